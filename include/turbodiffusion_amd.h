/*
 * turbodiffusion_amd.h — C-ABI of libturbodiffusion_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for the TurboDiffusion denoising hot path.  Every entry point takes
 * raw DEVICE pointers, sizes and an explicit HIP stream (void* == hipStream_t), never
 * allocates, never synchronises, and returns an int status (0 == TD_OK).  On failure the
 * reason is retrievable with td_last_error().  Unlike the reference's pybind module,
 * an unsupported shape is an ERROR, not a silent no-op (ops/gemm/launch.hpp:34-35).
 *
 * Reference interfaces replaced (paths relative to /root/reference/turbodiffusion):
 *   quant_cuda      ops/quant/quant.cu:28-75      -> td_quant_i8_block128
 *   gemm_cuda       ops/gemm/gemm.cu:27-72        -> td_gemm_w8a8
 *   rms_norm_cuda   ops/norm/rmsnorm.cu:12-59     -> td_rmsnorm
 *   layer_norm_cuda ops/norm/layernorm.cu:10-62   -> td_layernorm [+ fused modulate]
 *   Triton norms    ops/core.py:96-136,193-335    -> td_rmsnorm / td_layernorm
 *   AdaLN glue      rcm/networks/wan2pt1.py:404-413 -> td_layernorm [modulate], td_gated_residual
 *   rope_apply      rcm/networks/wan2pt1.py:156-178 -> td_qk_norm_rope
 *   SLA/utils.py get_block_map :55-67, mean_pool :43-52 -> td_seq_mean, td_sage_quant_pool, td_sla_topk
 *   spas_sage_attn.utils.get_vanilla_qk_quant (SLA/core.py:201-203) -> td_sage_quant_pool
 *   spas_sage_attn._qattn.qk_int8_sv_f16_*_block_sparse_attn (SLA/core.py:214)
 *                                              -> td_attn_i8   [sparse via LUT / dense]
 *   Triton _attn_fwd SLA/kernel.py:21-82       -> td_attn_16   [sparse via LUT / dense]
 *   SLA linear branch SLA/core.py:243-253      -> td_sla_linear_kv, td_sla_linear_out
 *   nn.Linear / F.linear in 16-bit (text MLP wan2pt1.py:678; C3's bf16 linears :226-232,375; umT5 rcm/utils/umt5.py:145-214;
 *     the VAE attention's products rcm/tokenizers/wan2pt1.py:229-248)      -> td_gemm_bf16, td_softmax_rows, td_t5_norm
 *
 * Tensor conventions: row-major contiguous unless a stride argument says otherwise;
 * 16-byte aligned base pointers; dtype codes below.
 */
#ifndef TURBODIFFUSION_AMD_H
#define TURBODIFFUSION_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TD_ABI_VERSION 5   /* round 6: + td_attn_16_qnorm_pieces, TD_TUNE_GEMM_W4; TD_SLA_NCH 32 -> 64 (workspace extents); default W8A8 dequant = one-VALU */

/* status codes */
#define TD_OK 0
#define TD_ERR_INVALID 1     /* bad argument (null pointer, negative size, ...) */
#define TD_ERR_UNSUPPORTED 2 /* shape / dtype this build cannot run (e.g. k % 128 != 0) */
#define TD_ERR_LAUNCH 3      /* HIP launch error */

/* dtype codes */
#define TD_F16 0
#define TD_BF16 1
#define TD_F32 2

/* GEMM epilogues */
#define TD_EPI_NONE 0
#define TD_EPI_GELU_TANH 1

typedef void* td_stream_t; /* hipStream_t */

int td_abi_version(void);
const char* td_last_error(void);

/* Kernel-selection knobs for benchmarking (results never depend on them: every variant of an
 * operator is bit-identical).  value 0 = automatic. */
#define TD_TUNE_GEMM_VARIANT 0 /* 1 = 128x128-tile kernel (default for m < 1024), 4 = 256x256 fine-interleaved on 16x16x64 MFMA (default),
                                  5 = the same pipeline on 32x32x32 MFMA (carries the opt-in fast dequant / schedules); 6 = variant 4 on 128 x 256 tiles
                                  (eight waves of 64 x 64), 7 = 4 or 6 by the launch planner, never mixed; 8 = the FOUR-wave form of variant 4
                                  (round 6: 128 x 256 tile, one wave per SIMD, two independent workgroups per CU); all exact variants are bit-identical */
#define TD_TUNE_GEMM_ABLATE 1  /* profiling instantiations only (s_memtime traces / phase stamps: 6, 9, 16, 19; tools/gemm_*.py) */
#define TD_TUNE_GEMM_LDPAD 2   /* profiling only: int8 operand row stride = k + value (buffers must be that large) */
#define TD_TUNE_GEMM_GROUP_M 3 /* m-tiles per raster group of the 256x256 kernels (0 = default 4) */
#define TD_TUNE_GEMM_SCHED 4   /* variant 5: bit 0 = barrier one chain earlier + refill spread over two chains, bit 1 = s_setprio for the
                                  younger half-workgroup (bit-identical results); variant 4: 1 = early LDS-DMA, 2 = L2 prefetch */
#define TD_TUNE_ATTN_TAU 5     /* attention: lazy running-max threshold in log2 units (0 = default 8, -1 = eager online softmax) */
#define TD_TUNE_GEMM_FAST 6     /* W8A8 GEMM dequant: 0 = build default (round 6: G = 8, the period instantiated for every epilogue and tile form; TD_GEMM_EXACT=1 in the environment makes it exact),
                                  1 = exact (bit-identical to the reference arithmetic, ops/gemm/utils.hpp:116-121),
                                  G in {2,4,8} = one-VALU dequant re-centred every G K blocks (|diff| <= 0.75 (G+1) sum_k s_k in the fp32
                                  accumulator: one bf16 rounding step on 1-8 % of the outputs).  bf16 + bias launches of the LDS-DMA kernels
                                  (m >= 1024: the model's linears); small problems and f16 / bias-free launches are always exact */
#define TD_TUNE_LIN_QB 7       /* linear branch, pass 2: Q blocks one workgroup walks (0 = default) */
#define TD_TUNE_ATTN_OCC 8     /* INT8/FP16-PV attention builds kept for comparison: 2 = two workgroups per CU with explicit fragment prefetch, 3 = Q64 (waves as 2 Q halves x 2 key halves; equal to rounding, not bit-identical), 4 = build 2 with the softmax denominator accumulated on the matrix pipe (round-4 experiment; equal to rounding), 5 = the production build with the denominator from the fp16-ROUNDED probabilities, two per v_dot2_f32_f16 (round-5 experiment: 16 instead of 32 VALU per lane and tile; measured equal, profiles/r05_attn_dot2.txt) */
#define TD_TUNE_VAE_CONV 9     /* td_vae_conv.  0 = default: the 2-D-tile kernel staged by LDS-DMA (csrc/vae_conv3.hip, frames-first tile order; 256-position tiles with two workgroups per CU, 512-position tiles for the 384-channel x 27-tap reductions) for the 3x3 spatial kernels with C_out % 96 == 0 or <= 32, the row-tile kernel (csrc/vae_conv.hip) for everything else; 8 / 9 = always 512 / 256 positions; 7 = 512 with the tiles of a frame first; 2 = the row-tile kernel everywhere (the default until round 4; cross-check); 1 = the first kernel (one gather per tap, flat position tiles), cross-check; 3 = row tiles of 512 columns, one workgroup per CU (experiment, slower), 4 = row tiles with 32-channel chunks in two LDS stages, 5 = row tiles, frames-first order, 6 = 4 with the chunk multiply unrolled (experiments: equal within 1 %) */
#define TD_TUNE_GEMM16 10      /* td_gemm_bf16, 16-bit outputs: 2 = the four-wave kernel (128x128 wave tiles, accumulators in AGPRs: round-4 experiment, measured equal to the default eight-wave kernel; bit-identical results) */
#define TD_TUNE_GELU_TABLE 11  /* td_gemm_w8a8_quant, bf16 + GELU-tanh (the FFN's first GEMM): 0 = the GELU of the fused epilogue as a lookup in the
                                  device-built 65 536-entry table of the GELU function of csrc/td_common.h: bit-identical, 1 = evaluated inline (cross-check, A/B) */
#define TD_TUNE_GEMM_COTENANT 12 /* 1 = the W8A8 GEMMs launched while this is set run BESIDE another GEMM on a second stream (the token-half
                                   split of a block's tail, wan.py): 256-row tiles only — the 128-row / mixed plans of gemm_w8a8_fi.hip price a
                                   launch as if it owned the 256 CUs, and half-height tiles cost 1.2x the matrix work per output (measured at
                                   N = 1 with the plans left on: -3.8 %, profiles/r05_gemm_plan_n1_ab.txt) */
#define TD_TUNE_GEMM_W4 13      /* round 6: which W8A8 GEMM launches take the FOUR-wave form (TD_TUNE_GEMM_VARIANT 8) when the variant is automatic —
                                   a bit mask over the epilogue kinds: 1 = fused output quantiser (ffn.0), 2 = V^T tiles (q|k|v), 4 = residual /
                                   row statistics (o, cross-q, cross-o, ffn.2), 8 = plain; 0 = the build default (csrc/gemm_w8a8_fi.hip), 16 = none */
#define TD_TUNE_COUNT 14
int td_set_tuning(int key, int value);
/* profiling: copy the n (<= 256) 64-bit s_memtime stamps of the last TD_TUNE_GEMM_ABLATE == 9 launch to host */
int td_debug_read(unsigned long long* host_dst, int n);
/* ---- f3: the two ends of the forward outside the blocks (csrc/embed_head.hip) ----
 * td_patch_embed: patchify "b c (t kt) (h kh) (w kw) -> b (t h w) (c kt kh kw)" + patch_embedding Linear
 *   (rcm/networks/wan2pt1.py:653-661; wan2pt2.py:644-645: y concatenated on channels = the second source x2 / c2).
 *   x [B, c1, T, Hin, Win], x2 [B, c2, T, Hin, Win] or NULL (c2 = 0), w [dim, (c1+c2)*4], bias [dim], all `dtype`
 *   (f16|bf16); y [B, rows, dim] = tokens [row0, row0 + rows) of every batch entry (a sequence-parallel rank's shard);
 *   fp32 accumulate, bias added in fp32, one rounding.  Patch (1, 2, 2); (c1+c2) % 4 == 0, <= 64; dim % 8 == 0.
 * td_head: Head.forward (wan2pt1.py:444-454) + unpatchify (:710-721): eager LayerNorm of x [B, rows, dim] (fp32 two-pass,
 *   rounded to `dtype`), fp32 modulate with scale / shift f32 [B, dim] (= e[1], e[0]), fp32 Linear w [out_dim*4, dim],
 *   bias [out_dim*4]; out f32 [B, out_dim, T, 2*Hh, 2*Ww] (unpatchify != 0; needs all tokens) or [B, rows, out_dim*4]. */
int td_patch_embed(const void* x, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t B, int64_t T, int64_t Hin,
                   int64_t Win, const void* w, const void* bias, void* y, int64_t dim, int64_t row0, int64_t rows,
                   td_stream_t stream);
int td_head(const void* x, int dtype, const float* scale, const float* shift, const float* w, const float* bias, float eps,
            float* out, int unpatchify, int64_t B, int64_t rows, int64_t dim, int64_t out_dim, int64_t T, int64_t Hh,
            int64_t Ww, int64_t row0, td_stream_t stream);

/* ---- f4: the convolutions of the Wan2.1 VAE decoder (csrc/vae_conv.hip; rcm/tokenizers/wan2pt1.py:37-55, 83-131, 177-209) ----
 * td_vae_conv: y = conv(x) + bias (+ res) on CHANNELS-LAST bf16 activations as one implicit GEMM on the bf16 matrix pipe.
 *   x [B, Ti, Hi, Wi, Ci] (batch stride in elements; Ci % 32 == 0), w [Co, kt*kh*kw*Ci] with K ordered (dt, dh, dw, c),
 *   bias [Co] or NULL, res like y or NULL; causal in time (kt - 1 zero frames on the left, CausalConv3d), symmetric
 *   zero padding in space; up2 != 0: nearest x2 up-sampling of H and W folded into the gather (Resample's
 *   Upsample + Conv2d, y is [B, Ti, 2Hi, 2Wi, Co]); interleave != 0: the time up-sampler's output mapping — channel n of
 *   frame t lands in frame 2t + n / (Co/2), channel n % (Co/2) (y is [B, 2Ti, Hi, Wi, Co/2]).  fp32 accumulate; bias added
 *   in fp32, rounded to bf16, residual added in bf16 (the reference's rounding points).
 * td_vae_chan_rms: RMS_norm over the channel axis of channels-last rows [rows, C] (C % 8 == 0, <= 512), optional SiLU,
 *   with every intermediate rounded to bf16 as the reference's bf16 path does (wan2pt1.py:69-70). */
int td_vae_conv(const void* x, int64_t x_batch_stride, const void* w, const void* bias, const void* res, void* y,
                int64_t y_batch_stride, int B, int Ti, int Hi, int Wi, int Ci, int Co, int kt, int kh, int kw, int up2,
                int interleave, td_stream_t stream);
/* td_vae_conv_ex: the same with an explicit output grid [To, Ho, Wo], output strides (1 | 2) and LEFT zero padding — the
 *   encoder's down-samplers (wan2pt1.py:98-102, 133-149): ZeroPad2d((0, 1, 0, 1)) + 3x3 stride-2 convolution =
 *   (stride_hw 2, pad_h = pad_w = 0, Ho = Hi / 2); the unpadded stride-2 (3,1,1) time convolution = (stride_t 2, pad_t 0,
 *   To = (Ti - 3) / 2 + 1).  Whatever the grid reaches beyond the right / bottom / last frame reads zero. */
int td_vae_conv_ex(const void* x, int64_t x_batch_stride, const void* w, const void* bias, const void* res, void* y,
                   int64_t y_batch_stride, int B, int Ti, int Hi, int Wi, int Ci, int Co, int kt, int kh, int kw, int up2,
                   int interleave, int To, int Ho, int Wo, int stride_t, int stride_hw, int pad_t, int pad_h, int pad_w,
                   td_stream_t stream);
int td_vae_chan_rms(const void* x, const void* gamma, void* y, int64_t rows, int C, int silu, td_stream_t stream);

/* ---- 16-bit GEMM, row softmax, T5LayerNorm (csrc/gemm_bf16.hip; ABI v3) ----
 * td_gemm_bf16: D[b][m, n] = sum_k A[b][m, k] * B[b][n, k] (+ bias[n]) (+ epilogue) on the 16-bit matrix pipe, fp32
 *   accumulate; `dtype` (TD_BF16 | TD_F16) = A, B, bias, res; `out_dtype` = dtype or TD_F32.  Row strides lda / ldb / ldd /
 *   ldr and batch strides in ELEMENTS (batch = 1: strides ignored); k % 64 == 0 (zero-pad shorter rows), lda, ldb % 8 == 0,
 *   16-bit outputs ldd % 8 == 0.  Rounding points = the reference's operator sequence (F.linear rounds, then each
 *   elementwise op rounds): cast(acc) -> + bias, cast -> epilogue, cast -> + res, cast; fp32 output: acc + bias, unrounded.
 *   epilogue: 0 none; 1 GELU-tanh (nn.GELU(approximate="tanh"), wan2pt1.py:375,678); 2 gated GELU of umT5's T5FeedForward
 *   (umt5.py:125-127,210: d [m, n/2] = fc1(x) * GELU(gate(x)) with B's rows = gate / fc1 interleaved in blocks of 32 rows
 *   — rows [64p, 64p+32) gate columns [32p, 32p+32), rows [64p+32, 64p+64) fc1 columns of the same range — and the 16-bit
 *   rounding of every elementwise step of the reference's explicit tanh formula); 3 exact (erf) GELU, nn.GELU() of Wan2.1 I2V's
 *   MLPProj (wan2pt1.py:462-466); res (16-bit, plain epilogue only): x + Linear.
 * td_softmax_rows: p[r, c] = softmax_c(scale * (s[r, c] (+ bias[r % bias_rows, c]))) in fp32, rounded once; columns
 *   [cols, ldp) of p are zero-filled (the next GEMM's k runs over the padded width).  s f32 or p's dtype (16-bit s: the
 *   bias add is rounded to that dtype first — `einsum(q, k) + attn_bias`, umt5.py:183); in place when s == p and lds == ldp.
 * td_t5_norm: T5LayerNorm (umt5.py:130-142): cast(x * rsqrt(mean(float(x)^2) + eps)) then w * that in the 16-bit dtype. */
int td_gemm_bf16(const void* a, const void* b, const void* bias, const void* res, void* d, int dtype, int out_dtype, int epilogue,
                 int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldd, int64_t ldr, int64_t batch,
                 int64_t stride_a, int64_t stride_b, int64_t stride_d, int64_t stride_r, td_stream_t stream);
/* split-K companion of td_gemm_bf16 for small m (a few dozen 256x256 tiles cannot pull the weights through the chip): run the
 * GEMM as a batch of `splits` K-slices with fp32 outputs ws [splits, m, n] (td_gemm_bf16 with out_dtype TD_F32, batch strides =
 * k / splits columns), then this pass adds the slices in fp32 and applies bias / epilogue / residual with td_gemm_bf16's
 * rounding points. */
int td_gemm_bf16_splitk_reduce(const float* ws, int splits, const void* bias, const void* res, void* d, int dtype, int epilogue,
                               int64_t m, int64_t n, int64_t ldd, int64_t ldr, td_stream_t stream);
int td_softmax_rows(const void* s, int s_dtype, void* p, int p_dtype, const void* bias, int64_t rows, int64_t cols, int64_t lds,
                    int64_t ldp, int64_t bias_rows, int64_t ldb, float scale, td_stream_t stream);
int td_t5_norm(const void* x, const void* w, void* y, int dtype, float eps, int64_t rows, int64_t n, int64_t ldx, int64_t ldy,
               td_stream_t stream);

/* time embedding (wan2pt1.py:144-153, 671-674) and the AdaLN vectors of all blocks:
 * td_time_sinusoid: t [B] (`dtype`, the bf16-rounded timesteps) -> out f32 [B, freq_dim] = cat(cos, sin)(t * 10000^(-j/half)), fp64;
 * td_gemv_f32: out f32 [B, N] = act(x f32 [B, K]) @ float(w [N, K])^T + float(bias [N]), act = SiLU when silu_input
 *   (the three Linears of time_embedding / time_projection in their fp32 island; w, bias 16-bit `dtype`); B <= 64;
 * td_bcast_add: out f32 [A, B, R, D] = m [A, R, D] + e [B, RE, D] (RE == R or 1): (modulation + e0) of every block at once
 *   (wan2pt1.py:400) and the head's (modulation + e) (:452). */
int td_time_sinusoid(const void* t, int dtype, float* out, int64_t B, int64_t freq_dim, td_stream_t stream);
int td_gemv_f32(const float* x, const void* w, const void* bias, int dtype, int silu_input, float* out, int64_t B, int64_t N,
                int64_t K, td_stream_t stream);
int td_bcast_add(const float* m, const float* e, float* out, int64_t A, int64_t B, int64_t R, int64_t RE, int64_t D,
                 td_stream_t stream);

/* a18: one update of the few-step rCM sampler on the fp64 state (inference/wan2.1_t2v_infer.py:134-139; the ODE form
 * wan2.2_i2v_infer.py:202-203 when eps == NULL), the reference's fp64 operator sequence operation for operation, fused with
 * the cast of the next step's network input: x f64 [n] in place; v f32 [n] (the network's velocity); eps f32 [n] N(0,1) or
 * NULL; x16 (optional) 16-bit [n] = x.to(dtype16) as torch casts a double (through float). */
int td_rcm_step(double* x, const float* v, const float* eps, void* x16, int dtype16, double t_cur, double t_next, int64_t n,
                td_stream_t stream);

/* measurement support (csrc/calib.hip; bench.py's "box" calibration — no reference counterpart: the reference ships no
 * benchmark code).  td_calib_mfma_i8: blocks x 256 threads, every wave issues iters x 4 v_mfma_i32_32x32x32_i8 (2*32^3 ops
 * each); td_calib_hbm_read: one streaming pass of 16-byte non-temporal loads over src[0, bytes); td_calib_clock_probe: one
 * wave writes {s_memtime start, end, s_memrealtime (100 MHz) start, end} (4 x u64, device) around a wait of ticks_100mhz. */
int td_calib_mfma_i8(int iters, int blocks, float* sink, td_stream_t stream);
int td_calib_hbm_read(const void* src, int64_t bytes, void* sink, td_stream_t stream);
int td_calib_clock_probe(int64_t ticks_100mhz, void* out, td_stream_t stream);

/* ---- a16: per-128x128-block INT8 quantiser (quant_cuda, ops/quant/quant.cu:28-71) ----
 * x [m,n] f16|bf16 -> q [m,n] int8, s [ceil(m/128), ceil(n/128)] f32.
 * amax = max(1e-8, max|x|) over the block's valid elements; q = sat_s8(rne(x*(128/amax)));
 * s = amax/128 (quant.hpp:91-98). Requires n % 8 == 0. */
int td_quant_i8_block128(const void* x, int dtype, int8_t* q, float* s, int64_t m, int64_t n,
                         td_stream_t stream);

/* ---- a17: block-scaled W8A8 GEMM (gemm_cuda, ops/gemm/gemm.cu:27-68) ----
 * d[m,n] = cast( sum_kb fma(float(sum_{k in kb} a[m,k]*b[n,k]), a_s[m/128,kb]*b_s[n/128,kb], acc) )
 * then, matching Int8Linear.forward (ops/core.py:408-412): + bias (rounded again in the
 * output dtype) and optionally GELU-tanh (the FFN activation, wan2pt1.py:375).
 * a [m,k] int8, a_s [ceil(m/128), k/128] f32, b [n,k] int8, b_s [ceil(n/128), k/128] f32,
 * bias [n] (out dtype) or NULL, d [m,n] f16|bf16 with row stride ldd (elements).
 * Requires k % 128 == 0 and n % 8 == 0; otherwise TD_ERR_UNSUPPORTED. */
int td_gemm_w8a8(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                 const void* bias, void* d, int out_dtype, int epilogue, int64_t m, int64_t n,
                 int64_t k, int64_t ldd, td_stream_t stream);

/* ---- a15 -> a16 fused: the same GEMM whose epilogue block-quantises its own (16-bit rounded) result for the next
 * Int8Linear: d_q int8 [m,n], d_s f32 [ceil(m/128), ceil(n/128)] == td_quant_i8_block128(td_gemm_w8a8(...)) bit for
 * bit (Int8Linear.forward -> int8_quant of the next Int8Linear, ops/core.py:408-412,12-25; the FFN pair
 * wan2pt1.py:375).  act_dtype = the 16-bit dtype the intermediate is rounded to.  Requires k % 128 == 0, n % 16 == 0. */
int td_gemm_w8a8_quant(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                       const void* bias, int8_t* d_q, float* d_s, int act_dtype, int epilogue,
                       int64_t m, int64_t n, int64_t k, td_stream_t stream);

/* ---- a15 -> a7 fused: the same GEMM whose epilogue applies the block's gated residual in place:
 * x[m,n] = x[m,n] + cast(cast(gemm+bias)[m,n] * cast(gate[n]))   (gate f32 [n], or NULL for a plain add; x f16|bf16 with
 * row stride ldx) == td_gated_residual(x, td_gemm_w8a8(...), gate) bit for bit (Int8Linear.forward ops/core.py:408-412
 * followed by `x + y * e.type_as(x)`, wan2pt1.py:405-406,412-413).  Requires k % 128 == 0, n % 8 == 0, ldx % 8 == 0. */
int td_gemm_w8a8_residual(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                          const void* bias, void* x, const float* gate, int dtype, int64_t m,
                          int64_t n, int64_t k, int64_t ldx, td_stream_t stream);

/* ---- a5: RMSNorm over the last dim (ops/core.py:139-191; rms_norm_cuda) ----
 * y = cast((x*rsqrt(mean(x^2)+eps))*w), fp32 math. x [m,n] in_dtype (f32|bf16|f16),
 * w [n] f32, y [m,n] out_dtype. Requires n % 8 == 0, n <= 8192. */
int td_rmsnorm(const void* x, int in_dtype, const float* w, void* y, int out_dtype, float eps,
               int64_t m, int64_t n, td_stream_t stream);

/* ---- a6 + a7: LayerNorm (+ fused AdaLN modulate) (ops/core.py:380-386; wan2pt1.py:404,411) ----
 * xn = cast_out((x-mean)*rstd [*w + b])            (w,b f32 [n] or NULL,NULL)
 * if scale != NULL: y = cast_out(float(xn)*(1+scale[bi]) + shift[bi]) with
 *   bi = row / rows_per_batch, scale/shift f32 [batch, n]   (the two roundings of the reference)
 * x [m,n] in_dtype, y [m,n] out_dtype. Requires n % 8 == 0, n <= 8192.
 * pad_cols: 0 = the textbook two-pass variance (the CUDA twin layer_norm_cuda, ops/norm/layernorm.hpp:68-77, and the
 *   eager WanLayerNorm).  The reference's TRITON LayerNorm — what FastLayerNorm.forward / ops.layernorm actually run
 *   (ops/core.py:380-386 -> :193-242, :293-335) — loads next_power_of_2(n) columns with the masked ones as 0 and sums
 *   (x - mean)^2 over ALL of them (:213-224, :313-324): var = (sum_valid (x-mean)^2 + pad_cols * mean^2) / n with
 *   pad_cols = next_power_of_2(n) - n (512 at n = 1536, 3072 at n = 5120).  Found by running those kernels on the
 *   MI355X (tests/golden/triton_leaves.pt); pass that pad_cols to reproduce the reference's shipped arithmetic. */
int td_layernorm(const void* x, int in_dtype, const float* w, const float* b, const float* scale,
                 const float* shift, int64_t rows_per_batch, void* y, int out_dtype, float eps, int64_t pad_cols,
                 int64_t m, int64_t n, td_stream_t stream);

/* ---- a6 + a7 -> a16 fused: LayerNorm (+ affine, + AdaLN modulate) whose 16-bit result is block-quantised for the
 * Int8Linear that consumes it: q int8 [m,n], qs f32 [ceil(m/128), ceil(n/128)] == td_quant_i8_block128(td_layernorm(...))
 * bit for bit (x, and the intermediate, in `dtype` = f16|bf16).  stats_ws: caller-owned scratch of 2*m floats (the rows'
 * mean / rstd between the two passes).  Requires n % 8 == 0, n <= 8192. */
int td_layernorm_quant(const void* x, int dtype, const float* w, const float* b, const float* scale,
                       const float* shift, int64_t rows_per_batch, int8_t* q, float* qs, float* stats_ws, float eps,
                       int64_t pad_cols, int64_t m, int64_t n, td_stream_t stream);

/* ---- a15 (+ a7) -> a5 / a6 statistics: row statistics of the NEXT norm from the producing GEMM's epilogue ----
 * td_gemm_w8a8_stats: td_gemm_w8a8 (residual == 0: d_or_x = d with row stride ld) or td_gemm_w8a8_residual (residual != 0:
 *   in place on x, gate f32 [n] or NULL) — same arithmetic, same bits — whose epilogue also writes, per output row and per
 *   64-column piece of it, (mean, M2 = sum of squared deviations from that mean) of the 16-bit values it stores:
 *   stats_ws float2 [m, n/64].  bf16, bias required, k % 128 == 0, n % 64 == 0, ld % 8 == 0.
 * td_row_stats_finalize: the pieces merged pairwise-update style (never E[x^2] - mean^2: rows with |mean| >> spread keep
 *   their digits, like the reference's two-pass statistics) -> mode 0: LayerNorm statistics out float2 [m] =
 *   (mean, 1/sqrt(var + eps)), var = (M2 + pad_cols * mean^2) / n (pad_cols: see td_layernorm);
 *   mode 1: RMSNorm out float [m] = 1/sqrt(E[x^2] + eps)   (== td_rms_stats up to summation order).  n == 64 * pieces.
 * td_layernorm_quant_stats: td_layernorm_quant's apply + quantise pass with the rows' (mean, rstd) supplied.
 * Together they remove the statistics pass of WanLayerNorm / the cross-attention q RMSNorm over [L, dim]
 * (wan2pt1.py:191-212,404-413): the residual stream is read once per LayerNorm instead of twice. */
int td_gemm_w8a8_stats(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                       void* d_or_x, const float* gate, int residual, int dtype, int64_t m, int64_t n, int64_t k,
                       int64_t ld, float* stats_ws, td_stream_t stream);

/* td_gemm_w8a8 of a fused q|k|v projection (wan2pt1.py:256-262 computes the three Linears; a15 each) whose V columns
 * [v_col0, n) leave the kernel as the V^T MFMA tiles of td_v_transpose — [head][ceil(m/64)][128][64], vt_dtype f16 (the cast
 * of SLA/core.py:213 `v.to(torch.float16)`) or the output dtype — instead of row-major: bit-identical to td_gemm_w8a8 followed
 * by td_v_transpose on those columns; d's V columns are NOT written.  bias required; n and v_col0 multiples of 256. */
int td_gemm_w8a8_vt(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias, void* d,
                    int out_dtype, int64_t m, int64_t n, int64_t k, int64_t ldd, int64_t v_col0, void* vt, int vt_dtype,
                    td_stream_t stream);
int td_row_stats_finalize(const float* ws, int pieces, int64_t n, float eps, int64_t pad_cols, int mode, float* out,
                          int64_t m, td_stream_t stream);
int td_layernorm_quant_stats(const void* x, int dtype, const float* w, const float* b, const float* scale,
                             const float* shift, int64_t rows_per_batch, int8_t* q, float* qs, const float* row_stats,
                             int64_t m, int64_t n, td_stream_t stream);

/* ---- a7: gated residual  x = x + y*gate.type_as(x)  (wan2pt1.py:405-406,412-413) ----
 * x,y [m,n] f16|bf16 (in place on x), gate f32 [batch,n] or NULL (plain x += y, :410).
 * Rounding order of the reference: t = round(y*round(gate)); x = round(x + t). */
int td_gated_residual(void* x, const void* y, const float* gate, int64_t rows_per_batch,
                      int dtype, int64_t m, int64_t n, td_stream_t stream);

/* ---- a3 + a8: q/k RMSNorm over the full model dim + interleaved RoPE + head-major relayout ----
 * (WanSelfAttention.forward wan2pt1.py:261-269; FastRMSNorm ops/core.py:441-442; rope_apply :156-178)
 * src [L, >=dim] 16-bit with row stride ld_src (elements) -> dst [H, L, D] (dim = H*D).
 * xn = cast(rmsnorm(x)*w); if cos != NULL: (x0,x1) -> cast(x0*c - x1*s, x0*s + x1*c) with
 * cos/sin f32 [L, D/2].  w may be NULL (no norm: plain relayout, used for V / cross K,V). */
int td_qk_norm_rope(const void* src, int64_t ld_src, const float* w, const float* cosv,
                    const float* sinv, void* dst, int dtype, float eps, int64_t L, int H, int D,
                    td_stream_t stream);

/* ---- V tile transpose for the PV MFMA: v 16-bit, element (h,l,d) at v + h*stride_h + l*stride_l + d
 * (so both [H,L,D] and a column slice of a [L, 3*dim] GEMM output work) ->
 * vt [H, ceil(L/64), D, 64] out_dtype, keys of each 16-group stored in MFMA operand order
 * (0-3,8-11,4-7,12-15), tail keys zero-filled.  Replaces v.to(float16) (SLA/core.py:213)
 * and spas_sage_attn._fused.transpose_pad_permute_cuda (:221). D must be 128. */
int td_v_transpose(const void* v, int in_dtype, int64_t stride_h, int64_t stride_l, void* vt,
                   int out_dtype, int64_t L, int H, int D, td_stream_t stream);

/* ---- a11/a13: per-head sequence mean of K  (k.mean(dim=-2), SLA/core.py:197, utils.py:56) ----
 * k [H, L, D] 16-bit -> km [H, D] same dtype (fp32 accumulate, one rounding). D == 128.
 * ws: f32 workspace [H, 64, D] (deterministic two-stage sum, no atomics). */
int td_seq_mean(const void* k, void* km, float* ws, int dtype, int64_t L, int H, int D,
                td_stream_t stream);
/* the two stages of td_seq_mean, exposed for sequence parallelism: each rank computes the 64 partial
 * sums of ITS tokens (ws [H, 64, D]); after an all-gather the final stage sums `nch` partials found at
 * ws + h*stride_h + c*stride_c (+d) in order, divides by the GLOBAL length and rounds once. */
int td_seq_sum_partial(const void* k, float* ws, int dtype, int64_t L, int H, int D, td_stream_t stream);
int td_seq_mean_final(const float* ws, int nch, int64_t stride_h, int64_t stride_c, void* km, int dtype,
                      int64_t L_total, int H, int D, td_stream_t stream);

/* ---- a11 + a13 fused pass over Q or K [H, L, D] (D == 128):
 *  - block mean pool (mean_pool, SLA/utils.py:43-52) of x (Q) or of cast(x - km) (K, smooth-K
 *    utils.py:56) with block `pool_blk`, fp32 sum / valid count, cast to dtype -> pooled [H, nb, D]
 *  - per-block INT8 quantisation (get_vanilla_qk_quant, SLA/core.py:201-203) with block
 *    `pool_blk`: xf = float(x) - float(km); scale = max|xf|/127 + 1e-7;
 *    q = trunc(xf/scale + 0.5*sign)  -> xq [H, L, D] int8, xs [H, nb] f32.
 * km may be NULL (Q). pooled or xq/xs may be NULL to skip that output.
 * pool_blk in {64, 128}. */
int td_sage_quant_pool(const void* x, const void* km, int dtype, int pool_blk, void* pooled,
                       int8_t* xq, float* xs, int64_t L, int H, int D, td_stream_t stream);

/* ---- a11: pooled score + top-k block selection (SLA/utils.py:59-66) ----
 * score[h,i,j] = cast(sum_d pq[h,i,d]*pk[h,j,d]) (16-bit rounding like the reference's
 * bf16 matmul); select the `topk` largest per row (ties -> lower index), write ascending
 * block ids to lut [H, Qb, topk] int32. Kb <= 2048. pk is [H, Kb_alloc, D] (Kb_alloc >= Kb rows
 * allocated per head; 0 means Kb). */
int td_sla_topk(const void* pq, const void* pk, int dtype, int32_t* lut, int H, int Qb, int Kb,
                int Kb_alloc, int D, int topk, td_stream_t stream);

/* ---- a13 / a9: SageAttention INT8-QK, FP16-PV, fp32 softmax+accumulate ----
 * q_i8 [H, L, 128] int8, q_s [H, ceil(L/128)], k_i8 [H, Lk, 128] int8, k_s [H, ceil(Lk/64)],
 * vt from td_v_transpose (f16) [H, ceil(Lk/64), 128, 64].
 * lut [H, Qb, nsel] ascending K-block ids, or NULL for dense attention (all K blocks).
 * o: 16-bit, element (h, l, d) at o + h*o_stride_h + l*o_stride_l + d.
 * sm_scale: softmax scale (D^-0.5).
 * Lk_alloc: K rows ALLOCATED per head in k_i8 (and Lk_alloc/64 blocks per head in k_s / vt); 0 = Lk.
 * (> Lk for the rank-padded gathered layout of sequence parallelism; must then be a multiple of 64.) */
int td_attn_i8(const int8_t* q_i8, const float* q_s, const int8_t* k_i8, const float* k_s,
               const void* vt, const int32_t* lut, int nsel, void* o, int out_dtype,
               int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk,
               int64_t Lk_alloc, int H, td_stream_t stream);

/* The same kernel with its two optional epilogue fusions (both leave the attention arithmetic untouched):
 *   add_t   : 16-bit addend in the lane-private layout written by td_sla_linear_out_t; the output becomes
 *             o_s + o_l (the 16-bit add of SLA/core.py:253) without a separate read-modify-write pass over o;
 *   q_out / q_scale : instead of o, write the [L, H*128] attention output block-quantised for the Int8Linear o
 *             projection: q_out int8 [L, H*128], q_scale f32 [ceil(L/128), H] == td_quant_i8_block128 of the
 *             16-bit output, bit for bit (a workgroup's 128-token x one-head tile is exactly one 128x128 block).
 * Without q_out the output strides must be multiples of 8 elements. */
int td_attn_i8_ex(const int8_t* q_i8, const float* q_s, const int8_t* k_i8, const float* k_s,
                  const void* vt, const int32_t* lut, int nsel, void* o, int out_dtype,
                  int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk,
                  int64_t Lk_alloc, int H, const void* add_t, int8_t* q_out, float* q_scale,
                  td_stream_t stream);

/* ---- a13, FP8-PV variant (the reference's sm89+ branch, SLA/core.py:217-239; SpargeAttn's
 * fused.transpose_pad_permute_cuda + fused.scale_fuse_quant_cuda and qk_int8_sv_f8_*_fuse_v_scale_* kernels) ----
 * td_v_fp8_tiles: V (element (h,l,d) at v + h*stride_h + l*stride_l + d, f16|bf16) ->
 *   v_scale f32 [H, 128] = max_l |v[h,l,d]| / scale_max (the reference passes 2.25), and
 *   vt8 [H, ceil(L/64), 128, 64] OCP e4m3 = e4m3(v / v_scale), keys of a block in the position order of the PV MFMA's
 *   B operand (position 32hi + 16g + r holds key 32g + (r&3) + 8(r>>2) + 4hi), tail keys zero.
 *   ws: f32 scratch [H, 64, 128] (partial maxima).
 * td_attn_i8_fp8pv: td_attn_i8_ex with P rounded to e4m3 and P.V on the fp8 MFMA (fp32 accumulate), the V channel
 *   scale applied in the epilogue. */
int td_v_fp8_tiles(const void* v, int in_dtype, int64_t stride_h, int64_t stride_l, uint8_t* vt8, float* v_scale,
                   float* ws, float scale_max, int64_t L, int H, int D, td_stream_t stream);
int td_attn_i8_fp8pv(const int8_t* q_i8, const float* q_s, const int8_t* k_i8, const float* k_s,
                     const uint8_t* vt8, const float* v_scale, const int32_t* lut, int nsel, void* o, int out_dtype,
                     int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk,
                     int64_t Lk_alloc, int H, const void* add_t, int8_t* q_out, float* q_scale,
                     td_stream_t stream);

/* ---- a12 / a9 / a4: 16-bit QK attention (SLA Triton arithmetic; dense cross-attention) ----
 * q [H, L, 128], k [H, Lk, 128] dtype (bf16|f16); vt [H, ceil(Lk/64), 128, 64] same dtype;
 * P is rounded to dtype before P@V (SLA/kernel.py:68). lut as above. */
int td_attn_16(const void* q, const void* k, const void* vt, const int32_t* lut, int nsel, void* o,
               int dtype, int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L,
               int64_t Lk, int64_t Lk_alloc, int H, td_stream_t stream);

int td_attn_16_ex(const void* q, const void* k, const void* vt, const int32_t* lut, int nsel, void* o,
                  int dtype, int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L,
                  int64_t Lk, int64_t Lk_alloc, int H, const void* add_t, int8_t* q_out, float* q_scale,
                  td_stream_t stream);

/* td_attn_16_ex with Q taken straight from a [L, ld_q] linear output (head h = columns [128h, 128h+128)) and its
 * full-width RMSNorm applied on load: q = cast(x * q_rstd[l] * q_w[c]) — td_qk_norm_rope without RoPE (cross-attention
 * Q, wan2pt1.py:289) minus the 4-byte-per-element round trip of the normalised copy.  q_rstd f32 [L] from td_rms_stats
 * (1/sqrt(mean(x^2)+eps) over the n = H*128 columns, same summation order as td_qk_norm_rope), q_w f32 [H*128]. */
int td_rms_stats(const void* src, int64_t ld_src, int dtype, float* rstd, float eps, int64_t L, int64_t n,
                 td_stream_t stream);
int td_attn_16_qnorm(const void* q_src, int64_t ld_q, const float* q_rstd, const float* q_w, const void* k,
                     const void* vt, const int32_t* lut, int nsel, void* o, int dtype, int64_t o_stride_h,
                     int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk, int64_t Lk_alloc, int H,
                     const void* add_t, int8_t* q_out, float* q_scale, td_stream_t stream);
/* round 6 (ABI v5): the same with the row statistic formed on load from the per-64-column pieces of td_gemm_w8a8_stats
 * (stats_ws float2 [L, pieces], pieces = H * 128 / 64): == td_row_stats_finalize(mode 1) + td_attn_16_qnorm bit for bit. */
int td_attn_16_qnorm_pieces(const void* q_src, int64_t ld_q, const float* stats_ws, int pieces, float eps, const float* q_w,
                            const void* k, const void* vt, const int32_t* lut, int nsel, void* o, int dtype, int64_t o_stride_h,
                            int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk, int64_t Lk_alloc, int H,
                            const void* add_t, int8_t* q_out, float* q_scale, td_stream_t stream);

/* ---- sequence parallelism: the same attention / block-map kernels reading the K side STRAIGHT from the
 * output of the per-layer all-gather, which is rank-major: K block j (64 keys) is block j % kb_per_rank of rank
 * j / kb_per_rank.  k / k_s / vt / pk point at rank 0's part ([H, kb_per_rank*64, 128], [H, kb_per_rank],
 * [H, kb_per_rank, 128, 64], [H, kb_per_rank, D]); the *_rank_stride arguments are the distances to the next rank's part
 * (k, vt: bytes, multiples of 16; k_s: floats; pk: elements).  No re-layout of the gathered state is needed.
 * (Replaces the role of the sequence all-to-alls around local attention, rcm/utils/a2a_cp.py:146-182.) */
int td_attn_i8_sp(const int8_t* q_i8, const float* q_s, const int8_t* k_i8, const float* k_s, const void* vt,
                  const int32_t* lut, int nsel, void* o, int out_dtype, int64_t o_stride_h, int64_t o_stride_l,
                  float sm_scale, int64_t L, int64_t Lk, int H, int kb_per_rank, int64_t k_rank_stride,
                  int64_t ks_rank_stride, int64_t vt_rank_stride, const void* add_t, int8_t* q_out, float* q_scale,
                  int q_heads_total, td_stream_t stream);
int td_attn_16_sp(const void* q, const void* k, const void* vt, const int32_t* lut, int nsel, void* o, int dtype,
                  int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk, int H,
                  int kb_per_rank, int64_t k_rank_stride, int64_t vt_rank_stride, const void* add_t, int8_t* q_out,
                  float* q_scale, int q_heads_total, td_stream_t stream);
/* (ABI v4) q_heads_total: the head count of the WHOLE [L, heads * 128] quantised output row when this launch covers a head
 * GROUP of it (q_out / q_scale then point at the group's first head: q_out + h0 * 128, q_scale + h0); 0 = H.
 *
 * Round-5 entry points of the sequence-parallel layer (fewer launches in front of its latency-bound exchanges):
 *  td_seq_sum            per-head column sums of k [H, L, 128] -> out f32 [H, 128]: td_seq_sum_partial's 64 row chunks per head + a
 *                        128-thread pass per head that adds the chunk partials in order (two launches; a one-launch form with an
 *                        atomic ticket cost 10x more: a device-scope fence per workgroup flushes the XCD's L2 — csrc/sla_prep.hip).
 *                        ws f32 [H, 64, 128] scratch.
 *  td_qk_norm_rope_pair  td_qk_norm_rope for the q AND the k columns of a fused projection in one launch (bit-identical).
 *  td_sage_quant_pool_packed_kmsum
 *                        td_sage_quant_pool_packed with the smooth-K mean either given (km) or formed in the kernel from km_n (<= 8)
 *                        per-rank column sums (km_parts: the gathered td_seq_sum outputs, km_stride floats apart) over km_rows
 *                        global rows — the arithmetic of td_seq_mean_final, without its launch — and a head layout of its own
 *                        (pool_hg, pool_gs_bytes; hg = 0: flat) for `pooled`. */
int td_seq_sum(const void* k, float* ws, float* out, int dtype, int64_t L, int H, int D, td_stream_t stream);
int td_qk_norm_rope_pair(const void* src_q, const void* src_k, int64_t ld_src, const float* w_q, const float* w_k,
                         const float* cosv, const float* sinv, void* dst_q, void* dst_k, int dtype, float eps, int64_t L, int H,
                         int D, td_stream_t stream);
int td_sage_quant_pool_packed_kmsum(const void* x, const void* km, const float* km_parts, int km_n, int64_t km_stride, int64_t km_rows,
                                    int dtype, int pool_blk, void* pooled, int pool_hg, int64_t pool_gs_bytes, int8_t* xq, float* xs,
                                    int64_t L, int64_t L_alloc, int hg, int64_t gs_bytes, int H, int D, td_stream_t stream);
int td_sla_topk_sp(const void* pq, const void* pk, int dtype, int32_t* lut, int H, int Qb, int Kb, int kb_per_rank,
                   int64_t pk_rank_stride, int D, int topk, td_stream_t stream);

#define TD_SLA_NCH 64 /* partial-sum chunks per head of td_sla_linear_kv* (workspace leading extent) */

/* ---- a14: linear-attention branch (SLA/core.py:243-253, feature_map = softmax) ----
 * pass 1: ck = cast(softmax_D(k)); kvsum[h] = cast(ck^T @ v) ; ksum[h] = cast(sum_L ck)
 *   k [H, L, D] dtype; vt = the V^T tiles of td_v_transpose (vt_dtype); ws_kv f32 [H,TD_SLA_NCH,D,D] and
 *   ws_ks f32 [H,TD_SLA_NCH,D] are scratch (partials summed in order: deterministic);
 *   outputs kvsum_t [H, D(d2), D(d1)] dtype (TRANSPOSED: the A operand of pass 2), ksum [H, D].
 *   Optional (both or neither NULL): ws_km f32 [H,TD_SLA_NCH,D] scratch + km [H, D] dtype = the smooth-K mean of
 *   td_seq_mean, accumulated during the same pass over K (so block-sparse SLA needs no separate td_seq_mean). */
int td_sla_linear_kv(const void* k, int dtype, const void* vt, int vt_dtype, float* ws_kv,
                     float* ws_ks, void* kvsum_t, void* ksum, float* ws_km, void* km, int64_t L, int H, int D,
                     td_stream_t stream);
/* the two stages of pass 1 (sequence parallelism: partial on each rank's tokens, all-gather, final).
 * final sums `nch` partials found at ws_kv + h*kv_stride_h + c*kv_stride_c (+i) [same for ks] and writes
 * out_dtype f16|bf16: kvsum_t TRANSPOSED + rounded, ksum rounded;  out_dtype f32: the un-rounded,
 * un-transposed fp32 sums [H,D,D] / [H,D] (a rank's contribution to the gather). */
int td_sla_linear_kv_partial(const void* k, int dtype, const void* vt, int vt_dtype, float* ws_kv,
                             float* ws_ks, int64_t L, int H, int D, td_stream_t stream);
int td_sla_linear_kv_final(const float* ws_kv, const float* ws_ks, int nch, int64_t kv_stride_h,
                           int64_t kv_stride_c, int64_t ks_stride_h, int64_t ks_stride_c, void* kv_out,
                           void* ks_out, int out_dtype, int H, int D, td_stream_t stream);
/* pass 2: cq = cast(softmax_D(q)); o_l = cast(cast(cq@kvsum)/cast(1e-5 + cast(sum_D cast(cq*ksum))));
 *   o_l = cast(o_l @ cast(Wp)^T + cast(bp)) (proj_l under autocast); o[h,l,:] = cast(o[h,l,:] + o_l)
 *   (o addressed with the same strides as td_attn_*; updated in place).
 *   q [H, L, D] dtype; Wp [D, D] f32 (nn.Linear weight [out,in]), bp [D] f32. */
int td_sla_linear_out(const void* q, int dtype, const void* kvsum_t, const void* ksum,
                      const float* wp, const float* bp, void* o, int64_t o_stride_h,
                      int64_t o_stride_l, int64_t L, int H, int D, td_stream_t stream);
/* The linear branch with the reference's other feature maps (SLA/core.py:57-64; ABI v3): feature_map 0 = softmax over D
 * (== td_sla_linear_kv / td_sla_linear_out), 1 = elu(x) + 1, 2 = relu — elementwise, with torch's 16-bit rounding after
 * every op.  Same buffers and layouts as the softmax entry points. */
int td_sla_linear_kv_fm(const void* k, int dtype, const void* vt, int vt_dtype, float* ws_kv, float* ws_ks, void* kvsum_t,
                        void* ksum, int feature_map, int64_t L, int H, int D, td_stream_t stream);
int td_sla_linear_out_fm(const void* q, int dtype, const void* kvsum_t, const void* ksum, const float* wp, const float* bp,
                         void* o, int64_t o_stride_h, int64_t o_stride_l, int feature_map, int64_t L, int H, int D,
                         td_stream_t stream);

/* pass 2 without the read-modify-write: o_l = cast(proj_l(...)) is written to t_out, 16-bit, in the lane-private
 * layout [H][ceil(L/128)][4 waves][16][64 lanes][4] that td_attn_*_ex consumes through add_t (run it BEFORE the
 * attention kernel: it only needs q, kvsum, ksum). */
int td_sla_linear_out_t(const void* q, int dtype, const void* kvsum_t, const void* ksum, const float* wp,
                        const float* bp, void* t_out, int64_t L, int H, int D, td_stream_t stream);


/* ---- sequence parallelism, producer side: write straight into the all-gather's send buffer ("pack") ----
 * The K-side state a rank sends per self-attention layer (seqpar.py; reference exchange: rcm/utils/a2a_cp.py:146-182) is
 * ONE buffer [groups][section][heads_per_group][...]: a group of heads is one contiguous range = one all-gather, so that
 * attention on the first heads runs under the later groups' transfers.  The *_packed entry points are the flat ones above
 * with the per-head OUTPUT (td_sla_linear_kv_partial_packed: the vt INPUT) placement generalised: head h of a section lives
 * at  section_base + (h / hg) * gs_bytes + (h % hg) * (that section's per-head bytes at L_alloc rows);  L_alloc >= L rows
 * (and ceil(L_alloc/blk) blocks, ceil(L_alloc/64) V^T tiles) are ALLOCATED per head — the rank-padded shard size — while L
 * rows are produced.  hg = 0 (with gs_bytes = 0, L_alloc = L) is exactly the flat call.  H % hg == 0; gs_bytes % 16 == 0. */
int td_sage_quant_pool_packed(const void* x, const void* km, int dtype, int pool_blk, void* pooled, int8_t* xq, float* xs,
                              int64_t L, int64_t L_alloc, int hg, int64_t gs_bytes, int H, int D, td_stream_t stream);
int td_v_transpose_packed(const void* v, int in_dtype, int64_t stride_h, int64_t stride_l, void* vt, int out_dtype,
                          int64_t L, int64_t L_alloc, int hg, int64_t gs_bytes, int H, int D, td_stream_t stream);
int td_sla_linear_kv_partial_packed(const void* k, int dtype, const void* vt, int vt_dtype, float* ws_kv, float* ws_ks,
                                    int64_t L, int64_t L_alloc, int hg, int64_t gs_bytes, int H, int D, td_stream_t stream);
int td_sla_linear_kv_final_packed(const float* ws_kv, const float* ws_ks, int nch, int64_t kv_stride_h, int64_t kv_stride_c,
                                  int64_t ks_stride_h, int64_t ks_stride_c, void* kv_out, void* ks_out, int out_dtype, int hg,
                                  int64_t gs_bytes, int H, int D, td_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TURBODIFFUSION_AMD_H */
