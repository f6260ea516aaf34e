// a17 (large-M path, v4) — block-scaled W8A8 INT8 GEMM, 256x256 tile, LDS-DMA staging, FINE-INTERLEAVED
// software pipeline: ONE s_barrier per K block and a homogeneous per-wave instruction stream.
//
// Same semantics (bit-identical) as gemm_w8a8.hip / gemm_w8a8_256.hip / gemm_w8a8_pp.hip
// (reference: ops/gemm/kernel.hpp:390-427, utils.hpp:116-121):
//   acc_f32[m,n] = sum over 128-deep K blocks (ascending) of
//                  fma(float(int32 sum_k a[m,k]*b[n,k]), a_s[m/128,kb]*b_s[n/128,kb], acc)
//
// Why a fourth kernel.  The W8A8 format forces 2 VALU ops (int32->fp32, FMA with the block scale) per
// output element per 128-deep K block = 4 VALU per v_mfma_i32_16x16x64_i8.  On gfx950 a SIMD issues one
// VALU-class instruction (MFMA included) per ~4.3 cycles whatever the number of waves (tools/ubench), so
// a K block of a SIMD's two 128x64 wave tiles costs (64 + 256) x 2 issues ~ 2750 cycles against 2048-2200
// cycles of matrix-pipe work: the kernel is VALU-issue bound and the only thing a schedule can do is keep
// BOTH pipes busy all the time.  The ping-pong kernel (v3) separates MFMA segments and VALU segments with
// 8 barriers per K block; measured, the VALU segments run at ~9 cycles/op beside the partner's MFMAs and
// every barrier costs a start-of-segment penalty: ~6000 cycles per K block.
//
// Here every wave runs the same homogeneous stream, one "slot" = 1 MFMA + 4 dequant VALU:
//   group g = (K block kb, 16-row m sub-tile i) = 8 slots:
//     slots 0-3: MFMA(k 0..63)   of sub-tiles (i, j=0..3)  ||  16 x v_add  (int32 bits -> fp32) of group g-1
//     slots 4-7: MFMA(k 64..127) of sub-tiles (i, j=0..3)  ||  16 x v_fmac (x block scale)     of group g-1
//   so the dequant of a group's results is issued one group (>= 8 MFMAs) after the MFMAs that produce
//   them: no MFMA->VALU dependency stall, no hazard, and the stream is the same mix at every point, so
//   the two waves of a SIMD interleave into a steady MFMA/VALU mix without any choreography.
//   * fragments: weights of the K block stay in 32 VGPRs; the activation fragment of group g+1 is read
//     (2 ds_read_b128) at the start of group g into the other half of a 2-deep ring.
//   * ONE barrier per K block, between groups 6 and 7: by then every LDS read of stage kb has returned
//     (the fragment of group 7 was read during group 6) and each wave's pieces of stage kb+1 have landed
//     (s_waitcnt vmcnt(0) lgkmcnt(0) just before).  Group 7 then reloads the weight fragments from stage
//     kb+1 half by half right behind the MFMAs that last used them, so the next K block starts without a
//     bubble, and the LDS-DMA of stage kb+2 into the freed buffer starts in group 7 and is spread over the
//     next K block's groups 0-4.
//   * int32 -> fp32 without v_cvt: each sub-tile's two-MFMA chain starts from C = 0x4B400000 (1.5*2^23);
//     |sum over a 128-deep block| < 2^22, so the int32 result reinterpreted as fp32 is exactly
//     12582912 + sum and one exact v_add_f32 recovers float(sum).
//   * MFMAs and VALU are asm volatile in program order (the order IS the design); the compiler only
//     allocates registers and inserts s_waitcnt for the LDS reads.
//   * epilogue identical to v3 (permlane32 swap -> 16-byte row-contiguous stores).
#include "td_common.h"

#define F_BN 256
#define F_DUMP 4096               // 2 KB landing area of the L2-prefetch dwords + 2 KB epilogue constants (bias, gate)
// per instantiation (NI = 8 | 4 sixteen-row sub-tiles per wave): tile rows F_BM = 32 NI = 256 | 128; one stage = an activation tile
// of F_BM x 128 B + a weight tile of 256 x 128 B; two stages = 128 | 96 KB (F_EPI = where the epilogue's scratch behind them
// starts: 128 KB in both forms, so that the GELU table fits)
// measured end to end on one box (profiles/r06_fast_dequant_ab.txt, three interleaved repetitions, one-VALU dequant in every arm):
// no four-wave launches 3.227 videos/s, the fused-quantiser GEMM (ffn.0) on it 3.257, + the q|k|v GEMM 3.256, every GEMM 3.221
#define TD_GEMM_W4_DEFAULT 3
// the re-centring period whose instantiations cover EVERY epilogue and tile form (the run-time twin of the exact kernel and the
// library default, csrc/capi.hip); the other periods (2, 4 | 8) exist for the plain / fused-quantiser / residual epilogues of the
// 256 x 256 tile only (tests, A/B)
#ifndef TD_GEMM_FAST_TWIN
#define TD_GEMM_FAST_TWIN 8
#endif
#define F_MAGIC_I 0x4B400000
#define F_MAGIC_F 12582912.0f

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ uint32_t f_swz(uint32_t row, uint32_t chunk) {
  return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4);
}

#define F_FENCE()                             \
  {                                           \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("" ::: "memory");            \
  }
#define F_BARRIER()                           \
  {                                           \
    F_FENCE()                                 \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  }

// GELU-tanh as a TABLE (QOUT epilogue, bf16; round 5).  The value the GELU is applied to has already been rounded to the
// 16-bit output dtype (cast(acc) + bias -> cast, ops/core.py:408-412) and its result is rounded to it again
// (wan2pt1.py:375 on a bf16 tensor): a function from 65 536 bit patterns to 65 536 bit patterns.  td_gelu_tanh costs 7
// VALU + 2 quarter-rate transcendentals + 1.5 conversions per element, i.e. most of the 19 VALU-class instructions per
// element that made ffn.0 (K = 1536: only 12 K blocks to amortise an epilogue over) the slowest GEMM of the block per FLOP
// (profiles/r04_ffn0_epilogue_instruction_count.txt).  The table is td_gelu_tanh evaluated ON THE DEVICE for every bit
// pattern (gelu_table_kernel, once per device) — bit-identical to the inline form by construction — and the epilogue
// fetches its 128 KB into the stage buffers (free after the main loop) by LDS-DMA while the accumulators are converted,
// then looks every element up with ds_read_u16_d16(_hi): 2 VALU (address) + 1 LDS gather per element.
__device__ uint16_t g_gelu_tab_bf16[65536];

__global__ void gelu_table_kernel(uint16_t* __restrict__ t) {
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  t[b] = (uint16_t)(pack2<TD_BF16>(td_gelu_tanh(bf16_bits_to_f32(b)), 0.f) & 0xffffu);
}

// the device's table, initialised on first use (eagerly and synchronously; a first use INSIDE a stream capture records the
// init kernel into the graph instead and leaves the device marked uninitialised, so that the first eager use still does it)
static const uint16_t* td_gelu_table_bf16(hipStream_t st) {
  static std::atomic<uint64_t> ready{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  uint16_t* p = nullptr;
  if (hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(g_gelu_tab_bf16)) != hipSuccess) return nullptr;
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0ull;
  if (bit && (ready.load(std::memory_order_acquire) & bit)) return p;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cs);
  gelu_table_kernel<<<256, 256, 0, st>>>(p);
  if (cs == hipStreamCaptureStatusNone) {
    (void)hipStreamSynchronize(st);   // once per device: launches on OTHER streams may follow immediately
    if (bit) ready.fetch_or(bit, std::memory_order_release);
  }
  return p;
}

// DBG = 1: profiling instantiation, records s_memtime at every group boundary of K blocks 8 and 9 for waves 0
// and 4 of workgroup 0 (read back with td_debug_read; tools/gemm_trace.py)
// SCHED bit 0: issue all 8 LDS-DMA pieces of stage kb+2 in groups 7 and 0 (right after the barrier) instead of
// spreading them over groups 7,0..4
// QOUT: the epilogue block-quantises the (bias/GELU'd, 16-bit-rounded) result for the NEXT W8A8 GEMM instead of
// storing it: D is int8 [M, ldd], QS the fp32 scales [ceil(M/128), ldqs] — bit-identical to td_gemm_w8a8 followed by
// td_quant_i8_block128 (quant.hip), minus one 2-byte write, one 2-byte read and a launch.
// RES: the epilogue applies the block's gated residual in place (a7, wan2pt1.py:405-406,412-413):
//   D[m,n] = D[m,n] + cast(cast(y[m,n]) * cast(gate[n]))  (gate == nullptr: plain add), y = this GEMM's 16-bit result —
// bit-identical to td_gemm_w8a8 followed by td_gated_residual, minus a 2-byte write, a 2-byte read and a launch.
// FAST = G > 0: ONE-VALU dequant (NOT bit-identical to the exact kernel, see the bound below).  The chain's int32 result,
// read as fp32, IS 1.5*2^23 + isum, so `acc = fma(raw, s, acc)` adds isum*s + M*s (M = 1.5*2^23) with one instruction
// instead of two; the M*s terms are taken out again every G K blocks by one v_add per accumulator with
// c = -(fp32)(sum of M*s_k over the group), formed in fp64 with the rounding remainder carried into the next group (and
// the last remainder applied in the drain), so that exactly M * sum_k s_k is subtracted in total.  What is lost: while
// up to G+1 offsets ride on an accumulator, each fma rounds at magnitude <= (G+1)*M*s instead of |acc|:
//   |fast - exact| <= 2^-24 * (G+1) * 1.5*2^23 * sum_k s_k = 0.75 (G+1) sum_k s_k     (fp32, before the 16-bit cast)
// i.e. < 4 counts of a K block's integer sum per block for G = 4, against typical |isum| ~ 1e4 (rel. ~1e-4..3e-4, an
// order below the bf16 rounding of the result).  1.25 VALU per element and K block instead of 2.
// STATS: the (plain or RES) epilogue also emits, per output row and per 64-column piece of it, (mean, M2 = sum of squared
// deviations from that mean) of the 16-bit values it stores — QS is then a float2 workspace [M, ldqs] with ldqs = N / 64 pieces per row.  td_row_stats_finalize
// turns the pieces into the row statistics of the LayerNorm / RMSNorm that reads this output next, which therefore needs no
// statistics pass of its own over the [M, N] tensor.
// VT = 1 | 2 (plain epilogue only): the columns n >= ldqs (a multiple of 256: whole tiles) are V of a fused q|k|v
// projection, heads of 128 columns — instead of storing them row-major the epilogue writes them as the attention kernels'
// V^T MFMA tiles (td_v_transpose's layout: [head][ceil(M/64)][128 d][64 key positions], position = key with bits 2/3
// swapped, rows m >= M zero), fp16 (VT = 1, the Sage PV operand: cast of the stored 16-bit value) or the output dtype
// (VT = 2), into QS (reinterpreted as 16-bit): bit-identical to td_gemm_w8a8 followed by td_v_transpose, minus a 2-byte
// write, a 2-byte read and a launch.  Each wave transposes its 128 keys x 64 d through a private LDS region (the stage
// buffers are free after the main loop) in two 64-key halves: 2-byte scatter writes, 16-byte row reads, full-line stores.
// NI = 4 (round 5): a 128(M) x 256(N) tile for the row counts of a sequence shard (M = 4096 per rank of 8: 16 x 18 = 288 tiles of
// 256 x 256 are 1.1 rounds of the 256 CUs, paid as 2; the 96 tiles at N = 1536 leave 160 CUs idle): the SAME eight waves, two
// per SIMD, each on 64 x 64 instead of 128 x 64 — a K block is four groups instead of eight, everything else (slot structure,
// one barrier per K block, fragment ring, epilogues) is the same code with NI for 8.  (A four-wave form with the unchanged
// 128 x 64 wave tile — one wave per SIMD — measured 0.75 of the per-FLOP rate: tools/gemm_small_m.py, profiles/r05_gemm_small_m.txt.)
// Bit-identical: the arithmetic per output element does not depend on the tile it is computed in.
// NW = 4 (round 6): the FOUR-wave form — one wave per SIMD, the unchanged 128 x 64 wave tile, waves side by side in N: a
// 128(M) x 256(N) tile per 256-thread workgroup and TWO independent workgroups per CU (68 KB of LDS and 256 VGPRs each).  What
// the eight-wave workgroup pays once per K block with nothing beside it — the barrier, the weight-fragment refill burst behind
// it, the prologue until the first stage has landed and the epilogue's store tail — one workgroup now pays while the other
// workgroup of the CU keeps both waves' worth of issue slots and the matrix pipe busy (the arrangement that bought the VAE
// convolution its short reductions, csrc/vae_conv3.hip).  LDS: the activation tile (128 rows x 128 B = 16 KB, read by all four
// waves) in two stages; the weight tile is NOT shared in this arrangement — wave w alone reads rows 64 w .. 64 w + 63 — so each
// wave stages its own 8 KB by its own eight LDS-DMA pieces into a PRIVATE region, single-buffered: the K block's weight fragments
// live in registers, the region is dead as soon as they are read (group 7 of the previous block) and the next block's pieces
// are issued right behind that read with a whole K block to land; no barrier guards it.  2 x 16 + 4 x 8 = 64 KB.
// Same slots, same fragment ring, same epilogues: bit-identical to the eight-wave form (test_gemm_four_wave_form_is_bit_identical).
template <int ODT, int EPI, bool HAS_BIAS, int DBG = 0, int SCHED = 0, bool QOUT = false, bool RES = false, int FAST = 0, bool STATS = false, int VT = 0, int NI = 8, int NW = 8>
__global__ __launch_bounds__(NW * 64, 2) void gemm_w8a8_fi_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ AS, const int8_t* __restrict__ B,
    const float* __restrict__ BS, const uint16_t* __restrict__ bias, uint16_t* __restrict__ D,
    int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int tiles_m, int tiles_n, int group_m,
    unsigned long long* __restrict__ dbg, float* __restrict__ QS, int64_t ldqs, const float* __restrict__ gate,
    const uint16_t* __restrict__ gelu_tab = nullptr, int tm0 = 0) {
  // tm0: index of this launch's first row tile (in tiles of F_BM rows) — a launch may cover a row range [tm0 F_BM, ...) of the
  // problem (launch_gemm_fi: whole rounds on 256-row tiles, the remainder on 128-row tiles); M stays the problem's row count
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  constexpr bool W4 = NW == 4;
  static_assert(NW == 8 || (NW == 4 && NI == 8 && DBG == 0 && SCHED == 0), "four-wave form: 128 x 64 wave tiles, no profiling instantiations");
  const int wm = W4 ? 0 : wave >> 2, wn = wave & 3;
  constexpr int WROWS = 16 * NI;                 // rows of a wave tile (128 | 64)
  constexpr int F_BM = W4 ? WROWS : 2 * WROWS;   // tile rows
  constexpr int F_TILE = F_BM * 128;             // activation tile bytes per K block
  constexpr int F_STAGE = W4 ? F_TILE : F_TILE + 256 * 128;    // + weight tile (four-wave form: a stage is the activation tile alone)
  constexpr int F_WBASE = W4 ? 2 * F_TILE : F_TILE;            // weight rows: behind the activation tile of a stage | the waves' private regions behind both stages
  constexpr int F_EPI = W4 ? 65536 : 131072;     // epilogue scratch (bias / gate constants, amax exchange) behind the stage area
  constexpr int NA = W4 ? 4 : NI / 2;            // activation chunks (8 rows x 128 B) a wave moves per stage (weights: 4 | 8)
  unsigned long long dbg_t[40];
  int dbg_n = 0;
  unsigned long long c_t0 = 0, c_t1 = 0, c_t2 = 0;
  if constexpr (DBG >= 2) c_t0 = __builtin_amdgcn_s_memtime();
#define F_STAMP()                                                                             \
  if constexpr (DBG == 1) {                                                                   \
    if ((kb == 8 || kb == 9) && dbg_n < 40) dbg_t[dbg_n++] = __builtin_amdgcn_s_memtime();    \
  }

  // ---- tile assignment: XCD remap, then m-grouped raster ----
  const uint32_t vid = xcd_remap(blockIdx.x, gridDim.x);
  const int per_group = group_m * tiles_n;
  const int gid = vid / per_group;
  const int first_m = gid * group_m;
  const int gsz = min(group_m, tiles_m - first_m);
  const int in_g = vid % per_group;
  const int tm = first_m + in_g % gsz + tm0;
  const int tn = in_g / gsz;
  const int64_t m0 = (int64_t)tm * F_BM, n0 = (int64_t)tn * F_BN;
  const int nk = (int)(K / 128);

  // ---- LDS-DMA pieces: wave w moves chunks c = w + 8t (8 rows x 128 B) of both operand tiles ----
  // four-wave form: wave w moves activation chunks c = w + 4t and ALL eight chunks of its own 64 weight rows — piece t of those
  // = rows 8t .. 8t + 7 of the wave's region; the row step rides on the instruction's scalar offset, the source swizzle depends
  // on t's parity only: two address registers (gb[0] even, gb[1] odd t)
  uint32_t ga[4], gb[4];     // (NI = 4: ga[0..1] used — the activation tile has 16 chunks, two per wave)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = wave + NW * t;
    const int row = 8 * c + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // source-side swizzle (LDS image is lane-linear)
    int64_t am = m0 + (t < NA ? row : 0); if (am > M - 1) am = M - 1;   // tail rows: clamp (never stored)
    int64_t bn = n0 + row; if (bn > N - 1) bn = N - 1;
    ga[t] = (uint32_t)(am * lda + chunk * 16);
    gb[t] = (uint32_t)(bn * ldb + chunk * 16);
    if constexpr (W4) {
      // rows >= N (n % 8 == 0: whole pieces) are beyond the descriptor's extent: the buffer unit answers them with zeros, and
      // nothing computed from them is ever stored
      const int lrow = 8 * (t & 1) + (lane >> 3);                        // local row of pieces t (mod 2): 8t + (lane >> 3), t < 2
      gb[t] = (uint32_t)((n0 + wave * 64 + lrow) * ldb + ((lane & 7) ^ ((lrow >> 1) & 7)) * 16);
    }
  }
  const uint32_t w4_step = (uint32_t)(8 * ldb);   // row step of a weight piece (scalar)
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(uint32_t)(M * lda), 0x00020000);
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(uint32_t)(N * ldb), 0x00020000);
  // L2 prefetch (SCHED bit 1): one buffer_load_dword per wave touches 64 cache lines (one 128-B line per lane) of a
  // stage several K blocks ahead, so that the LDS-DMA of that stage later hits in L2 instead of waiting on HBM.
  // waves 0-3: activation rows 64*(wave&3)+lane, waves 4-7: weight rows.  The loaded dword is never used.
  uint32_t pf_off;
  {
    const int row = (64 * (wave & 3) + lane) % F_BM;
    int64_t am = m0 + row; if (am > M - 1) am = M - 1;
    int64_t bn = n0 + row; if (bn > N - 1) bn = N - 1;
    pf_off = wave < 4 ? (uint32_t)(am * lda) : (uint32_t)(bn * ldb);
  }
  // (LDS-DMA form with a 4-byte element: no destination VGPR; the dwords land in a 2 KB dump area behind the stages)
#define F_PREFETCH(kb_)                                                                           \
  if constexpr ((SCHED & 2) != 0) {                                                               \
    char* dump_ = smem + F_EPI + wave * 256;                                                      \
    if (wave < 4)                                                                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)dump_, 4, pf_off, (kb_) * 128, 0, 0); \
    else                                                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lptr_t)dump_, 4, pf_off, (kb_) * 128, 0, 0); \
  }
  // piece p of stage kb_ into buffer (kb_ & 1): p = 0..3 activation chunks, 4..7 weight chunks
#define F_PIECE(kb_, p_)                                                                          \
  {                                                                                               \
    char* sb_ = smem + ((kb_) & 1) * F_STAGE + wave * 1024;                                       \
    if ((p_) < 4)                                                                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(sb_ + ((p_) & 3) * 8192), 16,     \
                                               ga[(p_) & 3], (kb_) * 128, 0, 0);                  \
    else                                                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lptr_t)(sb_ + F_TILE + ((p_) & 3) * 8192), \
                                               16, gb[(p_) & 3], (kb_) * 128, 0, 0);              \
  }

  // four-wave form: activation piece t (0..3) of stage kb_, weight piece t (0..7) of K block kb_ into the wave's private region
#define F_PIECE_A4(kb_, t_)                                                                       \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(smem + ((kb_) & 1) * F_STAGE + wave * 1024 + (t_) * 4096), 16, \
                                           ga[t_], (kb_) * 128, 0, 0);
#define F_PIECE_B4(kb_, t_)                                                                       \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lptr_t)(smem + F_WBASE + wave * 8192 + (t_) * 1024), 16, \
                                           gb[(t_) & 1], (kb_) * 128 + ((t_) >> 1) * 2 * w4_step, 0, 0);

  // ---- fragment read offsets (within a stage); weight rows use the bit-2/3-swapped order ----
  const int pr = (l16 & 3) | ((l16 & 4) << 1) | ((l16 & 8) >> 1);
  uint32_t xoff[2], woff[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    xoff[kc] = f_swz(wm * WROWS + l16, 4 * kc + lq);
    woff[kc] = W4 ? F_WBASE + wave * 8192 + f_swz(pr, 4 * kc + lq) : F_TILE + f_swz(wn * 64 + pr, 4 * kc + lq);
  }

  v4f accf[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) accf[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  // scale rows of this wave's 128x64 sub-tile (clamped for tail tiles)
  int64_t mb = (m0 + wm * WROWS) >> 7, nb = (n0 + wn * 64) >> 7;
  const int64_t mb_max = td_cdiv(M, 128) - 1, nb_max = td_cdiv(N, 128) - 1;
  if (mb > mb_max) mb = mb_max;
  if (nb > nb_max) nb = nb_max;
  const float* as_row = AS + mb * nk;
  const float* bs_row = BS + nb * nk;

  v4i magic = {F_MAGIC_I, F_MAGIC_I, F_MAGIC_I, F_MAGIC_I};
  asm volatile("" : "+v"(magic));  // opaque: keep it in 4 VGPRs, never re-materialised inside the loop
  v4i wf[4][2], xf[2][2], t[2][4];
  // ring slot 1 plays "group -1": 1.5*2^23 + 0 -> its dequant adds 0 * 0 to the (zero) accumulators
#pragma unroll
  for (int j = 0; j < 4; ++j) t[1][j] = magic;

#define F_LOAD_X(st_, i_, slot_)                                                                  \
  _Pragma("unroll") for (int kc = 0; kc < 2; ++kc)                                                \
    xf[slot_][kc] = *reinterpret_cast<const v4i*>((st_) + xoff[kc] + (i_) * 2048);
#define F_LOAD_W(st_, kc_)                                                                        \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
    wf[j][kc_] = *reinterpret_cast<const v4i*>((st_) + woff[kc_] + j * 2048);
  // first MFMA of a sub-tile's chain: C = magic (dst may not overlap the sources)
#define F_MFMA0(d_, a_, b_)                                                                       \
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(d_) : "v"(a_), "v"(b_), "v"(magic));
#define F_MFMA1(d_, a_, b_)                                                                       \
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(d_) : "v"(a_), "v"(b_));
#define F_ADD4(v_)                                                                                \
  _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
    asm volatile("v_add_f32 %0, %1, %0" : "+v"((v_)[r]) : "s"(-F_MAGIC_F));
#define F_FMAC4(acc_, v_, sc_)                                                                    \
  _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"((acc_)[r]) : "s"(sc_), "v"((v_)[r]));
#define F_ADDC4(acc_, c_)                                                                         \
  _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
    asm volatile("v_add_f32 %0, %1, %0" : "+v"((acc_)[r]) : "s"(c_));

  // ---- prologue: stage 0 and stage 1 in flight; wait for stage 0; fragments of group (0,0) ----
  if constexpr (W4) {
    // block 0 (4 + 8 pieces) and the activation tile of block 1 in flight; block 1's weights follow the first fragment read
#pragma unroll
    for (int t = 0; t < 4; ++t) F_PIECE_A4(0, t)
#pragma unroll
    for (int t = 0; t < 8; ++t) F_PIECE_B4(0, t)
    if (nk > 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) F_PIECE_A4(1, t)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    F_BARRIER()
    F_LOAD_W(smem, 0)
    F_LOAD_W(smem, 1)
    F_LOAD_X(smem, 0, 0)
    if (nk > 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the private region has been read: block 1's weights may land in it
#pragma unroll
      for (int t = 0; t < 8; ++t) F_PIECE_B4(1, t)
    }
  } else {
#pragma unroll
  for (int p = 0; p < 8; ++p) { if ((p & 3) < NA || p >= 4) F_PIECE(0, p) }
  if (nk > 1) {
#pragma unroll
    for (int p = 0; p < 8; ++p) { if ((p & 3) < NA || p >= 4) F_PIECE(1, p) }
    if constexpr (NI == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  F_PREFETCH(2)
  F_PREFETCH(3)
  F_BARRIER()
  if constexpr (DBG >= 2) c_t1 = __builtin_amdgcn_s_memtime();
  F_LOAD_W(smem, 0)
  F_LOAD_W(smem, 1)
  F_LOAD_X(smem, 0, 0)
  }

  // block scales live in SGPRs: sc_old = K block of the group being dequantised at i == 0 (the previous
  // block's last group), sc_new = this block's.  (sa*sb) formed first, kernel.hpp:418.
  float sc_old = 0.f;
  const float sv0 = as_row[0] * bs_row[0];
  float sc_new = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sv0)));
  // FAST: sum of M*s_k not yet taken out of the accumulators (fp64, per-wave uniform), fed from the VECTOR copy of each
  // scale so that the SGPR copies only have the fmacs as users
  double c_sum = FAST > 0 ? (double)sv0 * (double)F_MAGIC_F : 0.0;

  constexpr int UNR = FAST > 0 ? FAST : 1;   // FAST: the K loop is unrolled by the recentring period (no branch in the stream)
  float c_neg = 0.f;
  for (int kb0 = 0; kb0 < nk; kb0 += UNR) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int kb = kb0 + u;
    if (UNR > 1 && kb >= nk) break;
    // (no re-centring in the LAST K block: the drain takes everything that still rides on the accumulators out anyway — for
    //  K = 1536 that is one add per element less out of 17, round 6)
    const bool recentre = FAST > 0 && u == UNR - 1 && kb + 1 < nk;
    if constexpr (FAST > 0) {
      if (recentre) {
        const float c_hi = (float)c_sum;
        c_sum -= (double)c_hi;
        c_neg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -c_hi)));
      }
    }
    const char* st = smem + (kb & 1) * F_STAGE;
    const char* stn = smem + ((kb + 1) & 1) * F_STAGE;
    const char* wsn = W4 ? smem : stn;          // where the NEXT block's weight fragments are read (woff carries the private region)
    const bool more = kb + 1 < nk;
    const bool dma_tail = (kb >= 1) && more;   // rest of stage kb+1 (stage 1 was issued by the prologue)
    const bool dma_head = kb + 2 < nk;         // first pieces of stage kb+2, after this block's barrier
    float sa_n = 0.f, sb_n = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int cur = i & 1, prv = cur ^ 1;
      const int pi = (i + NI - 1) % NI;  // m sub-tile of the group being dequantised
      const float scl = (i == 0) ? sc_old : sc_new;
      F_STAMP()
      // -- fragment prefetch for the next group into the other ring slot (its last readers, the MFMAs of
      //    the previous group, have all been issued)
      if (i < NI - 1) { F_LOAD_X(st, i + 1, prv) }
      else if (more) { F_LOAD_X(stn, 0, prv) }
      F_FENCE()
      // -- slots 0-3
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        F_MFMA0(t[cur][j], wf[j][0], xf[cur][0])
        if constexpr (FAST == 0) { F_ADD4(t[prv][j]) }
        else if (i == 0 ? (u == 0 && kb > 0) : recentre) { F_ADDC4(accf[pi][j], c_neg) }   // (row group NI-1: one group later = the next block's first)
      }
      F_FENCE()
      if (i == NI - 1 && more) { F_LOAD_W(wsn, 0) }
      // -- LDS-DMA issue (VMEM issue slots of this wave only)
      if (i == NI - 1) { F_PREFETCH(kb + 4) }
      if constexpr (W4) {
        // behind this block's barrier (group 7): the activation tile two blocks ahead into the stage everyone has just left;
        // next block's groups 0-2: the rest of it and the weights of the block after — the private region was read in group 7
        // (lgkmcnt(0): the DMA engine does not order itself against this wave's outstanding LDS reads)
        if (i == 7) { if (dma_head) { F_PIECE_A4(kb + 2, 0) F_PIECE_A4(kb + 2, 1) } }
        else if (i == 0) { if (dma_tail) { F_PIECE_A4(kb + 1, 2) F_PIECE_A4(kb + 1, 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); F_PIECE_B4(kb + 1, 0) F_PIECE_B4(kb + 1, 1) } }
        else if (i == 1) { if (dma_tail) { F_PIECE_B4(kb + 1, 2) F_PIECE_B4(kb + 1, 3) F_PIECE_B4(kb + 1, 4) } }
        else if (i == 2) { if (dma_tail) { F_PIECE_B4(kb + 1, 5) F_PIECE_B4(kb + 1, 6) F_PIECE_B4(kb + 1, 7) } }
      } else if constexpr (NI == 4) {
        // six pieces per wave and stage (2 activation + 4 weight chunks), all of stage kb+2 right behind this block's barrier:
        // three in the last group, three in the next block's first — ~3 groups ahead of the barrier that needs them
        if (i == 3) { if (dma_head) { F_PIECE(kb + 2, 0) F_PIECE(kb + 2, 4) F_PIECE(kb + 2, 1) } }
        else if (i == 0) { if (dma_tail) { F_PIECE(kb + 1, 5) F_PIECE(kb + 1, 6) F_PIECE(kb + 1, 7) } }
      } else if constexpr (SCHED & 1) {
        if (i == 7) { if (dma_head) { F_PIECE(kb + 2, 0) F_PIECE(kb + 2, 4) F_PIECE(kb + 2, 1) F_PIECE(kb + 2, 5) } }
        else if (i == 0) { if (dma_tail) { F_PIECE(kb + 1, 2) F_PIECE(kb + 1, 6) F_PIECE(kb + 1, 3) F_PIECE(kb + 1, 7) } }
      } else {
        if (i == 7) { if (dma_head) { F_PIECE(kb + 2, 0) F_PIECE(kb + 2, 4) } }
        else if (i == 0) { if (dma_tail) { F_PIECE(kb + 1, 1) F_PIECE(kb + 1, 5) } }
        else if (i == 1) { if (dma_tail) { F_PIECE(kb + 1, 2) } }
        else if (i == 2) { if (dma_tail) { F_PIECE(kb + 1, 6) } }
        else if (i == 3) { if (dma_tail) { F_PIECE(kb + 1, 3) } }
        else if (i == 4) { if (dma_tail) { F_PIECE(kb + 1, 7) } }
      }
      if (i == NI - 3) { if (more) { sa_n = as_row[kb + 1]; sb_n = bs_row[kb + 1]; } }  // scalar loads
      F_FENCE()
      F_STAMP()
      // -- slots 4-7
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        F_MFMA1(t[cur][j], wf[j][1], xf[cur][1])
        F_FMAC4(accf[pi][j], t[prv][j], scl)
      }
      F_FENCE()
      if (i == NI - 1 && more) { F_LOAD_W(wsn, 1) }
      if (i == NI - 2) {
        // every LDS read of stage kb has returned (group 7's fragment was read at the top of this group),
        // this wave's pieces of stage kb+1 have landed; after the barrier: everyone's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        F_BARRIER()
      }
    }
    sc_old = sc_new;
    const float sv = sa_n * sb_n;
    sc_new = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sv)));
    if constexpr (FAST > 0) c_sum = __builtin_fma((double)sv, (double)F_MAGIC_F, c_sum);
  }
  }
  if constexpr (DBG == 1) {
    if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0)
      for (int q = 0; q < 40; ++q) dbg[(wave >> 2) * 64 + q] = q < dbg_n ? dbg_t[q] : 0ull;
  }
  if constexpr (DBG >= 2) c_t2 = __builtin_amdgcn_s_memtime();
  // ---- drain: dequant of the last group (nk-1, 7); its MFMAs were issued >= 4 slots ago, the last one
  //      just now: give the matrix pipe its 4 passes before the VALU reads ----
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  if constexpr (FAST == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { F_ADD4(t[1][j]) }
  }
  // (FAST: every row group takes a re-centring add right BEFORE the fmac of the re-centring block's own sums — for the last row
  //  group that fmac sits in the NEXT block's first group, and a re-centring block always has a next block: the same order of
  //  roundings for every output element whatever the tile form (NI = 4 | 8, eight | four waves), i.e. the one-VALU mode is
  //  bit-identical across the launch plans too)
#pragma unroll
  for (int j = 0; j < 4; ++j) { F_FMAC4(accf[NI - 1][j], t[1][j], sc_old) }
  if constexpr (FAST > 0) {
    // whatever M*s_k has not been taken out yet (the last, partial group + the fp64 -> fp32 remainders)
    const float c_hi = (float)c_sum;
    const float c_lo = (float)(c_sum - (double)c_hi);
    const float c_last = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -c_hi)));
    const float c_last2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -c_lo)));
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { F_ADDC4(accf[i][j], c_last) F_ADDC4(accf[i][j], c_last2) }
  }

  // ---- epilogue ----
  // lane owns m = ..+l16; accumulator (i,j) holds n_local = 16j + r + 8(lq&1) + 4(lq>>1).
  // After the swap lanes with lq<2 store 8 consecutive n of sub-tile ja, lanes with lq>=2 of jb.
  const int hi = lq >> 1;
  // TD_PHASE_MARKS (analysis builds only, tools/epilogue_valu_count.py): assembler comments at the phase boundaries of the
  // QOUT epilogue so that its instructions can be counted per phase in the .s file
#ifdef TD_PHASE_MARKS
#define F_MARK(name_) asm volatile("; TD_PHASE " name_ ::: "memory");
#else
#define F_MARK(name_)
#endif
  if constexpr (QOUT) {
    F_MARK("qout_begin")
    // TAB: the GELU as a table lookup (see g_gelu_tab_bf16).  The table's 128 KB replace the stage buffers: every wave must
    // be past its last fragment read (barrier), then each wave fetches its eighth (16 pieces of 1 KB) by LDS-DMA — in
    // flight while the accumulators are cast and biased below.
    constexpr bool TAB = (EPI == TD_EPI_GELU_TANH) && (ODT == TD_BF16);
    const bool use_tab = TAB && gelu_tab != nullptr;    // (kernel argument: workgroup-uniform)
    if constexpr (TAB) {
      if (use_tab) {
        F_BARRIER()
        const auto rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void*)gelu_tab, 0, 131072, 0x00020000);
#pragma unroll
        for (int t = 0; t < 16; ++t)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t, (lptr_t)(smem + (t * 8 + wave) * 1024), 16,
                                                   (uint32_t)((t * 8 + wave) * 1024 + lane * 16), 0, 0, 0);
      }
    }
    // (1) the 16-bit results exactly as the plain epilogue would store them, kept in 64 VGPRs
    uint32_t pk[NI][4][2];
    const bool tail = (m0 + F_BM > M) || (n0 + F_BN > N);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool m_ok = (m0 + wm * WROWS + i * 16 + l16) < M;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t n = n0 + wn * 64 + j * 16 + 8 * (lq & 1) + 4 * hi;
        const bool ok = m_ok && n < N;
        float bf[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) {
          if (n > N - 4) n = N - 4;
          const uint2 bb = *reinterpret_cast<const uint2*>(bias + n);
          unpack2<ODT>(bb.x, bf[0], bf[1]);
          unpack2<ODT>(bb.y, bf[2], bf[3]);
        }
        if (TAB && use_tab) {   // cast, + bias, cast — the GELU follows below
          pk[i][j][0] = td_gemm_epilogue2<ODT, TD_EPI_NONE, HAS_BIAS>(accf[i][j][0], accf[i][j][1], bf[0], bf[1]);
          pk[i][j][1] = td_gemm_epilogue2<ODT, TD_EPI_NONE, HAS_BIAS>(accf[i][j][2], accf[i][j][3], bf[2], bf[3]);
        } else {
          pk[i][j][0] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(accf[i][j][0], accf[i][j][1], bf[0], bf[1]);
          pk[i][j][1] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(accf[i][j][2], accf[i][j][3], bf[2], bf[3]);
        }
        if (tail && !ok) { pk[i][j][0] = 0u; pk[i][j][1] = 0u; }  // rows/cols outside the matrix: zero-filled (load.hpp:24-47); gelu(0) = 0
      }
    }
    if constexpr (TAB) {
      if (use_tab) {
        F_MARK("qout_table")
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the table have landed ...
        F_BARRIER()                                        // ... and everyone's
        const uint32_t tb = (uint32_t)(uintptr_t)(lptr_t)smem;
        // Measured (tools/ffn0_epilogue_split.py, tools/gelu_table_ab.py -> profiles/r05_gelu_table.txt): the inline GELU costs
        // a [32760 x 8960 x 1536] launch 80-85 us over the same fused quantiser without it, this form 68-75 — 1024 16-bit
        // gathers per tile at ~7 LDS cycles each (32 random addresses on 32 banks) instead of 2 quarter-rate transcendentals
        // + 7 VALU per element.  Splitting a wave's rows between the two forms (the two waves of a SIMD in opposite order, so
        // that the LDS serves one while the transcendental unit serves the other) measured BETWEEN the two, not below: dropped.
        // gfx950 runs with SRAM-ECC: a d16 load ZEROES the other half of its destination — two registers, OR-ed.
#define F_GELU_LOOKUP(i0_)                                                                          \
        _Pragma("unroll") for (int i = (i0_); i < (i0_) + 4; ++i) {                                 \
          uint32_t glo[4][2], ghi[4][2];                                                            \
          _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                         \
              const uint32_t w = pk[i][j][e];                                                       \
              const uint32_t a_lo = tb + ((w << 1) & 0x1fffeu), a_hi = tb + ((w >> 15) & 0x1fffeu); \
              asm volatile("ds_read_u16 %0, %1" : "=v"(glo[j][e]) : "v"(a_lo));                     \
              asm volatile("ds_read_u16_d16_hi %0, %1" : "=v"(ghi[j][e]) : "v"(a_hi));              \
            }                                                                                       \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
          _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                         \
              asm volatile("" : "+v"(glo[j][e]), "+v"(ghi[j][e]));   /* uses stay behind the wait */ \
              pk[i][j][e] = glo[j][e] | ghi[j][e];                                                  \
            }                                                                                       \
        }
        F_GELU_LOOKUP(0)
        if constexpr (NI == 8) { F_GELU_LOOKUP(4) }
#undef F_GELU_LOOKUP
      }
    }
    F_MARK("qout_amax")
    // (2) amax of this wave's 128x64 half of the 128x128 quant block: max over |x| as 15-bit magnitudes (both 16-bit
    //     formats are monotone in their magnitude bits), two per v_pk_max_u16
    uint32_t mx = 0u;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t a = pk[i][j][e] & 0x7fff7fffu;
          asm("v_pk_max_u16 %0, %0, %1" : "+v"(mx) : "v"(a));
        }
    uint32_t m16 = max(mx & 0xffffu, mx >> 16);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m16 = max(m16, (uint32_t)__shfl_xor((int)m16, o, 64));
    F_MARK("qout_exchange")
    // (3) the other half belongs to wave ^ 1: exchange through LDS (free: every wave is past the last barrier
    //     of the main loop, nothing reads or lands in the stages any more)
    uint32_t* red = reinterpret_cast<uint32_t*>(smem + F_EPI + 2048);   // (behind 128 KB: the stage area may hold the GELU table)
    if (lane == 0) red[wave] = m16;
    __syncthreads();
    m16 = max(red[wave], red[wave ^ 1]);
    if constexpr (NI == 4) m16 = max(m16, max(red[wave ^ 4], red[wave ^ 5]));   // 64-row wave tiles: the quant block spans both wm
    float amax = half_bits_to_f32<ODT>(m16);
    amax = fmaxf(amax, 1e-8f);
    const float mult = 128.0f / amax;  // IEEE division, as quant.hip
    {
      const int64_t mb_q = (m0 + wm * WROWS) >> 7, nb_q = (n0 + wn * 64) >> 7;
      if (lane == 0 && (wn & 1) == 0 && (NI == 8 || wm == 0) && (m0 + wm * WROWS) < M && (n0 + wn * 64) < N) QS[mb_q * ldqs + nb_q] = amax / 128.0f;
    }
    // (4) quantise: q = sat_s8(rne(x * mult)).  rne via the 1.5*2^23 add (exact for |v| < 2^22, same result as
    //     rintf of the rounded product); its low byte IS the two's-complement code.  |x*mult| <= 128(1+eps), so only
    //     +128 needs the clamp.  (5) two lane swaps gather 16 consecutive n per lane -> one 16-byte store per row.
    int8_t* Dq = reinterpret_cast<int8_t*>(D);
    F_MARK("qout_quantise_store")
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int64_t m = m0 + wm * WROWS + i * 16 + l16;
      uint32_t qd[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x[4];
        unpack2<ODT>(pk[i][j][0], x[0], x[1]);
        unpack2<ODT>(pk[i][j][1], x[2], x[3]);
        uint32_t w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = x[r] * mult;
          v = v + F_MAGIC_F;
          v = fminf(v, F_MAGIC_F + 127.0f);
          w[r] = __float_as_uint(v);
        }
        const uint32_t lo = __builtin_amdgcn_perm(w[1], w[0], 0x0c0c0400u);  // bytes: w0.b0, w1.b0, 0, 0
        const uint32_t hh = __builtin_amdgcn_perm(w[3], w[2], 0x04000c0cu);  // bytes: 0, 0, w2.b0, w3.b0
        qd[j] = lo | hh;
      }
      // permlane32: lower lanes get {own, partner} quads of tile ja, upper lanes of tile jb (8 consecutive n)
      auto p0 = __builtin_amdgcn_permlane32_swap(qd[0], qd[1], false, false);
      auto p1 = __builtin_amdgcn_permlane32_swap(qd[2], qd[3], false, false);
      // permlane16: rows (16 lanes) 1,3 of the first operand <-> rows 0,2 of the second: 16 consecutive n per lane,
      // of tile jt = ((lq & 1) << 1) | (lq >> 1)
      auto e0 = __builtin_amdgcn_permlane16_swap(p0[0], p1[0], false, false);
      auto e1 = __builtin_amdgcn_permlane16_swap(p0[1], p1[1], false, false);
      const uint4 v = make_uint4(e0[0], e1[0], e0[1], e1[1]);
      const int jt = ((lq & 1) << 1) | hi;
      const int64_t n = n0 + wn * 64 + jt * 16;
      if (m < M && n < N) *reinterpret_cast<uint4*>(Dq + m * ldd + n) = v;
    }
    F_MARK("qout_end")
    return;
  }
  // RES: the residual tile is fetched up front — 16 independent 16-byte loads per lane in flight, at the addresses
  // this lane will store to (row i*16 + l16, the 8 consecutive n it owns after the lane swap) — so their latency
  // overlaps the conversion of the accumulators instead of serialising load -> add -> store per row
  uint4 xres[RES ? NI : 1][2];
  if constexpr (RES) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int64_t m = m0 + wm * WROWS + i * 16 + l16;
        const int64_t n = n0 + wn * 64 + (hi ? 2 * jp + 1 : 2 * jp) * 16 + 8 * (lq & 1);
        xres[i][jp] = make_uint4(0, 0, 0, 0);
        if (m < M && n < N) xres[i][jp] = *reinterpret_cast<const uint4*>(D + m * ldd + n);
      }
  }
  // Full-line stores (round 2; tools/gemm_store_exp.py -> profiles/r02_gemm_store_exp.txt): after the lane swap a lane
  // holds 8 consecutive n of ONE row, so a direct store instruction writes 32-byte pieces of 16 different rows — the
  // epilogue of a K = 1536 tile took 11.7 k cycles.  The rows go through a wave-private LDS region instead (the stage
  // buffers are free after the main loop; [64 rows][64 cols + 8 pad] 16-bit, one 64-row half at a time: conflict-free
  // ds_write_b128 / ds_read_b128) and every store instruction writes 8 rows x one full 128-byte line: 6.9 k cycles,
  // -4...-5.5 % per GEMM, bit-identical.
  // The tile's 256 bias values (widened) and gate values (rounded to the output dtype, a7) go to LDS once: read per row
  // group with ds_read (lgkmcnt).  As global loads inside the row-group loop (the compiler does not hoist them) each one
  // carried an s_waitcnt vmcnt(0), and on gfx9 vmcnt counts STORES too: every row group waited for the previous group's
  // statistics store / the previous half's tile stores to complete.
  float* ep_bias = reinterpret_cast<float*>(smem + F_EPI + 2048);
  float* ep_gate = ep_bias + 256;
  if (tid < 256) {
    const int64_t n = n0 + tid;
    float bv = 0.f, gv = 0.f;
    if (n < N) {
      if constexpr (HAS_BIAS) bv = half_bits_to_f32<ODT>(bias[n]);
      if constexpr (RES) { if (gate != nullptr) gv = round_half<ODT>(gate[n]); }
    }
    ep_bias[tid] = bv;
    if constexpr (RES) ep_gate[tid] = gv;
  }
  __syncthreads();                    // every wave has read its last fragments: the stages may be overwritten
  uint16_t* stg_lds = reinterpret_cast<uint16_t*>(smem) + wave * (64 * 72);
  bool vtile = false;                 // VT: this tile's columns are V -> V^T tiles instead of row-major
  uint16_t* vt_lds = stg_lds;         // the wave's transposition buffer [64 d][64 positions (+8 pad)] 16-bit
  if constexpr (VT != 0) vtile = n0 >= ldqs;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int64_t m = m0 + wm * WROWS + i * 16 + l16;
    float st_s = 0.f, st_q = 0.f, st_c = 0.f;   // STATS: this lane's share of the row's 64-column piece (shifted by st_c)
    uint32_t pk[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float bf[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HAS_BIAS) {
        const float4 bb = *reinterpret_cast<const float4*>(ep_bias + wn * 64 + j * 16 + 8 * (lq & 1) + 4 * hi);
        bf[0] = bb.x; bf[1] = bb.y; bf[2] = bb.z; bf[3] = bb.w;
      }
      pk[j][0] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(accf[i][j][0], accf[i][j][1], bf[0], bf[1]);
      pk[j][1] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(accf[i][j][2], accf[i][j][3], bf[2], bf[3]);
    }
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const int ja = 2 * jp, jb = 2 * jp + 1;
      auto s0 = __builtin_amdgcn_permlane32_swap(pk[ja][0], pk[jb][0], false, false);
      auto s1 = __builtin_amdgcn_permlane32_swap(pk[ja][1], pk[jb][1], false, false);
      const uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      const int jt = hi ? jb : ja;
      const int64_t n = n0 + wn * 64 + jt * 16 + 8 * (lq & 1);
      if constexpr (VT != 0) {
        if (vtile) {   // (workgroup-uniform) this lane: key (i&3)*16 + l16 of the half, d = jt*16 + 8*(lq&1) + e
          const int key = (i & 3) * 16 + l16;
          const int pos = (key & 0x33) | ((key & 4) << 1) | ((key & 8) >> 1);
          const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
          uint16_t* col = vt_lds + (jt * 16 + 8 * (lq & 1)) * 72 + pos;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            uint32_t lo = w4[e2] & 0xffffu, hi16 = w4[e2] >> 16;
            if constexpr (VT == 1 && ODT != TD_F16) {
              float f0, f1;
              unpack2<ODT>(w4[e2], f0, f1);
              lo = f32_to_half_bits<TD_F16>(f0);
              hi16 = f32_to_half_bits<TD_F16>(f1);
            }
            if (m >= M) { lo = 0; hi16 = 0; }
            col[(2 * e2) * 72] = (uint16_t)lo;
            col[(2 * e2 + 1) * 72] = (uint16_t)hi16;
          }
          continue;
        }
      }
      uint4 outv = v;
      if constexpr (RES) {
        if (m < M && n < N) {
          float xf[8], yf[8];
          unpack8<ODT>(xres[i][jp], xf);
          unpack8<ODT>(v, yf);
          if (gate != nullptr) {
            const float* gp = ep_gate + wn * 64 + jt * 16 + 8 * (lq & 1);       // gate.type_as(x), rounded once (LDS)
            const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float t = round_half<ODT>(yf[e] * g[e]);    // y * gate -> x.dtype
              xf[e] = xf[e] + t;                                // x + t    -> x.dtype (rounded at pack)
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[e] = xf[e] + yf[e];
          }
          outv = pack8<ODT>(xf);
        }
      }
      if constexpr (STATS) {
        float sv[8];
        unpack8<ODT>(outv, sv);
        // sums of the values SHIFTED by one sample of the row's piece (the first value of the piece's first lane): a row
        // whose mean is large against its spread (DC-heavy channels) loses no digits to sum-of-squares cancellation
        if (jp == 0) st_c = __shfl(sv[0], l16, 64);
        if (m < M && n < N) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = sv[e] - st_c; st_s += d; st_q = fmaf(d, d, st_q); }
        }
      }
      *reinterpret_cast<uint4*>(stg_lds + ((i & 3) * 16 + l16) * 72 + jt * 16 + 8 * (lq & 1)) = outv;
    }
    if ((i & 3) == 3 && !vtile) {   // a 64-row half is complete in LDS: 8 rows x one 128-byte line per store instruction
      {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + (lane >> 3), chunk = lane & 7;
          const int64_t mm = m0 + wm * WROWS + (i >> 2) * 64 + row, nn = n0 + wn * 64 + chunk * 8;
          const uint4 r = *reinterpret_cast<const uint4*>(stg_lds + row * 72 + chunk * 8);
          if (mm < M && nn < N) *reinterpret_cast<uint4*>(D + mm * ldd + nn) = r;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    if constexpr (VT != 0) {
      if (vtile && (i & 3) == 3) {   // a 64-key half is complete in LDS: 8 d rows x 128 B per store instruction
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t Kb = (M + 63) >> 6;
        const int64_t kb = ((m0 + wm * WROWS) >> 6) + (i >> 2);
        const int64_t head = ((n0 - ldqs) >> 7) + (wn >> 1);
        uint16_t* dst = reinterpret_cast<uint16_t*>(QS) + (head * Kb + kb) * (128 * 64) + (int64_t)((wn & 1) * 64) * 64;
        if (kb < Kb) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int d = it * 8 + (lane >> 3), slot = lane & 7;
            const uint4 r = *reinterpret_cast<const uint4*>(vt_lds + d * 72 + slot * 8);
            *reinterpret_cast<uint4*>(dst + d * 64 + slot * 8) = r;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next half overwrites
      }
    }
    if constexpr (STATS) {
      // the row's 64 columns of this wave tile sit in the 4 lanes l16 + 16*lq: fixed-order butterfly, then one 8-byte store
      st_s += __shfl_xor(st_s, 16, 64); st_q += __shfl_xor(st_q, 16, 64);
      st_s += __shfl_xor(st_s, 32, 64); st_q += __shfl_xor(st_q, 32, 64);
      // the piece's 64 values as (mean, M2 = sum of squared deviations from that mean): S, Q are sums of d = v - c, so
      // mean = c + S/64 and M2 = Q - S^2/64 with |S/64| of the order of the piece's own spread (c is one of its values)
      if (lq == 0 && m < M && n0 + wn * 64 < N) {
        const float ds = st_s * (1.0f / 64.0f);
        reinterpret_cast<float2*>(QS)[m * ldqs + ((n0 + wn * 64) >> 6)] = make_float2(st_c + ds, fmaxf(fmaf(-ds, st_s, st_q), 0.f));
      }
    }
  }
  if constexpr (DBG >= 2) {
    // phase stamps of workgroups 0, 256, 512, ... (one per round on CU-slot 0): {start, first stage landed, main loop
    // done, stores issued, stores complete}
    const unsigned long long c_t3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c_t4 = __builtin_amdgcn_s_memtime();
    if ((blockIdx.x & 255) == 0 && wave == 0 && lane == 0 && (blockIdx.x >> 8) < 12) {
      unsigned long long* o = dbg + (blockIdx.x >> 8) * 5;
      o[0] = c_t0; o[1] = c_t1; o[2] = c_t2; o[3] = c_t3; o[4] = c_t4;
    }
    if (wave == 0 && lane == 0 && blockIdx.x < 21000) {   // every workgroup: {start, end, XCC id << 32 | HW_ID}
      unsigned long long* o = dbg + 256 + 3 * (unsigned long long)blockIdx.x;
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      o[0] = c_t0; o[1] = c_t4; o[2] = ((unsigned long long)xcc << 32) | hw;
    }
  }
}

// one launch over the row tiles [tm0, tm0 + tiles_m) (tiles of 32 NI rows; the four-wave form: 128 rows)
template <int ODT, int EPI, bool HAS_BIAS, int DBG, int SCHED, bool QOUT, bool RES, int FAST, bool STATS, int VT, int NI, int NW = 8>
static int launch_gemm_fi_range(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                                const void* bias, void* d, int64_t m, int64_t n, int64_t k, int64_t ldd,
                                hipStream_t st, float* qs, int64_t ldqs, const float* gate, int tm0, int tiles_m) {
  auto kern = gemm_w8a8_fi_kernel<ODT, EPI, HAS_BIAS, DBG, SCHED, QOUT, RES, FAST, STATS, VT, NI, NW>;
  // eight waves: 128 KB + scratch in both tile forms (one workgroup per CU either way); four waves: 64 KB + scratch, two per CU
  constexpr int F_LDS = NW == 4 ? 65536 : 131072;
  const uint16_t* gelu_tab = nullptr;
  if constexpr (QOUT && EPI == TD_EPI_GELU_TANH && ODT == TD_BF16 && NW == 8) {   // (the table takes 128 KB of LDS: eight-wave forms only)
    if (td_tuning(TD_TUNE_GELU_TABLE) != 1) gelu_tab = td_gelu_table_bf16(st);   // 1 = the inline form (cross-check / A-B)
  }
  static std::atomic<uint64_t> attr_mask{0};
  td_ensure_dyn_lds(reinterpret_cast<const void*>(kern), F_LDS + F_DUMP, attr_mask);
  const int tiles_n = (int)td_cdiv(n, F_BN);
  int group_m = td_tuning(TD_TUNE_GEMM_GROUP_M) > 0 ? td_tuning(TD_TUNE_GEMM_GROUP_M) : 4;
  if (NW == 4 && td_tuning(TD_TUNE_GEMM_GROUP_M) <= 0) group_m = 8;   // the same 1024 rows of activations per raster group
  const unsigned nwg = (unsigned)tiles_m * (unsigned)tiles_n;
  // profiling only: row stride of both int8 operands = k + pad (the caller's buffers must be that large)
  const int64_t ldab = k + td_tuning(TD_TUNE_GEMM_LDPAD);
  kern<<<nwg, NW * 64, F_LDS + F_DUMP, st>>>(a, a_s, b, b_s, (const uint16_t*)bias, (uint16_t*)d, m, n, k, ldab, ldab, ldd,
                                tiles_m, tiles_n, group_m, DBG ? td_dbg_buffer() : nullptr, qs, ldqs, gate, gelu_tab, tm0);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// Tile form(s) of a problem.  Time ~ rounds of the 256 CUs x relative tile cost (a 128-row tile: 0.5 of a 256 x 256 one in
// matrix work; measured — tools/gemm_small_m.py, profiles/r05_gemm_small_m.txt, shapes where both forms run whole rounds —
// 0.59-0.60 with the fused epilogues (quantiser, residual, row statistics), 0.79 with the plain 16-bit store).  Three plans:
//   (a) 256-row tiles only                                   ceil(T / 256)
//   (b) 128-row tiles only                                   c x ceil(2T' / 256)
//   (c) MIXED: as many whole rounds of 256-row tiles as fit, the remaining rows on 128-row tiles in a second launch
//       (ffn.0 of a rank of 8: 560 tiles = 2.19 rounds -> 2 rounds + 140 half tiles = 2.6 instead of 3)
// the cheapest wins; (c) must beat the better of (a) / (b) by 5 % to pay for its second launch.  Bit-identical whatever the
// plan.  TD_TUNE_GEMM_VARIANT = 4 forces (a), 6 forces (b), 7 allows only (a) / (b), 8 forces the four-wave form (128-row tiles,
// two workgroups per CU).
// Round 6: the dequant mode is a run-time choice for EVERY epilogue (td_gemm_fast_g() == TD_GEMM_FAST_TWIN re-dispatches the exact
// instantiation's call to its FAST = 4 twin; bf16 + bias — the model's linears); the plans apply to both modes.
template <int ODT, int EPI, bool HAS_BIAS, int DBG = 0, int SCHED = 0, bool QOUT = false, bool RES = false, int FAST = 0, bool STATS = false, int VT = 0, int NI = 8>
static int launch_gemm_fi(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                          const void* bias, void* d, int64_t m, int64_t n, int64_t k, int64_t ldd,
                          hipStream_t st, float* qs = nullptr, int64_t ldqs = 0, const float* gate = nullptr) {
  if constexpr (DBG == 0 && SCHED == 0 && FAST == 0 && ODT == TD_BF16 && HAS_BIAS) {
    if (td_gemm_fast_g() == TD_GEMM_FAST_TWIN)
      return launch_gemm_fi<ODT, EPI, HAS_BIAS, 0, 0, QOUT, RES, TD_GEMM_FAST_TWIN, STATS, VT, NI>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st, qs, ldqs, gate);
  }
  const int tm8 = (int)td_cdiv(m, 256), tm4 = (int)td_cdiv(m, 128), tn = (int)td_cdiv(n, F_BN);
  if constexpr (DBG == 0 && SCHED == 0 && (FAST == 0 || FAST == TD_GEMM_FAST_TWIN)) {
    int v = td_tuning(TD_TUNE_GEMM_VARIANT);
    // the four-wave form by epilogue kind (TD_TUNE_GEMM_W4; profiles/r06_gemm_forms_*.txt, r06_w4_ab.txt)
    constexpr int kind = QOUT ? 1 : (VT != 0 ? 2 : ((RES || STATS) ? 4 : 8));
    int w4 = td_tuning(TD_TUNE_GEMM_W4);
    if (w4 == 0) w4 = TD_GEMM_W4_DEFAULT;
    // (a PLAIN launch as wide as a fused q|k|v projection — the sequence-parallel path's, which packs V itself — is the same
    //  main loop as the V^T-epilogue one: 45.8 vs 55 us at m = 4096, profiles/r06_gemm_forms_m4096.txt, r06_timeline_emulated_rank_0_of_8.txt)
    const bool wide_plain = kind == 8 && (w4 & 2) && n >= 3072;
    if (v == 8 || (v == 0 && ((w4 & kind) || wide_plain) && m >= 1024))
      return launch_gemm_fi_range<ODT, EPI, HAS_BIAS, 0, 0, QOUT, RES, FAST, STATS, VT, 8, 4>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st, qs, ldqs, gate, 0, tm4);
    if (v == 0 && td_tuning(TD_TUNE_GEMM_COTENANT)) v = 4;   // beside another GEMM: a launch does not own the chip, whole tiles only
    const double c = (!QOUT && !RES && !STATS) ? 0.80 : 0.60;
    auto rounds = [](int64_t tiles) { return (double)td_cdiv(tiles, 256); };
    const double pa = rounds((int64_t)tm8 * tn), pb = c * rounds((int64_t)tm4 * tn);
    // (c): the largest row-tile count whose tiles fill whole rounds
    const int whole = (int)(((int64_t)tm8 * tn) / 256);            // full rounds available
    int m1 = 0;
    double pc = 1e30;
    if (whole >= 1 && v != 4 && v != 6 && v != 7) {
      m1 = (int)(((int64_t)whole * 256) / tn);                    // row tiles (of 256) in the first launch
      if (m1 >= tm8) m1 = tm8 - 1;
      if (m1 >= 1) {
        const int64_t rest4 = (int64_t)(tm4 - 2 * m1) * tn;        // 128-row tiles of the remaining rows
        pc = rounds((int64_t)m1 * tn) + c * rounds(rest4);
      }
    }
    if (v == 6 || (v != 4 && pb < pa && pb <= pc * 1.05))
      return launch_gemm_fi_range<ODT, EPI, HAS_BIAS, 0, 0, QOUT, RES, FAST, STATS, VT, 4>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st, qs, ldqs, gate, 0, tm4);
    if (v != 4 && v != 6 && pc * 1.05 < pa && pc * 1.05 < pb) {
      int rc = launch_gemm_fi_range<ODT, EPI, HAS_BIAS, 0, 0, QOUT, RES, FAST, STATS, VT, 8>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st, qs, ldqs, gate, 0, m1);
      if (rc != TD_OK) return rc;
      return launch_gemm_fi_range<ODT, EPI, HAS_BIAS, 0, 0, QOUT, RES, FAST, STATS, VT, 4>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st, qs, ldqs, gate, 2 * m1, tm4 - 2 * m1);
    }
  }
  return launch_gemm_fi_range<ODT, EPI, HAS_BIAS, DBG, SCHED, QOUT, RES, FAST, STATS, VT, NI>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st, qs, ldqs, gate, 0, tm8);
}

// called by td_gemm_w8a8 (gemm_w8a8.hip) after argument validation; needs ldd % 8 == 0
int td_gemm_w8a8_fi(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                    const void* bias, void* d, int out_dtype, int epilogue, int64_t m, int64_t n,
                    int64_t k, int64_t ldd, hipStream_t st) {
  if (td_tuning(TD_TUNE_GEMM_ABLATE) == 9)  // s_memtime trace instantiation
    return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 1>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
  if (td_tuning(TD_TUNE_GEMM_ABLATE) == 6)  // phase stamps (prologue / main loop / epilogue)
    return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
  if (td_tuning(TD_TUNE_GEMM_ABLATE) == 8)  // s_memtime trace, early-DMA schedule
    return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 1, 1>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
  if (epilogue == TD_EPI_NONE && bias && out_dtype == TD_BF16) {
    switch (td_tuning(TD_TUNE_GEMM_SCHED)) {
      case 1: return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 1>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
      case 2: return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
      case 3: return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 3>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
      default: break;
    }
  }
  if (out_dtype == TD_BF16 && bias && epilogue == TD_EPI_NONE) {  // one-VALU dequant (the model's instantiations only)
    switch (td_gemm_fast_g()) {
      case 2: return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, false, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
      case 4: return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, false, 4>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
      case 8: return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, false, 8>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
      default: break;
    }
  }
  if (td_tuning(TD_TUNE_GEMM_ABLATE) == 7)  // s_memtime trace, L2 prefetch
    return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 1, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
#define TD_GEMM_CASE(ODT)                                                                              \
  if (epilogue == TD_EPI_GELU_TANH) {                                                                  \
    return bias ? launch_gemm_fi<ODT, TD_EPI_GELU_TANH, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)  \
                : launch_gemm_fi<ODT, TD_EPI_GELU_TANH, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st); \
  } else {                                                                                             \
    return bias ? launch_gemm_fi<ODT, TD_EPI_NONE, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)    \
                : launch_gemm_fi<ODT, TD_EPI_NONE, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);  \
  }
  if (out_dtype == TD_BF16) { TD_GEMM_CASE(TD_BF16) } else { TD_GEMM_CASE(TD_F16) }
#undef TD_GEMM_CASE
}

// a15+a16 fused: d_q int8 [m, n] + d_s f32 [ceil(m/128), ceil(n/128)] = quant_block128(cast(gemm(...)+bias [gelu]))
int td_gemm_w8a8_fi_q(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                      int8_t* d_q, float* d_s, int act_dtype, int epilogue, int64_t m, int64_t n, int64_t k,
                      hipStream_t st) {
  const int64_t ldqs = td_cdiv(n, 128);
  if (act_dtype == TD_BF16 && bias && epilogue == TD_EPI_GELU_TANH) {
    switch (td_gemm_fast_g()) {
      case 2: return launch_gemm_fi<TD_BF16, TD_EPI_GELU_TANH, true, 0, 0, true, false, 2>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs);
      case 4: return launch_gemm_fi<TD_BF16, TD_EPI_GELU_TANH, true, 0, 0, true, false, 4>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs);
      case 8: return launch_gemm_fi<TD_BF16, TD_EPI_GELU_TANH, true, 0, 0, true, false, 8>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs);
      default: break;
    }
  }
#define TD_GEMM_CASE(ODT)                                                                                   \
  if (epilogue == TD_EPI_GELU_TANH) {                                                                       \
    return bias ? launch_gemm_fi<ODT, TD_EPI_GELU_TANH, true, 0, 0, true>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs)  \
                : launch_gemm_fi<ODT, TD_EPI_GELU_TANH, false, 0, 0, true>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs); \
  } else {                                                                                                  \
    return bias ? launch_gemm_fi<ODT, TD_EPI_NONE, true, 0, 0, true>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs)       \
                : launch_gemm_fi<ODT, TD_EPI_NONE, false, 0, 0, true>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs);     \
  }
  if (act_dtype == TD_BF16) { TD_GEMM_CASE(TD_BF16) } else { TD_GEMM_CASE(TD_F16) }
#undef TD_GEMM_CASE
}

// a15 + a7 fused: x[m, ldx] += cast(cast(gemm + bias) * cast(gate))   (gate f32 [n] or NULL for a plain add)
int td_gemm_w8a8_fi_res(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                        void* x, const float* gate, int dtype, int64_t m, int64_t n, int64_t k, int64_t ldx,
                        hipStream_t st) {
  if (dtype == TD_BF16 && bias) {
    switch (td_gemm_fast_g()) {
      case 2: return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, true, 2>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate);
      case 4: return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, true, 4>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate);
      case 8: return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, true, 8>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate);
      default: break;
    }
  }
  if (dtype == TD_BF16)
    return bias ? launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, true>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate)
                : launch_gemm_fi<TD_BF16, TD_EPI_NONE, false, 0, 0, false, true>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate);
  return bias ? launch_gemm_fi<TD_F16, TD_EPI_NONE, true, 0, 0, false, true>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate)
              : launch_gemm_fi<TD_F16, TD_EPI_NONE, false, 0, 0, false, true>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate);
}


// a15 (+a7) with the row-statistics partials of the NEXT norm (STATS epilogue): stats_ws float2 [m, n/64].
// x == nullptr: plain GEMM into d (ldd); else the residual form on x (ldx).  BF16 + bias only (the model's linears).
int td_gemm_w8a8_fi_stats(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                          void* d_or_x, const float* gate, int residual, int64_t m, int64_t n, int64_t k, int64_t ld,
                          float* stats_ws, hipStream_t st) {
  const int64_t pieces = n / 64;
  if (residual)
    return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, true, 0, true>(a, a_s, b, b_s, bias, d_or_x, m, n, k, ld, st,
                                                                                 stats_ws, pieces, gate);
  return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, false, 0, true>(a, a_s, b, b_s, bias, d_or_x, m, n, k, ld, st,
                                                                                stats_ws, pieces, nullptr);
}


// a15 of the fused q|k|v projection + the V^T tiles of td_v_transpose (VT epilogue): columns [v_col0, n) are NOT stored
// row-major, they go to vt as [head][ceil(m/64)][128][64] tiles (fp16 when vt_f16, else the output dtype).
int td_gemm_w8a8_fi_vt(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias, void* d,
                       int out_dtype, int64_t m, int64_t n, int64_t k, int64_t ldd, int64_t v_col0, void* vt, int vt_f16,
                       hipStream_t st) {
  float* vq = reinterpret_cast<float*>(vt);
  if (out_dtype == TD_BF16) {
    if (vt_f16) return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, false, 0, false, 1>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st, vq, v_col0);
    return launch_gemm_fi<TD_BF16, TD_EPI_NONE, true, 0, 0, false, false, 0, false, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st, vq, v_col0);
  }
  return launch_gemm_fi<TD_F16, TD_EPI_NONE, true, 0, 0, false, false, 0, false, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st, vq, v_col0);
}
