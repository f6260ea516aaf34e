// a9 / a12 / a13 / a4 — flash-style attention for gfx950, block-sparse (LUT) or dense.
//
// Two instantiations of one kernel:
//   td_attn_i8 : SageAttention arithmetic — INT8 Q.K^T on v_mfma_i32_32x32x32_i8 with
//                per-block scales, fp32 online softmax (exp2 domain), P rounded to fp16,
//                P.V on v_mfma_f32_32x32x16_f16, fp32 accumulate
//                (call site in the reference: SLA/core.py:211-216; arithmetic lives in the
//                un-vendored SpargeAttn dependency — see oracle/sla_ref.py:sage_sparse_attn).
//   td_attn_16 : bf16/f16 Q.K^T and P.V (SLA Triton kernel arithmetic, SLA/kernel.py:21-82;
//                also the dense cross-attention, rcm/utils/attention.py:120-167).
//
// MI355X design: one 256-thread workgroup (4 wavefronts) per 128-row Q block; each wave
// owns 32 Q rows and walks the selected 64-key K blocks.  Both GEMMs are issued TRANSPOSED
// so that every per-row softmax quantity is lane-local:
//   S^T[64 keys x 32 q] = K_tile . Q^T    (A operand = K rows from LDS, B operand = Q in VGPRs)
//   O^T[128 d x 32 q]  += V^T_tile . P^T  (A operand = V^T rows from LDS, B operand = P^T —
//        exactly the registers the first MFMA produced: lane = q row, registers = keys)
// so the row max is a 31-op in-lane reduction plus ONE exchange with lane^32, the rescale
// factor multiplies a lane's accumulators uniformly, and P never touches LDS.  V is
// pre-transposed per 64-key block by td_v_transpose with the keys of each 16-group stored in
// the order the MFMA B fragment delivers P (0-3,8-11 | 4-7,12-15), so a V^T A-fragment is one
// ds_read_b128.  K and V^T tiles sit in LDS with the 16-B slot XOR-swizzled so the 16-lane
// groups of ds_read_b128 are bank-conflict free; tiles arrive by LDS-DMA into a ring of buffers (three for the INT8
// kernel: fetch two iterations ahead of use; one barrier per K block).  Workgroup ids are XCD-remapped so one XCD's L2 serves one head's
// K/V at a time.
#include "td_common.h"

struct AttnParams {
  const void* q;       // int8 [H,L,128] or 16-bit [H,L,128]
  const float* q_s;    // [H, Qb] (int8 only)
  const void* k;       // int8 [H,Lk,128] or 16-bit
  const float* k_s;    // [H, Kb] (int8 only)
  const uint16_t* vt;  // [H, Kb, 128, 64] 16-bit
  const int32_t* lut;  // [H, Qb, nsel] or null
  uint16_t* o;
  int64_t o_stride_h, o_stride_l;
  float scale_log2;    // sm_scale * log2(e)
  int64_t L, Lk;
  int H, Qb, Kb, nsel;
  int64_t k_rows_alloc;  // K rows allocated per head (>= Lk; gathered sequence-parallel layout)
  int kb_alloc;          // K blocks allocated per head in vt / k_s
  float tau;             // lazy running-max threshold (log2 units)
  const uint16_t* add_t;  // optional: 16-bit addend in the lane-private layout of td_sla_linear_out_t (o = o_s + o_l)
  int8_t* q_out;          // optional: block-quantised output int8 [L, q_ld] instead of o ...
  float* q_scale;         // ... with scales [ceil(L/128), H]  (== td_quant_i8_block128 of the [L, H*128] output)
  int64_t q_ld;
  int q_heads;            // heads of the WHOLE output row (>= H: a head-group launch writes its H heads' columns / scales at a shifted origin)
  // optional (16-bit kernel): Q read straight from a [L, ld] GEMM output with its RMSNorm applied on load
  int64_t q_stride_h, q_stride_l;   // bytes; 0 = the packed [H, L, 128] layout
  const float* q_rstd;              // [L] 1/rms of the row over the FULL model dim (td_rms_stats), or null
  const float* q_w;                 // [H*128] RMSNorm weight
  // round 6: instead of q_rstd, the row statistic formed HERE from the per-64-column pieces (mean, M2) the producing GEMM's STATS
  // epilogue wrote — td_row_stats_finalize's arithmetic (mode 1) for the row this lane owns, without its launch
  const float* q_pieces;            // float2 [L][q_npieces], or null
  int q_npieces;
  float q_inv_n, q_eps;
  const float* v_scale;             // FP8-PV instantiation: per-(h, d) channel scale of the e4m3 V tiles [H, 128]
  // sequence-parallel GATHERED layout (kbp > 0): the K side is the output of an all-gather, rank-major — K block kb lives
  // on rank r = kb / kbp as that rank's block kb % kbp: K rows at k + r*k_rs + (h*kbp*64 + ...)*row bytes, scales at
  // k_s + r*ks_rs + h*kbp + ..., V^T tiles at vt + r*v_rs + (h*kbp + ...)*tile bytes.  k_rows_alloc = kbp*64 and
  // kb_alloc = kbp then describe ONE rank's part, so the per-head bases below need no change.
  int kbp;
  int64_t k_rs, v_rs, ks_rs;        // rank strides: bytes, bytes, floats
  // STAMP instantiations only (TD_TUNE_ATTN_OCC = 9, tools/attn_phases.py): s_memtime per workgroup {entry, Q ready + first
  // tile landed, K loop done, epilogue stores issued, hardware id}.  (As a run-time option of the production kernels the
  // stamps cost a spill and 50-70 us in situ, hence separate instantiations.)
  unsigned long long* dbg;
};

template <bool QK_I8> struct KTile {
  static constexpr int ROWB = QK_I8 ? 128 : 256;    // bytes per key row
  static constexpr int BYTES = 64 * ROWB;
  static constexpr int NVEC = BYTES / (256 * 16);   // 16-B vectors per thread
  __device__ static __forceinline__ uint32_t off(uint32_t row, uint32_t slot) {
    if constexpr (QK_I8) return row * 128u + ((slot ^ ((row >> 1) & 7u)) << 4);
    else return row * 256u + ((slot ^ (row & 15u)) << 4);
  }
};
typedef __attribute__((address_space(3))) void* lptr_a;
#define VT_BYTES (128 * 128)
#define VT8_BYTES (128 * 64)   // FP8-PV: e4m3 V^T tile [128 d][64 positions]
// FP8-PV: row = 64 bytes = 4 slots; (row >> 2) & 3 spreads the 16-lane groups of ds_read_b128 over all banks
__device__ __forceinline__ uint32_t vt8_off(uint32_t row, uint32_t slot) {
  return row * 64u + ((slot ^ ((row >> 2) & 3u)) << 4);
}
typedef int v8i_t __attribute__((ext_vector_type(8)));
// FP8-PV: P is scaled by PV8_C before the e4m3 conversion (the row sum carries the same factor, so it cancels); with the
// lazy running max P <= 2^PV8_TAU = 2, and PV8_C * 2 = 447 stays inside e4m3 (max 448).  The reference's kernels scale
// by 448 against the exact running max (oracle/sla_ref.py: sage_sparse_attn_fp8) — the same representation up to where
// in e4m3's uniform relative grid a value lands.
#define PV8_TAU 1.0f
#define PV8_LOG2C 7.8041310f   // log2(447 / 2)
// INT8 QK chains start from the INLINE constant 1/(2 pi) = 0x3E22F983 as the MFMA's C operand (no 16-register constant):
// |sum| <= 128 * 127 * 127 < 0x22F983 keeps 0x3E22F983 + sum inside the binade [0.125, 0.25), whose ulp is 2^-26 — the int32
// result read as fp32 is A_INV2PI_F + sum * 2^-26 ("raw"), monotone in the score.
#define A_INV2PI_F 0.15915494309189532f
#define A_QK_MFMA0(acc_, a_, b_) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0.15915494" : "=&v"(acc_) : "v"(a_), "v"(b_));
__device__ __forceinline__ uint32_t vt_off(uint32_t row, uint32_t slot) {
  return row * 128u + ((slot ^ ((row >> 1) & 7u)) << 4);
}

template <int DT> struct Mma16;
template <> struct Mma16<TD_F16> {
  typedef v8h frag;
  __device__ static __forceinline__ v16f mma(frag a, frag b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma16<TD_BF16> {
  typedef v8bf frag;
  __device__ static __forceinline__ v16f mma(frag a, frag b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// QK_I8: int8 QK; PDT: dtype of P / V^T (and of q,k when !QK_I8); ODT: output dtype
// OCC2 (INT8 kernels, experiment TD_TUNE_ATTN_OCC): two workgroups per CU with 256 VGPRs each instead of three with 168 —
// room for THREE tile buffers and for explicit fragment prefetch: all 8 K fragments of a tile are requested before the
// first QK MFMA (the 168-register build reuses one fragment register: ds_read -> wait -> MFMA, eight times), and the V
// fragments of d-block c+1 are requested before the MFMAs of d-block c.
// ROWSUM (experiment TD_TUNE_ATTN_OCC = 4, on the OCC2 build: it needs 20 more registers than three workgroups per CU leave):
// the softmax denominator on the MATRIX pipe — P is already rounded to 16 bits for O^T += V^T . P^T; four more MFMAs per
// tile against an all-ones A operand accumulate sum_k P[k][q] into one more accumulator block (every row of it is the row
// sum, complete over both half-waves), and the 32 v_add per lane and tile of `psum` leave the VALU stream.  The denominator
// is then the sum of the ROUNDED probabilities (what the numerator uses); results equal to rounding, not bit-identical.
template <bool QK_I8, int PDT, int ODT, bool PV8 = false, bool OCC2 = false, bool STAMP = false, bool ROWSUM = false, bool DOT2 = false>
// lut_all / ks_all / qs_all repeat p.lut / p.k_s / p.q_s as __restrict__ kernel arguments: only then are the per-iteration
// LUT entry and K scale SCALAR loads (s_load, lgkmcnt).  As vector loads they drag an s_waitcnt vmcnt(0) into the loop,
// which waits for every K/V tile in flight and undoes the fetch-ahead.
__global__ __launch_bounds__(256, (QK_I8 && !OCC2) ? 3 : 2) void attn_kernel(AttnParams p, const int32_t* __restrict__ lut_all,
                                                      const float* __restrict__ ks_all,
                                                      const float* __restrict__ qs_all) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef KTile<QK_I8> KT;
  constexpr int VTB = PV8 ? VT8_BYTES : VT_BYTES;
  constexpr int BUF = KT::BYTES + VTB;
  static_assert(!PV8 || QK_I8, "FP8 PV is the SageAttention variant");
  typedef typename Mma16<PDT>::frag frag16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  unsigned long long a_t0 = 0, a_t1 = 0, a_t2 = 0;
  if constexpr (STAMP) a_t0 = __builtin_amdgcn_s_memtime();

  const uint32_t vid = xcd_remap(blockIdx.x, gridDim.x);
  const int h = vid / p.Qb, qb = vid % p.Qb;

  int64_t qrow = (int64_t)qb * 128 + wave * 32 + li;
  const bool q_ok = qrow < p.L;
  if (!q_ok) qrow = p.L - 1;

  // ---- Q fragments (B operand): lane = q row, 16 B per 32-byte k-chunk half ----
  constexpr int NQ = QK_I8 ? 4 : 8;
  uint4 qf[NQ];
  {
    const char* qp = (const char*)p.q + ((int64_t)h * p.L + qrow) * (QK_I8 ? 128 : 256);
    if constexpr (!QK_I8) {
      if (p.q_stride_l != 0) qp = (const char*)p.q + (int64_t)h * p.q_stride_h + qrow * p.q_stride_l;
    }
#pragma unroll
    for (int kc = 0; kc < NQ; ++kc) qf[kc] = *reinterpret_cast<const uint4*>(qp + kc * 32 + hi * 16);
    if constexpr (!QK_I8) {
      if (p.q_rstd != nullptr || p.q_pieces != nullptr) {
        // q = cast(rmsnorm(x) * w) exactly as td_qk_norm_rope computes it (no RoPE: cross-attention), applied to the 64
        // elements this lane holds — the head-major normalised copy of Q is never written
        float rs;
        if (p.q_pieces != nullptr) {
          // row_stats_finalize_kernel (norm_quant.hip), mode 1, in ONE lane: its eight lanes' in-order partial sums over the
          // pieces j, j + 8, ... (each piece contributes M2 + 64 mean^2), then its 3-step butterfly's association
          // ((s0+s1)+(s2+s3)) + ((s4+s5)+(s6+s7)) — the same additions in the same order: the same bits
          const float2* pc = reinterpret_cast<const float2*>(p.q_pieces) + (int64_t)qrow * p.q_npieces;
          float sj[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            sj[j] = 0.f;
            for (int pi = j; pi < p.q_npieces; pi += 8) {
              const float2 v = pc[pi];
              sj[j] += fmaf(64.0f * v.x, v.x, v.y);
            }
          }
          const float qsum = ((sj[0] + sj[1]) + (sj[2] + sj[3])) + ((sj[4] + sj[5]) + (sj[6] + sj[7]));
          rs = 1.0f / sqrtf(qsum * p.q_inv_n + p.q_eps);
        } else {
          rs = p.q_rstd[qrow];
        }
#pragma unroll
        for (int kc = 0; kc < NQ; ++kc) {
          const float* wp_ = p.q_w + h * 128 + kc * 16 + hi * 8;
          const float4 w0 = *reinterpret_cast<const float4*>(wp_), w1 = *reinterpret_cast<const float4*>(wp_ + 4);
          const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
          float f[8];
          unpack8<PDT>(qf[kc], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = (f[e] * rs) * wv[e];
          qf[kc] = pack8<PDT>(f);
        }
      }
    }
  }

  const bool has_lut = lut_all != nullptr;
  const int32_t* __restrict__ lut = lut_all + ((int64_t)h * p.Qb + qb) * (has_lut ? p.nsel : 0);  // never read when !has_lut
  const int nsel = has_lut ? p.nsel : p.Kb;
  const float qs = QK_I8 ? qs_all[(int64_t)h * p.Qb + qb] : 1.0f;

  // ---- staging: LDS-DMA (buffer_load_dwordx4 ... lds), a ring of NBUF tile buffers ----
  // A K/V tile fetch is a scattered 24-32 KB read (the LUT picks the blocks).  The INT8 kernel keeps THREE tiles in
  // LDS (72 KB, two workgroups per CU) and fetches two iterations ahead; no staging VGPRs, no ds_write.  The LDS
  // image of a DMA piece is lane-linear, so the bank swizzle of the read side is applied to the global address.
  constexpr int NBUF = OCC2 ? 3 : 2;
  constexpr int KPIECES = KT::BYTES / 1024 / 4;  // per wave: 2 (int8 K) or 4 (16-bit K); V^T: 4
  constexpr int VPIECES = PV8 ? 2 : 4;
  constexpr int NPIECES = KPIECES + VPIECES;
  constexpr uint32_t K_ROWB = QK_I8 ? 128u : 256u;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.k + (int64_t)h * p.k_rows_alloc * KT::ROWB), 0, 0x7fffffff, 0x00020000);
  const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.vt + (int64_t)h * p.kb_alloc * VTB), 0, 0x7fffffff, 0x00020000);
  int krow[KPIECES];       // row of the K tile this lane fetches, per piece
  uint32_t kchunk[KPIECES], voffs[4];
#pragma unroll
  for (int t = 0; t < KPIECES; ++t) {
    const int c = wave_u + 4 * t;
    if constexpr (QK_I8) { krow[t] = 8 * c + (lane >> 3); kchunk[t] = (uint32_t)(((lane & 7) ^ ((krow[t] >> 1) & 7)) * 16); }
    else { krow[t] = 4 * c + (lane >> 4); kchunk[t] = (uint32_t)(((lane & 15) ^ (krow[t] & 15)) * 16); }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if constexpr (PV8) {   // piece = 16 rows x 64 B; LDS position (row, p) holds source slot p ^ ((row >> 2) & 3)
      const int row = 16 * (wave_u + 4 * t) + (lane >> 2);
      voffs[t] = (uint32_t)(row * 64 + (((lane & 3) ^ ((row >> 2) & 3)) * 16));
    } else {
      const int row = 8 * (wave_u + 4 * t) + (lane >> 3);
      voffs[t] = (uint32_t)(row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) * 16));
    }
  }
#define TISSUE(kb_, buf_)                                                                              \
  {                                                                                                    \
    char* base_ = smem + (buf_) * BUF;                                                                 \
    int wb_ = (kb_);                       /* block index within its rank's part (flat layout: the block id) */ \
    uint32_t ksoff_ = 0u, vsoff_ = 0u;     /* rank offsets, through the instruction's scalar offset */ \
    if (p.kbp > 0) {                                                                                   \
      const int r_ = (kb_) / p.kbp;                                                                    \
      wb_ = (kb_) - r_ * p.kbp;                                                                        \
      ksoff_ = (uint32_t)(r_ * p.k_rs);                                                                \
      vsoff_ = (uint32_t)(r_ * p.v_rs);                                                                \
    }                                                                                                  \
    const int64_t lastrow_ = p.Lk - 1 - (int64_t)(kb_) * 64;   /* rows of block kb_ past this one are past Lk */ \
    _Pragma("unroll") for (int t = 0; t < KPIECES; ++t) {                                              \
      int64_t kr_ = krow[t];                                                                           \
      if (kr_ > lastrow_) kr_ = lastrow_;                                                              \
      const uint32_t vo_ = (uint32_t)((int64_t)wb_ * 64 + kr_) * K_ROWB + kchunk[t]; /* (a temporary: an expression here loses the host stub) */ \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, (lptr_a)(base_ + (wave_u + 4 * t) * 1024), 16, \
                                               vo_, ksoff_, 0, 0);                                     \
    }                                                                                                  \
    _Pragma("unroll") for (int t = 0; t < VPIECES; ++t)                                                \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (lptr_a)(base_ + KT::BYTES + (wave_u + 4 * t) * 1024), 16, \
                                               voffs[t], vsoff_ + (uint32_t)wb_ * VTB, 0, 0);          \
  }
#define TWAIT(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory");

  v16f oacc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[c][r] = 0.f;
  float m_run = -INFINITY, l_part = 0.f;
  v16f lacc;
  frag16 ones16;
  if constexpr (ROWSUM) {
    static_assert(!ROWSUM || (OCC2 && !PV8), "the row-sum-by-MFMA experiment lives on the two-workgroup build");
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones16[e] = 1.0f;
  }

  // the Q fragments (plain loads) are older than every DMA piece, so the vmcnt waits below cover them too
  TISSUE(has_lut ? lut[0] : 0, 0)
  if (NBUF == 3 && nsel > 1) {
    TISSUE(has_lut ? lut[1] : 1, 1)
    if constexpr (NPIECES == 6) TWAIT(6) else TWAIT(8)
  } else {
    TWAIT(0)
  }
  __syncthreads();

  if constexpr (STAMP) a_t1 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < nsel; ++it) {
    const int cur = it % NBUF;
    const int kb = has_lut ? lut[it] : it;
    // fetch NBUF-1 tiles ahead into the buffer whose last readers passed the barrier at the end of iteration it-1
    if (it + NBUF - 1 < nsel) {
      const int nb_ = has_lut ? lut[it + NBUF - 1] : it + NBUF - 1;
      TISSUE(nb_, (it + NBUF - 1) % NBUF)
    }
    const char* kt = smem + cur * BUF;
    const char* vtile = kt + KT::BYTES;

    // ---- S^T = K . Q^T : two 32-key groups ----
    // INT8 path: every 4-MFMA chain starts from C = the inline constant 1/(2 pi) (A_INV2PI_F above), so the int32 result
    // reinterpreted as fp32 IS A_INV2PI_F + sum * 2^-26 — "raw".  raw is monotone in the score, so the row max is taken on
    // raw, and the softmax argument (sum*mult - m) is ONE fma per element: fma(raw, mult * 2^26, -(A_INV2PI_F * mult * 2^26 + m)).
    // The folded constant is ~1e3 with an ulp of ~1e-4 (log2 domain): a common factor per (row, K block) of relative
    // size < 1e-4, below the fp16 rounding of P.  No v_cvt, no separate scale multiply, no subtract.
    float s[2][16];
    float mult = p.scale_log2;
    if constexpr (QK_I8) {
      int64_t ksi = (int64_t)h * p.kb_alloc + kb;
      if (p.kbp > 0) { const int r_ = kb / p.kbp; ksi = (int64_t)r_ * p.ks_rs + (int64_t)h * p.kbp + (kb - r_ * p.kbp); }
      mult = ((qs * ks_all[ksi]) * p.scale_log2) * 67108864.0f;   // x 2^26: raw carries the sum in units of 2^-26
      if constexpr (OCC2) {
        v4i kfr[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) kfr[g][kc] = *reinterpret_cast<const v4i*>(kt + KT::off(32 * g + li, 2 * kc + hi));
        __builtin_amdgcn_sched_barrier(0);   // all eight reads are in flight before the first MFMA
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          v16i acc;
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) {
            v4i qv; qv[0] = qf[kc].x; qv[1] = qf[kc].y; qv[2] = qf[kc].z; qv[3] = qf[kc].w;
            if (kc == 0) A_QK_MFMA0(acc, kfr[g][0], qv)
            else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kfr[g][kc], qv, acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) s[g][r] = __int_as_float(acc[r]);
        }
      } else {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        v16i acc;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const v4i kf = *reinterpret_cast<const v4i*>(kt + KT::off(32 * g + li, 2 * kc + hi));
          v4i qv; qv[0] = qf[kc].x; qv[1] = qf[kc].y; qv[2] = qf[kc].z; qv[3] = qf[kc].w;
          if (kc == 0) A_QK_MFMA0(acc, kf, qv)
          else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[g][r] = __int_as_float(acc[r]);
      }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          const frag16 kf = *reinterpret_cast<const frag16*>(kt + KT::off(32 * g + li, 2 * kc + hi));
          frag16 qv = *reinterpret_cast<const frag16*>(&qf[kc]);
          acc = Mma16<PDT>::mma(kf, qv, acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[g][r] = acc[r];
      }
    }
    // ---- tail mask (only the last, partial K block) ----
    if ((int64_t)(kb + 1) * 64 > p.Lk) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t key = (int64_t)kb * 64 + 32 * g + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= p.Lk) s[g][r] = -INFINITY;
        }
    }
    // ---- online softmax, exp2 domain; lanes l and l^32 share a q row ----
    float mx = s[0][0];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[g][r]);
    {   // the other half-wave's maximum of the same q row: v_permlane32_swap (VALU) instead of a trip through the LDS crossbar
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    constexpr float OFFS = QK_I8 ? A_INV2PI_F : 0.0f;
    mx = (mx - OFFS) * mult;  // the scaled row max of this block (exact subtraction)
    // lazy running max: the reference point of the exponentials only moves when the row max grows by more than 2^8
    // — softmax is invariant to it as long as numerator and denominator use the same one; P stays <= 256 (fp16-safe)
    // and the 64 accumulator rescales per lane are skipped for almost every K block
    const float m_new = (mx > m_run + (PV8 ? fminf(p.tau, PV8_TAU) : p.tau)) ? mx : m_run;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first block: exp2(-inf) = 0
    m_run = m_new;
    const float cc = fmaf(-OFFS, mult, -m_new) + (PV8 ? PV8_LOG2C : 0.0f);
    float psum = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[g][r] = __builtin_amdgcn_exp2f(fmaf(s[g][r], mult, cc));
        if constexpr (!ROWSUM && !DOT2) psum += s[g][r];
      }
    if constexpr (ROWSUM) lacc[0] *= alpha;     // (only row 0 of the ones-product is ever read: every row holds the same sum)
    else l_part = l_part * alpha + psum;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[c][r] *= alpha;
    }
    if constexpr (PV8) {
      // ---- P^T as ONE e4m3 B operand: byte j = 16g + r of this half-wave's 32 key positions ----
      v8i_t p8;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const int g = w >> 2, r = 4 * (w & 3);
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(s[g][r], s[g][r + 1], 0, false);
        p8[w] = __builtin_amdgcn_cvt_pk_fp8_f32(s[g][r + 2], s[g][r + 3], pk, true);
      }
      // ---- O^T += V8^T . P8^T : one 32x32x64 MFMA per 32-row d block ----
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const v4i a0 = *reinterpret_cast<const v4i*>(vtile + vt8_off(32 * c + li, 2 * hi));
        const v4i a1 = *reinterpret_cast<const v4i*>(vtile + vt8_off(32 * c + li, 2 * hi + 1));
        v8i_t a;
        a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3]; a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
        oacc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, p8, oacc[c], 0, 0, 0, 0, 0, 0);
      }
    } else {
    // ---- P^T fragments (B operand): step ks = 2g+t uses registers 8t..8t+7 of group g ----
    uint4 pf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int g = ks >> 1, t = ks & 1;
      pf[ks] = pack8<PDT>(&s[g][8 * t]);
    }
    if constexpr (DOT2) {
      // round-5 experiment (TD_TUNE_ATTN_OCC = 5): the row sum from the ROUNDED probabilities — the values the PV product uses —
      // two per instruction: v_dot2_f32_f16 against (1, 1), fp32 accumulate; 16 instead of 32 VALU per lane and tile
      typedef _Float16 h2v __attribute__((ext_vector_type(2)));
      const h2v ones2 = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint32_t w4[4] = {pf[ks].x, pf[ks].y, pf[ks].z, pf[ks].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) psum = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2v, w4[e]), ones2, psum, false);
      }
      l_part += psum;     // (l_part was scaled by alpha above with psum = 0)
    }
    // ---- O^T += V^T . P^T ----
    if constexpr (OCC2) {
      frag16 vfr[2][4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) vfr[0][ks] = *reinterpret_cast<const frag16*>(vtile + vt_off(li, 2 * ks + hi));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c + 1 < 4) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            vfr[(c + 1) & 1][ks] = *reinterpret_cast<const frag16*>(vtile + vt_off(32 * (c + 1) + li, 2 * ks + hi));
        }
        __builtin_amdgcn_sched_barrier(0);   // the next d-block's reads are issued before this d-block's MFMAs
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          frag16 pv = *reinterpret_cast<const frag16*>(&pf[ks]);
          oacc[c] = Mma16<PDT>::mma(vfr[c & 1][ks], pv, oacc[c]);
        }
        if constexpr (ROWSUM) {   // one of the four row-sum MFMAs behind each d block's four
          frag16 pv = *reinterpret_cast<const frag16*>(&pf[c]);
          lacc = Mma16<PDT>::mma(ones16, pv, lacc);
        }
      }
    } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const frag16 vf = *reinterpret_cast<const frag16*>(vtile + vt_off(32 * c + li, 2 * ks + hi));
        frag16 pv = *reinterpret_cast<const frag16*>(&pf[ks]);
        oacc[c] = Mma16<PDT>::mma(vf, pv, oacc[c]);
      }
    }
    }
    }
    // tile it+1 must have landed (this wave's pieces; the barrier makes it everyone's); tile it+2 may stay in flight
    if (NBUF == 3 && it + 2 < nsel) {
      if constexpr (NPIECES == 6) TWAIT(6) else TWAIT(8)
    } else {
      TWAIT(0)
    }
    __syncthreads();
  }

  if constexpr (STAMP) a_t2 = __builtin_amdgcn_s_memtime();
  // ---- epilogue: lane = q row; oacc[c][r] is d = 32c + (r&3) + 8(r>>2) + 4hi ----
  // The tile [128 tokens x 128 d] is transposed through LDS (K/V buffers are free: the loop ended on a barrier) so
  // that global memory sees 16-byte row-contiguous accesses (a lane-per-row store touches 64 cache lines per
  // instruction), the linear branch's o_l is added from its lane-private layout (a coalesced read instead of a
  // strided read-modify-write pass), and — optionally — the tile, which is exactly one 128x128 quantisation block of
  // the [L, H*128] attention output, is block-quantised for the o projection right here.
  const float l_tot = ROWSUM ? lacc[0] : l_part + __shfl_xor(l_part, 32, 64);
  const float inv = 1.0f / l_tot;
  uint2 addv[16];
  if (p.add_t) {
    const uint2* ap = reinterpret_cast<const uint2*>(p.add_t) + (((int64_t)h * p.Qb + qb) * 4 + wave) * 16 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) addv[i] = ap[i * 64];
  }
  constexpr int SROW = 272;  // bytes per staged token row (256 + 16: conflict-free 8-byte writes and 16-byte reads)
  {
    char* srow = smem + (wave * 32 + li) * SROW;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float e0 = oacc[c][4 * g4] * inv, e1 = oacc[c][4 * g4 + 1] * inv, e2 = oacc[c][4 * g4 + 2] * inv, e3 = oacc[c][4 * g4 + 3] * inv;
        if constexpr (PV8) {   // the V channel scale, fused here (…fuse_v_scale… kernels, SLA/core.py:227-239)
          const float4 vs = *reinterpret_cast<const float4*>(p.v_scale + h * 128 + 32 * c + 8 * g4 + 4 * hi);
          e0 *= vs.x; e1 *= vs.y; e2 *= vs.z; e3 *= vs.w;
        }
        uint32_t w0 = pack2<ODT>(e0, e1);      // o_s in the output dtype
        uint32_t w1 = pack2<ODT>(e2, e3);
        if (p.add_t) {                                                                    // o = o_s + o_l (16-bit add)
          float a0, a1, a2, a3, b0, b1, b2, b3;
          unpack2<ODT>(w0, a0, a1); unpack2<ODT>(w1, a2, a3);
          unpack2<ODT>(addv[c * 4 + g4].x, b0, b1); unpack2<ODT>(addv[c * 4 + g4].y, b2, b3);
          w0 = pack2<ODT>(a0 + b0, a1 + b1);
          w1 = pack2<ODT>(a2 + b2, a3 + b3);
        }
        if (!q_ok) { w0 = 0u; w1 = 0u; }  // rows past L: zero (they only matter to the block amax)
        *reinterpret_cast<uint2*>(srow + (32 * c + 8 * g4 + 4 * hi) * 2) = make_uint2(w0, w1);
      }
  }
  __syncthreads();
  uint4 tv[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = tid + 256 * it, row = idx >> 4, ch = idx & 15;
    tv[it] = *reinterpret_cast<const uint4*>(smem + row * SROW + ch * 16);
  }
  if (p.q_out == nullptr) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = tid + 256 * it, row = idx >> 4, ch = idx & 15;
      const int64_t tok = (int64_t)qb * 128 + row;
      if (tok < p.L) *reinterpret_cast<uint4*>(p.o + (int64_t)h * p.o_stride_h + tok * p.o_stride_l + ch * 8) = tv[it];
    }
  } else {
    // per-128x128-block INT8 quantiser (quant.hip semantics: amax >= 1e-8, 128/amax, RNE, saturate; scale = amax/128)
    uint32_t mxb = 0u;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const uint32_t w[4] = {tv[it].x, tv[it].y, tv[it].z, tv[it].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t a = w[e] & 0x7fff7fffu;
        asm("v_pk_max_u16 %0, %0, %1" : "+v"(mxb) : "v"(a));
      }
    }
    uint32_t m16 = max(mxb & 0xffffu, mxb >> 16);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m16 = max(m16, (uint32_t)__shfl_xor((int)m16, o, 64));
    __syncthreads();  // everyone has read the staged tile; reuse its first words for the 4 wave maxima
    uint32_t* red = reinterpret_cast<uint32_t*>(smem);
    if (lane == 0) red[wave] = m16;
    __syncthreads();
    m16 = max(max(red[0], red[1]), max(red[2], red[3]));
    const float amax = fmaxf(half_bits_to_f32<ODT>(m16), 1e-8f);
    const float mult = 128.0f / amax;
    if (tid == 0) p.q_scale[(int64_t)qb * p.q_heads + h] = amax / 128.0f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = tid + 256 * it, row = idx >> 4, ch = idx & 15;
      const int64_t tok = (int64_t)qb * 128 + row;
      if (tok >= p.L) continue;
      float f[8];
      unpack8<ODT>(tv[it], f);
      uint32_t wd[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = rintf(f[j] * mult);  // RNE
        t = fminf(fmaxf(t, -128.0f), 127.0f);
        wd[j >> 2] |= ((uint32_t)(int)t & 0xffu) << (8 * (j & 3));
      }
      *reinterpret_cast<uint2*>(p.q_out + tok * p.q_ld + (int64_t)h * 128 + ch * 8) = make_uint2(wd[0], wd[1]);
    }
  }
  if constexpr (STAMP) {
    if (p.dbg != nullptr && tid == 0 && blockIdx.x < 5000) {
      unsigned long long* o = p.dbg + 256 + 5 * (unsigned long long)blockIdx.x;
      o[0] = a_t0; o[1] = a_t1; o[2] = a_t2; o[3] = __builtin_amdgcn_s_memtime();
      o[4] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Q64 build of the INT8-QK / FP16-PV kernel (TD_TUNE_ATTN_OCC = 3): the same workgroup (128 Q rows, 4 waves, the same tiles,
// DMA pieces and epilogue) with the waves arranged 2 (Q halves of 64 rows) x 2 (key halves of 32 keys) instead of 4 x 32 rows:
// a wave's K fragment (32 keys) and V^T fragments (32 keys) each feed TWO MFMAs (its two 32-row Q blocks), so a tile costs a
// wave 12 ds_read_b128 instead of 24 for the same 24 MFMAs, and the two Q blocks are two independent MFMA chains.  The price:
// 128 accumulator registers (two workgroups per CU, three tile buffers), a running max / row sum per key half — every wave
// is a split-K flash-attention stream over its half of the keys — and one exchange through LDS at the end: wave (wq, wk)
// hands the partial of Q block 1 - wk to wave (wq, 1 - wk) and finishes Q block wk, i.e. the 32 rows 64 wq + 32 wk + lane
// that wave 2 wq + wk of the 4 x 32 kernel owns, so the epilogue below is that kernel's.
// The QK chains start from the INLINE constant 1/(2 pi) = 0x3E22F983 (gemm_w8a8_m32.hip): |sum| <= 128 * 127 * 127 < 0x22F983
// keeps 0x3E22F983 + sum inside the binade [0.125, 0.25) (ulp 2^-26): the int32 result read as fp32 is M' + sum * 2^-26,
// monotone in the score, and the softmax argument is one fma(raw, mult * 2^26, -(M' * mult * 2^26 + m)) as in the 4 x 32 kernel,
// without its 16-register C operand.
// ------------------------------------------------------------------------------------------------------------------
template <int ODT>
__device__ __forceinline__ void attn_tile_out(const AttnParams& p, char* smem, int h, int qb, int tid, int lane, int wave) {
  // staged [128 tokens][272 B] tile in LDS -> global: 16-byte row-contiguous stores, or the 128x128 block quantiser
  constexpr int SROW = 272;
  uint4 tv[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = tid + 256 * it, row = idx >> 4, ch = idx & 15;
    tv[it] = *reinterpret_cast<const uint4*>(smem + row * SROW + ch * 16);
  }
  if (p.q_out == nullptr) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = tid + 256 * it, row = idx >> 4, ch = idx & 15;
      const int64_t tok = (int64_t)qb * 128 + row;
      if (tok < p.L) *reinterpret_cast<uint4*>(p.o + (int64_t)h * p.o_stride_h + tok * p.o_stride_l + ch * 8) = tv[it];
    }
    return;
  }
  uint32_t mxb = 0u;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t w[4] = {tv[it].x, tv[it].y, tv[it].z, tv[it].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t a = w[e] & 0x7fff7fffu;
      asm("v_pk_max_u16 %0, %0, %1" : "+v"(mxb) : "v"(a));
    }
  }
  uint32_t m16 = max(mxb & 0xffffu, mxb >> 16);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m16 = max(m16, (uint32_t)__shfl_xor((int)m16, o, 64));
  __syncthreads();
  uint32_t* red = reinterpret_cast<uint32_t*>(smem);
  if (lane == 0) red[wave] = m16;
  __syncthreads();
  m16 = max(max(red[0], red[1]), max(red[2], red[3]));
  const float amax = fmaxf(half_bits_to_f32<ODT>(m16), 1e-8f);
  const float mult = 128.0f / amax;
  if (tid == 0) p.q_scale[(int64_t)qb * p.q_heads + h] = amax / 128.0f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = tid + 256 * it, row = idx >> 4, ch = idx & 15;
    const int64_t tok = (int64_t)qb * 128 + row;
    if (tok >= p.L) continue;
    float f[8];
    unpack8<ODT>(tv[it], f);
    uint32_t wd[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = rintf(f[j] * mult);
      t = fminf(fmaxf(t, -128.0f), 127.0f);
      wd[j >> 2] |= ((uint32_t)(int)t & 0xffu) << (8 * (j & 3));
    }
    *reinterpret_cast<uint2*>(p.q_out + tok * p.q_ld + (int64_t)h * 128 + ch * 8) = make_uint2(wd[0], wd[1]);
  }
}

template <int ODT>
__global__ __launch_bounds__(256, 2) void attn_i8_q64_kernel(AttnParams p, const int32_t* __restrict__ lut_all,
                                                             const float* __restrict__ ks_all,
                                                             const float* __restrict__ qs_all) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef KTile<true> KT;
  constexpr int VTB = VT_BYTES;
  constexpr int BUF = KT::BYTES + VTB;
  constexpr int NBUF = 3;
  typedef typename Mma16<TD_F16>::frag frag16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wq = wave_u & 1, wk = wave_u >> 1;
  const uint32_t vid = xcd_remap(blockIdx.x, gridDim.x);
  const int h = vid / p.Qb, qb = vid % p.Qb;

  // ---- Q fragments of the wave's two 32-row blocks (B operands) ----
  uint4 qf[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int64_t qrow = (int64_t)qb * 128 + wq * 64 + j * 32 + li;
    if (qrow >= p.L) qrow = p.L - 1;
    const char* qp = (const char*)p.q + ((int64_t)h * p.L + qrow) * 128;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) qf[j][kc] = *reinterpret_cast<const uint4*>(qp + kc * 32 + hi * 16);
  }

  const bool has_lut = lut_all != nullptr;
  const int32_t* __restrict__ lut = lut_all + ((int64_t)h * p.Qb + qb) * (has_lut ? p.nsel : 0);
  const int nsel = has_lut ? p.nsel : p.Kb;
  const float qs = qs_all[(int64_t)h * p.Qb + qb];

  // ---- tile staging: exactly the 4 x 32 kernel's (TISSUE) ----
  constexpr int KPIECES = 2, VPIECES = 4;
  constexpr uint32_t K_ROWB = 128u;
  const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.k + (int64_t)h * p.k_rows_alloc * KT::ROWB), 0, 0x7fffffff, 0x00020000);
  const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.vt + (int64_t)h * p.kb_alloc * VTB), 0, 0x7fffffff, 0x00020000);
  int krow[KPIECES];
  uint32_t kchunk[KPIECES], voffs[4];
#pragma unroll
  for (int t = 0; t < KPIECES; ++t) {
    const int c = wave_u + 4 * t;
    krow[t] = 8 * c + (lane >> 3);
    kchunk[t] = (uint32_t)(((lane & 7) ^ ((krow[t] >> 1) & 7)) * 16);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = 8 * (wave_u + 4 * t) + (lane >> 3);
    voffs[t] = (uint32_t)(row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) * 16));
  }

  v16f oacc[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[j][c][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_part[2] = {0.f, 0.f};

  TISSUE(has_lut ? lut[0] : 0, 0)
  if (nsel > 1) {
    TISSUE(has_lut ? lut[1] : 1, 1)
    TWAIT(6)
  } else {
    TWAIT(0)
  }
  __syncthreads();

  for (int it = 0; it < nsel; ++it) {
    const int cur = it % NBUF;
    const int kb = has_lut ? lut[it] : it;
    if (it + 2 < nsel) {
      const int nb_ = has_lut ? lut[it + 2] : it + 2;
      TISSUE(nb_, (it + 2) % NBUF)
    }
    const char* kt = smem + cur * BUF;
    const char* vtile = kt + KT::BYTES;
    // a key half that lies entirely past Lk (second half of a short last block) contributes nothing
    if ((int64_t)kb * 64 + 32 * wk < p.Lk) {
      int64_t ksi = (int64_t)h * p.kb_alloc + kb;
      if (p.kbp > 0) { const int r_ = kb / p.kbp; ksi = (int64_t)r_ * p.ks_rs + (int64_t)h * p.kbp + (kb - r_ * p.kbp); }
      const float mult = ((qs * ks_all[ksi]) * p.scale_log2) * 67108864.0f;

      // ---- S^T[32 keys x 64 q] = K_half . Q^T : two chains of four MFMAs sharing every K fragment ----
      v4i kf[4];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) kf[kc] = *reinterpret_cast<const v4i*>(kt + KT::off(32 * wk + li, 2 * kc + hi));
      float s[2][16];
      {
        v16i acc[2];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            v4i qv; qv[0] = qf[j][kc].x; qv[1] = qf[j][kc].y; qv[2] = qf[j][kc].z; qv[3] = qf[j][kc].w;
            if (kc == 0) A_QK_MFMA0(acc[j], kf[0], qv)   // (a splat vector C operand of the builtin is materialised in 16 registers)
            else
              acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kc], qv, acc[j], 0, 0, 0);
          }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[j][r] = __int_as_float(acc[j][r]);
      }
      if ((int64_t)(kb + 1) * 64 > p.Lk) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t key = (int64_t)kb * 64 + 32 * wk + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= p.Lk) s[j][r] = -INFINITY;
          }
      }
      // ---- per Q block: online softmax over this wave's 32 keys (lazy running max), then O^T += V^T_half . P^T ----
      uint4 pf[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float mx = s[j][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[j][r]);
        {   // the other half-wave's maximum of the same q row: v_permlane32_swap (VALU) instead of a trip through the LDS crossbar
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
          mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        mx = (mx - A_INV2PI_F) * mult;
        const float m_new = (mx > m_run[j] + p.tau) ? mx : m_run[j];
        const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
        m_run[j] = m_new;
        const float cc = fmaf(-A_INV2PI_F, mult, -m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[j][r] = __builtin_amdgcn_exp2f(fmaf(s[j][r], mult, cc));
          psum += s[j][r];
        }
        l_part[j] = l_part[j] * alpha + psum;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[j][c][r] *= alpha;
        }
        pf[j][0] = pack8<TD_F16>(&s[j][0]);
        pf[j][1] = pack8<TD_F16>(&s[j][8]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const frag16 vf = *reinterpret_cast<const frag16*>(vtile + vt_off(32 * c + li, 2 * (2 * wk + t) + hi));
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            frag16 pv = *reinterpret_cast<const frag16*>(&pf[j][t]);
            oacc[j][c] = Mma16<TD_F16>::mma(vf, pv, oacc[j][c]);
          }
        }
    }
    if (it + 2 < nsel) {
      TWAIT(6)
    } else {
      TWAIT(0)
    }
    __syncthreads();
  }

  // ---- merge the two key halves: wave (wq, wk) finishes Q block wk, its partner (wave ^ 2) the other ----
  float4* xbuf = reinterpret_cast<float4*>(smem);           // per wave: 16 x 64 float4 + 64 float2
  float2* xml = reinterpret_cast<float2*>(smem + 4 * 16384);
  v16f ok[4];
  float m_s, l_s;
#define Q64_GIVE(jg_)                                                                              \
  {                                                                                                \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                  \
      _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4)                                             \
        xbuf[(wave_u * 16 + c * 4 + g4) * 64 + lane] = make_float4(oacc[jg_][c][4 * g4], oacc[jg_][c][4 * g4 + 1], \
                                                                  oacc[jg_][c][4 * g4 + 2], oacc[jg_][c][4 * g4 + 3]); \
    xml[wave_u * 64 + lane] = make_float2(m_run[jg_], l_part[jg_]);                                \
  }
#define Q64_KEEP(jk_)                                                                              \
  {                                                                                                \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) ok[c] = oacc[jk_][c];                            \
    m_s = m_run[jk_]; l_s = l_part[jk_];                                                           \
  }
  if (wk == 0) { Q64_GIVE(1) Q64_KEEP(0) } else { Q64_GIVE(0) Q64_KEEP(1) }
  __syncthreads();
  float l_part1;
  {
    const int pw = wave_u ^ 2;
    const float2 ml = xml[pw * 64 + lane];
    const float m = fmaxf(m_s, ml.x);
    const float a_s = __builtin_amdgcn_exp2f(m_s - m), a_o = __builtin_amdgcn_exp2f(ml.x - m);
    l_part1 = l_s * a_s + ml.y * a_o;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 o4 = xbuf[(pw * 16 + c * 4 + g4) * 64 + lane];
        ok[c][4 * g4] = ok[c][4 * g4] * a_s + o4.x * a_o;
        ok[c][4 * g4 + 1] = ok[c][4 * g4 + 1] * a_s + o4.y * a_o;
        ok[c][4 * g4 + 2] = ok[c][4 * g4 + 2] * a_s + o4.z * a_o;
        ok[c][4 * g4 + 3] = ok[c][4 * g4 + 3] * a_s + o4.w * a_o;
      }
  }
  __syncthreads();   // every partial has been read: the staging area may overwrite the exchange area

  // ---- epilogue of the 4 x 32 kernel with wave -> 2 wq + wk ----
  const int wv = 2 * wq + wk;
  const bool q_ok = (int64_t)qb * 128 + wv * 32 + li < p.L;
  const float l_tot = l_part1 + __shfl_xor(l_part1, 32, 64);
  const float inv = 1.0f / l_tot;
  uint2 addv[16];
  if (p.add_t) {
    const uint2* ap = reinterpret_cast<const uint2*>(p.add_t) + (((int64_t)h * p.Qb + qb) * 4 + wv) * 16 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) addv[i] = ap[i * 64];
  }
  {
    char* srow = smem + (wv * 32 + li) * 272;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float e0 = ok[c][4 * g4] * inv, e1 = ok[c][4 * g4 + 1] * inv, e2 = ok[c][4 * g4 + 2] * inv, e3 = ok[c][4 * g4 + 3] * inv;
        uint32_t w0 = pack2<ODT>(e0, e1);
        uint32_t w1 = pack2<ODT>(e2, e3);
        if (p.add_t) {
          float a0, a1, a2, a3, b0, b1, b2, b3;
          unpack2<ODT>(w0, a0, a1); unpack2<ODT>(w1, a2, a3);
          unpack2<ODT>(addv[c * 4 + g4].x, b0, b1); unpack2<ODT>(addv[c * 4 + g4].y, b2, b3);
          w0 = pack2<ODT>(a0 + b0, a1 + b1);
          w1 = pack2<ODT>(a2 + b2, a3 + b3);
        }
        if (!q_ok) { w0 = 0u; w1 = 0u; }
        *reinterpret_cast<uint2*>(srow + (32 * c + 8 * g4 + 4 * hi) * 2) = make_uint2(w0, w1);
      }
  }
  __syncthreads();
  attn_tile_out<ODT>(p, smem, h, qb, tid, lane, wave_u);
}

template <int ODT>
static int launch_attn_q64(const AttnParams& p_in, hipStream_t st) {
  AttnParams p = p_in;
  p.dbg = nullptr;
  auto kern = attn_i8_q64_kernel<ODT>;
  constexpr int lds = 3 * (KTile<true>::BYTES + VT_BYTES);   // three tile buffers; also holds the 66-KB exchange area
  static_assert(lds >= 4 * 16384 + 4 * 512, "exchange area");
  static std::atomic<uint64_t> attr_mask{0};
  td_ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, attr_mask);
  const unsigned nwg = (unsigned)p.H * (unsigned)p.Qb;
  kern<<<nwg, 256, lds, st>>>(p, p.lut, p.k_s, p.q_s);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

template <bool QK_I8, int PDT, int ODT, bool PV8 = false, bool OCC2 = false, bool STAMP = false, bool ROWSUM = false, bool DOT2 = false>
static int launch_attn(const AttnParams& p_in, hipStream_t st) {
  AttnParams p = p_in;
  p.dbg = nullptr;
  if constexpr (QK_I8 && !PV8 && !OCC2 && !STAMP && !DOT2) {
    if (td_tuning(TD_TUNE_ATTN_OCC) == 2) return launch_attn<QK_I8, PDT, ODT, PV8, true>(p, st);
    if (td_tuning(TD_TUNE_ATTN_OCC) == 3) return launch_attn_q64<ODT>(p, st);
    if (td_tuning(TD_TUNE_ATTN_OCC) == 4) return launch_attn<QK_I8, PDT, ODT, PV8, true, false, true>(p, st);
    if constexpr (PDT == TD_F16) { if (td_tuning(TD_TUNE_ATTN_OCC) == 5) return launch_attn<QK_I8, PDT, ODT, PV8, false, false, false, true>(p, st); }
  }
  if constexpr (!PV8 && !OCC2 && !STAMP && !DOT2 && ODT == TD_BF16 && (QK_I8 || PDT == TD_BF16)) {
    // profiling instantiations of the two kernels the model runs (bf16 outputs)
    if (td_tuning(TD_TUNE_ATTN_OCC) == 9) return launch_attn<QK_I8, PDT, ODT, PV8, false, true>(p, st);
  }
  if constexpr (STAMP) p.dbg = td_dbg_buffer();
  auto kern = attn_kernel<QK_I8, PDT, ODT, PV8, OCC2, STAMP, ROWSUM, DOT2>;
  // two (three: OCC2) tile buffers, and at least the 128 x 272-byte staging area of the epilogue's transpose
  constexpr int lds_tiles = (OCC2 ? 3 : 2) * (KTile<QK_I8>::BYTES + (PV8 ? VT8_BYTES : VT_BYTES));
  constexpr int lds = lds_tiles > 128 * 272 ? lds_tiles : 128 * 272;
  static std::atomic<uint64_t> attr_mask{0};
  td_ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, attr_mask);
  const unsigned nwg = (unsigned)p.H * (unsigned)p.Qb;
  kern<<<nwg, 256, lds, st>>>(p, p.lut, p.k_s, p.q_s);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

static int attn_common_checks(const char* who, const void* q, const void* k, const void* vt, void* o,
                              int nsel, int64_t L, int64_t Lk, int H, const int32_t* lut) {
  TD_REQUIRE(q && k && vt && o, TD_ERR_INVALID, "%s: null pointer", who);
  TD_REQUIRE(L > 0 && Lk > 0 && H > 0, TD_ERR_INVALID, "%s: L=%lld Lk=%lld H=%d", who, (long long)L,
             (long long)Lk, H);
  TD_REQUIRE(!lut || nsel >= 1, TD_ERR_INVALID, "%s: nsel=%d with a LUT", who, nsel);
  return TD_OK;
}
static int attn_stride_check(const char* who, int64_t o_stride_h, int64_t o_stride_l) {
  TD_REQUIRE(o_stride_h % 8 == 0 && o_stride_l % 8 == 0, TD_ERR_UNSUPPORTED,
             "%s: output strides must be multiples of 8 elements (16-byte row-contiguous stores)", who);
  return TD_OK;
}

struct AttnGather { int kbp; int64_t k_rs, v_rs, ks_rs; int q_heads; };
static const AttnGather kFlat = {0, 0, 0, 0, 0};

static int attn_i8_impl(const int8_t* q_i8, const float* q_s, const int8_t* k_i8, const float* k_s,
                             const void* vt, const float* v_scale, const int32_t* lut, int nsel, void* o, int out_dtype,
                             int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L,
                             int64_t Lk, int64_t Lk_alloc, int H, const void* add_t, int8_t* q_out,
                             float* q_scale, td_stream_t stream, const AttnGather& ga = kFlat) {
  int rc = attn_common_checks("td_attn_i8", q_i8, k_i8, vt, q_out ? (void*)q_out : o, nsel, L, Lk, H, lut);
  TD_REQUIRE((q_out == nullptr) == (q_scale == nullptr), TD_ERR_INVALID, "td_attn_i8: q_out/q_scale mismatch");
  if (!q_out) { rc = attn_stride_check("td_attn_i8", o_stride_h, o_stride_l); if (rc) return rc; }
  if (rc) return rc;
  TD_REQUIRE(q_s && k_s, TD_ERR_INVALID, "td_attn_i8: null scale pointer");
  TD_REQUIRE(out_dtype == TD_F16 || out_dtype == TD_BF16, TD_ERR_UNSUPPORTED,
             "td_attn_i8: out dtype %d", out_dtype);
  AttnParams p;
  p.q = q_i8; p.q_s = q_s; p.k = k_i8; p.k_s = k_s; p.vt = (const uint16_t*)vt; p.lut = lut;
  p.o = (uint16_t*)o; p.o_stride_h = o_stride_h; p.o_stride_l = o_stride_l;
  p.add_t = (const uint16_t*)add_t; p.q_out = q_out; p.q_scale = q_scale; p.q_heads = ga.q_heads > 0 ? ga.q_heads : H; p.q_ld = (int64_t)p.q_heads * 128;
  p.q_stride_h = 0; p.q_stride_l = 0; p.q_rstd = nullptr; p.q_w = nullptr; p.v_scale = v_scale;
  p.q_pieces = nullptr; p.q_npieces = 0; p.q_inv_n = 0.f; p.q_eps = 0.f;
  p.kbp = ga.kbp; p.k_rs = ga.k_rs; p.v_rs = ga.v_rs; p.ks_rs = ga.ks_rs;
  p.scale_log2 = sm_scale * 1.4426950408889634f;
  p.L = L; p.Lk = Lk; p.H = H; p.Qb = (int)td_cdiv(L, 128); p.Kb = (int)td_cdiv(Lk, 64); p.nsel = nsel;
  if (ga.kbp > 0) Lk_alloc = (int64_t)ga.kbp * 64;   // one rank's part
  else if (Lk_alloc == 0) Lk_alloc = Lk;
  TD_REQUIRE(ga.kbp > 0 || (Lk_alloc >= Lk && (Lk_alloc == Lk || Lk_alloc % 64 == 0)), TD_ERR_INVALID, "attn: Lk_alloc=%lld", (long long)Lk_alloc);
  p.k_rows_alloc = Lk_alloc; p.kb_alloc = (int)td_cdiv(Lk_alloc, 64);
  p.tau = td_tuning(TD_TUNE_ATTN_TAU) < 0 ? 0.0f : (td_tuning(TD_TUNE_ATTN_TAU) == 0 ? 8.0f : (float)td_tuning(TD_TUNE_ATTN_TAU));
  hipStream_t st = (hipStream_t)stream;
  if (v_scale) {
    if (out_dtype == TD_BF16) return launch_attn<true, TD_F16, TD_BF16, true>(p, st);
    return launch_attn<true, TD_F16, TD_F16, true>(p, st);
  }
  if (out_dtype == TD_BF16) return launch_attn<true, TD_F16, TD_BF16>(p, st);
  return launch_attn<true, TD_F16, TD_F16>(p, st);
}

extern "C" int td_attn_i8_ex(const int8_t* q_i8, const float* q_s, const int8_t* k_i8, const float* k_s,
                             const void* vt, const int32_t* lut, int nsel, void* o, int out_dtype,
                             int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L,
                             int64_t Lk, int64_t Lk_alloc, int H, const void* add_t, int8_t* q_out,
                             float* q_scale, td_stream_t stream) {
  return attn_i8_impl(q_i8, q_s, k_i8, k_s, vt, nullptr, lut, nsel, o, out_dtype, o_stride_h, o_stride_l, sm_scale, L, Lk,
                      Lk_alloc, H, add_t, q_out, q_scale, stream);
}

// a13, FP8-PV variant (the reference's sm89+ branch, SLA/core.py:217-239): the same kernel with P and V in OCP e4m3 and
// the PV contraction on v_mfma_f32_32x32x64_f8f6f4; vt8 / v_scale from td_v_fp8_tiles.
extern "C" int td_attn_i8_fp8pv(const int8_t* q_i8, const float* q_s, const int8_t* k_i8, const float* k_s,
                                const uint8_t* vt8, const float* v_scale, const int32_t* lut, int nsel, void* o,
                                int out_dtype, int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L,
                                int64_t Lk, int64_t Lk_alloc, int H, const void* add_t, int8_t* q_out,
                                float* q_scale, td_stream_t stream) {
  TD_REQUIRE(v_scale, TD_ERR_INVALID, "td_attn_i8_fp8pv: null v_scale");
  return attn_i8_impl(q_i8, q_s, k_i8, k_s, vt8, v_scale, lut, nsel, o, out_dtype, o_stride_h, o_stride_l, sm_scale, L, Lk,
                      Lk_alloc, H, add_t, q_out, q_scale, stream);
}

extern "C" int td_attn_i8(const int8_t* q_i8, const float* q_s, const int8_t* k_i8, const float* k_s,
                          const void* vt, const int32_t* lut, int nsel, void* o, int out_dtype,
                          int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L,
                          int64_t Lk, int64_t Lk_alloc, int H, td_stream_t stream) {
  return td_attn_i8_ex(q_i8, q_s, k_i8, k_s, vt, lut, nsel, o, out_dtype, o_stride_h, o_stride_l, sm_scale, L, Lk,
                       Lk_alloc, H, nullptr, nullptr, nullptr, stream);
}

static int attn_16_impl(const char* who, const void* q, int64_t q_stride_h, int64_t q_stride_l, const float* q_rstd,
                        const float* q_w, const void* k, const void* vt, const int32_t* lut, int nsel, void* o,
                        int dtype, int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk,
                        int64_t Lk_alloc, int H, const void* add_t, int8_t* q_out, float* q_scale, td_stream_t stream,
                        const AttnGather& ga = kFlat, const float* q_pieces = nullptr, int q_npieces = 0, float q_eps = 0.f) {
  int rc = attn_common_checks(who, q, k, vt, q_out ? (void*)q_out : o, nsel, L, Lk, H, lut);
  TD_REQUIRE((q_out == nullptr) == (q_scale == nullptr), TD_ERR_INVALID, "%s: q_out/q_scale mismatch", who);
  if (!q_out) { rc = attn_stride_check(who, o_stride_h, o_stride_l); if (rc) return rc; }
  if (rc) return rc;
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "%s: dtype %d", who, dtype);
  AttnParams p;
  p.q = q; p.q_s = nullptr; p.k = k; p.k_s = nullptr; p.vt = (const uint16_t*)vt; p.lut = lut;
  p.o = (uint16_t*)o; p.o_stride_h = o_stride_h; p.o_stride_l = o_stride_l;
  p.add_t = (const uint16_t*)add_t; p.q_out = q_out; p.q_scale = q_scale; p.q_heads = ga.q_heads > 0 ? ga.q_heads : H; p.q_ld = (int64_t)p.q_heads * 128;
  p.q_stride_h = q_stride_h; p.q_stride_l = q_stride_l; p.q_rstd = q_rstd; p.q_w = q_w; p.v_scale = nullptr;
  p.q_pieces = q_pieces; p.q_npieces = q_npieces; p.q_inv_n = q_npieces > 0 ? 1.0f / (float)((int64_t)q_npieces * 64) : 0.f; p.q_eps = q_eps;
  p.kbp = ga.kbp; p.k_rs = ga.k_rs; p.v_rs = ga.v_rs; p.ks_rs = ga.ks_rs;
  p.scale_log2 = sm_scale * 1.4426950408889634f;
  p.L = L; p.Lk = Lk; p.H = H; p.Qb = (int)td_cdiv(L, 128); p.Kb = (int)td_cdiv(Lk, 64); p.nsel = nsel;
  if (ga.kbp > 0) Lk_alloc = (int64_t)ga.kbp * 64;
  else if (Lk_alloc == 0) Lk_alloc = Lk;
  TD_REQUIRE(ga.kbp > 0 || (Lk_alloc >= Lk && (Lk_alloc == Lk || Lk_alloc % 64 == 0)), TD_ERR_INVALID, "attn: Lk_alloc=%lld", (long long)Lk_alloc);
  p.k_rows_alloc = Lk_alloc; p.kb_alloc = (int)td_cdiv(Lk_alloc, 64);
  p.tau = td_tuning(TD_TUNE_ATTN_TAU) < 0 ? 0.0f : (td_tuning(TD_TUNE_ATTN_TAU) == 0 ? 8.0f : (float)td_tuning(TD_TUNE_ATTN_TAU));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16) return launch_attn<false, TD_BF16, TD_BF16>(p, st);
  return launch_attn<false, TD_F16, TD_F16>(p, st);
}

extern "C" int td_attn_16_ex(const void* q, const void* k, const void* vt, const int32_t* lut, int nsel,
                             void* o, int dtype, int64_t o_stride_h, int64_t o_stride_l, float sm_scale,
                             int64_t L, int64_t Lk, int64_t Lk_alloc, int H, const void* add_t, int8_t* q_out,
                             float* q_scale, td_stream_t stream) {
  return attn_16_impl("td_attn_16", q, 0, 0, nullptr, nullptr, k, vt, lut, nsel, o, dtype, o_stride_h, o_stride_l,
                      sm_scale, L, Lk, Lk_alloc, H, add_t, q_out, q_scale, stream);
}

extern "C" int td_attn_16(const void* q, const void* k, const void* vt, const int32_t* lut, int nsel, void* o,
                          int dtype, int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L,
                          int64_t Lk, int64_t Lk_alloc, int H, td_stream_t stream) {
  return td_attn_16_ex(q, k, vt, lut, nsel, o, dtype, o_stride_h, o_stride_l, sm_scale, L, Lk, Lk_alloc, H, nullptr,
                       nullptr, nullptr, stream);
}

// Q taken straight from a [L, ld_q] linear output (head h = columns [128h, 128h+128)) with its full-width RMSNorm
// applied on load: q = cast(x * rstd[l] * w) — td_qk_norm_rope without RoPE (the cross-attention Q of
// WanT2VCrossAttention.forward, wan2pt1.py:289) minus its 4-byte-per-element round trip.  rstd from td_rms_stats.
extern "C" int td_attn_16_qnorm(const void* q_src, int64_t ld_q, const float* q_rstd, const float* q_w, const void* k,
                                const void* vt, const int32_t* lut, int nsel, void* o, int dtype, int64_t o_stride_h,
                                int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk, int64_t Lk_alloc, int H,
                                const void* add_t, int8_t* q_out, float* q_scale, td_stream_t stream) {
  TD_REQUIRE(q_rstd && q_w, TD_ERR_INVALID, "td_attn_16_qnorm: null rstd / weight");
  TD_REQUIRE(ld_q >= (int64_t)H * 128 && ld_q % 8 == 0, TD_ERR_INVALID, "td_attn_16_qnorm: ld_q=%lld", (long long)ld_q);
  return attn_16_impl("td_attn_16_qnorm", q_src, 256, ld_q * 2, q_rstd, q_w, k, vt, lut, nsel, o, dtype, o_stride_h,
                      o_stride_l, sm_scale, L, Lk, Lk_alloc, H, add_t, q_out, q_scale, stream);
}


// td_attn_16_qnorm with the row statistic taken from the STATS pieces of the GEMM that produced q_src (td_gemm_w8a8_stats:
// float2 [L, n / 64] = (mean, M2) per 64-column piece, n = H * 128 here): == td_row_stats_finalize(mode 1) + td_attn_16_qnorm
// bit for bit, one launch less per cross-attention (round 6).
extern "C" int td_attn_16_qnorm_pieces(const void* q_src, int64_t ld_q, const float* stats_ws, int pieces, float eps,
                                       const float* q_w, const void* k, const void* vt, const int32_t* lut, int nsel, void* o,
                                       int dtype, int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L,
                                       int64_t Lk, int64_t Lk_alloc, int H, const void* add_t, int8_t* q_out, float* q_scale,
                                       td_stream_t stream) {
  TD_REQUIRE(stats_ws && q_w, TD_ERR_INVALID, "td_attn_16_qnorm_pieces: null statistics / weight");
  TD_REQUIRE(pieces > 0 && pieces <= 128 && (int64_t)pieces * 64 == (int64_t)H * 128, TD_ERR_INVALID,
             "td_attn_16_qnorm_pieces: pieces=%d for H=%d heads (need 64 * pieces == 128 * H: the statistic is over the full width)", pieces, H);
  TD_REQUIRE(ld_q >= (int64_t)H * 128 && ld_q % 8 == 0, TD_ERR_INVALID, "td_attn_16_qnorm_pieces: ld_q=%lld", (long long)ld_q);
  return attn_16_impl("td_attn_16_qnorm_pieces", q_src, 256, ld_q * 2, nullptr, q_w, k, vt, lut, nsel, o, dtype, o_stride_h,
                      o_stride_l, sm_scale, L, Lk, Lk_alloc, H, add_t, q_out, q_scale, stream, kFlat, stats_ws, pieces, eps);
}


// ---- sequence-parallel entry points: the K side (K rows, K scales, V^T tiles) is read straight from the output of the
// all-gather, rank-major (see AttnParams::kbp): no re-layout of the gathered state.  k / k_s / vt point at rank 0's part;
// *_rank_stride are the distances to the next rank's (bytes, floats, bytes); kb_per_rank = blocks of 64 keys per rank.
extern "C" int td_attn_i8_sp(const int8_t* q_i8, const float* q_s, const int8_t* k_i8, const float* k_s,
                             const void* vt, const int32_t* lut, int nsel, void* o, int out_dtype,
                             int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk, int H,
                             int kb_per_rank, int64_t k_rank_stride, int64_t ks_rank_stride, int64_t vt_rank_stride,
                             const void* add_t, int8_t* q_out, float* q_scale, int q_heads_total, td_stream_t stream) {
  TD_REQUIRE(kb_per_rank > 0 && k_rank_stride > 0 && ks_rank_stride > 0 && vt_rank_stride > 0, TD_ERR_INVALID,
             "td_attn_i8_sp: gathered layout kb_per_rank=%d", kb_per_rank);
  TD_REQUIRE(k_rank_stride % 16 == 0 && vt_rank_stride % 16 == 0, TD_ERR_UNSUPPORTED, "td_attn_i8_sp: rank strides must be 16-byte multiples");
  TD_REQUIRE(q_heads_total == 0 || q_heads_total >= H, TD_ERR_INVALID, "td_attn_i8_sp: q_heads_total=%d < H=%d", q_heads_total, H);
  const AttnGather ga = {kb_per_rank, k_rank_stride, vt_rank_stride, ks_rank_stride, q_heads_total};
  return attn_i8_impl(q_i8, q_s, k_i8, k_s, vt, nullptr, lut, nsel, o, out_dtype, o_stride_h, o_stride_l, sm_scale, L, Lk,
                      0, H, add_t, q_out, q_scale, stream, ga);
}

extern "C" int td_attn_16_sp(const void* q, const void* k, const void* vt, const int32_t* lut, int nsel, void* o,
                             int dtype, int64_t o_stride_h, int64_t o_stride_l, float sm_scale, int64_t L, int64_t Lk,
                             int H, int kb_per_rank, int64_t k_rank_stride, int64_t vt_rank_stride, const void* add_t,
                             int8_t* q_out, float* q_scale, int q_heads_total, td_stream_t stream) {
  TD_REQUIRE(kb_per_rank > 0 && k_rank_stride > 0 && vt_rank_stride > 0, TD_ERR_INVALID,
             "td_attn_16_sp: gathered layout kb_per_rank=%d", kb_per_rank);
  TD_REQUIRE(k_rank_stride % 16 == 0 && vt_rank_stride % 16 == 0, TD_ERR_UNSUPPORTED, "td_attn_16_sp: rank strides must be 16-byte multiples");
  TD_REQUIRE(q_heads_total == 0 || q_heads_total >= H, TD_ERR_INVALID, "td_attn_16_sp: q_heads_total=%d < H=%d", q_heads_total, H);
  const AttnGather ga = {kb_per_rank, k_rank_stride, vt_rank_stride, 0, q_heads_total};
  return attn_16_impl("td_attn_16_sp", q, 0, 0, nullptr, nullptr, k, vt, lut, nsel, o, dtype, o_stride_h, o_stride_l,
                      sm_scale, L, Lk, 0, H, add_t, q_out, q_scale, stream, ga);
}
