// a17 (large-M path, v3) — block-scaled W8A8 INT8 GEMM, 256x256 tile, LDS-DMA staging, and a
// two-group PING-PONG schedule for gfx950.
//
// Same semantics (bit-identical) as gemm_w8a8.hip / gemm_w8a8_256.hip
// (reference: ops/gemm/kernel.hpp:390-427, utils.hpp:116-121):
//   acc_f32[m,n] = sum over 128-deep K blocks (ascending) of
//                  fma(float(int32 sum_k a[m,k]*b[n,k]), a_s[m/128,kb]*b_s[n/128,kb], acc)
//
// Why a third kernel: measured on MI355X, the one-barrier-per-K-block 256x256 kernel spends a K
// block in ~5400 cycles against 2048 cycles of matrix-pipe work — its eight waves run in lockstep,
// so the two waves that share a SIMD issue their MFMAs at the same time and their dequant VALU /
// ds_read / LDS-DMA issue at the same time, and neither overlaps the other.
//
// Schedule: a K block is 4 phases; phase p = two 16-row m sub-tiles x 4 n sub-tiles = 16 MFMAs (~256 cycles).
// Every wave alternates a C segment (the 16 MFMAs of a phase at raised priority, plus the issue of up to
// 3 LDS-DMA pieces of the next stage) with an O segment (int32->fp32 dequant of the 32 results, ds_read
// of the next phase's fragments), segments separated by s_barrier.  The waves of
// the second M half (waves 4-7, which share SIMDs with waves 0-3) execute ONE extra barrier up front,
// so they are permanently one segment behind: on every SIMD one wave is in a C segment while its
// partner is in an O segment — the matrix pipe always has a feeder, and the VALU/LDS/DMA issue of one
// wave hides under the MFMAs of the other.
//
//   * int32 -> fp32 without v_cvt: each sub-tile's two-MFMA chain starts from C = 0x4B400000
//     (the bits of 1.5*2^23); |sum over a 128-deep block| <= 128*128*128 < 2^22, so the int32 result
//     reinterpreted as fp32 is exactly 12582912 + sum, and one exact v_pk_add_f32 of -12582912
//     per two elements recovers float(sum).  Then one v_pk_fma_f32 per two elements — 4 VALU per
//     sub-tile instead of 8, all exact, bit-identical to cvt+fma.
//   * LDS-DMA for stage kb+1 is issued in the C segments C0..C2 of K block kb and waited for
//     (vmcnt(0)) in O2(kb); its first reader is O3(kb), >= one barrier later.  Its buffer (stage kb-1's)
//     is free: the later group's last read of stage kb-1 is in its O2(kb-1), >= one barrier earlier.
//   * weight rows inside each 16-row group are assigned to MFMA rows with bits 2,3 swapped, so lanes
//     l and l+32 own adjacent n quads and ONE v_permlane32_swap per dword turns the epilogue into
//     16-byte row-contiguous stores.
#include "td_common.h"

#define P_BM 256
#define P_BN 256
#define P_TILE (256 * 128)        // one operand tile per K block, bytes
#define P_STAGE (2 * P_TILE)      // activations + weights
#define P_LDS (2 * P_STAGE)       // two stages = 128 KB
#define P_MAGIC_I 0x4B400000
#define P_MAGIC_F 12582912.0f

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t p_swz(uint32_t row, uint32_t chunk) {
  return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4);
}


#define P_STAMP()                                                                     \
  if constexpr (DBG) {                                                                \
    if (dbg_on && dbg_n < 64) dbg_t[dbg_n++] = __builtin_amdgcn_s_memtime();          \
  }
#define P_SEG_END()                           \
  {                                           \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("" ::: "memory");            \
    P_STAMP()                                 \
    __builtin_amdgcn_s_barrier();             \
    P_STAMP()                                 \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  }

// DBG: profiling instantiation that records s_memtime at every segment boundary of K block 5 for waves
// 0 and 4 of workgroup 0 into g_td_dbg (read back with td_debug_read)
template <int ODT, int EPI, bool HAS_BIAS, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_w8a8_pp_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ AS, const int8_t* __restrict__ B,
    const float* __restrict__ BS, const uint16_t* __restrict__ bias, uint16_t* __restrict__ D,
    int64_t M, int64_t N, int64_t K, int64_t ldd, int tiles_m, int tiles_n, int group_m,
    unsigned long long* __restrict__ g_td_dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  unsigned long long dbg_t[64];
  int dbg_n = 0;
  bool dbg_on = false;

  // ---- tile assignment: XCD remap, then m-grouped raster ----
  const uint32_t vid = xcd_remap(blockIdx.x, gridDim.x);
  const int per_group = group_m * tiles_n;
  const int gid = vid / per_group;
  const int first_m = gid * group_m;
  const int gsz = min(group_m, tiles_m - first_m);
  const int in_g = vid % per_group;
  const int tm = first_m + in_g % gsz;
  const int tn = in_g / gsz;
  const int64_t m0 = (int64_t)tm * P_BM, n0 = (int64_t)tn * P_BN;
  const int nk = (int)(K / 128);

  // ---- LDS-DMA pieces: wave w moves chunks c = w + 8t (8 rows x 128 B) of both operand tiles ----
  // (uniform 64-bit base in SGPRs + 32-bit per-lane offset: the operands are < 4 GB)
  uint32_t ga[4], gb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = wave + 8 * t;
    const int row = 8 * c + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // source-side swizzle (LDS image is lane-linear)
    int64_t am = m0 + row; if (am > M - 1) am = M - 1;   // tail rows: clamp (never stored)
    int64_t bn = n0 + row; if (bn > N - 1) bn = N - 1;
    ga[t] = (uint32_t)(am * K + chunk * 16);
    gb[t] = (uint32_t)(bn * K + chunk * 16);
  }
  // piece p of stage kb_ into buffer (kb_ & 1): p = 0..3 activation chunks, 4..7 weight chunks.
  // buffer_load ... lds: descriptor in SGPRs, one 32-bit VGPR offset per piece, K offset in an SGPR.
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(uint32_t)(M * K), 0x00020000);
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(uint32_t)(N * K), 0x00020000);
#define P_PIECE(kb_, p_)                                                                          \
  {                                                                                               \
    char* sb_ = smem + ((kb_) & 1) * P_STAGE + wave * 1024;                                       \
    if ((p_) < 4)                                                                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(sb_ + ((p_) & 3) * 8192), 16,     \
                                               ga[(p_) & 3], (kb_) * 128, 0, 0);                  \
    else                                                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lptr_t)(sb_ + P_TILE + ((p_) & 3) * 8192), \
                                               16, gb[(p_) & 3], (kb_) * 128, 0, 0);              \
  }

  // ---- fragment read offsets (within a stage); weight rows use the bit-2/3-swapped order ----
  const int pr = (l16 & 3) | ((l16 & 4) << 1) | ((l16 & 8) >> 1);
  uint32_t xoff[2], woff[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    xoff[kc] = p_swz(wm * 128 + l16, 4 * kc + lq);
    woff[kc] = P_TILE + p_swz(wn * 64 + pr, 4 * kc + lq);
  }

  v4f accf[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) accf[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  // scale rows of this wave's 128x64 sub-tile (clamped for tail tiles)
  int64_t mb = (m0 + wm * 128) >> 7, nb = (n0 + wn * 64) >> 7;
  const int64_t mb_max = td_cdiv(M, 128) - 1, nb_max = td_cdiv(N, 128) - 1;
  if (mb > mb_max) mb = mb_max;
  if (nb > nb_max) nb = nb_max;
  const float* as_row = AS + mb * nk;
  const float* bs_row = BS + nb * nk;

  const v4i magic = {P_MAGIC_I, P_MAGIC_I, P_MAGIC_I, P_MAGIC_I};
  v4i wf[4][2] = {}, xf[2][2] = {}, t[2][4] = {};

#define P_LOAD_X(st_, i0_)                                                                        \
  _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                \
  _Pragma("unroll") for (int kc = 0; kc < 2; ++kc)                                                \
    if constexpr (DBG != 3)                                                                       \
      xf[ii][kc] = *reinterpret_cast<const v4i*>((st_) + xoff[kc] + ((i0_) + ii) * 2048);        \
    else asm volatile("" : "+v"(xf[ii][kc]));
#define P_LOAD_W(st_)                                                                             \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
  _Pragma("unroll") for (int kc = 0; kc < 2; ++kc)                                                \
    if constexpr (DBG != 3)                                                                       \
      wf[j][kc] = *reinterpret_cast<const v4i*>((st_) + woff[kc] + j * 2048);                     \
    else asm volatile("" : "+v"(wf[j][kc]));
#define P_MFMA_K0()                                                                               \
  _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
    if constexpr (DBG != 4)                                                                       \
      t[ii][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j][0], xf[ii][0], magic, 0, 0, 0);     \
    else asm volatile("" : "+v"(t[ii][j]) : "v"(wf[j][0]), "v"(xf[ii][0]));
#define P_MFMA_K1()                                                                               \
  _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
    if constexpr (DBG != 4)                                                                       \
      t[ii][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j][1], xf[ii][1], t[ii][j], 0, 0, 0);  \
    else asm volatile("" : "+v"(t[ii][j]) : "v"(wf[j][1]), "v"(xf[ii][1]));
  // exact: float(sum) = as_float(t) - 1.5*2^23 ; acc = fma(float(sum), sc, acc)  (utils.hpp:116-121)
#define P_DEQUANT(i0_, sc_)                                                                       \
  P_STAMP()                                                                                       \
  /* all 32 exact subtractions first (in place), then the 32 FMAs: no back-to-back dependent VALU. */ \
  /* Written as 4-byte VOP2 encodings (SGPR constant, v_fmac) in program order.                    */ \
  _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
  _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
    asm volatile("v_add_f32 %0, %1, %0" : "+v"(t[ii][j][r]) : "s"(-P_MAGIC_F));                   \
  P_STAMP()                                                                                       \
  _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
  _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(accf[(i0_) + ii][j][r]) : "v"(t[ii][j][r]), "v"(sc_));

  // ---- prologue: stage 0 and stage 1 in flight; wait for stage 0; fragments of phase 0 ----
#pragma unroll
  for (int p = 0; p < 8; ++p) P_PIECE(0, p)
  if (nk > 1) {
#pragma unroll
    for (int p = 0; p < 8; ++p) P_PIECE(1, p)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  P_SEG_END()
  P_LOAD_W(smem)
  P_LOAD_X(smem, 0)
  if (wm == 1) P_SEG_END()  // second M half runs one segment behind the first, forever

  float sc = as_row[0] * bs_row[0];  // (sa*sb) formed first, kernel.hpp:418
  for (int kb = 0; kb < nk; ++kb) {
    const char* st = smem + (kb & 1) * P_STAGE;
    const char* stn = smem + ((kb + 1) & 1) * P_STAGE;
    const bool dma = (DBG != 2) && (kb >= 1) && (kb + 1 < nk);  // stage 1 was issued by the prologue
    if constexpr (DBG) dbg_on = (kb == 40) && blockIdx.x == 0 && (wave == 0 || wave == 4);
    // phase p: C segment = 16 MFMAs of m sub-tiles 2p,2p+1 (+ the LDS-DMA issue, which only costs this
    // wave's issue slots while the matrix pipe drains its queue); O segment = next fragments + dequant
    // (no s_setprio around the MFMAs: raising the MFMA wave starves its partner's VALU issue - measured)
#define P_C_SEG(PIECES_)                                                                          \
    P_MFMA_K0()                                                                                   \
    if (dma) { PIECES_ }                                                                          \
    P_MFMA_K1()                                                                                   \
    P_SEG_END()
    // ---- phase 0..2 ----
    P_C_SEG(P_PIECE(kb + 1, 0) P_PIECE(kb + 1, 4) P_PIECE(kb + 1, 1))
    P_LOAD_X(st, 2)
    __builtin_amdgcn_sched_barrier(0);
    P_DEQUANT(0, sc)
    P_SEG_END()
    P_C_SEG(P_PIECE(kb + 1, 5) P_PIECE(kb + 1, 2) P_PIECE(kb + 1, 6))
    P_LOAD_X(st, 4)
    __builtin_amdgcn_sched_barrier(0);
    P_DEQUANT(2, sc)
    P_SEG_END()
    P_C_SEG(P_PIECE(kb + 1, 3) P_PIECE(kb + 1, 7))
    P_LOAD_X(st, 6)
    __builtin_amdgcn_sched_barrier(0);
    P_DEQUANT(4, sc)
    // this wave's pieces of stage kb+1 have landed, and its last ds_reads of stage kb have returned
    // (stage kb's buffer is refilled from the next K block's phase 0 on)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    P_SEG_END()
    // ---- phase 3: fragments of the next K block come from stage kb+1 (complete and visible:
    //      every wave waited, then passed a barrier) ----
    P_C_SEG()
    if (kb + 1 < nk) {
      P_LOAD_W(stn)
      P_LOAD_X(stn, 0)
    }
    float sa_n = 0.f, sb_n = 0.f;
    if (kb + 1 < nk) { sa_n = as_row[kb + 1]; sb_n = bs_row[kb + 1]; }  // scalar loads, hidden by the dequant
    __builtin_amdgcn_sched_barrier(0);
    P_DEQUANT(6, sc)
    sc = sa_n * sb_n;
    P_SEG_END()
  }
  if (wm == 0) P_SEG_END()  // balance the extra barrier of the second half
  if constexpr (DBG) {
    if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0)
      for (int i = 0; i < 64; ++i) g_td_dbg[(wave >> 2) * 64 + i] = i < dbg_n ? dbg_t[i] : 0ull;
  }

  // ---- epilogue ----
  // lane owns m = ..+l16; accumulator (i,j) holds n_local = 16j + r + 8(lq&1) + 4(lq>>1).
  // After the swap lanes with lq<2 store 8 consecutive n of sub-tile ja, lanes with lq>=2 of jb.
  const int hi = lq >> 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t m = m0 + wm * 128 + i * 16 + l16;
    uint32_t pk[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t n = n0 + wn * 64 + j * 16 + 8 * (lq & 1) + 4 * hi;
      float bf[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HAS_BIAS) {
        if (n > N - 4) n = N - 4;  // tail: clamp the read, the value is never stored
        const uint2 bb = *reinterpret_cast<const uint2*>(bias + n);
        unpack2<ODT>(bb.x, bf[0], bf[1]);
        unpack2<ODT>(bb.y, bf[2], bf[3]);
      }
      pk[j][0] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(accf[i][j][0], accf[i][j][1], bf[0], bf[1]);
      pk[j][1] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(accf[i][j][2], accf[i][j][3], bf[2], bf[3]);
    }
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const int ja = 2 * jp, jb = 2 * jp + 1;
      // swap(vdst = tile ja, vsrc = tile jb): upper lanes of ja <-> lower lanes of jb
      auto s0 = __builtin_amdgcn_permlane32_swap(pk[ja][0], pk[jb][0], false, false);
      auto s1 = __builtin_amdgcn_permlane32_swap(pk[ja][1], pk[jb][1], false, false);
      // lower lanes: {own ja quad, partner's ja quad}; upper lanes: {partner's jb quad, own jb quad}
      const uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      const int jt = hi ? jb : ja;
      const int64_t n = n0 + wn * 64 + jt * 16 + 8 * (lq & 1);
      if (m < M && n < N) *reinterpret_cast<uint4*>(D + m * ldd + n) = v;
    }
  }
}

template <int ODT, int EPI, bool HAS_BIAS, int DBG = 0>
static int launch_gemm_pp(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                          const void* bias, void* d, int64_t m, int64_t n, int64_t k, int64_t ldd,
                          hipStream_t st) {
  auto kern = gemm_w8a8_pp_kernel<ODT, EPI, HAS_BIAS, DBG>;
  static std::atomic<uint64_t> attr_mask{0};
  td_ensure_dyn_lds(reinterpret_cast<const void*>(kern), P_LDS, attr_mask);
  const int tiles_m = (int)td_cdiv(m, P_BM), tiles_n = (int)td_cdiv(n, P_BN);
  const int group_m = 4;
  const unsigned nwg = (unsigned)tiles_m * (unsigned)tiles_n;
  kern<<<nwg, 512, P_LDS, st>>>(a, a_s, b, b_s, (const uint16_t*)bias, (uint16_t*)d, m, n, k, ldd,
                                tiles_m, tiles_n, group_m, DBG ? td_dbg_buffer() : nullptr);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// called by td_gemm_w8a8 (gemm_w8a8.hip) after argument validation; needs ldd % 8 == 0
int td_gemm_w8a8_pp(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                    const void* bias, void* d, int out_dtype, int epilogue, int64_t m, int64_t n,
                    int64_t k, int64_t ldd, hipStream_t st) {
  switch (td_tuning(TD_TUNE_GEMM_ABLATE)) {  // profiling instantiations (s_memtime trace; 10-12: wrong results)
    case 9: return launch_gemm_pp<TD_BF16, TD_EPI_NONE, true, 1>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
    case 10: return launch_gemm_pp<TD_BF16, TD_EPI_NONE, true, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
    case 11: return launch_gemm_pp<TD_BF16, TD_EPI_NONE, true, 3>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
    case 12: return launch_gemm_pp<TD_BF16, TD_EPI_NONE, true, 4>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
    default: break;
  }
#define TD_GEMM_CASE(ODT)                                                                              \
  if (epilogue == TD_EPI_GELU_TANH) {                                                                  \
    return bias ? launch_gemm_pp<ODT, TD_EPI_GELU_TANH, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)  \
                : launch_gemm_pp<ODT, TD_EPI_GELU_TANH, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st); \
  } else {                                                                                             \
    return bias ? launch_gemm_pp<ODT, TD_EPI_NONE, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)    \
                : launch_gemm_pp<ODT, TD_EPI_NONE, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);  \
  }
  if (out_dtype == TD_BF16) { TD_GEMM_CASE(TD_BF16) } else { TD_GEMM_CASE(TD_F16) }
#undef TD_GEMM_CASE
}
