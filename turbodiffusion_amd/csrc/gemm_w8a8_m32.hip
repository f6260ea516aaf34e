// a17 (large-M path, v5) — block-scaled W8A8 INT8 GEMM on v_mfma_i32_32x32x32_i8: 256x256 tile, LDS-DMA staging,
// one s_barrier per K block, homogeneous per-wave stream (the v4 design, gemm_w8a8_fi.hip) on the 32x32 matrix op.
//
// Same semantics (bit-identical) as every other variant (reference: ops/gemm/kernel.hpp:390-427, utils.hpp:116-121):
//   acc_f32[m,n] = sum over 128-deep K blocks (ascending) of
//                  fma(float(int32 sum_k a[m,k]*b[n,k]), a_s[m/128,kb]*b_s[n/128,kb], acc)
//
// Why.  The format costs 2 VALU per output element per K block whatever the MFMA shape, and a gfx950 SIMD issues one
// VALU-class instruction (MFMA included, ~8 issue cycles) per ~4.3 cycles (tools/ubench/mix_rate.hip): the main loop
// is issue-bound.  A 32x32x32 MFMA does the work of two 16x16x64 ones for ONE issue, so a K block of a SIMD's two
// waves costs 512 VALU + 64 MFMA issues (~2700 cycles + LDS reads) instead of 512 + 128 (~3200 + LDS reads).
//
// Per wave: a 128(m) x 64(n) sub-tile = 4 (i) x 2 (jj) blocks of 32x32, fp32 accumulators acc[i][jj] (128 VGPRs).
// A "chain" = the 4 MFMAs (k 0..31, .., 96..127) of one block's K-block sum into t[jj] (16 int32 VGPRs, C = 0 on the
// first).  Chains run in the order (i, jj) = (0,0) (0,1) (1,0) ...; one slot = 1 MFMA + 8 dequant VALU:
//     slot (c, 0):  8 x v_fmac (second half of chain c-2, which lived in t[jj])   then   MFMA k0 -> t[jj]
//     slot (c, 1):  MFMA k1   then   8 x v_cvt_f32_i32 (first half of chain c-1, in t[jj^1])
//     slot (c, 2):  MFMA k2   then   8 x v_cvt_f32_i32 (second half of chain c-1)
//     slot (c, 3):  MFMA k3   then   8 x v_fmac        (first half of chain c-1)
//   so a chain's results are first read 18 instructions after its last MFMA (the 8-pass XDL -> VALU hazard needs 12)
//   and t[jj] is overwritten only after its last reader has been issued: two 16-register buffers are enough.
//   * fragments: the weights of the K block stay in 32 VGPRs (wf[jj][ks]); the activation fragments xf[ks] of block
//     row i are replaced IN PLACE by those of row i+1 right behind the last MFMA that reads them (chain (i,1)).
//   * ONE barrier per K block, after chain 6: by then every LDS read of stage kb has returned (row 3's fragments were
//     read during chain 5) and this wave's pieces of stage kb+1 have landed.  Chain 7 then refills all fragments from
//     stage kb+1, and the LDS-DMA of stage kb+2 into the freed buffer is spread over chains 7, 0..4.
//   * MFMAs and VALU are asm volatile in program order; the compiler allocates registers and places s_waitcnt.
#include "td_common.h"

#define G_BM 256
#define G_BN 256
#define G_TILE (256 * 128)
#define G_STAGE (2 * G_TILE)
#define G_LDS (2 * G_STAGE)  // 128 KB

typedef __attribute__((address_space(3))) void* g_lptr_t;

__device__ __forceinline__ uint32_t g_swz(uint32_t row, uint32_t chunk) {
  return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4);
}

#define G_FENCE()                             \
  {                                           \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("" ::: "memory");            \
  }
#define G_BARRIER()                           \
  {                                           \
    G_FENCE()                                 \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  }

// DBG = 2: phase stamps (prologue / main loop / epilogue) as in gemm_w8a8_fi.hip
// FAST = G > 0: one-VALU dequant, re-centred every G K blocks (gemm_w8a8_fi.hip has the derivation and the bound).
// A 16-register constant C operand does not fit beside 128 accumulators, so the chain starts from the INLINE constant
// 1/(2*pi) = 0x3E22F983: |isum| <= 128*128*128 = 2^21 < 0x22F983 keeps 0x3E22F983 + isum inside the binade
// [0.125, 0.25) (ulp 2^-26), i.e. the int32 result read as fp32 is M' + isum * 2^-26 with M' = 10680707 * 2^-26, and
// `acc = fma(raw, s * 2^26, acc)` adds isum*s + 10680707*s.  The two v_cvt blocks of a chain disappear; their slots
// carry the recentring adds when one is due.  |fast - exact| <= 2^-24 * (G+1) * 10680707 * sum_k s_k = 0.64 (G+1) sum_k s_k.
// SCHED bit 0: the K block's barrier sits in chain 6 (right after its first MFMA) instead of at its end, the weight
//   fragments wf[0][*] of the next stage are reloaded behind chain 6's own MFMAs (wf[1][*] and the activation row behind
//   chain 7's), and the LDS-DMA of stage kb+2 is issued in chains 6, 7, 0, 1, 2: the fragment refill is spread over two
//   chains instead of one burst of 12 reads per wave right after the barrier (tools/gemm_trace.py: chains 6 + 7 took
//   2400-3000 cycles against 350 for each of chains 0-5).
// SCHED bit 1: s_setprio 1 for the younger half of the workgroup (waves 4-7 lose the issue arbitration otherwise and
//   reach the barrier ~800 cycles after their partners).
template <int ODT, int EPI, bool HAS_BIAS, bool QOUT = false, bool RES = false, int DBG = 0, int FAST = 0, int SCHED = 0>
__global__ __launch_bounds__(512, 2) void gemm_w8a8_m32_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ AS, const int8_t* __restrict__ B,
    const float* __restrict__ BS, const uint16_t* __restrict__ bias, uint16_t* __restrict__ D,
    int64_t M, int64_t N, int64_t K, int64_t ldd, int tiles_m, int tiles_n, int group_m,
    float* __restrict__ QS, int64_t ldqs, const float* __restrict__ gate, unsigned long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  unsigned long long c_t0 = 0, c_t1 = 0, c_t2 = 0;
  if constexpr (DBG >= 2) c_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long dbg_t[40];
  int dbg_n = 0;
#define G_STAMP()                                                                             \
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
  const int tm = first_m + in_g % gsz;
  const int tn = in_g / gsz;
  const int64_t m0 = (int64_t)tm * G_BM, n0 = (int64_t)tn * G_BN;
  const int nk = (int)(K / 128);

  // ---- LDS-DMA pieces: wave w moves chunks c = w + 8t (8 rows x 128 B) of both operand tiles.  One VGPR offset per
  //      operand (chunk t = 0); chunk t adds 64 rows through the instruction's SGPR offset.  Rows past the end of the
  //      matrix are out of the buffer's range and read as zero (never stored).
  uint32_t ga0, gb0;
  {
    const int row = 8 * wave + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // source-side swizzle (LDS image is lane-linear); row + 64t keeps it
    ga0 = (uint32_t)((m0 + row) * K + chunk * 16);
    gb0 = (uint32_t)((n0 + row) * K + chunk * 16);
  }
  const uint32_t row64 = (uint32_t)(64 * K);
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(uint32_t)(M * K), 0x00020000);
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(uint32_t)(N * K), 0x00020000);
  // piece p of stage kb_ into buffer (kb_ & 1): p = 0..3 activation chunks, 4..7 weight chunks
#define G_PIECE(kb_, p_)                                                                          \
  {                                                                                               \
    char* sb_ = smem + ((kb_) & 1) * G_STAGE + wave * 1024;                                       \
    if ((p_) < 4)                                                                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (g_lptr_t)(sb_ + ((p_) & 3) * 8192), 16,   \
                                               ga0, (kb_) * 128 + ((p_) & 3) * row64, 0, 0);      \
    else                                                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (g_lptr_t)(sb_ + G_TILE + ((p_) & 3) * 8192), \
                                               16, gb0, (kb_) * 128 + ((p_) & 3) * row64, 0, 0);  \
  }

  // ---- fragment read offsets within a stage.  32x32x32 operand: lane holds row (lane & 31), k bytes 16*(lane >> 5)..+15
  //      of the 32-deep step, i.e. 16-byte chunk 2*ks + hi of the row.  Block row i adds i*4096, block column jj*4096
  //      (the row swizzle depends on row bits 1..3 only, which the block offset does not touch).
  uint32_t xoff[4], woff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    xoff[ks] = g_swz(wm * 128 + l32, 2 * ks + hi);
    woff[ks] = G_TILE + g_swz(wn * 64 + l32, 2 * ks + hi);
  }

  v16f acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // scale rows of this wave's 128x64 sub-tile (clamped for tail tiles)
  int64_t mb = (m0 + wm * 128) >> 7, nb = (n0 + wn * 64) >> 7;
  const int64_t mb_max = td_cdiv(M, 128) - 1, nb_max = td_cdiv(N, 128) - 1;
  if (mb > mb_max) mb = mb_max;
  if (nb > nb_max) nb = nb_max;
  const float* as_row = AS + mb * nk;
  const float* bs_row = BS + nb * nk;

  v4i wf[2][4], xf[4];
  v16i t[2];
  // "chains -1 and -2": int 0 -> 0.f, added with scale 0 to the zero accumulators
#pragma unroll
  for (int r = 0; r < 16; ++r) { t[0][r] = 0; t[1][r] = 0; }
  asm volatile("" : "+v"(t[0]), "+v"(t[1]));

#define G_LOAD_X(st_, i_, ks_) xf[ks_] = *reinterpret_cast<const v4i*>((st_) + xoff[ks_] + (i_) * 4096);
#define G_LOAD_W(st_, jj_, ks_) wf[jj_][ks_] = *reinterpret_cast<const v4i*>((st_) + woff[ks_] + (jj_) * 4096);
#define G_MFMA0(d_, a_, b_) \
  asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(d_) : "v"(a_), "v"(b_));
#define G_MFMA0M(d_, a_, b_) \
  asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0.15915494" : "=&v"(d_) : "v"(a_), "v"(b_));
#define G_ADDC8(acc_, h_, c_)                                                                     \
  _Pragma("unroll") for (int r = 0; r < 8; ++r)                                                   \
    asm volatile("v_add_f32 %0, %1, %0" : "+v"((acc_)[8 * (h_) + r]) : "s"(c_));
#define G_MFMA1(d_, a_, b_) \
  asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d_) : "v"(a_), "v"(b_));
#define G_CVT8(v_, h_)                                                                            \
  _Pragma("unroll") for (int r = 0; r < 8; ++r)                                                   \
    asm volatile("v_cvt_f32_i32 %0, %0" : "+v"((v_)[8 * (h_) + r]));
#define G_FMAC8(acc_, v_, h_, sc_)                                                                \
  _Pragma("unroll") for (int r = 0; r < 8; ++r)                                                   \
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"((acc_)[8 * (h_) + r]) : "s"(sc_), "v"((v_)[8 * (h_) + r]));

  // ---- prologue: stage 0 and stage 1 in flight; wait for stage 0; all fragments of chain (0,0)/(0,1) ----
#pragma unroll
  for (int p = 0; p < 8; ++p) G_PIECE(0, p)
  if (nk > 1) {
#pragma unroll
    for (int p = 0; p < 8; ++p) G_PIECE(1, p)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  G_BARRIER()
  if constexpr (DBG >= 2) c_t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    G_LOAD_W(smem, 0, ks)
    G_LOAD_X(smem, 0, ks)
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) G_LOAD_W(smem, 1, ks)

  // block scales live in SGPRs: sc_old = previous K block (its last two chains are dequantised during this block's
  // first two), sc_new = this block's.  (sa*sb) formed first, kernel.hpp:418.
  float sc_old = 0.f;
  // FAST: the scales are kept multiplied by 2^26 (exact): the fmac multiplies raw = M' + isum * 2^-26 by s * 2^26
  constexpr float G_S26 = FAST > 0 ? 67108864.0f : 1.0f;
  const float sv0 = (as_row[0] * bs_row[0]) * G_S26;
  float sc_new = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sv0)));
  // FAST: sum of M' * s'_k (s' = s * 2^26, M' = 0x3E22F983 = 10680707 * 2^-26) not yet taken out of the accumulators;
  // fp64, fed from the VECTOR copy of each scale so that the SGPR copies only have the fmacs as users
  double c_sum = FAST > 0 ? (double)sv0 * (10680707.0 / 67108864.0) : 0.0;

  if constexpr (SCHED & 2) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
  constexpr int UNR = FAST > 0 ? FAST : 1;   // FAST: the K loop is unrolled by the recentring period
  float c_neg = 0.f;
  for (int kb0 = 0; kb0 < nk; kb0 += UNR) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int kb = kb0 + u;
    if (UNR > 1 && kb >= nk) break;
    const bool recentre = FAST > 0 && u == UNR - 1;
    if constexpr (FAST > 0) {
      if (recentre) {
        const float c_hi = (float)c_sum;
        c_sum -= (double)c_hi;
        c_neg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -c_hi)));
      }
    }
    const char* st = smem + (kb & 1) * G_STAGE;
    const char* stn = smem + ((kb + 1) & 1) * G_STAGE;
    const bool more = kb + 1 < nk;
    const bool dma_tail = (kb >= 1) && more;   // rest of stage kb+1 (stage 1 was issued by the prologue)
    const bool dma_head = kb + 2 < nk;         // first pieces of stage kb+2, after this block's barrier
    float sa_n = 0.f, sb_n = 0.f;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const int i = ch >> 1, jj = ch & 1;
      // chain c-2 (same buffer t[jj]) and chain c-1 (buffer t[jj^1]): block coordinates and scale
      const int i2 = (i + 3) & 3;                         // chain c-2 = (i-1, jj)
      const int i1 = jj ? i : ((i + 3) & 3);              // chain c-1 = (i, 0) if jj else (i-1, 1)
      const float sc2 = (ch < 2) ? sc_old : sc_new;
      const float sc1 = (ch < 1) ? sc_old : sc_new;
      G_STAMP()
      // -- slot 0: second fmac half of chain c-2, then the first MFMA of this chain overwrites t[jj]
      G_FMAC8(acc[i2][jj], t[jj], 1, sc2)
      if constexpr (FAST > 0) { G_MFMA0M(t[jj], wf[jj][0], xf[0]) } else { G_MFMA0(t[jj], wf[jj][0], xf[0]) }
      G_FENCE()
      if constexpr (SCHED & 1) {
        if (ch == 6) {
          // every LDS read of stage kb has returned (block row 3's fragments were read during chain 5), this wave's
          // pieces of stage kb+1 (issued up to chain 2) have landed; after the barrier: everyone's
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          G_BARRIER()
        }
        if (ch == 6 && more) { G_LOAD_W(stn, 0, 0) }
        if (jj == 1) { if (i < 3) { G_LOAD_X(st, i + 1, 0) } else if (more) { G_LOAD_X(stn, 0, 0) } }
        if (ch == 7 && more) { G_LOAD_W(stn, 1, 0) }
        if (ch == 6) { if (dma_head) { G_PIECE(kb + 2, 0) G_PIECE(kb + 2, 4) } }
        else if (ch == 7) { if (dma_head) { G_PIECE(kb + 2, 1) G_PIECE(kb + 2, 5) } }
        else if (ch == 0) { if (dma_tail) { G_PIECE(kb + 1, 2) G_PIECE(kb + 1, 6) } }
        else if (ch == 1) { if (dma_tail) { G_PIECE(kb + 1, 3) } }
        else if (ch == 2) { if (dma_tail) { G_PIECE(kb + 1, 7) } }
      } else {
      if (ch == 7 && more) { G_LOAD_W(stn, 0, 0) }
      if (jj == 1) { if (i < 3) { G_LOAD_X(st, i + 1, 0) } else if (more) { G_LOAD_X(stn, 0, 0) } }
      if (ch == 7 && more) { G_LOAD_W(stn, 1, 0) }
      // -- LDS-DMA issue (VMEM issue slots of this wave only)
      if (ch == 7) { if (dma_head) { G_PIECE(kb + 2, 0) G_PIECE(kb + 2, 4) } }
      else if (ch == 0) { if (dma_tail) { G_PIECE(kb + 1, 1) G_PIECE(kb + 1, 5) } }
      else if (ch == 1) { if (dma_tail) { G_PIECE(kb + 1, 2) } }
      else if (ch == 2) { if (dma_tail) { G_PIECE(kb + 1, 6) } }
      else if (ch == 3) { if (dma_tail) { G_PIECE(kb + 1, 3) } }
      else if (ch == 4) { if (dma_tail) { G_PIECE(kb + 1, 7) } }
      }
      if (ch == 5) { if (more) { sa_n = as_row[kb + 1]; sb_n = bs_row[kb + 1]; } }  // scalar loads
      G_FENCE()
      // -- slot 1
      G_MFMA1(t[jj], wf[jj][1], xf[1])
      if constexpr (FAST == 0) { G_CVT8(t[jj ^ 1], 0) } else if (recentre) { G_ADDC8(acc[i][jj], 0, c_neg) }
      G_FENCE()
      if (ch == ((SCHED & 1) ? 6 : 7) && more) { G_LOAD_W(stn, 0, 1) }
      if (jj == 1) { if (i < 3) { G_LOAD_X(st, i + 1, 1) } else if (more) { G_LOAD_X(stn, 0, 1) } }
      if (ch == 7 && more) { G_LOAD_W(stn, 1, 1) }
      G_FENCE()
      // -- slot 2
      G_MFMA1(t[jj], wf[jj][2], xf[2])
      if constexpr (FAST == 0) { G_CVT8(t[jj ^ 1], 1) } else if (recentre) { G_ADDC8(acc[i][jj], 1, c_neg) }
      G_FENCE()
      if (ch == ((SCHED & 1) ? 6 : 7) && more) { G_LOAD_W(stn, 0, 2) }
      if (jj == 1) { if (i < 3) { G_LOAD_X(st, i + 1, 2) } else if (more) { G_LOAD_X(stn, 0, 2) } }
      if (ch == 7 && more) { G_LOAD_W(stn, 1, 2) }
      G_FENCE()
      // -- slot 3
      G_MFMA1(t[jj], wf[jj][3], xf[3])
      G_FMAC8(acc[i1][jj ^ 1], t[jj ^ 1], 0, sc1)
      G_FENCE()
      if (ch == ((SCHED & 1) ? 6 : 7) && more) { G_LOAD_W(stn, 0, 3) }
      if (jj == 1) { if (i < 3) { G_LOAD_X(st, i + 1, 3) } else if (more) { G_LOAD_X(stn, 0, 3) } }
      if (ch == 7 && more) { G_LOAD_W(stn, 1, 3) }
      if (ch == 6 && !(SCHED & 1)) {
        // every LDS read of stage kb has returned (block row 3's fragments were read during chain 5),
        // this wave's pieces of stage kb+1 have landed; after the barrier: everyone's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        G_BARRIER()
      }
      G_FENCE()
    }
    sc_old = sc_new;
    const float sv = (sa_n * sb_n) * G_S26;
    sc_new = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sv)));
    if constexpr (FAST > 0) c_sum = __builtin_fma((double)sv, 10680707.0 / 67108864.0, c_sum);
  }
  }
  if constexpr (DBG == 1) {
    if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0)
      for (int q = 0; q < 40; ++q) dbg[(wave >> 2) * 64 + q] = q < dbg_n ? dbg_t[q] : 0ull;
  }
  if constexpr (DBG >= 2) c_t2 = __builtin_amdgcn_s_memtime();
  // ---- drain: chain (3,0)'s second fmac half, then all of chain (3,1) (last MFMA just issued: 12 wait states) ----
  G_FMAC8(acc[3][0], t[0], 1, sc_old)
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  if constexpr (FAST == 0) {
    G_CVT8(t[1], 0)
    G_CVT8(t[1], 1)
  }
  G_FMAC8(acc[3][1], t[1], 0, sc_old)
  G_FMAC8(acc[3][1], t[1], 1, sc_old)
  if constexpr (FAST > 0) {
    // whatever M*s_k has not been taken out yet (the last, partial group + the fp64 -> fp32 remainders)
    const float c_hi = (float)c_sum;
    const float c_lo = (float)(c_sum - (double)c_hi);
    const float c_last = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -c_hi)));
    const float c_last2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -c_lo)));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) { G_ADDC8(acc[i][j], h, c_last) G_ADDC8(acc[i][j], h, c_last2) }
  }

  // ---- epilogue ----
  // lane owns row m = ..+32i+l32; accumulator (i, jj)[4q + r] holds n_local = 32jj + 8q + 4hi + r.  With j8 = 4jj + q:
  // n_local = 8*j8 + 4hi + r.  After a permlane32 swap of (j8 = ja, jb) the lower half-wave stores the 8 consecutive n of
  // ja, the upper one those of jb.
#define G_ACC(i_, j8_, r_) acc[i_][(j8_) >> 2][4 * ((j8_) & 3) + (r_)]
  if constexpr (QOUT) {
    // (1) the 16-bit results exactly as the plain epilogue would store them, kept in 64 VGPRs
    uint32_t pk[4][8][2];
    const bool tail = (m0 + G_BM > M) || (n0 + G_BN > N);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool m_ok = (m0 + wm * 128 + i * 32 + l32) < M;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int64_t n = n0 + wn * 64 + j * 8 + 4 * hi;
        const bool ok = m_ok && n < N;
        float bf[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) {
          if (n > N - 4) n = N - 4;
          const uint2 bb = *reinterpret_cast<const uint2*>(bias + n);
          unpack2<ODT>(bb.x, bf[0], bf[1]);
          unpack2<ODT>(bb.y, bf[2], bf[3]);
        }
        pk[i][j][0] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(G_ACC(i, j, 0), G_ACC(i, j, 1), bf[0], bf[1]);
        pk[i][j][1] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(G_ACC(i, j, 2), G_ACC(i, j, 3), bf[2], bf[3]);
        if (tail && !ok) { pk[i][j][0] = 0u; pk[i][j][1] = 0u; }  // rows/cols outside the matrix: zero-filled (load.hpp:24-47)
      }
    }
    // (2) amax of this wave's 128x64 half of the 128x128 quant block (15-bit magnitudes, two per v_pk_max_u16)
    uint32_t mx = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t a = pk[i][j][e] & 0x7fff7fffu;
          asm("v_pk_max_u16 %0, %0, %1" : "+v"(mx) : "v"(a));
        }
    uint32_t m16 = max(mx & 0xffffu, mx >> 16);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m16 = max(m16, (uint32_t)__shfl_xor((int)m16, o, 64));
    // (3) the other half belongs to wave ^ 1: exchange through LDS (every wave is past the last barrier of the main loop)
    uint32_t* red = reinterpret_cast<uint32_t*>(smem);
    if (lane == 0) red[wave] = m16;
    __syncthreads();
    m16 = max(red[wave], red[wave ^ 1]);
    float amax = half_bits_to_f32<ODT>(m16);
    amax = fmaxf(amax, 1e-8f);
    const float mult = 128.0f / amax;  // IEEE division, as quant.hip
    {
      const int64_t mb_q = (m0 + wm * 128) >> 7, nb_q = (n0 + wn * 64) >> 7;
      if (lane == 0 && (wn & 1) == 0 && (m0 + wm * 128) < M && (n0 + wn * 64) < N) QS[mb_q * ldqs + nb_q] = amax / 128.0f;
    }
    // (4) quantise: q = sat_s8(rne(x * mult)) via the 1.5*2^23 add; (5) two permlane32 swaps gather 16 consecutive n per
    //     lane (lower half-wave: n 0..15 of the 32-wide block, upper: 16..31) -> one 16-byte store per row and block.
    int8_t* Dq = reinterpret_cast<int8_t*>(D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + wm * 128 + i * 32 + l32;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        uint32_t qd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float x[4];
          unpack2<ODT>(pk[i][4 * jj + q][0], x[0], x[1]);
          unpack2<ODT>(pk[i][4 * jj + q][1], x[2], x[3]);
          uint32_t w[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = x[r] * mult;
            v = v + 12582912.0f;
            v = fminf(v, 12582912.0f + 127.0f);
            w[r] = __float_as_uint(v);
          }
          const uint32_t lo = __builtin_amdgcn_perm(w[1], w[0], 0x0c0c0400u);  // bytes: w0.b0, w1.b0, 0, 0
          const uint32_t hh = __builtin_amdgcn_perm(w[3], w[2], 0x04000c0cu);  // bytes: 0, 0, w2.b0, w3.b0
          qd[q] = lo | hh;
        }
        // lower lanes: {own q0 (n 0..3), partner's q0 (4..7)}, upper lanes: {partner's q2 (16..19), own q2 (20..23)}
        auto p0 = __builtin_amdgcn_permlane32_swap(qd[0], qd[2], false, false);
        auto p1 = __builtin_amdgcn_permlane32_swap(qd[1], qd[3], false, false);
        const uint4 v = make_uint4(p0[0], p0[1], p1[0], p1[1]);
        const int64_t n = n0 + wn * 64 + jj * 32 + 16 * hi;
        if (m < M && n < N) *reinterpret_cast<uint4*>(Dq + m * ldd + n) = v;
      }
    }
    return;
  }
  // RES: the residual tile is fetched up front (16 independent 16-byte loads per lane, at the addresses this lane will
  // store to) so their latency overlaps the conversion of the accumulators
  uint4 xres[RES ? 4 : 1][4];
  if constexpr (RES) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const int64_t m = m0 + wm * 128 + i * 32 + l32;
        const int64_t n = n0 + wn * 64 + (2 * jp + hi) * 8;
        xres[i][jp] = make_uint4(0, 0, 0, 0);
        if (m < M && n < N) xres[i][jp] = *reinterpret_cast<const uint4*>(D + m * ldd + n);
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 128 + i * 32 + l32;
    uint32_t pk[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int64_t n = n0 + wn * 64 + j * 8 + 4 * hi;
      float bf[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HAS_BIAS) {
        if (n > N - 4) n = N - 4;  // tail: clamp the read, the value is never stored
        const uint2 bb = *reinterpret_cast<const uint2*>(bias + n);
        unpack2<ODT>(bb.x, bf[0], bf[1]);
        unpack2<ODT>(bb.y, bf[2], bf[3]);
      }
      pk[j][0] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(G_ACC(i, j, 0), G_ACC(i, j, 1), bf[0], bf[1]);
      pk[j][1] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(G_ACC(i, j, 2), G_ACC(i, j, 3), bf[2], bf[3]);
    }
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const int ja = 2 * jp, jb = 2 * jp + 1;
      auto s0 = __builtin_amdgcn_permlane32_swap(pk[ja][0], pk[jb][0], false, false);
      auto s1 = __builtin_amdgcn_permlane32_swap(pk[ja][1], pk[jb][1], false, false);
      const uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      const int64_t n = n0 + wn * 64 + (hi ? jb : ja) * 8;
      if constexpr (RES) {
        if (m < M && n < N) {
          uint16_t* xp = D + m * ldd + n;
          float xv[8], yv[8];
          unpack8<ODT>(xres[i][jp], xv);
          unpack8<ODT>(v, yv);
          if (gate != nullptr) {
            const float4 g0 = *reinterpret_cast<const float4*>(gate + n), g1 = *reinterpret_cast<const float4*>(gate + n + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float gd = round_half<ODT>(g[e]);           // gate.type_as(x)
              const float tt = round_half<ODT>(yv[e] * gd);     // y * gate -> x.dtype
              xv[e] = xv[e] + tt;                               // x + t    -> x.dtype (rounded at pack)
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = xv[e] + yv[e];
          }
          *reinterpret_cast<uint4*>(xp) = pack8<ODT>(xv);
        }
      } else {
        if (m < M && n < N) *reinterpret_cast<uint4*>(D + m * ldd + n) = v;
      }
    }
  }
  if constexpr (DBG >= 2) {
    const unsigned long long c_t3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c_t4 = __builtin_amdgcn_s_memtime();
    if ((blockIdx.x & 255) == 0 && wave == 0 && lane == 0 && (blockIdx.x >> 8) < 12) {
      unsigned long long* o = dbg + (blockIdx.x >> 8) * 5;
      o[0] = c_t0; o[1] = c_t1; o[2] = c_t2; o[3] = c_t3; o[4] = c_t4;
    }
  }
}

template <int ODT, int EPI, bool HAS_BIAS, bool QOUT = false, bool RES = false, int DBG = 0, int FAST = 0, int SCHED = 0>
static int launch_gemm_m32(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                           const void* bias, void* d, int64_t m, int64_t n, int64_t k, int64_t ldd,
                           hipStream_t st, float* qs = nullptr, int64_t ldqs = 0, const float* gate = nullptr) {
  auto kern = gemm_w8a8_m32_kernel<ODT, EPI, HAS_BIAS, QOUT, RES, DBG, FAST, SCHED>;
  static std::atomic<uint64_t> attr_mask{0};
  td_ensure_dyn_lds(reinterpret_cast<const void*>(kern), G_LDS, attr_mask);
  const int tiles_m = (int)td_cdiv(m, G_BM), tiles_n = (int)td_cdiv(n, G_BN);
  const int group_m = td_tuning(TD_TUNE_GEMM_GROUP_M) > 0 ? td_tuning(TD_TUNE_GEMM_GROUP_M) : 4;
  const unsigned nwg = (unsigned)tiles_m * (unsigned)tiles_n;
  kern<<<nwg, 512, G_LDS, st>>>(a, a_s, b, b_s, (const uint16_t*)bias, (uint16_t*)d, m, n, k, ldd, tiles_m, tiles_n,
                                group_m, qs, ldqs, gate, DBG ? td_dbg_buffer() : nullptr);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// called by td_gemm_w8a8 (gemm_w8a8.hip) after argument validation; needs ldd % 8 == 0
int td_gemm_w8a8_m32(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                     const void* bias, void* d, int out_dtype, int epilogue, int64_t m, int64_t n,
                     int64_t k, int64_t ldd, hipStream_t st) {
  if (td_tuning(TD_TUNE_GEMM_ABLATE) == 6)  // phase stamps (prologue / main loop / epilogue)
    return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
  if (out_dtype == TD_BF16 && bias && epilogue == TD_EPI_NONE && td_tuning(TD_TUNE_GEMM_SCHED) > 0) {  // schedule experiments
    const int abl = td_tuning(TD_TUNE_GEMM_ABLATE), fg = td_gemm_fast_g();
#define TD_M32_SCHED(S_)                                                                                            \
    if (td_tuning(TD_TUNE_GEMM_SCHED) == S_) {                                                                      \
      if (abl == 9) return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 1, 0, S_>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st); \
      if (abl == 6) return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 2, 0, S_>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st); \
      if (fg == 4) return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 0, 4, S_>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);  \
      return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 0, 0, S_>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);               \
    }
    TD_M32_SCHED(1) TD_M32_SCHED(2) TD_M32_SCHED(3)
#undef TD_M32_SCHED
  }
  if (td_tuning(TD_TUNE_GEMM_ABLATE) == 19)  // the same trace of the one-VALU (FAST = 4) instantiation
    return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 1, 4>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
  if (td_tuning(TD_TUNE_GEMM_ABLATE) == 9)  // s_memtime at every chain start of K blocks 8 and 9 (tools/gemm_trace.py)
    return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 1>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
  if (out_dtype == TD_BF16 && bias && epilogue == TD_EPI_NONE) {
    const bool ph = td_tuning(TD_TUNE_GEMM_ABLATE) == 16;  // phase stamps of the FAST instantiation
    switch (td_gemm_fast_g()) {
      case 2: return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 0, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
      case 4: return ph ? launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 2, 4>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)
                        : launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 0, 4>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
      case 8: return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, false, 0, 8>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
      default: break;
    }
  }
#define TD_GEMM_CASE(ODT)                                                                              \
  if (epilogue == TD_EPI_GELU_TANH) {                                                                  \
    return bias ? launch_gemm_m32<ODT, TD_EPI_GELU_TANH, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)  \
                : launch_gemm_m32<ODT, TD_EPI_GELU_TANH, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st); \
  } else {                                                                                             \
    return bias ? launch_gemm_m32<ODT, TD_EPI_NONE, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)    \
                : launch_gemm_m32<ODT, TD_EPI_NONE, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);  \
  }
  if (out_dtype == TD_BF16) { TD_GEMM_CASE(TD_BF16) } else { TD_GEMM_CASE(TD_F16) }
#undef TD_GEMM_CASE
}

// a15+a16 fused: d_q int8 [m, n] + d_s f32 [ceil(m/128), ceil(n/128)] = quant_block128(cast(gemm(...)+bias [gelu]))
int td_gemm_w8a8_m32_q(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                       int8_t* d_q, float* d_s, int act_dtype, int epilogue, int64_t m, int64_t n, int64_t k,
                       hipStream_t st) {
  const int64_t ldqs = td_cdiv(n, 128);
  if (act_dtype == TD_BF16 && bias && epilogue == TD_EPI_GELU_TANH) {  // the model's instantiation: dequant mode x schedule
    const int fg = td_gemm_fast_g() ? 4 : 0, sch = td_tuning(TD_TUNE_GEMM_SCHED);
#define TD_M32_Q(F_, S_)                                                                                      \
    if (fg == F_ && sch == S_)                                                                                \
      return launch_gemm_m32<TD_BF16, TD_EPI_GELU_TANH, true, true, false, 0, F_, S_>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs);
    TD_M32_Q(4, 0) TD_M32_Q(4, 1) TD_M32_Q(4, 3) TD_M32_Q(0, 1) TD_M32_Q(0, 3)
#undef TD_M32_Q
  }
#define TD_GEMM_CASE(ODT)                                                                                   \
  if (epilogue == TD_EPI_GELU_TANH) {                                                                       \
    return bias ? launch_gemm_m32<ODT, TD_EPI_GELU_TANH, true, true>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs)  \
                : launch_gemm_m32<ODT, TD_EPI_GELU_TANH, false, true>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs); \
  } else {                                                                                                  \
    return bias ? launch_gemm_m32<ODT, TD_EPI_NONE, true, true>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs)       \
                : launch_gemm_m32<ODT, TD_EPI_NONE, false, true>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, d_s, ldqs);     \
  }
  if (act_dtype == TD_BF16) { TD_GEMM_CASE(TD_BF16) } else { TD_GEMM_CASE(TD_F16) }
#undef TD_GEMM_CASE
}

// a15 + a7 fused: x[m, ldx] += cast(cast(gemm + bias) * cast(gate))   (gate f32 [n] or NULL for a plain add)
int td_gemm_w8a8_m32_res(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                         void* x, const float* gate, int dtype, int64_t m, int64_t n, int64_t k, int64_t ldx,
                         hipStream_t st) {
  if (dtype == TD_BF16 && bias) {  // the model's instantiation: dequant mode x schedule
    const int fg = td_gemm_fast_g() ? 4 : 0, sch = td_tuning(TD_TUNE_GEMM_SCHED);
#define TD_M32_R(F_, S_)                                                                                      \
    if (fg == F_ && sch == S_)                                                                                \
      return launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, true, 0, F_, S_>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate);
    TD_M32_R(4, 0) TD_M32_R(4, 1) TD_M32_R(4, 3) TD_M32_R(0, 1) TD_M32_R(0, 3)
#undef TD_M32_R
  }
  if (dtype == TD_BF16)
    return bias ? launch_gemm_m32<TD_BF16, TD_EPI_NONE, true, false, true>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate)
                : launch_gemm_m32<TD_BF16, TD_EPI_NONE, false, false, true>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate);
  return bias ? launch_gemm_m32<TD_F16, TD_EPI_NONE, true, false, true>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate)
              : launch_gemm_m32<TD_F16, TD_EPI_NONE, false, false, true>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, nullptr, 0, gate);
}
