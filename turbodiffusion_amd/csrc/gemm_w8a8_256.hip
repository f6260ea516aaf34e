// a17 (large-M path) — block-scaled W8A8 INT8 GEMM, 256x256 tile, LDS-DMA pipelined (gfx950).
//
// Same semantics as gemm_w8a8.hip (reference: ops/gemm/kernel.hpp:390-427, utils.hpp:116-121):
//   acc_f32[m,n] = sum over 128-deep K blocks (ascending) of
//                  fma(float(int32 sum_k a[m,k]*b[n,k]), a_s[m/128,kb]*b_s[n/128,kb], acc)
// so the two kernels are bit-identical; this one exists because at the INT8 MFMA rate a 128x128
// tile needs 64 B/clk/CU of L2->LDS staging (more than an XCD's L2 delivers) and spends as many
// issue slots on the per-K-block int32->fp32 dequant as on MFMAs.
//
// Design (MI355X-first):
//   * 256(M) x 256(N) tile per 512-thread workgroup (8 wavefronts as 2(M) x 4(N), wave tile
//     128 x 64), K step = 128 B = one scale block: 32 B/clk/CU of staging at MFMA peak.
//   * staging is LDS-DMA (global_load_lds_dwordx4): each wave instruction lands 8 rows x 128 B;
//     the LDS image is lane-linear, so the bank swizzle (16-B slot ^ (row>>1)&7) is applied to
//     the per-lane GLOBAL address and again on the ds_read side.  Two 64 KB stages; the DMA for
//     K block kb+1 is in flight while block kb is multiplied; ONE s_barrier per K block, and
//     no VGPRs are spent on staging.
//   * v_mfma_i32_16x16x64_i8 with the WEIGHT fragment as the A operand (rows = n) and the
//     ACTIVATION fragment as B (cols = m): a lane owns 4 consecutive n of one m row per
//     accumulator -> 8-byte row-contiguous stores.  Each 16x16 sub-tile's K block is a chain of
//     two MFMAs starting from the inline constant 0 (no int32 accumulator file, no zeroing);
//     its 4 int32 results are converted and FMA'd into the fp32 accumulators while the matrix
//     pipe works on the next sub-tiles, so only 128 fp32 accumulators + 16 temporaries are live.
//   * 1-D grid, XCD-aware m-grouped tile order (an activation panel and the weight panels it
//     meets stay in one XCD's L2).
#include "td_common.h"

#define H_BM 256
#define H_BN 256
#define H_TILE (256 * 128)        // one operand tile per K block, bytes
#define H_STAGE (2 * H_TILE)      // activations + weights
#define H_LDS (2 * H_STAGE)       // two stages = 128 KB

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ uint32_t h_swz(uint32_t row, uint32_t chunk) {
  return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4);
}


// ABL: timing-only ablations for profiling (results are WRONG for ABL != 0; never dispatched unless
// the TD_TUNE_GEMM_ABLATE knob asks for it): 1 = no dequant VALU, 2 = no MFMA, 3 = no LDS-DMA in the loop
template <int ODT, int EPI, bool HAS_BIAS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_w8a8_256_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ AS, const int8_t* __restrict__ B,
    const float* __restrict__ BS, const uint16_t* __restrict__ bias, uint16_t* __restrict__ D,
    int64_t M, int64_t N, int64_t K, int64_t ldd, int tiles_m, int tiles_n, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;

  // ---- tile assignment: XCD remap, then m-grouped raster ----
  const uint32_t vid = xcd_remap(blockIdx.x, gridDim.x);
  const int per_group = group_m * tiles_n;
  const int gid = vid / per_group;
  const int first_m = gid * group_m;
  const int gsz = min(group_m, tiles_m - first_m);
  const int in_g = vid % per_group;
  const int tm = first_m + in_g % gsz;
  const int tn = in_g / gsz;
  const int64_t m0 = (int64_t)tm * H_BM, n0 = (int64_t)tn * H_BN;
  const int nk = (int)(K / 128);

  // ---- LDS-DMA staging: wave w moves chunks c = w + 8t (8 rows x 128 B each) of both tiles ----
  const int8_t* ga[4];
  const int8_t* gb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = wave + 8 * t;
    const int row = 8 * c + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // source-side swizzle (LDS image is lane-linear)
    int64_t am = m0 + row; if (am > M - 1) am = M - 1;   // tail rows: clamp (never stored)
    int64_t bn = n0 + row; if (bn > N - 1) bn = N - 1;
    ga[t] = A + am * K + chunk * 16;
    gb[t] = B + bn * K + chunk * 16;
  }
#define H_ISSUE(kb_, buf_)                                                                     \
  {                                                                                            \
    char* sb_ = smem + (buf_) * H_STAGE + wave * 1024;                                         \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                            \
      __builtin_amdgcn_global_load_lds((gptr_t)(ga[t] + (int64_t)(kb_) * 128),                \
                                       (lptr_t)(sb_ + t * 8192), 16, 0, 0);                    \
      __builtin_amdgcn_global_load_lds((gptr_t)(gb[t] + (int64_t)(kb_) * 128),                \
                                       (lptr_t)(sb_ + H_TILE + t * 8192), 16, 0, 0);           \
    }                                                                                          \
  }

  // ---- fragment read offsets (within a stage) ----
  uint32_t xoff[2], woff[2];  // per kc; sub-tile i/j adds 16 rows = 2048 B (swizzle term is periodic in 16 rows)
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    xoff[kc] = h_swz(wm * 128 + l16, 4 * kc + lq);
    woff[kc] = H_TILE + h_swz(wn * 64 + l16, 4 * kc + lq);
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

  H_ISSUE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int kb = 0; kb < nk; ++kb) {
    const int cur = kb & 1;
    if (ABL != 3 && kb + 1 < nk) H_ISSUE(kb + 1, cur ^ 1)
    const float sc = as_row[kb] * bs_row[kb];  // (sa*sb) formed first, kernel.hpp:418
    const char* st = smem + cur * H_STAGE;

    v4i wf[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
        wf[j][kc] = *reinterpret_cast<const v4i*>(st + woff[kc] + j * 2048);
    v4i xf[2][2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) xf[0][kc] = *reinterpret_cast<const v4i*>(st + xoff[kc]);

#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + 1 < 8) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
          xf[(i + 1) & 1][kc] = *reinterpret_cast<const v4i*>(st + xoff[kc] + (i + 1) * 2048);
      }
      v4i t[4];
      const v4i zero = {0, 0, 0, 0};
      if constexpr (ABL == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = wf[j][0] ^ xf[i & 1][0] ^ wf[j][1] ^ xf[i & 1][1];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          t[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j][0], xf[i & 1][0], zero, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          t[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j][1], xf[i & 1][1], t[j], 0, 0, 0);
      }
      // dequant this K block into the fp32 accumulators (one FMA per element, utils.hpp:116-121)
      if constexpr (ABL == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(t[j]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) accf[i][j][r] = fmaf((float)t[j][r], sc, accf[i][j][r]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage kb+1 has landed (this wave's share)
    __builtin_amdgcn_s_barrier();                     // ... everyone's; and stage kb is free again
  }

  // ---- epilogue: lane owns m = ..+l16; accumulator (i,j) holds n = ..+16j+4lq+{0..3} ----
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t m = m0 + wm * 128 + i * 16 + l16;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + 4 * lq;
      if (n >= N) continue;  // N % 8 == 0 and n % 4 == 0: the quad is all-in or all-out
      float bf[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HAS_BIAS) {
        const uint2 bb = *reinterpret_cast<const uint2*>(bias + n);
        unpack2<ODT>(bb.x, bf[0], bf[1]);
        unpack2<ODT>(bb.y, bf[2], bf[3]);
      }
      uint32_t ob[2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
        ob[e] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(accf[i][j][2 * e], accf[i][j][2 * e + 1], bf[2 * e], bf[2 * e + 1]);
      *reinterpret_cast<uint2*>(D + m * ldd + n) = make_uint2(ob[0], ob[1]);
    }
  }
}

template <int ODT, int EPI, bool HAS_BIAS, int ABL = 0>
static int launch_gemm256(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                          const void* bias, void* d, int64_t m, int64_t n, int64_t k, int64_t ldd,
                          hipStream_t st) {
  auto kern = gemm_w8a8_256_kernel<ODT, EPI, HAS_BIAS, ABL>;
  static std::atomic<uint64_t> attr_mask{0};
  td_ensure_dyn_lds(reinterpret_cast<const void*>(kern), H_LDS, attr_mask);
  const int tiles_m = (int)td_cdiv(m, H_BM), tiles_n = (int)td_cdiv(n, H_BN);
  const int group_m = 4;
  const unsigned nwg = (unsigned)tiles_m * (unsigned)tiles_n;
  kern<<<nwg, 512, H_LDS, st>>>(a, a_s, b, b_s, (const uint16_t*)bias, (uint16_t*)d, m, n, k, ldd,
                                tiles_m, tiles_n, group_m);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// called by td_gemm_w8a8 (gemm_w8a8.hip) after argument validation
int td_gemm_w8a8_256(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                     const void* bias, void* d, int out_dtype, int epilogue, int64_t m, int64_t n,
                     int64_t k, int64_t ldd, hipStream_t st) {
  const int abl = td_tuning(TD_TUNE_GEMM_ABLATE);
  if (abl == 1) return launch_gemm256<TD_BF16, TD_EPI_NONE, true, 1>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
  if (abl == 2) return launch_gemm256<TD_BF16, TD_EPI_NONE, true, 2>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
  if (abl == 3) return launch_gemm256<TD_BF16, TD_EPI_NONE, true, 3>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);
#define TD_GEMM_CASE(ODT)                                                                              \
  if (epilogue == TD_EPI_GELU_TANH) {                                                                  \
    return bias ? launch_gemm256<ODT, TD_EPI_GELU_TANH, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)  \
                : launch_gemm256<ODT, TD_EPI_GELU_TANH, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st); \
  } else {                                                                                             \
    return bias ? launch_gemm256<ODT, TD_EPI_NONE, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)    \
                : launch_gemm256<ODT, TD_EPI_NONE, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st);  \
  }
  if (out_dtype == TD_BF16) { TD_GEMM_CASE(TD_BF16) } else { TD_GEMM_CASE(TD_F16) }
#undef TD_GEMM_CASE
}
