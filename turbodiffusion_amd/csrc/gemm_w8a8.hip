// a17 — block-scaled W8A8 INT8 GEMM on CDNA4 matrix cores (gfx950).
//
// Semantics (reference: turbodiffusion/ops/gemm/kernel.hpp:390-427, utils.hpp:116-121):
//   acc_f32[m,n] = sum over 128-deep K blocks (ascending) of
//                  fma( float(int32 sum_{k in block} a[m,k]*b[n,k]),  a_s[m/128,kb]*b_s[n/128,kb],  acc )
//   d = cast(acc)  [ + bias, rounded again in the output dtype — Int8Linear.forward,
//   ops/core.py:408-412 ]  [ GELU-tanh on the rounded value — the FFN, wan2pt1.py:375 ].
//
// MI355X design (not the reference's 4x2-warp m16n8k32 / cp.async CuTe pipeline):
//   * 128(M) x 128(N) output tile per 256-thread workgroup (4 wavefronts in 2x2, 64x64 each),
//     K step = 128 = one scale block, so ONE scalar a_s*b_s per tile per K step.
//   * v_mfma_i32_32x32x32_i8: the WEIGHT fragment is the A operand (rows = n) and the
//     ACTIVATION fragment the B operand (cols = m), so each lane ends up owning 4 consecutive
//     n for one m-row per accumulator quad -> 8-byte row-contiguous stores, no LDS epilogue.
//   * both operands are K-contiguous: a lane's 16 int8 are one ds_read_b128; LDS tiles are
//     [128 rows][128 B] with the 16-B slot XOR-swizzled by (row>>1)&7 so the 16-lane groups of
//     ds_read_b128 hit 16 distinct slots of the 256-B bank row (conflict-free).
//   * double-buffered LDS, register-staged prefetch (global loads of K block kb+1 are issued
//     before the MFMAs of block kb, written to the other buffer after them; one barrier per
//     K block).  2 workgroups per CU (64 KB LDS each, <=256 VGPR) let one wave's int32->fp32
//     dequant VALU work overlap the other's MFMAs.
//   * 1-D grid with an XCD-aware, m-grouped tile order so an activation panel and the weight
//     panel it meets stay in one XCD's L2.
#include "td_common.h"

#define G_BM 128
#define G_BN 128
#define G_BK 128
#define G_TILE_BYTES (128 * 128)          // one operand tile, int8
#define G_BUF_BYTES (2 * G_TILE_BYTES)    // A + B
#define G_LDS_BYTES (2 * G_BUF_BYTES)     // double buffer = 64 KB

__device__ __forceinline__ uint32_t g_swz(uint32_t row, uint32_t slot) {
  return row * 128u + ((slot ^ ((row >> 1) & 7u)) << 4);
}


// MODE (the fused epilogues of the 256x256 kernel, gemm_w8a8_fi.hip, for problems too small to fill the chip with 256x256 tiles
// — the per-rank shapes of an 8-way sequence split: 96 tiles at M = 4096, N = 1536):
//   G_PLAIN   store the 16-bit result
//   G_RES     gated residual in place: D = D + cast(cast(y) * cast(gate))   (gate == nullptr: plain add) — bit-identical to the
//             256x256 kernel's RES epilogue
//   +G_STATS  (with G_PLAIN or G_RES) per row and 64-column piece (mean, M2) of the stored values -> QS float2 [M, N/64]
//             (shifted sums as in gemm_w8a8_fi.hip; another summation order: equal to rounding, not bit for bit)
//   G_QOUT    block-quantise the result for the next W8A8 GEMM: D = int8 [M, ldd], QS = scales [ceil(M/128), ldqs]; the
//             workgroup's 128x128 tile IS one quantisation block — bit-identical to the 256x256 kernel's QOUT epilogue
#define G_PLAIN 0
#define G_RES 1
#define G_STATS 2
#define G_QOUT 4
template <int ODT, int EPI, bool HAS_BIAS, int MODE = G_PLAIN>
__global__ __launch_bounds__(256, 2) void gemm_w8a8_kernel(
    const int8_t* __restrict__ A, const float* __restrict__ AS, const int8_t* __restrict__ B,
    const float* __restrict__ BS, const uint16_t* __restrict__ bias, uint16_t* __restrict__ D,
    int64_t M, int64_t N, int64_t K, int64_t ldd, int tiles_m, int tiles_n, int group_m,
    const float* __restrict__ gate = nullptr, float* __restrict__ QS = nullptr, int64_t ldqs = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;

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
  const int nk = (int)(K / G_BK);

  // ---- staging addresses: thread -> (row = i*32 + tid/8, slot = tid%8) ----
  const int srow = tid >> 3, sslot = tid & 7;
  const int8_t* ga[4];
  const int8_t* gb[4];
  uint32_t loff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 32 + srow;
    int64_t am = m0 + row; if (am > M - 1) am = M - 1;   // tail rows: clamp (never stored)
    int64_t bn = n0 + row; if (bn > N - 1) bn = N - 1;
    ga[i] = A + am * K + sslot * 16;
    gb[i] = B + bn * K + sslot * 16;
    loff[i] = g_swz(row, sslot);
  }
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GLOAD1(i_, kb_)                                                          \
  ra##i_ = *reinterpret_cast<const uint4*>(ga[i_] + (int64_t)(kb_) * G_BK);      \
  rb##i_ = *reinterpret_cast<const uint4*>(gb[i_] + (int64_t)(kb_) * G_BK);
#define GLOAD(kb_) GLOAD1(0, kb_) GLOAD1(1, kb_) GLOAD1(2, kb_) GLOAD1(3, kb_)
#define LSTORE1(i_, base_)                                                       \
  *reinterpret_cast<uint4*>((base_) + loff[i_]) = ra##i_;                        \
  *reinterpret_cast<uint4*>((base_) + G_TILE_BYTES + loff[i_]) = rb##i_;
#define LSTORE(buf_)                                                             \
  {                                                                              \
    char* base_ = smem + (buf_) * G_BUF_BYTES;                                   \
    LSTORE1(0, base_) LSTORE1(1, base_) LSTORE1(2, base_) LSTORE1(3, base_)      \
  }

  v16i acci[2][2];  // [i: m sub-tile][j: n sub-tile]
  v16f accf[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acci[i][j][r] = 0; accf[i][j][r] = 0.f; }
    }

  const float* as_row = AS + (int64_t)tm * nk;
  const float* bs_row = BS + (int64_t)tn * nk;

  GLOAD(0)
  LSTORE(0)
  __syncthreads();

  for (int kb = 0; kb < nk; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nk) { GLOAD(kb + 1) }
    const float sc = as_row[kb] * bs_row[kb];  // (sa*sb) formed first, kernel.hpp:418
    const char* xa = smem + cur * G_BUF_BYTES;        // activation tile
    const char* wb = xa + G_TILE_BYTES;               // weight tile
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      v4i xf[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        xf[i] = *reinterpret_cast<const v4i*>(xa + g_swz(wm * 64 + i * 32 + li, 2 * kc + hi));
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[j] = *reinterpret_cast<const v4i*>(wb + g_swz(wn * 64 + j * 32 + li, 2 * kc + hi));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], xf[i], acci[i][j], 0, 0, 0);
    }
    // dequant this K block into the fp32 accumulators (one FMA per element, utils.hpp:116-121)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          accf[i][j][r] = fmaf((float)acci[i][j][r], sc, accf[i][j][r]);
          acci[i][j][r] = 0;
        }
    if (kb + 1 < nk) { LSTORE(cur ^ 1) }
    __syncthreads();
  }

  // ---- epilogue: lane owns m = ..+li; accumulator quad g4 holds n = ..+8*g4+4*hi+{0..3} ----
  if constexpr (MODE == G_QOUT) {
    // the 16-bit results exactly as the plain epilogue would store them (rows / columns outside the matrix: zero,
    // ops/common/load.hpp:24-47), the block amax, then q = sat_s8(rne(x * (128 / amax))), scale = amax / 128 (quant.hip)
    uint32_t pk[2][2][4][2];
    float amax = 1e-8f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool m_ok = (m0 + wm * 64 + i * 32 + li) < M;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          int64_t n = n0 + wn * 64 + j * 32 + 8 * g4 + 4 * hi;
          const bool ok = m_ok && n < N;
          float bf[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (HAS_BIAS) {
            if (n > N - 4) n = N - 4;
            const uint2 bb = *reinterpret_cast<const uint2*>(bias + n);
            unpack2<ODT>(bb.x, bf[0], bf[1]);
            unpack2<ODT>(bb.y, bf[2], bf[3]);
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            uint32_t w = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(accf[i][j][4 * g4 + 2 * e], accf[i][j][4 * g4 + 2 * e + 1],
                                                               bf[2 * e], bf[2 * e + 1]);
            if (!ok) w = 0u;
            pk[i][j][g4][e] = w;
            float x0, x1;
            unpack2<ODT>(w, x0, x1);
            amax = fmaxf(amax, fmaxf(fabsf(x0), fabsf(x1)));
          }
        }
    }
    amax = wave_max(amax);
    float* red = reinterpret_cast<float*>(smem);   // (the main loop ended on a barrier: the stages are free)
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float mult = 128.0f / amax;  // IEEE division, as quant.hip
    if (tid == 0) QS[(int64_t)tm * ldqs + tn] = amax / 128.0f;
    int8_t* Dq = reinterpret_cast<int8_t*>(D);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t m = m0 + wm * 64 + i * 32 + li;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int64_t n = n0 + wn * 64 + j * 32 + 8 * g4 + 4 * hi;
          if (n >= N) continue;
          float x[4];
          unpack2<ODT>(pk[i][j][g4][0], x[0], x[1]);
          unpack2<ODT>(pk[i][j][g4][1], x[2], x[3]);
          uint32_t w = 0u;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = rintf(x[r] * mult);  // RNE
            v = fminf(fmaxf(v, -128.0f), 127.0f);
            w |= ((uint32_t)(int)v & 0xffu) << (8 * r);
          }
          *reinterpret_cast<uint32_t*>(Dq + m * ldd + n) = w;
        }
    }
    return;
  }
  constexpr bool RES = (MODE & G_RES) != 0, STATS = (MODE & G_STATS) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + wm * 64 + i * 32 + li;
    const bool m_ok = m < M;
    float st_s = 0.f, st_q = 0.f, st_c = 0.f;   // STATS: this lane's 32 of the row's 64 values of the wave's piece (shifted by st_c)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int64_t n = n0 + wn * 64 + j * 32 + 8 * g4 + 4 * hi;
        const bool ok = m_ok && n < N;   // N % 8 == 0 and n % 4 == 0: the quad is all-in or all-out
        float bf[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) {
          if (n < N) {
            const uint2 bb = *reinterpret_cast<const uint2*>(bias + n);
            unpack2<ODT>(bb.x, bf[0], bf[1]);
            unpack2<ODT>(bb.y, bf[2], bf[3]);
          }
        }
        uint32_t ob[2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
          ob[e] = td_gemm_epilogue2<ODT, EPI, HAS_BIAS>(accf[i][j][4 * g4 + 2 * e], accf[i][j][4 * g4 + 2 * e + 1],
                                                         bf[2 * e], bf[2 * e + 1]);
        if constexpr (RES) {
          if (ok) {
            const uint2 xr = *reinterpret_cast<const uint2*>(D + m * ldd + n);
            float xf[4], yf[4];
            unpack2<ODT>(xr.x, xf[0], xf[1]); unpack2<ODT>(xr.y, xf[2], xf[3]);
            unpack2<ODT>(ob[0], yf[0], yf[1]); unpack2<ODT>(ob[1], yf[2], yf[3]);
            if (gate != nullptr) {
              const float4 gv = *reinterpret_cast<const float4*>(gate + n);
              const float g[4] = {round_half<ODT>(gv.x), round_half<ODT>(gv.y), round_half<ODT>(gv.z), round_half<ODT>(gv.w)};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float t = round_half<ODT>(yf[e] * g[e]);    // y * gate -> x.dtype
                xf[e] = xf[e] + t;                                // x + t    -> x.dtype (rounded at pack)
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) xf[e] = xf[e] + yf[e];
            }
            ob[0] = pack2<ODT>(xf[0], xf[1]);
            ob[1] = pack2<ODT>(xf[2], xf[3]);
          }
        }
        if constexpr (STATS) {
          float sv[4];
          unpack2<ODT>(ob[0], sv[0], sv[1]); unpack2<ODT>(ob[1], sv[2], sv[3]);
          if (j == 0 && g4 == 0) st_c = __shfl(sv[0], li, 64);   // the piece's first column sits in the hi = 0 lane
          if (ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = sv[e] - st_c; st_s += d; st_q = fmaf(d, d, st_q); }
          }
        }
        if (ok) *reinterpret_cast<uint2*>(D + m * ldd + n) = make_uint2(ob[0], ob[1]);
      }
    }
    if constexpr (STATS) {
      st_s += __shfl_xor(st_s, 32, 64); st_q += __shfl_xor(st_q, 32, 64);
      if (hi == 0 && m_ok && n0 + wn * 64 < N) {
        const float ds = st_s * (1.0f / 64.0f);
        reinterpret_cast<float2*>(QS)[m * ldqs + ((n0 + wn * 64) >> 6)] = make_float2(st_c + ds, fmaxf(fmaf(-ds, st_s, st_q), 0.f));
      }
    }
  }
}

template <int ODT, int EPI, bool HAS_BIAS, int MODE = G_PLAIN>
static int launch_gemm(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                       const void* bias, void* d, int64_t m, int64_t n, int64_t k, int64_t ldd,
                       hipStream_t st, const float* gate = nullptr, float* qs = nullptr, int64_t ldqs = 0) {
  auto kern = gemm_w8a8_kernel<ODT, EPI, HAS_BIAS, MODE>;
  static std::atomic<uint64_t> attr_mask{0};
  td_ensure_dyn_lds(reinterpret_cast<const void*>(kern), G_LDS_BYTES, attr_mask);
  const int tiles_m = (int)td_cdiv(m, G_BM), tiles_n = (int)td_cdiv(n, G_BN);
  const int group_m = 4;
  const unsigned nwg = (unsigned)tiles_m * (unsigned)tiles_n;
  kern<<<nwg, 256, G_LDS_BYTES, st>>>(a, a_s, b, b_s, (const uint16_t*)bias, (uint16_t*)d, m, n, k,
                                      ldd, tiles_m, tiles_n, group_m, gate, qs, ldqs);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// Problems with fewer than 64 tiles of 128 x 256 go to the 128x128 kernel (two workgroups per CU).  Until round 5 this was "at
// most 128 tiles of 256 x 256" — the per-rank GEMMs of a wide sequence split (M = 4096, N = 1536: 96 such tiles); those now
// run on the LDS-DMA kernel's 128-row form (gemm_w8a8_fi.hip, NI = 4), which measured 0.72-0.81 of the 128x128 kernel's time on
// them (tools/gemm_small_m.py).  TD_TUNE_GEMM_VARIANT overrides (1 / 4 / 5 / 6 / 7 / 8).
static bool td_gemm_small(int64_t m, int64_t n) {
  const int v = td_tuning(TD_TUNE_GEMM_VARIANT);
  if (v == 1) return true;
  if (v != 0) return false;    // (4 / 5 / 6: a 256-column LDS-DMA kernel is forced)
  return td_cdiv(m, 128) * td_cdiv(n, 256) < 64;
}

// The 256x256 LDS-DMA kernels address the int8 operands through 32-bit buffer offsets (gemm_w8a8_fi.hip: ga / gb, the
// descriptor's num_records): m*k and n*k must stay below 2^32 bytes.  B = 1 configurations are far below (C4 ffn.2:
// 75 600 x 13 824 = 1.05e9); a batch of five of them is not — an ERROR here, never a silent wrap-around.
#define TD_REQUIRE_GEMM_EXTENT(who, m, n, k)                                                                      \
  TD_REQUIRE((m) * (k) < ((int64_t)1 << 32) && (n) * (k) < ((int64_t)1 << 32), TD_ERR_UNSUPPORTED,                \
             "%s: m*k = %lld or n*k = %lld reaches 2^32 bytes (32-bit operand offsets): split the rows", who,     \
             (long long)((m) * (k)), (long long)((n) * (k)))

extern "C" int td_gemm_w8a8(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                            const void* bias, void* d, int out_dtype, int epilogue, int64_t m,
                            int64_t n, int64_t k, int64_t ldd, td_stream_t stream) {
  TD_REQUIRE(a && a_s && b && b_s && d, TD_ERR_INVALID, "td_gemm_w8a8: null pointer");
  TD_REQUIRE(m >= 0 && n >= 0 && k >= 0, TD_ERR_INVALID, "td_gemm_w8a8: negative size");
  TD_REQUIRE(out_dtype == TD_F16 || out_dtype == TD_BF16, TD_ERR_UNSUPPORTED,
             "td_gemm_w8a8: out dtype %d (need f16|bf16)", out_dtype);
  TD_REQUIRE(k % 128 == 0 && k > 0, TD_ERR_UNSUPPORTED,
             "td_gemm_w8a8: k=%lld must be a positive multiple of 128 (one scale block per K step)",
             (long long)k);
  TD_REQUIRE(n % 8 == 0, TD_ERR_UNSUPPORTED, "td_gemm_w8a8: n=%lld must be a multiple of 8",
             (long long)n);
  TD_REQUIRE(ldd >= n && ldd % 4 == 0, TD_ERR_INVALID, "td_gemm_w8a8: bad ldd=%lld", (long long)ldd);
  TD_REQUIRE(epilogue == TD_EPI_NONE || epilogue == TD_EPI_GELU_TANH, TD_ERR_UNSUPPORTED,
             "td_gemm_w8a8: epilogue %d", epilogue);
  TD_REQUIRE_GEMM_EXTENT("td_gemm_w8a8", m, n, k);
  if (m == 0 || n == 0) return TD_OK;
  hipStream_t st = (hipStream_t)stream;
  const int variant = td_tuning(TD_TUNE_GEMM_VARIANT);
  // large problems: the fine-interleaved 256x256 LDS-DMA kernel (gemm_w8a8_fi.hip); every variant is bit-identical
  if (variant == 4 || variant == 6 || variant == 7 || variant == 8 || (variant == 0 && m >= 1024 && n >= 256 && ldd % 8 == 0 && !td_gemm_small(m, n))) {
    TD_REQUIRE(ldd % 8 == 0, TD_ERR_UNSUPPORTED, "td_gemm_w8a8: variants 4 / 6 / 7 / 8 need ldd %% 8 == 0");
    return td_gemm_w8a8_fi(a, a_s, b, b_s, bias, d, out_dtype, epilogue, m, n, k, ldd, st);
  }
  if (variant == 5) {  // 32x32x32-MFMA twin of variant 4 (gemm_w8a8_m32.hip)
    TD_REQUIRE(ldd % 8 == 0, TD_ERR_UNSUPPORTED, "td_gemm_w8a8: variant 5 needs ldd %% 8 == 0");
    return td_gemm_w8a8_m32(a, a_s, b, b_s, bias, d, out_dtype, epilogue, m, n, k, ldd, st);
  }
  TD_REQUIRE(variant != 2 && variant != 3, TD_ERR_UNSUPPORTED,
             "td_gemm_w8a8: kernel variants 2 / 3 (round-1 256x256 and ping-pong kernels) were removed; use 1, 4 or 5");
#define TD_GEMM_CASE(ODT)                                                                          \
  if (epilogue == TD_EPI_GELU_TANH) {                                                              \
    return bias ? launch_gemm<ODT, TD_EPI_GELU_TANH, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st) \
                : launch_gemm<ODT, TD_EPI_GELU_TANH, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st); \
  } else {                                                                                         \
    return bias ? launch_gemm<ODT, TD_EPI_NONE, true>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st)   \
                : launch_gemm<ODT, TD_EPI_NONE, false>(a, a_s, b, b_s, bias, d, m, n, k, ldd, st); \
  }
  if (out_dtype == TD_BF16) { TD_GEMM_CASE(TD_BF16) } else { TD_GEMM_CASE(TD_F16) }
#undef TD_GEMM_CASE
}

extern "C" int td_gemm_w8a8_quant(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                                  const void* bias, int8_t* d_q, float* d_s, int act_dtype, int epilogue,
                                  int64_t m, int64_t n, int64_t k, td_stream_t stream) {
  TD_REQUIRE(a && a_s && b && b_s && d_q && d_s, TD_ERR_INVALID, "td_gemm_w8a8_quant: null pointer");
  TD_REQUIRE(m >= 0 && n >= 0 && k >= 0, TD_ERR_INVALID, "td_gemm_w8a8_quant: negative size");
  TD_REQUIRE(act_dtype == TD_F16 || act_dtype == TD_BF16, TD_ERR_UNSUPPORTED,
             "td_gemm_w8a8_quant: intermediate dtype %d (need f16|bf16)", act_dtype);
  TD_REQUIRE(k % 128 == 0 && k > 0, TD_ERR_UNSUPPORTED,
             "td_gemm_w8a8_quant: k=%lld must be a positive multiple of 128", (long long)k);
  TD_REQUIRE(n % 16 == 0, TD_ERR_UNSUPPORTED, "td_gemm_w8a8_quant: n=%lld must be a multiple of 16", (long long)n);
  TD_REQUIRE(epilogue == TD_EPI_NONE || epilogue == TD_EPI_GELU_TANH, TD_ERR_UNSUPPORTED,
             "td_gemm_w8a8_quant: epilogue %d", epilogue);
  TD_REQUIRE_GEMM_EXTENT("td_gemm_w8a8_quant", m, n, k);
  if (m == 0 || n == 0) return TD_OK;
  if (td_gemm_small(m, n) && n % 128 == 0) {   // the workgroup tile is the quantisation block
    hipStream_t st = (hipStream_t)stream;
    const int64_t ldqs = td_cdiv(n, 128);
#define TD_GQ(ODT_, EPI_)                                                                                                   \
    return bias ? launch_gemm<ODT_, EPI_, true, G_QOUT>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, nullptr, d_s, ldqs)       \
                : launch_gemm<ODT_, EPI_, false, G_QOUT>(a, a_s, b, b_s, bias, d_q, m, n, k, n, st, nullptr, d_s, ldqs);
    if (act_dtype == TD_BF16) { if (epilogue == TD_EPI_GELU_TANH) { TD_GQ(TD_BF16, TD_EPI_GELU_TANH) } else { TD_GQ(TD_BF16, TD_EPI_NONE) } }
    else { if (epilogue == TD_EPI_GELU_TANH) { TD_GQ(TD_F16, TD_EPI_GELU_TANH) } else { TD_GQ(TD_F16, TD_EPI_NONE) } }
#undef TD_GQ
  }
  if (td_tuning(TD_TUNE_GEMM_VARIANT) == 5)
    return td_gemm_w8a8_m32_q(a, a_s, b, b_s, bias, d_q, d_s, act_dtype, epilogue, m, n, k, (hipStream_t)stream);
  return td_gemm_w8a8_fi_q(a, a_s, b, b_s, bias, d_q, d_s, act_dtype, epilogue, m, n, k, (hipStream_t)stream);
}

extern "C" int td_gemm_w8a8_residual(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                                     const void* bias, void* x, const float* gate, int dtype, int64_t m,
                                     int64_t n, int64_t k, int64_t ldx, td_stream_t stream) {
  TD_REQUIRE(a && a_s && b && b_s && x, TD_ERR_INVALID, "td_gemm_w8a8_residual: null pointer");
  TD_REQUIRE(m >= 0 && n >= 0 && k >= 0, TD_ERR_INVALID, "td_gemm_w8a8_residual: negative size");
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_gemm_w8a8_residual: dtype %d (need f16|bf16)", dtype);
  TD_REQUIRE(k % 128 == 0 && k > 0, TD_ERR_UNSUPPORTED,
             "td_gemm_w8a8_residual: k=%lld must be a positive multiple of 128", (long long)k);
  TD_REQUIRE(n % 8 == 0, TD_ERR_UNSUPPORTED, "td_gemm_w8a8_residual: n=%lld must be a multiple of 8", (long long)n);
  TD_REQUIRE(ldx >= n && ldx % 8 == 0, TD_ERR_INVALID, "td_gemm_w8a8_residual: bad ldx=%lld", (long long)ldx);
  TD_REQUIRE_GEMM_EXTENT("td_gemm_w8a8_residual", m, n, k);
  if (m == 0 || n == 0) return TD_OK;
  if (td_gemm_small(m, n)) {
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TD_BF16)
      return bias ? launch_gemm<TD_BF16, TD_EPI_NONE, true, G_RES>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, gate)
                  : launch_gemm<TD_BF16, TD_EPI_NONE, false, G_RES>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, gate);
    return bias ? launch_gemm<TD_F16, TD_EPI_NONE, true, G_RES>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, gate)
                : launch_gemm<TD_F16, TD_EPI_NONE, false, G_RES>(a, a_s, b, b_s, bias, x, m, n, k, ldx, st, gate);
  }
  if (td_tuning(TD_TUNE_GEMM_VARIANT) == 5)
    return td_gemm_w8a8_m32_res(a, a_s, b, b_s, bias, x, gate, dtype, m, n, k, ldx, (hipStream_t)stream);
  return td_gemm_w8a8_fi_res(a, a_s, b, b_s, bias, x, gate, dtype, m, n, k, ldx, (hipStream_t)stream);
}


// a15 (+ a7) -> a5 / a6 statistics: td_gemm_w8a8 (x == NULL form: d, ldd) or td_gemm_w8a8_residual (residual != 0: in
// place on x) whose epilogue also writes, per output row and 64-column piece, (mean, M2) of the stored 16-bit
// values: stats_ws float2 [m, n/64] for td_row_stats_finalize.  bf16, bias required, n % 64 == 0, m >= 1024.
extern "C" int td_gemm_w8a8_stats(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                                  void* d_or_x, const float* gate, int residual, int dtype, int64_t m, int64_t n,
                                  int64_t k, int64_t ld, float* stats_ws, td_stream_t stream) {
  TD_REQUIRE(a && a_s && b && b_s && d_or_x && bias && stats_ws, TD_ERR_INVALID, "td_gemm_w8a8_stats: null pointer");
  TD_REQUIRE(dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_gemm_w8a8_stats: dtype %d (bf16 only)", dtype);
  TD_REQUIRE(k % 128 == 0 && k > 0, TD_ERR_UNSUPPORTED, "td_gemm_w8a8_stats: k=%lld must be a positive multiple of 128", (long long)k);
  TD_REQUIRE(n % 64 == 0 && n > 0 && m > 0, TD_ERR_UNSUPPORTED, "td_gemm_w8a8_stats: n=%lld must be a positive multiple of 64", (long long)n);
  TD_REQUIRE(ld >= n && ld % 8 == 0, TD_ERR_INVALID, "td_gemm_w8a8_stats: bad ld=%lld", (long long)ld);
  TD_REQUIRE_GEMM_EXTENT("td_gemm_w8a8_stats", m, n, k);
  if (td_gemm_small(m, n)) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t pieces = n / 64;
    if (residual)
      return launch_gemm<TD_BF16, TD_EPI_NONE, true, G_RES | G_STATS>(a, a_s, b, b_s, bias, d_or_x, m, n, k, ld, st, gate, stats_ws, pieces);
    return launch_gemm<TD_BF16, TD_EPI_NONE, true, G_STATS>(a, a_s, b, b_s, bias, d_or_x, m, n, k, ld, st, nullptr, stats_ws, pieces);
  }
  return td_gemm_w8a8_fi_stats(a, a_s, b, b_s, bias, d_or_x, gate, residual, m, n, k, ld, stats_ws, (hipStream_t)stream);
}


// a15 of a fused q|k|v projection whose V columns [v_col0, n) leave the kernel as the attention kernels' V^T tiles
// (td_v_transpose's layout and values; vt_dtype f16 = the Sage PV operand, or the output dtype) instead of row-major:
// d's V columns are NOT written.  bias required; n, v_col0 multiples of 256 (whole tiles; heads of 128 columns).
extern "C" int td_gemm_w8a8_vt(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                               void* d, int out_dtype, int64_t m, int64_t n, int64_t k, int64_t ldd, int64_t v_col0,
                               void* vt, int vt_dtype, td_stream_t stream) {
  TD_REQUIRE(a && a_s && b && b_s && d && bias && vt, TD_ERR_INVALID, "td_gemm_w8a8_vt: null pointer");
  TD_REQUIRE(out_dtype == TD_F16 || out_dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_gemm_w8a8_vt: out dtype %d", out_dtype);
  TD_REQUIRE(vt_dtype == out_dtype || vt_dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_gemm_w8a8_vt: vt dtype %d (f16 or the output dtype)", vt_dtype);
  TD_REQUIRE(k % 128 == 0 && k > 0, TD_ERR_UNSUPPORTED, "td_gemm_w8a8_vt: k=%lld must be a positive multiple of 128", (long long)k);
  TD_REQUIRE(m > 0 && n > 0 && n % 256 == 0 && v_col0 % 256 == 0 && v_col0 >= 0 && v_col0 < n, TD_ERR_UNSUPPORTED,
             "td_gemm_w8a8_vt: n=%lld v_col0=%lld must be multiples of 256 with v_col0 < n", (long long)n, (long long)v_col0);
  TD_REQUIRE(ldd >= n && ldd % 8 == 0, TD_ERR_INVALID, "td_gemm_w8a8_vt: bad ldd=%lld", (long long)ldd);
  TD_REQUIRE_GEMM_EXTENT("td_gemm_w8a8_vt", m, n, k);
  return td_gemm_w8a8_fi_vt(a, a_s, b, b_s, bias, d, out_dtype, m, n, k, ldd, v_col0, vt, vt_dtype == TD_F16 && out_dtype != TD_F16,
                            (hipStream_t)stream);
}
