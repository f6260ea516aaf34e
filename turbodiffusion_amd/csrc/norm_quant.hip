// a6 + a7 -> a16 fused: LayerNorm (+ affine, + AdaLN modulate) whose output is block-quantised for the
// Int8Linear that consumes it (norm1 -> q|k|v, norm3 -> cross q, norm2 -> ffn.0 in WanAttentionBlock.forward,
// rcm/networks/wan2pt1.py:404-413 with the replaced modules of ops/core.py:380-412).
//
// Bit-identical to td_layernorm (norm.hip) followed by td_quant_i8_block128 (quant.hip): the row statistics use
// norm.hip's arithmetic order, the normalised value is rounded to the 16-bit activation dtype exactly where the
// unfused pair rounds it, the block amax / 128/amax / RNE / saturate are the quantiser's.  What disappears is the
// [m, n] 16-bit intermediate: 5 B/element of HBM traffic (2 + 2 read, 1 written) instead of 7 (2 read, 2 written,
// 2 read, 1 written), and the second read mostly hits the Infinity Cache.
//
// The two operators want different shapes — LayerNorm a whole row (n up to 5120) per reduction, the quantiser a
// 128x128 block per amax — and a workgroup that holds 128 full rows on chip needs >256 VGPRs or most of the LDS
// (measured: 66 us at C1, slower than the unfused pair's 37 + 28).  So the row statistics get their own pass:
//   1. ln_stats_kernel       one wave per row, the row in registers, 64-lane butterflies -> (mean, rstd) per row
//   2. ln_apply_quant_kernel one 256-thread workgroup per 128x128 quant block, quant.hip's mapping: each lane
//                            normalises / modulates its 8 x 8 elements with the rows' statistics, the block amax
//                            is a butterfly + 4-entry LDS exchange, 8 bytes of int8 per lane and row.
#include "td_common.h"

__device__ __forceinline__ void load8f(const float* p, float* f) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

template <int NV, int DT>
__global__ __launch_bounds__(256) void ln_stats_kernel(const uint16_t* __restrict__ x, float2* __restrict__ stats,
                                                       float eps, float pad_cols, int64_t m, int n) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  float f[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * 64 + lane) * 8;
    if (col < n) {
      unpack8<DT>(*reinterpret_cast<const uint4*>(x + row * n + col), f[v]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[v][j] = 0.f;
    }
  }
  // same order of operations as norm_rows_kernel<MODE 1> (norm.hip)
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += f[v][j];
  const float mean = wave_sum(sum) / (float)n;
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * 64 + lane) * 8;
    if (col < n) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f[v][j] = f[v][j] - mean;
        sq += f[v][j] * f[v][j];
      }
    }
  }
  const float var = fmaf(pad_cols, mean * mean, wave_sum(sq)) / (float)n;   // pad_cols: see norm_rows_kernel (norm.hip)
  const float rstd = 1.0f / sqrtf(var + eps);
  if (lane == 0) stats[row] = make_float2(mean, rstd);
}

// Straight-line code: affine / AdaLN are template flags, tail rows and columns are handled by zeroing the packed
// result (they must not reach the amax) and predicating the store, the batch index of the AdaLN vectors is one
// division per workgroup (rows_per_batch >= 128: a block touches at most two samples; the second one's vectors are
// fetched by the lanes that cross the boundary).
template <int DT, bool HAS_W, bool HAS_B, bool HAS_MOD>
__global__ __launch_bounds__(256) void ln_apply_quant_kernel(
    const uint16_t* __restrict__ x, const float2* __restrict__ stats, const float* __restrict__ w,
    const float* __restrict__ b, const float* __restrict__ scale, const float* __restrict__ shift,
    uint32_t rows_per_batch, int8_t* __restrict__ q, float* __restrict__ qs, int64_t m, int64_t n, int nb_n) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int bn = blockIdx.x, bm = blockIdx.y;
  const int c8 = tid & 15;   // which 8-element (16 B) column group of the 128-wide block
  const int r0 = tid >> 4;   // row within a 16-row group
  const int64_t col = (int64_t)bn * 128 + c8 * 8;
  const bool col_ok = col < n;  // n % 8 == 0, so a vector is all-in or all-out
  const int64_t colc = col_ok ? col : 0;
  const int64_t row_base = (int64_t)bm * 128 + r0;

  uint4 raw[8];
  float2 st[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    int64_t row = row_base + it * 16;
    if (row > m - 1) row = m - 1;   // clamped reads; the results of tail rows / columns are zeroed below
    raw[it] = *reinterpret_cast<const uint4*>(x + row * n + colc);
    st[it] = stats[row];
  }
  float wv[8], bv[8], sv1[8], hv[8];
  if constexpr (HAS_W) load8f(w + colc, wv);
  if constexpr (HAS_B) load8f(b + colc, bv);
  int64_t next_start = 0;
  bool straddle = false;
  if constexpr (HAS_MOD) {
    const uint32_t bi0 = (uint32_t)(bm * 128) / rows_per_batch;   // uniform: one division per workgroup
    next_start = (int64_t)(bi0 + 1) * rows_per_batch;
    int64_t blk_end = (int64_t)(bm + 1) * 128;
    if (blk_end > m) blk_end = m;
    straddle = next_start < blk_end;
    load8f(scale + (int64_t)bi0 * n + colc, sv1);
    load8f(shift + (int64_t)bi0 * n + colc, hv);
#pragma unroll
    for (int j = 0; j < 8; ++j) sv1[j] = 1.0f + sv1[j];
  }
  uint32_t mx = 0u;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int64_t row = row_base + it * 16;
    if constexpr (HAS_MOD) {
      if (straddle) {  // (uniform branch; rows only grow with `it`, so a lane crosses at most once)
        if (row >= next_start && row - 16 < next_start) {
          const int64_t bi1 = next_start / rows_per_batch;
          load8f(scale + bi1 * n + colc, sv1);
          load8f(shift + bi1 * n + colc, hv);
#pragma unroll
          for (int j = 0; j < 8; ++j) sv1[j] = 1.0f + sv1[j];
        }
      }
    }
    // (VALU diet: the kernel is instruction-bound before it is HBM-bound — hardware RNE packs (v_cvt_pk_*), 1 + scale
    //  formed once per column, the amax on packed 15-bit magnitudes)
    const bool valid = col_ok && row < m;
    uint32_t wds[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float o[2];
      unpack2<DT>(wds[p], o[0], o[1]);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * p + e;
        o[e] = (o[e] - st[it].x) * st[it].y;
        if constexpr (HAS_W) o[e] = o[e] * wv[j];
        if constexpr (HAS_B) o[e] = o[e] + bv[j];
      }
      if constexpr (HAS_MOD) {  // (norm(x).float() * (1 + scale) + shift).type_as(x)
        float x0, x1;
        unpack2<DT>(pack2<DT>(o[0], o[1]), x0, x1);  // the norm's own cast back to x.dtype
        o[0] = x0 * sv1[2 * p] + hv[2 * p];
        o[1] = x1 * sv1[2 * p + 1] + hv[2 * p + 1];
      }
      wds[p] = valid ? pack2<DT>(o[0], o[1]) : 0u;   // the 16-bit activation the unfused pair would have stored
      const uint32_t mag = wds[p] & 0x7fff7fffu;      // both 16-bit formats are monotone in their magnitude bits
      asm("v_pk_max_u16 %0, %0, %1" : "+v"(mx) : "v"(mag));
    }
    raw[it] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
  float amax = fmaxf(half_bits_to_f32<DT>(max(mx & 0xffffu, mx >> 16)), 1e-8f);
  amax = wave_max(amax);
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float mult = 128.0f / amax;  // IEEE division (no fast-math in this build)
  if (tid == 0) qs[(int64_t)bm * nb_n + bn] = amax / 128.0f;

#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int64_t row = row_base + it * 16;
    // q = sat_s8(rne(y * mult)): rne via the 1.5*2^23 add (exact for |v| < 2^22, the same integer as rintf of the
    // rounded product); the low byte of the sum IS the two's-complement code; |y * mult| <= 128 (1 + eps), so only
    // +128 needs the clamp
    const uint32_t wds[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
    uint32_t wd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t c[4];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float y0, y1;
        unpack2<DT>(wds[2 * h + p], y0, y1);
        float v0 = y0 * mult, v1 = y1 * mult;
        v0 = v0 + 12582912.0f; v1 = v1 + 12582912.0f;
        v0 = fminf(v0, 12582912.0f + 127.0f); v1 = fminf(v1, 12582912.0f + 127.0f);
        c[2 * p] = __float_as_uint(v0); c[2 * p + 1] = __float_as_uint(v1);
      }
      const uint32_t lo = __builtin_amdgcn_perm(c[1], c[0], 0x0c0c0400u);  // bytes: c0.b0, c1.b0, 0, 0
      const uint32_t hh = __builtin_amdgcn_perm(c[3], c[2], 0x04000c0cu);  // bytes: 0, 0, c2.b0, c3.b0
      wd[h] = lo | hh;
    }
    if (col_ok && row < m) *reinterpret_cast<uint2*>(q + row * n + col) = make_uint2(wd[0], wd[1]);
  }
}

// row statistics from the per-piece (mean, M2) a GEMM's STATS epilogue wrote (gemm_w8a8_fi.hip; every piece = 64 values,
// M2 = sum of squared deviations from the piece's own mean): merged with the pairwise-update formula of Chan et al. —
//   mean = sum_p mean_p / P,   M2 = sum_p M2_p + 64 * sum_p (mean_p - mean)^2
// — so nothing is ever formed as E[x^2] - mean^2: the result keeps fp32 accuracy for rows whose mean is large against
// their spread (the two-pass statistics of the reference, ops/core.py:293-335, have that property; a one-pass
// (sum, sum of squares) does not).  EIGHT lanes per row (lane j takes pieces j, j+8, ... in order, then a fixed 3-step
// butterfly: deterministic) — a wave reads 8 rows x 64 contiguous bytes per step.
// mode 0: LayerNorm -> out float2 [m] = (mean, 1/sqrt(M2/n + eps));
// mode 1: RMSNorm -> out float [m] = 1/sqrt(E[x^2] + eps), E[x^2] = sum_p (M2_p + 64 mean_p^2) / n (a sum of non-negatives).
__global__ __launch_bounds__(256) void row_stats_finalize_kernel(const float2* __restrict__ ws, int pieces, float inv_n,
                                                                float eps, float pad_cols, int mode,
                                                                float* __restrict__ out, int64_t m) {
  const int sub = threadIdx.x & 7;
  int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool ok = row < m;
  if (!ok) row = m - 1;
  // a lane's pieces (sub, sub + 8, ...: at most RSF_MAX of them — n <= 8192) stay in registers: ONE pass over the
  // workspace, all loads in flight together; the second (between-piece) sum runs on the registers
  constexpr int RSF_MAX = 16;
  float2 v[RSF_MAX];
#pragma unroll
  for (int i = 0; i < RSF_MAX; ++i) {
    const int p = sub + 8 * i;
    v[i] = (p < pieces) ? ws[row * pieces + p] : make_float2(0.f, 0.f);
  }
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < RSF_MAX; ++i) {
    if (sub + 8 * i < pieces) {
      s += v[i].x;
      q += (mode == 0) ? v[i].y : fmaf(64.0f * v[i].x, v[i].x, v[i].y);
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (mode == 0) {
    const float mean = s / (float)pieces;
    float b = 0.f;                       // between-piece part: 64 * sum (mean_p - mean)^2
#pragma unroll
    for (int i = 0; i < RSF_MAX; ++i) {
      if (sub + 8 * i < pieces) {
        const float d = v[i].x - mean;
        b = fmaf(d, d, b);
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) b += __shfl_xor(b, o, 64);
    if (!ok || sub != 0) return;
    const float var = fmaf(pad_cols, mean * mean, fmaf(64.0f, b, q)) * inv_n;   // pad_cols: see norm_rows_kernel (norm.hip)
    reinterpret_cast<float2*>(out)[row] = make_float2(mean, 1.0f / sqrtf(var + eps));
  } else {
    if (!ok || sub != 0) return;
    out[row] = 1.0f / sqrtf(q * inv_n + eps);
  }
}

extern "C" int td_row_stats_finalize(const float* ws, int pieces, int64_t n, float eps, int64_t pad_cols, int mode,
                                     float* out, int64_t m, td_stream_t stream) {
  TD_REQUIRE(pad_cols >= 0 && pad_cols <= 8192, TD_ERR_INVALID, "td_row_stats_finalize: pad_cols=%lld", (long long)pad_cols);
  TD_REQUIRE(ws && out, TD_ERR_INVALID, "td_row_stats_finalize: null pointer");
  TD_REQUIRE(pieces > 0 && pieces <= 128 && n == (int64_t)pieces * 64 && m >= 0 && (mode == 0 || mode == 1), TD_ERR_INVALID,
             "td_row_stats_finalize: pieces=%d n=%lld mode=%d (need n == 64 * pieces: every piece is 64 values)", pieces, (long long)n, mode);
  if (m == 0) return TD_OK;
  row_stats_finalize_kernel<<<(unsigned)td_cdiv(m, 32), 256, 0, (hipStream_t)stream>>>(
      reinterpret_cast<const float2*>(ws), pieces, 1.0f / (float)n, eps, (float)pad_cols, mode, out, m);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

static int ln_quant_impl(const void* x, int dtype, const float* w, const float* b, const float* scale,
                         const float* shift, int64_t rows_per_batch, int8_t* q, float* qs, float* stats_ws,
                         float eps, float pad_cols, int64_t m, int64_t n, td_stream_t stream, bool have_stats) {
  TD_REQUIRE(x && q && qs && stats_ws, TD_ERR_INVALID, "td_layernorm_quant: null pointer");
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_layernorm_quant: dtype %d (need f16|bf16)", dtype);
  TD_REQUIRE(m >= 0 && n > 0, TD_ERR_INVALID, "td_layernorm_quant: bad size");
  TD_REQUIRE(n % 8 == 0 && n <= 8192, TD_ERR_UNSUPPORTED, "td_layernorm_quant: n=%lld (need n %% 8 == 0, n <= 8192)",
             (long long)n);
  TD_REQUIRE((scale == nullptr) == (shift == nullptr), TD_ERR_INVALID, "td_layernorm_quant: scale/shift mismatch");
  TD_REQUIRE(scale == nullptr || (rows_per_batch >= 128 && m < ((int64_t)1 << 31)), TD_ERR_UNSUPPORTED,
             "td_layernorm_quant: rows_per_batch=%lld (need >= 128: a quant row block may touch at most two samples)",
             (long long)rows_per_batch);
  TD_REQUIRE(b == nullptr || w != nullptr, TD_ERR_INVALID, "td_layernorm_quant: bias without weight");
  if (m == 0) return TD_OK;
  const int nv = (int)td_cdiv(n, 512);
  const int nb_n = (int)td_cdiv(n, 128);
  hipStream_t st = (hipStream_t)stream;
  const uint16_t* xp = (const uint16_t*)x;
  float2* sp = reinterpret_cast<float2*>(stats_ws);
  dim3 g1((unsigned)td_cdiv(m, 4));
#define TD_LNS(NV_)                                                                          \
  {                                                                                          \
    if (dtype == TD_BF16) ln_stats_kernel<NV_, TD_BF16><<<g1, 256, 0, st>>>(xp, sp, eps, pad_cols, m, (int)n); \
    else ln_stats_kernel<NV_, TD_F16><<<g1, 256, 0, st>>>(xp, sp, eps, pad_cols, m, (int)n); \
  }
  if (!have_stats) {
  if (nv <= 1) TD_LNS(1) else if (nv <= 2) TD_LNS(2) else if (nv <= 3) TD_LNS(3) else if (nv <= 4) TD_LNS(4)
  else if (nv <= 6) TD_LNS(6) else if (nv <= 8) TD_LNS(8) else if (nv <= 10) TD_LNS(10) else TD_LNS(16)
  }
#undef TD_LNS
  TD_CHECK_LAUNCH();
  dim3 g2(nb_n, (unsigned)td_cdiv(m, 128));
  const uint32_t rpb = scale ? (uint32_t)rows_per_batch : 1u;
#define TD_LNA(DT_, W_, B_, M_)                                                                              \
  ln_apply_quant_kernel<DT_, W_, B_, M_><<<g2, 256, 0, st>>>(xp, sp, w, b, scale, shift, rpb, q, qs, m, n, nb_n)
#define TD_LNA_DT(DT_)                                                                                       \
  {                                                                                                          \
    if (scale) {                                                                                             \
      if (w && b) TD_LNA(DT_, true, true, true); else if (w) TD_LNA(DT_, true, false, true);                 \
      else TD_LNA(DT_, false, false, true);                                                                  \
    } else {                                                                                                 \
      if (w && b) TD_LNA(DT_, true, true, false); else if (w) TD_LNA(DT_, true, false, false);               \
      else TD_LNA(DT_, false, false, false);                                                                 \
    }                                                                                                        \
  }
  if (dtype == TD_BF16) TD_LNA_DT(TD_BF16) else TD_LNA_DT(TD_F16)
#undef TD_LNA_DT
#undef TD_LNA
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_layernorm_quant(const void* x, int dtype, const float* w, const float* b, const float* scale,
                                  const float* shift, int64_t rows_per_batch, int8_t* q, float* qs, float* stats_ws,
                                  float eps, int64_t pad_cols, int64_t m, int64_t n, td_stream_t stream) {
  TD_REQUIRE(pad_cols >= 0 && pad_cols <= 8192, TD_ERR_INVALID, "td_layernorm_quant: pad_cols=%lld", (long long)pad_cols);
  return ln_quant_impl(x, dtype, w, b, scale, shift, rows_per_batch, q, qs, stats_ws, eps, (float)pad_cols, m, n, stream,
                       false);
}

// the apply + quantise pass alone, with the rows' (mean, rstd) supplied (float2 [m], e.g. from td_row_stats_finalize of
// the producing GEMM's STATS epilogue): no statistics pass over x
extern "C" int td_layernorm_quant_stats(const void* x, int dtype, const float* w, const float* b, const float* scale,
                                        const float* shift, int64_t rows_per_batch, int8_t* q, float* qs,
                                        const float* row_stats, int64_t m, int64_t n, td_stream_t stream) {
  return ln_quant_impl(x, dtype, w, b, scale, shift, rows_per_batch, q, qs, const_cast<float*>(row_stats), 0.f, 0.f, m, n,
                       stream, true);
}
