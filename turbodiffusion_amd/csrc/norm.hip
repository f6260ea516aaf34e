// a5 / a6 / a7 / a3 / a8 — row-wise HBM-bound kernels for gfx950:
//   td_rmsnorm, td_layernorm(+AdaLN modulate), td_gated_residual, td_qk_norm_rope.
//
// Reference semantics: ops/core.py:96-136 (RMSNorm), :193-242,:293-335 (LayerNorm),
// rcm/networks/wan2pt1.py:404-413 (modulate / gated residual, incl. where the reference
// rounds to bf16), :156-178 (interleaved RoPE in fp32).
//
// Mapping: one 64-lane wavefront per row, the whole row held in registers (16 B per lane
// per load, fully coalesced), reductions are 64-lane butterflies — no LDS, no second pass
// over HBM.  4 rows per 256-thread workgroup.
#include "td_common.h"

template <int DT> struct RowIO {  // 8 consecutive elements per lane
  __device__ static __forceinline__ void load(const void* p, int64_t off, float* f) {
    if constexpr (DT == TD_F32) {
      const float4 a = *reinterpret_cast<const float4*>((const float*)p + off);
      const float4 b = *reinterpret_cast<const float4*>((const float*)p + off + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
      const uint4 v = *reinterpret_cast<const uint4*>((const uint16_t*)p + off);
      unpack8<DT>(v, f);
    }
  }
  __device__ static __forceinline__ float rnd(float v) {
    if constexpr (DT == TD_F32) return v;
    else return round_half<DT>(v);  // hardware RNE pack (v_cvt_pk_*): same value as the software rounding, 2 VALU not 7
  }
  __device__ static __forceinline__ void store(void* p, int64_t off, const float* f) {
    if constexpr (DT == TD_F32) {
      *reinterpret_cast<float4*>((float*)p + off) = make_float4(f[0], f[1], f[2], f[3]);
      *reinterpret_cast<float4*>((float*)p + off + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else {
      *reinterpret_cast<uint4*>((uint16_t*)p + off) = pack8<DT>(f);
    }
  }
};

__device__ __forceinline__ void load8f(const float* p, float* f) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// MODE 0: RMSNorm (w required). MODE 1: LayerNorm (w,b optional; scale/shift optional).
template <int NV, int IDT, int ODT, int MODE>
__global__ __launch_bounds__(256) void norm_rows_kernel(const void* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        int64_t rows_per_batch, void* __restrict__ y,
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
      RowIO<IDT>::load(x, row * n + col, f[v]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[v][j] = 0.f;
    }
  }
  float mean = 0.f;
  if constexpr (MODE == 1) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += f[v][j];
    mean = wave_sum(sum) / (float)n;
  }
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * 64 + lane) * 8;
    if (col < n) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MODE == 1) f[v][j] = f[v][j] - mean;
        sq += f[v][j] * f[v][j];
      }
    }
  }
  // pad_cols > 0 (LayerNorm only): the reference's Triton kernel sums (x - mean)^2 over next_power_of_2(n) columns with
  // the masked columns loaded as 0 (ops/core.py:213-224, 313-324) — each phantom column adds mean^2 to the sum
  const float var = fmaf(pad_cols, mean * mean, wave_sum(sq)) / (float)n;
  const float rstd = 1.0f / sqrtf(var + eps);
  const int64_t bi = (scale != nullptr) ? row / rows_per_batch : 0;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * 64 + lane) * 8;
    if (col >= n) continue;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f[v][j] * rstd;
    if (w != nullptr) {
      float wv[8];
      load8f(w + col, wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = o[j] * wv[j];
      if constexpr (MODE == 1) {
        if (b != nullptr) {
          float bv[8];
          load8f(b + col, bv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = o[j] + bv[j];
        }
      }
    }
    if constexpr (MODE == 1) {
      if (scale != nullptr) {  // (norm(x).float() * (1 + scale) + shift).type_as(x)
        float sv[8], hv[8];
        load8f(scale + bi * n + col, sv);
        load8f(shift + bi * n + col, hv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xn = RowIO<IDT>::rnd(o[j]);  // the norm's own cast back to x.dtype (.type_as(x))
          const float t = xn * (1.0f + sv[j]);
          o[j] = t + hv[j];
        }
      }
    }
    RowIO<ODT>::store(y, row * n + col, o);
  }
}

template <int IDT, int ODT, int MODE>
static int launch_norm(const void* x, const float* w, const float* b, const float* scale,
                       const float* shift, int64_t rpb, void* y, float eps, float pad_cols, int64_t m, int64_t n,
                       hipStream_t st) {
  const int nv = (int)td_cdiv(n, 512);
  dim3 grid((unsigned)td_cdiv(m, 4));
#define TD_NORM_NV(NV_)                                                                         \
  norm_rows_kernel<NV_, IDT, ODT, MODE><<<grid, 256, 0, st>>>(x, w, b, scale, shift, rpb, y, eps, \
                                                              pad_cols, m, (int)n)
  if (nv <= 1) TD_NORM_NV(1);
  else if (nv <= 2) TD_NORM_NV(2);
  else if (nv <= 3) TD_NORM_NV(3);
  else if (nv <= 4) TD_NORM_NV(4);
  else if (nv <= 6) TD_NORM_NV(6);
  else if (nv <= 8) TD_NORM_NV(8);
  else if (nv <= 10) TD_NORM_NV(10);
  else TD_NORM_NV(16);
#undef TD_NORM_NV
  TD_CHECK_LAUNCH();
  return TD_OK;
}

template <int MODE>
static int dispatch_norm(const char* who, const void* x, int idt, const float* w, const float* b,
                         const float* scale, const float* shift, int64_t rpb, void* y, int odt,
                         float eps, float pad_cols, int64_t m, int64_t n, hipStream_t st) {
  TD_REQUIRE(x && y, TD_ERR_INVALID, "%s: null pointer", who);
  TD_REQUIRE(m >= 0 && n > 0, TD_ERR_INVALID, "%s: bad size", who);
  TD_REQUIRE(n % 8 == 0 && n <= 8192, TD_ERR_UNSUPPORTED, "%s: n=%lld (need n%%8==0, n<=8192)", who,
             (long long)n);
  TD_REQUIRE((scale == nullptr) == (shift == nullptr), TD_ERR_INVALID, "%s: scale/shift mismatch", who);
  TD_REQUIRE(scale == nullptr || rpb > 0, TD_ERR_INVALID, "%s: rows_per_batch", who);
  if (m == 0) return TD_OK;
  if (idt == TD_BF16 && odt == TD_BF16)
    return launch_norm<TD_BF16, TD_BF16, MODE>(x, w, b, scale, shift, rpb, y, eps, pad_cols, m, n, st);
  if (idt == TD_F16 && odt == TD_F16)
    return launch_norm<TD_F16, TD_F16, MODE>(x, w, b, scale, shift, rpb, y, eps, pad_cols, m, n, st);
  if (idt == TD_F32 && odt == TD_F32)
    return launch_norm<TD_F32, TD_F32, MODE>(x, w, b, scale, shift, rpb, y, eps, pad_cols, m, n, st);
  if (idt == TD_F32 && odt == TD_BF16)
    return launch_norm<TD_F32, TD_BF16, MODE>(x, w, b, scale, shift, rpb, y, eps, pad_cols, m, n, st);
  if (idt == TD_BF16 && odt == TD_F32)  // the head: fp32 modulate of the bf16-rounded norm (wan2pt1.py:453)
    return launch_norm<TD_BF16, TD_F32, MODE>(x, w, b, scale, shift, rpb, y, eps, pad_cols, m, n, st);
  if (idt == TD_F16 && odt == TD_F32)
    return launch_norm<TD_F16, TD_F32, MODE>(x, w, b, scale, shift, rpb, y, eps, pad_cols, m, n, st);
  td_set_error("%s: unsupported dtype pair in=%d out=%d", who, idt, odt);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_rmsnorm(const void* x, int in_dtype, const float* w, void* y, int out_dtype,
                          float eps, int64_t m, int64_t n, td_stream_t stream) {
  TD_REQUIRE(w, TD_ERR_INVALID, "td_rmsnorm: null weight");
  return dispatch_norm<0>("td_rmsnorm", x, in_dtype, w, nullptr, nullptr, nullptr, 0, y, out_dtype, eps, 0.f,
                          m, n, (hipStream_t)stream);
}

extern "C" int td_layernorm(const void* x, int in_dtype, const float* w, const float* b,
                            const float* scale, const float* shift, int64_t rows_per_batch, void* y,
                            int out_dtype, float eps, int64_t pad_cols, int64_t m, int64_t n, td_stream_t stream) {
  TD_REQUIRE(pad_cols >= 0 && pad_cols <= 8192, TD_ERR_INVALID, "td_layernorm: pad_cols=%lld", (long long)pad_cols);
  return dispatch_norm<1>("td_layernorm", x, in_dtype, w, b, scale, shift, rows_per_batch, y, out_dtype,
                          eps, (float)pad_cols, m, n, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// gated residual: x = x + y * gate.type_as(x)   (two roundings, like the eager reference)
// ---------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void gated_residual_kernel(uint16_t* __restrict__ x,
                                                             const uint16_t* __restrict__ y,
                                                             const float* __restrict__ gate,
                                                             int64_t rows_per_batch, int64_t m, int n) {
  const int64_t nvec = m * (int64_t)(n / 8);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / (n / 8);
    const int col = (int)(i % (n / 8)) * 8;
    float xf[8], yf[8];
    const uint4 xv = *reinterpret_cast<const uint4*>(x + i * 8);
    const uint4 yv = *reinterpret_cast<const uint4*>(y + i * 8);
    unpack8<DT>(xv, xf);
    unpack8<DT>(yv, yf);
    if (gate != nullptr) {
      float g[8];
      load8f(gate + (row / rows_per_batch) * n + col, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gd = RowIO<DT>::rnd(g[j]);          // gate.type_as(x)
        const float t = RowIO<DT>::rnd(yf[j] * gd);     // y * gate  -> x.dtype
        xf[j] = xf[j] + t;                              // x + t     -> x.dtype (rounded at pack)
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = xf[j] + yf[j];
    }
    *reinterpret_cast<uint4*>(x + i * 8) = pack8<DT>(xf);
  }
}

extern "C" int td_gated_residual(void* x, const void* y, const float* gate, int64_t rows_per_batch,
                                 int dtype, int64_t m, int64_t n, td_stream_t stream) {
  TD_REQUIRE(x && y, TD_ERR_INVALID, "td_gated_residual: null pointer");
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_gated_residual: dtype %d", dtype);
  TD_REQUIRE(n > 0 && n % 8 == 0, TD_ERR_UNSUPPORTED, "td_gated_residual: n=%lld", (long long)n);
  TD_REQUIRE(gate == nullptr || rows_per_batch > 0, TD_ERR_INVALID, "td_gated_residual: rows_per_batch");
  if (m == 0) return TD_OK;
  const int64_t nvec = m * (n / 8);
  const unsigned grid = (unsigned)(td_cdiv(nvec, 256) < 4096 ? td_cdiv(nvec, 256) : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16)
    gated_residual_kernel<TD_BF16><<<grid, 256, 0, st>>>((uint16_t*)x, (const uint16_t*)y, gate,
                                                          rows_per_batch, m, (int)n);
  else
    gated_residual_kernel<TD_F16><<<grid, 256, 0, st>>>((uint16_t*)x, (const uint16_t*)y, gate,
                                                         rows_per_batch, m, (int)n);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// ---------------------------------------------------------------------------------------
// row statistic of RMSNorm alone: rstd[l] = 1/sqrt(mean(x[l,:]^2) + eps), same summation order as
// qk_norm_rope_kernel, for consumers that apply the normalisation themselves (td_attn_16_qnorm)
// ---------------------------------------------------------------------------------------
template <int NV, int DT>
__global__ __launch_bounds__(256) void rms_stats_kernel(const uint16_t* __restrict__ src, int64_t ld_src,
                                                        float* __restrict__ rstd, float eps, int64_t L, int n) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= L) return;
  float sq = 0.f;
  uint4 raw[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * 64 + lane) * 8;
    raw[v] = make_uint4(0, 0, 0, 0);
    if (col < n) raw[v] = *reinterpret_cast<const uint4*>(src + row * ld_src + col);
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    float f[8];
    unpack8<DT>(raw[v], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) sq += f[j] * f[j];
  }
  const float r = 1.0f / sqrtf(wave_sum(sq) / (float)n + eps);
  if (lane == 0) rstd[row] = r;
}

extern "C" int td_rms_stats(const void* src, int64_t ld_src, int dtype, float* rstd, float eps, int64_t L, int64_t n,
                            td_stream_t stream) {
  TD_REQUIRE(src && rstd, TD_ERR_INVALID, "td_rms_stats: null pointer");
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_rms_stats: dtype %d", dtype);
  TD_REQUIRE(n > 0 && n % 8 == 0 && n <= 8192 && ld_src >= n && ld_src % 8 == 0, TD_ERR_UNSUPPORTED,
             "td_rms_stats: n=%lld ld=%lld", (long long)n, (long long)ld_src);
  if (L == 0) return TD_OK;
  const int nv = (int)td_cdiv(n, 512);
  dim3 grid((unsigned)td_cdiv(L, 4));
  hipStream_t st = (hipStream_t)stream;
#define TD_RS_NV(NV_)                                                                                        \
  do {                                                                                                       \
    if (dtype == TD_BF16) rms_stats_kernel<NV_, TD_BF16><<<grid, 256, 0, st>>>((const uint16_t*)src, ld_src, rstd, eps, L, (int)n); \
    else rms_stats_kernel<NV_, TD_F16><<<grid, 256, 0, st>>>((const uint16_t*)src, ld_src, rstd, eps, L, (int)n);                   \
  } while (0)
  if (nv <= 1) TD_RS_NV(1);
  else if (nv <= 2) TD_RS_NV(2);
  else if (nv <= 3) TD_RS_NV(3);
  else if (nv <= 4) TD_RS_NV(4);
  else if (nv <= 6) TD_RS_NV(6);
  else if (nv <= 8) TD_RS_NV(8);
  else if (nv <= 10) TD_RS_NV(10);
  else TD_RS_NV(16);
#undef TD_RS_NV
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// ---------------------------------------------------------------------------------------
// q/k: RMSNorm over the full model dim -> cast -> interleaved RoPE (fp32) -> cast -> [H,L,D]
// ---------------------------------------------------------------------------------------
template <int NV, int DT>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const uint16_t* src,
                                                           int64_t ld_src, const float* w,
                                                           const float* __restrict__ cosv,
                                                           const float* __restrict__ sinv,
                                                           uint16_t* dst, float eps,
                                                           int64_t L, int H, int D,
                                                           const uint16_t* __restrict__ src2 = nullptr,
                                                           const float* __restrict__ w2 = nullptr,
                                                           uint16_t* __restrict__ dst2 = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= L) return;
  if (blockIdx.y == 1) { src = src2; w = w2; dst = dst2; }   // td_qk_norm_rope_pair: q and k of a fused projection in ONE launch
  const int n = H * D;
  float f[NV][8];
  float sq = 0.f;
  // the row's chunks are requested together (columns past n read column 0 and count as zeros): one HBM round trip per row
  // instead of one per chunk
  uint4 raw[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * 64 + lane) * 8;
    raw[v] = *reinterpret_cast<const uint4*>(src + row * ld_src + (col < n ? col : 0));
  }
  // every load (weight, cos / sin) is issued BEFORE the first store: on gfx9 stores count in vmcnt, so a load placed after
  // a store makes its s_waitcnt vmcnt(0) wait for that store's completion — once per 512-column chunk of the row
  float wv[NV][8];
  float4 cs[NV], sn[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[v][j] = 1.0f;
    cs[v] = make_float4(1.f, 1.f, 1.f, 1.f);
    sn[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (w != nullptr) {      // (uniform branches around straight-line load groups: columns past n read column 0, never used)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * 64 + lane) * 8;
      load8f(w + (col < n ? col : 0), wv[v]);
    }
  }
  if (cosv != nullptr) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * 64 + lane) * 8;
      const int d0 = (col < n ? col : 0) % D;
      cs[v] = *reinterpret_cast<const float4*>(cosv + row * (D / 2) + d0 / 2);
      sn[v] = *reinterpret_cast<const float4*>(sinv + row * (D / 2) + d0 / 2);
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * 64 + lane) * 8;
    unpack8<DT>(raw[v], f[v]);
    if (col < n) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sq += f[v][j] * f[v][j];
    }
  }
  float rstd = 1.0f;
  if (w != nullptr) rstd = 1.0f / sqrtf(wave_sum(sq) / (float)n + eps);
  // pin the arrival of every load here, before the first store (the compiler would otherwise wait for chunk v's weights
  // after chunk v-1's store, and that wait would include the store)
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(wv[v][j]));
    asm volatile("" : "+v"(cs[v].x), "+v"(cs[v].y), "+v"(cs[v].z), "+v"(cs[v].w));
    asm volatile("" : "+v"(sn[v].x), "+v"(sn[v].y), "+v"(sn[v].z), "+v"(sn[v].w));
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * 64 + lane) * 8;
    if (col >= n) continue;
    float o[8];
    if (w != nullptr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = RowIO<DT>::rnd((f[v][j] * rstd) * wv[v][j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f[v][j];
    }
    const int head = col / D, d0 = col % D;
    if (cosv != nullptr) {
      const float cc[4] = {cs[v].x, cs[v].y, cs[v].z, cs[v].w}, ss[4] = {sn[v].x, sn[v].y, sn[v].z, sn[v].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = o[2 * j], x1 = o[2 * j + 1];
        const float a0 = x0 * cc[j], a1 = x1 * ss[j];
        const float b0 = x0 * ss[j], b1 = x1 * cc[j];
        o[2 * j] = a0 - a1;
        o[2 * j + 1] = b0 + b1;
      }
    }
    RowIO<DT>::store(dst, ((int64_t)head * L + row) * D + d0, o);
  }
}

static int qk_norm_rope_impl(const void* src, int64_t ld_src, const float* w, const float* cosv,
                            const float* sinv, void* dst, const void* src2, const float* w2, void* dst2, int dtype, float eps,
                            int64_t L, int H, int D, td_stream_t stream) {
  TD_REQUIRE(src && dst, TD_ERR_INVALID, "td_qk_norm_rope: null pointer");
  TD_REQUIRE((src2 == nullptr) == (dst2 == nullptr) && (src2 != nullptr || w2 == nullptr) && (src2 == nullptr || (w == nullptr) == (w2 == nullptr)),
             TD_ERR_INVALID, "td_qk_norm_rope_pair: second source / weight / destination mismatch");
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_qk_norm_rope: dtype %d", dtype);
  TD_REQUIRE((cosv == nullptr) == (sinv == nullptr), TD_ERR_INVALID, "td_qk_norm_rope: cos/sin mismatch");
  TD_REQUIRE(H > 0 && D > 0 && D % 8 == 0, TD_ERR_UNSUPPORTED, "td_qk_norm_rope: H=%d D=%d", H, D);
  const int64_t n = (int64_t)H * D;
  TD_REQUIRE(n <= 8192 && ld_src >= n && ld_src % 8 == 0, TD_ERR_UNSUPPORTED,
             "td_qk_norm_rope: dim=%lld ld=%lld", (long long)n, (long long)ld_src);
  if (L == 0) return TD_OK;
  const int nv = (int)td_cdiv(n, 512);
  dim3 grid((unsigned)td_cdiv(L, 4), src2 ? 2u : 1u);
  hipStream_t st = (hipStream_t)stream;
#define TD_QK_NV(NV_)                                                                               \
  do {                                                                                              \
    if (dtype == TD_BF16)                                                                           \
      qk_norm_rope_kernel<NV_, TD_BF16><<<grid, 256, 0, st>>>((const uint16_t*)src, ld_src, w, cosv, \
                                                               sinv, (uint16_t*)dst, eps, L, H, D,   \
                                                               (const uint16_t*)src2, w2, (uint16_t*)dst2); \
    else                                                                                            \
      qk_norm_rope_kernel<NV_, TD_F16><<<grid, 256, 0, st>>>((const uint16_t*)src, ld_src, w, cosv,  \
                                                              sinv, (uint16_t*)dst, eps, L, H, D,    \
                                                              (const uint16_t*)src2, w2, (uint16_t*)dst2); \
  } while (0)
  if (nv <= 1) TD_QK_NV(1);
  else if (nv <= 2) TD_QK_NV(2);
  else if (nv <= 3) TD_QK_NV(3);
  else if (nv <= 4) TD_QK_NV(4);
  else if (nv <= 6) TD_QK_NV(6);
  else if (nv <= 8) TD_QK_NV(8);
  else if (nv <= 10) TD_QK_NV(10);
  else TD_QK_NV(16);
#undef TD_QK_NV
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_qk_norm_rope(const void* src, int64_t ld_src, const float* w, const float* cosv,
                               const float* sinv, void* dst, int dtype, float eps, int64_t L, int H,
                               int D, td_stream_t stream) {
  return qk_norm_rope_impl(src, ld_src, w, cosv, sinv, dst, nullptr, nullptr, nullptr, dtype, eps, L, H, D, stream);
}

// q AND k of a fused q|k|v projection in one launch (same row stride, same RoPE tables; each with its own RMSNorm weight):
// == td_qk_norm_rope(src_q, .., w_q, .., dst_q) and td_qk_norm_rope(src_k, .., w_k, .., dst_k), bit for bit
extern "C" int td_qk_norm_rope_pair(const void* src_q, const void* src_k, int64_t ld_src, const float* w_q, const float* w_k,
                                    const float* cosv, const float* sinv, void* dst_q, void* dst_k, int dtype, float eps,
                                    int64_t L, int H, int D, td_stream_t stream) {
  TD_REQUIRE(src_k && dst_k, TD_ERR_INVALID, "td_qk_norm_rope_pair: null pointer");
  return qk_norm_rope_impl(src_q, ld_src, w_q, cosv, sinv, dst_q, src_k, w_k, dst_k, dtype, eps, L, H, D, stream);
}
