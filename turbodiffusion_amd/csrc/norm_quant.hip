// a6 + a7 -> a16 fused: LayerNorm (+ affine, + AdaLN modulate) whose output is block-quantised for the
// Int8Linear that consumes it (norm1 -> q|k|v, norm3 -> cross q, norm2 -> ffn.0 in WanAttentionBlock.forward,
// rcm/networks/wan2pt1.py:404-413 with the replaced modules of ops/core.py:380-412).
//
// Bit-identical to td_layernorm (norm.hip) followed by td_quant_i8_block128 (quant.hip): the normalised value
// is rounded to the 16-bit activation dtype exactly where the unfused pair rounds it, the block amax / 128/amax /
// RNE / saturate are the quantiser's.  What disappears is the [m, n] 16-bit round trip through HBM:
// 3 B/element (2 read + 1 written) instead of 7.
//
// Mapping: one 512-thread workgroup (8 waves) per 128-row quant row block; a wave owns 16 rows, one row at a
// time entirely in registers (16-byte loads, 64-lane butterflies for mean / variance — same arithmetic order as
// norm.hip), the 16-bit results of its 16 rows stay on chip — packed in VGPRs, and for n > 1024 six of the sixteen
// rows in LDS (144 KB per workgroup) so that the kernel stays inside 256 VGPRs — while the
// per-column-block amax is reduced: 16 lanes (one 128-column block per quarter wave per vector) by shuffles, the 8
// waves through LDS.  Then every wave quantises its own rows: 8 bytes per lane per vector, 512 B contiguous
// per row.  n <= 1536 (NV <= 3); larger n uses the unfused pair.
#include "td_common.h"

// RPW = rows per wave (128 / RPW waves per workgroup); RL = rows (of a wave's RPW) whose results wait in LDS
template <int NV, int DT, int RL, int RPW>
__global__ __launch_bounds__(128 / RPW * 64) void layernorm_quant_kernel(
    const uint16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ scale, const float* __restrict__ shift, int64_t rows_per_batch,
    int8_t* __restrict__ q, float* __restrict__ qs, float eps, int64_t m, int n, int nb_n) {
  constexpr int NW = 128 / RPW;
  __shared__ uint32_t red[NW][NV * 4];
  __shared__ uint4 stash[RL > 0 ? NW * RL * NV * 64 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * RPW;
  uint32_t pk[RPW - RL][NV][4];  // rows RL..7 (raw 16-bit data first, the 16-bit result afterwards)
  uint32_t mx[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) mx[v] = 0u;

  // the loads of rows RL..7 are all issued up front ((8-RL) x NV x 16 B per lane in flight); a row's raw 16-bit data
  // lives in the registers that later hold its 16-bit result.  Rows 0..RL-1 go first, through a scratch vector.
#pragma unroll
  for (int r = RL; r < RPW; ++r) {
    const int64_t row = row0 + r;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * 64 + lane) * 8;
      uint4 raw = make_uint4(0, 0, 0, 0);
      if (row < m && col < n) raw = *reinterpret_cast<const uint4*>(x + row * n + col);
      pk[r - RL][v][0] = raw.x; pk[r - RL][v][1] = raw.y; pk[r - RL][v][2] = raw.z; pk[r - RL][v][3] = raw.w;
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int64_t row = row0 + r;
    const bool row_ok = row < m;
    asm volatile("" ::: "memory");  // keep the per-row parameter loads inside their row (register pressure)
    uint32_t cur[NV][4];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (r < RL) {
        const int col = (v * 64 + lane) * 8;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (row_ok && col < n) raw = *reinterpret_cast<const uint4*>(x + row * n + col);
        cur[v][0] = raw.x; cur[v][1] = raw.y; cur[v][2] = raw.z; cur[v][3] = raw.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[v][e] = pk[r < RL ? 0 : r - RL][v][e];
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float f[8];
      unpack8<DT>(make_uint4(cur[v][0], cur[v][1], cur[v][2], cur[v][3]), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += f[j];
    }
    const float mean = wave_sum(sum) / (float)n;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * 64 + lane) * 8;
      if (col < n) {
        float f[8];
        unpack8<DT>(make_uint4(cur[v][0], cur[v][1], cur[v][2], cur[v][3]), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = f[j] - mean;
          sq += d * d;
        }
      }
    }
    const float var = wave_sum(sq) / (float)n;
    const float rstd = 1.0f / sqrtf(var + eps);
    const int64_t bi = (scale != nullptr && row_ok) ? row / rows_per_batch : 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * 64 + lane) * 8;
      if (col < n && row_ok) {
        float o[8];
        unpack8<DT>(make_uint4(cur[v][0], cur[v][1], cur[v][2], cur[v][3]), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (o[j] - mean) * rstd;
        if (w != nullptr) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 wv = *reinterpret_cast<const float4*>(w + col + 4 * h);
            o[4 * h] *= wv.x; o[4 * h + 1] *= wv.y; o[4 * h + 2] *= wv.z; o[4 * h + 3] *= wv.w;
          }
          if (b != nullptr) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float4 bv = *reinterpret_cast<const float4*>(b + col + 4 * h);
              o[4 * h] += bv.x; o[4 * h + 1] += bv.y; o[4 * h + 2] += bv.z; o[4 * h + 3] += bv.w;
            }
          }
        }
        if (scale != nullptr) {  // (norm(x).float() * (1 + scale) + shift).type_as(x)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 sv = *reinterpret_cast<const float4*>(scale + bi * n + col + 4 * h);
            const float4 hv = *reinterpret_cast<const float4*>(shift + bi * n + col + 4 * h);
            const float s4[4] = {sv.x, sv.y, sv.z, sv.w}, h4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float xn = half_bits_to_f32<DT>(f32_to_half_bits<DT>(o[4 * h + j]));  // the norm's own cast to x.dtype
              const float t = xn * (1.0f + s4[j]);
              o[4 * h + j] = t + h4[j];
            }
          }
        }
        const uint4 p = pack8<DT>(o);
        cur[v][0] = p.x; cur[v][1] = p.y; cur[v][2] = p.z; cur[v][3] = p.w;
      } else {
        cur[v][0] = cur[v][1] = cur[v][2] = cur[v][3] = 0u;  // outside the matrix: zero-filled
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t a = cur[v][e] & 0x7fff7fffu;
        asm("v_pk_max_u16 %0, %0, %1" : "+v"(mx[v]) : "v"(a));
      }
      if (r < RL) {
        stash[((wave * RL + (r < RL ? r : 0)) * NV + v) * 64 + lane] = make_uint4(cur[v][0], cur[v][1], cur[v][2], cur[v][3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[r < RL ? 0 : r - RL][v][e] = cur[v][e];
      }
    }
  }
  // amax of column block (4v + lq): 16 lanes of this wave, then the 16 waves
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    uint32_t a = max(mx[v] & 0xffffu, mx[v] >> 16);
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) a = max(a, (uint32_t)__shfl_xor((int)a, o, 64));
    if ((lane & 15) == 0) red[wave][v * 4 + lq] = a;
  }
  __syncthreads();
  float mult[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    uint32_t a = 0u;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) a = max(a, red[wv][v * 4 + lq]);
    const float amax = fmaxf(half_bits_to_f32<DT>(a), 1e-8f);
    mult[v] = 128.0f / amax;  // IEEE division, as quant.hip
    const int cb = v * 4 + lq;
    if (wave == 0 && (lane & 15) == 0 && cb < nb_n) qs[(int64_t)blockIdx.x * nb_n + cb] = amax / 128.0f;
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int64_t row = row0 + r;
    if (row >= m) continue;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * 64 + lane) * 8;
      if (col >= n) continue;
      float f[8];
      if (r < RL) unpack8<DT>(stash[((wave * RL + (r < RL ? r : 0)) * NV + v) * 64 + lane], f);
      else unpack8<DT>(make_uint4(pk[r < RL ? 0 : r - RL][v][0], pk[r < RL ? 0 : r - RL][v][1], pk[r < RL ? 0 : r - RL][v][2],
                                  pk[r < RL ? 0 : r - RL][v][3]), f);
      uint32_t wd[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = rintf(f[j] * mult[v]);  // RNE
        t = fminf(fmaxf(t, -128.0f), 127.0f);
        wd[j >> 2] |= ((uint32_t)(int)t & 0xffu) << (8 * (j & 3));
      }
      *reinterpret_cast<uint2*>(q + row * n + col) = make_uint2(wd[0], wd[1]);
    }
  }
}

extern "C" int td_layernorm_quant(const void* x, int dtype, const float* w, const float* b, const float* scale,
                                  const float* shift, int64_t rows_per_batch, int8_t* q, float* qs, float eps,
                                  int64_t m, int64_t n, td_stream_t stream) {
  TD_REQUIRE(x && q && qs, TD_ERR_INVALID, "td_layernorm_quant: null pointer");
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_layernorm_quant: dtype %d (need f16|bf16)", dtype);
  TD_REQUIRE(m >= 0 && n > 0, TD_ERR_INVALID, "td_layernorm_quant: bad size");
  TD_REQUIRE(n % 8 == 0 && n <= 1536, TD_ERR_UNSUPPORTED,
             "td_layernorm_quant: n=%lld (need n %% 8 == 0 and n <= 1536; use td_layernorm + td_quant_i8_block128)", (long long)n);
  TD_REQUIRE((scale == nullptr) == (shift == nullptr), TD_ERR_INVALID, "td_layernorm_quant: scale/shift mismatch");
  TD_REQUIRE(scale == nullptr || rows_per_batch > 0, TD_ERR_INVALID, "td_layernorm_quant: rows_per_batch");
  TD_REQUIRE(b == nullptr || w != nullptr, TD_ERR_INVALID, "td_layernorm_quant: bias without weight");
  if (m == 0) return TD_OK;
  const int nv = (int)td_cdiv(n, 512);
  const int nb_n = (int)td_cdiv(n, 128);
  dim3 grid((unsigned)td_cdiv(m, 128));
  hipStream_t st = (hipStream_t)stream;
#define TD_LNQ(NV_, DT_)                                                                                  \
  layernorm_quant_kernel<NV_, DT_, (NV_ == 3 ? 6 : 0), 16><<<grid, 512, 0, st>>>((const uint16_t*)x, w, b, scale, shift,        \
                                                          rows_per_batch, q, qs, eps, m, (int)n, nb_n)
  if (dtype == TD_BF16) {
    if (nv <= 1) TD_LNQ(1, TD_BF16); else if (nv <= 2) TD_LNQ(2, TD_BF16); else TD_LNQ(3, TD_BF16);
  } else {
    if (nv <= 1) TD_LNQ(1, TD_F16); else if (nv <= 2) TD_LNQ(2, TD_F16); else TD_LNQ(3, TD_F16);
  }
#undef TD_LNQ
  TD_CHECK_LAUNCH();
  return TD_OK;
}
