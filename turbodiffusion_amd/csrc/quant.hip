// a16 — per-128x128-block INT8 quantiser for gfx950.
// Reference semantics: turbodiffusion/ops/quant/quant.hpp:91-98,122-164 (amax, 128/amax, RNE
// saturating convert, scale = amax/128) with the zero-fill of tail blocks from
// ops/common/load.hpp:24-47.  HBM-bound: 2 B read + 1 B written per element.
//
// Mapping (MI355X-first, not the reference's 256x64-element-per-thread CUDA tiling):
// one 256-thread workgroup (4 waves) per 128x128 block; a wave reads 4 rows x 256 B per
// instruction (16 B per lane, fully coalesced), all 8 row-groups are in flight before the
// first use; the block amax is a 64-lane butterfly + 4-entry LDS exchange; each lane then
// writes 8 B of int8 (128 B contiguous per row).
#include "td_common.h"

template <int DT>
__global__ __launch_bounds__(256) void quant_block128_kernel(const uint16_t* __restrict__ x,
                                                             int8_t* __restrict__ q,
                                                             float* __restrict__ s, int64_t m,
                                                             int64_t n, int nb_n) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int bn = blockIdx.x, bm = blockIdx.y;
  const int c8 = tid & 15;   // which 8-element (16 B) column group of the 128-wide block
  const int r0 = tid >> 4;   // row within a 16-row group
  const int64_t col = (int64_t)bn * 128 + c8 * 8;
  const bool col_ok = col < n;  // n % 8 == 0, so a vector is all-in or all-out

  uint4 raw[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int64_t row = (int64_t)bm * 128 + it * 16 + r0;
    raw[it] = make_uint4(0, 0, 0, 0);
    if (col_ok && row < m) raw[it] = *reinterpret_cast<const uint4*>(x + row * n + col);
  }
  float amax = 1e-8f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    float f[8];
    unpack8<DT>(raw[it], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
  }
  amax = wave_max(amax);
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float mult = 128.0f / amax;  // IEEE division (no fast-math in this build)
  if (tid == 0) s[(int64_t)bm * nb_n + bn] = amax / 128.0f;

#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int64_t row = (int64_t)bm * 128 + it * 16 + r0;
    if (!(col_ok && row < m)) continue;
    float f[8];
    unpack8<DT>(raw[it], f);
    uint32_t w[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = rintf(f[j] * mult);  // RNE
      v = fminf(fmaxf(v, -128.0f), 127.0f);
      w[j >> 2] |= ((uint32_t)(int)v & 0xffu) << (8 * (j & 3));
    }
    *reinterpret_cast<uint2*>(q + row * n + col) = make_uint2(w[0], w[1]);
  }
}

extern "C" int td_quant_i8_block128(const void* x, int dtype, int8_t* q, float* s, int64_t m,
                                    int64_t n, td_stream_t stream) {
  TD_REQUIRE(x && q && s, TD_ERR_INVALID, "td_quant_i8_block128: null pointer");
  TD_REQUIRE(m >= 0 && n >= 0, TD_ERR_INVALID, "td_quant_i8_block128: negative size");
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED,
             "td_quant_i8_block128: dtype %d (need f16|bf16)", dtype);
  TD_REQUIRE(n % 8 == 0, TD_ERR_UNSUPPORTED, "td_quant_i8_block128: n=%lld not a multiple of 8",
             (long long)n);
  if (m == 0 || n == 0) return TD_OK;
  const int nb_n = (int)td_cdiv(n, 128);
  dim3 grid(nb_n, (unsigned)td_cdiv(m, 128));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16)
    quant_block128_kernel<TD_BF16><<<grid, 256, 0, st>>>((const uint16_t*)x, q, s, m, n, nb_n);
  else
    quant_block128_kernel<TD_F16><<<grid, 256, 0, st>>>((const uint16_t*)x, q, s, m, n, nb_n);
  TD_CHECK_LAUNCH();
  return TD_OK;
}
