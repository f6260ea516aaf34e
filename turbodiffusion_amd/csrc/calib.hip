// Measurement support, not an operator: what THIS box sustains right now, so that two bench.py runs on two boxes of the pool
// can be compared (the same tree gave 2.80 ... 3.04 videos/s on seven boxes in round 2).  bench.py calls these before its
// timed region and reports them in "box"; rooflines are then quoted against the datasheet peak AND against the box.
//   td_calib_mfma_i8   : dense v_mfma_i32_32x32x32_i8 on every SIMD (4 waves each, 4 independent chains per wave) — the
//                        matrix-pipe rate at the clock the power manager grants a matrix-bound kernel
//   td_calib_hbm_read  : streaming non-temporal 16-byte loads over a buffer far larger than the 256 MB Infinity Cache
//   td_calib_clock_probe: ONE wave that reads the shader-clock counter (s_memtime) and the constant 100 MHz counter
//                        (s_memrealtime), sleeps until a given number of 100 MHz ticks have passed and reads both again —
//                        launched on a second stream beside a real kernel it gives the shader clock during that kernel
#include "td_common.h"

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void calib_mfma_i8_kernel(int iters, float* sink) {
  const int lane = threadIdx.x & 63;
  v4i a = {lane, 1, 2, 3}, b = {3, 2, 1, lane};
  v16i c[4] = {};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[j], 0, 0, 0);
  if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 0x7fffffff) *sink = 1.f;
}

__global__ __launch_bounds__(256) void calib_hbm_read_kernel(const v4u* __restrict__ src, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const v4u v = __builtin_nontemporal_load(src + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(64) void calib_clock_probe_kernel(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = c0; out[1] = c1; out[2] = r0; out[3] = r1; }
}

// ops issued = blocks * 4 waves * iters * 4 MFMAs * 2*32*32*32
extern "C" int td_calib_mfma_i8(int iters, int blocks, float* sink, td_stream_t stream) {
  TD_REQUIRE(iters > 0 && blocks > 0 && sink, TD_ERR_INVALID, "td_calib_mfma_i8: iters=%d blocks=%d", iters, blocks);
  calib_mfma_i8_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, sink);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_calib_hbm_read(const void* src, int64_t bytes, void* sink, td_stream_t stream) {
  TD_REQUIRE(src && sink && bytes >= 16 && bytes % 16 == 0, TD_ERR_INVALID, "td_calib_hbm_read: bytes=%lld", (long long)bytes);
  calib_hbm_read_kernel<<<256 * 16, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const v4u*>(src), (size_t)(bytes / 16),
                                                                  reinterpret_cast<uint32_t*>(sink));
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// out: 4 x u64 device words {s_memtime start, end, s_memrealtime start, end}; ticks_100mhz <= 1e7 (0.1 s)
extern "C" int td_calib_clock_probe(int64_t ticks_100mhz, void* out, td_stream_t stream) {
  TD_REQUIRE(out && ticks_100mhz > 0 && ticks_100mhz <= 10000000, TD_ERR_INVALID, "td_calib_clock_probe: ticks=%lld",
             (long long)ticks_100mhz);
  calib_clock_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>((unsigned long long)ticks_100mhz,
                                                             reinterpret_cast<unsigned long long*>(out));
  TD_CHECK_LAUNCH();
  return TD_OK;
}
