// Packed fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) issue rate vs the scalar forms in a VALU-only stream
// (the GEMM's GELU + quantiser epilogue is such a stream), one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>  // 0: 16 independent v_fma_f32 per iteration, 1: 8 independent v_pk_fma_f32 (same flops)
__global__ __launch_bounds__(512) void k_rate(int iters, float* sink, unsigned long long* out) {
  float a[16];
  v2f p[8];
  const float s = 1.0f + threadIdx.x * 1e-7f, t = 0.5f;
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) { p[i][0] = a[2 * i]; p[i][1] = a[2 * i + 1]; }
  const v2f s2 = {s, s}, t2 = {t, t};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(s2), "v"(t2));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += a[i];
  for (int i = 0; i < 8; ++i) r += p[i][0] + p[i][1];
  sink[blockIdx.x * 512 + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  float* sink; unsigned long long* out;
  hipMalloc(&sink, 1024 * 512 * 4); hipMalloc(&out, 8);
  const int iters = 2000;
  for (int threads : {256, 512}) {   // 1 or 2 waves per SIMD of the CU
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) k_rate<0><<<1, threads>>>(iters, sink, out); else k_rate<1><<<1, threads>>>(iters, sink, out);
        hipDeviceSynchronize();
      }
      unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
      // s_memtime ticks at 100 MHz; report ticks per iteration and per 16 lane-flops-pairs
      printf("%d waves/SIMD  %s: %.3f memtime ticks per iteration of 16 fma results (x%d waves)\n", threads / 256,
             mode ? "8 x v_pk_fma_f32" : "16 x v_fma_f32  ", (double)h / iters, threads / 256);
    }
  }
  return 0;
}
