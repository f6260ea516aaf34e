// v_mfma_(scale_)f32_32x32x64_f8f6f4 with fp8 e4m3 operands: operand layout check and issue rate.
//   hipcc --offload-arch=gfx950 -O3 -o fp8_mfma fp8_mfma.hip && ./fp8_mfma
// Assumed layout (CK xdlops_gemm.hpp: k_per_blk = 32, num_input_blks = 2): lane l holds, in 8 VGPRs = 32 bytes,
//   A[i = l % 32][k = 32 * (l / 32) + j], B[k = 32 * (l / 32) + j][n = l % 32], j = byte index 0..31;
// C/D as every 32x32 MFMA: lane l, reg r -> row (r & 3) + 8 (r >> 2) + 4 (l >> 5), col l & 31.
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_layout(const unsigned char* A, const unsigned char* B, float* D) {  // A [32][64], B [64][32] fp8 bytes
  const int l = threadIdx.x;
  v8i a, b;
  unsigned char ab[32], bb[32];
  for (int j = 0; j < 32; ++j) { ab[j] = A[(l % 32) * 64 + 32 * (l / 32) + j]; bb[j] = B[(32 * (l / 32) + j) * 32 + (l % 32)]; }
  for (int w = 0; w < 8; ++w) {
    a[w] = ab[4 * w] | (ab[4 * w + 1] << 8) | (ab[4 * w + 2] << 16) | (ab[4 * w + 3] << 24);
    b[w] = bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | (bb[4 * w + 3] << 24);
  }
  v16f c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

template <int MODE>  // 0: fp8 32x32x64 (unscaled lowering), 1: f16 32x32x16 for comparison
__global__ __launch_bounds__(256) void k_rate(int iters, float* sink, unsigned long long* out) {
  const int l = threadIdx.x & 63;
  v8i a, b;
  for (int w = 0; w < 8; ++w) { a[w] = 0x38383838 + l; b[w] = 0x38383838 + w; }
  v16f c[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
  typedef _Float16 v8h __attribute__((ext_vector_type(8)));
  v8h ah, bh;
  for (int w = 0; w < 8; ++w) { ah[w] = (_Float16)(l + w); bh[w] = (_Float16)w; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MODE == 0) c[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[q], 0, 0, 0, 0, 0, 0);
      else c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[q], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

static unsigned char f2fp8(float f) { __hip_fp8_e4m3 v(f); return *reinterpret_cast<unsigned char*>(&v); }

int main() {
  unsigned char hA[32 * 64], hB[64 * 32];
  float vA[32 * 64], vB[64 * 32];
  srand(1);
  const float vals[8] = {0.f, 1.f, -1.f, 2.f, 0.5f, -3.f, 1.5f, 4.f};   // exactly representable in e4m3
  for (int i = 0; i < 32 * 64; ++i) { vA[i] = vals[rand() % 8]; hA[i] = f2fp8(vA[i]); }
  for (int i = 0; i < 64 * 32; ++i) { vB[i] = vals[rand() % 8]; hB[i] = f2fp8(vB[i]); }
  unsigned char *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dA, dB, dD);
  float hD[32 * 32];
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
    double ref = 0;
    for (int k = 0; k < 64; ++k) ref += (double)vA[i * 64 + k] * vB[k * 32 + n];
    maxerr = fmax(maxerr, fabs(ref - hD[i * 32 + n]));
  }
  printf("layout check (asymmetric random A, B): max |D - ref| = %g  -> %s\n", maxerr, maxerr == 0 ? "LAYOUT OK" : "LAYOUT WRONG");
  float* sink; unsigned long long* out;
  hipMalloc(&sink, 1024 * 256 * 4); hipMalloc(&out, 8);
  for (int mode = 0; mode < 2; ++mode) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (mode == 0) k_rate<0><<<1024, 256>>>(10, sink, out); else k_rate<1><<<1024, 256>>>(10, sink, out);
    hipEventRecord(e0);
    if (mode == 0) k_rate<0><<<1024, 256>>>(iters, sink, out); else k_rate<1><<<1024, 256>>>(iters, sink, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    const double flop = (mode == 0 ? 2.0 * 32 * 32 * 64 : 2.0 * 32 * 32 * 16) * 4.0 * iters * 1024 * 4;
    printf("%s: %.1f TFLOP/s, %.1f s_memtime ticks per MFMA per wave (1 wave/SIMD, 4 independent accumulators)\n",
           mode == 0 ? "fp8 e4m3 32x32x64 (f8f6f4)" : "f16 32x32x16", flop / (ms * 1e-3) / 1e12, (double)h / (4.0 * iters));
  }
  return 0;
}
