// Microbenchmark: what rate does one gfx950 SIMD sustain on the W8A8 main-loop instruction mix
// (v_mfma_i32_16x16x64_i8 : v_add_f32 : v_fmac_f32 = 1 : 2 : 2), as a function of interleave granularity,
// waves per SIMD, and LDS / LDS-DMA traffic beside it?  Prints s_memtime ticks per MFMA slot and the
// effective shader clock (ticks / wall).
//   hipcc --offload-arch=gfx950 -O3 -o mix_rate mix_rate.hip && ./mix_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA0(d, a, b, c) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
#define MFMA1(d, a, b) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
#define ADD(x) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x) : "s"(-12582912.0f));
#define FMAC(acc, x, s) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(s), "v"(x));
#define CVT(x) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x));

// G = MFMAs per cluster (1, 2, 4, 8): cluster = G MFMAs then 4G VALU.  MODE: 0 mix, 1 MFMA only, 2 VALU only,
// 3 mix with v_cvt instead of the magic add, 4 mix + 3 ds_read_b128 per 8 slots, 5 mix with VALU first
template <int G, int MODE>
__global__ __launch_bounds__(512, 2) void k(int iters, int nwaves, unsigned long long* out, float* sink, float sc_in) {
  __shared__ v4i lds[4096];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  v4i a[4], b[2], magic = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
  for (int i = 0; i < 4; ++i) a[i] = (v4i){lane + i, lane * 3 + i, i, lane ^ i};
  b[0] = (v4i){lane, 1, 2, 3}; b[1] = (v4i){3, 2, 1, lane};
  asm volatile("" : "+v"(magic));
  v4i t[2][4];
  float acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = lane + i;
  for (int j = 0; j < 4; ++j) { t[0][j] = magic; t[1][j] = magic; }
  lds[threadIdx.x] = a[0]; lds[threadIdx.x + 512] = a[1];
  const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc_in)));
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < nwaves) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {  // two groups of 8 slots per iteration (ring of 2)
        const int cur = g, prv = g ^ 1;
        if (MODE == 4) {
          b[0] = lds[(lane + it) & 1023]; b[1] = lds[512 + ((lane + it) & 1023)]; a[3] = lds[1024 + lane];
        }
#pragma unroll
        for (int c = 0; c < 8 / G; ++c) {
          if (MODE == 5) {
#pragma unroll
            for (int s = 0; s < G; ++s) {
              const int slot = c * G + s, j = slot & 3;
#pragma unroll
              for (int r = 0; r < 4; ++r) { if (slot < 4) { ADD(t[prv][j][r]) } else { FMAC(acc[(g * 16 + j * 4 + r) & 31], t[prv][j][r], sc) } }
            }
          }
          if (MODE != 2) {
#pragma unroll
            for (int s = 0; s < G; ++s) {
              const int slot = c * G + s, j = slot & 3;
              if (slot < 4) { MFMA0(t[cur][j], a[j], b[0], magic) } else { MFMA1(t[cur][j], a[j], b[1]) }
            }
          }
          if (MODE != 1 && MODE != 5) {
#pragma unroll
            for (int s = 0; s < G; ++s) {
              const int slot = c * G + s, j = slot & 3;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (slot < 4) { if (MODE == 3) { CVT(t[prv][j][r]) } else { ADD(t[prv][j][r]) } }
                else { FMAC(acc[(g * 16 + j * 4 + r) & 31], t[prv][j][r], sc) }
              }
            }
          }
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 32; ++i) s += acc[i];
  for (int j = 0; j < 4; ++j) s += t[0][j][0] + t[1][j][1];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) out[wave] = t1 - t0;
}


typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define MFMA32_0(d, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
#define MFMA32_1(d, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
// 32x32x32: a sub-tile's K block = chain of 4 MFMAs (32 cycles each), 16 results -> 16 cvt + 16 fmac: slot = 1 MFMA + 8 VALU.
// MODE 0: mix, 1: MFMA only
template <int MODE>
__global__ __launch_bounds__(512, 2) void k32(int iters, int nwaves, unsigned long long* out, float* sink, float sc_in) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  v4i a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = (v4i){lane + i, lane * 3 + i, i, lane ^ i}; b[i] = (v4i){lane, i, 2, 3}; }
  v16i t[2];
  for (int r = 0; r < 16; ++r) { t[0][r] = r; t[1][r] = lane; }
  float acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = lane + i;
  const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc_in)));
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < nwaves) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int cur = g, prv = g ^ 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c == 0) { MFMA32_0(t[cur], a[0], b[0]) } else { MFMA32_1(t[cur], a[c], b[c]) }
          if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { if (c < 2) { CVT(t[prv][(c * 4 + r) * 2]) CVT(t[prv][(c * 4 + r) * 2 + 1]) }
              else { FMAC(acc[(g * 16 + (c - 2) * 8 + r * 2) & 31], t[prv][((c - 2) * 4 + r) * 2], sc) FMAC(acc[(g * 16 + (c - 2) * 8 + r * 2 + 1) & 31], t[prv][((c - 2) * 4 + r) * 2 + 1], sc) } }
          }
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 32; ++i) s += acc[i];
  s += t[0][0] + t[1][1];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) out[wave] = t1 - t0;
}
// packed VALU: slot = 1 MFMA 16x16x64 + 1 v_pk_add_f32 + 1 v_pk_fma_f32 (2 elements each)
__global__ __launch_bounds__(512, 2) void kpk(int iters, int nwaves, unsigned long long* out, float* sink, float sc_in) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  v4i a[4], b[2], magic = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
  for (int i = 0; i < 4; ++i) a[i] = (v4i){lane + i, lane * 3 + i, i, lane ^ i};
  b[0] = (v4i){lane, 1, 2, 3}; b[1] = (v4i){3, 2, 1, lane};
  asm volatile("" : "+v"(magic));
  v4i t[2][4];
  v2f acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (v2f){(float)lane, (float)i};
  for (int j = 0; j < 4; ++j) { t[0][j] = magic; t[1][j] = magic; }
  v2f dq[4][2];
  v2f psc = {sc_in, sc_in}, pm = {-12582912.0f, -12582912.0f};
  asm volatile("" : "+v"(psc), "+v"(pm));
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < nwaves) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int cur = g, prv = g ^ 1;
#pragma unroll
        for (int slot = 0; slot < 8; ++slot) {
          const int j = slot & 3;
          if (slot < 4) { MFMA0(t[cur][j], a[j], b[0], magic) } else { MFMA1(t[cur][j], a[j], b[1]) }
          if (slot < 4) {
            dq[j][0] = (v2f){__builtin_bit_cast(float, t[prv][j][0]), __builtin_bit_cast(float, t[prv][j][1])};
            dq[j][1] = (v2f){__builtin_bit_cast(float, t[prv][j][2]), __builtin_bit_cast(float, t[prv][j][3])};
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dq[j][0]) : "v"(pm));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dq[j][1]) : "v"(pm));
          } else {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(g * 8 + j * 2) & 15]) : "v"(dq[j][0]), "v"(psc));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(g * 8 + j * 2 + 1) & 15]) : "v"(dq[j][1]), "v"(psc));
          }
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
  for (int j = 0; j < 4; ++j) s += t[0][j][0] + t[1][j][1];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) out[wave] = t1 - t0;
}
template <class F>
void run2(const char* name, F kern, int nwaves, double slots_per_iter, unsigned long long* d, float* sink) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<256, 512>>>(10, nwaves, d, sink, 1.0001f);
  hipEventRecord(e0);
  kern<<<256, 512>>>(iters, nwaves, d, sink, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  const double slots = slots_per_iter * iters;
  printf("%-58s waves/SIMD %d  ticks/slot w0 %.2f w4 %.2f  wall %.3f ms  ticks/us %.0f\n", name, nwaves / 4,
         h[0] / slots, nwaves > 4 ? h[4] / slots : 0.0, ms, (nwaves > 4 ? h[4] : h[0]) / (ms * 1e3));
}

template <int G, int MODE>
void run(const char* name, int nwaves, unsigned long long* d, float* sink) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<G, MODE><<<256, 512>>>(10, nwaves, d, sink, 1.0001f);
  hipEventRecord(e0);
  k<G, MODE><<<256, 512>>>(iters, nwaves, d, sink, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  const double slots = 16.0 * iters;  // per wave
  printf("%-58s waves/SIMD %d  ticks/slot w0 %.2f w4 %.2f  wall %.3f ms  ticks/us %.0f  SIMD-cycles per 2-wave slot pair @2.4GHz %.1f\n", name, nwaves / 4,
         h[0] / slots, nwaves > 4 ? h[4] / slots : 0.0, ms, h[0] / (ms * 1e3), ms * 1e-3 * 2.4e9 / slots);
}

int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 64); hipMalloc(&sink, 256 * 512 * 4);
  run<1, 1>("MFMA only", 4, d, sink);
  run<1, 1>("MFMA only", 8, d, sink);
  run<1, 2>("VALU only (16 add + 16 fmac per 8 slots)", 4, d, sink);
  run<1, 2>("VALU only", 8, d, sink);
  run<1, 0>("mix 1 MFMA : 4 VALU", 4, d, sink);
  run<1, 0>("mix 1 MFMA : 4 VALU", 8, d, sink);
  run<2, 0>("mix 2 MFMA : 8 VALU", 8, d, sink);
  run<4, 0>("mix 4 MFMA : 16 VALU", 8, d, sink);
  run<8, 0>("mix 8 MFMA : 32 VALU", 8, d, sink);
  run<4, 0>("mix 4 MFMA : 16 VALU", 4, d, sink);
  run<1, 3>("mix 1:4 with v_cvt_f32_i32 instead of magic add", 8, d, sink);
  run<1, 4>("mix 1:4 + 3 ds_read_b128 per 8 slots", 8, d, sink);
  run<1, 5>("mix 1:4, VALU before the MFMA in each slot", 8, d, sink);
  run2("32x32x32: MFMA only (slot = 1 MFMA = 2x work of a 16x16x64)", k32<1>, 4, 8, d, sink);
  run2("32x32x32: MFMA only", k32<1>, 8, 8, d, sink);
  run2("32x32x32: mix 1 MFMA : 8 VALU (cvt+fmac)", k32<0>, 4, 8, d, sink);
  run2("32x32x32: mix 1 MFMA : 8 VALU (cvt+fmac)", k32<0>, 8, 8, d, sink);
  run2("16x16x64 + packed VALU: 1 MFMA : 1 pk_add + 1 pk_fma... x2", kpk, 4, 16, d, sink);
  run2("16x16x64 + packed VALU", kpk, 8, 16, d, sink);
  return 0;
}
