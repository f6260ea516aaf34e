// Microbenchmark: do the matrix pipe (v_mfma_i32_16x16x64_i8) and the VALU of ONE SIMD overlap when the
// MFMAs come from one wave and the VALU ops from its co-resident partner?  512-thread workgroups,
// 1 per CU: waves 0-3 and waves 4-7 share SIMDs.  mode bit0: waves 0-3 run MFMAs; bit1: waves 4-7 run VALU
// (v_cvt_f32_i32 + v_fma_f32 pairs, the W8A8 dequant); bit2: waves 4-7 run MFMAs too; bit3: waves 0-3 run
// the VALU mix interleaved 4:1 with their own MFMAs.  Prints cycles per MFMA / per VALU op.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void k(int mode, int iters, unsigned long long* out, float* sink) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  v4i a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, 1, 2, 3};
  v4i acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (v4i){0, 0, 0, 0};
  float f[32];
  int ti[32];
  for (int i = 0; i < 32; ++i) { f[i] = lane * 0.5f + i; ti[i] = lane + i; }
  const float sc = 1.0001f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const bool grp0 = wave < 4;
  const bool do_mfma = (grp0 && (mode & 1)) || (!grp0 && (mode & 4));
  const bool do_valu = (!grp0 && (mode & 2));
  const bool do_mix = (grp0 && (mode & 8)) || (!grp0 && (mode & 16));
  const bool do_pk = (!grp0 && (mode & 32));
  const bool do_mixpk = (grp0 && (mode & 64)) || (!grp0 && (mode & 128));
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f pf[16], pt[16];
  for (int i = 0; i < 16; ++i) { pf[i] = (v2f){lane * 1.f, i * 1.f}; pt[i] = (v2f){i * 2.f, lane * 3.f}; }
  const v2f psc = {sc, sc}, pm = {-12582912.0f, -12582912.0f};
  if (do_mix) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) f[4 * i + r] = fmaf((float)ti[4 * i + r], sc, f[4 * i + r]);
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(f[i]), "+v"(ti[i]));
    }
  } else if (do_mixpk) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          v2f u;
          asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(u) : "v"(pt[2 * i + r]), "v"(pm));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pf[2 * i + r]) : "v"(u), "v"(psc));
        }
      }
    }
  } else if (do_pk) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v2f u;
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(u) : "v"(pt[i]), "v"(pm));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pf[i]) : "v"(u), "v"(psc));
      }
    }
  } else if (!grp0 && (mode & 256)) {   // 8-byte encodings: v_add_f32 with a literal + VOP3 v_fma_f32
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, 0xcb400000, %0" : "+v"(f[i]));
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i + 7) & 31]), "v"(sc));
    }
  } else if (!grp0 && (mode & 512)) {   // 4-byte encodings: v_add_f32 with an SGPR + v_fmac_f32
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[i]) : "s"(-12582912.0f));
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 7) & 31]), "v"(sc));
    }
  } else if (grp0 && (mode & 1024)) {  // MFMAs with 8 distinct A and 2 distinct B fragments, C = constant VGPR quad (like the GEMM)
    v4i aa[8], bb[2], cst = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
    for (int i = 0; i < 8; ++i) aa[i] = (v4i){lane + i, lane * 3 + i, i, lane ^ i};
    bb[0] = b; bb[1] = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aa[2 * i], bb[0], cst, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aa[2 * i + 1], bb[1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[i]));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(aa[i]));
    }
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) f[i] = fmaf((float)ti[i], sc, f[i]);
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(f[i]), "+v"(ti[i]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 32; ++i) s += f[i];
  for (int i = 0; i < 16; ++i) s += pf[i][0] + pf[i][1];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) out[wave] = t1 - t0;
}

int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 64); hipMalloc(&sink, 256 * 512 * 4);
  const int iters = 2000;
  const int modes[] = {1, 1024, 1026, 1028, 257, 512, 513};
  const char* names[] = {"MFMA w0-3 only", "GEMM-like MFMA (distinct frags, const C) w0-3 only", "GEMM-like MFMA w0-3 || VALU w4-7", "GEMM-like MFMA w0-3 || plain MFMA w4-7", "8-byte VALU (literal add + VOP3 fma) w4-7 only",
                         "MFMA w0-3 || 8-byte VALU w4-7", "4-byte VALU (sgpr add + fmac) w4-7 only", "MFMA w0-3 || 4-byte VALU w4-7"};
  const char* names_old[] = {"MFMA on waves0-3 only", "VALU(cvt+fma) on waves4-7 only", "MFMA w0-3 || VALU w4-7",
                         "MFMA on both waves of each SIMD", "mix (8 MFMA + 32 cvt + 32 fma) on w0-3 only",
                         "mix w0-3 || VALU w4-7", "mix on BOTH waves of each SIMD", "PK (16 pk_add + 16 pk_fma = same work) w4-7 only",
                         "MFMA w0-3 || PK w4-7", "mixPK (8 MFMA + 16 pk_add + 16 pk_fma) w0-3 only", "mixPK on BOTH waves"};
  for (int m = 0; m < 4; ++m) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 512>>>(modes[m], 10, d, sink);
    hipEventRecord(e0);
    k<<<256, 512>>>(modes[m], iters, d, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("%-48s wall %.3f ms | ticks/iter wave0 %.1f wave4 %.1f  (iter = 8 MFMA and/or 32 cvt+32 fma)\n", names[m], ms,
           (double)h[0] / iters, (double)h[4] / iters);
  }
  return 0;
}
