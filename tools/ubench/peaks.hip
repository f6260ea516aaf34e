// Microbenchmark: what this MI355X actually sustains, to state next to the datasheet peaks used in the rooflines
// (SURVEY §8d): HBM read / write / copy bandwidth on buffers far larger than the 256 MB Infinity Cache, and the dense
// matrix-pipe rate of the MFMA instructions the hot path issues (whole chip, 4 waves per SIMD, independent chains).
//   hipcc --offload-arch=gfx950 -O3 -o peaks peaks.hip && ./peaks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
  } while (0)

__global__ __launch_bounds__(256) void k_read(const v4u* __restrict__ src, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const v4u v = __builtin_nontemporal_load(src + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(v4u* __restrict__ dst, size_t n) {
  const v4u v = {1u, 2u, 3u, threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v, dst + i);
}
__global__ __launch_bounds__(256) void k_copy(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// KIND 0: v_mfma_i32_16x16x64_i8, 1: v_mfma_i32_32x32x32_i8, 2: v_mfma_f32_32x32x16_f16, 3: v_mfma_f32_16x16x32_f16,
//      4: v_mfma_f32_32x32x16_bf16
template <int KIND>
__global__ __launch_bounds__(256) void k_mfma(int iters, float* sink) {
  const int lane = threadIdx.x & 63;
  if constexpr (KIND == 0) {
    v4i a = {lane, 1, 2, 3}, b = {3, 2, 1, lane}, c[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j)  // (asm: the builtin form makes this compiler shuffle the accumulators through AGPR copies)
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
    if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 0x7fffffff) *sink = 1.f;
  } else if constexpr (KIND == 1) {
    v4i a = {lane, 1, 2, 3}, b = {3, 2, 1, lane};
    v16i c[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[j], 0, 0, 0);
    if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 0x7fffffff) *sink = 1.f;
  } else if constexpr (KIND == 2) {
    v8h a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f); b[i] = (_Float16)0.5f; }
    v16f c[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
    if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 12345.f) *sink = 1.f;
  } else if constexpr (KIND == 3) {
    v8h a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f); b[i] = (_Float16)0.5f; }
    v4f c[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
    if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 12345.f) *sink = 1.f;
  } else {
    v8b a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f); b[i] = (__bf16)0.5f; }
    v16f c[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
    if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 12345.f) *sink = 1.f;
  }
}

template <typename F>
static float time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs, clock %d MHz, memory clock %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000,
         prop.memoryClockRate / 1000);
  const size_t bytes = (size_t)4 << 30;  // 4 GiB per buffer, 16x the Infinity Cache
  v4u *a = nullptr, *b = nullptr;
  uint32_t* sink = nullptr;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 256));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  const size_t n = bytes / 16;
  for (int wg_per_cu : {4, 8, 16}) {
    const int grid = cus * wg_per_cu;
    float ms = time_ms([&] { k_read<<<grid, 256>>>(a, n, sink); }, 5);
    printf("HBM read   %2d WG/CU: %7.1f GB/s\n", wg_per_cu, bytes / ms / 1e6);
    ms = time_ms([&] { k_write<<<grid, 256>>>(b, n); }, 5);
    printf("HBM write  %2d WG/CU: %7.1f GB/s\n", wg_per_cu, bytes / ms / 1e6);
    ms = time_ms([&] { k_copy<<<grid, 256>>>(a, b, n); }, 5);
    printf("HBM copy   %2d WG/CU: %7.1f GB/s (read + write bytes)\n", wg_per_cu, 2.0 * bytes / ms / 1e6);
  }
  {
    float ms = time_ms([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 5);
    printf("hipMemcpy D2D       : %7.1f GB/s (read + write bytes)\n", 2.0 * bytes / ms / 1e6);
  }
  const int iters = 20000;
  const int grid = cus * 4;   // 4 WGs x 4 waves per CU = 4 waves per SIMD
  const double waves = (double)grid * 4;
  struct { const char* name; double flop; int kind; } tests[] = {
      {"v_mfma_i32_16x16x64_i8  ", 2.0 * 16 * 16 * 64, 0}, {"v_mfma_i32_32x32x32_i8  ", 2.0 * 32 * 32 * 32, 1},
      {"v_mfma_f32_32x32x16_f16 ", 2.0 * 32 * 32 * 16, 2}, {"v_mfma_f32_16x16x32_f16 ", 2.0 * 16 * 16 * 32, 3},
      {"v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, 4}};
  for (auto& t : tests) {
    float ms = 0.f;
    float* fs = reinterpret_cast<float*>(sink);
    switch (t.kind) {
      case 0: ms = time_ms([&] { k_mfma<0><<<grid, 256>>>(iters, fs); }, 3); break;
      case 1: ms = time_ms([&] { k_mfma<1><<<grid, 256>>>(iters, fs); }, 3); break;
      case 2: ms = time_ms([&] { k_mfma<2><<<grid, 256>>>(iters, fs); }, 3); break;
      case 3: ms = time_ms([&] { k_mfma<3><<<grid, 256>>>(iters, fs); }, 3); break;
      default: ms = time_ms([&] { k_mfma<4><<<grid, 256>>>(iters, fs); }, 3); break;
    }
    const double total = waves * iters * 4.0 * t.flop;
    printf("%s: %8.1f TFLOP/s dense (%.3f ms)\n", t.name, total / ms / 1e9, ms);
  }
  return 0;
}
