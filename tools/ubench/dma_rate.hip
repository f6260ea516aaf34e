// Microbenchmark: sustained global -> LDS (LDS-DMA, buffer_load_dwordx4 ... lds) rate with every CU loading,
// as a function of where the data sits (own XCD's L2 / Infinity Cache / HBM), of how many 1-KB pieces each wave
// keeps in flight, and of the access shape (8 rows x 128 B with a row stride, like a GEMM operand tile, or 1 KB
// contiguous).  512-thread workgroups (8 waves), one per CU, like the W8A8 GEMM.
//   hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip && ./dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lptr_t;

// each WG walks `steps` K blocks of a [rows x ld] byte matrix region starting at its base; per K block it loads
// `tile_rows` rows x 128 B (tile_rows/8 pieces spread over 8 waves).  INFLIGHT pieces per wave before waiting.
template <int INFLIGHT>
__global__ __launch_bounds__(512, 2) void k(const int8_t* base, long long bytes_total, long long wg_stride,
                                            int share, int ld, int tile_rows, int steps, int wrap, int contiguous) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroups b, b+8, b+16, ... run on the same XCD (block b -> XCD b % 8): groups of `share` such WGs read the same data
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const long long region = ((long long)(idx / share) * 8 + xcd) * wg_stride;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + region % bytes_total), 0, 0x7fffffff, 0x00020000);
  const int pieces = tile_rows / 64;  // per wave per K block
  uint32_t off[8];
  for (int p = 0; p < 8; ++p) {
    if (contiguous) off[p] = (uint32_t)(((p * 8 + wave) * 64 + lane) * 16);
    else { const int row = (p * 8 + wave) * 8 + (lane >> 3); off[p] = (uint32_t)(row * ld + (lane & 7) * 16); }
  }
  int issued = 0;
  for (int s = 0; s < steps; ++s) {
    const int kk = s % wrap;
    const uint32_t soff = contiguous ? (uint32_t)kk * (uint32_t)(tile_rows * 128) : (uint32_t)kk * 128u;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if (p < pieces) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(smem + ((s & 1) * 8 + p) * 8192 + wave * 1024), 16, off[p], soff, 0, 0);
        if (++issued >= INFLIGHT) {
          if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          else if (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
          else if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

template <int INFLIGHT>
void run(const char* name, const int8_t* buf, long long bytes_total, long long wg_stride, int share, int ld,
         int tile_rows, int steps, int wrap, int contiguous) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  k<INFLIGHT><<<256, 512, 131072>>>(buf, bytes_total, wg_stride, share, ld, tile_rows, steps, wrap, contiguous);
  hipEventRecord(e0);
  k<INFLIGHT><<<256, 512, 131072>>>(buf, bytes_total, wg_stride, share, ld, tile_rows, steps, wrap, contiguous);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * steps * tile_rows * 128.0;
  printf("%-64s inflight/wave %2d  %.3f ms  %.2f TB/s  (%.1f GB/s per CU)\n", name, INFLIGHT, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  const long long total = 3ll << 30;
  int8_t* buf; hipMalloc(&buf, total + (64 << 20)); hipMemset(buf, 1, total);
  // GEMM-like tile: 512 rows (256 act + 256 weight) x 128 B per K block = 64 KB, row stride 1536, 12 K blocks re-read (L2 resident)
  run<8>("L2-resident, GEMM tile 512 rows x128B, ld 1536, private", buf, total, 1 << 20, 1, 1536, 512, 1200, 1, 0);
  run<8>("L2-resident, GEMM tile, ld 1536, shared by 4 WGs/XCD", buf, total, 1 << 20, 4, 1536, 512, 1200, 1, 0);
  run<8>("L2-resident, contiguous 64 KB per step, private", buf, total, 1 << 20, 1, 1536, 512, 1200, 1, 1);
  run<2>("L2-resident, GEMM tile, private", buf, total, 1 << 20, 1, 1536, 512, 1200, 1, 0);
  run<4>("L2-resident, GEMM tile, private", buf, total, 1 << 20, 1, 1536, 512, 1200, 1, 0);
  run<16>("L2-resident, GEMM tile, private", buf, total, 1 << 20, 1, 1536, 512, 1200, 1, 0);
  // Infinity-Cache resident: each WG streams 768 KB (x256 = 192 MB) repeatedly -> L2 misses, MALL hits
  run<8>("MALL-resident (192 MB total), GEMM tile ld 1536, private", buf, total, 768 << 10, 1, 1536, 512, 1200, 12 * 1, 0);
  run<8>("MALL-resident 192 MB, contiguous", buf, total, 768 << 10, 1, 1536, 512, 1200, 12, 1);
  // HBM: each WG streams 8 MB (x256 = 2 GB)
  run<8>("HBM (2 GB total), GEMM tile ld 8960 x 70 K blocks... private", buf, total, 8 << 20, 1, 8960, 512, 1400, 70, 0);
  run<8>("HBM 2 GB, contiguous 64 KB steps", buf, total, 8 << 20, 1, 1536, 512, 1400, 128, 1);
  run<16>("HBM 2 GB, contiguous 64 KB steps", buf, total, 8 << 20, 1, 1536, 512, 1400, 128, 1);
  // mixed like the GEMM: shared by 6 WGs per XCD, ld 8960, streaming 70 K blocks of fresh data (HBM first touch + 5 L2 hits)
  run<8>("GEMM-like: 6 WGs/XCD share a 512-row panel, ld 8960, stream", buf, total, 8 << 20, 6, 8960, 512, 1400, 1400, 0);
  run<8>("GEMM-like: 4 WGs/XCD share, ld 1536, 12-block panels re-read", buf, total, 1 << 20, 4, 1536, 512, 1200, 12, 0);
  return 0;
}
