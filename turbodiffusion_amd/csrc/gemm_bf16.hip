// 16-bit GEMM on the bf16 / fp16 matrix pipe:  D[b][m, n] = sum_k A[b][m, k] * B[b][n, k]  (+ bias[n]) (+ epilogue)
//
// What it is for (all PyTorch-library GEMMs of the product path until round 3):
//   * f4: the umT5 encoder's linears (rcm/utils/umt5.py:145-214: q|k|v, o, gate|fc1 with the gated-GELU product in the
//     epilogue, fc2), its per-head score / value products (batched), the Wan VAE middle-block attention as two batched
//     GEMMs around td_softmax_rows (rcm/tokenizers/wan2pt1.py:229-248);
//   * the text MLP of the DiT (wan2pt1.py:678) and the "bf16 linears" configuration (BASELINE config 3: nn.Linear in the
//     blocks, wan2pt1.py:226-232,375) — so that C3's number is this repo's kernel, not hipBLASLt's.
//
// Design: the W8A8 kernel's skeleton (gemm_w8a8_fi.hip) without its dequant.  A K step of 64 16-bit elements is 128 bytes
// per row — exactly the int8 kernel's 128-deep K block — so the LDS image (256 x 128 B per operand and stage, bank swizzle
// applied on the GLOBAL side of the LDS-DMA), the LDS-DMA piece schedule, the fragment addresses (one ds_read_b128 per
// 16-row x 32-k fragment: v_mfma_f32_16x16x32 takes 8 consecutive k per lane = 16 bytes, as v_mfma_i32_16x16x64_i8 takes
// 16), the one-barrier-per-K-step pipeline and the lane <-> (m, n) map of the results are the same; the accumulators are
// the MFMA's own fp32 C/D registers and the main loop contains no VALU at all.  256x256 tile, 512 threads (8 waves, wave
// tile 128 x 64), two stages of 64 KB.
//
// Arithmetic: fp32 accumulation in MFMA order (k ascending in steps of 32), one rounding to the output dtype, then the
// reference's operator sequence with its rounding points: + bias (rounded), GELU-tanh (rounded) | gated-GELU, + residual
// (rounded).  Parity is stated against an fp32 matmul of the same 16-bit operands (tests/test_gpu_f4.py).
#include "td_common.h"
#include <algorithm>

#define G_BM 256
#define G_BN 256
#define G_TILE (256 * 128)
#define G_STAGE (2 * G_TILE)
#define G_LDS (2 * G_STAGE)

typedef __attribute__((address_space(3))) void* g_lptr_t;

__device__ __forceinline__ uint32_t g_swz(uint32_t row, uint32_t chunk) {
  return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4);
}

#define G_FENCE()                             \
  {                                           \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("" ::: "memory");            \
  }
#define G_BARRIER()                           \
  {                                           \
    G_FENCE()                                 \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  }

struct Gemm16P {
  const uint16_t* A; const uint16_t* B; const uint16_t* bias; const uint16_t* R; void* D;
  int64_t M, N, K;                  // K % 64 == 0
  int64_t lda, ldb, ldd, ldr;       // row strides, elements
  int64_t sA, sB, sD, sR;           // batch strides, elements
  int tiles_m, tiles_n, group_m;
};

#define G_EPI_NONE 0
#define G_EPI_GELU 1     // GELU-tanh of the (bias-added, rounded) result, as the W8A8 kernels (td_gelu_tanh)
#define G_EPI_GELU_ERF 3 // exact GELU, nn.GELU() (Wan2.1 I2V's MLPProj on the CLIP tokens, wan2pt1.py:462-466): 0.5 x (1 + erf(x / sqrt 2))
#define G_EPI_GEGLU 2    // umT5 T5FeedForward (umt5.py:197-214): B's rows are gate / fc1 interleaved in blocks of 32
                         // (rows [64 p, 64 p + 32) = gate columns [32 p, 32 p + 32), rows [64 p + 32, 64 p + 64) = fc1
                         // columns of the same range); D [M, N / 2] = fc1(x) * GELU(gate(x)) with the reference's 16-bit
                         // rounding after every elementwise operation of its explicit tanh formula (umt5.py:125-127)

// umt5.py:125-127 on a value already rounded to the 16-bit dtype, every torch op rounding its result to that dtype:
// 0.5 * x * (1.0 + tanh(sqrt(2/pi) * (x + 0.044715 * pow(x, 3))))
template <int DT> __device__ __forceinline__ float g_t5_gelu(float x) {
  // torch.pow(x, 3.0) on a 16-bit tensor is `base * base * base` on the 16-bit scalar type (ATen's pow kernels, CPU and
  // GPU alike): the square is rounded before the third factor — two roundings (an fp32 cube rounded once differs in the last
  // place often enough to flip tanh's rounding further down: found by the GPU parity test, round 4)
  const float p3 = round_half<DT>(round_half<DT>(x * x) * x);
  const float a = round_half<DT>(0.044715f * p3);
  const float s = round_half<DT>(x + a);
  const float u = round_half<DT>(0.7978845608028654f * s);
  const float t = round_half<DT>(tanhf(u));
  const float o = round_half<DT>(1.0f + t);
  const float h = round_half<DT>(0.5f * x);
  return round_half<DT>(h * o);
}

// one more operator on a value already rounded to the output dtype (after bias): the exact GELU, rounded
template <int DT> __device__ __forceinline__ uint32_t g_gelu_erf2(uint32_t w) {
  float x0, x1;
  unpack2<DT>(w, x0, x1);
  return pack2<DT>(0.5f * x0 * (1.0f + erff(x0 * 0.7071067811865476f)), 0.5f * x1 * (1.0f + erff(x1 * 0.7071067811865476f)));
}

template <int IDT> struct g_mma;
template <> struct g_mma<TD_BF16> {
  __device__ static __forceinline__ void run(v4f& d, const v4i& a, const v4i& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
  }
};
template <> struct g_mma<TD_F16> {
  __device__ static __forceinline__ void run(v4f& d, const v4i& a, const v4i& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
  }
};

// IDT: operand dtype (bf16 | f16); ODT: output dtype (bf16 | f16 | f32; f32: plain epilogue only, no bias rounding games:
// acc + bias in fp32).  RES: D = round(D' + R) with D' the rounded epilogue result (x + Linear(...), 16-bit outputs only).
template <int IDT, int ODT, int EPI, bool HAS_BIAS, bool RES>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel(const Gemm16P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t M = p.M, N = p.N;
  const uint16_t* A = p.A + (int64_t)blockIdx.y * p.sA;
  const uint16_t* B = p.B + (int64_t)blockIdx.y * p.sB;

  // ---- tile assignment: XCD remap, then m-grouped raster (as gemm_w8a8_fi.hip) ----
  const uint32_t vid = xcd_remap(blockIdx.x, gridDim.x);
  const int per_group = p.group_m * p.tiles_n;
  const int gid = vid / per_group;
  const int first_m = gid * p.group_m;
  const int gsz = min(p.group_m, p.tiles_m - first_m);
  const int in_g = vid % per_group;
  const int tm = first_m + in_g % gsz;
  const int tn = in_g / gsz;
  const int64_t m0 = (int64_t)tm * G_BM, n0 = (int64_t)tn * G_BN;
  const int nk = (int)(p.K / 64);
  const int64_t ldab = p.lda * 2, ldbb = p.ldb * 2;   // bytes

  // ---- LDS-DMA pieces: wave w moves chunks c = w + 8t (8 rows x 128 B) of both operand tiles ----
  uint32_t ga[4], gb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = wave + 8 * t;
    const int row = 8 * c + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // source-side swizzle (LDS image is lane-linear)
    int64_t am = m0 + row; if (am > M - 1) am = M - 1;   // tail rows: clamp (never stored)
    int64_t bn = n0 + row; if (bn > N - 1) bn = N - 1;
    ga[t] = (uint32_t)(am * ldab + chunk * 16);
    gb[t] = (uint32_t)(bn * ldbb + chunk * 16);
  }
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(uint32_t)((M - 1) * ldab + p.K * 2), 0x00020000);
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(uint32_t)((N - 1) * ldbb + p.K * 2), 0x00020000);
#define G_PIECE(kb_, p_)                                                                          \
  {                                                                                               \
    char* sb_ = smem + ((kb_) & 1) * G_STAGE + wave * 1024;                                       \
    if ((p_) < 4)                                                                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (g_lptr_t)(sb_ + ((p_) & 3) * 8192), 16,   \
                                               ga[(p_) & 3], (kb_) * 128, 0, 0);                  \
    else                                                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (g_lptr_t)(sb_ + G_TILE + ((p_) & 3) * 8192), \
                                               16, gb[(p_) & 3], (kb_) * 128, 0, 0);              \
  }

  // ---- fragment read offsets (within a stage); B rows use the bit-2/3-swapped order so that after the MFMA a lane
  //      holds 4 CONSECUTIVE n (see the epilogue) ----
  const int pr = (l16 & 3) | ((l16 & 4) << 1) | ((l16 & 8) >> 1);
  uint32_t xoff[2], woff[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    xoff[kc] = g_swz(wm * 128 + l16, 4 * kc + lq);
    woff[kc] = G_TILE + g_swz(wn * 64 + pr, 4 * kc + lq);
  }

  v4f acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  v4i wf[4][2], xf[2][2];

#define G_LOAD_X(st_, i_, slot_)                                                                  \
  _Pragma("unroll") for (int kc = 0; kc < 2; ++kc)                                                \
    xf[slot_][kc] = *reinterpret_cast<const v4i*>((st_) + xoff[kc] + (i_) * 2048);
#define G_LOAD_W(st_, kc_)                                                                        \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
    wf[j][kc_] = *reinterpret_cast<const v4i*>((st_) + woff[kc_] + j * 2048);

  // ---- prologue: stage 0 and stage 1 in flight; wait for stage 0; fragments of group (0, 0) ----
#pragma unroll
  for (int q = 0; q < 8; ++q) G_PIECE(0, q)
  if (nk > 1) {
#pragma unroll
    for (int q = 0; q < 8; ++q) G_PIECE(1, q)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  G_BARRIER()
  G_LOAD_W(smem, 0)
  G_LOAD_W(smem, 1)
  G_LOAD_X(smem, 0, 0)

  for (int kb = 0; kb < nk; ++kb) {
    const char* st = smem + (kb & 1) * G_STAGE;
    const char* stn = smem + ((kb + 1) & 1) * G_STAGE;
    const bool more = kb + 1 < nk;
    const bool dma_tail = (kb >= 1) && more;   // rest of stage kb+1 (stage 1 was issued by the prologue)
    const bool dma_head = kb + 2 < nk;         // first pieces of stage kb+2, after this step's barrier
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int cur = i & 1, prv = cur ^ 1;
      // fragment of the next 16-row group into the other ring slot (its last readers were issued a group ago)
      if (i < 7) { G_LOAD_X(st, i + 1, prv) }
      else if (more) { G_LOAD_X(stn, 0, prv) }
      G_FENCE()
#pragma unroll
      for (int j = 0; j < 4; ++j) g_mma<IDT>::run(acc[i][j], wf[j][0], xf[cur][0]);
      G_FENCE()
      if (i == 7 && more) { G_LOAD_W(stn, 0) }
      // LDS-DMA issue, spread over the groups (VMEM issue slots of this wave only)
      if (i == 7) { if (dma_head) { G_PIECE(kb + 2, 0) G_PIECE(kb + 2, 4) } }
      else if (i == 0) { if (dma_tail) { G_PIECE(kb + 1, 1) G_PIECE(kb + 1, 5) } }
      else if (i == 1) { if (dma_tail) { G_PIECE(kb + 1, 2) } }
      else if (i == 2) { if (dma_tail) { G_PIECE(kb + 1, 6) } }
      else if (i == 3) { if (dma_tail) { G_PIECE(kb + 1, 3) } }
      else if (i == 4) { if (dma_tail) { G_PIECE(kb + 1, 7) } }
      G_FENCE()
#pragma unroll
      for (int j = 0; j < 4; ++j) g_mma<IDT>::run(acc[i][j], wf[j][1], xf[cur][1]);
      G_FENCE()
      if (i == 7 && more) { G_LOAD_W(stn, 1) }
      if (i == 6) {
        // every LDS read of stage kb has returned (group 7's fragment was read at the top of this group), this wave's
        // pieces of stage kb+1 have landed; after the barrier: everyone's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        G_BARRIER()
      }
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the last MFMAs' results before the VALU reads them

  // ---- epilogue.  A lane owns m = .. + l16; accumulator (i, j) holds n_local = 16 j + 8 (lq & 1) + 4 (lq >> 1) + r ----
  const int hi = lq >> 1;
  const uint16_t* bias = p.bias;
  if constexpr (ODT == TD_F32) {
    float* D = reinterpret_cast<float*>(p.D) + (int64_t)blockIdx.y * p.sD;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t m = m0 + wm * 128 + i * 16 + l16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = n0 + wn * 64 + j * 16 + 8 * (lq & 1) + 4 * hi;
        float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if constexpr (HAS_BIAS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < N) o[r] += half_bits_to_f32<IDT>(bias[n + r]);
        }
        if (m < M) {
          if (n + 3 < N && (p.ldd & 3) == 0) *reinterpret_cast<float4*>(D + m * p.ldd + n) = make_float4(o[0], o[1], o[2], o[3]);
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < N) D[m * p.ldd + n + r] = o[r];
          }
        }
      }
    }
    return;
  } else if constexpr (EPI == G_EPI_GEGLU) {
    uint16_t* D = reinterpret_cast<uint16_t*>(p.D) + (int64_t)blockIdx.y * p.sD;
    const int64_t No = N >> 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t m = m0 + wm * 128 + i * 16 + l16;
      uint32_t pk[2][2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = round_half<ODT>(acc[i][jj][r]);         // gate(x), rounded by the Linear
          const float f = round_half<ODT>(acc[i][jj + 2][r]);     // fc1(x)
          h[r] = f * g_t5_gelu<ODT>(g);                           // rounded at the pack
        }
        pk[jj][0] = pack2<ODT>(h[0], h[1]);
        pk[jj][1] = pack2<ODT>(h[2], h[3]);
      }
      auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
      auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
      const uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      const int jt = hi;   // lanes with lq < 2 hold 8 consecutive n of sub-tile 0, the others of sub-tile 1
      const int64_t n = (n0 >> 1) + wn * 32 + jt * 16 + 8 * (lq & 1);
      if (m < M && n < No) *reinterpret_cast<uint4*>(D + m * p.ldd + n) = v;
    }
    return;
  } else {
    uint16_t* D = reinterpret_cast<uint16_t*>(p.D) + (int64_t)blockIdx.y * p.sD;
    const uint16_t* R = RES ? p.R + (int64_t)blockIdx.y * p.sR : nullptr;
    __syncthreads();                    // every wave has read its last fragments: the stages may be overwritten
    float* ep_bias = reinterpret_cast<float*>(smem + 8 * (64 * 72) * 2);   // behind the 8 waves' staging regions
    if (tid < 256) {
      const int64_t n = n0 + tid;
      float bv = 0.f;
      if constexpr (HAS_BIAS) { if (n < N) bv = half_bits_to_f32<ODT>(bias[n]); }
      ep_bias[tid] = bv;
    }
    __syncthreads();
    uint16_t* stg = reinterpret_cast<uint16_t*>(smem) + wave * (64 * 72);
    constexpr int EP = EPI == G_EPI_GELU ? TD_EPI_GELU_TANH : TD_EPI_NONE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t pk[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float bf[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) {
          const float4 bb = *reinterpret_cast<const float4*>(ep_bias + wn * 64 + j * 16 + 8 * (lq & 1) + 4 * hi);
          bf[0] = bb.x; bf[1] = bb.y; bf[2] = bb.z; bf[3] = bb.w;
        }
        pk[j][0] = td_gemm_epilogue2<ODT, EP, HAS_BIAS>(acc[i][j][0], acc[i][j][1], bf[0], bf[1]);
        pk[j][1] = td_gemm_epilogue2<ODT, EP, HAS_BIAS>(acc[i][j][2], acc[i][j][3], bf[2], bf[3]);
        if constexpr (EPI == G_EPI_GELU_ERF) { pk[j][0] = g_gelu_erf2<ODT>(pk[j][0]); pk[j][1] = g_gelu_erf2<ODT>(pk[j][1]); }
      }
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int ja = 2 * jp, jb = 2 * jp + 1;
        auto s0 = __builtin_amdgcn_permlane32_swap(pk[ja][0], pk[jb][0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(pk[ja][1], pk[jb][1], false, false);
        const uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        const int jt = hi ? jb : ja;
        *reinterpret_cast<uint4*>(stg + ((i & 3) * 16 + l16) * 72 + jt * 16 + 8 * (lq & 1)) = v;
      }
      if ((i & 3) == 3) {   // a 64-row half is complete in LDS: 8 rows x one full 128-byte line per store instruction
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + (lane >> 3), chunk = lane & 7;
          const int64_t mm = m0 + wm * 128 + (i >> 2) * 64 + row, nn = n0 + wn * 64 + chunk * 8;
          uint4 r = *reinterpret_cast<const uint4*>(stg + row * 72 + chunk * 8);
          if (mm < M && nn < N) {
            if constexpr (RES) {
              float xf8[8], yf8[8];
              unpack8<ODT>(*reinterpret_cast<const uint4*>(R + mm * p.ldr + nn), xf8);
              unpack8<ODT>(r, yf8);
#pragma unroll
              for (int e = 0; e < 8; ++e) xf8[e] += yf8[e];
              r = pack8<ODT>(xf8);
            }
            if (nn + 8 <= N) *reinterpret_cast<uint4*>(D + mm * p.ldd + nn) = r;
            else {   // ragged N (not a multiple of 8): element stores for the last piece
              const uint32_t w4[4] = {r.x, r.y, r.z, r.w};
              for (int e = 0; e < 8 && nn + e < N; ++e) D[mm * p.ldd + nn + e] = (uint16_t)(w4[e >> 1] >> ((e & 1) * 16));
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// EXPERIMENT (round 4, TD_TUNE_GEMM16 = 2; measured EQUAL, not the default): the same 256x256 workgroup tile and LDS image,
// walked by FOUR waves of 128 x 128 instead of eight of 128 x 64.  Result (profiles/r04_gemm16_bench.jsonl, C3 shapes at L = 32 760,
// two runs): q|k|v 418-431 µs vs 464-473, o 164-166 vs 150-157, ffn.0 + GELU 861 vs 858, ffn.2 713-722 vs 755-819; C3 end to end 110.7
// vs 111.1 ms per DiT step — inside the spread: fewer LDS bytes per MFMA buy nothing once each SIMD holds ONE wave whose
// barrier and DMA waits nothing else covers.  (A trap on the way: with a bias / GELU / residual epilogue the `#pragma unroll`
// hint on the row-group loop was dropped, `acc[i]` became a dynamically indexed array and all 256 accumulators went to scratch
// — 10 x slower; `unroll(full)` fixed it.)  The idea was: per K step a 128 x 64 wave tile reads 24 KB of fragments for 64 MFMAs — 8 waves x 24 KB + 64 KB of
// DMA writes = 256 KB through a 128 B/clk LDS per 2048 matrix-pipe cycles: the kernel above is co-limited by LDS and the
// matrix pipe (counter: 0.52 busy, waves parked 45 %).  A 128 x 128 wave tile reads 32 KB for 128 MFMAs (4 waves x 32 KB + 64 KB
// = 192 KB per 2048 cycles: 73 % of the matrix time), at the price of 256 accumulator registers: they live in AGPRs (the MFMA's
// own C / D file; bf16 needs no VALU on them, which is what closes this road for the W8A8 kernel), one wave per SIMD with the
// whole 512-register file, latency hidden by the wave's own software pipeline (the next fragment is read while eight MFMAs
// of the current one issue).  Plain / bias / GELU / residual epilogues, 16-bit output; everything else stays on the kernel above.
// ------------------------------------------------------------------------------------------------------------------
template <int IDT> struct g_mma_b;
template <> struct g_mma_b<TD_BF16> {
  __device__ static __forceinline__ v4f run(const v4i& a, const v4i& b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
  }
};
template <> struct g_mma_b<TD_F16> {
  __device__ static __forceinline__ v4f run(const v4i& a, const v4i& b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
};

template <int IDT, int EPI, bool HAS_BIAS, bool RES>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4_kernel(const Gemm16P p) {
  constexpr int ODT = IDT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t M = p.M, N = p.N;
  const uint16_t* A = p.A + (int64_t)blockIdx.y * p.sA;
  const uint16_t* B = p.B + (int64_t)blockIdx.y * p.sB;

  const uint32_t vid = xcd_remap(blockIdx.x, gridDim.x);
  const int per_group = p.group_m * p.tiles_n;
  const int gid = vid / per_group;
  const int first_m = gid * p.group_m;
  const int gsz = min(p.group_m, p.tiles_m - first_m);
  const int in_g = vid % per_group;
  const int tm = first_m + in_g % gsz;
  const int tn = in_g / gsz;
  const int64_t m0 = (int64_t)tm * G_BM, n0 = (int64_t)tn * G_BN;
  const int nk = (int)(p.K / 64);
  const int64_t ldab = p.lda * 2, ldbb = p.ldb * 2;

  // ---- LDS-DMA pieces: wave w moves chunks c = w + 4t, t = 0..7 (8 rows x 128 B each) of both operand tiles ----
  uint32_t ga[8], gb[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int c = wave + 4 * t;
    const int row = 8 * c + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int64_t am = m0 + row; if (am > M - 1) am = M - 1;
    int64_t bn = n0 + row; if (bn > N - 1) bn = N - 1;
    ga[t] = (uint32_t)(am * ldab + chunk * 16);
    gb[t] = (uint32_t)(bn * ldbb + chunk * 16);
  }
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(uint32_t)((M - 1) * ldab + p.K * 2), 0x00020000);
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(uint32_t)((N - 1) * ldbb + p.K * 2), 0x00020000);
  // piece q of stage kb_ into buffer (kb_ & 1): q = 0..7 activation chunks, 8..15 weight chunks
#define W4_PIECE(kb_, q_)                                                                          \
  {                                                                                                \
    char* sb_ = smem + ((kb_) & 1) * G_STAGE + wave * 1024;                                        \
    if ((q_) < 8)                                                                                  \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (g_lptr_t)(sb_ + ((q_) & 7) * 4096), 16,    \
                                               ga[(q_) & 7], (kb_) * 128, 0, 0);                   \
    else                                                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (g_lptr_t)(sb_ + G_TILE + ((q_) & 7) * 4096), \
                                               16, gb[(q_) & 7], (kb_) * 128, 0, 0);               \
  }

  const int pr = (l16 & 3) | ((l16 & 4) << 1) | ((l16 & 8) >> 1);
  uint32_t xoff[2], woff[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    xoff[kc] = g_swz(wm * 128 + l16, 4 * kc + lq);
    woff[kc] = G_TILE + g_swz(wn * 128 + pr, 4 * kc + lq);
  }

  v4f acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  v4i wf[2][8], xf[2];   // weight fragments of BOTH halves of a K step (the other half is read while this one multiplies)

#define W4_LOAD_W(st_, kc_)                                                                        \
  _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                    \
    wf[kc_][j] = *reinterpret_cast<const v4i*>((st_) + woff[kc_] + j * 2048);
#define W4_LOAD_X(st_, kc_, i_, slot_) xf[slot_] = *reinterpret_cast<const v4i*>((st_) + xoff[kc_] + (i_) * 2048);

  // ---- prologue: stages 0 and 1 in flight; wait for stage 0 ----
#pragma unroll
  for (int q = 0; q < 16; ++q) W4_PIECE(0, q)
  if (nk > 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) W4_PIECE(1, q)
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  G_BARRIER()
  W4_LOAD_W(smem, 0)
  W4_LOAD_X(smem, 0, 0, 0)

  // One K step = 16 groups g = (kc, i): 8 MFMAs each against the 8 weight fragments of half kc.  The activation fragment of
  // group g + 1 is read at the top of group g; the weight fragments of the next half are re-read behind the last group that
  // uses the current ones.  ONE barrier per K step, before the last group: by then every LDS read of this stage has been
  // issued and returned and this wave's DMA pieces of the next-but-one stage... (see below) have landed.
  for (int kb = 0; kb < nk; ++kb) {
    const char* st = smem + (kb & 1) * G_STAGE;
    const char* stn = smem + ((kb + 1) & 1) * G_STAGE;
    const bool more = kb + 1 < nk;
    const bool dma = (kb >= 1) && more;          // stage kb + 1's pieces go out during THIS step (stage 1: the prologue)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int kc = g >> 3, i = g & 7;
      const int cur = g & 1, nxt = cur ^ 1;
      // next group's activation fragment
      if (g < 15) { W4_LOAD_X(st, (g + 1) >> 3, (g + 1) & 7, nxt) }
      else if (more) { W4_LOAD_X(stn, 0, 0, nxt) }
      // DMA of stage kb + 1 into the OTHER buffer: its last readers passed the barrier of step kb - 1.  All 16 pieces in the
      // first half of the step, so that the latest has ~1000 cycles before the wait below.
      if (dma && g < 8) { W4_PIECE(kb + 1, 2 * g) W4_PIECE(kb + 1, 2 * g + 1) }
      if (g == 1) { W4_LOAD_W(st, 1) }           // the second half's weight fragments: six groups before their first use
      if (g == 15 && more) { W4_LOAD_W(stn, 0) } // the next step's first half (behind the barrier of group 14), under this group's MFMAs
      G_FENCE()
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = g_mma_b<IDT>::run(wf[kc][j], xf[cur], acc[i][j]);
      G_FENCE()
      if (g == 14) {
        // every read of stage kb has been issued (group 15's fragment was read at the top of this group); wait for them and
        // for this wave's pieces of stage kb + 1, then meet the others: after the barrier stage kb + 1 is complete for
        // everyone and buffer kb & 1 may be overwritten by stage kb + 2 during the next step
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        G_BARRIER()
      }
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");

  // ---- epilogue: lane owns m = .. + l16; accumulator (i, j) holds n_local = 16 j + 8 (lq & 1) + 4 (lq >> 1) + r ----
  const int hi = lq >> 1;
  uint16_t* D = reinterpret_cast<uint16_t*>(p.D) + (int64_t)blockIdx.y * p.sD;
  const uint16_t* R = RES ? p.R + (int64_t)blockIdx.y * p.sR : nullptr;
  __syncthreads();                    // every wave has read its last fragments: the stages may be overwritten
  constexpr int WS = 136;             // staging row: 128 columns + 8 pad (16-bit)
  float* ep_bias = reinterpret_cast<float*>(smem + 4 * (64 * WS) * 2);
  if (tid < 256) {
    const int64_t n = n0 + tid;
    float bv = 0.f;
    if constexpr (HAS_BIAS) { if (n < N) bv = half_bits_to_f32<ODT>(p.bias[n]); }
    ep_bias[tid] = bv;
  }
  __syncthreads();
  uint16_t* stg = reinterpret_cast<uint16_t*>(smem) + wave * (64 * WS);
  constexpr int EP = EPI == G_EPI_GELU ? TD_EPI_GELU_TANH : TD_EPI_NONE;
  // (unroll(full), not the `unroll` hint: with a bias / GELU / residual body the hint is dropped, `acc[i]` becomes a dynamically
  // indexed array and all 256 accumulators move to scratch — the main loop then spills around every MFMA: 10 x slower, measured)
#pragma clang loop unroll(full)
  for (int i = 0; i < 8; ++i) {
    uint32_t pk[8][2];
#pragma clang loop unroll(full)
    for (int j = 0; j < 8; ++j) {
      float bf[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HAS_BIAS) {
        const float4 bb = *reinterpret_cast<const float4*>(ep_bias + wn * 128 + j * 16 + 8 * (lq & 1) + 4 * hi);
        bf[0] = bb.x; bf[1] = bb.y; bf[2] = bb.z; bf[3] = bb.w;
      }
      pk[j][0] = td_gemm_epilogue2<ODT, EP, HAS_BIAS>(acc[i][j][0], acc[i][j][1], bf[0], bf[1]);
      pk[j][1] = td_gemm_epilogue2<ODT, EP, HAS_BIAS>(acc[i][j][2], acc[i][j][3], bf[2], bf[3]);
      if constexpr (EPI == G_EPI_GELU_ERF) { pk[j][0] = g_gelu_erf2<ODT>(pk[j][0]); pk[j][1] = g_gelu_erf2<ODT>(pk[j][1]); }
    }
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const int ja = 2 * jp, jb = 2 * jp + 1;
      auto s0 = __builtin_amdgcn_permlane32_swap(pk[ja][0], pk[jb][0], false, false);
      auto s1 = __builtin_amdgcn_permlane32_swap(pk[ja][1], pk[jb][1], false, false);
      const uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      const int jt = hi ? jb : ja;
      *reinterpret_cast<uint4*>(stg + ((i & 3) * 16 + l16) * WS + jt * 16 + 8 * (lq & 1)) = v;
    }
    if ((i & 3) == 3) {   // a 64-row half is complete in LDS: 4 rows x two full 128-byte lines per store instruction
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4), chunk = lane & 15;
        const int64_t mm = m0 + wm * 128 + (i >> 2) * 64 + row, nn = n0 + wn * 128 + chunk * 8;
        uint4 r = *reinterpret_cast<const uint4*>(stg + row * WS + chunk * 8);
        if (mm < M && nn < N) {
          if constexpr (RES) {
            float xf8[8], yf8[8];
            unpack8<ODT>(*reinterpret_cast<const uint4*>(R + mm * p.ldr + nn), xf8);
            unpack8<ODT>(r, yf8);
#pragma unroll
            for (int e = 0; e < 8; ++e) xf8[e] += yf8[e];
            r = pack8<ODT>(xf8);
          }
          if (nn + 8 <= N) *reinterpret_cast<uint4*>(D + mm * p.ldd + nn) = r;
          else {
            const uint32_t w4[4] = {r.x, r.y, r.z, r.w};
            for (int e = 0; e < 8 && nn + e < N; ++e) D[mm * p.ldd + nn + e] = (uint16_t)(w4[e >> 1] >> ((e & 1) * 16));
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

template <int IDT, int EPI, bool HAS_BIAS, bool RES>
static int launch_gemm16_w4(const Gemm16P& p0, int batch, hipStream_t st) {
  auto kern = gemm_bf16_w4_kernel<IDT, EPI, HAS_BIAS, RES>;
  static std::atomic<uint64_t> attr_mask{0};
  td_ensure_dyn_lds(reinterpret_cast<const void*>(kern), G_LDS, attr_mask);
  Gemm16P p = p0;
  p.tiles_m = (int)td_cdiv(p.M, G_BM);
  p.tiles_n = (int)td_cdiv(p.N, G_BN);
  p.group_m = td_tuning(TD_TUNE_GEMM_GROUP_M) > 0 ? td_tuning(TD_TUNE_GEMM_GROUP_M) : 4;
  const dim3 grid((unsigned)p.tiles_m * (unsigned)p.tiles_n, (unsigned)batch, 1);
  kern<<<grid, 256, G_LDS, st>>>(p);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

template <int IDT, int ODT, int EPI, bool HAS_BIAS, bool RES>
static int launch_gemm16(const Gemm16P& p0, int batch, hipStream_t st) {
  auto kern = gemm_bf16_kernel<IDT, ODT, EPI, HAS_BIAS, RES>;
  static std::atomic<uint64_t> attr_mask{0};
  td_ensure_dyn_lds(reinterpret_cast<const void*>(kern), G_LDS, attr_mask);
  Gemm16P p = p0;
  p.tiles_m = (int)td_cdiv(p.M, G_BM);
  p.tiles_n = (int)td_cdiv(p.N, G_BN);
  p.group_m = td_tuning(TD_TUNE_GEMM_GROUP_M) > 0 ? td_tuning(TD_TUNE_GEMM_GROUP_M) : 4;
  const dim3 grid((unsigned)p.tiles_m * (unsigned)p.tiles_n, (unsigned)batch, 1);
  kern<<<grid, 512, G_LDS, st>>>(p);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

template <int IDT>
static int dispatch_gemm16(const Gemm16P& p, int out_dtype, int epilogue, int batch, hipStream_t st) {
  const bool hb = p.bias != nullptr, res = p.R != nullptr;
  if (out_dtype == TD_F32) {
    return hb ? launch_gemm16<IDT, TD_F32, G_EPI_NONE, true, false>(p, batch, st)
              : launch_gemm16<IDT, TD_F32, G_EPI_NONE, false, false>(p, batch, st);
  }
  // TD_TUNE_GEMM16 = 2: the four-wave kernel (experiment, kept selectable and tested: equal to the eight-wave kernel within
  // the run-to-run spread on every shape measured — see its header); default: the eight-wave kernel
  const int force = td_tuning(TD_TUNE_GEMM16);
  if (epilogue != G_EPI_GEGLU && force == 2) {
#define TD_W4(EPI_)                                                                                  \
    {                                                                                                \
      if (res) return hb ? launch_gemm16_w4<IDT, EPI_, true, true>(p, batch, st) : launch_gemm16_w4<IDT, EPI_, false, true>(p, batch, st); \
      return hb ? launch_gemm16_w4<IDT, EPI_, true, false>(p, batch, st) : launch_gemm16_w4<IDT, EPI_, false, false>(p, batch, st);      \
    }
    if (epilogue == G_EPI_NONE) TD_W4(G_EPI_NONE)
    if (!res && epilogue == G_EPI_GELU) TD_W4(G_EPI_GELU)
    if (!res && epilogue == G_EPI_GELU_ERF) TD_W4(G_EPI_GELU_ERF)
#undef TD_W4
  }
  if (epilogue == G_EPI_GEGLU) return launch_gemm16<IDT, IDT, G_EPI_GEGLU, false, false>(p, batch, st);
  if (epilogue == G_EPI_GELU) {
    return hb ? launch_gemm16<IDT, IDT, G_EPI_GELU, true, false>(p, batch, st)
              : launch_gemm16<IDT, IDT, G_EPI_GELU, false, false>(p, batch, st);
  }
  if (epilogue == G_EPI_GELU_ERF) {
    return hb ? launch_gemm16<IDT, IDT, G_EPI_GELU_ERF, true, false>(p, batch, st)
              : launch_gemm16<IDT, IDT, G_EPI_GELU_ERF, false, false>(p, batch, st);
  }
  if (res) {
    return hb ? launch_gemm16<IDT, IDT, G_EPI_NONE, true, true>(p, batch, st)
              : launch_gemm16<IDT, IDT, G_EPI_NONE, false, true>(p, batch, st);
  }
  return hb ? launch_gemm16<IDT, IDT, G_EPI_NONE, true, false>(p, batch, st)
            : launch_gemm16<IDT, IDT, G_EPI_NONE, false, false>(p, batch, st);
}

extern "C" int td_gemm_bf16(const void* a, const void* b, const void* bias, const void* res, void* d, int dtype, int out_dtype,
                            int epilogue, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldd, int64_t ldr,
                            int64_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_d, int64_t stride_r,
                            td_stream_t stream) {
  TD_REQUIRE(a && b && d, TD_ERR_INVALID, "td_gemm_bf16: null pointer");
  TD_REQUIRE(dtype == TD_BF16 || dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_gemm_bf16: operand dtype %d (bf16 | f16)", dtype);
  TD_REQUIRE(out_dtype == dtype || out_dtype == TD_F32, TD_ERR_UNSUPPORTED, "td_gemm_bf16: output dtype %d (the operand dtype | f32)", out_dtype);
  TD_REQUIRE(m > 0 && n > 0 && k > 0 && batch > 0 && batch < 65536, TD_ERR_INVALID, "td_gemm_bf16: empty problem or batch >= 65536");
  TD_REQUIRE(k % 64 == 0, TD_ERR_UNSUPPORTED, "td_gemm_bf16: k = %lld must be a multiple of 64 (pad the operands' rows with zeros)", (long long)k);
  TD_REQUIRE(lda >= k && ldb >= k && lda % 8 == 0 && ldb % 8 == 0, TD_ERR_INVALID, "td_gemm_bf16: lda / ldb must be >= k and multiples of 8");
  TD_REQUIRE(epilogue >= G_EPI_NONE && epilogue <= G_EPI_GELU_ERF, TD_ERR_INVALID, "td_gemm_bf16: epilogue %d", epilogue);
  const int64_t n_out = epilogue == G_EPI_GEGLU ? n / 2 : n;
  TD_REQUIRE(ldd >= n_out, TD_ERR_INVALID, "td_gemm_bf16: ldd < n");
  TD_REQUIRE(out_dtype == TD_F32 || ldd % 8 == 0, TD_ERR_UNSUPPORTED, "td_gemm_bf16: 16-bit output rows must start 16-byte aligned (ldd %% 8 == 0)");
  TD_REQUIRE(epilogue != G_EPI_GEGLU || (n % 64 == 0 && !bias && !res && out_dtype == dtype), TD_ERR_UNSUPPORTED,
             "td_gemm_bf16: the gated-GELU epilogue takes n %% 64 == 0 interleaved gate|fc1 rows, no bias, no residual");
  TD_REQUIRE(!res || (out_dtype == dtype && epilogue == G_EPI_NONE && ldr >= n && ldr % 8 == 0), TD_ERR_UNSUPPORTED,
             "td_gemm_bf16: the residual epilogue is 16-bit, plain, ldr %% 8 == 0");
  TD_REQUIRE(out_dtype != TD_F32 || epilogue == G_EPI_NONE, TD_ERR_UNSUPPORTED, "td_gemm_bf16: fp32 output has the plain epilogue only");
  // 32-bit byte offsets inside one batch entry's operands (buffer addressing)
  TD_REQUIRE((m - 1) * lda * 2 + k * 2 < (1ll << 32) && (n - 1) * ldb * 2 + k * 2 < (1ll << 32), TD_ERR_UNSUPPORTED,
             "td_gemm_bf16: an operand of one batch entry exceeds 4 GiB");
  Gemm16P p;
  p.A = (const uint16_t*)a; p.B = (const uint16_t*)b; p.bias = (const uint16_t*)bias; p.R = (const uint16_t*)res; p.D = d;
  p.M = m; p.N = n; p.K = k; p.lda = lda; p.ldb = ldb; p.ldd = ldd; p.ldr = ldr;
  p.sA = stride_a; p.sB = stride_b; p.sD = stride_d; p.sR = stride_r;
  p.tiles_m = p.tiles_n = p.group_m = 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TD_BF16 ? dispatch_gemm16<TD_BF16>(p, out_dtype, epilogue, (int)batch, st)
                          : dispatch_gemm16<TD_F16>(p, out_dtype, epilogue, (int)batch, st);
}

// ---- split-K (small M: umT5's linears at 64 ... 512 tokens, the text MLP).  A 256x256-tile GEMM with M <= 512 has a few dozen
// tiles for 256 CUs and every one walks the whole K: the weights stream through too few workgroups (86 us for the 100 MB of
// umT5's q|k|v at 64 tokens: 1.2 TB/s).  The host wrapper runs such a problem as a BATCH of S K-slices through the kernel
// above (batch strides = K/S columns of A and B, fp32 partial outputs [S, M, N]: no kernel change) and this pass adds the
// slices in fp32 and applies the epilogue with the same rounding points: cast -> + bias, cast -> GELU | gated GELU, cast ->
// + res, cast. ----
template <int DT, int EPI, bool HAS_BIAS, bool RES>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, int64_t M, int64_t N,
                                                            const uint16_t* __restrict__ bias, const uint16_t* __restrict__ R,
                                                            uint16_t* __restrict__ D, int64_t ldd, int64_t ldr) {
  const int64_t No = EPI == G_EPI_GEGLU ? N / 2 : N;
  const int64_t total = M * No;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / No, n = i % No;
    if constexpr (EPI == G_EPI_GEGLU) {
      const int64_t ng = (n >> 5) * 64 + (n & 31), nf = ng + 32;   // interleaved gate / fc1 rows of B = columns of the partials
      float g = 0.f, f = 0.f;
      for (int z = 0; z < S; ++z) { g += ws[((int64_t)z * M + m) * N + ng]; f += ws[((int64_t)z * M + m) * N + nf]; }
      g = round_half<DT>(g); f = round_half<DT>(f);
      D[m * ldd + n] = (uint16_t)f32_to_half_bits<DT>(f * g_t5_gelu<DT>(g));
    } else {
      float a = 0.f;
      for (int z = 0; z < S; ++z) a += ws[((int64_t)z * M + m) * N + n];
      a = round_half<DT>(a);
      if constexpr (HAS_BIAS) a = round_half<DT>(a + half_bits_to_f32<DT>(bias[n]));
      if constexpr (EPI == G_EPI_GELU) a = round_half<DT>(td_gelu_tanh(a));
      if constexpr (EPI == G_EPI_GELU_ERF) a = round_half<DT>(0.5f * a * (1.0f + erff(a * 0.7071067811865476f)));
      if constexpr (RES) a = a + half_bits_to_f32<DT>(R[m * ldr + n]);
      D[m * ldd + n] = (uint16_t)f32_to_half_bits<DT>(a);
    }
  }
}

extern "C" int td_gemm_bf16_splitk_reduce(const float* ws, int splits, const void* bias, const void* res, void* d, int dtype,
                                          int epilogue, int64_t m, int64_t n, int64_t ldd, int64_t ldr, td_stream_t stream) {
  TD_REQUIRE(ws && d && splits >= 1 && splits <= 64, TD_ERR_INVALID, "td_gemm_bf16_splitk_reduce: null pointer or splits = %d", splits);
  TD_REQUIRE(dtype == TD_BF16 || dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_gemm_bf16_splitk_reduce: dtype %d", dtype);
  TD_REQUIRE(m > 0 && n > 0 && epilogue >= G_EPI_NONE && epilogue <= G_EPI_GELU_ERF, TD_ERR_INVALID, "td_gemm_bf16_splitk_reduce: shape / epilogue");
  TD_REQUIRE(epilogue != G_EPI_GEGLU || (n % 64 == 0 && !bias && !res), TD_ERR_UNSUPPORTED, "td_gemm_bf16_splitk_reduce: gated GELU takes no bias / residual");
  TD_REQUIRE(!res || epilogue == G_EPI_NONE, TD_ERR_UNSUPPORTED, "td_gemm_bf16_splitk_reduce: residual with the plain epilogue only");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = m * (epilogue == G_EPI_GEGLU ? n / 2 : n);
  const unsigned grid = (unsigned)std::min<int64_t>(td_cdiv(total, 256), 4096);
#define TD_SKR(DT_, EPI_, HB_, RS_) \
  splitk_reduce_kernel<DT_, EPI_, HB_, RS_><<<grid, 256, 0, st>>>(ws, splits, m, n, (const uint16_t*)bias, (const uint16_t*)res, (uint16_t*)d, ldd, ldr)
#define TD_SKR_DT(DT_)                                                                             \
  if (epilogue == G_EPI_GEGLU) TD_SKR(DT_, G_EPI_GEGLU, false, false);                             \
  else if (epilogue == G_EPI_GELU) { if (bias) TD_SKR(DT_, G_EPI_GELU, true, false); else TD_SKR(DT_, G_EPI_GELU, false, false); } \
  else if (epilogue == G_EPI_GELU_ERF) { if (bias) TD_SKR(DT_, G_EPI_GELU_ERF, true, false); else TD_SKR(DT_, G_EPI_GELU_ERF, false, false); } \
  else if (res) { if (bias) TD_SKR(DT_, G_EPI_NONE, true, true); else TD_SKR(DT_, G_EPI_NONE, false, true); }                      \
  else { if (bias) TD_SKR(DT_, G_EPI_NONE, true, false); else TD_SKR(DT_, G_EPI_NONE, false, false); }
  if (dtype == TD_BF16) { TD_SKR_DT(TD_BF16) } else { TD_SKR_DT(TD_F16) }
#undef TD_SKR_DT
#undef TD_SKR
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// ---- row softmax (VAE middle-block attention: F.scaled_dot_product_attention's softmax(q k^T / sqrt(C)); umT5:
//      softmax(float(scores + position bias)), umt5.py:183-185) ----
// S [rows, lds] (f32, or the 16-bit dtype) -> P [rows, ldp] 16-bit: P[r, c] = softmax_c(scale * S[r, c] (+ bias[r % bias_rows, c])),
// columns [cols, ldp) are written as zeros (the P.V GEMM's K runs over the padded width).  With a 16-bit S the bias is added
// in that dtype first (one rounding: the reference's `einsum(...) + attn_bias`), the softmax itself is fp32 and rounded once.
template <int SDT, int PDT>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const void* __restrict__ S_, uint16_t* __restrict__ P, const uint16_t* __restrict__ bias,
                                                           int64_t rows, int cols, int64_t lds, int64_t ldp, int64_t bias_rows,
                                                           int64_t ldb, float scale) {
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ float red[8];
  auto load = [&](int c) -> float {
    float v;
    if constexpr (SDT == TD_F32) v = reinterpret_cast<const float*>(S_)[row * lds + c];
    else v = half_bits_to_f32<SDT>(reinterpret_cast<const uint16_t*>(S_)[row * lds + c]);
    if (bias != nullptr) {
      const float b = half_bits_to_f32<PDT>(bias[(row % bias_rows) * ldb + c]);
      v = (SDT == TD_F32) ? v + b : round_half<PDT>(v + b);
    }
    return v * scale;
  };
  float mx = -INFINITY;
  for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, load(c));
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid; c < cols; c += 256) sum += expf(load(c) - mx);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  const float inv = 1.0f / sum;
  for (int c = tid; c < (int)ldp; c += 256) {
    uint32_t o = 0;
    if (c < cols) o = f32_to_half_bits<PDT>(expf(load(c) - mx) * inv);
    P[row * ldp + c] = (uint16_t)o;
  }
}

extern "C" int td_softmax_rows(const void* s, int s_dtype, void* p, int p_dtype, const void* bias, int64_t rows, int64_t cols,
                               int64_t lds, int64_t ldp, int64_t bias_rows, int64_t ldb, float scale, td_stream_t stream) {
  TD_REQUIRE(s && p, TD_ERR_INVALID, "td_softmax_rows: null pointer");
  TD_REQUIRE(p_dtype == TD_BF16 || p_dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_softmax_rows: P dtype %d", p_dtype);
  TD_REQUIRE(s_dtype == TD_F32 || s_dtype == p_dtype, TD_ERR_UNSUPPORTED, "td_softmax_rows: S dtype %d (f32 or P's)", s_dtype);
  TD_REQUIRE(rows > 0 && rows < (1ll << 31) && cols > 0 && cols < (1 << 30) && lds >= cols && ldp >= cols, TD_ERR_INVALID,
             "td_softmax_rows: shape");
  TD_REQUIRE(!bias || (bias_rows > 0 && ldb >= cols), TD_ERR_INVALID, "td_softmax_rows: bias shape");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)rows;
#define TD_SM_CASE(SD, PD) \
  softmax_rows_kernel<SD, PD><<<g, 256, 0, st>>>(s, (uint16_t*)p, (const uint16_t*)bias, rows, (int)cols, lds, ldp, bias_rows ? bias_rows : 1, ldb, scale)
  if (p_dtype == TD_BF16) { if (s_dtype == TD_F32) TD_SM_CASE(TD_F32, TD_BF16); else TD_SM_CASE(TD_BF16, TD_BF16); }
  else { if (s_dtype == TD_F32) TD_SM_CASE(TD_F32, TD_F16); else TD_SM_CASE(TD_F16, TD_F16); }
#undef TD_SM_CASE
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// ---- T5LayerNorm (umt5.py:130-142): x * rsqrt(mean(float(x)^2) + eps) in fp32, cast to the weight's 16-bit dtype, THEN
//      multiplied by the weight in that dtype — two roundings (FastRMSNorm / td_rmsnorm multiplies in fp32 and rounds once).
//      x [rows, n] 16-bit (row stride ldx) -> y [rows, n] (row stride ldy); one 256-thread workgroup per row. ----
template <int DT>
__global__ __launch_bounds__(256) void t5_norm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                      uint16_t* __restrict__ y, int n, int64_t ldx, int64_t ldy, float eps) {
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ float red[4];
  const uint16_t* xr = x + row * ldx;
  float ss = 0.f;
  for (int c = tid; c < n; c += 256) { const float v = half_bits_to_f32<DT>(xr[c]); ss = fmaf(v, v, ss); }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  ss = (red[0] + red[1]) + (red[2] + red[3]);
  const float r = 1.0f / sqrtf(ss / (float)n + eps);
  for (int c = tid; c < n; c += 256) {
    const float xn = round_half<DT>(half_bits_to_f32<DT>(xr[c]) * r);
    y[row * ldy + c] = (uint16_t)f32_to_half_bits<DT>(half_bits_to_f32<DT>(w[c]) * xn);
  }
}

extern "C" int td_t5_norm(const void* x, const void* w, void* y, int dtype, float eps, int64_t rows, int64_t n, int64_t ldx,
                          int64_t ldy, td_stream_t stream) {
  TD_REQUIRE(x && w && y, TD_ERR_INVALID, "td_t5_norm: null pointer");
  TD_REQUIRE(dtype == TD_BF16 || dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_t5_norm: dtype %d (bf16 | f16)", dtype);
  TD_REQUIRE(rows > 0 && rows < (1ll << 31) && n > 0 && n < (1 << 30) && ldx >= n && ldy >= n, TD_ERR_INVALID, "td_t5_norm: shape");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16) t5_norm_kernel<TD_BF16><<<(unsigned)rows, 256, 0, st>>>((const uint16_t*)x, (const uint16_t*)w, (uint16_t*)y, (int)n, ldx, ldy, eps);
  else t5_norm_kernel<TD_F16><<<(unsigned)rows, 256, 0, st>>>((const uint16_t*)x, (const uint16_t*)w, (uint16_t*)y, (int)n, ldx, ldy, eps);
  TD_CHECK_LAUNCH();
  return TD_OK;
}
