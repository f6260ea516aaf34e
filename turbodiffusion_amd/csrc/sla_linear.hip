// a14 — SLA linear-attention branch on CDNA4 matrix cores (gfx950).
// Reference: SLA/core.py:243-253 (feature_map = softmax):
//   cq = softmax_D(q).to(dt); ck = softmax_D(k).to(dt)
//   kvsum = ck^T @ v ; ksum = sum_L ck
//   o_l = (cq @ kvsum) / (1e-5 + sum_D(cq*ksum)) ; o_l = proj_l(o_l) (autocast dt) ; o = o_s + o_l
//
// Two passes, both token-streaming (HBM-bound) with the small GEMMs on MFMA:
//   td_sla_linear_kv : per head, 16 workgroups each reduce a range of 64-token K blocks:
//        ck tile is written TRANSPOSED into LDS in the same key order as the V^T tiles produced
//        by td_v_transpose, so kv[d1,d2] += ck^T[d1,tok] v[tok,d2] is a plain A.B MFMA with both
//        operands read by ds_read_b128.  Partials go to a workspace (no atomics, deterministic),
//        a finalize kernel sums them in order and rounds once.
//   td_sla_linear_out: per 128-token block; lane = token (the two half-waves hold the two halves
//        of the row), so softmax_D, the normaliser and the final divide are lane-local;
//        num^T = kvsum^T.cq^T and out^T = Wp.o_l^T are chained MFMAs where the first one's
//        accumulator registers ARE the second one's B fragments (Wp is stored in LDS in the
//        matching k order).  The result is added to o_s in place.
#include "td_common.h"

#define LK_NCH TD_SLA_NCH  // partial-sum chunks per head (include/turbodiffusion_amd.h)

__device__ __forceinline__ uint32_t sw128(uint32_t row, uint32_t slot) {  // 128-B rows
  return row * 128u + ((slot ^ ((row >> 1) & 7u)) << 4);
}
__device__ __forceinline__ uint32_t sw256(uint32_t row, uint32_t slot) {  // 256-B rows
  return row * 256u + ((slot ^ (row & 15u)) << 4);
}
// position inside a 16-group that holds offset tt (0..15): order 0-3, 8-11, 4-7, 12-15
__device__ __forceinline__ int perm_pos(int tt) { return (tt & 3) + ((tt >> 3) & 1) * 4 + ((tt >> 2) & 1) * 8; }

template <int DT> struct MmaT;
template <> struct MmaT<TD_F16> {
  typedef v8h frag;
  __device__ static __forceinline__ v16f mma(frag a, frag b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaT<TD_BF16> {
  typedef v8bf frag;
  __device__ static __forceinline__ v16f mma(frag a, frag b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// softmax over D in the exp2 domain with v_exp_f32 / v_rcp_f32 (1 ulp each; the result is rounded to a
// 16-bit type right after, so this agrees with expf()/division to that rounding except in ~1e-4 of cases)
#define TD_LOG2E 1.4426950408889634f

// ---------------------------------------------------------------------------------------
// pass 1a: partial kv / ksum over a range of K blocks
// FM: the linear branch's feature map (SLA/core.py:57-72): 0 = softmax over D (the published checkpoints), 1 = elu(x) + 1, 2 = relu —
// the two elementwise maps with torch's 16-bit rounding after every op (F.elu rounds, "+ 1" rounds)
template <int DT, int FM> __device__ __forceinline__ float sla_feature(float x) {
  if constexpr (FM == 1) {
    const float e = round_half<DT>(x > 0.f ? x : expm1f(x));
    return round_half<DT>(e + 1.0f);
  } else {
    return fmaxf(x, 0.f);
  }
}
//   KDT: dtype of k (and of the rounded softmax ck);  VDT: dtype of the V^T tiles / MFMA
// Thread (tg = tid/16, c8 = tid%16) owns tokens 4tg..4tg+3 x channels 8c8..8c8+7 of each 64-token block:
// 4 coalesced 16-B row loads, the row softmax is a 16-lane butterfly, and the transposed image ck^T[d][tok]
// is written as 8-byte pieces (4 consecutive tokens of one channel are 4 consecutive MFMA positions).
// ---------------------------------------------------------------------------------------
template <int KDT, int VDT, bool WANT_KM, int FM = 0>
__global__ __launch_bounds__(256, 3) void linear_kv_partial_kernel(const uint16_t* __restrict__ k,
                                                                const uint16_t* __restrict__ vt,
                                                                float* __restrict__ ws_kv,
                                                                float* __restrict__ ws_ks,
                                                                float* __restrict__ ws_km, int64_t L,
                                                                int Kb, int Kb_alloc, int hg, int64_t gs) {
  // vt may live in the sequence-parallel pack (td_head_off, td_common.h): Kb_alloc tiles allocated per head, (hg, gs)
  // Round 6: the pass ran at 0.35 of the HBM rate — 384 workgroups (32 chunks x 12 heads) of 228 VGPRs, two per CU, each a
  // serial chain of [load 32 KB -> softmax -> LDS -> 16 MFMAs] with one tile of look-ahead staged through 16 VGPRs.  Now the V^T
  // tile (16 KB, contiguous) arrives by LDS-DMA into a two-slot ring (no staging registers, no ds_write), the rounded
  // softmax is kept packed, 48 KB of LDS and <= 168 VGPRs let THREE workgroups share a CU, and TD_SLA_NCH = 64 chunks per head
  // give 768 of them at C1: one full round.
  __shared__ __attribute__((aligned(16))) char ckT[128 * 128];      // [d1][64 positions] 16-bit
  __shared__ __attribute__((aligned(16))) char vT1[128 * 128];      // [d2][64 positions] 16-bit, lane-linear DMA image: ONE buffer —
  // the next tile's pieces are issued right behind the barrier that ends the MFMA phase and fly under the next softmax phase
  __shared__ float kmred[WANT_KM ? 16 : 1][128];                    // the smooth-K column sums accumulate HERE (8 registers per lane
  // more than the 168 three workgroups per CU allow: the compiler spilled 45 dwords and the pass took 130 us instead of 60)
  float (*ksred)[128] = reinterpret_cast<float (*)[128]>(&vT1[0]);  // (after the loop: 16 x 128 floats)
  typedef typename MmaT<VDT>::frag frag;
  typedef __attribute__((address_space(3))) void* lptr_k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int c8 = tid & 15, tg = tid >> 4;
  const int ch = blockIdx.x, h = blockIdx.y;
  const int per = (Kb + LK_NCH - 1) / LK_NCH;
  const int kb_lo = ch * per, kb_hi = min(Kb, kb_lo + per);
  const int wr = wave >> 1, wc = wave & 1;
  // tokens 4tg..4tg+3 sit at positions p0..p0+3 of 16-group tg/4: 8-byte piece (p0/4) of slot 2*(tg/4) + p0/8
  const int p0 = perm_pos((4 * tg) & 15);
  const uint32_t ck_slot = (uint32_t)((tg >> 2) * 2 + (p0 >> 3));
  const uint32_t ck_sub = (uint32_t)((p0 & 7) * 2);

  v16f acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ks_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // plain column sums of k (rows past L count as 0) -> the smooth-K mean: accumulated in LDS, slot [tg][8 c8 + j] is this lane's
  // alone.  Measured on one box (tools/glue_bench.py, the whole entry point incl. its two finalisers): round 5's kernel (32 chunks,
  // 228 VGPRs, two workgroups per CU, V^T staged through registers) 77.0 us with the mean, 65.7 without; this one 71.9 / 58.9;
  // with the sums in 8 registers the compiler spills 45 dwords at 168 VGPRs (130 us), at two workgroups per CU 85 us, pairs of
  // rows added first 77 us, LDS float atomics (ds_add_f32) 303 us.
  if constexpr (WANT_KM) {
#pragma unroll
    for (int j = 0; j < 8; ++j) kmred[tg][c8 * 8 + j] = 0.f;
  }
  // V^T tile kb -> ring slot kb & 1: 16 pieces of 1 KB (8 rows x 128 B), wave w moves pieces w + 4t.  The LDS image of a piece
  // is lane-linear, so the read side's bank swizzle (sw128) goes onto the global address; it does not depend on t (32 rows per
  // step), so ONE address register serves the four pieces through the instruction's scalar offset.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(vt + td_head_off(h, hg, gs, (int64_t)Kb_alloc * (128 * 64))), 0, 0x7fffffff, 0x00020000);
  const uint32_t v_off0 = (uint32_t)((8 * wave_u + (lane >> 3)) * 128 + (((lane & 7) ^ ((4 * wave_u + (lane >> 4)) & 7)) * 16));
#define LK_VDMA(kb_)                                                                                       \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                            \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (lptr_k)(&vT1[0] + (wave_u + 4 * t) * 1024), 16, v_off0, \
                                             (uint32_t)(kb_) * 16384u + (uint32_t)t * 4096u, 0, 0);
  // this thread's 4 K rows of a block: ALWAYS four load instructions per wave (rows past L re-read row L - 1 and are zeroed),
  // so that the counts behind the s_waitcnt below are the same for every wave
  uint4 kr[4];
#define LK_KLOAD(kb_)                                                                                      \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                          \
    int64_t l_ = (int64_t)(kb_) * 64 + 4 * tg + t;                                                         \
    if (l_ > L - 1) l_ = L - 1;                                                                            \
    kr[t] = *reinterpret_cast<const uint4*>(k + ((int64_t)h * L + l_) * 128 + c8 * 8);                     \
  }
  if (kb_lo < kb_hi) {
    LK_VDMA(kb_lo)
    LK_KLOAD(kb_lo)
  }
  for (int kb = kb_lo; kb < kb_hi; ++kb) {
    const bool more = kb + 1 < kb_hi;
    // softmax over D (16 lanes share a row), rounded to KDT, then to VDT: two rows at a time, kept PACKED (ckp[j][pair])
    uint32_t ckp[8][2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      float ck2[2][8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * pr + u;
        const bool ok = (int64_t)kb * 64 + 4 * tg + t < L;
        float f[8];
        if constexpr (WANT_KM) {   // (a row past L is a re-read of row L - 1: it must count as zeros in the column sums)
          if (!ok) kr[t] = make_uint4(0, 0, 0, 0);
        }
        unpack8<KDT>(kr[t], f);
        if constexpr (WANT_KM) {   // (same order of additions per slot as a register accumulator: bit-identical sums)
#pragma unroll
          for (int j = 0; j < 8; ++j) kmred[tg][c8 * 8 + j] += f[j];
        }
        if constexpr (FM != 0) {   // elementwise feature map (already a 16-bit value), zero for rows past the end
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            ck2[u][j] = ok ? sla_feature<KDT, FM>(f[j]) : 0.f;
            ks_acc[j] += ck2[u][j];
          }
          continue;
        }
        float mx = f[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) mx = fmaxf(mx, f[j]);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float mb = mx * TD_LOG2E;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { f[j] = __builtin_amdgcn_exp2f(fmaf(f[j], TD_LOG2E, -mb)); sum += f[j]; }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = ok ? __builtin_amdgcn_rcpf(sum) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const uint32_t w = pack2<KDT>(f[j] * inv, f[j + 1] * inv);  // softmax(...).to(dtype)
          unpack2<KDT>(w, ck2[u][j], ck2[u][j + 1]);
          ks_acc[j] += ck2[u][j];
          ks_acc[j + 1] += ck2[u][j + 1];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ckp[j][pr] = pack2<VDT>(ck2[0][j], ck2[1][j]);
    }
    if (more) { LK_KLOAD(kb + 1) }  // this thread's K rows of the next block (the current ones are consumed)
    // transposed write: for channel d1 = 8c8+j the 4 tokens are one 8-byte piece
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t d1 = (uint32_t)(c8 * 8 + j);
      *reinterpret_cast<uint2*>(ckT + sw128(d1, ck_slot) + ck_sub) = make_uint2(ckp[j][0], ckp[j][1]);
    }
    // this wave's pieces of tile kb are older than the K rows just consumed (loads return in order); younger and allowed to
    // stay in flight: the next block's 4 K-row loads
    if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* vT = &vT1[0];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      frag a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const frag*>(ckT + sw128(32 * (2 * wr + i) + li, 2 * ks + hi));
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const frag*>(vT + sw128(32 * (2 * wc + j) + li, 2 * ks + hi));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MmaT<VDT>::mma(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
    if (more) { LK_VDMA(kb + 1) }     // every wave has read tile kb's fragments: the buffer takes the next tile
  }
#undef LK_VDMA
#undef LK_KLOAD
  // partial kv: D[d1][d2], lane = d2 column, registers = d1 rows
  float* out = ws_kv + ((int64_t)h * LK_NCH + ch) * (128 * 128);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d1 = 32 * (2 * wr + i) + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int d2 = 32 * (2 * wc + j) + li;
        out[d1 * 128 + d2] = acc[i][j][r];
      }
#pragma unroll
  for (int j = 0; j < 8; ++j) ksred[tg][c8 * 8 + j] = ks_acc[j];
  __syncthreads();
  if (tid < 128) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += ksred[r][tid];
    ws_ks[((int64_t)h * LK_NCH + ch) * 128 + tid] = s;
  }
  if constexpr (WANT_KM) {  // first stage of td_seq_mean on the way (second stage: td_seq_mean_final over LK_NCH partials)
    __syncthreads();
    if (tid < 128) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += kmred[r][tid];
      ws_km[((int64_t)h * LK_NCH + ch) * 128 + tid] = s;
    }
  }
}

// pass 1b: sum partials in order, round once; kvsum is written TRANSPOSED ([d2][d1]) because
// that is the A operand of pass 2.  Partial (h, c) lives at ws + h*stride_h + c*stride_c, so the
// same kernel finishes the single-GPU workspace [H,16,..] and a rank-major gathered [N,H,..] one.
template <int DT, bool ROUND>
__global__ __launch_bounds__(256) void linear_kv_final_kernel(const float* __restrict__ ws_kv,
                                                              const float* __restrict__ ws_ks, int nch,
                                                              int64_t kv_sh, int64_t kv_sc, int64_t ks_sh,
                                                              int64_t ks_sc, void* __restrict__ kv_out,
                                                              void* __restrict__ ks_out, int hg, int64_t gs) {
  // (hg, gs in output elements): the outputs may live in the sequence-parallel pack (td_head_off, td_common.h)
  // 64 workgroups per head, one kv element per thread; the partials are summed in a fixed tree (four interleaved
  // running sums, then ((s0+s1)+(s2+s3))) with four loads in flight: deterministic, and not a chain of nch latencies
  const int h = blockIdx.x, part = blockIdx.y;
  {
    const int i = part * 256 + threadIdx.x;
    const float* p = ws_kv + h * kv_sh + i;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 3 < nch; c += 4) {
      s0 += p[(int64_t)c * kv_sc];
      s1 += p[(int64_t)(c + 1) * kv_sc];
      s2 += p[(int64_t)(c + 2) * kv_sc];
      s3 += p[(int64_t)(c + 3) * kv_sc];
    }
    for (; c < nch; ++c) s0 += p[(int64_t)c * kv_sc];
    const float s = (s0 + s1) + (s2 + s3);
    if constexpr (ROUND) {
      const int d1 = i >> 7, d2 = i & 127;
      ((uint16_t*)kv_out)[td_head_off(h, hg, gs, 128 * 128) + d2 * 128 + d1] = (uint16_t)f32_to_half_bits<DT>(s);
    } else {
      ((float*)kv_out)[td_head_off(h, hg, gs, 128 * 128) + i] = s;  // fp32, NOT transposed: still a partial
    }
  }
  if (part == 0 && threadIdx.x < 128) {
    const float* p = ws_ks + h * ks_sh + threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 3 < nch; c += 4) {
      s0 += p[(int64_t)c * ks_sc];
      s1 += p[(int64_t)(c + 1) * ks_sc];
      s2 += p[(int64_t)(c + 2) * ks_sc];
      s3 += p[(int64_t)(c + 3) * ks_sc];
    }
    for (; c < nch; ++c) s0 += p[(int64_t)c * ks_sc];
    const float s = (s0 + s1) + (s2 + s3);
    if constexpr (ROUND) ((uint16_t*)ks_out)[td_head_off(h, hg, gs, 128) + threadIdx.x] = (uint16_t)f32_to_half_bits<DT>(s);
    else ((float*)ks_out)[td_head_off(h, hg, gs, 128) + threadIdx.x] = s;
  }
}

static int sla_linear_kv_partial_impl(const void* k, int dtype, const void* vt, int vt_dtype, float* ws_kv,
                                      float* ws_ks, float* ws_km, int64_t L, int64_t L_alloc, int hg, int64_t gs_bytes,
                                      int H, int D, td_stream_t stream, int fm = 0) {
  TD_REQUIRE(fm >= 0 && fm <= 2, TD_ERR_INVALID, "td_sla_linear_kv: feature map %d (0 softmax, 1 elu + 1, 2 relu)", fm);
  TD_REQUIRE(L_alloc >= L && hg >= 0 && gs_bytes >= 0 && gs_bytes % 16 == 0 && (hg == 0 || H % hg == 0), TD_ERR_INVALID,
             "td_sla_linear_kv_partial: packed layout L_alloc=%lld hg=%d gs=%lld", (long long)L_alloc, hg, (long long)gs_bytes);
  TD_REQUIRE(k && vt && ws_kv && ws_ks, TD_ERR_INVALID, "td_sla_linear_kv_partial: null pointer");
  TD_REQUIRE(D == 128, TD_ERR_UNSUPPORTED, "td_sla_linear_kv_partial: D=%d (need 128)", D);
  TD_REQUIRE(L > 0 && H > 0, TD_ERR_INVALID, "td_sla_linear_kv_partial: L=%lld H=%d", (long long)L, H);
  const int Kb = (int)td_cdiv(L, 64), Kb_alloc = (int)td_cdiv(L_alloc, 64);
  dim3 grid(LK_NCH, H);
  hipStream_t st = (hipStream_t)stream;
#define TD_LKP(KD_, VD_)                                                                                         \
  {                                                                                                               \
    if (fm == 1) linear_kv_partial_kernel<KD_, VD_, false, 1><<<grid, 256, 0, st>>>((const uint16_t*)k, (const uint16_t*)vt, ws_kv, ws_ks, nullptr, L, Kb, Kb_alloc, hg, gs_bytes / 2); \
    else if (fm == 2) linear_kv_partial_kernel<KD_, VD_, false, 2><<<grid, 256, 0, st>>>((const uint16_t*)k, (const uint16_t*)vt, ws_kv, ws_ks, nullptr, L, Kb, Kb_alloc, hg, gs_bytes / 2); \
    else if (ws_km) linear_kv_partial_kernel<KD_, VD_, true><<<grid, 256, 0, st>>>((const uint16_t*)k, (const uint16_t*)vt, ws_kv, ws_ks, ws_km, L, Kb, Kb_alloc, hg, gs_bytes / 2); \
    else linear_kv_partial_kernel<KD_, VD_, false><<<grid, 256, 0, st>>>((const uint16_t*)k, (const uint16_t*)vt, ws_kv, ws_ks, ws_km, L, Kb, Kb_alloc, hg, gs_bytes / 2);     \
  }
  TD_REQUIRE(fm == 0 || ws_km == nullptr, TD_ERR_UNSUPPORTED, "td_sla_linear_kv: the k mean rides on the softmax feature map's pass only");
  if (dtype == TD_BF16 && vt_dtype == TD_F16) TD_LKP(TD_BF16, TD_F16)
  else if (dtype == TD_BF16 && vt_dtype == TD_BF16) TD_LKP(TD_BF16, TD_BF16)
  else if (dtype == TD_F16 && vt_dtype == TD_F16) TD_LKP(TD_F16, TD_F16)
  else {
    td_set_error("td_sla_linear_kv_partial: unsupported dtypes k=%d vt=%d", dtype, vt_dtype);
    return TD_ERR_UNSUPPORTED;
  }
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_sla_linear_kv_partial(const void* k, int dtype, const void* vt, int vt_dtype,
                                        float* ws_kv, float* ws_ks, int64_t L, int H, int D,
                                        td_stream_t stream) {
  return sla_linear_kv_partial_impl(k, dtype, vt, vt_dtype, ws_kv, ws_ks, nullptr, L, L, 0, 0, H, D, stream);
}

// vt read from the sequence-parallel pack (L_alloc rows allocated per head, heads grouped: td_common.h td_head_off)
extern "C" int td_sla_linear_kv_partial_packed(const void* k, int dtype, const void* vt, int vt_dtype, float* ws_kv,
                                               float* ws_ks, int64_t L, int64_t L_alloc, int hg, int64_t gs_bytes, int H,
                                               int D, td_stream_t stream) {
  return sla_linear_kv_partial_impl(k, dtype, vt, vt_dtype, ws_kv, ws_ks, nullptr, L, L_alloc, hg, gs_bytes, H, D, stream);
}

extern "C" int td_sla_linear_kv_final_packed(const float* ws_kv, const float* ws_ks, int nch, int64_t kv_stride_h,
                                             int64_t kv_stride_c, int64_t ks_stride_h, int64_t ks_stride_c,
                                             void* kv_out, void* ks_out, int out_dtype, int hg, int64_t gs_bytes, int H,
                                             int D, td_stream_t stream) {
  TD_REQUIRE(ws_kv && ws_ks && kv_out && ks_out, TD_ERR_INVALID, "td_sla_linear_kv_final: null pointer");
  TD_REQUIRE(hg >= 0 && gs_bytes >= 0 && gs_bytes % 16 == 0 && (hg == 0 || H % hg == 0), TD_ERR_INVALID,
             "td_sla_linear_kv_final: packed layout hg=%d gs=%lld", hg, (long long)gs_bytes);
  TD_REQUIRE(D == 128 && nch > 0 && H > 0, TD_ERR_UNSUPPORTED, "td_sla_linear_kv_final: D=%d nch=%d", D, nch);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == TD_BF16)
    linear_kv_final_kernel<TD_BF16, true><<<dim3(H, 64), 256, 0, st>>>(ws_kv, ws_ks, nch, kv_stride_h, kv_stride_c, ks_stride_h, ks_stride_c, kv_out, ks_out, hg, gs_bytes / 2);
  else if (out_dtype == TD_F16)
    linear_kv_final_kernel<TD_F16, true><<<dim3(H, 64), 256, 0, st>>>(ws_kv, ws_ks, nch, kv_stride_h, kv_stride_c, ks_stride_h, ks_stride_c, kv_out, ks_out, hg, gs_bytes / 2);
  else if (out_dtype == TD_F32)
    linear_kv_final_kernel<TD_F32, false><<<dim3(H, 64), 256, 0, st>>>(ws_kv, ws_ks, nch, kv_stride_h, kv_stride_c, ks_stride_h, ks_stride_c, kv_out, ks_out, hg, gs_bytes / 4);
  else {
    td_set_error("td_sla_linear_kv_final: out dtype %d", out_dtype);
    return TD_ERR_UNSUPPORTED;
  }
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_sla_linear_kv_final(const float* ws_kv, const float* ws_ks, int nch, int64_t kv_stride_h,
                                      int64_t kv_stride_c, int64_t ks_stride_h, int64_t ks_stride_c,
                                      void* kv_out, void* ks_out, int out_dtype, int H, int D,
                                      td_stream_t stream) {
  return td_sla_linear_kv_final_packed(ws_kv, ws_ks, nch, kv_stride_h, kv_stride_c, ks_stride_h, ks_stride_c, kv_out, ks_out,
                                       out_dtype, 0, 0, H, D, stream);
}

extern "C" int td_sla_linear_kv(const void* k, int dtype, const void* vt, int vt_dtype, float* ws_kv,
                                float* ws_ks, void* kvsum_t, void* ksum, float* ws_km, void* km, int64_t L, int H,
                                int D, td_stream_t stream) {
  TD_REQUIRE(kvsum_t && ksum, TD_ERR_INVALID, "td_sla_linear_kv: null pointer");
  TD_REQUIRE((ws_km == nullptr) == (km == nullptr), TD_ERR_INVALID, "td_sla_linear_kv: ws_km / km must come together");
  int rc = sla_linear_kv_partial_impl(k, dtype, vt, vt_dtype, ws_kv, ws_ks, ws_km, L, L, 0, 0, H, D, stream);
  if (rc) return rc;
  rc = td_sla_linear_kv_final(ws_kv, ws_ks, LK_NCH, (int64_t)LK_NCH * 128 * 128, 128 * 128,
                              (int64_t)LK_NCH * 128, 128, kvsum_t, ksum, dtype, H, D, stream);
  if (rc || km == nullptr) return rc;
  return td_seq_mean_final(ws_km, LK_NCH, (int64_t)LK_NCH * 128, 128, km, dtype, L, H, D, stream);
}

// ---------------------------------------------------------------------------------------
// pass 2: o += proj_l( (cq @ kvsum) / (1e-5 + sum(cq*ksum)) )
// ---------------------------------------------------------------------------------------
#define LO_QB_PER_WG 4   // 768 workgroups at the C1 shape = 3 per CU: 67 us vs 75 (8) / 84 (6) / 104 (12), tools/lin_qb_exp.py
template <int DT, int FM = 0>
__global__ __launch_bounds__(256, 2) void linear_out_kernel(const uint16_t* __restrict__ q,
                                                         const uint16_t* __restrict__ kvT,
                                                         const uint16_t* __restrict__ ksum,
                                                         const float* __restrict__ wp,
                                                         const float* __restrict__ bp,
                                                         uint16_t* __restrict__ o, int64_t o_stride_h,
                                                         int64_t o_stride_l, int64_t L, int Qb,
                                                         uint16_t* __restrict__ t_out, int qb_per_wg,
                                                         unsigned long long* __restrict__ dbg) {
  // dbg != nullptr (TD_TUNE_LIN_QB < 0 selects it, tools/lin_qb_exp.py): s_memtime at the phase boundaries of every Q block
  // of wave 0 of workgroup (0, 0): {block start, softmax done, GEMM 1 done, o_l done, GEMM 2 + stores issued}
  int dbg_n = 0;
#define LO_STAMP()                                                                                         \
  if (dbg != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && dbg_n < 60) {           \
    dbg[dbg_n++] = __builtin_amdgcn_s_memtime();                                                           \
  }
  extern __shared__ __attribute__((aligned(16))) char smem_lo[];
  char* kvs = smem_lo;               // kvsum^T [d2][d1] DT, 256-B rows, swizzled
  char* wps = smem_lo + 128 * 256;   // Wp [d3][d2 in MFMA k order] DT
  // proj_l bias, autocast-rounded once, in LDS: read per d-block with ds_read (lgkmcnt).  As a global load inside the
  // store loop it brought an s_waitcnt vmcnt(0) per d-block, i.e. a wait for the PREVIOUS d-block's stores (gfx9 counts
  // stores in vmcnt): 8-13 k of a Q block's 13-18 k cycles (tools/lin_qb_exp.py phase stamps)
  float* bps = reinterpret_cast<float*>(smem_lo + 2 * 128 * 256);
  typedef typename MmaT<DT>::frag frag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y;

  // q of the first Q block: issued before the LDS staging so that its latency overlaps it; inside the loop the next
  // block's q is fetched while the current one is processed (a workgroup walks LO_QB_PER_WG blocks one after the other
  // and only ~6 waves fit a CU beside the 64 KB of LDS, so an un-prefetched load is a fully exposed HBM round trip)
  uint4 qraw[8];
  {
    int64_t tok0 = (int64_t)blockIdx.x * qb_per_wg * 128 + wave * 32 + li;
    if (tok0 > L - 1) tok0 = L - 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      qraw[ks] = *reinterpret_cast<const uint4*>(q + ((int64_t)h * L + tok0) * 128 + 16 * ks + 8 * hi);
  }
  for (int i = tid; i < 128 * 16; i += 256) {  // 16-B vectors of kvsum^T
    const int row = i >> 4, slot = i & 15;
    *reinterpret_cast<uint4*>(kvs + sw256(row, slot)) =
        *reinterpret_cast<const uint4*>(kvT + (int64_t)h * 128 * 128 + row * 128 + slot * 8);
  }
  for (int i = tid; i < 128 * 16; i += 256) {  // Wp fp32 -> DT (autocast), permuted k order, 16 B per item
    const int d3 = i >> 4, slot = i & 15;         // slot = 2*(16-group) + half: d2 = 16g + 4*half + {0-3, 8-11}
    const float* src = wp + d3 * 128 + (slot >> 1) * 16 + (slot & 1) * 4;
    const float4 lo = *reinterpret_cast<const float4*>(src);
    const float4 hi4 = *reinterpret_cast<const float4*>(src + 8);
    *reinterpret_cast<uint4*>(wps + sw256(d3, slot)) =
        make_uint4(pack2<DT>(lo.x, lo.y), pack2<DT>(lo.z, lo.w), pack2<DT>(hi4.x, hi4.y), pack2<DT>(hi4.z, hi4.w));
  }
  if (tid < 128) bps[tid] = round_half<DT>(bp[tid]);
  __syncthreads();

  // this lane's half of ksum: d1 = 16*ks + 8*hi + e
  float ksf[8][8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    unpack8<DT>(*reinterpret_cast<const uint4*>(ksum + h * 128 + 16 * ks + 8 * hi), ksf[ks]);

  for (int qq = 0; qq < qb_per_wg; ++qq) {
    const int qb = blockIdx.x * qb_per_wg + qq;
    if (qb >= Qb) break;
    int64_t tok = (int64_t)qb * 128 + wave * 32 + li;
    const bool ok = tok < L;
    if (!ok) tok = L - 1;
    LO_STAMP()
    // ---- cq = softmax_D(q) rounded; lane holds d1 = 16ks + 8hi + e ----
    float qf[8][8];
    float mx = -INFINITY;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      unpack8<DT>(qraw[ks], qf[ks]);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, qf[ks][e]);
    }
    if (qq + 1 < qb_per_wg && qb + 1 < Qb) {
      int64_t tokn = (int64_t)(qb + 1) * 128 + wave * 32 + li;
      if (tokn > L - 1) tokn = L - 1;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        qraw[ks] = *reinterpret_cast<const uint4*>(q + ((int64_t)h * L + tokn) * 128 + 16 * ks + 8 * hi);
    }
    float inv = 1.0f;
    if constexpr (FM == 0) {
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mb = mx * TD_LOG2E;
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          qf[ks][e] = __builtin_amdgcn_exp2f(fmaf(qf[ks][e], TD_LOG2E, -mb));
          sum += qf[ks][e];
        }
      sum += __shfl_xor(sum, 32, 64);
      inv = __builtin_amdgcn_rcpf(sum);
    } else {   // elementwise feature map: already the 16-bit value, the pack below is exact
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ks][e] = sla_feature<DT, FM>(qf[ks][e]);
    }
    float den = 0.f;
    uint4 cqf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        w[e >> 1] = pack2<DT>(qf[ks][e] * inv, qf[ks][e + 1] * inv);  // cq = softmax(...).to(dt)
        float c0, c1;
        unpack2<DT>(w[e >> 1], c0, c1);
        float p0, p1;
        unpack2<DT>(pack2<DT>(c0 * ksf[ks][e], c1 * ksf[ks][e + 1]), p0, p1);  // (q * ksum) in dt
        den += p0 + p1;
      }
      cqf[ks] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    den += __shfl_xor(den, 32, 64);
    den = round_half<DT>(den);            // .sum(-1) -> dt
    den = round_half<DT>(1e-5f + den);    // 1e-5 + ... -> dt
    const float rden = __builtin_amdgcn_rcpf(den);
    LO_STAMP()
    // ---- num^T[d2][tok] = kvsum^T . cq^T ----
    v16f a1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[c][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const frag af = *reinterpret_cast<const frag*>(kvs + sw256(32 * c + li, 2 * ks + hi));
        const frag bf = *reinterpret_cast<const frag*>(&cqf[ks]);
        a1[c] = MmaT<DT>::mma(af, bf, a1[c]);
      }
    }
    LO_STAMP()
    // ---- o_l = dt(dt(num) / den); its registers are the B fragments of the projection ----
    uint4 olf[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float n0, n1;
        unpack2<DT>(pack2<DT>(a1[c][r], a1[c][r + 1]), n0, n1);  // num in dt
        t[r] = n0 * rden;
        t[r + 1] = n1 * rden;
      }
      olf[2 * c] = pack8<DT>(&t[0]);
      olf[2 * c + 1] = pack8<DT>(&t[8]);
    }
    LO_STAMP()
    // ---- out^T[d3][tok] = Wp . o_l^T ----
    uint16_t* op = o + (int64_t)h * o_stride_h + tok * o_stride_l;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v16f a2;
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const frag af = *reinterpret_cast<const frag*>(wps + sw256(32 * c + li, 2 * ks + hi));
        const frag bf = *reinterpret_cast<const frag*>(&olf[ks]);
        a2 = MmaT<DT>::mma(af, bf, a2);
      }
      if (t_out != nullptr) {
        // lane-private layout for the attention kernel's epilogue (same lane = token, register = d mapping there):
        // [h][qb][wave][c*4+g4][lane] x 8 bytes — every store instruction writes 512 contiguous bytes, no read of o
        uint2* tp = reinterpret_cast<uint2*>(t_out) + (((int64_t)h * Qb + qb) * 4 + wave) * 16 * 64 + lane;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d3 = 32 * c + 8 * g4 + 4 * hi;
          const float4 bb = *reinterpret_cast<const float4*>(bps + d3);                            // autocast bias (LDS)
          const float bias[4] = {bb.x, bb.y, bb.z, bb.w};
          uint32_t res[2];
#pragma unroll
          for (int e = 0; e < 4; e += 2)
            res[e >> 1] = pack2<DT>(a2[4 * g4 + e] + bias[e], a2[4 * g4 + e + 1] + bias[e + 1]);   // o_l in dt
          tp[(c * 4 + g4) * 64] = make_uint2(res[0], res[1]);
        }
      } else if (ok) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d3 = 32 * c + 8 * g4 + 4 * hi;
          const float4 bb = *reinterpret_cast<const float4*>(bps + d3);                            // autocast bias (LDS)
          const float bias[4] = {bb.x, bb.y, bb.z, bb.w};
          const uint2 ov = *reinterpret_cast<const uint2*>(op + d3);
          const uint32_t ob[4] = {ov.x & 0xffffu, ov.x >> 16, ov.y & 0xffffu, ov.y >> 16};
          uint32_t res[2];
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            float l0, l1;
            unpack2<DT>(pack2<DT>(a2[4 * g4 + e] + bias[e], a2[4 * g4 + e + 1] + bias[e + 1]), l0, l1);   // o_l in dt
            res[e >> 1] = pack2<DT>(half_bits_to_f32<DT>(ob[e]) + l0, half_bits_to_f32<DT>(ob[e + 1]) + l1);
          }
          *reinterpret_cast<uint2*>(op + d3) = make_uint2(res[0], res[1]);
        }
      }
    }
    LO_STAMP()
  }
}

static int sla_linear_out_impl(const void* q, int dtype, const void* kvsum_t, const void* ksum,
                               const float* wp, const float* bp, void* o, int64_t o_stride_h,
                               int64_t o_stride_l, int64_t L, int H, int D, void* t_out, td_stream_t stream, int fm = 0) {
  TD_REQUIRE(fm >= 0 && fm <= 2, TD_ERR_INVALID, "td_sla_linear_out: feature map %d (0 softmax, 1 elu + 1, 2 relu)", fm);
  TD_REQUIRE(q && kvsum_t && ksum && wp && bp && (o || t_out), TD_ERR_INVALID, "td_sla_linear_out: null pointer");
  TD_REQUIRE(D == 128, TD_ERR_UNSUPPORTED, "td_sla_linear_out: D=%d (need 128)", D);
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_sla_linear_out: dtype %d", dtype);
  TD_REQUIRE(L > 0 && H > 0, TD_ERR_INVALID, "td_sla_linear_out: L=%lld H=%d", (long long)L, H);
  TD_REQUIRE(o_stride_l % 4 == 0 && o_stride_h % 4 == 0, TD_ERR_UNSUPPORTED, "td_sla_linear_out: strides");
  const int Qb = (int)td_cdiv(L, 128);
  const int lds = 2 * 128 * 256 + 128 * 4;
  const int tune_ = td_tuning(TD_TUNE_LIN_QB);
  // small problems (a sequence-parallel rank's heads of one group: 32 Q blocks x 3 heads): fewer Q blocks per workgroup so
  // that the grid still covers the chip (24 workgroups took 34 us for 1/32 of the single-GPU work, round 4)
  int qpw_auto = LO_QB_PER_WG;
  while (qpw_auto > 1 && td_cdiv(Qb, qpw_auto) * H < 512) --qpw_auto;
  const int qpw = tune_ > 0 ? tune_ : (tune_ < 0 ? -tune_ : qpw_auto);
  unsigned long long* dbg = tune_ < 0 ? td_dbg_buffer() : nullptr;
  dim3 grid((unsigned)td_cdiv(Qb, qpw), H);
  hipStream_t st = (hipStream_t)stream;
#define TD_LO(DT_, FM_)                                                                                        \
  {                                                                                                              \
    static std::atomic<uint64_t> a{0};                                                                           \
    td_ensure_dyn_lds(reinterpret_cast<const void*>(linear_out_kernel<DT_, FM_>), lds, a);                       \
    linear_out_kernel<DT_, FM_><<<grid, 256, lds, st>>>((const uint16_t*)q, (const uint16_t*)kvsum_t,            \
        (const uint16_t*)ksum, wp, bp, (uint16_t*)o, o_stride_h, o_stride_l, L, Qb, (uint16_t*)t_out, qpw, dbg); \
  }
  if (dtype == TD_BF16) { if (fm == 1) TD_LO(TD_BF16, 1) else if (fm == 2) TD_LO(TD_BF16, 2) else TD_LO(TD_BF16, 0) }
  else { if (fm == 1) TD_LO(TD_F16, 1) else if (fm == 2) TD_LO(TD_F16, 2) else TD_LO(TD_F16, 0) }
#undef TD_LO
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_sla_linear_out(const void* q, int dtype, const void* kvsum_t, const void* ksum,
                                 const float* wp, const float* bp, void* o, int64_t o_stride_h,
                                 int64_t o_stride_l, int64_t L, int H, int D, td_stream_t stream) {
  TD_REQUIRE(o, TD_ERR_INVALID, "td_sla_linear_out: null output");
  return sla_linear_out_impl(q, dtype, kvsum_t, ksum, wp, bp, o, o_stride_h, o_stride_l, L, H, D, nullptr, stream);
}

extern "C" int td_sla_linear_out_t(const void* q, int dtype, const void* kvsum_t, const void* ksum,
                                   const float* wp, const float* bp, void* t_out, int64_t L, int H, int D,
                                   td_stream_t stream) {
  TD_REQUIRE(t_out, TD_ERR_INVALID, "td_sla_linear_out_t: null output");
  return sla_linear_out_impl(q, dtype, kvsum_t, ksum, wp, bp, nullptr, 4, 4, L, H, D, t_out, stream);
}


// The linear branch with another feature map (SparseLinearAttention(feature_map = "elu" | "relu"), SLA/core.py:57-64):
// the two passes as td_sla_linear_kv / td_sla_linear_out with feature_map = 0 (softmax) | 1 (elu + 1) | 2 (relu).
extern "C" int td_sla_linear_kv_fm(const void* k, int dtype, const void* vt, int vt_dtype, float* ws_kv, float* ws_ks,
                                   void* kvsum_t, void* ksum, int feature_map, int64_t L, int H, int D, td_stream_t stream) {
  TD_REQUIRE(kvsum_t && ksum, TD_ERR_INVALID, "td_sla_linear_kv_fm: null pointer");
  int rc = sla_linear_kv_partial_impl(k, dtype, vt, vt_dtype, ws_kv, ws_ks, nullptr, L, L, 0, 0, H, D, stream, feature_map);
  if (rc) return rc;
  return td_sla_linear_kv_final(ws_kv, ws_ks, LK_NCH, (int64_t)LK_NCH * 128 * 128, 128 * 128, (int64_t)LK_NCH * 128, 128,
                                kvsum_t, ksum, dtype, H, D, stream);
}

extern "C" int td_sla_linear_out_fm(const void* q, int dtype, const void* kvsum_t, const void* ksum, const float* wp,
                                    const float* bp, void* o, int64_t o_stride_h, int64_t o_stride_l, int feature_map,
                                    int64_t L, int H, int D, td_stream_t stream) {
  TD_REQUIRE(o, TD_ERR_INVALID, "td_sla_linear_out_fm: null output");
  return sla_linear_out_impl(q, dtype, kvsum_t, ksum, wp, bp, o, o_stride_h, o_stride_l, L, H, D, nullptr, stream, feature_map);
}
