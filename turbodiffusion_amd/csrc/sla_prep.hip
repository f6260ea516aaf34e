// a11 / a13 preparation kernels for gfx950 (all HBM-bound, one pass over their input):
//   td_v_transpose     V -> per-64-key transposed tiles in MFMA operand order
//   td_seq_mean        per-head sequence mean of K (smooth-K)
//   td_sage_quant_pool block mean pool (+ smooth-K) and per-block INT8 quantisation
//   td_sla_topk        pooled score + top-k -> ascending LUT
// Reference: SLA/utils.py:21-67, SLA/core.py:197-204,213,221 (see include/turbodiffusion_amd.h).
#include "td_common.h"

// ---------------------------------------------------------------------------------------
// V transpose: tile [64 keys][128 d] -> [128 d][64 positions], position p of a 16-key group
// holds key (p&3) + 8*((p>>2)&1) + 4*(p>>3)  i.e. key order 0-3, 8-11, 4-7, 12-15.
// ---------------------------------------------------------------------------------------
template <int IDT, int ODT>
__global__ __launch_bounds__(256) void v_transpose_kernel(const uint16_t* __restrict__ v,
                                                          int64_t stride_h, int64_t stride_l,
                                                          uint16_t* __restrict__ vt, int64_t L, int Kb_alloc,
                                                          int hg, int64_t gs) {
  // LDS tile [64 keys][128 d + 2 pad] 16-bit: the pad makes the column reads below conflict-light
  __shared__ uint16_t tile[64][130];
  const int tid = threadIdx.x;
  const int kb = blockIdx.x, h = blockIdx.y;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vec = tid + 256 * i;          // 1024 vectors of 8 elements
    const int key = vec >> 4, c8 = vec & 15;
    const int64_t l = (int64_t)kb * 64 + key;
    uint4 raw = make_uint4(0, 0, 0, 0);     // tail keys -> zeros (P is 0 there, V must be finite)
    if (l < L) raw = *reinterpret_cast<const uint4*>(v + h * stride_h + l * stride_l + c8 * 8);
    float f[8];
    unpack8<IDT>(raw, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[key][c8 * 8 + j] = (uint16_t)f32_to_half_bits<ODT>(f[j]);
  }
  __syncthreads();
  uint16_t* out = vt + td_head_off(h, hg, gs, (int64_t)Kb_alloc * (128 * 64)) + (int64_t)kb * (128 * 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vec = tid + 256 * i;          // output vector: row d = vec/8, slot = vec%8
    const int d = vec >> 3, slot = vec & 7;
    const int ks = slot >> 1, hi = slot & 1;
    uint32_t w[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      uint32_t pr[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = 2 * e2 + q;
        const int key = 16 * ks + 8 * (e >> 2) + 4 * hi + (e & 3);
        pr[q] = tile[key][d];
      }
      w[e2] = pr[0] | (pr[1] << 16);
    }
    *reinterpret_cast<uint4*>(out + d * 64 + slot * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

extern "C" int td_v_transpose_packed(const void* v, int in_dtype, int64_t stride_h, int64_t stride_l, void* vt,
                                     int out_dtype, int64_t L, int64_t L_alloc, int hg, int64_t gs_bytes, int H, int D,
                                     td_stream_t stream) {
  TD_REQUIRE(v && vt, TD_ERR_INVALID, "td_v_transpose: null pointer");
  TD_REQUIRE(L_alloc >= L && hg >= 0 && gs_bytes >= 0 && gs_bytes % 16 == 0 && (hg == 0 || H % hg == 0), TD_ERR_INVALID,
             "td_v_transpose: packed layout L_alloc=%lld hg=%d gs=%lld", (long long)L_alloc, hg, (long long)gs_bytes);
  TD_REQUIRE(D == 128, TD_ERR_UNSUPPORTED, "td_v_transpose: D=%d (need 128)", D);
  TD_REQUIRE(L > 0 && H > 0, TD_ERR_INVALID, "td_v_transpose: L=%lld H=%d", (long long)L, H);
  TD_REQUIRE(stride_l % 8 == 0 && stride_h % 8 == 0, TD_ERR_UNSUPPORTED, "td_v_transpose: strides");
  const int Kb = (int)td_cdiv(L, 64), Kb_alloc = (int)td_cdiv(L_alloc, 64);
  dim3 grid(Kb, H);
  hipStream_t st = (hipStream_t)stream;
#define TD_VT(I_, O_)                                                                            \
  v_transpose_kernel<I_, O_><<<grid, 256, 0, st>>>((const uint16_t*)v, stride_h, stride_l,       \
                                                   (uint16_t*)vt, L, Kb_alloc, hg, gs_bytes / 2)
  if (in_dtype == TD_BF16 && out_dtype == TD_F16) TD_VT(TD_BF16, TD_F16);
  else if (in_dtype == TD_BF16 && out_dtype == TD_BF16) TD_VT(TD_BF16, TD_BF16);
  else if (in_dtype == TD_F16 && out_dtype == TD_F16) TD_VT(TD_F16, TD_F16);
  else {
    td_set_error("td_v_transpose: unsupported dtypes in=%d out=%d", in_dtype, out_dtype);
    return TD_ERR_UNSUPPORTED;
  }
#undef TD_VT
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_v_transpose(const void* v, int in_dtype, int64_t stride_h, int64_t stride_l, void* vt,
                              int out_dtype, int64_t L, int H, int D, td_stream_t stream) {
  return td_v_transpose_packed(v, in_dtype, stride_h, stride_l, vt, out_dtype, L, L, 0, 0, H, D, stream);
}

// ---------------------------------------------------------------------------------------
// sequence mean: two deterministic stages (no atomics): partial sums over 64 row chunks,
// then a finalize that sums the chunks in order, divides by L and rounds once.
// ---------------------------------------------------------------------------------------
#define SM_CHUNKS 64
template <int DT>
__global__ __launch_bounds__(256) void seq_mean_partial_kernel(const uint16_t* __restrict__ k,
                                                               float* __restrict__ ws, int64_t L) {
  __shared__ float red[16][128];
  const int tid = threadIdx.x, c8 = tid & 15, r0 = tid >> 4;
  const int chunk = blockIdx.x, h = blockIdx.y;
  const int64_t rows_per = td_cdiv(L, SM_CHUNKS);
  const int64_t lo = chunk * rows_per, hi_ = (lo + rows_per < L) ? lo + rows_per : L;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t l = lo + r0; l < hi_; l += 16) {
    float f[8];
    unpack8<DT>(*reinterpret_cast<const uint4*>(k + ((int64_t)h * L + l) * 128 + c8 * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[r0][c8 * 8 + j] = acc[j];
  __syncthreads();
  if (tid < 128) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][tid];
    ws[((int64_t)h * SM_CHUNKS + chunk) * 128 + tid] = s;
  }
}
// second stage: 8 slices of the partial list per head in parallel (1024 threads), each summed in order with 4
// independent loads in flight, then the slices in order: a fixed summation tree, so still deterministic
template <int DT>
__global__ __launch_bounds__(1024) void seq_mean_final_kernel(const float* __restrict__ ws, int nch,
                                                              int64_t stride_h, int64_t stride_c,
                                                              uint16_t* __restrict__ km, int64_t L) {
  __shared__ float part[8][128];
  const int h = blockIdx.x, d = threadIdx.x & 127, sl = threadIdx.x >> 7;
  const float* p = ws + h * stride_h + d;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = sl;
  for (; c + 24 < nch; c += 32) {
    s0 += p[(int64_t)c * stride_c];
    s1 += p[(int64_t)(c + 8) * stride_c];
    s2 += p[(int64_t)(c + 16) * stride_c];
    s3 += p[(int64_t)(c + 24) * stride_c];
  }
  for (; c < nch; c += 8) s0 += p[(int64_t)c * stride_c];
  part[sl][d] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += part[q][d];
    km[h * 128 + d] = (uint16_t)f32_to_half_bits<DT>(s / (float)L);
  }
}

// The per-head column sums of a rank's K rows as f32 [H, 128] (round 5: the sequence-parallel layer's smooth-K exchange used
// td_seq_sum_partial + a library reduction + a contiguous copy = three launches in front of a latency-bound all-gather): the
// partial kernel above and this one, which adds a head's SM_CHUNKS partials IN ORDER — a fixed summation tree, deterministic.
// (A one-launch form — the last workgroup of a head to finish, by an atomic ticket, adds the partials — was built first and
// measured 57 us at [12, 4096, 128] against 4.8 + 2.5 us for the pair: every workgroup's device-scope release / acquire fence
// writes back / invalidates its XCD's whole L2, dirty with the previous kernel's output.  No inter-workgroup hand-off inside a
// kernel on this part.)
__global__ __launch_bounds__(128) void seq_sum_final_kernel(const float* __restrict__ ws, float* __restrict__ out) {
  const int h = blockIdx.x, d = threadIdx.x;
  const float* p = ws + (int64_t)h * SM_CHUNKS * 128 + d;
  float v[SM_CHUNKS];
#pragma unroll
  for (int c = 0; c < SM_CHUNKS; ++c) v[c] = p[c * 128];     // all loads in flight together
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < SM_CHUNKS; ++c) s += v[c];
  out[h * 128 + d] = s;
}

template <int DT>
__global__ __launch_bounds__(256) void seq_mean_partial_kernel(const uint16_t* __restrict__ k, float* __restrict__ ws, int64_t L);

extern "C" int td_seq_sum(const void* k, float* ws, float* out, int dtype, int64_t L, int H, int D, td_stream_t stream) {
  TD_REQUIRE(k && ws && out, TD_ERR_INVALID, "td_seq_sum: null pointer");
  TD_REQUIRE(D == 128, TD_ERR_UNSUPPORTED, "td_seq_sum: D=%d (need 128)", D);
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_seq_sum: dtype %d", dtype);
  TD_REQUIRE(L > 0 && H > 0, TD_ERR_INVALID, "td_seq_sum: L=%lld H=%d", (long long)L, H);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(SM_CHUNKS, H);
  if (dtype == TD_BF16) seq_mean_partial_kernel<TD_BF16><<<grid, 256, 0, st>>>((const uint16_t*)k, ws, L);
  else seq_mean_partial_kernel<TD_F16><<<grid, 256, 0, st>>>((const uint16_t*)k, ws, L);
  seq_sum_final_kernel<<<H, 128, 0, st>>>(ws, out);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_seq_sum_partial(const void* k, float* ws, int dtype, int64_t L, int H, int D,
                                  td_stream_t stream) {
  TD_REQUIRE(k && ws, TD_ERR_INVALID, "td_seq_sum_partial: null pointer");
  TD_REQUIRE(D == 128, TD_ERR_UNSUPPORTED, "td_seq_sum_partial: D=%d (need 128)", D);
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_seq_sum_partial: dtype %d", dtype);
  TD_REQUIRE(L > 0 && H > 0, TD_ERR_INVALID, "td_seq_sum_partial: L=%lld H=%d", (long long)L, H);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(SM_CHUNKS, H);
  if (dtype == TD_BF16) seq_mean_partial_kernel<TD_BF16><<<grid, 256, 0, st>>>((const uint16_t*)k, ws, L);
  else seq_mean_partial_kernel<TD_F16><<<grid, 256, 0, st>>>((const uint16_t*)k, ws, L);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_seq_mean_final(const float* ws, int nch, int64_t stride_h, int64_t stride_c, void* km,
                                 int dtype, int64_t L_total, int H, int D, td_stream_t stream) {
  TD_REQUIRE(ws && km, TD_ERR_INVALID, "td_seq_mean_final: null pointer");
  TD_REQUIRE(D == 128, TD_ERR_UNSUPPORTED, "td_seq_mean_final: D=%d (need 128)", D);
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_seq_mean_final: dtype %d", dtype);
  TD_REQUIRE(nch > 0 && L_total > 0 && H > 0, TD_ERR_INVALID, "td_seq_mean_final: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16) seq_mean_final_kernel<TD_BF16><<<H, 1024, 0, st>>>(ws, nch, stride_h, stride_c, (uint16_t*)km, L_total);
  else seq_mean_final_kernel<TD_F16><<<H, 1024, 0, st>>>(ws, nch, stride_h, stride_c, (uint16_t*)km, L_total);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_seq_mean(const void* k, void* km, float* ws, int dtype, int64_t L, int H, int D,
                           td_stream_t stream) {
  int rc = td_seq_sum_partial(k, ws, dtype, L, H, D, stream);
  if (rc) return rc;
  return td_seq_mean_final(ws, SM_CHUNKS, (int64_t)SM_CHUNKS * 128, 128, km, dtype, L, H, D, stream);
}

// ---------------------------------------------------------------------------------------
// fused mean-pool (+smooth-K) and per-block INT8 quantisation over one [BLK x 128] block
// ---------------------------------------------------------------------------------------
template <int DT, int BLK>
__global__ __launch_bounds__(256) void sage_quant_pool_kernel(const uint16_t* __restrict__ x,
                                                              const uint16_t* __restrict__ km,
                                                              uint16_t* __restrict__ pooled,
                                                              int8_t* __restrict__ xq,
                                                              float* __restrict__ xs, int64_t L, int64_t L_alloc,
                                                              int nb, int hg, int64_t gs,
                                                              const float* __restrict__ km_parts = nullptr, int km_n = 0,
                                                              int64_t km_stride = 0, float km_rows = 1.f, int phg = -1,
                                                              int64_t pgs = 0) {
  if (phg < 0) { phg = hg; pgs = gs; }    // (phg, pgs): the head layout of `pooled` when it differs from that of xq / xs
  // nb = blocks ALLOCATED per head in pooled / xs (>= the blocks of L), L_alloc = rows allocated per head in xq; (hg, gs
  // in bytes): the packed head layout of td_common.h for the three outputs
  __shared__ float red[16][128];
  __shared__ float wmax[4];
  constexpr int NIT = BLK / 16;
  const int tid = threadIdx.x, c8 = tid & 15, r0 = tid >> 4;
  const int blk = blockIdx.x, h = blockIdx.y;
  float kmf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (km != nullptr) unpack8<DT>(*reinterpret_cast<const uint4*>(km + h * 128 + c8 * 8), kmf);
  if (km_parts != nullptr) {
    // the smooth-K mean formed HERE from the km_n per-rank column sums f32 [km_n][H*128] (the gathered output of td_seq_sum):
    // summed in rank order, divided by the global row count, rounded to the 16-bit dtype — td_seq_mean_final's arithmetic
    // for up to 8 partials (its 8 slices then hold one partial each, added in order), without its launch
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < km_n; ++r) {
      const float4 a = *reinterpret_cast<const float4*>(km_parts + r * km_stride + h * 128 + c8 * 8);
      const float4 b = *reinterpret_cast<const float4*>(km_parts + r * km_stride + h * 128 + c8 * 8 + 4);
      s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w; s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) kmf[j] = half_bits_to_f32<DT>(f32_to_half_bits<DT>(s[j] / km_rows));
  }

  uint4 raw[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int64_t l = (int64_t)blk * BLK + it * 16 + r0;
    raw[it] = make_uint4(0, 0, 0, 0);
    if (l < L) raw[it] = *reinterpret_cast<const uint4*>(x + ((int64_t)h * L + l) * 128 + c8 * 8);
  }
  // (VALU diet: this kernel was instruction-bound at ~35 VALU per element; hardware RNE packs, the quotient by the
  //  block scale as a reciprocal + one exact-remainder correction, v_med3 clamp and a magic-number byte extraction
  //  bring it to ~16 and the kernel back under the HBM time)
  float psum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float amax = 0.f;
  float xf[NIT][8];   // x - km in fp32: the quantiser's input, kept for the second phase
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int64_t l = (int64_t)blk * BLK + it * 16 + r0;
    const bool ok = l < L;
    const uint32_t wds[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float a0, a1;
      unpack2<DT>(wds[p], a0, a1);
      a0 = a0 - kmf[2 * p];
      a1 = a1 - kmf[2 * p + 1];
      xf[it][2 * p] = a0;
      xf[it][2 * p + 1] = a1;
      if (ok) {
        amax = fmaxf(amax, fmaxf(fabsf(a0), fabsf(a1)));
        float r0_, r1_;
        unpack2<DT>(pack2<DT>(a0, a1), r0_, r1_);   // dtype-rounded (pool input)
        psum[2 * p] += r0_;
        psum[2 * p + 1] += r1_;
      }
    }
  }
  // pooled mean
  if (pooled != nullptr) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[r0][c8 * 8 + j] = psum[j];
  }
  amax = wave_max(amax);
  if ((tid & 63) == 0) wmax[tid >> 6] = amax;
  __syncthreads();
  if (pooled != nullptr && tid < 128) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][tid];
    const int64_t rem = L - (int64_t)blk * BLK;
    const float cnt = (float)(rem < BLK ? rem : BLK);
    pooled[td_head_off(h, phg, pgs / 2, (int64_t)nb * 128) + (int64_t)blk * 128 + tid] = (uint16_t)f32_to_half_bits<DT>(s / cnt);
  }
  if (xq == nullptr) return;
  amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  const float scale = amax / 127.0f + 1e-7f;
  if (tid == 0) xs[td_head_off(h, hg, gs / 4, nb) + blk] = scale;
  // y = x / scale, correctly rounded (== the IEEE division of the oracle): rinv = RN(1/scale) by one Newton step on
  // v_rcp_f32, then q0 = RN(x*rinv), r = x - q0*scale exactly (fma), q = RN(q0 + r*rinv)  (Markstein's correction)
  float rinv = __builtin_amdgcn_rcpf(scale);
  rinv = fmaf(fmaf(-scale, rinv, 1.0f), rinv, rinv);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int64_t l = (int64_t)blk * BLK + it * 16 + r0;
    uint32_t c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x_ = xf[it][j];
      const float q0 = x_ * rinv;
      const float rr = fmaf(-q0, scale, x_);
      float y = fmaf(rr, rinv, q0);
      y = y + __builtin_copysignf(0.5f, y);        // round half away from zero: +-0.5 then truncate
      y = truncf(y);
      y = __builtin_amdgcn_fmed3f(y, -128.f, 127.f);
      c[j] = __float_as_uint(y + 12582912.0f);     // integer in [-128, 127]: the low byte of 1.5*2^23 + y is its code
    }
    const uint32_t w0 = __builtin_amdgcn_perm(c[1], c[0], 0x0c0c0400u) | __builtin_amdgcn_perm(c[3], c[2], 0x04000c0cu);
    const uint32_t w1 = __builtin_amdgcn_perm(c[5], c[4], 0x0c0c0400u) | __builtin_amdgcn_perm(c[7], c[6], 0x04000c0cu);
    if (l < L) *reinterpret_cast<uint2*>(xq + td_head_off(h, hg, gs, L_alloc * 128) + l * 128 + c8 * 8) = make_uint2(w0, w1);
  }
}

static int sage_quant_pool_impl(const void* x, const void* km, const float* km_parts, int km_n, int64_t km_stride, int64_t km_rows,
                                int dtype, int pool_blk, void* pooled, int8_t* xq, float* xs, int64_t L, int64_t L_alloc, int hg,
                                int64_t gs_bytes, int H, int D, td_stream_t stream, int pool_hg = -1, int64_t pool_gs_bytes = 0) {
  TD_REQUIRE(pool_hg < 0 || (pool_gs_bytes >= 0 && pool_gs_bytes % 16 == 0 && (pool_hg == 0 || H % pool_hg == 0)), TD_ERR_INVALID,
             "td_sage_quant_pool: pooled layout hg=%d gs=%lld", pool_hg, (long long)pool_gs_bytes);
  TD_REQUIRE(x, TD_ERR_INVALID, "td_sage_quant_pool: null input");
  TD_REQUIRE(km_parts == nullptr || (km == nullptr && km_n >= 1 && km_n <= 8 && km_stride >= (int64_t)H * 128 && km_stride % 4 == 0 && km_rows > 0),
             TD_ERR_INVALID, "td_sage_quant_pool: smooth-K partials n=%d stride=%lld rows=%lld (need 1..8 partials, no km beside them)",
             km_n, (long long)km_stride, (long long)km_rows);
  TD_REQUIRE(L_alloc >= L && hg >= 0 && gs_bytes >= 0 && gs_bytes % 16 == 0 && (hg == 0 || H % hg == 0), TD_ERR_INVALID,
             "td_sage_quant_pool: packed layout L_alloc=%lld hg=%d gs=%lld", (long long)L_alloc, hg, (long long)gs_bytes);
  TD_REQUIRE(D == 128, TD_ERR_UNSUPPORTED, "td_sage_quant_pool: D=%d (need 128)", D);
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_sage_quant_pool: dtype %d", dtype);
  TD_REQUIRE(pool_blk == 64 || pool_blk == 128, TD_ERR_UNSUPPORTED, "td_sage_quant_pool: blk %d", pool_blk);
  TD_REQUIRE((xq == nullptr) == (xs == nullptr), TD_ERR_INVALID, "td_sage_quant_pool: xq/xs mismatch");
  TD_REQUIRE(pooled || xq, TD_ERR_INVALID, "td_sage_quant_pool: nothing to do");
  TD_REQUIRE(L > 0 && H > 0, TD_ERR_INVALID, "td_sage_quant_pool: L=%lld H=%d", (long long)L, H);
  const int nb = (int)td_cdiv(L, pool_blk), nb_alloc = (int)td_cdiv(L_alloc, pool_blk);
  dim3 grid(nb, H);
  hipStream_t st = (hipStream_t)stream;
#define TD_SQP(DT_, B_)                                                                              \
  sage_quant_pool_kernel<DT_, B_><<<grid, 256, 0, st>>>((const uint16_t*)x, (const uint16_t*)km,      \
                                                        (uint16_t*)pooled, xq, xs, L, L_alloc, nb_alloc, hg, gs_bytes, \
                                                        km_parts, km_n, km_stride, (float)km_rows, pool_hg, pool_gs_bytes)
  if (dtype == TD_BF16) { if (pool_blk == 64) TD_SQP(TD_BF16, 64); else TD_SQP(TD_BF16, 128); }
  else { if (pool_blk == 64) TD_SQP(TD_F16, 64); else TD_SQP(TD_F16, 128); }
#undef TD_SQP
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_sage_quant_pool_packed(const void* x, const void* km, int dtype, int pool_blk, void* pooled,
                                         int8_t* xq, float* xs, int64_t L, int64_t L_alloc, int hg, int64_t gs_bytes,
                                         int H, int D, td_stream_t stream) {
  return sage_quant_pool_impl(x, km, nullptr, 0, 0, 0, dtype, pool_blk, pooled, xq, xs, L, L_alloc, hg, gs_bytes, H, D, stream);
}

// td_sage_quant_pool_packed with (a) the smooth-K mean either given (km) or formed in the kernel from km_n (<= 8) per-rank column
// sums km_parts f32 [km_n][>= H*128] (km_stride floats apart; the all-gathered td_seq_sum outputs) over km_rows global rows, and (b) a
// head layout of its own (pool_hg, pool_gs_bytes) for `pooled` (the sequence-parallel pack keeps pooled K of ALL heads in its first piece)
extern "C" int td_sage_quant_pool_packed_kmsum(const void* x, const void* km, const float* km_parts, int km_n, int64_t km_stride,
                                               int64_t km_rows, int dtype, int pool_blk, void* pooled, int pool_hg,
                                               int64_t pool_gs_bytes, int8_t* xq, float* xs, int64_t L, int64_t L_alloc, int hg,
                                               int64_t gs_bytes, int H, int D, td_stream_t stream) {
  TD_REQUIRE((km == nullptr) || (km_parts == nullptr), TD_ERR_INVALID, "td_sage_quant_pool_packed_kmsum: km AND partials");
  return sage_quant_pool_impl(x, km, km_parts, km_n, km_stride, km_rows, dtype, pool_blk, pooled, xq, xs, L, L_alloc, hg,
                              gs_bytes, H, D, stream, pool_hg, pool_gs_bytes);
}

extern "C" int td_sage_quant_pool(const void* x, const void* km, int dtype, int pool_blk, void* pooled,
                                  int8_t* xq, float* xs, int64_t L, int H, int D, td_stream_t stream) {
  return td_sage_quant_pool_packed(x, km, dtype, pool_blk, pooled, xq, xs, L, L, 0, 0, H, D, stream);
}

// ---------------------------------------------------------------------------------------
// pooled score + top-k.  One 256-thread workgroup per 4 pooled-Q rows of one head (768 workgroups at the Wan 480p
// shape):
//   phase 1: scores of the 4 rows against 64 pooled keys at a time on v_mfma_f32_16x16x32 (the 4 rows are rows 0-3 of
//            the 16-row A operand; wave w owns keys 16w..16w+15 of the chunk), fp32 accumulate, rounded to the 16-bit
//            dtype like the reference's bf16 matmul (SLA/utils.py:59).  (The first version accumulated the 128 products
//            per score with scalar FMAs: 176 VALU + 48 LDS reads per thread and chunk made the kernel instruction-bound
//            at 37 us for 1.5 M scores.)
//   phase 2: wave w selects the top-k of row w: binary search on the order-preserving 16-bit key for the k-th
//            largest value, then an ordered ballot compaction (ties at the threshold -> lowest index first) =>
//            ascending LUT.
// ---------------------------------------------------------------------------------------
#define TK_ROWS 4
#define TK_MAXKB 2048
template <int DT> struct TkMma;
template <> struct TkMma<TD_BF16> {
  typedef v8bf frag;
  __device__ static __forceinline__ v4f mma(frag a, frag b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct TkMma<TD_F16> {
  typedef v8h frag;
  __device__ static __forceinline__ v4f mma(frag a, frag b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

template <int DT, int NPER>   // NPER >= ceil(Kb / 64): the wave's row lives in NPER registers per lane during the selection
__global__ __launch_bounds__(256) void sla_topk_kernel(const uint16_t* __restrict__ pq,
                                                       const uint16_t* __restrict__ pk,
                                                       int32_t* __restrict__ lut, int Qb, int Kb,
                                                       int Kb_alloc, int topk, int kbp, int64_t pk_rs) {
  // kbp > 0: pooled K in the rank-major layout of the sequence-parallel all-gather — block j is block j % kbp of rank
  // j / kbp, rows at pk + rank*pk_rs + (h*kbp + ...)*128 (Kb_alloc = kbp then)
#define TK_PKROW(key_) ((kbp > 0) ? ((int64_t)((key_) / kbp) * pk_rs + ((int64_t)h * kbp + ((key_) % kbp)) * 128) \
                                  : (((int64_t)h * Kb_alloc + (key_)) * 128))
  extern __shared__ __attribute__((aligned(16))) char smem_tk[];
  uint16_t* sc = reinterpret_cast<uint16_t*>(smem_tk);                      // [4][Kb] sortable keys
  uint4* ktile = reinterpret_cast<uint4*>(smem_tk + (((size_t)TK_ROWS * Kb * 2 + 15) & ~(size_t)15));
  typedef typename TkMma<DT>::frag frag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lq = lane >> 4;
  const int h = blockIdx.y, row0 = blockIdx.x * TK_ROWS;
  // A fragments: pooled-Q row l16 (rows 4..15 of the operand are zero), d = 32*ks + 8*lq .. +7
  uint4 qa[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qa[ks] = make_uint4(0, 0, 0, 0);
    if (l16 < TK_ROWS) {
      const int qr = row0 + l16 < Qb ? row0 + l16 : Qb - 1;
      qa[ks] = *reinterpret_cast<const uint4*>(pq + ((int64_t)h * Qb + qr) * 128 + 32 * ks + 8 * lq);
    }
  }
  // scores, 64 keys at a time: the 16 KB of pooled keys are fetched with fully coalesced 16-byte loads into LDS
  // (rows padded to 272 B: lane = key reads are conflict-free); the next chunk is requested before this one is used
  uint4 nxt[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + 256 * it, key = idx >> 4, v = idx & 15;
    nxt[it] = make_uint4(0, 0, 0, 0);
    if (key < Kb) nxt[it] = *reinterpret_cast<const uint4*>(pk + TK_PKROW(key) + v * 8);
  }
  for (int c0 = 0; c0 < Kb; c0 += 64) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + 256 * it, key = idx >> 4, v = idx & 15;
      ktile[key * 17 + v] = nxt[it];
    }
    if (c0 + 64 < Kb) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + 256 * it, key = idx >> 4, v = idx & 15;
        nxt[it] = make_uint4(0, 0, 0, 0);
        if (c0 + 64 + key < Kb) nxt[it] = *reinterpret_cast<const uint4*>(pk + TK_PKROW(c0 + 64 + key) + v * 8);
      }
    }
    __syncthreads();
    {
      v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 kb_ = ktile[(16 * wave + l16) * 17 + 4 * ks + lq];   // B: key 16*wave + l16, d = 32*ks + 8*lq .. +7
        acc = TkMma<DT>::mma(*reinterpret_cast<const frag*>(&qa[ks]), *reinterpret_cast<const frag*>(&kb_), acc);
      }
      // D[i][j]: lane holds key j = l16 and rows i = 4*lq + r; the 4 real rows are r = 0..3 of the lanes with lq == 0
      const int key = c0 + 16 * wave + l16;
      if (lq == 0 && key < Kb) {
#pragma unroll
        for (int r = 0; r < TK_ROWS; ++r) {
          uint32_t b = f32_to_half_bits<DT>(acc[r]);
          // order-preserving map of a 16-bit float to an unsigned key (larger value -> larger key)
          b = (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u);
          sc[r * Kb + key] = (uint16_t)b;
        }
      }
    }
    __syncthreads();
  }
  // selection (wave = row): every count is a sum of ballot popcounts — v_cmp + s_bcnt1, no cross-lane shuffles
  const int r = wave;
  if (row0 + r >= Qb) return;
  const uint16_t* keys = sc + r * Kb;
  // largest T with count(key >= T) >= topk
  // the row's Kb sortable keys: key t*64 + lane in register t (0 = below every real key, never selected); every count
  // of the search is then NPER compares + scalar popcounts, no memory access
  uint32_t kv[NPER];
#pragma unroll
  for (int t = 0; t < NPER; ++t) {
    const int j = t * 64 + lane;
    kv[t] = j < Kb ? keys[j] : 0u;
  }
  uint32_t lo = 1, hi = 0xffffu;  // invariant: count(>= lo) >= topk   (real keys are >= 1)
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < NPER; ++t) cnt += __popcll(__ballot(kv[t] >= mid));
    if (cnt >= topk) lo = mid; else hi = mid - 1;
  }
  const uint32_t T = lo;
  int gt = 0;
#pragma unroll
  for (int t = 0; t < NPER; ++t) gt += __popcll(__ballot(kv[t] > T));
  const int need_eq = topk - gt;  // how many ties at T to take (lowest index first)
  int32_t* out = lut + ((int64_t)h * Qb + row0 + r) * topk;
  int written = 0, eq_seen = 0;
#pragma unroll
  for (int t = 0; t < NPER; ++t) {
    const int j = t * 64 + lane;
    const bool is_gt = kv[t] > T;
    const bool is_eq = kv[t] == T;
    const unsigned long long meq = __ballot(is_eq);
    const int eq_before = __popcll(meq & ((1ull << lane) - 1ull));
    const bool take = is_gt || (is_eq && (eq_seen + eq_before) < need_eq);
    const unsigned long long mt = __ballot(take);
    if (take) out[written + __popcll(mt & ((1ull << lane) - 1ull))] = j;
    written += __popcll(mt);
    eq_seen += __popcll(meq);
  }
}

static int sla_topk_impl(const void* pq, const void* pk, int dtype, int32_t* lut, int H, int Qb,
                         int Kb, int Kb_alloc, int D, int topk, td_stream_t stream, int kbp, int64_t pk_rs) {
  if (Kb_alloc == 0) Kb_alloc = Kb;
  TD_REQUIRE(kbp > 0 || Kb_alloc >= Kb, TD_ERR_INVALID, "td_sla_topk: Kb_alloc=%d < Kb=%d", Kb_alloc, Kb);
  TD_REQUIRE(pq && pk && lut, TD_ERR_INVALID, "td_sla_topk: null pointer");
  TD_REQUIRE(D == 128, TD_ERR_UNSUPPORTED, "td_sla_topk: D=%d (need 128)", D);
  TD_REQUIRE(dtype == TD_F16 || dtype == TD_BF16, TD_ERR_UNSUPPORTED, "td_sla_topk: dtype %d", dtype);
  TD_REQUIRE(Kb >= 1 && Kb <= TK_MAXKB, TD_ERR_UNSUPPORTED, "td_sla_topk: Kb=%d (max %d)", Kb, TK_MAXKB);
  TD_REQUIRE(topk >= 1 && topk <= Kb, TD_ERR_INVALID, "td_sla_topk: topk=%d Kb=%d", topk, Kb);
  TD_REQUIRE(H > 0 && Qb > 0, TD_ERR_INVALID, "td_sla_topk: H=%d Qb=%d", H, Qb);
  const size_t lds = (((size_t)TK_ROWS * Kb * 2 + 15) & ~(size_t)15) + 64 * 17 * 16;
  dim3 grid((unsigned)td_cdiv(Qb, TK_ROWS), H);
  hipStream_t st = (hipStream_t)stream;
  const int nper = (Kb + 63) / 64;
#define TD_TK(DT_, NP_)                                                                                            \
  {                                                                                                                \
    static std::atomic<uint64_t> a{0};                                                                             \
    td_ensure_dyn_lds(reinterpret_cast<const void*>(sla_topk_kernel<DT_, NP_>), TK_ROWS * TK_MAXKB * 2 + 16 + 64 * 17 * 16, a); \
    sla_topk_kernel<DT_, NP_><<<grid, 256, lds, st>>>((const uint16_t*)pq, (const uint16_t*)pk, lut, Qb, Kb, Kb_alloc, topk, kbp, pk_rs); \
  }
#define TD_TK_DT(DT_)                                                                                              \
  {                                                                                                                \
    if (nper <= 8) TD_TK(DT_, 8) else if (nper <= 16) TD_TK(DT_, 16) else if (nper <= 24) TD_TK(DT_, 24) else TD_TK(DT_, 32) \
  }
  if (dtype == TD_BF16) TD_TK_DT(TD_BF16) else TD_TK_DT(TD_F16)
#undef TD_TK_DT
#undef TD_TK
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_sla_topk(const void* pq, const void* pk, int dtype, int32_t* lut, int H, int Qb,
                           int Kb, int Kb_alloc, int D, int topk, td_stream_t stream) {
  return sla_topk_impl(pq, pk, dtype, lut, H, Qb, Kb, Kb_alloc, D, topk, stream, 0, 0);
}

// sequence parallelism: pooled K straight from the all-gather's rank-major output — block j = block j % kb_per_rank of
// rank j / kb_per_rank, its row at pk + rank*pk_rank_stride + (h*kb_per_rank + ...)*D (strides in elements)
extern "C" int td_sla_topk_sp(const void* pq, const void* pk, int dtype, int32_t* lut, int H, int Qb, int Kb,
                              int kb_per_rank, int64_t pk_rank_stride, int D, int topk, td_stream_t stream) {
  TD_REQUIRE(kb_per_rank > 0 && pk_rank_stride > 0 && pk_rank_stride % 8 == 0, TD_ERR_INVALID,
             "td_sla_topk_sp: kb_per_rank=%d pk_rank_stride=%lld", kb_per_rank, (long long)pk_rank_stride);
  return sla_topk_impl(pq, pk, dtype, lut, H, Qb, Kb, kb_per_rank, D, topk, stream, kb_per_rank, pk_rank_stride);
}
