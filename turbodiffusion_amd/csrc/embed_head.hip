// f3 — the two ends of the DiT forward that are not transformer blocks:
//   td_patch_embed : patchify "b c (t kt) (h kh) (w kw) -> b (t h w) (c kt kh kw)" + patch_embedding Linear
//                    (rcm/networks/wan2pt1.py:653-661; wan2pt2.py:644-645 concatenates y on channels first) in ONE kernel:
//                    the token matrix [L, C*4] is never materialised, the channel concatenation is two source pointers.
//   td_head        : Head.forward (wan2pt1.py:444-454) + unpatchify (:710-721) in ONE kernel: eager LayerNorm (fp32 two-pass
//                    statistics, rounded to the activation dtype), fp32 modulate, fp32 Linear(dim -> out_dim*4), scattered
//                    straight into the [B, out_dim, T, 2H, 2W] video layout — the fp32 [L, dim] intermediate (200 MB at
//                    C1) is never written.
// Both are small next to the blocks (< 0.3 % of a step); they exist so that a captured forward contains no library kernel.
// Patch size (1, 2, 2) only (every Wan model, modify_model.py:86-127).
#include "td_common.h"
#include <algorithm>

// ------------------------------------------------------------------------------------------------------------------
// patch embedding: Y[l, n] = cast(sum_f X[l, f] W[n, f] + bias[n]), fp32 accumulate on v_mfma_f32_32x32x16_{bf16,f16},
// issued TRANSPOSED (A = weight rows, B = tokens) so that a lane owns one token and 4 consecutive n per register quad.
// Workgroup = 64 tokens x 256 outputs, 4 waves as 2 (token halves) x 2 (n halves).
// ------------------------------------------------------------------------------------------------------------------
template <int DT> struct PeMma;
template <> struct PeMma<TD_BF16> {
  typedef v8bf frag;
  __device__ static __forceinline__ v16f mma(frag a, frag b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct PeMma<TD_F16> {
  typedef v8h frag;
  __device__ static __forceinline__ v16f mma(frag a, frag b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <int DT>
__global__ __launch_bounds__(256) void patch_embed_kernel(const uint16_t* __restrict__ x, int c1, const uint16_t* __restrict__ x2,
                                                          int c2, int T, int Hin, int Win, const uint16_t* __restrict__ w,
                                                          const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
                                                          int dim, int64_t row0, int64_t rows, int tiles_per_batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int F = (c1 + c2) * 4, FS = F + 8;           // features per token; LDS row stride (elements): 16-byte aligned rows
  uint16_t* Wt = reinterpret_cast<uint16_t*>(smem);   // [256][FS]
  uint16_t* Xt = Wt + 256 * FS;                       // [64][FS]
  uint16_t* Ct = Wt;                                  // [64][264] (after the MFMAs)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / tiles_per_batch, tile = blockIdx.x % tiles_per_batch;
  const int n0 = blockIdx.y * 256;
  const int Hh = Hin >> 1, Ww = Win >> 1;
  // ---- gather the 64 x F token tile: thread -> one token, pairs (e = 0, 1 are adjacent in the source row) ----
  {
    const int tk = tid & 63;
    const int64_t lr = (int64_t)tile * 64 + tk;       // row of y within this batch entry
    const bool ok = lr < rows;
    const int64_t l = row0 + (ok ? lr : 0);
    const int t = (int)(l / ((int64_t)Hh * Ww)), hw = (int)(l % ((int64_t)Hh * Ww));
    const int h = hw / Ww, wq = hw % Ww;
    const int64_t plane = (int64_t)Hin * Win;
    for (int pp = tid >> 6; pp < F / 2; pp += 4) {
      const int c = pp >> 1, bs = pp & 1;
      const uint16_t* src = (c < c1) ? x + (((int64_t)b * c1 + c) * T + t) * plane
                                     : x2 + (((int64_t)b * c2 + (c - c1)) * T + t) * plane;
      uint32_t v = 0u;
      if (ok) v = *reinterpret_cast<const uint32_t*>(src + (int64_t)(2 * h + bs) * Win + 2 * wq);
      *reinterpret_cast<uint32_t*>(Xt + tk * FS + 2 * pp) = v;
    }
  }
  // ---- the 256 x F weight tile (rows past dim: zero) ----
  {
    const int cpr = F / 8;                             // 16-byte chunks per weight row
    for (int q = tid; q < 256 * cpr; q += 256) {
      const int row = q / cpr, cc = q % cpr;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (n0 + row < dim) v = *reinterpret_cast<const uint4*>(w + (int64_t)(n0 + row) * F + cc * 8);
      *reinterpret_cast<uint4*>(Wt + row * FS + cc * 8) = v;
    }
  }
  __syncthreads();
  typedef typename PeMma<DT>::frag frag;
  const int tg = wave & 1, nh = wave >> 1, li = lane & 31, hi = lane >> 5;
  v16f acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  for (int ks = 0; ks < F / 16; ++ks) {
    const frag xb = *reinterpret_cast<const frag*>(Xt + (tg * 32 + li) * FS + ks * 16 + hi * 8);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const frag wa = *reinterpret_cast<const frag*>(Wt + (nh * 128 + nb * 32 + li) * FS + ks * 16 + hi * 8);
      acc[nb] = PeMma<DT>::mma(wa, xb, acc[nb]);
    }
  }
  __syncthreads();                                     // every wave is done with Wt / Xt: Ct may overwrite Wt
  // ---- + bias, one rounding (the library GEMM's fp32 bias epilogue), through LDS to row-contiguous 16-byte stores ----
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int n = nh * 128 + nb * 32 + 8 * rq + 4 * hi;   // 4 consecutive outputs: registers 4rq .. 4rq+3
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (n0 + n < dim) {
        const uint2 bb = *reinterpret_cast<const uint2*>(bias + n0 + n);
        unpack2<DT>(bb.x, bv[0], bv[1]);
        unpack2<DT>(bb.y, bv[2], bv[3]);
      }
      const uint2 o = make_uint2(pack2<DT>(acc[nb][4 * rq] + bv[0], acc[nb][4 * rq + 1] + bv[1]),
                                 pack2<DT>(acc[nb][4 * rq + 2] + bv[2], acc[nb][4 * rq + 3] + bv[3]));
      *reinterpret_cast<uint2*>(Ct + (tg * 32 + li) * 264 + n) = o;
    }
  __syncthreads();
  for (int q = tid; q < 64 * 32; q += 256) {
    const int row = q >> 5, c16 = q & 31;
    const int64_t lr = (int64_t)tile * 64 + row;
    if (lr < rows && n0 + c16 * 8 < dim)
      *reinterpret_cast<uint4*>(y + ((int64_t)b * rows + lr) * dim + n0 + c16 * 8) = *reinterpret_cast<const uint4*>(Ct + row * 264 + c16 * 8);
  }
}

extern "C" int td_patch_embed(const void* x, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t B, int64_t T,
                              int64_t Hin, int64_t Win, const void* w, const void* bias, void* y, int64_t dim, int64_t row0,
                              int64_t rows, td_stream_t stream) {
  TD_REQUIRE(x && w && bias && y && (c2 == 0 || x2), TD_ERR_INVALID, "td_patch_embed: null pointer");
  TD_REQUIRE(dtype == TD_BF16 || dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_patch_embed: dtype %d (need f16|bf16)", dtype);
  TD_REQUIRE(c1 > 0 && c2 >= 0 && (c1 + c2) % 4 == 0 && c1 + c2 <= 64, TD_ERR_UNSUPPORTED,
             "td_patch_embed: channels %lld + %lld (need a multiple of 4, <= 64)", (long long)c1, (long long)c2);
  TD_REQUIRE(B > 0 && T > 0 && Hin > 0 && Win > 0 && Hin % 2 == 0 && Win % 2 == 0, TD_ERR_UNSUPPORTED,
             "td_patch_embed: latent %lld x %lld x %lld (patch (1, 2, 2): even H and W)", (long long)T, (long long)Hin, (long long)Win);
  TD_REQUIRE(dim > 0 && dim % 8 == 0, TD_ERR_UNSUPPORTED, "td_patch_embed: dim=%lld must be a multiple of 8", (long long)dim);
  const int64_t L = T * (Hin / 2) * (Win / 2);
  TD_REQUIRE(row0 >= 0 && rows >= 0 && row0 + rows <= L, TD_ERR_INVALID, "td_patch_embed: rows [%lld, %lld) of %lld tokens",
             (long long)row0, (long long)(row0 + rows), (long long)L);
  if (rows == 0) return TD_OK;
  const int F = (int)(c1 + c2) * 4, FS = F + 8;
  int lds = (256 + 64) * FS * 2;
  if (lds < 64 * 264 * 2) lds = 64 * 264 * 2;          // the output staging tile reuses the operand region
  const int tiles = (int)td_cdiv(rows, 64);
  dim3 grid((unsigned)(tiles * B), (unsigned)td_cdiv(dim, 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16) {
    static std::atomic<uint64_t> m{0};
    td_ensure_dyn_lds((const void*)patch_embed_kernel<TD_BF16>, lds, m);
    patch_embed_kernel<TD_BF16><<<grid, 256, lds, st>>>((const uint16_t*)x, (int)c1, (const uint16_t*)x2, (int)c2, (int)T, (int)Hin,
                                                      (int)Win, (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)y, (int)dim,
                                                      row0, rows, tiles);
  } else {
    static std::atomic<uint64_t> m{0};
    td_ensure_dyn_lds((const void*)patch_embed_kernel<TD_F16>, lds, m);
    patch_embed_kernel<TD_F16><<<grid, 256, lds, st>>>((const uint16_t*)x, (int)c1, (const uint16_t*)x2, (int)c2, (int)T, (int)Hin,
                                                     (int)Win, (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)y, (int)dim,
                                                     row0, rows, tiles);
  }
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// head: out[l, j] = sum_d hn[l, d] W[j, d] + bias[j] in fp32 (v_mfma_f32_32x32x2_f32, transposed: A = W rows, B = tokens),
// hn[l, d] = float(cast((x - mean) * rstd)) * (1 + scale[d]) + shift[d]  (wan2pt1.py:451-453).
// Workgroup = 64 tokens x P (<= 64) outputs; phase 1: row statistics (one wave per 16 rows, two passes over the row in
// registers, up to four rows in flight); phase 2: K chunks of 64 through LDS, 4 waves as 2 (token halves) x 2 (output halves).
// Measured 141 us at L = 32760, dim 1536 (profiles/NOTES_r03.md): 0.17 % of a video, the chunk size (128 or 64), the LDS
// read width and the rows in flight in phase 1 all left it within 3 % — the 512-workgroup grid is two per CU whatever the
// LDS allows, and each chunk is a load -> barrier -> MFMA -> barrier chain with nothing to overlap it.
// ------------------------------------------------------------------------------------------------------------------
#define HD_KC 64
#define HD_KS 68    // LDS row stride in words (16-byte aligned rows; 68 % 32 = 4: eight rows cover the 32 banks with 16 B each)
// Within a row of a chunk the k are stored EVEN k first, then ODD k (position of k = (k & 1) * HD_KC/2 + k / 2): the fp32 MFMA
// 32x32x2 gives lane (row, hi) the operand k = kk + hi of step kk, so the operands of four consecutive steps kk = 8m .. 8m+6
// are the 16 contiguous bytes at hi * HD_KC/2 + 4m — one ds_read_b128 per operand and four MFMAs instead of one ds_read_b32 each.

template <int DT, int NV>
__global__ __launch_bounds__(256) void head_kernel(const uint16_t* __restrict__ x, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float eps, float* __restrict__ out,
                                                   int unpatchify, int64_t rows, int dim, int P, int out_dim, int T, int Hh,
                                                   int Ww, int64_t row0, int tiles_per_batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* hn_s = reinterpret_cast<float*>(smem);           // [64][HD_KS]
  float* w_s = hn_s + 64 * HD_KS;                         // [64][HD_KS]
  float2* st_s = reinterpret_cast<float2*>(w_s + 64 * HD_KS);   // [64] (mean, rstd)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / tiles_per_batch, tile = blockIdx.x % tiles_per_batch;
  const int64_t lr0 = (int64_t)tile * 64;
  const uint16_t* xb = x + (int64_t)b * rows * dim;
  // ---- phase 1: LayerNorm statistics of 64 rows (the arithmetic of norm_rows_kernel<MODE 1>, norm.hip), R rows of a wave in
  //      flight together (a row is one 3-10 KB read: one at a time the wave only waits) ----
  constexpr int R = NV <= 4 ? 4 : (NV <= 10 ? 2 : 1);
  for (int rr = 0; rr < 16; rr += R) {
    float f[R][NV][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t lr = lr0 + wave * 16 + rr + r;
      if (lr >= rows) lr = rows - 1;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int col = (v * 64 + lane) * 8;
        if (col < dim) unpack8<DT>(*reinterpret_cast<const uint4*>(xb + lr * dim + col), f[r][v]);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[r][v][j] = 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float sum = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += f[r][v][j];
      const float mean = wave_sum(sum) / (float)dim;
      float sq = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int col = (v * 64 + lane) * 8;
        if (col < dim) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float d = f[r][v][j] - mean; sq += d * d; }
        }
      }
      const float var = wave_sum(sq) / (float)dim;
      if (lane == 0) st_s[wave * 16 + rr + r] = make_float2(mean, 1.0f / sqrtf(var + eps));
    }
  }
  __syncthreads();
  // ---- phase 2 ----
  const int tg = wave & 1, jh = wave >> 1, li = lane & 31, hi = lane >> 5;
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* sc = scale + (int64_t)b * dim;
  const float* sh = shift + (int64_t)b * dim;
  for (int k0 = 0; k0 < dim; k0 += HD_KC) {
    // hn chunk: 64 rows x 128 columns; thread -> (row = tid / 4 ... ) 16-byte pieces of 8 columns: 64 * 16 pieces
    for (int q = tid; q < 64 * (HD_KC / 8); q += 256) {
      const int row = q / (HD_KC / 8), c8 = q % (HD_KC / 8);
      int64_t lr = lr0 + row;
      if (lr >= rows) lr = rows - 1;
      const int col = k0 + c8 * 8;
      float hv[8];
      if (col < dim) {
        float xv[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(xb + lr * dim + col), xv);
        const float2 ms = st_s[row];
        const float4 s0 = *reinterpret_cast<const float4*>(sc + col), s1 = *reinterpret_cast<const float4*>(sc + col + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(sh + col), h1 = *reinterpret_cast<const float4*>(sh + col + 4);
        const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float tv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xn = round_half<DT>((xv[j] - ms.x) * ms.y);      // the norm's cast back to x.dtype (.type_as(x))
          const float t = xn * (1.0f + sv[j]);
          hv[j] = t + tv[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = 0.f;
      }
      *reinterpret_cast<float4*>(hn_s + row * HD_KS + c8 * 4) = make_float4(hv[0], hv[2], hv[4], hv[6]);        // even k
      *reinterpret_cast<float4*>(hn_s + row * HD_KS + HD_KC / 2 + c8 * 4) = make_float4(hv[1], hv[3], hv[5], hv[7]);   // odd k
    }
    // W chunk: P rows x 128 columns (rows >= P: zero)
    for (int q = tid; q < 64 * (HD_KC / 4); q += 256) {
      const int row = q / (HD_KC / 4), c4 = q % (HD_KC / 4);
      const int col = k0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < P && col < dim) v = *reinterpret_cast<const float4*>(w + (int64_t)row * dim + col);
      *reinterpret_cast<float2*>(w_s + row * HD_KS + c4 * 2) = make_float2(v.x, v.z);
      *reinterpret_cast<float2*>(w_s + row * HD_KS + HD_KC / 2 + c4 * 2) = make_float2(v.y, v.w);
    }
    __syncthreads();
#pragma unroll 4
    for (int m4 = 0; m4 < HD_KC / 8; ++m4) {
      const float4 a = *reinterpret_cast<const float4*>(w_s + (jh * 32 + li) * HD_KS + hi * (HD_KC / 2) + 4 * m4);
      const float4 bq = *reinterpret_cast<const float4*>(hn_s + (tg * 32 + li) * HD_KS + hi * (HD_KC / 2) + 4 * m4);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // ---- epilogue: lane = token tg*32 + li; register r -> output j = jh*32 + (r & 3) + 8 (r >> 2) + 4 hi ----
  const int64_t lr = lr0 + tg * 32 + li;
  if (lr >= rows) return;
  const int64_t l = row0 + lr;
  const int t = (int)(l / ((int64_t)Hh * Ww)), hw = (int)(l % ((int64_t)Hh * Ww));
  const int h = hw / Ww, wq = hw % Ww;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = jh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (j >= P) continue;
    const float v = acc[r] + bias[j];
    if (unpatchify) {   // j = (bi * 2 + ei) * out_dim + d  ->  [b, d, t, 2h + bi, 2w + ei]
      const int d = j % out_dim, be = j / out_dim, bi = be >> 1, ei = be & 1;
      out[((((int64_t)b * out_dim + d) * T + t) * (2 * Hh) + 2 * h + bi) * (2 * Ww) + 2 * wq + ei] = v;
    } else {
      out[((int64_t)b * rows + lr) * P + j] = v;
    }
  }
}

extern "C" int td_head(const void* x, int dtype, const float* scale, const float* shift, const float* w, const float* bias,
                       float eps, float* out, int unpatchify, int64_t B, int64_t rows, int64_t dim, int64_t out_dim, int64_t T,
                       int64_t Hh, int64_t Ww, int64_t row0, td_stream_t stream) {
  TD_REQUIRE(x && scale && shift && w && bias && out, TD_ERR_INVALID, "td_head: null pointer");
  TD_REQUIRE(dtype == TD_BF16 || dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_head: dtype %d (need f16|bf16)", dtype);
  TD_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= 8192, TD_ERR_UNSUPPORTED, "td_head: dim=%lld (need dim %% 8 == 0, <= 8192)", (long long)dim);
  TD_REQUIRE(out_dim > 0 && out_dim * 4 <= 64, TD_ERR_UNSUPPORTED, "td_head: out_dim=%lld (need out_dim * 4 <= 64)", (long long)out_dim);
  TD_REQUIRE(B > 0 && rows >= 0 && T > 0 && Hh > 0 && Ww > 0 && row0 >= 0 && row0 + rows <= T * Hh * Ww, TD_ERR_INVALID,
             "td_head: rows [%lld, %lld) of a %lld x %lld x %lld token grid", (long long)row0, (long long)(row0 + rows), (long long)T,
             (long long)Hh, (long long)Ww);
  TD_REQUIRE(!unpatchify || (row0 == 0 && rows == T * Hh * Ww), TD_ERR_INVALID, "td_head: unpatchify needs all tokens");
  if (rows == 0) return TD_OK;
  const int tiles = (int)td_cdiv(rows, 64);
  dim3 grid((unsigned)(tiles * B));
  hipStream_t st = (hipStream_t)stream;
  const int nv = (int)td_cdiv(dim, 512);
  const uint16_t* xp = (const uint16_t*)x;
  const int lds = 2 * 64 * HD_KS * 4 + 64 * 8;
#define TD_HEAD(DT_, NV_)                                                                                               \
  {                                                                                                                     \
    static std::atomic<uint64_t> m_{0};                                                                                 \
    td_ensure_dyn_lds((const void*)head_kernel<DT_, NV_>, lds, m_);                                                     \
    head_kernel<DT_, NV_><<<grid, 256, lds, st>>>(xp, scale, shift, w, bias, eps, out, unpatchify, rows, (int)dim,     \
                                                  (int)(out_dim * 4), (int)out_dim, (int)T, (int)Hh, (int)Ww, row0, tiles); \
  }
#define TD_HEAD_NV(DT_)                                                                                                 \
  {                                                                                                                     \
    if (nv <= 1) TD_HEAD(DT_, 1) else if (nv <= 2) TD_HEAD(DT_, 2) else if (nv <= 3) TD_HEAD(DT_, 3)                   \
    else if (nv <= 4) TD_HEAD(DT_, 4) else if (nv <= 6) TD_HEAD(DT_, 6) else if (nv <= 8) TD_HEAD(DT_, 8)              \
    else if (nv <= 10) TD_HEAD(DT_, 10) else TD_HEAD(DT_, 16)                                                           \
  }
  if (dtype == TD_BF16) TD_HEAD_NV(TD_BF16) else TD_HEAD_NV(TD_F16)
#undef TD_HEAD_NV
#undef TD_HEAD
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// time embedding (wan2pt1.py:144-153, 671-674): sinusoid of the (bf16-rounded) timestep in fp64, then three fp32 Linears
// on B rows — matrix-vector products: one wave per output, fp32 accumulate, 16-bit weights widened exactly (the
// reference's autocast(float32) island up-casts the bf16 parameters the same way).
// ------------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void time_sinusoid_kernel(const uint16_t* __restrict__ t, float* __restrict__ out, int B, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i % half;
  const double pos = (double)half_bits_to_f32<DT>(t[b]);
  // sinusoid = outer(position, pow(10000, -arange(half) / half))  (fp64, wan2pt1.py:148-151); cat([cos, sin], dim=1)
  const double ang = pos * pow(10000.0, -((double)j / (double)half));
  out[(int64_t)b * 2 * half + j] = (float)cos(ang);
  out[(int64_t)b * 2 * half + half + j] = (float)sin(ang);
}

extern "C" int td_time_sinusoid(const void* t, int dtype, float* out, int64_t B, int64_t freq_dim, td_stream_t stream) {
  TD_REQUIRE(t && out && B > 0 && freq_dim > 0 && freq_dim % 2 == 0, TD_ERR_INVALID, "td_time_sinusoid: bad argument");
  TD_REQUIRE(dtype == TD_BF16 || dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_time_sinusoid: dtype %d (need f16|bf16)", dtype);
  const int half = (int)(freq_dim / 2), n = (int)B * half;
  if (dtype == TD_BF16) time_sinusoid_kernel<TD_BF16><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>((const uint16_t*)t, out, (int)B, half);
  else time_sinusoid_kernel<TD_F16><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>((const uint16_t*)t, out, (int)B, half);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// out[b, n] = sum_k act(x[b, k]) * float(w[n, k]) + float(bias[n]);  act = identity | SiLU (nn.SiLU in fp32: x * sigmoid(x))
template <int DT, bool SILU>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ x, const uint16_t* __restrict__ w,
                                                        const uint16_t* __restrict__ bias, float* __restrict__ out, int B,
                                                        int64_t N, int K) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  for (int b = 0; b < B; ++b) {
    float acc = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
      float wv[8];
      unpack8<DT>(*reinterpret_cast<const uint4*>(w + n * K + k), wv);
      const float4 x0 = *reinterpret_cast<const float4*>(x + (int64_t)b * K + k), x1 = *reinterpret_cast<const float4*>(x + (int64_t)b * K + k + 4);
      float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (SILU) xv[j] = xv[j] / (1.0f + expf(-xv[j]));
        acc = fmaf(xv[j], wv[j], acc);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) out[(int64_t)b * N + n] = acc + half_bits_to_f32<DT>(bias[n]);
  }
}

extern "C" int td_gemv_f32(const float* x, const void* w, const void* bias, int dtype, int silu_input, float* out, int64_t B,
                           int64_t N, int64_t K, td_stream_t stream) {
  TD_REQUIRE(x && w && bias && out, TD_ERR_INVALID, "td_gemv_f32: null pointer");
  TD_REQUIRE(dtype == TD_BF16 || dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_gemv_f32: weight dtype %d (need f16|bf16)", dtype);
  TD_REQUIRE(B > 0 && B <= 64 && N > 0 && K > 0 && K % 8 == 0, TD_ERR_UNSUPPORTED, "td_gemv_f32: B=%lld N=%lld K=%lld (need B <= 64, K %% 8 == 0)",
             (long long)B, (long long)N, (long long)K);
  dim3 grid((unsigned)td_cdiv(N, 4));
  hipStream_t st = (hipStream_t)stream;
  const uint16_t* wp = (const uint16_t*)w;
  const uint16_t* bp = (const uint16_t*)bias;
  if (dtype == TD_BF16) {
    if (silu_input) gemv_rows_kernel<TD_BF16, true><<<grid, 256, 0, st>>>(x, wp, bp, out, (int)B, N, (int)K);
    else gemv_rows_kernel<TD_BF16, false><<<grid, 256, 0, st>>>(x, wp, bp, out, (int)B, N, (int)K);
  } else {
    if (silu_input) gemv_rows_kernel<TD_F16, true><<<grid, 256, 0, st>>>(x, wp, bp, out, (int)B, N, (int)K);
    else gemv_rows_kernel<TD_F16, false><<<grid, 256, 0, st>>>(x, wp, bp, out, (int)B, N, (int)K);
  }
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// out[a, b, r, d] = m[a, r, d] + e[b, r % re, d]: the AdaLN vectors of every block in one pass ((modulation + e0), wan2pt1.py:400:
// m = the blocks' modulation parameters [nblk, 6, dim], e = e0 [B, 6, dim], re = 6) and the head's ([1, 2, dim] + e [B, 1, dim], :452)
__global__ __launch_bounds__(256) void bcast_add_kernel(const float* __restrict__ m, const float* __restrict__ e, float* __restrict__ out,
                                                        int A, int B, int R, int RE, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)A * B * R * D;
  if (i >= total) return;
  const int d = (int)(i % D);
  const int r = (int)((i / D) % R);
  const int b = (int)((i / ((int64_t)D * R)) % B);
  const int a = (int)(i / ((int64_t)D * R * B));
  out[i] = m[((int64_t)a * R + r) * D + d] + e[((int64_t)b * RE + (r % RE)) * D + d];
}

extern "C" int td_bcast_add(const float* m, const float* e, float* out, int64_t A, int64_t B, int64_t R, int64_t RE, int64_t D,
                            td_stream_t stream) {
  TD_REQUIRE(m && e && out && A > 0 && B > 0 && R > 0 && D > 0 && (RE == R || RE == 1), TD_ERR_INVALID, "td_bcast_add: bad argument");
  const int64_t total = A * B * R * D;
  bcast_add_kernel<<<(unsigned)td_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(m, e, out, (int)A, (int)B, (int)R, (int)RE, (int)D);
  TD_CHECK_LAUNCH();
  return TD_OK;
}

// ---------------------------------------------------------------------------------------
// a18: one sampler update of the few-step rCM loop (inference/wan2.1_t2v_infer.py:134-139; ODE form wan2.2_i2v_infer.py:202-203)
// on the fp64 state, fused with the cast of the next step's network input:
//   SDE: x <- (1 - t_next) * (x - t_cur * double(v)) + double(float(t_next) * eps)   ODE (eps == NULL): x <- x - (t_cur - t_next) * double(v)
// the reference's operator sequence in fp64, operation for operation (no contraction: -ffp-contract=off), so the state is
// the bits torch's elementwise chain produces; x16 (optional) = x cast to the model's 16-bit dtype the way torch's
// `.to(dtype)` casts a double: through float (c10's BFloat16 / Half construct from float), i.e. two roundings.
// ---------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void rcm_step_kernel(double* __restrict__ x, const float* __restrict__ v, const float* __restrict__ eps,
                                                       uint16_t* __restrict__ x16, double t_cur, double t_next, int64_t n) {
  const double one_m = 1.0 - t_next, dt = t_cur - t_next;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double vv = (double)v[i];
    double xn;
    if (eps != nullptr) {
      const double a = x[i] - t_cur * vv;
      // `t_next * randn(dtype=float32)` is an fp32 product in the reference (a 0-dim fp64 tensor / Python float does not
      // promote an fp32 tensor): rounded to fp32 first, then added in fp64
      xn = one_m * a + (double)((float)t_next * eps[i]);
    } else {
      xn = x[i] - dt * vv;
    }
    x[i] = xn;
    if (x16 != nullptr) {
      x16[i] = (uint16_t)f32_to_half_bits<DT>((float)xn);
    }
  }
}

extern "C" int td_rcm_step(double* x, const float* v, const float* eps, void* x16, int dtype16, double t_cur, double t_next,
                           int64_t n, td_stream_t stream) {
  TD_REQUIRE(x && v && n > 0, TD_ERR_INVALID, "td_rcm_step: null pointer or empty state");
  TD_REQUIRE(!x16 || dtype16 == TD_BF16 || dtype16 == TD_F16, TD_ERR_UNSUPPORTED, "td_rcm_step: 16-bit dtype %d", dtype16);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)std::min<int64_t>(td_cdiv(n, 256), 2048);
  if (dtype16 == TD_F16) rcm_step_kernel<TD_F16><<<grid, 256, 0, st>>>(x, v, eps, (uint16_t*)x16, t_cur, t_next, n);
  else rcm_step_kernel<TD_BF16><<<grid, 256, 0, st>>>(x, v, eps, (uint16_t*)x16, t_cur, t_next, n);
  TD_CHECK_LAUNCH();
  return TD_OK;
}
