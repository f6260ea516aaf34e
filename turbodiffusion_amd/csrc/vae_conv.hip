// f4 — the convolutions of the Wan2.1 VAE decoder as ONE implicit-GEMM kernel on the bf16 matrix pipe,
// plus its channel RMS-norm (+ SiLU) pass.  Reference semantics: rcm/tokenizers/wan2pt1.py — CausalConv3d (:37-55: kt - 1
// zero frames on the LEFT of the time axis, symmetric spatial padding), Resample (:94-131: nearest x2 spatial up-sampling
// followed by a 3x3 convolution; time up-sampling = a (3,1,1) causal convolution to 2C channels whose two halves are
// interleaved in time), ResidualBlock (:195-209: conv(...) + shortcut), RMS_norm (:69-70).
//
// MI355X design.  Activations live channels-last, [B, T, H, W, C] bf16, so a convolution is the GEMM
//     Y[m, n] = sum_{tap, c} X[pos(m) + tap, c] * W[n, tap, c] + bias[n]
// with M = B*T*H*W output positions (tens of millions for a 480p clip), N = C_out, K = taps * C_in; the whole clip is one
// launch (no frame chunks, no cache of boundary frames: the clip's activations fit HBM many times over).  Workgroup tile
// 256 positions x 32*NB channels (NB = 3: C_out 96 / 192 / 384 / 768 are multiples of 96; NB = 1 for the 3-channel head),
// four waves of 64 positions each: per 16-wide K step a wave reads 2 A and NB B fragments (ds_read_b128) for 2*NB
// v_mfma_f32_32x32x16_bf16.  K advances in chunks of 64 = two 32-channel halves, each half with its own tap (C_in = 96 is
// three halves per tap), gathered straight from the input tensor — zero for taps that fall before the first frame or
// outside the image, and with the x2 up-sampling folded into the gather (source pixel = (h >> 1, w >> 1)) so the 4x larger
// up-sampled tensor is never written.  Global loads for chunk j + 1 are in flight while chunk j is multiplied (register
// prefetch, one LDS stage: 44 KB, two workgroups per CU); 128-byte LDS rows with the 16-byte slot XOR-swizzled by the row
// (the int8 K tile's scheme in attn.hip) keep both the row-per-thread writes and the 32-row fragment reads conflict-free.
// Epilogue: + bias (fp32) -> bf16 -> optional residual add in bf16 (the reference's rounding points) -> channels-last store;
// for the time up-sampler the output channel n lands in frame 2 t + n / (N/2), channel n % (N/2).
#include "td_common.h"
#include "vae_conv.h"
#include <type_traits>


#define VC_BM 256
#define VC_ROWB 128       // bytes per LDS row (64 bf16)
__device__ __forceinline__ uint32_t vc_off(uint32_t row, uint32_t slot) {
  return row * 128u + ((slot ^ ((row >> 1) & 7u)) << 4);
}

template <int NB>
__global__ __launch_bounds__(256, 2) void vae_conv_kernel(VaeConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_s = smem;                       // [256][128 B]
  char* b_s = smem + VC_BM * VC_ROWB;     // [32 NB][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int tiles_n = (p.Co + 32 * NB - 1) / (32 * NB);
  const int64_t tile_m = blockIdx.x / tiles_n;
  const int n0 = (int)(blockIdx.x % tiles_n) * 32 * NB;

  // ---- this thread's A row: output position m = tile_m * 256 + tid -> (b, t, h, w) ----
  int64_t m = tile_m * VC_BM + tid;
  if (m >= p.M) m = p.M - 1;
  int wo = (int)(m % p.Wo);
  int64_t r1 = m / p.Wo;
  int ho = (int)(r1 % p.Ho);
  r1 /= p.Ho;
  const int to = (int)(r1 % p.To), bb = (int)(r1 / p.To);
  const uint16_t* xb = p.x + (int64_t)bb * p.xs_b;
  const int hsrc = p.up2 ? 2 * p.Hi : p.Hi, wsrc = p.up2 ? 2 * p.Wi : p.Wi;   // extent of the (up-sampled) source image
  const int cpt = p.Ci >> 5;              // 32-channel halves per tap
  const int64_t ktot = (int64_t)p.halves * 32;

  // ---- B rows this thread loads: vector v = tid + 256 e -> (row v / 8, slot v % 8) ----
  constexpr int BV = NB == 3 ? 3 : 1;     // 96 x 8 = 768 vectors, or 32 x 8 = 256
  const uint16_t* bptr[BV];
#pragma unroll
  for (int e = 0; e < BV; ++e) {
    const int v = tid + 256 * e, row = v >> 3;
    int n = n0 + row;
    if (n >= p.Co) n = p.Co - 1;
    bptr[e] = p.w + (int64_t)n * ktot + (v & 7) * 8;
  }

  uint4 pa[8], pb[BV];
  // uniform (scalar) position of the next half to fetch: tap = (dt, dh, dw), channel offset cq * 32
  int f_dt = 0, f_dh = 0, f_dw = 0, f_cq = 0, f_q = 0;
  auto prefetch = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bool ok = f_q < p.halves;
      const int ts = to * p.st - p.pt + f_dt, hu = ho * p.ss - p.ph + f_dh, wu = wo * p.ss - p.pw + f_dw;
      ok = ok && ts >= 0 && ts < p.Ti && hu >= 0 && hu < hsrc && wu >= 0 && wu < wsrc;
      const int hs = p.up2 ? (hu >> 1) : hu, ws = p.up2 ? (wu >> 1) : wu;
      const uint16_t* src = xb + (((int64_t)ts * p.Hi + hs) * p.Wi + ws) * p.Ci + f_cq * 32;
#pragma unroll
      for (int v = 0; v < 4; ++v) pa[4 * s + v] = ok ? *reinterpret_cast<const uint4*>(src + 8 * v) : make_uint4(0u, 0u, 0u, 0u);
      // advance (uniform)
      ++f_q;
      if (++f_cq == cpt) {
        f_cq = 0;
        if (++f_dw == p.kw) { f_dw = 0; if (++f_dh == p.kh) { f_dh = 0; ++f_dt; } }
      }
    }
  };
  int64_t f_k = 0;   // first K index of the chunk being fetched
  auto prefetch_b = [&]() {
#pragma unroll
    for (int e = 0; e < BV; ++e) {
      const int v = tid + 256 * e;
      const int64_t kk = f_k + (v & 7) * 8;
      pb[e] = kk < ktot ? *reinterpret_cast<const uint4*>(bptr[e] + f_k) : make_uint4(0u, 0u, 0u, 0u);
    }
    f_k += 64;
  };

  v16f acc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][nb][r] = 0.f;

  const int chunks = (p.halves + 1) >> 1;
  prefetch();
  prefetch_b();
  for (int j = 0; j < chunks; ++j) {
    __syncthreads();   // chunk j - 1 has been multiplied by every wave
#pragma unroll
    for (int v = 0; v < 8; ++v) *reinterpret_cast<uint4*>(a_s + vc_off(tid, v)) = pa[v];
#pragma unroll
    for (int e = 0; e < BV; ++e) {
      const int v = tid + 256 * e;
      *reinterpret_cast<uint4*>(b_s + vc_off(v >> 3, v & 7)) = pb[e];
    }
    __syncthreads();
    if (j + 1 < chunks) {
      prefetch();
      prefetch_b();
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v8bf af[2], bf[NB];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const v8bf*>(a_s + vc_off(64 * wave + 32 * i + li, 2 * ks + hi));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bf[nb] = *reinterpret_cast<const v8bf*>(b_s + vc_off(32 * nb + li, 2 * ks + hi));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[nb], acc[i][nb], 0, 0, 0);
    }
  }

  // ---- epilogue: lane = output channel n0 + 32 nb + li; registers = positions (r & 3) + 8 (r >> 2) + 4 hi of the 32-row block ----
  float bias_v[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + 32 * nb + li;
    bias_v[nb] = (p.bias && n < p.Co) ? bf16_bits_to_f32(p.bias[n]) : 0.f;
  }
  const int half_c = p.Co >> 1;
  const int64_t per_b = (int64_t)p.To * p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t mm = tile_m * VC_BM + 64 * wave + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
      // element offset of (position mm, channel 0) and the channel stride pattern of the destination
      int64_t o0;
      if (p.interleave) {
        const int w_ = (int)(mm % p.Wo);
        int64_t q = mm / p.Wo;
        const int h_ = (int)(q % p.Ho);
        q /= p.Ho;
        const int t_ = (int)(q % p.To), b_ = (int)(q / p.To);
        o0 = (int64_t)b_ * p.ys_b + ((((int64_t)(2 * t_)) * p.Ho + h_) * p.Wo + w_) * half_c;
      } else {
        const int64_t b_ = mm / per_b;
        o0 = b_ * p.ys_b + (mm - b_ * per_b) * p.Co;
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + 32 * nb + li;
        if (mm < p.M && n < p.Co) {
          int64_t o = o0 + n;
          if (p.interleave && n >= half_c) o = o0 + (int64_t)p.Ho * p.Wo * half_c + (n - half_c);   // second half: the next frame
          float v = round_bf16(acc[i][nb][r] + bias_v[nb]);
          if (p.res) v = round_bf16(v + bf16_bits_to_f32(p.res[o]));
          p.y[o] = (uint16_t)f32_to_bf16_bits(v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// v2 (default): one workgroup = up to 256 consecutive columns of ONE image row (b, t, h) x 32*NB output channels.
// Measured on v1 above: the 3-channel head convolution cost as much as a 96-channel one (25 vs 27 ms at 480p) — the kernel
// was bound by the A gather (every input element fetched once per tap: 27 x), not by the matrix pipe.  Here the K loop
// walks (dt, dh, 64 channels): the source row segment [w0 - 1, w0 + 256] x 64 channels is brought to LDS ONCE (8 lanes per
// 128-byte row: coalesced) and serves all three dw taps — the A fragment of tap dw is the same LDS tile read one row
// further down — so the gather traffic and the LDS writes drop 3 x and there is one barrier pair per 3 x 64 K values
// (72 MFMAs per wave) instead of per 64.  Rows before the first frame / outside the image skip their iteration
// altogether.  Tiles never cross an image row, so nothing needs an integer division and the epilogue goes through LDS:
// every thread then owns one position's 32*NB contiguous channels — 16-byte global stores, 16-byte residual loads.
// ------------------------------------------------------------------------------------------------------------------
// WR = 32-row blocks per wave: 2 -> 256-column tiles, two workgroups per CU (the default); 4 -> 512-column tiles, wave tile
// 128 x 96 (7 fragment reads per 12 MFMAs instead of 5 per 6; 192 accumulators in AGPRs), ONE workgroup per CU with the
// 512-register file of a single wave per SIMD — built to test whether LDS bandwidth is what bounds the kernel: it is not, the
// 96-channel level runs at 549 instead of 700 TFLOP/s (one wave per SIMD has nothing to run beside its own LDS writes and
// barriers).  Kept selectable (TD_TUNE_VAE_CONV = 3) and tested.
typedef unsigned int vc_u4 __attribute__((ext_vector_type(4)));
// KC = channels per K chunk: 64 -> 128-byte LDS rows, ONE stage, two barriers per chunk (write phase, then multiply); 32 -> 64-byte
// rows (slot swizzle by (row >> 2) & 3), TWO stages in the same 70 KB: chunk j + 1 is written to the other stage at the top of
// iteration j (its global loads were issued an iteration earlier) while everyone multiplies chunk j — one barrier per chunk
// and no wave waits for another's LDS writes before it may start its MFMAs.
template <int KC> __device__ __forceinline__ uint32_t vc2_off(uint32_t row, uint32_t slot) {
  if constexpr (KC == 64) return row * 128u + ((slot ^ ((row >> 1) & 7u)) << 4);
  else return row * 64u + ((slot ^ ((row >> 2) & 3u)) << 4);
}
// KW > 0 (with KC = 32): the tap count along w as a compile-time constant — the whole multiply of a chunk (KW x 2 k-steps) is ONE
// basic block, so the fragment reads of a k-step can be scheduled over the MFMAs of the one before (with run-time bounds every
// k-step is its own block: five reads, a wait, six MFMAs, nothing overlapping within the wave).
// (amdgpu_waves_per_eu: LDS allows two workgroups per CU whatever the registers; without the pin the compiler spills the
// prefetch registers to scratch to reach the 168 VGPRs of three waves per SIMD.)
template <int NB, int WR, int KC = 64, int KW = 0>
__global__ __launch_bounds__(256, WR == 4 ? 1 : 2) __attribute__((amdgpu_waves_per_eu(WR == 4 ? 1 : 2, WR == 4 ? 1 : 2)))
void vae_conv2_kernel(VaeConvP p) {
  constexpr int BM = 64 * WR * 2;                // columns per tile: 256 | 512
  constexpr int AROWS = BM + 8;
  constexpr int ROWB = KC * 2, SLOTS = ROWB / 16, RPP = 256 / SLOTS;   // bytes per LDS row, 16-byte slots per row, rows per loader pass
  constexpr int NR = 32 * NB;
  constexpr int STAGE = (AROWS + 3 * NR) * ROWB;
  constexpr bool TWO = KC == 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int tiles_n = (p.Co + NR - 1) / NR, tiles_w = (p.Wo + BM - 1) / BM;
  uint32_t bid = blockIdx.x;
  const int n0 = (int)(bid % tiles_n) * NR;
  bid /= tiles_n;
  const int w0 = (int)(bid % tiles_w) * BM;
  bid /= tiles_w;
  int h, t, b;
  if (p.t_fast) {   // frames of one image row first: the (dt) neighbours of a tile are being read by the workgroups beside it
    t = (int)(bid % p.To);
    bid /= p.To;
    h = (int)(bid % p.Ho);
    b = (int)(bid / p.Ho);
  } else {
    h = (int)(bid % p.Ho);
    bid /= p.Ho;
    t = (int)(bid % p.To);
    b = (int)(bid / p.To);
  }
  const int ph = p.kh >> 1, pw = p.kw >> 1;
  const int64_t ktot = (int64_t)p.halves * 32;
  const uint16_t* xb = p.x + (int64_t)b * p.xs_b;

  // ---- loader roles: vector v = tid + 256 e -> (row tid / SLOTS + RPP e, 16-byte slot tid % SLOTS) ----
  const int slot = tid % SLOTS, r8 = tid / SLOTS;
  constexpr int AV = (BM + 2 + RPP - 1) / RPP;   // the last pass: rows BM.. (the right halo)
  constexpr int BVT = (3 * NR + RPP - 1) / RPP;  // B row passes for kw = 3
  vc_u4 pa[AV], pb[BVT];   // (a first-class vector type: arrays of the HIP uint4 STRUCT filled by plain loads stay in scratch)
  // Loop-invariant 32-bit byte offsets from a UNIFORM base pointer (the source row's start, the weights' (dt, dh, c0) start):
  // a load is one instruction with no address arithmetic and no branch.  Columns outside the image are never loaded into LDS
  // (their rows are zeroed once, below, and their stores skipped), so those lanes just read the row's first vector.
  uint32_t a_off[AV];
  uint32_t a_mask = 0u;                          // bit e: row r8 + RPP e is a column of the image
#pragma unroll
  for (int e = 0; e < AV; ++e) {
    const int r = r8 + RPP * e, wu = w0 - pw + r;
    const bool ok = r < BM + 2 * pw && wu >= 0 && wu < p.Wo;
    a_mask |= ok ? (1u << e) : 0u;
    const int col = ok ? (p.up2 ? (wu >> 1) : wu) : 0;
    a_off[e] = (uint32_t)(col * p.Ci) * 2u;
  }
  const int nbr = p.kw * NR;                     // B rows in use
  uint32_t b_off[BVT];                           // byte offset of (row n, tap dw) within the weights, without (dt, dh, c0)
#pragma unroll
  for (int e = 0; e < BVT; ++e) {
    int ridx = r8 + RPP * e;
    if (ridx >= nbr) ridx = nbr - 1;             // rows past the taps in use are never read: any valid address will do
    const int dw = ridx / NR;
    int n = n0 + (ridx - dw * NR);
    if (n >= p.Co) n = p.Co - 1;
    b_off[e] = (uint32_t)((int64_t)n * ktot + (int64_t)dw * p.Ci) * 2u;
  }
  // zero the A rows of both stages once: out-of-image columns (and the unused tail rows) stay zero for the whole kernel
  {
    constexpr int NST = KC == 32 ? 2 : 1;
    constexpr int STG = (AROWS + 3 * (32 * NB)) * ROWB;
    for (int st_ = 0; st_ < NST; ++st_)
      for (int v = tid; v < AROWS * SLOTS; v += 256)
        *reinterpret_cast<uint4*>(smem + st_ * STG + v * 16) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
  }

  v16f acc[WR][NB];
#pragma unroll
  for (int i = 0; i < WR; ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][nb][r] = 0.f;

  // ---- the list of iterations (dt, dh, c0) whose source row exists ----
  const int nc = (p.Ci + KC - 1) / KC;
  int it_dt = 0, it_dh = 0, it_c = 0;            // position of the NEXT iteration to fetch
  auto row_ok = [&](int dt, int dh) {
    const int ts = t - (p.kt - 1) + dt, hu = h + dh - ph;
    return ts >= 0 && hu >= 0 && hu < p.Ho;
  };
  auto advance_to_valid = [&]() {                // skip (dt, dh) pairs that fall outside; returns false at the end
    while (it_dt < p.kt && !row_ok(it_dt, it_dh)) {
      it_c = 0;
      if (++it_dh == p.kh) { it_dh = 0; ++it_dt; }
    }
    return it_dt < p.kt;
  };
  int f_kc = 0;                                  // channels of the fetched iteration (KC or the tail)
  auto fetch = [&]() {
    const int ts = t - (p.kt - 1) + it_dt, hu = h + it_dh - ph;
    const int hs = p.up2 ? (hu >> 1) : hu;
    const int c0 = it_c * KC;
    f_kc = min(KC, p.Ci - c0);
    // slots past the chunk's channels (the 32-channel tail of C_in = 96) are never read by the multiply: they load slot 0
    const uint32_t soff = (uint32_t)((slot * 8 < f_kc ? slot : 0) * 16);
    const char* srow = reinterpret_cast<const char*>(xb + (((int64_t)ts * p.Hi + hs) * p.Wi) * p.Ci + c0);
#pragma unroll
    for (int e = 0; e < AV; ++e) pa[e] = *reinterpret_cast<const vc_u4*>(srow + (a_off[e] + soff));
    const char* wrow = reinterpret_cast<const char*>(p.w + (int64_t)((it_dt * p.kh + it_dh) * p.kw) * p.Ci + c0);
#pragma unroll
    for (int e = 0; e < BVT; ++e) pb[e] = *reinterpret_cast<const vc_u4*>(wrow + (b_off[e] + soff));
    if (++it_c == nc) {
      it_c = 0;
      if (++it_dh == p.kh) { it_dh = 0; ++it_dt; }
    }
  };
  auto store = [&](char* st) {                   // the fetched chunk -> LDS stage st
    char* a_w = st;
    char* b_w = st + AROWS * ROWB;
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int r = r8 + RPP * e;
      if ((a_mask >> e) & 1u) *reinterpret_cast<vc_u4*>(a_w + vc2_off<KC>(r, slot)) = pa[e];
    }
#pragma unroll
    for (int e = 0; e < BVT; ++e)
      if (r8 + RPP * e < nbr) *reinterpret_cast<vc_u4*>(b_w + vc2_off<KC>(r8 + RPP * e, slot)) = pb[e];
  };
  // a wave whose columns all lie past the end of the image row (the last tile of a row: 832 = 3 x 256 + 64) only helps load
  const bool wave_live = w0 + 32 * WR * wave < p.Wo;
  auto multiply = [&](const char* st, int kc) {
    const char* a_r = st;
    const char* b_r = st + AROWS * ROWB;
    const int nks = kc >> 4;
    if constexpr (KW > 0) {
      auto steps = [&](auto nks_c) {
        constexpr int NKS = decltype(nks_c)::value;
#pragma unroll
        for (int dw = 0; dw < KW; ++dw)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            v8bf af[WR], bf[NB];
#pragma unroll
            for (int i = 0; i < WR; ++i) af[i] = *reinterpret_cast<const v8bf*>(a_r + vc2_off<KC>(32 * WR * wave + 32 * i + li + dw, 2 * ks + hi));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[nb] = *reinterpret_cast<const v8bf*>(b_r + vc2_off<KC>(dw * NR + 32 * nb + li, 2 * ks + hi));
#pragma unroll
            for (int i = 0; i < WR; ++i)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[nb], acc[i][nb], 0, 0, 0);
          }
      };
      if (wave_live) {
        if (KC == 32 || nks == KC / 16) steps(std::integral_constant<int, KC / 16>{});
        else steps(std::integral_constant<int, 2>{});      // the 32-channel tail chunk of C_in = 96
      }
    } else if (wave_live)
      for (int dw = 0; dw < p.kw; ++dw) {
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
          if (ks < nks) {
            v8bf af[WR], bf[NB];
#pragma unroll
            for (int i = 0; i < WR; ++i) af[i] = *reinterpret_cast<const v8bf*>(a_r + vc2_off<KC>(32 * WR * wave + 32 * i + li + dw, 2 * ks + hi));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[nb] = *reinterpret_cast<const v8bf*>(b_r + vc2_off<KC>(dw * NR + 32 * nb + li, 2 * ks + hi));
#pragma unroll
            for (int i = 0; i < WR; ++i)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[nb], acc[i][nb], 0, 0, 0);
          }
        }
      }
  };

  if constexpr (!TWO) {
    bool have = advance_to_valid();
    if (have) fetch();
    while (have) {
      const int kc = f_kc;
      __syncthreads();   // the previous iteration has been multiplied by every wave
      store(smem);
      __syncthreads();
      have = advance_to_valid();
      if (have) fetch();
      multiply(smem, kc);
    }
  } else {
    bool in_regs = advance_to_valid();            // (always true: the centre tap's row exists)
    fetch();
    store(smem);
    in_regs = advance_to_valid();
    if (in_regs) fetch();
    __syncthreads();
    int cur = 0;
    while (true) {
      const bool has_next = in_regs;
      if (has_next) store(smem + (cur ^ 1) * STAGE);   // its last readers passed the barrier that ended the previous iteration
      in_regs = has_next && advance_to_valid();
      if (in_regs) fetch();
      multiply(smem + cur * STAGE, KC);
      __syncthreads();   // stage cur may be overwritten; the stores to the other stage are visible
      if (!has_next) break;
      cur ^= 1;
    }
  }

  // ---- epilogue through LDS, 256 positions per pass: O[256][32 NB channels] bf16, row stride OS bytes ----
  constexpr int OS = NR * 2 + 16;
  float bias_v[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + 32 * nb + li;
    bias_v[nb] = (p.bias && n < p.Co) ? bf16_bits_to_f32(p.bias[n]) : 0.f;
  }
  const int half_c = p.Co >> 1;
#pragma unroll
  for (int q = 0; q < BM / 256; ++q) {
    __syncthreads();   // the K loop's (or the previous pass's) LDS contents are no longer needed
    if ((32 * WR * wave) / 256 == q) {
      const int rbase = 32 * WR * wave - 256 * q;
#pragma unroll
      for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
            *reinterpret_cast<uint16_t*>(smem + row * OS + (32 * nb + li) * 2) = (uint16_t)f32_to_bf16_bits(acc[i][nb][r] + bias_v[nb]);
          }
    }
    __syncthreads();
    const int w = w0 + 256 * q + tid;
    if (w < p.Wo) {
      int64_t o;
      if (p.interleave) {
        const int kk = n0 >= half_c ? 1 : 0;
        o = (int64_t)b * p.ys_b + ((((int64_t)(2 * t + kk)) * p.Ho + h) * p.Wo + w) * half_c + (n0 - kk * half_c);
      } else {
        o = (int64_t)b * p.ys_b + (((int64_t)t * p.Ho + h) * p.Wo + w) * p.Co + n0;
      }
      const char* orow = smem + tid * OS;
      if (n0 + NR <= p.Co) {
#pragma unroll
        for (int v = 0; v < NR / 8; ++v) {
          uint4 ov = *reinterpret_cast<const uint4*>(orow + 16 * v);
          if (p.res) {
            const uint4 rv = *reinterpret_cast<const uint4*>(p.res + o + 8 * v);
            float a[8], c[8];
            unpack8<TD_BF16>(ov, a);
            unpack8<TD_BF16>(rv, c);
            uint32_t qq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) qq[j] = f32_to_bf16_bits(a[j] + c[j]);
            ov.x = qq[0] | (qq[1] << 16); ov.y = qq[2] | (qq[3] << 16); ov.z = qq[4] | (qq[5] << 16); ov.w = qq[6] | (qq[7] << 16);
          }
          *reinterpret_cast<uint4*>(p.y + o + 8 * v) = ov;
        }
      } else {   // a partial channel tile (the 3-channel head): element stores
        for (int c = 0; n0 + c < p.Co; ++c) {
          float v = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(orow + 2 * c));
          if (p.res) v = round_bf16(v + bf16_bits_to_f32(p.res[o + c]));
          p.y[o + c] = (uint16_t)f32_to_bf16_bits(v);
        }
      }
    }
  }
}

extern "C" int td_vae_conv_ex(const void* x, int64_t x_batch_stride, const void* w, const void* bias, const void* res, void* y,
                              int64_t y_batch_stride, int B, int Ti, int Hi, int Wi, int Ci, int Co, int kt, int kh, int kw,
                              int up2, int interleave, int To, int Ho, int Wo, int stride_t, int stride_hw, int pad_t,
                              int pad_h, int pad_w, td_stream_t stream) {
  TD_REQUIRE(x && w && y, TD_ERR_INVALID, "td_vae_conv: null pointer");
  TD_REQUIRE(B > 0 && Ti > 0 && Hi > 0 && Wi > 0 && Co > 0, TD_ERR_INVALID, "td_vae_conv: empty problem");
  TD_REQUIRE(Ci > 0 && Ci % 32 == 0, TD_ERR_UNSUPPORTED, "td_vae_conv: C_in = %d must be a multiple of 32 (pad with zero channels)", Ci);
  TD_REQUIRE(kt >= 1 && kt <= 3 && (kh == 1 || kh == 3) && (kw == 1 || kw == 3), TD_ERR_UNSUPPORTED, "td_vae_conv: kernel %dx%dx%d", kt, kh, kw);
  TD_REQUIRE(!interleave || (Co % 2 == 0 && !res), TD_ERR_INVALID, "td_vae_conv: the time up-sampler needs an even C_out and takes no residual");
  VaeConvP p;
  p.x = (const uint16_t*)x; p.w = (const uint16_t*)w; p.bias = (const uint16_t*)bias; p.res = (const uint16_t*)res;
  p.y = (uint16_t*)y; p.xs_b = x_batch_stride; p.ys_b = y_batch_stride;
  p.B = B; p.Ti = Ti; p.Hi = Hi; p.Wi = Wi; p.Ci = Ci;
  TD_REQUIRE(To > 0 && Ho > 0 && Wo > 0 && (stride_t == 1 || stride_t == 2) && (stride_hw == 1 || stride_hw == 2) && pad_t >= 0 &&
             pad_h >= 0 && pad_w >= 0, TD_ERR_INVALID, "td_vae_conv: output grid %dx%dx%d strides %d/%d", To, Ho, Wo, stride_t, stride_hw);
  const bool plain = stride_t == 1 && stride_hw == 1 && pad_t == kt - 1 && pad_h == kh / 2 && pad_w == kw / 2 && To == Ti &&
                     Ho == (up2 ? 2 * Hi : Hi) && Wo == (up2 ? 2 * Wi : Wi);
  TD_REQUIRE(plain || !interleave, TD_ERR_UNSUPPORTED, "td_vae_conv: the time up-sampler mapping needs the plain geometry");
  p.To = To; p.Ho = Ho; p.Wo = Wo; p.Co = Co;
  p.st = stride_t; p.ss = stride_hw; p.pt = pad_t; p.ph = pad_h; p.pw = pad_w;
  p.t_fast = td_tuning(TD_TUNE_VAE_CONV) == 5 ? 1 : 0;
  p.kt = kt; p.kh = kh; p.kw = kw; p.up2 = up2; p.interleave = interleave;
  p.M = (int64_t)B * p.To * p.Ho * p.Wo;
  p.halves = kt * kh * kw * (Ci / 32);
  const int nbw = Co <= 32 ? 1 : 3;
  const int64_t tiles = td_cdiv(p.M, VC_BM) * td_cdiv(Co, 32 * nbw);
  TD_REQUIRE(tiles < (1ll << 31), TD_ERR_UNSUPPORTED, "td_vae_conv: %lld tiles", (long long)tiles);
  hipStream_t st = (hipStream_t)stream;
  const bool v2_ok = plain && (Co % 16 == 0 || Co <= 32) && (!interleave || (Co / 2) % (32 * nbw) == 0) &&
                     (int64_t)y_batch_stride % 8 == 0;
  {
    // DEFAULT for the 3x3 spatial kernels with C_out % 96 == 0 or <= 32 (every heavy convolution of the VAE): the 2-D-tile
    // LDS-DMA kernel (vae_conv3.hip), frames-first tile order.  0 = automatic tile size: 256 positions with two workgroups
    // per CU unless the reduction is 384 channels x 27 taps deep (one's prologue / epilogue under the other's main loop pays
    // -3...-6 % for the short reductions of the 96- / 192-channel levels, +3...+5 % for the deepest: profiles/r04_conv3_ab.txt);
    // 8 / 9 = always 512 / 256 positions; 7 = 512, tiles of a frame first.  2 (and the experiment values 3-6) = the row-tile
    // kernel below, which also takes everything conv3 does not: 1x1 / (3,1,1) kernels, the encoder's strided down-samplers.
    const int tv = td_tuning(TD_TUNE_VAE_CONV);
    if ((tv == 0 || (tv >= 7 && tv <= 9)) && vae_conv3_eligible(p, plain)) {
      const bool four = tv == 9 || (tv == 0 && (int64_t)Ci * kt * 9 < 384 * 27);
      return vae_conv3_launch(p, (tv == 7 ? 0 : 1) | (four ? 2 : 0), st);      // bit 0: frames first; bit 1: 256-position tiles
    }
  }
  if (td_tuning(TD_TUNE_VAE_CONV) != 1 && v2_ok) {
    const bool wide = nbw == 3 && td_tuning(TD_TUNE_VAE_CONV) == 3;   // measured slower (549 vs 700 TFLOP/s at 480p): opt-in
    const int bm = wide ? 512 : 256;
    const int64_t t2 = (int64_t)B * p.To * p.Ho * td_cdiv(p.Wo, bm) * td_cdiv(Co, 32 * nbw);
    TD_REQUIRE(t2 < (1ll << 31), TD_ERR_UNSUPPORTED, "td_vae_conv: %lld tiles", (long long)t2);
    if (nbw == 1) {
      constexpr int lds = (264 + 3 * 32) * VC_ROWB;
      static std::atomic<uint64_t> m21{0};
      td_ensure_dyn_lds((const void*)vae_conv2_kernel<1, 2>, lds, m21);
      vae_conv2_kernel<1, 2><<<(unsigned)t2, 256, lds, st>>>(p);
    } else if (!wide && td_tuning(TD_TUNE_VAE_CONV) == 4) {
      constexpr int lds = 2 * (264 + 3 * 96) * 64;
      static std::atomic<uint64_t> m232{0};
      td_ensure_dyn_lds((const void*)vae_conv2_kernel<3, 2, 32>, lds, m232);
      vae_conv2_kernel<3, 2, 32><<<(unsigned)t2, 256, lds, st>>>(p);
    } else if (!wide && td_tuning(TD_TUNE_VAE_CONV) == 6 && (kw == 3 || kw == 1)) {
      constexpr int lds = 2 * (264 + 3 * 96) * 64;
      static std::atomic<uint64_t> m2323{0}, m2321{0};
      if (kw == 3) {
        td_ensure_dyn_lds((const void*)vae_conv2_kernel<3, 2, 32, 3>, lds, m2323);
        vae_conv2_kernel<3, 2, 32, 3><<<(unsigned)t2, 256, lds, st>>>(p);
      } else {
        td_ensure_dyn_lds((const void*)vae_conv2_kernel<3, 2, 32, 1>, lds, m2321);
        vae_conv2_kernel<3, 2, 32, 1><<<(unsigned)t2, 256, lds, st>>>(p);
      }
    } else if (!wide) {
      constexpr int lds = (264 + 3 * 96) * VC_ROWB;
      static std::atomic<uint64_t> m23{0};
      td_ensure_dyn_lds((const void*)vae_conv2_kernel<3, 2>, lds, m23);
      vae_conv2_kernel<3, 2><<<(unsigned)t2, 256, lds, st>>>(p);
    } else {
      constexpr int lds = (520 + 3 * 96) * VC_ROWB;
      static std::atomic<uint64_t> m43{0};
      td_ensure_dyn_lds((const void*)vae_conv2_kernel<3, 4>, lds, m43);
      vae_conv2_kernel<3, 4><<<(unsigned)t2, 256, lds, st>>>(p);
    }
    TD_CHECK_LAUNCH();
    return TD_OK;
  }
  if (nbw == 1) {
    constexpr int lds = (VC_BM + 32) * VC_ROWB;
    static std::atomic<uint64_t> m1{0};
    td_ensure_dyn_lds((const void*)vae_conv_kernel<1>, lds, m1);
    vae_conv_kernel<1><<<(unsigned)tiles, 256, lds, st>>>(p);
  } else {
    constexpr int lds = (VC_BM + 96) * VC_ROWB;
    static std::atomic<uint64_t> m3{0};
    td_ensure_dyn_lds((const void*)vae_conv_kernel<3>, lds, m3);
    vae_conv_kernel<3><<<(unsigned)tiles, 256, lds, st>>>(p);
  }
  TD_CHECK_LAUNCH();
  return TD_OK;
}

extern "C" int td_vae_conv(const void* x, int64_t x_batch_stride, const void* w, const void* bias, const void* res, void* y,
                           int64_t y_batch_stride, int B, int Ti, int Hi, int Wi, int Ci, int Co, int kt, int kh, int kw,
                           int up2, int interleave, td_stream_t stream) {
  return td_vae_conv_ex(x, x_batch_stride, w, bias, res, y, y_batch_stride, B, Ti, Hi, Wi, Ci, Co, kt, kh, kw, up2, interleave, Ti,
                        up2 ? 2 * Hi : Hi, up2 ? 2 * Wi : Wi, 1, 1, kt - 1, kh / 2, kw / 2, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// RMS_norm over the channel axis (+ SiLU) on channels-last rows, with the rounding points of the reference's bf16 path
// (wan2pt1.py:69-70: F.normalize(x, dim=C) * sqrt(C) * gamma — every operator's result rounded to bf16):
//   n = bf16(||x||_2);  q = bf16(x / max(n, 1e-12));  q = bf16(q * sqrt(C));  q = bf16(q * gamma);  y = bf16(silu(q))
// G lanes per row (G = 16 / 32 / 64 for C <= 128 / 256 / 512), 8 channels per lane.
// ------------------------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void vae_chan_rms_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gamma,
                                                           uint16_t* __restrict__ y, int64_t rows, int C, float scale, int silu) {
  constexpr int RPB = 256 / G;   // rows per pass of the block
  constexpr int R = 8;           // passes per block: eight 16-byte loads in flight per lane (one row per lane group and pass
                                 // left the kernel at 1.7-2 TB/s: too few bytes in flight per CU)
  const int sub = threadIdx.x % G;
  const int64_t row0 = (int64_t)blockIdx.x * (RPB * R) + threadIdx.x / G;
  const bool lane_live = sub * 8 < C;
  uint4 v[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int64_t row = row0 + (int64_t)i * RPB;
    v[i] = (lane_live && row < rows) ? *reinterpret_cast<const uint4*>(x + row * C + sub * 8) : make_uint4(0u, 0u, 0u, 0u);
  }
  float g[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] = 0.f;
  if (lane_live) unpack8<TD_BF16>(*reinterpret_cast<const uint4*>(gamma + sub * 8), g);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int64_t row = row0 + (int64_t)i * RPB;
    float f[8];
    unpack8<TD_BF16>(v[i], f);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float nrm = fmaxf(round_bf16(sqrtf(ss)), 1e-12f);
    if (lane_live && row < rows) {
      // x / nrm correctly rounded without the division sequence: rinv = RN(1 / nrm) by one Newton step on v_rcp_f32, then
      // Markstein's correction (as in sla_prep.hip); every rounding to bf16 is the hardware pack (v_cvt_pk_bf16_f32), two
      // elements per instruction.  (The kernel is VALU-bound, not HBM-bound: with IEEE divisions and software rounding it
      // ran at 2 TB/s.)
      float rinv = __builtin_amdgcn_rcpf(nrm);
      rinv = fmaf(fmaf(-nrm, rinv, 1.0f), rinv, rinv);
      uint32_t o16[8];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        float q0, q1;
        {
          const float a0 = f[j] * rinv, a1 = f[j + 1] * rinv;
          q0 = fmaf(fmaf(-a0, nrm, f[j]), rinv, a0);
          q1 = fmaf(fmaf(-a1, nrm, f[j + 1]), rinv, a1);
        }
        unpack2<TD_BF16>(pack2<TD_BF16>(q0, q1), q0, q1);
        unpack2<TD_BF16>(pack2<TD_BF16>(q0 * scale, q1 * scale), q0, q1);
        uint32_t w = pack2<TD_BF16>(q0 * g[j], q1 * g[j + 1]);
        if (silu) {
          unpack2<TD_BF16>(w, q0, q1);
          // silu(q) = q / (1 + exp(-q)) = q * rcp(1 + exp2(-q * log2 e)); v_exp / v_rcp are within 1 ulp of fp32, far inside
          // the bf16 rounding that follows
          q0 = q0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * q0));
          q1 = q1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * q1));
          w = pack2<TD_BF16>(q0, q1);
        }
        o16[j] = w & 0xffffu;
        o16[j + 1] = w >> 16;
      }
      uint4 ov;
      ov.x = o16[0] | (o16[1] << 16); ov.y = o16[2] | (o16[3] << 16); ov.z = o16[4] | (o16[5] << 16); ov.w = o16[6] | (o16[7] << 16);
      *reinterpret_cast<uint4*>(y + row * C + sub * 8) = ov;
    }
  }
}

extern "C" int td_vae_chan_rms(const void* x, const void* gamma, void* y, int64_t rows, int C, int silu, td_stream_t stream) {
  TD_REQUIRE(x && gamma && y, TD_ERR_INVALID, "td_vae_chan_rms: null pointer");
  TD_REQUIRE(rows > 0 && C >= 8 && C % 8 == 0 && C <= 512, TD_ERR_UNSUPPORTED, "td_vae_chan_rms: rows=%lld C=%d (multiple of 8, <= 512)", (long long)rows, C);
  const float scale = sqrtf((float)C);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 128) vae_chan_rms_kernel<16><<<(unsigned)td_cdiv(rows, 16 * 8), 256, 0, st>>>((const uint16_t*)x, (const uint16_t*)gamma, (uint16_t*)y, rows, C, scale, silu);
  else if (C <= 256) vae_chan_rms_kernel<32><<<(unsigned)td_cdiv(rows, 8 * 8), 256, 0, st>>>((const uint16_t*)x, (const uint16_t*)gamma, (uint16_t*)y, rows, C, scale, silu);
  else vae_chan_rms_kernel<64><<<(unsigned)td_cdiv(rows, 4 * 8), 256, 0, st>>>((const uint16_t*)x, (const uint16_t*)gamma, (uint16_t*)y, rows, C, scale, silu);
  TD_CHECK_LAUNCH();
  return TD_OK;
}
