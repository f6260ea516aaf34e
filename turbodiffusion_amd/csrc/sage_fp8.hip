// a13, FP8-PV variant — V preparation for the INT8-QK / FP8-PV SageAttention kernel (attn.hip, PV8 instantiation).
//
// Reference: the sm89+ branch of SageSparseLinearAttention.forward (SLA/core.py:217-239):
//   fused.transpose_pad_permute_cuda(v, v_t, 1)            V -> [b, h, d, padded keys], keys permuted for the MMA operand
//   fused.scale_fuse_quant_cuda(v_t, v_fp8, v_scale, L, 2.25, 1)   per (h, d) channel: scale = max_l |v| / 2.25,
//                                                                  v_fp8 = e4m3(v / scale)
// (un-vendored SpargeAttn kernels; arithmetic stated in oracle/sla_ref.py: v_fp8_quant).
//
// MI355X layout: fp8 tiles [H, ceil(L/64), 128 d, 64 positions] (8 KB per 64-key block, half the fp16 tile).  The PV
// contraction runs on v_mfma_f32_32x32x64_f8f6f4 (one MFMA per 32 d x 32 q x 64 keys, 2x the fp16 rate): its B operand
// is the lane's 32 probabilities in the order the INT8 QK^T MFMA left them (key(g, r, hi) = 32g + (r&3) + 8(r>>2) + 4hi
// for byte j = 16g + r of half-wave hi), so the A operand — a V^T row — stores key(g, r, hi) at position 32hi + 16g + r.
#include "td_common.h"
#include <hip/hip_fp8.h>

#define VF_NCH 64   // partial-max chunks per head

// pass 1: per-(h, d) max |v| over this chunk's rows -> ws [H, VF_NCH, 128]
template <int IDT>
__global__ __launch_bounds__(256) void v_amax_partial_kernel(const uint16_t* __restrict__ v, int64_t stride_h,
                                                             int64_t stride_l, float* __restrict__ ws, int64_t L) {
  __shared__ float red[16][128];
  const int tid = threadIdx.x, c8 = tid & 15, rl = tid >> 4;
  const int ch = blockIdx.x, h = blockIdx.y;
  const int64_t rows = td_cdiv(L, VF_NCH);
  const int64_t r0 = ch * rows, r1 = (r0 + rows < L) ? r0 + rows : L;
  float m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t l = r0 + rl; l < r1; l += 16) {
    const uint4 raw = *reinterpret_cast<const uint4*>(v + h * stride_h + l * stride_l + c8 * 8);
    float f[8];
    unpack8<IDT>(raw, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], fabsf(f[j]));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][c8 * 8 + j] = m[j];
  __syncthreads();
  if (tid < 128) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a = fmaxf(a, red[r][tid]);
    ws[((int64_t)h * VF_NCH + ch) * 128 + tid] = a;
  }
}

// pass 1b: v_scale[h, d] = max over chunks / scale_max
__global__ void v_scale_final_kernel(const float* __restrict__ ws, float* __restrict__ v_scale, float scale_max) {
  const int h = blockIdx.x, d = threadIdx.x;
  float a = 0.f;
  for (int c = 0; c < VF_NCH; ++c) a = fmaxf(a, ws[((int64_t)h * VF_NCH + c) * 128 + d]);
  v_scale[h * 128 + d] = a / scale_max;
}

// pass 2: one workgroup per (64-key block, head): transpose, divide by the channel scale, convert to OCP e4m3 (RNE,
// saturating), store in the position order above.  Tail keys -> 0.
template <int IDT>
__global__ __launch_bounds__(256) void v_fp8_tiles_kernel(const uint16_t* __restrict__ v, int64_t stride_h, int64_t stride_l,
                                                          const float* __restrict__ v_scale, uint8_t* __restrict__ vt8,
                                                          int64_t L, int Kb) {
  __shared__ uint8_t tile[64][132];   // [key][d] bytes (+4 pad)
  const int tid = threadIdx.x;
  const int kb = blockIdx.x, h = blockIdx.y;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vec = tid + 256 * i;
    const int key = vec >> 4, c8 = vec & 15;
    const int64_t l = (int64_t)kb * 64 + key;
    uint4 raw = make_uint4(0, 0, 0, 0);
    if (l < L) raw = *reinterpret_cast<const uint4*>(v + h * stride_h + l * stride_l + c8 * 8);
    float f[8];
    unpack8<IDT>(raw, f);
    const float4 s0 = *reinterpret_cast<const float4*>(v_scale + h * 128 + c8 * 8);
    const float4 s1 = *reinterpret_cast<const float4*>(v_scale + h * 128 + c8 * 8 + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const float a = fminf(fmaxf(f[j] / fmaxf(sc[j], 1e-30f), -448.f), 448.f);
      const float b = fminf(fmaxf(f[j + 1] / fmaxf(sc[j + 1], 1e-30f), -448.f), 448.f);
      const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
      tile[key][c8 * 8 + j] = (uint8_t)(pk & 0xff);
      tile[key][c8 * 8 + j + 1] = (uint8_t)((pk >> 8) & 0xff);
    }
  }
  __syncthreads();
  uint8_t* out = vt8 + ((int64_t)h * Kb + kb) * (128 * 64);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int vec = tid + 256 * i;            // output vector: row d = vec / 4, 16-byte slot = vec % 4
    const int d = vec >> 2, slot = vec & 3;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int pos = slot * 16 + b;
      const int hi = pos >> 5, g = (pos >> 4) & 1, r = pos & 15;
      const int key = 32 * g + (r & 3) + 8 * (r >> 2) + 4 * hi;
      w[b >> 2] |= (uint32_t)tile[key][d] << (8 * (b & 3));
    }
    *reinterpret_cast<uint4*>(out + d * 64 + slot * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

extern "C" int td_v_fp8_tiles(const void* v, int in_dtype, int64_t stride_h, int64_t stride_l, uint8_t* vt8,
                              float* v_scale, float* ws, float scale_max, int64_t L, int H, int D, td_stream_t stream) {
  TD_REQUIRE(v && vt8 && v_scale && ws, TD_ERR_INVALID, "td_v_fp8_tiles: null pointer");
  TD_REQUIRE(D == 128, TD_ERR_UNSUPPORTED, "td_v_fp8_tiles: D=%d (need 128)", D);
  TD_REQUIRE(L > 0 && H > 0 && scale_max > 0.f, TD_ERR_INVALID, "td_v_fp8_tiles: L=%lld H=%d scale_max=%g", (long long)L, H, scale_max);
  TD_REQUIRE(stride_l % 8 == 0 && stride_h % 8 == 0, TD_ERR_UNSUPPORTED, "td_v_fp8_tiles: strides");
  TD_REQUIRE(in_dtype == TD_BF16 || in_dtype == TD_F16, TD_ERR_UNSUPPORTED, "td_v_fp8_tiles: dtype %d", in_dtype);
  hipStream_t st = (hipStream_t)stream;
  const int Kb = (int)td_cdiv(L, 64);
  if (in_dtype == TD_BF16) {
    v_amax_partial_kernel<TD_BF16><<<dim3(VF_NCH, H), 256, 0, st>>>((const uint16_t*)v, stride_h, stride_l, ws, L);
    v_scale_final_kernel<<<H, 128, 0, st>>>(ws, v_scale, scale_max);
    v_fp8_tiles_kernel<TD_BF16><<<dim3(Kb, H), 256, 0, st>>>((const uint16_t*)v, stride_h, stride_l, v_scale, vt8, L, Kb);
  } else {
    v_amax_partial_kernel<TD_F16><<<dim3(VF_NCH, H), 256, 0, st>>>((const uint16_t*)v, stride_h, stride_l, ws, L);
    v_scale_final_kernel<<<H, 128, 0, st>>>(ws, v_scale, scale_max);
    v_fp8_tiles_kernel<TD_F16><<<dim3(Kb, H), 256, 0, st>>>((const uint16_t*)v, stride_h, stride_l, v_scale, vt8, L, Kb);
  }
  TD_CHECK_LAUNCH();
  return TD_OK;
}
