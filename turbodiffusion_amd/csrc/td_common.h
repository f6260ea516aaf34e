// Shared device/host helpers for the gfx950 (MI355X / CDNA4) kernels.
// Everything here is written for 64-lane wavefronts; nothing is portable on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/turbodiffusion_amd.h"

#define TD_WAVE 64

typedef int      v4i  __attribute__((ext_vector_type(4)));
typedef int      v2i  __attribute__((ext_vector_type(2)));
typedef int      v16i __attribute__((ext_vector_type(16)));
typedef float    v16f __attribute__((ext_vector_type(16)));
typedef float    v4f  __attribute__((ext_vector_type(4)));
typedef _Float16 v8h  __attribute__((ext_vector_type(8)));
typedef short    v8s  __attribute__((ext_vector_type(8)));
typedef __bf16   v8bf __attribute__((ext_vector_type(8)));

// ---- error plumbing (host) -------------------------------------------------
void td_set_error(const char* fmt, ...);
#define TD_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      td_set_error(__VA_ARGS__);               \
      return (code);                           \
    }                                          \
  } while (0)
#define TD_CHECK_LAUNCH()                                              \
  do {                                                                 \
    hipError_t e_ = hipGetLastError();                                 \
    if (e_ != hipSuccess) {                                            \
      td_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,      \
                   hipGetErrorString(e_));                             \
      return TD_ERR_LAUNCH;                                            \
    }                                                                  \
  } while (0)

__host__ __device__ static inline int64_t td_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- scalar conversions (device) ------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// fp32 -> bf16, round-to-nearest-even (matches torch's .to(bfloat16))
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float round_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
  union { uint16_t u; _Float16 h; } c; c.u = (uint16_t)b; return (float)c.h;
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
  union { uint16_t u; _Float16 h; } c; c.h = (_Float16)f; return c.u;  // RNE
}

// load element i of a 16-bit typed array as float; DT = TD_BF16 | TD_F16
template <int DT> __device__ __forceinline__ float half_bits_to_f32(uint32_t b) {
  if constexpr (DT == TD_BF16) return bf16_bits_to_f32(b); else return f16_bits_to_f32(b);
}
template <int DT> __device__ __forceinline__ uint32_t f32_to_half_bits(float f) {
  if constexpr (DT == TD_BF16) return f32_to_bf16_bits(f); else return f32_to_f16_bits(f);
}

// unpack 8 x 16-bit (one 16-byte vector) to floats
template <int DT> __device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i]     = half_bits_to_f32<DT>(w[i] & 0xffffu);
    f[2 * i + 1] = half_bits_to_f32<DT>(w[i] >> 16);
  }
}
typedef __bf16   bf2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h2_t  __attribute__((ext_vector_type(2)));
typedef float    f2_t  __attribute__((ext_vector_type(2)));
// two fp32 -> one packed 32-bit word of two 16-bit values (lo = a, hi = b), RNE:
// v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950
template <int DT> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  f2_t v = {a, b};
  if constexpr (DT == TD_BF16) {
    bf2_t r = __builtin_convertvector(v, bf2_t);
    return *reinterpret_cast<uint32_t*>(&r);
  } else {
    h2_t r = __builtin_convertvector(v, h2_t);
    return *reinterpret_cast<uint32_t*>(&r);
  }
}
template <int DT> __device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack2<DT>(f[0], f[1]), pack2<DT>(f[2], f[3]), pack2<DT>(f[4], f[5]),
                    pack2<DT>(f[6], f[7]));
}

// ---- wave-level reductions (64 lanes) --------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a 1-D block id: consecutive *virtual* ids land on the
// same XCD (hardware round-robins physical ids over the 8 XCDs), so tiles that share an
// operand panel share one XCD's L2.  Pure speed: any placement is correct.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
  const uint32_t NX = 8;
  uint32_t xcd = bid % NX, idx = bid / NX;
  uint32_t q = nwg / NX, r = nwg % NX;
  uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
