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

// tuning knobs (td_set_tuning); 0 = automatic choice
int td_tuning(int key);
int td_gemm_fast_g(void);                  // resolved TD_TUNE_GEMM_FAST: 0 = exact dequant, else the recentring period G
unsigned long long* td_dbg_buffer(void);  // 256 x u64 device scratch for the DBG kernel instantiations
// internal kernel entry points shared between translation units
int td_gemm_w8a8_fi(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                    const void* bias, void* d, int out_dtype, int epilogue, int64_t m, int64_t n,
                    int64_t k, int64_t ldd, hipStream_t st);
int td_gemm_w8a8_fi_q(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                      int8_t* d_q, float* d_s, int act_dtype, int epilogue, int64_t m, int64_t n, int64_t k,
                      hipStream_t st);
int td_gemm_w8a8_fi_res(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                        void* x, const float* gate, int dtype, int64_t m, int64_t n, int64_t k, int64_t ldx,
                        hipStream_t st);
int td_gemm_w8a8_fi_stats(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                          void* d_or_x, const float* gate, int residual, int64_t m, int64_t n, int64_t k, int64_t ld,
                          float* stats_ws, hipStream_t st);
int td_gemm_w8a8_fi_vt(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias, void* d,
                       int out_dtype, int64_t m, int64_t n, int64_t k, int64_t ldd, int64_t v_col0, void* vt, int vt_f16,
                       hipStream_t st);
int td_gemm_w8a8_m32(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s,
                     const void* bias, void* d, int out_dtype, int epilogue, int64_t m, int64_t n,
                     int64_t k, int64_t ldd, hipStream_t st);
int td_gemm_w8a8_m32_q(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                       int8_t* d_q, float* d_s, int act_dtype, int epilogue, int64_t m, int64_t n, int64_t k,
                       hipStream_t st);
int td_gemm_w8a8_m32_res(const int8_t* a, const float* a_s, const int8_t* b, const float* b_s, const void* bias,
                         void* x, const float* gate, int dtype, int64_t m, int64_t n, int64_t k, int64_t ldx,
                         hipStream_t st);

__host__ __device__ static inline int64_t td_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: one process driving several GPUs (or
// several host threads) must set it once per device, not once per process.  `mask` = one static atomic per kernel
// instantiation, bit d = "set on device d" (devices >= 64 simply set it on every launch).
#include <atomic>
static inline void td_ensure_dyn_lds(const void* kern, int bytes, std::atomic<uint64_t>& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0ull;
  if (bit && (mask.load(std::memory_order_acquire) & bit)) return;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (bit) mask.fetch_or(bit, std::memory_order_release);
}

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

// GELU-tanh (wan2pt1.py:375 nn.GELU(approximate="tanh")) on a value already rounded to the output dtype.
// 0.5*x*(1+tanh(u)) == x*sigmoid(2u), u = sqrt(2/pi)*(x + 0.044715 x^3)  — an exact identity; evaluated as
// x * rcp(1 + exp2(x * (c + c*k1*x^2))), c = -2*sqrt(2/pi)*log2(e), on v_exp_f32 / v_rcp_f32: 5 plain VALU + 2
// transcendentals per element instead of 11 + 2 for the (1 + tanh) form, and no cancellation anywhere (the
// (1 + tanh u) form loses all its digits for x < -4, where torch's own fp32 result is off by up to 30 %; there the two
// differ by < 2e-7 absolute, everywhere else they round to the same 16-bit value).  Shared by every GEMM kernel so
// that all variants are bit-identical.
__device__ __forceinline__ float td_gelu_tanh(float x) {
  const float c = -2.302208198144167f;     // -2*sqrt(2/pi)*log2(e)
  const float ck1 = -0.10294324f;          // c * 0.044715
  const float p = fmaf(x * x, ck1, c);
  const float e = __builtin_amdgcn_exp2f(x * p);  // exp(-2u)
  return x * __builtin_amdgcn_rcpf(e + 1.0f);
}

// unpack one packed 32-bit word of two 16-bit values to floats
template <int DT> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
  if constexpr (DT == TD_BF16) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
  else { lo = f16_bits_to_f32(w & 0xffffu); hi = f16_bits_to_f32(w >> 16); }
}
// round an fp32 value to the 16-bit dtype and back (hardware RNE)
template <int DT> __device__ __forceinline__ float round_half(float x) {
  float lo, hi;
  unpack2<DT>(pack2<DT>(x, 0.f), lo, hi);
  return lo;
}
// GEMM epilogue for two adjacent outputs (Int8Linear.forward, ops/core.py:408-412 and the FFN GELU):
// cast(acc) -> [+ bias, cast] -> [gelu_tanh, cast]; every cast is the hardware RNE pack (v_cvt_pk_*).
// b0,b1: the bias values already widened to fp32.  Returns the packed 16-bit pair.
template <int ODT, int EPI, bool HAS_BIAS>
__device__ __forceinline__ uint32_t td_gemm_epilogue2(float a0, float a1, float b0, float b1) {
  uint32_t w = pack2<ODT>(a0, a1);
  if constexpr (HAS_BIAS) {
    float x0, x1; unpack2<ODT>(w, x0, x1);
    w = pack2<ODT>(x0 + b0, x1 + b1);
  }
  if constexpr (EPI == TD_EPI_GELU_TANH) {
    float x0, x1; unpack2<ODT>(w, x0, x1);
    w = pack2<ODT>(td_gelu_tanh(x0), td_gelu_tanh(x1));
  }
  return w;
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

// Sequence-parallel PACK layout (seqpar.py): the per-head outputs of a producer kernel are written straight into the send
// buffer of the K-side all-gather, which is [groups][section][heads_per_group][...] — head h of a section lives at
// section base + (h / hg) * gs + (h % hg) * hs  (hs = that section's natural per-head extent; gs = group stride, both in
// elements of the section's type).  hg == 0: the flat layout, h * hs.
__device__ __forceinline__ int64_t td_head_off(int h, int hg, int64_t gs, int64_t hs) {
  return hg > 0 ? (int64_t)(h / hg) * gs + (int64_t)(h % hg) * hs : (int64_t)h * hs;
}
