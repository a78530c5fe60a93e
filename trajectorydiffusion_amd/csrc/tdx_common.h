// Shared device/host helpers for the TrajUNet hot-path kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trajdiff_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define TDX_WAVE 64

// ---- error plumbing (no exceptions across the C ABI) ---------------------------------------
void tdx_set_error(const char* fmt, ...);
#define TDX_FAIL(code, ...)      \
  do {                           \
    tdx_set_error(__VA_ARGS__);  \
    return (code);               \
  } while (0)
#define TDX_CHECK_ARG(cond, ...) \
  do {                           \
    if (!(cond)) TDX_FAIL(TDX_ERR_ARG, __VA_ARGS__); \
  } while (0)
// Index of the current device into the library's per-device caches (kernel attributes set, CU counts, static scratch addresses): one
// process may drive several GPUs, and each holds its own copy of every code object.
constexpr int kTdxMaxDevices = 64;
inline int tdx_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev < kTdxMaxDevices ? dev : kTdxMaxDevices - 1;
}

#define TDX_CHECK_LAUNCH(name)                                                       \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) TDX_FAIL(TDX_ERR_HIP, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- the narrow (128 -> 128, k = 3) conv kernel of narrow_conv.hip, entered through tdx_conv_gemm ------------------------------
int tdx_narrow_conv_try(const tdx_conv_desc* d, hipStream_t st, int cus, int* handled);
bool tdx_narrow_conv_ok(const tdx_conv_desc* d);
void tdx_narrow_conv_disable(int off);

// ---- element traits -----------------------------------------------------------------------
template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int kVec = 4;  // elements per 16-byte vector
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void unpack(const u32x4& v, float (&o)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __uint_as_float(v[i]);
  }
  static __device__ __forceinline__ u32x4 pack(const float (&o)[4]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(o[i]);
    return v;
  }
};
template <>
struct Elem<__bf16> {
  static constexpr int kVec = 8;
  static __device__ __forceinline__ float ld(const __bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void unpack(const u32x4& v, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(v[i] << 16);
      o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&o)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // RNE, NaN-preserving; a two-element vector conversion is ONE v_cvt_pk_bf16_f32
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      const f32x2_t f = {o[2 * i], o[2 * i + 1]};
      v[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
    }
    return v;
  }
};

// x * sigmoid(x) with the hardware exp2 and reciprocal (about 1e-6 relative error; an IEEE divide costs ~5x more VALU)
__device__ __forceinline__ float tdx_swish(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tdx_leaky(float x) { return x > 0.f ? x : 0.1f * x; }
__device__ __forceinline__ float tdx_act(float x, int act) {
  return act == TDX_ACT_SWISH ? tdx_swish(x) : (act == TDX_ACT_LEAKY ? tdx_leaky(x) : x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- Philox4x32-10 ---------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4(uint64_t ctr, uint32_t stream, uint64_t seed, uint32_t (&o)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), stream, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = c[i];
}
