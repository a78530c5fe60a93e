// Stem Conv1d(2 -> C0) and head Conv1d(C0 -> 2): the two convs whose GEMM K or N is tiny.
// Both are pure HBM streams over a (B, L, C0) activation.
#include "tdx_common.h"

namespace {

// one thread = one (row, 16-byte channel vector); x is (B, 2, L) fp32 channels-first.
template <typename T>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ out, int B, int L, int C0) {
  constexpr int VEC = Elem<T>::kVec;
  const int vpr = C0 / VEC;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)B * L * vpr) return;
  const int cv = (int)(gid % vpr);
  const long n = gid / vpr;
  const int b = (int)(n / L), l = (int)(n % L);
  const float* xb = x + (size_t)b * 2 * L;
  float xin[2][3];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int u = l + t - 1;
      xin[ci][t] = (u >= 0 && u < L) ? xb[ci * L + u] : 0.f;
    }
  float v[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int co = cv * VEC + j;
    float a = bias[co];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int t = 0; t < 3; ++t) a = fmaf(w[(co * 2 + ci) * 3 + t], xin[ci][t], a);
    v[j] = a;
  }
  *reinterpret_cast<u32x4*>(out + (size_t)n * C0 + cv * VEC) = Elem<T>::pack(v);
}

// One wavefront per output row: lanes split the channels (16-byte vectors), each accumulates the
// row's contribution to out[l-1] (tap 2), out[l] (tap 1), out[l+1] (tap 0) ... instead we gather:
// out[b, o, l] = bias[o] + sum_t sum_c w[o, c, t] * a[b, l+t-1, c]; lanes split c, then a wave reduction.
template <typename T>
__global__ __launch_bounds__(256) void head_conv_kernel(const T* __restrict__ a, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B, int L, int C0) {
  constexpr int VEC = Elem<T>::kVec;
  const int lane = threadIdx.x & 63;
  const long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= (long)B * L) return;  // wave-uniform
  const int b = (int)(n / L), l = (int)(n % L);
  float acc0 = 0.f, acc1 = 0.f;
  for (int c0 = lane * VEC; c0 < C0; c0 += 64 * VEC) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int u = l + t - 1;
      if (u < 0 || u >= L) continue;  // wave-uniform
      float v[VEC];
      Elem<T>::unpack(*reinterpret_cast<const u32x4*>(a + ((size_t)b * L + u) * C0 + c0), v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        acc0 = fmaf(w[(0 * C0 + c0 + j) * 3 + t], v[j], acc0);
        acc1 = fmaf(w[(1 * C0 + c0 + j) * 3 + t], v[j], acc1);
      }
    }
  }
  acc0 = wave_sum(acc0);
  acc1 = wave_sum(acc1);
  if (lane == 0) {
    out[((size_t)b * 2 + 0) * L + l] = acc0 + bias[0];
    out[((size_t)b * 2 + 1) * L + l] = acc1 + bias[1];
  }
}

}  // namespace

extern "C" int tdx_stem_conv(const float* x, const float* w, const float* bias, void* out, int32_t B, int32_t L,
                             int32_t C0, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(x && w && bias && out, "tdx_stem_conv: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_stem_conv: bad dtype %d", dtype);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  TDX_CHECK_ARG(B > 0 && L > 0 && C0 > 0 && C0 % vec == 0, "tdx_stem_conv: bad shape B=%d L=%d C0=%d", B, L, C0);
  const long total = (long)B * L * (C0 / vec);
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDX_F32)
    hipLaunchKernelGGL(stem_conv_kernel<float>, grid, dim3(256), 0, st, x, w, bias, (float*)out, B, L, C0);
  else
    hipLaunchKernelGGL(stem_conv_kernel<__bf16>, grid, dim3(256), 0, st, x, w, bias, (__bf16*)out, B, L, C0);
  TDX_CHECK_LAUNCH("stem_conv_kernel");
  return TDX_OK;
}

extern "C" int tdx_head_conv(const void* a, const float* w, const float* bias, float* out, int32_t B, int32_t L,
                             int32_t C0, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(a && w && bias && out, "tdx_head_conv: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_head_conv: bad dtype %d", dtype);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  TDX_CHECK_ARG(B > 0 && L > 0 && C0 > 0 && C0 % vec == 0, "tdx_head_conv: bad shape B=%d L=%d C0=%d", B, L, C0);
  const long rows = (long)B * L;
  dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDX_F32)
    hipLaunchKernelGGL(head_conv_kernel<float>, grid, dim3(256), 0, st, (const float*)a, w, bias, out, B, L, C0);
  else
    hipLaunchKernelGGL(head_conv_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)a, w, bias, out, B, L, C0);
  TDX_CHECK_LAUNCH("head_conv_kernel");
  return TDX_OK;
}
