// Mid-block attention core: single head, L <= 64 tokens (13 at L=200), head dim = C.
// One workgroup per trajectory; scores live in LDS, q/k/v are read straight from L2.
#include "tdx_common.h"

namespace {

constexpr int kMaxTok = 64;

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int L, int C, float scale) {
  constexpr int VEC = Elem<T>::kVec;
  __shared__ float S[kMaxTok * kMaxTok];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* base = qkv + (size_t)b * L * 3 * C;
  const int nvec = C / VEC;

  // scores: one (i, j) pair per wavefront iteration, lanes split the channel vectors
  for (int pr = wave; pr < L * L; pr += 4) {
    const int i = pr / L, j = pr % L;
    const T* q = base + (size_t)i * 3 * C;
    const T* k = base + (size_t)j * 3 * C + C;
    float acc = 0.f;
    for (int cv = lane; cv < nvec; cv += 64) {
      float a[VEC], kk[VEC];
      Elem<T>::unpack(*reinterpret_cast<const u32x4*>(q + cv * VEC), a);
      Elem<T>::unpack(*reinterpret_cast<const u32x4*>(k + cv * VEC), kk);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc = fmaf(a[e], kk[e], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) S[i * kMaxTok + j] = acc * scale;
  }
  __syncthreads();
  // softmax over keys (dim=2 of (B, L, L))
  if (threadIdx.x < L) {
    float* row = S + threadIdx.x * kMaxTok;
    float m = row[0];
    for (int j = 1; j < L; ++j) m = fmaxf(m, row[j]);
    float s = 0.f;
    for (int j = 0; j < L; ++j) {
      const float e = expf(row[j] - m);
      row[j] = e;
      s += e;
    }
    const float inv = 1.0f / s;
    for (int j = 0; j < L; ++j) row[j] *= inv;
  }
  __syncthreads();
  // out = P @ v: one (channel vector, block of 4 query rows) item per thread iteration
  const int iblocks = (L + 3) / 4;
  for (int item = threadIdx.x; item < nvec * iblocks; item += 256) {
    const int cv = item % nvec, i0 = (item / nvec) * 4;
    float acc[4][VEC];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[ii][e] = 0.f;
    for (int j = 0; j < L; ++j) {
      float v[VEC];
      Elem<T>::unpack(*reinterpret_cast<const u32x4*>(base + (size_t)j * 3 * C + 2 * C + cv * VEC), v);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const float pij = (i0 + ii < L) ? S[(i0 + ii) * kMaxTok + j] : 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[ii][e] = fmaf(pij, v[e], acc[ii][e]);
      }
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
      if (i0 + ii < L)
        *reinterpret_cast<u32x4*>(out + ((size_t)b * L + i0 + ii) * C + cv * VEC) = Elem<T>::pack(acc[ii]);
  }
}

}  // namespace

extern "C" int tdx_attention(const void* qkv, void* out, int32_t B, int32_t L, int32_t C, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(qkv && out, "tdx_attention: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_attention: bad dtype %d", dtype);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  TDX_CHECK_ARG(B > 0 && L > 0 && C > 0 && C % vec == 0, "tdx_attention: bad shape B=%d L=%d C=%d", B, L, C);
  if (L > kMaxTok) TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_attention: %d tokens > %d supported", L, kMaxTok);
  const float scale = 1.0f / sqrtf((float)C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDX_F32)
    hipLaunchKernelGGL(attention_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)qkv, (float*)out, L, C, scale);
  else
    hipLaunchKernelGGL(attention_kernel<__bf16>, dim3(B), dim3(256), 0, st, (const __bf16*)qkv, (__bf16*)out, L, C, scale);
  TDX_CHECK_LAUNCH("attention_kernel");
  return TDX_OK;
}
