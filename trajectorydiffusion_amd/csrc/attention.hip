// Mid-block attention core: single head, L <= 16 tokens (13 at L=200), head dim = C  (TrajUNetBlocks.py:117-125).
//
// One WAVE per trajectory, every q/k/v element read from HBM exactly once:
//   1. S^T = K Q^T on MFMA.  Both fragments are plain 16-byte global loads (a lane holds 8 bf16 / 4 fp32 consecutive
//      channels of one token), tokens >= L are zero fragments.  The accumulator then holds, in lane (i = lane & 15,
//      g = lane >> 4), the scores of query i against keys 4g .. 4g+3.
//   2. softmax over the keys: 4 values in the lane, then lanes i, i+16, i+32, i+48 (xor-shuffles 16 and 32); fp32.
//   3. out = P V on the vector ALU: P goes through a 1 KB LDS patch per wave (broadcast reads), each lane owns
//      16-byte channel vectors and reads each V row once.
// kMaxTok = 64 tokens are still served by the general kernel below.
#include "tdx_common.h"

namespace {

constexpr int kMaxTok = 64;

template <typename T>
struct QK;
template <>
struct QK<__bf16> {  // 32 channels per step: v_mfma_f32_16x16x32_bf16, 8 channels per lane
  static constexpr int CH = 32;
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct QK<float> {  // 16 channels per step: four v_mfma_f32_16x16x4f32; element s of the lane's float4 is k-slot (lane >> 4) of
  static constexpr int CH = 16;  // MFMA s, for both operands alike (a consistent permutation of the summation index)
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[s]), __uint_as_float(b[s]), c, 0, 0, 0);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void attention16_kernel(const T* __restrict__ qkv, T* __restrict__ out, int B, int L, int C, float scale) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int VEC = Elem<T>::kVec;
  constexpr int CH = QK<T>::CH;
  __shared__ float P[4][16 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;  // wave-uniform; no block-wide barrier below
  const int i = lane & 15, g = lane >> 4;
  const T* base = qkv + (size_t)b * L * 3 * C;
  // ---- 1. S^T ----
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const bool tok = i < L;
  const T* qrow = base + (size_t)(tok ? i : 0) * 3 * C + g * VEC;  // channels c0 + VEC * g ...
  const T* krow = qrow + C;
  constexpr int UN = 8;
  for (int c0 = 0; c0 < C; c0 += CH * UN) {
    u32x4 fq[UN], fk[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool ok = tok && c0 + u * CH < C;
      fq[u] = ok ? *reinterpret_cast<const u32x4*>(qrow + c0 + u * CH) : (u32x4){0u, 0u, 0u, 0u};
      fk[u] = ok ? *reinterpret_cast<const u32x4*>(krow + c0 + u * CH) : (u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) QK<T>::mma(fk[u], fq[u], acc);  // A = K (rows = keys), B = Q (columns = queries)
  }
  // ---- 2. softmax over keys j = 4g + r ----
  float sc[4], m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sc[r] = 4 * g + r < L ? acc[r] * scale : -INFINITY;
    m = fmaxf(m, sc[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sc[r] = expf(sc[r] - m);
    sum += sc[r];
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  *reinterpret_cast<f32x4*>(&P[wave][i * 16 + 4 * g]) = (f32x4){sc[0] * inv, sc[1] * inv, sc[2] * inv, sc[3] * inv};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- 3. out = P V ----
  const T* vbase = base + 2 * C;
  T* obase = out + (size_t)b * L * C;
  for (int cv = lane; cv * VEC < C; cv += 64) {
    float v[16][VEC];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < L) Elem<T>::unpack(*reinterpret_cast<const u32x4*>(vbase + (size_t)j * 3 * C + cv * VEC), v[j]);
      else
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[j][e] = 0.f;
    }
    for (int q = 0; q < L; ++q) {
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4 pj = *reinterpret_cast<const f32x4*>(&P[wave][q * 16 + 4 * j4]);  // same address in every lane: broadcast
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = fmaf(pj[r], v[4 * j4 + r][e], o[e]);
      }
      *reinterpret_cast<u32x4*>(obase + (size_t)q * C + cv * VEC) = Elem<T>::pack(o);
    }
  }
#endif
}

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
  if (L <= 16 && C % (dtype == TDX_F32 ? 16 : 32) == 0) {
    const dim3 grid((unsigned)((B + 3) / 4));
    if (dtype == TDX_F32)
      hipLaunchKernelGGL(attention16_kernel<float>, grid, dim3(256), 0, st, (const float*)qkv, (float*)out, B, L, C, scale);
    else
      hipLaunchKernelGGL(attention16_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)qkv, (__bf16*)out, B, L, C, scale);
    TDX_CHECK_LAUNCH("attention16_kernel");
    return TDX_OK;
  }
  if (dtype == TDX_F32)
    hipLaunchKernelGGL(attention_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)qkv, (float*)out, L, C, scale);
  else
    hipLaunchKernelGGL(attention_kernel<__bf16>, dim3(B), dim3(256), 0, st, (const __bf16*)qkv, (__bf16*)out, L, C, scale);
  TDX_CHECK_LAUNCH("attention_kernel");
  return TDX_OK;
}
