// Small HBM-bound kernels: embedding-input gather, DDPM/DDIM updates (with in-kernel Philox),
// q-sample, and the device-resident step bookkeeping that lets one hipGraph replay all T steps.
#include "tdx_common.h"

namespace {

constexpr int kEmb = 128, kHid = 256;

// one workgroup per sample: 128 + 1280 + 256 outputs
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ time, const int64_t* __restrict__ cat,
                                                           const float* __restrict__ num, const float* __restrict__ pos_enc,
                                                           int max_time, const float* __restrict__ cell_emb,
                                                           const float* __restrict__ day_emb, const float* __restrict__ wide_w,
                                                           const float* __restrict__ wide_b, float* __restrict__ pe,
                                                           float* __restrict__ catvec, float* __restrict__ deep_in, int B,
                                                           int* __restrict__ err_flag) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const long t = time[b];
  const bool t_ok = t >= 0 && t < max_time;
  if (!t_ok && tid == 0) atomicOr(err_flag, 1);
  if (tid < kEmb) pe[(size_t)b * kEmb + tid] = t_ok ? pos_enc[(size_t)t * kEmb + tid] : 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const long idx = cat[(size_t)b * 5 + k];
    const int rows = k < 4 ? 256 : 33;
    const bool ok = idx >= 0 && idx < rows;
    if (!ok && tid == 0) atomicOr(err_flag, 2);
    const float* tab = k < 4 ? cell_emb : day_emb;
    catvec[(size_t)b * 5 * kHid + k * kHid + tid] = ok ? tab[(size_t)idx * kHid + tid] : 0.f;
  }
  const float n0 = num[(size_t)b * 3], n1 = num[(size_t)b * 3 + 1], n2 = num[(size_t)b * 3 + 2];
  // nn.Linear(3, 256): same left-to-right fp32 accumulation as a plain dot product
  float w = wide_w[tid * 3] * n0;
  w = fmaf(wide_w[tid * 3 + 1], n1, w);
  w = fmaf(wide_w[tid * 3 + 2], n2, w);
  w += wide_b[tid];
  deep_in[(size_t)b * 4 * kHid + 3 * kHid + tid] = tdx_leaky(w);
}

// (Philox4x32-10: tdx_common.h)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
  const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * __logf(u1));
  float s, c;
  __sincosf(6.28318530717958647692f * u2, &s, &c);
  z0 = r * c;
  z1 = r * s;
}

// 4 elements per thread; n % 4 == 0 is guaranteed by the host (per-sample size 2*L, L even or padded path)
__global__ __launch_bounds__(256) void ddpm_update_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                          const float* __restrict__ z, const float* __restrict__ coef, long n4,
                                                          long n, uint64_t seed, const int* __restrict__ step_ctr, long first_elem) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float c_eps = coef[0], sqrt_alpha = coef[1], sigma = coef[2];
  const long e0 = i * 4;
  float xv[4], ev[4], zv[4] = {0.f, 0.f, 0.f, 0.f};
  const bool full = e0 + 4 <= n;
  if (full) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + e0), e = *reinterpret_cast<const f32x4*>(eps + e0);
#pragma unroll
    for (int j = 0; j < 4; ++j) { xv[j] = a[j]; ev[j] = e[j]; }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) { xv[j] = e0 + j < n ? x[e0 + j] : 0.f; ev[j] = e0 + j < n ? eps[e0 + j] : 0.f; }
  }
  if (sigma != 0.f) {
    if (z) {
#pragma unroll
      for (int j = 0; j < 4; ++j) zv[j] = e0 + j < n ? z[e0 + j] : 0.f;
    } else {
      uint32_t r[4];
      // step_ctr = device int32[4] {step, seed_lo, seed_hi, stream}: a seed that can change between graph replays; the Philox
      // stream id is step_ctr[3] (the diffusion step t: tdx_step_prepare writes it), so every sampling path draws the same z
      const uint64_t dyn = step_ctr ? (((uint64_t)(uint32_t)step_ctr[2] << 32) | (uint32_t)step_ctr[1]) : 0ull;
      philox4((uint64_t)(first_elem + e0) >> 2, (uint32_t)(step_ctr ? step_ctr[3] : 0), seed ^ dyn, r);
      box_muller(r[0], r[1], zv[0], zv[1]);
      box_muller(r[2], r[3], zv[2], zv[3]);
    }
  }
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // DiffusionManager.py:50: mu = (x_t - (beta / sqrt(1-abar)) * eps) / sqrt(alpha), same op order, no contraction
    const float mu = (xv[j] - c_eps * ev[j]) / sqrt_alpha;
    o[j] = sigma != 0.f ? mu + sigma * zv[j] : mu;
  }
  if (full) {
    *reinterpret_cast<f32x4*>(x + e0) = (f32x4){o[0], o[1], o[2], o[3]};
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (e0 + j < n) x[e0 + j] = o[j];
  }
}

__global__ __launch_bounds__(256) void ddim_update_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                          const float* __restrict__ coef, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float s1m_t = coef[0], inv_sa_t = coef[1], sa_p = coef[2], s1m_p = coef[3];
  const float e = eps[i];
  const float x0 = (x[i] - s1m_t * e) * inv_sa_t;
  x[i] = sa_p * x0 + s1m_p * e;
}

__global__ __launch_bounds__(256) void step_prepare_kernel(int* __restrict__ step_ctr, const float* __restrict__ coef_table, int ncoef,
                                                           float* __restrict__ coef_cur, const float* __restrict__ tbias_table, int ntb,
                                                           float* __restrict__ tbias_cur, const int* __restrict__ step_to_t,
                                                           int64_t* __restrict__ t_vec, int B, int nsteps) {
  // every block reads the counter; only after ALL blocks are done may it advance, so the advance is
  // done by a separate single-thread kernel launched right after this one (step_advance_kernel).
  int step = *step_ctr;
  step = step < 0 ? 0 : (step >= nsteps ? nsteps - 1 : step);
  const long gid = (long)blockIdx.x * 256 + threadIdx.x, gsz = (long)gridDim.x * 256;
  for (long i = gid; i < ncoef; i += gsz) coef_cur[i] = coef_table[(size_t)step * ncoef + i];
  if (tbias_table)
    for (long i = gid; i < ntb; i += gsz) tbias_cur[i] = tbias_table[(size_t)step * ntb + i];
  if (t_vec) {
    const int64_t t = step_to_t[step];
    for (long i = gid; i < B; i += gsz) t_vec[i] = t;
  }
}
// step_ctr[3] = Philox stream id of this step's noise = the diffusion step t (or the step index without a step_to_t table)
__global__ void step_advance_kernel(int* __restrict__ step_ctr, const int* __restrict__ step_to_t, int nsteps) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int step = *step_ctr;
    const int s = step < 0 ? 0 : (step >= nsteps ? nsteps - 1 : step);
    step_ctr[3] = step_to_t ? step_to_t[s] : s;
    *step_ctr = step + 1;
  }
}

__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                                                       const int64_t* __restrict__ t, const float* __restrict__ sa,
                                                       const float* __restrict__ s1m, int T, float* __restrict__ out, int B,
                                                       int per_sample, int* __restrict__ err_flag) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * per_sample) return;
  const int b = (int)(i / per_sample);
  const long tb = t[b];
  if (tb < 0 || tb >= T) {
    atomicOr(err_flag, 4);
    out[i] = 0.f;
    return;
  }
  out[i] = sa[tb] * x0[i] + s1m[tb] * eps[i];  // DiffusionManager.py:33 (mul, mul, add: no fma contraction)
}

}  // namespace

extern "C" int tdx_embed_gather(const int64_t* time, const int64_t* cat, const float* num, const float* pos_enc,
                                int32_t max_time, const float* cell_emb, const float* day_emb, const float* wide_w,
                                const float* wide_b, float* pe, float* catvec, float* deep_in, int32_t B,
                                int32_t* err_flag, void* stream) {
  TDX_CHECK_ARG(time && cat && num && pos_enc && cell_emb && day_emb && wide_w && wide_b && pe && catvec && deep_in && err_flag,
                "tdx_embed_gather: null pointer");
  TDX_CHECK_ARG(B > 0 && max_time > 0, "tdx_embed_gather: bad B=%d max_time=%d", B, max_time);
  hipLaunchKernelGGL(embed_gather_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, time, cat, num, pos_enc, max_time,
                     cell_emb, day_emb, wide_w, wide_b, pe, catvec, deep_in, B, err_flag);
  TDX_CHECK_LAUNCH("embed_gather_kernel");
  return TDX_OK;
}

extern "C" int tdx_ddpm_update(float* x, const float* eps, const float* z, const float* coef, int64_t n, uint64_t seed,
                               const int32_t* step_ctr, int64_t first_elem, void* stream) {
  TDX_CHECK_ARG(x && eps && coef, "tdx_ddpm_update: null pointer");
  TDX_CHECK_ARG(n > 0, "tdx_ddpm_update: n=%lld", (long long)n);
  TDX_CHECK_ARG(first_elem % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)eps % 16 == 0),
                "tdx_ddpm_update: x/eps must be 16-byte aligned and first_elem a multiple of 4");
  const long n4 = (n + 3) / 4;
  hipLaunchKernelGGL(ddpm_update_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, eps, z, coef,
                     n4, (long)n, seed, step_ctr, (long)first_elem);
  TDX_CHECK_LAUNCH("ddpm_update_kernel");
  return TDX_OK;
}

extern "C" int tdx_ddim_update(float* x, const float* eps, const float* coef, int64_t n, void* stream) {
  TDX_CHECK_ARG(x && eps && coef && n > 0, "tdx_ddim_update: bad arguments");
  hipLaunchKernelGGL(ddim_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, eps, coef, (long)n);
  TDX_CHECK_LAUNCH("ddim_update_kernel");
  return TDX_OK;
}

extern "C" int tdx_step_prepare(int32_t* step_ctr, const float* coef_table, int32_t ncoef, float* coef_cur,
                                const float* tbias_table, int32_t ntb, float* tbias_cur, const int32_t* step_to_t,
                                int64_t* t_vec, int32_t B, int32_t nsteps, void* stream) {
  TDX_CHECK_ARG(step_ctr && coef_table && coef_cur && ncoef > 0 && nsteps > 0, "tdx_step_prepare: bad arguments");
  TDX_CHECK_ARG(!tbias_table || (tbias_cur && ntb > 0), "tdx_step_prepare: tbias_table needs tbias_cur/ntb");
  TDX_CHECK_ARG(!t_vec || (step_to_t && B > 0), "tdx_step_prepare: t_vec needs step_to_t and B");
  long work = ncoef;
  if (tbias_table && ntb > work) work = ntb;
  if (t_vec && B > work) work = B;
  unsigned blocks = (unsigned)((work + 255) / 256);
  if (blocks > 256) blocks = 256;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(step_prepare_kernel, dim3(blocks), dim3(256), 0, st, step_ctr, coef_table, ncoef, coef_cur, tbias_table, ntb,
                     tbias_cur, step_to_t, t_vec, B, nsteps);
  TDX_CHECK_LAUNCH("step_prepare_kernel");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, st, step_ctr, step_to_t, nsteps);
  TDX_CHECK_LAUNCH("step_advance_kernel");
  return TDX_OK;
}

extern "C" int tdx_q_sample(const float* x0, const float* eps, const int64_t* t, const float* sqrt_ab,
                            const float* sqrt_1m_ab, int32_t T, float* out, int32_t B, int32_t per_sample,
                            int32_t* err_flag, void* stream) {
  TDX_CHECK_ARG(x0 && eps && t && sqrt_ab && sqrt_1m_ab && out && err_flag, "tdx_q_sample: null pointer");
  TDX_CHECK_ARG(B > 0 && per_sample > 0 && T > 0, "tdx_q_sample: bad sizes");
  const long n = (long)B * per_sample;
  hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x0, eps, t, sqrt_ab,
                     sqrt_1m_ab, T, out, B, per_sample, err_flag);
  TDX_CHECK_LAUNCH("q_sample_kernel");
  return TDX_OK;
}
