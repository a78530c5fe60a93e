// Kernels of the latent path: SDUNet_CAR's cross-attention to the 7 condition tokens and its
// condition embedding, and the small layout / resampling / attention ops of VAE_InvTrans.  The GEMM-shaped work of both models
// (every Conv1d, nn.Linear and attention projection) runs on tdx_conv_gemm; what is here is bound by HBM or by launch latency.
#include "tdx_common.h"

namespace {

// ---- cross-attention core (Models/StableDiffusionBlocks.py:74, nn.MultiheadAttention(out_dim, 8) with 7 key/value tokens) ----
// HBM-bound: q in, out out, K | V of the trajectory once.  A lane owns one 16-byte channel vector of a query row, so a row takes
// LPR = C / VEC lanes and a head (C / 8 channels) LPR / 8 of them (2 .. 32, inside one wave): the N scores of a head are partial
// dot products reduced with xor-shuffles; softmax and the weighted sum of V stay in registers.  A workgroup serves TB whole
// trajectories (TB * L rows, 256 / LPR rows per pass); their K | V rows (TB x N x 2C) are staged once in LDS in the storage
// dtype with 16-byte loads.  kv_bs = element stride between trajectories (the K | V rows of all blocks of the model come out of
// one GEMM as column slices of one wide matrix).
template <typename T>
__global__ __launch_bounds__(256) void cross_attention_kernel(const T* __restrict__ q, int q_ld, const T* __restrict__ kv, int kv_ld, long kv_bs,
                                                             T* __restrict__ out, int out_ld, int B, int L, int N, int C, int TB, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kv_raw[];
  constexpr int VEC = Elem<T>::kVec;
  T* kvs = reinterpret_cast<T*>(kv_raw);   // [TB][N][2C]
  const int b0 = blockIdx.x * TB;
  const int ntb = min(TB, B - b0);
  const int vpt = 2 * C / VEC;             // 16-byte vectors per token row
  // staging in batches of 8 vectors per thread: all loads of a batch are in flight before the first LDS store (one load, one wait,
  // one store per trip would cost a memory round trip per 4 KB of a 28 KB working set)
  const int nvec = ntb * N * vpt;
  for (int i0 = threadIdx.x; i0 < nvec; i0 += 256 * 8) {
    u32x4 held[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * 256;
      const int ic = i < nvec ? i : nvec - 1;
      const int v = ic % vpt, tn = ic / vpt;
      const int n = tn % N, tb = tn / N;
      held[k] = *reinterpret_cast<const u32x4*>(kv + (size_t)(b0 + tb) * kv_bs + (size_t)n * kv_ld + v * VEC);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * 256;
      if (i < nvec) *reinterpret_cast<u32x4*>(kvs + (size_t)i * VEC) = held[k];
    }
  }
  const int lpr = C / VEC, lph = lpr >> 3, rpp = 256 / lpr;
  const int col = threadIdx.x % lpr, rip = threadIdx.x / lpr;
  const int rows = ntb * L;
  // the query vector of a pass is requested one pass ahead (the first one before the K | V staging is waited for): a pass is a
  // dependent chain load -> 7 dots -> softmax -> 7 fmas -> store, and L = 3 .. 24 rows give a workgroup only 1 .. 3 passes
  auto q_load = [&](int r0) -> u32x4 {
    const int r = r0 + rip;
    const int rc = r < rows ? r : rows - 1;
    const int tb = rc / L, l = rc - tb * L;
    return *reinterpret_cast<const u32x4*>(q + ((size_t)(b0 + tb) * L + l) * q_ld + col * VEC);
  };
  u32x4 q_next = q_load(0);
  __syncthreads();
  for (int r0 = 0; r0 < rows; r0 += rpp) {
    const int r = r0 + rip;
    const bool live = r < rows;
    const int rc = live ? r : rows - 1;    // dead lanes shadow the last row: the shuffles below stay convergent
    const int tb = rc / L, l = rc - tb * L;
    const size_t grow = (size_t)(b0 + tb) * L + l;
    float qv[VEC];
    Elem<T>::unpack(q_next, qv);
    if (r0 + rpp < rows) q_next = q_load(r0 + rpp);
    const T* kt = kvs + (size_t)tb * N * 2 * C + col * VEC;
    float s[8];
    float m = -3.0e38f;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      s[n] = -3.0e38f;
      if (n < N) {
        float kf[VEC];
        Elem<T>::unpack(*reinterpret_cast<const u32x4*>(kt + (size_t)n * 2 * C), kf);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) d = fmaf(qv[j], kf[j], d);
        for (int o = 1; o < lph; o <<= 1) d += __shfl_xor(d, o, 64);
        s[n] = d * scale;
        m = fmaxf(m, s[n]);
      }
    }
    float den = 0.f;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      s[n] = n < N ? __expf(s[n] - m) : 0.f;
      den += s[n];
    }
    const float inv = 1.0f / den;
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = 0.f;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      if (n < N) {
        float vf[VEC];
        Elem<T>::unpack(*reinterpret_cast<const u32x4*>(kt + (size_t)n * 2 * C + C), vf);
        const float pw = s[n] * inv;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = fmaf(pw, vf[j], o[j]);
      }
    }
    if (live) *reinterpret_cast<u32x4*>(out + grow * out_ld + col * VEC) = Elem<T>::pack(o);
  }
}

// ---- small multi-head self-attention (VAE_InvTrans's InvTransformerBlock, Models/VAEBlocks.py:103-121: tokens = channels,
//      embedding = sequence positions), fp32.  qkv (B, T, 3E) [q | k | v] -> out (B, T, E).  One workgroup per trajectory, K and V in
//      LDS, one thread per (token, head) walks the T keys with an online softmax. -----------------------------------------------
__global__ __launch_bounds__(256) void mha_small_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B, int T, int E, int heads,
                                                       float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // K [T][E] | V [T][E]
  const int b = blockIdx.x;
  float* ks = sm;
  float* vs = sm + T * E;
  for (int i = threadIdx.x; i < T * E; i += 256) {
    const int t = i / E, e = i - t * E;
    ks[i] = qkv[((size_t)b * T + t) * 3 * E + E + e];
    vs[i] = qkv[((size_t)b * T + t) * 3 * E + 2 * E + e];
  }
  __syncthreads();
  const int hd = E / heads;  // <= 16
  for (int w = threadIdx.x; w < T * heads; w += 256) {
    const int t = w / heads, h = w - t * heads;
    float qv[16], o[16];
    for (int j = 0; j < hd; ++j) {
      qv[j] = qkv[((size_t)b * T + t) * 3 * E + h * hd + j] * scale;
      o[j] = 0.f;
    }
    float m = -3.0e38f, den = 0.f;
    for (int n = 0; n < T; ++n) {
      float d = 0.f;
      for (int j = 0; j < hd; ++j) d = fmaf(qv[j], ks[n * E + h * hd + j], d);
      const float mn = fmaxf(m, d);
      const float corr = __expf(m - mn), pw = __expf(d - mn);
      den = den * corr + pw;
      for (int j = 0; j < hd; ++j) o[j] = fmaf(pw, vs[n * E + h * hd + j], o[j] * corr);
      m = mn;
    }
    const float inv = 1.0f / den;
    for (int j = 0; j < hd; ++j) out[((size_t)b * T + t) * E + h * hd + j] = o[j] * inv;
  }
}

// ---- layout: (B, C, L) fp32 channels-first <-> (B, L, ld) rows -------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void chanfirst_to_rows_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int C, int L, int ld) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * L * ld) return;
  const int c = (int)(i % ld);
  const long bl = i / ld;
  const int l = (int)(bl % L), b = (int)(bl / L);
  out[i] = (T)(c < C ? x[((size_t)b * C + c) * L + l] : 0.f);   // channels C .. ld-1 are zero padding
}
template <typename T>
__global__ __launch_bounds__(256) void rows_to_chanfirst_kernel(const T* __restrict__ rows, int ld, float* __restrict__ out, int B, int C, int L) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * C * L) return;
  const int l = (int)(i % L);
  const long bc = i / L;
  const int c = (int)(bc % C), b = (int)(bc / C);
  out[i] = Elem<T>::ld(rows + ((size_t)b * L + l) * ld + c);
}

// ---- resampling along the trajectory, rows (B, L, C) ------------------------------------------------------------------------------
// mode 0: nearest, src = min(floor(dst * (float)lin / lout), lin - 1) (nn.Upsample(scale_factor=2, mode='nearest'), Models/
//         StableDiffusion.py:163; ATen's index formula); mode 1: linear with align_corners=True (nn.Upsample(mode='linear',
//         align_corners=True), Models/VAE.py:137-168); mode 2: F.adaptive_avg_pool1d (Models/VAE.py:181,191).
template <typename T>
__global__ __launch_bounds__(256) void resample_rows_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int lin, int lout, int C, int mode,
                                                           float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * lout * C) return;
  const int c = (int)(i % C);
  const long bl = i / C;
  const int l = (int)(bl % lout), b = (int)(bl / lout);
  const T* xb = x + (size_t)b * lin * C + c;
  float v;
  if (mode == 0) {
    const int s = min((int)floorf((float)l * scale), lin - 1);
    v = Elem<T>::ld(xb + (size_t)s * C);
  } else if (mode == 1) {
    // ATen area_pixel_compute_source_index, align_corners: src = dst * (lin - 1) / (lout - 1)
    const float src = lout > 1 ? (float)l * scale : 0.f;
    const int i0 = (int)src;
    const int i1 = i0 + (i0 < lin - 1 ? 1 : 0);
    const float w1 = src - (float)i0, w0 = 1.0f - w1;
    v = w0 * Elem<T>::ld(xb + (size_t)i0 * C) + w1 * Elem<T>::ld(xb + (size_t)i1 * C);
  } else {
    const int s = (int)(((long)l * lin) / lout), e = (int)((((long)(l + 1)) * lin + lout - 1) / lout);
    float acc = 0.f;
    for (int k = s; k < e; ++k) acc += Elem<T>::ld(xb + (size_t)k * C);
    v = acc / (float)(e - s);
  }
  out[i] = (T)v;
}

// ---- ConditionMixBlock input gather (Models/StableDiffusionBlocks.py:131-140), fp32, hidden_dim = H ------------------------------
//   pe (B, H) = pos_enc[time];  catvec (B, 5H) = [cell_emb[cat0..3] | day_emb[cat4]] (optional raw copy, may be null);
//   mix_in (B, 7H) columns H..6H-1 = leaky0.1(catvec): the reference's forward concatenates the gathered embeddings as they are
//   (:136,140); `categorical_fc` (:109) is constructed but never called there.  Columns 6H..7H-1 = leaky0.1(numerical_fc(num));
//   columns 0..H-1 are written by the time GEMM.  The LeakyReLU that opens mix_fc (:116) acts on the whole concatenation, so each
//   producer applies it to its own slice.
__global__ __launch_bounds__(128) void condmix_gather_kernel(const int64_t* __restrict__ time, const int64_t* __restrict__ cat, const float* __restrict__ num,
                                                            const float* __restrict__ pos_enc, int max_time, const float* __restrict__ cell_emb,
                                                            const float* __restrict__ day_emb, const float* __restrict__ num_w, const float* __restrict__ num_b,
                                                            float* __restrict__ pe, float* __restrict__ catvec, float* __restrict__ mix_in, int H, int B,
                                                            int* __restrict__ err_flag) {
  const int b = blockIdx.x;
  long t = time[b];
  if (t < 0 || t >= max_time) {
    if (threadIdx.x == 0) atomicOr(err_flag, 1);
    t = 0;
  }
  long ci[5];
  for (int k = 0; k < 5; ++k) {
    ci[k] = cat[(size_t)b * 5 + k];
    const long lim = k < 4 ? 256 : 33;
    if (ci[k] < 0 || ci[k] >= lim) {
      if (threadIdx.x == 0) atomicOr(err_flag, 2);
      ci[k] = 0;
    }
  }
  const float n0 = num[(size_t)b * 3], n1 = num[(size_t)b * 3 + 1], n2 = num[(size_t)b * 3 + 2];
  for (int j = threadIdx.x; j < H; j += 128) {
    pe[(size_t)b * H + j] = pos_enc[(size_t)t * H + j];
    for (int k = 0; k < 5; ++k) {
      const float e = k < 4 ? cell_emb[(size_t)ci[k] * H + j] : day_emb[(size_t)ci[4] * H + j];
      if (catvec) catvec[(size_t)b * 5 * H + k * H + j] = e;
      mix_in[(size_t)b * 7 * H + (size_t)(k + 1) * H + j] = e > 0.f ? e : 0.1f * e;
    }
    // F.linear: x @ W^T + b, W (H, 3); accumulate in the order ATen's addmm does for K = 3 (bias first)
    const float v = ((num_b[j] + n0 * num_w[j * 3]) + n1 * num_w[j * 3 + 1]) + n2 * num_w[j * 3 + 2];
    mix_in[(size_t)b * 7 * H + 6 * H + j] = v > 0.f ? v : 0.1f * v;
  }
}

// ---- latent stem / head: Conv1d(k = 3, p = 1) with a handful of channels on one side (Models/StableDiffusion.py:127,150) ----------
// stem: x (B, CIN, L) fp32 channels-first -> rows (B, L, C0); one thread per (row, channel), CIN <= 8.
template <typename T>
__global__ __launch_bounds__(256) void stem_conv_n_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ out, int B, int L, int CIN, int C0) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * L * C0) return;
  const int c = (int)(i % C0);
  const long bl = i / C0;
  const int l = (int)(bl % L), b = (int)(bl / L);
  float a = bias[c];
  for (int ci = 0; ci < CIN; ++ci)
    for (int t = 0; t < 3; ++t) {
      const int u = l + t - 1;
      const float xv = (u >= 0 && u < L) ? x[((size_t)b * CIN + ci) * L + u] : 0.f;
      a = fmaf(w[(c * CIN + ci) * 3 + t], xv, a);
    }
  out[i] = (T)a;
}
// head: rows (B, L, C0) -> (B, COUT, L) fp32 channels-first; one wave per output row, lanes stride the channels, COUT <= 8.
template <typename T>
__global__ __launch_bounds__(256) void head_conv_n_kernel(const T* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int B, int L, int C0, int COUT) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * L) return;   // wave-uniform
  const int l = (int)(row % L), b = (int)(row / L);
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = 0.f;
  for (int c = lane; c < C0; c += 64)
    for (int t = 0; t < 3; ++t) {
      const int u = l + t - 1;
      if (u < 0 || u >= L) continue;
      const float v = Elem<T>::ld(a + ((size_t)b * L + u) * C0 + c);
#pragma unroll
      for (int o = 0; o < 8; ++o)
        if (o < COUT) acc[o] = fmaf(w[((size_t)o * C0 + c) * 3 + t], v, acc[o]);
    }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    if (o >= COUT) break;
    const float s = wave_sum(acc[o]);
    if (lane == 0) out[((size_t)b * COUT + o) * L + l] = s + bias[o];
  }
}

// Register-stationary forms of the two kernels above (C0 a multiple of the 16-byte vector): a lane owns one channel vector, keeps
// its weights in registers and walks `iters` rows; a per-thread weight reload per output (the simple kernels) is bound by
// vector-memory instruction issue, 20x off the HBM time.
template <typename T, int CIN>
__global__ __launch_bounds__(256) void stem_conv_n_vec_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             T* __restrict__ out, int B, int L, int C0, int iters) {
  constexpr int VEC = Elem<T>::kVec;
  const int vpr = C0 / VEC;                    // host: a power of two <= 256
  const int cv = threadIdx.x % vpr, rstep = 256 / vpr;
  float wr[VEC][CIN * 3], bs[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    bs[j] = bias[cv * VEC + j];
#pragma unroll
    for (int k = 0; k < CIN * 3; ++k) wr[j][k] = w[(cv * VEC + j) * CIN * 3 + k];
  }
  const long nrows = (long)B * L;
  const long r0 = (long)blockIdx.x * rstep * iters + threadIdx.x / vpr;
  for (int it = 0; it < iters; ++it) {
    const long n = r0 + (long)it * rstep;
    if (n >= nrows) break;
    const int l = (int)(n % L);
    const long b = n / L;
    float xin[CIN * 3];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int u = l + t - 1;
        xin[ci * 3 + t] = (u >= 0 && u < L) ? x[((size_t)b * CIN + ci) * L + u] : 0.f;
      }
    float v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float a = bs[j];
#pragma unroll
      for (int k = 0; k < CIN * 3; ++k) a = fmaf(wr[j][k], xin[k], a);   // same order as stem_conv_n_kernel
      v[j] = a;
    }
    *reinterpret_cast<u32x4*>(out + (size_t)n * C0 + cv * VEC) = Elem<T>::pack(v);
  }
}
// lpr = C0 / VEC lanes (a power of two <= 64) share a row; sums reduced with xor-shuffles inside the row's lanes.
template <typename T, int COUT>
__global__ __launch_bounds__(256) void head_conv_n_vec_kernel(const T* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ out, int B, int L, int C0, int lpr, int iters) {
  constexpr int VEC = Elem<T>::kVec;
  const int col = threadIdx.x % lpr, rstep = 256 / lpr;
  float wr[COUT][3][VEC];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < VEC; ++j) wr[o][t][j] = w[((size_t)o * C0 + col * VEC + j) * 3 + t];
  const long nrows = (long)B * L;
  const long r0 = (long)blockIdx.x * rstep * iters + threadIdx.x / lpr;
  for (int it = 0; it < iters; ++it) {
    const long n = r0 + (long)it * rstep;
    const bool live = n < nrows;   // dead lanes shadow the last row: the shuffles stay convergent
    const long nn = live ? n : nrows - 1;
    const int l = (int)(nn % L);
    const long b = nn / L;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int u = l + t - 1;
      const bool ok = u >= 0 && u < L;
      float v[VEC];
      Elem<T>::unpack(ok ? *reinterpret_cast<const u32x4*>(a + ((size_t)b * L + u) * C0 + col * VEC) : (u32x4){0u, 0u, 0u, 0u}, v);
#pragma unroll
      for (int o = 0; o < COUT; ++o)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[o] = fmaf(wr[o][t][j], v[j], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      float s = acc[o];
      for (int d = 1; d < lpr; d <<= 1) s += __shfl_xor(s, d, 64);
      if (live && col == 0) out[((size_t)b * COUT + o) * L + l] = s + bias[o];
    }
  }
}
template <typename T>
static bool launch_stem_vec(const float* x, const float* w, const float* bias, T* out, int B, int L, int cin, int C0, hipStream_t st) {
  const int vpr = C0 / Elem<T>::kVec, iters = 8;
  const long rows = (long)B * L, per_block = (long)(256 / vpr) * iters;
  const dim3 grid((unsigned)((rows + per_block - 1) / per_block));
  switch (cin) {
    case 1: hipLaunchKernelGGL((stem_conv_n_vec_kernel<T, 1>), grid, dim3(256), 0, st, x, w, bias, out, B, L, C0, iters); return true;
    case 2: hipLaunchKernelGGL((stem_conv_n_vec_kernel<T, 2>), grid, dim3(256), 0, st, x, w, bias, out, B, L, C0, iters); return true;
    case 3: hipLaunchKernelGGL((stem_conv_n_vec_kernel<T, 3>), grid, dim3(256), 0, st, x, w, bias, out, B, L, C0, iters); return true;
    case 4: hipLaunchKernelGGL((stem_conv_n_vec_kernel<T, 4>), grid, dim3(256), 0, st, x, w, bias, out, B, L, C0, iters); return true;
    default: return false;
  }
}
template <typename T>
static bool launch_head_vec(const T* a, const float* w, const float* bias, float* out, int B, int L, int C0, int cout, int lpr, hipStream_t st) {
  const int iters = 8;
  const long rows = (long)B * L, per_block = (long)(256 / lpr) * iters;
  const dim3 grid((unsigned)((rows + per_block - 1) / per_block));
  switch (cout) {
    case 1: hipLaunchKernelGGL((head_conv_n_vec_kernel<T, 1>), grid, dim3(256), 0, st, a, w, bias, out, B, L, C0, lpr, iters); return true;
    case 2: hipLaunchKernelGGL((head_conv_n_vec_kernel<T, 2>), grid, dim3(256), 0, st, a, w, bias, out, B, L, C0, lpr, iters); return true;
    case 3: hipLaunchKernelGGL((head_conv_n_vec_kernel<T, 3>), grid, dim3(256), 0, st, a, w, bias, out, B, L, C0, lpr, iters); return true;
    case 4: hipLaunchKernelGGL((head_conv_n_vec_kernel<T, 4>), grid, dim3(256), 0, st, a, w, bias, out, B, L, C0, lpr, iters); return true;
    default: return false;
  }
}

}  // namespace

extern "C" int tdx_stem_conv_n(const float* x, const float* w, const float* bias, void* out, int32_t B, int32_t L, int32_t cin, int32_t C0,
                               int32_t dtype, void* stream) {
  TDX_CHECK_ARG(x && w && bias && out, "tdx_stem_conv_n: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_stem_conv_n: bad dtype %d", dtype);
  TDX_CHECK_ARG(B > 0 && L > 0 && C0 > 0 && cin >= 1 && cin <= 8, "tdx_stem_conv_n: bad shape B=%d L=%d cin=%d C0=%d (cin 1..8)", B, L, cin, C0);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  const int vpr = C0 % vec == 0 ? C0 / vec : 0;
  if (vpr >= 1 && vpr <= 256 && (vpr & (vpr - 1)) == 0 && cin <= 4 && ((uintptr_t)out & 15) == 0) {
    const bool ok = dtype == TDX_F32 ? launch_stem_vec<float>(x, w, bias, (float*)out, B, L, cin, C0, (hipStream_t)stream)
                                     : launch_stem_vec<__bf16>(x, w, bias, (__bf16*)out, B, L, cin, C0, (hipStream_t)stream);
    if (ok) {
      TDX_CHECK_LAUNCH("stem_conv_n_vec_kernel");
      return TDX_OK;
    }
  }
  const long n = (long)B * L * C0;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == TDX_F32) hipLaunchKernelGGL(stem_conv_n_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, (float*)out, B, L, cin, C0);
  else hipLaunchKernelGGL(stem_conv_n_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, (__bf16*)out, B, L, cin, C0);
  TDX_CHECK_LAUNCH("stem_conv_n_kernel");
  return TDX_OK;
}

extern "C" int tdx_head_conv_n(const void* a, const float* w, const float* bias, float* out, int32_t B, int32_t L, int32_t C0, int32_t cout,
                               int32_t dtype, void* stream) {
  TDX_CHECK_ARG(a && w && bias && out, "tdx_head_conv_n: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_head_conv_n: bad dtype %d", dtype);
  TDX_CHECK_ARG(B > 0 && L > 0 && C0 > 0 && cout >= 1 && cout <= 8, "tdx_head_conv_n: bad shape B=%d L=%d C0=%d cout=%d (cout 1..8)", B, L, C0, cout);
  const long rows = (long)B * L;
  const int vec = dtype == TDX_F32 ? 4 : 8;
  const int lpr = C0 % vec == 0 ? C0 / vec : 0;
  if (lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && cout <= 4 && ((uintptr_t)a & 15) == 0) {
    const bool ok = dtype == TDX_F32 ? launch_head_vec<float>((const float*)a, w, bias, out, B, L, C0, cout, lpr, (hipStream_t)stream)
                                     : launch_head_vec<__bf16>((const __bf16*)a, w, bias, out, B, L, C0, cout, lpr, (hipStream_t)stream);
    if (ok) {
      TDX_CHECK_LAUNCH("head_conv_n_vec_kernel");
      return TDX_OK;
    }
  }
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == TDX_F32) hipLaunchKernelGGL(head_conv_n_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)a, w, bias, out, B, L, C0, cout);
  else hipLaunchKernelGGL(head_conv_n_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)a, w, bias, out, B, L, C0, cout);
  TDX_CHECK_LAUNCH("head_conv_n_kernel");
  return TDX_OK;
}

extern "C" int tdx_cross_attention(const void* q, int32_t q_ld, const void* kv, int32_t kv_ld, int64_t kv_bs, void* out, int32_t out_ld, int32_t B,
                                   int32_t L, int32_t N, int32_t C, int32_t heads, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(q && kv && out && B > 0 && L > 0, "tdx_cross_attention: bad arguments");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_cross_attention: bad dtype %d", dtype);
  TDX_CHECK_ARG(N >= 1 && N <= 8, "tdx_cross_attention: N=%d key/value tokens (1..8)", N);
  TDX_CHECK_ARG(heads == 8, "tdx_cross_attention: heads=%d (the lane layout is built for 8 heads, StableDiffusion.py:139)", heads);
  TDX_CHECK_ARG(C == 128 || C == 256 || C == 512 || C == 1024, "tdx_cross_attention: C=%d (128, 256, 512 or 1024)", C);
  const int es = dtype == TDX_F32 ? 4 : 2, vec = 16 / es;
  TDX_CHECK_ARG(q_ld >= C && out_ld >= C && kv_ld >= 2 * C && kv_bs >= (int64_t)N * kv_ld, "tdx_cross_attention: row pitches too small");
  TDX_CHECK_ARG(q_ld % vec == 0 && out_ld % vec == 0 && kv_ld % vec == 0 && kv_bs % vec == 0 && (((uintptr_t)q | (uintptr_t)kv | (uintptr_t)out) & 15) == 0,
                "tdx_cross_attention: pitches and pointers must be 16-byte aligned");
  // trajectories per workgroup: fill whole passes of 256 / (C / vec) rows, within 32 KB of LDS (one trajectory may need up to 56 KB)
  const int rpp = 256 / (C / vec);
  int g = L, r = rpp;
  while (r) { const int t = g % r; g = r; r = t; }   // gcd(L, rpp)
  int tb = rpp / g;
  const size_t per_traj = (size_t)N * 2 * C * es;
  while (tb > 1 && tb * per_traj > 32 * 1024) tb >>= 1;   // (more resident workgroups hide the staging latency better than fuller passes)
  if (tb > B) tb = B;
  const size_t lds = tb * per_traj;
  const float scale = 1.0f / sqrtf((float)(C / heads));
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((B + tb - 1) / tb));
  if (dtype == TDX_F32) {
    static bool attr_dev[kTdxMaxDevices] = {};
    bool& attr = attr_dev[tdx_device_slot()];
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attention_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
        TDX_FAIL(TDX_ERR_HIP, "tdx_cross_attention: cannot reserve LDS");
      attr = true;
    }
    hipLaunchKernelGGL(cross_attention_kernel<float>, grid, dim3(256), lds, st, (const float*)q, q_ld, (const float*)kv, kv_ld, (long)kv_bs, (float*)out,
                       out_ld, B, L, N, C, tb, scale);
  } else {
    static bool attr_dev[kTdxMaxDevices] = {};
    bool& attr = attr_dev[tdx_device_slot()];
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attention_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
        TDX_FAIL(TDX_ERR_HIP, "tdx_cross_attention: cannot reserve LDS");
      attr = true;
    }
    hipLaunchKernelGGL(cross_attention_kernel<__bf16>, grid, dim3(256), lds, st, (const __bf16*)q, q_ld, (const __bf16*)kv, kv_ld, (long)kv_bs,
                       (__bf16*)out, out_ld, B, L, N, C, tb, scale);
  }
  TDX_CHECK_LAUNCH("cross_attention_kernel");
  return TDX_OK;
}

extern "C" int tdx_mha_small(const float* qkv, float* out, int32_t B, int32_t T, int32_t E, int32_t heads, void* stream) {
  TDX_CHECK_ARG(qkv && out && B > 0 && T > 0 && E > 0 && heads > 0 && E % heads == 0, "tdx_mha_small: bad arguments");
  TDX_CHECK_ARG(E / heads <= 16 && (size_t)T * E * 8 <= 64 * 1024, "tdx_mha_small: head dim <= 16 and T * E <= 8192 (got T=%d E=%d heads=%d)", T, E, heads);
  hipLaunchKernelGGL(mha_small_kernel, dim3((unsigned)B), dim3(256), (size_t)T * E * 8, (hipStream_t)stream, qkv, out, B, T, E, heads,
                     1.0f / sqrtf((float)(E / heads)));
  TDX_CHECK_LAUNCH("mha_small_kernel");
  return TDX_OK;
}

extern "C" int tdx_chanfirst_to_rows(const float* x, void* out, int32_t B, int32_t C, int32_t L, int32_t ld, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(x && out && B > 0 && C > 0 && L > 0 && ld >= C, "tdx_chanfirst_to_rows: bad arguments");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_chanfirst_to_rows: bad dtype %d", dtype);
  const long n = (long)B * L * ld;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == TDX_F32) hipLaunchKernelGGL(chanfirst_to_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (float*)out, B, C, L, ld);
  else hipLaunchKernelGGL(chanfirst_to_rows_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, x, (__bf16*)out, B, C, L, ld);
  TDX_CHECK_LAUNCH("chanfirst_to_rows_kernel");
  return TDX_OK;
}

extern "C" int tdx_rows_to_chanfirst(const void* rows, int32_t ld, float* out, int32_t B, int32_t C, int32_t L, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(rows && out && B > 0 && C > 0 && L > 0 && ld >= C, "tdx_rows_to_chanfirst: bad arguments");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_rows_to_chanfirst: bad dtype %d", dtype);
  const long n = (long)B * C * L;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == TDX_F32) hipLaunchKernelGGL(rows_to_chanfirst_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)rows, ld, out, B, C, L);
  else hipLaunchKernelGGL(rows_to_chanfirst_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)rows, ld, out, B, C, L);
  TDX_CHECK_LAUNCH("rows_to_chanfirst_kernel");
  return TDX_OK;
}

extern "C" int tdx_resample_rows(const void* x, void* out, int32_t B, int32_t lin, int32_t lout, int32_t C, int32_t mode, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(x && out && B > 0 && lin > 0 && lout > 0 && C > 0, "tdx_resample_rows: bad arguments");
  TDX_CHECK_ARG(mode >= 0 && mode <= 2, "tdx_resample_rows: mode=%d (0 nearest, 1 linear align_corners, 2 adaptive average)", mode);
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_resample_rows: bad dtype %d", dtype);
  // nearest: ATen's scale = (float)lin / lout; linear, align_corners: (lin - 1) / (lout - 1), both one fp32 division
  const float scale = mode == 0 ? (float)lin / (float)lout : (lout > 1 ? (float)(lin - 1) / (float)(lout - 1) : 0.f);
  const long n = (long)B * lout * C;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == TDX_F32)
    hipLaunchKernelGGL(resample_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)out, B, lin, lout, C, mode, scale);
  else
    hipLaunchKernelGGL(resample_rows_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)out, B, lin, lout, C, mode, scale);
  TDX_CHECK_LAUNCH("resample_rows_kernel");
  return TDX_OK;
}

extern "C" int tdx_condmix_gather(const int64_t* time, const int64_t* cat, const float* num, const float* pos_enc, int32_t max_time,
                                  const float* cell_emb, const float* day_emb, const float* num_w, const float* num_b, float* pe, float* catvec,
                                  float* mix_in, int32_t hidden, int32_t B, int32_t* err_flag, void* stream) {
  TDX_CHECK_ARG(time && cat && num && pos_enc && cell_emb && day_emb && num_w && num_b && pe && mix_in && err_flag,
                "tdx_condmix_gather: null pointer");
  TDX_CHECK_ARG(B > 0 && max_time > 0 && hidden > 0, "tdx_condmix_gather: bad sizes");
  hipLaunchKernelGGL(condmix_gather_kernel, dim3((unsigned)B), dim3(128), 0, (hipStream_t)stream, time, cat, num, pos_enc, max_time, cell_emb, day_emb,
                     num_w, num_b, pe, catvec, mix_in, hidden, B, err_flag);
  TDX_CHECK_LAUNCH("condmix_gather_kernel");
  return TDX_OK;
}
