// GroupNorm (+ Swish) over channels-last rows: one wavefront per (trajectory, channel window).
//
// A window is W = LPR * (16 B / sizeof(T)) consecutive channels (>= one whole group); the wave
// sweeps the trajectory's L rows, 64/LPR rows per iteration, 16 bytes per lane, and reduces
// with wavefront xor-shuffles (no atomics).  Statistics are exact two-pass
// (mean, then centred sum of squares) in fp32.
// Long trajectories (L > 64) are split over S = 2 or 4 waves of one workgroup, each holding its share of the rows in
// registers: the waves exchange their partial sums through LDS (two barriers) and add them in a fixed order.  Shorter
// per-wave sweeps need fewer registers (more waves per SIMD) and finer work items (no partly filled last round).
#include "tdx_common.h"

#ifndef TDX_GN_WIDE_L
#define TDX_GN_WIDE_L 64
#endif
namespace {

// NIT = row iterations per lane kept in registers (0: re-read the rows from L2 in each of the three sweeps).
// Training-mode nn.Dropout behind the activation (TrajUNetBlocks.py:83-85), DROP = true: the mask of tdx_dropout for the output tensor
// (Philox4x32-10 keyed by (seed, step_ctr[0], layer) and the element quad (row * C + channel) / 4), applied as the rows are written.
struct GnDrop {
  uint32_t thresh, layer;
  float scale;
  uint64_t seed;
  const int32_t* step_ctr;
};
template <typename T, int NIT, int S, bool DROP = false>
__global__ __launch_bounds__(256) void groupnorm_act_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ out, int out_ld,
                                                           int out_off, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int B, int L, int C, int cpg,
                                                           int lpr, float eps, int act, int rows_per_part, GnDrop drop = GnDrop{}) {
  constexpr int VEC = Elem<T>::kVec;
  constexpr int H = VEC / 2;
  constexpr int NR = NIT > 0 ? NIT : 1;
  static_assert(S == 1 || NIT > 0, "split sweeps keep their rows in registers");
  __shared__ float red[S > 1 ? 2 : 1][4][64][2];  // [pass][wave][column][lo, hi]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = lpr * VEC;
  const int wins = C / W;
  const long wid = ((long)blockIdx.x * 4 + wave) / S;  // (trajectory, window) item; S consecutive waves share one
  const int part = wave % S;
  const bool active = wid < (long)B * wins;            // wave-uniform, the same for the S waves of an item
  if (S == 1 && !active) return;                       // (split sweeps: idle waves still take part in the barriers)
  const int b = active ? (int)(wid / wins) : 0, win = active ? (int)(wid % wins) : 0;
  const int col = lane % lpr, rl = lane / lpr, rpi = 64 / lpr;
  const int ch = win * W + col * VEC;
  const T* xb = x + ((size_t)b * L) * x_ld + ch;
  const bool split = cpg < VEC;  // two groups inside one 16-byte vector (cpg == VEC/2)
  // this wave's rows: [l0, l1)
  const int l0 = S > 1 ? part * rows_per_part : 0;
  const int l1 = !active ? 0 : (S > 1 ? (l0 + rows_per_part < L ? l0 + rows_per_part : L) : L);

  int pass = 0;
  auto group_sum = [&](float& lo, float& hi) {
    for (int o = 32; o >= lpr; o >>= 1) {
      lo += __shfl_xor(lo, o, 64);
      hi += __shfl_xor(hi, o, 64);
    }
    if constexpr (S > 1) {  // the other waves' rows: partial sums through LDS, added in the order of the parts
      if (rl == 0) *reinterpret_cast<float2*>(&red[pass][wave][col][0]) = make_float2(lo, hi);
      __syncthreads();
      const int w0 = wave - part;
      lo = 0.f;
      hi = 0.f;
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const float2 v = *reinterpret_cast<const float2*>(&red[pass][w0 + k][col][0]);
        lo += v.x;
        hi += v.y;
      }
      ++pass;
    }
    if (!split) {
      lo += hi;
      for (int o = 1; o * VEC < cpg; o <<= 1) lo += __shfl_xor(lo, o, 64);
      hi = lo;
    }
  };

  u32x4 held[NR];
  if constexpr (NIT > 0) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int l = l0 + rl + i * rpi;
      held[i] = l < l1 ? *reinterpret_cast<const u32x4*>(xb + (size_t)l * x_ld) : (u32x4){0u, 0u, 0u, 0u};
    }
  }
  auto row = [&](int i, int l) -> u32x4 {
    if constexpr (NIT > 0) return held[i];
    else return *reinterpret_cast<const u32x4*>(xb + (size_t)l * x_ld);
  };
  const int nit = NIT > 0 ? NIT : (L - rl + rpi - 1) / rpi;  // (NIT == 0 only with S == 1: l0 = 0, l1 = L)

  const float inv_cnt = 1.0f / ((float)cpg * (float)L);
  float s_lo = 0.f, s_hi = 0.f;
#pragma unroll
  for (int i = 0; i < nit; ++i) {
    const int l = l0 + rl + i * rpi;
    if (l >= l1) break;
    float v[VEC];
    Elem<T>::unpack(row(i, l), v);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      s_lo += v[j];
      s_hi += v[H + j];
    }
  }
  group_sum(s_lo, s_hi);
  const float m_lo = s_lo * inv_cnt, m_hi = s_hi * inv_cnt;

  float q_lo = 0.f, q_hi = 0.f;
#pragma unroll
  for (int i = 0; i < nit; ++i) {
    const int l = l0 + rl + i * rpi;
    if (l >= l1) break;
    float v[VEC];
    Elem<T>::unpack(row(i, l), v);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      float a = v[j] - m_lo, c = v[H + j] - m_hi;
      q_lo += a * a;
      q_hi += c * c;
    }
  }
  group_sum(q_lo, q_hi);
  const float r_lo = rsqrtf(q_lo * inv_cnt + eps), r_hi = rsqrtf(q_hi * inv_cnt + eps);

  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float g = gamma[ch + j], bt = beta[ch + j];
    const float r = j < H ? r_lo : r_hi, m = j < H ? m_lo : m_hi;
    sc[j] = r * g;
    sh[j] = bt - m * r * g;
  }
  T* ob = out + ((size_t)b * L) * out_ld + out_off + ch;
#pragma unroll
  for (int i = 0; i < nit; ++i) {
    const int l = l0 + rl + i * rpi;
    if (l >= l1) break;
    float v[VEC];
    Elem<T>::unpack(row(i, l), v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = tdx_act(fmaf(v[j], sc[j], sh[j]), act);
    if constexpr (DROP) {
      static_assert(!DROP || VEC == 4, "fused dropout: fp32 rows (one Philox call per 4 channels)");
      uint32_t r[4];
      philox4((uint64_t)(((long)b * L + l) * C + ch) >> 2, drop.layer, drop.seed ^ ((uint64_t)(uint32_t)drop.step_ctr[0] << 32), r);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = r[j] >= drop.thresh ? v[j] * drop.scale : 0.f;
    }
    *reinterpret_cast<u32x4*>(ob + (size_t)l * out_ld) = Elem<T>::pack(v);
  }
}

template <typename T, bool DROP = false>
void launch_gn(int nit, int parts, int rows_per_part, dim3 grid, hipStream_t st, const void* x, int x_ld, void* out, int out_ld, int out_off,
               const float* gamma, const float* beta, int B, int L, int C, int cpg, int lpr, float eps, int act, GnDrop drop = GnDrop{}) {
#define TDX_GN(N, S)                                                                                                                    \
  hipLaunchKernelGGL((groupnorm_act_kernel<T, N, S, DROP>), grid, dim3(256), 0, st, (const T*)x, x_ld, (T*)out, out_ld, out_off, gamma, \
                     beta, B, L, C, cpg, lpr, eps, act, rows_per_part, drop)
  if (parts == 4) {
    if (nit <= 8) TDX_GN(8, 4);
    else TDX_GN(16, 4);
  } else if (parts == 2) {
    if (nit <= 8) TDX_GN(8, 2);
    else TDX_GN(16, 2);
  } else if (nit <= 2) TDX_GN(2, 1);
  else if (nit <= 4) TDX_GN(4, 1);
  else if (nit <= 8) TDX_GN(8, 1);
  else if (nit <= 16) TDX_GN(16, 1);
  else if (nit <= 26) TDX_GN(26, 1);
  else TDX_GN(0, 1);
#undef TDX_GN
}

// Split GroupNorm, second half: block partial sums -> {rstd, -mean * rstd} per (trajectory, group).  One thread per entry; the
// 32 groups of a (trajectory, block) row are adjacent in memory, so a wave reads whole 256-byte rows.
__global__ __launch_bounds__(256) void groupnorm_finalize_kernel(const float* __restrict__ stats, int stats_ld, int stats_off,
                                                                float* __restrict__ mr, int mr_ld, int mr_off, int B, int nblk,
                                                                int groups, int gpg, double inv_cnt, float eps) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)B * groups) return;
  const int b = (int)(e / groups), g = (int)(e - (long)b * groups);
  const float* row = stats + (((size_t)b * nblk) * stats_ld + stats_off + (size_t)g * gpg) * 2;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nblk; ++k)          // fixed order: blocks, then granules
    for (int q = 0; q < gpg; ++q) {
      const float2 v = *reinterpret_cast<const float2*>(row + ((size_t)k * stats_ld + q) * 2);
      s1 += (double)v.x;
      s2 += (double)v.y;
    }
  const double mean = s1 * inv_cnt;
  double var = s2 * inv_cnt - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  *reinterpret_cast<float2*>(mr + ((size_t)b * mr_ld + mr_off + g) * 2) = make_float2(rstd, (float)(-mean * (double)rstd));
}

}  // namespace

extern "C" int tdx_groupnorm_finalize(const float* stats, int32_t stats_ld, int32_t stats_off, float* mr, int32_t mr_ld, int32_t mr_off,
                                      int32_t B, int32_t L, int32_t groups, int32_t cpg, float eps, void* stream) {
  TDX_CHECK_ARG(stats && mr && B > 0 && L > 0 && groups > 0, "tdx_groupnorm_finalize: bad arguments");
  TDX_CHECK_ARG(cpg >= 4 && cpg % 4 == 0, "tdx_groupnorm_finalize: cpg=%d must be a multiple of 4 (statistics come in 4-channel granules)", cpg);
  const int gpg = cpg / 4;
  TDX_CHECK_ARG(stats_off >= 0 && stats_off + groups * gpg <= stats_ld && mr_off >= 0 && mr_off + groups <= mr_ld,
                "tdx_groupnorm_finalize: groups do not fit (stats_ld=%d off=%d, mr_ld=%d off=%d, groups=%d, cpg=%d)", stats_ld, stats_off, mr_ld,
                mr_off, groups, cpg);
  TDX_CHECK_ARG(((uintptr_t)stats & 7) == 0 && ((uintptr_t)mr & 7) == 0, "tdx_groupnorm_finalize: 8-byte alignment");
  const long n = (long)B * groups;
  hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats, stats_ld,
                     stats_off, mr, mr_ld, mr_off, B, (L + 15) / 16, groups, gpg, 1.0 / ((double)cpg * (double)L), eps);
  TDX_CHECK_LAUNCH("groupnorm_finalize_kernel");
  return TDX_OK;
}

static int groupnorm_act_impl(const void* x, int32_t x_ld, void* out, int32_t out_ld, int32_t out_off,
                              const float* gamma, const float* beta, int32_t B, int32_t L, int32_t C,
                              int32_t groups, float eps, int32_t act, int32_t dtype, const GnDrop* drop, void* stream) {
  TDX_CHECK_ARG(x && out && gamma && beta, "tdx_groupnorm_act: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_groupnorm_act: bad dtype %d", dtype);
  TDX_CHECK_ARG(B > 0 && L > 0 && C > 0 && groups > 0 && C % groups == 0, "tdx_groupnorm_act: bad shape B=%d L=%d C=%d groups=%d", B, L, C, groups);
  TDX_CHECK_ARG(act == TDX_ACT_NONE || act == TDX_ACT_SWISH || act == TDX_ACT_LEAKY, "tdx_groupnorm_act: bad act %d", act);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  const int cpg = C / groups;
  if (cpg < 4 || (cpg & (cpg - 1)) != 0)
    TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_groupnorm_act: channels per group = %d; need a power of two >= 4", cpg);
  int lpr = 8;
  while (lpr * vec < cpg) lpr <<= 1;
  if (lpr > 64 || C % (lpr * vec) != 0)
    TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_groupnorm_act: C=%d not a multiple of the %d-channel window", C, lpr * vec);
  // Short trajectories: eight rows per sweep leave lanes idle (3 of 8 row slots at L = 3, 13 of 16 at L = 13) and a wave moves
  // little.  Widen the window instead (fewer rows per sweep, longer contiguous row segments): the narrowest window with the
  // best lane use at no more than 8 row iterations.  A function of L and C alone.
  if (L < TDX_GN_WIDE_L) {
    int best = lpr, best_used = 0, best_slots = 1;
    for (int c = lpr; c <= 64 && C % (c * vec) == 0; c <<= 1) {
      const int r = 64 / c, slots = (L + r - 1) / r * r;
      if (slots / r > 8) break;                                                                           // (at most 8 row iterations)
      if ((long)L * best_slots > (long)best_used * slots) best = c, best_used = L, best_slots = slots;    // (L / slots > best: ties stay with the narrower window)
    }
    lpr = best;
  }
  TDX_CHECK_ARG(x_ld % vec == 0 && out_ld % vec == 0 && out_off % vec == 0 && x_ld >= C && out_off + C <= out_ld,
                "tdx_groupnorm_act: x_ld/out_ld/out_off must be multiples of %d and fit", vec);
  // Long trajectories: 2 or 4 waves per (trajectory, window), at most 16 row iterations each (a function of L and the
  // window alone, so a trajectory is normalised the same way in every batch it is part of).
  const int rpi = 64 / lpr;
  int parts = L > 128 ? 4 : (L > 64 ? 2 : 1);
  int rows_per_part = ((L + parts - 1) / parts + rpi - 1) / rpi * rpi;
  if (parts > 1 && rows_per_part / rpi > 16) parts = 1;   // (very long trajectories: the one-wave sweep that re-reads its rows)
  if (parts == 1) rows_per_part = L;
  const long waves = (long)B * (C / (lpr * vec)) * parts;
  dim3 grid((unsigned)((waves + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  const int nit = (rows_per_part + rpi - 1) / rpi;
  if (drop)
    launch_gn<float, true>(nit, parts, rows_per_part, grid, st, x, x_ld, out, out_ld, out_off, gamma, beta, B, L, C, cpg, lpr, eps, act, *drop);
  else if (dtype == TDX_F32)
    launch_gn<float>(nit, parts, rows_per_part, grid, st, x, x_ld, out, out_ld, out_off, gamma, beta, B, L, C, cpg, lpr, eps, act);
  else
    launch_gn<__bf16>(nit, parts, rows_per_part, grid, st, x, x_ld, out, out_ld, out_off, gamma, beta, B, L, C, cpg, lpr, eps, act);
  TDX_CHECK_LAUNCH("groupnorm_act_kernel");
  return TDX_OK;
}

extern "C" int tdx_groupnorm_act(const void* x, int32_t x_ld, void* out, int32_t out_ld, int32_t out_off,
                                 const float* gamma, const float* beta, int32_t B, int32_t L, int32_t C,
                                 int32_t groups, float eps, int32_t act, int32_t dtype, void* stream) {
  return groupnorm_act_impl(x, x_ld, out, out_ld, out_off, gamma, beta, B, L, C, groups, eps, act, dtype, nullptr, stream);
}

extern "C" int tdx_groupnorm_act_dropout(const void* x, int32_t x_ld, void* out, int32_t out_ld, int32_t out_off, const float* gamma,
                                         const float* beta, int32_t B, int32_t L, int32_t C, int32_t groups, float eps, int32_t act,
                                         int32_t dtype, float p, uint64_t seed, uint32_t layer, const int32_t* step_ctr, void* stream) {
  TDX_CHECK_ARG(dtype == TDX_F32, "tdx_groupnorm_act_dropout: fp32 rows only (dtype %d)", dtype);
  TDX_CHECK_ARG(step_ctr && p >= 0.f && p < 1.f, "tdx_groupnorm_act_dropout: p=%f not in [0, 1) or no step counter", (double)p);
  TDX_CHECK_ARG(out_off == 0 && out_ld == C, "tdx_groupnorm_act_dropout: the mask is keyed by the element index of a plain (B, L, C) output");
  const GnDrop d{(uint32_t)((double)p * 4294967296.0), layer, 1.0f / (1.0f - p), seed, step_ctr};
  return groupnorm_act_impl(x, x_ld, out, out_ld, out_off, gamma, beta, B, L, C, groups, eps, act, dtype, &d, stream);
}
