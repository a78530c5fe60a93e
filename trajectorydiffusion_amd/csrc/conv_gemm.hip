// Conv1d / Linear as ONE gathered GEMM on MFMA (gfx950).
//
//   out[n, co] = act(sum_k W[co, k] * X[n, k] + biases) + res[n, co]
//
// n runs over the B*lout output rows (channels-last), k over (segment, 128-byte channel block, tap,
// channel in block): taps are the INNER loop, so the three shifted reads of a conv hit rows that are
// already in L2/LDS.  The X
// operand is never materialised: for every (segment, tap) each staged tile row is fetched
// through its own source-row pointer (conv zero padding, stride 2, nearest-upsampling, the
// decoder's channel concat and the 1x1 shortcut are all just different pointers), and
// out-of-range rows/chunks point at a zero row, so the MFMA loop itself is a plain GEMM loop.
//
// MFMA orientation: A = weights (M = co), B = activations (N = n).  Then each lane's four
// accumulator registers are four consecutive output channels of one row, i.e. one 8/16-byte
// channels-last store.
#include "tdx_common.h"

#include <cstdarg>
#include <cstdio>

namespace {

__device__ __attribute__((aligned(256))) uint32_t g_zero_row[4096];  // 16 KB of zeros (never written)

struct SegDev {
  const char* src;
  float up_scale;  // > 0: nearest-upsampled source, srow = min(floorf(u * up_scale), lin - 1) (ATen's formula)
  int32_t src_ld, ch_off, cin, taps, stride, lin, lvirt, k0;  // k0: first packed-K index of this segment
};

struct ConvArgs {
  SegDev seg[4];
  int32_t nseg, nsteps;
  const char* w;
  int32_t ktot;
  const float* bias;
  const float* ebias;
  const float* tbias;
  int32_t ebias_ld, ebias_off;
  const char* res;
  int32_t res_ld, act;
  char* out;
  int32_t out_ld, out_off;
  int32_t N, lout, cout, B;
  int32_t tile_rows;  // rows per row tile of the DMA kernels: 256, or k * L when the GroupNorm epilogue needs whole trajectories
  int32_t n_begin, row_tiles;  // DMA kernels: this launch covers row tiles [0, row_tiles) counted from output row n_begin
  uint64_t div_m;              // n / lout == (n * div_m) >> div_k for 0 <= n < 2^31 (stream kernel: no integer divides per tile)
  int32_t div_k;
  const float* gn_gamma;
  const float* gn_beta;
  char* gn_out;
  int32_t gn_ld, gn_off, gn_act, gn_cpg, gn_L;
  float gn_eps;
  int32_t gn_tpe;  // persistent form: threads per (trajectory, group) sum
  int32_t debug;  // ablation bits for scripts/conv_bench.py: 1 = no loads after the first stage, 2 = no MFMA, 4 = no stores
  // split-K (register-staged kernel only): blockIdx.y = slice z computes K-steps [z * steps_per_split, ...) into partial[z][n][co]
  float* partial;
  int32_t ksplit, steps_per_split;
#if TDX_STAMPS
  uint32_t* stamps;
#endif
};
#if TDX_STAMPS
uint32_t* g_stamps = nullptr;
#endif

#ifndef TDX_SPREAD_DMA
#define TDX_SPREAD_DMA 1
#endif
#ifndef TDX_DIRECT_EPI
#define TDX_DIRECT_EPI 1
#endif
#ifndef TDX_STAMPS  // diagnostic build: per-phase cycle counts of wave 0 of every workgroup of the stream kernel (scripts/conv_stamps.py)
#define TDX_STAMPS 0
#endif
#ifndef TDX_ABL   // timing ablations of the direct epilogue (never in a product build): 1 = no epilogue, 2 = no stores, 4 = no staged rows
#define TDX_ABL 0
#endif
int g_force = 0;  // test hook, see tdx_conv_gemm_force_small
int g_tail_max = 128;  // largest remainder (in 256-row tiles) handed to the small-tile tail launch
int g_debug = 0;
constexpr int kRowBytes = 128;  // one K-step of one tile row: 64 bf16 / 32 fp32
constexpr int kThreads = 256;
constexpr int kEbMaxTraj = 24;  // embed-bias rows staged in LDS per tile (trajectories of >= 11 rows)

// Run f(integral_constant<int, tap>) for a wave-uniform runtime tap in {0, 1, 2}: every per-tap register array is then
// indexed by a compile-time constant (a runtime index, or a select between elements, sends the array to scratch).
template <int N>
struct TapC { static constexpr int value = N; };
template <typename F>
__device__ __forceinline__ void with_tap(int tap, F&& f) {
  if (tap == 0) f(TapC<0>{});
  else if (tap == 1) f(TapC<1>{});
  else f(TapC<2>{});
}

template <typename T>
struct Mma;
template <>
struct Mma<__bf16> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
  }
};


// Epilogue shared by both kernels: bias (+ per-sample / per-step embed bias) -> act -> (+ residual) -> store.
// Lane (fr, fq) of a 16x16 sub-tile (mi, ni) holds 4 consecutive channels co..co+3 of output row n.
// One output row n, four consecutive channels co..co+3, accumulated value v: bias (+ per-step / per-trajectory embed rows) ->
// act -> (+ residual) -> store.  Shared by the register-staged kernel's epilogue and the split-K reduction.
template <typename T>
__device__ __forceinline__ void finish4(const ConvArgs& p, int n, int b, int co, float (&v)[4]) {
  // same association as the LDS-DMA epilogues: acc + (bias + time-step row) + per-trajectory row
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] += t[j];
  }
  if (p.tbias) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p.tbias + p.ebias_off + co);
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] += t[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] += bs[j];
  if (p.ebias) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p.ebias + (size_t)b * p.ebias_ld + p.ebias_off + co);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += t[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = tdx_act(v[j], p.act);
  if constexpr (sizeof(T) == 4) {
    if (p.res) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p.res + ((size_t)n * p.res_ld + co) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += t[j];
    }
    *reinterpret_cast<f32x4*>(p.out + ((size_t)n * p.out_ld + p.out_off + co) * 4) = (f32x4){v[0], v[1], v[2], v[3]};
  } else {
    if (p.res) {
      const u32x2 t = *reinterpret_cast<const u32x2*>(p.res + ((size_t)n * p.res_ld + co) * 2);
      v[0] += __uint_as_float(t[0] << 16);
      v[1] += __uint_as_float(t[0] & 0xffff0000u);
      v[2] += __uint_as_float(t[1] << 16);
      v[3] += __uint_as_float(t[1] & 0xffff0000u);
    }
    u32x2 o;
    o[0] = (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)v[0]) |
           ((uint32_t)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16);
    o[1] = (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)v[2]) |
           ((uint32_t)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16);
    *reinterpret_cast<u32x2*>(p.out + ((size_t)n * p.out_ld + p.out_off + co) * 2) = o;
  }
}

template <typename T, int MI, int NI>
__device__ __forceinline__ void epilogue(const ConvArgs& p, f32x4 (&acc)[MI][NI], int n_base, int co_base, int fr, int fq) {
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n_base + ni * 16 + fr;
    if (n >= p.N) continue;
    const int b = n / p.lout;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int co = co_base + mi * 16 + fq * 4;
      if (co >= p.cout) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][j];
      finish4<T>(p, n, b, co, v);
    }
  }
}
// Split-K: slice z leaves its raw fp32 sums in partial[z][n][co]; splitk_finish_kernel adds the slices in order.
template <int MI, int NI>
__device__ __forceinline__ void epilogue_partial(const ConvArgs& p, f32x4 (&acc)[MI][NI], int n_base, int co_base, int fr, int fq, int z) {
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n_base + ni * 16 + fr;
    if (n >= p.N) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int co = co_base + mi * 16 + fq * 4;
      if (co >= p.cout) continue;
      *reinterpret_cast<f32x4*>(p.partial + ((size_t)z * p.N + n) * p.cout + co) = acc[mi][ni];
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const ConvArgs p) {
  const int c4 = p.cout >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)p.N * c4) return;
  const int n = (int)(i / c4), co = (int)(i - (long)n * c4) * 4;
  f32x4 s = *reinterpret_cast<const f32x4*>(p.partial + (size_t)n * p.cout + co);
  for (int z = 1; z < p.ksplit; ++z) s += *reinterpret_cast<const f32x4*>(p.partial + ((size_t)z * p.N + n) * p.cout + co);
  float v[4] = {s[0], s[1], s[2], s[3]};
  finish4<T>(p, n, n / p.lout, co, v);
}


// Epilogue of the DMA kernel: each wave transposes its accumulators through a private LDS patch, 16 rows at a
// time, so that global traffic is whole 16-byte vectors of contiguous row segments (bias / embed-bias / residual
// reads and the output stores are all coalesced).  WCO = channels per wave tile, TCH = channels per block tile.
//
// GN = true additionally applies the CONSUMER's GroupNorm (+ activation) to the finished rows and stores that
// too (or only that, when p.out is null).  It needs row tiles made of whole trajectories (p.tile_rows = k * L),
// so every (trajectory, group) statistic is complete inside the workgroup.  Statistics are exact two-pass
// (mean, then centred squares) and bitwise reproducible: per-row partial sums by xor-shuffles -> LDS table ->
// fixed-order tree over the L rows.  No atomics.
template <typename T, int MI, int NI, int WCO, int TCH, int TRW, bool GN>
__device__ __forceinline__ void epilogue_lds(const ConvArgs& p, f32x4 (&acc)[MI][NI], char* smem, int wave, int wr, int wc, int lane,
                                             int n0, int co0) {
  constexpr int VEC = 16 / (int)sizeof(T);      // output elements per 16-byte store
  constexpr int STRIDE = WCO * 4 + 16;          // fp32 patch row, padded: conflict-free ds_write_b128
  constexpr int PATCH = 16 * STRIDE;
  constexpr int NWV = TRW / 32;                 // waves per workgroup
  constexpr int NT = NWV * 64;
  constexpr int CPR = WCO / VEC;                // 16-byte output chunks per row
  constexpr int RPP = 64 / CPR;                 // rows per pass
  constexpr int NP = 16 / RPP;
  static_assert(CPR <= 64 && NP >= 1, "wave tile too wide");
  char* patch = smem + wave * PATCH;
  const int fr = lane & 15, fq = lane >> 4;
  const int c = lane % CPR, r0 = lane / CPR;
  const int co = co0 + wc * WCO + c * VEC;
  const bool co_ok = co < p.cout;
  const int n_end = n0 + p.tile_rows < p.N ? n0 + p.tile_rows : p.N;
  float bs[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) bs[j] = 0.f;
  if (co_ok) {
    if (p.bias)
#pragma unroll
      for (int j = 0; j < VEC; ++j) bs[j] += p.bias[co + j];
    if (p.tbias)
#pragma unroll
      for (int j = 0; j < VEC; ++j) bs[j] += p.tbias[p.ebias_off + co + j];
  }
  // Everything the epilogue reads from global memory is requested up front, so the 16 transpose passes below never
  // wait on a load of their own: residual vectors go to registers, the tile's embed-bias rows to LDS (one row per
  // trajectory touched by the tile), GroupNorm parameters to registers.
  u32x4 resv[NI][NP];
  if (p.res) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int n = n0 + wr * 64 + ni * 16 + r0 + RPP * i;
        resv[ni][i] = (n < n_end && co_ok) ? *reinterpret_cast<const u32x4*>(p.res + ((size_t)n * p.res_ld + co) * sizeof(T))
                                           : (u32x4){0u, 0u, 0u, 0u};
      }
  }
  float gm[VEC], bt[VEC];
  if constexpr (GN) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      gm[j] = co_ok ? p.gn_gamma[co + j] : 0.f;
      bt[j] = co_ok ? p.gn_beta[co + j] : 0.f;
    }
  }
  // GroupNorm bookkeeping (GN only)
  const int cpg = GN ? p.gn_cpg : VEC;
  const int GT = TCH / cpg;                          // groups per block tile
  const int LPG = cpg >= VEC ? cpg / VEC : 1;        // lanes that share one group within a row
  const bool two = cpg < VEC;                        // two groups inside one 16-byte vector (cpg == VEC / 2)
  const int g0 = (wc * WCO + c * VEC) / cpg;         // this lane's (first) group inside the block tile
  float* rs = reinterpret_cast<float*>(smem + NWV * PATCH);   // [tile rows][GT] per-row partial sums
  float* tab_mean = rs + TRW * GT;                            // [trajectories in tile][GT]
  float* tab_var = tab_mean + (TRW / p.gn_L + 1) * GT;
  // [trajectories touched][TCH] staged embed-bias rows
  float* eb_lds = GN ? tab_var + (TRW / p.gn_L + 1) * GT : reinterpret_cast<float*>(smem + NWV * PATCH);
  const int b_first = n0 / p.lout;
  const int ntraj = (n_end - 1) / p.lout - b_first + 1;
  const bool eb_staged = p.ebias && ntraj <= kEbMaxTraj;
  constexpr int EBR = (kEbMaxTraj * (TCH / 4) + NT - 1) / NT;  // float4 slots per thread
  f32x4 ebv[EBR];
  const int tid_e = wave * 64 + lane;
  if (eb_staged) {
#pragma unroll
    for (int k = 0; k < EBR; ++k) {
      const int idx = tid_e + k * NT;
      const int t = idx / (TCH / 4), q4 = idx - t * (TCH / 4);
      const int cc = co0 + q4 * 4;
      ebv[k] = (t < ntraj && cc < p.cout) ? *reinterpret_cast<const f32x4*>(p.ebias + (size_t)(b_first + t) * p.ebias_ld + p.ebias_off + cc)
                                          : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();  // every wave is done reading the stage buffers: from here on LDS is epilogue scratch
  if (eb_staged) {
#pragma unroll
    for (int k = 0; k < EBR; ++k) {
      const int idx = tid_e + k * NT;
      if (idx < ntraj * (TCH / 4)) *reinterpret_cast<f32x4*>(eb_lds + idx * 4) = ebv[k];
    }
    __syncthreads();
  }
  float r[GN ? NI : 1][GN ? NP : 1][VEC];
  auto row_partial = [&](const float (&v)[VEC], int row_local) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < VEC / 2; ++j) {
      s0 += v[j];
      s1 += v[VEC / 2 + j];
    }
    if (!two) s0 += s1;
    for (int o = 1; o < LPG; o <<= 1) s0 += __shfl_xor(s0, o, 64);
    if ((c & (LPG - 1)) == 0) {
      rs[row_local * GT + g0] = s0;
      if (two) rs[row_local * GT + g0 + 1] = s1;
    }
  };
  // fixed-order reduction of the per-row partials over each trajectory's L rows -> tab[traj][group] * scale
  auto reduce_rows = [&](float* tab, float scale) {
    const int L = p.gn_L, E = (p.tile_rows / L) * GT;
    int tpe = 64;
    while (tpe > 1 && E * tpe > NT) tpe >>= 1;
    const int tid = wave * 64 + lane, sub = tid & (tpe - 1), per_round = NT / tpe;
    for (int e0 = 0; e0 < E; e0 += per_round) {
      const int e = e0 + tid / tpe;
      float sum = 0.f;
      if (e < E) {
        const int traj = e / GT, g = e - traj * GT;
        for (int l = sub; l < L; l += tpe) sum += rs[(traj * L + l) * GT + g];
      }
      for (int o = 1; o < tpe; o <<= 1) sum += __shfl_xor(sum, o, 64);
      if (e < E && sub == 0) tab[e] = sum * scale;
    }
  };

#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      *reinterpret_cast<f32x4*>(patch + fr * STRIDE + (mi * 16 + fq * 4) * 4) = acc[mi][ni];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = r0 + RPP * i;
      const int row_local = wr * 64 + ni * 16 + row;
      const int n = n0 + row_local;
      float v[VEC];
#pragma unroll
      for (int h = 0; h < VEC / 4; ++h) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(patch + row * STRIDE + (c * VEC + h * 4) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[h * 4 + j] = t[j];
      }
      const bool live = n < n_end && co_ok;
      if (live) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] += bs[j];
        if (p.ebias) {
          const int bb = n / p.lout;
          if (eb_staged) {  // (two separate branches: a pointer select would become flat loads, which wait for vmcnt(0))
            const float* e = eb_lds + (bb - b_first) * TCH + wc * WCO + c * VEC;
#pragma unroll
            for (int h = 0; h < VEC / 4; ++h) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(e + h * 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[h * 4 + j] += t[j];
            }
          } else {
            const float* e = p.ebias + (size_t)bb * p.ebias_ld + p.ebias_off + co;
#pragma unroll
            for (int h = 0; h < VEC / 4; ++h) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(e + h * 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[h * 4 + j] += t[j];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = tdx_act(v[j], p.act);
        if (p.res) {
          float rv[VEC];
          Elem<T>::unpack(resv[ni][i], rv);
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] += rv[j];
        }
        if (p.out) *reinterpret_cast<u32x4*>(p.out + ((size_t)n * p.out_ld + p.out_off + co) * sizeof(T)) = Elem<T>::pack(v);
      }
      if constexpr (GN) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) r[ni][i][j] = live ? v[j] : 0.f;
        row_partial(r[ni][i], row_local);
      }
    }
  }
  if constexpr (GN) {
    const float inv_cnt = 1.0f / ((float)cpg * (float)p.gn_L);
    __syncthreads();
    if (!(p.debug & 8)) reduce_rows(tab_mean, inv_cnt);
    __syncthreads();
    if (!(p.debug & 16))
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int row_local = wr * 64 + ni * 16 + r0 + RPP * i;
        const int tj = (row_local < p.tile_rows ? row_local : 0) / p.gn_L;
        const float m0 = tab_mean[tj * GT + g0], m1 = two ? tab_mean[tj * GT + g0 + 1] : m0;
        float d[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float t = r[ni][i][j] - (j < VEC / 2 ? m0 : m1);
          d[j] = t * t;
        }
        row_partial(d, row_local);
      }
    __syncthreads();
    if (!(p.debug & 8)) reduce_rows(tab_var, inv_cnt);
    __syncthreads();
    if (!(p.debug & 32))
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int row_local = wr * 64 + ni * 16 + r0 + RPP * i;
        const int n = n0 + row_local;
        if (!(n < n_end && co_ok)) continue;
        const int tj = row_local / p.gn_L;
        const float m0 = tab_mean[tj * GT + g0], m1 = two ? tab_mean[tj * GT + g0 + 1] : m0;
        const float q0 = rsqrtf(tab_var[tj * GT + g0] + p.gn_eps), q1 = two ? rsqrtf(tab_var[tj * GT + g0 + 1] + p.gn_eps) : q0;
        float y[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float m = j < VEC / 2 ? m0 : m1, q = j < VEC / 2 ? q0 : q1;
          const float sc = q * gm[j];
          y[j] = tdx_act(fmaf(r[ni][i][j], sc, bt[j] - m * sc), p.gn_act);
        }
        *reinterpret_cast<u32x4*>(p.gn_out + ((size_t)n * p.gn_ld + p.gn_off + co) * sizeof(T)) = Elem<T>::pack(y);
      }
  }
}

// ------------------------------------------------------------------------------------------------------
// Small / general kernel: TR x TC tile, 4 waves as 2 (channels) x 2 (rows), operands staged through VGPRs.
// Handles every legal descriptor (channel counts that are not whole 128-byte blocks, tiny problems).
template <typename T, int TR, int TC>
__global__ __launch_bounds__(kThreads) void conv_gemm_kernel(const ConvArgs p) {
  constexpr int BK = kRowBytes / (int)sizeof(T);
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int RA = TC / 32, RB = TR / 32;       // staged rows per thread
  constexpr int MI = TC / 32, NI = TR / 32;       // 16x16 sub-tiles per wave
  constexpr int A_BYTES = TC * kRowBytes, B_BYTES = TR * kRowBytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wc = wave & 1, wr = wave >> 1;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so ids that are equal mod 8 share
  // an L2.  Give each of those 8 streams its own row tiles and let it walk all channel tiles of a row tile
  // back to back: an activation tile is then fetched into ONE L2 and re-used there by every channel tile.
  const int nco = (p.cout + TC - 1) / TC;
  const int xcd = blockIdx.x & 7, jt = blockIdx.x >> 3;
  const int row_tile = (jt / nco) * 8 + xcd;
  const int n0 = row_tile * TR;
  if (n0 >= p.N) return;  // block-uniform: padding blocks of the last group of 8 row tiles
  const int co0 = (jt % nco) * TC;

  // ---- staging role: 16-byte chunk c of rows rbase + 32*i -----------------------------------
  const int c = tid & 7, rbase = tid >> 3;
  const int lds_stage = rbase * kRowBytes + ((c ^ (rbase & 7)) << 4);
  int rb_b[RB], rb_l[RB];
  bool rb_ok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int n = n0 + rbase + 32 * i;
    rb_ok[i] = n < p.N;
    int b = n / p.lout;
    rb_b[i] = b;
    rb_l[i] = n - b * p.lout;
  }
  const char* zero = reinterpret_cast<const char*>(g_zero_row) + (c << 4);
  const char* wrow[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int co = co0 + rbase + 32 * i;
    wrow[i] = co < p.cout ? p.w + ((size_t)co * p.ktot + c * VEC) * sizeof(T) : nullptr;
  }

  // ---- K iterator (segment, channel block, tap) --------------------------------------------------
  int it_seg = 0, it_tap = 0, it_ci = 0;
  const char* xrow[3][RB];  // per-tap source row pointers of the current segment (statically indexed only)
  auto set_rows = [&]() {
    const SegDev& s = p.seg[it_seg];
    const int pad = s.taps == 3 ? 1 : 0;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        int u = rb_l[i] * s.stride + t - pad;
        bool ok = t < s.taps && rb_ok[i] && u >= 0 && u < s.lvirt;
        int uu = ok ? u : 0;
        int srow = s.up_scale > 0.f ? min((int)floorf((float)uu * s.up_scale), s.lin - 1) : uu;
        xrow[t][i] = ok ? s.src + ((size_t)(rb_b[i] * s.lin + srow) * s.src_ld + s.ch_off + c * VEC) * sizeof(T) : nullptr;
      }
  };
  set_rows();

  auto advance = [&]() {
    if (++it_tap >= p.seg[it_seg].taps) {
      it_tap = 0;
      it_ci += BK;
      if (it_ci >= p.seg[it_seg].cin) {
        it_ci = 0;
        ++it_seg;
        if (it_seg < p.nseg) set_rows();
      }
    }
  };
  // split-K: this workgroup owns K-steps [s_begin, s_end)
  const int s_begin = p.ksplit > 1 ? (int)blockIdx.y * p.steps_per_split : 0;
  const int s_end = p.ksplit > 1 ? min(p.nsteps, s_begin + p.steps_per_split) : p.nsteps;
  for (int s = 0; s < s_begin; ++s) advance();

  u32x4 ra[RA], rx[RB];
  auto load_step = [&]() {
    const SegDev& s = p.seg[it_seg];
    const int blen = s.cin - it_ci < BK ? s.cin - it_ci : BK;
    const bool chunk_ok = c * VEC < blen;
    const size_t koff = (size_t)(s.k0 + it_ci * s.taps + it_tap * blen) * sizeof(T);
    const size_t xoff = (size_t)it_ci * sizeof(T);
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const char* g = (wrow[i] && chunk_ok) ? wrow[i] + koff : zero;
      ra[i] = *reinterpret_cast<const u32x4*>(g);
    }
    with_tap(it_tap, [&](auto tc) {
      constexpr int TAP = decltype(tc)::value;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const char* g = (xrow[TAP][i] && chunk_ok) ? xrow[TAP][i] + xoff : zero;
        rx[i] = *reinterpret_cast<const u32x4*>(g);
      }
    });
    advance();
  };
  auto store_step = [&](int buf) {
    char* a = smem + buf * (A_BYTES + B_BYTES);
    char* b = a + A_BYTES;
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<u32x4*>(a + lds_stage + i * 32 * kRowBytes) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<u32x4*>(b + lds_stage + i * 32 * kRowBytes) = rx[i];
  };

  // ---- MFMA role ---------------------------------------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  const int a_frag = (wc * (TC / 2) + fr) * kRowBytes;  // + mi*16 rows
  const int b_frag = (wr * (TR / 2) + fr) * kRowBytes;
  const int sw = fr & 7;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* a = smem + buf * (A_BYTES + B_BYTES);
    const char* b = a + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = (((kk << 2) + fq) ^ sw) << 4;
      u32x4 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const u32x4*>(a + a_frag + mi * 16 * kRowBytes + ch);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const u32x4*>(b + b_frag + ni * 16 * kRowBytes + ch);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Mma<T>::run(af[mi], bf[ni], acc[mi][ni]);
    }
  };

  // ---- main loop: register-staged, LDS double-buffered, one barrier per K-step ---------------------
  load_step();
  store_step(0);
  __syncthreads();
  for (int s = s_begin; s < s_end; ++s) {
    const bool more = s + 1 < s_end;
    const int buf = (s - s_begin) & 1;
    if (more) load_step();
    compute(buf);
    if (more) store_step(buf ^ 1);
    __syncthreads();
  }
  if (p.ksplit > 1) epilogue_partial<MI, NI>(p, acc, n0 + wr * (TR / 2), co0 + wc * (TC / 2), fr, fq, (int)blockIdx.y);
  else epilogue<T, MI, NI>(p, acc, n0 + wr * (TR / 2), co0 + wc * (TC / 2), fr, fq);
}

template <typename T, int TR, int TC>
int launch(const ConvArgs& a, hipStream_t st) {
  const int nco = (a.cout + TC - 1) / TC, nrow = (a.N + TR - 1) / TR;
  dim3 grid((unsigned)(nco * ((nrow + 7) / 8) * 8));
  size_t lds = 2 * (size_t)(TR + TC) * kRowBytes;
  static bool attr_done_dev[kTdxMaxDevices] = {};   // (per device: hipFuncSetAttribute acts on the current device's copy of the kernel)
  bool& attr_done = attr_done_dev[tdx_device_slot()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, TR, TC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      TDX_FAIL(TDX_ERR_HIP, "conv_gemm: cannot reserve %zu bytes of LDS", lds);
    attr_done = true;
  }
  grid.y = a.ksplit > 1 ? (unsigned)a.ksplit : 1u;
  hipLaunchKernelGGL((conv_gemm_kernel<T, TR, TC>), grid, dim3(kThreads), lds, st, a);
  if (a.ksplit > 1) {
    const long quads = (long)a.N * (a.cout >> 2);
    hipLaunchKernelGGL((splitk_finish_kernel<T>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a);
  }
  TDX_CHECK_LAUNCH("conv_gemm_kernel");
  return TDX_OK;
}

// ------------------------------------------------------------------------------------------------------
// Large-problem kernel: 256-row x TC-channel tile, 8 waves (2 channel x 4 row), both operands brought in by
// LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPR staging).  The DMA writes LDS lane-linearly, so the
// bank-conflict XOR swizzle lives in the per-lane SOURCE chunk (lane (r, c) of an 8-row group fetches chunk
// c ^ r) and in the fragment reads.  Out-of-range rows (conv padding, tile tails) use an out-of-range buffer
// offset: the DMA then writes zeros.  Weight stages are double-buffered, one barrier per K-step, the DMA of
// step s+1 in flight under the MFMAs of step s.
//
// WIN = false ("gather"): the activation tile of every (channel block, tap) is gathered row by row, which also
//   covers stride 2 and nearest-upsampled sources.
// WIN = true  ("window"): stride-1 identity-mapped segments.  One window of 256 + 2 consecutive source rows per
//   channel block is staged ONCE and serves all three taps: tap t of output row i reads window row i + t, or a
//   permanently-zero LDS row when l + t - 1 falls outside the trajectory (Conv1d's padding).  Activation
//   traffic into LDS drops 3x and the window for block j+1 streams in under the three tap steps of block j.
constexpr int kDmaRows = 256;  // largest row tile (bounds the 32-bit in-tile offsets)
constexpr int win_groups(int tr) { return tr / 8 + 1; }                      // 8-row DMA groups per window (rows 0 .. tr+7)
constexpr int win_bytes(int tr) { return (win_groups(tr) * 8 + 1) * 128; }   // + one zero row
constexpr uint32_t kOob = 0x80000000u;

// Buffer resource words {base_lo, base_hi (stride 0), num_records, flags}, forced into SGPRs.
__device__ __forceinline__ u32x4 make_srd(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

// One LDS-DMA wave-instruction: 64 lanes x 16 B from srd[voff + soff] to LDS[lds_addr + 16*lane]; out-of-range
// lanes write zeros.  Issued through asm on purpose: hipcc orders every later ds_read behind a DMA it can see
// (s_waitcnt vmcnt(0) before the first fragment read), which would serialise the loads of step s+1 with the MFMAs
// of step s.  The kernel waits for its DMAs itself (vmcnt(0) + barrier at the top of each K-step), and its main
// loop contains no compiler-visible vector-memory load, so hipcc inserts no vmcnt wait of its own there.
__device__ __forceinline__ void dma16(uint32_t lds_addr, uint32_t voff, const u32x4& srd, uint32_t soff) {
  uint32_t keep;
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);  // wave-uniform by construction; make it provably an SGPR
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
      : "memory");
}

// TR = 256: 8 waves, one workgroup per CU (big GEMMs).  TR = 128: 4 waves and half the LDS, two workgroups per CU, so one
// workgroup's epilogue (a pure HBM store phase) overlaps the other's K loop: used for the narrow (Cout <= 128) layers.
template <typename T, int TC, int TR, bool WIN, bool GN>
__global__ __launch_bounds__(TR * 2) void conv_gemm_dma_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
  constexpr int ES = (int)sizeof(T);
  constexpr int NW = TR / 32;                     // waves: 2 (channels) x TR/64 (rows)
  constexpr int kWinGroups = win_groups(TR), kWinBytes = win_bytes(TR);
  constexpr int BK = kRowBytes / ES;
  constexpr int MI = TC / 32, NI = 4;             // wave tile: (TC/2) channels x 64 rows
  constexpr int GA = TC / 8 / NW, GB = TR / 8 / NW;  // 8-row DMA groups per wave per stage
  constexpr int GW = (kWinGroups + NW - 1) / NW;  // window groups per wave (last one partial)
  constexpr int A_BYTES = TC * kRowBytes, B_BYTES = TR * kRowBytes;
  constexpr int A_STAGE = WIN ? A_BYTES : A_BYTES + B_BYTES;  // distance between the two weight stages
  constexpr int W_BASE = 2 * A_BYTES;                          // WIN: the two activation windows follow the weight stages
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wr = wave >> 1;
  const int nco = (p.cout + TC - 1) / TC;
  const int xcd = blockIdx.x & 7, jt = blockIdx.x >> 3;
  const int row_tile = (jt / nco) * 8 + xcd;
  const int n0 = p.n_begin + row_tile * p.tile_rows;
  if (row_tile >= p.row_tiles || n0 >= p.N) return;
  const int co0 = (jt % nco) * TC;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- weight (A) loader: linear walk over the packed K axis ---------------------------------------------
  const int r8 = lane >> 3;
  const uint32_t chunk_off = (uint32_t)(((lane & 7) ^ r8) << 4);
  const int rows_a = p.cout - co0 < TC ? p.cout - co0 : TC;
  const u32x4 rsrc_a = make_srd(p.w + (size_t)co0 * p.ktot * ES, (uint32_t)rows_a * (uint32_t)p.ktot * ES);
  uint32_t voff_a[GA];
#pragma unroll
  for (int i = 0; i < GA; ++i) voff_a[i] = (uint32_t)(8 * (wave + NW * i) + r8) * (uint32_t)(p.ktot * ES) + chunk_off;  // rows >= rows_a fall out of range
  uint32_t soff_a = 0;
  auto issue_a = [&](int buf) {
    const uint32_t a = lds_base + buf * A_STAGE + wave * 1024;
    const uint32_t so = __builtin_amdgcn_readfirstlane(soff_a);
#pragma unroll
    for (int i = 0; i < GA; ++i) dma16(a + i * NW * 1024, voff_a[i], rsrc_a, so);
    soff_a += kRowBytes;
  };

  // ---- activation (B) loader -------------------------------------------------------------------------------
  u32x4 rsrc_b;
  int ld_seg = 0, ld_ci = 0, ld_tap = 0;   // loader iterator (gather: per step; window: per channel block)
  int ld_taps = 1, ld_cin = 0;             // cached fields of segment ld_seg (no kernarg loads inside the K loop)
  // gather state
  uint32_t voff_b[3][WIN ? 1 : GB];
  int rb_b[WIN ? 1 : GB], rb_l[WIN ? 1 : GB];
  // window state
  uint32_t voff_w[WIN ? GW : 1];

  auto ld_setup = [&]() {  // per segment
    const SegDev& s = p.seg[ld_seg];
    ld_taps = s.taps;
    ld_cin = s.cin;
    if constexpr (WIN) {
      const int pad = s.taps == 3 ? 1 : 0;
      const int base_row = n0 - pad > 0 ? n0 - pad : 0;
      const size_t base = ((size_t)base_row * s.src_ld + s.ch_off) * ES;
      const size_t left = (size_t)p.N * s.src_ld * ES - base;
      rsrc_b = make_srd(s.src + base, left > 0x7fffffffull ? 0x7fffffffu : (uint32_t)left);
#pragma unroll
      for (int i = 0; i < GW; ++i) {
        const int srow = n0 - pad + 8 * (wave + NW * i) + r8;  // flat source row of this lane's window row
        voff_w[i] = (srow >= 0 && srow < p.N) ? (uint32_t)((srow - base_row) * s.src_ld) * ES + chunk_off : kOob;
      }
    } else {
      const int b0 = n0 / p.lout;
      const size_t base = ((size_t)b0 * s.lin * s.src_ld + s.ch_off) * ES;
      const size_t left = (size_t)p.B * s.lin * s.src_ld * ES - base;
      rsrc_b = make_srd(s.src + base, left > 0x7fffffffull ? 0x7fffffffu : (uint32_t)left);
      const int pad = s.taps == 3 ? 1 : 0;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < GB; ++i) {
          const int u = rb_l[i] * s.stride + t - pad;
          const bool ok = t < s.taps && rb_b[i] >= 0 && u >= 0 && u < s.lvirt;
          const int uu = ok ? u : 0;
          const int srow = s.up_scale > 0.f ? min((int)floorf((float)uu * s.up_scale), s.lin - 1) : uu;
          voff_b[t][i] = ok ? (uint32_t)((rb_b[i] * s.lin + srow) * s.src_ld) * ES + chunk_off : kOob;
        }
    }
  };
  if constexpr (!WIN) {
    const int b0 = n0 / p.lout;
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int i_loc = 8 * (wave + NW * i) + r8;
      const int n = n0 + i_loc;
      const int b = n / p.lout;
      rb_b[i] = (n < p.N && i_loc < p.tile_rows) ? b - b0 : -1;
      rb_l[i] = n - b * p.lout;
    }
  }
  ld_setup();

  // gather: one activation tile per K-step, next to its weight stage
  auto issue_b_gather = [&](int buf) {
    const uint32_t b = lds_base + buf * A_STAGE + A_BYTES + wave * 1024;
    const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)ld_ci * ES);
    with_tap(ld_tap, [&](auto tc) {
      constexpr int TAP = decltype(tc)::value;
#pragma unroll
      for (int i = 0; i < (WIN ? 0 : GB); ++i) dma16(b + i * NW * 1024, voff_b[TAP][i], rsrc_b, so);
    });
    if (++ld_tap >= ld_taps) {
      ld_tap = 0;
      ld_ci += BK;
      if (ld_ci >= ld_cin) {
        ld_ci = 0;
        ++ld_seg;
        if (ld_seg < p.nseg) ld_setup();
      }
    }
  };
  // window: one window per channel block
  auto issue_window = [&](int wbuf) {
    const uint32_t w = lds_base + W_BASE + wbuf * kWinBytes + wave * 1024;
    const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)ld_ci * ES);
#pragma unroll
    for (int i = 0; i < (WIN ? GW : 0); ++i)
      if (wave + NW * i < kWinGroups) dma16(w + i * NW * 1024, voff_w[i], rsrc_b, so);
    ld_ci += BK;
    if (ld_ci >= ld_cin) {
      ld_ci = 0;
      ++ld_seg;
      if (ld_seg < p.nseg) ld_setup();
    }
  };

  // ---- MFMA role ---------------------------------------------------------------------------------------
  const int a_frag = (wc * (TC / 2) + fr) * kRowBytes;
  const int sw = fr & 7;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // window mode: tap t of output row i reads window row i + t; padding taps (l + t - 1 outside the trajectory) and
  // rows past the end of the problem read the zero row instead.  One validity bit per (sub-tile, tap).
  int cp_seg = 0, cp_ci = 0, cp_tap = 0, cp_blk = 0;  // compute iterator (window mode)
  int cp_taps = 1, cp_cin = 0;
  uint32_t wmask = 0;
  int wl[WIN ? NI : 1];
  auto cp_setup = [&]() {
    const SegDev& s = p.seg[cp_seg];
    cp_taps = s.taps;
    cp_cin = s.cin;
    wmask = 0;
#pragma unroll
    for (int ni = 0; ni < (WIN ? NI : 0); ++ni)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int u = s.taps == 3 ? wl[ni] + t - 1 : wl[ni];
        const bool ok = t < s.taps && wl[ni] >= 0 && u >= 0 && u < p.lout;
        wmask |= (ok ? 1u : 0u) << (ni * 3 + t);
      }
  };
  if constexpr (WIN) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int i_loc = wr * 64 + ni * 16 + fr;
      const int n = n0 + i_loc;
      wl[ni] = (n < p.N && i_loc < p.tile_rows) ? n % p.lout : -1;
    }
    cp_setup();
    if (tid < 16)  // the two zero rows (never touched by the DMA)
      *reinterpret_cast<u32x4*>(smem + W_BASE + (tid >> 3) * kWinBytes + kWinGroups * 8 * kRowBytes + (tid & 7) * 16) =
          (u32x4){0u, 0u, 0u, 0u};
  }

  auto compute = [&](int abuf, int wbuf) {
    const char* a = smem + abuf * A_STAGE;
    const char* b = WIN ? smem + W_BASE + wbuf * kWinBytes : a + A_BYTES;
    int boff[NI];
    const int shift = WIN ? (cp_taps == 3 ? cp_tap : 0) : 0;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int wrow = wr * 64 + ni * 16 + fr + shift;
      boff[ni] = wrow * kRowBytes + ((fq ^ (wrow & 7)) << 4);
      if constexpr (WIN) boff[ni] = ((wmask >> (ni * 3 + cp_tap)) & 1u) ? boff[ni] : kWinGroups * 8 * kRowBytes;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = (((kk << 2) + fq) ^ sw) << 4;
      u32x4 af[MI], bf[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const u32x4*>(b + (boff[ni] ^ (kk << 6)));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const u32x4*>(a + a_frag + mi * 16 * kRowBytes + ch);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Mma<T>::run(af[mi], bf[ni], acc[mi][ni]);
    }
  };

  // ---- main loop ---------------------------------------------------------------------------------------------
  const bool no_load = p.debug & 1, no_mma = p.debug & 2;
  if constexpr (WIN) issue_window(0);
  issue_a(0);
  if constexpr (!WIN) issue_b_gather(0);
  for (int s = 0; s < p.nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs for step s have landed ...
    __syncthreads();                                    // ... and so have everyone's; all waves are done reading the buffers refilled below
    if (s + 1 < p.nsteps && !no_load) {
      issue_a((s + 1) & 1);
      if constexpr (!WIN) issue_b_gather((s + 1) & 1);
    }
    if constexpr (WIN) {
      if (cp_tap == 0 && ld_seg < p.nseg && !no_load) issue_window((cp_blk + 1) & 1);  // window of the NEXT channel block
    }
    if (!no_mma) compute(s & 1, cp_blk & 1);
    if constexpr (WIN) {
      if (++cp_tap >= cp_taps) {
        cp_tap = 0;
        ++cp_blk;
        cp_ci += BK;
        if (cp_ci >= cp_cin) {
          cp_ci = 0;
          ++cp_seg;
          if (cp_seg < p.nseg) cp_setup();
        }
      }
    }
  }
  if (p.debug & 4) {
    float t = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) t += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
    if (t == 123.456f) p.out[0] = 1;  // keeps the accumulators live without storing the tile
    return;
  }
  epilogue_lds<T, MI, NI, TC / 2, TC, TR, GN>(p, acc, smem, wave, wr, wc, lane, n0, co0);
#endif
}


// ------------------------------------------------------------------------------------------------------
// Streaming (persistent) form of the LDS-DMA kernel: one workgroup per CU slot walks a list of tiles, and the
// K-steps of all its tiles form ONE software pipeline.  At the end of a tile the loader moves on first: the
// next tile's first weight stage and activation window are requested BEFORE the epilogue runs, the epilogue
// transposes through the two buffers the last K-step has just released (nothing is reserved for it), and its
// global stores are issued from asm and never waited for: the first K-step of the next tile waits with a
// counted vmcnt that covers the DMAs (older) but not the stores (younger).  So the tile prologue (DMA latency),
// the store drain and the workgroup launch all disappear under MFMA work instead of recurring per tile.
// Explicit LDS accesses by 32-bit LDS address (a generic pointer that the compiler cannot trace back to the LDS array becomes
// a flat access, and every flat access waits for vmcnt(0)).
typedef __attribute__((address_space(3))) f32x4 lds_f32x4_t;
__device__ __forceinline__ f32x4 lds_ld16(uint32_t a) { return *reinterpret_cast<lds_f32x4_t*>(a); }
__device__ __forceinline__ void lds_st16(uint32_t a, const f32x4& v) { *reinterpret_cast<lds_f32x4_t*>(a) = v; }

__device__ __forceinline__ int fdiv(int n, uint64_t m, int k) { return (int)(((uint64_t)(uint32_t)n * m) >> k); }

// 16-byte store through a buffer descriptor; lanes with an out-of-range offset are dropped by the hardware, so the
// instruction is issued by every wave exactly once per call (the counted wait above depends on that).
__device__ __forceinline__ void buf_store16(const u32x4& v, uint32_t voff, const u32x4& srd) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd) : "memory");
}

template <typename T, int TC, int TR>
struct StreamCfg {
  static constexpr int NW = TR / 32;
  static constexpr int PATCH = 2 * 16 * (32 * 4 + 16);         // one wave's two transpose pieces (16 rows x 32 fp32 channels, padded)
  static constexpr int WU = win_groups(TR) * 8 * kRowBytes;    // window bytes below its zero row
  static constexpr int NPW = WU / PATCH < NW ? WU / PATCH : NW;  // window mode: waves whose pieces fit the released window
  static constexpr int EBT = TR == 256 ? 22 : 12;              // embed-bias rows staged per tile (trajectories touched)
  static constexpr int NST = 4 * (TC / 2 / 32) * (16 / (64 / (32 / (16 / (int)sizeof(T)))));  // stores per wave per tile
  static_assert((NW - NPW) * PATCH + (EBT + 1) * TC * 4 <= TC * kRowBytes, "window-mode epilogue scratch does not fit the released weight stage");
  static_assert(NW * PATCH + (EBT + 1) * TC * 4 <= (TC + TR) * kRowBytes, "gather-mode epilogue scratch does not fit the released stage");
};

// Epilogue of the stream kernel, in two parts.
//   prefetch: for 128-channel tiles every global READ of the epilogue (bias, per-step embed row, the tile's per-trajectory
//     embed rows, the residual vectors) is issued with the tile's FIRST K-step, into registers, and pin() makes the
//     compiler wait for them right after the last step's MFMAs, i.e. BEFORE the next tile's DMAs are issued (vmcnt
//     counts in issue order, so a read issued after those DMAs would wait for them too).
//   epi_finish: transposes the accumulators through per-wave patches (16 rows x 64 channels, fp32) in released
//     stage buffers and stores whole 128/256-byte row segments; no read of global memory is left in it, except the
//     residual of 256-channel tiles (no registers to hold it during the K loop).
template <typename T, int NI, int WCO, int TCH, int NWV, int EBT, bool PRE_ROWS, bool PRE_RES, bool RES, bool EB, bool DIRECT_, bool GNF_ = false>
struct Epi {
  static constexpr bool GNF = GNF_;
  // PRE_ROWS: the rows that get staged in LDS are read with the tile's first K-step (else in finish()); PRE_RES: so is the
  // residual.  Both are a question of register room in the K loop (scripts/check_spills.py).
  static constexpr bool PRE = PRE_ROWS;
  static constexpr bool DIRECT = DIRECT_;
  static constexpr int ES = (int)sizeof(T);
  static constexpr int VEC = 16 / ES;
  static constexpr int PC = 32, NQ = WCO / PC;  // channels per transposed piece, pieces per sub-tile row
  static constexpr int STRIDE = PC * 4 + 16;    // fp32 piece row, padded: conflict-free ds_write_b128
  static constexpr int PIECE = 16 * STRIDE;     // a wave owns two of them (double buffer)
  static constexpr int CPR = PC / VEC, RPP = 64 / CPR, NP = 16 / RPP;
  static constexpr int NT = NWV * 64;
  static constexpr int EBR = ((EBT + 1) * (TCH / 4) + NT - 1) / NT;  // staged rows: [0] = bias + time-step row, [1 + t] = trajectory t
  static constexpr int NST = NI * NQ * NP;      // store instructions per wave per tile
  f32x4 ebv[EBR];
  u32x4 resv[RES ? NI : 1][NQ][NP];
  int b_first, ntraj, n_end;
  bool eb_staged;

  // The lane index is passed through an opaque asm in each part, so that none of the lane-dependent offsets below is
  // loop-invariant for the compiler: hoisted out of the tile loop they would sit in VGPRs across the K loop, whose
  // register budget (128 accumulators + fragments) has no room for them (a spill reload there is a vector-memory
  // operation, and waiting for it waits for every DMA in flight).
  static __device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
  }
  // The lane index, recomputed from nothing (mbcnt): no register has to carry it across the K loop for the per-tile / per-segment code.
  static __device__ __forceinline__ int fresh_lane() {
    int v;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(v));
    return v;
  }

  __device__ __forceinline__ void load_res(const ConvArgs& p, int wr, int wc, int lane, int n0, int co0) {
    if constexpr (DIRECT) {  // the layout of finish_direct: lane (fr, fq), sub-tile row ni, channel chunk P
      const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int P = 0; P < NQ * NP; ++P) {
          const int n = n0 + wr * 64 + ni * 16 + fr;
          const int co = co0 + wc * WCO + P * 4 * VEC + fq * VEC;
          resv[ni][P / NP][P % NP] = (n < n_end && co < p.cout) ? *reinterpret_cast<const u32x4*>(p.res + ((size_t)n * p.res_ld + co) * ES)
                                                                : (u32x4){0u, 0u, 0u, 0u};
        }
      return;
    }
    const int c = lane % CPR, r0 = lane / CPR;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int n = n0 + wr * 64 + ni * 16 + r0 + RPP * i;
          const int co = co0 + wc * WCO + q * PC + c * VEC;
          resv[ni][q][i] = (n < n_end && co < p.cout) ? *reinterpret_cast<const u32x4*>(p.res + ((size_t)n * p.res_ld + co) * ES)
                                                      : (u32x4){0u, 0u, 0u, 0u};
        }
  }

  // Global reads of the epilogue into registers: the rows that go to LDS (bias + time-step row, the tile's per-trajectory
  // embed rows; absent operands read a zero row instead of branching, so all loads are issued back to back) and, when
  // PRE_RES, the residual vectors.
  __device__ __forceinline__ void prefetch(const ConvArgs& p, int wave, int wr, int wc, int lane_, int n0, int co0) {
    const int lane = fresh_lane();
    n_end = n0 + p.tile_rows < p.N ? n0 + p.tile_rows : p.N;
    b_first = fdiv(n0, p.div_m, p.div_k);
    ntraj = fdiv(n_end - 1, p.div_m, p.div_k) - b_first + 1;
    eb_staged = EB && p.ebias && ntraj <= EBT;
    const float* zeros = reinterpret_cast<const float*>(g_zero_row);
    const int tid_e = wave * 64 + lane;
#pragma unroll
    for (int k = 0; k < EBR; ++k) {
      const int idx = tid_e + k * NT;
      const int t = idx / (TCH / 4), q4 = idx - t * (TCH / 4);
      const int cc = co0 + q4 * 4;
      const bool col = cc < p.cout;
      if (k == 0 && (TCH / 4) <= NT) {  // slot 0 holds row 0 (and the first trajectory rows)
        const float* b1 = (t == 0 && p.bias && col) ? p.bias + cc : zeros;
        const float* b2 = (t == 0 && p.tbias && col) ? p.tbias + p.ebias_off + cc : zeros;
        const float* e = (t >= 1 && eb_staged && t - 1 < ntraj && col) ? p.ebias + (size_t)(b_first + t - 1) * p.ebias_ld + p.ebias_off + cc : zeros;
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(b1), v2 = *reinterpret_cast<const f32x4*>(b2), v3 = *reinterpret_cast<const f32x4*>(e);
#pragma unroll
        for (int j = 0; j < 4; ++j) ebv[k][j] = ((0.f + v1[j]) + v2[j]) + v3[j];  // row 0: bias + tbias (v3 = 0); rows >= 1: 0 + 0 + e
      } else {
        const float* e = (eb_staged && t >= 1 && t - 1 < ntraj && col) ? p.ebias + (size_t)(b_first + t - 1) * p.ebias_ld + p.ebias_off + cc : zeros;
        ebv[k] = *reinterpret_cast<const f32x4*>(e);
      }
    }
    if constexpr (PRE_RES && RES) load_res(p, wr, wc, lane, n0, co0);
  }

  // Consume every prefetched register in an (empty) asm statement: the compiler's wait for the loads lands here.
  __device__ __forceinline__ void pin(const ConvArgs& p) {
#pragma unroll
    for (int k = 0; k < EBR; ++k) asm volatile("" : "+v"(ebv[k]));
    if constexpr (PRE_RES && RES)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(resv[ni][q][i]));
  }

  // Direct epilogue.  The stream kernel DMAs the weight rows of a wave tile in a permuted order (perm_row below), so that the
  // accumulators a lane holds for TP = VEC / 4 adjacent 16-channel sub-tiles are VEC CONSECUTIVE output channels of one row:
  // 16 bytes per lane straight from the registers, 64-byte row segments per store instruction (what the transposing epilogue
  // stores too), and no LDS round trip for the values.  LDS holds only the staged bias / embed rows.  Same association of the
  // additions as finish(): (acc + (bias + step row)) + trajectory row -> act -> + residual.
  //   sub-tile mi, MFMA row m (lane group fq = m >> 2, register j = m & 3)  ->  channel (mi / TP) * 16 * TP + fq * VEC + (mi % TP) * 4 + j
  static __device__ __forceinline__ int perm_row(int x) {  // x = LDS row of the weight stage; returns the weight row that goes there
    if constexpr (VEC == 8) return (x & ~0x1c) | ((x & 0x08) << 1) | ((x & 0x04) << 1) | ((x & 0x10) >> 2);
    else return x;
  }
  // bias + time-step row and the tile's per-trajectory embed rows: registers (prefetch) -> LDS
  __device__ __forceinline__ void stage_rows(uint32_t eb_lds, int wave, int lane_) {
    const int tid_e = wave * 64 + fresh_lane();
#pragma unroll
    for (int k = 0; k < EBR; ++k) {
      const int idx = tid_e + k * NT;
      if (idx < (1 + (eb_staged ? ntraj : 0)) * (TCH / 4)) lds_st16(eb_lds + idx * 16, ebv[k]);
    }
  }
  template <int MI>
  __device__ __forceinline__ void finish_direct(const ConvArgs& p, f32x4 (&acc)[MI][NI], uint32_t eb_lds, int wave, int wr, int wc, int lane_,
                                                int n0, int co0) {
    constexpr int TP = VEC / 4, NPR = MI / TP;  // sub-tiles per 16-byte store, stores per sub-tile row
    static_assert(NPR == NQ * NP && NST == NI * NPR, "store count");
    const int lane = fresh_lane();
    const int fr = lane & 15, fq = lane >> 4;
    const int cw = co0 + wc * WCO;  // first channel of this wave's tile
    if constexpr (!PRE) prefetch(p, wave, wr, wc, lane, n0, co0);
    if constexpr (!PRE_RES && RES) load_res(p, wr, wc, lane, n0, co0);
    const size_t obase = ((size_t)n0 * p.out_ld + p.out_off + cw) * ES;
    const size_t oleft = (size_t)p.N * p.out_ld * ES - obase;
    const u32x4 rsrc_o = make_srd(p.out + obase, oleft > 0x7fffffffull ? 0x7fffffffu : (uint32_t)oleft);
    if constexpr ((TDX_ABL & 1) != 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
      return;
    }
    if constexpr ((TDX_ABL & 4) == 0) {
      __syncthreads();  // every wave is done with the K loop's last buffers: the staged rows go there
      stage_rows(eb_lds, wave, lane);
      __syncthreads();
    }
    const bool eb_lds_on = (TDX_ABL & 4) == 0 && EB && p.ebias && eb_staged;
    const int ch0 = wc * WCO + fq * VEC;  // this lane's first channel inside the block tile (chunk P adds P * 4 * VEC)
    // Items k = ni * NPR + P; the LDS reads of item k + 1 (bias vector, embed row of the row's trajectory) are requested
    // before the arithmetic of item k.
    struct Item {
      f32x4 bs[TP], eb[TP];
    };
    uint32_t eb_row = 0;  // LDS address of the staged embed row of sub-tile row ni's trajectory (+ this lane's channel offset)
    auto get = [&](int k, Item& x) {
      const int ni = k / NPR, P = k % NPR;
#pragma unroll
      for (int h = 0; h < TP; ++h) x.bs[h] = (TDX_ABL & 4) ? (f32x4){0.f, 0.f, 0.f, 0.f} : lds_ld16(eb_lds + (ch0 + P * 4 * VEC + h * 4) * 4);
      if constexpr (EB) if (eb_lds_on) {
        if (P == 0) {
          const int n = n0 + wr * 64 + ni * 16 + fr;
          const int t = fdiv(n < n_end ? n : n_end - 1, p.div_m, p.div_k) - b_first;  // (rows past the end are never stored)
          eb_row = eb_lds + ((1 + t) * TCH + ch0) * 4;
        }
#pragma unroll
        for (int h = 0; h < TP; ++h) x.eb[h] = lds_ld16(eb_row + (P * 4 * VEC + h * 4) * 4);
      }
    };
    Item cur, nxt;
    get(0, cur);
#pragma unroll
    for (int k = 0; k < NI * NPR; ++k) {
      const int ni = k / NPR, P = k % NPR;
      if (k + 1 < NI * NPR) get(k + 1, nxt);
      const int row_local = wr * 64 + ni * 16 + fr;
      const int n = n0 + row_local;
      const int chl = ch0 + P * 4 * VEC;
      const bool live = n < n_end && co0 + chl < p.cout;
      float v[VEC];
#pragma unroll
      for (int h = 0; h < TP; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[h * 4 + j] = acc[P * TP + h][ni][j] + cur.bs[h][j];
      if constexpr (EB) {
        if (eb_lds_on) {
#pragma unroll
          for (int h = 0; h < TP; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[h * 4 + j] += cur.eb[h][j];
        } else if (p.ebias && live) {  // more trajectories in the tile than staged rows: read the rows from global memory
          const int bb = fdiv(n, p.div_m, p.div_k);
          const float* e = p.ebias + (size_t)bb * p.ebias_ld + p.ebias_off + co0 + chl;
#pragma unroll
          for (int h = 0; h < TP; ++h) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(e + h * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[h * 4 + j] += t[j];
          }
        }
      }
      if (p.act == TDX_ACT_SWISH) {  // one wave-uniform branch per vector, not one per element
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = tdx_swish(v[j]);
      } else if (p.act == TDX_ACT_LEAKY) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = tdx_leaky(v[j]);
      }
      if constexpr (RES) {
        float rv[VEC];
        Elem<T>::unpack(resv[ni][P / NP][P % NP], rv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] += rv[j];
      }
      uint32_t voff = live ? (uint32_t)(row_local * p.out_ld + P * 4 * VEC + fq * VEC) * ES : kOob;
      if constexpr ((TDX_ABL & 8) != 0)  // timing only (wrong data): every store instruction writes 8 rows x 128 contiguous bytes
        voff = live ? (uint32_t)((wr * 64 + ni * 16 + (fr & 7) + 8 * (P & 1)) * p.out_ld + (P >> 1) * 8 * VEC + ((fr >> 3) * 4 + fq) * VEC) * ES : kOob;
      if constexpr ((TDX_ABL & 2) != 0) {
        const u32x4 o = Elem<T>::pack(v);
        asm volatile("" ::"v"(o), "v"(voff), "s"(rsrc_o));
      } else {
        buf_store16(Elem<T>::pack(v), voff, rsrc_o);
      }
      cur = nxt;
    }
  }

  // Direct epilogue with the CONSUMER's GroupNorm (+ activation) applied to the finished rows (p.gn_*): the persistent form of
  // epilogue_lds<GN = true>.  Row tiles hold whole trajectories (p.tile_rows = k * lout, tiles start on a trajectory), and a channel
  // tile holds whole groups, so every (trajectory, group) statistic is complete inside the workgroup.  The finished fp32 values stay
  // in the accumulator registers; per-row group sums go through LDS (rs[group][row], in the window the K loop released) and are
  // added per trajectory in a fixed order that depends only on the row's position INSIDE its trajectory: exact two-pass statistics
  // (mean, then centred squares), bitwise reproducible and independent of how the batch is cut into tiles or shards.  No atomics.
  //   rs_lds : GT x 256 floats (GT = TCH / cpg groups per channel tile);  tab_lds : mean[E] | rstd[E], E = (tile_rows / lout) * GT <= 1024
  //   (its own 8 KB); gb_lds : gamma[0 .. cout) at +0, beta at +4096 (staged once per workgroup at kernel start).
  template <int MI>
  __device__ __forceinline__ void finish_direct_gn(const ConvArgs& p, f32x4 (&acc)[MI][NI], uint32_t eb_lds, uint32_t rs_lds, uint32_t tab_lds,
                                                   uint32_t gb_lds, int wave, int wr, int wc, int lane_, int n0, int co0) {
    constexpr int TP = VEC / 4, NPR = MI / TP;
    static_assert(VEC == 8 && NPR == NQ * NP && NST == NI * NPR, "bf16 direct layout");
    typedef __attribute__((address_space(3))) float lds_f32_t;
    const int lane = fresh_lane();
    const int fr = lane & 15, fq = lane >> 4;
    const int cw = co0 + wc * WCO;
    if constexpr (!PRE) prefetch(p, wave, wr, wc, lane, n0, co0);
    // residual vectors that were not prefetched with the tile's first K-step (256-channel tiles): one sub-tile row ahead of their use,
    // 2 x NPR vectors live instead of all NI x NPR (with the 128 finished values in registers there is no room for more: a spill here
    // means scratch memory, and a kernel that needs scratch is not something to put into a replayed graph lightly)
    constexpr bool LATE_RES = RES && !PRE_RES;
    u32x4 rnext[LATE_RES ? NPR : 1];
    auto res_row = [&](int ni, u32x4 (&dst)[LATE_RES ? NPR : 1]) {
      const int n = n0 + wr * 64 + ni * 16 + fr;
#pragma unroll
      for (int P = 0; P < (LATE_RES ? NPR : 0); ++P) {
        const int co = co0 + wc * WCO + P * 4 * VEC + fq * VEC;
        dst[P] = (n < n_end && co < p.cout) ? *reinterpret_cast<const u32x4*>(p.res + ((size_t)n * p.res_ld + co) * ES) : (u32x4){0u, 0u, 0u, 0u};
      }
    };
    if constexpr (LATE_RES) res_row(0, rnext);
    const bool raw = p.out != nullptr;
    const size_t obase = ((size_t)n0 * p.out_ld + p.out_off + cw) * ES;
    const size_t oleft = raw ? (size_t)p.N * p.out_ld * ES - obase : 0;
    const u32x4 rsrc_o = make_srd(raw ? p.out + obase : p.gn_out, oleft > 0x7fffffffull ? 0x7fffffffu : (uint32_t)oleft);
    const size_t gbase = ((size_t)n0 * p.gn_ld + p.gn_off + cw) * ES;
    const size_t gleft = (size_t)p.N * p.gn_ld * ES - gbase;
    const u32x4 rsrc_g = make_srd(p.gn_out + gbase, gleft > 0x7fffffffull ? 0x7fffffffu : (uint32_t)gleft);
    __syncthreads();  // every wave is done with the K loop's last buffers: the staged rows and the statistics go there
    stage_rows(eb_lds, wave, lane);
    __syncthreads();
    const bool eb_lds_on = EB && p.ebias && eb_staged;
    const int ch0 = wc * WCO + fq * VEC;
    // ---- finished values (bias, embed rows, activation, residual) back into the accumulator registers; raw rows stored if wanted ----
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row_local = wr * 64 + ni * 16 + fr;
      const int n = n0 + row_local;
      uint32_t eb_row = 0;
      if constexpr (EB) if (eb_lds_on) {
        const int t = fdiv(n < n_end ? n : n_end - 1, p.div_m, p.div_k) - b_first;
        eb_row = eb_lds + ((1 + t) * TCH + ch0) * 4;
      }
      u32x4 rcur[LATE_RES ? NPR : 1];
      if constexpr (LATE_RES) {
#pragma unroll
        for (int P = 0; P < NPR; ++P) rcur[P] = rnext[P];
        if (ni + 1 < NI) res_row(ni + 1, rnext);
      }
#pragma unroll
      for (int P = 0; P < NPR; ++P) {
        const int chl = ch0 + P * 4 * VEC;
        const bool live = n < n_end && co0 + chl < p.cout;
        float v[VEC];
#pragma unroll
        for (int h = 0; h < TP; ++h) {
          const f32x4 bs = lds_ld16(eb_lds + (chl + h * 4) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[h * 4 + j] = acc[P * TP + h][ni][j] + bs[j];
        }
        if constexpr (EB) {
          if (eb_lds_on) {
#pragma unroll
            for (int h = 0; h < TP; ++h) {
              const f32x4 e = lds_ld16(eb_row + (P * 4 * VEC + h * 4) * 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[h * 4 + j] += e[j];
            }
          } else if (p.ebias && live) {
            const int bb = fdiv(n, p.div_m, p.div_k);
            const float* e = p.ebias + (size_t)bb * p.ebias_ld + p.ebias_off + co0 + chl;
#pragma unroll
            for (int h = 0; h < TP; ++h) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(e + h * 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[h * 4 + j] += t[j];
            }
          }
        }
        if (p.act == TDX_ACT_SWISH) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] = tdx_swish(v[j]);
        } else if (p.act == TDX_ACT_LEAKY) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] = tdx_leaky(v[j]);
        }
        if constexpr (RES) {
          float rv[VEC];
          if constexpr (LATE_RES) Elem<T>::unpack(rcur[P], rv);
          else Elem<T>::unpack(resv[ni][P / NP][P % NP], rv);
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] += rv[j];
        }
        buf_store16(Elem<T>::pack(v), live && raw ? (uint32_t)(row_local * p.out_ld + P * 4 * VEC + fq * VEC) * ES : kOob, rsrc_o);
#pragma unroll
        for (int h = 0; h < TP; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[P * TP + h][ni][j] = live ? v[h * 4 + j] : 0.f;
      }
    }
    // ---- statistics ----
    const int cpg = p.gn_cpg;                       // 8, 16, 32 or 64 channels per group
    const int cps = __builtin_ctz((unsigned)cpg);   // (powers of two: shifts instead of integer divisions)
    const int GT = TCH >> cps, gts = __builtin_ctz((unsigned)GT);
    const int lsh = cpg >= 32 ? 2 : (cpg >= 16 ? 1 : 0);   // a group spans 1 << lsh lanes (fq) of a row
    const bool pairP = cpg == 64;                   // ... and two 32-channel chunks
    const bool writer = (fq & ((1 << lsh) - 1)) == 0;
    const int L = p.gn_L;
    const int E = (p.tile_rows / L) * GT;
    const int tpe = p.gn_tpe;                       // power of two <= 16: threads that share one (trajectory, group) sum
    const int tid = wave * 64 + lane;
    const float inv_cnt = 1.0f / ((float)cpg * (float)L);
    auto lane_sum = [&](float s) {
      if (lsh >= 1) s += __shfl_xor(s, 16, 64);
      if (lsh >= 2) s += __shfl_xor(s, 32, 64);
      return s;
    };
    // rs[g][row] <- per-row group sums of f(value); then tab[e] <- fin(sum over the trajectory's rows * inv_cnt)
    auto reduce_rows = [&](uint32_t tab, auto fin) {
      const int tps = __builtin_ctz((unsigned)tpe);
      const int sub = tid & (tpe - 1), per_round = NT >> tps;
      for (int e0 = 0; e0 < E; e0 += per_round) {
        const int e = e0 + (tid >> tps);
        float sum = 0.f;
        if (e < E) {
          const int traj = e >> gts, g = e & (GT - 1);
          const uint32_t base = rs_lds + (uint32_t)(g * 256 + traj * L) * 4;
          for (int l = sub; l < L; l += tpe) sum += *reinterpret_cast<lds_f32_t*>(base + l * 4);
        }
        for (int o = 1; o < tpe; o <<= 1) sum += __shfl_xor(sum, o, 64);
        if (e < E && sub == 0) *reinterpret_cast<lds_f32_t*>(tab + e * 4) = fin(sum * inv_cnt);
      }
    };
    // trajectory (inside the tile) of each of this lane's four rows, and the group of each channel chunk
    int trow[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wr * 64 + ni * 16 + fr;
      trow[ni] = fdiv(n < n_end ? n : n_end - 1, p.div_m, p.div_k) - b_first;
    }
    const uint32_t tab_mean = tab_lds, tab_rstd = tab_lds + (uint32_t)E * 4;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row_local = wr * 64 + ni * 16 + fr;
        float sp[NPR];
#pragma unroll
        for (int P = 0; P < NPR; ++P) {
          const int g = (ch0 + P * 4 * VEC) >> cps;
          const float m = pass ? *reinterpret_cast<lds_f32_t*>(tab_mean + (trow[ni] * GT + g) * 4) : 0.f;
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = acc[P * TP][ni][j] - m, b = acc[P * TP + 1][ni][j] - m;
            s0 += pass ? a * a : a;
            s1 += pass ? b * b : b;
          }
          sp[P] = s0 + s1;
        }
#pragma unroll
        for (int P = 0; P < NPR; ++P) {
          if (pairP && (P & 1)) continue;
          const float s = lane_sum(pairP ? sp[P] + sp[P + 1 < NPR ? P + 1 : P] : sp[P]);
          const int g = (ch0 + P * 4 * VEC) >> cps;
          if (writer) *reinterpret_cast<lds_f32_t*>(rs_lds + (uint32_t)(g * 256 + row_local) * 4) = s;
        }
      }
      __syncthreads();
      const float eps = p.gn_eps;
      if (pass == 0) reduce_rows(tab_mean, [](float x) { return x; });
      else reduce_rows(tab_rstd, [eps](float x) { return rsqrtf(x + eps); });
      __syncthreads();
    }
    // ---- normalise, activate, store ----
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row_local = wr * 64 + ni * 16 + fr;
      const int n = n0 + row_local;
#pragma unroll
      for (int P = 0; P < NPR; ++P) {
        const int chl = ch0 + P * 4 * VEC;
        const bool live = n < n_end && co0 + chl < p.cout;
        const int g = chl >> cps;
        const float m = *reinterpret_cast<lds_f32_t*>(tab_mean + (trow[ni] * GT + g) * 4);
        const float q = *reinterpret_cast<lds_f32_t*>(tab_rstd + (trow[ni] * GT + g) * 4);
        float y[VEC];
#pragma unroll
        for (int h = 0; h < TP; ++h) {
          const f32x4 gm = lds_ld16(gb_lds + (co0 + chl + h * 4) * 4), bt = lds_ld16(gb_lds + 4096 + (co0 + chl + h * 4) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float sc = q * gm[j];
            y[h * 4 + j] = fmaf(acc[P * TP + h][ni][j], sc, bt[j] - m * sc);
          }
        }
        if (p.gn_act == TDX_ACT_SWISH) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) y[j] = tdx_swish(y[j]);
        } else if (p.gn_act == TDX_ACT_LEAKY) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) y[j] = tdx_leaky(y[j]);
        }
        buf_store16(Elem<T>::pack(y), live ? (uint32_t)(row_local * p.gn_ld + P * 4 * VEC + fq * VEC) * ES : kOob, rsrc_g);
      }
    }
  }

  // patch: LDS address of this wave's two pieces; eb_lds: LDS address of the staged rows.
  template <int MI>
  __device__ __forceinline__ void finish(const ConvArgs& p, f32x4 (&acc)[MI][NI], uint32_t patch, uint32_t eb_lds, int wave, int wr, int wc,
                                         int lane_, int n0, int co0) {
    static_assert(TCH / 4 <= NT, "one staging slot must cover a whole row");
    if constexpr (DIRECT) {
      finish_direct(p, acc, eb_lds, wave, wr, wc, lane_, n0, co0);
      return;
    }
    const int lane = fresh_lane();
    const int fr = lane & 15, fq = lane >> 4;
    const int c = lane % CPR, r0 = lane / CPR;
    const int cw = co0 + wc * WCO;  // first channel of this wave's tile
    if constexpr (!PRE) prefetch(p, wave, wr, wc, lane, n0, co0);
    if constexpr (!PRE_RES && RES) load_res(p, wr, wc, lane, n0, co0);
    // output descriptor: base = first element of this wave's tile, range = rest of the output matrix
    const size_t obase = ((size_t)n0 * p.out_ld + p.out_off + cw) * ES;
    const size_t oleft = (size_t)p.N * p.out_ld * ES - obase;
    const u32x4 rsrc_o = make_srd(p.out + obase, oleft > 0x7fffffffull ? 0x7fffffffu : (uint32_t)oleft);
    __syncthreads();  // every wave is done with the K loop's last buffers: they are epilogue scratch now
    {
      const int tid_e = wave * 64 + lane;
#pragma unroll
      for (int k = 0; k < EBR; ++k) {
        const int idx = tid_e + k * NT;
        if (idx < (1 + (eb_staged ? ntraj : 0)) * (TCH / 4)) lds_st16(eb_lds + idx * 16, ebv[k]);
      }
    }
    __syncthreads();
    // Pieces of 16 rows x 32 channels, two LDS buffers per wave, software-pipelined: piece k + 1 is written and piece k is
    // read back (transposed: a lane gets 16 bytes of output of one row) before the arithmetic of piece k starts, so no
    // LDS round trip is waited for with nothing else to do.  (LDS operations of a wave execute in order: the write of
    // piece k + 2 into the buffer of piece k cannot overtake the read of piece k.)
    constexpr int NPC = NI * NQ;
    auto put = [&](int k) {  // k = ni * NQ + q
      const int ni = k / NQ, q = k % NQ;
#pragma unroll
      for (int m = 0; m < PC / 16; ++m)
        lds_st16(patch + (k & 1) * PIECE + fr * STRIDE + (m * 16 + fq * 4) * 4, acc[q * (PC / 16) + m][ni]);
    };
    // what the arithmetic of one piece reads from LDS: its transposed values, the bias row and (staged) embed rows
    struct Piece {
      f32x4 d[NP][VEC / 4], bs[VEC / 4], eb[EB ? NP : 1][VEC / 4];
    };
    const bool eb_lds_on = EB && p.ebias && eb_staged;
    auto get = [&](int k, Piece& x) {
      const int ni = k / NQ, q = k % NQ;
      const int chl = wc * WCO + q * PC + c * VEC;
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int h = 0; h < VEC / 4; ++h) x.d[i][h] = lds_ld16(patch + (k & 1) * PIECE + (r0 + RPP * i) * STRIDE + (c * VEC + h * 4) * 4);
#pragma unroll
      for (int h = 0; h < VEC / 4; ++h) x.bs[h] = lds_ld16(eb_lds + (chl + h * 4) * 4);
      if constexpr (EB) if (eb_lds_on) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int n = n0 + wr * 64 + ni * 16 + r0 + RPP * i;
          int t = fdiv(n < n_end ? n : n_end - 1, p.div_m, p.div_k) - b_first;  // (rows past the end are never stored)
#pragma unroll
          for (int h = 0; h < VEC / 4; ++h) x.eb[i][h] = lds_ld16(eb_lds + ((1 + t) * TCH + chl + h * 4) * 4);
        }
      }
    };
    Piece cur, nxt;
    put(0);
    if (NPC > 1) put(1);
    get(0, cur);
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int ni = k / NQ, q = k % NQ;
      if (k + 1 < NPC) get(k + 1, nxt);
      if (k + 2 < NPC) put(k + 2);
      const int chl = wc * WCO + q * PC + c * VEC;  // channel inside the block tile
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int row_local = wr * 64 + ni * 16 + r0 + RPP * i;
        const int n = n0 + row_local;
        const bool live = n < n_end && co0 + chl < p.cout;
        float v[VEC];
#pragma unroll
        for (int h = 0; h < VEC / 4; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[h * 4 + j] = cur.d[i][h][j] + cur.bs[h][j];
        if constexpr (EB) if (eb_lds_on) {
#pragma unroll
          for (int h = 0; h < VEC / 4; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[h * 4 + j] += cur.eb[i][h][j];
        } else if (p.ebias && live) {  // more trajectories in the tile than staged rows: read the rows from global memory
          const int bb = fdiv(n, p.div_m, p.div_k);
          const float* e = p.ebias + (size_t)bb * p.ebias_ld + p.ebias_off + co0 + chl;
#pragma unroll
          for (int h = 0; h < VEC / 4; ++h) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(e + h * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[h * 4 + j] += t[j];
          }
        }
        if (p.act == TDX_ACT_SWISH) {  // one wave-uniform branch per vector, not one per element
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] = tdx_swish(v[j]);
        } else if (p.act == TDX_ACT_LEAKY) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] = tdx_leaky(v[j]);
        }
        if constexpr (RES) {
          float rv[VEC];
          Elem<T>::unpack(resv[ni][q][i], rv);
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] += rv[j];
        }
        const uint32_t voff = live ? (uint32_t)(row_local * p.out_ld + q * PC + c * VEC) * ES : kOob;
        buf_store16(Elem<T>::pack(v), voff, rsrc_o);
      }
      cur = nxt;
    }
  }
};

// MODE: 0 = no per-trajectory embed rows and no residual, 1 = embed rows, 2 = residual (both at once: one-tile kernel).
// GNF: the consumer's GroupNorm (+ activation) in the epilogue (Epi::finish_direct_gn; bf16 window kernels with 256-row tiles).
constexpr int kGnTabMax = 1024;              // (trajectory, group) statistics per tile
constexpr int kGnLds = 8192 + 8192 + 64;     // gamma | beta of up to 1024 output channels, mean | rstd tables, behind the (diagnostic) stamp words
template <typename T, int TC, int TR, bool WIN, int MODE, bool GNF = false>
__global__ __launch_bounds__(TR * 2) void conv_gemm_stream_kernel(const ConvArgs p) {
  constexpr bool RES = MODE == 2, EB = MODE == 1;
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = StreamCfg<T, TC, TR>;
  constexpr int ES = (int)sizeof(T);
  constexpr int NW = TR / 32;
  constexpr int kWinGroups = win_groups(TR), kWinBytes = win_bytes(TR);
  constexpr int BK = kRowBytes / ES;
  constexpr int MI = TC / 32, NI = 4;
  constexpr int GA = TC / 8 / NW, GB = TR / 8 / NW;
  constexpr int GW = (kWinGroups + NW - 1) / NW;
  constexpr int A_BYTES = TC * kRowBytes, B_BYTES = TR * kRowBytes;
  constexpr int A_STAGE = WIN ? A_BYTES : A_BYTES + B_BYTES;
  constexpr int W_BASE = 2 * A_BYTES;
  constexpr bool SPREAD = TDX_SPREAD_DMA && WIN && TC == 256;   // (same-box A/B: +1..4 % on the 256-channel tiles, -1.5 % on the 128-channel ones)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wr = wave >> 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int r8 = lane >> 3;
  const uint32_t chunk_off = (uint32_t)(((lane & 7) ^ r8) << 4);

  // ---- tile list of this workgroup: XCD x = blockIdx & 7 owns row tiles x, x + 8, ...; its workgroups walk
  //      (row tile, channel tile) pairs in the same order as the one-tile-per-workgroup kernel -----------------------
  const int nco = (p.cout + TC - 1) / TC;
  const int xcd = blockIdx.x & 7, nslot = gridDim.x >> 3;
  const int jt_end = nco * ((p.row_tiles + 7) >> 3);
  auto tile_of = [&](int jt, int& n0, int& co0) -> bool {
    const int rg = jt / nco, row_tile = rg * 8 + xcd;
    n0 = p.n_begin + row_tile * p.tile_rows;
    co0 = (jt - rg * nco) * TC;
    return jt < jt_end && row_tile < p.row_tiles && n0 < p.N;
  };
  int jt = blockIdx.x >> 3, cp_n0, cp_co0;
  if (!tile_of(jt, cp_n0, cp_co0)) return;

  // (the gathered 256-channel tiles keep the transposing epilogue: no register room for the direct one, scripts/check_spills.py)
  // (GroupNorm epilogue on 256-channel tiles with a residual: the staged rows are read in the epilogue, where the residual loads
  //  already wait for the next tile's first DMAs, instead of across the K loop: the registers they would hold there are not free)
  using EpiT = Epi<T, 4, TC / 2, TC, NW, Cfg::EBT, (TC == 128 || WIN) && !(GNF && RES && TC == 256), TC == 128, RES, EB, TDX_DIRECT_EPI && (WIN || TC == 128), GNF>;
  static_assert(!GNF || (EpiT::DIRECT && WIN && TR == 256 && sizeof(T) == 2), "fused GroupNorm: bf16 window kernels with 256-row tiles");
  const uint32_t gb_lds = lds_base + (uint32_t)(W_BASE + 2 * kWinBytes + 64);
  if constexpr (GNF) {   // gamma | beta of every output channel, once per workgroup (before any DMA is in flight)
    typedef __attribute__((address_space(3))) float lds_f32_t;
    for (int i = tid; i < p.cout; i += TR * 2) {
      *reinterpret_cast<lds_f32_t*>(gb_lds + i * 4) = p.gn_gamma[i];
      *reinterpret_cast<lds_f32_t*>(gb_lds + 4096 + i * 4) = p.gn_beta[i];
    }
  }
  // ---- loader state (always one K-step ahead of the MFMAs, also across tile boundaries) ------------------------
  int ld_n0 = 0;
  u32x4 rsrc_a, rsrc_b;
  // per-lane DMA offsets are kept as ONE base VGPR plus a wave-uniform stride per 8-row group (registers are the
  // scarce resource of this kernel): rows past the end of a buffer fall out of the descriptor's range by themselves
  // (direct epilogue: LDS row x of a weight stage holds weight row perm_row(x), see Epi::finish_direct)
  const uint32_t voff_a0 = (uint32_t)(EpiT::DIRECT ? EpiT::perm_row(8 * wave + r8) : 8 * wave + r8) * (uint32_t)(p.ktot * ES) + chunk_off;
  const uint32_t stride_a = (uint32_t)(8 * NW) * (uint32_t)(p.ktot * ES);
  uint32_t soff_a = 0;
  int ld_seg = 0, ld_ci = 0, ld_tap = 0, ld_taps = 1, ld_cin = 0;
  uint32_t voff_b[3][WIN ? 1 : GB];
  int rb_b[WIN ? 1 : GB], rb_l[WIN ? 1 : GB];
  uint32_t voff_w0 = 0, stride_w = 0;

  auto issue_a = [&](int buf) {
    const uint32_t a = lds_base + buf * A_STAGE + wave * 1024;
    const uint32_t so = __builtin_amdgcn_readfirstlane(soff_a);
#pragma unroll
    for (int i = 0; i < GA; ++i) dma16(a + i * NW * 1024, voff_a0 + i * stride_a, rsrc_a, so);
    soff_a += kRowBytes;
  };
  // the same stage, one DMA per call (spread over the MFMA rows of a K-step); the last piece advances the K offset
  auto issue_a_piece = [&](int buf, int i) {
    const uint32_t a = lds_base + buf * A_STAGE + wave * 1024;
    const uint32_t so = __builtin_amdgcn_readfirstlane(soff_a);
    dma16(a + i * NW * 1024, voff_a0 + i * stride_a, rsrc_a, so);
    if (i == GA - 1) soff_a += kRowBytes;
  };
  auto ld_setup = [&]() {  // per segment
    const SegDev& s = p.seg[ld_seg];
    ld_taps = s.taps;
    ld_cin = s.cin;
    const int pad = s.taps == 3 ? 1 : 0;
    // (recomputed from the lane index here, once per segment, rather than kept in a register across the K loop)
    const int lane_o = EpiT::fresh_lane();
    const uint32_t chunk_off = (uint32_t)(((lane_o & 7) ^ (lane_o >> 3)) << 4);
    if constexpr (WIN) {
      const int base_row = ld_n0 - pad > 0 ? ld_n0 - pad : 0;
      const size_t base = ((size_t)base_row * s.src_ld + s.ch_off) * ES;
      const size_t left = (size_t)p.N * s.src_ld * ES - base;
      rsrc_b = make_srd(s.src + base, left > 0x7fffffffull ? 0x7fffffffu : (uint32_t)left);
      // window row j of the tile = source row n0 - pad + j; row -1 (first tile) wraps to a huge offset, rows >= N lie
      // beyond num_records: both are out of range for the DMA, which then writes zeros
      voff_w0 = (uint32_t)((ld_n0 - pad - base_row + 8 * wave + (lane_o >> 3)) * s.src_ld * ES) + chunk_off;
      stride_w = (uint32_t)(8 * NW * s.src_ld * ES);
    } else {
      const int b0 = fdiv(ld_n0, p.div_m, p.div_k);
      const size_t base = ((size_t)b0 * s.lin * s.src_ld + s.ch_off) * ES;
      const size_t left = (size_t)p.B * s.lin * s.src_ld * ES - base;
      rsrc_b = make_srd(s.src + base, left > 0x7fffffffull ? 0x7fffffffu : (uint32_t)left);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < GB; ++i) {
          const int u = rb_l[i] * s.stride + t - pad;
          const bool ok = t < s.taps && rb_b[i] >= 0 && u >= 0 && u < s.lvirt;
          const int uu = ok ? u : 0;
          const int srow = s.up_scale > 0.f ? min((int)floorf((float)uu * s.up_scale), s.lin - 1) : uu;
          voff_b[t][i] = ok ? (uint32_t)((rb_b[i] * s.lin + srow) * s.src_ld) * ES + chunk_off : kOob;
        }
    }
  };
  auto loader_begin = [&](int n0, int co0) {  // per tile
    ld_n0 = n0;
    const int rows_a = p.cout - co0 < TC ? p.cout - co0 : TC;
    rsrc_a = make_srd(p.w + (size_t)co0 * p.ktot * ES, (uint32_t)rows_a * (uint32_t)p.ktot * ES);
    soff_a = 0;
    ld_seg = ld_ci = ld_tap = 0;
    if constexpr (!WIN) {
      const int b0 = fdiv(n0, p.div_m, p.div_k);
      const int r8_o = EpiT::fresh_lane() >> 3;
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        const int i_loc = 8 * (wave + NW * i) + r8_o;
        const int n = n0 + i_loc;
        const int b = fdiv(n, p.div_m, p.div_k);
        rb_b[i] = (n < p.N && i_loc < p.tile_rows) ? b - b0 : -1;
        rb_l[i] = n - b * p.lout;
      }
    }
    ld_setup();
  };
  auto issue_b_gather = [&](int buf) {
    const uint32_t b = lds_base + buf * A_STAGE + A_BYTES + wave * 1024;
    const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)ld_ci * ES);
    with_tap(ld_tap, [&](auto tc) {
      constexpr int TAP = decltype(tc)::value;
#pragma unroll
      for (int i = 0; i < (WIN ? 0 : GB); ++i) dma16(b + i * NW * 1024, voff_b[TAP][i], rsrc_b, so);
    });
    if (++ld_tap >= ld_taps) {
      ld_tap = 0;
      ld_ci += BK;
      if (ld_ci >= ld_cin) {
        ld_ci = 0;
        ++ld_seg;
        if (ld_seg < p.nseg) ld_setup();
      }
    }
  };
  // one window DMA per call (i = 0 .. GW - 1); the last piece moves the loader on to the next channel block
  auto issue_window_piece = [&](int wbuf, int i) {
    const uint32_t w = lds_base + W_BASE + wbuf * kWinBytes + wave * 1024;
    const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)ld_ci * ES);
    if (wave + NW * i < kWinGroups) dma16(w + i * NW * 1024, voff_w0 + i * stride_w, rsrc_b, so);
    if (i == GW - 1) {
      ld_ci += BK;
      if (ld_ci >= ld_cin) {
        ld_ci = 0;
        ++ld_seg;
        if (ld_seg < p.nseg) ld_setup();
      }
    }
  };
  auto issue_window = [&](int wbuf) {
    const uint32_t w = lds_base + W_BASE + wbuf * kWinBytes + wave * 1024;
    const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)ld_ci * ES);
#pragma unroll
    for (int i = 0; i < (WIN ? GW : 0); ++i)
      if (wave + NW * i < kWinGroups) dma16(w + i * NW * 1024, voff_w0 + i * stride_w, rsrc_b, so);
    ld_ci += BK;
    if (ld_ci >= ld_cin) {
      ld_ci = 0;
      ++ld_seg;
      if (ld_seg < p.nseg) ld_setup();
    }
  };

  // ---- MFMA state -------------------------------------------------------------------------------------------------
  const int a_frag = (wc * (TC / 2) + fr) * kRowBytes;
  const int sw = fr & 7;
  f32x4 acc[MI][NI];
  int cp_seg = 0, cp_ci = 0, cp_tap = 0, cp_taps = 1, cp_cin = 0;
  uint32_t wmask = 0, wm3 = 0, wm1 = 0;  // validity bits per (sub-tile, tap) of the current segment / of 3-tap and 1-tap segments
  auto cp_setup = [&]() {
    const SegDev& s = p.seg[cp_seg];
    cp_taps = s.taps;
    cp_cin = s.cin;
    wmask = s.taps == 3 ? wm3 : wm1;
  };
  auto compute_begin = [&](int n0) {  // per tile
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    cp_seg = cp_ci = cp_tap = 0;
    if constexpr (WIN) {
      const int fr_o = EpiT::fresh_lane() & 15;  // (keeps the per-tile index math out of the registers that live across the K loop)
      wm3 = wm1 = 0;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int i_loc = wr * 64 + ni * 16 + fr_o;
        const int n = n0 + i_loc;
        const bool row_ok = n < p.N && i_loc < p.tile_rows;
        const int l = n - fdiv(n, p.div_m, p.div_k) * p.lout;  // position inside the trajectory
        wm1 |= (row_ok ? 1u : 0u) << (ni * 3);
        wm3 |= ((row_ok && l > 0 ? 1u : 0u) | (row_ok ? 2u : 0u) | (row_ok && l + 1 < p.lout ? 4u : 0u)) << (ni * 3);
      }
    }
    cp_setup();
  };
  auto compute = [&](int abuf, int wbuf, auto&& mid) {
    const char* a = smem + abuf * A_STAGE;
    const char* b = WIN ? smem + W_BASE + wbuf * kWinBytes : a + A_BYTES;
    int boff[NI];
    const int shift = WIN ? (cp_taps == 3 ? cp_tap : 0) : 0;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int wrow = wr * 64 + ni * 16 + fr + shift;
      boff[ni] = wrow * kRowBytes + ((fq ^ (wrow & 7)) << 4);
      if constexpr (WIN) boff[ni] = ((wmask >> (ni * 3 + cp_tap)) & 1u) ? boff[ni] : kWinGroups * 8 * kRowBytes;
    }
    // Fragment reads are software-pipelined by hand: the weight fragment of sub-tile row t + 2 and the activation
    // fragments of the second K half are requested while the MFMAs of row t run (left to itself hipcc reads one
    // weight fragment, waits for it, issues its four MFMAs, and so exposes an LDS round trip per four MFMAs).
    constexpr int NT2 = 2 * MI;  // (kk, mi) pairs
    auto a_addr = [&](int t) { return a + a_frag + (t % MI) * 16 * kRowBytes + (((((t / MI) << 2) + fq) ^ sw) << 4); };
    u32x4 bf[2][NI], af[3];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bf[0][ni] = *reinterpret_cast<const u32x4*>(b + boff[ni]);
    af[0] = *reinterpret_cast<const u32x4*>(a_addr(0));
    af[1] = *reinterpret_cast<const u32x4*>(a_addr(1));
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      // The step's DMA issue sits behind the first row of MFMAs: right after the barrier the MFMA pipe is empty, and
      // the DMA instructions (m0 set-up, s_nop, issue) would keep it empty for their whole issue time.
      mid(t);
      if (t + 2 < NT2) af[(t + 2) % 3] = *reinterpret_cast<const u32x4*>(a_addr(t + 2));
      if (t < NI) bf[1][t] = *reinterpret_cast<const u32x4*>(b + (boff[t] ^ 64));  // second K half, one fragment per row of the first
      __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs of this row (the scheduler would sink them to their uses)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) Mma<T>::run(af[t % 3], bf[t / MI][ni], acc[t % MI][ni]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (WIN) {
    if (tid < 16)  // the two zero rows (never touched by the DMA, nor by the epilogue scratch)
      *reinterpret_cast<u32x4*>(smem + W_BASE + (tid >> 3) * kWinBytes + kWinGroups * 8 * kRowBytes + (tid & 7) * 16) = (u32x4){0u, 0u, 0u, 0u};
  }

  // ---- the pipeline: g counts K-steps over all tiles (weight stage g & 1), blk counts channel blocks (window blk & 1) ----
  static_assert(EpiT::NST == Cfg::NST, "store count must match StreamCfg::NST");
  EpiT epi;
#if TDX_STAMPS
  const uint32_t st_base = lds_base + (uint32_t)(WIN ? W_BASE + 2 * kWinBytes : 2 * A_STAGE);
  if (tid < 16) *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(st_base + tid * 4) = 0u;
  auto st_now = [&]() -> uint32_t { return (uint32_t)__builtin_amdgcn_s_memtime(); };
  auto st_add = [&](int k, uint32_t d) {
    if (tid == 0) *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(st_base + k * 4) += d;
  };
  // in-kernel clock: shader cycles (s_memtime) against the constant 100 MHz counter (s_memrealtime) over the whole kernel
  const uint32_t clk_t0 = st_now(), clk_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
  int g = 0, blk = 0;
  bool counted = false;  // the previous tile's epilogue stores are this wave's youngest vector-memory operations
  loader_begin(cp_n0, cp_co0);
  if constexpr (WIN) issue_window(0);
  issue_a(0);
  if constexpr (!WIN) issue_b_gather(0);
  // vmcnt counts in issue order.  A window is issued AFTER the weight stage of its step, so that the next step can wait
  // for that stage alone (vmcnt = number of window DMAs) and the window gets two K-steps to arrive instead of one.
  const int win_dmas = GW - ((wave + NW * (GW - 1) < kWinGroups) ? 0 : 1);  // window DMAs this wave issues per window
  int pend_w = 0;                                                            // ... of them issued in the previous K-step
  auto wait_vm = [&](int n) {
    if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == GW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW - 1) : "memory");
  };
  auto step = [&](auto first_c, auto last_c) {
    constexpr bool FIRST = decltype(first_c)::value != 0, LAST = decltype(last_c)::value != 0;
#if TDX_STAMPS
    const uint32_t st0 = st_now();
#endif
    // first step of a tile: DMAs landed, stores may still drain (GroupNorm epilogue: a raw and a normalised store per item, the raw ones
    // with every lane out of range when nobody wants the raw rows: the count stays a constant of the kernel)
    if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((GNF ? 2 : 1) * Cfg::NST) : "memory");
    else wait_vm(cp_tap == 0 ? 0 : pend_w);  // (a block's first step needs its window)
    counted = false;
#if TDX_STAMPS
    const uint32_t st1 = st_now();
#endif
    __syncthreads();
#if TDX_STAMPS
    const uint32_t st2 = st_now();
#endif
    const int g_now = g, blk_now = blk, tap_now = cp_tap;
    pend_w = 0;
    const bool win_go = WIN && tap_now == 0 && ld_seg < p.nseg;   // (ld_seg moves on only with the window's last piece)
    compute(g & 1, blk & 1, [&](int t) {
      // The step's DMA issue sits behind the first row of MFMAs, and (SPREAD) the weight stage goes out one DMA per MFMA row:
      // a burst of DMA returns into the LDS delays the just-in-time fragment reads of both waves of a SIMD at once.
      if constexpr (SPREAD && !LAST) {
        if (t >= 1 && t <= GA) issue_a_piece((g_now + 1) & 1, t - 1);
      }
      if constexpr (SPREAD) {   // window pieces behind the weight pieces, the rest of them with the last MFMA row
        constexpr int NT2 = 2 * MI;
        if (t > GA && win_go) {
          const int i0 = t - GA - 1;
          if (t < NT2 - 1) {
            if (i0 < GW) issue_window_piece((blk_now + 1) & 1, i0);
          } else {
#pragma unroll
            for (int i = 0; i < GW; ++i)
              if (i >= i0) issue_window_piece((blk_now + 1) & 1, i);
          }
          if (i0 == 0) pend_w = win_dmas;
        }
      }
      if (t != (SPREAD ? GA + 1 : 1)) return;
      if constexpr (!LAST) {
        if constexpr (!SPREAD) issue_a((g_now + 1) & 1);
        if constexpr (!WIN) issue_b_gather((g_now + 1) & 1);
      }
      if constexpr (WIN && !SPREAD) {
        if (win_go) {
          issue_window((blk_now + 1) & 1);
          pend_w = win_dmas;
        }
      }
      // the epilogue's global reads go out with the tile's FIRST K-step: they land under the rest of the K loop
      if constexpr (FIRST && EpiT::PRE) epi.prefetch(p, wave, wr, wc, lane, cp_n0, cp_co0);
    });
#if TDX_STAMPS
    {
      const uint32_t st3 = st_now();
      constexpr int SB = FIRST ? 0 : 6;
      st_add(SB, st1 - st0);
      st_add(SB + 1, st2 - st1);
      st_add(SB + 2, st3 - st2);
    }
#endif
    ++g;
    if (++cp_tap >= cp_taps) {
      cp_tap = 0;
      ++blk;
      cp_ci += BK;
      if (cp_ci >= cp_cin) {
        cp_ci = 0;
        ++cp_seg;
        if (cp_seg < p.nseg) cp_setup();
      }
    }
  };
  for (;;) {
    compute_begin(cp_n0);
    if constexpr (EpiT::PRE) {
      if (p.nsteps == 1) {
        step(TapC<1>{}, TapC<1>{});
      } else {
        step(TapC<1>{}, TapC<0>{});
        for (int s = 1; s + 1 < p.nsteps; ++s) step(TapC<0>{}, TapC<0>{});
        step(TapC<0>{}, TapC<1>{});
      }
    } else {
      for (int s = 0; s + 1 < p.nsteps; ++s) step(TapC<0>{}, TapC<0>{});
      step(TapC<0>{}, TapC<1>{});
    }
#if TDX_STAMPS
    const uint32_t sb0 = st_now();
#endif
    if constexpr (EpiT::PRE) epi.pin(p);
    // tile boundary: the loader goes ahead, then the epilogue runs in the buffers the last K-step released
    int nx_n0, nx_co0;
    const int jn = jt + nslot;
    const bool more = tile_of(jn, nx_n0, nx_co0);
    if (more) {
      loader_begin(nx_n0, nx_co0);
      if constexpr (WIN) issue_window(blk & 1);
      issue_a(g & 1);
      if constexpr (!WIN) issue_b_gather(g & 1);
    }
    const uint32_t scr_a = lds_base + ((g - 1) & 1) * A_STAGE;
    uint32_t patch, eb;
    if constexpr (WIN) {
      const uint32_t scr_w = lds_base + W_BASE + ((blk - 1) & 1) * kWinBytes;
      patch = wave < Cfg::NPW ? scr_w + wave * Cfg::PATCH : scr_a + (wave - Cfg::NPW) * Cfg::PATCH;
      eb = scr_a + (NW - Cfg::NPW) * Cfg::PATCH;
    } else {
      patch = scr_a + wave * Cfg::PATCH;
      eb = scr_a + NW * Cfg::PATCH;
    }
#if TDX_STAMPS
    const uint32_t sb1 = st_now();
#endif
    if constexpr (GNF) {
      const uint32_t scr_w = lds_base + W_BASE + ((blk - 1) & 1) * kWinBytes;
      epi.finish_direct_gn(p, acc, eb, scr_w, gb_lds + 8192u, gb_lds, wave, wr, wc, lane, cp_n0, cp_co0);
    } else {
      epi.finish(p, acc, patch, eb, wave, wr, wc, lane, cp_n0, cp_co0);
    }
#if TDX_STAMPS
    {
      const uint32_t sb2 = st_now();
      st_add(9, sb1 - sb0);
      st_add(11, sb2 - sb1);
      st_add(15, 1u);
    }
#endif
    counted = !(EpiT::DIRECT && (TDX_ABL & 3) != 0);
    if (!more) break;
    jt = jn;
    cp_n0 = nx_n0;
    cp_co0 = nx_co0;
  }
#if TDX_STAMPS
  st_add(12, st_now() - clk_t0);
  st_add(13, (uint32_t)__builtin_amdgcn_s_memrealtime() - clk_r0);
  __syncthreads();
  if (p.stamps && tid < 16) p.stamps[blockIdx.x * 16 + tid] = *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(st_base + tid * 4);
#endif
#endif
}

constexpr size_t dma_lds_bytes(int tc, int tr, bool win) {
  return win ? 2 * (size_t)tc * kRowBytes + 2 * (size_t)win_bytes(tr) : 2 * (size_t)(tr + tc) * kRowBytes;
}
// LDS the GroupNorm epilogue needs (256-row tiles only): transpose patches + per-row partials + mean/var tables + embed-bias rows
inline size_t gn_lds_bytes(int tc, int cpg, int L) {
  const size_t gt = tc / cpg;
  return 8 * 16 * (size_t)(tc / 2 * 4 + 16) + 256 * gt * 4 + 2 * (size_t)(256 / L + 1) * gt * 4 + (size_t)kEbMaxTraj * tc * 4;
}

template <typename T, int TC, int TR, bool WIN, bool GN>
int launch_dma(const ConvArgs& a, hipStream_t st) {
  const int nco = (a.cout + TC - 1) / TC, nrow = a.row_tiles;
  dim3 grid((unsigned)(nco * ((nrow + 7) / 8) * 8));
  size_t lds = dma_lds_bytes(TC, TR, WIN);
  static bool attr_done_dev[kTdxMaxDevices] = {};   // (per device: hipFuncSetAttribute acts on the current device's copy of the kernel)
  bool& attr_done = attr_done_dev[tdx_device_slot()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_dma_kernel<T, TC, TR, WIN, GN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      TDX_FAIL(TDX_ERR_HIP, "conv_gemm_dma: cannot reserve %zu bytes of LDS", lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_gemm_dma_kernel<T, TC, TR, WIN, GN>), grid, dim3(TR * 2), lds, st, a);
  TDX_CHECK_LAUNCH("conv_gemm_dma_kernel");
  return TDX_OK;
}

template <typename T>
int dispatch_dma(const ConvArgs& a, hipStream_t st, int tc, int tr, bool win, bool gn) {
  if (gn) {  // GroupNorm epilogue: 256-row tiles only
    if (win) return tc == 256 ? launch_dma<T, 256, 256, true, true>(a, st) : launch_dma<T, 128, 256, true, true>(a, st);
    return tc == 256 ? launch_dma<T, 256, 256, false, true>(a, st) : launch_dma<T, 128, 256, false, true>(a, st);
  }
  if (tr == 128) return win ? launch_dma<T, 128, 128, true, false>(a, st) : launch_dma<T, 128, 128, false, false>(a, st);
  if (win) return tc == 256 ? launch_dma<T, 256, 256, true, false>(a, st) : launch_dma<T, 128, 256, true, false>(a, st);
  return tc == 256 ? launch_dma<T, 256, 256, false, false>(a, st) : launch_dma<T, 128, 256, false, false>(a, st);
}

template <typename T, int TC, int TR, bool WIN, int MODE, bool GNF = false>
int launch_stream2(const ConvArgs& a, hipStream_t st, int cus) {
  const int nco = (a.cout + TC - 1) / TC;
  const int jt_end = nco * ((a.row_tiles + 7) / 8);
  const int per_xcd = (cus / 8) * (TR == 128 ? 2 : 1);  // resident workgroups per XCD
  const int nslot = jt_end < per_xcd ? jt_end : per_xcd;
  size_t lds = dma_lds_bytes(TC, TR, WIN);
  if (TDX_STAMPS) lds += 64;
  if (GNF) lds = dma_lds_bytes(TC, TR, WIN) + kGnLds;
  static bool attr_done_dev[kTdxMaxDevices] = {};   // (per device: hipFuncSetAttribute acts on the current device's copy of the kernel)
  bool& attr_done = attr_done_dev[tdx_device_slot()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_stream_kernel<T, TC, TR, WIN, MODE, GNF>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      TDX_FAIL(TDX_ERR_HIP, "conv_gemm_stream: cannot reserve %zu bytes of LDS", lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_gemm_stream_kernel<T, TC, TR, WIN, MODE, GNF>), dim3((unsigned)(8 * nslot)), dim3(TR * 2), lds, st, a);
  TDX_CHECK_LAUNCH("conv_gemm_stream_kernel");
  return TDX_OK;
}

// Shapes the persistent kernel is built for.  The rest (no register room: hipcc would spill inside the K loop, see
// scripts/check_spills.py) run on the one-tile-per-workgroup kernel, with identical results:
//   embed rows together with a residual; a residual on gathered 256-channel tiles; fp32 256-channel tiles with a residual,
//   or gathered with embed rows.
template <typename T>
constexpr bool stream_ok(int tc, bool win, int mode) {
  if (mode == 3) return false;
  if (tc == 256 && mode == 2 && (!win || sizeof(T) == 4)) return false;
  if (tc == 256 && mode == 1 && !win && sizeof(T) == 4) return false;
  return true;
}
inline int conv_mode(const ConvArgs& a) { return (a.res ? 2 : 0) | (a.ebias ? 1 : 0); }

template <typename T, int TC, int TR, bool WIN>
int launch_stream(const ConvArgs& a, hipStream_t st, int cus) {
  const int mode = conv_mode(a);
  if constexpr (stream_ok<T>(TC, WIN, 2)) if (mode == 2) return launch_stream2<T, TC, TR, WIN, 2>(a, st, cus);
  if constexpr (stream_ok<T>(TC, WIN, 1)) if (mode == 1) return launch_stream2<T, TC, TR, WIN, 1>(a, st, cus);
  if (mode == 0) return launch_stream2<T, TC, TR, WIN, 0>(a, st, cus);
  TDX_FAIL(TDX_ERR_UNSUPPORTED, "conv_gemm_stream: internal dispatch error (tc=%d win=%d mode=%d)", TC, (int)WIN, mode);
}

// Persistent kernel with the fused GroupNorm epilogue: bf16, window mode, 256-row tiles (plan_conv / gn_stream_ok decide).
template <int TC>
int launch_stream_gn(const ConvArgs& a, hipStream_t st, int cus) {
  const int mode = conv_mode(a);
  if (mode == 0) return launch_stream2<__bf16, TC, 256, true, 0, true>(a, st, cus);
  if (mode == 1) return launch_stream2<__bf16, TC, 256, true, 1, true>(a, st, cus);
  if (mode == 2) return launch_stream2<__bf16, TC, 256, true, 2, true>(a, st, cus);
  TDX_FAIL(TDX_ERR_UNSUPPORTED, "conv_gemm_stream (GroupNorm epilogue): embed rows together with a residual");
}

template <typename T>
int dispatch_stream(const ConvArgs& a, hipStream_t st, int tc, int tr, bool win, int cus) {
  if (tr == 128) return win ? launch_stream<T, 128, 128, true>(a, st, cus) : launch_stream<T, 128, 128, false>(a, st, cus);
  if (win) return tc == 256 ? launch_stream<T, 256, 256, true>(a, st, cus) : launch_stream<T, 128, 256, true>(a, st, cus);
  return tc == 256 ? launch_stream<T, 256, 256, false>(a, st, cus) : launch_stream<T, 128, 256, false>(a, st, cus);
}

}  // namespace

// Validate a descriptor, translate it to kernel arguments and choose the kernel.  kind: 0 = register-staged
// 128x128, 1 = LDS-DMA gather, 2 = LDS-DMA window; tc = channel tile of the DMA kernels; gn = fused GroupNorm.
struct Choice { int kind, tc, tr; bool gn; bool gn_stream = false; };

static int plan_conv(const tdx_conv_desc* d, ConvArgs& a, Choice& ch) {
  TDX_CHECK_ARG(d != nullptr, "tdx_conv_gemm: null descriptor");
  TDX_CHECK_ARG(d->dtype == TDX_F32 || d->dtype == TDX_BF16, "tdx_conv_gemm: bad dtype %d", d->dtype);
  const int es = d->dtype == TDX_F32 ? 4 : 2;
  const int vec = 16 / es, bk = kRowBytes / es;
  const bool want_gn = d->gn_gamma != nullptr;
  TDX_CHECK_ARG(d->nseg >= 1 && d->nseg <= 4, "tdx_conv_gemm: nseg=%d not in [1,4]", d->nseg);
  TDX_CHECK_ARG(d->B > 0 && d->lout > 0 && d->cout > 0, "tdx_conv_gemm: empty problem B=%d lout=%d cout=%d", d->B, d->lout, d->cout);
  TDX_CHECK_ARG((int64_t)d->B * d->lout < (1ll << 31), "tdx_conv_gemm: B*lout overflows int32");
  TDX_CHECK_ARG(d->w && (d->out || want_gn), "tdx_conv_gemm: null w/out");
  TDX_CHECK_ARG(d->cout % 4 == 0, "tdx_conv_gemm: cout=%d must be a multiple of 4", d->cout);
  TDX_CHECK_ARG(!d->out || (d->out_off % 4 == 0 && d->out_ld % 4 == 0 && d->out_off + d->cout <= d->out_ld),
                "tdx_conv_gemm: out_off/out_ld must be multiples of 4 and fit (cout=%d off=%d ld=%d)", d->cout, d->out_off, d->out_ld);
  TDX_CHECK_ARG(!d->res || (d->res_ld % 4 == 0 && d->res_ld >= d->cout), "tdx_conv_gemm: bad res_ld %d", d->res_ld);
  TDX_CHECK_ARG(!(d->ebias || d->tbias) || (d->ebias_off % 4 == 0 && d->ebias_ld % 4 == 0), "tdx_conv_gemm: ebias_off/ld must be multiples of 4");
  TDX_CHECK_ARG(d->act >= TDX_ACT_NONE && d->act <= TDX_ACT_SWISH, "tdx_conv_gemm: bad act %d", d->act);
  a = ConvArgs{};
  int k = 0, steps = 0;
  for (int i = 0; i < d->nseg; ++i) {
    const tdx_kseg& s = d->seg[i];
    TDX_CHECK_ARG(s.src != nullptr, "tdx_conv_gemm: seg %d null src", i);
    TDX_CHECK_ARG(s.taps == 1 || s.taps == 3, "tdx_conv_gemm: seg %d taps=%d (1 or 3)", i, s.taps);
    TDX_CHECK_ARG(s.stride == 1 || s.stride == 2, "tdx_conv_gemm: seg %d stride=%d (1 or 2)", i, s.stride);
    TDX_CHECK_ARG(s.cin > 0 && s.cin % vec == 0 && s.ch_off % vec == 0 && s.src_ld % vec == 0 && s.ch_off + s.cin <= s.src_ld,
                  "tdx_conv_gemm: seg %d channels (cin=%d off=%d ld=%d) must be multiples of %d and fit", i, s.cin, s.ch_off, s.src_ld, vec);
    TDX_CHECK_ARG(s.cin * es <= (int)sizeof(uint32_t) * 4096, "tdx_conv_gemm: seg %d cin=%d too wide", i, s.cin);
    TDX_CHECK_ARG(s.lin > 0 && s.lvirt > 0, "tdx_conv_gemm: seg %d bad lengths", i);
    TDX_CHECK_ARG(s.lvirt == s.lin || s.up_scale > 0.f, "tdx_conv_gemm: seg %d lvirt != lin needs up_scale = (float)lin / lvirt", i);
    const int pad = s.taps == 3 ? 1 : 0;
    const int lo = (s.lvirt + 2 * pad - s.taps) / s.stride + 1;
    TDX_CHECK_ARG(lo == d->lout, "tdx_conv_gemm: seg %d yields %d output rows per trajectory, lout=%d", i, lo, d->lout);
    a.seg[i] = SegDev{(const char*)s.src, s.lvirt != s.lin ? s.up_scale : 0.f, s.src_ld, s.ch_off, s.cin, s.taps, s.stride, s.lin, s.lvirt, k};
    k += s.taps * s.cin;
    steps += s.taps * ((s.cin + bk - 1) / bk);
  }
  a.nseg = d->nseg;
  a.nsteps = steps;
  a.w = (const char*)d->w;
  a.ktot = k;
  a.bias = d->bias;
  a.ebias = d->ebias;
  a.tbias = d->tbias;
  a.ebias_ld = d->ebias_ld;
  a.ebias_off = d->ebias_off;
  a.res = (const char*)d->res;
  a.res_ld = d->res_ld;
  a.act = d->act;
  a.out = (char*)d->out;
  a.out_ld = d->out_ld;
  a.out_off = d->out_off;
  a.N = d->B * d->lout;
  a.lout = d->lout;
  a.cout = d->cout;
  a.B = d->B;
  a.debug = g_debug;
  a.partial = nullptr;
#if TDX_STAMPS
  a.stamps = g_stamps;
#endif
  a.ksplit = 1;
  a.steps_per_split = 0;
  a.tile_rows = kDmaRows;
  a.gn_L = d->lout;
  a.gn_cpg = vec;
  // LDS-DMA kernels: whole 128-byte channel blocks only, 32-bit in-tile offsets, 16-byte epilogue vectors, and
  // enough 256-row tiles to fill the chip.  Window variant: every segment stride 1 on an un-resampled source.
  bool dma_ok = !(g_force & 1), win_ok = !(g_force & 2);
  for (int i = 0; i < d->nseg; ++i) {
    const tdx_kseg& s = d->seg[i];
    dma_ok = dma_ok && s.cin % bk == 0 && ((int64_t)(kDmaRows / d->lout + 2) * s.lin * s.src_ld * es < (1ll << 31));
    win_ok = win_ok && s.stride == 1 && s.lvirt == s.lin && s.lin == d->lout && ((int64_t)(kDmaRows + 16) * s.src_ld * es < (1ll << 31));
  }
  dma_ok = dma_ok && (int64_t)kDmaRows * k * es < (1ll << 31);
  dma_ok = dma_ok && d->cout % vec == 0 && (!d->out || (d->out_off % vec == 0 && d->out_ld % vec == 0)) && (!d->res || d->res_ld % vec == 0);
  dma_ok = dma_ok && (int64_t)(kDmaRows + 16) * d->out_ld * es < (1ll << 31);  // 32-bit in-tile store offsets
  int tc = (d->cout % 256 == 0 && !(g_force & 128)) ? 256 : 128;  // (128-wide tiles for Cout = 256 measured 17 % slower)
  int64_t tiles = (int64_t)((a.N + kDmaRows - 1) / kDmaRows) * ((d->cout + tc - 1) / tc);
  // Short trajectories under wide layers (the latent path: 3 .. 24 rows per trajectory, 256 .. 1024 channels): too few 256-wide
  // tiles for the LDS-DMA kernels, but 128 x 128 tiles (4x as many, two workgroups per CU) fill the chip.  Same K order per output.
  if (tc == 256 && tiles < 128 && !(g_force & 8)) {
    tc = 128;
    tiles *= 2;
  }
  // narrow outputs: 128-row tiles, two workgroups per CU (epilogue of one under the K loop of the other)
  const int tr = (tc == 128 && !(g_force & 8)) ? 128 : 256;
  ch = Choice{0, tc, tr, false};
  if (want_gn) {
    TDX_CHECK_ARG(d->gn_beta && d->gn_out, "tdx_conv_gemm: fused GroupNorm needs gn_beta and gn_out");
    TDX_CHECK_ARG(d->gn_act >= TDX_ACT_NONE && d->gn_act <= TDX_ACT_SWISH, "tdx_conv_gemm: bad gn_act %d", d->gn_act);
    TDX_CHECK_ARG(d->gn_cpg > 0 && d->cout % d->gn_cpg == 0, "tdx_conv_gemm: gn_cpg=%d must divide cout=%d", d->gn_cpg, d->cout);
    const int cpg = d->gn_cpg;
    bool ok = dma_ok && !(g_force & 4) && d->lout <= kDmaRows && cpg >= 4 && cpg >= vec / 2 && (cpg & (cpg - 1)) == 0 && cpg <= tc / 2 &&
              d->gn_off % vec == 0 && d->gn_ld % vec == 0 && d->gn_off + d->cout <= d->gn_ld;
    // Persistent form (Epi::finish_direct_gn): bf16 window convs, groups of 8 .. 64 channels inside one wave's half of the channel tile,
    // statistics tables that fit the scratch the K loop releases.  Everything else that is fusable takes the one-tile kernel.
    bool gs = ok && !(g_force & 32) && win_ok && d->dtype == TDX_BF16 && cpg >= 8 && cpg <= 64 && d->cout <= 1024 && !(d->res && d->ebias) &&
              d->gn_off % 8 == 0 && d->gn_ld % 8 == 0 && (int64_t)(kDmaRows + 16) * d->gn_ld * es < (1ll << 31);
    // (the statistics tables of a tile: judged for 256-channel tiles whatever the tile chosen below, so that the answer depends on the
    //  layer's shape alone and never on the batch size: batch shards must take the same arithmetic as the unsharded batch)
    gs = gs && (kDmaRows / d->lout) * (256 / cpg) <= kGnTabMax;
    if (gs) {
      // threads that share one (trajectory, group) row sum: a function of the trajectory length ALONE, so that the order of the
      // additions does not depend on the batch size or the tile shape (batch shards reproduce the unsharded run bit for bit)
      a.gn_tpe = d->lout >= 32 ? 8 : (d->lout >= 8 ? 4 : (d->lout >= 4 ? 2 : 1));
    }
    ok = ok && (gs || gn_lds_bytes(tc, cpg, d->lout) <= dma_lds_bytes(tc, 256, win_ok));
    if (!ok) TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_conv_gemm: this shape cannot fuse the GroupNorm (lout=%d cout=%d cpg=%d): run tdx_groupnorm_act", d->lout, d->cout, cpg);
    a.tile_rows = (kDmaRows / d->lout) * d->lout;
    a.gn_gamma = d->gn_gamma;
    a.gn_beta = d->gn_beta;
    a.gn_out = (char*)d->gn_out;
    a.gn_ld = d->gn_ld;
    a.gn_off = d->gn_off;
    a.gn_act = d->gn_act;
    a.gn_cpg = cpg;
    a.gn_eps = d->gn_eps;
    ch = Choice{win_ok ? 2 : 1, tc, 256, true, gs};
  } else if (dma_ok && tiles >= 128) {
    ch = Choice{win_ok ? 2 : 1, tc, tr, false};
    a.tile_rows = tr;
  }
  a.n_begin = 0;
  a.row_tiles = (a.N + a.tile_rows - 1) / a.tile_rows;
  int c = 0;
  while ((1ll << c) < d->lout) ++c;
  a.div_k = 32 + c;
  a.div_m = (uint64_t)((1ull << a.div_k) / (uint64_t)d->lout) + 1;
  return TDX_OK;
}

// Workgroup slots of 256-row tiles (one 8-wave workgroup per CU).
static int cu_count() {
  static int n_dev[kTdxMaxDevices] = {};   // per device
  int& n = n_dev[tdx_device_slot()];
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

template <typename T>
static int run_dma(ConvArgs& a, const Choice& ch, hipStream_t st) {
  const bool win = ch.kind == 2;
  if constexpr (sizeof(T) == 2) {
    if (ch.gn && ch.gn_stream) return ch.tc == 256 ? launch_stream_gn<256>(a, st, cu_count()) : launch_stream_gn<128>(a, st, cu_count());
  }
  // Persistent kernel for every tile shape; the general GroupNorm epilogue exists in the one-tile-per-workgroup form only.
  const bool stream = !ch.gn && !(g_force & 32);
  auto go = [&](const ConvArgs& x, int tc, int tr) {
    return stream && stream_ok<T>(tc, win, conv_mode(x)) ? dispatch_stream<T>(x, st, tc, tr, win, cu_count())
                                                         : dispatch_dma<T>(x, st, tc, tr, win, ch.gn);
  };
  // Tail split.  256-row tiles run one per CU in rounds of cu_count() tiles of equal duration, so a last round that is
  // only partly filled costs a whole round (e.g. 832 tiles = 3.25 rounds run as 4).  When the remainder is at most
  // half a round, its rows are given to a second launch of 128 x 128 tiles instead (4x as many, <= two per CU,
  // each about a fifth of a big tile's time).  Same K order per output element: results are bit-identical.
  if (ch.tr == 256 && !ch.gn && !(g_force & 16)) {
    const int cus = cu_count();
    const int nco = (a.cout + ch.tc - 1) / ch.tc;
    const int64_t tiles = (int64_t)a.row_tiles * nco;
    const int64_t full = tiles / cus * cus, rem = tiles - full;
    if (full > 0 && rem > 0 && rem <= g_tail_max && full % nco == 0) {
      ConvArgs m = a;
      m.row_tiles = (int32_t)(full / nco);
      int rc = go(m, ch.tc, 256);
      if (rc != TDX_OK) return rc;
      ConvArgs t = a;
      t.n_begin = m.row_tiles * kDmaRows;
      t.tile_rows = 128;
      t.row_tiles = (a.N - t.n_begin + 127) / 128;
      return go(t, 128, 128);
    }
  }
  return go(a, ch.tc, ch.tr);
}

#if TDX_STAMPS
extern "C" int tdx_conv_gemm_stamps(void* ptr) {
  g_stamps = (uint32_t*)ptr;
  return TDX_OK;
}
#endif

extern "C" int tdx_conv_gemm_gn_fusable(const tdx_conv_desc* d) {
  ConvArgs a;
  Choice ch;
  return d && d->gn_gamma && plan_conv(d, a, ch) == TDX_OK && ch.gn ? (ch.gn_stream ? 2 : 1) : 0;
}

extern "C" int tdx_conv_gemm(const tdx_conv_desc* d, void* stream) {
  ConvArgs a;
  Choice ch;
  const int rc = plan_conv(d, a, ch);
  if (rc != TDX_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  {  // 128 -> 128, k = 3 layers (and every descriptor that carries norm[] / stats_out): narrow_conv.hip
    int handled = 0;
    const int nrc = tdx_narrow_conv_try(d, st, cu_count(), &handled);
    if (nrc != TDX_OK || handled) return nrc;
  }
  if (ch.kind != 0) return d->dtype == TDX_F32 ? run_dma<float>(a, ch, st) : run_dma<__bf16>(a, ch, st);
  // Few tiles and a long K loop (small batches): cut K into slices that run as concurrent workgroups, if the caller gave room.
  const int64_t tiles128 = (int64_t)((a.N + 127) / 128) * ((a.cout + 127) / 128);
  if (d->ws && !(g_force & 64) && a.cout % 4 == 0 && tiles128 < 128 && a.nsteps >= 4 && ((uintptr_t)d->ws & 15) == 0) {
    const int64_t slice_bytes = (int64_t)a.N * a.cout * 4;
    int want = (int)std::min<int64_t>(a.nsteps / 2, (512 + tiles128 - 1) / tiles128);   // ~2 workgroups per CU, >= 2 K-steps each
    want = (int)std::min<int64_t>(want, d->ws_bytes / slice_bytes);
    if (want >= 2) {
      a.steps_per_split = (a.nsteps + want - 1) / want;
      a.ksplit = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;
      a.partial = (float*)d->ws;
    }
  }
  if (d->dtype == TDX_F32) return launch<float, 128, 128>(a, st);
  return launch<__bf16, 128, 128>(a, st);
}

// Test hooks (not part of the reference-facing surface).
// force bits: 1 = never use the LDS-DMA kernels, 2 = never use the window variant, 4 = report GroupNorm as not fusable,
//             8 = never use the 128-row (two workgroups per CU) tiles, 16 = no tail split, 32 = one tile per workgroup
//             (no persistent kernel), 64 = never split K, 128 = 128-channel tiles for every Cout;
//             bits 8..15, when non-zero: tail-split threshold in units of 2 tiles (ablation);
//             bit 16 = never take the narrow (128 -> 128, k = 3) kernel of narrow_conv.hip.
extern "C" int tdx_conv_gemm_norm_fusable(const tdx_conv_desc* d) {
  ConvArgs a;
  Choice ch;
  return d && plan_conv(d, a, ch) == TDX_OK && tdx_narrow_conv_ok(d) ? 1 : 0;
}

extern "C" int tdx_conv_gemm_force_small(int32_t bits) {
  tdx_narrow_conv_disable((bits >> 16) & 1);
  g_force = bits & 0xff;
  g_tail_max = (bits >> 8) & 0xff ? ((bits >> 8) & 0xff) * 2 : 128;
  return TDX_OK;
}
extern "C" int tdx_conv_gemm_debug(int32_t bits) {
  g_debug = bits;
  return TDX_OK;
}
