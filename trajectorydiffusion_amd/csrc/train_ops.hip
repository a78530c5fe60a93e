// Training-step building blocks (train.ipynb:198-209): eps-MSE loss and its gradient, column sums
// (bias gradients), convolution weight gradients, GroupNorm(+Swish) backward, fused AdamW.  fp32 arithmetic throughout;
// every reduction has a fixed order (two stages through a caller-provided workspace, no atomics): results are bitwise
// reproducible.  The step itself (saved activations, backward program) is assembled in trajectorydiffusion_amd/training.py.
#include <algorithm>

#include "tdx_common.h"

namespace {

constexpr int kRedBlocks = 1024;
int g_wgrad_force = 0;  // tests (tdx_conv_wgrad_debug): 1 = never take the large-tile weight-gradient kernel, 2 = one-thread-per-output slab sum,
                        // 4 = the former slab count (two rounds' worth), 8 = GroupNorm backward re-reads its rows in every sweep  // first-stage partials of the scalar / column reductions

__device__ __forceinline__ float block_sum_256(float v, float* red) {  // fixed-order tree: lanes, then the four waves
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

// ---- loss = mean((pred - target)^2), dpred = 2 (pred - target) / n  (torch.nn.MSELoss, train.ipynb:205) ----------------
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                         float* __restrict__ dpred, float* __restrict__ part, long n, float scale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = pred[i] - target[i];
    acc += d * d;
    if (dpred) dpred[i] = d * scale;
  }
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out, float scale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = t * scale;
}

// ---- column sums of a row-major (n, c) matrix: bias gradients, GroupNorm dgamma / dbeta ------------------------------------
// stage 1: block (row slab, column window) -> part[slab][c]; stage 2: fixed-order sum over slabs.
// fp32 with c % 4 == 0 (every call of the training step): a lane owns 4 consecutive columns (16-byte loads, a wave reads
// 1 KB of contiguous row data per load), four loads in flight per lane; lanes that share columns are combined with xor
// shuffles, the four waves through LDS.  Anything else: the scalar kernel below.
template <int UNROLL>
__global__ __launch_bounds__(256) void colsum_partial_v4_kernel(const float* __restrict__ x, int ld, int n, int c, int lpr,
                                                               float* __restrict__ part, int rows_per_slab, const float* __restrict__ x2) {
  __shared__ f32x4 red[4][64];
  if (blockIdx.z) {   // second matrix of a pair (tdx_colsum_pair): its partials follow the first one's
    x = x2;
    part += (size_t)gridDim.x * c;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rpw = 64 / lpr;                                  // rows one wave covers per load
  const int col = blockIdx.y * 256 + (lane % lpr) * 4;       // (lpr = min(64, c / 4): windows only exist when lpr == 64)
  const int r0 = blockIdx.x * rows_per_slab, r1 = min(n, r0 + rows_per_slab);
  const int rstep = 4 * rpw;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (col < c) {
    int r = r0 + wave * rpw + lane / lpr;
    for (; r + (UNROLL - 1) * rstep < r1; r += UNROLL * rstep) {
      f32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(r + u * rstep) * ld + col);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; r < r1; r += rstep) acc += *reinterpret_cast<const f32x4*>(x + (size_t)r * ld + col);
  }
  for (int o = 32; o >= lpr; o >>= 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && lane < lpr && col < c) {
    const f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * c + col) = t;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int ld, int n, int c, float* __restrict__ part, int slabs) {
  __shared__ float red[4][64];
  const int col = blockIdx.y * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int per = (n + slabs - 1) / slabs;
  const int r0 = blockIdx.x * per, r1 = min(n, r0 + per);
  float acc = 0.f;
  if (col < c)
    for (int r = r0 + sub; r < r1; r += 4) acc += Elem<T>::ld(x + (size_t)r * ld + col);
  red[sub][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sub == 0 && col < c) part[(size_t)blockIdx.x * c + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// block = 16 columns x 16 slab classes; class q adds slabs q, q + 16, ... in order, then the classes are added in order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int slabs, int c, float* __restrict__ out, int accumulate,
                                                          float* __restrict__ out2 = nullptr) {
  __shared__ float red[16][17];
  if (blockIdx.y) {
    part += (size_t)slabs * c;
    out = out2;
  }
  const int cj = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + cj;
  float acc = 0.f;
  if (col < c) {
    int s = sub;
    for (; s + 48 < slabs; s += 64) {
      const float a = part[(size_t)s * c + col], b = part[(size_t)(s + 16) * c + col], d = part[(size_t)(s + 32) * c + col],
                  e = part[(size_t)(s + 48) * c + col];
      acc += a;
      acc += b;
      acc += d;
      acc += e;
    }
    for (; s < slabs; s += 16) acc += part[(size_t)s * c + col];
  }
  red[sub][cj] = acc;
  __syncthreads();
  if (sub == 0 && col < c) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cj];
    out[col] = accumulate ? out[col] + t : t;
  }
}

// ---- weight re-layout for the training step ------------------------------------------------------------------------------
// Every step the parameters change, and with them the GEMM-layout copies the conv kernels read: the forward [Cout][K]
// matrix, the tap-flipped transpose for the data gradient, the two phase matrices of a stride-2 data gradient
// (engine.pack_conv_weight*).  All of them are "dst[row][(32-channel block, tap, channel)] = src[row*sr + ch*sc + tap_map[tap]*st]"
// for some strides and tap map, so ONE launch walks a table of such copies (blockIdx.y = table entry).
__global__ __launch_bounds__(256) void pack_weights_kernel(const tdx_pack_desc* __restrict__ table) {
  const tdx_pack_desc d = table[blockIdx.y];
  const int kk = d.cin * d.taps;
  const long total = (long)d.rows * kk;
  const float* __restrict__ src = reinterpret_cast<const float*>(d.src);
  float* __restrict__ dst = reinterpret_cast<float*>(d.dst);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int row = (int)(i / kk), pos = (int)(i - (long)row * kk);
    const int blk = pos / (32 * d.taps), within = pos - blk * 32 * d.taps;
    const int cb = min(32, d.cin - blk * 32);   // (a last, narrower channel block keeps its taps adjacent too)
    const int t = within / cb, c = within - t * cb;
    const int tm = d.tap_map[t];
    dst[(size_t)row * d.dst_ld + pos] = tm >= 0 ? src[(size_t)row * d.sr + (size_t)(blk * 32 + c) * d.sc + (size_t)tm * d.st] : 0.f;
  }
}

// ---- AdamW, one fused pass (torch.optim.AdamW single-tensor semantics, train.ipynb:131) --------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float omb1, float beta2, float omb2, float eps,
                                                   float decay, float step_size, float bc2_sqrt) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  float pi = p[i] * decay;                            // param.mul_(1 - lr * weight_decay)
  const float mi = m[i] + (gi - m[i]) * omb1;  // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = v[i] * beta2 + omb2 * gi * gi;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const float denom = sqrtf(vi) / bc2_sqrt + eps;     // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
  pi -= step_size * (mi / denom);                      // param.addcdiv_(exp_avg, denom, value=-step_size)
  p[i] = pi;
  m[i] = mi;
  v[i] = vi;
}
// every tensor of a parameter group in one launch: blockIdx.y = table entry, grid-stride over its elements (same arithmetic)
__global__ __launch_bounds__(256) void adamw_multi_kernel(const tdx_adamw_desc* __restrict__ table, float omb1, float beta2, float omb2, float eps,
                                                         float decay, float step_size, float bc2_sqrt) {
  const tdx_adamw_desc d = table[blockIdx.y];
  float* __restrict__ p = d.param;
  const float* __restrict__ g = d.grad;
  float* __restrict__ m = d.exp_avg;
  float* __restrict__ v = d.exp_avg_sq;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (long)gridDim.x * 256) {
    const float gi = g[i];
    float pi = p[i] * decay;
    const float mi = m[i] + (gi - m[i]) * omb1;
    const float vi = v[i] * beta2 + omb2 * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}


// ---- convolution weight gradient -----------------------------------------------------------------------------------------
// dW[co][ci][t] = sum over output rows n of dY[n][co] * X[src(n, t)][ci], src = the same row map as the forward gather (zero
// padding, stride 2, nearest-upsampled source: tdx_kseg).  A GEMM whose reduction runs over the B * lout rows: the grid splits
// the rows into `splits` slabs (partials in the workspace), a second kernel adds the slabs in fixed order.
// Tile: 64 co x 64 ci of one tap; 32 rows per LDS stage, converted to fp32; v_mfma_f32_16x16x4f32 with A = dY^T, B = X.
struct WgradArgs {
  const char* dy;
  const char* x;
  int32_t dy_ld, dy_off, cout;
  int32_t x_ld, x_off, cin, taps, stride, lin, lvirt;
  float up_scale;
  int32_t N, lout, rows_per_split;
  float* part;  // [splits][cout][cin * taps]
  float* bias_part;  // [splits][cout] column sums of dY per slab (large-tile kernel only), or null
};

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int RC = 32, LDS_LD = 80;  // row stride 80 floats: the four k rows of an MFMA operand fall on disjoint banks
  __shared__ float dys[RC * LDS_LD], xs[RC * LDS_LD];
  const int co0 = blockIdx.x * 64;
  const int ci_tiles = (p.cin + 63) / 64;
  const int tap = blockIdx.y / ci_tiles, ci0 = (blockIdx.y % ci_tiles) * 64;
  const int split = blockIdx.z;
  const int r_begin = split * p.rows_per_split, r_end = min(p.N, r_begin + p.rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = tid >> 3, c8 = (tid & 7) * 8;  // loader role: row lr of the chunk, 8 consecutive channels
  const int pad = p.taps == 3 ? 1 : 0;
  const T* dy = reinterpret_cast<const T*>(p.dy);
  const T* x = reinterpret_cast<const T*>(p.x);
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int r0 = r_begin; r0 < r_end; r0 += RC) {
    const int n = r0 + lr;
    const bool row_ok = n < r_end;
    const int b = row_ok ? n / p.lout : 0, l = row_ok ? n - b * p.lout : 0;
    const int u = l * p.stride + tap - pad;
    const bool src_ok = row_ok && u >= 0 && u < p.lvirt;
    const int uu = src_ok ? u : 0;
    const int srow = p.up_scale > 0.f ? min((int)floorf((float)uu * p.up_scale), p.lin - 1) : uu;
    const T* dyr = dy + (size_t)n * p.dy_ld + p.dy_off + co0 + c8;
    const T* xr = x + ((size_t)b * p.lin + srow) * p.x_ld + p.x_off + ci0 + c8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dys[lr * LDS_LD + c8 + j] = (row_ok && co0 + c8 + j < p.cout) ? Elem<T>::ld(dyr + j) : 0.f;
      xs[lr * LDS_LD + c8 + j] = (src_ok && ci0 + c8 + j < p.cin) ? Elem<T>::ld(xr + j) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < RC / 4; ++k4) {
      const int kr = k4 * 4 + (lane >> 4);
      const float a = dys[kr * LDS_LD + wave * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xs[kr * LDS_LD + j * 16 + (lane & 15)], acc[j], 0, 0, 0);
    }
    __syncthreads();
  }
  // acc[j][r] = dW[co0 + 16 * wave + 4 * (lane >> 4) + r][ci0 + 16 * j + (lane & 15)]
  const int kdim = p.cin * p.taps;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ci = ci0 + j * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wave * 16 + 4 * (lane >> 4) + r;
      if (co < p.cout && ci < p.cin) p.part[((size_t)split * p.cout + co) * kdim + ci * p.taps + tap] = acc[j][r];
    }
  }
#endif
}

// The large-tile variant for the shapes that carry the FLOPs (fp32, stride 1, cout and cin multiples of 128): a 64 x 64 tile
// of one tap re-reads dY and X from L2 at 16 flop/B, more than the caches can feed the fp32 MFMA pipe with.  Here one
// 512-thread workgroup owns 128 co x 128 ci x ALL taps: the chunk's X rows are staged once as a (32 + 2)-row window that
// the three taps read at row offsets 0/1/2 (96 flop/B), with per-row flags masking the rows that belong to the neighbouring
// trajectory.  Each of the 8 waves holds 64 co x 32 ci x TAPS accumulators (96 VGPRs); the next chunk's global loads are
// in flight (registers) while the current chunk is multiplied.  Rows are VIRTUAL rows (b, u) of the nearest-upsampled
// source, mapped to stored rows by the loader, so upsampled sources take this path too; stride-2 convs use a window of
// even / odd source rows per output row (S2, below).
template <int TAPS, bool S2>
__global__ __launch_bounds__(512) void conv_wgrad_big_kernel(const WgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(!S2 || TAPS == 3, "the stride-2 form exists for 3-tap convs only");
  // Stride 2 (S2): output row n = (b, l) reads source rows 2l - 1, 2l, 2l + 1.  The window holds, per output row, the EVEN row
  // 2l (rows 0..31) and the ODD row 2l + 1 of rows n0 - 1 .. n0 + 31 (rows 32..64): tap 1 reads even[n], tap 2 odd[n], tap 0
  // odd[n - 1] (masked on a trajectory's first row) -- indexed by output row, so trajectory boundaries need no special case.
  constexpr int RC = 32, PAD = TAPS == 3 ? 1 : 0, WR = S2 ? 2 * RC + 1 : RC + 2 * PAD, LD = 144;  // 144 mod 64 = 16: the 4 k rows hit disjoint banks
  constexpr int XV = (WR * 32 + 511) / 512;                                     // float4 loads of the X window per thread
  __shared__ __attribute__((aligned(16))) float dys[RC * LD];
  __shared__ __attribute__((aligned(16))) float xs[WR * LD];
  __shared__ int flags[WR];                                                      // bit 0: first row of a trajectory, bit 1: last
  const int co0 = blockIdx.x * 128, ci0 = blockIdx.y * 128, split = blockIdx.z;
  const int r_begin = split * p.rows_per_split, r_end = min(p.N, r_begin + p.rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cob = (wave & 1) * 64, cib = (wave >> 1) * 32;
  const int l16 = lane & 15, lk = lane >> 4;
  const float* dy = reinterpret_cast<const float*>(p.dy) + p.dy_off + co0;
  const float* x = reinterpret_cast<const float*>(p.x) + p.x_off + ci0;
  f32x4 acc[TAPS][4][2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 dreg[2], xreg[XV];
  f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};   // bias gradient: this thread's share of the column sums of dY (rows tid / 32 and + 16 of every chunk)
  const bool want_bias = p.bias_part != nullptr && blockIdx.y == 0;
  int freg = 0;
  auto fetch = [&](int r0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {  // dY chunk: 32 rows x 32 float4
      const int e = tid + q * 512, row = e >> 5, c4 = (e & 31) * 4;
      const int n = r0 + row;
      dreg[q] = n < r_end ? *reinterpret_cast<const f32x4*>(dy + (size_t)n * p.dy_ld + c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < XV; ++q) {  // X window: WR virtual rows x 32 float4
      const int e = tid + q * 512, row = e >> 5, c4 = (e & 31) * 4;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (S2) {
        const int odd = row >= RC ? 1 : 0;
        const int n = odd ? r0 - 1 + (row - RC) : r0 + row;
        if (row < WR && n >= 0 && n < p.N) {
          const int b = n / p.lout, u = 2 * (n - b * p.lout) + odd;
          if (u < p.lin) v = *reinterpret_cast<const f32x4*>(x + ((size_t)b * p.lin + u) * p.x_ld + c4);
        }
      } else {
        const int m = r0 - PAD + row;
        if (row < WR && m >= 0 && m < p.N) {
          const int b = m / p.lvirt, u = m - b * p.lvirt;
          const int srow = p.up_scale > 0.f ? min((int)floorf((float)u * p.up_scale), p.lin - 1) : u;
          v = *reinterpret_cast<const f32x4*>(x + ((size_t)b * p.lin + srow) * p.x_ld + c4);
        }
      }
      xreg[q] = v;
    }
    if (tid < WR) {
      const int m = S2 ? r0 + tid : r0 - PAD + tid;   // S2: flags are per OUTPUT row (only the first RC entries are read)
      const int len = S2 ? p.lout : p.lvirt;
      const int u = m >= 0 ? m % len : 0;
      freg = (u == 0 ? 1 : 0) | (u == len - 1 ? 2 : 0);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + q * 512;
      *reinterpret_cast<f32x4*>(&dys[(e >> 5) * LD + (e & 31) * 4]) = dreg[q];
    }
    if (want_bias) bsum = (bsum + dreg[0]) + dreg[1];   // (rows past the slab were fetched as zeros)
#pragma unroll
    for (int q = 0; q < XV; ++q) {
      const int e = tid + q * 512;
      if ((e >> 5) < WR) *reinterpret_cast<f32x4*>(&xs[(e >> 5) * LD + (e & 31) * 4]) = xreg[q];
    }
    if (tid < WR) flags[tid] = freg;
  };

  if (r_begin < r_end) fetch(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += RC) {
    stash();
    __syncthreads();
    if (r0 + RC < r_end) fetch(r0 + RC);
#pragma unroll
    for (int k4 = 0; k4 < RC / 4; ++k4) {
      const int kr = k4 * 4 + lk;
      float a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = dys[kr * LD + cob + i * 16 + l16];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int wr = S2 ? (t == 1 ? kr : RC + kr + (t == 2 ? 1 : 0)) : kr + t;  // window row that output row kr reads through tap t
        float bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = xs[wr * LD + cib + j * 16 + l16];
        if (TAPS == 3 && t != 1) {
          // tap 0 never reads a trajectory's last row, tap 2 never its first; S2: tap 0 of a first row is the only dead read
          const bool dead = S2 ? (t == 0 && (flags[kr] & 1) != 0) : (flags[wr] & (t == 0 ? 2 : 1)) != 0;
          bv[0] = dead ? 0.f : bv[0];
          bv[1] = dead ? 0.f : bv[1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[t][i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // acc[t][i][j][r] = dW[co0 + cob + 16 i + 4 (lane >> 4) + r][ci0 + cib + 16 j + (lane & 15)][t]
  const int kdim = p.cin * TAPS;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + cob + i * 16 + 4 * lk + r, ci = ci0 + cib + j * 16 + l16;
          p.part[((size_t)split * p.cout + co) * kdim + ci * TAPS + t] = acc[t][i][j][r];
        }
  if (want_bias) {   // the 16 row classes of a column quad, added in order (the K loop's last barrier has released dys)
    *reinterpret_cast<f32x4*>(&dys[(tid >> 5) * LD + (tid & 31) * 4]) = bsum;
    __syncthreads();
    if (tid < 32) {
      f32x4 t = *reinterpret_cast<const f32x4*>(&dys[tid * 4]);
#pragma unroll
      for (int k = 1; k < 16; ++k) t = t + *reinterpret_cast<const f32x4*>(&dys[k * LD + tid * 4]);
      *reinterpret_cast<f32x4*>(p.bias_part + (size_t)split * p.cout + co0 + tid * 4) = t;
    }
  }
#endif
}

// dw[co][ci_off + ci][t] (row length cin_total * taps) = (accumulate ? dw : 0) + sum over splits, in split order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int cout, int cin, int taps,
                                                          float* __restrict__ dw, int cin_total, int ci_off, int accumulate) {
  const long kdim = (long)cin * taps, total = (long)cout * kdim;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long co = i / kdim, k = i - co * kdim;
  float acc = 0.f;
  for (int s = 0; s < splits; ++s) acc += part[(size_t)s * total + i];
  float* o = dw + ((size_t)co * cin_total + ci_off) * taps + k;
  *o = accumulate ? *o + acc : acc;
}

// The same sum for gradients whose rows are whole float4s, with the slabs spread over 16 classes per output: a level-0 gradient
// (128 x 384 values) is cut into 512 slabs, and one thread per output walking them one after the other is 192 workgroups each
// waiting for 512 dependent-latency loads (0.13 ms for 100 MB).  Block = 16 float4 columns x 16 classes; class q adds slabs
// q, q + 16, ... in order, then the classes are added in order: a fixed association for a given slab count, no atomics.
// Blocks past the gradient's own (blockIdx.x >= wblocks) do the same for the bias gradient's slab sums (bias_part [splits][cout], cout a
// multiple of 4), so that tdx_conv_wgrad_bias needs no third launch.
__global__ __launch_bounds__(256) void wgrad_reduce_v4_kernel(const float* __restrict__ part, int splits, int cout, int kdim4,
                                                             float* __restrict__ dw, int ld4, int off4, int accumulate, int wblocks,
                                                             const float* __restrict__ bias_part, float* __restrict__ dbias, int bias_accumulate) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ f32x4 red[16][16];
  if ((int)blockIdx.x >= wblocks) {   // (block-uniform)
    part = bias_part;
    dw = dbias;
    kdim4 = cout >> 2;
    cout = 1;
    ld4 = off4 = 0;
    accumulate = bias_accumulate;
  }
  const long total4 = (long)cout * kdim4;
  const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
  const long i = (long)((int)blockIdx.x >= wblocks ? (int)blockIdx.x - wblocks : (int)blockIdx.x) * 16 + c;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (i < total4) {
    const f32x4* src = reinterpret_cast<const f32x4*>(part) + i;
    int s = q;
    for (; s + 48 < splits; s += 64) {   // four independent loads in flight per thread
      const f32x4 v0 = src[(size_t)s * total4], v1 = src[(size_t)(s + 16) * total4], v2 = src[(size_t)(s + 32) * total4], v3 = src[(size_t)(s + 48) * total4];
      acc = acc + v0;
      acc = acc + v1;
      acc = acc + v2;
      acc = acc + v3;
    }
    for (; s < splits; s += 16) acc = acc + src[(size_t)s * total4];
  }
  red[q][c] = acc;
  __syncthreads();
  if (q == 0 && i < total4) {
    f32x4 t = red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) t = t + red[k][c];
    const long co = i / kdim4, k4 = i - co * kdim4;
    f32x4* o = reinterpret_cast<f32x4*>(dw) + co * ld4 + off4 + k4;
    *o = accumulate ? *o + t : t;
  }
#endif
}

// ---- GroupNorm (+ Swish) backward ------------------------------------------------------------------------------------------
// forward: xh = (x - mean_g) * rstd_g ; u = xh * gamma + beta ; y = act(u).
// backward: du = dy * act'(u) ; dgamma_c = sum du * xh ; dbeta_c = sum du ; dxh = du * gamma ;
//           dx = rstd_g * (dxh - mean_g(dxh) - xh * mean_g(dxh * xh)).
// One wave per (trajectory, channel window), same decomposition as the forward kernel; four sweeps over the rows (statistics
// are recomputed, nothing but x was saved by the forward pass).  Per-channel sums are written per trajectory
// (dgamma_rows / dbeta_rows, (B, C) fp32); tdx_colsum over those rows gives dgamma / dbeta in a fixed order.
// NIT > 0: the wave's rows of x and dy (NIT row iterations per lane) are loaded ONCE, all at the start, and the four sweeps run on
// registers; a sweep that re-reads its rows pays one dependent memory round trip per row iteration, 100 of them at L = 200, and
// that latency, not the 315 MB of traffic, was the kernel's time (0.13 ms per full-resolution layer).  NIT == 0: the re-reading
// form, for trajectories too long for the register file.  Same arithmetic and summation order in both.
// DROP: dy is the gradient of dropout(act(...)): the mask of the forward pass (tdx_groupnorm_act_dropout, i.e. tdx_dropout's for the
// (B, L, C) tensor dy belongs to) is regenerated and applied as dy is read.
struct GnDropB {
  uint32_t thresh, layer;
  float scale;
  uint64_t seed;
  const int32_t* step_ctr;
};
template <typename T, int NIT, int S, bool DROP>
__global__ __launch_bounds__(256) void groupnorm_act_bwd_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ dy, int dy_ld,
                                                               int dy_off, T* __restrict__ dx, int dx_ld, int dx_off,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ dgamma_rows, float* __restrict__ dbeta_rows, int B,
                                                               int L, int C, int cpg, int lpr, float eps, int act, int accumulate,
                                                               int rows_per_part, GnDropB drop) {
  constexpr int VEC = Elem<T>::kVec;
  static_assert(!DROP || VEC == 4, "fused dropout: fp32 rows (one Philox call per 4 channels)");
  constexpr int H = VEC / 2;
  constexpr int NR = NIT > 0 ? NIT : 1;
  constexpr bool KEEP = NIT > 0 && sizeof(T) == 4;   // fp32 rows in registers: exact to overwrite dy with dxh
  static_assert(S == 1 || NIT > 0, "split sweeps keep their rows in registers");
  __shared__ float red[S > 1 ? 3 : 1][4][64][4];       // [exchange][wave][column][sums]
  __shared__ float redp[S > 1 ? 4 : 1][64][2 * VEC];   // per-channel dgamma | dbeta partials of the S waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = lpr * VEC;
  const int wins = C / W;
  const long wid = ((long)blockIdx.x * 4 + wave) / S;   // (trajectory, window) item; S consecutive waves share one
  const int part = wave % S;
  const bool active = wid < (long)B * wins;             // wave-uniform, the same for the S waves of an item
  if (S == 1 && !active) return;                        // (split sweeps: idle waves still take part in the barriers)
  const int b = active ? (int)(wid / wins) : 0, win = active ? (int)(wid % wins) : 0;
  const int col = lane % lpr, rl = lane / lpr, rpi = 64 / lpr;
  const int ch = win * W + col * VEC;
  const T* xb = x + ((size_t)b * L) * x_ld + ch;
  const T* dyb = dy + ((size_t)b * L) * dy_ld + dy_off + ch;
  T* dxb = dx + ((size_t)b * L) * dx_ld + dx_off + ch;
  const bool split = cpg < VEC;
  // this wave's rows: [l0, L1)
  const int l0 = S > 1 ? part * rows_per_part : 0;
  const int L1 = !active ? 0 : (S > 1 ? (l0 + rows_per_part < L ? l0 + rows_per_part : L) : L);
  int pass = 0;
  // sums over the wave's rows -> over the item's rows (the other waves' partials through LDS, added in the order of the parts) -> over the group
  auto group_sum2 = [&](float& lo, float& hi, float& lo2, float& hi2) {
    for (int o = 32; o >= lpr; o >>= 1) {
      lo += __shfl_xor(lo, o, 64);
      hi += __shfl_xor(hi, o, 64);
      lo2 += __shfl_xor(lo2, o, 64);
      hi2 += __shfl_xor(hi2, o, 64);
    }
    if constexpr (S > 1) {
      if (rl == 0) *reinterpret_cast<f32x4*>(&red[pass][wave][col][0]) = (f32x4){lo, hi, lo2, hi2};
      __syncthreads();
      const int w0 = wave - part;
      lo = hi = lo2 = hi2 = 0.f;
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&red[pass][w0 + k][col][0]);
        lo += v[0];
        hi += v[1];
        lo2 += v[2];
        hi2 += v[3];
      }
      ++pass;
    }
    if (!split) {
      lo += hi;
      lo2 += hi2;
      for (int o = 1; o * VEC < cpg; o <<= 1) {
        lo += __shfl_xor(lo, o, 64);
        lo2 += __shfl_xor(lo2, o, 64);
      }
      hi = lo;
      hi2 = lo2;
    }
  };
  auto group_sum = [&](float& lo, float& hi) {
    float z0 = 0.f, z1 = 0.f;
    group_sum2(lo, hi, z0, z1);
  };
  u32x4 hx[NR], hg[NR];
  if constexpr (NIT > 0) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int l = l0 + rl + i * rpi;
      hx[i] = l < L1 ? *reinterpret_cast<const u32x4*>(xb + (size_t)l * x_ld) : (u32x4){0u, 0u, 0u, 0u};
      hg[i] = l < L1 ? *reinterpret_cast<const u32x4*>(dyb + (size_t)l * dy_ld) : (u32x4){0u, 0u, 0u, 0u};
    }
  }
  auto mask = [&](int l, float (&g)[VEC]) {
    if constexpr (DROP) {
      uint32_t r[4];
      philox4((uint64_t)(((long)b * L + l) * C + ch) >> 2, drop.layer, drop.seed ^ ((uint64_t)(uint32_t)drop.step_ctr[0] << 32), r);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = r[j] >= drop.thresh ? g[j] * drop.scale : 0.f;
    }
  };
  if constexpr (DROP && NIT > 0) {   // once, in the registers
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int l = l0 + rl + i * rpi;
      if (l < L1) {
        float g[VEC];
        Elem<T>::unpack(hg[i], g);
        mask(l, g);
        hg[i] = Elem<T>::pack(g);
      }
    }
  }
  auto xrow = [&](int i, int l, float (&v)[VEC]) {
    if constexpr (NIT > 0) Elem<T>::unpack(hx[i], v);
    else Elem<T>::unpack(*reinterpret_cast<const u32x4*>(xb + (size_t)l * x_ld), v);
  };
  auto grow = [&](int i, int l, float (&v)[VEC]) {
    if constexpr (NIT > 0) Elem<T>::unpack(hg[i], v);
    else {
      Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dyb + (size_t)l * dy_ld), v);
      mask(l, v);
    }
  };
  const int nit = NIT > 0 ? NIT : (L - rl + rpi - 1) / rpi;   // (NIT == 0 only with S == 1: l0 = 0, L1 = L)
  const float inv_cnt = 1.0f / ((float)cpg * (float)L);
  float s_lo = 0.f, s_hi = 0.f;
#pragma unroll
  for (int i = 0; i < nit; ++i) {
    const int l = l0 + rl + i * rpi;
    if (l >= L1) break;
    float v[VEC];
    xrow(i, l, v);
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
    if (l >= L1) break;
    float v[VEC];
    xrow(i, l, v);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const float a = v[j] - m_lo, c = v[H + j] - m_hi;
      q_lo += a * a;
      q_hi += c * c;
    }
  }
  group_sum(q_lo, q_hi);
  const float r_lo = rsqrtf(q_lo * inv_cnt + eps), r_hi = rsqrtf(q_hi * inv_cnt + eps);
  float gm[VEC], bt[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    gm[j] = gamma[ch + j];
    bt[j] = beta[ch + j];
  }
  // du and the four sums
  auto du_of = [&](float xh, float dyv, int j) {
    const float u = xh * gm[j] + bt[j];
    if (act == TDX_ACT_SWISH) {
      const float sg = 1.0f / (1.0f + __expf(-u));
      return dyv * (sg * (1.0f + u * (1.0f - sg)));
    }
    if (act == TDX_ACT_LEAKY) return u > 0.f ? dyv : 0.1f * dyv;
    return dyv;
  };
  float a_lo = 0.f, a_hi = 0.f, c_lo = 0.f, c_hi = 0.f;  // group sums of dxh and dxh * xh
  float dg[VEC], db[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) dg[j] = db[j] = 0.f;
#pragma unroll
  for (int i = 0; i < nit; ++i) {
    const int l = l0 + rl + i * rpi;
    if (l >= L1) break;
    float v[VEC], g[VEC];
    xrow(i, l, v);
    grow(i, l, g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float xh = (v[j] - (j < H ? m_lo : m_hi)) * (j < H ? r_lo : r_hi);
      const float du = du_of(xh, g[j], j);
      dg[j] += du * xh;
      db[j] += du;
      const float dxh = du * gm[j];
      if (j < H) {
        a_lo += dxh;
        c_lo += dxh * xh;
      } else {
        a_hi += dxh;
        c_hi += dxh * xh;
      }
      g[j] = dxh;
    }
    if constexpr (KEEP) hg[i] = Elem<T>::pack(g);   // dy's registers now hold dxh: the last sweep does not evaluate the activation again
  }
  group_sum2(a_lo, a_hi, c_lo, c_hi);
  const float ma_lo = a_lo * inv_cnt, ma_hi = a_hi * inv_cnt, mc_lo = c_lo * inv_cnt, mc_hi = c_hi * inv_cnt;
#pragma unroll
  for (int j = 0; j < VEC; ++j)
    for (int o = 32; o >= lpr; o >>= 1) {
      dg[j] += __shfl_xor(dg[j], o, 64);
      db[j] += __shfl_xor(db[j], o, 64);
    }
  if constexpr (S > 1) {   // the S waves' per-channel sums, added in the order of the parts by the first of them
    if (rl == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        redp[wave][col][j] = dg[j];
        redp[wave][col][VEC + j] = db[j];
      }
    }
    __syncthreads();
    if (rl == 0 && part == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        dg[j] = db[j] = 0.f;
#pragma unroll
        for (int k = 0; k < S; ++k) {
          dg[j] += redp[wave + k][col][j];
          db[j] += redp[wave + k][col][VEC + j];
        }
      }
    }
  }
  if (rl == 0 && part == 0 && active) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      dgamma_rows[(size_t)b * C + ch + j] = dg[j];
      dbeta_rows[(size_t)b * C + ch + j] = db[j];
    }
  }
  auto dx_of = [&](int i, int l, float (&o)[VEC]) {
    float v[VEC], g[VEC];
    xrow(i, l, v);
    grow(i, l, g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float r = j < H ? r_lo : r_hi;
      const float xh = (v[j] - (j < H ? m_lo : m_hi)) * r;
      const float dxh = KEEP ? g[j] : du_of(xh, g[j], j) * gm[j];
      o[j] = r * (dxh - (j < H ? ma_lo : ma_hi) - xh * (j < H ? mc_lo : mc_hi));
    }
  };
  if constexpr (NIT > 0 && sizeof(T) == 4) {
    if (accumulate) {
      // the gradients take dy's registers, then the rows of dx already there are fetched together into x's (no load waits for a store)
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        float o[VEC];
        dx_of(i, l0 + rl + i * rpi, o);
        hg[i] = Elem<T>::pack(o);
      }
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int l = l0 + rl + i * rpi;
        if (l < L1) hx[i] = *reinterpret_cast<const u32x4*>(dxb + (size_t)l * dx_ld);
      }
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int l = l0 + rl + i * rpi;
        if (l >= L1) break;
        float o[VEC], d[VEC];
        Elem<T>::unpack(hx[i], o);
        Elem<T>::unpack(hg[i], d);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = o[j] + d[j];
        *reinterpret_cast<u32x4*>(dxb + (size_t)l * dx_ld) = Elem<T>::pack(o);
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < nit; ++i) {
    const int l = l0 + rl + i * rpi;
    if (l >= L1) break;
    float o[VEC], d[VEC];
    if (accumulate) Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dxb + (size_t)l * dx_ld), o);
    dx_of(i, l, d);
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = accumulate ? o[j] + d[j] : d[j];
    *reinterpret_cast<u32x4*>(dxb + (size_t)l * dx_ld) = Elem<T>::pack(o);
  }
}

template <typename T, bool DROP>
void launch_gn_bwd(int nit, int parts, int rows_per_part, dim3 grid, hipStream_t st, const void* x, int x_ld, const void* dy, int dy_ld, int dy_off,
                   void* dx, int dx_ld, int dx_off, const float* gamma, const float* beta, float* dgamma_rows, float* dbeta_rows, int B, int L, int C,
                   int cpg, int lpr, float eps, int act, int accumulate, GnDropB drop) {
#define TDX_GNB(N, S)                                                                                                                                  \
  hipLaunchKernelGGL((groupnorm_act_bwd_kernel<T, N, S, DROP>), grid, dim3(256), 0, st, (const T*)x, x_ld, (const T*)dy, dy_ld, dy_off, (T*)dx, dx_ld, \
                     dx_off, gamma, beta, dgamma_rows, dbeta_rows, B, L, C, cpg, lpr, eps, act, accumulate, rows_per_part, drop)
  if (parts == 4) TDX_GNB(8, 4);
  else if (parts == 2) TDX_GNB(8, 2);
  else if (nit <= 2) TDX_GNB(2, 1);
  else if (nit <= 4) TDX_GNB(4, 1);
  else if (nit <= 8) TDX_GNB(8, 1);
  else TDX_GNB(0, 1);
#undef TDX_GNB
}

// ---- row re-indexing of data gradients -----------------------------------------------------------------------------------
// interleave2: out[b][u] (+)= (u even ? even[b][u / 2] : odd[b][u / 2]): the two phases of a stride-2 conv's data gradient
// (even input rows see only the centre tap, odd rows the two outer taps), each computed by tdx_conv_gemm on dy.
template <typename T>
__global__ __launch_bounds__(256) void interleave2_kernel(const T* __restrict__ even, int le, const T* __restrict__ odd,
                                                         T* __restrict__ out, int out_ld, int out_off, int B, int L, int C, int accumulate) {
  constexpr int VEC = Elem<T>::kVec;
  const int vpr = C / VEC;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * L * vpr) return;
  const int cv = (int)(i % vpr);
  const long row = i / vpr;
  const int b = (int)(row / L), u = (int)(row - (long)b * L);
  const T* src = ((u & 1) ? odd : even) + ((size_t)b * le + (u >> 1)) * C;  // both phase tensors are (B, ceil(L / 2), C)
  T* dst = out + (size_t)row * out_ld + out_off + cv * VEC;
  float v[VEC];
  Elem<T>::unpack(*reinterpret_cast<const u32x4*>(src + cv * VEC), v);
  if (accumulate) {
    float o[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dst), o);
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] += o[j];
  }
  *reinterpret_cast<u32x4*>(dst) = Elem<T>::pack(v);
}

// upsample_fold: dx[b][s] (+)= sum of du[b][u] over the virtual rows u that F.interpolate(mode='nearest') takes from source
// row s, i.e. min(floor(u * scale), lin - 1) == s (ATen's index formula, the one the forward gather uses).
template <typename T>
__global__ __launch_bounds__(256) void upsample_fold_kernel(const T* __restrict__ du, int du_ld, int du_off, T* __restrict__ dx, int dx_ld,
                                                           int dx_off, int B, int lin, int lvirt, int C, float scale, int accumulate) {
  constexpr int VEC = Elem<T>::kVec;
  const int vpr = C / VEC;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * lin * vpr) return;
  const int cv = (int)(i % vpr);
  const long row = i / vpr;
  const int b = (int)(row / lin), sr = (int)(row - (long)b * lin);
  T* dst = dx + (size_t)row * dx_ld + dx_off + cv * VEC;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (accumulate) Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dst), acc);
  const int reach = (int)ceilf(1.0f / scale) + 2;  // the preimage of a source row is a run of at most ceil(1 / scale) virtual rows
  int u0 = (int)floorf((float)sr / scale) - 1;
  if (u0 < 0) u0 = 0;
  const int u1 = sr == lin - 1 ? lvirt : min(lvirt, u0 + reach + 1);  // (the clamp sends every late row to the last source row)
  for (int u = u0; u < u1; ++u) {
    if (min((int)floorf((float)u * scale), lin - 1) != sr) continue;
    float v[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(du + ((size_t)b * lvirt + u) * du_ld + du_off + cv * VEC), v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] += v[j];
  }
  *reinterpret_cast<u32x4*>(dst) = Elem<T>::pack(acc);
}

// ---- small pieces ---------------------------------------------------------------------------------------------------------
// act_fwd: y = act(z);  act_bwd: dz (+)= dy * act'(.), from the pre-activation z (swish) or from the OUTPUT y (leaky keeps the
// sign, so a conv that applied it in its epilogue needs nothing saved).  fp32, strided rows.
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ z, int z_ld, float* __restrict__ y, int y_ld, int n, int c, int act) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)n * c) return;
  const int r = (int)(i / c), k = (int)(i - (long)r * c);
  y[(size_t)r * y_ld + k] = tdx_act(z[(size_t)r * z_ld + k], act);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int dy_ld, const float* __restrict__ ref, int ref_ld,
                                                     float* __restrict__ dz, int dz_ld, int n, int c, int act, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)n * c) return;
  const int r = (int)(i / c), k = (int)(i - (long)r * c);
  const float g = dy[(size_t)r * dy_ld + k], v = ref[(size_t)r * ref_ld + k];
  float d = g;
  if (act == TDX_ACT_SWISH) {
    const float sg = 1.0f / (1.0f + __expf(-v));
    d = g * (sg * (1.0f + v * (1.0f - sg)));
  } else if (act == TDX_ACT_LEAKY) {
    d = v > 0.f ? g : 0.1f * g;
  }
  float* o = dz + (size_t)r * dz_ld + k;
  *o = accumulate ? *o + d : d;
}

// out[b][c] (+)= sum over the L rows of trajectory b of x[(b * L + l) * ld + c]: gradient of the per-trajectory embed bias
// that a conv epilogue broadcast over its rows (TrajUNetBlocks.py:96).  One thread per (b, c), fixed order.
__global__ __launch_bounds__(256) void traj_rowsum_kernel(const float* __restrict__ x, int ld, float* __restrict__ out, int out_ld, int B, int L,
                                                         int c, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * c) return;
  const int b = (int)(i / c), k = (int)(i - (long)b * c);
  float acc = 0.f;
  for (int l = 0; l < L; ++l) acc += x[((size_t)b * L + l) * ld + k];
  float* o = out + (size_t)b * out_ld + k;
  *o = accumulate ? *o + acc : acc;
}
// the same sums, four columns per thread and eight row loads in flight; rows are added in the same order (bit-identical)
__global__ __launch_bounds__(256) void traj_rowsum_v4_kernel(const float* __restrict__ x, int ld, float* __restrict__ out, int out_ld, int B, int L,
                                                            int c4, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * c4) return;
  const int b = (int)(i / c4), k = (int)(i - (long)b * c4) * 4;
  const float* src = x + (size_t)b * L * ld + k;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  int l = 0;
  for (; l + 8 <= L; l += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(l + u) * ld);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = acc + v[u];
  }
  for (; l < L; ++l) acc = acc + *reinterpret_cast<const f32x4*>(src + (size_t)l * ld);
  f32x4* o = reinterpret_cast<f32x4*>(out + (size_t)b * out_ld + k);
  *o = accumulate ? *o + acc : acc;
}

// nn.Embedding backward: dtable[r][c] (+)= sum over samples b and index columns j with idx[b * idx_ld + j] == r of
// dvec[b * dvec_ld + j * dim + c], in (b, j) order: fixed association, no atomics.  One workgroup per table row r.  Finding the matches
// is the parallel part: 4096 (b, j) pairs at a time, one pair per thread and load (16 loads in flight), a ballot per wave, the 64 masks
// through LDS; then every thread walks the set bits in pair order and adds its columns of the matching rows.  (One thread per table
// element scanning the batch by itself: 4096 dependent-latency iterations, 0.44 ms for the 256 x 128 cell table at B = 1024.)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ idx, int idx_ld, int ncols, const float* __restrict__ dvec,
                                                           int dvec_ld, float* __restrict__ dtable, int rows, int dim, int B, int accumulate) {
  __shared__ unsigned long long masks[16][4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long npairs = (long)B * ncols;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};   // columns tid + 256 k (dim <= 1024)
  for (long p0 = 0; p0 < npairs; p0 += 4096) {
    bool hit[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const long pr = p0 + k * 256 + tid;
      hit[k] = false;
      if (pr < npairs) {
        const long b = pr / ncols;
        hit[k] = idx[(size_t)b * idx_ld + (int)(pr - b * ncols)] == r;
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const unsigned long long m = __ballot(hit[k]);
      if (lane == 0) masks[k][wave] = m;
    }
    __syncthreads();
    for (int m = 0; m < 64; ++m) {
      const unsigned long long mv = masks[m >> 2][m & 3];
      unsigned long long mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mv >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mv);
      while (mask) {
        const int bit = __builtin_ctzll(mask);
        mask &= mask - 1;
        const long pr = p0 + (long)m * 64 + bit;
        const long b = pr / ncols;
        const float* v = dvec + (size_t)b * dvec_ld + (size_t)(pr - b * ncols) * dim;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = tid + 256 * k;
          if (c < dim) acc[k] += v[c];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = tid + 256 * k;
    if (c < dim) {
      float* o = dtable + (size_t)r * dim + c;
      *o = accumulate ? *o + acc[k] : acc[k];
    }
  }
}

// Mid-block attention backward (TrajUNetBlocks.py:117-125), fp32, L <= 16 tokens: one workgroup per trajectory.
//   S = Q K^T * scale, P = softmax(S), O = P V;  dV = P^T dO, dP = dO V^T, dS = P * (dP - rowsum(dP * P)),
//   dQ = dS K * scale, dK = dS^T Q * scale.   qkv / dqkv rows are (q | k | v) of 3 * C channels.
// Phase 1 (S and dP, two C-long dot products per token pair): a wave owns two query rows at a time and its lanes split the channels, so
// every load instruction reads 1 KB of one contiguous row and a K / V row is fetched once per row pair (one thread per (i, j) pair read
// 16 different rows per instruction and every row 16 times: 4 MB through the L1 per trajectory, 0.52 ms per step).  Phase 2: a thread
// owns a channel column, loads its dO / K / Q values of all tokens once and forms the column's dQ, dK, dV from P and dS in LDS.
template <bool VEC>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, float* __restrict__ dqkv,
                                                           int L, int C, float scale) {
  __shared__ float Sraw[16 * 16], dPraw[16 * 16], P[16 * 16], dS[16 * 16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* base = qkv + (size_t)b * L * 3 * C;
  const float* dob = dout + (size_t)b * L * C;
  float* dbase = dqkv + (size_t)b * L * 3 * C;
  constexpr int W = VEC ? 4 : 1;
  for (int i0 = 2 * wave; i0 < L; i0 += 8) {
    const int i1 = i0 + 1 < L ? i0 + 1 : i0;   // (an odd last row is computed twice and written once)
    float s[2][16], dp[2][16];
#pragma unroll
    for (int j = 0; j < 16; ++j) s[0][j] = s[1][j] = dp[0][j] = dp[1][j] = 0.f;
    for (int c = lane * W; c < C; c += 64 * W) {
      float q0[W], q1[W], g0[W], g1[W];
      if constexpr (VEC) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + (size_t)i0 * 3 * C + c), a1 = *reinterpret_cast<const f32x4*>(base + (size_t)i1 * 3 * C + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(dob + (size_t)i0 * C + c), g1v = *reinterpret_cast<const f32x4*>(dob + (size_t)i1 * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          q0[e] = a[e];
          q1[e] = a1[e];
          g0[e] = g[e];
          g1[e] = g1v[e];
        }
      } else {
        q0[0] = base[(size_t)i0 * 3 * C + c];
        q1[0] = base[(size_t)i1 * 3 * C + c];
        g0[0] = dob[(size_t)i0 * C + c];
        g1[0] = dob[(size_t)i1 * C + c];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < L) {
          float kk[W], vv[W];
          if constexpr (VEC) {
            const f32x4 k4 = *reinterpret_cast<const f32x4*>(base + (size_t)j * 3 * C + C + c), v4 = *reinterpret_cast<const f32x4*>(base + (size_t)j * 3 * C + 2 * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              kk[e] = k4[e];
              vv[e] = v4[e];
            }
          } else {
            kk[0] = base[(size_t)j * 3 * C + C + c];
            vv[0] = base[(size_t)j * 3 * C + 2 * C + c];
          }
#pragma unroll
          for (int e = 0; e < W; ++e) {
            s[0][j] = fmaf(q0[e], kk[e], s[0][j]);
            s[1][j] = fmaf(q1[e], kk[e], s[1][j]);
            dp[0][j] = fmaf(g0[e], vv[e], dp[0][j]);
            dp[1][j] = fmaf(g1[e], vv[e], dp[1][j]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < L) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float a = s[r][j], d = dp[r][j];
          for (int o = 32; o >= 1; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            d += __shfl_xor(d, o, 64);
          }
          if (lane == 0 && (r == 0 || i1 != i0)) {
            Sraw[(r ? i1 : i0) * 16 + j] = a * scale;
            dPraw[(r ? i1 : i0) * 16 + j] = d;
          }
        }
      }
    }
  }
  __syncthreads();
  const int i = tid >> 4, j = tid & 15;  // thread (i, j) owns score S[i][j] and dP[i][j]
  const bool in = i < L && j < L;
  const float sc = in ? Sraw[tid] : 0.f, dpv = in ? dPraw[tid] : 0.f;
  // softmax over j (16 consecutive lanes), masked to j < L
  float m = in ? sc : -INFINITY;
  for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const float e = in ? expf(sc - m) : 0.f;
  float sum = e;
  for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
  const float p = in ? e / sum : 0.f;
  float t = dpv * p;
  for (int o = 1; o < 16; o <<= 1) t += __shfl_xor(t, o, 64);
  P[tid] = p;
  dS[tid] = p * (dpv - t) * scale;   // already multiplied by the score scale
  __syncthreads();
  // dV[j][c] = sum_i P[i][j] dO[i][c];  dQ[i][c] = sum_j dS[i][j] K[j][c];  dK[j][c] = sum_i dS[i][j] Q[i][c]
  for (int c = tid; c < C; c += 256) {
    float go[16], kc[16], qc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const bool ok = o < L;
      go[o] = ok ? dob[(size_t)o * C + c] : 0.f;
      kc[o] = ok ? base[(size_t)o * 3 * C + C + c] : 0.f;
      qc[o] = ok ? base[(size_t)o * 3 * C + c] : 0.f;
    }
    for (int r = 0; r < L; ++r) {
      float dv = 0.f, dq = 0.f, dk = 0.f;
#pragma unroll
      for (int o = 0; o < 16; ++o) {   // (rows o >= L of P / dS are zero, and so are the padded column values)
        dv = fmaf(P[o * 16 + r], go[o], dv);
        dq = fmaf(dS[r * 16 + o], kc[o], dq);
        dk = fmaf(dS[o * 16 + r], qc[o], dk);
      }
      dbase[(size_t)r * 3 * C + c] = dq;
      dbase[(size_t)r * 3 * C + C + c] = dk;
      dbase[(size_t)r * 3 * C + 2 * C + c] = dv;
    }
  }
}

// nn.Dropout(p) in training mode (TrajUNetBlocks.py:85): x *= keep / (1 - p), keep ~ Bernoulli(1 - p) from Philox4x32-10 keyed by
// (seed, step, layer) and the element's logical index, so the backward pass regenerates the same mask from the same call.
__global__ __launch_bounds__(256) void dropout_kernel(float* __restrict__ x, int ld, int n, int c, uint32_t thresh, float scale, uint64_t seed,
                                                     uint32_t layer, const int32_t* __restrict__ step_ctr) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;   // one Philox call = 4 consecutive elements
  const long total = (long)n * c;
  if (q * 4 >= total) return;
  uint32_t r[4];
  philox4((uint64_t)q, layer, seed ^ ((uint64_t)(uint32_t)step_ctr[0] << 32), r);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long e = q * 4 + j;
    if (e >= total) break;
    const int row = (int)(e / c), k = (int)(e - (long)row * c);
    float* px = x + (size_t)row * ld + k;
    *px = r[j] >= thresh ? *px * scale : 0.f;
  }
}

}  // namespace

extern "C" int64_t tdx_train_workspace_floats(void) { return (int64_t)kRedBlocks * 4096; }

extern "C" int tdx_mse_loss_grad(const float* pred, const float* target, float* dpred, float* loss, int64_t n, float* ws, void* stream) {
  TDX_CHECK_ARG(pred && target && loss && ws && n > 0, "tdx_mse_loss_grad: null pointer or empty input");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((n + 255) / 256 < kRedBlocks ? (n + 255) / 256 : kRedBlocks);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, st, pred, target, dpred, ws, (long)n, 2.0f / (float)n);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, ws, blocks, loss, 1.0f / (float)n);
  TDX_CHECK_LAUNCH("mse_loss_grad");
  return TDX_OK;
}

// ---- the one entry point that allocates nothing (no reference counterpart; for callers that cannot allocate) --------------
// tdx_mse_loss_grad with its first-stage partial sums in a file-scope device array instead of a caller workspace.
// `g_mse_scratch` is, in host code, only a stand-in for the symbol: the device address is asked for once per device with
// hipGetSymbolAddress (each GPU holds its own copy of the array) and that pointer is what every kernel argument uses.
constexpr int64_t kStaticMaxN = (int64_t)1 << 28;    // elements per operand (1 GiB of fp32); 4 KiB of scratch per GPU
constexpr int kMaxDevices = 64;
static __device__ __align__(256) float g_mse_scratch[kRedBlocks];
static float* g_mse_scratch_ptr[kMaxDevices] = {};   // per device, filled on the first call made with that device current

static int static_scratch(float** out, hipStream_t st) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) TDX_FAIL(TDX_ERR_HIP, "tdx_mse_loss_grad_static: no current device (or index >= %d)", kMaxDevices);
  // the array's address belongs to ONE device: a stream of another GPU would launch there with this GPU's address
  if (st != nullptr) {
    int sdev = -1;
    if (hipStreamGetDevice(st, &sdev) == hipSuccess && sdev != dev)
      TDX_FAIL(TDX_ERR_ARG, "tdx_mse_loss_grad_static: the stream belongs to device %d but device %d is current; make the stream's device current", sdev, dev);
  }
  float* p = __atomic_load_n(&g_mse_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (p == nullptr) {
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g_mse_scratch));
    if (e != hipSuccess || sym == nullptr) TDX_FAIL(TDX_ERR_HIP, "tdx_mse_loss_grad_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
    p = (float*)sym;
    __atomic_store_n(&g_mse_scratch_ptr[dev], p, __ATOMIC_RELEASE);   // (two threads racing here store the same address)
  }
  *out = p;
  return TDX_OK;
}

extern "C" int tdx_mse_loss_grad_static(const float* pred, const float* target, float* dpred, float* loss, int64_t n, void* stream) {
  TDX_CHECK_ARG(pred && target && loss && n > 0, "tdx_mse_loss_grad_static: null pointer or empty input");
  TDX_CHECK_ARG(n <= kStaticMaxN, "tdx_mse_loss_grad_static: n=%lld exceeds the static scratch's limit of %lld elements; use tdx_mse_loss_grad with a workspace",
                (long long)n, (long long)kStaticMaxN);
  float* ws = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int rc = static_scratch(&ws, st);
  if (rc != TDX_OK) return rc;
  const int blocks = (int)((n + 255) / 256 < kRedBlocks ? (n + 255) / 256 : kRedBlocks);
  // every partial the second kernel reads is written by the first one of THIS call: nothing has to start at zero
  hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, st, pred, target, dpred, ws, (long)n, 2.0f / (float)n);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, ws, blocks, loss, 1.0f / (float)n);
  TDX_CHECK_LAUNCH("mse_loss_grad_static");
  return TDX_OK;
}

extern "C" int tdx_colsum(const void* x, int32_t ld, int32_t n, int32_t c, float* out, int32_t accumulate, int32_t dtype, float* ws, void* stream) {
  TDX_CHECK_ARG(x && out && ws && n > 0 && c > 0 && ld >= c, "tdx_colsum: bad arguments n=%d c=%d ld=%d", n, c, ld);
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_colsum: bad dtype %d", dtype);
  TDX_CHECK_ARG(c <= 4096, "tdx_colsum: c=%d exceeds the workspace (4096 columns)", c);
  hipStream_t st = (hipStream_t)stream;
  int slabs;
  if (dtype == TDX_F32 && c % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0 && (c <= 256 || c % 256 == 0)) {
    const int lpr = c >= 256 ? 64 : c / 4;                   // lanes per row; c < 256 must be a power of two times 4
    if ((lpr & (lpr - 1)) == 0) {
      const int windows = (c + 255) / 256, rows_iter = 4 * (64 / lpr);
      slabs = std::max(1, std::min(kRedBlocks, std::min(512 / windows, (n + 4 * rows_iter - 1) / (4 * rows_iter))));
      const int rows_per_slab = ((n + slabs - 1) / slabs + rows_iter - 1) / rows_iter * rows_iter;
      slabs = (n + rows_per_slab - 1) / rows_per_slab;
      hipLaunchKernelGGL(colsum_partial_v4_kernel<4>, dim3((unsigned)slabs, (unsigned)windows), dim3(256), 0, st, (const float*)x, ld, n, c, lpr,
                         ws, rows_per_slab, (const float*)nullptr);
      hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((c + 15) / 16)), dim3(256), 0, st, ws, slabs, c, out, accumulate, (float*)nullptr);
      TDX_CHECK_LAUNCH("colsum");
      return TDX_OK;
    }
  }
  slabs = (n + 255) / 256;
  if (slabs > kRedBlocks) slabs = kRedBlocks;
  const dim3 grid((unsigned)slabs, (unsigned)((c + 63) / 64));
  if (dtype == TDX_F32) hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ld, n, c, ws, slabs);
  else hipLaunchKernelGGL(colsum_partial_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)x, ld, n, c, ws, slabs);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((c + 15) / 16)), dim3(256), 0, st, ws, slabs, c, out, accumulate);
  TDX_CHECK_LAUNCH("colsum");
  return TDX_OK;
}

extern "C" int tdx_colsum_pair(const void* xa, const void* xb, int32_t ld, int32_t n, int32_t c, float* outa, float* outb, int32_t accumulate,
                               int32_t dtype, float* ws, void* stream) {
  TDX_CHECK_ARG(xa && xb && outa && outb && ws && n > 0 && c > 0 && ld >= c, "tdx_colsum_pair: bad arguments n=%d c=%d ld=%d", n, c, ld);
  const int lpr = c >= 256 ? 64 : c / 4;
  const bool v4 = dtype == TDX_F32 && c % 4 == 0 && ld % 4 == 0 && (((uintptr_t)xa | (uintptr_t)xb) & 15) == 0 && (c <= 256 || c % 256 == 0) && c <= 4096 &&
                  lpr > 0 && (lpr & (lpr - 1)) == 0;
  if (!v4) {   // shapes outside the 16-byte-load kernel: one after the other
    const int rc = tdx_colsum(xa, ld, n, c, outa, accumulate, dtype, ws, stream);
    return rc != TDX_OK ? rc : tdx_colsum(xb, ld, n, c, outb, accumulate, dtype, ws, stream);
  }
  hipStream_t st = (hipStream_t)stream;
  const int windows = (c + 255) / 256, rows_iter = 4 * (64 / lpr);
  int slabs = std::max(1, std::min(kRedBlocks / 2, std::min(512 / windows, (n + 4 * rows_iter - 1) / (4 * rows_iter))));   // (same rule as tdx_colsum: same sums)
  const int rows_per_slab = ((n + slabs - 1) / slabs + rows_iter - 1) / rows_iter * rows_iter;
  slabs = (n + rows_per_slab - 1) / rows_per_slab;
  hipLaunchKernelGGL(colsum_partial_v4_kernel<4>, dim3((unsigned)slabs, (unsigned)windows, 2u), dim3(256), 0, st, (const float*)xa, ld, n, c, lpr, ws,
                     rows_per_slab, (const float*)xb);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((c + 15) / 16), 2u), dim3(256), 0, st, ws, slabs, c, outa, accumulate, outb);
  TDX_CHECK_LAUNCH("colsum_pair");
  return TDX_OK;
}

extern "C" int tdx_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                              double beta2, double eps, double weight_decay, int32_t step, void* stream) {
  TDX_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0, "tdx_adamw_step: null pointer or empty tensor");
  TDX_CHECK_ARG(step >= 1, "tdx_adamw_step: step=%d (1-based)", step);
  // scalar bookkeeping in double, as torch does it in Python floats
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float decay = (float)(1.0 - lr * weight_decay), step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, (long)n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, decay, step_size, bc2_sqrt);
  TDX_CHECK_LAUNCH("adamw_kernel");
  return TDX_OK;
}

static int wgrad_cus() {   // CUs of the current device (per device: a process may drive several)
  static int n_dev[kTdxMaxDevices] = {};
  int& n = n_dev[tdx_device_slot()];
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

static int conv_wgrad_impl(const void* dy, int32_t dy_ld, int32_t dy_off, int32_t cout, const tdx_kseg* seg, int32_t B, int32_t lout,
                           float* dw, int32_t cin_total, int32_t ci_off, int32_t accumulate, int32_t dtype, float* ws, int64_t ws_floats,
                           float* dbias, int32_t dbias_accumulate, void* stream) {
  TDX_CHECK_ARG(dy && seg && seg->src && dw && ws, "tdx_conv_wgrad: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_conv_wgrad: bad dtype %d", dtype);
  TDX_CHECK_ARG(B > 0 && lout > 0 && cout > 0 && dy_off >= 0 && dy_off + cout <= dy_ld, "tdx_conv_wgrad: bad dY shape");
  TDX_CHECK_ARG((int64_t)B * lout < (1ll << 31), "tdx_conv_wgrad: B*lout overflows int32");
  const tdx_kseg& s = *seg;
  TDX_CHECK_ARG(s.taps == 1 || s.taps == 3, "tdx_conv_wgrad: taps=%d (1 or 3)", s.taps);
  TDX_CHECK_ARG(s.stride == 1 || s.stride == 2, "tdx_conv_wgrad: stride=%d (1 or 2)", s.stride);
  TDX_CHECK_ARG(s.cin > 0 && s.ch_off >= 0 && s.ch_off + s.cin <= s.src_ld && s.lin > 0 && s.lvirt > 0, "tdx_conv_wgrad: bad source shape");
  TDX_CHECK_ARG(s.lvirt == s.lin || s.up_scale > 0.f, "tdx_conv_wgrad: lvirt != lin needs up_scale = (float)lin / lvirt");
  const int pad = s.taps == 3 ? 1 : 0;
  TDX_CHECK_ARG((s.lvirt + 2 * pad - s.taps) / s.stride + 1 == lout, "tdx_conv_wgrad: source yields a different number of output rows than lout=%d", lout);
  TDX_CHECK_ARG(ci_off >= 0 && ci_off + s.cin <= cin_total, "tdx_conv_wgrad: ci_off/cin_total do not fit");
  const int N = B * lout;
  const int64_t per_split = (int64_t)cout * s.cin * s.taps;
  hipStream_t st = (hipStream_t)stream;
  const bool s2 = s.stride == 2;
  const bool big = !(g_wgrad_force & 1) && dtype == TDX_F32 && (!s2 || (s.taps == 3 && s.lvirt == s.lin)) && cout % 128 == 0 && s.cin % 128 == 0 && dy_ld % 4 == 0 &&
                   dy_off % 4 == 0 && s.src_ld % 4 == 0 && s.ch_off % 4 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)s.src & 15) == 0;
  int splits;
  if (big) {
    // one 512-thread workgroup per CU at a time, all of equal work: the fewest slabs whose workgroup count fills whole rounds of the
    // CUs (every slab costs one gradient-sized partial written and read back: 100 MB for 512 slabs of a 128 x 384 gradient)
    const int tiles = (cout / 128) * (s.cin / 128), cus = wgrad_cus();
    const int smax = std::max(1, std::min(512, (N + 255) / 256));
    double best = 0.0;
    for (int k = 1; k <= smax; ++k) best = std::max(best, (double)(tiles * k) / (double)(((tiles * k + cus - 1) / cus) * cus));
    splits = 1;
    while (splits < smax && (double)(tiles * splits) / (double)(((tiles * splits + cus - 1) / cus) * cus) < 0.97 * best) ++splits;
    if (g_wgrad_force & 4) splits = std::min(std::max(1, 512 / tiles), (N + 31) / 32);   // (the former rule: two rounds' worth)
  } else {
    // few tiles (the stem's 4 -> 128 and the head's 128 -> 4 channels): the rows are all the parallelism there is, so cut them finer
    // (about 2048 workgroups; 64 slabs left 384 workgroups of a latency-bound loop on 256 CUs: 0.35 ms for 105 MB of dY)
    const int tiles = ((cout + 63) / 64) * ((s.cin + 63) / 64) * s.taps;
    splits = std::max(std::min(64, (N + 2047) / 2048), std::min(512, 2048 / tiles));
    splits = std::max(1, std::min(splits, (N + 255) / 256));
  }
  const bool bias_here = dbias && big;   // the large-tile kernel sums the columns of the dY rows it stages anyway
  if (dbias && !big) {                   // every other shape: the stand-alone column sum first (it is done with ws before the slabs are written)
    TDX_CHECK_ARG(ws_floats >= (int64_t)kRedBlocks * 4096, "tdx_conv_wgrad_bias: workspace smaller than tdx_train_workspace_floats()");
    const int rc = tdx_colsum((const char*)dy + (size_t)dy_off * (dtype == TDX_F32 ? 4 : 2), dy_ld, N, cout, dbias, dbias_accumulate, dtype, ws, stream);
    if (rc != TDX_OK) return rc;
  }
  const int64_t per_slab = per_split + (bias_here ? cout : 0);
  if ((int64_t)splits * per_slab > ws_floats) splits = (int)(ws_floats / per_slab);
  TDX_CHECK_ARG(splits >= 1, "tdx_conv_wgrad: workspace of %lld floats is smaller than one gradient (%lld)", (long long)ws_floats, (long long)per_slab);
  WgradArgs a{(const char*)dy, (const char*)s.src, dy_ld, dy_off, cout, s.src_ld, s.ch_off, s.cin, s.taps, s.stride, s.lin, s.lvirt,
              s.lvirt != s.lin ? s.up_scale : 0.f, N, lout, ((N + splits - 1) / splits + 31) / 32 * 32, ws, nullptr};
  splits = (N + a.rows_per_split - 1) / a.rows_per_split;
  if (bias_here) a.bias_part = ws + (size_t)splits * per_split;   // (per_split is a multiple of 4 floats here: 16-byte aligned like ws)
  if (big) {
    const dim3 grid((unsigned)(cout / 128), (unsigned)(s.cin / 128), (unsigned)splits);
    if (s2) hipLaunchKernelGGL((conv_wgrad_big_kernel<3, true>), grid, dim3(512), 0, st, a);
    else if (s.taps == 3) hipLaunchKernelGGL((conv_wgrad_big_kernel<3, false>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_wgrad_big_kernel<1, false>), grid, dim3(512), 0, st, a);
  } else {
    const dim3 grid((unsigned)((cout + 63) / 64), (unsigned)(((s.cin + 63) / 64) * s.taps), (unsigned)splits);
    if (dtype == TDX_F32) hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv_wgrad_kernel<__bf16>, grid, dim3(256), 0, st, a);
  }
  const int kdim = s.cin * s.taps;
  if (!(g_wgrad_force & 2) && kdim % 4 == 0 && (cin_total * s.taps) % 4 == 0 && (ci_off * s.taps) % 4 == 0 && ((uintptr_t)dw & 15) == 0 && ((uintptr_t)ws & 15) == 0)
  {
    const int wblocks = (int)((per_split / 4 + 15) / 16), bblocks = bias_here && ((uintptr_t)dbias & 15) == 0 ? (cout / 4 + 15) / 16 : 0;
    hipLaunchKernelGGL(wgrad_reduce_v4_kernel, dim3((unsigned)(wblocks + bblocks)), dim3(256), 0, st, ws, splits, cout, kdim / 4, dw,
                       cin_total * s.taps / 4, ci_off * s.taps / 4, accumulate, wblocks, a.bias_part, dbias, dbias_accumulate);
    if (bias_here && !bblocks)
      hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cout + 15) / 16)), dim3(256), 0, st, a.bias_part, splits, cout, dbias, dbias_accumulate);
  } else {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per_split + 255) / 256)), dim3(256), 0, st, ws, splits, cout, s.cin, s.taps, dw,
                       cin_total, ci_off, accumulate);
    if (bias_here) hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cout + 15) / 16)), dim3(256), 0, st, a.bias_part, splits, cout, dbias, dbias_accumulate);
  }
  TDX_CHECK_LAUNCH("conv_wgrad");
  return TDX_OK;
}

extern "C" int tdx_conv_wgrad(const void* dy, int32_t dy_ld, int32_t dy_off, int32_t cout, const tdx_kseg* seg, int32_t B, int32_t lout,
                              float* dw, int32_t cin_total, int32_t ci_off, int32_t accumulate, int32_t dtype, float* ws, int64_t ws_floats,
                              void* stream) {
  return conv_wgrad_impl(dy, dy_ld, dy_off, cout, seg, B, lout, dw, cin_total, ci_off, accumulate, dtype, ws, ws_floats, nullptr, 0, stream);
}

extern "C" int tdx_conv_wgrad_bias(const void* dy, int32_t dy_ld, int32_t dy_off, int32_t cout, const tdx_kseg* seg, int32_t B, int32_t lout,
                                   float* dw, int32_t cin_total, int32_t ci_off, int32_t accumulate, float* dbias, int32_t dbias_accumulate,
                                   int32_t dtype, float* ws, int64_t ws_floats, void* stream) {
  TDX_CHECK_ARG(dbias, "tdx_conv_wgrad_bias: null dbias");
  return conv_wgrad_impl(dy, dy_ld, dy_off, cout, seg, B, lout, dw, cin_total, ci_off, accumulate, dtype, ws, ws_floats, dbias, dbias_accumulate, stream);
}

static int groupnorm_act_bwd_impl(const void* x, int32_t x_ld, const void* dy, int32_t dy_ld, int32_t dy_off, void* dx, int32_t dx_ld,
                                  int32_t dx_off, const float* gamma, const float* beta, float* dgamma_rows, float* dbeta_rows, int32_t B,
                                  int32_t L, int32_t C, int32_t groups, float eps, int32_t act, int32_t accumulate, int32_t dtype,
                                  const GnDropB* drop, void* stream) {
  TDX_CHECK_ARG(x && dy && dx && gamma && beta && dgamma_rows && dbeta_rows, "tdx_groupnorm_act_bwd: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_groupnorm_act_bwd: bad dtype %d", dtype);
  TDX_CHECK_ARG(B > 0 && L > 0 && C > 0 && groups > 0 && C % groups == 0, "tdx_groupnorm_act_bwd: bad shape B=%d L=%d C=%d groups=%d", B, L, C, groups);
  TDX_CHECK_ARG(act == TDX_ACT_NONE || act == TDX_ACT_SWISH || act == TDX_ACT_LEAKY, "tdx_groupnorm_act_bwd: bad act %d", act);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  const int cpg = C / groups;
  if (cpg < 4 || (cpg & (cpg - 1)) != 0 || cpg < vec / 2)
    TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_groupnorm_act_bwd: channels per group = %d; need a power of two >= %d", cpg, vec / 2 > 4 ? vec / 2 : 4);
  int lpr = 8;
  while (lpr * vec < cpg) lpr <<= 1;
  if (lpr > 64 || C % (lpr * vec) != 0)
    TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_groupnorm_act_bwd: C=%d not a multiple of the %d-channel window", C, lpr * vec);
  TDX_CHECK_ARG(x_ld % vec == 0 && dy_ld % vec == 0 && dy_off % vec == 0 && dx_ld % vec == 0 && dx_off % vec == 0 && x_ld >= C &&
                    dy_off + C <= dy_ld && dx_off + C <= dx_ld,
                "tdx_groupnorm_act_bwd: leading dimensions / offsets must be multiples of %d and fit", vec);
  // Long trajectories: 2 or 4 waves per (trajectory, window), at most 8 row iterations each, rows held in registers (a function of L and
  // the window alone, like the forward pass); longer than that: the one-wave sweep that re-reads its rows.
  const int rpi = 64 / lpr;
  int parts = L > 128 ? 4 : (L > 64 ? 2 : 1);
  int rows_per_part = ((L + parts - 1) / parts + rpi - 1) / rpi * rpi;
  if ((g_wgrad_force & 8) || (parts > 1 && rows_per_part / rpi > 8)) parts = 1;
  if (parts == 1) rows_per_part = L;
  const int nit = (g_wgrad_force & 8) ? 1 << 20 : (rows_per_part + rpi - 1) / rpi;
  const long waves = (long)B * (C / (lpr * vec)) * parts;
  const dim3 grid((unsigned)((waves + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (drop)
    launch_gn_bwd<float, true>(nit, parts, rows_per_part, grid, st, x, x_ld, dy, dy_ld, dy_off, dx, dx_ld, dx_off, gamma, beta, dgamma_rows, dbeta_rows, B,
                               L, C, cpg, lpr, eps, act, accumulate, *drop);
  else if (dtype == TDX_F32)
    launch_gn_bwd<float, false>(nit, parts, rows_per_part, grid, st, x, x_ld, dy, dy_ld, dy_off, dx, dx_ld, dx_off, gamma, beta, dgamma_rows, dbeta_rows, B,
                                L, C, cpg, lpr, eps, act, accumulate, GnDropB{});
  else
    launch_gn_bwd<__bf16, false>(nit, parts, rows_per_part, grid, st, x, x_ld, dy, dy_ld, dy_off, dx, dx_ld, dx_off, gamma, beta, dgamma_rows, dbeta_rows, B,
                                 L, C, cpg, lpr, eps, act, accumulate, GnDropB{});
  TDX_CHECK_LAUNCH("groupnorm_act_bwd_kernel");
  return TDX_OK;
}

extern "C" int tdx_groupnorm_act_bwd(const void* x, int32_t x_ld, const void* dy, int32_t dy_ld, int32_t dy_off, void* dx, int32_t dx_ld,
                                     int32_t dx_off, const float* gamma, const float* beta, float* dgamma_rows, float* dbeta_rows, int32_t B,
                                     int32_t L, int32_t C, int32_t groups, float eps, int32_t act, int32_t accumulate, int32_t dtype,
                                     void* stream) {
  return groupnorm_act_bwd_impl(x, x_ld, dy, dy_ld, dy_off, dx, dx_ld, dx_off, gamma, beta, dgamma_rows, dbeta_rows, B, L, C, groups, eps, act, accumulate,
                                dtype, nullptr, stream);
}

extern "C" int tdx_groupnorm_act_dropout_bwd(const void* x, int32_t x_ld, const void* dy, int32_t dy_ld, int32_t dy_off, void* dx, int32_t dx_ld,
                                             int32_t dx_off, const float* gamma, const float* beta, float* dgamma_rows, float* dbeta_rows,
                                             int32_t B, int32_t L, int32_t C, int32_t groups, float eps, int32_t act, int32_t accumulate,
                                             int32_t dtype, float p, uint64_t seed, uint32_t layer, const int32_t* step_ctr, void* stream) {
  TDX_CHECK_ARG(dtype == TDX_F32, "tdx_groupnorm_act_dropout_bwd: fp32 rows only (dtype %d)", dtype);
  TDX_CHECK_ARG(step_ctr && p >= 0.f && p < 1.f, "tdx_groupnorm_act_dropout_bwd: p=%f not in [0, 1) or no step counter", (double)p);
  TDX_CHECK_ARG(dy_off == 0 && dy_ld == C, "tdx_groupnorm_act_dropout_bwd: the mask is keyed by the element index of a plain (B, L, C) gradient");
  const GnDropB d{(uint32_t)((double)p * 4294967296.0), layer, 1.0f / (1.0f - p), seed, step_ctr};
  return groupnorm_act_bwd_impl(x, x_ld, dy, dy_ld, dy_off, dx, dx_ld, dx_off, gamma, beta, dgamma_rows, dbeta_rows, B, L, C, groups, eps, act, accumulate,
                                dtype, &d, stream);
}

extern "C" int tdx_interleave2(const void* even, const void* odd, void* out, int32_t out_ld, int32_t out_off, int32_t B, int32_t L, int32_t C,
                               int32_t accumulate, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(even && out && (odd || L < 2), "tdx_interleave2: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_interleave2: bad dtype %d", dtype);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  TDX_CHECK_ARG(B > 0 && L > 0 && C > 0 && C % vec == 0 && out_ld % vec == 0 && out_off % vec == 0 && out_off + C <= out_ld,
                "tdx_interleave2: bad shape B=%d L=%d C=%d ld=%d off=%d", B, L, C, out_ld, out_off);
  const long n = (long)B * L * (C / vec);
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  const int le = (L + 1) / 2;
  if (dtype == TDX_F32)
    hipLaunchKernelGGL(interleave2_kernel<float>, grid, dim3(256), 0, st, (const float*)even, le, (const float*)odd, (float*)out, out_ld, out_off, B, L, C, accumulate);
  else
    hipLaunchKernelGGL(interleave2_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)even, le, (const __bf16*)odd, (__bf16*)out, out_ld, out_off, B, L, C, accumulate);
  TDX_CHECK_LAUNCH("interleave2_kernel");
  return TDX_OK;
}

extern "C" int tdx_upsample_fold(const void* du, int32_t du_ld, int32_t du_off, void* dx, int32_t dx_ld, int32_t dx_off, int32_t B, int32_t lin,
                                 int32_t lvirt, int32_t C, int32_t accumulate, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(du && dx, "tdx_upsample_fold: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_upsample_fold: bad dtype %d", dtype);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  TDX_CHECK_ARG(B > 0 && lin > 0 && lvirt >= lin && C > 0 && C % vec == 0, "tdx_upsample_fold: bad shape B=%d lin=%d lvirt=%d C=%d", B, lin, lvirt, C);
  TDX_CHECK_ARG(du_ld % vec == 0 && du_off % vec == 0 && du_off + C <= du_ld && dx_ld % vec == 0 && dx_off % vec == 0 && dx_off + C <= dx_ld,
                "tdx_upsample_fold: leading dimensions / offsets must be multiples of %d and fit", vec);
  const long n = (long)B * lin * (C / vec);
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  const float scale = (float)lin / (float)lvirt;
  if (dtype == TDX_F32)
    hipLaunchKernelGGL(upsample_fold_kernel<float>, grid, dim3(256), 0, st, (const float*)du, du_ld, du_off, (float*)dx, dx_ld, dx_off, B, lin, lvirt, C, scale, accumulate);
  else
    hipLaunchKernelGGL(upsample_fold_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)du, du_ld, du_off, (__bf16*)dx, dx_ld, dx_off, B, lin, lvirt, C, scale, accumulate);
  TDX_CHECK_LAUNCH("upsample_fold_kernel");
  return TDX_OK;
}

extern "C" int tdx_act_fwd(const float* z, int32_t z_ld, float* y, int32_t y_ld, int32_t n, int32_t c, int32_t act, void* stream) {
  TDX_CHECK_ARG(z && y && n > 0 && c > 0 && z_ld >= c && y_ld >= c, "tdx_act_fwd: bad arguments");
  TDX_CHECK_ARG(act >= TDX_ACT_NONE && act <= TDX_ACT_SWISH, "tdx_act_fwd: bad act %d", act);
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)(((long)n * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, z_ld, y, y_ld, n, c, act);
  TDX_CHECK_LAUNCH("act_fwd_kernel");
  return TDX_OK;
}

extern "C" int tdx_act_bwd(const float* dy, int32_t dy_ld, const float* ref, int32_t ref_ld, float* dz, int32_t dz_ld, int32_t n, int32_t c,
                           int32_t act, int32_t accumulate, void* stream) {
  TDX_CHECK_ARG(dy && ref && dz && n > 0 && c > 0 && dy_ld >= c && ref_ld >= c && dz_ld >= c, "tdx_act_bwd: bad arguments");
  TDX_CHECK_ARG(act >= TDX_ACT_NONE && act <= TDX_ACT_SWISH, "tdx_act_bwd: bad act %d", act);
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)(((long)n * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dy_ld, ref, ref_ld, dz,
                     dz_ld, n, c, act, accumulate);
  TDX_CHECK_LAUNCH("act_bwd_kernel");
  return TDX_OK;
}

extern "C" int tdx_traj_rowsum(const float* x, int32_t ld, float* out, int32_t out_ld, int32_t B, int32_t L, int32_t c, int32_t accumulate,
                               void* stream) {
  TDX_CHECK_ARG(x && out && B > 0 && L > 0 && c > 0 && ld >= c && out_ld >= c, "tdx_traj_rowsum: bad arguments");
  if (c % 4 == 0 && ld % 4 == 0 && out_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(traj_rowsum_v4_kernel, dim3((unsigned)(((long)B * (c / 4) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, out, out_ld,
                       B, L, c / 4, accumulate);
  else
    hipLaunchKernelGGL(traj_rowsum_kernel, dim3((unsigned)(((long)B * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, out, out_ld, B, L, c,
                       accumulate);
  TDX_CHECK_LAUNCH("traj_rowsum_kernel");
  return TDX_OK;
}

extern "C" int tdx_embedding_bwd(const int64_t* idx, int32_t idx_ld, int32_t ncols, const float* dvec, int32_t dvec_ld, float* dtable, int32_t rows,
                                 int32_t dim, int32_t B, int32_t accumulate, void* stream) {
  TDX_CHECK_ARG(idx && dvec && dtable && ncols > 0 && rows > 0 && dim > 0 && B > 0 && idx_ld >= ncols && dvec_ld >= ncols * dim,
                "tdx_embedding_bwd: bad arguments");
  TDX_CHECK_ARG(dim <= 1024, "tdx_embedding_bwd: dim=%d exceeds 1024 columns per table row", dim);
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, idx, idx_ld, ncols,
                     dvec, dvec_ld, dtable, rows, dim, B, accumulate);
  TDX_CHECK_LAUNCH("embedding_bwd_kernel");
  return TDX_OK;
}

extern "C" int tdx_attention_bwd(const float* qkv, const float* dout, float* dqkv, int32_t B, int32_t L, int32_t C, void* stream) {
  TDX_CHECK_ARG(qkv && dout && dqkv && B > 0 && L > 0 && C > 0, "tdx_attention_bwd: bad arguments");
  if (L > 16) TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_attention_bwd: %d tokens > 16 supported", L);
  const float scale = 1.0f / sqrtf((float)C);
  if ((C & 3) == 0 && ((((uintptr_t)qkv | (uintptr_t)dout) & 15) == 0))
    hipLaunchKernelGGL(attention_bwd_kernel<true>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, qkv, dout, dqkv, L, C, scale);
  else
    hipLaunchKernelGGL(attention_bwd_kernel<false>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, qkv, dout, dqkv, L, C, scale);
  TDX_CHECK_LAUNCH("attention_bwd_kernel");
  return TDX_OK;
}

extern "C" int tdx_dropout(float* x, int32_t ld, int32_t n, int32_t c, float p, uint64_t seed, uint32_t layer, const int32_t* step_ctr,
                           void* stream) {
  TDX_CHECK_ARG(x && step_ctr && n > 0 && c > 0 && ld >= c, "tdx_dropout: bad arguments");
  TDX_CHECK_ARG(p >= 0.f && p < 1.f, "tdx_dropout: p=%f not in [0, 1)", (double)p);
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  const long quads = ((long)n * c + 3) / 4;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, n, c, thresh,
                     1.0f / (1.0f - p), seed, layer, step_ctr);
  TDX_CHECK_LAUNCH("dropout_kernel");
  return TDX_OK;
}

extern "C" int tdx_adamw_step_multi(const tdx_adamw_desc* table_dev, int32_t n, double lr, double beta1, double beta2, double eps,
                                    double weight_decay, int32_t step, void* stream) {
  TDX_CHECK_ARG(table_dev && n > 0 && n <= 65535, "tdx_adamw_step_multi: bad table (n=%d)", n);
  TDX_CHECK_ARG(step >= 1, "tdx_adamw_step_multi: step=%d (1-based)", step);
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float decay = (float)(1.0 - lr * weight_decay), step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(128, (unsigned)n), dim3(256), 0, (hipStream_t)stream, table_dev, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps, decay, step_size, bc2_sqrt);
  TDX_CHECK_LAUNCH("adamw_multi_kernel");
  return TDX_OK;
}

extern "C" int tdx_conv_wgrad_debug(int32_t force) {
  const int old = g_wgrad_force;
  g_wgrad_force = force;
  return old;
}

extern "C" int tdx_pack_weights(const tdx_pack_desc* table_dev, int32_t n, void* stream) {
  TDX_CHECK_ARG(table_dev && n > 0 && n <= 65535, "tdx_pack_weights: bad table (n=%d)", n);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream, table_dev);
  TDX_CHECK_LAUNCH("pack_weights_kernel");
  return TDX_OK;
}
