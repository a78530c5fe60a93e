// Evaluation metrics over sampled trajectories (reference: eval_trajunet.ipynb cell 12, where they are
// per-point Python loops): occupancy histogram of all / first / last points on an n x n grid spanning the set's own
// bounding box, histogram of path lengths, Jensen-Shannon divergence of two such histograms.
// HBM-bound streaming kernels: every coordinate is read twice (bounding box, then binning) or once (lengths).  Counts are
// integer atomics (order-independent), so results are bitwise reproducible.
#include <algorithm>

#include "tdx_common.h"

namespace {

constexpr int kMaxBlocks = 1024;

// torch.linspace(start, end, n)[i] in fp32 (ATen RangeFactories: two half-ranges, no contraction)
__device__ __forceinline__ float linspace_at(float start, float end, float step, int n, int i) {
  return i < n / 2 ? __fadd_rn(start, __fmul_rn(step, (float)i)) : __fsub_rn(end, __fmul_rn(step, (float)(n - 1 - i)));
}
// torch.searchsorted(grid, v) (side='left'): the number of grid values < v
__device__ __forceinline__ int searchsorted_left(float start, float end, float step, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (linspace_at(start, end, step, n, mid) < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void point_of(const float* __restrict__ traj, int L, int l0, int lc, long i, float& lon, float& lat) {
  const long b = i / lc;
  const int l = l0 + (int)(i - b * lc);
  lon = traj[(b * 2) * L + l];
  lat = traj[(b * 2 + 1) * L + l];
}

__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ traj, long npts, int L, int l0, int lc,
                                                            float* __restrict__ part) {
  __shared__ float red[4][4];
  float mn0 = INFINITY, mx0 = -INFINITY, mn1 = INFINITY, mx1 = -INFINITY;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npts; i += (long)gridDim.x * 256) {
    float lon, lat;
    point_of(traj, L, l0, lc, i, lon, lat);
    mn0 = fminf(mn0, lon);
    mx0 = fmaxf(mx0, lon);
    mn1 = fminf(mn1, lat);
    mx1 = fmaxf(mx1, lat);
  }
  for (int o = 32; o >= 1; o >>= 1) {
    mn0 = fminf(mn0, __shfl_xor(mn0, o, 64));
    mx0 = fmaxf(mx0, __shfl_xor(mx0, o, 64));
    mn1 = fminf(mn1, __shfl_xor(mn1, o, 64));
    mx1 = fmaxf(mx1, __shfl_xor(mx1, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    float* r = red[threadIdx.x >> 6];
    r[0] = mn0, r[1] = mx0, r[2] = mn1, r[3] = mx1;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    float v = red[0][k];
    for (int w = 1; w < 4; ++w) v = (k & 1) ? fmaxf(v, red[w][k]) : fminf(v, red[w][k]);
    part[blockIdx.x * 4 + k] = v;
  }
}
__global__ __launch_bounds__(64) void minmax_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out4) {
  const int k = threadIdx.x;
  if (k >= 4) return;
  float v = part[k];
  for (int i = 1; i < nparts; ++i) v = (k & 1) ? fmaxf(v, part[i * 4 + k]) : fminf(v, part[i * 4 + k]);
  out4[k] = v;  // min_lon, max_lon, min_lat, max_lat
}

__global__ __launch_bounds__(256) void point_hist_kernel(const float* __restrict__ traj, long npts, int L, int l0, int lc,
                                                        const float* __restrict__ box, int n, int* __restrict__ counts) {
  extern __shared__ int hist[];
  for (int i = threadIdx.x; i < n * n; i += 256) hist[i] = 0;
  __syncthreads();
  const float lo0 = box[0], hi0 = box[1], lo1 = box[2], hi1 = box[3];
  const float st0 = __fdiv_rn(__fsub_rn(hi0, lo0), (float)(n - 1)), st1 = __fdiv_rn(__fsub_rn(hi1, lo1), (float)(n - 1));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npts; i += (long)gridDim.x * 256) {
    float lon, lat;
    point_of(traj, L, l0, lc, i, lon, lat);
    // every point lies inside the bounding box, so both indices are < n (the maximum itself lands on grid[n - 1])
    const int a = min(searchsorted_left(lo0, hi0, st0, n, lon), n - 1), b = min(searchsorted_left(lo1, hi1, st1, n, lat), n - 1);
    atomicAdd(&hist[a * n + b], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n * n; i += 256)
    if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

// one wave per trajectory: path length = sum_l sqrt(dlon^2 + dlat^2), then its bin on linspace(lo, hi, n)
__global__ __launch_bounds__(256) void length_hist_kernel(const float* __restrict__ traj, int B, int L, float lo, float hi, int n,
                                                         int* __restrict__ counts, float* __restrict__ lengths, int* __restrict__ err) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;  // wave-uniform
  const float* x = traj + (size_t)b * 2 * L;
  const float* y = x + L;
  float acc = 0.f;
  for (int l = lane; l + 1 < L; l += 64) {
    const float dx = __fsub_rn(x[l + 1], x[l]), dy = __fsub_rn(y[l + 1], y[l]);
    acc += sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    if (lengths) lengths[b] = acc;
    const float step = __fdiv_rn(__fsub_rn(hi, lo), (float)(n - 1));
    const int idx = searchsorted_left(lo, hi, step, n, acc);
    if (idx < n) atomicAdd(&counts[idx], 1);
    else err[0] = 1;  // the reference indexes a 64-entry tensor with 64 here: IndexError
  }
}

// out[i] = (count[i] + add) / sum(count)   (cell 12: "(point_count + 1) / point_count.sum()", "(length_dist + 0.1) / length_dist.sum()")
__global__ __launch_bounds__(256) void hist_normalize_kernel(const int* __restrict__ counts, int n, float add, int normalize,
                                                            float* __restrict__ out) {
  __shared__ long red[256];
  long s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += counts[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float total = (float)red[0];
  for (int i = threadIdx.x; i < n; i += 256) out[i] = normalize ? __fdiv_rn(__fadd_rn((float)counts[i], add), total) : (float)counts[i];
}

// 0.5 * (kl_div(log P, M) + kl_div(log Q, M)) with M = (P + Q) / 2, reduction 'batchmean' (sum / batch)
__global__ __launch_bounds__(256) void jsd_kernel(const float* __restrict__ P, const float* __restrict__ Q, long n, int batch,
                                                 float* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float p = P[i], q = Q[i];
    const float m = 0.5f * (p + q);
    const float xl = m > 0.f ? m * logf(m) : 0.f;  // xlogy(m, m)
    acc += (double)(xl - m * logf(p)) + (double)(xl - m * logf(q));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(0.5 * red[0] / (double)batch);
}

}  // namespace

extern "C" int64_t tdx_metric_workspace_bytes(int32_t n_bins_total) { return (int64_t)kMaxBlocks * 16 + 16 + (int64_t)n_bins_total * 4; }

extern "C" int tdx_point_distribution(const float* traj, int32_t B, int32_t L, int32_t l_first, int32_t l_count, int32_t n_grids,
                                      int32_t normalize, float* out, void* ws, int64_t ws_bytes, void* stream) {
  TDX_CHECK_ARG(traj && out && ws, "tdx_point_distribution: null pointer");
  TDX_CHECK_ARG(B > 0 && L > 0 && l_first >= 0 && l_count > 0 && l_first + l_count <= L, "tdx_point_distribution: bad point range [%d, %d) of L=%d, B=%d",
                l_first, l_first + l_count, L, B);
  TDX_CHECK_ARG(n_grids >= 2 && n_grids <= 64, "tdx_point_distribution: n_grids=%d (2..64)", n_grids);
  const int nn = n_grids * n_grids;
  TDX_CHECK_ARG(ws_bytes >= tdx_metric_workspace_bytes(nn), "tdx_point_distribution: workspace of %lld bytes is too small", (long long)ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  float* box = part + kMaxBlocks * 4;
  int* counts = (int*)(box + 4);
  const long npts = (long)B * l_count;
  const int blocks = (int)std::min<long>(kMaxBlocks, (npts + 255) / 256);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(blocks), dim3(256), 0, st, traj, npts, L, l_first, l_count, part);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, st, part, blocks, box);
  if (hipMemsetAsync(counts, 0, (size_t)nn * 4, st) != hipSuccess) TDX_FAIL(TDX_ERR_HIP, "tdx_point_distribution: memset failed");
  hipLaunchKernelGGL(point_hist_kernel, dim3(blocks), dim3(256), (size_t)nn * 4, st, traj, npts, L, l_first, l_count, box, n_grids, counts);
  hipLaunchKernelGGL(hist_normalize_kernel, dim3(1), dim3(256), 0, st, counts, nn, 1.0f, normalize, out);
  TDX_CHECK_LAUNCH("point_distribution");
  return TDX_OK;
}

extern "C" int tdx_length_distribution(const float* traj, int32_t B, int32_t L, float lo, float hi, int32_t n_bins, float* out,
                                       float* lengths, int32_t* err, void* ws, int64_t ws_bytes, void* stream) {
  TDX_CHECK_ARG(traj && out && err && ws, "tdx_length_distribution: null pointer");
  TDX_CHECK_ARG(B > 0 && L > 0 && n_bins >= 2 && n_bins <= 4096 && hi > lo, "tdx_length_distribution: bad arguments B=%d L=%d bins=%d", B, L, n_bins);
  TDX_CHECK_ARG(ws_bytes >= tdx_metric_workspace_bytes(n_bins), "tdx_length_distribution: workspace of %lld bytes is too small", (long long)ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  int* counts = (int*)((float*)ws + kMaxBlocks * 4 + 4);
  if (hipMemsetAsync(counts, 0, (size_t)n_bins * 4, st) != hipSuccess) TDX_FAIL(TDX_ERR_HIP, "tdx_length_distribution: memset failed");
  hipLaunchKernelGGL(length_hist_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, traj, B, L, lo, hi, n_bins, counts, lengths, err);
  hipLaunchKernelGGL(hist_normalize_kernel, dim3(1), dim3(256), 0, st, counts, n_bins, 0.1f, 1, out);
  TDX_CHECK_LAUNCH("length_distribution");
  return TDX_OK;
}

extern "C" int tdx_jsd(const float* P, const float* Q, int64_t n, int32_t batch, float* out, void* stream) {
  TDX_CHECK_ARG(P && Q && out && n > 0 && batch > 0, "tdx_jsd: bad arguments");
  hipLaunchKernelGGL(jsd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, P, Q, (long)n, batch, out);
  TDX_CHECK_LAUNCH("jsd_kernel");
  return TDX_OK;
}
