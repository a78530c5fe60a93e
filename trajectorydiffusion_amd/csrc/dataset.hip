// DidiTrajectoryDataset.__getitem__ for a whole batch: z-score + derived attributes.
// One workgroup per trajectory (Dataset/DidiDataset.py:138-174).
#include "tdx_common.h"

namespace {

__device__ __forceinline__ int cell_index(const float* __restrict__ bounds, float v) {
  // torch.searchsorted(bounds, v) - 1 with right=False: (number of boundaries < v) - 1
  int lo = 0, hi = 257;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (bounds[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo - 1;
}

__global__ __launch_bounds__(256) void didi_items_kernel(const float* __restrict__ raw, const float* __restrict__ minutes,
                                                         const float* __restrict__ stats, const float* __restrict__ bounds,
                                                         float* __restrict__ lon_lat, int64_t* __restrict__ cat,
                                                         float* __restrict__ num, float* __restrict__ times, int L) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float mean_t = stats[0], mean_lon = stats[1], mean_lat = stats[2];
  const float std_t = stats[3], std_lon = stats[4], std_lat = stats[5];
  const float* r = raw + (size_t)b * L * 2;
  const float* m = minutes + (size_t)b * L;
  float seg = 0.f;
  for (int l = tid; l < L; l += 256) {
    const float lon = r[2 * l], lat = r[2 * l + 1];
    lon_lat[((size_t)b * 2 + 0) * L + l] = (lon - mean_lon) / std_lon;  // :162 (fp32 cast happened at load, :83)
    lon_lat[((size_t)b * 2 + 1) * L + l] = (lat - mean_lat) / std_lat;
    times[(size_t)b * L + l] = (m[l] - mean_t) / std_t;  // :152
    if (l + 1 < L) {
      const float dx = r[2 * l + 2] - lon, dy = r[2 * l + 3] - lat;
      seg += sqrtf(dx * dx + dy * dy);  // :155 on raw degrees
    }
  }
  seg = wave_sum(seg);
  if ((tid & 63) == 0) red[tid >> 6] = seg;
  __syncthreads();
  if (tid == 0) {
    const float total = (red[0] + red[1]) + (red[2] + red[3]);
    const float tl = total / (std_lon + std_lat);  // :156
    const float m0 = m[0];
    const float day = floorf(m0 / 1440.f) + 1.f;  // :150
    float rem = fmodf(m0, 1440.f);                 // :151 (python-style remainder)
    if (rem < 0.f) rem += 1440.f;
    num[(size_t)b * 3 + 0] = rem / 1440.f;
    num[(size_t)b * 3 + 1] = tl;
    num[(size_t)b * 3 + 2] = tl / (float)(L - 1);  // :159
    cat[(size_t)b * 5 + 4] = (int64_t)day;
  }
  if (tid >= 64 && tid < 68) {  // :166-167 on the normalised start / end points
    const int k = tid - 64;     // 0: src lon, 1: src lat, 2: dst lon, 3: dst lat
    const int l = k < 2 ? 0 : L - 1;
    const float v = (k & 1) ? (r[2 * l + 1] - mean_lat) / std_lat : (r[2 * l] - mean_lon) / std_lon;
    cat[(size_t)b * 5 + k] = cell_index(bounds, v);
  }
}

__global__ __launch_bounds__(256) void didi_denorm_kernel(const float* __restrict__ lon_lat, const float* __restrict__ stats,
                                                          float* __restrict__ out, int B, int L) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * L) return;
  const int b = (int)(i / L), l = (int)(i % L);
  out[2 * i] = lon_lat[((size_t)b * 2 + 0) * L + l] * stats[4] + stats[1];
  out[2 * i + 1] = lon_lat[((size_t)b * 2 + 1) * L + l] * stats[5] + stats[2];
}

__global__ __launch_bounds__(64) void cell_index_kernel(const float* __restrict__ v, const float* __restrict__ bounds, int64_t* __restrict__ out, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n) out[i] = cell_index(bounds, v[i]);
}

}  // namespace

extern "C" int tdx_cell_index(const float* values, const float* bounds, int64_t* out, int32_t n, void* stream) {
  TDX_CHECK_ARG(values && bounds && out && n > 0, "tdx_cell_index: bad arguments");
  hipLaunchKernelGGL(cell_index_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, values, bounds, out, n);
  TDX_CHECK_LAUNCH("cell_index_kernel");
  return TDX_OK;
}

extern "C" int tdx_didi_items(const float* lon_lat_raw, const float* minutes, const float* stats, const float* bounds,
                              float* lon_lat, int64_t* cat, float* num, float* times, int32_t B, int32_t L, void* stream) {
  TDX_CHECK_ARG(lon_lat_raw && minutes && stats && bounds && lon_lat && cat && num && times, "tdx_didi_items: null pointer");
  TDX_CHECK_ARG(B > 0 && L >= 2, "tdx_didi_items: need B > 0 and L >= 2 (B=%d L=%d)", B, L);
  hipLaunchKernelGGL(didi_items_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, lon_lat_raw, minutes, stats, bounds, lon_lat,
                     cat, num, times, L);
  TDX_CHECK_LAUNCH("didi_items_kernel");
  return TDX_OK;
}

extern "C" int tdx_didi_denorm(const float* lon_lat, const float* stats, float* out_deg, int32_t B, int32_t L, void* stream) {
  TDX_CHECK_ARG(lon_lat && stats && out_deg && B > 0 && L > 0, "tdx_didi_denorm: bad arguments");
  const long n = (long)B * L;
  hipLaunchKernelGGL(didi_denorm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lon_lat, stats,
                     out_deg, B, L);
  TDX_CHECK_LAUNCH("didi_denorm_kernel");
  return TDX_OK;
}
