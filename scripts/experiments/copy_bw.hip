// Device-copy bandwidth probe: which access pattern moves the most bytes on this part (the GroupNorm pass is priced against it).
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o scripts/experiments/libcopy_bw.so scripts/experiments/copy_bw.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT, bool CHUNKED>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n) {
  const long nthreads = (long)gridDim.x * 256;
  if (CHUNKED) {   // every workgroup walks one contiguous chunk
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (long i = lo + threadIdx.x; i < hi; i += 256L * UNROLL) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const long j = i + 256L * u;
        if (j < hi) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const long j = i + 256L * u;
        if (j < hi) {
          if (NT) __builtin_nontemporal_store(v[u], dst + j);
          else dst[j] = v[u];
        }
      }
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += nthreads * UNROLL) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const long j = i + nthreads * u;
        if (j < n) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const long j = i + nthreads * u;
        if (j < n) {
          if (NT) __builtin_nontemporal_store(v[u], dst + j);
          else dst[j] = v[u];
        }
      }
    }
  }
}

extern "C" int copy_probe(const void* src, void* dst, long bytes, int variant, int blocks, void* stream) {
  const long n = bytes / 16;
  hipStream_t st = (hipStream_t)stream;
  const u32x4* s = (const u32x4*)src;
  u32x4* d = (u32x4*)dst;
#define RUN(U, NT, CH) hipLaunchKernelGGL((copy_kernel<U, NT, CH>), dim3(blocks), dim3(256), 0, st, s, d, n)
  switch (variant) {
    case 0: RUN(1, false, false); break;
    case 1: RUN(4, false, false); break;
    case 2: RUN(8, false, false); break;
    case 3: RUN(4, true, false); break;
    case 4: RUN(8, true, false); break;
    case 5: RUN(4, false, true); break;
    case 6: RUN(8, false, true); break;
    case 7: RUN(8, true, true); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
