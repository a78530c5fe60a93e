// Stem Conv1d(2 -> C0) and head Conv1d(C0 -> 2): the two convs whose GEMM K or N is tiny.
// Both are pure HBM streams over a (B, L, C0) activation.
#include "tdx_common.h"

namespace {

// exact n / d for 0 <= n < 2^31: (n * m) >> k with m = floor(2^k / d) + 1, k = 32 + ceil(log2 d) (a 64-bit integer divide per
// row costs more than the row's arithmetic)
struct FastDiv {
  uint64_t m;
  int k, d;
};
inline FastDiv make_fastdiv(int d) {
  int c = 0;
  while ((1ll << c) < d) ++c;
  return FastDiv{(uint64_t)((1ull << (32 + c)) / (uint64_t)d) + 1, 32 + c, d};
}
__device__ __forceinline__ int fd_div(int n, const FastDiv& f) { return (int)(((uint64_t)(uint32_t)n * f.m) >> f.k); }

// Each thread owns one 16-byte channel vector (its 6*VEC weights stay in registers) and walks rows;
// x is (B, 2, L) fp32 channels-first.  A wave writes whole contiguous row segments.
template <typename T>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ out, int B, int L, int C0,
                                                       int rows_per_block, FastDiv fd) {
  constexpr int VEC = Elem<T>::kVec;
  const int vpr = C0 / VEC;                    // vectors per row; host guarantees 256 % vpr == 0 or vpr % 256 == 0
  const int tpr = vpr < 256 ? vpr : 256;       // threads that share one row at a time
  const int rstep = 256 / tpr;
  const int nrows = B * L;  // < 2^31 (checked on the host)
  const int r0 = blockIdx.x * rows_per_block;
  for (int cv = threadIdx.x % tpr; cv < vpr; cv += 256) {
    float wr[VEC][6], bs[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      bs[j] = bias[cv * VEC + j];
#pragma unroll
      for (int k = 0; k < 6; ++k) wr[j][k] = w[(cv * VEC + j) * 6 + k];
    }
    // four rows per trip, all input loads of the trip issued before the first use (one L2 round trip per trip, not per row)
    for (int rr0 = threadIdx.x / tpr; rr0 < rows_per_block; rr0 += 4 * rstep) {
      float xin[4][6];
      int nn[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = rr0 + q * rstep;
        const int n = r0 + rr;
        nn[q] = (rr < rows_per_block && n < nrows) ? n : -1;
        const int ns = nn[q] < 0 ? 0 : n;
        const int b = fd_div(ns, fd), l = ns - b * L;
        const float* xb = x + (size_t)b * 2 * L;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int u = l + t - 1;
            xin[q][ci * 3 + t] = (nn[q] >= 0 && u >= 0 && u < L) ? xb[ci * L + u] : 0.f;
          }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (nn[q] < 0) continue;
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float a = bs[j];
#pragma unroll
          for (int k = 0; k < 6; ++k) a = fmaf(wr[j][k], xin[q][k], a);
          v[j] = a;
        }
        *reinterpret_cast<u32x4*>(out + (size_t)nn[q] * C0 + cv * VEC) = Elem<T>::pack(v);
      }
    }
  }
}

// LPR lanes share one output row (each owns one 16-byte channel vector and keeps its 6*VEC weights in registers),
// 64/LPR rows per wavefront iteration, `iters` iterations per wavefront:
// out[b, o, l] = bias[o] + sum_t sum_c w[o, c, t] * a[b, l+t-1, c], reduced with xor-shuffles inside the row's lanes.
template <typename T>
__global__ __launch_bounds__(256) void head_conv_kernel(const T* __restrict__ a, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B, int L, int C0,
                                                       int lpr, int iters, FastDiv fd) {
  constexpr int VEC = Elem<T>::kVec;
  const int lane = threadIdx.x & 63;
  const int col = lane % lpr, rpw = 64 / lpr;
  const int nrows = B * L;  // < 2^31 (checked on the host)
  const int wave_row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw * iters;
  const float b0 = bias[0], b1 = bias[1];
  for (int cb = col * VEC; cb < C0; cb += lpr * VEC) {  // one pass when lpr * VEC >= C0 (the normal case)
    const bool first = cb == col * VEC;
    float w0[3][VEC], w1[3][VEC];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        w0[t][j] = w[(0 * C0 + cb + j) * 3 + t];
        w1[t][j] = w[(1 * C0 + cb + j) * 3 + t];
      }
    // four row groups per trip: all 12 loads of a trip are issued before the first is used
    for (int it0 = 0; it0 < iters; it0 += 4) {
      u32x4 raw[4][3];
      int bb[4], ll[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = wave_row0 + (it0 + q) * rpw + lane / lpr;
        const bool live = it0 + q < iters && n < nrows;
        bb[q] = live ? fd_div(n, fd) : -1;
        ll[q] = live ? n - bb[q] * L : 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int u = ll[q] + t - 1;
          raw[q][t] = (live && u >= 0 && u < L) ? *reinterpret_cast<const u32x4*>(a + ((size_t)bb[q] * L + u) * C0 + cb)
                                                : (u32x4){0u, 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          float v[VEC];
          Elem<T>::unpack(raw[q][t], v);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            acc0 = fmaf(w0[t][j], v[j], acc0);
            acc1 = fmaf(w1[t][j], v[j], acc1);
          }
        }
        for (int o = 1; o < lpr; o <<= 1) {
          acc0 += __shfl_xor(acc0, o, 64);
          acc1 += __shfl_xor(acc1, o, 64);
        }
        if (bb[q] >= 0 && col == 0) {
          float* o0 = out + ((size_t)bb[q] * 2 + 0) * L + ll[q];
          float* o1 = out + ((size_t)bb[q] * 2 + 1) * L + ll[q];
          *o0 = first ? acc0 + b0 : *o0 + acc0;
          *o1 = first ? acc1 + b1 : *o1 + acc1;
        }
      }
    }
  }
}

}  // namespace

extern "C" int tdx_stem_conv(const float* x, const float* w, const float* bias, void* out, int32_t B, int32_t L,
                             int32_t C0, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(x && w && bias && out, "tdx_stem_conv: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_stem_conv: bad dtype %d", dtype);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  TDX_CHECK_ARG(B > 0 && L > 0 && C0 > 0 && C0 % vec == 0, "tdx_stem_conv: bad shape B=%d L=%d C0=%d", B, L, C0);
  const int vpr = C0 / vec;
  if (!(256 % vpr == 0 || vpr % 256 == 0))
    TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_stem_conv: C0=%d gives %d vectors per row; need a divisor or multiple of 256", C0, vpr);
  const int rows_per_block = 64 * (256 / (vpr < 256 ? vpr : 256));
  const long rows = (long)B * L;
  TDX_CHECK_ARG(rows + 65536 < (1ll << 31), "tdx_stem_conv: B*L overflows int32");
  const FastDiv fd = make_fastdiv(L);
  dim3 grid((unsigned)((rows + rows_per_block - 1) / rows_per_block));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDX_F32)
    hipLaunchKernelGGL(stem_conv_kernel<float>, grid, dim3(256), 0, st, x, w, bias, (float*)out, B, L, C0, rows_per_block, fd);
  else
    hipLaunchKernelGGL(stem_conv_kernel<__bf16>, grid, dim3(256), 0, st, x, w, bias, (__bf16*)out, B, L, C0, rows_per_block, fd);
  TDX_CHECK_LAUNCH("stem_conv_kernel");
  return TDX_OK;
}

extern "C" int tdx_head_conv(const void* a, const float* w, const float* bias, float* out, int32_t B, int32_t L,
                             int32_t C0, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(a && w && bias && out, "tdx_head_conv: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_head_conv: bad dtype %d", dtype);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  TDX_CHECK_ARG(B > 0 && L > 0 && C0 > 0 && C0 % vec == 0, "tdx_head_conv: bad shape B=%d L=%d C0=%d", B, L, C0);
  const long rows = (long)B * L;
  TDX_CHECK_ARG(rows + 65536 < (1ll << 31), "tdx_head_conv: B*L overflows int32");
  const FastDiv fd = make_fastdiv(L);
  int lpr = 1;
  while (lpr < 64 && lpr * vec < C0) lpr <<= 1;  // lanes per row: enough 16-byte vectors to cover C0 in one step if possible
  const int iters = 16;
  const long per_block = 4 * (64 / lpr) * iters;
  dim3 grid((unsigned)((rows + per_block - 1) / per_block));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDX_F32)
    hipLaunchKernelGGL(head_conv_kernel<float>, grid, dim3(256), 0, st, (const float*)a, w, bias, out, B, L, C0, lpr, iters, fd);
  else
    hipLaunchKernelGGL(head_conv_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)a, w, bias, out, B, L, C0, lpr, iters, fd);
  TDX_CHECK_LAUNCH("head_conv_kernel");
  return TDX_OK;
}
