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

// Stem conv that also leaves the block partial sums of its output (the first half of the split GroupNorm that follows it):
// a wave owns 32 output channels (4 x 16-byte vectors) of ONE 16-row block of one trajectory, lane = (vector, row), so the
// 16 rows of a block sit in the 16 lanes of a DPP row and the sums are a fixed butterfly.  bf16 output.
template <int CTRL>
__device__ __forceinline__ float dpp_rot(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_total(float v) {
  v += dpp_rot<0x128>(v);
  v += dpp_rot<0x124>(v);
  v += dpp_rot<0x122>(v);
  v += dpp_rot<0x121>(v);
  return v;
}
__global__ __launch_bounds__(256) void stem_conv_stats_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, __bf16* __restrict__ out, float* __restrict__ stats,
                                                             int stats_ld, int stats_off, int B, int L, int C0, int nblk, int items_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int wpr = C0 / 32;                       // waves needed for one block's channels
  const long nitems = (long)B * nblk * wpr;      // (trajectory, block, 32-channel slice)
  for (int it = 0; it < items_per_wg; it += 4) {
    const long item = (long)blockIdx.x * items_per_wg + it + wave;
    if (item >= nitems) return;                  // wave-uniform
    const int cs = (int)(item % wpr);
    const long tb = item / wpr;
    const int blk = (int)(tb % nblk), b = (int)(tb / nblk);
    const int ch = cs * 32 + fq * 8;
    const int l = blk * 16 + fr;
    const bool ok = l < L;
    const float* xb = x + (size_t)b * 2 * L;
    float xin[6];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int u = l + t - 1;
        xin[ci * 3 + t] = (ok && u >= 0 && u < L) ? xb[ci * L + u] : 0.f;
      }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = bias[ch + j];
#pragma unroll
      for (int k = 0; k < 6; ++k) a = fmaf(w[(ch + j) * 6 + k], xin[k], a);   // same order as stem_conv_kernel
      v[j] = a;
    }
    if (ok) *reinterpret_cast<u32x4*>(out + ((size_t)b * L + l) * C0 + ch) = Elem<__bf16>::pack(v);
    float s[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float a0 = ok ? v[4 * h] : 0.f, a1 = ok ? v[4 * h + 1] : 0.f, a2 = ok ? v[4 * h + 2] : 0.f, a3 = ok ? v[4 * h + 3] : 0.f;
      s[2 * h] = row16_total((a0 + a1) + (a2 + a3));
      s[2 * h + 1] = row16_total((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3));
    }
    if (fr == 0)
      *reinterpret_cast<f32x4*>(stats + (((size_t)b * nblk + blk) * stats_ld + stats_off + (ch >> 2)) * 2) = (f32x4){s[0], s[1], s[2], s[3]};
  }
}

// Head conv on a RAW activation x with the head's GroupNorm (no activation, Models/TrajUNet.py:42-45) applied while it is
// loaded: a = (x * rstd + (-mean * rstd)) * gamma + beta per element, the conv's zero padding applies to a.  Same lane layout
// and reduction as head_conv_kernel.  bf16 input.
__global__ __launch_bounds__(256) void head_conv_norm_kernel(const __bf16* __restrict__ x, const float* __restrict__ mr, int mr_ld, int mr_off,
                                                            int cpg_shift, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                                                            int B, int L, int C0, int lpr, int iters, FastDiv fd) {
  constexpr int VEC = 8;
  const int lane = threadIdx.x & 63;
  const int col = lane % lpr, rpw = 64 / lpr;
  const int nrows = B * L;
  const int wave_row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw * iters;
  const float b0 = bias[0], b1 = bias[1];
  const int cb = col * VEC;                      // host guarantees lpr * VEC == C0
  float w0[3][VEC], w1[3][VEC], gm[VEC], bt[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    gm[j] = gamma[cb + j];
    bt[j] = beta[cb + j];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      w0[t][j] = w[(0 * C0 + cb + j) * 3 + t];
      w1[t][j] = w[(1 * C0 + cb + j) * 3 + t];
    }
  }
  const int g0 = cb >> cpg_shift, g1 = (cb + 4) >> cpg_shift;
  for (int it0 = 0; it0 < iters; it0 += 4) {
    u32x4 raw[4][3];
    float2 m0[4], m1[4];
    int bb[4], ll[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = wave_row0 + (it0 + q) * rpw + lane / lpr;
      const bool live = it0 + q < iters && n < nrows;
      bb[q] = live ? fd_div(n, fd) : -1;
      ll[q] = live ? n - bb[q] * L : 0;
      const float* mrow = mr + ((size_t)(live ? bb[q] : 0) * mr_ld + mr_off) * 2;
      m0[q] = *reinterpret_cast<const float2*>(mrow + g0 * 2);
      m1[q] = *reinterpret_cast<const float2*>(mrow + g1 * 2);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int u = ll[q] + t - 1;
        raw[q][t] = (live && u >= 0 && u < L) ? *reinterpret_cast<const u32x4*>(x + ((size_t)bb[q] * L + u) * C0 + cb) : (u32x4){0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int u = ll[q] + t - 1;
        const bool tap_ok = bb[q] >= 0 && u >= 0 && u < L;
        float v[VEC];
        Elem<__bf16>::unpack(raw[q][t], v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float a = fmaf(fmaf(v[j], j < 4 ? m0[q].x : m1[q].x, j < 4 ? m0[q].y : m1[q].y), gm[j], bt[j]);
          // (the stand-alone GroupNorm stores bf16: round here too, so both dataflows feed the conv the same values)
          const float ar = tap_ok ? (float)(__bf16)a : 0.f;
          acc0 = fmaf(w0[t][j], ar, acc0);
          acc1 = fmaf(w1[t][j], ar, acc1);
        }
      }
      for (int o = 1; o < lpr; o <<= 1) {
        acc0 += __shfl_xor(acc0, o, 64);
        acc1 += __shfl_xor(acc1, o, 64);
      }
      if (bb[q] >= 0 && col == 0) {
        out[((size_t)bb[q] * 2 + 0) * L + ll[q]] = acc0 + b0;
        out[((size_t)bb[q] * 2 + 1) * L + ll[q]] = acc1 + b1;
      }
    }
  }
}

}  // namespace

extern "C" int tdx_stem_conv_stats(const float* x, const float* w, const float* bias, void* out, float* stats, int32_t stats_ld,
                                   int32_t stats_off, int32_t B, int32_t L, int32_t C0, void* stream) {
  TDX_CHECK_ARG(x && w && bias && out && stats, "tdx_stem_conv_stats: null pointer");
  TDX_CHECK_ARG(B > 0 && L > 0 && C0 > 0 && C0 % 32 == 0, "tdx_stem_conv_stats: bad shape B=%d L=%d C0=%d (C0 must be a multiple of 32)", B, L, C0);
  TDX_CHECK_ARG(stats_off >= 0 && stats_off % 2 == 0 && stats_ld % 2 == 0 && stats_off + C0 / 4 <= stats_ld && ((uintptr_t)stats & 15) == 0 &&
                ((uintptr_t)out & 15) == 0, "tdx_stem_conv_stats: stats layout / alignment");
  const int nblk = (L + 15) / 16;
  const long items = (long)B * nblk * (C0 / 32);
  const int per_wg = 16;
  hipLaunchKernelGGL(stem_conv_stats_kernel, dim3((unsigned)((items + per_wg - 1) / per_wg)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                     (__bf16*)out, stats, stats_ld, stats_off, B, L, C0, nblk, per_wg);
  TDX_CHECK_LAUNCH("stem_conv_stats_kernel");
  return TDX_OK;
}

extern "C" int tdx_head_conv_norm(const void* x, const float* mr, int32_t mr_ld, int32_t mr_off, int32_t cpg, const float* gamma,
                                  const float* beta, const float* w, const float* bias, float* out, int32_t B, int32_t L, int32_t C0,
                                  void* stream) {
  TDX_CHECK_ARG(x && mr && gamma && beta && w && bias && out, "tdx_head_conv_norm: null pointer");
  TDX_CHECK_ARG(B > 0 && L > 0 && C0 >= 8 && C0 <= 512 && (C0 & (C0 - 1)) == 0, "tdx_head_conv_norm: C0=%d must be a power of two in [8, 512]", C0);
  TDX_CHECK_ARG(cpg >= 4 && (cpg & (cpg - 1)) == 0 && mr_off >= 0 && mr_off + C0 / cpg <= mr_ld, "tdx_head_conv_norm: cpg=%d / mr layout", cpg);
  const long rows = (long)B * L;
  TDX_CHECK_ARG(rows + 65536 < (1ll << 31), "tdx_head_conv_norm: B*L overflows int32");
  int shift = 2;
  while ((1 << shift) < cpg) ++shift;
  const FastDiv fd = make_fastdiv(L);
  const int lpr = C0 / 8, iters = 16;
  const long per_block = 4 * (64 / lpr) * iters;
  hipLaunchKernelGGL(head_conv_norm_kernel, dim3((unsigned)((rows + per_block - 1) / per_block)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)x, mr, mr_ld, mr_off, shift, gamma, beta, w, bias, out, B, L, C0, lpr, iters, fd);
  TDX_CHECK_LAUNCH("head_conv_norm_kernel");
  return TDX_OK;
}

extern "C" int tdx_stem_conv(const float* x, const float* w, const float* bias, void* out, int32_t B, int32_t L,
                             int32_t C0, int32_t dtype, void* stream) {
  TDX_CHECK_ARG(x && w && bias && out, "tdx_stem_conv: null pointer");
  TDX_CHECK_ARG(dtype == TDX_F32 || dtype == TDX_BF16, "tdx_stem_conv: bad dtype %d", dtype);
  const int vec = dtype == TDX_F32 ? 4 : 8;
  TDX_CHECK_ARG(B > 0 && L > 0 && C0 > 0 && C0 % vec == 0, "tdx_stem_conv: bad shape B=%d L=%d C0=%d", B, L, C0);
  const int vpr = C0 / vec;
  if (!(256 % vpr == 0 || vpr % 256 == 0))
    TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_stem_conv: C0=%d gives %d vectors per row; need a divisor or multiple of 256", C0, vpr);
  // 16 rows per thread (64 before: 0.051 -> 0.040 ms at B=2048, 800 -> 3200 workgroups at B=4096; the 56 weight / bias loads per
  // thread come from L1 and do not show)
  const int rows_per_block = 16 * (256 / (vpr < 256 ? vpr : 256));
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
  const int iters = 8;
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
