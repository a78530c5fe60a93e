// Conv1d / Linear as ONE gathered GEMM on MFMA (gfx950).
//
//   out[n, co] = act(sum_k W[co, k] * X[n, k] + biases) + res[n, co]
//
// n runs over the B*lout output rows (channels-last), k over (segment, tap, channel).  The X
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
  const int32_t* upmap;
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
  int32_t N, lout, cout;
};

constexpr int kRowBytes = 128;  // one K-step of one tile row: 64 bf16 / 32 fp32
constexpr int kThreads = 256;

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

// TR x TC output tile (rows x channels); 4 waves as 2 (channels) x 2 (rows); each wave (TC/2) x (TR/2).
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
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int row_tile = (j / nco) * 8 + xcd;
  const int n0 = row_tile * TR;
  if (n0 >= p.N) return;  // block-uniform: padding blocks of the last group of 8 row tiles
  const int co0 = (j % nco) * TC;

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

  // ---- K iterator (segment, tap, channel block) ------------------------------------------------
  int it_seg = 0, it_tap = 0, it_ci = 0;
  const char* xrow[RB];
  auto set_rows = [&]() {
    const SegDev& s = p.seg[it_seg];
    const int pad = s.taps == 3 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int u = rb_l[i] * s.stride + it_tap - pad;
      bool ok = rb_ok[i] && u >= 0 && u < s.lvirt;
      int uu = ok ? u : 0;
      int srow = s.upmap ? s.upmap[uu] : uu;
      xrow[i] = ok ? s.src + ((size_t)(rb_b[i] * s.lin + srow) * s.src_ld + s.ch_off + c * VEC) * sizeof(T) : nullptr;
    }
  };
  set_rows();

  u32x4 ra[RA], rx[RB];
  auto load_step = [&]() {
    const SegDev& s = p.seg[it_seg];
    const bool chunk_ok = it_ci + c * VEC < s.cin;
    const size_t koff = (size_t)(s.k0 + it_tap * s.cin + it_ci) * sizeof(T);
    const size_t xoff = (size_t)it_ci * sizeof(T);
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const char* g = (wrow[i] && chunk_ok) ? wrow[i] + koff : zero;
      ra[i] = *reinterpret_cast<const u32x4*>(g);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const char* g = (xrow[i] && chunk_ok) ? xrow[i] + xoff : zero;
      rx[i] = *reinterpret_cast<const u32x4*>(g);
    }
    // advance
    it_ci += BK;
    if (it_ci >= s.cin) {
      it_ci = 0;
      if (++it_tap >= s.taps) {
        it_tap = 0;
        ++it_seg;
      }
      if (it_seg < p.nseg) set_rows();
    }
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
  for (int s = 0; s < p.nsteps; ++s) {
    const bool more = s + 1 < p.nsteps;
    if (more) load_step();
    compute(s & 1);
    if (more) store_step((s + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: bias (+ per-sample / per-step embed bias) -> act -> (+ residual) -> store -----------
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wr * (TR / 2) + ni * 16 + fr;
    if (n >= p.N) continue;
    const int b = n / p.lout;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int co = co0 + wc * (TC / 2) + mi * 16 + fq * 4;
      if (co >= p.cout) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][j];
      if (p.bias) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += t[j];
      }
      if (p.ebias) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.ebias + (size_t)b * p.ebias_ld + p.ebias_off + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += t[j];
      }
      if (p.tbias) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.tbias + p.ebias_off + co);
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
  }
}

template <typename T, int TR, int TC>
int launch(const ConvArgs& a, hipStream_t st) {
  const int nco = (a.cout + TC - 1) / TC, nrow = (a.N + TR - 1) / TR;
  dim3 grid((unsigned)(nco * ((nrow + 7) / 8) * 8));
  size_t lds = 2 * (size_t)(TR + TC) * kRowBytes;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, TR, TC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      TDX_FAIL(TDX_ERR_HIP, "conv_gemm: cannot reserve %zu bytes of LDS", lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_gemm_kernel<T, TR, TC>), grid, dim3(kThreads), lds, st, a);
  TDX_CHECK_LAUNCH("conv_gemm_kernel");
  return TDX_OK;
}

}  // namespace

extern "C" int tdx_conv_gemm(const tdx_conv_desc* d, void* stream) {
  TDX_CHECK_ARG(d != nullptr, "tdx_conv_gemm: null descriptor");
  TDX_CHECK_ARG(d->dtype == TDX_F32 || d->dtype == TDX_BF16, "tdx_conv_gemm: bad dtype %d", d->dtype);
  const int es = d->dtype == TDX_F32 ? 4 : 2;
  const int vec = 16 / es, bk = kRowBytes / es;
  TDX_CHECK_ARG(d->nseg >= 1 && d->nseg <= 4, "tdx_conv_gemm: nseg=%d not in [1,4]", d->nseg);
  TDX_CHECK_ARG(d->B > 0 && d->lout > 0 && d->cout > 0, "tdx_conv_gemm: empty problem B=%d lout=%d cout=%d", d->B, d->lout, d->cout);
  TDX_CHECK_ARG((int64_t)d->B * d->lout < (1ll << 31), "tdx_conv_gemm: B*lout overflows int32");
  TDX_CHECK_ARG(d->cout % 4 == 0 && d->out_off % 4 == 0 && d->out_ld % 4 == 0 && d->out_off + d->cout <= d->out_ld,
                "tdx_conv_gemm: cout/out_off/out_ld must be multiples of 4 and fit (cout=%d off=%d ld=%d)", d->cout, d->out_off, d->out_ld);
  TDX_CHECK_ARG(d->w && d->out, "tdx_conv_gemm: null w/out");
  TDX_CHECK_ARG(!d->res || (d->res_ld % 4 == 0 && d->res_ld >= d->cout), "tdx_conv_gemm: bad res_ld %d", d->res_ld);
  TDX_CHECK_ARG(!(d->ebias || d->tbias) || (d->ebias_off % 4 == 0 && d->ebias_ld % 4 == 0), "tdx_conv_gemm: ebias_off/ld must be multiples of 4");
  TDX_CHECK_ARG(d->act >= TDX_ACT_NONE && d->act <= TDX_ACT_SWISH, "tdx_conv_gemm: bad act %d", d->act);
  ConvArgs a{};
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
    TDX_CHECK_ARG(s.upmap != nullptr || s.lvirt == s.lin, "tdx_conv_gemm: seg %d lvirt != lin needs an upmap", i);
    const int pad = s.taps == 3 ? 1 : 0;
    const int lo = (s.lvirt + 2 * pad - s.taps) / s.stride + 1;
    TDX_CHECK_ARG(lo == d->lout, "tdx_conv_gemm: seg %d yields %d output rows per trajectory, lout=%d", i, lo, d->lout);
    a.seg[i] = SegDev{(const char*)s.src, s.upmap, s.src_ld, s.ch_off, s.cin, s.taps, s.stride, s.lin, s.lvirt, k};
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
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == TDX_F32) return launch<float, 128, 128>(a, st);
  return launch<__bf16, 128, 128>(a, st);
}
