// Conv1d(128 -> 128, k=3) on the full-resolution levels of TrajUNet (L = 200 / 100): the layers where the GEMM is so
// narrow (K = 384, N = 128) that the conv is bound by HBM, not by the matrix cores, and where a separate GroupNorm pass
// over the activation costs as much as the conv itself.  One persistent kernel, built around that:
//
//   * the whole weight matrix (128 x 384 bf16 = 96 KB) lives in REGISTERS: 8 waves = 4 channel quarters x 2 row halves, a wave
//     keeps the 24 A fragments of its 32 output channels (96 VGPRs) for the life of the kernel, so the K loop issues no weight
//     traffic at all (the general kernel re-streams 96 KB of weights through the LDS for every 128-row tile);
//   * the activation window of a tile (all 128 input channels of the <= 258 source rows that 128 output rows touch) is
//     brought in by LDS-DMA, double-buffered ACROSS tiles: the window of tile t + 1 lands while tile t is processed;
//   * optional GroupNorm(+Swish) of the INPUT is applied in place in the LDS window, once per element, from block partial
//     sums that the producer of the tensor left behind (so the normalised tensor never exists in memory);
//   * the K loop (6 K-steps x 2 halves) then runs with NO barrier and NO memory instruction: LDS fragment reads + MFMA only;
//   * the epilogue works straight from the accumulator layout: the rows of the A operand are permuted so that a lane's two
//     16x16 sub-tiles hold 8 CONSECUTIVE output channels of one row, i.e. one 16-byte store (no LDS transpose), and it can
//     emit the block partial sums {sum, sum of squares} a later GroupNorm needs.
//
// Rows are processed in a PADDED row space: every trajectory occupies Lp = ceil(lout / 16) * 16 rows, so a 16-row MFMA
// sub-tile never straddles two trajectories.  That makes the statistics a fixed-order sum per (trajectory, 16-row block,
// 4-channel granule): bitwise reproducible, no atomics, and independent of how a batch is sharded.  The padded rows cost
// MFMA time only, which these layers have to spare.
//
// Results are bit-identical to the general kernels of conv_gemm.hip for descriptors both can run (same K order per output
// element, same epilogue association).
#include <type_traits>
#include <utility>

#include "tdx_common.h"

namespace {

// Two 4-wave workgroups share a CU (one wave of each per SIMD).  They run independently, so the phases of one (GroupNorm
// transform and epilogue on the vector ALU, output stores) fall under the K loop of the other; a single 8-wave workgroup has all
// its waves in the same phase at the same time, separated by barriers, and every unit idles while the others work.
// LDS plan, per workgroup (half of the CU's 160 KB).  The bytes a workgroup keeps in flight are the ring stages it has requested.  vmcnt is one
// in-order queue per wave: a load that is waited for also waits for every older request, so a tile's loads can only stay in
// flight for several tiles if NOTHING younger is ever waited for.  Hence every per-tile operand comes in by LDS-DMA, as one ring
// stage: the activation window (the <= 66 / <= 129 consecutive source rows that the tile's 64 output rows touch, as two
// 64-channel blocks of 128-byte rows with conv_gemm.hip's XOR swizzle, each followed by one permanently-zero row), the residual
// rows (RES), the tile's per-trajectory embed rows and its {rstd, -mean * rstd} rows.  The loop body contains no other load.
constexpr int kTileRows = 64;                    // output (padded) rows per tile
constexpr int kMaxTraj = 4;                      // trajectories a tile may touch: 64 padded rows of trajectories of Lp >= 32 padded rows start in at
                                                 // most 3 of them; the host gate (narrow_plan: tr / lp + 2 <= kMaxTraj) keeps one in hand
constexpr int kMrBytes = kMaxTraj * 32 * 8;      // {rstd, -mean * rstd} per (trajectory, group)
constexpr int kEbBytes = kMaxTraj * 128 * 4;     // per-trajectory embed rows
constexpr int kResBytes = kTileRows * 256;       // residual rows of the tile (16-byte chunks XOR-swizzled by the row)
constexpr int kGbBytes = 2 * 128 * 4;            // gamma | beta of the input GroupNorm
constexpr int kTailBytes = kGbBytes + 512 + 1024;   // gamma/beta, bias + per-step row, DMA dummy target
template <int MODE, bool RES>
struct Ring {
  static constexpr int WROWS = MODE == 1 ? 136 : 72;           // DMA area of one channel-block window, whole 8-row groups (>= 2 * 64 + 1 / 64 + 2 rows)
  static constexpr int WBYTES = (WROWS + 1) * 128;             // + the zero row
  static constexpr int ZERO_OFF = WROWS * 128;
  static constexpr int NG = (WROWS / 8 + 3) / 4;               // window groups per wave (every wave issues exactly NG x 2 window DMAs)
  static constexpr int RES_OFF = 2 * WBYTES;
  static constexpr int EB_OFF = RES_OFF + (RES ? kResBytes : 0);
  static constexpr int MR_OFF = EB_OFF + kEbBytes;
  static constexpr int STAGE = MR_OFF + kMrBytes;
  static constexpr int kBudget = 80 * 1024;                    // per workgroup: TWO workgroups share a CU's 160 KB
  static constexpr int NBUF = (kBudget - kTailBytes) / STAGE >= 4 ? 4 : ((kBudget - kTailBytes) / STAGE >= 3 ? 3 : 2);
  static constexpr int D = NBUF - 1;                           // tiles requested ahead
  static constexpr int NDMA = NG * 2 + (RES ? 4 : 0) + 1;      // DMA instructions per wave and stage
  static_assert(NBUF * STAGE + kTailBytes <= kBudget, "ring does not fit half of the LDS");
};
template <int MODE, bool RES>
constexpr int lds_bytes() { return Ring<MODE, RES>::NBUF * Ring<MODE, RES>::STAGE + kTailBytes; }
constexpr uint32_t kOob = 0x80000000u;

struct NarrowArgs {
  const char* src;
  int src_ld, ch_off, lin, lvirt;
  float up_scale;
  long src_rows;  // B * lin
  const char* w;
  const float* bias;
  const float* ebias;
  const float* tbias;
  int ebias_ld, ebias_off;
  const char* res;
  int res_ld, act;
  char* out;
  int out_ld, out_off;
  int B, lout, Lp, ntiles;
  uint64_t div_m, lin_m;  // p / Lp and srow / lin as multiply-shift
  int div_k, lin_k;
  const float* nmr;     // (B, nmr_ld) float2 {rstd, -mean * rstd} per (trajectory, group), from tdx_groupnorm_finalize
  const float* ngamma;
  const float* nbeta;
  int nact, nmr_ld, nmr_off, ncpg_shift;   // group of input channel c = c >> ncpg_shift
  float* sout;
  int sout_ld, sout_off, nblk_out;
  unsigned long long* dbg;   // diagnostic builds only (-DNARROW_STAMPS): per-phase cycle sums of wave 0 of every workgroup
};

// Pass a lane-dependent value through an empty asm so that nothing derived from it is loop-invariant for the compiler: hoisted
// out of the tile loop such values sit in VGPRs across the K loop (96 weight + 64 accumulator registers) and get spilled, and a
// spill reload is a vector-memory operation whose wait also waits for every LDS-DMA in flight.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ int fdiv(int n, uint64_t m, int k) { return (int)(((uint64_t)(uint32_t)n * m) >> k); }

__device__ __forceinline__ u32x4 make_srd(const void* base, uint64_t bytes) {
  const uint64_t a = (uint64_t)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes > 0x7fffffffull ? 0x7fffffffu : (uint32_t)bytes);
  r[3] = 0x00020000u;
  return r;
}
// 64 lanes x 16 B from srd[voff] to LDS[lds_addr + 16 * lane]; out-of-range lanes write zeros.  Issued from asm so that the
// compiler does not order later LDS reads behind it (the kernel waits for its DMAs itself: counted vmcnt + barrier).
__device__ __forceinline__ void dma16(uint32_t lds_addr, uint32_t voff, const u32x4& srd) {
  uint32_t keep;
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voff), "s"(srd)
      : "memory");
}
// 16-byte store through a buffer descriptor: lanes with an out-of-range offset are dropped by the hardware, so every wave issues
// the instruction exactly once per call (the counted vmcnt wait at the top of the tile loop depends on that).  A compiler builtin,
// not asm: the compiler then counts these stores in the vmcnt waits it inserts for the epilogue's loads (an asm store is
// invisible to it, and every later load wait would become vmcnt(0), i.e. wait for the stores to drain).
typedef uint32_t v4u32 __attribute__((__vector_size__(16)));
__device__ __forceinline__ void buf_store16(const u32x4& v, uint32_t voff, __amdgpu_buffer_rsrc_t srd) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, v), srd, (int)voff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void* base, uint64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, (short)0, bytes > 0x7fffffffull ? 0x7fffffff : (int)bytes, 0x00020000);
}
// Workgroup barrier without __syncthreads()'s release fence (which makes hipcc wait for vmcnt(0), i.e. for the next tile's
// DMAs and the previous tile's stores): LDS traffic of this wave is complete (lgkmcnt), then the barrier.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4_t;
typedef __attribute__((address_space(3))) float lds_f32_t;
__device__ __forceinline__ u32x4 lds_ld16(uint32_t a) { return *reinterpret_cast<lds_u32x4_t*>(a); }
__device__ __forceinline__ void lds_st16(uint32_t a, const u32x4& v) { *reinterpret_cast<lds_u32x4_t*>(a) = v; }

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// Sum over the 16 lanes of a DPP row (= the 16 rows of one MFMA sub-tile), fixed butterfly: every lane gets the total.
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0x128>(v);  // row_ror:8
  v += dpp_f<0x124>(v);  // row_ror:4
  v += dpp_f<0x122>(v);  // row_ror:2
  v += dpp_f<0x121>(v);  // row_ror:1
  return v;
}

// MODE 0: stride 1 on the source itself; 1: stride 2; 2: stride 1 on a nearest-upsampled source (lvirt != lin).
#ifdef NARROW_STAMPS
#define STAMP(i)                                            \
  do {                                                      \
    const unsigned long long t__ = __builtin_amdgcn_s_memtime(); \
    ph[i] += t__ - t_last;                                  \
    t_last = t__;                                           \
  } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, bool NORM, bool STATS, bool RES>
__global__ __launch_bounds__(256, 2) void narrow_conv_kernel(const NarrowArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int S = MODE == 1 ? 2 : 1;
  constexpr int TR = kTileRows;              // output (padded) rows per tile: 4 waves = 4 channel quarters, each all 64 rows
  constexpr int WR = TR;                     // rows per wave (more rows per wave need more accumulators and offsets next to the 96
                                             // weight registers: hipcc then spills weight fragments, and every reload is a
                                             // vector-memory wait inside the K loop)
  constexpr int NI = WR / 16;
  constexpr int NST = NI + (STATS ? NI : 0);  // stores per wave per tile (always issued; masked lanes fall out of range)
  using R = Ring<MODE, RES>;
  constexpr int kWinBytes = R::WBYTES, kZeroOff = R::ZERO_OFF, NBUF = R::NBUF, D = R::D, NG = R::NG, NDMA = R::NDMA;
  static_assert((D - 1) * NDMA + D * NST < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t gb_base = lds_base + NBUF * R::STAGE;   // gamma | beta
  const uint32_t bt_base = gb_base + kGbBytes;           // bias + per-step row
  const uint32_t dummy_base = bt_base + 512;             // where DMAs that have nothing to fetch put their zeros

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave, wrh = 0;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- per-tile geometry (wave-uniform) -------------------------------------------------------------------------------
  struct Geo {
    int p0, base, W, b_first, n_first;
    bool any;
  };
  auto map_u = [&](int u) -> int {
    if constexpr (MODE == 2) return min((int)floorf((float)u * p.up_scale), p.lin - 1);
    else return u;
  };
  auto geom = [&](int ti) -> Geo {
    Geo g;
    g.p0 = ti * TR;
    const int ptot = p.B * p.Lp;
    const int pe = (g.p0 + TR < ptot ? g.p0 + TR : ptot) - 1;
    int b0 = fdiv(g.p0, p.div_m, p.div_k), l0 = g.p0 - b0 * p.Lp;
    if (l0 >= p.lout) {
      ++b0;
      l0 = 0;
    }
    int b1 = fdiv(pe, p.div_m, p.div_k), l1 = pe - b1 * p.Lp;
    if (l1 >= p.lout) l1 = p.lout - 1;
    g.any = ti < p.ntiles && b0 < p.B && (b0 < b1 || (b0 == b1 && l0 <= l1));
    const int u0 = l0 * S - 1 > 0 ? l0 * S - 1 : 0;
    const int u1 = l1 * S + 1 < p.lvirt - 1 ? l1 * S + 1 : p.lvirt - 1;
    const int s0 = b0 * p.lin + map_u(u0), s1 = b1 * p.lin + map_u(u1);
    g.base = g.any ? s0 : 0;
    g.W = g.any ? s1 - s0 + 1 : 0;
    g.b_first = g.any ? b0 : 0;
    g.n_first = g.any ? b0 * p.lout + l0 : 0;
    return g;
  };

  // ---- stage loader: every wave issues exactly NDMA DMA instructions per stage (the counted waits depend on it), ONE per call of
  //      issue_part<k> so that they can be spread over the K loop (a DMA costs 100-200 issue cycles; in a burst behind the barrier
  //      the matrix pipe would idle for all of them).  What has nothing to fetch (groups past the window, absent operands, a tile
  //      that does not exist) reads out-of-range zeros into a dummy KB. ---------------------------------------------------------
  auto issue_part = [&](const Geo& g, int buf, auto kc) {
    constexpr int K = decltype(kc)::value;
    const uint32_t stage = lds_base + buf * R::STAGE;
    const int lane_o = opaque(lane), r8o = lane_o >> 3;
    if constexpr (K < 2 * NG) {  // activation window: 8-row group wave + 8 i of channel block cb
      constexpr int i = K / 2, cb = K % 2;
      const u32x4 srd = make_srd(p.src + ((size_t)g.base * p.src_ld + p.ch_off) * 2, (uint64_t)((size_t)(p.src_rows - g.base) * p.src_ld - p.ch_off) * 2);
      const int ngroups = (g.W + 7) >> 3;   // 0 when the tile has no rows
      const uint32_t row_bytes = (uint32_t)p.src_ld * 2;
      const int grp = wave + 4 * i;
      const bool real = grp < ngroups;
      const uint32_t voff = real ? (uint32_t)(grp * 8 + r8o) * row_bytes + (uint32_t)(((lane_o & 7) ^ r8o) << 4) + cb * 128 : kOob;
      dma16(real ? stage + cb * kWinBytes + grp * 1024 : dummy_base, voff, srd);
    } else if constexpr (RES && K < 2 * NG + 4) {
      // residual rows n_first + [0, 64): 4 rows (1 KB) per DMA, 16-byte chunk c of row r lands at chunk c ^ (r & 15)
      constexpr int i = K - 2 * NG;
      const u32x4 srd = make_srd(p.res + (size_t)g.n_first * p.res_ld * 2, g.any ? (uint64_t)((size_t)((long)p.B * p.lout - g.n_first) * p.res_ld) * 2 : 0);
      const int row = (wave * 4 + i) * 4 + (lane_o >> 4);
      const uint32_t voff = (uint32_t)row * (uint32_t)(p.res_ld * 2) + (uint32_t)(((lane_o & 15) ^ (row & 15)) << 4);
      dma16(stage + R::RES_OFF + (wave * 4 + i) * 1024, voff, srd);
    } else {
      // waves 0, 1: embed rows of trajectories b_first + 2 w, + 2 w + 1 (512 B each); wave 2: {rstd, -mean * rstd} rows of
      // trajectories b_first + [0, 4) (256 B each); wave 3: nothing
      const bool eb = wave < 2 && p.ebias != nullptr && g.any;
      const bool mr = NORM && wave == 2 && g.any;
      const int t_eb = 2 * wave + (lane_o >> 5), t_mr = lane_o >> 4;
      const size_t row0 = (size_t)g.b_first;
      u32x4 srd;
      uint32_t voff = kOob, dst = dummy_base;
      if (eb) {
        srd = make_srd(p.ebias + row0 * p.ebias_ld + p.ebias_off, (uint64_t)((size_t)(p.B - g.b_first) * p.ebias_ld - p.ebias_off) * 4);
        voff = (uint32_t)t_eb * (uint32_t)(p.ebias_ld * 4) + (uint32_t)((lane_o & 31) << 4);
        dst = stage + R::EB_OFF + wave * 1024;
      } else if (mr) {
        srd = make_srd(p.nmr + (row0 * p.nmr_ld + p.nmr_off) * 2, (uint64_t)((size_t)(p.B - g.b_first) * p.nmr_ld - p.nmr_off) * 8);
        voff = (uint32_t)t_mr * (uint32_t)(p.nmr_ld * 8) + (uint32_t)((lane_o & 15) << 4);
        dst = stage + R::MR_OFF;
      } else {
        srd = make_srd(p.w, 0);
      }
      dma16(dst, voff, srd);
    }
  };
  auto issue_stage = [&](const Geo& g, int buf) {   // the whole stage at once (prologue)
    [&]<int... Ks>(std::integer_sequence<int, Ks...>) { (issue_part(g, buf, std::integral_constant<int, Ks>{}), ...); }(std::make_integer_sequence<int, NDMA>{});
  };

  // ---- weights: this wave's A fragments, in registers for the life of the kernel -----------------------------------------
  // A row m of sub-tile mi <-> output channel wq*32 + (m >> 2)*8 + mi*4 + (m & 3): after the MFMA, lane (fq, fr) holds, for its
  // row fr, channels wq*32 + fq*8 + {0..3} (mi = 0) and + {4..7} (mi = 1): 8 consecutive channels.
  u32x4 afrag[2][6][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int co = wq * 32 + (fr >> 2) * 8 + mi * 4 + (fr & 3);
    const char* wrow = p.w + (size_t)co * 384 * 2;
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) afrag[mi][s][kk] = *reinterpret_cast<const u32x4*>(wrow + (64 * s + (kk * 4 + fq) * 8) * 2);
  }
  const int ch8 = wq * 32 + fq * 8;  // first of this lane's 8 output channels
  // loop-invariant rows go to LDS once and are re-read per tile (registers that must not live across the K loop):
  // bias + per-step embed row, and gamma | beta of the input GroupNorm
  if (tid < 128) {
    float t = 0.f;
    if (p.bias) t += p.bias[tid];
    if (p.tbias) t += p.tbias[p.ebias_off + tid];
    *reinterpret_cast<lds_f32_t*>(bt_base + tid * 4) = t;
  }
  if constexpr (NORM) {
    if (tid < 256) *reinterpret_cast<lds_f32_t*>(gb_base + tid * 4) = tid < 128 ? p.ngamma[tid] : p.nbeta[tid - 128];
  }
  // zero rows of every stage's two windows (never touched by the DMA or the transform)
  if (tid < NBUF * 16) lds_st16(lds_base + (tid >> 4) * R::STAGE + ((tid >> 3) & 1) * kWinBytes + kZeroOff + (tid & 7) * 16, (u32x4){0u, 0u, 0u, 0u});

  // ---- GroupNorm-on-load roles: thread -> (channel block, physical chunk, row mod 16) of the window; its LOGICAL chunk (= 8
  //      input channels, hence its gamma / beta and its two groups) is fixed -----------------------------------------------------------
  const int t_cb = (tid >> 3) & 1, t_row0 = tid >> 4;
  const int t_c8 = t_cb * 8 + ((tid & 7) ^ (t_row0 & 7));
  const int t_g0 = (t_c8 * 8) >> (NORM ? p.ncpg_shift : 2), t_g1 = (t_c8 * 8 + 4) >> (NORM ? p.ncpg_shift : 2);
  typedef __attribute__((address_space(3))) float2 lds_f2_t;

  const __amdgpu_buffer_rsrc_t srd_stats = make_rsrc(STATS ? (void*)p.sout : (void*)p.out, STATS ? (uint64_t)p.B * p.nblk_out * p.sout_ld * 8 : 0);
  const int ti0 = blockIdx.x, tstep = gridDim.x;
  if (ti0 >= p.ntiles) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight fragments have arrived: from here on only DMAs and stores are in flight
#pragma unroll
  for (int k = 0; k < D; ++k) issue_stage(geom(ti0 + k * tstep), k);

  u32x4 pend[NI];
  uint32_t pend_off[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    pend[k] = (u32x4){0u, 0u, 0u, 0u};
    pend_off[k] = kOob;   // the first tile's K loop has nothing to store: its NI store slots fall out of range
  }
  __amdgpu_buffer_rsrc_t srd_prev = make_rsrc(p.out, 0);
#ifdef NARROW_STAMPS
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
#endif
  for (int it = 0;; ++it) {
    const int ti = ti0 + it * tstep;
    const int buf = it % NBUF;
    const Geo cur = geom(ti);
    const bool more = ti + tstep < p.ntiles;
    // Stage `it` has landed.  Younger than its DMAs are: the DMAs of the next D - 1 stages and the stores of the tiles finished
    // since it was requested (min(it, D) tiles, NST stores each) - everything else this wave ever issued is older.
    {
      const int done = it < D ? it : D;
      if (done == 0) wait_vm<(D - 1) * NDMA>();
      else if (done == 1) wait_vm<(D - 1) * NDMA + NST>();
      else if (done == 2) wait_vm<(D - 1) * NDMA + (D >= 2 ? 2 : 1) * NST>();
      else wait_vm<(D - 1) * NDMA + (D >= 3 ? 3 : 1) * NST>();
    }
    STAMP(0);
    wg_barrier();   // ... for every wave; and every wave is done with the stage that is requested next (tile it - 1's)
    STAMP(1);
    const Geo far = geom(ti + D * tstep);   // requested during the K loop below, into the stage tile it - 1 has just released
    const int far_buf = (it + D) % NBUF;
    STAMP(2);
    const uint32_t stage = lds_base + buf * R::STAGE;
    if constexpr (NORM) {
      const uint32_t wbase = stage + t_cb * kWinBytes + (tid & 7) * 16;
      const uint32_t mrb = stage + R::MR_OFF;
      float gam[8], bet[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 gv = *reinterpret_cast<lds_f32x4_t*>(gb_base + (t_c8 * 8 + 4 * h) * 4), bv = *reinterpret_cast<lds_f32x4_t*>(gb_base + (128 + t_c8 * 8 + 4 * h) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gam[4 * h + j] = gv[j];
          bet[4 * h + j] = bv[j];
        }
      }
      // every read of this thread's (up to 5) chunks and of their {rstd, -mean * rstd} pairs first, then arithmetic + write back
      // chunk by chunk: no LDS round trip is waited for with nothing else to do
      constexpr int NRD = MODE == 0 ? (TR + 2 + 15) / 16 : (R::WROWS + 15) / 16;   // the host takes NORM in MODE 0 only: <= TR + 2 window rows
      u32x4 raw[NRD];
      float2 m0[NRD], m1[NRD];
#pragma unroll
      for (int k = 0; k < NRD; ++k) {
        const int row = t_row0 + 16 * k < cur.W ? t_row0 + 16 * k : 0;   // (rows past the window: read row 0, never written back)
        raw[k] = lds_ld16(wbase + row * 128);
        int tb = fdiv(cur.base + row, p.lin_m, p.lin_k) - cur.b_first;
        tb = tb < kMaxTraj ? tb : kMaxTraj - 1;
        m0[k] = *reinterpret_cast<lds_f2_t*>(mrb + (tb * 32 + t_g0) * 8);
        m1[k] = *reinterpret_cast<lds_f2_t*>(mrb + (tb * 32 + t_g1) * 8);
      }
      auto xform = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int k = 0; k < NRD; ++k) {
          float x[8];
          Elem<__bf16>::unpack(raw[k], x);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float y = fmaf(fmaf(x[j], j < 4 ? m0[k].x : m1[k].x, j < 4 ? m0[k].y : m1[k].y), gam[j], bet[j]);
            x[j] = ACT == TDX_ACT_SWISH ? tdx_swish(y) : (ACT == TDX_ACT_LEAKY ? tdx_leaky(y) : y);
          }
          if (t_row0 + 16 * k < cur.W) lds_st16(wbase + (t_row0 + 16 * k) * 128, Elem<__bf16>::pack(x));
        }
      };
      if (p.nact == TDX_ACT_SWISH) xform(std::integral_constant<int, TDX_ACT_SWISH>{});
      else if (p.nact == TDX_ACT_LEAKY) xform(std::integral_constant<int, TDX_ACT_LEAKY>{});
      else xform(std::integral_constant<int, TDX_ACT_NONE>{});
      STAMP(3);
      wg_barrier();
    }
    STAMP(4);

    // ---- fragment offsets of this lane's rows: window byte offset of (sub-tile ni, tap t), or the zero row ---------------
    uint32_t woff2[NI / 2][3];   // two 16-bit window offsets per register (sub-tiles 2k and 2k + 1)
    const int fr_o = opaque(fr), fq_o = opaque(fq);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int pp = cur.p0 + wrh * WR + ni * 16 + fr_o;
      const int b = fdiv(pp, p.div_m, p.div_k), l = pp - b * p.Lp;
      const bool row_ok = cur.any && b < p.B && l < p.lout;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int u = l * S + t - 1;
        const bool ok = row_ok && u >= 0 && u < p.lvirt;
        const int wrow = b * p.lin + map_u(ok ? u : 0) - cur.base;
        const uint32_t wo = ok ? (uint32_t)(wrow * 128 + ((fq_o ^ (wrow & 7)) << 4)) : (uint32_t)kZeroOff;
        woff2[ni / 2][t] = (ni & 1) ? (woff2[ni / 2][t] | (wo << 16)) : wo;
      }
    }

    // ---- K loop: 6 K-steps x 2 halves, LDS fragment reads + MFMA only -------------------------------------------------------
    f32x4 acc[2][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[0][ni] = acc[1][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Fragment reads are software-pipelined by hand, one batch (4 sub-tiles = 8 MFMAs) ahead, and pinned with sched_barrier:
    // left alone hipcc issues one ds_read, waits lgkmcnt(0), issues its two MFMAs, and exposes an LDS round trip per pair.
    constexpr int HB = NI >= 4 ? 4 : NI;      // sub-tiles per batch
    constexpr int NB = 12 * (NI / HB);        // batches per tile: (K-step, K half, sub-tile group)
    u32x4 bq[2][HB];
    auto rd = [&](int q, u32x4 (&dst)[HB]) {
      const int s = q / (2 * (NI / HB)), kk = (q / (NI / HB)) & 1, h = q % (NI / HB);
      const uint32_t wb = stage + (s / 3) * kWinBytes;
#pragma unroll
      for (int i = 0; i < HB; ++i) {
        const int ni = h * HB + i;
        const uint32_t wo = (ni & 1) ? (woff2[ni / 2][s % 3] >> 16) : (woff2[ni / 2][s % 3] & 0xffffu);
        dst[i] = lds_ld16(wb + (wo ^ (kk << 6)));
      }
    };
    rd(0, bq[0]);
    // Between the MFMA batches: first the NDMA DMA instructions of the stage requested D tiles ahead, then the NI deferred output
    // stores of the PREVIOUS tile (issued in a burst at the end of its epilogue they block every wave of the workgroup for the
    // ~3 000 cycles the CU's store path needs for 32 KB; here they drain under the matrix work).  All DMAs precede all stores:
    // the counted wait at the top of the loop relies on that order.
    static_assert(NDMA <= NB, "one DMA slot per batch");
    auto batch = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int s = q / (2 * (NI / HB)), kk = (q / (NI / HB)) & 1, h = q % (NI / HB);
      if constexpr (q + 1 < NB) rd(q + 1, bq[(q + 1) & 1]);
      if constexpr (q < NDMA) issue_part(far, far_buf, std::integral_constant<int, (q < NDMA ? q : 0)>{});
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if ((NDMA + k < NB ? NDMA + k : NB - 1) == q) buf_store16(pend[k], pend_off[k], srd_prev);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < HB; ++i) {
        const int ni = h * HB + i;
        acc[0][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afrag[0][s][kk]), __builtin_bit_cast(bf16x8, bq[q & 1][i]), acc[0][ni], 0, 0, 0);
        acc[1][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afrag[1][s][kk]), __builtin_bit_cast(bf16x8, bq[q & 1][i]), acc[1][ni], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... Qs>(std::integer_sequence<int, Qs...>) { (batch(std::integral_constant<int, Qs>{}), ...); }(std::make_integer_sequence<int, NB>{});

    STAMP(5);
    // ---- epilogue, straight from the accumulator layout; every operand is already in the LDS stage ----------------------------------
    const __amdgpu_buffer_rsrc_t srd_out = make_rsrc(p.out + ((size_t)cur.n_first * p.out_ld + p.out_off) * 2,
                                                     cur.any ? (uint64_t)((size_t)((long)p.B * p.lout - cur.n_first) * p.out_ld - p.out_off) * 2 : 0);
    const int fr_e = opaque(fr);
    const f32x4 t0 = *reinterpret_cast<lds_f32x4_t*>(bt_base + ch8 * 4), t1 = *reinterpret_cast<lds_f32x4_t*>(bt_base + ch8 * 4 + 16);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int pp_e = cur.p0 + wrh * WR + ni * 16 + fr_e;
      const int b_e = fdiv(pp_e, p.div_m, p.div_k), l_e = pp_e - b_e * p.Lp;
      const bool ok = cur.any && b_e < p.B && l_e < p.lout;
      const int b = ok ? b_e : cur.b_first;
      const int nrel = ok ? b * p.lout + l_e - cur.n_first : 0;   // row inside the tile's output range, [0, 64)
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[0][ni][j] + t0[j];
        v[4 + j] = acc[1][ni][j] + t1[j];
      }
      if (p.ebias) {
        const uint32_t ea = stage + R::EB_OFF + (b - cur.b_first) * 512 + ch8 * 4;
        const f32x4 e0 = *reinterpret_cast<lds_f32x4_t*>(ea), e1 = *reinterpret_cast<lds_f32x4_t*>(ea + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] += e0[j];
          v[4 + j] += e1[j];
        }
      }
      if (p.act == TDX_ACT_SWISH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tdx_swish(v[j]);
      } else if (p.act == TDX_ACT_LEAKY) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tdx_leaky(v[j]);
      }
      if constexpr (RES) {
        const u32x4 r = lds_ld16(stage + R::RES_OFF + nrel * 256 + ((((ch8 >> 3) ^ (nrel & 15))) << 4));
        float rv[8];
        Elem<__bf16>::unpack(r, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rv[j];
      }
      pend[ni] = Elem<__bf16>::pack(v);   // stored during the next tile's K loop
      pend_off[ni] = ok ? (uint32_t)(nrel * p.out_ld + ch8) * 2 : kOob;
      if constexpr (STATS) {
        // this sub-tile = rows 16 j .. 16 j + 15 of ONE trajectory: {sum, sum of squares} of the two 4-channel granules
        float s[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float a0 = ok ? v[4 * h] : 0.f, a1 = ok ? v[4 * h + 1] : 0.f, a2 = ok ? v[4 * h + 2] : 0.f, a3 = ok ? v[4 * h + 3] : 0.f;
          s[2 * h] = row16_sum((a0 + a1) + (a2 + a3));
          s[2 * h + 1] = row16_sum((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3));
        }
        // lane fr == 0 of each quarter writes its two granules (16 bytes); the sub-tile's trajectory / block from that lane's row
        const int pp0 = cur.p0 + wrh * WR + ni * 16;
        const int b0 = fdiv(pp0, p.div_m, p.div_k), l0 = pp0 - b0 * p.Lp;
        const bool w_ok = cur.any && fr == 0 && b0 < p.B && l0 < p.lout;
        const uint32_t so = (uint32_t)(((b0 * p.nblk_out + (l0 >> 4)) * p.sout_ld + p.sout_off + (ch8 >> 2)) * 8);
        buf_store16(__builtin_bit_cast(u32x4, (f32x4){s[0], s[1], s[2], s[3]}), w_ok ? so : kOob, srd_stats);
      }
    }
    srd_prev = srd_out;
    STAMP(6);
    if (!more) break;
  }
#pragma unroll
  for (int k = 0; k < NI; ++k) buf_store16(pend[k], pend_off[k], srd_prev);   // the last tile's rows
#ifdef NARROW_STAMPS
  if (p.dbg && lane == 0 && wave == 0)
    for (int i = 0; i < 8; ++i) p.dbg[blockIdx.x * 8 + i] = ph[i];
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (stages requested past the last tile fetched nothing, but they must not outlive the LDS)
#endif
}

template <int MODE, bool NORM, bool STATS, bool RES>
int launch_narrow(const NarrowArgs& a, hipStream_t st, int cus) {
  constexpr int kLds = lds_bytes<MODE, RES>();
  static bool attr_done_dev[kTdxMaxDevices] = {};   // (per device: hipFuncSetAttribute acts on the current device's copy of the kernel)
  bool& attr_done = attr_done_dev[tdx_device_slot()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&narrow_conv_kernel<MODE, NORM, STATS, RES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLds) != hipSuccess)
      TDX_FAIL(TDX_ERR_HIP, "narrow_conv: cannot reserve %d bytes of LDS", kLds);
    attr_done = true;
  }
  const int grid = a.ntiles < 2 * cus ? a.ntiles : 2 * cus;   // two resident workgroups per CU
  hipLaunchKernelGGL((narrow_conv_kernel<MODE, NORM, STATS, RES>), dim3((unsigned)grid), dim3(256), kLds, st, a);
  TDX_CHECK_LAUNCH("narrow_conv_kernel");
  return TDX_OK;
}

template <int MODE>
int dispatch_narrow(const NarrowArgs& a, hipStream_t st, int cus) {
  const bool norm = a.nmr != nullptr, stats = a.sout != nullptr, res = a.res != nullptr;
  if constexpr (MODE == 0) {
    if (norm && res) return stats ? launch_narrow<0, true, true, true>(a, st, cus) : launch_narrow<0, true, false, true>(a, st, cus);
    if (norm) return stats ? launch_narrow<0, true, true, false>(a, st, cus) : launch_narrow<0, true, false, false>(a, st, cus);
  }
  if constexpr (MODE != 1) {   // (a stride-2 window plus a residual tile does not fit two ring stages: the host refuses the pair)
    if (res) return stats ? launch_narrow<MODE, false, true, true>(a, st, cus) : launch_narrow<MODE, false, false, true>(a, st, cus);
  }
  return stats ? launch_narrow<MODE, false, true, false>(a, st, cus) : launch_narrow<MODE, false, false, false>(a, st, cus);
}

void magic(int d, uint64_t& m, int& k) {
  int c = 0;
  while ((1ll << c) < d) ++c;
  k = 32 + c;
  m = (uint64_t)((1ull << k) / (uint64_t)d) + 1;
}

int g_narrow_off = 0;                        // test hook: 1 = never take this kernel (tdx_conv_gemm_force_small bit 16, mask 0x10000)
unsigned long long* g_narrow_dbg = nullptr;  // diagnostic builds (-DNARROW_STAMPS): per-phase cycle sums, see tdx_narrow_conv_stamps

}  // namespace

void tdx_narrow_conv_disable(int off) { g_narrow_off = off; }
extern "C" void tdx_narrow_conv_stamps(unsigned long long* dev_buf) { g_narrow_dbg = dev_buf; }   // diagnostic builds (-DNARROW_STAMPS)

// Can (and should) this descriptor run on the narrow kernel?  `why` receives the reason when it cannot.
static bool narrow_plan(const tdx_conv_desc* d, NarrowArgs& a, const char** why) {
  *why = "";
  const bool wants = d->norm[0].mr || d->norm[1].mr || d->norm[2].mr || d->norm[3].mr || d->stats_out;
  auto no = [&](const char* w) {
    *why = w;
    return false;
  };
  if (g_narrow_off && !wants) return no("disabled");
  if (d->dtype != TDX_BF16) return no("bf16 only");
  if (d->nseg != 1) return no("one segment only");
  const tdx_kseg& s = d->seg[0];
  if (s.cin != 128 || s.taps != 3 || d->cout != 128) return no("128 -> 128 channels, k = 3 only");
  if (d->gn_gamma || d->ws) return no("no fused output GroupNorm / split-K here");
  if (s.src_ld % 8 || s.ch_off % 8 || (uintptr_t)s.src % 16 || (uintptr_t)d->w % 16) return no("16-byte alignment of src / w");
  if (!d->out || d->out_ld % 8 || d->out_off % 8 || (uintptr_t)d->out % 16) return no("16-byte alignment of out");
  if (d->res && (d->res_ld % 8 || (uintptr_t)d->res % 16)) return no("16-byte alignment of res");
  if ((d->ebias || d->tbias) && ((d->ebias_off % 4) || (d->ebias_ld % 4))) return no("alignment of ebias");
  int mode;
  if (s.stride == 2 && s.lvirt == s.lin) mode = 1;
  else if (s.stride == 1 && s.lvirt == s.lin) mode = 0;
  else if (s.stride == 1 && s.lvirt > s.lin) mode = 2;
  else return no("stride / resampling combination");
  const int tr = kTileRows;
  const int lp = (d->lout + 15) / 16 * 16;
  if ((int64_t)d->B * lp + tr >= (1ll << 31) || (int64_t)d->B * s.lin >= (1ll << 31)) return no("row count overflows int32");
  // a tile's window: <= 258 consecutive source rows touching <= 16 trajectories.  Stride 1: lin <= Lp rows of source per Lp padded
  // rows; stride 2: lin <= 2 lout <= 2 Lp, 128 padded rows per tile; upsampled: fewer source rows than output rows.
  const int traj_per_tile = tr / lp + 2;
  if (mode == 1 && s.lin > 2 * d->lout) return no("stride-2 source longer than 2 lout");
  if (mode == 2 && s.lin > d->lout) return no("resampled source longer than the output");
  if (traj_per_tile > kMaxTraj || s.lin < 16) return no("trajectories too short for the tile's row tables");
  if (d->res && d->res_ld * 2 * 128 >= (1 << 30)) return no("res_ld too large");
  if (d->res && mode == 1) return no("no residual on stride-2 convs");
  if (lp < 32) return no("trajectories shorter than 17 rows");
  if ((int64_t)264 * s.src_ld * 2 >= (1ll << 31) || (int64_t)264 * d->out_ld * 2 >= (1ll << 31)) return no("row pitch too large");
  const int64_t tiles = ((int64_t)d->B * lp + tr - 1) / tr;
  // Without norm[] / stats_out the general kernels are as fast (stride 1) or faster (resampled sources): only the stride-2 conv
  // (gathered taps there, one contiguous window here) comes here by itself
  if (!wants && (tiles < 256 || mode != 1)) return no("the general kernel is at least as fast for this descriptor");
  const tdx_seg_norm& nm = d->norm[0];
  if (d->norm[1].mr || d->norm[2].mr || d->norm[3].mr) return no("norm on segment 0 only");
  int cshift = 2;
  if (nm.mr) {
    if (!nm.gamma || !nm.beta) return no("input GroupNorm needs gamma and beta");
    if (nm.cpg < 4 || nm.cpg > 128 || (nm.cpg & (nm.cpg - 1))) return no("input GroupNorm: channels per group must be a power of two in [4, 128]");
    while ((1 << cshift) < nm.cpg) ++cshift;
    if (nm.mr_ld < 128 / nm.cpg || nm.mr_off < 0 || nm.mr_off + 128 / nm.cpg > nm.mr_ld) return no("input GroupNorm: mr_ld / mr_off");
    if (nm.act < TDX_ACT_NONE || nm.act > TDX_ACT_SWISH) return no("bad norm act");
    if (mode != 0) return no("input GroupNorm: stride-1, un-resampled sources only");
    if ((uintptr_t)nm.mr % 8) return no("alignment of mr");
  }
  if (d->stats_out) {
    if (d->stats_out_ld < 32 || d->stats_out_off < 0 || d->stats_out_off % 2 || d->stats_out_ld % 2 || d->stats_out_off + 32 > d->stats_out_ld || (uintptr_t)d->stats_out % 16)
      return no("stats_out layout");
    if ((int64_t)d->B * ((d->lout + 15) / 16) * d->stats_out_ld * 8 >= (1ll << 31)) return no("stats_out too large");
  }
  a = NarrowArgs{};
  a.src = (const char*)s.src;
  a.src_ld = s.src_ld;
  a.ch_off = s.ch_off;
  a.lin = s.lin;
  a.lvirt = s.lvirt;
  a.up_scale = s.up_scale;
  a.src_rows = (long)d->B * s.lin;
  a.w = (const char*)d->w;
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
  a.B = d->B;
  a.lout = d->lout;
  a.Lp = lp;
  a.ntiles = (int)tiles;
  magic(lp, a.div_m, a.div_k);
  magic(s.lin, a.lin_m, a.lin_k);
  a.nmr = nm.mr;
  a.ngamma = nm.gamma;
  a.nbeta = nm.beta;
  a.nact = nm.act;
  a.nmr_ld = nm.mr_ld;
  a.nmr_off = nm.mr_off;
  a.ncpg_shift = cshift;
  a.sout = d->stats_out;
  a.sout_ld = d->stats_out_ld;
  a.sout_off = d->stats_out_off;
  a.nblk_out = (d->lout + 15) / 16;
  a.dbg = g_narrow_dbg;
  return true;
}

// Called by tdx_conv_gemm after its own validation.  *handled = 1 when the conv was launched here.
int tdx_narrow_conv_try(const tdx_conv_desc* d, hipStream_t st, int cus, int* handled) {
  NarrowArgs a;
  const char* why;
  *handled = 0;
  if (!narrow_plan(d, a, &why)) {
    if (d->norm[0].mr || d->norm[1].mr || d->norm[2].mr || d->norm[3].mr || d->stats_out)
      TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_conv_gemm: norm[] / stats_out are not supported for this shape (%s): ask tdx_conv_gemm_norm_fusable()", why);
    return TDX_OK;
  }
  *handled = 1;
  const tdx_kseg& s = d->seg[0];
  if (s.stride == 2) return dispatch_narrow<1>(a, st, cus);
  if (s.lvirt != s.lin) return dispatch_narrow<2>(a, st, cus);
  return dispatch_narrow<0>(a, st, cus);
}

bool tdx_narrow_conv_ok(const tdx_conv_desc* d) {
  NarrowArgs a;
  const char* why;
  return narrow_plan(d, a, &why);
}
