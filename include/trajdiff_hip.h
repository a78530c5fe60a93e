/*
 * trajdiff_hip.h — C ABI of libtrajdiff_hip.so: the MI355X (gfx950) kernels behind the
 * TrajectoryDiffusion denoising hot path (TrajUNet forward + DiffusionManager steps).
 *
 * The reference (JinmingWang/TrajectoryDiffusion) has no FFI: its hot path is Python calling
 * PyTorch ATen ops.  Every entry point below therefore cites the reference *call site* whose
 * ATen work it replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless named host_*;
 *   - `stream` is a hipStream_t passed as void*; all entry points are asynchronous and
 *     stream-ordered, allocate nothing, free nothing and never synchronise;
 *   - return 0 on success, a TDX_ERR_* code otherwise (tdx_last_error() gives the text);
 *     arguments are validated on the host BEFORE any launch;
 *   - activations are channels-last rows: tensor (B, L, C) is B*L rows of C contiguous
 *     elements of `dtype` (TDX_F32 or TDX_BF16); model inputs/outputs stay (B, 2, L) fp32
 *     exactly as the reference's forward() takes/returns them.
 */
#ifndef TRAJDIFF_HIP_H
#define TRAJDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TDX_OK = 0, TDX_ERR_ARG = 1, TDX_ERR_HIP = 2, TDX_ERR_UNSUPPORTED = 3 };
enum { TDX_F32 = 0, TDX_BF16 = 1 };
enum { TDX_ACT_NONE = 0, TDX_ACT_LEAKY = 1, TDX_ACT_SWISH = 2 };

/* Library / error state ---------------------------------------------------------------- */
const char* tdx_last_error(void);   /* thread-local text of the last failure */
int tdx_version(void);
/* Launch geometry used for the last tdx_conv_gemm call of this thread (for bench/roofline). */
int tdx_device_ok(void);            /* 0 iff the current device is gfx950 */

/* One K-segment of the implicit GEMM: a (source tensor, taps) pair.  Output row n = (b, l)
 * consumes, for tap t, source row  b*lin + map(l*stride + t - (taps==3))  or zeros when that
 * virtual index u falls outside [0, lvirt) — i.e. Conv1d's zero padding.  map is the identity when
 * lvirt == lin; otherwise it is F.interpolate(mode='nearest', size=lvirt)'s source index, computed
 * exactly as ATen does: min((int)floorf(u * up_scale), lin - 1) with up_scale = (float)lin / lvirt. */
typedef struct {
  const void* src;      /* (B, lin, src_ld) activations */
  float up_scale;       /* (float)lin / (float)lvirt when lvirt != lin, else ignored */
  int32_t reserved0;
  int32_t src_ld;       /* elements per source row */
  int32_t ch_off;       /* first source channel consumed */
  int32_t cin;          /* channels consumed; multiple of 64 (bf16) / 32 (fp32) */
  int32_t taps;         /* 1 or 3 */
  int32_t stride;       /* 1 or 2 */
  int32_t lin;          /* source trajectory length */
  int32_t lvirt;        /* virtual (post-upsample, pre-stride) length */
  int32_t reserved;
} tdx_kseg;

/* GroupNorm(+act) applied to ONE K-segment while tdx_conv_gemm loads it; see tdx_conv_desc.norm. */
typedef struct {
  const float* mr;      /* (B, mr_ld, 2) fp32 {rstd, -mean * rstd} per (trajectory, group), from tdx_groupnorm_finalize; or NULL */
  const float* gamma;   /* [cin]: affine parameters of the segment's channels */
  const float* beta;
  int32_t cpg;          /* channels per group: a power of two >= 4 */
  int32_t act;          /* TDX_ACT_* applied after the affine */
  int32_t mr_ld;        /* groups per trajectory row of `mr` */
  int32_t mr_off;       /* group that holds the segment's channel 0 */
} tdx_seg_norm;

/* Conv1d (k=3 / k=1, stride 1/2, optional nearest-upsampled input, optional channel-concat of
 * several sources, optional fused 1x1 shortcut as an extra segment) as ONE gathered GEMM on MFMA:
 *   out[n, co] = act( sum_seg sum_t sum_ci  w[co, k(seg,t,ci)] * src_seg[row(n,t), ch_off+ci]
 *                     + bias[co] + ebias[b(n), co] + tbias[co] ) + res[n, co]
 * Replaces: nn.Conv1d call sites Models/TrajUNet.py:52,61 and Models/TrajUNetBlocks.py:77,86,89,
 * 110,112, the embed_proj add (:98-99), the shortcut add (:99), torch.cat (TrajUNet.py:90) and
 * F.interpolate (TrajUNet.py:93); nn.Linear call sites TrajUNetBlocks.py:20-40 with lout = 1. */
typedef struct {
  tdx_kseg seg[4];
  int32_t nseg;
  int32_t dtype;        /* TDX_F32 / TDX_BF16: element type of src, w, res, out */
  const void* w;        /* packed weights [cout][ktot], ktot = sum_seg taps*cin, K order = seg, tap, ci */
  const float* bias;    /* [cout] or NULL */
  const float* ebias;   /* per-sample bias rows (B, ebias_ld), or NULL */
  const float* tbias;   /* per-call bias vector (indexed like one ebias row), or NULL */
  int32_t ebias_ld;
  int32_t ebias_off;    /* column of ebias/tbias that maps to co = 0 */
  const void* res;      /* residual (B*lout, res_ld) added after act, or NULL */
  int32_t res_ld;
  int32_t act;          /* TDX_ACT_* applied before the residual add */
  void* out;            /* (B*lout, out_ld) */
  int32_t out_ld;
  int32_t out_off;      /* first output channel written (concat writes) */
  int32_t B, lout, cout;
  int32_t reserved;
  /* Optional fused GroupNorm(+act) of the CONSUMER, applied to the finished rows r = out[n, :] (fp32, before
   * rounding):  gn_out[n, gn_off + co] = gn_act((r - mean[b, g]) * rsqrt(var[b, g] + gn_eps) * gn_gamma[co] + gn_beta[co])
   * with groups of gn_cpg consecutive output channels over the trajectory's lout rows (biased variance, exact
   * two-pass, bitwise reproducible).  `out` may be NULL when only the normalised tensor is wanted.
   * Replaces the nn.GroupNorm + Swish that FOLLOWS a conv (Models/TrajUNetBlocks.py:75-76, :83-84, :109;
   * Models/TrajUNet.py:43).  Only some shapes can be fused: ask tdx_conv_gemm_gn_fusable() first. */
  const float* gn_gamma; /* NULL = no fused GroupNorm */
  const float* gn_beta;
  void* gn_out;
  int32_t gn_ld, gn_off, gn_act, gn_cpg;
  float gn_eps;
  int32_t reserved2;
  /* Optional split-K workspace (device, fp32): when given, a problem too small to fill the chip (few output tiles, long K:
   * the single-trajectory / small-batch serving case) is cut along K into up to ws_bytes / (4 * B * lout * cout) slices that
   * run as concurrent workgroups; a second kernel adds the slices in order and applies the epilogue.  Results are
   * reproducible, but the fp32 summation order differs from the unsplit kernels (and so may the last bit).  NULL = never. */
  void* ws;
  int64_t ws_bytes;
  /* Optional GroupNorm(+act) of the INPUT of segment i, applied while the segment is loaded (norm[i].mr != NULL):
   *   x'[b, l, c] = norm_act((x[b, l, c] - mean[b, g]) * rstd[b, g] * gamma[c] + beta[c]),  g = c / cpg (c counted from the
   * segment's channel 0 inside the normalised tensor, i.e. gamma/beta/stats already point at that channel),
   * with rstd and -mean * rstd read from norm[i].mr (tdx_groupnorm_finalize turns the block partial sums that the PRODUCER of
   * x wrote through its stats_out into that table).  The conv's zero
   * padding applies to x' (as in the reference, where the padding pads the normalised tensor).  This replaces the
   * nn.GroupNorm + Swish that PRECEDES a conv (Models/TrajUNetBlocks.py:75-77, :83-86) without a normalised copy in memory.
   * Only some shapes support it: ask tdx_conv_gemm_norm_fusable() first. */
  tdx_seg_norm norm[4];
  /* Optional statistics of the OUTPUT for a later consumer's norm[]: for every trajectory b, 16-row block j = l / 16 and
   * 4-channel granule q = co / 4, stats_out[((b * ceil(lout / 16) + j) * stats_out_ld + stats_out_off + q) * 2 + {0, 1}] =
   * {sum, sum of squares} of the finished fp32 values out[b, 16 j .. 16 j + 15, 4 q .. 4 q + 3] (fixed summation order per
   * block: bitwise reproducible and independent of the batch size).  NULL = none. */
  float* stats_out;
  int32_t stats_out_ld, stats_out_off;
} tdx_conv_desc;
int tdx_conv_gemm(const tdx_conv_desc* host_desc, void* stream);
/* > 0 if tdx_conv_gemm can run this descriptor WITH its fused GroupNorm, 0 if the caller must run
 * tdx_groupnorm_act separately (trajectories longer than a 256-row tile, odd channel counts, ...).
 * 2 = the persistent kernel takes it (bf16, stride-1 un-resampled sources, groups of 8 .. 64 channels, cout <= 1024): the form that
 * is faster than conv + stand-alone GroupNorm when tiles of whole trajectories fill their 256 rows; 1 = the one-tile kernel. */
int tdx_conv_gemm_gn_fusable(const tdx_conv_desc* host_desc);
/* 1 if tdx_conv_gemm can run this descriptor WITH its norm[] / stats_out fields (today: bf16, one 128-channel k=3 segment,
 * 128 output channels, enough rows to fill the chip: the HBM-bound full-resolution layers, where a separate GroupNorm pass
 * costs as much as the conv), 0 if the caller must run tdx_groupnorm_act separately and leave those fields empty. */
int tdx_conv_gemm_norm_fusable(const tdx_conv_desc* host_desc);

/* GroupNorm(groups, C, eps, affine) [+ Swish] over channels-last rows, one pass to memory:
 *   out[b, l, out_off + c] = act((x[b,l,c] - mean[b,g]) * rstd[b,g] * gamma[c] + beta[c])
 * Replaces nn.GroupNorm + Swish at Models/TrajUNetBlocks.py:75-76, :83-84, :109,
 * Models/TrajUNet.py:43 and Models/ModelUtils.py:14.  C/groups must be a power of two >= 4. */
int tdx_groupnorm_act(const void* x, int32_t x_ld, void* out, int32_t out_ld, int32_t out_off,
                      const float* gamma, const float* beta, int32_t B, int32_t L, int32_t C,
                      int32_t groups, float eps, int32_t act, int32_t dtype, void* stream);

/* Second half of the split GroupNorm: block partial sums (tdx_conv_desc.stats_out, tdx_stem_conv_stats) -> per (trajectory, group)
 *   mr[b, mr_off + g] = {rstd, -mean * rstd},  mean / biased variance over the L rows x cpg channels of group g
 * (g < groups; group g = granules [g * cpg / 4, (g + 1) * cpg / 4) of stats starting at granule stats_off; blocks and granules
 * are added in a fixed order, in double).  stats is (B, ceil(L / 16), stats_ld, 2) fp32, mr is (B, mr_ld, 2) fp32.
 * Replaces the statistics half of nn.GroupNorm (Models/TrajUNetBlocks.py:75,83; Models/TrajUNet.py:43). */
int tdx_groupnorm_finalize(const float* stats, int32_t stats_ld, int32_t stats_off, float* mr, int32_t mr_ld, int32_t mr_off,
                           int32_t B, int32_t L, int32_t groups, int32_t cpg, float eps, void* stream);

/* Stem Conv1d(2 -> C0, k3, p1): x (B, 2, L) fp32 channels-first -> out (B, L, C0) rows.
 * w is the reference layout (C0, 2, 3) fp32.  Replaces Models/TrajUNet.py:21,107. */
int tdx_stem_conv(const float* x, const float* w, const float* bias, void* out, int32_t B, int32_t L,
                  int32_t C0, int32_t dtype, void* stream);

/* tdx_stem_conv (bf16 output) that also writes the block partial sums of its output, in the layout of tdx_conv_desc.stats_out,
 * for the GroupNorm that follows the stem (Models/TrajUNetBlocks.py:75 of down_blocks.0.0).  C0 % 32 == 0. */
int tdx_stem_conv_stats(const float* x, const float* w, const float* bias, void* out, float* stats, int32_t stats_ld,
                        int32_t stats_off, int32_t B, int32_t L, int32_t C0, void* stream);

/* Head GroupNorm + Conv1d in one pass over the RAW rows x (B, L, C0) bf16: a = (x * rstd + (-mean * rstd)) * gamma + beta from
 * the {rstd, -mean * rstd} table of tdx_groupnorm_finalize, then the k3 conv to (B, 2, L) fp32.  Replaces Models/TrajUNet.py:42-45,112
 * without a normalised copy in memory.  C0 a power of two <= 512. */
int tdx_head_conv_norm(const void* x, const float* mr, int32_t mr_ld, int32_t mr_off, int32_t cpg, const float* gamma,
                       const float* beta, const float* w, const float* bias, float* out, int32_t B, int32_t L, int32_t C0,
                       void* stream);

/* Head Conv1d(C0 -> 2, k3, p1) on already group-normalised rows a (B, L, C0) -> eps (B, 2, L)
 * fp32 channels-first.  w is (2, C0, 3) fp32.  Replaces Models/TrajUNet.py:44,112. */
int tdx_head_conv(const void* a, const float* w, const float* bias, float* out, int32_t B, int32_t L,
                  int32_t C0, int32_t dtype, void* stream);

/* Mid-block single-head attention core on qkv rows (B, L, 3C) [q | k | v]:
 *   o[b] = softmax(q k^T / sqrt(C)) v   -> (B, L, C).   L <= 64.
 * Replaces transpose/split/bmm/softmax/bmm at Models/TrajUNetBlocks.py:121-123. */
int tdx_attention(const void* qkv, void* out, int32_t B, int32_t L, int32_t C, int32_t dtype, void* stream);

/* WideAndDeepEmbedBlock input gather (Models/TrajUNetBlocks.py:53-58): builds, all fp32,
 *   pe      (B, 128)   = pos_enc[time]
 *   catvec  (B, 1280)  = [cell_emb[cat0..3] | day_emb[cat4]]
 *   deep_in (B, 1024)  columns 768..1023 = leaky0.1(wide_fc(num))   (columns 0..767 are written by
 *                        the categorical_fc GEMM).  Indices are validated on the device: an
 *                        out-of-range index sets *err_flag (device int32) instead of faulting. */
int tdx_embed_gather(const int64_t* time, const int64_t* cat, const float* num, const float* pos_enc,
                     int32_t max_time, const float* cell_emb, const float* day_emb, const float* wide_w,
                     const float* wide_b, float* pe, float* catvec, float* deep_in, int32_t B,
                     int32_t* err_flag, void* stream);

/* DDPM posterior step, DiffusionManager.py:46-55, fused into one coalesced pass:
 *   x = (x - c_eps * eps) / sqrt_alpha + sigma * z     coef = {beta/sqrt(1-abar), sqrt(alpha), sqrt(beta)} (device, 3 floats)
 * z = injected noise tensor, or NULL to draw N(0,1) in-kernel (Philox4x32-10 + Box-Muller keyed by
 * (seed ^ dyn_seed, stream, first_elem + element index) so results do not depend on how the batch is sharded;
 * step_ctr is NULL or a device int32[4] = {step, dyn_seed_lo, dyn_seed_hi, stream}, read at run time so
 * a captured graph can be replayed with a new step / seed; stream = the diffusion step t in every sampling path, so the
 * eager per-step call and the graph-replayed loop draw the same noise).  sigma == 0 (t == 0) adds no noise.  In place on x. */
int tdx_ddpm_update(float* x, const float* eps, const float* z, const float* coef, int64_t n,
                    uint64_t seed, const int32_t* step_ctr, int64_t first_elem, void* stream);

/* DDIM (eta = 0) step on the same alpha-bar table (not in the reference):
 *   x0 = (x - s1m_t*eps)/sa_t ; x = sa_p*x0 + s1m_p*eps     coef = {s1m_t, 1/sa_t, sa_p, s1m_p}. */
int tdx_ddim_update(float* x, const float* eps, const float* coef, int64_t n, void* stream);

/* Sampling-loop bookkeeping kept on the device so one captured hipGraph can be replayed T times:
 * reads step = *step_ctr, copies coef_table[step, 0:ncoef] -> coef_cur and
 * tbias_table[step, 0:ntb] -> tbias_cur, fills t_vec[0:B] = step_to_t[step], sets step_ctr[3] = step_to_t[step] (the Philox
 * stream id tdx_ddpm_update reads; = step without a table), then *step_ctr = step + 1.
 * Replaces the host-side indexing in DiffusionManager.py:46-48,70-72. */
int tdx_step_prepare(int32_t* step_ctr, const float* coef_table, int32_t ncoef, float* coef_cur,
                     const float* tbias_table, int32_t ntb, float* tbias_cur, const int32_t* step_to_t,
                     int64_t* t_vec, int32_t B, int32_t nsteps, void* stream);

/* q-sample, DiffusionManager.py:33: x_t = sqrt_ab[t[b]] * x0 + sqrt_1m_ab[t[b]] * eps. */
int tdx_q_sample(const float* x0, const float* eps, const int64_t* t, const float* sqrt_ab,
                 const float* sqrt_1m_ab, int32_t T, float* out, int32_t B, int32_t per_sample,
                 int32_t* err_flag, void* stream);

/* DidiTrajectoryDataset.__getitem__ batched (Dataset/DidiDataset.py:138-174):
 *   lon_lat_raw (B, L, 2) fp32 degrees, minutes (B, L) fp32, stats = {mean_t, mean_lon, mean_lat,
 *   std_t, std_lon, std_lat}, bounds = 257 fp32 Gaussian-quantile cell boundaries.
 * Outputs lon_lat (B, 2, L) fp32 z-scored, cat (B, 5) int64, num (B, 3) fp32, times (B, L) fp32. */
int tdx_didi_items(const float* lon_lat_raw, const float* minutes, const float* stats, const float* bounds,
                   float* lon_lat, int64_t* cat, float* num, float* times, int32_t B, int32_t L, void* stream);
/* Inverse z-score (not in the reference): lon_lat (B, 2, L) -> degrees (B, L, 2). */
int tdx_didi_denorm(const float* lon_lat, const float* stats, float* out_deg, int32_t B, int32_t L, void* stream);
/* DidiTrajectoryDataset.getCellIndex (Dataset/DidiDataset.py:127-135) for n values: out[i] = torch.searchsorted(bounds, values[i]) - 1
 * over the 257 fp32 cell boundaries (the search tdx_didi_items runs on the normalised start / end points, :166-167). */
int tdx_cell_index(const float* values, const float* bounds, int64_t* out, int32_t n, void* stream);

/* ---- Training-step building blocks (train.ipynb:198-209).  fp32 arithmetic; every reduction has a
 * fixed order (two stages through a caller-provided workspace, no atomics).  trajectorydiffusion_amd/training.py assembles the step
 * (forward program, tape-recorded backward program, fused AdamW, data-parallel gradient mean) from these. ---- */

/* Floats a workspace `ws` must hold for tdx_mse_loss_grad and tdx_colsum. */
int64_t tdx_train_workspace_floats(void);

/* torch.nn.MSELoss()(pred, target) and its gradient (train.ipynb:205-206): *loss = mean((pred - target)^2),
 * dpred[i] = 2 (pred[i] - target[i]) / n (dpred may be null). */
int tdx_mse_loss_grad(const float* pred, const float* target, float* dpred, float* loss, int64_t n, float* ws, void* stream);

/* tdx_mse_loss_grad without a workspace argument and without any allocation.  No counterpart in the reference: it exists for
 * callers that cannot allocate or pass scratch memory (code under HIP graph capture, tight serving/training loops, harnesses
 * that forbid hipMalloc).  The first-stage partial sums live in a 4 KiB file-scope device array of the library (one copy per
 * GPU, held until the process ends); n <= 2^28 elements, larger inputs are rejected with TDX_ERR_ARG.  Same kernels and
 * launch geometry as tdx_mse_loss_grad: results are bit-identical to it.  Calls that overlap on ONE GPU (two streams, two
 * threads) share that GPU's array and race; serialise them.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress): make that call outside graph capture. */
int tdx_mse_loss_grad_static(const float* pred, const float* target, float* dpred, float* loss, int64_t n, void* stream);

/* out[c] (+)= sum over the n rows of x[r * ld + c], c < C <= 4096: bias gradients of Conv1d / Linear, and dgamma / dbeta from the
 * per-trajectory rows of tdx_groupnorm_act_bwd. */
int tdx_colsum(const void* x, int32_t ld, int32_t n, int32_t c, float* out, int32_t accumulate, int32_t dtype, float* ws, void* stream);
/* Two matrices of one shape in one launch pair (GroupNorm's dgamma and dbeta rows, tdx_groupnorm_act_bwd): outa (+)= column sums of xa,
 * outb (+)= column sums of xb; the same sums, in the same order, as two tdx_colsum calls. */
int tdx_colsum_pair(const void* xa, const void* xb, int32_t ld, int32_t n, int32_t c, float* outa, float* outb, int32_t accumulate, int32_t dtype,
                    float* ws, void* stream);

/* Weight gradient of one K-segment of tdx_conv_gemm (autograd of nn.Conv1d, TrajUNetBlocks.py:77,86,89):
 *   dw[co][ci_off + ci][t] (+)= sum over output rows n of dy[n * dy_ld + dy_off + co] * X[src(n, t)][ci],
 * dw in the reference layout (Cout, cin_total, taps) fp32, src(n, t) = the forward row map of `seg` (padding, stride 2,
 * nearest-upsampled source).  ws holds splits * cout * cin * taps floats (splits <= 512, chosen to fit ws_floats); the
 * partial sums are added in split order, so the result does not depend on scheduling. */
int tdx_conv_wgrad(const void* dy, int32_t dy_ld, int32_t dy_off, int32_t cout, const tdx_kseg* seg, int32_t B, int32_t lout,
                   float* dw, int32_t cin_total, int32_t ci_off, int32_t accumulate, int32_t dtype, float* ws, int64_t ws_floats,
                   void* stream);
/* The same, plus the bias gradient dbias[co] (+)= sum over all rows of dY[n][co] (nn.Conv1d's bias, TrajUNetBlocks.py:77,86,89) from the
 * same launch where the large-tile kernel runs (it sums the columns of the dY rows it stages anyway: no second 100 MB read of dY);
 * every other shape runs tdx_colsum first, through the same workspace (ws_floats >= tdx_train_workspace_floats() then). */
int tdx_conv_wgrad_bias(const void* dy, int32_t dy_ld, int32_t dy_off, int32_t cout, const tdx_kseg* seg, int32_t B, int32_t lout, float* dw,
                        int32_t cin_total, int32_t ci_off, int32_t accumulate, float* dbias, int32_t dbias_accumulate, int32_t dtype, float* ws,
                        int64_t ws_floats, void* stream);
/* ---- Evaluation metrics over sampled trajectories (eval_trajunet.ipynb cell 12; next-row widening) ----------
 * Workspace for the calls below: tdx_metric_workspace_bytes(number of histogram bins) bytes of device memory. */
int64_t tdx_metric_workspace_bytes(int32_t n_bins_total);
/* getPointDistribution / getStartDistribution / getEndDistribution (cell 12): traj (B, 2, L) fp32; the points
 * l_first <= l < l_first + l_count of every trajectory are counted on an n_grids x n_grids grid whose boundaries are
 * torch.linspace(min, max, n_grids) of the points' own lon / lat range, bin = torch.searchsorted(grid, value);
 * out[lon_bin * n_grids + lat_bin] = normalize ? (count + 1) / sum(count) : count. */
int tdx_point_distribution(const float* traj, int32_t B, int32_t L, int32_t l_first, int32_t l_count, int32_t n_grids,
                           int32_t normalize, float* out, void* ws, int64_t ws_bytes, void* stream);
/* getLengthDistribution (cell 12): path length sum_l |p[l+1] - p[l]| of every trajectory, binned with
 * torch.searchsorted(torch.linspace(lo, hi, n_bins), length); out = (count + 0.1) / sum(count).  lengths (B) is optional.
 * err[0] is set to 1 (never cleared) when a length exceeds hi: the reference raises IndexError there. */
int tdx_length_distribution(const float* traj, int32_t B, int32_t L, float lo, float hi, int32_t n_bins, float* out,
                            float* lengths, int32_t* err, void* ws, int64_t ws_bytes, void* stream);
/* JSD (cell 12): 0.5 * (kl_div(log P, M) + kl_div(log Q, M)), M = (P + Q) / 2, 'batchmean' over `batch` = P.shape[0]. */
int tdx_jsd(const float* P, const float* Q, int64_t n, int32_t batch, float* out, void* stream);

/* One re-layout of a conv weight into the K order tdx_conv_gemm reads (fp32): for row < rows, channel < cin, tap < taps
 *   dst[row * dst_ld + (channel / 32) * 32 * taps + tap * min(32, cin - 32 * (channel / 32)) + channel % 32]
 *     = tap_map[tap] < 0 ? 0 : src[row * sr + channel * sc + tap_map[tap] * st]        (strides in elements).
 * Forward weights (TrajUNetBlocks.py:77,86,89): rows = Cout, sr/sc/st = the parameter's strides, tap_map = {0,1,2};
 * data-gradient weights: rows = Cin, sr and sc exchanged, tap_map = {2,1,0}. */
typedef struct tdx_pack_desc {
  const void* src;
  void* dst;
  int32_t rows, cin, taps, dst_ld;
  int64_t sr, sc, st;
  int32_t tap_map[3];
  int32_t reserved;
} tdx_pack_desc;
/* Runs every entry of a DEVICE-resident table of n descriptors in one launch (the training step refreshes ~200 weight
 * copies per step; one by one they are launch-bound).  No reference counterpart: torch keeps one layout. */
int tdx_pack_weights(const tdx_pack_desc* table_dev, int32_t n, void* stream);


/* Backward of tdx_groupnorm_act (nn.GroupNorm + Swish, TrajUNetBlocks.py:75-76,83-84): from the forward INPUT x and the
 * gradient dy of the activated output, dx (+)= gradient of x; dgamma_rows / dbeta_rows (B, C) fp32 receive each
 * trajectory's share of the parameter gradients (sum them with tdx_colsum). */
int tdx_groupnorm_act_bwd(const void* x, int32_t x_ld, const void* dy, int32_t dy_ld, int32_t dy_off, void* dx, int32_t dx_ld,
                          int32_t dx_off, const float* gamma, const float* beta, float* dgamma_rows, float* dbeta_rows, int32_t B,
                          int32_t L, int32_t C, int32_t groups, float eps, int32_t act, int32_t accumulate, int32_t dtype,
                          void* stream);
/* nn.GroupNorm + Swish + nn.Dropout(p) in training mode as one pass each way (ResnetBlock.stage2, TrajUNetBlocks.py:83-85): the forward
 * writes dropout(act(groupnorm(x))) with tdx_dropout's mask for the plain (B, L, C) fp32 output (out_ld == C, out_off == 0; the mask is
 * keyed by the element index), the backward takes the gradient dy of that output (dy_ld == C, dy_off == 0), regenerates the mask and
 * otherwise is tdx_groupnorm_act_bwd.  Same results as tdx_groupnorm_act + tdx_dropout and tdx_dropout + tdx_groupnorm_act_bwd. */
int tdx_groupnorm_act_dropout(const void* x, int32_t x_ld, void* out, int32_t out_ld, int32_t out_off, const float* gamma, const float* beta,
                              int32_t B, int32_t L, int32_t C, int32_t groups, float eps, int32_t act, int32_t dtype, float p, uint64_t seed,
                              uint32_t layer, const int32_t* step_ctr, void* stream);
int tdx_groupnorm_act_dropout_bwd(const void* x, int32_t x_ld, const void* dy, int32_t dy_ld, int32_t dy_off, void* dx, int32_t dx_ld,
                                  int32_t dx_off, const float* gamma, const float* beta, float* dgamma_rows, float* dbeta_rows, int32_t B,
                                  int32_t L, int32_t C, int32_t groups, float eps, int32_t act, int32_t accumulate, int32_t dtype, float p,
                                  uint64_t seed, uint32_t layer, const int32_t* step_ctr, void* stream);

/* Data gradient of a STRIDE-1 Conv1d: tdx_conv_gemm on dy with the weights of engine.pack_conv_weight_dgrad (taps flipped, channel
 * roles exchanged).  Stride 2 (TrajUNet.py:44): the even input rows see only the centre tap (a k=1 conv of dy with W[:,:,1]^T), the
 * odd rows the two outer taps (a k=3 conv of dy with taps {0, W[:,:,2]^T, W[:,:,0]^T}); both phase tensors are (B, ceil(L/2), C),
 * and tdx_interleave2 merges them: out[b][u] (+)= u even ? even[b][u/2] : odd[b][u/2], u < L (engine.conv_dgrad_stride2). */
int tdx_interleave2(const void* even, const void* odd, void* out, int32_t out_ld, int32_t out_off, int32_t B, int32_t L, int32_t C,
                    int32_t accumulate, int32_t dtype, void* stream);

/* Data gradient through F.interpolate(mode='nearest', size=lvirt) (TrajUNet.py:93): dx[b][s] (+)= sum of du[b][u] over the
 * virtual rows u with min(floor(u * lin / lvirt), lin - 1) == s. */
int tdx_upsample_fold(const void* du, int32_t du_ld, int32_t du_off, void* dx, int32_t dx_ld, int32_t dx_off, int32_t B, int32_t lin,
                      int32_t lvirt, int32_t C, int32_t accumulate, int32_t dtype, void* stream);

/* y = act(z) and dz (+)= dy * act'(ref) on fp32 rows (n x c, row strides *_ld): ref = the pre-activation z for TDX_ACT_SWISH (Swish,
 * ModelUtils.py:14), the pre-activation or the OUTPUT for TDX_ACT_LEAKY (LeakyReLU(0.1) keeps the sign). */
int tdx_act_fwd(const float* z, int32_t z_ld, float* y, int32_t y_ld, int32_t n, int32_t c, int32_t act, void* stream);
int tdx_act_bwd(const float* dy, int32_t dy_ld, const float* ref, int32_t ref_ld, float* dz, int32_t dz_ld, int32_t n, int32_t c, int32_t act,
                int32_t accumulate, void* stream);

/* out[b][c] (+)= sum over the L rows of trajectory b of x: gradient of the embed projection that conv1's epilogue broadcast over
 * its rows (x + embed_proj(embed), TrajUNetBlocks.py:96). */
int tdx_traj_rowsum(const float* x, int32_t ld, float* out, int32_t out_ld, int32_t B, int32_t L, int32_t c, int32_t accumulate, void* stream);

/* nn.Embedding backward (TrajUNetBlocks.py:26-27,53-54): dtable[r][:] (+)= sum of dvec[b][j*dim : (j+1)*dim] over (b, j) with
 * idx[b][j] == r; fixed summation order. */
int tdx_embedding_bwd(const int64_t* idx, int32_t idx_ld, int32_t ncols, const float* dvec, int32_t dvec_ld, float* dtable, int32_t rows,
                      int32_t dim, int32_t B, int32_t accumulate, void* stream);

/* Backward of tdx_attention (fp32, L <= 16): dqkv (B, L, 3C) from qkv and the gradient dout (B, L, C) of its output. */
int tdx_attention_bwd(const float* qkv, const float* dout, float* dqkv, int32_t B, int32_t L, int32_t C, void* stream);

/* nn.Dropout(p), training mode (TrajUNetBlocks.py:85), in place on fp32 rows: x *= keep / (1 - p).  The mask comes from Philox4x32-10
 * keyed by (seed, step_ctr[0], layer, element index): calling it again with the same key on the gradient is the backward pass. */
int tdx_dropout(float* x, int32_t ld, int32_t n, int32_t c, float p, uint64_t seed, uint32_t layer, const int32_t* step_ctr, void* stream);

/* torch.optim.AdamW single-tensor update (train.ipynb:131: lr 5e-4, betas (0.9, 0.999), eps 1e-8, weight_decay 0.01);
 * step counts from 1.  Hyper-parameters are doubles, as torch keeps them: 1 - beta2 must be formed before rounding to fp32. */
int tdx_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int32_t step, void* stream);
/* The same update for every tensor of one parameter group in ONE launch (torch.optim.AdamW(model.parameters(), ...), train.ipynb:131,208:
 * ~200 tensors share one set of hyper-parameters, and one launch per tensor is launch-bound).  table_dev: a DEVICE-resident array of n
 * descriptors; the element arithmetic is tdx_adamw_step's, so the result is bit-identical to n single-tensor calls. */
typedef struct tdx_adamw_desc {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;
} tdx_adamw_desc;
int tdx_adamw_step_multi(const tdx_adamw_desc* table_dev, int32_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                         int32_t step, void* stream);

/* ---- Latent path: SDUNet_CAR (Models/StableDiffusion.py:114-222) between
 * VAE_InvTrans.encode / decode (Models/VAE.py:135-192).  Every Conv1d / nn.Linear / attention projection of both models is a
 * tdx_conv_gemm; the entry points below are the remaining ops.  Rows are channels-last (B, L, C) as everywhere else. ---- */

/* Core of nn.MultiheadAttention(C, 8, batch_first=True)(x, cond, cond) (StableDiffusionBlocks.py:58,77) after the input
 * projections: q (B, L, C) rows; K | V rows of the N <= 8 condition tokens (7 in the reference): token n of trajectory b at
 * kv + b * kv_bs + n * kv_ld (elements), 2C wide [K | V]; out (B, L, C) = per head softmax(q_h K_h^T / sqrt(C / 8)) V_h, heads
 * concatenated (the out_proj GEMM follows).  C in {128, 256, 512, 1024}; pitches and pointers 16-byte aligned. */
int tdx_cross_attention(const void* q, int32_t q_ld, const void* kv, int32_t kv_ld, int64_t kv_bs, void* out, int32_t out_ld, int32_t B,
                        int32_t L, int32_t N, int32_t C, int32_t heads, int32_t dtype, void* stream);

/* Core of the self-attention of InvTransformerBlock (VAEBlocks.py:109,120: tokens = channels, embedding = positions), fp32:
 * qkv (B, T, 3E) [q | k | v] -> out (B, T, E); E / heads <= 16 and T * E <= 8192. */
int tdx_mha_small(const float* qkv, float* out, int32_t B, int32_t T, int32_t E, int32_t heads, void* stream);

/* Layout: (B, C, L) fp32 channels-first -> rows (B, L, ld) of `dtype` (channels C .. ld-1 zero: padding up to the GEMM's
 * vector width), and back (VAEBlocks.py:120 switches between the two views of one tensor; the model I/O is channels-first). */
int tdx_chanfirst_to_rows(const float* x, void* out, int32_t B, int32_t C, int32_t L, int32_t ld, int32_t dtype, void* stream);
int tdx_rows_to_chanfirst(const void* rows, int32_t ld, float* out, int32_t B, int32_t C, int32_t L, int32_t dtype, void* stream);

/* Resampling along the trajectory on rows (B, lin, C) -> (B, lout, C):
 *   mode 0  nn.Upsample(mode='nearest') (StableDiffusion.py:166): src = min(floor(dst * (float)lin / lout), lin - 1);
 *   mode 1  nn.Upsample(mode='linear', align_corners=True) (VAE.py:142,156,160): src = dst * (lin - 1) / (lout - 1);
 *   mode 2  F.adaptive_avg_pool1d (VAE.py:181,192): mean over [floor(i lin / lout), ceil((i + 1) lin / lout)). */
int tdx_resample_rows(const void* x, void* out, int32_t B, int32_t lin, int32_t lout, int32_t C, int32_t mode, int32_t dtype, void* stream);

/* ConditionMixBlock's gathers (StableDiffusionBlocks.py:132-140), fp32, hidden width H: pe (B, H) = pos_enc[time];
 * mix_in (B, 7H) columns H..6H-1 = LeakyReLU([cell_emb[cat 0..3] | day_emb[cat 4]]) (the reference's forward concatenates the
 * gathered embeddings as they are, :136,140: `categorical_fc` is constructed, :109, and never called), columns 6H.. =
 * LeakyReLU(numerical_fc(num)); columns 0..H-1 are written by the time GEMM (each producer applies mix_fc's leading LeakyReLU,
 * :116, to its own slice).  catvec (B, 5H), optional (may be null): the raw gathered embeddings.
 * err_flag bits as tdx_embed_gather (1: time out of range, 2: categorical index out of range). */
int tdx_condmix_gather(const int64_t* time, const int64_t* cat, const float* num, const float* pos_enc, int32_t max_time,
                       const float* cell_emb, const float* day_emb, const float* num_w, const float* num_b, float* pe, float* catvec,
                       float* mix_in, int32_t hidden, int32_t B, int32_t* err_flag, void* stream);

/* Stem Conv1d(cin -> C0, 3, 1, 1), cin <= 8, from x (B, cin, L) fp32 to rows (B, L, C0) (StableDiffusion.py:127), and head
 * Conv1d(C0 -> cout, 3, 1, 1), cout <= 8, from rows to (B, cout, L) fp32 (:150); w in the reference layout, fp32. */
int tdx_stem_conv_n(const float* x, const float* w, const float* bias, void* out, int32_t B, int32_t L, int32_t cin, int32_t C0,
                    int32_t dtype, void* stream);
int tdx_head_conv_n(const void* a, const float* w, const float* bias, float* out, int32_t B, int32_t L, int32_t C0, int32_t cout,
                    int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRAJDIFF_HIP_H */
