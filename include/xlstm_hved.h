/* xlstm_hved.h -- C ABI of libxlstm_hved_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the volumetric forward/backward hot path of XLSTM-HVED.  The reference is pure
 * PyTorch (no custom ops), so every entry point below replaces an ATen op *sequence* issued by a reference
 * nn.Module.forward; the file:line it replaces is cited per function (paths relative to the reference repo).
 * The host side (the Python files under xlstm-hved_amd/) binds these with ctypes and keeps the reference's nn.Module surface.
 *
 * Conventions
 *  - Every function is asynchronous on `stream` (a hipStream_t passed as void*), does no allocation, no
 *    host synchronisation and keeps no pointer after returning: capture-safe (hipGraph).
 *  - Returns 0 on success, <0 on error (XH_ERR_ARG bad shape/unsupported combo, XH_ERR_DTYPE, XH_ERR_HIP).
 *  - Activations are contiguous NCDHW per sample; a `*_bs` argument is the batch stride in ELEMENTS, which
 *    lets a call read/write a channel slice of a larger tensor (virtual concat / split).
 *  - dtype: XH_F32 (0), XH_BF16 (1) or XH_F16 (2) is the STORAGE type of activations and activation gradients
 *    (fp16 = IEEE half, the reference's own AMP dtype, train.py:218; its 5-bit exponent needs the caller's loss scaling
 *    in backward exactly like the reference's GradScaler, train.py:207,265-268 -- the library never scales).
 *    Parameters, parameter gradients, statistics and coefficients are always fp32 (sums: fp64).
 *    Arithmetic is fp32 in every kernel; the MFMA conv kernels multiply 16-bit operands and accumulate in fp32.
 *  - The caller owns all memory, including the `red` reduction buffers which it must zero before a call that
 *    accumulates into them (documented per call).
 */
#ifndef XLSTM_HVED_H
#define XLSTM_HVED_H
#ifdef __cplusplus
extern "C" {
#endif

#define XH_MAX_WPTR 8            /* weight / bias / gradient pointers per conv call: one per group for up to 8 groups */
#define XH_F32 0
#define XH_BF16 1
#define XH_F16 2
#define XH_ACT_NONE 0
#define XH_ACT_RELU 1
#define XH_ACT_LRELU 2
#define XH_ACT_SIGMOID 3

int xh_abi_version(void);
/* key 0: use the bf16-MFMA implicit-GEMM conv kernels where eligible (default 1); for A/B tests.
 * key 1: MFMA kernel ablation mask (microbenchmarks only).
 * key 2: disable mask for specialised kernels: bit 0 sliding-window depthwise conv (+wgrad), bit 1 exact-2x trilinear
 *        kernels, bit 2 vectorised stride-2 conv forward / data gradient, bit 3 chunk-recurrent mLSTM (falls back to the
 *        tiled O(S^2) contraction; A/B tests), bit 4 quad-channel k3 forward / data-gradient kernel, bit 5 quad-channel k3
 *        weight-gradient kernel (both fall back to the implicit-GEMM kernels), bit 6 statistics fan-in (direct atomics),
 *        bit 7 depthwise k3 convs through the quad-channel kernel (fall back to the sliding-window vector kernel),
 *        bit 8 input-channel split inside the blocks of the stride-2 vector conv on small outputs, bit 9 shared launches of the
 *        k=1 and stride-2 weight gradients of xh_conv3d_wgrad_batch (one launch per problem instead), bit 10 one-pass DuSE gate backward.
 * key 3: target workgroup count of the k3 MFMA forward kernel (default 512 = 2 per CU; microbenchmarks: 1024-4096 were 7-25 % slower).
 * key 4: 1 = <=128-VGPR instances of the k3 MFMA forward kernel for <= 8 input channels (microbenchmarks: spills, 2x slower).
 * key 5: K step of the discriminator's implicit GEMM in 32-channel quarters (1 | 2, default 2).
 * key 10-13: launch-plan limits of the generator's conv kernels (microbenchmarks; defaults are the measured optima).
 * key 16: workgroup cap of the 1<->2-channel k3 stencil kernels (default 512).
 * key 17: workgroup count below which a quad-channel k3 launch walks 4, then 2 output planes per workgroup instead of 8
 *         (default 512; 0 = always 8).
 * key 18: REMOVED in round 6 (returns XH_ERR_ARG): the arithmetic of fp32 storage is xh_conv_desc.arith, a field of every call.
 * key 19: persistent, tile-pipelined variant of the quad-channel k3 kernel (csrc/conv3d_q4p.hip): 0 = never, 1 = forward launches with
 *         several input quads per tile and >= 2048 stages (default: where it was measured to win), 2 = every 8-plane launch.
 * key 20: mask of the row widths on which quad-channel k3 convs take the full-row tiles of csrc/conv3d_q4w.hip: bit 1 = 128 voxels,
 *         bit 0 = 64 voxels (default 3); 0 = always the 32-wide tiles of csrc/conv3d_q4.hip (A/B switch; the outputs are
 *         bit-identical).  Bit 2 = 1: no workgroups of three output quads (the 4 -> 12 data gradients stage their tile once per
 *         output quad, as before round 6; bit-identical too).
 * key 21: 0 = no full-row weight-gradient kernel (csrc/conv3d_wgrad_q5.hip; the 32-wide tile kernel instead).  key 22: its
 *         workgroup budget per launch (default 256 = one per CU).  key 23: 0 = rows of 32 voxels stay with the tile kernel (default 1 since round 6).
 * key 24: input-stationary 7^3 gate-conv kernel (csrc/conv7_mfma.hip): 0 never, 1 volumes >= 2^20 voxels (default), 2 B fragments in
 *         registers, 3 every volume.  key 25: REMOVED in round 6 (xh_conv_desc.arith bit XH_ARITH_K7_VECTOR).
 * key 26: groups of 8 class workgroups per launch of the discriminator's source-block weight gradient (default 32 = 256 workgroups).
 * key 27: workgroup target of the row-streaming norm / element-wise kernels (default 2048).
 * key 28: full-row weight-gradient kernel on rows of 64 voxels: bit 0 = a unit stages two input quads, bit 1 = two output quads
 *         bit 2 = rows of 128 voxels: three input quads against one staging of dY; bit 3 = fp32 storage with fp16 operands takes the
 *         full-row kernel too (measured slower: off).  Default 7; 0 = one quad of each per unit.
 * key 14: discriminator conv A/B mask (csrc/dconv.hip): bit 0 one launch per parity class, bit 1 no 256x64 tiles, bit 2 no
 *         small tiles, bit 3 64x128 instead of 64x64, bit 4 no tap pairs in the 64-channel weight gradient, bit 5 / 7 register
 *         prefetch of 4 / 2 K steps on the 256x16 tile, bit 6 no XCD remap, bit 8 no 256x128 tiles, bit 10 / 11 generic kernel instead of the LDS-halo
 *         kernels of the first conv (data gradient / forward), bit 14 / 17 gather kernels instead of the source-block kernels of the
 *         stride-2 data gradients / k = 4 weight gradients.
 *
 * PROCESS-GLOBAL STATE (the only two exceptions to "no mutable state in the library"): the option table behind
 * xh_set_option (plain ints, e.g. g_q4_maxc / g_q4_wgs in csrc/conv3d_q4.hip) and the name buffer behind xh_last_conv_kernel
 * are per PROCESS, not per device, stream or call.  They are development / measurement knobs: every option has a default that
 * is the measured optimum, the deployment model is one process per GPU, and the package's Python files write none of them
 * (bench.py, tools/ and tests/ do).
 * WHAT AN OPTION CAN CHANGE.  No option selects an arithmetic MODE any more: the one that did (key 18, with 25) is now
 * xh_conv_desc.arith.  The remaining keys choose between kernels / launch plans that compute the same function:
 *   - bit-identical results whatever the value: keys 3, 10-13, 16, 17, 19, 20, 22, 26, 27 and key 14's tile / remap bits;
 *   - same products, sums taken in another order (differences at fp32 / fp64 round-off of the sums): key 2 bits 0, 1, 2, 6, 8, 9,
 *     10; keys 5, 21, 23, 28; key 14's kernel-choice bits;
 *   - a different kernel FAMILY for the same conv, one operand rounding apart in 16-bit storage (tests hold both to the same
 *     bounds): key 0 (MFMA vs vector kernels), key 2 bits 3, 4, 5, 7, key 24 (input- vs output-stationary 7^3 kernel).
 * Neither function is thread-safe against concurrent launches from other host threads: set options before the first launch;
 * read xh_last_conv_kernel on the thread that made the call.  Every other entry point is re-entrant: all device memory,
 * workspaces and the statistics fan-in block are the caller's, the stream and the arithmetic mode are arguments. */
int xh_set_option(int key, int value);
/* Host-side launch plan of the multi-problem weight-gradient launches whose workgroups are all resident together (no device work;
 * exported so that the plan can be tested without a GPU).  Problem i has units[i] units of cost[i] each (any unit of time); a unit
 * takes wq[i] workgroups (1 .. cap[i]) that share it equally, so the launch lasts max_i cost[i] / wq[i].  Fills wq with the
 * assignment that minimises that maximum subject to sum_i units[i] * wq[i] <= budget (when even one workgroup per unit exceeds the
 * budget: one each), spare workgroups going to the slowest units; returns the planned duration, < 0 on bad arguments. */
double xh_wgrad_plan_minmax(int n, const double* cost, const int* units, const int* cap, int budget, int* wq);
/* Name of the kernel template instance the most recent xh_conv3d_fwd / xh_conv3d_wgrad call of THIS PROCESS launched (static
 * storage, overwritten by the next call on any thread; the same spelling rocprofv3 prints), so measurements can be attributed
 * to a kernel without a profiler attached.  Measurement aid only. */
const char* xh_last_conv_kernel(void);

/* ------------------------------------------------------------------------------------------------
 * 3D convolution family.  Replaces nn.Conv3d together with the norm/activation modules wrapped around it
 * in create_conv/SingleConv (buildingblocks.py:381-461), BasicConv (buildingblocks.py:13-31), DWConvNorm
 * (sa_modules/sa_module.py:79-85), AttenModule2's 7^3 grouped convs (buildingblocks.py:271-274,283-296),
 * DuSEAttention's squeeze/adjust convs (modules/DuSFE.py:135-144) and the 1x1 heads (RA_HVED.py:148-149,
 * 192-196,480,640).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int dtype;
  int N, Cin, Cout, groups;     /* Cin/Cout are totals over all groups */
  int D, H, W;                  /* input spatial size */
  int Do, Ho, Wo;               /* output spatial size: (D + 2*(k/2) - k)/stride + 1 */
  int k, stride;                /* k in {1,3,7}; padding = k/2; stride in {1,2} (2 only with k=3) */
  int Ca;                       /* input channels [0,Ca) come from xa, [Ca,Cin) from xb (Ca==Cin: xb unused) */
  long long xa_bs, xb_bs, y_bs; /* batch strides (elements) */
  int n_wptr;                   /* 1: w[0] holds all groups; ==groups (<= XH_MAX_WPTR): w[g] holds group g */
  int transposed;               /* 1: compute the data-gradient correlation using FORWARD-layout weights
                                   [Cin][Cout/groups][k^3] (roles swapped, taps flipped); stride must be 1 */
  int pre;                      /* 1: input transform v = leaky(x*pre_sc[n,c] + pre_sh[n,c], pre_slope)
                                   applied before zero padding (slope 1 = affine only)
                                   2: (data gradients, see xh_conv3d_fuses_norm_bwd) the InstanceNorm backward of the stage behind
                                   this conv applied on load: v = A*xa + C*px + B, the coefficients of xh_in_bwd_apply derived
                                   in-kernel from nb_red / nb_mean / nb_rstd; one source (Ca == Cin) */
  float pre_slope;
  int act;                      /* epilogue activation XH_ACT_* (after bias) */
  float act_slope;
  int epi;                      /* 0 none
                                   1 activation/norm backward: g = out * leaky'(e*e_sc+e_sh); store g;
                                     red[n][c][0] += g, red[n][c][1] += g*e   (e = raw pre-norm value)
                                   2 output moments: red[n][c][0] += y, red[n][c][1] += y*y (y as stored) */
  int Cea;                      /* epi==1: e channels [0,Cea) from ea, rest from eb */
  long long ea_bs, eb_bs;
  float e_slope;
  long long px_bs, pd_bs;       /* pre == 2: batch strides of px and pd */
  int arith;                    /* ARITHMETIC of this call when dtype == XH_F32 (16-bit storage ignores it): XH_ARITH_* bits.
                                   0 (default): the fp32 vector (FMA) kernels.
                                   XH_ARITH_F32_SPLIT: the matrix cores with every fp32 value as a two-term fp16 pair (csrc/conv3d_q4s.hip:
                                   three fp16 MFMA products per fp32 product, ~22 significand bits, fp32 accumulation) for k = 3 stride-1
                                   convs and their data gradients; their weight gradients and the 7^3 gate convs with operands rounded
                                   ONCE to fp16.  In the backward pass the activation gradients then need the caller's loss scale, as
                                   with fp16 storage.
                                   XH_ARITH_K7_VECTOR (with F32_SPLIT): the 7^3 gate convs stay on the fp32 vector kernels.
                                   The mode is part of the CALL (round 6; it was process state behind xh_set_option keys 18 / 25):
                                   two models of different modes, or a captured graph next to eager calls, cannot disturb each other. */
  int bcast;                    /* 0, or 4: a BROADCAST operand (round 6).  Logical channel c of the operand reads channel c / 4 of the
                                   tensor given, which has a quarter of the channels: the four channels of a group are one stored
                                   channel seen through four different (pre_sc, pre_sh) / (e_sc, e_sh) pairs.
                                     xh_conv3d_fwd, transposed == 0: the operand is the input xa (Ca == Cin, 4 input channels per group);
                                     xh_conv3d_fwd, transposed == 1, epi == 1: the operand is e (ea; Cea == Cout, 4 per group), and y may
                                       be NULL: the masked gradient is then only summed (red), not stored;
                                     xh_conv3d_wgrad[_batch]: the operand is the forward input xa.
                                   This is how the first conv of XLSTM_HVED's encoders reads the init blocks' 1x1 convs WITHOUT their
                                   output ever being stored: InstanceNorm(w_c x + b_c) = sign(w_c) (x - mean) / sqrt(var + eps / w_c^2), a
                                   per-channel affine of the input modality itself (RA_HVED.py:345-349,548-553; xh_init_fold_fwd).
                                   Served by the full-row quad-channel kernels only (16-bit storage, k = 3, stride 1, rows of 64 / 128
                                   voxels, H a multiple of 8, D >= 4: xh_conv3d_supports_bcast); anything else returns XH_ERR_ARG. */
} xh_conv_desc;
#define XH_ARITH_F32_SPLIT 1
#define XH_ARITH_K7_VECTOR 2

typedef struct {
  const void* xa; const void* xb;
  const float* w[XH_MAX_WPTR]; const float* b[XH_MAX_WPTR];       /* b[i] may be NULL (no bias) */
  const float* pre_sc; const float* pre_sh;   /* [N][Cin] when pre */
  void* y;
  const void* ea; const void* eb;             /* epi==1 */
  const float* e_sc; const float* e_sh;       /* [N][Cout] */
  double* red;                                /* [N][Cout][2], caller zeroes, epi!=0 */
  void* ws; long long ws_bytes;               /* scratch for packed bf16 MFMA weight fragments (may be NULL:
                                                 the vector kernel is used); size from xh_conv3d_workspace_bytes */
  /* Optional fused InstanceNorm finalisation (k = 3: the MFMA path, and the stride-2 convs; needs pre == 1): when fin_red is
   * given, the conv kernel itself turns the raw sums fin_red[n][c] = (sum x, sum x^2) over fin_count voxels into scale = rstd,
   * shift = -mean*rstd (every workgroup for its own input channels, in fp64) and ALSO WRITES pre_sc / pre_sh / fin_mean /
   * fin_rstd (kept for the backward), replacing a separate xh_norm_finalize launch.  A stride-1 call that cannot take the MFMA
   * path returns an error; a stride-2 call whose kernel cannot take it (fp32 storage, unaligned rows) finalises with a launch
   * of its own first. */
  const double* fin_red; float* fin_mean; float* fin_rstd; long long fin_count;
  int ws_packed;                              /* 1: ws already holds this conv's fragments (xh_conv3d_prepack): no pack launch */
  /* Optional statistics fan-in workspace (epi != 0): xh_fanin_bytes() bytes, 128-byte aligned, ZERO on entry; the launch
   * leaves it zero again.  With it, launches of >= 256 workgroups per (sample, channel block) sum their epilogue statistics
   * through a two-level tree in this block instead of same-cache-line atomics on red[] (csrc/fanin.h).  The block is the only
   * state such a launch shares with others: launches that may run CONCURRENTLY (different streams, parallel branches of one
   * graph) must be given different blocks; launches ordered on one stream may share one.  NULL / too small: direct atomics. */
  void* fan; long long fan_bytes;
  /* pre == 2 (replaces a separate xh_in_bwd_apply pass between two chained data gradients, buildingblocks.py:464-507 backward):
   * xa = g, the activation-masked data gradient the NEXT conv's backward left (its epi == 1 output), px = the raw tensor that
   * stage normalised (same layout as xa), nb_red[N][Cin][2] = that launch's sums (sum g, sum g*px), nb_mean / nb_rstd [N][Cin] the
   * InstanceNorm statistics of px, nb_count = voxels per channel.  pd (optional): the transformed tensor A*g + C*px + B is ALSO
   * stored there (layout of xa) -- the weight gradient of this conv reads it; written by this launch, complete when it ends. */
  const void* px; void* pd;
  const double* nb_red; const float* nb_mean; const float* nb_rstd; long long nb_count;
  /* Optional with fin_red, N == 1 (xh_conv3d_fuses_bn_finalize): the fused finalisation is a training-mode BatchNorm3d instead of an
   * InstanceNorm -- scale = rstd * fin_gamma[c], shift = fin_beta[c] - mean * rstd * fin_gamma[c] (one sample: batch statistics ==
   * instance statistics), and the launch also advances the running statistics fin_rm / fin_rv by fin_steps momentum-0.1 updates
   * with the unbiased variance (what xh_norm_finalize mode 1 does in a launch of its own; sa_modules/sa_module.py:79-85). */
  const float* fin_gamma; const float* fin_beta; float* fin_rm; float* fin_rv; int fin_steps;
  /* Optional with a broadcast e operand (xh_conv_desc.bcast, transposed, epi == 1): e_ctr[N][Cout], a per-channel centre; the second
   * sum is then red[n][c][1] += g * (e - e_ctr[n][c]).  With e_ctr = the channel mean of e the caller gets sum g (e - mean) without
   * the cancellation of sum g e - mean sum g (five digits for an input modality whose spread is small against its mean). */
  const float* e_ctr;
} xh_conv_ptrs;

/* Size in bytes of a statistics fan-in workspace (xh_conv_ptrs.fan).  The library allocates no device memory and keeps no
 * device state of its own: every buffer, this one included, belongs to the caller. */
long long xh_fanin_bytes(void);

/* y = act(conv(pre(x)) + b)  [+ epilogue].  Also serves as the data-gradient of a stride-1 conv
 * (transposed=1).  Reference: F.conv3d as used throughout RA_HVED.py:510-648. */
int xh_conv3d_fwd(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p);
/* Two INDEPENDENT k = 3 stride-1 convolutions of one shape (same storage type, extents, Cout, pre / epi variant) in ONE launch of
 * the full-row quad-channel kernel: returns 0 when launched, 1 when the two are not such a pair (nothing launched: the caller calls
 * xh_conv3d_fwd twice), < 0 on a bad argument.  Both must have their fragments packed (ws_packed).  Same results as two calls.  The
 * decoder's recon | seg streams (RA_HVED.py:171-183) run convs of identical shape on different inputs; at 64^3 each is half a
 * resident round of workgroups. */
int xh_conv3d_fwd_pair(void* stream, const xh_conv_desc* d0, const xh_conv_ptrs* p0, const xh_conv_desc* d1, const xh_conv_ptrs* p1);
/* 1 when xh_conv3d_fwd takes this desc (pre == 1) with the BatchNorm flavour of the fused finalisation (fin_gamma ...): the
 * quad-channel MFMA kernel, one sample. */
int xh_conv3d_fuses_bn_finalize(const xh_conv_desc* d);
/* 1 when xh_conv3d_fwd takes this desc with pre == 2 (the quad-channel MFMA kernel: 16-bit storage, k = 3, stride 1, rows of 32
 * voxels, <= 48 channels per group, ...); 0: the caller materialises the tensor with xh_in_bwd_apply and calls with pre == 0. */
int xh_conv3d_fuses_norm_bwd(const xh_conv_desc* d);
/* 1 when every call of the broadcast form of this desc's convolution (forward, data gradient with e broadcast, weight gradient)
 * is served: d describes the FORWARD conv with bcast set. */
int xh_conv3d_supports_bcast(const xh_conv_desc* d);
/* The init blocks' 1x1 convs (RA_HVED.py:345-349: one modality -> B channels, X_c = w_c x_m + b_c) folded into the InstanceNorm that
 * is their only consumer: IN(X_c) = sc_c x_m + sh_c with sc_c = w_c R_c, sh_c = -w_c mean_m R_c, R_c = 1 / sqrt(w_c^2 var_m + eps)
 * (the bias drops out).  red_x[N][M][2] = fp64 sums (sum x, sum x^2) of the M stored channels over `count` voxels; w[m] = the B
 * weights of modality m; sc / sh / rstd / ctr: [N][M * B], ctr = fp32(mean_m) (the e_ctr of the data gradient).  The first encoder
 * conv then reads x itself (xh_conv_desc.bcast). */
int xh_init_fold_fwd(void* stream, const double* red_x, long long count, int N, int M, int B, const float* const w[XH_MAX_WPTR],
                     float eps, float* sc, float* sh, float* rstd, float* ctr);
/* Backward of the fold: red_g[N][M * B][2] = (sum g, sum g (x - ctr)) left by the first conv's data gradient (epi == 1 with the
 * broadcast e operand and e_ctr = ctr); dw[m][j] += eps R^3 sum g (x - mean), the exact gradient of the init weight (the bias
 * gradient is exactly zero). */
int xh_init_fold_bwd(void* stream, const double* red_x, long long count, int N, int M, int B, const float* const w[XH_MAX_WPTR],
                     float eps, const double* red_g, float* const dw[XH_MAX_WPTR]);
/* Bytes of p->ws the bf16-MFMA implicit-GEMM path wants for this desc (0: shape not eligible, vector kernel). */
long long xh_conv3d_workspace_bytes(const xh_conv_desc* d);
/* Weights are constant within a training step: packs the MFMA weight fragments of n convolutions (forward and data-gradient
 * calls alike; d[i] / p[i] as xh_conv3d_fwd will receive them -- only shape, weights and ws are read) into their p[i]->ws with
 * ONE launch per 24 convolutions, instead of one small launch in front of every convolution.  A later xh_conv3d_fwd with
 * ws_packed = 1 and the same ws skips its own pack.  Convolutions that are not on the MFMA path are skipped.  The caller
 * re-packs whenever the weights change (xlstm-hved_amd/ops.py: once per forward, keyed by the parameters' version counters).
 * Reference: the parameters of every nn.Conv3d(k=3) of RA_HVED.py:510-648 / buildingblocks.py:381-461. */
int xh_conv3d_prepack(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p);
/* The same packing in ONE launch for any number of convolutions (up to 256 pack jobs), for callers that pack the same set every step:
 * xh_conv3d_prepack_table writes the job table of the n convolutions (pointers and shapes; what xh_conv3d_prepack would put into its
 * kernel arguments) into xh_conv3d_prepack_table_bytes() bytes of HOST memory -- no launch, no device access; the caller copies the
 * table to device memory it owns (once, and again whenever a weight or workspace ADDRESS or the set changes -- weight VALUES may
 * change freely) and calls xh_conv3d_prepack_run(stream, device copy, nblocks) every step, with nblocks = the second int of the
 * table.  Replaces ceil(jobs / 24) launches of 10-15 us by one (the 54 convolutions of XLSTM_HVED: three -> one). */
long long xh_conv3d_prepack_table_bytes(void);
int xh_conv3d_prepack_table(int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, void* host_table);
int xh_conv3d_prepack_run(void* stream, const void* dev_table, int nblocks);

/* Data gradient of a k=3, stride=2, pad=1 conv (the DRB SingleConv, RA_HVED.py:396-397).  Desc fields
 * describe the FORWARD conv (Cin,D,H,W = forward input; Cout,Do,Ho,Wo = forward output); x* = dY
 * (Cout channels, Do..), y = dX (Cin channels).  epi 0/1 as above with e = forward input. */
int xh_conv3d_dgrad_s2(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p);

/* Weight/bias gradient: dw[co][ci][tap] += sum_{n,p} dy[n,co,p] * pre(x)[n,ci,p*stride+tap-pad],
 * db[co] += sum dy.  Desc describes the FORWARD conv; ptrs: xa/xb/pre_* = forward input, `ea` = dY with
 * batch stride ea_bs.  dw[i]/db[i] are laid out like w[i]/b[i], fp32, ACCUMULATED into (caller zeroes). */
int xh_conv3d_wgrad(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR]);
/* The weight gradients of `n` convolutions in as few launches as possible: d[i] / p[i] / dw[i] / db[i] are what
 * xh_conv3d_wgrad would take for problem i (db may be NULL, db[i][j] may be NULL).  Weight gradients are off the critical
 * path of a backward pass, so a caller can collect them and issue this once at the end: the k=3 MFMA problems of a
 * (storage type, volume class) share launches 7 at a time (their workgroups run side by side instead of 16 latency-bound
 * launches one after the other), the k=1 problems of a storage type 20 at a time, the 7^3 gate problems 4 at a time, the 1<->2-channel k=3 stencils 8 at a time, the vectorised k=3 stride-2
 * problems 4 at a time, everything else is forwarded to xh_conv3d_wgrad.  Same accumulate (+=) semantics. */
int xh_conv3d_wgrad_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p,
                          float* const (*dw)[XH_MAX_WPTR], float* const (*db)[XH_MAX_WPTR]);
/* Scratch (bytes) xh_conv3d_wgrad wants in p->ws for this shape (per-workgroup partial gradients of the 7^3 MFMA weight
 * gradient); 0 = none.  With less, the call uses the vector kernel. */
long long xh_conv3d_wgrad_workspace_bytes(const xh_conv_desc* d);

/* ------------------------------------------------------------------------------------------------
 * Normalisation statistics and elementwise stages.
 * ------------------------------------------------------------------------------------------------ */
/* red[n][c][0] += sum x, red[n][c][1] += sum x^2 over DHW.  red has row stride red_rs doubles per n
 * (so a slice [N][Ctot][2] can be filled from several sources).  nn.InstanceNorm3d / BatchNorm3d /
 * GroupNorm statistics (buildingblocks.py:429-433; sa_module.py:75; DuSFE.py:108-110). */
int xh_moments(void* stream, int dtype, const void* x, long long x_bs, int N, int C, long long DHW,
               double* red, long long red_rs);
/* The same over a virtual concat (xa | xb) -- the decoder's torch.cat((enc, x), 1) input (buildingblocks.py:732) -- in one
 * launch: red[n][0..CA+CB) gets both. */
int xh_moments2(void* stream, int dtype, const void* xa, long long xa_bs, int CA, const void* xb, long long xb_bs, int CB,
                int N, long long DHW, double* red, long long red_rs);

/* Turns moments into the affine pre-transform (sc, sh) consumed by xh_conv3d_fwd/xh_affine_act and the
 * saved (mean, rstd).
 *  mode 0 InstanceNorm: per (n,c).   mode 1 BatchNorm(train): per c over n, updates running stats
 *  `steps` times (momentum 0.1, unbiased var) -- steps=4 reproduces the 4x evaluation at RA_HVED.py:548-552.
 *  mode 2 BatchNorm(eval): uses running stats, ignores red.   mode 3 GroupNorm: per (n, group of gs ch).
 *  gamma/beta may be NULL (=1/0).  Outputs sc,sh,mean,rstd are [N][C]. */
int xh_norm_finalize(void* stream, int mode, const double* red, int N, int C, long long count, int gs,
                     float eps, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, int steps, float* sc, float* sh, float* mean, float* rstd);

/* y = act(x*sc[n,c] + sh[n,c]) -- materialises norm+activation (BasicConv tail, BatchNorm apply). */
int xh_affine_act(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs, int N, int C,
                  long long DHW, const float* sc, const float* sh, int act, float slope);
/* The same pass with the InstanceNorm finalisation inside: (sc, sh) are derived in the kernel from the raw channel sums
 * red[N][C][2] a conv epilogue left (count = DHW, eps 1e-5, no affine), and sc / sh / mean / rstd [N][C] are written for the
 * backward pass.  Replaces xh_norm_finalize(mode 0) + xh_affine_act for BasicConv (buildingblocks.py:13-31) in one launch. */
int xh_in_affine_act(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs, int N, int C,
                     long long DHW, const double* red, int act, float slope, float* sc, float* sh, float* mean, float* rstd);
/* BatchNorm3d flavour (mode 1 train: batch statistics from red, running statistics updated `steps` times as in
 * xh_norm_finalize; mode 2 eval: running statistics): DuSEAttention's output norms (modules/DuSFE.py:108-110,151-153). */
int xh_bn_affine_act(void* stream, int dtype, int mode, const void* x, long long x_bs, void* y, long long y_bs, int N, int C,
                     long long DHW, const double* red, float eps, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, int steps, int act, float slope, float* sc, float* sh, float* mean, float* rstd);

/* Backward reduce through y = leaky(x*sc+sh): g = dy*leaky'(.), red[n][c][0] += g, red[n][c][1] += g*x. */
int xh_act_bwd_reduce(void* stream, int dtype, const void* dy, long long dy_bs, const void* x, long long x_bs,
                      int N, int C, long long DHW, const float* sc, const float* sh, float slope, double* red);

/* Coefficients of the norm backward from the reduced sums: dx = A*g + Cc*x + B.
 * mode as in xh_norm_finalize (0 IN, 1 BN train, 2 BN eval, 3 GN).  Also emits dgamma/dbeta (ACCUMULATED)
 * when the norm is affine.  A,B,Cc are [N][C]. */
int xh_norm_bwd_coef(void* stream, int mode, const double* red, int N, int C, long long count, int gs,
                     const float* gamma, const float* mean, const float* rstd, float* A, float* B, float* Cc,
                     float* dgamma, float* dbeta);

/* dx (+)= A[n,c]*g + Cc[n,c]*x + B[n,c], with g = dy (have_g=1: dy already holds g) or
 * g = dy*leaky'(x*sc+sh) (have_g=0).  accumulate=1 adds into dx. */
int xh_norm_bwd_apply(void* stream, int dtype, const void* dy, long long dy_bs, const void* x, long long x_bs,
                      void* dx, long long dx_bs, int N, int C, long long DHW, const float* A, const float* B,
                      const float* Cc, int have_g, const float* sc, const float* sh, float slope, int accumulate);
/* xh_norm_bwd_coef + xh_norm_bwd_apply in one launch for BatchNorm (mode 1 train, 2 eval) and GroupNorm (mode 3): the
 * coefficients are derived inside the kernel from red[N][C][2] = {sum g, sum g*x}, mean, rstd [N][C] (count = DHW), g = dy;
 * dgamma / dbeta [C] are accumulated as xh_norm_bwd_coef does.  Same arithmetic, same results. */
int xh_norm_bwd_fused(void* stream, int dtype, int mode, const void* dy, long long dy_bs, const void* x, long long x_bs,
                      void* dx, long long dx_bs, int N, int C, long long DHW, const double* red, int gs,
                      const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta);
/* Two BatchNorm3d modules over the two channel halves of ONE tensor -- DuSEAttention's bn_fuse_ch1 / bn_fuse_ch2 applied to the
 * recon | seg pair in one launch each way (modules/DuSFE.py:151-154): channels [0, Chalf) take the first parameter set,
 * [Chalf, C) the second (its arrays indexed from 0).  Otherwise xh_bn_affine_act / xh_norm_bwd_fused (mode 1 | 2). */
int xh_bn_affine_act2(void* stream, int dtype, int mode, const void* x, long long x_bs, void* y, long long y_bs, int N, int C, int Chalf,
                      long long DHW, const double* red, float eps, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, const float* gamma2, const float* beta2, float* running_mean2, float* running_var2, int steps,
                      int act, float slope, float* sc, float* sh, float* mean, float* rstd);
int xh_norm_bwd_fused2(void* stream, int dtype, int mode, const void* dy, long long dy_bs, const void* x, long long x_bs, void* dx,
                       long long dx_bs, int N, int C, int Chalf, long long DHW, const double* red, const float* gamma,
                       const float* gamma2, const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dgamma2,
                       float* dbeta2);
/* InstanceNorm backward in one launch (autograd of nn.InstanceNorm3d in create_conv, buildingblocks.py:431, and in BasicConv,
 * buildingblocks.py:21-24): coefficients derived per (n, c) row from the raw sums red = (sum g, sum g*x) of
 * xh_act_bwd_reduce / the conv epilogue, and the forward's mean / rstd.  red, mean, rstd (and sc, sh when have_g == 0)
 * point at this tensor's first channel inside arrays whose rows are stat_rs channels wide (virtual concat). */
int xh_in_bwd_apply(void* stream, int dtype, const void* dy, long long dy_bs, const void* x, long long x_bs, void* dx,
                    long long dx_bs, int N, int C, long long DHW, const double* red, const float* mean, const float* rstd,
                    int stat_rs, int have_g, const float* sc, const float* sh, float slope, int accumulate);
/* InstanceNorm backward of a virtual concat in one launch: dy / red / mean / rstd are CA+CB wide, the first CA channels
 * read xa and write dxa, the others xb / dxb (g given: have_g = 1 of xh_in_bwd_apply).  accumulate: bit 0: dxa +=, bit 1:
 * dxb += (the buffer already holds the gradient share of another consumer of that tensor). */
int xh_in_bwd_apply2(void* stream, int dtype, const void* dy, long long dy_bs, const void* xa, long long xa_bs, void* dxa,
                     long long dxa_bs, int CA, const void* xb, long long xb_bs, void* dxb, long long dxb_bs, int CB, int N,
                     long long DHW, const double* red, const float* mean, const float* rstd, int accumulate);

/* nn.MaxPool3d(2) (buildingblocks.py:635-636,656-657) and its backward (first maximum in scan order wins). */
int xh_maxpool2_fwd(void* stream, int dtype, const void* x, void* y, int NC, int D, int H, int W);
int xh_maxpool2_bwd(void* stream, int dtype, const void* x, const void* dy, void* dx, int NC, int D, int H, int W,
                    int accumulate);

/* F.interpolate(mode='trilinear', align_corners=False) to an arbitrary size (buildingblocks.py:785-787,
 * RA_HVED.py:600-601) and its adjoint. */
int xh_upsample_trilinear_fwd(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs,
                              int N, int C, int D, int H, int W, int Do, int Ho, int Wo);
int xh_upsample_trilinear_bwd(void* stream, int dtype, const void* dy, long long dy_bs, void* dx, long long dx_bs,
                              int N, int C, int D, int H, int W, int Do, int Ho, int Wo, int accumulate);
/* BasicConv's InstanceNorm3d + LeakyReLU (buildingblocks.py:13-31) applied inside the exact-2x upsampling that follows it
 * (RA_HVED.py:599-601): y (N, C, 2D, 2H, 2W) = up2x(leaky(IN(x))), IN finalised from the conv epilogue's raw channel sums
 * red[n][c] = (sum x, sum x^2); sc / sh / mean / rstd (N x C, fp32) are written for the backward pass.  Returns 1 with nothing
 * launched when the exact-2x kernel does not take the layout (then: xh_in_affine_act + xh_upsample_trilinear_fwd). */
int xh_upsample2x_in_act_fwd(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs, int N, int C,
                             int D, int H, int W, const double* red, float slope, float* sc, float* sh, float* mean, float* rstd);
/* Its adjoint: dx (N, C, D, H, W) = up2x^T(dy), and red[n][c] += (sum dz, sum dz * y0) with dz = dx * leaky'(y0 * sc + sh) --
 * the sums of xh_act_bwd_reduce over the values as stored (caller zeroes red).  Returns 1 when not taken (then:
 * xh_upsample_trilinear_bwd + xh_act_bwd_reduce). */
int xh_upsample2x_bwd_act_reduce(void* stream, int dtype, const void* dy, long long dy_bs, void* dx, long long dx_bs, int N, int C,
                                 int D, int H, int W, const void* y0, long long y0_bs, const float* sc, const float* sh,
                                 float slope, double* red);

/* ------------------------------------------------------------------------------------------------
 * Multi-problem launches of the five passes of the latent path (RA_HVED.py:599-603: per fusion level PoE output -> BasicConv 1x1
 * -> 2x upsampling -> BasicConv depthwise 3^3).  The four levels are independent of each other and, except the finest, far too
 * small to fill the chip: a launch of 4 - 64 workgroups costs its ~5 us of dispatch whatever it holds.  Each entry point below runs
 * the SAME pass for up to XH_LEVELS_MAX problems in ONE launch (a table of per-problem arguments travels in the kernel arguments;
 * a workgroup finds its problem from its index).  Semantics per problem = the single-problem entry point named in the struct; the
 * results are the same bits (same kernel bodies, same per-problem grids).  All problems of a call share dtype.  Returns 1 with
 * nothing launched when a problem's layout is not taken by the kernel the pass uses (then: the single-problem calls).
 * ------------------------------------------------------------------------------------------------ */
#define XH_LEVELS_MAX 4
typedef struct {                 /* xh_in_affine_act */
  const void* x; long long x_bs; void* y; long long y_bs; int N, C; long long DHW; const double* red; int act; float slope;
  float *sc, *sh, *mean, *rstd;
} xh_in_affine_act_args;
typedef struct {                 /* xh_act_bwd_reduce */
  const void* dy; long long dy_bs; const void* x; long long x_bs; int N, C; long long DHW; const float *sc, *sh; float slope; double* red;
} xh_act_bwd_reduce_args;
typedef struct {                 /* xh_in_bwd_apply */
  const void* dy; long long dy_bs; const void* x; long long x_bs; void* dx; long long dx_bs; int N, C; long long DHW; const double* red;
  const float *mean, *rstd; int stat_rs, have_g; const float *sc, *sh; float slope; int accumulate;
} xh_in_bwd_apply_args;
typedef struct {                 /* xh_upsample2x_in_act_fwd */
  const void* x; long long x_bs; void* y; long long y_bs; int N, C, D, H, W; const double* red; float slope; float *sc, *sh, *mean, *rstd;
} xh_upsample2x_in_act_args;
typedef struct {                 /* xh_upsample2x_bwd_act_reduce */
  const void* dy; long long dy_bs; void* dx; long long dx_bs; int N, C, D, H, W; const void* y0; long long y0_bs; const float *sc, *sh;
  float slope; double* red;
} xh_upsample2x_bwd_act_reduce_args;
/* n <= XH_LEVELS_MAX k = 1 convolutions (xh_conv3d_fwd with k == 1: same descriptors) in one launch; all of one storage type, one
 * epilogue (epi 0 or 2), no activation epilogue.  Returns 1 with nothing launched when that does not hold. */
int xh_conv1x1_multi(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p);
int xh_in_affine_act_multi(void* stream, int dtype, int n, const xh_in_affine_act_args* a);
int xh_act_bwd_reduce_multi(void* stream, int dtype, int n, const xh_act_bwd_reduce_args* a);
int xh_in_bwd_apply_multi(void* stream, int dtype, int n, const xh_in_bwd_apply_args* a);
int xh_upsample2x_in_act_multi(void* stream, int dtype, int n, const xh_upsample2x_in_act_args* a);
int xh_upsample2x_bwd_act_reduce_multi(void* stream, int dtype, int n, const xh_upsample2x_bwd_act_reduce_args* a);

/* generic elementwise helpers */
/* y = a + b  (b may be NULL: copy) with independent batch strides */
int xh_add(void* stream, int dtype, const void* a, long long a_bs, const void* b, long long b_bs, void* y,
           long long y_bs, int N, long long CDHW);
/* dx = dy * f'(y) for y = act(x): act RELU uses y>0, SIGMOID uses y(1-y) */
int xh_act_bwd(void* stream, int dtype, const void* dy, const void* y, void* dx, long long n, int act);

/* ------------------------------------------------------------------------------------------------
 * S-MVAE: product of experts + reparameterisation (buildingblocks.py:853-886, RA_HVED.py:573-597,741-753).
 * feat: the 4 DRB outputs stacked along channels: [N][4][2L][dhw] (modality m, first L = mu, last L = logvar).
 * keep: [N][4] floats (1 = modality present).  eps: [N][L][dhw] noise or NULL (valid=True).
 * Outputs: z [N][L][dhw]; mu_stack, lv_stack [N][5][L][dhw] (index 0 = prior, logvar clipped to +-50;
 * mask_mu=1 zeroes dropped modalities' mu like ProductOfExperts2's in-place ZeroLayerF).
 * ------------------------------------------------------------------------------------------------ */
int xh_poe_fwd(void* stream, int dtype, const void* feat, const float* keep, const void* eps, void* z,
               void* mu_stack, void* lv_stack, int N, int L, long long dhw, int mask_mu);
/* dfeat = d/dfeat of (z, mu_stack, lv_stack); dmu_stack/dlv_stack may be NULL. */
int xh_poe_bwd(void* stream, int dtype, const void* feat, const float* keep, const void* eps, const void* dz,
               const void* dmu_stack, const void* dlv_stack, void* dfeat, int N, int L, long long dhw, int mask_mu);
/* The PoE of several latent levels in ONE launch per direction (the levels of a forward pass are independent): job i carries what
 * xh_poe_fwd (bwd = 0: feat, keep, eps, z, mu_stack, lv_stack) / xh_poe_bwd (bwd = 1: + dz, dmu_stack, dlv_stack, dfeat) take for
 * level i.  Job array in host memory, read during the call. */
#define XH_POE_MAX 8
#define XH_RNG_LINES 32
#define XH_RNG_WORDS (16 * (2 + XH_RNG_LINES))
typedef struct {
  const void* feat; const float* keep; const void* eps;
  void *z, *mu_stack, *lv_stack;
  const void *dz, *dmu_stack, *dlv_stack; void* dfeat;
  long long dhw; int N, L, mask_mu;
  /* In-kernel reparameterisation noise (RA_HVED.py:741-747 draws eps ~ N(0,1) in fp32 with normal_()): when eps == NULL and
   * rng_used != NULL, element i of this job takes eps_i = the standard normal of Philox4x32-10 keyed by
   * (seed, *rng_used, rng_stream, i) -- fp32 whatever the storage type, no noise tensor in HBM, no generator launch.
   * rng_used: TWO 64-bit device words {draw counter, seed} of this forward; the backward call passes the same words and
   * regenerates the same eps.  See xh_poe_multi for who writes them. */
  const unsigned long long* rng_used; int rng_stream;
} xh_poe_job;
/* rng (optional, forward only; required when a job has rng_used and no eps): the caller's generator state, XH_RNG_WORDS 64-bit
 * device words: {seed, counter, 14 unused, then 1 + XH_RNG_LINES ticket words on cache lines of their own}, all but the first two
 * zero on entry.  Every workgroup of the launch reads `counter` when it starts and takes a ticket when it is done;
 * the last workgroup to finish copies {counter, seed} into every job's rng_used words, advances `counter` by one and leaves the ticket words zero -- so
 * each launch (each replay of a captured graph included) draws fresh noise with no host involvement, and launches ordered on a
 * stream may share one state.  In a backward call rng is NULL: the jobs' rng_used words already hold their forward's counter and seed. */
int xh_poe_multi(void* stream, int dtype, int bwd, int n, const xh_poe_job* jobs, unsigned long long* rng);
/* out[i] = the standard normal xh_poe_multi draws for element i of a job with (seed, counter, rng_stream) (raw = 0: n fp32 values),
 * or the four raw Philox4x32-10 output words of counter block i (raw = 1: 4 n uint32 values).  Test / inspection aid. */
int xh_philox_normal(void* stream, unsigned long long seed, unsigned long long counter, int rng_stream, void* out, long long n, int raw);

/* ------------------------------------------------------------------------------------------------
 * Channel attention glue.
 * ------------------------------------------------------------------------------------------------ */
/* ChannelPool (buildingblocks.py:136-138): y[:,0] = max_c x, y[:,1] = mean_c x, written at y (2 channels). */
int xh_channel_pool_fwd(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs, int N,
                        int C, long long DHW);
/* dx (+)= dy0 * [c == first argmax] + dy1 / C */
int xh_channel_pool_bwd(void* stream, int dtype, const void* x, long long x_bs, const void* dy, long long dy_bs,
                        void* dx, long long dx_bs, int N, int C, long long DHW, int accumulate);
/* y = x * (1 + s[n,0,p])  (AttenModule2 scaling buildingblocks.py:287,297; skip-return x_i = a*x_i + x_i,
 * RA_HVED.py:552).  s has 1 channel. */
int xh_gate_fwd(void* stream, int dtype, const void* x, long long x_bs, const void* s, long long s_bs, void* y,
                long long y_bs, int N, int C, long long DHW);
/* dx (+)= dy*(1+s);  ds (+)= sum_c dy*x */
/* AttenModule2's two pooled / gated tensors in one launch each (buildingblocks.py:279-299): a = the upsampled seg feature (Ca
 * channels), b = the encoder feature (Cb).  channel_pool2: y (N, 4, ..) = [max_c a, mean_c a, max_c b, mean_c b]; gate2: y (N,
 * Ca + Cb, ..) = [a (1 + E[:,0]) | b (1 + E[:,1])], E (N, 2, ..); the backward forms take an accumulate flag per input gradient
 * (1: += into a buffer that already holds another consumer's share) and gate2_bwd writes dE (N, 2, ..). */
int xh_channel_pool2_fwd(void* stream, int dtype, const void* xa, long long xa_bs, int Ca, const void* xb, long long xb_bs, int Cb,
                         void* y, long long y_bs, int N, long long DHW);
int xh_channel_pool2_bwd(void* stream, int dtype, const void* xa, long long xa_bs, int Ca, const void* xb, long long xb_bs, int Cb,
                         const void* dy, long long dy_bs, void* dxa, long long dxa_bs, int acc_a, void* dxb, long long dxb_bs, int acc_b,
                         int N, long long DHW);
/* red (may be NULL): [N][Ca + Cb][2] fp64, zeroed by the caller: the channel sums (sum y, sum y^2) of the stored output, left
 * by the same pass for the InstanceNorm of the conv that follows (no xh_moments launch). */
int xh_gate2_fwd(void* stream, int dtype, const void* xa, long long xa_bs, int Ca, const void* xb, long long xb_bs, int Cb, const void* E,
                 long long E_bs, void* y, long long y_bs, int N, long long DHW, double* red);
int xh_gate2_bwd(void* stream, int dtype, const void* xa, long long xa_bs, int Ca, const void* xb, long long xb_bs, int Cb, const void* E,
                 long long E_bs, const void* dy, long long dy_bs, void* dxa, long long dxa_bs, int acc_a, void* dxb, long long dxb_bs,
                 int acc_b, void* dE, long long dE_bs, int N, long long DHW, int sig_bwd);
/* (sig_bwd = 1: E is the output of a sigmoid -- AttenModule2's gates, buildingblocks.py:286,296 -- and dE receives the gradient of its
 * PRE-activation, dE * E (1 - E): the sigmoid's backward rides in this pass instead of a launch of its own.) */
/* Gate + MaxPool3d(2) in one pass (RA_HVED.py:552 followed by buildingblocks.py:655-657): y = maxpool2(x * (1 + s)), the gated
 * values rounded to the storage type before the maximum (bit-identical to xh_gate_fwd + xh_maxpool2_fwd); red (optional):
 * red[n][c][0..1] += (sum y, sum y^2) of the stored output for the InstanceNorm that follows.  D, H even, W a multiple of 8,
 * batch strides multiples of 8.  fwd with s == NULL: a plain MaxPool3d(2) that leaves the channel sums (bit-identical to
 * xh_maxpool2_fwd + xh_moments).  bwd: dx = dy routed to the first maximum of its window, times (1 + s); ds = sum over channels
 * of the routed dy times x (s has one channel). */
int xh_gate_maxpool_fwd(void* stream, int dtype, const void* x, long long x_bs, const void* s, long long s_bs, void* y, long long y_bs,
                        int N, int C, int D, int H, int W, double* red, int Cg);
/* Cg: the gate applies to channels [0, Cg) only (0 or > C: all) -- the skip stream of the skip-return path rides through the
 * modality streams' pooling launch ungated (RA_HVED.py:552 gates the four streams, :621 pools the skip stream without a gate);
 * its channels take no part in ds. */
int xh_gate_maxpool_bwd(void* stream, int dtype, const void* x, long long x_bs, const void* s, long long s_bs, const void* dy,
                        long long dy_bs, void* dx, long long dx_bs, void* ds, long long ds_bs, int N, int C, int D, int H, int W,
                        int acc_dx /* 1: dx += (another consumer's gradient share is already there) */, int Cg);
int xh_gate_bwd(void* stream, int dtype, const void* x, long long x_bs, const void* s, long long s_bs,
                const void* dy, long long dy_bs, void* dx, long long dx_bs, void* ds, long long ds_bs, int N,
                int C, long long DHW, int acc_dx, int acc_ds);

/* DuSEAttention gating (modules/DuSFE.py:130-153): u = x * (1 + ch[n,c] + sp[n,0,p]); ch = channel gate
 * (already sigmoid'ed, fp32 [N][C]), sp = spatial gate (1 channel). */
int xh_duse_gate_fwd(void* stream, int dtype, const void* x, long long x_bs, const float* ch, const void* sp,
                     long long sp_bs, void* u, long long u_bs, int N, int C, long long DHW);
/* The same pass, also leaving the channel sums of u (red[N][C][2] += {sum u, sum u^2}, the caller zeroes red) for the
 * BatchNorm3d that follows (modules/DuSFE.py:151-153): replaces a separate xh_moments pass over u. */
int xh_duse_gate_fwd_stats(void* stream, int dtype, const void* x, long long x_bs, const float* ch, const void* sp,
                           long long sp_bs, void* u, long long u_bs, int N, int C, long long DHW, double* red);
/* dx = du*(1+ch+sp); dsp = sum_c du*x; dch[n][c] += sum_p du*x (double accum buffer red1 [N][C]).  sigmoid_bwd = 1: dsp is stored
 * times sp*(1-sp), i.e. as the gradient in front of the sigmoid that produced sp (modules/DuSFE.py:113-155) -- only where
 * xh_duse_gate_bwd_fuses(C) says the one-pass kernel takes the channel count (4 / 8 / 16), XH_ERR_ARG otherwise. */
int xh_duse_gate_bwd(void* stream, int dtype, const void* x, long long x_bs, const float* ch, const void* sp,
                     long long sp_bs, const void* du, long long du_bs, void* dx, long long dx_bs, void* dsp,
                     long long dsp_bs, double* dch, int N, int C, long long DHW, int sigmoid_bwd);
int xh_duse_gate_bwd_fuses(int C);

/* Skip-return attention tail (sa_module.py:133-135 + attention_blocks.py:119-126):
 * r = relu(relu(t*sc+sh) + x); a = sigmoid(w0*max_c r + w1*mean_c r).  a has 1 channel. */
int xh_skr_tail_fwd(void* stream, int dtype, const void* t, const void* x, const float* sc, const float* sh,
                    const float* w2, void* a, int N, int C, long long DHW);
/* xh_skr_tail_fwd with the training-mode BatchNorm3d in front of the tail (sa_modules/sa_module.py:79-85) finalised in the same
 * launch, one sample, C <= 64: red [C][2] = raw sums (sum t, sum t^2) of t left by the conv epilogue; sc / sh / mean / rstd [C] are
 * written for the backward pass, running_mean / running_var advanced by `steps` momentum-0.1 updates (unbiased variance). */
int xh_skr_tail_bn_fwd(void* stream, int dtype, const void* t, const void* x, const double* red, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int steps, const float* w2, void* a, int C, long long DHW, float* sc,
                       float* sh, float* mean, float* rstd);
/* Given da: dtg = gradient w.r.t. the BatchNorm output t*sc+sh (already multiplied by both relu'),
 * dx (+)= gradient through the residual branch, and the gradient of the 1x1 conv's two weights ACCUMULATED into dw2[0..1]
 * (fp64) or, when dw2_f32 is given, into dw2_f32[0..1] (the parameter's fp32 gradient buffer; dw2 may then be NULL). */
int xh_skr_tail_bwd(void* stream, int dtype, const void* t, const void* x, const float* sc, const float* sh,
                    const float* w2, const void* a, const void* da, void* dtg, void* dx, double* dw2, int N, int C,
                    long long DHW, int acc_dx, float* dw2_f32);

/* dx += w[c]*d[n,0,p] + k[n,c]: rank-1 data gradient of a C->1 1x1 conv plus a per-(n,c) constant (the
 * global-average-pool gradient) -- finishes DuSEAttention's input gradient (modules/DuSFE.py:118,135-140). */
int xh_rank1_add(void* stream, int dtype, void* dx, long long dx_bs, const void* d, long long d_bs, const float* w,
                 const float* k, int N, int C, long long DHW);
/* DuSEAttention's channel excitation (modules/DuSFE.py:113-133) inside the passes of the recon | seg PAIR (one sample; the pair
 * (1, 2C, ...) viewed as (2, C, ...)): xh_duse_gate_fc_fwd = xh_duse_fc_fwd + xh_duse_gate_fwd(_stats) in one launch -- every
 * workgroup derives its channel's gate from the pair's raw sums red_in [2C][2]; ch_out [2][C], g_out [C], means_out [2C] are left
 * for the backward pass.  xh_rank1_add_fc = xh_duse_fc_bwd + xh_rank1_add: dx[c] += w[c] * d + d(mean)[c] with the pooled-mean
 * gradient derived in-kernel from dch [2][C] (xh_duse_gate_bwd) and the saved ch / g / means; the first workgroup ACCUMULATES the
 * six parameter gradients of fc_comb / fc_ch1 / fc_ch2. */
int xh_duse_gate_fc_fwd(void* stream, int dtype, const void* x, long long x_bs, const void* sp, long long sp_bs, void* u, long long u_bs, int C,
                        long long DHW, const double* red_in, const float* wc, const float* bc, const float* w1, const float* b1,
                        const float* w2, const float* b2, float* ch_out, float* g_out, float* means_out, double* red_out);
int xh_rank1_add_fc(void* stream, int dtype, void* dx, long long dx_bs, const void* d, long long d_bs, const float* w, int C2, long long DHW,
                    const float* means, const float* g, const float* ch, const double* dch, const float* wc, const float* w1,
                    const float* w2, float* dwc, float* dbc, float* dw1, float* db1, float* dw2, float* db2);

/* tiny dense layers on pooled features (DuSFE.py:118-127): handled on device, fp32.
 * in: channel sums red_r, red_s [N][C][2] (fp64, from moments or a producer's epilogue); out: ch1, ch2 [N][C] (sigmoid'ed), the
 * saved g [N][C] and, when `means` is given, the pooled means [N][2C] for the backward call (which then needs no red_r / red_s:
 * the sums may live in scratch storage that is gone by then). */
int xh_duse_fc_fwd(void* stream, const double* red_r, const double* red_s, long long count, int N, int C,
                   const float* w_comb, const float* b_comb, const float* w1, const float* b1, const float* w2,
                   const float* b2, float* g, float* ch1, float* ch2, float* means);
/* means: what xh_duse_fc_fwd left (red_r / red_s are ignored and may be NULL), or NULL: the means are taken from red_r, red_s. */
int xh_duse_fc_bwd(void* stream, const double* red_r, const double* red_s, long long count, int N, int C,
                   const float* w_comb, const float* w1, const float* w2, const float* g, const float* ch1,
                   const float* ch2, const double* dch1, const double* dch2, float* dw_comb, float* db_comb,
                   float* dw1, float* db1, float* dw2, float* db2, float* dmean_r, float* dmean_s, const float* means);

/* ------------------------------------------------------------------------------------------------
 * ViL / mLSTM (UxLSTMEnc_3d.py:42-87, vision_lstm.py:48-506).  All fp32.  Token t = flattened (d,h,w),
 * W fastest; B = batch, S tokens, C model dim, inner = 2C, NH heads of DH = inner/NH.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* norm_w;      /* [C]        vil.norm.weight (LayerNorm weight is 1+w, no bias) */
  const float* proj_up;     /* [4C][C]    */
  const float* conv_w;      /* [2C][1][4] causal depthwise conv1d */
  const float* conv_b;      /* [2C] */
  const float* q_w; const float* k_w; const float* v_w;   /* [2C/4][4][4] block-diagonal */
  const float* ig_w; const float* ig_b;                   /* [NH][6C], [NH] */
  const float* fg_w; const float* fg_b;
  const float* outnorm_w;   /* [2C] */
  const float* skip;        /* [2C] learnable_skip */
  const float* proj_down;   /* [C][2C] */
} xh_vil_params;

typedef struct {            /* same layout, gradients (ACCUMULATED, caller zeroes) */
  float* norm_w; float* proj_up; float* conv_w; float* conv_b; float* q_w; float* k_w; float* v_w;
  float* ig_w; float* ig_b; float* fg_w; float* fg_b; float* outnorm_w; float* skip; float* proj_down;
} xh_vil_grads;

/* workspace size in floats for B,S,C (forward saves what backward needs inside it) */
long long xh_vil_workspace_floats(int B, int S, int C);
/* t = xa (+ xb);  out = (add_xa ? xa : 0) + ViLBlock(t), ViLBlock(t) = t + layer(norm(t)).
 * xa/xb/out are NCDHW tensors [B][C][S] of storage `dtype` (xb may be NULL).  add_xa=1 is the bottleneck use
 * rec0 + mViL(rec0 + skip) (RA_HVED.py:626); add_xa=0 is DoubleConv_ViL (buildingblocks.py:554-555). */
int xh_vil_fwd(void* stream, int dtype, const void* xa, const void* xb, void* out, int B, int S, int C, int NH,
               int add_xa, const xh_vil_params* p, float* ws);
/* dxin = dout * dViLBlock/dt (the gradient w.r.t. t = xa+xb; the caller adds dout for xa when add_xa). */
int xh_vil_bwd(void* stream, int dtype, const void* xa, const void* xb, const void* dout, void* dxin, int B,
               int S, int C, int NH, const xh_vil_params* p, const xh_vil_grads* g, float* ws);

/* ------------------------------------------------------------------------------------------------
 * Parameter compositions (fp32, parameter-sized).  Linear stages with no non-linearity between them are applied as ONE
 * conv with composed weights; these build the composed weights and scatter their gradients back (accumulating, +=).
 *  - AttenModule2 (buildingblocks.py:271-274,283-296): grouped k^3 conv (E outputs per input channel) followed by a
 *    1x1 conv to one channel, for the seg gate (NS pooled channels) and the enc gate (NE pooled channels):
 *    w[2][NE][K3] (row 0 = seg gate, zero for channels >= NS), b[2].
 *  - DuSEAttention (modules/DuSFE.py:129-144): conv_comb o [conv_squeeze_ch1|conv_squeeze_ch2] -> sqw[2C], sqb[1], and
 *    conv_adjust_ch1/ch2 stacked -> adjw[2][27], adjb[2].  params/grads order: comb_w, comb_b, sq1_w, sq1_b, sq2_w, sq2_b,
 *    adj1_w, adj1_b, adj2_w, adj2_b. */
int xh_compose_atten_fwd(void* stream, const float* seg_w, const float* seg_b, const float* seg2_w, const float* seg2_b,
                         const float* enc_w, const float* enc_b, const float* enc2_w, const float* enc2_b, int NS, int NE,
                         int E, int K3, float* w, float* b);
int xh_compose_atten_bwd(void* stream, const float* seg_w, const float* seg_b, const float* seg2_w, const float* enc_w,
                         const float* enc_b, const float* enc2_w, int NS, int NE, int E, int K3, const float* gw,
                         const float* gb, float* d_seg_w, float* d_seg_b, float* d_seg2_w, float* d_seg2_b, float* d_enc_w,
                         float* d_enc_b, float* d_enc2_w, float* d_enc2_b);
/* All parameter compositions of a step in ONE launch per direction.  Job arrays are host memory, read during the call.
 *  xh_atten_job: p / g = the eight AttenModule2 parameters / their gradient buffers in the order of xh_compose_atten_fwd; forward
 *    writes w, b; backward reads gw, gb and ACCUMULATES into g.   xh_duse_job: p / g as xh_compose_duse_fwd; out = {sqw, sqb, adjw,
 *    adjb}, gout their gradients.   xh_head_job: the segmentation head final_conv o sfinals (RA_HVED.py:192-199,640): wf [Co][Cm],
 *    bf [Co], ws [Cm][Ci], bs [Cm] -> w [Co][Ci] = wf ws, b = wf bs + bf; backward accumulates dwf, dbf, dws, dbs from gw, gb. */
#define XH_COMPOSE_MAX 4
typedef struct { const float* p[8]; float* w; float* b; float* g[8]; const float* gw; const float* gb; int NS, NE, E, K3; } xh_atten_job;
typedef struct { const float* p[10]; float* out[4]; float* g[10]; const float* gout[4]; int C; } xh_duse_job;
typedef struct { const float *wf, *bf, *ws, *bs; float *w, *b; float *dwf, *dbf, *dws, *dbs; const float *gw, *gb; int Co, Cm, Ci; } xh_head_job;
/* xh_sep_job: a depthwise k^3 conv followed by a pointwise 1x1 conv with nothing in between (DWConvNorm, sa_modules/sa_module.py:79-85:
 *   dwconv -> pwconv -> norm) as ONE dense k^3 conv: w [C][C][K3] = pw[co][ci] * dw[ci][t] (the pointwise bias stays the dense conv's
 *   bias); backward reads gw [C][C][K3] and ACCUMULATES into g_dw [C][K3], g_pw [C][C]. */
#define XH_SEP_MAX 8
typedef struct { const float *dw, *pw; float* w; const float* gw; float *g_dw, *g_pw; int C, K3; } xh_sep_job;
/* zero_buf / zero_n: `zero_n` floats the same launch clears (the caller's gradient buffers of the composed tensors: the convs that
 * use them accumulate into those during the backward pass); NULL / 0: nothing. */
int xh_compose_multi(void* stream, int bwd, int na, const xh_atten_job* aj, int nd, const xh_duse_job* dj, int nh, const xh_head_job* hj,
                     int ns, const xh_sep_job* sj, float* zero_buf, long long zero_n);
int xh_compose_duse_fwd(void* stream, const float* const params[10], int C, float* sqw, float* sqb, float* adjw, float* adjb);
int xh_compose_duse_bwd(void* stream, const float* const params[10], int C, const float* dsqw, const float* dsqb,
                        const float* dadjw, const float* dadjb, float* const grads[10]);

/* ------------------------------------------------------------------------------------------------
 * Loss / metric epilogues of the training step (train.py:232-262,288-296): one pass per tensor pair.
 * ------------------------------------------------------------------------------------------------ */
/* red[n][c][0..5] += (sum a'b, sum a'^2, sum b^2, sum (a'-b)^2, sum a', sum b) over DHW, a' = a or (a > thr ? 1 : 0) when
 * thr_on.  b may be fp32 (b_dtype = XH_F32) while a is in a 16-bit storage type, or NULL (= the constant bval).
 * Serves DiceLoss (loss.py:257-285: slots 0,1,2), nn.MSELoss / GANLoss (train.py:173, loss.py:167-186: slot 3) and the
 * thresholded DiceCoefficient / DiceRegion metrics (metrics.py:27-107: slots 0,4,5).  red: fp64 [N][C][6], caller zeroes. */
int xh_pair_sums(void* stream, int dtype, const void* a, long long a_bs, int b_dtype, const void* b, long long b_bs, float bval,
                 int N, int C, long long DHW, int thr_on, float thr, double* red);
/* out[n,c,:] (+)= ca[n,c]*a + cb[n,c]*b + cc[n,c]: the backward of the losses above (their gradient w.r.t. a is linear in
 * a and b with per-(n,c) coefficients).  out has a's dtype; cc may be NULL. */
int xh_lincomb(void* stream, int dtype, const void* a, long long a_bs, int b_dtype, const void* b, long long b_bs, float bval,
               void* out, long long o_bs, int N, int C, long long DHW, const float* ca, const float* cb, const float* cc,
               const float* gscale, int accumulate);
/* Sums of xh_pair_sums -> scalar loss / per-channel metric + the per-(n,c) coefficients of xh_lincomb, on the device:
 * kind 0 DiceLoss, 1 mean squared difference (count = number of elements), 2 thresholded Dice metric (out[C]), 3 mean of a,
 * 4 weighted sum over the rows of red (slot 4) with the INPUT weights ca[N*C] (several means finished by one launch).
 * gscale (xh_lincomb, xh_kld_bwd): optional device scalar multiplying the result (the upstream gradient). */
int xh_loss_finalize(void* stream, int kind, const double* red, int N, int C, double count, double eps, float* out, float* ca,
                     float* cb);
/* compute_KLD (loss.py:29-40,85-115) for ONE modality subset on the (N,5,L,dhw) stacks xh_poe_fwd returns: product of
 * the kept experts and the prior, then KL(posterior || prior) summed over all latent voxels: red[0] += sum
 * (0.5 * mean is applied by the caller).  keep as in xh_poe_fwd.  bwd: d(scale * sum)/d(mu_stack, lv_stack). */
int xh_kld_fwd(void* stream, int dtype, const void* mu_stack, const void* lv_stack, const float* keep, int N, int L, long long dhw,
               double* red);
int xh_kld_bwd(void* stream, int dtype, const void* mu_stack, const void* lv_stack, const float* keep, int N, int L, long long dhw,
               float scale, const float* gscale, void* dmu_stack, void* dlv_stack);
/* Nested tumour-region weight map (train.py:244-248,252-256): from the 3 sigmoid channels (WT, TC, ET) of seg,
 * w = p0 > .5 ? p0 : 0, overridden by p1 where p1 > .5, then by p2 where p2 > .5.  w has 1 channel. */
int xh_nested_weight(void* stream, int dtype, const void* seg, long long seg_bs, void* w, long long w_bs, int N, long long DHW);
/* out[0..n) = v (* gscale[0] when given) in the storage type (constant upstream gradients of mean-type losses). */
int xh_fill(void* stream, int dtype, void* out, long long n, float v, const float* gscale);
/* Multi-tensor forms (at most XH_MULTI_MAX tensors of ONE storage type per launch; host arrays, read during the call):
 * xh_multi_sum: red[(row0[t] + b % rows[t]) * 6 + 4] += partial sums of tensor t (fp64; the slot layout of xh_pair_sums, so
 * xh_loss_finalize kind 4 turns the rows into sum_t w_t * sum(t)); rows[t] spreads a large tensor's workgroups over several
 * accumulator rows (at most 64 adders each).  xh_multi_fill: tensor t = values[t] (* gscale[0] when given).
 * Reference use: the SURVEY 8(d) benchmark loss seg.mean() + rec.mean() + sum_l (mu_l.mean() + logvar_l.mean()). */
#define XH_MULTI_MAX 16
int xh_multi_sum(void* stream, int dtype, int nt, const void* const* ptrs, const long long* numels, const int* row0, const int* rows,
                 double* red);
int xh_multi_fill(void* stream, int dtype, int nt, void* const* ptrs, const long long* numels, const float* values, const float* gscale);

/* Scalar glue of a loss (train.py:240,262,280: weighted sums of loss terms; loss.py:85-115 averaged over the latent levels at
 * train.py:235-239): out[0] = sum_i coef[i] * src[i][0] over n <= XH_SCALAR_MAX DEVICE scalars, each fp32 or fp64 (is_f64[i]);
 * host arrays are read during the call.  xh_scalar_fanout is its backward: out32[i] = out64[i] = coef[i] * g[0]. */
#define XH_SCALAR_MAX 16
int xh_scalar_lincomb(void* stream, int n, const void* const* src, const int* is_f64, const double* coef, float* out);
int xh_scalar_fanout(void* stream, int n, const double* coef, const float* g, float* out32, double* out64);

/* ------------------------------------------------------------------------------------------------
 * Discriminator (RA_HVED.py:204-236, buildingblocks.py:342-358): convolutions 7 -> 64 -> 128 -> 256 -> 512 -> 1 with kernel
 * size ks = 4 (train.py:146, Pretrain.py:150) or 3 (the class default), padding 1, strides 1,2,2,2,1, as implicit GEMMs on the
 * matrix cores.  Inside the discriminator activations are CHANNELS-LAST [N][D][H][W][C], 16-bit (dtype XH_BF16 / XH_F16),
 * C a multiple of 8; weights are re-packed per step by xh_dconv_pack.  Every entry point returns XH_ERR_ARG for ks outside
 * {3, 4} and for extents that are not those of a padding-1 convolution: out = (in + 2 - ks) / stride + 1.
 * ------------------------------------------------------------------------------------------------ */
/* Exact fp32 route of the Discriminator (parity mode; buildingblocks.py:350,354: nn.Conv3d(c, 2c, ks, stride, padding=1)): direct
 * NCDHW fp32 convolutions for any kernel size 1..7 with padding 1, stride 1 | 2.  mode 0: y = conv(x, w) + b (b optional);
 * mode 1: data gradient, x = dY (N, Cout, Do..), y = dX (N, Cin, D..) written; mode 2: weight / bias gradient, x = input, y = dY,
 * w = dW and b = dB (optional) ACCUMULATED into (the const qualifiers of w / b are cast away in this mode).  ~1 TFLOP/s: for small
 * patches and tests; the matrix-core entry points below are the product path. */
int xh_dconv_exact(void* stream, int mode, const float* x, const float* w, const float* b, float* y, int N, int Cin, int Cout, int D, int H,
                   int W, int Do, int Ho, int Wo, int ks, int stride);
/* mode 0 forward: x [N][Di,Hi,Wi][Cs] -> y [N][Do,Ho,Wo][Cn], Do = (Di+2-ks)/stride+1; Cs a multiple of 32, or 8 (the padded
 * 7-channel input: weights packed with mode 2).  bias [Cn] optional; act XH_ACT_NONE / XH_ACT_LRELU(slope); red optional:
 * red[n][cn][0..1] += (sum y, sum y^2) of the stored output (InstanceNorm statistics).
 * mode 1 data gradient: x = dY [N][Di..][Cs = Cout(_pad)], y = dX [N][Do..][Cn = Cin(_pad)], Di = (Do+2-ks)/stride+1, weights
 * packed with mode 1; bias / act unused.
 * mask (optional, [N][Do..][Cn] like y): the result is multiplied by (mask > 0 ? 1 : slope) before it is rounded, stored and
 * summed into red -- the LeakyReLU backward of the layer below (mask = its stored activation; leaky keeps the sign) fused into
 * the data gradient, with red[..][0] then carrying that layer's bias gradient. */
int xh_dconv_cl(void* stream, int dtype, int mode, int stride, int ks, const void* x, const void* w, const float* bias, void* y,
                double* red, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int Cs, int Cn, int act, float slope,
                const void* mask);
/* dwp[ks^3][Cn][Cs] (fp32, caller zeroes) += sum over output voxels m of dY[m][cn] * X[src(m, tap)][cs]; x = the forward input
 * [N][Di..][Cs], dy [N][Do..][Cn]; Cs, Cn multiples of 8. */
int xh_dconv_wgrad_cl(void* stream, int dtype, int stride, int ks, const void* x, const void* dy, float* dwp, int N, int Di, int Hi,
                      int Wi, int Do, int Ho, int Wo, int Cs, int Cn);
/* The discriminator's head, `last` = Conv3d(C, 1, ks, stride 1, padding 1, bias=False) (RA_HVED.py:223): ONE output channel is a
 * C * ks^3-long dot product per voxel, a reduction rather than a GEMM (as a 16-column MFMA tile the three passes ran at 0.002 of the
 * matrix peak: 37-97 us each, latency of a few workgroups).  x [N][Di..][C] channels-last 16-bit, C a multiple of 512 (a wave's 64
 * lanes x 8 channels); wp = the mode-0 image of the weight, [ks^3][C] 16-bit (xh_dconv_pack with Cout = 1); y / dy [N][Do..] 16-bit.
 *   xh_dlast_fwd:   y[o] = sum_{tap, c} wp[tap][c] x[o + tap - 1][c]                       (fp32 accumulation)
 *   xh_dlast_dgrad: dx[i][c] = sum_tap dy[i + 1 - tap] wp[tap][c]
 *   xh_dlast_wgrad: dw[0][c][tap] += scale * sum_{n, o} dy[n][o] x[n][o + tap - 1][c]      (dw: the fp32 PARAMETER layout [1][C][ks^3],
 *                   accumulated in place: no packed temporary, no unpack launch) */
int xh_dlast_fwd(void* stream, int dtype, int ks, const void* x, const void* wp, void* y, int N, int Di, int Hi, int Wi, int Do, int Ho,
                 int Wo, int C);
int xh_dlast_dgrad(void* stream, int dtype, int ks, const void* dy, const void* wp, void* dx, int N, int Di, int Hi, int Wi, int Do, int Ho,
                   int Wo, int C);
int xh_dlast_wgrad(void* stream, int dtype, int ks, const void* x, const void* dy, float* dw, float scale, int N, int Di, int Hi, int Wi,
                   int Do, int Ho, int Wo, int C);
/* fp32 nn.Conv3d weight [Cout][Cin][ks^3] -> 16-bit operand image.  mode 0: [ks^3][Cout][CinPad]; mode 1: [ks^3][CinPad][CoutPad];
 * mode 2 (CinPad == 8): [ks^2][Cout][32] (k = kw * 8 + ci).  xh_dconv_unpack_grad: dw[Cout][Cin][ks^3] += dwp[ks^3][CoutPad][CinPad]. */
int xh_dconv_pack(void* stream, int dtype, int mode, int ks, const float* w, void* out, int Cout, int Cin, int CoutPad, int CinPad);
int xh_dconv_unpack_grad(void* stream, int ks, const float* dwp, float* dw, int Cout, int Cin, int CoutPad, int CinPad);
/* NCDHW (xa: CA channels, xb: CB channels or NULL) -> channels-last [N][V][Cpad] (zero padded), and its adjoint. */
int xh_cl_from_ncdhw(void* stream, int dtype, const void* xa, long long xa_bs, int CA, const void* xb, long long xb_bs, int CB, void* out,
                     int Cpad, int N, long long V);
int xh_cl_to_ncdhw(void* stream, int dtype, const void* g, int Cpad, void* da, long long da_bs, int CA, void* db, long long db_bs, int CB,
                   int N, long long V);
/* y = leaky(x * sc[n,c] + sh[n,c], slope) on channels-last tensors (InstanceNorm3d + LeakyReLU(0.2) of discriminator_block). */
int xh_cl_affine_act(void* stream, int dtype, const void* x, void* y, const float* sc, const float* sh, float slope, int N, int C,
                     long long V);
/* g = dy * leaky'(x*sc + sh) (sc/sh NULL: 1/0).  mode 0: red[n][c][0..1] += (sum g, sum g*x); mode 1: dx = A*g + Cc*x + B
 * (coefficients from xh_norm_bwd_coef); mode 2: dx = g, red[n][c][0] += sum g (bias gradient). */
int xh_cl_act_bwd(void* stream, int dtype, int mode, const void* dy, const void* x, void* dx, const float* sc, const float* sh, float slope,
                  const float* A, const float* B, const float* Cc, double* red, int N, int C, long long V);

#ifdef __cplusplus
}
#endif
#endif
