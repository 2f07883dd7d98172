// Bandwidth-bound stages of the XLSTM-HVED path: normalisation statistics and their backward, pooling,
// trilinear resampling, product-of-experts + reparameterisation, attention gates.
// One lane handles 4 consecutive voxels of one (n, c) row (16 B fp32 / 8 B bf16 accesses) when the row
// length and strides allow it, otherwise a scalar tail path.  Reductions: fp32 in the lane, fp32 across
// the block, fp64 atomics across blocks (few workgroups per address: red_grid below).
#include "common.h"
#include "conv_pack.h"
#include "../../include/xlstm_hved.h"

#define EW_BLOCK 256
// voxels per lane: 16-byte accesses for both storage types (4 fp32 or 8 x 16-bit)

// VEC: the layout allows 16-byte accesses (vec_ok below) -- a compile-time flag, so the wide path is straight-line code
// whose loads the scheduler can issue back to back (a run-time flag puts every load in its own branch with the wait for
// it right behind).
template <bool VEC, typename T, int N>
__device__ __forceinline__ void ldrow(const T* p, long long q, int valid, float (&o)[N]) {
  if constexpr (VEC) {
    ldvec(p, q, o);
  } else {
#pragma unroll
    for (int v = 0; v < N; ++v) o[v] = v < valid ? ldf(p, q + v) : 0.f;
  }
}
template <bool VEC, typename T, int N>
__device__ __forceinline__ void strow(T* p, long long q, int valid, const float (&o)[N]) {
  if constexpr (VEC) {
    stvec(p, q, o);
  } else {
#pragma unroll
    for (int v = 0; v < N; ++v)
      if (v < valid) stf(p, q + v, o[v]);
  }
}
template <typename T>
static inline bool vec_ok(long long dhw, std::initializer_list<long long> strides) {
  constexpr int VW = VWT<T>::v;
  if (dhw % VW) return false;
  for (long long s : strides)
    if (s % VW) return false;
  return true;
}
// ~2048 workgroups per launch whatever the channel count: few workgroups per (n,c) row when there are many rows (so
// the per-workgroup fp64 atomics of the reducing kernels do not pile up on one address), many when there are few.
int g_row_wgs = 2048;   // xh_set_option(27, n): workgroup target of the row-streaming kernels (experiments)
template <typename T>
static inline dim3 row_grid(long long dhw, int C, int N) {
  const long long maxb = (dhw + EW_BLOCK * VWT<T>::v - 1) / (EW_BLOCK * VWT<T>::v);
  long long want = (g_row_wgs + (long long)C * N - 1) / ((long long)C * N);
  if (want < 1) want = 1;
  return dim3((unsigned)(want < maxb ? want : maxb), C, N);
}
// Kernels that end in fp64 atomics on one address per (n, c) row (moments, act_bwd_reduce, duse_gate_bwd_row): the
// device-scope atomics of one address serialise across the 8 XCDs at ~50 ns each, so 512 workgroups per row cost 28 us
// whatever the row holds (4 channels @128^3: 27.5 us with 512 per row, 8.6 us with 64).  At most 64 workgroups per row,
// ~1024 per launch, and no workgroup below 16 KB (measured optimum at every shape of the step, tools/microbench_small.py
// --reducers).
int g_red_wgs = 0;      // xh_set_option(9, n): overrides the ~1024 workgroup target (experiments)
template <typename T>
static inline dim3 red_grid(long long dhw, int C, int N) {
  const long long maxb = (dhw + EW_BLOCK * VWT<T>::v - 1) / (EW_BLOCK * VWT<T>::v);
  const long long rows = (long long)C * N;
  long long want = ((g_red_wgs > 0 ? g_red_wgs : 1024) + rows - 1) / rows;
  if (g_red_wgs <= 0) {
    const long long by_bytes = dhw * (long long)sizeof(T) / (16 * 1024);
    if (want > 64) want = 64;
    if (want > by_bytes) want = by_bytes;
  }
  if (want < 1) want = 1;
  return dim3((unsigned)(want < maxb ? want : maxb), C, N);
}

#define ROW_LOOP_BEGIN                                                                         \
  constexpr int VW = VWT<T>::v;                                                                \
  const int c = blockIdx.y, n = blockIdx.z;                                                    \
  const long long q_per = ((dhw + gridDim.x - 1) / gridDim.x + EW_BLOCK * VW - 1) / (EW_BLOCK * VW) * (EW_BLOCK * VW); \
  const long long q_end = min(dhw, (long long)(blockIdx.x + 1) * q_per);                       \
  for (long long q = (long long)blockIdx.x * q_per + threadIdx.x * VW; q < q_end; q += EW_BLOCK * VW) { \
    const int valid = (int)min((long long)VW, dhw - q);
#define ROW_LOOP_END }
// The reducing kernels run with at most 64 workgroups per row (red_grid), i.e. a thread walks 8 - 16 vectors: four of them per trip,
// their loads issued together (one load per trip is one exposed latency per trip on a launch with one workgroup per CU).
// ROW_LOOP4_BEGIN opens the four-vector trips (q, u = 0..3 inside the caller's loops over u; S = the trip's vector stride),
// ROW_LOOP4_TAIL the single-vector trips behind them.
#define ROW_LOOP4_BEGIN                                                                        \
  constexpr int VW = VWT<T>::v;                                                                \
  const int c = blockIdx.y, n = blockIdx.z;                                                    \
  const long long q_per = ((dhw + gridDim.x - 1) / gridDim.x + EW_BLOCK * VW - 1) / (EW_BLOCK * VW) * (EW_BLOCK * VW); \
  const long long q_end = min(dhw, (long long)(blockIdx.x + 1) * q_per);                       \
  constexpr long long S = (long long)EW_BLOCK * VW;                                            \
  long long q = (long long)blockIdx.x * q_per + threadIdx.x * VW;                              \
  for (; q + 3 * S < q_end; q += 4 * S) {
#define ROW_LOOP4_TAIL }                                                                       \
  for (; q < q_end; q += S) {                                                                  \
    const int valid = (int)min((long long)VW, dhw - q);

// ---------------------------------------------------------------------------------------- moments
// xb != nullptr: a virtual concat (xa | xb): channels >= ca come from xb (xh_moments2: the decoder's torch.cat input in ONE launch)
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void moments_kernel(const T* x, long long x_bs, long long dhw, double* red,
                                                          long long red_rs, const T* xb, long long xb_bs, int ca) {
  // Per-lane sums in fp64: the statistics feed var = E[x^2] - mean^2, which cancels mean^2/var digits, and the network
  // amplifies any error in them ~1e4x; fp64 adds are free next to the HBM stream.
  __shared__ double s_red[4 * 2];
  double s[2] = {0.0, 0.0};
  const T* xp;
  {
    const int c = blockIdx.y, n = blockIdx.z;
    xp = (xb && c >= ca) ? xb + n * xb_bs + (long long)(c - ca) * dhw : x + n * x_bs + (long long)c * dhw;
  }
  ROW_LOOP4_BEGIN
    float v4[4][VW];
#pragma unroll
    for (int u = 0; u < 4; ++u) ldrow<VEC>(xp, q + u * S, (int)min((long long)VW, dhw - (q + u * S)), v4[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t0 = 0.f, t1 = 0.f;                         // one vector's worth in fp32, then folded into the fp64 sums
#pragma unroll
      for (int i = 0; i < VW; ++i) { t0 += v4[u][i]; t1 = fmaf(v4[u][i], v4[u][i], t1); }
      s[0] += (double)t0;
      s[1] += (double)t1;
    }
  ROW_LOOP4_TAIL
    float v[VW];
    ldrow<VEC>(xp, q, valid, v);
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int i = 0; i < VW; ++i) { t0 += v[i]; t1 = fmaf(v[i], v[i], t1); }
    s[0] += (double)t0;
    s[1] += (double)t1;
    (void)c; (void)n;
  ROW_LOOP_END
  block_sum_d<2>(s, s_red, EW_BLOCK >> 6);
  if (threadIdx.x < 2) atomicAdd(&red[blockIdx.z * red_rs + blockIdx.y * 2 + threadIdx.x], s_red[threadIdx.x]);
}

static int launch_moments(void* stream, int dtype, const void* x, long long x_bs, const void* xb, long long xb_bs, int ca, int N, int C,
                          long long DHW, double* red, long long red_rs) {
  if (!x || !red || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  const bool vec32 = vec_ok<float>(DHW, {x_bs, xb_bs}), vec16 = vec_ok<bf16_t>(DHW, {x_bs, xb_bs});
  const dim3 grid32 = red_grid<float>(DHW, C, N), grid16 = red_grid<bf16_t>(DHW, C, N);
#define MO(T, V, G) hipLaunchKernelGGL((moments_kernel<T, V>), G, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)x, x_bs, DHW, red, red_rs, (const T*)xb, xb_bs, ca)
  if (dtype == XH_F32) { if (vec32) MO(float, true, grid32); else MO(float, false, grid32); }
  else if (dtype == XH_BF16) { if (vec16) MO(bf16_t, true, grid16); else MO(bf16_t, false, grid16); }
  else if (dtype == XH_F16) { if (vec16) MO(f16_t, true, grid16); else MO(f16_t, false, grid16); }
  else return XH_ERR_DTYPE;
#undef MO
  return xh_launch_status();
}
extern "C" int xh_moments(void* stream, int dtype, const void* x, long long x_bs, int N, int C, long long DHW,
                          double* red, long long red_rs) {
  return launch_moments(stream, dtype, x, x_bs, nullptr, 0, C, N, C, DHW, red, red_rs);
}
extern "C" int xh_moments2(void* stream, int dtype, const void* xa, long long xa_bs, int CA, const void* xb, long long xb_bs, int CB,
                           int N, long long DHW, double* red, long long red_rs) {
  if (!xb || CA <= 0 || CB <= 0) return XH_ERR_ARG;
  return launch_moments(stream, dtype, xa, xa_bs, xb, xb_bs, CA, N, CA + CB, DHW, red, red_rs);
}

// ---------------------------------------------------------------------------------------- norm finalize
__global__ void norm_finalize_kernel(int mode, const double* red, int N, int C, long long count, int gs, float eps,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     int steps, float* sc, float* sh, float* mean_o, float* rstd_o) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  double mean, var;
  if (mode == 0) {
    mean = red[i * 2] / (double)count;
    var = red[i * 2 + 1] / (double)count - mean * mean;
  } else if (mode == 1) {
    double s0 = 0, s1 = 0;
    for (int k = 0; k < N; ++k) { s0 += red[(k * C + c) * 2]; s1 += red[(k * C + c) * 2 + 1]; }
    const double M = (double)count * N;
    mean = s0 / M;
    var = s1 / M - mean * mean;
    if (n == 0 && running_mean && running_var) {
      const double keep = pow(0.9, (double)steps);
      const double unb = var * M / (M > 1 ? M - 1 : 1);
      running_mean[c] = (float)(keep * running_mean[c] + (1 - keep) * mean);
      running_var[c] = (float)(keep * running_var[c] + (1 - keep) * unb);
    }
  } else if (mode == 2) {
    mean = running_mean[c];
    var = running_var[c];
  } else {
    const int g0 = (c / gs) * gs;
    double s0 = 0, s1 = 0;
    for (int k = g0; k < g0 + gs; ++k) { s0 += red[(n * C + k) * 2]; s1 += red[(n * C + k) * 2 + 1]; }
    const double M = (double)count * gs;
    mean = s0 / M;
    var = s1 / M - mean * mean;
  }
  if (var < 0) var = 0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
  sc[i] = (float)(rstd * ga);
  sh[i] = (float)(be - mean * rstd * ga);
  if (mean_o) mean_o[i] = (float)mean;
  if (rstd_o) rstd_o[i] = (float)rstd;
}

extern "C" int xh_norm_finalize(void* stream, int mode, const double* red, int N, int C, long long count, int gs,
                                float eps, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, int steps, float* sc, float* sh, float* mean, float* rstd) {
  if (mode < 0 || mode > 3 || N <= 0 || C <= 0 || !sc || !sh) return XH_ERR_ARG;
  if (mode != 2 && (!red || count <= 0)) return XH_ERR_ARG;
  if (mode == 2 && (!running_mean || !running_var)) return XH_ERR_ARG;
  if (mode == 3 && (gs <= 0 || C % gs)) return XH_ERR_ARG;
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, mode, red, N, C,
                     count, gs, eps, gamma, beta, running_mean, running_var, steps, sc, sh, mean, rstd);
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- affine + act
// The norm finalisation can ride inside this pass (no xh_norm_finalize launch between the producer of the sums and here): every
// workgroup evaluates the few flops for its own channel and the first workgroup of the channel leaves sc / sh / mean / rstd
// behind for the backward pass (and updates the BatchNorm running statistics).
//   mode -1: sc / sh given;  0: InstanceNorm from raw sums (conv_pack.h: in_finalize);  1 / 2: BatchNorm train / eval, the
//   arithmetic of norm_finalize_kernel
struct AffFin {
  int mode;
  const double* red; int N; double count; float eps;
  const float* gamma; const float* beta; float* running_mean; float* running_var; int steps;
  float* o_sc; float* o_sh; float* o_mean; float* o_rstd;
  // two BatchNorm modules over one tensor (xh_bn_affine_act2): channels >= chalf take the second parameter set, indexed from 0
  int chalf; const float* gamma2; const float* beta2; float* running_mean2; float* running_var2;
};
// (kernel bodies that also run inside a multi-problem launch -- "multi kernels" at the end of this file -- take the block
// coordinates as PARAMETERS named like the built-ins, so the same text serves both)
template <typename T, bool VEC>
__device__ __forceinline__ void affine_act_body(const uint3 blockIdx, const uint3 gridDim, const T* x, long long x_bs, T* y, long long y_bs, int C,
                                                long long dhw, const float* sc, const float* sh, int act, float slope, const AffFin& f) {
  float a, b;
  const int cc = blockIdx.y;
  const long long nc = (long long)blockIdx.z * C + cc;
  const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
  if (f.mode == 0) {
    float m, r;
    in_finalize(f.red[nc * 2], f.red[nc * 2 + 1], 1.0 / f.count, a, b, m, r);
    if (writer) { f.o_sc[nc] = a; f.o_sh[nc] = b; f.o_mean[nc] = m; f.o_rstd[nc] = r; }
  } else if (f.mode > 0) {
    double mean, var;
    const bool second = f.chalf > 0 && cc >= f.chalf;
    const int pc = second ? cc - f.chalf : cc;          // index into the module's own parameter arrays
    const float* gam = second ? f.gamma2 : f.gamma;
    const float* bet = second ? f.beta2 : f.beta;
    float* rmean = second ? f.running_mean2 : f.running_mean;
    float* rvar = second ? f.running_var2 : f.running_var;
    if (f.mode == 1) {
      double s0 = 0, s1 = 0;
      for (int k = 0; k < f.N; ++k) { s0 += f.red[((long long)k * C + cc) * 2]; s1 += f.red[((long long)k * C + cc) * 2 + 1]; }
      const double M = f.count * f.N;
      mean = s0 / M;
      var = s1 / M - mean * mean;
      if (writer && blockIdx.z == 0 && rmean && rvar) {
        const double keep = pow(0.9, (double)f.steps);
        const double unb = var * M / (M > 1 ? M - 1 : 1);
        rmean[pc] = (float)(keep * rmean[pc] + (1 - keep) * mean);
        rvar[pc] = (float)(keep * rvar[pc] + (1 - keep) * unb);
      }
    } else {
      mean = rmean[pc];
      var = rvar[pc];
    }
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)f.eps);
    const double ga = gam ? (double)gam[pc] : 1.0, be = bet ? (double)bet[pc] : 0.0;
    a = (float)(rstd * ga);
    b = (float)(be - mean * rstd * ga);
    if (writer) { f.o_sc[nc] = a; f.o_sh[nc] = b; f.o_mean[nc] = (float)mean; f.o_rstd[nc] = (float)rstd; }
  } else {
    a = sc ? sc[nc] : 1.f;
    b = sh ? sh[nc] : 0.f;
  }
  ROW_LOOP_BEGIN
    const T* xp = x + n * x_bs + (long long)c * dhw;
    T* yp = y + n * y_bs + (long long)c * dhw;
    float v[VW];
    ldrow<VEC>(xp, q, valid, v);
#pragma unroll
    for (int i = 0; i < VW; ++i) v[i] = apply_act(v[i] * a + b, act, slope);
    strow<VEC>(yp, q, valid, v);
  ROW_LOOP_END
}
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void affine_act_kernel(const T* x, long long x_bs, T* y, long long y_bs, int C,
                                                             long long dhw, const float* sc, const float* sh, int act,
                                                             float slope, const AffFin f) {
  affine_act_body<T, VEC>(uint3{blockIdx.x, blockIdx.y, blockIdx.z}, uint3{gridDim.x, gridDim.y, gridDim.z}, x, x_bs, y, y_bs, C, dhw, sc, sh, act, slope, f);
}

static int launch_affine_act(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs, int N, int C, long long DHW,
                             const float* sc, const float* sh, int act, float slope, const AffFin& f) {
  if (!x || !y || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  const bool vec32 = vec_ok<float>(DHW, {x_bs, y_bs}), vec16 = vec_ok<bf16_t>(DHW, {x_bs, y_bs});
  const dim3 grid32 = row_grid<float>(DHW, C, N), grid16 = row_grid<bf16_t>(DHW, C, N);
#define AA(T, V, G) hipLaunchKernelGGL((affine_act_kernel<T, V>), G, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)x, x_bs, (T*)y, y_bs, C, DHW, sc, sh, act, slope, f)
  if (dtype == XH_F32) { if (vec32) AA(float, true, grid32); else AA(float, false, grid32); }
  else if (dtype == XH_BF16) { if (vec16) AA(bf16_t, true, grid16); else AA(bf16_t, false, grid16); }
  else if (dtype == XH_F16) { if (vec16) AA(f16_t, true, grid16); else AA(f16_t, false, grid16); }
  else return XH_ERR_DTYPE;
#undef AA
  return xh_launch_status();
}

extern "C" int xh_affine_act(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs, int N,
                             int C, long long DHW, const float* sc, const float* sh, int act, float slope) {
  AffFin f{};
  f.mode = -1;
  return launch_affine_act(stream, dtype, x, x_bs, y, y_bs, N, C, DHW, sc, sh, act, slope, f);
}

extern "C" int xh_in_affine_act(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs, int N, int C,
                                long long DHW, const double* red, int act, float slope, float* sc, float* sh, float* mean, float* rstd) {
  if (!red || !sc || !sh || !mean || !rstd) return XH_ERR_ARG;
  AffFin f{};
  f.mode = 0; f.red = red; f.N = N; f.count = (double)DHW;
  f.o_sc = sc; f.o_sh = sh; f.o_mean = mean; f.o_rstd = rstd;
  return launch_affine_act(stream, dtype, x, x_bs, y, y_bs, N, C, DHW, nullptr, nullptr, act, slope, f);
}

extern "C" int xh_bn_affine_act(void* stream, int dtype, int mode, const void* x, long long x_bs, void* y, long long y_bs, int N, int C,
                                long long DHW, const double* red, float eps, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, int steps, int act, float slope, float* sc, float* sh, float* mean, float* rstd) {
  if ((mode != 1 && mode != 2) || !sc || !sh || !mean || !rstd) return XH_ERR_ARG;
  if (mode == 1 && !red) return XH_ERR_ARG;
  if (mode == 2 && (!running_mean || !running_var)) return XH_ERR_ARG;
  AffFin f{};
  f.mode = mode; f.red = red; f.N = N; f.count = (double)DHW; f.eps = eps;
  f.gamma = gamma; f.beta = beta; f.running_mean = running_mean; f.running_var = running_var; f.steps = steps;
  f.o_sc = sc; f.o_sh = sh; f.o_mean = mean; f.o_rstd = rstd;
  return launch_affine_act(stream, dtype, x, x_bs, y, y_bs, N, C, DHW, nullptr, nullptr, act, slope, f);
}

// Two BatchNorm modules over the two channel halves of ONE tensor (DuSEAttention's bn_fuse_ch1 / bn_fuse_ch2 on the recon | seg
// pair, modules/DuSFE.py:151-154): channels [0, Chalf) use the first parameter set, [Chalf, C) the second.
extern "C" int xh_bn_affine_act2(void* stream, int dtype, int mode, const void* x, long long x_bs, void* y, long long y_bs, int N, int C,
                                 int Chalf, long long DHW, const double* red, float eps, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, const float* gamma2, const float* beta2, float* running_mean2,
                                 float* running_var2, int steps, int act, float slope, float* sc, float* sh, float* mean, float* rstd) {
  if ((mode != 1 && mode != 2) || !sc || !sh || !mean || !rstd || Chalf <= 0 || Chalf >= C) return XH_ERR_ARG;
  if (mode == 1 && !red) return XH_ERR_ARG;
  if (mode == 2 && (!running_mean || !running_var || !running_mean2 || !running_var2)) return XH_ERR_ARG;
  AffFin f{};
  f.mode = mode; f.red = red; f.N = N; f.count = (double)DHW; f.eps = eps;
  f.gamma = gamma; f.beta = beta; f.running_mean = running_mean; f.running_var = running_var; f.steps = steps;
  f.chalf = Chalf; f.gamma2 = gamma2; f.beta2 = beta2; f.running_mean2 = running_mean2; f.running_var2 = running_var2;
  f.o_sc = sc; f.o_sh = sh; f.o_mean = mean; f.o_rstd = rstd;
  return launch_affine_act(stream, dtype, x, x_bs, y, y_bs, N, C, DHW, nullptr, nullptr, act, slope, f);
}

// ---------------------------------------------------------------------------------------- act/norm backward
template <typename T, bool VEC>
__device__ __forceinline__ void act_bwd_reduce_body(const uint3 blockIdx, const uint3 gridDim, const T* dy, long long dy_bs, const T* x,
                                                    long long x_bs, int C, long long dhw, const float* sc, const float* sh, float slope,
                                                    double* red) {
  __shared__ double s_red[4 * 2];
  const float a = sc[blockIdx.z * C + blockIdx.y], b = sh[blockIdx.z * C + blockIdx.y];
  double s[2] = {0.0, 0.0};
  ROW_LOOP4_BEGIN
    float g4[4][VW], x4[4][VW];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int vu = (int)min((long long)VW, dhw - (q + u * S));
      ldrow<VEC>(dy + n * dy_bs + (long long)c * dhw, q + u * S, vu, g4[u]);
      ldrow<VEC>(x + n * x_bs + (long long)c * dhw, q + u * S, vu, x4[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        const float gg = g4[u][i] * ((x4[u][i] * a + b) > 0.f ? 1.f : slope);
        t0 += gg;
        t1 = fmaf(gg, x4[u][i], t1);
      }
      s[0] += (double)t0;
      s[1] += (double)t1;
    }
  ROW_LOOP4_TAIL
    float g[VW], xv[VW];
    ldrow<VEC>(dy + n * dy_bs + (long long)c * dhw, q, valid, g);
    ldrow<VEC>(x + n * x_bs + (long long)c * dhw, q, valid, xv);
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      const float gg = g[i] * ((xv[i] * a + b) > 0.f ? 1.f : slope);
      t0 += gg;
      t1 = fmaf(gg, xv[i], t1);
    }
    s[0] += (double)t0;
    s[1] += (double)t1;
  ROW_LOOP_END
  block_sum_d<2>(s, s_red, EW_BLOCK >> 6);
  if (threadIdx.x < 2) atomicAdd(&red[((long long)blockIdx.z * C + blockIdx.y) * 2 + threadIdx.x], s_red[threadIdx.x]);
}
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void act_bwd_reduce_kernel(const T* dy, long long dy_bs, const T* x, long long x_bs,
                                                                 int C, long long dhw, const float* sc, const float* sh,
                                                                 float slope, double* red) {
  act_bwd_reduce_body<T, VEC>(uint3{blockIdx.x, blockIdx.y, blockIdx.z}, uint3{gridDim.x, gridDim.y, gridDim.z}, dy, dy_bs, x, x_bs, C, dhw, sc, sh, slope, red);
}

extern "C" int xh_act_bwd_reduce(void* stream, int dtype, const void* dy, long long dy_bs, const void* x, long long x_bs,
                                 int N, int C, long long DHW, const float* sc, const float* sh, float slope, double* red) {
  if (!dy || !x || !sc || !sh || !red || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  const bool vec32 = vec_ok<float>(DHW, {dy_bs, x_bs}), vec16 = vec_ok<bf16_t>(DHW, {dy_bs, x_bs});
  const dim3 grid32 = red_grid<float>(DHW, C, N), grid16 = red_grid<bf16_t>(DHW, C, N);
  if (dtype == XH_F32)
    { if (vec32) hipLaunchKernelGGL((act_bwd_reduce_kernel<float, true>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)dy, dy_bs, (const float*)x, x_bs, C, DHW, sc, sh, slope, red); else hipLaunchKernelGGL((act_bwd_reduce_kernel<float, false>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)dy, dy_bs, (const float*)x, x_bs, C, DHW, sc, sh, slope, red); }
  else if (dtype == XH_BF16)
    { if (vec16) hipLaunchKernelGGL((act_bwd_reduce_kernel<bf16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)dy, dy_bs, (const bf16_t*)x, x_bs, C, DHW, sc, sh, slope, red); else hipLaunchKernelGGL((act_bwd_reduce_kernel<bf16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)dy, dy_bs, (const bf16_t*)x, x_bs, C, DHW, sc, sh, slope, red); }
  else if (dtype == XH_F16)
    { if (vec16) hipLaunchKernelGGL((act_bwd_reduce_kernel<f16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)dy, dy_bs, (const f16_t*)x, x_bs, C, DHW, sc, sh, slope, red); else hipLaunchKernelGGL((act_bwd_reduce_kernel<f16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)dy, dy_bs, (const f16_t*)x, x_bs, C, DHW, sc, sh, slope, red); }
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

// red[n][c] = {sum g, sum g*x};  dx = A*g + Cc*x + B
__global__ void norm_bwd_coef_kernel(int mode, const double* red, int N, int C, long long count, int gs,
                                     const float* gamma, const float* mean, const float* rstd, float* A, float* B,
                                     float* Cc, float* dgamma, float* dbeta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  const double ga = gamma ? (double)gamma[c] : 1.0;
  const double mu = mean[i], rs = rstd[i];
  // P_c = sum g*xhat for this (n,c)
  auto P_of = [&](int k) { return (double)rstd[k] * (red[k * 2 + 1] - (double)mean[k] * red[k * 2]); };
  double S0 = 0, P = 0, M = 1;
  if (mode == 0) {
    S0 = ga * red[i * 2]; P = ga * P_of(i); M = (double)count;
  } else if (mode == 1 || mode == 2) {
    for (int k = 0; k < N; ++k) { S0 += red[(k * C + c) * 2]; P += P_of(k * C + c); }
    if (n == 0) {
      if (dgamma) dgamma[c] += (float)P;
      if (dbeta) dbeta[c] += (float)S0;
    }
    S0 *= ga; P *= ga; M = (double)count * N;
  } else {
    const int g0 = (c / gs) * gs;
    for (int k = g0; k < g0 + gs; ++k) {
      const double gk = gamma ? (double)gamma[k] : 1.0;
      S0 += gk * red[(n * C + k) * 2];
      P += gk * P_of(n * C + k);
    }
    M = (double)count * gs;
    // per-channel affine gradients need the sum over n of this channel's own sums: done by the n==0 lane
    if (n == 0) {
      double p = 0, s = 0;
      for (int k = 0; k < N; ++k) { p += P_of(k * C + c); s += red[(k * C + c) * 2]; }
      if (dgamma) dgamma[c] += (float)p;
      if (dbeta) dbeta[c] += (float)s;
    }
  }
  if (mode == 2) {
    A[i] = (float)(ga * rs); B[i] = 0.f; Cc[i] = 0.f;
  } else {
    A[i] = (float)(ga * rs);
    Cc[i] = (float)(-rs * rs * P / M);
    B[i] = (float)(-rs * S0 / M + rs * rs * mu * P / M);
  }
}

extern "C" int xh_norm_bwd_coef(void* stream, int mode, const double* red, int N, int C, long long count, int gs,
                                const float* gamma, const float* mean, const float* rstd, float* A, float* B, float* Cc,
                                float* dgamma, float* dbeta) {
  if (mode < 0 || mode > 3 || !red || !mean || !rstd || !A || !B || !Cc || N <= 0 || C <= 0 || count <= 0) return XH_ERR_ARG;
  if (mode == 3 && (gs <= 0 || C % gs)) return XH_ERR_ARG;
  hipLaunchKernelGGL(norm_bwd_coef_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, mode, red, N, C,
                     count, gs, gamma, mean, rstd, A, B, Cc, dgamma, dbeta);
  return xh_launch_status();
}

template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void norm_bwd_apply_kernel(const T* dy, long long dy_bs, const T* x, long long x_bs,
                                                                 T* dx, long long dx_bs, int C, long long dhw,
                                                                 const float* A, const float* B, const float* Cc,
                                                                 int have_g, const float* sc, const float* sh, float slope,
                                                                 int accumulate) {
  const int k = blockIdx.z * C + blockIdx.y;
  const float a_ = A[k], b_ = B[k], c_ = Cc[k];
  float tsc = 1.f, tsh = 0.f;
  if (!have_g) { tsc = sc[k]; tsh = sh[k]; }
  ROW_LOOP_BEGIN
    float g[VW], xv[VW], o[VW];
    ldrow<VEC>(dy + n * dy_bs + (long long)c * dhw, q, valid, g);
    ldrow<VEC>(x + n * x_bs + (long long)c * dhw, q, valid, xv);
    T* dp = dx + n * dx_bs + (long long)c * dhw;
    if (accumulate) ldrow<VEC>((const T*)dp, q, valid, o);
    else {
#pragma unroll
      for (int i = 0; i < VW; ++i) o[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      float gg = g[i];
      if (!have_g) gg *= ((xv[i] * tsc + tsh) > 0.f ? 1.f : slope);
      o[i] += a_ * gg + c_ * xv[i] + b_;
    }
    strow<VEC>(dp, q, valid, o);
  ROW_LOOP_END
}

// GroupNorm / BatchNorm flavour with the coefficient step folded in (xh_norm_bwd_fused): the arithmetic of
// norm_bwd_coef_kernel, evaluated by every workgroup for its own (n, c) -- a loop over the gs channels of the group or the N
// samples of the batch -- and the affine gradients added by the first workgroup of sample 0.  g = dy (activation already undone).
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void norm_bwd_fused_kernel(const T* dy, long long dy_bs, const T* x, long long x_bs, T* dx,
                                                                 long long dx_bs, int mode, const double* red, int N, int C,
                                                                 long long dhw, int gs, const float* gamma, const float* mean,
                                                                 const float* rstd, float* dgamma, float* dbeta, int chalf,
                                                                 const float* gamma2, float* dgamma2, float* dbeta2) {
  const int cc = blockIdx.y, nn = blockIdx.z;
  const int i0 = nn * C + cc;
  // chalf > 0 (BatchNorm modes only): two modules over the channel halves, the second one's arrays indexed from 0
  if (chalf > 0 && cc >= chalf) { gamma = gamma2 ? gamma2 - chalf : nullptr; dgamma = dgamma2 ? dgamma2 - chalf : nullptr; dbeta = dbeta2 ? dbeta2 - chalf : nullptr; }
  const double ga = gamma ? (double)gamma[cc] : 1.0;
  const double mu = mean[i0], rs = rstd[i0];
  auto P_of = [&](int k) { return (double)rstd[k] * (red[k * 2 + 1] - (double)mean[k] * red[k * 2]); };
  double S0 = 0, P = 0, M = 1;
  const bool writer = blockIdx.x == 0 && nn == 0 && threadIdx.x == 0;
  if (mode == 1 || mode == 2) {
    for (int k = 0; k < N; ++k) { S0 += red[(k * C + cc) * 2]; P += P_of(k * C + cc); }
    if (writer) {
      if (dgamma) dgamma[cc] += (float)P;
      if (dbeta) dbeta[cc] += (float)S0;
    }
    S0 *= ga; P *= ga; M = (double)dhw * N;
  } else {
    const int g0 = (cc / gs) * gs;
    for (int k = g0; k < g0 + gs; ++k) {
      const double gk = gamma ? (double)gamma[k] : 1.0;
      S0 += gk * red[(nn * C + k) * 2];
      P += gk * P_of(nn * C + k);
    }
    M = (double)dhw * gs;
    if (writer) {
      double pp = 0, ss = 0;
      for (int k = 0; k < N; ++k) { pp += P_of(k * C + cc); ss += red[(k * C + cc) * 2]; }
      if (dgamma) dgamma[cc] += (float)pp;
      if (dbeta) dbeta[cc] += (float)ss;
    }
  }
  float a_ = (float)(ga * rs), b_ = 0.f, c_ = 0.f;
  if (mode != 2) {
    c_ = (float)(-rs * rs * P / M);
    b_ = (float)(-rs * S0 / M + rs * rs * mu * P / M);
  }
  ROW_LOOP_BEGIN
    float g[VW], xv[VW], o[VW];
    ldrow<VEC>(dy + n * dy_bs + (long long)c * dhw, q, valid, g);
    ldrow<VEC>(x + n * x_bs + (long long)c * dhw, q, valid, xv);
#pragma unroll
    for (int i = 0; i < VW; ++i) o[i] = a_ * g[i] + c_ * xv[i] + b_;
    strow<VEC>(dx + n * dx_bs + (long long)c * dhw, q, valid, o);
  ROW_LOOP_END
}

static int launch_norm_bwd_fused(void* stream, int dtype, int mode, const void* dy, long long dy_bs, const void* x, long long x_bs,
                                 void* dx, long long dx_bs, int N, int C, long long DHW, const double* red, int gs,
                                 const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta, int chalf,
                                 const float* gamma2, float* dgamma2, float* dbeta2) {
  if (mode < 1 || mode > 3 || !dy || !x || !dx || !red || !mean || !rstd || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535)
    return XH_ERR_ARG;
  if (mode == 3 && (gs <= 0 || C % gs)) return XH_ERR_ARG;
  if (chalf && (mode == 3 || chalf < 0 || chalf >= C)) return XH_ERR_ARG;
  if (mode != 3) gs = 1;
  const bool vec32 = vec_ok<float>(DHW, {dy_bs, x_bs, dx_bs}), vec16 = vec_ok<bf16_t>(DHW, {dy_bs, x_bs, dx_bs});
  const dim3 grid32 = row_grid<float>(DHW, C, N), grid16 = row_grid<bf16_t>(DHW, C, N);
#define NB(T, V, G) hipLaunchKernelGGL((norm_bwd_fused_kernel<T, V>), G, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)dy, dy_bs, (const T*)x, x_bs, (T*)dx, dx_bs, mode, red, N, C, DHW, gs, gamma, mean, rstd, dgamma, dbeta, chalf, gamma2, dgamma2, dbeta2)
  if (dtype == XH_F32) { if (vec32) NB(float, true, grid32); else NB(float, false, grid32); }
  else if (dtype == XH_BF16) { if (vec16) NB(bf16_t, true, grid16); else NB(bf16_t, false, grid16); }
  else if (dtype == XH_F16) { if (vec16) NB(f16_t, true, grid16); else NB(f16_t, false, grid16); }
  else return XH_ERR_DTYPE;
#undef NB
  return xh_launch_status();
}
extern "C" int xh_norm_bwd_fused(void* stream, int dtype, int mode, const void* dy, long long dy_bs, const void* x, long long x_bs,
                                 void* dx, long long dx_bs, int N, int C, long long DHW, const double* red, int gs,
                                 const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta) {
  return launch_norm_bwd_fused(stream, dtype, mode, dy, dy_bs, x, x_bs, dx, dx_bs, N, C, DHW, red, gs, gamma, mean, rstd, dgamma, dbeta, 0,
                               nullptr, nullptr, nullptr);
}
// BatchNorm backward of two modules over the channel halves of one tensor (see xh_bn_affine_act2); mode 1 | 2.
extern "C" int xh_norm_bwd_fused2(void* stream, int dtype, int mode, const void* dy, long long dy_bs, const void* x, long long x_bs,
                                  void* dx, long long dx_bs, int N, int C, int Chalf, long long DHW, const double* red,
                                  const float* gamma, const float* gamma2, const float* mean, const float* rstd, float* dgamma,
                                  float* dbeta, float* dgamma2, float* dbeta2) {
  if (Chalf <= 0) return XH_ERR_ARG;
  return launch_norm_bwd_fused(stream, dtype, mode, dy, dy_bs, x, x_bs, dx, dx_bs, N, C, DHW, red, 1, gamma, mean, rstd, dgamma, dbeta, Chalf,
                               gamma2, dgamma2, dbeta2);
}

// InstanceNorm flavour with the coefficient step folded in: every workgroup derives A, B, C of its (n, c) row from the
// raw fp64 sums (sum g, sum g*x), mean and rstd -- a handful of flops -- so the one-block coefficient launch disappears.
// The statistics arrays may be wider than this tensor's channel count (virtual concat): row stride `stat_rs`.
template <typename T, bool VEC>
__device__ __forceinline__ void in_bwd_apply_body(const uint3 blockIdx, const uint3 gridDim, const T* dy, long long dy_bs, const T* x,
                                                  long long x_bs, T* dx, long long dx_bs, int C, long long dhw, const double* red,
                                                  const float* mean, const float* rstd, int stat_rs, double count, int have_g,
                                                  const float* sc, const float* sh, float slope, int accumulate, const T* xb,
                                                  long long xb_bs, T* dxb, long long dxb_bs, int ca) {
  // xb != nullptr: virtual concat (xa | xb): channels >= ca read xb and write dxb (xh_in_bwd_apply2)
  const int k = blockIdx.z * stat_rs + blockIdx.y;
  const T* xrow;
  T* drow;
  {
    const int c = blockIdx.y, n = blockIdx.z;
    if (xb && c >= ca) { xrow = xb + n * xb_bs + (long long)(c - ca) * dhw; drow = dxb + n * dxb_bs + (long long)(c - ca) * dhw; }
    else { xrow = x + n * x_bs + (long long)c * dhw; drow = dx + n * dx_bs + (long long)c * dhw; }
  }
  const double rs = rstd[k], mu = mean[k];
  const double S0 = red[k * 2], P = rs * (red[k * 2 + 1] - mu * red[k * 2]);
  const float a_ = (float)rs, c_ = (float)(-rs * rs * P / count), b_ = (float)(-rs * S0 / count + rs * rs * mu * P / count);
  float tsc = 1.f, tsh = 0.f;
  if (!have_g) { tsc = sc[k]; tsh = sh[k]; }
  ROW_LOOP_BEGIN
    float g[VW], xv[VW], o[VW];
    ldrow<VEC>(dy + n * dy_bs + (long long)c * dhw, q, valid, g);
    ldrow<VEC>(xrow, q, valid, xv);
    T* dp = drow;
    if (xb && blockIdx.y >= ca ? (accumulate & 2) : (accumulate & 1)) ldrow<VEC>((const T*)dp, q, valid, o);     // bit 0: dx, bit 1: dxb
    else {
#pragma unroll
      for (int i = 0; i < VW; ++i) o[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      float gg = g[i];
      if (!have_g) gg *= ((xv[i] * tsc + tsh) > 0.f ? 1.f : slope);
      o[i] += a_ * gg + c_ * xv[i] + b_;
    }
    strow<VEC>(dp, q, valid, o);
  ROW_LOOP_END
}
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void in_bwd_apply_kernel(const T* dy, long long dy_bs, const T* x, long long x_bs,
                                                               T* dx, long long dx_bs, int C, long long dhw,
                                                               const double* red, const float* mean, const float* rstd,
                                                               int stat_rs, double count, int have_g, const float* sc,
                                                               const float* sh, float slope, int accumulate, const T* xb,
                                                               long long xb_bs, T* dxb, long long dxb_bs, int ca) {
  in_bwd_apply_body<T, VEC>(uint3{blockIdx.x, blockIdx.y, blockIdx.z}, uint3{gridDim.x, gridDim.y, gridDim.z}, dy, dy_bs, x, x_bs, dx, dx_bs, C, dhw, red, mean, rstd, stat_rs, count, have_g, sc, sh, slope,
                            accumulate, xb, xb_bs, dxb, dxb_bs, ca);
}
static int launch_in_bwd_apply(void* stream, int dtype, const void* dy, long long dy_bs, const void* x, long long x_bs, void* dx,
                               long long dx_bs, const void* xb, long long xb_bs, void* dxb, long long dxb_bs, int ca, int N, int C,
                               long long DHW, const double* red, const float* mean, const float* rstd, int stat_rs, int have_g,
                               const float* sc, const float* sh, float slope, int accumulate) {
  if (!dy || !x || !dx || !red || !mean || !rstd || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535 || stat_rs < C)
    return XH_ERR_ARG;
  if (!have_g && (!sc || !sh)) return XH_ERR_ARG;
  const bool vec32 = vec_ok<float>(DHW, {dy_bs, x_bs, dx_bs, xb_bs, dxb_bs}), vec16 = vec_ok<bf16_t>(DHW, {dy_bs, x_bs, dx_bs, xb_bs, dxb_bs});
  const dim3 grid32 = row_grid<float>(DHW, C, N), grid16 = row_grid<bf16_t>(DHW, C, N);
#define IB(T, V, G) hipLaunchKernelGGL((in_bwd_apply_kernel<T, V>), G, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)dy, dy_bs, (const T*)x, x_bs, (T*)dx, dx_bs, C, DHW, red, mean, rstd, stat_rs, (double)DHW, have_g, sc, sh, slope, accumulate, (const T*)xb, xb_bs, (T*)dxb, dxb_bs, ca)
  if (dtype == XH_F32) { if (vec32) IB(float, true, grid32); else IB(float, false, grid32); }
  else if (dtype == XH_BF16) { if (vec16) IB(bf16_t, true, grid16); else IB(bf16_t, false, grid16); }
  else if (dtype == XH_F16) { if (vec16) IB(f16_t, true, grid16); else IB(f16_t, false, grid16); }
  else return XH_ERR_DTYPE;
#undef IB
  return xh_launch_status();
}
extern "C" int xh_in_bwd_apply(void* stream, int dtype, const void* dy, long long dy_bs, const void* x, long long x_bs,
                               void* dx, long long dx_bs, int N, int C, long long DHW, const double* red,
                               const float* mean, const float* rstd, int stat_rs, int have_g, const float* sc,
                               const float* sh, float slope, int accumulate) {
  return launch_in_bwd_apply(stream, dtype, dy, dy_bs, x, x_bs, dx, dx_bs, nullptr, 0, nullptr, 0, C, N, C, DHW, red, mean, rstd, stat_rs,
                             have_g, sc, sh, slope, accumulate);
}
extern "C" int xh_in_bwd_apply2(void* stream, int dtype, const void* dy, long long dy_bs, const void* xa, long long xa_bs, void* dxa,
                                long long dxa_bs, int CA, const void* xb, long long xb_bs, void* dxb, long long dxb_bs, int CB, int N,
                                long long DHW, const double* red, const float* mean, const float* rstd, int accumulate) {
  if (!xb || !dxb || CA <= 0 || CB <= 0 || accumulate < 0 || accumulate > 3) return XH_ERR_ARG;
  return launch_in_bwd_apply(stream, dtype, dy, dy_bs, xa, xa_bs, dxa, dxa_bs, xb, xb_bs, dxb, dxb_bs, CA, N, CA + CB, DHW, red, mean,
                             rstd, CA + CB, 1, nullptr, nullptr, 1.f, accumulate);
}

extern "C" int xh_norm_bwd_apply(void* stream, int dtype, const void* dy, long long dy_bs, const void* x, long long x_bs,
                                 void* dx, long long dx_bs, int N, int C, long long DHW, const float* A, const float* B,
                                 const float* Cc, int have_g, const float* sc, const float* sh, float slope,
                                 int accumulate) {
  if (!dy || !x || !dx || !A || !B || !Cc || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  if (!have_g && (!sc || !sh)) return XH_ERR_ARG;
  const bool vec32 = vec_ok<float>(DHW, {dy_bs, x_bs, dx_bs}), vec16 = vec_ok<bf16_t>(DHW, {dy_bs, x_bs, dx_bs});
  const dim3 grid32 = row_grid<float>(DHW, C, N), grid16 = row_grid<bf16_t>(DHW, C, N);
  if (dtype == XH_F32)
    { if (vec32) hipLaunchKernelGGL((norm_bwd_apply_kernel<float, true>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)dy, dy_bs, (const float*)x, x_bs, (float*)dx, dx_bs, C, DHW, A, B, Cc, have_g, sc, sh, slope, accumulate); else hipLaunchKernelGGL((norm_bwd_apply_kernel<float, false>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)dy, dy_bs, (const float*)x, x_bs, (float*)dx, dx_bs, C, DHW, A, B, Cc, have_g, sc, sh, slope, accumulate); }
  else if (dtype == XH_BF16)
    { if (vec16) hipLaunchKernelGGL((norm_bwd_apply_kernel<bf16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)dy, dy_bs, (const bf16_t*)x, x_bs, (bf16_t*)dx, dx_bs, C, DHW, A, B, Cc, have_g, sc, sh, slope, accumulate); else hipLaunchKernelGGL((norm_bwd_apply_kernel<bf16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)dy, dy_bs, (const bf16_t*)x, x_bs, (bf16_t*)dx, dx_bs, C, DHW, A, B, Cc, have_g, sc, sh, slope, accumulate); }
  else if (dtype == XH_F16)
    { if (vec16) hipLaunchKernelGGL((norm_bwd_apply_kernel<f16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)dy, dy_bs, (const f16_t*)x, x_bs, (f16_t*)dx, dx_bs, C, DHW, A, B, Cc, have_g, sc, sh, slope, accumulate); else hipLaunchKernelGGL((norm_bwd_apply_kernel<f16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)dy, dy_bs, (const f16_t*)x, x_bs, (f16_t*)dx, dx_bs, C, DHW, A, B, Cc, have_g, sc, sh, slope, accumulate); }
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- max pool 2^3
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* x, T* y, long long total, int D, int H, int W) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow = (int)(i % Wo);
    long long r = i / Wo;
    const int oh = (int)(r % Ho); r /= Ho;
    const int od = (int)(r % Do);
    const long long nc = r / Do;
    const T* p = x + ((nc * D + 2 * od) * H + 2 * oh) * (long long)W + 2 * ow;
    float m = -INFINITY;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const float v = ldf(p, ((long long)dz * H + dy) * W + dx);
          m = (v > m || v != v) ? v : m;
        }
    stf(y, i, m);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* x, const T* dy, T* dx, long long total, int D, int H,
                                                          int W, int accumulate) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow = (int)(i % Wo);
    long long r = i / Wo;
    const int oh = (int)(r % Ho); r /= Ho;
    const int od = (int)(r % Do);
    const long long nc = r / Do;
    const long long base = ((nc * D + 2 * od) * H + 2 * oh) * (long long)W + 2 * ow;
    float m = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = ldf(x, base + ((long long)(k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1));
      if (v > m || v != v) { m = v; arg = k; }
    }
    const float g = ldf(dy, i);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long long o = base + ((long long)(k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1);
      const float prev = accumulate ? ldf((const T*)dx, o) : 0.f;
      stf(dx, o, prev + (k == arg ? g : 0.f));
    }
  }
}
static inline int flat_grid(long long total) {
  long long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  return (int)(b < 1 ? 1 : b);
}
extern "C" int xh_maxpool2_fwd(void* stream, int dtype, const void* x, void* y, int NC, int D, int H, int W) {
  if (!x || !y || NC <= 0 || D < 2 || H < 2 || W < 2 || (D & 1) || (H & 1) || (W & 1)) return XH_ERR_ARG;
  const long long total = (long long)NC * (D / 2) * (H / 2) * (W / 2);
  if (dtype == XH_F32)
    hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, total, D, H, W);
  else if (dtype == XH_BF16)
    hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, total, D, H, W);
  else if (dtype == XH_F16)
    hipLaunchKernelGGL(maxpool2_fwd_kernel<f16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, (f16_t*)y, total, D, H, W);
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}
extern "C" int xh_maxpool2_bwd(void* stream, int dtype, const void* x, const void* dy, void* dx, int NC, int D, int H,
                               int W, int accumulate) {
  if (!x || !dy || !dx || NC <= 0 || D < 2 || H < 2 || W < 2 || (D & 1) || (H & 1) || (W & 1)) return XH_ERR_ARG;
  const long long total = (long long)NC * (D / 2) * (H / 2) * (W / 2);
  if (dtype == XH_F32)
    hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dy, (float*)dx, total, D, H, W, accumulate);
  else if (dtype == XH_BF16)
    hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, total, D, H, W, accumulate);
  else if (dtype == XH_F16)
    hipLaunchKernelGGL(maxpool2_bwd_kernel<f16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, (const f16_t*)dy, (f16_t*)dx, total, D, H, W, accumulate);
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- trilinear
// PyTorch area_pixel_compute_source_index(align_corners=False): src = max((dst+0.5)*scale-0.5, 0)
__device__ __forceinline__ void lin_src(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  float s = ((float)o + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}
template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* x, long long x_bs, T* y, long long y_bs, int C, int D,
                                                          int H, int W, int Do, int Ho, int Wo, long long total) {
  const float sd = (float)D / Do, sh = (float)H / Ho, sw = (float)W / Wo;
  const long long dhw = (long long)D * H * W, odhw = (long long)Do * Ho * Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow = (int)(i % Wo);
    long long r = i / Wo;
    const int oh = (int)(r % Ho); r /= Ho;
    const int od = (int)(r % Do); r /= Do;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    int d0, d1, h0, h1, w0, w1;
    float ld0, ld1, lh0, lh1, lw0, lw1;
    lin_src(od, sd, D, d0, d1, ld0, ld1);
    lin_src(oh, sh, H, h0, h1, lh0, lh1);
    lin_src(ow, sw, W, w0, w1, lw0, lw1);
    const T* p = x + n * x_bs + c * dhw;
    auto at = [&](int d, int h, int w) { return ldf(p, ((long long)d * H + h) * W + w); };
    const float v = ld0 * (lh0 * (lw0 * at(d0, h0, w0) + lw1 * at(d0, h0, w1)) + lh1 * (lw0 * at(d0, h1, w0) + lw1 * at(d0, h1, w1))) +
                    ld1 * (lh0 * (lw0 * at(d1, h0, w0) + lw1 * at(d1, h0, w1)) + lh1 * (lw0 * at(d1, h1, w0) + lw1 * at(d1, h1, w1)));
    stf(y + n * y_bs + c * odhw, ((long long)od * Ho + oh) * Wo + ow, v);
  }
}
// weight of output o on input i along one axis (exactly the forward's coefficients)
__device__ __forceinline__ float lin_w(int o, int i, float scale, int in) {
  int i0, i1; float l0, l1;
  lin_src(o, scale, in, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}
__device__ __forceinline__ void cand(int i, float scale, int out, int& lo, int& hi) {
  lo = (int)floorf(((float)i - 0.5f) / scale - 0.5f);
  hi = (int)ceilf(((float)i + 1.5f) / scale - 0.5f);
  lo = lo < 0 ? 0 : lo;
  hi = hi > out - 1 ? out - 1 : hi;
}
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* dy, long long dy_bs, T* dx, long long dx_bs, int C,
                                                          int D, int H, int W, int Do, int Ho, int Wo, long long total,
                                                          int accumulate) {
  const float sd = (float)D / Do, sh = (float)H / Ho, sw = (float)W / Wo;
  const long long dhw = (long long)D * H * W, odhw = (long long)Do * Ho * Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int w = (int)(i % W);
    long long r = i / W;
    const int h = (int)(r % H); r /= H;
    const int d = (int)(r % D); r /= D;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    int dlo, dhi, hlo, hhi, wlo, whi;
    cand(d, sd, Do, dlo, dhi);
    cand(h, sh, Ho, hlo, hhi);
    cand(w, sw, Wo, wlo, whi);
    // per-axis adjoint weights, evaluated once per lane (<= 8 candidates per axis; exactly 4 for 2x upsampling)
    constexpr int MC = 8;
    float wd[MC], wh[MC], ww[MC];
    dhi = min(dhi, dlo + MC - 1); hhi = min(hhi, hlo + MC - 1); whi = min(whi, wlo + MC - 1);
#pragma unroll
    for (int k = 0; k < MC; ++k) {
      wd[k] = (dlo + k <= dhi) ? lin_w(dlo + k, d, sd, D) : 0.f;
      wh[k] = (hlo + k <= hhi) ? lin_w(hlo + k, h, sh, H) : 0.f;
      ww[k] = (wlo + k <= whi) ? lin_w(wlo + k, w, sw, W) : 0.f;
    }
    const T* p = dy + n * dy_bs + c * odhw;
    float acc = 0.f;
#pragma unroll
    for (int kd = 0; kd < MC; ++kd) {
      if (wd[kd] == 0.f) continue;
#pragma unroll
      for (int kh = 0; kh < MC; ++kh) {
        if (wh[kh] == 0.f) continue;
        const T* row = p + ((long long)(dlo + kd) * Ho + (hlo + kh)) * Wo + wlo;
        float rowacc = 0.f;
#pragma unroll
        for (int kw = 0; kw < MC; ++kw)
          if (ww[kw] != 0.f) rowacc = fmaf(ww[kw], ldf(row, kw), rowacc);
        acc = fmaf(wd[kd] * wh[kh], rowacc, acc);
      }
    }
    T* o = dx + n * dx_bs + c * dhw;
    const long long sp = ((long long)d * H + h) * W + w;
    stf(o, sp, acc + (accumulate ? ldf((const T*)o, sp) : 0.f));
  }
}
// ---- exact 2x (Do=2D, Ho=2H, Wo=2W; every use in the network) -------------------------------------------------------
// align_corners=False at scale 2 is a fixed stencil per axis:  out[2i] = .25 x[i-1] + .75 x[i],
// out[2i+1] = .75 x[i] + .25 x[i+1], with the index clamped at the borders.  A lane owns one 8-byte input run (4 bf16 /
// 2 fp32 voxels = one 16-byte output run), marches through the input planes of its depth segment, reads each plane
// once (3 rows, W neighbours by wave shuffle) and keeps the H/W-interpolated previous plane in registers.
template <int N>
__device__ __forceinline__ void ldhalf(const float* p, long long q, float (&o)[N]) {
  static_assert(N == 2, "");
  const float2 t = *reinterpret_cast<const float2*>(p + q); o[0] = t.x; o[1] = t.y;
}
template <int N, int F>
__device__ __forceinline__ void ldhalf(const h16<F>* p, long long q, float (&o)[N]) {
  static_assert(N == 4, "");
  ld4(p, q, o);
}
__device__ __forceinline__ void sthalf(float* p, long long q, const float (&o)[2]) {
  *reinterpret_cast<float2*>(p + q) = make_float2(o[0], o[1]);
}
template <int F> __device__ __forceinline__ void sthalf(h16<F>* p, long long q, const float (&o)[4]) { st4(p, q, o); }
// the same runs as raw registers (requested one plane ahead of their use: 2 / 4 registers per run in flight) and their conversion
template <typename T> __device__ __forceinline__ uint2 ldraw8(const T* p, long long q) { return *reinterpret_cast<const uint2*>(p + q); }
template <typename T> __device__ __forceinline__ uint4 ldraw16(const T* p, long long q) { return *reinterpret_cast<const uint4*>(p + q); }
__device__ __forceinline__ void cvtraw(const float*, uint2 t, float (&o)[2]) { o[0] = __uint_as_float(t.x); o[1] = __uint_as_float(t.y); }
template <int F> __device__ __forceinline__ void cvtraw(const h16<F>*, uint2 t, float (&o)[4]) {
  o[0] = cvt_lo<F>(t.x); o[1] = cvt_hi<F>(t.x); o[2] = cvt_lo<F>(t.y); o[3] = cvt_hi<F>(t.y);
}
__device__ __forceinline__ void cvtraw(const float*, uint4 t, float (&o)[4]) {
  o[0] = __uint_as_float(t.x); o[1] = __uint_as_float(t.y); o[2] = __uint_as_float(t.z); o[3] = __uint_as_float(t.w);
}
template <int F> __device__ __forceinline__ void cvtraw(const h16<F>*, uint4 t, float (&o)[8]) {
  const unsigned u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { o[2 * k] = cvt_lo<F>(u[k]); o[2 * k + 1] = cvt_hi<F>(u[k]); }
}

// Both kernels march through the planes of a depth segment; a plane's rows are REQUESTED one step before they are used (raw
// registers), so the march is not a chain of exposed memory round trips (sd + 2 of them: 10 us for any volume before).
// PRE: the input is a conv output whose InstanceNorm + LeakyReLU have not been applied yet (BasicConv + Upsampling,
// RA_HVED.py:599-601): finalised here from the raw channel sums `fin.red` and applied to every value on load.
struct UpFin {
  const double* red; double inv_count; float slope;
  float *o_sc, *o_sh, *o_mean, *o_rstd;
};
template <typename T, int TXN, bool PRE = false>
__device__ __forceinline__ void upsample2x_fwd_body(const uint3 blockIdx, const T* __restrict__ x, long long x_bs, T* __restrict__ y, long long y_bs,
                                                    int C, int D, int H, int W, int sd, int tilesW, int tilesH, const UpFin& fin) {
  constexpr int VO = VWT<T>::v, VI = VO / 2, TH = 256 / TXN;
  const int tid = threadIdx.x, tx = tid % TXN, ty = tid / TXN;
  const int c = blockIdx.y, n = blockIdx.z;
  float pa = 1.f, pb = 0.f;
  double fs1 = 0.0, fs2 = 0.0;
  if (PRE) { fs1 = fin.red[((long long)n * C + c) * 2]; fs2 = fin.red[((long long)n * C + c) * 2 + 1]; }
  const float pslope = fin.slope;
  auto pre = [&](float v) { if (PRE) { v = fmaf(v, pa, pb); v = v > 0.f ? v : v * pslope; } return v; };
  int t = blockIdx.x;
  const int tw = t % tilesW; t /= tilesW;
  const int th = t % tilesH;
  const int ds = t / tilesH;
  const int h = th * TH + ty, w0 = (tw * TXN + tx) * VI;
  const bool active = h < H && w0 < W;
  const int hc = min(h, H - 1), wc = active ? w0 : 0;
  const int d_begin = ds * sd, d_end = min(D, d_begin + sd);
  const long long hw = (long long)H * W;
  const int Ho = 2 * H, Wo = 2 * W;
  const long long ohw = (long long)Ho * Wo;
  const T* src = x + n * x_bs + (long long)c * D * hw;
  T* dst = y + n * y_bs + (long long)c * 2 * D * ohw;
  const int r0 = max(hc - 1, 0) * W, r1 = hc * W, r2 = min(hc + 1, H - 1) * W;
  const bool edge_l = tx == 0, edge_r = tx == TXN - 1 || w0 + VI >= W;
  const bool glob_l = edge_l && tilesW > 1 && w0 > 0, glob_r = edge_r && tilesW > 1 && w0 + VI < W;
  float prev[2][VO], cur[2][VO];
  const int ro[3] = {r0, r1, r2};
  uint2 raw[3], nxt[3];
  const T* pl = src + (long long)min(max(d_begin - 1, 0), D - 1) * hw;
#pragma unroll
  for (int k = 0; k < 3; ++k) raw[k] = ldraw8(pl, ro[k] + wc);
  if (PRE) {                                             // behind the first plane's requests
    const long long nc = (long long)n * C + c;
    float m, r;
    in_finalize(fs1, fs2, fin.inv_count, pa, pb, m, r);
    if (blockIdx.x == 0 && tid == 0) { fin.o_sc[nc] = pa; fin.o_sh[nc] = pb; fin.o_mean[nc] = m; fin.o_rstd[nc] = r; }
  }
  for (int p = d_begin - 1; p <= d_end; ++p) {
    const T* pn = src + (long long)min(max(p + 1, 0), D - 1) * hw;      // clamped: always a valid plane, unused past d_end
#pragma unroll
    for (int k = 0; k < 3; ++k) nxt[k] = ldraw8(pn, ro[k] + wc);
    float rows[3][VI + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v[VI];
      cvtraw(pl, raw[k], v);
#pragma unroll
      for (int j = 0; j < VI; ++j) { v[j] = pre(v[j]); rows[k][j + 1] = v[j]; }
      float l = __shfl_up(v[VI - 1], 1, 64), r = __shfl_down(v[0], 1, 64);
      if (edge_l) l = glob_l ? pre(ldf(pl, ro[k] + w0 - 1)) : v[0];          // clamped index at the volume border
      if (edge_r) r = glob_r ? pre(ldf(pl, ro[k] + w0 + VI)) : v[VI - 1];
      rows[k][0] = l;
      rows[k][VI + 1] = r;
    }
    pl = pn;
#pragma unroll
    for (int k = 0; k < 3; ++k) raw[k] = nxt[k];
    // W interpolation of the three rows, then H interpolation into the two output rows of this input row
    float wi[3][VO];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < VI; ++j) {
        wi[k][2 * j] = 0.25f * rows[k][j] + 0.75f * rows[k][j + 1];
        wi[k][2 * j + 1] = 0.75f * rows[k][j + 1] + 0.25f * rows[k][j + 2];
      }
#pragma unroll
    for (int v = 0; v < VO; ++v) {
      cur[0][v] = 0.25f * wi[0][v] + 0.75f * wi[1][v];
      cur[1][v] = 0.75f * wi[1][v] + 0.25f * wi[2][v];
    }
    if (p > d_begin - 1 && active) {
      // output plane 2(p-1)+1 = .75 U[p-1] + .25 U[p];  output plane 2p = .25 U[p-1] + .75 U[p]
      const int q = p - 1;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        float o[VO];
        if (q >= d_begin) {
#pragma unroll
          for (int v = 0; v < VO; ++v) o[v] = 0.75f * prev[rr][v] + 0.25f * cur[rr][v];
          stvec(dst, (long long)(2 * q + 1) * ohw + (long long)(2 * h + rr) * Wo + 2 * w0, o);
        }
        if (p < d_end) {
#pragma unroll
          for (int v = 0; v < VO; ++v) o[v] = 0.25f * prev[rr][v] + 0.75f * cur[rr][v];
          stvec(dst, (long long)(2 * p) * ohw + (long long)(2 * h + rr) * Wo + 2 * w0, o);
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int v = 0; v < VO; ++v) prev[rr][v] = cur[rr][v];
  }
}

template <typename T, int TXN, bool PRE = false>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ x, long long x_bs, T* __restrict__ y, long long y_bs, int C, int D,
                                                            int H, int W, int sd, int tilesW, int tilesH, const UpFin fin) {
  upsample2x_fwd_body<T, TXN, PRE>(uint3{blockIdx.x, blockIdx.y, blockIdx.z}, x, x_bs, y, y_bs, C, D, H, W, sd, tilesW, tilesH, fin);
}

// Adjoint of the same stencil: dx[i] = .25 dy[2i-1] + .75 dy[2i] + .75 dy[2i+1] + .25 dy[2i+2] per axis, indices clamped
// (the clamped forward taps fold back onto the border voxel).  Same lane role, marching through the dy planes.
// RED: dx is the gradient of leaky(y0 * sc + sh) (the same BasicConv + Upsampling pair): the kernel also leaves the two sums the
// InstanceNorm backward needs, red[n][c] += (sum dz, sum dz * y0) with dz = dx * leaky'(.), dx as stored (xh_act_bwd_reduce's).
struct UpRed {
  const void* y0; long long y0_bs; const float *sc, *sh; float slope; double* red;
};
__device__ __forceinline__ float up_stored(const float*, float v) { return v; }
template <int F> __device__ __forceinline__ float up_stored(const h16<F>*, float v) { return cvt_lo<F>(cvt_pack<F>(v, 0.f)); }
template <typename T, int TXN, bool RED = false>
__device__ __forceinline__ void upsample2x_bwd_body(const uint3 blockIdx, const T* __restrict__ dy, long long dy_bs, T* __restrict__ dx, long long dx_bs,
                                                    int C, int D, int H, int W, int sd, int tilesW, int tilesH, int accumulate, const UpRed& ur) {
  constexpr int VO = VWT<T>::v, VI = VO / 2, TH = 256 / TXN;
  const int tid = threadIdx.x, tx = tid % TXN, ty = tid / TXN;
  const int c = blockIdx.y, n = blockIdx.z;
  float ra = 1.f, rb = 0.f, rs0 = 0.f, rs1 = 0.f;
  const T* y0p = nullptr;
  if (RED) {
    ra = ur.sc[(long long)n * C + c]; rb = ur.sh[(long long)n * C + c];
    y0p = (const T*)ur.y0 + n * ur.y0_bs + (long long)c * D * H * W;
  }
  int t = blockIdx.x;
  const int tw = t % tilesW; t /= tilesW;
  const int th = t % tilesH;
  const int ds = t / tilesH;
  const int h = th * TH + ty, w0 = (tw * TXN + tx) * VI;
  const bool active = h < H && w0 < W;
  const int hc = min(h, H - 1), wc = active ? w0 : 0;
  const int d_begin = ds * sd, d_end = min(D, d_begin + sd);
  const long long hw = (long long)H * W;
  const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
  const long long ohw = (long long)Ho * Wo;
  const T* src = dy + n * dy_bs + (long long)c * Do * ohw;
  T* dst = dx + n * dx_bs + (long long)c * D * hw;
  int ro[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ro[k] = min(max(2 * hc - 1 + k, 0), Ho - 1) * Wo;
  const bool edge_l = tx == 0, edge_r = tx == TXN - 1 || w0 + VI >= W;
  const bool glob_l = edge_l && tilesW > 1 && w0 > 0, glob_r = edge_r && tilesW > 1 && w0 + VI < W;
  const float ch[4] = {0.25f, 0.75f, 0.75f, 0.25f};
  float acc_prev[VI], acc_cur[VI], acc_next[VI];
#pragma unroll
  for (int j = 0; j < VI; ++j) acc_prev[j] = acc_cur[j] = acc_next[j] = 0.f;
  // a dy plane = 4 rows of this lane's 16-byte run; requested one step ahead (clamped plane index: always a valid address)
  auto request = [&](int od, uint4 (&raw)[4]) {
    const T* pl = src + (long long)min(max(od, 0), Do - 1) * ohw;
#pragma unroll
    for (int k = 0; k < 4; ++k) raw[k] = ldraw16(pl, ro[k] + 2 * wc);
  };
  // V(od)[j]: the H/W-reduced dy plane od for this lane's VI inputs
  auto plane = [&](int od, const uint4 (&raw)[4], float (&V)[VI]) {
    const T* pl = src + (long long)min(max(od, 0), Do - 1) * ohw;
#pragma unroll
    for (int j = 0; j < VI; ++j) V[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v[VO], r[VO + 2];
      cvtraw(pl, raw[k], v);
#pragma unroll
      for (int i = 0; i < VO; ++i) r[i + 1] = v[i];
      float l = __shfl_up(v[VO - 1], 1, 64), rr = __shfl_down(v[0], 1, 64);
      if (edge_l) l = glob_l ? ldf(pl, ro[k] + 2 * w0 - 1) : v[0];
      if (edge_r) rr = glob_r ? ldf(pl, ro[k] + 2 * w0 + VO) : v[VO - 1];
      r[0] = l;
      r[VO + 1] = rr;
#pragma unroll
      for (int j = 0; j < VI; ++j)
        V[j] = fmaf(ch[k], 0.25f * r[2 * j] + 0.75f * r[2 * j + 1] + 0.75f * r[2 * j + 2] + 0.25f * r[2 * j + 3], V[j]);
    }
  };
  uint4 re[4], rodd[4], ne[4], no[4];                    // planes 2q, 2q + 1 of this step and of the next one
  request(2 * (d_begin - 1), re);
  request(2 * (d_begin - 1) + 1, rodd);
  for (int q = d_begin - 1; q <= d_end; ++q) {          // block-uniform
    request(2 * q + 2, ne);
    request(2 * q + 3, no);
    uint2 yraw = {0u, 0u};                               // RED: the y0 run of the plane this step completes, requested up front
    if (RED) yraw = ldraw8(y0p, (long long)min(max(q - 1, 0), D - 1) * hw + (long long)hc * W + wc);
    float V[VI];
    if (2 * q >= 2 * d_begin - 1) {                      // even plane 2q: .75 -> q, .25 -> q-1
      plane(2 * q, re, V);
#pragma unroll
      for (int j = 0; j < VI; ++j) { acc_cur[j] = fmaf(0.75f, V[j], acc_cur[j]); acc_prev[j] = fmaf(0.25f, V[j], acc_prev[j]); }
    }
    const int d = q - 1;                                 // complete once plane 2q = 2d+2 is in
    if (d >= d_begin && d < d_end && active) {
      float o[VI];
      const long long sp = (long long)d * hw + (long long)h * W + w0;
      if (accumulate) {
        ldhalf(dst, sp, o);
#pragma unroll
        for (int j = 0; j < VI; ++j) o[j] += acc_prev[j];
      } else {
#pragma unroll
        for (int j = 0; j < VI; ++j) o[j] = acc_prev[j];
      }
      sthalf(dst, sp, o);
      if (RED) {
        float yv[VI];
        cvtraw(y0p, yraw, yv);
#pragma unroll
        for (int j = 0; j < VI; ++j) {
          const float gg = up_stored(dst, o[j]) * (fmaf(yv[j], ra, rb) > 0.f ? 1.f : ur.slope);
          rs0 += gg;
          rs1 = fmaf(gg, yv[j], rs1);
        }
      }
    }
    if (2 * q + 1 <= 2 * d_end) {                        // odd plane 2q+1: .75 -> q, .25 -> q+1
      plane(2 * q + 1, rodd, V);
#pragma unroll
      for (int j = 0; j < VI; ++j) { acc_cur[j] = fmaf(0.75f, V[j], acc_cur[j]); acc_next[j] = fmaf(0.25f, V[j], acc_next[j]); }
    }
#pragma unroll
    for (int j = 0; j < VI; ++j) { acc_prev[j] = acc_cur[j]; acc_cur[j] = acc_next[j]; acc_next[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { re[k] = ne[k]; rodd[k] = no[k]; }
  }
  if (RED) {
    // a lane's partial covers at most sd * VI values in fp32; fp64 from the workgroup level on, as in act_bwd_reduce_kernel
    __shared__ double s_red[4 * 2];
    double v[2] = {(double)rs0, (double)rs1};
    block_sum_d<2>(v, s_red, 4);
    if (tid < 2) atomicAdd(&ur.red[((long long)n * C + c) * 2 + tid], s_red[tid]);
  }
}

template <typename T, int TXN, bool RED = false>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ dy, long long dy_bs, T* __restrict__ dx, long long dx_bs, int C, int D,
                                                            int H, int W, int sd, int tilesW, int tilesH, int accumulate, const UpRed ur) {
  upsample2x_bwd_body<T, TXN, RED>(uint3{blockIdx.x, blockIdx.y, blockIdx.z}, dy, dy_bs, dx, dx_bs, C, D, H, W, sd, tilesW, tilesH, accumulate, ur);
}

template <typename T>
static bool upsample2x_plan(int N, int C, int D, int H, int W, long long a_bs, long long b_bs, int& txn, int& tilesW,
                            int& tilesH, int& sd, int& dsegs) {
  constexpr int VO = VWT<T>::v, VI = VO / 2;
  if (W % VI || a_bs % VO || b_bs % VO || ((long long)D * H * W) % VO) return false;
  if (C > 65535 || N > 65535) return false;
  txn = 4;
  while (txn < 64 && txn * VI < W) txn *= 2;
  tilesW = (W + txn * VI - 1) / (txn * VI);
  tilesH = (H + 256 / txn - 1) / (256 / txn);
  const int base = tilesW * tilesH * C * N;
  dsegs = (1024 + base - 1) / base;
  if (dsegs > D / 4) dsegs = D / 4;
  if (dsegs < 1) dsegs = 1;
  sd = (D + dsegs - 1) / dsegs;
  dsegs = (D + sd - 1) / sd;
  return true;
}
#define UP2X_LAUNCH(KERNEL, T, F, ...)                                                                                  \
  switch (txn) {                                                                                                        \
    case 4: hipLaunchKernelGGL((KERNEL<T, 4, F>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;          \
    case 8: hipLaunchKernelGGL((KERNEL<T, 8, F>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;          \
    case 16: hipLaunchKernelGGL((KERNEL<T, 16, F>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;        \
    case 32: hipLaunchKernelGGL((KERNEL<T, 32, F>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;        \
    default: hipLaunchKernelGGL((KERNEL<T, 64, F>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);               \
  }
template <typename T>
static int upsample2x_fwd_try(void* stream, const void* x, long long x_bs, void* y, long long y_bs, int N, int C, int D, int H,
                              int W, const UpFin* fin = nullptr) {
  int txn, tilesW, tilesH, sd, dsegs;
  if (!upsample2x_plan<T>(N, C, D, H, W, x_bs, y_bs, txn, tilesW, tilesH, sd, dsegs)) return 1;
  dim3 grid(tilesW * tilesH * dsegs, C, N);
  if (fin) {
    UP2X_LAUNCH(upsample2x_fwd_kernel, T, true, (const T*)x, x_bs, (T*)y, y_bs, C, D, H, W, sd, tilesW, tilesH, *fin)
  } else {
    UpFin none{};
    UP2X_LAUNCH(upsample2x_fwd_kernel, T, false, (const T*)x, x_bs, (T*)y, y_bs, C, D, H, W, sd, tilesW, tilesH, none)
  }
  return xh_launch_status();
}
template <typename T>
static int upsample2x_bwd_try(void* stream, const void* dy, long long dy_bs, void* dx, long long dx_bs, int N, int C, int D,
                              int H, int W, int accumulate, const UpRed* ur = nullptr) {
  int txn, tilesW, tilesH, sd, dsegs;
  if (!upsample2x_plan<T>(N, C, D, H, W, dx_bs, dy_bs, txn, tilesW, tilesH, sd, dsegs)) return 1;
  dim3 grid(tilesW * tilesH * dsegs, C, N);
  if (ur) {
    UP2X_LAUNCH(upsample2x_bwd_kernel, T, true, (const T*)dy, dy_bs, (T*)dx, dx_bs, C, D, H, W, sd, tilesW, tilesH, accumulate, *ur)
  } else {
    UpRed none{};
    UP2X_LAUNCH(upsample2x_bwd_kernel, T, false, (const T*)dy, dy_bs, (T*)dx, dx_bs, C, D, H, W, sd, tilesW, tilesH, accumulate, none)
  }
  return xh_launch_status();
}

// BasicConv's InstanceNorm + LeakyReLU applied inside the exact-2x trilinear upsampling that follows it (RA_HVED.py:599-601):
// y = up2x(leaky(IN(x))) with IN finalised from the raw channel sums red[n][c] = (sum x, sum x^2) of the conv epilogue; sc / sh /
// mean / rstd (N x C) are written for the backward pass.  Returns 1 (nothing launched) when the exact-2x kernel does not take
// the layout: the caller then runs xh_in_affine_act + xh_upsample_trilinear_fwd.
extern "C" int xh_upsample2x_in_act_fwd(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs, int N, int C,
                                        int D, int H, int W, const double* red, float slope, float* sc, float* sh, float* mean,
                                        float* rstd) {
  if (!x || !y || !red || !sc || !sh || !mean || !rstd || N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return XH_ERR_ARG;
  if (g_xh_disable & 2) return 1;
  UpFin f{red, 1.0 / ((double)D * H * W), slope, sc, sh, mean, rstd};
  XH_DISPATCH_T(dtype, return upsample2x_fwd_try<T>(stream, x, x_bs, y, y_bs, N, C, D, H, W, &f););
}
// ... and its adjoint: dx = up2x^T(dy) (N, C, D, H, W) together with red[n][c] += (sum dz, sum dz * y0), dz = dx * leaky'(y0 * sc
// + sh): what xh_upsample_trilinear_bwd + xh_act_bwd_reduce leave.  red is accumulated into (caller zeroes).  Returns 1 when
// the exact-2x kernel does not take the layout.
extern "C" int xh_upsample2x_bwd_act_reduce(void* stream, int dtype, const void* dy, long long dy_bs, void* dx, long long dx_bs, int N,
                                            int C, int D, int H, int W, const void* y0, long long y0_bs, const float* sc,
                                            const float* sh, float slope, double* red) {
  if (!dy || !dx || !y0 || !sc || !sh || !red || N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return XH_ERR_ARG;
  if (g_xh_disable & 2) return 1;
  const int vo = dtype == XH_F32 ? 4 : 8;
  if (y0_bs % vo) return 1;
  UpRed r{y0, y0_bs, sc, sh, slope, red};
  XH_DISPATCH_T(dtype, return upsample2x_bwd_try<T>(stream, dy, dy_bs, dx, dx_bs, N, C, D, H, W, 0, &r););
}

extern "C" int xh_upsample_trilinear_fwd(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs,
                                         int N, int C, int D, int H, int W, int Do, int Ho, int Wo) {
  if (!x || !y || N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return XH_ERR_ARG;
  const long long total = (long long)N * C * Do * Ho * Wo;
  if (Do == 2 * D && Ho == 2 * H && Wo == 2 * W && (dtype == XH_F32 || dtype == XH_BF16 || dtype == XH_F16) && !(g_xh_disable & 2)) {
    const int r = dtype == XH_F32    ? upsample2x_fwd_try<float>(stream, x, x_bs, y, y_bs, N, C, D, H, W)
                  : dtype == XH_BF16 ? upsample2x_fwd_try<bf16_t>(stream, x, x_bs, y, y_bs, N, C, D, H, W)
                                     : upsample2x_fwd_try<f16_t>(stream, x, x_bs, y, y_bs, N, C, D, H, W);
    if (r != 1) return r;
  }
  if (dtype == XH_F32)
    hipLaunchKernelGGL(upsample_fwd_kernel<float>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_bs, (float*)y, y_bs, C, D, H, W, Do, Ho, Wo, total);
  else if (dtype == XH_BF16)
    hipLaunchKernelGGL(upsample_fwd_kernel<bf16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (bf16_t*)y, y_bs, C, D, H, W, Do, Ho, Wo, total);
  else if (dtype == XH_F16)
    hipLaunchKernelGGL(upsample_fwd_kernel<f16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (f16_t*)y, y_bs, C, D, H, W, Do, Ho, Wo, total);
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}
extern "C" int xh_upsample_trilinear_bwd(void* stream, int dtype, const void* dy, long long dy_bs, void* dx, long long dx_bs,
                                         int N, int C, int D, int H, int W, int Do, int Ho, int Wo, int accumulate) {
  if (!dy || !dx || N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return XH_ERR_ARG;
  if (Do > 3 * D || Ho > 3 * H || Wo > 3 * W) return XH_ERR_ARG;     // adjoint keeps <= 8 candidate outputs per axis
  const long long total = (long long)N * C * D * H * W;
  if (Do == 2 * D && Ho == 2 * H && Wo == 2 * W && (dtype == XH_F32 || dtype == XH_BF16 || dtype == XH_F16) && !(g_xh_disable & 2)) {
    const int r = dtype == XH_F32    ? upsample2x_bwd_try<float>(stream, dy, dy_bs, dx, dx_bs, N, C, D, H, W, accumulate)
                  : dtype == XH_BF16 ? upsample2x_bwd_try<bf16_t>(stream, dy, dy_bs, dx, dx_bs, N, C, D, H, W, accumulate)
                                     : upsample2x_bwd_try<f16_t>(stream, dy, dy_bs, dx, dx_bs, N, C, D, H, W, accumulate);
    if (r != 1) return r;
  }
  if (dtype == XH_F32)
    hipLaunchKernelGGL(upsample_bwd_kernel<float>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, dy_bs, (float*)dx, dx_bs, C, D, H, W, Do, Ho, Wo, total, accumulate);
  else if (dtype == XH_BF16)
    hipLaunchKernelGGL(upsample_bwd_kernel<bf16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, dy_bs, (bf16_t*)dx, dx_bs, C, D, H, W, Do, Ho, Wo, total, accumulate);
  else if (dtype == XH_F16)
    hipLaunchKernelGGL(upsample_bwd_kernel<f16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)dy, dy_bs, (f16_t*)dx, dx_bs, C, D, H, W, Do, Ho, Wo, total, accumulate);
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- add / act bwd
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void add_kernel(const T* a, long long a_bs, const T* b, long long b_bs, T* y,
                                                      long long y_bs, long long dhw) {
  ROW_LOOP_BEGIN
    (void)c;
    float u[VW], v[VW];
    ldrow<VEC>(a + n * a_bs, q, valid, u);
    if (b) {
      ldrow<VEC>(b + n * b_bs, q, valid, v);
#pragma unroll
      for (int i = 0; i < VW; ++i) u[i] += v[i];
    }
    strow<VEC>(y + n * y_bs, q, valid, u);
  ROW_LOOP_END
}
extern "C" int xh_add(void* stream, int dtype, const void* a, long long a_bs, const void* b, long long b_bs, void* y,
                      long long y_bs, int N, long long CDHW) {
  if (!a || !y || N <= 0 || CDHW <= 0 || N > 65535) return XH_ERR_ARG;
  const bool vec32 = vec_ok<float>(CDHW, {a_bs, b_bs, y_bs}), vec16 = vec_ok<bf16_t>(CDHW, {a_bs, b_bs, y_bs});
  const dim3 grid32 = row_grid<float>(CDHW, 1, N), grid16 = row_grid<bf16_t>(CDHW, 1, N);
  if (dtype == XH_F32)
    { if (vec32) hipLaunchKernelGGL((add_kernel<float, true>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)a, a_bs, (const float*)b, b_bs, (float*)y, y_bs, CDHW); else hipLaunchKernelGGL((add_kernel<float, false>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)a, a_bs, (const float*)b, b_bs, (float*)y, y_bs, CDHW); }
  else if (dtype == XH_BF16)
    { if (vec16) hipLaunchKernelGGL((add_kernel<bf16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)a, a_bs, (const bf16_t*)b, b_bs, (bf16_t*)y, y_bs, CDHW); else hipLaunchKernelGGL((add_kernel<bf16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)a, a_bs, (const bf16_t*)b, b_bs, (bf16_t*)y, y_bs, CDHW); }
  else if (dtype == XH_F16)
    { if (vec16) hipLaunchKernelGGL((add_kernel<f16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)a, a_bs, (const f16_t*)b, b_bs, (f16_t*)y, y_bs, CDHW); else hipLaunchKernelGGL((add_kernel<f16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)a, a_bs, (const f16_t*)b, b_bs, (f16_t*)y, y_bs, CDHW); }
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* dy, const T* y, T* dx, long long n, int act) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float yv = ldf(y, i), g = ldf(dy, i);
    float o = g;
    if (act == XH_ACT_RELU) o = yv > 0.f ? g : 0.f;
    else if (act == XH_ACT_SIGMOID) o = g * yv * (1.f - yv);
    stf(dx, i, o);
  }
}
extern "C" int xh_act_bwd(void* stream, int dtype, const void* dy, const void* y, void* dx, long long n, int act) {
  if (!dy || !y || !dx || n <= 0) return XH_ERR_ARG;
  if (dtype == XH_F32)
    hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)y, (float*)dx, n, act);
  else if (dtype == XH_BF16)
    hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, n, act);
  else if (dtype == XH_F16)
    hipLaunchKernelGGL(act_bwd_kernel<f16_t>, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)dy, (const f16_t*)y, (f16_t*)dx, n, act);
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- init-block fold
// The init blocks are 1x1 convs from ONE modality to B channels (RA_HVED.py:345-349): X_c = w_c x_m + b_c, and their only consumer
// is an InstanceNorm (the first SingleConv of the encoder, buildingblocks.py:406-433).  IN(X_c) = (x_m - mean_m) w_c R_c with
// R_c = 1 / sqrt(w_c^2 var_m + eps): an affine of the INPUT -- so the 16-channel tensor X is never stored; the first conv reads x
// through (sc, sh) per logical channel (xh_conv_desc.bcast).  These two parameter-sized kernels are the bookkeeping.
struct FoldW { const float* w[XH_MAX_WPTR]; float* dw[XH_MAX_WPTR]; };
__global__ __launch_bounds__(64) void init_fold_fwd_k(const double* __restrict__ red_x, double inv_count, int N, int M, int B, FoldW fw, float eps,
                                                      float* sc, float* sh, float* rstd, float* ctr) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= N * M * B) return;
  const int c = i % (M * B), n = i / (M * B), m = c / B;
  const double mean = red_x[2 * (n * M + m)] * inv_count;
  double var = red_x[2 * (n * M + m) + 1] * inv_count - mean * mean;
  if (var < 0) var = 0;
  const double w = (double)fw.w[m][c - m * B];
  const double R = 1.0 / sqrt(w * w * var + (double)eps);
  sc[i] = (float)(w * R);
  sh[i] = (float)(-w * mean * R);
  rstd[i] = (float)R;
  ctr[i] = (float)mean;
}
// backward: the data gradient of the first conv left S0 = sum g, S1 = sum g (x - ctr) per logical channel (g = d loss / d IN(X_c),
// masked by the activation; e = the raw input, centred on ctr = fp32(mean_m): xh_conv_ptrs.e_ctr -- sum g x - mean sum g would cancel
// five digits).  Chain rule through mean and variance of X_c = w_c x + b_c:
//   d loss / d w_c = eps R_c^3 sum g (x - mean_m) = eps R_c^3 (S1 - (mean_m - ctr) S0),   d loss / d b_c = 0
// (both exactly; for |w_c| sigma << sqrt(eps) the factor eps R^3 reaches eps^-1/2: these gradients are NOT small)
__global__ __launch_bounds__(64) void init_fold_bwd_k(const double* __restrict__ red_x, double inv_count, int N, int M, int B, FoldW fw, float eps,
                                                      const double* __restrict__ red_g) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= M * B) return;
  const int m = c / B;
  const double w = (double)fw.w[m][c - m * B];
  double acc = 0.0;
  for (int n = 0; n < N; ++n) {
    const double mean = red_x[2 * (n * M + m)] * inv_count;
    double var = red_x[2 * (n * M + m) + 1] * inv_count - mean * mean;
    if (var < 0) var = 0;
    const double R = 1.0 / sqrt(w * w * var + (double)eps);
    const double S0 = red_g[2 * (n * M * B + c)], S1 = red_g[2 * (n * M * B + c) + 1];
    acc += (double)eps * R * R * R * (S1 - (mean - (double)(float)mean) * S0);
  }
  fw.dw[m][c - m * B] += (float)acc;
}
extern "C" int xh_init_fold_fwd(void* stream, const double* red_x, long long count, int N, int M, int B, const float* const w[XH_MAX_WPTR],
                                float eps, float* sc, float* sh, float* rstd, float* ctr) {
  if (!red_x || count <= 0 || N <= 0 || M <= 0 || M > XH_MAX_WPTR || B <= 0 || !w || !sc || !sh || !rstd || !ctr) return XH_ERR_ARG;
  FoldW fw;
  for (int i = 0; i < XH_MAX_WPTR; ++i) { fw.w[i] = i < M ? w[i] : nullptr; fw.dw[i] = nullptr; if (i < M && !w[i]) return XH_ERR_ARG; }
  hipLaunchKernelGGL(init_fold_fwd_k, dim3((N * M * B + 63) / 64), dim3(64), 0, (hipStream_t)stream, red_x, 1.0 / (double)count, N, M, B, fw, eps,
                     sc, sh, rstd, ctr);
  return xh_launch_status();
}
extern "C" int xh_init_fold_bwd(void* stream, const double* red_x, long long count, int N, int M, int B, const float* const w[XH_MAX_WPTR],
                                float eps, const double* red_g, float* const dw[XH_MAX_WPTR]) {
  if (!red_x || count <= 0 || N <= 0 || M <= 0 || M > XH_MAX_WPTR || B <= 0 || !w || !red_g || !dw) return XH_ERR_ARG;
  FoldW fw;
  for (int i = 0; i < XH_MAX_WPTR; ++i) {
    fw.w[i] = i < M ? w[i] : nullptr; fw.dw[i] = i < M ? dw[i] : nullptr;
    if (i < M && (!w[i] || !dw[i])) return XH_ERR_ARG;
  }
  hipLaunchKernelGGL(init_fold_bwd_k, dim3((M * B + 63) / 64), dim3(64), 0, (hipStream_t)stream, red_x, 1.0 / (double)count, N, M, B, fw, eps, red_g);
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- product of experts
#define POE_EPS 1e-8f
// Philox4x32-10 (Salmon et al., SC'11; the generator behind torch's device RNG): counter-based, so element i of draw c is a pure
// function of (seed, c, stream, i) -- the backward pass regenerates the forward's noise instead of reading it from HBM.
struct PhiloxKey { unsigned long long seed, ctr; int stream; };
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ void philox_block(const PhiloxKey& k, long long i, unsigned (&o)[4]) {
  philox4x32_10((unsigned)i, (unsigned)((unsigned long long)i >> 32) | ((unsigned)k.stream << 24), (unsigned)k.ctr, (unsigned)(k.ctr >> 32),
                (unsigned)k.seed, (unsigned)(k.seed >> 32), o);
}
// one standard normal per element (Box-Muller on two 24-bit uniforms in (0, 1): |eps| <= 5.8)
__device__ __forceinline__ float philox_normal(const PhiloxKey& k, long long i) {
  unsigned o[4];
  philox_block(k, i, o);
  const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.f / 16777216.f), u2 = ((float)(o[1] >> 8) + 0.5f) * (1.f / 16777216.f);
  // v_log_f32 (log2) and v_cos_f32 (argument in revolutions: cos(2 pi u2) is one instruction) instead of the library's logf / cosf,
  // whose argument reduction made the PoE launches 10 us longer than the pass they replace a noise tensor in
  return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
}
template <typename T>
__device__ __forceinline__ void poe_fwd_body(const T* feat, const float* keep, const T* eps, T* z, T* mu_stack, T* lv_stack, int L,
                                             long long dhw, long long total, int mask_mu, int bx, int gdx, bool draw = false, PhiloxKey rk = PhiloxKey{0, 0, 0}) {
  for (long long i = (long long)bx * 256 + threadIdx.x; i < total; i += (long long)gdx * 256) {
    const long long p = i % dhw;
    const int l = (int)((i / dhw) % L);
    const int n = (int)(i / (dhw * L));
    float tsum = 1.f / (1.f + POE_EPS), musum = 0.f;
    const long long so = (((long long)n * 5) * L + l) * dhw + p;   // stack offset of expert 0
    stf(mu_stack, so, 0.f);
    stf(lv_stack, so, 0.f);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const long long fo = (((long long)n * 4 + m) * 2 * L + l) * dhw + p;
      const float mu = ldf(feat, fo);
      float lv = ldf(feat, fo + (long long)L * dhw);
      lv = fminf(fmaxf(lv, -50.f), 50.f);
      const float k = keep[n * 4 + m];
      const float t = k / (__expf(lv) + POE_EPS);
      tsum += t;
      musum = fmaf(mu, t, musum);
      stf(mu_stack, so + (long long)(m + 1) * L * dhw, mask_mu ? mu * k : mu);
      stf(lv_stack, so + (long long)(m + 1) * L * dhw, lv);
    }
    const float pmu = musum / tsum;
    float out = pmu;
    if (eps) out = fmaf(ldf(eps, i), __expf(-0.5f * __logf(tsum)), pmu);
    else if (draw) out = fmaf(philox_normal(rk, i), __expf(-0.5f * __logf(tsum)), pmu);
    stf(z, i, out);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void poe_fwd_kernel(const T* feat, const float* keep, const T* eps, T* z, T* mu_stack,
                                                     T* lv_stack, int L, long long dhw, long long total, int mask_mu) {
  poe_fwd_body<T>(feat, keep, eps, z, mu_stack, lv_stack, L, dhw, total, mask_mu, blockIdx.x, gridDim.x);
}
template <typename T>
__device__ __forceinline__ void poe_bwd_body(const T* feat, const float* keep, const T* eps, const T* dz, const T* dmu_stack,
                                             const T* dlv_stack, T* dfeat, int L, long long dhw, long long total, int mask_mu, int bx,
                                             int gdx, bool draw = false, PhiloxKey rk = PhiloxKey{0, 0, 0}) {
  for (long long i = (long long)bx * 256 + threadIdx.x; i < total; i += (long long)gdx * 256) {
    const long long p = i % dhw;
    const int l = (int)((i / dhw) % L);
    const int n = (int)(i / (dhw * L));
    float mu[4], lvr[4], lv[4], t[4], k[4], ex[4];
    float tsum = 1.f / (1.f + POE_EPS), musum = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const long long fo = (((long long)n * 4 + m) * 2 * L + l) * dhw + p;
      mu[m] = ldf(feat, fo);
      lvr[m] = ldf(feat, fo + (long long)L * dhw);
      lv[m] = fminf(fmaxf(lvr[m], -50.f), 50.f);
      k[m] = keep[n * 4 + m];
      ex[m] = __expf(lv[m]);
      t[m] = k[m] / (ex[m] + POE_EPS);
      tsum += t[m];
      musum = fmaf(mu[m], t[m], musum);
    }
    const float pmu = musum / tsum;
    const float g = ldf(dz, i);
    float dlvp = 0.f;
    if (eps) dlvp = g * ldf(eps, i) * 0.5f * __expf(-0.5f * __logf(tsum));
    else if (draw) dlvp = g * philox_normal(rk, i) * 0.5f * __expf(-0.5f * __logf(tsum));
    const float dtsum = -dlvp / tsum - g * pmu / tsum;
    const long long so = (((long long)n * 5) * L + l) * dhw + p;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const long long fo = (((long long)n * 4 + m) * 2 * L + l) * dhw + p;
      float dmu = g * t[m] / tsum;
      float dlv = (g * mu[m] / tsum + dtsum) * (-t[m] * ex[m] / (ex[m] + POE_EPS));
      if (dmu_stack) dmu += ldf(dmu_stack, so + (long long)(m + 1) * L * dhw) * (mask_mu ? k[m] : 1.f);
      if (dlv_stack) dlv += ldf(dlv_stack, so + (long long)(m + 1) * L * dhw);
      if (lvr[m] < -50.f || lvr[m] > 50.f) dlv = 0.f;
      stf(dfeat, fo, dmu);
      stf(dfeat, fo + (long long)L * dhw, dlv);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void poe_bwd_kernel(const T* feat, const float* keep, const T* eps, const T* dz,
                                                     const T* dmu_stack, const T* dlv_stack, T* dfeat, int L, long long dhw,
                                                     long long total, int mask_mu) {
  poe_bwd_body<T>(feat, keep, eps, dz, dmu_stack, dlv_stack, dfeat, L, dhw, total, mask_mu, blockIdx.x, gridDim.x);
}
// The PoE of every latent level of a forward pass (RA_HVED.py:573-597 runs it per level; the levels are independent functions of
// the encoder outputs) in one launch per direction: four launches of 2 - 1 024 workgroups were four launch floors.
struct PoeMulti {
  int n, bwd;
  int off[XH_POE_MAX + 1];
  xh_poe_job j[XH_POE_MAX];
  unsigned long long* rng;        // forward with in-kernel noise: {seed, counter, ticket, -}; else NULL
};
template <typename T>
__global__ __launch_bounds__(256) void poe_multi_kernel(const PoeMulti m) {
  int pi = 0;
  for (int k = 1; k < XH_POE_MAX; ++k)
    if (k < m.n && (int)blockIdx.x >= m.off[k]) pi = k;
  const xh_poe_job& j = m.j[pi];
  const int bx = blockIdx.x - m.off[pi], gdx = m.off[pi + 1] - m.off[pi];
  const long long total = (long long)j.N * j.L * j.dhw;
  // in-kernel noise: forward reads the generator's counter (every workgroup, before it takes its ticket below -- the counter moves
  // only after ALL tickets are taken); backward reads the word its forward left in rng_used
  PhiloxKey rk = {0ull, 0ull, 0};
  const bool draw = !j.eps && j.rng_used && (m.bwd || m.rng);
  if (m.rng) { rk.seed = m.rng[0]; rk.ctr = m.rng[1]; }
  if (m.bwd && draw) { rk.seed = j.rng_used[1]; rk.ctr = j.rng_used[0]; }
  rk.stream = j.rng_stream;
  if (!m.bwd) poe_fwd_body<T>((const T*)j.feat, j.keep, (const T*)j.eps, (T*)j.z, (T*)j.mu_stack, (T*)j.lv_stack, j.L, j.dhw, total, j.mask_mu, bx, gdx, draw, rk);
  else poe_bwd_body<T>((const T*)j.feat, j.keep, (const T*)j.eps, (const T*)j.dz, (const T*)j.dmu_stack, (const T*)j.dlv_stack, (T*)j.dfeat, j.L, j.dhw, total, j.mask_mu, bx, gdx, draw, rk);
  if (m.rng) {                                                     // (uniform per launch)
    // No fences: the ticket only has to come after this workgroup's READS of the counter, and those have completed -- their values
    // went into the stores above (the barrier collects the lanes); the words written below are read by LATER launches only.
    __syncthreads();
    if (threadIdx.x == 0) {
      // two-level ticket: 32 first-level words on cache lines of their own (1 360 tickets on ONE word serialise at the memory side:
      // +16 us on this launch), the workgroup that completes a word takes a second-level ticket; the last of those is the last
      // workgroup of the launch.  Words are left zero for the next launch.
      const unsigned G = gridDim.x, nl = G < XH_RNG_LINES ? G : XH_RNG_LINES, l = blockIdx.x % XH_RNG_LINES;
      unsigned long long* w1 = m.rng + 16 * (2 + l);
      unsigned long long* w2 = m.rng + 16;
      bool last = false;
      if (atomicAdd(w1, 1ull) == (unsigned long long)((G - l + XH_RNG_LINES - 1) / XH_RNG_LINES) - 1) {
        *w1 = 0;
        if (atomicAdd(w2, 1ull) == (unsigned long long)nl - 1) { *w2 = 0; last = true; }
      }
      if (last) {                                                  // the last workgroup: record the draw, advance the generator
        for (int k = 0; k < m.n; ++k)
          if (m.j[k].rng_used) {
            unsigned long long* u = const_cast<unsigned long long*>(m.j[k].rng_used);
            u[0] = rk.ctr; u[1] = rk.seed;
          }
        m.rng[1] = rk.ctr + 1;
      }
    }
  }
}
extern "C" int xh_poe_multi(void* stream, int dtype, int bwd, int n, const xh_poe_job* jobs, unsigned long long* rng) {
  if (n <= 0 || n > XH_POE_MAX || !jobs) return XH_ERR_ARG;
  if (bwd && rng) return XH_ERR_ARG;
  PoeMulti m;
  m.n = n; m.bwd = bwd ? 1 : 0; m.off[0] = 0;
  m.rng = rng;
  for (int i = 0; i < n; ++i) {
    const xh_poe_job& j = jobs[i];
    if (!j.feat || !j.keep || j.N <= 0 || j.L <= 0 || j.dhw <= 0) return XH_ERR_ARG;
    if (bwd ? (!j.dz || !j.dfeat) : (!j.z || !j.mu_stack || !j.lv_stack)) return XH_ERR_ARG;
    if (!bwd && !j.eps && j.rng_used && !rng) return XH_ERR_ARG;   // a drawing job needs the generator state
    if (j.rng_stream < 0 || j.rng_stream > 255) return XH_ERR_ARG;
    m.j[i] = j;
    int nb = flat_grid((long long)j.N * j.L * j.dhw);
    m.off[i + 1] = m.off[i] + nb;
  }
  for (int i = n; i < XH_POE_MAX; ++i) m.off[i + 1] = m.off[n];
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL(poe_multi_kernel<T>, dim3(m.off[n]), dim3(256), 0, (hipStream_t)stream, m););
  return xh_launch_status();
}
__global__ __launch_bounds__(256) void philox_normal_kernel(PhiloxKey k, void* out, long long n, int raw) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    if (raw) {
      unsigned o[4];
      philox_block(k, i, o);
      reinterpret_cast<uint4*>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
      reinterpret_cast<float*>(out)[i] = philox_normal(k, i);
    }
  }
}
extern "C" int xh_philox_normal(void* stream, unsigned long long seed, unsigned long long counter, int rng_stream, void* out, long long n, int raw) {
  if (!out || n <= 0 || rng_stream < 0 || rng_stream > 255) return XH_ERR_ARG;
  PhiloxKey k;
  k.seed = seed; k.ctr = counter; k.stream = rng_stream;
  hipLaunchKernelGGL(philox_normal_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, k, out, n, raw);
  return xh_launch_status();
}
extern "C" int xh_poe_fwd(void* stream, int dtype, const void* feat, const float* keep, const void* eps, void* z,
                          void* mu_stack, void* lv_stack, int N, int L, long long dhw, int mask_mu) {
  if (!feat || !keep || !z || !mu_stack || !lv_stack || N <= 0 || L <= 0 || dhw <= 0) return XH_ERR_ARG;
  const long long total = (long long)N * L * dhw;
  if (dtype == XH_F32)
    hipLaunchKernelGGL(poe_fwd_kernel<float>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)feat, keep, (const float*)eps, (float*)z, (float*)mu_stack, (float*)lv_stack, L, dhw, total, mask_mu);
  else if (dtype == XH_BF16)
    hipLaunchKernelGGL(poe_fwd_kernel<bf16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feat, keep, (const bf16_t*)eps, (bf16_t*)z, (bf16_t*)mu_stack, (bf16_t*)lv_stack, L, dhw, total, mask_mu);
  else if (dtype == XH_F16)
    hipLaunchKernelGGL(poe_fwd_kernel<f16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)feat, keep, (const f16_t*)eps, (f16_t*)z, (f16_t*)mu_stack, (f16_t*)lv_stack, L, dhw, total, mask_mu);
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}
extern "C" int xh_poe_bwd(void* stream, int dtype, const void* feat, const float* keep, const void* eps, const void* dz,
                          const void* dmu_stack, const void* dlv_stack, void* dfeat, int N, int L, long long dhw,
                          int mask_mu) {
  if (!feat || !keep || !dz || !dfeat || N <= 0 || L <= 0 || dhw <= 0) return XH_ERR_ARG;
  const long long total = (long long)N * L * dhw;
  if (dtype == XH_F32)
    hipLaunchKernelGGL(poe_bwd_kernel<float>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)feat, keep, (const float*)eps, (const float*)dz, (const float*)dmu_stack, (const float*)dlv_stack, (float*)dfeat, L, dhw, total, mask_mu);
  else if (dtype == XH_BF16)
    hipLaunchKernelGGL(poe_bwd_kernel<bf16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feat, keep, (const bf16_t*)eps, (const bf16_t*)dz, (const bf16_t*)dmu_stack, (const bf16_t*)dlv_stack, (bf16_t*)dfeat, L, dhw, total, mask_mu);
  else if (dtype == XH_F16)
    hipLaunchKernelGGL(poe_bwd_kernel<f16_t>, dim3(flat_grid(total)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)feat, keep, (const f16_t*)eps, (const f16_t*)dz, (const f16_t*)dmu_stack, (const f16_t*)dlv_stack, (f16_t*)dfeat, L, dhw, total, mask_mu);
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- channel pool / gates
// One lane per 16-byte run of voxels (8 x 16-bit / 4 x fp32; element-wise tail when the layout does not allow the wide
// accesses); loops over channels (C <= 128).  grid: (chunks over the runs, 1, N)
#define VOX_LOOP_BEGIN                                                                                               \
  constexpr int VW = VWT<T>::v;                                                                                      \
  const int n = blockIdx.z;                                                                                          \
  const long long nrun = (dhw + VW - 1) / VW;                                                                        \
  for (long long run = (long long)blockIdx.x * 256 + threadIdx.x; run < nrun; run += (long long)gridDim.x * 256) {   \
    const long long q = run * VW;                                                                                    \
    const int valid = (int)min((long long)VW, dhw - q);
#define VOX_LOOP_END }
// The same with VWX voxels per lane when VWX > 0 (the deep levels: two -- four times the lanes of the 16-byte runs, and room in the
// registers for four times the channels per round of loads)
#define VOX_LOOP_BEGIN_W(VWX)                                                                                        \
  constexpr int VW = (VWX) > 0 ? (VWX) : VWT<T>::v;                                                                  \
  const int n = blockIdx.z;                                                                                          \
  const long long nrun = (dhw + VW - 1) / VW;                                                                        \
  for (long long run = (long long)blockIdx.x * 256 + threadIdx.x; run < nrun; run += (long long)gridDim.x * 256) {   \
    const long long q = run * VW;                                                                                    \
    const int valid = (int)min((long long)VW, dhw - q);
// Channels are walked CB at a time with all of a batch's loads issued before the first use: a lane's channel loop is
// otherwise one full memory latency per channel (in-order issue, the wait sits right behind each load).
constexpr int CB = 4;
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void channel_pool_fwd_kernel(const T* __restrict__ x, long long x_bs, T* __restrict__ y,
                                                              long long y_bs, int C, long long dhw) {
  VOX_LOOP_BEGIN
    const T* xp = x + n * x_bs;
    float m[VW], s[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) { m[v] = -INFINITY; s[v] = 0.f; }
    for (int c0 = 0; c0 < C; c0 += CB) {
      float xv[CB][VW];
#pragma unroll
      for (int j = 0; j < CB; ++j) ldrow<VEC>(xp + (long long)min(c0 + j, C - 1) * dhw, q, valid, xv[j]);
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          m[v] = (xv[j][v] > m[v] || xv[j][v] != xv[j][v]) ? xv[j][v] : m[v];
          s[v] += xv[j][v];
        }
              }
      }
    }
#pragma unroll
    for (int v = 0; v < VW; ++v) s[v] = s[v] / (float)C;
    strow<VEC>(y + n * y_bs, q, valid, m);
    strow<VEC>(y + n * y_bs + dhw, q, valid, s);
  VOX_LOOP_END
}
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void channel_pool_bwd_kernel(const T* __restrict__ x, long long x_bs, const T* __restrict__ dy,
                                                              long long dy_bs, T* dx, long long dx_bs, int C, long long dhw,
                                                              int accumulate) {
  VOX_LOOP_BEGIN
    const T* xp = x + n * x_bs;
    float m[VW], g0[VW], g1[VW];
    int arg[VW];
    ldrow<VEC>(dy + n * dy_bs, q, valid, g0);
    ldrow<VEC>(dy + n * dy_bs + dhw, q, valid, g1);
#pragma unroll
    for (int v = 0; v < VW; ++v) { m[v] = -INFINITY; arg[v] = 0; }
    for (int c0 = 0; c0 < C; c0 += CB) {
      float xv[CB][VW];
#pragma unroll
      for (int j = 0; j < CB; ++j) ldrow<VEC>(xp + (long long)min(c0 + j, C - 1) * dhw, q, valid, xv[j]);
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
#pragma unroll
        for (int v = 0; v < VW; ++v)
          if (xv[j][v] > m[v] || xv[j][v] != xv[j][v]) { m[v] = xv[j][v]; arg[v] = c0 + j; }
              }
      }
    }
#pragma unroll
    for (int v = 0; v < VW; ++v) g1[v] = g1[v] / (float)C;
    T* dp = dx + n * dx_bs;
    for (int c0 = 0; c0 < C; c0 += CB) {
      float o[CB][VW];
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (accumulate) {
          ldrow<VEC>((const T*)dp + (long long)min(c0 + j, C - 1) * dhw, q, valid, o[j]);
        } else {
#pragma unroll
          for (int v = 0; v < VW; ++v) o[j][v] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
#pragma unroll
        for (int v = 0; v < VW; ++v) o[j][v] = o[j][v] + g1[v] + (c0 + j == arg[v] ? g0[v] : 0.f);
        strow<VEC>(dp + (long long)(c0 + j) * dhw, q, valid, o[j]);
              }
      }
    }
  VOX_LOOP_END
}
template <typename T>
static inline dim3 vox_grid(long long dhw, int N, int cap = 4096) {
  const long long nrun = (dhw + VWT<T>::v - 1) / VWT<T>::v;
  long long b = (nrun + 255) / 256;
  if (b > cap) b = cap;
  return dim3((unsigned)b, 1, N);
}
// The deep levels: a few dozen workgroups whose lanes walk many channels.  Their run time is the lane's chain of dependent
// load rounds (C / CB of them, a memory latency each), so those launches take the instances with 8 or 16 channels per round.
static inline bool deep_grid(const dim3& g, int C) { return C >= 8 && (long long)g.x * g.y * g.z < 256; }
// Round 6: those instances also take TWO voxels per lane instead of eight (DEEP_VW) and DEEP_CB channels per round: at 32 workgroups
// for the 32^3 level a launch was one wave per SIMD on an eighth of the chip walking 64 - 128 channels in rounds of 8 or 16, twice --
// 20 us of exposed latency for a megabyte of data.  Four times the lanes, a quarter of the rounds.
constexpr int DEEP_VW = 2, DEEP_CB = 32;
template <typename T>
static inline dim3 vox_grid_deep(long long dhw, int N, int cap = 4096) {
  const long long nrun = (dhw + DEEP_VW - 1) / DEEP_VW;
  long long b = (nrun + 255) / 256;
  if (b > cap) b = cap;
  return dim3((unsigned)b, 1, N);
}
extern "C" int xh_channel_pool_fwd(void* stream, int dtype, const void* x, long long x_bs, void* y, long long y_bs, int N,
                                   int C, long long DHW) {
  if (!x || !y || N <= 0 || C <= 0 || DHW <= 0 || N > 65535) return XH_ERR_ARG;
  if (dtype == XH_F32)
    { if (vec_ok<float>(DHW, {x_bs, y_bs})) hipLaunchKernelGGL((channel_pool_fwd_kernel<float, true>), vox_grid<float>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_bs, (float*)y, y_bs, C, DHW); else hipLaunchKernelGGL((channel_pool_fwd_kernel<float, false>), vox_grid<float>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_bs, (float*)y, y_bs, C, DHW); }
  else if (dtype == XH_BF16)
    { if (vec_ok<bf16_t>(DHW, {x_bs, y_bs})) hipLaunchKernelGGL((channel_pool_fwd_kernel<bf16_t, true>), vox_grid<bf16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (bf16_t*)y, y_bs, C, DHW); else hipLaunchKernelGGL((channel_pool_fwd_kernel<bf16_t, false>), vox_grid<bf16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (bf16_t*)y, y_bs, C, DHW); }
  else if (dtype == XH_F16)
    { if (vec_ok<f16_t>(DHW, {x_bs, y_bs})) hipLaunchKernelGGL((channel_pool_fwd_kernel<f16_t, true>), vox_grid<f16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (f16_t*)y, y_bs, C, DHW); else hipLaunchKernelGGL((channel_pool_fwd_kernel<f16_t, false>), vox_grid<f16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (f16_t*)y, y_bs, C, DHW); }
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}
extern "C" int xh_channel_pool_bwd(void* stream, int dtype, const void* x, long long x_bs, const void* dy, long long dy_bs,
                                   void* dx, long long dx_bs, int N, int C, long long DHW, int accumulate) {
  if (!x || !dy || !dx || N <= 0 || C <= 0 || DHW <= 0 || N > 65535) return XH_ERR_ARG;
  if (dtype == XH_F32)
    { if (vec_ok<float>(DHW, {x_bs, dy_bs, dx_bs})) hipLaunchKernelGGL((channel_pool_bwd_kernel<float, true>), vox_grid<float>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_bs, (const float*)dy, dy_bs, (float*)dx, dx_bs, C, DHW, accumulate); else hipLaunchKernelGGL((channel_pool_bwd_kernel<float, false>), vox_grid<float>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_bs, (const float*)dy, dy_bs, (float*)dx, dx_bs, C, DHW, accumulate); }
  else if (dtype == XH_BF16)
    { if (vec_ok<bf16_t>(DHW, {x_bs, dy_bs, dx_bs})) hipLaunchKernelGGL((channel_pool_bwd_kernel<bf16_t, true>), vox_grid<bf16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (const bf16_t*)dy, dy_bs, (bf16_t*)dx, dx_bs, C, DHW, accumulate); else hipLaunchKernelGGL((channel_pool_bwd_kernel<bf16_t, false>), vox_grid<bf16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (const bf16_t*)dy, dy_bs, (bf16_t*)dx, dx_bs, C, DHW, accumulate); }
  else if (dtype == XH_F16)
    { if (vec_ok<f16_t>(DHW, {x_bs, dy_bs, dx_bs})) hipLaunchKernelGGL((channel_pool_bwd_kernel<f16_t, true>), vox_grid<f16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (const f16_t*)dy, dy_bs, (f16_t*)dx, dx_bs, C, DHW, accumulate); else hipLaunchKernelGGL((channel_pool_bwd_kernel<f16_t, false>), vox_grid<f16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (const f16_t*)dy, dy_bs, (f16_t*)dx, dx_bs, C, DHW, accumulate); }
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

// y = x*(1+s)
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void gate_fwd_kernel(const T* x, long long x_bs, const T* s, long long s_bs, T* y,
                                                           long long y_bs, long long dhw) {
  ROW_LOOP_BEGIN
    float xv[VW], sv[VW];
    ldrow<VEC>(x + n * x_bs + (long long)c * dhw, q, valid, xv);
    ldrow<VEC>(s + n * s_bs, q, valid, sv);
#pragma unroll
    for (int i = 0; i < VW; ++i) xv[i] *= (1.f + sv[i]);
    strow<VEC>(y + n * y_bs + (long long)c * dhw, q, valid, xv);
  ROW_LOOP_END
}
// lane per voxel, loop over channels: dx = dy*(1+s), ds = sum_c dy*x
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const T* __restrict__ x, long long x_bs, const T* __restrict__ s,
                                                      long long s_bs, const T* __restrict__ dy, long long dy_bs, T* dx,
                                                      long long dx_bs, T* ds, long long ds_bs, int C, long long dhw,
                                                      int acc_dx, int acc_ds) {
  VOX_LOOP_BEGIN
    float g1[VW], a[VW];
    ldrow<VEC>(s + n * s_bs, q, valid, g1);
#pragma unroll
    for (int v = 0; v < VW; ++v) { g1[v] = 1.f + g1[v]; a[v] = 0.f; }
    for (int c0 = 0; c0 < C; c0 += CB) {
      float g[CB][VW], xv[CB][VW], o[CB][VW];
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const long long off = (long long)min(c0 + j, C - 1) * dhw;
        ldrow<VEC>(dy + n * dy_bs + off, q, valid, g[j]);
        ldrow<VEC>(x + n * x_bs + off, q, valid, xv[j]);
        if (dx && acc_dx) {
          ldrow<VEC>((const T*)dx + n * dx_bs + off, q, valid, o[j]);
        } else {
#pragma unroll
          for (int v = 0; v < VW; ++v) o[j][v] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
#pragma unroll
        for (int v = 0; v < VW; ++v) a[v] = fmaf(g[j][v], xv[j][v], a[v]);
        if (dx) {
#pragma unroll
          for (int v = 0; v < VW; ++v) o[j][v] = g[j][v] * g1[v] + o[j][v];
          strow<VEC>(dx + n * dx_bs + (long long)(c0 + j) * dhw, q, valid, o[j]);
        }
              }
      }
    }
    if (ds) {
      T* sp = ds + n * ds_bs;
      float o[VW];
      if (acc_ds) {
        ldrow<VEC>((const T*)sp, q, valid, o);
      } else {
#pragma unroll
        for (int v = 0; v < VW; ++v) o[v] = 0.f;
      }
#pragma unroll
      for (int v = 0; v < VW; ++v) o[v] = a[v] + o[v];
      strow<VEC>(sp, q, valid, o);
    }
  VOX_LOOP_END
}
extern "C" int xh_gate_fwd(void* stream, int dtype, const void* x, long long x_bs, const void* s, long long s_bs, void* y,
                           long long y_bs, int N, int C, long long DHW) {
  if (!x || !s || !y || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  const bool vec32 = vec_ok<float>(DHW, {x_bs, s_bs, y_bs}), vec16 = vec_ok<bf16_t>(DHW, {x_bs, s_bs, y_bs});
  const dim3 grid32 = row_grid<float>(DHW, C, N), grid16 = row_grid<bf16_t>(DHW, C, N);
  if (dtype == XH_F32)
    { if (vec32) hipLaunchKernelGGL((gate_fwd_kernel<float, true>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)x, x_bs, (const float*)s, s_bs, (float*)y, y_bs, DHW); else hipLaunchKernelGGL((gate_fwd_kernel<float, false>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)x, x_bs, (const float*)s, s_bs, (float*)y, y_bs, DHW); }
  else if (dtype == XH_BF16)
    { if (vec16) hipLaunchKernelGGL((gate_fwd_kernel<bf16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (const bf16_t*)s, s_bs, (bf16_t*)y, y_bs, DHW); else hipLaunchKernelGGL((gate_fwd_kernel<bf16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (const bf16_t*)s, s_bs, (bf16_t*)y, y_bs, DHW); }
  else if (dtype == XH_F16)
    { if (vec16) hipLaunchKernelGGL((gate_fwd_kernel<f16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (const f16_t*)s, s_bs, (f16_t*)y, y_bs, DHW); else hipLaunchKernelGGL((gate_fwd_kernel<f16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (const f16_t*)s, s_bs, (f16_t*)y, y_bs, DHW); }
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}
extern "C" int xh_gate_bwd(void* stream, int dtype, const void* x, long long x_bs, const void* s, long long s_bs,
                           const void* dy, long long dy_bs, void* dx, long long dx_bs, void* ds, long long ds_bs, int N,
                           int C, long long DHW, int acc_dx, int acc_ds) {
  if (!x || !s || !dy || N <= 0 || C <= 0 || DHW <= 0 || N > 65535) return XH_ERR_ARG;
  if (dtype == XH_F32)
    { if (vec_ok<float>(DHW, {x_bs, s_bs, dy_bs, dx_bs, ds_bs})) hipLaunchKernelGGL((gate_bwd_kernel<float, true>), vox_grid<float>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_bs, (const float*)s, s_bs, (const float*)dy, dy_bs, (float*)dx, dx_bs, (float*)ds, ds_bs, C, DHW, acc_dx, acc_ds); else hipLaunchKernelGGL((gate_bwd_kernel<float, false>), vox_grid<float>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_bs, (const float*)s, s_bs, (const float*)dy, dy_bs, (float*)dx, dx_bs, (float*)ds, ds_bs, C, DHW, acc_dx, acc_ds); }
  else if (dtype == XH_BF16)
    { if (vec_ok<bf16_t>(DHW, {x_bs, s_bs, dy_bs, dx_bs, ds_bs})) hipLaunchKernelGGL((gate_bwd_kernel<bf16_t, true>), vox_grid<bf16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (const bf16_t*)s, s_bs, (const bf16_t*)dy, dy_bs, (bf16_t*)dx, dx_bs, (bf16_t*)ds, ds_bs, C, DHW, acc_dx, acc_ds); else hipLaunchKernelGGL((gate_bwd_kernel<bf16_t, false>), vox_grid<bf16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (const bf16_t*)s, s_bs, (const bf16_t*)dy, dy_bs, (bf16_t*)dx, dx_bs, (bf16_t*)ds, ds_bs, C, DHW, acc_dx, acc_ds); }
  else if (dtype == XH_F16)
    { if (vec_ok<f16_t>(DHW, {x_bs, s_bs, dy_bs, dx_bs, ds_bs})) hipLaunchKernelGGL((gate_bwd_kernel<f16_t, true>), vox_grid<f16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (const f16_t*)s, s_bs, (const f16_t*)dy, dy_bs, (f16_t*)dx, dx_bs, (f16_t*)ds, ds_bs, C, DHW, acc_dx, acc_ds); else hipLaunchKernelGGL((gate_bwd_kernel<f16_t, false>), vox_grid<f16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (const f16_t*)s, s_bs, (const f16_t*)dy, dy_bs, (f16_t*)dx, dx_bs, (f16_t*)ds, ds_bs, C, DHW, acc_dx, acc_ds); }
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- AttenModule2 pairs
// AttenModule2 (buildingblocks.py:279-299) pools and gates TWO tensors -- the upsampled seg feature a and the encoder feature b:
// pooled = [ChannelPool(a) | ChannelPool(b)], out = [a (1 + E0) | b (1 + E1)].  As one launch per tensor that is eight launches
// per level (two pools, two gates, forward and backward), most of them latency at the 64^3 / 32^3 levels; here each of the four
// steps is ONE launch whose second grid dimension (lane-per-voxel kernels) or channel range (row kernel) selects the tensor.
template <typename T> struct Pair2 {
  const T* x[2]; long long x_bs[2]; int C[2];
  T* dx[2]; long long dx_bs[2]; int acc[2];
};
// y (N, 4, ...): channels 2 w, 2 w + 1 = (max_c, mean_c) of x[w]
template <typename T, bool VEC, int CBT = CB, int VWX = 0>
__global__ __launch_bounds__(256) void channel_pool2_fwd_kernel(const Pair2<T> p, T* __restrict__ y, long long y_bs, long long dhw) {
  const int w = blockIdx.y, C = p.C[w];
  VOX_LOOP_BEGIN_W(VWX)
    const T* xp = p.x[w] + n * p.x_bs[w];
    float m[VW], s[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) { m[v] = -INFINITY; s[v] = 0.f; }
    for (int c0 = 0; c0 < C; c0 += CBT) {
      float xv[CBT][VW];
#pragma unroll
      for (int j = 0; j < CBT; ++j) ldrow<VEC>(xp + (long long)min(c0 + j, C - 1) * dhw, q, valid, xv[j]);
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          m[v] = (xv[j][v] > m[v] || xv[j][v] != xv[j][v]) ? xv[j][v] : m[v];
          s[v] += xv[j][v];
        }
              }
      }
    }
#pragma unroll
    for (int v = 0; v < VW; ++v) s[v] = s[v] / (float)C;
    strow<VEC>(y + n * y_bs + (long long)(2 * w) * dhw, q, valid, m);
    strow<VEC>(y + n * y_bs + (long long)(2 * w + 1) * dhw, q, valid, s);
  VOX_LOOP_END
}
// dy (N, 4, ...) as above -> dx[w] (+)= the pooled gradients routed back (first maximum; mean to every channel)
template <typename T, bool VEC, int CBT = CB, int VWX = 0>
__global__ __launch_bounds__(256) void channel_pool2_bwd_kernel(const Pair2<T> p, const T* __restrict__ dy, long long dy_bs, long long dhw) {
  const int w = blockIdx.y, C = p.C[w], accumulate = p.acc[w];
  VOX_LOOP_BEGIN_W(VWX)
    const T* xp = p.x[w] + n * p.x_bs[w];
    float m[VW], g0[VW], g1[VW];
    int arg[VW];
    ldrow<VEC>(dy + n * dy_bs + (long long)(2 * w) * dhw, q, valid, g0);
    ldrow<VEC>(dy + n * dy_bs + (long long)(2 * w + 1) * dhw, q, valid, g1);
#pragma unroll
    for (int v = 0; v < VW; ++v) { m[v] = -INFINITY; arg[v] = 0; }
    for (int c0 = 0; c0 < C; c0 += CBT) {
      float xv[CBT][VW];
#pragma unroll
      for (int j = 0; j < CBT; ++j) ldrow<VEC>(xp + (long long)min(c0 + j, C - 1) * dhw, q, valid, xv[j]);
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
#pragma unroll
        for (int v = 0; v < VW; ++v)
          if (xv[j][v] > m[v] || xv[j][v] != xv[j][v]) { m[v] = xv[j][v]; arg[v] = c0 + j; }
              }
      }
    }
#pragma unroll
    for (int v = 0; v < VW; ++v) g1[v] = g1[v] / (float)C;
    T* dp = p.dx[w] + n * p.dx_bs[w];
    for (int c0 = 0; c0 < C; c0 += CBT) {
      float o[CBT][VW];
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        if (accumulate) {
          ldrow<VEC>((const T*)dp + (long long)min(c0 + j, C - 1) * dhw, q, valid, o[j]);
        } else {
#pragma unroll
          for (int v = 0; v < VW; ++v) o[j][v] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
#pragma unroll
        for (int v = 0; v < VW; ++v) o[j][v] = o[j][v] + g1[v] + (c0 + j == arg[v] ? g0[v] : 0.f);
        strow<VEC>(dp + (long long)(c0 + j) * dhw, q, valid, o[j]);
              }
      }
    }
  VOX_LOOP_END
}
// y (N, C0 + C1, ...) = [x0 (1 + E[:,0]) | x1 (1 + E[:,1])]; grid.y = C0 + C1
template <typename T, bool VEC>
// red != nullptr: also the channel sums (sum y, sum y^2) of the rounded output, for the InstanceNorm of the conv that follows
// (launched on the reduction grid then, like duse_gate_fwd_kernel)
__global__ __launch_bounds__(EW_BLOCK) void gate2_fwd_kernel(const Pair2<T> p, const T* E, long long E_bs, T* y, long long y_bs, long long dhw,
                                                            double* red) {
  __shared__ double s_red[4 * 2];
  const int w = (int)blockIdx.y >= p.C[0] ? 1 : 0, cl = blockIdx.y - (w ? p.C[0] : 0);
  double s[2] = {0.0, 0.0};
  ROW_LOOP_BEGIN
    float xv[VW], sv[VW];
    ldrow<VEC>(p.x[w] + n * p.x_bs[w] + (long long)cl * dhw, q, valid, xv);
    ldrow<VEC>(E + n * E_bs + (long long)w * dhw, q, valid, sv);
#pragma unroll
    for (int i = 0; i < VW; ++i) xv[i] *= (1.f + sv[i]);
    strow<VEC>(y + n * y_bs + (long long)c * dhw, q, valid, xv);
    if (red) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int i = 0; i < VW; ++i)
        if (i < valid) { const float r = rnd_as((const T*)nullptr, xv[i]); t0 += r; t1 = fmaf(r, r, t1); }
      s[0] += (double)t0;
      s[1] += (double)t1;
    }
  ROW_LOOP_END
  if (red) {
    block_sum_d<2>(s, s_red, EW_BLOCK >> 6);
    if (threadIdx.x < 2) atomicAdd(&red[((long long)blockIdx.z * gridDim.y + blockIdx.y) * 2 + threadIdx.x], s_red[threadIdx.x]);
  }
}
// dy (N, C0 + C1, ...) -> dx[w] (+)= dy (1 + E[:,w]),  dE[:,w] = sum_c dy x[w]
template <typename T, bool VEC, int CBT = CB>
__global__ __launch_bounds__(256) void gate2_bwd_kernel(const Pair2<T> p, const T* __restrict__ E, long long E_bs, const T* __restrict__ dy,
                                                       long long dy_bs, T* dE, long long dE_bs, long long dhw, int sig_bwd) {
  const int w = blockIdx.y, C = p.C[w], acc_dx = p.acc[w];
  const long long dyo = (long long)(w ? p.C[0] : 0) * dhw;
  VOX_LOOP_BEGIN
    float g1[VW], a[VW];
    ldrow<VEC>(E + n * E_bs + (long long)w * dhw, q, valid, g1);
#pragma unroll
    for (int v = 0; v < VW; ++v) { g1[v] = 1.f + g1[v]; a[v] = 0.f; }
    T* dxp = p.dx[w];
    for (int c0 = 0; c0 < C; c0 += CBT) {
      float g[CBT][VW], xv[CBT][VW], o[CBT][VW];
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        const long long off = (long long)min(c0 + j, C - 1) * dhw;
        ldrow<VEC>(dy + n * dy_bs + dyo + off, q, valid, g[j]);
        ldrow<VEC>(p.x[w] + n * p.x_bs[w] + off, q, valid, xv[j]);
        if (acc_dx) {
          ldrow<VEC>((const T*)dxp + n * p.dx_bs[w] + off, q, valid, o[j]);
        } else {
#pragma unroll
          for (int v = 0; v < VW; ++v) o[j][v] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          a[v] = fmaf(g[j][v], xv[j][v], a[v]);
          o[j][v] = g[j][v] * g1[v] + o[j][v];
        }
        strow<VEC>(dxp + n * p.dx_bs[w] + (long long)(c0 + j) * dhw, q, valid, o[j]);
              }
      }
    }
    if (sig_bwd) {                        // E = sigmoid(pre): store the gradient of the pre-activation, dE * E (1 - E)
#pragma unroll
      for (int v = 0; v < VW; ++v) { const float e = g1[v] - 1.f; a[v] *= e * (1.f - e); }
    }
    strow<VEC>(dE + n * dE_bs + (long long)w * dhw, q, valid, a);
  VOX_LOOP_END
}
template <typename T>
static Pair2<T> make_pair2(const void* xa, long long xa_bs, int Ca, const void* xb, long long xb_bs, int Cb, void* da, long long da_bs,
                           int acc_a, void* db, long long db_bs, int acc_b) {
  Pair2<T> p;
  p.x[0] = (const T*)xa; p.x[1] = (const T*)xb; p.x_bs[0] = xa_bs; p.x_bs[1] = xb_bs; p.C[0] = Ca; p.C[1] = Cb;
  p.dx[0] = (T*)da; p.dx[1] = (T*)db; p.dx_bs[0] = da_bs; p.dx_bs[1] = db_bs; p.acc[0] = acc_a; p.acc[1] = acc_b;
  return p;
}
#define PAIR_ARGS_OK (xa && xb && N > 0 && N <= 65535 && Ca > 0 && Cb > 0 && Ca <= 128 && Cb <= 128 && DHW > 0)
extern "C" int xh_channel_pool2_fwd(void* stream, int dtype, const void* xa, long long xa_bs, int Ca, const void* xb, long long xb_bs, int Cb,
                                    void* y, long long y_bs, int N, long long DHW) {
  if (!PAIR_ARGS_OK || !y) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, {
    const Pair2<T> p = make_pair2<T>(xa, xa_bs, Ca, xb, xb_bs, Cb, nullptr, 0, 0, nullptr, 0, 0);
    dim3 grid = vox_grid<T>(DHW, N); grid.y = 2;
    if (vec_ok<T>(DHW, {xa_bs, xb_bs, y_bs}) && deep_grid(grid, Ca < Cb ? Ca : Cb)) {
      grid = vox_grid_deep<T>(DHW, N); grid.y = 2;
      hipLaunchKernelGGL((channel_pool2_fwd_kernel<T, true, DEEP_CB, DEEP_VW>), grid, dim3(256), 0, st, p, (T*)y, y_bs, DHW);
    }
    else if (vec_ok<T>(DHW, {xa_bs, xb_bs, y_bs})) hipLaunchKernelGGL((channel_pool2_fwd_kernel<T, true>), grid, dim3(256), 0, st, p, (T*)y, y_bs, DHW);
    else hipLaunchKernelGGL((channel_pool2_fwd_kernel<T, false>), grid, dim3(256), 0, st, p, (T*)y, y_bs, DHW);
  });
  return xh_launch_status();
}
extern "C" int xh_channel_pool2_bwd(void* stream, int dtype, const void* xa, long long xa_bs, int Ca, const void* xb, long long xb_bs, int Cb,
                                    const void* dy, long long dy_bs, void* dxa, long long dxa_bs, int acc_a, void* dxb, long long dxb_bs,
                                    int acc_b, int N, long long DHW) {
  if (!PAIR_ARGS_OK || !dy || !dxa || !dxb) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, {
    const Pair2<T> p = make_pair2<T>(xa, xa_bs, Ca, xb, xb_bs, Cb, dxa, dxa_bs, acc_a, dxb, dxb_bs, acc_b);
    dim3 grid = vox_grid<T>(DHW, N); grid.y = 2;
    if (vec_ok<T>(DHW, {xa_bs, xb_bs, dy_bs, dxa_bs, dxb_bs}) && deep_grid(grid, Ca < Cb ? Ca : Cb)) {
      grid = vox_grid_deep<T>(DHW, N); grid.y = 2;
      hipLaunchKernelGGL((channel_pool2_bwd_kernel<T, true, DEEP_CB, DEEP_VW>), grid, dim3(256), 0, st, p, (const T*)dy, dy_bs, DHW);
    }
    else if (vec_ok<T>(DHW, {xa_bs, xb_bs, dy_bs, dxa_bs, dxb_bs}))
      hipLaunchKernelGGL((channel_pool2_bwd_kernel<T, true>), grid, dim3(256), 0, st, p, (const T*)dy, dy_bs, DHW);
    else hipLaunchKernelGGL((channel_pool2_bwd_kernel<T, false>), grid, dim3(256), 0, st, p, (const T*)dy, dy_bs, DHW);
  });
  return xh_launch_status();
}
extern "C" int xh_gate2_fwd(void* stream, int dtype, const void* xa, long long xa_bs, int Ca, const void* xb, long long xb_bs, int Cb,
                            const void* E, long long E_bs, void* y, long long y_bs, int N, long long DHW, double* red) {
  if (!PAIR_ARGS_OK || !E || !y) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, {
    const Pair2<T> p = make_pair2<T>(xa, xa_bs, Ca, xb, xb_bs, Cb, nullptr, 0, 0, nullptr, 0, 0);
    const dim3 grid = red ? red_grid<T>(DHW, Ca + Cb, N) : row_grid<T>(DHW, Ca + Cb, N);
    if (vec_ok<T>(DHW, {xa_bs, xb_bs, E_bs, y_bs}))
      hipLaunchKernelGGL((gate2_fwd_kernel<T, true>), grid, dim3(EW_BLOCK), 0, st, p, (const T*)E, E_bs, (T*)y, y_bs, DHW, red);
    else hipLaunchKernelGGL((gate2_fwd_kernel<T, false>), grid, dim3(EW_BLOCK), 0, st, p, (const T*)E, E_bs, (T*)y, y_bs, DHW, red);
  });
  return xh_launch_status();
}
extern "C" int xh_gate2_bwd(void* stream, int dtype, const void* xa, long long xa_bs, int Ca, const void* xb, long long xb_bs, int Cb,
                            const void* E, long long E_bs, const void* dy, long long dy_bs, void* dxa, long long dxa_bs, int acc_a, void* dxb,
                            long long dxb_bs, int acc_b, void* dE, long long dE_bs, int N, long long DHW, int sig_bwd) {
  if (!PAIR_ARGS_OK || !E || !dy || !dxa || !dxb || !dE) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, {
    const Pair2<T> p = make_pair2<T>(xa, xa_bs, Ca, xb, xb_bs, Cb, dxa, dxa_bs, acc_a, dxb, dxb_bs, acc_b);
    dim3 grid = vox_grid<T>(DHW, N); grid.y = 2;
    if (vec_ok<T>(DHW, {xa_bs, xb_bs, E_bs, dy_bs, dxa_bs, dxb_bs, dE_bs}) && deep_grid(grid, Ca < Cb ? Ca : Cb))
      hipLaunchKernelGGL((gate2_bwd_kernel<T, true, 8>), grid, dim3(256), 0, st, p, (const T*)E, E_bs, (const T*)dy, dy_bs, (T*)dE, dE_bs, DHW, sig_bwd);
    else if (vec_ok<T>(DHW, {xa_bs, xb_bs, E_bs, dy_bs, dxa_bs, dxb_bs, dE_bs}))
      hipLaunchKernelGGL((gate2_bwd_kernel<T, true>), grid, dim3(256), 0, st, p, (const T*)E, E_bs, (const T*)dy, dy_bs, (T*)dE, dE_bs, DHW, sig_bwd);
    else hipLaunchKernelGGL((gate2_bwd_kernel<T, false>), grid, dim3(256), 0, st, p, (const T*)E, E_bs, (const T*)dy, dy_bs, (T*)dE, dE_bs, DHW, sig_bwd);
  });
  return xh_launch_status();
}
#undef PAIR_ARGS_OK

// ---------------------------------------------------------------------------------------- gate + max-pool (+ moments)
// The skip-return attention gates every modality stream, x_i = a * x_i + x_i (RA_HVED.py:552), right before the next encoder
// pools it (buildingblocks.py:655-657) and normalises the pooled tensor (the first InstanceNorm of the DoubleConv).  As three
// launches that is a write + a read of the full-resolution gated tensor (16 channels x 128^3 at level 1) and a moments pass over
// the pooled one; here ONE pass reads x and the gate, keeps the gated values in registers (rounded to the storage type, so the
// pooled maxima -- and the arg-max the backward recomputes -- are those of the unfused path bit for bit), writes the pooled
// tensor and leaves its channel sums.  A lane owns 4 pooled voxels of a row = 8 input voxels x 2 rows x 2 planes.
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&o)[8]) {
  float a[4], b[4];
  ld4(p, 0, a); ld4(p, 4, b);
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&o)[8]) { ldvec(p, 0, o); }
template <> __device__ __forceinline__ void ld8<f16_t>(const f16_t* p, float (&o)[8]) { ldvec(p, 0, o); }
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&o)[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&o)[8]) {
  const float a[4] = {o[0], o[1], o[2], o[3]}, b[4] = {o[4], o[5], o[6], o[7]};
  st4(p, 0, a); st4(p, 4, b);
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&o)[8]) { stvec(p, 0, o); }
template <> __device__ __forceinline__ void st8<f16_t>(f16_t* p, const float (&o)[8]) { stvec(p, 0, o); }

// grid (blocks over runs, C, N); run = 4 pooled voxels along W
template <typename T>
__global__ __launch_bounds__(256) void gate_maxpool_fwd_kernel(const T* x, long long x_bs, const T* s, long long s_bs, T* y, long long y_bs,
                                                              int D, int H, int W, double* red, int Cg) {
  __shared__ double s_red[4 * 2];
  const int c = blockIdx.y, n = blockIdx.z, C = gridDim.y;
  if (c >= Cg) s = nullptr;                             // channels [Cg, C) are pooled ungated (the skip stream riding along)
  const int Do = D / 2, Ho = H / 2, Wo = W / 2, Wr = Wo / 4;
  const long long hw = (long long)H * W, dhw = (long long)D * hw, odhw = (long long)Do * Ho * Wo;
  const T* xp = x + n * x_bs + (long long)c * dhw;
  const T* sp = s ? s + n * s_bs : nullptr;            // no gate: a plain MaxPool3d(2) that leaves the channel sums of its output
  T* yp = y + n * y_bs + (long long)c * odhw;
  const long long runs = (long long)Do * Ho * Wr;
  double acc[2] = {0.0, 0.0};
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < runs; r += (long long)gridDim.x * 256) {
    const int wr = (int)(r % Wr);
    long long t = r / Wr;
    const int oh = (int)(t % Ho), od = (int)(t / Ho);
    const long long base = ((long long)(2 * od) * H + 2 * oh) * W + 8 * wr;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < 4; ++k) {                       // (dz, dy) rows of the window, in the scan order of max_pool3d
      const long long o = base + (long long)(k >> 1) * hw + (long long)(k & 1) * W;
      float xv[8], sv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      ld8<T>(xp + o, xv);
      if (sp) ld8<T>(sp + o, sv);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float v = rnd_as(yp, xv[2 * j + e] * (1.f + sv[2 * j + e]));
          m[j] = (v > m[j] || v != v) ? v : m[j];
        }
    }
    st4(yp, ((long long)od * Ho + oh) * Wo + 4 * wr, m);
    if (red) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { t0 += m[j]; t1 = fmaf(m[j], m[j], t1); }
      acc[0] += (double)t0; acc[1] += (double)t1;
    }
  }
  if (red) {
    block_sum_d<2>(acc, s_red, 4);
    if (threadIdx.x < 2) atomicAdd(&red[((long long)n * C + c) * 2 + threadIdx.x], s_red[threadIdx.x]);
  }
}
// dx = (arg-max of its window ? dy * (1 + s) : 0), ds = sum_c (arg-max ? dy * x : 0).  A workgroup owns 64 runs; lane = run,
// wave w takes the channels w, w + 4, ... (a lane-per-run loop over all 16 channels left 256 workgroups of serial work at level
// 1: one per CU); the four waves' partial ds rows meet in LDS and wave k writes window row k.
template <typename T>
__global__ __launch_bounds__(256) void gate_maxpool_bwd_kernel(const T* __restrict__ x, long long x_bs, const T* __restrict__ s, long long s_bs,
                                                              const T* __restrict__ dy, long long dy_bs, T* dx, long long dx_bs, T* ds,
                                                              long long ds_bs, int C, int D, int H, int W, int acc_dx, int Cg) {
  __shared__ float s_ds[4 * 4 * 8 * 64];                  // [wave][window row][element][lane]
  const int n = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int Do = D / 2, Ho = H / 2, Wo = W / 2, Wr = Wo / 4;
  const long long hw = (long long)H * W, dhw = (long long)D * hw, odhw = (long long)Do * Ho * Wo;
  const long long runs = (long long)Do * Ho * Wr;
  const long long r = (long long)blockIdx.x * 64 + lane;
  const bool live = r < runs;
  const long long rc = live ? r : 0;
  const int wr = (int)(rc % Wr);
  const long long t = rc / Wr;
  const int oh = (int)(t % Ho), od = (int)(t / Ho);
  const long long base = ((long long)(2 * od) * H + 2 * oh) * W + 8 * wr;
  const long long obase = ((long long)od * Ho + oh) * Wo + 4 * wr;
  float g1[4][8], dsv[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ld8<T>(s + n * s_bs + base + (long long)(k >> 1) * hw + (long long)(k & 1) * W, g1[k]);
#pragma unroll
    for (int e = 0; e < 8; ++e) { g1[k][e] = 1.f + g1[k][e]; dsv[k][e] = 0.f; }
  }
  for (int c = wv; c < C; c += 4) {
    const T* xp = x + n * x_bs + (long long)c * dhw + base;
    const bool gated = c < Cg;                          // channels [Cg, C): plain max-pool backward, no share in ds
    float xv[4][8], g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ld8<T>(xp + (long long)(k >> 1) * hw + (long long)(k & 1) * W, xv[k]);
    ld4(dy + n * dy_bs + (long long)c * odhw, obase, g);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0, 0, 0, 0};                          // window position 2 * k + e of the first maximum (scan order)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float v = rnd_as(x, xv[k][2 * j + e] * (gated ? g1[k][2 * j + e] : 1.f));
          if (v > m[j] || v != v) { m[j] = v; arg[j] = 2 * k + e; }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float gg = arg[j] == 2 * k + e ? g[j] : 0.f;
          o[2 * j + e] = gated ? gg * g1[k][2 * j + e] : gg;
          dsv[k][2 * j + e] = fmaf(gated ? gg : 0.f, xv[k][2 * j + e], dsv[k][2 * j + e]);
        }
      T* dp = dx + n * dx_bs + (long long)c * dhw + base + (long long)(k >> 1) * hw + (long long)(k & 1) * W;
      if (acc_dx && live) {                               // += : the gradient buffer already holds another consumer's share
        float prev[8];
        ld8<T>((const T*)dp, prev);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += prev[e];
      }
      if (live) st8<T>(dp, o);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) s_ds[((wv * 4 + k) * 8 + e) * 64 + lane] = dsv[k][e];
  __syncthreads();
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e)                             // the same order of additions whatever C is: waves 0, 1, 2, 3
    o[e] = ((s_ds[((0 * 4 + wv) * 8 + e) * 64 + lane] + s_ds[((1 * 4 + wv) * 8 + e) * 64 + lane]) +
            s_ds[((2 * 4 + wv) * 8 + e) * 64 + lane]) + s_ds[((3 * 4 + wv) * 8 + e) * 64 + lane];
  if (live) st8<T>(ds + n * ds_bs + base + (long long)(wv >> 1) * hw + (long long)(wv & 1) * W, o);
}
// The same for small volumes (a few dozen workgroups of the kernel above, each wave walking C / 4 channels one memory latency
// at a time: 64 channels @32^3 took 30 us on 16 workgroups): a workgroup takes 16 runs, the 64 lanes of a wave are 16 runs x 4
// channel slots, so 16 channels are in flight per workgroup step and there are four times as many workgroups.
template <typename T>
__global__ __launch_bounds__(256) void gate_maxpool_bwd_deep_kernel(const T* __restrict__ x, long long x_bs, const T* __restrict__ s, long long s_bs,
                                                                   const T* __restrict__ dy, long long dy_bs, T* dx, long long dx_bs, T* ds,
                                                                   long long ds_bs, int C, int D, int H, int W, int acc_dx, int Cg) {
  constexpr int RUNS = 16, SL = 64 / RUNS;
  __shared__ float s_ds[4 * 4 * 8 * 64];                  // [wave][window row][element][lane]
  const int n = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rl = lane & (RUNS - 1), slot = lane / RUNS;
  const int Do = D / 2, Ho = H / 2, Wo = W / 2, Wr = Wo / 4;
  const long long hw = (long long)H * W, dhw = (long long)D * hw, odhw = (long long)Do * Ho * Wo;
  const long long runs = (long long)Do * Ho * Wr;
  const long long r = (long long)blockIdx.x * RUNS + rl;
  const bool live = r < runs;
  const long long rc = live ? r : 0;
  const int wr = (int)(rc % Wr);
  const long long t = rc / Wr;
  const int oh = (int)(t % Ho), od = (int)(t / Ho);
  const long long base = ((long long)(2 * od) * H + 2 * oh) * W + 8 * wr;
  const long long obase = ((long long)od * Ho + oh) * Wo + 4 * wr;
  float g1[4][8], dsv[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ld8<T>(s + n * s_bs + base + (long long)(k >> 1) * hw + (long long)(k & 1) * W, g1[k]);
#pragma unroll
    for (int e = 0; e < 8; ++e) { g1[k][e] = 1.f + g1[k][e]; dsv[k][e] = 0.f; }
  }
  for (int c = wv + 4 * slot; c < C; c += 4 * SL) {
    const T* xp = x + n * x_bs + (long long)c * dhw + base;
    const bool gated = c < Cg;                          // channels [Cg, C): plain max-pool backward, no share in ds
    float xv[4][8], g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ld8<T>(xp + (long long)(k >> 1) * hw + (long long)(k & 1) * W, xv[k]);
    ld4(dy + n * dy_bs + (long long)c * odhw, obase, g);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0, 0, 0, 0};                          // window position 2 * k + e of the first maximum (scan order)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float v = rnd_as(x, xv[k][2 * j + e] * (gated ? g1[k][2 * j + e] : 1.f));
          if (v > m[j] || v != v) { m[j] = v; arg[j] = 2 * k + e; }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float gg = arg[j] == 2 * k + e ? g[j] : 0.f;
          o[2 * j + e] = gated ? gg * g1[k][2 * j + e] : gg;
          dsv[k][2 * j + e] = fmaf(gated ? gg : 0.f, xv[k][2 * j + e], dsv[k][2 * j + e]);
        }
      T* dp = dx + n * dx_bs + (long long)c * dhw + base + (long long)(k >> 1) * hw + (long long)(k & 1) * W;
      if (acc_dx && live) {
        float prev[8];
        ld8<T>((const T*)dp, prev);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += prev[e];
      }
      if (live) st8<T>(dp, o);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) s_ds[((wv * 4 + k) * 8 + e) * 64 + lane] = dsv[k][e];
  __syncthreads();
  // thread (k, run) of the first 4 * RUNS: the sum over waves and channel slots, always in the same order
  if (threadIdx.x < 4 * RUNS) {
    const int k = threadIdx.x / RUNS, rr = threadIdx.x % RUNS;
    const long long r2 = (long long)blockIdx.x * RUNS + rr;
    if (r2 < runs) {
      const int wr2 = (int)(r2 % Wr);
      const long long t2 = r2 / Wr;
      const int oh2 = (int)(t2 % Ho), od2 = (int)(t2 / Ho);
      const long long base2 = ((long long)(2 * od2) * H + 2 * oh2) * W + 8 * wr2;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float acc = 0.f;
        for (int w = 0; w < 4; ++w)
          for (int sl = 0; sl < SL; ++sl) acc += s_ds[((w * 4 + k) * 8 + e) * 64 + sl * RUNS + rr];
        o[e] = acc;
      }
      st8<T>(ds + n * ds_bs + base2 + (long long)(k >> 1) * hw + (long long)(k & 1) * W, o);
    }
  }
}
static bool gmp_ok(int D, int H, int W, std::initializer_list<long long> strides) {
  if (D < 2 || H < 2 || W < 8 || (D & 1) || (H & 1) || (W & 7)) return false;
  for (long long v : strides)
    if (v & 7) return false;
  return true;
}
extern "C" int xh_gate_maxpool_fwd(void* stream, int dtype, const void* x, long long x_bs, const void* s, long long s_bs, void* y, long long y_bs,
                                   int N, int C, int D, int H, int W, double* red, int Cg) {
  if (!x || !y || N <= 0 || C <= 0 || C > 65535 || N > 65535 || Cg < 0) return XH_ERR_ARG;
  if (Cg == 0 || Cg > C) Cg = C;
  if (!gmp_ok(D, H, W, {x_bs, s ? s_bs : 0, y_bs})) return XH_ERR_ARG;
  const long long runs = (long long)(D / 2) * (H / 2) * (W / 8);
  long long nb = (runs + 255) / 256;
  if (nb > 64) nb = 64;                                   // at most 64 adders per statistics address
  dim3 grid((unsigned)nb, C, N);
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL(gate_maxpool_fwd_kernel<T>, grid, dim3(256), 0, st, (const T*)x, x_bs, (const T*)s, s_bs, (T*)y, y_bs,
                                          D, H, W, red, Cg););
  return xh_launch_status();
}
extern "C" int xh_gate_maxpool_bwd(void* stream, int dtype, const void* x, long long x_bs, const void* s, long long s_bs, const void* dy,
                                   long long dy_bs, void* dx, long long dx_bs, void* ds, long long ds_bs, int N, int C, int D, int H, int W,
                                   int acc_dx, int Cg) {
  if (!x || !s || !dy || !dx || !ds || N <= 0 || C <= 0 || N > 65535 || Cg < 0) return XH_ERR_ARG;
  if (Cg == 0 || Cg > C) Cg = C;
  if (!gmp_ok(D, H, W, {x_bs, s_bs, dx_bs, ds_bs}) || (dy_bs & 3)) return XH_ERR_ARG;
  const long long runs = (long long)(D / 2) * (H / 2) * (W / 8);
  const long long nb = (runs + 63) / 64;
  if (nb >= (1ll << 31)) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (nb * N < 256 && C >= 16) {                          // the deep levels: 16 runs x 16 channel slots per workgroup
    dim3 grid((unsigned)((runs + 15) / 16), N);
    XH_DISPATCH_T(dtype, hipLaunchKernelGGL(gate_maxpool_bwd_deep_kernel<T>, grid, dim3(256), 0, st, (const T*)x, x_bs, (const T*)s, s_bs,
                                            (const T*)dy, dy_bs, (T*)dx, dx_bs, (T*)ds, ds_bs, C, D, H, W, acc_dx, Cg););
    return xh_launch_status();
  }
  dim3 grid((unsigned)nb, N);
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL(gate_maxpool_bwd_kernel<T>, grid, dim3(256), 0, st, (const T*)x, x_bs, (const T*)s, s_bs, (const T*)dy,
                                          dy_bs, (T*)dx, dx_bs, (T*)ds, ds_bs, C, D, H, W, acc_dx, Cg););
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- DuSE gates
// red != nullptr: the channel sums of the (rounded) output for the BatchNorm that follows (red[n][c][0..1] += sum u, sum u^2),
// as xh_moments would find them in u -- launched on the reduction grid then (few, long workgroups per atomic address)
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void duse_gate_fwd_kernel(const T* x, long long x_bs, const float* ch, const T* sp,
                                                                long long sp_bs, T* u, long long u_bs, int C, long long dhw, double* red) {
  __shared__ double s_red[4 * 2];
  const float cg = 1.f + ch[blockIdx.z * C + blockIdx.y];
  double s[2] = {0.0, 0.0};
  ROW_LOOP_BEGIN
    float xv[VW], sv[VW];
    ldrow<VEC>(x + n * x_bs + (long long)c * dhw, q, valid, xv);
    ldrow<VEC>(sp + n * sp_bs, q, valid, sv);
#pragma unroll
    for (int i = 0; i < VW; ++i) xv[i] *= (cg + sv[i]);
    strow<VEC>(u + n * u_bs + (long long)c * dhw, q, valid, xv);
    if (red) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int i = 0; i < VW; ++i)
        if (i < valid) { const float r = rnd_as((const T*)nullptr, xv[i]); t0 += r; t1 = fmaf(r, r, t1); }
      s[0] += (double)t0;
      s[1] += (double)t1;
    }
  ROW_LOOP_END
  if (red) {
    block_sum_d<2>(s, s_red, EW_BLOCK >> 6);
    if (threadIdx.x < 2) atomicAdd(&red[((long long)blockIdx.z * C + blockIdx.y) * 2 + threadIdx.x], s_red[threadIdx.x]);
  }
}
// lane per voxel over channels; dch via block reduction per channel would need C reductions: instead each
// block owns one (n, c) row for dch and the dsp accumulation goes through a second voxel-major kernel.
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void duse_gate_bwd_row_kernel(const T* x, long long x_bs, const float* ch, const T* sp,
                                                                    long long sp_bs, const T* du, long long du_bs, T* dx,
                                                                    long long dx_bs, double* dch, int C, long long dhw) {
  __shared__ double s_red[4];
  const float cg = 1.f + ch[blockIdx.z * C + blockIdx.y];
  double s[1] = {0.0};
  ROW_LOOP_BEGIN
    float xv[VW], sv[VW], g[VW], o[VW];
    ldrow<VEC>(x + n * x_bs + (long long)c * dhw, q, valid, xv);
    ldrow<VEC>(sp + n * sp_bs, q, valid, sv);
    ldrow<VEC>(du + n * du_bs + (long long)c * dhw, q, valid, g);
    float t0 = 0.f;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      o[i] = g[i] * (cg + sv[i]);
      t0 = fmaf(g[i], xv[i], t0);
    }
    s[0] += (double)t0;
    strow<VEC>(dx + n * dx_bs + (long long)c * dhw, q, valid, o);
  ROW_LOOP_END
  block_sum_d<1>(s, s_red, EW_BLOCK >> 6);
  if (threadIdx.x == 0) atomicAdd(&dch[blockIdx.z * C + blockIdx.y], s_red[0]);
}
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void duse_gate_bwd_sp_kernel(const T* __restrict__ x, long long x_bs, const T* __restrict__ du,
                                                              long long du_bs, T* __restrict__ dsp, long long dsp_bs, int C,
                                                              long long dhw) {
  VOX_LOOP_BEGIN
    float a[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) a[v] = 0.f;
    for (int c0 = 0; c0 < C; c0 += CB) {
      float g[CB][VW], xv[CB][VW];
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const long long off = (long long)min(c0 + j, C - 1) * dhw;
        ldrow<VEC>(du + n * du_bs + off, q, valid, g[j]);
        ldrow<VEC>(x + n * x_bs + off, q, valid, xv[j]);
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
#pragma unroll
        for (int v = 0; v < VW; ++v) a[v] = fmaf(g[j][v], xv[j][v], a[v]);
              }
      }
    }
    strow<VEC>(dsp + n * dsp_bs, q, valid, a);
  VOX_LOOP_END
}
// Both halves in ONE voxel-major pass for C = 4 / 8 / 16 (the decoder levels of the network): a lane walks the channels of its
// run once -- dx_c = du_c (1 + ch_c + sp), the voxel sum dsp = sum_c du_c x_c, the channel sums dch_c in registers until one block
// reduction -- instead of reading x and du in a row-major and again in a voxel-major launch.  SIG: dsp is stored times
// sp (1 - sp), i.e. already through the sigmoid that produced sp (the act_bwd pass behind it).
template <typename T, bool VEC, int C, bool SIG>
__global__ __launch_bounds__(256) void duse_gate_bwd_fused_kernel(const T* __restrict__ x, long long x_bs, const float* __restrict__ ch,
                                                                 const T* __restrict__ sp, long long sp_bs, const T* __restrict__ du,
                                                                 long long du_bs, T* __restrict__ dx, long long dx_bs,
                                                                 T* __restrict__ dsp, long long dsp_bs, double* dch, long long dhw) {
  __shared__ double s_red[4 * C];
  float cg[C], part[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { cg[c] = 1.f + ch[blockIdx.z * C + c]; part[c] = 0.f; }
  double tot[C];
#pragma unroll
  for (int c = 0; c < C; ++c) tot[c] = 0.0;
  VOX_LOOP_BEGIN
    float sv[VW], a[VW];
    ldrow<VEC>(sp + n * sp_bs, q, valid, sv);
#pragma unroll
    for (int v = 0; v < VW; ++v) a[v] = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += 4) {
      float g[4][VW], xv[4][VW];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ldrow<VEC>(du + n * du_bs + (long long)(c0 + j) * dhw, q, valid, g[j]);
        ldrow<VEC>(x + n * x_bs + (long long)(c0 + j) * dhw, q, valid, xv[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float o[VW], t = 0.f;
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          const float gv = v < valid ? g[j][v] : 0.f;
          o[v] = gv * (cg[c0 + j] + sv[v]);
          const float pr = gv * xv[j][v];
          a[v] += pr;
          t += pr;
        }
        part[c0 + j] += t;
        strow<VEC>(dx + n * dx_bs + (long long)(c0 + j) * dhw, q, valid, o);
      }
    }
    if (SIG) {
#pragma unroll
      for (int v = 0; v < VW; ++v) a[v] *= sv[v] * (1.f - sv[v]);
    }
    strow<VEC>(dsp + n * dsp_bs, q, valid, a);
    // fp32 partial over one run, fp64 across the lane's runs (as the row kernel: fp64 from the run level on)
#pragma unroll
    for (int c = 0; c < C; ++c) { tot[c] += (double)part[c]; part[c] = 0.f; }
  VOX_LOOP_END
  block_sum_d<C>(tot, s_red, 4);
  if (threadIdx.x < C) atomicAdd(&dch[blockIdx.z * C + threadIdx.x], s_red[threadIdx.x]);
}
template <typename T>
static bool duse_gate_bwd_fused_try(void* stream, const void* x, long long x_bs, const float* ch, const void* sp, long long sp_bs,
                                    const void* du, long long du_bs, void* dx, long long dx_bs, void* dsp, long long dsp_bs, double* dch,
                                    int N, int C, long long DHW, int sig) {
  if ((C != 4 && C != 8 && C != 16) || (g_xh_disable & 1024)) return false;
  const bool vec = vec_ok<T>(DHW, {x_bs, sp_bs, du_bs, dx_bs, dsp_bs});
  const dim3 grid = vox_grid<T>(DHW, N, 512);           // every workgroup ends in C same-address fp64 atomics
#define DGF(CC, V, S) hipLaunchKernelGGL((duse_gate_bwd_fused_kernel<T, V, CC, S>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, x_bs, ch, (const T*)sp, sp_bs, (const T*)du, du_bs, (T*)dx, dx_bs, (T*)dsp, dsp_bs, dch, DHW)
#define DGS(CC, V) do { if (sig) DGF(CC, V, true); else DGF(CC, V, false); } while (0)
#define DGV(CC) do { if (vec) DGS(CC, true); else DGS(CC, false); } while (0)
  if (C == 4) DGV(4); else if (C == 8) DGV(8); else DGV(16);
#undef DGV
#undef DGS
#undef DGF
  return true;
}
static int launch_duse_gate_fwd(void* stream, int dtype, const void* x, long long x_bs, const float* ch, const void* sp, long long sp_bs,
                                void* u, long long u_bs, int N, int C, long long DHW, double* red) {
  if (!x || !ch || !sp || !u || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  const bool vec32 = vec_ok<float>(DHW, {x_bs, sp_bs, u_bs}), vec16 = vec_ok<bf16_t>(DHW, {x_bs, sp_bs, u_bs});
  const dim3 grid32 = red ? red_grid<float>(DHW, C, N) : row_grid<float>(DHW, C, N);
  const dim3 grid16 = red ? red_grid<bf16_t>(DHW, C, N) : row_grid<bf16_t>(DHW, C, N);
#define DG(T, V, G) hipLaunchKernelGGL((duse_gate_fwd_kernel<T, V>), G, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)x, x_bs, ch, (const T*)sp, sp_bs, (T*)u, u_bs, C, DHW, red)
  if (dtype == XH_F32) { if (vec32) DG(float, true, grid32); else DG(float, false, grid32); }
  else if (dtype == XH_BF16) { if (vec16) DG(bf16_t, true, grid16); else DG(bf16_t, false, grid16); }
  else if (dtype == XH_F16) { if (vec16) DG(f16_t, true, grid16); else DG(f16_t, false, grid16); }
  else return XH_ERR_DTYPE;
#undef DG
  return xh_launch_status();
}
extern "C" int xh_duse_gate_bwd_fuses(int C) { return (C == 4 || C == 8 || C == 16) && !(g_xh_disable & 1024); }
extern "C" int xh_duse_gate_fwd(void* stream, int dtype, const void* x, long long x_bs, const float* ch, const void* sp,
                                long long sp_bs, void* u, long long u_bs, int N, int C, long long DHW) {
  return launch_duse_gate_fwd(stream, dtype, x, x_bs, ch, sp, sp_bs, u, u_bs, N, C, DHW, nullptr);
}
extern "C" int xh_duse_gate_fwd_stats(void* stream, int dtype, const void* x, long long x_bs, const float* ch, const void* sp,
                                      long long sp_bs, void* u, long long u_bs, int N, int C, long long DHW, double* red) {
  if (!red) return XH_ERR_ARG;
  return launch_duse_gate_fwd(stream, dtype, x, x_bs, ch, sp, sp_bs, u, u_bs, N, C, DHW, red);
}
extern "C" int xh_duse_gate_bwd(void* stream, int dtype, const void* x, long long x_bs, const float* ch, const void* sp,
                                long long sp_bs, const void* du, long long du_bs, void* dx, long long dx_bs, void* dsp,
                                long long dsp_bs, double* dch, int N, int C, long long DHW, int sigmoid_bwd) {
  if (!x || !ch || !sp || !du || !dx || !dsp || !dch || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  {
    bool done = false;
    XH_DISPATCH_T(dtype, done = duse_gate_bwd_fused_try<T>(stream, x, x_bs, ch, sp, sp_bs, du, du_bs, dx, dx_bs, dsp, dsp_bs, dch, N, C, DHW, sigmoid_bwd););
    if (done) return xh_launch_status();
  }
  if (sigmoid_bwd) return XH_ERR_ARG;                  // only the fused pass applies it (C = 4 / 8 / 16): the caller asks xh_duse_gate_bwd_fuses()
  const bool vec32 = vec_ok<float>(DHW, {x_bs, sp_bs, du_bs, dx_bs}), vec16 = vec_ok<bf16_t>(DHW, {x_bs, sp_bs, du_bs, dx_bs});
  const dim3 grid32 = red_grid<float>(DHW, C, N), grid16 = red_grid<bf16_t>(DHW, C, N);
  if (dtype == XH_F32) {
    { if (vec32) hipLaunchKernelGGL((duse_gate_bwd_row_kernel<float, true>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)x, x_bs, ch, (const float*)sp, sp_bs, (const float*)du, du_bs, (float*)dx, dx_bs, dch, C, DHW); else hipLaunchKernelGGL((duse_gate_bwd_row_kernel<float, false>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)x, x_bs, ch, (const float*)sp, sp_bs, (const float*)du, du_bs, (float*)dx, dx_bs, dch, C, DHW); }
    { if (vec_ok<float>(DHW, {x_bs, du_bs, dsp_bs})) hipLaunchKernelGGL((duse_gate_bwd_sp_kernel<float, true>), vox_grid<float>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_bs, (const float*)du, du_bs, (float*)dsp, dsp_bs, C, DHW); else hipLaunchKernelGGL((duse_gate_bwd_sp_kernel<float, false>), vox_grid<float>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_bs, (const float*)du, du_bs, (float*)dsp, dsp_bs, C, DHW); }
  } else if (dtype == XH_BF16) {
    { if (vec16) hipLaunchKernelGGL((duse_gate_bwd_row_kernel<bf16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, ch, (const bf16_t*)sp, sp_bs, (const bf16_t*)du, du_bs, (bf16_t*)dx, dx_bs, dch, C, DHW); else hipLaunchKernelGGL((duse_gate_bwd_row_kernel<bf16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, ch, (const bf16_t*)sp, sp_bs, (const bf16_t*)du, du_bs, (bf16_t*)dx, dx_bs, dch, C, DHW); }
    { if (vec_ok<bf16_t>(DHW, {x_bs, du_bs, dsp_bs})) hipLaunchKernelGGL((duse_gate_bwd_sp_kernel<bf16_t, true>), vox_grid<bf16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (const bf16_t*)du, du_bs, (bf16_t*)dsp, dsp_bs, C, DHW); else hipLaunchKernelGGL((duse_gate_bwd_sp_kernel<bf16_t, false>), vox_grid<bf16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_bs, (const bf16_t*)du, du_bs, (bf16_t*)dsp, dsp_bs, C, DHW); }
  } else if (dtype == XH_F16) {
    { if (vec16) hipLaunchKernelGGL((duse_gate_bwd_row_kernel<f16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, ch, (const f16_t*)sp, sp_bs, (const f16_t*)du, du_bs, (f16_t*)dx, dx_bs, dch, C, DHW); else hipLaunchKernelGGL((duse_gate_bwd_row_kernel<f16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, ch, (const f16_t*)sp, sp_bs, (const f16_t*)du, du_bs, (f16_t*)dx, dx_bs, dch, C, DHW); }
    { if (vec_ok<f16_t>(DHW, {x_bs, du_bs, dsp_bs})) hipLaunchKernelGGL((duse_gate_bwd_sp_kernel<f16_t, true>), vox_grid<f16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (const f16_t*)du, du_bs, (f16_t*)dsp, dsp_bs, C, DHW); else hipLaunchKernelGGL((duse_gate_bwd_sp_kernel<f16_t, false>), vox_grid<f16_t>(DHW, N), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, x_bs, (const f16_t*)du, du_bs, (f16_t*)dsp, dsp_bs, C, DHW); }
  } else {
    return XH_ERR_DTYPE;
  }
  return xh_launch_status();
}

// dx (+)= w[c]*d[n,0,p] + k[n,c]   -- finishes the DuSE input gradient: squeeze-conv data gradient (rank-1) plus
// the global-average-pool gradient.
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void rank1_add_kernel(T* dx, long long dx_bs, const T* d, long long d_bs,
                                                            const float* w, const float* k, int C, long long dhw) {
  const float wc = w[blockIdx.y], kc = k ? k[blockIdx.z * C + blockIdx.y] : 0.f;
  ROW_LOOP_BEGIN
    float o[VW], dv[VW];
    T* dp = dx + n * dx_bs + (long long)c * dhw;
    ldrow<VEC>((const T*)dp, q, valid, o);
    ldrow<VEC>(d + n * d_bs, q, valid, dv);
#pragma unroll
    for (int i = 0; i < VW; ++i) o[i] += fmaf(wc, dv[i], kc);
    strow<VEC>(dp, q, valid, o);
  ROW_LOOP_END
}
extern "C" int xh_rank1_add(void* stream, int dtype, void* dx, long long dx_bs, const void* d, long long d_bs,
                            const float* w, const float* k, int N, int C, long long DHW) {
  if (!dx || !d || !w || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  const bool vec32 = vec_ok<float>(DHW, {dx_bs, d_bs}), vec16 = vec_ok<bf16_t>(DHW, {dx_bs, d_bs});
  const dim3 grid32 = row_grid<float>(DHW, C, N), grid16 = row_grid<bf16_t>(DHW, C, N);
  if (dtype == XH_F32)
    { if (vec32) hipLaunchKernelGGL((rank1_add_kernel<float, true>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (float*)dx, dx_bs, (const float*)d, d_bs, w, k, C, DHW); else hipLaunchKernelGGL((rank1_add_kernel<float, false>), grid32, dim3(EW_BLOCK), 0, (hipStream_t)stream, (float*)dx, dx_bs, (const float*)d, d_bs, w, k, C, DHW); }
  else if (dtype == XH_BF16)
    { if (vec16) hipLaunchKernelGGL((rank1_add_kernel<bf16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (bf16_t*)dx, dx_bs, (const bf16_t*)d, d_bs, w, k, C, DHW); else hipLaunchKernelGGL((rank1_add_kernel<bf16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (bf16_t*)dx, dx_bs, (const bf16_t*)d, d_bs, w, k, C, DHW); }
  else if (dtype == XH_F16)
    { if (vec16) hipLaunchKernelGGL((rank1_add_kernel<f16_t, true>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (f16_t*)dx, dx_bs, (const f16_t*)d, d_bs, w, k, C, DHW); else hipLaunchKernelGGL((rank1_add_kernel<f16_t, false>), grid16, dim3(EW_BLOCK), 0, (hipStream_t)stream, (f16_t*)dx, dx_bs, (const f16_t*)d, d_bs, w, k, C, DHW); }
  else
    return XH_ERR_DTYPE;
  return xh_launch_status();
}

// tiny dense layers of the DuSE channel excitation (single block)
__global__ void duse_fc_fwd_kernel(const double* red_r, const double* red_s, long long count, int N, int C,
                                   const float* wc, const float* bc, const float* w1, const float* b1, const float* w2,
                                   const float* b2, float* g, float* ch1, float* ch2, float* means) {
  extern __shared__ float sm[];   // [N][2C] means, [N][C] g
  float* mean = sm;
  float* gs = sm + N * 2 * C;
  for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
    const int n = i / C, c = i % C;
    mean[n * 2 * C + c] = (float)(red_r[i * 2] / (double)count);
    mean[n * 2 * C + C + c] = (float)(red_s[i * 2] / (double)count);
    if (means) { means[n * 2 * C + c] = mean[n * 2 * C + c]; means[n * 2 * C + C + c] = mean[n * 2 * C + C + c]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
    const int n = i / C, c = i % C;
    float a = bc[c];
    for (int k = 0; k < 2 * C; ++k) a = fmaf(wc[c * 2 * C + k], mean[n * 2 * C + k], a);
    gs[i] = a;
    g[i] = a;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
    const int n = i / C, c = i % C;
    float a1 = b1[c], a2 = b2[c];
    for (int k = 0; k < C; ++k) { a1 = fmaf(w1[c * C + k], gs[n * C + k], a1); a2 = fmaf(w2[c * C + k], gs[n * C + k], a2); }
    ch1[i] = sigmoidf_(a1);
    ch2[i] = sigmoidf_(a2);
  }
}
__global__ void duse_fc_bwd_kernel(const double* red_r, const double* red_s, long long count, int N, int C,
                                   const float* wc, const float* w1, const float* w2, const float* g, const float* ch1,
                                   const float* ch2, const double* dch1, const double* dch2, float* dwc, float* dbc,
                                   float* dw1, float* db1, float* dw2, float* db2, float* dmean_r, float* dmean_s, const float* means) {
  extern __shared__ float sm[];   // mean [N][2C], p1 [N][C], p2 [N][C], dg [N][C]
  float* mean = sm;
  float* p1 = sm + N * 2 * C;
  float* p2 = p1 + N * C;
  float* dg = p2 + N * C;
  for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
    const int n = i / C, c = i % C;
    mean[n * 2 * C + c] = means ? means[n * 2 * C + c] : (float)(red_r[i * 2] / (double)count);
    mean[n * 2 * C + C + c] = means ? means[n * 2 * C + C + c] : (float)(red_s[i * 2] / (double)count);
    p1[i] = (float)dch1[i] * ch1[i] * (1.f - ch1[i]);
    p2[i] = (float)dch2[i] * ch2[i] * (1.f - ch2[i]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
    const int n = i / C, k = i % C;
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w1[c * C + k] * p1[n * C + c] + w2[c * C + k] * p2[n * C + c];
    dg[i] = a;
  }
  __syncthreads();
  // parameter gradients (sum over n)
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
    const int c = i / C, k = i % C;
    float a1 = 0.f, a2 = 0.f;
    for (int n = 0; n < N; ++n) { a1 += p1[n * C + c] * g[n * C + k]; a2 += p2[n * C + c] * g[n * C + k]; }
    dw1[i] += a1;
    dw2[i] += a2;
  }
  for (int i = threadIdx.x; i < C * 2 * C; i += blockDim.x) {
    const int c = i / (2 * C), k = i % (2 * C);
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += dg[n * C + c] * mean[n * 2 * C + k];
    dwc[i] += a;
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int n = 0; n < N; ++n) { a1 += p1[n * C + c]; a2 += p2[n * C + c]; a3 += dg[n * C + c]; }
    db1[c] += a1; db2[c] += a2; dbc[c] += a3;
  }
  for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
    const int n = i / C, k = i % C;
    float ar = 0.f, as = 0.f;
    for (int c = 0; c < C; ++c) { ar += wc[c * 2 * C + k] * dg[n * C + c]; as += wc[c * 2 * C + C + k] * dg[n * C + c]; }
    dmean_r[i] = ar / (float)count;     // already divided: d(mean)/dx = 1/count
    dmean_s[i] = as / (float)count;
  }
}
extern "C" int xh_duse_fc_fwd(void* stream, const double* red_r, const double* red_s, long long count, int N, int C,
                              const float* w_comb, const float* b_comb, const float* w1, const float* b1, const float* w2,
                              const float* b2, float* g, float* ch1, float* ch2, float* means) {
  if (!red_r || !red_s || !w_comb || !b_comb || !w1 || !b1 || !w2 || !b2 || !g || !ch1 || !ch2 || N <= 0 || C <= 0 || count <= 0) return XH_ERR_ARG;
  const size_t shm = (size_t)N * 3 * C * sizeof(float);
  if (shm > 60000) return XH_ERR_ARG;
  hipLaunchKernelGGL(duse_fc_fwd_kernel, dim3(1), dim3(256), shm, (hipStream_t)stream, red_r, red_s, count, N, C, w_comb, b_comb, w1, b1, w2, b2, g, ch1, ch2, means);
  return xh_launch_status();
}
extern "C" int xh_duse_fc_bwd(void* stream, const double* red_r, const double* red_s, long long count, int N, int C,
                              const float* w_comb, const float* w1, const float* w2, const float* g, const float* ch1,
                              const float* ch2, const double* dch1, const double* dch2, float* dw_comb, float* db_comb,
                              float* dw1, float* db1, float* dw2, float* db2, float* dmean_r, float* dmean_s, const float* means) {
  if ((!means && (!red_r || !red_s)) || !w_comb || !w1 || !w2 || !g || !ch1 || !ch2 || !dch1 || !dch2 || !dw_comb || !db_comb || !dw1 ||
      !db1 || !dw2 || !db2 || !dmean_r || !dmean_s || N <= 0 || C <= 0 || count <= 0)
    return XH_ERR_ARG;
  const size_t shm = (size_t)N * 5 * C * sizeof(float);
  if (shm > 60000) return XH_ERR_ARG;
  hipLaunchKernelGGL(duse_fc_bwd_kernel, dim3(1), dim3(256), shm, (hipStream_t)stream, red_r, red_s, count, N, C, w_comb, w1, w2, g, ch1, ch2, dch1, dch2, dw_comb, db_comb, dw1, db1, dw2, db2, dmean_r, dmean_s, means);
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- DuSE pair: the tiny dense layers inside the passes
// The channel excitation of DuSEAttention (modules/DuSFE.py:113-133: two global average pools -> fc_comb -> fc_ch1 / fc_ch2 ->
// sigmoid) is 2C^2 + 2C^2 multiply-adds on 2C numbers; as launches of their own (one workgroup forward, one backward) they are
// ~5 + ~9 us of pure launch latency per decoder level.  For the recon | seg PAIR (one sample: x viewed as (2, C, ...)) every
// workgroup of the gate pass derives its own channel's gate from the raw channel sums, and every workgroup of the pass that
// finishes the input gradient derives its channel's pooled-mean gradient; the first workgroup also leaves what the backward
// pass / the optimizer needs (gate vector, pooled means; parameter gradients).
struct DuseFc {
  const double* red;                 // [2C][2] raw sums of the pair (sum x, sum x^2): recon channels first
  double inv_count;
  const float *wc, *bc, *w1, *b1, *w2, *b2;
  float *g_out, *ch_out, *means_out; // [C], [2][C], [2C]
};
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void duse_gate_fc_fwd_kernel(const T* x, long long x_bs, const T* sp, long long sp_bs, T* u, long long u_bs,
                                                                   int C, long long dhw, double* red, const DuseFc f) {
  __shared__ double s_red[4 * 2];
  __shared__ float s_mean[64], s_g[32], s_gate;
  const int half = blockIdx.z, cc = blockIdx.y, tid = threadIdx.x;
  for (int k = tid; k < 2 * C; k += EW_BLOCK) s_mean[k] = (float)(f.red[2 * k] * f.inv_count);
  __syncthreads();
  for (int j = tid; j < C; j += EW_BLOCK) {
    float a = f.bc[j];
    for (int k = 0; k < 2 * C; ++k) a = fmaf(f.wc[j * 2 * C + k], s_mean[k], a);
    s_g[j] = a;
  }
  __syncthreads();
  if (tid == 0) {
    const float* w = half ? f.w2 : f.w1;
    float a = (half ? f.b2 : f.b1)[cc];
    for (int k = 0; k < C; ++k) a = fmaf(w[cc * C + k], s_g[k], a);
    const float ch = sigmoidf_(a);
    s_gate = 1.f + ch;
    if (blockIdx.x == 0) {
      f.ch_out[half * C + cc] = ch;
      if (half == 0 && cc == 0) {
        for (int k = 0; k < 2 * C; ++k) f.means_out[k] = s_mean[k];
        for (int k = 0; k < C; ++k) f.g_out[k] = s_g[k];
      }
    }
  }
  __syncthreads();
  const float cg = s_gate;
  double s[2] = {0.0, 0.0};
  ROW_LOOP_BEGIN
    float xv[VW], sv[VW];
    ldrow<VEC>(x + n * x_bs + (long long)c * dhw, q, valid, xv);
    ldrow<VEC>(sp + n * sp_bs, q, valid, sv);
#pragma unroll
    for (int i = 0; i < VW; ++i) xv[i] *= (cg + sv[i]);
    strow<VEC>(u + n * u_bs + (long long)c * dhw, q, valid, xv);
    if (red) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int i = 0; i < VW; ++i)
        if (i < valid) { const float r = rnd_as((const T*)nullptr, xv[i]); t0 += r; t1 = fmaf(r, r, t1); }
      s[0] += (double)t0;
      s[1] += (double)t1;
    }
  ROW_LOOP_END
  if (red) {
    block_sum_d<2>(s, s_red, EW_BLOCK >> 6);
    if (threadIdx.x < 2) atomicAdd(&red[((long long)blockIdx.z * C + blockIdx.y) * 2 + threadIdx.x], s_red[threadIdx.x]);
  }
}
// x: the pair as (2, C, ...) (batch stride x_bs = C * dhw for a contiguous (1, 2C, ...) tensor); sp (2, 1, ...); red_in: the pair's
// raw sums [2C][2]; ch_out [2][C], g_out [C], means_out [2C] are written for the backward pass; red_out (optional) [2][C][2] zeroed.
extern "C" int xh_duse_gate_fc_fwd(void* stream, int dtype, const void* x, long long x_bs, const void* sp, long long sp_bs, void* u, long long u_bs,
                                   int C, long long DHW, const double* red_in, const float* wc, const float* bc, const float* w1,
                                   const float* b1, const float* w2, const float* b2, float* ch_out, float* g_out, float* means_out,
                                   double* red_out) {
  if (!x || !sp || !u || !red_in || !wc || !bc || !w1 || !b1 || !w2 || !b2 || !ch_out || !g_out || !means_out || C <= 0 || C > 32 || DHW <= 0)
    return XH_ERR_ARG;
  DuseFc f{red_in, 1.0 / (double)DHW, wc, bc, w1, b1, w2, b2, g_out, ch_out, means_out};
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, {
    dim3 grid = red_out ? red_grid<T>(DHW, C, 2) : row_grid<T>(DHW, C, 2);
    if (vec_ok<T>(DHW, {x_bs, sp_bs, u_bs})) hipLaunchKernelGGL((duse_gate_fc_fwd_kernel<T, true>), grid, dim3(EW_BLOCK), 0, st, (const T*)x, x_bs, (const T*)sp, sp_bs, (T*)u, u_bs, C, DHW, red_out, f);
    else hipLaunchKernelGGL((duse_gate_fc_fwd_kernel<T, false>), grid, dim3(EW_BLOCK), 0, st, (const T*)x, x_bs, (const T*)sp, sp_bs, (T*)u, u_bs, C, DHW, red_out, f);
  });
  return xh_launch_status();
}
struct DuseFcBwd {
  const float *means, *g, *ch;       // [2C], [C], [2][C] saved by the forward pass
  const double* dch;                 // [2][C]: sum over voxels of du * x per channel (xh_duse_gate_bwd)
  const float *wc, *w1, *w2;
  float *dwc, *dbc, *dw1, *db1, *dw2, *db2;     // ACCUMULATED into by the first workgroup
  float inv_count;
};
// dx[c] += w[c] * d + dmean[c] for the pair (1, 2C, ...): xh_rank1_add with the pooled-mean gradient derived in-kernel
template <typename T, bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void rank1_add_fc_kernel(T* dx, long long dx_bs, const T* d, long long d_bs, const float* w, int C2,
                                                               long long dhw, const DuseFcBwd f) {
  __shared__ float s_p[64], s_dg[32], s_k;
  const int C = C2 / 2, tid = threadIdx.x, kk = blockIdx.y;
  for (int i = tid; i < C2; i += EW_BLOCK) { const float ch = f.ch[i]; s_p[i] = (float)f.dch[i] * ch * (1.f - ch); }
  __syncthreads();
  for (int k = tid; k < C; k += EW_BLOCK) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += f.w1[c * C + k] * s_p[c] + f.w2[c * C + k] * s_p[C + c];
    s_dg[k] = a;
  }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += f.wc[c * C2 + kk] * s_dg[c];
    s_k = a * f.inv_count;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0) {               // parameter gradients, once
    for (int i = tid; i < C * C; i += EW_BLOCK) {
      const int c = i / C, k = i % C;
      f.dw1[i] += s_p[c] * f.g[k];
      f.dw2[i] += s_p[C + c] * f.g[k];
    }
    for (int i = tid; i < C * C2; i += EW_BLOCK) f.dwc[i] += s_dg[i / C2] * f.means[i % C2];
    for (int c = tid; c < C; c += EW_BLOCK) { f.db1[c] += s_p[c]; f.db2[c] += s_p[C + c]; f.dbc[c] += s_dg[c]; }
  }
  __syncthreads();
  const float wc = w[blockIdx.y], kc = s_k;
  ROW_LOOP_BEGIN
    float o[VW], dv[VW];
    T* dp = dx + n * dx_bs + (long long)c * dhw;
    ldrow<VEC>((const T*)dp, q, valid, o);
    ldrow<VEC>(d + n * d_bs, q, valid, dv);
#pragma unroll
    for (int i = 0; i < VW; ++i) o[i] += fmaf(wc, dv[i], kc);
    strow<VEC>(dp, q, valid, o);
  ROW_LOOP_END
}
extern "C" int xh_rank1_add_fc(void* stream, int dtype, void* dx, long long dx_bs, const void* d, long long d_bs, const float* w, int C2,
                               long long DHW, const float* means, const float* g, const float* ch, const double* dch, const float* wc,
                               const float* w1, const float* w2, float* dwc, float* dbc, float* dw1, float* db1, float* dw2, float* db2) {
  if (!dx || !d || !w || !means || !g || !ch || !dch || !wc || !w1 || !w2 || !dwc || !dbc || !dw1 || !db1 || !dw2 || !db2 || C2 <= 0 || (C2 & 1) ||
      C2 > 64 || DHW <= 0)
    return XH_ERR_ARG;
  DuseFcBwd f{means, g, ch, dch, wc, w1, w2, dwc, dbc, dw1, db1, dw2, db2, 1.f / (float)DHW};
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, {
    dim3 grid = row_grid<T>(DHW, C2, 1);
    if (vec_ok<T>(DHW, {dx_bs, d_bs})) hipLaunchKernelGGL((rank1_add_fc_kernel<T, true>), grid, dim3(EW_BLOCK), 0, st, (T*)dx, dx_bs, (const T*)d, d_bs, w, C2, DHW, f);
    else hipLaunchKernelGGL((rank1_add_fc_kernel<T, false>), grid, dim3(EW_BLOCK), 0, st, (T*)dx, dx_bs, (const T*)d, d_bs, w, C2, DHW, f);
  });
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- skip-return tail
// r_c = relu(relu(t_c*sc+sh) + x_c);  a = sigmoid(w0*max_c r + w1*mean_c r)
// fin.red != nullptr (one sample): the training-mode BatchNorm in front of the tail is finalised HERE from the raw channel sums of
// t (every workgroup for all C <= 64 channels; the first one leaves sc / sh / mean / rstd for the backward pass and advances
// the running statistics) -- the one-workgroup xh_norm_finalize launch between the conv and this pass disappears.
struct SkrFin {
  const double* red; double inv_count;
  const float *gamma, *beta; float *rm, *rv; int steps;
  float *o_sc, *o_sh, *o_mean, *o_rstd;
};
template <typename T, bool VEC, int CBT = CB, int VWX = 0>
__global__ __launch_bounds__(256) void skr_tail_fwd_kernel(const T* __restrict__ t, const T* __restrict__ x, const float* sc,
                                                          const float* sh, const float* w2, T* __restrict__ a, int C,
                                                          long long dhw, const SkrFin fin) {
  __shared__ float s_sc[64], s_sh[64];
  if (fin.red) {
    const int c = threadIdx.x;
    if (c < C) {
      float sc_, sh_, m_, r_;
      in_finalize(fin.red[2 * c], fin.red[2 * c + 1], fin.inv_count, sc_, sh_, m_, r_);
      const float g_ = fin.gamma ? fin.gamma[c] : 1.f, b_ = fin.beta ? fin.beta[c] : 0.f;
      sc_ *= g_; sh_ = fmaf(sh_, g_, b_);
      s_sc[c] = sc_; s_sh[c] = sh_;
      if (blockIdx.x == 0 && blockIdx.z == 0) {
        fin.o_sc[c] = sc_; fin.o_sh[c] = sh_; fin.o_mean[c] = m_; fin.o_rstd[c] = r_;
        if (fin.rm && fin.rv && fin.steps > 0) {
          const double M = 1.0 / fin.inv_count, mean = fin.red[2 * c] * fin.inv_count;
          double var = fin.red[2 * c + 1] * fin.inv_count - mean * mean;
          if (var < 0) var = 0;
          const double keep = pow(0.9, (double)fin.steps), unb = var * M / (M > 1 ? M - 1 : 1);
          fin.rm[c] = (float)(keep * fin.rm[c] + (1 - keep) * mean);
          fin.rv[c] = (float)(keep * fin.rv[c] + (1 - keep) * unb);
        }
      }
    }
    __syncthreads();
    sc = s_sc; sh = s_sh;                                // (one sample: n * C + c == c)
  }
  const float w0 = w2[0], w1 = w2[1];
  VOX_LOOP_BEGIN_W(VWX)
    float m[VW], s[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) { m[v] = -INFINITY; s[v] = 0.f; }
    for (int c0 = 0; c0 < C; c0 += CBT) {
      float tv[CBT][VW], xv[CBT][VW];
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        const long long o = ((long long)n * C + min(c0 + j, C - 1)) * dhw;
        ldrow<VEC>(t + o, q, valid, tv[j]);
        ldrow<VEC>(x + o, q, valid, xv[j]);
      }
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
        const float scv = sc[n * C + c0 + j], shv = sh[n * C + c0 + j];
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          float y = tv[j][v] * scv + shv;
          y = y > 0.f ? y : 0.f;
          float r = y + xv[j][v];
          r = r > 0.f ? r : 0.f;
          m[v] = r > m[v] ? r : m[v];
          s[v] += r;
        }
              }
      }
    }
    float out[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) out[v] = sigmoidf_(w0 * m[v] + w1 * s[v] / (float)C);
    strow<VEC>(a + (long long)n * dhw, q, valid, out);
  VOX_LOOP_END
}
template <typename T, bool VEC, int CBT = CB, int VWX = 0>
__global__ __launch_bounds__(256) void skr_tail_bwd_kernel(const T* __restrict__ t, const T* __restrict__ x, const float* sc,
                                                          const float* sh, const float* w2, const T* __restrict__ a,
                                                          const T* __restrict__ da, T* __restrict__ dtg, T* dx,
                                                          double* dw2acc, int C, long long dhw, int acc_dx, float* dw2f) {
  __shared__ double s_red[4 * 2];
  const float w0 = w2[0], w1 = w2[1];
  double sacc[2] = {0.0, 0.0};
  VOX_LOOP_BEGIN_W(VWX)
    float m[VW], s[VW], av[VW], dpre[VW];
    int arg[VW];
    ldrow<VEC>(a + (long long)n * dhw, q, valid, av);
    ldrow<VEC>(da + (long long)n * dhw, q, valid, dpre);
#pragma unroll
    for (int v = 0; v < VW; ++v) { m[v] = -INFINITY; s[v] = 0.f; arg[v] = 0; }
    for (int c0 = 0; c0 < C; c0 += CBT) {
      float tv[CBT][VW], xv[CBT][VW];
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        const long long o = ((long long)n * C + min(c0 + j, C - 1)) * dhw;
        ldrow<VEC>(t + o, q, valid, tv[j]);
        ldrow<VEC>(x + o, q, valid, xv[j]);
      }
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
        const float scv = sc[n * C + c0 + j], shv = sh[n * C + c0 + j];
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          float y = tv[j][v] * scv + shv;
          y = y > 0.f ? y : 0.f;
          float r = y + xv[j][v];
          r = r > 0.f ? r : 0.f;
          if (r > m[v]) { m[v] = r; arg[v] = c0 + j; }
          s[v] += r;
        }
              }
      }
    }
    float t0 = 0.f, t1 = 0.f;                           // lanes past the row end: a = da = 0, so they add nothing
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      dpre[v] = dpre[v] * av[v] * (1.f - av[v]);
      t0 = fmaf(dpre[v], m[v], t0);
      t1 = fmaf(dpre[v], s[v] / (float)C, t1);
    }
    sacc[0] += (double)t0;
    sacc[1] += (double)t1;
    for (int c0 = 0; c0 < C; c0 += CBT) {
      float tv[CBT][VW], xv[CBT][VW], odx[CBT][VW];
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        const long long o = ((long long)n * C + min(c0 + j, C - 1)) * dhw;
        ldrow<VEC>(t + o, q, valid, tv[j]);
        ldrow<VEC>(x + o, q, valid, xv[j]);
        if (acc_dx) {
          ldrow<VEC>((const T*)dx + o, q, valid, odx[j]);
        } else {
#pragma unroll
          for (int v = 0; v < VW; ++v) odx[j][v] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < CBT; ++j) {
        if (c0 + j < C) {                 // predicate, not break: a break in the unrolled loop sends the arrays to scratch
        const long long o = ((long long)n * C + c0 + j) * dhw;
        const float scv = sc[n * C + c0 + j], shv = sh[n * C + c0 + j];
        float odt[VW];
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          const float yraw = tv[j][v] * scv + shv;
          const float y = yraw > 0.f ? yraw : 0.f;
          const float r = y + xv[j][v];
          float dr = dpre[v] * w1 / (float)C + (c0 + j == arg[v] ? dpre[v] * w0 : 0.f);
          dr = r > 0.f ? dr : 0.f;
          odx[j][v] = dr + odx[j][v];
          odt[v] = yraw > 0.f ? dr : 0.f;
        }
        strow<VEC>(dx + o, q, valid, odx[j]);
        strow<VEC>(dtg + o, q, valid, odt);
              }
      }
    }
  VOX_LOOP_END
  block_sum_d<2>(sacc, s_red, 4);
  if (threadIdx.x < 2) {
    if (dw2f) atomicAdd(&dw2f[threadIdx.x], (float)s_red[threadIdx.x]);
    else atomicAdd(&dw2acc[threadIdx.x], s_red[threadIdx.x]);
  }
}
static int launch_skr_tail_fwd(void* stream, int dtype, const void* t, const void* x, const float* sc, const float* sh, const float* w2,
                               void* a, int N, int C, long long DHW, const SkrFin& fin) {
  if (!t || !x || !w2 || !a || N <= 0 || C <= 0 || DHW <= 0 || N > 65535) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, {
    const dim3 grid = vox_grid<T>(DHW, N);
    if (vec_ok<T>(DHW, {}) && deep_grid(grid, C))
      hipLaunchKernelGGL((skr_tail_fwd_kernel<T, true, DEEP_CB, DEEP_VW>), vox_grid_deep<T>(DHW, N), dim3(256), 0, st, (const T*)t, (const T*)x, sc, sh, w2, (T*)a, C, DHW, fin);
    else if (vec_ok<T>(DHW, {}))
      hipLaunchKernelGGL((skr_tail_fwd_kernel<T, true>), grid, dim3(256), 0, st, (const T*)t, (const T*)x, sc, sh, w2, (T*)a, C, DHW, fin);
    else
      hipLaunchKernelGGL((skr_tail_fwd_kernel<T, false>), grid, dim3(256), 0, st, (const T*)t, (const T*)x, sc, sh, w2, (T*)a, C, DHW, fin);
  });
  return xh_launch_status();
}
extern "C" int xh_skr_tail_fwd(void* stream, int dtype, const void* t, const void* x, const float* sc, const float* sh, const float* w2,
                               void* a, int N, int C, long long DHW) {
  if (!sc || !sh) return XH_ERR_ARG;
  return launch_skr_tail_fwd(stream, dtype, t, x, sc, sh, w2, a, N, C, DHW, SkrFin{});
}
// xh_skr_tail_fwd with the training-mode BatchNorm in front of it finalised in the same launch (one sample, C <= 64): red [C][2]
// raw sums of t; sc / sh / mean / rstd [C] are written (backward pass), running_mean / running_var advanced by `steps` updates.
extern "C" int xh_skr_tail_bn_fwd(void* stream, int dtype, const void* t, const void* x, const double* red, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, int steps, const float* w2, void* a, int C,
                                  long long DHW, float* sc, float* sh, float* mean, float* rstd) {
  if (!red || !sc || !sh || !mean || !rstd || C > 64 || steps < 0) return XH_ERR_ARG;
  SkrFin f{red, 1.0 / (double)DHW, gamma, beta, running_mean, running_var, steps, sc, sh, mean, rstd};
  return launch_skr_tail_fwd(stream, dtype, t, x, sc, sh, w2, a, 1, C, DHW, f);
}
extern "C" int xh_skr_tail_bwd(void* stream, int dtype, const void* t, const void* x, const float* sc, const float* sh,
                               const float* w2, const void* a, const void* da, void* dtg, void* dx, double* dw2,
                               int N, int C, long long DHW, int acc_dx, float* dw2_f32) {
  if (!t || !x || !sc || !sh || !w2 || !a || !da || !dtg || !dx || (!dw2 && !dw2_f32) || N <= 0 || C <= 0 || DHW <= 0 || N > 65535) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define SKRB(V, ...) hipLaunchKernelGGL((skr_tail_bwd_kernel<T, V, ##__VA_ARGS__>), grid, dim3(256), 0, st, (const T*)t, (const T*)x, sc, sh, w2, \
                                        (const T*)a, (const T*)da, (T*)dtg, (T*)dx, dw2, C, DHW, acc_dx, dw2_f32)
  XH_DISPATCH_T(dtype, {
    // one pair of fp64 atomics per workgroup on dw2: at most 1024 of them
    // (the narrow-lane instance -- DEEP_VW -- was measured here too: 21.0 -> 22.4 us at 16^3, 14.1 -> 24.4 us at 64^3: not used)
    const dim3 grid = vox_grid<T>(DHW, N, 1024);
    if (vec_ok<T>(DHW, {}) && deep_grid(grid, C)) SKRB(true, 8);
    else if (vec_ok<T>(DHW, {})) SKRB(true);
    else SKRB(false);
  });
#undef SKRB
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------- parameter compositions
// Linear stages with no non-linearity between them are applied as one conv with composed weights.
// The composition and its adjoint are parameter-sized: one small kernel each instead of ~40 framework launches.
//
// AttenModule2 (buildingblocks.py:271-296): grouped k^3 conv (groups = input channels, E outputs each) followed by a 1x1
// conv to one channel.  w[0] = seg gate over input channels [0, NS) (the remaining NE-NS rows are zero), w[1] = enc gate
// over all NE input channels;  w[r][ci][tap] = sum_e w2[ci*E+e] * w1[(ci*E+e)][tap],  b[r] = sum_o w2[o]*b1[o] + b2.
struct AttenCompose {
  const float *seg_w, *seg_b, *seg2_w, *seg2_b, *enc_w, *enc_b, *enc2_w, *enc2_b;
  int NS, NE, E, K3;
};
__device__ __forceinline__ void compose_atten_fwd_body(const AttenCompose& a, float* w, float* b, int bx, int nbx) {
  const int total = 2 * a.NE * a.K3;
  for (int idx = bx * 256 + threadIdx.x; idx < total; idx += nbx * 256) {
    const int tap = idx % a.K3, ci = (idx / a.K3) % a.NE, r = idx / (a.K3 * a.NE);
    const float* w1 = r ? a.enc_w : a.seg_w;
    const float* w2 = r ? a.enc2_w : a.seg2_w;
    float v = 0.f;
    if (r == 1 || ci < a.NS)
      for (int e = 0; e < a.E; ++e) v = fmaf(w2[ci * a.E + e], w1[(long long)(ci * a.E + e) * a.K3 + tap], v);
    w[idx] = v;
  }
  if (bx == 0 && threadIdx.x < 2) {
    const int r = threadIdx.x;
    const float* b1 = r ? a.enc_b : a.seg_b;
    const float* w2 = r ? a.enc2_w : a.seg2_w;
    const int no = (r ? a.NE : a.NS) * a.E;
    float v = r ? a.enc2_b[0] : a.seg2_b[0];
    for (int o = 0; o < no; ++o) v = fmaf(w2[o], b1[o], v);
    b[r] = v;
  }
}
__global__ __launch_bounds__(256) void compose_atten_fwd_kernel(AttenCompose a, float* w, float* b) {
  compose_atten_fwd_body(a, w, b, blockIdx.x, gridDim.x);
}
struct AttenComposeGrad { float *seg_w, *seg_b, *seg2_w, *seg2_b, *enc_w, *enc_b, *enc2_w, *enc2_b; };
// one workgroup per first-stage output channel o (seg: NS*E of them, then enc: NE*E); every gradient is accumulated (+=)
__device__ __forceinline__ void compose_atten_bwd_body(const AttenCompose& a, const AttenComposeGrad& g, const float* gw, const float* gb,
                                                       int bx, float* s_red) {
  const int nso = a.NS * a.E;
  if (bx >= nso + a.NE * a.E) return;                    // (multi-job launches: the grid is sized for the largest job)
  const int r = bx >= nso ? 1 : 0;
  const int o = r ? bx - nso : bx;
  const int ci = o / a.E;
  const float* w1 = (r ? a.enc_w : a.seg_w) + (long long)o * a.K3;
  const float w2 = (r ? a.enc2_w : a.seg2_w)[o];
  float* dw1 = (r ? g.enc_w : g.seg_w) + (long long)o * a.K3;
  const float* grow = gw + (long long)(r * a.NE + ci) * a.K3;
  float acc[1] = {0.f};
  for (int tap = threadIdx.x; tap < a.K3; tap += 256) {
    const float gv = grow[tap];
    dw1[tap] += w2 * gv;
    acc[0] = fmaf(w1[tap], gv, acc[0]);
  }
  block_sum<1>(acc, s_red, 4);
  if (threadIdx.x == 0) {
    const float b1 = (r ? a.enc_b : a.seg_b)[o];
    (r ? g.enc2_w : g.seg2_w)[o] += s_red[0] + b1 * gb[r];
    (r ? g.enc_b : g.seg_b)[o] += w2 * gb[r];
    if (o == 0) (r ? g.enc2_b : g.seg2_b)[0] += gb[r];
  }
}
__global__ __launch_bounds__(256) void compose_atten_bwd_kernel(AttenCompose a, AttenComposeGrad g, const float* gw,
                                                               const float* gb) {
  __shared__ float s_red[4];
  compose_atten_bwd_body(a, g, gw, gb, blockIdx.x, s_red);
}
extern "C" int xh_compose_atten_fwd(void* stream, const float* seg_w, const float* seg_b, const float* seg2_w,
                                    const float* seg2_b, const float* enc_w, const float* enc_b, const float* enc2_w,
                                    const float* enc2_b, int NS, int NE, int E, int K3, float* w, float* b) {
  if (!seg_w || !seg_b || !seg2_w || !seg2_b || !enc_w || !enc_b || !enc2_w || !enc2_b || !w || !b) return XH_ERR_ARG;
  if (NS <= 0 || NE < NS || E <= 0 || K3 <= 0) return XH_ERR_ARG;
  AttenCompose a{seg_w, seg_b, seg2_w, seg2_b, enc_w, enc_b, enc2_w, enc2_b, NS, NE, E, K3};
  hipLaunchKernelGGL(compose_atten_fwd_kernel, dim3(cdiv(2 * NE * K3, 256)), dim3(256), 0, (hipStream_t)stream, a, w, b);
  return xh_launch_status();
}
extern "C" int xh_compose_atten_bwd(void* stream, const float* seg_w, const float* seg_b, const float* seg2_w,
                                    const float* enc_w, const float* enc_b, const float* enc2_w, int NS, int NE, int E,
                                    int K3, const float* gw, const float* gb, float* d_seg_w, float* d_seg_b,
                                    float* d_seg2_w, float* d_seg2_b, float* d_enc_w, float* d_enc_b, float* d_enc2_w,
                                    float* d_enc2_b) {
  if (!seg_w || !seg_b || !seg2_w || !enc_w || !enc_b || !enc2_w || !gw || !gb || !d_seg_w || !d_seg_b || !d_seg2_w ||
      !d_seg2_b || !d_enc_w || !d_enc_b || !d_enc2_w || !d_enc2_b)
    return XH_ERR_ARG;
  if (NS <= 0 || NE < NS || E <= 0 || K3 <= 0) return XH_ERR_ARG;
  AttenCompose a{seg_w, seg_b, seg2_w, nullptr, enc_w, enc_b, enc2_w, nullptr, NS, NE, E, K3};
  AttenComposeGrad g{d_seg_w, d_seg_b, d_seg2_w, d_seg2_b, d_enc_w, d_enc_b, d_enc2_w, d_enc2_b};
  hipLaunchKernelGGL(compose_atten_bwd_kernel, dim3((NS + NE) * E), dim3(256), 0, (hipStream_t)stream, a, g, gw, gb);
  return xh_launch_status();
}

// DuSEAttention (modules/DuSFE.py:129-144): conv_comb o [conv_squeeze_ch1 | conv_squeeze_ch2] as one 2C->1 1x1 conv
// (sqw[2C], sqb[1]) and the two 3^3 adjust convs stacked (adjw[2][27], adjb[2]).
struct DuseCompose {
  const float *comb_w, *comb_b, *sq1_w, *sq1_b, *sq2_w, *sq2_b, *adj1_w, *adj1_b, *adj2_w, *adj2_b;
  int C;
};
__device__ __forceinline__ void compose_duse_fwd_body(const DuseCompose& a, float* sqw, float* sqb, float* adjw, float* adjb) {
  const float w0 = a.comb_w[0], w1 = a.comb_w[1];
  for (int j = threadIdx.x; j < 2 * a.C; j += 256) sqw[j] = j < a.C ? w0 * a.sq1_w[j] : w1 * a.sq2_w[j - a.C];
  for (int j = threadIdx.x; j < 54; j += 256) adjw[j] = j < 27 ? a.adj1_w[j] : a.adj2_w[j - 27];
  if (threadIdx.x == 0) {
    sqb[0] = w0 * a.sq1_b[0] + w1 * a.sq2_b[0] + a.comb_b[0];
    adjb[0] = a.adj1_b[0];
    adjb[1] = a.adj2_b[0];
  }
}
__global__ __launch_bounds__(256) void compose_duse_fwd_kernel(DuseCompose a, float* sqw, float* sqb, float* adjw, float* adjb) {
  compose_duse_fwd_body(a, sqw, sqb, adjw, adjb);
}
struct DuseComposeGrad { float *comb_w, *comb_b, *sq1_w, *sq1_b, *sq2_w, *sq2_b, *adj1_w, *adj1_b, *adj2_w, *adj2_b; };
__device__ __forceinline__ void compose_duse_bwd_body(const DuseCompose& a, const DuseComposeGrad& g, const float* dsqw, const float* dsqb,
                                                      const float* dadjw, const float* dadjb, float* s_red) {
  const float w0 = a.comb_w[0], w1 = a.comb_w[1];
  float acc[2] = {0.f, 0.f};
  for (int j = threadIdx.x; j < a.C; j += 256) {
    g.sq1_w[j] += w0 * dsqw[j];
    g.sq2_w[j] += w1 * dsqw[a.C + j];
    acc[0] = fmaf(a.sq1_w[j], dsqw[j], acc[0]);
    acc[1] = fmaf(a.sq2_w[j], dsqw[a.C + j], acc[1]);
  }
  for (int j = threadIdx.x; j < 27; j += 256) { g.adj1_w[j] += dadjw[j]; g.adj2_w[j] += dadjw[27 + j]; }
  block_sum<2>(acc, s_red, 4);
  if (threadIdx.x == 0) {
    const float db = dsqb[0];
    g.comb_w[0] += s_red[0] + a.sq1_b[0] * db;
    g.comb_w[1] += s_red[1] + a.sq2_b[0] * db;
    g.sq1_b[0] += w0 * db;
    g.sq2_b[0] += w1 * db;
    g.comb_b[0] += db;
    g.adj1_b[0] += dadjb[0];
    g.adj2_b[0] += dadjb[1];
  }
}
__global__ __launch_bounds__(256) void compose_duse_bwd_kernel(DuseCompose a, DuseComposeGrad g, const float* dsqw,
                                                              const float* dsqb, const float* dadjw, const float* dadjb) {
  __shared__ float s_red[4 * 2];
  compose_duse_bwd_body(a, g, dsqw, dsqb, dadjw, dadjb, s_red);
}

// ---- all parameter compositions of a step in ONE launch per direction (xh_compose_multi): the three AttenModule2 gates, the
// three DuSE blocks and the segmentation head final_conv o sfinals were twelve + ~ten launches of one to 24 workgroups each,
// i.e. pure launch latency (~5 us apiece in a replayed graph).  blockIdx.y = job.
// head: W[co][ci] = sum_m wf[co][m] ws[m][ci], b[co] = bf[co] + sum_m wf[co][m] bs[m]      (RA_HVED.py:192-199,640: 1x1 o 1x1)
struct ComposeJobs {
  int na, nd, nh, bwd;
  AttenCompose a[XH_COMPOSE_MAX]; AttenComposeGrad ag[XH_COMPOSE_MAX];
  float* aw[XH_COMPOSE_MAX]; float* ab[XH_COMPOSE_MAX]; const float* agw[XH_COMPOSE_MAX]; const float* agb[XH_COMPOSE_MAX];
  DuseCompose d[XH_COMPOSE_MAX]; DuseComposeGrad dg[XH_COMPOSE_MAX];
  float* dout[XH_COMPOSE_MAX][4]; const float* dgout[XH_COMPOSE_MAX][4];
  xh_head_job h;
  int ns; xh_sep_job s[XH_SEP_MAX];                     // depthwise 3^3 o pointwise 1x1 (sa_modules/sa_module.py:79-85) as one dense 3^3 conv
  float* zero_buf; long long zero_n;                    // forward: also cleared (the composed tensors' gradient buffers of the step)
};
__global__ __launch_bounds__(256) void compose_multi_kernel(const ComposeJobs j) {
  __shared__ float s_red[4 * 2];
  const int job = blockIdx.y;
  if (j.zero_buf)
    for (long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < j.zero_n;
         i += (long long)gridDim.x * gridDim.y * 256)
      j.zero_buf[i] = 0.f;
  if (job < j.na) {
    if (!j.bwd) compose_atten_fwd_body(j.a[job], j.aw[job], j.ab[job], blockIdx.x, gridDim.x);
    else compose_atten_bwd_body(j.a[job], j.ag[job], j.agw[job], j.agb[job], blockIdx.x, s_red);
    return;
  }
  if (job >= j.na + j.nd + j.nh) {
    // W[co][ci][t] = pw[co][ci] * dw[ci][t]:  (pointwise o depthwise)(x) as ONE dense k^3 conv (no non-linearity between them).
    // Every workgroup of the job's grid row takes a share of the elements (as ONE workgroup per job the backward's 256 + 432 dot
    // products of 27 / 16 strided loads were the launch's long pole: 15 us)
    const xh_sep_job& q = j.s[job - j.na - j.nd - j.nh];
    const int C = q.C, K3 = q.K3, t = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
    if (!j.bwd) {
      for (int i = t; i < C * C * K3; i += nt) {
        const int tap = i % K3, ci = (i / K3) % C, co = i / (K3 * C);
        q.w[i] = q.pw[co * C + ci] * q.dw[ci * K3 + tap];
      }
    } else {
      for (int i = t; i < C * C; i += nt) {               // d pw[co][ci] += sum_t gw[co][ci][t] dw[ci][t]
        const int ci = i % C;
        float v = 0.f;
        for (int tap = 0; tap < K3; ++tap) v = fmaf(q.gw[(long long)i * K3 + tap], q.dw[ci * K3 + tap], v);
        q.g_pw[i] += v;
      }
      for (int i = t; i < C * K3; i += nt) {              // d dw[ci][t] += sum_co gw[co][ci][t] pw[co][ci]
        const int tap = i % K3, ci = i / K3;
        float v = 0.f;
        for (int co = 0; co < C; ++co) v = fmaf(q.gw[((long long)co * C + ci) * K3 + tap], q.pw[co * C + ci], v);
        q.g_dw[i] += v;
      }
    }
    return;
  }
  if (blockIdx.x != 0) return;
  if (job < j.na + j.nd) {
    const int k = job - j.na;
    if (!j.bwd) compose_duse_fwd_body(j.d[k], j.dout[k][0], j.dout[k][1], j.dout[k][2], j.dout[k][3]);
    else compose_duse_bwd_body(j.d[k], j.dg[k], j.dgout[k][0], j.dgout[k][1], j.dgout[k][2], j.dgout[k][3], s_red);
    return;
  }
  const xh_head_job& h = j.h;
  const int t = threadIdx.x;
  if (!j.bwd) {
    for (int i = t; i < h.Co * h.Ci; i += 256) {
      const int co = i / h.Ci, ci = i % h.Ci;
      float v = 0.f;
      for (int m = 0; m < h.Cm; ++m) v = fmaf(h.wf[co * h.Cm + m], h.ws[m * h.Ci + ci], v);
      h.w[i] = v;
    }
    for (int co = t; co < h.Co; co += 256) {
      float v = h.bf[co];
      for (int m = 0; m < h.Cm; ++m) v = fmaf(h.wf[co * h.Cm + m], h.bs[m], v);
      h.b[co] = v;
    }
  } else {
    for (int i = t; i < h.Co * h.Cm; i += 256) {          // d wf[co][m] += sum_ci gw[co][ci] ws[m][ci] + gb[co] bs[m]
      const int co = i / h.Cm, m = i % h.Cm;
      float v = h.gb[co] * h.bs[m];
      for (int ci = 0; ci < h.Ci; ++ci) v = fmaf(h.gw[co * h.Ci + ci], h.ws[m * h.Ci + ci], v);
      h.dwf[i] += v;
    }
    for (int i = t; i < h.Cm * h.Ci; i += 256) {          // d ws[m][ci] += sum_co wf[co][m] gw[co][ci]
      const int m = i / h.Ci, ci = i % h.Ci;
      float v = 0.f;
      for (int co = 0; co < h.Co; ++co) v = fmaf(h.wf[co * h.Cm + m], h.gw[co * h.Ci + ci], v);
      h.dws[i] += v;
    }
    for (int m = t; m < h.Cm; m += 256) {
      float v = 0.f;
      for (int co = 0; co < h.Co; ++co) v = fmaf(h.wf[co * h.Cm + m], h.gb[co], v);
      h.dbs[m] += v;
    }
    for (int co = t; co < h.Co; co += 256) h.dbf[co] += h.gb[co];
  }
}
extern "C" int xh_compose_multi(void* stream, int bwd, int na, const xh_atten_job* aj, int nd, const xh_duse_job* dj, int nh,
                                const xh_head_job* hj, int ns, const xh_sep_job* sj, float* zero_buf, long long zero_n) {
  if (na < 0 || nd < 0 || nh < 0 || ns < 0 || na > XH_COMPOSE_MAX || nd > XH_COMPOSE_MAX || nh > 1 || ns > XH_SEP_MAX || na + nd + nh + ns == 0)
    return XH_ERR_ARG;
  if ((na && !aj) || (nd && !dj) || (nh && !hj) || (ns && !sj) || zero_n < 0 || (zero_n > 0 && !zero_buf)) return XH_ERR_ARG;
  ComposeJobs j;
  j.na = na; j.nd = nd; j.nh = nh; j.ns = ns; j.bwd = bwd ? 1 : 0;
  for (int i = 0; i < ns; ++i) {
    j.s[i] = sj[i];
    if (!sj[i].dw || !sj[i].pw || sj[i].C <= 0 || sj[i].K3 <= 0 || (bwd ? (!sj[i].gw || !sj[i].g_dw || !sj[i].g_pw) : !sj[i].w)) return XH_ERR_ARG;
  }
  j.zero_buf = zero_n > 0 ? zero_buf : nullptr; j.zero_n = zero_n;
  int gx = ns > 0 ? 8 : 1;                               // the separable-conv jobs spread over their grid row
  for (int i = 0; i < na; ++i) {
    const xh_atten_job& s = aj[i];
    if (s.NS <= 0 || s.NE < s.NS || s.E <= 0 || s.K3 <= 0) return XH_ERR_ARG;
    for (int k = 0; k < 8; ++k)
      if (!s.p[k] || (bwd && !s.g[k])) return XH_ERR_ARG;
    if (bwd ? (!s.gw || !s.gb) : (!s.w || !s.b)) return XH_ERR_ARG;
    j.a[i] = AttenCompose{s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], s.p[5], s.p[6], s.p[7], s.NS, s.NE, s.E, s.K3};
    j.ag[i] = AttenComposeGrad{s.g[0], s.g[1], s.g[2], s.g[3], s.g[4], s.g[5], s.g[6], s.g[7]};
    j.aw[i] = s.w; j.ab[i] = s.b; j.agw[i] = s.gw; j.agb[i] = s.gb;
    const int need = bwd ? (s.NS + s.NE) * s.E : cdiv(2 * s.NE * s.K3, 256);
    if (need > gx) gx = need;
  }
  for (int i = 0; i < nd; ++i) {
    const xh_duse_job& s = dj[i];
    if (s.C <= 0) return XH_ERR_ARG;
    for (int k = 0; k < 10; ++k)
      if (!s.p[k] || (bwd && !s.g[k])) return XH_ERR_ARG;
    for (int k = 0; k < 4; ++k)
      if (bwd ? !s.gout[k] : !s.out[k]) return XH_ERR_ARG;
    j.d[i] = DuseCompose{s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], s.p[5], s.p[6], s.p[7], s.p[8], s.p[9], s.C};
    j.dg[i] = DuseComposeGrad{s.g[0], s.g[1], s.g[2], s.g[3], s.g[4], s.g[5], s.g[6], s.g[7], s.g[8], s.g[9]};
    for (int k = 0; k < 4; ++k) { j.dout[i][k] = s.out[k]; j.dgout[i][k] = s.gout[k]; }
  }
  if (nh) {
    j.h = *hj;
    if (!j.h.wf || !j.h.bf || !j.h.ws || !j.h.bs || j.h.Co <= 0 || j.h.Cm <= 0 || j.h.Ci <= 0) return XH_ERR_ARG;
    if (bwd ? (!j.h.gw || !j.h.gb || !j.h.dwf || !j.h.dbf || !j.h.dws || !j.h.dbs) : (!j.h.w || !j.h.b)) return XH_ERR_ARG;
  }
  hipLaunchKernelGGL(compose_multi_kernel, dim3(gx, na + nd + nh + ns), dim3(256), 0, (hipStream_t)stream, j);
  return xh_launch_status();
}
extern "C" int xh_compose_duse_fwd(void* stream, const float* const params[10], int C, float* sqw, float* sqb, float* adjw,
                                   float* adjb) {
  if (!params || !sqw || !sqb || !adjw || !adjb || C <= 0) return XH_ERR_ARG;
  for (int i = 0; i < 10; ++i)
    if (!params[i]) return XH_ERR_ARG;
  DuseCompose a{params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7], params[8], params[9], C};
  hipLaunchKernelGGL(compose_duse_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, sqw, sqb, adjw, adjb);
  return xh_launch_status();
}
extern "C" int xh_compose_duse_bwd(void* stream, const float* const params[10], int C, const float* dsqw, const float* dsqb,
                                   const float* dadjw, const float* dadjb, float* const grads[10]) {
  if (!params || !grads || !dsqw || !dsqb || !dadjw || !dadjb || C <= 0) return XH_ERR_ARG;
  for (int i = 0; i < 10; ++i)
    if (!params[i] || !grads[i]) return XH_ERR_ARG;
  DuseCompose a{params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7], params[8], params[9], C};
  DuseComposeGrad g{grads[0], grads[1], grads[2], grads[3], grads[4], grads[5], grads[6], grads[7], grads[8], grads[9]};
  hipLaunchKernelGGL(compose_duse_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, g, dsqw, dsqb, dadjw, dadjb);
  return xh_launch_status();
}


// ======================================================================================================================
// Multi kernels (include/xlstm_hved.h: "Multi-problem launches of the five passes of the latent path").  One launch, a table of up
// to XH_LEVELS_MAX problems in the kernel arguments, a 1-D grid: workgroup b belongs to the problem whose [off[i], off[i + 1])
// holds it and runs that problem's ordinary kernel body at the block coordinates the problem's own launch would have had.
struct MultiHdr { int n; int off[XH_LEVELS_MAX + 1]; int gx[XH_LEVELS_MAX], gy[XH_LEVELS_MAX], gz[XH_LEVELS_MAX]; };
__device__ __forceinline__ int multi_find(const MultiHdr& h, uint3& vb, uint3& vg) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < XH_LEVELS_MAX; ++k)
    if (k < h.n && (int)blockIdx.x >= h.off[k]) i = k;
  const int local = (int)blockIdx.x - h.off[i];
  const int gx = h.gx[i], gy = h.gy[i], gz = h.gz[i];
  if (local >= gx * gy * gz) return -1;
  const int zy = local / gx;
  vb = uint3{(unsigned)(local - zy * gx), (unsigned)(zy % gy), (unsigned)(zy / gy)};
  vg = uint3{(unsigned)gx, (unsigned)gy, (unsigned)gz};
  return i;
}
static void multi_layout(MultiHdr& h, int i, dim3 g) {
  h.gx[i] = (int)g.x; h.gy[i] = (int)g.y; h.gz[i] = (int)g.z;
  h.off[i + 1] = h.off[i] + (int)(g.x * g.y * g.z);
}

struct AffMulti { MultiHdr h; struct P { const void* x; long long x_bs; void* y; long long y_bs; int C; long long dhw; int act; float slope; int vec; AffFin f; } p[XH_LEVELS_MAX]; };
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void affine_act_multi_kernel(const AffMulti m) {
  uint3 vb, vg;
  const int i = multi_find(m.h, vb, vg);
  if (i < 0) return;
  const AffMulti::P& a = m.p[i];
  if (a.vec) affine_act_body<T, true>(vb, vg, (const T*)a.x, a.x_bs, (T*)a.y, a.y_bs, a.C, a.dhw, nullptr, nullptr, a.act, a.slope, a.f);
  else affine_act_body<T, false>(vb, vg, (const T*)a.x, a.x_bs, (T*)a.y, a.y_bs, a.C, a.dhw, nullptr, nullptr, a.act, a.slope, a.f);
}
extern "C" int xh_in_affine_act_multi(void* stream, int dtype, int n, const xh_in_affine_act_args* q) {
  if (n < 1 || n > XH_LEVELS_MAX || !q) return XH_ERR_ARG;
  AffMulti m{};
  m.h.n = n;
  for (int i = 0; i < n; ++i) {
    const xh_in_affine_act_args& a = q[i];
    if (!a.x || !a.y || !a.red || !a.sc || !a.sh || !a.mean || !a.rstd || a.N <= 0 || a.C <= 0 || a.DHW <= 0 || a.C > 65535 || a.N > 65535)
      return XH_ERR_ARG;
    AffMulti::P& p = m.p[i];
    p.x = a.x; p.x_bs = a.x_bs; p.y = a.y; p.y_bs = a.y_bs; p.C = a.C; p.dhw = a.DHW; p.act = a.act; p.slope = a.slope;
    p.f = AffFin{};
    p.f.mode = 0; p.f.red = a.red; p.f.count = (double)a.DHW; p.f.N = a.N;
    p.f.o_sc = a.sc; p.f.o_sh = a.sh; p.f.o_mean = a.mean; p.f.o_rstd = a.rstd;
    if (dtype == XH_F32) { p.vec = vec_ok<float>(a.DHW, {a.x_bs, a.y_bs}); multi_layout(m.h, i, row_grid<float>(a.DHW, a.C, a.N)); }
    else { p.vec = vec_ok<bf16_t>(a.DHW, {a.x_bs, a.y_bs}); multi_layout(m.h, i, row_grid<bf16_t>(a.DHW, a.C, a.N)); }
  }
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL((affine_act_multi_kernel<T>), dim3(m.h.off[n]), dim3(EW_BLOCK), 0, (hipStream_t)stream, m););
  return xh_launch_status();
}

struct AbrMulti { MultiHdr h; struct P { const void* dy; long long dy_bs; const void* x; long long x_bs; int C; long long dhw; const float *sc, *sh; float slope; double* red; int vec; } p[XH_LEVELS_MAX]; };
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void act_bwd_reduce_multi_kernel(const AbrMulti m) {
  uint3 vb, vg;
  const int i = multi_find(m.h, vb, vg);
  if (i < 0) return;
  const AbrMulti::P& a = m.p[i];
  if (a.vec) act_bwd_reduce_body<T, true>(vb, vg, (const T*)a.dy, a.dy_bs, (const T*)a.x, a.x_bs, a.C, a.dhw, a.sc, a.sh, a.slope, a.red);
  else act_bwd_reduce_body<T, false>(vb, vg, (const T*)a.dy, a.dy_bs, (const T*)a.x, a.x_bs, a.C, a.dhw, a.sc, a.sh, a.slope, a.red);
}
extern "C" int xh_act_bwd_reduce_multi(void* stream, int dtype, int n, const xh_act_bwd_reduce_args* q) {
  if (n < 1 || n > XH_LEVELS_MAX || !q) return XH_ERR_ARG;
  AbrMulti m{};
  m.h.n = n;
  for (int i = 0; i < n; ++i) {
    const xh_act_bwd_reduce_args& a = q[i];
    if (!a.dy || !a.x || !a.sc || !a.sh || !a.red || a.N <= 0 || a.C <= 0 || a.DHW <= 0 || a.C > 65535 || a.N > 65535) return XH_ERR_ARG;
    AbrMulti::P& p = m.p[i];
    p.dy = a.dy; p.dy_bs = a.dy_bs; p.x = a.x; p.x_bs = a.x_bs; p.C = a.C; p.dhw = a.DHW; p.sc = a.sc; p.sh = a.sh; p.slope = a.slope; p.red = a.red;
    if (dtype == XH_F32) { p.vec = vec_ok<float>(a.DHW, {a.dy_bs, a.x_bs}); multi_layout(m.h, i, red_grid<float>(a.DHW, a.C, a.N)); }
    else { p.vec = vec_ok<bf16_t>(a.DHW, {a.dy_bs, a.x_bs}); multi_layout(m.h, i, red_grid<bf16_t>(a.DHW, a.C, a.N)); }
  }
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL((act_bwd_reduce_multi_kernel<T>), dim3(m.h.off[n]), dim3(EW_BLOCK), 0, (hipStream_t)stream, m););
  return xh_launch_status();
}

struct IbaMulti {
  MultiHdr h;
  struct P { const void* dy; long long dy_bs; const void* x; long long x_bs; void* dx; long long dx_bs; int C; long long dhw; const double* red;
             const float *mean, *rstd; int stat_rs, have_g; const float *sc, *sh; float slope; int accumulate, vec; } p[XH_LEVELS_MAX];
};
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void in_bwd_apply_multi_kernel(const IbaMulti m) {
  uint3 vb, vg;
  const int i = multi_find(m.h, vb, vg);
  if (i < 0) return;
  const IbaMulti::P& a = m.p[i];
  if (a.vec) in_bwd_apply_body<T, true>(vb, vg, (const T*)a.dy, a.dy_bs, (const T*)a.x, a.x_bs, (T*)a.dx, a.dx_bs, a.C, a.dhw, a.red, a.mean, a.rstd,
                                        a.stat_rs, (double)a.dhw, a.have_g, a.sc, a.sh, a.slope, a.accumulate, (const T*)nullptr, 0, (T*)nullptr, 0, a.C);
  else in_bwd_apply_body<T, false>(vb, vg, (const T*)a.dy, a.dy_bs, (const T*)a.x, a.x_bs, (T*)a.dx, a.dx_bs, a.C, a.dhw, a.red, a.mean, a.rstd,
                                   a.stat_rs, (double)a.dhw, a.have_g, a.sc, a.sh, a.slope, a.accumulate, (const T*)nullptr, 0, (T*)nullptr, 0, a.C);
}
extern "C" int xh_in_bwd_apply_multi(void* stream, int dtype, int n, const xh_in_bwd_apply_args* q) {
  if (n < 1 || n > XH_LEVELS_MAX || !q) return XH_ERR_ARG;
  IbaMulti m{};
  m.h.n = n;
  for (int i = 0; i < n; ++i) {
    const xh_in_bwd_apply_args& a = q[i];
    if (!a.dy || !a.x || !a.dx || !a.red || !a.mean || !a.rstd || a.N <= 0 || a.C <= 0 || a.DHW <= 0 || a.C > 65535 || a.N > 65535 || a.stat_rs < a.C)
      return XH_ERR_ARG;
    if (!a.have_g && (!a.sc || !a.sh)) return XH_ERR_ARG;
    IbaMulti::P& p = m.p[i];
    p.dy = a.dy; p.dy_bs = a.dy_bs; p.x = a.x; p.x_bs = a.x_bs; p.dx = a.dx; p.dx_bs = a.dx_bs; p.C = a.C; p.dhw = a.DHW; p.red = a.red;
    p.mean = a.mean; p.rstd = a.rstd; p.stat_rs = a.stat_rs; p.have_g = a.have_g; p.sc = a.sc; p.sh = a.sh; p.slope = a.slope; p.accumulate = a.accumulate;
    if (dtype == XH_F32) { p.vec = vec_ok<float>(a.DHW, {a.dy_bs, a.x_bs, a.dx_bs}); multi_layout(m.h, i, row_grid<float>(a.DHW, a.C, a.N)); }
    else { p.vec = vec_ok<bf16_t>(a.DHW, {a.dy_bs, a.x_bs, a.dx_bs}); multi_layout(m.h, i, row_grid<bf16_t>(a.DHW, a.C, a.N)); }
  }
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL((in_bwd_apply_multi_kernel<T>), dim3(m.h.off[n]), dim3(EW_BLOCK), 0, (hipStream_t)stream, m););
  return xh_launch_status();
}

// the exact-2x upsampling pair: the lane tiling (TXN = lanes along W) is a template argument and differs between the levels, so the
// multi kernel carries every instance a problem can ask for and branches on the problem's (workgroup-uniform) choice
struct UpfMulti { MultiHdr h; struct P { const void* x; long long x_bs; void* y; long long y_bs; int C, D, H, W, sd, tilesW, tilesH, txn; UpFin fin; } p[XH_LEVELS_MAX]; };
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_fwd_multi_kernel(const UpfMulti m) {
  uint3 vb, vg;
  const int i = multi_find(m.h, vb, vg);
  if (i < 0) return;
  const UpfMulti::P& a = m.p[i];
#define UPF(N_) upsample2x_fwd_body<T, N_, true>(vb, (const T*)a.x, a.x_bs, (T*)a.y, a.y_bs, a.C, a.D, a.H, a.W, a.sd, a.tilesW, a.tilesH, a.fin)
  switch (a.txn) { case 4: UPF(4); break; case 8: UPF(8); break; case 16: UPF(16); break; case 32: UPF(32); break; default: UPF(64); }
#undef UPF
}
extern "C" int xh_upsample2x_in_act_multi(void* stream, int dtype, int n, const xh_upsample2x_in_act_args* q) {
  if (n < 1 || n > XH_LEVELS_MAX || !q) return XH_ERR_ARG;
  if (g_xh_disable & 2) return 1;
  UpfMulti m{};
  m.h.n = n;
  for (int i = 0; i < n; ++i) {
    const xh_upsample2x_in_act_args& a = q[i];
    if (!a.x || !a.y || !a.red || !a.sc || !a.sh || !a.mean || !a.rstd || a.N <= 0 || a.C <= 0 || a.D <= 0 || a.H <= 0 || a.W <= 0) return XH_ERR_ARG;
    UpfMulti::P& p = m.p[i];
    int dsegs;
    const bool ok = dtype == XH_F32 ? upsample2x_plan<float>(a.N, a.C, a.D, a.H, a.W, a.x_bs, a.y_bs, p.txn, p.tilesW, p.tilesH, p.sd, dsegs)
                                    : upsample2x_plan<bf16_t>(a.N, a.C, a.D, a.H, a.W, a.x_bs, a.y_bs, p.txn, p.tilesW, p.tilesH, p.sd, dsegs);
    if (!ok) return 1;
    p.x = a.x; p.x_bs = a.x_bs; p.y = a.y; p.y_bs = a.y_bs; p.C = a.C; p.D = a.D; p.H = a.H; p.W = a.W;
    p.fin = UpFin{a.red, 1.0 / ((double)a.D * a.H * a.W), a.slope, a.sc, a.sh, a.mean, a.rstd};
    multi_layout(m.h, i, dim3(p.tilesW * p.tilesH * dsegs, a.C, a.N));
  }
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL((upsample2x_fwd_multi_kernel<T>), dim3(m.h.off[n]), dim3(256), 0, (hipStream_t)stream, m););
  return xh_launch_status();
}

struct UpbMulti { MultiHdr h; struct P { const void* dy; long long dy_bs; void* dx; long long dx_bs; int C, D, H, W, sd, tilesW, tilesH, txn; UpRed ur; } p[XH_LEVELS_MAX]; };
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_multi_kernel(const UpbMulti m) {
  uint3 vb, vg;
  const int i = multi_find(m.h, vb, vg);
  if (i < 0) return;
  const UpbMulti::P& a = m.p[i];
#define UPB(N_) upsample2x_bwd_body<T, N_, true>(vb, (const T*)a.dy, a.dy_bs, (T*)a.dx, a.dx_bs, a.C, a.D, a.H, a.W, a.sd, a.tilesW, a.tilesH, 0, a.ur)
  switch (a.txn) { case 4: UPB(4); break; case 8: UPB(8); break; case 16: UPB(16); break; case 32: UPB(32); break; default: UPB(64); }
#undef UPB
}
extern "C" int xh_upsample2x_bwd_act_reduce_multi(void* stream, int dtype, int n, const xh_upsample2x_bwd_act_reduce_args* q) {
  if (n < 1 || n > XH_LEVELS_MAX || !q) return XH_ERR_ARG;
  if (g_xh_disable & 2) return 1;
  UpbMulti m{};
  m.h.n = n;
  const int vo = dtype == XH_F32 ? 4 : 8;
  for (int i = 0; i < n; ++i) {
    const xh_upsample2x_bwd_act_reduce_args& a = q[i];
    if (!a.dy || !a.dx || !a.y0 || !a.sc || !a.sh || !a.red || a.N <= 0 || a.C <= 0 || a.D <= 0 || a.H <= 0 || a.W <= 0) return XH_ERR_ARG;
    if (a.y0_bs % vo) return 1;
    UpbMulti::P& p = m.p[i];
    int dsegs;
    const bool ok = dtype == XH_F32 ? upsample2x_plan<float>(a.N, a.C, a.D, a.H, a.W, a.dx_bs, a.dy_bs, p.txn, p.tilesW, p.tilesH, p.sd, dsegs)
                                    : upsample2x_plan<bf16_t>(a.N, a.C, a.D, a.H, a.W, a.dx_bs, a.dy_bs, p.txn, p.tilesW, p.tilesH, p.sd, dsegs);
    if (!ok) return 1;
    p.dy = a.dy; p.dy_bs = a.dy_bs; p.dx = a.dx; p.dx_bs = a.dx_bs; p.C = a.C; p.D = a.D; p.H = a.H; p.W = a.W;
    p.ur = UpRed{a.y0, a.y0_bs, a.sc, a.sh, a.slope, a.red};
    multi_layout(m.h, i, dim3(p.tilesW * p.tilesH * dsegs, a.C, a.N));
  }
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL((upsample2x_bwd_multi_kernel<T>), dim3(m.h.off[n]), dim3(256), 0, (hipStream_t)stream, m););
  return xh_launch_status();
}
