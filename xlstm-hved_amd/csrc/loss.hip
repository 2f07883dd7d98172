// Loss / metric epilogues of the XLSTM-HVED training step, gfx950.
//
// What the reference computes with chains of full-resolution ATen ops after the generator's forward (train.py:232-262):
//   DiceLoss            loss.py:188-209,257-285    per-channel  2 sum(p t) / max(sum p^2 + sum t^2, eps)
//   MSELoss / GANLoss   train.py:173, loss.py:167-186  mean (a - b)^2, b a tensor or a constant label
//   compute_KLD         loss.py:29-40,85-115       PoE over a modality subset, then KL(posterior || prior), mean
//   nested-weight maps  train.py:242-259           w = p0>.5 ? p0 : 0, overridden by p1, p2 where those exceed .5
//   DiceCoefficient / DiceRegion   metrics.py:10-107   thresholded Dice per channel / region
// Every one of them is ONE pass over the tensors here: a reduction kernel that leaves per-(n, channel) sums in fp64, and
// for the differentiable ones a linear-combination kernel for the backward (dL/da = ca*a + cb*b + cc per (n, c)).
// Bandwidth-bound; the target tensor may be fp32 while the prediction is in the 16-bit storage type.
#include "common.h"
#include "../../include/xlstm_hved.h"

#define LS_BLOCK 256

template <bool VEC, typename T>
__device__ __forceinline__ void ld4any(const T* p, long long q, int valid, float (&o)[4]) {
  if constexpr (VEC) {                                // compile-time: straight-line loads (see ldrow in eltwise.hip)
    ld4(p, q, o);
  } else {
#pragma unroll
    for (int v = 0; v < 4; ++v) o[v] = v < valid ? ldf(p, q + v) : 0.f;
  }
}

// red[n][c][0..5] += ( sum a'b, sum a'^2, sum b^2, sum (a'-b)^2, sum a', sum b ),  a' = a or (a > thr) when thr_on
template <typename TA, typename TB, bool VEC>
__global__ __launch_bounds__(LS_BLOCK) void pair_sums_kernel(const TA* a, long long a_bs, const TB* b, long long b_bs, float bval,
                                                            long long dhw, int thr_on, float thr, double* red) {
  __shared__ double s_red[4 * 6];
  const int c = blockIdx.y, n = blockIdx.z;
  const TA* ap = a + n * a_bs + (long long)c * dhw;
  const TB* bp = b ? b + n * b_bs + (long long)c * dhw : nullptr;
  double s[6] = {0, 0, 0, 0, 0, 0};
  const long long per = ((dhw + gridDim.x - 1) / gridDim.x + LS_BLOCK * 4 - 1) / (LS_BLOCK * 4) * (LS_BLOCK * 4);
  const long long q_end = min(dhw, (long long)(blockIdx.x + 1) * per);
  long long q = (long long)blockIdx.x * per + threadIdx.x * 4;
  if constexpr (VEC) {
    // four runs per trip, the eight loads issued together (one 8-byte load pair per trip left the three 128^3 reductions of the
    // training step at 1.5 TB/s); the runs are summed in the same order as one by one
    constexpr long long S = (long long)LS_BLOCK * 4;
    for (; q + 3 * S < q_end; q += 4 * S) {
      float a4[4][4], b4[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld4(ap, q + u * S, a4[u]);
      if (bp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ld4(bp, q + u * S, b4[u]);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) b4[u][i] = bval;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x = thr_on ? (a4[u][i] > thr ? 1.f : 0.f) : a4[u][i];
          const float y = b4[u][i], d = x - y;
          t[0] = fmaf(x, y, t[0]); t[1] = fmaf(x, x, t[1]); t[2] = fmaf(y, y, t[2]); t[3] = fmaf(d, d, t[3]); t[4] += x; t[5] += y;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) s[k] += (double)t[k];
      }
    }
  }
  for (; q < q_end; q += LS_BLOCK * 4) {
    const int valid = (int)min(4LL, dhw - q);
    float av[4], bv[4] = {bval, bval, bval, bval};
    ld4any<VEC>(ap, q, valid, av);
    if (bp) ld4any<VEC>(bp, q, valid, bv);
    float t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= valid) break;
      const float x = thr_on ? (av[i] > thr ? 1.f : 0.f) : av[i];
      const float y = bv[i], d = x - y;
      t[0] = fmaf(x, y, t[0]); t[1] = fmaf(x, x, t[1]); t[2] = fmaf(y, y, t[2]); t[3] = fmaf(d, d, t[3]); t[4] += x; t[5] += y;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] += (double)t[k];
  }
  block_sum_d<6>(s, s_red, LS_BLOCK >> 6);
  if (threadIdx.x < 6) atomicAdd(&red[((long long)n * gridDim.y + c) * 6 + threadIdx.x], s_red[threadIdx.x]);
}

// out[n,c,:] (+)= ca[n,c]*a + cb[n,c]*b + cc[n,c]     (b NULL: the constant bval)
template <typename TA, typename TB, bool VEC>
__global__ __launch_bounds__(LS_BLOCK) void lincomb_kernel(const TA* a, long long a_bs, const TB* b, long long b_bs, float bval,
                                                          TA* out, long long o_bs, long long dhw, const float* ca, const float* cb,
                                                          const float* cc, const float* gs, int accumulate) {
  const int c = blockIdx.y, n = blockIdx.z, C = gridDim.y;
  const TA* ap = a + n * a_bs + (long long)c * dhw;
  const TB* bp = b ? b + n * b_bs + (long long)c * dhw : nullptr;
  TA* op = out + n * o_bs + (long long)c * dhw;
  const float g = gs ? gs[0] : 1.f;                    // upstream scalar gradient (device resident: no host sync)
  const float fa = g * ca[n * C + c], fb = g * cb[n * C + c], fc = cc ? g * cc[n * C + c] : 0.f;
  const long long per = ((dhw + gridDim.x - 1) / gridDim.x + LS_BLOCK * 4 - 1) / (LS_BLOCK * 4) * (LS_BLOCK * 4);
  const long long q_end = min(dhw, (long long)(blockIdx.x + 1) * per);
  for (long long q = (long long)blockIdx.x * per + threadIdx.x * 4; q < q_end; q += LS_BLOCK * 4) {
    const int valid = (int)min(4LL, dhw - q);
    float av[4], bv[4] = {bval, bval, bval, bval}, ov[4] = {0, 0, 0, 0};
    ld4any<VEC>(ap, q, valid, av);
    if (bp) ld4any<VEC>(bp, q, valid, bv);
    if (accumulate) ld4any<VEC>((const TA*)op, q, valid, ov);
#pragma unroll
    for (int i = 0; i < 4; ++i) ov[i] += fmaf(fa, av[i], fmaf(fb, bv[i], fc));
    if constexpr (VEC) {
      st4(op, q, ov);
    } else {
      for (int i = 0; i < valid; ++i) stf(op, q + i, ov[i]);
    }
  }
}

// launches KERNEL<TA, TB, vec> on `grid` / `st` (both in scope) with the layout flag as a compile-time argument
#define LS_VEC(KERNEL, TA, TB, ...)                                                                      \
  do {                                                                                                   \
    if (vec) hipLaunchKernelGGL((KERNEL<TA, TB, true>), grid, dim3(LS_BLOCK), 0, st, __VA_ARGS__);       \
    else hipLaunchKernelGGL((KERNEL<TA, TB, false>), grid, dim3(LS_BLOCK), 0, st, __VA_ARGS__);          \
  } while (0)
static inline dim3 ls_grid(long long dhw, int C, int N) {
  const long long maxb = (dhw + LS_BLOCK * 4 - 1) / (LS_BLOCK * 4);
  long long want = (2048 + (long long)C * N - 1) / ((long long)C * N);
  if (want < 1) want = 1;
  return dim3((unsigned)(want < maxb ? want : maxb), C, N);
}

// The reducing pass ends in six fp64 atomics per workgroup on the row's six addresses, and atomics on one address retire one after
// the other (~45 ns): the 683 workgroups per row of a 3-channel 128^3 loss spent 30 us queueing them for 38 MB of input.  At most 64
// workgroups per row (like the norm reducers of eltwise.hip); a thread's longer walk runs four runs per trip.
static inline dim3 ls_red_grid(long long dhw, int C, int N) {
  dim3 g = ls_grid(dhw, C, N);
  if (g.x > 64) g.x = 64;
  return g;
}
extern "C" int xh_pair_sums(void* stream, int dtype, const void* a, long long a_bs, int b_dtype, const void* b, long long b_bs,
                            float bval, int N, int C, long long DHW, int thr_on, float thr, double* red) {
  if (!a || !red || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  if (b && b_dtype != dtype && b_dtype != XH_F32) return XH_ERR_DTYPE;
  const bool vec = DHW % 4 == 0 && a_bs % 4 == 0 && (!b || b_bs % 4 == 0);
  const dim3 grid = ls_red_grid(DHW, C, N);
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype,
    if (b && b_dtype == XH_F32 && dtype != XH_F32)
      LS_VEC(pair_sums_kernel, T, float, (const T*)a, a_bs, (const float*)b, b_bs, bval, DHW, thr_on, thr, red);
    else
      LS_VEC(pair_sums_kernel, T, T, (const T*)a, a_bs, (const T*)b, b_bs, bval, DHW, thr_on, thr, red););
  return xh_launch_status();
}

extern "C" int xh_lincomb(void* stream, int dtype, const void* a, long long a_bs, int b_dtype, const void* b, long long b_bs,
                          float bval, void* out, long long o_bs, int N, int C, long long DHW, const float* ca, const float* cb,
                          const float* cc, const float* gscale, int accumulate) {
  if (!a || !out || !ca || !cb || N <= 0 || C <= 0 || DHW <= 0 || C > 65535 || N > 65535) return XH_ERR_ARG;
  if (b && b_dtype != dtype && b_dtype != XH_F32) return XH_ERR_DTYPE;
  const bool vec = DHW % 4 == 0 && a_bs % 4 == 0 && o_bs % 4 == 0 && (!b || b_bs % 4 == 0);
  const dim3 grid = ls_grid(DHW, C, N);
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype,
    if (b && b_dtype == XH_F32 && dtype != XH_F32)
      LS_VEC(lincomb_kernel, T, float, (const T*)a, a_bs, (const float*)b, b_bs, bval, (T*)out, o_bs, DHW, ca, cb, cc, gscale, accumulate);
    else
      LS_VEC(lincomb_kernel, T, T, (const T*)a, a_bs, (const T*)b, b_bs, bval, (T*)out, o_bs, DHW, ca, cb, cc, gscale, accumulate););
  return xh_launch_status();
}

// ------------------------------------------------------------------------------------------------ KLD
// stacks: [N][5][L][dhw], index 0 = prior.  keep[n][m] selects the experts of the subset (the prior always takes part).
// Per latent voxel: T_i = 1/(e^{lv_i} + 1e-8), P = sum T_i, m = sum mu_i T_i / P, var1 = 1/P;
// term = -1 + lv_prior + log P + (1/P + (m - mu_prior)^2) / (e^{lv_prior} + 1e-8)        (loss.py:29-40,52-64)
// fwd: red[0] += sum term.   bwd: dmu_i / dlv_i of  scale * sum term  for the prior and the kept experts, zeros elsewhere.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void kld_kernel(const T* mu, const T* lv, const float* keep, int L, long long dhw, long long total,
                                                  double* red, float scale_h, const float* gs, T* dmu, T* dlv) {
  __shared__ double s_red[4];
  double acc[1] = {0.0};
  // forward: a grid-stride walk on at most 256 workgroups (xh_kld_fwd; 128: 14.1 us, 1 024: 15.4 us for the largest level) -- every workgroup ends in one fp64 atomic on ONE address, and
  // the 1 024 workgroups of the largest level queued them for 15 us; backward: one element per thread, no reduction
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long ldhw = (long long)L * dhw;
    const int n = (int)(i / ldhw);
    const long long r = i % ldhw;
    const long long base = (long long)n * 5 * ldhw + r;
    float m_[5], l_[5], t_[5], kp[5];
    kp[0] = 1.f;
#pragma unroll
    for (int e = 1; e < 5; ++e) kp[e] = keep[n * 4 + e - 1];
    float P = 0.f, num = 0.f;
#pragma unroll
    for (int e = 0; e < 5; ++e) {
      m_[e] = ldf(mu, base + e * ldhw);
      l_[e] = ldf(lv, base + e * ldhw);
      t_[e] = kp[e] / (expf(l_[e]) + 1e-8f);
      P += t_[e];
      num = fmaf(m_[e], t_[e], num);
    }
    const float m = num / P;
    const float v2e = expf(l_[0]) + 1e-8f;
    const float d = m - m_[0];
    if (!BWD) {
      acc[0] += (double)(-1.f + l_[0] + logf(P) + (1.f / P + d * d) / v2e);
    } else {
      const float scale = scale_h * (gs ? gs[0] : 1.f);
      const float gm = 2.f * d / v2e * scale;
      const float gP = (1.f / P - 1.f / (P * P * v2e)) * scale;
#pragma unroll
      for (int e = 0; e < 5; ++e) {
        float gmu = 0.f, glv = 0.f;
        if (kp[e] != 0.f) {                              // as an expert of the product (the prior always is one)
          gmu = gm * t_[e] / P;
          glv = (gm * (m_[e] - m) / P + gP) * (-expf(l_[e]) * t_[e] * t_[e]);
        }
        if (e == 0) {                                    // ... and the prior is also the KL's second distribution
          gmu -= gm;
          glv += scale * (1.f - (1.f / P + d * d) * expf(l_[0]) / (v2e * v2e));
        }
        stf(dmu, base + e * ldhw, gmu);
        stf(dlv, base + e * ldhw, glv);
      }
    }
  }
  if (!BWD) {
    block_sum_d<1>(acc, s_red, 4);
    if (threadIdx.x == 0) atomicAdd(red, s_red[0]);
  }
}

extern "C" int xh_kld_fwd(void* stream, int dtype, const void* mu_stack, const void* lv_stack, const float* keep, int N, int L,
                          long long dhw, double* red) {
  if (!mu_stack || !lv_stack || !keep || !red || N <= 0 || L <= 0 || dhw <= 0) return XH_ERR_ARG;
  const long long total = (long long)N * L * dhw;
  unsigned nb = (unsigned)((total + 255) / 256);
  if (nb > 256) nb = 256;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL((kld_kernel<T, false>), dim3(nb), dim3(256), 0, st, (const T*)mu_stack, (const T*)lv_stack, keep,
                                          L, dhw, total, red, 0.f, (const float*)nullptr, (T*)nullptr, (T*)nullptr););
  return xh_launch_status();
}
extern "C" int xh_kld_bwd(void* stream, int dtype, const void* mu_stack, const void* lv_stack, const float* keep, int N, int L,
                          long long dhw, float scale, const float* gscale, void* dmu_stack, void* dlv_stack) {
  if (!mu_stack || !lv_stack || !keep || !dmu_stack || !dlv_stack || N <= 0 || L <= 0 || dhw <= 0) return XH_ERR_ARG;
  const long long total = (long long)N * L * dhw;
  const unsigned nb = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL((kld_kernel<T, true>), dim3(nb), dim3(256), 0, st, (const T*)mu_stack, (const T*)lv_stack, keep,
                                          L, dhw, total, (double*)nullptr, scale, gscale, (T*)dmu_stack, (T*)dlv_stack););
  return xh_launch_status();
}

// ------------------------------------------------------------------------------------------------ nested weights
// train.py:244-248: w = where(p > .5, p, 0) per channel, then w0 overridden by w1 where p1 > .5, then by w2 where p2 > .5
template <typename T>
__global__ __launch_bounds__(256) void nested_weight_kernel(const T* seg, long long s_bs, T* w, long long w_bs, long long dhw, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i / dhw);
  const long long q = i % dhw;
  const T* sp = seg + n * s_bs + q;
  const float p0 = ldf(sp, 0), p1 = ldf(sp, dhw), p2 = ldf(sp, 2 * dhw);
  float v = p0 > 0.5f ? p0 : 0.f;
  if (p1 > 0.5f) v = p1;
  if (p2 > 0.5f) v = p2;
  stf(w + n * w_bs, q, v);
}
extern "C" int xh_nested_weight(void* stream, int dtype, const void* seg, long long seg_bs, void* w, long long w_bs, int N, long long DHW) {
  if (!seg || !w || N <= 0 || DHW <= 0) return XH_ERR_ARG;
  const long long total = (long long)N * DHW;
  const unsigned nb = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL(nested_weight_kernel<T>, dim3(nb), dim3(256), 0, st, (const T*)seg, seg_bs, (T*)w, w_bs, DHW, total););
  return xh_launch_status();
}

// ------------------------------------------------------------------------------------------------ finalisation
// One workgroup turns the fp64 sums into the scalar loss / per-channel metric and the per-(n,c) backward coefficients,
// so the host never touches them (no .item(), no chain of tiny ATen ops).
//  kind 0  DiceLoss (loss.py:188-209,257-285): dice_c = 2 I_c / max(A_c + B_c, eps) over batch and space, loss = 1 - mean_c;
//          d loss / d a = ca*a + cb*b with ca = 4 I/(C D^2), cb = -2/(C D)   (clamped: ca = 0, cb = -2/(C eps))
//  kind 1  mean squared difference (nn.MSELoss, GANLoss): loss = sum d^2 / count; ca = 2/count, cb = -2/count
//  kind 2  thresholded Dice metric (metrics.py:40-48,99-107): out[c] = mean_n (2 I + eps) / (sum a' + sum b + eps)
//  kind 3  mean of a (the SURVEY 8(d) benchmark loss): out[0] = sum a / count
//  kind 4  sum_i w[i] * (sum of tensor i), w = ca (an INPUT here): several means in one finalisation
__global__ __launch_bounds__(64) void loss_finalize_kernel(int kind, const double* red, int N, int C, double count, double eps,
                                                          float* out, float* ca, float* cb) {
  const int t = threadIdx.x;
  if (kind == 0) {
    __shared__ double s_d[64];
    double dice = 0.0;
    if (t < C) {
      double I = 0, D = 0;
      for (int n = 0; n < N; ++n) { const double* r = red + ((long long)n * C + t) * 6; I += r[0]; D += r[1] + r[2]; }
      const bool cl = D < eps;
      const double Dc = cl ? eps : D;
      dice = 2.0 * I / Dc;
      for (int n = 0; n < N; ++n) {
        ca[n * C + t] = (float)(cl ? 0.0 : 4.0 * I / (C * Dc * Dc));
        cb[n * C + t] = (float)(-2.0 / (C * Dc));
      }
    }
    s_d[t] = dice;
    __syncthreads();
    if (t == 0) { double m = 0; for (int c = 0; c < C; ++c) m += s_d[c]; out[0] = (float)(1.0 - m / C); }
  } else if (kind == 1) {
    double sacc = 0;                                  // one row per lane, then a wave sum: a serial loop of dependent loads on
    for (int i = t; i < N * C; i += 64) sacc += red[(long long)i * 6 + 3];          // lane 0 cost ~0.4 us per row
    sacc = wave_sum(sacc);
    if (t == 0) out[0] = (float)(sacc / count);
    for (int i = t; i < N * C; i += 64) { ca[i] = (float)(2.0 / count); cb[i] = (float)(-2.0 / count); }
  } else if (kind == 2) {
    if (t < C) {
      double m = 0;
      for (int n = 0; n < N; ++n) { const double* r = red + ((long long)n * C + t) * 6; m += (2.0 * r[0] + eps) / (r[4] + r[5] + eps); }
      out[t] = (float)(m / N);
    }
  } else if (kind == 3) {                             // plain mean of a (slot 4)
    double sacc = 0;
    for (int i = t; i < N * C; i += 64) sacc += red[(long long)i * 6 + 4];
    sacc = wave_sum(sacc);
    if (t == 0) out[0] = (float)(sacc / count);
  } else {                                            // kind 4: weighted sum of several tensors' sums: sum_i red[i][4] * ca[i]
    double sacc = 0;
    for (int i = t; i < N * C; i += 64) sacc += red[(long long)i * 6 + 4] * (double)ca[i];
    sacc = wave_sum(sacc);
    if (t == 0) out[0] = (float)sacc;
  }
}
extern "C" int xh_loss_finalize(void* stream, int kind, const double* red, int N, int C, double count, double eps, float* out,
                                float* ca, float* cb) {
  if (!red || !out || N <= 0 || C <= 0 || (C > 64 && (kind == 0 || kind == 2)) || kind < 0 || kind > 4) return XH_ERR_ARG;
  if ((kind < 2 && (!ca || !cb)) || (kind == 4 && !ca)) return XH_ERR_ARG;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, kind, red, N, C, count, eps, out, ca, cb);
  return xh_launch_status();
}

// fill: out[i] = v  (the constant upstream gradients of mean-type losses, in the storage type)
template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(T* out, long long n, float v, const float* gs) {
  if (gs) v *= gs[0];
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float o[4] = {v, v, v, v};
    st4(out, i, o);
  } else {
    for (long long k = i; k < n; ++k) stf(out, k, v);
  }
}
// ------------------------------------------------------------------------------------------------ multi-tensor passes
// A loss that is a sum over MANY tensors (the SURVEY 8(d) benchmark loss: seg, recon and the eight latent stacks) costs one
// reduction launch and one gradient-fill launch PER tensor when done tensor by tensor -- twenty launches of which sixteen are
// small enough to be pure launch latency (~5 us each in a replayed graph).  Here the tensors of one dtype travel in ONE
// launch: the job table rides in the kernel arguments, a workgroup finds its tensor from a prefix of workgroup counts.
struct XhMulti {
  const void* p[XH_MULTI_MAX];
  long long n[XH_MULTI_MAX];
  float v[XH_MULTI_MAX];         // multi_fill: the value;  multi_sum: unused
  int row0[XH_MULTI_MAX];        // multi_sum: first row of red[] this tensor adds to
  int rows[XH_MULTI_MAX];        //            ... and how many rows it spreads its workgroups over (<= 64 adders per row)
  int wg0[XH_MULTI_MAX + 1];     // prefix of workgroup counts
  int nt;
};
// red[(row0[t] + b % rows[t]) * 6 + 4] += sum of workgroup b's chunk of tensor t   (slot 4 = "sum a" of xh_pair_sums' layout,
// so xh_loss_finalize kind 4 serves both)
template <typename T>
__global__ __launch_bounds__(LS_BLOCK) void multi_sum_kernel(const XhMulti m, double* red) {
  __shared__ double s_red[4];
  int t = 0;
  for (int k = 1; k < m.nt; ++k)
    if ((int)blockIdx.x >= m.wg0[k]) t = k;
  const int b = blockIdx.x - m.wg0[t], nb = m.wg0[t + 1] - m.wg0[t];
  const T* p = (const T*)m.p[t];
  const long long n = m.n[t];
  const long long per = ((n + nb - 1) / nb + LS_BLOCK * 4 - 1) / (LS_BLOCK * 4) * (LS_BLOCK * 4);
  const long long q_end = min(n, (long long)(b + 1) * per);
  const bool vec = (n % 4 == 0) && (((unsigned long long)p & (4 * sizeof(T) - 1)) == 0);
  double s[1] = {0.0};
  long long q = (long long)b * per + threadIdx.x * 4;
  if (vec) {                                            // four runs per trip, their loads issued together (the order of the sums stays)
    constexpr long long S = (long long)LS_BLOCK * 4;
    for (; q + 3 * S < q_end; q += 4 * S) {
      float a4[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld4any<true>(p, q + u * S, 4, a4[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) s[0] += (double)(((0.f + a4[u][0]) + a4[u][1]) + a4[u][2] + a4[u][3]);
    }
  }
  for (; q < q_end; q += LS_BLOCK * 4) {
    const int valid = (int)min(4LL, n - q);
    float av[4];
    if (vec) ld4any<true>(p, q, valid, av); else ld4any<false>(p, q, valid, av);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < valid) acc += av[i];
    s[0] += (double)acc;
  }
  block_sum_d<1>(s, s_red, LS_BLOCK >> 6);
  if (threadIdx.x == 0) atomicAdd(&red[(long long)(m.row0[t] + b % m.rows[t]) * 6 + 4], s_red[0]);
}
template <typename T>
__global__ __launch_bounds__(256) void multi_fill_kernel(const XhMulti m, const float* gs) {
  int t = 0;
  for (int k = 1; k < m.nt; ++k)
    if ((int)blockIdx.x >= m.wg0[k]) t = k;
  const int b = blockIdx.x - m.wg0[t], nb = m.wg0[t + 1] - m.wg0[t];
  T* out = (T*)const_cast<void*>(m.p[t]);
  const long long n = m.n[t];
  float v = m.v[t];
  if (gs) v *= gs[0];
  const bool vec = ((unsigned long long)out & (4 * sizeof(T) - 1)) == 0;
  const float o[4] = {v, v, v, v};
  for (long long i = ((long long)b * 256 + threadIdx.x) * 4; i < n; i += (long long)nb * 1024) {
    if (vec && i + 3 < n) st4(out, i, o);
    else
      for (long long k = i; k < n && k < i + 4; ++k) stf(out, k, v);
  }
}
static int multi_plan(XhMulti* m, int nt, const void* const* ptrs, const long long* numels, const float* values, const int* row0,
                      const int* rows, long long per_wg) {
  if (nt <= 0 || nt > XH_MULTI_MAX || !ptrs || !numels) return -1;
  m->nt = nt;
  int tot = 0;
  for (int t = 0; t < nt; ++t) {
    if (!ptrs[t] || numels[t] <= 0) return -1;
    m->p[t] = ptrs[t]; m->n[t] = numels[t];
    m->v[t] = values ? values[t] : 0.f;
    m->row0[t] = row0 ? row0[t] : 0;
    m->rows[t] = rows ? (rows[t] > 0 ? rows[t] : 1) : 1;
    long long nb = (numels[t] + per_wg - 1) / per_wg;
    if (rows && nb > 64LL * m->rows[t]) nb = 64LL * m->rows[t];      // at most 64 adders per atomic address
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    m->wg0[t] = tot;
    tot += (int)nb;
  }
  m->wg0[nt] = tot;
  return tot;
}
extern "C" int xh_multi_sum(void* stream, int dtype, int nt, const void* const* ptrs, const long long* numels, const int* row0, const int* rows,
                            double* red) {
  XhMulti m;
  if (!red || !row0 || !rows) return XH_ERR_ARG;
  const int tot = multi_plan(&m, nt, ptrs, numels, nullptr, row0, rows, 16384);
  if (tot <= 0) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL(multi_sum_kernel<T>, dim3((unsigned)tot), dim3(LS_BLOCK), 0, st, m, red););
  return xh_launch_status();
}
extern "C" int xh_multi_fill(void* stream, int dtype, int nt, void* const* ptrs, const long long* numels, const float* values,
                             const float* gscale) {
  XhMulti m;
  if (!values) return XH_ERR_ARG;
  const int tot = multi_plan(&m, nt, (const void* const*)ptrs, numels, values, nullptr, nullptr, 16384);
  if (tot <= 0) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL(multi_fill_kernel<T>, dim3((unsigned)tot), dim3(256), 0, st, m, gscale););
  return xh_launch_status();
}

extern "C" int xh_fill(void* stream, int dtype, void* out, long long n, float v, const float* gscale) {
  if (!out || n <= 0) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL(fill_kernel<T>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, (T*)out, n, v, gscale););
  return xh_launch_status();
}

// ------------------------------------------------------------------------------------------------ scalar glue of a loss
// train.py:240,262,280 add five loss terms with three weights, average four KL terms, halve a sum of two: as ATen ops on device
// scalars that is one ~5 us launch per operation and as many again in backward (60 of the training step's 82 ATen launches).
// One launch forms the weighted sum of up to XH_SCALAR_MAX device scalars (fp32 or fp64), one launch fans the upstream gradient
// out to them.
struct XhScalars {
  const void* p[XH_SCALAR_MAX];
  double c[XH_SCALAR_MAX];
  int f64[XH_SCALAR_MAX];
  int n;
};
__global__ __launch_bounds__(64) void scalar_lincomb_kernel(const XhScalars s, float* out) {
  const int i = threadIdx.x;
  double v = 0.0;
  if (i < s.n) v = s.c[i] * (s.f64[i] ? *reinterpret_cast<const double*>(s.p[i]) : (double)*reinterpret_cast<const float*>(s.p[i]));
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);            // XH_SCALAR_MAX = 16 lanes carry terms
  if (i == 0) out[0] = (float)v;
}
__global__ __launch_bounds__(64) void scalar_fanout_kernel(const XhScalars s, const float* g, float* o32, double* o64) {
  const int i = threadIdx.x;
  if (i < s.n) {
    const double v = s.c[i] * (double)g[0];
    o32[i] = (float)v;
    o64[i] = v;
  }
}
extern "C" int xh_scalar_lincomb(void* stream, int n, const void* const* src, const int* is_f64, const double* coef, float* out) {
  if (n <= 0 || n > XH_SCALAR_MAX || !src || !is_f64 || !coef || !out) return XH_ERR_ARG;
  XhScalars s;
  s.n = n;
  for (int i = 0; i < XH_SCALAR_MAX; ++i) {
    s.p[i] = i < n ? src[i] : nullptr; s.c[i] = i < n ? coef[i] : 0.0; s.f64[i] = i < n ? is_f64[i] : 0;
    if (i < n && !src[i]) return XH_ERR_ARG;
  }
  hipLaunchKernelGGL(scalar_lincomb_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, s, out);
  return xh_launch_status();
}
extern "C" int xh_scalar_fanout(void* stream, int n, const double* coef, const float* g, float* out32, double* out64) {
  if (n <= 0 || n > XH_SCALAR_MAX || !coef || !g || !out32 || !out64) return XH_ERR_ARG;
  XhScalars s;
  s.n = n;
  for (int i = 0; i < XH_SCALAR_MAX; ++i) { s.p[i] = nullptr; s.c[i] = i < n ? coef[i] : 0.0; s.f64[i] = 0; }
  hipLaunchKernelGGL(scalar_fanout_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, s, g, out32, out64);
  return xh_launch_status();
}
