// 3x3x3 stride-1 convolution between ONE and TWO channels (forward 1 -> 2 with the sigmoid, data gradient 2 -> 1, weight
// gradient): the spatial branch of DuSEAttention (buildingblocks.py:313-327: conv 2C -> 1, conv3 1 -> 2, sigmoid), three levels
// per step.  113 MFLOP on 12 MB at 128^3 -- the generic LDS-tiled kernels took 27 / 34 / 39 us for the three directions there
// (halo tile + barrier per tile, a run-time switched epilogue); this is a streaming stencil:
//  * a lane owns one 16-byte run of an output row (8 voxels at 16-bit storage, 4 at fp32), the lanes of a row sit next to each
//    other in a wave: the w - 1 / w + 1 neighbours come by wave shuffle, every global access is a whole aligned run;
//  * the 9 input rows of a channel are loaded from clamped (always valid) addresses, all of them before the first is used,
//    and masked arithmetically (zero padding);
//  * weights, bias, activation and the channel counts are compile-time or scalar: straight-line code, no LDS, no barrier;
//  * the weight gradient keeps its 54 + 2 sums in registers across a grid-stride loop over rows and ends with one block
//    reduction and one pass of fp32 atomics per workgroup.
#include <algorithm>
#include <vector>
#include "common.h"
#include "../../include/xlstm_hved.h"

namespace {
struct TinyK {
  const void* x; const void* dy; void* y;
  const float* w; const float* b;
  float* dw; float* db;
  long long x_bs, dy_bs, y_bs;
  int D, H, W, LW, transposed;
};

// r[1 .. VW] = the run, r[0] / r[VW + 1] = its neighbours in the row (0 outside the volume); m = 0 for a padding row
template <typename T, int VW>
__device__ __forceinline__ void tiny_row(const T* src, long long off, float m, bool first, bool last, float (&r)[VW + 2]) {
  float x[VW];
  ldvec(src, off, x);
#pragma unroll
  for (int v = 0; v < VW; ++v) r[v + 1] = x[v] * m;
  const float l = __shfl_up(r[VW], 1, 64), rr = __shfl_down(r[1], 1, 64);
  r[0] = first ? 0.f : l;
  r[VW + 1] = last ? 0.f : rr;
}

template <typename T, int CI, int CO, bool SIG>
__global__ __launch_bounds__(256) void conv3_tiny_kernel(const TinyK a) {
  constexpr int VW = VWT<T>::v;
  const int n = blockIdx.y;
  const int D = a.D, H = a.H, W = a.W, LW = a.LW;
  const long long dhw = (long long)D * H * W;
  float wgt[CO][CI][27], bias[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    bias[co] = a.b ? a.b[co] : 0.f;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int t = 0; t < 27; ++t) wgt[co][ci][t] = a.transposed ? a.w[(ci * CO + co) * 27 + 26 - t] : a.w[(co * CI + ci) * 27 + t];
  }
  const long long lane_id = (long long)blockIdx.x * 256 + threadIdx.x;
  const int tx = (int)(lane_id % LW);
  const long long row = lane_id / LW;
  const bool ok = row < (long long)D * H;
  const int oh = (int)(row % H), od = (int)min(row / H, (long long)D - 1);
  const int ow = tx * VW;
  float acc[CO][VW];
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int v = 0; v < VW; ++v) acc[co][v] = bias[co];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) {
    const T* src = (const T*)a.x + n * a.x_bs + (long long)ci * dhw + ow;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int gd = od - 1 + kd, gh = oh - 1 + kh;
        const float m = ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H) ? 1.f : 0.f;
        float r[VW + 2];
        tiny_row<T, VW>(src, ((long long)min(max(gd, 0), D - 1) * H + min(max(gh, 0), H - 1)) * W, m, tx == 0, tx == LW - 1, r);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int co = 0; co < CO; ++co) {
            const float wv = wgt[co][ci][(kd * 3 + kh) * 3 + kw];
#pragma unroll
            for (int v = 0; v < VW; ++v) acc[co][v] = fmaf(wv, r[v + kw], acc[co][v]);
          }
      }
  }
  if (!ok) return;
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    if (SIG) {
#pragma unroll
      for (int v = 0; v < VW; ++v) acc[co][v] = sigmoidf_(acc[co][v]);
    }
    stvec((T*)a.y + n * a.y_bs + (long long)co * dhw, ((long long)od * H + oh) * W + ow, acc[co]);
  }
}

template <typename T, int CI, int CO>
__device__ __forceinline__ void tiny_wgrad_body(const TinyK& a, const int bx, const int n, const int gdx) {
  constexpr int VW = VWT<T>::v, NACC = 27 * CI * CO + CO;
  __shared__ float s_red[4 * NACC];
  const int D = a.D, H = a.H, W = a.W, LW = a.LW;
  const long long dhw = (long long)D * H * W, rows = (long long)D * H, lanes = rows * LW;
  float acc[CO][CI][27], dbs[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    dbs[co] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int t = 0; t < 27; ++t) acc[co][ci][t] = 0.f;
  }
  // whole workgroups iterate together (the shuffles need every lane of a row)
  for (long long lane_id = (long long)bx * 256 + threadIdx.x; lane_id - threadIdx.x < lanes; lane_id += (long long)gdx * 256) {
    const int tx = (int)(lane_id % LW);
    const long long row = lane_id / LW;
    const float okm = row < rows ? 1.f : 0.f;
    const int oh = (int)(row % H), od = (int)min(row / H, (long long)D - 1);
    const int ow = tx * VW;
    float dyv[CO][VW];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      ldvec((const T*)a.dy + n * a.dy_bs + (long long)co * dhw, ((long long)od * H + oh) * W + ow, dyv[co]);
#pragma unroll
      for (int v = 0; v < VW; ++v) { dyv[co][v] *= okm; dbs[co] += dyv[co][v]; }
    }
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      const T* src = (const T*)a.x + n * a.x_bs + (long long)ci * dhw + ow;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int gd = od - 1 + kd, gh = oh - 1 + kh;
          const float m = ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H) ? 1.f : 0.f;
          float r[VW + 2];
          tiny_row<T, VW>(src, ((long long)min(max(gd, 0), D - 1) * H + min(max(gh, 0), H - 1)) * W, m, tx == 0, tx == LW - 1, r);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int co = 0; co < CO; ++co) {
              float t = acc[co][ci][(kd * 3 + kh) * 3 + kw];
#pragma unroll
              for (int v = 0; v < VW; ++v) t = fmaf(r[v + kw], dyv[co][v], t);
              acc[co][ci][(kd * 3 + kh) * 3 + kw] = t;
            }
        }
    }
  }
  float v[NACC];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    v[27 * CI * CO + co] = dbs[co];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int t = 0; t < 27; ++t) v[(co * CI + ci) * 27 + t] = acc[co][ci][t];
  }
  block_sum<NACC>(v, s_red, 4);
  const int tid = threadIdx.x;
  if (tid < 27 * CI * CO) atomicAdd(&a.dw[tid], s_red[tid]);                  // dw is [CO][CI][27]
  else if (tid < NACC && a.db) atomicAdd(&a.db[tid - 27 * CI * CO], s_red[tid]);
}
template <typename T, int CI, int CO>
__global__ __launch_bounds__(256) void conv3_tiny_wgrad_kernel(const TinyK a) {
  tiny_wgrad_body<T, CI, CO>(a, blockIdx.x, blockIdx.y, gridDim.x);
}
// the problems of a batch (xh_conv3d_wgrad_batch: the DuSE adjust convs of a step) in one launch; workgroup b belongs to problem
// i with off[i] <= b < off[i + 1]
constexpr int TINY_MULTI = 8;
struct TinyMulti {
  int n;
  int off[TINY_MULTI + 1];
  int gx[TINY_MULTI];
  TinyK p[TINY_MULTI];
};
template <typename T, int CI, int CO>
__global__ __launch_bounds__(256) void conv3_tiny_wgrad_multi_kernel(const TinyMulti m) {
  int pi = 0;
  for (int k = 1; k < TINY_MULTI; ++k)
    if (k < m.n && (int)blockIdx.x >= m.off[k]) pi = k;
  const int local = blockIdx.x - m.off[pi], gx = m.gx[pi];
  tiny_wgrad_body<T, CI, CO>(m.p[pi], local % gx, local / gx, gx);
}

template <typename T>
bool tiny_ok(const xh_conv_desc* d, const xh_conv_ptrs* p, int& lw) {
  constexpr int VW = VWT<T>::v;
  if (d->k != 3 || d->stride != 1 || d->groups != 1 || d->n_wptr != 1 || d->pre || d->epi) return false;
  if (!((d->Cin == 1 && d->Cout == 2) || (d->Cin == 2 && d->Cout == 1))) return false;
  if (d->Ca != d->Cin || d->Do != d->D || d->Ho != d->H || d->Wo != d->W || d->N > 65535) return false;
  if (d->W % VW) return false;
  lw = d->W / VW;
  if (lw > 64 || 64 % lw) return false;                                       // the lanes of a row share a wave
  const long long dhw = (long long)d->D * d->H * d->W;
  if (dhw % VW || d->xa_bs % VW) return false;
  (void)p;
  return true;
}
}  // namespace

void xh_note_kernel(const char* fmt, ...);
int g_tiny_wgs = 512;   // xh_set_option(16, n): workgroups of the weight-gradient kernel (each ends with 56 same-address atomics)

// XH_OK if launched, 1 if the shape is not for these kernels
int xh_conv3_tiny_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p) {
  if (d->act != XH_ACT_NONE && d->act != XH_ACT_SIGMOID) return 1;
  int lw = 0;
  bool ok = false;
  XH_DISPATCH_T(d->dtype, ok = tiny_ok<T>(d, p, lw) && d->y_bs % VWT<T>::v == 0;);
  if (!ok) return 1;
  TinyK a;
  a.x = p->xa; a.dy = nullptr; a.y = p->y; a.w = p->w[0]; a.b = p->b[0]; a.dw = a.db = nullptr;
  a.x_bs = d->xa_bs; a.dy_bs = 0; a.y_bs = d->y_bs;
  a.D = d->D; a.H = d->H; a.W = d->W; a.LW = lw; a.transposed = d->transposed;
  const long long lanes = (long long)d->D * d->H * lw;
  dim3 grid((unsigned)((lanes + 255) / 256), d->N);
  hipStream_t st = (hipStream_t)stream;
  const bool sig = d->act == XH_ACT_SIGMOID;
  xh_note_kernel("conv3_tiny_kernel<%d -> %d%s>", d->Cin, d->Cout, sig ? ", sigmoid" : "");
#define TINY(CI, CO)                                                                                     \
  do {                                                                                                   \
    if (sig) hipLaunchKernelGGL((conv3_tiny_kernel<T, CI, CO, true>), grid, dim3(256), 0, st, a);         \
    else hipLaunchKernelGGL((conv3_tiny_kernel<T, CI, CO, false>), grid, dim3(256), 0, st, a);            \
  } while (0)
  XH_DISPATCH_T(d->dtype, { if (d->Cin == 1) TINY(1, 2); else TINY(2, 1); });
#undef TINY
  return xh_launch_status();
}

// weight / bias gradient of the same convs (dw: [Cout][Cin][27] fp32, accumulated); XH_OK if launched, 1 if not eligible
static int tiny_wgrad_plan(const xh_conv_desc* d, const xh_conv_ptrs* p, float* dw, float* db, TinyK* a_, int* gx) {
  if (d->transposed || !p->ea || !dw) return 1;
  int lw = 0;
  bool ok = false;
  XH_DISPATCH_T(d->dtype, ok = tiny_ok<T>(d, p, lw) && d->ea_bs % VWT<T>::v == 0;);
  if (!ok) return 1;
  TinyK& a = *a_;
  a.x = p->xa; a.dy = p->ea; a.y = nullptr; a.w = nullptr; a.b = nullptr; a.dw = dw; a.db = db;
  a.x_bs = d->xa_bs; a.dy_bs = d->ea_bs; a.y_bs = 0;
  a.D = d->D; a.H = d->H; a.W = d->W; a.LW = lw; a.transposed = 0;
  const long long lanes = (long long)d->D * d->H * lw;
  long long nb = (lanes + 255) / 256;
  if (nb > g_tiny_wgs) nb = g_tiny_wgs;
  *gx = (int)nb;
  return XH_OK;
}
int xh_conv3_tiny_wgrad_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* dw, float* db) {
  TinyK a;
  int nb = 0;
  const int rcp = tiny_wgrad_plan(d, p, dw, db, &a, &nb);
  if (rcp != XH_OK) return rcp;
  dim3 grid((unsigned)nb, d->N);
  hipStream_t st = (hipStream_t)stream;
  xh_note_kernel("conv3_tiny_wgrad_kernel<%d -> %d>", d->Cin, d->Cout);
  XH_DISPATCH_T(d->dtype, {
    if (d->Cin == 1) hipLaunchKernelGGL((conv3_tiny_wgrad_kernel<T, 1, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3_tiny_wgrad_kernel<T, 2, 1>), grid, dim3(256), 0, st, a);
  });
  return xh_launch_status();
}

// The problems of a batch this file's weight-gradient kernel takes, TINY_MULTI per launch, storage type and channel shape
// (marked in handled[]; a lone problem is left to the ordinary entry point)
int xh_tiny_wgrad_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, float* const (*dw)[XH_MAX_WPTR],
                        float* const (*db)[XH_MAX_WPTR], char* handled) {
  extern int g_xh_disable;
  int xh_check_conv(const xh_conv_desc* d, const xh_conv_ptrs* p);
  if (g_xh_disable & 512) return XH_OK;
  hipStream_t st = (hipStream_t)stream;
  int rc_all = XH_OK;
  const int types[3] = {XH_F32, XH_BF16, XH_F16};
  TinyMulti* m = new TinyMulti;
  for (int t = 0; t < 3; ++t)
    for (int cin = 1; cin <= 2; ++cin) {
      std::vector<int> idx;
      std::vector<TinyK> ks;
      std::vector<int> gxs;
      for (int i = 0; i < n; ++i) {
        if (handled[i] || !d[i] || !p[i] || !dw[i] || d[i]->dtype != types[t] || d[i]->Cin != cin) continue;
        if (xh_check_conv(d[i], p[i])) continue;
        TinyK a; int gx = 0;
        if (tiny_wgrad_plan(d[i], p[i], dw[i][0], db ? db[i][0] : nullptr, &a, &gx) != XH_OK) continue;
        idx.push_back(i); ks.push_back(a); gxs.push_back(gx);
      }
      if (idx.size() < 2) continue;
      for (size_t i0 = 0; i0 < idx.size(); i0 += TINY_MULTI) {
        m->n = (int)std::min<size_t>(TINY_MULTI, idx.size() - i0);
        m->off[0] = 0;
        for (int k = 0; k < TINY_MULTI; ++k) {
          if (k >= m->n) { m->off[k + 1] = m->off[m->n]; m->gx[k] = 1; continue; }
          m->p[k] = ks[i0 + k]; m->gx[k] = gxs[i0 + k];
          m->off[k + 1] = m->off[k] + gxs[i0 + k] * d[idx[i0 + k]]->N;
          handled[idx[i0 + k]] = 1;
        }
        xh_note_kernel("conv3_tiny_wgrad_multi_kernel<%d -> %d>", cin, 3 - cin);
        auto launch = [&]() -> int {
          XH_DISPATCH_T(types[t], {
            if (cin == 1) hipLaunchKernelGGL((conv3_tiny_wgrad_multi_kernel<T, 1, 2>), dim3(m->off[m->n]), dim3(256), 0, st, *m);
            else hipLaunchKernelGGL((conv3_tiny_wgrad_multi_kernel<T, 2, 1>), dim3(m->off[m->n]), dim3(256), 0, st, *m);
          });
          return XH_OK;
        };
        if (launch() != XH_OK || xh_launch_status() != XH_OK) rc_all = XH_ERR_HIP;
      }
    }
  delete m;
  return rc_all;
}
