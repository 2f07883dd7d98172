// Direct 3D convolution kernels for tiny channel counts (4..48 per group) on gfx950.
//
// Layout: NCDHW, W contiguous.  A workgroup of 256 threads (4 waves) owns an output tile of
// TW x 8 x TD voxels (TW = 4*TXN, TD = 32/TXN); each thread produces 4 consecutive outputs along W for COB
// output channels.  The input halo tile is staged into LDS once per chunk of input channels with the
// producing norm + activation applied on the way in (x*sc+sh, leaky) so normalised activations never
// round-trip through HBM; weights for the chunk sit next to it in LDS laid out [ci][tap][co] so one
// broadcast ds_read feeds COB FMAs per lane.  Accumulation is fp32.
//
// The same kernel is the stride-1 data gradient (transposed=1: roles of Cin/Cout swapped and taps flipped
// while reading the forward-layout weights) with a fused epilogue that multiplies by leaky'() of the forward
// input's normalised value and block-reduces the two sums the InstanceNorm/BatchNorm backward needs.
#include <cstdarg>
#include <cstdio>
#include "common.h"
#include "conv_pack.h"
#include "fanin.h"
#include "../../include/xlstm_hved.h"

extern "C" int xh_norm_finalize(void* stream, int mode, const double* red, int N, int C, long long count, int gs, float eps,
                                const float* gamma, const float* beta, float* running_mean, float* running_var, int steps, float* sc,
                                float* sh, float* mean, float* rstd);                                                   // eltwise.hip
int xh_conv3_tiny_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p);                                   // conv3_tiny.hip
int xh_conv3_tiny_wgrad_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* dw, float* db);

struct ConvK {
  xh_conv_desc d;
  xh_conv_ptrs p;
  int Cin_g, Cout_g, ncob, tilesW, tilesH, tilesD;
  unsigned char* fan;           // statistics fan-in block of this launch (fanin.h), or nullptr: direct atomics
};

// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float conv_weight(const ConvK& a, int g, int co_g, int ci_g, int tap, int K3) {
  const int gpp = a.d.groups / a.d.n_wptr;   // groups per weight pointer
  const float* wp = a.p.w[g / gpp];
  const int gl = g % gpp;
  if (!a.d.transposed) return wp[((long long)(gl * a.Cout_g + co_g) * a.Cin_g + ci_g) * K3 + tap];
  return wp[((long long)(gl * a.Cin_g + ci_g) * a.Cout_g + co_g) * K3 + (K3 - 1 - tap)];
}
__device__ __forceinline__ float conv_bias(const ConvK& a, int g, int co_g) {
  const int gpp = a.d.groups / a.d.n_wptr;
  const float* bp = a.p.b[g / gpp];
  return bp ? bp[(g % gpp) * a.Cout_g + co_g] : 0.f;
}

template <typename T>
__device__ __forceinline__ const T* in_plane(const ConvK& a, int n, int c, long long dhw) {
  return c < a.d.Ca ? (const T*)a.p.xa + n * a.d.xa_bs + (long long)c * dhw
                    : (const T*)a.p.xb + n * a.d.xb_bs + (long long)(c - a.d.Ca) * dhw;
}
template <typename T>
__device__ __forceinline__ const T* epi_plane(const ConvK& a, int n, int c, long long dhw) {
  return c < a.d.Cea ? (const T*)a.p.ea + n * a.d.ea_bs + (long long)c * dhw
                     : (const T*)a.p.eb + n * a.d.eb_bs + (long long)(c - a.d.Cea) * dhw;
}

// Epilogue for NV consecutive outputs of one (n, c) row starting at spatial offset `sp`; `valid` = how many of
// them exist.  Returns partial sums through s0/s1.
// MODE >= 0: the epilogue variant is a compile-time constant (MODE = xh_conv_desc.epi, no activation); MODE < 0: both are read
// from the descriptor at run time -- a switch per output value, which for a lane with 64 outputs compiles to a branch forest
// of several thousand instructions (the k = 1 kernel spent 6 of its 11 us there).  ALIGNED: the launch plan has checked that
// every run is complete and 16-byte aligned (no scalar fallback paths in the code).
template <typename T, int NV, int MODE = -1, bool ALIGNED = false>
__device__ __forceinline__ void conv_epilogue(const ConvK& a, int n, int c, long long odhw, long long sp, int valid,
                                              float bias, float (&val)[NV], double& s0, double& s1) {
  T* yp = (T*)a.p.y + n * a.d.y_bs + (long long)c * odhw + sp;
  // whole-vector accesses (16 B, or 8 B for 4 bf16) when the run is complete and aligned
  constexpr bool WIDE = NV == VWT<T>::v;
  const bool vec = ALIGNED || ((NV == 4 || WIDE) && valid == NV && ((odhw | sp) % NV) == 0);
  if (ALIGNED) valid = NV;
  const int epi = MODE == 3 ? 0 : MODE >= 0 ? MODE : a.d.epi;       // MODE 3: sigmoid (stored threshold-exact, common.h), no epilogue
  float ev[NV];
  float esc = 0.f, esh = 0.f;
  if (epi == 1) {
    esc = a.p.e_sc[n * a.d.Cout + c];
    esh = a.p.e_sh[n * a.d.Cout + c];
    const T* ep = epi_plane<T>(a, n, c, odhw) + sp;
    if (ALIGNED || (vec && ((a.d.ea_bs | a.d.eb_bs) % NV) == 0)) {
      if constexpr (WIDE) {
        ldvec(ep, 0, ev);
      } else if constexpr (NV == 4) {
        ld4(ep, 0, ev);
      }
    } else {
#pragma unroll
      for (int v = 0; v < NV; ++v) ev[v] = v < valid ? ldf(ep, v) : 0.f;
    }
  }
  float t0 = 0.f, t1 = 0.f;       // this run in fp32; the lane's running sums are fp64 (statistics precision, common.h)
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    float o = MODE == 3 ? apply_act_as(yp, val[v] + bias, XH_ACT_SIGMOID, 0.f)
                        : MODE >= 0 ? val[v] + bias : apply_act_as(yp, val[v] + bias, a.d.act, a.d.act_slope);
    if (v < valid) {
      if (epi == 1) {
        o = rnd_as(yp, o * ((ev[v] * esc + esh) > 0.f ? 1.f : a.d.e_slope));
        t0 += o;
        t1 += o * ev[v];
      } else if (epi == 2) {
        o = rnd_as(yp, o);
        t0 += o;
        t1 += o * o;
      }
    }
    val[v] = o;
  }
  if (epi) { s0 += (double)t0; s1 += (double)t1; }
  if (ALIGNED || (vec && (a.d.y_bs % NV) == 0)) {
    if constexpr (WIDE) {
      stvec(yp, 0, val);
    } else if constexpr (NV == 4) {
      st4(yp, 0, val);
    }
  } else {
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (v < valid) stf(yp, v, val[v]);
  }
}

template <int COB>
__device__ __forceinline__ void conv_reduce_out(const ConvK& a, int n, int g, int cob, double (&s0)[COB],
                                                double (&s1)[COB], double* s_red) {
  double v[2 * COB];
#pragma unroll
  for (int i = 0; i < COB; ++i) { v[2 * i] = s0[i]; v[2 * i + 1] = s1[i]; }
  block_sum_d<2 * COB>(v, s_red, blockDim.x >> 6);
  // many workgroups per (sample, channel block): two-level fan-in instead of gridDim.x same-line atomics (fanin.h)
  if (a.fan && !fan_in<2 * COB>(a.fan + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * FAN_UNIT_BYTES, blockIdx.x, gridDim.x,
                                s_red, reinterpret_cast<int*>(s_red + 2 * COB)))
    return;
  if ((int)threadIdx.x < 2 * COB) {
    const int co_g = cob * COB + (threadIdx.x >> 1);
    if (co_g < a.Cout_g) {
      const int c = g * a.Cout_g + co_g;
      atomicAdd(&a.p.red[((long long)n * a.d.Cout + c) * 2 + (threadIdx.x & 1)], s_red[threadIdx.x]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// k = 3 / 7 forward (and stride-1 dgrad)
// ---------------------------------------------------------------------------------------------------
template <int K, int S>
struct ConvCfg {
  static constexpr int CIC = (K == 3 && S == 1) ? 4 : (K == 7 ? 2 : 1);
};

template <typename T, int K, int S, int COB, int TXN>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvK a) {
  constexpr int VW = 4, TW = TXN * VW, TH = 8, TD = 32 / TXN;
  constexpr int IW = (TW - 1) * S + K, IH = (TH - 1) * S + K, ID = (TD - 1) * S + K;
  constexpr int IWP = (IW + 3) / 4 * 4;
  constexpr int CIC = ConvCfg<K, S>::CIC;
  constexpr int K3 = K * K * K;
  constexpr int ROWN = (VW - 1) * S + K;
  constexpr int PAD = K / 2;
  constexpr int UNR = (K == 3) ? 3 : 1;
  __shared__ __attribute__((aligned(16))) float s_in[CIC * ID * IH * IWP];
  __shared__ __attribute__((aligned(16))) float s_w[CIC * K3 * COB];
  __shared__ double s_red[4 * 2 * COB];

  const int tid = threadIdx.x;
  const int tx = tid % TXN, ty = (tid / TXN) % TH, tz = tid / (TXN * TH);
  int t = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tw = t % a.tilesW; t /= a.tilesW;
  const int th = t % a.tilesH;
  const int td = t / a.tilesH;
  const int cob = blockIdx.y;
  const int n = blockIdx.z / a.d.groups, g = blockIdx.z % a.d.groups;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long dhw = (long long)D * H * W;
  const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
  const int id0 = od0 * S - PAD, ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;

  float acc[COB][VW];
#pragma unroll
  for (int i = 0; i < COB; ++i)
#pragma unroll
    for (int v = 0; v < VW; ++v) acc[i][v] = 0.f;

  for (int c0 = 0; c0 < a.Cin_g; c0 += CIC) {
    const int ncc = min(CIC, a.Cin_g - c0);           // live input channels of this chunk (1 for depthwise convs)
    __syncthreads();
    // ---- stage inputs (transform applied before zero padding) ----
    for (int cc = 0; cc < ncc; ++cc) {
      const int ci_g = c0 + cc;
      float* dst = s_in + cc * (ID * IH * IWP);
      if (ci_g < a.Cin_g) {
        const int c = g * a.Cin_g + ci_g;
        const T* src = in_plane<T>(a, n, c, dhw);
        float sc = 1.f, sh = 0.f;
        if (a.d.pre) { sc = a.p.pre_sc[n * a.d.Cin + c]; sh = a.p.pre_sh[n * a.d.Cin + c]; }
        for (int idx = tid; idx < ID * IH * IWP; idx += 256) {
          const int wx = idx % IWP;
          const int r = idx / IWP;
          const int hy = r % IH, dz = r / IH;
          const int gd = id0 + dz, gh = ih0 + hy, gw = iw0 + wx;
          float v = 0.f;
          if (wx < IW && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W) {
            v = ldf(src, ((long long)gd * H + gh) * W + gw);
            if (a.d.pre) v = leaky(v * sc + sh, a.d.pre_slope);
          }
          dst[idx] = v;
        }
      } else {
        for (int idx = tid; idx < ID * IH * IWP; idx += 256) dst[idx] = 0.f;
      }
    }
    // ---- stage weights [cc][tap][co] ----
    for (int idx = tid; idx < ncc * K3 * COB; idx += 256) {
      const int co = idx % COB;
      const int r = idx / COB;
      const int tap = r % K3, cc = r / K3;
      const int ci_g = c0 + cc, co_g = cob * COB + co;
      s_w[idx] = (ci_g < a.Cin_g && co_g < a.Cout_g) ? conv_weight(a, g, co_g, ci_g, tap, K3) : 0.f;
    }
    __syncthreads();
    // ---- compute ----
    const float* base = s_in + ((tz * S) * IH + ty * S) * IWP + tx * VW * S;
#pragma unroll 1
    for (int cc = 0; cc < ncc; ++cc) {
#pragma unroll UNR
      for (int kd = 0; kd < K; ++kd) {
#pragma unroll UNR
        for (int kh = 0; kh < K; ++kh) {
          const float* row = base + ((cc * ID + kd) * IH + kh) * IWP;
          float r[ROWN];
#pragma unroll
          for (int i = 0; i + 3 < ROWN; i += 4) {
            const float4 q = *reinterpret_cast<const float4*>(row + i);
            r[i] = q.x; r[i + 1] = q.y; r[i + 2] = q.z; r[i + 3] = q.w;
          }
#pragma unroll
          for (int i = ROWN / 4 * 4; i < ROWN; ++i) r[i] = row[i];
          const float* wrow = s_w + ((cc * K3) + (kd * K + kh) * K) * COB;
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            float wr[COB];
#pragma unroll
            for (int co = 0; co < COB; ++co) wr[co] = wrow[kw * COB + co];
#pragma unroll
            for (int co = 0; co < COB; ++co)
#pragma unroll
              for (int v = 0; v < VW; ++v) acc[co][v] = fmaf(wr[co], r[v * S + kw], acc[co][v]);
          }
        }
      }
    }
  }
  // ---- epilogue ----
  const int od = od0 + tz, oh = oh0 + ty, ow = ow0 + tx * VW;
  const int Do = a.d.Do, Ho = a.d.Ho, Wo = a.d.Wo;
  const long long odhw = (long long)Do * Ho * Wo;
  int valid = 0;
  if (od < Do && oh < Ho && ow < Wo) valid = min(VW, Wo - ow);
  double s0[COB], s1[COB];
#pragma unroll
  for (int co = 0; co < COB; ++co) {
    s0[co] = 0.0; s1[co] = 0.0;
    const int co_g = cob * COB + co;
    if (co_g < a.Cout_g && valid > 0) {
      const int c = g * a.Cout_g + co_g;
      conv_epilogue<T, VW>(a, n, c, odhw, ((long long)od * Ho + oh) * Wo + ow, valid, conv_bias(a, g, co_g), acc[co],
                           s0[co], s1[co]);
    }
  }
  if (a.d.epi) conv_reduce_out<COB>(a, n, g, cob, s0, s1, s_red);
}

// ---------------------------------------------------------------------------------------------------
// Depthwise k = 3, stride 1 (BasicConv conv_blocks, the skip-return ResBlock's dwconvs, and their data gradients).
// 27 FMA per output against 2+2 bytes: bandwidth bound, so no LDS tile -- each lane produces 4 consecutive outputs of
// one channel from 9 row segments read straight through L1/L2 (rows are shared by neighbouring lanes/blocks).
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_dw3_kernel(const ConvK a) {
  __shared__ double s_red[4 * 2];
  const int tid = threadIdx.x;
  const int tx = tid & 7, ty = (tid >> 3) & 7, tz = tid >> 6;
  const int c = blockIdx.y, n = blockIdx.z;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long dhw = (long long)D * H * W;
  const int gpp = a.d.groups / a.d.n_wptr;
  const float* wp = a.p.w[c / gpp] + (long long)(c % gpp) * 27;
  float wgt[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) wgt[i] = wp[a.d.transposed ? 26 - i : i];
  float sc = 1.f, sh = 0.f;
  if (a.d.pre) { sc = a.p.pre_sc[n * a.d.Cin + c]; sh = a.p.pre_sh[n * a.d.Cin + c]; }
  const T* src = in_plane<T>(a, n, c, dhw);
  const bool vec4 = (W % 4 == 0) && (dhw % 4 == 0) && (a.d.xa_bs % 4 == 0) && (a.d.xb_bs % 4 == 0);
  double s0 = 0.0, s1 = 0.0;
  const float* bp = a.p.b[c / gpp];
  const float bias = bp ? bp[c % gpp] : 0.f;
  // persistent over tiles: the fused reduction costs one atomic per workgroup, not one per tile (same-address fp64
  // atomics from thousands of tiny workgroups serialise at the memory side)
  const int ntiles = a.tilesW * a.tilesH * a.tilesD;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  int t = tile;
  const int tw = t % a.tilesW; t /= a.tilesW;
  const int th = t % a.tilesH;
  const int td = t / a.tilesH;
  const int od = td * 4 + tz, oh = th * 8 + ty, ow = tw * 32 + tx * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int valid = 0;
  if (od < D && oh < H && ow < W) {
    valid = min(4, W - ow);
    // Every load is unconditional on a clamped (always valid) address and masked arithmetically afterwards, so the
    // 27 loads issue back to back behind ONE wait instead of 27 exec-masked branches with a wait each.
    float q[9][6], rmask[9];
    const int owl = max(ow - 1, 0), owr = min(ow + 4, W - 1);
    const float lmask = ow > 0 ? 1.f : 0.f, rmk = (ow + 4 < W) ? 1.f : 0.f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int gd = od - 1 + kd, gh = oh - 1 + kh;
        const int ri = kd * 3 + kh;
        rmask[ri] = ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H) ? 1.f : 0.f;
        const T* row = src + ((long long)min(max(gd, 0), D - 1) * H + min(max(gh, 0), H - 1)) * W;
        if (vec4 && valid == 4) {                    // block-uniform in practice (W % 4 == 0)
          float t4[4];
          ld4(row, ow, t4);
          q[ri][1] = t4[0]; q[ri][2] = t4[1]; q[ri][3] = t4[2]; q[ri][4] = t4[3];
        } else {
#pragma unroll
          for (int i = 1; i < 5; ++i) q[ri][i] = ldf(row, min(ow - 1 + i, W - 1));
        }
        q[ri][0] = ldf(row, owl);
        q[ri][5] = ldf(row, owr);
      }
#pragma unroll
    for (int ri = 0; ri < 9; ++ri) {
      float m[6];
      m[0] = rmask[ri] * lmask;
      m[5] = rmask[ri] * rmk;
#pragma unroll
      for (int i = 1; i < 5; ++i) m[i] = (ow - 1 + i < W) ? rmask[ri] : 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float v = q[ri][i];
        if (a.d.pre) v = leaky(v * sc + sh, a.d.pre_slope);
        q[ri][i] = v * m[i];                         // zero padding is applied after the transform
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = fmaf(wgt[ri * 3 + kw], q[ri][v + kw], acc[v]);
    }
  }
  if (valid > 0) conv_epilogue<T, 4>(a, n, c, dhw, ((long long)od * H + oh) * W + ow, valid, bias, acc, s0, s1);
  }
  if (a.d.epi) {
    double v[2] = {s0, s1};
    block_sum_d<2>(v, s_red, 4);
    if (a.fan && !fan_in<2>(a.fan + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * FAN_UNIT_BYTES, blockIdx.x, gridDim.x, s_red,
                            reinterpret_cast<int*>(s_red + 2)))
      return;
    if (tid < 2) atomicAdd(&a.p.red[((long long)n * a.d.Cout + c) * 2 + tid], s_red[tid]);
  }
}

// ---------------------------------------------------------------------------------------------------
// Depthwise k = 3, stride 1, large volumes: sliding window along D.  A lane owns one 16-byte run of one output row
// (8 bf16 / 4 fp32 voxels) and marches through the planes of its depth segment.  Each input plane is read once per lane
// (3 rows x one 16-byte load; the +-1 neighbours along W come from the adjacent lanes by wave shuffle) and is folded into
// the three output planes it touches, whose accumulators rotate through registers.  ~1.1 load instructions per output
// instead of 6.75, which is what bounds the tile kernel above (the L1/TA path, not HBM).
// ---------------------------------------------------------------------------------------------------
template <typename T, int TXN>
__global__ __launch_bounds__(256) void conv_dw3_slide_kernel(const ConvK a, int sd) {
  constexpr int VW = VWT<T>::v, TH = 256 / TXN;
  __shared__ double s_red[4 * 2];
  const int tid = threadIdx.x;
  const int tx = tid % TXN, ty = tid / TXN;
  const int c = blockIdx.y, n = blockIdx.z;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  int t = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tw = t % a.tilesW; t /= a.tilesW;
  const int th = t % a.tilesH;
  const int ds = t / a.tilesH;
  const int oh = th * TH + ty, ow = tw * TXN * VW + tx * VW;
  const bool active = oh < H && ow < W;               // W % VW == 0: a run is whole or absent
  const int owc = active ? ow : 0;
  const int d_begin = ds * sd, d_end = min(D, d_begin + sd);
  const int gpp = a.d.groups / a.d.n_wptr;
  const float* wp = a.p.w[c / gpp] + (long long)(c % gpp) * 27;
  float wgt[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) wgt[i] = wp[a.d.transposed ? 26 - i : i];
  float sc = 1.f, sh = 0.f;
  if (a.d.pre) { sc = a.p.pre_sc[n * a.d.Cin + c]; sh = a.p.pre_sh[n * a.d.Cin + c]; }
  const float* bp = a.p.b[c / gpp];
  const float bias = bp ? bp[c % gpp] : 0.f;
  const T* src = in_plane<T>(a, n, c, dhw);
  // the three rows this lane reads in every plane (clamped: always a valid address; masked arithmetically)
  int roff[3];
  float rmask[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int gh = oh - 1 + kh;
    rmask[kh] = (active && (unsigned)gh < (unsigned)H) ? 1.f : 0.f;
    roff[kh] = min(max(gh, 0), H - 1) * W;
  }
  const bool multi_w = a.tilesW > 1;                  // only then can a tile edge be interior to the volume
  const bool edge_l = tx == 0, edge_r = tx == TXN - 1 || ow + VW >= W;
  const bool need_l = multi_w && edge_l && ow > 0, need_r = multi_w && edge_r && ow + VW < W;
  const int col_l = max(ow - 1, 0), col_r = min(ow + VW, W - 1);

  float accA[VW], accB[VW], accC[VW];
#pragma unroll
  for (int v = 0; v < VW; ++v) accA[v] = accB[v] = accC[v] = 0.f;
  double s0 = 0.0, s1 = 0.0;
  float nxt[3][VW], nhl[3] = {0.f, 0.f, 0.f}, nhr[3] = {0.f, 0.f, 0.f};
  auto load_plane = [&](int p) {
    const T* pl = src + (long long)p * hw;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      ldvec(pl, roff[kh] + owc, nxt[kh]);
      if (multi_w) {                                  // block-uniform
        nhl[kh] = ldf(pl, roff[kh] + col_l);
        nhr[kh] = ldf(pl, roff[kh] + col_r);
      }
    }
  };
  bool nxt_ok = d_begin - 1 >= 0;
  if (nxt_ok) load_plane(d_begin - 1);
  for (int p = d_begin - 1; p <= d_end; ++p) {        // p is block-uniform
    float cur[3][VW], hl[3], hr[3];
    const bool cur_ok = nxt_ok;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int v = 0; v < VW; ++v) cur[kh][v] = nxt[kh][v];
      hl[kh] = nhl[kh];
      hr[kh] = nhr[kh];
    }
    if (p + 1 <= d_end) {                             // prefetch the next plane behind this plane's arithmetic
      nxt_ok = p + 1 < D;
      if (nxt_ok) load_plane(p + 1);
    }
    if (cur_ok) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        float r[VW + 2];
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          float x = cur[kh][v];
          if (a.d.pre) x = leaky(x * sc + sh, a.d.pre_slope);
          r[v + 1] = x * rmask[kh];                   // zero padding is applied after the transform
        }
        float l = __shfl_up(r[VW], 1, 64), rr = __shfl_down(r[1], 1, 64);
        if (edge_l) {
          float x = hl[kh];
          if (a.d.pre) x = leaky(x * sc + sh, a.d.pre_slope);
          l = need_l ? x * rmask[kh] : 0.f;
        }
        if (edge_r) {
          float x = hr[kh];
          if (a.d.pre) x = leaky(x * sc + sh, a.d.pre_slope);
          rr = need_r ? x * rmask[kh] : 0.f;
        }
        r[0] = l;
        r[VW + 1] = rr;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int v = 0; v < VW; ++v) {
            accC[v] = fmaf(wgt[0 * 9 + kh * 3 + kw], r[v + kw], accC[v]);   // plane p is tap kd=0 of output p+1
            accB[v] = fmaf(wgt[1 * 9 + kh * 3 + kw], r[v + kw], accB[v]);   //            tap kd=1 of output p
            accA[v] = fmaf(wgt[2 * 9 + kh * 3 + kw], r[v + kw], accA[v]);   //            tap kd=2 of output p-1
          }
      }
    }
    const int od = p - 1;
    if (od >= d_begin && active)
      conv_epilogue<T, VW>(a, n, c, dhw, ((long long)od * H + oh) * W + ow, VW, bias, accA, s0, s1);
#pragma unroll
    for (int v = 0; v < VW; ++v) { accA[v] = accB[v]; accB[v] = accC[v]; accC[v] = 0.f; }
  }
  if (a.d.epi) {
    double v[2] = {s0, s1};
    block_sum_d<2>(v, s_red, 4);
    if (a.fan && !fan_in<2>(a.fan + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * FAN_UNIT_BYTES, blockIdx.x, gridDim.x, s_red,
                            reinterpret_cast<int*>(s_red + 2)))
      return;
    if (tid < 2) atomicAdd(&a.p.red[((long long)n * a.d.Cout + c) * 2 + tid], s_red[tid]);
  }
}

// ---------------------------------------------------------------------------------------------------
// k = 1 forward / dgrad: no halo, inputs straight from global memory (4 voxels per lane, vectorised).
// ---------------------------------------------------------------------------------------------------
// MODE: see conv_epilogue (>= 0: no activation, epilogue variant MODE, straight-line code)
// (the body takes the block coordinates as parameters named like the built-ins: it also runs inside conv1x1_multi_kernel)
template <typename T, int COB, bool VEC, int CIC = 4, int MODE = -1>   // CIC input channels per step: their loads are issued together
__device__ __forceinline__ void conv1x1_body(const uint3 blockIdx, const uint3 gridDim, const ConvK& a) {
  constexpr int VW = VEC ? VWT<T>::v : 4;              // 16-byte runs when the layout allows
  __shared__ float s_w[132 * COB];
  __shared__ double s_red[4 * 2 * COB];
  const int tid = threadIdx.x;
  const int cob = blockIdx.y;
  const int n = blockIdx.z / a.d.groups, g = blockIdx.z % a.d.groups;
  const long long dhw = (long long)a.d.D * a.d.H * a.d.W;
  const int cin_pad = (a.Cin_g + CIC - 1) / CIC * CIC;
  for (int idx = tid; idx < cin_pad * COB; idx += 256) {
    const int co = idx % COB, ci_g = idx / COB, co_g = cob * COB + co;
    s_w[idx] = (co_g < a.Cout_g && ci_g < a.Cin_g) ? conv_weight(a, g, co_g, ci_g, 0, 1) : 0.f;   // pad rows: zero weight
  }
  float bias[COB];                                      // requested before the barrier: not a round trip of its own in the epilogue
#pragma unroll
  for (int co = 0; co < COB; ++co) bias[co] = cob * COB + co < a.Cout_g ? conv_bias(a, g, cob * COB + co) : 0.f;
  __syncthreads();
  double s0[COB], s1[COB];
#pragma unroll
  for (int co = 0; co < COB; ++co) { s0[co] = 0.0; s1[co] = 0.0; }
  for (long long q0 = ((long long)blockIdx.x * 256 + tid) * VW; q0 < dhw; q0 += (long long)gridDim.x * 256 * VW) {
    const int valid = (int)min((long long)VW, dhw - q0);
    float acc[COB][VW];
#pragma unroll
    for (int i = 0; i < COB; ++i)
#pragma unroll
      for (int v = 0; v < VW; ++v) acc[i][v] = 0.f;
    for (int ci0 = 0; ci0 < a.Cin_g; ci0 += CIC) {
      float x[CIC][VW];
#pragma unroll
      for (int j = 0; j < CIC; ++j) {                  // clamped channel: always a valid address, its weight is zero
        const int c = g * a.Cin_g + min(ci0 + j, a.Cin_g - 1);
        const T* src = in_plane<T>(a, n, c, dhw) + q0;
        if constexpr (VEC) {
          ldvec(src, 0, x[j]);
        } else {
#pragma unroll
          for (int v = 0; v < VW; ++v) x[j][v] = v < valid ? ldf(src, v) : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < CIC; ++j) {
        if (a.d.pre) {
          const int c = g * a.Cin_g + min(ci0 + j, a.Cin_g - 1);
          const float sc = a.p.pre_sc[n * a.d.Cin + c], sh = a.p.pre_sh[n * a.d.Cin + c];
#pragma unroll
          for (int v = 0; v < VW; ++v) x[j][v] = leaky(x[j][v] * sc + sh, a.d.pre_slope);
        }
#pragma unroll
        for (int co = 0; co < COB; ++co) {
          const float w = s_w[(ci0 + j) * COB + co];
#pragma unroll
          for (int v = 0; v < VW; ++v) acc[co][v] = fmaf(w, x[j][v], acc[co][v]);
        }
      }
    }
#pragma unroll
    for (int co = 0; co < COB; ++co) {
      const int co_g = cob * COB + co;
      if (co_g < a.Cout_g) {
        const int c = g * a.Cout_g + co_g;
        conv_epilogue<T, VW, MODE, VEC>(a, n, c, dhw, q0, valid, bias[co], acc[co], s0[co], s1[co]);
      }
    }
  }
  if (MODE >= 0 ? (MODE == 1 || MODE == 2) : a.d.epi != 0) conv_reduce_out<COB>(a, n, g, cob, s0, s1, s_red);
}
template <typename T, int COB, bool VEC, int CIC = 4, int MODE = -1>
__global__ __launch_bounds__(256) void conv1x1_kernel(const ConvK a) {
  conv1x1_body<T, COB, VEC, CIC, MODE>(uint3{blockIdx.x, blockIdx.y, blockIdx.z}, uint3{gridDim.x, gridDim.y, gridDim.z}, a);
}
// Up to XH_LEVELS_MAX k = 1 convolutions in ONE launch (xh_conv1x1_multi: the four fusion levels' VU-block convs and their data
// gradients, RA_HVED.py:599-601): a workgroup finds its problem from its index and runs that problem's ordinary body -- with the
// output-channel block / channel-step instance the single launch would have picked -- at the problem's own block coordinates.
struct C1Multi { int n; int off[XH_LEVELS_MAX + 1]; int gx[XH_LEVELS_MAX], gy[XH_LEVELS_MAX], gz[XH_LEVELS_MAX], cob[XH_LEVELS_MAX], cic[XH_LEVELS_MAX]; ConvK p[XH_LEVELS_MAX]; };
static_assert(sizeof(C1Multi) <= 4096, "the problem table travels in the kernel arguments");
template <typename T, int MODE>
__global__ __launch_bounds__(256) void conv1x1_multi_kernel(const C1Multi m) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < XH_LEVELS_MAX; ++k)
    if (k < m.n && (int)blockIdx.x >= m.off[k]) i = k;
  const int local = (int)blockIdx.x - m.off[i];
  const int gx = m.gx[i], gy = m.gy[i], gz = m.gz[i];
  if (local >= gx * gy * gz) return;
  const int zy = local / gx;
  const uint3 vb = {(unsigned)(local - zy * gx), (unsigned)(zy % gy), (unsigned)(zy / gy)}, vg = {(unsigned)gx, (unsigned)gy, (unsigned)gz};
  const ConvK& a = m.p[i];
  if (m.cic[i] == 16) { conv1x1_body<T, 2, true, 16, MODE>(vb, vg, a); return; }
  switch (m.cob[i]) {
    case 1: conv1x1_body<T, 1, true, 4, MODE>(vb, vg, a); break;
    case 2: conv1x1_body<T, 2, true, 4, MODE>(vb, vg, a); break;
    case 4: conv1x1_body<T, 4, true, 4, MODE>(vb, vg, a); break;
    default: conv1x1_body<T, 8, true, 4, MODE>(vb, vg, a);
  }
}

// ---------------------------------------------------------------------------------------------------
// k = 3, stride = 2 forward (the DRB convs, RA_HVED.py:396-397): outputs are 1/8 of the inputs and the tensors are
// small, so a direct gather (one lane per output voxel, COB output channels, weights in LDS) beats tiling: no
// per-chunk barriers, full-chip parallelism even for 8^3 outputs.
// ---------------------------------------------------------------------------------------------------
template <typename T, int COB>
__global__ __launch_bounds__(256) void conv3_s2_gather_kernel(const ConvK a) {
  extern __shared__ float s_dyn[];                    // [Cin_g][27][COB] weights, then reduction scratch
  float* s_w = s_dyn;
  double* s_red = reinterpret_cast<double*>(s_dyn + ((a.Cin_g * 27 * COB + 1) & ~1));
  const int tid = threadIdx.x;
  const int cob = blockIdx.y;
  const int n = blockIdx.z / a.d.groups, g = blockIdx.z % a.d.groups;
  for (int idx = tid; idx < a.Cin_g * 27 * COB; idx += 256) {
    const int co = idx % COB;
    const int r = idx / COB;
    const int tap = r % 27, ci_g = r / 27;
    const int co_g = cob * COB + co;
    s_w[idx] = co_g < a.Cout_g ? conv_weight(a, g, co_g, ci_g, tap, 27) : 0.f;
  }
  __syncthreads();
  const int D = a.d.D, H = a.d.H, W = a.d.W, Do = a.d.Do, Ho = a.d.Ho, Wo = a.d.Wo;
  const long long dhw = (long long)D * H * W, odhw = (long long)Do * Ho * Wo;
  const long long q = (long long)blockIdx.x * 256 + tid;
  const bool ok = q < odhw;
  float acc[COB];
#pragma unroll
  for (int i = 0; i < COB; ++i) acc[i] = 0.f;
  if (ok) {
    const int ow = (int)(q % Wo), oh = (int)((q / Wo) % Ho), od = (int)(q / ((long long)Wo * Ho));
    for (int ci_g = 0; ci_g < a.Cin_g; ++ci_g) {
      const int c = g * a.Cin_g + ci_g;
      const T* src = in_plane<T>(a, n, c, dhw);
      float sc = 1.f, sh = 0.f;
      if (a.d.pre) { sc = a.p.pre_sc[n * a.d.Cin + c]; sh = a.p.pre_sh[n * a.d.Cin + c]; }
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int gd = 2 * od - 1 + kd;
        if ((unsigned)gd >= (unsigned)D) continue;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int gh = 2 * oh - 1 + kh;
          if ((unsigned)gh >= (unsigned)H) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int gw = 2 * ow - 1 + kw;
            if ((unsigned)gw >= (unsigned)W) continue;
            float v = ldf(src, ((long long)gd * H + gh) * W + gw);
            if (a.d.pre) v = leaky(v * sc + sh, a.d.pre_slope);
            const float* wr = s_w + (ci_g * 27 + (kd * 3 + kh) * 3 + kw) * COB;
#pragma unroll
            for (int co = 0; co < COB; ++co) acc[co] = fmaf(wr[co], v, acc[co]);
          }
        }
      }
    }
  }
  double s0[COB], s1[COB];
#pragma unroll
  for (int co = 0; co < COB; ++co) {
    s0[co] = 0.0; s1[co] = 0.0;
    const int co_g = cob * COB + co;
    if (co_g < a.Cout_g && ok) {
      float v1[1] = {acc[co]};
      conv_epilogue<T, 1>(a, n, g * a.Cout_g + co_g, odhw, q, 1, conv_bias(a, g, co_g), v1, s0[co], s1[co]);
    }
  }
  if (a.d.epi) conv_reduce_out<COB>(a, n, g, cob, s0, s1, s_red);
}

// ---------------------------------------------------------------------------------------------------
// k = 3, stride = 2 forward, vectorised: a lane produces VW/2 consecutive outputs of one row from 16-byte input runs
// (input columns 2*ow0 .. 2*ow0+VW-1; the one column to the left comes from the neighbouring lane by wave shuffle),
// i.e. 9 wide loads per input channel instead of 27 two-byte ones per output.  Lanes of a wave tile whole rows.
// ---------------------------------------------------------------------------------------------------
// KS > 1 (small outputs: the deep levels launch a few dozen workgroups whose lanes each walk Cin_g x 27 x COB FMAs -- 32 -> 64
// channels @16^3 -> 8^3 ran 63 us on 32 workgroups): a block keeps 256 / KS lanes and its KS thread groups split the input
// channels, partial sums meet in LDS (part_off floats into the dynamic segment) and group 0 runs the epilogue.
template <typename T, int COB>
__global__ __launch_bounds__(256) void conv3_s2_vec_kernel(const ConvK a, int LW, int KS, int part_off, int fin_off) {
  constexpr int VW = VWT<T>::v, OW = VW / 2;
  extern __shared__ float s_dyn[];                    // [Cin_g][27][COB] weights, then reduction scratch
  float* s_w = s_dyn;
  double* s_red = reinterpret_cast<double*>(s_dyn + ((a.Cin_g * 27 * COB + 1) & ~1));
  const int tid = threadIdx.x;
  const int cob = blockIdx.y;
  const int n = blockIdx.z / a.d.groups, g = blockIdx.z % a.d.groups;
  for (int idx = tid; idx < a.Cin_g * 27 * COB; idx += 256) {
    const int co = idx % COB;
    const int r = idx / COB;
    const int tap = r % 27, ci_g = r / 27;
    const int co_g = cob * COB + co;
    s_w[idx] = co_g < a.Cout_g ? conv_weight(a, g, co_g, ci_g, tap, 27) : 0.f;
  }
  // fused InstanceNorm finalisation (xh_conv_ptrs.fin_red; 16-bit storage): the group's scale / shift from the raw sums into
  // LDS, the same bits in every workgroup; workgroup (0, 0, 0) also leaves them and mean / rstd of ALL channels in memory for
  // the backward pass -- no norm_finalize launch in front of the stride-2 convs
  const bool fin = a.p.fin_red != nullptr;
  float* s_fin = s_dyn + fin_off;                     // [2][Cin_g]
  if (fin) {
    const double inv = 1.0 / (double)a.p.fin_count;
    if (tid < a.Cin_g) {
      const int c = n * a.d.Cin + g * a.Cin_g + tid;
      float m_, r_;
      in_finalize(a.p.fin_red[2 * c], a.p.fin_red[2 * c + 1], inv, s_fin[tid], s_fin[a.Cin_g + tid], m_, r_);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
      for (int i = tid; i < a.d.N * a.d.Cin; i += 256)
        in_finalize(a.p.fin_red[2 * i], a.p.fin_red[2 * i + 1], inv, const_cast<float*>(a.p.pre_sc)[i], const_cast<float*>(a.p.pre_sh)[i],
                    a.p.fin_mean[i], a.p.fin_rstd[i]);
  }
  __syncthreads();
  const int D = a.d.D, H = a.d.H, W = a.d.W, Do = a.d.Do, Ho = a.d.Ho, Wo = a.d.Wo;
  const long long dhw = (long long)D * H * W, odhw = (long long)Do * Ho * Wo;
  const int LPB = 256 / KS, ks = tid / LPB, tl = tid - ks * LPB;
  const long long lane_id = (long long)blockIdx.x * LPB + tl;
  const int tx = (int)(lane_id % LW);
  const long long row = lane_id / LW;
  bool ok = row < (long long)Do * Ho;
  const int oh = (int)(row % Ho), od = (int)min(row / Ho, (long long)Do - 1);
  const int ow0 = tx * OW;
  // accumulators as PAIRS of consecutive outputs (v_pk_fma_f32): output j of a lane reads columns 2 j + kw of its run, i.e. the
  // even columns for kw = 0 and 2 and the odd ones for kw = 1 -- with the run split into its even and odd columns every tap is
  // one packed FMA per output pair (the kernel is bound by its vector FMAs: 27 Cin_g COB per output)
  constexpr int NP = OW / 2;
  f32x2_t acc2[COB][NP];
#pragma unroll
  for (int i = 0; i < COB; ++i)
#pragma unroll
    for (int j = 0; j < NP; ++j) acc2[i][j] = f32x2_t{0.f, 0.f};
  const int cpk = (a.Cin_g + KS - 1) / KS;
  const int ci_end = min(a.Cin_g, (ks + 1) * cpk);
  for (int ci_g = ks * cpk; ci_g < ci_end; ++ci_g) {
    const int c = g * a.Cin_g + ci_g;
    const T* src = in_plane<T>(a, n, c, dhw) + 2 * ow0;
    float sc = 1.f, sh = 0.f;
    if (fin) { sc = s_fin[ci_g]; sh = s_fin[a.Cin_g + ci_g]; }
    else if (a.d.pre) { sc = a.p.pre_sc[n * a.d.Cin + c]; sh = a.p.pre_sh[n * a.d.Cin + c]; }
    float x[9][VW], m[9];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {                // clamped rows: always valid addresses, masked below
        const int gd = 2 * od - 1 + kd, gh = 2 * oh - 1 + kh;
        m[kd * 3 + kh] = ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H) ? 1.f : 0.f;
        ldvec(src, ((long long)min(max(gd, 0), D - 1) * H + min(max(gh, 0), H - 1)) * W, x[kd * 3 + kh]);
      }
#pragma unroll
    for (int r9 = 0; r9 < 9; ++r9) {
      float r[VW + 1];
#pragma unroll
      for (int v = 0; v < VW; ++v) {
        float xv = x[r9][v];
        if (a.d.pre) xv = leaky(xv * sc + sh, a.d.pre_slope);
        r[v + 1] = xv * m[r9];                        // zero padding after the transform
      }
      const float l = __shfl_up(r[VW], 1, 64);
      r[0] = tx == 0 ? 0.f : l;                       // column -1 of the volume is padding
      const float* wr = s_w + (ci_g * 27 + r9 * 3) * COB;
      f32x2_t e0[NP], e1[NP], o0[NP];                 // (r[4p], r[4p+2]), (r[4p+2], r[4p+4]), (r[4p+1], r[4p+3])
#pragma unroll
      for (int p2 = 0; p2 < NP; ++p2) {
        e0[p2] = f32x2_t{r[4 * p2], r[4 * p2 + 2]};
        e1[p2] = f32x2_t{r[4 * p2 + 2], r[4 * p2 + 4]};
        o0[p2] = f32x2_t{r[4 * p2 + 1], r[4 * p2 + 3]};
      }
#pragma unroll
      for (int co = 0; co < COB; ++co) {
        const float w0 = wr[co], w1 = wr[COB + co], w2 = wr[2 * COB + co];
#pragma unroll
        for (int p2 = 0; p2 < NP; ++p2)               // same order of the three taps per output as the scalar loop had
          acc2[co][p2] = f32x2_t{w2, w2} * e1[p2] + (f32x2_t{w1, w1} * o0[p2] + (f32x2_t{w0, w0} * e0[p2] + acc2[co][p2]));
      }
    }
  }
  float acc[COB][OW];
#pragma unroll
  for (int i = 0; i < COB; ++i)
#pragma unroll
    for (int j = 0; j < NP; ++j) { acc[i][2 * j] = acc2[i][j].x; acc[i][2 * j + 1] = acc2[i][j].y; }
  if (KS > 1) {                                         // block-uniform
    float* s_part = s_dyn + part_off;                   // [KS - 1][LPB][COB * OW]
    if (ks > 0) {
#pragma unroll
      for (int i = 0; i < COB; ++i)
#pragma unroll
        for (int j = 0; j < OW; ++j) s_part[((ks - 1) * LPB + tl) * (COB * OW) + i * OW + j] = acc[i][j];
    }
    __syncthreads();
    if (ks == 0) {
      for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
        for (int i = 0; i < COB; ++i)
#pragma unroll
          for (int j = 0; j < OW; ++j) acc[i][j] += s_part[((k2 - 1) * LPB + tl) * (COB * OW) + i * OW + j];
    } else {
      ok = false;
    }
  }
  double s0[COB], s1[COB];
#pragma unroll
  for (int co = 0; co < COB; ++co) {
    s0[co] = 0.0; s1[co] = 0.0;
    const int co_g = cob * COB + co;
    if (co_g < a.Cout_g && ok)
      conv_epilogue<T, OW>(a, n, g * a.Cout_g + co_g, odhw, ((long long)od * Ho + oh) * Wo + ow0, OW, conv_bias(a, g, co_g),
                           acc[co], s0[co], s1[co]);
  }
  if (a.d.epi) conv_reduce_out<COB>(a, n, g, cob, s0, s1, s_red);
}

// ---------------------------------------------------------------------------------------------------
// k = 3, stride = 2 data gradient (DRB).  One lane per forward-input voxel, CIB input channels per lane;
// gathers the <= 8 contributing output voxels per tap parity.  Small tensors only (latent resolution).
// ---------------------------------------------------------------------------------------------------
template <typename T, int CIB>
__global__ __launch_bounds__(256) void conv3_dgrad_s2_kernel(const ConvK a) {
  // here a.d describes the FORWARD conv: Cin_g/Cout_g forward; xa = dY (Cout channels), y = dX
  extern __shared__ float s_dyn[];   // [Cout_g][CIB][27] weights + reduction scratch
  float* s_w = s_dyn;
  double* s_red = reinterpret_cast<double*>(s_dyn + ((a.Cout_g * CIB * 27 + 1) & ~1));
  const int tid = threadIdx.x;
  const int cib = blockIdx.y;
  const int n = blockIdx.z / a.d.groups, g = blockIdx.z % a.d.groups;
  const int gpp = a.d.groups / a.d.n_wptr;
  const float* wp = a.p.w[g / gpp];
  const int gl = g % gpp;
  for (int idx = tid; idx < a.Cout_g * CIB * 27; idx += 256) {
    const int tap = idx % 27;
    const int r = idx / 27;
    const int ci = r % CIB, co_g = r / CIB;
    const int ci_g = cib * CIB + ci;
    s_w[idx] = ci_g < a.Cin_g ? wp[((long long)(gl * a.Cout_g + co_g) * a.Cin_g + ci_g) * 27 + tap] : 0.f;
  }
  __syncthreads();
  const int D = a.d.D, H = a.d.H, W = a.d.W, Do = a.d.Do, Ho = a.d.Ho, Wo = a.d.Wo;
  const long long dhw = (long long)D * H * W, odhw = (long long)Do * Ho * Wo;
  const long long q = (long long)blockIdx.x * 256 + tid;
  const bool ok = q < dhw;
  float acc[CIB];
#pragma unroll
  for (int i = 0; i < CIB; ++i) acc[i] = 0.f;
  if (ok) {
    const int w_ = (int)(q % W), h_ = (int)((q / W) % H), d_ = (int)(q / ((long long)W * H));
    for (int kd = 0; kd < 3; ++kd) {
      const int od2 = d_ + 1 - kd;
      if (od2 < 0 || (od2 & 1) || (od2 >> 1) >= Do) continue;
      for (int kh = 0; kh < 3; ++kh) {
        const int oh2 = h_ + 1 - kh;
        if (oh2 < 0 || (oh2 & 1) || (oh2 >> 1) >= Ho) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int ow2 = w_ + 1 - kw;
          if (ow2 < 0 || (ow2 & 1) || (ow2 >> 1) >= Wo) continue;
          const long long osp = ((long long)(od2 >> 1) * Ho + (oh2 >> 1)) * Wo + (ow2 >> 1);
          const int tap = (kd * 3 + kh) * 3 + kw;
          for (int co_g = 0; co_g < a.Cout_g; ++co_g) {
            const int co = g * a.Cout_g + co_g;
            const float dy = ldf((const T*)a.p.xa + n * a.d.xa_bs + (long long)co * odhw, osp);
#pragma unroll
            for (int ci = 0; ci < CIB; ++ci) acc[ci] = fmaf(s_w[(co_g * CIB + ci) * 27 + tap], dy, acc[ci]);
          }
        }
      }
    }
  }
  // epilogue: output channels are forward-input channels; epi sc/sh arrays are [N][Cin]
  float s0[CIB], s1[CIB];
#pragma unroll
  for (int ci = 0; ci < CIB; ++ci) {
    s0[ci] = 0.f; s1[ci] = 0.f;
    const int ci_g = cib * CIB + ci;
    if (ci_g < a.Cin_g && ok) {
      const int c = g * a.Cin_g + ci_g;
      T* yp = (T*)a.p.y + n * a.d.y_bs + (long long)c * dhw + q;
      float o = acc[ci];
      if (a.d.epi == 1) {
        const float esc = a.p.e_sc[n * a.d.Cin + c], esh = a.p.e_sh[n * a.d.Cin + c];
        const T* ep = (c < a.d.Cea ? (const T*)a.p.ea + n * a.d.ea_bs + (long long)c * dhw
                                   : (const T*)a.p.eb + n * a.d.eb_bs + (long long)(c - a.d.Cea) * dhw);
        const float ev = ldf(ep, q);
        o = rnd_as(yp, o * ((ev * esc + esh) > 0.f ? 1.f : a.d.e_slope));
        s0[ci] = o;
        s1[ci] = o * ev;
      }
      stf(yp, 0, o);
    }
  }
  if (a.d.epi == 1) {
    double v[2 * CIB];
#pragma unroll
    for (int i = 0; i < CIB; ++i) { v[2 * i] = (double)s0[i]; v[2 * i + 1] = (double)s1[i]; }
    block_sum_d<2 * CIB>(v, s_red, 4);
    if (tid < 2 * CIB) {
      const int ci_g = cib * CIB + (tid >> 1);
      if (ci_g < a.Cin_g) {
        const int c = g * a.Cin_g + ci_g;
        atomicAdd(&a.p.red[((long long)n * a.d.Cin + c) * 2 + (tid & 1)], s_red[tid]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient, k = 3 / 7.  A workgroup fixes (group, input channel, COB output channels[, kd plane for
// k=7]) and walks a slab of spatial tiles keeping NT*COB partial sums per lane in registers; one block
// reduction + fp32 atomics at the end.
// ---------------------------------------------------------------------------------------------------
struct WgradK {
  ConvK c;
  float* dw[XH_MAX_WPTR];
  float* db[XH_MAX_WPTR];
  int tiles_total;      // N * tiles per sample
  int tiles_per_block;
};

template <typename T, int K, int S, int COB, int TXN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradK wa) {
  const ConvK& a = wa.c;
  constexpr int VW = 4, TW = TXN * VW, TH = 8, TD = 32 / TXN;
  constexpr int KDN = (K == 3) ? 3 : 1;                 // kd planes handled per block
  constexpr int IW = (TW - 1) * S + K, IH = (TH - 1) * S + K, ID = (TD - 1) * S + KDN;
  constexpr int IWP = (IW + 3) / 4 * 4;
  constexpr int NT = KDN * K * K;
  constexpr int K3 = K * K * K;
  constexpr int ROWN = (VW - 1) * S + K;
  constexpr int PAD = K / 2;
  constexpr int NACC = NT * COB + COB;
  __shared__ __attribute__((aligned(16))) float s_in[ID * IH * IWP];
  __shared__ float s_red[4 * NACC];

  const int tid = threadIdx.x;
  const int tx = tid % TXN, ty = (tid / TXN) % TH, tz = tid / (TXN * TH);
  int yy = blockIdx.y;
  const int ci_g = yy % a.Cin_g; yy /= a.Cin_g;
  const int cob = yy % a.ncob;
  const int kd0 = (K == 3) ? 0 : yy / a.ncob;
  const int g = blockIdx.z;
  const int c_in = g * a.Cin_g + ci_g;
  const int D = a.d.D, H = a.d.H, W = a.d.W, Do = a.d.Do, Ho = a.d.Ho, Wo = a.d.Wo;
  const long long dhw = (long long)D * H * W, odhw = (long long)Do * Ho * Wo;
  const int tiles = a.tilesW * a.tilesH * a.tilesD;

  float acc[NT][COB];
  float dbacc[COB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int co = 0; co < COB; ++co) acc[t][co] = 0.f;
#pragma unroll
  for (int co = 0; co < COB; ++co) dbacc[co] = 0.f;

  const int t_begin = blockIdx.x * wa.tiles_per_block;
  const int t_end = min(wa.tiles_total, t_begin + wa.tiles_per_block);
  for (int tl = t_begin; tl < t_end; ++tl) {
    const int n = tl / tiles;
    int t = tl % tiles;
    const int tw = t % a.tilesW; t /= a.tilesW;
    const int th = t % a.tilesH;
    const int td = t / a.tilesH;
    const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
    const int id0 = od0 * S - PAD + kd0, ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
    __syncthreads();
    {
      const T* src = in_plane<T>(a, n, c_in, dhw);
      float sc = 1.f, sh = 0.f;
      if (a.d.pre) { sc = a.p.pre_sc[n * a.d.Cin + c_in]; sh = a.p.pre_sh[n * a.d.Cin + c_in]; }
      for (int idx = tid; idx < ID * IH * IWP; idx += 256) {
        const int wx = idx % IWP;
        const int r = idx / IWP;
        const int hy = r % IH, dz = r / IH;
        const int gd = id0 + dz, gh = ih0 + hy, gw = iw0 + wx;
        float v = 0.f;
        if (wx < IW && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W) {
          v = ldf(src, ((long long)gd * H + gh) * W + gw);
          if (a.d.pre) v = leaky(v * sc + sh, a.d.pre_slope);
        }
        s_in[idx] = v;
      }
    }
    __syncthreads();
    // this lane's dY values
    const int od = od0 + tz, oh = oh0 + ty, ow = ow0 + tx * VW;
    float dy[COB][VW];
#pragma unroll
    for (int co = 0; co < COB; ++co) {
      const int co_g = cob * COB + co;
#pragma unroll
      for (int v = 0; v < VW; ++v) dy[co][v] = 0.f;
      if (co_g < a.Cout_g && od < Do && oh < Ho) {
        const T* dp = (const T*)a.p.ea + n * a.d.ea_bs + (long long)(g * a.Cout_g + co_g) * odhw +
                      ((long long)od * Ho + oh) * Wo;
#pragma unroll
        for (int v = 0; v < VW; ++v)
          if (ow + v < Wo) dy[co][v] = ldf(dp, ow + v);
      }
#pragma unroll
      for (int v = 0; v < VW; ++v) dbacc[co] += dy[co][v];
    }
    const float* base = s_in + ((tz * S) * IH + ty * S) * IWP + tx * VW * S;
#pragma unroll
    for (int kd = 0; kd < KDN; ++kd) {
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const float* row = base + (kd * IH + kh) * IWP;
        float r[ROWN];
#pragma unroll
        for (int i = 0; i + 3 < ROWN; i += 4) {
          const float4 q = *reinterpret_cast<const float4*>(row + i);
          r[i] = q.x; r[i + 1] = q.y; r[i + 2] = q.z; r[i + 3] = q.w;
        }
#pragma unroll
        for (int i = ROWN / 4 * 4; i < ROWN; ++i) r[i] = row[i];
#pragma unroll
        for (int kw = 0; kw < K; ++kw)
#pragma unroll
          for (int co = 0; co < COB; ++co)
#pragma unroll
            for (int v = 0; v < VW; ++v)
              acc[(kd * K + kh) * K + kw][co] = fmaf(r[v * S + kw], dy[co][v], acc[(kd * K + kh) * K + kw][co]);
      }
    }
  }
  // ---- block reduction + atomics ----
  float v[NACC];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int co = 0; co < COB; ++co) v[t * COB + co] = acc[t][co];
#pragma unroll
  for (int co = 0; co < COB; ++co) v[NT * COB + co] = dbacc[co];
  __syncthreads();
  block_sum<NACC>(v, s_red, 4);
  const int gpp = a.d.groups / a.d.n_wptr;
  const int gl = g % gpp;
  for (int i = tid; i < NACC; i += 256) {
    const int co = i % COB;
    const int co_g = cob * COB + co;
    if (co_g >= a.Cout_g) continue;
    if (i < NT * COB) {
      const int tap = (K == 3) ? (i / COB) : (kd0 * K * K + i / COB);
      atomicAdd(&wa.dw[g / gpp][((long long)(gl * a.Cout_g + co_g) * a.Cin_g + ci_g) * K3 + tap], s_red[i]);
    } else if (ci_g == 0 && kd0 == 0 && wa.db[g / gpp]) {
      atomicAdd(&wa.db[g / gpp][gl * a.Cout_g + co_g], s_red[i]);
    }
  }
}

// k = 1 weight gradient: CIB x COB partial products per lane over a grid-strided range of 16-byte runs
// (4 fp32 / 8 bf16 voxels) when the layout allows it.
template <typename T, int CIB, int COB, bool VEC>
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const WgradK wa) {
  const ConvK& a = wa.c;
  constexpr int NACC = CIB * COB + COB;
  constexpr int VW = VEC ? VWT<T>::v : 1;              // 16-byte runs when the layout allows
  __shared__ float s_red[4 * NACC];
  const int tid = threadIdx.x;
  int yy = blockIdx.y;
  const int ncib = ((a.Cin_g + CIB - 1) / CIB);
  const int cib = yy % ncib;
  const int cob = yy / ncib;
  const int g = blockIdx.z;
  const long long dhw = (long long)a.d.D * a.d.H * a.d.W;
  float acc[CIB][COB], dbacc[COB];
#pragma unroll
  for (int i = 0; i < CIB; ++i)
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[i][j] = 0.f;
#pragma unroll
  for (int j = 0; j < COB; ++j) dbacc[j] = 0.f;
  // channels past the end are clamped (always valid addresses, all loads issue together) and masked arithmetically
  int cx[CIB], cy[COB];
  float mx[CIB], my[COB];
#pragma unroll
  for (int i = 0; i < CIB; ++i) {
    const int ci_g = cib * CIB + i;
    mx[i] = ci_g < a.Cin_g ? 1.f : 0.f;
    cx[i] = g * a.Cin_g + min(ci_g, a.Cin_g - 1);
  }
#pragma unroll
  for (int j = 0; j < COB; ++j) {
    const int co_g = cob * COB + j;
    my[j] = co_g < a.Cout_g ? 1.f : 0.f;
    cy[j] = g * a.Cout_g + min(co_g, a.Cout_g - 1);
  }
  const long long per_n = dhw / VW;
  const long long total = (long long)a.d.N * per_n;
  for (long long q = (long long)blockIdx.x * 256 + tid; q < total; q += (long long)gridDim.x * 256) {
    const int n = (int)(q / per_n);
    const long long sp = (q % per_n) * VW;
    float x[CIB][VW], dy[COB][VW];
#pragma unroll
    for (int i = 0; i < CIB; ++i) {
      const T* src = in_plane<T>(a, n, cx[i], dhw);
      if constexpr (VEC) ldvec(src, sp, x[i]); else x[i][0] = ldf(src, sp);
    }
#pragma unroll
    for (int j = 0; j < COB; ++j) {
      const T* dp = (const T*)a.p.ea + n * a.d.ea_bs + (long long)cy[j] * dhw;
      if constexpr (VEC) ldvec(dp, sp, dy[j]); else dy[j][0] = ldf(dp, sp);
    }
#pragma unroll
    for (int i = 0; i < CIB; ++i) {
      float sc = 1.f, sh = 0.f;
      if (a.d.pre) { sc = a.p.pre_sc[n * a.d.Cin + cx[i]]; sh = a.p.pre_sh[n * a.d.Cin + cx[i]]; }
#pragma unroll
      for (int v = 0; v < VW; ++v) {
        float xv = x[i][v];
        if (a.d.pre) xv = leaky(xv * sc + sh, a.d.pre_slope);
        x[i][v] = xv * mx[i];
      }
    }
#pragma unroll
    for (int j = 0; j < COB; ++j)
#pragma unroll
      for (int v = 0; v < VW; ++v) {
        dy[j][v] *= my[j];
        dbacc[j] += dy[j][v];
      }
#pragma unroll
    for (int i = 0; i < CIB; ++i)
#pragma unroll
      for (int j = 0; j < COB; ++j)
#pragma unroll
        for (int v = 0; v < VW; ++v) acc[i][j] = fmaf(x[i][v], dy[j][v], acc[i][j]);
  }
  float v[NACC];
#pragma unroll
  for (int i = 0; i < CIB; ++i)
#pragma unroll
    for (int j = 0; j < COB; ++j) v[i * COB + j] = acc[i][j];
#pragma unroll
  for (int j = 0; j < COB; ++j) v[CIB * COB + j] = dbacc[j];
  block_sum<NACC>(v, s_red, 4);
  const int gpp = a.d.groups / a.d.n_wptr;
  const int gl = g % gpp;
  if (tid < NACC) {
    if (tid < CIB * COB) {
      const int ci_g = cib * CIB + tid / COB, co_g = cob * COB + tid % COB;
      if (ci_g < a.Cin_g && co_g < a.Cout_g)
        atomicAdd(&wa.dw[g / gpp][(long long)(gl * a.Cout_g + co_g) * a.Cin_g + ci_g], s_red[tid]);
    } else {
      const int co_g = cob * COB + (tid - CIB * COB);
      if (cib == 0 && co_g < a.Cout_g && wa.db[g / gpp]) atomicAdd(&wa.db[g / gpp][gl * a.Cout_g + co_g], s_red[tid]);
    }
  }
}

// Several k = 1 weight gradients in ONE launch (xh_conv3d_wgrad_batch): the 1x1 problems of a training step are independent
// of each other, a dozen of them are deferred to the end of the backward pass, and each alone is a few hundred workgroups of
// 7 - 17 us (launch ramp, a short load chain, the atomics tail).  Workgroup b of the launch belongs to problem i with
// off[i] <= b < off[i + 1]; inside a problem the decomposition is conv1x1_wgrad_kernel's (4 x 4 channel tile, grid-strided
// 16-byte runs).  The table travels in the kernel arguments.
struct C1WP {
  const void *xa, *xb, *dy;
  const float *pre_sc, *pre_sh;
  float* dw[4];                                         // (k = 1 problems with more than 4 weight pointers run one by one)
  float* db[4];
  long long xa_bs, xb_bs, ea_bs, dhw;
  int N, Cin, Ca, Cin_g, Cout_g, groups, n_wptr, pre, gx, ny;
  float pre_slope;
  int dtype;
};
constexpr int C1W_MULTI = 20;
struct C1WMulti {
  int n;
  int off[C1W_MULTI + 1];
  C1WP p[C1W_MULTI];
};
static_assert(sizeof(C1WMulti) <= 3800, "kernel-argument table");

template <typename T>
__global__ __launch_bounds__(256) void conv1x1_wgrad_multi_kernel(const C1WMulti m) {
  constexpr int CIB = 4, COB = 4, NACC = CIB * COB + COB, VW = VWT<T>::v;
  __shared__ float s_red[4 * NACC];
  const int tid = threadIdx.x;
  int pi = 0;
  for (int k = 1; k < C1W_MULTI; ++k)
    if (k < m.n && (int)blockIdx.x >= m.off[k]) pi = k;
  const C1WP& a = m.p[pi];
  const int local = blockIdx.x - m.off[pi];
  const int gx = a.gx, ny = a.ny;
  const int bx = local % gx, r1 = local / gx;
  const int yy = r1 % ny, g = r1 / ny;
  const int ncib = (a.Cin_g + CIB - 1) / CIB;
  const int cib = yy % ncib, cob = yy / ncib;
  const long long dhw = a.dhw;
  float acc[CIB][COB], dbacc[COB];
#pragma unroll
  for (int i = 0; i < CIB; ++i)
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[i][j] = 0.f;
#pragma unroll
  for (int j = 0; j < COB; ++j) dbacc[j] = 0.f;
  // channels past the end are clamped (always valid addresses, all loads issue together) and masked arithmetically
  const T* xp[CIB]; const T* yp[COB];
  float mx[CIB], my[COB], sc[CIB], sh[CIB];
  long long xbs[CIB];
#pragma unroll
  for (int i = 0; i < CIB; ++i) {
    const int ci_g = cib * CIB + i;
    mx[i] = ci_g < a.Cin_g ? 1.f : 0.f;
    const int c = g * a.Cin_g + min(ci_g, a.Cin_g - 1);
    xp[i] = c < a.Ca ? (const T*)a.xa + (long long)c * dhw : (const T*)a.xb + (long long)(c - a.Ca) * dhw;
    xbs[i] = c < a.Ca ? a.xa_bs : a.xb_bs;
    sc[i] = 1.f; sh[i] = 0.f;
    if (a.pre && a.N == 1) { sc[i] = a.pre_sc[c]; sh[i] = a.pre_sh[c]; }
  }
#pragma unroll
  for (int j = 0; j < COB; ++j) {
    const int co_g = cob * COB + j;
    my[j] = co_g < a.Cout_g ? 1.f : 0.f;
    yp[j] = (const T*)a.dy + (long long)(g * a.Cout_g + min(co_g, a.Cout_g - 1)) * dhw;
  }
  const long long per_n = dhw / VW;
  const long long total = (long long)a.N * per_n;
  const float slope = a.pre_slope;
  for (long long q = (long long)bx * 256 + tid; q < total; q += (long long)gx * 256) {
    const int n = (int)(q / per_n);
    const long long sp = (q - n * per_n) * VW;
    float x[CIB][VW], dy[COB][VW];
#pragma unroll
    for (int i = 0; i < CIB; ++i) ldvec(xp[i] + n * xbs[i], sp, x[i]);
#pragma unroll
    for (int j = 0; j < COB; ++j) ldvec(yp[j] + n * a.ea_bs, sp, dy[j]);
#pragma unroll
    for (int i = 0; i < CIB; ++i) {
      if (a.pre && a.N > 1) {
        const int c = g * a.Cin_g + min(cib * CIB + i, a.Cin_g - 1);
        sc[i] = a.pre_sc[n * a.Cin + c]; sh[i] = a.pre_sh[n * a.Cin + c];
      }
#pragma unroll
      for (int v = 0; v < VW; ++v) {
        float xv = x[i][v];
        if (a.pre) xv = leaky(xv * sc[i] + sh[i], slope);
        x[i][v] = xv * mx[i];
      }
    }
#pragma unroll
    for (int j = 0; j < COB; ++j)
#pragma unroll
      for (int v = 0; v < VW; ++v) {
        dy[j][v] *= my[j];
        dbacc[j] += dy[j][v];
      }
#pragma unroll
    for (int i = 0; i < CIB; ++i)
#pragma unroll
      for (int j = 0; j < COB; ++j)
#pragma unroll
        for (int v = 0; v < VW; ++v) acc[i][j] = fmaf(x[i][v], dy[j][v], acc[i][j]);
  }
  float v[NACC];
#pragma unroll
  for (int i = 0; i < CIB; ++i)
#pragma unroll
    for (int j = 0; j < COB; ++j) v[i * COB + j] = acc[i][j];
#pragma unroll
  for (int j = 0; j < COB; ++j) v[CIB * COB + j] = dbacc[j];
  block_sum<NACC>(v, s_red, 4);
  const int gpp = a.groups / a.n_wptr;
  const int gl = g % gpp;
  if (tid < NACC) {
    if (tid < CIB * COB) {
      const int ci_g = cib * CIB + tid / COB, co_g = cob * COB + tid % COB;
      if (ci_g < a.Cin_g && co_g < a.Cout_g)
        atomicAdd(&a.dw[g / gpp][(long long)(gl * a.Cout_g + co_g) * a.Cin_g + ci_g], s_red[tid]);
    } else {
      const int co_g = cob * COB + (tid - CIB * COB);
      if (cib == 0 && co_g < a.Cout_g && a.db[g / gpp]) atomicAdd(&a.db[g / gpp][gl * a.Cout_g + co_g], s_red[tid]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static int check_desc(const xh_conv_desc* d, const xh_conv_ptrs* p) {
  if (!d || !p) return XH_ERR_ARG;
  if (d->dtype != XH_F32 && d->dtype != XH_BF16 && d->dtype != XH_F16) return XH_ERR_DTYPE;
  if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->groups <= 0) return XH_ERR_ARG;
  if (d->Cin % d->groups || d->Cout % d->groups) return XH_ERR_ARG;
  if (!(d->k == 1 || d->k == 3 || d->k == 7)) return XH_ERR_ARG;
  if (!(d->stride == 1 || (d->stride == 2 && d->k == 3))) return XH_ERR_ARG;
  const int pad = d->k / 2;
  if (d->Do != (d->D + 2 * pad - d->k) / d->stride + 1) return XH_ERR_ARG;
  if (d->Ho != (d->H + 2 * pad - d->k) / d->stride + 1) return XH_ERR_ARG;
  if (d->Wo != (d->W + 2 * pad - d->k) / d->stride + 1) return XH_ERR_ARG;
  if (d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Do <= 0 || d->Ho <= 0 || d->Wo <= 0) return XH_ERR_ARG;
  if (d->Ca < 0 || d->Ca > d->Cin) return XH_ERR_ARG;
  if (!p->xa || (d->Ca < d->Cin && !p->xb)) return XH_ERR_ARG;
  if (!(d->n_wptr == 1 || (d->n_wptr == d->groups && d->groups <= XH_MAX_WPTR))) return XH_ERR_ARG;
  for (int i = 0; i < d->n_wptr; ++i)
    if (!p->w[i]) return XH_ERR_ARG;
  if (d->pre < 0 || d->pre > 2) return XH_ERR_ARG;
  if (d->pre == 1 && (!p->pre_sc || !p->pre_sh)) return XH_ERR_ARG;
  if (d->pre == 2 && (!p->px || !p->nb_red || !p->nb_mean || !p->nb_rstd || p->nb_count <= 0)) return XH_ERR_ARG;
  if (d->N * d->groups > 65535) return XH_ERR_ARG;
  return XH_OK;
}

// The statistics fan-in block (fanin.h) is the caller's (xh_conv_ptrs.fan): zero on entry, left zero by the launch.
extern "C" long long xh_fanin_bytes(void) { return FAN_BLOCK_BYTES; }
unsigned char* xh_fan_block(void* fan, long long fan_bytes, long long units, long long wgs) {
  if (!fan || fan_bytes < FAN_BLOCK_BYTES || ((unsigned long long)fan & 127)) return nullptr;
  if (units < 1 || units > FAN_UNITS || wgs < FAN_MIN_WGS || (g_xh_disable & 64)) return nullptr;
  return (unsigned char*)fan;
}

static ConvK make_k(const xh_conv_desc* d, const xh_conv_ptrs* p, int cob, int txn) {
  ConvK a;
  a.d = *d;
  a.p = *p;
  a.Cin_g = d->Cin / d->groups;
  a.Cout_g = d->Cout / d->groups;
  a.ncob = cdiv(a.Cout_g, cob);
  a.tilesW = cdiv(d->Wo, 4 * txn);
  a.tilesH = cdiv(d->Ho, 8);
  a.tilesD = cdiv(d->Do, 32 / txn);
  a.fan = nullptr;
  return a;
}
static int pick_txn(int Wo) { return Wo > 16 ? 8 : (Wo > 8 ? 4 : 2); }
static int pick_cob(int cout_g, int maxc) {
  int c = 1;
  while (c < cout_g && c < maxc) c <<= 1;
  return c;
}

template <typename T> static const char* tname() { return FmtOf<T>::v == 0 ? "bf16_t" : FmtOf<T>::v == 1 ? "f16_t" : "float"; }
#define LAUNCH_FWD(T, K, S, COB, TXN)                                                                              \
  do {                                                                                                             \
    xh_note_kernel("conv_fwd_kernel<%s, %d, %d, %d, %d>", tname<T>(), K, S, COB, TXN);                            \
    hipLaunchKernelGGL((conv_fwd_kernel<T, K, S, COB, TXN>), grid, dim3(256), 0, (hipStream_t)stream, a);          \
  } while (0)
#define FWD_TXN(T, K, S, COB)                                        \
  do {                                                               \
    if (txn == 8) { LAUNCH_FWD(T, K, S, COB, 8); }                   \
    else if (txn == 4) { LAUNCH_FWD(T, K, S, COB, 4); }              \
    else { LAUNCH_FWD(T, K, S, COB, 2); }                            \
  } while (0)

// Launch-plan tunables of the small streaming kernels (xh_set_option keys 6..8; 0 = the built-in rule, which is what
// tools/microbench_small.py measured best at this network's shapes)
static int g_dw_minsd = 0;     // key 6: fewest planes a depthwise sliding-window workgroup marches through
static int g_dw_target = 1024; // key 7: workgroup count the depth split of the sliding-window kernels aims at
static int g_c1w_wgs = 320;    // key 8: workgroup count the k = 1 weight gradient aims at
static int g_c1_cap = 0;       // key 10: workgroup cap of the k = 1 forward kernel (0 = the built-in rule)
static int g_wg_slabs = 512;   // key 13: workgroup target of the vector weight-gradient kernel (1 -> 2 k3 @128^3: 75 us with 2048, 40 us with 512)
static int g_s2w_cap = 512;    // key 12: workgroup cap of the vectorised stride-2 weight gradient (4096: 98 us at 128^3, 512: 41 us -- each workgroup
                               // ends with a 56-value block reduction and 54 atomics on addresses shared by the whole (channel, group))

// Planes per sliding-window segment: at 128^3 the kernels are issue-bound and the two halo planes per segment cost more
// than the extra workgroups return (8 planes); at 64^3 / 32^3 the launch is a latency chain of one load per plane over
// too few waves, and shorter segments win (64^3: 22 -> 16 us with 4; 32^3: 22 -> 11 us with 2).
static int dw_min_planes(long long dhw) {
  if (g_dw_minsd > 0) return g_dw_minsd;
  return dhw >= (1 << 21) ? 8 : dhw >= (1 << 18) ? 4 : 2;
}

template <typename T>
static int conv_fwd_dispatch(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p) {
  const int cout_g = d->Cout / d->groups, cin_g = d->Cin / d->groups;
  if (d->k == 1) {
    if (cin_g > 128) return XH_ERR_ARG;
    const int cob = pick_cob(cout_g, 8);
    ConvK a = make_k(d, p, cob, 8);
    const long long dhw = (long long)d->D * d->H * d->W;
    constexpr int VW1 = VWT<T>::v;
    const bool vec = (dhw % VW1 == 0) && (d->xa_bs % VW1 == 0) && (d->xb_bs % VW1 == 0) && (d->y_bs % VW1 == 0) &&
                     (d->epi != 1 || (d->ea_bs % VW1 == 0 && d->eb_bs % VW1 == 0));
    // compile-time epilogue for the common launches: no activation (mode = epi), or the sigmoid head (mode 3)
    const int mode = d->act == XH_ACT_NONE ? d->epi : (d->act == XH_ACT_SIGMOID && d->epi == 0) ? 3 : -1;
    long long gx1 = (dhw + 256 * VW1 - 1) / (256 * VW1);
    // few lanes (the deep levels): the run time is the per-lane chain of Cin/CIC dependent load steps, so take narrow
    // output blocks (more workgroups) with 16 channels in flight per step
    if (vec && cin_g >= 16 && cout_g >= 2 && gx1 * a.ncob * d->N * d->groups < 128) {
      a = make_k(d, p, 2, 8);
      dim3 grid((unsigned)gx1, a.ncob, d->N * d->groups);
      if (d->epi) a.fan = xh_fan_block(p->fan, p->fan_bytes, (long long)grid.y * grid.z, grid.x);
      xh_note_kernel("conv1x1_kernel<%s, 2, true, 16>", tname<T>());
      switch (mode) {
        case 0: hipLaunchKernelGGL((conv1x1_kernel<T, 2, true, 16, 0>), grid, dim3(256), 0, (hipStream_t)stream, a); break;
        case 1: hipLaunchKernelGGL((conv1x1_kernel<T, 2, true, 16, 1>), grid, dim3(256), 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL((conv1x1_kernel<T, 2, true, 16, 2>), grid, dim3(256), 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL((conv1x1_kernel<T, 2, true, 16, -1>), grid, dim3(256), 0, (hipStream_t)stream, a);
      }
      return xh_launch_status();
    }
    const long long cap1 = cdiv(g_c1_cap > 0 ? g_c1_cap : 2048, a.ncob * d->N * d->groups);
    if (gx1 > cap1) gx1 = cap1;
    dim3 grid((unsigned)gx1, a.ncob, d->N * d->groups);
    if (d->epi) a.fan = xh_fan_block(p->fan, p->fan_bytes, (long long)grid.y * grid.z, grid.x);
#define L1M(COB, M) hipLaunchKernelGGL((conv1x1_kernel<T, COB, true, 4, M>), grid, dim3(256), 0, (hipStream_t)stream, a)
#define L1(COB)                                                                                                  \
  do {                                                                                                           \
    xh_note_kernel("conv1x1_kernel<%s, %d, %s>", tname<T>(), COB, vec ? "true" : "false");                       \
    if (!vec) hipLaunchKernelGGL((conv1x1_kernel<T, COB, false>), grid, dim3(256), 0, (hipStream_t)stream, a);   \
    else if (mode == 0) L1M(COB, 0);                                                                             \
    else if (mode == 1) L1M(COB, 1);                                                                             \
    else if (mode == 2) L1M(COB, 2);                                                                             \
    else if (mode == 3) L1M(COB, 3);                                                                             \
    else L1M(COB, -1);                                                                                           \
  } while (0)
    switch (cob) { case 1: L1(1); break; case 2: L1(2); break; case 4: L1(4); break; default: L1(8); }
#undef L1
#undef L1M
    return xh_launch_status();
  }
  const int txn = pick_txn(d->Wo);
  if (d->k == 3 && d->stride == 1 && cin_g == 1 && cout_g == 1 && d->Cin < 65536) {
    ConvK a = make_k(d, p, 1, 8);
    {
      constexpr int VW = VWT<T>::v;
      const long long dhw3 = (long long)d->D * d->H * d->W;
      const bool al = d->W % VW == 0 && dhw3 % VW == 0 && d->xa_bs % VW == 0 && d->xb_bs % VW == 0 && d->y_bs % VW == 0 &&
                      (d->epi != 1 || (d->ea_bs % VW == 0 && d->eb_bs % VW == 0));
      if (al && d->H >= 32 && d->W >= 4 * VW && d->D >= 8 && !(g_xh_disable & 1)) {
        int txn = 4;
        while (txn < 32 && txn * VW < d->W) txn *= 2;
        a.tilesW = cdiv(d->W, txn * VW);
        a.tilesH = cdiv(d->H, 256 / txn);
        const int base = a.tilesW * a.tilesH * d->Cin * d->N;
        int dsegs = cdiv(g_dw_target, base);
        if (dsegs > d->D / dw_min_planes(dhw3)) dsegs = d->D / dw_min_planes(dhw3);
        if (dsegs < 1) dsegs = 1;
        const int sd = cdiv(d->D, dsegs);
        dsegs = cdiv(d->D, sd);
        dim3 grid(a.tilesW * a.tilesH * dsegs, d->Cin, d->N);
        if (d->epi) a.fan = xh_fan_block(p->fan, p->fan_bytes, (long long)grid.y * grid.z, grid.x);
        xh_note_kernel("conv_dw3_slide_kernel<%s, %d>", tname<T>(), txn);
        switch (txn) {
          case 4: hipLaunchKernelGGL((conv_dw3_slide_kernel<T, 4>), grid, dim3(256), 0, (hipStream_t)stream, a, sd); break;
          case 8: hipLaunchKernelGGL((conv_dw3_slide_kernel<T, 8>), grid, dim3(256), 0, (hipStream_t)stream, a, sd); break;
          case 16: hipLaunchKernelGGL((conv_dw3_slide_kernel<T, 16>), grid, dim3(256), 0, (hipStream_t)stream, a, sd); break;
          default: hipLaunchKernelGGL((conv_dw3_slide_kernel<T, 32>), grid, dim3(256), 0, (hipStream_t)stream, a, sd);
        }
        return xh_launch_status();
      }
    }
    int gx = a.tilesW * a.tilesH * a.tilesD;
    const int cap = cdiv(2048, d->Cin * d->N);
    if (gx > cap) gx = cap;
    dim3 grid(gx, d->Cin, d->N);
    xh_note_kernel("conv_dw3_kernel<%s>", tname<T>());
    hipLaunchKernelGGL((conv_dw3_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, a);
    return xh_launch_status();
  }
  if (d->k == 3 && d->stride == 1) {
    int cob = pick_cob(cout_g, 8);
    ConvK a = make_k(d, p, cob, txn);
    // small volumes (the 16^3 / 8^3 levels in fp32 storage): a few dozen workgroups whose threads each walk Cin_g x 27 x COB FMAs
    // -- 80 -> 80 g5 @16^3 ran 108 us on 40 workgroups.  Narrower output-channel blocks: COB times the workgroups, 1 / COB the chain
    // (a voxel's accumulation order does not depend on the block: same bits)
    while (cob > 1 && (long long)a.tilesW * a.tilesH * a.tilesD * a.ncob * d->N * d->groups < 256) {
      cob >>= 1;
      a = make_k(d, p, cob, txn);
    }
    dim3 grid(a.tilesW * a.tilesH * a.tilesD, a.ncob, d->N * d->groups);
    switch (cob) {
      case 1: FWD_TXN(T, 3, 1, 1); break;
      case 2: FWD_TXN(T, 3, 1, 2); break;
      case 4: FWD_TXN(T, 3, 1, 4); break;
      default: FWD_TXN(T, 3, 1, 8);
    }
    return xh_launch_status();
  }
  if (d->k == 3 && d->stride == 2) {
    if (d->transposed) return XH_ERR_ARG;
    int cob = pick_cob(cout_g, 8) < 2 ? 2 : pick_cob(cout_g, 8);
    // small outputs (the deep DRB levels): few lanes, so the per-lane chain cin_g x 27 x COB is the run time -- use
    // narrow channel blocks to get 4x more workgroups with 4x shorter chains
    if ((long long)d->Do * d->Ho * d->Wo * d->N * d->groups <= (1 << 16) && cob > 2) cob = 2;
    ConvK a = make_k(d, p, cob, txn);
    const long long odhw = (long long)d->Do * d->Ho * d->Wo;
    dim3 grid((unsigned)((odhw + 255) / 256), a.ncob, d->N * d->groups);
    const size_t shm = ((size_t)cin_g * 27 * cob + 2 + 2 * 4 * 2 * cob) * sizeof(float);
    if (shm > 60 * 1024) return XH_ERR_ARG;
    {
      constexpr int VW = VWT<T>::v;
      const int lw = d->W / VW;
      const long long dhw2 = (long long)d->D * d->H * d->W;
      const bool al = d->W % VW == 0 && d->Wo * 2 == d->W && lw >= 1 && lw <= 64 && (64 % lw) == 0 && dhw2 % VW == 0 &&
                      d->xa_bs % VW == 0 && d->xb_bs % VW == 0 && !(g_xh_disable & 4);
      if (al) {
        const long long lanes = (long long)d->Do * d->Ho * lw;
        // input-channel split inside the block while the launch has fewer than 256 workgroups (narrow channel blocks only)
        int ks = 1;
        if (cob <= 4 && !(g_xh_disable & 256))
          while (ks < 8 && cdiv((int)lanes, 256 / ks) * a.ncob * d->N * d->groups < 256 && 256 / (2 * ks) >= lw &&
                 (256 / (2 * ks)) % lw == 0 && cin_g / (2 * ks) >= 2)
            ks *= 2;
        const int lpb = 256 / ks;
        const int part_off = (int)((shm / sizeof(float) + 1) & ~(size_t)1);
        size_t shm2 = ks > 1 ? (size_t)part_off * sizeof(float) + (size_t)(ks - 1) * lpb * cob * (VW / 2) * sizeof(float) : shm;
        if (p->fin_red && sizeof(T) != 2) {             // fp32 storage: the finalisation of xh_norm_finalize, bit for bit
          const int rc = xh_norm_finalize(stream, 0, p->fin_red, d->N, d->Cin, p->fin_count, 1, 1e-5f, nullptr, nullptr, nullptr, nullptr, 1,
                                          const_cast<float*>(p->pre_sc), const_cast<float*>(p->pre_sh), p->fin_mean, p->fin_rstd);
          if (rc) return rc;
          a.p.fin_red = nullptr;
        }
        const int fin_off = (int)((shm2 / sizeof(float) + 1) & ~(size_t)1);
        if (a.p.fin_red) shm2 = (size_t)fin_off * sizeof(float) + (size_t)2 * cin_g * sizeof(float);
        dim3 gridv((unsigned)((lanes + lpb - 1) / lpb), a.ncob, d->N * d->groups);
        xh_note_kernel("conv3_s2_vec_kernel<%s, %d>", tname<T>(), cob);
        switch (cob) {
          case 2: hipLaunchKernelGGL((conv3_s2_vec_kernel<T, 2>), gridv, dim3(256), shm2, (hipStream_t)stream, a, lw, ks, part_off, fin_off); break;
          case 4: hipLaunchKernelGGL((conv3_s2_vec_kernel<T, 4>), gridv, dim3(256), shm2, (hipStream_t)stream, a, lw, ks, part_off, fin_off); break;
          default: hipLaunchKernelGGL((conv3_s2_vec_kernel<T, 8>), gridv, dim3(256), shm2, (hipStream_t)stream, a, lw, ks, part_off, fin_off);
        }
        return xh_launch_status();
      }
    }
    if (p->fin_red) {                                   // the gather kernel takes finished scale / shift
      const int rc = xh_norm_finalize(stream, 0, p->fin_red, d->N, d->Cin, p->fin_count, 1, 1e-5f, nullptr, nullptr, nullptr, nullptr, 1,
                                      const_cast<float*>(p->pre_sc), const_cast<float*>(p->pre_sh), p->fin_mean, p->fin_rstd);
      if (rc) return rc;
      a.p.fin_red = nullptr;
    }
    xh_note_kernel("conv3_s2_gather_kernel<%s, %d>", tname<T>(), cob);
    switch (cob) {
      case 2: hipLaunchKernelGGL((conv3_s2_gather_kernel<T, 2>), grid, dim3(256), shm, (hipStream_t)stream, a); break;
      case 4: hipLaunchKernelGGL((conv3_s2_gather_kernel<T, 4>), grid, dim3(256), shm, (hipStream_t)stream, a); break;
      default: hipLaunchKernelGGL((conv3_s2_gather_kernel<T, 8>), grid, dim3(256), shm, (hipStream_t)stream, a);
    }
    return xh_launch_status();
  }
  if (d->k == 7) {
    const int cob = pick_cob(cout_g, 4) < 2 ? 2 : pick_cob(cout_g, 4);
    ConvK a = make_k(d, p, cob, txn);
    dim3 grid(a.tilesW * a.tilesH * a.tilesD, a.ncob, d->N * d->groups);
    switch (cob) {
      case 2: FWD_TXN(T, 7, 1, 2); break;
      default: FWD_TXN(T, 7, 1, 4);
    }
    return xh_launch_status();
  }
  return XH_ERR_ARG;
}

int xh_conv3_mfma_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p);   // conv3d_mfma.hip
int xh_conv7_mfma_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p);   // conv7_mfma.hip
int xh_conv7_wgrad_mfma_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR]);
int xh_conv3_wgrad_mfma_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR]);
int g_use_mfma = 1;
int g_xh_disable = 0;
static char g_last_kernel[96] = "";
void xh_note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
  va_end(ap);
}
extern "C" const char* xh_last_conv_kernel(void) { return g_last_kernel; }
extern int g_mfma_abl;
// Launch plan of a k = 1 convolution (shared by the single launch in conv_fwd_dispatch and xh_conv1x1_multi): output-channel
// block, channels per step, grid.  false: the layout does not allow 16-byte runs (the single launch then takes its scalar instance).
template <typename T>
static bool c1_plan(const xh_conv_desc* d, const xh_conv_ptrs* p, ConvK& a, int& cob, int& cic, dim3& grid) {
  const int cout_g = d->Cout / d->groups, cin_g = d->Cin / d->groups;
  cob = pick_cob(cout_g, 8);
  cic = 4;
  a = make_k(d, p, cob, 8);
  const long long dhw = (long long)d->D * d->H * d->W;
  constexpr int VW1 = VWT<T>::v;
  const bool vec = (dhw % VW1 == 0) && (d->xa_bs % VW1 == 0) && (d->xb_bs % VW1 == 0) && (d->y_bs % VW1 == 0) &&
                   (d->epi != 1 || (d->ea_bs % VW1 == 0 && d->eb_bs % VW1 == 0));
  long long gx1 = (dhw + 256 * VW1 - 1) / (256 * VW1);
  if (vec && cin_g >= 16 && cout_g >= 2 && gx1 * a.ncob * d->N * d->groups < 128) {
    cob = 2; cic = 16;
    a = make_k(d, p, 2, 8);
    grid = dim3((unsigned)gx1, a.ncob, d->N * d->groups);
    return true;
  }
  const long long cap1 = cdiv(g_c1_cap > 0 ? g_c1_cap : 2048, a.ncob * d->N * d->groups);
  if (gx1 > cap1) gx1 = cap1;
  grid = dim3((unsigned)gx1, a.ncob, d->N * d->groups);
  return vec;
}
static int check_desc(const xh_conv_desc* d, const xh_conv_ptrs* p);
extern "C" int xh_conv1x1_multi(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p) {
  if (n < 1 || n > XH_LEVELS_MAX || !d || !p) return XH_ERR_ARG;
  C1Multi* m = new C1Multi;
  m->n = n;
  m->off[0] = 0;
  const int dtype = d[0] ? d[0]->dtype : -1, epi = d[0] ? d[0]->epi : -1;
  for (int i = 0; i < n; ++i) {
    const int rc = check_desc(d[i], p[i]);
    if (rc) { delete m; return rc; }
    if (!p[i]->y || (d[i]->epi == 2 && !p[i]->red)) { delete m; return XH_ERR_ARG; }
    // one storage type, one epilogue (none | output moments), no activation, no cin > 128: else the caller launches them one by one
    if (d[i]->k != 1 || d[i]->dtype != dtype || d[i]->epi != epi || (epi != 0 && epi != 2) || d[i]->act != XH_ACT_NONE ||
        d[i]->Cin / d[i]->groups > 128 || d[i]->pre == 2) { delete m; return 1; }
    dim3 g;
    bool vec = false;
    XH_DISPATCH_T(dtype, vec = c1_plan<T>(d[i], p[i], m->p[i], m->cob[i], m->cic[i], g););
    if (!vec) { delete m; return 1; }
    m->gx[i] = (int)g.x; m->gy[i] = (int)g.y; m->gz[i] = (int)g.z;
    m->off[i + 1] = m->off[i] + (int)(g.x * g.y * g.z);
  }
  xh_note_kernel("conv1x1_multi_kernel<%d problems, epi %d>", n, epi);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(m->off[n]);
  XH_DISPATCH_T(dtype, {
    if (epi == 2) hipLaunchKernelGGL((conv1x1_multi_kernel<T, 2>), grid, dim3(256), 0, st, *m);
    else hipLaunchKernelGGL((conv1x1_multi_kernel<T, 0>), grid, dim3(256), 0, st, *m);
  });
  delete m;
  return xh_launch_status();
}

extern "C" int xh_set_option(int key, int value) {
  if (key == 0) { g_use_mfma = value; return XH_OK; }
  if (key == 1) { g_mfma_abl = value; return XH_OK; }
  if (key == 2) { g_xh_disable = value; return XH_OK; }
  if (key == 3) { extern int g_mfma_wgs; g_mfma_wgs = value > 0 ? value : 512; return XH_OK; }
  if (key == 4) { extern int g_mfma_occ; g_mfma_occ = value; return XH_OK; }
  if (key == 6) { g_dw_minsd = value < 0 ? 0 : value; return XH_OK; }
  if (key == 7) { g_dw_target = value < 1 ? 1 : value; return XH_OK; }
  if (key == 8) { g_c1w_wgs = value < 1 ? 320 : value; return XH_OK; }
  if (key == 10) { g_c1_cap = value < 0 ? 0 : value; return XH_OK; }
  if (key == 12) { g_s2w_cap = value < 16 ? 16 : value; return XH_OK; }
  if (key == 13) { g_wg_slabs = value < 16 ? 16 : value; return XH_OK; }
  if (key == 11) { extern int g_q4_maxc; g_q4_maxc = value < 4 ? 4 : value > 48 ? 48 : value; return XH_OK; }
  if (key == 9) { extern int g_red_wgs; g_red_wgs = value < 0 ? 0 : value; return XH_OK; }
  if (key == 5) { extern int g_dconv_kq; g_dconv_kq = value == 1 ? 1 : 2; return XH_OK; }
  if (key == 14) { extern int g_dconv_cfg; g_dconv_cfg = value; return XH_OK; }
  if (key == 15) { extern int g_dconv_big; g_dconv_big = value < 1 ? 1 : value; return XH_OK; }
  if (key == 17) { extern int g_q4_wgs; g_q4_wgs = value < 0 ? 0 : value; return XH_OK; }
  if (key == 20) { extern int g_q4_wide; g_q4_wide = value & 7; return XH_OK; }
  if (key == 19) { extern int g_q4_persist; g_q4_persist = value < 0 ? 0 : value; return XH_OK; }
  if (key == 16) { extern int g_tiny_wgs; g_tiny_wgs = value < 1 ? 1 : value; return XH_OK; }
  if (key == 21) { extern int g_q5_on; g_q5_on = value ? 1 : 0; return XH_OK; }
  if (key == 27) { extern int g_row_wgs; if (value < 64 || value > (1 << 20)) return XH_ERR_ARG; g_row_wgs = value; return XH_OK; }
  if (key == 26) { extern int g_dwh_groups; if (value < 1 || value > 4096) return XH_ERR_ARG; g_dwh_groups = value; return XH_OK; }
  if (key == 24) { extern int g_c7_as; g_c7_as = value < 0 ? 0 : value > 3 ? 3 : value; return XH_OK; }
  if (key == 23) { extern int g_q5_w32; g_q5_w32 = value ? 1 : 0; return XH_OK; }
  if (key == 22) { extern int g_q5_wgs; g_q5_wgs = value < 8 ? 8 : value; return XH_OK; }
  if (key == 28) { extern int g_q5_uq; g_q5_uq = value & 15; return XH_OK; }
  return XH_ERR_ARG;
}

int xh_conv3_q4_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p);          // conv3d_q4.hip
long long xh_conv3_q4_workspace_bytes(const xh_conv_desc* d);
extern "C" int xh_conv3d_fuses_bn_finalize(const xh_conv_desc* d) {
  if (!d || !g_use_mfma || d->pre != 1 || d->N != 1 || d->k != 3) return 0;
  return xh_conv3_q4_workspace_bytes(d) > 0 ? 1 : 0;
}
extern "C" int xh_conv3d_supports_bcast(const xh_conv_desc* d) {
  if (!d || !g_use_mfma || d->bcast != 4 || d->transposed || d->k != 3 || d->stride != 1 || (d->dtype != XH_BF16 && d->dtype != XH_F16)) return 0;
  if (d->groups <= 0 || d->Cin != 4 * d->groups || d->Cout != 4 * d->groups || d->Ca != d->Cin) return 0;   // 4 -> 4 per group (the data gradient's e is the input)
  if ((d->W != 128 && d->W != 64) || d->H % 8 || d->H < 8 || d->D < 4) return 0;
  const long long dhw = (long long)d->D * d->H * d->W;
  if (4 * dhw >= (1ll << 27)) return 0;
  return xh_conv3_q4_workspace_bytes(d) > 0 ? 1 : 0;
}
extern "C" int xh_conv3d_fuses_norm_bwd(const xh_conv_desc* d) {
  if (!d || !g_use_mfma || d->epi == 2 || d->act != XH_ACT_NONE) return 0;
  xh_conv_desc t = *d;
  t.pre = 2;
  return xh_conv3_q4_workspace_bytes(&t) > 0 ? 1 : 0;
}
int xh_conv3_q4_pair_try(void* stream, const xh_conv_desc* d0, const xh_conv_ptrs* p0, const xh_conv_desc* d1, const xh_conv_ptrs* p1);
// Two independent k = 3 stride-1 convolutions of one shape in one launch; returns 1 (nothing launched) when they are not such a pair.
extern "C" int xh_conv3d_fwd_pair(void* stream, const xh_conv_desc* d0, const xh_conv_ptrs* p0, const xh_conv_desc* d1, const xh_conv_ptrs* p1) {
  const xh_conv_desc* ds[2] = {d0, d1};
  const xh_conv_ptrs* ps[2] = {p0, p1};
  for (int i = 0; i < 2; ++i) {
    const int rc = check_desc(ds[i], ps[i]);
    if (rc) return rc;
    const xh_conv_desc* d = ds[i];
    const xh_conv_ptrs* p = ps[i];
    if (!p->y || d->k != 3 || d->stride != 1 || d->bcast || d->pre == 2 || p->fin_gamma) return 1;
    if (d->epi == 1 && (!p->ea || !p->e_sc || !p->e_sh || !p->red || (d->Cea < d->Cout && !p->eb))) return XH_ERR_ARG;
    if (d->epi == 2 && !p->red) return XH_ERR_ARG;
    if (d->epi < 0 || d->epi > 2) return XH_ERR_ARG;
    if (p->fin_red && (d->pre != 1 || !p->fin_mean || !p->fin_rstd || p->fin_count <= 0)) return XH_ERR_ARG;
  }
  if (!g_use_mfma) return 1;
  return xh_conv3_q4_pair_try(stream, d0, p0, d1, p1);
}
extern "C" int xh_conv3d_fwd(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p) {
  int rc = check_desc(d, p);
  if (rc) return rc;
  if (!p->y && !(d->bcast && d->transposed && d->epi == 1)) return XH_ERR_ARG;
  if (d->transposed && d->stride != 1) return XH_ERR_ARG;
  if (d->bcast) {                                      // broadcast operand: the full-row quad-channel kernel or nothing
    if (d->bcast != 4 || d->Ca != d->Cin || p->fin_red || d->pre == 2) return XH_ERR_ARG;
    if (d->epi == 1 && (!p->ea || !p->e_sc || !p->e_sh || !p->red)) return XH_ERR_ARG;
    if (d->epi == 2 && !p->red) return XH_ERR_ARG;
    const int r = g_use_mfma ? xh_conv3_q4_try(stream, d, p) : 1;
    return r == 1 ? XH_ERR_ARG : r;
  }
  if (d->epi == 1 && (!p->ea || !p->e_sc || !p->e_sh || !p->red || (d->Cea < d->Cout && !p->eb))) return XH_ERR_ARG;
  if (d->epi == 2 && !p->red) return XH_ERR_ARG;
  if (d->epi < 0 || d->epi > 2) return XH_ERR_ARG;
  if (p->fin_red && (d->pre != 1 || !p->fin_mean || !p->fin_rstd || p->fin_count <= 0 || d->k != 3)) return XH_ERR_ARG;
  if (p->fin_gamma) {                                 // BatchNorm flavour of the fused finalisation: the quad-channel kernel, one sample
    if (!p->fin_red || !p->fin_beta || d->N != 1 || p->fin_steps < 0) return XH_ERR_ARG;
    const int r = g_use_mfma ? xh_conv3_q4_try(stream, d, p) : 1;
    return r == 1 ? XH_ERR_ARG : r;
  }
  if (d->pre == 2) {                                  // norm-backward input: the quad-channel kernel only (xh_conv3d_fuses_norm_bwd)
    const int r = g_use_mfma ? xh_conv3_q4_try(stream, d, p) : 1;
    return r == 1 ? XH_ERR_ARG : r;
  }
  if (g_use_mfma) {
    const int r = d->k == 7 ? xh_conv7_mfma_try(stream, d, p) : xh_conv3_mfma_try(stream, d, p);
    if (r != 1) return r;
  }
  // the fused finalisation exists on the MFMA path and in front of the stride-2 convs (which fall back to a finalisation launch
  // of their own where their kernel cannot take it)
  if (p->fin_red && !(d->k == 3 && d->stride == 2)) return XH_ERR_ARG;
  {
    const int r = xh_conv3_tiny_try(stream, d, p);    // conv3_tiny.hip: 1 <-> 2 channels
    if (r != 1) return r;
  }
  xh_note_kernel("conv k%d s%d (vector kernel family)", d->k, d->stride);
  XH_DISPATCH_T(d->dtype, return conv_fwd_dispatch<T>(stream, d, p););
}

__device__ __forceinline__ void ldhalf_c(const float* p, float (&o)[2]) {
  const float2 t = *reinterpret_cast<const float2*>(p); o[0] = t.x; o[1] = t.y;
}
template <int F> __device__ __forceinline__ void ldhalf_c(const h16<F>* p, float (&o)[4]) { ld4(p, 0, o); }

// ---------------------------------------------------------------------------------------------------
// k = 3, stride = 2 data gradient, vectorised.  A lane produces one 16-byte run of dX (VW voxels of one row, CIB input
// channels).  Along each axis an input voxel v receives tap k from output (v+1-k)/2 when v+1-k is even: an even voxel
// only tap 1, an odd voxel taps 0 and 2.  Workgroups are sorted by the (d, h) parity class of their rows, so the tap
// set is uniform per workgroup; along W the VW voxels of a lane need dY columns c0 .. c0+VW/2 (one VW/2-wide load +
// the next lane's first column by wave shuffle).
// ---------------------------------------------------------------------------------------------------
template <typename T, int CIB>
__global__ __launch_bounds__(256) void conv3_dgrad_s2_vec_kernel(const ConvK a, int LW, int rows_per_class) {
  constexpr int VW = VWT<T>::v, OW = VW / 2;
  extern __shared__ float s_dyn[];   // [Cout_g][CIB][27] weights + reduction scratch
  float* s_w = s_dyn;
  double* s_red = reinterpret_cast<double*>(s_dyn + ((a.Cout_g * CIB * 27 + 1) & ~1));
  const int tid = threadIdx.x;
  const int cib = blockIdx.y;
  const int n = blockIdx.z / a.d.groups, g = blockIdx.z % a.d.groups;
  const int gpp = a.d.groups / a.d.n_wptr;
  const float* wp = a.p.w[g / gpp];
  const int gl = g % gpp;
  for (int idx = tid; idx < a.Cout_g * CIB * 27; idx += 256) {
    const int tap = idx % 27;
    const int r = idx / 27;
    const int ci = r % CIB, co_g = r / CIB;
    const int ci_g = cib * CIB + ci;
    s_w[idx] = ci_g < a.Cin_g ? wp[((long long)(gl * a.Cout_g + co_g) * a.Cin_g + ci_g) * 27 + tap] : 0.f;
  }
  __syncthreads();
  const int D = a.d.D, H = a.d.H, W = a.d.W, Do = a.d.Do, Ho = a.d.Ho, Wo = a.d.Wo;
  const long long dhw = (long long)D * H * W, odhw = (long long)Do * Ho * Wo;
  const int blocks_per_class = gridDim.x / 4;
  const int cls = blockIdx.x / blocks_per_class;                        // (d & 1) * 2 + (h & 1)
  const int pd = cls >> 1, ph = cls & 1;
  const long long lane_id = (long long)(blockIdx.x % blocks_per_class) * 256 + tid;
  const int tx = (int)(lane_id % LW);
  const long long row = lane_id / LW;
  const int HH = (H - ph + 1) / 2, DD = (D - pd + 1) / 2;               // rows of this parity along H and D
  const bool ok = row < (long long)DD * HH;
  const int hh = (int)(row % max(HH, 1)), dd = (int)min(row / max(HH, 1), (long long)max(DD - 1, 0));
  const int d_ = 2 * dd + pd, h_ = 2 * hh + ph;
  const int w0 = tx * VW, c0 = tx * OW;
  // taps per axis: even -> {1}; odd -> {0, 2}.  output index = (v + 1 - k) / 2
  const int nkd = pd ? 2 : 1, nkh = ph ? 2 : 1;
  float acc[CIB][VW];
#pragma unroll
  for (int i = 0; i < CIB; ++i)
#pragma unroll
    for (int v = 0; v < VW; ++v) acc[i][v] = 0.f;
  for (int td = 0; td < nkd; ++td) {                                    // block-uniform trip counts
    const int kd = pd ? 2 * td : 1;
    const int od = (d_ + 1 - kd) >> 1;
    const float md = (od >= 0 && od < Do) ? 1.f : 0.f;
    for (int th = 0; th < nkh; ++th) {
      const int kh = ph ? 2 * th : 1;
      const int oh = (h_ + 1 - kh) >> 1;
      const float m = (oh >= 0 && oh < Ho && ok) ? md : 0.f;
      const long long osp = ((long long)min(max(od, 0), Do - 1) * Ho + min(max(oh, 0), Ho - 1)) * Wo + c0;
      const int tap0 = (kd * 3 + kh) * 3;
      // the dY loads of U output channels are issued together: with one load per trip the walk is a chain of Cout_g x taps exposed
      // memory latencies (16 x 2.25 of them at 128^3; the kernel ran at a quarter of its traffic bound)
      auto trip = [&](int co_g, const float (&t)[OW]) __attribute__((always_inline)) {
        float dv[OW + 1];
#pragma unroll
        for (int j = 0; j < OW; ++j) dv[j] = t[j] * m;
        const float nx = __shfl_down(dv[0], 1, 64);
        dv[OW] = (tx == LW - 1) ? 0.f : nx;                             // column Wo is outside the volume
        const float* wr = s_w + co_g * CIB * 27 + tap0;
#pragma unroll
        for (int ci = 0; ci < CIB; ++ci) {
          const float w0_ = wr[ci * 27 + 0], w1_ = wr[ci * 27 + 1], w2_ = wr[ci * 27 + 2];
#pragma unroll
          for (int j = 0; j < OW; ++j) {
            acc[ci][2 * j] = fmaf(w1_, dv[j], acc[ci][2 * j]);                                   // even voxel: kw = 1
            acc[ci][2 * j + 1] = fmaf(w0_, dv[j + 1], fmaf(w2_, dv[j], acc[ci][2 * j + 1]));     // odd: kw = 0 and 2
          }
        }
      };
      const T* dy0 = (const T*)a.p.xa + n * a.d.xa_bs + (long long)(g * a.Cout_g) * odhw + osp;
      constexpr int U = 4;
      int co_g = 0;
      for (; co_g + U <= a.Cout_g; co_g += U) {
        float t[U][OW];
#pragma unroll
        for (int u = 0; u < U; ++u) ldhalf_c(dy0 + (long long)(co_g + u) * odhw, t[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) trip(co_g + u, t[u]);
      }
      for (; co_g < a.Cout_g; ++co_g) {
        float t[OW];
        ldhalf_c(dy0 + (long long)co_g * odhw, t);
        trip(co_g, t);
      }
    }
  }
  // epilogue: output channels are forward-input channels; epi sc/sh arrays are [N][Cin]
  double s0[CIB], s1[CIB];
  const long long q = ((long long)d_ * H + h_) * W + w0;
#pragma unroll
  for (int ci = 0; ci < CIB; ++ci) {
    s0[ci] = 0.0; s1[ci] = 0.0;
    const int ci_g = cib * CIB + ci;
    if (ci_g < a.Cin_g && ok) {
      const int c = g * a.Cin_g + ci_g;
      T* yp = (T*)a.p.y + n * a.d.y_bs + (long long)c * dhw + q;
      float o[VW];
#pragma unroll
      for (int v = 0; v < VW; ++v) o[v] = acc[ci][v];
      if (a.d.epi == 1) {
        const float esc = a.p.e_sc[n * a.d.Cin + c], esh = a.p.e_sh[n * a.d.Cin + c];
        const T* ep = (c < a.d.Cea ? (const T*)a.p.ea + n * a.d.ea_bs + (long long)c * dhw
                                   : (const T*)a.p.eb + n * a.d.eb_bs + (long long)(c - a.d.Cea) * dhw);
        float ev[VW];
        ldvec(ep, q, ev);
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          o[v] = rnd_as(yp, o[v] * ((ev[v] * esc + esh) > 0.f ? 1.f : a.d.e_slope));
          t0 += o[v];
          t1 += o[v] * ev[v];
        }
        s0[ci] = t0;
        s1[ci] = t1;
      }
      stvec(yp, 0, o);
    }
  }
  if (a.d.epi == 1) {
    double v[2 * CIB];
#pragma unroll
    for (int i = 0; i < CIB; ++i) { v[2 * i] = s0[i]; v[2 * i + 1] = s1[i]; }
    block_sum_d<2 * CIB>(v, s_red, 4);
    // many workgroups per (sample, group, channel block) -- 1 024 at 128^3: two-level fan-in instead of that many same-line fp64
    // atomics (fanin.h; the forward kernels have had it since round 3, this one queued 8 - 50 ns x 1 024 behind its last store)
    if (a.fan && !fan_in<2 * CIB>(a.fan + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * FAN_UNIT_BYTES, blockIdx.x, gridDim.x,
                                  s_red, reinterpret_cast<int*>(s_red + 2 * CIB)))
      return;
    if (tid < 2 * CIB) {
      const int ci_g = cib * CIB + (tid >> 1);
      if (ci_g < a.Cin_g) {
        const int c = g * a.Cin_g + ci_g;
        atomicAdd(&a.p.red[((long long)n * a.d.Cin + c) * 2 + (tid & 1)], s_red[tid]);
      }
    }
  }
}

template <typename T>
static int dgrad_s2_dispatch(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p) {
  const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  const int cib = pick_cob(cin_g, 4);
  ConvK a = make_k(d, p, 1, 8);
  const long long dhw = (long long)d->D * d->H * d->W;
  dim3 grid((unsigned)((dhw + 255) / 256), cdiv(cin_g, cib), d->N * d->groups);
  const size_t shm = ((size_t)cout_g * cib * 27 + 2 + 2 * 4 * 2 * cib) * sizeof(float);
  if (shm > 64 * 1024) return XH_ERR_ARG;
  {
    constexpr int VW = VWT<T>::v;
    const int lw = d->W / VW;
    const bool al = d->W % VW == 0 && d->Wo * 2 == d->W && d->D % 2 == 0 && d->H % 2 == 0 && lw >= 1 && lw <= 64 &&
                    (64 % lw) == 0 && dhw % VW == 0 && d->xa_bs % (VW / 2) == 0 && d->y_bs % VW == 0 &&
                    ((long long)d->Do * d->Ho * d->Wo) % (VW / 2) == 0 &&
                    (d->epi != 1 || (d->ea_bs % VW == 0 && d->eb_bs % VW == 0)) && !(g_xh_disable & 4);
    if (al) {
      const long long rows = (long long)(d->D / 2) * (d->H / 2);        // per (d, h) parity class
      const int bpc = (int)((rows * lw + 255) / 256);
      dim3 gridv(4 * bpc, cdiv(cin_g, cib), d->N * d->groups);
      if (d->epi == 1) a.fan = xh_fan_block(p->fan, p->fan_bytes, (long long)gridv.y * gridv.z, gridv.x);
      switch (cib) {
        case 1: hipLaunchKernelGGL((conv3_dgrad_s2_vec_kernel<T, 1>), gridv, dim3(256), shm, (hipStream_t)stream, a, lw, (int)rows); break;
        case 2: hipLaunchKernelGGL((conv3_dgrad_s2_vec_kernel<T, 2>), gridv, dim3(256), shm, (hipStream_t)stream, a, lw, (int)rows); break;
        default: hipLaunchKernelGGL((conv3_dgrad_s2_vec_kernel<T, 4>), gridv, dim3(256), shm, (hipStream_t)stream, a, lw, (int)rows);
      }
      return xh_launch_status();
    }
  }
  switch (cib) {
    case 1: hipLaunchKernelGGL((conv3_dgrad_s2_kernel<T, 1>), grid, dim3(256), shm, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL((conv3_dgrad_s2_kernel<T, 2>), grid, dim3(256), shm, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL((conv3_dgrad_s2_kernel<T, 4>), grid, dim3(256), shm, (hipStream_t)stream, a);
  }
  return xh_launch_status();
}

extern "C" int xh_conv3d_dgrad_s2(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p) {
  if (!d || !p || !p->xa || !p->y) return XH_ERR_ARG;
  if (d->dtype != XH_F32 && d->dtype != XH_BF16 && d->dtype != XH_F16) return XH_ERR_DTYPE;
  if (d->k != 3 || d->stride != 2 || d->groups <= 0 || d->Cin % d->groups || d->Cout % d->groups) return XH_ERR_ARG;
  if (d->Do != (d->D - 1) / 2 + 1 || d->Ho != (d->H - 1) / 2 + 1 || d->Wo != (d->W - 1) / 2 + 1) return XH_ERR_ARG;
  if (!(d->n_wptr == 1 || (d->n_wptr == d->groups && d->groups <= XH_MAX_WPTR))) return XH_ERR_ARG;
  for (int i = 0; i < d->n_wptr; ++i)
    if (!p->w[i]) return XH_ERR_ARG;
  if (d->epi == 1 && (!p->ea || !p->e_sc || !p->e_sh || !p->red || (d->Cea < d->Cin && !p->eb))) return XH_ERR_ARG;
  if (d->epi != 0 && d->epi != 1) return XH_ERR_ARG;
  if (d->N * d->groups > 65535) return XH_ERR_ARG;
  XH_DISPATCH_T(d->dtype, return dgrad_s2_dispatch<T>(stream, d, p););
}

#define LAUNCH_WG(T, K, S, COB, TXN)                                                                               \
  do {                                                                                                             \
    xh_note_kernel("conv_wgrad_kernel<%s, %d, %d, %d, %d>", tname<T>(), K, S, COB, TXN);                          \
    hipLaunchKernelGGL((conv_wgrad_kernel<T, K, S, COB, TXN>), grid, dim3(256), 0, (hipStream_t)stream, wa);       \
  } while (0)
#define WG_TXN(T, K, S, COB)                                        \
  do {                                                              \
    if (txn == 8) { LAUNCH_WG(T, K, S, COB, 8); }                   \
    else if (txn == 4) { LAUNCH_WG(T, K, S, COB, 4); }              \
    else { LAUNCH_WG(T, K, S, COB, 2); }                            \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// Depthwise k = 3 weight gradient, large volumes: the same sliding window as conv_dw3_slide_kernel.  A lane owns one
// 16-byte run of a row and marches along D; each x plane is read once (3 rows + wave-shuffle neighbours, producer
// norm/activation applied), each dY plane once, and the 27 tap sums live in registers until one block reduction.
// ---------------------------------------------------------------------------------------------------
template <typename T, int TXN>
__global__ __launch_bounds__(256) void conv_dw3_wgrad_slide_kernel(const WgradK wa, int sd) {
  const ConvK& a = wa.c;
  constexpr int VW = VWT<T>::v, TH = 256 / TXN;
  __shared__ float s_red[4 * 28];
  const int tid = threadIdx.x;
  const int tx = tid % TXN, ty = tid / TXN;
  const int c = blockIdx.y, n = blockIdx.z;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  int t = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tw = t % a.tilesW; t /= a.tilesW;
  const int th = t % a.tilesH;
  const int ds = t / a.tilesH;
  const int oh = th * TH + ty, ow = tw * TXN * VW + tx * VW;
  const bool active = oh < H && ow < W;
  const int owc = active ? ow : 0, ohc = min(oh, H - 1);
  const int d_begin = ds * sd, d_end = min(D, d_begin + sd);
  float sc = 1.f, sh = 0.f;
  if (a.d.pre) { sc = a.p.pre_sc[n * a.d.Cin + c]; sh = a.p.pre_sh[n * a.d.Cin + c]; }
  const T* src = in_plane<T>(a, n, c, dhw);
  const T* dyp = (const T*)a.p.ea + n * a.d.ea_bs + (long long)c * dhw + (long long)ohc * W + owc;
  int roff[3];
  float rmask[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int gh = oh - 1 + kh;
    rmask[kh] = (active && (unsigned)gh < (unsigned)H) ? 1.f : 0.f;
    roff[kh] = min(max(gh, 0), H - 1) * W;
  }
  const bool multi_w = a.tilesW > 1;
  const bool edge_l = tx == 0, edge_r = tx == TXN - 1 || ow + VW >= W;
  const bool need_l = multi_w && edge_l && ow > 0, need_r = multi_w && edge_r && ow + VW < W;
  const int col_l = max(ow - 1, 0), col_r = min(ow + VW, W - 1);
  const float amask = active ? 1.f : 0.f;

  float acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.f;
  float dbs = 0.f;
  float dyA[VW], dyB[VW], dyC[VW], dyN[VW];            // dY planes q-1, q, q+1 and the prefetched q+2
#pragma unroll
  for (int v = 0; v < VW; ++v) dyA[v] = dyB[v] = dyC[v] = dyN[v] = 0.f;
  float nxt[3][VW], nhl[3] = {0.f, 0.f, 0.f}, nhr[3] = {0.f, 0.f, 0.f};
  auto load_x = [&](int q) {
    const T* pl = src + (long long)q * hw;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      ldvec(pl, roff[kh] + owc, nxt[kh]);
      if (multi_w) {
        nhl[kh] = ldf(pl, roff[kh] + col_l);
        nhr[kh] = ldf(pl, roff[kh] + col_r);
      }
    }
  };
  auto load_dy = [&](int p, float (&o)[VW]) {           // zero outside this block's planes
    if (p >= d_begin && p < d_end) {
      ldvec(dyp, (long long)p * hw, o);
#pragma unroll
      for (int v = 0; v < VW; ++v) o[v] *= amask;
    } else {
#pragma unroll
      for (int v = 0; v < VW; ++v) o[v] = 0.f;
    }
  };
  // x planes q = d_begin-1 .. d_end pair with dY planes q+1 (kd=0), q (kd=1), q-1 (kd=2)
  bool nxt_ok = d_begin - 1 >= 0;
  if (nxt_ok) load_x(d_begin - 1);
  load_dy(d_begin, dyC);                                // q = d_begin-1: dyA = dy[q-1] = 0, dyB = dy[q] = 0 (not ours), dyC = dy[q+1]
  load_dy(d_begin + 1, dyN);
  for (int q = d_begin - 1; q <= d_end; ++q) {          // block-uniform
    float cur[3][VW], hl[3], hr[3];
    const bool cur_ok = nxt_ok;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int v = 0; v < VW; ++v) cur[kh][v] = nxt[kh][v];
      hl[kh] = nhl[kh];
      hr[kh] = nhr[kh];
    }
    float dyQ[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) dyQ[v] = 0.f;
    if (q + 1 <= d_end) {
      nxt_ok = q + 1 < D;
      if (nxt_ok) load_x(q + 1);
      load_dy(q + 3, dyQ);                              // becomes dyN after the rotation below
    }
    if (cur_ok) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        float r[VW + 2];
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          float x = cur[kh][v];
          if (a.d.pre) x = leaky(x * sc + sh, a.d.pre_slope);
          r[v + 1] = x * rmask[kh];
        }
        float l = __shfl_up(r[VW], 1, 64), rr = __shfl_down(r[1], 1, 64);
        if (edge_l) {
          float x = hl[kh];
          if (a.d.pre) x = leaky(x * sc + sh, a.d.pre_slope);
          l = need_l ? x * rmask[kh] : 0.f;
        }
        if (edge_r) {
          float x = hr[kh];
          if (a.d.pre) x = leaky(x * sc + sh, a.d.pre_slope);
          rr = need_r ? x * rmask[kh] : 0.f;
        }
        r[0] = l;
        r[VW + 1] = rr;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int v = 0; v < VW; ++v) {
            s0 = fmaf(r[v + kw], dyC[v], s0);           // kd = 0: dY plane q+1
            s1 = fmaf(r[v + kw], dyB[v], s1);           // kd = 1: dY plane q
            s2 = fmaf(r[v + kw], dyA[v], s2);           // kd = 2: dY plane q-1
          }
          acc[0 * 9 + kh * 3 + kw] += s0;
          acc[1 * 9 + kh * 3 + kw] += s1;
          acc[2 * 9 + kh * 3 + kw] += s2;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      dbs += dyB[v];                                     // every owned dY plane is dyB exactly once
      dyA[v] = dyB[v]; dyB[v] = dyC[v]; dyC[v] = dyN[v]; dyN[v] = dyQ[v];
    }
  }
  float v28[28];
#pragma unroll
  for (int i = 0; i < 27; ++i) v28[i] = acc[i];
  v28[27] = dbs;
  block_sum<28>(v28, s_red, 4);
  const int gpp = a.d.groups / a.d.n_wptr;
  if (tid < 27) atomicAdd(&wa.dw[c / gpp][(long long)(c % gpp) * 27 + tid], s_red[tid]);
  if (tid == 27 && wa.db[c / gpp]) atomicAdd(&wa.db[c / gpp][c % gpp], s_red[27]);
}

// ---------------------------------------------------------------------------------------------------
// k = 3, stride = 2 weight gradient, vectorised (the DRB convs).  A workgroup fixes (group, input channel ci); a lane
// owns VW/2 consecutive outputs of one output row: it reads the 9 input rows of its channel as 16-byte runs (left
// neighbour column by wave shuffle, producer norm/activation applied) and the dY run of every output channel of the
// group, and keeps the 27 x COUT_G partial sums in registers until one block reduction.
// ---------------------------------------------------------------------------------------------------
template <typename T, int CO>
__device__ __forceinline__ void s2_wgrad_body(const WgradK& wa, int LW, int bx, int by, int bz, int gdx) {
  const ConvK& a = wa.c;
  constexpr int VW = VWT<T>::v, OW = VW / 2, NACC = 27 * CO + CO;
  __shared__ float s_red[4 * NACC];
  const int tid = threadIdx.x;
  // by = (chunk of CO output channels of the group, input channel): groups of 8 / 16 output channels (the deep DRB
  // convs) run as 2 / 4 chunks of 4 instead of the generic tiled kernel (39 us for 64 -> 32 g4 @32^3)
  const int ci_g = by % a.Cin_g, co0 = (by / a.Cin_g) * CO;
  const int n = bz / a.d.groups, g = bz % a.d.groups;
  const int D = a.d.D, H = a.d.H, W = a.d.W, Do = a.d.Do, Ho = a.d.Ho, Wo = a.d.Wo;
  const long long dhw = (long long)D * H * W, odhw = (long long)Do * Ho * Wo;
  const int c = g * a.Cin_g + ci_g;
  float sc = 1.f, sh = 0.f;
  if (a.d.pre) { sc = a.p.pre_sc[n * a.d.Cin + c]; sh = a.p.pre_sh[n * a.d.Cin + c]; }
  float acc[27][CO], dbs[CO];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[t][j] = 0.f;
#pragma unroll
  for (int j = 0; j < CO; ++j) dbs[j] = 0.f;
  const long long rows = (long long)Do * Ho;
  const long long lanes = rows * LW;
  for (long long lane_id = (long long)bx * 256 + tid; lane_id - tid < lanes; lane_id += (long long)gdx * 256) {
    const int tx = (int)(lane_id % LW);
    const long long row = lane_id / LW;
    const bool ok = row < rows;
    const int oh = (int)(row % Ho), od = (int)min(row / Ho, (long long)Do - 1);
    const int ow0 = tx * OW;
    const float okm = ok ? 1.f : 0.f;
    float dyv[CO][OW];
#pragma unroll
    for (int j = 0; j < CO; ++j) {
      const T* dp = (const T*)a.p.ea + n * a.d.ea_bs + (long long)(g * a.Cout_g + co0 + j) * odhw + ((long long)od * Ho + oh) * Wo + ow0;
      ldhalf_c(dp, dyv[j]);
#pragma unroll
      for (int v = 0; v < OW; ++v) { dyv[j][v] *= okm; dbs[j] += dyv[j][v]; }
    }
    const T* src = in_plane<T>(a, n, c, dhw) + 2 * ow0;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int gd = 2 * od - 1 + kd, gh = 2 * oh - 1 + kh;
        const float m = ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H) ? 1.f : 0.f;
        float x[VW], r[VW + 1];
        ldvec(src, ((long long)min(max(gd, 0), D - 1) * H + min(max(gh, 0), H - 1)) * W, x);
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          float xv = x[v];
          if (a.d.pre) xv = leaky(xv * sc + sh, a.d.pre_slope);
          r[v + 1] = xv * m;
        }
        const float l = __shfl_up(r[VW], 1, 64);
        r[0] = tx == 0 ? 0.f : l;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int j = 0; j < CO; ++j) {
            float t = acc[(kd * 3 + kh) * 3 + kw][j];
#pragma unroll
            for (int v = 0; v < OW; ++v) t = fmaf(r[2 * v + kw], dyv[j][v], t);
            acc[(kd * 3 + kh) * 3 + kw][j] = t;
          }
      }
  }
  float v[NACC];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int j = 0; j < CO; ++j) v[t * CO + j] = acc[t][j];
#pragma unroll
  for (int j = 0; j < CO; ++j) v[27 * CO + j] = dbs[j];
  block_sum<NACC>(v, s_red, 4);
  const int gpp = a.d.groups / a.d.n_wptr, gl = g % gpp;
  if (tid < 27 * CO) {
    const int t = tid / CO, j = tid % CO;
    atomicAdd(&wa.dw[g / gpp][((long long)(gl * a.Cout_g + co0 + j) * a.Cin_g + ci_g) * 27 + t], s_red[tid]);
  } else if (tid < NACC && ci_g == 0 && wa.db[g / gpp]) {
    atomicAdd(&wa.db[g / gpp][gl * a.Cout_g + co0 + (tid - 27 * CO)], s_red[tid]);
  }
}
template <typename T, int CO>
__global__ __launch_bounds__(256) void conv3_s2_wgrad_vec_kernel(const WgradK wa, int LW) {
  s2_wgrad_body<T, CO>(wa, LW, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}
// up to S2W_MULTI problems of one launch (xh_conv3d_wgrad_batch): workgroup b belongs to problem i with off[i] <= b < off[i + 1]
constexpr int S2W_MULTI = 4;
struct S2WMulti {
  int n;
  int off[S2W_MULTI + 1];
  int lw[S2W_MULTI], gx[S2W_MULTI], gy[S2W_MULTI], co[S2W_MULTI];     // co: output channels per workgroup (2: groups of two; else 4)
  WgradK p[S2W_MULTI];
};
static_assert(sizeof(S2WMulti) <= 3900, "kernel-argument table");
// Both instances of the body in one kernel: the two-channel problem (the level-0 DRB conv, 41 us alone on 512 workgroups -- bound by
// the latency of its load chain, not by bandwidth) runs NEXT TO the four-channel problems of the deeper levels instead of after them.
template <typename T>
__global__ __launch_bounds__(256) void conv3_s2_wgrad_vec_multi_kernel(const S2WMulti m) {
  int pi = 0;
  for (int k = 1; k < S2W_MULTI; ++k)
    if (k < m.n && (int)blockIdx.x >= m.off[k]) pi = k;
  const int local = blockIdx.x - m.off[pi];
  const int gx = m.gx[pi], gy = m.gy[pi];
  const int r = local / gx;
  if (m.co[pi] == 2) s2_wgrad_body<T, 2>(m.p[pi], m.lw[pi], local - r * gx, r % gy, r / gy, gx);
  else s2_wgrad_body<T, 4>(m.p[pi], m.lw[pi], local - r * gx, r % gy, r / gy, gx);
}

// launch plan of the vectorised stride-2 weight gradient; false: not eligible
static bool s2w_plan(const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR], WgradK* wa, int* lw_,
                     dim3* grid);

template <typename T>
static int wgrad_dispatch(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR],
                          float* const db[XH_MAX_WPTR]) {
  const int cout_g = d->Cout / d->groups, cin_g = d->Cin / d->groups;
  WgradK wa;
  for (int i = 0; i < XH_MAX_WPTR; ++i) { wa.dw[i] = i < d->n_wptr ? dw[i] : nullptr; wa.db[i] = (i < d->n_wptr && db) ? db[i] : nullptr; }
  if (d->k == 1) {
    wa.c = make_k(d, p, 4, 8);
    const long long dhw1 = (long long)d->D * d->H * d->W;
    constexpr int VW1 = VWT<T>::v;
    const bool vec = (dhw1 % VW1 == 0) && (d->xa_bs % VW1 == 0) && (d->xb_bs % VW1 == 0) && (d->ea_bs % VW1 == 0);
    const long long total = (long long)d->N * dhw1 / (vec ? VW1 : 1);
    const int ny1 = cdiv(cin_g, 4) * cdiv(cout_g, 4) * d->groups;
    // Every workgroup ends in a pass of same-address device-scope atomics, which serialise (~65 ns each across the 8
    // XCDs): a few hundred workgroups is the optimum between the per-lane load chain and that tail.
    int gx = (int)((total + 255) / 256);
    const int cap = cdiv(g_c1w_wgs, ny1);
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    dim3 grid(gx, cdiv(cin_g, 4) * cdiv(cout_g, 4), d->groups);
    xh_note_kernel("conv1x1_wgrad_kernel<%s, 4, 4, %s>", tname<T>(), vec ? "true" : "false");
    if (vec) hipLaunchKernelGGL((conv1x1_wgrad_kernel<T, 4, 4, true>), grid, dim3(256), 0, (hipStream_t)stream, wa);
    else hipLaunchKernelGGL((conv1x1_wgrad_kernel<T, 4, 4, false>), grid, dim3(256), 0, (hipStream_t)stream, wa);
    return xh_launch_status();
  }
  if (d->k == 3 && d->stride == 1 && cin_g == 1 && cout_g == 1 && !(g_xh_disable & 1)) {
    constexpr int VW = VWT<T>::v;
    const long long dhw3 = (long long)d->D * d->H * d->W;
    const bool al = d->W % VW == 0 && dhw3 % VW == 0 && d->xa_bs % VW == 0 && d->xb_bs % VW == 0 && d->ea_bs % VW == 0;
    if (al && d->H >= 32 && d->W >= 4 * VW && d->D >= 8 && d->Cin < 65536 && d->N < 65536) {
      wa.c = make_k(d, p, 1, 8);
      int txn = 4;
      while (txn < 32 && txn * VW < d->W) txn *= 2;
      wa.c.tilesW = cdiv(d->W, txn * VW);
      wa.c.tilesH = cdiv(d->H, 256 / txn);
      const int base = wa.c.tilesW * wa.c.tilesH * d->Cin * d->N;
      int dsegs = cdiv(g_dw_target, base);
      if (dsegs > d->D / dw_min_planes(dhw3)) dsegs = d->D / dw_min_planes(dhw3);
      if (dsegs < 1) dsegs = 1;
      const int sd = cdiv(d->D, dsegs);
      dsegs = cdiv(d->D, sd);
      dim3 grid(wa.c.tilesW * wa.c.tilesH * dsegs, d->Cin, d->N);
      xh_note_kernel("conv_dw3_wgrad_slide_kernel<%s, %d>", tname<T>(), txn);
      switch (txn) {
        case 4: hipLaunchKernelGGL((conv_dw3_wgrad_slide_kernel<T, 4>), grid, dim3(256), 0, (hipStream_t)stream, wa, sd); break;
        case 8: hipLaunchKernelGGL((conv_dw3_wgrad_slide_kernel<T, 8>), grid, dim3(256), 0, (hipStream_t)stream, wa, sd); break;
        case 16: hipLaunchKernelGGL((conv_dw3_wgrad_slide_kernel<T, 16>), grid, dim3(256), 0, (hipStream_t)stream, wa, sd); break;
        default: hipLaunchKernelGGL((conv_dw3_wgrad_slide_kernel<T, 32>), grid, dim3(256), 0, (hipStream_t)stream, wa, sd);
      }
      return xh_launch_status();
    }
  }
  {
    int lw;
    dim3 grid;
    if (s2w_plan(d, p, dw, db, &wa, &lw, &grid)) {
      xh_note_kernel("conv3_s2_wgrad_vec_kernel<%s, %d>", tname<T>(), cout_g == 2 ? 2 : 4);
      if (cout_g == 2) hipLaunchKernelGGL((conv3_s2_wgrad_vec_kernel<T, 2>), grid, dim3(256), 0, (hipStream_t)stream, wa, lw);
      else hipLaunchKernelGGL((conv3_s2_wgrad_vec_kernel<T, 4>), grid, dim3(256), 0, (hipStream_t)stream, wa, lw);
      return xh_launch_status();
    }
  }
  const int txn = pick_txn(d->Wo);
  const int cob = (d->k == 7) ? pick_cob(cout_g, 2) : pick_cob(cout_g, 4);
  wa.c = make_k(d, p, cob, txn);
  const int tiles = wa.c.tilesW * wa.c.tilesH * wa.c.tilesD;
  wa.tiles_total = tiles * d->N;
  // enough slabs to fill the chip, few enough that the end-of-block reduction amortises
  const int ny = cin_g * wa.c.ncob * (d->k == 7 ? 7 : 1) * d->groups;
  int slabs = cdiv(g_wg_slabs, ny);
  if (slabs > wa.tiles_total) slabs = wa.tiles_total;
  if (slabs < 1) slabs = 1;
  wa.tiles_per_block = cdiv(wa.tiles_total, slabs);
  slabs = cdiv(wa.tiles_total, wa.tiles_per_block);
  dim3 grid(slabs, cin_g * wa.c.ncob * (d->k == 7 ? 7 : 1), d->groups);
  if (grid.y > 65535) return XH_ERR_ARG;
  if (d->k == 3 && d->stride == 1) {
    switch (cob) { case 1: WG_TXN(T, 3, 1, 1); break; case 2: WG_TXN(T, 3, 1, 2); break; default: WG_TXN(T, 3, 1, 4); }
  } else if (d->k == 3 && d->stride == 2) {
    switch (cob) { case 1: WG_TXN(T, 3, 2, 1); break; case 2: WG_TXN(T, 3, 2, 2); break; default: WG_TXN(T, 3, 2, 4); }
  } else {
    switch (cob) { case 1: WG_TXN(T, 7, 1, 1); break; default: WG_TXN(T, 7, 1, 2); }
  }
  return xh_launch_status();
}

extern "C" int xh_conv3d_wgrad(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR],
                               float* const db[XH_MAX_WPTR]) {
  int rc = check_desc(d, p);
  if (rc) return rc;
  if (!p->ea || !dw || d->transposed) return XH_ERR_ARG;
  for (int i = 0; i < d->n_wptr; ++i)
    if (!dw[i]) return XH_ERR_ARG;
  if (d->bcast) {                                      // broadcast input: the full-row quad-channel kernel or nothing
    if (d->bcast != 4 || d->Ca != d->Cin || d->k != 3) return XH_ERR_ARG;
    const int r = g_use_mfma ? xh_conv3_wgrad_mfma_try(stream, d, p, dw, db) : 1;
    return r == 1 ? XH_ERR_ARG : r;
  }
  if (g_use_mfma) {
    const int r = d->k == 7 ? xh_conv7_wgrad_mfma_try(stream, d, p, dw, db) : xh_conv3_wgrad_mfma_try(stream, d, p, dw, db);
    if (r != 1) return r;
  }
  {
    const int r = xh_conv3_tiny_wgrad_try(stream, d, p, dw[0], db ? db[0] : nullptr);     // conv3_tiny.hip: 1 <-> 2 channels
    if (r != 1) return r;
  }
  xh_note_kernel("conv wgrad k%d s%d (vector kernel family)", d->k, d->stride);
  XH_DISPATCH_T(d->dtype, return wgrad_dispatch<T>(stream, d, p, dw, db););
}

// k = 1 weight gradients of a batch (xh_conv3d_wgrad_batch): those conv1x1_wgrad_multi_kernel can take (16-byte runs) go
// C1W_MULTI per launch and storage type and are marked in handled[]; the others are left to the caller's one-by-one path.
static bool c1w_plan(const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR], C1WP* o) {
  if (d->n_wptr > 4) return false;
  if (check_desc(d, p) || d->k != 1 || d->stride != 1 || d->transposed || !p->ea || !dw) return false;
  if (d->n_wptr < 1 || d->groups % d->n_wptr) return false;
  for (int i = 0; i < d->n_wptr; ++i)
    if (!dw[i]) return false;
  const int vw = d->dtype == XH_F32 ? 4 : 8;
  const long long dhw = (long long)d->D * d->H * d->W;
  if (dhw % vw || d->xa_bs % vw || d->xb_bs % vw || d->ea_bs % vw) return false;
  const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  o->xa = p->xa; o->xb = p->xb; o->dy = p->ea;
  o->pre_sc = p->pre_sc; o->pre_sh = p->pre_sh;
  for (int i = 0; i < 4; ++i) { o->dw[i] = i < d->n_wptr ? dw[i] : nullptr; o->db[i] = (i < d->n_wptr && db) ? db[i] : nullptr; }
  o->xa_bs = d->xa_bs; o->xb_bs = d->xb_bs; o->ea_bs = d->ea_bs; o->dhw = dhw;
  o->N = d->N; o->Cin = d->Cin; o->Ca = d->Ca; o->Cin_g = cin_g; o->Cout_g = cout_g; o->groups = d->groups; o->n_wptr = d->n_wptr;
  o->pre = d->pre; o->pre_slope = d->pre_slope; o->dtype = d->dtype;
  o->ny = cdiv(cin_g, 4) * cdiv(cout_g, 4);
  // every workgroup ends in same-address device-scope atomics, which serialise: a few hundred workgroups per problem (key 8)
  const long long total = (long long)d->N * dhw / vw;
  long long gx = (total + 255) / 256;
  const long long cap = cdiv(g_c1w_wgs, o->ny * d->groups);
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  o->gx = (int)gx;
  return (long long)o->gx * o->ny * d->groups < (1 << 20);
}
static int c1w_launch(void* stream, int dtype, const C1WP* v, int n) {
  C1WMulti m;
  m.n = n; m.off[0] = 0;
  for (int i = 0; i < n; ++i) { m.p[i] = v[i]; m.off[i + 1] = m.off[i] + v[i].gx * v[i].ny * v[i].groups; }
  for (int i = n; i < C1W_MULTI; ++i) m.off[i + 1] = m.off[n];
  xh_note_kernel("conv1x1_wgrad_multi_kernel<%s>", dtype == XH_F32 ? "float" : (dtype == XH_F16 ? "f16_t" : "bf16_t"));
  XH_DISPATCH_T(dtype, hipLaunchKernelGGL((conv1x1_wgrad_multi_kernel<T>), dim3(m.off[n]), dim3(256), 0, (hipStream_t)stream, m););
  return xh_launch_status();
}
int xh_c1w_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, float* const (*dw)[XH_MAX_WPTR],
                 float* const (*db)[XH_MAX_WPTR], char* handled) {
  if (g_xh_disable & 512) return XH_OK;
  int rc_all = XH_OK;
  const int types[3] = {XH_F32, XH_BF16, XH_F16};
  for (int t = 0; t < 3; ++t) {
    C1WP v[C1W_MULTI];
    int k = 0;
    for (int i = 0; i < n; ++i) {
      if (handled[i] || !d[i] || !p[i] || d[i]->dtype != types[t]) continue;
      if (!c1w_plan(d[i], p[i], dw[i], db ? db[i] : nullptr, &v[k])) continue;
      handled[i] = 1;
      if (++k == C1W_MULTI) {
        const int rc = c1w_launch(stream, types[t], v, k);
        if (rc != XH_OK) rc_all = rc;
        k = 0;
      }
    }
    if (k) {
      const int rc = c1w_launch(stream, types[t], v, k);
      if (rc != XH_OK) rc_all = rc;
    }
  }
  return rc_all;
}

static bool s2w_plan(const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR], WgradK* wa, int* lw_,
                     dim3* grid) {
  const int cout_g = d->Cout / d->groups, cin_g = d->Cin / d->groups;
  if (!(d->k == 3 && d->stride == 2 && (cout_g == 2 || cout_g % 4 == 0) && !(g_xh_disable & 4))) return false;
  const int VW = d->dtype == XH_F32 ? 4 : 8;
  const int lw = d->W / VW;
  const long long dhw2 = (long long)d->D * d->H * d->W, odhw2 = (long long)d->Do * d->Ho * d->Wo;
  const bool al = d->W % VW == 0 && d->Wo * 2 == d->W && lw >= 1 && lw <= 64 && (64 % lw) == 0 && dhw2 % VW == 0 &&
                  odhw2 % (VW / 2) == 0 && d->xa_bs % VW == 0 && d->xb_bs % VW == 0 && d->ea_bs % (VW / 2) == 0 &&
                  (long long)cin_g * (cout_g / (cout_g == 2 ? 2 : 4)) <= 65535 && (long long)d->N * d->groups <= 65535;
  if (!al) return false;
  for (int i = 0; i < XH_MAX_WPTR; ++i) { wa->dw[i] = i < d->n_wptr ? dw[i] : nullptr; wa->db[i] = (i < d->n_wptr && db) ? db[i] : nullptr; }
  wa->c = make_k(d, p, 1, 8);
  wa->tiles_total = wa->tiles_per_block = 0;
  const int chunks = cout_g == 2 ? 1 : cout_g / 4;
  const long long lanes = (long long)d->Do * d->Ho * lw;
  long long gx = (lanes + 255) / 256;
  const long long cap = cdiv(g_s2w_cap, cin_g * chunks * d->groups * d->N);   // few enough workgroups that the atomics tail stays small
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  *grid = dim3((unsigned)gx, cin_g * chunks, d->N * d->groups);
  *lw_ = lw;
  return true;
}
// k = 3 stride-2 weight gradients of a batch with groups of 4 k output channels: S2W_MULTI per launch and storage type
int xh_s2w_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, float* const (*dw)[XH_MAX_WPTR],
                 float* const (*db)[XH_MAX_WPTR], char* handled) {
  if (g_xh_disable & 512) return XH_OK;
  int rc_all = XH_OK;
  const int types[3] = {XH_F32, XH_BF16, XH_F16};
  S2WMulti* m = new S2WMulti;
  for (int t = 0; t < 3; ++t) {
    m->n = 0; m->off[0] = 0;
    auto launch = [&]() -> int {
      XH_DISPATCH_T(types[t], hipLaunchKernelGGL((conv3_s2_wgrad_vec_multi_kernel<T>), dim3(m->off[m->n]), dim3(256), 0,
                                                 (hipStream_t)stream, *m););
      return XH_OK;
    };
    auto flush = [&]() {
      if (m->n == 0) return;
      for (int i = m->n; i < S2W_MULTI; ++i) m->off[i + 1] = m->off[m->n];
      xh_note_kernel("conv3_s2_wgrad_vec_multi_kernel<%s>", types[t] == XH_F32 ? "float" : (types[t] == XH_F16 ? "f16_t" : "bf16_t"));
      if (launch() != XH_OK || xh_launch_status() != XH_OK) rc_all = XH_ERR_HIP;
      m->n = 0; m->off[0] = 0;
    };
    for (int i = 0; i < n; ++i) {
      if (handled[i] || !d[i] || !p[i] || d[i]->dtype != types[t]) continue;
      if (check_desc(d[i], p[i]) || !p[i]->ea || !dw[i] || d[i]->transposed || d[i]->groups <= 0) continue;
      bool ok = true;
      for (int j = 0; j < d[i]->n_wptr; ++j) ok = ok && dw[i][j];
      dim3 grid;
      const int k = m->n;
      if (!ok || !s2w_plan(d[i], p[i], dw[i], db ? db[i] : nullptr, &m->p[k], &m->lw[k], &grid)) continue;
      const long long blocks = (long long)grid.x * grid.y * grid.z;
      if (blocks > (1 << 22)) continue;
      handled[i] = 1;
      m->gx[k] = (int)grid.x; m->gy[k] = (int)grid.y;
      m->co[k] = (d[i]->Cout / d[i]->groups) == 2 ? 2 : 4;
      m->off[k + 1] = m->off[k] + (int)blocks;
      if (++m->n == S2W_MULTI) flush();
    }
    flush();
  }
  delete m;
  return rc_all;
}

int xh_check_conv(const xh_conv_desc* d, const xh_conv_ptrs* p) { return check_desc(d, p); }   // for the batch paths of other files
extern "C" int xh_abi_version(void) { return 2; }
