// bf16-MFMA weight gradient of the 3x3x3 stride-1 convolution (bf16 storage) for gfx950.
//
//   dW[co][ci][kd,kh,kw] = sum over voxels p of dY[co][p] * xa[ci][p + (kd,kh,kw) - 1],   xa = leaky(x*sc+sh)
//
// GEMM view: D[16 co][16 columns] += A[16 co][32 voxels] * B[32 voxels][16 columns], K = voxels.
//  * A (dY): lane (g = l>>4, m = l&15) supplies 8 consecutive voxels along W of output channel m: one 16-byte global
//    load, reused by every MFMA of the K-step.
//  * B (xa): column n = (input channel n % CP, row selector n / CP); an MFMA "tile" t fixes kw and assigns the
//    16/CP row selectors to consecutive (kd,kh) rows, so all lanes of an MFMA share the W shift kw-1.  The input is
//    kept channel-planar in LDS ([ci][h][48 voxels] bf16, sliding 4-plane ring along D like the forward kernel); for a
//    (kd,kh) row a lane reads ONE aligned 16-byte chunk plus the dword on either side and builds the three kw-shifted
//    fragments with v_alignbyte: 3 LDS reads + 8 VALU per 3 MFMAs.
//  * Each wave keeps all 3*ceil(9/(16/CP)) accumulator tiles in registers for the whole run of planes; one LDS
//    reduction across waves and one pass of contiguous fp32 atomics per workgroup at the end.
//  * Narrow volumes (W = 16 / 8, the deep levels): the 32 voxels of a K step are 2 / 4 consecutive rows of 16 / 8 voxels
//    (a.rs rows per step); only the lane -> (row, chunk) mapping changes.
#include "common.h"
#include "../../include/xlstm_hved.h"
#include "wgrad_q4.h"
#include <vector>
#include <algorithm>

typedef h16x8 bf16x8;     // 8 raw 16-bit values (either format)
typedef f32x4_t f32x4;

struct WgMK {
  xh_conv_desc d;
  xh_conv_ptrs p;
  float* dw[XH_MAX_WPTR];
  float* db[XH_MAX_WPTR];
  int Cin_g, Cout_g;
  int tilesW, tilesH, sd, dsegs;
  int gs;           // groups per set
  int cin_set;      // input channels of a set (gs * Cin_g)
  int cout_set;     // output channels of a set
  int ntile;        // 16-wide output-channel tiles per set
  int nctile;       // CP-wide input-channel tiles per set
  int rs;           // rows per MFMA K group: 1 (W % 32 == 0), 2 (W == 16), 4 (W == 8)
};

template <int FMT, int CP, int NT>
__device__ __forceinline__ void conv3_wgrad_body(const WgMK& a, const int bid_x, const int bid_y, const int grid_x) {
  typedef h16<FMT> ST;                                // storage type: bf16_t or f16_t
  constexpr int NWV = NT / 64;
  constexpr int TH = 8, IH = TH + 2;
  constexpr int ROWB = 96;                            // 48 voxels: [ow0-8, ow0+40)
  constexpr int CHS = IH * ROWB + 16;                 // channel stride, padded so 16 channels hit 16 distinct bank groups
  constexpr int PLANE = CP * CHS;
  constexpr int R = 16 / CP;                          // (kd,kh) rows per MFMA tile
  constexpr int TPK = (9 + R - 1) / R;                // tiles per kw
  constexpr int NITEM = CP * IH * 6;                  // staging items per plane (one 16-byte chunk each)
  constexpr int NIT = (NITEM + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_in = smem;                         // 4 * PLANE
  float* s_dw = reinterpret_cast<float*>(smem);       // reused after the plane loop: [16 co][CP ci][27] + [16] bias sums

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g4 = lane >> 4, nn = lane & 15;
  int y = bid_y;
  const int ct = y % a.nctile; y /= a.nctile;
  const int nt = y % a.ntile;
  const int set = y / a.ntile;
  const int n_first = 0;
  (void)n_first;
  const int ci0 = set * a.cin_set + ct * CP;          // first input channel of this tile
  const int ci_lim = min(CP, a.cin_set - ct * CP);
  const int co_base = set * a.cout_set + nt * 16;
  const int co_lim = min(16, a.cout_set - nt * 16);
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  const int Do = a.d.Do, Ho = a.d.Ho, Wo = a.d.Wo;
  const long long odhw = (long long)Do * Ho * Wo;
  int wk = xcd_swizzle(bid_x, grid_x);
  const int tw = wk % a.tilesW; wk /= a.tilesW;
  const int th = wk % a.tilesH; wk /= a.tilesH;
  const int ds = wk % a.dsegs;
  const int n = wk / a.dsegs;
  const int oh0 = th * TH, ow0 = tw * 32;
  const int d_begin = ds * a.sd, d_end = min(Do, d_begin + a.sd);

  // ---- column role of this lane ----
  const int cil = nn % CP, rsel = nn / CP;
  const int rs = a.rs, cw = 4 / rs;                   // rows per K step, 16-byte chunks per row
  const int rsub = g4 / cw, chunk = g4 % cw;          // this lane's row within the step and chunk within the row
  const int nrg = TH / rs;                            // row groups (K steps) per plane
  int boff[TPK], bkd[TPK];                            // in-plane byte offset (row kh, channel, chunk g4+1) and kd per tile
#pragma unroll
  for (int t = 0; t < TPK; ++t) {
    int r9 = t * R + rsel;
    if (r9 > 8) r9 = 8;                               // column unused: any valid address
    bkd[t] = r9 / 3;
    boff[t] = cil * CHS + (r9 % 3 + rsub) * ROWB + (chunk + 1) * 16;
  }
  // ---- A operand source: dY row of channel co_base + nn ----
  const bool a_ok = nn < co_lim;
  const ST* dyp = (const ST*)a.p.ea + n * a.d.ea_bs + (long long)(co_base + (a_ok ? nn : 0)) * odhw + ow0 + chunk * 8;

  // ---- staging plan ----
  const ST* sp_src[NIT];
  float sp_sc[NIT], sp_sh[NIT];
  int sp_lds[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = tid + it * NT;
    const int gi = item % 6;
    int r = item / 6;
    const int hy = r % IH;
    const int cl = r / IH;
    const int gh = oh0 - 1 + hy, gw = ow0 + (gi - 1) * 8;
    const int c = ci0 + cl;
    sp_src[it] = nullptr;
    sp_sc[it] = 0.f; sp_sh[it] = 0.f;
    sp_lds[it] = item < NITEM ? cl * CHS + hy * ROWB + gi * 16 : -1;
    if (item < NITEM && cl < ci_lim && (unsigned)gh < (unsigned)H && gw >= 0 && gw < W) {
      sp_src[it] = (c < a.d.Ca ? (const ST*)a.p.xa + n * a.d.xa_bs + (long long)c * dhw
                               : (const ST*)a.p.xb + n * a.d.xb_bs + (long long)(c - a.d.Ca) * dhw) +
                   (long long)gh * W + gw;
      sp_sc[it] = 1.f;
      if (a.d.pre) { sp_sc[it] = a.p.pre_sc[n * a.d.Cin + c]; sp_sh[it] = a.p.pre_sh[n * a.d.Cin + c]; }
    }
  }
  uint4 raw[NIT];
  auto load_plane = [&](int gd) {
    const bool dok = (unsigned)gd < (unsigned)D;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      raw[it] = make_uint4(0, 0, 0, 0);
      if (dok && sp_src[it]) raw[it] = *reinterpret_cast<const uint4*>(sp_src[it] + (long long)gd * hw);
    }
  };
  auto store_plane = [&](int gd) {
    const int slot = ((gd + 4) & 3) * PLANE;
    const bool dok = (unsigned)gd < (unsigned)D;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (sp_lds[it] < 0) continue;
      const float sc = sp_sc[it], sh = (dok && sp_src[it]) ? sp_sh[it] : 0.f;
      const unsigned u[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
      unsigned o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float lo = cvt_lo<FMT>(u[k]) * sc + sh, hi = cvt_hi<FMT>(u[k]) * sc + sh;
        lo = fmaxf(lo, lo * a.d.pre_slope);           // leaky for 0 <= slope <= 1
        hi = fmaxf(hi, hi * a.d.pre_slope);
        o[k] = cvt_pack<FMT>(lo, hi);
      }
      *reinterpret_cast<uint4*>(s_in + slot + sp_lds[it]) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  };

  f32x4 acc[3][TPK];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int t = 0; t < TPK; ++t) acc[kw][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbsum = 0.f;

  load_plane(d_begin - 1);
  store_plane(d_begin - 1);
  load_plane(d_begin);
  store_plane(d_begin);
  load_plane(d_begin + 1);
  // The A operand (dY rows) comes straight from global memory: fetch it one plane ahead so its latency hides behind the
  // previous plane's MFMAs instead of stalling every row.
  constexpr int RPW = (TH + NWV - 1) / NWV;
  uint4 a_nxt[RPW];
  auto load_dy = [&](int d) {
#pragma unroll
    for (int ri = 0; ri < RPW; ++ri) {
      const int rg = wv + ri * NWV;
      const int oh = oh0 + rg * rs + rsub;
      a_nxt[ri] = make_uint4(0, 0, 0, 0);
      if (a_ok && rg < nrg && oh < Ho) a_nxt[ri] = *reinterpret_cast<const uint4*>(dyp + ((long long)d * Ho + oh) * Wo);
    }
  };
  load_dy(d_begin);
  for (int d = d_begin; d < d_end; ++d) {
    store_plane(d + 1);
    __syncthreads();
    if (d + 1 < d_end) load_plane(d + 2);
    uint4 a_cur[RPW];
#pragma unroll
    for (int ri = 0; ri < RPW; ++ri) a_cur[ri] = a_nxt[ri];
    if (d + 1 < d_end) load_dy(d + 1);
    const int sbase = d + 3;
#pragma unroll
    for (int ri = 0; ri < RPW; ++ri) {
      const int rg = wv + ri * NWV;
      const int rr = rg * rs;                           // first row of the step (block-uniform per wave)
      if (rg >= nrg || oh0 + rr >= Ho) continue;
      // A fragment: 8 voxels of dY
      const uint4 araw = a_cur[ri];
      const bf16x8 av = __builtin_bit_cast(bf16x8, araw);
      {
        const unsigned u[4] = {araw.x, araw.y, araw.z, araw.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) dbsum += cvt_lo<FMT>(u[k]) + cvt_hi<FMT>(u[k]);
      }
#pragma unroll
      for (int t = 0; t < TPK; ++t) {
        const unsigned char* bp = s_in + ((sbase + bkd[t]) & 3) * PLANE + rr * ROWB + boff[t];
        const uint4 cur = *reinterpret_cast<const uint4*>(bp);
        const unsigned prv = *reinterpret_cast<const unsigned*>(bp - 4);
        const unsigned nxt = *reinterpret_cast<const unsigned*>(bp + 16);
        // kw = 0: window starts one voxel (2 bytes) earlier; kw = 2: one voxel later
        uint4 b0, b2;
        b0.x = __builtin_amdgcn_alignbyte(cur.x, prv, 2);
        b0.y = __builtin_amdgcn_alignbyte(cur.y, cur.x, 2);
        b0.z = __builtin_amdgcn_alignbyte(cur.z, cur.y, 2);
        b0.w = __builtin_amdgcn_alignbyte(cur.w, cur.z, 2);
        b2.x = __builtin_amdgcn_alignbyte(cur.y, cur.x, 2);
        b2.y = __builtin_amdgcn_alignbyte(cur.z, cur.y, 2);
        b2.z = __builtin_amdgcn_alignbyte(cur.w, cur.z, 2);
        b2.w = __builtin_amdgcn_alignbyte(nxt, cur.w, 2);
        acc[0][t] = mfma16x16x32<FMT>(av, __builtin_bit_cast(bf16x8, b0), acc[0][t]);
        acc[1][t] = mfma16x16x32<FMT>(av, __builtin_bit_cast(bf16x8, cur), acc[1][t]);
        acc[2][t] = mfma16x16x32<FMT>(av, __builtin_bit_cast(bf16x8, b2), acc[2][t]);
      }
    }
  }
  // ---- reduce across waves in LDS (layout of the weight tensor slice [16 co][CP ci][27]) ----
  // Every wave STORES its accumulators into a slice of its own and the slices are summed by the threads that issue the global
  // atomics.  (Round 6: this was 36 ds_add_f32 per lane into one shared slice.  LDS floating-point atomics retire about one lane
  // per two cycles: the 18 432 of a workgroup took 17 of the 26 us of a 16^3 problem -- measured by switching the tail off.)
  constexpr int NSL = 16 * CP * 27 + 16;                 // floats per slice: [16 co][CP ci][27] + 16 bias sums
  constexpr bool SLICES = (size_t)NWV * NSL * sizeof(float) <= 64 * 1024;
  __syncthreads();
  if constexpr (SLICES) {
    float* my = s_dw + wv * NSL;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int t = 0; t < TPK; ++t) {
        const int r9 = t * R + rsel;
        if (r9 < 9) {
#pragma unroll
          for (int r = 0; r < 4; ++r) my[((g4 * 4 + r) * CP + cil) * 27 + r9 * 3 + kw] = acc[kw][t][r];
        }
      }
    dbsum += __shfl_xor(dbsum, 16, 64);
    dbsum += __shfl_xor(dbsum, 32, 64);
    if (lane < 16) my[16 * CP * 27 + lane] = dbsum;
  } else {
    for (int i = tid; i < NSL; i += NT) s_dw[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int t = 0; t < TPK; ++t) {
        const int r9 = t * R + rsel;
        if (r9 < 9) {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(&s_dw[((g4 * 4 + r) * CP + cil) * 27 + r9 * 3 + kw], acc[kw][t][r]);
        }
      }
    dbsum += __shfl_xor(dbsum, 16, 64);
    dbsum += __shfl_xor(dbsum, 32, 64);
    if (lane < 16) atomicAdd(&s_dw[16 * CP * 27 + lane], dbsum);
  }
  __syncthreads();
  auto total = [&](int i) {
    if constexpr (SLICES) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) v += s_dw[w * NSL + i];
      return v;
    } else {
      return s_dw[i];
    }
  };
  const int gpp = a.d.groups / a.d.n_wptr;
  for (int i = tid; i < 16 * CP * 27; i += NT) {
    const int tap = i % 27;
    const int r = i / 27;
    const int cl = r % CP, co_l = r / CP;
    if (co_l >= co_lim || cl >= ci_lim) continue;
    const int co = co_base + co_l, ci = ci0 + cl;
    const int g = co / a.Cout_g;
    if (ci / a.Cin_g != g) continue;                  // off the block diagonal
    float* dst = a.dw[g / gpp] + ((long long)((g % gpp) * a.Cout_g + co % a.Cout_g) * a.Cin_g + ci % a.Cin_g) * 27 + tap;
    atomicAdd(dst, total(i));
  }
  if (ct == 0 && tid < co_lim) {
    const int co = co_base + tid, g = co / a.Cout_g;
    float* dbp = a.db[g / gpp];
    if (dbp) atomicAdd(&dbp[(g % gpp) * a.Cout_g + co % a.Cout_g], total(16 * CP * 27 + tid));
  }
}

template <int FMT, int CP, int NT>
__global__ __launch_bounds__(NT, 2) void conv3_wgrad_mfma_kernel(const WgMK a) {
  conv3_wgrad_body<FMT, CP, NT>(a, blockIdx.x, blockIdx.y, gridDim.x);
}

// Several weight-gradient problems in ONE launch.  The weight gradients of a backward pass are off its critical path
// (nothing reads them before the optimizer / the all-reduce), and the small-volume ones are latency-bound on their own --
// a few dozen workgroups each, 35 us whatever the shape.  Deferred to the end of backward and launched together, their
// workgroups fill the chip side by side.  The problem table travels in the kernel arguments (no device-side table, no
// host-to-device copy: capture-safe); a workgroup finds its problem from the prefix of workgroup counts.
constexpr int WG_MULTI = 7;
struct WgMulti {
  int n;
  int off[WG_MULTI + 1];      // first linear workgroup of problem i: a multiple of 8, so that the workgroup -> XCD dealing
                              // (round robin over the launch) is the one the problem's own XCD remap assumes
  int gx[WG_MULTI];           // grid x extent of problem i
  int nb[WG_MULTI];           // workgroups of problem i (gx * gy); the padding up to off[i + 1] exits at once
  WgMK p[WG_MULTI];
};
template <int FMT, int CP, int NT>
__global__ __launch_bounds__(NT, 2) void conv3_wgrad_mfma_multi_kernel(const WgMulti m) {
  const int b = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < WG_MULTI; ++k)
    if (k < m.n && b >= m.off[k]) i = k;
  const int local = b - m.off[i];
  if (local >= m.nb[i]) return;                          // alignment padding
  conv3_wgrad_body<FMT, CP, NT>(m.p[i], local % m.gx[i], local / m.gx[i], m.gx[i]);
}

struct WgPlan { WgMK a; unsigned gx; int ny; size_t shm; bool big; int cp; };
// fills the launch plan; returns false when the shape is not eligible for the MFMA weight gradient
static bool wg_plan(const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR], WgPlan* pl) {
  if ((d->dtype != XH_BF16 && d->dtype != XH_F16) || d->k != 3 || d->stride != 1 || d->bcast) return false;
  if ((d->W % 32 != 0 && d->W != 16 && d->W != 8) || d->Wo != d->W) return false;
  const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  if (cin_g < 4) return false;
  if ((d->xa_bs & 7) || (d->xb_bs & 7) || (d->ea_bs & 7)) return false;
  if (((long long)d->D * d->H * d->W) % 8) return false;
  if (d->pre && !(d->pre_slope >= 0.f && d->pre_slope <= 1.f)) return false;
  WgMK& a = pl->a;
  a.d = *d; a.p = *p;
  if (!d->pre) a.d.pre_slope = 1.f;
  for (int i = 0; i < XH_MAX_WPTR; ++i) { a.dw[i] = i < d->n_wptr ? dw[i] : nullptr; a.db[i] = (i < d->n_wptr && db) ? db[i] : nullptr; }
  a.Cin_g = cin_g; a.Cout_g = cout_g;
  int gs = 1;
  while (gs * 2 <= d->groups && d->groups % (gs * 2) == 0 && gs * 2 * cin_g <= 16 && gs * 2 * cout_g <= 16) gs *= 2;
  a.gs = gs;
  a.cin_set = gs * cin_g;
  a.cout_set = gs * cout_g;
  // Input channels per workgroup.  4 wins at every shape of this network (measured, tools/microbench_conv.py --wgrad
  // --cp): 4x more workgroups on the small volumes (16->16 @32^3: 27 us vs 87 us with 16) and a 4x smaller LDS
  // reduction + atomics tail per workgroup, for 36 instead of 27 MFMAs per 16 channels.  Wider tiles stay selectable
  // for experiments (xh_set_option(1, 128 | 256)).
  int cp = 4;
  {
    extern int g_mfma_abl;
    if ((g_mfma_abl & 128) && a.cin_set > 4) cp = 8;
    if ((g_mfma_abl & 256) && a.cin_set > 8) cp = 16;
  }
  pl->cp = cp;
  a.ntile = cdiv(a.cout_set, 16);
  a.nctile = cdiv(a.cin_set, cp);
  const int ny = (d->groups / gs) * a.ntile * a.nctile;
  if (ny > 65535) return false;
  a.rs = d->W >= 32 ? 1 : 32 / d->W;
  a.tilesW = cdiv(d->W, 32); a.tilesH = cdiv(d->Ho, 8);
  const int cols = a.tilesW * a.tilesH;
  // small volumes: fewer, longer runs so the per-workgroup LDS reduction + atomics pass amortises
  const bool big_vol = (long long)d->Do * d->Ho * d->Wo >= (1 << 20);
  int dsegs = cdiv(big_vol ? 512 : 128, cols * ny * d->N);
  const int max_segs = d->Do >= 4 ? d->Do / 4 : 1;
  if (dsegs > max_segs) dsegs = max_segs;
  if (dsegs < 1) dsegs = 1;
  a.sd = cdiv(d->Do, dsegs);
  a.dsegs = cdiv(d->Do, a.sd);
  const long long gx = (long long)cols * a.dsegs * d->N;
  if (gx > 2147483647LL) return false;
  pl->gx = (unsigned)gx;
  pl->ny = ny;
  const size_t ring = (size_t)4 * cp * (10 * 96 + 16);
  // one slice per wave when that fits 64 KB (the kernel's SLICES), else the single atomically summed slice
  const size_t slice = (size_t)(16 * cp * 27 + 16) * sizeof(float), nwv = (big_vol ? 256 : 512) / 64;
  const size_t red = nwv * slice <= 64 * 1024 ? nwv * slice : slice;
  pl->shm = ring > red ? ring : red;
  pl->big = big_vol;
  return true;
}

// returns XH_OK if launched, 1 if not eligible
int xh_conv3_wgrad_mfma_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR]) {
  {                                                   // few channels per group: the quad-channel kernel (conv3d_wgrad_q4.hip)
    WgQ4 q;
    if (xh_wgrad_q4_plan(d, p, dw, db, &q)) {
      xh_wgrad_q4_launch((hipStream_t)stream, d->dtype == XH_F32 ? 2 : d->dtype == XH_F16 ? 1 : 0, &q, 1);
      return xh_launch_status();
    }
  }
  if (d->bcast) return 1;                              // (a broadcast input is read by the full-row quad-channel kernel only)
  WgPlan pl;
  if (!wg_plan(d, p, dw, db, &pl)) return 1;
  const WgMK& a = pl.a;
  dim3 grid(pl.gx, pl.ny, 1);
  const size_t shm = pl.shm;
  hipStream_t st = (hipStream_t)stream;
  const bool big = pl.big;
  const int cp = pl.cp;
#define LW(C)                                                                                     \
  do {                                                                                            \
    if (d->dtype == XH_F16) {                                                                     \
      if (big) hipLaunchKernelGGL((conv3_wgrad_mfma_kernel<1, C, 256>), grid, dim3(256), shm, st, a);  \
      else hipLaunchKernelGGL((conv3_wgrad_mfma_kernel<1, C, 512>), grid, dim3(512), shm, st, a);      \
    } else {                                                                                      \
      if (big) hipLaunchKernelGGL((conv3_wgrad_mfma_kernel<0, C, 256>), grid, dim3(256), shm, st, a);  \
      else hipLaunchKernelGGL((conv3_wgrad_mfma_kernel<0, C, 512>), grid, dim3(512), shm, st, a);      \
    }                                                                                             \
  } while (0)
  xh_note_kernel("conv3_wgrad_mfma_kernel<%d, %d, %d>", d->dtype == XH_F16 ? 1 : 0, cp, big ? 256 : 512);
  switch (cp) {
    case 4: LW(4); break;
    case 8: LW(8); break;
    default: LW(16);
  }
#undef LW
  return xh_launch_status();
}

// Weight gradients of `n` convolutions (see xlstm_hved.h).  Problems the MFMA kernel can take (k = 3, stride 1, 16-bit
// storage, >= 4 input channels per group, the default 4-channel tiles) are grouped by (storage format, volume class) and
// launched WG_MULTI at a time; the others go through xh_conv3d_wgrad one by one.
extern "C" int xh_conv3d_wgrad(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR]);
extern int g_use_mfma;
int xh_c1w_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, float* const (*dw)[XH_MAX_WPTR],
                 float* const (*db)[XH_MAX_WPTR], char* handled);
int xh_tiny_wgrad_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, float* const (*dw)[XH_MAX_WPTR],
                        float* const (*db)[XH_MAX_WPTR], char* handled);
int xh_wg7_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, float* const (*dw)[XH_MAX_WPTR],
                 float* const (*db)[XH_MAX_WPTR], char* handled);
int xh_s2w_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, float* const (*dw)[XH_MAX_WPTR],
                 float* const (*db)[XH_MAX_WPTR], char* handled);
extern "C" int xh_conv3d_wgrad_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p,
                                     float* const (*dw)[XH_MAX_WPTR], float* const (*db)[XH_MAX_WPTR]) {
  if (n < 0 || (n > 0 && (!d || !p || !dw))) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int rc_all = XH_OK;
  // quad-channel problems first: WQ_MULTI per launch and storage format
  std::vector<char> handled(n > 0 ? n : 1, 0);
  if (g_use_mfma) {
    for (int cls = 0; cls < 12; ++cls) {              // (storage format: bf16 / fp16 / fp32 with fp16 operands, input-channel quads per group);
                                                      // 9, 10, 11: the full-row kernel's problems (bf16 / fp16 / fp32, conv3d_wgrad_q5.hip)
      const int full = cls >= 9, fmt = full ? cls - 9 : cls / 3, ci4 = full ? 1 : cls % 3 + 1;
      std::vector<WgQ4> cl;
      WgQ4 q;
      for (int i = 0; i < n; ++i) {
        if (!d[i] || !p[i]) return XH_ERR_ARG;
        if (handled[i] || (d[i]->dtype == XH_F32 ? 2 : d[i]->dtype == XH_F16 ? 1 : 0) != fmt) continue;
        if (!xh_wgrad_q4_plan(d[i], p[i], dw[i], db ? db[i] : nullptr, &q) || q.ci4 != ci4 || q.full != full) continue;
        handled[i] = 1;
        cl.push_back(q);
      }
      if (cl.empty()) continue;
      // as few launches as the problem table allows, of EQUAL work: in arrival order 17 problems went 8 + 8 + 1, and the last
      // launch -- one 128^3 problem alone on the chip -- took as long as a full one (77 / 152 / 78 us).  Largest first into the
      // currently lightest launch
      const int per = full ? Q5_MULTI : WQ_MULTI;
      const int L = ((int)cl.size() + per - 1) / per;
      auto cost = [](const WgQ4& w) {                   // (full-row units of several quads: as xh_wgrad_q5_launch prices them)
        const double uf = !w.full || w.uqx * w.uqy == 1 ? 1.0 : w.uqx * w.uqy == 2 ? 1.45 : w.uqx == 3 ? 2.0 : 1.9;
        return (double)w.N * w.D * w.H * (w.full ? (w.W == 128 ? 128.0 * uf : w.W == 64 ? 90.0 * uf : 70.0 * uf) : (double)w.W) * w.nq * w.ci4;
      };
      std::vector<int> order(cl.size());
      for (size_t i = 0; i < cl.size(); ++i) order[i] = (int)i;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(cl[a]) > cost(cl[b]); });
      std::vector<std::vector<WgQ4>> bucket(L);
      std::vector<double> load(L, 0.0);
      for (int idx : order) {
        int best = -1;
        for (int b = 0; b < L; ++b)
          if ((int)bucket[b].size() < per && (best < 0 || load[b] < load[best])) best = b;
        bucket[best].push_back(cl[idx]);
        load[best] += cost(cl[idx]);
      }
      if (full && L > 1) {
        // full-row launches: a launch lasts as long as its slowest workgroup under the min-max plan (xh_wgrad_q5_plan), which is not
        // additive in the problems -- improve the greedy split by single moves and swaps while the sum of planned durations falls
        extern int g_q5_wgs;
        auto planned = [&](const std::vector<WgQ4>& v) { int wq[Q5_MULTI]; return v.empty() ? 0.0 : xh_wgrad_q5_plan(v.data(), (int)v.size(), g_q5_wgs, wq); };
        std::vector<double> T(L);
        for (int b = 0; b < L; ++b) T[b] = planned(bucket[b]);
        for (int iter = 0; iter < 32; ++iter) {
          double gain = 1e-9; int ba = -1, bb = -1, ia = -1, ib = -1;
          for (int a = 0; a < L; ++a)
            for (int b2 = 0; b2 < L; ++b2) {
              if (a == b2) continue;
              for (int i = 0; i < (int)bucket[a].size(); ++i) {
                for (int j = -1; j < (int)bucket[b2].size(); ++j) {     // j = -1: move i to b2; else swap i <-> j (a < b2 only)
                  if (j < 0 ? ((int)bucket[b2].size() >= per || bucket[a].size() <= 1) : a > b2) continue;
                  std::vector<WgQ4> va = bucket[a], vb = bucket[b2];
                  if (j < 0) { vb.push_back(va[i]); va.erase(va.begin() + i); }
                  else std::swap(va[i], vb[j]);
                  const double g = T[a] + T[b2] - planned(va) - planned(vb);
                  if (g > gain) { gain = g; ba = a; bb = b2; ia = i; ib = j; }
                }
              }
            }
          if (ba < 0) break;
          if (ib < 0) { bucket[bb].push_back(bucket[ba][ia]); bucket[ba].erase(bucket[ba].begin() + ia); }
          else std::swap(bucket[ba][ia], bucket[bb][ib]);
          T[ba] = planned(bucket[ba]); T[bb] = planned(bucket[bb]);
        }
      }
      for (int b = 0; b < L; ++b) xh_wgrad_q4_launch(st, fmt, bucket[b].data(), (int)bucket[b].size());
    }
  }
  {                                                   // k = 1 problems: conv1x1_wgrad_multi_kernel (conv3d.hip)
    const int rc = xh_c1w_batch(stream, n, d, p, dw, db, handled.data());
    if (rc != XH_OK) rc_all = rc;
    const int rc2 = xh_s2w_batch(stream, n, d, p, dw, db, handled.data());      // stride-2 k = 3 problems likewise
    if (rc2 != XH_OK) rc_all = rc2;
    const int rc4 = xh_tiny_wgrad_batch(stream, n, d, p, dw, db, handled.data());    // 1 <-> 2 channel k = 3 stencils (conv3_tiny.hip)
    if (rc4 != XH_OK) rc_all = rc4;
    const int rc3 = g_use_mfma ? xh_wg7_batch(stream, n, d, p, dw, db, handled.data()) : XH_OK;   // the 7^3 gate convs (conv7_wgrad_mfma.hip)
    if (rc3 != XH_OK) rc_all = rc3;
  }
  WgMulti* m = new WgMulti;
  for (int cls = 0; cls < 4; ++cls) {                 // (fmt, big)
    const int fmt = cls >> 1, big = cls & 1;
    m->n = 0; m->off[0] = 0;
    size_t shm = 0;
    auto flush = [&]() {
      if (m->n == 0) return;
      const int total = m->off[m->n];
      xh_note_kernel("conv3_wgrad_mfma_multi_kernel<%d, 4, %d>", fmt, big ? 256 : 512);
      if (fmt) {
        if (big) hipLaunchKernelGGL((conv3_wgrad_mfma_multi_kernel<1, 4, 256>), dim3(total), dim3(256), shm, st, *m);
        else hipLaunchKernelGGL((conv3_wgrad_mfma_multi_kernel<1, 4, 512>), dim3(total), dim3(512), shm, st, *m);
      } else {
        if (big) hipLaunchKernelGGL((conv3_wgrad_mfma_multi_kernel<0, 4, 256>), dim3(total), dim3(256), shm, st, *m);
        else hipLaunchKernelGGL((conv3_wgrad_mfma_multi_kernel<0, 4, 512>), dim3(total), dim3(512), shm, st, *m);
      }
      m->n = 0; m->off[0] = 0; shm = 0;
    };
    for (int i = 0; i < n; ++i) {
      if (!d[i] || !p[i]) { delete m; return XH_ERR_ARG; }
      if (handled[i]) continue;
      WgPlan pl;
      const bool ok = g_use_mfma && !d[i]->transposed && p[i]->ea && wg_plan(d[i], p[i], dw[i], db ? db[i] : nullptr, &pl) && pl.cp == 4;
      if (!ok) {
        if (cls == 0) {                               // not batchable: the ordinary entry point, once
          const int rc = xh_conv3d_wgrad(stream, d[i], p[i], dw[i], db ? db[i] : nullptr);
          if (rc != XH_OK) rc_all = rc;
        }
        continue;
      }
      if ((d[i]->dtype == XH_F16 ? 1 : 0) != fmt || (pl.big ? 1 : 0) != big) continue;
      const long long blocks = (long long)pl.gx * pl.ny;
      if (blocks > (1 << 24)) {                       // too large to share a launch: on its own
        if (xh_conv3_wgrad_mfma_try(stream, d[i], p[i], dw[i], db ? db[i] : nullptr) != XH_OK) rc_all = XH_ERR_HIP;
        continue;
      }
      const int k = m->n;
      m->p[k] = pl.a;
      m->gx[k] = (int)pl.gx;
      m->nb[k] = (int)blocks;
      m->off[k + 1] = m->off[k] + (int)((blocks + 7) / 8 * 8);
      if (pl.shm > shm) shm = pl.shm;
      m->n = k + 1;
      if (m->n == WG_MULTI) flush();
    }
    flush();
  }
  delete m;
  if (rc_all != XH_OK) return rc_all;
  return xh_launch_status();
}
