// Persistent, software-pipelined variant of the quad-channel 3x3x3 convolution (conv3d_q4.hip) for the launches that have
// several tiles per workgroup slot -- the 128^3-class convs of XLSTM_HVED (RA_HVED.py:510-648; semantics of
// buildingblocks.py:406-433).
//
// Why: conv3_q4_kernel is a chain per workgroup -- request the tile's inputs, wait, build the LDS image, barrier, matrix phase,
// request the epilogue operand, wait, store -- and a launch of 4 096 workgroups runs it in four rounds of 1 024 resident ones.
// tools/microbench_big.py --abl (operands rotated through 1.6 GB so that they come from HBM, 16 -> 16 g4 @128^3 data gradient):
// 60.6 us, of which 29.5 us remain when the kernel stops after the LDS image is built and NOTHING changes when the MFMAs are
// removed: the time is the two exposed memory round trips per workgroup (+ its launch), four workgroups per CU are not enough
// to hide them, and the SIMDs sit in s_waitcnt (SQ_WAIT_ANY 41 %, MFMA busy 12 %).
//
// Here a workgroup is resident for the whole launch and takes every (G / 8)-th tile of its XCD's eighth of the tile sequence
// (the workgroups of an XCD sweep its slab together, so halos are re-read from its own L2).  A "stage" is one (tile, input quad):
//   * the global loads of stage s + 1 are issued right after stage s has been transformed into the LDS image -- their
//     destination registers are free from then on -- and stay in flight under the barrier, the matrix phase and the whole
//     epilogue of stage s;
//   * the weight fragments of a stage and the epilogue operand of its tile (EPI == 1) are requested at the START of the stage,
//     BEFORE the next stage's operands: vmcnt retires in order, so waiting for them does not wait for the younger prefetch
//     (requested after it, `s_waitcnt vmcnt(0)` in front of the matrix phase drained the prefetch: 80 us instead of 60);
//   * the weight fragments are requested at the start of a stage, the epilogue operand behind its transform, both BEFORE the prefetch;
//   * every load is unconditional (a load under a branch makes hipcc's wait counts conservative on every path that joins);
//   * statistics: fp32 per tile, fp64 across the tiles of one (sample, output quad), one set of fp64 atomics per run.
// 150 - 230 VGPRs: two workgroups per CU (three spill, and a spill reload waits for vmcnt(0)).
// WHAT IT MEASURED: the memory waits go (SQ_WAIT_ANY -60 %) and single-quad tiles get SLOWER (16 -> 16 g4 data gradient
// 59.7 -> 67.5 us): the saturated unit was the texture path (half-line requests), which conv3d_q4w.hip then fixed with full-row
// tiles.  This kernel remains for 8-plane multi-quad forward launches on rows that are not 64 / 128 voxels wide.
#include "conv_q4.h"

int g_q4_persist = 1;             // xh_set_option(19, n): 0 = never, 1 = where it wins (xh_conv3_q4p_try), 2 = every 8-plane launch

template <int FMT> __device__ __forceinline__ f32x2_t q4p_xf(unsigned u, float sc, float sh, float slope) {
  const f32x2_t v = cvt2_in<FMT>(u) * f32x2_t{sc, sc} + f32x2_t{sh, sh};
  return max2(v, v * f32x2_t{slope, slope});
}

template <int FMT, int PRE, int EPI, bool MULTI, int OCC>
__global__ __launch_bounds__(256, OCC) void conv3_q4p_kernel(const ConvQ4 a, const int ntile_sp, const int ntotal) {
  typedef h16<FMT> ST;
  constexpr int TD = 8, ID = TD + 2, TILE_BYTES = ID * PLANE, NROWS = ID * IH, NITEM = NROWS * 4, NEDGE = NROWS * 2;
  constexpr int NIT = (NITEM + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* s_red = reinterpret_cast<double*>(smem + TILE_BYTES);      // [4 waves][8], then [8] totals
  float* s_fin = reinterpret_cast<float*>(smem + TILE_BYTES + 48 * sizeof(double));   // [2][Q4_MAXC]: in-kernel InstanceNorm scale / shift

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nn = lane & 15, g4 = lane >> 4;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  const long long dhw_b = dhw * (long long)sizeof(ST);
  const int Do = a.d.Do, Ho = a.d.Ho;
  const int OQ = a.d.Cout / 4;
  // this workgroup's tiles: XCD x (= blockIdx % 8) owns the slab [x T / 8, (x + 1) T / 8) of the tile sequence and its G / 8
  // workgroups sweep through it TOGETHER (workgroup slot s takes slab tiles s, s + G/8, s + 2 G/8, ...): the tiles in flight at
  // any time are neighbours, as in a launch of one workgroup per tile, so the halo rows and planes a tile shares with its H / D
  // neighbours are still in the XCD's L2.  (Contiguous runs per workgroup put the H neighbour 4 steps = 14 MB of streamed data
  // later: TCC_MISS 1.03 M -> 2.35 M per launch, 60 -> 79 us.)
  const int G = gridDim.x, b = blockIdx.x;
  const int nx = (G & 7) ? 1 : 8, xcd = (G & 7) ? 0 : (b & 7), slot = (G & 7) ? b : (b >> 3), step = G / nx;
  const int s0 = (int)((long long)ntotal * xcd / nx), s1 = (int)((long long)ntotal * (xcd + 1) / nx);
  const int t0 = s0 + slot, t1 = s1;                 // tiles t0, t0 + step, ... < t1
  if (t0 >= t1) return;

  // ---- tile-independent part of the staging plan ----
  int i_dz[NIT], i_hy[NIT], i_gq8[NIT], i_lds[NIT], i_par[NIT];
  bool i_do[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = tid + it * 256;
    i_do[it] = item < NITEM;
    const int gq = item & 3, row = min(item >> 2, NROWS - 1);
    const int dz = row / IH, hy = row - dz * IH;
    i_dz[it] = dz - 1; i_hy[it] = hy - 1; i_gq8[it] = gq * 8;
    i_lds[it] = row * PITCH;
    i_par[it] = (row & 1) | ((1 + 4 * gq) << 1);
  }
  const bool e_do = tid >= 256 - NEDGE;
  int e_dz, e_hy, e_side, e_lds;
  {
    const int ei = max(255 - tid, 0) < NEDGE ? 255 - tid : 0;
    const int row = ei >> 1, side = ei & 1;
    const int dz = row / IH, hy = row - dz * IH;
    e_dz = dz - 1; e_hy = hy - 1; e_side = side;
    e_lds = row * PITCH + (((side ? 17 : 0) ^ (row & 1)) << 4);
  }
  const int ur = (nn >> 2) & 1;
  const int qw = (nn >> 3) | ((nn & 3) << 1);
  const int rowbase = (2 * wv + ur) * PITCH;
  const int b_off0 = rowbase + ((2 * qw + g4) ^ ur) * 16;
  const int b_off1 = rowbase + ((2 * qw + g4) ^ ur ^ 1) * 16;
  const long long odhw = (long long)Do * Ho * a.d.Wo;
  const unsigned spd_b = (unsigned)(Ho * a.d.Wo) * (unsigned)sizeof(ST);
  const float pslope = a.d.pre_slope, eslope = a.d.e_slope;
  const bool fin = PRE == 1 && a.p.fin_red != nullptr;
  const int ncq = MULTI ? a.ci4 : 1;

  // a tile's coordinates and the tile-dependent part of the plan (byte offsets inside a channel volume, liveness bits)
  struct Plan { int n, oq, od0, oh0, ow0; unsigned off[NIT], e_off, live; };
  auto make_plan = [&](int t, Plan& p) {
    const int q = t / ntile_sp, sp = t - q * ntile_sp;
    p.n = q / OQ; p.oq = q - p.n * OQ;
    const int wk1 = udiv_fast(sp, a.tilesW, a.mW), tw = sp - wk1 * a.tilesW;
    const int td = udiv_fast(wk1, a.tilesH, a.mH), th = wk1 - td * a.tilesH;
    p.od0 = td * TD; p.oh0 = th * TH; p.ow0 = tw * TW;
    unsigned live = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int gd = p.od0 + i_dz[it], gh = p.oh0 + i_hy[it];
      live |= ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H) ? 1u << it : 0u;
      const int gdc = min(max(gd, 0), D - 1), ghc = min(max(gh, 0), H - 1);
      p.off[it] = (unsigned)((((long long)gdc * H + ghc) * W + p.ow0 + i_gq8[it]) * (long long)sizeof(ST));
    }
    {
      const int gd = p.od0 + e_dz, gh = p.oh0 + e_hy;
      const int gw = e_side ? p.ow0 + TW : p.ow0 - 2;
      live |= ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw >= 0 && gw < W) ? 1u << NIT : 0u;
      const int gdc = min(max(gd, 0), D - 1), ghc = min(max(gh, 0), H - 1), gwc = min(max(gw, 0), W - 2);
      p.e_off = (unsigned)((((long long)gdc * H + ghc) * W + gwc) * (long long)sizeof(ST));
    }
    p.live = live;
  };
  auto src_of = [&](const Plan& p, int cq) -> const char* {
    const int grp = udiv_fast(p.oq, a.oq_g, a.mQ);
    const int c0 = grp * a.Cin_g + cq * 4;
    return reinterpret_cast<const char*>(c0 < a.d.Ca ? (const ST*)a.p.xa + p.n * a.d.xa_bs + (long long)c0 * dhw
                                                      : (const ST*)a.p.xb + p.n * a.d.xb_bs + (long long)(c0 - a.d.Ca) * dhw);
  };
  // the staged operands of ONE stage: every global load of the thread, requested back to back
  uint4 raw[NIT][4];
  unsigned eraw4[4];
  float nsc[4] = {1.f, 1.f, 1.f, 1.f}, nsh[4] = {0.f, 0.f, 0.f, 0.f};     // PRE == 1 without fused finalisation: scale / shift of the stage
  auto issue = [&](const Plan& p, int cq) {
    const char* src = src_of(p, cq);
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) raw[it][cc] = *reinterpret_cast<const uint4*>(src + cc * dhw_b + p.off[it]);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) eraw4[cc] = *reinterpret_cast<const unsigned*>(src + cc * dhw_b + p.e_off);
    if (PRE == 1 && !fin) {
      const int grp = udiv_fast(p.oq, a.oq_g, a.mQ);
      const int c0 = grp * a.Cin_g + cq * 4;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) { nsc[cc] = a.p.pre_sc[p.n * a.d.Cin + c0 + cc]; nsh[cc] = a.p.pre_sh[p.n * a.d.Cin + c0 + cc]; }
    }
  };

  Plan cur, nxt;
  make_plan(t0, cur);
  issue(cur, 0);
  nxt = cur;

  frag8 wfrag[9];
  int fkey = -1;
  double run_s = 0.0, run_q = 0.0;                   // statistics of the current (sample, output quad) run, lanes nn == 0
  const f32x2_t esl2 = {eslope, eslope};

  for (int t = t0; t < t1; t += step) {
    const int n = cur.n, oq = cur.oq, co0 = oq * 4;
    const int grp = udiv_fast(oq, a.oq_g, a.mQ);
    const int cin_base = grp * a.Cin_g;
    const int od0 = cur.od0, oh0 = cur.oh0, ow0 = cur.ow0;
    // ---- epilogue lane role of this tile ----
    const int oh = oh0 + 2 * wv + ur;
    const bool row_ok = oh < Ho;
    const int ndz = min(TD, Do - od0);
    float bias, esc = 0.f, esh = 0.f;
    {
      const int wp = udiv_fast(grp, a.gpp, a.mG);
      const float* bp = a.p.b[wp];                    // (no load under a branch: without a bias, a word of the weight workspace times 0)
      const float* bq = bp ? bp + (grp - wp * a.gpp) * a.Cout_g + (oq - grp * a.oq_g) * 4 + g4 : reinterpret_cast<const float*>(a.p.ws);
      bias = *bq;
      if (!bp) bias = 0.f;
    }
    const unsigned lane_b = (unsigned)(((long long)g4 * odhw + (long long)(row_ok ? oh : 0) * a.d.Wo + ow0 + 4 * qw) * (long long)sizeof(ST));
    const unsigned lane_bo = row_ok ? lane_b : Q4_OOB;
    __amdgpu_buffer_rsrc_t ers = q4_window(a.p.y), yrs;
    if (EPI == 1) {
      esc = a.p.e_sc[n * a.d.Cout + co0 + g4];
      esh = a.p.e_sh[n * a.d.Cout + co0 + g4];
      ers = q4_window(reinterpret_cast<const char*>(co0 < a.d.Cea ? (const ST*)a.p.ea + n * a.d.ea_bs + (long long)co0 * odhw
                                                                    : (const ST*)a.p.eb + n * a.d.eb_bs + (long long)(co0 - a.d.Cea) * odhw) +
                      (long long)od0 * spd_b);
    }
    yrs = q4_window(reinterpret_cast<char*>((ST*)a.p.y + n * a.d.y_bs + (long long)co0 * odhw) + (long long)od0 * spd_b);

    // ---- fused InstanceNorm finalisation of this group's input channels (xh_conv_ptrs.fin_red), once per (sample, group) ----
    if (fin && fkey != n * a.d.groups + grp) {
      fkey = n * a.d.groups + grp;
      const float* gam = a.p.fin_gamma;
      const float* bet = a.p.fin_beta;
      __syncthreads();                                // nobody still reads the previous group's coefficients
      if (tid < a.Cin_g) {
        float m_, r_, sc_, sh_;
        in_finalize(a.p.fin_red[2 * (n * a.d.Cin + cin_base + tid)], a.p.fin_red[2 * (n * a.d.Cin + cin_base + tid) + 1], a.fin_inv, sc_, sh_, m_, r_);
        if (gam) { const float g_ = gam[cin_base + tid]; sc_ *= g_; sh_ = fmaf(sh_, g_, bet[cin_base + tid]); }
        s_fin[tid] = sc_; s_fin[Q4_MAXC + tid] = sh_;
      }
      if (b == 0 && t == t0)                          // the workgroup of tile 0: the coefficients of ALL channels for the backward pass
        for (int i = tid; i < a.d.N * a.d.Cin; i += 256) {
          float sc_, sh_, m_, r_;
          in_finalize(a.p.fin_red[2 * i], a.p.fin_red[2 * i + 1], a.fin_inv, sc_, sh_, m_, r_);
          if (gam) {                                  // N == 1: i is the channel
            sc_ *= gam[i]; sh_ = fmaf(sh_, gam[i], bet[i]);
            if (a.p.fin_rm && a.p.fin_rv && a.p.fin_steps > 0) {
              const double M = 1.0 / a.fin_inv, mean = a.p.fin_red[2 * i] * a.fin_inv;
              double var = a.p.fin_red[2 * i + 1] * a.fin_inv - mean * mean;
              if (var < 0) var = 0;
              const double keep = pow(0.9, (double)a.p.fin_steps), unb = var * M / (M > 1 ? M - 1 : 1);
              a.p.fin_rm[i] = (float)(keep * a.p.fin_rm[i] + (1 - keep) * mean);
              a.p.fin_rv[i] = (float)(keep * a.p.fin_rv[i] + (1 - keep) * unb);
            }
          }
          const_cast<float*>(a.p.pre_sc)[i] = sc_; const_cast<float*>(a.p.pre_sh)[i] = sh_;
          a.p.fin_mean[i] = m_; a.p.fin_rstd[i] = r_;
        }
      __syncthreads();
    }

    f32x4 acc[TD];
#pragma unroll
    for (int i = 0; i < TD; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint2 eraw[TD];

    for (int cq = 0; cq < ncq; ++cq) {
      const bool last_cq = MULTI ? cq + 1 == ncq : true;
      // ---- what THIS stage needs after its barrier is requested first: the memory counter retires in order, so everything
      // requested before the next stage's operands can be waited for without waiting for those ----
      {
        const frag8* wpk = reinterpret_cast<const frag8*>(a.p.ws) + ((long long)oq * a.ci4 + cq) * 9 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 9; ++i) wfrag[i] = wpk[i * 64];
      }
      // ---- stage (t, cq): its operands were requested one stage ago ----
      float sc[4], sh[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        if (PRE == 1 && fin) { sc[cc] = s_fin[cq * 4 + cc]; sh[cc] = s_fin[Q4_MAXC + cq * 4 + cc]; }
        else { sc[cc] = nsc[cc]; sh[cc] = nsh[cc]; }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (!i_do[it]) continue;
        const bool lv_ = (cur.live >> it) & 1;
        const int par = i_par[it] & 1, slot0 = i_par[it] >> 1;
        uint4 outv[4];
        if (PRE) {
          const float lv = lv_ ? 1.f : 0.f;
          f32x2_t v[4][4];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float s1 = sc[cc] * lv, s2 = sh[cc] * lv;
            const unsigned u[4] = {raw[it][cc].x, raw[it][cc].y, raw[it][cc].z, raw[it][cc].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) v[cc][k] = q4p_xf<FMT>(u[k], s1, s2, pslope);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            outv[j].x = cvt2_pack<FMT>(v[0][j].x, v[1][j].x);
            outv[j].y = cvt2_pack<FMT>(v[2][j].x, v[3][j].x);
            outv[j].z = cvt2_pack<FMT>(v[0][j].y, v[1][j].y);
            outv[j].w = cvt2_pack<FMT>(v[2][j].y, v[3][j].y);
          }
        } else {
          const unsigned se = lv_ ? 0x05040100u : 0x0c0c0c0cu, so = lv_ ? 0x07060302u : 0x0c0c0c0cu;
          const unsigned u[4][4] = {{raw[it][0].x, raw[it][0].y, raw[it][0].z, raw[it][0].w},
                                    {raw[it][1].x, raw[it][1].y, raw[it][1].z, raw[it][1].w},
                                    {raw[it][2].x, raw[it][2].y, raw[it][2].z, raw[it][2].w},
                                    {raw[it][3].x, raw[it][3].y, raw[it][3].z, raw[it][3].w}};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            outv[j].x = __builtin_amdgcn_perm(u[1][j], u[0][j], se);
            outv[j].y = __builtin_amdgcn_perm(u[3][j], u[2][j], se);
            outv[j].z = __builtin_amdgcn_perm(u[1][j], u[0][j], so);
            outv[j].w = __builtin_amdgcn_perm(u[3][j], u[2][j], so);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<uint4*>(smem + i_lds[it] + (((slot0 + j) ^ par) << 4)) = outv[j];
      }
      if (e_do) {
        const bool lv_ = (cur.live >> NIT) & 1;
        uint4 o;
        if (PRE) {
          const float lv = lv_ ? 1.f : 0.f;
          f32x2_t v[4];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) v[cc] = q4p_xf<FMT>(eraw4[cc], sc[cc] * lv, sh[cc] * lv, pslope);
          o.x = cvt2_pack<FMT>(v[0].x, v[1].x);
          o.y = cvt2_pack<FMT>(v[2].x, v[3].x);
          o.z = cvt2_pack<FMT>(v[0].y, v[1].y);
          o.w = cvt2_pack<FMT>(v[2].y, v[3].y);
        } else {
          const unsigned se = lv_ ? 0x05040100u : 0x0c0c0c0cu, so = lv_ ? 0x07060302u : 0x0c0c0c0cu;
          o.x = __builtin_amdgcn_perm(eraw4[1], eraw4[0], se);
          o.y = __builtin_amdgcn_perm(eraw4[3], eraw4[2], se);
          o.z = __builtin_amdgcn_perm(eraw4[1], eraw4[0], so);
          o.w = __builtin_amdgcn_perm(eraw4[3], eraw4[2], so);
        }
        *reinterpret_cast<uint4*>(smem + e_lds) = o;
      }
      // ---- the next stage's operands: in flight from here until its own transform ----
      // the tile's epilogue operand: behind the transform (its registers are not live there), in front of the prefetch
      if (EPI == 1) {                                 // (stages before the tile's last one: out-of-window offsets, no traffic, same count)
        const unsigned eo = last_cq ? lane_b : Q4_OOB;
#pragma unroll
        for (int dz = 0; dz < TD; ++dz)
          eraw[dz] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(ers, (int)eo, (int)((unsigned)min(dz, ndz - 1) * spd_b), 0));
      }
      // (ALWAYS requested, through ONE instruction sequence: hipcc's s_waitcnt bookkeeping takes the minimum over the paths
      // that join, so a path without the prefetch -- last tile of the run, or a branch per input quad -- made every later
      // wait drain it.  The run's last stage re-requests its own tile; nobody reads that.)
      if (last_cq) make_plan(t + step < t1 ? t + step : t, nxt);
      {
        Plan q = cur;
        if (last_cq) q = nxt;
        issue(q, last_cq ? 0 : cq + 1);
      }
      __syncthreads();                                // the image of stage (t, cq) is complete
#pragma unroll
      for (int pz = 0; pz < ID; ++pz) {
        frag8 bf[3];
        bf[0] = *reinterpret_cast<const frag8*>(smem + pz * PLANE + b_off0);
        bf[1] = *reinterpret_cast<const frag8*>(smem + pz * PLANE + PITCH + b_off1);
        bf[2] = *reinterpret_cast<const frag8*>(smem + pz * PLANE + 2 * PITCH + b_off0);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
          const int dz = pz - kd;
          if (dz < 0 || dz >= TD) continue;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) acc[dz] = mfma16x16x32<FMT>(wfrag[kd * 3 + kh], bf[kh], acc[dz]);
        }
      }
      __syncthreads();                                // every wave is done reading the image: the next stage may overwrite it
    }

    // ---- epilogue of the tile ----
    f32x2_t ps = {0.f, 0.f}, pq = {0.f, 0.f};
    const f32x2_t bias2 = {bias, bias}, esc2 = {esc, esc}, esh2 = {esh, esh};
#pragma unroll
    for (int dz = 0; dz < TD; ++dz) {
      const bool live = dz < ndz;                     // uniform
      f32x2_t v[2] = {f32x2_t{acc[dz][0], acc[dz][1]} + bias2, f32x2_t{acc[dz][2], acc[dz][3]} + bias2};
      uint2 pk;
      if (EPI == 1) {
        const f32x2_t e[2] = {cvt2_in<FMT>(eraw[dz].x), cvt2_in<FMT>(eraw[dz].y)};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x2_t z = e[q] * esc2 + esh2, vs = v[q] * esl2;
          v[q] = f32x2_t{z.x > 0.f ? v[q].x : vs.x, z.y > 0.f ? v[q].y : vs.y};
        }
        pk.x = cvt2_pack<FMT>(v[0].x, v[0].y); pk.y = cvt2_pack<FMT>(v[1].x, v[1].y);
        if (live) {
          const f32x2_t r0 = cvt2_in<FMT>(pk.x), r1 = cvt2_in<FMT>(pk.y);
          ps += r0 + r1;
          pq += r0 * e[0] + r1 * e[1];
        }
      } else {
        pk.x = cvt2_pack<FMT>(v[0].x, v[0].y); pk.y = cvt2_pack<FMT>(v[1].x, v[1].y);
        if (EPI == 2 && live) {
          const f32x2_t r0 = cvt2_in<FMT>(pk.x), r1 = cvt2_in<FMT>(pk.y);
          ps += r0 + r1;
          pq += r0 * r0 + r1 * r1;
        }
      }
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(2))), pk), yrs,
                                            (int)(live ? lane_bo : Q4_OOB), (int)((unsigned)dz * spd_b), 0);
    }
    if (EPI) {
      // the 16 lanes of a DPP row hold one channel of one wave's two rows; fp64 from the tile level on
      const float s0 = row16_sum(row_ok ? ps.x + ps.y : 0.f), s1 = row16_sum(row_ok ? pq.x + pq.y : 0.f);
      run_s += (double)s0; run_q += (double)s1;
      const bool flush = t + step >= t1 || nxt.n != n || nxt.oq != oq;      // uniform
      if (flush) {
        if (nn == 0) { s_red[wv * 8 + g4 * 2] = run_s; s_red[wv * 8 + g4 * 2 + 1] = run_q; }
        run_s = 0.0; run_q = 0.0;
        __syncthreads();
        if (tid < 8)
          atomicAdd(&a.p.red[((long long)n * a.d.Cout + co0 + (tid >> 1)) * 2 + (tid & 1)],
                    s_red[tid] + s_red[8 + tid] + s_red[16 + tid] + s_red[24 + tid]);
        __syncthreads();
      }
    }
    cur = nxt;
  }
}

// workgroups of a persistent launch: at most `slots` resident ones, every one with (nearly) the same number of tiles
static int q4p_grid(long long ntotal, int slots) {
  const long long per = (ntotal + slots - 1) / slots;
  long long g = (ntotal + per - 1) / per;
  g = (g + 7) & ~7ll;
  if (g > slots) g = slots;
  return (int)g;
}

// XH_OK if launched, 1 if this launch stays with conv3_q4_kernel
int xh_conv3_q4p_try(hipStream_t st, const ConvQ4& a) {
  if (!g_q4_persist || a.td != 8 || a.act_slope != 1.f || a.d.pre == 2 || a.d.dtype == XH_F32) return 1;
  const int ntile_sp = a.tilesW * a.tilesH * a.tilesD;
  const long long ntotal = (long long)ntile_sp * (a.d.Cout / 4) * a.d.N;
  if (ntotal >= (1ll << 30)) return 1;
  // default: the launches it was measured to win (below): several input quads per tile, forward (producer norm on load)
  if (g_q4_persist == 1 && (a.ci4 < 2 || a.d.pre != 1 || ntotal * a.ci4 < 2048)) return 1;
  // workgroups per CU: 2 (no instance spills at 256 VGPRs; a spill reload is a scratch load that waits for vmcnt(0) and drains the
  // prefetch).  Ablation bit 32768: 3 per CU (168 VGPRs: the single-quad instances without the norm-backward epilogue fit)
  const int occ = (a.abl & 32768) ? 3 : 2;
  const int grid = q4p_grid(ntotal, 256 * occ);
  const size_t shm = q4_tile_bytes(8) + 48 * sizeof(double) + 3 * Q4_MAXC * sizeof(float);
  const int f = a.d.dtype == XH_F16 ? 1 : 0;
  const bool multi = a.ci4 > 1;
  xh_note_kernel("conv3_q4p_kernel<%d, %d, %d, %s, %d>", f, a.d.pre, a.d.epi, multi ? "true" : "false", occ);
#define QPO(F, P, E, M)                                                                                                         \
  do {                                                                                                                          \
    if (occ == 3) hipLaunchKernelGGL((conv3_q4p_kernel<F, P, E, M, 3>), dim3(grid), dim3(256), shm, st, a, ntile_sp, (int)ntotal); \
    else hipLaunchKernelGGL((conv3_q4p_kernel<F, P, E, M, 2>), dim3(grid), dim3(256), shm, st, a, ntile_sp, (int)ntotal);          \
  } while (0)
#define QPL(F, P, E)              \
  do {                            \
    if (multi) QPO(F, P, E, true); \
    else QPO(F, P, E, false);     \
  } while (0)
#define QPE(F, P)                        \
  do {                                   \
    if (a.d.epi == 0) QPL(F, P, 0);      \
    else if (a.d.epi == 1) QPL(F, P, 1); \
    else QPL(F, P, 2);                   \
  } while (0)
  if (f) { if (a.d.pre) QPE(1, 1); else QPE(1, 0); }
  else { if (a.d.pre) QPE(0, 1); else QPE(0, 0); }
#undef QPE
#undef QPL
#undef QPO
  return xh_launch_status();
}
