// Full-row variant of the quad-channel 3x3x3 convolution (conv3d_q4.hip) for volumes whose rows are 128 voxels wide -- the
// 128^3 patches of XLSTM_HVED (RA_HVED.py:510-648; semantics of buildingblocks.py:406-433), where the conv stages spend their
// time.
//
// Why (counters of conv3_q4_kernel on HBM-resident operands): its 8 x 8 x 32 tile reads, per staged row and channel, one
// 64-byte HALF of a 128-byte line plus two 4-byte edge voxels that sit in lines of their own -- 1 200 line requests for a tile
// image of 28.8 KB, the texture path busy 59 % of the launch and stalled by the cache for most of that, while neither more
// resident waves nor prefetching across tiles (conv3d_q4p.hip) changed the 3 us per tile and CU.
//
// Here a tile spans the WHOLE row: 4 planes x 8 rows x 128 voxels, one workgroup of 8 waves.
//   * no W halo at all: the two voxels left and right of a row are the conv's zero padding (two constant LDS slots per staged
//     row); every global load is a 16-byte piece of a FULL line, 8 consecutive lanes per line: 60 rows x 4 channels x 2 lines =
//     480 line requests for 4 096 output voxels (2 400 for the same voxels before);
//   * D / H halo 6 x 10 planes-rows for 4 x 8 (1.875; 1.56 before) -- those re-reads are L2 hits of the neighbouring tiles;
//   * wave w owns output row w and both 64-voxel halves of it (two N tiles of the W-Toeplitz GEMM): per staged plane 6 LDS reads and
//     up to 18 MFMAs; same products in the same order as conv3_q4_kernel: the outputs are BIT-IDENTICAL;
//   * LDS image [plane][row][66 slots of 16 bytes] (slot = 2 voxels x 4 channels + two padding slots), slots and the lane -> quad
//     assignment permuted so that both the staging writes and the fragment reads are bank-conflict free (q4w_slot below).
// (Measured and dropped, round 5: requesting the EPI == 1 operand in front of the last matrix phase instead of at the start of the
// epilogue -- 16 more live registers, 128 with 4 spilled: 16 -> 16 g4 data gradient 49.4 -> 57.1 us, 4 -> 4 16.8 -> 18.8, 12 -> 4 38.4 -> 44.5.)
// (Measured and dropped: 2 output planes per tile on launches of at most one resident round -- twice the workgroups, but the D halo
// goes from 1.5 to 2 staged planes per output plane: 4 -> 4 @128^3 16.0 -> 19.9 us, 12 -> 4 30.8 -> 37.2, 8 -> 8 @64^3 9.8 -> 13.0.)
// Same template axes as conv3_q4_kernel except: no activation epilogue, no norm-backward-on-load (pre == 2 lives on tensors below
// 2^22 elements, i.e. not on 128-wide rows); 63 KB of LDS, two workgroups per CU.
#include "conv_q4.h"

int g_q4_wide = 3;                // xh_set_option(20, mask): bit 1 = rows of 128 voxels, bit 0 = rows of 64 voxels take conv3_q4w_kernel;
                                  // bit 2: NO workgroups of three output quads (the 4 -> 12 data gradients stage their tile per output quad)

namespace {
constexpr int WTD = 4, WTH = 8;              // output planes / rows per workgroup
constexpr int WID = WTD + 2, WIH = WTH + 2;
constexpr int WNROWS = WID * WIH;            // 60 staged rows
// NH = 64-voxel halves per row: 2 (rows of 128 voxels) or 1 (rows of 64)
template <int NH> struct QW {
  static constexpr int WW = 64 * NH;                     // row width the instance is built for
  static constexpr int SLOTS = WW / 2 + 2;               // 66 | 34
  static constexpr int PITCH = SLOTS * 16;               // 1056 | 544 bytes per staged row
  static constexpr int PLANE = WIH * PITCH;
  static constexpr int TILE = WID * PLANE;               // 63 360 | 32 640 bytes
  static constexpr int PR = 8 * NH;                      // 8-voxel pieces per row
  static constexpr int NITEM = WNROWS * PR;              // (row, piece)
  static constexpr int NIT = (NITEM + 511) / 512;        // 2 | 1
};
}
// LDS layout of a staged row.  Logical slot s = 16 bytes = voxels 2 s - 2, 2 s - 1 (x 4 channels); s = 0 and s = SLOTS - 1 are the
// conv's zero padding.  Bank rules (MI355X_MICROARCH.md, LDS): ds_write_b128 is served in 8 groups of 8 CONTIGUOUS lanes on 32
// banks (8 slots), ds_read_b128 in 4 groups of 16 lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... on 64 banks (16 slots).
//  * the 8-voxel piece j of a row is the logical slots 1 + 4 j + k, k = 0..3: physical slot 32 (j >> 3) + 8 k + (j & 7): the k-th
//    write of 8 consecutive pieces covers 8 consecutive slots -- conflict-free;
//  * lane (nn, g4) of a fragment read takes quad q = q4w_quad(nn) -- even quads on lanes {0-3, 12-15}, odd quads on {4-11} -- and
//    reads logical slot 32 h + 2 q + g4: inside every read group the 8 lanes of either g4 land on 8 different (j & 7) and the two
//    g4 on different k parities, i.e. on the two halves of the 16-slot bank row -- conflict-free (one 2-way hit per padding read);
//  * the padding slots sit behind the data (physical SLOTS - 2, SLOTS - 1).
// (The first version XORed bit 4 into the slot index, which suits CONTIGUOUS 16-lane groups: SQ_LDS_BANK_CONFLICT 2.09 M cycles per
// launch against 0.97 M LDS-active cycles.)
template <int SLOTS> __device__ __forceinline__ int q4w_slot(int s) {
  const int t = s - 1;
  return s == 0 ? SLOTS - 2 : s == SLOTS - 1 ? SLOTS - 1 : (t & ~31) + 8 * (t & 3) + ((t >> 2) & 7);
}
__device__ __forceinline__ int q4w_quad(int nn) { return nn < 4 ? 2 * nn : nn < 12 ? 2 * (nn - 4) + 1 : 2 * (nn - 8); }

template <int FMT> __device__ __forceinline__ f32x2_t q4w_xf(unsigned u, float sc, float sh, float slope) {
  const f32x2_t v = cvt2_in<FMT>(u) * f32x2_t{sc, sc} + f32x2_t{sh, sh};
  return max2(v, v * f32x2_t{slope, slope});
}

// The body takes its block coordinates as arguments (bx, by, bz of a grid gdx x gdy x N): conv3_q4w_kernel passes the launch's own,
// conv3_q4w_pair_kernel those of the problem a workgroup belongs to -- the same arithmetic on the same data either way.
// BC: a broadcast operand (xh_conv_desc.bcast) -- a template argument so that the ordinary instances carry none of its registers: as
// run-time flags they cost the dominant data-gradient instance <0, 0, 1, false, 2> six spilled registers (128 of 128 in use).
// NOQ (round 6): output quads of ONE group a workgroup produces from one staged image, one after the other -- the 4 -> 12 data
// gradients of the first decoder convs staged the same 4-channel tile once per output quad (three workgroups, three transforms of
// 60 rows); the matrix phase and the epilogue of an output quad are what they were, so the outputs are the same bits.  Single input
// quad only (the image of a later input quad would overwrite the one the next output quad needs).
template <int FMT, int PRE, int EPI, bool MULTI, int NH, bool BC = false, int NOQ = 1>
__device__ __forceinline__ void q4w_body(const ConvQ4& a, unsigned char* smem, const int bx, const int by, const int bz, const int gdx, const int gdy) {
  static_assert(NOQ == 1 || !MULTI, "several output quads per workgroup: one input quad");
  typedef h16<FMT> ST;
  typedef QW<NH> Q;
  constexpr int WW = Q::WW, WSLOTS = Q::SLOTS, WPITCH = Q::PITCH, WPLANE = Q::PLANE, WTILE = Q::TILE, WNITEM = Q::NITEM, WNIT = Q::NIT;
  double* s_red = reinterpret_cast<double*>(smem + WTILE);             // [8 waves][8], then [8] totals + the fan-in flag
  float* s_fin = reinterpret_cast<float*>(smem + WTILE + 80 * sizeof(double));   // [2][Q4_MAXC]: in-kernel InstanceNorm scale / shift

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nn = lane & 15, g4 = lane >> 4;
  const int qd = q4w_quad(nn);                       // the quad (4 consecutive voxels of a 64-voxel half) this lane's N column is
  const int oq0 = by * NOQ, n = bz;
  const int grp = udiv_fast(oq0, a.oq_g, a.mQ);
  const int cin_base = grp * a.Cin_g;
  const int D = a.d.D, H = a.d.H;
  const long long hw = (long long)H * WW, dhw = (long long)D * hw;
  const int Do = a.d.Do, Ho = a.d.Ho;
  const int tilesH = (Ho + WTH - 1) / WTH;
  const int wk = xcd_swizzle(bx, gdx);
  const int td = wk / tilesH, th = wk - td * tilesH;
  const int od0 = td * WTD, oh0 = th * WTH;
  // raw InstanceNorm sums of this group's input channels (fused finalisation): requested first, used behind the staging loads
  double fs1 = 0.0, fs2 = 0.0;
  if (PRE == 1 && a.p.fin_red && tid < a.Cin_g) {
    fs1 = a.p.fin_red[2 * (n * a.d.Cin + cin_base + tid)];
    fs2 = a.p.fin_red[2 * (n * a.d.Cin + cin_base + tid) + 1];
  }
  // the padding slots of every staged row: zero for the whole launch (the staging never writes them)
  if (tid < WNROWS * 2) {
    const int row = tid >> 1, s = (tid & 1) ? WSLOTS - 1 : 0;
    *reinterpret_cast<uint4*>(smem + row * WPITCH + q4w_slot<WSLOTS>(s) * 16) = make_uint4(0, 0, 0, 0);
  }

  // ---- staging plan (the same for every input-channel quad): item = (row, piece j of 8 voxels) ----
  unsigned i_off[WNIT];
  int i_lds[WNIT];              // LDS byte address of the piece's first voxel pair; the k-th pair: + 128 k
  bool i_live[WNIT], i_do[WNIT];
#pragma unroll
  for (int it = 0; it < WNIT; ++it) {
    const int item = tid + it * 512;
    i_do[it] = item < WNITEM;
    const int j = item & (Q::PR - 1), row = min(item / Q::PR, WNROWS - 1);
    const int dz = row / WIH, hy = row - dz * WIH;
    const int gd = od0 - 1 + dz, gh = oh0 - 1 + hy;
    i_live[it] = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H;
    const int gdc = min(max(gd, 0), D - 1), ghc = min(max(gh, 0), H - 1);
    i_off[it] = (unsigned)((((long long)gdc * H + ghc) * WW + j * 8) * (long long)sizeof(ST));
    i_lds[it] = row * WPITCH + (((j >> 3) << 5) + (j & 7)) * 16;   // physical slot 32 (j >> 3) + (j & 7) + 8 k
  }
  // ---- B (data) fragments: lane (quad nn, k-group g4) of half h reads logical slot 32 h + 2 nn + g4 of input row wv + kh ----
  int b_off[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) b_off[h] = q4w_slot<WSLOTS>(32 * h + 2 * qd + g4) * 16;
  // ---- epilogue lane role: lane (quad, channel) owns 4 consecutive voxels of output row oh0 + wv in each half ----
  const int oh = oh0 + wv;
  const bool row_ok = oh < Ho;
  const int ndz = min(WTD, Do - od0);
  const long long odhw = (long long)Do * Ho * WW;
  const unsigned spd_b = (unsigned)(Ho * WW) * (unsigned)sizeof(ST);
  const unsigned lane_b = (unsigned)(((long long)g4 * odhw + (long long)(row_ok ? oh : 0) * WW + 4 * qd) * (long long)sizeof(ST));
  // broadcast operands (xh_conv_desc.bcast): the forward input / the data gradient's e operand is ONE stored channel per group, seen
  // through four (scale, shift) pairs; a data gradient without y only sums (every store lands outside the window and is dropped)
  const bool bc_in = BC && !a.d.transposed, bc_e = BC && a.d.transposed;
  const bool st_ok = BC ? (row_ok && a.p.y != nullptr) : row_ok;
  const unsigned lane_e = bc_e ? (unsigned)(((long long)(row_ok ? oh : 0) * WW + 4 * qd) * (long long)sizeof(ST)) : lane_b;
  const unsigned lane_bo = st_ok ? lane_b : Q4_OOB;
  constexpr unsigned HALF_B = 64 * sizeof(ST);         // byte distance of the two halves of a row
  const int ncq = MULTI ? a.ci4 : 1;
  const float pslope = a.d.pre_slope;
  const bool fin = PRE == 1 && a.p.fin_red != nullptr;
  const long long dhw_b = dhw * (long long)sizeof(ST);
  const long long cs_b = bc_in ? 0 : dhw_b;              // byte distance of the quad's channel planes (0: one stored channel)

#pragma unroll 1
  for (int oi = 0; oi < NOQ; ++oi) {                     // (NOQ == 1: the body as it always was)
  const int oq = oq0 + oi;
  const int co0 = oq * 4;
  float bias = 0.f, esc = 0.f, esh = 0.f, ectr = 0.f;
  {
    const int wp = udiv_fast(grp, a.gpp, a.mG);
    const float* bp = a.p.b[wp];
    if (bp) bias = bp[(grp - wp * a.gpp) * a.Cout_g + (oq - grp * a.oq_g) * 4 + g4];
  }
  __amdgpu_buffer_rsrc_t ers = q4_window(a.p.y), yrs;
  if (EPI == 1) {
    esc = a.p.e_sc[n * a.d.Cout + co0 + g4];
    esh = a.p.e_sh[n * a.d.Cout + co0 + g4];
    if (BC && bc_e && a.p.e_ctr) ectr = a.p.e_ctr[n * a.d.Cout + co0 + g4];
    ers = q4_window(reinterpret_cast<const char*>(bc_e ? (const ST*)a.p.ea + n * a.d.ea_bs + (long long)(co0 >> 2) * odhw
                                                  : co0 < a.d.Cea ? (const ST*)a.p.ea + n * a.d.ea_bs + (long long)co0 * odhw
                                                                  : (const ST*)a.p.eb + n * a.d.eb_bs + (long long)(co0 - a.d.Cea) * odhw) +
                    (long long)od0 * spd_b);
  }
  yrs = q4_window(reinterpret_cast<char*>((ST*)a.p.y + n * a.d.y_bs + (long long)co0 * odhw) + (long long)od0 * spd_b);

  f32x4 acc[WTD][NH];
#pragma unroll
  for (int i = 0; i < WTD; ++i)
#pragma unroll
    for (int h = 0; h < NH; ++h) acc[i][h] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int cq = 0; cq < ncq; ++cq) {
    const int c0 = cin_base + cq * 4;
    if (NOQ == 1 || oi == 0) {                           // (a later output quad of the workgroup finds the image staged)
    const char* src = reinterpret_cast<const char*>(bc_in ? (const ST*)a.p.xa + n * a.d.xa_bs + (long long)(c0 >> 2) * dhw
                                                    : c0 < a.d.Ca ? (const ST*)a.p.xa + n * a.d.xa_bs + (long long)c0 * dhw
                                                                  : (const ST*)a.p.xb + n * a.d.xb_bs + (long long)(c0 - a.d.Ca) * dhw);
    // ---- all global loads of this thread, back to back ----
    uint4 raw[WNIT][4];
#pragma unroll
    for (int it = 0; it < WNIT; ++it)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) raw[it][cc] = *reinterpret_cast<const uint4*>(src + cc * cs_b + i_off[it]);
    if (cq > 0) __syncthreads();                      // every wave is done reading the previous quad's image
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (PRE == 1) {
      if (fin) {
        // fused InstanceNorm finalisation (xh_conv_ptrs.fin_red): as in conv3_q4_kernel
        if (cq == 0) {
          const float* gam = a.p.fin_gamma;
          const float* bet = a.p.fin_beta;
          if (tid < a.Cin_g) {
            float m_, r_, sc_, sh_;
            in_finalize(fs1, fs2, a.fin_inv, sc_, sh_, m_, r_);
            if (gam) { const float g_ = gam[cin_base + tid]; sc_ *= g_; sh_ = fmaf(sh_, g_, bet[cin_base + tid]); }
            s_fin[tid] = sc_; s_fin[Q4_MAXC + tid] = sh_;
          }
          if (bx == 0 && by == 0 && bz == 0)
            for (int i = tid; i < a.d.N * a.d.Cin; i += 512) {
              float sc_, sh_, m_, r_;
              in_finalize(a.p.fin_red[2 * i], a.p.fin_red[2 * i + 1], a.fin_inv, sc_, sh_, m_, r_);
              if (gam) {                                // N == 1: i is the channel
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
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { sc[cc] = s_fin[cq * 4 + cc]; sh[cc] = s_fin[Q4_MAXC + cq * 4 + cc]; }
      } else {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { sc[cc] = a.p.pre_sc[n * a.d.Cin + c0 + cc]; sh[cc] = a.p.pre_sh[n * a.d.Cin + c0 + cc]; }
      }
    }
    // ---- transform + channels-last LDS image ----
#pragma unroll
    for (int it = 0; it < WNIT; ++it) {
      if (!i_do[it]) continue;
      uint4 outv[4];                                  // slot k = voxels 2k, 2k+1 x 4 channels
      if (PRE) {
        const float lv = i_live[it] ? 1.f : 0.f;
        f32x2_t v[4][4];                              // [channel][voxel pair]
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float s1 = sc[cc] * lv, s2 = sh[cc] * lv;
          const unsigned u[4] = {raw[it][cc].x, raw[it][cc].y, raw[it][cc].z, raw[it][cc].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) v[cc][k] = q4w_xf<FMT>(u[k], s1, s2, pslope);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          outv[k].x = cvt2_pack<FMT>(v[0][k].x, v[1][k].x);
          outv[k].y = cvt2_pack<FMT>(v[2][k].x, v[3][k].x);
          outv[k].z = cvt2_pack<FMT>(v[0][k].y, v[1][k].y);
          outv[k].w = cvt2_pack<FMT>(v[2][k].y, v[3][k].y);
        }
      } else {
        const unsigned se = i_live[it] ? 0x05040100u : 0x0c0c0c0cu, so = i_live[it] ? 0x07060302u : 0x0c0c0c0cu;
        const unsigned u[4][4] = {{raw[it][0].x, raw[it][0].y, raw[it][0].z, raw[it][0].w},
                                  {raw[it][1].x, raw[it][1].y, raw[it][1].z, raw[it][1].w},
                                  {raw[it][2].x, raw[it][2].y, raw[it][2].z, raw[it][2].w},
                                  {raw[it][3].x, raw[it][3].y, raw[it][3].z, raw[it][3].w}};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          outv[k].x = __builtin_amdgcn_perm(u[1][k], u[0][k], se);
          outv[k].y = __builtin_amdgcn_perm(u[3][k], u[2][k], se);
          outv[k].z = __builtin_amdgcn_perm(u[1][k], u[0][k], so);
          outv[k].w = __builtin_amdgcn_perm(u[3][k], u[2][k], so);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<uint4*>(smem + i_lds[it] + 128 * k) = outv[k];
    }
    }
    // A (weight) fragments of this (output quad, input quad): issued here so that they travel while the workgroup gathers
    frag8 wfrag[9];
    {
      const frag8* wpk = reinterpret_cast<const frag8*>(a.p.ws) + ((long long)oq * a.ci4 + cq) * 9 * 64 + lane;
#pragma unroll
      for (int i = 0; i < 9; ++i) wfrag[i] = wpk[i * 64];
    }
    if (NOQ == 1 || oi == 0) __syncthreads();
    // ---- matrix phase: the wave's output row, both halves, walking the 6 staged planes once ----
#pragma unroll
    for (int pz = 0; pz < WID; ++pz) {
      // (kh outside kd: NH fragments live at a time instead of 3 NH; an accumulator still meets its taps in the same order --
      // for a given output plane the staged plane fixes kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        frag8 bf[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) bf[h] = *reinterpret_cast<const frag8*>(smem + pz * WPLANE + (wv + kh) * WPITCH + b_off[h]);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
          const int dz = pz - kd;
          if (dz < 0 || dz >= WTD) continue;
#pragma unroll
          for (int h = 0; h < NH; ++h) acc[dz][h] = mfma16x16x32<FMT>(wfrag[kd * 3 + kh], bf[h], acc[dz][h]);
        }
      }
    }
  }

  // ---- epilogue ----
  f32x2_t ps = {0.f, 0.f}, pq = {0.f, 0.f};
  const f32x2_t bias2 = {bias, bias}, esc2 = {esc, esc}, esh2 = {esh, esh};
  const f32x2_t esl2 = {a.d.e_slope, a.d.e_slope}, ectr2 = {ectr, ectr};
  uint2 eraw[WTD][NH];
  if (EPI == 1) {
#pragma unroll
    for (int dz = 0; dz < WTD; ++dz)
#pragma unroll
      for (int h = 0; h < NH; ++h)
        eraw[dz][h] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(ers, (int)(lane_e + h * HALF_B),
                                                                                      (int)((unsigned)min(dz, ndz - 1) * spd_b), 0));
  }
#pragma unroll
  for (int dz = 0; dz < WTD; ++dz) {
    const bool live = dz < ndz;                       // uniform
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      f32x2_t v[2] = {f32x2_t{acc[dz][h][0], acc[dz][h][1]} + bias2, f32x2_t{acc[dz][h][2], acc[dz][h][3]} + bias2};
      uint2 pk;
      if (EPI == 1) {
        const f32x2_t e[2] = {cvt2_in<FMT>(eraw[dz][h].x), cvt2_in<FMT>(eraw[dz][h].y)};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x2_t z = e[q] * esc2 + esh2, vs = v[q] * esl2;
          v[q] = f32x2_t{z.x > 0.f ? v[q].x : vs.x, z.y > 0.f ? v[q].y : vs.y};
        }
        pk.x = cvt2_pack<FMT>(v[0].x, v[0].y); pk.y = cvt2_pack<FMT>(v[1].x, v[1].y);
        if (live) {
          const f32x2_t r0 = cvt2_in<FMT>(pk.x), r1 = cvt2_in<FMT>(pk.y);         // the values as stored
          ps += r0 + r1;
          if (BC) pq += r0 * (e[0] - ectr2) + r1 * (e[1] - ectr2);      // (a broadcast e operand: the second sum around its centre)
          else pq += r0 * e[0] + r1 * e[1];
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
                                            (int)(live ? lane_bo + (st_ok ? h * HALF_B : 0u) : Q4_OOB), (int)((unsigned)dz * spd_b), 0);
    }
  }
  if (EPI) {
    // lanes of one channel: the 16 lanes nn = 0..15 of a lane group g4 = one DPP row
    const float t0 = row16_sum(row_ok ? ps.x + ps.y : 0.f), t1 = row16_sum(row_ok ? pq.x + pq.y : 0.f);
    if (nn == 0) { s_red[wv * 8 + g4 * 2] = (double)t0; s_red[wv * 8 + g4 * 2 + 1] = (double)t1; }
    __syncthreads();
    if (tid < 8) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) tot += s_red[w * 8 + tid];
      s_red[64 + tid] = tot;
    }
    double* s_tot = s_red + 64;
    if (a.fan && !fan_in<8>(a.fan + ((long long)n * gdy + oq) * FAN_UNIT_BYTES, bx, gdx, s_tot,
                            reinterpret_cast<int*>(s_tot + 8)))
      return;
    if (!a.fan) __syncthreads();
    if (tid < 8) atomicAdd(&a.p.red[((long long)n * a.d.Cout + co0 + (tid >> 1)) * 2 + (tid & 1)], s_tot[tid]);
  }
  }   // output quads of the workgroup
}

template <int FMT, int PRE, int EPI, bool MULTI, int NH, bool BC = false, int NOQ = 1>
__global__ __launch_bounds__(512, 4) void conv3_q4w_kernel(const ConvQ4 a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  q4w_body<FMT, PRE, EPI, MULTI, NH, BC, NOQ>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}
// Two independent convolutions of one shape in ONE launch (xh_conv3d_fwd_pair): grid z = 2 N, the first N planes of workgroups are
// problem 0.  The recon | seg streams' first decoder convs (buildingblocks.py:732-735: different inputs, the same shapes) are a
// single resident round each at 64^3 (256 workgroups on 256 CUs x 2 slots) -- together they fill the round.
template <int FMT, int PRE, int EPI, bool MULTI, int NH>
__global__ __launch_bounds__(512, 4) void conv3_q4w_pair_kernel(const ConvQ4 a0, const ConvQ4 a1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = a0.d.N;
  if ((int)blockIdx.z >= N) q4w_body<FMT, PRE, EPI, MULTI, NH>(a1, smem, blockIdx.x, blockIdx.y, (int)blockIdx.z - N, gridDim.x, gridDim.y);
  else q4w_body<FMT, PRE, EPI, MULTI, NH>(a0, smem, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// XH_OK if launched, 1 if the two plans do not share an instance / a grid (the caller launches them one by one)
int xh_conv3_q4w_pair_try(hipStream_t st, ConvQ4& a, ConvQ4& b) {
  const int nh = a.d.W == 128 ? 2 : a.d.W == 64 ? 1 : 0;
  if (!nh || !(g_q4_wide & nh) || a.d.dtype == XH_F32 || a.act_slope != 1.f || a.d.pre == 2 || a.d.bcast || b.d.bcast) return 1;
  if (a.d.dtype != b.d.dtype || a.d.W != b.d.W || a.d.H != b.d.H || a.d.D != b.d.D || a.d.N != b.d.N || a.d.Cout != b.d.Cout ||
      a.d.pre != b.d.pre || a.d.epi != b.d.epi || (a.ci4 > 1) != (b.ci4 > 1) || b.act_slope != 1.f || b.d.pre == 2) return 1;
  if (a.d.D < 4 || a.d.H < 8 || 2 * a.d.N > 65535) return 1;
  const int tilesD = (a.d.Do + WTD - 1) / WTD, tilesH = (a.d.Ho + WTH - 1) / WTH;
  dim3 grid(tilesD * tilesH, a.d.Cout / 4, 2 * a.d.N);
  a.fan = b.fan = nullptr;                              // (two problems in flight: direct atomics)
  const size_t shm = (nh == 2 ? QW<2>::TILE : QW<1>::TILE) + 80 * sizeof(double) + 2 * Q4_MAXC * sizeof(float);
  const int f = a.d.dtype == XH_F16 ? 1 : 0;
  const bool multi = a.ci4 > 1;
  xh_note_kernel("conv3_q4w_pair_kernel<%d, %d, %d, %s, %d>", f, a.d.pre, a.d.epi, multi ? "true" : "false", nh);
#define QPN(F, P, E, M)                                                                                        \
  do {                                                                                                         \
    if (nh == 2) hipLaunchKernelGGL((conv3_q4w_pair_kernel<F, P, E, M, 2>), grid, dim3(512), shm, st, a, b);   \
    else hipLaunchKernelGGL((conv3_q4w_pair_kernel<F, P, E, M, 1>), grid, dim3(512), shm, st, a, b);           \
  } while (0)
#define QPM(F, P, E)               \
  do {                             \
    if (multi) QPN(F, P, E, true); \
    else QPN(F, P, E, false);      \
  } while (0)
  // (instances: forward with output moments and the data gradient with norm-backward sums -- what the decoder pairs launch)
  if (a.d.pre == 1 && a.d.epi == 2) { if (f) QPM(1, 1, 2); else QPM(0, 1, 2); }
  else if (a.d.pre == 0 && a.d.epi == 1) { if (f) QPM(1, 0, 1); else QPM(0, 0, 1); }
  else return 1;
#undef QPM
#undef QPN
  return xh_launch_status();
}

// XH_OK if launched, 1 if this launch stays with conv3_q4_kernel.  `a` is a filled plan (q4_plan + pointers).
int xh_conv3_q4w_try(hipStream_t st, ConvQ4& a) {
  const int nh = a.d.W == 128 ? 2 : a.d.W == 64 ? 1 : 0;
  if (!nh || !(g_q4_wide & nh) || a.d.dtype == XH_F32 || a.act_slope != 1.f || a.d.pre == 2) return 1;
  if (a.d.D < 4 || a.d.H < 8) return 1;
  const int tilesD = (a.d.Do + WTD - 1) / WTD, tilesH = (a.d.Ho + WTH - 1) / WTH;
  dim3 grid(tilesD * tilesH, a.d.Cout / 4, a.d.N);
  const long long nwg = (long long)grid.x * grid.y * grid.z;
  // 512 workgroups are resident at once.  Rows of 64 voxels: launches between one and three rounds keep the 32-wide tiles, whose
  // 256-thread workgroups quantise better (20 -> 20 g5 @64^3: 640 workgroups = 1.25 rounds here, 12.6 -> 13.6 us; 8 -> 8, 16 -> 16 g2,
  // 24 -> 8 fit one round: 13.2 -> 10.1, 17.7 -> 13.6, 22.3 -> 18.6 us)
  if (nh == 1 && nwg > 512 && nwg < 1536 && !(a.abl & 262144) && !a.d.bcast) return 1;      // (a broadcast operand is read here only)
  a.fan = a.d.epi ? xh_fan_block(a.p.fan, a.p.fan_bytes, (long long)grid.y * grid.z, grid.x) : nullptr;
  // The fan-in ends every workgroup with a returning-atomic round trip: worth it when the launch is ONE resident round (512
  // workgroups: the tail of same-line atomics is exposed: 4 -> 4 @128^3 data gradient 18.3 -> 17.1 us), a loss when later rounds
  // would have covered the plain atomics of earlier ones (16 -> 16 g4 forward + moments 48.2 -> 43.6 us).  Bit 131072: never.
  if (nwg > 512 || (a.abl & 131072)) a.fan = nullptr;
  const size_t shm = (nh == 2 ? QW<2>::TILE : QW<1>::TILE) + 80 * sizeof(double) + 2 * Q4_MAXC * sizeof(float);
  const int f = a.d.dtype == XH_F16 ? 1 : 0;
  const bool multi = a.ci4 > 1;
  xh_note_kernel("conv3_q4w_kernel<%d, %d, %d, %s, %d>", f, a.d.pre, a.d.epi, multi ? "true" : "false", nh);
#define QWN(F, P, E, M)                                                                                    \
  do {                                                                                                     \
    if (nh == 2) hipLaunchKernelGGL((conv3_q4w_kernel<F, P, E, M, 2>), grid, dim3(512), shm, st, a);       \
    else hipLaunchKernelGGL((conv3_q4w_kernel<F, P, E, M, 1>), grid, dim3(512), shm, st, a);               \
  } while (0)
#define QWM(F, P, E)               \
  do {                             \
    if (multi) QWN(F, P, E, true); \
    else QWN(F, P, E, false);      \
  } while (0)
#define QWE(F, P)                        \
  do {                                   \
    if (a.d.epi == 0) QWM(F, P, 0);      \
    else if (a.d.epi == 1) QWM(F, P, 1); \
    else QWM(F, P, 2);                   \
  } while (0)
  if (a.d.bcast) {
    // broadcast operand: the two instances the init fold launches (forward + moments; data gradient + norm-backward sums), one quad per group
    if (multi || !((a.d.pre == 1 && a.d.epi == 2 && !a.d.transposed) || (a.d.pre == 0 && a.d.epi == 1 && a.d.transposed))) return XH_ERR_ARG;
#define QWB(F, P, E)                                                                                               \
  do {                                                                                                              \
    if (nh == 2) hipLaunchKernelGGL((conv3_q4w_kernel<F, P, E, false, 2, true>), grid, dim3(512), shm, st, a);      \
    else hipLaunchKernelGGL((conv3_q4w_kernel<F, P, E, false, 1, true>), grid, dim3(512), shm, st, a);              \
  } while (0)
    if (a.d.pre == 1) { if (f) QWB(1, 1, 2); else QWB(0, 1, 2); }
    else { if (f) QWB(1, 0, 1); else QWB(0, 0, 1); }
#undef QWB
    return xh_launch_status();
  }
  // several output quads of a group from one staged image (NOQ = 3): the data gradient of an n -> 4 conv with 3 k input quads
  // (4 -> 12: the decoders' first convs), single input quad, norm-backward sums by direct atomics
  if (!multi && a.d.pre == 0 && a.d.epi == 1 && a.oq_g % 3 == 0 && (g_q4_wide & 4) == 0) {
    grid.y = a.d.Cout / 12;
    a.fan = nullptr;
    if (f) { if (nh == 2) hipLaunchKernelGGL((conv3_q4w_kernel<1, 0, 1, false, 2, false, 3>), grid, dim3(512), shm, st, a);
             else hipLaunchKernelGGL((conv3_q4w_kernel<1, 0, 1, false, 1, false, 3>), grid, dim3(512), shm, st, a); }
    else { if (nh == 2) hipLaunchKernelGGL((conv3_q4w_kernel<0, 0, 1, false, 2, false, 3>), grid, dim3(512), shm, st, a);
           else hipLaunchKernelGGL((conv3_q4w_kernel<0, 0, 1, false, 1, false, 3>), grid, dim3(512), shm, st, a); }
    return xh_launch_status();
  }
  if (f) { if (a.d.pre) QWE(1, 1); else QWE(1, 0); }
  else { if (a.d.pre) QWE(0, 1); else QWE(0, 0); }
#undef QWE
#undef QWM
#undef QWN
  return xh_launch_status();
}
