// Quad-channel 3x3x3 stride-1 convolution for gfx950 (forward and data gradient), 16-bit storage: the kernel for convs
// whose groups have FEW channels (4 -> 4, 12 -> 4, 4 -> 12, the four-stream 16 -> 16 g4 of the encoder, ...), i.e. every
// 128^3-class conv of XLSTM_HVED (RA_HVED.py:510-648).  A plain implicit GEMM with N = output channels leaves 12 of the 16
// MFMA columns empty there and, worse, spends its time in per-voxel vector instructions around the matrix cores
// (measured on conv3_mfma_tile4_kernel, 4 -> 4 @128^3: 51 us, of which 33 us remain with loads, MFMAs and stores removed).
//
// W-Toeplitz GEMM on mfma_f32_16x16x32:   D[(c, p)][q] += A[(c, p)][(s, ci)] * B[(s, ci)][q]
//   q  = one of 16 output "quads" (4 consecutive voxels along W)            -> the N axis
//   (c, p) = output channel c of the quad of 4, position p inside the quad   -> the M axis (16 rows, all useful)
//   (s, ci) = input voxel 4q - 2 + s (s = 0..7), input channel ci of 4       -> the K axis (32, 24 of them non-zero)
//   A[(c, p)][(s, ci)] = W[c][ci][kd][kh][kw = s - p - 1] (0 outside 0..2)  -> one fragment per (kd, kh), in registers
// so one MFMA per (kd, kh) yields 64 voxels x 4 channels: 9 MFMAs per 64 voxels instead of 20, and
//  * the B operand of a lane is ONE aligned 16-byte LDS read (input planes are staged channels-last, 8 bytes per voxel);
//    a wave walks along D and reads every staged row once for its three depth taps (3 reads per 9 MFMAs);
//  * the accumulator layout is the output layout: lane (q, c) holds 4 consecutive voxels of channel c -> one 8-byte
//    NCDHW store per lane, all 64 lanes busy, 64-byte runs per row;
//  * staging applies the producer's InstanceNorm + LeakyReLU on the way in with packed fp32 arithmetic and writes 16-byte
//    LDS chunks; out-of-volume chunks get scale = shift = 0 (norm path) or a zero byte selector (copy path): no guarded
//    loads, every global load of a workgroup is in flight before the first one is used;
//  * the epilogue variant (none / output moments / norm-backward sums) and the input transform are template arguments.
// One workgroup (4 waves) = a TD x 8 x 32 (D x H x W) block of outputs of ONE output-channel quad, TD = 8 (4 / 2 on launches that
// would leave most CUs without a workgroup, see TD below); more input channels are walked quad by quad through the same LDS
// tile (28.8 KB at TD = 8; 4-5 workgroups per CU hide each other's staging phase).
//
// Measured on MI355X -- TWO regimes, do not mix them:
//  (a) microbenchmark with ONE operand set (tools/microbench_big.py --abl, XH_ROT=1): the 201 MB of a 16 -> 16 g4 @128^3 launch sit
//      in the 256 MB last-level cache.  Forward without statistics 39.5 us: staging 23.7, matrix phase + epilogue 13.2, launch 2.6;
//      statistics + fan-in add 10 us.  SQ counters there: issue slots 91 % busy (4.1 cycles per instruction).
//  (b) in the training step / with operands rotated through 1.6 GB (XH_ROT=8), i.e. from HBM: the same launch 49.2 us (data gradient
//      with the norm-backward epilogue 59.7 us = 3.4 TB/s = 0.42 of the 8 TB/s peak), staging alone 29.5 us.  SQ counters in the step
//      (profiles/r04c_pmc_sq.json): issue 17 %, SQ_WAIT_ANY 41 %, MFMA busy 12 %; memory-side counters: TA busy 59 % of
//      the launch and mostly stalled by the cache, L1 hit rate 60 % -- the tile's 64-byte half-lines and 4-byte edge loads are
//      what saturates, which is why neither more resident waves nor the persistent prefetching variant (conv3d_q4p.hip) helps
//      single-quad tiles.
// Tried and dropped: loads of the next input quad issued under the matrix phase of the current one
// (needs 154-168 VGPRs = 3 workgroups per CU: 12 -> 4 @128^3 33 -> 38 us, the 64^3 / 32^3 shapes unchanged -- those are
// latency chains of a few dozen workgroups); the epilogue of output plane pz - 2 run inside the plane walk (its loads requested
// two or four planes ahead, branch-free): 16 -> 16 g4 data gradient 56 -> 48 us in the microbenchmark, whose operands sit in the
// 256 MB last-level cache, but 58 -> 62 us inside the training step, where they come from HBM (step 5.16 -> 5.19 ms); four of
// the eight epilogue loads requested before the matrix phase: no change.
#include "common.h"
#include "conv_pack.h"
#include "fanin.h"
#include "../../include/xlstm_hved.h"

#include "conv_q4.h"
int g_q4_maxc = 48;               // xh_set_option(11, n): most channels per group the quad-channel kernel takes (<= 48)
int g_q4_wgs = 512;               // xh_set_option(17, n): workgroup count below which a launch takes 4, then 2 output planes per workgroup (0: always 8)

// two values of one channel -> leaky(x * sc + sh) in fp32 (packed fma / mul; leaky = max(v, slope * v) for 0 <= slope <= 1)
template <int FMT> __device__ __forceinline__ f32x2_t q4_xf(unsigned u, float sc, float sh, float slope) {
  const f32x2_t v = cvt2_in<FMT>(u) * f32x2_t{sc, sc} + f32x2_t{sh, sh};
  return max2(v, v * f32x2_t{slope, slope});
}

// PRE: 0 = the input as stored, 1 = leaky(x * sc + sh) (the producer's norm + activation), 2 = the InstanceNorm BACKWARD of
// the stage behind this data gradient applied on load: v = A g + C x + B with g = xa (the masked data gradient the next
// conv's backward left), x = px (that stage's saved raw input) and per-(n, c) coefficients derived in-kernel from the raw
// sums nb_red = (sum g, sum g x), mean, rstd -- what xh_in_bwd_apply computes in a pass of its own (reads g, x; writes dx)
// before this conv reads dx again.  The workgroups of the first output quad of a group also STORE v for the voxels their tile
// owns (pd): the weight gradient of this conv needs the tensor materialised.
// (body with the block coordinates as arguments: conv3_q4_kernel passes the launch's own, conv3_q4_pair_kernel those of the problem a
// workgroup belongs to)
template <int FMT, int PRE, int EPI, bool ACT, bool MULTI, int TD = 8>
__device__ __forceinline__ void q4_body(const ConvQ4& a, unsigned char* smem, const int bx, const int by, const int bz, const int gdx, const int gdy) {
  typedef h16<FMT> ST;
  constexpr int ID = TD + 2, TILE_BYTES = ID * PLANE, NROWS = ID * IH, NITEM = NROWS * 4, NEDGE = NROWS * 2;
  constexpr int NIT = (NITEM + 255) / 256;           // interior items (row, 8-voxel group) per thread; edge items: (row, side)
  double* s_red = reinterpret_cast<double*>(smem + TILE_BYTES);      // [4 waves][8], then [8] totals + the fan-in flag
  float* s_fin = reinterpret_cast<float*>(smem + TILE_BYTES + 48 * sizeof(double));   // [2][Q4_MAXC]: in-kernel InstanceNorm scale / shift
                                                                                      // (PRE == 2: [3][Q4_MAXC] = A, C, B)

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nn = lane & 15, g4 = lane >> 4;
  const int oq = by, n = bz;
  const int co0 = oq * 4;
  const int grp = udiv_fast(oq, a.oq_g, a.mQ);
  const int cin_base = grp * a.Cin_g;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  const int Do = a.d.Do, Ho = a.d.Ho;
  const int wk = xcd_swizzle(bx, gdx);
  const int wk1 = udiv_fast(wk, a.tilesW, a.mW), tw = wk - wk1 * a.tilesW;
  const int td = udiv_fast(wk1, a.tilesH, a.mH), th = wk1 - td * a.tilesH;
  const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
  if (a.abl & 4096) return;
  // raw InstanceNorm sums of this group's input channels (fused finalisation): requested first, used behind the staging loads
  double fs1 = 0.0, fs2 = 0.0;
  if (PRE == 1 && a.p.fin_red && tid < a.Cin_g) {
    fs1 = a.p.fin_red[2 * (n * a.d.Cin + cin_base + tid)];
    fs2 = a.p.fin_red[2 * (n * a.d.Cin + cin_base + tid) + 1];
  }

  // ---- staging plan (the same for every input-channel quad) ----
  unsigned i_off[NIT];          // byte offset of the item's 8 voxels inside a channel volume (clamped into the volume)
  int i_lds[NIT];               // LDS byte address of the item's first 16-byte chunk, before the per-chunk XOR
  int i_par[NIT];
  bool i_live[NIT], i_do[NIT];
  bool i_own[NIT];              // PRE == 2: this tile owns the item's voxels (not halo, inside the volume) -> side store
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = tid + it * 256;
    i_do[it] = item < NITEM;
    const int gq = item & 3, row = min(item >> 2, NROWS - 1);
    const int dz = row / IH, hy = row - dz * IH;
    const int gd = od0 - 1 + dz, gh = oh0 - 1 + hy;
    i_live[it] = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H;
    i_own[it] = i_live[it] && i_do[it] && dz >= 1 && dz <= TD && hy >= 1 && hy <= TH;
    const int gdc = min(max(gd, 0), D - 1), ghc = min(max(gh, 0), H - 1);
    i_off[it] = (unsigned)((((long long)gdc * H + ghc) * W + ow0 + gq * 8) * (long long)sizeof(ST));
    i_lds[it] = row * PITCH;
    i_par[it] = (row & 1) | ((1 + 4 * gq) << 1);           // bit 0: row parity, rest: first 16-byte slot of the item
  }
  unsigned e_off; int e_lds; bool e_live;
  const bool e_do = tid >= 256 - NEDGE;                // edge items go to the threads with the fewest interior items
  {
    const int ei = max(255 - tid, 0) < NEDGE ? 255 - tid : 0;
    const int row = ei >> 1, side = ei & 1;
    const int dz = row / IH, hy = row - dz * IH;
    const int gd = od0 - 1 + dz, gh = oh0 - 1 + hy;
    const int gw = side ? ow0 + TW : ow0 - 2;
    e_live = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw >= 0 && gw < W;
    const int gdc = min(max(gd, 0), D - 1), ghc = min(max(gh, 0), H - 1), gwc = min(max(gw, 0), W - 2);
    e_off = (unsigned)((((long long)gdc * H + ghc) * W + gwc) * (long long)sizeof(ST));
    e_lds = row * PITCH + (((side ? 17 : 0) ^ (row & 1)) << 4);
  }
  // ---- B (data) fragment addresses: lane (nn, g4) reads quad (r, qw) of a 2-row x 32-voxel unit; the quad order and the
  // row-parity XOR of the 16-byte slot make the ds_read_b128 bank-conflict free at a 288-byte pitch ----
  const int ur = (nn >> 2) & 1;
  const int qw = (nn >> 3) | ((nn & 3) << 1);
  const int rowbase = (2 * wv + ur) * PITCH;
  const int b_off0 = rowbase + ((2 * qw + g4) ^ ur) * 16;          // rows whose parity equals ur's (kh = 0, 2)
  const int b_off1 = rowbase + ((2 * qw + g4) ^ ur ^ 1) * 16;      // kh = 1
  // ---- epilogue lane role: lane (quad, channel) owns 4 consecutive voxels of one output row; everything that differs
  // between lanes is ONE 32-bit byte offset (a.d: dhw < 2^29), the rest of every address is scalar ----
  const int oh = oh0 + 2 * wv + ur;
  const bool row_ok = oh < Ho;
  const int ndz = min(TD, Do - od0);                  // output planes of this tile inside the volume
  float bias = 0.f, esc = 0.f, esh = 0.f;
  {
    const int wp = udiv_fast(grp, a.gpp, a.mG);
    const float* bp = a.p.b[wp];
    if (bp) bias = bp[(grp - wp * a.gpp) * a.Cout_g + (oq - grp * a.oq_g) * 4 + g4];
  }
  const long long odhw = (long long)Do * Ho * a.d.Wo;
  const unsigned spd_b = (unsigned)(Ho * a.d.Wo) * (unsigned)sizeof(ST);                       // bytes per output plane
  const unsigned lane_b = (unsigned)(((long long)g4 * odhw + (long long)(row_ok ? oh : 0) * a.d.Wo + ow0 + 4 * qw) * (long long)sizeof(ST));
  const unsigned lane_bo = row_ok ? lane_b : Q4_OOB;      // rows below the volume: the stores fall outside the window
  __amdgpu_buffer_rsrc_t ers = q4_window(a.p.y), yrs;
  if (EPI == 1) {
    esc = a.p.e_sc[n * a.d.Cout + co0 + g4];
    esh = a.p.e_sh[n * a.d.Cout + co0 + g4];
    // a quad never straddles the ea / eb boundary (Cea % 4 == 0)
    ers = q4_window(reinterpret_cast<const char*>(co0 < a.d.Cea ? (const ST*)a.p.ea + n * a.d.ea_bs + (long long)co0 * odhw
                                                                  : (const ST*)a.p.eb + n * a.d.eb_bs + (long long)(co0 - a.d.Cea) * odhw) +
                    (long long)od0 * spd_b);
  }
  yrs = q4_window(reinterpret_cast<char*>((ST*)a.p.y + n * a.d.y_bs + (long long)co0 * odhw) + (long long)od0 * spd_b);

  f32x4 acc[TD];
#pragma unroll
  for (int i = 0; i < TD; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ncq = MULTI ? a.ci4 : 1;
  uint2 eraw[TD];
  // statistics of this lane: packed fp32 partial sums over its 32 values (the planes of one row); fp64 from the workgroup
  // level on (16-bit storage: the 2^-24 of a 512-term fp32 partial is far below the storage rounding of every term)
  f32x2_t ps = {0.f, 0.f}, pq = {0.f, 0.f};
  const f32x2_t aslope2 = {a.act_slope, a.act_slope}, bias2 = {bias, bias}, esc2 = {esc, esc}, esh2 = {esh, esh};
  const float eslope = a.d.e_slope;
  // no branch around the loads and stores of the epilogue: planes past the volume are loaded from the last valid plane and
  // stored outside the window
  auto epi_load = [&](int dz) {
    if (EPI == 1)
      eraw[dz] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(ers, (int)lane_b, (int)((unsigned)min(dz, ndz - 1) * spd_b), 0));
  };
  auto epi_do = [&](int dz) {
    const bool live = dz < ndz;                       // uniform
    f32x2_t v[2] = {f32x2_t{acc[dz][0], acc[dz][1]} + bias2, f32x2_t{acc[dz][2], acc[dz][3]} + bias2};
    if (ACT) { v[0] = max2(v[0], v[0] * aslope2); v[1] = max2(v[1], v[1] * aslope2); }
    uint2 pk;
    if (EPI == 1) {
      const f32x2_t e[2] = {cvt2_in<FMT>(eraw[dz].x), cvt2_in<FMT>(eraw[dz].y)};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x2_t z = e[q] * esc2 + esh2, vs = v[q] * f32x2_t{eslope, eslope};
        v[q] = f32x2_t{z.x > 0.f ? v[q].x : vs.x, z.y > 0.f ? v[q].y : vs.y};
      }
      pk.x = cvt2_pack<FMT>(v[0].x, v[0].y); pk.y = cvt2_pack<FMT>(v[1].x, v[1].y);
      if (live) {
        const f32x2_t r0 = cvt2_in<FMT>(pk.x), r1 = cvt2_in<FMT>(pk.y);         // the values as stored
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
  };

  const float pslope = a.d.pre_slope;
  const bool fin = PRE == 1 && a.p.fin_red != nullptr;
  for (int cq = 0; cq < ncq; ++cq) {
    const int c0 = cin_base + cq * 4;
    const char* src = reinterpret_cast<const char*>(c0 < a.d.Ca ? (const ST*)a.p.xa + n * a.d.xa_bs + (long long)c0 * dhw
                                                                  : (const ST*)a.p.xb + n * a.d.xb_bs + (long long)(c0 - a.d.Ca) * dhw);
    const long long dhw_b = dhw * (long long)sizeof(ST);
    // ---- all global loads of this thread, back to back ----
    uint4 raw[NIT][4];
    unsigned eraw4[4];
    uint4 rawx[PRE == 2 ? NIT : 1][4];
    unsigned erawx4[4];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        raw[it][cc] = *reinterpret_cast<const uint4*>(src + cc * dhw_b + i_off[it]);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) eraw4[cc] = *reinterpret_cast<const unsigned*>(src + cc * dhw_b + e_off);
    if (PRE == 2) {
      const char* srcx = reinterpret_cast<const char*>((const ST*)a.p.px + n * a.d.px_bs + (long long)c0 * dhw);
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          rawx[it][cc] = *reinterpret_cast<const uint4*>(srcx + cc * dhw_b + i_off[it]);
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) erawx4[cc] = *reinterpret_cast<const unsigned*>(srcx + cc * dhw_b + e_off);
    }
    if (cq > 0) __syncthreads();                      // every wave is done reading the previous quad's tile
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    float cx[4] = {0.f, 0.f, 0.f, 0.f};               // PRE == 2: v = sc * g + cx * x + sh
    if (PRE == 2) {
      if (cq == 0) {
        // the coefficients of xh_in_bwd_apply from the raw sums of this group's channels (fp64, the same bits in every workgroup)
        if (tid < a.Cin_g) {
          const long long k = (long long)n * a.d.Cin + cin_base + tid;
          const double rs = a.p.nb_rstd[k], mu = a.p.nb_mean[k], cnt = (double)a.p.nb_count;
          const double S0 = a.p.nb_red[k * 2], P = rs * (a.p.nb_red[k * 2 + 1] - mu * S0);
          s_fin[tid] = (float)rs;
          s_fin[Q4_MAXC + tid] = (float)(-rs * rs * P / cnt);
          s_fin[2 * Q4_MAXC + tid] = (float)(-rs * S0 / cnt + rs * rs * mu * P / cnt);
        }
        __syncthreads();
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        sc[cc] = s_fin[cq * 4 + cc]; cx[cc] = s_fin[Q4_MAXC + cq * 4 + cc]; sh[cc] = s_fin[2 * Q4_MAXC + cq * 4 + cc];
      }
    }
    if (PRE == 1) {
      if (fin) {
        // fused InstanceNorm finalisation (xh_conv_ptrs.fin_red), behind the loads just issued: the raw sums of this
        // group's channels -> scale / shift in LDS (fp64, the same bits in every workgroup); workgroup (0, 0, 0) also
        // leaves them and mean / rstd of ALL channels in memory for the backward pass
        if (cq == 0) {
          // (fin_gamma: a training-mode BatchNorm of ONE sample -- instance statistics, then the affine; xh_conv_ptrs)
          const float* gam = a.p.fin_gamma;
          const float* bet = a.p.fin_beta;
          if (tid < a.Cin_g) {
            float m_, r_, sc_, sh_;
            in_finalize(fs1, fs2, a.fin_inv, sc_, sh_, m_, r_);
            if (gam) { const float g_ = gam[cin_base + tid]; sc_ *= g_; sh_ = fmaf(sh_, g_, bet[cin_base + tid]); }
            s_fin[tid] = sc_; s_fin[Q4_MAXC + tid] = sh_;
          }
          if (bx == 0 && by == 0 && bz == 0)
            for (int i = tid; i < a.d.N * a.d.Cin; i += 256) {
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
    const bool side = PRE == 2 && a.p.pd != nullptr && oq == grp * a.oq_g;      // one workgroup per (tile, input quad) stores
    char* dstd = PRE == 2 ? reinterpret_cast<char*>((ST*)a.p.pd + n * a.d.pd_bs + (long long)c0 * dhw) : nullptr;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (!i_do[it]) continue;
      const int par = i_par[it] & 1, slot0 = i_par[it] >> 1;
      uint4 outv[4];                                  // chunk j = voxels 2j, 2j+1 x 4 channels
      if (PRE) {
        const float lv = i_live[it] ? 1.f : 0.f;
        f32x2_t v[4][4];                              // [channel][voxel pair]
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float s1 = sc[cc] * lv, s2 = sh[cc] * lv;
          const unsigned u[4] = {raw[it][cc].x, raw[it][cc].y, raw[it][cc].z, raw[it][cc].w};
          if (PRE == 2) {
            const float s3 = cx[cc] * lv;
            const unsigned ux[4] = {rawx[it][cc].x, rawx[it][cc].y, rawx[it][cc].z, rawx[it][cc].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
              v[cc][k] = cvt2_in<FMT>(u[k]) * f32x2_t{s1, s1} + (cvt2_in<FMT>(ux[k]) * f32x2_t{s3, s3} + f32x2_t{s2, s2});
            if (side && i_own[it]) {                  // the materialised tensor (channel-planar, as stored by xh_in_bwd_apply)
              uint4 o;
              o.x = cvt2_pack<FMT>(v[cc][0].x, v[cc][0].y); o.y = cvt2_pack<FMT>(v[cc][1].x, v[cc][1].y);
              o.z = cvt2_pack<FMT>(v[cc][2].x, v[cc][2].y); o.w = cvt2_pack<FMT>(v[cc][3].x, v[cc][3].y);
              *reinterpret_cast<uint4*>(dstd + cc * dhw_b + i_off[it]) = o;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[cc][k] = q4_xf<FMT>(u[k], s1, s2, pslope);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          outv[j].x = cvt2_pack<FMT>(v[0][j].x, v[1][j].x);
          outv[j].y = cvt2_pack<FMT>(v[2][j].x, v[3][j].x);
          outv[j].z = cvt2_pack<FMT>(v[0][j].y, v[1][j].y);
          outv[j].w = cvt2_pack<FMT>(v[2][j].y, v[3][j].y);
        }
      } else {
        const unsigned se = i_live[it] ? 0x05040100u : 0x0c0c0c0cu, so = i_live[it] ? 0x07060302u : 0x0c0c0c0cu;
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
      uint4 o;
      if (PRE) {
        const float lv = e_live ? 1.f : 0.f;
        f32x2_t v[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          if (PRE == 2)
            v[cc] = cvt2_in<FMT>(eraw4[cc]) * f32x2_t{sc[cc] * lv, sc[cc] * lv} +
                    (cvt2_in<FMT>(erawx4[cc]) * f32x2_t{cx[cc] * lv, cx[cc] * lv} + f32x2_t{sh[cc] * lv, sh[cc] * lv});
          else v[cc] = q4_xf<FMT>(eraw4[cc], sc[cc] * lv, sh[cc] * lv, pslope);
        }
        o.x = cvt2_pack<FMT>(v[0].x, v[1].x);
        o.y = cvt2_pack<FMT>(v[2].x, v[3].x);
        o.z = cvt2_pack<FMT>(v[0].y, v[1].y);
        o.w = cvt2_pack<FMT>(v[2].y, v[3].y);
      } else {
        const unsigned se = e_live ? 0x05040100u : 0x0c0c0c0cu, so = e_live ? 0x07060302u : 0x0c0c0c0cu;
        o.x = __builtin_amdgcn_perm(eraw4[1], eraw4[0], se);
        o.y = __builtin_amdgcn_perm(eraw4[3], eraw4[2], se);
        o.z = __builtin_amdgcn_perm(eraw4[1], eraw4[0], so);
        o.w = __builtin_amdgcn_perm(eraw4[3], eraw4[2], so);
      }
      *reinterpret_cast<uint4*>(smem + e_lds) = o;
    }
    // A (weight) fragments of this (output quad, input quad): 9 x 16 bytes per lane, L2 resident; issued here so that
    // they travel while the workgroup gathers at the barrier (the staging registers are dead by now)
    frag8 wfrag[9];
    {
      const frag8* wpk = reinterpret_cast<const frag8*>(a.p.ws) + ((long long)oq * a.ci4 + cq) * 9 * 64 + lane;
#pragma unroll
      for (int i = 0; i < 9; ++i) wfrag[i] = wpk[i * 64];
    }
    __syncthreads();
    if (a.abl & 8192) continue;
    // ---- matrix phase: the wave's two output rows, walking the 10 staged planes once ----
    {
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
    }
  }
  if (a.abl & 8192) return;

  {
#pragma unroll
    for (int dz = 0; dz < TD; ++dz) epi_load(dz);
#pragma unroll
    for (int dz = 0; dz < TD; ++dz) epi_do(dz);
  }
  if (EPI) {
    // lanes of one channel: the 16 lanes nn = 0..15 of a lane group g4 = one DPP row
    const float t0 = row16_sum(row_ok ? ps.x + ps.y : 0.f), t1 = row16_sum(row_ok ? pq.x + pq.y : 0.f);
    if (nn == 0) { s_red[wv * 8 + g4 * 2] = (double)t0; s_red[wv * 8 + g4 * 2 + 1] = (double)t1; }
    __syncthreads();
    // the 8 channel sums of the workgroup, then the two-level fan-in of fanin.h (one fp64 atomic per workgroup and value on
    // red[] is 1024 .. 4096 requests on ONE cache line, retired one after the other: 8 us of a 22 us launch)
    if (tid < 8) {
      const double tot = s_red[tid] + s_red[8 + tid] + s_red[16 + tid] + s_red[24 + tid];
      s_red[32 + tid] = tot;
    }
    double* s_tot = s_red + 32;
    if (a.fan && !fan_in<8>(a.fan + ((long long)n * gdy + oq) * FAN_UNIT_BYTES, bx, gdx, s_tot,
                            reinterpret_cast<int*>(s_tot + 8)))
      return;
    if (!a.fan) __syncthreads();
    if (tid < 8) atomicAdd(&a.p.red[((long long)n * a.d.Cout + co0 + (tid >> 1)) * 2 + (tid & 1)], s_tot[tid]);
  }
}

template <int FMT, int PRE, int EPI, bool ACT, bool MULTI, int TD = 8>
__global__ __launch_bounds__(256, PRE == 2 ? 3 : 4) void conv3_q4_kernel(const ConvQ4 a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  q4_body<FMT, PRE, EPI, ACT, MULTI, TD>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}
// two independent convolutions of one shape in one launch (xh_conv3d_fwd_pair; rows of 32 voxels: the 32^3 decoder level)
template <int FMT, int PRE, int EPI, bool MULTI, int TD>
__global__ __launch_bounds__(256, 4) void conv3_q4_pair_kernel(const ConvQ4 a0, const ConvQ4 a1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = a0.d.N;
  if ((int)blockIdx.z >= N) q4_body<FMT, PRE, EPI, false, MULTI, TD>(a1, smem, blockIdx.x, blockIdx.y, (int)blockIdx.z - N, gridDim.x, gridDim.y);
  else q4_body<FMT, PRE, EPI, false, MULTI, TD>(a0, smem, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

static bool q4_plan(const xh_conv_desc* d, ConvQ4* a) {
  extern int g_xh_disable;
  if (g_xh_disable & 16) return false;
  const bool f32 = d->dtype == XH_F32;                 // fp32 storage: conv3d_q4s.hip (two-term fp16 operands), opt-in
  if (f32 && !(d->arith & XH_ARITH_F32_SPLIT)) return false;
  if ((d->dtype != XH_BF16 && d->dtype != XH_F16 && !f32) || d->k != 3 || d->stride != 1) return false;
  if (f32 && (d->pre == 2 || d->act != XH_ACT_NONE)) return false;
  if (d->groups <= 0 || d->Cin % d->groups || d->Cout % d->groups) return false;
  if (d->W % TW != 0 || d->Wo != d->W || d->Ho != d->H || d->Do != d->D) return false;
  int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  // a depthwise conv (one channel per group) rides as groups of 4 channels whose 4 x 4 weight blocks are diagonal: three
  // quarters of the MACs multiply zeros, but they are matrix-core MACs -- the vector-ALU sliding-window kernel needs 27 FMAs
  // per voxel and channel and ran 4 -> 4 @128^3 in 31 us; this kernel takes 17
  const bool dw = cin_g == 1 && cout_g == 1 && d->groups % 4 == 0 && d->n_wptr > 0 && (d->groups / 4) % d->n_wptr == 0 &&
                  !(g_xh_disable & 128);
  if (dw) cin_g = cout_g = 4;
  if (cin_g % 4 || cout_g % 4) return false;
  { extern int g_q4_maxc; if (cin_g > g_q4_maxc || cout_g > g_q4_maxc) return false; }   // denser groups: the plain implicit GEMM
  if (d->Ca % 4) return false;
  if (d->epi == 1 && d->Cea % 4) return false;
  if (d->bcast) {                                      // broadcast operand (xh_conv_desc.bcast): 16-bit storage, 4 -> 4 per group, one source
    if (d->bcast != 4 || f32 || dw || d->pre == 2 || d->Ca != d->Cin || cin_g != 4 || cout_g != 4) return false;
    if (d->transposed && (d->epi != 1 || d->Cea != d->Cout)) return false;
  }
  if ((d->xa_bs & 7) || (d->xb_bs & 7) || (d->y_bs & 7) || (d->ea_bs & 7) || (d->eb_bs & 7)) return false;
  const long long dhw = (long long)d->D * d->H * d->W;
  if (dhw % 8 || dhw >= (f32 ? (1ll << 27) : (1ll << 28))) return false;      // 31-bit byte offsets inside a quad of channel volumes (q4_window)
  if (d->pre == 1 && !(d->pre_slope >= 0.f && d->pre_slope <= 1.f)) return false;
  if (d->pre == 2 && (d->Ca != d->Cin || (d->px_bs & 7) || (d->pd_bs & 7))) return false;     // one source, 16-byte runs
  float as = 1.f;
  if (d->act == XH_ACT_RELU) as = 0.f;
  else if (d->act == XH_ACT_LRELU) as = d->act_slope;
  else if (d->act != XH_ACT_NONE) return false;
  if (!(as >= 0.f && as <= 1.f)) return false;
  if (d->N > 65535 || d->Cout / 4 > 65535) return false;
  if (d->D < 4 || d->H < 8) return false;
  a->d = *d;
  a->dw = dw ? 1 : 0;
  if (dw) a->d.groups = d->groups / 4;
  a->Cin_g = cin_g; a->Cout_g = cout_g; a->ci4 = cin_g / 4;
  a->tilesW = d->W / TW; a->tilesH = cdiv(d->Ho, TH);
  {
    // fewer planes per workgroup while the launch would leave most of the 256 CUs without one (the 64^3 / 32^3 levels)
    extern int g_q4_wgs;
    const long long cols = (long long)a->tilesW * a->tilesH * (d->Cout / 4) * d->N;
    int td = 8;
    while (td > 2 && cols * cdiv(d->Do, td) < g_q4_wgs) td >>= 1;
    if (as != 1.f) td = 8;                              // the conv + activation instances exist for 8 planes only
    if (f32 && td > 4) td = 4;                          // two LDS images per tile: 4 planes keep three workgroups per CU
    a->td = td;
  }
  a->tilesD = cdiv(d->Do, a->td);
  a->oq_g = cout_g / 4; a->gpp = a->d.groups / d->n_wptr;
  // udiv_fast is exact while index * divisor < 2^32
  if ((long long)a->tilesW * a->tilesH * a->tilesD * (a->tilesW > a->tilesH ? a->tilesW : a->tilesH) >= (1ll << 32)) return false;
  a->mW = udiv_magic(a->tilesW); a->mH = udiv_magic(a->tilesH); a->mQ = udiv_magic(a->oq_g); a->mG = udiv_magic(a->gpp);
  a->act_slope = as;
  a->abl = g_mfma_abl;
  return true;
}

static void q4_pack_job(const ConvQ4& a, PackJob* j) {
  for (int i = 0; i < XH_MAX_WPTR; ++i) j->w[i] = a.p.w[i];
  j->ws = a.p.ws;
  j->kind = 1;
  j->f16 = a.d.dtype == XH_F32 ? 2 : a.d.dtype == XH_F16;      // 2: hi / lo fp16 images (conv3d_q4s.hip)
  j->groups = a.d.groups; j->n_wptr = a.d.n_wptr; j->transposed = a.d.transposed;
  j->Cin_g = a.Cin_g; j->Cout_g = a.Cout_g;
  j->ntile = j->cin_stride = j->cin_off = j->cin_blk = j->cout_set = j->nm = j->nch = j->cpr = j->cinp = 0;
  j->ci4 = a.ci4;
  j->dw = a.dw;
  j->nelem = (a.d.Cout / 4) * a.ci4 * 9 * 512;
}
bool xh_conv3_q4_pack_job(const xh_conv_desc* d, const xh_conv_ptrs* p, PackJob* j) {
  ConvQ4 a;
  if (!q4_plan(d, &a)) return false;
  a.p = *p;
  q4_pack_job(a, j);
  return true;
}
void xh_launch_pack_single(hipStream_t st, const PackJob& j);                   // conv3d_mfma.hip

long long xh_conv3_q4_workspace_bytes(const xh_conv_desc* d) {
  ConvQ4 a;
  if (!q4_plan(d, &a)) return 0;
  return (long long)(d->Cout / 4) * a.ci4 * 9 * 1024 * (d->dtype == XH_F32 ? 2 : 1);
}
int xh_conv3_q4s_launch(hipStream_t st, const ConvQ4& a, dim3 grid);           // conv3d_q4s.hip
int xh_conv3_q4p_try(hipStream_t st, const ConvQ4& a);                         // conv3d_q4p.hip: persistent, pipelined over tiles
int xh_conv3_q4w_try(hipStream_t st, ConvQ4& a);                               // conv3d_q4w.hip: full-row tiles for 128-wide rows

int xh_conv3_q4w_pair_try(hipStream_t st, ConvQ4& a, ConvQ4& b);              // conv3d_q4w.hip
static inline bool xh_conv3_q4p_pair_blocked(const ConvQ4&) { return false; }  // (the persistent variant is a single-launch plan; pairs take the tile kernel)
// Two convolutions in one launch: XH_OK if launched, 1 if they are not a pair the full-row kernel takes
int xh_conv3_q4_pair_try(void* stream, const xh_conv_desc* d0, const xh_conv_ptrs* p0, const xh_conv_desc* d1, const xh_conv_ptrs* p1) {
  ConvQ4 a, b;
  if (!q4_plan(d0, &a) || !q4_plan(d1, &b)) return 1;
  if (!p0->ws || p0->ws_bytes < xh_conv3_q4_workspace_bytes(d0) || !p1->ws || p1->ws_bytes < xh_conv3_q4_workspace_bytes(d1)) return 1;
  if (!p0->ws_packed || !p1->ws_packed) return 1;      // (packed up front: xh_conv3d_prepack)
  a.p = *p0; b.p = *p1;
  a.fin_inv = p0->fin_count > 0 ? 1.0 / (double)p0->fin_count : 0.0;
  b.fin_inv = p1->fin_count > 0 ? 1.0 / (double)p1->fin_count : 0.0;
  const int r = xh_conv3_q4w_pair_try((hipStream_t)stream, a, b);
  if (r != 1) return r;
  // rows of 32 voxels: the tile kernel, forward with output moments (the 32^3 level's first decoder convs)
  if (d0->dtype == XH_F32 || d0->dtype != d1->dtype || d0->W != d1->W || d0->H != d1->H || d0->D != d1->D || d0->N != d1->N ||
      d0->Cout != d1->Cout || d0->pre != 1 || d1->pre != 1 || d0->epi != 2 || d1->epi != 2 || a.act_slope != 1.f || b.act_slope != 1.f ||
      (a.ci4 > 1) != (b.ci4 > 1) || a.td != b.td || a.tilesW != b.tilesW || a.tilesH != b.tilesH || a.tilesD != b.tilesD || d0->bcast ||
      d1->bcast || 2 * d0->N > 65535)
    return 1;
  if (a.ci4 > 1 && xh_conv3_q4p_pair_blocked(a)) return 1;
  dim3 grid(a.tilesW * a.tilesH * a.tilesD, d0->Cout / 4, 2 * d0->N);
  a.fan = b.fan = nullptr;
  const size_t shm = q4_tile_bytes(a.td) + 48 * sizeof(double) + 3 * Q4_MAXC * sizeof(float);
  const int f = d0->dtype == XH_F16 ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  xh_note_kernel("conv3_q4_pair_kernel<%d, 1, 2, %s, %d>", f, a.ci4 > 1 ? "true" : "false", a.td);
#define Q4P(F, M, T) hipLaunchKernelGGL((conv3_q4_pair_kernel<F, 1, 2, M, T>), grid, dim3(256), shm, st, a, b)
#define Q4PT(F, M)                 \
  do {                             \
    if (a.td == 8) Q4P(F, M, 8);   \
    else if (a.td == 4) Q4P(F, M, 4); \
    else Q4P(F, M, 2);             \
  } while (0)
  if (f) { if (a.ci4 > 1) Q4PT(1, true); else Q4PT(1, false); }
  else { if (a.ci4 > 1) Q4PT(0, true); else Q4PT(0, false); }
#undef Q4PT
#undef Q4P
  return xh_launch_status();
}

// XH_OK if launched, 1 if the shape is not eligible
int xh_conv3_q4_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p) {
  ConvQ4 a;
  if (!q4_plan(d, &a)) return 1;
  const long long need = xh_conv3_q4_workspace_bytes(d);
  if (!p->ws || p->ws_bytes < need) return 1;
  a.p = *p;
  a.fin_inv = p->fin_count > 0 ? 1.0 / (double)p->fin_count : 0.0;

  hipStream_t st = (hipStream_t)stream;
  const int f = d->dtype == XH_F16 ? 1 : 0;
  if (!p->ws_packed) {
    PackJob pj;
    q4_pack_job(a, &pj);
    xh_launch_pack_single(st, pj);
  }
  if (d->dtype != XH_F32) {
    int r = xh_conv3_q4w_try(st, a);                   // rows of 128 voxels: full-row tiles
    if (r != 1) return r;
    if (d->bcast) return XH_ERR_ARG;                   // (only the full-row kernel reads a broadcast operand)
    r = xh_conv3_q4p_try(st, a);                       // multi-quad forward launches with several tiles per workgroup slot: the persistent kernel
    if (r != 1) return r;
  }
  dim3 grid(a.tilesW * a.tilesH * a.tilesD, d->Cout / 4, d->N);
  a.fan = d->epi ? xh_fan_block(p->fan, p->fan_bytes, (long long)grid.y * grid.z, grid.x) : nullptr;
  if (d->bcast) return XH_ERR_ARG;
  if (d->dtype == XH_F32) return xh_conv3_q4s_launch(st, a, grid);
  const size_t shm = q4_tile_bytes(a.td) + 48 * sizeof(double) + 3 * Q4_MAXC * sizeof(float);
  if (d->pre == 2 && (!p->px || !p->nb_red || !p->nb_mean || !p->nb_rstd || p->nb_count <= 0 || d->epi == 2)) return XH_ERR_ARG;
  const bool act = a.act_slope != 1.f;
  xh_note_kernel("conv3_q4_kernel<%d, %d, %d, %s, %s, %d>", f, d->pre, d->epi, act ? "true" : "false", a.ci4 > 1 ? "true" : "false", a.td);
#define Q4L(F, P, E, A, T)                                                                                      \
  do {                                                                                                          \
    if (a.ci4 > 1) hipLaunchKernelGGL((conv3_q4_kernel<F, P, E, A, true, T>), grid, dim3(256), shm, st, a);     \
    else hipLaunchKernelGGL((conv3_q4_kernel<F, P, E, A, false, T>), grid, dim3(256), shm, st, a);              \
  } while (0)
#define Q4A(F, P, E)                               \
  do {                                             \
    if (act) Q4L(F, P, E, true, 8);                \
    else if (a.td == 8) Q4L(F, P, E, false, 8);    \
    else if (a.td == 4) Q4L(F, P, E, false, 4);    \
    else Q4L(F, P, E, false, 2);                   \
  } while (0)
#define Q4E(F, P)                       \
  do {                                  \
    if (d->epi == 0) Q4A(F, P, 0);      \
    else if (d->epi == 1) Q4A(F, P, 1); \
    else Q4A(F, P, 2);                  \
  } while (0)
  // (pre == 2 exists for the data gradients that carry the norm-backward epilogue or none: epi 0 / 1, no activation)
#define Q4N(F)                                                        \
  do {                                                                \
    if (act) return XH_ERR_ARG;                                       \
    if (d->epi == 1) {                                                \
      if (a.td == 8) Q4L(F, 2, 1, false, 8);                          \
      else if (a.td == 4) Q4L(F, 2, 1, false, 4);                     \
      else Q4L(F, 2, 1, false, 2);                                    \
    } else {                                                          \
      if (a.td == 8) Q4L(F, 2, 0, false, 8);                          \
      else if (a.td == 4) Q4L(F, 2, 0, false, 4);                     \
      else Q4L(F, 2, 0, false, 2);                                    \
    }                                                                 \
  } while (0)
  if (d->pre == 2) { if (f) Q4N(1); else Q4N(0); }
  else if (f) { if (d->pre) Q4E(1, 1); else Q4E(1, 0); }
  else { if (d->pre) Q4E(0, 1); else Q4E(0, 0); }
#undef Q4N
#undef Q4E
#undef Q4A
#undef Q4L
  return xh_launch_status();
}
