// bf16-MFMA 7x7x7 stride-1 convolution for the AttenModule2 gate convs (buildingblocks.py:271-296 after composition:
// 4 pooled channels -> 2 gates forward, 2 -> 4 data gradient), bf16 storage, fp32 accumulation.
//
// With only 2 (4) output channels a plain implicit GEMM wastes 14 (12) of the 16 MFMA columns.  Here the N dimension
// carries OUTPUT ROWS as well: for one input row rr of the halo tile and one depth tap kd
//
//     D[16 voxels along W][(j, co)] += A[16 voxels][(kw, ci)] * B[(kw, ci)][(j, co)],   B = W[co][ci][kd][kh = rr - j][kw]
//
// i.e. the weight matrix is Toeplitz in H: input row rr feeds output row j through tap kh = rr - j (zero outside 0..6).
// N = JR rows x CO channels = 16 exactly (JR = 8 for CO = 2, 4 for CO = 4); K = 8 kw slots (7 used) x CI channels, and
// for CI = 2 two depth taps share one MFMA so K is 32 either way.  MFMAs per output voxel drop 4x against channel
// padding, and the accumulator layout (lane = 4 consecutive voxels of one (row, channel)) stores straight to NCDHW.
//
//  * input planes: channels-last in LDS ([row][w][CI] bf16), ring of 8 slots, sliding along D like the 3x3x3 kernel
//    (one new plane per output plane, loaded to registers behind the MFMAs, written to the slot nobody reads);
//  * B fragments: a 8 KB LDS table [kd][kh (+1 zero row)][k-group][co] built by the workgroup from the fp32 weights;
//    every lane indexes it with its own kh = rr - j, one 16-byte read per (kd, rr), shared by the wave's two M tiles;
//  * a workgroup is 2 waves (one JR-row tile each) on a 32-wide column: 62 KB of LDS, two workgroups per CU.
#include "common.h"
#include "../../include/xlstm_hved.h"

typedef h16x8 bf16x8;     // 8 raw 16-bit values (either format)
typedef f32x4_t f32x4;

struct Conv7K {
  xh_conv_desc d;
  xh_conv_ptrs p;
  int tilesW, tilesH, sd, dsegs;
};

// KSPLIT: the depth taps of an output plane are shared by KSPLIT wave pairs (2 KSPLIT waves per workgroup on the same LDS tiles): a
// fraction of the MFMA chain per wave and KSPLIT waves per SIMD instead of one -- the kernel is parked on LDS fragment reads (6 % MFMA busy
// with one wave per SIMD), and at 32^3 / 64^3 its run time is the chain of one workgroup.  The partial tiles meet in LDS.
template <int FMT, int CI, int CO, int KSPLIT = 4>
__global__ __launch_bounds__(128 * KSPLIT, 2) void conv7_mfma_kernel(const Conv7K a) {
  typedef h16<FMT> ST;                                // storage type: ST or f16_t
  constexpr int PPM = 32 / (8 * CI);                  // depth taps per MFMA (1 for CI = 4, 2 for CI = 2)
  constexpr int NKD = (7 + PPM - 1) / PPM;            // MFMA steps along kd
  constexpr int JR = 16 / CO;                         // output rows per N tile
  constexpr int RT = 2;                               // row tiles per workgroup (one per wave)
  constexpr int TW = 32, MT = 2;                      // tile width, M tiles per wave
  constexpr int IH = RT * JR + 6, IWP = TW + 8;       // halo tile rows / columns (38 used + 2 for the kw = 7 over-read)
  constexpr int VB = CI * 2;                          // bytes per voxel in LDS
  constexpr int PLANE = IH * IWP * VB;
  constexpr int NG = TW / 8 + 2;                      // aligned 8-voxel groups covering [ow0 - 8, ow0 + TW + 8)
  constexpr int NTHR = 128 * KSPLIT;
  constexpr int NITEM = IH * NG, NIT = (NITEM + NTHR - 1) / NTHR;
  constexpr int GS = 4 / PPM;                         // k-groups per depth tap (4 or 2)
  constexpr int TBL = 8 * 8 * GS * CO * 16;           // B table bytes: [kd 8][kh 8][gslot][co] x 16 B
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_in = smem;                         // 8 * PLANE
  unsigned char* s_tb = smem + 8 * PLANE;             // TBL
  float* s_part = reinterpret_cast<float*>(smem + 8 * PLANE + TBL);   // [KSPLIT - 1][RT waves][MT][4][64]: partial tiles of the other tap shares

  const int tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & (RT - 1), kq = tid >> 7;
  const int g4 = lane >> 4, nn = lane & 15;
  const int n = blockIdx.z;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  int wk = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tw = wk % a.tilesW; wk /= a.tilesW;
  const int th = wk % a.tilesH;
  const int ds = wk / a.tilesH;
  const int oh0 = th * (RT * JR), ow0 = tw * TW;
  const int d_begin = ds * a.sd, d_end = min(D, d_begin + a.sd);

  // ---- B table: entry (kd, kh, gslot, co) = 8 bf16: element e -> (kw, ci) of the k-group ----
  {
    const float* wp = a.p.w[0];
    for (int idx = tid; idx < 8 * 8 * GS * CO * 8; idx += NTHR) {
      const int e = idx & 7;
      int r = idx >> 3;
      const int co = r % CO; r /= CO;
      const int gs = r % GS; r /= GS;
      const int kh = r & 7, kd = r >> 3;
      const int kw = gs * (8 / CI) + e / CI, ci = e % CI;
      float v = 0.f;
      if (kd < 7 && kh < 7 && kw < 7) {
        const int tap = (kd * 7 + kh) * 7 + kw;
        v = a.d.transposed ? wp[((long long)ci * CO + co) * 343 + (342 - tap)] : wp[((long long)co * CI + ci) * 343 + tap];
      }
      reinterpret_cast<unsigned short*>(s_tb)[idx] = cvt_out<FMT>(v);
    }
  }
  // lane roles: N column nn = (j, co); this lane's B row offset for (kd step, rr) and A offsets
  const int jn = nn / CO, con = nn % CO;
  const int kd_l = PPM == 1 ? 0 : (g4 >> 1);          // which of the step's depth taps this lane's k-group belongs to
  const int gs_l = PPM == 1 ? g4 : (g4 & 1);
  const int a_col = PPM == 1 ? 2 * g4 : 4 * (g4 & 1); // first voxel (kw) of this lane's k-group
  float bias = 0.f;
  if (a.p.b[0]) bias = a.p.b[0][con];

  // ---- staging plan ----
  const ST* sp_src[NIT][CI];
  int sp_lds[NIT], sp_gq[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = tid + it * NTHR;
    const int gi = item % NG, hy = item / NG;
    const int gq = gi - 1;
    const int gh = oh0 - 3 + hy, gw = ow0 + gq * 8;
    const bool inb = item < NITEM && (unsigned)gh < (unsigned)H && gw >= 0 && gw < W;
    sp_gq[it] = item < NITEM ? gq : 100;
    sp_lds[it] = hy * IWP * VB;
#pragma unroll
    for (int c = 0; c < CI; ++c)
      sp_src[it][c] = inb ? (const ST*)a.p.xa + n * a.d.xa_bs + (long long)c * dhw + (long long)gh * W + gw : nullptr;
  }
  uint4 raw[NIT][CI];
  auto load_plane = [&](int gd) {
    const bool dok = (unsigned)gd < (unsigned)D;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        raw[it][c] = make_uint4(0, 0, 0, 0);
        if (dok && sp_src[it][c]) raw[it][c] = *reinterpret_cast<const uint4*>(sp_src[it][c] + (long long)gd * hw);
      }
  };
  auto store_plane = [&](int gd) {
    const int slot = ((gd + 8) & 7) * PLANE;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      unsigned short v[CI][8];
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        const unsigned u[4] = {raw[it][c].x, raw[it][c].y, raw[it][c].z, raw[it][c].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[c][2 * k] = (unsigned short)(u[k] & 0xffffu); v[c][2 * k + 1] = (unsigned short)(u[k] >> 16); }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int col = sp_gq[it] * 8 + k + 3;        // tile column of voxel gw + k (column 0 = ow0 - 3)
        if (col >= 0 && col < IWP) {
          unsigned char* dst = s_in + slot + sp_lds[it] + col * VB;
          if (CI == 4) {
            uint2 pk;
            pk.x = (unsigned)v[0][k] | ((unsigned)v[1][k] << 16);
            pk.y = (unsigned)v[2][k] | ((unsigned)v[3][k] << 16);
            *reinterpret_cast<uint2*>(dst) = pk;
          } else {
            *reinterpret_cast<unsigned*>(dst) = (unsigned)v[0][k] | ((unsigned)v[1][k] << 16);
          }
        }
      }
    }
  };

  // ---- prologue: input planes d_begin-3 .. d_begin+3 ----
  for (int q = d_begin - 3; q <= d_begin + 3; ++q) {
    load_plane(q);
    store_plane(q);
  }
  __syncthreads();

  ST* ybase = (ST*)a.p.y + n * a.d.y_bs + (long long)con * dhw;
  for (int d = d_begin; d < d_end; ++d) {
    const bool more = d + 1 < d_end;
    if (more) load_plane(d + 4);                      // lands behind the MFMAs
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks = kq; ks < NKD; ks += KSPLIT) {
      const int kd = ks * PPM + kd_l;                 // this lane's depth tap; kd = 7 is the dummy half of the last step:
      const int kda = kd < 7 ? kd : 6;                // zero weights, and a resident (finite) plane to multiply them with
      const unsigned char* pl = s_in + ((d + kda - 3 + 8) & 7) * PLANE;
      const unsigned char* tb = s_tb + ((kd * 8) * GS + gs_l) * CO * 16 + con * 16;
#pragma unroll
      for (int rr = 0; rr < JR + 6; ++rr) {
        const int kh = rr - jn;
        const int khx = (unsigned)kh < 7u ? kh : 7;   // row 7 of the table is zero
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(tb + khx * GS * CO * 16);
        const unsigned char* ar = pl + ((wv * JR + rr) * IWP + nn + a_col) * VB;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned char* ap = ar + mt * 16 * VB;
          bf16x8 av;
          if (CI == 4) {
            const uint2 lo = *reinterpret_cast<const uint2*>(ap);
            const uint2 hi = *reinterpret_cast<const uint2*>(ap + 8);
            av = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          } else {
            const unsigned* a32 = reinterpret_cast<const unsigned*>(ap);
            av = __builtin_bit_cast(bf16x8, make_uint4(a32[0], a32[1], a32[2], a32[3]));
          }
          acc[mt] = mfma16x16x32<FMT>(av, bv, acc[mt]);
        }
      }
    }
    if (more) store_plane(d + 4);
    if (KSPLIT > 1) {                                  // the other tap shares hand their partial tiles over
      if (kq > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s_part[((((kq - 1) * RT + wv) * MT + mt) * 4 + r) * 64 + lane] = acc[mt][r];
      }
      __syncthreads();
      if (kq == 0) {
#pragma unroll
        for (int q = 0; q < KSPLIT - 1; ++q)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] += s_part[(((q * RT + wv) * MT + mt) * 4 + r) * 64 + lane];
      }
    }
    // ---- epilogue: lane = voxels 4*g4 .. +3 of M tile mt, output row jn of this wave's row tile, channel con ----
    const int oh = oh0 + wv * JR + jn;
    if (oh < H && kq == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = apply_act(acc[mt][r] + bias, a.d.act, a.d.act_slope);
        st4(ybase, ((long long)d * H + oh) * W + ow0 + mt * 16 + 4 * g4, o);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Input-stationary variant (round 5).  conv7_mfma_kernel above is OUTPUT-plane stationary: for every output plane it re-reads the 7
// input planes of its depth taps, one A fragment (1 KB per wave) per MFMA plus a B fragment per two -- 1.5 KB of LDS reads per
// MFMA, i.e. the LDS pipe (128 B / clk / CU) caps the four SIMDs at a third of their matrix rate, and the counters agreed (76 % of
// wave cycles waiting on LDS, MFMA busy 10.7 %, unchanged for two rounds).  Here
//   * the INPUT plane is stationary: an A fragment of base plane q is read ONCE and multiplied into the accumulators of all the
//     output planes it reaches, d = q + 3 - ks PPM (7 for CI = 4; 4 for CI = 2, where an MFMA pairs two depth taps): seven
//     output planes are in flight per wave, their accumulators rotate through a loop unrolled 7 times (static register indices);
//   * the B (weight) fragments do not depend on the plane or the column at all: a wave keeps those of ITS input rows in registers
//     for the whole run (the workgroup's KSPLIT wave pairs split the tile's JR + 6 input rows, not the depth taps);
//   * so a step is 1 LDS fragment read per 7 (4) MFMAs, and the ring needs the two planes being read plus two incoming: 4 slots.
// Partial tiles (other wave pairs' rows) meet in LDS when an output plane completes, as before; one barrier per plane.
// ~220 registers: one workgroup of 8 waves per CU.
// FMT 2 = fp32 STORAGE (the fp32_mfma mode, xh_set_option(18, 1)): x is read and y written as fp32, the operands are rounded ONCE to fp16
// on their way into LDS / the weight table, fp32 accumulation -- like the weight gradients of that mode (conv7_wgrad_mfma_multi_kernel<2>).
template <int FMT> struct C7Store { typedef h16<FMT> T; };
template <> struct C7Store<2> { typedef float T; };
template <int FMT, int CI, int CO, int KSPLIT = 4, bool BREG = false>
__global__ __launch_bounds__(128 * KSPLIT, BREG ? 2 : 4) void conv7_as_kernel(const Conv7K a) {
  typedef typename C7Store<FMT>::T ST;
  constexpr int CF = FMT == 2 ? 1 : FMT;               // format of the LDS images / MFMA operands
  constexpr bool F32S = FMT == 2;
  constexpr int PPM = 32 / (8 * CI);
  constexpr int NKD = (7 + PPM - 1) / PPM;
  constexpr int JR = 16 / CO;
  constexpr int RT = 2, TW = 32, MT = 2;
  constexpr int IH = RT * JR + 6, IWP = TW + 16;      // tile columns = the aligned 8-voxel groups [ow0 - 8, ow0 + TW + 8): column 0 = ow0 - 8
  constexpr int VB = CI * 2;
  constexpr int PLANE = IH * IWP * VB;
  constexpr int NG = TW / 8 + 2;
  constexpr int NTHR = 128 * KSPLIT;
  constexpr int NITEM = IH * NG, NIT = (NITEM + NTHR - 1) / NTHR;
  constexpr int GS = 4 / PPM;
  constexpr int TBL = 8 * 8 * GS * CO * 16;
  constexpr int NR = (JR + 6 + KSPLIT - 1) / KSPLIT;  // input rows per wave pair, at most
  constexpr int NS = 7;                               // output planes in flight
  constexpr int PART = (KSPLIT - 1) * RT * MT * 4 * 64;   // floats per partial-tile buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_in = smem;                         // 4 * PLANE
  unsigned char* s_tb = smem + 4 * PLANE;             // TBL
  float* s_part = reinterpret_cast<float*>(smem + 4 * PLANE + TBL);   // 2 buffers

  const int tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & (RT - 1), kq = tid >> 7;
  const int g4 = lane >> 4, nn = lane & 15;
  const int n = blockIdx.z;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  int wk = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tw = wk % a.tilesW; wk /= a.tilesW;
  const int th = wk % a.tilesH;
  const int ds = wk / a.tilesH;
  const int oh0 = th * (RT * JR), ow0 = tw * TW;
  const int d_begin = ds * a.sd, d_end = min(D, d_begin + a.sd);

  // ---- B table (as in conv7_mfma_kernel) and a zeroed plane ring: a stale slot is read (against zero weights, or into
  // accumulators that are never stored) and must hold finite values ----
  {
    const float* wp = a.p.w[0];
    for (int idx = tid; idx < 8 * 8 * GS * CO * 8; idx += NTHR) {
      const int e = idx & 7;
      int r = idx >> 3;
      const int co = r % CO; r /= CO;
      const int gs = r % GS; r /= GS;
      const int kh = r & 7, kd = r >> 3;
      const int kw = gs * (8 / CI) + e / CI, ci = e % CI;
      float v = 0.f;
      if (kd < 7 && kh < 7 && kw < 7) {
        const int tap = (kd * 7 + kh) * 7 + kw;
        v = a.d.transposed ? wp[((long long)ci * CO + co) * 343 + (342 - tap)] : wp[((long long)co * CI + ci) * 343 + tap];
      }
      reinterpret_cast<unsigned short*>(s_tb)[idx] = cvt_out<CF>(v);
    }
    for (int i = tid; i < 4 * PLANE / 16; i += NTHR) reinterpret_cast<uint4*>(s_in)[i] = make_uint4(0, 0, 0, 0);
  }
  const int jn = nn / CO, con = nn % CO;
  const int kd_l = PPM == 1 ? 0 : (g4 >> 1);
  const int gs_l = PPM == 1 ? g4 : (g4 & 1);
  const int a_col = PPM == 1 ? 2 * g4 : 4 * (g4 & 1);
  float bias = 0.f;
  if (a.p.b[0]) bias = a.p.b[0][con];

  // ---- staging: item = (tile row, aligned 8-voxel group) x CI channels: CI 16-byte loads from clamped addresses, zeroed by a mask
  // when the row / group / plane is outside the volume, interleaved channels-last with v_perm and written as 64 (32) contiguous
  // bytes.  (conv7_mfma_kernel unpacks to 16-bit values and writes voxel by voxel under a column test: ~180 vector instructions per
  // item, executed by all 8 waves for the 132 items of a plane -- SQ_INSTS_VALU 10.5 M per launch against 1.6 M MFMAs.)
  static_assert(NIT == 1, "one item per thread");
  const bool has_item = tid < NITEM;
  const int s_gi = (has_item ? tid : 0) % NG, s_hy = (has_item ? tid : 0) / NG;
  const int s_gh = oh0 - 3 + s_hy, s_gw = ow0 + (s_gi - 1) * 8;
  const bool s_inb = has_item && (unsigned)s_gh < (unsigned)H && s_gw >= 0 && s_gw < W;
  const ST* s_src = (const ST*)a.p.xa + n * a.d.xa_bs + (long long)min(max(s_gh, 0), H - 1) * W + min(max(s_gw, 0), W - 8);
  const int s_lds = (s_hy * IWP + s_gi * 8) * VB;
  uint4 raw[CI], raw2[F32S ? CI : 1];                   // fp32 storage: 8 voxels = two 16-byte loads per channel
  auto load_plane = [&](int gd) {
    const long long po = (long long)min(max(gd, 0), D - 1) * hw;
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      raw[c] = *reinterpret_cast<const uint4*>(s_src + (long long)c * dhw + po);
      if (F32S) raw2[c] = *reinterpret_cast<const uint4*>(s_src + (long long)c * dhw + po + 4);
    }
  };
  auto to16 = [&](int c) -> uint4 {                     // the channel's 8 voxels as 16-bit pairs
    if (!F32S) return raw[c];
    return make_uint4(cvt2_pack<CF>(__uint_as_float(raw[c].x), __uint_as_float(raw[c].y)), cvt2_pack<CF>(__uint_as_float(raw[c].z), __uint_as_float(raw[c].w)),
                      cvt2_pack<CF>(__uint_as_float(raw2[F32S ? c : 0].x), __uint_as_float(raw2[F32S ? c : 0].y)),
                      cvt2_pack<CF>(__uint_as_float(raw2[F32S ? c : 0].z), __uint_as_float(raw2[F32S ? c : 0].w)));
  };
  auto store_plane = [&](int gd) {
    if (!has_item) return;
    const unsigned m = (s_inb && (unsigned)gd < (unsigned)D) ? 0xffffffffu : 0u;
    unsigned char* dst = s_in + ((gd + 8) & 3) * PLANE + s_lds;
    if (CI == 4) {
      const uint4 r0 = to16(0), r1 = to16(1), r2 = to16(CI > 2 ? 2 : 0), r3 = to16(CI > 3 ? 3 : 0);
      const unsigned u[4][4] = {{r0.x, r0.y, r0.z, r0.w}, {r1.x, r1.y, r1.z, r1.w}, {r2.x, r2.y, r2.z, r2.w}, {r3.x, r3.y, r3.z, r3.w}};
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {                  // voxels 2 k2, 2 k2 + 1: [c0 c1 | c2 c3] each
        uint4 o;
        o.x = __builtin_amdgcn_perm(u[1][k2], u[0][k2], 0x05040100u) & m;
        o.y = __builtin_amdgcn_perm(u[3][k2], u[2][k2], 0x05040100u) & m;
        o.z = __builtin_amdgcn_perm(u[1][k2], u[0][k2], 0x07060302u) & m;
        o.w = __builtin_amdgcn_perm(u[3][k2], u[2][k2], 0x07060302u) & m;
        *reinterpret_cast<uint4*>(dst + k2 * 16) = o;
      }
    } else {
      const uint4 r0 = to16(0), r1 = to16(1);
      const unsigned u0[4] = {r0.x, r0.y, r0.z, r0.w}, u1[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {                  // voxels 4 h2 .. 4 h2 + 3: [c0 c1] each
        uint4 o;
        o.x = __builtin_amdgcn_perm(u1[2 * h2], u0[2 * h2], 0x05040100u) & m;
        o.y = __builtin_amdgcn_perm(u1[2 * h2], u0[2 * h2], 0x07060302u) & m;
        o.z = __builtin_amdgcn_perm(u1[2 * h2 + 1], u0[2 * h2 + 1], 0x05040100u) & m;
        o.w = __builtin_amdgcn_perm(u1[2 * h2 + 1], u0[2 * h2 + 1], 0x07060302u) & m;
        *reinterpret_cast<uint4*>(dst + h2 * 16) = o;
      }
    }
  };
  __syncthreads();                                      // table built, ring zeroed
  // ---- this wave pair's input rows rr = kq, kq + KSPLIT, ... and their B fragments, kept for the whole run ----
  // BREG: kept in registers for the whole run (112 registers for CI = 4: one workgroup per CU); else re-read from the table per use
  // (one 16-byte read per two MFMAs, the ring and table of TWO workgroups fit a CU and their phases overlap)
  bf16x8 bfr[BREG ? NR : 1][BREG ? NKD : 1];
  const int z_off = (7 * GS + gs_l) * CO * 16 + con * 16;  // (kd 0, kh 7): a zero row of the table
  int b_off[NR];                                        // table offset of (kd = kd_l, kh = rr - jn); + ks * PPM * 8 * GS * CO * 16 per tap step
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int rr = kq + i * KSPLIT;
    const int kh = rr - jn;
    const int khx = (rr < JR + 6 && (unsigned)kh < 7u) ? kh : 7;       // row 7 of the table is zero
    b_off[i] = (((kd_l * 8) + khx) * GS + gs_l) * CO * 16 + con * 16;
    if (BREG) {
#pragma unroll
      for (int ks = 0; ks < NKD; ++ks) bfr[i][ks] = *reinterpret_cast<const bf16x8*>(s_tb + b_off[i] + ks * PPM * 8 * GS * CO * 16);
    }
  }
  f32x4 acc[NS][MT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[s][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: base planes q0, q0 + 1 ----
  const int q0 = d_begin - 3, nq = (d_end - d_begin) + 6;
  load_plane(q0); store_plane(q0);
  load_plane(q0 + 1); store_plane(q0 + 1);
  __syncthreads();
  ST* ybase = (ST*)a.p.y + n * a.d.y_bs + (long long)con * dhw;
  const int oh = oh0 + wv * JR + jn;
  for (int qq = 0; qq < nq; qq += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int qi = qq + u;
      if (qi < nq) {                                    // uniform
        const int q = q0 + qi;
        load_plane(q + 2);                              // lands behind the MFMAs
        const unsigned char* pl = s_in + ((q + kd_l + 8) & 3) * PLANE;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int rr = kq + i * KSPLIT;
          if (rr < JR + 6) {                            // uniform per wave
            const unsigned char* ar = pl + ((wv * JR + rr) * IWP + nn + a_col + 5) * VB;   // column 0 = ow0 - 8; tap kw reads ow - 3 + kw
            bf16x8 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const unsigned char* ap = ar + mt * 16 * VB;
              if (CI == 4) {
                const uint2 lo = *reinterpret_cast<const uint2*>(ap);
                const uint2 hi = *reinterpret_cast<const uint2*>(ap + 8);
                av[mt] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
              } else {
                const unsigned* a32 = reinterpret_cast<const unsigned*>(ap);
                av[mt] = __builtin_bit_cast(bf16x8, make_uint4(a32[0], a32[1], a32[2], a32[3]));
              }
            }
#pragma unroll
            for (int ks = 0; ks < NKD; ++ks) {
              const int d = q + 3 - ks * PPM;           // the output plane this (base plane, tap step) pair feeds
              // planes outside the run take the table's ZERO row instead of a branch around the MFMA (the branches -- 56 per plane --
              // kept the compiler from issuing the fragment reads ahead of their MFMAs)
              const bool live = d >= d_begin && d < d_end;
              const bf16x8 bv = BREG ? (live ? bfr[BREG ? i : 0][BREG ? ks : 0] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0})
                                     : *reinterpret_cast<const bf16x8*>(s_tb + (live ? b_off[i] + ks * PPM * 8 * GS * CO * 16 : z_off));
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                acc[(u + 3 - ks * PPM + 2 * NS) % NS][mt] = mfma16x16x32<CF>(av[mt], bv, acc[(u + 3 - ks * PPM + 2 * NS) % NS][mt]);
            }
          }
        }
        // output plane d = q - 3 is complete in slot (u + 4) % NS
        const int d = q - 3;
        const bool emit = d >= d_begin;                 // (d < d_end always holds: q <= d_end + 2)
        float* part = s_part + (qi & 1) * PART;
        if (emit && kq > 0) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[((((kq - 1) * RT + wv) * MT + mt) * 4 + r) * 64 + lane] = acc[(u + 4) % NS][mt][r];
        }
        store_plane(q + 2);
        __syncthreads();                                // partial tiles visible; plane q + 2 staged; plane q no longer read
        if (emit && kq == 0) {
#pragma unroll
          for (int p_ = 0; p_ < KSPLIT - 1; ++p_)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[(u + 4) % NS][mt][r] += part[(((p_ * RT + wv) * MT + mt) * 4 + r) * 64 + lane];
          if (oh < H) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              float o[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = apply_act(acc[(u + 4) % NS][mt][r] + bias, a.d.act, a.d.act_slope);
              st4(ybase, ((long long)d * H + oh) * W + ow0 + mt * 16 + 4 * g4, o);
            }
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[(u + 4) % NS][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
}

int g_c7_as = 1;                                       // xh_set_option(24, v): 0 conv7_mfma_kernel, 1 conv7_as_kernel (weight fragments from LDS,
                                                       // two workgroups per CU) on volumes of >= 2^20 voxels, 2 the same with the weight
                                                       // fragments in registers, 3 conv7_as_kernel on every volume (tests)
// returns XH_OK if launched, 1 if the shape is not eligible (caller falls back to the vector kernel)
int xh_conv7_mfma_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p) {
  // fp32 storage with operands rounded once to fp16: the call's arithmetic mode (xh_conv_desc.arith)
  const bool f32s = d->dtype == XH_F32 && (d->arith & XH_ARITH_F32_SPLIT) && !(d->arith & XH_ARITH_K7_VECTOR);
  if ((d->dtype != XH_BF16 && d->dtype != XH_F16 && !f32s) || d->k != 7 || d->stride != 1 || d->groups != 1 || d->n_wptr != 1) return 1;
  if (!((d->Cin == 4 && d->Cout == 2) || (d->Cin == 2 && d->Cout == 4))) return 1;
  if (d->pre || d->epi || d->Ca != d->Cin) return 1;
  if (d->W % 32 != 0 || d->Wo != d->W || d->Ho != d->H || d->Do != d->D) return 1;
  if ((d->xa_bs & 7) || (d->y_bs & 3) || (((long long)d->D * d->H * d->W) & 7)) return 1;
  if (f32s && (d->act != XH_ACT_NONE && d->act != XH_ACT_SIGMOID && d->act != XH_ACT_LRELU && d->act != XH_ACT_RELU)) return 1;
  if (d->N > 65535) return 1;
  Conv7K a;
  a.d = *d;
  a.p = *p;
  const int rows = 2 * (16 / d->Cout);
  a.tilesW = d->W / 32;
  a.tilesH = cdiv(d->H, rows);
  const int cols = a.tilesW * a.tilesH;
  extern int g_c7_as;
  // input-stationary variant: measured (tools/microbench_k7.py, bf16) forward / data gradient at 128^3 45.2 / 41.1 us against 50.6 / 43.4
  // (weight fragments in registers, one workgroup per CU: 48.3 / 51.6), at 64^3 17.8 / 14.0 against 15.3 / 11.2, at 32^3 16.9 / 11.2
  // against 14.2 / 8.7: runs of 2 planes there stage 8 planes for 2 outputs either way and the longer prologue loses -- so it takes the
  // volumes of >= 2^20 voxels only.  The LDS traffic per MFMA fell from 1.5 KB to 0.64 KB (0.14 KB with the weights in registers) and the
  // run time by a tenth: the kernel was never bound by LDS BANDWIDTH but by the read -> MFMA dependency chains of few resident waves.
  const bool as = f32s || (g_c7_as != 0 && (g_c7_as > 2 || (long long)d->D * d->H * d->W >= (1 << 20)));
  int dsegs = cdiv(as && (f32s || g_c7_as == 2 || d->Cin == 4) ? 256 : 512, cols * d->N);   // weight fragments in registers: one workgroup per CU
  // runs of >= 8 planes (6 halo planes are staged per run) -- but on small volumes (<= 64^3) that leaves 8-64 workgroups on
  // 256 CUs and the run time is the serial chain of one workgroup (81 us at 32^3 and 64^3, like 128^3): runs of 2 there
  const int min_run = (long long)d->D * d->H * d->W <= (1 << 18) ? 2 : 8;
  const int max_segs = d->D >= min_run ? d->D / min_run : 1;
  if (dsegs > max_segs) dsegs = max_segs;
  if (dsegs < 1) dsegs = 1;
  a.sd = cdiv(d->D, dsegs);
  a.dsegs = cdiv(d->D, a.sd);
  dim3 grid(cols * a.dsegs, 1, d->N);
  hipStream_t st = (hipStream_t)stream;
  const int f = f32s ? 2 : d->dtype == XH_F16 ? 1 : 0;
  if (as) {
    static bool done_as[XH_MAX_DEV] = {};
    if (xh_attr_needed(done_as)) {
#define C7ATTR(F, CI_, CO_)                                                                                                             \
  (void)hipFuncSetAttribute((const void*)conv7_as_kernel<F, CI_, CO_, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);  \
  (void)hipFuncSetAttribute((const void*)conv7_as_kernel<F, CI_, CO_, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)
      C7ATTR(0, 4, 2); C7ATTR(1, 4, 2); C7ATTR(0, 2, 4); C7ATTR(1, 2, 4); C7ATTR(2, 4, 2); C7ATTR(2, 2, 4);
#undef C7ATTR
    }
    const size_t part = (size_t)2 * 3 * 2 * 2 * 4 * 64 * sizeof(float);
    // measured at 128^3 (after the staging rewrite): forward (CI = 4) 42.9 us with the weight fragments in registers, 46.7 from the table,
    // 49.3 output-stationary; data gradient (CI = 2) 43.9 / 37.4 / 44.1
    const bool breg = f32s || g_c7_as == 2 || (g_c7_as == 1 && d->Cin == 4);
#define C7AS(F, CI_, CO_, shm)                                                                                  \
  do {                                                                                                          \
    if (breg) hipLaunchKernelGGL((conv7_as_kernel<F, CI_, CO_, 4, true>), grid, dim3(512), shm, st, a);         \
    else hipLaunchKernelGGL((conv7_as_kernel<F, CI_, CO_, 4, false>), grid, dim3(512), shm, st, a);             \
  } while (0)
    if (d->Cin == 4) {
      const size_t shm = (size_t)4 * (2 * 8 + 6) * 48 * 8 + 8 * 8 * 4 * 2 * 16 + part;
      xh_note_kernel("conv7_as_kernel<%d, 4, 2>", f);
      if (f == 2) C7AS(2, 4, 2, shm); else if (f) C7AS(1, 4, 2, shm); else C7AS(0, 4, 2, shm);
    } else {
      const size_t shm = (size_t)4 * (2 * 4 + 6) * 48 * 4 + 8 * 8 * 2 * 4 * 16 + part;
      xh_note_kernel("conv7_as_kernel<%d, 2, 4>", f);
      if (f == 2) C7AS(2, 2, 4, shm); else if (f) C7AS(1, 2, 4, shm); else C7AS(0, 2, 4, shm);
    }
#undef C7AS
    return xh_launch_status();
  }
  if (d->Cin == 4) {
    const size_t shm = (size_t)8 * (2 * 8 + 6) * 40 * 8 + 8 * 8 * 4 * 2 * 16 + 3 * 2 * 2 * 4 * 64 * sizeof(float);
    static bool done[XH_MAX_DEV] = {};
    if (xh_attr_needed(done)) {
      (void)hipFuncSetAttribute((const void*)conv7_mfma_kernel<0, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      (void)hipFuncSetAttribute((const void*)conv7_mfma_kernel<1, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    }
    xh_note_kernel("conv7_mfma_kernel<%d, 4, 2>", f);
    if (f) hipLaunchKernelGGL((conv7_mfma_kernel<1, 4, 2>), grid, dim3(512), shm, st, a);
    else hipLaunchKernelGGL((conv7_mfma_kernel<0, 4, 2>), grid, dim3(512), shm, st, a);
  } else {
    const size_t shm = (size_t)8 * (2 * 4 + 6) * 40 * 4 + 8 * 8 * 2 * 4 * 16 + 3 * 2 * 2 * 4 * 64 * sizeof(float);
    xh_note_kernel("conv7_mfma_kernel<%d, 2, 4>", f);
    if (f) hipLaunchKernelGGL((conv7_mfma_kernel<1, 2, 4>), grid, dim3(512), shm, st, a);
    else hipLaunchKernelGGL((conv7_mfma_kernel<0, 2, 4>), grid, dim3(512), shm, st, a);
  }
  return xh_launch_status();
}
