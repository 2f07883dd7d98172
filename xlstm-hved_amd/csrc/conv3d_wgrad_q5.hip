// Full-row weight / bias gradient of the quad-channel 3x3x3 stride-1 convolution on rows of 128 or 64 voxels (round 5): every
// 128^3- and 64^3-level conv of XLSTM_HVED (buildingblocks.py:406-433 backward), 16-bit storage, gfx950.
//
//   dW[co][ci][kd][kh][kw] = sum over voxels v of dY[co][v] * xa[ci][v + (kd, kh, kw) - 1],   xa = leaky(x * sc + sh)
//
// Same GEMM as conv3d_wgrad_q4.hip (mfma_f32_16x16x32, K = 32 voxels of one row, M = (co of 4, kw), N = (ci of 4, kh), a wave
// walks along D so a B fragment meets the dY fragments of three planes), different staging.  What bound the 2 x 64 / 4 x 32 tile
// kernel was instruction issue (profiles/r03d_wgrad_q4_counters.txt: 185 vector + 135 scalar instructions per 128 voxels and
// channel-quad pair, two thirds of them building THREE shifted copies of every dY row -- 4 v_alignbit, DPP neighbour exchange,
// edge dwords, masks, three ds_write_b64 per 4 voxels -- because an MFMA operand wants its 8 voxels in one aligned 16-byte
// LDS read and the kw shift runs along the contraction axis), plus a 1.5 - 2x halo on x rows.
//
// Here dY is staged ONCE, unshifted, as a plain 16-byte copy (no vector arithmetic beyond the depth-segment mask) and the kw shift
// is applied when the A fragment is READ:
//   * lane (co, kw, g) reads the FIVE dwords that hold voxels 32 c + 8 g - 2 .. + 7 (kw = 2: one voxel to the left) or 32 c + 8 g ..
//     + 9 (kw = 1, 0) of its line -- the start is a per-lane address -- and funnels neighbouring dwords together with four
//     v_alignbit whose shift is 16 (kw = 0, 2) or 0 (kw = 1): no selects, no masks, no DPP, and each dY element is written to LDS
//     once instead of three times.  (gfx950 also serves one ds_read_b128 from a 2-byte-aligned address -- hipcc emits it for an
//     align-2 access -- but tools/probe/lds_unaligned.hip measured it at 1/9 of the aligned rate: 9.7 against 88 TB/s, 19 TB/s at
//     4-byte and 38 TB/s at 8-byte alignment; a fragment per 3 MFMAs at that rate makes the LDS pipe the bound.);
//   * every (row, channel) line has 16 bytes of zeros in front and behind, so the voxels beside a row are the conv's zero padding
//     and a row has no edge loads;
//   * a tile is 8 FULL rows (8 waves, wave w owns row w, 2 or 4 chunks of 32 voxels): every global load is a 16-byte piece of a
//     full line, 16 (8) consecutive lanes per 256-byte (128-byte) row; x halo 10 rows for 8 (1.25; 2.0 / 1.5 before);
//   * x is loaded as 16-byte items, the producer's InstanceNorm + LeakyReLU applied once per element (packed fp32), rows / planes
//     outside the volume become zero when staged: the plane step has no masks;
//   * 4-slot LDS ring of planes (slot p & 3 holds x plane p and dY plane p + 1), rounds of two planes, ONE barrier per round;
//     the loads of round r + 1 are in flight during the matrix phase of round r;
//   * a UNIT = (NQX input quads, NQY output quads) of one group staged once, every pair multiplied (round 6).  Rows of 128 voxels:
//     one quad of each (20 KB per plane slot) -- except groups of 3 k input quads (12 -> 4: the first decoder convs), three input quads
//     against ONE staging of dY on rounds of one plane and a two-slot ring (84 KB).  Rows of 64 / 32 voxels: up to two quads of each
//     operand (88 / 51 KB): 24 -> 8 at 64^3 was 12 single-quad units that moved 108 channel-volumes for 32 algorithmic, now 3 units
//     and 54.  A / B fragment reads per MFMA fall with it (2 + 2 reads for 12 MFMAs instead of 4 + 4);
//   * persistent workgroups (1 per CU: 239 registers), accumulators kept across a workgroup's tiles, ONE LDS reduction + pass of fp32
//     atomics;
//   * launch plan: all workgroups of a launch are resident together, so a launch lasts as long as its slowest workgroup -- workgroups
//     per unit by the min-max rule (xh_wgrad_q5_plan), problems dealt to launches by planned duration (xh_conv3d_wgrad_batch).
// Step of a wave and chunk (one quad of each): 5 ds_read_b32 + 4 v_alignbit (A), 1 ds_read_b128 (B), 3 MFMAs.
#include "common.h"
#include "../../include/xlstm_hved.h"
#include "wgrad_q4.h"
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <type_traits>

typedef h16x8 frag8;
typedef f32x4_t f32x4;

struct WgQ5Multi {
  int n;
  int off[Q5_MULTI + 1];
  WgQ4 p[Q5_MULTI];
};
static_assert(sizeof(WgQ5Multi) <= 4096, "the problem table travels in the kernel arguments");

namespace {
constexpr int Q5_TH = 8;                      // rows per tile = waves per workgroup
// NCH = 32-voxel K chunks per row: rows of 128 | 64 | 32 voxels; NQX / NQY = input / output channel quads of a unit (round 6: a
// unit of rows of 64 voxels stages up to two quads of each operand ONCE and multiplies every pair)
// PPR = planes per round of loads (2; 1 for the three-input-quad units on rows of 128 voxels, whose plane slot is 42 KB)
template <int NCH, int NQX = 1, int NQY = 1, int PPR = 2> struct Q5 {
  static constexpr int W = 32 * NCH;
  static constexpr int PR = W / 8;            // 16-byte pieces per row
  static constexpr int NC = NCH;
  static constexpr int CP = 2 * W + 16;       // bytes per (row, channel) line of x (16 B spare: bank spreading of the B reads)
  static constexpr int XROW = 4 * CP + 16;    // bytes per staged x row (4 channels)
  static constexpr int XB1 = (Q5_TH + 2) * XROW;        // one input quad
  static constexpr int XB = NQX * XB1;
  static constexpr int YP = 2 * W + 32;       // bytes per (row, co) line of dY: [16 B zeros][row][16 B zeros]  (2 W + 64, which takes the
                                              // four co lines of a row to distinct bank groups, measured 7 - 10 % SLOWER: 46.1 -> 49.5 us)
  static constexpr int YB1 = Q5_TH * 4 * YP;            // one output quad
  static constexpr int YB = NQY * YB1;
  static constexpr int SLOT = XB + YB;
  static constexpr int NSLOT = 2 * PPR;          // the round being read + the round being staged
  static constexpr int RING = NSLOT * SLOT;
  static constexpr int CONSTB = RING;         // [16 B of ones][16 B of zeros]
  static constexpr int TAIL = Q5_TH * NQX * NQY * (4 * 4 * 27 + 4) * 4;   // the per-wave slices of partial sums after the plane loops
  static constexpr int BYTES = RING + 32 > TAIL ? RING + 32 : TAIL;      // (two quads of each operand on rows of 32 voxels: the tail is larger)
  static constexpr int NXI = PPR * NQX * (Q5_TH + 2) * 4 * PR; // x items of a round (two planes)
  static constexpr int NIX = (NXI + 511) / 512;          // per thread: 3 | 2 | 1 (one quad)
  static constexpr int NYI = PPR * NQY * Q5_TH * 4 * PR; // dY items of a round
  static constexpr int NIY = (NYI + 511) / 512;          // 2 | 1 | 1 (rows of 32 voxels: half the threads)
};
}

template <int FMT, int NCH, int PD, int NQX = 1, int NQY = 1, int PPR = 2>
__device__ __forceinline__ void wgrad_q5_body(const WgQ4& a, int b, unsigned char* smem) {
  // FMT = 2 (round 6): fp32 STORAGE, operands rounded once to fp16 while staging (xh_conv_desc.arith XH_ARITH_F32_SPLIT; the same
  // arithmetic as conv3_wgrad_q4_multi_kernel<2, ...>): an item is still 8 voxels = two 16-byte loads, the LDS images are fp16
  constexpr bool F32S = FMT == 2;
  constexpr int CF = F32S ? 1 : FMT;                   // format of the LDS images / MFMA operands
  typedef typename std::conditional<F32S, float, h16<CF>>::type ST;
  typedef Q5<NCH, NQX, NQY, PPR> Q;
  constexpr int NSLOT = Q::NSLOT;
  constexpr int NP = NQX * NQY;
  constexpr int W = Q::W, PR = Q::PR, NC = Q::NC, NIX = Q::NIX, NIY = Q::NIY;
  constexpr unsigned ONE2 = CF == 0 ? 0x3F803F80u : 0x3C003C00u;
  float* s_dw = reinterpret_cast<float*>(smem);        // after the plane loops
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nn = lane & 15, g = lane >> 4;
  const int unit = b / a.wpu, w = b - unit * a.wpu;
  const int oq = unit / a.nchunk, chunk = unit - oq * a.nchunk;
  // A unit's work is the linear sequence of its (column = (sample, 8-row tile), plane) pairs; workgroup w of the unit takes the
  // w-th of wpu equal shares of it -- a contiguous run of planes, split where it crosses into the next column -- so every workgroup
  // of a launch has the same number of plane steps to within one (tiles of a fixed depth left 5 - 30 % of the 256 workgroups
  // short of work or unused: the count per unit had to divide the tile count)
  const int D_ = a.D;
  const long long planes = (long long)a.tilesH * a.N * D_;
  long long p_lo = planes * w / a.wpu;
  const long long p_hi = planes * (w + 1) / a.wpu;
  f32x4 acc[NP][3];                                    // pair (qx, qy) = qx * NQY + qy
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) acc[p][kd] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int D = a.D, H = a.H;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  const int co0 = oq * 4 * NQY;                        // a unit's output quads lie in one group (Cout_g % (4 NQY) == 0)
  const int grp = co0 / a.Cout_g;
  const int cin_base = grp * a.Cin_g + chunk * 4 * NQX;
  const float pslope = a.pre ? a.pre_slope : 1.f;
  const f32x2_t ps2 = {pslope, pslope};

  // ---- constants of the launch: the zero blocks in front of and behind every dY line, the ones / zeros block ----
  for (int i = tid; i < NSLOT * NQY * Q5_TH * 4 * 2; i += 512) {
    const int slot = i / (NQY * Q5_TH * 4 * 2), l = (i >> 1) % (NQY * Q5_TH * 4);
    *reinterpret_cast<uint4*>(smem + slot * Q::SLOT + Q::XB + l * Q::YP + ((i & 1) ? 16 + 2 * W : 0)) = make_uint4(0, 0, 0, 0);
  }
  if (tid < 8) *reinterpret_cast<unsigned*>(smem + Q::CONSTB + tid * 4) = tid < 4 ? ONE2 : 0u;

  // ---- fragment addresses ----
  // B: lane (ci = nn & 3, kh = nn >> 2; g) reads x row wv + kh, channel ci, voxels 32 c + 8 g ..; kh = 3: the constant column (bias gradient)
  const int ci_l = nn & 3, khB = nn >> 2;
  const int b_off = khB == 3 ? Q::CONSTB + (ci_l == 0 ? 0 : 16) : (wv + khB) * Q::XROW + ci_l * Q::CP + g * 16;
  const int b_slot = khB == 3 ? 0 : 1;                 // the constant block is not in a plane slot
  const int b_cstep = b_slot ? 64 : 0, b_qstep = b_slot ? Q::XB1 : 0;
  // A: lane (co = nn >> 2, kw = nn & 3; g) wants dY row wv, channel co, voxels 32 c + 8 g + 1 - kw .. + 7 (kw = 3: an unused
  // accumulator row, as kw = 1): five dwords from the one that holds its first voxel, funnel-shifted by 16 bits unless kw = 1
  const int kwA = (nn & 3) == 3 ? 1 : (nn & 3);
  const int a_off = Q::XB + (wv * 4 + (nn >> 2)) * Q::YP + 16 + 16 * g - (kwA == 2 ? 4 : 0);
  const unsigned a_sh = kwA == 1 ? 0u : 16u;

  // ---- staging plan, tile-independent part.  x item k of this thread: it = tid + 512 k = (plane pp, row r, channel c, piece j) ----
  int x_lds[NIX], x_pp[NIX], x_r[NIX], x_c[NIX];
  bool x_do[NIX];
#pragma unroll
  for (int k = 0; k < NIX; ++k) {
    const int it = tid + 512 * k;
    x_do[k] = it < Q::NXI;
    const int itc = x_do[k] ? it : 0;
    const int j = itc % PR, c = (itc / PR) & 3, r = (itc / (4 * PR)) % (Q5_TH + 2);
    const int qx = (itc / (4 * PR * (Q5_TH + 2))) % NQX, pp = itc / (4 * PR * (Q5_TH + 2) * NQX);
    x_pp[k] = pp; x_r[k] = r; x_c[k] = qx * 4 + c;     // channel inside the unit
    x_lds[k] = pp * Q::SLOT + qx * Q::XB1 + r * Q::XROW + c * Q::CP + j * 16;
  }
  int y_lds[NIY], y_pp[NIY];
  bool y_do[NIY];
  unsigned y_goff[NIY];                                // element offset inside the (sample, output quad) block, row h0 excluded
#pragma unroll
  for (int k = 0; k < NIY; ++k) {
    y_do[k] = tid + 512 * k < Q::NYI;
    const int it = y_do[k] ? tid + 512 * k : 0;
    const int j = it % PR, co = (it / PR) & 3, r = (it / (4 * PR)) % Q5_TH;
    const int qy = (it / (4 * PR * Q5_TH)) % NQY, pp = it / (4 * PR * Q5_TH * NQY);
    y_pp[k] = pp;
    y_lds[k] = pp * Q::SLOT + Q::XB + qy * Q::YB1 + (r * 4 + co) * Q::YP + 16 + j * 16;
    y_goff[k] = (unsigned)((long long)(qy * 4 + co) * dhw + (long long)r * W + 8 * j);
  }

  while (p_lo < p_hi) {
    const int col = (int)(p_lo / D_);
    const int d0 = (int)(p_lo - (long long)col * D_), d1 = (int)min((long long)D_, d0 + (p_hi - p_lo));
    p_lo += d1 - d0;
    const int th = col % a.tilesH, n = col / a.tilesH;
    const int h0 = th * Q5_TH;
    // ---- per-tile part of the x plan ----
    // the source of an input quad (Ca % 4 == 0: a quad lies in one of the two sources); one quad per unit: one pointer
    auto quad_src = [&](int cq) {
      return a.bcast ? (const ST*)a.xa + n * a.xa_bs + (long long)(cq >> 2) * dhw                     // one stored channel, four transforms
                     : (cq < a.Ca ? (const ST*)a.xa + n * a.xa_bs + (long long)cq * dhw
                                  : (const ST*)a.xb + n * a.xb_bs + (long long)(cq - a.Ca) * dhw);
    };
    const ST* xsrc[NIX];
    unsigned x_goff[NIX];
    float x_sc[NIX], x_sh[NIX];
#pragma unroll
    for (int k = 0; k < NIX; ++k) {
      const int row = h0 - 1 + x_r[k];
      const bool rok = (unsigned)row < (unsigned)H;
      const int it = tid + 512 * k;
      const int j = (x_do[k] ? it : 0) % PR;
      xsrc[k] = quad_src(cin_base + (x_c[k] & ~3));
      x_goff[k] = (unsigned)((a.bcast ? 0ll : (long long)(x_c[k] & 3) * dhw) + (long long)min(max(row, 0), H - 1) * W + 8 * j);
      float sc = 1.f, sh = 0.f;
      if (a.pre) { sc = a.pre_sc[n * a.Cin + cin_base + x_c[k]]; sh = a.pre_sh[n * a.Cin + cin_base + x_c[k]]; }
      x_sc[k] = rok ? sc : 0.f;
      x_sh[k] = rok ? sh : 0.f;
    }
    const ST* ysrc = (const ST*)a.dy + n * a.dy_bs + (long long)co0 * dhw + (long long)h0 * W;

    // PD rounds of loads in flight per thread, in PD register buffers used round-robin (the loop is unrolled PD times so that a
    // buffer is a fixed set of registers).  Every issue is unconditional -- a round beyond the tile's last one loads one dead
    // line (all offsets collapse to the source's first bytes) -- so that the compiler's vmcnt bookkeeping stays exact: a load
    // under a branch makes it wait for (nearly) everything at the next use, i.e. one round in flight whatever PD says.
    struct RB { uint4 x[NIX]; uint4 y[NIY]; uint4 x2[F32S ? NIX : 1]; uint4 y2[F32S ? NIY : 1]; };   // x2 / y2: the second half of an fp32 item
    RB q[PD];
    const int nround = (d1 - d0 + 2 + PPR - 1) / PPR;    // x planes d0 - 1 .. d1
    auto issue = [&](int r, RB& rb) {
      const int p0 = d0 - 1 + PPR * r;
      const bool live = r < nround;
      const unsigned lm = live ? 0xffffffffu : 0u;
#pragma unroll
      for (int k = 0; k < NIX; ++k) {
        const long long po = live ? (long long)min(max(p0 + x_pp[k], 0), D - 1) * hw : 0ll;
        rb.x[k] = *reinterpret_cast<const uint4*>(xsrc[k] + po + (x_goff[k] & lm));
        if constexpr (F32S) rb.x2[k] = *reinterpret_cast<const uint4*>(xsrc[k] + po + (x_goff[k] & lm) + 4);
      }
#pragma unroll
      for (int k = 0; k < NIY; ++k) {
        const long long po = live ? (long long)min(max(p0 + 1 + y_pp[k], 0), D - 1) * hw : 0ll;
        rb.y[k] = *reinterpret_cast<const uint4*>(ysrc + po + (y_goff[k] & lm));
        if constexpr (F32S) rb.y2[k] = *reinterpret_cast<const uint4*>(ysrc + po + (y_goff[k] & lm) + 4);
      }
    };
    auto commit = [&](int r, const RB& rb) {             // round r -> slots (2 r) & 3, (2 r + 1) & 3  (PPR = 1: slot r & 1)
      const int p0 = d0 - 1 + PPR * r;
      unsigned char* dst = smem + ((PPR * r) & (NSLOT - 1)) * Q::SLOT;
#pragma unroll
      for (int k = 0; k < NIX; ++k) {
        if (!x_do[k]) continue;
        const float pm = (unsigned)(p0 + x_pp[k]) < (unsigned)D ? 1.f : 0.f;
        const float sc = x_sc[k] * pm, sh = x_sh[k] * pm;
        const f32x2_t sc2 = {sc, sc}, sh2 = {sh, sh};
        const uint4 xh = F32S ? rb.x2[k] : rb.x[k];
        const unsigned u[8] = {rb.x[k].x, rb.x[k].y, rb.x[k].z, rb.x[k].w, xh.x, xh.y, xh.z, xh.w};
        uint4 o;
        unsigned* op = &o.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2_t xin = F32S ? f32x2_t{__uint_as_float(u[2 * e]), __uint_as_float(u[2 * e + 1])} : cvt2_in<CF>(u[e]);
          const f32x2_t v = xin * sc2 + sh2;
          const f32x2_t y = max2(v, v * ps2);
          op[e] = cvt2_pack<CF>(y.x, y.y);
        }
        *reinterpret_cast<uint4*>(dst + x_lds[k]) = o;
      }
#pragma unroll
      for (int k = 0; k < NIY; ++k) {
        if (!y_do[k]) continue;
        // out plane v = p + 1 belongs to this tile when v < d1 (v >= d0 always holds)
        const unsigned am = (p0 + 1 + y_pp[k] < d1) ? 0xffffffffu : 0u;
        uint4 yv = rb.y[k];
        if constexpr (F32S) {
          const uint4 a0 = rb.y[k], a1 = rb.y2[k];
          yv = make_uint4(cvt2_pack<CF>(__uint_as_float(a0.x), __uint_as_float(a0.y)), cvt2_pack<CF>(__uint_as_float(a0.z), __uint_as_float(a0.w)),
                          cvt2_pack<CF>(__uint_as_float(a1.x), __uint_as_float(a1.y)), cvt2_pack<CF>(__uint_as_float(a1.z), __uint_as_float(a1.w)));
        }
        *reinterpret_cast<uint4*>(dst + y_lds[k]) = make_uint4(yv.x & am, yv.y & am, yv.z & am, yv.w & am);
      }
    };
    frag8 af_m1[NQY][NC], af_0[NQY][NC];
#pragma unroll
    for (int qy = 0; qy < NQY; ++qy)
#pragma unroll
      for (int c = 0; c < NC; ++c) af_m1[qy][c] = af_0[qy][c] = frag8{0, 0, 0, 0, 0, 0, 0, 0};
    auto step = [&](int s) {                             // plane slot s: x plane p, dY plane p + 1
      const unsigned char* src = smem + s * Q::SLOT;
      const unsigned char* bsrc = smem + (b_slot ? s * Q::SLOT : 0) + b_off;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        frag8 bf[NQX];
#pragma unroll
        for (int qx = 0; qx < NQX; ++qx) bf[qx] = *reinterpret_cast<const frag8*>(bsrc + c * b_cstep + qx * b_qstep);
#pragma unroll
        for (int qy = 0; qy < NQY; ++qy) {
          const unsigned* ap = reinterpret_cast<const unsigned*>(src + a_off + qy * Q::YB1 + c * 64);
          const unsigned e0 = ap[0], e1 = ap[1], e2 = ap[2], e3 = ap[3], e4 = ap[4];
          const frag8 af_p1 = __builtin_bit_cast(frag8, make_uint4(__builtin_amdgcn_alignbit(e1, e0, a_sh), __builtin_amdgcn_alignbit(e2, e1, a_sh),
                                                                   __builtin_amdgcn_alignbit(e3, e2, a_sh), __builtin_amdgcn_alignbit(e4, e3, a_sh)));
#pragma unroll
          for (int qx = 0; qx < NQX; ++qx) {
            f32x4* ac = acc[qx * NQY + qy];
            ac[0] = mfma16x16x32<CF>(af_p1, bf[qx], ac[0]);
            ac[1] = mfma16x16x32<CF>(af_0[qy][c], bf[qx], ac[1]);
            ac[2] = mfma16x16x32<CF>(af_m1[qy][c], bf[qx], ac[2]);
          }
          af_m1[qy][c] = af_0[qy][c];
          af_0[qy][c] = af_p1;
        }
      }
    };
    __syncthreads();                                     // the previous tile's planes are no longer read (first tile: constants written)
#pragma unroll
    for (int u = 0; u < PD; ++u) issue(u, q[u]);
    commit(0, q[0]);
    issue(PD, q[0]);
    for (int r0 = 0; r0 < nround; r0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int r = r0 + u;
        if (r < nround) {                                // uniform
          __syncthreads();                               // round r staged; round r - 1 fully read
#pragma unroll
          for (int i = 0; i < PPR; ++i) step((PPR * r + i) & (NSLOT - 1));
        }
        // round r + 1 into the slots of round r - 1 (a dead round is not committed: past the last barrier its target slots may
        // still be read); its buffer then takes round r + 1 + PD
        if (r + 1 < nround) commit(r + 1, q[(u + 1) % PD]);
        issue(r + 1 + PD, q[(u + 1) % PD]);
      }
    }
  }   // tiles

  // ---- sum the eight rows of the workgroup in LDS (the plane ring is free now), then one pass of fp32 atomics ----
  // Every wave STORES its accumulators into a slice of its own; the threads that issue the global atomics sum the eight slices.
  // (It was 28 ds_add_f32 per lane into one slice: LDS floating-point atomics retire about a lane per two cycles -- 14 336 of them
  // per workgroup, ~13 us at the end of every workgroup of a 115 us launch; found on the MFMA weight gradient of the deep levels,
  // conv3d_wgrad_mfma.hip.)
  constexpr int NW = 4 * 4 * 27, NSL = NW + 4;
  static_assert(Q5_TH * NP * NSL * 4 <= Q::BYTES, "the tail's slices fit the launch's LDS");
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    float* my = s_dw + (wv * NP + p) * NSL;
    const int kh = nn >> 2, ci = nn & 3;                // accumulator column; rows 4 g + r = (co = g, kw = r)
    if (kh < 3) {
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int r = 0; r < 3; ++r) my[(g * 4 + ci) * 27 + kd * 9 + kh * 3 + r] = acc[p][kd][r];
    } else if (ci == 0) {
      my[NW + g] = acc[p][1][1];                         // column of ones x dY row (co = g, kw = 1)
    }
  }
  __syncthreads();
  for (int i = tid; i < NP * NSL; i += 512) {            // wave 0's slices become the sums (each element is read and written by one thread)
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < Q5_TH; ++w) v += s_dw[w * NP * NSL + i];
    s_dw[i] = v;
  }
  __syncthreads();
  if (a.abl & 2048) return;                              // ablation: no global atomics
  const int gpp = a.groups / a.n_wptr;
  float* dwp = a.dw[grp / gpp];
  const int gl = grp % gpp;
  for (int i = tid; i < NP * NW; i += 512) {
    const int p = i / NW, e = i - p * NW;                // pair (qx, qy), element of its 4 x 4 x 27 block
    const int qx = p / NQY, qy = p - qx * NQY;
    const int tap = e % 27;
    const int r = e / 27;
    const int ci = r & 3, c = r >> 2;
    const int co_g = (co0 + 4 * qy + c) % a.Cout_g;
    const float v = s_dw[p * NSL + e];
    if (a.dwm) {                                         // depthwise: dw[C][1][27], the off-diagonal products are not gradients (one quad per unit)
      if (ci == c) atomicAdd(dwp + (long long)(gl * 4 + c) * 27 + tap, v);
      continue;
    }
    atomicAdd(dwp + ((long long)(gl * a.Cout_g + co_g) * a.Cin_g + (chunk * NQX + qx) * 4 + ci) * 27 + tap, v);
  }
  float* dbp = a.db[grp / gpp];                          // bias gradient: the pairs of the first input quad of the group
  if (dbp && chunk == 0 && tid < 4 * NQY) atomicAdd(&dbp[gl * a.Cout_g + (co0 + tid) % a.Cout_g], s_dw[(tid >> 2) * NSL + NW + (tid & 3)]);
}

// Q5_PD = rounds of loads in flight per thread.  The first version had ONE (128 registers, two workgroups per CU): a round then
// lasts one memory latency whatever else overlaps -- 16 -> 16 g4 @128^3 60 us, no better than the tile kernel's 58, while 4 -> 4
// (28 against 38 us) and the 64^3 problems (19 - 35 against 25 - 43 us) already won on their lower instruction count.  Three rounds
// in 204 registers = ONE workgroup per CU with 3 x 36 KB of loads in flight all the time.
constexpr int Q5_PD = 3;
template <int FMT>
__global__ __launch_bounds__(512, 2) void conv3_wgrad_q5_multi_kernel(const WgQ5Multi m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < Q5_MULTI; ++k)
    if (k < m.n && b >= m.off[k]) i = k;
  const int local = b - m.off[i];
  const WgQ4& a = m.p[i];
  if (local >= a.nb) return;
  constexpr int PD = FMT == 2 ? 2 : Q5_PD;             // fp32 storage: a round in flight is twice the registers (and bytes)
  if (a.W == 128) {
    if (a.uqx == 3) wgrad_q5_body<FMT, 4, PD, 3, 1, 1>(a, local, smem);
    else wgrad_q5_body<FMT, 4, PD>(a, local, smem);
  } else if (a.W == 64) {
    if (a.uqx == 2 && a.uqy == 2) wgrad_q5_body<FMT, 2, PD, 2, 2>(a, local, smem);
    else if (a.uqy == 2) wgrad_q5_body<FMT, 2, PD, 1, 2>(a, local, smem);
    else if (a.uqx == 2) wgrad_q5_body<FMT, 2, PD, 2, 1>(a, local, smem);
    else wgrad_q5_body<FMT, 2, PD>(a, local, smem);
  } else {
    if (a.uqx == 2 && a.uqy == 2) wgrad_q5_body<FMT, 1, PD, 2, 2>(a, local, smem);
    else if (a.uqy == 2) wgrad_q5_body<FMT, 1, PD, 1, 2>(a, local, smem);
    else if (a.uqx == 2) wgrad_q5_body<FMT, 1, PD, 2, 1>(a, local, smem);
    else wgrad_q5_body<FMT, 1, PD>(a, local, smem);
  }
}

// Re-plans a quad-channel problem (xh_wgrad_q4_plan has filled `a`) for the full-row kernel; false: it stays with the tile kernel
int g_q5_on = 1;                                         // xh_set_option(21, 0 / 1)
int g_q5_w32 = 1;                                        // xh_set_option(23, 0 / 1): rows of 32 voxels take the full-row kernel too (default since
                                                         // units hold two quads of each operand: round 6)
int g_q5_wgs = 256;                                      // xh_set_option(22, n): workgroups per launch (one per CU is resident)
int g_q5_uq = 7;                                         // xh_set_option(28, bits): rows of 64 voxels, bit 0: two input quads per unit, bit 1: two output quads;
                                                         // bit 2: rows of 128 voxels, three input quads per unit (one plane per round)
                                                         // bit 3 (off): fp32 storage takes this kernel too (FMT = 2: measured SLOWER than the tile
                                                         // kernel in the fp32_mfma step, 2 x 348 us against 592 us: 32-byte items as two 16-byte loads
                                                         // at a lane stride of 32 bytes, two rounds in flight in 251 registers)
bool xh_wgrad_q5_replan(const xh_conv_desc* d, WgQ4* a) {
  { static bool env_done = false;                        // (measurement overrides of the option defaults)
    if (!env_done) { env_done = true; const char* e = getenv("XH_Q5_UQ"); if (e) g_q5_uq = atoi(e) & 15;
      e = getenv("XH_Q5_W32"); if (e) g_q5_w32 = atoi(e) ? 1 : 0; } }
  const bool f32 = d->dtype == XH_F32 && (d->arith & XH_ARITH_F32_SPLIT) && (g_q5_uq & 8);   // fp32 storage, fp16 operands
  if (!g_q5_on || (d->dtype != XH_BF16 && d->dtype != XH_F16 && !f32)) return false;
  extern int g_q5_w32;
  // rows of 32 voxels (xh_set_option(23, 0) sends them back to the tile kernel): with one quad of each operand per unit the 32^3
  // problems of the network were ~200 units of a few tiles each and the launch waited for them (batch of the step's 24 problems: 433 us
  // with them here, 392 us all on the tile kernel); as units of two quads of each operand they are 59 units and ride in the two
  // full-row launches (+10 us each) instead of three tile-kernel launches (55 us)
  if ((d->W != 128 && d->W != 64 && !(d->W == 32 && g_q5_w32)) || d->H % Q5_TH || d->D < 4) return false;
  const long long dhw = (long long)d->D * d->H * d->W;
  if (4 * dhw >= (1ll << 31)) return false;              // 32-bit element offsets inside a channel quad
  a->full = 1;
  a->ci4 = 1;
  // rows of 64 voxels: a unit stages two input and / or two output quads of a group once and multiplies every pair (the LDS ring of
  // two quads of each operand is 88 KB there; at 128 voxels a second input quad alone takes it to 125 KB and ~250 registers).
  // One quad per unit re-stages dY once per input quad and x once per output quad: 24 -> 8 moved 108 channel-volumes for 32.
  a->uqx = a->uqy = 1;
  if ((d->W == 64 || d->W == 32) && !a->dwm && !d->bcast && 8 * dhw < (1ll << 31)) {
    if ((g_q5_uq & 1) && (a->Cin_g / 4) % 2 == 0) a->uqx = 2;
    if ((g_q5_uq & 2) && (a->Cout_g / 4) % 2 == 0) a->uqy = 2;
  }
  // rows of 128 voxels, groups of 3 k input quads (12 -> 4: the recon | seg decoders' first convs at 128^3): three quads against ONE
  // staging of dY -- 42 KB per plane slot, so a round is one plane and the ring two slots (84 KB)
  if (d->W == 128 && !a->dwm && !d->bcast && (g_q5_uq & 4) && (a->Cin_g / 4) % 3 == 0 && 12 * dhw < (1ll << 31)) a->uqx = 3;
  a->nchunk = a->Cin_g / (4 * a->uqx);
  a->nq = (d->Cout / (4 * a->uqy)) * a->nchunk;
  a->wide = 0;
  a->tilesW = 1;
  a->tilesH = d->H / Q5_TH;
  return true;                                           // dsegs / sd / ntile / wpu / nb: per launch (xh_wgrad_q5_launch)
}

// Cost of ONE unit of a problem in rounds of a single-quad unit on rows of 128 voxels (a round = 8 rows x 2 planes; its cost is mostly
// instruction issue and grows slowly with the row width: measured ~3 / ~2 us at 128 / 64 voxels; units of several quads stage
// (uqx + uqy) / 2 x the bytes and run uqx uqy x the matrix work: factors fitted to the step's batch, tools/scan_q5_step.sh)
static double q5_ucost(const WgQ4& w) {
  static double f64 = -1.0, f2, f4, f3, f32;
  if (f64 < 0) {                                         // (measurement overrides)
    const char* e;
    f2 = (e = getenv("XH_Q5_F2")) ? atof(e) : 1.45; f4 = (e = getenv("XH_Q5_F4")) ? atof(e) : 1.9; f3 = (e = getenv("XH_Q5_F3")) ? atof(e) : 2.4;
    f32 = (e = getenv("XH_Q5_F32")) ? atof(e) : 0.55;
    f64 = (e = getenv("XH_Q5_F64")) ? atof(e) : 0.7;
  }
  const int pairs = w.uqx * w.uqy;
  return (double)w.N * w.D * (w.H / Q5_TH) * (w.W == 128 ? (w.uqx == 3 ? f3 : 1.0) : (w.W == 64 ? f64 : f32) * (pairs == 1 ? 1.0 : pairs == 2 ? f2 : f4));
}

// Workgroups per unit of the problems of ONE launch: every workgroup of a unit walks an equal run of the unit's planes and all of a
// launch's workgroups are resident together (one per CU), so the launch lasts as long as its slowest workgroup: the plan is the
// smallest T with  sum_i nq_i ceil(cost_i / T) <= budget  (min-max; never more than `budget` workgroups unless one per unit
// already is -- a 257th workgroup would wait for a CU and double the launch).  Until round 6 this was "integer part of the
// proportional share, then largest remainders": units of 3.3 shares got 3 workgroups and ran 10 % longer than the rest.
// Returns the planned duration max_i cost_i / wq_i (xh_conv3d_wgrad_batch balances its launches with it).
extern "C" double xh_wgrad_plan_minmax(int n, const double* cost, const int* units, const int* cap, int budget, int* wq) {
  if (n <= 0 || !cost || !units || !cap || !wq || budget < 1) return -1.0;
  double total = 0.0, hi = 0.0;
  for (int i = 0; i < n; ++i) {
    if (!(cost[i] > 0.0) || units[i] < 1 || cap[i] < 1) return -1.0;
    total += units[i] * cost[i];
    hi = cost[i] > hi ? cost[i] : hi;
  }
  auto need = [&](double T, int* out) {
    long long used = 0;
    for (int i = 0; i < n; ++i) {
      const double q = ceil(cost[i] / T - 1e-9);
      int k = q > (double)cap[i] ? cap[i] : (int)q;
      k = k < 1 ? 1 : k;
      if (out) out[i] = k;
      used += (long long)units[i] * k;
    }
    return used;
  };
  if (need(hi, wq) <= budget) {                          // (else: more units than workgroups -- one each, already in wq)
    double lo = total / budget * 0.999;                  // below the smallest conceivable T
    for (int it = 0; it < 48; ++it) {                    // need() falls with T: bisect to the smallest feasible T
      const double mid = 0.5 * (lo + hi);
      if (need(mid, nullptr) <= budget) hi = mid; else lo = mid;
    }
    long long used = need(hi, wq);
    // workgroups left over (they cannot lower the maximum): to the units that are slowest now, while whole units fit
    for (;;) {
      int best = -1;
      for (int i = 0; i < n; ++i)
        if (used + units[i] <= budget && wq[i] + 1 <= cap[i] && (best < 0 || cost[i] / wq[i] > cost[best] / wq[best])) best = i;
      if (best < 0) break;
      ++wq[best];
      used += units[best];
    }
  }
  double T = 0.0;
  for (int i = 0; i < n; ++i) T = cost[i] / wq[i] > T ? cost[i] / wq[i] : T;
  return T;
}

double xh_wgrad_q5_plan(const WgQ4* probs, int n, int budget, int* wq) {
  double c[Q5_MULTI];
  int cap[Q5_MULTI], nq[Q5_MULTI];
  for (int i = 0; i < n; ++i) {
    c[i] = q5_ucost(probs[i]);
    const long long planes = (long long)probs[i].tilesH * probs[i].N * probs[i].D;
    cap[i] = (int)(planes / 4 > 0 ? planes / 4 : 1);     // runs of at least 4 planes
    nq[i] = probs[i].nq;
  }
  return xh_wgrad_plan_minmax(n, c, nq, cap, budget, wq);
}

// launches up to Q5_MULTI re-planned problems of one storage format
void xh_wgrad_q5_launch(hipStream_t st, int fmt, const WgQ4* probs, int n) {
  WgQ5Multi m;
  m.n = n;
  m.off[0] = 0;
  extern int g_q5_wgs;
  const int budget = g_q5_wgs;                           // resident workgroups: 1 per CU
  int wq[Q5_MULTI];
  (void)xh_wgrad_q5_plan(probs, n, budget, wq);
  for (int i = 0; i < n; ++i) {
    m.p[i] = probs[i];
    WgQ4& a = m.p[i];
    a.sd = a.D; a.dsegs = 1; a.ntile = a.tilesH * a.N;             // (informational: the kernel cuts the plane sequence itself)
    a.wpu = wq[i];
    a.nb = a.nq * wq[i];
    static const bool dump = getenv("XH_Q5_DUMP") != nullptr;       // (measurement aid: the plan of every launch on stderr)
    if (dump) fprintf(stderr, "q5 plan %d: %d->%d g%d W%d uq %dx%d nq %d wpu %d\n", i, a.Cin, a.Cout, a.groups, a.W, a.uqx, a.uqy, a.nq, a.wpu);
    m.off[i + 1] = m.off[i] + a.nb;
  }
  static bool attr_done[XH_MAX_DEV] = {};
  if (xh_attr_needed(attr_done)) {
    constexpr int m0 = Q5<2, 2, 2>::BYTES > Q5<4>::BYTES ? Q5<2, 2, 2>::BYTES : Q5<4>::BYTES;
    constexpr int mx = Q5<4, 3, 1, 1>::BYTES > m0 ? Q5<4, 3, 1, 1>::BYTES : m0;
    (void)hipFuncSetAttribute((const void*)conv3_wgrad_q5_multi_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute((const void*)conv3_wgrad_q5_multi_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute((const void*)conv3_wgrad_q5_multi_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
  }
  size_t shm = 0;
  for (int i = 0; i < n; ++i) {
    const WgQ4& w = probs[i];
    const size_t need = w.W == 128 ? (w.uqx == 3 ? Q5<4, 3, 1, 1>::BYTES : Q5<4>::BYTES)
                        : w.W == 32 ? (w.uqx == 2 && w.uqy == 2 ? Q5<1, 2, 2>::BYTES : w.uqy == 2 ? Q5<1, 1, 2>::BYTES : w.uqx == 2 ? Q5<1, 2, 1>::BYTES : Q5<1>::BYTES)
                        : w.uqx == 2 && w.uqy == 2 ? Q5<2, 2, 2>::BYTES : w.uqy == 2 ? Q5<2, 1, 2>::BYTES : w.uqx == 2 ? Q5<2, 2, 1>::BYTES : Q5<2>::BYTES;
    shm = need > shm ? need : shm;
  }
  xh_note_kernel("conv3_wgrad_q5_multi_kernel<%d>", fmt);
  if (fmt == 2) hipLaunchKernelGGL((conv3_wgrad_q5_multi_kernel<2>), dim3(m.off[n]), dim3(512), shm, st, m);
  else if (fmt) hipLaunchKernelGGL((conv3_wgrad_q5_multi_kernel<1>), dim3(m.off[n]), dim3(512), shm, st, m);
  else hipLaunchKernelGGL((conv3_wgrad_q5_multi_kernel<0>), dim3(m.off[n]), dim3(512), shm, st, m);
}
