// Weight / bias gradient of the 3x3x3 stride-1 convolution for groups of FEW channels (4 -> 4, 12 -> 4, the four-stream
// 16 -> 16 g4, ...: every 128^3-class conv of XLSTM_HVED), 16-bit storage, gfx950.  Companion of conv3d_q4.hip.
//
//   dW[co][ci][kd][kh][kw] = sum over voxels v of dY[co][v] * xa[ci][v + (kd, kh, kw) - 1],   xa = leaky(x * sc + sh)
//
// GEMM on mfma_f32_16x16x32 with K = 32 voxels of one row, BOTH operands straight from global memory (no LDS staging, no
// barrier in the main loop):
//   M = (co of 4, kw):  A[(co, kw)][w] = dY[co][row][w + 1 - kw]   -- the kw shift is applied to dY: an aligned 16-byte load
//                        plus the neighbouring dword and four v_alignbit; dY never needs a halo, edge elements are masked
//   N = (ci of 4, kh):  B[w][(ci, kh)] = xa[ci][row + kh - 1][w]   -- ONE aligned 16-byte load per lane, the producer's
//                        InstanceNorm + LeakyReLU applied in registers; rows / planes outside the volume get scale = shift = 0
//   kd: a wave walks along D; the x fragment of plane p meets the dY fragments of planes p+1, p, p-1 (three accumulator
//       tiles), so every x and dY element is loaded once per wave.
//   Column (ci = 0, kh = 3) -- unused by the taps -- holds the constant 1, so the same MFMAs deliver the bias gradient.
// A lane's accumulator tile is dW[co = lane>>4][ci = lane&3][kd][kh = (lane>>2)&3][kw = register]: the four waves of a workgroup
// (four consecutive rows) are summed through LDS and leave one pass of fp32 atomics.
//
// Measured on MI355X (tools/microbench_wgrad_q4.py, 4 -> 4 @128^3: 26 us without / 44 us with the atomics tail of a lone
// problem; batched launches hide the tail): the main loop sits at ~1.7 TB/s of algorithmic traffic and did NOT move with
// prefetch depth 2 -> 4, 4 vs 6 waves per SIMD, two output rows per wave (x fragment of four rows shared by two dY rows: half
// the input transforms, 2/3 of the x fetches), 64-voxel rows per wave (whole 128-byte lines) or non-power-of-two volumes
// (96^3, 160^3: same time per voxel) -- those variants were removed again; SQ counters: 51 % of wave cycles waiting on memory,
// 34 % waiting to issue, SIMD issue slots 59 % busy.  Compile-time ablation (tools/abl_wq4.sh, 16 -> 16 g4 @128^3, us without
// the atomics tail): full 86, MFMAs removed 86, x loads removed 52, dY loads removed 45, both removed 31: the loads cost 55 us
// ON TOP of 31 us of vector / scalar work -- they do not overlap with it although 12 load sets per SIMD are in flight, and
// 134 MB in 55 us is 2.4 TB/s: the vector-memory path (16 distinct half-lines per x load, 4-fold replicated dY lanes), not HBM.
#include "common.h"
#include "../../include/xlstm_hved.h"
#include "wgrad_q4.h"

typedef h16x8 frag8;
typedef f32x4_t f32x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Buffer descriptor of a wave-uniform base pointer (readfirstlane makes the uniformity provable to the compiler: without it
// every buffer_load is wrapped in a waterfall loop).  Loads then take a 32-bit lane offset and a 32-bit scalar offset:
// no 64-bit address arithmetic in vector registers.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t q4_rsrc(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}

struct WgQ4Multi {
  int n;
  int off[WQ_MULTI + 1];         // first workgroup of problem i (multiples of 8: the XCD remap of each problem stays valid)
  WgQ4 p[WQ_MULTI];
};

template <int FMT, int CI4>
__device__ __forceinline__ void wgrad_q4_body(const WgQ4& a, int b, float* s_dw) {
  typedef h16<FMT> ST;
  constexpr int DEPTH = CI4 == 2 ? 3 : 2;              // steps of loads in flight per wave (register budget, wq4_waves)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nn = lane & 15, g = lane >> 4;
  // Persistent workgroups: unit = (problem, output-channel quad); its a.wpu workgroups share the unit's a.ntile spatial
  // tiles and keep their accumulators across tiles, so the LDS reduction + atomics tail is paid once per workgroup (a
  // workgroup per tile left 1024 x 432 same-address atomics per unit: 35 of the 46 us of a 4 -> 4 @128^3 problem).
  // Workgroups that share an XCD (equal id mod 8) walk one contiguous eighth of the tiles side by side.
  const int unit = b / a.wpu, w = b - unit * a.wpu;
  const int oq = unit / a.nchunk, chunk = unit - oq * a.nchunk;   // unit = (output quad, chunk of CI4 input quads)
  int t_first, t_stride, t_end;
  if ((a.ntile & 7) == 0 && (a.wpu & 7) == 0) {
    const int per = a.ntile >> 3;
    t_first = (w & 7) * per + (w >> 3); t_stride = a.wpu >> 3; t_end = ((w & 7) + 1) * per;
  } else {
    t_first = w; t_stride = a.wpu; t_end = a.ntile;
  }
  f32x4 acc[CI4][3];
#pragma unroll
  for (int cq = 0; cq < CI4; ++cq)
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) acc[cq][kd] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int D = a.D, H = a.H, W = a.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  const int co0 = oq * 4;
  const int grp = co0 / a.Cout_g;
  const int cin_base = grp * a.Cin_g + chunk * 4 * CI4;
  for (int t = t_first; t < t_end; t += t_stride) {
  int wk = t;
  const int tw = wk % a.tilesW; wk /= a.tilesW;
  const int th = wk % a.tilesH; wk /= a.tilesH;
  const int ds = wk % a.dsegs;
  const int n = wk / a.dsegs;
  const int h = th * 4 + wv, w0 = tw * 32;
  const bool hrow = h < H;
  const int d0 = ds * a.sd, d1 = min(D, d0 + a.sd);

  // ---- B role: column (ci = nn & 3, kh = nn >> 2) ----
  const int ci_l = nn & 3, khB = nn >> 2;
  const int xrow = h + khB - 1;
  const bool rowok = hrow && khB < 3 && (unsigned)xrow < (unsigned)H;
  const int xrow_c = min(max(xrow, 0), H - 1);
  // loads are "uniform base (SGPRs) + 32-bit lane offset": the lane part is fixed per tile, the plane advances the base
  const unsigned x_off = (unsigned)(((long long)ci_l * dhw + (long long)xrow_c * W + w0 + 8 * g) * 2);     // bytes
  __amdgpu_buffer_rsrc_t xrs[CI4];
  float scl[CI4], shl[CI4];
#pragma unroll
  for (int cq = 0; cq < CI4; ++cq) {
    const int c0 = cin_base + cq * 4, c = c0 + ci_l;
    xrs[cq] = q4_rsrc(c0 < a.Ca ? (const ST*)a.xa + n * a.xa_bs + (long long)c0 * dhw
                                : (const ST*)a.xb + n * a.xb_bs + (long long)(c0 - a.Ca) * dhw);
    float sc = 1.f, sh = 0.f;
    if (a.pre) { sc = a.pre_sc[n * a.Cin + c]; sh = a.pre_sh[n * a.Cin + c]; }
    if (!rowok) { sc = 0.f; sh = 0.f; }
    if (khB == 3 && cq == 0 && ci_l == 0) { sc = 0.f; sh = 1.f; }             // the column of ones: bias gradient
    scl[cq] = sc; shl[cq] = sh;
  }
  const bool onescol = khB == 3;                       // not subject to the plane mask
  const float pslope = a.pre ? a.pre_slope : 1.f;
  // ---- A role: row (co = nn >> 2, kw = nn & 3) ----
  const int co_l = nn >> 2, kwA = nn & 3;
  const int hc = min(h, H - 1);
  const __amdgpu_buffer_rsrc_t dyrs = q4_rsrc((const ST*)a.dy + n * a.dy_bs + (long long)co0 * dhw);
  const unsigned dy_off = (unsigned)(((long long)co_l * dhw + (long long)hc * W + w0 + 8 * g) * 2);          // bytes
  const int wpos = w0 + 8 * g;
  const bool left = kwA == 2, right = kwA == 0;
  // neighbouring dword: elements (wpos - 2, wpos - 1) for kw = 2, (wpos + 8, wpos + 9) for kw = 0; at a row end any valid address
  const int e_off = (left && wpos > 0) ? -2 : (right && wpos + 8 < W) ? 8 : 0;
  const unsigned m0 = (left && wpos == 0) ? 0xffff0000u : 0xffffffffu;       // element w = -1 does not exist
  const unsigned m3 = (right && wpos + 8 == W) ? 0x0000ffffu : 0xffffffffu;  // element w = W does not exist
  const unsigned shbits = (left || right) ? 16u : 0u;

  const unsigned dye_off = dy_off + 2 * e_off;
  const int hw2 = (int)(hw * 2);                       // bytes per plane
  auto load_x = [&](int p, uint4 (&raw)[CI4]) {
    const int po = min(max(p, 0), D - 1) * hw2;
#pragma unroll
    for (int cq = 0; cq < CI4; ++cq)
#ifdef WQ4_ABL_NOX
      raw[cq] = make_uint4(po, x_off, 0, 0);
#else
      raw[cq] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs[cq], (int)x_off, po, 0));
#endif
  };
  auto load_dy = [&](int v, uint4& cur, unsigned& ex) {
    const int po = min(max(v, 0), D - 1) * hw2;
#ifdef WQ4_ABL_NODY
    cur = make_uint4(po, dy_off, 0, 0); ex = dye_off;
#else
    // the four kw lanes of an output channel want the same 16 bytes: only the kw = 1 lane loads them (a quarter of the lane
    // addresses through the vector-memory path), make_a hands them to its quad by DPP; the neighbour dword is needed by the
    // kw = 0 / 2 lanes only
    cur = make_uint4(0, 0, 0, 0);
    ex = 0;
    if (kwA == 1) cur = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(dyrs, (int)dy_off, po, 0));
    if (left || right) ex = __builtin_amdgcn_raw_buffer_load_b32(dyrs, (int)dye_off, po, 0);
#endif
  };
  auto make_a = [&](const uint4& cl, unsigned ex) -> frag8 {
    // lane 1 of every quad of lanes (kw = 1) holds the loaded 16 bytes: quad broadcast (DPP quad_perm [1, 1, 1, 1])
    uint4 c;
    c.x = (unsigned)__builtin_amdgcn_mov_dpp((int)cl.x, 0x55, 0xf, 0xf, true);
    c.y = (unsigned)__builtin_amdgcn_mov_dpp((int)cl.y, 0x55, 0xf, 0xf, true);
    c.z = (unsigned)__builtin_amdgcn_mov_dpp((int)cl.z, 0x55, 0xf, 0xf, true);
    c.w = (unsigned)__builtin_amdgcn_mov_dpp((int)cl.w, 0x55, 0xf, 0xf, true);
    // kw = 2: window one voxel to the left, kw = 0: one to the right, kw = 1: as loaded
    const unsigned l0 = left ? ex : c.x, l1 = left ? c.x : c.y, l2 = left ? c.y : c.z, l3 = left ? c.z : c.w;
    const unsigned h0 = left ? c.x : c.y, h1 = left ? c.y : c.z, h2 = left ? c.z : c.w, h3 = left ? c.w : ex;
    uint4 o;                                           // centre rows: shift 0 returns the low operand = the loaded dword
    o.x = __builtin_amdgcn_alignbit(h0, l0, shbits) & m0;
    o.y = __builtin_amdgcn_alignbit(h1, l1, shbits);
    o.z = __builtin_amdgcn_alignbit(h2, l2, shbits);
    o.w = __builtin_amdgcn_alignbit(h3, l3, shbits) & m3;
    return __builtin_bit_cast(frag8, o);
  };
  auto make_b = [&](const uint4& r, float sc, float sh) -> frag8 {
    const unsigned u[4] = {r.x, r.y, r.z, r.w};
    uint4 o;
    unsigned* op = &o.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float lo = cvt_lo<FMT>(u[k]) * sc + sh, hi = cvt_hi<FMT>(u[k]) * sc + sh;
      lo = fmaxf(lo, lo * pslope);
      hi = fmaxf(hi, hi * pslope);
      op[k] = cvt_pack<FMT>(lo, hi);
    }
    return __builtin_bit_cast(frag8, o);
  };

  if (hrow) {
    // A wave is alone with its memory latency (no barrier, no partner doing other work): the loads of DEPTH steps are in
    // flight at any time, in a register ring unrolled DEPTH times.  Step p consumes x plane p and dY plane p + 1, then
    // refills its slot with step p + DEPTH.  The steady-state loop is branch-free -- loads from clamped planes, dY
    // fragments of planes outside [d0, d1) zeroed when they are built, every MFMA unconditional -- because a branch
    // inside it makes hipcc wait for vmcnt(0) at every step, i.e. one step in flight whatever DEPTH says; steps are
    // rounded up to a multiple of DEPTH (the extra ones multiply zeros).
    frag8 af_m1 = frag8{0, 0, 0, 0, 0, 0, 0, 0}, af_0 = af_m1;                   // dY fragments of out planes p - 1, p
    uint4 xraw[DEPTH][CI4], dcur[DEPTH];
    unsigned dex[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) { load_x(d0 - 1 + u, xraw[u]); load_dy(d0 + u, dcur[u], dex[u]); }
    const int p_end = d0 - 1 + (d1 - d0 + 2 + DEPTH - 1) / DEPTH * DEPTH;
    for (int p0 = d0 - 1; p0 < p_end; p0 += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int p = p0 + u;
        const unsigned amask = p + 1 < d1 ? 0xffffffffu : 0u;                    // out plane p + 1 belongs to this tile
        const uint4 ar = __builtin_bit_cast(uint4, make_a(dcur[u], dex[u]));
        const frag8 af_p1 = __builtin_bit_cast(frag8, make_uint4(ar.x & amask, ar.y & amask, ar.z & amask, ar.w & amask));
        const float pm = (unsigned)p < (unsigned)D ? 1.f : 0.f;
        frag8 bf[CI4];
#pragma unroll
        for (int cq = 0; cq < CI4; ++cq) {
          const float sc = onescol ? scl[cq] : scl[cq] * pm, sh = onescol ? shl[cq] : shl[cq] * pm;
          bf[cq] = make_b(xraw[u][cq], sc, sh);
        }
        load_x(p + DEPTH, xraw[u]);
        load_dy(p + DEPTH + 1, dcur[u], dex[u]);
#ifdef WQ4_ABL_NOMFMA
#pragma unroll
        for (int cq = 0; cq < CI4; ++cq) asm volatile("" ::"v"(af_p1), "v"(bf[cq]));
#else
#pragma unroll
        for (int cq = 0; cq < CI4; ++cq) {             // out plane v = p + 1 - kd
          acc[cq][0] = mfma16x16x32<FMT>(af_p1, bf[cq], acc[cq][0]);
          acc[cq][1] = mfma16x16x32<FMT>(af_0, bf[cq], acc[cq][1]);
          acc[cq][2] = mfma16x16x32<FMT>(af_m1, bf[cq], acc[cq][2]);
        }
#endif
        af_m1 = af_0;
        af_0 = af_p1;
      }
    }
  }

  }   // tiles

  // ---- sum the four rows of the workgroup in LDS: s_dw[(co_l * 4 CI4 + ci) * 27 + tap], then [4] bias sums ----
  // (every wave STORES into a slice of its own, the slices are summed by the threads that issue the global atomics: LDS floating-
  // point atomics retire about a lane per two cycles -- see conv3d_wgrad_mfma.hip)
  constexpr int NW = 4 * 4 * CI4 * 27, NSL = NW + 4;
  {
    float* my = s_dw + (tid >> 6) * NSL;
    const int kh = nn >> 2, ci = nn & 3;               // accumulator column
    if (kh < 3) {
#pragma unroll
      for (int cq = 0; cq < CI4; ++cq)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
          for (int r = 0; r < 3; ++r)                  // D row 4 * g + r = (co = g, kw = r)
            my[(g * 4 * CI4 + cq * 4 + ci) * 27 + kd * 9 + kh * 3 + r] = acc[cq][kd][r];
    } else if (ci == 0) {
      my[NW + g] = acc[0][1][1];                       // column of ones x dY row (co = g, kw = 1)
    }
  }
  __syncthreads();
  for (int i = tid; i < NSL; i += 256) s_dw[i] = ((s_dw[i] + s_dw[NSL + i]) + s_dw[2 * NSL + i]) + s_dw[3 * NSL + i];
  __syncthreads();
  if (a.abl & 2048) return;                            // ablation: no global atomics
  const int gpp = a.groups / a.n_wptr;
  float* dwp = a.dw[grp / gpp];
  const int gl = grp % gpp;
  for (int i = tid; i < NW; i += 256) {
    const int tap = i % 27;
    const int r = i / 27;
    const int ci = r % (4 * CI4), c = r / (4 * CI4);
    const int co_g = (co0 + c) % a.Cout_g;
    if (a.dwm) {                                         // depthwise: dw[C][1][27], the off-diagonal products are not gradients
      if (ci == c) atomicAdd(dwp + (long long)(gl * 4 + c) * 27 + tap, s_dw[i]);
      continue;
    }
    atomicAdd(dwp + ((long long)(gl * a.Cout_g + co_g) * a.Cin_g + chunk * 4 * CI4 + ci) * 27 + tap, s_dw[i]);
  }
  float* dbp = a.db[grp / gpp];
  if (dbp && chunk == 0 && tid < 4) atomicAdd(&dbp[gl * a.Cout_g + (co0 + tid) % a.Cout_g], s_dw[NW + tid]);
}

// ---------------------------------------------------------------------------------------------------------------
// Variant with BOTH operands staged through LDS (the one the launches use).  In the body above every wave loads (and
// transforms) its own three x rows and builds its dY fragment per lane: a workgroup's four waves put 4 KB of lane loads into the
// vector-memory path per plane for 1.5 KB of distinct x, and 64 lanes each spend ~22 vector instructions (quad broadcast,
// selects, v_alignbit, masks) on a dY fragment whose distinct content is 256 bytes -- the kernel is bound by that path and by
// vector-instruction issue (SQ counters: 61 vector + 33 scalar instructions per step of 3 MFMAs), not by HBM.  Here, per plane:
//  * the 6 distinct x rows (h0 - 1 .. h0 + 4, 4 CI4 channels, 64 bytes each) are loaded ONCE per workgroup as 16-byte items,
//    transformed once (InstanceNorm + LeakyReLU; rows / planes outside the volume become zero) and written to the plane slot;
//  * the 4 dY rows (4 output channels, 64 bytes each) are loaded once as 8-byte items (one per thread); the item's thread builds
//    the three kw windows (one voxel left / centre / one voxel right: 4 v_alignbit, the neighbour dwords by DPP from the lanes
//    next to it, the two row-end dwords by one extra 4-byte load) and writes them interleaved [row][g][co][kw], so that the
//    A fragment of lane (co, kw, g) is ONE aligned ds_read_b128 and the 16 lanes of a g read 256 consecutive bytes;
//  * planes outside the depth segment and rows outside the volume are zeroed when staged: the step has no masks;
//  * the constant column (bias gradient) reads a 16-byte block of ones that sits in every plane slot.
// A step = 1 + CI4 ds_read_b128 and 3 CI4 MFMAs.  4-plane LDS ring, one barrier per two planes, the global loads of a round
// issued two rounds ahead (3 + 4 NIX VGPRs per round in flight; every thread has the same loads, none under a branch).
// FMT 2 = fp32 STORAGE (xh_set_option(18, 1)): both operands are read as fp32, the x operand transformed in fp32, and both go to
// LDS as fp16 (CF below): single-rounded fp16 operands, fp32 accumulation.  A weight gradient is a sum over ~10^6 voxels whose
// operand roundings are independent: the deviation of the sum is ~3e-4 of its size, far inside the parity mode's gradient band;
// the activation gradients (dY) need the caller's loss scale to sit in fp16's range, like the data gradients of conv3d_q4s.hip.
template <int FMT> struct WqStore { typedef h16<FMT> T; };
template <> struct WqStore<2> { typedef float T; };
template <int FMT, int CI4>
__device__ __forceinline__ void wgrad_q4_body_lds(const WgQ4& a, int b, unsigned char* smem) {
  typedef typename WqStore<FMT>::T ST;
  constexpr int CF = FMT == 2 ? 1 : FMT;               // format of the LDS images / MFMA operands
  constexpr bool F32S = FMT == 2;
  // Two tile shapes, chosen per problem at run time (a.wide; everything that differs is address arithmetic outside the plane loop):
  //   narrow: 4 rows x 32 voxels -- wave w owns row w; 6 x rows of 64 bytes per channel and plane;
  //   wide:   2 rows x 64 voxels -- wave w owns row w >> 1, 32-voxel half w & 1; 4 x rows of 128 bytes per channel and plane.  For
  //           rows of 64 / 128 voxels: every x and dY load is then a piece of a FULL 128-byte line and a row has one pair of
  //           edge dwords per 64 voxels instead of per 32 (the texture path, not HBM, bounds this kernel: DESIGN 3.6).  Same item
  //           count per thread (128 CI4 x items per plane instead of 96 CI4 padded to 128 CI4), same dY windows, same MFMAs.
  const bool wide = a.wide != 0;
  const int CB = wide ? 144 : 64;                     // bytes per channel inside a staged x row (wide: 128 + 16, bank spreading)
  const int ROWB = 4 * CI4 * CB + 16;                 // bytes per staged x row; + 16 B so that the 12 (ci, kh) lanes of a B-fragment
                                                      // read fall on different 16-byte bank groups
  constexpr int XB = (6 * (4 * CI4 * 64 + 16) > 4 * (4 * CI4 * 144 + 16)) ? 6 * (4 * CI4 * 64 + 16) : 4 * (4 * CI4 * 144 + 16);   // x rows of a plane
  constexpr int CONSTB = XB;                          // [16 B of ones][16 B of zeros]
  constexpr int DYB = XB + 32;                        // dY windows: [row 4][g 4][co 4][kw 4] x 16 B
  constexpr int PLB = DYB + 4096;                     // bytes per plane slot
  const int NITX = (wide ? 128 : 96) * CI4;           // 16-byte x items per plane
  constexpr int NIX = CI4;                            // x items per thread (a thread serves one of the two planes of a round)
  constexpr unsigned ONE2 = CF == 0 ? 0x3F803F80u : 0x3C003C00u;
  float* s_dw = reinterpret_cast<float*>(smem);       // after the plane loops
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nn = lane & 15, g = lane >> 4;
  const int unit = b / a.wpu, w = b - unit * a.wpu;
  const int oq = unit / a.nchunk, chunk = unit - oq * a.nchunk;
  int t_first, t_stride, t_end;
  if ((a.ntile & 7) == 0 && (a.wpu & 7) == 0) {
    const int per = a.ntile >> 3;
    t_first = (w & 7) * per + (w >> 3); t_stride = a.wpu >> 3; t_end = ((w & 7) + 1) * per;
  } else {
    t_first = w; t_stride = a.wpu; t_end = a.ntile;
  }
  f32x4 acc[CI4][3];
#pragma unroll
  for (int cq = 0; cq < CI4; ++cq)
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) acc[cq][kd] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int D = a.D, H = a.H, W = a.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  const int co0 = oq * 4;
  const int grp = co0 / a.Cout_g;
  const int cin_base = grp * a.Cin_g + chunk * 4 * CI4;
  const float pslope = a.pre ? a.pre_slope : 1.f;
  // plane of the round this thread stages (wave-uniform: the plane addresses are scalar arithmetic), id among its 128 threads
  const int pp = __builtin_amdgcn_readfirstlane(tid >> 7), tl = tid & 127;
  // B fragment of lane (ci = nn & 3, kh = nn >> 2) in a staged plane; kh = 3 is the constant column (bias gradient): ones for
  // (quad 0, ci 0), zeros elsewhere
  const int ci_l = nn & 3, khB = nn >> 2;
  int b_off[CI4];
#pragma unroll
  for (int cq = 0; cq < CI4; ++cq)
    b_off[cq] = khB == 3 ? CONSTB + ((cq == 0 && ci_l == 0) ? 0 : 16)
                         : ((wide ? (wv >> 1) : wv) + khB) * ROWB + (cq * 4 + ci_l) * CB + (wide ? (wv & 1) * 64 : 0) + g * 16;
  // A fragment of lane (co = nn >> 2, kw = nn & 3): window kw of row wv (kw = 3: an unused accumulator row, reads window 1).
  // Slot of window (co, kw) inside the 256-byte block of (row, g): co * 4 + (kw ^ g) -- the 16 lanes of a g read 16 different
  // slots, and the 32 lanes of a staging write (one kw, all (g, co), both halves) cover the 64 banks once
  const int a_off = DYB + ((wv * 4 + g) * 16 + (nn >> 2) * 4 + (((nn & 3) == 3 ? 1 : (nn & 3)) ^ g)) * 16;
  // the constant blocks of the four plane slots (never overwritten by the staging)
  if (tid < 32) {
    const int slot = tid >> 3, d = tid & 7;
    *reinterpret_cast<unsigned*>(smem + slot * PLB + CONSTB + d * 4) = d < 4 ? ONE2 : 0u;
  }
  // dY item of this thread: row r, output channel co, 8-byte chunk gq (4 voxels) of the 64-byte row.  Every thread has one (and
  // the same number of loads in flight: a load under a branch makes hipcc wait for nearly all of them at each commit)
  // (wide: 16 chunks per 128-byte row; the row's two halves are the "rows" 2 r, 2 r + 1 of the window image, i.e. of the waves)
  const int y_q = wide ? (tl & 15) : (tl & 7);        // chunk inside the global row
  const int y_gq = y_q & 7, y_co = wide ? (tl >> 4) & 3 : (tl >> 3) & 3, y_r = wide ? tl >> 6 : tl >> 5;
  const int y_vr = wide ? 2 * y_r + (y_q >> 3) : y_r;
  const bool y_first = y_q == 0, y_last = y_q == (wide ? 15 : 7);
  const int y_g = y_gq >> 1;
  const int y_lds = DYB + ((y_vr * 4 + y_g) * 16 + y_co * 4) * 16 + (y_gq & 1) * 8;             // + (kw ^ g) * 16
  const int y_lds0 = y_lds + (0 ^ y_g) * 16, y_lds1 = y_lds + (1 ^ y_g) * 16, y_lds2 = y_lds + (2 ^ y_g) * 16;

  for (int t = t_first; t < t_end; t += t_stride) {
    int wk = t;
    const int tw = wk % a.tilesW; wk /= a.tilesW;
    const int th = wk % a.tilesH; wk /= a.tilesH;
    const int ds = wk % a.dsegs;
    const int n = wk / a.dsegs;
    const int h0 = th * (wide ? 2 : 4), w0 = tw * (wide ? 64 : 32), tww = wide ? 64 : 32;
    const int d0 = ds * a.sd, d1 = min(D, d0 + a.sd);
    // ---- x staging plan of this thread: items tl, tl + 128, ... of a plane = (row r, quad cq, channel ci, chunk gq) ----
    const ST* i_src[NIX];
    float i_sc[NIX], i_sh[NIX];
    int i_lds[NIX];
    bool i_do[NIX];
#pragma unroll
    for (int k = 0; k < NIX; ++k) {
      const int it = tl + 128 * k;
      i_do[k] = it < NITX;
      const int itc = i_do[k] ? it : 0;
      const int gq = wide ? itc & 7 : itc & 3, ci = wide ? (itc >> 3) & 3 : (itc >> 2) & 3;
      const int cq = (wide ? itc >> 5 : itc >> 4) % CI4, r = itc / ((wide ? 32 : 16) * CI4);
      const int row = h0 - 1 + r;
      const bool rok = (unsigned)row < (unsigned)H;
      const int c = cin_base + cq * 4 + ci;
      i_src[k] = (c < a.Ca ? (const ST*)a.xa + n * a.xa_bs + (long long)c * dhw : (const ST*)a.xb + n * a.xb_bs + (long long)(c - a.Ca) * dhw) +
                 (long long)min(max(row, 0), H - 1) * W + w0 + 8 * gq;
      float sc = 1.f, sh = 0.f;
      if (a.pre) { sc = a.pre_sc[n * a.Cin + c]; sh = a.pre_sh[n * a.Cin + c]; }
      i_sc[k] = rok ? sc : 0.f;
      i_sh[k] = rok ? sh : 0.f;
      i_lds[k] = r * ROWB + (cq * 4 + ci) * CB + gq * 16;
    }
    // ---- dY staging plan ----
    const int yrow = h0 + y_r;
    const ST* y_src = (const ST*)a.dy + n * a.dy_bs + (long long)(co0 + y_co) * dhw + (long long)min(yrow, H - 1) * W + w0 + 4 * y_q;
    const bool y_edge_l = y_first && w0 > 0, y_edge_r = y_last && w0 + tww < W;
    const ST* y_esrc = y_src + (y_edge_l ? -2 : y_edge_r ? 4 : 0);               // the dword beyond the 64-byte row, if there is one
    const unsigned y_rowmask = yrow < H ? 0xffffffffu : 0u;
    const unsigned y_lmask = y_first ? (y_edge_l ? 0xffffffffu : 0u) : 0xffffffffu;   // chunk 0 at w = 0: voxel -1 does not exist
    const unsigned y_rmask = y_last ? (y_edge_r ? 0xffffffffu : 0u) : 0xffffffffu;    // last chunk at the row end: voxel W does not exist
    // ---- staging: round r covers x planes d0 - 1 + 2 r (+ pp) and the dY planes one above them ----
    // a staged item: 8 voxels of x (16 bytes of 16-bit storage, 32 of fp32), 4 voxels of dY + the 2 voxels beyond the row end
    struct XQ { uint4 a; uint4 b; };                   // b: fp32 storage only
    struct YQ { uint4 v; uint2 e; };                   // 16-bit storage: v.x, v.y and e.x only
    auto issue = [&](int r, XQ (&q)[NIX], YQ& yq) {
      const int p = d0 - 1 + 2 * r + pp;
      const long long po = (long long)min(max(p, 0), D - 1) * hw;
#pragma unroll
      for (int k = 0; k < NIX; ++k) {
        q[k].a = *reinterpret_cast<const uint4*>(i_src[k] + po);
        if (F32S) q[k].b = *reinterpret_cast<const uint4*>(i_src[k] + po + 4);
        else q[k].b = q[k].a;                          // unused
      }
      const long long yo = (long long)min(max(p + 1, 0), D - 1) * hw;
      if (F32S) {
        yq.v = *reinterpret_cast<const uint4*>(y_src + yo);
        yq.e = *reinterpret_cast<const uint2*>(y_esrc + yo);
      } else {
        const uint2 t = *reinterpret_cast<const uint2*>(y_src + yo);
        yq.v = uint4{t.x, t.y, 0u, 0u};
        yq.e.y = 0u;
        yq.e.x = *reinterpret_cast<const unsigned*>(y_esrc + yo);
      }
    };
    auto commit = [&](int r, const XQ (&q)[NIX], const YQ& yq) {
      const int p = d0 - 1 + 2 * r + pp;
      const float pm = (unsigned)p < (unsigned)D ? 1.f : 0.f;
      unsigned char* dst = smem + ((2 * r + pp) & 3) * PLB;
#pragma unroll
      for (int k = 0; k < NIX; ++k) {
        if (!i_do[k]) continue;
        const float sc = i_sc[k] * pm, sh = i_sh[k] * pm;
        const unsigned u[8] = {q[k].a.x, q[k].a.y, q[k].a.z, q[k].a.w, q[k].b.x, q[k].b.y, q[k].b.z, q[k].b.w};
        uint4 o;
        unsigned* op = &o.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2_t xin = F32S ? f32x2_t{__uint_as_float(u[2 * e]), __uint_as_float(u[2 * e + 1])} : cvt2_in<CF>(u[e]);
          const f32x2_t v = xin * f32x2_t{sc, sc} + f32x2_t{sh, sh};
          const f32x2_t y = max2(v, v * f32x2_t{pslope, pslope});
          op[e] = cvt2_pack<CF>(y.x, y.y);
        }
        *reinterpret_cast<uint4*>(dst + i_lds[k]) = o;
      }
      {
        // out plane v = p + 1 belongs to this tile when d0 <= v < d1 (v >= d0 always holds: p >= d0 - 1)
        const unsigned am = (p + 1 < d1 ? 0xffffffffu : 0u) & y_rowmask;
        unsigned y0 = yq.v.x, y1 = yq.v.y, ye = yq.e.x;
        if (F32S) {                                     // four fp32 voxels -> two packed fp16 pairs; the two beyond the row end -> one
          y0 = cvt2_pack<CF>(__uint_as_float(yq.v.x), __uint_as_float(yq.v.y));
          y1 = cvt2_pack<CF>(__uint_as_float(yq.v.z), __uint_as_float(yq.v.w));
          ye = cvt2_pack<CF>(__uint_as_float(yq.e.x), __uint_as_float(yq.e.y));
        }
        const unsigned c0 = y0 & am, c1 = y1 & am, e = ye & am;
        // neighbour dwords inside the row: the 8 lanes y_gq = 0..7 of a DPP row half hold its chunks (row_shr:1 / row_shl:1;
        // the lanes at a row end take the extra dword instead)
        const unsigned pl = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c1, 0x111, 0xf, 0xf, true);
        const unsigned nx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c0, 0x101, 0xf, 0xf, true);
        const unsigned prev = (y_first ? e : pl) & y_lmask, next = (y_last ? e : nx) & y_rmask;
        uint2 wl, wr;                                    // kw = 2: one voxel to the left; kw = 0: one voxel to the right
        wl.x = __builtin_amdgcn_alignbit(c0, prev, 16); wl.y = __builtin_amdgcn_alignbit(c1, c0, 16);
        wr.x = __builtin_amdgcn_alignbit(c1, c0, 16);   wr.y = __builtin_amdgcn_alignbit(next, c1, 16);
        *reinterpret_cast<uint2*>(dst + y_lds0) = wr;                             // kw = 0
        *reinterpret_cast<uint2*>(dst + y_lds1) = make_uint2(c0, c1);             // kw = 1
        *reinterpret_cast<uint2*>(dst + y_lds2) = wl;                             // kw = 2
      }
    };
    frag8 af_m1 = frag8{0, 0, 0, 0, 0, 0, 0, 0}, af_0 = af_m1;
    auto step = [&](int p) {                                 // x plane p and dY plane p + 1, both in slot (p - (d0 - 1)) & 3
      const unsigned char* src = smem + ((p - (d0 - 1)) & 3) * PLB;
      const frag8 af_p1 = *reinterpret_cast<const frag8*>(src + a_off);
      frag8 bf[CI4];
#pragma unroll
      for (int cq = 0; cq < CI4; ++cq) bf[cq] = *reinterpret_cast<const frag8*>(src + b_off[cq]);
#pragma unroll
      for (int cq = 0; cq < CI4; ++cq) {
        acc[cq][0] = mfma16x16x32<CF>(af_p1, bf[cq], acc[cq][0]);
        acc[cq][1] = mfma16x16x32<CF>(af_0, bf[cq], acc[cq][1]);
        acc[cq][2] = mfma16x16x32<CF>(af_m1, bf[cq], acc[cq][2]);
      }
      af_m1 = af_0;
      af_0 = af_p1;
    };
    const int nround = ((d1 - d0 + 2 + 1) / 2 + 1) & ~1;     // rounds of two planes, an even number of them
    XQ qa[NIX], qb[NIX];
    YQ ya, yb;
    __syncthreads();                                         // the previous tile's planes are no longer read
    issue(0, qa, ya);
    issue(1, qb, yb);
    commit(0, qa, ya);
    issue(2, qa, ya);
    for (int r = 0; r < nround; r += 2) {
      __syncthreads();                                       // round r staged; round r - 1 fully read
      commit(r + 1, qb, yb);
      issue(r + 3, qb, yb);
      step(d0 - 1 + 2 * r);
      step(d0 + 2 * r);
      __syncthreads();                                       // round r + 1 staged; round r fully read
      commit(r + 2, qa, ya);
      issue(r + 4, qa, ya);
      step(d0 + 1 + 2 * r);
      step(d0 + 2 + 2 * r);
    }
  }   // tiles

  // ---- sum the four rows of the workgroup in LDS (the plane ring is free now), then one pass of fp32 atomics ----
  constexpr int NW = 4 * 4 * CI4 * 27, NSL = NW + 4;
  __syncthreads();
  {
    float* my = s_dw + (tid >> 6) * NSL;                 // a slice per wave, stored (not atomically added: see wgrad_q4_body)
    const int kh = nn >> 2, ci = nn & 3;
    if (kh < 3) {
#pragma unroll
      for (int cq = 0; cq < CI4; ++cq)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
          for (int r = 0; r < 3; ++r)
            my[(g * 4 * CI4 + cq * 4 + ci) * 27 + kd * 9 + kh * 3 + r] = acc[cq][kd][r];
    } else if (ci == 0) {
      my[NW + g] = acc[0][1][1];
    }
  }
  __syncthreads();
  for (int i = tid; i < NSL; i += 256) s_dw[i] = ((s_dw[i] + s_dw[NSL + i]) + s_dw[2 * NSL + i]) + s_dw[3 * NSL + i];
  __syncthreads();
  if (a.abl & 2048) return;
  const int gpp = a.groups / a.n_wptr;
  float* dwp = a.dw[grp / gpp];
  const int gl = grp % gpp;
  for (int i = tid; i < NW; i += 256) {
    const int tap = i % 27;
    const int r = i / 27;
    const int ci = r % (4 * CI4), c = r / (4 * CI4);
    const int co_g = (co0 + c) % a.Cout_g;
    if (a.dwm) {                                         // depthwise: dw[C][1][27], the off-diagonal products are not gradients
      if (ci == c) atomicAdd(dwp + (long long)(gl * 4 + c) * 27 + tap, s_dw[i]);
      continue;
    }
    atomicAdd(dwp + ((long long)(gl * a.Cout_g + co_g) * a.Cin_g + chunk * 4 * CI4 + ci) * 27 + tap, s_dw[i]);
  }
  float* dbp = a.db[grp / gpp];
  if (dbp && chunk == 0 && tid < 4) atomicAdd(&dbp[gl * a.Cout_g + (co0 + tid) % a.Cout_g], s_dw[NW + tid]);
}

// resident workgroups per CU (= waves per SIMD) each instance is built for: 75 / 127 / 143 VGPRs, no spills
constexpr int wq4_waves(int ci4) { return ci4 == 1 ? 6 : ci4 == 2 ? 4 : 3; }
// ... of the LDS-staged form; fp32 storage (FMT 2) stages 8 floats per piece: at the 16-bit instances' budgets (128 / 168 registers)
// hipcc spilled its loop-invariant addresses and reloaded them every step behind s_waitcnt vmcnt(0) -- one workgroup less per CU
constexpr int wq4_resident(int fmt, int ci4) { return (ci4 == 1 ? 5 : ci4 == 2 ? 4 : 3) - (fmt == 2 && ci4 > 1 ? 1 : 0); }
template <int FMT, int CI4, bool LDSX>
__global__ __launch_bounds__(256, LDSX ? wq4_resident(FMT, CI4) : wq4_waves(CI4)) void conv3_wgrad_q4_multi_kernel(const WgQ4Multi m) {
  constexpr int XBN = 6 * (4 * CI4 * 64 + 16), XBW = 4 * (4 * CI4 * 144 + 16);  // x rows of a plane: narrow / wide tiles (wgrad_q4_body_lds)
  constexpr int RING = 4 * ((XBN > XBW ? XBN : XBW) + 32 + 4096);             // four plane slots: x rows, constant block, dY windows
  constexpr int RED = 4 * (4 * 4 * CI4 * 27 + 4) * 4;                        // a slice of partial sums per wave
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDSX && RING > RED ? RING : RED];
  const int b = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < WQ_MULTI; ++k)
    if (k < m.n && b >= m.off[k]) i = k;
  const int local = b - m.off[i];
  const WgQ4& a = m.p[i];
  if (local >= a.nb) return;
  if constexpr (LDSX || FMT == 2) wgrad_q4_body_lds<FMT, CI4>(a, local, smem);
  else wgrad_q4_body<FMT, CI4>(a, local, reinterpret_cast<float*>(smem));      // A/B: every wave loads its own x rows
}

// fills the plan; false when the shape is not for this kernel
bool xh_wgrad_q4_plan(const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR], WgQ4* a) {
  extern int g_xh_disable;
  if (g_xh_disable & 32) return false;
  const bool f32 = d->dtype == XH_F32 && (d->arith & XH_ARITH_F32_SPLIT);     // fp32 storage, fp16 operands (xh_conv_desc.arith)
  if ((d->dtype != XH_BF16 && d->dtype != XH_F16 && !f32) || d->k != 3 || d->stride != 1 || d->transposed) return false;
  if (d->groups <= 0 || d->Cin % d->groups || d->Cout % d->groups) return false;
  if (d->W % 32 != 0 || d->Wo != d->W || d->Ho != d->H || d->Do != d->D) return false;
  int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups, groups = d->groups;
  // depthwise: groups of 4 channels, the diagonal of each 4 x 4 block is the gradient (conv3d_q4.hip: q4_plan)
  const bool dwm = cin_g == 1 && cout_g == 1 && groups % 4 == 0 && d->n_wptr > 0 && (groups / 4) % d->n_wptr == 0 && !(g_xh_disable & 128);
  if (dwm) { cin_g = cout_g = 4; groups /= 4; }
  if (cin_g % 4 || cout_g % 4 || cin_g > 48 || cout_g > 48) return false;
  if (d->Ca % 4) return false;
  if ((d->xa_bs & 7) || (d->xb_bs & 7) || (d->ea_bs & 7)) return false;
  const long long dhw = (long long)d->D * d->H * d->W;
  if (dhw % 8 || dhw >= (f32 ? (1ll << 26) : (1ll << 27))) return false;     // 32-bit byte offsets inside a channel quad
  if (d->pre && !(d->pre_slope >= 0.f && d->pre_slope <= 1.f)) return false;
  if (!p->ea || !p->xa) return false;
  if (d->D < 4 || d->H < 4) return false;
  a->xa = p->xa; a->xb = p->xb; a->dy = p->ea;
  a->pre_sc = p->pre_sc; a->pre_sh = p->pre_sh;
  for (int i = 0; i < XH_MAX_WPTR; ++i) { a->dw[i] = i < d->n_wptr ? dw[i] : nullptr; a->db[i] = (i < d->n_wptr && db) ? db[i] : nullptr; }
  a->xa_bs = d->xa_bs; a->xb_bs = d->xb_bs; a->dy_bs = d->ea_bs;
  a->N = d->N; a->Cin = d->Cin; a->Cout = d->Cout; a->groups = groups; a->n_wptr = d->n_wptr; a->Ca = d->Ca;
  a->dwm = dwm ? 1 : 0;
  a->D = d->D; a->H = d->H; a->W = d->W;
  // input quads per unit: a workgroup keeps 3 accumulator tiles per quad, so groups of more than 3 quads are cut into
  // equal chunks of 3, 2 or 1 (each chunk re-reads the dY rows)
  const int q_all = cin_g / 4;
  const int cs = q_all <= 3 ? q_all : q_all % 3 == 0 ? 3 : q_all % 2 == 0 ? 2 : 1;
  a->Cin_g = cin_g; a->Cout_g = cout_g; a->ci4 = cs; a->nchunk = q_all / cs; a->pre = d->pre; a->pre_slope = d->pre_slope;
  // rows of 64 / 128 voxels: 2 x 64 tiles (full 128-byte lines); ablation bits 8192 (no LDS staging) / 524288 keep the 4 x 32 tiles
  { extern int g_mfma_abl; a->wide = (d->W % 64 == 0 && !(g_mfma_abl & (8192 | 524288))) ? 1 : 0; }
  if (d->dtype == XH_F32) a->wide = 0;                 // fp32 rows of 32 voxels are full lines already
  a->tilesW = a->wide ? d->W / 64 : d->W / 32; a->tilesH = cdiv(d->H, a->wide ? 2 : 4);
  a->nq = (d->Cout / 4) * a->nchunk;
  // depth segments: tiles of >= 8 planes (x is read sd + 2 planes per tile), ~1024 tiles per unit
  const long long cols = (long long)a->tilesW * a->tilesH * d->N;
  int dsegs = (int)((1024 + cols - 1) / cols);
  const int max_segs = d->D >= 8 ? d->D / 8 : 1;
  if (dsegs > max_segs) dsegs = max_segs;
  if (dsegs < 1) dsegs = 1;
  a->sd = cdiv(d->D, dsegs);
  a->dsegs = cdiv(d->D, a->sd);
  const long long nt = cols * a->dsegs;
  if (nt > (1 << 24)) return false;
  a->ntile = (int)nt;
  a->wpu = a->nb = 0;                                  // set per launch (xh_wgrad_q4_launch)
  { extern int g_mfma_abl; a->abl = g_mfma_abl; }
  a->full = 0;
  a->bcast = d->bcast ? 1 : 0;
  if (d->bcast && (d->bcast != 4 || f32 || dwm || cin_g != 4 || d->Ca != d->Cin)) return false;
  if (!(a->abl & (8192 | 524288))) xh_wgrad_q5_replan(d, a);   // rows of 64 / 128 voxels: the full-row kernel (conv3d_wgrad_q5.hip)
  if (d->bcast && !a->full) return false;               // (only the full-row kernel reads a broadcast input)
  return true;
}

// launches up to WQ_MULTI planned problems of one storage format and one input-quad count (probs[i].ci4 all equal)
void xh_wgrad_q4_launch(hipStream_t st, int fmt, const WgQ4* probs, int n) {
  if (probs[0].full) { xh_wgrad_q5_launch(st, fmt, probs, n); return; }
  WgQ4Multi m;
  m.n = n;
  m.off[0] = 0;
  const int ci4 = probs[0].ci4;
  // workgroups per launch: all resident at once, dealt to the units in proportion to their tiles
  extern int g_mfma_abl;
  const int budget = ((g_mfma_abl & 8192) ? wq4_waves(ci4) : wq4_resident(fmt, ci4)) * 256;
  double total = 0.0;
  for (int i = 0; i < n; ++i) total += (double)probs[i].nq * probs[i].ntile;
  for (int i = 0; i < n; ++i) {
    m.p[i] = probs[i];
    WgQ4& a = m.p[i];
    int w = (int)(budget * (double)a.ntile / total);
    if (w >= 8) w &= ~7;
    if (w > a.ntile) w = a.ntile;
    if (w < 1) w = 1;
    a.wpu = w;
    a.nb = a.nq * w;
    m.off[i + 1] = m.off[i] + ((a.nb + 7) & ~7);
  }
  const bool ldsx = fmt == 2 || !(g_mfma_abl & 8192);  // ablation bit 8192: the variant in which every wave loads its own x rows
  xh_note_kernel("conv3_wgrad_q4_multi_kernel<%d, %d, %s>", fmt, ci4, ldsx ? "true" : "false");
#define WQL(F, C)                                                                                                        \
  do {                                                                                                                   \
    if (ldsx) hipLaunchKernelGGL((conv3_wgrad_q4_multi_kernel<F, C, true>), dim3(m.off[n]), dim3(256), 0, st, m);         \
    else hipLaunchKernelGGL((conv3_wgrad_q4_multi_kernel<F, C, false>), dim3(m.off[n]), dim3(256), 0, st, m);             \
  } while (0)
  if (fmt == 2) {                                      // fp32 storage: the LDS variant only
    if (ci4 == 1) hipLaunchKernelGGL((conv3_wgrad_q4_multi_kernel<2, 1, true>), dim3(m.off[n]), dim3(256), 0, st, m);
    else if (ci4 == 2) hipLaunchKernelGGL((conv3_wgrad_q4_multi_kernel<2, 2, true>), dim3(m.off[n]), dim3(256), 0, st, m);
    else hipLaunchKernelGGL((conv3_wgrad_q4_multi_kernel<2, 3, true>), dim3(m.off[n]), dim3(256), 0, st, m);
  } else if (fmt) { if (ci4 == 1) WQL(1, 1); else if (ci4 == 2) WQL(1, 2); else WQL(1, 3); }
  else { if (ci4 == 1) WQL(0, 1); else if (ci4 == 2) WQL(0, 2); else WQL(0, 3); }
#undef WQL
}
