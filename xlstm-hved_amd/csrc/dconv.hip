// Discriminator convolutions (RA_HVED.py:204-236, buildingblocks.py:342-358) as implicit GEMMs on the matrix cores.
//
// The discriminator is the one MFMA-bound object of a training step (7 -> 64 -> 128 -> 256 -> 512 -> 1 channels,
// strides 1,2,2,2,1, padding 1; kernel size K = 4 as train.py:146 builds it -- 128 -> 127 -> 63 -> 31 -> 15 -> 14 voxels per axis,
// ~560 GFLOP per forward at 128^3 -- or K = 3, the class default; three passes per step) and it is self-contained, so inside it the
// activations are kept CHANNELS-LAST ([n][d][h][w][C], 16-bit): the K axis of the GEMM (input channels of one tap) is then
// contiguous in memory and a 16 x 32 MFMA operand row is one 64-byte run of one voxel -- no im2col buffer, no transposition.
//
//   forward      Y[m][co] = sum_{tap, ci} X[src(m, tap)][ci] * W[co][ci][tap]          M = voxels, N = Cout, K = K^3 * Cin
//   data grad    dX[m][ci] = sum_{tap, co} dY[src'(m, tap)][co] * W[co][ci][tap]       same kernel, roles swapped; for stride 2
//                the destination voxels are processed in their 8 parity classes so that only the taps that reach a class are
//                walked (K = 3: 27 tap visits in total instead of 8 x 27; K = 4: 8 taps for every class instead of 64)
//   weight grad  dW[tap][co][ci] = sum_m dY[m][co] * X[src(m, tap)][ci]                M = Cout, N = Cin, K = voxels: both operands
//                are needed K(voxel)-major, i.e. transposed -- staged row-major in LDS and read with ds_read_b64_tr_b16
//
// Workgroup = 4 waves, tile 128 x 128 (or 256 x 16 for the 8-channel ends), K step 32, two LDS buffers, register-staged
// loads one step ahead, mfma_f32_16x16x32 with a 4 x 4 accumulator block per wave.
#include "common.h"
#include "../../include/xlstm_hved.h"

typedef short s4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct DTaps { int n; int t[4]; int off[4]; };

// One class of destination voxels.  Forward / stride-1 data gradient: a single class = all of them.  Stride-2 data gradient:
// the 8 parity classes (destination = 2 j + p per axis), each with the taps that reach it (1 or 2 per axis); they travel in ONE
// launch, heaviest first (8 taps .. 1 tap), so that the short classes fill the tail of the long ones instead of running as
// eight under-filled launches one after the other.
struct DClass {
  int Jd, Jh, Jw;          // rows of this class per sample: (jd, jh, jw), destination index = j * omul + p
  int pd, ph, pw;
  int tile0, ntile;        // row tiles (blockIdx.x) of the class: [tile0, tile0 + ntile)
  DTaps td, th, tw;
};

struct DConvK {
  const u16* x; const u16* w; const float* bias; u16* y; double* red;
  const u16* mask;         // optional, [N][Do..][Cn] like y: the result is multiplied by leaky'(mask) = (mask > 0 ? 1 : slope) before
                           // it is rounded, stored and summed -- the LeakyReLU backward of the layer below, fused into this
                           // data gradient's epilogue (the separate pass read and wrote the 64-channel 127^3 tensor once more)
  int N, Di, Hi, Wi, Do, Ho, Wo, Cs, Cn;
  int omul;
  int smul;                // source index = j * smul + off[tap]
  int ncls, xcd;
  DClass c[8];
  int rowmode;             // 1: Cs == 8: one K step = the kw taps x 8 channels of a (kd, kh) row (K = 3: + 8 zero-weight), tw ignored
  int act; float slope;
  int K;                   // kernel size per axis (3 or 4), padding 1
  int wtap_stride;         // elements between consecutive taps in w (= Cn * Kc)
  int Kc;                  // K extent of one step group in w rows (Cs, or 32 in rowmode)
};

// LDS images: rows of 64 bytes (32 x 16-bit), the 16-byte chunk index XORed with (-(row >> 2)) & 3.  A ds_read_b128 is served in four
// groups of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md): a group mixes k-group kg
// on rows 0-3 / 12-15 with kg + 1 on rows 4-11, so the XOR value per 4-row block must send (kg, block 0), (kg, block 3),
// (kg ^ 1, block 1), (kg ^ 1, block 2) to four different chunks: 0, 3, 2, 1 does; the earlier (row >> 2) & 3 = 0, 1, 2, 3 put two
// lanes on every bank (SQ_LDS_BANK_CONFLICT as large as the LDS-active cycles of the 64 -> 128 forward)
__device__ __forceinline__ int sw64(int row, int chunk) { return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4); }
// rows of 64 * KQ bytes: KQ = 2 (128-byte rows): XOR with row & 7 over the 8 chunks
// KQ = 4 (256-byte rows = the whole bank width): XOR with row & 15 over the 16 chunks
template <int KQ> __device__ __forceinline__ int swr(int row, int chunk) {
  return KQ == 1 ? sw64(row, chunk) : KQ == 2 ? row * 128 + ((chunk ^ (row & 7)) << 4) : row * 256 + ((chunk ^ (row & 15)) << 4);
}

// KQ = 32-channel quarters per K step (1: K step 32, 2: K step 64 -- half the barriers, twice the matrix work between them)
// PF = K steps in flight in registers (1: the next step only).  Measured on the narrow 256 x 16 tile (64 -> 8 data gradient
// @128^3, 4 MFMAs per wave per step): PF 1 / 2 / 4 = 753 / 783 / 998 us -- the gather is bound by cache throughput (each dY row
// is fetched 27 times), not by latency, and the extra registers only cost occupancy.  Kept as an experiment switch
// TM = 16-row blocks per wave (4: a wave owns 64 rows; 8: 128 rows = a 256 x 128 workgroup tile: 12 fragment reads per 32 MFMAs
// instead of 16)
template <int FMT, int WGN, int TN, int KQ = 1, int PF = 1, int TM = 4>
__global__ __launch_bounds__(256, 2) void dconv_cl_kernel(const DConvK a) {
  constexpr int WGM = 4 / WGN;
  constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
  constexpr int RB = 64 * KQ, CPRW = 4 * KQ;             // bytes / 16-byte chunks per LDS row
  constexpr int AB = BM * RB, BB = BN * RB;              // bytes per A / B buffer
  constexpr int NA = BM * CPRW / 256, NB = (BN * CPRW + 255) / 256;
  constexpr int SM = 2 * (AB + BB);
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  // per-wave-row column sums of the epilogue; where the two staging buffers already fill the 64 KB of static LDS (KQ = 4) they live in
  // the first buffer, which nobody reads after the last step's barrier
  constexpr bool STAT_IN_SMEM = SM + (int)sizeof(double) * WGM * BN * 2 > 65536;
  __shared__ double s_stat_own[STAT_IN_SMEM ? 1 : WGM * BN * 2];
  double* s_stat = STAT_IN_SMEM ? reinterpret_cast<double*>(smem) : s_stat_own;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WGN, wn = wv % WGN;
  const int r16 = lane & 15, kg = lane >> 4;
  // (a.xcd & 4) column tile = XCD (forward, 8 % column tiles == 0; 1-D grid): the weights of ONE column tile (BN x K^3 Cs x 2 B, 1 - 2 MB)
  // then stay in that XCD's 4 MB L2 for the whole launch and every row tile streams its x rows once -- no lockstep between
  // workgroups needed.  With the (x, y, z) grid an XCD saw every column tile: 256 -> 512 @15^3 fetched 800 MB past L2 for 32 MB of
  // operands (TCC miss 46 %), the launch ran at the fabric's rate.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd & 4) {
    const int ny = (a.Cn + BN - 1) / BN, r = 8 / ny;       // r XCDs share a column tile
    const int k = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int t = a.c[0].ntile;
    const int xx = sl * r + k / ny;                       // row tile x sample
    by = k % ny;
    if (xx >= t * a.N) return;
    bx = xx % t; bz = xx / t;
  }
  const int n = bz, cn0 = by * BN;
  int ci = 0;
  for (int k = 1; k < a.ncls; ++k)
    if (bx >= a.c[k].tile0) ci = k;
  const DClass& cl = a.c[ci];
  // workgroups go round the 8 XCDs in blockIdx order: hand every XCD a contiguous run of the class's row tiles, so that the
  // rows its taps share (h +- 1, d +- 1) are re-read from its own L2 instead of the fabric
  int mt = bx - cl.tile0;
  if ((a.xcd & 1) && !(cl.tile0 & 7) && !(cl.ntile & 7)) mt = xcd_swizzle(mt, cl.ntile);
  const int R = cl.Jd * cl.Jh * cl.Jw;

  // ---- staging plan: A rows (source voxel base coordinates), B columns ----
  int a_row[NA], a_sd[NA], a_sh[NA], a_sw[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = tid / CPRW + i * (256 / CPRW);
    const int m = mt * BM + row;
    a_row[i] = row;
    a_ok[i] = m < R;
    const int mm = a_ok[i] ? m : 0;
    const int jw = mm % cl.Jw, t2 = mm / cl.Jw;
    const int jh = t2 % cl.Jh, jd = t2 / cl.Jh;
    a_sd[i] = jd * a.smul; a_sh[i] = jh * a.smul; a_sw[i] = jw * a.smul;
  }
  const int ch = tid % CPRW;
  const long long xs_n = (long long)n * a.Di * a.Hi * a.Wi;
  const int nsteps_c = a.rowmode ? 1 : a.Cs / (32 * KQ);
  const int ntw = a.rowmode ? 1 : cl.tw.n;
  const int nsteps = cl.td.n * cl.th.n * ntw * nsteps_c;

  // every load of a step is issued back to back from a clamped (always valid) address; what was out of range is zeroed when
  // the step is written to LDS (mask bits travel with it): a guarded load (`if (in range) load`) costs a branch and an
  // s_waitcnt each and serialises the step's ~8 loads.
  // Round 3: the step decode and the addresses used to be recomputed per step -- four scalar divisions, and per staged row three
  // coordinate sums, six clamps, three range checks and a 64-bit address (SQ counters: 9 vector + 6 scalar instructions per
  // MFMA, the kernel was bound by their issue, not by LDS or the matrix cores).  Now a row keeps ONE 32-bit element offset and a
  // 12-bit mask of the taps that stay inside the volume; the step (tap, channel quarter) is walked incrementally in scalar
  // registers and contributes one scalar offset: per row and step an add, a clamp and a mask test.
  int a_base[NA];
  unsigned a_msk[NA];
  const int x_lim = a.Di * a.Hi * a.Wi * a.Cs - 8;        // last 16-byte run of a sample (host: fits 31 bits)
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    a_base[i] = ((a_sd[i] * a.Hi + a_sh[i]) * a.Wi + a_sw[i]) * a.Cs + ch * 8;
    unsigned m = 0;
    for (int k = 0; k < cl.td.n; ++k) m |= (a_ok[i] && (unsigned)(a_sd[i] + cl.td.off[k]) < (unsigned)a.Di) ? 1u << k : 0u;
    for (int k = 0; k < cl.th.n; ++k) m |= (a_ok[i] && (unsigned)(a_sh[i] + cl.th.off[k]) < (unsigned)a.Hi) ? 16u << k : 0u;
    for (int k = 0; k < cl.tw.n; ++k) m |= (a_ok[i] && (unsigned)(a_sw[i] + cl.tw.off[k]) < (unsigned)a.Wi) ? 256u << k : 0u;
    a_msk[i] = m;
  }
  int b_base[NB];
  unsigned b_okm = 0;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int col = tid / CPRW + i * (256 / CPRW);
    const bool ok = col < BN && cn0 + col < a.Cn;
    b_okm |= ok ? 1u << (NA + i) : 0u;
    b_base[i] = min(cn0 + col, a.Cn - 1) * a.Kc + ch * 8;
  }
  const u16* x_n = a.x + xs_n * a.Cs;
  // Tap table (round 5): what a K step needs of its tap -- the source-row offset, the weight offset, the mask-bit selector -- sits in
  // LDS, written once.  The scalar walk used to fetch the tap's t[] / off[] entries from the kernel arguments inside every step:
  // 6-7 s_load_dword + s_waitcnt lgkmcnt(0), which also drains the step's LDS reads (SQ counters of 64 -> 128 @63^3: 63 scalar
  // instructions per step, 37 % of the wave time parked on waits).
  constexpr bool TAPTAB = true;                           // (false: the scalar walk, kept for A/B)
  __shared__ __attribute__((aligned(16))) int s_tap[64 * 4];
  if (TAPTAB && !a.rowmode) {
    const int ntap = cl.td.n * cl.th.n * cl.tw.n;
    for (int ti = tid; ti < ntap; ti += 256) {
      const int iw = ti % cl.tw.n, t3 = ti / cl.tw.n;
      const int ih = t3 % cl.th.n, id = t3 / cl.th.n;
      const int tapk = (cl.td.t[id] * a.K + cl.th.t[ih]) * a.K + cl.tw.t[iw];
      s_tap[ti * 4] = ((cl.td.off[id] * a.Hi + cl.th.off[ih]) * a.Wi + cl.tw.off[iw]) * a.Cs;
      s_tap[ti * 4 + 1] = tapk * a.wtap_stride;
      s_tap[ti * 4 + 2] = (int)((1u << id) | (16u << ih) | (256u << iw));
    }
    __syncthreads();
  }
  // x through a buffer descriptor of the sample: a row whose tap falls outside the volume loads from an out-of-range offset and
  // gets zeros back -- no clamp, no select when the step is written to LDS
  __amdgpu_buffer_rsrc_t x_rs;
  {
    const unsigned long long v = (unsigned long long)x_n;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (x_lim + 8) * 2, 0x00020000);
  }
  // the step walk (scalar): channel quarter fastest, then the w, h, d taps of the class; it stops at the last step
  int it_cs = 0, it_w = 0, it_h = 0, it_d = 0, it_s = 0;
  auto load_step = [&](int s, uint4 (&qa)[NA], uint4 (&qb)[NB], unsigned& okm) __attribute__((always_inline)) {
    if (a.rowmode) {
      const int cs = s % nsteps_c; int t = s / nsteps_c;
      const int iw = t % ntw; t /= ntw;
      const int ih = t % cl.th.n, id = t / cl.th.n;
      const int tap = cl.td.t[id] * a.K + cl.th.t[ih];
      unsigned m = 0;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int d = a_sd[i] + cl.td.off[id], h = a_sh[i] + cl.th.off[ih];
        const int w = a_sw[i] + (ch - 1);
        const bool ok = a_ok[i] && (unsigned)d < (unsigned)a.Di && (unsigned)h < (unsigned)a.Hi && (unsigned)w < (unsigned)a.Wi;
        m |= ok ? 1u << i : 0u;
        const int dc = min(max(d, 0), a.Di - 1), hc = min(max(h, 0), a.Hi - 1), wc = min(max(w, 0), a.Wi - 1);
        const u16* p = a.x + ((xs_n + ((long long)dc * a.Hi + hc) * a.Wi + wc) * a.Cs);
        qa[i] = *reinterpret_cast<const uint4*>(p);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int col = tid / CPRW + i * (256 / CPRW);
        const bool ok = col < BN && cn0 + col < a.Cn;
        m |= ok ? 1u << (NA + i) : 0u;
        const int cc = min(cn0 + col, a.Cn - 1);
        qb[i] = *reinterpret_cast<const uint4*>(a.w + (long long)tap * a.wtap_stride + (long long)cc * a.Kc + cs * 32 * KQ + ch * 8);
      }
      okm = m;
      (void)iw;
      return;
    }
    // bring the walk to step s (callers ask for consecutive steps, the last one possibly several times)
    int toff, woff;
    unsigned sel;
    if constexpr (TAPTAB) {
      while (it_s < s) {
        ++it_s;
        if (++it_cs == nsteps_c) { it_cs = 0; ++it_w; }   // it_w: the tap's index in s_tap
      }
      const int4 e = *reinterpret_cast<const int4*>(&s_tap[it_w * 4]);
      toff = e.x + it_cs * 32 * KQ;
      woff = e.y + it_cs * 32 * KQ;
      sel = (unsigned)e.z;
    } else {
      while (it_s < s) {
        ++it_s;
        if (++it_cs == nsteps_c) { it_cs = 0; if (++it_w == cl.tw.n) { it_w = 0; if (++it_h == cl.th.n) { it_h = 0; ++it_d; } } }
      }
      const int tap = (cl.td.t[it_d] * a.K + cl.th.t[it_h]) * a.K + cl.tw.t[it_w];
      toff = ((cl.td.off[it_d] * a.Hi + cl.th.off[it_h]) * a.Wi + cl.tw.off[it_w]) * a.Cs + it_cs * 32 * KQ;
      woff = tap * a.wtap_stride + it_cs * 32 * KQ;
      sel = (1u << it_d) | (16u << it_h) | (256u << it_w);
    }
    unsigned m = b_okm | ((1u << NA) - 1u);            // x rows need no mask: invalid ones arrive as zeros
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const unsigned off = ((a_msk[i] & sel) == sel) ? (unsigned)(a_base[i] + toff) * 2u : 0xFFFFFFF0u;
      qa[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) qb[i] = *reinterpret_cast<const uint4*>(a.w + (unsigned)(b_base[i] + woff));
    okm = m;
  };
  auto store_step = [&](int buf, const uint4 (&qa)[NA], const uint4 (&qb)[NB], unsigned okm) __attribute__((always_inline)) {
    unsigned char* As = smem + buf * (AB + BB);
    unsigned char* Bs = As + AB;
    const uint4 z = make_uint4(0, 0, 0, 0);
    if (a.rowmode) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        uint4 v = qa[i];                                // (not `ok ? qa[i] : z`: a conditional of two lvalues selects an ADDRESS and
        if (!((okm >> i) & 1)) v = z;                   //  drags the staging registers into scratch memory)
        *reinterpret_cast<uint4*>(As + swr<KQ>(a_row[i], ch)) = v;
      }
    } else {                                            // rows outside the volume came back as zeros from the buffer load
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<uint4*>(As + swr<KQ>(a_row[i], ch)) = qa[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int col = tid / CPRW + i * (256 / CPRW);
      uint4 v = qb[i];
      if (!((okm >> (NA + i)) & 1)) v = z;
      if (col < BN) *reinterpret_cast<uint4*>(Bs + swr<KQ>(col, ch)) = v;
    }
  };
  auto mfma_step = [&](int buf, f32x4_t (&acc)[TM][TN]) __attribute__((always_inline)) {
    const unsigned char* As = smem + buf * (AB + BB);
    const unsigned char* Bs = As + AB;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      h16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const h16x8*>(As + swr<KQ>((wm * TM + i) * 16 + r16, kq * 4 + kg));
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const h16x8*>(Bs + swr<KQ>((wn * TN + j) * 16 + r16, kq * 4 + kg));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma16x16x32<FMT>(bf[j], af[i], acc[i][j]);      // D^T: channels on the rows (epilogue)
    }
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if constexpr (PF == 1) {
    uint4 ra[NA], rb[NB];
    unsigned okm;
    load_step(0, ra, rb, okm);
    store_step(0, ra, rb, okm);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int buf = s & 1;
      if (s + 1 < nsteps) load_step(s + 1, ra, rb, okm);
      mfma_step(buf, acc);
      if (s + 1 < nsteps) store_step(buf ^ 1, ra, rb, okm);
      __syncthreads();
    }
  } else {
    // slot u holds step k with k % PF == u; a step's slot is refilled (step + PF) as soon as the step sits in LDS.  The main loop
    // has no guards (loads past the end re-read the last step, the surplus LDS write goes to the buffer nobody reads again);
    // the nsteps % PF steps left over are finished from the slots that already hold them
    static_assert(PF % 2 == 0, "LDS buffer parity rides on the slot index");
    uint4 ra[PF][NA], rb[PF][NB];
    unsigned okm[PF];
    const int last = nsteps - 1;
#pragma unroll
    for (int u = 0; u < PF; ++u) load_step(min(u, last), ra[u], rb[u], okm[u]);
    store_step(0, ra[0], rb[0], okm[0]);
    __syncthreads();
    const int nmain = nsteps / PF * PF;
    for (int s0 = 0; s0 < nmain; s0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        load_step(min(s0 + u + PF, last), ra[u], rb[u], okm[u]);
        mfma_step(u & 1, acc);
        store_step((u + 1) & 1, ra[(u + 1) % PF], rb[(u + 1) % PF], okm[(u + 1) % PF]);
        __syncthreads();
      }
    }
    const int rem = nsteps - nmain;
#pragma unroll
    for (int u = 0; u < PF - 1; ++u)
      if (u < rem) {
        mfma_step(u & 1, acc);
        if (u + 1 < rem) store_step((u + 1) & 1, ra[u + 1], rb[u + 1], okm[u + 1]);
        __syncthreads();
      }
  }

  // ---- epilogue: bias, activation, rounding, column sums (InstanceNorm statistics) ----
  // The MFMAs ran with the operands swapped (weights on the M side), so a lane holds, for voxel block i and channel block j,
  // voxel r16 and the four CONSECUTIVE channels 4 kg + r: 8 contiguous bytes of the channels-last output, stored straight
  // from the registers (the earlier transpose of the tile through LDS -- 64 two-byte LDS writes per lane, two barriers -- cost
  // 113 of the 290 us of the first conv's forward and 73 of the 349 us of the 64 <- 128 data gradient).
  if (a.xcd & 2) {                                        // ablation: no epilogue (keeps the accumulators alive)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) a.y[0] = 1;
    return;
  }
  const long long ys_n = (long long)n * a.Do * a.Ho * a.Wo;
  const bool vec_ok = (a.Cn & 3) == 0;
  float b4[TN][4], s0[TN][4], s1[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cn = cn0 + (wn * TN + j) * 16 + 4 * kg + r;
      b4[j][r] = (a.bias && cn < a.Cn) ? a.bias[cn] : 0.f;
      s0[j][r] = 0.f; s1[j][r] = 0.f;
    }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mt * BM + (wm * TM + i) * 16 + r16;
    const bool ok = m < R;
    const int mm = ok ? m : 0;
    const int jw = mm % cl.Jw, t2 = mm / cl.Jw;
    const int jh = t2 % cl.Jh, jd = t2 / cl.Jh;
    const long long vox = ys_n + ((long long)(jd * a.omul + cl.pd) * a.Ho + (jh * a.omul + cl.ph)) * a.Wo + (jw * a.omul + cl.pw);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cn = cn0 + (wn * TN + j) * 16 + 4 * kg;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] + b4[j][r];
        if (a.act == XH_ACT_LRELU) v[r] = v[r] > 0.f ? v[r] : v[r] * a.slope;
      }
      if (a.mask && ok) {
        if (vec_ok && cn + 4 <= a.Cn) {
          const uint2 mm = *reinterpret_cast<const uint2*>(a.mask + vox * a.Cn + cn);
          if (!(cvt_lo<FMT>(mm.x) > 0.f)) v[0] *= a.slope;
          if (!(cvt_hi<FMT>(mm.x) > 0.f)) v[1] *= a.slope;
          if (!(cvt_lo<FMT>(mm.y) > 0.f)) v[2] *= a.slope;
          if (!(cvt_hi<FMT>(mm.y) > 0.f)) v[3] *= a.slope;
        } else {
          for (int e = 0; e < 4; ++e)
            if (cn + e < a.Cn && !(cvt_in<FMT>(a.mask[vox * a.Cn + cn + e]) > 0.f)) v[e] *= a.slope;
        }
      }
      const unsigned q0 = cvt_pack<FMT>(v[0], v[1]), q1 = cvt_pack<FMT>(v[2], v[3]);
      if (a.red && ok) {                                  // sums of the ROUNDED values: what the next layer normalises
        const float w0 = cvt_lo<FMT>(q0), w1 = cvt_hi<FMT>(q0), w2 = cvt_lo<FMT>(q1), w3 = cvt_hi<FMT>(q1);
        s0[j][0] += w0; s0[j][1] += w1; s0[j][2] += w2; s0[j][3] += w3;
        s1[j][0] += w0 * w0; s1[j][1] += w1 * w1; s1[j][2] += w2 * w2; s1[j][3] += w3 * w3;
      }
      if (ok) {
        if (vec_ok && cn + 4 <= a.Cn) {
          *reinterpret_cast<uint2*>(a.y + vox * a.Cn + cn) = make_uint2(q0, q1);
        } else {
          const u16 qs[4] = {(u16)(q0 & 0xffff), (u16)(q0 >> 16), (u16)(q1 & 0xffff), (u16)(q1 >> 16)};
          for (int e = 0; e < 4; ++e)
            if (cn + e < a.Cn) a.y[vox * a.Cn + cn + e] = qs[e];
        }
      }
    }
  }
  if (a.red) {
    // a lane's partial covers 4 voxels; the 16 lanes of a DPP row (same kg) hold the other voxels of the wave's 64: fp32 over
    // those 64 values (squares of 16-bit values are exact in fp32), fp64 from there on (across waves, workgroups, launches)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t0 = row16_sum(s0[j][r]), t1 = row16_sum(s1[j][r]);
        if (r16 == 0) {
          const int col = (wn * TN + j) * 16 + 4 * kg + r;
          s_stat[(wm * BN + col) * 2] = (double)t0;
          s_stat[(wm * BN + col) * 2 + 1] = (double)t1;
        }
      }
    __syncthreads();
    for (int i = tid; i < BN * 2; i += 256) {             // one fp64 atomic per (column, moment) per workgroup
      const int col = i >> 1;
      if (cn0 + col < a.Cn) {
        double t = 0.0;
#pragma unroll
        for (int m2 = 0; m2 < WGM; ++m2) t += s_stat[(m2 * BN + col) * 2 + (i & 1)];
        atomicAdd(&a.red[((long long)n * a.Cn + cn0 + col) * 2 + (i & 1)], t);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Stride-2 data gradient with the SOURCE BLOCK staged once per channel slice (round 5).
//
// dconv_cl_kernel gathers, for every tap and 32-channel K step, the A rows of its 256 destination voxels from global memory: the
// data gradient 64 <- 128 @63^3 (the dominant launch of a training step: 2 x 1.08 ms at 0.145 of the dense MFMA peak) moves
// M x K x 2 B = 2.05 M rows x 1 024 x 2 B = 4.2 GB through L1 / L2 for a 64 MB dY -- every dY row is fetched once per tap and per
// destination parity class, 64 times in all -- plus 1 GB of weight rows, at 20 KB per 1 MFLOP: the launch runs at the rate the
// cache hierarchy delivers those re-reads (4.7 TB/s), not at the matrix cores' rate.
//
// A parity class of a stride-2 data gradient reaches only 2 (K = 4; 1 or 2 for K = 3) source offsets per axis, so a block of
// 8 x 8 x 8 destination voxels of the class needs a source block of at most 9 x 9 x 9 voxels.  Here a workgroup of 8 waves owns such
// a block (M = 512 rows) x 64 destination channels; per 32-channel slice of dY it stages
//   * the source block ONCE: <= 729 rows x 64 B = 46 KB (out-of-volume rows arrive as zeros from the buffer load), and
//   * the slice's weight rows of all the class's taps: <= 8 x 64 x 64 B = 32 KB,
// and then runs the taps from LDS: the A fragment of destination row m and tap t is the staged row m + offset(t) -- an address.
// Per tile that is 4 x 78 KB = 0.3 MB of loads for 67 MFLOP (215 flop / B instead of 52); LDS fragment reads as before (4 + 4 per
// 16 MFMAs).  One LDS buffer of 79 KB, two workgroups per CU: one stages while the other multiplies.
// Wave w owns destination plane w of the block (4 M tiles of 2 rows x 8 voxels) and all 4 N tiles; epilogue as dconv_cl_kernel
// (bias, LeakyReLU mask of the layer below, rounding, channel sums).
// K4: k = 4, stride 2 -- every class has 2 x 2 x 2 taps, so the block extents (9 x 9 x 9 source rows) are compile-time constants and
// the staging plan's row decomposition costs multiplies instead of runtime divisions (64 <- 128 @127^3, two samples: 891 -> 865 us).
// Measured and not kept (round 5): a PERSISTENT form -- one workgroup per CU, both LDS images twice, the next slice (of the next
// block) in flight under the taps, channel sums carried in registers -- ran 1033 us against 865: with everything but its skeleton
// switched off (no taps, no loads, no stores) it still took 425 us (plan, LDS stores, four barriers per block at 2 waves per SIMD);
// the counters of this kernel show 63 % of its wave time parked on waits and LDS bank conflicts in the A-fragment reads (the 8 + 8
// rows of a fragment start at multiples of 9, not 4, so the chunk swizzle of sw64 does not separate them)
template <int FMT, bool K4 = false>
__global__ __launch_bounds__(512, 4) void dconv_dgrad_halo_kernel(const DConvK a) {
  constexpr int BT = 8, BN = 64, TM = 4, TN = 4;
  constexpr int A_BYTES = 729 * 64, B_BYTES = 8 * BN * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Bs = smem + A_BYTES;
  double* s_stat = reinterpret_cast<double*>(smem);      // epilogue: [8 waves][BN][2], after the last matrix phase
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int n = blockIdx.z, cn0 = blockIdx.y * BN;
  int ci = 0;
  for (int k = 1; k < a.ncls; ++k)
    if ((int)blockIdx.x >= a.c[k].tile0) ci = k;
  const DClass& cl = a.c[ci];
  int mt = blockIdx.x - cl.tile0;
  if ((a.xcd & 1) && !(cl.tile0 & 7) && !(cl.ntile & 7)) mt = xcd_swizzle(mt, cl.ntile);
  const int tnw = (cl.Jw + BT - 1) / BT, tnh = (cl.Jh + BT - 1) / BT;
  const int tjw = mt % tnw, t1 = mt / tnw;
  const int tjh = t1 % tnh, tjd = t1 / tnh;
  const int j0d = tjd * BT, j0h = tjh * BT, j0w = tjw * BT;
  const int nd = K4 ? 2 : cl.td.n, nh = K4 ? 2 : cl.th.n, nw = K4 ? 2 : cl.tw.n, ntap = nd * nh * nw;
  int mind = cl.td.off[0], minh = cl.th.off[0], minw = cl.tw.off[0];
  for (int k = 1; k < nd; ++k) mind = min(mind, cl.td.off[k]);
  for (int k = 1; k < nh; ++k) minh = min(minh, cl.th.off[k]);
  for (int k = 1; k < nw; ++k) minw = min(minw, cl.tw.off[k]);
  const int HD = BT + nd - 1, HH = BT + nh - 1, HW = BT + nw - 1;
  const int nrows = HD * HH * HW;
  const long long xs_n = (long long)n * a.Di * a.Hi * a.Wi;
  const u16* x_n = a.x + xs_n * a.Cs;
  const int x_lim = a.Di * a.Hi * a.Wi * a.Cs - 8;
  __amdgpu_buffer_rsrc_t x_rs;
  {
    const unsigned long long v = (unsigned long long)x_n;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (x_lim + 8) * 2, 0x00020000);
  }
  // ---- staging plan: A items (source row, 16-byte chunk), B items (tap, destination channel, chunk) ----
  constexpr int NA = (729 * 4 + 511) / 512, NB = (8 * BN * 4) / 512;     // 6, 4
  unsigned a_off[NA];                                    // byte offset of the row's chunk in the sample (slice 0), or out of range
  int a_lds[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int it = tid + 512 * i;
    const int row = it >> 2, ch = it & 3;
    const bool live = row < nrows;
    const int rr = live ? row : 0;
    const int hw_ = rr % HW, t2 = rr / HW;
    const int hh_ = t2 % HH, hd_ = t2 / HH;
    const int sd = (j0d + hd_) * a.smul + mind, sh = (j0h + hh_) * a.smul + minh, sw = (j0w + hw_) * a.smul + minw;
    const bool inb = live && (unsigned)sd < (unsigned)a.Di && (unsigned)sh < (unsigned)a.Hi && (unsigned)sw < (unsigned)a.Wi;
    a_off[i] = inb ? (unsigned)((((sd * a.Hi + sh) * a.Wi + sw) * a.Cs + ch * 8) * 2) : 0xFFFFFFF0u;
    a_lds[i] = live ? row * 64 + ((ch ^ ((hh_ & 1) << 1)) << 4) : -1;       // (the source image's own swizzle: see the fragment rows below)
  }
  int b_src[NB], b_lds[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int it = tid + 512 * i;
    const int ch = it & 3, col = (it >> 2) & (BN - 1), t = it >> 8;
    const bool live = t < ntap;
    const int tt = live ? t : 0;
    const int iw = tt % nw, t3 = tt / nw;
    const int ih = t3 % nh, id = t3 / nh;
    const int tapk = (cl.td.t[id] * a.K + cl.th.t[ih]) * a.K + cl.tw.t[iw];
    b_src[i] = live ? tapk * a.wtap_stride + min(cn0 + col, a.Cn - 1) * a.Kc + ch * 8 : -1;
    b_lds[i] = t * (BN * 64) + sw64(col, ch);
  }
  // ---- fragment rows: wave w = destination plane w of the block; M tile i = the 4 x 4 voxels (h = 4 (i >> 1) + (r16 >> 2), w = 4 (i & 1)
  // + (r16 & 3)).  Four runs of four consecutive source rows, whatever the tap offset: consecutive rows differ in row & 3 (their 64-byte
  // quarter of the banks), and the chunk index is XORed with 2 (h & 1) of the SOURCE row's h coordinate, which sends the four runs
  // of a ds_read_b128 lane group -- k-group kg on runs 0 and 3, kg ^ 1 on runs 1 and 2 -- to four different chunks for every
  // alignment (2 x 8 tiles under sw64 had SQ_LDS_BANK_CONFLICT at twice the LDS-active cycles: the runs started at multiples of 9)
  int a_row[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_row[i] = (wv * HH + 4 * (i >> 1) + (r16 >> 2)) * HW + 4 * (i & 1) + (r16 & 3);
  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int rd0 = (cl.td.off[0] - mind) * HH * HW, rd1 = (cl.td.off[nd > 1 ? 1 : 0] - mind) * HH * HW;
  const int dh0 = cl.th.off[0] - minh, dh1 = cl.th.off[nh > 1 ? 1 : 0] - minh;
  const int rh0 = dh0 * HW, rh1 = dh1 * HW;
  const int rw0 = cl.tw.off[0] - minw, rw1 = cl.tw.off[nw > 1 ? 1 : 0] - minw;
  const int nslice = a.Cs / 32;
  for (int sl = 0; sl < nslice; ++sl) {
    // Two groups, each issued whole and then written: with all ten pieces requested at once hipcc (128 registers) kept the four weight
    // pieces in SCRATCH -- load, wait, spill, four times in a row -- so a slice paid five exposed memory latencies; now it pays two.
    {
      uint4 qa[NA];
#pragma unroll
      for (int i = 0; i < NA; ++i)
        qa[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)(a_off[i] == 0xFFFFFFF0u ? a_off[i] : a_off[i] + sl * 64), 0, 0));
      if (sl > 0) __syncthreads();                       // every wave is done with the previous slice's image
#pragma unroll
      for (int i = 0; i < NA; ++i)
        if (a_lds[i] >= 0) *reinterpret_cast<uint4*>(As + a_lds[i]) = qa[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      uint4 qb[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) qb[i] = *reinterpret_cast<const uint4*>(a.w + (unsigned)(max(b_src[i], 0) + sl * 32));
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (b_src[i] >= 0) *reinterpret_cast<uint4*>(Bs + b_lds[i]) = qb[i];
    }
    __syncthreads();
#pragma unroll 1
    for (int id = 0; id < nd; ++id)
#pragma unroll 1
      for (int ih = 0; ih < nh; ++ih)
#pragma unroll 1
        for (int iw = 0; iw < nw; ++iw) {
          const int t = (id * nh + ih) * nw + iw;
          // (K4: the two offsets per axis sit in scalar registers -- indexing the class's tables by the loop counters fetched them from
          // the kernel arguments inside every tap, with an s_waitcnt that also drained the fragment reads)
          const int roff = K4 ? (id ? rd1 : rd0) + (ih ? rh1 : rh0) + (iw ? rw1 : rw0)
                              : ((cl.td.off[id] - mind) * HH + (cl.th.off[ih] - minh)) * HW + (cl.tw.off[iw] - minw);
          h16x8 af[TM];
          const int dh = K4 ? (ih ? dh1 : dh0) : cl.th.off[ih] - minh;            // h offset of the tap inside the source block
          const int hsw = ((((r16 >> 2) + dh) & 1) << 1) ^ kg;                     // chunk of this lane's k-group in its source row
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const h16x8*>(As + (a_row[i] + roff) * 64 + (hsw << 4));
          const unsigned char* bt = Bs + t * (BN * 64);
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const h16x8 bf = *reinterpret_cast<const h16x8*>(bt + sw64(j * 16 + r16, kg));
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = mfma16x16x32<FMT>(bf, af[i], acc[i][j]);      // D^T: channels on the rows
          }
        }
  }

  // ---- epilogue (as dconv_cl_kernel): bias, mask of the layer below, rounding, channel sums ----
  const long long ys_n = (long long)n * a.Do * a.Ho * a.Wo;
  float b4[TN][4], s0[TN][4], s1[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cn = cn0 + j * 16 + 4 * kg + r;
      b4[j][r] = (a.bias && cn < a.Cn) ? a.bias[cn] : 0.f;
      s0[j][r] = 0.f; s1[j][r] = 0.f;
    }
  const int jd = j0d + wv;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int jh = j0h + 4 * (i >> 1) + (r16 >> 2), jw = j0w + 4 * (i & 1) + (r16 & 3);
    const bool ok = jd < cl.Jd && jh < cl.Jh && jw < cl.Jw;
    const long long vox = ok ? ys_n + ((long long)(jd * a.omul + cl.pd) * a.Ho + (jh * a.omul + cl.ph)) * a.Wo + (jw * a.omul + cl.pw) : ys_n;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cn = cn0 + j * 16 + 4 * kg;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] + b4[j][r];
        if (a.act == XH_ACT_LRELU) v[r] = v[r] > 0.f ? v[r] : v[r] * a.slope;
      }
      if (a.mask && ok) {
        const uint2 mm = *reinterpret_cast<const uint2*>(a.mask + vox * a.Cn + cn);
        if (!(cvt_lo<FMT>(mm.x) > 0.f)) v[0] *= a.slope;
        if (!(cvt_hi<FMT>(mm.x) > 0.f)) v[1] *= a.slope;
        if (!(cvt_lo<FMT>(mm.y) > 0.f)) v[2] *= a.slope;
        if (!(cvt_hi<FMT>(mm.y) > 0.f)) v[3] *= a.slope;
      }
      const unsigned q0 = cvt_pack<FMT>(v[0], v[1]), q1 = cvt_pack<FMT>(v[2], v[3]);
      if (a.red && ok) {
        const float w0 = cvt_lo<FMT>(q0), w1 = cvt_hi<FMT>(q0), w2 = cvt_lo<FMT>(q1), w3 = cvt_hi<FMT>(q1);
        s0[j][0] += w0; s0[j][1] += w1; s0[j][2] += w2; s0[j][3] += w3;
        s1[j][0] += w0 * w0; s1[j][1] += w1 * w1; s1[j][2] += w2 * w2; s1[j][3] += w3 * w3;
      }
      if (ok) *reinterpret_cast<uint2*>(a.y + vox * a.Cn + cn) = make_uint2(q0, q1);
    }
  }
  if (a.red) {
    __syncthreads();                                     // the LDS images are no longer read
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t0 = row16_sum(s0[j][r]), t1 = row16_sum(s1[j][r]);
        if (r16 == 0) {
          const int col = j * 16 + 4 * kg + r;
          s_stat[(wv * BN + col) * 2] = (double)t0;
          s_stat[(wv * BN + col) * 2 + 1] = (double)t1;
        }
      }
    __syncthreads();
    for (int i = tid; i < BN * 2; i += 512) {
      const int col = i >> 1;
      double t = 0.0;
#pragma unroll
      for (int m2 = 0; m2 < 8; ++m2) t += s_stat[(m2 * BN + col) * 2 + (i & 1)];
      atomicAdd(&a.red[((long long)n * a.Cn + cn0 + col) * 2 + (i & 1)], t);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: dw[tap][cn][cs] += sum_m dY[m][cn] * X[src(m, tap)][cs]   (fp32, packed order; split over voxel ranges)
struct DWgK {
  const u16* x; const u16* dy; float* dw;
  int N, Di, Hi, Wi, Do, Ho, Wo, Cs, Cn, stride;
  int msplit;              // voxel-range splits (grid z = ntz * msplit)
  long long M;             // N * Do * Ho * Wo
  int K, ntap, ntz;        // kernel size per axis, K^3, tap slots on grid z (ntap, or ceil(ntap / 2) for tap pairs)
};
// LDS image of a [32 voxel][128 channel] tile for transposed reads: 256-byte rows, the 16-byte chunk index XORed with
// ((row & 3) << 2) | ((row >> 2) & 3)  (conflict-free ds_read_b64_tr_b16 of 4-row blocks)
__device__ __forceinline__ int sw256(int row, int chunk) { return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

// PAIR (Cs == 64): a 128-wide X tile would be half padding, so two consecutive taps share it -- columns 0..63 = tap 2z,
// 64..127 = tap 2z + 1 (K = 3: 14 pairs, the odd one out has an empty second half: dY is streamed 14 times instead of 27;
// K = 4: 32 pairs)
template <int FMT, int WGN, int TMW, int TNW, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void dwgrad_cl_kernel(const DWgK a) {
  // waves: WGM x WGN over (cn, cs); a wave owns TMW x TNW tiles of 16 x 16
  constexpr int WGM = 4 / WGN;
  constexpr int BMc = WGM * TMW * 16, BNc = WGN * TNW * 16;   // channels of dY / of X per workgroup (<= 128 each)
  constexpr int TB = 32 * 256;                              // bytes per staged tile (32 voxels x 128 channels)
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WGN, wn = wv % WGN;
  const int r16 = lane & 15, kg = lane >> 4, q = r16 >> 2, p = r16 & 3;
  const int cs0 = blockIdx.x * BNc, cn0 = blockIdx.y * BMc;
  const int tz = blockIdx.z % a.ntz, split = blockIdx.z / a.ntz;
  // the X chunk a thread stages is the same in every step (chunk = tid & 15): its tap is a per-thread constant
  const int tap = PAIR ? 2 * tz + ((tid & 15) >> 3) : tz;
  const bool tap_ok = tap < a.ntap;
  const int kd = tap / (a.K * a.K), kh = (tap / a.K) % a.K, kw = tap % a.K;
  const long long per = ((a.M + a.msplit - 1) / a.msplit + 31) / 32 * 32;
  const long long m_begin = (long long)split * per, m_end = m_begin + per < a.M ? m_begin + per : a.M;
  // staging: a tile is 32 rows x 16 chunks = 512 chunks -> 2 per thread per operand.  A thread's two rows move on by 32 voxels
  // per step: their (n, od, oh, ow) are carried along instead of being divided out of m every step (three runtime divisions
  // per load were more VALU work than the step's 16 MFMAs), and every load is issued from a clamped address and masked
  // afterwards (guards around loads serialise them)
  uint4 ry[2], rx[2];
  unsigned okm = 0;                                       // bits 0, 1: dY pieces to keep, bits 2, 3: X pieces
  const int chn = tid & 15;
  const bool y_ok = cn0 + chn * 8 < a.Cn && chn * 8 < BMc;
  const bool x_ok = PAIR ? tap_ok : (cs0 + chn * 8 < a.Cs && chn * 8 < BNc);
  const int y_off = y_ok ? cn0 + chn * 8 : 0;
  const int x_off = x_ok ? (PAIR ? (chn & 7) * 8 : cs0 + chn * 8) : 0;
  int ow[2], oh[2], od[2], nn[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long long t = m_begin + (tid >> 4) + i * 16;
    ow[i] = (int)(t % a.Wo); t /= a.Wo;
    oh[i] = (int)(t % a.Ho); t /= a.Ho;
    od[i] = (int)(t % a.Do); nn[i] = (int)(t / a.Do);
  }
  auto load_step = [&](long long m0) __attribute__((always_inline)) {     // called with m0 = m_begin, m_begin + 32, ... in order
    bool oky[2], okx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long m = m0 + (tid >> 4) + i * 16;
      const bool live = m < m_end;
      const unsigned mc = (unsigned)(m < a.M ? m : a.M - 1);        // 32-bit element offsets (host: M * Cn and the x tensor fit 31 bits)
      oky[i] = live && y_ok;
      ry[i] = *reinterpret_cast<const uint4*>(a.dy + (mc * (unsigned)a.Cn + (unsigned)y_off));
      const int d = od[i] * a.stride + kd - 1, h = oh[i] * a.stride + kh - 1, w = ow[i] * a.stride + kw - 1;
      okx[i] = live && x_ok && (unsigned)d < (unsigned)a.Di && (unsigned)h < (unsigned)a.Hi && (unsigned)w < (unsigned)a.Wi;
      const int dc = min(max(d, 0), a.Di - 1), hc = min(max(h, 0), a.Hi - 1), wc = min(max(w, 0), a.Wi - 1);
      const int nc = min(nn[i], a.N - 1);
      rx[i] = *reinterpret_cast<const uint4*>(a.x + (unsigned)((((nc * a.Di + dc) * a.Hi + hc) * a.Wi + wc) * a.Cs + x_off));
    }
    okm = 0;                                              // zeroed in store_step (masks next to the loads would wait for them here)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      okm |= (oky[i] ? (1u << i) : 0u) | (okx[i] ? (4u << i) : 0u);
      ow[i] += 32;                                        // the row this slot stages next step
      while (ow[i] >= a.Wo) {
        ow[i] -= a.Wo;
        if (++oh[i] == a.Ho) { oh[i] = 0; if (++od[i] == a.Do) { od[i] = 0; ++nn[i]; } }
      }
    }
  };
  auto store_step = [&](int buf) {
    unsigned char* Ys = smem + buf * 2 * TB;
    unsigned char* Xs = Ys + TB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256;
      const int row = c >> 4, chn = c & 15;
      uint4 vy = ry[i], vx = rx[i];
      if (!((okm >> i) & 1)) vy = make_uint4(0, 0, 0, 0);
      if (!((okm >> (2 + i)) & 1)) vx = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(Ys + sw256(row, chn)) = vy;
      *reinterpret_cast<uint4*>(Xs + sw256(row, chn)) = vx;
    }
  };
  f32x4_t acc[TMW][TNW];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (m_begin < m_end) {
    load_step(m_begin);
    store_step(0);
  }
  __syncthreads();
  // two steps per trip: the LDS buffer index is a literal in each half (fragment addresses = per-lane base + immediate offset)
  auto step = [&](long long m0, int buf) __attribute__((always_inline)) {
    if (m0 + 32 < m_end) load_step(m0 + 32);
    const unsigned char* Ys = smem + buf * 2 * TB;
    const unsigned char* Xs = Ys + TB;
    // transposed fragments: lane (i = r16, kg) needs tile[8 kg + e][c0 + i], e = 0..7 = two 4-row blocks; within a 16-lane
    // group lane 4q + p supplies the address of row (block row q), columns c0 + 4p .. +3
    h16x8 af[TMW], bf[TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      const int c0 = (wm * TMW + i) * 16;                   // first channel of the tile
      const int chn = (c0 >> 3) + (p >> 1);
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Ys + sw256(8 * kg + q, chn) + 8 * (p & 1)));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Ys + sw256(8 * kg + 4 + q, chn) + 8 * (p & 1)));
      af[i] = h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int c0 = (wn * TNW + j) * 16;
      const int chn = (c0 >> 3) + (p >> 1);
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Xs + sw256(8 * kg + q, chn) + 8 * (p & 1)));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Xs + sw256(8 * kg + 4 + q, chn) + 8 * (p & 1)));
      bf[j] = h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j) acc[i][j] = mfma16x16x32<FMT>(af[i], bf[j], acc[i][j]);
    if (m0 + 32 < m_end) store_step(buf ^ 1);
    __syncthreads();
  };
  for (long long m0 = m_begin; m0 < m_end; m0 += 64) {
    step(m0, 0);
    if (m0 + 32 < m_end) step(m0 + 32, 1);
  }
  // D[i = cn][j = cs]: lane holds column cs = r16, rows cn = 4 kg + r
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int col = (wn * TNW + j) * 16 + r16;
      const int cs = PAIR ? (col & 63) : cs0 + col;
      const int otap = PAIR ? 2 * tz + (col >> 6) : tz;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cn = cn0 + (wm * TMW + i) * 16 + kg * 4 + r;
        if (cn < a.Cn && cs < a.Cs && otap < a.ntap) atomicAdd(&a.dw[((long long)otap * a.Cn + cn) * a.Cs + cs], acc[i][j][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient of the stride-2 k = 4 convs from a staged SOURCE BLOCK (round 5).  dwgrad_cl_kernel gathers the X rows of one
// tap pair per 32-voxel step: 16 KB of loads per 2 x 16 MFMAs per wave = 64 flop per loaded byte, and the kernel sits at 0.2 of
// the MFMA peak waiting on L1.  Here a workgroup of 8 waves owns ONE parity class of taps (k = 4, stride 2: the 64 taps fall
// into 8 classes of 2 x 2 x 2 taps which read source voxels of one parity) x 64 source channels x 128 destination channels, and
// walks 4 x 4 x 4 blocks of output voxels: per block it stages the 64 dY rows (16 KB) and the 5 x 5 x 5 source rows of the class
// ONCE (16 KB) and every wave (= one tap of the class) takes its B fragments from the block at its tap's offset: 32 KB of loads
// per 8 x 64 MFMAs = 256 flop per byte.  The fragments are transposed reads (ds_read_b64_tr_b16) as in dwgrad_cl_kernel: rows =
// voxels, a lane supplies the address of one row, so the rows of a tap's fragment may sit anywhere in the block.
// Source-block layout in LDS: row = hz * 32 + hy * 6 + hx (128 bytes each), 16-byte chunk index XORed with ((row >> 1) & 3) << 1:
// the 16 rows one transposed read touches (x = 0..3 at two y and two z) then spread evenly over the banks.
struct DWgHK {
  const u16* x; const u16* dy; float* dw;
  int N, Di, Hi, Wi, Do, Ho, Wo, Cs, Cn;
  int nbz, nby, nbx, nblk;   // 4 x 4 x 4 blocks of output voxels per axis; N * nbz * nby * nbx
  int nsplit, ncs, ncn;      // block-range splits; 64-channel tiles of Cs; 128-channel tiles of Cn
  int ngroup;                // nsplit * ncs * ncn; workgroup id = class * ngroup + group (ngroup % 8 == 0: a group's 8 classes,
};                           // which stream the same dY rows, land on one XCD and share them through its L2)
template <int FMT>
__global__ __launch_bounds__(512, 2) void dwgrad_halo_kernel(const DWgHK a) {
  constexpr int YB = 64 * 256, XB = 160 * 128, BUF = YB + XB;
  extern __shared__ __attribute__((aligned(256))) unsigned char dwh_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4, q = r16 >> 2, p = r16 & 3;
  const int cls = blockIdx.x / a.ngroup, grp = blockIdx.x % a.ngroup;
  const int ez = cls >> 2, ey = (cls >> 1) & 1, ex = cls & 1;          // parity bit of the class's taps per axis
  const int split = grp / (a.ncs * a.ncn), tile = grp % (a.ncs * a.ncn);
  const int cs0 = (tile % a.ncs) * 64, cn0 = (tile / a.ncs) * 128;
  const int per = (a.nblk + a.nsplit - 1) / a.nsplit;
  const int b_begin = split * per, b_end = min(a.nblk, b_begin + per);
  // staging: dY 64 rows x 16 chunks, X 125 rows x 8 chunks -> two 16-byte pieces of each per thread, the same pieces of every block
  const int ych = tid & 15, yx = (tid >> 4) & 3, yy = (tid >> 6) & 3, yz0 = tid >> 8;         // piece i: z = yz0 + 2 i
  const int xpc = tid & 7;
  int xz[2], xy[2], xx[2], xrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int l = (tid >> 3) + 64 * i;                     // 0..127, rows >= 125 do not exist
    xz[i] = l / 25; xy[i] = (l % 25) / 5; xx[i] = l % 5;
    xrow[i] = l < 125 ? xz[i] * 32 + xy[i] * 6 + xx[i] : -1;
  }
  // the loads of a block are only ISSUED here (clamped addresses); what falls outside the volume is zeroed when the registers are
  // written to LDS a whole block of MFMAs later -- masking next to the loads makes hipcc wait for them on the spot
  uint4 ry[2], rx[2];
  unsigned okm = 0;                                        // bits 0, 1: dY pieces inside, bits 2, 3: X pieces inside
  int nb_x = 0, nb_y = 0, nb_z = 0, nb_n = 0;              // coordinates of the next block to load (carried, not divided out)
  {
    int t = b_begin;
    nb_x = t % a.nbx; t /= a.nbx;
    nb_y = t % a.nby; t /= a.nby;
    nb_z = t % a.nbz; nb_n = t / a.nbz;
  }
  auto load_block = [&]() __attribute__((always_inline)) {   // called for b_begin, b_begin + 1, ... in order
    const int bx = nb_x, by = nb_y, bz = nb_z, n = nb_n;
    okm = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int od = bz * 4 + yz0 + 2 * i, oh = by * 4 + yy, ow = bx * 4 + yx;
      okm |= (od < a.Do && oh < a.Ho && ow < a.Wo) ? (1u << i) : 0u;
      const unsigned row = (unsigned)(((n * a.Do + min(od, a.Do - 1)) * a.Ho + min(oh, a.Ho - 1)) * a.Wo + min(ow, a.Wo - 1));
      ry[i] = *reinterpret_cast<const uint4*>(a.dy + (row * (unsigned)a.Cn + (unsigned)(cn0 + ych * 8)));
      const int d = 2 * (bz * 4 + xz[i]) + ez - 1, h = 2 * (by * 4 + xy[i]) + ey - 1, w = 2 * (bx * 4 + xx[i]) + ex - 1;
      okm |= (xrow[i] >= 0 && (unsigned)d < (unsigned)a.Di && (unsigned)h < (unsigned)a.Hi && (unsigned)w < (unsigned)a.Wi) ? (4u << i) : 0u;
      const int dc = min(max(d, 0), a.Di - 1), hc = min(max(h, 0), a.Hi - 1), wc = min(max(w, 0), a.Wi - 1);
      rx[i] = *reinterpret_cast<const uint4*>(a.x + ((unsigned)(((n * a.Di + dc) * a.Hi + hc) * a.Wi + wc) * (unsigned)a.Cs + (unsigned)(cs0 + xpc * 8)));
    }
    if (++nb_x == a.nbx) { nb_x = 0; if (++nb_y == a.nby) { nb_y = 0; if (++nb_z == a.nbz) { nb_z = 0; ++nb_n; } } }
  };
  auto store_block = [&](int buf) __attribute__((always_inline)) {
    unsigned char* Ys = dwh_smem + buf * BUF;
    unsigned char* Xs = Ys + YB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<uint4*>(Ys + sw256(((yz0 + 2 * i) * 4 + yy) * 4 + yx, ych)) = (okm >> i) & 1 ? ry[i] : make_uint4(0, 0, 0, 0);
      if (xrow[i] >= 0)
        *reinterpret_cast<uint4*>(Xs + xrow[i] * 128 + ((xpc ^ (((xrow[i] >> 1) & 3) << 1)) << 4)) = (okm >> (2 + i)) & 1 ? rx[i] : make_uint4(0, 0, 0, 0);
    }
  };
  // fragment addresses: dY rows 32 s + 8 kg + q (+ 4), X rows of the voxels (z = 2 s + (kg >> 1), y = 2 (kg & 1) (+ 1), x = q) at the
  // wave's tap (jz, jy, jx); the tile index of a read enters as an XOR of bits 5..7 (the swizzles only touch the chunk bits)
  const int jz = wv >> 2, jy = (wv >> 1) & 1, jx = wv & 1;
  const int ph = p >> 1, pb = (p & 1) * 8;
  unsigned a_base[2], b_base[2];
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int row = 8 * kg + q + 4 * hi;
    const int ca = ((row & 3) << 2) | ((row >> 2) & 3);
    a_base[hi] = (unsigned)(row * 256 + ((ph ^ ca) << 4) + pb);
    const int hr = ((kg >> 1) + jz) * 32 + (2 * (kg & 1) + hi + jy) * 6 + q + jx;
    b_base[hi] = (unsigned)(YB + hr * 128 + (((hr >> 1) & 3) << 5) + (ph << 4) + pb);
  }
  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (b_begin < b_end) {
    load_block();
    store_block(0);
  }
  __syncthreads();
  auto step = [&](int b, int buf) __attribute__((always_inline)) {
    if (b + 1 < b_end) load_block();
    const unsigned char* S = dwh_smem + buf * BUF;
    auto rd_a = [&](int n) __attribute__((always_inline)) {        // dY fragment n = 8 s + tile
      const int s = n >> 3, i = n & 7;
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(S + (a_base[0] ^ (unsigned)(i << 5)) + s * 32 * 256));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(S + (a_base[1] ^ (unsigned)(i << 5)) + s * 32 * 256));
      return h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto rd_b = [&](int s, int j) __attribute__((always_inline)) {
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(S + (b_base[0] ^ (unsigned)(j << 5)) + s * 64 * 128));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(S + (b_base[1] ^ (unsigned)(j << 5)) + s * 64 * 128));
      return h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // the transposed reads come back after ~300 cycles when eight waves queue them, a group of four MFMAs lasts 64: the dY fragments
    // are requested AHEAD fragments before their group (hipcc's own schedule keeps one in flight and the matrix cores wait on LDS)
    constexpr int AHEAD = 3;
    h16x8 bf[2][4], aq[AHEAD + 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[0][j] = rd_b(0, j);
#pragma unroll
    for (int n = 0; n < AHEAD; ++n) aq[n] = rd_a(n);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      if (n + AHEAD < 16) aq[(n + AHEAD) % (AHEAD + 1)] = rd_a(n + AHEAD);
      if (n >= 2 && n < 6) bf[1][n - 2] = rd_b(1, n - 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[n & 7][j] = mfma16x16x32<FMT>(aq[n % (AHEAD + 1)], bf[n >> 3][j], acc[n & 7][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (b + 1 < b_end) store_block(buf ^ 1);
    __syncthreads();
  };
  for (int b = b_begin; b < b_end; b += 2) {
    step(b, 0);
    if (b + 1 < b_end) step(b + 1, 1);
  }
  // D[i = cn][j = cs]: lane holds column cs = r16, rows cn = 4 kg + r, of the wave's tap
  const int otap = ((2 * jz + ez) * 4 + 2 * jy + ey) * 4 + 2 * jx + ex;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(&a.dw[((long long)otap * a.Cn + cn0 + i * 16 + kg * 4 + r) * a.Cs + cs0 + j * 16 + r16], acc[i][j][r]);
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient of the FIRST conv (8 padded input channels): with so few channels per tap the 27 taps become the N axis
// of the GEMM -- dW[cn][(tap, ci)] = sum_m dY[m][cn] * X[src(m, tap)][ci], N = 27 x 8 = 216 (14 tiles of 16) -- so that dY
// is streamed ONCE (not once per tap) and a 32-voxel step carries 14 MFMAs per wave instead of one.
__device__ __forceinline__ int sw512(int row, int chunk) { return row * 512 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

template <int FMT>
__global__ __launch_bounds__(256, 2) void dwgrad_c8_kernel(const DWgK a) {
  constexpr int YB = 32 * 256, XB = 32 * 512;            // bytes per staged dY / X-columns tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (YB + XB)];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4, q = r16 >> 2, p = r16 & 3;
  const int cn0 = blockIdx.x * 64;
  const int split = blockIdx.y;
  const long long per = ((a.M + a.msplit - 1) / a.msplit + 31) / 32 * 32;
  const long long m_begin = (long long)split * per, m_end = m_begin + per < a.M ? m_begin + per : a.M;
  uint4 ry, rx[4];
  unsigned okm = 0;                                       // bits 0..3: X pieces to keep, bit 4: the dY piece
  // a thread stages the same tap for its four rows (slot = tid + 256 i: tap = tid & 31, row = (tid >> 5) + 8 i), and each row
  // moves on by 32 voxels per step: coordinates carried along, loads unconditional from clamped addresses (see dwgrad_cl_kernel)
  const int xtap = tid & 31;
  const bool xtap_ok = xtap < 27;
  const int xkd = xtap / 9, xkh = (xtap / 3) % 3, xkw = xtap % 3;
  const int ychn = tid & 7;
  const bool y_ok = cn0 + ychn * 8 < a.Cn;
  const int y_off = y_ok ? cn0 + ychn * 8 : 0;
  int ow[4], oh[4], od[4], nn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long t = m_begin + (tid >> 5) + i * 8;
    ow[i] = (int)(t % a.Wo); t /= a.Wo;
    oh[i] = (int)(t % a.Ho); t /= a.Ho;
    od[i] = (int)(t % a.Do); nn[i] = (int)(t / a.Do);
  }
  auto load_step = [&](long long m0) __attribute__((always_inline)) {     // m0 = m_begin, m_begin + 32, ... in order
    bool oky, okx[4];
    {
      const long long m = m0 + (tid >> 3);
      oky = m < m_end && y_ok;
      const long long mc = m < a.M ? m : a.M - 1;
      ry = *reinterpret_cast<const uint4*>(a.dy + mc * a.Cn + y_off);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long m = m0 + (tid >> 5) + i * 8;
      const int d = od[i] * a.stride + xkd - 1, h = oh[i] * a.stride + xkh - 1, w = ow[i] * a.stride + xkw - 1;
      okx[i] = m < m_end && xtap_ok && (unsigned)d < (unsigned)a.Di && (unsigned)h < (unsigned)a.Hi && (unsigned)w < (unsigned)a.Wi;
      const int dc = min(max(d, 0), a.Di - 1), hc = min(max(h, 0), a.Hi - 1), wc = min(max(w, 0), a.Wi - 1);
      const int nc = min(nn[i], a.N - 1);
      rx[i] = *reinterpret_cast<const uint4*>(a.x + ((((long long)nc * a.Di + dc) * a.Hi + hc) * a.Wi + wc) * 8);
    }
    okm = oky ? 16u : 0u;                                 // zeroed in store_step (masks next to the loads would wait for them here)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      okm |= okx[i] ? (1u << i) : 0u;
      ow[i] += 32;
      while (ow[i] >= a.Wo) {
        ow[i] -= a.Wo;
        if (++oh[i] == a.Ho) { oh[i] = 0; if (++od[i] == a.Do) { od[i] = 0; ++nn[i]; } }
      }
    }
  };
  auto store_step = [&](int buf) {
    unsigned char* Ys = smem + buf * (YB + XB);
    unsigned char* Xs = Ys + YB;
    uint4 vy = ry;
    if (!(okm & 16u)) vy = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(Ys + sw256(tid >> 3, tid & 7)) = vy;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int slot = tid + i * 256;
      uint4 vx = rx[i];
      if (!((okm >> i) & 1)) vx = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(Xs + sw512(slot >> 5, slot & 31)) = vx;
    }
  };
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (m_begin < m_end) {
    load_step(m_begin);
    store_step(0);
  }
  __syncthreads();
  int buf = 0;
  for (long long m0 = m_begin; m0 < m_end; m0 += 32, buf ^= 1) {
    if (m0 + 32 < m_end) load_step(m0 + 32);
    const unsigned char* Ys = smem + buf * (YB + XB);
    const unsigned char* Xs = Ys + YB;
    h16x8 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int chn = 2 * i + (p >> 1);
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Ys + sw256(8 * kg + q, chn) + 8 * (p & 1)));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Ys + sw256(8 * kg + 4 + q, chn) + 8 * (p & 1)));
      af[i] = h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nt = wv + 4 * j;                              // N tile (16 of the 224 (tap, ci) columns); tiles 14, 15 are all padding
      const int chn = 2 * nt + (p >> 1);                      // < 32 always (nt <= 15)
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Xs + sw512(8 * kg + q, chn) + 8 * (p & 1)));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Xs + sw512(8 * kg + 4 + q, chn) + 8 * (p & 1)));
      bf[j] = h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x16x32<FMT>(af[i], bf[j], acc[i][j]);
    if (m0 + 32 < m_end) store_step(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = (wv + 4 * j) * 16 + r16;
      const int tap = col >> 3, ci = col & 7;
      if (tap >= 27) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cn = cn0 + i * 16 + kg * 4 + r;
        if (cn < a.Cn) atomicAdd(&a.dw[((long long)tap * a.Cn + cn) * 8 + ci], acc[i][j][r]);
      }
    }
}

// The same weight gradient (8 -> 64, stride 1) without the per-step gather: a workgroup walks 4 x 4 x 16 blocks of output voxels;
// per block the (3 + K)^2 x (15 + K) halo of X (16 bytes per voxel, 10 / 15 KB) and the block's 256 x 64 dY values (eight
// 32-voxel K steps in the transposed-read layout above) are put in LDS ONCE, and the X operand of a (tap pair, ci) column tile is
// read straight from the halo -- a tap is an address offset -- with ds_read_b64_tr_b16.  One barrier pair per 256 voxels instead
// of one per 32, no K^3-fold re-fetch of X, the next block's loads in flight behind the 128 MFMAs of the current one.
// K = 4: 64 taps x 8 channels = 512 columns = 32 column tiles: the taps are split in two halves over blockIdx.y (16 tiles each,
// none of them padding; K = 3 uses 14 of its 16).  (D, H, W) are the OUTPUT extents, the input has (D, H, W) + K - 3.
struct DWg8HK {
  const u16* x; const u16* dy; float* dw;
  int N, D, H, W;
  int td, th, tw, ntile;
};
template <int FMT, int K>
__global__ __launch_bounds__(256, 2) void dwgrad_c8_halo_kernel(const DWg8HK a) {
  constexpr int PD = 3 + K, PH = 3 + K, PW = 15 + K, NV = PD * PH * PW, NX = (NV + 255) / 256, NTAP = K * K * K;
  const int Dx = a.D + K - 3, Hx = a.H + K - 3, Wx = a.W + K - 3;
  const int tap0 = blockIdx.y * 32;
  constexpr int YB = 32 * 256;                             // one K step of dY: 32 voxels x (64 of 128) channels, sw256 image
  __shared__ __attribute__((aligned(16))) unsigned char ys[8 * YB];
  __shared__ __attribute__((aligned(16))) unsigned char xt[(NV + 1) * 16];          // + one zero voxel for the padding columns
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4, q = r16 >> 2, p = r16 & 3;
  // X fragment of column tile nt = wv + 4 j: columns = (tap 2 nt + (p >> 1), ci 4 (p & 1) ..): the tap is a constant byte offset
  int xoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = tap0 + 2 * (wv + 4 * j) + (p >> 1);
    const int kd = t / (K * K), kh = (t / K) % K, kw = t % K;
    xoff[j] = t < NTAP ? ((kd * PH + kh) * PW + kw) * 16 + 8 * (p & 1) : -1;
  }
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (tid == 0) *reinterpret_cast<uint4*>(xt + NV * 16) = make_uint4(0, 0, 0, 0);
  uint4 py[8], px[NX];
  unsigned okm = 0;                                       // bits 0..7: dY pieces inside the volume, bits 8..: X pieces
  const int per_n = a.td * a.th * a.tw;
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int n = t / per_n; int b = t - n * per_n;
    const int tw_i = b % a.tw; b /= a.tw;
    const int th_i = b % a.th, td_i = b / a.th;
    const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
    const u16* yn = a.dy + (long long)n * a.D * a.H * a.W * 64;
    const u16* xn = a.x + (long long)n * Dx * Hx * Wx * 8;
    okm = 0;                                              // zeroing waits until store_tile: masks next to the loads make hipcc wait for them here
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + i * 256;
      const int v = idx >> 3, c = idx & 7;
      const int rb = v >> 4, w = w0 + (v & 15);
      const int d = d0 + (rb >> 2), h = h0 + (rb & 3);
      okm |= (d < a.D && h < a.H && w < a.W) ? (1u << i) : 0u;
      const int dc = min(d, a.D - 1), hc = min(h, a.H - 1), wc = min(w, a.W - 1);
      py[i] = *reinterpret_cast<const uint4*>(yn + (((long long)dc * a.H + hc) * a.W + wc) * 64 + c * 8);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int pi = min(tid + i * 256, NV - 1);
      const int pw = pi % PW, r = pi / PW;
      const int ph = r % PH, pd = r / PH;
      const int d = d0 - 1 + pd, h = h0 - 1 + ph, w = w0 - 1 + pw;
      okm |= ((unsigned)d < (unsigned)Dx && (unsigned)h < (unsigned)Hx && (unsigned)w < (unsigned)Wx) ? (256u << i) : 0u;
      const int dc = min(max(d, 0), Dx - 1), hc = min(max(h, 0), Hx - 1), wc = min(max(w, 0), Wx - 1);
      px[i] = *reinterpret_cast<const uint4*>(xn + (((long long)dc * Hx + hc) * Wx + wc) * 8);
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + i * 256;
      const int v = idx >> 3, c = idx & 7;
      const int rb = v >> 4;                              // (dz, hy) row of the block; K step = rb >> 1, row in the step = 16 (rb & 1) + w
      uint4 v4 = py[i];
      if (!((okm >> i) & 1)) v4 = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(ys + (rb >> 1) * YB + sw256((rb & 1) * 16 + (v & 15), c)) = v4;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int pi = tid + i * 256;
      uint4 v4 = px[i];
      if (!((okm >> (8 + i)) & 1)) v4 = make_uint4(0, 0, 0, 0);
      if (pi < NV) *reinterpret_cast<uint4*>(xt + pi * 16) = v4;
    }
  };
  int t = blockIdx.x;
  if (t < a.ntile) load_tile(t);
  for (; t < a.ntile; t += gridDim.x) {
    __syncthreads();                                      // everybody is done with the previous block's tiles
    store_tile();
    __syncthreads();
    if (t + (int)gridDim.x < a.ntile) load_tile(t + gridDim.x);
#pragma unroll 2
    for (int s8 = 0; s8 < 8; ++s8) {
      const unsigned char* Ys = ys + s8 * YB;
      h16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int chn = 2 * i + (p >> 1);
        const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Ys + sw256(8 * kg + q, chn) + 8 * (p & 1)));
        const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(Ys + sw256(8 * kg + 4 + q, chn) + 8 * (p & 1)));
        af[i] = h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      // K rows 8 kg + q (+ 4) of the step = voxels w = 8 (kg & 1) + q (+ 4) of block row rb = 2 s8 + (kg >> 1)
      const int rb = 2 * s8 + (kg >> 1);
      const int hb = (((rb >> 2) * PH + (rb & 3)) * PW + 8 * (kg & 1) + q) * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o_lo = xoff[j] >= 0 ? hb + xoff[j] : NV * 16 + 8 * (p & 1);
        const int o_hi = xoff[j] >= 0 ? o_lo + 64 : o_lo;
        const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(xt + o_lo));
        const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(xt + o_hi));
        bf[j] = h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x16x32<FMT>(af[i], bf[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = (wv + 4 * j) * 16 + r16;
      const int tap = tap0 + (col >> 3), ci = col & 7;
      if (tap >= NTAP) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cn = i * 16 + kg * 4 + r;
        atomicAdd(&a.dw[((long long)tap * 64 + cn) * 8 + ci], acc[i][j][r]);
      }
    }
}

// data gradient of the FIRST conv (64 -> 8 padded input channels, stride 1): the generic kernel gathers every dY row 27 times
// through the caches for 4 MFMAs per wave per step (750 us @128^3, bound by cache throughput).  Here a workgroup owns a
// 4 x 4 x 16 block of destination voxels and keeps the 6 x 6 x 18 halo of dY it needs in LDS (32 channels at a time, 80-byte
// voxel pitch: the 16 lanes of a fragment read sit 20 banks apart), so the 27 taps are 27 shifted LDS reads of the same tile:
// no barrier and no global activation traffic inside the tap loop; the weight fragment of a tap (1 KB) comes from L1/L2, one
// tap ahead.  Weights ride on the M side (8 of 16 rows used): a lane ends up with 4 consecutive channels of one voxel.
// (Round 5: a dense 64-byte pitch with a chunk XOR that is conflict-free for the real ds_read_b128 lane groups -- they mix two
// k-groups, this pitch was built for one -- removed every bank conflict and ran 12 % SLOWER: at the 80-byte pitch all 16 fragment
// reads of a (kd, kh) row are one base register + immediates; the XOR costs vector instructions per read in a loop of 4 MFMAs per tap.)
// K = 4: the halo is 7 x 7 x 19 (74 KB at the 80-byte pitch: two workgroups per CU), 64 taps; destination d takes tap kd from
// source d + 1 - kd, so the halo starts at d0 - (K - 2).  (D, H, W) are the DESTINATION extents (the conv's input), the
// gradient g has (D, H, W) + 3 - K.
struct DDg8K {
  const u16* g; const u16* w; u16* dx;
  int N, D, H, W;
  int td, th, tw;          // tiles per axis
};
template <int FMT, int K>
__global__ __launch_bounds__(256, K == 3 ? 3 : 2) void dconv_dgrad_c8_kernel(const DDg8K a) {
  constexpr int PD = 3 + K, PH = 3 + K, PW = 15 + K, NV = PD * PH * PW, PITCH = 80, NTAP = K * K * K;
  const int Dg = a.D + 3 - K, Hg = a.H + 3 - K, Wg = a.W + 3 - K;
  __shared__ __attribute__((aligned(16))) unsigned char tile[NV * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int per_n = a.td * a.th * a.tw;
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int n = b / per_n; b -= n * per_n;
  const int tw_i = b % a.tw, t2 = b / a.tw;
  const int th_i = t2 % a.th, td_i = t2 / a.th;
  const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
  const u16* gn = a.g + (long long)n * Dg * Hg * Wg * 64;
  f32x4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // weights: w[tap][ci (8 rows)][co (64)]; lanes r16 >= 8 are the padding rows of the M side
  const bool wrow = r16 < 8;
  const u16* wl = a.w + (wrow ? r16 : 0) * 64 + kg * 8;
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();                           // every wave is done reading the first 32 channels
    // the halo pieces in groups of 8: every load of a group is issued (from a clamped address) before the first is written -- one
    // load, one wait, one store per trip made a block pay ~30 memory latencies for 3.4 us of MFMAs (round 5)
    constexpr int NPC = NV * 4, STG = 8;
#pragma unroll 1
    for (int i0 = 0; i0 < NPC; i0 += 256 * STG) {
      uint4 v[STG];
      unsigned okm = 0;
#pragma unroll
      for (int u = 0; u < STG; ++u) {
        const int idx = min(i0 + u * 256 + tid, NPC - 1);
        const int p = idx >> 2, c = idx & 3;
        const int pw = p % PW, q = p / PW;
        const int ph = q % PH, pd = q / PH;
        const int d = d0 - (K - 2) + pd, h = h0 - (K - 2) + ph, w = w0 - (K - 2) + pw;
        okm |= ((unsigned)d < (unsigned)Dg && (unsigned)h < (unsigned)Hg && (unsigned)w < (unsigned)Wg) ? 1u << u : 0u;
        const int dc = min(max(d, 0), Dg - 1), hc = min(max(h, 0), Hg - 1), wc = min(max(w, 0), Wg - 1);
        v[u] = *reinterpret_cast<const uint4*>(gn + (((long long)dc * Hg + hc) * Wg + wc) * 64 + half * 32 + c * 8);
      }
#pragma unroll
      for (int u = 0; u < STG; ++u) {
        const int idx = i0 + u * 256 + tid;
        uint4 t = v[u];
        if (!((okm >> u) & 1)) t = make_uint4(0, 0, 0, 0);
        if (idx < NPC) *reinterpret_cast<uint4*>(tile + (idx >> 2) * PITCH + (idx & 3) * 16) = t;
      }
    }
    __syncthreads();
    // weight fragments a whole (kd, kh) row ahead (K taps = 4 K MFMAs per wave between request and use; one tap ahead left the L2
    // latency of every fragment exposed)
    uint4 wrow_cur[K], wrow_nxt[K];
#pragma unroll
    for (int kw = 0; kw < K; ++kw) wrow_cur[kw] = *reinterpret_cast<const uint4*>(wl + (long long)kw * 512 + half * 32);   // (padding rows: zeroed at use)
#pragma unroll 1
    for (int kd = 0; kd < K; ++kd)
#pragma unroll 1
      for (int kh = 0; kh < K; ++kh) {
        {
          const int nrow = min(kd * K + kh + 1, K * K - 1);
#pragma unroll
          for (int kw = 0; kw < K; ++kw)
            wrow_nxt[kw] = *reinterpret_cast<const uint4*>(wl + (long long)(nrow * K + kw) * 512 + half * 32);
        }
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const h16x8 bw = __builtin_bit_cast(h16x8, wrow ? wrow_cur[kw] : make_uint4(0, 0, 0, 0));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rb = wv * 4 + i;                   // (dz, hy) row of the block
            const int dz = rb >> 2, hy = rb & 3;
            // destination (dz, hy, r16) takes tap (kd, kh, kw) from source (+1 - kd, +1 - kh, +1 - kw); halo origin is -(K - 2)
            const int p = ((dz + K - 1 - kd) * PH + (hy + K - 1 - kh)) * PW + (r16 + K - 1 - kw);
            const h16x8 av = *reinterpret_cast<const h16x8*>(tile + p * PITCH + kg * 16);
            acc[i] = mfma16x16x32<FMT>(bw, av, acc[i]);
          }
        }
#pragma unroll
        for (int kw = 0; kw < K; ++kw) wrow_cur[kw] = wrow_nxt[kw];
      }
  }
  // lane: channels 4 kg .. 4 kg + 3 (kg < 2) of voxel (d0 + dz, h0 + hy, w0 + r16)
  if (kg < 2) {
    u16* dn = a.dx + (long long)n * a.D * a.H * a.W * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rb = wv * 4 + i;
      const int d = d0 + (rb >> 2), h = h0 + (rb & 3), w = w0 + r16;
      if (d < a.D && h < a.H && w < a.W)
        *reinterpret_cast<uint2*>(dn + (((long long)d * a.H + h) * a.W + w) * 8 + kg * 4) =
            make_uint2(cvt_pack<FMT>(acc[i][0], acc[i][1]), cvt_pack<FMT>(acc[i][2], acc[i][3]));
    }
  }
}

// forward of the FIRST conv (8 padded input channels -> 64, stride 1, bias + LeakyReLU): same block shape as the kernel above.
// The 6 x 6 x 18 input halo is 10 KB of LDS (16 bytes per voxel); one K step = one (kd, kh) row = 3 kw taps x 8 channels
// (+ 8 zero-weight lanes), read as ONE 16-byte LDS load per lane: lane (voxel r16, k-group kg) takes voxel r16 + kg of the row.
// 9 steps x 16 MFMAs per wave, weights (M side, 4 blocks of 16 channels) from L1/L2 one step ahead.  The assignment of output
// channels to MFMA rows is free, so block j, row m carries channel (j >> 1) * 32 + (m >> 2) * 8 + (j & 1) * 4 + (m & 3): a lane
// then owns 8 CONSECUTIVE channels per block pair and the four k-groups of a voxel write 64 contiguous bytes per store.
// K = 4: 7 x 7 x 19 halo, 16 (kd, kh) steps whose 4 kw taps x 8 channels fill the 32-deep MFMA exactly (no zero-weight lanes).
// (D, H, W) are the OUTPUT extents; the input has (D, H, W) + K - 3.
struct DFw8K {
  const u16* x; const u16* w; const float* bias; u16* y;
  int N, D, H, W;
  int td, th, tw;
  int act; float slope;
};
template <int FMT, int K>
__global__ __launch_bounds__(256, 2) void dconv_fwd_c8_kernel(const DFw8K a) {
  constexpr int PD = 3 + K, PH = 3 + K, PW = 15 + K, NV = PD * PH * PW;
  const int Dx = a.D + K - 3, Hx = a.H + K - 3, Wx = a.W + K - 3;
  __shared__ __attribute__((aligned(16))) unsigned char tile[(NV + 2) * 16];      // + 2 voxels: the zero-weight lanes of the last row (K = 3)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int per_n = a.td * a.th * a.tw;
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int n = b / per_n; b -= n * per_n;
  const int tw_i = b % a.tw, t2 = b / a.tw;
  const int th_i = t2 % a.th, td_i = t2 / a.th;
  const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
  const u16* xn = a.x + (long long)n * Dx * Hx * Wx * 8;
  {
    // all pieces of the halo requested (clamped addresses) before the first is written: one exposed latency instead of four
    constexpr int NSTG = (NV + 2 + 255) / 256;
    uint4 v[NSTG];
    unsigned okm = 0;
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
      const int p = min(tid + u * 256, NV - 1);
      const int pw = p % PW, q = p / PW;
      const int ph = q % PH, pd = q / PH;
      const int d = d0 - 1 + pd, h = h0 - 1 + ph, w = w0 - 1 + pw;
      okm |= (tid + u * 256 < NV && (unsigned)d < (unsigned)Dx && (unsigned)h < (unsigned)Hx && (unsigned)w < (unsigned)Wx) ? 1u << u : 0u;
      const int dc = min(max(d, 0), Dx - 1), hc = min(max(h, 0), Hx - 1), wc = min(max(w, 0), Wx - 1);
      v[u] = *reinterpret_cast<const uint4*>(xn + (((long long)dc * Hx + hc) * Wx + wc) * 8);
    }
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
      uint4 t = v[u];
      if (!((okm >> u) & 1)) t = make_uint4(0, 0, 0, 0);
      if (tid + u * 256 < NV + 2) *reinterpret_cast<uint4*>(tile + (tid + u * 256) * 16) = t;
    }
  }
  // weights: w[r9][co (64)][k (32)]; this lane supplies row m = r16 of block j = channel co_of(j)
  int wrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wrow[j] = ((j >> 1) * 32 + (r16 >> 2) * 8 + (j & 1) * 4 + (r16 & 3)) * 32 + kg * 8;
  // weight fragments AH steps ahead (a step = 16 MFMAs per wave = 256 cycles; the 36 / 64 KB of weights do not stay in L1: with one
  // step of lookahead SQ_WAIT_ANY was 74 % of the wave time, round 5: 200 -> 178 us); the step loop is unrolled so that the ring index
  // is static.  (The same depth in the 8-channel data gradient needed its tap loops unrolled: 222 registers, 431 -> 470 us, dropped.)
  constexpr int AH = 4;
  uint4 wf[AH][4];
#pragma unroll
  for (int u = 0; u < AH; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[u][j] = *reinterpret_cast<const uint4*>(a.w + (u < K * K ? u : K * K - 1) * 2048 + wrow[j]);
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
#pragma unroll
  for (int r9 = 0; r9 < K * K; ++r9) {
    const int kd = r9 / K, kh = r9 - kd * K;
    h16x8 bw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bw[j] = __builtin_bit_cast(h16x8, wf[r9 % AH][j]);
    if (r9 + AH < K * K) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[r9 % AH][j] = *reinterpret_cast<const uint4*>(a.w + (r9 + AH) * 2048 + wrow[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rb = wv * 4 + i;
      const int dz = rb >> 2, hy = rb & 3;
      // destination (dz, hy, r16), tap (kd, kh, kw = kg): source (+kd - 1, +kh - 1, +kg - 1), halo origin -1
      const int p = ((dz + kd) * PH + (hy + kh)) * PW + r16 + kg;
      const h16x8 av = *reinterpret_cast<const h16x8*>(tile + p * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x16x32<FMT>(bw[j], av, acc[i][j]);
    }
  }
  // lane: block pair j2 -> channels j2 * 32 + kg * 8 + (0..7) of voxel (d0 + dz, h0 + hy, w0 + r16)
  float bs[2][8];
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[j2][e] = a.bias ? a.bias[j2 * 32 + kg * 8 + e] : 0.f;
  u16* yn = a.y + (long long)n * a.D * a.H * a.W * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rb = wv * 4 + i;
    const int d = d0 + (rb >> 2), h = h0 + (rb & 3), w = w0 + r16;
    if (d < a.D && h < a.H && w < a.W) {
      u16* yp = yn + (((long long)d * a.H + h) * a.W + w) * 64 + kg * 8;
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = acc[i][2 * j2 + (e >> 2)][e & 3] + bs[j2][e];
          if (a.act == XH_ACT_LRELU) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
        }
        *reinterpret_cast<uint4*>(yp + j2 * 32) =
            make_uint4(cvt_pack<FMT>(v[0], v[1]), cvt_pack<FMT>(v[2], v[3]), cvt_pack<FMT>(v[4], v[5]), cvt_pack<FMT>(v[6], v[7]));
      }
    }
  }
}

// forward of the LAST conv (Cout = 1): a dot product of K^3 x Cs values per output voxel -- one wave per voxel.
// (D, H, W) are the output extents, the input has (D, H, W) + K - 3
template <int FMT>
__global__ __launch_bounds__(256) void dconv_cout1_kernel(const u16* x, const u16* w, u16* y, int N, int D, int H, int W, int Cs, int K) {
  const int lane = threadIdx.x & 63;
  const long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long V = (long long)D * H * W;
  if (v >= (long long)N * V) return;
  const int n = (int)(v / V);
  long long t = v % V;
  const int ow = (int)(t % W); t /= W;
  const int oh = (int)(t % H); const int od = (int)(t / H);
  const int Dx = D + K - 3, Hx = H + K - 3, Wx = W + K - 3;
  float acc = 0.f;
  for (int tap = 0; tap < K * K * K; ++tap) {
    const int d = od + tap / (K * K) - 1, h = oh + (tap / K) % K - 1, ww = ow + tap % K - 1;
    if ((unsigned)d >= (unsigned)Dx || (unsigned)h >= (unsigned)Hx || (unsigned)ww >= (unsigned)Wx) continue;     // wave-uniform
    const u16* xp = x + ((((long long)n * Dx + d) * Hx + h) * Wx + ww) * Cs;
    const u16* wp = w + (long long)tap * Cs;
    for (int c = lane * 8; c < Cs; c += 512) {
      const uint4 a = *reinterpret_cast<const uint4*>(xp + c);
      const uint4 b = *reinterpret_cast<const uint4*>(wp + c);
      const unsigned ua[4] = {a.x, a.y, a.z, a.w}, ub[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc = fmaf(cvt_lo<FMT>(ua[k]), cvt_lo<FMT>(ub[k]), acc);
        acc = fmaf(cvt_hi<FMT>(ua[k]), cvt_hi<FMT>(ub[k]), acc);
      }
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) y[v] = cvt_out<FMT>(acc);
}

// ------------------------------------------------------------------------------------------------------------------
// parameter-sized helpers: weight packing / gradient unpacking
//  mode 0 forward:   out[tap][co (< Cout)][ci (< CinPad)]        = w[co][ci][tap]      (zero for ci >= Cin)
//  mode 1 data grad: out[tap][ci (< CinPad)][co (< CoutPad)]     = w[co][ci][tap]      (zero rows / columns beyond Cin / Cout)
//  mode 2 row mode (CinPad == 8): out[(kd,kh)][co][kw * 8 + ci]  = w[co][ci][(kd,kh,kw)], zero for kw >= K (K = 3: k >= 24)
__global__ __launch_bounds__(256) void dpack_kernel(const float* w, u16* out, int Cout, int Cin, int CoutPad, int CinPad, int mode, int fmt,
                                                   long long total, int K) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int K3 = K * K * K;
  float v = 0.f;
  if (mode == 0) {
    const int ci = (int)(i % CinPad); long long t = i / CinPad;
    const int co = (int)(t % Cout); const int tap = (int)(t / Cout);
    if (ci < Cin) v = w[((long long)co * Cin + ci) * K3 + tap];
  } else if (mode == 1) {
    const int co = (int)(i % CoutPad); long long t = i / CoutPad;
    const int ci = (int)(t % CinPad); const int tap = (int)(t / CinPad);
    if (ci < Cin && co < Cout) v = w[((long long)co * Cin + ci) * K3 + tap];
  } else {
    const int k = (int)(i % 32); long long t = i / 32;
    const int co = (int)(t % Cout); const int r9 = (int)(t / Cout);
    const int kw = k >> 3, ci = k & 7;
    if (kw < K && ci < Cin) v = w[((long long)co * Cin + ci) * K3 + r9 * K + kw];
  }
  out[i] = fmt ? f2hf(v) : f2bf(v);
}
// Modes 0 / 1 for unpadded multiples of 64 channels: a workgroup moves a (64 inner channels x K^3 taps) tile through LDS.  The element-
// per-thread kernel above reads fp32 weights K^3 floats apart and writes two bytes per thread (the 256 -> 512 layer: 40 us per image, two
// images per step); here the tile is read as whole 128-byte lines (mode 0: one contiguous run) and written as 128-byte runs of one tap.
//   src(j, tap) = w[base + j * jstride + tap],   dst(tap, j) = out[dbase + tap * tstride + j],   j < 64
template <int FMT>
__global__ __launch_bounds__(256) void dpack_tile_kernel(const float* __restrict__ w, u16* __restrict__ out, int K3, long long jstride,
                                                        long long tstride, long long src_x, long long src_y, long long dst_x, long long dst_y) {
  extern __shared__ float s_pk[];                          // [64][K3 + 1]
  const float* src = w + blockIdx.x * src_x + blockIdx.y * src_y;
  u16* dst = out + blockIdx.x * dst_x + blockIdx.y * dst_y;
  const int P = K3 + 1;
  for (int e = threadIdx.x; e < 64 * K3; e += 256) {
    const int j = e / K3, tap = e - j * K3;
    s_pk[j * P + tap] = src[j * jstride + tap];
  }
  __syncthreads();
  for (int it = threadIdx.x; it < K3 * 8; it += 256) {
    const int j8 = it & 7, tap = it >> 3;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = s_pk[(j8 * 8 + q) * P + tap];
    *reinterpret_cast<uint4*>(dst + tap * tstride + j8 * 8) =
        make_uint4(cvt_pack<FMT>(v[0], v[1]), cvt_pack<FMT>(v[2], v[3]), cvt_pack<FMT>(v[4], v[5]), cvt_pack<FMT>(v[6], v[7]));
  }
}
// dw_param[co][ci][tap] += dwp[tap][co (row stride CoutPad rows)][ci (< CinPad)]
__global__ __launch_bounds__(256) void dunpack_kernel(const float* dwp, float* dw, int Cout, int Cin, int CoutPad, int CinPad, long long total,
                                                     int K3) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int tap = (int)(i % K3); long long t = i / K3;
  const int ci = (int)(t % Cin); const int co = (int)(t / Cin);
  dw[i] += dwp[((long long)tap * CoutPad + co) * CinPad + ci];
}

// The same through an LDS tile for unpadded multiples of 64 input channels: workgroup (x, co) reads the 64 x K^3 values
// dwp[tap][co][64 x + j] as 256-byte runs and adds them into the contiguous block dw[co][64 x .. 64 x + 63][0 .. K^3) (the element
// kernel reads one float per thread Cout * Cin floats apart: 48 us for the 256 -> 512 layer).
__global__ __launch_bounds__(256) void dunpack_tile_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin, int K3) {
  extern __shared__ float s_pk[];                          // [64][K3 + 1]
  const int co = blockIdx.y, x = blockIdx.x;
  const int P = K3 + 1;
  const float* src = dwp + (long long)co * Cin + 64 * x;
  for (int e = threadIdx.x; e < 64 * K3; e += 256) {
    const int j = e & 63, tap = e >> 6;
    s_pk[j * P + tap] = src[(long long)tap * Cout * Cin + j];
  }
  __syncthreads();
  float* dst = dw + ((long long)co * Cin + 64 * x) * K3;
  for (int e = threadIdx.x; e < 64 * K3; e += 256) {
    const int j = e / K3, tap = e - j * K3;
    dst[e] += s_pk[j * P + tap];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// channels-last elementwise stages
//  cl_from: NCDHW sources (CA channels of xa, CB of xb) -> [n][v][Cpad] (zero padded);  cl_to: the adjoint (split back)
template <typename T>
__global__ __launch_bounds__(256) void cl_from_kernel(const T* xa, long long xa_bs, int CA, const T* xb, long long xb_bs, int CB, u16* out,
                                                     int Cpad, long long V, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // over N * V
  if (i >= total) return;
  const long long n = i / V, v = i % V;
  for (int c = 0; c < Cpad; ++c) {
    float f = 0.f;
    if (c < CA) f = ldf(xa, n * xa_bs + (long long)c * V + v);
    else if (c < CA + CB) f = ldf(xb, n * xb_bs + (long long)(c - CA) * V + v);
    out[i * Cpad + c] = cvt_out<FmtOf<T>::v>(f);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void cl_to_kernel(const u16* g, int Cpad, T* da, long long da_bs, int CA, T* db, long long db_bs, int CB,
                                                   long long V, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long n = i / V, v = i % V;
  for (int c = 0; c < CA + CB; ++c) {
    const float f = cvt_in<FmtOf<T>::v>(g[i * Cpad + c]);
    if (c < CA) stf(da, n * da_bs + (long long)c * V + v, f);
    else stf(db, n * db_bs + (long long)(c - CA) * V + v, f);
  }
}

// y = leaky(x * sc[n,c] + sh[n,c], slope) on [n][v][C]; one thread = 8 channels of one voxel
template <int FMT>
__global__ __launch_bounds__(256) void cl_affine_act_kernel(const u16* x, u16* y, const float* sc, const float* sh, float slope, int C,
                                                           long long V, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // over N * V * C/8
  if (i >= total) return;
  const int c8 = C >> 3;
  const int cc = (int)(i % c8) * 8;
  const long long n = (i / c8) / V;
  const uint4 t = *reinterpret_cast<const uint4*>(x + i * 8);
  const unsigned u[4] = {t.x, t.y, t.z, t.w};
  unsigned o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* s0 = sc + n * C + cc + 2 * k;
    const float* h0 = sh + n * C + cc + 2 * k;
    const float a0 = leaky(cvt_lo<FMT>(u[k]) * s0[0] + h0[0], slope), a1 = leaky(cvt_hi<FMT>(u[k]) * s0[1] + h0[1], slope);
    o[k] = cvt_pack<FMT>(a0, a1);
  }
  *reinterpret_cast<uint4*>(y + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

// backward through y = leaky(x*sc + sh): g = dy * leaky'(.)
//  MODE 0: red[n][c][0] += sum_v g, red[n][c][1] += sum_v g*x        (reduce only)
//  MODE 1: dx = A[n,c]*g + Cc[n,c]*x + B[n,c]                       (norm backward apply)
//  MODE 2: dx = g and red[n][c][0] += sum_v g                       (plain activation backward + bias gradient)
// grid (blocks over voxels, C/8 groups... ) -- one thread walks voxels of its 8 channels
template <int FMT, int MODE>
__global__ __launch_bounds__(256) void cl_bwd_kernel(const u16* dy, const u16* x, u16* dx, const float* sc, const float* sh, float slope,
                                                    const float* A, const float* B, const float* Cc, double* red, int C, long long V,
                                                    int vchunk) {
  // block: 256 threads = (256 / c8) voxel lanes x c8 channel groups when c8 <= 256
  const int c8 = C >> 3;
  const int cg = threadIdx.x % c8, vl = threadIdx.x / c8, nvl = 256 / c8;
  const int n = blockIdx.y;
  const long long v0 = (long long)blockIdx.x * vchunk, v1 = v0 + vchunk < V ? v0 + vchunk : V;
  const int cc = cg * 8;
  float fs[8], fh[8], fA[8], fB[8], fC[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    fs[k] = sc ? sc[n * C + cc + k] : 1.f; fh[k] = sh ? sh[n * C + cc + k] : 0.f;
    if (MODE == 1) { fA[k] = A[n * C + cc + k]; fB[k] = B[n * C + cc + k]; fC[k] = Cc[n * C + cc + k]; }
  }
  double s0[8], s1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s0[k] = 0.0; s1[k] = 0.0; }
  auto one = [&](long long o, const uint4 td, const uint4 tx) __attribute__((always_inline)) {
      const unsigned ud[4] = {td.x, td.y, td.z, td.w}, ux[4] = {tx.x, tx.y, tx.z, tx.w};
      unsigned oo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float g[2], xv[2] = {cvt_lo<FMT>(ux[k]), cvt_hi<FMT>(ux[k])};
        g[0] = cvt_lo<FMT>(ud[k]); g[1] = cvt_hi<FMT>(ud[k]);
        float r[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int kk = 2 * k + e;
          if ((xv[e] * fs[kk] + fh[kk]) <= 0.f) g[e] *= slope;
          if (MODE == 0) { s0[kk] += (double)g[e]; s1[kk] += (double)g[e] * (double)xv[e]; }
          if (MODE == 1) r[e] = fA[kk] * g[e] + fC[kk] * xv[e] + fB[kk];
          if (MODE == 2) { r[e] = g[e]; s0[kk] += (double)cvt_in<FMT>(cvt_out<FMT>(g[e])); }
        }
        if (MODE != 0) oo[k] = cvt_pack<FMT>(r[0], r[1]);
      }
      if (MODE != 0) *reinterpret_cast<uint4*>(dx + o) = make_uint4(oo[0], oo[1], oo[2], oo[3]);
  };
  if (vl < nvl) {
    // four voxels' loads in flight per thread: the reducing modes run with few workgroups (see xh_cl_act_bwd), so a thread's walk is
    // long and one load pair per trip would be a chain of exposed latencies
    constexpr int U = 4;
    long long v = v0 + vl;
    for (; v + (U - 1) * (long long)nvl < v1; v += U * (long long)nvl) {
      uint4 td[U], tx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long o = ((long long)n * V + v + u * (long long)nvl) * C + cc;
        td[u] = *reinterpret_cast<const uint4*>(dy + o);
        tx[u] = *reinterpret_cast<const uint4*>(x + o);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) one(((long long)n * V + v + u * (long long)nvl) * C + cc, td[u], tx[u]);
    }
    for (; v < v1; v += nvl) {
      const long long o = ((long long)n * V + v) * C + cc;
      one(o, *reinterpret_cast<const uint4*>(dy + o), *reinterpret_cast<const uint4*>(x + o));
    }
  }
  if (MODE != 1) {
    // every thread STORES its eight partial sums ([voxel lane][channel]: nvl * C = 2 048 values whatever C is) and one thread per
    // channel adds the voxel lanes up -- not LDS atomics: floating-point atomics in LDS retire about a lane per two cycles
    // (conv3d_wgrad_mfma.hip), 4 096 fp64 ones per workgroup here
    __shared__ double s_acc[2][2048];
    if (vl < nvl)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s_acc[0][vl * C + cc + k] = s0[k];
        if (MODE == 0) s_acc[1][vl * C + cc + k] = s1[k];
      }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      double t0 = 0.0, t1 = 0.0;
      for (int v = 0; v < nvl; ++v) { t0 += s_acc[0][v * C + c]; if (MODE == 0) t1 += s_acc[1][v * C + c]; }
      atomicAdd(&red[((long long)n * C + c) * 2], t0);
      if (MODE == 0) atomicAdd(&red[((long long)n * C + c) * 2 + 1], t1);
    }
  }
}

// =================================================================================================================
// host entry points
// =================================================================================================================
int g_dconv_kq = 2;        // xh_set_option(5, 1|2): K step of the discriminator's implicit GEMM in 32-channel quarters
static void fill_taps(DTaps* t, int mode, int stride, int parity, int K) {
  // mode 0 forward: all taps, source = j*stride + (t - 1);  mode 1 data gradient: stride 1: source = j + 1 - t;
  // stride 2: destination o = 2j + parity is reached from source s through tap t = o + 1 - 2s: the taps of parity 1 - parity,
  // s = j + (parity + 1 - t) / 2.  K = 3: parity 0 -> tap 1 (source j), parity 1 -> taps 0 (j + 1), 2 (j);
  // K = 4: parity 0 -> taps 1 (j), 3 (j - 1), parity 1 -> taps 0 (j + 1), 2 (j): two per axis for every class
  t->n = 0;
  for (int k = 0; k < K; ++k) {
    if (mode == 1 && stride == 2 && (k & 1) == parity) continue;
    t->t[t->n] = k;
    t->off[t->n] = mode == 0 ? k - 1 : (stride == 1 ? 1 - k : (parity + 1 - k) / 2);
    ++t->n;
  }
}

// Tile shape per launch (xh_set_option(14, mask) switches the choices off one by one for A/B measurements):
//   Cn <= 16          256 x 16   (the 8-channel ends)
//   Cn <= 64          256 x 64   (a 128-wide tile would be half padding: first conv forward, 64 <- 128 data gradient); K step 32
//   < 256 tiles       64 x 64    (256 -> 512 @32^3 forward has 128 tiles of 128 x 128: half of the CUs idle, and one workgroup
//                                per CU cannot hide the gather latency of a K step: 184 us; 64 x 128: 147 us; 64 x 64: 125 us)
//   >= 1024 tiles     256 x 128, 128 x 64 per wave, K step 32: 12 fragment reads per 32 MFMAs instead of 16 (the loop is bound by
//                                LDS reads), 220 VGPRs / 52 KB = two workgroups per CU: 64 -> 128 forward 238 -> 218 us, 128 <- 256
//                                data gradient 123 -> 111 us.  (With K step 64 the tile needs 96 KB = one workgroup per CU: 349 us;
//                                512 x 64 for the 64-channel data gradient: 340 -> 389 us.)
//   else              128 x 128, K step 64
int g_dconv_big = 1024;    // xh_set_option(15, n): 256 x 128 tiles from this many 128 x 128 tiles on
int g_dwh_groups = 32;    // source-block weight gradient: groups of 8 class workgroups per launch (option 26)
int g_dconv_cfg = 0;       // bit 0: one launch per parity class, bit 1: no 256 x 64, bit 2: no small tiles, bit 8: no 256 x 128 tiles, bit 3: 64 x 128 instead of 64 x 64, bit 4: no tap pairs in the weight gradient, bit 5 / 7: 4 / 2 steps in flight for 256 x 16, bit 6: no XCD remap, bit 12: element-per-thread weight pack for every image, bit 10 / 11 / 13: no LDS-halo kernel for the 64 -> 8 data gradient / the 8 -> 64 forward / its weight gradient, bit 14: no source-block kernel for the stride-2 data gradients, bit 15: that kernel on small volumes too, bit 16: one K step of prefetch on the 64 x 64 tiles, bit 17: no source-block kernel for the stride-2 k = 4 weight gradients, bit 19: runtime block extents in the k = 4 source-block data gradient, bit 20: (x, y, z) grid instead of column tile = XCD in the forward convs, bit 21: no 128-channel K steps on the 64 x 64 tiles
template <int FMT>
static void launch_dconv(hipStream_t st, DConvK& a, int N) {
  extern int g_dconv_kq;
  const bool kq2 = !a.rowmode && (a.Cs % 64) == 0 && g_dconv_kq == 2;
  int bm, bn, cfg;
  if (a.Cn <= 16) { bm = 256; bn = 16; cfg = 0; }
  else if (a.Cn <= 64 && !(g_dconv_cfg & 2)) { bm = 256; bn = 64; cfg = 1; }
  else {
    long long tiles = 0;
    for (int k = 0; k < a.ncls; ++k) tiles += cdiv(a.c[k].Jd * a.c[k].Jh * a.c[k].Jw, 128);
    tiles *= (long long)cdiv(a.Cn, 128) * N;
    if (tiles < 256 && !(g_dconv_cfg & 4)) { bm = 64; bn = (g_dconv_cfg & 8) ? 128 : 64; cfg = (g_dconv_cfg & 8) ? 3 : 4; }
    else if (!(g_dconv_cfg & 256) && tiles >= g_dconv_big) { bm = 256; bn = 128; cfg = 5; }
    else { bm = 128; bn = 128; cfg = 2; }
  }
  int t = 0;
  for (int k = 0; k < a.ncls; ++k) { a.c[k].tile0 = t; a.c[k].ntile = cdiv(a.c[k].Jd * a.c[k].Jh * a.c[k].Jw, bm); t += a.c[k].ntile; }
  dim3 grid(t, cdiv(a.Cn, bn), N);
  a.xcd = ((t & 7) == 0 || grid.y * grid.z == 1) && !(g_dconv_cfg & 64);
  if (g_dconv_cfg & 512) a.xcd |= 2;
  a.N = N;
  if (a.ncls == 1 && grid.y >= 2 && (8 % grid.y) == 0 && !a.rowmode && !(g_dconv_cfg & 1048576) &&
      (long long)bn * a.K * a.K * a.K * a.Cs * 2 <= (3ll << 20)) {        // a column tile's weights fit an XCD's L2 with room for the x stream
    const int r = 8 / (int)grid.y;
    a.xcd = 4;
    grid = dim3(8 * cdiv(t * N, r), 1, 1);
  }
  if (cfg == 0) {
    if (g_dconv_cfg & 32) hipLaunchKernelGGL((dconv_cl_kernel<FMT, 1, 1, 1, 4>), grid, dim3(256), 0, st, a);
    else if (g_dconv_cfg & 128) hipLaunchKernelGGL((dconv_cl_kernel<FMT, 1, 1, 1, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dconv_cl_kernel<FMT, 1, 1>), grid, dim3(256), 0, st, a);
  }
  else if (cfg == 1) hipLaunchKernelGGL((dconv_cl_kernel<FMT, 1, 4>), grid, dim3(256), 0, st, a);
  else if (cfg == 4) {
    // few workgroups (< 256 tiles of 128 x 128), each a chain of hundreds of K steps that one step of prefetch does not cover: TWO
    // K steps of loads in flight (round 5; 256 -> 512 @31^3 forward 219.7 -> 188.7 us, four steps 186.5; bit 16: one step as before)
    // ... and K steps of 128 channels where the layer has them: a wave's 64 x 16 share is 4 MFMAs per 32 channels against ~100 scalar /
    // vector / LDS instructions of step overhead (round 5: the 256 -> 512 launch was bound by their issue)
    // (one resident round of workgroups only: two samples of 256 -> 512 @15^3 are 848 workgroups and ran 297 against 266 us)
    if (kq2 && (a.Cs % 128) == 0 && (long long)grid.x * grid.y * grid.z <= 512 && !(g_dconv_cfg & 2097152))
      hipLaunchKernelGGL((dconv_cl_kernel<FMT, 4, 1, 4, 2>), grid, dim3(256), 0, st, a);
    else if (kq2 && !(g_dconv_cfg & 65536)) hipLaunchKernelGGL((dconv_cl_kernel<FMT, 4, 1, 2, 2>), grid, dim3(256), 0, st, a);
    else if (kq2) hipLaunchKernelGGL((dconv_cl_kernel<FMT, 4, 1, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dconv_cl_kernel<FMT, 4, 1>), grid, dim3(256), 0, st, a);
  } else if (cfg == 3) {
    if (kq2) hipLaunchKernelGGL((dconv_cl_kernel<FMT, 4, 2, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dconv_cl_kernel<FMT, 4, 2>), grid, dim3(256), 0, st, a);
  } else {
    if (cfg == 5) hipLaunchKernelGGL((dconv_cl_kernel<FMT, 2, 4, 1, 1, 8>), grid, dim3(256), 0, st, a);
    else if (kq2) hipLaunchKernelGGL((dconv_cl_kernel<FMT, 2, 4, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dconv_cl_kernel<FMT, 2, 4>), grid, dim3(256), 0, st, a);
  }
}

// Channels-last convolution of the discriminator, kernel size ks (3 or 4), padding 1.  mode 0: forward (x: [N][Di..][Cs] ->
// y: [N][Do..][Cn], Do = (Di + 2 - ks)/stride + 1); mode 1: data gradient (x = dY [N][Di..][Cs = Cout], y = dX [N][Do..][Cn = Cin_pad],
// Di = (Do + 2 - ks)/stride + 1).  w: weights packed by xh_dconv_pack (mode 0/2 for forward, 1 for the data gradient).
// bias/red optional (forward).
extern "C" int xh_dconv_cl(void* stream, int dtype, int mode, int stride, int ks, const void* x, const void* w, const float* bias, void* y,
                           double* red, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int Cs, int Cn, int act, float slope,
                           const void* mask) {
  if (!x || !w || !y || N <= 0 || N > 65535 || Cs <= 0 || Cn <= 0) return XH_ERR_ARG;
  if (dtype != XH_BF16 && dtype != XH_F16) return XH_ERR_DTYPE;
  if (!(stride == 1 || stride == 2) || (mode != 0 && mode != 1) || (ks != 3 && ks != 4)) return XH_ERR_ARG;
  {                                                       // extents must be those of a padding-1 convolution
    const int bi[3] = {mode == 0 ? Di : Do, mode == 0 ? Hi : Ho, mode == 0 ? Wi : Wo};
    const int sm[3] = {mode == 0 ? Do : Di, mode == 0 ? Ho : Hi, mode == 0 ? Wo : Wi};
    for (int k = 0; k < 3; ++k)
      if (bi[k] + 2 < ks || sm[k] != (bi[k] + 2 - ks) / stride + 1) return XH_ERR_ARG;
  }
  const bool rowmode = mode == 0 && Cs == 8;
  if (!rowmode && (Cs % 32)) return XH_ERR_ARG;
  // the staging keeps 32-bit element offsets inside one sample of x and inside w
  if ((long long)Di * Hi * Wi * Cs >= (1ll << 31) || (long long)ks * ks * ks * Cn * (rowmode ? 32 : Cs) >= (1ll << 31)) return XH_ERR_ARG;
  if (rowmode && stride != 1) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  DConvK a;
  a.x = (const u16*)x; a.w = (const u16*)w; a.bias = bias; a.y = (u16*)y; a.red = red; a.mask = (const u16*)mask;
  a.N = N; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.Cs = Cs; a.Cn = Cn;
  a.rowmode = rowmode ? 1 : 0; a.act = act; a.slope = slope; a.K = ks;
  a.Kc = rowmode ? 32 : Cs;
  a.wtap_stride = Cn * a.Kc;
  if (mode == 0 && Cn == 1 && stride == 1 && !bias && !red && !mask && act == XH_ACT_NONE && !rowmode && (Cs % 8) == 0) {
    const long long waves = (long long)N * Do * Ho * Wo;
    const unsigned nb = (unsigned)((waves + 3) / 4);
    if (dtype == XH_F16) hipLaunchKernelGGL(dconv_cout1_kernel<1>, dim3(nb), dim3(256), 0, st, (const u16*)x, (const u16*)w, (u16*)y, N, Do, Ho, Wo, Cs, ks);
    else hipLaunchKernelGGL(dconv_cout1_kernel<0>, dim3(nb), dim3(256), 0, st, (const u16*)x, (const u16*)w, (u16*)y, N, Do, Ho, Wo, Cs, ks);
    return xh_launch_status();
  }
  if (rowmode && Cn == 64 && !red && !mask && (act == XH_ACT_NONE || act == XH_ACT_LRELU) && !(g_dconv_cfg & 2048)) {
    DFw8K k;
    k.x = (const u16*)x; k.w = (const u16*)w; k.bias = bias; k.y = (u16*)y;
    k.N = N; k.D = Do; k.H = Ho; k.W = Wo; k.act = act; k.slope = slope;
    k.td = cdiv(Do, 4); k.th = cdiv(Ho, 4); k.tw = cdiv(Wo, 16);
    const long long nb = (long long)N * k.td * k.th * k.tw;
    if (nb < (1LL << 31)) {
      if (ks == 3) {
        if (dtype == XH_F16) hipLaunchKernelGGL((dconv_fwd_c8_kernel<1, 3>), dim3((unsigned)nb), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((dconv_fwd_c8_kernel<0, 3>), dim3((unsigned)nb), dim3(256), 0, st, k);
      } else {
        if (dtype == XH_F16) hipLaunchKernelGGL((dconv_fwd_c8_kernel<1, 4>), dim3((unsigned)nb), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((dconv_fwd_c8_kernel<0, 4>), dim3((unsigned)nb), dim3(256), 0, st, k);
      }
      return xh_launch_status();
    }
  }
  if (mode == 1 && stride == 1 && Cs == 64 && Cn == 8 && !bias && !red && !mask && act == XH_ACT_NONE && !(g_dconv_cfg & 1024)) {
    DDg8K k;
    k.g = (const u16*)x; k.w = (const u16*)w; k.dx = (u16*)y;
    k.N = N; k.D = Do; k.H = Ho; k.W = Wo;
    k.td = cdiv(Do, 4); k.th = cdiv(Ho, 4); k.tw = cdiv(Wo, 16);
    const long long nb = (long long)N * k.td * k.th * k.tw;
    if (nb < (1LL << 31)) {
      if (ks == 3) {
        if (dtype == XH_F16) hipLaunchKernelGGL((dconv_dgrad_c8_kernel<1, 3>), dim3((unsigned)nb), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((dconv_dgrad_c8_kernel<0, 3>), dim3((unsigned)nb), dim3(256), 0, st, k);
      } else {
        if (dtype == XH_F16) hipLaunchKernelGGL((dconv_dgrad_c8_kernel<1, 4>), dim3((unsigned)nb), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((dconv_dgrad_c8_kernel<0, 4>), dim3((unsigned)nb), dim3(256), 0, st, k);
      }
      return xh_launch_status();
    }
  }
  const int classes = (mode == 1 && stride == 2) ? 8 : 1;
  a.omul = classes == 8 ? 2 : 1;
  a.smul = mode == 0 ? stride : 1;
  // heaviest class first: parity 1 on an axis = 2 taps on it
  static const int order[8] = {7, 6, 5, 3, 4, 2, 1, 0};
  DClass all[8];
  int nc = 0;
  for (int k = 0; k < classes; ++k) {
    const int cls = classes == 8 ? order[k] : 0;
    const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
    DClass& c = all[nc];
    c.pd = pd; c.ph = ph; c.pw = pw; c.tile0 = 0;
    c.Jd = classes == 8 ? (Do - pd + 1) / 2 : Do;
    c.Jh = classes == 8 ? (Ho - ph + 1) / 2 : Ho;
    c.Jw = classes == 8 ? (Wo - pw + 1) / 2 : Wo;
    if (c.Jd <= 0 || c.Jh <= 0 || c.Jw <= 0) continue;
    fill_taps(&c.td, mode, stride, pd, ks); fill_taps(&c.th, mode, stride, ph, ks); fill_taps(&c.tw, mode, stride, pw, ks);
    ++nc;
  }
  // measured (tools/microbench_disc.py, ks = 4, bf16): 64 <- 128 @127^3 691 -> 451 us, 128 <- 256 @63^3 185 -> 178 us, 256 <- 512 @31^3
  // 97 -> 114 us (256 workgroups of 16 K slices each: the gather kernel's 64 x 64 tiles fill the chip better there)
  if (mode == 1 && stride == 2 && nc > 0 && (Cs % 32) == 0 && (Cn % 64) == 0 && !(g_dconv_cfg & 16384) &&
      ((long long)Do * Ho * Wo >= 200000 || (g_dconv_cfg & 32768)) &&
      (long long)Di * Hi * Wi * Cs * 2 < (1ll << 31)) {        // the kernel's source offsets and buffer resource are 32-bit bytes
    // the source-block kernel (dconv_dgrad_halo_kernel): tiles of 8 x 8 x 8 destination voxels per parity class
    a.ncls = nc;
    int t = 0;
    for (int k = 0; k < nc; ++k) {
      a.c[k] = all[k];
      a.c[k].tile0 = t;
      a.c[k].ntile = cdiv(all[k].Jd, 8) * cdiv(all[k].Jh, 8) * cdiv(all[k].Jw, 8);
      t += a.c[k].ntile;
    }
    dim3 grid(t, Cn / 64, N);
    a.xcd = ((t & 7) == 0 || grid.y * grid.z == 1) && !(g_dconv_cfg & 64);
    const size_t shm = 729 * 64 + 8 * 64 * 64;
    static bool attr_done[XH_MAX_DEV] = {};
    if (xh_attr_needed(attr_done)) {
      (void)hipFuncSetAttribute((const void*)dconv_dgrad_halo_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      (void)hipFuncSetAttribute((const void*)dconv_dgrad_halo_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      (void)hipFuncSetAttribute((const void*)dconv_dgrad_halo_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      (void)hipFuncSetAttribute((const void*)dconv_dgrad_halo_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    }
    if (ks == 4 && nc == 8 && !(g_dconv_cfg & 524288)) {
      if (dtype == XH_F16) hipLaunchKernelGGL((dconv_dgrad_halo_kernel<1, true>), grid, dim3(512), shm, st, a);
      else hipLaunchKernelGGL((dconv_dgrad_halo_kernel<0, true>), grid, dim3(512), shm, st, a);
    } else {
      if (dtype == XH_F16) hipLaunchKernelGGL((dconv_dgrad_halo_kernel<1, false>), grid, dim3(512), shm, st, a);
      else hipLaunchKernelGGL((dconv_dgrad_halo_kernel<0, false>), grid, dim3(512), shm, st, a);
    }
    return xh_launch_status();
  }
  if (g_dconv_cfg & 1) {
    for (int k = 0; k < nc; ++k) {
      a.ncls = 1; a.c[0] = all[k];
      if (dtype == XH_F16) launch_dconv<1>(st, a, N); else launch_dconv<0>(st, a, N);
    }
  } else if (nc > 0) {
    a.ncls = nc;
    for (int k = 0; k < nc; ++k) a.c[k] = all[k];
    if (dtype == XH_F16) launch_dconv<1>(st, a, N); else launch_dconv<0>(st, a, N);
  }
  return xh_launch_status();
}

// dwp[tap][Cn][Cs] (fp32, caller zeroes) += sum over output voxels of dY[m][cn] * X[src][cs];  ks^3 taps, padding 1
extern "C" int xh_dconv_wgrad_cl(void* stream, int dtype, int stride, int ks, const void* x, const void* dy, float* dwp, int N, int Di, int Hi,
                                 int Wi, int Do, int Ho, int Wo, int Cs, int Cn) {
  if (!x || !dy || !dwp || N <= 0 || (Cs % 8) || (Cn % 8) || (ks != 3 && ks != 4) || (stride != 1 && stride != 2)) return XH_ERR_ARG;
  if (dtype != XH_BF16 && dtype != XH_F16) return XH_ERR_DTYPE;
  if (Do != (Di + 2 - ks) / stride + 1 || Ho != (Hi + 2 - ks) / stride + 1 || Wo != (Wi + 2 - ks) / stride + 1) return XH_ERR_ARG;
  DWgK a;
  a.K = ks; a.ntap = ks * ks * ks; a.ntz = a.ntap;
  a.x = (const u16*)x; a.dy = (const u16*)dy; a.dw = dwp;
  a.N = N; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.Cs = Cs; a.Cn = Cn; a.stride = stride;
  a.M = (long long)N * Do * Ho * Wo;
  // the staging of dwgrad_cl_kernel keeps 32-bit element offsets into x and dY
  if ((long long)N * Di * Hi * Wi * Cs >= (1ll << 31) || a.M * Cn >= (1ll << 31)) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (Cs == 8 && Cn == 64 && stride == 1 && !(g_dconv_cfg & 8192)) {       // first conv of the network: LDS-halo kernel
    DWg8HK k;
    k.x = (const u16*)x; k.dy = (const u16*)dy; k.dw = dwp;
    k.N = N; k.D = Do; k.H = Ho; k.W = Wo;
    k.td = cdiv(Do, 4); k.th = cdiv(Ho, 4); k.tw = cdiv(Wo, 16);
    const long long nt = (long long)N * k.td * k.th * k.tw;
    if (nt < (1LL << 30)) {
      k.ntile = (int)nt;
      const unsigned nwg = (unsigned)(nt < 512 ? nt : 512);
      if (ks == 3) {
        if (dtype == XH_F16) hipLaunchKernelGGL((dwgrad_c8_halo_kernel<1, 3>), dim3(nwg), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((dwgrad_c8_halo_kernel<0, 3>), dim3(nwg), dim3(256), 0, st, k);
      } else {                                            // the 64 taps in two halves (grid y)
        if (dtype == XH_F16) hipLaunchKernelGGL((dwgrad_c8_halo_kernel<1, 4>), dim3(nwg, 2), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((dwgrad_c8_halo_kernel<0, 4>), dim3(nwg, 2), dim3(256), 0, st, k);
      }
      return xh_launch_status();
    }
  }
  if (stride == 2 && ks == 4 && (Cs % 64) == 0 && (Cn % 128) == 0 && !(g_dconv_cfg & 131072) &&
      (long long)Di * Hi * Wi * Cs * 2 < (1ll << 31) && (long long)Do * Ho * Wo * Cn * 2 < (1ll << 31)) {   // source-block kernel (32-bit byte offsets per sample)
    DWgHK k;
    k.x = (const u16*)x; k.dy = (const u16*)dy; k.dw = dwp;
    k.N = N; k.Di = Di; k.Hi = Hi; k.Wi = Wi; k.Do = Do; k.Ho = Ho; k.Wo = Wo; k.Cs = Cs; k.Cn = Cn;
    k.nbz = cdiv(Do, 4); k.nby = cdiv(Ho, 4); k.nbx = cdiv(Wo, 4);
    const long long nblk = (long long)N * k.nbz * k.nby * k.nbx;
    k.ncs = Cs / 64; k.ncn = Cn / 128;
    const int tiles = k.ncs * k.ncn;
    if (nblk < (1LL << 30) && tiles <= 4096) {
      k.nblk = (int)nblk;
      int nsplit = tiles >= g_dwh_groups ? 1 : g_dwh_groups / tiles;       // 8 classes x groups workgroups, one per CU
      if (nsplit > k.nblk) nsplit = k.nblk;
      k.nsplit = nsplit; k.ngroup = nsplit * tiles;
      const size_t shm = 2 * (64 * 256 + 160 * 128);
      static bool attr_done[XH_MAX_DEV] = {};
      if (xh_attr_needed(attr_done)) {
        (void)hipFuncSetAttribute((const void*)dwgrad_halo_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        (void)hipFuncSetAttribute((const void*)dwgrad_halo_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      }
      if (dtype == XH_F16) hipLaunchKernelGGL((dwgrad_halo_kernel<1>), dim3(8 * k.ngroup), dim3(512), shm, st, k);
      else hipLaunchKernelGGL((dwgrad_halo_kernel<0>), dim3(8 * k.ngroup), dim3(512), shm, st, k);
      return xh_launch_status();
    }
  }
  if (Cs == 8 && ks == 3) {                               // first conv: the taps ride on the N axis
    const int tiles = cdiv(Cn, 64);
    int msplit = cdiv(1024, tiles);
    const long long max_split = a.M / 256 > 0 ? a.M / 256 : 1;
    if (msplit > max_split) msplit = (int)max_split;
    a.msplit = msplit;
    dim3 grid(tiles, msplit);
    if (dtype == XH_F16) hipLaunchKernelGGL(dwgrad_c8_kernel<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dwgrad_c8_kernel<0>, grid, dim3(256), 0, st, a);
    return xh_launch_status();
  }
  const bool narrow = Cs <= 16;
  const bool pair = Cs == 64 && !(g_dconv_cfg & 16);
  const int ntz = pair ? (a.ntap + 1) / 2 : a.ntap;
  a.ntz = ntz;
  const int bn = narrow ? 16 : 128, bm = narrow ? 64 : 128;
  const int tiles = cdiv(Cs, bn) * cdiv(Cn, bm);
  int msplit = cdiv(1024, tiles * ntz);
  const long long max_split = a.M / 512 > 0 ? a.M / 512 : 1;
  if (msplit > max_split) msplit = (int)max_split;
  if (msplit < 1) msplit = 1;
  a.msplit = msplit;
  dim3 grid(cdiv(Cs, bn), cdiv(Cn, bm), ntz * msplit);
  if (pair) {
    if (dtype == XH_F16) hipLaunchKernelGGL((dwgrad_cl_kernel<1, 2, 4, 4, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwgrad_cl_kernel<0, 2, 4, 4, true>), grid, dim3(256), 0, st, a);
  } else if (narrow) {
    if (dtype == XH_F16) hipLaunchKernelGGL((dwgrad_cl_kernel<1, 1, 1, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwgrad_cl_kernel<0, 1, 1, 1>), grid, dim3(256), 0, st, a);
  } else {
    if (dtype == XH_F16) hipLaunchKernelGGL((dwgrad_cl_kernel<1, 2, 4, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwgrad_cl_kernel<0, 2, 4, 4>), grid, dim3(256), 0, st, a);
  }
  return xh_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// The head: Conv3d(C, 1, ks, stride 1, padding 1, bias=False) (RA_HVED.py:223) as three reduction kernels (xlstm_hved.h: xh_dlast_*).
// A wave owns 4 consecutive voxels of a row; lane l holds channels 8 l .. 8 l + 7 of each 512-channel chunk (one 16-byte load per
// voxel and chunk); fp32 arithmetic on 16-bit operands.
template <int FMT> __device__ __forceinline__ void dl_cvt8(const uint4& u, float (&o)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { o[2 * k] = cvt_lo<FMT>(w[k]); o[2 * k + 1] = cvt_hi<FMT>(w[k]); }
}
constexpr int DL_OW = 4;
// Branch-free bodies with K a template argument, walking the K^2 (kd, kh) rows TWO at a time in a rolled loop: a row outside the
// volume reads a clamped address and is multiplied by zero, so the loads of two rows (22 pieces of 16 bytes) are in flight together.
// (With `continue` on the bounds every row was one exposed memory latency: 26 us for 16 steps; fully unrolled the compiler hoisted
// every load and the kernels took 256 registers, the data gradient spilling 539.)
template <int FMT, int K>
__global__ __launch_bounds__(256) void dlast_fwd_kernel(const u16* __restrict__ x, const u16* __restrict__ wp, u16* __restrict__ y, int N, int Di,
                                                       int Hi, int Wi, int Do, int Ho, int Wo, int C) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int gw = (Wo + DL_OW - 1) / DL_OW;
  if (wave >= (long long)N * Do * Ho * gw) return;
  const int owg = (int)(wave % gw); long long t = wave / gw;
  const int oh = (int)(t % Ho); t /= Ho;
  const int od = (int)(t % Do); const int n = (int)(t / Do);
  const int ow0 = owg * DL_OW;
  constexpr int NS = DL_OW + K - 1;
  float acc[DL_OW] = {0.f, 0.f, 0.f, 0.f};
  float cm[NS];                                           // column masks
#pragma unroll
  for (int s = 0; s < NS; ++s) cm[s] = (unsigned)(ow0 - 1 + s) < (unsigned)Wi ? 1.f : 0.f;
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
#pragma unroll 1
    for (int t2 = 0; t2 < (K * K + 1) / 2; ++t2) {
      uint4 xr[2][NS], wr[2][K];
      float rm[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tt = min(2 * t2 + h, K * K - 1), kd = tt / K, kh = tt - kd * K;
        const int id = od - 1 + kd, ih = oh - 1 + kh;
        rm[h] = (2 * t2 + h < K * K && (unsigned)id < (unsigned)Di && (unsigned)ih < (unsigned)Hi) ? 1.f : 0.f;
        const u16* row = x + (((long long)n * Di + min(max(id, 0), Di - 1)) * Hi + min(max(ih, 0), Hi - 1)) * (long long)Wi * C + c0;
#pragma unroll
        for (int s = 0; s < NS; ++s) xr[h][s] = *reinterpret_cast<const uint4*>(row + (long long)min(max(ow0 - 1 + s, 0), Wi - 1) * C);
#pragma unroll
        for (int kw = 0; kw < K; ++kw) wr[h][kw] = *reinterpret_cast<const uint4*>(wp + (long long)(tt * K + kw) * C + c0);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float xs[NS][8];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          dl_cvt8<FMT>(xr[h][s], xs[s]);
          const float m = cm[s] * rm[h];
#pragma unroll
          for (int e = 0; e < 8; ++e) xs[s][e] *= m;
        }
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          float wv[8];
          dl_cvt8<FMT>(wr[h][kw], wv);
#pragma unroll
          for (int o = 0; o < DL_OW; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o] = fmaf(xs[o + kw][e], wv[e], acc[o]);
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < DL_OW; ++o) {
    const float s = wave_sum(acc[o]);
    if (lane == 0 && ow0 + o < Wo) y[(((long long)n * Do + od) * Ho + oh) * Wo + ow0 + o] = cvt_out<FMT>(s);
  }
}
template <int FMT, int K>
__global__ __launch_bounds__(256) void dlast_dgrad_kernel(const u16* __restrict__ dy, const u16* __restrict__ wp, u16* __restrict__ dx, int N,
                                                         int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int gw = (Wi + DL_OW - 1) / DL_OW;
  if (wave >= (long long)N * Di * Hi * gw) return;
  const int iwg = (int)(wave % gw); long long t = wave / gw;
  const int ih = (int)(t % Hi); t /= Hi;
  const int id = (int)(t % Di); const int n = (int)(t / Di);
  const int iw0 = iwg * DL_OW;
  constexpr int NS = DL_OW + K - 1;
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
    float acc[DL_OW][8];
#pragma unroll
    for (int v = 0; v < DL_OW; ++v)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[v][e] = 0.f;
#pragma unroll 1
    for (int t2 = 0; t2 < (K * K + 1) / 2; ++t2) {
      uint4 wr[2][K];
      float g[2][NS];                                     // dY row (od, oh), columns iw0 + 1 - (K - 1) .. iw0 + DL_OW (one address per wave: broadcast loads)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tt = min(2 * t2 + h, K * K - 1), kd = tt / K, kh = tt - kd * K;
        const int od = id + 1 - kd, oh = ih + 1 - kh;
        const bool rok = 2 * t2 + h < K * K && (unsigned)od < (unsigned)Do && (unsigned)oh < (unsigned)Ho;
        const u16* drow = dy + (((long long)n * Do + min(max(od, 0), Do - 1)) * Ho + min(max(oh, 0), Ho - 1)) * Wo;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int ow = iw0 + 1 - (K - 1) + s;
          const float v = cvt_in<FMT>(drow[min(max(ow, 0), Wo - 1)]);
          g[h][s] = rok && (unsigned)ow < (unsigned)Wo ? v : 0.f;
        }
#pragma unroll
        for (int kw = 0; kw < K; ++kw) wr[h][kw] = *reinterpret_cast<const uint4*>(wp + (long long)(tt * K + kw) * C + c0);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          float wv[8];
          dl_cvt8<FMT>(wr[h][kw], wv);
#pragma unroll
          for (int v = 0; v < DL_OW; ++v) {
            const float gv = g[h][v + (K - 1) - kw];                  // column ow = iw0 + v + 1 - kw
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[v][e] = fmaf(gv, wv[e], acc[v][e]);
          }
        }
    }
#pragma unroll
    for (int v = 0; v < DL_OW; ++v) {
      if (iw0 + v >= Wi) break;
      uint4 o;
      o.x = cvt_pack<FMT>(acc[v][0], acc[v][1]); o.y = cvt_pack<FMT>(acc[v][2], acc[v][3]);
      o.z = cvt_pack<FMT>(acc[v][4], acc[v][5]); o.w = cvt_pack<FMT>(acc[v][6], acc[v][7]);
      *reinterpret_cast<uint4*>(dx + ((((long long)n * Di + id) * Hi + ih) * Wi + iw0 + v) * (long long)C + c0) = o;
    }
  }
}
// grid (ks^3 taps, splits of the output ROWS); the 16 waves of a workgroup take interleaved rows (n, od, oh) and walk each row in
// groups of 7 voxels whose loads are issued together (unconditional: a voxel outside the input reads a clamped address and is
// multiplied by zero -- 86 dependent load -> fma steps per wave made the first version 140 us); the waves are summed in LDS and one
// lane per channel adds into the parameter gradient
constexpr int DLW_WAVES = 16, DLW_U = 7;
template <int FMT>
__global__ __launch_bounds__(64 * DLW_WAVES) void dlast_wgrad_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, float* __restrict__ dw,
                                                                    float scale, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int K) {
  __shared__ float s_acc[DLW_WAVES - 1][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tap = blockIdx.x, kw = tap % K, kh = (tap / K) % K, kd = tap / (K * K);
  const int R = N * Do * Ho;
  const int per = (R + gridDim.y - 1) / gridDim.y;
  const int r0 = per * blockIdx.y, r1 = r0 + per < R ? r0 + per : R;
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + wv; r < r1; r += DLW_WAVES) {
      const int oh = r % Ho, t = r / Ho;
      const int od = t % Do, n = t / Do;
      const int id = od - 1 + kd, ih = oh - 1 + kh;
      if ((unsigned)id >= (unsigned)Di || (unsigned)ih >= (unsigned)Hi) continue;      // (wave-uniform)
      const u16* xrow = x + (((long long)n * Di + id) * Hi + ih) * (long long)Wi * C + c0;
      const u16* drow = dy + (long long)r * Wo;
      for (int ow0 = 0; ow0 < Wo; ow0 += DLW_U) {
        uint4 xv[DLW_U];
        float g[DLW_U];
#pragma unroll
        for (int u = 0; u < DLW_U; ++u) {
          const int ow = ow0 + u, iw = ow - 1 + kw;
          const bool ok = ow < Wo && (unsigned)iw < (unsigned)Wi;
          xv[u] = *reinterpret_cast<const uint4*>(xrow + (long long)min(max(iw, 0), Wi - 1) * C);
          g[u] = ok ? cvt_in<FMT>(drow[min(ow, Wo - 1)]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < DLW_U; ++u) {
          float xf[8];
          dl_cvt8<FMT>(xv[u], xf);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] = fmaf(g[u], xf[e], acc[e]);
        }
      }
    }
    __syncthreads();
    if (wv > 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) s_acc[wv - 1][lane * 8 + e] = acc[e];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s = acc[e];
#pragma unroll
        for (int w = 0; w < DLW_WAVES - 1; ++w) s += s_acc[w][lane * 8 + e];
        atomicAdd(&dw[(long long)(c0 + e) * (K * K * K) + tap], s * scale);
      }
    }
  }
}
static bool dlast_ok(int dtype, int ks, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C) {
  if ((dtype != XH_BF16 && dtype != XH_F16) || (ks != 3 && ks != 4) || N <= 0 || C <= 0 || C % 512) return false;
  if (Do != Di + 2 - ks + 1 || Ho != Hi + 2 - ks + 1 || Wo != Wi + 2 - ks + 1 || Do <= 0 || Ho <= 0 || Wo <= 0) return false;
  return (long long)N * Di * Hi * Wi * C < (1ll << 40);
}
extern "C" int xh_dlast_fwd(void* stream, int dtype, int ks, const void* x, const void* wp, void* y, int N, int Di, int Hi, int Wi, int Do,
                            int Ho, int Wo, int C) {
  if (!x || !wp || !y || !dlast_ok(dtype, ks, N, Di, Hi, Wi, Do, Ho, Wo, C)) return XH_ERR_ARG;
  const long long waves = (long long)N * Do * Ho * cdiv(Wo, DL_OW);
  const dim3 grid((unsigned)((waves + 3) / 4));
#define DLF(F, K) hipLaunchKernelGGL((dlast_fwd_kernel<F, K>), grid, dim3(256), 0, (hipStream_t)stream, (const u16*)x, (const u16*)wp, (u16*)y, N, Di, Hi, Wi, Do, Ho, Wo, C)
  if (dtype == XH_F16) { if (ks == 4) DLF(1, 4); else DLF(1, 3); }
  else { if (ks == 4) DLF(0, 4); else DLF(0, 3); }
#undef DLF
  return xh_launch_status();
}
extern "C" int xh_dlast_dgrad(void* stream, int dtype, int ks, const void* dy, const void* wp, void* dx, int N, int Di, int Hi, int Wi, int Do,
                              int Ho, int Wo, int C) {
  if (!dy || !wp || !dx || !dlast_ok(dtype, ks, N, Di, Hi, Wi, Do, Ho, Wo, C)) return XH_ERR_ARG;
  const long long waves = (long long)N * Di * Hi * cdiv(Wi, DL_OW);
  const dim3 grid((unsigned)((waves + 3) / 4));
#define DLD(F, K) hipLaunchKernelGGL((dlast_dgrad_kernel<F, K>), grid, dim3(256), 0, (hipStream_t)stream, (const u16*)dy, (const u16*)wp, (u16*)dx, N, Di, Hi, Wi, Do, Ho, Wo, C)
  if (dtype == XH_F16) { if (ks == 4) DLD(1, 4); else DLD(1, 3); }
  else { if (ks == 4) DLD(0, 4); else DLD(0, 3); }
#undef DLD
  return xh_launch_status();
}
extern "C" int xh_dlast_wgrad(void* stream, int dtype, int ks, const void* x, const void* dy, float* dw, float scale, int N, int Di, int Hi,
                              int Wi, int Do, int Ho, int Wo, int C) {
  if (!x || !dy || !dw || !dlast_ok(dtype, ks, N, Di, Hi, Wi, Do, Ho, Wo, C)) return XH_ERR_ARG;
  const int R = N * Do * Ho;
  int splits = R / (2 * DLW_WAVES) > 0 ? R / (2 * DLW_WAVES) : 1;      // >= 2 rows per wave; ks^3 x splits workgroups of 16 waves
  if (splits > 8) splits = 8;
  const dim3 grid(ks * ks * ks, splits);
  if (dtype == XH_F16) hipLaunchKernelGGL(dlast_wgrad_kernel<1>, grid, dim3(64 * DLW_WAVES), 0, (hipStream_t)stream, (const u16*)x, (const u16*)dy, dw, scale, N, Di, Hi, Wi, Do, Ho, Wo, C, ks);
  else hipLaunchKernelGGL(dlast_wgrad_kernel<0>, grid, dim3(64 * DLW_WAVES), 0, (hipStream_t)stream, (const u16*)x, (const u16*)dy, dw, scale, N, Di, Hi, Wi, Do, Ho, Wo, C, ks);
  return xh_launch_status();
}

extern "C" int xh_dconv_pack(void* stream, int dtype, int mode, int ks, const float* w, void* out, int Cout, int Cin, int CoutPad, int CinPad) {
  if (!w || !out || mode < 0 || mode > 2 || CoutPad < Cout || CinPad < Cin || (ks != 3 && ks != 4)) return XH_ERR_ARG;
  if (mode == 2 && CinPad != 8) return XH_ERR_ARG;
  const long long k3 = (long long)ks * ks * ks;
  const long long total = mode == 2 ? (long long)ks * ks * Cout * 32 : (mode == 0 ? k3 * Cout * CinPad : k3 * CinPad * CoutPad);
  if (mode != 2 && CinPad == Cin && CoutPad == Cout && !(g_dconv_cfg & 4096) && (mode == 0 ? Cin % 64 == 0 : Cout % 64 == 0)) {
    // (xh_set_option(14, 4096): the element-per-thread kernel for every image, A/B and tests)
    const size_t shm = (size_t)64 * (k3 + 1) * sizeof(float);
    const dim3 grid(mode == 0 ? Cin / 64 : Cout / 64, mode == 0 ? Cout : Cin);
    // mode 0: tile (co = y, ci = 64 x + j): src = (co Cin + ci) K3, dst = tap Cout Cin + co Cin + ci
    // mode 1: tile (ci = y, co = 64 x + j): src = (co Cin + ci) K3, dst = tap Cin Cout + ci Cout + co
    const long long jstride = mode == 0 ? k3 : (long long)Cin * k3, tstride = (long long)Cout * Cin;
    const long long src_x = mode == 0 ? 64 * k3 : 64ll * Cin * k3, src_y = mode == 0 ? (long long)Cin * k3 : k3;
    const long long dst_x = 64, dst_y = mode == 0 ? Cin : Cout;
    if (dtype == XH_F16) hipLaunchKernelGGL(dpack_tile_kernel<1>, grid, dim3(256), shm, (hipStream_t)stream, w, (u16*)out, (int)k3, jstride, tstride, src_x, src_y, dst_x, dst_y);
    else hipLaunchKernelGGL(dpack_tile_kernel<0>, grid, dim3(256), shm, (hipStream_t)stream, w, (u16*)out, (int)k3, jstride, tstride, src_x, src_y, dst_x, dst_y);
    return xh_launch_status();
  }
  hipLaunchKernelGGL(dpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (u16*)out, Cout, Cin, CoutPad,
                     CinPad, mode, dtype == XH_F16 ? 1 : 0, total, ks);
  return xh_launch_status();
}
extern "C" int xh_dconv_unpack_grad(void* stream, int ks, const float* dwp, float* dw, int Cout, int Cin, int CoutPad, int CinPad) {
  if (!dwp || !dw || CoutPad < Cout || CinPad < Cin || (ks != 3 && ks != 4)) return XH_ERR_ARG;
  const int k3 = ks * ks * ks;
  const long long total = (long long)k3 * Cout * Cin;
  if (CinPad == Cin && CoutPad == Cout && Cin % 64 == 0 && !(g_dconv_cfg & 4096)) {
    hipLaunchKernelGGL(dunpack_tile_kernel, dim3(Cin / 64, Cout), dim3(256), (size_t)64 * (k3 + 1) * sizeof(float), (hipStream_t)stream, dwp, dw,
                       Cout, Cin, k3);
    return xh_launch_status();
  }
  hipLaunchKernelGGL(dunpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dwp, dw, Cout, Cin, CoutPad, CinPad,
                     total, k3);
  return xh_launch_status();
}

extern "C" int xh_cl_from_ncdhw(void* stream, int dtype, const void* xa, long long xa_bs, int CA, const void* xb, long long xb_bs, int CB,
                                void* out, int Cpad, int N, long long V) {
  if (!xa || !out || CA + CB > Cpad) return XH_ERR_ARG;
  if (dtype != XH_BF16 && dtype != XH_F16) return XH_ERR_DTYPE;
  const long long total = (long long)N * V;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == XH_F16) hipLaunchKernelGGL(cl_from_kernel<f16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const f16_t*)xa, xa_bs, CA, (const f16_t*)xb, xb_bs, CB, (u16*)out, Cpad, V, total);
  else hipLaunchKernelGGL(cl_from_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)xa, xa_bs, CA, (const bf16_t*)xb, xb_bs, CB, (u16*)out, Cpad, V, total);
  return xh_launch_status();
}
extern "C" int xh_cl_to_ncdhw(void* stream, int dtype, const void* g, int Cpad, void* da, long long da_bs, int CA, void* db, long long db_bs,
                              int CB, int N, long long V) {
  if (!g || !da || CA + CB > Cpad) return XH_ERR_ARG;
  if (dtype != XH_BF16 && dtype != XH_F16) return XH_ERR_DTYPE;
  const long long total = (long long)N * V;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == XH_F16) hipLaunchKernelGGL(cl_to_kernel<f16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const u16*)g, Cpad, (f16_t*)da, da_bs, CA, (f16_t*)db, db_bs, CB, V, total);
  else hipLaunchKernelGGL(cl_to_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const u16*)g, Cpad, (bf16_t*)da, da_bs, CA, (bf16_t*)db, db_bs, CB, V, total);
  return xh_launch_status();
}

extern "C" int xh_cl_affine_act(void* stream, int dtype, const void* x, void* y, const float* sc, const float* sh, float slope, int N, int C,
                                long long V) {
  if (!x || !y || !sc || !sh || (C % 8)) return XH_ERR_ARG;
  if (dtype != XH_BF16 && dtype != XH_F16) return XH_ERR_DTYPE;
  const long long total = (long long)N * V * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == XH_F16) hipLaunchKernelGGL(cl_affine_act_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const u16*)x, (u16*)y, sc, sh, slope, C, V, total);
  else hipLaunchKernelGGL(cl_affine_act_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const u16*)x, (u16*)y, sc, sh, slope, C, V, total);
  return xh_launch_status();
}

// mode 0: reduce (red += [sum g, sum g*x]); 1: dx = A*g + Cc*x + B; 2: dx = g, red[..][0] += sum g.   g = dy * leaky'(x*sc + sh)
extern "C" int xh_cl_act_bwd(void* stream, int dtype, int mode, const void* dy, const void* x, void* dx, const float* sc, const float* sh,
                             float slope, const float* A, const float* B, const float* Cc, double* red, int N, int C, long long V) {
  if (!dy || !x || (C % 8) || C > 512 || (256 % (C / 8)) || mode < 0 || mode > 2) return XH_ERR_ARG;
  if (dtype != XH_BF16 && dtype != XH_F16) return XH_ERR_DTYPE;
  if ((mode != 0 && !dx) || (mode != 1 && !red) || (mode == 1 && (!A || !B || !Cc))) return XH_ERR_ARG;
  const int nvl = 256 / (C / 8);
  long long blocks = (V + nvl * 16 - 1) / (nvl * 16);
  // the reducing modes end in one fp64 atomic per (channel, moment) and workgroup, and atomics on ONE address retire one after the
  // other (~65 ns each): with a workgroup per 16 voxel rows the 128-channel layer of a 128^3 patch (977 workgroups per sample) spent
  // 64 of its 101 us queueing them.  At most 256 workgroups per sample there; the apply mode keeps its finer grid
  const long long cap = mode == 1 ? 2048 : 256;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int vchunk = (int)((V + blocks - 1) / blocks);
  dim3 grid((unsigned)((V + vchunk - 1) / vchunk), N);
  hipStream_t st = (hipStream_t)stream;
#define LB(F, M) hipLaunchKernelGGL((cl_bwd_kernel<F, M>), grid, dim3(256), 0, st, (const u16*)dy, (const u16*)x, (u16*)dx, sc, sh, slope, A, B, Cc, red, C, V, vchunk)
  if (dtype == XH_F16) { if (mode == 0) LB(1, 0); else if (mode == 1) LB(1, 1); else LB(1, 2); }
  else { if (mode == 0) LB(0, 0); else if (mode == 1) LB(0, 1); else LB(0, 2); }
#undef LB
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Exact fp32 route of the Discriminator (parity mode, small patches): plain NCDHW fp32 direct convolutions for any kernel size
// with padding 1 -- nn.Conv3d(c, 2c, ks, stride, padding=1) of buildingblocks.py:350,354 -- forward, data gradient and weight /
// bias gradient.  No matrix cores, no 16-bit operands: fp32 FMA in a fixed order per output, so a training step in fp32 storage
// can be compared with the CPU oracle end to end (tests/test_gpu_trainstep.py).  One thread per output element / one workgroup
// per (co, ci) filter: ~1 TFLOP/s, meant for 32^3 .. 64^3 patches; the matrix-core kernels above are the product path.
struct DxArgs {
  const float* x; const float* w; const float* b; float* y;      // fwd: x, w, b -> y;  dgrad: x = dy, y = dx;  wgrad: x, y = dy -> w = dw, b = db
  int N, Cin, Cout, D, H, W, Do, Ho, Wo, ks, stride;
};
__global__ __launch_bounds__(256) void dconv_exact_fwd_kernel(const DxArgs a) {
  const long long total = (long long)a.N * a.Cout * a.Do * a.Ho * a.Wo;
  const int k3 = a.ks * a.ks * a.ks;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow = (int)(i % a.Wo);
    long long r = i / a.Wo;
    const int oh = (int)(r % a.Ho); r /= a.Ho;
    const int od = (int)(r % a.Do); r /= a.Do;
    const int co = (int)(r % a.Cout);
    const int n = (int)(r / a.Cout);
    float acc = a.b ? a.b[co] : 0.f;
    for (int ci = 0; ci < a.Cin; ++ci) {
      const float* xp = a.x + ((long long)n * a.Cin + ci) * a.D * a.H * a.W;
      const float* wp = a.w + ((long long)co * a.Cin + ci) * k3;
      for (int kd = 0; kd < a.ks; ++kd) {
        const int d = od * a.stride - 1 + kd;
        if ((unsigned)d >= (unsigned)a.D) continue;
        for (int kh = 0; kh < a.ks; ++kh) {
          const int h = oh * a.stride - 1 + kh;
          if ((unsigned)h >= (unsigned)a.H) continue;
          for (int kw = 0; kw < a.ks; ++kw) {
            const int w = ow * a.stride - 1 + kw;
            if ((unsigned)w >= (unsigned)a.W) continue;
            acc = fmaf(xp[((long long)d * a.H + h) * a.W + w], wp[(kd * a.ks + kh) * a.ks + kw], acc);
          }
        }
      }
    }
    a.y[i] = acc;
  }
}
// dx[n][ci][d][h][w] = sum_co sum_taps dy[n][co][(d + 1 - kd) / s][..] w[co][ci][kd][kh][kw]   (where the division is exact)
__global__ __launch_bounds__(256) void dconv_exact_dgrad_kernel(const DxArgs a) {
  const long long total = (long long)a.N * a.Cin * a.D * a.H * a.W;
  const int k3 = a.ks * a.ks * a.ks;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int w = (int)(i % a.W);
    long long r = i / a.W;
    const int h = (int)(r % a.H); r /= a.H;
    const int d = (int)(r % a.D); r /= a.D;
    const int ci = (int)(r % a.Cin);
    const int n = (int)(r / a.Cin);
    float acc = 0.f;
    for (int co = 0; co < a.Cout; ++co) {
      const float* gp = a.x + ((long long)n * a.Cout + co) * a.Do * a.Ho * a.Wo;
      const float* wp = a.w + ((long long)co * a.Cin + ci) * k3;
      for (int kd = 0; kd < a.ks; ++kd) {
        const int td = d + 1 - kd;
        if (td < 0 || td % a.stride) continue;
        const int od = td / a.stride;
        if (od >= a.Do) continue;
        for (int kh = 0; kh < a.ks; ++kh) {
          const int th = h + 1 - kh;
          if (th < 0 || th % a.stride) continue;
          const int oh = th / a.stride;
          if (oh >= a.Ho) continue;
          for (int kw = 0; kw < a.ks; ++kw) {
            const int tw = w + 1 - kw;
            if (tw < 0 || tw % a.stride) continue;
            const int ow = tw / a.stride;
            if (ow >= a.Wo) continue;
            acc = fmaf(gp[((long long)od * a.Ho + oh) * a.Wo + ow], wp[(kd * a.ks + kh) * a.ks + kw], acc);
          }
        }
      }
    }
    a.y[i] = acc;
  }
}
// one workgroup per (co, ci): dw[co][ci][tap] += sum_n sum_out dy[n][co][out] x[n][ci][out * s - 1 + tap]; db[co] += sum dy (ci == 0)
__global__ __launch_bounds__(256) void dconv_exact_wgrad_kernel(const DxArgs a) {
  __shared__ double s_red[4];
  const int co = blockIdx.x / a.Cin, ci = blockIdx.x % a.Cin;
  const int k3 = a.ks * a.ks * a.ks;
  const long long ovol = (long long)a.Do * a.Ho * a.Wo, vol = (long long)a.D * a.H * a.W;
  for (int tap = 0; tap <= k3; ++tap) {                 // tap == k3: the bias gradient (ci == 0 only)
    if (tap == k3 && (ci != 0 || !a.b)) break;
    const int kd = tap / (a.ks * a.ks), kh = (tap / a.ks) % a.ks, kw = tap % a.ks;
    double acc = 0.0;
    for (long long i = threadIdx.x; i < (long long)a.N * ovol; i += 256) {
      const int n = (int)(i / ovol);
      const long long o = i - (long long)n * ovol;
      const float g = a.y[((long long)n * a.Cout + co) * ovol + o];
      if (tap == k3) { acc += (double)g; continue; }
      const int ow = (int)(o % a.Wo), oh = (int)((o / a.Wo) % a.Ho), od = (int)(o / ((long long)a.Wo * a.Ho));
      const int d = od * a.stride - 1 + kd, h = oh * a.stride - 1 + kh, w = ow * a.stride - 1 + kw;
      if ((unsigned)d >= (unsigned)a.D || (unsigned)h >= (unsigned)a.H || (unsigned)w >= (unsigned)a.W) continue;
      acc += (double)g * (double)a.x[((long long)n * a.Cin + ci) * vol + ((long long)d * a.H + h) * a.W + w];
    }
    double v[1] = {acc};
    block_sum_d<1>(v, s_red, 4);
    if (threadIdx.x == 0) {
      if (tap == k3) const_cast<float*>(a.b)[co] += (float)s_red[0];
      else const_cast<float*>(a.w)[((long long)co * a.Cin + ci) * k3 + tap] += (float)s_red[0];
    }
    __syncthreads();
  }
}
static int dx_check(const DxArgs& a) {
  if (!a.x || !a.w || !a.y || a.N <= 0 || a.Cin <= 0 || a.Cout <= 0 || a.ks < 1 || a.ks > 7 || a.stride < 1 || a.stride > 2) return XH_ERR_ARG;
  if (a.Do != (a.D + 2 - a.ks) / a.stride + 1 || a.Ho != (a.H + 2 - a.ks) / a.stride + 1 || a.Wo != (a.W + 2 - a.ks) / a.stride + 1) return XH_ERR_ARG;
  if (a.Do <= 0 || a.Ho <= 0 || a.Wo <= 0) return XH_ERR_ARG;
  return XH_OK;
}
// mode 0: y = conv(x, w) + b;  1: dx (written to y) from dy (= x) and w;  2: dw (= w) += , db (= b, may be NULL) += from x and dy (= y)
extern "C" int xh_dconv_exact(void* stream, int mode, const float* x, const float* w, const float* b, float* y, int N, int Cin, int Cout,
                              int D, int H, int W, int Do, int Ho, int Wo, int ks, int stride) {
  DxArgs a{x, w, b, y, N, Cin, Cout, D, H, W, Do, Ho, Wo, ks, stride};
  const int rc = dx_check(a);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    const long long total = (long long)N * Cout * Do * Ho * Wo;
    hipLaunchKernelGGL(dconv_exact_fwd_kernel, dim3((unsigned)min((total + 255) / 256, (long long)65536)), dim3(256), 0, st, a);
  } else if (mode == 1) {
    const long long total = (long long)N * Cin * D * H * W;
    hipLaunchKernelGGL(dconv_exact_dgrad_kernel, dim3((unsigned)min((total + 255) / 256, (long long)65536)), dim3(256), 0, st, a);
  } else if (mode == 2) {
    if ((long long)Cout * Cin > 0x7fffffffll) return XH_ERR_ARG;
    hipLaunchKernelGGL(dconv_exact_wgrad_kernel, dim3((unsigned)(Cout * Cin)), dim3(256), 0, st, a);
  } else return XH_ERR_ARG;
  return xh_launch_status();
}
