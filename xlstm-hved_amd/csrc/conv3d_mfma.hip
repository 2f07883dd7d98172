// bf16-MFMA implicit-GEMM 3x3x3 stride-1 convolution for gfx950 (forward and data gradient), bf16 storage.
//
// GEMM view per output row segment of 16 voxels along W:   D[16 vox][16 cout] += A[16 vox][K] * B[K][16 cout]
// with K = 9 (kd,kh) row segments x (3 taps x CINP channels), walked in 16-byte chunks of 8 bf16.
//
//  * Input planes live in LDS channels-last ([h][w][CINP] bf16) so that for a fixed (kd,kh) the 3 taps x CINP
//    channels of output voxel w are ONE contiguous run starting at tile column w: an A fragment is a single 16-byte LDS
//    read per lane, no im2col buffer.  The producer's normalisation + LeakyReLU is applied while staging (NCDHW planes
//    -> 4-channel x 8-voxel register transpose -> 8-byte LDS writes), zero padding after it.
//  * Sliding window along D: a persistent workgroup owns an 8 x 32 (H x W) column and a run of SD output planes and keeps
//    a ring of 4 input planes in LDS.  Per output plane exactly ONE new input plane is fetched; its global loads are
//    issued before the MFMA phase of the current plane and land in registers while the matrix cores work (one barrier
//    per plane).  Depth halo re-reads drop from (TD+2)/TD to (SD+2)/SD.
//  * mfma_f32_16x16x32_bf16: lane l supplies A[row l&15][k 8*(l>>4)..+7] and B[k 8*(l>>4)..+7][col l&15]; lane group
//    g = l>>4 therefore owns chunk 4*i+g of the K walk in MFMA i.  Chunks past the end of a row segment read the next
//    voxel's (finite) data against zero weights.
//  * Weights are packed ONCE per launch by a small kernel into fragment order (caller workspace) and held in registers
//    (B operands) for the workgroup's whole life; groups are block-diagonal zeros inside a 16-wide output tile.
//  * Epilogue: straight from the accumulator layout -- lane (nn, g4) owns output channel nn and voxels 4*g4..+3 of each
//    16-voxel segment: bias / activation / fused reductions in registers, one 8-byte NCDHW store per segment.  The
//    fused reductions (output moments for the next InstanceNorm, or the leaky'-masked gradient sums of the norm
//    backward) are accumulated in fp64 registers over the whole run and reduced once.
#include "common.h"
#include "conv_pack.h"
#include "fanin.h"
#include "../../include/xlstm_hved.h"

typedef h16x8 bf16x8;     // 8 raw 16-bit values (either format)
typedef f32x4_t f32x4;

struct ConvMK {
  xh_conv_desc d;
  xh_conv_ptrs p;
  int Cin_g, Cout_g;
  int tilesW, tilesH, tw, th;
  int sd, dsegs;    // output planes per worker, number of depth segments
  int cin_blk;      // input channels staged per block (<= CINP)
  int ntile;        // 16-wide output tiles per set
  int cout_set;     // output channels per set
  int cinp, cpr, nch, nm;
  int abl;          // ablation mask for microbenchmarks (0 in production)
  // split-K over input channels (ungrouped convs with more than 24 of them: the decoders' virtual concats).  Launch s
  // covers channels [cin_off, cin_off + cin_blk); all but the last write fp32 partial sums to `part`, all but the first
  // add the partials in; only the last runs the epilogue proper.
  int nsplit, cin_off, part_in, part_out;
  int cin_stride;   // input channels between consecutive sets (= cin_blk unless split: then the whole group's cin_g)
  float* part;
  unsigned char* fan;  // statistics fan-in block of this launch (fanin.h), or nullptr: direct atomics
};
int g_mfma_abl = 0;
int g_mfma_wgs = 512;      // xh_set_option(3, n): target workgroup count of the k3 MFMA forward kernel (experiments)
int g_mfma_occ = 0;        // xh_set_option(4, 1): high-occupancy (<=128 VGPR) instances for CINP <= 8 at 128^3-class volumes

static void mk_pack_job(const ConvMK& a, PackJob* j) {
  j->dw = 0;
  for (int i = 0; i < XH_MAX_WPTR; ++i) j->w[i] = a.p.w[i];
  j->ws = a.p.ws;
  j->kind = 0;
  j->f16 = a.d.dtype == XH_F16;
  j->groups = a.d.groups; j->n_wptr = a.d.n_wptr; j->transposed = a.d.transposed;
  j->Cin_g = a.Cin_g; j->Cout_g = a.Cout_g;
  j->ntile = a.ntile; j->cin_stride = a.cin_stride; j->cin_off = a.cin_off; j->cin_blk = a.cin_blk;
  j->cout_set = a.cout_set; j->nm = a.nm; j->nch = a.nch; j->cpr = a.cpr; j->cinp = a.cinp;
  j->ci4 = 0;
  const int gs = a.nsplit > 1 ? 1 : a.cin_blk / a.Cin_g;
  j->nelem = (a.d.groups / gs) * a.ntile * a.nm * 512;
}

// one convolution's fragments (the launch in front of a convolution whose caller did not prepack)
__global__ __launch_bounds__(256) void conv3_pack_kernel(const PackJob j) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < j.nelem; idx += gridDim.x * 256) pack_elem(j, idx);
}
// the fragments of up to XH_PACK_MAX_JOBS convolutions (xh_conv3d_prepack)
__global__ __launch_bounds__(256) void conv3_pack_multi_kernel(const PackMulti m) {
  int i = 0;
  while (i + 1 < m.n && (int)blockIdx.x >= m.first_block[i + 1]) ++i;
  const PackJob& j = m.job[i];
  const int base = ((int)blockIdx.x - m.first_block[i]) * XH_PACK_PER_BLOCK;
#pragma unroll
  for (int u = 0; u < XH_PACK_PER_BLOCK / 256; ++u) {
    const int idx = base + u * 256 + threadIdx.x;
    if (idx < j.nelem) pack_elem(j, idx);
  }
}
// The same for ANY number of convolutions in ONE launch: the job table lives in device memory (xh_conv3d_prepack_table: the caller
// builds it on the host once per set of convolutions and keeps a device copy), so it is not bound by the 4 KB of kernel arguments.
// Table: {int n; int nblocks; int first_block[XH_PACK_TABLE_MAX + 1]; PackJob job[n]}.
#define XH_PACK_TABLE_MAX 256
struct PackTableHead { int n, nblocks; int first_block[XH_PACK_TABLE_MAX + 1]; int pad_; };
__global__ __launch_bounds__(256) void conv3_pack_table_kernel(const PackTableHead* __restrict__ h) {
  const PackJob* jobs = reinterpret_cast<const PackJob*>(h + 1);
  // the job whose block range holds blockIdx.x = the number of range starts (entries 1 .. n - 1) at or below it.  Thread t tests entry
  // t + 1: one load latency and a counting barrier instead of a binary search's chain of dependent loads (no measurable difference
  // by itself: what bounds the launch are the weight loads, see pack_q4_stage)
  const int nj = h->n;
  const int lo = __builtin_amdgcn_readfirstlane(__syncthreads_count((int)threadIdx.x + 1 < nj && h->first_block[threadIdx.x + 1] <= (int)blockIdx.x));
  // a REFERENCE into the table at a wave-uniform index: the fields arrive by scalar loads.  A copy of the record (`PackJob j = jobs[lo]`,
  // its pointer array indexed by a run-time group) was placed in scratch memory -- 160 bytes per lane, every field access a trip to
  // memory, wave launch gated by the scratch ring: the launch held 15 % of the wave slots and took 23 us for 6 MB of output
  const PackJob& j = jobs[lo];
  const int base = ((int)blockIdx.x - h->first_block[lo]) * XH_PACK_PER_BLOCK;
  if (j.kind == 1) {                                    // quad-channel image: the four fragments' 192 weights through LDS (conv_pack.h)
    __shared__ float s_w[4 * 48];
    pack_q4_stage(j, base, s_w);
    __syncthreads();
    const int idx8 = base + 8 * threadIdx.x;
    if (idx8 < j.nelem) pack_elem8_q4_lds(j, idx8, s_w);
    return;
  }
  if (j.kind == 0 && j.f16 != 2) {                      // a lane's 16 bytes at once (nelem is a multiple of 512 in both layouts)
    const int idx8 = base + 8 * threadIdx.x;
    if (idx8 < j.nelem) pack_elem8_mk(j, idx8);
    return;
  }
#pragma unroll
  for (int u = 0; u < XH_PACK_PER_BLOCK / 256; ++u) {
    const int idx = base + u * 256 + threadIdx.x;
    if (idx < j.nelem) pack_elem(j, idx);
  }
}
void xh_launch_pack_single(hipStream_t st, const PackJob& j) {
  hipLaunchKernelGGL(conv3_pack_kernel, dim3(min(cdiv(j.nelem, 2048), 64)), dim3(256), 0, st, j);
}

// LDS swizzle: XOR the 16-byte chunk index (bits 4..6) with the 256-byte block index (bits 8..10).  Both the 8-voxel
// group stride (8*VB) and the row stride are multiples of 128 B for the common layouts, which would make the staging
// writes (lanes differ in group / row) many-way bank conflicted; the XOR spreads them over all banks.  It is a
// bijection inside each 128-byte block, so fragment reads apply the same function.
__device__ __forceinline__ int swz(int off) { return off ^ (((off >> 8) & 7) << 4); }

template <int FMT, int CINP, int NT, int TW, int TH = 8, int MW = (TH == 4 ? 3 : 2)>
__global__ __launch_bounds__(NT, MW) void conv3_mfma_kernel(const ConvMK a) {
  typedef h16<FMT> ST;                                // storage type of activations: ST or f16_t
  constexpr int NWV = NT / 64;
  constexpr int NSEG = TW / 16;
  constexpr int IH = TH + 2;
  constexpr int IWP = TW + 4;                         // halo (2) + 2 spare columns for the over-reading tail chunk
  constexpr int VB = CINP * 2;                        // bytes per voxel in LDS
  constexpr int PLANE = IH * IWP * VB;                // bytes per ring slot
  constexpr int CPR = (3 * CINP + 7) / 8;             // 16-byte chunks per (kd,kh) row segment
  constexpr int NCH = 9 * CPR;
  constexpr int NM = (NCH + 3) / 4;                   // MFMAs per 16-voxel segment
  constexpr bool A16 = (CINP % 8) == 0;               // 16-byte aligned fragments
  constexpr int NQ = CINP / 4, NG = TW / 8 + 2;       // channel quads; aligned 8-voxel groups covering [ow0-8, ow0+TW+8)
  constexpr int NITEM = IH * NG * NQ;                 // staging items per plane
  constexpr int NIT = (NITEM + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_in = smem;                                                     // 4 * PLANE
  double* s_red = reinterpret_cast<double*>(smem + 4 * PLANE);                    // [NWV waves][32] after the plane loop

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g4 = lane >> 4, nn = lane & 15;
  const int y = blockIdx.y;
  const int set = y / a.ntile, nt = y % a.ntile;
  const int n = blockIdx.z;
  const int cin0 = set * a.cin_stride + a.cin_off;
  const int cin_end = (set + 1) * a.cin_stride;
  const int co_base = set * a.cout_set + nt * 16;
  const int co_lim = min(16, a.cout_set - nt * 16);
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W;
  const long long dhw = (long long)D * hw;
  const int Do = a.d.Do, Ho = a.d.Ho, Wo = a.d.Wo;
  const long long odhw = (long long)Do * Ho * Wo;
  int wk = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tw = wk % a.tilesW; wk /= a.tilesW;
  const int th = wk % a.tilesH;
  const int ds = wk / a.tilesH;
  const int oh0 = th * TH, ow0 = tw * TW;
  const int d_begin = ds * a.sd, d_end = min(Do, d_begin + a.sd);

  // ---- B fragments (registers, whole kernel) and A offsets ----
  bf16x8 bfrag[NM];
  int aoff[NM], akd[NM];
  {
    const bf16x8* wpk = reinterpret_cast<const bf16x8*>(a.p.ws) + (long long)y * NM * 64;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      bfrag[i] = wpk[i * 64 + lane];
      const int c = 4 * i + g4;
      const int cc = c < NCH ? c : 0;                 // dummy chunk: any valid address, its weights are zero
      const int r9 = cc / CPR, j = cc % CPR;
      akd[i] = r9 / 3;
      aoff[i] = ((r9 % 3) * IWP + nn) * VB + j * 16;
    }
  }
  // ---- epilogue lane role: the accumulator layout itself -- lane (nn, g4) holds output channel nn, voxels 4*g4 .. +3 of
  // each 16-voxel segment: bias / activation / fused sums in registers and one 8-byte NCDHW store per segment, with no
  // transpose through LDS (a 16-byte-store epilogue via a wave-private LDS pad cost two LDS round trips and two wave
  // barriers per row, and 18 KB of LDS) ----
  const int eco = nn;
  const int co = co_base + eco;
  const bool co_ok = eco < co_lim;
  float bias = 0.f, esc = 0.f, esh = 0.f;
  const ST* eplane = nullptr;
  if (co_ok) {
    const int g = co / a.Cout_g, gpp = a.d.groups / a.d.n_wptr;
    const float* bp = a.p.b[g / gpp];
    if (bp) bias = bp[(g % gpp) * a.Cout_g + co % a.Cout_g];
    if (a.d.epi == 1) {
      esc = a.p.e_sc[n * a.d.Cout + co];
      esh = a.p.e_sh[n * a.d.Cout + co];
      eplane = co < a.d.Cea ? (const ST*)a.p.ea + n * a.d.ea_bs + (long long)co * odhw
                            : (const ST*)a.p.eb + n * a.d.eb_bs + (long long)(co - a.d.Cea) * odhw;
    }
  }
  ST* yplane = (ST*)a.p.y + n * a.d.y_bs + (long long)(co_ok ? co : co_base) * odhw;
  double s0 = 0.0, s1 = 0.0;     // running statistics in fp64 (block_sum_d note in common.h)

  // ---- fused InstanceNorm finalisation (xh_conv_ptrs.fin_red): every workgroup turns the raw sums of ITS input channels
  // into scale / shift (fp64, bit-identical across workgroups); workgroup (0, 0, 0) of the first split also leaves them and
  // mean / rstd in memory for the backward pass ----
  const bool fin = a.p.fin_red != nullptr;
  float* s_fin = reinterpret_cast<float*>(s_red);     // [2][CINP] until the plane loop is over
  if (fin) {
    const double inv = 1.0 / (double)a.p.fin_count;
    if (tid < a.cin_blk && cin0 + tid < cin_end) {
      float m_, r_;
      const int i = n * a.d.Cin + cin0 + tid;
      in_finalize(a.p.fin_red[2 * i], a.p.fin_red[2 * i + 1], inv, s_fin[tid], s_fin[CINP + tid], m_, r_);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && a.cin_off == 0)
      for (int i = tid; i < a.d.N * a.d.Cin; i += NT)
        in_finalize(a.p.fin_red[2 * i], a.p.fin_red[2 * i + 1], inv, const_cast<float*>(a.p.pre_sc)[i],
                    const_cast<float*>(a.p.pre_sh)[i], a.p.fin_mean[i], a.p.fin_rstd[i]);
    __syncthreads();
  }
  // ---- per-thread staging plan (identical for every plane) ----
  const ST* sp_src[NIT][4];     // channel plane base + in-plane offset, or nullptr
  float sp_sc[NIT][4], sp_sh[NIT][4];
  int sp_lds[NIT], sp_gq[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = tid + it * NT;
    const int gi = item % NG;                         // group fastest: 6 x 16 B = one contiguous 96-byte run per row
    int r = item / NG;
    const int q = r % NQ;
    const int hy = r / NQ;
    const int gq = gi - 1;
    const int gh = oh0 - 1 + hy, gw = ow0 + gq * 8;
    const bool inb = item < NITEM && (unsigned)gh < (unsigned)H && gw >= 0 && gw < W;
    sp_gq[it] = item < NITEM ? gq : 100;              // 100: no LDS column is ever valid
    sp_lds[it] = (hy * IWP) * VB + q * 8;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int cl = q * 4 + cc;
      const int c = cin0 + cl;
      sp_src[it][cc] = nullptr;
      sp_sc[it][cc] = 1.f; sp_sh[it][cc] = 0.f;
      if (inb && cl < a.cin_blk && c < cin_end) {
        sp_src[it][cc] = (c < a.d.Ca ? (const ST*)a.p.xa + n * a.d.xa_bs + (long long)c * dhw
                                     : (const ST*)a.p.xb + n * a.d.xb_bs + (long long)(c - a.d.Ca) * dhw) +
                         (long long)gh * W + gw;
        if (fin) { sp_sc[it][cc] = s_fin[cl]; sp_sh[it][cc] = s_fin[CINP + cl]; }
        else if (a.d.pre) { sp_sc[it][cc] = a.p.pre_sc[n * a.d.Cin + c]; sp_sh[it][cc] = a.p.pre_sh[n * a.d.Cin + c]; }
      }
    }
  }
  // Small channel counts have too little MFMA work per plane to hide a global load behind it: their staging loads are
  // issued TWO planes ahead (a second register buffer), so a load has a whole plane iteration to land.
  constexpr bool DEEP = CINP <= 8;
  constexpr int AHEAD = DEEP ? 3 : 2;
  uint4 rawA[NIT][4], rawB[DEEP ? NIT : 1][4];
  auto load_plane = [&](int gd, uint4 (&raw)[NIT][4]) {
    const bool dok = (unsigned)gd < (unsigned)D && !(a.abl & 2);
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        raw[it][cc] = make_uint4(0, 0, 0, 0);
        if (dok && sp_src[it][cc]) raw[it][cc] = *reinterpret_cast<const uint4*>(sp_src[it][cc] + (long long)gd * hw);
      }
  };
  auto store_plane = [&](int gd, uint4 (&raw)[NIT][4]) {
    if (a.abl & 16) return;
    const int slot = ((gd + 4) & 3) * PLANE;
    const bool dok = (unsigned)gd < (unsigned)D;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      float v[4][8];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const bool live = dok && sp_src[it][cc] != nullptr;
        const unsigned u[4] = {raw[it][cc].x, raw[it][cc].y, raw[it][cc].z, raw[it][cc].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float lo = cvt_lo<FMT>(u[k]), hi = cvt_hi<FMT>(u[k]);
          if (a.d.pre) {
            lo = leaky(lo * sp_sc[it][cc] + sp_sh[it][cc], a.d.pre_slope);
            hi = leaky(hi * sp_sc[it][cc] + sp_sh[it][cc], a.d.pre_slope);
          }
          v[cc][2 * k] = live ? lo : 0.f;
          v[cc][2 * k + 1] = live ? hi : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int wx = sp_gq[it] * 8 + k + 1;         // tile column of voxel gw+k (column 0 = ow0-1)
        if (wx >= 0 && wx < IWP) {
          uint2 pk;
          pk.x = cvt_pack<FMT>(v[0][k], v[1][k]);
          pk.y = cvt_pack<FMT>(v[2][k], v[3][k]);
          *reinterpret_cast<uint2*>(s_in + swz(slot + sp_lds[it] + wx * VB)) = pk;
        }
      }
    }
  };

  // ---- prologue: planes d_begin-1, d_begin, d_begin+1 into the ring ----
  load_plane(d_begin - 1, rawA);
  store_plane(d_begin - 1, rawA);
  load_plane(d_begin, rawA);
  store_plane(d_begin, rawA);
  load_plane(d_begin + 1, rawA);
  store_plane(d_begin + 1, rawA);
  __syncthreads();

  constexpr int RPW = (TH + NWV - 1) / NWV;           // output rows per wave
  auto plane_step = [&](int d, uint4 (&rl)[NIT][4], uint4 (&rs)[NIT][4]) {
    // Order inside a plane: (1) issue the global loads of plane d+2, (2) MFMA pass over this wave's rows, (3) the loads
    // have landed: transform + LDS-write plane d+2 into the slot nobody reads, (4) epilogue: transpose, global stores,
    // (5) barrier.  The stores are the last memory operations of the iteration, so the only vmcnt wait (step 3 of the
    // NEXT plane) finds them long drained instead of stalling every plane on their acknowledgement.
    const bool more = d + 2 <= d_end;                 // plane d+2 is still to be staged
    if (d + AHEAD <= d_end) load_plane(d + AHEAD, rl);
    const int sbase = d + 3;                          // slot of input plane d-1+kd = (sbase + kd) & 3
    f32x4 accs[RPW][NSEG];
#pragma unroll
    for (int ri = 0; ri < RPW; ++ri) {
      const int rr = wv + ri * NWV;
#pragma unroll
      for (int wt = 0; wt < NSEG; ++wt) {
        const int rowoff = (rr * IWP + wt * 16) * VB;
        accs[ri][wt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rr < TH && oh0 + rr < Ho && !(a.abl & 4))   // wave-uniform
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          const int ao = ((sbase + akd[i]) & 3) * PLANE + rowoff + aoff[i];
          bf16x8 av;
          if (A16) {
            av = *reinterpret_cast<const bf16x8*>(s_in + swz(ao));
          } else {
            const uint2 lo = *reinterpret_cast<const uint2*>(s_in + swz(ao));
            const uint2 hi = *reinterpret_cast<const uint2*>(s_in + swz(ao + 8));
            const uint4 q4 = make_uint4(lo.x, lo.y, hi.x, hi.y);
            av = __builtin_bit_cast(bf16x8, q4);
          }
          accs[ri][wt] = mfma16x16x32<FMT>(av, bfrag[i], accs[ri][wt]);
        }
      }
    }
    if (more) store_plane(d + 2, rs);
#pragma unroll
    for (int ri = 0; ri < RPW; ++ri) {
      const int rr = wv + ri * NWV;
      const int oh = oh0 + rr;
      if (rr >= TH || oh >= Ho) continue;             // wave-uniform
      f32x4 (&acc)[NSEG] = accs[ri];
      if (a.abl & 32) continue;
      if (!co_ok) continue;                            // lanes of unused output columns
#pragma unroll
      for (int wt = 0; wt < NSEG; ++wt) {
        const long long sp = ((long long)d * Ho + oh) * Wo + ow0 + wt * 16 + 4 * g4;
        float o[4] = {acc[wt][0], acc[wt][1], acc[wt][2], acc[wt][3]};
        if (a.part_in | a.part_out) {                   // split-K partial sums (fp32, [n][cout][dhw])
          float* pp = a.part + ((long long)n * a.d.Cout + co) * odhw + sp;
          if (a.part_in) {
            const f32x4 pv = *reinterpret_cast<const f32x4*>(pp);
            o[0] += pv[0]; o[1] += pv[1]; o[2] += pv[2]; o[3] += pv[3];
          }
          if (a.part_out) {
            *reinterpret_cast<f32x4*>(pp) = f32x4{o[0], o[1], o[2], o[3]};
            continue;
          }
        }
        if (a.abl & 8) continue;
        float ev[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.d.epi == 1) ld4(eplane, sp, ev);
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = apply_act(o[r] + bias, a.d.act, a.d.act_slope);
          if (a.d.epi == 1) {
            v = cvt_in<FMT>(cvt_out<FMT>(v * ((ev[r] * esc + esh) > 0.f ? 1.f : a.d.e_slope)));
            t0 += v; t1 += v * ev[r];
          } else if (a.d.epi == 2) {
            v = cvt_in<FMT>(cvt_out<FMT>(v));
            t0 += v; t1 += v * v;
          }
          o[r] = v;
        }
        if (a.d.epi) { s0 += (double)t0; s1 += (double)t1; }
        st4(yplane, sp, o);
      }
    }
    __syncthreads();                                  // plane d+2 visible; everyone is done reading planes d-1..d+1
  };
  if constexpr (DEEP) {
    if (d_begin + 2 <= d_end) load_plane(d_begin + 2, rawA);
    for (int d = d_begin; d < d_end; d += 2) {
      plane_step(d, rawB, rawA);                      // loads plane d+3 into B, stages plane d+2 from A
      if (d + 1 < d_end) plane_step(d + 1, rawA, rawB);
    }
  } else {
    for (int d = d_begin; d < d_end; ++d) plane_step(d, rawA, rawA);
  }
  if (a.d.epi && !a.part_out) {
    s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    __syncthreads();
    if (lane < 16) { s_red[wv * 32 + eco * 2] = s0; s_red[wv * 32 + eco * 2 + 1] = s1; }
    __syncthreads();
    double tot = 0.0;
    if (tid < 32) {
#pragma unroll
      for (int w8 = 0; w8 < NWV; ++w8) tot += s_red[w8 * 32 + tid];
    }
    if (a.fan) {                                       // two-level fan-in instead of gridDim.x same-line atomics (fanin.h)
      __syncthreads();
      if (tid < 32) s_red[tid] = tot;
      if (!fan_in<32>(a.fan + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * FAN_UNIT_BYTES, blockIdx.x, gridDim.x, s_red,
                      reinterpret_cast<int*>(s_red + 32)))
        return;
      tot = tid < 32 ? s_red[tid] : 0.0;
    }
    if (tid < 32) {
      const int c = tid >> 1;
      if (c < co_lim) atomicAdd(&a.p.red[((long long)n * a.d.Cout + co_base + c) * 2 + (tid & 1)], tot);
    }
  }
}

static int mfma_plan(const xh_conv_desc* d, ConvMK* a) {
  if ((d->dtype != XH_BF16 && d->dtype != XH_F16) || d->k != 3 || d->stride != 1) return 1;
  if (d->W % 16 != 0 || d->Wo != d->W) return 1;
  const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  if (cin_g < 4) return 1;                            // depthwise / single-channel convs stay on the vector kernel
  if ((d->xa_bs & 7) || (d->xb_bs & 7) || (d->y_bs & 7) || (d->ea_bs & 7) || (d->eb_bs & 7)) return 1;
  const long long dhw = (long long)d->D * d->H * d->W;
  if (dhw % 8) return 1;
  int gs = 1;                                         // groups per set: as many as fit 24 input / 16 output channels
  while (gs * 2 <= d->groups && d->groups % (gs * 2) == 0 && gs * 2 * cin_g <= 24 && gs * 2 * cout_g <= 16) gs *= 2;
  int cin_blk = gs * cin_g;
  a->nsplit = 1;
  a->cin_stride = cin_blk;
  if (cin_blk > 24) {                                 // gs == 1 here: one group (or the whole ungrouped conv) per set
    a->nsplit = cdiv(cin_blk, 24);
    cin_blk = cdiv(cdiv(cin_blk, a->nsplit), 4) * 4;  // equal chunks, whole channel quads
    if (cin_blk > 24 || a->nsplit > 8) return 1;
  }
  a->cin_off = 0; a->part_in = a->part_out = 0; a->part = nullptr;
  if (d->N > 65535) return 1;
  a->d = *d;
  a->Cin_g = cin_g; a->Cout_g = cout_g;
  a->tw = (d->W % 32 == 0) ? 32 : 16;
  // 4-row tiles (ablation bit 64): 37 KB of LDS per workgroup -> 3 resident workgroups per CU instead of 2
  // (measured: 158 us vs 103 us for 16->16 @128^3 -- the per-plane fixed work per workgroup dominates -- so not used)
  a->th = (g_mfma_abl & 64) ? 4 : 8;
  a->tilesW = d->W / a->tw; a->tilesH = cdiv(d->Ho, a->th);
  a->cin_blk = cin_blk;
  a->cout_set = gs * cout_g;
  a->ntile = cdiv(a->cout_set, 16);
  a->cinp = cin_blk <= 4 ? 4 : cin_blk <= 8 ? 8 : cin_blk <= 12 ? 12 : cin_blk <= 16 ? 16 : 24;
  a->cpr = (3 * a->cinp + 7) / 8;
  a->nch = 9 * a->cpr;
  a->nm = (a->nch + 3) / 4;
  a->abl = g_mfma_abl;
  const int ny = (d->groups / gs) * a->ntile;
  if (ny > 65535) return 1;
  // depth segments: enough workers for ~2 resident workgroups per CU, but runs of at least 4 planes
  const int cols = a->tilesW * a->tilesH;
  int dsegs = cdiv(a->th == 4 ? 1024 : g_mfma_wgs, cols * ny * d->N);
  // runs of >= 4 planes (2 on small volumes, where workgroup count matters more than the 2 halo planes per run)
  const int min_run = ((long long)d->Do * d->Ho * d->Wo <= (1 << 16) && !(g_mfma_abl & 512)) ? 2 : 4;
  const int max_segs = d->Do >= min_run ? d->Do / min_run : 1;
  if (dsegs > max_segs) dsegs = max_segs;
  if (dsegs < 1) dsegs = 1;
  a->sd = cdiv(d->Do, dsegs);
  a->dsegs = cdiv(d->Do, a->sd);
  return 0;
}

static long long pack_bytes(const xh_conv_desc* d, const ConvMK& a) {
  const int gs = a.nsplit > 1 ? 1 : a.cin_blk / a.Cin_g;
  return (long long)(d->groups / gs) * a.ntile * a.nm * 1024;
}
long long xh_conv3_q4_workspace_bytes(const xh_conv_desc* d);                            // conv3d_q4.hip
int xh_conv3_q4_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p);
extern "C" long long xh_conv3d_workspace_bytes(const xh_conv_desc* d) {
  ConvMK a;
  if (!d || d->groups <= 0 || d->Cin % d->groups || d->Cout % d->groups) return 0;
  if (const long long q4 = xh_conv3_q4_workspace_bytes(d)) return q4;
  if (mfma_plan(d, &a)) return 0;
  long long need = pack_bytes(d, a) * a.nsplit;
  if (a.nsplit > 1) need += (long long)d->N * d->Cout * d->Do * d->Ho * d->Wo * (long long)sizeof(float);
  return need;
}

// returns XH_OK if launched, 1 if the shape is not eligible (caller falls back to the vector kernel)
int xh_conv3_mfma_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p) {
  ConvMK a;
  {                                                   // few channels per group: the quad-channel W-Toeplitz kernel
    const int r = xh_conv3_q4_try(stream, d, p);
    if (r != 1) return r;
  }
  if (mfma_plan(d, &a)) return 1;
  const long long need = xh_conv3d_workspace_bytes(d);
  if (!p->ws || p->ws_bytes < need) return 1;
  a.p = *p;
  const long long pb = pack_bytes(d, a);
  const int gs = a.nsplit > 1 ? 1 : a.cin_blk / a.Cin_g;
  const int ny = (d->groups / gs) * a.ntile;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(a.tilesW * a.tilesH * a.dsegs, ny, d->N);
  a.fan = (d->epi && a.nsplit == 1) ? xh_fan_block(p->fan, p->fan_bytes, (long long)grid.y * grid.z, grid.x) : nullptr;
  const size_t shm = (size_t)4 * (a.th + 2) * (a.tw + 4) * a.cinp * 2 + (size_t)8 * 32 * sizeof(double);
  // 8-wave workgroups hide the per-plane serial chain better on small volumes; 4-wave ones win on 128^3-class volumes
  const bool big = (long long)d->Do * d->Ho * d->Wo >= (1 << 20);
  xh_note_kernel("conv3_mfma_kernel<%d, %d, %d, %d, %d, %d>", d->dtype == XH_F16 ? 1 : 0, a.cinp, a.tw == 16 ? 512 : (big ? 256 : 512),
                 a.tw == 16 ? 16 : 32, a.th, a.th == 4 ? 3 : ((big && g_mfma_occ && a.cinp <= 8 && a.tw != 16) ? 4 : 2));
  for (int sidx = 0; sidx < a.nsplit; ++sidx) {
  if (a.nsplit > 1) {
    a.cin_off = sidx * a.cin_blk;
    a.part_in = sidx > 0;
    a.part_out = sidx + 1 < a.nsplit;
    a.part = reinterpret_cast<float*>((char*)p->ws + pb * a.nsplit);
    a.p.ws = (char*)p->ws + pb * sidx;
  }
  if (!p->ws_packed) {
    PackJob pj;
    mk_pack_job(a, &pj);
    xh_launch_pack_single(st, pj);
  }
#define LM(F, C)                                                                                                \
  do {                                                                                                          \
    static bool attr_done[XH_MAX_DEV] = {};                                                                              \
    if (xh_attr_needed(attr_done)) {                                                                                           \
      (void)hipFuncSetAttribute((const void*)conv3_mfma_kernel<F, C, 256, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
      (void)hipFuncSetAttribute((const void*)conv3_mfma_kernel<F, C, 512, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
      (void)hipFuncSetAttribute((const void*)conv3_mfma_kernel<F, C, 512, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
    }                                                                                                           \
    if (a.tw == 16) hipLaunchKernelGGL((conv3_mfma_kernel<F, C, 512, 16>), grid, dim3(512), shm, st, a);        \
    else if (a.th == 4) hipLaunchKernelGGL((conv3_mfma_kernel<F, C, 256, 32, 4>), grid, dim3(256), shm, st, a); \
    else if (big && g_mfma_occ && C <= 8) hipLaunchKernelGGL((conv3_mfma_kernel<F, (C <= 8 ? C : 4), 256, 32, 8, 4>), grid, dim3(256), shm, st, a); \
    else if (big) hipLaunchKernelGGL((conv3_mfma_kernel<F, C, 256, 32>), grid, dim3(256), shm, st, a);          \
    else hipLaunchKernelGGL((conv3_mfma_kernel<F, C, 512, 32>), grid, dim3(512), shm, st, a);                   \
  } while (0)
#define LMF(C)                            \
  do {                                    \
    if (d->dtype == XH_F16) LM(1, C);     \
    else LM(0, C);                        \
  } while (0)
  switch (a.cinp) {
    case 4: LMF(4); break;
    case 8: LMF(8); break;
    case 12: LMF(12); break;
    case 16: LMF(16); break;
    default: LMF(24);
  }
  }
#undef LMF
#undef LM
  return xh_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// xh_conv3d_prepack: the weight fragments of n convolutions in ceil(jobs / 24) launches (a split-K convolution is one
// job per split).  d[i] / p[i] are what xh_conv3d_fwd will be called with (only the weights, the workspace and the shape
// are read); a convolution that is not on the MFMA path is skipped.  Afterwards the caller sets xh_conv_ptrs.ws_packed.
bool xh_conv3_q4_pack_job(const xh_conv_desc* d, const xh_conv_ptrs* p, PackJob* j);        // conv3d_q4.hip
extern int g_use_mfma;
// every pack job of the n convolutions, in order, handed to `push`; XH_ERR_ARG on a null entry
template <typename PUSH> static int pack_jobs_of(int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, PUSH push) {
  for (int i = 0; i < n; ++i) {
    if (!d[i] || !p[i]) return XH_ERR_ARG;
    if (d[i]->k != 3 || d[i]->stride != 1 || d[i]->groups <= 0 || d[i]->Cin % d[i]->groups || d[i]->Cout % d[i]->groups) continue;
    const long long need = xh_conv3d_workspace_bytes(d[i]);
    if (!need || !p[i]->ws || p[i]->ws_bytes < need) continue;
    PackJob j;
    if (xh_conv3_q4_pack_job(d[i], p[i], &j)) { push(j); continue; }
    ConvMK a;
    if (mfma_plan(d[i], &a)) continue;
    a.p = *p[i];
    const long long pb = pack_bytes(d[i], a);
    for (int sidx = 0; sidx < a.nsplit; ++sidx) {
      if (a.nsplit > 1) {
        a.cin_off = sidx * a.cin_blk;
        a.p.ws = (char*)p[i]->ws + pb * sidx;
      }
      mk_pack_job(a, &j);
      push(j);
    }
  }
  return XH_OK;
}
extern "C" int xh_conv3d_prepack(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p) {
  if (n < 0 || (n > 0 && (!d || !p))) return XH_ERR_ARG;
  if (!g_use_mfma) return XH_OK;
  hipStream_t st = (hipStream_t)stream;
  PackMulti m;
  m.n = 0;
  m.first_block[0] = 0;
  auto flush = [&]() {
    if (m.n == 0) return;
    hipLaunchKernelGGL(conv3_pack_multi_kernel, dim3(m.first_block[m.n]), dim3(256), 0, st, m);
    m.n = 0;
  };
  const int rc = pack_jobs_of(n, d, p, [&](const PackJob& j) {
    m.job[m.n] = j;
    m.first_block[m.n + 1] = m.first_block[m.n] + cdiv(j.nelem, XH_PACK_PER_BLOCK);
    if (++m.n == XH_PACK_MAX_JOBS) flush();
  });
  if (rc) return rc;
  flush();
  return xh_launch_status();
}
extern "C" long long xh_conv3d_prepack_table_bytes(void) { return (long long)sizeof(PackTableHead) + (long long)XH_PACK_TABLE_MAX * sizeof(PackJob); }
extern "C" int xh_conv3d_prepack_table(int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, void* host_table) {
  if (n < 0 || (n > 0 && (!d || !p)) || !host_table) return XH_ERR_ARG;
  PackTableHead* h = reinterpret_cast<PackTableHead*>(host_table);
  PackJob* jobs = reinterpret_cast<PackJob*>(h + 1);
  h->n = 0; h->nblocks = 0; h->first_block[0] = 0; h->pad_ = 0;
  if (!g_use_mfma) return XH_OK;
  bool full = false;
  const int rc = pack_jobs_of(n, d, p, [&](const PackJob& j) {
    if (h->n >= XH_PACK_TABLE_MAX) { full = true; return; }
    jobs[h->n] = j;
    h->first_block[h->n + 1] = h->first_block[h->n] + cdiv(j.nelem, XH_PACK_PER_BLOCK);
    ++h->n;
  });
  if (rc) return rc;
  if (full) return XH_ERR_ARG;
  h->nblocks = h->first_block[h->n];
  return XH_OK;
}
extern "C" int xh_conv3d_prepack_run(void* stream, const void* dev_table, int nblocks) {
  if (nblocks < 0 || (nblocks > 0 && !dev_table)) return XH_ERR_ARG;
  if (nblocks == 0) return XH_OK;
  hipLaunchKernelGGL(conv3_pack_table_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const PackTableHead*>(dev_table));
  return xh_launch_status();
}
