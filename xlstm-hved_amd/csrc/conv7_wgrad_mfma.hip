// bf16-MFMA weight gradient of the 7x7x7 gate conv (4 pooled channels -> 2 gates), bf16 storage, fp32 accumulation.
//
//   dW[co][ci][kd][kh][kw] = sum_{d,h,w} x[ci][d+kd-3][h+kh-3][w+kw-3] * dY[co][d][h][w]
//
// GEMM per (x plane p, x row r, 32-voxel run along W) and depth tap kd, with K = the 32 voxels:
//
//   C_kd[(kw, ci)][(kh, co)] += A[(kw, ci)][w] * B_kd[w][(kh, co)],   A = x[ci][p][r][w + kw - 3],
//                                                                       B_kd = dY[co][p - kd + 3][r - kh + 3][w]
//
// M = 8 kw slots x 4 ci = two 16-row tiles, N = 8 kh slots x 2 co = 16 (slot 7 of kw / kh is a dummy), so one x row feeds
// 14 MFMAs (7 kd x 2 M tiles) from 2 A and 7 B fragments, and the 7^3 x 8 weight gradient lives in 14 accumulator tiles
// per wave for the whole run.
//  * the kw shift is a 2-byte-granular shift along the contraction axis, which a 16-byte LDS fragment read cannot do;
//    so the x rows are staged as 8 pre-shifted copies ([ci][row][kw][32 w], built in registers with constant byte
//    aligns from the neighbouring 16-byte chunks): every A fragment is one aligned 16-byte read;
//  * dY planes p-3 .. p+3 live in an 8-slot LDS ring ([co][14 rows][32 w]); the workgroup slides along D over x planes;
//  * per-workgroup partial gradients go to a scratch buffer (plain stores) and a second tiny kernel adds them into dW/db:
//    512 workgroups x 2744 same-address atomics would serialise.
#include "common.h"
#include "../../include/xlstm_hved.h"
#include <algorithm>
#include <vector>

typedef h16x8 bf16x8;     // 8 raw 16-bit values (either format)
typedef f32x4_t f32x4;

struct Wg7K {
  xh_conv_desc d;
  const void* x;        // (N, 4, D, H, W), 16-bit or (FMT 2) fp32 elements
  const void* dy;       // (N, 2, D, H, W)
  long long x_bs, dy_bs;
  float* part;          // [workgroups][NPART]
  int tilesW, tilesH, sd, dsegs;
};
constexpr int W7_XBUF = 4 * (8 * 8 * 96 + 64);      // LDS: x image of one plane (conv7_wgrad_body: 4 channels x (8 rows x 8 copies x 96 B + 64))
constexpr int W7_COS = 14 * 64 + 32, W7_DYPL = 2 * W7_COS;
constexpr size_t W7_SHM = 2 * W7_XBUF + 8 * W7_DYPL + 64;
constexpr int K7_NW = 2 * 4 * 343;          // 2744 weight gradients
constexpr int K7_NPART = K7_NW + 8;         // + 2 bias gradients (padded)

// FMT 0 / 1: bf16 / fp16 storage.  FMT 2: fp32 STORAGE (xh_set_option(18, 1)): 8-voxel chunks are read as 32 bytes and rounded once
// to fp16 on their way into LDS (CF = 1 below), fp32 accumulation -- like conv3_wgrad_q4_multi_kernel<2, ...>: a weight gradient
// sums ~10^6 independently rounded products; dY needs the caller's loss scale for fp16's range.
template <int FMT> struct W7Store { typedef bf16_t T; };
template <> struct W7Store<2> { typedef float T; };
template <int FMT>
__device__ __forceinline__ void conv7_wgrad_body(const Wg7K& a, const int bid, const int nwg) {
  typedef typename W7Store<FMT>::T ST;
  constexpr int CF = FMT == 2 ? 1 : FMT;              // format of the LDS images / MFMA operands
  constexpr int TH = 8, TW = 32;
  // LDS strides chosen for the ds_read_b128 bank groups (MI355X_MICROARCH.md: a wave's read is served in four groups of 16 lanes,
  // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...; 64 banks x 4 B = sixteen 16-byte chunks per cycle).  Round 5's layout (copies
  // 64 B apart, channels 4 096 B apart, dY channels 896 B apart) put the four input channels of an A fragment on the SAME chunk
  // (4-way conflict) and the two gate channels of a B fragment two-way: 352 LDS-pipe cycles per row against 224 of MFMA, on a pipe
  // that all 8 waves of a CU share -- the launch was bound by it (SQ: MFMA busy 0.086, issue 0.14).  Copies 96 B apart, channels
  // = 64 mod 256 B apart and dY channels = 32 mod 256 B apart make every group hit sixteen distinct chunks.
  constexpr int KWS = 96;                             // bytes between the shifted copies of a row (64 of data)
  constexpr int XROW = 8 * KWS;                       // bytes per (ci, row): 8 shifted copies x 32 bf16
  constexpr int CIS = TH * XROW + 64;                 // bytes per input channel
  constexpr int XBUF = W7_XBUF;                       // 4 channels
  constexpr int DYROWS = TH + 6, COS = W7_COS, DYPL = W7_DYPL;   // bytes per dY channel / plane slot
  static_assert(XBUF == 4 * CIS && COS == DYROWS * 64 + 32 && DYPL == 2 * COS, "host-side LDS size");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_x = smem;                          // 2 * XBUF (double buffer)
  unsigned char* s_dy = smem + 2 * XBUF;              // 8 * DYPL
  unsigned char* s_zero = s_dy + 8 * DYPL;            // 64 B of zeros (dummy kh = 7 columns)
  float* s_red = reinterpret_cast<float*>(smem);      // after the plane loop: [K7_NPART]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g4 = lane >> 4, nn = lane & 15;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  int wk = xcd_swizzle(bid, nwg);
  const int tw = wk % a.tilesW; wk /= a.tilesW;
  const int th = wk % a.tilesH; wk /= a.tilesH;
  const int ds = wk % a.dsegs;
  const int n = wk / a.dsegs;
  const int oh0 = th * TH, ow0 = tw * TW;
  const int p_begin = ds * a.sd, p_end = min(D, p_begin + a.sd);
  if (tid < 16) reinterpret_cast<unsigned*>(s_zero)[tid] = 0u;

  // ---- staging roles: threads 0..127 stage x (ci, row, chunk); threads 128..255 stage dY (co, row, chunk) ----
  const bool xrole = tid < 128;
  const int it = tid & 127;
  // x item
  const int xc = it & 3, xrow = (it >> 2) & 7, xci = it >> 5;
  const int xh = oh0 + xrow;
  const bool x_ok = xh < H;
  const ST* xsrc = (const ST*)a.x + n * a.x_bs + (long long)xci * dhw + (long long)min(xh, H - 1) * W + ow0 + 8 * xc;
  const bool x_prev = ow0 + 8 * xc - 8 >= 0, x_next = ow0 + 8 * xc + 8 < W;
  // dY item (112 of the 128 threads)
  const int yc = it & 3, yrow = (it >> 2) % DYROWS, yco = (it >> 2) / DYROWS;
  const int yh = oh0 - 3 + yrow;
  const bool y_item = !xrole && it < 2 * DYROWS * 4;
  const bool y_ok = y_item && (unsigned)yh < (unsigned)H;
  const ST* ysrc = (const ST*)a.dy + n * a.dy_bs + (long long)min(yco, 1) * dhw + (long long)min(max(yh, 0), H - 1) * W + ow0 + 8 * yc;
  const bool y_own_row = yrow >= 3 && yrow < 3 + TH;  // a row of the tile proper (bias gradient counts those once)

  float dbs = 0.f;
  // 8 voxels at s as 8 packed 16-bit values (fp32 storage: two 16-byte loads, rounded to fp16)
  auto ld8 = [&](const ST* s) -> uint4 {
    if constexpr (FMT == 2) {
      const float4 lo = *reinterpret_cast<const float4*>(s), hi = *reinterpret_cast<const float4*>(s + 4);
      return make_uint4(cvt2_pack<1>(lo.x, lo.y), cvt2_pack<1>(lo.z, lo.w), cvt2_pack<1>(hi.x, hi.y), cvt2_pack<1>(hi.z, hi.w));
    } else {
      return *reinterpret_cast<const uint4*>(s);
    }
  };
  // Staging, round 6: EVERY thread requests the same three pieces (its chunk and the two neighbours; a dY thread's neighbours are
  // its own chunk again) from a clamped, always valid address, unconditionally and TWO planes ahead of their use -- the first version
  // loaded under `if (in range)` one plane ahead: a load under a branch makes hipcc wait for everything at the next use, so every plane
  // step exposed a full memory latency (16 steps + a prologue of 7 serial dY loads = most of the launch's 55 us at 128^3).  What is
  // outside the volume is masked when the piece is committed to LDS.
  struct Stage { uint4 prev, cur, next; };
  const ST* src_t = xrole ? xsrc : ysrc;                // this thread's row (plane 0)
  const int off_prev = xrole && x_prev ? -8 : 0, off_next = xrole && x_next ? 8 : 0;
  const int plane_off = xrole ? 0 : 3;                  // a dY thread stages plane p + 3 + k when an x thread stages p + k
  auto issue = [&](int pl, Stage& st) {                 // pl: x plane index; the dY thread takes pl + 3
    const ST* s = src_t + (long long)min(max(pl + plane_off, 0), D - 1) * hw;
    st.cur = ld8(s);
    st.prev = ld8(s + off_prev);
    st.next = ld8(s + off_next);
  };
  auto commit_x = [&](int pl, const Stage& st, int buf) {
    const unsigned mk = (x_ok && (unsigned)pl < (unsigned)D) ? 0xffffffffu : 0u;
    const unsigned mp = x_prev ? mk : 0u, mn = x_next ? mk : 0u;
    const unsigned w[12] = {st.prev.x & mp, st.prev.y & mp, st.prev.z & mp, st.prev.w & mp, st.cur.x & mk, st.cur.y & mk, st.cur.z & mk, st.cur.w & mk,
                            st.next.x & mn, st.next.y & mn, st.next.z & mn, st.next.w & mn};
    unsigned char* dst = s_x + buf * XBUF + xci * CIS + xrow * XROW + xc * 16;
#pragma unroll
    for (int kw = 0; kw < 8; ++kw) {
      const int o = 8 + kw - 3;                       // first element of the shifted window inside prev|cur|next
      unsigned q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        q[j] = (o & 1) ? __builtin_amdgcn_alignbyte(w[(o + 1) / 2 + j], w[(o - 1) / 2 + j], 2) : w[o / 2 + j];
      *reinterpret_cast<uint4*>(dst + kw * KWS) = make_uint4(q[0], q[1], q[2], q[3]);
    }
  };
  auto commit_dy = [&](int q, const Stage& st) {       // q: dY plane index
    if (!y_item) return;
    const unsigned mk = (y_ok && (unsigned)q < (unsigned)D) ? 0xffffffffu : 0u;
    const uint4 v = make_uint4(st.cur.x & mk, st.cur.y & mk, st.cur.z & mk, st.cur.w & mk);
    *reinterpret_cast<uint4*>(s_dy + ((q + 8) & 7) * DYPL + yco * COS + yrow * 64 + yc * 16) = v;
    if (y_own_row && q >= p_begin && q < p_end) {
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) dbs += cvt_lo<CF>(u[k]) + cvt_hi<CF>(u[k]);
    }
  };

  // ---- prologue: x plane p_begin, dY planes p_begin-3 .. p_begin+3: all requested, then committed ----
  Stage s0, s1;
  {
    Stage pro[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) issue(xrole ? p_begin : p_begin - 6 + k, pro[k]);   // x: plane p_begin (7 times the same lines: L1 hits); dY: planes p_begin - 3 + k
    issue(p_begin + 1, s0);                           // x plane p_begin + 1 / dY plane p_begin + 4
    issue(p_begin + 2, s1);
    if (xrole) commit_x(p_begin, pro[0], 0);
    else {
#pragma unroll
      for (int k = 0; k < 7; ++k) commit_dy(p_begin - 3 + k, pro[k]);
    }
  }
  __syncthreads();

  // lane roles in the MFMAs
  const int a_off0 = (nn & 3) * CIS + (nn >> 2) * KWS + g4 * 16;            // M tile 0: kw = nn>>2, ci = nn&3
  const int a_off1 = a_off0 + 4 * KWS;                                      // M tile 1: kw + 4
  const int kh_l = nn >> 1, co_l = nn & 1;
  f32x4 acc[7][2];
#pragma unroll
  for (int kd = 0; kd < 7; ++kd) { acc[kd][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[kd][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  int buf = 0;
  for (int p = p_begin; p < p_end; ++p) {
    const bool more = p + 1 < p_end;
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
      const int r = wv * 2 + ri;                      // x row of the tile (wave-uniform)
      const unsigned char* xr = s_x + buf * XBUF + r * XROW;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(xr + a_off0);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(xr + a_off1);
      const int ry = r + 6 - kh_l;                    // dY row (ring-relative) this lane's column pairs with
#pragma unroll
      for (int kd = 0; kd < 7; ++kd) {
        const unsigned char* bp = kh_l < 7 ? s_dy + ((p - kd + 3 + 8) & 7) * DYPL + co_l * COS + ry * 64 + g4 * 16
                                           : s_zero + g4 * 16;
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bp);
        acc[kd][0] = mfma16x16x32<CF>(a0, bv, acc[kd][0]);
        acc[kd][1] = mfma16x16x32<CF>(a1, bv, acc[kd][1]);
      }
    }
    // stage s0 (requested two iterations ago) holds x plane p + 1 / dY plane p + 4; then s1 moves up and plane p + 3 is requested
    if (more) { if (xrole) commit_x(p + 1, s0, buf ^ 1); else commit_dy(p + 4, s0); }
    s0 = s1;
    issue(p + 3, s1);
    buf ^= 1;
    __syncthreads();
  }

  // ---- reduce the four waves' tiles in LDS: C[(kw, ci)][(kh, co)]: lane holds rows 4*g4 + r -> kw = 4*mt + g4, ci = r ----
  // Every wave STORES its 2 744 values into a slice of its own (it holds each of them exactly once) and the slices are summed on the
  // way out.  Round 6: this was 49 ds_add_f32 per lane into one slice, and LDS floating-point atomics retire about a lane per two
  // cycles -- 12 500 per workgroup, a fifth of the launch (found on conv3d_wgrad_mfma.hip by switching its tail off).
  static_assert((size_t)4 * K7_NPART * sizeof(float) <= W7_SHM, "four slices of partial sums in the staging buffers");
  if (tid < 8) s_red[K7_NW + tid] = 0.f;                 // the bias sums stay atomic (two addresses, a few dozen lanes), in slice 0
  __syncthreads();
  if (kh_l < 7) {
    float* my = s_red + (tid >> 6) * K7_NPART;
#pragma unroll
    for (int kd = 0; kd < 7; ++kd)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int kw = 4 * mt + g4;
        if (kw < 7) {
#pragma unroll
          for (int r = 0; r < 4; ++r) my[(co_l * 4 + r) * 343 + (kd * 7 + kh_l) * 7 + kw] = acc[kd][mt][r];
        }
      }
  }
  if (!xrole && y_item) atomicAdd(&s_red[K7_NW + yco], dbs);
  __syncthreads();
  float* out = a.part + (long long)bid * K7_NPART;
  for (int i = tid; i < K7_NPART; i += 256)
    out[i] = i < K7_NW ? ((s_red[i] + s_red[K7_NPART + i]) + s_red[2 * K7_NPART + i]) + s_red[3 * K7_NPART + i] : s_red[i];
}
template <int FMT>
__global__ __launch_bounds__(256, 2) void conv7_wgrad_mfma_kernel(const Wg7K a) {
  conv7_wgrad_body<FMT>(a, blockIdx.x, gridDim.x);
}
// The 7^3 weight gradients of a batch (xh_conv3d_wgrad_batch: the three AttenModule2 gates of a step) in one launch: alone the
// 64^3 / 32^3 problems are 128 / 64 workgroups of 27 / 21 us next to the 128^3 one's 512 of 55 us.  Workgroup b belongs to
// problem i with off[i] <= b < off[i + 1] (largest problem first, so the long workgroups start first).
constexpr int WG7_MULTI = 4;
struct Wg7Multi {
  int n;
  int off[WG7_MULTI + 1];
  Wg7K p[WG7_MULTI];
};
struct Wg7Red {
  const float* part[WG7_MULTI];
  float* dw[WG7_MULTI];
  float* db[WG7_MULTI];
  int nparts[WG7_MULTI];
};
template <int FMT>
__global__ __launch_bounds__(256, 2) void conv7_wgrad_mfma_multi_kernel(const Wg7Multi m) {
  int pi = 0;
  for (int k = 1; k < WG7_MULTI; ++k)
    if (k < m.n && (int)blockIdx.x >= m.off[k]) pi = k;
  conv7_wgrad_body<FMT>(m.p[pi], blockIdx.x - m.off[pi], m.off[pi + 1] - m.off[pi]);
}
__global__ __launch_bounds__(256) void conv7_wgrad_reduce_multi_kernel(const Wg7Red r) {
  const int i = blockIdx.x * 256 + threadIdx.x, pi = blockIdx.z;
  if (i >= K7_NW + 2) return;
  const float* part = r.part[pi];
  const int nparts = r.nparts[pi];
  float s = 0.f;
#pragma unroll 8
  for (int k = blockIdx.y; k < nparts; k += gridDim.y) s += part[(long long)k * K7_NPART + i];
  if (i < K7_NW) atomicAdd(&r.dw[pi][i], s);
  else if (r.db[pi]) atomicAdd(&r.db[pi][i - K7_NW], s);
}

// second stage: 2746 outputs x 32 slices of the partials; one float atomic per (output, slice)
__global__ __launch_bounds__(256) void conv7_wgrad_reduce_kernel(const float* part, int nparts, float* dw, float* db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K7_NW + 2) return;
  float s = 0.f;
#pragma unroll 8
  for (int k = blockIdx.y; k < nparts; k += gridDim.y) s += part[(long long)k * K7_NPART + i];
  if (i < K7_NW) atomicAdd(&dw[i], s);
  else if (db) atomicAdd(&db[i - K7_NW], s);
}

static bool wg7_eligible(const xh_conv_desc* d) {
  const bool f32s = d->dtype == XH_F32 && (d->arith & XH_ARITH_F32_SPLIT) && !(d->arith & XH_ARITH_K7_VECTOR);   // xh_conv_desc.arith
  if ((d->dtype != XH_BF16 && d->dtype != XH_F16 && !f32s) || d->k != 7 || d->stride != 1 || d->groups != 1 ||
      d->n_wptr != 1)
    return false;
  if (d->Cin != 4 || d->Cout != 2 || d->pre || d->Ca != d->Cin || d->transposed) return false;
  if (d->W % 32 != 0 || d->Wo != d->W || d->Ho != d->H || d->Do != d->D) return false;
  if ((d->xa_bs & 7) || (d->ea_bs & 7) || (((long long)d->D * d->H * d->W) & 7)) return false;
  return true;
}
static void wg7_plan(const xh_conv_desc* d, Wg7K* a) {
  a->tilesW = d->W / 32;
  a->tilesH = cdiv(d->H, 8);
  const int cols = a->tilesW * a->tilesH;
  int dsegs = cdiv(512, cols * d->N);
  // small volumes: more, shorter runs (conv7_mfma.hip); measured: 32^3 27.9 -> 22.7 us, but 64^3 29.7 -> 42.9 us (more partials
  // for the second-stage reduction), so only up to 32^3 here
  const int min_run = (long long)d->D * d->H * d->W <= (1 << 15) ? 2 : 8;
  const int max_segs = d->D >= min_run ? d->D / min_run : 1;
  if (dsegs > max_segs) dsegs = max_segs;
  if (dsegs < 1) dsegs = 1;
  a->sd = cdiv(d->D, dsegs);
  a->dsegs = cdiv(d->D, a->sd);
}
extern "C" long long xh_conv3d_wgrad_workspace_bytes(const xh_conv_desc* d) {
  if (!d || !wg7_eligible(d)) return 0;
  Wg7K a;
  wg7_plan(d, &a);
  return (long long)a.tilesW * a.tilesH * a.dsegs * d->N * K7_NPART * (long long)sizeof(float);
}

// returns XH_OK if launched, 1 if not eligible (caller falls back to the vector kernel)
int xh_conv7_wgrad_mfma_try(void* stream, const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR]) {
  if (!wg7_eligible(d)) return 1;
  const long long need = xh_conv3d_wgrad_workspace_bytes(d);
  if (!p->ws || p->ws_bytes < need) return 1;
  Wg7K a;
  a.d = *d;
  wg7_plan(d, &a);
  a.x = p->xa; a.x_bs = d->xa_bs;
  a.dy = p->ea; a.dy_bs = d->ea_bs;
  a.part = (float*)p->ws;
  const int nwg = a.tilesW * a.tilesH * a.dsegs * d->N;
  const size_t shm = W7_SHM;
  hipStream_t st = (hipStream_t)stream;
  const int fmt1 = d->dtype == XH_F32 ? 2 : d->dtype == XH_F16 ? 1 : 0;
  xh_note_kernel("conv7_wgrad_mfma_kernel<%d>", fmt1);
  if (fmt1 == 2) hipLaunchKernelGGL(conv7_wgrad_mfma_kernel<2>, dim3(nwg), dim3(256), shm, st, a);
  else if (fmt1 == 1) hipLaunchKernelGGL(conv7_wgrad_mfma_kernel<1>, dim3(nwg), dim3(256), shm, st, a);
  else hipLaunchKernelGGL(conv7_wgrad_mfma_kernel<0>, dim3(nwg), dim3(256), shm, st, a);
  hipLaunchKernelGGL(conv7_wgrad_reduce_kernel, dim3(cdiv(K7_NW + 2, 256), nwg < 32 ? nwg : 32), dim3(256), 0, st,
                     (const float*)a.part, nwg, dw[0], db ? db[0] : nullptr);
  return xh_launch_status();
}

// 7^3 weight gradients of a batch: WG7_MULTI per launch and storage format (marked in handled[]); see conv7_wgrad_mfma_multi_kernel
int xh_wg7_batch(void* stream, int n, const xh_conv_desc* const* d, const xh_conv_ptrs* const* p, float* const (*dw)[XH_MAX_WPTR],
                 float* const (*db)[XH_MAX_WPTR], char* handled) {
  extern int g_xh_disable;
  int xh_check_conv(const xh_conv_desc* d, const xh_conv_ptrs* p);
  if (g_xh_disable & 512) return XH_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t shm = W7_SHM;
  int rc_all = XH_OK;
  for (int fmt = 0; fmt < 3; ++fmt) {                 // bf16 / fp16 / fp32 storage with fp16 operands
    std::vector<int> idx;
    for (int i = 0; i < n; ++i) {
      if (handled[i] || !d[i] || !p[i] || !dw[i] || !dw[i][0] || !p[i]->xa || !p[i]->ea) continue;
      if ((d[i]->dtype == XH_F32 ? 2 : d[i]->dtype == XH_F16 ? 1 : 0) != fmt || !wg7_eligible(d[i]) || xh_check_conv(d[i], p[i])) continue;
      if (!p[i]->ws || p[i]->ws_bytes < xh_conv3d_wgrad_workspace_bytes(d[i])) continue;
      idx.push_back(i);
    }
    if (idx.size() < 2) continue;                       // a single problem: the ordinary entry point
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
      return (long long)d[a]->N * d[a]->D * d[a]->H * d[a]->W > (long long)d[b]->N * d[b]->D * d[b]->H * d[b]->W;
    });
    for (size_t i0 = 0; i0 < idx.size(); i0 += WG7_MULTI) {
      Wg7Multi m;
      Wg7Red r;
      m.n = (int)std::min<size_t>(WG7_MULTI, idx.size() - i0);
      m.off[0] = 0;
      int maxp = 1;
      for (int k = 0; k < WG7_MULTI; ++k) {
        if (k >= m.n) { m.off[k + 1] = m.off[m.n]; r.part[k] = nullptr; r.dw[k] = r.db[k] = nullptr; r.nparts[k] = 0; continue; }
        const int i = idx[i0 + k];
        Wg7K& a = m.p[k];
        a.d = *d[i];
        wg7_plan(d[i], &a);
        a.x = p[i]->xa; a.x_bs = d[i]->xa_bs;
        a.dy = p[i]->ea; a.dy_bs = d[i]->ea_bs;
        a.part = (float*)p[i]->ws;
        const int nwg = a.tilesW * a.tilesH * a.dsegs * d[i]->N;
        m.off[k + 1] = m.off[k] + nwg;
        r.part[k] = a.part; r.dw[k] = dw[i][0]; r.db[k] = db ? db[i][0] : nullptr; r.nparts[k] = nwg;
        if (nwg > maxp) maxp = nwg;
        handled[i] = 1;
      }
      xh_note_kernel("conv7_wgrad_mfma_multi_kernel<%d>", fmt);
      if (fmt == 2) hipLaunchKernelGGL(conv7_wgrad_mfma_multi_kernel<2>, dim3(m.off[m.n]), dim3(256), shm, st, m);
      else if (fmt) hipLaunchKernelGGL(conv7_wgrad_mfma_multi_kernel<1>, dim3(m.off[m.n]), dim3(256), shm, st, m);
      else hipLaunchKernelGGL(conv7_wgrad_mfma_multi_kernel<0>, dim3(m.off[m.n]), dim3(256), shm, st, m);
      hipLaunchKernelGGL(conv7_wgrad_reduce_multi_kernel, dim3(cdiv(K7_NW + 2, 256), maxp < 32 ? maxp : 32, m.n), dim3(256), 0, st, r);
      if (xh_launch_status() != XH_OK) rc_all = XH_ERR_HIP;
    }
  }
  return rc_all;
}
