// Weight-fragment packing shared by the k=3 MFMA convolution kernels (conv3d_mfma.hip: implicit-GEMM fragments;
// conv3d_q4.hip: quad-channel W-Toeplitz fragments), and the in-kernel InstanceNorm finalisation they both use.
//
// Weights are constant within a training step, so a caller may pack the fragments of EVERY convolution of the step with one
// launch up front (xh_conv3d_prepack, conv3_pack_multi_kernel) instead of one small launch in front of each convolution; the
// job record below is everything a pack needs, small enough for two dozen of them to travel in the kernel arguments.
#pragma once
#include "../../include/xlstm_hved.h"
#include "common.h"

struct PackJob {
  const float* w[XH_MAX_WPTR];
  void* ws;
  int kind;                                  // 0: implicit-GEMM fragments, 1: quad-channel Toeplitz fragments
  int f16, groups, n_wptr, transposed, Cin_g, Cout_g;
  int ntile, cin_stride, cin_off, cin_blk, cout_set, nm, nch, cpr, cinp;   // kind 0 (ConvMK fields of the same names)
  int ci4;                                                                  // kind 1
  int dw;                                    // kind 1: depthwise conv run as groups of 4 with diagonal 4 x 4 weights (w: [C][1][27])
  int nelem;                                 // 16-bit elements to write
};

// weight of (output channel co, input channel ci, tap) in kernel view (absolute channels; block-diagonal over groups);
// transposed = data gradient: forward-layout weights with roles swapped and taps flipped
__device__ __forceinline__ float pack_weight(const PackJob& j, int co, int ci, int tap) {
  const int g = co / j.Cout_g;
  if (ci / j.Cin_g != g) return 0.f;
  const int co_g = co % j.Cout_g, ci_g = ci % j.Cin_g;
  const int gpp = j.groups / j.n_wptr;
  const float* wp = j.w[g / gpp];
  const int gl = g % gpp;
  if (j.dw) return co_g == ci_g ? wp[(long long)(gl * 4 + co_g) * 27 + (j.transposed ? 26 - tap : tap)] : 0.f;
  if (!j.transposed) return wp[((long long)(gl * j.Cout_g + co_g) * j.Cin_g + ci_g) * 27 + tap];
  return wp[((long long)(gl * j.Cin_g + ci_g) * j.Cout_g + co_g) * 27 + (26 - tap)];
}

__device__ __forceinline__ void pack_elem(const PackJob& j, int idx) {
  float v = 0.f;
  if (j.kind == 0) {
    // ws[y][i][lane][8] = B fragment (k = 8*(lane>>4)..+7, col lane&15) of MFMA i for channel tile y = set*ntile + nt
    const int per = j.nm * 512;
    const int y = idx / per, r = idx - y * per;
    const int set = y / j.ntile, nt = y % j.ntile;
    const int cin0 = set * j.cin_stride + j.cin_off;
    const int cin_end = (set + 1) * j.cin_stride;
    const int co_base = set * j.cout_set + nt * 16;
    const int co_lim = min(16, j.cout_set - nt * 16);
    const int e = r & 7, l = (r >> 3) & 63, i = r >> 9;
    const int c = 4 * i + (l >> 4);
    if (c < j.nch && (l & 15) < co_lim) {
      const int r9 = c / j.cpr, q = c % j.cpr;
      const int flat = q * 8 + e;                       // position inside the row segment: kw*CINP + ci
      const int kw = flat / j.cinp, ci = flat % j.cinp;
      if (kw < 3 && ci < j.cin_blk && cin0 + ci < cin_end) v = pack_weight(j, co_base + (l & 15), cin0 + ci, r9 * 3 + kw);
    }
  } else {
    // ws[((oq * ci4 + cq) * 9 + r9) * 64 + lane][8]: A fragment of (kd, kh) = r9 for output quad oq, input quad cq:
    // row m = (c, p) = output channel c of the quad, position p in a voxel quad; k = (s, ci) = input voxel 4q - 2 + s, channel ci
    const int per = j.ci4 * 9 * 512;
    const int oq = idx / per, r = idx - oq * per;
    const int grp = (oq * 4) / j.Cout_g;
    const int e = r & 7, l = (r >> 3) & 63, f = r >> 9;
    const int r9 = f % 9, cq = f / 9;
    const int m = l & 15, c = m >> 2, pp = m & 3, g = l >> 4;
    const int s = 2 * g + (e >> 2), ci = e & 3;
    const int kw = s - pp - 1;
    if (kw >= 0 && kw <= 2) v = pack_weight(j, oq * 4 + c, grp * j.Cin_g + cq * 4 + ci, r9 * 3 + kw);
  }
  if (j.f16 == 2) {                                     // two-term fp16 image of an fp32 weight: v = hi + 2^-11 lo (conv3d_q4s.hip)
    const unsigned short hi = f2hf(v);
    reinterpret_cast<unsigned short*>(j.ws)[idx] = hi;
    reinterpret_cast<unsigned short*>(j.ws)[(long long)j.nelem + idx] = f2hf((v - hf2f(hi)) * 2048.f);
    return;
  }
  reinterpret_cast<unsigned short*>(j.ws)[idx] = j.f16 ? f2hf(v) : f2bf(v);
}

// Eight consecutive elements of a quad-channel (kind 1) image -- one A-fragment lane: they share output channel, input quad, (kd, kh)
// and voxel-quad position, so the index arithmetic (a dozen integer divisions per element in pack_elem) is done once and the lane's
// 16 bytes are stored at once.  idx8 is a multiple of 8.  Same values as pack_elem.
__device__ __forceinline__ void pack_elem8_q4(const PackJob& j, int idx8) {
  const int per = j.ci4 * 9 * 512;
  const int oq = idx8 / per, r = idx8 - oq * per;
  const int l = (r >> 3) & 63, f = r >> 9;
  const int r9 = f % 9, cq = f / 9;
  const int m = l & 15, c = m >> 2, pp = m & 3, g = l >> 4;
  const int co = oq * 4 + c;
  const int grp = co / j.Cout_g, co_g = co - grp * j.Cout_g;
  const int gpp = j.groups / j.n_wptr;
  const float* wp = j.w[grp / gpp];
  const int gl = grp % gpp;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int s = 2 * g + (e >> 2), ci = e & 3;
    const int kw = s - pp - 1;
    float x = 0.f;
    if (kw >= 0 && kw <= 2) {
      const int ci_g = cq * 4 + ci, tap = r9 * 3 + kw;
      if (j.dw) x = co_g == ci_g ? wp[(long long)(gl * 4 + co_g) * 27 + (j.transposed ? 26 - tap : tap)] : 0.f;
      else if (!j.transposed) x = wp[((long long)(gl * j.Cout_g + co_g) * j.Cin_g + ci_g) * 27 + tap];
      else x = wp[((long long)(gl * j.Cin_g + ci_g) * j.Cout_g + co_g) * 27 + (26 - tap)];
    }
    v[e] = x;
  }
  unsigned short* ws = reinterpret_cast<unsigned short*>(j.ws);
  if (j.f16 == 2) {                                     // two-term fp16 image of an fp32 weight (conv3d_q4s.hip)
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { hi[e] = f2hf(v[e]); lo[e] = f2hf((v[e] - hf2f(hi[e])) * 2048.f); }
    uint4 a, b;
    a.x = hi[0] | ((unsigned)hi[1] << 16); a.y = hi[2] | ((unsigned)hi[3] << 16); a.z = hi[4] | ((unsigned)hi[5] << 16); a.w = hi[6] | ((unsigned)hi[7] << 16);
    b.x = lo[0] | ((unsigned)lo[1] << 16); b.y = lo[2] | ((unsigned)lo[3] << 16); b.z = lo[4] | ((unsigned)lo[5] << 16); b.w = lo[6] | ((unsigned)lo[7] << 16);
    *reinterpret_cast<uint4*>(ws + idx8) = a;
    *reinterpret_cast<uint4*>(ws + (long long)j.nelem + idx8) = b;
    return;
  }
  uint4 o;
  if (j.f16) { o.x = cvt_pack<1>(v[0], v[1]); o.y = cvt_pack<1>(v[2], v[3]); o.z = cvt_pack<1>(v[4], v[5]); o.w = cvt_pack<1>(v[6], v[7]); }
  else { o.x = cvt_pack<0>(v[0], v[1]); o.y = cvt_pack<0>(v[2], v[3]); o.z = cvt_pack<0>(v[4], v[5]); o.w = cvt_pack<0>(v[6], v[7]); }
  *reinterpret_cast<uint4*>(ws + idx8) = o;
}

// The quad-channel image again, with the weights of the workgroup's FOUR fragments (4 co x 4 ci x 3 kw of one (kd, kh) each = 48
// floats) staged in LDS first: pack_elem8_q4 issues 8 scattered 4-byte loads per lane -- 2 048 per workgroup for 192 distinct values
// (the Toeplitz expansion repeats every weight ~10 times), and the launch was bound by those load instructions, not by its 6 MB of
// output.  s_w: 4 x 48 floats; the caller has put a barrier behind stage.  Same values as pack_elem8_q4.
__device__ __forceinline__ void pack_q4_stage(const PackJob& j, int base, float* s_w) {
  const int t = threadIdx.x;
  if (t >= 192) return;
  const int fr = t / 48, i48 = t - fr * 48;
  const int idx = base + fr * 512;
  float x = 0.f;
  if (idx < j.nelem) {
    const int per = j.ci4 * 9 * 512;
    const int oq = idx / per, f = (idx - oq * per) >> 9;
    const int r9 = f % 9, cq = f / 9;
    const int cc = i48 / 12, cii = (i48 / 3) & 3, kw = i48 % 3;       // s_w[fragment][co of the quad][ci of the quad][kw]
    const int co = oq * 4 + cc;
    const int grp = co / j.Cout_g, co_g = co - grp * j.Cout_g;
    const int gpp = j.groups / j.n_wptr;
    const float* wp = j.w[grp / gpp];
    const int gl = grp % gpp;
    const int ci_g = cq * 4 + cii, tap = r9 * 3 + kw;
    if (j.dw) x = co_g == ci_g ? wp[(long long)(gl * 4 + co_g) * 27 + (j.transposed ? 26 - tap : tap)] : 0.f;
    else if (!j.transposed) x = wp[((long long)(gl * j.Cout_g + co_g) * j.Cin_g + ci_g) * 27 + tap];
    else x = wp[((long long)(gl * j.Cin_g + ci_g) * j.Cout_g + co_g) * 27 + (26 - tap)];
  }
  s_w[t] = x;
}
__device__ __forceinline__ void pack_elem8_q4_lds(const PackJob& j, int idx8, const float* s_w) {
  const int l = (idx8 >> 3) & 63;
  const float* sf = s_w + (threadIdx.x >> 6) * 48;       // thread t packs fragment t / 64 of the workgroup's four
  const int m = l & 15, c = m >> 2, pp = m & 3, g = l >> 4;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int s = 2 * g + (e >> 2), ci = e & 3;
    const int kw = s - pp - 1;
    v[e] = (kw >= 0 && kw <= 2) ? sf[c * 12 + ci * 3 + kw] : 0.f;
  }
  unsigned short* ws = reinterpret_cast<unsigned short*>(j.ws);
  if (j.f16 == 2) {                                     // two-term fp16 image of an fp32 weight (conv3d_q4s.hip)
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { hi[e] = f2hf(v[e]); lo[e] = f2hf((v[e] - hf2f(hi[e])) * 2048.f); }
    uint4 a, b;
    a.x = hi[0] | ((unsigned)hi[1] << 16); a.y = hi[2] | ((unsigned)hi[3] << 16); a.z = hi[4] | ((unsigned)hi[5] << 16); a.w = hi[6] | ((unsigned)hi[7] << 16);
    b.x = lo[0] | ((unsigned)lo[1] << 16); b.y = lo[2] | ((unsigned)lo[3] << 16); b.z = lo[4] | ((unsigned)lo[5] << 16); b.w = lo[6] | ((unsigned)lo[7] << 16);
    *reinterpret_cast<uint4*>(ws + idx8) = a;
    *reinterpret_cast<uint4*>(ws + (long long)j.nelem + idx8) = b;
    return;
  }
  uint4 o;
  if (j.f16) { o.x = cvt_pack<1>(v[0], v[1]); o.y = cvt_pack<1>(v[2], v[3]); o.z = cvt_pack<1>(v[4], v[5]); o.w = cvt_pack<1>(v[6], v[7]); }
  else { o.x = cvt_pack<0>(v[0], v[1]); o.y = cvt_pack<0>(v[2], v[3]); o.z = cvt_pack<0>(v[4], v[5]); o.w = cvt_pack<0>(v[6], v[7]); }
  *reinterpret_cast<uint4*>(ws + idx8) = o;
}

// The same for the implicit-GEMM (kind 0) image: eight consecutive elements are one lane's B-fragment row segment -- they share the
// channel tile, the MFMA index, the column (output channel) and the (kd, kh) row; only (kw, ci) = divmod(q * 8 + e, cinp) moves.
__device__ __forceinline__ void pack_elem8_mk(const PackJob& j, int idx8) {
  const int per = j.nm * 512;
  const int y = idx8 / per, r = idx8 - y * per;
  const int set = y / j.ntile, nt = y - set * j.ntile;
  const int cin0 = set * j.cin_stride + j.cin_off;
  const int cin_end = (set + 1) * j.cin_stride;
  const int co_lim = min(16, j.cout_set - nt * 16);
  const int l = (r >> 3) & 63, i = r >> 9;
  const int c = 4 * i + (l >> 4);
  // EVERY load is issued, from an address clamped into the weight tensor, and the element is selected afterwards: with the loads
  // under their validity tests hipcc waits for each one where it is issued -- eight exposed latencies in a row per lane, and the
  // 107 workgroups of these images were the long pole (~20 us) of the step's pack launch
  float v[8];
  {
    const bool lane_ok = c < j.nch && (l & 15) < co_lim;
    const int cc = lane_ok ? c : 0;
    const int r9 = cc / j.cpr, q = cc - r9 * j.cpr;
    const int co = lane_ok ? set * j.cout_set + nt * 16 + (l & 15) : 0;
    const int g = co / j.Cout_g, co_g = co - g * j.Cout_g;
    const int gpp = j.groups / j.n_wptr;
    const float* wp = j.w[g / gpp];
    const int gl = g % gpp;
    long long off[8];
    bool ok[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int flat = q * 8 + e;
      const int kw = flat / j.cinp, cib = flat - kw * j.cinp;
      const int ci = cin0 + cib;
      bool k = lane_ok && kw < 3 && cib < j.cin_blk && ci < cin_end && ci / j.Cin_g == g;
      const int ci_g = k ? ci - g * j.Cin_g : 0, tap = k ? r9 * 3 + kw : 0;
      if (j.dw) { k = k && co_g == ci_g; off[e] = (long long)(gl * 4 + co_g) * 27 + (j.transposed ? 26 - tap : tap); }
      else if (!j.transposed) off[e] = ((long long)(gl * j.Cout_g + co_g) * j.Cin_g + ci_g) * 27 + tap;
      else off[e] = ((long long)(gl * j.Cin_g + ci_g) * j.Cout_g + co_g) * 27 + (26 - tap);
      ok[e] = k;
    }
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = wp[off[e]];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ok[e] ? x[e] : 0.f;
  }
  unsigned short* ws = reinterpret_cast<unsigned short*>(j.ws);
  uint4 o;
  if (j.f16) { o.x = cvt_pack<1>(v[0], v[1]); o.y = cvt_pack<1>(v[2], v[3]); o.z = cvt_pack<1>(v[4], v[5]); o.w = cvt_pack<1>(v[6], v[7]); }
  else { o.x = cvt_pack<0>(v[0], v[1]); o.y = cvt_pack<0>(v[2], v[3]); o.z = cvt_pack<0>(v[4], v[5]); o.w = cvt_pack<0>(v[6], v[7]); }
  *reinterpret_cast<uint4*>(ws + idx8) = o;
}

#define XH_PACK_MAX_JOBS 24
struct PackMulti {
  PackJob job[XH_PACK_MAX_JOBS];
  int first_block[XH_PACK_MAX_JOBS + 1];     // workgroup prefix: job i owns blocks [first_block[i], first_block[i+1])
  int n;
};
#define XH_PACK_PER_BLOCK 2048                // elements per workgroup (256 threads x 8)

// InstanceNorm statistics of a channel from its raw sums (s1 = sum x, s2 = sum x^2; xh_conv_ptrs.fin_red): mean and
// variance in fp64 (the cancellation in E[x^2] - mean^2 needs it), the reciprocal square root in fp32 with one Newton step
// (<= 1 ulp of the fp32 value xh_norm_finalize rounds to; these kernels feed 16-bit storage).  Cheap enough for every
// workgroup to evaluate for its own channels; all of them get the same bits.
__device__ __forceinline__ void in_finalize(double s1, double s2, double inv_count, float& sc, float& sh, float& mean, float& rstd) {
  const double m = s1 * inv_count;
  double var = fma(-m, m, s2 * inv_count);
  if (var < 0) var = 0;
  const float v = (float)(var + 1e-5);
  float r = __frsqrt_rn(v);
  r = r * (1.5f - 0.5f * v * r * r);
  sc = r;
  sh = (float)(-m * (double)r);
  mean = (float)m;
  rstd = r;
}
