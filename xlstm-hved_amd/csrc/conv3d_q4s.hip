// Quad-channel 3x3x3 stride-1 convolution (forward and data gradient) for FP32 STORAGE on the matrix cores of gfx950:
// the W-Toeplitz GEMM of conv3d_q4.hip with TWO-TERM fp16 operands.  The fp32 storage mode is the parity mode of this path
// (Dice within 1e-4 of the reference); its convolutions ran on LDS-tiled fp32 FMA kernels at ~12 % of the vector
// peak (conv3d.hip: 5.5 of the mode's 14 ms per step).  gfx950 has no fp32-input MFMA faster than the vector units, but
//
//     x = hi + 2^-11 lo,   hi = fp16(x),   lo = fp16((x - hi) 2^11)        (22 significand bits; lo stays a normal fp16)
//     x w = hi_x hi_w + 2^-11 (hi_x lo_w + lo_x hi_w) + 2^-22 lo_x lo_w    (the last term is dropped: relative 2^-22)
//
// turns one fp32 product into THREE fp16 MFMA products accumulated in two fp32 tiles (main, correction).  The bf16 / fp16
// kernel runs its matrix cores ~11 % of the time (profiles/r03e_pmc_sq.json), so three times the MFMAs are affordable; what
// doubles is the LDS image (hi and lo planes) and the bytes.  Everything else is conv3d_q4.hip's: staging plan, fragment
// addresses, accumulator layout = store layout, in-kernel InstanceNorm finalisation, epilogue statistics through the fan-in.
//
// Range.  Forward operands are normalised activations and weights: O(1).  In the BACKWARD pass the data operand is an
// activation gradient; like fp16 STORAGE it needs the caller's loss scale to sit inside fp16's range (bench.py / TrainStep
// apply 65536 exactly as for fp16 storage, and unscale the fp32 parameter gradients).  Opt-in: xh_set_option(18, 1).
//
// Workgroup = TD x 8 x 32 outputs of one output-channel quad, TD = 4 (2 on small launches): two LDS images of (TD + 2) planes.
#include "conv_q4.h"

namespace {
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// fp32 pair -> (hi, lo) packed fp16 pairs
__device__ __forceinline__ void split2(f32x2_t v, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2_t r = (v - __builtin_convertvector(h, f32x2_t)) * f32x2_t{2048.f, 2048.f};
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ f32x4 mfma_h(frag8 a, frag8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
}

template <int PRE, int EPI, bool MULTI, int TD>
__global__ __launch_bounds__(256, 3) void conv3_q4s_kernel(const ConvQ4 a) {
  constexpr int ID = TD + 2, IMG = ID * PLANE, TILE_BYTES = 2 * IMG, NROWS = ID * IH, NITEM = NROWS * 4, NEDGE = NROWS * 2;
  static_assert(NITEM <= 256 && NEDGE <= 256, "one interior item and at most one edge item per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* s_red = reinterpret_cast<double*>(smem + TILE_BYTES);
  float* s_fin = reinterpret_cast<float*>(smem + TILE_BYTES + 48 * sizeof(double));

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nn = lane & 15, g4 = lane >> 4;
  const int oq = blockIdx.y, n = blockIdx.z;
  const int co0 = oq * 4;
  const int grp = udiv_fast(oq, a.oq_g, a.mQ);
  const int cin_base = grp * a.Cin_g;
  const int D = a.d.D, H = a.d.H, W = a.d.W;
  const long long hw = (long long)H * W, dhw = (long long)D * hw;
  const int Do = a.d.Do, Ho = a.d.Ho;
  const int wk = xcd_swizzle(blockIdx.x, gridDim.x);
  const int wk1 = udiv_fast(wk, a.tilesW, a.mW), tw = wk - wk1 * a.tilesW;
  const int td = udiv_fast(wk1, a.tilesH, a.mH), th = wk1 - td * a.tilesH;
  const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
  double fs1 = 0.0, fs2 = 0.0;
  if (PRE == 1 && a.p.fin_red && tid < a.Cin_g) {
    fs1 = a.p.fin_red[2 * (n * a.d.Cin + cin_base + tid)];
    fs2 = a.p.fin_red[2 * (n * a.d.Cin + cin_base + tid) + 1];
  }

  // ---- staging plan: thread tid owns interior item tid (row, 8-voxel group) and, the last NEDGE threads, an edge pair ----
  const bool i_do = tid < NITEM;
  unsigned i_off; int i_lds, i_par; bool i_live;
  {
    const int item = min(tid, NITEM - 1);
    const int gq = item & 3, row = item >> 2;
    const int dz = row / IH, hy = row - dz * IH;
    const int gd = od0 - 1 + dz, gh = oh0 - 1 + hy;
    i_live = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H;
    const int gdc = min(max(gd, 0), D - 1), ghc = min(max(gh, 0), H - 1);
    i_off = (unsigned)((((long long)gdc * H + ghc) * W + ow0 + gq * 8) * 4ll);
    i_lds = row * PITCH;
    i_par = (row & 1) | ((1 + 4 * gq) << 1);
  }
  unsigned e_off; int e_lds; bool e_live;
  const bool e_do = tid >= 256 - NEDGE;
  {
    const int ei = max(255 - tid, 0) < NEDGE ? 255 - tid : 0;
    const int row = ei >> 1, side = ei & 1;
    const int dz = row / IH, hy = row - dz * IH;
    const int gd = od0 - 1 + dz, gh = oh0 - 1 + hy;
    const int gw = side ? ow0 + TW : ow0 - 2;
    e_live = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw >= 0 && gw < W;
    const int gdc = min(max(gd, 0), D - 1), ghc = min(max(gh, 0), H - 1), gwc = min(max(gw, 0), W - 2);
    e_off = (unsigned)((((long long)gdc * H + ghc) * W + gwc) * 4ll);
    e_lds = row * PITCH + (((side ? 17 : 0) ^ (row & 1)) << 4);
  }
  const int ur = (nn >> 2) & 1;
  const int qw = (nn >> 3) | ((nn & 3) << 1);
  const int rowbase = (2 * wv + ur) * PITCH;
  const int b_off0 = rowbase + ((2 * qw + g4) ^ ur) * 16;
  const int b_off1 = rowbase + ((2 * qw + g4) ^ ur ^ 1) * 16;
  const int oh = oh0 + 2 * wv + ur;
  const bool row_ok = oh < Ho;
  const int ndz = min(TD, Do - od0);
  float bias = 0.f, esc = 0.f, esh = 0.f;
  {
    const int wp = udiv_fast(grp, a.gpp, a.mG);
    const float* bp = a.p.b[wp];
    if (bp) bias = bp[(grp - wp * a.gpp) * a.Cout_g + (oq - grp * a.oq_g) * 4 + g4];
  }
  const long long odhw = (long long)Do * Ho * a.d.Wo;
  const unsigned spd_b = (unsigned)(Ho * a.d.Wo) * 4u;
  const unsigned lane_b = (unsigned)(((long long)g4 * odhw + (long long)(row_ok ? oh : 0) * a.d.Wo + ow0 + 4 * qw) * 4ll);
  const unsigned lane_bo = row_ok ? lane_b : Q4_OOB;
  __amdgpu_buffer_rsrc_t ers = q4_window(a.p.y), yrs;
  if (EPI == 1) {
    esc = a.p.e_sc[n * a.d.Cout + co0 + g4];
    esh = a.p.e_sh[n * a.d.Cout + co0 + g4];
    ers = q4_window(reinterpret_cast<const char*>(co0 < a.d.Cea ? (const float*)a.p.ea + n * a.d.ea_bs + (long long)co0 * odhw
                                                                  : (const float*)a.p.eb + n * a.d.eb_bs + (long long)(co0 - a.d.Cea) * odhw) +
                    (long long)od0 * spd_b);
  }
  yrs = q4_window(reinterpret_cast<char*>((float*)a.p.y + n * a.d.y_bs + (long long)co0 * odhw) + (long long)od0 * spd_b);

  f32x4 acc[TD], accl[TD];                             // main products, and the two cross products (x 2^-11)
#pragma unroll
  for (int i = 0; i < TD; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int ncq = MULTI ? a.ci4 : 1;
  const float pslope = a.d.pre_slope;
  const bool fin = PRE == 1 && a.p.fin_red != nullptr;
  const long long nfrag = (long long)gridDim.y * a.ci4 * 9 * 64;          // fragments of one image (hi), the lo image follows
  for (int cq = 0; cq < ncq; ++cq) {
    const int c0 = cin_base + cq * 4;
    const char* src = reinterpret_cast<const char*>(c0 < a.d.Ca ? (const float*)a.p.xa + n * a.d.xa_bs + (long long)c0 * dhw
                                                                  : (const float*)a.p.xb + n * a.d.xb_bs + (long long)(c0 - a.d.Ca) * dhw);
    const long long dhw_b = dhw * 4ll;
    uint4 raw[4][2];
    uint2 eraw[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      raw[cc][0] = *reinterpret_cast<const uint4*>(src + cc * dhw_b + i_off);
      raw[cc][1] = *reinterpret_cast<const uint4*>(src + cc * dhw_b + i_off + 16);
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) eraw[cc] = *reinterpret_cast<const uint2*>(src + cc * dhw_b + e_off);
    if (cq > 0) __syncthreads();
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (PRE == 1) {
      if (fin) {
        if (cq == 0) {
          const float* gam = a.p.fin_gamma;
          const float* bet = a.p.fin_beta;
          if (tid < a.Cin_g) {
            float m_, r_, sc_, sh_;
            in_finalize(fs1, fs2, a.fin_inv, sc_, sh_, m_, r_);
            if (gam) { const float g_ = gam[cin_base + tid]; sc_ *= g_; sh_ = fmaf(sh_, g_, bet[cin_base + tid]); }
            s_fin[tid] = sc_; s_fin[Q4_MAXC + tid] = sh_;
          }
          if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
            for (int i = tid; i < a.d.N * a.d.Cin; i += 256) {
              float sc_, sh_, m_, r_;
              in_finalize(a.p.fin_red[2 * i], a.p.fin_red[2 * i + 1], a.fin_inv, sc_, sh_, m_, r_);
              if (gam) {
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
    // ---- transform in fp32, split, channels-last hi / lo images ----
    if (i_do) {
      const int par = i_par & 1, slot0 = i_par >> 1;
      const float lv = i_live ? 1.f : 0.f;
      f32x2_t v[4][4];                                  // [channel][voxel pair]
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float s1 = sc[cc] * lv, s2 = sh[cc] * lv;
        const unsigned u[8] = {raw[cc][0].x, raw[cc][0].y, raw[cc][0].z, raw[cc][0].w, raw[cc][1].x, raw[cc][1].y, raw[cc][1].z, raw[cc][1].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f32x2_t t = f32x2_t{__uint_as_float(u[2 * k]), __uint_as_float(u[2 * k + 1])};
          t = t * f32x2_t{s1, s1} + f32x2_t{s2, s2};
          if (PRE == 1) t = max2(t, t * f32x2_t{pslope, pslope});
          v[cc][k] = t;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 oh_, ol_;
        split2(f32x2_t{v[0][j].x, v[1][j].x}, oh_.x, ol_.x);
        split2(f32x2_t{v[2][j].x, v[3][j].x}, oh_.y, ol_.y);
        split2(f32x2_t{v[0][j].y, v[1][j].y}, oh_.z, ol_.z);
        split2(f32x2_t{v[2][j].y, v[3][j].y}, oh_.w, ol_.w);
        const int ad = i_lds + (((slot0 + j) ^ par) << 4);
        *reinterpret_cast<uint4*>(smem + ad) = oh_;
        *reinterpret_cast<uint4*>(smem + IMG + ad) = ol_;
      }
    }
    if (e_do) {
      const float lv = e_live ? 1.f : 0.f;
      f32x2_t v[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        f32x2_t t = f32x2_t{__uint_as_float(eraw[cc].x), __uint_as_float(eraw[cc].y)};
        t = t * f32x2_t{sc[cc] * lv, sc[cc] * lv} + f32x2_t{sh[cc] * lv, sh[cc] * lv};
        if (PRE == 1) t = max2(t, t * f32x2_t{pslope, pslope});
        v[cc] = t;
      }
      uint4 oh_, ol_;
      split2(f32x2_t{v[0].x, v[1].x}, oh_.x, ol_.x);
      split2(f32x2_t{v[2].x, v[3].x}, oh_.y, ol_.y);
      split2(f32x2_t{v[0].y, v[1].y}, oh_.z, ol_.z);
      split2(f32x2_t{v[2].y, v[3].y}, oh_.w, ol_.w);
      *reinterpret_cast<uint4*>(smem + e_lds) = oh_;
      *reinterpret_cast<uint4*>(smem + IMG + e_lds) = ol_;
    }
    // weight fragments (hi and lo) of this (output quad, input quad)
    frag8 wh[9], wl[9];
    {
      const frag8* wpk = reinterpret_cast<const frag8*>(a.p.ws) + ((long long)oq * a.ci4 + cq) * 9 * 64 + lane;
#pragma unroll
      for (int i = 0; i < 9; ++i) { wh[i] = wpk[i * 64]; wl[i] = wpk[nfrag + i * 64]; }
    }
    __syncthreads();
#pragma unroll
    for (int pz = 0; pz < ID; ++pz) {
      frag8 bh[3], bl[3];
      bh[0] = *reinterpret_cast<const frag8*>(smem + pz * PLANE + b_off0);
      bh[1] = *reinterpret_cast<const frag8*>(smem + pz * PLANE + PITCH + b_off1);
      bh[2] = *reinterpret_cast<const frag8*>(smem + pz * PLANE + 2 * PITCH + b_off0);
      bl[0] = *reinterpret_cast<const frag8*>(smem + IMG + pz * PLANE + b_off0);
      bl[1] = *reinterpret_cast<const frag8*>(smem + IMG + pz * PLANE + PITCH + b_off1);
      bl[2] = *reinterpret_cast<const frag8*>(smem + IMG + pz * PLANE + 2 * PITCH + b_off0);
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int dz = pz - kd;
        if (dz < 0 || dz >= TD) continue;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          acc[dz] = mfma_h(wh[kd * 3 + kh], bh[kh], acc[dz]);
          accl[dz] = mfma_h(wh[kd * 3 + kh], bl[kh], accl[dz]);
          accl[dz] = mfma_h(wl[kd * 3 + kh], bh[kh], accl[dz]);
        }
      }
    }
  }

  // ---- epilogue: y = main + 2^-11 correction + bias; 16-byte fp32 stores; statistics as stored ----
  f32x2_t ps = {0.f, 0.f}, pq = {0.f, 0.f};
  const float eslope = a.d.e_slope;
  f32x4 er[TD];
  if (EPI == 1) {
#pragma unroll
    for (int dz = 0; dz < TD; ++dz)
      er[dz] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ers, (int)lane_b, (int)((unsigned)min(dz, ndz - 1) * spd_b), 0));
  }
#pragma unroll
  for (int dz = 0; dz < TD; ++dz) {
    const bool live = dz < ndz;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaf(accl[dz][r], 1.f / 2048.f, acc[dz][r]) + bias;
    if (EPI == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = fmaf(er[dz][r], esc, esh);
        v[r] = z > 0.f ? v[r] : v[r] * eslope;
      }
      if (live) {
        ps += f32x2_t{v[0] + v[1], v[2] + v[3]};
        pq += f32x2_t{v[0] * er[dz][0] + v[1] * er[dz][1], v[2] * er[dz][2] + v[3] * er[dz][3]};
      }
    } else if (EPI == 2 && live) {
      ps += f32x2_t{v[0] + v[1], v[2] + v[3]};
      pq += f32x2_t{v[0] * v[0] + v[1] * v[1], v[2] * v[2] + v[3] * v[3]};
    }
    // (two 8-byte stores: a 16-byte buffer store followed by the DPP sums below was seen to write stale data for a few lanes of
    // the workgroup's last wave, run-to-run varying -- the store's data registers were reused before it had read them)
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned so = live ? lane_bo : Q4_OOB;
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}, yrs, (int)so, (int)((unsigned)dz * spd_b), 0);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v[2]), __float_as_uint(v[3])}, yrs, (int)(live ? lane_bo + 8u : Q4_OOB),
                                          (int)((unsigned)dz * spd_b), 0);
  }
  asm volatile("s_nop 7" ::: "memory");
  if (EPI) {
    // fp32 storage: the per-lane partial sums of up to 4 * TD values go to fp64 right away
    const double t0 = row16_sum(row_ok ? (double)ps.x + (double)ps.y : 0.0), t1 = row16_sum(row_ok ? (double)pq.x + (double)pq.y : 0.0);
    if (nn == 0) { s_red[wv * 8 + g4 * 2] = t0; s_red[wv * 8 + g4 * 2 + 1] = t1; }
    __syncthreads();
    if (tid < 8) {
      const double tot = s_red[tid] + s_red[8 + tid] + s_red[16 + tid] + s_red[24 + tid];
      s_red[32 + tid] = tot;
    }
    double* s_tot = s_red + 32;
    if (a.fan && !fan_in<8>(a.fan + ((long long)n * gridDim.y + oq) * FAN_UNIT_BYTES, blockIdx.x, gridDim.x, s_tot,
                            reinterpret_cast<int*>(s_tot + 8)))
      return;
    if (!a.fan) __syncthreads();
    if (tid < 8) atomicAdd(&a.p.red[((long long)n * a.d.Cout + co0 + (tid >> 1)) * 2 + (tid & 1)], s_tot[tid]);
  }
}

// launch (plan and fan block prepared by conv3d_q4.hip: xh_conv3_q4_try)
int xh_conv3_q4s_launch(hipStream_t st, const ConvQ4& a, dim3 grid) {
  const xh_conv_desc* d = &a.d;
  const size_t shm = 2 * q4_tile_bytes(a.td) + 48 * sizeof(double) + 3 * Q4_MAXC * sizeof(float);
  xh_note_kernel("conv3_q4s_kernel<%d, %d, %s, %d>", d->pre, d->epi, a.ci4 > 1 ? "true" : "false", a.td);
#define QSL(P, E, T)                                                                                   \
  do {                                                                                                  \
    if (a.ci4 > 1) hipLaunchKernelGGL((conv3_q4s_kernel<P, E, true, T>), grid, dim3(256), shm, st, a);  \
    else hipLaunchKernelGGL((conv3_q4s_kernel<P, E, false, T>), grid, dim3(256), shm, st, a);           \
  } while (0)
#define QST(P, E)                      \
  do {                                 \
    if (a.td == 4) QSL(P, E, 4);       \
    else QSL(P, E, 2);                 \
  } while (0)
#define QSE(P)                           \
  do {                                   \
    if (d->epi == 0) QST(P, 0);          \
    else if (d->epi == 1) QST(P, 1);     \
    else QST(P, 2);                      \
  } while (0)
  if (d->pre) QSE(1);
  else QSE(0);
#undef QSE
#undef QST
#undef QSL
  return xh_launch_status();
}
