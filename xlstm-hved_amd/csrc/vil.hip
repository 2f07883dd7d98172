// ViL block (LayerNorm -> mLSTM layer -> residual) on flattened 3D patch tokens, fp32, gfx950.
//
// Reference semantics: UxLSTMEnc_3d.py:42-87 (outer ViLLayer), vision_lstm.py:48-130 (stabilised parallel
// mLSTM), :133-221 (headwise linear, causal conv), :224-287 (norms), :290-348 (cell), :351-477 (layer),
// :480-506 (block).
//
// mLSTM formulation used here.  With lf_t = logsigmoid(f_t), F_t = cumsum(lf)_t, g_s = i_s - F_s and
// G_t = max_{s<=t} g_s (a prefix maximum), the reference's row stabiliser is m_t = F_t + G_t and
//     P_ts = (q_t . k_s / sqrt(DH)) * exp(g_s - G_t)   (s <= t),   b_t = sum_s P_ts,   a_t = sum_s P_ts v_s,
//     h_t  = a_t / (max(|b_t|, exp(-m_t)) + 1e-6).
// Every exponent is <= 0 and known before the contraction starts, so the S x S matrix is never materialised and no
// online rescaling is needed: F and G come from two wavefront-shuffle scans (one wave per (batch, head)), the
// contraction is tiled over 64-key LDS tiles.  The backward uses the row/column-sum identities
//     dF_t = q_t.dq_t - k_t.dk_t (+ stabiliser terms),  di_s = k_s.dk_s (+ stabiliser terms)
// so it needs only dq, dk, dv from two tiled passes plus a reverse scan.
#include "common.h"
#include "../../include/xlstm_hved.h"

#define VIL_EPS 1e-5f
#define MLSTM_EPS 1e-6f
#define NH 4

// workspace layout (floats), all token-major unless noted
struct VilWs {
  float *tok, *ln_mean, *ln_rstd, *xm, *z, *xc, *q, *k, *v, *ig, *fg, *F, *G, *h, *bden;
  int* arg;
  // backward scratch
  float *dh, *dz, *dxa, *dap, *dbp, *dm, *dq, *dk, *dv, *di, *df, *dxc, *dxm, *rq, *ck, *dscat;
  // chunk-recurrent mLSTM: per (batch, head, chunk of 64 tokens) the carried matrix memory C (16x16) + normaliser n (16)
  // in front of the chunk, and the mirror-image state R, r of the backward (contributions of all LATER chunks)
  float *cst, *rst;
};
constexpr int CL = 64;                  // chunk length = one wavefront of tokens
constexpr int STF = 16 * 16 + 16;       // floats per chunk state
static long long ws_layout(VilWs* w, float* base, int B, int S, int C) {
  const long long I = 2 * C, BS = (long long)B * S;
  long long o = 0;
  auto take = [&](long long n) { float* p = base ? base + o : nullptr; o += (n + 3) / 4 * 4; return p; };
  VilWs t;
  t.tok = take(BS * C); t.ln_mean = take(BS); t.ln_rstd = take(BS);
  t.xm = take(BS * I); t.z = take(BS * I); t.xc = take(BS * I);
  t.q = take(BS * I); t.k = take(BS * I); t.v = take(BS * I);
  t.ig = take(BS * NH); t.fg = take(BS * NH); t.F = take(BS * NH); t.G = take(BS * NH);
  t.h = take(BS * I); t.bden = take(BS * NH);
  t.arg = (int*)take(BS * NH);
  t.dh = take(BS * I); t.dz = take(BS * I); t.dxa = take(BS * I);
  t.dap = take(BS * I); t.dbp = take(BS * NH); t.dm = take(BS * NH);
  t.dq = take(BS * I); t.dk = take(BS * I); t.dv = take(BS * I);
  t.di = take(BS * NH); t.df = take(BS * NH);
  t.dxc = take(BS * I); t.dxm = take(BS * I);
  t.rq = take(BS * NH); t.ck = take(BS * NH); t.dscat = take(BS * NH);
  const long long nst = (long long)B * NH * ((S + CL - 1) / CL) * STF;
  t.cst = take(nst); t.rst = take(nst);
  if (w) *w = t;
  return o;
}
extern "C" long long xh_vil_workspace_floats(int B, int S, int C) { return ws_layout(nullptr, nullptr, B, S, C); }

__device__ __forceinline__ float silu_(float x) { return x / (1.f + expf(-x)); }
__device__ __forceinline__ float dsilu_(float x) {
  const float s = 1.f / (1.f + expf(-x));
  return s * (1.f + x * (1.f - s));
}
__device__ __forceinline__ float logsigmoid_(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

template <typename T>
__device__ __forceinline__ float ld_in(const T* xa, const T* xb, long long i) {
  float v = ldf(xa, i);
  if (xb) v += ldf(xb, i);
  return v;
}

// accumulate dW[m][n] += sum_t a[t][m]*b[t][n] from LDS rows (row strides lda/ldb) with fp32 atomics
__device__ __forceinline__ void outer_accum(float* dW, const float* a, int lda, int M, const float* b, int ldb, int Nn,
                                            int T) {
  for (int e = threadIdx.x; e < M * Nn; e += blockDim.x) {
    const int m = e / Nn, n = e % Nn;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s = fmaf(a[t * lda + m], b[t * ldb + n], s);
    atomicAdd(&dW[e], s);
  }
}

// -------------------------------------------------------------------------------------------------
// forward stage 1: gather tokens, LayerNorm (weight 1+w, no bias), proj_up
// block: PRE1_TT tokens x PRE1_LP lanes (round 6: 16 x 16 -- 256 workgroups for 4 096 tokens and 8 outputs per lane; it was 32 x 8:
// one wave per SIMD on half the CUs walking 16 outputs x C products each)
// -------------------------------------------------------------------------------------------------
constexpr int PRE1_TT = 16, PRE1_LP = 16;
template <typename T, int C>
__global__ __launch_bounds__(PRE1_TT * PRE1_LP) void vil_pre1_kernel(const T* xa, const T* xb, int S, xh_vil_params p, VilWs w) {
  constexpr int I = 2 * C, TT = PRE1_TT, LP = PRE1_LP, NT = TT * LP, O = 4 * C;
  __shared__ float s_w[O * (C + 1)];
  __shared__ float s_t[TT * (C + 1)];
  const int tid = threadIdx.x, b = blockIdx.y, s0 = blockIdx.x * TT;
  for (int i = tid; i < O * C; i += NT) s_w[(i / C) * (C + 1) + (i % C)] = p.proj_up[i];
  for (int i = tid; i < TT * C; i += NT) {
    const int tk = i % TT, c = i / TT;
    const int s = s0 + tk;
    s_t[tk * (C + 1) + c] = s < S ? ld_in(xa, xb, ((long long)b * C + c) * S + s) : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < TT * C; i += NT) {
    const int tk = i / C, c = i % C;
    if (s0 + tk < S) w.tok[((long long)b * S + s0 + tk) * C + c] = s_t[tk * (C + 1) + c];
  }
  const int tk = tid / LP, sub = tid % LP;
  const int s = s0 + tk;
  float sum = 0.f, sq = 0.f;
  for (int c = sub; c < C; c += LP) { const float v = s_t[tk * (C + 1) + c]; sum += v; }
#pragma unroll
  for (int o = 1; o < LP; o <<= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / C;
  for (int c = sub; c < C; c += LP) { const float v = s_t[tk * (C + 1) + c] - mean; sq = fmaf(v, v, sq); }
#pragma unroll
  for (int o = 1; o < LP; o <<= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = rsqrtf(sq / C + VIL_EPS);
  if (sub == 0 && s < S) { w.ln_mean[(long long)b * S + s] = mean; w.ln_rstd[(long long)b * S + s] = rstd; }
  __syncthreads();
  for (int c = sub; c < C; c += LP) s_t[tk * (C + 1) + c] = (s_t[tk * (C + 1) + c] - mean) * rstd * (1.f + p.norm_w[c]);
  __syncthreads();
  if (s < S) {
    for (int j = 0; j < O / LP; ++j) {
      const int o = j * LP + sub;                       // a token's lanes take consecutive outputs: weight rows one bank apart, stores coalesced
      float a = 0.f;
#pragma unroll 8
      for (int k = 0; k < C; ++k) a = fmaf(s_w[o * (C + 1) + k], s_t[tk * (C + 1) + k], a);
      if (o < I) w.xm[((long long)b * S + s) * I + o] = a;
      else w.z[((long long)b * S + s) * I + (o - I)] = a;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// forward stage 2: causal conv1d (k=4) + SiLU, block-diagonal q/k/v, gate pre-activations
// block: PRE2_TT tokens x 16 lanes at C = 32 (lane = head x 4-channel block of the head; 8 lanes at C = 16).  Round 6: it was 32 tokens x NH lanes, a lane walking its
// head's 16 channels -- 64 conv taps, 192 projection products and two 192-long gate dot products in a row, their weights fetched from
// global memory inside the loops, on ONE wave per SIMD (128 workgroups of 128 threads for 4 096 tokens): 22 us of latency.  Four times
// the lanes, a quarter of the chain each; the gate dot products are split over the head's four lanes and meet by two shuffles.
// -------------------------------------------------------------------------------------------------
constexpr int PRE2_TT = 16;     // tokens per workgroup: 256 workgroups for the 4 096 tokens of the 128^3 patch's deepest level (one per CU)
template <int C>
__global__ __launch_bounds__(PRE2_TT * NH * (2 * C / NH / 4)) void vil_pre2_kernel(int S, xh_vil_params p, VilWs w) {
  constexpr int I = 2 * C, DH = I / NH, TT = PRE2_TT, NB4 = DH / 4;
  static_assert(NB4 == 4 || NB4 == 2, "lanes per token = NH x (DH / 4): 16 (C = 32) or 8 (C = 16)");
  constexpr int LPT = NH * NB4;
  __shared__ float s_xm[(TT + 3) * (I + 1)];
  __shared__ float s_qkv[TT * (3 * I + 1)];
  __shared__ float s_gw[2 * NH * 3 * I];
  __shared__ __attribute__((aligned(16))) float s_pw[4 * I * 4];     // conv_w | q_w | k_w | v_w, each [I][4]: a lane reads rows of four
  const int tid = threadIdx.x, b = blockIdx.y, s0 = blockIdx.x * TT;
  for (int i = tid; i < (TT + 3) * I; i += blockDim.x) {
    const int r = i / I, c = i % I;
    const int s = s0 - 3 + r;
    s_xm[r * (I + 1) + c] = (s >= 0 && s < S) ? w.xm[((long long)b * S + s) * I + c] : 0.f;
  }
  for (int i = tid; i < NH * 3 * I; i += blockDim.x) { s_gw[i] = p.ig_w[i]; s_gw[NH * 3 * I + i] = p.fg_w[i]; }
  for (int i = tid; i < I * 4; i += blockDim.x) {
    s_pw[i] = p.conv_w[i]; s_pw[I * 4 + i] = p.q_w[i]; s_pw[2 * I * 4 + i] = p.k_w[i]; s_pw[3 * I * 4 + i] = p.v_w[i];
  }
  __syncthreads();
  const int tk = tid / LPT, h = (tid / NB4) % NH, blk = tid % NB4;
  const int s = s0 + tk;
  const int gb = h * NB4 + blk;          // global 4x4 block index
  const int c0 = h * DH + blk * 4;       // first of the lane's four channels
  float xa[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int c = c0 + d;
    const float4 cw = *reinterpret_cast<const float4*>(s_pw + c * 4);
    float a = p.conv_b[c];
    a = fmaf(cw.x, s_xm[(tk + 0) * (I + 1) + c], a);
    a = fmaf(cw.y, s_xm[(tk + 1) * (I + 1) + c], a);
    a = fmaf(cw.z, s_xm[(tk + 2) * (I + 1) + c], a);
    a = fmaf(cw.w, s_xm[(tk + 3) * (I + 1) + c], a);
    xa[d] = silu_(a);
    if (s < S) w.xc[((long long)b * S + s) * I + c] = a;
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const float4 qw = *reinterpret_cast<const float4*>(s_pw + I * 4 + (gb * 4 + o) * 4);
    const float4 kw = *reinterpret_cast<const float4*>(s_pw + 2 * I * 4 + (gb * 4 + o) * 4);
    const float4 vw = *reinterpret_cast<const float4*>(s_pw + 3 * I * 4 + (gb * 4 + o) * 4);
    const float* xm3 = s_xm + (tk + 3) * (I + 1) + c0;
    const float q = fmaf(qw.w, xa[3], fmaf(qw.z, xa[2], fmaf(qw.y, xa[1], qw.x * xa[0])));
    const float k = fmaf(kw.w, xa[3], fmaf(kw.z, xa[2], fmaf(kw.y, xa[1], kw.x * xa[0])));
    const float v = fmaf(vw.w, xm3[3], fmaf(vw.z, xm3[2], fmaf(vw.y, xm3[1], vw.x * xm3[0])));
    const int j = blk * 4 + o;
    s_qkv[tk * (3 * I + 1) + h * DH + j] = q;
    s_qkv[tk * (3 * I + 1) + I + h * DH + j] = k;
    s_qkv[tk * (3 * I + 1) + 2 * I + h * DH + j] = v;
    if (s < S) {
      const long long o_ = (((long long)b * NH + h) * S + s) * DH + j;
      w.q[o_] = q; w.k[o_] = k; w.v[o_] = v;
    }
  }
  __syncthreads();
  float gi = 0.f, gf = 0.f;
  for (int m = blk; m < 3 * I; m += NB4) {               // the head's lanes take every NB4-th element of the token's [q, k, v]
    const float x = s_qkv[tk * (3 * I + 1) + m];
    gi = fmaf(s_gw[h * 3 * I + m], x, gi);
    gf = fmaf(s_gw[NH * 3 * I + h * 3 * I + m], x, gf);
  }
  gi += __shfl_xor(gi, 1, 64); gf += __shfl_xor(gf, 1, 64);
  if (NB4 == 4) { gi += __shfl_xor(gi, 2, 64); gf += __shfl_xor(gf, 2, 64); }
  if (s < S && blk == 0) {
    w.ig[((long long)b * NH + h) * S + s] = gi + p.ig_b[h];
    w.fg[((long long)b * NH + h) * S + s] = gf + p.fg_b[h];
  }
}

// -------------------------------------------------------------------------------------------------
// gate scans: one 1024-lane workgroup per (batch, head).  F = inclusive cumsum of logsigmoid(f) (fp64 carry),
// G = inclusive prefix max of g = i - F with its arg index.  A lane owns a short contiguous chunk (4 tokens at
// S = 4096, so a wave reads 1 KB runs); carries move through wavefront shuffles, then across the 16 waves through LDS.
// -------------------------------------------------------------------------------------------------
constexpr int SCAN_T = 1024, SCAN_W = SCAN_T / 64;
__global__ __launch_bounds__(SCAN_T) void vil_scan_kernel(int S, VilWs w) {
  __shared__ double s_sum[SCAN_W];
  __shared__ float s_max[SCAN_W];
  __shared__ int s_arg[SCAN_W];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long base = (long long)blockIdx.x * S;
  const int chunk = (S + SCAN_T - 1) / SCAN_T;
  const int t0 = min(S, tid * chunk), t1 = min(S, t0 + chunk);
  // A lane's chunk is short (4 tokens at S = 4096): it is loaded ONCE into registers -- the three walks over it (sum, prefix + local
  // maximum, running maximum) were three round trips to memory, the last one re-reading the F this kernel had just written
  constexpr int CH = 8;
  const bool reg = chunk <= CH;                   // block-uniform
  float lfv[CH], gv[CH];
  double loc = 0.0;
  if (reg) {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int t = t0 + k;
      const bool in = t < t1;
      lfv[k] = in ? logsigmoid_(w.fg[base + (in ? t : 0)]) : 0.f;
      gv[k] = in ? w.ig[base + (in ? t : 0)] : 0.f;
      loc += (double)lfv[k];
    }
  } else {
    for (int t = t0; t < t1; ++t) loc += (double)logsigmoid_(w.fg[base + t]);
  }
  double inc = loc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) s_sum[wv] = inc;
  __syncthreads();
  double run = inc - loc;   // exclusive prefix within the wave ...
  for (int i = 0; i < wv; ++i) run += s_sum[i];   // ... plus the waves before it
  float lmax = -INFINITY;
  int larg = t0;
  if (reg) {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int t = t0 + k;
      if (t < t1) {
        run += (double)lfv[k];
        const float Ft = (float)run;
        w.F[base + t] = Ft;
        gv[k] -= Ft;                              // g_t = i_t - F_t
        if (gv[k] > lmax) { lmax = gv[k]; larg = t; }
      }
    }
  } else {
    for (int t = t0; t < t1; ++t) {
      run += (double)logsigmoid_(w.fg[base + t]);
      const float Ft = (float)run;
      w.F[base + t] = Ft;
      const float g = w.ig[base + t] - Ft;
      if (g > lmax) { lmax = g; larg = t; }
    }
  }
  float pm = lmax;
  int pa = larg;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float um = __shfl_up(pm, o, 64);
    const int ua = __shfl_up(pa, o, 64);
    if (lane >= o && um >= pm) { pm = um; pa = ua; }   // earlier index wins ties
  }
  if (lane == 63) { s_max[wv] = pm; s_arg[wv] = pa; }
  __syncthreads();
  float em = -INFINITY;
  int ea = 0;
  for (int i = 0; i < wv; ++i)
    if (s_max[i] > em) { em = s_max[i]; ea = s_arg[i]; }
  const float xm = __shfl_up(pm, 1, 64);
  const int xa = __shfl_up(pa, 1, 64);
  if (lane > 0 && xm > em) { em = xm; ea = xa; }
  if (reg) {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int t = t0 + k;
      if (t < t1) {
        if (gv[k] > em) { em = gv[k]; ea = t; }
        w.G[base + t] = em;
        w.arg[base + t] = ea;
      }
    }
    return;
  }
  for (int t = t0; t < t1; ++t) {
    const float g = w.ig[base + t] - w.F[base + t];
    if (g > em) { em = g; ea = t; }
    w.G[base + t] = em;
    w.arg[base + t] = ea;
  }
}

// reverse scan for the backward: df_t = sigmoid(-f_t) * sum_{t'>=t} dF_t'
__global__ __launch_bounds__(SCAN_T) void vil_rscan_kernel(int S, VilWs w) {
  __shared__ double s_sum[SCAN_W];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long base = (long long)blockIdx.x * S;
  const int chunk = (S + SCAN_T - 1) / SCAN_T;
  const int t0 = min(S, tid * chunk), t1 = min(S, t0 + chunk);
  // dF_t = rq_t - ck_t + dm_t - dscat_t ;  di_t = ck_t + dscat_t
  constexpr int CH = 8;                           // short chunks live in registers (see vil_scan_kernel)
  const bool reg = chunk <= CH;
  float dFv[CH], div[CH], sgv[CH];
  double loc = 0.0;
  if (reg) {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int t = t0 + k;
      const bool in = t < t1;
      const long long o = base + (in ? t : 0);
      const float ck = w.ck[o], ds = w.dscat[o];
      dFv[k] = in ? w.rq[o] - ck + w.dm[o] - ds : 0.f;
      div[k] = ck + ds;
      sgv[k] = 1.f / (1.f + expf(w.fg[o]));
      loc += (double)dFv[k];
    }
  } else {
    for (int t = t0; t < t1; ++t) loc += (double)(w.rq[base + t] - w.ck[base + t] + w.dm[base + t] - w.dscat[base + t]);
  }
  double inc = loc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double dn = __shfl_down(inc, o, 64);
    if (lane + o < 64) inc += dn;
  }
  if (lane == 0) s_sum[wv] = inc;
  __syncthreads();
  double run = inc - loc;   // sum over the later lanes of the wave ...
  for (int i = wv + 1; i < SCAN_W; ++i) run += s_sum[i];   // ... and the later waves
  if (reg) {
#pragma unroll
    for (int k = CH - 1; k >= 0; --k) {
      const int t = t0 + k;
      if (t < t1) {
        run += (double)dFv[k];
        w.df[base + t] = (float)run * sgv[k];
        w.di[base + t] = div[k];
      }
    }
    return;
  }
  for (int t = t1 - 1; t >= t0; --t) {
    run += (double)(w.rq[base + t] - w.ck[base + t] + w.dm[base + t] - w.dscat[base + t]);
    const float f = w.fg[base + t];
    w.df[base + t] = (float)run * (1.f / (1.f + expf(f)));
    w.di[base + t] = w.ck[base + t] + w.dscat[base + t];
  }
}

// -------------------------------------------------------------------------------------------------
// mLSTM forward contraction.  block: 32 queries x 8 key slices; key tiles of 64 staged in LDS.
// All exponents are known before the contraction (no online rescaling), so partial sums over disjoint key sets simply
// add: the key tiles of a query tile are dealt round-robin to MSPLIT workgroups (balanced under the causal limit, 4x
// shorter serial chain, 4x more workgroups) that accumulate numerator and denominator with float atomics;
// mlstm_norm_kernel then divides.
// -------------------------------------------------------------------------------------------------
constexpr int MSPLIT = 4;
template <int DH>
__global__ __launch_bounds__(256) void mlstm_fwd_kernel(int S, VilWs w) {
  constexpr int KT = 64, LD = DH + 4;
  __shared__ float s_k[KT * LD], s_v[KT * LD], s_g[KT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const long long hb = ((long long)b * NH + h) * S;
  const int t0 = (blockIdx.x / MSPLIT) * 32, ks = blockIdx.x % MSPLIT;
  const int t = t0 + wv * 8 + (lane >> 3), sl = lane & 7;
  const bool tv = t < S;
  float q[DH], num[DH];
  float den = 0.f, Gt = 0.f;
#pragma unroll
  for (int j = 0; j < DH; ++j) { q[j] = tv ? w.q[(hb + t) * DH + j] : 0.f; num[j] = 0.f; }
  if (tv) Gt = w.G[hb + t];
  const float isq = rsqrtf((float)DH);
  const int s_last = min(S - 1, t0 + 31);
  for (int st0 = ks * KT; st0 <= s_last; st0 += MSPLIT * KT) {
    __syncthreads();
    for (int i = tid; i < KT * DH; i += 256) {
      const int r = i / DH, c = i % DH;
      const int s = st0 + r;
      s_k[r * LD + c] = s < S ? w.k[(hb + s) * DH + c] : 0.f;
      s_v[r * LD + c] = s < S ? w.v[(hb + s) * DH + c] : 0.f;
    }
    if (tid < KT) { const int s = st0 + tid; s_g[tid] = s < S ? w.ig[hb + s] - w.F[hb + s] : -INFINITY; }
    __syncthreads();
#pragma unroll 2
    for (int jj = 0; jj < KT / 8; ++jj) {
      const int r = sl + 8 * jj;
      const int s = st0 + r;
      if (tv && s <= t) {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < DH; ++j) dot = fmaf(q[j], s_k[r * LD + j], dot);
        const float pw = dot * isq * expf(s_g[r] - Gt);
        den += pw;
#pragma unroll
        for (int j = 0; j < DH; ++j) num[j] = fmaf(pw, s_v[r * LD + j], num[j]);
      }
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    den += __shfl_xor(den, o, 64);
#pragma unroll
    for (int j = 0; j < DH; ++j) num[j] += __shfl_xor(num[j], o, 64);
  }
  if (tv && sl == 0) {                                 // partial sums of this key subset (h / bden are zeroed before the launch)
    atomicAdd(&w.bden[hb + t], den);
#pragma unroll
    for (int j = 0; j < DH; ++j) atomicAdd(&w.h[(hb + t) * DH + j], num[j]);
  }
}
// h_t = num_t / (max(|den_t|, exp(-m_t)) + eps) once all key subsets have been added
template <int DH>
__global__ __launch_bounds__(256) void mlstm_norm_kernel(long long total, VilWs w) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over B*NH*S rows
  if (i >= total) return;
  const float m = w.F[i] + w.G[i];
  const float inv = 1.f / (fmaxf(fabsf(w.bden[i]), expf(-m)) + MLSTM_EPS);
#pragma unroll
  for (int j = 0; j < DH; ++j) w.h[i * DH + j] *= inv;
}

// -------------------------------------------------------------------------------------------------
// forward stage 3: per-head norm, learnable skip, output gate, proj_down, residuals, scatter to NCDHW
// block: POST_TT tokens x POST_LP lanes (round 6: 16 x 16, a lane = (head, quarter of the head's channels) in the norm and C / 16
// outputs of proj_down; it was 32 x 8 with four of a token's eight lanes idle through the norm)
// -------------------------------------------------------------------------------------------------
constexpr int POST_TT = 16, POST_LP = 16;
template <typename T, int C>
__global__ __launch_bounds__(POST_TT * POST_LP) void vil_post_kernel(const T* xa, T* out, int S, int add_xa, xh_vil_params p, VilWs w) {
  constexpr int I = 2 * C, DH = I / NH, TT = POST_TT, LP = POST_LP, NT = TT * LP;
  constexpr int LH = LP / NH, JH = DH / LH;            // lanes per head, channels per lane in the norm
  static_assert(LP % NH == 0 && DH % LH == 0 && C % LP == 0, "lane map of vil_post_kernel");
  __shared__ float s_w[C * (I + 1)];
  __shared__ float s_hg[TT * (I + 1)];
  __shared__ float s_o[TT * (C + 1)];
  const int tid = threadIdx.x, b = blockIdx.y, s0 = blockIdx.x * TT;
  for (int i = tid; i < C * I; i += NT) s_w[(i / I) * (I + 1) + (i % I)] = p.proj_down[i];
  const int tk = tid / LP, sub = tid % LP;
  const int s = s0 + tk;
  {
    const int h = sub / LH, part = sub % LH;
    const bool live = s < S;
    const long long ho = (((long long)b * NH + h) * (live ? S : 0) + (live ? s : 0)) * DH + part * JH;
    float hv[JH], mean = 0.f, var = 0.f;
#pragma unroll
    for (int j = 0; j < JH; ++j) { hv[j] = live ? w.h[ho + j] : 0.f; mean += hv[j]; }
#pragma unroll
    for (int o = 1; o < LH; o <<= 1) mean += __shfl_xor(mean, o, 64);
    mean /= DH;
#pragma unroll
    for (int j = 0; j < JH; ++j) { const float d = hv[j] - mean; var = fmaf(d, d, var); }
#pragma unroll
    for (int o = 1; o < LH; o <<= 1) var += __shfl_xor(var, o, 64);
    const float rstd = rsqrtf(var / DH + VIL_EPS);
    if (live) {
#pragma unroll
      for (int j = 0; j < JH; ++j) {
        const int c = h * DH + part * JH + j;
        const long long to = ((long long)b * S + s) * I + c;
        const float hn = (hv[j] - mean) * rstd * (1.f + p.outnorm_w[c]);
        const float hs = hn + p.skip[c] * silu_(w.xc[to]);
        s_hg[tk * (I + 1) + c] = hs * silu_(w.z[to]);
      }
    }
  }
  __syncthreads();
  if (s < S) {
    for (int j = 0; j < C / LP; ++j) {
      const int o = j * LP + sub;
      float a = w.tok[((long long)b * S + s) * C + o];
#pragma unroll 8
      for (int k = 0; k < I; ++k) a = fmaf(s_w[o * (I + 1) + k], s_hg[tk * (I + 1) + k], a);
      s_o[tk * (C + 1) + o] = a;
    }
  }
  __syncthreads();
  for (int i = tid; i < TT * C; i += NT) {
    const int tk2 = i % TT, c = i / TT;
    const int s2 = s0 + tk2;
    if (s2 < S) {
      const long long o = ((long long)b * C + c) * S + s2;
      stf(out, o, (add_xa ? ldf(xa, o) : 0.f) + s_o[tk2 * (C + 1) + c]);
    }
  }
}

// =================================================================================================
// Chunk-recurrent mLSTM on the matrix cores (head dimension 16).
//
// Tokens are cut into chunks of CL = 64 (one wavefront).  With w_s = exp(g_s - M) for a reference M >= every g_s involved
// (all exponents <= 0, like the tiled form above), everything a query t of chunk c needs from EARLIER chunks is the
// 16x16 matrix memory  C_c = sum_{s < cL} w_s k_s v_s^T  and the normaliser  n_c = sum w_s k_s,  referenced to
// M_c = G_{cL-1}:
//     a_t = isq [ sum_{s in chunk, s<=t} (q_t.k_s) e^{g_s-G_t} v_s  +  e^{M_c-G_t} C_c^T q_t ],   b_t likewise with n_c
// (vision_lstm.py:99-128 evaluates the same sums as one S x S matrix; the recurrent form of oracle.mlstm_recurrent carries
// exactly this C, n).  Three launches: (1) every chunk's own contribution dC_c in parallel, (2) a 272-lane scan over the
// chunks of one (batch, head) that turns them into the carried states, (3) every chunk's intra-chunk 64x64 block + the
// carried state in parallel.  All contractions -- Q K^T, P V, K^T (w V), Q C -- are v_mfma_f32_16x16x4_f32: exact fp32
// like the reference's fp32-forced ViL (UxLSTMEnc_3d.py:77-80), at the vector rate but off the VALU.  O(S * 64 * DH)
// instead of O(S^2 * DH): S = 4096 costs what S = 64 x 64 costs, and S = 32 768 (DoubleConv_ViL at 128^3) is 8x that.
// The backward mirrors it with the reverse state R_c = sum_{t >= (c+1)L} e^{Mr_c - G_t} q_t da'_t^T (+ r_c with db'_t),
// Mr_c = G_{(c+1)L}, and recomputes the 64x64 blocks per chunk.
// =================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// MFMA 16x16x4 f32 lane roles: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]; D[i = 4*(lane>>4) + r][j = lane&15]

// (1) local states.  REV = 0: dC_c[k][j] = sum_{s in c} e^{g_s - G_end(c)} k_s[k] v_s[j], dn_c[k] likewise (forward).
//                    REV = 1: dR_c[a][b] = sum_{t in c} e^{G_first(c) - G_t} q_t[a] da'_t[b], dr_c[a] = sum ... db'_t q_t[a].
template <int REV>
__global__ __launch_bounds__(64) void mlstm_chunk_dstate_kernel(int S, int nchunk, VilWs w) {
  const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const long long hb = ((long long)b * NH + h) * S;
  const int s_base = c * CL, s_end = min(S, s_base + CL);
  const float ref = REV ? w.G[hb + s_base] : w.G[hb + s_end - 1];
  const int kj = lane & 15, sg = lane >> 4;
  const float* ap = REV ? w.q : w.k;
  const float* bp = REV ? w.dap : w.v;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float nacc = 0.f;
#pragma unroll 4
  for (int s0 = 0; s0 < CL; s0 += 4) {
    const int s = s_base + s0 + sg;
    float a = 0.f, bv = 0.f, nw = 1.f;
    if (s < S) {
      const float wt = REV ? expf(ref - w.G[hb + s]) : expf(w.ig[hb + s] - w.F[hb + s] - ref);
      a = wt * ap[(hb + s) * 16 + kj];
      bv = bp[(hb + s) * 16 + kj];
      if (REV) nw = w.dbp[hb + s];
    }
    acc = MFMA4(a, bv, acc);
    nacc = fmaf(a, nw, nacc);
  }
  float* st = (REV ? w.rst : w.cst) + ((((long long)b * NH + h) * nchunk) + c) * STF;
#pragma unroll
  for (int r = 0; r < 4; ++r) st[(sg * 4 + r) * 16 + kj] = acc[r];
  nacc += __shfl_xor(nacc, 16, 64);
  nacc += __shfl_xor(nacc, 32, 64);
  if (lane < 16) st[256 + lane] = nacc;
}

// (2) exclusive scan of the local states over the chunks of one (batch, head), in place; one lane per state element.
//     forward: st[c] <- state carried INTO chunk c (reference M_c = G_{cL-1});  reverse: st[c] <- state of all chunks AFTER
//     c (reference Mr_c = G_{(c+1)L}).
//     The recurrence is one fma per chunk, but written as "load, fma, store" per chunk it was a chain of 64 dependent global
//     round trips (22 us at S = 4096 on four workgroups): the chunk values of a block of SCB chunks are now requested together
//     (independent loads into registers), the decays of the block are computed once per workgroup into LDS, and only the fma
//     chain itself is serial.
constexpr int SCB = 64;
template <int REV>
__global__ __launch_bounds__(320) void mlstm_chunk_scan_kernel(int S, int nchunk, VilWs w) {
  __shared__ float s_dec[SCB];
  const int e = threadIdx.x;
  const bool live = e < STF;
  const long long hb = (long long)blockIdx.x * S;
  float* st = (REV ? w.rst : w.cst) + (long long)blockIdx.x * nchunk * STF;
  float run = 0.f;
  for (int b0 = 0; b0 < nchunk; b0 += SCB) {
    // chunk index of position i of this block, in scan order (forward: ascending, reverse: descending)
    auto cidx = [&](int i) { return REV ? nchunk - 1 - (b0 + i) : b0 + i; };
    __syncthreads();                                    // the previous block's decays are no longer read
    if (e < SCB && b0 + e < nchunk) {
      const int c = cidx(e);
      float dec;
      if (!REV) dec = c > 0 ? expf(w.G[hb + c * CL - 1] - w.G[hb + min(S, (c + 1) * CL) - 1]) : 0.f;
      else dec = c < nchunk - 1 ? expf(w.G[hb + c * CL] - w.G[hb + (c + 1) * CL]) : 0.f;
      s_dec[e] = dec;
    }
    float d[SCB];
#pragma unroll
    for (int i = 0; i < SCB; ++i) d[i] = (live && b0 + i < nchunk) ? st[(long long)cidx(i) * STF + e] : 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SCB; ++i) {
      if (b0 + i < nchunk) {
        const float out = run;
        run = fmaf(run, s_dec[i], d[i]);
        d[i] = out;
      }
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < SCB; ++i)
        if (b0 + i < nchunk) st[(long long)cidx(i) * STF + e] = d[i];
    }
  }
}

constexpr int CLD = 17;      // LDS row stride of the 64 x 16 token tiles (conflict-free A/B operand reads)
constexpr int CPD = 66;      // LDS row stride of the 64 x 64 score tiles

__device__ __forceinline__ void chunk_load_rows(float* dst, const float* src, long long row0, int rows_valid, int tid) {
  // 64 rows x 16 floats, 256 threads: one float4 each
  const int r = tid >> 2, c4 = (tid & 3) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < rows_valid) v = *reinterpret_cast<const float4*>(src + (row0 + r) * 16 + c4);
  dst[r * CLD + c4] = v.x; dst[r * CLD + c4 + 1] = v.y; dst[r * CLD + c4 + 2] = v.z; dst[r * CLD + c4 + 3] = v.w;
}

// (3) forward: one workgroup (4 waves) per chunk; wave wv owns query rows 16*wv .. +15.
__global__ __launch_bounds__(256) void mlstm_chunk_fwd_kernel(int S, int nchunk, VilWs w) {
  __shared__ float s_q[CL * CLD], s_k[CL * CLD], s_v[CL * CLD], s_P[CL * CPD];
  __shared__ float s_g[CL], s_G[CL], s_F[CL], s_C[STF];
  const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int kj = lane & 15, sg = lane >> 4;
  const long long hb = ((long long)b * NH + h) * S;
  const int s_base = c * CL, nv = min(CL, S - s_base);
  chunk_load_rows(s_q, w.q, hb + s_base, nv, tid);
  chunk_load_rows(s_k, w.k, hb + s_base, nv, tid);
  chunk_load_rows(s_v, w.v, hb + s_base, nv, tid);
  if (tid < CL) {
    const bool ok = tid < nv;
    const float Fv = ok ? w.F[hb + s_base + tid] : 0.f;
    s_F[tid] = Fv;
    s_g[tid] = ok ? w.ig[hb + s_base + tid] - Fv : -INFINITY;
    s_G[tid] = ok ? w.G[hb + s_base + tid] : 0.f;
  }
  const float* cst = w.cst + ((((long long)b * NH + h) * nchunk) + c) * STF;
  for (int i = tid; i < STF; i += 256) s_C[i] = cst[i];
  __syncthreads();
  const float isq = 0.25f;                        // 1 / sqrt(16)
  float Gt[4], den[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) Gt[r] = s_G[16 * wv + sg * 4 + r];
  for (int kb = 0; kb <= wv; ++kb) {              // causal: key blocks up to the wave's own
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4)
      acc = MFMA4(s_q[(16 * wv + kj) * CLD + k0 + sg], s_k[(16 * kb + kj) * CLD + k0 + sg], acc);
    const int sl = 16 * kb + kj;
    const float gs = s_g[sl];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tl = 16 * wv + sg * 4 + r;
      const float pv = (sl <= tl && tl < nv) ? isq * acc[r] * expf(gs - Gt[r]) : 0.f;
      den[r] += pv;
      s_P[tl * CPD + sl] = pv;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) den[r] += __shfl_xor(den[r], o, 64);
  f32x4 accA = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < 16 * (wv + 1); s0 += 4)    // P rows of this wave x V
    accA = MFMA4(s_P[(16 * wv + kj) * CPD + s0 + sg], s_v[(s0 + sg) * CLD + kj], accA);
  f32x4 accI = {0.f, 0.f, 0.f, 0.f}, accN = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k0 = 0; k0 < 16; k0 += 4) {
    const float a = s_q[(16 * wv + kj) * CLD + k0 + sg];
    accI = MFMA4(a, s_C[(k0 + sg) * 16 + kj], accI);          // (C^T q_t)[j]
    accN = MFMA4(a, s_C[256 + k0 + sg], accN);                // q_t . n   (every column)
  }
  const float Mc = c > 0 ? w.G[hb + s_base - 1] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int tl = 16 * wv + sg * 4 + r;
    if (tl >= nv) continue;
    const float sc = c > 0 ? isq * expf(Mc - Gt[r]) : 0.f;
    const float num = fmaf(sc, accI[r], accA[r]);
    const float bd = fmaf(sc, accN[r], den[r]);
    const float m = s_F[tl] + Gt[r];
    const float nrm = fmaxf(fabsf(bd), expf(-m)) + MLSTM_EPS;
    w.h[(hb + s_base + tl) * 16 + kj] = num / nrm;
    if (kj == 0) w.bden[hb + s_base + tl] = bd;
  }
}

// (3') backward: dq, dk, dv of one chunk.  Phase (i): wave wv computes the masked, weighted 16-row strips
// EW[t][s] = (da'_t.v_s + db'_t) isq e^{g_s-G_t} and SW[t][s] = (q_t.k_s) isq e^{g_s-G_t} of its query rows and dq of
// those rows; phase (ii) (after a barrier): the same wave, now owning KEY rows 16*wv.., reads the strips column-wise for
// dk_s = sum_t EW[t][s] q_t, dv_s = sum_t SW[t][s] da'_t, and adds the later chunks' contribution through R, r.
__global__ __launch_bounds__(256) void mlstm_chunk_bwd_kernel(int S, int nchunk, VilWs w) {
  __shared__ float s_q[CL * CLD], s_k[CL * CLD], s_v[CL * CLD], s_da[CL * CLD], s_EW[CL * CPD], s_SW[CL * CPD];
  __shared__ float s_g[CL], s_G[CL], s_db[CL], s_C[STF], s_R[STF];
  const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int kj = lane & 15, sg = lane >> 4;
  const long long hb = ((long long)b * NH + h) * S;
  const int s_base = c * CL, nv = min(CL, S - s_base);
  chunk_load_rows(s_q, w.q, hb + s_base, nv, tid);
  chunk_load_rows(s_k, w.k, hb + s_base, nv, tid);
  chunk_load_rows(s_v, w.v, hb + s_base, nv, tid);
  chunk_load_rows(s_da, w.dap, hb + s_base, nv, tid);
  if (tid < CL) {
    const bool ok = tid < nv;
    s_g[tid] = ok ? w.ig[hb + s_base + tid] - w.F[hb + s_base + tid] : -INFINITY;
    s_G[tid] = ok ? w.G[hb + s_base + tid] : 0.f;
    s_db[tid] = ok ? w.dbp[hb + s_base + tid] : 0.f;
  }
  const long long so = ((((long long)b * NH + h) * nchunk) + c) * STF;
  for (int i = tid; i < STF; i += 256) { s_C[i] = w.cst[so + i]; s_R[i] = w.rst[so + i]; }
  __syncthreads();
  const float isq = 0.25f;
  float Gt[4], dbt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { Gt[r] = s_G[16 * wv + sg * 4 + r]; dbt[r] = s_db[16 * wv + sg * 4 + r]; }
  // ---- phase (i): strips of the wave's query rows, dq ----
  for (int kb = 0; kb < 4; ++kb) {
    if (kb > wv) {                                   // above the diagonal: zeros (phase (ii) reads whole columns)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_EW[(16 * wv + sg * 4 + r) * CPD + 16 * kb + kj] = 0.f;
        s_SW[(16 * wv + sg * 4 + r) * CPD + 16 * kb + kj] = 0.f;
      }
      continue;
    }
    f32x4 aS = {0.f, 0.f, 0.f, 0.f}, aE = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) {
      aS = MFMA4(s_q[(16 * wv + kj) * CLD + k0 + sg], s_k[(16 * kb + kj) * CLD + k0 + sg], aS);
      aE = MFMA4(s_da[(16 * wv + kj) * CLD + k0 + sg], s_v[(16 * kb + kj) * CLD + k0 + sg], aE);
    }
    const int sl = 16 * kb + kj;
    const float gs = s_g[sl];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tl = 16 * wv + sg * 4 + r;
      const float wg = (sl <= tl && tl < nv) ? isq * expf(gs - Gt[r]) : 0.f;      // padded query rows: exactly zero
      s_EW[tl * CPD + sl] = (aE[r] + dbt[r]) * wg;
      s_SW[tl * CPD + sl] = aS[r] * wg;
    }
  }
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accX = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < 16 * (wv + 1); s0 += 4)
      acc = MFMA4(s_EW[(16 * wv + kj) * CPD + s0 + sg], s_k[(s0 + sg) * CLD + kj], acc);
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4)              // (C da'_t)[j] = sum_k C[j][k] da'_t[k]
      accX = MFMA4(s_da[(16 * wv + kj) * CLD + k0 + sg], s_C[kj * 16 + k0 + sg], accX);
    const float Mc = c > 0 ? w.G[hb + s_base - 1] : 0.f;
    const float nj = s_C[256 + kj];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tl = 16 * wv + sg * 4 + r;
      if (tl >= nv) continue;
      const float sc = c > 0 ? isq * expf(Mc - Gt[r]) : 0.f;
      w.dq[(hb + s_base + tl) * 16 + kj] = fmaf(sc, fmaf(dbt[r], nj, accX[r]), acc[r]);
    }
  }
  __syncthreads();
  // ---- phase (ii): the wave's key rows ----
  {
    f32x4 aK = {0.f, 0.f, 0.f, 0.f}, aV = {0.f, 0.f, 0.f, 0.f}, xK = {0.f, 0.f, 0.f, 0.f}, xV = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = 16 * wv; t0 < CL; t0 += 4) {
      aK = MFMA4(s_EW[(t0 + sg) * CPD + 16 * wv + kj], s_q[(t0 + sg) * CLD + kj], aK);
      aV = MFMA4(s_SW[(t0 + sg) * CPD + 16 * wv + kj], s_da[(t0 + sg) * CLD + kj], aV);
    }
    const bool later = c < nchunk - 1;
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) {
      xK = MFMA4(s_v[(16 * wv + kj) * CLD + k0 + sg], s_R[kj * 16 + k0 + sg], xK);     // (R v_s)[j]
      xV = MFMA4(s_k[(16 * wv + kj) * CLD + k0 + sg], s_R[(k0 + sg) * 16 + kj], xV);   // (R^T k_s)[j]
    }
    const float Mr = later ? w.G[hb + s_base + CL] : 0.f;
    const float rj = s_R[256 + kj];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int sl = 16 * wv + sg * 4 + r;
      if (sl >= nv) continue;
      const float sc = later ? isq * expf(s_g[sl] - Mr) : 0.f;
      w.dk[(hb + s_base + sl) * 16 + kj] = fmaf(sc, xK[r] + rj, aK[r]);
      w.dv[(hb + s_base + sl) * 16 + kj] = fmaf(sc, xV[r], aV[r]);
    }
  }
}

// =================================================================================================
// backward
// =================================================================================================
// stage 3 backward: dout (NCDHW) -> dh, dz, dxa(partial: skip path) ; grads of proj_down, skip, outnorm
// block: 32 tokens x 16 lanes (round 6; it was 32 x 8 with the per-head norm backward -- 16 channels, four passes over them -- on four of
// a token's eight lanes).  The token count per workgroup stays: every workgroup ends in one global atomic per parameter-gradient
// element, and atomics on one address retire one after the other.
template <typename T, int C>
__global__ __launch_bounds__(512) void vil_post_bwd_kernel(const T* dout, int S, xh_vil_params p, xh_vil_grads g, VilWs w) {
  constexpr int I = 2 * C, DH = I / NH, TT = 32, LP = 16, NT = TT * LP;
  constexpr int LH = LP / NH, JH = DH / LH;             // lanes per head, channels per lane in the norm backward
  static_assert(LP % NH == 0 && DH % LH == 0 && I % LP == 0, "lane map of vil_post_bwd_kernel");
  __shared__ float s_w[C * (I + 1)];
  __shared__ float s_do[TT * (C + 1)];
  __shared__ float s_hg[TT * (I + 1)];    // forward hg (for dW_down), later reused
  __shared__ float s_dhg[TT * (I + 1)];
  __shared__ float s_acc[2 * I];           // dskip, doutnorm partials
  const int tid = threadIdx.x, b = blockIdx.y, s0 = blockIdx.x * TT;
  for (int i = tid; i < C * I; i += NT) s_w[(i / I) * (I + 1) + (i % I)] = p.proj_down[i];
  for (int i = tid; i < 2 * I; i += NT) s_acc[i] = 0.f;
  for (int i = tid; i < TT * C; i += NT) {
    const int tk = i % TT, c = i / TT;
    const int s = s0 + tk;
    s_do[tk * (C + 1) + c] = s < S ? ldf(dout, ((long long)b * C + c) * S + s) : 0.f;
  }
  __syncthreads();
  const int tk = tid / LP, sub = tid % LP;
  const int s = s0 + tk;
  const bool live = s < S;
  // dhg = W^T dout
  for (int j = 0; j < I / LP; ++j) {
    const int k = j * LP + sub;
    float a = 0.f;
#pragma unroll 8
    for (int o = 0; o < C; ++o) a = fmaf(s_w[o * (I + 1) + k], s_do[tk * (C + 1) + o], a);
    s_dhg[tk * (I + 1) + k] = live ? a : 0.f;
  }
  __syncthreads();
  {
    const int h = sub / LH, part = sub % LH, cb = h * DH + part * JH;
    const long long ho = (((long long)b * NH + h) * S + (live ? s : 0)) * DH + part * JH;
    const long long tb = ((long long)b * S + (live ? s : 0)) * I + cb;
    float hv[JH], xh[JH], dn[JH], acc_sk[JH], acc_nw[JH], mean = 0.f, var = 0.f;
#pragma unroll
    for (int j = 0; j < JH; ++j) { hv[j] = live ? w.h[ho + j] : 0.f; mean += hv[j]; }
#pragma unroll
    for (int o = 1; o < LH; o <<= 1) mean += __shfl_xor(mean, o, 64);
    mean /= DH;
#pragma unroll
    for (int j = 0; j < JH; ++j) { const float d = hv[j] - mean; var = fmaf(d, d, var); }
#pragma unroll
    for (int o = 1; o < LH; o <<= 1) var += __shfl_xor(var, o, 64);
    const float rstd = rsqrtf(var / DH + VIL_EPS);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      const int c = cb + j;
      xh[j] = (hv[j] - mean) * rstd;
      const float gam = 1.f + p.outnorm_w[c];
      const float xcv = live ? w.xc[tb + j] : 0.f, zv = live ? w.z[tb + j] : 0.f;
      const float xav = silu_(xcv);
      const float hs = xh[j] * gam + p.skip[c] * xav;
      const float dhg = s_dhg[tk * (I + 1) + c];
      s_hg[tk * (I + 1) + c] = live ? hs * silu_(zv) : 0.f;
      const float dhs = dhg * silu_(zv);
      if (live) {
        w.dz[tb + j] = dhg * hs * dsilu_(zv);
        w.dxa[tb + j] = dhs * p.skip[c];
      }
      acc_sk[j] = live ? dhs * xav : 0.f;        // dskip
      acc_nw[j] = live ? dhs * xh[j] : 0.f;      // d outnorm weight
      dn[j] = dhs * gam;
      m1 += dn[j];
      m2 = fmaf(dn[j], xh[j], m2);
    }
#pragma unroll
    for (int o = 1; o < LH; o <<= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
    m1 /= DH; m2 /= DH;
    if (live) {
#pragma unroll
      for (int j = 0; j < JH; ++j) w.dh[ho + j] = rstd * (dn[j] - m1 - xh[j] * m2);
    }
    // the wave's 4 tokens (lane bits 4, 5) summed by shuffles, then one LDS atomic per wave and value
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      float a = acc_sk[j], c2 = acc_nw[j];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      c2 += __shfl_xor(c2, 16, 64); c2 += __shfl_xor(c2, 32, 64);
      if ((tid & 63) < LP) { atomicAdd(&s_acc[cb + j], a); atomicAdd(&s_acc[I + cb + j], c2); }
    }
  }
  __syncthreads();
  outer_accum(g.proj_down, s_do, C + 1, C, s_hg, I + 1, I, TT);
  for (int i = tid; i < I; i += NT) { atomicAdd(&g.skip[i], s_acc[i]); atomicAdd(&g.outnorm_w[i], s_acc[I + i]); }
}

// per-token preparation for the mLSTM backward: da' = dh/den', db', dm
// one lane per ELEMENT (row, j) -- DH lanes share a row and meet by shuffles: every load and store of the pass is contiguous across the
// wave (round 6; a lane per row walked its DH elements 64 bytes apart from its neighbours': 14 us for 1.3 MB)
template <int DH>
__global__ __launch_bounds__(256) void mlstm_bwd_prep_kernel(int S, long long total, VilWs w) {
  static_assert((DH & (DH - 1)) == 0 && DH <= 64, "DH lanes of a wave share a row");
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;   // over B*NH*S*DH
  const bool ok = e < total * DH;
  const long long ec = ok ? e : 0, i = ec / DH;
  const int j = (int)(ec - i * DH);
  const float bd = w.bden[i];
  const float m = w.F[i] + w.G[i];
  const float em = expf(-m);
  const float nrm = fmaxf(fabsf(bd), em) + MLSTM_EPS;
  const float d = w.dh[ec];
  float dot = d * w.h[ec];
#pragma unroll
  for (int o = 1; o < DH; o <<= 1) dot += __shfl_xor(dot, o, 64);
  if (!ok) return;
  w.dap[e] = d / nrm;
  w.dq[e] = 0.f; w.dk[e] = 0.f; w.dv[e] = 0.f;                     // atomically accumulated below
  if (j == 0) {
    const float dden = -dot / nrm;
    w.dbp[i] = (fabsf(bd) > em) ? (bd > 0.f ? dden : -dden) : 0.f;
    w.dm[i] = dden * MLSTM_EPS;
    w.dscat[i] = 0.f;
  }
}
// scatter the stabiliser gradient onto the arg-max key of every row
__global__ __launch_bounds__(256) void mlstm_bwd_scatter_kernel(int S, long long total, VilWs w) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long row0 = (i / S) * S;
  atomicAdd(&w.dscat[row0 + w.arg[i]], w.dm[i]);
}

// pass A: dq_t = sum_{s<=t} (da'_t.v_s + db'_t) * exp(g_s-G_t) * k_s/sqrt(DH) ; rq_t = q_t.dq_t
template <int DH>
__global__ __launch_bounds__(256) void mlstm_bwd_q_kernel(int S, VilWs w) {
  constexpr int KT = 64, LD = DH + 4;
  __shared__ float s_k[KT * LD], s_v[KT * LD], s_g[KT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const long long hb = ((long long)b * NH + h) * S;
  const int t0 = (blockIdx.x / MSPLIT) * 32, ks = blockIdx.x % MSPLIT;
  const int t = t0 + wv * 8 + (lane >> 3), sl = lane & 7;
  const bool tv = t < S;
  float da[DH], dq[DH];
  float db = 0.f, Gt = 0.f;
#pragma unroll
  for (int j = 0; j < DH; ++j) { da[j] = tv ? w.dap[(hb + t) * DH + j] : 0.f; dq[j] = 0.f; }
  if (tv) { Gt = w.G[hb + t]; db = w.dbp[hb + t]; }
  const float isq = rsqrtf((float)DH);
  const int s_last = min(S - 1, t0 + 31);
  for (int st0 = ks * KT; st0 <= s_last; st0 += MSPLIT * KT) {
    __syncthreads();
    for (int i = tid; i < KT * DH; i += 256) {
      const int r = i / DH, c = i % DH;
      const int s = st0 + r;
      s_k[r * LD + c] = s < S ? w.k[(hb + s) * DH + c] : 0.f;
      s_v[r * LD + c] = s < S ? w.v[(hb + s) * DH + c] : 0.f;
    }
    if (tid < KT) { const int s = st0 + tid; s_g[tid] = s < S ? w.ig[hb + s] - w.F[hb + s] : -INFINITY; }
    __syncthreads();
#pragma unroll 2
    for (int jj = 0; jj < KT / 8; ++jj) {
      const int r = sl + 8 * jj;
      const int s = st0 + r;
      if (tv && s <= t) {
        float dp = db;
#pragma unroll
        for (int j = 0; j < DH; ++j) dp = fmaf(da[j], s_v[r * LD + j], dp);
        const float c = dp * isq * expf(s_g[r] - Gt);
#pragma unroll
        for (int j = 0; j < DH; ++j) dq[j] = fmaf(c, s_k[r * LD + j], dq[j]);
      }
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1)
#pragma unroll
    for (int j = 0; j < DH; ++j) dq[j] += __shfl_xor(dq[j], o, 64);
  if (tv && sl == 0) {
#pragma unroll
    for (int j = 0; j < DH; ++j) atomicAdd(&w.dq[(hb + t) * DH + j], dq[j]);
  }
}
// rq_t = q_t.dq_t and ck_s = k_s.dk_s once every partial has been added
template <int DH>
__global__ __launch_bounds__(256) void mlstm_bwd_dots_kernel(long long total, VilWs w) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over B*NH*S rows
  if (i >= total) return;
  float r = 0.f, c = 0.f;
#pragma unroll
  for (int j = 0; j < DH; ++j) {
    r = fmaf(w.dq[i * DH + j], w.q[i * DH + j], r);
    c = fmaf(w.dk[i * DH + j], w.k[i * DH + j], c);
  }
  w.rq[i] = r;
  w.ck[i] = c;
}

// pass B: per key s: dk_s, dv_s over queries t >= s ; ck_s = k_s.dk_s
template <int DH>
__global__ __launch_bounds__(256) void mlstm_bwd_kv_kernel(int S, VilWs w) {
  constexpr int QT = 64, LD = DH + 4;
  __shared__ float s_q[QT * LD], s_da[QT * LD], s_G[QT], s_db[QT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const long long hb = ((long long)b * NH + h) * S;
  const int s0 = (blockIdx.x / MSPLIT) * 32, ks = blockIdx.x % MSPLIT;
  const int s = s0 + wv * 8 + (lane >> 3), sl = lane & 7;
  const bool sv = s < S;
  float kk[DH], vv[DH], dk[DH], dv[DH];
  float gs = 0.f;
#pragma unroll
  for (int j = 0; j < DH; ++j) {
    kk[j] = sv ? w.k[(hb + s) * DH + j] : 0.f;
    vv[j] = sv ? w.v[(hb + s) * DH + j] : 0.f;
    dk[j] = 0.f; dv[j] = 0.f;
  }
  if (sv) gs = w.ig[hb + s] - w.F[hb + s];
  const float isq = rsqrtf((float)DH);
  for (int qt0 = (s0 / QT + ks) * QT; qt0 < S; qt0 += MSPLIT * QT) {
    __syncthreads();
    for (int i = tid; i < QT * DH; i += 256) {
      const int r = i / DH, c = i % DH;
      const int t = qt0 + r;
      s_q[r * LD + c] = t < S ? w.q[(hb + t) * DH + c] : 0.f;
      s_da[r * LD + c] = t < S ? w.dap[(hb + t) * DH + c] : 0.f;
    }
    if (tid < QT) {
      const int t = qt0 + tid;
      s_G[tid] = t < S ? w.G[hb + t] : INFINITY;
      s_db[tid] = t < S ? w.dbp[hb + t] : 0.f;
    }
    __syncthreads();
#pragma unroll 2
    for (int jj = 0; jj < QT / 8; ++jj) {
      const int r = sl + 8 * jj;
      const int t = qt0 + r;
      if (sv && t >= s && t < S) {
        float qk = 0.f, dp = s_db[r];
#pragma unroll
        for (int j = 0; j < DH; ++j) {
          qk = fmaf(s_q[r * LD + j], kk[j], qk);
          dp = fmaf(s_da[r * LD + j], vv[j], dp);
        }
        const float wgt = isq * expf(gs - s_G[r]);
        const float c1 = dp * wgt, c2 = qk * wgt;
#pragma unroll
        for (int j = 0; j < DH; ++j) {
          dk[j] = fmaf(c1, s_q[r * LD + j], dk[j]);
          dv[j] = fmaf(c2, s_da[r * LD + j], dv[j]);
        }
      }
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1)
#pragma unroll
    for (int j = 0; j < DH; ++j) { dk[j] += __shfl_xor(dk[j], o, 64); dv[j] += __shfl_xor(dv[j], o, 64); }
  if (sv && sl == 0) {
#pragma unroll
    for (int j = 0; j < DH; ++j) {
      atomicAdd(&w.dk[(hb + s) * DH + j], dk[j]);
      atomicAdd(&w.dv[(hb + s) * DH + j], dv[j]);
    }
  }
}

// stage 2 backward (a): gates + q/k/v projections -> dxc (pre-SiLU conv output grad), dxm via v; param grads
// block: 32 tokens x NH * (DH / 4) lanes -- a lane = one 4 x 4 block of the block-diagonal projections (round 6; it was 32 tokens x NH
// lanes, a lane walking its head's 16 channels through every stage on one wave per SIMD; the loops over the workgroup's parameter-
// gradient elements run on four times the threads)
template <int C>
__global__ __launch_bounds__(32 * NH * (2 * C / NH / 4)) void vil_pre2_bwd_kernel(int S, xh_vil_params p, xh_vil_grads g, VilWs w) {
  constexpr int I = 2 * C, DH = I / NH, TT = 32, NB4 = DH / 4, LPT = NH * NB4;
  __shared__ float s_qkv[TT * (3 * I + 1)];    // forward [q,k,v]
  __shared__ float s_d[TT * (3 * I + 1)];      // d[q,k,v]
  __shared__ float s_gate[TT * (2 * NH + 1)];  // di, df per head
  __shared__ float s_gw[2 * NH * 3 * I];
  __shared__ __attribute__((aligned(16))) float s_pw[3 * I * 4];     // q_w | k_w | v_w, each [I][4]
  float* s_x = s_qkv;                          // after the gate gradients: [TT][2 * I + 1] = silu(xc) | xm of the tile's tokens
  const int tid = threadIdx.x, b = blockIdx.y, s0 = blockIdx.x * TT;
  for (int i = tid; i < NH * 3 * I; i += blockDim.x) { s_gw[i] = p.ig_w[i]; s_gw[NH * 3 * I + i] = p.fg_w[i]; }
  for (int i = tid; i < I * 4; i += blockDim.x) { s_pw[i] = p.q_w[i]; s_pw[I * 4 + i] = p.k_w[i]; s_pw[2 * I * 4 + i] = p.v_w[i]; }
  const int tk = tid / LPT, h = (tid / NB4) % NH, blk = tid % NB4;
  const int s = s0 + tk;
  const bool ok = s < S;
  const int cb = h * DH + blk * 4;              // the lane's four channels
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long o_ = (((long long)b * NH + h) * S + (ok ? s : 0)) * DH + blk * 4 + j;
    s_qkv[tk * (3 * I + 1) + cb + j] = ok ? w.q[o_] : 0.f;
    s_qkv[tk * (3 * I + 1) + I + cb + j] = ok ? w.k[o_] : 0.f;
    s_qkv[tk * (3 * I + 1) + 2 * I + cb + j] = ok ? w.v[o_] : 0.f;
    s_d[tk * (3 * I + 1) + cb + j] = ok ? w.dq[o_] : 0.f;
    s_d[tk * (3 * I + 1) + I + cb + j] = ok ? w.dk[o_] : 0.f;
    s_d[tk * (3 * I + 1) + 2 * I + cb + j] = ok ? w.dv[o_] : 0.f;
  }
  if (blk == 0) {
    s_gate[tk * (2 * NH + 1) + h] = ok ? w.di[((long long)b * NH + h) * S + s] : 0.f;
    s_gate[tk * (2 * NH + 1) + NH + h] = ok ? w.df[((long long)b * NH + h) * S + s] : 0.f;
  }
  __syncthreads();
  // gate weight/bias grads
  outer_accum(g.ig_w, s_gate, 2 * NH + 1, NH, s_qkv, 3 * I + 1, 3 * I, TT);
  outer_accum(g.fg_w, s_gate + NH, 2 * NH + 1, NH, s_qkv, 3 * I + 1, 3 * I, TT);
  if (tid < 2 * NH) {
    float a = 0.f;
    for (int t = 0; t < TT; ++t) a += s_gate[t * (2 * NH + 1) + tid];
    atomicAdd(tid < NH ? &g.ig_b[tid] : &g.fg_b[tid - NH], a);
  }
  __syncthreads();
  // d[q,k,v] += W_i^T di + W_f^T df   (each lane updates its own 3 x 4 entries)
  for (int part = 0; part < 3; ++part)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = part * I + cb + j;
      float a = s_d[tk * (3 * I + 1) + m];
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        a = fmaf(s_gw[hh * 3 * I + m], s_gate[tk * (2 * NH + 1) + hh], a);
        a = fmaf(s_gw[NH * 3 * I + hh * 3 * I + m], s_gate[tk * (2 * NH + 1) + NH + hh], a);
      }
      s_d[tk * (3 * I + 1) + m] = a;
    }
  __syncthreads();
  // through the block-diagonal projections
  if (ok) {
    const int gb = h * NB4 + blk;
    float xa4[4], xm4[4], dxa4[4] = {0, 0, 0, 0}, dxm4[4] = {0, 0, 0, 0};
    const long long to0 = ((long long)b * S + s) * I + cb;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      xa4[d] = silu_(w.xc[to0 + d]);
      xm4[d] = w.xm[to0 + d];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int m = cb + o;
      const float dqv = s_d[tk * (3 * I + 1) + m], dkv = s_d[tk * (3 * I + 1) + I + m], dvv = s_d[tk * (3 * I + 1) + 2 * I + m];
      const float4 qw = *reinterpret_cast<const float4*>(s_pw + (gb * 4 + o) * 4);
      const float4 kw = *reinterpret_cast<const float4*>(s_pw + I * 4 + (gb * 4 + o) * 4);
      const float4 vw = *reinterpret_cast<const float4*>(s_pw + 2 * I * 4 + (gb * 4 + o) * 4);
      const float qv[4] = {qw.x, qw.y, qw.z, qw.w}, kv[4] = {kw.x, kw.y, kw.z, kw.w}, vv[4] = {vw.x, vw.y, vw.z, vw.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        dxa4[d] = fmaf(qv[d], dqv, dxa4[d]);
        dxa4[d] = fmaf(kv[d], dkv, dxa4[d]);
        dxm4[d] = fmaf(vv[d], dvv, dxm4[d]);
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float dxa_tot = dxa4[d] + w.dxa[to0 + d];
      w.dxc[to0 + d] = dxa_tot * dsilu_(w.xc[to0 + d]);
      w.dxm[to0 + d] = dxm4[d];      // v-path part; conv part added in the next stage
      s_x[tk * (2 * I + 1) + cb + d] = xa4[d];
      s_x[tk * (2 * I + 1) + I + cb + d] = xm4[d];
    }
  } else {
#pragma unroll
    for (int d = 0; d < 4; ++d) { s_x[tk * (2 * I + 1) + cb + d] = 0.f; s_x[tk * (2 * I + 1) + I + cb + d] = 0.f; }
  }
  __syncthreads();
  // dq_w / dk_w / dv_w [m][d] = sum over the tile's tokens of d{q,k,v}[t][m] * x[t][4 * (m / 4) + d]: one output per thread and
  // round, the token sum read from LDS (LDS atomics from every token's thread were 32 colliding adds per value: 54 us per launch)
  for (int idx = tid; idx < 3 * I * 4; idx += blockDim.x) {
    const int part = idx / (I * 4), r = idx - part * I * 4;
    const int m = r >> 2, xch = (m & ~3) + (r & 3) + (part == 2 ? I : 0);
    float a = 0.f;
#pragma unroll 8
    for (int t = 0; t < TT; ++t) a = fmaf(s_d[t * (3 * I + 1) + part * I + m], s_x[t * (2 * I + 1) + xch], a);
    atomicAdd(&(part == 0 ? g.q_w : part == 1 ? g.k_w : g.v_w)[r], a);
  }
}

// stage 2 backward (b) + stage 1 backward: conv1d backward, proj_up backward, LayerNorm backward, scatter
// block: 32 tokens x 16 lanes (round 6: twice the threads on the same tile -- every loop over the tile's elements and over the
// parameter-gradient elements halves, the token count per workgroup, i.e. the depth of the global atomics, stays)
constexpr int PRE1B_LP = 16, PRE1B_NT = 32 * PRE1B_LP;
template <typename T, int C>
__global__ __launch_bounds__(PRE1B_NT) void vil_pre1_bwd_kernel(const T* dout, T* dxin, int S, xh_vil_params p, xh_vil_grads g,
                                                          VilWs w) {
  constexpr int I = 2 * C, TT = 32, O = 4 * C, LP = PRE1B_LP, NT = PRE1B_NT;
  static_assert(C % LP == 0, "lane map of vil_pre1_bwd_kernel");
  __shared__ float s_w[O * (C + 1)];
  __shared__ float s_dxc[(TT + 3) * (I + 1)];
  __shared__ float s_xm[(TT + 3) * (I + 1)];
  __shared__ float s_din[TT * (O + 1)];     // d[xm, z]
  __shared__ float s_ln[TT * (C + 1)];      // ln output (for dW_up), then reused for dln
  __shared__ float s_dt[TT * (C + 1)];      // dtok
  __shared__ float s_cw[I * 5];             // dconv_w (4) + dconv_b
  __shared__ float s_nw[C];
  const int tid = threadIdx.x, b = blockIdx.y, s0 = blockIdx.x * TT;
  for (int i = tid; i < O * C; i += NT) s_w[(i / C) * (C + 1) + (i % C)] = p.proj_up[i];
  for (int i = tid; i < I * 5; i += NT) s_cw[i] = 0.f;
  for (int i = tid; i < C; i += NT) s_nw[i] = 0.f;
  // dxc rows s0 .. s0+TT+2 ; xm rows s0-3 .. s0+TT-1
  for (int i = tid; i < (TT + 3) * I; i += NT) {
    const int r = i / I, c = i % I;
    const int sd = s0 + r, sx = s0 - 3 + r;
    s_dxc[r * (I + 1) + c] = sd < S ? w.dxc[((long long)b * S + sd) * I + c] : 0.f;
    s_xm[r * (I + 1) + c] = (sx >= 0 && sx < S) ? w.xm[((long long)b * S + sx) * I + c] : 0.f;
  }
  __syncthreads();
  // conv backward: dxm[s][c] = sum_j w[c][j]*dxc[s+3-j][c];  dw[c][j] += sum_s dxc[s][c]*xm[s-3+j][c]
  for (int i = tid; i < TT * I; i += NT) {
    const int tk = i / I, c = i % I;
    const int s = s0 + tk;
    float a = 0.f;
    if (s < S) {
      a = w.dxm[((long long)b * S + s) * I + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) a = fmaf(p.conv_w[c * 4 + j], s_dxc[(tk + 3 - j) * (I + 1) + c], a);
    }
    s_din[tk * (O + 1) + c] = a;
    s_din[tk * (O + 1) + I + c] = s < S ? w.dz[((long long)b * S + s) * I + c] : 0.f;
  }
  for (int i = tid; i < I * 5; i += NT) {
    const int c = i / 5, j = i % 5;
    float a = 0.f;
    for (int tk = 0; tk < TT; ++tk) {
      const float d = s_dxc[tk * (I + 1) + c];          // dxc at token s0+tk (zero beyond S)
      a = fmaf(d, j < 4 ? s_xm[(tk + j) * (I + 1) + c] : 1.f, a);
    }
    s_cw[i] = a;
  }
  // recompute LayerNorm output of the tokens
  const int tk = tid / LP, sub = tid % LP;
  const int s = s0 + tk;
  const bool ok = s < S;
  float mean = 0.f, rstd = 0.f;
  if (ok) { mean = w.ln_mean[(long long)b * S + s]; rstd = w.ln_rstd[(long long)b * S + s]; }
  for (int c = sub; c < C; c += LP) {
    const float xh = ok ? (w.tok[((long long)b * S + s) * C + c] - mean) * rstd : 0.f;
    s_ln[tk * (C + 1) + c] = xh * (1.f + p.norm_w[c]);
    s_dt[tk * (C + 1) + c] = xh;   // keep xhat
  }
  __syncthreads();
  for (int i = tid; i < I * 5; i += NT) {
    const int c = i / 5, j = i % 5;
    atomicAdd(j < 4 ? &g.conv_w[c * 4 + j] : &g.conv_b[c], s_cw[i]);
  }
  outer_accum(g.proj_up, s_din, O + 1, O, s_ln, C + 1, C, TT);
  __syncthreads();
  // dln = W_up^T d[xm,z]
  float dln[C / LP], xh[C / LP];
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int j = 0; j < C / LP; ++j) {
    const int c = sub + LP * j;
    float a = 0.f;
    for (int o = 0; o < O; ++o) a = fmaf(s_w[o * (C + 1) + c], s_din[tk * (O + 1) + o], a);
    xh[j] = s_dt[tk * (C + 1) + c];
    atomicAdd(&s_nw[c], a * xh[j]);
    dln[j] = a * (1.f + p.norm_w[c]);
    m1 += dln[j];
    m2 = fmaf(dln[j], xh[j], m2);
  }
#pragma unroll
  for (int o = 1; o < LP; o <<= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
  m1 /= C; m2 /= C;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < C / LP; ++j) {
    const int c = sub + LP * j;
    s_dt[tk * (C + 1) + c] = rstd * (dln[j] - m1 - xh[j] * m2);
  }
  __syncthreads();
  for (int i = tid; i < C; i += NT) atomicAdd(&g.norm_w[i], s_nw[i]);
  for (int i = tid; i < TT * C; i += NT) {
    const int tk2 = i % TT, c = i / TT;
    const int s2 = s0 + tk2;
    if (s2 < S) {
      const long long o = ((long long)b * C + c) * S + s2;
      stf(dxin, o, ldf(dout, o) + s_dt[tk2 * (C + 1) + c]);
    }
  }
}

// =================================================================================================
// host entry points
// =================================================================================================
template <typename T, int C>
static int vil_fwd_impl(hipStream_t st, const T* xa, const T* xb, T* out, int B, int S, int add_xa, const xh_vil_params* p, float* ws) {
  constexpr int DH = 2 * C / NH;
  VilWs w;
  ws_layout(&w, ws, B, S, C);
  hipLaunchKernelGGL((vil_pre1_kernel<T, C>), dim3(cdiv(S, PRE1_TT), B), dim3(PRE1_TT * PRE1_LP), 0, st, xa, xb, S, *p, w);
  hipLaunchKernelGGL((vil_pre2_kernel<C>), dim3(cdiv(S, PRE2_TT), B), dim3(PRE2_TT * NH * (2 * C / NH / 4)), 0, st, S, *p, w);
  hipLaunchKernelGGL(vil_scan_kernel, dim3(B * NH), dim3(SCAN_T), 0, st, S, w);
  const long long rows_f = (long long)B * NH * S;
  if (DH == 16 && !(g_xh_disable & 8)) {              // chunk-recurrent form on the matrix cores
    const int nchunk = cdiv(S, CL);
    hipLaunchKernelGGL(mlstm_chunk_dstate_kernel<0>, dim3(nchunk, NH, B), dim3(64), 0, st, S, nchunk, w);
    hipLaunchKernelGGL(mlstm_chunk_scan_kernel<0>, dim3(B * NH), dim3(320), 0, st, S, nchunk, w);
    hipLaunchKernelGGL(mlstm_chunk_fwd_kernel, dim3(nchunk, NH, B), dim3(256), 0, st, S, nchunk, w);
  } else {
    (void)hipMemsetAsync(w.h, 0, (size_t)rows_f * DH * sizeof(float), st);
    (void)hipMemsetAsync(w.bden, 0, (size_t)rows_f * sizeof(float), st);
    hipLaunchKernelGGL((mlstm_fwd_kernel<DH>), dim3(cdiv(S, 32) * MSPLIT, NH, B), dim3(256), 0, st, S, w);
    hipLaunchKernelGGL((mlstm_norm_kernel<DH>), dim3((unsigned)((rows_f + 255) / 256)), dim3(256), 0, st, rows_f, w);
  }
  hipLaunchKernelGGL((vil_post_kernel<T, C>), dim3(cdiv(S, POST_TT), B), dim3(POST_TT * POST_LP), 0, st, xa, out, S, add_xa, *p, w);
  return xh_launch_status();
}
template <typename T, int C>
static int vil_bwd_impl(hipStream_t st, const T* dout, T* dxin, int B, int S, const xh_vil_params* p, const xh_vil_grads* g,
                        float* ws) {
  constexpr int DH = 2 * C / NH;
  VilWs w;
  ws_layout(&w, ws, B, S, C);
  const long long rows = (long long)B * NH * S;
  hipLaunchKernelGGL((vil_post_bwd_kernel<T, C>), dim3(cdiv(S, 32), B), dim3(512), 0, st, dout, S, *p, *g, w);
  hipLaunchKernelGGL((mlstm_bwd_prep_kernel<DH>), dim3((unsigned)((rows * DH + 255) / 256)), dim3(256), 0, st, S, rows, w);
  hipLaunchKernelGGL(mlstm_bwd_scatter_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, S, rows, w);
  if (DH == 16 && !(g_xh_disable & 8)) {
    // the forward's carried states C, n are still in the workspace; the reverse states are built here
    const int nchunk = cdiv(S, CL);
    hipLaunchKernelGGL(mlstm_chunk_dstate_kernel<1>, dim3(nchunk, NH, B), dim3(64), 0, st, S, nchunk, w);
    hipLaunchKernelGGL(mlstm_chunk_scan_kernel<1>, dim3(B * NH), dim3(320), 0, st, S, nchunk, w);
    hipLaunchKernelGGL(mlstm_chunk_bwd_kernel, dim3(nchunk, NH, B), dim3(256), 0, st, S, nchunk, w);
  } else {
    hipLaunchKernelGGL((mlstm_bwd_q_kernel<DH>), dim3(cdiv(S, 32) * MSPLIT, NH, B), dim3(256), 0, st, S, w);
    hipLaunchKernelGGL((mlstm_bwd_kv_kernel<DH>), dim3(cdiv(S, 32) * MSPLIT, NH, B), dim3(256), 0, st, S, w);
  }
  hipLaunchKernelGGL((mlstm_bwd_dots_kernel<DH>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, rows, w);
  hipLaunchKernelGGL(vil_rscan_kernel, dim3(B * NH), dim3(SCAN_T), 0, st, S, w);
  hipLaunchKernelGGL((vil_pre2_bwd_kernel<C>), dim3(cdiv(S, 32), B), dim3(32 * NH * (2 * C / NH / 4)), 0, st, S, *p, *g, w);
  hipLaunchKernelGGL((vil_pre1_bwd_kernel<T, C>), dim3(cdiv(S, 32), B), dim3(PRE1B_NT), 0, st, dout, dxin, S, *p, *g, w);
  return xh_launch_status();
}

static int vil_check(int dtype, int B, int S, int C, int nh, const xh_vil_params* p, const float* ws) {
  if (dtype != XH_F32 && dtype != XH_BF16 && dtype != XH_F16) return XH_ERR_DTYPE;
  if (B <= 0 || S <= 0 || B > 65535 || nh != NH || !p || !ws) return XH_ERR_ARG;
  if (!(C == 16 || C == 32)) return XH_ERR_ARG;
  if (!p->norm_w || !p->proj_up || !p->conv_w || !p->conv_b || !p->q_w || !p->k_w || !p->v_w || !p->ig_w || !p->ig_b ||
      !p->fg_w || !p->fg_b || !p->outnorm_w || !p->skip || !p->proj_down)
    return XH_ERR_ARG;
  return XH_OK;
}

extern "C" int xh_vil_fwd(void* stream, int dtype, const void* xa, const void* xb, void* out, int B, int S, int C, int nh,
                          int add_xa, const xh_vil_params* p, float* ws) {
  int rc = vil_check(dtype, B, S, C, nh, p, ws);
  if (rc) return rc;
  if (!xa || !out) return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype,
    if (C == 32) return vil_fwd_impl<T, 32>(st, (const T*)xa, (const T*)xb, (T*)out, B, S, add_xa, p, ws);
    return vil_fwd_impl<T, 16>(st, (const T*)xa, (const T*)xb, (T*)out, B, S, add_xa, p, ws););
}

extern "C" int xh_vil_bwd(void* stream, int dtype, const void* xa, const void* xb, const void* dout, void* dxin, int B,
                          int S, int C, int nh, const xh_vil_params* p, const xh_vil_grads* g, float* ws) {
  int rc = vil_check(dtype, B, S, C, nh, p, ws);
  if (rc) return rc;
  (void)xa; (void)xb;
  if (!dout || !dxin || !g) return XH_ERR_ARG;
  if (!g->norm_w || !g->proj_up || !g->conv_w || !g->conv_b || !g->q_w || !g->k_w || !g->v_w || !g->ig_w || !g->ig_b ||
      !g->fg_w || !g->fg_b || !g->outnorm_w || !g->skip || !g->proj_down)
    return XH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  XH_DISPATCH_T(dtype,
    if (C == 32) return vil_bwd_impl<T, 32>(st, (const T*)dout, (T*)dxin, B, S, p, g, ws);
    return vil_bwd_impl<T, 16>(st, (const T*)dout, (T*)dxin, B, S, p, g, ws););
}
