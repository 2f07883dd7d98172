// Shared device helpers for the XLSTM-HVED gfx950 kernel library.
// Storage types: float, bf16 or fp16 (raw 16-bit), arithmetic is always fp32 (double for cross-block sums).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define XH_OK 0
#define XH_ERR_ARG (-1)       // bad shape / unsupported combination
#define XH_ERR_DTYPE (-2)
#define XH_ERR_HIP (-3)       // launch failed

#define XH_F32 0
#define XH_BF16 1
#define XH_F16 2

#define XH_ACT_NONE 0
#define XH_ACT_RELU 1
#define XH_ACT_LRELU 2
#define XH_ACT_SIGMOID 3

// 16-bit storage formats: one struct template so that every helper below is written once.  FMT 0 = bfloat16 (8 exponent /
// 7 mantissa bits), FMT 1 = IEEE half (5 / 10: the reference's own AMP dtype, train.py:218).
template <int FMT> struct h16 { unsigned short v; };
typedef h16<0> bf16_t;
typedef h16<1> f16_t;

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float hf2f(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ unsigned short f2hf(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }   // RNE, overflow -> inf

// format-generic scalar / packed-pair conversions
template <int FMT> __device__ __forceinline__ float cvt_in(unsigned short v) { return FMT == 0 ? bf2f(v) : hf2f(v); }
template <int FMT> __device__ __forceinline__ unsigned short cvt_out(float f) { return FMT == 0 ? f2bf(f) : f2hf(f); }
template <int FMT> __device__ __forceinline__ float cvt_lo(unsigned u) {       // low half of a packed pair
  return FMT == 0 ? __uint_as_float(u << 16) : hf2f((unsigned short)(u & 0xffffu));
}
template <int FMT> __device__ __forceinline__ float cvt_hi(unsigned u) {
  return FMT == 0 ? __uint_as_float(u & 0xffff0000u) : hf2f((unsigned short)(u >> 16));
}
template <int FMT> __device__ __forceinline__ unsigned cvt_pack(float a, float b) {
  return (unsigned)cvt_out<FMT>(a) | ((unsigned)cvt_out<FMT>(b) << 16);
}
// Explicit two-wide forms: a <2 x float> is what hipcc turns into v_pk_fma_f32 / v_pk_mul_f32, and a vector conversion is
// ONE v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 of exactly the pair given (two scalar conversions of values that end up in one
// dword let the compiler pair them its own way and then re-shuffle the halves: and + lshl + 2 x or_sdwa per dword).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <int FMT> __device__ __forceinline__ f32x2_t cvt2_in(unsigned u) { return f32x2_t{cvt_lo<FMT>(u), cvt_hi<FMT>(u)}; }
template <int FMT> __device__ __forceinline__ unsigned cvt2_pack(float a, float b) {      // RNE, a -> low half
  const f32x2_t v = {a, b};
  if constexpr (FMT == 0) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ f32x2_t max2(f32x2_t a, f32x2_t b) { return f32x2_t{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
template <typename T> struct FmtOf { static constexpr int v = -1; };
template <int FMT> struct FmtOf<h16<FMT>> { static constexpr int v = FMT; };

__device__ __forceinline__ float ldf(const float* p, long long i) { return p[i]; }
template <int F> __device__ __forceinline__ float ldf(const h16<F>* p, long long i) { return cvt_in<F>(p[i].v); }
__device__ __forceinline__ void stf(float* p, long long i, float v) { p[i] = v; }
template <int F> __device__ __forceinline__ void stf(h16<F>* p, long long i, float v) { p[i].v = cvt_out<F>(v); }
// value as it will be read back from storage
__device__ __forceinline__ float rnd_as(const float*, float v) { return v; }
template <int F> __device__ __forceinline__ float rnd_as(const h16<F>*, float v) { return cvt_in<F>(cvt_out<F>(v)); }

// 4-wide contiguous access (caller guarantees alignment: 16 B for float, 8 B for the 16-bit formats)
__device__ __forceinline__ void ld4(const float* p, long long i, float (&o)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p + i); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
template <int F> __device__ __forceinline__ void ld4(const h16<F>* p, long long i, float (&o)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p + i);
  o[0] = cvt_lo<F>(t.x); o[1] = cvt_hi<F>(t.x);
  o[2] = cvt_lo<F>(t.y); o[3] = cvt_hi<F>(t.y);
}
__device__ __forceinline__ void st4(float* p, long long i, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]);
}
template <int F> __device__ __forceinline__ void st4(h16<F>* p, long long i, const float (&v)[4]) {
  uint2 t;
  t.x = cvt_pack<F>(v[0], v[1]);
  t.y = cvt_pack<F>(v[2], v[3]);
  *reinterpret_cast<uint2*>(p + i) = t;
}

// widest (16-byte) contiguous access per lane: 4 floats or 8 16-bit values
template <typename T> struct VWT { static constexpr int v = 4; };
template <int F> struct VWT<h16<F>> { static constexpr int v = 8; };
__device__ __forceinline__ void ldvec(const float* p, long long q, float (&o)[4]) { ld4(p, q, o); }
__device__ __forceinline__ void stvec(float* p, long long q, const float (&o)[4]) { st4(p, q, o); }
template <int F> __device__ __forceinline__ void ldvec(const h16<F>* p, long long q, float (&o)[8]) {
  const uint4 t = *reinterpret_cast<const uint4*>(p + q);
  const unsigned u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { o[2 * k] = cvt_lo<F>(u[k]); o[2 * k + 1] = cvt_hi<F>(u[k]); }
}
template <int F> __device__ __forceinline__ void stvec(h16<F>* p, long long q, const float (&o)[8]) {
  uint4 t;
  t.x = cvt_pack<F>(o[0], o[1]);
  t.y = cvt_pack<F>(o[2], o[3]);
  t.z = cvt_pack<F>(o[4], o[5]);
  t.w = cvt_pack<F>(o[6], o[7]);
  *reinterpret_cast<uint4*>(p + q) = t;
}

// two-element runs (4 bytes of 16-bit storage, 8 of fp32): the narrow lanes of the deep-level element-wise kernels (eltwise.hip)
__device__ __forceinline__ void ldvec(const float* p, long long q, float (&o)[2]) {
  const float2 t = *reinterpret_cast<const float2*>(p + q); o[0] = t.x; o[1] = t.y;
}
__device__ __forceinline__ void stvec(float* p, long long q, const float (&o)[2]) { *reinterpret_cast<float2*>(p + q) = make_float2(o[0], o[1]); }
template <int F> __device__ __forceinline__ void ldvec(const h16<F>* p, long long q, float (&o)[2]) {
  const unsigned u = *reinterpret_cast<const unsigned*>(p + q);
  o[0] = cvt_lo<F>(u); o[1] = cvt_hi<F>(u);
}
template <int F> __device__ __forceinline__ void stvec(h16<F>* p, long long q, const float (&o)[2]) {
  *reinterpret_cast<unsigned*>(p + q) = cvt_pack<F>(o[0], o[1]);
}

// 16x16x32 MFMA on either 16-bit format (operands carried as 8 raw shorts = 4 VGPRs; fp32 accumulate)
typedef short h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int FMT> __device__ __forceinline__ f32x4_t mfma16x16x32(h16x8 a, h16x8 b, f32x4_t c) {
  if constexpr (FMT == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// one dtype switch for every entry point: expands BODY with T = float / bf16_t / f16_t (else: return XH_ERR_DTYPE)
#define XH_DISPATCH_T(dtype, ...)                                   \
  do {                                                              \
    if ((dtype) == XH_F32) { typedef float T; __VA_ARGS__ }         \
    else if ((dtype) == XH_BF16) { typedef bf16_t T; __VA_ARGS__ }  \
    else if ((dtype) == XH_F16) { typedef f16_t T; __VA_ARGS__ }    \
    else return XH_ERR_DTYPE;                                       \
  } while (0)

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case XH_ACT_RELU: return v > 0.f ? v : 0.f;
    case XH_ACT_LRELU: return leaky(v, slope);
    case XH_ACT_SIGMOID: return sigmoidf_(v);
    default: return v;
  }
}

// Sigmoid whose STORED value keeps the side of 0.5 its logit is on.  The segmentation is thresholded at `> 0.5` (metrics.py:85-107,
// RA_HVED.py:640-641): with 16-bit storage a probability in (0.5, 0.5 + ulp / 2] rounds to exactly 0.5 and lands on the wrong side
// although the logit is positive -- tools/precision_sweep.py: 209 of 6.3 M voxels at bf16, 17 at fp16 from this rounding alone.
// Such a value is stored as the format's next value above 0.5 instead (an error of <= 1 ulp instead of <= 1/2 ulp on these voxels
// only); a probability below 0.5 that rounds up to 0.5 already fails `> 0.5` as it should.  fp32 storage: unchanged.
__device__ __forceinline__ float apply_act_as(const float*, float v, int act, float slope) { return apply_act(v, act, slope); }
template <int F> __device__ __forceinline__ float apply_act_as(const h16<F>* yp, float v, int act, float slope) {
  float o = apply_act(v, act, slope);
  if (act == XH_ACT_SIGMOID && v > 0.f && rnd_as(yp, o) <= 0.5f) o = F == 0 ? 0.50390625f : 0.50048828125f;
  return o;
}

// 64-lane wave reductions.  The four steps inside a 16-lane row are DPP moves (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror: full-rate vector instructions); only the two steps across rows go through the LDS crossbar (ds_bpermute).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row; every lane gets it
template <typename V> __device__ __forceinline__ V row16_sum(V v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum of NV per-thread values (block = NW waves); result valid in threads 0..NV-1 as ret[...] via smem.
// smem must hold NW*NV floats.  After the call smem[0..NV) (first row) holds the totals (all threads may read
// them after the trailing barrier).
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* smem, int nwaves) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = wave_sum(v[i]);
    if (lane == 0) smem[wid * NV + i] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < NV) {
    float s = 0.f;
    for (int w = 0; w < nwaves; ++w) s += smem[w * NV + threadIdx.x];
    smem[threadIdx.x] = s;
  }
  __syncthreads();
}

// fp64 flavour for the statistics reductions (norm moments, norm-backward sums): the network amplifies an error in them
// ~1e4x, and fp64 adds are free next to the memory stream.  smem must hold NW*NV doubles; totals land in smem[0..NV).
template <int NV>
__device__ __forceinline__ void block_sum_d(double (&v)[NV], double* smem, int nwaves) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double s = wave_sum(v[i]);
    if (lane == 0) smem[wid * NV + i] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < NV) {
    double s = 0.0;
    for (int w = 0; w < nwaves; ++w) s += smem[w * NV + threadIdx.x];
    smem[threadIdx.x] = s;
  }
  __syncthreads();
}

// Workgroups are dealt round-robin to the 8 XCDs (each with a private 4 MiB L2).  Remapping the linear block id so that
// XCD x owns the contiguous range [x*nb/8, (x+1)*nb/8) makes spatially adjacent tiles share an L2, so halo re-reads hit
// there instead of going to the fabric.  Identity when the grid is not a multiple of 8.
__device__ __forceinline__ int xcd_swizzle(int b, int nb) { return (nb & 7) ? b : (b & 7) * (nb >> 3) + (b >> 3); }

// name of the template instance the last conv launch used (bench.py's roofline object reports it; rocprof names agree)
void xh_note_kernel(const char* fmt, ...);
extern int g_xh_disable;     // xh_set_option(2, mask): bit 0 = no sliding-window depthwise kernel, bit 1 = no exact-2x upsample kernels
// hipFuncSetAttribute (dynamic LDS above 64 KB) holds per DEVICE: a launch site remembers the devices it has set it on
#define XH_MAX_DEV 64
static inline bool xh_attr_needed(bool (&done)[XH_MAX_DEV]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= XH_MAX_DEV) return true;
  if (done[dev]) return false;
  done[dev] = true;
  return true;
}
static inline int xh_launch_status() { return hipGetLastError() == hipSuccess ? XH_OK : XH_ERR_HIP; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Division of a small non-negative index by a launch constant without the 25-instruction reciprocal sequence:
// m = udiv_magic(d) on the host, q = udiv_fast(n, d, m) on the device; exact while n * d < 2^32 (hosts check their grids).
static inline unsigned udiv_magic(int d) { return d > 1 ? (unsigned)((1ull << 32) / (unsigned)d) + 1u : 0u; }
__device__ __forceinline__ int udiv_fast(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }

