// Probe (round 5): throughput of the transposed LDS read ds_read_b64_tr_b16 (the fragment read of the discriminator's weight
// gradients) next to plain ds_read_b64 / ds_read_b128, with the two address patterns dwgrad_halo_kernel uses.
// (A first version of this probe spent ~8 vector instructions per read and measured vector issue, ~62 B/clk for every pattern.)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_tr_read tools/probe/lds_tr_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sw256(int row, int chunk) { return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

// MODE 0: tr read, dY pattern (256-byte rows, sw256); 1: tr read, source-block pattern (128-byte rows hz*32+hy*6+hx);
// 2: plain b64 at the dY pattern's addresses; 3: plain b128 linear; 4: tr read, linear rows of 128 bytes without swizzle;
// 5: source-block rows without swizzle; 8: dense 5 x 5 x 5 rows; 6: 256-byte rows without swizzle; 7: two panels of 128-byte rows
template <int MODE>
__global__ __launch_bounds__(512) void timeit(unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(256))) unsigned char s[];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(s)[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4, q = r16 >> 2, p = r16 & 3, ph = p >> 1, pb = (p & 1) * 8;
  unsigned base[2];
  for (int hi = 0; hi < 2; ++hi) {
    const int row = 8 * kg + q + 4 * hi;
    if (MODE == 0 || MODE == 2) {
      const int ca = ((row & 3) << 2) | ((row >> 2) & 3);
      base[hi] = row * 256 + ((ph ^ ca) << 4) + pb;
    } else if (MODE == 1) {
      const int jz = wv >> 2, jy = (wv >> 1) & 1, jx = wv & 1;
      const int hr = ((kg >> 1) + jz) * 32 + (2 * (kg & 1) + hi + jy) * 6 + q + jx;
      base[hi] = hr * 128 + (((hr >> 1) & 3) << 5) + (ph << 4) + pb;
    } else if (MODE == 4 || MODE == 7) {
      base[hi] = row * 128 + (ph << 4) + pb;
    } else if (MODE == 5) {
      const int jz = wv >> 2, jy = (wv >> 1) & 1, jx = wv & 1;
      const int hr = ((kg >> 1) + jz) * 32 + (2 * (kg & 1) + hi + jy) * 6 + q + jx;
      base[hi] = hr * 128 + (ph << 4) + pb;
    } else if (MODE == 6) {
      base[hi] = row * 256 + (ph << 4) + pb;
    } else if (MODE == 8) {
      const int hr = ((kg >> 1)) * 25 + (2 * (kg & 1) + hi) * 5 + q + wv;
      base[hi] = hr * 128 + (ph << 4) + pb;
    } else {
      base[hi] = (lane + 64 * hi) * 16;
    }
  }
  unsigned acc = 0, acc2 = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = (MODE == 1 || MODE == 4 || MODE == 5 || MODE == 8) ? (u & 3) + (u >> 2) * 1024 : (MODE == 7 ? (u & 3) + (u >> 2) * 256 : u);   // (distinct addresses: equal ones would be merged)
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) {
        const unsigned a = (base[hi] ^ (unsigned)(t << 5)) + ((it & 1) << 13) + ((MODE == 3) ? u * 2048 : 0);
        if (MODE == 3) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(s + a);
          acc ^= v.x ^ v.w; acc2 ^= v.y ^ v.z;
        } else if (MODE == 2) {
          const u32x2 v = *reinterpret_cast<const u32x2*>(s + a);
          acc ^= v.x; acc2 ^= v.y;
        } else {
          const s4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_t __attribute__((address_space(3)))*)(s + a));
          const u32x2 w = __builtin_bit_cast(u32x2, v);
          acc ^= w.x; acc2 ^= w.y;                          // two VALU per read: the loop stays bound by the LDS, not by vector issue
        }
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc2;
}

int main() {
  unsigned* dacc;
  hipMalloc(&dacc, 1024 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* nm, int bytes_per_lane) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    kern<<<256, 512, 65536>>>(dacc, 10); hipDeviceSynchronize();
    hipEventRecord(e0); kern<<<256, 512, 65536>>>(dacc, 4000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * 512 * 4000 * 16 * bytes_per_lane;
    printf("%s: %.3f ms, %.1f TB/s = %.1f B/clk/CU at 2.4 GHz\n", nm, ms, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / (256 * 2.4e9));
  };
  run(timeit<0>, "tr_b16 b64, dY pattern (256 B rows, swizzled)     ", 8);
  run(timeit<1>, "tr_b16 b64, source-block pattern (128 B rows)     ", 8);
  run(timeit<4>, "tr_b16 b64, 128 B rows without swizzle            ", 8);
  run(timeit<5>, "tr_b16 b64, source-block rows without swizzle     ", 8);
  run(timeit<8>, "tr_b16 b64, dense 5x5x5 source rows, no swizzle   ", 8);
  run(timeit<6>, "tr_b16 b64, 256 B rows without swizzle            ", 8);
  run(timeit<7>, "tr_b16 b64, two panels of 128 B rows, no swizzle  ", 8);
  run(timeit<2>, "plain b64 at the dY pattern's addresses           ", 8);
  run(timeit<3>, "plain b128, linear                                ", 16);
  return 0;
}
