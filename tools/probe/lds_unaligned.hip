// Probe (round 5): does gfx950 serve a 16-byte LDS read from a 2-byte-aligned address (hipcc emits one ds_read_b128 for it), is the
// data right, and what does it cost next to an aligned read?   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_unaligned tools/probe/lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U16B { u32x4 v; };

__global__ void check(const unsigned short* in, unsigned short* out, int sh) {
  __shared__ __attribute__((aligned(16))) unsigned short s[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = in[i];
  __syncthreads();
  const U16B* p = reinterpret_cast<const U16B*>(s + threadIdx.x * 8 + sh);
  const u32x4 v = p->v;
  *reinterpret_cast<u32x4*>(out + threadIdx.x * 8) = v;
}

template <int SH>
__global__ void timeit(const unsigned short* in, unsigned* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned short s[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = in[i & 8191];
  __syncthreads();
  unsigned acc = 0;
  int base = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 1024 + SH;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const U16B* p = reinterpret_cast<const U16B*>(s + ((base + u * 512) & 8191));
      const u32x4 v = p->v;
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    base += 8;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  std::vector<unsigned short> h(8192);
  for (int i = 0; i < 8192; ++i) h[i] = (unsigned short)(i * 7 + 3);
  unsigned short *din, *dout;
  unsigned* dacc;
  hipMalloc(&din, 8192 * 2); hipMalloc(&dout, 512 * 8 * 2); hipMalloc(&dacc, 1024 * 256 * 4);
  hipMemcpy(din, h.data(), 8192 * 2, hipMemcpyHostToDevice);
  int bad = 0;
  for (int sh = 0; sh < 8; ++sh) {
    check<<<1, 512>>>(din, dout, sh);
    std::vector<unsigned short> o(512 * 8);
    hipMemcpy(o.data(), dout, o.size() * 2, hipMemcpyDeviceToHost);
    int b = 0;
    for (int t = 0; t < 512; ++t) for (int j = 0; j < 8; ++j) b += o[t * 8 + j] != h[t * 8 + sh + j];
    printf("shift %d elements (%d bytes): %s\n", sh, sh * 2, b ? "WRONG" : "ok");
    bad += b;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* nm) {
    kern<<<1024, 256>>>(din, dacc, 10); hipDeviceSynchronize();
    hipEventRecord(e0); kern<<<1024, 256>>>(din, dacc, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 1024.0 * 256 * 2000 * 8 * 16;
    printf("%s: %.3f ms, %.1f TB/s of LDS reads (256 CUs x 128 B/clk x 2.4 GHz = 78.6 TB/s)\n", nm, ms, bytes / ms / 1e9);
  };
  run(timeit<0>, "aligned        ");
  run(timeit<1>, "shift 2 bytes  ");
  run(timeit<2>, "shift 4 bytes  ");
  run(timeit<4>, "shift 8 bytes  ");
  run(timeit<7>, "shift 14 bytes ");
  return bad != 0;
}
