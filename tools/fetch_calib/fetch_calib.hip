// Known-bytes calibration of the TCC FETCH_SIZE counter on gfx950 for the access patterns of this repo's kernels
// (VERDICT r2 #3c: conv3_wgrad_q4_multi_kernel showed FETCH_SIZE x 2 = 1.75 x its algorithmic bytes -- real re-reads, or the
// "x 2" rule of MI355X_MICROARCH.md over-counting loads that are not 16 B per lane over a whole wave?).
//
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib
//
// Every kernel reads EXACTLY `bytes` distinct bytes of a 1 GiB buffer once (cold: a 2 GiB write to another buffer precedes
// each launch) and writes 4 bytes per workgroup; tools/fetch_calib/summarize.py divides FETCH_SIZE (KiB) by the known bytes.
//   full16    every lane loads 16 B, a wave covers 1 KiB contiguous            (the pattern the x2 rule was stated for)
//   full8     every lane loads 8 B, a wave covers 512 B contiguous
//   full4     every lane loads 4 B, a wave covers 256 B contiguous
//   quad16    one lane in four loads 16 B (the kw = 1 lane of each quad in conv3_wgrad_q4: dY), the 16 loading lanes of a
//             wave cover 256 B contiguous; the other lanes are masked off
//   quad16rep all four lanes of a quad load the SAME 16 B (replicated-lane form of the same fetch)
//   row64     a wave loads 16 distinct 64-byte row pieces (4 lanes x 16 B each) at a 256-byte pitch: the x operand of
//             conv3_wgrad_q4 (16 distinct half-lines per load)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void sink(unsigned acc, unsigned* out) { if (threadIdx.x == 0 || acc == 0x12345u) out[blockIdx.x] = acc; }

__global__ __launch_bounds__(256) void full16(const uint4* p, unsigned* out, long long n16) {
  unsigned acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  sink(acc, out);
}
__global__ __launch_bounds__(256) void full8(const uint2* p, unsigned* out, long long n8) {
  unsigned acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) { const uint2 v = p[i]; acc ^= v.x ^ v.y; }
  sink(acc, out);
}
__global__ __launch_bounds__(256) void full4(const unsigned* p, unsigned* out, long long n4) {
  unsigned acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) acc ^= p[i];
  sink(acc, out);
}
// 64 loading lanes per workgroup (one per quad): chunk index = wg-iteration * 64 + quad
__global__ __launch_bounds__(256) void quad16(const uint4* p, unsigned* out, long long n16) {
  unsigned acc = 0;
  const int quad = threadIdx.x >> 2;
  for (long long i = (long long)blockIdx.x * 64 + quad; i < n16; i += (long long)gridDim.x * 64)
    if ((threadIdx.x & 3) == 1) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  sink(acc, out);
}
__global__ __launch_bounds__(256) void quad16rep(const uint4* p, unsigned* out, long long n16) {
  unsigned acc = 0;
  const int quad = threadIdx.x >> 2;
  for (long long i = (long long)blockIdx.x * 64 + quad; i < n16; i += (long long)gridDim.x * 64) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  sink(acc, out);
}
// a wave = 16 rows x 64 B; rows 256 B apart; the four 64-byte columns of a 256-byte row group are taken by the 4 waves of the
// workgroup, so that every byte is still read exactly once
__global__ __launch_bounds__(256) void row64(const uint4* p, unsigned* out, long long n16) {
  unsigned acc = 0;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = lane >> 2, piece = lane & 3;
  // block of 16 rows x 256 B = 4 KiB = 256 chunks of 16 B per workgroup iteration
  for (long long b = blockIdx.x; b * 256 < n16; b += gridDim.x) {
    const long long i = b * 256 + row * 16 + wv * 4 + piece;
    if (i < n16) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  }
  sink(acc, out);
}
__global__ __launch_bounds__(256) void flush_write(uint4* p, long long n16) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) p[i] = make_uint4(1, 2, 3, 4);
}

int main() {
  const long long bytes = 1ll << 30;
  void *buf, *junk;
  unsigned* out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&junk, 2 * bytes));
  CHECK(hipMalloc(&out, 1 << 20));
  CHECK(hipMemset(buf, 1, bytes));
  const int grid = 4096;
  for (int rep = 0; rep < 3; ++rep) {
#define COLD() hipLaunchKernelGGL(flush_write, dim3(8192), dim3(256), 0, 0, (uint4*)junk, 2 * bytes / 16)
    COLD(); hipLaunchKernelGGL(full16, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, out, bytes / 16);
    COLD(); hipLaunchKernelGGL(full8, dim3(grid), dim3(256), 0, 0, (const uint2*)buf, out, bytes / 8);
    COLD(); hipLaunchKernelGGL(full4, dim3(grid), dim3(256), 0, 0, (const unsigned*)buf, out, bytes / 4);
    COLD(); hipLaunchKernelGGL(quad16, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, out, bytes / 16);
    COLD(); hipLaunchKernelGGL(quad16rep, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, out, bytes / 16);
    COLD(); hipLaunchKernelGGL(row64, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, out, bytes / 16);
  }
  CHECK(hipDeviceSynchronize());
  printf("bytes_per_kernel %lld\n", bytes);
  return 0;
}
