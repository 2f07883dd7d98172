// Shared declarations of the quad-channel W-Toeplitz convolution kernels: conv3d_q4.hip (16-bit storage) and conv3d_q4s.hip (fp32
// storage, two-term fp16 operands on the matrix cores).
#pragma once
#include "common.h"
#include "conv_pack.h"
#include "fanin.h"
#include "../../include/xlstm_hved.h"

typedef h16x8 frag8;
typedef f32x4_t f32x4;

struct ConvQ4 {
  xh_conv_desc d;
  xh_conv_ptrs p;
  int Cin_g, Cout_g, ci4;       // channels per group, input-channel quads per group
  int tilesW, tilesH, tilesD;
  unsigned mW, mH, mQ, mG;      // reciprocals (udiv_magic) of tilesW, tilesH, oq_g, gpp
  int oq_g, gpp;                // output-channel quads per group, groups per weight pointer
  int td;                       // output planes per workgroup (8 | 4 | 2)
  int dw;                       // depthwise conv presented as groups of 4 channels with diagonal weights (plan, pack only)
  float act_slope;              // effective epilogue slope: 1 = identity, 0 = ReLU, else LeakyReLU
  double fin_inv;               // 1 / fin_count
  unsigned char* fan;           // statistics fan-in block of this launch (fanin.h), or nullptr: direct atomics
  int abl;
};
extern int g_mfma_abl;
extern int g_q4_maxc, g_q4_wgs;

namespace {
constexpr int TW = 32, TH = 8, IH = TH + 2;
constexpr int PITCH = 288;                  // 36 voxels (ow0 - 2 .. ow0 + 33) x 8 bytes
constexpr int PLANE = IH * PITCH;
constexpr int Q4_MAXC = 48;                 // most channels per group the kernel can be asked to take
// TD = output planes per workgroup: 8 (28.8 KB tile, 100 staged rows = two 8-voxel items per thread) for the volumes that fill
// the chip; 4 / 2 for the 64^3 / 32^3 launches, which are chains of load -> transform -> matrix phase per input quad on a few
// dozen workgroups: a quarter of the planes per workgroup = four times the workgroups and a shorter chain each (the halo
// planes are re-read from L2)
constexpr int q4_tile_bytes(int td) { return (td + 2) * PLANE; }
}

// Buffer descriptor of a wave-uniform base pointer, 2 GiB window: accesses at a 32-bit lane offset >= 0x80000000 are out of range,
// i.e. a load returns 0 and a store is dropped -- predication without a branch (a branch around a vector-memory instruction makes
// hipcc's s_waitcnt bookkeeping fall back to "wait for everything", which serialises loads that were requested ahead of their use)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t q4_window(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x80000000, 0x00020000);
}
constexpr unsigned Q4_OOB = 0xFFFFFFF0u;

