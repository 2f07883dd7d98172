// Problem record of the quad-channel weight-gradient kernel (conv3d_wgrad_q4.hip), shared with the batching entry point.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/xlstm_hved.h"

struct WgQ4 {
  const void* xa; const void* xb; const void* dy;
  const float* pre_sc; const float* pre_sh;
  float* dw[XH_MAX_WPTR]; float* db[XH_MAX_WPTR];
  long long xa_bs, xb_bs, dy_bs;
  int N, Cin, Cout, groups, n_wptr, Ca, D, H, W;
  int Cin_g, Cout_g, ci4, pre;   // ci4 = input-channel quads per unit (1..3)
  float pre_slope;
  int tilesW, tilesH, dsegs, sd;
  int nchunk;                    // chunks of ci4 input quads a group's input channels are cut into
  int nq;                        // units of the problem: (output-channel quad, input chunk) pairs = Cout / 4 * nchunk
  int ntile;                     // spatial tiles of a unit (tilesW * tilesH * dsegs * N)
  int wpu;                       // workgroups per unit (each walks ntile / wpu tiles)
  int nb;                        // workgroups of the problem (nq * wpu)
  int abl;                       // ablation mask (microbenchmarks)
  int dwm;                       // depthwise problem run as groups of 4: only the diagonal of a 4 x 4 block is a gradient
  int wide;                      // tile shape: 0 = 4 rows x 32 voxels, 1 = 2 rows x 64 voxels (rows of 64 / 128 voxels: full-line loads)
  int bcast;                     // xh_conv_desc.bcast: the group's four input channels are one stored channel of xa (full-row kernel only)
  int full;                      // 1: planned for the full-row kernel (conv3d_wgrad_q5.hip: 8 rows x W tiles)
  int uqx, uqy;                  // full-row kernel: input / output channel quads a unit stages (1 | 2; 2 on rows of 64 voxels only)
};

constexpr int WQ_MULTI = 8;         // problems per launch (the table travels in the kernel arguments)
constexpr int Q5_MULTI = 12;        // the same for the full-row kernel (4 KB of kernel arguments)
bool xh_wgrad_q4_plan(const xh_conv_desc* d, const xh_conv_ptrs* p, float* const dw[XH_MAX_WPTR], float* const db[XH_MAX_WPTR], WgQ4* a);
void xh_wgrad_q4_launch(hipStream_t st, int fmt, const WgQ4* probs, int n);
// conv3d_wgrad_q5.hip: re-plans a planned problem for the full-row kernel (rows of 64 / 128 voxels, 16-bit storage); launch of up to
// Q5_MULTI such problems of one storage format (xh_wgrad_q4_launch forwards to it)
bool xh_wgrad_q5_replan(const xh_conv_desc* d, WgQ4* a);
void xh_wgrad_q5_launch(hipStream_t st, int fmt, const WgQ4* probs, int n);
// min-max workgroup plan of one full-row launch (wq[i] workgroups per unit of problem i); returns its planned duration in unit rounds
double xh_wgrad_q5_plan(const WgQ4* probs, int n, int budget, int* wq);
