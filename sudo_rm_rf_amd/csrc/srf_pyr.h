// Arguments of the register-resident pyramid kernels (srf_pyramid_reg.hip), filled by srf_pyramid() (srf_pyramid.hip).
#pragma once
#include "srf_common.h"

struct PyrRegArgs {
  const float* y1;     // pass 1 input
  float* d0;           // unused by the register kernels (the LDS kernels of srf_pyramid.hip stage level 0 here)
  float* merged;       // pass 2 output (may alias y1)
  SrfNormDev in_norm;  // proj_1x1 GlobLN (+PReLU)
  const float* in_mr;  // [groups][2] pre-finalised {mean, rstd} of in_norm (non-persistent pass 1 only)
  double in_inv_count; // 1 / (C * L)
  const float* w[SRF_MAX_DEPTH];
  const float* bias[SRF_MAX_DEPTH];
  const float* gamma[SRF_MAX_DEPTH];
  const float* beta[SRF_MAX_DEPTH];
  const float* lvl;    // [groups][D][2] {mean, rstd} per level (pass 2)
  double* mom;         // [rows][D][5] (pass 1; zeroed by the host)
  double* out_sums;    // merged statistics (pass 2)
  int rows;            // groups * C
  int rpw;             // rows per (persistent) wavefront
  int C, L, D, tiles, own;   // own = own chunks per tile
  float* lv_out[SRF_MAX_DEPTH];   // training forward (SAVE): the raw (pre-norm) conv output of every level, or null
  int save = 0;        // SAVE requested (lv_out[1 .. D-1] set; lv_out[0] may be null: level 0 is not kept -- round 6, the backward
                       // re-computes d_0 from y1)
};

bool srf_pyramid_reg_supported(int L, int D);
int srf_pyramid_reg_launch(PyrRegArgs a, bool moments, long rows, hipStream_t st);
