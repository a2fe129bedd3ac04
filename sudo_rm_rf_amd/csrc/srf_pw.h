// Shared pieces of the pointwise-conv GEMM kernels (srf_pwconv.hip, srf_pwconv_bf16x3.hip).
#pragma once
#include "srf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PwArgs {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  const float* residual;
  double* out_sums;
  const float* mul;
  SrfNormDev nrm;
  double inv_count;
  int Cin, Cout, L, Bt;
  int mul_channels;
  int epi_mask;
};


// XCD-aware tile numbering: hardware places block id on XCD id%8; give each XCD a contiguous run of
// virtual ids so the Cout/BM blocks that share one X tile hit the same L2 (bijective for any total).
__device__ __forceinline__ int srf_xcd_remap(int id, int total) {
  const int xcd = id & 7, slot = id >> 3;
  const int qn = total >> 3, rn = total & 7;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
}

// Epilogue of one 32x32 MFMA accumulator tile whose top-left element is (m_base, l_base).
// C/D layout of v_mfma_f32_32x32x*: col = lane&31 (time), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
__device__ __forceinline__ void srf_pw_epilogue_tile(const PwArgs& a, const f32x16& acc, long b, int m_base,
                                                     int l_base, int lane, float& s, float& q) {
  const int l = l_base + (lane & 31);
  const int kh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (m < a.Cout && l < a.L) {
      const size_t idx = ((size_t)b * a.Cout + m) * a.L + l;
      float v = acc[r] + a.bias[m];
      if (a.residual) v += a.residual[idx];
      if (a.epi_mask)
        v = fmaxf(v, 0.f) * a.mul[((size_t)b * a.mul_channels + (m % a.mul_channels)) * a.L + l];
      a.y[idx] = v;
      s += v;
      q = fmaf(v, v, q);
    }
  }
}
