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
  // srf_pwconv_small.hip only: the conv's input is u = x + GlobLN(pre_q) (GroupComm's TAC norm + residual,
  // groupcomm_sudormrf_v2.py:381-384, folded into this conv's load); u is also written to pre_u (the block's residual)
  const float* pre_q = nullptr;
  float* pre_u = nullptr;
  SrfNormDev pre_nrm = {nullptr, nullptr, nullptr, nullptr};
  double pre_inv_count = 0.0;
};


// Two 1x1 convolutions back to back in one launch (srf_pwconv_x3f.hip): y = W1 f(x) + b1 (+ residual) with Cout = 256, then
// y2 = W2 y + b2 and the statistics of y2.  wpack1 / wpack2: the paired-block image of the two weights (srf_x3p_packed_bytes).
struct PwPairArgs {
  // conv 1
  const float* x;
  const float* residual;
  const float* bias1;
  float* y;
  SrfNormDev nrm;
  double inv_count;
  const char* wpack1;
  // conv 2
  const char* wpack2;
  const float* bias2;
  float* y2;
  double* out_sums2;
  int K1, C2, L, Bt, nLt, total;
};


// XCD-aware tile numbering: hardware places block id on XCD id%8; give each XCD a contiguous run of
// virtual ids so the Cout/BM blocks that share one X tile hit the same L2 (bijective for any total).
__device__ __forceinline__ int srf_xcd_remap(int id, int total) {
  const int xcd = id & 7, slot = id >> 3;
  const int qn = total >> 3, rn = total & 7;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
}

// Epilogue of a 32(row) x 64(time) strip held as two 32x32 MFMA accumulator tiles (accL | accR), staged
// through a wave-private LDS strip so that global traffic is row-contiguous 16-B per lane
// (4 rows x 256 B per wavefront instruction) instead of 4-B scattered stores:
//   C/D layout of v_mfma_f32_32x32x*: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
// bias / residual / ReLU*enc / {sum,sumsq} are applied on the float4 side.  `strip` = 32*SRF_EPI_PITCH
// floats, 16-B aligned, private to the calling wavefront (all 64 lanes must call).
constexpr int SRF_EPI_PITCH = 68;

__device__ __forceinline__ void srf_pw_epilogue_strip(const PwArgs& a, const f32x16& accL, const f32x16& accR,
                                                      float* strip, long b, int m_base, int l_base, int lane,
                                                      float& s, float& q) {
  const int col = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
    strip[row * SRF_EPI_PITCH + col] = accL[r];
    strip[row * SRF_EPI_PITCH + 32 + col] = accR[r];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int c4 = (lane & 15) * 4;
  const int l = l_base + c4;
  const bool l_ok = l < a.L;  // L % 4 == 0 -> the whole float4 is in range
  const size_t lc = l_ok ? l : 0;
  const int mulC = a.mul_channels;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = i * 4 + (lane >> 4);
    const int m = m_base + row;
    const bool ok = l_ok && m < a.Cout;
    const int mc = m < a.Cout ? m : 0;  // clamped: loads stay unconditional (no branch, no vmcnt drain)
    float4 v = *reinterpret_cast<const float4*>(strip + row * SRF_EPI_PITCH + c4);
    const float bs = a.bias[mc];
    v.x += bs;
    v.y += bs;
    v.z += bs;
    v.w += bs;
    const size_t idx = ((size_t)b * a.Cout + mc) * a.L + lc;
    if (a.residual) {
      const float4 rv = *reinterpret_cast<const float4*>(a.residual + idx);
      v.x += rv.x;
      v.y += rv.y;
      v.z += rv.z;
      v.w += rv.w;
    }
    if (a.epi_mask & 1) {
      const float4 e = *reinterpret_cast<const float4*>(a.mul + ((size_t)b * mulC + (mc % mulC)) * a.L + lc);
      v.x = fmaxf(v.x, 0.f) * e.x;
      v.y = fmaxf(v.y, 0.f) * e.y;
      v.z = fmaxf(v.z, 0.f) * e.z;
      v.w = fmaxf(v.w, 0.f) * e.w;
    }
    if (ok) {
      *reinterpret_cast<float4*>(a.y + idx) = v;
      s += (v.x + v.y) + (v.z + v.w);
      q = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, q))));
    }
  }
}

// ---- 32 x 32 variant (one accumulator tile per call), shared by the persistent kernels ----
constexpr int SRF_EPI_PITCH_H = 36;

// one 32x32 accumulator tile -> rows x 128 B float4 stores (see srf_pw_epilogue_strip)
__device__ __forceinline__ void srf_pw_epilogue_half(const PwArgs& a, const f32x16& acc, float* strip, long b,
                                                     int m_base, int l_base, int lane, float& s, float& q) {
  const int col = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
    strip[row * SRF_EPI_PITCH_H + col] = acc[r];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int c4 = (lane & 7) * 4;
  const int l = l_base + c4;
  const bool l_ok = l < a.L;
  const size_t lc = l_ok ? l : 0;
  const int mulC = a.mul_channels;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + (lane >> 3);
    const int m = m_base + row;
    const bool ok = l_ok && m < a.Cout;
    const int mc = m < a.Cout ? m : 0;
    float4 v = *reinterpret_cast<const float4*>(strip + row * SRF_EPI_PITCH_H + c4);
    const float bs = a.bias[mc];
    v.x += bs;
    v.y += bs;
    v.z += bs;
    v.w += bs;
    const size_t idx = ((size_t)b * a.Cout + mc) * a.L + lc;
    if (a.residual) {
      const float4 rv = *reinterpret_cast<const float4*>(a.residual + idx);
      v.x += rv.x;
      v.y += rv.y;
      v.z += rv.z;
      v.w += rv.w;
    }
    if (a.epi_mask & 1) {
      const float4 e = *reinterpret_cast<const float4*>(a.mul + ((size_t)b * mulC + (mc % mulC)) * a.L + lc);
      v.x = fmaxf(v.x, 0.f) * e.x;
      v.y = fmaxf(v.y, 0.f) * e.y;
      v.z = fmaxf(v.z, 0.f) * e.z;
      v.w = fmaxf(v.w, 0.f) * e.w;
    }
    if (ok) {
      *reinterpret_cast<float4*>(a.y + idx) = v;
      s += (v.x + v.y) + (v.z + v.w);
      q = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, q))));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

