// K6 -- transform-average-concatenate group communication (TAC.forward up to TAC_norm,
// reference: groupcomm_sudormrf_v2.py:356-377).  Rows of the reference's Linear layers are
// (batch, time, group); here one lane owns one (batch, time) column and walks the G groups, so
// every activation load/store is coalesced along time and every weight is a wave-uniform scalar.
//   z_g  = PReLU(Wi x_g + bi)                      Linear(n -> H) + PReLU          :365-366
//   zbar = mean_g z_g                                                               :369
//   q    = PReLU(Wm zbar + bm)                     Linear(H -> H) + PReLU          :372-373
//   o_g  = PReLU(Wo [z_g ; q] + bo)                Linear(2H -> n) + PReLU         :375-377
// The group mean needs all z_g before any o_g: z_g is recomputed in a second sweep instead of being
// spilled (n*H MACs are cheaper than 4*H bytes of traffic per group and column).  The q-half of Wo
// is group-independent and folded into a per-column vector r once.
// The output is the PRE-norm tensor; {sum,sumsq} per (batch, group) feed TAC_norm (GlobLN).
#include "srf_common.h"

struct TacArgs {
  const float* x;
  float* q;
  const float *wi, *bi, *ai, *wm, *bm, *am, *wo, *bo, *ao;
  double* out_sums;
  int G, L;
};

template <int NN>
__global__ __launch_bounds__(128) void srf_tac_kernel(TacArgs a) {
  constexpr int HH = 3 * NN;
  const int l = blockIdx.x * 128 + threadIdx.x;
  const long b = blockIdx.y;
  const bool valid = l < a.L;
  const int G = a.G, L = a.L;
  const float ai = a.ai[0], am = a.am[0], ao = a.ao[0];
  const float* xb = a.x + (size_t)b * G * NN * L;
  float* qb = a.q + (size_t)b * G * NN * L;

  float zbar[HH];
#pragma unroll
  for (int j = 0; j < HH; ++j) zbar[j] = 0.f;
  for (int g = 0; g < G; ++g) {
    float xg[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) xg[i] = valid ? xb[((size_t)g * NN + i) * L + l] : 0.f;
#pragma unroll
    for (int j = 0; j < HH; ++j) {
      float z = 0.f;
#pragma unroll
      for (int i = 0; i < NN; ++i) z = fmaf(a.wi[j * NN + i], xg[i], z);
      z = srf_prelu(z + a.bi[j], ai);
      zbar[j] += z;
    }
  }
  const float fG = (float)G;
#pragma unroll
  for (int j = 0; j < HH; ++j) zbar[j] = zbar[j] / fG;

  // r = bo + Wo[:, H:2H] * PReLU(Wm zbar + bm)
  float r[NN];
#pragma unroll
  for (int i = 0; i < NN; ++i) r[i] = 0.f;
  for (int j = 0; j < HH; ++j) {
    float qj = 0.f;
#pragma unroll
    for (int i = 0; i < HH; ++i) qj = fmaf(a.wm[j * HH + i], zbar[i], qj);
    qj = srf_prelu(qj + a.bm[j], am);
#pragma unroll
    for (int i = 0; i < NN; ++i) r[i] = fmaf(a.wo[i * 2 * HH + HH + j], qj, r[i]);
  }

  for (int g = 0; g < G; ++g) {
    float xg[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) xg[i] = valid ? xb[((size_t)g * NN + i) * L + l] : 0.f;
    float o[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) o[i] = 0.f;
    for (int j = 0; j < HH; ++j) {
      float z = 0.f;
#pragma unroll
      for (int i = 0; i < NN; ++i) z = fmaf(a.wi[j * NN + i], xg[i], z);
      z = srf_prelu(z + a.bi[j], ai);
#pragma unroll
      for (int i = 0; i < NN; ++i) o[i] = fmaf(a.wo[i * 2 * HH + j], z, o[i]);
    }
    float s = 0.f, qq = 0.f;
#pragma unroll
    for (int i = 0; i < NN; ++i) {
      const float v = srf_prelu((o[i] + r[i]) + a.bo[i], ao);
      if (valid) {
        qb[((size_t)g * NN + i) * L + l] = v;
        s += v;
        qq = fmaf(v, v, qq);
      }
    }
    if (a.out_sums) {
      const double ds = srf_wave_sum((double)s), dq = srf_wave_sum((double)qq);
      if ((threadIdx.x & 63) == 0) {
        double* dst = srf_stat_slot(a.out_sums, b * G + g, blockIdx.x * 2 + (threadIdx.x >> 6));
        atomicAdd(dst, ds);
        atomicAdd(dst + 1, dq);
      }
    }
  }
}

extern "C" int srf_tac(const float* x, float* q, const float* const* params, int Bt, int G, int n, int H,
                       int L, double* out_sums, void* stream) {
  SRF_CHECK_ARG(x && q && params, "srf_tac: null pointer");
  SRF_CHECK_ARG(Bt > 0 && G > 0 && n > 0 && L > 0 && Bt <= 65535, "srf_tac: bad sizes");
  SRF_CHECK_ARG(H == 3 * n, "srf_tac: hidden size must be 3*n (got n=%d H=%d)", n, H);
  for (int i = 0; i < 9; ++i) SRF_CHECK_ARG(params[i] != nullptr, "srf_tac: null parameter %d", i);
  TacArgs a;
  a.x = x;
  a.q = q;
  a.wi = params[0];
  a.bi = params[1];
  a.ai = params[2];
  a.wm = params[3];
  a.bm = params[4];
  a.am = params[5];
  a.wo = params[6];
  a.bo = params[7];
  a.ao = params[8];
  a.out_sums = out_sums;
  a.G = G;
  a.L = L;
  dim3 grid((L + 127) / 128, Bt), block(128);
  hipStream_t st = (hipStream_t)stream;
  switch (n) {
    case 2: hipLaunchKernelGGL(srf_tac_kernel<2>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(srf_tac_kernel<4>, grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL(srf_tac_kernel<8>, grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL(srf_tac_kernel<16>, grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL(srf_tac_kernel<32>, grid, block, 0, st, a); break;
    default:
      srf_set_error("srf_tac: channels per group n=%d unsupported (2,4,8,16,32)", n);
      return SRF_EINVAL;
  }
  SRF_CHECK_LAUNCH("tac", st);
  return SRF_OK;
}
