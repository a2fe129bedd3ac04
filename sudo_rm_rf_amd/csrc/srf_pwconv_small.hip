// K2s -- pointwise Conv1d for SMALL channel counts (GroupComm's per-group proj_1x1 / res_conv,
// groupcomm_sudormrf_v2.py:167-170,192-197: out_channels/G -> in_channels/G and back, e.g. 16 <-> 32
// with the batch folded to B*G).  At these shapes the op is a pure HBM stream (2*Cin*Cout flops against
// 4*(Cin+Cout) bytes per time step, < 11 flop/B): no LDS, no MFMA.  One lane owns VEC consecutive time
// steps of one folded-batch row, holds all Cin inputs in registers (every x element is loaded exactly
// once, as 8/16-byte coalesced vectors), and produces the Cout outputs 8 at a time with the weights as
// wave-uniform scalar operands.  Prologue (GlobLN / PReLU on load) and epilogue (bias, residual,
// GlobLN {sum,sumsq}) match the GEMM kernels (srf_pwconv.hip); the ReLU*encoder epilogue is not needed
// at these shapes and stays with the other kernels.
#include "srf_pw.h"

template <int VEC>
struct SrfVec;
template <>
struct SrfVec<2> {
  typedef float type __attribute__((ext_vector_type(2)));
};
template <>
struct SrfVec<4> {
  typedef float type __attribute__((ext_vector_type(4)));
};

template <int CIN, int COUT, int VEC>
__global__ __launch_bounds__(256) void srf_pw_small_kernel(
    PwArgs a, int wavesPerRow, int totalWaves,
    // weights / bias / affine again as noalias kernel arguments: only then are they provably not
    // clobbered by the stores to y and fetched with scalar loads (SGPR operands) instead of VMEM
    const float* __restrict__ wgt, const float* __restrict__ bias, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ pre_gamma, const float* __restrict__ pre_beta) {
  typedef typename SrfVec<VEC>::type vecf;
  constexpr int MC = 4;   // outputs per pass: MC*CIN weights must fit the SGPR file
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (w >= totalWaves) return;
  const long b = w / wavesPerRow;
  const int wr = w - (int)b * wavesPerRow;
  const int l = (wr * 64 + lane) * VEC;
  const bool valid = l < a.L;          // L % VEC == 0: a vector is entirely inside or outside the row
  const int lc = valid ? l : 0;        // clamped: loads stay unconditional

  const bool has_norm = a.nrm.sums != nullptr;
  const bool act = a.nrm.prelu != nullptr;
  float mean = 0.f, rstd = 1.f;
  if (has_norm) srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
  const float slope = act ? a.nrm.prelu[0] : 1.f;

  const float* xb = a.x + (size_t)b * CIN * a.L + lc;
  vecf x[CIN];
  // (round 4: NON-TEMPORAL loads for the three streamed inputs -- every byte is read exactly once; 105.7 -> 95.7 us per launch at
  // cfg 3, same box; non-temporal stores on top measured worse for the forward: the next kernel reads what is written here)
#pragma unroll
  for (int k = 0; k < CIN; ++k) x[k] = __builtin_nontemporal_load(reinterpret_cast<const vecf*>(xb + (size_t)k * a.L));
  if (a.pre_q) {
    // u = x + GlobLN(q) on load -- exactly srf_gln_apply_kernel<true>'s arithmetic (fmaf(q, gamma rstd, beta - mean gamma rstd),
    // then the add), so the fused forward is bitwise the unfused one -- and u goes out once, for the block's residual.
    // Saves the separate kernel's launch and one read of the [Bt, B, L] tensor per block.
    float pm, pr;
    srf_finalize_stats(a.pre_nrm.sums, b, a.pre_inv_count, pm, pr);
    const float* qb = a.pre_q + (size_t)b * CIN * a.L + lc;
    float* ub = a.pre_u + (size_t)b * CIN * a.L + lc;
#pragma unroll
    for (int k = 0; k < CIN; ++k) {
      const vecf qv = __builtin_nontemporal_load(reinterpret_cast<const vecf*>(qb + (size_t)k * a.L));
      const float sc = pre_gamma[k] * pr;
      const float sh = pre_beta[k] - pm * sc;
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[k][v] = x[k][v] + fmaf(qv[v], sc, sh);
      if (valid) *reinterpret_cast<vecf*>(ub + (size_t)k * a.L) = x[k];
    }
  }
  if (has_norm) {
#pragma unroll
    for (int k = 0; k < CIN; ++k) {
      const float sc = gamma[k] * rstd;
      const float sh = beta[k] - mean * sc;
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[k][v] = fmaf(x[k][v], sc, sh);
    }
  }
  if (act) {
#pragma unroll
    for (int k = 0; k < CIN; ++k)
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[k][v] = srf_prelu(x[k][v], slope);
  }

  float s = 0.f, q = 0.f;
  const size_t ybase = (size_t)b * COUT * a.L + lc;
  // not unrolled: a fully unrolled body lets the scheduler hoist all Cin*Cout scalar weight loads to
  // the top, where they spill out of the SGPR file into VGPRs (measured: 215-256 VGPRs, occupancy 1-2)
#pragma unroll 1
  for (int m0 = 0; m0 < COUT; m0 += MC) {
    vecf acc[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) acc[i] = (vecf)(0.f);
#pragma unroll
    for (int k = 0; k < CIN; ++k) {
#pragma unroll
      for (int i = 0; i < MC; ++i) {
        const float wv = wgt[(m0 + i) * CIN + k];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[i][v] = fmaf(wv, x[k][v], acc[i][v]);
      }
    }
#pragma unroll
    for (int i = 0; i < MC; ++i) {
      const size_t idx = ybase + (size_t)(m0 + i) * a.L;
      vecf o = acc[i] + bias[m0 + i];
      if (a.residual) o += __builtin_nontemporal_load(reinterpret_cast<const vecf*>(a.residual + idx));
      if (valid) {
        *reinterpret_cast<vecf*>(a.y + idx) = o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          s += o[v];
          q = fmaf(o[v], o[v], q);
        }
      }
    }
  }
  if (a.out_sums) {
    const double ds = srf_wave_sum((double)s), dq = srf_wave_sum((double)q);
    if (lane == 0) {
      double* dst = srf_stat_slot(a.out_sums, b, wr);
      atomicAdd(dst, ds);
      atomicAdd(dst + 1, dq);
    }
  }
}

bool srf_pw_small_supported(int Cin, int Cout, int L) {
  const bool cin_ok = Cin == 8 || Cin == 16 || Cin == 32;
  const bool cout_ok = Cout == 8 || Cout == 16 || Cout == 32 || Cout == 64;
  return cin_ok && cout_ok && (L % 4 == 0);
}

template <int CIN, int COUT>
static void srf_pw_small_go(const PwArgs& a, hipStream_t st) {
  constexpr int VEC = CIN <= 16 ? 4 : 2;
  const int wavesPerRow = (a.L / VEC + 63) / 64;
  const long total = (long)a.Bt * wavesPerRow;
  dim3 grid((unsigned)((total + 3) / 4)), block(256);
  hipLaunchKernelGGL((srf_pw_small_kernel<CIN, COUT, VEC>), grid, block, 0, st, a, wavesPerRow, (int)total,
                     a.w, a.bias, a.nrm.gamma, a.nrm.beta, a.pre_nrm.gamma, a.pre_nrm.beta);
}

template <int CIN>
static void srf_pw_small_cout(const PwArgs& a, hipStream_t st) {
  switch (a.Cout) {
    case 8: srf_pw_small_go<CIN, 8>(a, st); break;
    case 16: srf_pw_small_go<CIN, 16>(a, st); break;
    case 32: srf_pw_small_go<CIN, 32>(a, st); break;
    default: srf_pw_small_go<CIN, 64>(a, st); break;
  }
}

int srf_pw_small_launch(const PwArgs& a, hipStream_t st) {
  SRF_CHECK_ARG((long)a.Bt * ((a.L / 2 + 63) / 64) < (1L << 31), "srf_pw_conv: too many rows");
  switch (a.Cin) {
    case 8: srf_pw_small_cout<8>(a, st); break;
    case 16: srf_pw_small_cout<16>(a, st); break;
    default: srf_pw_small_cout<32>(a, st); break;
  }
  SRF_CHECK_LAUNCH("pw_conv_small", st);
  return SRF_OK;
}
