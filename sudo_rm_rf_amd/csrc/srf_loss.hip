// Training loss of the reference runners, forward and backward, on the device:
//     l = clamp( PITLossWrapper(PairwiseNegSDR("sisdr"), pit_from='pw_mtx')(est, tgt), -30, +30 )
//         experiments/run_improved_sudormrf.py:63-66,169-171
//     PairwiseNegSDR.forward (zero-mean, SI-SDR, 10 log10)      losses/sisdr.py:426-458
//     PITLossWrapper.forward / find_best_perm                   losses/sisdr.py:254-311 / 342-387
// Everything the loss needs from the waveforms is a handful of inner products per example:
//     sum e_i, sum e_i^2, sum t_j, sum t_j^2, sum e_i t_j          (4S + S^2 numbers, fp64)
// so the forward is ONE streaming pass over est/tgt (srf_pit_stats_kernel) plus a tiny per-example
// finalize (pairwise matrix, best permutation in itertools order with first-minimum tie-breaking like
// torch.argmin, batch mean, clamp), and the gradient w.r.t. the estimates is a two-term combination
//     dl/d e_i = A_i (e_i - mean e_i) + B_i (t_m(i) - mean t_m(i))
// of the estimate and its matched target (second streaming pass, srf_pit_grad_kernel).  The reference
// materialises [B,S,S,T] broadcast tensors for the same quantities (~10 passes, autograd doubling them).
#include "srf_common.h"

#define SRF_LOSS_MAX_SRC 4      // the streaming statistics kernels with everything in registers
#define SRF_LOSS_MAX_SRC_ANY 9  // the reference accepts n_src < 10 (losses/sisdr.py:275, find_best_perm :342-387): 5..9 sources run on
                                // srf_pit_stats_any_kernel (one estimate per block: the targets are re-read per estimate) and the
                                // finalize kernels' MAXS = 9 instantiation (S! permutations per example: 362 880 for 9 -- slow,
                                // as is the reference's [batch, S!, S] gather)

__global__ __launch_bounds__(256) void srf_pit_stats_kernel(const float* __restrict__ est,
                                                            const float* __restrict__ tgt, double* __restrict__ work,
                                                            int S, int T, int per_block) {
  __shared__ double red[4][SRF_LOSS_MAX_SRC * (4 + SRF_LOSS_MAX_SRC)];
  const long b = blockIdx.y;
  const int beg = blockIdx.x * per_block, end = min(beg + per_block, T);
  const int nstat = 4 * S + S * S;
  double acc[SRF_LOSS_MAX_SRC * (4 + SRF_LOSS_MAX_SRC)];
#pragma unroll
  for (int k = 0; k < SRF_LOSS_MAX_SRC * (4 + SRF_LOSS_MAX_SRC); ++k) acc[k] = 0.0;
  const float* eb = est + b * (long)S * T;
  const float* tb = tgt + b * (long)S * T;
  for (int t = beg + threadIdx.x; t < end; t += 256) {
    float e[SRF_LOSS_MAX_SRC], g[SRF_LOSS_MAX_SRC];
#pragma unroll
    for (int i = 0; i < SRF_LOSS_MAX_SRC; ++i) {
      e[i] = i < S ? eb[(long)i * T + t] : 0.f;
      g[i] = i < S ? tb[(long)i * T + t] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SRF_LOSS_MAX_SRC; ++i) {
      if (i < S) {
        acc[i] += (double)e[i];
        acc[S + i] += (double)e[i] * (double)e[i];
        acc[2 * S + i] += (double)g[i];
        acc[3 * S + i] += (double)g[i] * (double)g[i];
#pragma unroll
        for (int j = 0; j < SRF_LOSS_MAX_SRC; ++j)
          if (j < S) acc[4 * S + i * S + j] += (double)e[i] * (double)g[j];
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < SRF_LOSS_MAX_SRC * (4 + SRF_LOSS_MAX_SRC); ++k) {
    if (k < nstat) {
      const double v = srf_wave_sum(acc[k]);
      if (lane == 0) red[wave][k] = v;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < nstat) {
    const int k = threadIdx.x;
    atomicAdd(work + b * nstat + k, (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]));
  }
}

// S > SRF_LOSS_MAX_SRC: block (x, i, b) accumulates estimate i's {sum e, sum e^2, sum e t_j for every j}; the blocks of i == 0 also the
// targets' {sum t_j, sum t_j^2}.  Same work layout as srf_pit_stats_kernel.
__global__ __launch_bounds__(256) void srf_pit_stats_any_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                                double* __restrict__ work, int S, int T, int per_block) {
  constexpr int MS = SRF_LOSS_MAX_SRC_ANY, NA = 2 + 3 * MS;
  __shared__ double red[4][NA];
  const long b = blockIdx.z;
  const int i = blockIdx.y;
  const int beg = blockIdx.x * per_block, end = min(beg + per_block, T);
  double acc[NA];   // [0] sum e, [1] sum e^2, [2 + j] sum e t_j, [2 + MS + j] sum t_j, [2 + 2 MS + j] sum t_j^2
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  const float* eb = est + (b * S + i) * (long)T;
  const float* tb = tgt + b * (long)S * T;
  for (int t = beg + threadIdx.x; t < end; t += 256) {
    const double e = (double)eb[t];
    acc[0] += e;
    acc[1] += e * e;
#pragma unroll
    for (int j = 0; j < MS; ++j) {
      if (j < S) {
        const double g = (double)tb[(long)j * T + t];
        acc[2 + j] += e * g;
        if (i == 0) {
          acc[2 + MS + j] += g;
          acc[2 + 2 * MS + j] += g * g;
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const double v = srf_wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < NA) {
    const int k = threadIdx.x;
    const double v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    const int nstat = 4 * S + S * S;
    double* w = work + b * nstat;
    int dst = -1;
    if (k == 0) dst = i;
    else if (k == 1) dst = S + i;
    else if (k < 2 + MS) dst = (k - 2 < S) ? 4 * S + i * S + (k - 2) : -1;
    else if (k < 2 + 2 * MS) dst = (i == 0 && k - 2 - MS < S) ? 2 * S + (k - 2 - MS) : -1;
    else dst = (i == 0 && k - 2 - 2 * MS < S) ? 3 * S + (k - 2 - 2 * MS) : -1;
    if (dst >= 0) atomicAdd(w + dst, v);
  }
}

struct PitOut {
  float* pw;      // [Bt][S][S] (est, tgt) or null
  int* match;     // [Bt][S]: estimate matched with target j
  float* coef;    // [Bt][S][4]: per estimate i {A, B, mean e_i, mean of its matched target}
  int* tmatch;    // [Bt][S]: target matched with estimate i
  float* loss;    // {clamped batch mean, raw batch mean}
};

// sdr_type: 0 = "sisdr", 1 = "sdsdr", 2 = "snr" (losses/sisdr.py:440-455); zero_mean / take_log: the constructor flags.
// With P = |projection|^2 and N = |noise|^2 + 1e-8 as functions of (ee = <e,e>, d = <e,t>) for a fixed target:
//   sisdr: P = d^2 tt/tau^2, N0 = ee - 2 d^2/tau + d^2 tt/tau^2;  sdsdr: same P, N0 = ee - 2 d + tt;  snr: P = tt, N0 as sdsdr
//   d l / d e = 2 l_N (e - mean e) + (l_P P_d + l_N N_d) (t - mean t)       (means 0 without zero_mean)
template <int MAXS>
__global__ __launch_bounds__(256) void srf_pit_finalize_kernel(const double* __restrict__ work, PitOut o, int Bt,
                                                               int S, int T, float clamp, int sdr_type, int zero_mean,
                                                               int take_log) {
  __shared__ double red[4];
  const int nstat = 4 * S + S * S;
  double mysum = 0.0;
  for (int b = threadIdx.x; b < Bt; b += 256) {
    const double* w = work + (long)b * nstat;
    double pw[MAXS][MAXS], cA[MAXS][MAXS],
        cB[MAXS][MAXS];
    const double dT = (double)T;
    const double zm = zero_mean ? 1.0 : 0.0;
    for (int i = 0; i < S; ++i) {
      const double me = zm * w[i] / dT, ee = w[S + i] - dT * me * me;
      for (int j = 0; j < S; ++j) {
        const double mt = zm * w[2 * S + j] / dT;
        const double tau0 = w[3 * S + j] - dT * mt * mt, tau = tau0 + 1e-8;
        const double d = w[4 * S + i * S + j] - dT * me * mt;
        const double al = d / tau;
        double P, N, P_d, N_d;
        if (sdr_type == 2) {
          P = tau0;
          P_d = 0.0;
        } else {
          P = al * al * tau0;
          P_d = 2.0 * al * tau0 / tau;
        }
        if (sdr_type == 0) {
          N = ee - 2.0 * al * d + al * al * tau0 + 1e-8;
          N_d = -2.0 * al * (2.0 - tau0 / tau);
        } else {
          N = ee - 2.0 * d + tau0 + 1e-8;
          N_d = -2.0;
        }
        const double sdr = P / N;
        pw[i][j] = take_log ? -10.0 * log10(sdr + 1e-8) : -sdr;
        // d pw / d e_i = cA * (e_i - me) + cB * (t_j - mt)
        const double k = take_log ? -(10.0 / log(10.0)) / (sdr + 1e-8) : -1.0;
        cA[i][j] = k * (-2.0 * P / (N * N));
        cB[i][j] = k * (P_d / N - P * N_d / (N * N));
        if (o.pw) o.pw[((long)b * S + i) * S + j] = (float)pw[i][j];
      }
    }
    // permutations of (0..S-1) in lexicographic (= itertools.permutations) order; perm[j] = estimate for target j
    int perm[MAXS], best[MAXS];
    for (int j = 0; j < S; ++j) perm[j] = best[j] = j;
    double best_loss = 0.0;
    bool first = true;
    for (;;) {
      double l = 0.0;
      for (int j = 0; j < S; ++j) l += pw[perm[j]][j];
      l /= (double)S;
      if (first || l < best_loss) {   // strict: the first minimum wins, like torch.argmin
        best_loss = l;
        for (int j = 0; j < S; ++j) best[j] = perm[j];
        first = false;
      }
      // next lexicographic permutation
      int p = S - 2;
      while (p >= 0 && perm[p] > perm[p + 1]) --p;
      if (p < 0) break;
      int q = S - 1;
      while (perm[q] < perm[p]) --q;
      int tmp = perm[p];
      perm[p] = perm[q];
      perm[q] = tmp;
      for (int lo = p + 1, hi = S - 1; lo < hi; ++lo, --hi) {
        tmp = perm[lo];
        perm[lo] = perm[hi];
        perm[hi] = tmp;
      }
    }
    mysum += best_loss;
    const double scale = 1.0 / ((double)S * (double)Bt);   // mean over the matched pairs, then over the batch
    for (int j = 0; j < S; ++j) {
      const int i = best[j];
      o.match[(long)b * S + j] = i;
      o.tmatch[(long)b * S + i] = j;
      float* c = o.coef + ((long)b * S + i) * 4;
      c[0] = (float)(cA[i][j] * scale);
      c[1] = (float)(cB[i][j] * scale);
      c[2] = (float)(zm * w[i] / dT);
      c[3] = (float)(zm * w[2 * S + j] / dT);
    }
  }
  mysum = srf_wave_sum(mysum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mysum;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double raw = ((red[0] + red[1]) + (red[2] + red[3])) / (double)Bt;
    double cl = raw;
    if (clamp > 0.f) cl = cl < -(double)clamp ? -(double)clamp : (cl > (double)clamp ? (double)clamp : cl);
    o.loss[0] = (float)cl;
    o.loss[1] = (float)raw;
  }
}

__global__ __launch_bounds__(256) void srf_pit_grad_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                           const float* __restrict__ coef, const int* __restrict__ tmatch,
                                                           const float* __restrict__ loss, float* __restrict__ grad,
                                                           int S, int T, float clamp, const float* __restrict__ upstream) {
  const long b = blockIdx.z;
  const int i = blockIdx.y;
  const float raw = loss[1];
  // d clamp(raw) / d raw: 1 strictly inside the interval, 0 outside (torch.clamp passes the gradient on the
  // boundary as well; a loss of exactly +-30.0 is not worth a special case)
  const float gate = (clamp > 0.f && (raw < -clamp || raw > clamp)) ? 0.f : (upstream ? upstream[0] : 1.f);
  const float* c = coef + (b * S + i) * 4;
  const float A = c[0] * gate, B = c[1] * gate, me = c[2], mt = c[3];
  const int j = tmatch[b * S + i];
  const float* e = est + (b * S + i) * (long)T;
  const float* t = tgt + (b * S + j) * (long)T;
  float* g = grad + (b * S + i) * (long)T;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int x = blockIdx.x * 1024 + u * 256 + threadIdx.x;
    if (x < T) g[x] = fmaf(A, e[x] - me, B * (t[x] - mt));
  }
}

extern "C" size_t srf_pit_sisdr_work_bytes(int Bt, int S) {
  if (Bt <= 0 || S <= 0 || S > SRF_LOSS_MAX_SRC_ANY) return 0;
  // fp64 statistics | coef [Bt][S][4] f32 | match, tmatch [Bt][S] i32
  return (size_t)Bt * ((4 * S + S * S) * sizeof(double) + S * 4 * sizeof(float) + 2 * S * sizeof(int));
}

static void pit_carve(void* work, int Bt, int S, double** stats, float** coef, int** match, int** tmatch) {
  *stats = reinterpret_cast<double*>(work);
  *coef = reinterpret_cast<float*>(*stats + (size_t)Bt * (4 * S + S * S));
  *match = reinterpret_cast<int*>(*coef + (size_t)Bt * S * 4);
  *tmatch = *match + (size_t)Bt * S;
}

extern "C" int srf_pit_sdr_forward(const float* est, const float* tgt, int Bt, int S, int T, float clamp, int sdr_type,
                                   int zero_mean, int take_log, void* work, float* pw, float* loss, void* stream);

extern "C" int srf_pit_sisdr_forward(const float* est, const float* tgt, int Bt, int S, int T, float clamp,
                                     void* work, float* pw, float* loss, void* stream) {
  return srf_pit_sdr_forward(est, tgt, Bt, S, T, clamp, 0, 1, 1, work, pw, loss, stream);
}

extern "C" int srf_pit_sdr_forward(const float* est, const float* tgt, int Bt, int S, int T, float clamp, int sdr_type,
                                   int zero_mean, int take_log, void* work, float* pw, float* loss, void* stream) {
  SRF_CHECK_ARG(sdr_type >= 0 && sdr_type <= 2, "srf_pit_sdr_forward: sdr_type %d (0 sisdr, 1 sdsdr, 2 snr)", sdr_type);
  SRF_CHECK_ARG(est && tgt && work && loss, "srf_pit_sisdr_forward: null pointer");
  SRF_CHECK_ARG(Bt > 0 && Bt <= 65535 && T > 0, "srf_pit_sisdr_forward: bad sizes");
  SRF_CHECK_ARG(S >= 1 && S <= SRF_LOSS_MAX_SRC_ANY, "srf_pit_sisdr_forward: %d sources unsupported (1..%d)", S,
                SRF_LOSS_MAX_SRC_ANY);
  hipStream_t st = (hipStream_t)stream;
  double* stats;
  PitOut o;
  pit_carve(work, Bt, S, &stats, &o.coef, &o.match, &o.tmatch);
  o.pw = pw;
  o.loss = loss;
  SRF_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * (size_t)Bt * (4 * S + S * S), st));
  const int per_block = 256 * 16;
  if (S <= SRF_LOSS_MAX_SRC) {
    dim3 grid((unsigned)((T + per_block - 1) / per_block), (unsigned)Bt);
    hipLaunchKernelGGL(srf_pit_stats_kernel, grid, dim3(256), 0, st, est, tgt, stats, S, T, per_block);
    SRF_CHECK_LAUNCH("pit_sisdr_stats", st);
    hipLaunchKernelGGL(srf_pit_finalize_kernel<SRF_LOSS_MAX_SRC>, dim3(1), dim3(256), 0, st, stats, o, Bt, S, T, clamp, sdr_type,
                       zero_mean ? 1 : 0, take_log ? 1 : 0);
  } else {
    dim3 grid((unsigned)((T + per_block - 1) / per_block), (unsigned)S, (unsigned)Bt);
    hipLaunchKernelGGL(srf_pit_stats_any_kernel, grid, dim3(256), 0, st, est, tgt, stats, S, T, per_block);
    SRF_CHECK_LAUNCH("pit_sisdr_stats", st);
    hipLaunchKernelGGL(srf_pit_finalize_kernel<SRF_LOSS_MAX_SRC_ANY>, dim3(1), dim3(256), 0, st, stats, o, Bt, S, T, clamp,
                       sdr_type, zero_mean ? 1 : 0, take_log ? 1 : 0);
  }
  SRF_CHECK_LAUNCH("pit_sisdr_finalize", st);
  return SRF_OK;
}

extern "C" int srf_pit_sisdr_match(const void* work, int Bt, int S, int* match_out, void* stream) {
  SRF_CHECK_ARG(work && match_out && Bt > 0 && S >= 1 && S <= SRF_LOSS_MAX_SRC_ANY, "srf_pit_sisdr_match: bad arguments");
  double* stats;
  float* coef;
  int *match, *tmatch;
  pit_carve(const_cast<void*>(work), Bt, S, &stats, &coef, &match, &tmatch);
  SRF_CHECK_HIP(hipMemcpyAsync(match_out, match, sizeof(int) * (size_t)Bt * S, hipMemcpyDeviceToDevice,
                               (hipStream_t)stream));
  return SRF_OK;
}

extern "C" int srf_pit_sisdr_backward(const float* est, const float* tgt, int Bt, int S, int T, float clamp,
                                      const void* work, const float* loss, const float* upstream, float* grad_est,
                                      void* stream) {
  SRF_CHECK_ARG(est && tgt && work && loss && grad_est, "srf_pit_sisdr_backward: null pointer");
  SRF_CHECK_ARG(Bt > 0 && Bt <= 65535 && T > 0 && S >= 1 && S <= SRF_LOSS_MAX_SRC_ANY, "srf_pit_sisdr_backward: bad sizes");
  double* stats;
  float* coef;
  int *match, *tmatch;
  pit_carve(const_cast<void*>(work), Bt, S, &stats, &coef, &match, &tmatch);
  dim3 grid((unsigned)((T + 1023) / 1024), (unsigned)S, (unsigned)Bt);
  hipLaunchKernelGGL(srf_pit_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, est, tgt, coef, tmatch, loss,
                     grad_est, S, T, clamp, upstream);
  SRF_CHECK_LAUNCH("pit_sisdr_grad", stream);
  return SRF_OK;
}

// =============================================================================================
// Validation metric of the runners: PermInvariantSISDR (losses/sisdr.py:66-196; used as
// PermInvariantSISDR(zero_mean=True, backward_loss=False, improvement=True, return_individual_results=True),
// experiments/run_improved_sudormrf.py:82-85,201-205).  Same inner products as the training loss plus the
// mixture's {sum m, sum m^2, sum m t_j}; per example: SI-SNR of every (estimate, target) pair with the class's own
// eps placement  s = <p,t>/(<t,t>+eps) t,  10 log10(<s,s> / (<p-s,p-s> + eps)),  the mean over sources for every
// permutation in itertools order, the maximum (first maximum wins, like torch.max), and the SI-SNR of the mixture
// against every target (the "improvement" baseline, which the class subtracts as ONE batch-and-source mean).
// =============================================================================================
template <int MAXS>
__global__ __launch_bounds__(256) void srf_mix_stats_kernel(const float* __restrict__ mix, const float* __restrict__ tgt,
                                                            double* __restrict__ work, int S, int T, int per_block) {
  __shared__ double red[4][2 + MAXS];
  const long b = blockIdx.y;
  const int beg = blockIdx.x * per_block, end = min(beg + per_block, T);
  double acc[2 + MAXS];
#pragma unroll
  for (int k = 0; k < 2 + MAXS; ++k) acc[k] = 0.0;
  const float* mb = mix + b * (long)T;
  const float* tb = tgt + b * (long)S * T;
  for (int t = beg + threadIdx.x; t < end; t += 256) {
    const double m = (double)mb[t];
    acc[0] += m;
    acc[1] += m * m;
#pragma unroll
    for (int j = 0; j < MAXS; ++j)
      if (j < S) acc[2 + j] += m * (double)tb[(long)j * T + t];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 2 + MAXS; ++k) {
    if (k < 2 + S) {
      const double v = srf_wave_sum(acc[k]);
      if (lane == 0) red[wave][k] = v;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 + S) {
    const int k = threadIdx.x;
    atomicAdd(work + b * (2 + S) + k, (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]));
  }
}

template <int MAXS>
__global__ __launch_bounds__(256) void srf_perm_inv_finalize_kernel(const double* __restrict__ work,
                                                                    const double* __restrict__ mwork, int Bt, int S, int T,
                                                                    int zero_mean, double eps, float* __restrict__ best_out,
                                                                    int* __restrict__ perm_out, float* __restrict__ base_out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= Bt) return;
  const int nstat = 4 * S + S * S;
  const double* w = work + (long)b * nstat;
  const double dT = (double)T, zm = zero_mean ? 1.0 : 0.0;
  auto sisnr = [&](double pp, double tt, double pt) {   // inner products of the (centred) signals
    const double al = pt / (tt + eps);
    const double ss = al * al * tt;
    double ee = pp - 2.0 * al * pt + al * al * tt;
    ee = ee < 0.0 ? 0.0 : ee;
    return 10.0 * log10(ss / (ee + eps));
  };
  double sn[MAXS][MAXS];
  for (int i = 0; i < S; ++i) {
    const double me = zm * w[i] / dT, pp = w[S + i] - dT * me * me;
    for (int j = 0; j < S; ++j) {
      const double mt = zm * w[2 * S + j] / dT;
      sn[i][j] = sisnr(pp, w[3 * S + j] - dT * mt * mt, w[4 * S + i * S + j] - dT * me * mt);
    }
  }
  // permutations in itertools order; permuted_pr[:, j] = pr[:, perm[j]] is scored against target j
  int perm[MAXS];
  for (int j = 0; j < S; ++j) perm[j] = j;
  double best = 0.0;
  int best_idx = 0, idx = 0;
  for (;;) {
    double l = 0.0;
    for (int j = 0; j < S; ++j) l += sn[perm[j]][j];
    l /= (double)S;
    if (idx == 0 || l > best) {
      best = l;
      best_idx = idx;
    }
    ++idx;
    int p = S - 2;
    while (p >= 0 && perm[p] > perm[p + 1]) --p;
    if (p < 0) break;
    int q = S - 1;
    while (perm[q] < perm[p]) --q;
    int tmp = perm[p];
    perm[p] = perm[q];
    perm[q] = tmp;
    for (int lo = p + 1, hi = S - 1; lo < hi; ++lo, --hi) {
      tmp = perm[lo];
      perm[lo] = perm[hi];
      perm[hi] = tmp;
    }
  }
  best_out[b] = (float)best;
  perm_out[b] = best_idx;
  if (base_out && mwork) {
    const double* mw = mwork + (long)b * (2 + S);
    const double mm_ = zm * mw[0] / dT, pp = mw[1] - dT * mm_ * mm_;
    for (int j = 0; j < S; ++j) {
      const double mt = zm * w[2 * S + j] / dT;
      base_out[(long)b * S + j] = (float)sisnr(pp, w[3 * S + j] - dT * mt * mt, mw[2 + j] - dT * mm_ * mt);
    }
  }
}

extern "C" size_t srf_perm_inv_sisdr_work_bytes(int Bt, int S) {
  if (Bt <= 0 || S <= 0 || S > SRF_LOSS_MAX_SRC_ANY) return 0;
  return (size_t)Bt * ((4 * S + S * S) + (2 + S)) * sizeof(double);
}

extern "C" int srf_perm_inv_sisdr(const float* pr, const float* tgt, const float* mix, int Bt, int S, int T, int zero_mean,
                                  double eps, void* work, float* best, int* best_perm, float* base, void* stream) {
  SRF_CHECK_ARG(pr && tgt && work && best && best_perm, "srf_perm_inv_sisdr: null pointer");
  SRF_CHECK_ARG(Bt > 0 && Bt <= 65535 && T > 0, "srf_perm_inv_sisdr: bad sizes");
  SRF_CHECK_ARG(S >= 1 && S <= SRF_LOSS_MAX_SRC_ANY, "srf_perm_inv_sisdr: %d sources unsupported (1..%d)", S, SRF_LOSS_MAX_SRC_ANY);
  SRF_CHECK_ARG(!base || mix, "srf_perm_inv_sisdr: the improvement baseline needs the mixtures");
  hipStream_t st = (hipStream_t)stream;
  double* stats = reinterpret_cast<double*>(work);
  double* mstats = stats + (size_t)Bt * (4 * S + S * S);
  SRF_CHECK_HIP(hipMemsetAsync(work, 0, srf_perm_inv_sisdr_work_bytes(Bt, S), st));
  const int per_block = 256 * 16;
  dim3 grid((unsigned)((T + per_block - 1) / per_block), (unsigned)Bt);
  if (S <= SRF_LOSS_MAX_SRC) {
    hipLaunchKernelGGL(srf_pit_stats_kernel, grid, dim3(256), 0, st, pr, tgt, stats, S, T, per_block);
  } else {
    dim3 grid3(grid.x, (unsigned)S, (unsigned)Bt);
    hipLaunchKernelGGL(srf_pit_stats_any_kernel, grid3, dim3(256), 0, st, pr, tgt, stats, S, T, per_block);
  }
  if (mix && base) {
    if (S <= SRF_LOSS_MAX_SRC) hipLaunchKernelGGL(srf_mix_stats_kernel<SRF_LOSS_MAX_SRC>, grid, dim3(256), 0, st, mix, tgt, mstats, S, T, per_block);
    else hipLaunchKernelGGL(srf_mix_stats_kernel<SRF_LOSS_MAX_SRC_ANY>, grid, dim3(256), 0, st, mix, tgt, mstats, S, T, per_block);
  }
  if (S <= SRF_LOSS_MAX_SRC)
    hipLaunchKernelGGL(srf_perm_inv_finalize_kernel<SRF_LOSS_MAX_SRC>, dim3((unsigned)((Bt + 255) / 256)), dim3(256), 0, st, stats,
                       (mix && base) ? mstats : nullptr, Bt, S, T, zero_mean, eps, best, best_perm, base);
  else
    hipLaunchKernelGGL(srf_perm_inv_finalize_kernel<SRF_LOSS_MAX_SRC_ANY>, dim3((unsigned)((Bt + 255) / 256)), dim3(256), 0, st, stats,
                       (mix && base) ? mstats : nullptr, Bt, S, T, zero_mean, eps, best, best_perm, base);
  SRF_CHECK_LAUNCH("perm_inv_sisdr", st);
  return SRF_OK;
}
