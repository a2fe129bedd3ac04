// On-GPU online remix augmentation of the reference's training loop (SURVEY.md §8f rank 3;
// experiments/run_improved_sudormrf.py:150-164, normalize_tensor_wav :127-131):
//     energies = sum(clean^2, -1);  random_wavs = clean[:, randperm(S)]
//     new_s_j  = random_wavs[randperm(B), j] * sqrt(energies[:, j] / sum(new_s_j^2, -1))         j = 0..S-1
//     mixture  = normalize(sum_j new_s_j);  clean[:, j] = normalize(new_s_j)     normalize = (x - mean)/(std + 1e-8)
// = ~25 ATen kernels and 10 passes over [B,S,T] per step in the reference.  Here: one pass for the row statistics
// of `clean` (everything about a re-scaled copy of a row follows from its {sum, sum of squares}: energy, mean,
// unbiased std), one pass for the mixture's statistics (the only quantity with cross terms), one pass that
// writes the normalised mixture and sources.  The permutations stay on the host side (torch.randperm, same RNG
// consumption as the runner) and arrive as index tensors.
#include "srf_common.h"

#define SRF_AUG_MAX_SRC 4

__global__ __launch_bounds__(256) void srf_row_stats_kernel(const float* __restrict__ x, double* __restrict__ sums,
                                                            int T) {
  __shared__ double red[8];
  const long r = blockIdx.x;
  const float* xr = x + r * (long)T;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < T; i += 256) {
    const double v = (double)xr[i];
    s += v;
    q += v * v;
  }
  s = srf_wave_sum(s);
  q = srf_wave_sum(q);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = s;
    red[4 + (threadIdx.x >> 6)] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[2 * r] = (red[0] + red[1]) + (red[2] + red[3]);
    sums[2 * r + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

struct RemixArgs {
  const float* clean;     // [B,S,T]
  const double* rs;       // [B*S][2] row statistics of clean
  const int* src_b;       // [S][B]: batch index the new source j of example b is taken from
  const int* src_s;       // [S]: source index it is taken from (the randperm over sources)
  double* ms;             // [B][2] mixture statistics
  float* mix;             // [B,T]
  float* out;             // [B,S,T]
  int B, S, T;
  float eps;
};

__device__ __forceinline__ void remix_rows(const RemixArgs& a, int b, const float* (&row)[SRF_AUG_MAX_SRC],
                                           float (&scale)[SRF_AUG_MAX_SRC], long (&src)[SRF_AUG_MAX_SRC]) {
#pragma unroll
  for (int j = 0; j < SRF_AUG_MAX_SRC; ++j) {
    if (j < a.S) {
      src[j] = (long)a.src_b[j * a.B + b] * a.S + a.src_s[j];
      row[j] = a.clean + src[j] * (long)a.T;
      // sqrt(energies[b, j] / sum(new_s^2)): target energy = ORIGINAL row (b, j), current = the gathered row's
      scale[j] = (float)sqrt(a.rs[2 * ((long)b * a.S + j) + 1] / a.rs[2 * src[j] + 1]);
    } else {
      src[j] = 0;
      row[j] = a.clean;
      scale[j] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void srf_remix_stats_kernel(RemixArgs a) {
  __shared__ double red[8];
  const int b = blockIdx.x;
  const float* row[SRF_AUG_MAX_SRC];
  float scale[SRF_AUG_MAX_SRC];
  long src[SRF_AUG_MAX_SRC];
  remix_rows(a, b, row, scale, src);
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < a.T; i += 256) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < SRF_AUG_MAX_SRC; ++j)
      if (j < a.S) m += row[j][i] * scale[j];
    s += (double)m;
    q += (double)m * (double)m;
  }
  s = srf_wave_sum(s);
  q = srf_wave_sum(q);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = s;
    red[4 + (threadIdx.x >> 6)] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.ms[2 * b] = (red[0] + red[1]) + (red[2] + red[3]);
    a.ms[2 * b + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

__global__ __launch_bounds__(256) void srf_remix_write_kernel(RemixArgs a) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.T) return;
  const float* row[SRF_AUG_MAX_SRC];
  float scale[SRF_AUG_MAX_SRC];
  long src[SRF_AUG_MAX_SRC];
  remix_rows(a, b, row, scale, src);
  const double dT = (double)a.T, dn = (double)(a.T > 1 ? a.T - 1 : 1);
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < SRF_AUG_MAX_SRC; ++j) {
    if (j < a.S) {
      const float v = row[j][i] * scale[j];
      m += v;
      // mean / unbiased std of the re-scaled row from the source row's statistics
      const double mu = a.rs[2 * src[j]] / dT;
      double var = (a.rs[2 * src[j] + 1] - dT * mu * mu) / dn;
      var = var < 0.0 ? 0.0 : var;
      const float mean = scale[j] * (float)mu, sd = scale[j] * (float)sqrt(var);
      a.out[((long)b * a.S + j) * a.T + i] = (v - mean) / (sd + a.eps);
    }
  }
  const double mu = a.ms[2 * b] / dT;
  double var = (a.ms[2 * b + 1] - dT * mu * mu) / dn;
  var = var < 0.0 ? 0.0 : var;
  a.mix[(long)b * a.T + i] = (m - (float)mu) / ((float)sqrt(var) + a.eps);
}

extern "C" size_t srf_online_remix_scratch_bytes(int B, int S) {
  return B > 0 && S > 0 ? sizeof(double) * 2 * ((size_t)B * S + B) : 0;
}

// clean: [B,S,T]; src_b: [S][B] int32 (device); src_s: [S] int32 (device); mix: [B,T]; out: [B,S,T] (must not alias clean)
extern "C" int srf_online_remix(const float* clean, const int* src_b, const int* src_s, int B, int S, int T, float eps,
                                float* mix, float* out, void* scratch, void* stream) {
  SRF_CHECK_ARG(clean && src_b && src_s && mix && out && scratch, "srf_online_remix: null pointer");
  SRF_CHECK_ARG(B > 0 && T > 0 && S >= 1 && S <= SRF_AUG_MAX_SRC && B <= 65535, "srf_online_remix: bad sizes (S <= 4)");
  SRF_CHECK_ARG(out != clean, "srf_online_remix: out must not alias clean (rows are gathered across the batch)");
  hipStream_t st = (hipStream_t)stream;
  RemixArgs a;
  a.clean = clean;
  a.rs = reinterpret_cast<double*>(scratch);
  a.ms = reinterpret_cast<double*>(scratch) + 2 * (size_t)B * S;
  a.src_b = src_b;
  a.src_s = src_s;
  a.mix = mix;
  a.out = out;
  a.B = B;
  a.S = S;
  a.T = T;
  a.eps = eps;
  hipLaunchKernelGGL(srf_row_stats_kernel, dim3((unsigned)(B * S)), dim3(256), 0, st, clean,
                     reinterpret_cast<double*>(scratch), T);
  SRF_CHECK_LAUNCH("remix_row_stats", st);
  hipLaunchKernelGGL(srf_remix_stats_kernel, dim3((unsigned)B), dim3(256), 0, st, a);
  SRF_CHECK_LAUNCH("remix_mix_stats", st);
  hipLaunchKernelGGL(srf_remix_write_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)B), dim3(256), 0, st, a);
  SRF_CHECK_LAUNCH("remix_write", st);
  return SRF_OK;
}
