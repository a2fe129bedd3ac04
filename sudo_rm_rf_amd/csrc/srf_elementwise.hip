// Stand-alone GlobLN pieces, TAC_norm + residual, mixture consistency, and the two small helpers of
// the decoder (weight transpose, overlap-add).  All are pure HBM streaming kernels.
//   GlobLN                      improved_sudormrf.py:30-47 (+ :24-27 gain/bias)
//   TAC_norm + residual         groupcomm_sudormrf_v2.py:378-382
//   mixture_consistency.apply   experiments/utils/mixture_consistency.py:14-36
//   ConvTranspose1d overlap-add improved_sudormrf.py:272-279,300 and crop :316-318
#include "srf_common.h"

// ---- GlobLN statistics ------------------------------------------------------------------------
// grid: (chunks, groups); each block reduces a contiguous slice of its group.
__global__ __launch_bounds__(256) void srf_gln_stats_kernel(const float* __restrict__ x,
                                                            double* __restrict__ sums, long per_group,
                                                            long per_block) {
  __shared__ double red[8];
  const long g = blockIdx.y;
  const long beg = (long)blockIdx.x * per_block;
  long end = beg + per_block;
  if (end > per_group) end = per_group;
  const float* xg = x + g * per_group;
  double ds = 0.0, dq = 0.0;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const float v = xg[i];
    ds += (double)v;
    dq += (double)v * (double)v;
  }
  srf_block_stats_atomic<4>(ds, dq, srf_stat_slot(sums, g, blockIdx.x), red);
}

extern "C" int srf_gln_stats(const float* x, double* sums, int groups, long per_group, void* stream) {
  SRF_CHECK_ARG(x && sums && groups > 0 && per_group > 0, "srf_gln_stats: bad arguments");
  SRF_CHECK_ARG(groups <= 65535, "srf_gln_stats: too many groups");
  const long per_block = 256 * 32;
  const long chunks = (per_group + per_block - 1) / per_block;
  dim3 grid((unsigned)chunks, (unsigned)groups);
  hipLaunchKernelGGL(srf_gln_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, sums, per_group,
                     per_block);
  SRF_CHECK_LAUNCH("gln_stats", stream);
  return SRF_OK;
}

// ---- GlobLN apply (optionally + residual input) ------------------------------------------------
// One block per (row = (g,c), chunk of 1024 time steps).  ADD: y = xres + GlobLN(q).
template <bool ADD>
__global__ __launch_bounds__(256) void srf_gln_apply_kernel(const float* __restrict__ xres,
                                                            const float* __restrict__ q,
                                                            float* __restrict__ y, SrfNormDev nrm,
                                                            double inv_count, int channels, int length,
                                                            int chunks) {
  const long row = blockIdx.x / chunks;
  const int chunk = blockIdx.x - row * chunks;
  const int c = (int)(row % channels);
  const long g = row / channels;
  float sc = 1.f, sh = 0.f;
  if (nrm.sums) {
    float mean, rstd;
    srf_finalize_stats(nrm.sums, g, inv_count, mean, rstd);
    sc = nrm.gamma[c] * rstd;
    sh = nrm.beta[c] - mean * sc;
  }
  const bool act = nrm.prelu != nullptr;
  const float slope = act ? nrm.prelu[0] : 1.f;
  const size_t base = (size_t)row * length;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int l = chunk * 1024 + u * 256 + threadIdx.x;
    if (l < length) {
      float v = fmaf(q[base + l], sc, sh);
      if (act) v = srf_prelu(v, slope);
      if (ADD) v = xres[base + l] + v;
      y[base + l] = v;
    }
  }
}

static int gln_apply_launch(const float* xres, const float* q, float* y, const srf_norm* norm, int groups,
                            int channels, int length, void* stream, bool add) {
  SRF_CHECK_ARG(q && y && norm && groups > 0 && channels > 0 && length > 0, "srf_gln_apply: bad arguments");
  const int chunks = (length + 1023) / 1024;
  const long blocks = (long)groups * channels * chunks;
  SRF_CHECK_ARG(blocks < (1L << 31), "srf_gln_apply: tensor too large");
  const double inv_count = 1.0 / ((double)channels * (double)length);
  SrfNormDev nd = srf_norm_dev(norm);
  if (add)
    hipLaunchKernelGGL(srf_gln_apply_kernel<true>, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, xres, q, y, nd, inv_count, channels, length, chunks);
  else
    hipLaunchKernelGGL(srf_gln_apply_kernel<false>, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, xres, q, y, nd, inv_count, channels, length, chunks);
  SRF_CHECK_LAUNCH(add ? "gln_apply_add" : "gln_apply", stream);
  return SRF_OK;
}

extern "C" int srf_gln_apply(const float* x, float* y, const srf_norm* norm, int groups, int channels,
                             int length, void* stream) {
  return gln_apply_launch(nullptr, x, y, norm, groups, channels, length, stream, false);
}

extern "C" int srf_gln_apply_add(const float* x, const float* q, float* y, const srf_norm* norm,
                                 int groups, int channels, int length, void* stream) {
  SRF_CHECK_ARG(x != nullptr, "srf_gln_apply_add: null x");
  return gln_apply_launch(x, q, y, norm, groups, channels, length, stream, true);
}

// ---- mixture consistency (uniform weights) -----------------------------------------------------
__global__ __launch_bounds__(256) void srf_mixcons_kernel(const float* __restrict__ pr,
                                                          const float* __restrict__ mix,
                                                          float* __restrict__ out, int S, int T,
                                                          long total /* Bt*T */) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long b = i / T;
  const int t = (int)(i - b * T);
  const float* p = pr + (size_t)b * S * T + t;
  float sum = 0.f;
  for (int s = 0; s < S; ++s) sum += p[(size_t)s * T];
  const float corr = (1.0f / (float)S) * (mix[i] - sum);
  float* o = out + (size_t)b * S * T + t;
  for (int s = 0; s < S; ++s) o[(size_t)s * T] = p[(size_t)s * T] + corr;
}

extern "C" int srf_mixture_consistency(const float* pr, const float* mix, float* out, int Bt, int S,
                                       int T, void* stream) {
  SRF_CHECK_ARG(pr && mix && out && Bt > 0 && S > 0 && T > 0, "srf_mixture_consistency: bad arguments");
  const long total = (long)Bt * T;
  hipLaunchKernelGGL(srf_mixcons_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, pr, mix, out, S, T, total);
  SRF_CHECK_LAUNCH("mixture_consistency", stream);
  return SRF_OK;
}

// ---- mixture consistency, 'magsq' weights (mixture_consistency.py:26-28) --------------------------
// w[b,s] = E[b,s] / (sum_s E[b,s] + 1e-9), E = mean_t pr^2.  Pass 1: one block per (b,s) row, fp64 sum of squares
// -> E as fp32 (the reference's torch.mean is fp32 as well); pass 2: the correction with the per-row weights.
__global__ __launch_bounds__(256) void srf_row_meansq_kernel(const float* __restrict__ x, float* __restrict__ e, int T) {
  __shared__ double red[4];
  const float* xr = x + (size_t)blockIdx.x * T;
  double q = 0.0;
  for (int t = threadIdx.x; t < T; t += 256) q += (double)xr[t] * (double)xr[t];
  q = srf_wave_sum(q);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
  __syncthreads();
  if (threadIdx.x == 0) e[blockIdx.x] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)T);
}

__global__ __launch_bounds__(256) void srf_mixcons_magsq_kernel(const float* __restrict__ pr, const float* __restrict__ mix,
                                                                const float* __restrict__ e, float* __restrict__ out,
                                                                int S, int T, long total /* Bt*T */) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long b = i / T;
  const int t = (int)(i - b * T);
  const float* p = pr + (size_t)b * S * T + t;
  const float* eb = e + b * S;
  float sum = 0.f, esum = 0.f;
  for (int s = 0; s < S; ++s) {
    sum += p[(size_t)s * T];
    esum += eb[s];
  }
  const float resid = mix[i] - sum;
  float* o = out + (size_t)b * S * T + t;
  for (int s = 0; s < S; ++s) o[(size_t)s * T] = p[(size_t)s * T] + (eb[s] / (esum + 1e-9f)) * resid;
}

extern "C" int srf_mixture_consistency_magsq(const float* pr, const float* mix, float* out, int Bt, int S, int T,
                                             float* work, void* stream) {
  SRF_CHECK_ARG(pr && mix && out && work && Bt > 0 && S > 0 && T > 0, "srf_mixture_consistency_magsq: bad arguments");
  SRF_CHECK_ARG((long)Bt * S < (1L << 31), "srf_mixture_consistency_magsq: too many rows");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(srf_row_meansq_kernel, dim3((unsigned)(Bt * S)), dim3(256), 0, st, pr, work, T);
  const long total = (long)Bt * T;
  hipLaunchKernelGGL(srf_mixcons_magsq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pr, mix, work,
                     out, S, T, total);
  SRF_CHECK_LAUNCH("mixture_consistency_magsq", stream);
  return SRF_OK;
}

// ---- decoder helpers ---------------------------------------------------------------------------
// wt[m][ci] = w[ci][m]   (w: [Ci][M] = ConvTranspose1d weight with (o,k) flattened to m)
__global__ __launch_bounds__(256) void srf_transpose_kernel(const float* __restrict__ w,
                                                            float* __restrict__ wt, int Ci, int M) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8) {
    const int ci = c0 + r, m = m0 + tx;
    tile[r][tx] = (ci < Ci && m < M) ? w[(size_t)ci * M + m] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, ci = c0 + tx;
    if (m < M && ci < Ci) wt[(size_t)m * Ci + ci] = tile[tx][r];
  }
}

int srf_transpose_launch(const float* w, float* wt, int Ci, int M, hipStream_t st) {
  dim3 grid((M + 31) / 32, (Ci + 31) / 32);
  hipLaunchKernelGGL(srf_transpose_kernel, grid, dim3(256), 0, st, w, wt, Ci, M);
  SRF_CHECK_LAUNCH("transpose", st);
  return SRF_OK;
}

// out[b,o,t] = sum over the <=3 (frame l, tap k) pairs with h*l + k - h = t of z[b, o*K + k, l]
// q = t / h, r = t % h:  (l=q-1,k=2h) when r == 0, (l=q,k=r+h), (l=q+1,k=r) -- added in that order.
// z: [Bt][nparts][Co * K][L]; nparts > 1 = the fused tail's per-256-channel partial frames (srf_pwconv_x3w.hip, EPI 4), summed
// here in part order (fixed order: run-to-run identical).
// A block owns OA_Q frames of one example: the [Co K] x [OA_Q + 2] frame values it needs are read ROW-WISE (coalesced; a thread
// per output sample reads 2-3 different rows at a stride of L floats -- fine for one part, not for the 4-32 of the fused tail)
// into LDS, summed over the parts on the way; then one thread per output sample (and, POST, all Co sources of it: the callers'
// post-processing of README.md:106-114 folded in -- out = est * std + mean and, with `mc`, mixture_consistency.apply against
// the mixture re-normalised on the fly from the raw waveform).
constexpr int OA_Q = 64;      // frames per block (fewer when Co * K rows of them would not fit 64 KB of LDS)
template <bool POST>
__global__ __launch_bounds__(256) void srf_overlap_add_kernel(const float* __restrict__ z, float* __restrict__ out, int Co, int K,
                                                              int L, int T, int nparts, int qw, const float* __restrict__ stats,
                                                              const float* __restrict__ wav, int mc) {
  extern __shared__ float oa_s[];                 // [Co * K][qw + 2]: frames q0 - 1 .. q0 + qw
  const int W = qw + 2;
  const int q0 = blockIdx.x * qw;
  const long b = blockIdx.y;
  const int M = Co * K, h = K / 2;
  const size_t pstride = (size_t)M * L;
  const float* zb = z + (size_t)b * nparts * pstride;
  // (four frames per load, four parts in flight: 64 B per thread keep enough bytes in flight to cover the HBM latency -- one
  // dword at a time measured 2.7 TB/s on cfg 5's 1.1 GB of partial frames)
  const int W4 = W >> 2;                          // W % 4 == 0 (host)
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  for (int idx = threadIdx.x; idx < M * W4; idx += 256) {
    const int m = idx / W4, j = (idx - m * W4) * 4;
    const int l = q0 - 1 + j;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (l >= 0 && l + 3 < L) {
      const float* src = zb + (size_t)m * L + l;
      int p = 0;
      for (; p + 4 <= nparts; p += 4, src += 4 * pstride) {
        const f32x4u t0 = *reinterpret_cast<const f32x4u*>(src), t1 = *reinterpret_cast<const f32x4u*>(src + pstride),
                     t2 = *reinterpret_cast<const f32x4u*>(src + 2 * pstride),
                     t3 = *reinterpret_cast<const f32x4u*>(src + 3 * pstride);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (t0[e] + t1[e]) + (t2[e] + t3[e]);
      }
      for (; p < nparts; ++p, src += pstride) {
        const f32x4u t0 = *reinterpret_cast<const f32x4u*>(src);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += t0[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (l + e >= 0 && l + e < L) {
          const float* src = zb + (size_t)m * L + l + e;
          int p = 0;
          for (; p + 4 <= nparts; p += 4, src += 4 * pstride) v[e] += (src[0] + src[pstride]) + (src[2 * pstride] + src[3 * pstride]);
          for (; p < nparts; ++p, src += pstride) v[e] += *src;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) oa_s[m * W + j + e] = v[e];
  }
  __syncthreads();
  float mean = 0.f, sd = 1.f;
  if constexpr (POST) {
    mean = stats[2 * b];
    sd = stats[2 * b + 1];
  }
  for (int tt = threadIdx.x; tt < qw * h; tt += 256) {
    const int t = q0 * h + tt;
    if (t >= T) break;
    const int qi = tt / h, r = tt - qi * h;       // frame q0 + qi is column qi + 1
    auto est = [&](int o) {
      const float* row = oa_s + (size_t)o * K * W;
      float acc = 0.f;
      if (r == 0) acc += row[(2 * h) * W + qi];
      acc += row[(r + h) * W + qi + 1];
      acc += row[r * W + qi + 2];
      return POST ? acc * sd + mean : acc;
    };
    float corr = 0.f;
    if (POST && mc) {
      float tot = 0.f;
      for (int o = 0; o < Co; ++o) tot += est(o);
      corr = ((wav[b * (long)T + t] - mean) / (sd + 1e-9f) - tot) * (1.f / (float)Co);
    }
    for (int o = 0; o < Co; ++o) {
      const float v = est(o);
      out[((size_t)b * Co + o) * T + t] = (POST && mc) ? v + corr : v;
    }
  }
}

// stats: null = plain overlap-add; else [Bt][2] {mean, std} of the raw mixture `wav` [Bt][T] (mc: also mixture consistency)
int srf_overlap_add_launch(const float* z, float* out, int Bt, int Co, int K, int L, int T, int nparts, const float* stats,
                           const float* wav, int mc, hipStream_t st) {
  // frames per block: rows of 128 floats (fewer partly used cache lines at the unaligned row ends) when there are partial
  // frames to sum and the launch still has >= 4 blocks per CU, else rows of 64 (the single-part case is latency-bound:
  // more, smaller blocks); fewer when Co * K rows would not fit 64 KB of LDS.  qw + 2 must be a multiple of 4.
  const int cap = ((int)(64 * 1024 / sizeof(float) / ((size_t)Co * K)) & ~3) - 2;
  const int nfr = (T + K / 2 - 1) / (K / 2);
  int qw = OA_Q - 2;
  if (nparts >= 4 && (long)((nfr + 2 * OA_Q - 3) / (2 * OA_Q - 2)) * Bt >= 1024) qw = 2 * OA_Q - 2;
  qw = qw > cap ? cap : qw;
  SRF_CHECK_ARG(qw >= 2, "srf_decoder: %d x %d frame rows exceed the overlap-add's LDS tile", Co, K);
  const size_t lds = sizeof(float) * (size_t)Co * K * (qw + 2);
  const int h = K / 2;
  dim3 grid(((T + h - 1) / h + qw - 1) / qw, Bt);
  if (stats)
    hipLaunchKernelGGL(srf_overlap_add_kernel<true>, grid, dim3(256), lds, st, z, out, Co, K, L, T, nparts, qw, stats, wav, mc);
  else
    hipLaunchKernelGGL(srf_overlap_add_kernel<false>, grid, dim3(256), lds, st, z, out, Co, K, L, T, nparts, qw, stats, wav, mc);
  SRF_CHECK_LAUNCH("overlap_add", st);
  return SRF_OK;
}

// ---- caller-side pre/post-processing -------------------------------------
// Every caller of the reference wraps model() in the same few torch ops (README.md:100-114,
// simple_whamr_evaluation.py:142-148):
//     std = x.std(-1, keepdim=True); mean = x.mean(-1, keepdim=True)      (std is UNBIASED, torch default)
//     y = model(((x - mean) / (std + 1e-9)).unsqueeze(1))
//     y = y * std + mean;  [GroupComm:] y = mixture_consistency.apply(y, normalised_mixture)
// srf_wav_normalize = the first two lines (one block per row: mean, then sum of squared deviations, both
// fp64, then the normalised row; the row stays in L2 between the three sweeps); srf_wav_denormalize = the
// last line in one pass.
__global__ __launch_bounds__(256) void srf_wav_normalize_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                float* __restrict__ stats, int T) {
  __shared__ double red[8];
  __shared__ double bc;
  const long r = blockIdx.x;
  const float* xr = x + r * (long)T;
  const int tid = threadIdx.x;
  auto block_sum = [&](double v) {
    v = srf_wave_sum(v);
    __syncthreads();   // red / bc free again
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) bc = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return bc;
  };
  double s = 0.0;
  for (int i = tid; i < T; i += 256) s += (double)xr[i];
  const double mean = block_sum(s) / (double)T;
  double q = 0.0;
  for (int i = tid; i < T; i += 256) {
    const double d = (double)xr[i] - mean;
    q += d * d;
  }
  const double var = block_sum(q) / (double)(T > 1 ? T - 1 : 1);
  const float fm = (float)mean, fs = (float)sqrt(var);
  if (tid == 0) {
    stats[2 * r] = fm;
    stats[2 * r + 1] = fs;
  }
  if (!y) return;   // statistics only (srf_wav_stats: the normalisation itself happens in the encoder's load)
  const float den = fs + 1e-9f;
  float* yr = y + r * (long)T;
  for (int i = tid; i < T; i += 256) yr[i] = (xr[i] - fm) / den;
}

extern "C" int srf_wav_stats(const float* wav, float* stats, int rows, int T, void* stream) {
  SRF_CHECK_ARG(wav && stats && rows > 0 && T > 0, "srf_wav_stats: bad arguments");
  hipLaunchKernelGGL(srf_wav_normalize_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, wav, (float*)nullptr,
                     stats, T);
  SRF_CHECK_LAUNCH("wav_stats", stream);
  return SRF_OK;
}

extern "C" int srf_wav_normalize(const float* wav, float* out, float* stats, int rows, int T, void* stream) {
  SRF_CHECK_ARG(wav && out && stats && rows > 0 && T > 0, "srf_wav_normalize: bad arguments");
  hipLaunchKernelGGL(srf_wav_normalize_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, wav, out,
                     stats, T);
  SRF_CHECK_LAUNCH("wav_normalize", stream);
  return SRF_OK;
}

// est [Bt,S,T] (model output for the normalised mixture), stats [Bt][2], mix_norm [Bt,1,T] or null
__global__ __launch_bounds__(256) void srf_wav_denormalize_kernel(const float* __restrict__ est,
                                                                  const float* __restrict__ stats,
                                                                  const float* __restrict__ mix, float* __restrict__ out,
                                                                  int S, int T) {
  const long b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float mean = stats[2 * b], sd = stats[2 * b + 1];
  const float* e = est + b * (long)S * T + t;
  float* o = out + b * (long)S * T + t;
  if (!mix) {
    for (int s = 0; s < S; ++s) o[(long)s * T] = e[(long)s * T] * sd + mean;
  } else {
    float tot = 0.f;
    for (int s = 0; s < S; ++s) tot += e[(long)s * T] * sd + mean;
    const float corr = (mix[b * (long)T + t] - tot) * (1.f / (float)S);
    for (int s = 0; s < S; ++s) o[(long)s * T] = (e[(long)s * T] * sd + mean) + corr;
  }
}

extern "C" int srf_wav_denormalize(const float* est, const float* stats, const float* mix_norm, float* out, int Bt,
                                   int S, int T, void* stream) {
  SRF_CHECK_ARG(est && stats && out && Bt > 0 && S > 0 && T > 0 && Bt <= 65535, "srf_wav_denormalize: bad arguments");
  dim3 grid((unsigned)((T + 255) / 256), (unsigned)Bt);
  hipLaunchKernelGGL(srf_wav_denormalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, est, stats, mix_norm, out,
                     S, T);
  SRF_CHECK_LAUNCH("wav_denormalize", stream);
  return SRF_OK;
}
