// K1 -- encoder filterbank: Conv1d(A->N, k=K, stride=h=K/2, padding=h, bias=False)
// (reference: improved_sudormrf.py:247-251,286; the zero right-padding of
// pad_to_appropriate_length :303-314 is folded into the bounds check t < T).
//
// Write-bound: [Bt,N,L] fp32 out (N*K MACs per frame, 4 B stored per K MACs).  Lanes run along
// time (coalesced 256-B stores per channel row); channels are wave-uniform so the filter taps are
// scalar (SGPR) operands of the FMAs.  GlobLN {sum,sumsq} of the output are produced here so that
// `ln` never needs its own pass over the tensor.
#include "srf_common.h"

// Fast path: one audio channel, compile-time K.  Each lane owns two frames (l, l+64) whose KT-sample
// windows live in registers; the block's input window is staged once through LDS.
template <int KT>
__global__ __launch_bounds__(256) void srf_encoder_fast_kernel(const float* __restrict__ wav,
                                                               const float* __restrict__ w,
                                                               float* __restrict__ out,
                                                               double* __restrict__ sums, int T, int N,
                                                               int L, const float* __restrict__ in_stats) {
  constexpr int H = KT / 2;
  constexpr int FR = 128;                  // frames per block
  constexpr int WIN = (FR - 1) * H + KT;   // samples needed by FR frames
  __shared__ float win[WIN];
  __shared__ double red[8];
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * FR;
  const float* xb = wav + (size_t)b * T;
  // in_stats (caller-side recipe folded in, README.md:100-104): the model sees (x - mean) / (std + 1e-9); the conv's zero
  // padding applies to the NORMALISED signal
  const float im = in_stats ? in_stats[2 * b] : 0.f, iden = in_stats ? in_stats[2 * b + 1] + 1e-9f : 1.f;
  for (int i = threadIdx.x; i < WIN; i += 256) {
    const int t = H * l0 - H + i;
    win[i] = (t >= 0 && t < T) ? (in_stats ? (xb[t] - im) / iden : xb[t]) : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float xw0[KT], xw1[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    xw0[k] = win[H * lane + k];
    xw1[k] = win[H * (lane + 64) + k];
  }
  const int la = l0 + lane, lb = l0 + 64 + lane;
  const bool va = la < L, vb = lb < L;
  double ds = 0.0, dq = 0.0;
  // blockIdx.z: a slice of the basis (small batches: 25 blocks per 4-s example would leave 9 of 10 CUs idle)
  const int per = ((N + (int)gridDim.z - 1) / (int)gridDim.z + 3) & ~3;
  const int n_lo = blockIdx.z * per, n_hi = min(N, n_lo + per);
  for (int n = n_lo + wave; n < n_hi; n += 4) {
    const float* wn = w + (size_t)n * KT;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const float wk = wn[k];
      a0 = fmaf(wk, xw0[k], a0);
      a1 = fmaf(wk, xw1[k], a1);
    }
    float* o = out + ((size_t)b * N + n) * L;
    float s = 0.f, q = 0.f;
    if (va) {
      o[la] = a0;
      s += a0;
      q = fmaf(a0, a0, q);
    }
    if (vb) {
      o[lb] = a1;
      s += a1;
      q = fmaf(a1, a1, q);
    }
    ds += (double)s;
    dq += (double)q;
  }
  if (sums) srf_block_stats_atomic<4>(ds, dq, srf_stat_slot(sums, b, blockIdx.x + blockIdx.z * gridDim.x), red);
}

// Generic path: any A, any odd K.  64 frames per block, window in dynamic LDS, taps read from LDS in
// the inner loop.
__global__ __launch_bounds__(256) void srf_encoder_generic_kernel(const float* __restrict__ wav,
                                                                  const float* __restrict__ w,
                                                                  float* __restrict__ out,
                                                                  double* __restrict__ sums, int A,
                                                                  int T, int N, int K, int L,
                                                                  const float* __restrict__ in_stats) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* red = reinterpret_cast<double*>(smem_raw);        // 8 doubles
  float* win = reinterpret_cast<float*>(smem_raw + 64);     // [A][WIN]
  const int H = K / 2;
  const int FR = 64;
  const int WIN = (FR - 1) * H + K;
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * FR;
  for (int i = threadIdx.x; i < A * WIN; i += 256) {
    const int a = i / WIN, j = i - a * WIN;
    const int t = H * l0 - H + j;
    float v = 0.f;
    if (t >= 0 && t < T) {
      v = wav[((size_t)b * A + a) * T + t];
      if (in_stats) v = (v - in_stats[2 * ((size_t)b * A + a)]) / (in_stats[2 * ((size_t)b * A + a) + 1] + 1e-9f);
    }
    win[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l = l0 + lane;
  double ds = 0.0, dq = 0.0;
  for (int n = wave; n < N; n += 4) {
    float acc = 0.f;
    for (int a = 0; a < A; ++a) {
      const float* wn = w + ((size_t)n * A + a) * K;
      const float* xw = win + a * WIN + H * lane;
      for (int k = 0; k < K; ++k) acc = fmaf(wn[k], xw[k], acc);
    }
    if (l < L) {
      out[((size_t)b * N + n) * L + l] = acc;
      ds += (double)acc;
      dq += (double)acc * (double)acc;
    }
  }
  if (sums) srf_block_stats_atomic<4>(ds, dq, srf_stat_slot(sums, b, blockIdx.x), red);
}

int srf_encoder_impl(const float* wav, const float* w, float* out, double* sums, int Bt, int A, int T, int N, int K, int L,
                     const float* in_stats, void* stream);
extern "C" int srf_encoder(const float* wav, const float* w, float* out, double* sums, int Bt, int A,
                           int T, int N, int K, int L, void* stream) {
  return srf_encoder_impl(wav, w, out, sums, Bt, A, T, N, K, L, nullptr, stream);
}
// in_stats: null, or [Bt * A][2] {mean, std} per input row: the input is normalised on load (srf_separate)
int srf_encoder_impl(const float* wav, const float* w, float* out, double* sums, int Bt, int A, int T, int N, int K, int L,
                     const float* in_stats, void* stream) {
  SRF_CHECK_ARG(wav && w && out, "srf_encoder: null pointer");
  SRF_CHECK_ARG(Bt > 0 && A > 0 && T > 0 && N > 0 && L > 0, "srf_encoder: bad sizes");
  SRF_CHECK_ARG(K >= 3 && (K & 1), "srf_encoder: enc_kernel_size must be odd (got %d)", K);
  SRF_CHECK_ARG(Bt <= 65535, "srf_encoder: batch %d too large for one launch", Bt);
  hipStream_t st = (hipStream_t)stream;
  if (srf_kernel_mode() != 1 && A == 1 && K == 21) {
    // basis slices per (time tile, example): many small blocks -- measured cfg 2 / 4 / 5: 96 / 365 / 1199 us with one slice,
    // 74 / 284 / 1105 at 8 blocks per CU, 74 / 245 / 857 at 64 (each function's 21 scalar tap loads sit in front of its
    // FMAs; more resident wavefronts hide them.  Taps through the vector-memory path with a software prefetch: slower, 108 us)
    const long bxy = (long)((L + 127) / 128) * Bt, want = 64L * srf_device_cus();
    int nz = bxy >= want ? 1 : (int)((want + bxy - 1) / bxy);
    nz = nz > N / 16 ? (N / 16 > 0 ? N / 16 : 1) : nz;       // at least 16 basis functions per block
    dim3 grid((L + 127) / 128, Bt, nz);
    hipLaunchKernelGGL(srf_encoder_fast_kernel<21>, grid, dim3(256), 0, st, wav, w, out, sums, T, N, L, in_stats);
  } else {
    const int H = K / 2;
    const size_t lds = 64 + sizeof(float) * (size_t)A * ((64 - 1) * H + K);
    SRF_CHECK_ARG(lds <= 64 * 1024, "srf_encoder: window does not fit LDS (A=%d K=%d)", A, K);
    dim3 grid((L + 63) / 64, Bt);
    hipLaunchKernelGGL(srf_encoder_generic_kernel, grid, dim3(256), lds, st, wav, w, out, sums, A, T, N,
                       K, L, in_stats);
  }
  SRF_CHECK_LAUNCH("encoder", st);
  return SRF_OK;
}
