// Fused gradient clipping + Adam over all parameters (last item): the tail of the reference
// runner's step,
//     torch.nn.utils.clip_grad_norm_(model.parameters(), clip)        run_improved_sudormrf.py:172-174
//     opt.step()                         (torch.optim.Adam, lr 1e-3)  :121-122,:176
// as two launches over every parameter tensor at once (multi-tensor: a device table of {param, grad, exp_avg,
// exp_avg_sq, size} + a chunk table), instead of torch's ~10 foreach kernels + a host sync for the norm:
//   1. sum of squares of all gradients -> fp64 buckets (no host round trip: the clip coefficient is computed on
//      the device by the update kernel from the buckets);
//   2. g' = g * min(1, max_norm / (norm + 1e-6));  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
//      p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)       -- torch.optim.Adam (no amsgrad / decay).
// Arithmetic order follows torch's _single_tensor_adam so that the two agree to fp32 round-off.
#include "srf_common.h"

struct srf_opt_tensor {   // mirrored in sudo_rm_rf_amd/_lib.py
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;
};

constexpr int OPT_CHUNK = 4096;

__global__ __launch_bounds__(256) void srf_grad_sqnorm_kernel(const srf_opt_tensor* __restrict__ tens,
                                                              const int2* __restrict__ chunks, double* __restrict__ buckets) {
  __shared__ double red[4];
  const int2 ch = chunks[blockIdx.x];
  const srf_opt_tensor t = tens[ch.x];
  const long beg = (long)ch.y * OPT_CHUNK;
  const long end = beg + OPT_CHUNK < t.n ? beg + OPT_CHUNK : t.n;
  double s = 0.0;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const double g = (double)t.g[i];
    s += g * g;
  }
  s = srf_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(buckets + (blockIdx.x & (SRF_STAT_BUCKETS - 1)), (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void srf_adam_kernel(const srf_opt_tensor* __restrict__ tens, const int2* __restrict__ chunks,
                                                       const double* __restrict__ buckets, float max_norm, float lr,
                                                       float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                                       float* __restrict__ norm_out) {
  // total gradient norm from the buckets (every wavefront redundantly: 64 loads)
  const double part = buckets[threadIdx.x & (SRF_STAT_BUCKETS - 1)];
  const float norm = (float)sqrt(srf_wave_sum(part));
  float clip = 1.f;
  if (max_norm > 0.f) {
    clip = max_norm / (norm + 1e-6f);          // torch.nn.utils.clip_grad_norm_: clip_coef clamped to 1
    clip = clip > 1.f ? 1.f : clip;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
  const int2 ch = chunks[blockIdx.x];
  const srf_opt_tensor t = tens[ch.x];
  const long beg = (long)ch.y * OPT_CHUNK;
  const long end = beg + OPT_CHUNK < t.n ? beg + OPT_CHUNK : t.n;
  const float step_size = lr / bc1;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const float g = t.g[i] * clip;
    const float m = t.m[i] + (g - t.m[i]) * (1.f - beta1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float v = t.v[i] * beta2 + (1.f - beta2) * g * g;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    t.m[i] = m;
    t.v[i] = v;
    t.p[i] = t.p[i] - step_size * (m / denom);                      // param.addcdiv_(exp_avg, denom, value=-step_size)
  }
}

// tensors: device array of n_tensors descriptors; chunks: device array of n_chunks {tensor index, chunk index}
// (chunk = 4096 elements); buckets: SRF_STAT_BUCKETS doubles of scratch; step: 1-based Adam step count.
// max_norm <= 0: no clipping.  norm_out (device, 1 float, optional) receives the pre-clip total gradient norm.
extern "C" int srf_clip_adam_step(const void* tensors, const void* chunks, int n_chunks, double* buckets, float max_norm,
                                  float lr, float beta1, float beta2, float eps, int step, float* norm_out,
                                  void* stream) {
  SRF_CHECK_ARG(tensors && chunks && buckets && n_chunks > 0 && step >= 1, "srf_clip_adam_step: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  SRF_CHECK_HIP(hipMemsetAsync(buckets, 0, sizeof(double) * SRF_STAT_BUCKETS, st));
  const srf_opt_tensor* T = reinterpret_cast<const srf_opt_tensor*>(tensors);
  const int2* C = reinterpret_cast<const int2*>(chunks);
  hipLaunchKernelGGL(srf_grad_sqnorm_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, T, C, buckets);
  SRF_CHECK_LAUNCH("grad_sqnorm", st);
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(srf_adam_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, T, C, buckets, max_norm, lr, beta1,
                     beta2, eps, bc1, bc2_sqrt, norm_out);
  SRF_CHECK_LAUNCH("clip_adam", st);
  return SRF_OK;
}

extern "C" int srf_opt_chunk_size(void) { return OPT_CHUNK; }
