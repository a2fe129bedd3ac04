// Shared device/host helpers for the gfx950 SuDoRM-RF kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#define SRF_DIAGNOSTICS 1   // the library itself sees (and defines) the diagnostics switches
#include "../../include/sudormrf_hip.h"

#define SRF_WAVE 64
#define SRF_MAX_DEPTH 8  // deepest upsampling_depth the merge kernel / plan support

// ---------------------------------------------------------------------------------------------
// host side: error reporting (thread-local string, never throws across the ABI)
// ---------------------------------------------------------------------------------------------
void srf_set_error(const char* fmt, ...);
int srf_kernel_mode();  // 0 = fast paths, 1 = generic kernels only, 2 = fast paths with exact-fp32 MFMA GEMMs
int srf_kernel_mode_override(int mode);   // per-thread override (-1 = none); returns the previous override
int srf_debug_flags();  // kernel-variant switches for A/B runs and tests (table: include/sudormrf_hip.h, SRF_DIAGNOSTICS)
bool srf_profiling();
void srf_prof_hold(int delta);   // +1 / -1 around all but the last launch of an operation that is one profiler interval
void srf_prof_mark(const char* name, hipStream_t st);
// Per-device launch geometry caches (srf_api.hip).  The library is called from one thread per GPU (DataParallel replicas,
// SURVEY.md §8b), so nothing device-dependent may live in a plain function-static: these are indexed by the CALLING
// thread's current device and guarded by a mutex.
#define SRF_MAX_DEVICES 64
int srf_current_device();                       // hipGetDevice, -1 on error
int srf_device_cus();                           // multiProcessorCount of the current device (256 if the query fails)
// Slot `slot` (0 .. SRF_OCC_SLOTS-1) of the current device's occupancy cache: returns the cached value or, when empty,
// evaluates `compute` once (under the lock) and stores it.
#define SRF_OCC_SLOTS 16
long srf_device_cached(int slot, long (*compute)(void*), void* arg);

#define SRF_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      srf_set_error(__VA_ARGS__);         \
      return SRF_EINVAL;                  \
    }                                     \
  } while (0)

// after every kernel launch: surface launch errors; in profiling mode drop a HIP event on the stream
#define SRF_CHECK_LAUNCH(name, st)                                                    \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      srf_set_error("%s: kernel launch failed: %s", name, hipGetErrorString(e__));    \
      return SRF_EHIP;                                                                \
    }                                                                                 \
    if (srf_profiling()) srf_prof_mark(name, (hipStream_t)(st));                      \
  } while (0)

#define SRF_CHECK_HIP(expr)                                                            \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess) {                                                          \
      srf_set_error("%s failed: %s", #expr, hipGetErrorString(e__));                  \
      return SRF_EHIP;                                                                \
    }                                                                                 \
  } while (0)

static inline bool srf_aligned16(const void* p) { return (((size_t)p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
struct SrfNormDev {  // by-value kernel argument mirror of srf_norm
  const double* sums;
  const float* gamma;
  const float* beta;
  const float* prelu;
};

static inline SrfNormDev srf_norm_dev(const srf_norm* n) {
  SrfNormDev d{nullptr, nullptr, nullptr, nullptr};
  if (n) {
    d.sums = n->sums;
    d.gamma = n->gamma;
    d.beta = n->beta;
    d.prelu = n->prelu;
  }
  return d;
}

// GlobLN statistics of one group from its fp64 {sum, sumsq}: mean and 1/sqrt(var_biased + 1e-8)
// (reference: improved_sudormrf.py:44-47).  fp64 keeps E[x^2]-mu^2 free of cancellation trouble.
//
// Statistics are spread over SRF_STAT_BUCKETS {sum,sumsq} pairs per group so that concurrent producers
// do not serialise on one address (measured: 512 same-address fp64 atomics cost ~150 us per launch on
// MI355X, a fixed floor under every streaming kernel).  Each lane of the calling wavefront loads one
// bucket and a shuffle reduction adds them: MUST be called by a full, converged 64-lane wavefront.
__device__ __forceinline__ double srf_wave_sum(double v);
__device__ __forceinline__ void srf_finalize_stats(const double* sums, long g, double inv_count,
                                                   float& mean, float& rstd) {
  const double2 bk =
      reinterpret_cast<const double2*>(sums)[g * SRF_STAT_BUCKETS + (threadIdx.x & (SRF_STAT_BUCKETS - 1))];
  const double s = srf_wave_sum(bk.x), q = srf_wave_sum(bk.y);
  const double m = s * inv_count;
  double v = q * inv_count - m * m;
  v = v < 0.0 ? 0.0 : v;
  mean = (float)m;
  rstd = (float)(1.0 / sqrt(v + 1e-8));
}

__device__ __forceinline__ double srf_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Address of bucket `bucket` (any non-negative number, reduced mod SRF_STAT_BUCKETS) of group g.
__device__ __forceinline__ double* srf_stat_slot(double* sums, long g, long bucket) {
  return sums + 2 * (g * SRF_STAT_BUCKETS + (bucket & (SRF_STAT_BUCKETS - 1)));
}

// Block-wide {sum, sumsq} -> one fp64 atomic pair.  `sh` = 2*NWAVES doubles of LDS.
// Must be reached by every thread of the block.
template <int NWAVES>
__device__ __forceinline__ void srf_block_stats_atomic(double s, double q, double* dst, double* sh) {
  s = srf_wave_sum(s);
  q = srf_wave_sum(q);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    sh[w] = s;
    sh[NWAVES + w] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i) {
      a += sh[i];
      b += sh[NWAVES + i];
    }
    atomicAdd(dst, a);
    atomicAdd(dst + 1, b);
  }
}

// Neighbour-lane exchange as DPP wavefront shifts (VALU, full rate) instead of __shfl_up/down, which lower to
// ds_bpermute_b32 (LDS pipe, ~100 cycles of latency on every level's critical path).  Lane 0 / lane 63 receive 0:
// their chunks are halo chunks whose results are discarded, exactly as the clamped shuffle's were.
__device__ __forceinline__ float srf_lane_up(float v) {     // lane l <- lane l-1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));   // wave_shr:1
}
__device__ __forceinline__ float srf_lane_down(float v) {   // lane l <- lane l+1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));   // wave_shl:1
}

// Wavefront sum with DPP row operations (VALU only; __shfl_xor lowers to ds_bpermute, an LDS-pipe
// instruction with ~50+ cycles of latency per step).  The total ends up in lane 63.
__device__ __forceinline__ float srf_dpp_wave_sum(float v) {
  auto dpp = [](float x, int ctrl_id) {
    const int xi = __float_as_int(x);
    int r;
    switch (ctrl_id) {
      case 0: r = __builtin_amdgcn_update_dpp(0, xi, 0x111, 0xf, 0xf, true); break;   // row_shr:1
      case 1: r = __builtin_amdgcn_update_dpp(0, xi, 0x112, 0xf, 0xf, true); break;   // row_shr:2
      case 2: r = __builtin_amdgcn_update_dpp(0, xi, 0x114, 0xf, 0xf, true); break;   // row_shr:4
      case 3: r = __builtin_amdgcn_update_dpp(0, xi, 0x118, 0xf, 0xf, true); break;   // row_shr:8
      case 4: r = __builtin_amdgcn_update_dpp(0, xi, 0x142, 0xa, 0xf, true); break;   // row_bcast:15 -> rows 1,3
      default: r = __builtin_amdgcn_update_dpp(0, xi, 0x143, 0xc, 0xf, true); break;  // row_bcast:31 -> rows 2,3
    }
    return __int_as_float(r);
  };
  v += dpp(v, 0);
  v += dpp(v, 1);
  v += dpp(v, 2);
  v += dpp(v, 3);   // lane 15 of every row holds its row sum
  v += dpp(v, 4);   // lanes 31 / 63 hold the sum of rows 0-1 / 2-3
  v += dpp(v, 5);   // lane 63 holds the wavefront sum
  return v;
}

// fp64 variant (two 32-bit DPP moves per step); total in lane 63.  For dependent chains of small reductions, where the
// ~100-cycle ds_bpermute steps of srf_wave_sum(double) are the critical path.
__device__ __forceinline__ double srf_dpp_wave_sum(double v) {
  auto step = [](double x, auto ctrl_tag) {
    constexpr int CTRL = decltype(ctrl_tag)::ctrl, RMASK = decltype(ctrl_tag)::rmask;
    const long long xi = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(xi & 0xffffffffLL), CTRL, RMASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(xi >> 32), CTRL, RMASK, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  };
  struct T1 { enum { ctrl = 0x111, rmask = 0xf }; };
  struct T2 { enum { ctrl = 0x112, rmask = 0xf }; };
  struct T4 { enum { ctrl = 0x114, rmask = 0xf }; };
  struct T8 { enum { ctrl = 0x118, rmask = 0xf }; };
  struct B15 { enum { ctrl = 0x142, rmask = 0xa }; };
  struct B31 { enum { ctrl = 0x143, rmask = 0xc }; };
  v += step(v, T1{});
  v += step(v, T2{});
  v += step(v, T4{});
  v += step(v, T8{});
  v += step(v, B15{});
  v += step(v, B31{});
  return v;
}

// srf_finalize_stats with the bucket reduction on the VALU (DPP) instead of 24 dependent ds_bpermute steps -- for
// kernels that finalise inside their main loop (persistent GEMM: once per tile).  Same value up to the fp64 summation
// order (<= 1 ulp of the fp32 results).  MUST be called by a full, converged wavefront.
__device__ __forceinline__ void srf_finalize_stats_dpp(const double* sums, long g, double inv_count, float& mean,
                                                       float& rstd) {
  const double2 bk = reinterpret_cast<const double2*>(sums)[g * SRF_STAT_BUCKETS + (threadIdx.x & (SRF_STAT_BUCKETS - 1))];
  auto bcast63 = [](double x) {
    const long long xi = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(xi & 0xffffffffLL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(xi >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  };
  const double s = bcast63(srf_dpp_wave_sum(bk.x)), q = bcast63(srf_dpp_wave_sum(bk.y));
  const double m = s * inv_count;
  double v = q * inv_count - m * m;
  v = v < 0.0 ? 0.0 : v;
  mean = (float)m;
  rstd = (float)(1.0 / sqrt(v + 1e-8));
}

// PReLU_a(x) = x >= 0 ? x : a x, as ONE multiply and ONE v_med3_f32 for any slope: max(x, a x) when a <= 1,
// min(x, a x) when a > 1, i.e. the median of {x, a x, +inf} resp. {x, a x, -inf}; the third operand depends on the
// (wave-uniform, loop-invariant) slope only.  The compare + select form costs three VALU instructions per element
// and showed up in every kernel that applies the activation on load (res_conv GEMM -4 %, pyramid pass 1 -7 %).
// Non-temporal (streaming) loads for tensors a kernel reads exactly once: they do not displace what the NEXT kernel will read.
// NT = false: the plain load (same code path for A/B).
template <bool NT>
__device__ __forceinline__ float4 srf_ld4(const float* p) {
  if constexpr (NT) {
    typedef float srf_f4v __attribute__((ext_vector_type(4)));
    const srf_f4v t = __builtin_nontemporal_load(reinterpret_cast<const srf_f4v*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}
template <bool NT>
__device__ __forceinline__ float2 srf_ld2(const float* p) {
  if constexpr (NT) {
    typedef float srf_f2v __attribute__((ext_vector_type(2)));
    const srf_f2v t = __builtin_nontemporal_load(reinterpret_cast<const srf_f2v*>(p));
    return make_float2(t[0], t[1]);
  } else {
    return *reinterpret_cast<const float2*>(p);
  }
}

__device__ __forceinline__ float srf_prelu(float x, float a) {
  return __builtin_amdgcn_fmed3f(x, a * x, a <= 1.f ? __builtin_inff() : -__builtin_inff());
}
