// How many bytes per clock can ONE CU pull through its vector-memory path while all 256 CUs do the same?
// (round 4: the 256 x 128 GEMMs move ~14 B / clock / CU whatever their instruction mix -- is that a ceiling?)
//
// One 512-thread block per CU (100 KB of LDS claimed), every wavefront keeps DEPTH 16-byte loads per lane in flight.
//   src 0: every block streams the SAME small buffer (L2-resident after the first pass: the GEMM's packed weights)
//   src 1: every block streams its own slice of a large buffer (HBM / MALL: the GEMM's activations)
//   src 2: half the wavefronts do src 0, half src 1 (the GEMM's mix)
//   kind 0: buffer_load_dwordx4 -> VGPRs      kind 1: global_load_lds_dwordx4 -> LDS (DMA)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/cu_stream_probe.hip -o tools/probes/cu_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ small_buf, size_t small_bytes, const char* __restrict__ big,
                                             size_t slice_bytes, int src, int passes, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool use_small = src == 0 || (src == 2 && (wave & 1) == 0);
  const char* base = use_small ? small_buf : big + (size_t)blockIdx.x * slice_bytes;
  const size_t bytes = use_small ? small_bytes : slice_bytes;
  const size_t per_wave = bytes / 8;                  // each wavefront walks its own eighth, 1 KB per instruction
  const char* p0 = base + (size_t)wave * per_wave;
  const int n_inst = (int)(per_wave / 1024);
  u32x4 acc = {0, 0, 0, 0};
  for (int pass = 0; pass < passes; ++pass) {
    for (int i = 0; i < n_inst; i += DEPTH) {
      if constexpr (KIND == 0) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const u32x4*>(p0 + (size_t)(i + d) * 1024 + lane * 16);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
      } else {
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) void*)(smem + wave * (DEPTH * 1024))));
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(p0 + (size_t)(i + d) * 1024 + lane * 16), "s"(dst + d * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  if (KIND == 1) acc[0] = *reinterpret_cast<unsigned*>(smem + tid * 4);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[tid] = acc[0];
}

// Timed co-run: wavefronts with (wave & mask) == 0 stream the SAME small (L2-resident) buffer, the others their block's own slice
// (HBM), all until s_memrealtime passes `ticks` (100 MHz); bytes per class are counted.  Is an L2 hit delayed by the HBM misses
// queued ahead of it in the same CU's vector-memory pipe?
template <int DEPTH, int AUX_L2 = 0, int AUX_HBM = 0>
__global__ __launch_bounds__(512) void corun(const char* __restrict__ small_buf, size_t small_bytes, const char* __restrict__ big,
                                             size_t slice_bytes, int mask, unsigned ticks, unsigned long long* __restrict__ counts,
                                             unsigned* __restrict__ sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool use_small = mask >= 0 && (wave & mask) == 0;
  const bool idle = mask == -2 && wave >= 4;          // (mask -2: only four HBM wavefronts, the others exit)
  if (idle) return;
  const char* base = use_small ? small_buf : big + (size_t)blockIdx.x * slice_bytes;
  const size_t bytes = use_small ? small_bytes : slice_bytes;
  const size_t per_wave = bytes / 8;
  const char* p0 = base + (size_t)wave * per_wave;
  const int n_inst = (int)(per_wave / 1024);
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long n = 0;
  int i = 0;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p0), 0, (int)per_wave, 0x00020000);
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (use_small) v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (i + d) * 1024, AUX_L2);
      else v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (i + d) * 1024, AUX_HBM);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    i += DEPTH;
    if (i + DEPTH > n_inst) i = 0;
    n += DEPTH;
  }
  if (lane == 0) atomicAdd(&counts[use_small ? 0 : 1], n * 1024ull);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[tid] = acc[0];
}

template <int AUX_L2 = 0, int AUX_HBM = 0>
void run_corun(const char* name, const char* small_buf, size_t small_bytes, const char* big, size_t slice, int mask, unsigned* sink,
               int ncu) {
  unsigned long long* counts;
  CHECK(hipMalloc(&counts, 16));
  const unsigned ticks = 200000;      // 2 ms
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(counts, 0, 16));
    hipLaunchKernelGGL((corun<8, AUX_L2, AUX_HBM>), dim3(ncu), dim3(512), 0, 0, small_buf, small_bytes, big, slice, mask, ticks, counts, sink);
    CHECK(hipDeviceSynchronize());
  }
  unsigned long long h[2];
  CHECK(hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost));
  printf("%-58s L2-resident class %7.1f GB/s per CU | HBM class %6.1f GB/s per CU (%.2f TB/s chip)\n", name,
         h[0] / 2e-3 / 1e9 / ncu, h[1] / 2e-3 / 1e9 / ncu, h[1] / 2e-3 / 1e12);
  CHECK(hipFree(counts));
}

// The GEMM's request pattern per "k-step" and CU: 32 KB of packed weights (the same 512 KB for everybody: L2 hits) by LDS-DMA and
// 16 KB of activations (own slice: HBM) -- who issues them, and in which order, decides how long a weight piece sits in the
// CU's in-order memory pipe.  All pieces by LDS-DMA (1 KB per wave instruction).  Steps per microsecond per CU are reported.
//   mode 0: all 8 wavefronts, each 4 weight + 2 activation pieces per step, interleaved (the round-3 kernel's shape)
//   mode 1: ONE wavefront issues the step's 32 weight pieces, then its 16 activation pieces (bursts), one step in flight
//   mode 2: one wavefront for the weights, another for the activations, each with one step in flight
//   mode 3: as mode 1, two steps in flight        mode 4: as mode 1, activation burst first
template <int MODE>
__global__ __launch_bounds__(512) void pattern(const char* __restrict__ small_buf, size_t small_bytes, const char* __restrict__ big,
                                               size_t slice_bytes, unsigned ticks, unsigned long long* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwork = MODE == 0 ? 8 : MODE == 2 ? 2 : 1;
  if (wave >= nwork) return;
  const char* xs = big + (size_t)blockIdx.x * slice_bytes;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) void*)smem));
  auto dma = [&](const char* src, unsigned dst) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src + lane * 16), "s"(dst) : "memory");
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long steps = 0;
  size_t ao = 0, xo = 0;       // byte cursors of the step inside the weight image / the activation slice
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    const unsigned stage = (unsigned)(steps % 3) * 49152u;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dma(small_buf + ao + (wave * 4 + i) * 1024, lds0 + stage + (wave * 4 + i) * 1024);
        if (i & 1) dma(xs + xo + (wave * 2 + (i >> 1)) * 1024, lds0 + stage + 32768 + (wave * 2 + (i >> 1)) * 1024);
      }
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if constexpr (MODE == 1 || MODE == 3 || MODE == 4) {
      if constexpr (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) dma(xs + xo + i * 1024, lds0 + stage + 32768 + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) dma(small_buf + ao + i * 1024, lds0 + stage + i * 1024);
      if constexpr (MODE != 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) dma(xs + xo + i * 1024, lds0 + stage + 32768 + i * 1024);
      }
      if constexpr (MODE == 3) {
        // (vmcnt saturates at 63: two steps = 96 pieces cannot be counted; wait for all but the newest 63)
        asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      }
    } else {
      if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) dma(small_buf + ao + i * 1024, lds0 + stage + i * 1024);
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) dma(xs + xo + i * 1024, lds0 + stage + 32768 + i * 1024);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      }
    }
    ao += 32768;
    if (ao + 32768 > small_bytes) ao = 0;
    xo += 16384;
    if (xo + 16384 > slice_bytes) xo = 0;
    ++steps;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && wave == 0) atomicAdd(&counts[0], steps);
  if (lane == 0 && wave == 1 && MODE == 2) atomicAdd(&counts[1], steps);
}

// The role-split GEMM without its synchronisation: wavefronts 0-7 loop over {16 ds_read_b128, 24 MFMAs} (WORK bit 0: the reads,
// bit 1: the MFMAs), wavefronts 8-9 stream the weight bursts and 10-11 the activation bursts by LDS-DMA (16 / 8 pieces each per
// step, one step in flight).  Nobody waits for anybody: what does each side lose to the other?
typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));
typedef float f32x16p __attribute__((ext_vector_type(16)));
template <int WORK, int LOAD>
__global__ __launch_bounds__(768) void split_probe(const char* __restrict__ small_buf, size_t small_bytes, const char* __restrict__ big,
                                                   size_t slice_bytes, unsigned ticks, unsigned long long* __restrict__ counts,
                                                   float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long steps = 0;
  // LOAD 4 / 5: 8 (12) wavefronts; the loaders are the wavefronts with (wave & 3) == 3 -- they share ONE SIMD (a block's wavefronts
  // go to the SIMDs round-robin) -- one streams the weight bursts, the other the activation bursts, 32 / 16 pieces per step
  const bool one_simd = LOAD >= 4;
  const bool is_mult = one_simd ? (wave & 3) != 3 : wave < 8;
  if (is_mult) {
    if (WORK == 0) return;
    if (LOAD == 3) __builtin_amdgcn_s_setprio(0);
    f32x16p acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8p f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) f[i][e] = (__bf16)(0.001f * (lane + i + e));
    while ((WORK & 4) ? steps < 2000 : __builtin_amdgcn_s_memrealtime() - t0 < ticks) {
      const int stage = (int)(steps % 3) * 49152;
      if constexpr (WORK & 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          f[i] = *reinterpret_cast<const bf16x8p*>(smem + stage + ((wave * 16 + i) * 1024 + lane * 16) % 49152);
      }
      if constexpr (WORK & 2) {
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(p * 4 + i) & 15], f[(p + i + 5) & 15], acc[i], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(f[i]));
      }
      if constexpr (WORK & 4) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      ++steps;
    }
    if (WORK & 4) {
      if (lane == 0 && wave == 0) counts[3] = __builtin_amdgcn_s_memrealtime() - t0;
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][7];
    if (t == 123.456f) sink[threadIdx.x] = t;
    if (lane == 0) atomicAdd(&counts[2], steps);
    return;
  }
  if (LOAD == 0) return;
  if (LOAD == 2) __builtin_amdgcn_s_setprio(3);       // loaders outrank the (older) multipliers at the instruction arbiter
  const char* xs = big + (size_t)blockIdx.x * slice_bytes;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) void*)smem));
  auto dma = [&](const char* src, unsigned dst) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src + lane * 16), "s"(dst) : "memory");
  };
  const int half = wave & 1;
  size_t ao = 0, xo = 0;
  if (one_simd) {
    if (wave > 7) return;
    while ((WORK & 4) ? steps < 2000 : __builtin_amdgcn_s_memrealtime() - t0 < ticks) {
      const unsigned stage = (unsigned)(steps % 3) * 49152u;
      if (wave == 3) {
#pragma unroll
        for (int i = 0; i < 32; ++i) dma(small_buf + ao + i * 1024, lds0 + stage + i * 1024);
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) dma(xs + xo + i * 1024, lds0 + stage + 32768 + i * 1024);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      }
      if constexpr (WORK & 4) __builtin_amdgcn_s_barrier();
      ao += 32768;
      if (ao + 32768 > small_bytes) ao = 0;
      xo += 16384;
      if (xo + 16384 > slice_bytes) xo = 0;
      ++steps;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && wave == 3) atomicAdd(&counts[0], steps);
    if (lane == 0 && wave == 7) atomicAdd(&counts[1], steps);
    return;
  }
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    const unsigned stage = (unsigned)(steps % 3) * 49152u;
    if (wave < 10) {
#pragma unroll
      for (int i = 0; i < 16; ++i) dma(small_buf + ao + (half * 16 + i) * 1024, lds0 + stage + (half * 16 + i) * 1024);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) dma(xs + xo + (half * 8 + i) * 1024, lds0 + stage + 32768 + (half * 8 + i) * 1024);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    ao += 32768;
    if (ao + 32768 > small_bytes) ao = 0;
    xo += 16384;
    if (xo + 16384 > slice_bytes) xo = 0;
    ++steps;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && wave == 8) atomicAdd(&counts[0], steps);
  if (lane == 0 && wave == 10) atomicAdd(&counts[1], steps);
}

template <int WORK, int LOAD>
void run_split(const char* name, const char* small_buf, size_t small_bytes, const char* big, size_t slice, int ncu, float* sink) {
  unsigned long long* counts;
  CHECK(hipMalloc(&counts, 32));
  const unsigned ticks = 200000;      // 2 ms
  if (WORK & 4) {
    CHECK(hipFuncSetAttribute((const void*)&split_probe<WORK, LOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(counts, 0, 32));
      hipLaunchKernelGGL((split_probe<WORK, LOAD>), dim3(ncu), dim3(512), 3 * 49152, 0, small_buf, small_bytes, big, slice, ticks, counts, sink);
      CHECK(hipDeviceSynchronize());
    }
    unsigned long long h[4];
    CHECK(hipMemcpy(h, counts, 32, hipMemcpyDeviceToHost));
    const double us = h[3] / 100.0;
    printf("%-58s 2000 steps with a block barrier each: %.3f us per step (block 0); 48 MFMAs per SIMD and step = %.0f %% of the pipe at 2.4 GHz\n",
           name, us / 2000, 100.0 * 48 * 32 / 2400.0 / (us / 2000));
    CHECK(hipFree(counts));
    return;
  }
  CHECK(hipFuncSetAttribute((const void*)&split_probe<WORK, LOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(counts, 0, 32));
    hipLaunchKernelGGL((split_probe<WORK, LOAD>), dim3(ncu), dim3(LOAD == 4 ? 512 : 768), 3 * 49152, 0, small_buf, small_bytes, big, slice, ticks, counts, sink);
    CHECK(hipDeviceSynchronize());
  }
  unsigned long long h[3];
  CHECK(hipMemcpy(h, counts, 24, hipMemcpyDeviceToHost));
  printf("%-58s weight steps %.3f /us | activation steps %.3f /us | multiplier steps %.3f /us per wavefront (24 MFMAs each: %.0f %% of the pipe at 2.4 GHz)\n",
         name, h[0] / 2e3 / ncu, h[1] / 2e3 / ncu, h[2] / 2e3 / ncu / (LOAD == 4 ? 6 : LOAD == 5 ? 9 : 8),
         100.0 * (h[2] / 2e3 / ncu / (LOAD == 4 ? 6 : LOAD == 5 ? 9 : 8)) * (LOAD == 5 ? 3 : 2) * 24 * 32 / 2400.0);
  CHECK(hipFree(counts));
}

template <int MODE>
void run_pattern(const char* name, const char* small_buf, size_t small_bytes, const char* big, size_t slice, int ncu) {
  unsigned long long* counts;
  CHECK(hipMalloc(&counts, 16));
  const unsigned ticks = 200000;      // 2 ms
  CHECK(hipFuncSetAttribute((const void*)&pattern<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(counts, 0, 16));
    hipLaunchKernelGGL((pattern<MODE>), dim3(ncu), dim3(512), 3 * 49152, 0, small_buf, small_bytes, big, slice, ticks, counts);
    CHECK(hipDeviceSynchronize());
  }
  unsigned long long h[2];
  CHECK(hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost));
  const double sps = h[0] / 2e3 / ncu;    // steps per us per CU
  printf("%-66s %.3f steps/us/CU = %.2f us/step, %5.1f GB/s per CU (HBM share %.2f TB/s chip)%s\n", name, sps, 1.0 / sps,
         sps * 49152 / 1e3, sps * 16384 * ncu / 1e6, MODE == 2 ? " [weight wave]" : "");
  if (MODE == 2) printf("%-66s %.3f steps/us/CU [activation wave]\n", "", h[1] / 2e3 / ncu);
  CHECK(hipFree(counts));
}

template <int KIND, int DEPTH>
void run(const char* name, const char* small_buf, size_t small_bytes, const char* big, size_t slice, int src, unsigned* sink, int ncu) {
  const size_t lds = 100 * 1024;
  CHECK(hipFuncSetAttribute((const void*)&probe<KIND, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const size_t per_block = src == 0 ? small_bytes : src == 1 ? slice : (small_bytes + slice) / 2;
  const int passes = (int)((size_t)(64 << 20) / per_block) + 1;       // ~64 MB per CU
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<KIND, DEPTH>), dim3(ncu), dim3(512), lds, 0, small_buf, small_bytes, big, slice, src, passes, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)per_block * passes;
  printf("%-34s kind %d depth %2d: %7.1f GB/s per CU, %6.2f TB/s chip (%d blocks, %.2f ms)\n", name, KIND, DEPTH,
         bytes / (ms * 1e-3) / 1e9, bytes * ncu / (ms * 1e-3) / 1e12, ncu, ms);
}

int main(int argc, char** argv) {
  const int ncu = argc > 1 ? atoi(argv[1]) : 256;
  const size_t small_bytes = 512 << 10, slice = 8 << 20;
  char *small_buf, *big;
  unsigned* sink;
  CHECK(hipMalloc(&small_buf, small_bytes));
  CHECK(hipMalloc(&big, slice * 256));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemset(small_buf, 1, small_bytes));
  CHECK(hipMemset(big, 1, slice * 256));
  run<0, 4>("same 512 KB in every block (L2)", small_buf, small_bytes, big, slice, 0, sink, ncu);
  run<0, 8>("same 512 KB in every block (L2)", small_buf, small_bytes, big, slice, 0, sink, ncu);
  run<1, 4>("same 512 KB in every block (L2)", small_buf, small_bytes, big, slice, 0, sink, ncu);
  run<1, 8>("same 512 KB in every block (L2)", small_buf, small_bytes, big, slice, 0, sink, ncu);
  run<0, 8>("own 8 MB slice per block (HBM)", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run<1, 8>("own 8 MB slice per block (HBM)", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run<0, 8>("half the waves each (the GEMM mix)", small_buf, small_bytes, big, slice, 2, sink, ncu);
  run<1, 8>("half the waves each (the GEMM mix)", small_buf, small_bytes, big, slice, 2, sink, ncu);
  run_corun("timed: all 8 wavefronts L2-resident", small_buf, small_bytes, big, slice, 0, sink, ncu);
  run_corun("timed: all 8 wavefronts HBM", small_buf, small_bytes, big, slice, -1, sink, ncu);
  run_corun("timed: 4 wavefronts HBM, 4 idle", small_buf, small_bytes, big, slice, -2, sink, ncu);
  run_corun("timed: 4 wavefronts L2-resident + 4 HBM in every CU", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run_corun("timed: 2 wavefronts L2-resident + 6 HBM in every CU", small_buf, small_bytes, big, slice, 3, sink, ncu);
  run_pattern<0>("pattern: 8 wavefronts, weight + activation pieces interleaved", small_buf, small_bytes, big, slice, ncu);
  run_pattern<1>("pattern: 1 wavefront, weight burst then activation burst", small_buf, small_bytes, big, slice, ncu);
  run_pattern<4>("pattern: 1 wavefront, activation burst then weight burst", small_buf, small_bytes, big, slice, ncu);
  run_pattern<3>("pattern: 1 wavefront, bursts, ~two steps in flight", small_buf, small_bytes, big, slice, ncu);
  run_pattern<2>("pattern: 1 weight wavefront + 1 activation wavefront", small_buf, small_bytes, big, slice, ncu);
  run_split<3, 0>("split: multipliers alone (reads + MFMAs)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<2, 0>("split: multipliers alone (MFMAs only)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<1, 0>("split: multipliers alone (reads only)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<0, 1>("split: loaders alone", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<3, 1>("split: loaders + multipliers (reads + MFMAs)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<2, 1>("split: loaders + multipliers (MFMAs only)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<1, 1>("split: loaders + multipliers (reads only)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<3, 4>("ONE SIMD for the 2 loaders, 6 multipliers (reads + MFMAs)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<2, 4>("ONE SIMD for the 2 loaders, 6 multipliers (MFMAs only)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<7, 4>("ONE SIMD for the 2 loaders, 6 multipliers, BARRIER per step", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<6, 4>("same, multipliers without the LDS reads", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<3, 5>("ONE SIMD for the 2 loaders, 9 multipliers (reads + MFMAs)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<3, 2>("split: loaders at s_setprio 3 + multipliers (reads + MFMAs)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  run_split<2, 2>("split: loaders at s_setprio 3 + multipliers (MFMAs only)", small_buf, small_bytes, big, slice, ncu, (float*)sink);
  // cache-policy bits of the two classes (aux: 1 = sc0, 2 = nt, 16 = sc1)
  run_corun<2, 0>("4 + 4, L2 class nt", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run_corun<0, 2>("4 + 4, HBM class nt", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run_corun<2, 2>("4 + 4, both nt", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run_corun<16, 0>("4 + 4, L2 class sc1", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run_corun<0, 16>("4 + 4, HBM class sc1", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run_corun<17, 17>("4 + 4, both sc0 sc1", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run_corun<1, 0>("4 + 4, L2 class sc0", small_buf, small_bytes, big, slice, 1, sink, ncu);
  run_corun<0, 18>("4 + 4, HBM class nt sc1", small_buf, small_bytes, big, slice, 1, sink, ncu);
  return 0;
}
