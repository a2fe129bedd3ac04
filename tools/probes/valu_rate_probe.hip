// Probe: issue rate of the fp32 VALU forms the pyramid kernels could use, per SIMD, at 1 / 2 / 4 wavefronts per SIMD:
//   v_fma_f32 (wave64), v_pk_fma_f32, v_pk_mul_f32 / v_pk_add_f32, v_pk_mov_b32 (interleave), DPP wave_shr move,
//   v_cndmask_b32, v_max_f32, and v_fma_f32 with an SGPR operand.
// Question: is a v_pk_fma_f32 (2 FMAs per lane) as cheap as a v_fma_f32?  (Decides whether a two-rows-per-lane packed
// cascade is worth building; the SQ counters say a plain VALU instruction occupies one quad-cycle.)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/valu_rate_probe.hip -o tools/probes/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int UNROLL = 16, ITERS = 2048;

template <int FORM>
__global__ __launch_bounds__(256) void rate(const float* __restrict__ in, float* __restrict__ out, unsigned long long* cyc, float s) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f2 a[UNROLL];
  f2 b = {in[tid & 1023] + 1.0f, in[(tid + 1) & 1023] - 1.0f};
  f2 c = {0.001f, -0.002f};
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) a[u] = f2{in[(tid + u) & 1023], in[(tid + 2 * u) & 1023]};
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (FORM == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[u].x) : "v"(b.x), "v"(c.x));
      if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c));
      if (FORM == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
      if (FORM == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[u]) : "v"(c));
      if (FORM == 4) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(a[u]) : "v"(a[(u + 1) % UNROLL]), "v"(a[(u + 2) % UNROLL]));
      if (FORM == 5) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[u].x) : "v"(a[(u + 1) % UNROLL].y));
      if (FORM == 6) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[u].x) : "v"(b.x));
      if (FORM == 7) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[u].x) : "v"(b.x));
      if (FORM == 8) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[u].x) : "v"(b.x), "s"(s));
      if (FORM == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[u]) : "v"(b), "v"(c));   // src1 broadcast lo
      if (FORM == 10) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[u].x) : "v"(b.x), "v"(c.x));
      if (FORM == 11) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[u].x) : "v"(b.x));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) acc += a[u].x + a[u].y;
  out[tid] = acc;
  if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

static const char* NAMES[] = {"v_fma_f32 (VGPR x3)", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mov_b32 op_sel", "v_mov_b32_dpp wave_shr:1",
                              "v_cndmask_b32", "v_max_f32", "v_fma_f32 (SGPR operand)", "v_pk_fma_f32 bcast src1.lo", "v_fmac_f32", "v_mul_f32"};

template <int FORM>
void run(const float* in, float* out, unsigned long long* cyc, int cus) {
  for (int wps = 1; wps <= 4; wps *= 2) {   // wavefronts per SIMD: blocks of 256 threads (4 waves = 1 per SIMD) x wps per CU
    const int blocks = cus * wps;
    hipLaunchKernelGGL(rate<FORM>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 1.0001f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate<FORM>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 1.0001f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[64];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < 64; ++i) mean += (double)h[i];
    mean /= 64;
    const double n = (double)UNROLL * ITERS;
    // s_memtime ticks at a constant 100 MHz on gfx950; wall time gives the rate independent of that
    const double inst_per_simd = n * wps;
    printf("%-28s waves/SIMD %d: %8.3f ms wall -> %6.2f ns per instruction per SIMD (= %.2f cycles at 2.4 GHz), memtime ticks/instr %.3f\n",
           NAMES[FORM], wps, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4, mean / n);
  }
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
  float *in, *out;
  unsigned long long* cyc;
  CK(hipMalloc(&in, 1024 * 4));
  CK(hipMalloc(&out, (size_t)cus * 4 * 256 * 4));
  CK(hipMalloc(&cyc, (size_t)cus * 4 * 4 * 8));
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 17) - 0.005f;
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  run<0>(in, out, cyc, cus);
  run<10>(in, out, cyc, cus);
  run<8>(in, out, cyc, cus);
  run<11>(in, out, cyc, cus);
  run<1>(in, out, cyc, cus);
  run<9>(in, out, cyc, cus);
  run<2>(in, out, cyc, cus);
  run<3>(in, out, cyc, cus);
  run<4>(in, out, cyc, cus);
  run<5>(in, out, cyc, cus);
  run<6>(in, out, cyc, cus);
  run<7>(in, out, cyc, cus);
  return 0;
}
