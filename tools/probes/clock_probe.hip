// Probe: what do s_memtime (clock64) and s_memrealtime (wall_clock64) count, and how many of each does a
// dependent chain of N v_fma_f32 take?  Build: hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(float* out, long long* t, int n) {
  float x = out[threadIdx.x];
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) x = fmaf(x, 1.0000001f, 0.5f);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) {
    t[blockIdx.x * 2] = c1 - c0;
    t[blockIdx.x * 2 + 1] = w1 - w0;
  }
}
int main() {
  float* out;
  long long* t;
  hipMalloc(&out, 4096);
  hipMemset(out, 0, 4096);
  hipMalloc(&t, 16 * 4096);
  for (int blocks : {1, 256, 2048}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      const int n = 20000;
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, out, t, n);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long h[2];
      hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
      printf("blocks %5d: event %.3f ms  clock64 %lld  wall_clock64 %lld  -> clock64 %.1f MHz, wall %.1f MHz; "
             "%.2f clock64 ticks per dependent fma\n",
             blocks, ms, h[0], h[1], h[0] / (ms * 1e3), h[1] / (ms * 1e3), (double)h[0] / (64.0 * n));
    }
  }
  return 0;
}
