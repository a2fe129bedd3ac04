// Probe the semantics of ds_read_b64_tr_b16 on gfx950: fill LDS with 16-bit values equal to their own element
// index, let lane l read from byte address base[l] (printed), dump the four 16-bit values each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int* addr, uint16_t* out, int n16) {
  __shared__ uint16_t lds[8192];
  for (int i = threadIdx.x; i < n16; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x];
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
  int h_addr[64];
  int* d_addr;
  uint16_t *d_out, h_out[256];
  hipMalloc(&d_addr, 256);
  hipMalloc(&d_out, 512);
  // experiment A: lane l -> row (l & 15) of a [16 rows][pitch 64 B] image, 8-byte column block (l >> 4)
  for (int exp = 0; exp < 3; ++exp) {
    for (int l = 0; l < 64; ++l) {
      if (exp == 0) h_addr[l] = (l & 15) * 64 + (l >> 4) * 8;          // rows = lanes within 16-group, pitch 64 B
      if (exp == 1) h_addr[l] = (l & 15) * 8 + (l >> 4) * 128;         // contiguous 8-B pieces
      if (exp == 2) h_addr[l] = (l & 3) * 64 + ((l >> 2) & 3) * 8 + (l >> 4) * 256;   // 4 rows x 4 col-blocks per 16 lanes
    }
    hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out, 8192);
    hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
    printf("== experiment %d (element index = byte/2)\n", exp);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d addr elem %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2],
             h_out[l * 4 + 3]);
    }
  }
  return 0;
}
