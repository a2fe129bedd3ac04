// Probe: do packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) return wrong results when a
// wavefront of ANOTHER kernel issues MFMAs on the same SIMD?  (root cause of the GroupComm two-stream
// corruption and of the GlobLN-prologue GEMM instantiation.)  A "victim" kernel evaluates one operand form of a packed
// instruction in a loop and checks every result against the two scalar instructions it stands for; an "aggressor" kernel
// on a second stream keeps every SIMD busy with one kind of instruction.  Prints mismatches per form x aggressor and the
// lanes / halves hit.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/pk_opsel_probe.hip -o tools/probes/pk_opsel_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// out[tid*4 + {0,1}] = mismatch counts of the low / high result, out[tid*4+2] = iterations run
template <int FORM>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, unsigned* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f2 a = {in[2 * tid], in[2 * tid + 1]};
  f2 b = {in[2 * tid + 1] * 0.5f + 1.f, in[2 * tid] * 0.25f - 1.f};
  f2 c = {0.125f * in[2 * tid], -0.375f * in[2 * tid + 1]};
  unsigned bad0 = 0, bad1 = 0;
  for (int it = 0; it < iters; ++it) {
    f2 r;
    float e0, e1;
    if (FORM == 0) {   // default: lo = a.lo*b.lo, hi = a.hi*b.hi
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      e0 = a.x * b.x; e1 = a.y * b.y;
    } else if (FORM == 1) {   // src1 swapped: lo = a.lo*b.hi, hi = a.lo*b.lo  (the GEMM prologue's form)
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
      e0 = a.x * b.y; e1 = a.x * b.x;
    } else if (FORM == 2) {   // broadcast src0.lo: lo = a.lo*b.lo, hi = a.lo*b.hi
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
      e0 = a.x * b.x; e1 = a.x * b.y;
    } else if (FORM == 3) {   // broadcast src0.hi: lo = a.hi*b.lo, hi = a.hi*b.hi
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
      e0 = a.y * b.x; e1 = a.y * b.y;
    } else if (FORM == 4) {   // fma, src0 broadcast hi (TAC's form): lo = a.hi*b.lo + c.lo, hi = a.hi*b.hi + c.hi
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
      e0 = __builtin_fmaf(a.y, b.x, c.x); e1 = __builtin_fmaf(a.y, b.y, c.y);
    } else if (FORM == 5) {   // fma default
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
      e0 = __builtin_fmaf(a.x, b.x, c.x); e1 = __builtin_fmaf(a.y, b.y, c.y);
    } else if (FORM == 6) {   // add default
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      e0 = a.x + b.x; e1 = a.y + b.y;
    } else if (FORM == 7) {   // add, src1 swapped
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
      e0 = a.x + b.y; e1 = a.y + b.x;
    } else if (FORM == 8) {   // fma, src2 swapped
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
      e0 = __builtin_fmaf(a.x, b.x, c.y); e1 = __builtin_fmaf(a.y, b.y, c.x);
    } else if (FORM == 10) {  // fma, src1 swapped: lo = a.lo*b.hi + c.lo, hi = a.hi*b.lo + c.hi
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
      e0 = __builtin_fmaf(a.x, b.y, c.x); e1 = __builtin_fmaf(a.y, b.x, c.y);
    } else if (FORM == 11) {  // add, src1 = SGPR pair, high half broadcast (TAC's bias add): lo = a.lo + s.hi, hi = a.hi + s.hi
      const float s0 = __builtin_amdgcn_readfirstlane(it) * 0.5f, s1 = 3.f - __builtin_amdgcn_readfirstlane(it);
      const unsigned long long sp = ((unsigned long long)__float_as_uint(s1) << 32) | __float_as_uint(s0);
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "s"(sp));
      e0 = a.x + s1; e1 = a.y + s1;
    } else if (FORM == 12) {  // fma, src0 = SGPR pair, high half broadcast (TAC's weight form)
      const float s0 = __builtin_amdgcn_readfirstlane(it) * 0.5f, s1 = 3.f - __builtin_amdgcn_readfirstlane(it);
      const unsigned long long sp = ((unsigned long long)__float_as_uint(s1) << 32) | __float_as_uint(s0);
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "s"(sp), "v"(b), "v"(c));
      e0 = __builtin_fmaf(s1, b.x, c.x); e1 = __builtin_fmaf(s1, b.y, c.y);
    } else if (FORM == 13) {  // add, src1 broadcast LOW (the replacement form): lo = a.lo + b.lo, hi = a.hi + b.lo
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
      e0 = a.x + b.x; e1 = a.y + b.x;
    } else if (FORM == 14) {  // v_pk_mov_b32: lo = a.hi, hi = b.lo
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
      e0 = a.y; e1 = b.x;
    } else if (FORM == 15) {  // v_pk_mov_b32: lo = a.lo, hi... src1 high for the low?  lo = a.lo (src0), hi = b.hi; op_sel:[0,1]: hi lane from src1
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
      e0 = a.x; e1 = b.y;
    } else {                  // (FORM 9) control: DPP row_mirror + scalar add (no packed instruction)
      int m;
      asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(m) : "v"(__float_as_int(a.x)));
      r.x = __int_as_float(m);
      r.y = a.y;
      e0 = __shfl(a.x, (threadIdx.x & 48) | (15 - (threadIdx.x & 15)), 64); e1 = a.y;
    }
    // keep the scalar reference on scalar instructions (no re-association, no fusing: the products above are single ops)
    asm volatile("" : "+v"(e0), "+v"(e1));
    bad0 += (__float_as_uint(r.x) != __float_as_uint(e0));
    bad1 += (__float_as_uint(r.y) != __float_as_uint(e1));
    a.x = a.x * 1.0009765625f + 0.001f;
    a.y = a.y * 0.9990234375f - 0.002f;
    b.x += 0.01f;
    b.y -= 0.003f;
  }
  out[tid * 4 + 0] = bad0;
  out[tid * 4 + 1] = bad1;
  out[tid * 4 + 2] = iters;
}

// aggressors: KIND 5 = bf16 MFMA 32x32x16 on two independent accumulators (back-to-back issue); KIND 0 = bf16 MFMA 32x32x16, 1 = fp32 MFMA 32x32x2, 2 = bf16 MFMA 16x16x32, 3 = v_cvt_pk_bf16_f32 + VALU, 4 = LDS traffic
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* sink, int iters) {
  __shared__ float lds[4096];
  const int t = threadIdx.x;
  f32x16 acc = {0}, accb = {0};
  f32x4 acc4 = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (t + j)); b[j] = (__bf16)(0.02f * (t - j)); }
  float s = 0.f, x = 0.001f * t;
  lds[t] = x; lds[t + 256] = -x;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else if (KIND == 5) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, accb, 0, 0, 0);
      }
    } else if (KIND == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, s + 1.f, acc, 0, 0, 0);
    } else if (KIND == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4, 0, 0, 0);
    } else if (KIND == 3) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const __bf16 h = (__bf16)x;
        x = x * 1.001f + (x - (float)h);
        s += (float)h;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += lds[(t * 4 + u * 64 + it) & 4095];
        lds[(t + u * 256) & 4095] = s;
      }
    }
  }
  float r = s + x;
  for (int j = 0; j < 16; ++j) r += acc[j] + accb[j];
  for (int j = 0; j < 4; ++j) r += acc4[j];
  sink[blockIdx.x * 256 + t] = r;
}

template <int FORM>
static void launch_victim(const float* in, unsigned* out, int blocks, int iters, hipStream_t st) {
  hipLaunchKernelGGL(victim<FORM>, dim3(blocks), dim3(256), 0, st, in, out, iters);
}
static void launch_aggr(int kind, float* sink, int blocks, int iters, hipStream_t st) {
  switch (kind) {
    case 0: hipLaunchKernelGGL(aggressor<0>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    case 1: hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    case 2: hipLaunchKernelGGL(aggressor<2>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    case 3: hipLaunchKernelGGL(aggressor<3>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    case 4: hipLaunchKernelGGL(aggressor<4>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    default: hipLaunchKernelGGL(aggressor<5>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
  }
}

int main(int argc, char** argv) {
  const int vblocks = 256 * 4, ablocks = 256 * 4, viters = argc > 1 ? atoi(argv[1]) : 20000;
  const int nthreads = vblocks * 256;
  std::vector<float> h(2 * nthreads);
  unsigned seed = 12345;
  for (auto& v : h) { seed = seed * 1664525u + 1013904223u; v = ((seed >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *in, *sink;
  unsigned* out;
  CK(hipMalloc(&in, sizeof(float) * h.size()));
  CK(hipMalloc(&out, sizeof(unsigned) * 4 * nthreads));
  CK(hipMalloc(&sink, sizeof(float) * ablocks * 256));
  CK(hipMemcpy(in, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sb));
  const char* forms[] = {"pk_mul default", "pk_mul src1 swapped (op_sel:[0,1] op_sel_hi:[0,0])", "pk_mul bcast src0.lo (op_sel_hi:[0,1])",
                         "pk_mul bcast src0.hi (op_sel:[1,0])", "pk_fma bcast src0.hi (op_sel:[1,0,0])", "pk_fma default",
                         "pk_add default", "pk_add src1 swapped", "pk_fma src2 swapped", "control: dpp row_mirror",
                         "pk_fma src1 swapped (op_sel:[0,1,0])", "pk_add src1 = SGPR pair, op_sel:[0,1]", "pk_fma src0 = SGPR pair, op_sel:[1,0,0]",
                         "pk_add src1 bcast LOW (op_sel_hi:[1,0])", "pk_mov_b32 op_sel:[1,0]", "pk_mov_b32 op_sel:[0,1]"};
  const char* aggr[] = {"none", "bf16 MFMA 32x32x16", "fp32 MFMA 32x32x2", "bf16 MFMA 16x16x32", "cvt_pk_bf16 + VALU", "LDS traffic",
                        "bf16 MFMA 32x32x16 x2 acc"};
  std::vector<unsigned> ho(4 * nthreads);
  for (int ag = 0; ag < 7; ++ag) {
    for (int f = 0; f < 16; ++f) {
      CK(hipMemset(out, 0, sizeof(unsigned) * 4 * nthreads));
      CK(hipDeviceSynchronize());
      if (ag > 0) launch_aggr(ag - 1, sink, ablocks, 60000, sb);
      switch (f) {
        case 0: launch_victim<0>(in, out, vblocks, viters, sa); break;
        case 1: launch_victim<1>(in, out, vblocks, viters, sa); break;
        case 2: launch_victim<2>(in, out, vblocks, viters, sa); break;
        case 3: launch_victim<3>(in, out, vblocks, viters, sa); break;
        case 4: launch_victim<4>(in, out, vblocks, viters, sa); break;
        case 5: launch_victim<5>(in, out, vblocks, viters, sa); break;
        case 6: launch_victim<6>(in, out, vblocks, viters, sa); break;
        case 7: launch_victim<7>(in, out, vblocks, viters, sa); break;
        case 8: launch_victim<8>(in, out, vblocks, viters, sa); break;
        case 9: launch_victim<9>(in, out, vblocks, viters, sa); break;
        case 10: launch_victim<10>(in, out, vblocks, viters, sa); break;
        case 11: launch_victim<11>(in, out, vblocks, viters, sa); break;
        case 12: launch_victim<12>(in, out, vblocks, viters, sa); break;
        case 13: launch_victim<13>(in, out, vblocks, viters, sa); break;
        case 14: launch_victim<14>(in, out, vblocks, viters, sa); break;
        default: launch_victim<15>(in, out, vblocks, viters, sa); break;
      }
      CK(hipStreamSynchronize(sa));
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(ho.data(), out, sizeof(unsigned) * 4 * nthreads, hipMemcpyDeviceToHost));
      unsigned long long lo = 0, hi = 0, row[4] = {0, 0, 0, 0};
      for (int t = 0; t < nthreads; ++t) {
        lo += ho[4 * t];
        hi += ho[4 * t + 1];
        row[(t & 63) >> 4] += ho[4 * t] + ho[4 * t + 1];
      }
      printf("aggressor %-22s | %-52s | mismatches lo %10llu hi %10llu | by lane row [%llu %llu %llu %llu]\n", aggr[ag], forms[f], lo,
             hi, row[0], row[1], row[2], row[3]);
      fflush(stdout);
    }
  }
  return 0;
}
