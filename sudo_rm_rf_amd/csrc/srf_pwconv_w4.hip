// K2, small launches -- the split-precision 1x1-conv GEMM of srf_pwconv_bf16x3.hip on 64(M) x 64(time) x 32(K) tiles.
//
// A batch-1 forward (the README's inference recipe, README.md:100-106; SURVEY.md §8 cfg 1) gives proj_1x1 4 x 25 = 100 and
// res_conv 2 x 25 = 50 tiles of 128 x 128 on a 256-CU chip: most CUs idle, and each busy one walks its whole K loop with two
// wavefronts per SIMD (26 us per launch, round 2 / 3 profiles).  The work is latency- and issue-bound per block, not
// bandwidth-bound, so the cure is more, smaller blocks: 64 x 64 tiles make 400 / 200 of them, four wavefronts each (2 x 2, one
// 32 x 32 accumulator per wavefront), 40 KB of LDS -- up to four blocks per CU -- and a quarter of the MFMA / split work per
// k-step on every block's critical path.
//
// Everything else is the 128 x 128 kernel's: operands split on the fly into bf16 hi | lo ([row][32 k] LDS images, 80-B pitch),
// three MFMAs per product block in the order lo*hi, hi*lo, hi*hi (so the results are BITWISE those of the other split-bf16
// kernels: same products, same summation order along K), global loads two k-tiles ahead, GlobLN / PReLU prologue on the
// activation operand, bias / residual / ReLU x mul / {sum, sumsq} epilogue through a wave-private LDS strip.
// (reference sites: improved_sudormrf.py:256-259, :174, :196, :220, :268-269, :295-298)
#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int W4_BM = 64, W4_BN = 64, W4_BK = 32;
constexpr int W4_PITCH = 80;                    // bytes per LDS row (32 bf16 = 64 B + 16 B pad: conflict-free b128 fragment reads)
constexpr int W4_IMG = W4_BM * W4_PITCH;        // one [64][32] bf16 image
constexpr int W4_STAGE = 4 * W4_IMG;            // A_hi, A_lo, B_hi, B_lo = 20 KB

__device__ __forceinline__ void w4_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

// PRO: 0 = identity, 1 = GlobLN, 2 = GlobLN + PReLU, 3 = PReLU only
template <int PRO>
__global__ __launch_bounds__(256, 4) void srf_pw_w4_kernel(PwArgs a, int nMt, int nLt, int total) {
  __shared__ __attribute__((aligned(16))) char smem[2 * W4_STAGE];   // 40 KB

  const int v = srf_xcd_remap(blockIdx.x, total);
  const int mt = v % nMt;
  const int lt = (v / nMt) % nLt;
  const long b = v / (nMt * nLt);
  const int m0 = mt * W4_BM, l0 = lt * W4_BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // 2 x 2 wavefronts, 32 x 32 each

  float mean = 0.f, rstd = 1.f, slope = 1.f;
  if (PRO == 1 || PRO == 2) srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
  if (PRO == 2 || PRO == 3) slope = a.nrm.prelu[0];

  const int Cin = a.Cin, L = a.L, Cout = a.Cout;
  const int nk = Cin / W4_BK;   // even (host checks Cin % 64 == 0)

  // A (weights [m][k]): thread -> row tid >> 2, 8-k packet tid & 3 (rows beyond Cout read 0 from the descriptor)
  const int a_m = tid >> 2, a_pk = tid & 3;
  const int a_lds = a_m * W4_PITCH + a_pk * 16;
  // B (X_b [k][time]): thread -> time step tid & 63, k-group = wavefront (8 k rows each)
  const int b_n = tid & 63, b_kg = wave * 8;
  const int b_lds = b_n * W4_PITCH + b_kg * 2;
  __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, Cout * Cin * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.Bt * Cin * L * 4, 0x00020000);
  const int a_vo = ((m0 + a_m) * Cin + a_pk * 8) * 4;
  const int b_vo = (((int)b * Cin + b_kg) * L + min(l0 + b_n, L - 1)) * 4;   // clamped column: never stored

  struct Regs {
    float4 a[2];
    float b[8];
  };
  Regs r0, r1;
  auto gload = [&](Regs& r, int k0) {
    const auto w0 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo, k0 * 4, 0);
    const auto w1 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo, k0 * 4 + 16, 0);
    r.a[0] = make_float4(__uint_as_float(w0[0]), __uint_as_float(w0[1]), __uint_as_float(w0[2]), __uint_as_float(w0[3]));
    r.a[1] = make_float4(__uint_as_float(w1[0]), __uint_as_float(w1[1]), __uint_as_float(w1[2]), __uint_as_float(w1[3]));
#pragma unroll
    for (int j = 0; j < 8; ++j) r.b[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, b_vo, (k0 + j) * L * 4, 0));
  };
  auto lds_store = [&](const Regs& r, int stage, int k0) {
    char* base = smem + stage * W4_STAGE;
    const float va[8] = {r.a[0].x, r.a[0].y, r.a[0].z, r.a[0].w, r.a[1].x, r.a[1].y, r.a[1].z, r.a[1].w};
    bf16x8 hi, lo;
    w4_split8(va, hi, lo);
    float vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x0 = r.b[j];
      if (PRO == 1 || PRO == 2) {
        const int k = k0 + b_kg + j;
        const float sc = a.nrm.gamma[k] * rstd;
        x0 = fmaf(x0, sc, a.nrm.beta[k] - mean * sc);
      }
      if (PRO == 2 || PRO == 3) x0 = srf_prelu(x0, slope);
      vb[j] = x0;
    }
    bf16x8 hib, lob;
    w4_split8(vb, hib, lob);
    *reinterpret_cast<bf16x8*>(base + 0 * W4_IMG + a_lds) = hi;
    *reinterpret_cast<bf16x8*>(base + 1 * W4_IMG + a_lds) = lo;
    *reinterpret_cast<bf16x8*>(base + 2 * W4_IMG + b_lds) = hib;
    *reinterpret_cast<bf16x8*>(base + 3 * W4_IMG + b_lds) = lob;
  };

  f32x16 acc = {0};
  const int frag = (lane & 31) * W4_PITCH + (lane >> 5) * 16;
  const int a_row = (wm * 32) * W4_PITCH + frag, b_row = (wn * 32) * W4_PITCH + frag;
  auto mma_tile = [&](int stage) {
    const char* base = smem + stage * W4_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks * 32;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + 0 * W4_IMG + a_row + ko);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + 1 * W4_IMG + a_row + ko);
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(base + 2 * W4_IMG + b_row + ko);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(base + 3 * W4_IMG + b_row + ko);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  };
  auto step = [&](Regs& nx, int kt) {
    if (kt + 1 < nk) lds_store(nx, (kt + 1) & 1, (kt + 1) * W4_BK);
    gload(nx, min(kt + 3, nk - 1) * W4_BK);
    mma_tile(kt & 1);
    __syncthreads();
  };

  gload(r0, 0);
  gload(r1, W4_BK);
  // What the epilogue reads -- bias, residual or mask multiplier of this thread's four output rows -- is requested NOW, behind
  // the first operand loads: a block lives for 8-16 k-steps, and four dependent global round trips at its end (the shared
  // strip epilogue's form: load, wait, store, next row) were a third of that.
  const int e_c4 = (lane & 7) * 4, e_l = l0 + wn * 32 + e_c4;
  const bool e_lok = e_l < L;   // L % 4 == 0: the whole float4 is in range
  float e_bias[4];
  float4 e_ext[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 32 + i * 8 + (lane >> 3);
    const int mc = m < Cout ? m : 0;
    e_bias[i] = a.bias[mc];
    e_ext[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.residual)
      e_ext[i] = *reinterpret_cast<const float4*>(a.residual + ((size_t)b * Cout + mc) * L + (e_lok ? e_l : 0));
    else if (a.epi_mask & 1)
      e_ext[i] = *reinterpret_cast<const float4*>(a.mul + ((size_t)b * a.mul_channels + (mc % a.mul_channels)) * L + (e_lok ? e_l : 0));
  }
  lds_store(r0, 0, 0);
  gload(r0, min(2, nk - 1) * W4_BK);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    step(r1, kt);
    step(r0, kt + 1);
  }

  // epilogue: accumulator (MFMA C layout: column lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) -> wave-private strip
  // -> rows x 128-B float4 stores
  float s = 0.f, q = 0.f;
  float* strip = reinterpret_cast<float*>(smem) + wave * (32 * SRF_EPI_PITCH_H);
  {
    const int col = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * kh) * SRF_EPI_PITCH_H + col] = acc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 8 + (lane >> 3);
      const int m = m0 + wm * 32 + row;
      float4 o = *reinterpret_cast<const float4*>(strip + row * SRF_EPI_PITCH_H + e_c4);
      const float bs = e_bias[i];
      o.x += bs; o.y += bs; o.z += bs; o.w += bs;
      const float4 e = e_ext[i];
      if (a.residual) {
        o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
      } else if (a.epi_mask & 1) {
        o.x = fmaxf(o.x, 0.f) * e.x;
        o.y = fmaxf(o.y, 0.f) * e.y;
        o.z = fmaxf(o.z, 0.f) * e.z;
        o.w = fmaxf(o.w, 0.f) * e.w;
      }
      if (e_lok && m < Cout) {
        *reinterpret_cast<float4*>(a.y + ((size_t)b * Cout + m) * L + e_l) = o;
        s += (o.x + o.y) + (o.z + o.w);
        q = fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, fmaf(o.w, o.w, q))));
      }
    }
  }
  if (a.out_sums) {
    __syncthreads();
    srf_block_stats_atomic<4>((double)s, (double)q, srf_stat_slot(a.out_sums, b, v),
                              reinterpret_cast<double*>(smem + 4 * 32 * SRF_EPI_PITCH_H * 4));
  }
}

// Worth it when 128 x 128 tiles would leave MOST CUs idle: four times the blocks, a quarter of the MFMA work each -- but twice the
// operand staging (split + ds_write) per product, which is what these kernels are bound on once the chip is busy.  Measured
// (profiles/r03_NOTES.md): 50-100 tiles of 128 x 128 (batch 1-2): 19-21 vs 26 us per launch; 200 tiles (res_conv at batch 4):
// 37 vs 30 us.  Hence: at most one 128 x 128 tile per two CUs.  Debug flag 2048 (the one-tile-per-block 128 x 128 kernel
// everywhere) switches it off for A/B.
bool srf_pw_w4_wanted(const PwArgs& a) {
  if (srf_debug_flags() & 2048) return false;
  if (a.Cin % 64 || a.L % 4 || (long)a.Bt * a.Cin * a.L * 4 >= (1L << 31) || (long)a.Cout * a.Cin * 4 >= (1L << 31)) return false;
  const long tiles128 = (long)a.Bt * ((a.Cout + 127) / 128) * ((a.L + 127) / 128);
  return 2 * tiles128 <= srf_device_cus();
}

int srf_pw_w4_launch(const PwArgs& a, int pro, hipStream_t st) {
  const int nMt = (a.Cout + W4_BM - 1) / W4_BM, nLt = (a.L + W4_BN - 1) / W4_BN;
  const long total = (long)a.Bt * nMt * nLt;
  SRF_CHECK_ARG(total < (1L << 31), "srf_pw_conv: too many tiles");
  dim3 grid((unsigned)total), block(256);
  switch (pro) {
    case 0: hipLaunchKernelGGL(srf_pw_w4_kernel<0>, grid, block, 0, st, a, nMt, nLt, (int)total); break;
    case 1: hipLaunchKernelGGL(srf_pw_w4_kernel<1>, grid, block, 0, st, a, nMt, nLt, (int)total); break;
    case 2: hipLaunchKernelGGL(srf_pw_w4_kernel<2>, grid, block, 0, st, a, nMt, nLt, (int)total); break;
    default: hipLaunchKernelGGL(srf_pw_w4_kernel<3>, grid, block, 0, st, a, nMt, nLt, (int)total); break;
  }
  SRF_CHECK_LAUNCH("pw_conv_bf16x3_w4", st);
  return SRF_OK;
}
