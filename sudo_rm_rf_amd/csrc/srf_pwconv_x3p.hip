// K2 -- split-precision MFMA GEMM with PRE-PACKED weights (8-wave 128x128x32 tile).
//
// Same arithmetic, tile and pipeline as srf_pw_bf16x3_w8_kernel (srf_pwconv_bf16x3.hip), but the weights
// are split into bf16 hi|lo and laid out tile-by-tile ONCE per forward (srf_pack_pw_weights, one launch
// for all 1x1 convolutions of the model), so the A operand is a plain 16-B global->LDS copy: the
// per-k-tile VALU work of a wave drops from ~90 to ~55 instructions and there is no row mask.
// (A first packed variant with a 256x128 tile / one block per CU measured 220 us vs 152 us and was
// dropped: lock-step waves cannot hide each other's conversion phase.)
#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int P_BM = 128, P_BN = 128, P_BK = 32;
// packed weight image of one (m-tile, k-tile): [part hi|lo][128 rows][32 bf16] = 2 * 128 * 64 B
constexpr int P_WTILE_BYTES = 2 * P_BM * 64;    // 16384

// ---------------------------------------------------------------------------------------------
// weight packing: W[Cout][Cin] fp32 -> bf16 hi/lo tiles, rows >= Cout zero-filled
// ---------------------------------------------------------------------------------------------
#define SRF_MAX_PACK 96
struct PackEntry {
  const float* w;
  char* dst;
  int Cout, Cin;
};
struct PackTable {
  PackEntry e[SRF_MAX_PACK];
};

__global__ __launch_bounds__(256) void srf_pack_pw_weights_kernel(PackTable t) {
  const PackEntry e = t.e[blockIdx.y];
  const int nKt = e.Cin / P_BK;
  const int nMt = (e.Cout + P_BM - 1) / P_BM;
  // one thread per 8-k packet: packet id -> (mt, kt, row, pk)
  const long total = (long)nMt * nKt * P_BM * 4;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int pk = (int)(id & 3);
    const int row = (int)((id >> 2) % P_BM);
    const long tile = (id >> 2) / P_BM;
    const int kt = (int)(tile % nKt), mt = (int)(tile / nKt);
    const int m = mt * P_BM + row;
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (m < e.Cout) ? e.w[(size_t)m * e.Cin + kt * P_BK + pk * 8 + j] : 0.f;
      const __bf16 h = (__bf16)v;
      hi[j] = h;
      lo[j] = (__bf16)(v - (float)h);
    }
    char* base = e.dst + (size_t)tile * P_WTILE_BYTES + row * 64 + pk * 16;
    *reinterpret_cast<bf16x8*>(base) = hi;
    *reinterpret_cast<bf16x8*>(base + P_BM * 64) = lo;
  }
}

size_t srf_x3p_packed_bytes(int Cout, int Cin) {
  const size_t nMt = (Cout + P_BM - 1) / P_BM, nKt = Cin / P_BK;
  return nMt * nKt * (size_t)P_WTILE_BYTES;
}

bool srf_x3p_supported(int Cin, int Cout, int L) {
  return (Cin % 64 == 0) && Cin >= 64 && (L % 4 == 0) && Cout >= 64;
}

// Packs up to SRF_MAX_PACK weight matrices in ONE launch.
int srf_x3p_pack_launch(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n,
                        hipStream_t st) {
  for (int base = 0; base < n; base += SRF_MAX_PACK) {
    PackTable t;
    const int cnt = (n - base) < SRF_MAX_PACK ? (n - base) : SRF_MAX_PACK;
    for (int i = 0; i < SRF_MAX_PACK; ++i) {
      const int j = base + (i < cnt ? i : 0);
      t.e[i] = PackEntry{w[j], dst[j], Cout[j], Cin[j]};
    }
    hipLaunchKernelGGL(srf_pack_pw_weights_kernel, dim3(64, cnt), dim3(256), 0, st, t);
    SRF_CHECK_LAUNCH("pack_pw_weights", st);
  }
  return SRF_OK;
}

// ---------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // first-class 16-B value (HIP uint4 is a struct)

__device__ __forceinline__ void srf_split8p(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

constexpr int P_PITCH_B = 80;
constexpr int P_IMG = P_BM * P_PITCH_B;     // one [128][32] bf16 image, 80-B row pitch
constexpr int P_STAGE_B = 4 * P_IMG;        // A_hi, A_lo, B_hi, B_lo

// PRO: 0 = identity, 1 = GlobLN, 2 = GlobLN + PReLU, 3 = PReLU only
template <int PRO>
__global__ __launch_bounds__(512, 4) void srf_pw_x3p_kernel(PwArgs a, const char* __restrict__ wpack, int nMt,
                                                            int nLt, int total) {
  __shared__ __attribute__((aligned(16))) char smem[2 * P_STAGE_B];   // exactly 80 KB -> 2 blocks / CU

  const int v = srf_xcd_remap(blockIdx.x, total);
  const int mt = v % nMt;
  const int lt = (v / nMt) % nLt;
  const long b = v / (nMt * nLt);
  const int m0 = mt * P_BM, l0 = lt * P_BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 waves, 32 x 64 each

  float mean = 0.f, rstd = 1.f, slope = 1.f;
  if (PRO == 1 || PRO == 2) srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
  if (PRO == 2 || PRO == 3) slope = a.nrm.prelu[0];

  const int Cin = a.Cin, L = a.L;
  const int nk = Cin / P_BK;   // even (host checks Cin % 64 == 0)
  const float* xb = a.x + (size_t)b * Cin * L;

  // A: packed tile (mt, kt) = 1024 x 16-B pieces in [part][row][4] order; thread -> pieces tid, tid + 512
  //    piece p: part = p >> 9, row = (p >> 2) & 127, chunk = p & 3  ->  i = 0: hi image, i = 1: lo image
  const char* a_src = wpack + (size_t)mt * nk * P_WTILE_BYTES + tid * 16;
  const int a_lds = (tid >> 2) * P_PITCH_B + (tid & 3) * 16;
  // B: time step n = tid & 127, k-group kg = tid >> 7 (wave-uniform), 8 k rows x 1 time step
  const int b_n = tid & 127, b_kg = (wave >> 1) * 8;
  const bool b_ok = (l0 + b_n) < L;
  const float* b_src = xb + (size_t)b_kg * L + (b_ok ? (l0 + b_n) : 0);   // clamped; never stored if !ok
  const int b_lds = b_n * P_PITCH_B + b_kg * 2;

  struct Regs {
    u32x4 a0, a1;   // (clang ext-vectors: HIP's uint4 struct ended up in scratch memory here)
    float b[8];
  };
  Regs r0, r1;
  auto gload = [&](Regs& r, int kt) {
    const char* ap = a_src + (size_t)kt * P_WTILE_BYTES;
    r.a0 = *reinterpret_cast<const u32x4*>(ap);
    r.a1 = *reinterpret_cast<const u32x4*>(ap + 8192);
    const float* bp = b_src + (size_t)kt * P_BK * L;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.b[j] = bp[(size_t)j * L];
  };
  auto lds_store = [&](const Regs& r, int stage, int kt) {
    char* base = smem + stage * P_STAGE_B;
    *reinterpret_cast<u32x4*>(base + 0 * P_IMG + a_lds) = r.a0;
    *reinterpret_cast<u32x4*>(base + 1 * P_IMG + a_lds) = r.a1;
    float vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x0 = r.b[j];
      if (PRO == 1 || PRO == 2) {
        const int k = kt * P_BK + b_kg + j;
        const float sc = a.nrm.gamma[k] * rstd;
        x0 = fmaf(x0, sc, a.nrm.beta[k] - mean * sc);
      }
      if (PRO == 2 || PRO == 3) x0 = srf_prelu(x0, slope);
      vb[j] = x0;
    }
    bf16x8 hi, lo;
    srf_split8p(vb, hi, lo);
    *reinterpret_cast<bf16x8*>(base + 2 * P_IMG + b_lds) = hi;
    *reinterpret_cast<bf16x8*>(base + 3 * P_IMG + b_lds) = lo;
  };

  f32x16 acc0 = {0}, acc1 = {0};
  const int frag = (lane & 31) * P_PITCH_B + (lane >> 5) * 16;
  const int a_row = (wm * 32) * P_PITCH_B + frag;
  const int b_row0 = (wn * 64) * P_PITCH_B + frag, b_row1 = b_row0 + 32 * P_PITCH_B;
  auto mma_tile = [&](int stage) {
    const char* base = smem + stage * P_STAGE_B;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks * 32;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + 0 * P_IMG + a_row + ko);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + 1 * P_IMG + a_row + ko);
      const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(base + 2 * P_IMG + b_row0 + ko);
      const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(base + 2 * P_IMG + b_row1 + ko);
      const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(base + 3 * P_IMG + b_row0 + ko);
      const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(base + 3 * P_IMG + b_row1 + ko);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc1, 0, 0, 0);
    }
  };
  auto step = [&](Regs& nx, int kt) {
    if (kt + 1 < nk) lds_store(nx, (kt + 1) & 1, kt + 1);
    gload(nx, min(kt + 3, nk - 1));
    mma_tile(kt & 1);
    __syncthreads();
  };

  gload(r0, 0);
  gload(r1, 1);
  lds_store(r0, 0, 0);
  gload(r0, min(2, nk - 1));
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    step(r1, kt);
    step(r0, kt + 1);
  }

  float s = 0.f, q = 0.f;
  float* strip = reinterpret_cast<float*>(smem) + wave * (32 * SRF_EPI_PITCH);
  srf_pw_epilogue_strip(a, acc0, acc1, strip, b, m0 + wm * 32, l0 + wn * 64, lane, s, q);
  __syncthreads();
  if (a.out_sums)
    srf_block_stats_atomic<8>((double)s, (double)q, srf_stat_slot(a.out_sums, b, v),
                              reinterpret_cast<double*>(smem));
}

int srf_pw_x3p_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st) {
  const int nMt = (a.Cout + P_BM - 1) / P_BM, nLt = (a.L + P_BN - 1) / P_BN;
  const long total = (long)a.Bt * nMt * nLt;
  SRF_CHECK_ARG(total < (1L << 31), "srf_pw_conv: too many tiles");
  dim3 grid((unsigned)total), block(512);
  switch (pro) {
    case 0: hipLaunchKernelGGL(srf_pw_x3p_kernel<0>, grid, block, 0, st, a, wpack, nMt, nLt, (int)total); break;
    case 1: hipLaunchKernelGGL(srf_pw_x3p_kernel<1>, grid, block, 0, st, a, wpack, nMt, nLt, (int)total); break;
    case 2: hipLaunchKernelGGL(srf_pw_x3p_kernel<2>, grid, block, 0, st, a, wpack, nMt, nLt, (int)total); break;
    default: hipLaunchKernelGGL(srf_pw_x3p_kernel<3>, grid, block, 0, st, a, wpack, nMt, nLt, (int)total); break;
  }
  SRF_CHECK_LAUNCH("pw_conv_x3p", st);
  return SRF_OK;
}
