// K2 (main path) -- split-precision MFMA GEMM with PRE-PACKED weights.
//
// Same arithmetic as srf_pwconv_bf16x3.hip (x = bf16 hi + bf16 lo, three v_mfma_f32_32x32x16_bf16 per
// product block, fp32 accumulate; whole-model error vs the reference <= 4e-6, bar 1e-4) but organised
// around what the PMC counters of that kernel showed (SQ_VALU_MFMA_BUSY 21 %, 200 VALU instructions
// per 24 MFMAs per wave, the split/convert work repeated for every 128-row M tile):
//   * weights are split and laid out ONCE per forward by srf_pack_pw_weights (bf16 hi|lo images in
//     exactly the LDS row order), so the A operand is a plain 16-B global->LDS copy: no VALU;
//   * block tile 256(M) x 128(time) x 32(K), 8 wavefronts (4 along M x 2 along time, 64x64 each):
//     the activation tile is converted once per 256 output rows (half / quarter of the VALU work);
//   * two register sets, loads issued two k-tiles ahead, all loads unconditional (clamped) so the
//     compiler's vmcnt bookkeeping stays exact;
//   * conversion of tile t+1 and the MFMAs of tile t sit in ONE basic block with
//     sched_group_barrier hints so the matrix pipe runs under the VALU work of the same wave (the 8
//     waves of the single resident block move in lock-step, so there is no other wave to hide it).
// LDS: A hi|lo 2 x 256 x 80 B + B hi|lo 2 x 128 x 80 B = 60 KB per stage, 2 stages = 120 KB, 1 block/CU.
#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int P_BM = 256, P_BN = 128, P_BK = 32;
constexpr int P_PITCH = 80;                     // LDS row pitch in bytes (64 B data + 16 B pad)
constexpr int P_AIMG = P_BM * P_PITCH;          // 20480
constexpr int P_BIMG = P_BN * P_PITCH;          // 10240
constexpr int P_STAGE = 2 * P_AIMG + 2 * P_BIMG;  // 61440
// packed weight image of one (m-tile, k-tile): [part hi|lo][256 rows][32 bf16] = 2 * 256 * 64 B
constexpr int P_WTILE_BYTES = 2 * P_BM * 64;    // 32768

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
  return (Cin % 64 == 0) && Cin >= 64 && (L % 4 == 0) && Cout >= 192;
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
__device__ __forceinline__ void srf_split8p(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

// PRO: 0 = identity, 1 = GlobLN, 2 = GlobLN + PReLU, 3 = PReLU only
template <int PRO, bool SCHED>
__global__ __launch_bounds__(512, 2) void srf_pw_x3p_kernel(PwArgs a, const char* __restrict__ wpack, int nMt,
                                                            int nLt, int total) {
  __shared__ __attribute__((aligned(16))) char smem[2 * P_STAGE];   // 120 KB

  const int v = srf_xcd_remap(blockIdx.x, total);
  const int mt = v % nMt;
  const int lt = (v / nMt) % nLt;
  const long b = v / (nMt * nLt);
  const int m0 = mt * P_BM, l0 = lt * P_BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  float mean = 0.f, rstd = 1.f, slope = 1.f;
  if (PRO == 1 || PRO == 2) srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
  if (PRO == 2 || PRO == 3) slope = a.nrm.prelu[0];

  const int Cin = a.Cin, L = a.L;
  const int nk = Cin / P_BK;  // even: host checks Cin % 64 == 0
  const float* xb = a.x + (size_t)b * Cin * L;

  // ---- staging assignment
  // A: packed tile (mt, kt) = 2048 x 16-B pieces in [part][row][4] order; thread -> piece tid + 512*i
  const char* a_src = wpack + (size_t)mt * nk * P_WTILE_BYTES + tid * 16;
  //    LDS offset of piece p = tid + 512*i: part = p >> 10, row = (p >> 2) & 255, chunk = p & 3
  const int a_lds = ((tid >> 2) & 255) * P_PITCH + (tid & 3) * 16;   // (+ 128 rows for odd i, + image for part)
  // B: thread -> time step n = tid & 127, k-group kg = tid >> 7 (wave-uniform): 8 k rows x 1 time step
  const int b_n = tid & 127;
  const int b_kg = (wave >> 1) * 8;
  const bool b_ok = (l0 + b_n) < L;
  const float* b_src = xb + (size_t)b_kg * L + (b_ok ? (l0 + b_n) : 0);   // clamped: never stored if !ok
  const int b_lds = b_n * P_PITCH + b_kg * 2;

  struct Regs {
    uint4 a[4];
    float b[8];
  };
  Regs r0, r1;
  auto gload = [&](Regs& r, int kt) {
    const char* ap = a_src + (size_t)kt * P_WTILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.a[i] = *reinterpret_cast<const uint4*>(ap + i * 8192);
    const float* bp = b_src + (size_t)kt * P_BK * L;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.b[j] = bp[(size_t)j * L];
  };
  auto lds_store = [&](const Regs& r, int stage, int kt) {
    char* base = smem + stage * P_STAGE;
    // pieces i = 0..3: p = tid + 512 i -> part = i >> 1, row = (tid >> 2) + 128 * (i & 1)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(base + (i >> 1) * P_AIMG + (i & 1) * 128 * P_PITCH + a_lds) = r.a[i];
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
    *reinterpret_cast<bf16x8*>(base + 2 * P_AIMG + b_lds) = hi;
    *reinterpret_cast<bf16x8*>(base + 2 * P_AIMG + P_BIMG + b_lds) = lo;
  };

  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  const int frag = (lane & 31) * P_PITCH + (lane >> 5) * 16;
  const int a_row0 = (wm * 64) * P_PITCH + frag, a_row1 = a_row0 + 32 * P_PITCH;
  const int b_row0 = (wn * 64) * P_PITCH + frag, b_row1 = b_row0 + 32 * P_PITCH;
  auto mma_tile = [&](int stage) {
    const char* base = smem + stage * P_STAGE;
    const char* bb = base + 2 * P_AIMG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks * 32;
      const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(base + a_row0 + ko);
      const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(base + a_row1 + ko);
      const bf16x8 al0 = *reinterpret_cast<const bf16x8*>(base + P_AIMG + a_row0 + ko);
      const bf16x8 al1 = *reinterpret_cast<const bf16x8*>(base + P_AIMG + a_row1 + ko);
      const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(bb + b_row0 + ko);
      const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(bb + b_row1 + ko);
      const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(bb + P_BIMG + b_row0 + ko);
      const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(bb + P_BIMG + b_row1 + ko);
      acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh1, acc11, 0, 0, 0);
      acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl1, acc11, 0, 0, 0);
      acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh1, acc11, 0, 0, 0);
    }
  };
  // One pipeline step for tile kt (stage kt&1).  `nx` holds tile kt+1 (loaded two steps ago): it is
  // converted/stored into the other stage, then re-armed with tile kt+3.  Everything is unconditional
  // (indices clamped; a surplus store lands in the stage nobody reads any more) so that the step is ONE
  // basic block the scheduler can interleave: MFMA pipe under the VALU/LDS work of the same wave.
  auto step = [&](Regs& nx, int kt) {
    lds_store(nx, (kt + 1) & 1, min(kt + 1, nk - 1));
    gload(nx, min(kt + 3, nk - 1));
    mma_tile(kt & 1);
    if (SCHED) {
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
        if (i % 4 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 DS write
        if (i % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
      }
    }
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

  // ---- epilogue: per-wave LDS strips (8 waves x 2 strips x 32 x 68 floats = 136 KB?  no: one strip
  // per wave used twice = 8 x 8704 B = 68 KB, inside the dead operand stages)
  float s = 0.f, q = 0.f;
  const int mb = m0 + wm * 64, lb = l0 + wn * 64;
  float* strip = reinterpret_cast<float*>(smem) + wave * (32 * SRF_EPI_PITCH);
  srf_pw_epilogue_strip(a, acc00, acc01, strip, b, mb, lb, lane, s, q);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  srf_pw_epilogue_strip(a, acc10, acc11, strip, b, mb + 32, lb, lane, s, q);
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
#define X3P_LAUNCH(P, S) \
  hipLaunchKernelGGL((srf_pw_x3p_kernel<P, S>), grid, block, 0, st, a, wpack, nMt, nLt, (int)total)
  const bool sched = (srf_debug_flags() & 1) == 0;
  switch (pro * 2 + (sched ? 1 : 0)) {
    case 0: X3P_LAUNCH(0, false); break;
    case 1: X3P_LAUNCH(0, true); break;
    case 2: X3P_LAUNCH(1, false); break;
    case 3: X3P_LAUNCH(1, true); break;
    case 4: X3P_LAUNCH(2, false); break;
    case 5: X3P_LAUNCH(2, true); break;
    case 6: X3P_LAUNCH(3, false); break;
    default: X3P_LAUNCH(3, true); break;
  }
#undef X3P_LAUNCH
  SRF_CHECK_LAUNCH("pw_conv_x3p", st);
  return SRF_OK;
}
