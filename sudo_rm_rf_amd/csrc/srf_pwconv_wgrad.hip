// Weight gradient of the pointwise convs (training step):
//     dW[m, n] = sum_b sum_l G[b, m, l] * f(X[b, n, l])        dbias[m] = sum_b sum_l G[b, m, l]
// for y = W f(x) + bias (improved_sudormrf.py:174,196,256-259,268-269), f = the same GlobLN / PReLU prologue the
// forward GEMM applies on load (it is never materialised, so it is re-applied here).
// A GEMM with a tiny output (Cout x Cin) and a huge reduction (Bt * L = 102400 for cfg 2): both operands are
// [row][time] with the reduction index contiguous, which is exactly what v_mfma_f32_32x32x16_bf16 wants (8
// consecutive k per lane) -- no transposition while staging, unlike the forward.  Same split-bf16 scheme as the
// forward (hi + lo, three MFMAs per product, fp32 accumulate), same 128x128x32 tile / 8 wavefronts / LDS images.
// Split-K: the (example, time-chunk) pieces are dealt round-robin to P blocks per output tile, each block
// keeps its 128x128 accumulator over all its pieces and writes ONE partial tile; a second kernel sums the P
// partials (deterministic, no atomics).
#include <type_traits>

#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_BM = 128, WG_BN = 128, WG_BK = 32;
constexpr int WG_PITCH = 80;                  // bytes per LDS row (32 bf16 + 16 B pad): conflict-free b128 reads
constexpr int WG_IMG = WG_BM * WG_PITCH;
constexpr int WG_STAGE = 4 * WG_IMG;          // G_hi, G_lo, X_hi, X_lo

struct WgArgs {
  const float* g;      // [Bt][M][L]
  const float* x;      // [Bt][N][L]
  float* part;         // [P][M][N]
  float* bias_part;    // [P][M] or null
  SrfNormDev nrm;      // prologue on x (per example statistics, per channel n affine)
  double inv_count;
  int M, N, L, Bt;
  int kc_len, nKc, P;  // time-chunk length (multiple of 32), chunks per example, partials
  int nMt, nNt;
  int xcd_map;         // block -> (tile, partial) mapping that puts ALL output tiles of a partial on one XCD (P % 8 == 0)
};

__device__ __forceinline__ void wg_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

// PRO: 0 identity, 1 GlobLN, 2 GlobLN + PReLU, 3 PReLU
// FULL (round 6): M and N multiples of 128 and every time chunk a whole number of 32-wide k-tiles (L % 32 == 0) -- no row / tail masks.
// The kernel is VALU-issue-bound, not MFMA- or HBM-bound (r06 counters: 13.1 VALU instructions per MFMA, VALU issue 64 % of the kernel's
// cycles, MFMA pipe 39 %): the masks were 16 multiplies + 16 address selects of the 136 VALU instructions per k-tile.
template <int PRO, bool FULL>
__global__ __launch_bounds__(512, 4) void srf_pw_wgrad_kernel(WgArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * WG_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = a.nMt * a.nNt;
  // Every output tile of partial p streams the SAME (example, time-chunk) pieces: G rows are shared by the nNt tiles of an
  // m-tile, X rows by the nMt tiles of an n-tile.  Blocks go to XCDs round-robin by blockIdx, so the round-3 mapping
  // (tile = blockIdx % ntiles; with 8 tiles: tile t lives on XCD t) made every XCD read its rows through its own L2 -- the
  // operands crossed the fabric (nNt M + nMt N) / (M + N) times: 2.7 x for 256 x 512.  Round 4: XCD x
  // owns the partials p = x (mod 8), its co-resident blocks walk the tiles of one partial together, and the re-reads hit L2.
  int tile, p;
  if (a.xcd_map) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % ntiles;
    p = (j / ntiles) * 8 + xcd;
  } else {
    tile = blockIdx.x % ntiles;
    p = blockIdx.x / ntiles;
  }
  const int m0 = (tile % a.nMt) * WG_BM, n0 = (tile / a.nMt) * WG_BN;
  const int L = a.L, M = a.M, N = a.N;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;

  // staging: thread -> row r = tid>>2 of BOTH tiles, 8-k packet pk = tid&3 (2 float4 each)
  const int r = tid >> 2, pk = tid & 3;
  const bool g_ok = FULL || (m0 + r) < M, x_ok = FULL || (n0 + r) < N;
  const int gm = g_ok ? m0 + r : 0, xn = x_ok ? n0 + r : 0;
  const float g_msk = g_ok ? 1.f : 0.f, x_msk = x_ok ? 1.f : 0.f;
  const int lds_off = r * WG_PITCH + pk * 16;
  float gam = 1.f, bet = 0.f;
  if (PRO == 1 || PRO == 2) {
    gam = a.nrm.gamma[xn];
    bet = a.nrm.beta[xn];
  }

  f32x16 acc0 = {0}, acc1 = {0};
  float bsum = 0.f;
  const int frag = (lane & 31) * WG_PITCH + (lane >> 5) * 16;
  const int a_row = (wm * 32) * WG_PITCH + frag;
  const int b_row0 = (wn * 64) * WG_PITCH + frag, b_row1 = b_row0 + 32 * WG_PITCH;

  const int nchunks = a.Bt * a.nKc;
  for (int c = p; c < nchunks; c += a.P) {
    const int b = c / a.nKc, kc = c - b * a.nKc;
    const int l_beg = kc * a.kc_len;
    const int l_end = min(l_beg + a.kc_len, L);
    const int nk = (l_end - l_beg + WG_BK - 1) / WG_BK;
    float sc = 1.f, sh = 0.f;
    if (PRO == 1 || PRO == 2) {
      float mean, rstd;
      srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
      sc = gam * rstd;
      sh = bet - mean * sc;
    }
    const float* gsrc = a.g + ((size_t)b * M + gm) * L + l_beg + pk * 8;
    const float* xsrc = a.x + ((size_t)b * N + xn) * L + l_beg + pk * 8;

    float4 rg[2], rx[2];
    float km[2];   // 1 inside the chunk, 0 beyond its end (L % 4 == 0: a float4 is entirely in or out)
    auto gload_g = [&](int kt) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if constexpr (FULL) {
          rg[h] = *reinterpret_cast<const float4*>(gsrc + kt * WG_BK + h * 4);
        } else {
          const int o = kt * WG_BK + h * 4;
          const bool in = l_beg + pk * 8 + o < l_end;
          rg[h] = *reinterpret_cast<const float4*>(gsrc + (in ? o : 0));
          km[h] = in ? 1.f : 0.f;
        }
      }
    };
    auto gload_x = [&](int kt) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if constexpr (FULL) {
          rx[h] = *reinterpret_cast<const float4*>(xsrc + kt * WG_BK + h * 4);
        } else {
          const int o = kt * WG_BK + h * 4;
          const bool in = l_beg + pk * 8 + o < l_end;
          rx[h] = *reinterpret_cast<const float4*>(xsrc + (in ? o : 0));
        }
      }
    };
    // the loaded packets, prologue applied, split into the stage's images -- G and X separately (8 live result registers, not 16)
    auto put_g = [&](int stage) {
      float vg[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float gg[4] = {rg[h].x, rg[h].y, rg[h].z, rg[h].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) vg[4 * h + j] = FULL ? gg[j] : gg[j] * (km[h] * g_msk);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum += vg[j];
      bf16x8 hi, lo;
      wg_split8(vg, hi, lo);
      char* base = smem + stage * WG_STAGE + lds_off;
      *reinterpret_cast<bf16x8*>(base + 0 * WG_IMG) = hi;
      *reinterpret_cast<bf16x8*>(base + 1 * WG_IMG) = lo;
    };
    auto put_x = [&](int stage, float kmx0, float kmx1) {
      float vx[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float xx[4] = {rx[h].x, rx[h].y, rx[h].z, rx[h].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = xx[j];
          if (PRO == 1 || PRO == 2) v = fmaf(v, sc, sh);
          if (PRO == 2 || PRO == 3) v = srf_prelu(v, slope);
          vx[4 * h + j] = FULL ? v : v * ((h ? kmx1 : kmx0) * x_msk);
        }
      }
      bf16x8 hi, lo;
      wg_split8(vx, hi, lo);
      char* base = smem + stage * WG_STAGE + lds_off;
      *reinterpret_cast<bf16x8*>(base + 2 * WG_IMG) = hi;
      *reinterpret_cast<bf16x8*>(base + 3 * WG_IMG) = lo;
    };
    auto mma_tile = [&](int stage) {
      const char* base = smem + stage * WG_STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ko = ks * 32;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + 0 * WG_IMG + a_row + ko);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + 1 * WG_IMG + a_row + ko);
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(base + 2 * WG_IMG + b_row0 + ko);
        const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(base + 2 * WG_IMG + b_row1 + ko);
        const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(base + 3 * WG_IMG + b_row0 + ko);
        const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(base + 3 * WG_IMG + b_row1 + ko);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc1, 0, 0, 0);
      }
    };

    auto mma_half = [&](int stage, int ks) {
      const char* base = smem + stage * WG_STAGE;
      const int ko = ks * 32;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + 0 * WG_IMG + a_row + ko);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + 1 * WG_IMG + a_row + ko);
      const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(base + 2 * WG_IMG + b_row0 + ko);
      const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(base + 2 * WG_IMG + b_row1 + ko);
      const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(base + 3 * WG_IMG + b_row0 + ko);
      const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(base + 3 * WG_IMG + b_row1 + ko);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc1, 0, 0, 0);
    };

    // one register set, LDS double-buffered: tile kt+1 is split into the other stage and tile kt+2's loads
    // are in flight while tile kt's MFMAs run.  Round 6: the steady state is ONE basic block (no kt + 1 / kt + 2 conditionals): the first
    // half k-tile's MFMAs with G's split in their shadow, the second half's with X's, each operand's next loads issued as soon as
    // its registers are free.
    gload_g(0);
    gload_x(0);
    __syncthreads();          // previous chunk's last MFMAs are done with both stages
    {
      const float k0 = km[0], k1 = km[1];
      put_g(0);
      put_x(0, k0, k1);
    }
    if (nk > 1) {
      gload_g(1);
      gload_x(1);
    }
    __syncthreads();
    int kt = 0;
    if constexpr (FULL) {
      // Steady state written out with the order FIXED (sched_barrier(0): nothing moves across): each MFMA (8 passes = 32 cycles of
      // the matrix pipe) is followed by the split of two of tile kt + 1's sixteen values (about 12 VALU instructions), so the
      // wavefront's own VALU work sits in its MFMAs' shadow; left to itself the scheduler issues the MFMAs in a burst and the
      // hundred VALU instructions after it.  G first (its registers are then free for tile kt + 2's loads), then X.
      for (; kt + 2 < nk; ++kt) {
        const int st = kt & 1;
        const char* rb = smem + st * WG_STAGE;
        char* wb = smem + (st ^ 1) * WG_STAGE + lds_off;
        bf16x8 ah, al, bh0, bh1, bl0, bl1, hi, lo;
        auto frags = [&](int ko) {
          ah = *reinterpret_cast<const bf16x8*>(rb + 0 * WG_IMG + a_row + ko);
          al = *reinterpret_cast<const bf16x8*>(rb + 1 * WG_IMG + a_row + ko);
          bh0 = *reinterpret_cast<const bf16x8*>(rb + 2 * WG_IMG + b_row0 + ko);
          bh1 = *reinterpret_cast<const bf16x8*>(rb + 2 * WG_IMG + b_row1 + ko);
          bl0 = *reinterpret_cast<const bf16x8*>(rb + 3 * WG_IMG + b_row0 + ko);
          bl1 = *reinterpret_cast<const bf16x8*>(rb + 3 * WG_IMG + b_row1 + ko);
        };
        auto pair_g = [&](int q) {          // values 2 q, 2 q + 1 of G's packet
          const float v0 = q == 0 ? rg[0].x : q == 1 ? rg[0].z : q == 2 ? rg[1].x : rg[1].z;
          const float v1 = q == 0 ? rg[0].y : q == 1 ? rg[0].w : q == 2 ? rg[1].y : rg[1].w;
          bsum += v0;
          bsum += v1;
          const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
          hi[2 * q] = h0;
          hi[2 * q + 1] = h1;
          lo[2 * q] = (__bf16)(v0 - (float)h0);
          lo[2 * q + 1] = (__bf16)(v1 - (float)h1);
        };
        auto pair_x = [&](int q) {
          float v0 = q == 0 ? rx[0].x : q == 1 ? rx[0].z : q == 2 ? rx[1].x : rx[1].z;
          float v1 = q == 0 ? rx[0].y : q == 1 ? rx[0].w : q == 2 ? rx[1].y : rx[1].w;
          if (PRO == 1 || PRO == 2) {
            v0 = fmaf(v0, sc, sh);
            v1 = fmaf(v1, sc, sh);
          }
          if (PRO == 2 || PRO == 3) {
            v0 = srf_prelu(v0, slope);
            v1 = srf_prelu(v1, slope);
          }
          const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
          hi[2 * q] = h0;
          hi[2 * q + 1] = h1;
          lo[2 * q] = (__bf16)(v0 - (float)h0);
          lo[2 * q + 1] = (__bf16)(v1 - (float)h1);
        };
#define WG_SB() __builtin_amdgcn_sched_barrier(0)
#define WG_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
        frags(0);
        WG_SB();
        acc0 = WG_MFMA(al, bh0, acc0);
        pair_g(0);
        WG_SB();
        acc1 = WG_MFMA(al, bh1, acc1);
        pair_g(1);
        WG_SB();
        acc0 = WG_MFMA(ah, bl0, acc0);
        pair_g(2);
        WG_SB();
        acc1 = WG_MFMA(ah, bl1, acc1);
        pair_g(3);
        WG_SB();
        acc0 = WG_MFMA(ah, bh0, acc0);
        *reinterpret_cast<bf16x8*>(wb + 0 * WG_IMG) = hi;
        *reinterpret_cast<bf16x8*>(wb + 1 * WG_IMG) = lo;
        gload_g(kt + 2);
        WG_SB();
        acc1 = WG_MFMA(ah, bh1, acc1);
        frags(32);
        pair_x(0);
        WG_SB();
        acc0 = WG_MFMA(al, bh0, acc0);
        pair_x(1);
        WG_SB();
        acc1 = WG_MFMA(al, bh1, acc1);
        pair_x(2);
        WG_SB();
        acc0 = WG_MFMA(ah, bl0, acc0);
        pair_x(3);
        WG_SB();
        acc1 = WG_MFMA(ah, bl1, acc1);
        *reinterpret_cast<bf16x8*>(wb + 2 * WG_IMG) = hi;
        *reinterpret_cast<bf16x8*>(wb + 3 * WG_IMG) = lo;
        gload_x(kt + 2);
        WG_SB();
        acc0 = WG_MFMA(ah, bh0, acc0);
        acc1 = WG_MFMA(ah, bh1, acc1);
        WG_SB();
#undef WG_SB
#undef WG_MFMA
        __syncthreads();
      }
    } else {
      for (; kt + 2 < nk; ++kt) {
        const int st = kt & 1, nx = st ^ 1;
        const float k0 = km[0], k1 = km[1];       // (tile kt + 1's masks, before gload_g(kt + 2) overwrites them)
        mma_half(st, 0);
        put_g(nx);
        gload_g(kt + 2);
        mma_half(st, 1);
        put_x(nx, k0, k1);
        gload_x(kt + 2);
        __syncthreads();
      }
    }
    for (; kt < nk; ++kt) {
      if (kt + 1 < nk) {
        const float k0 = km[0], k1 = km[1];
        put_g((kt + 1) & 1);
        put_x((kt + 1) & 1, k0, k1);
      }
      mma_half(kt & 1, 0);
      mma_half(kt & 1, 1);
      __syncthreads();
    }
  }

  // ---- partial tile out: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = a.part + (size_t)p * M * N;
  const int col = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int row = (rr & 3) + 8 * (rr >> 2) + 4 * kh;
    const int m = m0 + wm * 32 + row;
    const int n = n0 + wn * 64 + col;
    if (m < M) {
      if (n < N) out[(size_t)m * N + n] = acc0[rr];
      if (n + 32 < N) out[(size_t)m * N + n + 32] = acc1[rr];
    }
  }
  if (a.bias_part && (tile / a.nMt) == 0) {
    // the 4 threads of a row hold disjoint k packets
    bsum += __shfl_xor(bsum, 1, 64);
    bsum += __shfl_xor(bsum, 2, 64);
    if (pk == 0 && g_ok) a.bias_part[(size_t)p * M + m0 + r] = bsum;
  }
}

// ---------------------------------------------------------------------------------------------
// Round 6: the 256 x 128 (or 128 x 256) tile, ONE block per CU.
// What binds the 128 x 128 kernel above (ablation builds, profiles/r06_wgrad_ablation_and_wide_tile.txt): not the MFMAs (without
// them 95.6 of 99.0 us), not the split's VALU work (96.3), not the barriers or the LDS writes -- the GLOBAL LOADS (without them 75.5 us):
// with 8 tiles of 128 x 128 for a 256 x 512 gradient every G row is fetched four times and every X row twice, 840 MB through L2 for 315 MB
// of operands, by wavefronts that all issue their loads at the same point of a barrier-locked loop.  A tile that spans the whole of the
// smaller operand (BM = 256 for Cout = 256, BN = 256 for Cin = 256) fetches that operand's rows once per output COLUMN block and the
// other operand once: 630 MB; 24 MFMAs per wavefront and k-tile against 12 with 24 instead of 16 values to split (1 value per MFMA
// instead of 1.33) and 0.67 fragment reads per MFMA instead of 1; eight wavefronts with 64 x 64 accumulators (2 per SIMD, up to 256
// VGPRs), the split of tile kt + 1 written out between the MFMAs of tile kt as in the FULL form above.
// Full shapes only (Cout % BM == 0, Cin % BN == 0, L % 32 == 0); same partial-tile output and reduction as the kernel above.
// ---------------------------------------------------------------------------------------------
template <int PRO, int BM, int BN>
__global__ __launch_bounds__(512, 2) void srf_pw_wgrad_wide_kernel(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wsmem[];
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  // Staging in FULL 128-byte lines: eight lanes load the eight float4 of one row's k-tile, a wavefront instruction covers 8 rows x 128 B
  // (the first build had four lanes per row, 16 rows x 64 B per instruction -- twice the address / tag work in the texture path for the
  // same bytes; MI355X guide, "fragment-shaped loads").  Thread -> segment sg = tid & 7 of rows rr + 64 u, u = 0 .. NR - 1, of the
  // concatenated [G tile rows; X tile rows].
  constexpr int NGR = BM / 64, NXR = BN / 64, NR = NGR + NXR;     // rows per thread and k-tile: G rows, X rows (6 in all)
  constexpr int IMG_G = BM * WG_PITCH, IMG_X = BN * WG_PITCH;
  constexpr int OFF_GH = 0, OFF_GL = IMG_G, OFF_XH = 2 * IMG_G, OFF_XL = 2 * IMG_G + IMG_X;
  constexpr int STAGE = 2 * (IMG_G + IMG_X);
  constexpr int WN = BN / 64;                                    // wavefronts along N (8 / WN along M)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntiles = a.nMt * a.nNt;
  int tile, p;
  if (a.xcd_map) {        // XCD x owns the partials p = x (mod 8): the tiles of one partial share their operand rows through one L2
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % ntiles;
    p = (j / ntiles) * 8 + xcd;
  } else {
    tile = blockIdx.x % ntiles;
    p = blockIdx.x / ntiles;
  }
  const int m0 = (tile % a.nMt) * BM, n0 = (tile / a.nMt) * BN;
  const int L = a.L, M = a.M, N = a.N;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;

  const int rr = tid >> 3, sg = tid & 7;
  const int lds_off = rr * WG_PITCH + sg * 8;
  float gam[NXR], bet[NXR];
#pragma unroll
  for (int q = 0; q < NXR; ++q) {
    gam[q] = 1.f;
    bet[q] = 0.f;
    if (PRO == 1 || PRO == 2) {
      gam[q] = a.nrm.gamma[n0 + rr + 64 * q];
      bet[q] = a.nrm.beta[n0 + rr + 64 * q];
    }
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float bsum[NGR];
#pragma unroll
  for (int q = 0; q < NGR; ++q) bsum[q] = 0.f;
  const int frag = (lane & 31) * WG_PITCH + (lane >> 5) * 16;
  const int a_row = (wm * 64) * WG_PITCH + frag, b_row = (wn * 64) * WG_PITCH + frag;

  const int nchunks = a.Bt * a.nKc;
  for (int c = p; c < nchunks; c += a.P) {
    const int b = c / a.nKc, kc = c - b * a.nKc;
    const int l_beg = kc * a.kc_len;
    const int l_end = min(l_beg + a.kc_len, L);
    const int nk = (l_end - l_beg) / WG_BK;
    float sc[NXR], sh[NXR];
#pragma unroll
    for (int q = 0; q < NXR; ++q) {
      sc[q] = 1.f;
      sh[q] = 0.f;
    }
    if (PRO == 1 || PRO == 2) {
      float mean, rstd;
      srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
#pragma unroll
      for (int q = 0; q < NXR; ++q) {
        sc[q] = gam[q] * rstd;
        sh[q] = bet[q] - mean * sc[q];
      }
    }
    const float* gsrc = a.g + ((size_t)b * M + m0) * L + l_beg;
    const float* xsrc = a.x + ((size_t)b * N + n0) * L + l_beg;
    const unsigned tbyte = ((unsigned)rr * (unsigned)L + (unsigned)sg * 4u) * 4u;      // (< 2^31: 64 rows of one example)
    const size_t row64 = (size_t)64 * L;

    // One register set, as in the 128 x 128 kernel: while tile kt's MFMAs run, tile kt + 1 is split out of it into the other LDS stage and
    // its rows are re-loaded with tile kt + 2.  (Two sets -- two k-tiles of loads in flight per thread, 96 KB per CU -- were built and
    // measured: 250-256 VGPRs with 12-29 spilled in the GlobLN forms, 101 us against 93.5 us for one set at the cfg-2 shapes.)
    float4 rv[NR];
    auto gload = [&](int u, int kt) {        // row u of this thread: G rows first, then X rows
      const float* base = u < NGR ? gsrc + u * row64 : xsrc + (u - NGR) * row64;
      rv[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base + kt * WG_BK) + tbyte);
    };
    // values 2 e, 2 e + 1 (e = 0, 1) of row u's float4: prologue, hi / lo split
    auto pair = [&](int u, int e, bf16x4& hi, bf16x4& lo) {
      float v0 = e ? rv[u].z : rv[u].x, v1 = e ? rv[u].w : rv[u].y;
      if (u < NGR) {
        bsum[u < NGR ? u : 0] += v0;
        bsum[u < NGR ? u : 0] += v1;
      } else {
        const int q = u < NGR ? 0 : u - NGR;
        if (PRO == 1 || PRO == 2) {
          v0 = fmaf(v0, sc[q], sh[q]);
          v1 = fmaf(v1, sc[q], sh[q]);
        }
        if (PRO == 2 || PRO == 3) {
          v0 = srf_prelu(v0, slope);
          v1 = srf_prelu(v1, slope);
        }
      }
      const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
      hi[2 * e] = h0;
      hi[2 * e + 1] = h1;
      lo[2 * e] = (__bf16)(v0 - (float)h0);
      lo[2 * e + 1] = (__bf16)(v1 - (float)h1);
    };
    auto store = [&](int u, int stage, const bf16x4& hi, const bf16x4& lo) {
      char* base = wsmem + stage * STAGE + lds_off;
      if (u < NGR) {
        *reinterpret_cast<bf16x4*>(base + OFF_GH + u * 64 * WG_PITCH) = hi;
        *reinterpret_cast<bf16x4*>(base + OFF_GL + u * 64 * WG_PITCH) = lo;
      } else {
        *reinterpret_cast<bf16x4*>(base + OFF_XH + (u - NGR) * 64 * WG_PITCH) = hi;
        *reinterpret_cast<bf16x4*>(base + OFF_XL + (u - NGR) * 64 * WG_PITCH) = lo;
      }
    };
    bf16x8 ah[2], al[2], bh[2], bl[2];
    auto frags = [&](int stage, int ko) {
      const char* rb = wsmem + stage * STAGE;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(rb + OFF_GH + a_row + i * 32 * WG_PITCH + ko);
        al[i] = *reinterpret_cast<const bf16x8*>(rb + OFF_GL + a_row + i * 32 * WG_PITCH + ko);
        bh[i] = *reinterpret_cast<const bf16x8*>(rb + OFF_XH + b_row + i * 32 * WG_PITCH + ko);
        bl[i] = *reinterpret_cast<const bf16x8*>(rb + OFF_XL + b_row + i * 32 * WG_PITCH + ko);
      }
    };
    // MFMA e of a half k-tile (0 .. 11): pass e / 4 = lo x hi, hi x lo, hi x hi (small terms first); accumulator (i, j) = ((e % 4) / 2, e % 2)
    auto mfma = [&](int e) {
      const int pass = e >> 2, i = (e & 3) >> 1, j = e & 1;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pass == 0 ? al[i] : ah[i], pass == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
    };
    // One k-tile with the order FIXED (sched_barrier(0)): the 24 MFMAs of tile kt and, after every second one, the split of two of tile
    // kt + 1's 4 NR values (SPLIT); a row's register takes tile kt + 2's load as soon as its second pair has been split (RELOAD).
    auto step = [&](int kt, auto SPLIT, auto RELOAD) {
      const int st = kt & 1;
      bf16x4 hi, lo;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        frags(st, ks * 32);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
          mfma(2 * s6);
          mfma(2 * s6 + 1);
          const int w = ks * 6 + s6;           // work slot 0 .. 11
          if (decltype(SPLIT)::value && w < 2 * NR) {
            const int u = w >> 1, e = w & 1;
            pair(u, e, hi, lo);
            if (e == 1) {
              store(u, st ^ 1, hi, lo);
              if (decltype(RELOAD)::value) gload(u, kt + 2);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    };
    using Yes = std::true_type;
    using No = std::false_type;

#pragma unroll
    for (int u = 0; u < NR; ++u) gload(u, 0);
    __syncthreads();          // the previous chunk's last MFMAs are done with both stages
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      bf16x4 hi, lo;
      pair(u, 0, hi, lo);
      pair(u, 1, hi, lo);
      store(u, 0, hi, lo);
    }
    if (nk > 1) {
#pragma unroll
      for (int u = 0; u < NR; ++u) gload(u, 1);
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; ++kt) step(kt, Yes{}, Yes{});
    for (; kt < nk; ++kt) {
      if (kt + 1 < nk) step(kt, Yes{}, No{});
      else step(kt, No{}, No{});
    }
  }

  // ---- partial tile out (C/D layout of the 32x32 MFMA: col = lane&31, row = (rr&3) + 8*(rr>>2) + 4*(lane>>5))
  float* out = a.part + (size_t)p * M * N;
  const int col = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * kh;
        out[(size_t)(m0 + wm * 64 + i * 32 + row) * N + n0 + wn * 64 + j * 32 + col] = acc[i][j][q];
      }
  if (a.bias_part && (tile / a.nMt) == 0) {
#pragma unroll
    for (int q = 0; q < NGR; ++q) {         // the 8 threads of a row hold disjoint segments
      float t = bsum[q];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      if (sg == 0) a.bias_part[(size_t)p * M + m0 + rr + 64 * q] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Small-channel variant (GroupComm's per-group convs: Cout, Cin <= 64 with the batch folded to Bt*G = 512): the
// 128x128 MFMA tile above would be 97 % padding and the op is a stream (2*M*N flops against 4*(M+N) bytes per
// column).  A block stages [M+N][LC] columns of one folded example in LDS; thread = (4x4 output tile, column
// slice): per float4 column it reads 4+4 float4 and does 64 FMAs; slices are summed through LDS, blocks write
// partial [M][N] tiles that the same reduction kernel sums.
// ---------------------------------------------------------------------------------------------
template <int PRO>
__global__ __launch_bounds__(256) void srf_pw_wgrad_small_kernel(WgArgs a, int LC, int npiece_l) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  const int M = a.M, N = a.N, L = a.L;
  const int pitch = LC + 4;
  float* sg = wsm;                  // [M][pitch]
  float* sx = wsm + M * pitch;      // [N][pitch]
  const int tilesN = N >> 2, ntile = (M >> 2) * tilesN;
  const int nslice = 256 / ntile;   // host guarantees ntile <= 256; threads beyond nslice * ntile only help staging
  const int tile = threadIdx.x % ntile, slice = threadIdx.x / ntile;
  const bool worker = slice < nslice;
  const int tm = tile / tilesN, tn = tile - tm * tilesN;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;
  float acc[4][4] = {{0.f}};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // a thread's staging rows do not depend on the piece (see below): their GlobLN affine once, in registers
  constexpr int NIT = 12;
  const int nf4 = LC >> 2, sh4 = 31 - __builtin_clz(nf4);      // LC is 64, 128 or 256
  const int rpi = 256 >> sh4, r0 = threadIdx.x >> sh4, c4 = threadIdx.x & (nf4 - 1);
  float gam[NIT], bet[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int n = r0 + i * rpi - M;
    const bool isx = (PRO == 1 || PRO == 2) && n >= 0 && n < N;
    gam[i] = isx ? a.nrm.gamma[n] : 1.f;
    bet[i] = isx ? a.nrm.beta[n] : 0.f;
  }
  const int pieces = a.Bt * npiece_l;
  for (int pc = blockIdx.x; pc < pieces; pc += gridDim.x) {
    const int b = pc / npiece_l, l0 = (pc - b * npiece_l) * LC;
    float mean = 0.f, rstd = 1.f;
    if (PRO == 1 || PRO == 2) srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
    __syncthreads();   // previous piece fully consumed
    // ---- stage: rows of G then rows of X, float4 along time, zero beyond L.  Round 6: ALL of a thread's loads first (<= 12 float4: up to 48,
    // 96 or 128 rows of 256, 128 or 64 columns over 256 threads), then the prologue and the LDS stores -- the rolled loop of rounds 3-5 made
    // twelve dependent round trips to HBM per piece with one float4 in flight per thread (12 KB per CU: 3.3 TB/s by Little's law, which is
    // what it ran at).
    // Element e = threadIdx.x + 256 i sits in row (threadIdx.x >> sh4) + i * (256 >> sh4) and -- 256 being a multiple of the 16, 32 or 64
    // float4 of a row -- always in column c4 = threadIdx.x & (nf4 - 1).
    const int l = l0 + c4 * 4;
    const bool l_ok = l < L;
    const float* gb = a.g + (size_t)b * M * L + l;
    const float* xb = a.x + (size_t)b * N * L + l;
    float4 stg[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row = r0 + i * rpi;
      stg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < M + N && l_ok) {
        const bool isx = row >= M;
        const unsigned idx = (unsigned)(isx ? row - M : row) * (unsigned)L;
        stg[i] = *reinterpret_cast<const float4*>((isx ? xb : gb) + idx);
      }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row = r0 + i * rpi;
      if (row < M + N) {
        float4 v = stg[i];
        if (row >= M && l_ok) {
          if (PRO == 1 || PRO == 2) {
            const float sc = gam[i] * rstd, sh = bet[i] - mean * sc;
            v.x = fmaf(v.x, sc, sh);
            v.y = fmaf(v.y, sc, sh);
            v.z = fmaf(v.z, sc, sh);
            v.w = fmaf(v.w, sc, sh);
          }
          if (PRO == 2 || PRO == 3) {
            v.x = srf_prelu(v.x, slope);
            v.y = srf_prelu(v.y, slope);
            v.z = srf_prelu(v.z, slope);
            v.w = srf_prelu(v.w, slope);
          }
        }
        *reinterpret_cast<float4*>(wsm + row * pitch + c4 * 4) = v;
      }
    }
    __syncthreads();
    if (worker) {
      for (int c4 = slice; c4 < nf4; c4 += nslice) {
        float4 gv[4], xv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gv[r] = *reinterpret_cast<const float4*>(sg + (tm * 4 + r) * pitch + c4 * 4);
          xv[r] = *reinterpret_cast<const float4*>(sx + (tn * 4 + r) * pitch + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j] = fmaf(gv[i].x, xv[j].x, acc[i][j]);
            acc[i][j] = fmaf(gv[i].y, xv[j].y, acc[i][j]);
            acc[i][j] = fmaf(gv[i].z, xv[j].z, acc[i][j]);
            acc[i][j] = fmaf(gv[i].w, xv[j].w, acc[i][j]);
          }
          if (tn == 0) bsum[i] += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
        }
      }
    }
  }
  // ---- sum the column slices (LDS), write this block's partial tile
  __syncthreads();
  float* red = wsm;   // [nslice][M*N + M]
  const int per = M * N + M;
  if (worker) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[slice * per + (tm * 4 + i) * N + tn * 4 + j] = acc[i][j];
      if (tn == 0) red[slice * per + M * N + tm * 4 + i] = bsum[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < per; e += 256) {
    float t = 0.f;
    for (int sidx = 0; sidx < nslice; ++sidx) t += red[sidx * per + e];
    if (e < M * N)
      a.part[(size_t)blockIdx.x * M * N + e] = t;
    else if (a.bias_part)
      a.bias_part[(size_t)blockIdx.x * M + (e - M * N)] = t;
  }
}

static bool wg_small_ok(int M, int N, int L) {
  if (M > 64 || N > 64 || (M & 3) || (N & 3) || (L & 3)) return false;
  const int ntile = (M >> 2) * (N >> 2);
  return ntile <= 256;
}
static int wg_small_lc(int M, int N) { return (M + N) <= 48 ? 256 : ((M + N) <= 96 ? 128 : 64); }
static size_t wg_small_lds(int M, int N) {
  const int LC = wg_small_lc(M, N), ntile = (M >> 2) * (N >> 2);
  const size_t stage = sizeof(float) * (size_t)(M + N) * (LC + 4), red = sizeof(float) * (size_t)(256 / ntile) * ((size_t)M * N + M);
  return stage > red ? stage : red;
}
static int wg_small_blocks(int Bt, int M, int N, int L) {
  const int LC = wg_small_lc(M, N);
  const long pieces = (long)Bt * ((L + LC - 1) / LC);
  // ONE resident round of blocks (round 6): as many as the CUs hold at once by LDS (<= 8 of 4 wavefronts).  The fixed 1024 of rounds
  // 3-5 were 1.33 rounds at the GroupComm shapes (3 blocks of 50 KB per CU = 768 slots): the last third of the pieces ran on a third
  // of the chip.  (Debug flag 1 << 18: 1024.)
  long slots = 1024;
  if (!(srf_debug_flags() & (1 << 18))) {
    long per_cu = (long)(160 * 1024 / wg_small_lds(M, N));
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    slots = per_cu * srf_device_cus();
  }
  return (int)(pieces < slots ? pieces : slots);
}

// out[m][n] (n < cols_out) = sum_p part[p][m][n]   (part rows have `cols` entries)
__global__ __launch_bounds__(256) void srf_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               int rows, int cols, int cols_out, int ld_out, int P,
                                                               float beta) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * cols_out) return;
  const int m = (int)(i / cols_out), n = (int)(i - (long)m * cols_out);
  const size_t src = (size_t)m * cols + n, stride = (size_t)rows * cols, dst = (size_t)m * ld_out + n;
  // gridDim.y > 1: the partials are split over blockIdx.y and atomically added (the host pre-zeroes `out` unless
  // it accumulates) -- a small output with many partials would otherwise be summed by a handful of threads
  const int per = (P + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(p0 + per, P);
  float s = 0.f;
  for (int p = p0; p < p1; ++p) s += part[(size_t)p * stride + src];
  if (gridDim.y > 1)
    atomicAdd(out + dst, s);
  else
    out[dst] = beta != 0.f ? fmaf(beta, out[dst], s) : s;
}

// The weight AND the bias partials of one wgrad in one launch (round 3: the separate bias reduction was a 12-us launch for
// 256-512 sums): threads [0, rows * cols_out) fold dW as srf_wgrad_reduce_kernel does, the next `rows` threads the bias.
// (round 4: four consecutive columns per thread -- float4 loads of the partials, four independent chains -- when the row
// lengths allow it; V = 1 is the scalar form.  cfg 2: 21 -> see profiles/r04_NOTES.md)
template <int V>
__global__ __launch_bounds__(256) void srf_wgrad_reduce2_kernel(const float* __restrict__ part, float* __restrict__ out, int rows,
                                                                int cols, int cols_out, int ld_out,
                                                                const float* __restrict__ bias_part, float* __restrict__ bias_out,
                                                                int P, float beta) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int cv = cols_out / V;
  const long nw = (long)rows * cv;
  if (i < nw) {
    const int m = (int)(i / cv), n = (int)(i - (long)m * cv) * V;
    const size_t src = (size_t)m * cols + n, stride = (size_t)rows * cols, dst = (size_t)m * ld_out + n;
    if constexpr (V == 4) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
      for (int p = 0; p < P; ++p) {
        const float4 t = *reinterpret_cast<const float4*>(part + (size_t)p * stride + src);
        s.x += t.x;
        s.y += t.y;
        s.z += t.z;
        s.w += t.w;
      }
      float4* o = reinterpret_cast<float4*>(out + dst);
      if (beta != 0.f) {
        const float4 c = *o;
        s = make_float4(fmaf(beta, c.x, s.x), fmaf(beta, c.y, s.y), fmaf(beta, c.z, s.z), fmaf(beta, c.w, s.w));
      }
      *o = s;
    } else {
      float s = 0.f;
      for (int p = 0; p < P; ++p) s += part[(size_t)p * stride + src];
      out[dst] = beta != 0.f ? fmaf(beta, out[dst], s) : s;
    }
  } else if (i < nw + rows) {
    const int m = (int)(i - nw);
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += bias_part[(size_t)p * rows + m];
    bias_out[m] = beta != 0.f ? fmaf(beta, bias_out[m], s) : s;
  }
}

// The same fold with four times the parallelism (round 5): a 256 x 512 gradient is 32 K float4 outputs = 129 blocks of the kernel
// above, every thread walking all P = 64 partials of its output one dependent-latency load after the other -- 18-21 us for 33 MB
// (1.7 TB/s), 36 times per cfg-2 step.  Here a block is 4 groups x 64 outputs: group g sums the partials [g P/4, (g + 1) P/4) of
// its output, the groups' sums are folded through LDS in group order (deterministic; the summation ORDER differs from the
// kernel above: ((p0..p15) + (p16..p31)) + ... instead of p0 + p1 + ...).
// ALIGNED_OUT = false: `out` rows are not 16-byte aligned (a gradient view at an odd offset of the flat gradient buffer: half of
// a model's weights sit behind a one-element PReLU slope) -- the partials are still read as float4, the result goes out as four
// dword stores.
template <bool ALIGNED_OUT>
__global__ __launch_bounds__(256) void srf_wgrad_reduce2g_kernel(const float* __restrict__ part, float* __restrict__ out, int rows,
                                                                 int cols, int cols_out, int ld_out,
                                                                 const float* __restrict__ bias_part, float* __restrict__ bias_out,
                                                                 int P, float beta) {
  __shared__ float4 red[3][64];
  const int grp = threadIdx.x >> 6, o = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 64 + o;
  const int cv = cols_out / 4;
  const long nw = (long)rows * cv;
  const int per = (P + 3) >> 2;
  const int p0 = grp * per, p1 = min(p0 + per, P);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  size_t dst = 0;
  const bool is_w = i < nw, is_b = !is_w && i < nw + rows;
  if (is_w) {
    const int m = (int)(i / cv), n = (int)(i - (long)m * cv) * 4;
    const size_t src = (size_t)m * cols + n, stride = (size_t)rows * cols;
    dst = (size_t)m * ld_out + n;
#pragma unroll 8
    for (int p = p0; p < p1; ++p) {
      const float4 t = *reinterpret_cast<const float4*>(part + (size_t)p * stride + src);
      s.x += t.x;
      s.y += t.y;
      s.z += t.z;
      s.w += t.w;
    }
  } else if (is_b) {
    const int m = (int)(i - nw);
#pragma unroll 8
    for (int p = p0; p < p1; ++p) s.x += bias_part[(size_t)p * rows + m];
  }
  if (grp) red[grp - 1][o] = s;
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float4 t = red[g][o];
      s.x += t.x;
      s.y += t.y;
      s.z += t.z;
      s.w += t.w;
    }
    if (is_w) {
      if constexpr (ALIGNED_OUT) {
        float4* op = reinterpret_cast<float4*>(out + dst);
        if (beta != 0.f) {
          const float4 c = *op;
          s = make_float4(fmaf(beta, c.x, s.x), fmaf(beta, c.y, s.y), fmaf(beta, c.z, s.z), fmaf(beta, c.w, s.w));
        }
        *op = s;
      } else {
        float* op = out + dst;
        if (beta != 0.f) s = make_float4(fmaf(beta, op[0], s.x), fmaf(beta, op[1], s.y), fmaf(beta, op[2], s.z), fmaf(beta, op[3], s.w));
        op[0] = s.x;
        op[1] = s.y;
        op[2] = s.z;
        op[3] = s.w;
      }
    } else if (is_b) {
      const int m = (int)(i - nw);
      bias_out[m] = beta != 0.f ? fmaf(beta, bias_out[m], s.x) : s.x;
    }
  }
}

// dst [rows][ld] <- sum over P partials [P][rows][cols] (first cols_out columns)
static int wg_reduce_launch(const float* part, float* out, int rows, int cols, int cols_out, int ld_out, int P,
                            int accumulate, hipStream_t st) {
  const long nw = (long)rows * cols_out;
  int psplit = 1;
  if (P >= 64 && nw * 4 <= 65536) psplit = P / 16;   // >= 16 partials per thread
  if (psplit > 64) psplit = 64;
  if (psplit > 1 && !accumulate)
    SRF_CHECK_HIP(hipMemset2DAsync(out, sizeof(float) * ld_out, 0, sizeof(float) * cols_out, rows, st));
  dim3 grid((unsigned)((nw + 255) / 256), (unsigned)psplit);
  hipLaunchKernelGGL(srf_wgrad_reduce_kernel, grid, dim3(256), 0, st, part, out, rows, cols, cols_out, ld_out, P,
                     accumulate ? 1.f : 0.f);
  return SRF_OK;
}

static int wg_pick_partials(int ntiles, int nchunks) {
  int P = (2 * 256 + ntiles - 1) / ntiles;   // one resident wave of blocks (2 per CU): fewer, longer partials
  if (P > nchunks) P = nchunks;
  if (P > 128) P = 128;
  // A multiple of the XCD count (the kernel's xcd_map) where that costs little: rounded UP when the chunk count allows, DOWN
  // otherwise, and left alone (plain mapping) when rounding down would drop more than a fifth of the blocks (ADVICE r4:
  // ntiles = 36 gave P = 15 -> 8, half a wave of blocks).  The models' shapes (8, 16, 32, 64 tiles) land on multiples as is.
  if (P >= 8 && (P & 7)) {
    const int up = (P + 7) & ~7, down = P & ~7;
    if (up <= nchunks && up <= 128) P = up;
    else if (5 * down >= 4 * P) P = down;
  }
  if (P < 1) P = 1;
  return P;
}

static void wg_geometry(int M, int N, int L, int Bt, WgArgs* a) {
  a->nMt = (M + WG_BM - 1) / WG_BM;
  a->nNt = (N + WG_BN - 1) / WG_BN;
  a->kc_len = 800;                              // 25 k-tiles per piece
  if (a->kc_len > L) a->kc_len = (L + 31) / 32 * 32;
  a->nKc = (L + a->kc_len - 1) / a->kc_len;
  a->P = wg_pick_partials(a->nMt * a->nNt, Bt * a->nKc);
  a->xcd_map = (a->P % 8 == 0 && !(srf_debug_flags() & 4096)) ? 1 : 0;
}

// The wide tile (srf_pw_wgrad_wide_kernel): 1 = 256 x 128, 2 = 128 x 256, 0 = not for this shape.  (Debug flag 1 << 18: never -- A/B.)
static int wg_wide_form(int M, int N, int L) {
  if (L % WG_BK || srf_kernel_mode() == 1 || (srf_debug_flags() & (1 << 18))) return 0;
  if (M % 256 == 0 && N % 128 == 0) return 1;
  if (M % 128 == 0 && N % 256 == 0) return 2;
  return 0;
}
static void wg_geometry_wide(int form, int M, int N, int L, int Bt, WgArgs* a) {
  a->nMt = M / (form == 1 ? 256 : 128);
  a->nNt = N / (form == 1 ? 128 : 256);
  a->kc_len = 800;
  if (a->kc_len > L) a->kc_len = L;
  a->nKc = (L + a->kc_len - 1) / a->kc_len;
  const int ntiles = a->nMt * a->nNt, nchunks = Bt * a->nKc;
  int P = (256 + ntiles - 1) / ntiles;          // ONE block per CU
  if (P > nchunks) P = nchunks;
  if (P > 128) P = 128;
  if (P >= 8 && (P & 7)) {
    const int up = (P + 7) & ~7, down = P & ~7;
    if (up <= nchunks && up <= 128) P = up;
    else if (5 * down >= 4 * P) P = down;
  }
  a->P = P < 1 ? 1 : P;
  a->xcd_map = (a->P % 8 == 0) ? 1 : 0;
  // k chunks per block -> one chunk k times as long where the time chunks divide evenly: one pipeline fill per block instead of k
  // (cfg 2: 2 x 25 k-tiles -> 1 x 50)
  const int k = nchunks / a->P;
  if (k >= 2 && nchunks % a->P == 0 && a->nKc % k == 0 && L % (a->kc_len * k) == 0 && !(srf_debug_flags() & (1 << 20))) {
    a->kc_len *= k;
    a->nKc /= k;
  }
}

extern "C" size_t srf_pw_wgrad_scratch_bytes(int Bt, int Cout, int Cin, int L) {
  if (Bt <= 0 || Cout <= 0 || Cin <= 0 || L <= 0) return 0;
  WgArgs a;
  wg_geometry(Cout, Cin, L, Bt, &a);
  size_t P = a.P;
  if ((L % WG_BK) == 0 && ((Cout % 256 == 0 && Cin % 128 == 0) || (Cout % 128 == 0 && Cin % 256 == 0))) {
    WgArgs w;
    wg_geometry_wide(Cout % 256 == 0 && Cin % 128 == 0 ? 1 : 2, Cout, Cin, L, Bt, &w);
    if ((size_t)w.P > P) P = w.P;
  }
  if (wg_small_ok(Cout, Cin, L)) {
    const size_t ps = wg_small_blocks(Bt, Cout, Cin, L);
    if (ps > P) P = ps;
  }
  return sizeof(float) * P * ((size_t)Cout * Cin + Cout);
}

// g: [Bt,Cout,L] gradient w.r.t. the conv output; x: [Bt,Cin,L] the conv's (pre-prologue) input;
// dw: [Cout,dw_cols] (first dw_cols columns of the Cout x Cin product); dbias: [Cout] or NULL.
// accumulate != 0: dw += ..., dbias += ... (else overwrite).
extern "C" int srf_pw_wgrad_ld(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout,
                               int L, float* dw, int dw_cols, int dw_ld, float* dbias, int accumulate, void* scratch,
                               void* stream);

extern "C" int srf_pw_wgrad_cols(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout,
                                 int L, float* dw, int dw_cols, float* dbias, int accumulate, void* scratch,
                                 void* stream) {
  return srf_pw_wgrad_ld(g, x, in_norm, Bt, Cin, Cout, L, dw, dw_cols, dw_cols, dbias, accumulate, scratch, stream);
}

// dw rows have a pitch of dw_ld floats (>= dw_cols): writes a column block of a wider matrix
extern "C" int srf_pw_wgrad_ld(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout,
                               int L, float* dw, int dw_cols, int dw_ld, float* dbias, int accumulate, void* scratch,
                               void* stream) {
  SRF_CHECK_ARG(g && x && dw && scratch, "srf_pw_wgrad: null pointer");
  SRF_CHECK_ARG(Bt > 0 && Cin > 0 && Cout > 0 && L > 0 && (L % 4) == 0, "srf_pw_wgrad: bad sizes (L %% 4 == 0 required)");
  SRF_CHECK_ARG(dw_cols > 0 && dw_cols <= Cin && dw_ld >= dw_cols, "srf_pw_wgrad: dw_cols / dw_ld out of range");
  SRF_CHECK_ARG(srf_aligned16(g) && srf_aligned16(x), "srf_pw_wgrad: operands must be 16-byte aligned");
  WgArgs a;
  a.g = g;
  a.x = x;
  a.nrm = srf_norm_dev(in_norm);
  a.inv_count = 1.0 / ((double)Cin * (double)L);
  a.M = Cout;
  a.N = Cin;
  a.L = L;
  a.Bt = Bt;
  wg_geometry(Cout, Cin, L, Bt, &a);
  a.part = reinterpret_cast<float*>(scratch);
  a.bias_part = dbias ? a.part + (size_t)a.P * Cout * Cin : nullptr;
  if (a.nrm.sums) SRF_CHECK_ARG(a.nrm.gamma && a.nrm.beta, "srf_pw_wgrad: norm without gamma/beta");
  hipStream_t st = (hipStream_t)stream;
  const int pro = a.nrm.sums ? (a.nrm.prelu ? 2 : 1) : (a.nrm.prelu ? 3 : 0);
  if (wg_small_ok(Cout, Cin, L) && srf_kernel_mode() != 1) {
    const int LC = wg_small_lc(Cout, Cin), npl = (L + LC - 1) / LC;
    a.P = wg_small_blocks(Bt, Cout, Cin, L);
    a.bias_part = dbias ? a.part + (size_t)a.P * Cout * Cin : nullptr;
    const size_t lds = wg_small_lds(Cout, Cin);
    dim3 gs((unsigned)a.P), bs(256);
    switch (pro) {
      case 0: hipLaunchKernelGGL(srf_pw_wgrad_small_kernel<0>, gs, bs, lds, st, a, LC, npl); break;
      case 1: hipLaunchKernelGGL(srf_pw_wgrad_small_kernel<1>, gs, bs, lds, st, a, LC, npl); break;
      case 2: hipLaunchKernelGGL(srf_pw_wgrad_small_kernel<2>, gs, bs, lds, st, a, LC, npl); break;
      default: hipLaunchKernelGGL(srf_pw_wgrad_small_kernel<3>, gs, bs, lds, st, a, LC, npl); break;
    }
    SRF_CHECK_LAUNCH("pw_wgrad_small", st);
  } else if (const int wide = wg_wide_form(Cout, Cin, L)) {
    wg_geometry_wide(wide, Cout, Cin, L, Bt, &a);
    a.bias_part = dbias ? a.part + (size_t)a.P * Cout * Cin : nullptr;
    constexpr int lds = 2 * 2 * (256 + 128) * WG_PITCH;
    const long ok = srf_device_cached(8, [](void*) -> long {
      bool good = true;
      const void* fns[] = {(const void*)&srf_pw_wgrad_wide_kernel<0, 256, 128>, (const void*)&srf_pw_wgrad_wide_kernel<1, 256, 128>,
                           (const void*)&srf_pw_wgrad_wide_kernel<2, 256, 128>, (const void*)&srf_pw_wgrad_wide_kernel<3, 256, 128>,
                           (const void*)&srf_pw_wgrad_wide_kernel<0, 128, 256>, (const void*)&srf_pw_wgrad_wide_kernel<1, 128, 256>,
                           (const void*)&srf_pw_wgrad_wide_kernel<2, 128, 256>, (const void*)&srf_pw_wgrad_wide_kernel<3, 128, 256>};
      for (const void* f : fns) good &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
      return good ? 1 : 0;
    }, nullptr);
    SRF_CHECK_ARG(ok == 1, "srf_pw_wgrad: cannot reserve %d bytes of LDS", lds);
    dim3 grid((unsigned)(a.nMt * a.nNt * a.P)), block(512);
#define WG_WIDE(P_) \
  if (wide == 1) hipLaunchKernelGGL((srf_pw_wgrad_wide_kernel<P_, 256, 128>), grid, block, lds, st, a); \
  else hipLaunchKernelGGL((srf_pw_wgrad_wide_kernel<P_, 128, 256>), grid, block, lds, st, a)
    switch (pro) {
      case 0: WG_WIDE(0); break;
      case 1: WG_WIDE(1); break;
      case 2: WG_WIDE(2); break;
      default: WG_WIDE(3); break;
    }
#undef WG_WIDE
    SRF_CHECK_LAUNCH("pw_wgrad", st);
  } else {
  dim3 grid((unsigned)(a.nMt * a.nNt * a.P)), block(512);
  // (debug flag 1 << 19: the masked form for full shapes too -- A/B)
  const bool full = Cout % WG_BM == 0 && Cin % WG_BN == 0 && L % WG_BK == 0 && a.kc_len % WG_BK == 0 && !(srf_debug_flags() & (1 << 19));
#define WG_GO(P_) \
  if (full) hipLaunchKernelGGL((srf_pw_wgrad_kernel<P_, true>), grid, block, 0, st, a); \
  else hipLaunchKernelGGL((srf_pw_wgrad_kernel<P_, false>), grid, block, 0, st, a)
  switch (pro) {
    case 0: WG_GO(0); break;
    case 1: WG_GO(1); break;
    case 2: WG_GO(2); break;
    default: WG_GO(3); break;
  }
#undef WG_GO
  SRF_CHECK_LAUNCH("pw_wgrad", st);
  }
  int rc;
  if (dbias && !(a.P >= 64 && (long)Cout * dw_cols * 4 <= 65536)) {   // (large outputs: no partial split, one launch for both)
    const bool p4 = (Cin % 4 == 0) && (dw_cols % 4 == 0) && srf_aligned16(a.part);      // float4 reads of the partials
    const bool v4 = p4 && (dw_ld % 4 == 0) && srf_aligned16(dw);                          // ... and float4 stores of the result
    const bool four_groups = p4 && a.P >= 16 && !(srf_debug_flags() & (1 << 25));        // (debug flag 1 << 25: the one-chain fold, A/B)
    const long n = (long)Cout * (dw_cols / ((v4 || four_groups) ? 4 : 1)) + Cout;
    if (four_groups) {
      if (v4)
        hipLaunchKernelGGL(srf_wgrad_reduce2g_kernel<true>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, a.part, dw, Cout, Cin,
                           dw_cols, dw_ld, a.bias_part, dbias, a.P, accumulate ? 1.f : 0.f);
      else
        hipLaunchKernelGGL(srf_wgrad_reduce2g_kernel<false>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, a.part, dw, Cout, Cin,
                           dw_cols, dw_ld, a.bias_part, dbias, a.P, accumulate ? 1.f : 0.f);
    } else if (v4)
      hipLaunchKernelGGL(srf_wgrad_reduce2_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, dw, Cout, Cin,
                         dw_cols, dw_ld, a.bias_part, dbias, a.P, accumulate ? 1.f : 0.f);
    else
      hipLaunchKernelGGL(srf_wgrad_reduce2_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, dw, Cout, Cin,
                         dw_cols, dw_ld, a.bias_part, dbias, a.P, accumulate ? 1.f : 0.f);
  } else {
    rc = wg_reduce_launch(a.part, dw, Cout, Cin, dw_cols, dw_ld, a.P, accumulate, st);
    if (rc) return rc;
    if (dbias) {
      rc = wg_reduce_launch(a.bias_part, dbias, Cout, 1, 1, 1, a.P, accumulate, st);
      if (rc) return rc;
    }
  }
  SRF_CHECK_LAUNCH("pw_wgrad_reduce", st);
  return SRF_OK;
}

extern "C" int srf_pw_wgrad(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout, int L,
                            float* dw, float* dbias, int accumulate, void* scratch, void* stream) {
  return srf_pw_wgrad_cols(g, x, in_norm, Bt, Cin, Cout, L, dw, Cin, dbias, accumulate, scratch, stream);
}
