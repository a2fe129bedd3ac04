// K2 (fast path) -- the 256 x 128 split-precision MFMA GEMM with pre-split weights (round 2's kernel, git tag r3-gemm-x3v-kernel,
// with its memory pipeline repaired in round 3).
//
// Block tile 256 (M) x 128 (time) x 32 (K): for Cout = 256 one block owns all of M, so an activation tile goes through GlobLN /
// PReLU / split exactly once; the weights are split into bf16 hi | lo ONCE per forward (srf_x3w_pack_kernel) and stored tile by
// tile as the exact, XOR-swizzled LDS image of a stage (64-byte rows of 32 bf16, 16-byte chunks swizzled by (row >> 2) & 3:
// conflict-free ds_read_b128 fragments), so the A operand is a global_load_lds_dwordx4 DMA; 3 LDS stages of 48 KB (A_hi | A_lo |
// B_hi | B_lo), one 512-thread block per CU, ONE barrier per k-tile; persistent blocks, operand pipeline running across tile
// boundaries; the leftover tiles of the last round cut into quarter tiles (256 x 32) dealt to all blocks.  (x = hi + lo
// in bf16, three v_mfma_f32_32x32x16_bf16 per product block, fp32 accumulate; reference sites
// improved_sudormrf.py:256-259, :174, :196, :220, :268-269, :295-298).  What changed, and why (round-3 reading of the round-2
// kernel's device assembly, tools/isa_waits.py -- VERDICT r2 weak 3: res_conv 0.40 / proj_1x1 0.32 of the HBM ceiling, "issue
// bound"; the round-2 ablation had shown the k-loop + epilogue costing far more together than apart):
//   1. hipcc placed FULL DRAINS of the memory pipeline (s_waitcnt vmcnt(0)) inside the persistent loop:
//        * once per tile in the GlobLN-prologue kernels: the statistics of the tile after next were fetched with a vector load
//          right behind the activation loads of two k-tiles -- its wait is a wait for everything in flight.  Now every block
//          finalises {mean, rstd} of all examples once, at kernel start, into an LDS table (one table per launch, <= 8 KB);
//        * every second k-step in the prologue-free kernels (proj_1x1, mask): the conversion of BOTH activation register sets
//          was hoisted to the top of the two-step loop body, i.e. the set requested one step earlier was waited for with
//          vmcnt(0).  The operand values are now made opaque at their point of use (an empty volatile asm), which pins the
//          conversion -- and the compiler's counted wait -- behind the previous step's barrier;
//        * at the start of every epilogue: under the register pressure of 64 accumulators + 64 prefetched residual values
//          the allocator re-used destination registers of in-flight activation loads.  The epilogue is now specialised at compile
//          time (EPI): the launches without a residual / mask multiplier (proj_1x1, bottleneck) carry no such registers.
//   2. Every k-step began with an LDS bubble: all eight wavefronts read their first fragments right behind the barrier, with
//      the matrix pipe idle.  The fragments of a k-tile's first half are now requested BEHIND the barrier that completes the
//      tile, under the MFMAs of the previous tile's second half (two fragment sets of 32 registers; the last two steps of a tile,
//      where the epilogue's operands are in flight, read just in time instead).
// Measured and removed again, round 3 (all bit-identical to this kernel; code: git tag r3-gemm-experiments, numbers:
// profiles/r03_NOTES.md): a DRIP epilogue (a finished tile's accumulators parked in 64 registers and stored during the next
// tile's first 8 / 16 k-steps as dword buffer stores straight from the MFMA layout: proj_1x1 138 vs 119 us), per-XCD WORK
// QUEUES with phase seeds (blocks start out of phase on 0-3 quarter tiles, then draw tiles with one atomicAdd each: 125 vs
// 115 us -- a quarter tile costs ~0.6 of a full tile's time), start-up stagger (114-121 vs 108 us).  Why none of the
// re-orderings pays -- round 4's finding (profiles/r04_NOTES.md, tools/probes/cu_stream_probe.hip; round 3's "power cap" reading
// is RETRACTED: with the MFMAs compiled out the launches keep 81-93 % of their time at 2.4 GHz and 1.05 kW): on gfx950 a
// wavefront's vector-memory instructions do not issue while other wavefronts of its SIMD run MFMAs back to back, so in a block
// whose eight waves all multiply, loads issue only in the gaps (barriers, LDS waits) and the waves phase-lock; and one CU's L2
// hits queue behind its own HBM misses.  Round 4 built three kernels around that, all bit-identical to this one: loader waves / a
// SIMD kept free of MFMAs (csrc/experiments/, lab build only: 124-170 us against this kernel's 109-118 us on res_conv -- the
// epilogue and the extra barriers cost what the overlap wins), and TWO CO-RESIDENT HALF-SIZE BLOCKS PER CU (srf_pwconv_x3p.hip),
// a tie in isolation that srf_forward uses because its blocks leave room for the other stream's kernels (srf_pwconv.hip,
// srf_pw_256_launch).  This kernel serves every other caller, the mask / fused-tail GEMM and the training forward.
// Round 4 also added NP = 4: operands split into two FP16 parts (3 f16 MFMAs per product block, weights pre-scaled by 2^4) --
// the training forward's default (srf_pwconv.hip: srf_train_f16_split).
// Prologue / epilogue semantics are those of srf_pw.h (PwArgs).
#include <type_traits>

#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int W_BM = 256, W_BN = 128, W_BK = 32;
constexpr int W_A_IMG = W_BM * 64;                       // [256][32] bf16, 64-B rows
constexpr int W_B_IMG = W_BN * 64;                       // [128][32] bf16
constexpr int W_STAGE = 2 * W_A_IMG + 2 * W_B_IMG;       // A_hi | A_lo | B_hi | B_lo = 48 KB
constexpr int W_NSTAGE = 3;
constexpr int W_WTILE_BYTES = 2 * W_A_IMG;               // packed weights of one (m-tile, k-tile) (srf_x3w_pack_kernel's format)
constexpr int W_MAX_STAT_EXAMPLES = 1024;                // LDS statistics table: 8 KB behind the stages

__host__ __device__ __forceinline__ int w_swz(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

__device__ __forceinline__ void w_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

// NP 4: the two parts are fp16 (11 + 11 mantissa bits instead of bf16's 8 + 8): hi = fp16(x), lo = fp16(x - hi).
// The weights are stored times 2^4 (W_F16_WSCALE) so that the lo parts of typical weights (~0.03) stay in fp16's normal range;
// the epilogue multiplies the accumulators by 2^-4 (exact).  Packets keep the bf16x8 container type (16 bytes).
// Range contract of this form: |operand| < 65520 (beyond: non-finite outputs, loudly -- see w_split8_f16); operands below ~2^-14
// lose their lo part to fp16's subnormal range, i.e. they carry 11 instead of 22 bits -- an absolute error of <= 3e-8 per term,
// invisible next to O(1) terms, but a tensor that is tiny AS A WHOLE (every |x| < 1e-3) reaches only the two-part bf16
// kernel's accuracy class; the three-part bf16 form (flag 16384) has neither limit.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float W_F16_WSCALE = 16.f;
__device__ __forceinline__ void w_split8_f16(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  f16x8 h8, l8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // No range guard (round 5): an operand beyond fp16's range (|v| >= 65520, e.g. a diverged step) converts to
    // +-inf and the lo part to NaN, so it POISONS the output like it would in the fp32 reference's overflow / NaN case --
    // rounds 3-4 clamped to +-6e4 (fmed3, which also turned NaN into -6e4): a finite, plausible, wrong result.  The remedy
    // for legitimately large activations is the three-part bf16 form (debug flag 16384: fp32's exponent range).
    const float vc = v[j];
    const _Float16 h = (_Float16)vc;
    h8[j] = h;
    l8[j] = (_Float16)(vc - (float)h);
  }
  hi = __builtin_bit_cast(bf16x8, h8);
  lo = __builtin_bit_cast(bf16x8, l8);
}

#define W_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// PRO: 0 = identity, 1 = GlobLN, 2 = GlobLN + PReLU, 3 = PReLU only.
// EPI: 0 = bias (+ statistics), 1 = bias + residual, 2 = ReLU(bias + .) x mul (mask epilogue), 3 = decided at run time (any),
//      4 = the mask epilogue FUSED with the decoder's contraction (K5): the masked values never leave the chip -- see the EPI 4
//          block in the tile loop; bias_r = the bias again as a noalias argument (scalar loads), wdpack = the decoder weights as
//          MFMA A fragments (srf_x3w_pack_dec_kernel), zpart = [Bt][nMt][zM][L] partial decoder frames, zM = sources x taps.
// gamma / beta come again as noalias kernel arguments so that they are fetched with scalar loads.
// Work distribution, barrier protocol and LDS images: as in round 2's kernel (file header).
// CP: cache policy of the streamed tensors (see the launch function).
// NP: 2 = two bf16 parts per operand (hi | lo, three MFMAs per product block: the inference GEMM); 3 = THREE parts (h | m | l =
//     24 mantissa bits, the exact-fp32 class) and six MFMAs -- l*h, h*l, m*m, m*h, h*m, h*h -- for the training forward
//     (VERDICT r2 next 4).  Same stages, DMA, swizzle and fragment reads: a k-tile then holds 16 k instead of 32 -- the "hi"
//     image carries h (slots 0-15) and m (slots 16-31) of those 16 k, the "lo" image l (slots 0-15) -- so what was the second
//     k-sub-step's fragment pair is the m part (srf_x3w_pack3_kernel writes the weights that way, lds_store the activations).
template <int PRO, int EPI, int CP = 0, int NP = 2>
__global__ __launch_bounds__(512, 2) void srf_pw_x3w_kernel(PwArgs a, const char* __restrict__ wpack, int nMt, int nLt,
                                                            int total, int rounds, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ bias_r,
                                                            const char* __restrict__ wdpack, float* __restrict__ zpart, int zM, int mgrp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // W_NSTAGE * W_STAGE (+ the statistics table)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // full tiles: 4 x 2 wavefronts, 64 x 64 each
  const int Cin = a.Cin, L = a.L;
  constexpr int KT = NP == 3 ? 16 : W_BK;    // k per pipeline step
  const int nk = Cin / KT;                   // even, >= 4 (host checks)
  const int nblk = gridDim.x;
  const int nquart = 4 * (total - rounds * nblk);                                   // quarter tiles of the leftover round
  const int nq_mine = ((int)blockIdx.x < nquart) ? (nquart - (int)blockIdx.x + nblk - 1) / nblk : 0;
  const int ntile = rounds + nq_mine;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;
  const int x_bytes = a.Bt * Cin * L * 4;
  constexpr bool kHasExt = EPI != 0 && EPI != 4;   // the strip epilogue reads a second tensor (residual or mask multiplier)

  // ---- GlobLN statistics of every example, once per block: {mean, rstd} in LDS behind the stages.  (A vector load per tile
  // -- the round-2 form -- is waited for with vmcnt(0) right behind two k-tiles' worth of activation loads.)  Wavefront w
  // finalises examples 4w .. 4w+3 (+32 ...): the four bucket loads are issued together, then reduced (DPP, VALU only).
  float2* stat_tab = reinterpret_cast<float2*>(smem + W_NSTAGE * W_STAGE);
  if constexpr (PRO == 1 || PRO == 2) {
    for (int b0 = wave * 4; b0 < a.Bt; b0 += 32) {
      double2 bk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = min(b0 + u, a.Bt - 1);
        bk[u] = reinterpret_cast<const double2*>(a.nrm.sums)[(size_t)b * SRF_STAT_BUCKETS + lane];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double s = srf_dpp_wave_sum(bk[u].x), q = srf_dpp_wave_sum(bk[u].y);   // totals in lane 63
        const double m = s * a.inv_count;
        double v = q * a.inv_count - m * m;
        v = v < 0.0 ? 0.0 : v;
        if (lane == 63 && b0 + u < a.Bt) stat_tab[b0 + u] = make_float2((float)m, (float)(1.0 / sqrt(v + 1e-8)));
      }
    }
    __syncthreads();
  }

  // Tile order: virtual id v = mi + G (lt + nLt (b + Bt grp)), m-tile = grp G + mi.  G = nMt (one group; the default) is the
  // order of round 2's kernel: the nMt blocks that share one activation tile are neighbours on one XCD.  G < nMt (host: when
  // the packed weights exceed what an XCD's 4-MB L2 can keep -- the mask GEMMs of cfg 4 / cfg 5, 8 / 16 MB) gives every XCD a
  // GROUP of G m-tiles whose weight slabs stay L2-resident while it walks the (example, time) tiles; all XCDs walk those in the
  // same order at the same pace, so the activation tile one of them fetched from HBM is a MALL hit for the others.  Before:
  // every XCD re-fetched ALL weights every round -- PMC, cfg-5 mask GEMM: 35.2 GB of L2 misses per launch against 3.8 GB
  // algorithmic; with groups 15.6 GB and 6.31-6.36 vs 6.48-6.49 ms (same box).  (Non-temporal loads of the encoder multiplier
  // on top: 12.7 GB, but no faster at cfg 5 and 1 % slower at cfg 4 -- not kept.)
  struct TileCur {
    int i, v, mt, lt, b, q, mi, grp;   // q: -1 = full tile, 0..3 = quarter of its parent tile; v < 0: no tile
  };
  const int G = mgrp;
  const int vstep = nblk >> 3;
  const int st_mi = vstep % G, st_r1 = vstep / G;
  const int st_l = st_r1 % nLt, st_r2 = st_r1 / nLt;
  const int st_b = st_r2 % a.Bt, st_g = st_r2 / a.Bt;
  const int qfirst = (a.epi_mask >> 12) & 1 ? nq_mine : 0;     // tiles [0, qfirst) are quarter tiles, then the full ones
  auto is_quarter = [&](int i) { return qfirst ? i < qfirst : i >= rounds; };
  auto cur_from_v = [&](TileCur& c, int v) {
    const int t = v / G;
    c.v = v;
    c.mi = v - t * G;
    const int u = t / nLt;
    c.lt = t - u * nLt;
    c.grp = u / a.Bt;
    c.b = u - c.grp * a.Bt;
    c.mt = c.grp * G + c.mi;
  };
  auto cur_set = [&](TileCur& c, int i) {   // by division: a block's first tile and its quarter tiles
    int p = blockIdx.x + (i - qfirst) * nblk;
    c.q = -1;
    if (is_quarter(i)) {
      const int qi = blockIdx.x + (qfirst ? i : i - rounds) * nblk;
      p = rounds * nblk + (qi >> 2);
      c.q = qi & 3;
    }
    c.i = i;
    cur_from_v(c, srf_xcd_remap(p, total));
  };
  auto cur_next = [&](TileCur& c) {
    const int i = c.i + 1;
    if (is_quarter(i) || c.q >= 0) {
      cur_set(c, i);
      return;
    }
    c.i = i;
    c.v += vstep;
    c.mi += st_mi;
    int cy = c.mi >= G ? 1 : 0;
    c.mi -= cy ? G : 0;
    c.lt += st_l + cy;
    cy = c.lt >= nLt ? 1 : 0;
    c.lt -= cy ? nLt : 0;
    c.b += st_b + cy;
    cy = c.b >= a.Bt ? 1 : 0;
    c.b -= cy ? a.Bt : 0;
    c.grp += st_g + cy;
    c.mt = c.grp * G + c.mi;
  };
  // ---- B staging geometry: thread -> time step n = tid & 127, k-group kg = tid >> 7 (wave-uniform), 8 k rows
  //      (NP 3: 4 k rows per thread; h -> chunk b_c >> 1 of the hi image, m -> chunk 2 + (b_c >> 1), l -> chunk b_c >> 1 of the lo
  //      image, at byte (b_c & 1) * 8 of the 16-byte chunk)
  const int b_n = tid & 127, b_c = wave >> 1, b_kg = NP == 3 ? b_c * 4 : b_c * 8;
  const int b_lds = NP == 3 ? 2 * W_A_IMG + w_swz(b_n, b_c >> 1) + (b_c & 1) * 8 : 2 * W_A_IMG + w_swz(b_n, b_c);
  const int b_lds_m = 2 * W_A_IMG + w_swz(b_n, 2 + (b_c >> 1)) + (b_c & 1) * 8;   // NP 3 only
  __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);

  struct TileP {
    const char* a_src;   // this wavefront's 4-KB slice of the tile's packed weights, k-tile 0 (wave-uniform)
    int b_vo;            // per-lane byte offset of (example, k row b_kg, column) inside X; out of range = fetch nothing
    float mean, rstd;    // GlobLN statistics of the tile's example (PRO 1 / 2)
  };
  auto make_tile = [&](const TileCur& c) {
    TileP t;
    t.a_src = wpack + (size_t)c.mt * nk * W_WTILE_BYTES + wave * 4096;
    const int col0 = c.lt * W_BN + (c.q < 0 ? 0 : c.q * 32);
    const int width = c.q < 0 ? W_BN : 32;
    const int off = ((c.b * Cin + b_kg) * L + min(col0 + b_n, L - 1)) * 4;   // columns >= L are never stored
    t.b_vo = b_n < width ? off : x_bytes;
    t.mean = 0.f;
    t.rstd = 1.f;
    if constexpr (PRO == 1 || PRO == 2) {
      const float2 mr = stat_tab[c.b];     // LDS: counted on lgkmcnt, never behind the activation loads
      t.mean = mr.x;
      t.rstd = mr.y;
    }
    return t;
  };

  struct Regs {
    float b[8];
  };
  // A: LDS DMA through inline asm (invisible to hipcc's vmcnt bookkeeping: counted by hand, see `step`)
  auto gload_a = [&](const TileP& t, int kt, int stage) __attribute__((always_inline)) {
    const char* src = t.a_src + (size_t)kt * W_WTILE_BYTES + lane * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)W_LDS(smem + stage * W_STAGE + wave * 4096));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src + i * 1024), "s"(dst + i * 1024)
          : "memory");
    }
  };
  auto gload_b = [&](Regs& r, const TileP& t, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < (NP == 3 ? 4 : 8); ++j)
      r.b[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, t.b_vo, (kt * KT + j) * L * 4, (CP & 4) ? 2 : 0));
  };
  // GlobLN / PReLU / split of k-tile kt (tile t) -> B images of `stage`
  auto lds_store = [&](const Regs& r, const TileP& t, int kt, int stage) __attribute__((always_inline)) {
    char* base = smem + stage * W_STAGE + b_lds;
    if constexpr (NP == 3) {
      float x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = r.b[j];
      asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 ph, pm, pl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x0 = x[j];
        if (PRO == 1 || PRO == 2) {
          const int k = kt * KT + b_kg + j;
          const float sc = gamma[k] * t.rstd;
          x0 = fmaf(x0, sc, beta[k] - t.mean * sc);
        }
        if (PRO == 2 || PRO == 3) x0 = srf_prelu(x0, slope);
        const __bf16 h = (__bf16)x0;
        const float r1 = x0 - (float)h;
        const __bf16 m = (__bf16)r1;
        ph[j] = h;
        pm[j] = m;
        pl[j] = (__bf16)(r1 - (float)m);
      }
      *reinterpret_cast<bf16x4*>(base) = ph;
      *reinterpret_cast<bf16x4*>(smem + stage * W_STAGE + b_lds_m) = pm;
      *reinterpret_cast<bf16x4*>(base + W_B_IMG) = pl;
      return;
    }
    float vb[8];
    // Opaque at the point of use: volatile asm statements keep their order, so the conversion below -- and with it the
    // compiler's counted wait for this register set's loads -- cannot be hoisted above the previous step's barrier.  (One
    // statement for the whole set: per-element statements also keep the gamma / beta scalar loads from merging.)
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = r.b[j];
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x0 = x[j];
      if (PRO == 1 || PRO == 2) {
        const int k = kt * W_BK + b_kg + j;
        const float sc = gamma[k] * t.rstd;
        x0 = fmaf(x0, sc, beta[k] - t.mean * sc);
      }
      if (PRO == 2 || PRO == 3) x0 = srf_prelu(x0, slope);
      vb[j] = x0;
    }
    bf16x8 hi, lo;
    if constexpr (NP == 4) w_split8_f16(vb, hi, lo);
    else w_split8(vb, hi, lo);
    *reinterpret_cast<bf16x8*>(base) = hi;
    *reinterpret_cast<bf16x8*>(base + W_B_IMG) = lo;
  };

  // ---- MFMA.  Fragment rows of this lane: A operand row = <wave's first row> + mi*32 + (lane & 31), chunk = 2 ks + (lane >> 5).
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int fr = lane & 31, fc = lane >> 5;
  int a_off[2][2], b_off[2][2];   // [mi | ni][ks]
  auto set_off = [&](bool quarter) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        a_off[t][ks] = w_swz(quarter ? wave * 32 + fr : wm * 64 + t * 32 + fr, 2 * ks + fc);
        b_off[t][ks] = 2 * W_A_IMG + w_swz(quarter ? fr : wn * 64 + t * 32 + fr, 2 * ks + fc);
      }
  };
  struct Frags {   // the fragments of one k-sub-step (16 of the k-tile's 32 k): 32 registers
    bf16x8 ah[2], al[2], bh[2], bl[2];   // [mi | ni]
  };
  auto read_frags = [&](Frags& f, int stage, int ks, auto full_tag) __attribute__((always_inline)) {
    constexpr int NT = decltype(full_tag)::value ? 2 : 1;
    const char* base = smem + stage * W_STAGE;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f.ah[t] = *reinterpret_cast<const bf16x8*>(base + a_off[t][ks]);
      f.bh[t] = *reinterpret_cast<const bf16x8*>(base + b_off[t][ks]);
      if (NP == 3 && ks == 1) continue;     // (NP 3: "sub-step 1" = the m parts, which live in the hi images only)
      f.al[t] = *reinterpret_cast<const bf16x8*>(base + W_A_IMG + a_off[t][ks]);
      f.bl[t] = *reinterpret_cast<const bf16x8*>(base + W_B_IMG + b_off[t][ks]);
    }
  };
  auto mma = [&](const Frags& f, auto full_tag) __attribute__((always_inline)) {
    constexpr int NT = decltype(full_tag)::value ? 2 : 1;
    // pass-major order: independent accumulators between two MFMAs on the same one (and the summation order of every split-bf16 kernel of the library)
    auto mf = [](const bf16x8& x, const bf16x8& y, const f32x16& c) __attribute__((always_inline)) {
      if constexpr (NP == 4)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
      else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
    };
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi) acc[mi][ni] = mf(f.al[mi], f.bh[ni], acc[mi][ni]);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi) acc[mi][ni] = mf(f.ah[mi], f.bl[ni], acc[mi][ni]);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi) acc[mi][ni] = mf(f.ah[mi], f.bh[ni], acc[mi][ni]);
  };
  // NP 3: f = {h, l} fragments (sub-step 0), g = {m} fragments (sub-step 1).  Smallest terms first: l*h, h*l | m*m, m*h, h*m, h*h
  auto mma3_a = [&](const Frags& f, auto full_tag) __attribute__((always_inline)) {
    constexpr int NT = decltype(full_tag)::value ? 2 : 1;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bl[ni], acc[mi][ni], 0, 0, 0);
  };
  auto mma3_b = [&](const Frags& f, const Frags& g, auto full_tag) __attribute__((always_inline)) {
    constexpr int NT = decltype(full_tag)::value ? 2 : 1;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.ah[mi], g.bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], g.bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
  };

  // One pipeline step = k-tile kt of the current tile (in stage s0): split k-tile kt+1 into stage s1, start the DMA of k-tile kt+2
  // into stage s2 and the activation loads of k-tile kt+3, multiply k-tile kt.  k-tile indices >= nk belong to the NEXT tile.
  // Barrier protocol (one barrier per step; before it every wavefront's ds_writes of k-tile kt+1 are
  // done and its DMA pieces of k-tile kt+1 -- issued a step ago, >= 20 memory operations ago -- have landed).
  // Fragment schedule.  HAVE0: the first half's fragments (f0) were requested by the previous step, behind ITS barrier, i.e.
  // under the previous k-tile's last 12 MFMAs.  PREF: this step does the same for the next k-tile (stage s1 is complete for
  // every reader once this step's barrier is passed; its reads go into f0, which the MFMAs issued before the barrier have
  // consumed).  The register sets f0 / f1 are therefore both live across the step: 64 registers, against the 32 of reading
  // just in time (HAVE0 = PREF = false: the last two steps of a tile, where the epilogue's operands occupy registers).
  int s0 = 0;
  TileP tc, tn;
  Frags f0, f1;
  auto pick = [&](int k, int& kk) __attribute__((always_inline)) {
    const bool nx = k >= nk;   // wave-uniform
    kk = nx ? k - nk : k;
    TileP t;
    t.a_src = nx ? tn.a_src : tc.a_src;
    t.b_vo = nx ? tn.b_vo : tc.b_vo;
    t.mean = nx ? tn.mean : tc.mean;
    t.rstd = nx ? tn.rstd : tc.rstd;
    return t;
  };
  auto step = [&](Regs& nx, int kt, auto full_tag, auto have0_tag, auto pref_tag) __attribute__((always_inline)) {
    // (NP 3: always the just-in-time flavour -- the h fragments of sub-step 0 are needed again with the m fragments)
    constexpr bool HAVE0 = NP == 3 ? false : decltype(have0_tag)::value, PREF = NP == 3 ? false : decltype(pref_tag)::value;
    const int s1 = s0 == W_NSTAGE - 1 ? 0 : s0 + 1, s2 = s1 == W_NSTAGE - 1 ? 0 : s1 + 1;
    int k1, k2, k3;
    const TileP t1 = pick(kt + 1, k1), t2 = pick(kt + 2, k2), t3 = pick(kt + 3, k3);
    if constexpr (!HAVE0) read_frags(f0, s0, 0, full_tag);
    if constexpr (HAVE0) read_frags(f1, s0, 1, full_tag);
    lds_store(nx, t1, k1, s1);
    gload_a(t2, k2, s2);
    gload_b(nx, t3, k3);
    if constexpr (NP == 3) {
      mma3_a(f0, full_tag);
      read_frags(f1, s0, 1, full_tag);
      mma3_b(f0, f1, full_tag);
    } else {
      mma(f0, full_tag);
      if constexpr (!HAVE0) {
        read_frags(f1, s0, 1, full_tag);
        mma(f1, full_tag);
      }
    }
    // (in flight behind the DMA of k-tile kt + 1: last step's activation loads, this step's DMA and activation loads)
    if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (PREF) read_frags(f0, s1, 0, full_tag);     // next k-tile's first half, under this one's second
    if constexpr (HAVE0) mma(f1, full_tag);
    s0 = s1;
  };

  TileCur cur, nxc;
  {
    cur_set(cur, 0);
    nxc = cur;
    if (ntile > 1) cur_next(nxc);
    else nxc.v = -1;
  }
  tc = make_tile(cur);
  tn = nxc.v >= 0 ? make_tile(nxc) : tc;   // past the last tile the pipeline re-reads that tile (harmless)
  Regs r0, r1;
  gload_a(tc, 0, 0);
  gload_b(r0, tc, 0);                   // k-tile 0 -> stage 0 (A), r0 (B)
  gload_a(tc, 1, 1);
  gload_b(r1, tc, 1);                   // k-tile 1 -> stage 1,     r1     (nk >= 4)
  lds_store(r0, tc, 0, 0);              // B of k-tile 0 -> stage 0
  gload_b(r0, tc, 2);                   // k-tile 2 -> r0
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // A of k-tiles 0 and 1 landed
  __builtin_amdgcn_s_barrier();

  using T = std::true_type;
  using F = std::false_type;
  for (int i = 0; cur.v >= 0; ++i) {
    const bool quarter = cur.q >= 0;
    set_off(quarter);
    const int m0 = cur.mt * W_BM, v = cur.v;
    const int l0 = cur.lt * W_BN + (quarter ? cur.q * 32 : 0);
    const long b = cur.b;
    const int NT = quarter ? 1 : 2;
    const int mrow = quarter ? m0 + wave * 32 : m0 + wm * 64;     // first output row / column of this wavefront
    const int lcol = quarter ? l0 : l0 + wn * 64;
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;
    const float* ext = nullptr;
    int extC = 1;
    if constexpr (EPI == 1) {
      ext = a.residual;
      extC = a.Cout;
    } else if constexpr (EPI == 2) {
      ext = a.mul;
      extC = a.mul_channels;
    } else if constexpr (EPI == 3) {
      ext = a.residual ? a.residual : ((a.epi_mask & 1) ? a.mul : nullptr);
      extC = a.residual ? a.Cout : a.mul_channels;
    }
    const bool is_res = EPI == 1 || (EPI == 3 && a.residual != nullptr);
    const bool is_mask = EPI == 2 || (EPI == 3 && !a.residual && (a.epi_mask & 1));
    const float* extb = ext ? ext + (size_t)b * extC * L : nullptr;
    float* yb = a.y + (size_t)b * a.Cout * L;
    float4 rext[kHasExt ? 2 : 1][kHasExt ? 2 : 1][kHasExt ? 4 : 1];   // [mi][ni][ii]
    float rbias[2][4];
    auto epi_row = [&](int mi, int ii, int& mc) __attribute__((always_inline)) {
      const int m = mrow + mi * 32 + ii * 8 + rsub;
      mc = m < a.Cout ? m : 0;
      return m < a.Cout;
    };
    auto epi_col = [&](int ni, int& lc) __attribute__((always_inline)) {
      const int l = lcol + ni * 32 + c4;
      lc = l < L ? l : 0;
      return l < L;
    };
    // What the epilogue reads from global memory: the bias and the UPPER half (mi = 0: 32 registers) of the residual / mask
    // multiplier are requested TWO STEPS before the tile's k-loop ends, so that they arrive under the last MFMAs; the lower
    // half (mi = 1) right after the last step, when the fragment registers are free -- it lands while the upper half's strips
    // are processed.  (All 64 registers two steps ahead, the round-2 form, spilled next to 64 accumulators + the fragments.)
    auto epi_issue = [&](auto half_tag) __attribute__((always_inline)) {
      constexpr int mi = decltype(half_tag)::value;
      {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          int mc, lc;
          epi_row(mi, ii, mc);
          if (mi == 0) {
            rbias[0][ii] = a.bias[mc];
            int mc1;
            epi_row(1, ii, mc1);
            rbias[1][ii] = a.bias[mc1];
          }
          if constexpr (kHasExt) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              epi_col(ni, lc);
              if (ext && mi < NT && ni < NT) {
                const int me = is_res ? mc : mc % extC;
                if constexpr (CP & 8) {
                  typedef float f32x4v __attribute__((ext_vector_type(4)));
                  const f32x4v t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(extb + me * L + lc));
                  rext[mi][ni][ii] = make_float4(t4[0], t4[1], t4[2], t4[3]);
                } else
                  rext[mi][ni][ii] = *reinterpret_cast<const float4*>(extb + me * L + lc);
              } else {
                rext[mi][ni][ii] = make_float4(0.f, 0.f, 0.f, 0.f);
              }
            }
          }
        }
      }
    };
    // EPI 4: the encoder multiplier in the MFMA C layout (register r of accumulator tile (mi, ni) = row (r & 3) + 8 (r >> 2) +
    // 4 (lane >> 5), column lane & 31), one dword per register; upper half (mi = 0) two steps ahead, lower half after the last step
    float rxc[EPI == 4 ? 2 : 1][EPI == 4 ? 2 : 1][EPI == 4 ? 16 : 1];
    // (buffer loads: ONE per-lane offset -- 4 rows down for the upper lane half, the column -- and the row as the scalar
    // offset; mul_channels % 8 == 0 and Cout % 8 == 0 (host) keep an 8-row group on one side of the wrap / the bound)
    auto epi4_issue = [&](auto half_tag) __attribute__((always_inline)) {
      if constexpr (EPI == 4) {
        constexpr int mi = decltype(half_tag)::value;
        __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.mul) + (size_t)b * a.mul_channels * L, 0, a.mul_channels * L * 4, 0x00020000);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int col = lcol + ni * 32 + (lane & 31);
          const int vo = (4 * (lane >> 5) * L + (col < L ? col : 0)) * 4;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row8 = mrow + mi * 32 + 8 * g;
            const int so = ((row8 < a.Cout ? row8 : 0) % a.mul_channels) * L * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              rxc[mi][ni][4 * g + j] = (mi < NT && ni < NT)
                                           ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrs, vo, so + j * L * 4, 0))
                                           : 0.f;
          }
        }
      }
    };
    // EPI 4: this wavefront's Wd^T fragments of 32-channel chunk c of the tile: [block][mb][ks][hi | lo], 1 KB each, lane l at
    // 16 l (L2-resident; chunk 0 is requested before the tile's last step, chunk c + 1 under chunk c's MFMAs)
    struct WdFrag {
      bf16x8 f[EPI == 4 ? 2 : 1][EPI == 4 ? 2 : 1];
    };
    WdFrag wd0, wd1;
    auto wd_load = [&](WdFrag& w, int c) __attribute__((always_inline)) {
      if constexpr (EPI == 4) {
        // (buffer loads: the lane offset is the only vector address; block / mb / fragment go into the scalar offset)
        __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wdpack), 0, nMt * 8 * 8192, 0x00020000);
        const int so = (((m0 >> 5) + c) * 2 + (wave >> 2)) * 4096;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int part = 0; part < 2; ++part)
            w.f[ks][part] = __builtin_bit_cast(bf16x8, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, so + (ks * 2 + part) * 1024, 0));
      }
    };
    auto k_loop = [&](auto full_tag) __attribute__((always_inline)) {
      read_frags(f0, s0, 0, full_tag);     // this tile's k-tile 0 (complete since the barrier that ended the previous tile)
      for (int kt = 0; kt + 4 < nk; kt += 2) {
        step(r1, kt, full_tag, T{}, T{});        // converts k-tile kt+1 (odd: r1), loads k-tile kt+3 into r1
        step(r0, kt + 1, full_tag, T{}, T{});    // converts k-tile kt+2 (even: r0), loads k-tile kt+4 into r0
      }
      step(r1, nk - 4, full_tag, T{}, T{});
      step(r0, nk - 3, full_tag, T{}, F{});      // last prefetching step: nothing requested for k-tile nk-2
      if constexpr (EPI == 4) epi4_issue(std::integral_constant<int, 0>{});
      else epi_issue(std::integral_constant<int, 0>{});
      step(r1, nk - 2, full_tag, F{}, F{});
      if constexpr (EPI == 4) {      // (just-in-time fragments in these steps leave room for the second half one step ahead)
        epi4_issue(std::integral_constant<int, 1>{});
        wd_load(wd0, 0);
      }
      step(r0, nk - 1, full_tag, F{}, F{});
      if constexpr (kHasExt) epi_issue(std::integral_constant<int, 1>{});
    };
    if (quarter)
      k_loop(F{});
    else
      k_loop(T{});
    // parameters of the tile after next (LDS table + integer arithmetic: no memory wait)
    TileCur nnc = nxc;
    if (i + 2 < ntile) cur_next(nnc);
    else nnc.v = -1;
    const bool has_nn = nnc.v >= 0 && nxc.v >= 0;
    const TileP tnn = has_nn ? make_tile(nnc) : tn;
    // epilogue through wave-private strips in the stage the tile's last k-tile has just freed
    const int free_stage = s0 == 0 ? W_NSTAGE - 1 : s0 - 1;
    if constexpr (EPI == 4) {
      // ================= K5 fused: out frames Z[(o, k), l] = sum_ci Wd[ci, (o, k)] * relu(y + bias)[ci, l] * enc[ci % N, l] =================
      // (improved_sudormrf.py:295-300 + the ConvTranspose1d of :272-279 as a frame GEMM.)  The tile's 256 x 128 masked values
      // exist only in the accumulators: 32-channel chunks of them go through a 16-KB LDS image (the exact layout of a B-operand
      // stage: [128 columns][32 k] bf16, hi | lo, swizzled) -- two buffers in the stage the last k-tile freed -- and ALL eight
      // wavefronts contract a chunk against the matching pre-packed Wd^T fragments (global, L2-resident) into ONE 32 x 32
      // accumulator each: wavefront w owns frame rows (w >> 2) * 32 .. and columns (w & 3) * 32 ...  Summing over the chunk
      // index IS the reduction over the tile's 256 channels: fixed order, no atomics.  The tile's [zM x 128] partial frames go to
      // zpart[b][mt]; the overlap-add kernel adds the nMt partials.  Instead of 128 KB of masked values a tile writes 21 KB.
      // (the lane id through an opaque copy: otherwise every LDS / store offset below is hoisted out of the tile loop as a
      // loop invariant and, with all 256 registers taken by the k-loop, spilled -- one scratch round trip per use)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int lhalf = lane_o >> 5, lcol32 = lane_o & 31;
      char* stage_base = smem + free_stage * W_STAGE;
      const int zmb = wave >> 2, znb = wave & 3;
      auto fused_tail = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int NTc = FULL ? 2 : 1;
        // (1) v = relu(acc + bias) * enc in place (bias: scalar loads, the row is wave-uniform up to the lane half)
#pragma unroll
        for (int mi = 0; mi < NTc; ++mi) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row8 = mrow + mi * 32 + 8 * g;
            const float* bp = bias_r + (row8 < a.Cout ? row8 : 0);        // 8 consecutive scalars (Cout % 8 == 0)
            float bl[4], bh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              bl[j] = bp[j];
              bh[j] = bp[4 + j];
            }
#pragma unroll
            for (int ni = 0; ni < NTc; ++ni)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                const float v = acc[mi][ni][r] + (lhalf ? bh[j] : bl[j]);
                acc[mi][ni][r] = fmaxf(v, 0.f) * rxc[mi][ni][r];
              }
          }
        }
        // (2) chunk loop: 8 x 32 channels = the tile's 256 rows
        f32x16 zacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) zacc[r] = 0.f;
        const bool z_active = FULL ? true : znb == 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          char* buf = stage_base + ((c & 1) ? 0 : 2 * W_A_IMG);      // even chunks: the stage's B region, odd: its A region
          // the owners write their 32 channels x (64 | 32) columns of chunk c (full tiles: the two wavefronts of row group
          // c >> 1, accumulator row half c & 1; quarter tiles: wavefront c)
          const bool owner = FULL ? wm == (c >> 1) : wave == c;
          if (owner) {
#pragma unroll
            for (int ni = 0; ni < NTc; ++ni) {
              const f32x16& t = FULL ? acc[c & 1][ni] : acc[0][0];
              const int row = FULL ? wn * 64 + ni * 32 + lcol32 : lcol32;         // image row = tile column
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 hi4, lo4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float v = t[4 * g + j];
                  const __bf16 h = (__bf16)v;
                  hi4[j] = h;
                  lo4[j] = (__bf16)(v - (float)h);
                }
                char* dst = buf + w_swz(row, g) + 8 * lhalf;
                *reinterpret_cast<bf16x4*>(dst) = hi4;
                *reinterpret_cast<bf16x4*>(dst + W_B_IMG) = lo4;
              }
            }
          }
          WdFrag& wc = (c & 1) ? wd1 : wd0;
          WdFrag& wn_ = (c & 1) ? wd0 : wd1;
          if (c + 1 < 8) wd_load(wn_, c + 1);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (z_active) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const int off = w_swz(znb * 32 + lcol32, 2 * ks + lhalf);
              const bf16x8 vh = *reinterpret_cast<const bf16x8*>(buf + off);
              const bf16x8 vl = *reinterpret_cast<const bf16x8*>(buf + W_B_IMG + off);
              zacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc.f[ks][1], vh, zacc, 0, 0, 0);
              zacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc.f[ks][0], vl, zacc, 0, 0, 0);
              zacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc.f[ks][0], vh, zacc, 0, 0, 0);
            }
          }
        }
        // (3) the tile's partial frames
        // (buffer stores: row in the scalar offset; lanes beyond L / zM get an offset past the descriptor's range = dropped)
        if (z_active) {
          float* zb = zpart + ((size_t)b * nMt + cur.mt) * (size_t)zM * L;
          __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(zb, 0, zM * L * 4, 0x00020000);
          const int col = l0 + znb * 32 + lcol32;
          const int vo = (4 * lhalf * L + col) * 4;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int mu = zmb * 32 + (r & 3) + 8 * (r >> 2);           // wave-uniform; + 4 for the upper lane half
            if (mu < zM) {
              const bool ok = col < L && mu + 4 * lhalf < zM;
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(zacc[r]), zrs, ok ? vo : 0x7ffffff0, mu * L * 4, 0);
            }
          }
        }
      };
      if (quarter)
        fused_tail(F{});
      else
        fused_tail(T{});
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
      cur = nxc;
      nxc = nnc;
      tc = tn;
      tn = tnn;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      continue;
    }
    float* strip = reinterpret_cast<float*>(smem + free_stage * W_STAGE) + wave * (32 * SRF_EPI_PITCH_H);
    float s = 0.f, q = 0.f;
    {
      const int col = lane & 31, kh = lane >> 5;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if (mi < NT && ni < NT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * kh) * SRF_EPI_PITCH_H + col] = acc[mi][ni][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
              float4 o = *reinterpret_cast<const float4*>(strip + (ii * 8 + rsub) * SRF_EPI_PITCH_H + c4);
              const float bs = rbias[mi][ii];
              if constexpr (NP == 4) {       // (the weights were stored times 2^4)
                constexpr float inv = 1.f / W_F16_WSCALE;
                o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
              }
              o.x += bs; o.y += bs; o.z += bs; o.w += bs;
              if constexpr (kHasExt) {
                const float4 e = rext[mi][ni][ii];
                if (is_res) {
                  o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
                } else if (is_mask) {
                  o.x = fmaxf(o.x, 0.f) * e.x;
                  o.y = fmaxf(o.y, 0.f) * e.y;
                  o.z = fmaxf(o.z, 0.f) * e.z;
                  o.w = fmaxf(o.w, 0.f) * e.w;
                }
              }
              int mc, lc;
              const bool okr = epi_row(mi, ii, mc), okc = epi_col(ni, lc);
              if (okr && okc) {
                if constexpr ((CP & 3) == 0) {
                  *reinterpret_cast<float4*>(yb + mc * L + lc) = o;
                } else {
                  // cache-policy experiment: the output tensor is written once and read by a LATER kernel -- keep it from
                  // displacing the L2-resident weight image that every CU re-reads for every tile
                  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                  const u32x4 ov = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                  __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb, 0, a.Cout * L * 4, 0x00020000);
                  __builtin_amdgcn_raw_buffer_store_b128(ov, yrs, (mc * L + lc) * 4, 0, (CP & 3) == 1 ? 2 : (CP & 3) == 2 ? 16 : 18);
                }
                s += (o.x + o.y) + (o.z + o.w);
                q = fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, fmaf(o.w, o.w, q))));
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
        }
    }
    if (a.out_sums) {
      const double ds = srf_dpp_wave_sum((double)s), dq = srf_dpp_wave_sum((double)q);
      if (lane == 63) {
        double* dst = srf_stat_slot(a.out_sums, b, (long)v * 32 + wave + (quarter ? 8 * (cur.q + 1) : 0));
        atomicAdd(dst, ds);
        atomicAdd(dst + 1, dq);
      }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    cur = nxc;
    nxc = nnc;
    tc = tn;
    tn = tnn;
    // strip reads done before the next step's DMA overwrites that stage
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // surplus DMA of the pipeline tail must not outlive the block's LDS
}

bool srf_x3w_supported(int Bt, int pro) { return !(pro == 1 || pro == 2) || Bt <= W_MAX_STAT_EXAMPLES; }

// ---- weight packing: W[Cout][Cin] fp32 -> per (m-tile, k-tile) [hi image | lo image], rows >= Cout zero ----------------------
bool srf_x3w_shape_supported(int Cin, int Cout, int L) {
  return (Cin % 64 == 0) && Cin >= 128 && (L % 4 == 0) && Cout >= 192 && (long)Cout * Cin * 4 < (1L << 31);
}
size_t srf_x3w_packed_bytes(int Cout, int Cin) {
  const size_t nMt = (Cout + W_BM - 1) / W_BM, nKt = Cin / W_BK;
  return nMt * nKt * (size_t)W_WTILE_BYTES;
}
constexpr int SRF_W_MAX_PACK = 96;
struct WPackEntry {
  const float* w;
  char* dst;
  char* dst16;     // optional second image of the same parts for the paired-block kernel (srf_pwconv_x3p.hip): per (m-tile, 16-k
                   // step) [256 rows][hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15], same swizzle -- written by the same launch
  int Cout, Cin;   // Cin < 0: w is stored TRANSPOSED ([|Cin|][Cout]) -- the backward's data-gradient GEMMs use W^T
};
struct WPackTable {
  WPackEntry e[SRF_W_MAX_PACK];
};
// F16: fp16 parts of 2^4 w (NP 4) instead of bf16 parts of w
template <bool F16>
__global__ __launch_bounds__(256) void srf_x3w_pack_kernel(WPackTable t) {
  WPackEntry e = t.e[blockIdx.y];
  const bool trans = e.Cin < 0;
  e.Cin = trans ? -e.Cin : e.Cin;
  const int nKt = e.Cin / W_BK;
  const int nMt = (e.Cout + W_BM - 1) / W_BM;
  const long total = (long)nMt * nKt * W_BM * 4;   // one thread per 8-k packet
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int c = (int)(id & 3);
    const int row = (int)((id >> 2) % W_BM);
    const long tile = (id >> 2) / W_BM;
    const int kt = (int)(tile % nKt), mt = (int)(tile / nKt);
    const int m = mt * W_BM + row;
    bf16x8 hi, lo;
    float vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kt * W_BK + c * 8 + j;
      const float v = (m < e.Cout) ? (trans ? e.w[(size_t)k * e.Cout + m] : e.w[(size_t)m * e.Cin + k]) : 0.f;
      vv[j] = F16 ? v * W_F16_WSCALE : v;
    }
    if constexpr (F16) w_split8_f16(vv, hi, lo);
    else w_split8(vv, hi, lo);
    char* base = e.dst + (size_t)tile * W_WTILE_BYTES + w_swz(row, c);
    *reinterpret_cast<bf16x8*>(base) = hi;
    *reinterpret_cast<bf16x8*>(base + W_A_IMG) = lo;
    if (e.dst16) {              // 16-k step 2 kt + (c >> 1) of the m-tile, 8-k packet c & 1
      char* b16 = e.dst16 + ((size_t)mt * (2 * nKt) + 2 * kt + (c >> 1)) * (size_t)(W_BM * 64);
      *reinterpret_cast<bf16x8*>(b16 + w_swz(row, c & 1)) = hi;
      *reinterpret_cast<bf16x8*>(b16 + w_swz(row, 2 + (c & 1))) = lo;
    }
  }
}
static int srf_x3w_pack_launch_any(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n, hipStream_t st,
                                   bool f16, char* const* dst16 = nullptr) {
  for (int base = 0; base < n; base += SRF_W_MAX_PACK) {
    WPackTable t;
    const int cnt = (n - base) < SRF_W_MAX_PACK ? (n - base) : SRF_W_MAX_PACK;
    for (int i = 0; i < SRF_W_MAX_PACK; ++i) {
      const int j = base + (i < cnt ? i : 0);
      t.e[i] = WPackEntry{w[j], dst[j], dst16 ? dst16[j] : nullptr, Cout[j], Cin[j]};
    }
    if (f16) hipLaunchKernelGGL(srf_x3w_pack_kernel<true>, dim3(64, cnt), dim3(256), 0, st, t);
    else hipLaunchKernelGGL(srf_x3w_pack_kernel<false>, dim3(64, cnt), dim3(256), 0, st, t);
    SRF_CHECK_LAUNCH(f16 ? "pack_pw_weights_f16" : "pack_pw_weights", st);
  }
  return SRF_OK;
}
int srf_x3w_pack_launch(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n, hipStream_t st) {
  return srf_x3w_pack_launch_any(w, dst, Cout, Cin, n, st, false);
}
// both images of every entry in ONE launch: dst (this kernel's) and dst16 (the paired-block kernel's)
int srf_x3w_pack2_launch(const float* const* w, char* const* dst, char* const* dst16, const int* Cout, const int* Cin, int n,
                         hipStream_t st) {
  return srf_x3w_pack_launch_any(w, dst, Cout, Cin, n, st, false, dst16);
}
// fp16 parts (NP 4): the same image layout and size as the bf16 two-part image; dst16 (optional): the paired-block layout of the
// same parts behind it (what the fused pair of the training forward streams, srf_pwconv_x3f.hip F16)
int srf_x3w_pack_f16_launch(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n, hipStream_t st,
                            char* const* dst16) {
  return srf_x3w_pack_launch_any(w, dst, Cout, Cin, n, st, true, dst16);
}

// fuse_wd != null: the mask epilogue fused with the decoder's contraction (EPI 4; pro 3, mask epilogue, fuse_M <= 64)
static int srf_pw_x3w_launch_any(const PwArgs& a, const char* wpack, int pro, const char* fuse_wd, float* fuse_z, int fuse_M,
                                 hipStream_t st, int np = 2);
int srf_pw_x3w_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st) {
  return srf_pw_x3w_launch_any(a, wpack, pro, nullptr, nullptr, 0, st);
}
// three-part operands, six MFMAs per product block (the training forward); wpack3: srf_x3w_pack3_launch's image
// two fp16 parts, three MFMAs per product block (the training forward's default); wpack: srf_x3w_pack_f16_launch's image
int srf_pw_x3w4_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st) {
  return srf_pw_x3w_launch_any(a, wpack, pro, nullptr, nullptr, 0, st, 4);
}
int srf_pw_x3w3_launch(const PwArgs& a, const char* wpack3, int pro, hipStream_t st) {
  return srf_pw_x3w_launch_any(a, wpack3, pro, nullptr, nullptr, 0, st, 3);
}

// ---- K5: decoder weights as MFMA A fragments.  w: the ConvTranspose1d weight [Ci][M] (M = sources x taps <= 64), split into
// bf16 hi | lo.  Image: [ceil(Ci / 256) * 8 blocks of 32 channels][mb 0..1][ks 0..1][hi | lo][lane 0..63][8 bf16]: lane l of
// fragment (block, mb, ks) holds Wd[block * 32 + 16 ks + 8 (l >> 5) + 0..7][mb * 32 + (l & 31)]; zero beyond Ci / M.
__global__ __launch_bounds__(256) void srf_x3w_pack_dec_kernel(const float* __restrict__ w, char* __restrict__ dst, int Ci, int M,
                                                               int nblocks) {
  const int idx = blockIdx.x * 256 + threadIdx.x;          // one thread per (block, mb, ks, lane): 8 values, both parts
  if (idx >= nblocks * 4 * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) & 1, mb = (idx >> 7) & 1, blk = idx >> 8;
  const int m = mb * 32 + (lane & 31);
  bf16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = blk * 32 + 16 * ks + 8 * (lane >> 5) + e;
    const float v = (ci < Ci && m < M) ? w[(size_t)ci * M + m] : 0.f;
    const __bf16 h = (__bf16)v;
    hi[e] = h;
    lo[e] = (__bf16)(v - (float)h);
  }
  char* o = dst + ((size_t)(blk * 2 + mb) * 4 + ks * 2) * 1024 + lane * 16;
  *reinterpret_cast<bf16x8*>(o) = hi;
  *reinterpret_cast<bf16x8*>(o + 1024) = lo;
}

size_t srf_x3w_dec_pack_bytes(int Ci) { return (size_t)((Ci + W_BM - 1) / W_BM) * 8 * 8192; }

int srf_x3w_pack_dec_launch(const float* w, void* dst, int Ci, int M, hipStream_t st) {
  SRF_CHECK_ARG(w && dst && Ci > 0 && M > 0 && M <= 64, "srf_pack_decoder: bad arguments");
  const int nblocks = (Ci + W_BM - 1) / W_BM * 8;
  const int threads = nblocks * 4 * 64;
  hipLaunchKernelGGL(srf_x3w_pack_dec_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, w, (char*)dst, Ci, M, nblocks);
  SRF_CHECK_LAUNCH("pack_decoder", st);
  return SRF_OK;
}

// ---- three-part weights (NP 3): per (m-tile, 16-k tile) the stage image of srf_x3w_pack_kernel -- [256 rows][32 slots] bf16,
// XOR-swizzled 16-byte chunks, "hi" image then "lo" image -- with h | m of the tile's 16 k in the hi image (chunks 0-1 | 2-3)
// and l in chunks 0-1 of the lo image (chunks 2-3: zero, never read).  w = h + m + l to 24 bits.
struct W3PackEntry {
  const float* w;
  char* dst;
  int Cout, Cin;
};
constexpr int SRF_W3_MAX_PACK = 48;
struct W3PackTable {
  W3PackEntry e[SRF_W3_MAX_PACK];
};
__global__ __launch_bounds__(256) void srf_x3w_pack3_kernel(W3PackTable t) {
  const W3PackEntry e = t.e[blockIdx.y];
  const int nKt = e.Cin / 16;
  const int nMt = (e.Cout + W_BM - 1) / W_BM;
  const long total = (long)nMt * nKt * W_BM * 2;   // one thread per (row, 8-k packet)
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int c = (int)(id & 1);
    const int row = (int)((id >> 1) % W_BM);
    const long tile = (id >> 1) / W_BM;
    const int kt = (int)(tile % nKt), mt = (int)(tile / nKt);
    const int m = mt * W_BM + row;
    bf16x8 ph, pm, pl, zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (m < e.Cout) ? e.w[(size_t)m * e.Cin + kt * 16 + c * 8 + j] : 0.f;
      const __bf16 h = (__bf16)v;
      const float r1 = v - (float)h;
      const __bf16 mm = (__bf16)r1;
      ph[j] = h;
      pm[j] = mm;
      pl[j] = (__bf16)(r1 - (float)mm);
      zero[j] = (__bf16)0.f;
    }
    char* base = e.dst + (size_t)tile * W_WTILE_BYTES;
    *reinterpret_cast<bf16x8*>(base + w_swz(row, c)) = ph;
    *reinterpret_cast<bf16x8*>(base + w_swz(row, 2 + c)) = pm;
    *reinterpret_cast<bf16x8*>(base + W_A_IMG + w_swz(row, c)) = pl;
    *reinterpret_cast<bf16x8*>(base + W_A_IMG + w_swz(row, 2 + c)) = zero;
  }
}
size_t srf_x3w_packed3_bytes(int Cout, int Cin) {
  return (size_t)((Cout + W_BM - 1) / W_BM) * (size_t)(Cin / 16) * (size_t)W_WTILE_BYTES;
}
int srf_x3w_pack3_launch(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n, hipStream_t st) {
  for (int base = 0; base < n; base += SRF_W3_MAX_PACK) {
    W3PackTable t;
    const int cnt = (n - base) < SRF_W3_MAX_PACK ? (n - base) : SRF_W3_MAX_PACK;
    for (int i = 0; i < SRF_W3_MAX_PACK; ++i) {
      const int j = base + (i < cnt ? i : 0);
      t.e[i] = W3PackEntry{w[j], dst[j], Cout[j], Cin[j]};
    }
    hipLaunchKernelGGL(srf_x3w_pack3_kernel, dim3(64, cnt), dim3(256), 0, st, t);
    SRF_CHECK_LAUNCH("pack_pw_weights3", st);
  }
  return SRF_OK;
}

// The model's tail in one GEMM launch: partial decoder frames zpart[Bt][ceil(Cout / 256)][M][L] instead of the masked tensor
int srf_pw_x3w_fused_tail_launch(const PwArgs& a, const char* wpack, const char* wdpack, float* zpart, int M, hipStream_t st) {
  SRF_CHECK_ARG(wdpack && zpart && M > 0 && M <= 64, "srf_mask_decode: bad arguments");
  SRF_CHECK_ARG(a.nrm.prelu && !a.nrm.sums && (a.epi_mask & 1) && !a.residual && !a.out_sums && a.mul,
                "srf_mask_decode: needs the PReLU prologue and the mask epilogue");
  return srf_pw_x3w_launch_any(a, wpack, 3, wdpack, zpart, M, st);
}

static int srf_pw_x3w_launch_any(const PwArgs& a, const char* wpack, int pro, const char* fuse_wd, float* fuse_z, int fuse_M,
                                 hipStream_t st, int np) {
  const int nMt = (a.Cout + W_BM - 1) / W_BM, nLt = (a.L + W_BN - 1) / W_BN;
  const long total = (long)a.Bt * nMt * nLt;
  SRF_CHECK_ARG(total < (1L << 31), "srf_pw_conv: too many tiles");
  SRF_CHECK_ARG((long)a.Bt * a.Cin * a.L * 4 < (1L << 31), "srf_pw_conv: activation tensor too large for buffer loads");
  SRF_CHECK_ARG(srf_x3w_supported(a.Bt, pro), "srf_pw_conv: too many examples for the statistics table");
  const size_t lds = (size_t)W_NSTAGE * W_STAGE + ((pro == 1 || pro == 2) ? (size_t)a.Bt * sizeof(float2) : 0);
  // dynamic LDS beyond 64 KB needs the attribute once per device (per-device cache, srf_common.h)
  const long ok = srf_device_cached(3, [](void*) -> long {
    const int bytes = W_NSTAGE * W_STAGE + W_MAX_STAT_EXAMPLES * (int)sizeof(float2);
    bool good = true;
    const void* fns[] = {
        // the forms the models launch, each with its cache policy (see the launch function)
        (const void*)&srf_pw_x3w_kernel<0, 0, 0>, (const void*)&srf_pw_x3w_kernel<1, 0, 5>,
        (const void*)&srf_pw_x3w_kernel<2, 1, 5>, (const void*)&srf_pw_x3w_kernel<3, 2, 5>,
        (const void*)&srf_pw_x3w_kernel<3, 4, 4>, (const void*)&srf_pw_x3w_kernel<0, 1, 0>,
        // two fp16 parts per operand (training forward, the default)
        (const void*)&srf_pw_x3w_kernel<0, 0, 0, 4>, (const void*)&srf_pw_x3w_kernel<1, 0, 5, 4>,
        (const void*)&srf_pw_x3w_kernel<2, 1, 5, 4>, (const void*)&srf_pw_x3w_kernel<3, 0, 5, 4>,
        // three bf16 parts per operand (training forward, debug flag 16384)
        (const void*)&srf_pw_x3w_kernel<0, 0, 0, 3>, (const void*)&srf_pw_x3w_kernel<1, 0, 5, 3>,
        (const void*)&srf_pw_x3w_kernel<2, 1, 5, 3>, (const void*)&srf_pw_x3w_kernel<3, 0, 5, 3>,
        // any other prologue / epilogue combination (unit tests, stand-alone srf_pw_conv callers)
        (const void*)&srf_pw_x3w_kernel<0, 3>, (const void*)&srf_pw_x3w_kernel<1, 3>,
        (const void*)&srf_pw_x3w_kernel<2, 3>, (const void*)&srf_pw_x3w_kernel<3, 3>};
    for (const void* f : fns) good &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    return good ? 1 : 0;
  }, nullptr);
  SRF_CHECK_ARG(ok == 1, "srf_pw_conv: cannot reserve %zu bytes of LDS", lds);
  long nb = srf_device_cus();
  nb -= nb % 8;
  if (nb < 8) nb = 8;
  if (nb > total) nb = total - total % 8;   // (the tile cursors need nb % 8 == 0; the host dispatches this kernel for total >= #CUs)
  SRF_CHECK_ARG(nb >= 8, "srf_pw_conv: too few tiles for the 256 x 128 kernel");
  const int rounds = (srf_debug_flags() & 256) ? (int)((total + nb - 1) / nb) : (int)(total / nb);
  // m-tile groups (see the tile cursors): as many m-tiles as keep their packed weight slabs (256 x Cin x 4 B each) within half
  // of an XCD's 4-MB L2, when the whole weight image does not fit it.  Debug flag 2 (the round-2 kernel's fragment-read order,
  // unused by this kernel) = one group = the round-2 tile order, for A/B.
  int mgrp = nMt;
  {
    const long slab = (long)W_BM * a.Cin * 4 * (np == 3 ? 2 : 1), image = slab * nMt;
    if (image > (3L << 20) && !(srf_debug_flags() & 2)) {
      int g = (int)((2L << 20) / slab);
      while (g > 1 && nMt % g) --g;
      if (g >= 1 && g < nMt) mgrp = g;
    }
  }
  dim3 grid((unsigned)nb), block(512);
  PwArgs ap = a;
  if (!(srf_debug_flags() & 512)) ap.epi_mask |= 1 << 12;   // quarter tiles first (flag 512: last)
  const bool res = a.residual != nullptr, mask = !res && (a.epi_mask & 1);
#define W_GO(P, E, C) hipLaunchKernelGGL((srf_pw_x3w_kernel<P, E, C>), grid, block, lds, st, ap, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta, a.bias, fuse_wd, fuse_z, fuse_M, mgrp)
  // Cache policy (CP) of the four model forms: bit 0 = non-temporal output stores, bit 2 = non-temporal activation loads,
  // bit 3 = non-temporal residual / mask-multiplier loads.  Every CU re-reads the whole packed weight image from L2 for every
  // tile, while activations, residuals and outputs stream through once: marked non-temporal they stop displacing the weights.
  // Defaults = what measured fastest INSIDE the forward (profiles/r03_NOTES.md: isolated launches mislead -- non-temporal
  // stores looked 10 % faster on proj_1x1 alone and cost 5 % in the model, where the next kernel reads that tensor): proj_1x1
  // plain (CP 0); bottleneck, res_conv, mask non-temporal activation loads + stores (CP 5; the fused mask + decoder GEMM
  // has no output tensor to mark: CP 4).
  if (np == 4) {
#define W_GO4(P, E, C) hipLaunchKernelGGL((srf_pw_x3w_kernel<P, E, C, 4>), grid, block, lds, st, ap, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta, a.bias, fuse_wd, fuse_z, fuse_M, mgrp)
    SRF_CHECK_ARG(!mask && !fuse_wd && (pro == 2) == res && !(pro != 2 && res), "srf_pw_conv (fp16 two-part): form not built");
    if (pro == 0) W_GO4(0, 0, 0);
    else if (pro == 1) W_GO4(1, 0, 5);
    else if (pro == 2) W_GO4(2, 1, 5);
    else W_GO4(3, 0, 5);
#undef W_GO4
    static const char* const kLabel4[4] = {"pw_conv_x3w4<0>", "pw_conv_x3w4<1>", "pw_conv_x3w4<2>", "pw_conv_x3w4<3>"};
    SRF_CHECK_LAUNCH(kLabel4[pro < 0 || pro > 3 ? 3 : pro], st);
    return SRF_OK;
  }
  if (np == 3) {
#define W_GO3(P, E, C) hipLaunchKernelGGL((srf_pw_x3w_kernel<P, E, C, 3>), grid, block, lds, st, ap, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta, a.bias, fuse_wd, fuse_z, fuse_M, mgrp)
    SRF_CHECK_ARG(!mask && !fuse_wd && (pro == 2) == res && !(pro != 2 && res), "srf_pw_conv (three-part): form not built");
    if (pro == 0) W_GO3(0, 0, 0);
    else if (pro == 1) W_GO3(1, 0, 5);
    else if (pro == 2) W_GO3(2, 1, 5);
    else W_GO3(3, 0, 5);
#undef W_GO3
    static const char* const kLabel3[4] = {"pw_conv_x3w3<0>", "pw_conv_x3w3<1>", "pw_conv_x3w3<2>", "pw_conv_x3w3<3>"};
    SRF_CHECK_LAUNCH(kLabel3[pro < 0 || pro > 3 ? 3 : pro], st);
    return SRF_OK;
  }
  if (fuse_wd) {
    W_GO(3, 4, 4);
    SRF_CHECK_LAUNCH("pw_mask_decode", st);
    return SRF_OK;
  }
  // the forms the models use are specialised on their epilogue; anything else runs the run-time-switched one
  if (pro == 0 && !res && !mask) W_GO(0, 0, 0);
  else if (pro == 1 && !res && !mask) W_GO(1, 0, 5);
  else if (pro == 2 && res) W_GO(2, 1, 5);
  else if (pro == 0 && res) W_GO(0, 1, 0);      // (the backward's data-gradient GEMM of proj_1x1: W^T g + skip gradient)
  else if (pro == 3 && mask) W_GO(3, 2, 5);
  else if (pro == 0) W_GO(0, 3, 0);
  else if (pro == 1) W_GO(1, 3, 0);
  else if (pro == 2) W_GO(2, 3, 0);
  else W_GO(3, 3, 0);
#undef W_GO
  static const char* const kLabel[4] = {"pw_conv_x3w<0>", "pw_conv_x3w<1>", "pw_conv_x3w<2>", "pw_conv_x3w<3>"};
  SRF_CHECK_LAUNCH(kLabel[pro < 0 || pro > 3 ? 3 : pro], st);
  return SRF_OK;
}
