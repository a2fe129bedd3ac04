// K2 (round-4 fast path) -- the 256 x 128 split-precision MFMA GEMM with its wavefronts SPLIT BY ROLE.
//
// Same arithmetic, tiles, packed weight image, LDS stage layout, work distribution and results (bit for bit) as
// srf_pwconv_x3w.hip: y = W f(x) + bias (+ residual | ReLU(.) x mul), x = hi + lo in bf16, three v_mfma_f32_32x32x16_bf16 per
// product block, fp32 accumulate (reference sites improved_sudormrf.py:256-259, :174, :196, :220, :268-269, :295-298).
//
// What round 3's kernel could not do, and the measurement that says so (profiles/r04_NOTES.md): with every
// MFMA compiled out it kept 81-93 % of its time (proj_1x1 93 of 115 us, res_conv 108 of 116) while drawing 1.05 kW at the full
// 2.4 GHz -- not power-bound; its memory skeleton is what is slow.  In that kernel each of the eight wavefronts did everything,
// in program order: wait for its activation loads, GlobLN / PReLU / split them (~100 VALU), write them to LDS, issue 12 memory
// instructions (each queues behind the other wavefronts' at the CU's one address pipe), 24 MFMAs, one barrier -- and the
// barrier keeps all eight in the same phase, so the matrix pipe idles while a SIMD's two wavefronts convert and queue, and
// the memory pipe idles while they multiply.  A step took ~4 000 cycles for 1 536 cycles of MFMA per SIMD.
//
// Here a block has TWELVE wavefronts (three per SIMD, <= 168 registers each):
//   * wavefronts 0..7 (two per SIMD) only read MFMA fragments from LDS and multiply (4 x 2 layout, 64 x 64 each, exactly the
//     accumulators of the old kernel), then run the tile's epilogue; their only global-memory instructions are the
//     epilogue's (bias / residual loads, output stores), so nothing in the k-loop ever waits behind a store;
//   * wavefronts 8..11 (one per SIMD) are LOADERS: the packed weight k-tiles by LDS-DMA (8 KB each per step), the activation
//     k-tile by 8-byte buffer loads (wavefront = 8 k rows x 128 columns, two adjacent columns per lane), GlobLN / PReLU /
//     hi | lo split, ds_write of the B images.  They never store to global memory: their counted vmcnt waits see loads only.
// The hardware interleaves the three instruction streams of a SIMD; one s_barrier per k-step (all twelve wavefronts) keeps
// the old stage protocol: at the barrier ending step kt the B image of k-tile kt+1 is written and its DMA has landed.
//
// Prologue / epilogue semantics: srf_pw.h (PwArgs).  PRO / EPI / CP as in srf_pwconv_x3w.hip.
#include <type_traits>

#include "../srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int S_BM = 256, S_BN = 128, S_BK = 32;
constexpr int S_A_IMG = S_BM * 64;                       // [256][32] bf16, 64-B rows
constexpr int S_B_IMG = S_BN * 64;                       // [128][32] bf16
constexpr int S_STAGE = 2 * S_A_IMG + 2 * S_B_IMG;       // A_hi | A_lo | B_hi | B_lo = 48 KB
constexpr int S_NSTAGE = 3;
constexpr int S_WTILE_BYTES = 2 * S_A_IMG;               // packed weights of one (m-tile, k-tile) (srf_x3v_pack_kernel's format)
constexpr int S_MAX_STAT_EXAMPLES = 1024;                // LDS statistics table: 8 KB behind the stages
constexpr int S_THREADS = 768, S_MMA_WAVES = 8;

__device__ __forceinline__ int s_swz(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

__device__ __forceinline__ void s_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

#define S_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// PRO: 0 = identity, 1 = GlobLN, 2 = GlobLN + PReLU, 3 = PReLU only.
// EPI: 0 = bias (+ statistics), 1 = bias + residual, 2 = ReLU(bias + .) x mul (mask epilogue), 3 = decided at run time (any).
// CP:  bit 0 = non-temporal output stores, bit 2 = non-temporal activation loads, bit 3 = non-temporal residual / multiplier loads.
// TL (diagnostics, tools/gemm_timeline_x3s.py): every wavefront sums the s_memtime ticks it spends in its barriers / in its
// explicit waits / in the epilogue and writes {total, barrier, wait, epilogue, steps} to `a.mul` (the trace buffer) at the end.
template <int PRO, int EPI, int CP = 0, int TL = 0>
__global__ __launch_bounds__(S_THREADS) void srf_pw_x3s_kernel(PwArgs a, const char* __restrict__ wpack, int nMt, int nLt,
                                                               int total, int rounds, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int mgrp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // S_NSTAGE * S_STAGE (+ the statistics table)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= S_MMA_WAVES;          // wave-uniform roles: 0..7 multiply, 8..9 stage activations, 10..11 weight DMA
  const bool xrole = wave == 8 || wave == 9;
  const int lw = wave & 1;                          // X loaders: k rows 16 lw .. 16 lw + 15 of the k-tile; DMA: half lw of the image
  const int wm = (wave >> 1) & 3, wn = wave & 1;    // multipliers: 4 x 2 wavefronts, 64 x 64 each
  const int Cin = a.Cin, L = a.L;
  const int nk = Cin / S_BK;                        // even, >= 4 (host checks)
  const int nblk = gridDim.x;
  const int nquart = 4 * (total - rounds * nblk);                                   // quarter tiles of the leftover round
  const int nq_mine = ((int)blockIdx.x < nquart) ? (nquart - (int)blockIdx.x + nblk - 1) / nblk : 0;
  const int ntile = rounds + nq_mine;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;
  const int x_bytes = a.Bt * Cin * L * 4;
  constexpr bool kHasExt = EPI != 0;   // the strip epilogue reads a second tensor (residual or mask multiplier)

  // ---- GlobLN statistics of every example, once per block: {mean, rstd} in LDS behind the stages (multiplier wavefront w
  // finalises examples 4w .. 4w+3 (+32 ...); the loaders read the table per tile: LDS, never behind a memory wait)
  float2* stat_tab = reinterpret_cast<float2*>(smem + S_NSTAGE * S_STAGE);
  if constexpr (PRO == 1 || PRO == 2) {
    if (!loader) {
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
    }
    __syncthreads();
  }

  // ---- tile order (srf_pwconv_x3w.hip): virtual id v = mi + G (lt + nLt (b + Bt grp)), m-tile = grp G + mi; quarter tiles
  // (256 x 32) of the leftover round first
  struct TileCur {
    int i, v, mt, lt, b, q, mi, grp;   // q: -1 = full tile, 0..3 = quarter of its parent tile; v < 0: no tile
  };
  const int G = mgrp;
  const int vstep = nblk >> 3;
  const int st_mi = vstep % G, st_r1 = vstep / G;
  const int st_l = st_r1 % nLt, st_r2 = st_r1 / nLt;
  const int st_b = st_r2 % a.Bt, st_g = st_r2 / a.Bt;
  const int qfirst = nq_mine;     // tiles [0, qfirst) are quarter tiles, then the full ones
  auto is_quarter = [&](int i) { return i < qfirst; };
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
      const int qi = blockIdx.x + i * nblk;
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

  // ---- X-loader geometry: lane -> columns 2 lane, 2 lane + 1 of the tile; wavefront -> k rows 16 lw .. 16 lw + 15 of the k-tile
  const int b_kg = lw * 16;
  __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
  struct TileP {
    const char* a_src;   // this DMA wavefront's 16-KB half of the tile's packed weights, k-tile 0 (wave-uniform)
    int b_vo;            // per-lane byte offset of (example, k row b_kg, column pair) inside X; out of range = fetch nothing
    float mean, rstd;    // GlobLN statistics of the tile's example (PRO 1 / 2)
  };
  auto make_tile = [&](const TileCur& c) {
    TileP t;
    t.a_src = wpack + (size_t)c.mt * nk * S_WTILE_BYTES + (loader ? lw : 0) * 16384;
    const int col0 = c.lt * S_BN + (c.q < 0 ? 0 : c.q * 32);
    const int width = c.q < 0 ? S_BN : 32;
    const int off = ((c.b * Cin + (loader ? b_kg : 0)) * L + min(col0 + 2 * lane, L - 2)) * 4;   // columns >= L are never stored
    t.b_vo = 2 * lane < width ? off : x_bytes;
    t.mean = 0.f;
    t.rstd = 1.f;
    if constexpr (PRO == 1 || PRO == 2) {
      const float2 mr = stat_tab[c.b];
      t.mean = mr.x;
      t.rstd = mr.y;
    }
    return t;
  };
  struct Regs {
    float b[32];     // [16 g + 2 j + c]: k row 8 g + j of the wavefront's 16, column c of the lane's pair
  };
  // A: LDS DMA through inline asm (invisible to hipcc's vmcnt bookkeeping: counted by hand, see lstep)
  auto gload_a = [&](const TileP& t, int kt, int stage) __attribute__((always_inline)) {
    const char* src = t.a_src + (size_t)kt * S_WTILE_BYTES + lane * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)S_LDS(smem + stage * S_STAGE + lw * 16384));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src + i * 1024), "s"(dst + i * 1024)
          : "memory");
    }
  };
  auto gload_b = [&](Regs& r, const TileP& t, int kt) __attribute__((always_inline)) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(b_rs, t.b_vo, (kt * S_BK + j) * L * 4, (CP & 4) ? 2 : 0);
      r.b[2 * j] = __uint_as_float(v[0]);
      r.b[2 * j + 1] = __uint_as_float(v[1]);
    }
  };
  // GlobLN / PReLU / split of k-tile kt (tile t) -> B images of `stage`: image row = tile column, 16-byte chunk = k-row group
  auto lds_store = [&](const Regs& r, const TileP& t, int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = r.b[16 * g + j];
      // opaque at the point of use: the conversion stays behind the previous step's barrier (srf_pwconv_x3w.hip)
      asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
      asm volatile("" : "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
      float sc[8], sh[8];
      if constexpr (PRO == 1 || PRO == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kt * S_BK + b_kg + 8 * g + j;
          sc[j] = gamma[k] * t.rstd;
          sh[j] = beta[k] - t.mean * sc[j];
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float vb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float x0 = x[2 * j + c];
          if constexpr (PRO == 1 || PRO == 2) x0 = fmaf(x0, sc[j], sh[j]);
          if constexpr (PRO == 2 || PRO == 3) x0 = srf_prelu(x0, slope);
          vb[j] = x0;
        }
        bf16x8 hi, lo;
        s_split8(vb, hi, lo);
        char* base = smem + stage * S_STAGE + 2 * S_A_IMG + s_swz(2 * lane + c, 2 * lw + g);
        *reinterpret_cast<bf16x8*>(base) = hi;
        *reinterpret_cast<bf16x8*>(base + S_B_IMG) = lo;
      }
    }
  };

  // ---- the two roles run DISJOINT instantiations of the tile loop (one generic lambda, the role a compile-time tag): in one
  // shared loop the accumulators / fragments of the multipliers and the load registers of the loaders are all live at every
  // join of the role branches, and 168 registers do not hold their union (first build: 215-615 spilled registers)
  auto run = [&](auto role_tag) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_tag)::value;      // 0 = multiply, 1 = activation staging, 2 = weight DMA
    constexpr bool LOADER = ROLE != 0;
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
          a_off[t][ks] = s_swz(quarter ? (wave & 7) * 32 + fr : wm * 64 + t * 32 + fr, 2 * ks + fc);
          b_off[t][ks] = 2 * S_A_IMG + s_swz(quarter ? fr : wn * 64 + t * 32 + fr, 2 * ks + fc);
        }
    };
    struct Frags {   // the fragments of one k-sub-step (16 of the k-tile's 32 k): 32 registers
      bf16x8 ah[2], al[2], bh[2], bl[2];   // [mi | ni]
    };
    auto read_frags = [&](Frags& f, int stage, int ks, auto full_tag) __attribute__((always_inline)) {
      constexpr int NT = decltype(full_tag)::value ? 2 : 1;
      const char* base = smem + stage * S_STAGE;
  #pragma unroll
      for (int t = 0; t < NT; ++t) {
        f.ah[t] = *reinterpret_cast<const bf16x8*>(base + a_off[t][ks]);
        f.bh[t] = *reinterpret_cast<const bf16x8*>(base + b_off[t][ks]);
        f.al[t] = *reinterpret_cast<const bf16x8*>(base + S_A_IMG + a_off[t][ks]);
        f.bl[t] = *reinterpret_cast<const bf16x8*>(base + S_B_IMG + b_off[t][ks]);
      }
    };
    auto mma = [&](const Frags& f, auto full_tag) __attribute__((always_inline)) {
      constexpr int NT = decltype(full_tag)::value ? 2 : 1;
      // pass-major order: independent accumulators between two MFMAs on the same one (and the summation order of x3v / x3w)
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
  #pragma unroll
      for (int ni = 0; ni < NT; ++ni)
  #pragma unroll
        for (int mi = 0; mi < NT; ++mi)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
    };

    // One pipeline step = k-tile kt of the current tile (in stage s0).
    //   loaders:     split k-tile kt+1 into stage s1, start the DMA of k-tile kt+2 into stage s2 and the activation loads of
    //                k-tile kt+3 (k-tile indices >= nk belong to the NEXT tile), wait until the DMA of k-tile kt+1 has landed;
    //   multipliers: k-tile kt.  Fragment schedule as in srf_pwconv_x3w.hip: HAVE0 = the first half's fragments were requested
    //                behind the previous step's barrier; PREF = request the next k-tile's behind this step's.
    // One barrier per step for all twelve wavefronts.
    int s0 = 0;
    TileP tc, tn;
    Frags f0, f1;
    unsigned tl_bar = 0, tl_wait = 0, tl_epi = 0, tl_steps = 0;
    const unsigned tl_t0 = TL ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    auto tl_now = [&]() __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_s_memtime(); };
    auto barrier_tl = [&]() __attribute__((always_inline)) {
      if constexpr (TL) {
        const unsigned t0 = tl_now();
        __builtin_amdgcn_s_barrier();
        tl_bar += tl_now() - t0;
        ++tl_steps;
      } else {
        __builtin_amdgcn_s_barrier();
      }
    };
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
    auto lstep = [&](Regs& nx, int kt) __attribute__((always_inline)) {
      const int s1 = s0 == S_NSTAGE - 1 ? 0 : s0 + 1, s2 = s1 == S_NSTAGE - 1 ? 0 : s1 + 1;
      int k1, k2, k3;
      const TileP t1 = pick(kt + 1, k1), t2 = pick(kt + 2, k2), t3 = pick(kt + 3, k3);
      if constexpr (ROLE == 1) {
        // (every memory instruction of this wavefront is a compiler-counted load: the wait hipcc places ahead of the
        // conversion -- vmcnt(16), the other register set may stay in flight -- is exact, the loads have two full steps)
        const unsigned w0 = TL ? tl_now() : 0u;
        lds_store(nx, t1, k1, s1);
        if constexpr (TL) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          tl_wait += tl_now() - w0;       // conversion incl. its wait for the loads
        }
        gload_b(nx, t3, k3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else {
        // (hand-counted: behind the DMA of k-tile kt + 1 -- issued a step ago, at that step's START -- only this step's 16
        // pieces are in flight: the weights have two full steps to land, and no compiler-placed wait can shorten that)
        gload_a(t2, k2, s2);
        const unsigned w0 = TL ? tl_now() : 0u;
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if constexpr (TL) tl_wait += tl_now() - w0;
      }
      barrier_tl();
      s0 = s1;
    };
    auto cstep = [&](auto full_tag, auto have0_tag, auto pref_tag) __attribute__((always_inline)) {
      constexpr bool HAVE0 = decltype(have0_tag)::value, PREF = decltype(pref_tag)::value;
      const int s1 = s0 == S_NSTAGE - 1 ? 0 : s0 + 1;
      if constexpr (!HAVE0) read_frags(f0, s0, 0, full_tag);
      if constexpr (HAVE0) read_frags(f1, s0, 1, full_tag);
      mma(f0, full_tag);
      if constexpr (!HAVE0) {
        read_frags(f1, s0, 1, full_tag);
        mma(f1, full_tag);
      }
      {
        const unsigned w0 = TL ? tl_now() : 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TL) tl_wait += tl_now() - w0;
      }
      barrier_tl();
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
    if constexpr (ROLE == 1) {
      gload_b(r0, tc, 0);                   // k-tile 0 -> r0
      gload_b(r1, tc, 1);                   // k-tile 1 -> r1     (nk >= 4)
      lds_store(r0, tc, 0, 0);              // B of k-tile 0 -> stage 0
      gload_b(r0, tc, 2);                   // k-tile 2 -> r0
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (ROLE == 2) {
      gload_a(tc, 0, 0);                    // k-tiles 0 and 1 -> stages 0 and 1
      gload_a(tc, 1, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    using T = std::true_type;
    using F = std::false_type;
    for (int i = 0; cur.v >= 0; ++i) {
      const bool quarter = cur.q >= 0;
      if constexpr (!LOADER) set_off(quarter);
      const int m0 = cur.mt * S_BM, v = cur.v;
      const int l0 = cur.lt * S_BN + (quarter ? cur.q * 32 : 0);
      const long b = cur.b;
      const int NT = quarter ? 1 : 2;
      const int mrow = quarter ? m0 + (wave & 7) * 32 : m0 + wm * 64;     // first output row / column of this wavefront
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
      // multiplier are requested before the tile's last step, the lower half (mi = 1) right after it, when the fragment
      // registers are free -- it lands while the upper half's strips are processed.
      auto epi_issue = [&](auto half_tag) __attribute__((always_inline)) {
        constexpr int mi = decltype(half_tag)::value;
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
      };
      auto k_loop = [&](auto full_tag) __attribute__((always_inline)) {
        if constexpr (LOADER) {
          for (int kt = 0; kt + 4 < nk; kt += 2) {
            lstep(r1, kt);        // converts k-tile kt+1 (odd: r1), loads k-tile kt+3 into r1
            lstep(r0, kt + 1);    // converts k-tile kt+2 (even: r0), loads k-tile kt+4 into r0
          }
          lstep(r1, nk - 4);
          lstep(r0, nk - 3);
          lstep(r1, nk - 2);
          lstep(r0, nk - 1);
        } else {
          read_frags(f0, s0, 0, full_tag);     // this tile's k-tile 0 (complete since the barrier that ended the previous tile)
          for (int kt = 0; kt + 4 < nk; kt += 2) {
            cstep(full_tag, T{}, T{});
            cstep(full_tag, T{}, T{});
          }
          cstep(full_tag, T{}, T{});
          cstep(full_tag, T{}, T{});
          cstep(full_tag, T{}, F{});      // last prefetching step: nothing requested for k-tile nk-1
          epi_issue(std::integral_constant<int, 0>{});
          cstep(full_tag, F{}, F{});
          if constexpr (kHasExt) epi_issue(std::integral_constant<int, 1>{});
        }
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
      // epilogue (multipliers) through wave-private strips in the stage the tile's last k-tile has just freed
      const int free_stage = s0 == 0 ? S_NSTAGE - 1 : s0 - 1;
      const unsigned tl_e0 = TL ? tl_now() : 0u;
      if constexpr (!LOADER) {
        float* strip = reinterpret_cast<float*>(smem + free_stage * S_STAGE) + wave * (32 * SRF_EPI_PITCH_H);
        float s = 0.f, q = 0.f;
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
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 ov = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb, 0, a.Cout * L * 4, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(ov, yrs, (mc * L + lc) * 4, 0, 2);
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
      }
      cur = nxc;
      nxc = nnc;
      tc = tn;
      tn = tnn;
      // strip reads done before the next step's DMA overwrites that stage
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (TL) tl_epi += tl_now() - tl_e0;
      barrier_tl();
    }
    if constexpr (TL) {
      unsigned* out = reinterpret_cast<unsigned*>(const_cast<float*>(a.mul)) + ((size_t)blockIdx.x * 12 + wave) * 8;
      if (lane == 0) {
        out[0] = tl_now() - tl_t0;
        out[1] = tl_bar;
        out[2] = tl_wait;
        out[3] = tl_epi;
        out[4] = tl_steps;
      }
    }
    if constexpr (ROLE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // surplus DMA of the pipeline tail must not outlive the block's LDS
  };
  if (!loader) run(std::integral_constant<int, 0>{});
  else if (xrole) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 2>{});
}

bool srf_x3s_supported(int Bt, int pro) { return !(pro == 1 || pro == 2) || Bt <= S_MAX_STAT_EXAMPLES; }

int srf_pw_x3s_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st) {
  const int nMt = (a.Cout + S_BM - 1) / S_BM, nLt = (a.L + S_BN - 1) / S_BN;
  const long total = (long)a.Bt * nMt * nLt;
  SRF_CHECK_ARG(total < (1L << 31), "srf_pw_conv: too many tiles");
  SRF_CHECK_ARG((long)a.Bt * a.Cin * a.L * 4 < (1L << 31), "srf_pw_conv: activation tensor too large for buffer loads");
  SRF_CHECK_ARG(srf_x3s_supported(a.Bt, pro), "srf_pw_conv: too many examples for the statistics table");
  const size_t lds = (size_t)S_NSTAGE * S_STAGE + ((pro == 1 || pro == 2) ? (size_t)a.Bt * sizeof(float2) : 0);
  // dynamic LDS beyond 64 KB needs the attribute once per device (per-device cache, srf_common.h)
  const long ok = srf_device_cached(4, [](void*) -> long {
    const int bytes = S_NSTAGE * S_STAGE + S_MAX_STAT_EXAMPLES * (int)sizeof(float2);
    bool good = true;
    const void* fns[] = {
        (const void*)&srf_pw_x3s_kernel<0, 0, 0>, (const void*)&srf_pw_x3s_kernel<1, 0, 5>,
        (const void*)&srf_pw_x3s_kernel<2, 1, 5>, (const void*)&srf_pw_x3s_kernel<3, 2, 5>,
        (const void*)&srf_pw_x3s_kernel<0, 1, 0>,
        (const void*)&srf_pw_x3s_kernel<0, 0, 0, 1>, (const void*)&srf_pw_x3s_kernel<2, 1, 5, 1>,
        (const void*)&srf_pw_x3s_kernel<0, 3>, (const void*)&srf_pw_x3s_kernel<1, 3>,
        (const void*)&srf_pw_x3s_kernel<2, 3>, (const void*)&srf_pw_x3s_kernel<3, 3>};
    for (const void* f : fns) good &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    return good ? 1 : 0;
  }, nullptr);
  SRF_CHECK_ARG(ok == 1, "srf_pw_conv: cannot reserve %zu bytes of LDS", lds);
  long nb = srf_device_cus();
  nb -= nb % 8;
  if (nb < 8) nb = 8;
  if (nb > total) nb = total - total % 8;   // (the tile cursors need nb % 8 == 0; the host dispatches this kernel for total >= #CUs)
  SRF_CHECK_ARG(nb >= 8, "srf_pw_conv: too few tiles for the 256 x 128 kernel");
  const int rounds = (int)(total / nb);
  int mgrp = nMt;     // m-tile groups where the packed weights exceed an XCD's L2 (srf_pwconv_x3w.hip)
  {
    const long slab = (long)S_BM * a.Cin * 4, image = slab * nMt;
    if (image > (3L << 20)) {
      int g = (int)((2L << 20) / slab);
      while (g > 1 && nMt % g) --g;
      if (g >= 1 && g < nMt) mgrp = g;
    }
  }
  dim3 grid((unsigned)nb), block(S_THREADS);
  const bool res = a.residual != nullptr, mask = !res && (a.epi_mask & 1);
#define S_GO(P, E, C) hipLaunchKernelGGL((srf_pw_x3s_kernel<P, E, C>), grid, block, lds, st, a, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta, mgrp)
  // the forms the models use are specialised on their epilogue and cache policy (srf_pwconv_x3w.hip: proj_1x1 plain;
  // bottleneck, res_conv, mask non-temporal activation loads + stores); anything else runs the run-time-switched epilogue
  const bool tl = getenv("SRF_X3S_TL") && atoi(getenv("SRF_X3S_TL"));
  if (tl && pro == 0 && !res && !mask) hipLaunchKernelGGL((srf_pw_x3s_kernel<0, 0, 0, 1>), grid, block, lds, st, a, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta, mgrp);
  else if (tl && pro == 2 && res) hipLaunchKernelGGL((srf_pw_x3s_kernel<2, 1, 5, 1>), grid, block, lds, st, a, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta, mgrp);
  else if (pro == 0 && !res && !mask) S_GO(0, 0, 0);
  else if (pro == 1 && !res && !mask) S_GO(1, 0, 5);
  else if (pro == 2 && res) S_GO(2, 1, 5);
  else if (pro == 0 && res) S_GO(0, 1, 0);      // (the backward's data-gradient GEMM of proj_1x1: W^T g + skip gradient)
  else if (pro == 3 && mask) S_GO(3, 2, 5);
  else if (pro == 0) S_GO(0, 3, 0);
  else if (pro == 1) S_GO(1, 3, 0);
  else if (pro == 2) S_GO(2, 3, 0);
  else S_GO(3, 3, 0);
#undef S_GO
  static const char* const kLabel[4] = {"pw_conv_x3w<0>", "pw_conv_x3w<1>", "pw_conv_x3w<2>", "pw_conv_x3w<3>"};
  SRF_CHECK_LAUNCH(kLabel[pro < 0 || pro > 3 ? 3 : pro], st);
  return SRF_OK;
}
