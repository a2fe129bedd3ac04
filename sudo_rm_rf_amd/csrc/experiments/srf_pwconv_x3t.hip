// K2 (round-4 fast path) -- split-precision MFMA GEMM for the 1x1 convolutions with ONE SIMD OF EVERY CU RESERVED FOR MEMORY WORK.
//
// y = W f(x) + bias (+ residual | ReLU(.) x mul), x = hi + lo in bf16, three v_mfma_f32_32x32x16_bf16 per product block, fp32
// accumulate: the arithmetic, the splits and the per-accumulator summation order of srf_pwconv_x3w.hip (results are bit-identical;
// reference sites improved_sudormrf.py:256-259, :174, :196, :220, :268-269, :295-298).
//
// Why a new structure (profiles/r04_NOTES.md; tools/probes/cu_stream_probe.hip).  Round 3's kernel ran at 0.40 of the HBM ceiling and
// kept 81-93 % of its time with every MFMA compiled out, at the full 2.4 GHz and 1.05 kW: not power-bound.  The probe shows what
// binds it: A WAVEFRONT'S VECTOR-MEMORY INSTRUCTIONS DO NOT ISSUE WHILE OTHER WAVEFRONTS OF ITS SIMD KEEP THE MATRIX PIPE BUSY.
// Two wavefronts issuing MFMAs back to back on a SIMD starve a third wavefront's LDS-DMA stream on that SIMD completely
// (0.001 of its stand-alone rate; s_setprio changes nothing); with the loaders on a SIMD of their own both sides run at full
// speed (loaders 1.2 steps / us of the GEMM's 48 KB per step, multipliers 83 % of their pipes).  So in every earlier form of this
// kernel -- all wavefronts doing everything (rounds 1-3), or loader wavefronts spread over the four SIMDs (srf_pwconv_x3s.hip,
// round 4, bit-identical and 5 % slower) -- a step cost (MFMA time) + (memory-instruction issue time), never the larger of the two.
//
// Here a block has eight wavefronts of 256 registers.  A block's wavefronts go to the SIMDs round-robin, so wavefronts 3 and 7 share
// one SIMD: they are the LOADERS; wavefronts 0, 1, 2, 4, 5, 6 (two on each of the other three SIMDs) are the MULTIPLIERS.
//   * tile 256 (M) x 192 (time) x 16 (k per step); multipliers as 2 (M) x 3 (N), 128 x 64 each = eight 32 x 32 accumulators
//     (128 registers), 12 fragment reads for 24 MFMAs per step, two fragment sets: the reads of step s run under the MFMAs of
//     step s-1.  In the k-loop a multiplier issues ds_read_b128, MFMA, one barrier -- nothing else.
//   * loaders: everything else.  Packed weights (16 KB per step) and RAW fp32 activations (12 KB per step) by LDS-DMA -- no load
//     lands in a register, every wait is hand-counted -- then GlobLN / PReLU / hi | lo split of the raw stage into the B image of
//     the next step (one loader per 8-k half; 24 values per lane and step, on a SIMD that runs no MFMA).
//   * LDS: A ring 3 x 16 KB, raw ring 3 x 12 KB, B images 2 x 12 KB, six epilogue strips, the statistics table: 143 KB.
//   * 192-column tiles also cut the weight traffic per output by a third (the weights are re-read from L2 for every tile).
//   * work: full tiles in rounds over the persistent blocks; the leftover tiles are cut into thirds (256 x 64: only the multipliers
//     of that column block work) dealt to all blocks first.
// Packed weights: srf_x3t_pack_kernel's image -- per (m-tile, 16-k step) [256 rows][hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15], 64-byte
// rows, 16-byte chunks XOR-swizzled exactly as the B image, so that a stage is one 16-KB DMA and both operands' fragments are
// conflict-free ds_read_b128.
// Prologue / epilogue semantics: srf_pw.h (PwArgs).
#include <type_traits>

#include "../srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int T_BM = 256, T_BN = 192, T_BK = 16;
constexpr int T_A_STAGE = T_BM * 64;                    // 16 KB
constexpr int T_B_STAGE = T_BN * 64;                    // 12 KB
constexpr int T_R_STAGE = T_BK * T_BN * 4;              // 12 KB raw fp32 [16 k][192 columns]
constexpr int T_NA = 3, T_NR = 3, T_NB = 2;
constexpr int T_OFF_A = 0, T_OFF_B = T_OFF_A + T_NA * T_A_STAGE, T_OFF_R = T_OFF_B + T_NB * T_B_STAGE;
constexpr int T_STRIP = 32 * SRF_EPI_PITCH_H * 4;       // 4608 B per multiplier
constexpr int T_OFF_STRIP = T_OFF_R + T_NR * T_R_STAGE, T_OFF_STAT = T_OFF_STRIP + 6 * T_STRIP;
constexpr int T_MAX_STAT_EXAMPLES = 1024;
constexpr int T_LDS_BYTES = T_OFF_STAT + T_MAX_STAT_EXAMPLES * 8;
constexpr int T_THREADS = 512;
static_assert(T_LDS_BYTES <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int t_swz(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

#define T_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// ---- packed weights ------------------------------------------------------------------------------------------------------
struct TPackEntry {
  const float* w;
  char* dst;
  int Cout, Cin;     // Cin < 0: w is [|Cin|][Cout] and the image is that of its transpose (the backward's data-gradient GEMMs)
};
constexpr int SRF_T_MAX_PACK = 48;
struct TPackTable {
  TPackEntry e[SRF_T_MAX_PACK];
};
__global__ __launch_bounds__(256) void srf_x3t_pack_kernel(TPackTable t) {
  TPackEntry e = t.e[blockIdx.y];
  const bool trans = e.Cin < 0;
  e.Cin = trans ? -e.Cin : e.Cin;
  const int nKt = e.Cin / T_BK;
  const int nMt = (e.Cout + T_BM - 1) / T_BM;
  const long total = (long)nMt * nKt * T_BM * 2;   // one thread per (row, 8-k packet): both parts
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int c = (int)(id & 1);
    const int row = (int)((id >> 1) % T_BM);
    const long tile = (id >> 1) / T_BM;
    const int kt = (int)(tile % nKt), mt = (int)(tile / nKt);
    const int m = mt * T_BM + row;
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kt * T_BK + c * 8 + j;
      const float v = (m < e.Cout) ? (trans ? e.w[(size_t)k * e.Cout + m] : e.w[(size_t)m * e.Cin + k]) : 0.f;
      const __bf16 h = (__bf16)v;
      hi[j] = h;
      lo[j] = (__bf16)(v - (float)h);
    }
    char* base = e.dst + (size_t)tile * T_A_STAGE;
    *reinterpret_cast<bf16x8*>(base + t_swz(row, c)) = hi;
    *reinterpret_cast<bf16x8*>(base + t_swz(row, 2 + c)) = lo;
  }
}
size_t srf_x3t_packed_bytes(int Cout, int Cin) {
  return (size_t)((Cout + T_BM - 1) / T_BM) * (size_t)(Cin / T_BK) * (size_t)T_A_STAGE;
}
int srf_x3t_pack_launch(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n, hipStream_t st) {
  for (int base = 0; base < n; base += SRF_T_MAX_PACK) {
    TPackTable t;
    const int cnt = (n - base) < SRF_T_MAX_PACK ? (n - base) : SRF_T_MAX_PACK;
    for (int i = 0; i < SRF_T_MAX_PACK; ++i) {
      const int j = base + (i < cnt ? i : 0);
      t.e[i] = TPackEntry{w[j], dst[j], Cout[j], Cin[j]};
    }
    hipLaunchKernelGGL(srf_x3t_pack_kernel, dim3(64, cnt), dim3(256), 0, st, t);
    SRF_CHECK_LAUNCH("pack_pw_weights", st);
  }
  return SRF_OK;
}

// ---- GEMM ------------------------------------------------------------------------------------------------------------------
// PRO: 0 = identity, 1 = GlobLN, 2 = GlobLN + PReLU, 3 = PReLU only.
// EPI: 0 = bias (+ statistics), 1 = bias + residual, 2 = ReLU(bias + .) x mul (mask epilogue), 3 = decided at run time (any).
// CP:  bit 0 = non-temporal output stores.
// TL (diagnostics): per-role s_memtime totals to `a.mul` (tools/gemm_timeline_x3s.py).
// ABL (diagnostics, results wrong): 1 = no conversion, 2 = no fragment reads, 4 = no MFMAs, 8 = no epilogue, 16 = no DMA.
template <int PRO, int EPI, int CP = 0, int TL = 0, int ABL = 0>
__global__ __launch_bounds__(T_THREADS, 2) void srf_pw_x3t_kernel(PwArgs a, const char* __restrict__ wpack, int nMt, int nLt,
                                                                  int total, int rounds, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int mgrp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = (wave & 3) == 3;              // wavefronts 3 and 7: one SIMD
  const int lq = wave >> 2;                         // loaders: 0 / 1 = k rows 0-7 / 8-15 of a step, pieces [8 lq, ..) / [6 lq, ..)
  const int mw = wave - (wave >> 2);                // multipliers: 0..5
  const int wm = mw / 3, wn = mw - 3 * wm;          // 2 (M) x 3 (N), 128 x 64 each
  const int Cin = a.Cin, L = a.L;
  const int nk = Cin / T_BK;                        // steps per tile, >= 8 (host checks)
  const int nblk = gridDim.x;
  const int nthird = 3 * (total - rounds * nblk);   // thirds of the leftover round's tiles
  const int nq_mine = ((int)blockIdx.x < nthird) ? (nthird - (int)blockIdx.x + nblk - 1) / nblk : 0;
  const int ntile = rounds + nq_mine;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;
  constexpr bool kHasExt = EPI != 0;

  // ---- GlobLN statistics of every example, once per block: {mean, rstd} in LDS (the loaders read the table per tile)
  float2* stat_tab = reinterpret_cast<float2*>(smem + T_OFF_STAT);
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

  // ---- tile order: virtual id v = mi + G (lt + nLt (b + Bt grp)), m-tile = grp G + mi (srf_pwconv_x3w.hip: G < nMt keeps a
  // group of weight slabs L2-resident per XCD where the whole image exceeds it); a block's thirds first, then its full tiles
  struct TileCur {
    int i, v, mt, lt, b, q, mi, grp;   // q: -1 = full tile, 0..2 = third of its parent tile; v < 0: no tile
  };
  const int G = mgrp;
  const int vstep = nblk >> 3;
  const int st_mi = vstep % G, st_r1 = vstep / G;
  const int st_l = st_r1 % nLt, st_r2 = st_r1 / nLt;
  const int st_b = st_r2 % a.Bt, st_g = st_r2 / a.Bt;
  const int qfirst = nq_mine;
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
  auto cur_set = [&](TileCur& c, int i) {
    int p = blockIdx.x + (i - qfirst) * nblk;
    c.q = -1;
    if (i < qfirst) {
      const int qi = blockIdx.x + i * nblk;
      p = rounds * nblk + qi / 3;
      c.q = qi - 3 * (qi / 3);
    }
    c.i = i;
    cur_from_v(c, srf_xcd_remap(p, total));
  };
  auto cur_next = [&](TileCur& c) {
    const int i = c.i + 1;
    if (i < qfirst || c.q >= 0) {
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
  auto tile_col0 = [&](const TileCur& c) { return c.lt * T_BN + (c.q < 0 ? 0 : c.q * 64); };
  auto tile_ncb = [&](const TileCur& c) {      // 64-column blocks of the tile that hold any column < L
    const int left = L - tile_col0(c);
    const int nb = left <= 0 ? 0 : (left + 63) >> 6;
    return c.q < 0 ? (nb < 3 ? nb : 3) : (nb < 1 ? nb : 1);
  };

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

  if (loader) {
    // =============================================== LOADERS ===============================================
    // Global step counter g (continues across tiles): A(g) -> ring slot g % 3, raw(g) -> slot g % 3, B image(g) -> slot g & 1.
    // In phase 2 g (see the multipliers' phase loop) a loader issues A(g + 2) and raw(g + 3), waits for what it issued a step ago
    // (A(g + 1), raw(g + 2)) and meets the barrier; in phase 2 g + 1 it only meets the barrier.  Ring safety: the slot of A(g + 2)
    // held A(g - 1), last read in phase 2 g - 1 (group 1's LOAD(g - 1)); the slot of raw(g + 3) held raw(g), last read there too.  (The multipliers turn raw(g + 1) into B image(g + 1) meanwhile: VALU and LDS work issues beside
    // MFMAs, only vector-memory instructions do not.  With the conversion on the loaders -- first version -- their SIMD, two
    // wavefronts, was the bottleneck: 1.5 us per step, the multipliers idle at the barrier for 40-50 % of the time.)
    struct LTile {
      const char* a_src;       // packed weights of the m-tile, step 0, this loader's 8 pieces (wave-uniform)
      const float* x_src;      // x + ((b Cin) L + col0): the tile's first raw row (wave-uniform)
      int cmax;                // last column offset (multiple of 4) a lane may fetch: L - 4 - col0 (>= 0), else clamped there
    };
    auto make_ltile = [&](const TileCur& c) {
      LTile t;
      t.a_src = wpack + (size_t)c.mt * nk * T_A_STAGE + lq * 8192;
      const int col0 = min(tile_col0(c), L - 4);
      t.x_src = a.x + ((size_t)c.b * Cin) * L + col0;
      t.cmax = L - 4 - col0;
      return t;
    };
    // raw piece p (0..11) = LDS bytes [1024 p, +1024) of the stage [16 k][192 columns] fp32; lane l -> byte 1024 p + 16 l ->
    // k = byte / 768, column = (byte % 768) / 4.  This loader's pieces: 6 lq .. 6 lq + 5.
    int r_k[6], r_c[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int byte = 1024 * (6 * lq + i) + 16 * lane;
      r_k[i] = byte / 768;
      r_c[i] = (byte - 768 * r_k[i]) >> 2;
    }
    auto dma = [&](const char* sbase, unsigned voff, unsigned dst) __attribute__((always_inline)) {
      if (ABL & 16) return;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(voff), "s"(sbase), "s"(dst)
                   : "memory");
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)T_LDS(smem));
    auto issue_a = [&](const LTile& t, int k, int g) __attribute__((always_inline)) {      // A of the tile's step k -> ring slot g % 3
      const char* src = t.a_src + (size_t)k * T_A_STAGE;
      const unsigned dst = lds0 + T_OFF_A + (g % T_NA) * T_A_STAGE + lq * 8192;
#pragma unroll
      for (int i = 0; i < 8; ++i) dma(src + i * 1024, lane * 16, dst + i * 1024);
    };
    auto issue_r = [&](const LTile& t, int k, int g) __attribute__((always_inline)) {      // raw rows of the tile's step k
      const char* src = reinterpret_cast<const char*>(t.x_src + (size_t)k * T_BK * L);
      const unsigned dst = lds0 + T_OFF_R + (g % T_NR) * T_R_STAGE + lq * 6144;
#pragma unroll
      for (int i = 0; i < 6; ++i) dma(src, (unsigned)(r_k[i] * L + min(r_c[i], t.cmax)) * 4u, dst + i * 1024);
    };
    TileCur cur, nxc;
    cur_set(cur, 0);
    nxc = cur;
    if (ntile > 1) cur_next(nxc);
    else nxc.v = -1;
    LTile tc = make_ltile(cur), tn = nxc.v >= 0 ? make_ltile(nxc) : tc;
    // (a step index k >= nk addresses the NEXT tile; past the last tile the pipeline re-reads that tile: harmless)
    auto pk = [&](int k, int& kk) -> const LTile& {
      const bool nx = k >= nk;
      kk = nx ? k - nk : k;
      return nx ? tn : tc;
    };
    // prologue: A(0), A(1), raw(0), raw(1), raw(2) in flight; the multipliers convert raw(0) between the two barriers
    int g = 0;
    {
      int kk;
      issue_a(tc, 0, 0);
      issue_r(tc, 0, 0);
      issue_a(tc, 1, 1);
      issue_r(tc, 1, 1);
      const LTile& t2 = pk(2, kk);
      issue_r(t2, kk, 2);
      asm volatile("s_waitcnt vmcnt(20)" ::: "memory");     // A(0), raw(0) landed (14 + 6 later pieces may fly)
      __builtin_amdgcn_s_barrier();                         // raw(0) is visible to the multipliers
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // A(1), raw(1) landed
      __builtin_amdgcn_s_barrier();                         // step 0 may start
    }
    for (int i = 0; cur.v >= 0; ++i) {
      for (int k = 0; k < nk; ++k, ++g) {
        int k2, k3;
        const LTile& t2 = pk(k + 2, k2);
        issue_a(t2, k2, g + 2);
        const unsigned w0 = TL ? tl_now() : 0u;
        // (behind A(g + 1), issued a step ago: raw(g + 2) and this phase's A(g + 2) = 14 pieces)
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        if constexpr (TL) tl_wait += tl_now() - w0;
        barrier_tl();       // phase 2 g (group 0 LOAD(g), group 1 COMPUTE(g - 1))
        const LTile& t3 = pk(k + 3, k3);
        issue_r(t3, k3, g + 3);
        // (behind raw(g + 2): A(g + 2) and this phase's raw(g + 3) = 14 pieces)
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        barrier_tl();       // phase 2 g + 1
      }
      TileCur nnc = nxc;
      if (i + 2 < ntile) cur_next(nnc);
      else nnc.v = -1;
      const bool has_nn = nnc.v >= 0 && nxc.v >= 0;
      cur = nxc;
      nxc = nnc;
      tc = tn;
      if (has_nn) tn = make_ltile(nnc);
    }
    barrier_tl();           // (group 1's last COMPUTE phase)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // surplus DMA of the pipeline tail must not outlive the block's LDS
  } else {
    // ============================================= MULTIPLIERS =============================================
    f32x16 acc[4][2];      // [t: 32-row block][u: 32-column block]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    const int fr = lane & 31, fc = lane >> 5;
    // fragment addresses inside a stage: A row = wm 128 + 32 t + fr (the swizzle term repeats every 16 rows), chunk fc (hi) /
    // 2 + fc (lo); B row (= column) wn 64 + 32 u + fr
    const int a_hi = t_swz(wm * 128 + fr, fc), a_lo = t_swz(wm * 128 + fr, 2 + fc);
    const int b_hi = t_swz(wn * 64 + fr, fc), b_lo = t_swz(wn * 64 + fr, 2 + fc);
    // ONE fragment set of a whole step (48 registers): a multiplier alternates a LOAD phase (the step's 12 fragment reads, its 8
    // values of the next step's conversion) with a COMPUTE phase (the step's 24 MFMAs, nothing else) -- see the phase loop.
    struct Frags {
      bf16x8 ah[4], al[4], bh[2], bl[2];
    };
    auto read_frags = [&](Frags& f, int g) __attribute__((always_inline)) {
      if (ABL & 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(f.ah[t]), "+v"(f.al[t]));
        asm volatile("" : "+v"(f.bh[0]), "+v"(f.bl[0]), "+v"(f.bh[1]), "+v"(f.bl[1]));
        return;
      }
      const char* pa = smem + T_OFF_A + (g % T_NA) * T_A_STAGE;
      const char* pb = smem + T_OFF_B + (g & 1) * T_B_STAGE;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f.bh[u] = *reinterpret_cast<const bf16x8*>(pb + b_hi + u * 2048);
        f.bl[u] = *reinterpret_cast<const bf16x8*>(pb + b_lo + u * 2048);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f.ah[t] = *reinterpret_cast<const bf16x8*>(pa + a_hi + t * 2048);
        f.al[t] = *reinterpret_cast<const bf16x8*>(pa + a_lo + t * 2048);
      }
    };
    auto mma = [&](const Frags& f) __attribute__((always_inline)) {
      if (ABL & 4) {
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(f.ah[t]), "v"(f.al[t]));
        asm volatile("" ::"v"(f.bh[0]), "v"(f.bl[0]), "v"(f.bh[1]), "v"(f.bl[1]));
        return;
      }
      // pass-major: lo x hi, hi x lo, hi x hi -- the per-accumulator order of srf_pwconv_x3w.hip (bit-identical sums); eight
      // independent accumulators between two MFMAs on the same one
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[t], f.bh[u], acc[t][u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[t], f.bl[u], acc[t][u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[t], f.bh[u], acc[t][u], 0, 0, 0);
    };
    Frags f;
    // GlobLN / PReLU / hi | lo split of raw(g1) -> B image(g1): this lane owns column 64 wn + lane of the tile and the k rows
    // 8 wm .. 8 wm + 7 of the step (k1 = the step's index inside ITS tile, for gamma / beta)
    const int cv_raw = (8 * wm) * (T_BN * 4) + (64 * wn + lane) * 4;
    const int cv_hi = t_swz(64 * wn + lane, wm), cv_lo = t_swz(64 * wn + lane, 2 + wm);
    auto convert = [&](int g1, int k1, float mean, float rstd) __attribute__((always_inline)) {
      if (ABL & 1) return;
      const char* raw = smem + T_OFF_R + (g1 % T_NR) * T_R_STAGE + cv_raw;
      char* img = smem + T_OFF_B + (g1 & 1) * T_B_STAGE;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(raw + j * (T_BN * 4));
      bf16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x0 = v[j];
        if constexpr (PRO == 1 || PRO == 2) {
          const int kk = k1 * T_BK + 8 * wm + j;
          const float sc = gamma[kk] * rstd;
          x0 = fmaf(x0, sc, beta[kk] - mean * sc);
        }
        if constexpr (PRO == 2 || PRO == 3) x0 = srf_prelu(x0, slope);
        const __bf16 h = (__bf16)x0;
        hi[j] = h;
        lo[j] = (__bf16)(x0 - (float)h);
      }
      *reinterpret_cast<bf16x8*>(img + cv_hi) = hi;
      *reinterpret_cast<bf16x8*>(img + cv_lo) = lo;
    };
    auto tile_stats = [&](const TileCur& c, float& mean, float& rstd) {
      mean = 0.f;
      rstd = 1.f;
      if constexpr (PRO == 1 || PRO == 2) {
        const float2 mr = stat_tab[c.b];
        mean = mr.x;
        rstd = mr.y;
      }
    };
    float* strip = reinterpret_cast<float*>(smem + T_OFF_STRIP + mw * T_STRIP);

    TileCur cur, nxc;
    cur_set(cur, 0);
    nxc = cur;
    if (ntile > 1) cur_next(nxc);
    float mean_c, rstd_c, mean_n, rstd_n;
    tile_stats(cur, mean_c, rstd_c);
    tile_stats(nxc, mean_n, rstd_n);
    int g = 0;
    __builtin_amdgcn_s_barrier();      // (the loaders' two prologue barriers: raw(0) has landed)
    if (wn < tile_ncb(cur)) convert(0, 0, mean_c, rstd_c);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // PHASES.  Every step is two phases, each closed by a barrier of all eight wavefronts.  The multipliers of rows 0-127 (group 0:
    // wavefronts 0, 1, 2) run LOAD(g) in phase 2 g and COMPUTE(g) in phase 2 g + 1; the multipliers of rows 128-255 (group 1:
    // wavefronts 4, 5, 6 -- the SIMD partners of 0, 1, 2) run one phase behind (one extra barrier here, one fewer at the end).  So
    // on every SIMD one wavefront owns the matrix pipe for 24 back-to-back MFMAs while its partner reads LDS and converts:
    //   LOAD(g):    fragments of step g (A ring slot g % 3, B image g & 1); this wavefront's share (8 k rows x its 64 columns) of
    //               raw(g + 1) -> B image(g + 1) -- complete one phase after group 1's LOAD(g), read first in group 0's LOAD(g + 1)
    //   COMPUTE(g): 24 MFMAs from registers.
    // (First version: every multiplier read, converted and multiplied within one step, all six in the same order at the same time
    // -- the pipe idled while both wavefronts of a SIMD waited for LDS or converted: a step took twice its MFMA time.)
    if (wm == 1) barrier_tl();
    for (int i = 0; i < ntile; ++i) {
      const int col0 = tile_col0(cur), ncb = tile_ncb(cur);
      const bool active = wn < ncb;                      // (wave-uniform) this multiplier's 64 columns hold any column < L
      const bool next_active = i + 1 < ntile && wn < tile_ncb(nxc);
      for (int k = 0; k < nk; ++k, ++g) {
        const unsigned w0 = TL ? tl_now() : 0u;
        if (active) read_frags(f, g);
        if (k + 1 < nk) {
          if (active) convert(g + 1, k + 1, mean_c, rstd_c);
        } else if (next_active) {
          convert(g + 1, 0, mean_n, rstd_n);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TL) tl_wait += tl_now() - w0;        // (multipliers: "wait" = the LOAD phases' own time)
        barrier_tl();
        if (active && k + 1 < nk) mma(f);
        if (k + 1 < nk) barrier_tl();
      }
      // ---- epilogue (the loaders are already two steps into the next tile)
      const unsigned tl_e0 = TL ? tl_now() : 0u;
      if (active) {
        const int m0 = cur.mt * T_BM + wm * 128, lcol = col0 + wn * 64;
        const long b = cur.b;
        // (the lane id through an opaque copy: otherwise every per-lane epilogue value is computed ahead of the tile loop as a
        // loop invariant and kept in registers through the k-loop, next to 128 accumulators and two fragment sets)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int c4 = (lane_o & 7) * 4, rsub = lane_o >> 3;
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
        // buffer descriptors of the example's output / second tensor: per-lane addresses are 32-bit offsets
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * a.Cout * L, 0, a.Cout * L * 4, 0x00020000);
        __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(ext ? ext + (size_t)b * extC * L : a.y), 0, ext ? extC * L * 4 : 0, 0x00020000);
        auto epi_row = [&](int t, int ii, int& mc) __attribute__((always_inline)) {
          const int m = m0 + t * 32 + ii * 8 + rsub;
          mc = m < a.Cout ? m : 0;
          return m < a.Cout;
        };
        auto epi_col = [&](int u, int& lc) __attribute__((always_inline)) {
          const int l = lcol + u * 32 + c4;
          lc = l < L ? l : 0;
          return l < L;
        };
        // the second tensor of one 32 x 32 block: 4 float4 per lane (+ the block's bias rows), requested two blocks ahead
        struct Ext {
          float4 e[4];
          float bias[4];
        };
        auto ext_issue = [&](Ext& x, int j) __attribute__((always_inline)) {      // block j = (t = j >> 1, u = j & 1)
          const int t = j >> 1, u = j & 1;
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) {
            int mc, lc;
            epi_row(t, ii, mc);
            x.bias[ii] = a.bias[mc];
            if constexpr (kHasExt) {
              epi_col(u, lc);
              const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ers, ((is_res ? mc : mc % extC) * L + lc) * 4, 0, 0);   // (no tensor: range 0 = zeros)
              x.e[ii] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
          }
        };
        Ext xa, xb;
        ext_issue(xa, 0);
        asm volatile("" ::: "memory");
        mma(f);                                // COMPUTE of the tile's last step ...
        barrier_tl();                          // ... and the barrier that closes that phase; the epilogue opens this wavefront's next LOAD phase
        ext_issue(xb, 1);
        asm volatile("" ::: "memory");
        float s = 0.f, q = 0.f;
        if constexpr ((ABL & 8) == 0) {
        const int col = lane_o & 31, kh = lane_o >> 5;
        auto epi_block = [&](const Ext& x, const f32x16& c, int j) __attribute__((always_inline)) {
          const int t = j >> 1, u = j & 1;
#pragma unroll
          for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * kh) * SRF_EPI_PITCH_H + col] = c[r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) {
            float4 o = *reinterpret_cast<const float4*>(strip + (ii * 8 + rsub) * SRF_EPI_PITCH_H + c4);
            const float bs = x.bias[ii];
            o.x += bs; o.y += bs; o.z += bs; o.w += bs;
            if constexpr (kHasExt) {
              const float4 e = x.e[ii];
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
            const bool okr = epi_row(t, ii, mc), okc = epi_col(u, lc);
            if (okr && okc) {
              const u32x4 ov = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
              __builtin_amdgcn_raw_buffer_store_b128(ov, yrs, (mc * L + lc) * 4, 0, (CP & 1) ? 2 : 0);
              s += (o.x + o.y) + (o.z + o.w);
              q = fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, fmaf(o.w, o.w, q))));
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        // (compiler fences: without them hipcc hoists the loads of all eight blocks to the top -- 128 registers next to the accumulators)
#define T_FENCE() asm volatile("" ::: "memory")
        epi_block(xa, acc[0][0], 0);
        T_FENCE();
        ext_issue(xa, 2);
        T_FENCE();
        epi_block(xb, acc[0][1], 1);
        T_FENCE();
        ext_issue(xb, 3);
        T_FENCE();
        epi_block(xa, acc[1][0], 2);
        T_FENCE();
        ext_issue(xa, 4);
        T_FENCE();
        epi_block(xb, acc[1][1], 3);
        T_FENCE();
        ext_issue(xb, 5);
        T_FENCE();
        epi_block(xa, acc[2][0], 4);
        T_FENCE();
        ext_issue(xa, 6);
        T_FENCE();
        epi_block(xb, acc[2][1], 5);
        T_FENCE();
        ext_issue(xb, 7);
        T_FENCE();
        epi_block(xa, acc[3][0], 6);
        T_FENCE();
        epi_block(xb, acc[3][1], 7);
#undef T_FENCE
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(acc[t][0]), "v"(acc[t][1]));
        }
        if (a.out_sums) {
          const double ds = srf_dpp_wave_sum((double)s), dq = srf_dpp_wave_sum((double)q);
          if (lane_o == 63) {
            double* dst = srf_stat_slot(a.out_sums, b, (long)cur.v * 24 + mw + 6 * (cur.q + 1));
            atomicAdd(dst, ds);
            atomicAdd(dst + 1, dq);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
      } else {
        barrier_tl();                            // (closes COMPUTE of the tile's last step)
      }
      if constexpr (TL) tl_epi += tl_now() - tl_e0;
      cur = nxc;
      mean_c = mean_n;
      rstd_c = rstd_n;
      if (i + 2 < ntile) {
        cur_next(nxc);
        tile_stats(nxc, mean_n, rstd_n);
      }
    }
    if (wm == 0) barrier_tl();      // (group 1's last COMPUTE phase)
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
}

bool srf_x3t_supported(int Bt, int Cin, int pro) {
  return (Cin % 32 == 0) && Cin >= 128 && (!(pro == 1 || pro == 2) || Bt <= T_MAX_STAT_EXAMPLES);
}

int srf_pw_x3t_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st) {
  const int nMt = (a.Cout + T_BM - 1) / T_BM, nLt = (a.L + T_BN - 1) / T_BN;
  const long total = (long)a.Bt * nMt * nLt;
  SRF_CHECK_ARG(total < (1L << 30), "srf_pw_conv: too many tiles");
  SRF_CHECK_ARG(srf_x3t_supported(a.Bt, a.Cin, pro) && a.L % 4 == 0 && a.L >= 4, "srf_pw_conv: shape not served by the 256 x 192 kernel");
  const size_t lds = (size_t)T_OFF_STAT + ((pro == 1 || pro == 2) ? (size_t)a.Bt * sizeof(float2) : 0);
  const long ok = srf_device_cached(5, [](void*) -> long {
    bool good = true;
    const void* fns[] = {
        (const void*)&srf_pw_x3t_kernel<0, 0, 0>, (const void*)&srf_pw_x3t_kernel<1, 0, 1>,
        (const void*)&srf_pw_x3t_kernel<2, 1, 1>, (const void*)&srf_pw_x3t_kernel<3, 2, 1>,
        (const void*)&srf_pw_x3t_kernel<0, 1, 0>,
        (const void*)&srf_pw_x3t_kernel<0, 3>, (const void*)&srf_pw_x3t_kernel<1, 3>,
        (const void*)&srf_pw_x3t_kernel<2, 3>, (const void*)&srf_pw_x3t_kernel<3, 3>,
        (const void*)&srf_pw_x3t_kernel<0, 0, 0, 1>, (const void*)&srf_pw_x3t_kernel<2, 1, 1, 1>,
        (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 1>, (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 2>,
        (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 4>, (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 8>,
        (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 16>, (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 9>,
        (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 13>, (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 15>,
        (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 31>, (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 24>,
        (const void*)&srf_pw_x3t_kernel<0, 0, 0, 0, 27>};
    for (const void* f : fns) good &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES) == hipSuccess;
    return good ? 1 : 0;
  }, nullptr);
  SRF_CHECK_ARG(ok == 1, "srf_pw_conv: cannot reserve %zu bytes of LDS", lds);
  long nb = srf_device_cus();
  nb -= nb % 8;
  if (nb < 8) nb = 8;
  if (nb > total) nb = total - total % 8;
  SRF_CHECK_ARG(nb >= 8, "srf_pw_conv: too few tiles for the 256 x 192 kernel");
  const int rounds = (int)(total / nb);
  int mgrp = nMt;
  {
    const long slab = (long)T_BM * a.Cin * 4, image = slab * nMt;
    if (image > (3L << 20)) {
      int g = (int)((2L << 20) / slab);
      while (g > 1 && nMt % g) --g;
      if (g >= 1 && g < nMt) mgrp = g;
    }
  }
  dim3 grid((unsigned)nb), block(T_THREADS);
  const bool res = a.residual != nullptr, mask = !res && (a.epi_mask & 1);
#define T_GO(...) hipLaunchKernelGGL((srf_pw_x3t_kernel<__VA_ARGS__>), grid, block, lds, st, a, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta, mgrp)
  const bool tl = getenv("SRF_X3S_TL") && atoi(getenv("SRF_X3S_TL"));
  const int abl = getenv("SRF_X3W_ABL") ? atoi(getenv("SRF_X3W_ABL")) : 0;
  if (abl && pro == 0 && !res && !mask) {
    switch (abl) {
      case 1: T_GO(0, 0, 0, 0, 1); break;
      case 2: T_GO(0, 0, 0, 0, 2); break;
      case 4: T_GO(0, 0, 0, 0, 4); break;
      case 8: T_GO(0, 0, 0, 0, 8); break;
      case 16: T_GO(0, 0, 0, 0, 16); break;
      case 9: T_GO(0, 0, 0, 0, 9); break;
      case 13: T_GO(0, 0, 0, 0, 13); break;
      case 15: T_GO(0, 0, 0, 0, 15); break;
      case 24: T_GO(0, 0, 0, 0, 24); break;
      case 27: T_GO(0, 0, 0, 0, 27); break;
      default: T_GO(0, 0, 0, 0, 31); break;
    }
  } else if (tl && pro == 0 && !res && !mask) T_GO(0, 0, 0, 1);
  else if (tl && pro == 2 && res) T_GO(2, 1, 1, 1);
  else if (pro == 0 && !res && !mask) T_GO(0, 0, 0);
  else if (pro == 1 && !res && !mask) T_GO(1, 0, 1);
  else if (pro == 2 && res) T_GO(2, 1, 1);
  else if (pro == 0 && res) T_GO(0, 1, 0);      // (the backward's data-gradient GEMM of proj_1x1: W^T g + skip gradient)
  else if (pro == 3 && mask) T_GO(3, 2, 1);
  else if (pro == 0) T_GO(0, 3);
  else if (pro == 1) T_GO(1, 3);
  else if (pro == 2) T_GO(2, 3);
  else T_GO(3, 3);
#undef T_GO
  static const char* const kLabel[4] = {"pw_conv_x3w<0>", "pw_conv_x3w<1>", "pw_conv_x3w<2>", "pw_conv_x3w<3>"};
  SRF_CHECK_LAUNCH(kLabel[pro < 0 || pro > 3 ? 3 : pro], st);
  return SRF_OK;
}
