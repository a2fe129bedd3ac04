// EXPERIMENT (round 2): the 256 x 128 split-bf16 GEMM of srf_pwconv_x3v.hip with FOUR wavefronts per block -- one per SIMD, up
// to 512 registers each -- instead of eight: wave tile 128 x 64 (4 x 2 accumulator tiles), 96 instead of 128 KB of LDS fragment
// reads per k-step, half as many barrier participants, and register room for a deferred epilogue.  Same packed weights,
// same LDS images, same arithmetic (bit-identical results).  Dispatched only under debug flag 2 (A/B); see the header of
// srf_pwconv_x3v.hip for everything that is not about the wavefront count.
#include <type_traits>

#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int V_BM = 256, V_BN = 128, V_BK = 32;
constexpr int V_A_IMG = V_BM * 64;
constexpr int V_B_IMG = V_BN * 64;
constexpr int V_STAGE = 2 * V_A_IMG + 2 * V_B_IMG;
constexpr int V_NSTAGE = 3;
constexpr int V_WTILE_BYTES = 2 * V_A_IMG;
__host__ __device__ __forceinline__ int v_swz(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

// ---- (kernel body generated from srf_pwconv_x3v.hip, then edited)
// ---------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void v_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

#define V_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define V_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

// PRO: 0 = identity, 1 = GlobLN, 2 = GlobLN + PReLU, 3 = PReLU only.
// gamma / beta come again as noalias kernel arguments so that they are fetched with scalar loads (they are wave-uniform;
// through the PwArgs struct the compiler cannot rule out that the stores to y clobber them).
// (A sched_group_barrier layout of the step -- one MFMA, then three VALU in its shadow -- measured no gain over hipcc's own
// order once the fragment reads were hoisted, and its build faulted in round 2's ablation run: removed.)
// ABL (diagnostics, results are wrong when != 0): 1 = no activation loads, 2 = no weight DMA, 4 = no MFMAs,
// 8 = no GlobLN / PReLU / split / ds_write, 16 = no epilogue, 32 = epilogue without its global stores
//
// Work distribution: `total` full tiles (256 x 128) are dealt round-robin to the persistent blocks, rounds = total / nblk each;
// the R = total % nblk leftover tiles are cut into 4 R QUARTER tiles (256 x 32) dealt to all blocks, so the last round costs a
// quarter of a tile per block instead of a whole tile on R blocks (res_conv at batch 32: 800 tiles on 256 CUs = 3.125 rounds,
// was 4).  A quarter tile uses all eight wavefronts as 8 (M) x 1 (N), one 32 x 32 accumulator each; its columns sit at rows
// 0..31 of the B images, lanes beyond column 31 fetch nothing (out-of-range buffer offsets return 0 without a memory access).
template <int PRO>
__global__ __launch_bounds__(256, 1) void srf_pw_x4w_kernel(PwArgs a, const char* __restrict__ wpack, int nMt, int nLt,
                                                            int total, int rounds, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // V_NSTAGE * V_STAGE

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // full tiles: 2 x 2 wavefronts, 128 x 64 each
  const int Cin = a.Cin, L = a.L;
  const int nk = Cin / V_BK;                 // even, >= 4 (host checks)
  const int nblk = gridDim.x;
  const int nquart = 4 * (total - rounds * nblk);                                   // quarter tiles of the leftover round
  const int nq_mine = ((int)blockIdx.x < nquart) ? (nquart - (int)blockIdx.x + nblk - 1) / nblk : 0;
  const int ntile = rounds + nq_mine;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;
  const int x_bytes = a.Bt * Cin * L * 4;
  // One-off start-up stagger (epi_mask bits 8..11 = units of ~4K cycles, 4 phases by block id; default none)
  {
    const int units = (a.epi_mask >> 8) & 15;
    const int phase = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < units * phase; ++i) __builtin_amdgcn_s_sleep(64);
  }

  struct TileCur {
    int i, v, mt, lt, b, q;   // q: -1 = full tile, 0..3 = quarter of its parent tile
  };
  const int vstep = nblk >> 3;
  const int st_b = vstep / (nMt * nLt), st_r = vstep - st_b * (nMt * nLt);
  const int st_l = st_r / nMt, st_m = st_r - st_l * nMt;
  // Order of a block's tiles: its quarter tiles FIRST when epi_mask bit 12 is set.  Blocks that own a quarter tile then run
  // a fraction of a tile ahead of / behind the others for the whole launch, so the chip's epilogues (read + write bursts)
  // no longer all coincide.
  const int qfirst = (a.epi_mask >> 12) & 1 ? nq_mine : 0;     // tiles [0, qfirst) are quarter tiles, then the full ones
  auto is_quarter = [&](int i) { return qfirst ? i < qfirst : i >= rounds; };
  auto cur_set = [&](TileCur& c, int i) {   // by division: a block's first tile and its quarter tiles
    int p = blockIdx.x + (i - qfirst) * nblk;
    c.q = -1;
    if (is_quarter(i)) {
      const int qi = blockIdx.x + (qfirst ? i : i - rounds) * nblk;
      p = rounds * nblk + (qi >> 2);
      c.q = qi & 3;
    }
    const int v = srf_xcd_remap(p, total);
    const int t = v / nMt;
    c.i = i;
    c.v = v;
    c.mt = v - t * nMt;
    c.b = t / nLt;
    c.lt = t - c.b * nLt;
  };
  auto cur_next = [&](TileCur& c) {
    const int i = c.i + 1;
    if (is_quarter(i) || c.q >= 0) {
      cur_set(c, i);
      return;
    }
    c.i = i;
    c.v += vstep;
    c.mt += st_m;
    int cy = c.mt >= nMt ? 1 : 0;
    c.mt -= cy ? nMt : 0;
    c.lt += st_l + cy;
    cy = c.lt >= nLt ? 1 : 0;
    c.lt -= cy ? nLt : 0;
    c.b += st_b + cy;
  };

  // ---- B staging geometry: thread -> time step n = tid & 127, k-group kg = tid >> 7 (wave-uniform), 8 k rows
  const int b_n = tid & 127, b_c = wave >> 1, b_kg = b_c * 8;   // this thread: k-groups b_c and b_c + 2 (8 k rows each)
  const int b_lds = 2 * V_A_IMG + v_swz(b_n, b_c), b_lds2 = 2 * V_A_IMG + v_swz(b_n, b_c + 2);
  __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);

  // Everything a pipeline step needs to know about a tile, computed ONCE per tile outside the k-loop so that a step is
  // straight-line code (no branches between its MFMAs, its VALU work and its loads: one scheduling region).
  struct TileP {
    const char* a_src;   // this wavefront's 4-KB slice of the tile's packed weights, k-tile 0 (wave-uniform)
    int b_vo;            // per-lane byte offset of (example, k row b_kg, column) inside X; out of range = fetch nothing
    float mean, rstd;    // GlobLN statistics of the tile's example (PRO 1 / 2)
  };
  auto make_tile = [&](const TileCur& c) {
    TileP t;
    t.a_src = wpack + (size_t)c.mt * nk * V_WTILE_BYTES + wave * 8192;
    const int col0 = c.lt * V_BN + (c.q < 0 ? 0 : c.q * 32);
    const int width = c.q < 0 ? V_BN : 32;
    const int off = ((c.b * Cin + b_kg) * L + min(col0 + b_n, L - 1)) * 4;   // columns >= L are never stored
    t.b_vo = b_n < width ? off : x_bytes;
    t.mean = 0.f;
    t.rstd = 1.f;
    if (PRO == 1 || PRO == 2) srf_finalize_stats_dpp(a.nrm.sums, c.b, a.inv_count, t.mean, t.rstd);
    return t;
  };

  struct Regs {
    float b[16];
  };
  // The DMA is issued through inline asm on purpose: with a global_load_lds BUILTIN in flight hipcc waits vmcnt(0) at the
  // next use of any ordinary load's result and cannot leave the epilogue's stores in flight behind the next tile's loads
  // (cdna_hip_programming.md "mixing load kinds"); hidden in asm, the compiler counts only its own loads and stores --
  // exactly, in order -- and its waits are merely a little early (they do not see the 4 DMA pieces per step), never late.
  // The DMA's own completion is waited for by hand (end of step, see `step`).  "memory" clobber: nothing moves across.
  auto gload_a = [&](const TileP& t, int kt, int stage) __attribute__((always_inline)) {
    if (0) return;
    const char* src = t.a_src + (size_t)kt * V_WTILE_BYTES + lane * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)V_LDS(smem + stage * V_STAGE + wave * 8192));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src + i * 1024), "s"(dst + i * 1024)
          : "memory");
    }
  };
  // ... B into registers
  auto gload_b = [&](Regs& r, const TileP& t, int kt) __attribute__((always_inline)) {
    if (0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(r.b[j]));
      return;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      r.b[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, t.b_vo, (kt * V_BK + (j & 7) + (j >> 3) * 16) * L * 4, 0));
  };
  // GlobLN / PReLU / split of k-tile kt (tile t) -> B images of `stage`
  auto lds_store = [&](const Regs& r, const TileP& t, int kt, int stage) __attribute__((always_inline)) {
    if (0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(r.b[j]));
      return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      char* base = smem + stage * V_STAGE + (h ? b_lds2 : b_lds);
      float vb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x0 = r.b[8 * h + j];
        if (PRO == 1 || PRO == 2) {
          const int k = kt * V_BK + b_kg + 16 * h + j;
          const float sc = gamma[k] * t.rstd;
          x0 = fmaf(x0, sc, beta[k] - t.mean * sc);
        }
        if (PRO == 2 || PRO == 3) x0 = srf_prelu(x0, slope);
        vb[j] = x0;
      }
      bf16x8 hi, lo;
      v_split8(vb, hi, lo);
      *reinterpret_cast<bf16x8*>(base) = hi;
      *reinterpret_cast<bf16x8*>(base + V_B_IMG) = lo;
    }
  };

  // ---- MFMA.  Fragment rows of this lane: A operand row = <wave's first row> + mi*32 + (lane & 31), chunk = 2 ks + (lane >> 5).
  // Full tile: wavefront (wm, wn) owns rows wm*128.., columns wn*64.. (4 x 2 accumulator tiles); quarter tile: wavefront w owns
  // rows w*64.., columns 0..31 (2 x 1 accumulator tiles).
  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int fr = lane & 31, fc = lane >> 5;
  int a_off[4][2], b_off[2][2];   // [mi | ni][ks]
  auto set_off = [&](bool quarter) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < 4; ++t) a_off[t][ks] = v_swz(quarter ? wave * 64 + (t & 1) * 32 + fr : wm * 128 + t * 32 + fr, 2 * ks + fc);
#pragma unroll
      for (int t = 0; t < 2; ++t) b_off[t][ks] = 2 * V_A_IMG + v_swz(quarter ? fr : wn * 64 + t * 32 + fr, 2 * ks + fc);
    }
  };
  // One wavefront per SIMD: nothing else can cover this wavefront's LDS latency, so the second k-sub-step's fragments are
  // read behind the first one's MFMAs (and 512 registers leave room for both sets).
  struct Frags {
    bf16x8 ah[2][4], al[2][4], bh[2][2], bl[2][2];   // [ks][mi | ni]
  };
  auto read_frags_ks = [&](Frags& f, int stage, int ks, auto full_tag) __attribute__((always_inline)) {
    constexpr int MT = decltype(full_tag)::value ? 4 : 2, NT = decltype(full_tag)::value ? 2 : 1;
    const char* base = smem + stage * V_STAGE;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      f.ah[ks][t] = *reinterpret_cast<const bf16x8*>(base + a_off[t][ks]);
      f.al[ks][t] = *reinterpret_cast<const bf16x8*>(base + V_A_IMG + a_off[t][ks]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f.bh[ks][t] = *reinterpret_cast<const bf16x8*>(base + b_off[t][ks]);
      f.bl[ks][t] = *reinterpret_cast<const bf16x8*>(base + V_B_IMG + b_off[t][ks]);
    }
  };
  auto mma_ks = [&](const Frags& f, int ks, auto full_tag) __attribute__((always_inline)) {
    constexpr int MT = decltype(full_tag)::value ? 4 : 2, NT = decltype(full_tag)::value ? 2 : 1;
    // pass-major order: independent accumulators between two MFMAs on the same one
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[ks][mi], f.bh[ks][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ks][mi], f.bl[ks][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ks][mi], f.bh[ks][ni], acc[mi][ni], 0, 0, 0);
  };

  // One pipeline step = k-tile kt of the current tile `tc` (in stage s0): split k-tile kt+1 into stage s1, start the DMA of
  // k-tile kt+2 into stage s2 and the activation loads of k-tile kt+3 (two register sets: the B values of pipeline k-tile j
  // live in r[j & 1]), multiply k-tile kt.  k-tile indices >= nk belong to the NEXT tile `tn` (selected without branches:
  // a step is one scheduling region).
  // Barrier protocol (one per step): before it every wavefront has finished its ds_writes of k-tile kt+1 (lgkmcnt) and its
  // DMA pieces of k-tile kt+1 have landed: they were issued a step ago, and at least 16 + 8 + 16 memory operations were issued
  // after them (more at a tile boundary: the epilogue's loads and stores; loads return in order), so "at most 40
  // outstanding" implies they are complete.  After it k-tile kt+1 is complete for every reader, and stage s0 may be
  // overwritten by the next step's DMA.
  int s0 = 0;
  TileP tc, tn;
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
  auto step = [&](Regs& nx, int kt, auto full_tag) __attribute__((always_inline)) {
    const int s1 = s0 == V_NSTAGE - 1 ? 0 : s0 + 1, s2 = s1 == V_NSTAGE - 1 ? 0 : s1 + 1;
    int k1, k2, k3;
    const TileP t1 = pick(kt + 1, k1), t2 = pick(kt + 2, k2), t3 = pick(kt + 3, k3);
    Frags f;
    read_frags_ks(f, s0, 0, full_tag);
    lds_store(nx, t1, k1, s1);
    gload_a(t2, k2, s2);
    gload_b(nx, t3, k3);
    read_frags_ks(f, s0, 1, full_tag);   // (issued before the first MFMA: they arrive while the first sub-step multiplies)
    mma_ks(f, 0, full_tag);
    mma_ks(f, 1, full_tag);
    // per step and wavefront: 8 DMA pieces + 16 activation loads; the DMA of k-tile kt+1 is a step old, 16 + 8 + 16 operations
    // were issued after it
    asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    s0 = s1;
  };

  TileCur cur, nxc;
  cur_set(cur, 0);
  tc = make_tile(cur);
  nxc = cur;
  if (ntile > 1) cur_next(nxc);
  tn = ntile > 1 ? make_tile(nxc) : tc;   // past the last tile the pipeline re-reads that tile (harmless)
  Regs r0, r1;
  gload_a(tc, 0, 0);
  gload_b(r0, tc, 0);                   // k-tile 0 -> stage 0 (A), r0 (B)
  gload_a(tc, 1, 1);
  gload_b(r1, tc, 1);                   // k-tile 1 -> stage 1,     r1     (nk >= 4)
  lds_store(r0, tc, 0, 0);              // B of k-tile 0 -> stage 0
  gload_b(r0, tc, 2);                   // k-tile 2 -> r0
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // A of k-tiles 0 and 1 landed
  __builtin_amdgcn_s_barrier();

  for (int i = 0; i < ntile; ++i) {
    const bool quarter = cur.q >= 0;
    set_off(quarter);
    const int m0 = cur.mt * V_BM, v = cur.v;
    const int l0 = cur.lt * V_BN + (quarter ? cur.q * 32 : 0);
    const long b = cur.b;
    // ---- what the epilogue reads from global memory (residual or mask multiplier, bias) is requested TWO STEPS before the
    // tile's k-loop ends -- one batch, up to 64 + 8 registers -- so that it arrives under the last MFMAs: the first version
    // loaded inside the per-piece loop and spent half of the kernel there (89 of 178 us on res_conv).
    const int MT = quarter ? 2 : 4, NT = quarter ? 1 : 2;
    const int mrow = quarter ? m0 + wave * 64 : m0 + wm * 128;     // first output row / column of this wavefront
    const int lcol = quarter ? l0 : l0 + wn * 64;
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;
    const float* ext = a.residual ? a.residual : ((a.epi_mask & 1) ? a.mul : nullptr);
    const int extC = a.residual ? a.Cout : a.mul_channels;
    const float* extb = ext ? ext + (size_t)b * extC * L : nullptr;
    float* yb = a.y + (size_t)b * a.Cout * L;
    float4 rext[4][2][4];
    float rbias[4][4];
    // (row / column bookkeeping is recomputed in the epilogue rather than kept in 32 registers across the last two steps)
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
    auto epi_issue = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          int mc, lc;
          epi_row(mi, ii, mc);
          rbias[mi][ii] = a.bias[mc];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            epi_col(ni, lc);
            if (ext && mi < MT && ni < NT) {
              const int me = a.residual ? mc : mc % extC;
              rext[mi][ni][ii] = *reinterpret_cast<const float4*>(extb + me * L + lc);
            } else {
              rext[mi][ni][ii] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        }
    };
    auto k_loop = [&](auto full_tag) __attribute__((always_inline)) {
      for (int kt = 0; kt + 2 < nk; kt += 2) {
        step(r1, kt, full_tag);        // converts k-tile kt+1 (odd: r1), loads k-tile kt+3 into r1
        step(r0, kt + 1, full_tag);    // converts k-tile kt+2 (even: r0), loads k-tile kt+4 into r0
      }
      epi_issue();
      step(r1, nk - 2, full_tag);
      step(r0, nk - 1, full_tag);
    };
    if (quarter)
      k_loop(std::false_type{});
    else
      k_loop(std::true_type{});
    // parameters of the tile after next (a statistics load + its wait): BEFORE the epilogue's stores are issued, so that no
    // compiler-inserted wait for a load ever stands behind them
    TileCur nnc = nxc;
    const bool has_nn = i + 2 < ntile;
    if (has_nn) cur_next(nnc);
    const TileP tnn = has_nn ? make_tile(nnc) : tn;
    // epilogue through wave-private strips in the stage the tile's last k-tile has just freed (every wavefront is past the
    // barrier that ended its reads); the other two stages hold the next tile's k-tiles 0 (complete) and 1 (DMA in flight).
    // No wait for memory here except the compiler's own (exact, counted) one for the batch requested above; the stores are
    // left in flight behind the next tile's first steps.
    const int free_stage = s0 == 0 ? V_NSTAGE - 1 : s0 - 1;
    float* strip = reinterpret_cast<float*>(smem + free_stage * V_STAGE) + wave * (32 * SRF_EPI_PITCH_H);
    float s = 0.f, q = 0.f;
    {
      const int col = lane & 31, kh = lane >> 5;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if (mi < MT && ni < NT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * kh) * SRF_EPI_PITCH_H + col] = acc[mi][ni][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
              float4 o = *reinterpret_cast<const float4*>(strip + (ii * 8 + rsub) * SRF_EPI_PITCH_H + c4);
              const float bs = rbias[mi][ii];
              const float4 e = rext[mi][ni][ii];
              o.x += bs; o.y += bs; o.z += bs; o.w += bs;
              if (a.residual) {
                o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
              } else if (a.epi_mask & 1) {
                o.x = fmaxf(o.x, 0.f) * e.x;
                o.y = fmaxf(o.y, 0.f) * e.y;
                o.z = fmaxf(o.z, 0.f) * e.z;
                o.w = fmaxf(o.w, 0.f) * e.w;
              }
              int mc, lc;
              const bool okr = epi_row(mi, ii, mc), okc = epi_col(ni, lc);
              if (okr && okc) {
                *reinterpret_cast<float4*>(yb + mc * L + lc) = o;
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
      // (DPP wave sums: VALU only; the __shfl_xor form is 24 ds_bpermute per tile on every wavefront's critical path)
      const double ds = srf_dpp_wave_sum((double)s), dq = srf_dpp_wave_sum((double)q);
      if (lane == 63) {
        double* dst = srf_stat_slot(a.out_sums, b, (long)v * 32 + wave + (quarter ? 8 * (cur.q + 1) : 0));
        atomicAdd(dst, ds);
        atomicAdd(dst + 1, dq);
      }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
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


int srf_pw_x4w_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st) {
  const int nMt = (a.Cout + V_BM - 1) / V_BM, nLt = (a.L + V_BN - 1) / V_BN;
  const long total = (long)a.Bt * nMt * nLt;
  SRF_CHECK_ARG(total < (1L << 31), "srf_pw_conv: too many tiles");
  SRF_CHECK_ARG((long)a.Bt * a.Cin * a.L * 4 < (1L << 31), "srf_pw_conv: activation tensor too large for buffer loads");
  const size_t lds = (size_t)V_NSTAGE * V_STAGE;
  struct Q {
    size_t lds;
  } q{lds};
  const long ok = srf_device_cached(6, [](void* p) -> long {
    const int bytes = (int)static_cast<Q*>(p)->lds;
    bool good = true;
    const void* fns[] = {(const void*)&srf_pw_x4w_kernel<0>, (const void*)&srf_pw_x4w_kernel<1>,
                         (const void*)&srf_pw_x4w_kernel<2>, (const void*)&srf_pw_x4w_kernel<3>};
    for (const void* f : fns) good &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    return good ? 1 : 0;
  }, &q);
  SRF_CHECK_ARG(ok == 1, "srf_pw_conv: cannot reserve %zu bytes of LDS", lds);
  long nb = srf_device_cus();
  nb -= nb % 8;
  if (nb < 8) nb = 8;
  if (nb > total) nb = total - total % 8;
  SRF_CHECK_ARG(nb >= 8, "srf_pw_conv: too few tiles for the 256 x 128 kernel");
  const int rounds = (int)(total / nb);
  dim3 grid((unsigned)nb), block(256);
  PwArgs ap = a;
  ap.epi_mask |= 1 << 12;   // quarter tiles first
#define V_LAUNCH(P) hipLaunchKernelGGL((srf_pw_x4w_kernel<P>), grid, block, lds, st, ap, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta)
  switch (pro) {
    case 0: V_LAUNCH(0); break;
    case 1: V_LAUNCH(1); break;
    case 2: V_LAUNCH(2); break;
    default: V_LAUNCH(3); break;
  }
#undef V_LAUNCH
  static const char* const kLabel[4] = {"pw_conv_x3v<0>", "pw_conv_x3v<1>", "pw_conv_x3v<2>", "pw_conv_x3v<3>"};   // (same labels: A/B)
  SRF_CHECK_LAUNCH(kLabel[pro < 0 || pro > 3 ? 3 : pro], st);
  return SRF_OK;
}
