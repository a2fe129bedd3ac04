// K2 (round-2 fast path) -- split-precision MFMA GEMM with a 256 x 128 block tile and pre-split weights.
//
// Same arithmetic as srf_pwconv_bf16x3.hip (x = hi + lo in bf16, three v_mfma_f32_32x32x16_bf16 per product block, fp32
// accumulate; reference sites improved_sudormrf.py:256-259, :174, :196,:220, :268-269,:295-298), restructured after the
// round-1 profile (VERDICT r1 "what's weak" 4: 0.33 of the HBM roofline, issue-bound: every X tile was transformed and
// split once per 128-row M tile, every W tile once per block, a barrier every 12 MFMAs per wavefront):
//   * block tile 256 (M) x 128 (time) x 32 (K): for Cout = 256 (res_conv, bottleneck) ONE block owns all of M, so an
//     activation tile goes through GlobLN / PReLU / split exactly once (twice for proj_1x1's 512 rows, was 4x);
//   * the weights are split into bf16 hi | lo ONCE per forward (srf_x3v_pack_launch) and stored tile by tile as the exact
//     LDS image of a stage, so the A operand is a global_load_lds_dwordx4 DMA: no VALU, no VGPRs, no ds_write;
//   * LDS rows are 64 B (32 bf16) with the 16-B chunks XOR-swizzled by (row >> 2) & 3: conflict-free ds_read_b128 fragment
//     reads and ds_write_b128 packet writes without padding (the packed weight image is stored pre-swizzled);
//   * 3 LDS stages of 48 KB (A_hi | A_lo | B_hi | B_lo) = 144 KB, one 512-thread block per CU, ONE barrier per k-tile:
//     step g reads stage g % 3 (24 MFMAs per wavefront on a 64 x 64 wave tile = 4 accumulator tiles), writes the split
//     activations of k-tile g+1 into stage (g+1) % 3 and starts the DMA + the activation loads of k-tile g+2;
//   * persistent blocks walk over tiles with the operand pipeline running across tile boundaries; the leftover tiles of the
//     last round are cut into quarter tiles (256 x 32) dealt to ALL blocks.
// Prologue / epilogue semantics are those of srf_pw.h (PwArgs).
#include <type_traits>

#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int V_BM = 256, V_BN = 128, V_BK = 32;
constexpr int V_A_IMG = V_BM * 64;                       // [256][32] bf16, 64-B rows
constexpr int V_B_IMG = V_BN * 64;                       // [128][32] bf16
constexpr int V_STAGE = 2 * V_A_IMG + 2 * V_B_IMG;       // A_hi | A_lo | B_hi | B_lo = 48 KB
constexpr int V_NSTAGE = 3;
constexpr int V_WTILE_BYTES = 2 * V_A_IMG;               // packed weights of one (m-tile, k-tile) = the A half of a stage

// byte offset of 16-B chunk `c` (8 consecutive k) of row `r` inside an image
__host__ __device__ __forceinline__ int v_swz(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

// ---------------------------------------------------------------------------------------------
// weight packing: W[Cout][Cin] fp32 -> per (m-tile, k-tile) [hi image | lo image], rows >= Cout zero
// ---------------------------------------------------------------------------------------------
#define SRF_V_MAX_PACK 96
struct VPackEntry {
  const float* w;
  char* dst;
  int Cout, Cin;   // Cin < 0: w is stored TRANSPOSED ([|Cin|][Cout]) -- the backward's data-gradient GEMMs use W^T
};
struct VPackTable {
  VPackEntry e[SRF_V_MAX_PACK];
};

__global__ __launch_bounds__(256) void srf_x3v_pack_kernel(VPackTable t) {
  VPackEntry e = t.e[blockIdx.y];
  const bool trans = e.Cin < 0;
  e.Cin = trans ? -e.Cin : e.Cin;
  const int nKt = e.Cin / V_BK;
  const int nMt = (e.Cout + V_BM - 1) / V_BM;
  const long total = (long)nMt * nKt * V_BM * 4;   // one thread per 8-k packet
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int c = (int)(id & 3);
    const int row = (int)((id >> 2) % V_BM);
    const long tile = (id >> 2) / V_BM;
    const int kt = (int)(tile % nKt), mt = (int)(tile / nKt);
    const int m = mt * V_BM + row;
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kt * V_BK + c * 8 + j;
      const float v = (m < e.Cout) ? (trans ? e.w[(size_t)k * e.Cout + m] : e.w[(size_t)m * e.Cin + k]) : 0.f;
      const __bf16 h = (__bf16)v;
      hi[j] = h;
      lo[j] = (__bf16)(v - (float)h);
    }
    char* base = e.dst + (size_t)tile * V_WTILE_BYTES + v_swz(row, c);
    *reinterpret_cast<bf16x8*>(base) = hi;
    *reinterpret_cast<bf16x8*>(base + V_A_IMG) = lo;
  }
}

size_t srf_x3v_packed_bytes(int Cout, int Cin) {
  const size_t nMt = (Cout + V_BM - 1) / V_BM, nKt = Cin / V_BK;
  return nMt * nKt * (size_t)V_WTILE_BYTES;
}

bool srf_x3v_supported(int Cin, int Cout, int L) {
  return (Cin % 64 == 0) && Cin >= 128 && (L % 4 == 0) && Cout >= 192 && (long)Cout * Cin * 4 < (1L << 31);
}

int srf_x3v_pack_launch(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n, hipStream_t st) {
  for (int base = 0; base < n; base += SRF_V_MAX_PACK) {
    VPackTable t;
    const int cnt = (n - base) < SRF_V_MAX_PACK ? (n - base) : SRF_V_MAX_PACK;
    for (int i = 0; i < SRF_V_MAX_PACK; ++i) {
      const int j = base + (i < cnt ? i : 0);
      t.e[i] = VPackEntry{w[j], dst[j], Cout[j], Cin[j]};
    }
    hipLaunchKernelGGL(srf_x3v_pack_kernel, dim3(64, cnt), dim3(256), 0, st, t);
    SRF_CHECK_LAUNCH("pack_pw_weights", st);
  }
  return SRF_OK;
}

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
template <int PRO, int ABL = 0, int KS = 0>
__global__ __launch_bounds__(512, 2) void srf_pw_x3v_kernel(PwArgs a, const char* __restrict__ wpack, int nMt, int nLt,
                                                            int total, int rounds, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // V_NSTAGE * V_STAGE

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // full tiles: 4 x 2 wavefronts, 64 x 64 each
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
  const int b_n = tid & 127, b_c = wave >> 1, b_kg = b_c * 8;
  const int b_lds = 2 * V_A_IMG + v_swz(b_n, b_c);
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
    t.a_src = wpack + (size_t)c.mt * nk * V_WTILE_BYTES + wave * 4096;
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
    float b[8];
  };
  // The DMA is issued through inline asm on purpose: with a global_load_lds BUILTIN in flight hipcc waits vmcnt(0) at the
  // next use of any ordinary load's result and cannot leave the epilogue's stores in flight behind the next tile's loads
  // (cdna_hip_programming.md "mixing load kinds"); hidden in asm, the compiler counts only its own loads and stores --
  // exactly, in order -- and its waits are merely a little early (they do not see the 4 DMA pieces per step), never late.
  // The DMA's own completion is waited for by hand (end of step, see `step`).  "memory" clobber: nothing moves across.
  auto gload_a = [&](const TileP& t, int kt, int stage) __attribute__((always_inline)) {
    if (ABL & 2) return;
    const char* src = t.a_src + (size_t)kt * V_WTILE_BYTES + lane * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)V_LDS(smem + stage * V_STAGE + wave * 4096));
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
  // ... B into registers
  auto gload_b = [&](Regs& r, const TileP& t, int kt) __attribute__((always_inline)) {
    if (ABL & 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(r.b[j]));
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      r.b[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, t.b_vo, (kt * V_BK + j) * L * 4, 0));
  };
  // GlobLN / PReLU / split of k-tile kt (tile t) -> B images of `stage`
  auto lds_store = [&](const Regs& r, const TileP& t, int kt, int stage) __attribute__((always_inline)) {
    if (ABL & 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(r.b[j]));
      return;
    }
    char* base = smem + stage * V_STAGE + b_lds;
    float vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x0 = r.b[j];
      if (PRO == 1 || PRO == 2) {
        const int k = kt * V_BK + b_kg + j;
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
  };

  // ---- MFMA.  Fragment rows of this lane: A operand row = <wave's first row> + mi*32 + (lane & 31), chunk = 2 ks + (lane >> 5).
  // Full tile: wavefront (wm, wn) owns rows wm*64.., columns wn*64.. (2 x 2 accumulator tiles); quarter tile: wavefront w owns
  // rows w*32.., columns 0..31 (one accumulator tile).
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
        a_off[t][ks] = v_swz(quarter ? wave * 32 + fr : wm * 64 + t * 32 + fr, 2 * ks + fc);
        b_off[t][ks] = 2 * V_A_IMG + v_swz(quarter ? fr : wn * 64 + t * 32 + fr, 2 * ks + fc);
      }
  };
  // The LDS -> register half and the MFMA half of a k-tile are separate: the fragment reads are issued FIRST in a step, ahead
  // of the step's ds_writes and DMA (which the compiler must assume to alias them: the stage indices are run-time values),
  // so that the MFMAs depend on registers only and can be interleaved with the split's VALU work.
  struct Frags {
    bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];   // [ks][mi | ni]
  };
  auto read_frags = [&](Frags& f, int stage, auto full_tag) __attribute__((always_inline)) {
    constexpr int NT = decltype(full_tag)::value ? 2 : 1;
    const char* base = smem + stage * V_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f.ah[ks][t] = *reinterpret_cast<const bf16x8*>(base + a_off[t][ks]);
        f.al[ks][t] = *reinterpret_cast<const bf16x8*>(base + V_A_IMG + a_off[t][ks]);
        f.bh[ks][t] = *reinterpret_cast<const bf16x8*>(base + b_off[t][ks]);
        f.bl[ks][t] = *reinterpret_cast<const bf16x8*>(base + V_B_IMG + b_off[t][ks]);
      }
  };
  auto read_frags_ks = [&](Frags& f, int stage, int ks, auto full_tag) __attribute__((always_inline)) {
    constexpr int NT = decltype(full_tag)::value ? 2 : 1;
    const char* base = smem + stage * V_STAGE;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f.ah[ks][t] = *reinterpret_cast<const bf16x8*>(base + a_off[t][ks]);
      f.al[ks][t] = *reinterpret_cast<const bf16x8*>(base + V_A_IMG + a_off[t][ks]);
      f.bh[ks][t] = *reinterpret_cast<const bf16x8*>(base + b_off[t][ks]);
      f.bl[ks][t] = *reinterpret_cast<const bf16x8*>(base + V_B_IMG + b_off[t][ks]);
    }
  };
  auto mma_ks = [&](const Frags& f, int ks, auto full_tag) __attribute__((always_inline)) {
    if (ABL & 4) return;
    constexpr int NT = decltype(full_tag)::value ? 2 : 1;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[ks][mi], f.bh[ks][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ks][mi], f.bl[ks][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < NT; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ks][mi], f.bh[ks][ni], acc[mi][ni], 0, 0, 0);
  };
  auto mma_tile = [&](const Frags& f, auto full_tag) __attribute__((always_inline)) {
    if (ABL & 4) {
      asm volatile("" ::"v"(f.ah[0][0]), "v"(f.al[1][0]), "v"(f.bh[0][0]), "v"(f.bl[1][0]));
      return;
    }
    constexpr int NT = decltype(full_tag)::value ? 2 : 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // pass-major order: independent accumulators between two MFMAs on the same one
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < NT; ++mi)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[ks][mi], f.bh[ks][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < NT; ++mi)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ks][mi], f.bl[ks][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < NT; ++mi)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ks][mi], f.bh[ks][ni], acc[mi][ni], 0, 0, 0);
    }
  };

  // One pipeline step = k-tile kt of the current tile `tc` (in stage s0): split k-tile kt+1 into stage s1, start the DMA of
  // k-tile kt+2 into stage s2 and the activation loads of k-tile kt+3 (two register sets: the B values of pipeline k-tile j
  // live in r[j & 1]), multiply k-tile kt.  k-tile indices >= nk belong to the NEXT tile `tn` (selected without branches:
  // a step is one scheduling region).
  // Barrier protocol (one per step): before it every wavefront has finished its ds_writes of k-tile kt+1 (lgkmcnt) and its
  // DMA pieces of k-tile kt+1 have landed: they were issued a step ago, and at least 8 + 4 + 8 memory operations were issued
  // after them (more at a tile boundary: the epilogue's loads and stores; loads return in order), so "at most 20
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
    if constexpr (KS == 1) {
      // second k-sub-step's fragments read behind the first one's MFMAs: 32 instead of 64 fragment registers.  Same-box A/B on
      // cfg 2: the kernels with a GlobLN prologue (res_conv 131 -> 127 us, bottleneck 116 -> 113: their 9 / 1 spilled
      // registers are gone) gain, proj_1x1 (no prologue, 8 k-steps) loses (121 -> 129) and stays on the hoisted form
      read_frags_ks(f, s0, 0, full_tag);
      lds_store(nx, t1, k1, s1);
      gload_a(t2, k2, s2);
      gload_b(nx, t3, k3);
      mma_ks(f, 0, full_tag);
      read_frags_ks(f, s0, 1, full_tag);
      mma_ks(f, 1, full_tag);
    } else {
      read_frags(f, s0, full_tag);
      lds_store(nx, t1, k1, s1);
      gload_a(t2, k2, s2);
      gload_b(nx, t3, k3);
      mma_tile(f, full_tag);
    }
    asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
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
  if (ABL & 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r0.b[j] = r1.b[j] = 0.5f * j + lane;
  }
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
    const int NT = quarter ? 1 : 2;
    const int mrow = quarter ? m0 + wave * 32 : m0 + wm * 64;     // first output row / column of this wavefront
    const int lcol = quarter ? l0 : l0 + wn * 64;
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;
    const float* ext = a.residual ? a.residual : ((a.epi_mask & 1) ? a.mul : nullptr);
    const int extC = a.residual ? a.Cout : a.mul_channels;
    const float* extb = ext ? ext + (size_t)b * extC * L : nullptr;
    float* yb = a.y + (size_t)b * a.Cout * L;
    float4 rext[2][2][4];
    float rbias[2][4];
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
      if (ABL & 16) return;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          int mc, lc;
          epi_row(mi, ii, mc);
          rbias[mi][ii] = a.bias[mc];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            epi_col(ni, lc);
            if (ext && mi < NT && ni < NT) {
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
    if (ABL & 16) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) asm volatile("" ::"v"(acc[mi][0]), "v"(acc[mi][1]));
    } else {
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
                if (ABL & 32)
                  asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));   // diagnostics: epilogue without its stores
                else
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

// (Measured and removed again, round 2: a DEFERRED epilogue -- the accumulators of tile i parked in 64 registers and leaving
// 4 (Cin >= 512) or 8 (Cin = 256) registers per k-step of tile i+1 as plain dword buffer stores, their residual / bias
// requested one step ahead, the last tile drained after its k-loop.  Motivation: the ablation of this kernel on res_conv
// (tools/gemm_ablate2.py: 141 us full, 95 us without the epilogue, 142 us without the epilogue's stores only, 75 us without
// any memory operation) says the chip-wide read + write burst at each tile boundary is exposed.  Three forms were built, all
// bit-identical to this kernel: (a) a run-time switch over the 16 pieces inside the step: hipcc's s_waitcnt insertion drains
// vmcnt(0) at the joins and every spill reload is a vmcnt(0) too: res_conv 172 us vs 132; (b) the next group's loads issued
// before this group's stores, two register sets: 177 us; (c) the first 16 steps unrolled so that the piece index is a
// compile-time constant (straight-line code, exact waits, 4 of 16 steps still with a scratch reload): 151 us, and 304 vs
// 119 us for proj_1x1, whose two pieces per step spill ~100 registers.  With 64 + 64 accumulators + 32 fragment registers
// the 256-register budget of two wavefronts per SIMD leaves no room for the k-loop's own pipeline, and the extra ~75 VALU
// + ~80 SALU instructions per step eat what the overlap wins.  Code: git history, round 2.)
// (Measured and removed again, round 2: the same kernel with FOUR wavefronts per block -- one per SIMD, wave tile 128 x 64,
// 428-490 registers, 96 instead of 128 KB of LDS fragment reads per k-step, second sub-step's fragments read ahead of the
// first MFMA.  Bit-identical at the first run and 20 % slower: res_conv 153 vs 127 us, proj_1x1 146 vs 120, mask 304 vs 262.
// hipcc interleaves its ~250 other instructions between the step's 48 MFMAs, but with nothing else resident on the SIMD
// every s_waitcnt and every barrier release is an MFMA bubble.  srf_pwconv_x4w.hip in the round's history.)
// (Measured and removed again, round 2: launches with fewer tiles than CUs -- small batches -- dealt out entirely as quarter
// tiles (rounds = 0: 100 instead of 25-50 busy CUs for a batch-1 GEMM).  Slower than the 128 x 128 kernels they replace:
// cfg 1 batch 1 0.87 vs 0.79 ms per forward, cfg 2 at batch 4 2.14 vs 1.75 ms -- a quarter tile streams the whole 256-row
// weight image for a quarter of the MFMA work.)
// (Measured and removed again: the same pipeline as TWO independent 256-thread blocks per CU -- 256 x 64 tiles,
// 4 wavefronts, 2 LDS stages of 40 KB -- so that one block's epilogue and operand waits sit under the other's MFMAs.  Correct
// at the first run, but slower inside the forward: res_conv 144 vs 130 us, proj_1x1 133 vs 119 us, mask 287 vs 251 us; it
// doubles the L2 -> LDS weight traffic and halves the reuse of every fragment of W.)

int srf_pw_x3v_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st) {
  const int nMt = (a.Cout + V_BM - 1) / V_BM, nLt = (a.L + V_BN - 1) / V_BN;
  const long total = (long)a.Bt * nMt * nLt;
  SRF_CHECK_ARG(total < (1L << 31), "srf_pw_conv: too many tiles");
  SRF_CHECK_ARG((long)a.Bt * a.Cin * a.L * 4 < (1L << 31), "srf_pw_conv: activation tensor too large for buffer loads");
  const size_t lds = (size_t)V_NSTAGE * V_STAGE;
  struct Q {
    size_t lds;
  } q{lds};
  // dynamic LDS beyond 64 KB needs the attribute once per device (per-device cache, srf_common.h)
  const long ok = srf_device_cached(2, [](void* p) -> long {
    const int bytes = (int)static_cast<Q*>(p)->lds;
    bool good = true;
    const void* fns[] = {(const void*)&srf_pw_x3v_kernel<0>, (const void*)&srf_pw_x3v_kernel<1>,
                         (const void*)&srf_pw_x3v_kernel<2>, (const void*)&srf_pw_x3v_kernel<3>,
                         (const void*)&srf_pw_x3v_kernel<0, 0, 1>, (const void*)&srf_pw_x3v_kernel<1, 0, 1>,
                         (const void*)&srf_pw_x3v_kernel<2, 0, 1>, (const void*)&srf_pw_x3v_kernel<3, 0, 1>,
                         (const void*)&srf_pw_x3v_kernel<2, 1>, (const void*)&srf_pw_x3v_kernel<2, 2>,
                         (const void*)&srf_pw_x3v_kernel<2, 3>, (const void*)&srf_pw_x3v_kernel<2, 4>,
                         (const void*)&srf_pw_x3v_kernel<2, 8>, (const void*)&srf_pw_x3v_kernel<2, 16>,
                         (const void*)&srf_pw_x3v_kernel<2, 19>, (const void*)&srf_pw_x3v_kernel<2, 12>,
                         (const void*)&srf_pw_x3v_kernel<2, 31>, (const void*)&srf_pw_x3v_kernel<2, 32>};
    for (const void* f : fns) good &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    return good ? 1 : 0;
  }, &q);
  SRF_CHECK_ARG(ok == 1, "srf_pw_conv: cannot reserve %zu bytes of LDS", lds);
  long nb = srf_device_cus();
  nb -= nb % 8;
  if (nb < 8) nb = 8;
  if (nb > total) nb = total - total % 8;   // (the tile cursors need nb % 8 == 0; the host dispatches this kernel for total >= #CUs)
  SRF_CHECK_ARG(nb >= 8, "srf_pw_conv: too few tiles for the 256 x 128 kernel");
  // full rounds of whole tiles; the leftover tiles go out as quarter tiles (debug flag 256: as whole tiles)
  const int rounds = (srf_debug_flags() & 256) ? (int)((total + nb - 1) / nb) : (int)(total / nb);
  dim3 grid((unsigned)nb), block(512);
  PwArgs ap = a;
  {
    const int su = (srf_debug_flags() >> 20) & 15;   // start-up stagger units (default none: a delayed block ends late)
    ap.epi_mask |= (su == 15 ? 0 : su) << 8;
    if (!(srf_debug_flags() & 512)) ap.epi_mask |= 1 << 12;   // quarter tiles first (flag 512: last)
  }
  const int abl = (srf_debug_flags() >> 16) & 31;     // diagnostics: ablated pipelines (GlobLN + PReLU prologue only)
  if (abl && pro == 2) {
#define V_ABL(A) hipLaunchKernelGGL((srf_pw_x3v_kernel<2, A>), grid, block, lds, st, ap, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta)
    switch (abl) {
      case 1: V_ABL(1); break;
      case 2: V_ABL(2); break;
      case 3: V_ABL(3); break;
      case 4: V_ABL(4); break;
      case 8: V_ABL(8); break;
      case 16: V_ABL(16); break;
      case 19: V_ABL(19); break;
      case 12: V_ABL(12); break;
      case 30: V_ABL(32); break;   // (flag value 30: no epilogue stores)
      default: V_ABL(31); break;
    }
#undef V_ABL
    SRF_CHECK_LAUNCH("pw_conv_x3v_ablated", st);
    return SRF_OK;
  }
  // fragment-read order (see `step`): per k-sub-step for the GlobLN-prologue kernels, hoisted for the others; debug flag 2 swaps
  const int ksv = ((pro == 1 || pro == 2) ? 1 : 0) ^ ((srf_debug_flags() & 2) ? 1 : 0);
#define V_LAUNCH(P) do { if (ksv == 1) hipLaunchKernelGGL((srf_pw_x3v_kernel<P, 0, 1>), grid, block, lds, st, ap, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta); \
    else hipLaunchKernelGGL((srf_pw_x3v_kernel<P>), grid, block, lds, st, ap, wpack, nMt, nLt, (int)total, rounds, a.nrm.gamma, a.nrm.beta); } while (0)
  switch (pro) {
    case 0: V_LAUNCH(0); break;
    case 1: V_LAUNCH(1); break;
    case 2: V_LAUNCH(2); break;
    default: V_LAUNCH(3); break;
  }
#undef V_LAUNCH
  static const char* const kLabel[4] = {"pw_conv_x3v<0>", "pw_conv_x3v<1>", "pw_conv_x3v<2>", "pw_conv_x3v<3>"};
  SRF_CHECK_LAUNCH(kLabel[pro < 0 || pro > 3 ? 3 : pro], st);
  return SRF_OK;
}
