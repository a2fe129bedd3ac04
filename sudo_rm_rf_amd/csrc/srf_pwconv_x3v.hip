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
//     last round run as half tiles (128 -> 64 columns).
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
  int Cout, Cin;
};
struct VPackTable {
  VPackEntry e[SRF_V_MAX_PACK];
};

__global__ __launch_bounds__(256) void srf_x3v_pack_kernel(VPackTable t) {
  const VPackEntry e = t.e[blockIdx.y];
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
      const float v = (m < e.Cout) ? e.w[(size_t)m * e.Cin + kt * V_BK + c * 8 + j] : 0.f;
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
// gamma / beta / bias come again as noalias kernel arguments so that they are fetched with scalar loads (they are
// wave-uniform; through the PwArgs struct the compiler cannot rule out that the stores to y clobber them).
template <int PRO>
__global__ __launch_bounds__(512, 2) void srf_pw_x3v_kernel(PwArgs a, const char* __restrict__ wpack, int nMt, int nLt,
                                                            int total, int nhalf, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // V_NSTAGE * V_STAGE

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 wavefronts, 64 x 64 each
  const int Cin = a.Cin, L = a.L;
  const int nk = Cin / V_BK;                 // even (host checks Cin % 64 == 0)
  const int nblk = gridDim.x;
  // leftover round as half tiles, see srf_pwconv_bf16x3.hip
  const int rounds = nhalf ? total / nblk : 0;
  const int ntile = nhalf ? rounds + ((int)blockIdx.x < nhalf ? 1 : 0) : (total - (int)blockIdx.x + nblk - 1) / nblk;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;

  struct TileCur {
    int i, v, mt, lt, b, half;
  };
  const int vstep = nblk >> 3;
  const int st_b = vstep / (nMt * nLt), st_r = vstep - st_b * (nMt * nLt);
  const int st_l = st_r / nMt, st_m = st_r - st_l * nMt;
  auto cur_set = [&](TileCur& c, int i) {
    int q = blockIdx.x + i * nblk;
    c.half = -1;
    if (nhalf && i == rounds) {
      q = rounds * nblk + ((int)blockIdx.x >> 1);
      c.half = blockIdx.x & 1;
    }
    const int v = srf_xcd_remap(q, total);
    const int t = v / nMt;
    c.i = i;
    c.v = v;
    c.mt = v - t * nMt;
    c.b = t / nLt;
    c.lt = t - c.b * nLt;
  };
  auto cur_next = [&](TileCur& c) {
    const int i = c.i + 1;
    if (nhalf && i == rounds) {
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
  __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.Bt * Cin * L * 4, 0x00020000);

  // Everything a pipeline step needs to know about a tile, computed ONCE per tile outside the k-loop so that a step is
  // straight-line code (no branches between its MFMAs, its VALU work and its loads: one scheduling region).
  struct TileP {
    const char* a_src;   // this wavefront's 4-KB slice of the tile's packed weights, k-tile 0 (wave-uniform)
    int b_vo;            // per-lane byte offset of (example, k row b_kg, column) inside X
    float mean, rstd;    // GlobLN statistics of the tile's example (PRO 1 / 2)
  };
  auto make_tile = [&](const TileCur& c) {
    TileP t;
    t.a_src = wpack + (size_t)c.mt * nk * V_WTILE_BYTES + wave * 4096;
    t.b_vo = ((c.b * Cin + b_kg) * L + min(c.lt * V_BN + b_n, L - 1)) * 4;   // columns >= L are never stored
    t.mean = 0.f;
    t.rstd = 1.f;
    if (PRO == 1 || PRO == 2) srf_finalize_stats_dpp(a.nrm.sums, c.b, a.inv_count, t.mean, t.rstd);
    return t;
  };

  struct Regs {
    float b[8];
  };
  // fetch k-tile kt of tile t: A by DMA into `stage` ...
  auto gload_a = [&](const TileP& t, int kt, int stage) {
    const char* src = t.a_src + (size_t)kt * V_WTILE_BYTES + lane * 16;
    char* dst = smem + stage * V_STAGE + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(V_GLB(src + i * 1024), V_LDS(dst + i * 1024), 16, 0, 0);
  };
  // ... B into registers
  auto gload_b = [&](Regs& r, const TileP& t, int kt) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      r.b[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, t.b_vo, (kt * V_BK + j) * L * 4, 0));
  };
  // GlobLN / PReLU / split of k-tile kt (tile t) -> B images of `stage`
  auto lds_store = [&](const Regs& r, const TileP& t, int kt, int stage) {
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

  // ---- MFMA: fragment rows of this lane.  A operand: row = wm*64 + mi*32 + (lane & 31), chunk = 2 ks + (lane >> 5)
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int fr = lane & 31, fc = lane >> 5;
  int a_off[2][2], b_off[2][2];   // [mi | ni][ks]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[t][ks] = v_swz(wm * 64 + t * 32 + fr, 2 * ks + fc);
  int cur_half = -1;
  auto set_b_off = [&]() {
    // full tile: columns wn*64 + ni*32; half tile h: columns h*64 + wn*32 (ni = 0 only)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int col = cur_half < 0 ? wn * 64 + t * 32 + fr : cur_half * 64 + wn * 32 + fr;
        b_off[t][ks] = 2 * V_A_IMG + v_swz(col, 2 * ks + fc);
      }
  };
  auto mma_tile = [&](int stage, auto full_tag) {
    constexpr bool full = decltype(full_tag)::value;
    const char* base = smem + stage * V_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(base + a_off[t][ks]);
        al[t] = *reinterpret_cast<const bf16x8*>(base + V_A_IMG + a_off[t][ks]);
      }
      bh[0] = *reinterpret_cast<const bf16x8*>(base + b_off[0][ks]);
      bl[0] = *reinterpret_cast<const bf16x8*>(base + V_B_IMG + b_off[0][ks]);
      if (full) {
        bh[1] = *reinterpret_cast<const bf16x8*>(base + b_off[1][ks]);
        bl[1] = *reinterpret_cast<const bf16x8*>(base + V_B_IMG + b_off[1][ks]);
      }
      // pass-major order: independent accumulators between two MFMAs on the same one
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[0], acc[mi][0], 0, 0, 0);
      if (full) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[1], acc[mi][1], 0, 0, 0);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[0], acc[mi][0], 0, 0, 0);
      if (full) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[1], acc[mi][1], 0, 0, 0);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[0], acc[mi][0], 0, 0, 0);
      if (full) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[1], acc[mi][1], 0, 0, 0);
      }
    }
  };

  // One pipeline step = k-tile kt of the current tile `tc` (in stage s0): split k-tile kt+1 into stage s1, start the DMA of
  // k-tile kt+2 into stage s2 and the activation loads of k-tile kt+3 (two register sets: the B values of pipeline k-tile j
  // live in r[j & 1]), multiply k-tile kt.  k-tile indices >= nk belong to the NEXT tile `tn` (selected without branches:
  // a step is one scheduling region).
  // Barrier protocol (one per step): before it every wavefront has finished its ds_writes of k-tile kt+1 (lgkmcnt) and its
  // DMA pieces of k-tile kt+1 have landed: they were issued a step ago and only 8 + 4 + 8 loads were issued after them
  // (loads return in order) -- vmcnt(20).  After it k-tile kt+1 is complete for every reader, and stage s0 may be
  // overwritten by the next step's DMA.
  int s0 = 0;
  TileP tc, tn;
  auto pick = [&](int k, int& kk) {
    const bool nx = k >= nk;   // wave-uniform
    kk = nx ? k - nk : k;
    TileP t;
    t.a_src = nx ? tn.a_src : tc.a_src;
    t.b_vo = nx ? tn.b_vo : tc.b_vo;
    t.mean = nx ? tn.mean : tc.mean;
    t.rstd = nx ? tn.rstd : tc.rstd;
    return t;
  };
  auto step = [&](Regs& nx, int kt, auto full_tag) {
    const int s1 = s0 == V_NSTAGE - 1 ? 0 : s0 + 1, s2 = s1 == V_NSTAGE - 1 ? 0 : s1 + 1;
    int k1, k2, k3;
    const TileP t1 = pick(kt + 1, k1), t2 = pick(kt + 2, k2), t3 = pick(kt + 3, k3);
    lds_store(nx, t1, k1, s1);
    gload_a(t2, k2, s2);
    gload_b(nx, t3, k3);
    mma_tile(s0, full_tag);
    asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    s0 = s1;
  };

  TileCur cur;
  cur_set(cur, 0);
  tc = make_tile(cur);
  tn = tc;
  Regs r0, r1;
  gload_a(tc, 0, 0);
  gload_b(r0, tc, 0);                   // k-tile 0 -> stage 0 (A), r0 (B)
  gload_a(tc, 1, 1);
  gload_b(r1, tc, 1);                   // k-tile 1 -> stage 1,     r1     (nk >= 4)
  lds_store(r0, tc, 0, 0);              // B of k-tile 0 -> stage 0
  gload_b(r0, tc, 2);                   // k-tile 2 -> r0
  asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");   // A of k-tile 0 landed
  __builtin_amdgcn_s_barrier();

  for (int i = 0; i < ntile; ++i) {
    cur_half = cur.half;
    set_b_off();
    TileCur nxc = cur;
    const bool has_next = i + 1 < ntile;
    if (has_next) cur_next(nxc);
    tn = has_next ? make_tile(nxc) : tc;   // past the last tile the pipeline re-reads that tile (harmless)
    if (cur_half < 0) {
      for (int kt = 0; kt < nk; kt += 2) {
        step(r1, kt, std::true_type{});       // converts k-tile kt+1 (odd: r1), loads kt+3 into r1
        step(r0, kt + 1, std::true_type{});   // converts k-tile kt+2 (even: r0), loads kt+4 into r0
      }
    } else {
      for (int kt = 0; kt < nk; kt += 2) {
        step(r1, kt, std::false_type{});
        step(r0, kt + 1, std::false_type{});
      }
    }
    // epilogue through wave-private strips in the stage the tile's last k-tile has just freed (every wavefront is past the
    // barrier that ended its reads); the other two stages hold the next tile's k-tiles 0 (complete) and 1 (DMA in flight)
    const int free_stage = s0 == 0 ? V_NSTAGE - 1 : s0 - 1;
    float* strip = reinterpret_cast<float*>(smem + free_stage * V_STAGE) + wave * (32 * SRF_EPI_PITCH_H);
    const int m0 = cur.mt * V_BM, l0 = cur.lt * V_BN, v = cur.v;
    const long b = cur.b;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      if (cur_half < 0) {
        srf_pw_epilogue_half(a, acc[mi][0], strip, b, m0 + wm * 64 + mi * 32, l0 + wn * 64, lane, s, q);
        srf_pw_epilogue_half(a, acc[mi][1], strip, b, m0 + wm * 64 + mi * 32, l0 + wn * 64 + 32, lane, s, q);
      } else {
        srf_pw_epilogue_half(a, acc[mi][0], strip, b, m0 + wm * 64 + mi * 32, l0 + cur_half * 64 + wn * 32, lane, s, q);
      }
    }
    if (a.out_sums) {
      const double ds = srf_wave_sum((double)s), dq = srf_wave_sum((double)q);
      if (lane == 0) {
        double* dst = srf_stat_slot(a.out_sums, b, (long)v * 16 + wave + (cur_half > 0 ? 8 : 0));
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
    tc = tn;
    // strip reads done before the next step's DMA overwrites that stage
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // surplus DMA of the pipeline tail must not outlive the block's LDS
}

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
    good &= hipFuncSetAttribute(reinterpret_cast<const void*>(&srf_pw_x3v_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    good &= hipFuncSetAttribute(reinterpret_cast<const void*>(&srf_pw_x3v_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    good &= hipFuncSetAttribute(reinterpret_cast<const void*>(&srf_pw_x3v_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    good &= hipFuncSetAttribute(reinterpret_cast<const void*>(&srf_pw_x3v_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    return good ? 1 : 0;
  }, &q);
  SRF_CHECK_ARG(ok == 1, "srf_pw_conv: cannot reserve %zu bytes of LDS", lds);
  long nb = srf_device_cus();
  nb -= nb % 8;
  if (nb < 8) nb = 8;
  if (nb > total) nb = total;   // (then nb % 8 may be != 0: the XCD remap stays bijective, tile cursors use division)
  const long rem = total % nb;
  const int nhalf = (nb % 8 == 0 && total >= nb && rem > 0 && 2 * rem <= nb && !(srf_debug_flags() & 256)) ? (int)(2 * rem) : 0;
  dim3 grid((unsigned)nb), block(512);
  switch (pro) {
    case 0: hipLaunchKernelGGL(srf_pw_x3v_kernel<0>, grid, block, lds, st, a, wpack, nMt, nLt, (int)total, nhalf, a.nrm.gamma, a.nrm.beta); break;
    case 1: hipLaunchKernelGGL(srf_pw_x3v_kernel<1>, grid, block, lds, st, a, wpack, nMt, nLt, (int)total, nhalf, a.nrm.gamma, a.nrm.beta); break;
    case 2: hipLaunchKernelGGL(srf_pw_x3v_kernel<2>, grid, block, lds, st, a, wpack, nMt, nLt, (int)total, nhalf, a.nrm.gamma, a.nrm.beta); break;
    default: hipLaunchKernelGGL(srf_pw_x3v_kernel<3>, grid, block, lds, st, a, wpack, nMt, nLt, (int)total, nhalf, a.nrm.gamma, a.nrm.beta); break;
  }
  static const char* const kLabel[4] = {"pw_conv_x3v<0>", "pw_conv_x3v<1>", "pw_conv_x3v<2>", "pw_conv_x3v<3>"};
  SRF_CHECK_LAUNCH(kLabel[pro < 0 || pro > 3 ? 3 : pro], st);
  return SRF_OK;
}
