// K2 (fast path) -- pointwise conv as a split-precision MFMA GEMM: every fp32 operand is split into
// two bf16 terms  x = hi + lo  (hi = bf16_rne(x), lo = bf16_rne(x - hi), |x - hi - lo| <= 2^-18 |x|)
// and each 32x32x16 product block is accumulated in fp32 from THREE bf16 MFMAs
//     acc += a_lo*b_hi;  acc += a_hi*b_lo;  acc += a_hi*b_hi          (a_lo*b_lo ~ 2^-18 dropped)
// The bf16 matrix pipe runs 16x the exact-fp32 MFMA rate, so three passes are still ~5x faster than
// v_mfma_f32_32x32x2_f32 while the result stays in the fp32 round-off class: the whole forward stays
// <= 1e-6 max-abs from the reference (bar: 1e-4); plain bf16 operands measured 3-4e-4 and fail it.
// Same sites / prologue / epilogue as srf_pwconv.hip (reference: improved_sudormrf.py:256-259, :174,
// :196,:220, :268-269,:295-298).
//
// Tiling: block 128(M) x 128(N=time) x 32(K), 8 wavefronts (4 x 2), each owning 32(M) x 64(time) = two 32x32
// MFMA tiles (32 fp32 accumulators per lane).  v_mfma_f32_32x32x16_bf16 wants 8 consecutive-k bf16 per lane
// for BOTH operands (A: row l&31, B: column l&31, k = 8*(l>>5)+j), but X_b is [k][time] with time contiguous:
// the transposition happens in registers while staging -- each thread loads 8 k-rows x 1 time step (coalesced
// 256-B row segments per wavefront), applies the GlobLN/PReLU prologue, splits, and writes the 8-k packets as
// 16-B ds_write_b128 into [time][k] LDS images (row pitch 80 B: conflict-free ds_read_b128 fragment fetches).
// W is [m][k] row-major already.  LDS: 4 images x 2 stages = 80 KB -> 2 blocks = 16 wavefronts per CU; global
// loads run two k-tiles ahead of their use.  Two kernels: one tile per block (srf_pw_bf16x3_w8_kernel) and the persistent
// one (srf_pw_bf16x3_p8_kernel, default for >= 3 tiles per block slot).
// WHAT THIS FILE SERVES since the 256 x 128 kernels (srf_pwconv_x3w.hip / _x3p.hip / _x3f.hip) took the large launches: 1x1
// convs with Cin % 64 == 0 whose launch has FEWER 256 x 128 tiles than the chip has CUs (batches of 2-8 at cfg 2: the 128 x 128
// tiles here make twice as many blocks), Cout < 192, callers that pass no packed weight image, the decoder's frame GEMM
// (800 tiles) and the backward's small data-gradient GEMMs.  Batch 1 goes to the 64 x 64 tiles of srf_pwconv_w4.hip.
#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 32;
constexpr int X3_PITCH = 80;                    // bytes per LDS row (32 bf16 = 64 B + 16 B pad)
constexpr int X3_IMG = X3_BM * X3_PITCH;        // one [128][32] bf16 image
constexpr int X3_STAGE = 4 * X3_IMG;            // A_hi, A_lo, B_hi, B_lo

__device__ __forceinline__ void srf_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

// PRO: 0 = identity, 1 = GlobLN, 2 = GlobLN + PReLU, 3 = PReLU only
//
// Variants that were built, measured on proj_1x1 (cfg 2) and removed again (git history has them):
//   * 4 wavefronts x (2x2 MFMA tiles): 164 us vs 152 us -- too few co-resident waves to overlap the split (VALU)
//     and MFMA phases (PMC: matrix pipe 21 % busy);
//   * wave-specialised (4 loader + 4 MMA wavefronts per block): 165 us -- the loader side is the bottleneck;
//   * activation prefetch 4 k-tiles ahead (separate register rings): no gain, spills with a prologue;
//   * 256x128 tile at 1 block per CU with pre-packed weights and sched_group_barrier interleave: slower;
//   * start-up stagger of every block of the one-tile-per-block kernel: slower (it only pays once per persistent
//     block, see below);
//   * the activation tile as 2 coalesced dwordx4 loads per thread instead of 8 dword loads (what a [k][time] LDS
//     image read back with ds_read_b64_tr_b16 would issue; tools/probes/tr_probe.hip documents that instruction):
//     timing-only experiment inside the forward, 132 vs 137 us -- the VMEM instruction count is not the limiter.
//   * static s_setprio 1 for the block's younger four wavefronts (persistent kernel): 164-166 vs 161 us on res_conv,
//     125-127 vs 124 us on proj_1x1 (same-box A/B, tools/lib_ab.sh) -- slower.

// ---------------------------------------------------------------------------------------------
// One tile per block.
// ---------------------------------------------------------------------------------------------
// (An instruction-interleave variant -- unconditional store + sched_group_barrier(MFMA 1 / VALU 7) -- was
// measured: 131 us vs 151 us on proj_1x1 but slower on the prologue variants and miscompiled for PRO 0;
// dropped.)
// BUF: operand loads as buffer loads (see the persistent kernel); false = 64-bit pointer form for tensors whose
// byte offsets do not fit 32 bits.
template <int PRO, bool BUF = true>
__global__ __launch_bounds__(512, 4) void srf_pw_bf16x3_w8_kernel(PwArgs a, int nMt, int nLt, int total) {
  __shared__ __attribute__((aligned(16))) char smem[2 * X3_STAGE];   // exactly 80 KB

  const int v = srf_xcd_remap(blockIdx.x, total);
  const int mt = v % nMt;
  const int lt = (v / nMt) % nLt;
  const long b = v / (nMt * nLt);
  const int m0 = mt * X3_BM, l0 = lt * X3_BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 waves, 32 x 64 each


  float mean = 0.f, rstd = 1.f, slope = 1.f;
  if (PRO == 1 || PRO == 2) srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
  if (PRO == 2 || PRO == 3) slope = a.nrm.prelu[0];

  const int Cin = a.Cin, L = a.L, Cout = a.Cout;
  const int nk_ = Cin / X3_BK;   // even (host checks Cin % 64 == 0)
  const float* xb = a.x + (size_t)b * Cin * L;

  // A (weights [m][k]): thread -> row m = tid>>2, 8-k packet pk = tid&3 (2 float4 = 32 B; a wavefront
  // covers 16 rows x 128 B = whole cache lines)
  const int a_m = tid >> 2, a_pk = tid & 3;
  const bool a_ok = (m0 + a_m) < Cout;
  const float* a_src = a.w + (size_t)(a_ok ? (m0 + a_m) : 0) * Cin + a_pk * 8;
  const float a_msk = (BUF || a_ok) ? 1.f : 0.f;   // BUF: rows beyond Cout read 0 from the descriptor
  const int a_lds = a_m * X3_PITCH + a_pk * 16;
  // B (X_b [k][time]): thread -> time step n = tid&127, k-group kg = tid>>7 (wave-uniform), 8 k rows
  const int b_n = tid & 127, b_kg = (wave >> 1) * 8;
  const bool b_ok = (l0 + b_n) < L;
  const float* b_src = xb + (size_t)b_kg * L + (b_ok ? (l0 + b_n) : 0);   // clamped; never stored if !ok
  const int b_lds = b_n * X3_PITCH + b_kg * 2;
  __amdgpu_buffer_rsrc_t a_rs, b_rs;
  int a_vo = 0, b_vo = 0;
  if (BUF) {
    a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, Cout * Cin * 4, 0x00020000);
    b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.Bt * Cin * L * 4, 0x00020000);
    a_vo = ((m0 + a_m) * Cin + a_pk * 8) * 4;
    b_vo = (((int)b * Cin + b_kg) * L + min(l0 + b_n, L - 1)) * 4;
  }

  struct Regs {
    float4 a[2];
    float b[8];
  };
  Regs r0, r1;
  auto gload = [&](Regs& r, int k0) {
    {
      if (BUF) {
        const auto w0 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo, k0 * 4, 0);
        const auto w1 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo, k0 * 4 + 16, 0);
        r.a[0] = make_float4(__uint_as_float(w0[0]), __uint_as_float(w0[1]), __uint_as_float(w0[2]), __uint_as_float(w0[3]));
        r.a[1] = make_float4(__uint_as_float(w1[0]), __uint_as_float(w1[1]), __uint_as_float(w1[2]), __uint_as_float(w1[3]));
      } else {
        r.a[0] = *reinterpret_cast<const float4*>(a_src + k0);
        r.a[1] = *reinterpret_cast<const float4*>(a_src + k0 + 4);
      }
    }
    {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        r.b[j] = BUF ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, b_vo, (k0 + j) * L * 4, 0))
                     : b_src[(size_t)(k0 + j) * L];
    }
  };
  auto lds_store = [&](const Regs& r, int stage, int k0) {
    char* base = smem + stage * X3_STAGE;
    const float va[8] = {r.a[0].x * a_msk, r.a[0].y * a_msk, r.a[0].z * a_msk, r.a[0].w * a_msk,
                         r.a[1].x * a_msk, r.a[1].y * a_msk, r.a[1].z * a_msk, r.a[1].w * a_msk};
    bf16x8 hi, lo;
    srf_split8(va, hi, lo);
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
    srf_split8(vb, hib, lob);
    *reinterpret_cast<bf16x8*>(base + 0 * X3_IMG + a_lds) = hi;
    *reinterpret_cast<bf16x8*>(base + 1 * X3_IMG + a_lds) = lo;
    *reinterpret_cast<bf16x8*>(base + 2 * X3_IMG + b_lds) = hib;
    *reinterpret_cast<bf16x8*>(base + 3 * X3_IMG + b_lds) = lob;
  };

  f32x16 acc0 = {0}, acc1 = {0};
  const int frag = (lane & 31) * X3_PITCH + (lane >> 5) * 16;
  const int a_row = (wm * 32) * X3_PITCH + frag;
  const int b_row0 = (wn * 64) * X3_PITCH + frag, b_row1 = b_row0 + 32 * X3_PITCH;
  auto mma_tile = [&](int stage) {
    const char* base = smem + stage * X3_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks * 32;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + 0 * X3_IMG + a_row + ko);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + 1 * X3_IMG + a_row + ko);
      const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(base + 2 * X3_IMG + b_row0 + ko);
      const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(base + 2 * X3_IMG + b_row1 + ko);
      const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(base + 3 * X3_IMG + b_row0 + ko);
      const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(base + 3 * X3_IMG + b_row1 + ko);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc1, 0, 0, 0);
    }
  };
  auto step = [&](Regs& nx, int kt) {
    if (kt + 1 < nk_) lds_store(nx, (kt + 1) & 1, (kt + 1) * X3_BK);
    gload(nx, min(kt + 3, nk_ - 1) * X3_BK);
    mma_tile(kt & 1);
    __syncthreads();
  };

  gload(r0, 0);
  gload(r1, X3_BK);
  lds_store(r0, 0, 0);
  gload(r0, min(2, nk_ - 1) * X3_BK);
  __syncthreads();
  for (int kt = 0; kt < nk_; kt += 2) {
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

// ---------------------------------------------------------------------------------------------
// Persistent 8-wave kernel.  In-kernel timeline of the kernel above (tools/gemm_timeline.py, proj_1x1):
// a block spends ~14 % of its life in the prologue (first operands from HBM), ~60 % in the k-loop and
// ~26 % in the epilogue, where every resident block of the chip writes its 64 KB at the same moment
// (the launch without its stores runs in 120 us instead of 161 us).  Here 2 blocks per CU stay
// resident and walk over tiles: the operand pipeline (global loads 2-3 k-tiles ahead, split + LDS
// store 1 ahead) runs straight across tile boundaries, so the next tile's first operands arrive under
// the current tile's MFMAs, and the epilogue's global stores are issued and left in flight while the
// next tile's k-loop runs.  The epilogue strip lives in the LDS stage the last k-tile has just freed
// (two 32x32 halves per wavefront, 36.9 KB), the other stage already holds the next tile's k-tile 0.
// (Measured and dropped: issuing the current tile's MFMAs first and splitting the next tile in their shadow
// -- MFMA(ks0), split A, MFMA(ks1), split B, stores, loads, barrier, pinned with sched_barrier -- changes
// nothing inside the forward: 129 vs 131 us on proj_1x1, 178 vs 170 us on res_conv.)
// ---------------------------------------------------------------------------------------------
// BUF: operand loads as buffer loads -- SGPR descriptor + per-lane byte offset fixed per tile + a SCALAR offset that
// walks the k rows, so the k-loop does no per-load 64-bit address arithmetic on the VALU (11 v_lshl_add_u64 per
// k-tile in the pointer form) and rows / columns beyond the tensor simply read 0 (no clamp, no row mask).
template <int PRO, bool BUF = false>
__global__ __launch_bounds__(512, 4) void srf_pw_bf16x3_p8_kernel(PwArgs a, int nMt, int nLt, int total, int nhalf) {
  __shared__ __attribute__((aligned(16))) char smem[2 * X3_STAGE];   // exactly 80 KB

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 waves, 32 x 64 each
  const int Cin = a.Cin, L = a.L, Cout = a.Cout;
  const int nk_ = Cin / X3_BK;   // even (host checks Cin % 64 == 0)
  const int nblk = gridDim.x;
  // Tail: total = rounds * nblk + R tiles.  With 0 < R <= nblk/2 the leftover round would run R blocks against
  // nblk slots (res_conv: 1600 tiles on 512 slots = 3 rounds + 64 tiles = the cost of 4); instead the R leftover
  // tiles are dealt as nhalf = 2R half tiles (128 x 64 columns) to the first 2R blocks: twice the blocks, half
  // the MFMA work each.  A half tile still stages the whole 128-column operand tile (simplest; its loads and
  // split are wasted for the other half) but every wavefront computes one 32x32 accumulator instead of two.
  const int rounds = nhalf ? total / nblk : 0;
  const int ntile = nhalf ? rounds + ((int)blockIdx.x < nhalf ? 1 : 0)
                          : (total - (int)blockIdx.x + nblk - 1) / nblk;   // >= 1 (host: grid <= total)
  const int nsteps = ntile * nk_;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;
  // one-off start-up stagger: blocks start in four phases ~4K cycles apart, so that the chip's epilogue write bursts stop
  // coinciding (measured in round 2 against none / two phases / longer units)
  {
    const int phase = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < phase; ++i) __builtin_amdgcn_s_sleep(64);
  }

  // tile i of this block -> virtual tile id (XCD-contiguous runs, see srf_xcd_remap; nblk % 8 == 0 keeps
  // every tile of a block on the block's own XCD) -> (M tile, time tile, example).
  // half: -1 = full tile, 0 / 1 = the half tile covering columns [64 half, 64 half + 64) of its parent tile.
  // Consecutive tiles of a block are nblk/8 virtual ids apart, so the three cursors below (load, convert, epilogue)
  // step their coordinates with two carries instead of dividing: the divisions by the run-time nMt / nLt cost
  // ~80 scalar instructions per evaluation, three evaluations per tile -- 14 % of a K = 256 tile's instruction
  // stream, on a kernel that is instruction-issue bound.
  struct TileCur {
    int i, v, mt, lt, b, half;
  };
  const int vstep = nblk >> 3;
  const int st_b = vstep / (nMt * nLt), st_r = vstep - st_b * (nMt * nLt);
  const int st_l = st_r / nMt, st_m = st_r - st_l * nMt;
  auto cur_set = [&](TileCur& c, int i) {   // by division: a block's first tile and the half-tile round
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

  const int a_m = tid >> 2, a_pk = tid & 3;
  const int a_lds = a_m * X3_PITCH + a_pk * 16;
  const int b_n = tid & 127, b_kg = (wave >> 1) * 8;
  const int b_lds = b_n * X3_PITCH + b_kg * 2;

  // ---- load cursor: (tile, k offset) of the next k-tile to fetch
  int ld_k = 0;
  TileCur ldc;
  const float* a_src;
  const float* b_src;
  __amdgpu_buffer_rsrc_t a_rs, b_rs;     // BUF: descriptors of W and of the tile's example X_b
  int a_vo = 0, b_vo = 0;                // BUF: per-lane byte offsets inside them
  // both descriptors are loop-invariant (whole W, whole X): only the per-lane offsets change from tile to tile.  (A
  // per-example descriptor rebuilt at every tile boundary mis-compiled in one instantiation: the flags word shared a
  // scalar register with a load offset.)
  if (BUF) {
    a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, Cout * Cin * 4, 0x00020000);
    b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.Bt * Cin * L * 4, 0x00020000);
  }
  auto ld_tile = [&]() {
    const int m0 = ldc.mt * X3_BM, l0 = ldc.lt * X3_BN;
    const long b = ldc.b;
    if (BUF) {
      a_vo = ((m0 + a_m) * Cin + a_pk * 8) * 4;                         // rows >= Cout land beyond the descriptor: 0
      b_vo = (((int)b * Cin + b_kg) * L + min(l0 + b_n, L - 1)) * 4;    // columns >= L are never stored
    } else {
      const bool a_ok = (m0 + a_m) < Cout;
      a_src = a.w + (size_t)(a_ok ? (m0 + a_m) : 0) * Cin + a_pk * 8;
      const bool b_ok = (l0 + b_n) < L;
      b_src = a.x + ((size_t)b * Cin + b_kg) * L + (b_ok ? (l0 + b_n) : 0);   // clamped; never stored if !ok
    }
  };
  cur_set(ldc, 0);
  ld_tile();
  struct Regs {
    float4 a[2];
    float b[8];
  };
  auto gload = [&](Regs& r) {
    if (BUF) {
      const auto w0 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo, ld_k * 4, 0);
      const auto w1 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo, ld_k * 4 + 16, 0);
      r.a[0] = make_float4(__uint_as_float(w0[0]), __uint_as_float(w0[1]), __uint_as_float(w0[2]), __uint_as_float(w0[3]));
      r.a[1] = make_float4(__uint_as_float(w1[0]), __uint_as_float(w1[1]), __uint_as_float(w1[2]), __uint_as_float(w1[3]));
#pragma unroll
      for (int j = 0; j < 8; ++j)
        r.b[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, b_vo, (ld_k + j) * L * 4, 0));
    } else {
      r.a[0] = *reinterpret_cast<const float4*>(a_src + ld_k);
      r.a[1] = *reinterpret_cast<const float4*>(a_src + ld_k + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.b[j] = b_src[(size_t)(ld_k + j) * L];
    }
    ld_k += X3_BK;
    if (ld_k == Cin) {   // wave-uniform; past the last tile the cursor re-reads that tile (harmless)
      ld_k = 0;
      if (ldc.i + 1 < ntile) {
        cur_next(ldc);
        ld_tile();
      }
    }
  };

  // ---- convert cursor: (tile, k offset) of the next k-tile to split into LDS
  int cv_k = 0;
  TileCur cvc;
  float a_msk = 0.f, mean = 0.f, rstd = 1.f;
  auto cv_tile = [&]() {
    a_msk = (BUF || (cvc.mt * X3_BM + a_m) < Cout) ? 1.f : 0.f;
    if (PRO == 1 || PRO == 2) srf_finalize_stats_dpp(a.nrm.sums, cvc.b, a.inv_count, mean, rstd);
  };
  cur_set(cvc, 0);
  cv_tile();
  auto lds_store = [&](const Regs& r, int stage) {
    char* base = smem + stage * X3_STAGE;
    const float va[8] = {r.a[0].x * a_msk, r.a[0].y * a_msk, r.a[0].z * a_msk, r.a[0].w * a_msk,
                         r.a[1].x * a_msk, r.a[1].y * a_msk, r.a[1].z * a_msk, r.a[1].w * a_msk};
    bf16x8 hi, lo;
    srf_split8(va, hi, lo);
    *reinterpret_cast<bf16x8*>(base + 0 * X3_IMG + a_lds) = hi;
    *reinterpret_cast<bf16x8*>(base + 1 * X3_IMG + a_lds) = lo;
    float vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x0 = r.b[j];
      if (PRO == 1 || PRO == 2) {
        const int k = cv_k + b_kg + j;
        const float sc = a.nrm.gamma[k] * rstd;
        x0 = fmaf(x0, sc, a.nrm.beta[k] - mean * sc);
      }
      if (PRO == 2 || PRO == 3) x0 = srf_prelu(x0, slope);
      vb[j] = x0;
    }
    srf_split8(vb, hi, lo);
    *reinterpret_cast<bf16x8*>(base + 2 * X3_IMG + b_lds) = hi;
    *reinterpret_cast<bf16x8*>(base + 3 * X3_IMG + b_lds) = lo;
    cv_k += X3_BK;
    if (cv_k == Cin) {
      cv_k = 0;
      if (cvc.i + 1 < ntile) {
        cur_next(cvc);
        cv_tile();
      }
    }
  };

  f32x16 acc0 = {0}, acc1 = {0};
  const int frag = (lane & 31) * X3_PITCH + (lane >> 5) * 16;
  const int a_row = (wm * 32) * X3_PITCH + frag;
  const int b_row0 = (wn * 64) * X3_PITCH + frag;
  int cur_half = -1;   // mode of the tile whose k-loop is running (wave-uniform)
  int br0 = b_row0;    // B fragment rows of the first accumulator for the running tile
  auto mma_tile = [&](int stage) {
    const char* base = smem + stage * X3_STAGE;
    const bool full = cur_half < 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks * 32;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + 0 * X3_IMG + a_row + ko);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + 1 * X3_IMG + a_row + ko);
      const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(base + 2 * X3_IMG + br0 + ko);
      const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(base + 3 * X3_IMG + br0 + ko);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc0, 0, 0, 0);
      if (full) {
        const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(base + 2 * X3_IMG + br0 + 32 * X3_PITCH + ko);
        const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(base + 3 * X3_IMG + br0 + 32 * X3_PITCH + ko);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc1, 0, 0, 0);
      }
    }
  };
  auto step = [&](Regs& nx, int g) {
    if (g + 1 < nsteps) lds_store(nx, (g + 1) & 1);
    gload(nx);
    mma_tile(g & 1);
    __syncthreads();
  };

  Regs r0, r1;
  gload(r0);
  gload(r1);
  lds_store(r0, 0);
  gload(r0);
  __syncthreads();
  float* strip = reinterpret_cast<float*>(smem + X3_STAGE) + wave * (32 * SRF_EPI_PITCH_H);   // stage 1
  int g = 0;
  TileCur epc;
  cur_set(epc, 0);
  for (int i = 0; i < ntile; ++i) {
    cur_half = (nhalf && i == rounds) ? (int)(blockIdx.x & 1) : -1;
    br0 = cur_half < 0 ? b_row0 : (cur_half * 64 + wn * 32) * X3_PITCH + frag;
    for (int kt = 0; kt < nk_; kt += 2) {
      step(r1, g);
      step(r0, g + 1);
      g += 2;
    }
    // the tile's last k-tile sat in stage 1, which every wave has finished reading (barrier above);
    // stage 0 already holds the next tile's first k-tile
    const int m0 = epc.mt * X3_BM, l0 = epc.lt * X3_BN, v = epc.v;
    const long b = epc.b;
    float s = 0.f, q = 0.f;
    if (cur_half < 0) {
      srf_pw_epilogue_half(a, acc0, strip, b, m0 + wm * 32, l0 + wn * 64, lane, s, q);
      srf_pw_epilogue_half(a, acc1, strip, b, m0 + wm * 32, l0 + wn * 64 + 32, lane, s, q);
    } else {
      srf_pw_epilogue_half(a, acc0, strip, b, m0 + wm * 32, l0 + cur_half * 64 + wn * 32, lane, s, q);
    }
    if (a.out_sums) {
      // (DPP wave sums: VALU only; the __shfl_xor form is 24 ds_bpermute per tile on every wavefront's critical path)
      const double ds = srf_dpp_wave_sum((double)s), dq = srf_dpp_wave_sum((double)q);
      if (lane == 63) {
        double* dst = srf_stat_slot(a.out_sums, b, (long)v * 16 + wave + (cur_half > 0 ? 8 : 0));
        atomicAdd(dst, ds);
        atomicAdd(dst + 1, dq);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    if (i + 1 < ntile) cur_next(epc);
    __syncthreads();   // strip reads done before the next step's split overwrites stage 1
  }
}

// buffer loads need 32-bit byte offsets inside W and inside X (debug flag 1<<27 forces the pointer form)
static bool srf_pw_buffer_ok(const PwArgs& a) {
  return (long)a.Cout * a.Cin * 4 < (1L << 31) && (long)a.Bt * a.Cin * a.L * 4 < (1L << 31) &&
         !(srf_debug_flags() & (1 << 27));
}

int srf_pw_bf16x3_launch(const PwArgs& a, int pro, hipStream_t st) {
  const int nMt = (a.Cout + X3_BM - 1) / X3_BM, nLt = (a.L + X3_BN - 1) / X3_BN;
  const long total = (long)a.Bt * nMt * nLt;
  SRF_CHECK_ARG(total < (1L << 31), "srf_pw_conv: too many tiles");
  {
    dim3 grid8((unsigned)total), block8(512);
    // Persistent blocks (2 per CU) whenever every block gets >= 3 tiles; fewer tiles and the idle slots
    // of the last round cost more than the pipelining across tiles gains (decoder frame GEMM: 800 tiles).
    // Debug flag 2048 forces the one-tile-per-block kernel.
    long nb = 2L * srf_device_cus();
    nb -= nb % 8;
    if (!(srf_debug_flags() & 2048) && nb >= 8 && total >= 3 * nb) {
      dim3 gridp((unsigned)nb);
      PwArgs ap = a;
      // leftover tiles of the last round as half tiles when they fill at most half of it (debug flag 256: off)
      const long rem = total % nb;
      const int nhalf = (rem > 0 && 2 * rem <= nb && !(srf_debug_flags() & 256)) ? (int)(2 * rem) : 0;
      // (Round 1 shipped without the <1, true> instantiation: it "computed wrong tiles in every run".  Cause, found in
      // round 2: the SLP vectorizer had packed that prologue's scale computation into v_pk_mul_f32 ... op_sel:[0,1], a form
      // that returns wrong low results in lanes 48..63 next to other wavefronts' MFMAs on gfx950 -- see build.py's ISA
      // lint.  This file is compiled with -fno-slp-vectorize now and all four prologues use buffer loads.)
      if (srf_pw_buffer_ok(a)) {
        switch (pro) {
          case 0: hipLaunchKernelGGL((srf_pw_bf16x3_p8_kernel<0, true>), gridp, block8, 0, st, ap, nMt, nLt, (int)total, nhalf); break;
          case 1: hipLaunchKernelGGL((srf_pw_bf16x3_p8_kernel<1, true>), gridp, block8, 0, st, ap, nMt, nLt, (int)total, nhalf); break;
          case 2: hipLaunchKernelGGL((srf_pw_bf16x3_p8_kernel<2, true>), gridp, block8, 0, st, ap, nMt, nLt, (int)total, nhalf); break;
          default: hipLaunchKernelGGL((srf_pw_bf16x3_p8_kernel<3, true>), gridp, block8, 0, st, ap, nMt, nLt, (int)total, nhalf); break;
        }
      } else
      switch (pro) {
        case 0: hipLaunchKernelGGL(srf_pw_bf16x3_p8_kernel<0>, gridp, block8, 0, st, ap, nMt, nLt, (int)total, nhalf); break;
        case 1: hipLaunchKernelGGL(srf_pw_bf16x3_p8_kernel<1>, gridp, block8, 0, st, ap, nMt, nLt, (int)total, nhalf); break;
        case 2: hipLaunchKernelGGL(srf_pw_bf16x3_p8_kernel<2>, gridp, block8, 0, st, ap, nMt, nLt, (int)total, nhalf); break;
        default: hipLaunchKernelGGL(srf_pw_bf16x3_p8_kernel<3>, gridp, block8, 0, st, ap, nMt, nLt, (int)total, nhalf); break;
      }
      // profiler labels per prologue variant = per rocprofv3 kernel name (srf_pw_bf16x3_p8_kernel<PRO, ...>)
      static const char* const kLabel[4] = {"pw_conv_bf16x3_p8<0>", "pw_conv_bf16x3_p8<1>", "pw_conv_bf16x3_p8<2>",
                                            "pw_conv_bf16x3_p8<3>"};
      SRF_CHECK_LAUNCH(kLabel[pro < 0 || pro > 3 ? 3 : pro], st);
      return SRF_OK;
    }
    const PwArgs& aw = a;
    if (!srf_pw_buffer_ok(a)) {
      switch (pro) {
        case 0: hipLaunchKernelGGL((srf_pw_bf16x3_w8_kernel<0, false>), grid8, block8, 0, st, aw, nMt, nLt, (int)total); break;
        case 1: hipLaunchKernelGGL((srf_pw_bf16x3_w8_kernel<1, false>), grid8, block8, 0, st, aw, nMt, nLt, (int)total); break;
        case 2: hipLaunchKernelGGL((srf_pw_bf16x3_w8_kernel<2, false>), grid8, block8, 0, st, aw, nMt, nLt, (int)total); break;
        default: hipLaunchKernelGGL((srf_pw_bf16x3_w8_kernel<3, false>), grid8, block8, 0, st, aw, nMt, nLt, (int)total); break;
      }
    } else
    switch (pro) {
      case 0: hipLaunchKernelGGL(srf_pw_bf16x3_w8_kernel<0>, grid8, block8, 0, st, aw, nMt, nLt, (int)total); break;
      case 1: hipLaunchKernelGGL(srf_pw_bf16x3_w8_kernel<1>, grid8, block8, 0, st, aw, nMt, nLt, (int)total); break;
      case 2: hipLaunchKernelGGL(srf_pw_bf16x3_w8_kernel<2>, grid8, block8, 0, st, aw, nMt, nLt, (int)total); break;
      default: hipLaunchKernelGGL(srf_pw_bf16x3_w8_kernel<3>, grid8, block8, 0, st, aw, nMt, nLt, (int)total); break;
    }
    SRF_CHECK_LAUNCH("pw_conv_bf16x3_w8", st);
    return SRF_OK;
  }
}
