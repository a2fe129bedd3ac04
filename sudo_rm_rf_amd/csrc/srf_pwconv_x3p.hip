// K2 (fast path, round 4) -- the 256 x 128 split-bf16 GEMM as TWO CO-RESIDENT BLOCKS PER CU (first form).
//
// srf_pwconv_x3w.hip keeps one 512-thread block per CU (144 KB of LDS, ~240 registers).  Here a block is half as heavy -- 16-k
// pipeline steps, three 24-KB stages (A image 16 KB by LDS-DMA, B image 8 KB), <= 128 registers (64 accumulators, ONE fragment
// set, four 4-register activation sets = loads four steps ahead) -- so that two blocks share a CU, 4 wavefronts per SIMD.  In
// isolation that is a tie with the one-block kernel (profiles/r04_NOTES.md: a SIMD either issues vector-memory instructions or
// feeds the matrix pipe, so more wavefronts add contenders, not overlap; one such block per CU: 130 us, two: 107-114, x3w:
// 106-118).  What it buys is INSIDE THE FORWARD: the engine runs a batch as two sub-batches on two streams (DESIGN.md "Two
// streams per forward"), and a GEMM block that owns half a CU instead of all of it lets the other stream's pyramid kernels
// (VALU- and HBM-bound, no LDS) co-reside with it: cfg 2 6.74 -> 6.36-6.45 ms on the same box, where the one-block kernel
// gains nothing from the split (profiles/r04_gemm_x3p_in_forward_ab.txt).
// Nothing else is clever: no fragment prefetch, static round-robin tiles, whole tiles only; the epilogue goes through
// wave-private LDS strips (float4 stores / residual loads: a quarter of the vector-memory instructions of storing straight from
// the MFMA layout, which measured 6.49 against 6.375 ms per cfg-2 forward and 22.26 against 21.78 ms at cfg 4, same box --
// profiles/r04_gemm_x3p_in_forward_ab.txt).
// Same arithmetic as srf_pwconv_x3w.hip in the same order (per 16 k: lo*hi, hi*lo, hi*hi; bias, then residual): the outputs are
// BIT-IDENTICAL to it; the statistics (fp64 buckets of fp32 partial sums) agree to rounding.
// Packed weights (second image of srf_x3w_pack_kernel): per (m-tile, 16-k step) [256 rows][hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15], 64-byte
// rows, 16-byte chunks XOR-swizzled like the B image: a stage's A image is one contiguous 16-KB DMA and both operands' fragments
// are conflict-free ds_read_b128.
// Forms: PRO 0-3 x EPI 0 (bias + statistics) | 1 (bias + residual) -- proj_1x1, res_conv, bottleneck and the backward's
// data-gradient GEMMs; Cout % 64 == 0.  The mask / fused-tail GEMM, the fp16-part training forward and every other shape stay
// with srf_pwconv_x3w.hip.  Reference sites: improved_sudormrf.py:256-259, :174, :196, :220.
#include <type_traits>

#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int P_BM = 256, P_BN = 128, P_KT = 16;
constexpr int P_A_IMG = P_BM * 64;                 // 16 KB
constexpr int P_B_IMG = P_BN * 64;                 // 8 KB
constexpr int P_STAGE = P_A_IMG + P_B_IMG;         // 24 KB
constexpr int P_NSTAGE = 3;
constexpr int P_MAX_STAT_EXAMPLES = 512;           // 4 KB behind the stages
constexpr int P_LDS_BYTES = P_NSTAGE * P_STAGE + P_MAX_STAT_EXAMPLES * 8;
static_assert(2 * P_LDS_BYTES <= 160 * 1024, "two blocks per CU");

__device__ __forceinline__ int p_swz(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

#define P_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// ---- packed weights: the image is written by srf_x3w_pack_kernel (srf_pwconv_x3w.hip, WPackEntry::dst16) in the launch that
// writes the one-block kernel's image -- same bf16 parts, laid out per 16-k step ---------------------------------------------
size_t srf_x3p_packed_bytes(int Cout, int Cin) {
  return (size_t)((Cout + P_BM - 1) / P_BM) * (size_t)(Cin / P_KT) * (size_t)P_A_IMG;
}

// CP (cache policy, as in srf_pwconv_x3w.hip): bit 0 = non-temporal output stores, bit 2 = non-temporal activation loads -- for the
// forms whose streamed tensors would otherwise displace the weight image every CU re-reads from L2 for every tile.
template <int PRO, int EPI, int CP = 0>
__global__ __launch_bounds__(512, 4) void srf_pw_x3p_kernel(PwArgs a, const char* __restrict__ wpack, int nMt, int nLt, int total,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ bias_r) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 wavefronts, 64 x 64 each
  const int Cin = a.Cin, L = a.L;
  const int nk = Cin / P_KT;                 // multiple of 4, >= 8 (host checks)
  const int nblk = gridDim.x;
  const int ntile = (total - (int)blockIdx.x + nblk - 1) / nblk;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;
  const int x_bytes = a.Bt * Cin * L * 4;

  // GlobLN statistics of every example, once per block (as in the shipped kernel)
  float2* stat_tab = reinterpret_cast<float2*>(smem + P_NSTAGE * P_STAGE);
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

  // ---- tiles: virtual id v = mt + nMt (lt + nLt b) (the nMt blocks that share an activation tile are neighbours on one XCD)
  struct TileCur {
    int v, mt, lt, b;
  };
  auto cur_set = [&](TileCur& c, int i) {
    const int v = srf_xcd_remap(blockIdx.x + i * nblk, total);
    const int t = v / nMt;
    c.v = v;
    c.mt = v - t * nMt;
    c.b = t / nLt;
    c.lt = t - c.b * nLt;
  };
  // ---- B staging: thread -> time step n = tid & 127, k rows 4 kg .. 4 kg + 3 (kg = tid >> 7, wave-uniform)
  const int b_n = tid & 127, kg = wave >> 1;
  const int b_hi = P_A_IMG + p_swz(b_n, kg >> 1) + (kg & 1) * 8;
  const int b_lo = P_A_IMG + p_swz(b_n, 2 + (kg >> 1)) + (kg & 1) * 8;
  __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);

  struct TileP {
    const char* a_src;   // this wavefront's 2-KB slice of the tile's packed weights, k-step 0 (wave-uniform)
    int b_vo;            // per-lane byte offset of (example, k row 4 kg, column) inside X; out of range = fetch nothing
    float mean, rstd;
  };
  auto make_tile = [&](const TileCur& c) {
    TileP t;
    t.a_src = wpack + (size_t)c.mt * nk * P_A_IMG + wave * 2048;
    const int col = c.lt * P_BN + b_n;
    t.b_vo = col < L ? ((c.b * Cin + 4 * kg) * L + col) * 4 : x_bytes;
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
    float b[4];
  };
  auto gload_a = [&](const TileP& t, int kt, int stage) __attribute__((always_inline)) {
    const char* src = t.a_src + (size_t)kt * P_A_IMG + lane * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)P_LDS(smem + stage * P_STAGE + wave * 2048));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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
    for (int j = 0; j < 4; ++j)
      r.b[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, t.b_vo, (kt * P_KT + j) * L * 4, (CP & 4) ? 2 : 0));
  };
  auto lds_store = [&](const Regs& r, const TileP& t, int kt, int stage) __attribute__((always_inline)) {
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = r.b[j];
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));   // (pins the conversion -- and its wait -- here)
    bf16x4 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x0 = x[j];
      if (PRO == 1 || PRO == 2) {
        const int k = kt * P_KT + 4 * kg + j;
        const float sc = gamma[k] * t.rstd;
        x0 = fmaf(x0, sc, beta[k] - t.mean * sc);
      }
      if (PRO == 2 || PRO == 3) x0 = srf_prelu(x0, slope);
      const __bf16 h = (__bf16)x0;
      ph[j] = h;
      pl[j] = (__bf16)(x0 - (float)h);
    }
    char* base = smem + stage * P_STAGE;
    *reinterpret_cast<bf16x4*>(base + b_hi) = ph;
    *reinterpret_cast<bf16x4*>(base + b_lo) = pl;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int fr = lane & 31, fc = lane >> 5;
  int a_off[2], b_off[2];   // hi fragments; the lo fragment of the same row sits at offset ^ 32 (logical chunk + 2)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    a_off[t] = p_swz(wm * 64 + t * 32 + fr, fc);
    b_off[t] = P_A_IMG + p_swz(wn * 64 + t * 32 + fr, fc);
  }
  struct Frags {
    bf16x8 ah[2], al[2], bh[2], bl[2];
  };
  auto read_frags = [&](Frags& f, int stage) __attribute__((always_inline)) {
    const char* base = smem + stage * P_STAGE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f.ah[t] = *reinterpret_cast<const bf16x8*>(base + a_off[t]);
      f.bh[t] = *reinterpret_cast<const bf16x8*>(base + b_off[t]);
      f.al[t] = *reinterpret_cast<const bf16x8*>(base + (a_off[t] ^ 32));
      f.bl[t] = *reinterpret_cast<const bf16x8*>(base + (b_off[t] ^ 32));
    }
  };
  auto mma = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
  };

  // One pipeline step = k-step kt of the current tile (stage s0): convert k-step kt+1 (register set (kt+1) & 3, loaded four steps
  // ago) into stage s1, start the DMA of k-step kt+2 into stage s2, reload the register set with k-step kt+5, multiply k-step kt.
  // k-step indices >= nk belong to the NEXT tile.  One barrier per step: before it every wavefront's ds_writes of k-step kt+1
  // are done and its DMA pieces of k-step kt+1 -- issued a step ago, 10 memory operations ago -- have landed.
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
  auto step = [&](Regs& set, int kt) __attribute__((always_inline)) {
    const int s1 = s0 == P_NSTAGE - 1 ? 0 : s0 + 1, s2 = s1 == P_NSTAGE - 1 ? 0 : s1 + 1;
    int k1, k2, k5;
    const TileP t1 = pick(kt + 1, k1), t2 = pick(kt + 2, k2), t5 = pick(kt + 5, k5);
    Frags f;
    read_frags(f, s0);
    lds_store(set, t1, k1, s1);
    gload_a(t2, k2, s2);
    gload_b(set, t5, k5);
    mma(f);
    asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    s0 = s1;
  };

  TileCur cur;
  cur_set(cur, 0);
  tc = make_tile(cur);
  TileCur nxc = cur;
  if (ntile > 1) cur_set(nxc, 1);
  tn = ntile > 1 ? make_tile(nxc) : tc;   // past the last tile the pipeline re-reads that tile (harmless)
  Regs r0, r1, r2, r3;
  gload_a(tc, 0, 0);
  gload_a(tc, 1, 1);
  gload_b(r0, tc, 0);
  gload_b(r1, tc, 1);
  gload_b(r2, tc, 2);
  gload_b(r3, tc, 3);
  lds_store(r0, tc, 0, 0);
  gload_b(r0, tc, 4);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int i = 0; i < ntile; ++i) {
    for (int kt = 0; kt < nk; kt += 4) {
      step(r1, kt);         // converts k-step kt+1 (r1), reloads it with k-step kt+5
      step(r2, kt + 1);
      step(r3, kt + 2);
      step(r0, kt + 3);
    }
    // ---- epilogue
    const int b = cur.b, m0 = cur.mt * P_BM + wm * 64, l0 = cur.lt * P_BN + wn * 64, v = cur.v;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));     // (keeps the offsets below from being hoisted out of the tile loop and spilled)
    const int lhalf = lane_o >> 5, lcol = lane_o & 31;
    float s = 0.f, q = 0.f;
    if (m0 < a.Cout) {      // (Cout % 64 == 0: a wavefront's 64 rows are in range together or not at all)
      __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * a.Cout * L, 0, a.Cout * L * 4, 0x00020000);
      __amdgpu_buffer_rsrc_t rrs = yrs;
      if constexpr (EPI == 1)
        rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.residual) + (size_t)b * a.Cout * L, 0, a.Cout * L * 4, 0x00020000);
      // Through a wave-private LDS strip in the stage the tile's last k-step has just freed (16 rows x 32 columns at a time,
      // pitch 36 floats): the accumulators go in by MFMA layout (register r of a 32 x 32 block = row (r & 3) + 8 (r >> 2) +
      // 4 (lane >> 5), column lane & 31) and come out as float4 along time, so that the residual loads and the stores are
      // 16-byte accesses -- 16 + 16 vector-memory instructions per wavefront and tile instead of 64 + 64 dword ones (a
      // wavefront's vector-memory instructions are what competes with its SIMD's MFMAs; LDS instructions do not).
      // Unit u = (mi, ni, half): the residual / bias of unit u + 1 is requested before unit u is evaluated.
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const int free_stage = s0 == 0 ? P_NSTAGE - 1 : s0 - 1;
      float* strip = reinterpret_cast<float*>(smem + free_stage * P_STAGE) + wave * (16 * SRF_EPI_PITCH_H);
      const int c4 = (lane_o & 7) * 4, rsub = lane_o >> 3;
      float4 rv[2][2];
      float bv[2][2];
      auto unit_off = [&](int u, int it, int& row, bool& ok) __attribute__((always_inline)) {
        const int mi = u >> 2, ni = (u >> 1) & 1, h = u & 1;
        row = m0 + mi * 32 + 16 * h + it * 8 + rsub;
        const int col = l0 + ni * 32 + c4;
        ok = col < L;                                  // (L % 4 == 0: a float4 is in range as a whole)
        return ok ? (row * L + col) * 4 : 0x7ffffff0;
      };
      auto res_issue = [&](int slot, int u) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          int row;
          bool ok;
          const int off = unit_off(u, it, row, ok);
          bv[slot][it] = a.bias[row];
          if constexpr (EPI == 1) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rrs, off, 0, 0);
            rv[slot][it] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
          }
        }
      };
      res_issue(0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int mi = u >> 2, ni = (u >> 1) & 1, h = u & 1;
        if (u + 1 < 8) res_issue((u + 1) & 1, u + 1);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          strip[((e & 3) + 8 * (e >> 2) + 4 * lhalf) * SRF_EPI_PITCH_H + lcol] = acc[mi][ni][8 * h + e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          int row;
          bool ok;
          const int off = unit_off(u, it, row, ok);
          float4 o = *reinterpret_cast<const float4*>(strip + (it * 8 + rsub) * SRF_EPI_PITCH_H + c4);
          const float bs = bv[u & 1][it];
          o.x += bs;
          o.y += bs;
          o.z += bs;
          o.w += bs;
          if constexpr (EPI == 1) {
            const float4 e4 = rv[u & 1][it];
            o.x += e4.x;
            o.y += e4.y;
            o.z += e4.z;
            o.w += e4.w;
          }
          const u32x4 ov = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
          __builtin_amdgcn_raw_buffer_store_b128(ov, yrs, off, 0, (CP & 1) ? 2 : 0);
          if constexpr (EPI == 0) {
            if (ok) {
              s += (o.x + o.y) + (o.z + o.w);
              q = fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, fmaf(o.w, o.w, q))));
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (EPI == 0 && a.out_sums) {
      const double ds = srf_dpp_wave_sum((double)s), dq = srf_dpp_wave_sum((double)q);
      if (lane == 63) {
        double* dst = srf_stat_slot(a.out_sums, b, (long)v * 32 + wave);
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
    // every wavefront's strip reads are done before the next step's DMA overwrites that stage
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cur = nxc;
    tc = tn;
    if (i + 2 < ntile) {
      cur_set(nxc, i + 2);
      tn = make_tile(nxc);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // surplus DMA of the pipeline tail must not outlive the block's LDS
}

bool srf_x3p_supported(const PwArgs& a, int pro) {
  if (a.Cin % 64 || a.Cin < 128 || a.Cout % 64 || a.L % 4) return false;   // (whole 64-row wavefront tiles: no row guards)
  if ((pro == 1 || pro == 2) && a.Bt > P_MAX_STAT_EXAMPLES) return false;
  if (a.epi_mask & 1) return false;                  // (the mask epilogue stays with the shipped kernel)
  if (a.residual && a.out_sums) return false;
  return true;
}

int srf_pw_x3p_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st) {
  const int nMt = (a.Cout + P_BM - 1) / P_BM, nLt = (a.L + P_BN - 1) / P_BN;
  const long total = (long)a.Bt * nMt * nLt;
  SRF_CHECK_ARG(total < (1L << 30), "srf_pw_conv: too many tiles");
  SRF_CHECK_ARG(srf_x3p_supported(a, pro), "srf_pw_conv: shape not served by the paired 256 x 128 kernel");
  const long ok = srf_device_cached(6, [](void*) -> long {
    bool good = true;
    const void* fns[] = {(const void*)&srf_pw_x3p_kernel<0, 0>, (const void*)&srf_pw_x3p_kernel<1, 0>,
                         (const void*)&srf_pw_x3p_kernel<2, 0>, (const void*)&srf_pw_x3p_kernel<3, 0>,
                         (const void*)&srf_pw_x3p_kernel<0, 1>, (const void*)&srf_pw_x3p_kernel<1, 1>,
                         (const void*)&srf_pw_x3p_kernel<2, 1>, (const void*)&srf_pw_x3p_kernel<3, 1>,
                         (const void*)&srf_pw_x3p_kernel<1, 0, 4>, (const void*)&srf_pw_x3p_kernel<2, 1, 4>};
    for (const void* f : fns) good &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES) == hipSuccess;
    return good ? 1 : 0;
  }, nullptr);
  SRF_CHECK_ARG(ok == 1, "srf_pw_conv: cannot reserve %d bytes of LDS", P_LDS_BYTES);
  long nb = 2L * srf_device_cus();      // two resident blocks per CU
  nb -= nb % 8;
  if (nb > total) nb = total;
  dim3 grid((unsigned)nb), block(512);
  const bool res = a.residual != nullptr;
#define P_GO(...) hipLaunchKernelGGL((srf_pw_x3p_kernel<__VA_ARGS__>), grid, block, P_LDS_BYTES, st, a, wpack, nMt, nLt, (int)total, a.nrm.gamma, a.nrm.beta, a.bias)
  if (pro == 1 && !res && !(srf_debug_flags() & 2)) {
    P_GO(1, 0, 4);                // bottleneck
  } else if (pro == 2 && res && !(srf_debug_flags() & 2)) {
    P_GO(2, 1, 4);                // res_conv (debug flag 2: the plain cache policy)
  } else if (!res) {
    switch (pro) {
      case 0: P_GO(0, 0); break;
      case 1: P_GO(1, 0); break;
      case 2: P_GO(2, 0); break;
      default: P_GO(3, 0); break;
    }
  } else {
    switch (pro) {
      case 0: P_GO(0, 1); break;
      case 1: P_GO(1, 1); break;
      case 2: P_GO(2, 1); break;
      default: P_GO(3, 1); break;
    }
  }
#undef P_GO
  static const char* const kLabel[4] = {"pw_conv_x3p<0>", "pw_conv_x3p<1>", "pw_conv_x3p<2>", "pw_conv_x3p<3>"};
  SRF_CHECK_LAUNCH(kLabel[pro < 0 || pro > 3 ? 3 : pro], st);
  return SRF_OK;
}
