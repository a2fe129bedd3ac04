// K2 (round 5) -- TWO 1x1 convolutions back to back in one launch, the intermediate tensor handed over IN REGISTERS:
//
//     y  = W1 f(x) + b1 (+ residual)        res_conv   improved_sudormrf.py:196,:220  (f = final_norm + PReLU, :195,:218)
//                                           bottleneck improved_sudormrf.py:256-259,:292 (f = ln, :254,:290)
//     y2 = W2 y + b2, {sum, sumsq}(y2)      proj_1x1 of the NEXT block, improved_sudormrf.py:174,:205 (no norm between :220 and :205)
//
// Why (profiles/r04_NOTES.md): in every form of the 256 x 128 GEMM a k-step cost MFMA time PLUS the time its
// wavefronts need to issue their vector-memory instructions, and proj_1x1 re-read from HBM what res_conv had just written from
// registers.  Here the k-loop of the second GEMM issues no activation load at all and the first GEMM's needs no LDS round trip:
//   * a wavefront owns ALL 256 output rows of conv 1 for 32 time steps (8 accumulator tiles of 32 x 32, 128 registers).  Its B
//     operand (the activations of its own 32 columns, 16 k rows per step) arrives as TWO 16-byte loads per lane and step, four
//     steps ahead, is transposed through the wavefront's private LDS strip into the MFMA B layout -- lane (n, h) holds
//     k = 16 kt + 8 h + 0..7 of column n -- and normalised / activated / split into bf16 hi | lo in registers, one step ahead
//     of its use: no shared B image, no block barrier on the activation path;
//   * after bias (+ residual) the 256 x 32 result is stored (float4 rows through a wave-private LDS strip, as srf_pwconv_x3p.hip)
//     and read back from the strip in B-operand order, split once into bf16 hi | lo: 16 k-blocks x (4 + 4) registers = the COMPLETE
//     K = 256 operand of conv 2 for these 32 columns, in registers;
//   * conv 2 then runs as C2 / 128 passes of 128 output rows (4 accumulator tiles) x 8 steps of 32 k with WEIGHT DMA ONLY.
// Weights: both convs stream the paired-block kernel's packed image (srf_x3w_pack_kernel's second image: per (256-row m-tile,
// 16-k step) [256 rows][hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15], XOR-swizzled 16-byte chunks) by LDS-DMA into three 16-KB
// stages; a stage is 256 "virtual rows" of 64 B: conv 1: the 256 output rows of one 16-k step; conv 2: two 16-k steps x 128
// output rows (two contiguous 8-KB halves of two images).  Every step of either phase is therefore the same: one 16-KB
// DMA (four 1-KB pieces per wavefront, two steps ahead), 16 conflict-free ds_read_b128 fragments, 24 MFMAs per wavefront, one
// barrier.  Blocks are 4 wavefronts / 75 KB of LDS / <= 256 registers: two per CU.
// Arithmetic and order per accumulator are those of srf_pwconv_x3p.hip / _x3w.hip (per 16 k: lo*hi, hi*lo, hi*hi; bias, then
// residual; the split of y is the split proj_1x1's prologue would make): y AND y2 are BIT-IDENTICAL to the two separate launches;
// the statistics (fp64 buckets of fp32 partial sums) agree to rounding.
// Shapes: conv 1 Cout = 256 (one m-tile: the block holds every k row of conv 2), K1 % 64 == 0, 128 <= K1 <= 512; conv 2
// C2 % 128 == 0, C2 <= 512; L % 4 == 0; the activation tensors within 32-bit buffer reach (srf_x3f_supported).
// Three users: srf_forward (PRO 1 / 2, bf16 parts: bottleneck / res_conv + the next proj_1x1), srf_forward_train (the same pairs
// on two fp16 parts, F16: bit-identical to srf_pwconv_x3w.hip NP = 4) and srf_backward (PRO 0, bf16 parts: the data gradients
// W_proj^T g + skip gradient, then W_res^T of it).
#include <type_traits>

#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// F16 (the training forward, srf_pwconv_x3w.hip NP = 4): operands split into two FP16 parts (hi = fp16(x), lo = fp16(x - hi): 22
// mantissa bits), weights stored times 2^4 (their lo parts stay normal), accumulators times 2^-4 in the epilogues (exact);
// `v_mfma_f32_32x32x16_f16` -- same shapes, same three products per block, same order.  Packets keep the bf16x8 container.
constexpr float F_F16_WSCALE_INV = 1.f / 16.f;
template <bool F16>
__device__ __forceinline__ f32x16 f_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// element j of the hi | lo packets of x0 (no range guard in the fp16 form: see w_split8_f16 in srf_pwconv_x3w.hip)
template <bool F16>
__device__ __forceinline__ void f_split1(float x0, bf16x8& hi, bf16x8& lo, int j) {
  if constexpr (F16) {
    const _Float16 hh = (_Float16)x0;
    hi[j] = __builtin_bit_cast(__bf16, hh);
    lo[j] = __builtin_bit_cast(__bf16, (_Float16)(x0 - (float)hh));
  } else {
    const __bf16 hh = (__bf16)x0;
    hi[j] = hh;
    lo[j] = (__bf16)(x0 - (float)hh);
  }
}

constexpr int F_BM = 256;                                  // rows of conv 1 = k of conv 2
constexpr int F_STAGE = 16384, F_NSTAGE = 3;
constexpr int F_STRIP_FLOATS = 32 * SRF_EPI_PITCH_H;       // 32 rows x 36 floats per wavefront
constexpr int F_MAX_K1 = 512, F_MAX_C2 = 512;
constexpr int F_G = 1;             // 32-row units per epilogue group (its stores / residual loads go out as one batch)
constexpr int F_OFF_STRIP = F_NSTAGE * F_STAGE;
constexpr int F_OFF_GB = F_OFF_STRIP + 4 * F_STRIP_FLOATS * 4;
constexpr int F_OFF_BIAS = F_OFF_GB + 2 * F_MAX_K1 * 4;
constexpr int F_LDS_BYTES = F_OFF_BIAS + (F_BM + F_MAX_C2) * 4;
static_assert(2 * F_LDS_BYTES <= 160 * 1024, "two blocks per CU");

__device__ __forceinline__ int f_swz(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }
#define F_LDS(p) ((__attribute__((address_space(3))) void*)(p))


template <int V>
using f_int = std::integral_constant<int, V>;

// PRO: 0 = none (the backward's data-gradient pair: W_proj^T g + skip gradient, then W_res^T of that), 1 = GlobLN,
// 2 = GlobLN + PReLU (conv 1's operand load).  EPI: 0 = bias, 1 = bias + residual (conv 1's epilogue).
// DRAIN: every counted wait of the DMA pipeline becomes vmcnt(0) -- the conservative form (debug flag 1 << 23), kept so that a
// test can hold the counted waits against it bit for bit (same results, ~1.5 % slower).
template <int PRO, int EPI, bool DRAIN, bool F16 = false>
__global__ __launch_bounds__(256, 2) void srf_pw_x3f_kernel(PwPairArgs a, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta) {
  static_assert(PRO >= 0 && PRO <= 2, "conv 1's prologue: none, GlobLN, GlobLN + PReLU");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int L = a.L, K1 = a.K1, C2 = a.C2;
  const int nk1 = K1 >> 4;                 // 16-k steps of conv 1 (multiple of 4, >= 8: host checks)
  const int npass = C2 >> 7;               // 128-row passes of conv 2
  const int nblk = gridDim.x;
  const int ntile = (a.total - (int)blockIdx.x + nblk - 1) / nblk;   // >= 1 (grid <= total)
  const float slope = PRO == 2 ? a.nrm.prelu[0] : 1.f;
  const int x_bytes = a.Bt * K1 * L * 4;

  float* strip = reinterpret_cast<float*>(smem + F_OFF_STRIP) + wave * F_STRIP_FLOATS;
  float* g_tab = reinterpret_cast<float*>(smem + F_OFF_GB);
  float* b_tab = g_tab + F_MAX_K1;
  float* bias1_t = reinterpret_cast<float*>(smem + F_OFF_BIAS);
  float* bias2_t = bias1_t + F_BM;

  // ---- per-block tables: biases, gamma / beta
  bias1_t[tid] = a.bias1[tid];
  for (int i = tid; i < C2; i += 256) bias2_t[i] = a.bias2[i];
  if constexpr (PRO != 0) {
    for (int i = tid; i < K1; i += 256) {
      g_tab[i] = gamma[i];
      b_tab[i] = beta[i];
    }
  }
  __syncthreads();

  // ---- tiles: tile v = 128 columns (4 wavefronts x 32) of one example; static round-robin over the persistent blocks
  struct Tile {
    int b, colw;         // example, first column of this wavefront's 32-column strip
    int x_vo;            // per-lane byte offset of (example, k row lane >> 3, this lane's float4 column) inside X; out of range = fetch nothing
    float mean, rstd;
  };
  auto make_tile = [&](int i) {
    int v = (int)blockIdx.x + i * nblk;
    Tile t;
    t.b = v / a.nLt;
    t.colw = (v - t.b * a.nLt) * 128 + wave * 32;
    const int col = t.colw + (lane & 7) * 4;       // (L % 4 == 0: a float4 is in range as a whole)
    t.x_vo = col < L ? ((t.b * K1 + (lane >> 3)) * L + col) * 4 : x_bytes;
    // GlobLN statistics of the tile's example: every wavefront reduces the 64 fp64 buckets itself (DPP; the same sums in the
    // same order as srf_pwconv_x3p.hip's table: identical {mean, rstd})
    t.mean = 0.f;
    t.rstd = 1.f;
    if constexpr (PRO != 0) srf_finalize_stats_dpp(a.nrm.sums, t.b, a.inv_count, t.mean, t.rstd);
    return t;
  };
  __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);

  // Everything below that is invariant across tiles (row offsets, DMA sources of unrolled steps ...) would be hoisted out of the
  // tile loop by hipcc and spilled (~300 registers in the first build): the tile loop re-derives it from values made opaque once per
  // tile (Lt, w1 / w2, lane_o), so it stays a handful of SALU / VALU instructions at the point of use.
  int Lt = L;                      // = L (opaque per tile)
  const char* w1 = a.wpack1;       // = a.wpack1 / a.wpack2 (opaque per tile)
  const char* w2 = a.wpack2;

  // ---- conv 1's B operand.  A 16-k step needs 16 k rows x 32 columns of X per wavefront.  Loaded in MFMA B order (lane (n, h):
  // rows 8 h + j of column n) that is eight dword loads per lane and step -- and on this chip the vector-memory instructions a
  // wavefront issues, not their bytes, are what a k-step costs on top of its MFMAs (profiles/r04_NOTES.md; the first build of
  // this kernel: 12 per step and wavefront, 3300 cycles per step against 770 of MFMA).  So the step's tile comes as TWO
  // 16-byte loads per lane (8 rows x 128 B each, non-temporal: read once) and is transposed through the wavefront's LDS strip
  // (idle during conv 1): float4 rows in, B-operand columns out -- the access pattern of epilogue 1.
  struct Regs {
    float4 v[2];
  };
  auto gload_b = [&](Regs& r, const Tile& t, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(x_rs, t.x_vo, (kt * 16 + it * 8) * Lt * 4, 2);
      r.v[it] = make_float4(__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3]));
    }
  };
  const int st_row = (lane >> 3) * SRF_EPI_PITCH_H + (lane & 7) * 4;     // float4 side of the strip: row lane >> 3 (+ 8 ..), 4 columns
  const int st_col = 8 * h * SRF_EPI_PITCH_H + n;                       // operand side: rows 8 h + j (+ 16) of column n
  // raw x of one 16-k step -> this lane's eight operand values (rows 8 h + j of column n)
  auto stage_x = [&](const Regs& r, float (&x)[8]) __attribute__((always_inline)) {
    *reinterpret_cast<float4*>(strip + st_row) = r.v[0];
    *reinterpret_cast<float4*>(strip + st_row + 8 * SRF_EPI_PITCH_H) = r.v[1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = strip[st_col + j * SRF_EPI_PITCH_H];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // element j of k-step kt's operand: GlobLN affine (+ PReLU), split into bf16 hi | lo (the arithmetic of srf_pwconv_x3p.hip)
  const float* g_lane = g_tab + 8 * h;
  const float* b_lane = b_tab + 8 * h;
  auto cvt1 = [&](float xv, int kt, int j, const Tile& t, bf16x8& bh, bf16x8& bl) __attribute__((always_inline)) {
    float x0 = xv;
    if constexpr (PRO != 0) {
      const float sc = g_lane[kt * 16 + j] * t.rstd;
      x0 = fmaf(xv, sc, b_lane[kt * 16 + j] - t.mean * sc);
    }
    if (PRO == 2) x0 = srf_prelu(x0, slope);
    f_split1<F16>(x0, bh, bl, j);
  };

  // ---- weight stages: step g of a tile's sequence (conv 1: g < nk1; conv 2: pass (g - nk1) >> 3, 32-k step (g - nk1) & 7).
  // Scalar base + per-lane 32-bit offsets (a per-lane 64-bit pointer per piece is what hipcc hoists out of the tile loop and
  // spills).  No instruction offset: it would be added on the LDS side as well (M0 already carries the piece's position).
  const int dma_l0 = lane * 16, dma_l1 = dma_l0 + 1024, dma_l2 = dma_l0 + 2048, dma_l3 = dma_l0 + 3072;
  // `after`: a value the statement pretends to read, so that hipcc cannot move it above the code that produces it.  (Where a
  // wavefront's vector-memory instructions sit matters: they do not issue while the SIMD's other wavefront runs its MFMA burst,
  // and the wavefront waits behind them in order -- placed in front of its own conversion work they would keep that work from
  // overlapping the partner's burst; placed behind it, right before its own burst, only the short issue itself is exposed.)
  auto dma_issue = [&](const char* src_any, int stage, unsigned after = 0u) __attribute__((always_inline)) {
    const size_t sa = (size_t)src_any;             // (wave-uniform by construction; hipcc needs to be told)
    const char* src = (const char*)(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(sa >> 32)) << 32) |
                                    (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa));
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)F_LDS(smem + stage * F_STAGE + wave * 4096));
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, %6, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
        "s_add_u32 m0, %6, 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
        "s_add_u32 m0, %6, 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(dma_l0), "v"(dma_l1), "v"(dma_l2), "v"(dma_l3), "s"(src), "s"(dst), "v"(after)
        : "memory", "scc");
  };
  // this wavefront's 4-KB quarter of the image of conv 1's 16-k step kt / of conv 2's (pass p, 32-k step ks): virtual rows
  // 0-127 = 16-k step 2 ks, rows 128-255 = step 2 ks + 1; both: rows 128 (p & 1) .. + 127 of m-tile p >> 1
  auto srcA = [&](int kt) __attribute__((always_inline)) { return w1 + (size_t)kt * F_STAGE + wave * 4096; };
  auto srcB = [&](int p, int ks) __attribute__((always_inline)) {
    return w2 + (size_t)((p >> 1) * (F_BM / 16) + 2 * ks + (wave >> 1)) * F_STAGE + (p & 1) * 8192 + (wave & 1) * 4096;
  };
  const int a_hi0 = f_swz(n, h), a_lo0 = a_hi0 ^ 32;   // fragment of virtual row 32 m + n: + 2048 m
  int s0 = 0;                                          // stage of the current step
  auto stage_after = [](int s, int d) __attribute__((always_inline)) {
    s += d;
    return s >= F_NSTAGE ? s - F_NSTAGE : s;
  };
  // End of a step: this wavefront's pieces of the NEXT step's image have landed (VM = vector-memory operations it has issued
  // since: counted per call site, see the table at the tile loop), its fragment reads are done; the barrier publishes the
  // next stage and frees the current one.
  auto end_step = [&](auto vm_tag) __attribute__((always_inline)) {
    constexpr int VM = DRAIN ? 0 : decltype(vm_tag)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM) : "memory");
    __builtin_amdgcn_s_barrier();
    s0 = stage_after(s0, 1);
  };

  // Issue order of a step's 16 fragment reads and 24 MFMAs: two fragment pairs in flight (16 registers), the next pair requested
  // behind the previous pair's MFMAs.  Left to itself hipcc hoists all 16 reads (64 registers) to the top of the step, runs out
  // of registers next to the 128 accumulators and starts copying accumulator tiles around the loop.
  auto frag_schedule = []() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
  };
  f32x16 acc[8];
  bf16x8 x2h[16], x2l[16];
  Tile tc = make_tile(0);

  // per-tile epilogue addressing: one per-lane offset (row lane >> 3 of a 16-row unit, this lane's float4 column), the unit's
  // first row in the scalar offset
  int ep_vo = 0;                   // (rsub Lt + col4) 4, or out of range
  float4 rv[F_G][4];               // residual of a group of F_G 32-row units (4 float4 rows per lane and unit)
  // part 0 .. 2 F_G - 1 of group g: two of its 4 F_G loads
  auto res_issue = [&](__amdgpu_buffer_rsrc_t rrs, int g, int part) __attribute__((always_inline)) {
    if constexpr (EPI == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int uu = part >> 1, it = 2 * (part & 1) + i;
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rrs, ep_vo, (32 * (F_G * g + uu) + it * 8) * Lt * 4, 2);
        rv[uu][it] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
      }
    }
  };

  // conv 1, one 16-k step kt.  The step multiplies the operand converted during the PREVIOUS step (bhC | blC) and, between its
  // MFMAs, converts the next one: STAGE = k-step kt + 1 (held raw in `set`) goes through the strip and is converted element by
  // element behind the eight accumulator tiles' MFMAs (the matrix pipe takes an MFMA every 32 cycles: ~6 issue slots per MFMA
  // are free for this wavefront's own VALU / LDS work -- in the first build the conversion ran in front of the burst, 1400
  // cycles per step with the pipe idle unless the SIMD's other wavefront happened to multiply); LOADS = `set` is reloaded with
  // k-step kt + 5; RES = part of the first residual group requested; DMAB = the stage fetched is conv 2's step DMAB (else conv
  // 1's kt + 2).
  bf16x8 bhC, blC;
  auto stepA = [&](auto stage_tag, auto loads_tag, auto res_tag, auto dmab_tag, auto vm_tag, Regs& set, int kt, const Tile& t,
                   __amdgpu_buffer_rsrc_t rrs) __attribute__((always_inline)) {
    constexpr bool STAGE = decltype(stage_tag)::value != 0, LOADS = decltype(loads_tag)::value != 0;
    constexpr int RES = decltype(res_tag)::value, DMAB = decltype(dmab_tag)::value;
    float xin[8];
    unsigned dep = 0u;
    if constexpr (STAGE) {
      stage_x(set, xin);
      dep = __float_as_uint(xin[7]);
    }
    if constexpr (DMAB < 0) dma_issue(srcA(kt + 2), stage_after(s0, 2), dep);
    else dma_issue(srcB(0, DMAB), stage_after(s0, 2), dep);
    if constexpr (LOADS) gload_b(set, t, kt + 5);
    if constexpr (RES >= 0) res_issue(rrs, 0, RES);
    const char* base = smem + s0 * F_STAGE;
    bf16x8 ah[2], al[2], bhN, blN;
    ah[0] = *reinterpret_cast<const bf16x8*>(base + a_hi0);
    al[0] = *reinterpret_cast<const bf16x8*>(base + a_lo0);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      if (mt + 1 < 8) {      // the next tile's pair is requested before this one's MFMAs
        ah[(mt + 1) & 1] = *reinterpret_cast<const bf16x8*>(base + a_hi0 + (mt + 1) * 2048);
        al[(mt + 1) & 1] = *reinterpret_cast<const bf16x8*>(base + a_lo0 + (mt + 1) * 2048);
      }
      acc[mt] = f_mfma<F16>(al[mt & 1], bhC, acc[mt]);
      acc[mt] = f_mfma<F16>(ah[mt & 1], blC, acc[mt]);
      acc[mt] = f_mfma<F16>(ah[mt & 1], bhC, acc[mt]);
      if constexpr (STAGE) cvt1(xin[mt], kt + 1, mt, t, bhN, blN);
    }
    end_step(vm_tag);
    if constexpr (STAGE) {
      bhC = bhN;
      blC = blN;
    }
  };

  // ---- prologue: the first two weight stages and the first four activation sets
  Regs r0, r1, r2, r3;
  dma_issue(srcA(0), 0);
  dma_issue(srcA(1), 1);
  gload_b(r0, tc, 0);
  gload_b(r1, tc, 1);
  gload_b(r2, tc, 2);
  gload_b(r3, tc, 3);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // Counted waits (vector-memory operations a wavefront issues after the last DMA piece of step g + 1, up to the end of step g;
  // a smaller number is always safe).  A load step issues DMA x 4, then 2 loads; a no-load step DMA x 4, then 2 residual loads
  // (EPI 1, the last four steps); epilogue 1: per two-unit group 8 stores + the next group's 8 residual loads; a conv-2 step
  // DMA x 4; epilogue 2: 16 stores (+ 2 atomics after the last pass); the next tile's 8 activation loads go out before the last
  // epilogue 2, its first set's reload (2) in front of its step 0.
  //   conv 1, load step after a load step         2 + 4 + 2        = 8    (also used for step 0 of a tile: >= 32 there)
  //   conv 1, first no-load step                  2 + 4            = 6
  //   conv 1, second no-load step                 0 + 4 (+ 2)      = 4 / 6
  //   conv 1, later no-load steps                 (2 +) 4 (+ 2)    = 4 / 8
  //   conv 2, first step of a pass                >= 16 stores + 4 = 20   (pass 0: 32 stores + 4)
  //   conv 2, other steps                                            4
  constexpr int VM_A = 8, VM_A_NL0 = 6, VM_A_NL1 = EPI == 1 ? 6 : 4, VM_A_NL = EPI == 1 ? 8 : 4, VM_BP = 20, VM_B = 4;

  int i = 0;
  do {
    Tile tn = tc;
    if (i + 1 < ntile) tn = make_tile(i + 1);
    int lane_o = lane;
    asm volatile("" : "+s"(Lt), "+s"(w1), "+s"(w2), "+v"(lane_o));   // (see above: nothing below is hoisted out of this loop)
    const int b = tc.b;
    const int c4 = (lane_o & 7) * 4, rsub = lane_o >> 3;
    const int col4 = tc.colw + c4;
    const bool ok4 = col4 < Lt;                        // (L % 4 == 0: a float4 is in range as a whole)
    ep_vo = ok4 ? (rsub * Lt + col4) * 4 : 0x7ffffff0;
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * F_BM * Lt, 0, F_BM * Lt * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rrs = yrs;
    if constexpr (EPI == 1)
      rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.residual) + (size_t)b * F_BM * Lt, 0, F_BM * Lt * 4, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // ================= conv 1: nk1 steps of 16 k =================
    {   // the first step's operand (the only conversion a tile does in front of its MFMAs); its set goes back out for k-step 4
      float x0[8];
      stage_x(r0, x0);
#pragma unroll
      for (int j = 0; j < 8; ++j) cvt1(x0[j], 0, j, tc, bhC, blC);
      gload_b(r0, tc, 4);
    }
    // (do-while: in the for form the loop exits at its header, the accumulators' header values are live out of the loop, the
    // first MFMA of the body no longer kills its accumulator operand, hipcc picks the untied MFMA form for it and every
    // accumulator tile exists twice -- +128 registers, ~300 spilled)
    int kt = 0;
    if (nk1 > 8) {
      do {     // step kt stages / reloads the set of k-step kt + 1: r1, r2, r3, r0
        stepA(f_int<1>{}, f_int<1>{}, f_int<-1>{}, f_int<-1>{}, f_int<VM_A>{}, r1, kt, tc, rrs);
        stepA(f_int<1>{}, f_int<1>{}, f_int<-1>{}, f_int<-1>{}, f_int<VM_A>{}, r2, kt + 1, tc, rrs);
        stepA(f_int<1>{}, f_int<1>{}, f_int<-1>{}, f_int<-1>{}, f_int<VM_A>{}, r3, kt + 2, tc, rrs);
        stepA(f_int<1>{}, f_int<1>{}, f_int<-1>{}, f_int<-1>{}, f_int<VM_A>{}, r0, kt + 3, tc, rrs);
        kt += 4;
      } while (kt < nk1 - 8);
    }
    // the last eight steps: reloads stop after three (k-step kt + 5 >= nk1), the last four request the first residual group,
    // the last two fetch conv 2's first stages, the last one has no next operand
    stepA(f_int<1>{}, f_int<1>{}, f_int<-1>{}, f_int<-1>{}, f_int<VM_A>{}, r1, kt, tc, rrs);
    stepA(f_int<1>{}, f_int<1>{}, f_int<-1>{}, f_int<-1>{}, f_int<VM_A>{}, r2, kt + 1, tc, rrs);
    stepA(f_int<1>{}, f_int<1>{}, f_int<-1>{}, f_int<-1>{}, f_int<VM_A>{}, r3, kt + 2, tc, rrs);
    stepA(f_int<1>{}, f_int<0>{}, f_int<-1>{}, f_int<-1>{}, f_int<VM_A_NL0>{}, r0, kt + 3, tc, rrs);
    static_assert(F_G == 1 || F_G == 2, "residual parts of the first group: the last 2 F_G steps");
    // (counted waits: the step before a residual part issued none -> VM_A_NL1, one -> VM_A_NL; no part in this step or the last: 4)
    stepA(f_int<1>{}, f_int<0>{}, f_int<(F_G == 2 ? 0 : -1)>{}, f_int<-1>{}, f_int<(F_G == 2 ? VM_A_NL1 : 4)>{}, r1, kt + 4, tc, rrs);
    stepA(f_int<1>{}, f_int<0>{}, f_int<(F_G == 2 ? 1 : -1)>{}, f_int<-1>{}, f_int<(F_G == 2 ? VM_A_NL : 4)>{}, r2, kt + 5, tc, rrs);
    stepA(f_int<1>{}, f_int<0>{}, f_int<(F_G == 2 ? 2 : 0)>{}, f_int<0>{}, f_int<(F_G == 2 ? VM_A_NL : VM_A_NL1)>{}, r3, kt + 6, tc, rrs);
    stepA(f_int<0>{}, f_int<0>{}, f_int<(F_G == 2 ? 3 : 1)>{}, f_int<1>{}, f_int<VM_A_NL>{}, r0, kt + 7, tc, rrs);

    // ================= epilogue 1: y = acc + bias (+ residual) -> HBM, and -> conv 2's B operand =================
    // Unit t = accumulator tile t (rows 32 t .. 32 t + 31): MFMA layout -> strip -> float4 rows (bias, residual; the sum goes
    // back into the strip and into a register batch) -> strip columns in B-operand order (lane (n, h): rows 16 c + 8 h + j =
    // conv 2's k-block 2 t + c).  The stores and the next group's residual loads go out in batches of a two-unit group -- 16
    // vector-memory instructions back to back (see dma_issue: a wavefront stalls at a vector-memory instruction for as long as
    // its SIMD partner multiplies, with everything behind it).
#pragma unroll
    for (int g2 = 0; g2 < 8 / F_G; ++g2) {
      u32x4 ob[F_G][4];
#pragma unroll
      for (int uu = 0; uu < F_G; ++uu) {
        const int t = F_G * g2 + uu;
#pragma unroll
        for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * h) * SRF_EPI_PITCH_H + n] = acc[t][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (all reads of a phase first, ONE wait, then the arithmetic, then all writes: read - modify - write per row makes hipcc
        // keep the may-alias order and pay an LDS round trip per row; same for the 16 operand reads below)
        float4 o[4];
        float bs[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          o[it] = *reinterpret_cast<const float4*>(strip + (it * 8 + rsub) * SRF_EPI_PITCH_H + c4);
          bs[it] = bias1_t[32 * t + it * 8 + rsub];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) asm volatile("" : "+v"(o[it].x), "+v"(o[it].y), "+v"(o[it].z), "+v"(o[it].w), "+v"(bs[it]));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if constexpr (F16) {       // (the weights were stored times 2^4)
            o[it].x *= F_F16_WSCALE_INV;
            o[it].y *= F_F16_WSCALE_INV;
            o[it].z *= F_F16_WSCALE_INV;
            o[it].w *= F_F16_WSCALE_INV;
          }
          o[it].x += bs[it];
          o[it].y += bs[it];
          o[it].z += bs[it];
          o[it].w += bs[it];
          if constexpr (EPI == 1) {
            const float4 e4 = rv[uu][it];
            o[it].x += e4.x;
            o[it].y += e4.y;
            o[it].z += e4.z;
            o[it].w += e4.w;
          }
          ob[uu][it] = u32x4{__float_as_uint(o[it].x), __float_as_uint(o[it].y), __float_as_uint(o[it].z), __float_as_uint(o[it].w)};
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<float4*>(strip + (it * 8 + rsub) * SRF_EPI_PITCH_H + c4) = o[it];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float xr[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) xr[j] = strip[st_col + ((j >> 3) * 16 + (j & 7)) * SRF_EPI_PITCH_H];
        asm volatile("" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]),
                          "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]), "+v"(xr[15]));
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          bf16x8 ph, pl;
#pragma unroll
          for (int j = 0; j < 8; ++j) f_split1<F16>(xr[8 * c + j], ph, pl, j);
          x2h[2 * t + c] = ph;
          x2l[2 * t + c] = pl;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
#pragma unroll
      for (int uu = 0; uu < F_G; ++uu)
#pragma unroll
        for (int it = 0; it < 4; ++it)
          __builtin_amdgcn_raw_buffer_store_b128(ob[uu][it], yrs, ep_vo, (32 * (F_G * g2 + uu) + it * 8) * Lt * 4, 2);
      if (g2 + 1 < 8 / F_G) {
#pragma unroll
        for (int part = 0; part < 2 * F_G; ++part) res_issue(rrs, g2 + 1, part);
      }
    }

    // ================= conv 2: npass passes of 128 rows x 8 steps of 32 k, B operand from registers =================
    __amdgpu_buffer_rsrc_t y2rs = __builtin_amdgcn_make_buffer_rsrc(a.y2 + (size_t)b * C2 * Lt, 0, C2 * Lt * 4, 0x00020000);
    float s = 0.f, q = 0.f;
    f32x16 acc2[4];
    auto passB = [&](int p) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks < 6) {
          dma_issue(srcB(p, ks + 2), stage_after(s0, 2));
        } else {       // the next pass's first two images; the last pass fetches conv 1's first two for the next tile
          const char* sa = srcA(ks - 6);
          const char* sb = srcB(p + 1, ks - 6);
          dma_issue(p + 1 < npass ? sb : sa, stage_after(s0, 2));
        }
        const char* base = smem + s0 * F_STAGE;
        bf16x8 ah[2], al[2];
        ah[0] = *reinterpret_cast<const bf16x8*>(base + a_hi0);
        al[0] = *reinterpret_cast<const bf16x8*>(base + a_lo0);
#pragma unroll
        for (int f = 0; f < 8; ++f) {            // fragment pair f = 4 kb + mt (virtual rows 32 f ..)
          const int kb = f >> 2, mt = f & 3;
          if (f + 1 < 8) {
            ah[(f + 1) & 1] = *reinterpret_cast<const bf16x8*>(base + a_hi0 + (f + 1) * 2048);
            al[(f + 1) & 1] = *reinterpret_cast<const bf16x8*>(base + a_lo0 + (f + 1) * 2048);
          }
          acc2[mt] = f_mfma<F16>(al[f & 1], x2h[2 * ks + kb], acc2[mt]);
          acc2[mt] = f_mfma<F16>(ah[f & 1], x2l[2 * ks + kb], acc2[mt]);
          acc2[mt] = f_mfma<F16>(ah[f & 1], x2h[2 * ks + kb], acc2[mt]);
        }
        frag_schedule();
        if (ks == 0) end_step(f_int<VM_BP>{});
        else end_step(f_int<VM_B>{});
      }
    };
    // epilogue 2 of pass p: y2 rows 128 p .. 128 p + 127 = acc2 + bias, {sum, sumsq}
    auto epiB = [&](int p) __attribute__((always_inline)) {
#pragma unroll
      for (int g2 = 0; g2 < 4 / F_G; ++g2) {
        u32x4 ob[F_G][4];
#pragma unroll
        for (int uu = 0; uu < F_G; ++uu) {
          const int t = F_G * g2 + uu;
#pragma unroll
          for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * h) * SRF_EPI_PITCH_H + n] = acc2[t][r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          float4 o4[4];
          float bs[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            o4[it] = *reinterpret_cast<const float4*>(strip + (it * 8 + rsub) * SRF_EPI_PITCH_H + c4);
            bs[it] = bias2_t[128 * p + 32 * t + it * 8 + rsub];
          }
#pragma unroll
          for (int it = 0; it < 4; ++it) asm volatile("" : "+v"(o4[it].x), "+v"(o4[it].y), "+v"(o4[it].z), "+v"(o4[it].w), "+v"(bs[it]));
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            float4 o = o4[it];
            if constexpr (F16) {
              o.x *= F_F16_WSCALE_INV;
              o.y *= F_F16_WSCALE_INV;
              o.z *= F_F16_WSCALE_INV;
              o.w *= F_F16_WSCALE_INV;
            }
            o.x += bs[it];
            o.y += bs[it];
            o.z += bs[it];
            o.w += bs[it];
            ob[uu][it] = u32x4{__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
            if (ok4) {
              s += (o.x + o.y) + (o.z + o.w);
              q = fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, fmaf(o.w, o.w, q))));
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int uu = 0; uu < F_G; ++uu)
#pragma unroll
          for (int it = 0; it < 4; ++it)
            __builtin_amdgcn_raw_buffer_store_b128(ob[uu][it], y2rs, ep_vo, (128 * p + 32 * (F_G * g2 + uu) + it * 8) * Lt * 4, 0);
      }
    };
    for (int p = 0; p < npass - 1; ++p) {
      passB(p);
      epiB(p);
    }
    passB(npass - 1);
    // the next tile's first four activation sets travel during the last epilogue (conv 2's operand registers are free now)
    if (i + 1 < ntile) {
      gload_b(r0, tn, 0);
      gload_b(r1, tn, 1);
      gload_b(r2, tn, 2);
      gload_b(r3, tn, 3);
    }
    epiB(npass - 1);
    if (a.out_sums2) {
      const double ds = srf_dpp_wave_sum((double)s), dq = srf_dpp_wave_sum((double)q);
      if (lane == 63) {
        double* dst = srf_stat_slot(a.out_sums2, b, ((long)blockIdx.x + (long)i * nblk) * 4 + wave);
        atomicAdd(dst, ds);
        atomicAdd(dst + 1, dq);
      }
    }
    tc = tn;
  } while (++i < ntile);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // surplus DMA of the pipeline tail must not outlive the block's LDS
}

bool srf_x3f_supported(int Bt, int K1, int C2, int L) {
  if (K1 % 64 || K1 < 128 || K1 > F_MAX_K1 || C2 % 128 || C2 < 128 || C2 > F_MAX_C2 || L % 4) return false;
  if (Bt < 1) return false;
  // 32-bit buffer reach: x as one resource over the batch; y / the residual (256 rows) and y2 (C2 rows) as one resource per example
  if ((long)Bt * K1 * L * 4 >= (1L << 31) || (long)C2 * L * 4 >= (1L << 31) || (long)F_BM * L * 4 >= (1L << 31)) return false;
  return true;
}

// wpack1 / wpack2: the PAIRED-BLOCK image of the two weights (srf_x3p_packed_bytes; the second image of a packed buffer)
// f16: the training forward's form (packed3 fp16 images, PRO 1 / 2 only)
int srf_pw_x3f_launch(const PwPairArgs& a0, int pro, hipStream_t st, bool f16) {
  PwPairArgs a = a0;
  SRF_CHECK_ARG(srf_x3f_supported(a.Bt, a.K1, a.C2, a.L), "srf_pw_conv_pair: shape not served by the fused pair kernel");
  SRF_CHECK_ARG(pro >= 0 && pro <= 2, "srf_pw_conv_pair: prologue %d", pro);
  SRF_CHECK_ARG((pro != 1) == (a.residual != nullptr),
                "srf_pw_conv_pair: built forms: GlobLN + PReLU with residual, GlobLN without, no prologue with residual");
  a.nLt = (a.L + 127) / 128;
  const long total = (long)a.Bt * a.nLt;
  SRF_CHECK_ARG(total < (1L << 30), "srf_pw_conv_pair: too many tiles");
  a.total = (int)total;
  const bool drain = (srf_debug_flags() & (1 << 23)) != 0;
  const long ok = srf_device_cached(7, [](void*) -> long {
    bool good = true;
    const void* fns[] = {(const void*)&srf_pw_x3f_kernel<1, 0, false>, (const void*)&srf_pw_x3f_kernel<2, 1, false>,
                         (const void*)&srf_pw_x3f_kernel<1, 0, true>, (const void*)&srf_pw_x3f_kernel<2, 1, true>,
                         (const void*)&srf_pw_x3f_kernel<0, 1, false>, (const void*)&srf_pw_x3f_kernel<0, 1, true>,
                         (const void*)&srf_pw_x3f_kernel<1, 0, false, true>, (const void*)&srf_pw_x3f_kernel<2, 1, false, true>};
    for (const void* f : fns) good &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES) == hipSuccess;
    return good ? 1 : 0;
  }, nullptr);
  SRF_CHECK_ARG(ok == 1, "srf_pw_conv_pair: cannot reserve %d bytes of LDS", F_LDS_BYTES);
  // ONE TILE PER BLOCK up to 16 rounds of the chip's 2 x CUs block slots (beyond: persistent blocks, static round-robin).  A tile
  // takes a block the same time whether or not a second block shares its CU, so what matters is that no slot idles while tiles
  // are left: with one tile per block the dispatcher hands the next tile to whichever slot frees first (cfg 2: 800 tiles on 512
  // slots = 1.56 rounds instead of the 2 full rounds of a static split).
  const long slots = 2L * srf_device_cus();
  long nb = total <= 16 * slots && !(srf_debug_flags() & (1 << 21)) ? total : slots - slots % 8;   // (flag 1 << 21: always persistent -- tests)
  if (nb > total) nb = total;
  dim3 grid((unsigned)nb), block(256);
#define F_GO(...) hipLaunchKernelGGL((srf_pw_x3f_kernel<__VA_ARGS__>), grid, block, F_LDS_BYTES, st, a, a.nrm.gamma, a.nrm.beta)
  if (f16) {
    SRF_CHECK_ARG(pro == 1 || pro == 2, "srf_pw_conv_pair (fp16 parts): prologue %d not built", pro);
    if (pro == 1) F_GO(1, 0, false, true);
    else F_GO(2, 1, false, true);
  } else if (pro == 1) {
    if (drain) F_GO(1, 0, true);
    else F_GO(1, 0, false);
  } else if (pro == 2) {
    if (drain) F_GO(2, 1, true);
    else F_GO(2, 1, false);
  } else {
    if (drain) F_GO(0, 1, true);
    else F_GO(0, 1, false);
  }
#undef F_GO
  static const char* const kLabel[3] = {"pw_pair_x3f<0>", "pw_pair_x3f<1>", "pw_pair_x3f<2>"};
  static const char* const kLabel4[3] = {"pw_pair_x3f4<0>", "pw_pair_x3f4<1>", "pw_pair_x3f4<2>"};
  SRF_CHECK_LAUNCH(f16 ? kLabel4[pro] : kLabel[pro], st);
  return SRF_OK;
}
