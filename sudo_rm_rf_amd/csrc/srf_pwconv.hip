// K2 -- pointwise (kernel_size=1) Conv1d as a GEMM over [channel, time], with the neighbouring
// GlobLN / PReLU folded into the operand load and bias / residual / ReLU-mask / GlobLN statistics
// folded into the epilogue.  Sites (improved_sudormrf.py): bottleneck :256-259,:292 (prologue = ln),
// proj_1x1 :174,:205 (epilogue statistics for its GlobLN), res_conv :196,:220 (prologue =
// final_norm + PReLU, epilogue = + residual), mask_net :268-269,:295-298 (prologue = PReLU,
// epilogue = ReLU * encoder output), and the decoder's frame GEMM (srf_api.hip).
//
//   y[b, m, l] = bias[m] + sum_k w[m, k] * f(x[b, k, l])        W: [Cout, Cin] row-major (Conv1d
//   weight [Cout, Cin, 1]), X_b: [Cin, L] row-major -> time is the contiguous GEMM-N dimension.
//
// MFMA path: exact-fp32 v_mfma_f32_32x32x2_f32 (gfx950 has no xf32/TF32; bf16 operands break the
// 1e-4 parity bar), 128x128 block tile, BK=16, 4 wavefronts each owning a 64x64 quadrant (2x2 MFMA
// tiles, 64 accumulator VGPRs), operands staged global -> registers (transform) -> LDS in [k][m] /
// [k][n] layout so every MFMA operand fetch is a conflict-free ds_read_b32, register prefetch of the
// next k-tile under the current tile's MFMAs, one barrier per k-tile (double-buffered LDS).
// Blocks are numbered so that the M-tiles sharing one X tile run back-to-back on one XCD (L2 reuse).
// Generic path (any shape, small channel counts): one thread per time step, MT outputs in registers,
// weights as scalar operands.
#include "srf_common.h"

#include "srf_pw.h"
#include <iterator>
#include <map>
#include <mutex>
#include <vector>

// ---------------------------------------------------------------------------------------------
// generic VALU kernel
// ---------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void srf_pw_generic_kernel(PwArgs a) {
  __shared__ double red[8];
  const int l = blockIdx.x * 256 + threadIdx.x;
  const int m0 = blockIdx.y * MT;
  const long b = blockIdx.z;
  const bool valid = l < a.L;
  float mean = 0.f, rstd = 1.f;
  const bool has_norm = a.nrm.sums != nullptr;
  if (has_norm) srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
  const bool act = a.nrm.prelu != nullptr;
  const float slope = act ? a.nrm.prelu[0] : 1.f;
  float acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = 0.f;
  const float* xb = a.x + (size_t)b * a.Cin * a.L;
  for (int k = 0; k < a.Cin; ++k) {
    float xv = valid ? xb[(size_t)k * a.L + l] : 0.f;
    if (has_norm) {
      const float sc = a.nrm.gamma[k] * rstd;
      xv = fmaf(xv, sc, a.nrm.beta[k] - mean * sc);
    }
    if (act) xv = srf_prelu(xv, slope);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + i;
      const float wv = (m < a.Cout) ? a.w[(size_t)m * a.Cin + k] : 0.f;
      acc[i] = fmaf(wv, xv, acc[i]);
    }
  }
  double ds = 0.0, dq = 0.0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + i;
    if (m < a.Cout && valid) {
      const size_t idx = ((size_t)b * a.Cout + m) * a.L + l;
      float v = acc[i] + a.bias[m];
      if (a.residual) v += a.residual[idx];
      if (a.epi_mask & 1)
        v = fmaxf(v, 0.f) * a.mul[((size_t)b * a.mul_channels + (m % a.mul_channels)) * a.L + l];
      a.y[idx] = v;
      ds += (double)v;
      dq += (double)v * (double)v;
    }
  }
  if (a.out_sums)
    srf_block_stats_atomic<4>(ds, dq, srf_stat_slot(a.out_sums, b, blockIdx.x + blockIdx.y * gridDim.x), red);
}

// ---------------------------------------------------------------------------------------------
// MFMA kernel
// ---------------------------------------------------------------------------------------------
constexpr int PW_BM = 128, PW_BN = 128, PW_BK = 16;
constexpr int PW_LDA = PW_BM + 4;  // padded row pitch (floats) of the [k][m] / [k][n] LDS tiles
constexpr int PW_LDB = PW_BN + 4;

// PRO: 0 = identity, 1 = GlobLN, 2 = GlobLN + PReLU, 3 = PReLU only
template <int PRO>
__global__ __launch_bounds__(256) void srf_pw_mfma_kernel(PwArgs a, int nMt, int nLt, int total) {
  // operand tiles, later re-used as the epilogue's per-wave staging strips (4 waves x 2 strips)
  constexpr int kTileFloats = 2 * PW_BK * PW_LDA + 2 * PW_BK * PW_LDB;
  constexpr int kEpiFloats = 4 * 32 * SRF_EPI_PITCH;   // one strip per wave, used twice
  __shared__ __attribute__((aligned(16))) float smem_f[kTileFloats > kEpiFloats ? kTileFloats : kEpiFloats];
  __shared__ double red[8];
  float(*As)[PW_BK][PW_LDA] = reinterpret_cast<float(*)[PW_BK][PW_LDA]>(smem_f);
  float(*Bs)[PW_BK][PW_LDB] = reinterpret_cast<float(*)[PW_BK][PW_LDB]>(smem_f + 2 * PW_BK * PW_LDA);

  // ---- XCD-aware tile numbering: hardware places block id on XCD id%8; give each XCD a contiguous
  // run of virtual ids so the nMt blocks that share one X tile hit the same L2 (bijective remap).
  const int id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  const int qn = total >> 3, rn = total & 7;
  const int v = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  const int mt = v % nMt;
  const int lt = (v / nMt) % nLt;
  const long b = v / (nMt * nLt);
  const int m0 = mt * PW_BM, l0 = lt * PW_BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  float mean = 0.f, rstd = 1.f, slope = 1.f;
  if (PRO == 1 || PRO == 2) srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
  if (PRO == 2 || PRO == 3) slope = a.nrm.prelu[0];

  const float* xb = a.x + (size_t)b * a.Cin * a.L;
  const int Cin = a.Cin, L = a.L, Cout = a.Cout;

  // staging assignment
  const int a_m = tid >> 2;        // 0..63 (+64 second pass)
  const int a_kq = (tid & 3) * 4;  // k offset of this thread's float4 inside the k-tile
  const int b_r = tid >> 5;        // 0..7 (+8 second pass)
  const int b_c = (tid & 31) * 4;  // column offset of this thread's float4

  float4 ra[2], rb[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      // unconditional load from a clamped row, masked afterwards (a "cond ? load : 0" select makes
      // hipcc branch around each load and drain vmcnt(0), serialising the prefetch)
      const int m = m0 + a_m + 64 * p;
      const float msk = (m < Cout) ? 1.f : 0.f;
      float4 wv = *reinterpret_cast<const float4*>(a.w + (size_t)(m < Cout ? m : 0) * Cin + k0 + a_kq);
      wv.x *= msk;
      wv.y *= msk;
      wv.z *= msk;
      wv.w *= msk;
      ra[p] = wv;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int k = k0 + b_r + 8 * p;
      const int l = l0 + b_c;
      float4 xv = *reinterpret_cast<const float4*>(xb + (size_t)k * L + (l < L ? l : 0));  // cols >= L never stored
      if (PRO == 1 || PRO == 2) {
        const float sc = a.nrm.gamma[k] * rstd;
        const float sh = a.nrm.beta[k] - mean * sc;
        xv.x = fmaf(xv.x, sc, sh);
        xv.y = fmaf(xv.y, sc, sh);
        xv.z = fmaf(xv.z, sc, sh);
        xv.w = fmaf(xv.w, sc, sh);
      }
      if (PRO == 2 || PRO == 3) {
        xv.x = srf_prelu(xv.x, slope);
        xv.y = srf_prelu(xv.y, slope);
        xv.z = srf_prelu(xv.z, slope);
        xv.w = srf_prelu(xv.w, slope);
      }
      rb[p] = xv;
    }
  };
  auto lds_store = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int m = a_m + 64 * p;
      As[buf][a_kq + 0][m] = ra[p].x;
      As[buf][a_kq + 1][m] = ra[p].y;
      As[buf][a_kq + 2][m] = ra[p].z;
      As[buf][a_kq + 3][m] = ra[p].w;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
      *reinterpret_cast<float4*>(&Bs[buf][b_r + 8 * p][b_c]) = rb[p];
  };

  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  const int nk = Cin / PW_BK;
  gload(0);
  lds_store(0);
  __syncthreads();
  const int kh = lane >> 5, col = lane & 31;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * PW_BK);
#pragma unroll
    for (int kk = 0; kk < PW_BK / 2; ++kk) {
      const float a0 = As[buf][2 * kk + kh][wm * 64 + col];
      const float a1 = As[buf][2 * kk + kh][wm * 64 + 32 + col];
      const float b0 = Bs[buf][2 * kk + kh][wn * 64 + col];
      const float b1 = Bs[buf][2 * kk + kh][wn * 64 + 32 + col];
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
    }
    if (kt + 1 < nk) lds_store(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue (shared with the bf16x3 kernel): per-wave LDS strips -> row-contiguous float4 I/O
  float s = 0.f, q = 0.f;
  const int mb = m0 + wm * 64, lb = l0 + wn * 64;
  float* strip = smem_f + wave * (32 * SRF_EPI_PITCH);   // tiles are dead after the last barrier
  srf_pw_epilogue_strip(a, acc00, acc01, strip, b, mb, lb, lane, s, q);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // strip reads done before it is rewritten
  __builtin_amdgcn_wave_barrier();
  srf_pw_epilogue_strip(a, acc10, acc11, strip, b, mb + 32, lb, lane, s, q);
  if (a.out_sums) srf_block_stats_atomic<4>((double)s, (double)q, srf_stat_slot(a.out_sums, b, v), red);
}

int srf_pw_bf16x3_launch(const PwArgs& a, int pro, hipStream_t st);
int srf_pw_w4_launch(const PwArgs& a, int pro, hipStream_t st);   // 64 x 64 tiles for small launches (srf_pwconv_w4.hip)
bool srf_pw_w4_wanted(const PwArgs& a);
int srf_pw_x3w_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st);   // the 256 x 128 kernel (srf_pwconv_x3w.hip)
bool srf_x3w_supported(int Bt, int pro);
bool srf_x3w_shape_supported(int Cin, int Cout, int L);
size_t srf_x3w_packed_bytes(int Cout, int Cin);
int srf_x3w_pack_launch(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n, hipStream_t st);
// Round 4: the paired-block form of the 256 x 128 kernel (srf_pwconv_x3p.hip: two co-resident blocks per CU; bit-identical
// outputs).  In isolation it ties with the one-block kernel on proj_1x1 / bottleneck and loses on res_conv (127 vs 113 us) and
// in the single-stream training step (+3.5 % / +4.5 % at cfg 2 / cfg 4); INSIDE srf_forward, whose caller runs two sub-batches
// on two streams, its half-CU blocks let the other stream's kernels co-reside (cfg 2: 6.74 -> 6.36-6.48 ms).  So srf_forward
// asks for it (srf_pw_prefer_paired, thread-local, scoped to the call) and every other caller -- srf_pw_conv_packed on its
// own, the backward's data-gradient GEMMs -- keeps the one-block kernel.  Debug flag 8192 swaps the two choices (A/B, tests).
// A packed weight buffer holds TWO images: [x3w image | x3p image].
int srf_pw_x3p_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st);
bool srf_x3p_supported(const PwArgs& a, int pro);
size_t srf_x3p_packed_bytes(int Cout, int Cin);
int srf_x3w_pack2_launch(const float* const* w, char* const* dst, char* const* dst16, const int* Cout, const int* Cin, int n,
                         hipStream_t st);
static thread_local int g_pw_prefer_paired = 0;
void srf_pw_prefer_paired(bool on) { g_pw_prefer_paired = on ? 1 : 0; }      // (srf_forward: around its launches)
static bool srf_pw_paired_wanted() { return g_pw_prefer_paired != 0; }
static int srf_pw_256_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st) {
  const char* wpack_p = wpack + srf_x3w_packed_bytes(a.Cout, a.Cin);
  const bool paired = srf_pw_paired_wanted() != ((srf_debug_flags() & 8192) != 0);
  if (paired && srf_x3p_supported(a, pro)) return srf_pw_x3p_launch(a, wpack_p, pro, st);
  return srf_pw_x3w_launch(a, wpack, pro, st);
}
int srf_pw_small_launch(const PwArgs& a, hipStream_t st);
bool srf_pw_small_supported(int Cin, int Cout, int L);

// Round 5: two 1x1 convolutions back to back in one launch, the intermediate handed over in registers (srf_pwconv_x3f.hip):
// res_conv (or the bottleneck) and the proj_1x1 that consumes its output.  Both weights come as packed buffers of
// srf_pack_pw_weights; the kernel streams their paired-block images.
bool srf_x3f_supported(int Bt, int K1, int C2, int L);
int srf_pw_x3f_launch(const PwPairArgs& a, int pro, hipStream_t st, bool f16 = false);
// Whether the fused pair serves (Bt, Cin1 -> 256 -> Cout2, L): the kernel's shape limits, the default kernel mode, and a launch
// that fills the chip (fewer 128-column tiles than CUs: the separate launches' small-launch kernels do better).
static bool srf_pw_pair_shape_ok(int Bt, int Cin1, int Cmid, int Cout2, int L) {      // (whatever the kernel mode)
  if (srf_debug_flags() & (1 | 4 | 8)) return false;
  if (Cmid != 256 || !srf_x3f_supported(Bt, Cin1, Cout2, L)) return false;
  if (!srf_x3w_shape_supported(Cin1, Cmid, L) || !srf_x3w_shape_supported(Cmid, Cout2, L)) return false;
  return (long)Bt * ((L + 127) / 128) >= srf_device_cus();
}
extern "C" int srf_pw_conv_pair_supported(int Bt, int Cin1, int Cmid, int Cout2, int L) {
  return srf_kernel_mode() == 0 && srf_pw_pair_shape_ok(Bt, Cin1, Cmid, Cout2, L) ? 1 : 0;
}
// y = W1 f(x) + bias1 (+ residual), f = in_norm (GlobLN, or GlobLN + PReLU: then the residual is required -- the two forms the
// model has -- or NULL: no prologue, residual required -- the backward's data-gradient pair); y2 = W2 y + bias2; out_sums2
// (nullable) += {sum, sumsq} of y2.  w1_packed / w2_packed: srf_pack_pw_weights.
extern "C" int srf_pw_conv_pair(const float* x, const void* w1_packed, const float* bias1, float* y, const srf_norm* in_norm,
                                const float* residual, const void* w2_packed, const float* bias2, float* y2, double* out_sums2,
                                int Bt, int Cin1, int Cmid, int Cout2, int L, void* stream) {
  SRF_CHECK_ARG(x && w1_packed && bias1 && y && w2_packed && bias2 && y2, "srf_pw_conv_pair: null pointer");
  SRF_CHECK_ARG(!in_norm || (in_norm->sums && in_norm->gamma && in_norm->beta), "srf_pw_conv_pair: a prologue needs statistics, gamma and beta");
  SRF_CHECK_ARG(srf_pw_conv_pair_supported(Bt, Cin1, Cmid, Cout2, L), "srf_pw_conv_pair: unsupported shape / mode (Bt=%d %d->%d->%d L=%d)",
                Bt, Cin1, Cmid, Cout2, L);
  SRF_CHECK_ARG(srf_aligned16(x) && srf_aligned16(y) && srf_aligned16(y2) && srf_aligned16(w1_packed) && srf_aligned16(w2_packed) &&
                    (!residual || srf_aligned16(residual)),
                "srf_pw_conv_pair: unaligned operand");
  PwPairArgs a;
  a.x = x;
  a.residual = residual;
  a.bias1 = bias1;
  a.y = y;
  a.nrm = srf_norm_dev(in_norm);
  a.inv_count = 1.0 / ((double)Cin1 * (double)L);
  a.wpack1 = reinterpret_cast<const char*>(w1_packed) + srf_x3w_packed_bytes(Cmid, Cin1);
  a.wpack2 = reinterpret_cast<const char*>(w2_packed) + srf_x3w_packed_bytes(Cout2, Cmid);
  a.bias2 = bias2;
  a.y2 = y2;
  a.out_sums2 = out_sums2;
  a.K1 = Cin1;
  a.C2 = Cout2;
  a.L = L;
  a.Bt = Bt;
  a.nLt = 0;
  a.total = 0;
  return srf_pw_x3f_launch(a, !in_norm ? 0 : (a.nrm.prelu ? 2 : 1), (hipStream_t)stream);
}

// THE predicate of the 256 x 128 dispatch for a whole launch (srf_pw_conv_packed, srf_pw_conv_packed3 and srf_pw_packed_only all
// use it -- ADVICE r3: the backward's "the fp32 weight is never read" shortcut was a hand-written copy of the dispatch test).
static bool srf_pw_256_serves(const void* w_packed, const float* x, int Bt, int Cin, int Cout, int L, int pro) {
  if (!w_packed || srf_kernel_mode() != 0 || (srf_debug_flags() & 4)) return false;
  if ((Cin % PW_BK) || (L % 4) || Cout < 32 || Cin < 32 || !srf_aligned16(x) || !srf_aligned16(w_packed)) return false;
  if (!srf_x3w_shape_supported(Cin, Cout, L) || !srf_x3w_supported(Bt, pro)) return false;
  if ((long)Bt * Cin * L * 4 >= (1L << 31)) return false;      // (beyond the 32-bit buffer range: the chunked form decides per chunk)
  return (long)Bt * ((Cout + 255) / 256) * ((L + 127) / 128) >= srf_device_cus();
}

// ---- K5 (library-internal; srf_forward's tail): mask GEMM + decoder contraction in one launch (srf_pwconv_x3w.hip, EPI 4)
int srf_pw_x3w_fused_tail_launch(const PwArgs& a, const char* wpack, const char* wdpack, float* zpart, int M, hipStream_t st);
size_t srf_x3w_dec_pack_bytes(int Ci);
int srf_x3w_pack_dec_launch(const float* w, void* dst, int Ci, int M, hipStream_t st);

// Whether srf_forward may run its tail fused: the shapes the round-3 256 x 128 kernel is dispatched for (same test as in
// srf_pw_conv_packed) and a decoder of at most 64 frame rows (sources x taps: 42 for the reference's 2 x 21).
// Debug flag 32768 = without (the masked tensor is then materialised: srf_debug_fetch(2), A/B).
bool srf_mask_decode_supported(int Bt, int Cin, int Cout, int L, int M) {
  if (srf_kernel_mode() != 0 || (srf_debug_flags() & (4 | 8 | 32768))) return false;
  if (M <= 0 || M > 64 || Cout % 8 || Cin % PW_BK || L % 4 || !srf_x3w_shape_supported(Cin, Cout, L) || !srf_x3w_supported(Bt, 3)) return false;
  if ((long)Bt * Cin * L * 4 >= (1L << 31)) return false;
  return (long)Bt * ((Cout + 255) / 256) * ((L + 127) / 128) >= srf_device_cus();
}
size_t srf_mask_decode_pack_bytes(int Cout) { return srf_x3w_dec_pack_bytes(Cout); }
int srf_mask_decode_pack(const float* wd, void* dst, int Ci, int M, hipStream_t st) {
  return srf_x3w_pack_dec_launch(wd, dst, Ci, M, st);
}
// zpart[Bt][ceil(Cout / 256)][M][L] = per-256-channel partial sums of Wd^T (relu(W prelu(x) + bias) * mul)
int srf_mask_decode(const float* x, const float* w, const void* w_packed, const float* bias, const float* prelu,
                    const float* mul, int mul_channels, const void* wd_packed, float* zpart, int Bt, int Cin, int Cout, int L,
                    int M, hipStream_t st) {
  SRF_CHECK_ARG(x && w && w_packed && bias && prelu && mul && wd_packed && zpart, "srf_mask_decode: null pointer");
  SRF_CHECK_ARG(srf_mask_decode_supported(Bt, Cin, Cout, L, M) && mul_channels > 0 && mul_channels % 8 == 0,
                "srf_mask_decode: unsupported shape");
  SRF_CHECK_ARG(srf_aligned16(x) && srf_aligned16(w_packed) && srf_aligned16(wd_packed), "srf_mask_decode: unaligned operand");
  PwArgs a;
  a.x = x;
  a.w = w;
  a.bias = bias;
  a.y = zpart;          // (never written by this form)
  a.residual = nullptr;
  a.out_sums = nullptr;
  a.mul = mul;
  srf_norm nm{nullptr, nullptr, nullptr, prelu};
  a.nrm = srf_norm_dev(&nm);
  a.inv_count = 1.0 / ((double)Cin * (double)L);
  a.Cin = Cin;
  a.Cout = Cout;
  a.L = L;
  a.Bt = Bt;
  a.mul_channels = mul_channels;
  a.epi_mask = 1;
  return srf_pw_x3w_fused_tail_launch(a, reinterpret_cast<const char*>(w_packed), reinterpret_cast<const char*>(wd_packed), zpart,
                                      M, st);
}

// ---- three-part split GEMM (six bf16 MFMAs per product block: the exact-fp32 class at ~1.6 x the 3-MFMA kernel's time instead
// of the exact-fp32 MFMA kernel's 2.5 x) -- the training forward's 1x1 convolutions (srf_forward_train) ----------------------
int srf_pw_x3w3_launch(const PwArgs& a, const char* wpack3, int pro, hipStream_t st);
int srf_pw_x3w4_launch(const PwArgs& a, const char* wpack, int pro, hipStream_t st);
int srf_x3w_pack_f16_launch(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n, hipStream_t st,
                            char* const* dst16);
// The training forward's GEMMs run on TWO fp16 parts per operand (hi = fp16(x), lo = fp16(x - hi): 22 mantissa bits, 3 MFMAs per
// product block -- round 4) unless debug flag 16384 selects round 3's three bf16 parts (24 bits, 6 MFMAs).
// Measured (tools/f16_split_probe.py -> profiles/r04_f16_split_probe.txt): error against fp64 on a model-sized GEMM 2.6e-6 --
// below the three-part kernel's 4.4e-6 and the exact-fp32 MFMA kernel's 4.8e-6 (all three sit on the fp32 accumulation floor);
// for inputs of magnitude 0.02, where the fp16 lo parts go subnormal, 1.0e-7 against 7.7e-8; 61 us per launch against 89 us;
// every gradient and trajectory fixture passes at unchanged bars; cfg 2 step 35.2 -> 33.1 ms, cfg 4 111.2 -> 102.0 ms.
// Range: fp16 parts end at 65504 where bf16's do not.  GlobLN'ed tensors are O(1), the residual stream O(1..100) in every model
// we have; an operand beyond the range (or NaN / inf) makes its output column non-finite -- loud, like the fp32 reference's own
// overflow, not clamped (round 5) -- and flag 16384 (three bf16 parts) is the form without the limit.
// The packed3 buffer then holds the fp16 image (half its size).
static bool srf_train_f16_split() { return (srf_debug_flags() & 16384) == 0; }
// The two image formats differ in layout and size, and which one a buffer holds is decided at PACK time: every
// buffer srf_pack3_pw_weights has written is remembered with its format (device, address), and a launch whose flag disagrees
// with the buffer it is handed is refused with SRF_EINVAL instead of reinterpreting the bits.  (A buffer this process never
// packed -- e.g. a device-to-device copy of one -- is not in the table and is taken as what the caller says it is.)
static std::mutex g_pk3_mu;
struct Pk3Entry {
  int format;       // 4 = fp16 x 2, 3 = bf16 x 3
  size_t bytes;     // the image's extent: a later pack that overlaps it replaces it
};
static std::map<unsigned long long, Pk3Entry> g_pk3_format;          // key: device << 56 | address (ordered: range queries)
static unsigned long long srf_pk3_key(const void* p) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return ((unsigned long long)dev << 56) | ((unsigned long long)(size_t)p & ((1ull << 56) - 1));
}
// Round 6: entries no longer live for ever.  A pack drops every recorded image its own extent overlaps (an
// allocator that re-uses a freed address hands it to the next pack or to something else; in the first case the stale record
// is replaced here), and srf_pack3_forget drops the record of a buffer the caller is about to free or re-use for other data.
static void srf_pk3_record(const void* p, size_t bytes, int format) {     // (g_pk3_mu held)
  const unsigned long long k0 = srf_pk3_key(p), k1 = k0 + bytes;
  auto it = g_pk3_format.lower_bound(k0);
  if (it != g_pk3_format.begin()) {
    auto prev = std::prev(it);
    if (prev->first + prev->second.bytes > k0) it = prev;
  }
  while (it != g_pk3_format.end() && it->first < k1) it = g_pk3_format.erase(it);
  g_pk3_format[k0] = Pk3Entry{format, bytes};
}
extern "C" void srf_pack3_forget(const void* packed) {
  if (!packed) return;
  std::lock_guard<std::mutex> lk(g_pk3_mu);
  g_pk3_format.erase(srf_pk3_key(packed));
}
size_t srf_x3w_packed3_bytes(int Cout, int Cin);
int srf_x3w_pack3_launch(const float* const* w, char* const* dst, const int* Cout, const int* Cin, int n, hipStream_t st);

extern "C" size_t srf_packed3_pw_weight_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || !srf_x3w_shape_supported(Cin, Cout, 4)) return 0;
  return srf_x3w_packed3_bytes(Cout, Cin);
}
extern "C" int srf_pack3_pw_weights(const float* const* w, void* const* packed, const int* Cout, const int* Cin, int n,
                                    void* stream) {
  SRF_CHECK_ARG(w && packed && Cout && Cin && n > 0, "srf_pack3_pw_weights: bad arguments");
  for (int i = 0; i < n; ++i)
    SRF_CHECK_ARG(w[i] && packed[i] && srf_packed3_pw_weight_bytes(Cout[i], Cin[i]) > 0 && srf_aligned16(packed[i]),
                  "srf_pack3_pw_weights: entry %d unsupported (Cout=%d Cin=%d)", i, Cout[i], Cin[i]);
  const bool f16 = srf_train_f16_split();
  {
    std::lock_guard<std::mutex> lk(g_pk3_mu);
    for (int i = 0; i < n; ++i) srf_pk3_record(packed[i], srf_packed3_pw_weight_bytes(Cout[i], Cin[i]), f16 ? 4 : 3);
  }
  if (f16) {
    // the fp16 image takes the first half of a packed3 buffer (sized for the three-part image); the second half gets the same
    // parts in the paired-block layout, for the training forward's fused pairs (srf_pw_conv_pair_packed3)
    std::vector<char*> second(n);
    for (int i = 0; i < n; ++i) second[i] = reinterpret_cast<char*>(packed[i]) + srf_x3w_packed_bytes(Cout[i], Cin[i]);
    return srf_x3w_pack_f16_launch(w, reinterpret_cast<char* const*>(packed), Cout, Cin, n, (hipStream_t)stream, second.data());
  }
  return srf_x3w_pack3_launch(w, reinterpret_cast<char* const*>(packed), Cout, Cin, n, (hipStream_t)stream);
}
// y = W f(x) + bias (+ residual), out_sums as in srf_pw_conv; w_packed3 from srf_pack3_pw_weights (NULL, a shape the
// 256 x 128 kernel does not take, or a launch of fewer tiles than CUs: srf_pw_conv under the caller's kernel mode).
extern "C" int srf_pw_conv_packed3(const float* x, const float* w, const void* w_packed3, const float* bias, float* y, int Bt,
                                   int Cin, int Cout, int L, const srf_norm* in_norm, const float* residual, double* out_sums,
                                   void* stream) {
  SRF_CHECK_ARG(x && w && bias && y, "srf_pw_conv: null pointer");
  SRF_CHECK_ARG(Bt > 0 && Cin > 0 && Cout > 0 && L > 0, "srf_pw_conv: bad sizes");
  const SrfNormDev nd = srf_norm_dev(in_norm);
  const int pro = nd.sums ? (nd.prelu ? 2 : 1) : (nd.prelu ? 3 : 0);
  const bool form_ok = (pro == 2) == (residual != nullptr);      // the built forms: res_conv has the residual, the others none
  // (kernel mode 2 -- the exact-fp32 MFMA kernels -- still takes this three-part split GEMM: it IS the exact-fp32 class)
  const bool ok = form_ok && srf_kernel_mode() != 1 && srf_aligned16(y) && (!residual || srf_aligned16(residual)) &&
                  (srf_kernel_mode() == 0 ? srf_pw_256_serves(w_packed3, x, Bt, Cin, Cout, L, pro)
                                          : (w_packed3 && srf_x3w_shape_supported(Cin, Cout, L) && srf_x3w_supported(Bt, pro) &&
                                             srf_aligned16(x) && srf_aligned16(w_packed3) && (long)Bt * Cin * L * 4 < (1L << 31) &&
                                             !(srf_debug_flags() & 4) &&
                                             (long)Bt * ((Cout + 255) / 256) * ((L + 127) / 128) >= srf_device_cus()));
  if (!ok) return srf_pw_conv(x, w, bias, y, Bt, Cin, Cout, L, in_norm, residual, out_sums, 0, nullptr, 0, stream);
  if (nd.sums) SRF_CHECK_ARG(nd.gamma && nd.beta, "srf_pw_conv: norm without gamma/beta");
  PwArgs a;
  a.x = x;
  a.w = w;
  a.bias = bias;
  a.y = y;
  a.residual = residual;
  a.out_sums = out_sums;
  a.mul = nullptr;
  a.nrm = nd;
  a.inv_count = 1.0 / ((double)Cin * (double)L);
  a.Cin = Cin;
  a.Cout = Cout;
  a.L = L;
  a.Bt = Bt;
  a.mul_channels = 1;
  a.epi_mask = 0;
  const bool f16 = srf_train_f16_split();
  {
    std::lock_guard<std::mutex> lk(g_pk3_mu);
    const auto it = g_pk3_format.find(srf_pk3_key(w_packed3));
    SRF_CHECK_ARG(it == g_pk3_format.end() || it->second.format == (f16 ? 4 : 3),
                  "srf_pw_conv_packed3: the weight image was packed as %s but debug flag 16384 now asks for %s -- pack again",
                  it->second.format == 4 ? "two fp16 parts" : "three bf16 parts", f16 ? "two fp16 parts" : "three bf16 parts");
  }
  if (f16) return srf_pw_x3w4_launch(a, reinterpret_cast<const char*>(w_packed3), pro, (hipStream_t)stream);
  return srf_pw_x3w3_launch(a, reinterpret_cast<const char*>(w_packed3), pro, (hipStream_t)stream);
}

// The training forward's fused pair (round 5): srf_pw_conv_pair on the two-fp16-part images of srf_pack3_pw_weights -- y and y2
// bit-identical to two srf_pw_conv_packed3 launches (res_conv / bottleneck, then proj_1x1 with its statistics).  Served: what
// srf_pw_conv_pair serves, with the fp16 form selected (no debug flag 16384) and both buffers packed as fp16 by this process.
static bool srf_pk3_is_f16(const void* p) {
  std::lock_guard<std::mutex> lk(g_pk3_mu);
  const auto it = g_pk3_format.find(srf_pk3_key(p));
  return it != g_pk3_format.end() && it->second.format == 4;
}
extern "C" int srf_pw_conv_pair_packed3_supported(int Bt, int Cin1, int Cmid, int Cout2, int L) {
  // (kernel mode 2 -- srf_forward_train's -- as well as 0: like srf_pw_conv_packed3, this IS the exact-fp32 class)
  return srf_train_f16_split() && srf_kernel_mode() != 1 && srf_pw_pair_shape_ok(Bt, Cin1, Cmid, Cout2, L) &&
                 srf_x3w_packed3_bytes(Cmid, Cin1) >= 2 * srf_x3w_packed_bytes(Cmid, Cin1) &&
                 srf_x3w_packed3_bytes(Cout2, Cmid) >= 2 * srf_x3w_packed_bytes(Cout2, Cmid)
             ? 1 : 0;
}
extern "C" int srf_pw_conv_pair_packed3(const float* x, const void* w1_packed3, const float* bias1, float* y, const srf_norm* in_norm,
                                        const float* residual, const void* w2_packed3, const float* bias2, float* y2,
                                        double* out_sums2, int Bt, int Cin1, int Cmid, int Cout2, int L, void* stream) {
  SRF_CHECK_ARG(x && w1_packed3 && bias1 && y && in_norm && w2_packed3 && bias2 && y2, "srf_pw_conv_pair_packed3: null pointer");
  SRF_CHECK_ARG(in_norm->sums && in_norm->gamma && in_norm->beta, "srf_pw_conv_pair_packed3: conv 1 needs a GlobLN prologue");
  SRF_CHECK_ARG(srf_pw_conv_pair_packed3_supported(Bt, Cin1, Cmid, Cout2, L),
                "srf_pw_conv_pair_packed3: unsupported shape / mode (Bt=%d %d->%d->%d L=%d)", Bt, Cin1, Cmid, Cout2, L);
  SRF_CHECK_ARG(srf_pk3_is_f16(w1_packed3) && srf_pk3_is_f16(w2_packed3),
                "srf_pw_conv_pair_packed3: both weight images must have been packed as two fp16 parts by srf_pack3_pw_weights");
  SRF_CHECK_ARG(srf_aligned16(x) && srf_aligned16(y) && srf_aligned16(y2) && srf_aligned16(w1_packed3) && srf_aligned16(w2_packed3) &&
                    (!residual || srf_aligned16(residual)),
                "srf_pw_conv_pair_packed3: unaligned operand");
  PwPairArgs a;
  a.x = x;
  a.residual = residual;
  a.bias1 = bias1;
  a.y = y;
  a.nrm = srf_norm_dev(in_norm);
  a.inv_count = 1.0 / ((double)Cin1 * (double)L);
  a.wpack1 = reinterpret_cast<const char*>(w1_packed3) + srf_x3w_packed_bytes(Cmid, Cin1);
  a.wpack2 = reinterpret_cast<const char*>(w2_packed3) + srf_x3w_packed_bytes(Cout2, Cmid);
  a.bias2 = bias2;
  a.y2 = y2;
  a.out_sums2 = out_sums2;
  a.K1 = Cin1;
  a.C2 = Cout2;
  a.L = L;
  a.Bt = Bt;
  a.nLt = 0;
  a.total = 0;
  return srf_pw_x3f_launch(a, a.nrm.prelu ? 2 : 1, (hipStream_t)stream, true);
}

// ---- GroupComm (library-internal; srf_forward): proj_1x1 with the TAC's "x + GlobLN(q)" folded into its operand load ----
// y = W u + bias, u = x + GlobLN_qnorm(q) written to `u` as well (the block's residual); out_sums as in srf_pw_conv.
// Only the register-resident thin-shape kernel (srf_pwconv_small.hip) has this prologue.
bool srf_pw_conv_preadd_supported(int Cin, int Cout, int L, const void* const* ptrs, int nptrs) {
  if (srf_kernel_mode() == 1 || !srf_pw_small_supported(Cin, Cout, L)) return false;
  for (int i = 0; i < nptrs; ++i)
    if (!srf_aligned16(ptrs[i])) return false;
  return true;
}
int srf_pw_conv_preadd(const float* x, const float* q, const srf_norm* qnorm, float* u, const float* w, const float* bias,
                       float* y, int Bt, int Cin, int Cout, int L, double* out_sums, hipStream_t st) {
  SRF_CHECK_ARG(x && q && qnorm && qnorm->sums && qnorm->gamma && qnorm->beta && u && w && bias && y, "srf_pw_conv_preadd: null pointer");
  const void* ptrs[4] = {x, q, u, y};
  SRF_CHECK_ARG(srf_pw_conv_preadd_supported(Cin, Cout, L, ptrs, 4), "srf_pw_conv_preadd: unsupported shape");
  PwArgs a;
  a.x = x;
  a.w = w;
  a.bias = bias;
  a.y = y;
  a.residual = nullptr;
  a.out_sums = out_sums;
  a.mul = nullptr;
  a.nrm = srf_norm_dev(nullptr);
  a.inv_count = 1.0 / ((double)Cin * (double)L);
  a.Cin = Cin;
  a.Cout = Cout;
  a.L = L;
  a.Bt = Bt;
  a.mul_channels = 1;
  a.epi_mask = 0;
  a.pre_q = q;
  a.pre_u = u;
  a.pre_nrm = srf_norm_dev(qnorm);
  a.pre_inv_count = 1.0 / ((double)Cin * (double)L);
  return srf_pw_small_launch(a, st);
}

extern "C" size_t srf_packed_pw_weight_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || !srf_x3w_shape_supported(Cin, Cout, 4)) return 0;
  return srf_x3w_packed_bytes(Cout, Cin) + srf_x3p_packed_bytes(Cout, Cin);
}
static int srf_pack_both(const float* const* w, void* const* packed, const int* Cout, const int* Cin_signed, int n, hipStream_t st) {
  std::vector<char*> second(n);     // the x3p image of every entry behind its x3w image (one launch writes both)
  for (int i = 0; i < n; ++i)
    second[i] = reinterpret_cast<char*>(packed[i]) + srf_x3w_packed_bytes(Cout[i], Cin_signed[i] < 0 ? -Cin_signed[i] : Cin_signed[i]);
  int rc = srf_x3w_pack2_launch(w, reinterpret_cast<char* const*>(packed), second.data(), Cout, Cin_signed, n, st);
  return rc;
}

// Whether srf_pw_conv_packed will serve this launch (no prologue: the backward's data-gradient GEMMs) from the packed image
// alone -- the 256 x 128 kernel on the whole launch -- so that the fp32 weight argument is never read (the backward skips its
// transposed copy).
bool srf_pw_packed_only(const void* w_packed, const float* x, int Bt, int Cin, int Cout, int L) {
  if (srf_pw_small_supported(Cin, Cout, L)) return false;     // (the thin-shape kernel has precedence and reads the fp32 weights)
  return srf_pw_256_serves(w_packed, x, Bt, Cin, Cout, L, 0);
}

// (library-internal: the backward's data-gradient GEMMs) w[i] is the FORWARD weight [Cin][Cout]; the image is that of its
// transpose [Cout][Cin] -- no transposed copy is made
int srf_pack_pw_weights_transposed(const float* const* w, void* const* packed, const int* Cout, const int* Cin, int n,
                                   hipStream_t st) {
  std::vector<int> neg(Cin, Cin + n);
  for (int i = 0; i < n; ++i) {
    SRF_CHECK_ARG(w[i] && packed[i] && srf_packed_pw_weight_bytes(Cout[i], Cin[i]) > 0 && srf_aligned16(packed[i]),
                  "srf_pack_pw_weights: entry %d unsupported (Cout=%d Cin=%d)", i, Cout[i], Cin[i]);
    neg[i] = -Cin[i];
  }
  return srf_pack_both(w, packed, Cout, neg.data(), n, st);
}

extern "C" int srf_pack_pw_weights(const float* const* w, void* const* packed, const int* Cout, const int* Cin,
                                   int n, void* stream) {
  SRF_CHECK_ARG(w && packed && Cout && Cin && n > 0, "srf_pack_pw_weights: bad arguments");
  for (int i = 0; i < n; ++i)
    SRF_CHECK_ARG(w[i] && packed[i] && srf_packed_pw_weight_bytes(Cout[i], Cin[i]) > 0 &&
                      srf_aligned16(packed[i]),
                  "srf_pack_pw_weights: entry %d unsupported (Cout=%d Cin=%d)", i, Cout[i], Cin[i]);
  return srf_pack_both(w, packed, Cout, Cin, n, (hipStream_t)stream);
}

extern "C" int srf_pw_conv(const float* x, const float* w, const float* bias, float* y, int Bt, int Cin,
                           int Cout, int L, const srf_norm* in_norm, const float* residual,
                           double* out_sums, int epilogue_mask, const float* mul, int mul_channels,
                           void* stream) {
  return srf_pw_conv_packed(x, w, nullptr, bias, y, Bt, Cin, Cout, L, in_norm, residual, out_sums,
                            epilogue_mask, mul, mul_channels, stream);
}

extern "C" int srf_pw_conv_packed(const float* x, const float* w, const void* w_packed, const float* bias,
                                  float* y, int Bt, int Cin, int Cout, int L, const srf_norm* in_norm,
                                  const float* residual, double* out_sums, int epilogue_mask, const float* mul,
                                  int mul_channels, void* stream) {
  SRF_CHECK_ARG(x && w && bias && y, "srf_pw_conv: null pointer");
  SRF_CHECK_ARG(Bt > 0 && Cin > 0 && Cout > 0 && L > 0, "srf_pw_conv: bad sizes");
  SRF_CHECK_ARG(!epilogue_mask || (mul && mul_channels > 0), "srf_pw_conv: mask epilogue needs mul");
  PwArgs a;
  a.x = x;
  a.w = w;
  a.bias = bias;
  a.y = y;
  a.residual = residual;
  a.out_sums = out_sums;
  a.mul = mul;
  a.nrm = srf_norm_dev(in_norm);
  a.inv_count = 1.0 / ((double)Cin * (double)L);
  a.Cin = Cin;
  a.Cout = Cout;
  a.L = L;
  a.Bt = Bt;
  a.mul_channels = mul_channels > 0 ? mul_channels : 1;
  a.epi_mask = epilogue_mask ? 1 : 0;
  if (a.nrm.sums) SRF_CHECK_ARG(a.nrm.gamma && a.nrm.beta, "srf_pw_conv: norm without gamma/beta");
  hipStream_t st = (hipStream_t)stream;

  const int mode = srf_kernel_mode();
  const bool mfma_ok = mode != 1 && (Cin % PW_BK == 0) && (L % 4 == 0) && Cout >= 32 && Cin >= 32 &&
                       srf_aligned16(x) && srf_aligned16(w);
  const int pro_sel = a.nrm.sums ? (a.nrm.prelu ? 2 : 1) : (a.nrm.prelu ? 3 : 0);
  // GroupComm's per-group convs (<= 32 -> <= 64 channels): register-resident streaming kernel
  if (mode != 1 && !(a.epi_mask & 1) && srf_pw_small_supported(Cin, Cout, L) && srf_aligned16(x) &&
      srf_aligned16(y) && (!residual || srf_aligned16(residual)))
    return srf_pw_small_launch(a, st);
  // 256 x 128 tiles with pre-split weights: whenever the packed image is there and the launch fills the chip (fewer tiles
  // than CUs: the 128 x 128 kernels below make twice as many).  (That kernel never reads the fp32 weights.)
  if (srf_pw_256_serves(w_packed, x, Bt, Cin, Cout, L, pro_sel))
    return srf_pw_256_launch(a, reinterpret_cast<const char*>(w_packed), pro_sel, st);
  // An activation tensor beyond the 2 GB reach of the kernel's 32-bit buffer offsets (cfg 5's bottleneck: 16 x 4096 x 12800
  // floats = 3.4 GB) goes out as several launches over runs of whole examples: examples are independent, every per-example
  // pointer (statistics slots included) just moves along.
  if (mfma_ok && mode == 0 && w_packed && srf_x3w_shape_supported(Cin, Cout, L) && srf_x3w_supported(Bt, pro_sel) && srf_aligned16(w_packed) &&
      (long)Bt * Cin * L * 4 >= (1L << 31) && (long)Cin * L * 4 < (1L << 31) && !(srf_debug_flags() & 4)) {
    const int cap = (int)(((1L << 31) - 1) / ((long)Cin * L * 4));   // examples one launch can address
    const int nch = (Bt + cap - 1) / cap, per = (Bt + nch - 1) / nch;  // balanced runs of whole examples
    const int last = Bt - (nch - 1) * per;
    if ((long)last * ((Cout + 255) / 256) * ((L + 127) / 128) >= srf_device_cus()) {
      for (int b0 = 0; b0 < Bt; b0 += per) {
        PwArgs c = a;
        c.Bt = Bt - b0 < per ? Bt - b0 : per;
        c.x = a.x + (size_t)b0 * Cin * L;
        c.y = a.y + (size_t)b0 * Cout * L;
        if (a.residual) c.residual = a.residual + (size_t)b0 * Cout * L;
        if (a.mul) c.mul = a.mul + (size_t)b0 * a.mul_channels * L;
        if (a.out_sums) c.out_sums = a.out_sums + (size_t)b0 * SRF_STAT_BUCKETS * 2;
        if (a.nrm.sums) c.nrm.sums = a.nrm.sums + (size_t)b0 * SRF_STAT_BUCKETS * 2;
        const bool more = b0 + per < Bt;    // (one profiler interval for the whole operation: only the last launch is marked)
        if (more) srf_prof_hold(+1);
        const int rc = srf_pw_256_launch(c, reinterpret_cast<const char*>(w_packed), pro_sel, st);
        if (more) srf_prof_hold(-1);
        if (rc) return rc;
      }
      return SRF_OK;
    }
  }
  // launches that would leave CUs idle on 128 x 128 tiles (a batch-1 forward): 64 x 64 tiles, four times the blocks
  if (mfma_ok && mode == 0 && (Cin % 64 == 0) && srf_aligned16(y) && srf_pw_w4_wanted(a)) return srf_pw_w4_launch(a, pro_sel, st);
  if (mfma_ok && mode == 0 && (Cin % 64 == 0)) return srf_pw_bf16x3_launch(a, pro_sel, st);
  if (mfma_ok) {
    const int nMt = (Cout + PW_BM - 1) / PW_BM, nLt = (L + PW_BN - 1) / PW_BN;
    const long total = (long)Bt * nMt * nLt;
    SRF_CHECK_ARG(total < (1L << 31), "srf_pw_conv: too many tiles");
    const int pro = a.nrm.sums ? (a.nrm.prelu ? 2 : 1) : (a.nrm.prelu ? 3 : 0);
    dim3 grid((unsigned)total), block(256);
    switch (pro) {
      case 0: hipLaunchKernelGGL(srf_pw_mfma_kernel<0>, grid, block, 0, st, a, nMt, nLt, (int)total); break;
      case 1: hipLaunchKernelGGL(srf_pw_mfma_kernel<1>, grid, block, 0, st, a, nMt, nLt, (int)total); break;
      case 2: hipLaunchKernelGGL(srf_pw_mfma_kernel<2>, grid, block, 0, st, a, nMt, nLt, (int)total); break;
      default: hipLaunchKernelGGL(srf_pw_mfma_kernel<3>, grid, block, 0, st, a, nMt, nLt, (int)total); break;
    }
  } else {
    SRF_CHECK_ARG(Bt <= 65535, "srf_pw_conv: batch too large for the generic kernel");
    constexpr int MT = 16;
    dim3 grid((L + 255) / 256, (Cout + MT - 1) / MT, Bt);
    hipLaunchKernelGGL(srf_pw_generic_kernel<MT>, grid, dim3(256), 0, st, a);
  }
  SRF_CHECK_LAUNCH(mfma_ok ? "pw_conv_mfma" : "pw_conv_generic", st);
  return SRF_OK;
}
