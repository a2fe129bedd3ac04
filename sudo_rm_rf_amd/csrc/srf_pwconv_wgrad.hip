// Weight gradient of the pointwise convs (training step):
//     dW[m, n] = sum_b sum_l G[b, m, l] * f(X[b, n, l])        dbias[m] = sum_b sum_l G[b, m, l]
// for y = W f(x) + bias (improved_sudormrf.py:174,196,256-259,268-269), f = the same GlobLN / PReLU prologue the
// forward GEMM applies on load (it is never materialised, so it is re-applied here).
// A GEMM with a tiny output (Cout x Cin) and a huge reduction (Bt * L = 102400 for cfg 2): both operands are
// [row][time] with the reduction index contiguous, which is exactly what v_mfma_f32_32x32x16_bf16 wants (8
// consecutive k per lane) -- no transposition while staging, unlike the forward.  Same split-bf16 scheme as the
// forward (hi + lo, three MFMAs per product, fp32 accumulate), same 128x128x32 tile / 8 wavefronts / LDS images.
// Split-K: the (example, time-chunk) pieces are dealt round-robin to P blocks per output tile, each block
// keeps its 128x128 accumulator over all its pieces and writes ONE partial tile; a second kernel sums the P
// partials (deterministic, no atomics).
#include "srf_pw.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_BM = 128, WG_BN = 128, WG_BK = 32;
constexpr int WG_PITCH = 80;                  // bytes per LDS row (32 bf16 + 16 B pad): conflict-free b128 reads
constexpr int WG_IMG = WG_BM * WG_PITCH;
constexpr int WG_STAGE = 4 * WG_IMG;          // G_hi, G_lo, X_hi, X_lo

struct WgArgs {
  const float* g;      // [Bt][M][L]
  const float* x;      // [Bt][N][L]
  float* part;         // [P][M][N]
  float* bias_part;    // [P][M] or null
  SrfNormDev nrm;      // prologue on x (per example statistics, per channel n affine)
  double inv_count;
  int M, N, L, Bt;
  int kc_len, nKc, P;  // time-chunk length (multiple of 32), chunks per example, partials
  int nMt, nNt;
};

__device__ __forceinline__ void wg_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

// PRO: 0 identity, 1 GlobLN, 2 GlobLN + PReLU, 3 PReLU
template <int PRO>
__global__ __launch_bounds__(512, 2) void srf_pw_wgrad_kernel(WgArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * WG_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = a.nMt * a.nNt;
  const int tile = blockIdx.x % ntiles, p = blockIdx.x / ntiles;
  const int m0 = (tile % a.nMt) * WG_BM, n0 = (tile / a.nMt) * WG_BN;
  const int L = a.L, M = a.M, N = a.N;
  const float slope = (PRO == 2 || PRO == 3) ? a.nrm.prelu[0] : 1.f;

  // staging: thread -> row r = tid>>2 of BOTH tiles, 8-k packet pk = tid&3 (2 float4 each)
  const int r = tid >> 2, pk = tid & 3;
  const bool g_ok = (m0 + r) < M, x_ok = (n0 + r) < N;
  const int gm = g_ok ? m0 + r : 0, xn = x_ok ? n0 + r : 0;
  const float g_msk = g_ok ? 1.f : 0.f, x_msk = x_ok ? 1.f : 0.f;
  const int lds_off = r * WG_PITCH + pk * 16;
  float gam = 1.f, bet = 0.f;
  if (PRO == 1 || PRO == 2) {
    gam = a.nrm.gamma[xn];
    bet = a.nrm.beta[xn];
  }

  f32x16 acc0 = {0}, acc1 = {0};
  float bsum = 0.f;
  const int frag = (lane & 31) * WG_PITCH + (lane >> 5) * 16;
  const int a_row = (wm * 32) * WG_PITCH + frag;
  const int b_row0 = (wn * 64) * WG_PITCH + frag, b_row1 = b_row0 + 32 * WG_PITCH;

  const int nchunks = a.Bt * a.nKc;
  for (int c = p; c < nchunks; c += a.P) {
    const int b = c / a.nKc, kc = c - b * a.nKc;
    const int l_beg = kc * a.kc_len;
    const int l_end = min(l_beg + a.kc_len, L);
    const int nk = (l_end - l_beg + WG_BK - 1) / WG_BK;
    float sc = 1.f, sh = 0.f;
    if (PRO == 1 || PRO == 2) {
      float mean, rstd;
      srf_finalize_stats(a.nrm.sums, b, a.inv_count, mean, rstd);
      sc = gam * rstd;
      sh = bet - mean * sc;
    }
    const float* gsrc = a.g + ((size_t)b * M + gm) * L;
    const float* xsrc = a.x + ((size_t)b * N + xn) * L;

    float4 rg[2], rx[2];
    float km[2];   // 1 inside the chunk, 0 beyond its end (L % 4 == 0: a float4 is entirely in or out)
    auto gload = [&](int kt) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int l = l_beg + kt * WG_BK + pk * 8 + h * 4;
        const bool in = l < l_end;
        const int lc = in ? l : l_beg;
        rg[h] = *reinterpret_cast<const float4*>(gsrc + lc);
        rx[h] = *reinterpret_cast<const float4*>(xsrc + lc);
        km[h] = in ? 1.f : 0.f;
      }
    };
    auto lds_store = [&](int stage) {
      char* base = smem + stage * WG_STAGE;
      float vg[8], vx[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float mg = km[h] * g_msk, mx = km[h] * x_msk;
        const float gg[4] = {rg[h].x, rg[h].y, rg[h].z, rg[h].w};
        const float xx[4] = {rx[h].x, rx[h].y, rx[h].z, rx[h].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vg[4 * h + j] = gg[j] * mg;
          float v = xx[j];
          if (PRO == 1 || PRO == 2) v = fmaf(v, sc, sh);
          if (PRO == 2 || PRO == 3) v = srf_prelu(v, slope);
          vx[4 * h + j] = v * mx;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum += vg[j];
      bf16x8 hi, lo;
      wg_split8(vg, hi, lo);
      *reinterpret_cast<bf16x8*>(base + 0 * WG_IMG + lds_off) = hi;
      *reinterpret_cast<bf16x8*>(base + 1 * WG_IMG + lds_off) = lo;
      wg_split8(vx, hi, lo);
      *reinterpret_cast<bf16x8*>(base + 2 * WG_IMG + lds_off) = hi;
      *reinterpret_cast<bf16x8*>(base + 3 * WG_IMG + lds_off) = lo;
    };
    auto mma_tile = [&](int stage) {
      const char* base = smem + stage * WG_STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ko = ks * 32;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + 0 * WG_IMG + a_row + ko);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + 1 * WG_IMG + a_row + ko);
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(base + 2 * WG_IMG + b_row0 + ko);
        const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(base + 2 * WG_IMG + b_row1 + ko);
        const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(base + 3 * WG_IMG + b_row0 + ko);
        const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(base + 3 * WG_IMG + b_row1 + ko);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc1, 0, 0, 0);
      }
    };

    // one register set, LDS double-buffered: tile kt+1 is split into the other stage and tile kt+2's loads
    // are in flight while tile kt's MFMAs run
    gload(0);
    __syncthreads();          // previous chunk's last MFMAs are done with both stages
    lds_store(0);
    if (nk > 1) gload(1);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) lds_store((kt + 1) & 1);
      if (kt + 2 < nk) gload(kt + 2);
      mma_tile(kt & 1);
      __syncthreads();
    }
  }

  // ---- partial tile out: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = a.part + (size_t)p * M * N;
  const int col = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int row = (rr & 3) + 8 * (rr >> 2) + 4 * kh;
    const int m = m0 + wm * 32 + row;
    const int n = n0 + wn * 64 + col;
    if (m < M) {
      if (n < N) out[(size_t)m * N + n] = acc0[rr];
      if (n + 32 < N) out[(size_t)m * N + n + 32] = acc1[rr];
    }
  }
  if (a.bias_part && (tile / a.nMt) == 0) {
    // the 4 threads of a row hold disjoint k packets
    bsum += __shfl_xor(bsum, 1, 64);
    bsum += __shfl_xor(bsum, 2, 64);
    if (pk == 0 && g_ok) a.bias_part[(size_t)p * M + m0 + r] = bsum;
  }
}

// out[m][n] (n < cols_out) = sum_p part[p][m][n]   (part rows have `cols` entries)
__global__ __launch_bounds__(256) void srf_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               int rows, int cols, int cols_out, int ld_out, int P,
                                                               float beta) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * cols_out) return;
  const int m = (int)(i / cols_out), n = (int)(i - (long)m * cols_out);
  const size_t src = (size_t)m * cols + n, stride = (size_t)rows * cols, dst = (size_t)m * ld_out + n;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += part[(size_t)p * stride + src];
  out[dst] = beta != 0.f ? fmaf(beta, out[dst], s) : s;
}

static int wg_pick_partials(int ntiles, int nchunks) {
  int P = (2 * 256 + ntiles - 1) / ntiles;   // one resident wave of blocks (2 per CU): fewer, longer partials
  if (P > nchunks) P = nchunks;
  if (P > 128) P = 128;
  if (P < 1) P = 1;
  return P;
}

static void wg_geometry(int M, int N, int L, int Bt, WgArgs* a) {
  a->nMt = (M + WG_BM - 1) / WG_BM;
  a->nNt = (N + WG_BN - 1) / WG_BN;
  a->kc_len = 800;                              // 25 k-tiles per piece
  if (a->kc_len > L) a->kc_len = (L + 31) / 32 * 32;
  a->nKc = (L + a->kc_len - 1) / a->kc_len;
  a->P = wg_pick_partials(a->nMt * a->nNt, Bt * a->nKc);
}

extern "C" size_t srf_pw_wgrad_scratch_bytes(int Bt, int Cout, int Cin, int L) {
  if (Bt <= 0 || Cout <= 0 || Cin <= 0 || L <= 0) return 0;
  WgArgs a;
  wg_geometry(Cout, Cin, L, Bt, &a);
  return sizeof(float) * (size_t)a.P * ((size_t)Cout * Cin + Cout);
}

// g: [Bt,Cout,L] gradient w.r.t. the conv output; x: [Bt,Cin,L] the conv's (pre-prologue) input;
// dw: [Cout,dw_cols] (first dw_cols columns of the Cout x Cin product); dbias: [Cout] or NULL.
// accumulate != 0: dw += ..., dbias += ... (else overwrite).
extern "C" int srf_pw_wgrad_ld(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout,
                               int L, float* dw, int dw_cols, int dw_ld, float* dbias, int accumulate, void* scratch,
                               void* stream);

extern "C" int srf_pw_wgrad_cols(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout,
                                 int L, float* dw, int dw_cols, float* dbias, int accumulate, void* scratch,
                                 void* stream) {
  return srf_pw_wgrad_ld(g, x, in_norm, Bt, Cin, Cout, L, dw, dw_cols, dw_cols, dbias, accumulate, scratch, stream);
}

// dw rows have a pitch of dw_ld floats (>= dw_cols): writes a column block of a wider matrix
extern "C" int srf_pw_wgrad_ld(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout,
                               int L, float* dw, int dw_cols, int dw_ld, float* dbias, int accumulate, void* scratch,
                               void* stream) {
  SRF_CHECK_ARG(g && x && dw && scratch, "srf_pw_wgrad: null pointer");
  SRF_CHECK_ARG(Bt > 0 && Cin > 0 && Cout > 0 && L > 0 && (L % 4) == 0, "srf_pw_wgrad: bad sizes (L %% 4 == 0 required)");
  SRF_CHECK_ARG(dw_cols > 0 && dw_cols <= Cin && dw_ld >= dw_cols, "srf_pw_wgrad: dw_cols / dw_ld out of range");
  SRF_CHECK_ARG(srf_aligned16(g) && srf_aligned16(x), "srf_pw_wgrad: operands must be 16-byte aligned");
  WgArgs a;
  a.g = g;
  a.x = x;
  a.nrm = srf_norm_dev(in_norm);
  a.inv_count = 1.0 / ((double)Cin * (double)L);
  a.M = Cout;
  a.N = Cin;
  a.L = L;
  a.Bt = Bt;
  wg_geometry(Cout, Cin, L, Bt, &a);
  a.part = reinterpret_cast<float*>(scratch);
  a.bias_part = dbias ? a.part + (size_t)a.P * Cout * Cin : nullptr;
  if (a.nrm.sums) SRF_CHECK_ARG(a.nrm.gamma && a.nrm.beta, "srf_pw_wgrad: norm without gamma/beta");
  hipStream_t st = (hipStream_t)stream;
  const int pro = a.nrm.sums ? (a.nrm.prelu ? 2 : 1) : (a.nrm.prelu ? 3 : 0);
  dim3 grid((unsigned)(a.nMt * a.nNt * a.P)), block(512);
  switch (pro) {
    case 0: hipLaunchKernelGGL(srf_pw_wgrad_kernel<0>, grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL(srf_pw_wgrad_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(srf_pw_wgrad_kernel<2>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(srf_pw_wgrad_kernel<3>, grid, block, 0, st, a); break;
  }
  SRF_CHECK_LAUNCH("pw_wgrad", st);
  const long nw = (long)Cout * dw_cols;
  hipLaunchKernelGGL(srf_wgrad_reduce_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, a.part, dw, Cout,
                     Cin, dw_cols, dw_ld, a.P, accumulate ? 1.f : 0.f);
  if (dbias)
    hipLaunchKernelGGL(srf_wgrad_reduce_kernel, dim3((unsigned)((Cout + 255) / 256)), dim3(256), 0, st, a.bias_part,
                       dbias, Cout, 1, 1, 1, a.P, accumulate ? 1.f : 0.f);
  SRF_CHECK_LAUNCH("pw_wgrad_reduce", st);
  return SRF_OK;
}

extern "C" int srf_pw_wgrad(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout, int L,
                            float* dw, float* dbias, int accumulate, void* scratch, void* stream) {
  return srf_pw_wgrad_cols(g, x, in_norm, Bt, Cin, Cout, L, dw, Cin, dbias, accumulate, scratch, stream);
}
