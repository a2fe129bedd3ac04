// K3+K4 fused -- the whole depthwise pyramid of a U-ConvBlock (D depthwise k=5 convs, each followed by
// a GlobLN, then the bottom-up nearest-x2 upsample+add; reference: improved_sudormrf.py:178-194,
// :206-216) in TWO passes over the expanded tensor instead of D+1 materialising kernels:
//
//   pass 1  srf_pyramid_moments   read y1 once; per (batch,channel) row run the *un-normalised* conv
//                                  cascade C_0 = d_0, C_k = w_k (*) C_{k-1} on chip and emit only row
//                                  moments {sum C_k, sum C_k^2, C_k[0], C_k[1], C_k[last]}
//   finalize srf_pyramid_finalize  per example: every level's GlobLN {mean, rstd} from those moments
//   pass 2  srf_pyramid_merge     read y1 again; recompute the cascade with the real GlobLNs on chip,
//                                  write the merged tensor (+ its {sum,sumsq} for final_norm)
//
// Why pass 1 works without knowing any statistics: GlobLN is affine per row, n = a*d + b*1, and the
// zero-padded conv is linear, so with alpha_k = prod a_m and a data-independent edge/constant term G_k
//      d_k = alpha_k * C_k + G_k,     G_k = bias_k + a_{k-1} conv_k(G_{k-1}) + b_{k-1} conv_k(1)
// G_k is a constant kappa_k in the interior and differs only at j = 0, 1 and L_k-1 (zero padding),
// hence  sum d_k   = alpha sum C + (L-3) kappa + g0 + g1 + gl
//        sum d_k^2 = alpha^2 sum C^2 + 2 alpha (kappa (sum C - e0 - e1 - el) + g0 e0 + g1 e1 + gl el)
//                    + (L-3) kappa^2 + g0^2 + g1^2 + gl^2          (e* = edge values of C_k)
// (validated against direct evaluation to 1e-16 in fp64).  HBM traffic per block drops from
// 7.75 C*L (D+1 kernels, every level written and re-read) to 3 C*L (y1 read twice, merged written).
//
// One 256-thread block per row; the row's levels live in LDS ([0..3] and [L_k+4..L_k+7] are physical
// zero pads so that taps never need masks).  Requires L % (4 * 2^(D-1)) == 0 and L >> (D-1) >= 8;
// otherwise srf_forward falls back to the per-level kernels (srf_dwconv.hip).
#include "srf_common.h"

struct PyrArgs {
  const float* y1;
  float* merged;
  SrfNormDev in_norm;   // GlobLN (+PReLU) of proj_1x1, applied to y1 on load
  double in_inv_count;
  const float* w[SRF_MAX_DEPTH];
  const float* bias[SRF_MAX_DEPTH];
  const float* gamma[SRF_MAX_DEPTH];
  const float* beta[SRF_MAX_DEPTH];
  const float* lvl;     // [groups][D][2] {mean, rstd}    (pass 2)
  const float* in_mr;   // [groups][2] {mean, rstd} of in_norm, pre-finalised (tiled kernels)
  float* d0;            // [rows][L] raw level-0 conv output: written by tiled pass 1, read by tiled pass 2
  double* mom;          // [rows][D][5]                   (pass 1)
  double* out_sums;     // merged statistics              (pass 2)
  long rows;
  int C, L, D;
};

__device__ __forceinline__ float srf_dot5(const float* w, float x0, float x1, float x2, float x3, float x4,
                                          float init) {
  return fmaf(w[4], x4, fmaf(w[3], x3, fmaf(w[2], x2, fmaf(w[1], x1, fmaf(w[0], x0, init)))));
}

// MOMENTS = true : pass 1 (raw cascade, moments out);  false : pass 2 (real GlobLNs, merged out)
template <bool MOMENTS>
__global__ __launch_bounds__(256) void srf_pyramid_kernel(PyrArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = a.L, D = a.D, C = a.C;
  const int tid = threadIdx.x;
  const int n4 = L >> 2;
  // Persistent blocks: each block walks rows blockIdx.x, +gridDim.x, ... and keeps the NEXT row's first
  // 1024 float4 groups in flight (registers) while it works on the current row, so the global-load
  // latency of a row is hidden behind the previous row's on-chip phases.
  // Each block owns a CONTIGUOUS chunk of rows (so the example index, hence the proj GlobLN statistics,
  // rarely changes inside a block: the fp64 finalisation is redone only when it does).
  const long rows_per_block = (a.rows + gridDim.x - 1) / gridDim.x;
  const long row_beg = (long)blockIdx.x * rows_per_block;
  const long row_end = min(row_beg + rows_per_block, a.rows);
  if (row_beg >= row_end) return;
  float4 pre[4];
  {
    const float4* y0 = reinterpret_cast<const float4*>(a.y1 + (size_t)row_beg * L);
#pragma unroll
    for (int i = 0; i < 4; ++i) pre[i] = y0[min(tid + 256 * i, n4 - 1)];
  }
  long cur_g = -1;
  float in_mean = 0.f, in_rstd = 1.f;
  for (long row = row_beg; row < row_end; ++row) {
  const int c = (int)(row % C);
  const long g = row / C;

  // LDS carve: bufB = level 0 (L+8 floats), bufA = proj output o, later levels 1..D-1 back to back
  float* bufB = lds;
  float* bufA = lds + (L + 8);
  int offA[SRF_MAX_DEPTH];
  int lv_total = 0;
  offA[0] = 0;
#pragma unroll
  for (int k = 1; k < SRF_MAX_DEPTH; ++k) {
    offA[k] = lv_total;
    if (k < D) lv_total += (L >> k) + 8;
  }
  const int sizeA = max(L + 8, lv_total);
  double* red = reinterpret_cast<double*>(lds + (L + 8) + ((sizeA + 3) & ~3));   // 8-B aligned: sizes % 4 == 0

  // ---- per-row coefficients
  float sc = 1.f, sh = 0.f;
  if (a.in_norm.sums) {
    if (g != cur_g) {   // block-uniform branch
      srf_finalize_stats(a.in_norm.sums, g, a.in_inv_count, in_mean, in_rstd);
      cur_g = g;
    }
    sc = a.in_norm.gamma[c] * in_rstd;
    sh = a.in_norm.beta[c] - in_mean * sc;
  }
  const bool act = a.in_norm.prelu != nullptr;
  const float slope = act ? a.in_norm.prelu[0] : 1.f;
  // per-level coefficients are fetched right where a level is computed (wave-uniform -> scalar loads
  // with short live ranges; holding all D levels at once cost ~190 VGPRs and 2/3 of the occupancy)
  struct Lv {
    float w[5], b, a, c;
  };
  auto level_coef = [&](int k) {
    Lv lv;
#pragma unroll
    for (int t = 0; t < 5; ++t) lv.w[t] = a.w[k][c * 5 + t];
    lv.b = a.bias[k][c];
    if (MOMENTS) {
      lv.a = 1.f;   // raw cascade: identity "norm", no bias below level 0
      lv.c = 0.f;
      if (k > 0) lv.b = 0.f;
    } else {
      const float m = a.lvl[(g * D + k) * 2 + 0], r = a.lvl[(g * D + k) * 2 + 1];
      lv.a = a.gamma[k][c] * r;
      lv.c = a.beta[k][c] - m * lv.a;
    }
    return lv;
  };

  // ---- zero pads of every buffer (4 floats each side)
  if (tid < 4) {
    bufB[tid] = 0.f;
    bufB[4 + L + tid] = 0.f;
    bufA[tid] = 0.f;
    bufA[4 + L + tid] = 0.f;
  }

  // ---- step 1: o = PReLU(GlobLN(y1)) -> bufA.  Groups 0..1023 come from the prefetch registers,
  // longer rows load their remaining groups here (4 loads in flight per thread).
  const float4* yrow = reinterpret_cast<const float4*>(a.y1 + (size_t)row * L);
  auto put_o = [&](float4 v, int q) {
    v.x = fmaf(v.x, sc, sh);
    v.y = fmaf(v.y, sc, sh);
    v.z = fmaf(v.z, sc, sh);
    v.w = fmaf(v.w, sc, sh);
    if (act) {
      v.x = srf_prelu(v.x, slope);
      v.y = srf_prelu(v.y, slope);
      v.z = srf_prelu(v.z, slope);
      v.w = srf_prelu(v.w, slope);
    }
    if (q < n4) *reinterpret_cast<float4*>(bufA + 4 + 4 * q) = v;
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) put_o(pre[i], tid + 256 * i);
  for (int base = 1024; base < n4; base += 1024) {
    float4 t4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t4[i] = yrow[min(base + tid + 256 * i, n4 - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i) put_o(t4[i], base + tid + 256 * i);
  }
  {  // prefetch the next row of this block (clamped to a valid row: surplus loads are harmless)
    const long nrow = (row + 1 < row_end) ? row + 1 : row;
    const float4* yn = reinterpret_cast<const float4*>(a.y1 + (size_t)nrow * L);
#pragma unroll
    for (int i = 0; i < 4; ++i) pre[i] = yn[min(tid + 256 * i, n4 - 1)];
  }
  __syncthreads();

  // row moments: <= 16 values per thread and level in fp32, fp32 wave reduction, fp64 only across waves
  float s1[SRF_MAX_DEPTH], s2[SRF_MAX_DEPTH];
#pragma unroll
  for (int k = 0; k < SRF_MAX_DEPTH; ++k) s1[k] = s2[k] = 0.f;

  // ---- step 2: level 0 (stride 1) from bufA -> bufB
  const Lv l0 = level_coef(0);
  for (int q = tid; q < n4; q += 256) {
    const float* p = bufA + 4 + 4 * q;   // o[4q]
    const float2 lo = *reinterpret_cast<const float2*>(p - 2);
    const float4 mi = *reinterpret_cast<const float4*>(p);
    const float2 hi = *reinterpret_cast<const float2*>(p + 4);
    float4 d;
    d.x = srf_dot5(l0.w, lo.x, lo.y, mi.x, mi.y, mi.z, l0.b);
    d.y = srf_dot5(l0.w, lo.y, mi.x, mi.y, mi.z, mi.w, l0.b);
    d.z = srf_dot5(l0.w, mi.x, mi.y, mi.z, mi.w, hi.x, l0.b);
    d.w = srf_dot5(l0.w, mi.y, mi.z, mi.w, hi.x, hi.y, l0.b);
    if (MOMENTS) {
      s1[0] += (d.x + d.y) + (d.z + d.w);
      s2[0] += fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w)));
      if (q == 0) {
        a.mom[(row * D + 0) * 5 + 2] = (double)d.x;
        a.mom[(row * D + 0) * 5 + 3] = (double)d.y;
      }
      if (q == n4 - 1) a.mom[(row * D + 0) * 5 + 4] = (double)d.w;
    } else {
      d.x = fmaf(d.x, l0.a, l0.c);
      d.y = fmaf(d.y, l0.a, l0.c);
      d.z = fmaf(d.z, l0.a, l0.c);
      d.w = fmaf(d.w, l0.a, l0.c);
    }
    *reinterpret_cast<float4*>(bufB + 4 + 4 * q) = d;
  }
  __syncthreads();   // bufA (o) is dead from here on: it now hosts levels 1..D-1

  // ---- step 3: levels 1..D-1 (stride 2); level k reads level k-1
#pragma unroll
  for (int k = 1; k < SRF_MAX_DEPTH; ++k) {
    if (k < D) {
      const int Lk = L >> k;
      const Lv lk = level_coef(k);
      const float* src = (k == 1) ? bufB : (bufA + offA[k - 1]);
      float* dst = bufA + offA[k];
      if (tid < 4) {
        dst[tid] = 0.f;
        dst[4 + Lk + tid] = 0.f;
      }
      const int nq = Lk >> 2;
      for (int q = tid; q < nq; q += 256) {
        const float* p = src + 4 + 8 * q;   // in[2 * (4q)]
        const float2 l2 = *reinterpret_cast<const float2*>(p - 2);
        const float4 A = *reinterpret_cast<const float4*>(p);
        const float4 B = *reinterpret_cast<const float4*>(p + 4);
        const float r = p[8];
        float4 d;
        d.x = srf_dot5(lk.w, l2.x, l2.y, A.x, A.y, A.z, lk.b);
        d.y = srf_dot5(lk.w, A.x, A.y, A.z, A.w, B.x, lk.b);
        d.z = srf_dot5(lk.w, A.z, A.w, B.x, B.y, B.z, lk.b);
        d.w = srf_dot5(lk.w, B.x, B.y, B.z, B.w, r, lk.b);
        if (MOMENTS) {
          s1[k] += (d.x + d.y) + (d.z + d.w);
          s2[k] += fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w)));
          if (q == 0) {
            a.mom[(row * D + k) * 5 + 2] = (double)d.x;
            a.mom[(row * D + k) * 5 + 3] = (double)d.y;
          }
          if (q == nq - 1) a.mom[(row * D + k) * 5 + 4] = (double)d.w;
        } else {
          d.x = fmaf(d.x, lk.a, lk.c);
          d.y = fmaf(d.y, lk.a, lk.c);
          d.z = fmaf(d.z, lk.a, lk.c);
          d.w = fmaf(d.w, lk.a, lk.c);
        }
        *reinterpret_cast<float4*>(dst + 4 + 4 * q) = d;
      }
      __syncthreads();
    }
  }

  if (MOMENTS) {
    // ---- block reduction of the 2D row moments -> mom[row][k][0..1]
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int k = 0; k < SRF_MAX_DEPTH; ++k) {
      if (k < D) {
        float r1 = s1[k], r2 = s2[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          r1 += __shfl_xor(r1, o, 64);
          r2 += __shfl_xor(r2, o, 64);
        }
        if (lane == 0) {
          red[(w * SRF_MAX_DEPTH + k) * 2 + 0] = (double)r1;
          red[(w * SRF_MAX_DEPTH + k) * 2 + 1] = (double)r2;
        }
      }
    }
    __syncthreads();
    if (tid < 2 * D) {
      const int k = tid >> 1, j = tid & 1;
      double t = 0.0;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) t += red[(ww * SRF_MAX_DEPTH + k) * 2 + j];
      a.mom[(row * D + k) * 5 + j] = t;
    }
  } else {
    // ---- step 4: merged[j] = n_0[j] + (n_1[j>>1] + (... + n_{D-1}[j>>(D-1)]))   (reference order)
    float4* mrow = reinterpret_cast<float4*>(a.merged + (size_t)row * L);
    double ms = 0.0, mq = 0.0;
    for (int q = tid; q < n4; q += 256) {
      float t = 0.f;
      bool have = false;
#pragma unroll
      for (int k = SRF_MAX_DEPTH - 1; k >= 2; --k) {
        if (k < D) {
          const float nk = bufA[offA[k] + 4 + ((4 * q) >> k)];
          t = have ? nk + t : nk;
          have = true;
        }
      }
      float4 o = *reinterpret_cast<const float4*>(bufB + 4 + 4 * q);
      if (D > 1) {
        const float2 e = *reinterpret_cast<const float2*>(bufA + offA[1] + 4 + 2 * q);
        const float ta = have ? e.x + t : e.x, tb = have ? e.y + t : e.y;
        o.x += ta;
        o.y += ta;
        o.z += tb;
        o.w += tb;
      }
      mrow[q] = o;
      ms += (double)((o.x + o.y) + (o.z + o.w));
      mq += (double)fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, o.w * o.w)));
    }
    if (a.out_sums) srf_block_stats_atomic<4>(ms, mq, srf_stat_slot(a.out_sums, g, row), red);
  }
  __syncthreads();   // the next row re-uses every LDS buffer
  }  // row loop
}

// ---------------------------------------------------------------------------------------------
// Tiled variant (preferred): one WAVEFRONT per (row, time tile of TS level-0 outputs), no block
// barriers at all.  The block-per-row kernel above is latency-bound (7 barriers per row, <= 16 waves
// per CU: measured 157 + 188 us per U-ConvBlock, no better than the per-level kernels); here every
// wave owns a private ~6 KB LDS slice, recomputes a halo of h_k = 2 h_{k+1} + 4 positions per level
// (h_{D-1} = 0) instead of synchronising with its neighbours, and ~20+ independent waves per CU keep
// HBM busy.  Row moments of pass 1 are accumulated with fp64 atomics over the few tiles of a row.
// ---------------------------------------------------------------------------------------------
struct PyrTile {
  int TS, tiles;          // level-0 outputs per tile, tiles per row
  int h[SRF_MAX_DEPTH];   // halo per level (level-k units)
  int wave_floats;        // LDS floats per wave
  long tasks;             // rows * tiles
};

template <bool MOMENTS>
__global__ __launch_bounds__(256) void srf_pyramid_tile_kernel(PyrArgs a, PyrTile t) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = a.L, D = a.D, C = a.C;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long task = (long)blockIdx.x * 4 + wave;
  if (task >= t.tasks) return;   // wave-uniform; no block-level synchronisation anywhere below
  const long row = task / t.tiles;
  const int tile = (int)(task - row * t.tiles);
  const int c = (int)(row % C);
  const long g = row / C;
  const int j0 = tile * t.TS;

  float* bufB = lds + wave * t.wave_floats;        // level 0
  const int len0 = t.TS + 2 * t.h[0] + 8;          // o / level-0 buffer length incl. 4 + 4 pad
  float* bufA = bufB + len0;                       // o, later levels 1..D-1
  int offA[SRF_MAX_DEPTH];
  {
    int o = 0;
    offA[0] = 0;
#pragma unroll
    for (int k = 1; k < SRF_MAX_DEPTH; ++k) {
      offA[k] = o;
      if (k < D) o += (t.TS >> k) + 2 * t.h[k] + 8;
    }
  }

  // ---- per-task coefficient table in LDS: ONE batch of (vector) loads for all D levels, issued together
  // with the first data loads, instead of D dependent rounds of scalar loads (each ~1-2 us under load)
  float* coef = bufB + t.wave_floats - 8 * SRF_MAX_DEPTH;   // [level][8] = w0..w4, bias, a, c
  {
    const int k = lane >> 3, j = lane & 7;
    float v = 0.f;
    if (k < D) {
      if (j < 5) {
        v = a.w[k][c * 5 + j];
      } else if (j == 5) {
        v = (MOMENTS && k > 0) ? 0.f : a.bias[k][c];
      } else if (MOMENTS) {
        v = (j == 6) ? 1.f : 0.f;
      } else {
        const float m = a.lvl[(g * D + k) * 2 + 0], r = a.lvl[(g * D + k) * 2 + 1];
        const float aa = a.gamma[k][c] * r;
        v = (j == 6) ? aa : (a.beta[k][c] - m * aa);
      }
    }
    coef[lane] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // table visible to the whole wave
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float sc = 1.f, sh = 0.f;
  if (a.in_norm.sums) {   // statistics pre-finalised once per example by srf_stats_finalize_kernel
    const float mean = a.in_mr[2 * g], rstd = a.in_mr[2 * g + 1];
    sc = a.in_norm.gamma[c] * rstd;
    sh = a.in_norm.beta[c] - mean * sc;
  }
  const bool act = a.in_norm.prelu != nullptr;
  const float slope = act ? a.in_norm.prelu[0] : 1.f;
  struct Lv {
    float w[5], b, a, c;
  };
  auto level_coef = [&](int k) {   // broadcast reads (uniform address)
    const float4 c0 = *reinterpret_cast<const float4*>(coef + 8 * k);
    const float4 c1 = *reinterpret_cast<const float4*>(coef + 8 * k + 4);
    Lv lv;
    lv.w[0] = c0.x;
    lv.w[1] = c0.y;
    lv.w[2] = c0.z;
    lv.w[3] = c0.w;
    lv.w[4] = c1.x;
    lv.b = c1.y;
    lv.a = c1.z;
    lv.c = c1.w;
    return lv;
  };
  auto wave_sync = [&]() {   // LDS ops of one wave execute in order; this only pins the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  const int lo0 = j0 - t.h[0];
  const int n4 = L >> 2;
  float s1[SRF_MAX_DEPTH], s2[SRF_MAX_DEPTH];
#pragma unroll
  for (int k = 0; k < SRF_MAX_DEPTH; ++k) s1[k] = s2[k] = 0.f;

  if (MOMENTS) {
    // ---- pass 1, step 1: o = PReLU(GlobLN(y1)) on [lo0 - 4, lo0 + TS + 2 h0 + 4)  (zero outside the row)
    const float4* yrow = reinterpret_cast<const float4*>(a.y1 + (size_t)row * L);
    for (int q = lane; q < (len0 >> 2); q += 64) {
      const int p4 = ((lo0 - 4) >> 2) + q;   // float4 group index in the row (may be out of range)
      const bool ok = p4 >= 0 && p4 < n4;
      float4 v = yrow[min(max(p4, 0), n4 - 1)];
      v.x = fmaf(v.x, sc, sh);
      v.y = fmaf(v.y, sc, sh);
      v.z = fmaf(v.z, sc, sh);
      v.w = fmaf(v.w, sc, sh);
      if (act) {
        v.x = srf_prelu(v.x, slope);
        v.y = srf_prelu(v.y, slope);
        v.z = srf_prelu(v.z, slope);
        v.w = srf_prelu(v.w, slope);
      }
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(bufA + 4 * q) = v;
    }
    wave_sync();
    // ---- pass 1, step 2: raw level 0 on [lo0, lo0 + TS + 2 h0); the tile's own range goes to HBM (d0)
    // so that pass 2 starts from it instead of redoing the GlobLN + PReLU + conv of y1
    const Lv l0 = level_coef(0);
    float4* drow = reinterpret_cast<float4*>(a.d0 + (size_t)row * L);
    const int nq = (t.TS + 2 * t.h[0]) >> 2;
    for (int q = lane; q < nq; q += 64) {
      const float* p = bufA + 4 + 4 * q;
      const float2 lo = *reinterpret_cast<const float2*>(p - 2);
      const float4 mi = *reinterpret_cast<const float4*>(p);
      const float2 hi = *reinterpret_cast<const float2*>(p + 4);
      float4 d;
      d.x = srf_dot5(l0.w, lo.x, lo.y, mi.x, mi.y, mi.z, l0.b);
      d.y = srf_dot5(l0.w, lo.y, mi.x, mi.y, mi.z, mi.w, l0.b);
      d.z = srf_dot5(l0.w, mi.x, mi.y, mi.z, mi.w, hi.x, l0.b);
      d.w = srf_dot5(l0.w, mi.y, mi.z, mi.w, hi.x, hi.y, l0.b);
      const int pos = lo0 + 4 * q;
      if (pos >= j0 && pos < j0 + t.TS) {   // own range only (halo belongs to the neighbours)
        drow[pos >> 2] = d;
        s1[0] += (d.x + d.y) + (d.z + d.w);
        s2[0] += fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w)));
        if (pos == 0) {
          a.mom[(row * D + 0) * 5 + 2] = (double)d.x;
          a.mom[(row * D + 0) * 5 + 3] = (double)d.y;
        }
        if (pos == L - 4) a.mom[(row * D + 0) * 5 + 4] = (double)d.w;
      }
      if (pos < 0 || pos >= L) d = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding of the next conv
      *reinterpret_cast<float4*>(bufB + 4 + 4 * q) = d;
    }
  } else {
    // ---- pass 2, steps 1+2: n_0 = GlobLN_0(d0) straight from HBM on [lo0, lo0 + TS + 2 h0)
    const Lv l0 = level_coef(0);
    const float4* drow = reinterpret_cast<const float4*>(a.d0 + (size_t)row * L);
    const int nq = (t.TS + 2 * t.h[0]) >> 2;
    for (int q = lane; q < nq; q += 64) {
      const int p4 = (lo0 >> 2) + q;
      const bool ok = p4 >= 0 && p4 < n4;
      float4 d = drow[min(max(p4, 0), n4 - 1)];
      d.x = fmaf(d.x, l0.a, l0.c);
      d.y = fmaf(d.y, l0.a, l0.c);
      d.z = fmaf(d.z, l0.a, l0.c);
      d.w = fmaf(d.w, l0.a, l0.c);
      if (!ok) d = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(bufB + 4 + 4 * q) = d;
    }
  }
  wave_sync();

  // ---- step 3: levels 1..D-1
#pragma unroll
  for (int k = 1; k < SRF_MAX_DEPTH; ++k) {
    if (k < D) {
      const int Lk = L >> k;
      const Lv lk = level_coef(k);
      const float* src = (k == 1) ? bufB : (bufA + offA[k - 1]);
      float* dst = bufA + offA[k];
      const int own_lo = j0 >> k, own_len = t.TS >> k;
      const int lo = own_lo - t.h[k];
      const int nq = (own_len + 2 * t.h[k]) >> 2;
      for (int q = lane; q < nq; q += 64) {
        const float* p = src + 8 + 8 * q;   // input position 2 * (lo + 4q)
        const float2 l2 = *reinterpret_cast<const float2*>(p - 2);
        const float4 A = *reinterpret_cast<const float4*>(p);
        const float4 B = *reinterpret_cast<const float4*>(p + 4);
        const float r = p[8];
        float4 d;
        d.x = srf_dot5(lk.w, l2.x, l2.y, A.x, A.y, A.z, lk.b);
        d.y = srf_dot5(lk.w, A.x, A.y, A.z, A.w, B.x, lk.b);
        d.z = srf_dot5(lk.w, A.z, A.w, B.x, B.y, B.z, lk.b);
        d.w = srf_dot5(lk.w, B.x, B.y, B.z, B.w, r, lk.b);
        const int pos = lo + 4 * q;
        if (MOMENTS) {
          if (pos >= own_lo && pos < own_lo + own_len) {
            s1[k] += (d.x + d.y) + (d.z + d.w);
            s2[k] += fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w)));
            if (pos == 0) {
              a.mom[(row * D + k) * 5 + 2] = (double)d.x;
              a.mom[(row * D + k) * 5 + 3] = (double)d.y;
            }
            if (pos == Lk - 4) a.mom[(row * D + k) * 5 + 4] = (double)d.w;
          }
        } else {
          d.x = fmaf(d.x, lk.a, lk.c);
          d.y = fmaf(d.y, lk.a, lk.c);
          d.z = fmaf(d.z, lk.a, lk.c);
          d.w = fmaf(d.w, lk.a, lk.c);
        }
        if (pos < 0 || pos >= Lk) d = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(dst + 4 + 4 * q) = d;
      }
      wave_sync();
    }
  }

  if (MOMENTS) {
#pragma unroll
    for (int k = 0; k < SRF_MAX_DEPTH; ++k) {
      if (k < D) {
        float r1 = s1[k], r2 = s2[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          r1 += __shfl_xor(r1, o, 64);
          r2 += __shfl_xor(r2, o, 64);
        }
        if (lane == 0) {
          atomicAdd(&a.mom[(row * D + k) * 5 + 0], (double)r1);
          atomicAdd(&a.mom[(row * D + k) * 5 + 1], (double)r2);
        }
      }
    }
  } else {
    // ---- step 4: merge the tile's own level-0 range
    float4* mrow = reinterpret_cast<float4*>(a.merged + (size_t)row * L + j0);
    float ms = 0.f, mq = 0.f;
    for (int q = lane; q < (t.TS >> 2); q += 64) {
      float tt = 0.f;
      bool have = false;
#pragma unroll
      for (int k = SRF_MAX_DEPTH - 1; k >= 2; --k) {
        if (k < D) {
          const float nk = bufA[offA[k] + 4 + t.h[k] + ((4 * q) >> k)];
          tt = have ? nk + tt : nk;
          have = true;
        }
      }
      float4 o = *reinterpret_cast<const float4*>(bufB + 4 + t.h[0] + 4 * q);
      if (D > 1) {
        const float2 e = *reinterpret_cast<const float2*>(bufA + offA[1] + 4 + t.h[1] + 2 * q);
        const float ta = have ? e.x + tt : e.x, tb = have ? e.y + tt : e.y;
        o.x += ta;
        o.y += ta;
        o.z += tb;
        o.w += tb;
      }
      mrow[q] = o;
      ms += (o.x + o.y) + (o.z + o.w);
      mq += fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, o.w * o.w)));
    }
    if (a.out_sums) {
      const double ds = srf_wave_sum((double)ms), dq = srf_wave_sum((double)mq);
      if (lane == 0) {
        double* dst = srf_stat_slot(a.out_sums, g, task);
        atomicAdd(dst, ds);
        atomicAdd(dst + 1, dq);
      }
    }
  }
}

// {sum,sumsq} buckets -> {mean, rstd} per group, once (one wavefront per group)
__global__ __launch_bounds__(64) void srf_stats_finalize_kernel(const double* sums, double inv_count,
                                                                float* out) {
  float mean, rstd;
  srf_finalize_stats(sums, blockIdx.x, inv_count, mean, rstd);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = mean;
    out[2 * blockIdx.x + 1] = rstd;
  }
}

// ---------------------------------------------------------------------------------------------
// finalize: per example, all levels' {mean, rstd} from the row moments (fp64)
// ---------------------------------------------------------------------------------------------
struct PyrFinArgs {
  const double* mom;   // [groups*C][D][5] {S1, S2, e0, e1, el}
  const float* w[SRF_MAX_DEPTH];
  const float* bias[SRF_MAX_DEPTH];
  const float* gamma[SRF_MAX_DEPTH];
  const float* beta[SRF_MAX_DEPTH];
  float* lvl;          // [groups][D][2]
  const double* in_sums;   // statistics of the pyramid's input norm (may be null) -> in_mr[g] = {mean, rstd}
  double in_inv_count;
  float* in_mr;
  int C, L, D;
  double* lv_sums[SRF_MAX_DEPTH];   // training forward: every level's {sum d_k, sum d_k^2} goes to bucket 0 of its
                                    // GlobLN statistic slot [groups][SRF_STAT_BUCKETS][2] (zeroed by the caller), or null
};

#define SRF_FIN_CPT 8   // channels per thread (C <= 256 * 8)
// CPT = channels per thread (C <= 256 * CPT): 2 covers the published models (C = 512) and keeps the whole example's
// moments in registers; 8 is the general fallback (moments fetched level by level).
template <int CPT>
__global__ __launch_bounds__(256) void srf_pyramid_finalize_kernel(PyrFinArgs a) {
  __shared__ double red[16];
  const long g = blockIdx.x;
  const int tid = threadIdx.x, C = a.C, D = a.D;
  if (a.in_sums && tid < 64) {   // wavefront 0: the input norm's {mean, rstd} for pass 2
    float m, r;
    srf_finalize_stats(a.in_sums, g, a.in_inv_count, m, r);
    if (tid == 0) {
      a.in_mr[2 * g] = m;
      a.in_mr[2 * g + 1] = r;
    }
  }
  double alpha[CPT], kap[CPT], g0[CPT], g1[CPT], gl[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    alpha[i] = 1.0;
    kap[i] = g0[i] = g1[i] = gl[i] = 0.0;
  }
  // Latency, not work, is what this kernel costs (one block per example, D dependent block reductions): all of the
  // example's moments are requested up front, and every level's coefficient loads are issued before that level's
  // reduction so that their latency overlaps the shuffles and barriers (was 15 us per launch with the loads inside
  // the dependent chain).
  constexpr bool PRE = CPT <= 2;
  double mo[PRE ? SRF_MAX_DEPTH : 1][CPT][5];
  auto load_level = [&](int slot, int k) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = min(tid + 256 * i, C - 1);   // clamped: unconditional loads
      const double* m = a.mom + ((g * C + c) * D + k) * 5;
#pragma unroll
      for (int e = 0; e < 5; ++e) mo[slot][i][e] = m[e];
    }
  };
  if (PRE) {
#pragma unroll
    for (int k = 0; k < SRF_MAX_DEPTH; ++k) load_level(PRE ? k : 0, k < D ? k : 0);
  }
#pragma unroll
  for (int k = 0; k < SRF_MAX_DEPTH; ++k) {
    if (k >= D) break;
    const int Lk = a.L >> k;
    const int slot = PRE ? k : 0;
    if (!PRE) load_level(0, k);
    // next level's coefficients (independent of this level's statistics)
    float gam_[CPT], bet_[CPT], w_[CPT][5], bias_[CPT];
    if (k + 1 < D) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int c = min(tid + 256 * i, C - 1);
        gam_[i] = a.gamma[k][c];
        bet_[i] = a.beta[k][c];
        bias_[i] = a.bias[k + 1][c];
#pragma unroll
        for (int t = 0; t < 5; ++t) w_[i][t] = a.w[k + 1][c * 5 + t];
      }
    }
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + 256 * i;
      if (c < C) {
        const double S1 = mo[slot][i][0], S2 = mo[slot][i][1], e0 = mo[slot][i][2], e1 = mo[slot][i][3], el = mo[slot][i][4];
        s += alpha[i] * S1 + (Lk - 3) * kap[i] + g0[i] + g1[i] + gl[i];
        q += alpha[i] * alpha[i] * S2 +
             2.0 * alpha[i] * (kap[i] * (S1 - e0 - e1 - el) + g0[i] * e0 + g1[i] * e1 + gl[i] * el) +
             (Lk - 3) * kap[i] * kap[i] + g0[i] * g0[i] + g1[i] * g1[i] + gl[i] * gl[i];
      }
    }
    s = srf_dpp_wave_sum(s);   // VALU-only reductions: this chain is the kernel's critical path
    q = srf_dpp_wave_sum(q);
    if ((tid & 63) == 63) {
      red[(tid >> 6) * 2] = s;
      red[(tid >> 6) * 2 + 1] = q;
    }
    __syncthreads();
    // every thread finalises the level itself (same inputs, same arithmetic: identical values) -- no second
    // barrier-and-broadcast round
    float mean_f, rstd_f;
    {
      double ts = 0.0, tq = 0.0;
      for (int w = 0; w < 4; ++w) {
        ts += red[2 * w];
        tq += red[2 * w + 1];
      }
      const double n = (double)C * (double)Lk;
      const double mean = ts / n;
      double var = tq / n - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      mean_f = (float)mean;
      rstd_f = (float)(1.0 / sqrt(var + 1e-8));
      if (tid == 0) {
        a.lvl[(g * D + k) * 2 + 0] = mean_f;
        a.lvl[(g * D + k) * 2 + 1] = rstd_f;
        if (a.lv_sums[k]) {   // what the un-fused path's per-level statistics would hold (consumer: the backward)
          double* dst = a.lv_sums[k] + (size_t)g * SRF_STAT_BUCKETS * 2;
          dst[0] = ts;
          dst[1] = tq;
        }
      }
    }
    if (k + 1 < D) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int c = tid + 256 * i;
        if (c < C) {
          // exactly the fp32 coefficients pass 2 will use
          const float af = gam_[i] * rstd_f;
          const float bf = bet_[i] - mean_f * af;
          const double A = (double)af, B = (double)bf;
          const double w0 = w_[i][0], w1 = w_[i][1], w2 = w_[i][2], w3 = w_[i][3], w4 = w_[i][4];
          const double bs = (double)bias_[i];
          const double sw = w0 + w1 + w2 + w3 + w4;
          const double nk = bs + A * kap[i] * sw + B * sw;
          const double n0 = bs + A * (w2 * g0[i] + w3 * g1[i] + w4 * kap[i]) + B * (w2 + w3 + w4);
          const double n1 = bs + A * (w0 * g0[i] + w1 * g1[i] + (w2 + w3 + w4) * kap[i]) + B * sw;
          const double nl = bs + A * ((w0 + w1 + w2) * kap[i] + w3 * gl[i]) + B * (w0 + w1 + w2 + w3);
          kap[i] = nk;
          g0[i] = n0;
          g1[i] = n1;
          gl[i] = nl;
          alpha[i] *= A;
        }
      }
    }
    __syncthreads();   // red[] is rewritten by the next level
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
static size_t pyr_lds_bytes(int L, int D) {
  size_t sizeA = (size_t)L + 8, lv = 0;
  for (int k = 1; k < D; ++k) lv += (size_t)(L >> k) + 8;
  if (lv > sizeA) sizeA = lv;
  sizeA = (sizeA + 3) & ~(size_t)3;
  return sizeof(float) * ((size_t)L + 8 + sizeA) + sizeof(double) * (4 * SRF_MAX_DEPTH * 2 + 8);
}

static void srf_pyramid_finalize_launch(const PyrFinArgs& f, int groups, int C, hipStream_t st) {
  if (C <= 512)
    hipLaunchKernelGGL(srf_pyramid_finalize_kernel<2>, dim3((unsigned)groups), dim3(256), 0, st, f);
  else
    hipLaunchKernelGGL((srf_pyramid_finalize_kernel<SRF_FIN_CPT>), dim3((unsigned)groups), dim3(256), 0, st, f);
}

bool srf_pyramid_reg_supported(int L, int D);
extern "C" int srf_pyramid_supported(int C, int L, int D) {
  if (D < 1 || D > SRF_MAX_DEPTH || C > 256 * SRF_FIN_CPT) return 0;
  if ((L >> (D - 1)) < 8 || (L % (1 << (D - 1))) != 0) return 0;   // finalize needs distinct edge positions
  if (!(srf_debug_flags() & 64) && srf_pyramid_reg_supported(L, D)) return 1;   // register-resident kernels
  if (L % (4 << (D - 1)) != 0) return 0;                              // LDS kernels work on float4 groups
  return pyr_lds_bytes(L, D) <= 160 * 1024 - 1024;
}

// scratch layout: moments [groups*C][D][5] fp64 | {mean,rstd} per level [groups][D][2] + of in_norm
// [groups][2] fp32 | (256-B aligned) raw level-0 tensor d0 [groups*C][L] fp32
static size_t pyr_small_bytes(int groups, int C, int D) {
  const size_t b = sizeof(double) * (size_t)groups * C * D * 5 + sizeof(float) * (size_t)groups * (D * 2 + 2);
  return (b + 255) & ~(size_t)255;
}
extern "C" size_t srf_pyramid_scratch_bytes(int groups, int C, int L, int D) {
  return pyr_small_bytes(groups, C, D) + sizeof(float) * (size_t)groups * C * L;
}

// Tile choice for the wave-per-tile kernels: TS = q * 4 * 2^(D-1) with q | (L / (4*2^(D-1))), 384..1024
// level-0 outputs per tile when such a divisor exists.  Returns false -> use the block-per-row kernel.
static bool pyr_pick_tile(int L, int D, PyrTile* t) {
  const int unit = 4 << (D - 1);
  const int m = L / unit;
  int best = 0;
  for (int q = 1; q <= m; ++q) {
    if (m % q) continue;
    const int ts = q * unit;
    if (ts >= 384 && ts <= 1024) {
      if (!best || abs(ts - 640) < abs(best - 640)) best = ts;
    }
  }
  if (!best || best == L) return false;
  t->TS = best;
  t->tiles = L / best;
  int h = 0;
  for (int k = SRF_MAX_DEPTH - 1; k >= 0; --k) {
    if (k >= D) {
      t->h[k] = 0;
      continue;
    }
    t->h[k] = (k == D - 1) ? 0 : 2 * h + 4;
    h = t->h[k];
  }
  const int len0 = best + 2 * t->h[0] + 8;
  int lv = 0;
  for (int k = 1; k < D; ++k) lv += (best >> k) + 2 * t->h[k] + 8;
  const int lenA = lv > len0 ? lv : len0;
  t->wave_floats = len0 + ((lenA + 3) & ~3) + 8 * SRF_MAX_DEPTH;   // + coefficient table
  return (size_t)t->wave_floats * 4 * sizeof(float) <= 64 * 1024;
}

// register-resident kernels (srf_pyramid_reg.hip)
#include "srf_pyr.h"


// lv_out / lv_sums (both or neither; register-resident kernels only -- srf_pyramid_reg_supported): the training
// forward's extra outputs, see PyrRegArgs::lv_out and PyrFinArgs::lv_sums.
int srf_pyramid_impl(const float* y1, float* merged, const srf_norm* in_norm, const float* const* w,
                     const float* const* bias, const float* const* gamma, const float* const* beta, int groups, int C,
                     int L, int D, void* scratch, double* out_sums, float* const* lv_out, double* const* lv_sums,
                     void* stream);

extern "C" int srf_pyramid(const float* y1, float* merged, const srf_norm* in_norm,
                           const float* const* w, const float* const* bias, const float* const* gamma,
                           const float* const* beta, int groups, int C, int L, int D, void* scratch,
                           double* out_sums, void* stream) {
  return srf_pyramid_impl(y1, merged, in_norm, w, bias, gamma, beta, groups, C, L, D, scratch, out_sums, nullptr, nullptr,
                          stream);
}

int srf_pyramid_impl(const float* y1, float* merged, const srf_norm* in_norm, const float* const* w,
                     const float* const* bias, const float* const* gamma, const float* const* beta, int groups, int C,
                     int L, int D, void* scratch, double* out_sums, float* const* lv_out, double* const* lv_sums,
                     void* stream) {
  SRF_CHECK_ARG((lv_out == nullptr) == (lv_sums == nullptr), "srf_pyramid: level outputs and level sums come together");
  SRF_CHECK_ARG(y1 && merged && w && bias && gamma && beta && scratch, "srf_pyramid: null pointer");
  SRF_CHECK_ARG(groups > 0 && C > 0 && L > 0, "srf_pyramid: bad sizes");
  SRF_CHECK_ARG(srf_pyramid_supported(C, L, D), "srf_pyramid: unsupported shape C=%d L=%d D=%d", C, L, D);
  SRF_CHECK_ARG(srf_aligned16(y1) && srf_aligned16(merged) && srf_aligned16(scratch),
                "srf_pyramid: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)groups * C;
  SRF_CHECK_ARG(rows < (1L << 31), "srf_pyramid: too many rows");
  double* mom = reinterpret_cast<double*>(scratch);
  float* lvl = reinterpret_cast<float*>(mom + (size_t)rows * D * 5);
  PyrArgs a;
  PyrFinArgs f;
  a.y1 = y1;
  a.merged = merged;
  a.in_norm = srf_norm_dev(in_norm);
  a.in_inv_count = 1.0 / ((double)C * (double)L);
  for (int k = 0; k < SRF_MAX_DEPTH; ++k) {
    const int kk = k < D ? k : 0;
    SRF_CHECK_ARG(w[kk] && bias[kk] && gamma[kk] && beta[kk], "srf_pyramid: null level parameter %d", kk);
    a.w[k] = f.w[k] = w[kk];
    a.bias[k] = f.bias[k] = bias[kk];
    a.gamma[k] = f.gamma[k] = gamma[kk];
    a.beta[k] = f.beta[k] = beta[kk];
  }
  a.lvl = lvl;
  a.in_mr = lvl + (size_t)groups * D * 2;
  a.d0 = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + pyr_small_bytes(groups, C, D));
  a.mom = mom;
  a.out_sums = out_sums;
  a.rows = rows;
  a.C = C;
  a.L = L;
  a.D = D;
  f.mom = mom;
  f.lvl = lvl;
  f.in_sums = nullptr;   // the LDS kernels get in_mr from srf_stats_finalize_kernel before pass 1
  f.in_inv_count = a.in_inv_count;
  f.in_mr = const_cast<float*>(a.in_mr);
  f.C = C;
  f.L = L;
  f.D = D;
  for (int k = 0; k < SRF_MAX_DEPTH; ++k) f.lv_sums[k] = (lv_sums && k < D) ? lv_sums[k] : nullptr;
  SRF_CHECK_ARG(!lv_out || (!(srf_debug_flags() & (64 | 128)) && srf_pyramid_reg_supported(L, D)),
                "srf_pyramid: level outputs need the register-resident kernels");
  if (!(srf_debug_flags() & 64) && srf_pyramid_reg_supported(L, D)) {
    PyrRegArgs r;
    for (int k = 0; k < SRF_MAX_DEPTH; ++k) r.lv_out[k] = (lv_out && k < D) ? lv_out[k] : nullptr;
    r.save = lv_out != nullptr;
    r.y1 = y1;
    r.d0 = a.d0;
    r.merged = merged;
    r.in_norm = a.in_norm;
    r.in_mr = a.in_mr;
    r.in_inv_count = a.in_inv_count;
    for (int k = 0; k < SRF_MAX_DEPTH; ++k) {
      r.w[k] = a.w[k];
      r.bias[k] = a.bias[k];
      r.gamma[k] = a.gamma[k];
      r.beta[k] = a.beta[k];
    }
    r.lvl = lvl;
    r.mom = mom;
    r.out_sums = out_sums;
    r.C = C;
    r.L = L;
    r.D = D;
    r.rows = r.rpw = 0;
    r.tiles = r.own = 0;
    if (srf_debug_flags() & 128) {   // non-persistent pass 1: atomics into mom + pre-finalised statistics
      SRF_CHECK_HIP(hipMemsetAsync(mom, 0, sizeof(double) * (size_t)rows * D * 5, st));
      if (a.in_norm.sums) {
        hipLaunchKernelGGL(srf_stats_finalize_kernel, dim3((unsigned)groups), dim3(64), 0, st, a.in_norm.sums,
                           a.in_inv_count, const_cast<float*>(a.in_mr));
        SRF_CHECK_LAUNCH("stats_finalize", st);
      }
    }
    SRF_CHECK_ARG(merged != y1, "srf_pyramid: merged must not alias y1 (pass 2 re-reads y1 with halos)");
    int rc = srf_pyramid_reg_launch(r, true, rows, st);
    if (rc) return rc;
    if (!(srf_debug_flags() & 128)) f.in_sums = a.in_norm.sums;
    srf_pyramid_finalize_launch(f, groups, C, st);
    SRF_CHECK_LAUNCH("pyramid_finalize", st);
    return srf_pyramid_reg_launch(r, false, rows, st);
  }
  PyrTile tile;
  if (!(srf_debug_flags() & 32) && pyr_pick_tile(L, D, &tile)) {
    tile.tasks = rows * tile.tiles;
    const size_t tl = (size_t)tile.wave_floats * 4 * sizeof(float);
    const unsigned nb = (unsigned)((tile.tasks + 3) / 4);
    SRF_CHECK_HIP(hipMemsetAsync(mom, 0, sizeof(double) * (size_t)rows * D * 5, st));
    if (a.in_norm.sums) {
      hipLaunchKernelGGL(srf_stats_finalize_kernel, dim3((unsigned)groups), dim3(64), 0, st, a.in_norm.sums,
                         a.in_inv_count, const_cast<float*>(a.in_mr));
      SRF_CHECK_LAUNCH("stats_finalize", st);
    }
    hipLaunchKernelGGL(srf_pyramid_tile_kernel<true>, dim3(nb), dim3(256), tl, st, a, tile);
    SRF_CHECK_LAUNCH("pyramid_moments", st);
    srf_pyramid_finalize_launch(f, groups, C, st);
    SRF_CHECK_LAUNCH("pyramid_finalize", st);
    hipLaunchKernelGGL(srf_pyramid_tile_kernel<false>, dim3(nb), dim3(256), tl, st, a, tile);
    SRF_CHECK_LAUNCH("pyramid_merge", st);
    return SRF_OK;
  }
  const size_t ldsb = pyr_lds_bytes(L, D);
  // (Fallback path -- shapes neither the register-resident nor the wave-per-tile kernels take.  No cached state: the
  // dynamic-LDS attribute is per device and the occupancy depends on L, so both are simply evaluated per call; a
  // function-static cache here was wrong for one thread per GPU)
  if (ldsb > 64 * 1024) {
    SRF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&srf_pyramid_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SRF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&srf_pyramid_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  // persistent grids: exactly as many blocks as are co-resident (registers + LDS), rows strided
  auto resident_blocks = [&](const void* fn) -> long {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, ldsb) != hipSuccess || per_cu < 1)
      per_cu = 1;
    const long n = (long)per_cu * srf_device_cus();
    return n < rows ? n : rows;
  };
  const long nblk1 = resident_blocks(reinterpret_cast<const void*>(&srf_pyramid_kernel<true>));
  const long nblk2 = resident_blocks(reinterpret_cast<const void*>(&srf_pyramid_kernel<false>));
  hipLaunchKernelGGL(srf_pyramid_kernel<true>, dim3((unsigned)nblk1), dim3(256), ldsb, st, a);
  SRF_CHECK_LAUNCH("pyramid_moments", st);
  srf_pyramid_finalize_launch(f, groups, C, st);
  SRF_CHECK_LAUNCH("pyramid_finalize", st);
  hipLaunchKernelGGL(srf_pyramid_kernel<false>, dim3((unsigned)nblk2), dim3(256), ldsb, st, a);
  SRF_CHECK_LAUNCH("pyramid_merge", st);
  return SRF_OK;
}
