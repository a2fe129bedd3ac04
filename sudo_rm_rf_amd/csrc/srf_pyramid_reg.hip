// Register-resident variant of the fused depthwise pyramid (see srf_pyramid.hip for the algebra and
// the three-step structure moments -> finalize -> merge).
//
// PMC counters of the LDS-tiled kernels showed them LDS-bound (LDS array busy ~80 % of the kernel, half
// of it bank conflicts of the stride-2 level reads) with ~600 VALU + ~380 SALU instructions per
// 640-sample tile.  Here every lane owns CH = 16 (D <= 5) or 32 (D <= 6) CONSECUTIVE time steps of a
// row, so a lane's level-k values (CH >> k of them) are exactly the ones its own outputs upsample
// from: the whole cascade and the bottom-up merge run in registers, the only cross-lane traffic is
// the 2-left / <=2-right conv halo per level (wavefront shuffles), and there is no LDS and no barrier.
// A wavefront covers 64 chunks = 2 halo chunks + up to 60 own chunks + 2 halo chunks (the dependency
// cone of the deepest level reaches 30 (62) level-0 samples = 2 chunks); halo lanes recompute what the
// neighbouring wavefront owns.  Global accesses are 4 (8) float4 per lane at a 64-B (128-B) lane
// stride: every byte of a line is used by the same wavefront, L1 merges the pieces.
//
// Round 2, measured and removed again (code in git history / the round's notes): TWO rows per lane -- channel c of
// examples g and g + Bt/2 -- so that every conv tap, mask, moment and merge is a packed-fp32 instruction on naturally aligned
// (row A, row B) register pairs (weights as scalar pairs broadcast through op_sel on src0, interleave free on load).
// Bit-compatible results, 39 % fewer VALU instructions per row in pass 2 and 29 % in pass 1 -- and no gain: pass 2 stayed
// at 83-84 us (it moves 419 MB at 5.0 TB/s: the read + write stream is what bounds it, not the VALU), pass 1 went from 61
// to 69 us (155 instead of 71 registers: 3 instead of 6 wavefronts per SIMD to cover the loads, and a v_pk_fma_f32 holds the
// SIMD ~5.5 cycles against 4.6 for a v_fma_f32, tools/probes/valu_rate_probe.hip).  What did help pass 1: building this
// file WITHOUT the SLP vectorizer (70 -> 62 us: its v_pk_fma_f32 came with a v_mov per operand pair), see build.py.
#include "srf_pyr.h"


template <int N>
__device__ __forceinline__ void srf_zero(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = 0.f;
}

// stride-1 k=5 conv over a lane's chunk: out[i] = b + sum_t w[t] * x[i + t - 2], halo from neighbours
template <int N>
__device__ __forceinline__ void srf_conv_s1(const float (&x)[N], float (&out)[N], const float* w, float b) {
  const float l2 = srf_lane_up(x[N - 2]), l1 = srf_lane_up(x[N - 1]);
  const float r0 = srf_lane_down(x[0]), r1 = srf_lane_down(x[1]);
  float e[N + 4];
  e[0] = l2;
  e[1] = l1;
#pragma unroll
  for (int i = 0; i < N; ++i) e[i + 2] = x[i];
  e[N + 2] = r0;
  e[N + 3] = r1;
#pragma unroll
  for (int i = 0; i < N; ++i)
    out[i] = fmaf(w[4], e[i + 4], fmaf(w[3], e[i + 3], fmaf(w[2], e[i + 2], fmaf(w[1], e[i + 1], fmaf(w[0], e[i], b)))));
}

// stride-2 k=5 conv: out[j] = b + sum_t w[t] * x[2j + t - 2]
template <int N>
__device__ __forceinline__ void srf_conv_s2(const float (&x)[N], float (&out)[N / 2], const float* w, float b) {
  const float l2 = srf_lane_up(x[N - 2]), l1 = srf_lane_up(x[N - 1]);
  const float r0 = srf_lane_down(x[0]);
  float e[N + 3];
  e[0] = l2;
  e[1] = l1;
#pragma unroll
  for (int i = 0; i < N; ++i) e[i + 2] = x[i];
  e[N + 2] = r0;
#pragma unroll
  for (int j = 0; j < N / 2; ++j)
    out[j] = fmaf(w[4], e[2 * j + 4],
                  fmaf(w[3], e[2 * j + 3], fmaf(w[2], e[2 * j + 2], fmaf(w[1], e[2 * j + 1], fmaf(w[0], e[2 * j], b)))));
}

// v <- valid ? a v + c : 0.  The mask goes into the coefficients (two selects per call) instead of one select per
// element: lanes outside the row compute 0 * v + 0 (v is finite: it comes from clamped loads of real data), lanes
// inside compute exactly fmaf(v, a, c) as before.  Both pyramid kernels are VALU-bound; the per-element selects were
// 15 % of their VALU instructions.
template <int N>
__device__ __forceinline__ void srf_affine_mask(float (&v)[N], float a, float c, bool valid) {
  const float am = valid ? a : 0.f, cm = valid ? c : 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = fmaf(v[i], am, cm);
}

template <int N>
__device__ __forceinline__ void srf_acc_moments(const float (&v)[N], float& s, float& q) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    s += v[i];
    q = fmaf(v[i], v[i], q);
  }
}

struct LvCoef {
  float w[5], b, a, c;
};

// pass-1 edge bookkeeping: level-k values at row positions 0, 1 and L_k - 1 (N = values per lane)
template <int N>
__device__ __forceinline__ void srf_pyr_edges(double* m5, const float (&v)[N], int ci, int nchunks) {
  if (ci == 0) {
    m5[2] = (double)v[0];
    if (N >= 2) m5[3] = (double)v[N >= 2 ? 1 : 0];
  }
  if (N == 1 && ci == 1) m5[3] = (double)v[0];
  if (ci == nchunks - 1) m5[4] = (double)v[N - 1];
}

// Own chunk j = lane - 2 of a wavefront -> global row, through the wave-private strip: chunk j sits at float4 offset
// j*(NF+1) (odd pitch in 16-B units: conflict-free both ways), then every store instruction writes 64 consecutive
// float4 = 1 KB.  Ends with a wavefront barrier so the strip can be reused.
template <int CH>
__device__ __forceinline__ void srf_pyr_store_chunks(float4* strip, const float (&v)[CH], bool own, float* row_base,
                                                     int first_chunk, int cnt, int lane) {
  constexpr int NF = CH / 4, PF = NF + 1;
  const int j = lane - 2;
  if (own) {
#pragma unroll
    for (int i = 0; i < NF; ++i) strip[j * PF + i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float4* dst = reinterpret_cast<float4*>(row_base + (size_t)first_chunk * CH);
#pragma unroll
  for (int it = 0; it < (60 * NF + 63) / 64; ++it) {
    const int f = it * 64 + lane;
    if (f < cnt * NF) dst[f] = strip[(f / NF) * PF + (f % NF)];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a lane's N consecutive values of a deeper level (N * 4 bytes, naturally aligned: neighbouring lanes are contiguous)
template <int N>
__device__ __forceinline__ void srf_pyr_store_level(float* dst, const float (&v)[N]) {
  if constexpr (N >= 4) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i)
      reinterpret_cast<float4*>(dst)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else if constexpr (N == 2) {
    *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
  } else {
    dst[0] = v[0];
  }
}

// MOMENTS: pass 1 (y1 -> raw cascade -> row moments; writes nothing else).  !MOMENTS: pass 2 (y1 -> level 0
// again, normalised cascade -> merged).  Recomputing level 0 (5 MAC/elem) is cheaper than the d0 round trip
// through HBM that an earlier version made (pass 1 measured 138 us with the d0 store, 81 us without).
// PERSIST: wavefronts loop over a contiguous task range with next-task prefetch (measured faster for
// pass 1, slower for pass 2 whose one-task-per-wave form keeps 8 waves per SIMD resident).
// SAVE (pass 2 only): also write every level's raw conv output d_k -- what the training backward needs -- so the
// training forward is the same two fused passes instead of D depthwise kernels + a merge kernel (7.75 -> 4.94 C*L of
// traffic per block).
template <bool MOMENTS, int CH, bool PERSIST = MOMENTS, bool SAVE = false>
// (no occupancy attribute: pinning 6 wavefronts per SIMD made the persistent pass 1 spill VGPRs to scratch memory)
__global__ __launch_bounds__(256) void srf_pyramid_reg_kernel(PyrRegArgs a) {
  __shared__ float4 pyr_strip[MOMENTS ? 1 : 4 * 60 * (CH / 4 + 1)];   // pass 2: store transposition
  const int lane = threadIdx.x & 63;
  const int L = a.L, D = a.D, C = a.C;
  const int nchunks = L / CH;
  // Task = (row, tile).  All index arithmetic is 32-bit and division-free inside the loop: the earlier
  // flat 64-bit task index cost ~300 scalar instructions per task in divisions and 64-bit multiplies
  // (the scalar unit is shared by the CU's four SIMDs).
  //  * persistent (pass 1): wave w owns the whole rows [w*rpw, (w+1)*rpw) -- the row moments stay in
  //    registers across the row's tiles -- and keeps the NEXT task's input chunk in flight while it
  //    computes the current one;
  //  * one task per wave (pass 2): grid = (C, ceil(tiles/4), groups), wave = tile within the group of 4.
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int row, row_end, tile, g, c;
  if (PERSIST) {
    row = (blockIdx.x * 4 + wave) * a.rpw;
    if (row >= a.rows) return;   // wave-uniform
    row_end = min(row + a.rpw, a.rows);
    g = row / C;
    c = row - g * C;
    tile = 0;
  } else {
    c = blockIdx.x;
    g = blockIdx.z;
    row = g * C + c;
    row_end = row + 1;
    tile = blockIdx.y * 4 + wave;
    if (tile >= a.tiles) return;   // wave-uniform
  }
  const float* in_base = a.y1;   // both passes start from y1: pass 2 recomputes level 0 instead of re-reading it
  auto chunk_src = [&](int row_, int tile_) {
    const int ci_ = min(max(tile_ * a.own - 2 + lane, 0), nchunks - 1);   // clamped: loads unconditional
    return reinterpret_cast<const float4*>(in_base + (size_t)row_ * L + ci_ * CH);
  };
  float s1[6], s2[6];
  int cur_g = -1;                // persistent pass 1 finalises the proj statistics itself, once per example
  float in_mean = 0.f, in_rstd = 1.f;
  float4 pre[CH / 4];
  {
    const float4* src = chunk_src(row, tile);
#pragma unroll
    for (int i = 0; i < CH / 4; ++i) pre[i] = src[i];
  }
  for (;;) {
  // the task after this one (persistent only)
  int nrow = row, ntile = tile + 1;
  if (ntile == a.tiles) {
    ntile = 0;
    nrow = row + 1;
  }
  const bool last_task = !PERSIST || nrow >= row_end;
  const int ci = tile * a.own - 2 + lane;          // this lane's chunk index in the row
  const bool valid = ci >= 0 && ci < nchunks;      // inside the row (else: zero padding)
  const bool own = valid && lane >= 2 && lane < 2 + a.own;

  // per-level coefficients (wave-uniform -> scalar loads, all issued up front)
  LvCoef lc[SRF_MAX_DEPTH];
#pragma unroll
  for (int k = 0; k < SRF_MAX_DEPTH; ++k) {
    if (k < D && (CH >> k) >= 1) {
#pragma unroll
      for (int t = 0; t < 5; ++t) lc[k].w[t] = a.w[k][c * 5 + t];
      lc[k].b = (MOMENTS && k > 0) ? 0.f : a.bias[k][c];
      if (MOMENTS) {
        lc[k].a = 1.f;
        lc[k].c = 0.f;
      } else {
        const float m = a.lvl[(g * D + k) * 2 + 0], r = a.lvl[(g * D + k) * 2 + 1];
        lc[k].a = a.gamma[k][c] * r;
        lc[k].c = a.beta[k][c] - m * lc[k].a;
      }
    }
  }

  float x0[CH];
  {
    // ---- o = PReLU(GlobLN(y1)), then the level-0 conv (+ bias)
    float sc = 1.f, sh = 0.f;
    if (a.in_norm.sums) {
      if (PERSIST) {
        if (g != cur_g) {   // wave-uniform
          srf_finalize_stats(a.in_norm.sums, g, a.in_inv_count, in_mean, in_rstd);
          cur_g = g;
        }
      } else {
        in_mean = a.in_mr[2 * g];
        in_rstd = a.in_mr[2 * g + 1];
      }
      sc = a.in_norm.gamma[c] * in_rstd;
      sh = a.in_norm.beta[c] - in_mean * sc;
    }
    const bool act = a.in_norm.prelu != nullptr;
    const float slope = act ? a.in_norm.prelu[0] : 1.f;
    float o[CH];
#pragma unroll
    for (int i = 0; i < CH / 4; ++i) {
      const float4 v = pre[i];
      o[4 * i + 0] = v.x;
      o[4 * i + 1] = v.y;
      o[4 * i + 2] = v.z;
      o[4 * i + 3] = v.w;
    }
    const float scm = valid ? sc : 0.f, shm = valid ? sh : 0.f;   // mask in the coefficients: PReLU(0) = 0
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      float v = fmaf(o[i], scm, shm);
      if (act) v = srf_prelu(v, slope);
      o[i] = v;
    }
    srf_conv_s1<CH>(o, x0, lc[0].w, lc[0].b);
    if (SAVE && a.lv_out[0])      // (kernel-uniform; null = level 0 is not kept)
      srf_pyr_store_chunks<CH>(pyr_strip + (threadIdx.x >> 6) * (60 * (CH / 4 + 1)), x0, own, a.lv_out[0] + (size_t)row * L,
                               tile * a.own, min(a.own, nchunks - tile * a.own), lane);
    const float vmask = valid ? 1.f : 0.f;
    if (MOMENTS) {
#pragma unroll
      for (int i = 0; i < CH; ++i) x0[i] *= vmask;
    } else {
      srf_affine_mask<CH>(x0, lc[0].a, lc[0].c, valid);   // n_0 = GlobLN_0(d_0)
    }
  }

  if (PERSIST) {  // prefetch the next task's chunk (the last task re-reads its own: surplus loads are harmless)
    const float4* src = chunk_src(last_task ? row : nrow, last_task ? tile : ntile);
#pragma unroll
    for (int i = 0; i < CH / 4; ++i) pre[i] = src[i];
  }

  // ---- levels 1..D-1 in registers (each level halves the per-lane count)
  float x1[CH / 2], x2[CH / 4], x3[CH / 8], x4[CH / 16], x5[CH / 32 > 0 ? CH / 32 : 1];
  srf_zero(x1);
  srf_zero(x2);
  srf_zero(x3);
  srf_zero(x4);
  srf_zero(x5);
  // pass 1, persistent: a wave owns whole rows, so the row moments are
  // accumulated in registers over the row's tiles and written with plain stores at its last tile
  if (!PERSIST || tile == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) s1[k] = s2[k] = 0.f;
  }
  double* mrow = a.mom + (size_t)row * D * 5;
  if (MOMENTS && own) {
    srf_acc_moments<CH>(x0, s1[0], s2[0]);
    srf_pyr_edges<CH>(mrow, x0, ci, nchunks);
  }
  if (D > 1) {
    srf_conv_s2<CH>(x0, x1, lc[1].w, lc[1].b);
    if (SAVE && own) srf_pyr_store_level<CH / 2>(a.lv_out[1] + (size_t)row * (L >> 1) + (size_t)ci * (CH / 2), x1);
    srf_affine_mask<CH / 2>(x1, lc[1].a, lc[1].c, valid);
    if (MOMENTS && own) {
      srf_acc_moments<CH / 2>(x1, s1[1], s2[1]);
      srf_pyr_edges<CH / 2>(mrow + 5, x1, ci, nchunks);
    }
  }
  if (D > 2) {
    srf_conv_s2<CH / 2>(x1, x2, lc[2].w, lc[2].b);
    if (SAVE && own) srf_pyr_store_level<CH / 4>(a.lv_out[2] + (size_t)row * (L >> 2) + (size_t)ci * (CH / 4), x2);
    srf_affine_mask<CH / 4>(x2, lc[2].a, lc[2].c, valid);
    if (MOMENTS && own) {
      srf_acc_moments<CH / 4>(x2, s1[2], s2[2]);
      srf_pyr_edges<CH / 4>(mrow + 10, x2, ci, nchunks);
    }
  }
  if (D > 3) {
    srf_conv_s2<CH / 4>(x2, x3, lc[3].w, lc[3].b);
    if (SAVE && own) srf_pyr_store_level<CH / 8>(a.lv_out[3] + (size_t)row * (L >> 3) + (size_t)ci * (CH / 8), x3);
    srf_affine_mask<CH / 8>(x3, lc[3].a, lc[3].c, valid);
    if (MOMENTS && own) {
      srf_acc_moments<CH / 8>(x3, s1[3], s2[3]);
      srf_pyr_edges<CH / 8>(mrow + 15, x3, ci, nchunks);
    }
  }
  if (D > 4) {
    srf_conv_s2<CH / 8>(x3, x4, lc[4].w, lc[4].b);
    if (SAVE && own) srf_pyr_store_level<CH / 16>(a.lv_out[4] + (size_t)row * (L >> 4) + (size_t)ci * (CH / 16), x4);
    srf_affine_mask<CH / 16>(x4, lc[4].a, lc[4].c, valid);
    if (MOMENTS && own) {
      srf_acc_moments<CH / 16>(x4, s1[4], s2[4]);
      srf_pyr_edges<CH / 16>(mrow + 20, x4, ci, nchunks);
    }
  }
  if constexpr (CH >= 32) {
    if (D > 5) {
      srf_conv_s2<CH / 16>(x4, x5, lc[5].w, lc[5].b);
      if (SAVE && own) srf_pyr_store_level<CH / 32>(a.lv_out[5] + (size_t)row * (L >> 5) + (size_t)ci * (CH / 32), x5);
      srf_affine_mask<CH / 32>(x5, lc[5].a, lc[5].c, valid);
      if (MOMENTS && own) {
        srf_acc_moments<CH / 32>(x5, s1[5], s2[5]);
        srf_pyr_edges<CH / 32>(mrow + 25, x5, ci, nchunks);
      }
    }
  }

  if (MOMENTS) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (PERSIST && tile != a.tiles - 1) break;   // the row's last tile flushes
      if (k < D) {
        const float r1 = srf_dpp_wave_sum(s1[k]);
        const float r2 = srf_dpp_wave_sum(s2[k]);
        if (lane == 63) {
          if (PERSIST) {
            mrow[k * 5 + 0] = (double)r1;
            mrow[k * 5 + 1] = (double)r2;
          } else {
            atomicAdd(&mrow[k * 5 + 0], (double)r1);
            atomicAdd(&mrow[k * 5 + 1], (double)r2);
          }
        }
      }
    }
  } else {
    // ---- merged[i] = n0[i] + (n1[i>>1] + (n2[i>>2] + ...))  -- everything is lane-local
    float ms = 0.f, mq = 0.f;
    float outv[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      float t = 0.f;
      bool have = false;
      if (D > 5 && CH >= 32) {
        t = x5[i >> 5];
        have = true;
      }
      if (D > 4) {
        const float v = x4[i >> 4];
        t = have ? v + t : v;
        have = true;
      }
      if (D > 3) {
        const float v = x3[i >> 3];
        t = have ? v + t : v;
        have = true;
      }
      if (D > 2) {
        const float v = x2[i >> 2];
        t = have ? v + t : v;
        have = true;
      }
      if (D > 1) {
        const float v = x1[i >> 1];
        t = have ? v + t : v;
        have = true;
      }
      const float v = have ? x0[i] + t : x0[i];
      outv[i] = v;
      ms += v;
      mq = fmaf(v, v, mq);
    }
    // Store through a wave-private LDS strip so that every store instruction writes 1 KB of consecutive
    // addresses (a lane's own 64/128 B at a 64/128-B lane stride measured 17-22 us slower per launch).
    if (!own) {
      ms = 0.f;
      mq = 0.f;
    }
    srf_pyr_store_chunks<CH>(pyr_strip + wave * (60 * (CH / 4 + 1)), outv, own, a.merged + (size_t)row * L, tile * a.own,
                             min(a.own, nchunks - tile * a.own), lane);
    if (a.out_sums) {
      // (DPP wave sums: VALU only; the __shfl_xor form is 24 ds_bpermute per tile on every wavefront's critical path)
      const double ds = srf_dpp_wave_sum((double)ms), dq = srf_dpp_wave_sum((double)mq);
      if (lane == 63) {
        double* dst = srf_stat_slot(a.out_sums, g, c * a.tiles + tile);
        atomicAdd(dst, ds);
        atomicAdd(dst + 1, dq);
      }
    }
  }
  if (last_task) break;
  if (ntile == 0 && ++c == C) {
    c = 0;
    ++g;
  }
  row = nrow;
  tile = ntile;
  }  // task loop
}

bool srf_pyramid_reg_supported(int L, int D) {
  if (D <= 5) return L % 16 == 0 && L / 16 >= 4;
  if (D == 6) return L % 32 == 0 && L / 32 >= 4;
  return false;
}

// moments / finalize / merge launches are driven by srf_pyramid() in srf_pyramid.hip
int srf_pyramid_reg_launch(PyrRegArgs a, bool moments, long rows, hipStream_t st) {
  const int CH = a.D <= 5 ? 16 : 32;
  const int nchunks = a.L / CH;
  a.tiles = (nchunks + 59) / 60;
  a.own = (nchunks + a.tiles - 1) / a.tiles;
  SRF_CHECK_ARG(rows * a.tiles < (1L << 31) && rows / a.C <= 65535, "srf_pyramid: too many rows");
  a.rows = (int)rows;
  // pass 1 persistent (grid = co-resident wavefronts, cached occupancy query) unless debug flag 128
  const bool persist = moments && !(srf_debug_flags() & 128);
  // co-resident wavefronts of the persistent pass 1 on THIS device (per-device cache, srf_common.h)
  long cw = 0;
  if (persist) {
    struct Q { int ch; } q{CH};
    cw = srf_device_cached(CH == 16 ? 0 : 1, [](void* p) -> long {
      const int ch = static_cast<Q*>(p)->ch;
      int per_cu = 0;
      const void* fn = ch == 16 ? (const void*)&srf_pyramid_reg_kernel<true, 16, true>
                                : (const void*)&srf_pyramid_reg_kernel<true, 32, true>;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
      return (long)srf_device_cus() * per_cu * 4;
    }, &q);
  }
  dim3 grid;
  if (persist) {
    // Round 6: ONE row per wavefront (the row's moments still stay in registers over its tiles), i.e. rows / 4 blocks whatever
    // the number of co-resident wavefronts.  Rounds 2-5 sized the grid to the co-resident wavefronts (3 rows per wave at cfg 2:
    // 5.3 blocks per CU); pass 1 is bound by the loads it has in flight, not by its VALU count (profiles/r06_pyramid_pass1_*):
    // 1 / 2 / 3 / 4 rows per wave = 54.5 / 55.5 / 61 / 62.8 us at cfg 2, same box.  Debug flag 1 << 17: the old sizing.
    long nwaves = cw < rows ? cw : rows;
    a.rpw = (srf_debug_flags() & (1 << 17)) ? (int)((rows + nwaves - 1) / nwaves) : 1;   // whole rows per wave
    nwaves = (rows + a.rpw - 1) / a.rpw;
    grid = dim3((unsigned)((nwaves + 3) / 4));
  } else {
    a.rpw = 1;
    grid = dim3((unsigned)a.C, (unsigned)((a.tiles + 3) / 4), (unsigned)(rows / a.C));
  }
  const bool save = !moments && a.save != 0;
  if (save)
    for (int k = 0; k < a.D; ++k)
      SRF_CHECK_ARG((a.lv_out[k] || (k == 0 && a.D > 1)) && srf_aligned16(a.lv_out[k]), "srf_pyramid: level output %d missing / unaligned", k);
  if (CH == 16) {
    if (moments && persist)
      hipLaunchKernelGGL((srf_pyramid_reg_kernel<true, 16, true>), grid, dim3(256), 0, st, a);
    else if (moments)
      hipLaunchKernelGGL((srf_pyramid_reg_kernel<true, 16, false>), grid, dim3(256), 0, st, a);
    else if (save)
      hipLaunchKernelGGL((srf_pyramid_reg_kernel<false, 16, false, true>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((srf_pyramid_reg_kernel<false, 16, false>), grid, dim3(256), 0, st, a);
  } else {
    if (moments && persist)
      hipLaunchKernelGGL((srf_pyramid_reg_kernel<true, 32, true>), grid, dim3(256), 0, st, a);
    else if (moments)
      hipLaunchKernelGGL((srf_pyramid_reg_kernel<true, 32, false>), grid, dim3(256), 0, st, a);
    else if (save)
      hipLaunchKernelGGL((srf_pyramid_reg_kernel<false, 32, false, true>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((srf_pyramid_reg_kernel<false, 32, false>), grid, dim3(256), 0, st, a);
  }
  SRF_CHECK_LAUNCH(moments ? "pyramid_moments" : (save ? "pyramid_merge_save" : "pyramid_merge"), st);
  return SRF_OK;
}
