// Backward kernels of the streaming ops (training step).  Each mirrors one autograd node
// of the reference's forward graph; the per-kernel entry points exist for unit parity against torch autograd.
//   GlobLN (+PReLU) backward        improved_sudormrf.py:30-47 (+ :13-27 affine, PReLU of ConvNormAct/NormAct)
//   Upsample(x2 nearest)+add merge  improved_sudormrf.py:190-194,214-216
//   depthwise k=5 conv (s=1|2)      improved_sudormrf.py:138-159,178-189
//   mask ReLU * encoder output      improved_sudormrf.py:296-298
//   conv_transpose1d / conv1d frame gathers for the decoder / encoder weight gradients  :247-251,272-279
#include "srf_common.h"

// The streamed operands of the row kernels below (gradients, saved activations: each read once per kernel) as non-temporal loads
// (round 4; -DSRF_BWD_NT=false for the A/B build)
#ifndef SRF_BWD_NT
#define SRF_BWD_NT true
#endif
#include <algorithm>
#include <vector>

// =============================================================================================
// GlobLN (+PReLU) backward.   z = gamma_c * xh + beta_c, xh = (x - mean_b) * rstd_b, out = PReLU_a(z)
//   g_z   = g_out * (z >= 0 ? 1 : a)                 d a    += sum g_out * z [z < 0]
//   d gamma_c += sum_{b,t} g_z xh                    d beta_c += sum_{b,t} g_z
//   g_x   = rstd_b * (gamma_c g_z - S1_b / n - xh * S2_b / n),   S1_b = sum_{c,t} gamma_c g_z,  S2_b = sum gamma_c g_z xh
// Pass A: one block per row (b, c): {sum g_z, sum g_z xh, slope term} -> rowpart (plain stores) and S1/S2 into
// the per-example fp64 buckets (same bucketed layout as the forward statistics: no same-address atomics).
// Pass A2: per-channel reduction of rowpart over the batch.  Pass B: apply.
// =============================================================================================
struct GlnBwdArgs {
  const float* gout;
  const float* gout2;   // optional second gradient contribution, added on load
  const float* x;
  SrfNormDev nrm;
  double inv_count;
  float* rowpart;       // [rows][4]
  double* bsums;        // [groups][SRF_STAT_BUCKETS][2]  {S1, S2}
  float* gx;
  int C, L, accumulate;
  // apply pass only (srf_gln_bwd_apply_v4_kernel): the merge backward folded in -- gx is then g_merged = g_n_0 and level k's
  // gradient g_n_k[j] = g_n_{k-1}[2j] + g_n_{k-1}[2j+1] goes to mlv[k] ([rows, L >> k], k = 1 .. mD - 1); mD <= 1 = off
  float* mlv[SRF_MAX_DEPTH] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int mD = 0;
};

__global__ __launch_bounds__(256) void srf_gln_bwd_reduce_kernel(GlnBwdArgs a) {
  __shared__ float red[4][3];
  const long row = blockIdx.x;
  const int c = (int)(row % a.C);
  const long g = row / a.C;
  float mean, rstd;
  srf_finalize_stats(a.nrm.sums, g, a.inv_count, mean, rstd);
  const float gam = a.nrm.gamma[c], bet = a.nrm.beta[c];
  const bool act = a.nrm.prelu != nullptr;
  const float slope = act ? a.nrm.prelu[0] : 1.f;
  const float* go = a.gout + row * (long)a.L;
  const float* go2 = a.gout2 ? a.gout2 + row * (long)a.L : nullptr;
  const float* xr = a.x + row * (long)a.L;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int l = threadIdx.x; l < a.L; l += 256) {
    float gv = go[l];
    if (go2) gv += go2[l];
    const float xh = (xr[l] - mean) * rstd;
    const float z = fmaf(gam, xh, bet);
    float gz = gv;
    if (act && z < 0.f) {
      s2 = fmaf(gv, z, s2);
      gz = gv * slope;
    }
    s0 += gz;
    s1 = fmaf(gz, xh, s1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    red[w][0] = s0;
    red[w][1] = s1;
    red[w][2] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t0 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    const float t1 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const float t2 = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    float* rp = a.rowpart + row * 4;
    rp[0] = t0;
    rp[1] = t1;
    rp[2] = t2;
    double* dst = srf_stat_slot(a.bsums, g, c);
    atomicAdd(dst, (double)gam * (double)t0);
    atomicAdd(dst + 1, (double)gam * (double)t1);
  }
}

// Sums of the per-row partials over the batch.  Block = 32 channels x 8 group slices over a chunk of 64 groups
// (GroupComm folds 512 groups onto 16 channels: one thread per channel walking all groups took 130 us); results are
// atomically added to the parameter gradients.
// GlobLN: rowpart[.][0] -> dbeta, [1] -> dgamma, [2] -> slope (all channels into one scalar)
// ---- deferred parameter-gradient reductions (round 3) --------------------------------------------------------------------
// Every GlobLN / depthwise-conv backward ends with a tiny kernel that folds its per-row partials into the parameter gradients:
// ~190 launches of ~5 us per cfg-2 training step, each in the dependent chain of the stream.  When the caller gives every call
// its OWN scratch slice (srf_backward does, for the blocks' norms and convs), the partials stay valid, the call
// only records a descriptor, and srf_bwd_ctx_flush() folds them all in a handful of batched launches at the end of the backward.
struct GlnParamsDesc {
  const float* rowpart;
  float *dgamma, *dbeta, *dslope;
  int groups, C;
};
struct DwParamsDesc {
  const float* rowpart;
  float *dw, *dbias;
  int groups, C;
};
constexpr int SRF_PB_MAX = 40;
struct GlnParamsTable {
  GlnParamsDesc d[SRF_PB_MAX];
};
struct DwParamsTable {
  DwParamsDesc d[SRF_PB_MAX];
};
// What one srf_backward call carries across its kernel-level calls: the deferred reductions above, and the one-shot request
// "fold the merge backward into the NEXT GlobLN backward's apply pass".  An explicit object owned by the caller (rounds 3-4
// kept it in thread_local variables); the public per-kernel entry points pass none.
struct SrfBwdCtx {
  bool defer = false;
  std::vector<GlnParamsDesc> gln;
  std::vector<DwParamsDesc> dw;
  float* merge_lv[SRF_MAX_DEPTH] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int merge_D = 0;
  bool merge_taken = false;
};
SrfBwdCtx* srf_bwd_ctx_new() { return new SrfBwdCtx; }
void srf_bwd_ctx_free(SrfBwdCtx* c) { delete c; }
void srf_bwd_ctx_defer(SrfBwdCtx* c, bool on) { c->defer = on; }

__device__ __forceinline__ void srf_gln_bwd_params_body(const float* __restrict__ rowpart, int groups, int C, float* dgamma,
                                                        float* dbeta, float* dslope) {
  __shared__ float red[8][32][3];
  const int cl = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int g0 = blockIdx.y * 64, g1 = min(g0 + 64, groups);
  float sb = 0.f, sg = 0.f, ss = 0.f;
  if (c < C) {
    for (int g = g0 + slice; g < g1; g += 8) {
      const float* rp = rowpart + ((long)g * C + c) * 4;
      sb += rp[0];
      sg += rp[1];
      ss += rp[2];
    }
  }
  red[slice][cl][0] = sb;
  red[slice][cl][1] = sg;
  red[slice][cl][2] = ss;
  __syncthreads();
  if (slice == 0) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
      t0 += red[sidx][cl][0];
      t1 += red[sidx][cl][1];
      t2 += red[sidx][cl][2];
    }
    if (c < C) {
      if (dbeta) atomicAdd(dbeta + c, t0);
      if (dgamma) atomicAdd(dgamma + c, t1);
    }
    if (dslope) {
      // the 32 channels of this block -> one atomic (lanes 0..31 of wavefront 0; lanes 32..63 hold slice 1: excluded)
      t2 = c < C ? t2 : 0.f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) t2 += __shfl_xor(t2, o, 64);
      if (cl == 0 && t2 != 0.f) atomicAdd(dslope, t2);
    }
  }
}

__global__ __launch_bounds__(256) void srf_gln_bwd_params_kernel(const float* __restrict__ rowpart, int groups, int C,
                                                                 float* dgamma, float* dbeta, float* dslope) {
  srf_gln_bwd_params_body(rowpart, groups, C, dgamma, dbeta, dslope);
}
__global__ __launch_bounds__(256) void srf_gln_bwd_params_batch_kernel(GlnParamsTable t) {
  const GlnParamsDesc d = t.d[blockIdx.z];
  if ((int)blockIdx.x * 32 >= d.C || (int)blockIdx.y * 64 >= d.groups) return;   // (block-uniform)
  srf_gln_bwd_params_body(d.rowpart, d.groups, d.C, d.dgamma, d.dbeta, d.dslope);
}

__global__ __launch_bounds__(256) void srf_gln_bwd_apply_kernel(GlnBwdArgs a, int chunks) {
  const long row = blockIdx.x / chunks;
  const int chunk = blockIdx.x - row * chunks;
  const int c = (int)(row % a.C);
  const long g = row / a.C;
  float mean, rstd;
  srf_finalize_stats(a.nrm.sums, g, a.inv_count, mean, rstd);
  // S1, S2 from the buckets (same access pattern as srf_finalize_stats)
  const double2 bk = reinterpret_cast<const double2*>(a.bsums)[g * SRF_STAT_BUCKETS + (threadIdx.x & (SRF_STAT_BUCKETS - 1))];
  const float m1 = (float)(srf_wave_sum(bk.x) * a.inv_count), m2 = (float)(srf_wave_sum(bk.y) * a.inv_count);
  const float gam = a.nrm.gamma[c], bet = a.nrm.beta[c];
  const bool act = a.nrm.prelu != nullptr;
  const float slope = act ? a.nrm.prelu[0] : 1.f;
  const size_t base = (size_t)row * a.L;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int l = chunk * 1024 + u * 256 + threadIdx.x;
    if (l < a.L) {
      float gv = a.gout[base + l];
      if (a.gout2) gv += a.gout2[base + l];
      const float xh = (a.x[base + l] - mean) * rstd;
      const float z = fmaf(gam, xh, bet);
      const float gz = (act && z < 0.f) ? gv * slope : gv;
      float r = rstd * (gam * gz - m1 - xh * m2);
      if (a.accumulate) r += a.gx[base + l];
      a.gx[base + l] = r;
    }
  }
}

// Fast variants (L % 4 == 0, 16-byte aligned tensors): ONE WAVEFRONT PER ROW, 16-byte accesses, four float4 per lane
// and tensor in flight per trip, the row sums reduced with DPP (no LDS, no barrier).  The dword kernels above ran at
// 2.7 (reduce) / 3.1 TB/s (apply) on cfg 2 and remain the fallback for odd lengths.
__device__ __forceinline__ void srf_gln_bwd_elem(float gv, float x, float mean, float rstd, float gam, float bet,
                                                 bool act, float slope, float& xh, float& gz, float& sneg) {
  xh = (x - mean) * rstd;
  const float z = fmaf(gam, xh, bet);
  const bool neg = act && z < 0.f;
  sneg = fmaf(neg ? gv : 0.f, z, sneg);
  gz = neg ? gv * slope : gv;
}

__global__ __launch_bounds__(256) void srf_gln_bwd_reduce_v4_kernel(GlnBwdArgs a, long rows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;   // wave-uniform
  const int c = (int)(row % a.C);
  const long g = row / a.C;
  const int L4 = a.L >> 2;
  const float4* go = reinterpret_cast<const float4*>(a.gout + row * (long)a.L);
  const float4* go2 = a.gout2 ? reinterpret_cast<const float4*>(a.gout2 + row * (long)a.L) : nullptr;
  const float4* xr = reinterpret_cast<const float4*>(a.x + row * (long)a.L);
  float mean, rstd;
  srf_finalize_stats(a.nrm.sums, g, a.inv_count, mean, rstd);
  const float gam = a.nrm.gamma[c], bet = a.nrm.beta[c];
  const bool act = a.nrm.prelu != nullptr;
  const float slope = act ? a.nrm.prelu[0] : 1.f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int f0 = 0; f0 < L4; f0 += 256) {
    float4 gv[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int fc = min(f0 + u * 64 + lane, L4 - 1);   // clamped: loads unconditional
      gv[u] = srf_ld4<SRF_BWD_NT>(reinterpret_cast<const float*>(go + fc));
      xv[u] = srf_ld4<SRF_BWD_NT>(reinterpret_cast<const float*>(xr + fc));
    }
    if (go2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 t = srf_ld4<SRF_BWD_NT>(reinterpret_cast<const float*>(go2 + min(f0 + u * 64 + lane, L4 - 1)));
        gv[u].x += t.x;
        gv[u].y += t.y;
        gv[u].z += t.z;
        gv[u].w += t.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = f0 + u * 64 + lane < L4;
      const float ge[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
      const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xh, gz;
        srf_gln_bwd_elem(ok ? ge[e] : 0.f, xe[e], mean, rstd, gam, bet, act, slope, xh, gz, s2);
        s0 += gz;
        s1 = fmaf(gz, xh, s1);
      }
    }
  }
  s0 = srf_dpp_wave_sum(s0);
  s1 = srf_dpp_wave_sum(s1);
  s2 = srf_dpp_wave_sum(s2);
  if (lane == 63) {
    float* rp = a.rowpart + row * 4;
    rp[0] = s0;
    rp[1] = s1;
    rp[2] = s2;
    double* dst = srf_stat_slot(a.bsums, g, c);
    atomicAdd(dst, (double)gam * (double)s0);
    atomicAdd(dst + 1, (double)gam * (double)s1);
  }
}

__global__ __launch_bounds__(256) void srf_gln_bwd_apply_v4_kernel(GlnBwdArgs a, long rows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;   // wave-uniform
  const int c = (int)(row % a.C);
  const long g = row / a.C;
  const int L4 = a.L >> 2;
  const float4* go = reinterpret_cast<const float4*>(a.gout + row * (long)a.L);
  const float4* go2 = a.gout2 ? reinterpret_cast<const float4*>(a.gout2 + row * (long)a.L) : nullptr;
  const float4* xr = reinterpret_cast<const float4*>(a.x + row * (long)a.L);
  float4* gx = reinterpret_cast<float4*>(a.gx + row * (long)a.L);
  // the first trip's operands are requested before the statistics are finalised
  float4 gv[4], xv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int fc = min(u * 64 + lane, L4 - 1);
    gv[u] = srf_ld4<SRF_BWD_NT>(reinterpret_cast<const float*>(go + fc));
    xv[u] = srf_ld4<SRF_BWD_NT>(reinterpret_cast<const float*>(xr + fc));
  }
  float mean, rstd;
  srf_finalize_stats(a.nrm.sums, g, a.inv_count, mean, rstd);
  const double2 bk = reinterpret_cast<const double2*>(a.bsums)[g * SRF_STAT_BUCKETS + lane];
  const float m1 = (float)(srf_wave_sum(bk.x) * a.inv_count), m2 = (float)(srf_wave_sum(bk.y) * a.inv_count);
  const float gam = a.nrm.gamma[c], bet = a.nrm.beta[c];
  const bool act = a.nrm.prelu != nullptr;
  const float slope = act ? a.nrm.prelu[0] : 1.f;
  for (int f0 = 0; f0 < L4; f0 += 256) {
    if (f0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int fc = min(f0 + u * 64 + lane, L4 - 1);
        gv[u] = srf_ld4<SRF_BWD_NT>(reinterpret_cast<const float*>(go + fc));
        xv[u] = srf_ld4<SRF_BWD_NT>(reinterpret_cast<const float*>(xr + fc));
      }
    }
    if (go2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 t = go2[min(f0 + u * 64 + lane, L4 - 1)];
        gv[u].x += t.x;
        gv[u].y += t.y;
        gv[u].z += t.z;
        gv[u].w += t.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = f0 + u * 64 + lane;
      if (f < L4) {
        const float ge[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
        const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xh, gz, unused = 0.f;
          srf_gln_bwd_elem(ge[e], xe[e], mean, rstd, gam, bet, act, slope, xh, gz, unused);
          r[e] = rstd * (gam * gz - m1 - xh * m2);
        }
        float4 o = make_float4(r[0], r[1], r[2], r[3]);
        if (a.accumulate) {
          const float4 t = gx[f];
          o.x += t.x;
          o.y += t.y;
          o.z += t.z;
          o.w += t.w;
        }
        gx[f] = o;
        // ---- merge backward on the way out (round 3): the chain of pair sums srf_merge_bwd would compute from the tensor just
        // written -- same additions, same order (bitwise equal) -- without reading it back.  A lane holds 4 consecutive samples:
        // levels 1 and 2 are lane-local, from level 3 on 2, 4, 8 neighbouring lanes (consecutive float4 of the row) combine.
        // (L % 2^(mD-1) == 0, so a group of lanes never straddles the row's end: all of its lanes are inside this branch.)
        if (a.mD > 1) {
          const float p0 = o.x + o.y, p1 = o.z + o.w;
          reinterpret_cast<float2*>(a.mlv[1] + row * (long)(a.L >> 1))[f] = make_float2(p0, p1);
          float sacc = p0 + p1;
          if (a.mD > 2) a.mlv[2][row * (long)(a.L >> 2) + f] = sacc;
#pragma unroll
          for (int k = 3; k < SRF_MAX_DEPTH; ++k) {
            if (k < a.mD) {
              const int grp = 1 << (k - 2);                       // lanes per output
              sacc = sacc + __shfl_down(sacc, grp >> 1, 64);      // (lane, lane + grp/2): left + right half, like the chain
              if ((lane & (grp - 1)) == 0) a.mlv[k][row * (long)(a.L >> k) + (f >> (k - 2))] = sacc;
            }
          }
        }
      }
    }
  }
}

extern "C" size_t srf_gln_bwd_scratch_bytes(int groups, int C) {
  if (groups <= 0 || C <= 0) return 0;
  return sizeof(double) * (size_t)groups * SRF_STAT_BUCKETS * 2 + sizeof(float) * (size_t)groups * C * 4;
}

// gout (+gout2): [groups,C,L] gradient w.r.t. PReLU(GlobLN(x)) (PReLU only if norm->prelu); x: the GlobLN input;
// norm: statistics of x, gamma, beta, slope.  gx: [groups,C,L] (accumulate_gx != 0: added to).  dgamma/dbeta [C],
// dslope [1] are ACCUMULATED into (NULL = skip).
// mode bit 0 (pre-reduced): `scratch` already holds this norm's row partials and S1/S2 buckets (written by the fused
// srf_dwconv5_bwd_impl for exactly this gout/x pair) -- no reduce pass.  mode bit 1: no apply pass (the consumer,
// srf_dwconv5_bwd_impl in apply-on-load form, evaluates it from `scratch`; gx may be NULL).
// One-shot request to the NEXT srf_gln_bwd_impl call that gets this context: fold the merge backward (srf_merge_bwd on its
// output gx) into its apply pass.  srf_bwd_ctx_merge_taken() tells afterwards whether that happened (vectorised apply kernel,
// aligned level buffers); if not, the caller runs srf_merge_bwd itself.
void srf_bwd_ctx_merge_sink(SrfBwdCtx* c, float* const* levels, int D) {
  c->merge_D = (levels && D > 1 && D <= SRF_MAX_DEPTH) ? D : 0;
  c->merge_taken = false;
  for (int k = 0; k < SRF_MAX_DEPTH; ++k) c->merge_lv[k] = (k >= 1 && k < c->merge_D) ? levels[k] : nullptr;
}
bool srf_bwd_ctx_merge_taken(const SrfBwdCtx* c) { return c->merge_taken; }

int srf_gln_bwd_impl(const float* gout, const float* gout2, const float* x, const srf_norm* norm, int groups, int C,
                     int L, float* gx, int accumulate_gx, float* dgamma, float* dbeta, float* dslope, void* scratch,
                     int mode, void* stream, SrfBwdCtx* ctx) {
  const int sink_D = ctx ? ctx->merge_D : 0;      // (consumed by this call, whatever happens)
  if (ctx) ctx->merge_D = 0;
  const bool defer = ctx && ctx->defer;
  const int pre_reduced = mode & 1, no_apply = (mode >> 1) & 1;
  SRF_CHECK_ARG(gout && x && norm && norm->sums && norm->gamma && norm->beta && (gx || no_apply) && scratch,
                "srf_gln_bwd: null pointer");
  SRF_CHECK_ARG(groups > 0 && C > 0 && L > 0, "srf_gln_bwd: bad sizes");
  const long rows = (long)groups * C;
  const int chunks = (L + 1023) / 1024;
  SRF_CHECK_ARG(rows * chunks < (1L << 31), "srf_gln_bwd: tensor too large");
  hipStream_t st = (hipStream_t)stream;
  GlnBwdArgs a;
  a.gout = gout;
  a.gout2 = gout2;
  a.x = x;
  a.nrm = srf_norm_dev(norm);
  a.inv_count = 1.0 / ((double)C * (double)L);
  a.bsums = reinterpret_cast<double*>(scratch);
  a.rowpart = reinterpret_cast<float*>(a.bsums + (size_t)groups * SRF_STAT_BUCKETS * 2);
  a.gx = gx;
  a.C = C;
  a.L = L;
  a.accumulate = accumulate_gx;
  // (deferred mode: the caller's scratch slices are zeroed once per backward)
  if (!pre_reduced && !defer) SRF_CHECK_HIP(hipMemsetAsync(a.bsums, 0, sizeof(double) * (size_t)groups * SRF_STAT_BUCKETS * 2, st));
  const bool v4 = (L % 4) == 0 && srf_aligned16(gout) && srf_aligned16(x) && (!gx || srf_aligned16(gx)) &&
                  (!gout2 || srf_aligned16(gout2)) && srf_kernel_mode() != 1 && !(srf_debug_flags() & (1 << 30));
  const dim3 grid4((unsigned)((rows + 3) / 4));
  if (!pre_reduced) {
    if (v4)
      hipLaunchKernelGGL(srf_gln_bwd_reduce_v4_kernel, grid4, dim3(256), 0, st, a, rows);
    else
      hipLaunchKernelGGL(srf_gln_bwd_reduce_kernel, dim3((unsigned)rows), dim3(256), 0, st, a);
    SRF_CHECK_LAUNCH("gln_bwd_reduce", st);
  }
  if (dgamma || dbeta || (dslope && norm->prelu)) {
    if (defer) {
      ctx->gln.push_back(GlnParamsDesc{a.rowpart, dgamma, dbeta, norm->prelu ? dslope : nullptr, groups, C});
    } else {
      hipLaunchKernelGGL(srf_gln_bwd_params_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((groups + 63) / 64)),
                         dim3(256), 0, st, a.rowpart, groups, C, dgamma, dbeta, norm->prelu ? dslope : nullptr);
      SRF_CHECK_LAUNCH("gln_bwd_params", st);
    }
  }
  if (no_apply) return SRF_OK;
  if (v4 && sink_D > 1 && (L % (1 << (sink_D - 1))) == 0 && !accumulate_gx) {
    bool ok = true;
    for (int k = 1; k < sink_D; ++k) ok = ok && ctx->merge_lv[k] && srf_aligned16(ctx->merge_lv[k]);
    if (ok) {
      a.mD = sink_D;
      for (int k = 1; k < sink_D; ++k) a.mlv[k] = ctx->merge_lv[k];
      ctx->merge_taken = true;
    }
  }
  if (v4)
    hipLaunchKernelGGL(srf_gln_bwd_apply_v4_kernel, grid4, dim3(256), 0, st, a, rows);
  else
    hipLaunchKernelGGL(srf_gln_bwd_apply_kernel, dim3((unsigned)(rows * chunks)), dim3(256), 0, st, a, chunks);
  SRF_CHECK_LAUNCH("gln_bwd_apply", st);
  return SRF_OK;
}

extern "C" int srf_gln_bwd(const float* gout, const float* gout2, const float* x, const srf_norm* norm, int groups, int C,
                           int L, float* gx, int accumulate_gx, float* dgamma, float* dbeta, float* dslope,
                           void* scratch, void* stream) {
  return srf_gln_bwd_impl(gout, gout2, x, norm, groups, C, L, gx, accumulate_gx, dgamma, dbeta, dslope, scratch, 0, stream,
                          nullptr);
}

// =============================================================================================
// Merge backward.  merged[i] = sum_k n_k[i >> k]  (improved_sudormrf.py:214-216)  =>
//   g_n_0 = g_merged,  g_n_k[j] = g_n_{k-1}[2j] + g_n_{k-1}[2j+1]      (a chain of pair sums)
// =============================================================================================
__global__ __launch_bounds__(256) void srf_pairsum_kernel(const float* __restrict__ in, float* __restrict__ out, long n_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const float2 v = reinterpret_cast<const float2*>(in)[i];
  out[i] = v.x + v.y;
}

// The whole chain in one pass (round 3): a thread owns 2^(D-1) consecutive samples of g_merged and writes its 2^(D-1-k) sums
// of every level k -- the same additions in the same order as the chain of pair-sum launches (bitwise equal), but g_merged is
// read once and no level is read back: 406 MB instead of 590 MB per cfg-2 block, one launch instead of four.
struct MergeBwdArgs {
  const float* g;
  float* lv[SRF_MAX_DEPTH];
  long chunks;    // rows * (L >> (D - 1))
};
template <int D>
__global__ __launch_bounds__(256) void srf_merge_bwd_fused_kernel(MergeBwdArgs a) {
  constexpr int N = 1 << (D - 1);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.chunks) return;
  float v[N];
  if constexpr (N >= 4) {
    const float4* src = reinterpret_cast<const float4*>(a.g) + i * (N / 4);
#pragma unroll
    for (int j = 0; j < N / 4; ++j) {
      const float4 t = src[j];
      v[4 * j] = t.x;
      v[4 * j + 1] = t.y;
      v[4 * j + 2] = t.z;
      v[4 * j + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = a.g[i * N + j];
  }
#pragma unroll
  for (int k = 1; k < D; ++k) {
    const int n = N >> k;                 // this thread's outputs at level k
#pragma unroll
    for (int j = 0; j < n; ++j) v[j] = v[2 * j] + v[2 * j + 1];
    float* dst = a.lv[k] + i * n;
    if (n >= 4) {
#pragma unroll
      for (int j = 0; j < n / 4; ++j) reinterpret_cast<float4*>(dst)[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    } else if (n == 2) {
      *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
    } else {
      dst[0] = v[0];
    }
  }
}

// g_levels[0] is not written (it IS g_merged); g_levels[k] : [rows, L >> k] for k = 1..D-1
extern "C" int srf_merge_bwd(const float* g_merged, float* const* g_levels, int D, long rows, int L, void* stream) {
  SRF_CHECK_ARG(g_merged && g_levels && D >= 1 && D <= SRF_MAX_DEPTH && rows > 0 && L > 0, "srf_merge_bwd: bad arguments");
  SRF_CHECK_ARG((L % (1 << (D - 1))) == 0, "srf_merge_bwd: L must be a multiple of 2^(D-1)");
  for (int k = 1; k < D; ++k) SRF_CHECK_ARG(g_levels[k] != nullptr, "srf_merge_bwd: null level %d", k);
  if (D == 1) return SRF_OK;
  bool aligned = srf_aligned16(g_merged);
  for (int k = 1; k < D; ++k) aligned = aligned && srf_aligned16(g_levels[k]);
  if (aligned && D >= 3 && srf_kernel_mode() != 1) {
    MergeBwdArgs a;
    a.g = g_merged;
    for (int k = 0; k < SRF_MAX_DEPTH; ++k) a.lv[k] = (k >= 1 && k < D) ? g_levels[k] : nullptr;
    a.chunks = rows * (long)(L >> (D - 1));
    const dim3 grid((unsigned)((a.chunks + 255) / 256)), block(256);
    switch (D) {
      case 3: hipLaunchKernelGGL(srf_merge_bwd_fused_kernel<3>, grid, block, 0, (hipStream_t)stream, a); break;
      case 4: hipLaunchKernelGGL(srf_merge_bwd_fused_kernel<4>, grid, block, 0, (hipStream_t)stream, a); break;
      case 5: hipLaunchKernelGGL(srf_merge_bwd_fused_kernel<5>, grid, block, 0, (hipStream_t)stream, a); break;
      default: hipLaunchKernelGGL(srf_merge_bwd_fused_kernel<6>, grid, block, 0, (hipStream_t)stream, a); break;
    }
    SRF_CHECK_LAUNCH("merge_bwd", stream);
    return SRF_OK;
  }
  const float* src = g_merged;
  for (int k = 1; k < D; ++k) {
    const long n_out = rows * (long)(L >> k);
    hipLaunchKernelGGL(srf_pairsum_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       g_levels[k], n_out);
    src = g_levels[k];
  }
  SRF_CHECK_LAUNCH("merge_bwd", stream);
  return SRF_OK;
}

// =============================================================================================
// Depthwise k=5 conv backward (stride 1 | 2, padding 2).   d[c,j] = bias[c] + sum_t w[c,t] u[c, s j + t - 2],
// u = f(xin) = the forward's operand prologue (GlobLN of the previous level, + PReLU for level 0).
//   g_u[c,i]  = sum_t w[c,t] g_d[c,(i + 2 - t)/s]   (terms with s | (i+2-t) and 0 <= j < Lout)
//   d w[c,t] += sum_{b,j} g_d[b,c,j] u[b,c,s j + t - 2],   d bias[c] += sum_{b,j} g_d[b,c,j]
// One block per row (b,c): row partials of the 6 parameter sums (reduced over the batch by a second kernel) and
// the input gradient.
// =============================================================================================
struct DwBwdArgs {
  const float* gd;
  const float* xin;
  SrfNormDev nrm;
  double inv_count;
  const float* w;
  float* gin;
  float* rowpart;   // [rows][8]
  int C, Lin, Lout, stride;
  // fused GlobLN-backward reduction of the prologue norm (row kernel only; see srf_dwconv5_bwd_fused)
  const float* gadd;     // optional second contribution to the prologue-output gradient, added before the store
  float* nrm_rowpart;    // [rows][4]
  double* nrm_bsums;     // [groups][SRF_STAT_BUCKETS][2]
  // apply-on-load (row kernel only): gd is the gradient w.r.t. the OUTPUT of this conv's own norm `anrm`, whose input
  // (this conv's output) is ax and whose reduced sums are a_bsums; the apply pass of srf_gln_bwd runs on load
  const float* ax;
  SrfNormDev anrm;
  double a_inv_count;
  const double* a_bsums;
};

// Block = (row, chunk of 2048 input positions); a thread owns 8 consecutive input positions i0..i0+7 and the
// conv outputs j that start inside them (stride 1: j = i0..i0+7, stride 2: j = i0/2..i0/2+3), so the input
// gradient and the parameter sums come from two 12-value register windows (u and g_d around the chunk).
template <int S>
__global__ __launch_bounds__(256) void srf_dwconv5_bwd_kernel(DwBwdArgs a) {
  __shared__ float red[4][6];
  const int c = blockIdx.y;
  const long g = blockIdx.z;
  const long row = g * a.C + c;
  float sc = 1.f, sh = 0.f;
  if (a.nrm.sums) {
    float mean, rstd;
    srf_finalize_stats(a.nrm.sums, g, a.inv_count, mean, rstd);
    sc = a.nrm.gamma[c] * rstd;
    sh = a.nrm.beta[c] - mean * sc;
  }
  const bool act = a.nrm.prelu != nullptr;
  const float slope = act ? a.nrm.prelu[0] : 1.f;
  float w[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) w[t] = a.w[c * 5 + t];
  const float* gd = a.gd + row * (long)a.Lout;
  const float* xr = a.xin + row * (long)a.Lin;
  const int Lin = a.Lin, Lout = a.Lout;
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 8;
  float p[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i0 < Lin) {
    // u window: positions i0-2 .. i0+9 (zero outside the row: the conv pads the prologue's OUTPUT)
    float u[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      const int i = i0 - 2 + e;
      float v = 0.f;
      if (i >= 0 && i < Lin) {
        v = fmaf(xr[i], sc, sh);
        if (act) v = srf_prelu(v, slope);
      }
      u[e] = v;
    }
    if (S == 1) {
      // g_d window: j = i0-2 .. i0+9
      float gw[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) {
        const int j = i0 - 2 + e;
        gw[e] = (j >= 0 && j < Lout) ? gd[j] : 0.f;
      }
      // parameters: j = i0 + e (e < 8): u index = j + t - 2 -> window slot e + t
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gv = gw[e + 2];
        p[5] += gv;
#pragma unroll
        for (int t = 0; t < 5; ++t) p[t] = fmaf(gv, u[e + t], p[t]);
      }
      // input gradient: g_u[i] = sum_t w[t] g_d[i + 2 - t] -> window slot (i - i0) + 4 - t
      if (a.gin) {
        float* gi = a.gin + row * (long)Lin;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float acc = 0.f;
#pragma unroll
          for (int t = 0; t < 5; ++t) acc = fmaf(w[t], gw[e + 4 - t], acc);
          if (i0 + e < Lin) gi[i0 + e] = acc;
        }
      }
    } else {
      // stride 2: outputs j = j0-1 .. j0+4 with j0 = i0/2 (i0 is a multiple of 8)
      const int j0 = i0 >> 1;
      float gw[6];
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int j = j0 - 1 + e;
        gw[e] = (j >= 0 && j < Lout) ? gd[j] : 0.f;
      }
      // parameters: j = j0 + q (q < 4): u index 2j + t - 2 = i0 + 2q + t - 2 -> window slot 2q + t
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float gv = gw[q + 1];
        p[5] += gv;
#pragma unroll
        for (int t = 0; t < 5; ++t) p[t] = fmaf(gv, u[2 * q + t], p[t]);
      }
      // input gradient: i = i0 + e; terms with (i + 2 - t) even: j = (i + 2 - t)/2 -> window slot j - j0 + 1
      if (a.gin) {
        float* gi = a.gin + row * (long)Lin;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float acc = 0.f;
#pragma unroll
          for (int t = 0; t < 5; ++t) {
            if (((e + 2 - t) & 1) == 0) acc = fmaf(w[t], gw[(e + 2 - t) / 2 + 1], acc);   // (e+2-t) in [-2, 9]
          }
          if (i0 + e < Lin) gi[i0 + e] = acc;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p[k] += __shfl_xor(p[k], o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red[threadIdx.x >> 6][k] = p[k];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    const float v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    if (gridDim.x == 1)
      a.rowpart[row * 8 + k] = v;
    else
      atomicAdd(&a.rowpart[row * 8 + k], v);
  }
}

// Fast variant (Lin % 4 == 0, 16-byte aligned rows): a thread owns 4 consecutive input positions (one float4 of
// xin / of the input gradient, one float4 or float2 of g_d), neighbouring lanes are neighbouring positions, so
// every global access is a fully coalesced 1 KB (512 B) per wavefront instruction; the 2-sample halos of both
// windows come from the neighbouring lanes (wavefront shuffles), the wavefront's two edge lanes load theirs.
template <int S>
__global__ __launch_bounds__(256) void srf_dwconv5_bwd_fast_kernel(DwBwdArgs a) {
  __shared__ float red[4][6];
  const int c = blockIdx.y;
  const long g = blockIdx.z;
  const long row = g * a.C + c;
  float sc = 1.f, sh = 0.f;
  if (a.nrm.sums) {
    float mean, rstd;
    srf_finalize_stats(a.nrm.sums, g, a.inv_count, mean, rstd);
    sc = a.nrm.gamma[c] * rstd;
    sh = a.nrm.beta[c] - mean * sc;
  }
  const bool act = a.nrm.prelu != nullptr;
  const float slope = act ? a.nrm.prelu[0] : 1.f;
  float w[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) w[t] = a.w[c * 5 + t];
  const float* gd = a.gd + row * (long)a.Lout;
  const float* xr = a.xin + row * (long)a.Lin;
  const int Lin = a.Lin, Lout = a.Lout;
  const int lane = threadIdx.x & 63;
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool valid = i0 < Lin;
  auto pro = [&](float v) {
    v = fmaf(v, sc, sh);
    return act ? srf_prelu(v, slope) : v;
  };
  float u[8];
  {
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) xv = *reinterpret_cast<const float4*>(xr + i0);
    u[2] = valid ? pro(xv.x) : 0.f;
    u[3] = valid ? pro(xv.y) : 0.f;
    u[4] = valid ? pro(xv.z) : 0.f;
    u[5] = valid ? pro(xv.w) : 0.f;
    u[0] = __shfl_up(u[4], 1, 64);
    u[1] = __shfl_up(u[5], 1, 64);
    u[6] = __shfl_down(u[2], 1, 64);
    u[7] = __shfl_down(u[3], 1, 64);
    if (lane == 0) {
      u[0] = (i0 - 2 >= 0 && i0 - 2 < Lin) ? pro(xr[i0 - 2]) : 0.f;
      u[1] = (i0 - 1 >= 0 && i0 - 1 < Lin) ? pro(xr[i0 - 1]) : 0.f;
    }
    if (lane == 63) {
      u[6] = (i0 + 4 < Lin) ? pro(xr[i0 + 4]) : 0.f;
      u[7] = (i0 + 5 < Lin) ? pro(xr[i0 + 5]) : 0.f;
    }
  }
  float p[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float gi[4];
  if (S == 1) {
    float gw[8];
    float4 gv4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) gv4 = *reinterpret_cast<const float4*>(gd + i0);
    gw[2] = gv4.x;
    gw[3] = gv4.y;
    gw[4] = gv4.z;
    gw[5] = gv4.w;
    gw[0] = __shfl_up(gw[4], 1, 64);
    gw[1] = __shfl_up(gw[5], 1, 64);
    gw[6] = __shfl_down(gw[2], 1, 64);
    gw[7] = __shfl_down(gw[3], 1, 64);
    if (lane == 0) {
      gw[0] = (i0 - 2 >= 0 && i0 - 2 < Lout) ? gd[i0 - 2] : 0.f;
      gw[1] = (i0 - 1 >= 0 && i0 - 1 < Lout) ? gd[i0 - 1] : 0.f;
    }
    if (lane == 63) {
      gw[6] = (i0 + 4 < Lout) ? gd[i0 + 4] : 0.f;
      gw[7] = (i0 + 5 < Lout) ? gd[i0 + 5] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gv = gw[e + 2];
      p[5] += gv;
#pragma unroll
      for (int t = 0; t < 5; ++t) p[t] = fmaf(gv, u[e + t], p[t]);
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 5; ++t) acc = fmaf(w[t], gw[e + 4 - t], acc);
      gi[e] = acc;
    }
  } else {
    const int j0 = i0 >> 1;
    float gq[4];
    float2 g2 = make_float2(0.f, 0.f);
    if (valid) g2 = *reinterpret_cast<const float2*>(gd + j0);   // Lout = Lin/2: both outputs exist
    gq[1] = g2.x;
    gq[2] = g2.y;
    gq[0] = __shfl_up(gq[2], 1, 64);
    gq[3] = __shfl_down(gq[1], 1, 64);
    if (lane == 0) gq[0] = (j0 - 1 >= 0 && j0 - 1 < Lout) ? gd[j0 - 1] : 0.f;
    if (lane == 63) gq[3] = (j0 + 2 < Lout) ? gd[j0 + 2] : 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float gv = gq[q + 1];
      p[5] += gv;
#pragma unroll
      for (int t = 0; t < 5; ++t) p[t] = fmaf(gv, u[2 * q + t], p[t]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        if (((e + 2 - t) & 1) == 0) acc = fmaf(w[t], gq[(e + 2 - t) / 2 + 1], acc);   // (e+2-t) in [-2, 5]
      }
      gi[e] = acc;
    }
  }
  if (a.gin && valid) *reinterpret_cast<float4*>(a.gin + row * (long)Lin + i0) = make_float4(gi[0], gi[1], gi[2], gi[3]);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p[k] += __shfl_xor(p[k], o, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red[threadIdx.x >> 6][k] = p[k];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    const float v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    if (gridDim.x == 1)
      a.rowpart[row * 8 + k] = v;
    else
      atomicAdd(&a.rowpart[row * 8 + k], v);
  }
}

// Row-per-wavefront variant of the fast kernel (same preconditions; rows that fit 32-bit float4 counts): the
// wavefront walks its row 64 float4 at a time, keeps the six parameter sums in registers across the row and reduces
// them once with DPP -- no LDS, no barrier, no atomics, no memset of rowpart, the statistics finalised once per row
// instead of once per 1024 positions.  Halos between neighbouring lanes are DPP wavefront shifts; the two edge lanes
// fetch theirs with one 8-byte (4-byte) load issued together with the main loads.
// FUSE: the gradient written is (conv input gradient + gadd) = the complete gradient w.r.t. the prologue's output, and
// the kernel also emits what srf_gln_bwd's reduce pass would compute for that prologue norm from it (row partials +
// S1/S2 buckets): the input tensor is in registers anyway, so the norm's backward needs no reduce pass of its own.
// APPLY: a.gd holds the gradient w.r.t. the output of the GlobLN that FOLLOWS this conv; its apply pass
// (srf_gln_bwd_apply: g_d = rstd (gamma g_z - S1/n - xh S2/n)) runs on the loaded values, so g_d never exists in HBM.
template <int S, bool FUSE, bool APPLY>
__global__ __launch_bounds__(256) void srf_dwconv5_bwd_row_kernel(DwBwdArgs a, long rows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;   // wave-uniform
  const int c = (int)(row % a.C);
  const long g = row / a.C;
  float sc = 1.f, sh = 0.f, mean = 0.f, rstd = 1.f, gam = 1.f, bet = 0.f;
  if (a.nrm.sums) {
    srf_finalize_stats(a.nrm.sums, g, a.inv_count, mean, rstd);
    gam = a.nrm.gamma[c];
    bet = a.nrm.beta[c];
    sc = gam * rstd;
    sh = bet - mean * sc;
  }
  const bool act = a.nrm.prelu != nullptr;
  const float slope = act ? a.nrm.prelu[0] : 1.f;
  float w[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) w[t] = a.w[c * 5 + t];
  const float* gd = a.gd + row * (long)a.Lout;
  const float* xr = a.xin + row * (long)a.Lin;
  float* gin = a.gin ? a.gin + row * (long)a.Lin : nullptr;
  const float* gadd = (FUSE && a.gadd) ? a.gadd + row * (long)a.Lin : nullptr;
  const float* ax = APPLY ? a.ax + row * (long)a.Lout : nullptr;
  float amean = 0.f, arstd = 1.f, agam = 1.f, abet = 0.f, am1 = 0.f, am2 = 0.f, aslope = 1.f;
  bool aact = false;
  if (APPLY) {
    srf_finalize_stats(a.anrm.sums, g, a.a_inv_count, amean, arstd);
    const double2 bk = reinterpret_cast<const double2*>(a.a_bsums)[g * SRF_STAT_BUCKETS + lane];
    am1 = (float)(srf_wave_sum(bk.x) * a.a_inv_count);
    am2 = (float)(srf_wave_sum(bk.y) * a.a_inv_count);
    agam = a.anrm.gamma[c];
    abet = a.anrm.beta[c];
    aact = a.anrm.prelu != nullptr;
    aslope = aact ? a.anrm.prelu[0] : 1.f;
  }
  auto gd_of = [&](float gv, float x) {   // APPLY: one element of the following norm's backward
    float xh, gz, unused = 0.f;
    srf_gln_bwd_elem(gv, x, amean, arstd, agam, abet, aact, aslope, xh, gz, unused);
    return arstd * (agam * gz - am1 - xh * am2);
  };
  float n0 = 0.f, n1 = 0.f, n2 = 0.f;   // FUSE: sum g_z, sum g_z xh, slope term of the prologue norm
  const int Lin = a.Lin, L4 = a.Lin >> 2;
  auto pro = [&](float v) {
    v = fmaf(v, sc, sh);
    return act ? srf_prelu(v, slope) : v;
  };
  const bool edge = lane == 0 || lane == 63;
  float p[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int f0 = 0; f0 < L4; f0 += 64) {
    const int f = f0 + lane;
    const bool valid = f < L4;
    const int i0 = f * 4;
    // ---- all loads of the trip first
    const float4 xv = srf_ld4<SRF_BWD_NT>(xr + (valid ? i0 : 0));
    float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FUSE && gadd) ga = srf_ld4<SRF_BWD_NT>(gadd + (valid ? i0 : 0));
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 g2 = make_float2(0.f, 0.f);
    float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 d2 = make_float2(0.f, 0.f);
    if (S == 1) {
      g4 = srf_ld4<SRF_BWD_NT>(gd + (valid ? i0 : 0));
      if (APPLY) d4 = srf_ld4<SRF_BWD_NT>(ax + (valid ? i0 : 0));
    } else {
      g2 = srf_ld2<SRF_BWD_NT>(gd + (valid ? (i0 >> 1) : 0));   // Lout = Lin/2: both outputs exist
      if (APPLY) d2 = srf_ld2<SRF_BWD_NT>(ax + (valid ? (i0 >> 1) : 0));
    }
    // edge lanes: the two positions left of lane 0 / right of lane 63 (i0 % 4 == 0: in range together or not at all)
    const int hi = lane == 0 ? i0 - 2 : i0 + 4;
    const bool hok = edge && valid && hi >= 0 && hi < Lin;
    float2 hx = make_float2(0.f, 0.f), hg = make_float2(0.f, 0.f), hd = make_float2(0.f, 0.f);
    if (hok) {
      hx = *reinterpret_cast<const float2*>(xr + hi);
      if (S == 1) {
        hg = *reinterpret_cast<const float2*>(gd + hi);
        if (APPLY) hd = *reinterpret_cast<const float2*>(ax + hi);
      } else {
        const int hj = lane == 0 ? (i0 >> 1) - 1 : (i0 >> 1) + 2;   // in range exactly when hi is
        hg.x = gd[hj];
        if (APPLY) hd.x = ax[hj];
      }
    }
    if (APPLY) {
      g4 = make_float4(gd_of(g4.x, d4.x), gd_of(g4.y, d4.y), gd_of(g4.z, d4.z), gd_of(g4.w, d4.w));
      g2 = make_float2(gd_of(g2.x, d2.x), gd_of(g2.y, d2.y));
      hg = hok ? make_float2(gd_of(hg.x, hd.x), gd_of(hg.y, hd.y)) : make_float2(0.f, 0.f);
    }
    // ---- prologue'd input window u[0..7] = positions i0-2 .. i0+5
    float u[8];
    u[2] = valid ? pro(xv.x) : 0.f;
    u[3] = valid ? pro(xv.y) : 0.f;
    u[4] = valid ? pro(xv.z) : 0.f;
    u[5] = valid ? pro(xv.w) : 0.f;
    u[0] = srf_lane_up(u[4]);
    u[1] = srf_lane_up(u[5]);
    u[6] = srf_lane_down(u[2]);
    u[7] = srf_lane_down(u[3]);
    const float e0 = hok ? pro(hx.x) : 0.f, e1 = hok ? pro(hx.y) : 0.f;
    if (lane == 0) {
      u[0] = e0;
      u[1] = e1;
    }
    if (lane == 63) {
      u[6] = e0;
      u[7] = e1;
    }
    float gi[4];
    if (S == 1) {
      float gw[8];
      gw[2] = valid ? g4.x : 0.f;
      gw[3] = valid ? g4.y : 0.f;
      gw[4] = valid ? g4.z : 0.f;
      gw[5] = valid ? g4.w : 0.f;
      gw[0] = srf_lane_up(gw[4]);
      gw[1] = srf_lane_up(gw[5]);
      gw[6] = srf_lane_down(gw[2]);
      gw[7] = srf_lane_down(gw[3]);
      if (lane == 0) {
        gw[0] = hg.x;
        gw[1] = hg.y;
      }
      if (lane == 63) {
        gw[6] = hg.x;
        gw[7] = hg.y;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gv = gw[e + 2];
        p[5] += gv;
#pragma unroll
        for (int t = 0; t < 5; ++t) p[t] = fmaf(gv, u[e + t], p[t]);
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) acc = fmaf(w[t], gw[e + 4 - t], acc);
        gi[e] = acc;
      }
    } else {
      float gq[4];
      gq[1] = valid ? g2.x : 0.f;
      gq[2] = valid ? g2.y : 0.f;
      gq[0] = srf_lane_up(gq[2]);
      gq[3] = srf_lane_down(gq[1]);
      if (lane == 0) gq[0] = hg.x;
      if (lane == 63) gq[3] = hg.x;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float gv = gq[q + 1];
        p[5] += gv;
#pragma unroll
        for (int t = 0; t < 5; ++t) p[t] = fmaf(gv, u[2 * q + t], p[t]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          if (((e + 2 - t) & 1) == 0) acc = fmaf(w[t], gq[(e + 2 - t) / 2 + 1], acc);   // (e+2-t) in [-2, 5]
        }
        gi[e] = acc;
      }
    }
    if (FUSE) {
      gi[0] += ga.x;
      gi[1] += ga.y;
      gi[2] += ga.z;
      gi[3] += ga.w;
      const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xh, gz;
        srf_gln_bwd_elem(valid ? gi[e] : 0.f, xe[e], mean, rstd, gam, bet, act, slope, xh, gz, n2);
        n0 += gz;
        n1 = fmaf(gz, xh, n1);
      }
    }
    if (gin && valid) *reinterpret_cast<float4*>(gin + i0) = make_float4(gi[0], gi[1], gi[2], gi[3]);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) p[k] = srf_dpp_wave_sum(p[k]);
  if (FUSE) {
    n0 = srf_dpp_wave_sum(n0);
    n1 = srf_dpp_wave_sum(n1);
    n2 = srf_dpp_wave_sum(n2);
  }
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 6; ++k) a.rowpart[row * 8 + k] = p[k];
    if (FUSE) {
      float* rp = a.nrm_rowpart + row * 4;
      rp[0] = n0;
      rp[1] = n1;
      rp[2] = n2;
      double* dst = srf_stat_slot(a.nrm_bsums, g, c);
      atomicAdd(dst, (double)gam * (double)n0);
      atomicAdd(dst + 1, (double)gam * (double)n1);
    }
  }
}

// dw[c][t] += sum_g rowpart[g][c][t] (t < 5), dbias[c] += sum_g rowpart[g][c][5]; same blocking as above
__device__ __forceinline__ void srf_dwconv5_bwd_params_body(const float* __restrict__ rowpart, int groups, int C, float* dw,
                                                            float* dbias) {
  __shared__ float red[8][32][6];
  const int cl = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int g0 = blockIdx.y * 64, g1 = min(g0 + 64, groups);
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    for (int g = g0 + slice; g < g1; g += 8) {
      const float* rp = rowpart + ((long)g * C + c) * 8;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += rp[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) red[slice][cl][k] = acc[k];
  __syncthreads();
  if (slice == 0 && c < C) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      float t = 0.f;
#pragma unroll
      for (int sidx = 0; sidx < 8; ++sidx) t += red[sidx][cl][k];
      if (k < 5) {
        if (dw) atomicAdd(dw + c * 5 + k, t);
      } else if (dbias) {
        atomicAdd(dbias + c, t);
      }
    }
  }
}

__global__ __launch_bounds__(256) void srf_dwconv5_bwd_params_kernel(const float* __restrict__ rowpart, int groups, int C,
                                                                     float* dw, float* dbias) {
  srf_dwconv5_bwd_params_body(rowpart, groups, C, dw, dbias);
}
__global__ __launch_bounds__(256) void srf_dwconv5_bwd_params_batch_kernel(DwParamsTable t) {
  const DwParamsDesc d = t.d[blockIdx.z];
  if ((int)blockIdx.x * 32 >= d.C || (int)blockIdx.y * 64 >= d.groups) return;   // (block-uniform)
  srf_dwconv5_bwd_params_body(d.rowpart, d.groups, d.C, d.dw, d.dbias);
}

// Fold every recorded partial into its parameter gradients (batched launches of up to SRF_PB_MAX reductions) and clear the list.
int srf_bwd_ctx_flush(SrfBwdCtx* c, hipStream_t st) {
  for (size_t base = 0; base < c->gln.size(); base += SRF_PB_MAX) {
    GlnParamsTable t;
    const int cnt = (int)std::min<size_t>(SRF_PB_MAX, c->gln.size() - base);
    int maxC = 0, maxG = 0;
    for (int i = 0; i < SRF_PB_MAX; ++i) {
      t.d[i] = c->gln[base + (i < cnt ? i : 0)];
      if (i < cnt) {
        maxC = std::max(maxC, t.d[i].C);
        maxG = std::max(maxG, t.d[i].groups);
      }
    }
    hipLaunchKernelGGL(srf_gln_bwd_params_batch_kernel, dim3((unsigned)((maxC + 31) / 32), (unsigned)((maxG + 63) / 64), (unsigned)cnt),
                       dim3(256), 0, st, t);
    SRF_CHECK_LAUNCH("gln_bwd_params", st);
  }
  c->gln.clear();
  for (size_t base = 0; base < c->dw.size(); base += SRF_PB_MAX) {
    DwParamsTable t;
    const int cnt = (int)std::min<size_t>(SRF_PB_MAX, c->dw.size() - base);
    int maxC = 0, maxG = 0;
    for (int i = 0; i < SRF_PB_MAX; ++i) {
      t.d[i] = c->dw[base + (i < cnt ? i : 0)];
      if (i < cnt) {
        maxC = std::max(maxC, t.d[i].C);
        maxG = std::max(maxG, t.d[i].groups);
      }
    }
    hipLaunchKernelGGL(srf_dwconv5_bwd_params_batch_kernel, dim3((unsigned)((maxC + 31) / 32), (unsigned)((maxG + 63) / 64), (unsigned)cnt),
                       dim3(256), 0, st, t);
    SRF_CHECK_LAUNCH("dwconv5_bwd_params", st);
  }
  c->dw.clear();
  return SRF_OK;
}

extern "C" size_t srf_dwconv5_bwd_scratch_bytes(int groups, int C) {
  return groups > 0 && C > 0 ? sizeof(float) * (size_t)groups * C * 8 : 0;
}

// gd: [groups,C,Lout]; xin: [groups,C,Lin] the PRE-norm input tensor of this conv, in_norm its prologue (NULL =
// identity); gin: [groups,C,Lin] gradient w.r.t. the prologue's OUTPUT (overwritten; NULL = skip); dw [C,5] and
// dbias [C] are ACCUMULATED into.
// gln_scratch != NULL asks for the fused form: gin <- conv input gradient + gadd (gadd may be NULL), and the reduce
// pass of srf_gln_bwd for the prologue norm `in_norm` over that complete gradient written into gln_scratch (layout of
// srf_gln_bwd's scratch).  *fused reports whether that happened (row kernel preconditions); when it did not, gin
// holds the plain conv input gradient and the caller runs the unfused sequence.
// ax != NULL asks for the apply-on-load form on top (only together with the fused form; check
// srf_dwconv5_bwd_rowwise_ok first): gd is then the gradient w.r.t. the OUTPUT of the norm `anorm` that follows this
// conv, ax that norm's input (= this conv's output) and a_scratch its reduced sums (srf_gln_bwd_impl mode bit 1).
bool srf_dwconv5_bwd_rowwise_ok(int Lin, int stride, const void* const* ptrs, int nptrs) {
  if ((Lin % 4) != 0 || srf_kernel_mode() == 1 || (srf_debug_flags() & ((1 << 29) | (1 << 30)))) return false;
  if (stride == 2 && ((Lin - 1) / 2 + 1) * 2 != Lin) return false;
  for (int i = 0; i < nptrs; ++i)
    if (ptrs[i] && !srf_aligned16(ptrs[i])) return false;
  return true;
}

int srf_dwconv5_bwd_impl(const float* gd, const float* xin, const srf_norm* in_norm, const float* w, int groups, int C,
                         int Lin, int stride, float* gin, float* dw, float* dbias, void* scratch, const float* gadd,
                         void* gln_scratch, int* fused, const float* ax, const srf_norm* anorm, const void* a_scratch,
                         void* stream, SrfBwdCtx* ctx) {
  if (fused) *fused = 0;
  const bool defer = ctx && ctx->defer;
  SRF_CHECK_ARG(gd && xin && w && scratch, "srf_dwconv5_bwd: null pointer");
  SRF_CHECK_ARG(groups > 0 && C > 0 && Lin > 0 && (stride == 1 || stride == 2), "srf_dwconv5_bwd: bad sizes");
  const long rows = (long)groups * C;
  SRF_CHECK_ARG(rows < (1L << 31), "srf_dwconv5_bwd: too many rows");
  DwBwdArgs a;
  a.gd = gd;
  a.xin = xin;
  a.nrm = srf_norm_dev(in_norm);
  a.inv_count = 1.0 / ((double)C * (double)Lin);
  a.w = w;
  a.gin = gin;
  a.rowpart = reinterpret_cast<float*>(scratch);
  a.C = C;
  a.Lin = Lin;
  a.Lout = (Lin - 1) / stride + 1;
  a.stride = stride;
  a.gadd = nullptr;
  a.nrm_rowpart = nullptr;
  a.nrm_bsums = nullptr;
  a.ax = nullptr;
  a.anrm = SrfNormDev{};
  a.a_inv_count = 0.0;
  a.a_bsums = nullptr;
  hipStream_t st = (hipStream_t)stream;
  SRF_CHECK_ARG(groups <= 65535 && C <= 65535, "srf_dwconv5_bwd: groups / channels exceed 65535");
  const bool fast = (Lin % 4) == 0 && srf_aligned16(gd) && srf_aligned16(xin) && (!gin || srf_aligned16(gin)) &&
                    srf_kernel_mode() != 1;
  const int per_block = fast ? 1024 : 2048;
  const int chunks = (Lin + per_block - 1) / per_block;
  // stride 2 with an odd output count (Lin % 8 == 4) keeps the chunked kernel: its float2 loads assume Lout = Lin / 2
  const bool rowwise = fast && (stride == 1 || a.Lout * 2 == Lin) && !(srf_debug_flags() & (1 << 29));
  if (rowwise) {
    const dim3 grid4((unsigned)((rows + 3) / 4));
    const bool fuse = gln_scratch && gin && in_norm && in_norm->sums && in_norm->gamma && in_norm->beta &&
                      (!gadd || srf_aligned16(gadd)) && !(srf_debug_flags() & (1 << 30));
    SRF_CHECK_ARG(!ax || (fuse && anorm && anorm->sums && anorm->gamma && anorm->beta && a_scratch && srf_aligned16(ax)),
                  "srf_dwconv5_bwd: apply-on-load needs the fused row kernel");
    if (fuse) {
      a.gadd = gadd;
      a.nrm_bsums = reinterpret_cast<double*>(gln_scratch);
      a.nrm_rowpart = reinterpret_cast<float*>(a.nrm_bsums + (size_t)groups * SRF_STAT_BUCKETS * 2);
      if (!defer) SRF_CHECK_HIP(hipMemsetAsync(a.nrm_bsums, 0, sizeof(double) * (size_t)groups * SRF_STAT_BUCKETS * 2, st));
      if (ax) {
        a.ax = ax;
        a.anrm = srf_norm_dev(anorm);
        a.a_inv_count = 1.0 / ((double)C * (double)a.Lout);
        a.a_bsums = reinterpret_cast<const double*>(a_scratch);
        if (stride == 1)
          hipLaunchKernelGGL((srf_dwconv5_bwd_row_kernel<1, true, true>), grid4, dim3(256), 0, st, a, rows);
        else
          hipLaunchKernelGGL((srf_dwconv5_bwd_row_kernel<2, true, true>), grid4, dim3(256), 0, st, a, rows);
      } else if (stride == 1) {
        hipLaunchKernelGGL((srf_dwconv5_bwd_row_kernel<1, true, false>), grid4, dim3(256), 0, st, a, rows);
      } else {
        hipLaunchKernelGGL((srf_dwconv5_bwd_row_kernel<2, true, false>), grid4, dim3(256), 0, st, a, rows);
      }
      if (fused) *fused = 1;
    } else if (stride == 1) {
      hipLaunchKernelGGL((srf_dwconv5_bwd_row_kernel<1, false, false>), grid4, dim3(256), 0, st, a, rows);
    } else {
      hipLaunchKernelGGL((srf_dwconv5_bwd_row_kernel<2, false, false>), grid4, dim3(256), 0, st, a, rows);
    }
    SRF_CHECK_LAUNCH("dwconv5_bwd", st);
  } else {
    SRF_CHECK_ARG(!ax, "srf_dwconv5_bwd: apply-on-load needs the fused row kernel");
  if (chunks > 1) SRF_CHECK_HIP(hipMemsetAsync(a.rowpart, 0, sizeof(float) * (size_t)rows * 8, st));
  dim3 grid((unsigned)chunks, (unsigned)C, (unsigned)groups);
  if (fast && stride == 1)
    hipLaunchKernelGGL(srf_dwconv5_bwd_fast_kernel<1>, grid, dim3(256), 0, st, a);
  else if (fast)
    hipLaunchKernelGGL(srf_dwconv5_bwd_fast_kernel<2>, grid, dim3(256), 0, st, a);
  else if (stride == 1)
    hipLaunchKernelGGL(srf_dwconv5_bwd_kernel<1>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(srf_dwconv5_bwd_kernel<2>, grid, dim3(256), 0, st, a);
  SRF_CHECK_LAUNCH("dwconv5_bwd", st);
  }
  if (dw || dbias) {
    if (defer) {
      ctx->dw.push_back(DwParamsDesc{a.rowpart, dw, dbias, groups, C});
    } else {
      hipLaunchKernelGGL(srf_dwconv5_bwd_params_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((groups + 63) / 64)),
                         dim3(256), 0, st, a.rowpart, groups, C, dw, dbias);
      SRF_CHECK_LAUNCH("dwconv5_bwd_params", st);
    }
  }
  return SRF_OK;
}

extern "C" int srf_dwconv5_bwd(const float* gd, const float* xin, const srf_norm* in_norm, const float* w, int groups,
                               int C, int Lin, int stride, float* gin, float* dw, float* dbias, void* scratch,
                               void* stream) {
  return srf_dwconv5_bwd_impl(gd, xin, in_norm, w, groups, C, Lin, stride, gin, dw, dbias, scratch, nullptr, nullptr,
                              nullptr, nullptr, nullptr, nullptr, stream, nullptr);
}

// =============================================================================================
// Round 6: the HEAD of a block's pyramid backward in two passes over {G_0, y1} instead of 7 C*L of traffic.
//   forward   o = PReLU(GlobLN_p(y1))          proj_1x1's norm + activation      improved_sudormrf.py:174-175,:205
//             d_0 = conv_0(o)  (k = 5, s = 1)   n_0 = GlobLN_0(d_0)               :178-189,:206
//   backward  G_0 = d loss / d n_0 (complete: merge part + what level 1 sent down; written by level 1's kernel, which
//             also leaves norm 0's reduced sums S1 / S2)
//             g_d0 = GlobLN_0'(G_0; d_0),   g_o = conv_0^T(g_d0),   g_y1 = (GlobLN_p + PReLU)'(g_o; y1)
// Before: the level-0 conv kernel read G_0, d_0, y1 and wrote g_o (4 C*L) while reducing norm p's sums, then norm p's apply
// pass read g_o, y1 and wrote g_y1 (3 C*L).  g_o is needed twice only because norm p's sums must be complete before its
// apply -- but g_o is cheap to RE-COMPUTE, and so is d_0 (five FMAs per element from y1, which both passes read anyway):
//   pass A (reduce): read G_0, y1 -> conv 0's parameter sums, norm p's row partials and S1 / S2 buckets.  Writes no tensor.
//   pass B (apply) : read G_0, y1 -> g_y1.                                         Together 5 C*L, and d_0 is not read at all.
// One wavefront per row; a trip covers 64 consecutive float4 of the row = 62 own + one halo float4 on either side: g_o at
// an own position needs g_d0 at +-2, which the neighbouring LANE holds (DPP shifts); a halo lane's g_d0 is only needed at
// its two positions next to the own range, and those need o at +-2 of THEM -- inside the trip's 64 float4 on both sides.
// So everything is lane-local or one DPP shift away; no LDS, no barrier, every access a full 1-KB line run.
// The recomputed d_0 is the forward's d_0 bit for bit (same expressions in the same order as srf_pyramid_reg.hip).
// =============================================================================================
struct L0pArgs {
  const float* G0;
  const float* y1;
  float* gy1;               // pass B
  SrfNormDev pn;            // proj_1x1's norm (+ PReLU): statistics of y1
  double pn_inv;
  SrfNormDev n0;            // level 0's norm: statistics of d_0
  double n0_inv;
  const double* n0_bsums;   // S1 / S2 of norm 0 (level 1's kernel reduced them)
  double* pn_bsums;         // S1 / S2 of norm p: pass A accumulates, pass B reads
  float* pn_rowpart;        // [rows][4] pass A
  const float* w;           // conv 0 [C][5]
  const float* bias;        // conv 0 [C]
  float* dw_rowpart;        // [rows][8] pass A
  int C, L;
};

template <bool APPLY>
__global__ __launch_bounds__(256) void srf_bwd_l0p_kernel(L0pArgs a, long rows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;   // wave-uniform
  const int c = (int)(row % a.C);
  const long g = row / a.C;
  const int L4 = a.L >> 2;
  const float* yr = a.y1 + row * (long)a.L;
  const float* gr = a.G0 + row * (long)a.L;
  // the first trip's operands travel while the statistics are finalised
  auto f_of = [&](int f0) { return f0 - 1 + lane; };
  auto ld = [&](const float* base, int f) { return srf_ld4<SRF_BWD_NT>(base + 4 * (size_t)min(max(f, 0), L4 - 1)); };
  float4 xv = ld(yr, f_of(0)), gv = ld(gr, f_of(0));
  float pmean, prstd, zmean, zrstd;
  srf_finalize_stats(a.pn.sums, g, a.pn_inv, pmean, prstd);
  srf_finalize_stats(a.n0.sums, g, a.n0_inv, zmean, zrstd);
  const double2 bk0 = reinterpret_cast<const double2*>(a.n0_bsums)[g * SRF_STAT_BUCKETS + lane];
  const float zm1 = (float)(srf_wave_sum(bk0.x) * a.n0_inv), zm2 = (float)(srf_wave_sum(bk0.y) * a.n0_inv);
  float pm1 = 0.f, pm2 = 0.f;
  if (APPLY) {
    const double2 bkp = reinterpret_cast<const double2*>(a.pn_bsums)[g * SRF_STAT_BUCKETS + lane];
    pm1 = (float)(srf_wave_sum(bkp.x) * a.pn_inv);
    pm2 = (float)(srf_wave_sum(bkp.y) * a.pn_inv);
  }
  const float pgam = a.pn.gamma[c], pbet = a.pn.beta[c], zgam = a.n0.gamma[c];
  const float psc = pgam * prstd, psh = pbet - pmean * psc;          // (the forward's expressions: srf_pyramid_reg.hip)
  const bool act = a.pn.prelu != nullptr;
  const float slope = act ? a.pn.prelu[0] : 1.f;
  float w[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) w[t] = a.w[c * 5 + t];
  const float b0 = a.bias[c];
  float p[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float n0 = 0.f, n1 = 0.f, n2 = 0.f;
  for (int f0 = 0; f0 < L4; f0 += 62) {
    const int f = f_of(f0);
    const bool inrow = f >= 0 && f < L4;
    const bool own = inrow && lane >= 1 && lane <= 62;
    const float4 xc = xv, gc = gv;
    if (f0 + 62 < L4) {      // next trip's operands
      xv = ld(yr, f_of(f0 + 62));
      gv = ld(gr, f_of(f0 + 62));
    }
    const float xe[4] = {xc.x, xc.y, xc.z, xc.w};
    const float ge[4] = {gc.x, gc.y, gc.z, gc.w};
    // ---- o = PReLU(GlobLN_p(y1)) on positions 4 f - 2 .. 4 f + 5 (zero outside the row: the conv pads o)
    const float scm = inrow ? psc : 0.f, shm = inrow ? psh : 0.f;
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = fmaf(xe[e], scm, shm);
      if (act) v = srf_prelu(v, slope);
      o[e + 2] = v;
    }
    o[0] = srf_lane_up(o[4]);
    o[1] = srf_lane_up(o[5]);
    o[6] = srf_lane_down(o[2]);
    o[7] = srf_lane_down(o[3]);
    // ---- d_0 (the forward's FMA chain), norm 0's backward on load: g_d0 (zero outside the row)
    float gw[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = fmaf(w[4], o[e + 4], fmaf(w[3], o[e + 3], fmaf(w[2], o[e + 2], fmaf(w[1], o[e + 1], fmaf(w[0], o[e], b0)))));
      const float xh = (d0 - zmean) * zrstd;
      const float gd = zrstd * (zgam * ge[e] - zm1 - xh * zm2);
      gw[e + 2] = inrow ? gd : 0.f;
    }
    gw[0] = srf_lane_up(gw[4]);
    gw[1] = srf_lane_up(gw[5]);
    gw[6] = srf_lane_down(gw[2]);
    gw[7] = srf_lane_down(gw[3]);
    // ---- conv 0: parameter sums (pass A) and the input gradient g_o
    float gi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!APPLY) {
        const float gd = own ? gw[e + 2] : 0.f;
        p[5] += gd;
#pragma unroll
        for (int t = 0; t < 5; ++t) p[t] = fmaf(gd, o[e + t], p[t]);
      }
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 5; ++t) acc = fmaf(w[t], gw[e + 4 - t], acc);
      gi[e] = acc;
    }
    // ---- norm p (+ PReLU) backward: reduce (pass A) or apply (pass B)
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xh, gz;
      srf_gln_bwd_elem(own ? gi[e] : 0.f, xe[e], pmean, prstd, pgam, pbet, act, slope, xh, gz, n2);
      if (APPLY) {
        r[e] = prstd * (pgam * gz - pm1 - xh * pm2);
      } else {
        n0 += gz;
        n1 = fmaf(gz, xh, n1);
      }
    }
    if (APPLY && own) *reinterpret_cast<float4*>(a.gy1 + row * (long)a.L + 4 * (size_t)f) = make_float4(r[0], r[1], r[2], r[3]);
  }
  if (!APPLY) {
#pragma unroll
    for (int k = 0; k < 6; ++k) p[k] = srf_dpp_wave_sum(p[k]);
    n0 = srf_dpp_wave_sum(n0);
    n1 = srf_dpp_wave_sum(n1);
    n2 = srf_dpp_wave_sum(n2);
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 6; ++k) a.dw_rowpart[row * 8 + k] = p[k];
      float* rp = a.pn_rowpart + row * 4;
      rp[0] = n0;
      rp[1] = n1;
      rp[2] = n2;
      double* dst = srf_stat_slot(a.pn_bsums, g, c);
      atomicAdd(dst, (double)pgam * (double)n0);
      atomicAdd(dst + 1, (double)pgam * (double)n1);
    }
  }
}

// G0: [groups,C,L] complete gradient w.r.t. n_0; y1: the block's proj_1x1 conv output; pn / n0: the two norms (pn with its PReLU
// slope); w0 / b0: conv 0; n0_scratch: norm 0's REDUCED scratch slice (srf_gln_bwd layout: buckets, then row partials);
// pn_scratch: norm p's slice -- pass A writes it (buckets zeroed by the caller: the deferred mode's arena), pass B reads it;
// dw_scratch: conv 0's row-partial slice.  dw / dbias: conv 0's parameter gradients (deferred through ctx, or reduced here).
// gy1 must not alias G0 (halo lanes re-read what a neighbouring trip's own lanes would have overwritten).
bool srf_bwd_level0_proj_ok(int L, const void* const* ptrs, int nptrs) {
  if ((L % 4) != 0 || L < 8 || srf_kernel_mode() == 1 || (srf_debug_flags() & ((1 << 16) | (1 << 29) | (1 << 30)))) return false;
  for (int i = 0; i < nptrs; ++i)
    if (!ptrs[i] || !srf_aligned16(ptrs[i])) return false;
  return true;
}
int srf_bwd_level0_proj(const float* G0, const float* y1, const srf_norm* pn, const srf_norm* n0, const float* w0, const float* b0,
                        const void* n0_scratch, void* pn_scratch, void* dw_scratch, float* dw, float* dbias, float* gy1,
                        int groups, int C, int L, void* stream, SrfBwdCtx* ctx) {
  SRF_CHECK_ARG(G0 && y1 && pn && n0 && w0 && b0 && n0_scratch && pn_scratch && dw_scratch && gy1 && gy1 != G0,
                "srf_bwd_level0_proj: bad pointers");
  SRF_CHECK_ARG(pn->sums && pn->gamma && pn->beta && n0->sums && n0->gamma && n0->beta, "srf_bwd_level0_proj: incomplete norms");
  const long rows = (long)groups * C;
  SRF_CHECK_ARG(groups > 0 && C > 0 && rows < (1L << 31), "srf_bwd_level0_proj: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  L0pArgs a;
  a.G0 = G0;
  a.y1 = y1;
  a.gy1 = gy1;
  a.pn = srf_norm_dev(pn);
  a.n0 = srf_norm_dev(n0);
  a.pn_inv = a.n0_inv = 1.0 / ((double)C * (double)L);
  a.n0_bsums = reinterpret_cast<const double*>(n0_scratch);
  a.pn_bsums = reinterpret_cast<double*>(pn_scratch);
  a.pn_rowpart = reinterpret_cast<float*>(a.pn_bsums + (size_t)groups * SRF_STAT_BUCKETS * 2);
  a.w = w0;
  a.bias = b0;
  a.dw_rowpart = reinterpret_cast<float*>(dw_scratch);
  a.C = C;
  a.L = L;
  const bool defer = ctx && ctx->defer;
  if (!defer) SRF_CHECK_HIP(hipMemsetAsync(a.pn_bsums, 0, sizeof(double) * (size_t)groups * SRF_STAT_BUCKETS * 2, st));
  const dim3 grid4((unsigned)((rows + 3) / 4));
  hipLaunchKernelGGL(srf_bwd_l0p_kernel<false>, grid4, dim3(256), 0, st, a, rows);
  SRF_CHECK_LAUNCH("bwd_l0p_reduce", st);
  if (dw || dbias) {
    if (defer) {
      ctx->dw.push_back(DwParamsDesc{a.dw_rowpart, dw, dbias, groups, C});
    } else {
      hipLaunchKernelGGL(srf_dwconv5_bwd_params_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((groups + 63) / 64)), dim3(256),
                         0, st, a.dw_rowpart, groups, C, dw, dbias);
      SRF_CHECK_LAUNCH("dwconv5_bwd_params", st);
    }
  }
  hipLaunchKernelGGL(srf_bwd_l0p_kernel<true>, grid4, dim3(256), 0, st, a, rows);
  SRF_CHECK_LAUNCH("bwd_l0p_apply", st);
  return SRF_OK;
}

// ---- level 1 in the same style: conv 1's backward (stride 2) with its own norm's apply on load, the complete G_0 and norm 0's
// reduce pass -- srf_dwconv5_bwd_row_kernel<2, true, true> -- but with conv 1's INPUT n_0 = GlobLN_0(conv_0(o)) re-computed from
// y1 instead of read from the saved d_0: with the fused head above nothing reads d_0 any more, so the training forward does not
// write it (one C*L less in its pyramid pass).  Same trip layout as srf_bwd_l0p_kernel (62 own float4 of level-0 positions +
// one halo float4 per side; a lane's two level-1 positions are 2 f, 2 f + 1).
struct L1hArgs {
  const float* G1;          // [rows][L/2] gradient w.r.t. n_1 (complete)
  const float* d1;          // [rows][L/2] conv 1's output (norm 1's input)
  const float* y1;          // [rows][L]
  const float* gadd;        // [rows][L] the merge part of G_0 (g_merged)
  float* G0;                // [rows][L] out
  SrfNormDev pn, n0, n1;    // proj norm (+ PReLU), level-0 norm, level-1 norm
  double inv0, inv1;        // 1 / (C L), 1 / (C L / 2)
  const double* n1_bsums;   // S1 / S2 of norm 1 (reduced)
  double* n0_bsums;         // norm 0: reduce pass output
  float* n0_rowpart;        // [rows][4]
  const float* w0;          // conv 0 [C][5], bias0 [C]
  const float* b0;
  const float* w1;          // conv 1 [C][5]
  float* dw_rowpart;        // conv 1 [rows][8]
  int C, L;
};

__global__ __launch_bounds__(256) void srf_bwd_l1h_kernel(L1hArgs a, long rows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;   // wave-uniform
  const int c = (int)(row % a.C);
  const long g = row / a.C;
  const int L4 = a.L >> 2;
  const float* yr = a.y1 + row * (long)a.L;
  const float* mr = a.gadd + row * (long)a.L;
  const float* g1r = a.G1 + row * (long)(a.L >> 1);
  const float* d1r = a.d1 + row * (long)(a.L >> 1);
  auto fc = [&](int f0) { return min(max(f0 - 1 + lane, 0), L4 - 1); };
  float4 xv = srf_ld4<SRF_BWD_NT>(yr + 4 * (size_t)fc(0)), mv = srf_ld4<SRF_BWD_NT>(mr + 4 * (size_t)fc(0));
  float2 gv = srf_ld2<SRF_BWD_NT>(g1r + 2 * (size_t)fc(0)), dv = srf_ld2<SRF_BWD_NT>(d1r + 2 * (size_t)fc(0));
  float pmean, prstd, zmean, zrstd, omean, orstd;
  srf_finalize_stats(a.pn.sums, g, a.inv0, pmean, prstd);
  srf_finalize_stats(a.n0.sums, g, a.inv0, zmean, zrstd);
  srf_finalize_stats(a.n1.sums, g, a.inv1, omean, orstd);
  const double2 bk1 = reinterpret_cast<const double2*>(a.n1_bsums)[g * SRF_STAT_BUCKETS + lane];
  const float om1 = (float)(srf_wave_sum(bk1.x) * a.inv1), om2 = (float)(srf_wave_sum(bk1.y) * a.inv1);
  const float psc = a.pn.gamma[c] * prstd, psh = a.pn.beta[c] - pmean * psc;
  const bool act = a.pn.prelu != nullptr;
  const float slope = act ? a.pn.prelu[0] : 1.f;
  const float zgam = a.n0.gamma[c], zbet = a.n0.beta[c];
  const float zsc = zgam * zrstd, zsh = zbet - zmean * zsc;           // n_0 = d_0 zsc + zsh (the forward's affine)
  const float ogam = a.n1.gamma[c], obet = a.n1.beta[c];
  float w0[5], w1[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    w0[t] = a.w0[c * 5 + t];
    w1[t] = a.w1[c * 5 + t];
  }
  const float b0 = a.b0[c];
  float p[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float n0 = 0.f, n1 = 0.f, n2 = 0.f;
  for (int f0 = 0; f0 < L4; f0 += 62) {
    const int f = f0 - 1 + lane;
    const bool inrow = f >= 0 && f < L4;
    const bool own = inrow && lane >= 1 && lane <= 62;
    const float4 xc = xv, mc = mv;
    const float2 gc = gv, dc = dv;
    if (f0 + 62 < L4) {
      const int fn = fc(f0 + 62);
      xv = srf_ld4<SRF_BWD_NT>(yr + 4 * (size_t)fn);
      mv = srf_ld4<SRF_BWD_NT>(mr + 4 * (size_t)fn);
      gv = srf_ld2<SRF_BWD_NT>(g1r + 2 * (size_t)fn);
      dv = srf_ld2<SRF_BWD_NT>(d1r + 2 * (size_t)fn);
    }
    const float xe[4] = {xc.x, xc.y, xc.z, xc.w};
    // ---- o on positions 4 f - 2 .. 4 f + 5, then d_0 and n_0 (zero outside the row: conv 1 pads n_0) on the own four
    const float scm = inrow ? psc : 0.f, shm = inrow ? psh : 0.f;
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = fmaf(xe[e], scm, shm);
      if (act) v = srf_prelu(v, slope);
      o[e + 2] = v;
    }
    o[0] = srf_lane_up(o[4]);
    o[1] = srf_lane_up(o[5]);
    o[6] = srf_lane_down(o[2]);
    o[7] = srf_lane_down(o[3]);
    float d0[4], u[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d0[e] = fmaf(w0[4], o[e + 4], fmaf(w0[3], o[e + 3], fmaf(w0[2], o[e + 2], fmaf(w0[1], o[e + 1], fmaf(w0[0], o[e], b0)))));
      u[e + 2] = inrow ? fmaf(d0[e], zsc, zsh) : 0.f;
    }
    u[0] = srf_lane_up(u[4]);
    u[1] = srf_lane_up(u[5]);
    u[6] = srf_lane_down(u[2]);
    u[7] = srf_lane_down(u[3]);
    // ---- norm 1's backward on load: g_d1 at level-1 positions 2 f, 2 f + 1 (+ one on either side from the neighbours)
    float gq[4];
    {
      float xh, gz, unused = 0.f;
      srf_gln_bwd_elem(gc.x, dc.x, omean, orstd, ogam, obet, false, 1.f, xh, gz, unused);
      gq[1] = inrow ? orstd * (ogam * gz - om1 - xh * om2) : 0.f;
      srf_gln_bwd_elem(gc.y, dc.y, omean, orstd, ogam, obet, false, 1.f, xh, gz, unused);
      gq[2] = inrow ? orstd * (ogam * gz - om1 - xh * om2) : 0.f;
    }
    gq[0] = srf_lane_up(gq[2]);
    gq[3] = srf_lane_down(gq[1]);
    // ---- conv 1's parameter sums (own lanes) and input gradient; + the merge part = the complete G_0
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float gd = own ? gq[q + 1] : 0.f;
      p[5] += gd;
#pragma unroll
      for (int t = 0; t < 5; ++t) p[t] = fmaf(gd, u[2 * q + t], p[t]);
    }
    float gi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        if (((e + 2 - t) & 1) == 0) acc = fmaf(w1[t], gq[(e + 2 - t) / 2 + 1], acc);   // (e + 2 - t) in [-2, 5]
      }
      gi[e] = acc;
    }
    gi[0] += mc.x;
    gi[1] += mc.y;
    gi[2] += mc.z;
    gi[3] += mc.w;
    // ---- norm 0's reduce pass over the complete gradient (its input d_0 is in registers)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xh, gz;
      srf_gln_bwd_elem(own ? gi[e] : 0.f, d0[e], zmean, zrstd, zgam, zbet, false, 1.f, xh, gz, n2);
      n0 += gz;
      n1 = fmaf(gz, xh, n1);
    }
    if (own) *reinterpret_cast<float4*>(a.G0 + row * (long)a.L + 4 * (size_t)f) = make_float4(gi[0], gi[1], gi[2], gi[3]);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) p[k] = srf_dpp_wave_sum(p[k]);
  n0 = srf_dpp_wave_sum(n0);
  n1 = srf_dpp_wave_sum(n1);
  n2 = srf_dpp_wave_sum(n2);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 6; ++k) a.dw_rowpart[row * 8 + k] = p[k];
    float* rp = a.n0_rowpart + row * 4;
    rp[0] = n0;
    rp[1] = n1;
    rp[2] = n2;
    double* dst = srf_stat_slot(a.n0_bsums, g, c);
    atomicAdd(dst, (double)zgam * (double)n0);
    atomicAdd(dst + 1, (double)zgam * (double)n1);
  }
}

// G1 / d1: [groups,C,L/2]; n1_scratch: norm 1's REDUCED slice; n0_scratch: norm 0's slice (written: buckets + row partials);
// dw_scratch: conv 1's row-partial slice; dw1 / db1: conv 1's parameter gradients.  G0 must not alias gadd / y1.
int srf_bwd_level1_head(const float* G1, const float* d1, const srf_norm* n1, const void* n1_scratch, const float* y1,
                        const srf_norm* pn, const srf_norm* n0, const float* w0, const float* b0, const float* w1, const float* gadd,
                        float* G0, void* n0_scratch, void* dw_scratch, float* dw1, float* db1, int groups, int C, int L,
                        void* stream, SrfBwdCtx* ctx) {
  SRF_CHECK_ARG(G1 && d1 && n1 && n1_scratch && y1 && pn && n0 && w0 && b0 && w1 && gadd && G0 && n0_scratch && dw_scratch &&
                    G0 != gadd && G0 != y1, "srf_bwd_level1_head: bad pointers");
  SRF_CHECK_ARG((L % 4) == 0 && L >= 8, "srf_bwd_level1_head: L must be a multiple of 4");
  SRF_CHECK_ARG(srf_aligned16(G1) && srf_aligned16(d1) && srf_aligned16(y1) && srf_aligned16(gadd) && srf_aligned16(G0),
                "srf_bwd_level1_head: unaligned tensor");
  const long rows = (long)groups * C;
  SRF_CHECK_ARG(groups > 0 && C > 0 && rows < (1L << 31), "srf_bwd_level1_head: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  L1hArgs a;
  a.G1 = G1;
  a.d1 = d1;
  a.y1 = y1;
  a.gadd = gadd;
  a.G0 = G0;
  a.pn = srf_norm_dev(pn);
  a.n0 = srf_norm_dev(n0);
  a.n1 = srf_norm_dev(n1);
  a.inv0 = 1.0 / ((double)C * (double)L);
  a.inv1 = 1.0 / ((double)C * (double)(L >> 1));
  a.n1_bsums = reinterpret_cast<const double*>(n1_scratch);
  a.n0_bsums = reinterpret_cast<double*>(n0_scratch);
  a.n0_rowpart = reinterpret_cast<float*>(a.n0_bsums + (size_t)groups * SRF_STAT_BUCKETS * 2);
  a.w0 = w0;
  a.b0 = b0;
  a.w1 = w1;
  a.dw_rowpart = reinterpret_cast<float*>(dw_scratch);
  a.C = C;
  a.L = L;
  const bool defer = ctx && ctx->defer;
  if (!defer) SRF_CHECK_HIP(hipMemsetAsync(a.n0_bsums, 0, sizeof(double) * (size_t)groups * SRF_STAT_BUCKETS * 2, st));
  hipLaunchKernelGGL(srf_bwd_l1h_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a, rows);
  SRF_CHECK_LAUNCH("bwd_l1h", st);
  if (dw1 || db1) {
    if (defer) {
      ctx->dw.push_back(DwParamsDesc{a.dw_rowpart, dw1, db1, groups, C});
    } else {
      hipLaunchKernelGGL(srf_dwconv5_bwd_params_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((groups + 63) / 64)), dim3(256),
                         0, st, a.dw_rowpart, groups, C, dw1, db1);
      SRF_CHECK_LAUNCH("dwconv5_bwd_params", st);
    }
  }
  return SRF_OK;
}

// =============================================================================================
// Mask application and its backward (improved_sudormrf.py:296-298): v[b, s N + n, l] = relu(m[b, s N + n, l]) * e[b, n, l]
// The inference path fuses this into the mask GEMM's epilogue; training needs the pre-activation m afterwards,
// so the training forward writes m and applies the mask here.
//   g_m = g_v * e * [m > 0]            g_e[b,n,l] += sum_s g_v[b, s N + n, l] * relu(m[b, s N + n, l])
// =============================================================================================
__global__ __launch_bounds__(256) void srf_mask_apply_kernel(const float* __restrict__ m, const float* __restrict__ e,
                                                             float* __restrict__ v, int SA, int N, int L) {
  const long b = blockIdx.z;
  const int n = blockIdx.y;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const float ev = e[(b * N + n) * (long)L + l];
  for (int s = 0; s < SA; ++s) {
    const size_t idx = ((size_t)b * SA * N + (size_t)s * N + n) * L + l;
    v[idx] = fmaxf(m[idx], 0.f) * ev;
  }
}

__global__ __launch_bounds__(256) void srf_mask_bwd_kernel(const float* __restrict__ gv, const float* __restrict__ m,
                                                           const float* __restrict__ e, float* __restrict__ gm,
                                                           float* __restrict__ ge, int SA, int N, int L, int acc_ge) {
  const long b = blockIdx.z;
  const int n = blockIdx.y;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const size_t eidx = (b * N + n) * (size_t)L + l;
  const float ev = e[eidx];
  float accum = acc_ge ? ge[eidx] : 0.f;
  for (int s = 0; s < SA; ++s) {
    const size_t idx = ((size_t)b * SA * N + (size_t)s * N + n) * L + l;
    const float mv = m[idx], g = gv[idx];
    gm[idx] = mv > 0.f ? g * ev : 0.f;
    accum = fmaf(g, fmaxf(mv, 0.f), accum);
  }
  ge[eidx] = accum;
}

extern "C" int srf_mask_apply(const float* m, const float* enc, float* v, int Bt, int SA, int N, int L, void* stream) {
  SRF_CHECK_ARG(m && enc && v && Bt > 0 && SA > 0 && N > 0 && L > 0 && Bt <= 65535 && N <= 65535, "srf_mask_apply: bad arguments");
  dim3 grid((unsigned)((L + 255) / 256), (unsigned)N, (unsigned)Bt);
  hipLaunchKernelGGL(srf_mask_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, m, enc, v, SA, N, L);
  SRF_CHECK_LAUNCH("mask_apply", stream);
  return SRF_OK;
}

// gv, m, gm: [Bt, SA*N, L]; enc, genc: [Bt, N, L]; gm may alias gv; accumulate_genc != 0: genc += ...
extern "C" int srf_mask_bwd(const float* gv, const float* m, const float* enc, float* gm, float* genc, int accumulate_genc,
                            int Bt, int SA, int N, int L, void* stream) {
  SRF_CHECK_ARG(gv && m && enc && gm && genc && Bt > 0 && SA > 0 && N > 0 && L > 0 && Bt <= 65535 && N <= 65535,
                "srf_mask_bwd: bad arguments");
  dim3 grid((unsigned)((L + 255) / 256), (unsigned)N, (unsigned)Bt);
  hipLaunchKernelGGL(srf_mask_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gv, m, enc, gm, genc, SA, N, L,
                     accumulate_genc);
  SRF_CHECK_LAUNCH("mask_bwd", stream);
  return SRF_OK;
}

// =============================================================================================
// PReLU backward for a stand-alone PReLU (mask_net.0, improved_sudormrf.py:268): gx = gout * (x >= 0 ? 1 : a),
// d a += sum gout * x [x < 0].  gx may alias gout.
// =============================================================================================
// (round 4: a persistent grid with float4 accesses -- the first form's one block per 1 024 elements ended in 25 600 atomicAdds on
// ONE address for the mask net's PReLU at cfg 2: 236 us for 315 MB; now <= 2 048 of them)
__global__ __launch_bounds__(256) void srf_prelu_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                            const float* __restrict__ slope, float* __restrict__ gx,
                                                            float* dslope, long n, int vec) {
  __shared__ double red[4];
  const float a = slope[0];
  double acc = 0.0;
  const long stride = (long)gridDim.x * 256;
  if (vec) {                                    // n % 4 == 0, 16-byte aligned tensors
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 g = reinterpret_cast<const float4*>(gout)[i], xv = reinterpret_cast<const float4*>(x)[i];
      float4 o;
      o.x = xv.x < 0.f ? g.x * a : g.x;
      o.y = xv.y < 0.f ? g.y * a : g.y;
      o.z = xv.z < 0.f ? g.z * a : g.z;
      o.w = xv.w < 0.f ? g.w * a : g.w;
      acc += (xv.x < 0.f ? (double)g.x * (double)xv.x : 0.0) + (xv.y < 0.f ? (double)g.y * (double)xv.y : 0.0);
      acc += (xv.z < 0.f ? (double)g.z * (double)xv.z : 0.0) + (xv.w < 0.f ? (double)g.w * (double)xv.w : 0.0);
      reinterpret_cast<float4*>(gx)[i] = o;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      const float g = gout[i], xv = x[i];
      if (xv < 0.f) {
        acc += (double)g * (double)xv;
        gx[i] = g * a;
      } else {
        gx[i] = g;
      }
    }
  }
  acc = srf_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && dslope) {
    const double t = (red[0] + red[1]) + (red[2] + red[3]);
    if (t != 0.0) atomicAdd(dslope, (float)t);
  }
}

extern "C" int srf_prelu_bwd(const float* gout, const float* x, const float* slope, float* gx, float* dslope, long n,
                             void* stream) {
  SRF_CHECK_ARG(gout && x && slope && gx && n > 0, "srf_prelu_bwd: bad arguments");
  const int vec = (n % 4 == 0) && srf_aligned16(gout) && srf_aligned16(x) && srf_aligned16(gx) ? 1 : 0;
  const long work = vec ? n / 4 : n;
  long blocks = (work + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(srf_prelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gout, x, slope, gx,
                     dslope, n, vec);
  SRF_CHECK_LAUNCH("prelu_bwd", stream);
  return SRF_OK;
}

// =============================================================================================
// Frame gather: out[b, r*K + k, l] = src[b, r, hop*l + k - pad]  (0 outside [0,T); rows >= R*K up to rows_out are 0).
// Turns the encoder's weight gradient (improved_sudormrf.py:247-251: dW[n,k] = sum_{b,l} g_s[b,n,l] x[b, hop l + k - pad])
// and the decoder's backward (:272-279: g_v = W_d g_frames, dW_d = v g_frames^T with g_frames[(o,k), l] =
// g_out[o, hop l + k - pad]) into the pointwise GEMM / weight-gradient GEMM.
// =============================================================================================
__global__ __launch_bounds__(256) void srf_frames_gather_kernel(const float* __restrict__ src, float* __restrict__ out,
                                                                int R, int T, int K, int hop, int pad, int L,
                                                                int rows_out) {
  const long b = blockIdx.z;
  const int row = blockIdx.y;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  float v = 0.f;
  if (row < R * K) {
    const int r = row / K, k = row - r * K;
    const int t = hop * l + k - pad;
    if (t >= 0 && t < T) v = src[(b * R + r) * (long)T + t];
  }
  out[(b * rows_out + row) * (long)L + l] = v;
}

extern "C" int srf_frames_gather(const float* src, float* out, int Bt, int R, int T, int K, int hop, int pad, int L,
                                 int rows_out, void* stream) {
  SRF_CHECK_ARG(src && out && Bt > 0 && R > 0 && T > 0 && K > 0 && hop > 0 && L > 0 && rows_out >= R * K &&
                    Bt <= 65535 && rows_out <= 65535,
                "srf_frames_gather: bad arguments");
  dim3 grid((unsigned)((L + 255) / 256), (unsigned)rows_out, (unsigned)Bt);
  hipLaunchKernelGGL(srf_frames_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, out, R, T, K, hop, pad, L,
                     rows_out);
  SRF_CHECK_LAUNCH("frames_gather", stream);
  return SRF_OK;
}

// dst += src (gradient accumulation where no producer epilogue can do it)
__global__ __launch_bounds__(256) void srf_accumulate_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 a = *reinterpret_cast<float4*>(dst + i);
    const float4 b = *reinterpret_cast<const float4*>(src + i);
    a.x += b.x;
    a.y += b.y;
    a.z += b.z;
    a.w += b.w;
    *reinterpret_cast<float4*>(dst + i) = a;
  } else {
    for (long k = i; k < n; ++k) dst[k] += src[k];
  }
}

int srf_accumulate_launch(float* dst, const float* src, long n, hipStream_t st) {
  hipLaunchKernelGGL(srf_accumulate_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, dst, src, n);
  SRF_CHECK_LAUNCH("accumulate", st);
  return SRF_OK;
}
