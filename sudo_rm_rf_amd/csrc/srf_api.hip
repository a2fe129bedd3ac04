// Plan + whole-forward orchestration behind the C ABI (include/sudormrf_hip.h).
//   srf_forward  <-  SuDORMRF.forward            improved_sudormrf.py:283-301
//                    GroupCommSudoRmRf.forward   groupcomm_sudormrf_v2.py:302-322
// One immutable plan per (config, batch, T): parameter indices in state_dict() order, workspace
// carve-up, GlobLN statistic slots.  The library allocates nothing on the device: the caller hands
// in one workspace buffer.  All launches go to the caller's stream; nothing synchronises.
#include <string.h>

#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "srf_common.h"

// ---------------------------------------------------------------------------------------------
// error string / kernel mode
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static std::atomic<int> g_kernel_mode{0};   // process-wide diagnostics switch (tests, bench A/B); per-thread override below

void srf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
// per-thread override (srf_forward_train runs its GEMMs in exact fp32 by default, see srf_train.hip)
static thread_local int g_kernel_mode_override = -1;
int srf_kernel_mode() { return g_kernel_mode_override >= 0 ? g_kernel_mode_override : g_kernel_mode.load(std::memory_order_relaxed); }
int srf_kernel_mode_override(int mode) {
  const int prev = g_kernel_mode_override;
  g_kernel_mode_override = mode;
  return prev;
}
static std::atomic<int> g_debug_flags{0};   // process-wide diagnostics switch: kernel-variant A/B only, never set by the product path
int srf_debug_flags() { return g_debug_flags.load(std::memory_order_relaxed); }
extern "C" void srf_set_debug_flags(int f) { g_debug_flags.store(f, std::memory_order_relaxed); }

// ---- per-device caches (see srf_common.h) ----------------------------------------------------------
static std::mutex g_dev_mu;
static int g_dev_cus[SRF_MAX_DEVICES];
static long g_dev_occ[SRF_MAX_DEVICES][SRF_OCC_SLOTS];
static bool g_dev_occ_set[SRF_MAX_DEVICES][SRF_OCC_SLOTS];

int srf_current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}
int srf_device_cus() {
  const int dev = srf_current_device();
  if (dev < 0 || dev >= SRF_MAX_DEVICES) return 256;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  if (!g_dev_cus[dev]) {
    hipDeviceProp_t prop;
    g_dev_cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  return g_dev_cus[dev];
}
long srf_device_cached(int slot, long (*compute)(void*), void* arg) {
  const int dev = srf_current_device();
  if (dev < 0 || dev >= SRF_MAX_DEVICES || slot < 0 || slot >= SRF_OCC_SLOTS) return compute(arg);
  {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_dev_occ_set[dev][slot]) return g_dev_occ[dev][slot];
  }
  // evaluated OUTSIDE the lock (compute may call srf_device_cus(), which takes it); two threads racing here compute the
  // same value for the same device
  const long v = compute(arg);
  std::lock_guard<std::mutex> lk(g_dev_mu);
  g_dev_occ[dev][slot] = v;
  g_dev_occ_set[dev][slot] = true;
  return v;
}

// ---- in-library HIP-event profiler: one event after every kernel launch on the caller's stream ------
struct ProfMark {
  const char* name;
  hipEvent_t ev;
  hipStream_t st;     // the stream the event was recorded on (timeline mode: launches of several streams in one list)
};
static bool g_prof = false;
static std::vector<ProfMark> g_marks;
static std::vector<hipEvent_t> g_pool;
static std::vector<float> g_ms;

static thread_local int g_prof_hold = 0;   // > 0: launches of a multi-launch operation are not marked one by one
bool srf_profiling() { return g_prof && g_prof_hold == 0; }
void srf_prof_hold(int delta) { g_prof_hold += delta; }
void srf_prof_mark(const char* name, hipStream_t st) {
  hipEvent_t ev;
  if (!g_pool.empty()) {
    ev = g_pool.back();
    g_pool.pop_back();
  } else if (hipEventCreate(&ev) != hipSuccess) {
    return;
  }
  (void)hipEventRecord(ev, st);
  g_marks.push_back(ProfMark{name, ev, st});
}

extern "C" int srf_profile_begin(void* stream) {
  for (auto& m : g_marks) g_pool.push_back(m.ev);
  g_marks.clear();
  g_ms.clear();
  g_prof = true;
  srf_prof_mark("begin", (hipStream_t)stream);
  return SRF_OK;
}

extern "C" int srf_profile_end(void* stream, int* count) {
  g_prof = false;
  SRF_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  g_ms.assign(g_marks.size(), 0.f);
  for (size_t i = 1; i < g_marks.size(); ++i) {
    float ms = 0.f;
    SRF_CHECK_HIP(hipEventElapsedTime(&ms, g_marks[i - 1].ev, g_marks[i].ev));
    g_ms[i] = ms;
  }
  if (count) *count = g_marks.empty() ? 0 : (int)g_marks.size() - 1;
  return SRF_OK;
}

// Timeline view of the same marks (the forward AS TIMED runs on two streams, rocprofv3's kernel trace
// serialises them): mark i's completion time in ms since mark 0 ("begin") and the index of its stream in order of first
// appearance.  The caller synchronises the device before srf_profile_end when more than one stream was used.
extern "C" int srf_profile_timeline(int i, const char** name, float* t_ms, int* stream_index) {
  SRF_CHECK_ARG(i >= 0 && (size_t)(i + 1) < g_marks.size() && name && t_ms && stream_index, "srf_profile_timeline: bad index %d", i);
  float ms = 0.f;
  SRF_CHECK_HIP(hipEventElapsedTime(&ms, g_marks[0].ev, g_marks[i + 1].ev));
  int idx = 0;
  std::vector<hipStream_t> seen;
  for (size_t j = 1; j <= (size_t)(i + 1); ++j) {
    bool known = false;
    for (size_t k = 0; k < seen.size(); ++k)
      if (seen[k] == g_marks[j].st) {
        known = true;
        if (j == (size_t)(i + 1)) idx = (int)k;
      }
    if (!known) {
      if (j == (size_t)(i + 1)) idx = (int)seen.size();
      seen.push_back(g_marks[j].st);
    }
  }
  *name = g_marks[i + 1].name;
  *t_ms = ms;
  *stream_index = idx;
  return SRF_OK;
}

extern "C" int srf_profile_get(int i, const char** name, float* ms) {
  SRF_CHECK_ARG(i >= 0 && (size_t)(i + 1) < g_marks.size() && name && ms, "srf_profile_get: bad index %d", i);
  *name = g_marks[i + 1].name;
  *ms = g_ms[i + 1];
  return SRF_OK;
}

extern "C" const char* srf_last_error(void) { return g_err; }
extern "C" int srf_abi_version(void) { return SRF_ABI_VERSION; }
extern "C" void srf_set_kernel_mode(int mode) { g_kernel_mode.store((mode == 1 || mode == 2) ? mode : 0, std::memory_order_relaxed); }
extern "C" int srf_get_kernel_mode(void) { return g_kernel_mode.load(std::memory_order_relaxed); }

// Zero fill as a kernel of our own rather than hipMemsetAsync: the inference forward is replayed from a captured HIP graph
// for small batches (engine.py), and a graph made of kernel nodes only replayed correctly where one with memset nodes did
// not (NaN outputs after an unrelated host-synchronising copy, ROCm 7.0 runtime; tools/diag_graph.py).  16-byte granules.
__global__ __launch_bounds__(256) void srf_zero_kernel(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}
static int srf_zero_launch(void* p, size_t bytes, hipStream_t st) {
  SRF_CHECK_ARG(srf_aligned16(p) && bytes % 16 == 0, "internal: zero fill needs 16-byte granules");
  const size_t n16 = bytes / 16;
  if (!n16) return SRF_OK;
  const size_t blocks = (n16 + 255) / 256;
  hipLaunchKernelGGL(srf_zero_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, (uint4*)p, n16);
  SRF_CHECK_LAUNCH("zero_fill", st);
  return SRF_OK;
}

int srf_transpose_launch(const float* w, float* wt, int Ci, int M, hipStream_t st);
int srf_overlap_add_launch(const float* z, float* out, int Bt, int Co, int K, int L, int T, int nparts, const float* stats,
                           const float* wav, int mc, hipStream_t st);
bool srf_pw_conv_preadd_supported(int Cin, int Cout, int L, const void* const* ptrs, int nptrs);
int srf_pw_conv_preadd(const float* x, const float* q, const srf_norm* qnorm, float* u, const float* w, const float* bias,
                       float* y, int Bt, int Cin, int Cout, int L, double* out_sums, hipStream_t st);
bool srf_mask_decode_supported(int Bt, int Cin, int Cout, int L, int M);
size_t srf_mask_decode_pack_bytes(int Cout);
int srf_mask_decode_pack(const float* wd, void* dst, int Ci, int M, hipStream_t st);
int srf_mask_decode(const float* x, const float* w, const void* w_packed, const float* bias, const float* prelu,
                    const float* mul, int mul_channels, const void* wd_packed, float* zpart, int Bt, int Cin, int Cout, int L,
                    int M, hipStream_t st);
int srf_encoder_impl(const float* wav, const float* w, float* out, double* sums, int Bt, int A, int T, int N, int K, int L,
                     const float* in_stats, void* stream);
extern "C" int srf_wav_stats(const float* wav, float* stats, int rows, int T, void* stream);
extern "C" int srf_pw_conv_pair_supported(int Bt, int Cin1, int Cmid, int Cout2, int L);
extern "C" int srf_pw_conv_pair(const float* x, const void* w1_packed, const float* bias1, float* y, const srf_norm* in_norm,
                                const float* residual, const void* w2_packed, const float* bias2, float* y2, double* out_sums2,
                                int Bt, int Cin1, int Cmid, int Cout2, int L, void* stream);

// ---------------------------------------------------------------------------------------------
// decoder = transpose(weight) -> frame GEMM (K2) -> overlap-add + crop
// ---------------------------------------------------------------------------------------------
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t srf_decoder_scratch_floats(int Bt, int Ci, int Co, int K, int L) {
  const size_t M = (size_t)Co * K;
  // wt [M][Ci] | zero bias [M] | z [Bt][M][L]   (each section 64-float aligned)
  return align_up(M * Ci, 64) + align_up(M, 64) + align_up((size_t)Bt * M * L, 64);
}

static int srf_decoder_impl(const float* v, const float* w, float* out, int Bt, int Ci, int Co, int K, int L, int T,
                            float* scratch, const float* post_stats, const float* post_wav, int post_mc, void* stream);
extern "C" int srf_decoder(const float* v, const float* w, float* out, int Bt, int Ci, int Co, int K,
                           int L, int T, float* scratch, void* stream) {
  return srf_decoder_impl(v, w, out, Bt, Ci, Co, K, L, T, scratch, nullptr, nullptr, 0, stream);
}
// post_stats / post_wav / post_mc: the callers' rescale (+ mixture consistency) folded into the overlap-add (srf_separate)
static int srf_decoder_impl(const float* v, const float* w, float* out, int Bt, int Ci, int Co, int K, int L, int T,
                            float* scratch, const float* post_stats, const float* post_wav, int post_mc, void* stream) {
  SRF_CHECK_ARG(v && w && out && scratch, "srf_decoder: null pointer");
  SRF_CHECK_ARG(Bt > 0 && Ci > 0 && Co > 0 && L > 0 && T > 0, "srf_decoder: bad sizes");
  SRF_CHECK_ARG(K >= 3 && (K & 1), "srf_decoder: kernel size must be odd (got %d)", K);
  SRF_CHECK_ARG(T <= (K / 2) * L, "srf_decoder: T=%d exceeds hop*L=%d", T, (K / 2) * L);
  SRF_CHECK_ARG(Co <= 65535 && Bt <= 65535, "srf_decoder: too many channels / batch");
  hipStream_t st = (hipStream_t)stream;
  const int M = Co * K;
  float* wt = scratch;
  float* zb = wt + align_up((size_t)M * Ci, 64);
  float* z = zb + align_up((size_t)M, 64);
  int rc = srf_transpose_launch(w, wt, Ci, M, st);
  if (rc) return rc;
  rc = srf_zero_launch(zb, sizeof(float) * align_up((size_t)M, 64), st);
  if (rc) return rc;
  rc = srf_pw_conv(v, wt, zb, z, Bt, Ci, M, L, nullptr, nullptr, nullptr, 0, nullptr, 0, stream);
  if (rc) return rc;
  return srf_overlap_add_launch(z, out, Bt, Co, K, L, T, 1, post_stats, post_wav, post_mc, st);
}

// ---------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------
#include "srf_plan.h"

static int plan_fail(srf_plan* p, int rc) {
  delete p;
  return rc;
}

extern "C" int srf_plan_create(const srf_config* c, int batch, int T, srf_plan** out) {
  SRF_CHECK_ARG(c && out, "srf_plan_create: null pointer");
  *out = nullptr;
  SRF_CHECK_ARG(c->variant == SRF_VARIANT_IMPROVED || c->variant == SRF_VARIANT_GROUPCOMM,
                "srf_plan_create: unknown variant %d", c->variant);
  SRF_CHECK_ARG(batch > 0 && T > 0, "srf_plan_create: batch and T must be positive");
  SRF_CHECK_ARG(c->out_channels > 0 && c->in_channels > 0 && c->num_blocks > 0 && c->enc_num_basis > 0 &&
                    c->num_sources > 0,
                "srf_plan_create: non-positive model dimension");
  SRF_CHECK_ARG(c->upsampling_depth >= 1 && c->upsampling_depth <= SRF_MAX_DEPTH,
                "srf_plan_create: upsampling_depth %d unsupported (1..%d)", c->upsampling_depth,
                SRF_MAX_DEPTH);
  SRF_CHECK_ARG(c->enc_kernel_size >= 3 && (c->enc_kernel_size & 1),
                "srf_plan_create: enc_kernel_size must be odd (the reference's mask multiply breaks "
                "for even sizes)");
  srf_plan* p = new (std::nothrow) srf_plan();
  SRF_CHECK_ARG(p != nullptr, "srf_plan_create: out of host memory");
  p->cfg = *c;
  const bool gc = c->variant == SRF_VARIANT_GROUPCOMM;
  const int G = gc ? c->group_size : 1;
  p->A = gc ? c->in_audio_channels : 1;
  if (G <= 0 || p->A <= 0) {
    srf_set_error("srf_plan_create: group_size / in_audio_channels must be positive");
    return plan_fail(p, SRF_EINVAL);
  }
  if (gc && (c->out_channels % G || c->in_channels % G)) {
    srf_set_error("srf_plan_create: channels (%d,%d) not divisible by group_size %d", c->out_channels,
                  c->in_channels, G);
    return plan_fail(p, SRF_EINVAL);
  }
  if (gc) {
    const int n = c->out_channels / G;
    if (!(n == 2 || n == 4 || n == 8 || n == 16 || n == 32)) {
      srf_set_error("srf_plan_create: out_channels/group_size = %d unsupported (2,4,8,16,32)", n);
      return plan_fail(p, SRF_EINVAL);
    }
  }
  const int D = c->upsampling_depth, U = c->num_blocks, K = c->enc_kernel_size;
  const int h = K / 2;
  const long nls = (long)h << D;  // n_least_samples_req, improved_sudormrf.py:244
  long Tp = (T < nls) ? nls : ((T / nls) + (T % nls ? 1 : 0)) * nls;
  p->Bt = batch;
  p->T = T;
  p->Tp = (int)Tp;
  p->L = (int)((Tp + 2 * h - K) / h + 1);
  p->SA = c->num_sources * p->A;
  p->Bg = batch * G;
  p->nB = c->out_channels / G;
  p->nC = c->in_channels / G;
  if (p->L % (1 << (D - 1)) != 0) {
    srf_set_error("srf_plan_create: internal: L=%d not divisible by 2^(D-1)", p->L);
    return plan_fail(p, SRF_EINVAL);
  }
  if ((long)p->Bg > 65535) {
    srf_set_error("srf_plan_create: batch*group_size=%d too large (max 65535 per call)", p->Bg);
    return plan_fail(p, SRF_EINVAL);
  }
  // ---- parameter indexing (state_dict order)
  const int ublock_params = 10 + 4 * D;
  p->p_block0 = 5;
  p->p_ublock_off = gc ? 11 : 0;
  p->p_block_stride = ublock_params + p->p_ublock_off;
  p->p_tail = 5 + U * p->p_block_stride;
  p->n_params = p->p_tail + 4;
  // ---- statistic slots: ln | per block: [tac] proj d0..d{D-1} merged
  p->slots_per_block = D + 2 + (gc ? 1 : 0);
  p->n_slots = 1 + U * p->slots_per_block;
  // ---- workspace
  const size_t F = sizeof(float);
  const size_t L = p->L;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  p->stats_bytes = (size_t)p->n_slots * p->Bg * SRF_STAT_BUCKETS * 2 * sizeof(double);
  p->off_stats = take(p->stats_bytes);
  p->off_enc = take(F * batch * c->enc_num_basis * L);
  p->off_xa = take(F * batch * c->out_channels * L);
  p->off_xb = take(F * batch * c->out_channels * L);
  p->off_xq = gc ? take(F * batch * c->out_channels * L) : 0;
  p->off_xu = gc ? take(F * batch * c->out_channels * L) : 0;
  p->off_y1 = take(F * batch * c->in_channels * L);
  for (int k = 0; k < D; ++k) p->off_lv[k] = take(F * batch * c->in_channels * (L >> k));
  p->off_masked = take(F * batch * p->SA * c->enc_num_basis * L);
  p->off_dec = take(F * srf_decoder_scratch_floats(batch, p->SA * c->enc_num_basis, p->SA, K, p->L));
  p->fused_pyramid = srf_pyramid_supported(p->nC, p->L, D);   // (evaluated with the default debug flags)
  p->off_pyr = p->fused_pyramid ? take(srf_pyramid_scratch_bytes(p->Bg, p->nC, p->L, D)) : 0;
  // packed weights for the split-precision GEMM (only shapes the kernel supports)
  p->pk_of_param.assign(p->n_params, 0);
  auto add_pack = [&](int param, int cout, int cin) {
    const size_t bytes = srf_packed_pw_weight_bytes(cout, cin);
    if (!bytes) return;
    const size_t o = take(bytes);
    p->pk_param.push_back(param);
    p->pk_cout.push_back(cout);
    p->pk_cin.push_back(cin);
    p->pk_off.push_back(o);
    p->pk_of_param[param] = o;
  };
  add_pack(3, c->out_channels, c->enc_num_basis);
  for (int i = 0; i < U; ++i) {
    const int pu = p->p_block0 + i * p->p_block_stride + p->p_ublock_off;
    add_pack(pu + 0, p->nC, p->nB);
    add_pack(pu + 5 + 4 * D + 3, p->nB, p->nC);
  }
  add_pack(p->p_tail + 1, p->SA * c->enc_num_basis, c->out_channels);
  // decoder weights as MFMA fragments for the fused tail (K5); whether it runs is decided per forward
  p->off_wdpack = (p->SA * K <= 64 && p->pk_of_param[p->p_tail + 1]) ? take(srf_mask_decode_pack_bytes(p->SA * c->enc_num_basis)) : 0;
  p->total_bytes = off;
  p->n_launches = 1 /*memset*/ + 2 + U * (D + 3 + (gc ? 2 : 0)) + 1 + 4;
  *out = p;
  return SRF_OK;
}

extern "C" void srf_plan_destroy(srf_plan* p) { delete p; }
extern "C" size_t srf_plan_workspace_bytes(const srf_plan* p) { return p ? p->total_bytes : 0; }
extern "C" int srf_plan_num_params(const srf_plan* p) { return p ? p->n_params : 0; }
extern "C" int srf_plan_frames(const srf_plan* p) { return p ? p->L : 0; }
extern "C" int srf_plan_padded_length(const srf_plan* p) { return p ? p->Tp : 0; }
extern "C" int srf_plan_num_launches(const srf_plan* p) { return p ? p->n_launches : 0; }

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
static int srf_forward_impl(const srf_plan* p, const float* const* P, int num_params, const float* wav, float* out,
                            void* workspace, size_t workspace_bytes, const float* wav_stats, int mixture_consistency,
                            void* stream);
extern "C" int srf_forward(const srf_plan* p, const float* const* P, int num_params, const float* wav,
                           float* out, void* workspace, size_t workspace_bytes, void* stream) {
  return srf_forward_impl(p, P, num_params, wav, out, workspace, workspace_bytes, nullptr, 0, stream);
}

// The reference's caller-side inference recipe as ONE pass (README.md:100-114, experiments/simple_whamr_evaluation.py:
// 142-148): stats = per-row {mean, unbiased std} of the raw mixture (srf_wav_stats), the encoder normalises on load, the
// decoder's overlap-add rescales the estimates with the mixture's statistics and -- mixture_consistency != 0, what the README
// prescribes for the GroupComm models -- applies mixture_consistency.apply against the normalised mixture.  wav: the RAW
// mixture [Bt, in_audio_channels, T]; stats: [Bt * in_audio_channels][2] device floats (written here).
extern "C" int srf_separate(const srf_plan* p, const float* const* P, int num_params, const float* wav, float* out,
                            float* stats, int mixture_consistency, void* workspace, size_t workspace_bytes, void* stream) {
  SRF_CHECK_ARG(p && wav && stats, "srf_separate: null pointer");
  SRF_CHECK_ARG(p->A == 1, "srf_separate: the recipe is defined for single-channel mixtures (in_audio_channels = %d)", p->A);
  int rc = srf_wav_stats(wav, stats, p->Bt * p->A, p->T, stream);
  if (rc) return rc;
  return srf_forward_impl(p, P, num_params, wav, out, workspace, workspace_bytes, stats, mixture_consistency, stream);
}

static int srf_forward_body(const srf_plan* p, const float* const* P, int num_params, const float* wav, float* out,
                            void* workspace, size_t workspace_bytes, const float* wav_stats, int mixture_consistency,
                            void* stream);
void srf_pw_prefer_paired(bool on);   // srf_pwconv.hip: the paired-block form of the 256 x 128 GEMM for this thread's launches
static int srf_forward_impl(const srf_plan* p, const float* const* P, int num_params, const float* wav, float* out,
                            void* workspace, size_t workspace_bytes, const float* wav_stats, int mixture_consistency,
                            void* stream) {
  // The inference forward is what the engine runs as two sub-batches on two streams: its GEMMs take the half-CU blocks of
  // srf_pwconv_x3p.hip so that the other stream's kernels can co-reside (srf_pwconv.hip, srf_pw_256_launch).
  struct Paired {
    Paired() { srf_pw_prefer_paired(true); }
    ~Paired() { srf_pw_prefer_paired(false); }
  } paired_for_this_call;
  return srf_forward_body(p, P, num_params, wav, out, workspace, workspace_bytes, wav_stats, mixture_consistency, stream);
}
static int srf_forward_body(const srf_plan* p, const float* const* P, int num_params, const float* wav, float* out,
                            void* workspace, size_t workspace_bytes, const float* wav_stats, int mixture_consistency,
                            void* stream) {
  SRF_CHECK_ARG(p && P && wav && out && workspace, "srf_forward: null pointer");
  SRF_CHECK_ARG(num_params == p->n_params, "srf_forward: expected %d parameter tensors, got %d",
                p->n_params, num_params);
  if (workspace_bytes < p->total_bytes) {
    srf_set_error("srf_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, p->total_bytes);
    return SRF_EWORKSPACE;
  }
  SRF_CHECK_ARG((((size_t)workspace) & 255) == 0, "srf_forward: workspace must be 256-byte aligned");
  for (int i = 0; i < num_params; ++i) SRF_CHECK_ARG(P[i] != nullptr, "srf_forward: parameter %d is null", i);
  // profiler: intervals run from mark to mark, so without this one the first kernel's interval would also hold the
  // host-side gap since the previous forward
  if (srf_profiling()) srf_prof_mark("(gap)", (hipStream_t)stream);

  const srf_config& c = p->cfg;
  const bool gc = c.variant == SRF_VARIANT_GROUPCOMM;
  const int G = gc ? c.group_size : 1;
  const int D = c.upsampling_depth, U = c.num_blocks, N = c.enc_num_basis, K = c.enc_kernel_size;
  const int Bt = p->Bt, L = p->L, Bg = p->Bg, nB = p->nB, nC = p->nC;
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  auto fptr = [&](size_t o) { return (float*)(ws + o); };
  double* stats = (double*)(ws + p->off_stats);
  auto slot = [&](int s) { return stats + (size_t)s * Bg * SRF_STAT_BUCKETS * 2; };
  int rc;

  rc = srf_zero_launch(stats, p->stats_bytes, st);
  if (rc) return rc;
  // split + lay out every 1x1 weight for the 256 x 128 split-precision GEMMs (srf_pwconv_x3w.hip / _x3p.hip), one launch per form and forward
  // (kernel mode 0 only; debug flag 8 = without: the 128 x 128 kernels that split W on the fly)
  const bool use_pack = srf_kernel_mode() == 0 && !(srf_debug_flags() & 8) && !p->pk_param.empty();
  if (use_pack) {
    std::vector<const float*> pw(p->pk_param.size());
    std::vector<void*> pd(p->pk_param.size());
    for (size_t i = 0; i < p->pk_param.size(); ++i) {
      pw[i] = P[p->pk_param[i]];
      pd[i] = ws + p->pk_off[i];
    }
    rc = srf_pack_pw_weights(pw.data(), pd.data(), p->pk_cout.data(), p->pk_cin.data(), (int)pw.size(), stream);
    if (rc) return rc;
  }
  auto packed = [&](int param_index) -> const void* {
    return (use_pack && p->pk_of_param[param_index]) ? (const void*)(ws + p->pk_of_param[param_index]) : nullptr;
  };

  // ---- front end: encoder (+ ln statistics), ln folded into the bottleneck GEMM's operand load
  float* enc = fptr(p->off_enc);
  rc = srf_encoder_impl(wav, P[0], enc, slot(0), Bt, p->A, p->T, N, K, L, wav_stats, stream);
  if (rc) return rc;
  float* cur = fptr(p->off_xa);
  float* nxt = fptr(p->off_xb);
  float* y1 = fptr(p->off_y1);
  // Round 5: a 1x1 conv with 256 output channels and the proj_1x1 that consumes its output run as ONE launch, the 256-channel
  // tensor handed over in registers (srf_pwconv_x3f.hip): bottleneck -> proj_1x1 of block 0, res_conv of block i -> proj_1x1 of
  // block i + 1 (improved_sudormrf.py:292 -> :205, :220 -> :205).  Needs the fused pyramid (its merged tensor has a buffer of
  // its own: the pair kernel writes y1 while it reads the merged tensor).  Debug flag 1 = separate launches.
  const bool pyr_fused_now = p->fused_pyramid && srf_kernel_mode() != 1 && !(srf_debug_flags() & 16);
  const int pu0 = p->p_block0 + p->p_ublock_off;
  const bool pair_res = !gc && use_pack && pyr_fused_now && packed(pu0) && packed(pu0 + 5 + 4 * D + 3) &&
                        srf_pw_conv_pair_supported(Bt, nC, nB, nC, L);
  const bool pair_head = pair_res && packed(3) && srf_pw_conv_pair_supported(Bt, N, nB, nC, L);
  bool y1_ready = false;      // proj_1x1 of the coming block has already been computed (with its statistics) by a pair launch
  {
    srf_norm ln{slot(0), P[1], P[2], nullptr};
    if (pair_head) {
      rc = srf_pw_conv_pair(enc, packed(3), P[4], cur, &ln, nullptr, packed(pu0), P[pu0 + 1], y1, slot(1), Bt, N, nB, nC, L, stream);
      y1_ready = true;
    } else {
      rc = srf_pw_conv_packed(enc, P[3], packed(3), P[4], cur, Bt, N, c.out_channels, L, &ln, nullptr, nullptr,
                              0, nullptr, 0, stream);
    }
    if (rc) return rc;
  }

  // ---- separation module
  for (int i = 0; i < U; ++i) {
    const float* const* Pb = P + p->p_block0 + (size_t)i * p->p_block_stride;
    const float* const* Pu = Pb + p->p_ublock_off;
    int s0 = 1 + i * p->slots_per_block;
    const float* xin = cur;
    bool tac_norm_fused = false;
    if (gc) {
      // TAC (groupcomm_sudormrf_v2.py:356-384): q = TAC MLPs, u = x + GlobLN_(b,g)(q)
      float* xq = fptr(p->off_xq);
      float* xu = fptr(p->off_xu);
      rc = srf_tac(cur, xq, Pb, Bt, G, nB, 3 * nB, L, slot(s0), stream);
      if (rc) return rc;
      srf_norm tn{slot(s0), Pb[9], Pb[10], nullptr};
      // u = x + GlobLN(q): folded into the proj conv's operand load where the thin-shape kernel runs it (it writes u for
      // the block's residual as it goes); else its own kernel
      const void* al[4] = {cur, xq, xu, y1};
      tac_norm_fused = srf_pw_conv_preadd_supported(nB, nC, L, al, 4);
      if (tac_norm_fused) {
        rc = srf_pw_conv_preadd(cur, xq, &tn, xu, Pu[0], Pu[1], y1, Bg, nB, nC, L, slot(s0 + 1), st);
      } else {
        rc = srf_gln_apply_add(cur, xq, xu, &tn, Bg, nB, L, stream);
      }
      if (rc) return rc;
      xin = xu;
      s0 += 1;
    }
    // proj_1x1 conv (+ statistics for its GlobLN)            improved_sudormrf.py:205
    const int pu_index = p->p_block0 + i * p->p_block_stride + p->p_ublock_off;
    if (!tac_norm_fused && !y1_ready) {
      rc = srf_pw_conv_packed(xin, Pu[0], packed(pu_index), Pu[1], y1, Bg, nB, nC, L, nullptr, nullptr, slot(s0),
                              0, nullptr, 0, stream);
      if (rc) return rc;
    }
    y1_ready = false;
    // depthwise pyramid + upsample/add                         :206-216
    // unfused path: the merged tensor aliases y1 (dead once every level has been produced); fused path:
    // its own buffer (the otherwise unused level-0 buffer), because pass 2 re-reads y1 with halos
    const bool fused = p->fused_pyramid && srf_kernel_mode() != 1 && !(srf_debug_flags() & 16);
    float* merged = fused ? fptr(p->off_lv[0]) : y1;
    if (fused) {
      // two passes with every level kept on chip (srf_pyramid.hip)
      const float *pw[SRF_MAX_DEPTH], *pb[SRF_MAX_DEPTH], *pg[SRF_MAX_DEPTH], *pbe[SRF_MAX_DEPTH];
      for (int k = 0; k < D; ++k) {
        const float* const* Pk = Pu + 5 + 4 * k;
        pw[k] = Pk[0];
        pb[k] = Pk[1];
        pg[k] = Pk[2];
        pbe[k] = Pk[3];
      }
      srf_norm in{slot(s0), Pu[2], Pu[3], Pu[4]};
      rc = srf_pyramid(y1, merged, &in, pw, pb, pg, pbe, Bg, nC, L, D, ws + p->off_pyr, slot(s0 + 1 + D), stream);
      if (rc) return rc;
    } else {
    const float* levels[SRF_MAX_DEPTH];
    srf_norm norms[SRF_MAX_DEPTH];
    for (int k = 0; k < D; ++k) {
      const float* const* Pk = Pu + 5 + 4 * k;
      float* dk = fptr(p->off_lv[k]);
      srf_norm in;
      const float* src;
      int Lin, stride;
      if (k == 0) {
        in = srf_norm{slot(s0), Pu[2], Pu[3], Pu[4]};  // proj_1x1.norm + act
        src = y1;
        Lin = L;
        stride = 1;
      } else {
        const float* const* Pprev = Pu + 5 + 4 * (k - 1);
        in = srf_norm{slot(s0 + k), Pprev[2], Pprev[3], nullptr};  // previous level's norm
        src = fptr(p->off_lv[k - 1]);
        Lin = L >> (k - 1);
        stride = 2;
      }
      rc = srf_dwconv5(src, Pk[0], Pk[1], dk, Bg, nC, Lin, stride, &in, slot(s0 + 1 + k), stream);
      if (rc) return rc;
      levels[k] = dk;
      norms[k] = srf_norm{slot(s0 + 1 + k), Pk[2], Pk[3], nullptr};
    }
    // upsample + add                                          :214-216   (output aliases y1: dead)
    rc = srf_merge(levels, norms, D, merged, Bg, nC, L, slot(s0 + 1 + D), stream);
    if (rc) return rc;
    }
    // final_norm + PReLU folded into res_conv, + residual     :218-220
    const float* const* Pf = Pu + 5 + 4 * D;
    srf_norm fn{slot(s0 + 1 + D), Pf[0], Pf[1], Pf[2]};
    if (pair_res && fused && i + 1 < U) {
      // res_conv of this block + proj_1x1 of the next one (its output into y1 -- dead since this block's pyramid -- and its
      // statistics into the next block's first slot)
      const int pn = pu_index + p->p_block_stride;
      rc = srf_pw_conv_pair(merged, packed(pu_index + 5 + 4 * D + 3), Pf[4], nxt, &fn, xin, packed(pn), P[pn + 1], y1,
                            slot(1 + (i + 1) * p->slots_per_block), Bt, nC, nB, nC, L, stream);
      y1_ready = true;
    } else {
      rc = srf_pw_conv_packed(merged, Pf[3], packed(pu_index + 5 + 4 * D + 3), Pf[4], nxt, Bg, nC, nB, L, &fn, xin,
                              nullptr, 0, nullptr, 0, stream);
    }
    if (rc) return rc;
    float* t = cur;
    cur = nxt;
    nxt = t;
  }

  // ---- mask estimation + decoder                            :295-301
  const float* const* Pt = P + p->p_tail;
  float* masked = fptr(p->off_masked);
  // K5: mask GEMM and decoder contraction in ONE launch -- the [Bt, S N, L] masked tensor (the largest of the forward) never
  // reaches HBM; the GEMM leaves per-256-channel partial decoder frames (in the masked tensor's workspace region, <= 1/4 of
  // it) and the overlap-add sums them.  Only where the 256 x 128 GEMM would have run the mask conv anyway.
  if (p->off_wdpack && use_pack && packed(p->p_tail + 1) && N % 8 == 0 &&
      srf_mask_decode_supported(Bt, c.out_channels, p->SA * N, L, p->SA * K)) {
    const int M = p->SA * K, nparts = (p->SA * N + 255) / 256;
    rc = srf_mask_decode_pack(Pt[3], ws + p->off_wdpack, p->SA * N, M, st);
    if (rc) return rc;
    rc = srf_mask_decode(cur, Pt[1], packed(p->p_tail + 1), Pt[2], Pt[0], enc, N, ws + p->off_wdpack, masked, Bt,
                         c.out_channels, p->SA * N, L, M, st);
    if (rc) return rc;
    return srf_overlap_add_launch(masked, out, Bt, p->SA, K, L, p->T, nparts, wav_stats, wav, mixture_consistency, st);
  }
  {
    srf_norm pre{nullptr, nullptr, nullptr, Pt[0]};
    rc = srf_pw_conv_packed(cur, Pt[1], packed(p->p_tail + 1), Pt[2], masked, Bt, c.out_channels, p->SA * N, L,
                            &pre, nullptr, nullptr, 1, enc, N, stream);
    if (rc) return rc;
  }
  rc = srf_decoder_impl(masked, Pt[3], out, Bt, p->SA * N, p->SA, K, L, p->T, fptr(p->off_dec), wav_stats, wav,
                        mixture_consistency, stream);
  return rc;
}

extern "C" int srf_debug_fetch(const srf_plan* p, const void* workspace, int what, float* dst,
                               size_t dst_floats, void* stream) {
  SRF_CHECK_ARG(p && workspace && dst, "srf_debug_fetch: null pointer");
  const srf_config& c = p->cfg;
  const char* ws = (const char*)workspace;
  size_t off = 0, n = 0;
  if (what == 0) {
    off = p->off_enc;
    n = (size_t)p->Bt * c.enc_num_basis * p->L;
  } else if (what == 1) {
    off = (c.num_blocks % 2 == 0) ? p->off_xa : p->off_xb;
    n = (size_t)p->Bt * c.out_channels * p->L;
  } else if (what == 2) {
    // the fused tail (K5) never materialises the masked tensor: its workspace region holds partial decoder frames
    if (p->off_wdpack && srf_kernel_mode() == 0 && !(srf_debug_flags() & 8) && c.enc_num_basis % 8 == 0 &&
        srf_mask_decode_supported(p->Bt, c.out_channels, p->SA * c.enc_num_basis, p->L, p->SA * c.enc_kernel_size)) {
      srf_set_error("srf_debug_fetch: the masked tensor is not materialised at this shape (mask GEMM and decoder run fused); "
                    "set debug flag 32768 to run the tail unfused");
      return SRF_EINVAL;
    }
    off = p->off_masked;
    n = (size_t)p->Bt * p->SA * c.enc_num_basis * p->L;
  } else {
    srf_set_error("srf_debug_fetch: unknown selector %d", what);
    return SRF_EINVAL;
  }
  SRF_CHECK_ARG(dst_floats >= n, "srf_debug_fetch: destination too small (%zu < %zu)", dst_floats, n);
  SRF_CHECK_HIP(hipMemcpyAsync(dst, ws + off, n * sizeof(float), hipMemcpyDeviceToDevice,
                               (hipStream_t)stream));
  return SRF_OK;
}
