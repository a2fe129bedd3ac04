/*
 * sudormrf_hip.h -- C ABI of libsudormrf_hip.so, the MI355X (gfx950) hot path of
 * SuDoRM-RF (Improved SuDORMRF and GroupComm SuDoRM-RF v2) inference forward.
 *
 * Boundary replaced (reference is pure PyTorch, paths relative to
 * /root/reference/sudo_rm_rf/dnn/):
 *   srf_forward            <- SuDORMRF.forward            models/improved_sudormrf.py:283-301
 *                             GroupCommSudoRmRf.forward   models/groupcomm_sudormrf_v2.py:302-322
 *   srf_encoder            <- self.encoder (nn.Conv1d)    models/improved_sudormrf.py:247-251,286
 *                             + pad_to_appropriate_length :303-314 (folded into bounds checks)
 *   srf_gln_stats/_apply   <- GlobLN.forward              models/improved_sudormrf.py:30-47
 *   srf_pw_conv            <- nn.Conv1d(kernel_size=1) sites :256-259 (bottleneck), :174 (proj_1x1),
 *                             :196,:220 (res_conv + residual), :268-269,:295-298 (mask_net + ReLU + *s)
 *                             with the neighbouring GlobLN / PReLU folded into prologue / epilogue
 *   srf_dwconv5            <- DilatedConvNorm.conv (depthwise k=5, stride 1|2) :152-153,:206-211
 *   srf_merge              <- Upsample(x2 nearest) + add loop          :190-194,:214-216
 *   srf_decoder            <- self.decoder (nn.ConvTranspose1d)        :272-279,:300 + crop :316-318
 *   srf_tac                <- TAC.forward                 models/groupcomm_sudormrf_v2.py:356-377
 *   srf_gln_apply_add      <- TAC_norm + residual add     models/groupcomm_sudormrf_v2.py:378-382
 *   srf_mixture_consistency<- mixture_consistency.apply   experiments/utils/mixture_consistency.py:14-36
 *   srf_wav_normalize / srf_wav_denormalize <- the callers' normalise / rescale lines   README.md:100-114
 *   srf_pit_sisdr_*        <- PITLossWrapper(PairwiseNegSDR("sisdr")) fwd/bwd        losses/sisdr.py:254-311,426-458
 *   srf_perm_inv_sisdr     <- PermInvariantSISDR.forward (validation metric)        losses/sisdr.py:66-196
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (or fp64 for GlobLN sums) owned by the
 *     caller; the library allocates nothing on the device; global state = a thread-local error string,
 *     the kernel-mode switch and the (off by default) profiler;
 *   - activations are [batch, channel, time] contiguous, exactly as the reference's tensors;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is asynchronous
 *     on that stream, nothing synchronises;
 *   - return value: 0 on success, a negative SRF_E* code otherwise (srf_last_error() has the text);
 *     nothing throws across the ABI;
 *   - GlobLN statistics travel as fp64 {sum, sum_of_squares} pairs ("sums",
 *     [groups][SRF_STAT_BUCKETS][2]); producers ACCUMULATE into them (atomics), so the caller zeroes
 *     them first (srf_forward does it itself).
 */
#ifndef SUDORMRF_HIP_H
#define SUDORMRF_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRF_ABI_VERSION 15

/* GlobLN statistics layout: "sums" = fp64 [groups][SRF_STAT_BUCKETS][2] {sum, sum of squares}; the
 * statistic of a group is the total over its buckets (producers spread their atomics over buckets). */
#define SRF_STAT_BUCKETS 64

#define SRF_OK 0
#define SRF_EINVAL (-1)   /* bad argument / unsupported shape */
#define SRF_EHIP (-2)     /* a HIP runtime call or kernel launch failed */
#define SRF_EWORKSPACE (-3) /* workspace too small */

#define SRF_VARIANT_IMPROVED 0
#define SRF_VARIANT_GROUPCOMM 1

/* Constructor arguments of the reference models, same meaning
 * (improved_sudormrf.py:224-231, groupcomm_sudormrf_v2.py:232-241). */
typedef struct srf_config {
  int variant;           /* SRF_VARIANT_* */
  int in_audio_channels; /* 1 for Improved */
  int out_channels;      /* B */
  int in_channels;       /* C */
  int num_blocks;        /* U */
  int upsampling_depth;  /* D */
  int enc_kernel_size;   /* K (odd) */
  int enc_num_basis;     /* N */
  int num_sources;       /* S */
  int group_size;        /* G (1 for Improved) */
} srf_config;

/* "Apply GlobLN (+ optional PReLU) to this tensor when it is loaded". */
typedef struct srf_norm {
  const double* sums;  /* [groups][SRF_STAT_BUCKETS][2] over (channel,time); NULL = no normalisation */
  const float* gamma;  /* [channels] */
  const float* beta;   /* [channels] */
  const float* prelu;  /* [1] shared slope, or NULL = no activation */
} srf_norm;

typedef struct srf_plan srf_plan;

int srf_abi_version(void);
const char* srf_last_error(void);

/* Kernel-variant switch for A/B measurements:
 *   0 = fast paths where the shape allows (default); 1x1 convs run as split-precision MFMA GEMMs
 *       (each fp32 operand = bf16 hi + bf16 lo, three bf16 MFMAs per product block, fp32 accumulate);
 *   1 = force the generic (shape-agnostic, scalar-load, fp32 FMA) kernels everywhere;
 *   2 = fast paths, but 1x1 convs on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32). */
void srf_set_kernel_mode(int mode);
int srf_get_kernel_mode(void);
/* In-library profiler (bench.py): between begin/end every kernel launched through this library is
 * followed by a HIP event on the caller's stream; end() synchronises the stream and get(i) returns
 * the kernel family name and the elapsed ms between the previous event and launch i's event
 * (i.e. that launch's duration including its launch gap).  Not thread-safe; off by default. */
int srf_profile_begin(void* stream);
int srf_profile_end(void* stream, int* count);
int srf_profile_get(int i, const char** name, float* ms);
/* The same marks as a timeline: completion time of launch i in ms since srf_profile_begin and the index (order of first
 * appearance) of the stream it ran on -- for forwards whose sub-batches run on several streams (synchronise the device before
 * srf_profile_end); tools/two_stream_events.py. */
int srf_profile_timeline(int i, const char** name, float* t_ms, int* stream_index);

/* ---- SRF_DIAGNOSTICS -----------------------------------------------------------------------------------------------------
 * NOT part of the drop-in surface: process-wide switches between kernel variants for A/B measurements and bisection
 * (tools/, bench.py --debug-flags, a handful of tests).  They act on every thread's subsequent launches; a caller that does
 * not define SRF_DIAGNOSTICS before including this header does not see them.  Default 0 = the shipped paths.
 *   1        srf_forward WITHOUT the fused conv pairs (round 5: res_conv / bottleneck + the next proj_1x1 in one launch)
 *   2        256 x 128 GEMMs: no m-tile groups (round 2's tile order); paired-block form: plain cache policy
 *   4        without the 256 x 128 GEMM (128 x 128 kernels)
 *   8        WITHOUT pre-packed weights (srf_forward packs by default)
 *   16       per-level depthwise + merge kernels instead of the fused pyramid (inference and training)
 *   32 / 64  LDS pyramid kernels instead of the register ones      128       non-persistent pyramid pass 1
 *   256      leftover GEMM tiles as whole tiles (no quarter tiles) 512       quarter tiles last
 *   1024     TAC forward with one time step per lane               2048      one-tile-per-block 128 x 128 GEMM everywhere
 *                                                                            (also: no 64 x 64 tiles for small launches)
 *   4096     weight-gradient GEMM: round 3's block -> (tile, partial) mapping (every XCD re-reads its rows through its own L2)
 *   8192     swap the two forms of the 256 x 128 GEMM: srf_forward / srf_separate run the one-block-per-CU kernel (srf_pwconv_x3w.hip),
 *            every other caller the paired-block kernel (srf_pwconv_x3p.hip) -- default: the paired form inside the forward only
 *   16384    training forward: three bf16 parts per operand (6 MFMAs, round 3) instead of two fp16 parts (3 MFMAs, round 4)
 *   32768    WITHOUT the fused tail: mask GEMM -> masked tensor -> decoder frame GEMM -> overlap-add as separate launches
 *   1<<17    pyramid pass 1 on a grid of co-resident wavefronts, several rows each (rounds 2-5) -- default since round 6: one row per wavefront
 *   1<<16    srf_backward WITHOUT the fused head of the blocks' pyramid backward (round 6: level 0 + proj_1x1's norm as two passes
 *            over {G_0, y1}): the level-0 conv kernel + the norm's apply pass of rounds 3-5
 *   1<<18    weight-gradient GEMM WITHOUT the wide tile (round 6: 256 x 128 / 128 x 256, one block per CU): the 128 x 128 kernel;
 *            small-channel form on a fixed 1024 blocks (rounds 3-5) instead of one resident round
 *   1<<19    weight-gradient GEMM, 128 x 128 kernel: the masked form for full shapes too (rounds 3-5)
 *   1<<20    weight-gradient GEMM, wide tile: 800-column time chunks (several per block) instead of one long chunk per block
 *   1<<21    fused conv pair on persistent blocks (2 per CU, several tiles each) whatever the launch size -- default: one tile per block
 *   1<<23    fused conv pair with every counted wait of its DMA pipeline as a full drain (bisection aid, same results)
 *   1<<22    TAC forward / backward on the VALU kernels instead of the MFMA forms (n = 16, G = 16)
 *   1<<24    TAC forward on the generic kernel (no lane-per-time-step form)      1<<26   its lane form with four tiles per block
 *   1<<25    weight-gradient partials folded by one chain per output (rounds 3-4) instead of four groups per output (round 5)
 *   1<<27    64-bit pointer loads in the 128 x 128 GEMM (no buffer loads)
 *   1<<28    training forward on the split-bf16 GEMMs (faster; gradients then differ from the reference by ~3e-3)
 *   1<<29 / 1<<30  chunked depthwise-backward / scalar GlobLN-backward kernels and no backward fusion
 *   1<<31    training forward on the exact-fp32 MFMA kernel instead of the three-part split GEMM (pass INT_MIN) */
#ifdef SRF_DIAGNOSTICS
void srf_set_debug_flags(int flags);
#endif

/* ---- whole-model path ---------------------------------------------------------------------- */
int srf_plan_create(const srf_config* cfg, int batch, int T, srf_plan** out);
void srf_plan_destroy(srf_plan* plan);
size_t srf_plan_workspace_bytes(const srf_plan* plan);
int srf_plan_num_params(const srf_plan* plan);    /* tensors in state_dict() order */
int srf_plan_frames(const srf_plan* plan);        /* L */
int srf_plan_padded_length(const srf_plan* plan); /* T' */
int srf_plan_num_launches(const srf_plan* plan);  /* kernel launches per forward (informational) */

/* params: host array of num_params device pointers in the reference's state_dict() order
 * (SURVEY.md Appendix A).  wav: [batch, in_audio_channels, T].  out: [batch, S*in_audio, T]. */
int srf_forward(const srf_plan* plan, const float* const* params, int num_params,
                const float* wav, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* The whole caller-side inference recipe in ONE forward (README.md:100-114): per-example {mean,
 * unbiased std} of the RAW mixture (written to `stats`, [batch][2] device floats), normalisation folded into the encoder's
 * load, "estimates * std + mean" and -- mixture_consistency != 0, as the README prescribes for the GroupComm models --
 * mixture_consistency.apply against the normalised mixture folded into the decoder's overlap-add.  Single-channel
 * mixtures (in_audio_channels = 1).  Same buffers and rules as srf_forward. */
int srf_separate(const srf_plan* plan, const float* const* params, int num_params, const float* wav, float* out,
                 float* stats, int mixture_consistency, void* workspace, size_t workspace_bytes, void* stream);

/* Copy an intermediate of the LAST srf_forward on this workspace into dst (for parity tests).
 * what: 0 = encoder output [Bt,N,L], 1 = separation-module output [Bt,B,L], 2 = masked [Bt,S*A*N,L].
 * 2 fails (SRF_EINVAL) at shapes whose forward runs the mask GEMM fused with the decoder (launches with at least as many
 * 256 x 128 tiles as the GPU has CUs): the masked tensor then never exists in memory. */
int srf_debug_fetch(const srf_plan* plan, const void* workspace, int what, float* dst, size_t dst_floats,
                    void* stream);

/* ---- per-kernel entry points (unit parity + building blocks) -------------------------------- */

/* out[b,n,l] = sum_{a,k} w[n,a,k] * xpad[b,a,h*l+k-h], h=K/2; samples outside [0,T) are zero, so the
 * reference's right zero-padding is implicit in L.  sums (nullable): [Bt][SRF_STAT_BUCKETS][2] += {sum, sumsq}. */
int srf_encoder(const float* wav, const float* w, float* out, double* sums,
                int Bt, int A, int T, int N, int K, int L, void* stream);

/* sums[g][bucket][0..1] += {sum, sumsq} of x[g, :, :] (x: [groups, channels*length]). */
int srf_gln_stats(const float* x, double* sums, int groups, long per_group, void* stream);
/* y = gamma_c * (x - mu_g) / sqrt(var_g + 1e-8) + beta_c, then optional PReLU. */
int srf_gln_apply(const float* x, float* y, const srf_norm* norm, int groups, int channels, int length,
                  void* stream);
/* y = x + GlobLN(q)  (TAC_norm + residual). */
int srf_gln_apply_add(const float* x, const float* q, float* y, const srf_norm* norm, int groups,
                      int channels, int length, void* stream);

/* 1x1 convolution y[b,m,l] = bias[m] + sum_k w[m,k] * f(x[b,k,l])  (+ residual[b,m,l]),
 * f = in_norm (GlobLN and/or PReLU on load; NULL = identity).
 * epilogue_mask != 0:  y = relu(y) * mul[b, m % mul_channels, l]   (mask_nl_class + "* s.unsqueeze(1)").
 * out_sums (nullable): [Bt][SRF_STAT_BUCKETS][2] += {sum, sumsq} of the stored y. */
int srf_pw_conv(const float* x, const float* w, const float* bias, float* y,
                int Bt, int Cin, int Cout, int L, const srf_norm* in_norm, const float* residual,
                double* out_sums, int epilogue_mask, const float* mul, int mul_channels, void* stream);

/* Pre-packed weights for the split-precision GEMM (kernel mode 0): the fp32 weight [Cout,Cin] is split
 * into bf16 hi/lo and laid out tile-by-tile ONCE (srf_forward does it at the start of every forward for
 * all its 1x1 convolutions, in one launch).  srf_packed_pw_weight_bytes() = 0 when the shape does not
 * qualify (needs Cin % 64 == 0, Cout >= 192); srf_pw_conv_packed() with w_packed = NULL (or a
 * non-qualifying shape / mode) is exactly srf_pw_conv().  packed buffers: 16-B aligned device memory.
 * Round 4: a packed buffer holds TWO layouts of the same bf16 parts, written by the one pack launch -- the image of the
 * one-block-per-CU kernel (srf_pwconv_x3w.hip: what srf_pw_conv_packed runs) and the image of the paired-block kernel
 * (srf_pwconv_x3p.hip: two co-resident blocks per CU, bit-identical outputs, what srf_forward / srf_separate run for their
 * proj_1x1 / res_conv / bottleneck GEMMs so that a caller's second stream can share the CUs); srf_packed_pw_weight_bytes
 * covers both.  Always size the buffer with that function. */
size_t srf_packed_pw_weight_bytes(int Cout, int Cin);
int srf_pack_pw_weights(const float* const* w, void* const* packed, const int* Cout, const int* Cin, int n,
                        void* stream);
int srf_pw_conv_packed(const float* x, const float* w, const void* w_packed, const float* bias, float* y,
                       int Bt, int Cin, int Cout, int L, const srf_norm* in_norm, const float* residual,
                       double* out_sums, int epilogue_mask, const float* mul, int mul_channels, void* stream);

/* Round 5: TWO 1x1 convolutions back to back in ONE launch (csrc/srf_pwconv_x3f.hip) -- a conv with Cmid = 256 output channels
 * and the conv that consumes its output, the 256-channel tensor handed over in registers (it is still written to y: it is the
 * model's residual stream):
 *     y  = W1 f(x) + bias1 (+ residual)     f = in_norm: GlobLN (bottleneck, improved_sudormrf.py:292: no residual),
 *                                           GlobLN + PReLU (res_conv, :218-220: residual required), or NULL: no prologue,
 *                                           residual required (the backward's data-gradient pair W_proj^T g + skip, W_res^T of it)
 *     y2 = W2 y + bias2,  out_sums2 (nullable) += {sum, sumsq} of y2          (proj_1x1 of the next block, :205)
 * Results are BIT-IDENTICAL to srf_pw_conv_packed(x -> y) followed by srf_pw_conv_packed(y -> y2) (statistics: to rounding).
 * w1_packed / w2_packed: buffers of srf_pack_pw_weights for [Cmid, Cin1] / [Cout2, Cmid].  srf_pw_conv_pair_supported: the
 * shapes served (Cmid = 256, Cin1 % 64 == 0, 128 <= Cin1 <= 512, Cout2 % 128 == 0, Cout2 <= 512, L % 4 == 0, at least as
 * many 128-column tiles as CUs) under the default kernel mode; srf_forward uses the pair wherever this says 1. */
int srf_pw_conv_pair_supported(int Bt, int Cin1, int Cmid, int Cout2, int L);
int srf_pw_conv_pair(const float* x, const void* w1_packed, const float* bias1, float* y, const srf_norm* in_norm,
                     const float* residual, const void* w2_packed, const float* bias2, float* y2, double* out_sums2,
                     int Bt, int Cin1, int Cmid, int Cout2, int L, void* stream);

/* The same GEMM in the exact-fp32 class for the training forward.  Round 4 (default): TWO FP16 parts per operand (22 mantissa
 * bits, three MFMAs per product block, the inference kernel's speed; range: |operand| < 65520 -- beyond it, and for NaN / inf operands,
 * the affected outputs are non-finite (round 5: no silent clamp), the remedy is flag 16384).  Debug flag 16384:
 * round 3's form, THREE bf16 parts per operand (h + m + l = 24 mantissa bits) and six MFMAs per product block: results in
 * the exact-fp32 class (what srf_forward_train needs: the two-part kernel's 2^-17 representation error is amplified by the
 * early layers' gradients) at ~1.6 x the two-part kernel's time.  Weights packed by srf_pack3_pw_weights (bytes:
 * srf_packed3_pw_weight_bytes, 0 = shape not taken).  No mask epilogue; residual only together with a GlobLN + PReLU
 * prologue (the res_conv form).  w_packed3 = NULL or a shape / launch size the kernel does not take: exactly srf_pw_conv. */
size_t srf_packed3_pw_weight_bytes(int Cout, int Cin);
int srf_pack3_pw_weights(const float* const* w, void* const* packed, const int* Cout, const int* Cin, int n, void* stream);
/* The library remembers, per (device, address), in which of the two forms a packed3 image was written (so that a launch under
 * the other setting of flag 16384 is refused instead of reading a foreign layout).  A pack replaces every record its extent
 * overlaps; srf_pack3_forget drops the record of a buffer the caller frees or re-uses for other data (ABI 15). */
void srf_pack3_forget(const void* packed);
int srf_pw_conv_packed3(const float* x, const float* w, const void* w_packed3, const float* bias, float* y, int Bt, int Cin,
                        int Cout, int L, const srf_norm* in_norm, const float* residual, double* out_sums, void* stream);
/* The fused pair of the training forward (ABI 14): srf_pw_conv_pair on the two-fp16-part images -- y and y2 BIT-IDENTICAL to
 * srf_pw_conv_packed3(x -> y) followed by srf_pw_conv_packed3(y -> y2, out_sums2).  w1_packed3 / w2_packed3: buffers written by
 * srf_pack3_pw_weights in this process under the default (fp16) form -- it keeps the paired-block layout of the same parts in
 * the buffer's second half.  in_norm required (GlobLN: no residual; GlobLN + PReLU: residual required). */
int srf_pw_conv_pair_packed3_supported(int Bt, int Cin1, int Cmid, int Cout2, int L);
int srf_pw_conv_pair_packed3(const float* x, const void* w1_packed3, const float* bias1, float* y, const srf_norm* in_norm,
                             const float* residual, const void* w2_packed3, const float* bias2, float* y2, double* out_sums2,
                             int Bt, int Cin1, int Cmid, int Cout2, int L, void* stream);

/* Depthwise k=5, padding 2: y[r,j] = bias[c] + sum_k w[c,k] * f(x[r, stride*j+k-2]), r=(b,c), zero
 * outside AFTER f (the reference pads the normalised tensor).  x: [Bt,C,Lin], y: [Bt,C,Lout],
 * Lout = (Lin-1)/stride + 1. */
int srf_dwconv5(const float* x, const float* w, const float* bias, float* y,
                int Bt, int C, int Lin, int stride, const srf_norm* in_norm, double* out_sums,
                void* stream);

/* General Conv1d (round 6, ABI 15): any kernel size / stride / dilation / zero padding / groups, weights [Cout, Cin/groups, K]
 * (nn.Conv1d layout), bias nullable; y: [Bt, Cout, Lout], Lout = (Lin + 2 padding - dilation (K - 1) - 1) / stride + 1.
 * Not on the model's path (that builds kSize 1 and depthwise k = 5 only): it serves the reference's building blocks
 * ConvNormAct / DilatedConvNorm (improved_sudormrf.py:50-73,:138-159) when a user instantiates them with other shapes.
 * out_sums (nullable): [Bt][SRF_STAT_BUCKETS][2] += {sum, sumsq} of y. */
int srf_conv1d(const float* x, const float* w, const float* bias, float* y, int Bt, int Cin, int Cout, int Lin, int K,
               int stride, int padding, int dilation, int groups, double* out_sums, void* stream);

/* Bottom-up nearest-x2 upsample-and-add of D normalised levels:
 * y[b,c,j] = n_0[j] + (n_1[j>>1] + (... + n_{D-1}[j>>(D-1)])),  n_k = GlobLN_k(levels[k]).
 * levels[k]: [Bt,C,L>>k]; norms[k] describes level k (prelu ignored). */
int srf_merge(const float* const* levels, const srf_norm* norms, int D, float* y,
              int Bt, int C, int L, double* out_sums, void* stream);

/* Fused depthwise pyramid of one U-ConvBlock (all D depthwise convs + their GlobLNs + the upsample/add
 * merge) in two passes over y1: y1 [groups,C,L] = proj_1x1 conv output (in_norm = its GlobLN + PReLU,
 * applied on load) -> merged [groups,C,L] (+ out_sums for final_norm).  w/bias/gamma/beta: D pointers
 * each (spp_dw[k].conv.weight/.bias, spp_dw[k].norm.gamma/.beta).  Equivalent to D x srf_dwconv5 +
 * srf_merge but moves 3 C*L instead of 7.75 C*L through HBM.  srf_pyramid_supported() tells whether
 * the shape qualifies (L % (4*2^(D-1)) == 0, L >> (D-1) >= 8, row fits LDS).  merged must NOT alias y1
 * (pass 2 re-reads y1 with halos while other wavefronts write merged). */
int srf_pyramid_supported(int C, int L, int D);
size_t srf_pyramid_scratch_bytes(int groups, int C, int L, int D);
int srf_pyramid(const float* y1, float* merged, const srf_norm* in_norm, const float* const* w,
                const float* const* bias, const float* const* gamma, const float* const* beta, int groups,
                int C, int L, int D, void* scratch, double* out_sums, void* stream);

/* Transposed conv synthesis + crop: out[b,o,t] = sum_{ci,l,k: h*l+k-h=t} v[b,ci,l]*w[ci,o,k], t<T.
 * v: [Bt,Ci,L], w: [Ci,Co,K] (ConvTranspose1d layout), out: [Bt,Co,T].
 * scratch: device buffer of srf_decoder_scratch_floats() floats, 16-byte aligned (the decoder zero-fills a part of it with
 * 16-byte stores; an unaligned scratch is rejected with SRF_EINVAL). */
size_t srf_decoder_scratch_floats(int Bt, int Ci, int Co, int K, int L);
int srf_decoder(const float* v, const float* w, float* out, int Bt, int Ci, int Co, int K, int L, int T,
                float* scratch, void* stream);

/* TAC up to (not including) TAC_norm: q[b,g,:,l] = PReLU(Wo [z_g ; PReLU(Wm mean_g z_g + bm)] + bo),
 * z_g = PReLU(Wi x[b,g,:,l] + bi).  x,q: [Bt,G,n,L].  params: the 9 TAC tensors in state_dict order
 * (TAC_input.0.weight/.bias, TAC_input.1.weight, TAC_mean.0.weight/.bias, TAC_mean.1.weight,
 * TAC_output.0.weight/.bias, TAC_output.1.weight).  out_sums: [Bt*G][SRF_STAT_BUCKETS][2]. */
int srf_tac(const float* x, float* q, const float* const* params, int Bt, int G, int n, int H, int L,
            double* out_sums, void* stream);

/* pr + w * (mix - sum_s pr), uniform weights (w = 1/S).  pr,out: [Bt,S,T], mix: [Bt,1,T]. */
int srf_mixture_consistency(const float* pr, const float* mix, float* out, int Bt, int S, int T,
                            void* stream);

/* The 'magsq' variant (mixture_consistency.py:26-28): w[b,s] = mean_t pr[b,s]^2 / (sum_s mean_t pr[b,s]^2 + 1e-9).
 * work: Bt*S floats of device scratch (receives the per-source energies). */
int srf_mixture_consistency_magsq(const float* pr, const float* mix, float* out, int Bt, int S, int T, float* work,
                                  void* stream);

/* Caller-side pre/post-processing that every user of the reference wraps around model() (README.md:100-114,
 * experiments/simple_whamr_evaluation.py:142-148):
 *   srf_wav_normalize:   out = (wav - mean) / (std + 1e-9) per row, std unbiased (torch.std default);
 *                        stats[row] = {mean, std}.  wav,out: [rows,T].
 *   srf_wav_denormalize: out = est * std + mean; with mix_norm != NULL additionally
 *                        mixture_consistency.apply(out, mix_norm) (uniform), as the README prescribes for the
 *                        GroupComm models.  est,out: [Bt,S,T]; stats: [Bt][2]; mix_norm: [Bt,1,T]. */
int srf_wav_normalize(const float* wav, float* out, float* stats, int rows, int T, void* stream);
int srf_wav_stats(const float* wav, float* stats, int rows, int T, void* stream);   /* the statistics alone */
int srf_wav_denormalize(const float* est, const float* stats, const float* mix_norm, float* out, int Bt, int S,
                        int T, void* stream);

/* ---- training loss: clamp(PITLossWrapper(PairwiseNegSDR("sisdr"), pit_from='pw_mtx'), +-clamp)
 * reference: losses/sisdr.py:426-458 (pairwise SI-SDR), :254-311,:342-387 (PIT), runner clamp
 * experiments/run_improved_sudormrf.py:169-171.  est, tgt, grad_est: [Bt,S,T]; S <= 9 -- the reference's own limit
 * (sisdr.py:275); 1..4 sources on the streaming kernels, 5..9 on the generic ones (S! permutations per example).
 *   work  : srf_pit_sisdr_work_bytes(Bt,S) bytes, 8-byte aligned, written by _forward and read by _backward;
 *   pw    : optional [Bt,S,S] pairwise losses (estimate, target);
 *   loss  : 2 floats {clamp(batch mean), raw batch mean}; clamp <= 0 disables the clamp;
 *   srf_pit_sisdr_match: [Bt,S] int32, the estimate matched with target j (best permutation);
 *   _backward: grad_est = upstream[0] * d loss[0] / d est (zero when the raw mean is outside +-clamp);
 *              upstream is a DEVICE scalar (NULL = 1), so an autograd chain never synchronises. */
size_t srf_pit_sisdr_work_bytes(int Bt, int S);
int srf_pit_sisdr_forward(const float* est, const float* tgt, int Bt, int S, int T, float clamp, void* work,
                          float* pw, float* loss, void* stream);
/* The other PairwiseNegSDR configurations (losses/sisdr.py:418-424,440-458): sdr_type 0 = "sisdr", 1 = "sdsdr",
 * 2 = "snr"; zero_mean / take_log as the constructor flags.  srf_pit_sisdr_forward == (0, 1, 1).  Same work buffer,
 * srf_pit_sisdr_match / _backward apply unchanged. */
int srf_pit_sdr_forward(const float* est, const float* tgt, int Bt, int S, int T, float clamp, int sdr_type,
                        int zero_mean, int take_log, void* work, float* pw, float* loss, void* stream);
int srf_pit_sisdr_match(const void* work, int Bt, int S, int* match_out, void* stream);
int srf_pit_sisdr_backward(const float* est, const float* tgt, int Bt, int S, int T, float clamp, const void* work,
                           const float* loss, const float* upstream, float* grad_est, void* stream);

/* ---- validation metric of the runners: PermInvariantSISDR.forward (losses/sisdr.py:66-196; constructed at
 * experiments/run_improved_sudormrf.py:82-85, called :201-205).  pr, tgt: [Bt,S,T], mix: [Bt,1,T] or NULL; S <= 9.
 *   best      [Bt]    max over permutations (itertools order) of the source-mean SI-SNR in dB, eps as the class
 *                     places it: s = <p,t>/(<t,t>+eps) t, 10 log10(<s,s>/(<p-s,p-s>+eps));
 *   best_perm [Bt]    index of that permutation in itertools.permutations(range(S)) (first maximum);
 *   base      [Bt*S]  optional (needs mix): SI-SNR of the mixture against every target -- the class subtracts
 *                     mean(base) over batch AND sources from `best` when improvement=True;
 *   zero_mean         subtract the time mean of every signal first (perform_zero_mean);
 *   work              srf_perm_inv_sisdr_work_bytes(Bt,S) bytes, 8-byte aligned. */
size_t srf_perm_inv_sisdr_work_bytes(int Bt, int S);
int srf_perm_inv_sisdr(const float* pr, const float* tgt, const float* mix, int Bt, int S, int T, int zero_mean,
                       double eps, void* work, float* best, int* best_perm, float* base, void* stream);

/* ---- training step, backward kernels (one entry point per kernel for unit parity) ---- */

/* Weight / bias gradient of a pointwise conv y = W f(x) + bias (improved_sudormrf.py:174,196,256-259,268-269):
 *   dw[m,n] = sum_{b,l} g[b,m,l] * f(x[b,n,l]),  dbias[m] = sum_{b,l} g[b,m,l]
 * f = the forward's operand prologue (in_norm: GlobLN statistics of x + gamma/beta and/or PReLU slope; NULL = none).
 * g: [Bt,Cout,L], x: [Bt,Cin,L], dw: [Cout,Cin], dbias: [Cout] or NULL; accumulate != 0 adds to dw / dbias.
 * scratch: srf_pw_wgrad_scratch_bytes(...) bytes.  L % 4 == 0.
 * srf_pw_wgrad_cols: dw is [Cout, dw_cols] and only the first dw_cols <= Cin columns are produced (x rows beyond
 * them are padding, e.g. the decoder's 42 frame rows padded to 64 for the GEMM). */
size_t srf_pw_wgrad_scratch_bytes(int Bt, int Cout, int Cin, int L);
int srf_pw_wgrad(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout, int L, float* dw,
                 float* dbias, int accumulate, void* scratch, void* stream);
int srf_pw_wgrad_cols(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout, int L,
                      float* dw, int dw_cols, float* dbias, int accumulate, void* scratch, void* stream);
/* ... and with a row pitch of dw_ld floats (a column block of a wider matrix, e.g. the two halves of TAC_output) */
int srf_pw_wgrad_ld(const float* g, const float* x, const srf_norm* in_norm, int Bt, int Cin, int Cout, int L,
                    float* dw, int dw_cols, int dw_ld, float* dbias, int accumulate, void* scratch, void* stream);

/* TAC backward (groupcomm_sudormrf_v2.py:356-377 under autograd).  x, go, gx: [Bt,G,n,L]; go = gradient w.r.t. the
 * TAC MLP output (before TAC_norm); gx = gradient through the MLP only; params / grads: the 9 TAC tensors as in
 * srf_tac / same shapes, gradients ACCUMULATED into.  n in {2,4,8,16}, G in {2,4,8,16}, L % 4 == 0. */
size_t srf_tac_bwd_scratch_bytes(int Bt, int G, int n, int L);
int srf_tac_bwd(const float* x, const float* go, const float* const* params, float* const* grads, int Bt, int G, int n,
                int H, int L, float* gx, void* scratch, void* stream);

/* GlobLN (+PReLU when norm->prelu) backward (improved_sudormrf.py:30-47, PReLU of ConvNormAct :73 / NormAct :113).
 * gout (+ optional gout2, added on load): gradient w.r.t. the normalised (activated) tensor; x: the GlobLN input;
 * norm: {sums of x, gamma, beta, slope}.  gx (accumulate_gx != 0: added to).  dgamma, dbeta [C], dslope [1] are
 * ACCUMULATED into (NULL = skip).  scratch: srf_gln_bwd_scratch_bytes(groups, C). */
size_t srf_gln_bwd_scratch_bytes(int groups, int C);
int srf_gln_bwd(const float* gout, const float* gout2, const float* x, const srf_norm* norm, int groups, int C, int L,
                float* gx, int accumulate_gx, float* dgamma, float* dbeta, float* dslope, void* scratch, void* stream);

/* Merge backward (improved_sudormrf.py:214-216): g_levels[k][j] = sum of g_merged over the 2^k samples level k was
 * upsampled to, k = 1..D-1 ([rows, L >> k]); level 0's gradient IS g_merged (g_levels[0] is ignored). */
int srf_merge_bwd(const float* g_merged, float* const* g_levels, int D, long rows, int L, void* stream);

/* Depthwise k=5 conv backward (improved_sudormrf.py:138-159,178-189).  gd: [groups,C,Lout] gradient w.r.t. the conv
 * output; xin: [groups,C,Lin] the conv's PRE-prologue input, in_norm its prologue (NULL = identity); gin: gradient
 * w.r.t. the prologue's output (overwritten; NULL = skip); dw [C,5], dbias [C]: ACCUMULATED into (NULL = skip). */
size_t srf_dwconv5_bwd_scratch_bytes(int groups, int C);
int srf_dwconv5_bwd(const float* gd, const float* xin, const srf_norm* in_norm, const float* w, int groups, int C,
                    int Lin, int stride, float* gin, float* dw, float* dbias, void* scratch, void* stream);

/* Mask application v = relu(m) * enc (improved_sudormrf.py:296-298; the inference path fuses it into the mask GEMM)
 * and its backward: gm = gv * enc * [m > 0] (may alias gv), genc (+)= sum_s gv * relu(m).
 * m, v, gv, gm: [Bt, SA*N, L]; enc, genc: [Bt, N, L]. */
int srf_mask_apply(const float* m, const float* enc, float* v, int Bt, int SA, int N, int L, void* stream);
int srf_mask_bwd(const float* gv, const float* m, const float* enc, float* gm, float* genc, int accumulate_genc, int Bt,
                 int SA, int N, int L, void* stream);

/* Stand-alone PReLU backward (mask_net.0, improved_sudormrf.py:268): gx = gout * (x >= 0 ? 1 : a) (may alias gout),
 * dslope[0] += sum gout * x [x < 0] (NULL = skip). */
int srf_prelu_bwd(const float* gout, const float* x, const float* slope, float* gx, float* dslope, long n, void* stream);

/* out[b, r*K + k, l] = src[b, r, hop*l + k - pad] (0 outside [0,T); rows R*K..rows_out-1 are zero).  src: [Bt,R,T],
 * out: [Bt,rows_out,L].  Feeds the encoder's weight gradient and the decoder's backward (:247-251, :272-279). */
int srf_frames_gather(const float* src, float* out, int Bt, int R, int T, int K, int hop, int pad, int L, int rows_out,
                      void* stream);

/* ---- training step (both models; SURVEY.md §8b proposal: srf_forward_train / srf_backward) ----
 * srf_forward_train: the forward of srf_forward, un-fused where the backward needs an intermediate, keeping what
 *   the backward needs in `saved` (srf_train_saved_bytes: GlobLN statistics, encoder output, residual stream,
 *   per block y1 / D levels / merged, mask pre-activation, masked encoding).
 * srf_backward: grad_out [Bt, S, T] -> parameter gradients ACCUMULATED into grads[i] (same order and shapes as
 *   params; the caller zeroes them like optimizer.zero_grad()).  Reference: torch autograd over
 *   SuDORMRF.forward (improved_sudormrf.py:283-301), run_improved_sudormrf.py:167-172.
 * saved / scratch: 256-byte aligned device buffers of srf_train_saved_bytes / srf_train_scratch_bytes; `saved`
 *   must stay untouched between the two calls, `scratch` may be reused by anything in between. */
size_t srf_train_saved_bytes(const srf_plan* plan);
size_t srf_train_scratch_bytes(const srf_plan* plan);
int srf_forward_train(const srf_plan* plan, const float* const* params, int num_params, const float* wav, float* out,
                      void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);
int srf_backward(const srf_plan* plan, const float* const* params, float* const* grads, int num_params,
                 const float* wav, const float* grad_out, const void* saved, size_t saved_bytes, void* scratch,
                 size_t scratch_bytes, void* stream);
/* srf_backward plus the gradient w.r.t. the INPUT waveform (ABI 15): grad_wav [Bt, in_audio_channels, T], overwritten --
 * what torch autograd over the reference's forward returns for a mixture that requires grad (improved_sudormrf.py:283-301;
 * the encoder's Conv1d :247-251 transposed, applied to the encoder-output gradient the backward forms anyway). */
int srf_backward_wav(const srf_plan* plan, const float* const* params, float* const* grads, int num_params,
                     const float* wav, const float* grad_out, const void* saved, size_t saved_bytes, void* scratch,
                     size_t scratch_bytes, float* grad_wav, void* stream);

/* On-GPU online remix augmentation of the training loop (experiments/run_improved_sudormrf.py:150-164): new source j
 * of example b = clean[src_b[j][b], src_s[j]] re-scaled to the energy of clean[b, j]; mix = normalize(sum_j),
 * out[:, j] = normalize(new source j), normalize = (x - mean)/(std + eps) with the unbiased std (:127-131).
 * clean, out: [B,S,T] (out must not alias clean), mix: [B,T], src_b: [S][B] and src_s: [S] int32 on the device,
 * S <= 4; scratch: srf_online_remix_scratch_bytes(B, S). */
size_t srf_online_remix_scratch_bytes(int B, int S);
int srf_online_remix(const float* clean, const int* src_b, const int* src_s, int B, int S, int T, float eps, float* mix,
                     float* out, void* scratch, void* stream);

/* Fused clip_grad_norm_ + Adam step over all parameters (run_improved_sudormrf.py:172-176; torch.optim.Adam without
 * amsgrad / weight decay).  tensors: device array of {float* p; const float* g; float* m; float* v; long n;};
 * chunks: device array of {int tensor, int chunk} covering every tensor in srf_opt_chunk_size()-element pieces;
 * buckets: SRF_STAT_BUCKETS doubles of device scratch; step: 1-based step count; max_norm <= 0: no clipping;
 * norm_out: optional device float receiving the total gradient norm before clipping. */
int srf_opt_chunk_size(void);
int srf_clip_adam_step(const void* tensors, const void* chunks, int n_chunks, double* buckets, float max_norm, float lr,
                       float beta1, float beta2, float eps, int step, float* norm_out, void* stream);

/* Input feeder.  Replaces, for this path, the reference's Dataset.__getitem__ + torch DataLoader
 * (dataset_loader/wham.py:171-226): a pool of host threads reads the WAV files of a batch in parallel -- mixture = stream
 * 0, then the sources; one crop start per example, shared by its files (:183-186,:201); float32 values exactly as
 * scipy.io.wavfile.read + torch.tensor(dtype=float32) give them; zero pad to time_samples (:157-166) -- straight into
 * CALLER-OWNED buffers (pinned host memory if the copy to the device is to be asynchronous):
 *   wave [batch][n_streams][time_samples] float32, len [batch][n_streams] int32 (valid samples of every stream: a source
 *   file may be shorter than its mixture), stat [batch][2] float32 = {mean, unbiased std} of the mixture over the range the
 *   reference normalises it on (the crop when it crops, else the whole file: it truncates after normalising, :183-191;
 *   only computed when the feeder was created with normalize != 0).
 * paths: n_items * n_streams file names, item-major.  augment: random crop start when a file is longer than time_samples
 * (splitmix64 of seed, epoch and item: reproducible, unlike the reference's time-seeded numpy generator).  shuffle /
 * drop_last: the DataLoader's (get_generator, :219-224).
 * srf_feeder_submit queues the next batch of the epoch (returns 1, queues nothing, when the epoch is exhausted);
 * srf_feeder_wait blocks until the OLDEST submitted batch is complete and hands its buffers back.  Any number of batches may
 * be in flight.  srf_wav_info / srf_wav_read: the reader on its own (RIFF/WAVE mono, PCM 8/16/24/32, IEEE float 32/64).
 * srf_feeder_create_sharded: the rank-aware form for one process per GPU (the reference feeds its DataParallel replicas
 * from ONE DataLoader and scatters each batch, wham.py:219-226 + run_improved_sudormrf.py:118): every rank builds the same
 * epoch order from (seed, epoch); a GLOBAL batch is batch * world consecutive items of it and this feeder delivers items
 * [rank * batch, (rank + 1) * batch) of every global batch -- disjoint across ranks, their concatenation in rank order is the
 * single-process batch of size batch * world, an epoch covers every item once over all ranks.  world > 1 requires drop_last.
 * normalize = 0 skips the mixture statistics (and the second read of files longer than time_samples they need).
 * srf_feeder_create = the same with normalize 1, rank 0, world 1.  srf_feeder_epoch_items: the items this rank delivers in
 * the current epoch, in order.  srf_feeder_read_example: one example synchronously in the calling thread (Dataset[i]). */
typedef struct srf_feeder srf_feeder;
int srf_wav_info(const char* path, int* rate, int* channels, int* bits, long* frames);
int srf_wav_read(const char* path, long start, long n, float* dst, long* frames);
int srf_feeder_create(const char* const* paths, int n_items, int n_streams, int time_samples, int batch, int n_threads,
                      int augment, int shuffle, int drop_last, unsigned long long seed, srf_feeder** out);
int srf_feeder_create_sharded(const char* const* paths, int n_items, int n_streams, int time_samples, int batch,
                              int n_threads, int augment, int shuffle, int drop_last, unsigned long long seed, int normalize,
                              int rank, int world, srf_feeder** out);
void srf_feeder_destroy(srf_feeder* f);
long srf_feeder_epoch_items(srf_feeder* f, int* items, long capacity);
int srf_feeder_read_example(const char* const* paths, int n_streams, int time_samples, long start, int augment, int normalize,
                            float* wave, int* len, float* stat);
long srf_feeder_batches_per_epoch(const srf_feeder* f);
long srf_feeder_item_frames(const srf_feeder* f, int item);
int srf_feeder_start_epoch(srf_feeder* f, int epoch);
int srf_feeder_submit(srf_feeder* f, float* wave, int* len, float* stat);
int srf_feeder_wait(srf_feeder* f, float** wave, int** len, float** stat, int* n_valid);
/* The Dataset's normalisation on a whole batch, on the device (wham.py:189-217): raw [B][n_streams][T] as delivered by the
 * feeder (len [B][n_streams]) -> mix [B][T], src [B][n_streams-1][T].  normalize = 0: copy; 1: every stream
 * (x - mean)/(std + eps) over ITS OWN valid samples (the mixture with `stat`), zero pad, then (x - mean_T)/(mix_std + eps)
 * with the population std of the padded mixture. */
int srf_feeder_normalize(const float* raw, const int* len, const float* stat, int B, int n_streams, int T, int normalize,
                         float eps, float* mix, float* src, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUDORMRF_HIP_H */
