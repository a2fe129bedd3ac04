"""Algorithmic (fusion-minimal) HBM bytes and FLOPs of the SuDoRM-RF forward, per example and per
kernel launch (derivation in DESIGN.md).  Pure arithmetic, used by bench.py."""

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak (exact fp32)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # v_mfma_f32_32x32x16_bf16 dense peak (no sparsity)


def frames(T, K, D):
    h = K // 2
    n = h * 2 ** D
    Tp = n if T < n else (T // n + (1 if T % n else 0)) * n
    return (Tp + 2 * h - K) // h + 1


def bytes_per_example(variant, B, C, U, D, K, N, S, T, G=1):
    """materialise only at GlobLN barriers, fp32, weights ignored."""
    L = frames(T, K, D)
    xfac = 7 if variant == "groupcomm" else 3
    return 4.0 * ((1 + S) * T + 3 * N * L + 2 * B * L + U * (xfac * B * L + (10 - 2.0 ** (3 - D)) * C * L))


def flops_per_example(variant, B, C, U, D, K, N, S, T, G=1):
    L = frames(T, K, D)
    dw = 5 * C * L * (2 - 2.0 ** (1 - D))
    if variant == "groupcomm":
        n, h, c = B // G, 3 * B // G, C // G
        blk = L * (G * 3 * n * h + h * h) + G * (2 * n * c * L) + dw
    else:
        blk = 2 * B * C * L + dw
    mac = N * K * L + N * B * L + U * blk + B * S * N * L + S * N * S * K * L
    return 2.0 * mac


def pyramid_fused(C, L, D):
    """Mirror of srf_pyramid_supported() (csrc/srf_pyramid.hip)."""
    if D < 1 or D > 8 or C > 2048:
        return False
    if (L >> (D - 1)) < 8 or L % (1 << (D - 1)):
        return False
    if (D <= 5 and L % 16 == 0 and L // 16 >= 4) or (D == 6 and L % 32 == 0 and L // 32 >= 4):
        return True
    if L % (4 << (D - 1)):
        return False
    size_a = max(L + 8, sum((L >> k) + 8 for k in range(1, D)))
    return 4 * (L + 8 + ((size_a + 3) & ~3)) + 8 * 72 <= 160 * 1024 - 1024


def pyramid_regs(L, D):
    """Mirror of srf_pyramid_reg_supported(): register-resident pyramid kernels (no stats pre-kernel)."""
    return (D <= 5 and L % 16 == 0 and L // 16 >= 4) or (D == 6 and L % 32 == 0 and L // 32 >= 4)


def pyramid_tiled(L, D):
    """True when srf_pyramid() runs its stats_finalize pre-kernel: register-resident kernels
    (srf_pyramid_reg_supported) or the LDS-tiled ones (pyr_pick_tile)."""
    if (D <= 5 and L % 16 == 0 and L // 16 >= 4) or (D == 6 and L % 32 == 0 and L // 32 >= 4):
        return True
    unit = 4 << (D - 1)
    m = L // unit
    cands = [q * unit for q in range(1, m + 1) if m % q == 0 and 384 <= q * unit <= 1024]
    if not cands:
        return False
    best = min(cands, key=lambda ts: (abs(ts - 640), ts))
    return best != L


def _packable(cin, cout):
    """Mirror of srf_x3w_shape_supported(): 1x1 convs whose weights srf_forward pre-splits (srf_pwconv_x3w.hip)."""
    return cin % 64 == 0 and cin >= 128 and cout >= 192


def fused_tail(B, N, SA, K, L, Bt, kernel_mode=0, packed=True, cus=256):
    """Mirror of srf_mask_decode_supported() (srf_pwconv.hip): whether srf_forward runs its tail as one GEMM launch."""
    return (kernel_mode == 0 and packed and SA * K <= 64 and _packable(B, SA * N) and (SA * N) % 8 == 0 and N % 8 == 0 and
            B % 32 == 0 and L % 4 == 0 and Bt <= 1024 and Bt * B * L * 4 < 2 ** 31 and
            Bt * ((SA * N + 255) // 256) * ((L + 127) // 128) >= cus)


def conv_pair(variant, B, C, D, N, L, Bt, kernel_mode=0, packed=True, cus=256, head=False):
    """Mirror of srf_forward's pair_res / pair_head (srf_api.hip, srf_pw_conv_pair_supported): whether res_conv (head: the
    bottleneck) and the proj_1x1 that follows run as one launch (srf_pwconv_x3f.hip)."""
    k1 = N if head else C
    return (variant == "improved" and kernel_mode == 0 and packed and B == 256 and pyramid_fused(C, L, D) and
            _packable(k1, B) and _packable(B, C) and k1 % 64 == 0 and 128 <= k1 <= 512 and C % 128 == 0 and C <= 512 and
            L % 4 == 0 and Bt * k1 * L * 4 < 2 ** 31 and Bt * ((L + 127) // 128) >= cus)


def launch_model(variant, B, C, U, D, K, N, S, T, Bt, G=1, A=1, kernel_mode=0, packed=True, fuse_tail=True, pairs=True):
    """(family, algorithmic_bytes, flops) for every kernel launch of one srf_forward, in launch order
    (mirrors srf_api.hip).  Bytes = tensors each kernel must read + write once, fp32.  pairs = False: debug flag 1."""
    L = frames(T, K, D)
    SA = S * A
    Bg, nB, nC = Bt * G, B // G, C // G
    f = 4.0
    out = [("zero_fill", 16.0 * 64 * Bg * (1 + U * (D + 2 + (1 if variant == "groupcomm" else 0))), 0.0)]   # GlobLN statistic slots
    convs = [(N, B)] + [(nB, nC), (nC, nB)] * U + [(B, SA * N)]
    pk = [(ci, co) for ci, co in convs if _packable(ci, co)]
    if kernel_mode == 0 and packed and pk:      # one launch per forward: fp32 weights -> bf16 hi|lo tile images (two layouts:
        out.append(("pack_pw_weights", sum(12.0 * ci * co for ci, co in pk), 0.0))      # the one-block and the paired-block GEMM's)
    out.append(("encoder", f * Bt * (A * T + N * L), 2.0 * Bt * N * A * K * L))

    def pw(cin, cout, bt, extra_in=0):
        return ("pw_conv", f * bt * L * (cin + cout + extra_in), 2.0 * bt * cin * cout * L)

    def pair(cin1, cout2, bt, residual):      # y = W1 f(x) + b1 (+ residual) [256 rows, written: the residual stream]; y2 = W2 y + b2
        return ("pw_pair", f * bt * L * (cin1 + (2 if residual else 1) * nB + cout2), 2.0 * bt * L * nB * (cin1 + cout2))

    pair_res = pairs and conv_pair(variant, B, C, D, N, L, Bt, kernel_mode, packed)
    pair_head = pair_res and conv_pair(variant, B, C, D, N, L, Bt, kernel_mode, packed, head=True)
    out.append(pair(N, nC, Bt, False) if pair_head else pw(N, B, Bt))
    y1_ready = pair_head
    preadd = False
    for blk in range(U):
        if variant == "groupcomm":
            n, h = nB, 3 * nB
            out.append(("tac", f * Bt * B * L * 2, 2.0 * Bt * L * (2 * G * n * h + h * h + n * h + G * n * h)))
            # u = x + GlobLN(q): folded into the proj conv's load on the thin-shape kernel (srf_pw_conv_preadd_supported)
            preadd = kernel_mode != 1 and nB in (8, 16, 32) and nC in (8, 16, 32, 64) and L % 4 == 0
            if preadd:
                out.append(("pw_conv", f * Bg * L * (3 * nB + nC), 2.0 * Bg * nB * nC * L + 3.0 * Bt * B * L))
            else:
                out.append(("gln_apply_add", f * Bt * B * L * 3, 3.0 * Bt * B * L))
        if not (variant == "groupcomm" and preadd) and not y1_ready:
            out.append(pw(nB, nC, Bg))
        y1_ready = False
        dw_flops = 2.0 * 5 * Bg * nC * sum(L >> k for k in range(D))
        if kernel_mode != 1 and pyramid_fused(nC, L, D):
            if pyramid_tiled(L, D) and not pyramid_regs(L, D):
                out.append(("stats_finalize", 8.0 * Bg * 130, 0.0))
            out.append(("pyramid_moments", f * Bg * nC * L, dw_flops))
            out.append(("pyramid_finalize", 8.0 * Bg * nC * D * 5, 0.0))
            out.append(("pyramid_merge", f * Bg * nC * L * 2, dw_flops + 2.0 * D * Bg * nC * L))
        else:
            for k in range(D):
                lin = L if k == 0 else L >> (k - 1)
                lout = L >> k
                out.append(("dwconv5", f * Bg * nC * (lin + lout), 2.0 * 5 * Bg * nC * lout))
            out.append(("merge", f * Bg * nC * (sum(L >> k for k in range(D)) + L), 2.0 * D * Bg * nC * L))
        if pair_res and blk + 1 < U:
            out.append(pair(nC, nC, Bt, True))
            y1_ready = True
        else:
            out.append(pw(nC, nB, Bg, extra_in=nB))
    if fuse_tail and fused_tail(B, N, SA, K, L, Bt, kernel_mode, packed):
        # K5: mask GEMM + decoder contraction in one launch (srf_pwconv_x3w.hip EPI 4): the masked tensor is never stored;
        # per-256-channel partial frames [Bt, nparts, SA K, L] instead, summed by the overlap-add
        M, nparts = SA * K, (SA * N + 255) // 256
        out.append(("pack_decoder", f * SA * N * M + 8192.0 * 8 * nparts, 0.0))
        out.append(("pw_mask_decode", f * Bt * L * (B + N + nparts * M), 2.0 * Bt * L * SA * N * (B + M)))
        out.append(("overlap_add", f * Bt * (nparts * M * L + SA * T), 3.0 * nparts * Bt * SA * T))
        return out
    out.append(pw(B, SA * N, Bt, extra_in=N))
    out.append(("transpose", f * 2 * SA * N * SA * K, 0.0))
    out.append(("zero_fill", f * 64 * ((SA * K + 63) // 64), 0.0))   # the frame GEMM's zero bias
    out.append(pw(SA * N, SA * K, Bt))
    out.append(("overlap_add", f * Bt * (SA * K * L + SA * T), 3.0 * Bt * SA * T))
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# Training step (forward with saved activations, PIT-SI-SDR, backward, clip, Adam): run_improved_sudormrf.py:146-177
# ---------------------------------------------------------------------------------------------------------------------------
def n_params(variant, B, C, U, D, K, N, S, G=1, A=1):
    """Parameter count of the model (Appendix A of SURVEY.md; checked against the README's #Params there)."""
    SA = S * A
    if variant == "groupcomm":
        n, h, c = B // G, 3 * B // G, C // G
        tac = (h * n + h + 1) + (h * h + h + 1) + (n * 2 * h + n + 1) + 2 * n
        ub = (c * n + c) + 2 * c + 1 + D * (5 * c + c + 2 * c) + 2 * c + 1 + (n * c + n)
        blk = tac + ub
    else:
        blk = (C * B + C) + 2 * C + 1 + D * (5 * C + C + 2 * C) + 2 * C + 1 + (B * C + B)
    return N * A * K + 2 * N + (B * N + B) + U * blk + 1 + (SA * N * B + SA * N) + SA * N * SA * K


def train_bytes_per_example(variant, B, C, U, D, K, N, S, T, G=1):
    """Fusion-minimal HBM bytes of one TRAINING step per example, on the rules of SURVEY.md 8(d) extended to the backward
    (DESIGN.md 6c): every tensor the forward materialises at a GlobLN barrier is read once more by the backward (the local
    derivative needs the activation), and every forward pass over a tensor has a mirror pass over its gradient.  So
        step = 2 x forward + (one extra read of each materialised activation) + what the training forward must keep that the
               inference forward fuses away (the mask pre-activation [S N, L]: written, read by its ReLU' and by the enc gradient),
    weights / optimizer state ignored as in 8(d) (<= 0.4 GB per step against >= 80 GB of activations at cfg 4)."""
    L = frames(T, K, D)
    fwd = bytes_per_example(variant, B, C, U, D, K, N, S, T, G)
    xre = 2 if variant == "groupcomm" else 1            # block input (GroupComm: x and u)
    reread = 4.0 * (N * L + U * (xre * B * L + (4 - 2.0 ** (1 - D)) * C * L) + B * L)
    mask = 4.0 * (3 * S * N * L)
    return 2.0 * fwd + reread + mask


def train_flops_per_example(variant, B, C, U, D, K, N, S, T, G=1):
    """Forward + data gradient + weight gradient of every contraction (~3 x the forward's FLOPs)."""
    return 3.0 * flops_per_example(variant, B, C, U, D, K, N, S, T, G)


def _train_family_model_groupcomm(B, C, U, D, K, N, S, T, Bt, G, A, fused_head=True):
    """GroupComm step (groupcomm_sudormrf_v2.py:232-339; csrc/srf_train.hip with gc = true): per block a TAC (MLP over the G groups
    of n = B / G channels, hidden H = 3 n) and a U-ConvBlock on the (batch x group)-folded tensor [Bt G, B / G -> C / G, L].  Folding
    does not change a family's bytes (Bt G rows of C / G channels = Bt rows of C), so the pyramid / norm / depthwise entries are
    the Improved model's; the per-group 1x1 convs and their gradients run on the thin-shape kernels."""
    L = frames(T, K, D)
    f = 4.0 * Bt
    n = B // G
    H = 3 * n
    GH = G * H                                   # rows of the TAC's hidden tensors per example
    SN, SK = S * A * N, S * A * K
    P = n_params("groupcomm", B, C, U, D, K, N, S, G, A)
    lv = [L >> k for k in range(D)]
    lev_sum = sum(lv)
    fam = {}

    def add(name, nbytes, flops=0.0):
        b0, f0 = fam.get(name, (0.0, 0.0))
        fam[name] = (b0 + nbytes, f0 + flops)

    tac_flops = 2.0 * Bt * L * (2 * G * n * H + H * H + n * H + G * n * H)
    # ---- forward (training)
    add("encoder", f * (A * T + N * L), 2.0 * Bt * N * A * K * L)
    add("pw_conv_x3w4<1>", f * L * (N + B), 2.0 * Bt * N * B * L)
    add("tac_mfma", U * f * L * 2 * B, U * tac_flops)
    add("pw_conv_small", U * f * L * (3 * B + C), U * 2.0 * Bt * B * C * L / G)        # proj: x, q in; u, y out
    dw = 2.0 * 5 * Bt * C * lev_sum
    add("pyramid_moments", U * f * C * L, U * dw)
    add("pyramid_finalize", U * 8.0 * Bt * C * D * 5)
    add("pyramid_merge_save", U * f * C * (2 * L + lev_sum - (L if fused_head and D > 1 else 0)), U * (dw + 2.0 * D * Bt * C * L))
    add("pw_conv_small", U * f * L * (C + 2 * B), U * 2.0 * Bt * B * C * L / G)        # res_conv + residual
    add("pw_conv_x3w4<3>", f * L * (B + SN), 2.0 * Bt * B * SN * L)
    add("mask_apply", f * L * (2 * SN + N), 2.0 * Bt * SN * L)
    add("pw_conv_bf16x3_w8", f * L * (SN + SK), 2.0 * Bt * SN * SK * L)
    add("overlap_add", f * (SK * L + S * A * T), 3.0 * Bt * S * A * T)
    add("pit_sisdr_stats", f * 2 * S * T, 2.0 * Bt * (4 * S + S * S) * T)
    add("pit_sisdr_grad", f * 3 * S * T, 4.0 * Bt * S * T)
    # ---- backward: tail (as the Improved model)
    add("frames_gather", f * (S * A * T + SK * L) + f * (A * T + A * K * L))
    add("pw_wgrad", f * L * (SN + SK), 2.0 * Bt * SN * SK * L)
    add("pw_conv_bf16x3_p8<0>", f * L * (SK + SN), 2.0 * Bt * SN * SK * L)
    add("mask_bwd", f * L * (3 * SN + 2 * N), 3.0 * Bt * SN * L)
    add("pw_wgrad", f * L * (SN + B), 2.0 * Bt * SN * B * L)
    add("pw_conv_x3w<0>", f * L * (SN + B), 2.0 * Bt * SN * B * L)
    add("prelu_bwd", f * 3 * B * L, 2.0 * Bt * B * L)
    # ---- backward: U blocks
    add("pw_wgrad_small", U * 2 * f * L * (B + C), U * 2 * 2.0 * Bt * B * C * L / G)   # res_conv / proj weight gradients
    add("pw_conv_small", U * (f * L * (B + C) + f * L * (C + 2 * B)), U * 2 * 2.0 * Bt * B * C * L / G)   # their data gradients
    add("gln_bwd_reduce", U * f * C * 2 * (L + lv[-1]), U * 4.0 * Bt * C * (L + lv[-1]))
    _block_head_families(add, U, f, C, L, lv, lev_sum, D, Bt, fused_head)
    # TAC backward: x, g_o in (g_po re-read); g_x, Z, GPZ, GPO out (+ the [Bt, H | n, L] tensors of the group mean path)
    add("tac_bwd_mfma", U * f * L * (5 * B + 2 * GH + 4 * H + n), U * 3.0 * tac_flops)
    add("gln_bwd_reduce", U * f * B * 2 * L, U * 4.0 * Bt * B * L)                     # TAC_norm
    add("gln_bwd_apply", U * f * B * 3 * L, U * 8.0 * Bt * B * L)
    add("accumulate", U * f * B * 3 * L)
    # the TAC's four weight gradients: g_pz x^T, g_po z^T over (batch, group, time); (sum_g g_po) q^T, g_pq zbar^T over (batch, time)
    add("pw_wgrad_small", U * (2 * f * L * (GH + B) + f * L * (n + H) + f * L * 2 * H),
        U * 2.0 * Bt * L * (2 * G * n * H + n * H + H * H))
    # ---- backward: head
    add("pw_wgrad", f * L * (B + N), 2.0 * Bt * B * N * L)
    add("pw_conv_x3w<0>", f * L * (B + N), 2.0 * Bt * B * N * L)
    add("gln_bwd_reduce", f * N * 2 * L, 4.0 * Bt * N * L)
    add("gln_bwd_apply", f * N * 4 * L, 8.0 * Bt * N * L)
    add("pw_wgrad", f * L * (N + A * K), 2.0 * Bt * N * A * K * L)
    add("grad_sqnorm", 4.0 * P)
    add("clip_adam", 28.0 * P, 12.0 * P)
    return fam


def _block_head_families(add, U, f, C, L, lv, lev_sum, D, Bt, fused_head):
    """The norm-apply and depthwise-backward families of the U blocks.  Per level k >= 1 the conv backward reads g_out, d_k (its
    own norm's apply on load), the conv input, the merge part and writes g_in.  Level 0 and proj_1x1's norm: round 6's fused head
    (srf_backward.hip, srf_bwd_l0p_kernel) = a reduce pass and an apply pass over {G_0, y1} (2 + 3 C L; d_0 and g_o are
    re-computed); without it (D = 1, odd lengths, debug flag 1 << 16) the level-0 conv kernel (4 C L) + the norm's apply pass (3)."""
    head = fused_head and D > 1
    deep = sum(2 * lv[k] + 3 * lv[k - 1] for k in range(2 if head else 1, D))
    add("gln_bwd_apply", U * f * C * (3 * L + (lev_sum - L)) + (0 if head else U * f * C * 3 * L), U * (8.0 if head else 16.0) * Bt * C * L)
    if deep or not head:
        add("dwconv5_bwd", U * f * C * (deep + (0 if head else 4 * L)),
            U * 2.0 * 15 * Bt * C * (lev_sum - ((L + lv[1]) if head else 0)))
    if head:
        add("bwd_l1h", U * f * C * (2 * lv[1] + 3 * L), U * 2.0 * 22 * Bt * C * L)       # level 1, conv input re-computed from y1
        add("bwd_l0p_reduce", U * f * C * 2 * L, U * 2.0 * 24 * Bt * C * L)
        add("bwd_l0p_apply", U * f * C * 3 * L, U * 2.0 * 20 * Bt * C * L)


def train_family_model(variant, B, C, U, D, K, N, S, T, Bt, G=1, A=1, dgrad_pairs=None, fused_head=True):
    """{profiler family: (algorithmic bytes per step, FLOPs per step)} of srf_forward_train + the loss + srf_backward + the
    optimizer for the IMPROVED model (csrc/srf_train.hip is the launch sequence this mirrors; launches per step come from the
    in-library profiler, so per-launch figures = these totals / the launches counted).  Bytes = tensors a kernel family must read
    + write once per launch, fp32, weights ignored.  GroupComm (round 4): the families of its blocks -- TAC forward / backward on
    the matrix pipe, the thin-shape convs and weight gradients, the (batch x group)-folded pyramid and norm backwards -- plus the
    head / tail families it shares with the Improved model."""
    if variant not in ("improved", "groupcomm"):
        return None
    if variant == "groupcomm":
        return _train_family_model_groupcomm(B, C, U, D, K, N, S, T, Bt, G, A, fused_head)
    L = frames(T, K, D)
    # round 5: the B = 256 models' data gradients of proj_1x1(i) and res_conv(i - 1) run as ONE launch (srf_pwconv_x3f.hip without
    # prologue) when a launch has at least one 128-column tile per CU -- U - 1 pairs; the first res_conv gradient and the last
    # proj_1x1 gradient stay launches of their own.  dgrad_pairs=None: decide from the shape as srf_backward does.
    if dgrad_pairs is None:
        dgrad_pairs = B == 256 and C % 128 == 0 and C <= 512 and Bt * ((L + 127) // 128) >= 256
    npair = (U - 1) if dgrad_pairs else 0
    f = 4.0 * Bt
    SN, SK = S * A * N, S * A * K
    P = n_params(variant, B, C, U, D, K, N, S, G, A)
    lv = [L >> k for k in range(D)]
    lev_sum = sum(lv)                                  # C-rows of all pre-norm levels: L (2 - 2^(1-D))
    fam = {}

    def add(name, nbytes, flops=0.0):
        b0, f0 = fam.get(name, (0.0, 0.0))
        fam[name] = (b0 + nbytes, f0 + flops)

    # ---- forward (training): GEMMs on the fp16 two-part split kernel (x3w4), fused pyramid with saved levels, un-fused tail
    add("encoder", f * (A * T + N * L), 2.0 * Bt * N * A * K * L)
    # round 5: with the pairs (same condition as the backward's), the forward's res_conv(i) + proj_1x1(i + 1) are U - 1 launches
    # of the fp16-part pair kernel, bottleneck + proj_1x1(0) one more when N fits the kernel (<= 512): only the last res_conv stays
    head_pair = bool(npair) and N % 64 == 0 and 128 <= N <= 512
    if head_pair:
        add("pw_pair_x3f4<1>", f * L * (N + B + C), 2.0 * Bt * (N * B + B * C) * L)
    else:
        add("pw_conv_x3w4<1>", f * L * (N + B), 2.0 * Bt * N * B * L)
    nproj = U - npair - (1 if head_pair else 0)
    if nproj:
        add("pw_conv_x3w4<0>", nproj * f * L * (B + C), nproj * 2.0 * Bt * B * C * L)
    if npair:
        add("pw_pair_x3f4<2>", npair * f * L * (C + 2 * B + C), npair * 2 * 2.0 * Bt * B * C * L)
    dw = 2.0 * 5 * Bt * C * lev_sum
    add("pyramid_moments", U * f * C * L, U * dw)
    add("pyramid_finalize", U * 8.0 * Bt * C * D * 5)
    add("pyramid_merge_save", U * f * C * (2 * L + lev_sum - (L if fused_head and D > 1 else 0)),
        U * (dw + 2.0 * D * Bt * C * L))     # pass 2 + the levels on the side (level 0 not kept with the fused backward head)
    add("pw_conv_x3w4<2>", (U - npair) * f * L * (C + 2 * B), (U - npair) * 2.0 * Bt * B * C * L)
    add("pw_conv_x3w4<3>", f * L * (B + SN), 2.0 * Bt * B * SN * L)
    add("mask_apply", f * L * (2 * SN + N), 2.0 * Bt * SN * L)
    # decoder (stand-alone form): weight transpose, frame GEMM S N -> S K, overlap-add
    add("pw_conv_bf16x3_w8", f * L * (SN + SK), 2.0 * Bt * SN * SK * L)            # (K = S N, 42 rows: the 128 x 128 split kernel -- round 6: no longer the exact-fp32 MFMA kernel)
    add("overlap_add", f * (SK * L + S * A * T), 3.0 * Bt * S * A * T)
    # ---- loss: one streaming pass over estimates + targets, the gradient pass
    add("pit_sisdr_stats", f * 2 * S * T, 2.0 * Bt * (4 * S + S * S) * T)
    add("pit_sisdr_grad", f * 3 * S * T, 4.0 * Bt * S * T)
    # ---- backward: tail
    add("frames_gather", f * (S * A * T + SK * L) + f * (A * T + A * K * L))
    add("pw_wgrad", f * L * (SN + SK), 2.0 * Bt * SN * SK * L)                       # decoder weight
    add("pw_conv_bf16x3_p8<0>", f * L * (SK + SN), 2.0 * Bt * SN * SK * L)                   # decoder data gradient (frames -> g_v; K = 42 padded to 64: the 128 x 128 split kernel)
    add("mask_bwd", f * L * (3 * SN + 2 * N), 3.0 * Bt * SN * L)
    add("pw_wgrad", f * L * (SN + B), 2.0 * Bt * SN * B * L)                         # mask_net weight
    add("pw_conv_x3w<0>", f * L * (SN + B), 2.0 * Bt * SN * B * L)                   # mask_net data gradient
    add("prelu_bwd", f * 3 * B * L, 2.0 * Bt * B * L)
    # ---- backward: U blocks
    add("pw_wgrad", U * f * L * (B + C), U * 2.0 * Bt * B * C * L)                   # res_conv weight (prologue re-applied on load)
    add("pw_conv_x3w<0>", (U - npair) * f * L * (B + C), (U - npair) * 2.0 * Bt * B * C * L)     # res_conv data gradient (un-paired)
    add("gln_bwd_reduce", U * f * C * 2 * (L + lv[-1]), U * 4.0 * Bt * C * (L + lv[-1]))      # final_norm + the deepest level
    _block_head_families(add, U, f, C, L, lv, lev_sum, D, Bt, fused_head)
    add("pw_wgrad", U * f * L * (B + C), U * 2.0 * Bt * B * C * L)                   # proj_1x1 weight
    add("pw_conv_x3w<0>", (U - npair) * f * L * (C + 2 * B), (U - npair) * 2.0 * Bt * B * C * L)   # proj_1x1 data gradient + skip (un-paired)
    if npair:        # g_y1 and the skip gradient in, g_x(i) out, g_f(i - 1) out: the 256-channel g_x is not re-read
        add("pw_pair_x3f<0>", npair * f * L * (C + 2 * B + C), npair * 2 * 2.0 * Bt * B * C * L)
    # ---- backward: head
    add("pw_wgrad", f * L * (B + N), 2.0 * Bt * B * N * L)
    add("pw_conv_x3w<0>", f * L * (B + N), 2.0 * Bt * B * N * L)                     # bottleneck data gradient
    add("gln_bwd_reduce", f * N * 2 * L, 4.0 * Bt * N * L)
    add("gln_bwd_apply", f * N * 4 * L, 8.0 * Bt * N * L)
    add("pw_wgrad", f * L * (N + A * K), 2.0 * Bt * N * A * K * L)                   # encoder weight
    # ---- optimizer: sum of squares, then clip coefficient + Adam in one pass (p, g, m, v read; p, m, v written)
    add("grad_sqnorm", 4.0 * P)
    add("clip_adam", 28.0 * P, 12.0 * P)
    return fam
