"""Explicit-index numpy restatement of the SuDoRM-RF forward path.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Every op is written from its index formula, not through
a convolution library, so it is independent of ATen's conv semantics.  fp64 by
default (the "truth" the fp32 paths are measured against); pass
``dtype=np.float32`` for an fp32 run.  Each function cites the reference lines
it restates (paths relative to /root/reference/sudo_rm_rf/dnn/).
"""
import numpy as np

from .schema import ModelConfig


# ----------------------------------------------------------------------------
# primitive ops
# ----------------------------------------------------------------------------
def pad_to_appropriate_length(x, cfg: ModelConfig):
    """models/improved_sudormrf.py:303-314 -- zero right-pad the last axis to a
    multiple of (K//2)*2^D (or up to that value when shorter)."""
    T = x.shape[-1]
    Tp = cfg.padded_length(T)
    out = np.zeros(x.shape[:-1] + (Tp,), dtype=x.dtype)
    out[..., :T] = x
    return out


def encoder(x, w):
    """models/improved_sudormrf.py:247-251,286 -- Conv1d(A->N, k=K, stride=K//2,
    padding=K//2, bias=False):  v[b,n,l] = sum_{a,k} w[n,a,k] * xp[b,a,h*l+k],
    xp = x zero-padded by h=K//2 on both sides."""
    Bt, A, T = x.shape
    N, A2, K = w.shape
    assert A == A2
    h = K // 2
    L = (T + 2 * h - K) // h + 1
    xp = np.zeros((Bt, A, T + 2 * h), dtype=x.dtype)
    xp[:, :, h:h + T] = x
    out = np.zeros((Bt, N, L), dtype=x.dtype)
    for k in range(K):
        # samples h*l + k for l = 0..L-1
        seg = xp[:, :, k:k + h * (L - 1) + 1:h]            # [Bt, A, L]
        out += np.einsum("na,bal->bnl", w[:, :, k], seg)
    return out


def glob_ln(x, gamma, beta, eps=1e-8):
    """models/improved_sudormrf.py:30-47 + :24-27 -- per leading index: mean and
    BIASED variance over (channel, time); y = gamma_c*(x-mu)/sqrt(var+eps)+beta_c."""
    mu = x.mean(axis=(1, 2), keepdims=True)
    var = ((x - mu) ** 2).mean(axis=(1, 2), keepdims=True)
    y = (x - mu) / np.sqrt(var + eps)
    return gamma[None, :, None] * y + beta[None, :, None]


def conv1x1(x, w, b=None):
    """nn.Conv1d(kernel_size=1): y[b,m,l] = sum_k w[m,k,0] x[b,k,l] + bias[m]
    (improved_sudormrf.py:256-259 bottleneck, :174 proj_1x1, :196 res_conv,
    :268 mask conv)."""
    y = np.einsum("mk,bkl->bml", w[:, :, 0], x)
    if b is not None:
        y = y + b[None, :, None]
    return y


def prelu(x, a):
    """nn.PReLU() with ONE shared slope (improved_sudormrf.py:68,109,269)."""
    a = np.asarray(a).reshape(())
    return np.where(x >= 0, x, a * x)


def dwconv5(x, w, b, stride):
    """models/improved_sudormrf.py:152-153 with groups=C, kSize=5, d=1, padding=2:
    y[b,c,j] = bias[c] + sum_{k=0..4} w[c,0,k] * x[b,c, stride*j + k - 2] (0 outside)."""
    Bt, C, Lin = x.shape
    Lout = (Lin + 4 - 5) // stride + 1
    xp = np.zeros((Bt, C, Lin + 4), dtype=x.dtype)
    xp[:, :, 2:2 + Lin] = x
    y = np.zeros((Bt, C, Lout), dtype=x.dtype) + b[None, :, None]
    for k in range(5):
        y = y + w[None, :, 0, k, None] * xp[:, :, k:k + stride * (Lout - 1) + 1:stride]
    return y


def upsample_nearest2(x):
    """torch.nn.Upsample(scale_factor=2) default mode='nearest'
    (improved_sudormrf.py:190-194): up[j] = x[j // 2]."""
    return np.repeat(x, 2, axis=-1)


def decoder(v, w):
    """models/improved_sudormrf.py:272-279,300 -- ConvTranspose1d(Ci->Co, k=K,
    stride=h, padding=h, output_padding=h-1, groups=1, bias=False), h=K//2:
    y[b,o,t] = sum_{ci,l,k : h*l + k - h = t} v[b,ci,l] * w[ci,o,k],  t in [0, h*L)."""
    Bt, Ci, L = v.shape
    Ci2, Co, K = w.shape
    assert Ci == Ci2
    h = K // 2
    Tout = (L - 1) * h - 2 * h + K + (h - 1)
    full = np.zeros((Bt, Co, (L - 1) * h + K), dtype=v.dtype)   # un-cropped overlap-add
    for k in range(K):
        z = np.einsum("co,bcl->bol", w[:, :, k], v)             # [Bt, Co, L]
        full[:, :, k:k + h * (L - 1) + 1:h] += z
    return full[:, :, h:h + Tout]


# ----------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------
def uconv_block(x, sd, p, D, trace=None):
    """UConvBlock.forward, models/improved_sudormrf.py:198-220."""
    g = lambda k: sd[p + k]
    y1 = conv1x1(x, g("proj_1x1.conv.weight"), g("proj_1x1.conv.bias"))
    o1 = prelu(glob_ln(y1, g("proj_1x1.norm.gamma"), g("proj_1x1.norm.beta")),
               g("proj_1x1.act.weight"))
    raw = []       # pre-norm depthwise outputs d_k
    outs = []      # normalised levels
    cur = o1
    for k in range(D):
        d = dwconv5(cur, g(f"spp_dw.{k}.conv.weight"), g(f"spp_dw.{k}.conv.bias"),
                    1 if k == 0 else 2)
        raw.append(d)
        cur = glob_ln(d, g(f"spp_dw.{k}.norm.gamma"), g(f"spp_dw.{k}.norm.beta"))
        outs.append(cur)
    # bottom-up nearest-upsample-and-add, :214-216
    for _ in range(D - 1):
        top = outs.pop(-1)
        outs[-1] = outs[-1] + upsample_nearest2(top)
    merged = outs[-1]
    e = prelu(glob_ln(merged, g("final_norm.norm.gamma"), g("final_norm.norm.beta")),
              g("final_norm.act.weight"))
    out = conv1x1(e, g("res_conv.weight"), g("res_conv.bias")) + x
    if trace is not None:
        trace[p + "y1"] = y1
        for k, d in enumerate(raw):
            trace[p + f"d{k}"] = d
        trace[p + "merged"] = merged
        trace[p + "out"] = out
    return out


def tac(x4, sd, p, trace=None):
    """TAC.forward, models/groupcomm_sudormrf_v2.py:356-384.  x4: [Bt, G, n, L]."""
    Bt, G, n, L = x4.shape
    g = lambda k: sd[p + k]
    rows = np.transpose(x4, (0, 3, 1, 2)).reshape(-1, n)                   # (b,t,g) rows
    z = prelu(rows @ g("TAC_input.0.weight").T + g("TAC_input.0.bias"), g("TAC_input.1.weight"))
    H = z.shape[-1]
    z = z.reshape(Bt, L, G, H)
    zbar = z.mean(axis=2).reshape(Bt * L, H)
    q = prelu(zbar @ g("TAC_mean.0.weight").T + g("TAC_mean.0.bias"), g("TAC_mean.1.weight"))
    q = np.broadcast_to(q[:, None, :], (Bt * L, G, H))
    cat = np.concatenate([z.reshape(Bt * L, G, H), q], axis=2).reshape(-1, 2 * H)
    o = prelu(cat @ g("TAC_output.0.weight").T + g("TAC_output.0.bias"), g("TAC_output.1.weight"))
    o = np.transpose(o.reshape(Bt, L, G, n), (0, 2, 3, 1))                 # [Bt,G,n,L]
    o_raw = o.reshape(Bt * G, n, L)
    o_n = glob_ln(o_raw, g("TAC_norm.gamma"), g("TAC_norm.beta"))          # per (b,g)
    out = x4 + o_n.reshape(x4.shape)
    if trace is not None:
        trace[p + "q_raw"] = o_raw
        trace[p + "out"] = out
    return out


def gc_uconv_block(x, sd, p, D, G, trace=None):
    """GC_UConvBlock.forward, models/groupcomm_sudormrf_v2.py:405-418."""
    Bt, B, L = x.shape
    u = tac(x.reshape(Bt, G, B // G, L), sd, p + "TAC.", trace).reshape(Bt * G, B // G, L)
    y = uconv_block(u, sd, p + "UBlock.", D, trace)
    return y.reshape(Bt, B, L)


# ----------------------------------------------------------------------------
# whole models
# ----------------------------------------------------------------------------
def forward(cfg: ModelConfig, sd, wav, dtype=np.float64, trace=None):
    """SuDORMRF.forward (models/improved_sudormrf.py:283-301) and
    GroupCommSudoRmRf.forward (models/groupcomm_sudormrf_v2.py:302-322).

    sd: dict key -> ndarray (state_dict schema), wav: [Bt, A, T].
    Returns [Bt, S*A, T] in ``dtype``.  ``trace`` (dict) collects intermediates.
    """
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items()}
    wav = np.asarray(wav, dtype=dtype)
    T = wav.shape[-1]
    x = pad_to_appropriate_length(wav, cfg)
    s = encoder(x, sd["encoder.weight"])
    x = glob_ln(s, sd["ln.gamma"], sd["ln.beta"])
    x = conv1x1(x, sd["bottleneck.weight"], sd["bottleneck.bias"])
    if trace is not None:
        trace["enc"] = s
        trace["bottleneck"] = x
    for i in range(cfg.num_blocks):
        if cfg.variant == "improved":
            x = uconv_block(x, sd, f"sm.{i}.", cfg.upsampling_depth, trace)
        else:
            x = gc_uconv_block(x, sd, f"sm.{i}.", cfg.upsampling_depth, cfg.group_size, trace)
    m = conv1x1(prelu(x, sd["mask_net.0.weight"]), sd["mask_net.1.weight"], sd["mask_net.1.bias"])
    Bt, _, L = m.shape
    SA = cfg.num_sources * (cfg.in_audio_channels if cfg.variant == "groupcomm" else 1)
    m = np.maximum(m.reshape(Bt, SA, cfg.enc_num_basis, L), 0)          # ReLU masks :296-297
    v = m * s[:, None, :, :]                                            # :298
    if trace is not None:
        trace["masked"] = v.reshape(Bt, -1, L)
    y = decoder(v.reshape(Bt, -1, L), sd["decoder.weight"])
    return y[..., :T]                                                   # :316-318


def mixture_consistency(pr_batch, input_mixture, mix_weights_type="uniform"):
    """experiments/utils/mixture_consistency.py:14-36."""
    S = pr_batch.shape[1]
    pr_mix = pr_batch.sum(axis=1, keepdims=True)
    if mix_weights_type == "magsq":
        w = (pr_batch ** 2).mean(axis=-1, keepdims=True)
        w = w / (w.sum(axis=1, keepdims=True) + 1e-9)
    elif mix_weights_type == "uniform":
        w = 1.0 / S
    else:
        raise ValueError("Invalid mixture consistency weight type: {}".format(mix_weights_type))
    return pr_batch + w * (input_mixture - pr_mix)
