"""Functional torch-CPU restatement of the SuDoRM-RF forward path.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Same ATen op sequence as the reference's nn.Modules (so: same arithmetic, same
rounding order and the same CPU cost profile -- GlobLN stays the unfused
mean/pow/mean/sqrt/div chain of improved_sudormrf.py:44-47), but written as
plain functions over a state_dict.  Used for full-size parity and as
``cpu_baseline`` (kind "port") in bench.py, because /root/reference is not
present on the GPU box.  Pinned to the real reference by tests/golden.
Citations are relative to /root/reference/sudo_rm_rf/dnn/.
"""
import torch
import torch.nn.functional as F

from .schema import ModelConfig


def _gln(x, gamma, beta):
    # models/improved_sudormrf.py:44-47 and :24-27 (biased var, eps inside sqrt)
    dims = list(range(1, x.dim()))
    mean = x.mean(dim=dims, keepdim=True)
    var = torch.pow(x - mean, 2).mean(dim=dims, keepdim=True)
    normed = (x - mean) / (var + 1e-8).sqrt()
    return (gamma * normed.transpose(1, -1) + beta).transpose(1, -1)


def _ublock(x, sd, p, D, trace=None):
    # UConvBlock.forward, models/improved_sudormrf.py:198-220
    y1 = F.conv1d(x, sd[p + "proj_1x1.conv.weight"], sd[p + "proj_1x1.conv.bias"])
    cur = F.prelu(_gln(y1, sd[p + "proj_1x1.norm.gamma"], sd[p + "proj_1x1.norm.beta"]),
                  sd[p + "proj_1x1.act.weight"])
    C = cur.shape[1]
    outs, raw = [], []
    for k in range(D):
        d = F.conv1d(cur, sd[p + f"spp_dw.{k}.conv.weight"], sd[p + f"spp_dw.{k}.conv.bias"],
                     stride=1 if k == 0 else 2, padding=2, groups=C)
        raw.append(d)
        cur = _gln(d, sd[p + f"spp_dw.{k}.norm.gamma"], sd[p + f"spp_dw.{k}.norm.beta"])
        outs.append(cur)
    for _ in range(D - 1):
        top = outs.pop(-1)
        outs[-1] = outs[-1] + F.interpolate(top, scale_factor=2, mode="nearest")
    merged = outs[-1]
    e = F.prelu(_gln(merged, sd[p + "final_norm.norm.gamma"], sd[p + "final_norm.norm.beta"]),
                sd[p + "final_norm.act.weight"])
    out = F.conv1d(e, sd[p + "res_conv.weight"], sd[p + "res_conv.bias"]) + x
    if trace is not None:
        trace[p + "y1"] = y1
        for k, d in enumerate(raw):
            trace[p + f"d{k}"] = d
        trace[p + "merged"] = merged
        trace[p + "out"] = out
    return out


def _tac(x4, sd, p, trace=None):
    # TAC.forward, models/groupcomm_sudormrf_v2.py:356-384
    Bt, G, n, L = x4.shape
    rows = x4.permute(0, 3, 1, 2).contiguous().view(-1, n)
    z = F.prelu(F.linear(rows, sd[p + "TAC_input.0.weight"], sd[p + "TAC_input.0.bias"]),
                sd[p + "TAC_input.1.weight"]).view(Bt, L, G, -1)
    zbar = z.mean(2).view(Bt * L, -1)
    z = z.view(Bt * L, G, -1)
    q = F.prelu(F.linear(zbar, sd[p + "TAC_mean.0.weight"], sd[p + "TAC_mean.0.bias"]),
                sd[p + "TAC_mean.1.weight"]).unsqueeze(1).expand_as(z).contiguous()
    cat = torch.cat([z, q], 2)
    o = F.prelu(F.linear(cat.view(-1, cat.shape[-1]), sd[p + "TAC_output.0.weight"],
                         sd[p + "TAC_output.0.bias"]), sd[p + "TAC_output.1.weight"])
    o = o.view(Bt, L, G, -1).permute(0, 2, 3, 1).contiguous()
    o_raw = o.view(Bt * G, n, L)
    o_n = _gln(o_raw, sd[p + "TAC_norm.gamma"], sd[p + "TAC_norm.beta"])
    out = x4 + o_n.view(x4.shape)
    if trace is not None:
        trace[p + "q_raw"] = o_raw
        trace[p + "out"] = out
    return out


def forward(cfg: ModelConfig, sd, wav, trace=None):
    """SuDORMRF.forward (models/improved_sudormrf.py:283-301) /
    GroupCommSudoRmRf.forward (models/groupcomm_sudormrf_v2.py:302-322).
    sd: dict key -> torch CPU tensor; wav: [Bt, A, T]; dtype follows the inputs."""
    T = wav.shape[-1]
    Tp = cfg.padded_length(T)
    x = torch.zeros(list(wav.shape[:-1]) + [Tp], dtype=wav.dtype)
    x[..., :T] = wav
    h = cfg.enc_kernel_size // 2
    s = F.conv1d(x, sd["encoder.weight"], None, stride=h, padding=h)
    x = _gln(s, sd["ln.gamma"], sd["ln.beta"])
    x = F.conv1d(x, sd["bottleneck.weight"], sd["bottleneck.bias"])
    if trace is not None:
        trace["enc"] = s
        trace["bottleneck"] = x
    D = cfg.upsampling_depth
    for i in range(cfg.num_blocks):
        if cfg.variant == "improved":
            x = _ublock(x, sd, f"sm.{i}.", D, trace)
        else:
            Bt, B, L = x.shape
            G = cfg.group_size
            u = _tac(x.view(Bt, G, -1, L), sd, f"sm.{i}.TAC.", trace).view(Bt * G, -1, L)
            x = _ublock(u, sd, f"sm.{i}.UBlock.", D, trace).view(Bt, B, L)
    m = F.conv1d(F.prelu(x, sd["mask_net.0.weight"]), sd["mask_net.1.weight"], sd["mask_net.1.bias"])
    SA = cfg.num_sources * (cfg.in_audio_channels if cfg.variant == "groupcomm" else 1)
    m = torch.relu(m.view(m.shape[0], SA, cfg.enc_num_basis, -1))
    v = m * s.unsqueeze(1)
    v = v.view(v.shape[0], -1, v.shape[-1])
    if trace is not None:
        trace["masked"] = v
    y = F.conv_transpose1d(v, sd["decoder.weight"], None, stride=h, padding=h, output_padding=h - 1)
    return y[..., :T]


def mixture_consistency(pr_batch, input_mixture, mix_weights_type="uniform"):
    """experiments/utils/mixture_consistency.py:14-36."""
    S = pr_batch.shape[1]
    pr_mix = torch.sum(pr_batch, 1, keepdim=True)
    if mix_weights_type == "magsq":
        w = torch.mean(pr_batch ** 2, -1, keepdim=True)
        w = w / (torch.sum(w, 1, keepdim=True) + 1e-9)
    elif mix_weights_type == "uniform":
        w = 1.0 / S
    else:
        raise ValueError("Invalid mixture consistency weight type: {}".format(mix_weights_type))
    return pr_batch + w * (input_mixture - pr_mix)


def to_torch(sd_np, dtype=torch.float32):
    return {k: torch.as_tensor(v).to(dtype) for k, v in sd_np.items()}
