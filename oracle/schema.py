"""state_dict schema of the reference models (TEST INFRASTRUCTURE, see oracle/__init__.py).

Key names, shapes and ORDER follow what ``nn.Module.state_dict()`` yields for
the reference classes:
  * Improved ``SuDORMRF``      -- /root/reference/sudo_rm_rf/dnn/models/improved_sudormrf.py:224-281
    (blocks: UConvBlock :162-196, ConvNormAct :50-68, DilatedConvNorm :138-155,
    NormAct :99-109, GlobLN/_LayerNorm :13-22)
  * ``GroupCommSudoRmRf``      -- groupcomm_sudormrf_v2.py:232-300, TAC :343-354,
    GC_UConvBlock :388-403
"""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class ModelConfig:
    """Constructor arguments of the two reference models (same names)."""
    variant: str = "improved"        # "improved" | "groupcomm"
    out_channels: int = 128          # B
    in_channels: int = 512           # C
    num_blocks: int = 16             # U
    upsampling_depth: int = 4        # D
    enc_kernel_size: int = 21        # K
    enc_num_basis: int = 512         # N
    num_sources: int = 2             # S
    in_audio_channels: int = 1       # GroupComm only
    group_size: int = 16             # G, GroupComm only

    def ctor_kwargs(self):
        kw = dict(out_channels=self.out_channels, in_channels=self.in_channels,
                  num_blocks=self.num_blocks, upsampling_depth=self.upsampling_depth,
                  enc_kernel_size=self.enc_kernel_size, enc_num_basis=self.enc_num_basis,
                  num_sources=self.num_sources)
        if self.variant == "groupcomm":
            kw.update(in_audio_channels=self.in_audio_channels, group_size=self.group_size)
        return kw

    def as_dict(self):
        return asdict(self)

    @property
    def hop(self):
        return self.enc_kernel_size // 2

    @property
    def n_least_samples_req(self):
        # improved_sudormrf.py:244
        return (self.enc_kernel_size // 2) * 2 ** self.upsampling_depth

    def padded_length(self, T):
        # improved_sudormrf.py:303-310
        n = self.n_least_samples_req
        if T < n:
            return n
        return (T // n + (1 if T % n else 0)) * n

    def frames(self, T):
        # Conv1d(k=K, stride=K//2, padding=K//2) on the padded length
        Tp = self.padded_length(T)
        h = self.hop
        return (Tp + 2 * h - self.enc_kernel_size) // h + 1


# BASELINE.json configs (hyper-parameters per SURVEY.md §8 table).
CONFIGS = {
    "cfg1_improved_u8": ModelConfig("improved", 256, 512, 8, 5, 21, 512, 2),
    "cfg2_improved_u16": ModelConfig("improved", 256, 512, 16, 5, 21, 512, 2),
    "cfg3_groupcomm_u8": ModelConfig("groupcomm", 256, 512, 8, 5, 21, 512, 2, 1, 16),
    "cfg4_improved_u36_n2048": ModelConfig("improved", 512, 512, 36, 6, 21, 2048, 2),
    "cfg5_improved_u36_n4096": ModelConfig("improved", 512, 512, 36, 6, 21, 4096, 2),
}


def _ublock_schema(prefix, B, C, D):
    out = [
        (f"{prefix}proj_1x1.conv.weight", (C, B, 1)),
        (f"{prefix}proj_1x1.conv.bias", (C,)),
        (f"{prefix}proj_1x1.norm.gamma", (C,)),
        (f"{prefix}proj_1x1.norm.beta", (C,)),
        (f"{prefix}proj_1x1.act.weight", (1,)),
    ]
    for k in range(D):
        out += [
            (f"{prefix}spp_dw.{k}.conv.weight", (C, 1, 5)),
            (f"{prefix}spp_dw.{k}.conv.bias", (C,)),
            (f"{prefix}spp_dw.{k}.norm.gamma", (C,)),
            (f"{prefix}spp_dw.{k}.norm.beta", (C,)),
        ]
    out += [
        (f"{prefix}final_norm.norm.gamma", (C,)),
        (f"{prefix}final_norm.norm.beta", (C,)),
        (f"{prefix}final_norm.act.weight", (1,)),
        (f"{prefix}res_conv.weight", (B, C, 1)),
        (f"{prefix}res_conv.bias", (B,)),
    ]
    return out


def state_dict_schema(cfg: ModelConfig):
    """Ordered list of (key, shape) exactly as the reference's state_dict()."""
    B, C, U, D = cfg.out_channels, cfg.in_channels, cfg.num_blocks, cfg.upsampling_depth
    K, N, S = cfg.enc_kernel_size, cfg.enc_num_basis, cfg.num_sources
    if cfg.variant == "improved":
        A = 1
    elif cfg.variant == "groupcomm":
        A = cfg.in_audio_channels
    else:
        raise ValueError(cfg.variant)
    out = [
        ("encoder.weight", (N, A, K)),
        ("ln.gamma", (N,)),
        ("ln.beta", (N,)),
        ("bottleneck.weight", (B, N, 1)),
        ("bottleneck.bias", (B,)),
    ]
    for i in range(U):
        if cfg.variant == "improved":
            out += _ublock_schema(f"sm.{i}.", B, C, D)
        else:
            G = cfg.group_size
            n, h, c = B // G, B * 3 // G, C // G
            p = f"sm.{i}.TAC."
            out += [
                (p + "TAC_input.0.weight", (h, n)), (p + "TAC_input.0.bias", (h,)),
                (p + "TAC_input.1.weight", (1,)),
                (p + "TAC_mean.0.weight", (h, h)), (p + "TAC_mean.0.bias", (h,)),
                (p + "TAC_mean.1.weight", (1,)),
                (p + "TAC_output.0.weight", (n, 2 * h)), (p + "TAC_output.0.bias", (n,)),
                (p + "TAC_output.1.weight", (1,)),
                (p + "TAC_norm.gamma", (n,)), (p + "TAC_norm.beta", (n,)),
            ]
            out += _ublock_schema(f"sm.{i}.UBlock.", n, c, D)
    out += [
        ("mask_net.0.weight", (1,)),
        ("mask_net.1.weight", (S * N * A, B, 1)),
        ("mask_net.1.bias", (S * N * A,)),
        ("decoder.weight", (S * N * A, S * A, K)),
    ]
    return out


def num_params(cfg: ModelConfig):
    tot = 0
    for _, shp in state_dict_schema(cfg):
        n = 1
        for s in shp:
            n *= s
        tot += n
    return tot
