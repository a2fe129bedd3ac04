"""Deterministic weight / input generators (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference's default init (gamma=1, beta=0, PReLU slope=0.25:
improved_sudormrf.py:19-22, nn.PReLU default) hides bugs in the affine / slope
paths, so every parameter here is drawn at random (SURVEY.md §8c "perturb").
numpy's ``default_rng`` (PCG64) stream is stable across numpy versions, so the
same (config, seed) reproduces the same weights in the build container (where
the golden outputs are produced from the real reference) and on the GPU box.
"""
import numpy as np

from .schema import ModelConfig, state_dict_schema


def make_state_dict(cfg: ModelConfig, seed: int = 0):
    """Ordered dict key -> float32 ndarray following state_dict_schema(cfg)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in state_dict_schema(cfg):
        leaf = key.split(".")[-1]
        if leaf == "gamma":
            w = rng.uniform(0.5, 1.5, size=shape)
        elif leaf == "beta":
            w = rng.uniform(-0.3, 0.3, size=shape)
        elif shape == (1,):                      # PReLU slopes
            w = rng.uniform(0.05, 0.45, size=shape)
        elif key in ("encoder.weight", "decoder.weight"):
            # xavier-uniform like improved_sudormrf.py:252,280
            rf = shape[2]
            fan_in, fan_out = shape[1] * rf, shape[0] * rf
            b = np.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-b, b, size=shape)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            b = 1.0 / np.sqrt(fan_in)
            w = rng.uniform(-b, b, size=shape)
        elif leaf == "bias":
            w = rng.uniform(-0.2, 0.2, size=shape)
        else:
            raise KeyError(key)
        sd[key] = np.ascontiguousarray(w, dtype=np.float32)
    return sd


def make_mixture(batch: int, T: int, seed: int = 0, channels: int = 1, normalize: bool = True):
    """Synthetic mixtures [batch, channels, T], normalised per example like the
    callers do (README.md:100-103, simple_whamr_evaluation.py:142-144)."""
    rng = np.random.default_rng(1000003 * (seed + 1) + 17)
    x = rng.standard_normal(size=(batch, channels, T))
    if normalize and T > 1:
        m = x.mean(-1, keepdims=True)
        s = x.std(-1, ddof=1, keepdims=True)
        x = (x - m) / (s + 1e-9)
    return np.ascontiguousarray(x, dtype=np.float32)
