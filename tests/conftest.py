import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)


def load_case(manifest, name):
    """(cfg, state_dict(np), wav(np), golden arrays) for a golden case."""
    from oracle.schema import ModelConfig
    from oracle.weights import make_state_dict, make_mixture
    meta = manifest["cases"][name]
    cfg = ModelConfig(**meta["config"])
    sd = make_state_dict(cfg, meta["weight_seed"])
    A = cfg.in_audio_channels if cfg.variant == "groupcomm" else 1
    wav = make_mixture(meta["batch"], meta["T"], meta["input_seed"], channels=A)
    gold = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return cfg, sd, wav, gold
