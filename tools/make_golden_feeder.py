#!/usr/bin/env python3
"""Golden fixtures for the input feeder: the UNMODIFIED reference Dataset (/root/reference/sudo_rm_rf/dnn/dataset_loader/
wham.py) run on the deterministic miniature WHAM tree of oracle/feeder_oracle.make_fake_wham; stores what its __getitem__
returned per utterance (keyed by file name: the reference keeps glob order).  Run in the build container only."""
import json
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import feeder_oracle  # noqa: E402

# the reference imports glob2 / tqdm at module level: glob2 is not installed here -> equivalent stand-in (glob.glob)
if "glob2" not in sys.modules:
    import glob
    g2 = types.ModuleType("glob2")
    g2.glob = glob.glob
    sys.modules["glob2"] = g2
sys.path.insert(0, "/root/reference")
for m in [k for k in sys.modules if k == "sudo_rm_rf" or k.startswith("sudo_rm_rf.")]:
    del sys.modules[m]
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("ref_abstract", "/root/reference/sudo_rm_rf/dnn/dataset_loader/abstract_dataset.py")
ref_abstract = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_abstract)
pkg = types.ModuleType("sudo_rm_rf"); pkg.__path__ = []
for name in ("sudo_rm_rf", "sudo_rm_rf.dnn", "sudo_rm_rf.dnn.dataset_loader"):
    mod = types.ModuleType(name); mod.__path__ = []
    sys.modules[name] = mod
sys.modules["sudo_rm_rf.dnn.dataset_loader.abstract_dataset"] = ref_abstract
sys.modules["sudo_rm_rf.dnn.dataset_loader"].abstract_dataset = ref_abstract
spec = importlib.util.spec_from_file_location("ref_wham", "/root/reference/sudo_rm_rf/dnn/dataset_loader/wham.py")
ref_wham = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_wham)

CASES = {
    "feeder_sep_clean_norm_pad": dict(task="sep_clean", timelength=0.5, normalize_audio=True, zero_pad=True, augment=False),
    "feeder_sep_clean_raw_pad": dict(task="sep_clean", timelength=0.5, normalize_audio=False, zero_pad=True, augment=False),
    "feeder_sep_noisy_norm_nopad": dict(task="sep_noisy", timelength=0.3, normalize_audio=True, zero_pad=False, augment=False),
    "feeder_enh_single_norm_pad": dict(task="enh_single", timelength=0.45, normalize_audio=True, zero_pad=True, augment=False),
}
out_dir = os.path.join(ROOT, "tests", "golden")
manifest = {}
for name, c in CASES.items():
    with tempfile.TemporaryDirectory() as tmp:
        feeder_oracle.make_fake_wham(tmp, task=c["task"], seed=11)
        ds = ref_wham.Dataset(root_dirpath=tmp, task=c["task"], split="tr", sample_rate=8000, timelength=c["timelength"],
                              normalize_audio=c["normalize_audio"], n_samples=0, zero_pad=c["zero_pad"], augment=c["augment"],
                              min_or_max="min")
        z = {}
        for i in range(len(ds)):
            mix, src = ds[i]
            z["mix:" + ds.file_names[i]] = mix.numpy()
            z["src:" + ds.file_names[i]] = src.numpy()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **z)
        manifest[name] = dict(c, seed=11, n_items=len(ds))
        print(name, len(ds), "items")
json.dump(manifest, open(os.path.join(out_dir, "FEEDER_MANIFEST.json"), "w"), indent=1, sort_keys=True)
