"""Checkpoint compatibility helpers.

The reference produces two kinds of files:
  * training checkpoints: ``torch.save(model.module.cpu().state_dict(), ...)`` in the current runners
    (run_improved_sudormrf.py:221-227) but ``model.state_dict()`` of the ``DataParallel`` wrapper -- every key
    prefixed with ``module.`` -- in older ones and in many user forks;
  * the published pre-trained models: whole-module pickles (README.md:75-98 does ``torch.load(path)`` and then
    copies ``.state_dict()`` into a freshly built model).
Both load into this package's models: the class paths the pickles name (``sudo_rm_rf.dnn.models.*``) resolve
to the HIP-backed modules through the ``sudo_rm_rf`` shim package, and the helpers below normalise prefixes.
"""
from collections import OrderedDict

import torch

_PREFIX = "module."


def strip_module_prefix(state_dict):
    """Remove a leading ``module.`` (DataParallel / DistributedDataParallel) from every key that has it."""
    return OrderedDict((k[len(_PREFIX):] if k.startswith(_PREFIX) else k, v) for k, v in state_dict.items())


def add_module_prefix(state_dict):
    """The inverse: make the keys loadable into a DataParallel-wrapped model."""
    return OrderedDict((k if k.startswith(_PREFIX) else _PREFIX + k, v) for k, v in state_dict.items())


def extract_state_dict(obj):
    """state_dict of whatever a reference checkpoint file holds: a state_dict, a dict with a 'state_dict' /
    'model_state_dict' entry, a module, or a DataParallel-wrapped module."""
    if isinstance(obj, torch.nn.Module):
        obj = obj.module if hasattr(obj, "module") and isinstance(obj.module, torch.nn.Module) else obj
        return strip_module_prefix(obj.state_dict())
    if isinstance(obj, dict):
        for key in ("state_dict", "model_state_dict", "model"):
            if key in obj and isinstance(obj[key], (dict, torch.nn.Module)):
                return extract_state_dict(obj[key])
        return strip_module_prefix(obj)
    raise TypeError("cannot find a state_dict in a %s" % type(obj).__name__)


def config_from_module(module):
    """Constructor kwargs of a (pickled) reference module: the 7 (8) public attributes the README reads."""
    names = ["out_channels", "in_channels", "num_blocks", "upsampling_depth", "enc_kernel_size", "enc_num_basis",
             "num_sources"]
    kw = {n: getattr(module, n) for n in names}
    if hasattr(module, "in_audio_channels"):
        kw["in_audio_channels"] = module.in_audio_channels
        # group_size is not stored as an attribute by the reference: recover it from the TAC weight shape
        sd = module.state_dict()
        n = sd["sm.0.TAC.TAC_input.0.weight"].shape[1]
        kw["group_size"] = kw["out_channels"] // n
    return kw


def load_checkpoint(path_or_obj, model=None, map_location="cpu", strict=True):
    """Load a reference checkpoint (path or already-loaded object) into `model`; with model=None a whole-module
    pickle is rebuilt as the matching HIP-backed class.  Returns the model."""
    obj = path_or_obj
    if isinstance(path_or_obj, (str, bytes)) or hasattr(path_or_obj, "read"):
        # whole-module pickles need weights_only=False; the classes they name are this package's own
        obj = torch.load(path_or_obj, map_location=map_location, weights_only=False)
    sd = extract_state_dict(obj)
    if model is None:
        mod = obj.module if hasattr(obj, "module") and isinstance(getattr(obj, "module"), torch.nn.Module) else obj
        if not isinstance(mod, torch.nn.Module):
            raise TypeError("a bare state_dict needs the `model` argument")
        from .dnn.models import groupcomm_sudormrf_v2, improved_sudormrf
        kw = config_from_module(mod)
        cls = groupcomm_sudormrf_v2.GroupCommSudoRmRf if "group_size" in kw else improved_sudormrf.SuDORMRF
        model = cls(**kw)
    target = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
    target.load_state_dict(sd, strict=strict)
    return model
