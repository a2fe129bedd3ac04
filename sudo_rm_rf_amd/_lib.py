"""ctypes binding of libsudormrf_hip.so (C ABI: include/sudormrf_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails,
an exception is raised.  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  (loads torch's libamdhip64 first so the extension binds to the same runtime)

_PKG = os.path.dirname(os.path.abspath(__file__))
# SRF_LIB: an alternative build of the same library (same-box A/B of kernel variants, tools/); default = the in-tree build
LIB_PATH = os.environ.get("SRF_LIB") or os.path.join(_PKG, "libsudormrf_hip.so")
ABI_VERSION = 15
STAT_BUCKETS = 64

SRF_OK = 0
VARIANT_IMPROVED, VARIANT_GROUPCOMM = 0, 1


class SrfError(RuntimeError):
    pass


class srf_config(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "variant", "in_audio_channels", "out_channels", "in_channels", "num_blocks",
        "upsampling_depth", "enc_kernel_size", "enc_num_basis", "num_sources", "group_size")]


class srf_norm(C.Structure):
    _fields_ = [("sums", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("prelu", C.c_void_p)]


_vp, _i, _sz, _l = C.c_void_p, C.c_int, C.c_size_t, C.c_long
_PROTOS = {
    "srf_abi_version": (_i, []),
    "srf_last_error": (C.c_char_p, []),
    "srf_set_kernel_mode": (None, [_i]),
    "srf_get_kernel_mode": (_i, []),
    "srf_profile_begin": (_i, [_vp]),
    "srf_profile_end": (_i, [_vp, C.POINTER(_i)]),
    "srf_profile_get": (_i, [_i, C.POINTER(C.c_char_p), C.POINTER(C.c_float)]),
    "srf_profile_timeline": (_i, [_i, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(_i)]),
    "srf_plan_create": (_i, [C.POINTER(srf_config), _i, _i, C.POINTER(_vp)]),
    "srf_plan_destroy": (None, [_vp]),
    "srf_plan_workspace_bytes": (_sz, [_vp]),
    "srf_plan_num_params": (_i, [_vp]),
    "srf_plan_frames": (_i, [_vp]),
    "srf_plan_padded_length": (_i, [_vp]),
    "srf_plan_num_launches": (_i, [_vp]),
    "srf_forward": (_i, [_vp, C.POINTER(_vp), _i, _vp, _vp, _vp, _sz, _vp]),
    "srf_debug_fetch": (_i, [_vp, _vp, _i, _vp, _sz, _vp]),
    "srf_encoder": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "srf_gln_stats": (_i, [_vp, _vp, _i, _l, _vp]),
    "srf_gln_apply": (_i, [_vp, _vp, C.POINTER(srf_norm), _i, _i, _i, _vp]),
    "srf_gln_apply_add": (_i, [_vp, _vp, _vp, C.POINTER(srf_norm), _i, _i, _i, _vp]),
    "srf_pw_conv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(srf_norm), _vp, _vp, _i, _vp, _i, _vp]),
    "srf_set_debug_flags": (None, [_i]),
    "srf_packed_pw_weight_bytes": (_sz, [_i, _i]),
    "srf_pack_pw_weights": (_i, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), _i, _vp]),
    "srf_pw_conv_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(srf_norm), _vp, _vp, _i, _vp, _i, _vp]),
    "srf_pw_conv_pair_supported": (_i, [_i, _i, _i, _i, _i]),
    "srf_pw_conv_pair": (_i, [_vp, _vp, _vp, _vp, C.POINTER(srf_norm), _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srf_pw_conv_pair_packed3_supported": (_i, [_i, _i, _i, _i, _i]),
    "srf_pw_conv_pair_packed3": (_i, [_vp, _vp, _vp, _vp, C.POINTER(srf_norm), _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srf_packed3_pw_weight_bytes": (_sz, [_i, _i]),
    "srf_pack3_pw_weights": (_i, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), _i, _vp]),
    "srf_pack3_forget": (None, [_vp]),
    "srf_pw_conv_packed3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(srf_norm), _vp, _vp, _vp]),
    "srf_pit_sisdr_work_bytes": (_sz, [_i, _i]),
    "srf_perm_inv_sisdr_work_bytes": (_sz, [_i, _i]),
    "srf_perm_inv_sisdr": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "srf_pit_sisdr_forward": (_i, [_vp, _vp, _i, _i, _i, C.c_float, _vp, _vp, _vp, _vp]),
    "srf_pit_sdr_forward": (_i, [_vp, _vp, _i, _i, _i, C.c_float, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "srf_pit_sisdr_match": (_i, [_vp, _i, _i, _vp, _vp]),
    "srf_pit_sisdr_backward": (_i, [_vp, _vp, _i, _i, _i, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "srf_pw_wgrad_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "srf_pw_wgrad_cols": (_i, [_vp, _vp, C.POINTER(srf_norm), _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "srf_pw_wgrad_ld": (_i, [_vp, _vp, C.POINTER(srf_norm), _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "srf_pw_wgrad": (_i, [_vp, _vp, C.POINTER(srf_norm), _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "srf_gln_bwd_scratch_bytes": (_sz, [_i, _i]),
    "srf_gln_bwd": (_i, [_vp, _vp, _vp, C.POINTER(srf_norm), _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "srf_merge_bwd": (_i, [_vp, C.POINTER(_vp), _i, C.c_long, _i, _vp]),
    "srf_dwconv5_bwd_scratch_bytes": (_sz, [_i, _i]),
    "srf_dwconv5_bwd": (_i, [_vp, _vp, C.POINTER(srf_norm), _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "srf_mask_apply": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srf_mask_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srf_prelu_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_long, _vp]),
    "srf_frames_gather": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "srf_train_saved_bytes": (_sz, [_vp]),
    "srf_train_scratch_bytes": (_sz, [_vp]),
    "srf_forward_train": (_i, [_vp, C.POINTER(_vp), _i, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "srf_backward": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "srf_backward_wav": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "srf_tac_bwd_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "srf_tac_bwd": (_i, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "srf_online_remix_scratch_bytes": (_sz, [_i, _i]),
    "srf_online_remix": (_i, [_vp, _vp, _vp, _i, _i, _i, C.c_float, _vp, _vp, _vp, _vp]),
    "srf_opt_chunk_size": (_i, []),
    "srf_clip_adam_step": (_i, [_vp, _vp, _i, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _i, _vp, _vp]),
    "srf_wav_normalize": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "srf_wav_stats": (_i, [_vp, _vp, _i, _i, _vp]),
    "srf_separate": (_i, [_vp, C.POINTER(_vp), _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "srf_wav_denormalize": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srf_dwconv5": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(srf_norm), _vp, _vp]),
    "srf_conv1d": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "srf_pyramid_supported": (_i, [_i, _i, _i]),
    "srf_pyramid_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "srf_pyramid": (_i, [_vp, _vp, C.POINTER(srf_norm), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                        C.POINTER(_vp), _i, _i, _i, _i, _vp, _vp, _vp]),
    "srf_merge": (_i, [C.POINTER(_vp), C.POINTER(srf_norm), _i, _vp, _i, _i, _i, _vp, _vp]),
    "srf_decoder_scratch_floats": (_sz, [_i, _i, _i, _i, _i]),
    "srf_decoder": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "srf_tac": (_i, [_vp, _vp, C.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _vp]),
    "srf_mixture_consistency": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "srf_mixture_consistency_magsq": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "srf_wav_info": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_l)]),
    "srf_wav_read": (_i, [C.c_char_p, _l, _l, _vp, C.POINTER(_l)]),
    "srf_feeder_create": (_i, [C.POINTER(C.c_char_p), _i, _i, _i, _i, _i, _i, _i, _i, C.c_ulonglong, C.POINTER(_vp)]),
    "srf_feeder_create_sharded": (_i, [C.POINTER(C.c_char_p), _i, _i, _i, _i, _i, _i, _i, _i, C.c_ulonglong, _i, _i, _i,
                                       C.POINTER(_vp)]),
    "srf_feeder_destroy": (None, [_vp]),
    "srf_feeder_epoch_items": (_l, [_vp, _vp, _l]),
    "srf_feeder_read_example": (_i, [C.POINTER(C.c_char_p), _i, _i, _l, _i, _i, _vp, _vp, _vp]),
    "srf_feeder_batches_per_epoch": (_l, [_vp]),
    "srf_feeder_item_frames": (_l, [_vp, _i]),
    "srf_feeder_start_epoch": (_i, [_vp, _i]),
    "srf_feeder_submit": (_i, [_vp, _vp, _vp, _vp]),
    "srf_feeder_wait": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i)]),
    "srf_feeder_normalize": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.c_float, _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None
_lock = threading.Lock()


def load():
    """Load the shared library (once).  Raises SrfError when it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SrfError(
                "libsudormrf_hip.so not found at %s -- build it with `python -m sudo_rm_rf_amd.build` "
                "(or __graft_entry__.build()).  There is no CPU / PyTorch fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise SrfError("libsudormrf_hip.so lacks symbol %s (stale build?)" % name) from e
            fn.restype = res
            fn.argtypes = args
        if lib.srf_abi_version() != ABI_VERSION:
            raise SrfError("libsudormrf_hip.so ABI version %d != expected %d" %
                           (lib.srf_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(rc, what="libsudormrf_hip"):
    if rc != SRF_OK:
        msg = load().srf_last_error()
        raise SrfError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_norm(sums=None, gamma=None, beta=None, prelu=None):
    n = srf_norm()
    n.sums = sums.data_ptr() if sums is not None else None
    n.gamma = gamma.data_ptr() if gamma is not None else None
    n.beta = beta.data_ptr() if beta is not None else None
    n.prelu = prelu.data_ptr() if prelu is not None else None
    return n
