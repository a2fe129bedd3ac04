"""Mixture consistency (reference: sudo_rm_rf/dnn/experiments/utils/mixture_consistency.py:14-36).

``apply(pr_batch, input_mixture, mix_weights_type='uniform')`` -- same signature and errors.  The
uniform case (the only one the reference's runners use, run_sudormrf_gc_v2.py:154-155) is one HIP
kernel; 'magsq' is not on the hot path and raises NotImplementedError.
"""
import torch

from .... import ops


def apply(pr_batch, input_mixture, mix_weights_type='uniform'):
    """pr_batch: [batch, n_sources, time]; input_mixture: [batch, 1, time]."""
    if mix_weights_type == 'magsq':
        raise NotImplementedError("mix_weights_type='magsq' is not implemented on the HIP path")
    elif mix_weights_type != 'uniform':
        raise ValueError('Invalid mixture consistency weight type: {}'
                         ''.format(mix_weights_type))
    if pr_batch.device.type != "cuda":
        raise RuntimeError("sudo_rm_rf_amd.mixture_consistency runs on an MI355X only (no CPU fallback)")
    if pr_batch.dim() != 3 or input_mixture.dim() != 3 or input_mixture.shape[1] != 1 or \
            input_mixture.shape[0] != pr_batch.shape[0] or input_mixture.shape[2] != pr_batch.shape[2]:
        raise RuntimeError("expected pr_batch [B,S,T] and input_mixture [B,1,T], got %s and %s" %
                           (tuple(pr_batch.shape), tuple(input_mixture.shape)))
    return ops.mixture_consistency(pr_batch.detach().to(torch.float32).contiguous(),
                                   input_mixture.detach().to(torch.float32).contiguous())
