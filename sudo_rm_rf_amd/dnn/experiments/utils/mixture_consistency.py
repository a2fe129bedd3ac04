"""Mixture consistency (reference: sudo_rm_rf/dnn/experiments/utils/mixture_consistency.py:14-36).

``apply(pr_batch, input_mixture, mix_weights_type='uniform')`` -- same signature and errors.  The
uniform case (the only one the reference's runners use, run_sudormrf_gc_v2.py:154-155) is one HIP
kernel; 'magsq' (per-source energy weights) is two (row energies, correction) and inference-only: it raises under
autograd, no runner trains through it.  The GroupComm runner applies the uniform form INSIDE the
training graph (between the model and the loss), so it carries autograd: the map is linear in the estimates,
out = pr + (mix - sum_s pr)/S, hence grad_pr = g - mean_s g = the same kernel applied to g with a zero mixture, and
grad_mix = mean_s g.
"""
import torch

from .... import ops


class _MixtureConsistency(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pr, mix):
        ctx.in_dtypes = (pr.dtype, mix.dtype)
        ctx.need_mix = mix.requires_grad
        return ops.mixture_consistency(pr.detach().to(torch.float32).contiguous(),
                                       mix.detach().to(torch.float32).contiguous())

    @staticmethod
    def backward(ctx, g):
        g = g.detach().to(torch.float32).contiguous()
        zero = torch.zeros((g.shape[0], 1, g.shape[2]), dtype=torch.float32, device=g.device)
        g_pr = ops.mixture_consistency(g, zero).to(ctx.in_dtypes[0])
        g_mix = g.mean(dim=1, keepdim=True).to(ctx.in_dtypes[1]) if ctx.need_mix else None
        return g_pr, g_mix


def apply(pr_batch, input_mixture, mix_weights_type='uniform'):
    """pr_batch: [batch, n_sources, time]; input_mixture: [batch, 1, time]."""
    if mix_weights_type not in ('magsq', 'uniform'):
        raise ValueError('Invalid mixture consistency weight type: {}'
                         ''.format(mix_weights_type))
    if pr_batch.device.type != "cuda":
        raise RuntimeError("sudo_rm_rf_amd.mixture_consistency runs on an MI355X only (no CPU fallback)")
    if pr_batch.dim() != 3 or input_mixture.dim() != 3 or input_mixture.shape[1] != 1 or \
            input_mixture.shape[0] != pr_batch.shape[0] or input_mixture.shape[2] != pr_batch.shape[2]:
        raise RuntimeError("expected pr_batch [B,S,T] and input_mixture [B,1,T], got %s and %s" %
                           (tuple(pr_batch.shape), tuple(input_mixture.shape)))
    if mix_weights_type == 'magsq':
        if torch.is_grad_enabled() and (pr_batch.requires_grad or input_mixture.requires_grad):
            raise NotImplementedError("mix_weights_type='magsq' has no backward on the HIP path (the reference's "
                                      "runners train through the uniform form only)")
        return ops.mixture_consistency(pr_batch.detach().to(torch.float32).contiguous(),
                                       input_mixture.detach().to(torch.float32).contiguous(), 'magsq')
    return _MixtureConsistency.apply(pr_batch, input_mixture)
