"""The reference's caller-side inference recipe as one call.

README.md:100-114 (and experiments/simple_whamr_evaluation.py:142-148) wrap every model() call in the same
lines: per-example mean/std normalisation of the mixture, the forward, rescaling of the estimates with the
mixture's statistics and -- for the GroupComm models -- mixture consistency.  On a GPU those are 5-7 extra
passes over [batch, sources, time] tensors in separate ATen kernels; here they are folded INTO the forward
(srf_separate): one statistics kernel over the raw mixture, the normalisation in the encoder's operand load, the
rescale and the mixture consistency in the decoder's overlap-add.  The stand-alone kernels (srf_wav_normalize /
srf_wav_denormalize, ``ops``) remain for callers that need the normalised mixture itself."""
import torch

from . import ops


def separate(model, mixture, mixture_consistency=None):
    """mixture: float tensor [batch, time] or [batch, 1, time] on the model's MI355X.
    Returns the estimated sources [batch, num_sources, time] in the mixture's own scale.

    mixture_consistency: None = apply it exactly when the model is a GroupCommSudoRmRf (what the README
    prescribes for the pre-trained GroupComm models), True / False to force."""
    if mixture.dim() == 2:
        mixture = mixture.unsqueeze(1)
    if mixture.dim() != 3 or mixture.shape[1] != 1:
        raise RuntimeError("separate() expects [batch, time] or [batch, 1, time], got %s" % (tuple(mixture.shape),))
    if mixture_consistency is None:
        mixture_consistency = type(model).__name__ == "GroupCommSudoRmRf"
    x = mixture.detach().to(torch.float32).contiguous()
    with torch.no_grad():
        if getattr(model, "in_audio_channels", 1) == 1 and hasattr(model, "_engine"):
            return model._engine().separate(model, x, bool(mixture_consistency))
        norm, stats = ops.wav_normalize(x)          # (multi-channel front ends: the three-kernel form)
        est = model(norm)
        return ops.wav_denormalize(est, stats, norm if mixture_consistency else None)
