"""`MseStructuralSimilarityLoss(input_min_value, input_max_value)(prediction, target)` -- the loss
plugin point of the reference (d3f/loss_functions/structural_similarity_loss.py:5-26):

    (MSE(prediction, target) + (1 - SSIM(clip01(prediction), clip01(target)))) / 2

Forward value and d loss / d prediction come from two fused HIP stencil passes (csrc/loss.hip)
instead of piqa's ~15 torch kernels; `target` receives no gradient, as in the reference's use
(`criterion(prediction, image)` with `image` a data batch, train_denoiser/lit_module.py:119).
"""
import torch
import torch.nn as nn

from .. import ops


class _MseSsimFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, target, lo, hi):
        out, grad = ops.mse_ssim_loss(prediction.detach(), target.detach(), lo, hi)
        ctx.save_for_backward(grad)
        ctx.parts = out  # {loss, mse, ssim} on device, for logging without extra kernels
        return out[0]  # a view of this call's own result buffer (no copy kernel on the step's dependent chain)

    @staticmethod
    def backward(ctx, grad_output):
        (grad,) = ctx.saved_tensors
        return grad * grad_output, None, None, None


class MseStructuralSimilarityLoss(nn.Module):
    def __init__(self, input_min_value, input_max_value):
        super().__init__()
        self.input_min_value = input_min_value
        self.input_max_value = input_max_value

    def forward(self, prediction, target):
        if prediction.shape != target.shape:
            raise RuntimeError(f"prediction {tuple(prediction.shape)} and target {tuple(target.shape)} differ")
        return _MseSsimFunction.apply(prediction, target, float(self.input_min_value),
                                      float(self.input_max_value))

    def normalise_between_zero_and_one(self, x):
        # structural_similarity_loss.py:23-26 (host-side helper kept for API parity)
        x = (x - self.input_min_value) / (self.input_max_value - self.input_min_value)
        return x.clip(0.0, 1.0)
