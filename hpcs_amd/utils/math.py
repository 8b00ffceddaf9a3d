"""Mirror of the training-path part of hpcs/utils/math.py (Artanh :61-78, tanh :81-82)."""
import torch

from .. import _lib


class Artanh(torch.autograd.Function):
    """Input clamped to +-(1-1e-5), evaluated in fp64 on device, grad g/(1-x^2) with the clamped x."""

    @staticmethod
    def forward(ctx, x):
        # |x| as a 1-vector row: hyp_dist_o = 2*artanh(|row|) -> reuse that kernel (sign restored below)
        xc = x.clamp(-1 + 1e-5, 1 - 1e-5)
        ctx.save_for_backward(xc)
        flat = xc.reshape(-1, 1).contiguous().float()
        out = torch.empty(flat.shape[0], device=x.device, dtype=torch.float32)
        _lib.call("hpcs_hyp_dist_o", flat.abs(), flat.shape[0], 1, out)
        return (0.5 * out * torch.sign(flat[:, 0])).reshape(x.shape).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return g / (1 - xc ** 2)


def arctanh(x):
    return Artanh.apply(x)


def tanh(x):
    return x.clamp(-15, 15).tanh()
