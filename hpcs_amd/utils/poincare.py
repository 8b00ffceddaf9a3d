"""Mirror of hpcs/utils/poincare.py (the symbols the embedder uses) on row-wise HIP kernels."""
import torch

from .. import _lib

MIN_NORM = 1e-15
BALL_EPS = {torch.float32: 4e-3, torch.float64: 1e-5}


def _rows(x):
    if not x.is_cuda:
        raise _lib.HpcsNativeError("hpcs_amd operators need HIP-device tensors (there is no CPU fallback)")
    E = x.shape[-1]
    return x.reshape(-1, E).contiguous().float(), E


class _ExpMap0(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        flat, E = _rows(u)
        out = torch.empty_like(flat)
        _lib.call("hpcs_expmap0_fwd", flat, flat.shape[0], E, out)
        ctx.save_for_backward(flat)
        ctx.shape = u.shape
        return out.view(u.shape)

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        E = flat.shape[1]
        gu = torch.empty_like(flat)
        _lib.call("hpcs_expmap0_bwd", flat, g.reshape(-1, E).contiguous().float(), flat.shape[0], E, gu)
        return gu.view(ctx.shape)


def expmap0(u):
    """expmap_1(u, 0) = tanh(clamp(|u|,15)) u/|u|   (hpcs/utils/poincare.py:50-54 with p = 0)."""
    return _ExpMap0.apply(u)


def _nograd(name, *tensors):
    flat0, E = _rows(tensors[0])
    flats = [flat0] + [_rows(t)[0] for t in tensors[1:]]
    out = torch.empty_like(flat0)
    _lib.call(name, *flats, flat0.shape[0], E, out)
    return out.view(tensors[0].shape)


def lambda_(x):
    x_sqnorm = torch.sum(x.data.pow(2), dim=-1, keepdim=True)
    return 2 / (1. - x_sqnorm).clamp_min(MIN_NORM)


def expmap_1(u, p):
    """General exp-map at p.  Forward-only kernel (the training path only uses p = 0 -> expmap0)."""
    if not torch.is_grad_enabled() or not (u.requires_grad or p.requires_grad):
        return _nograd("hpcs_expmap_fwd", u, p)
    if bool((p == 0).all()):
        return expmap0(u)
    raise NotImplementedError("differentiable expmap at p != 0 is not on the HPCS training path")


def mobius_add(x, y):
    return _nograd("hpcs_mobius_add", x.detach(), y.detach())


def project(x):
    return _nograd("hpcs_project", x.detach())


def hyp_dist_o(x):
    flat, E = _rows(x.detach())
    out = torch.empty(flat.shape[0], device=x.device, dtype=torch.float32)
    _lib.call("hpcs_hyp_dist_o", flat, flat.shape[0], E, out)
    return out.view(*x.shape[:-1], 1)


def mobius_mul(x, t):
    """t (*) x = tanh(t artanh|x|) x/|x|   (hpcs/utils/poincare.py mobius_mul; forward-only toolbox op)."""
    from .math import arctanh, tanh
    n = x.norm(dim=-1, p=2, keepdim=True).clamp_min(MIN_NORM)
    return tanh(t * arctanh(n)) * (x / n)
