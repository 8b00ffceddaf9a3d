"""Mirror of hpcs/distances/lca.py:37-52 (hyp_lca) on the closed-form HIP kernel (DESIGN.md "hyp_lca")."""
import torch

from .. import _lib


class _HypLcaDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        E = a.shape[-1]
        af, bf = a.reshape(-1, E).contiguous().float(), b.reshape(-1, E).contiguous().float()
        d = torch.empty(af.shape[0], device=a.device, dtype=torch.float32)
        _lib.call("hpcs_hyp_lca_fwd", af, bf, af.shape[0], E, d, None)
        ctx.save_for_backward(af, bf)
        ctx.shape = a.shape
        return d.view(*a.shape[:-1], 1)

    @staticmethod
    def backward(ctx, g):
        af, bf = ctx.saved_tensors
        E = af.shape[1]
        ga, gb = torch.empty_like(af), torch.empty_like(bf)
        _lib.call("hpcs_hyp_lca_bwd", af, bf, g.reshape(-1).contiguous().float(), af.shape[0], E, ga, gb)
        return ga.view(ctx.shape), gb.view(ctx.shape)


class _HypLcaCoord(torch.autograd.Function):
    """The projection point itself, differentiable like the reference's (lca.py:37-52)."""

    @staticmethod
    def forward(ctx, a, b):
        E = a.shape[-1]
        af, bf = a.reshape(-1, E).contiguous().float(), b.reshape(-1, E).contiguous().float()
        coord = torch.empty_like(af)
        _lib.call("hpcs_hyp_lca_fwd", af, bf, af.shape[0], E, None, coord)
        ctx.save_for_backward(af, bf)
        ctx.shape = a.shape
        return coord.view(a.shape)

    @staticmethod
    def backward(ctx, g):
        af, bf = ctx.saved_tensors
        E = af.shape[1]
        ga, gb = torch.empty_like(af), torch.empty_like(bf)
        _lib.call("hpcs_hyp_lca_coord_bwd", af, bf, g.reshape(-1, E).contiguous().float(), af.shape[0], E, ga, gb)
        return ga.view(ctx.shape), gb.view(ctx.shape)


def hyp_lca(a, b, return_coord=True):
    """Projection of the origin on the geodesic a-b, or (return_coord=False) its distance to the origin [..,1]."""
    if not a.is_cuda:
        raise _lib.HpcsNativeError("hpcs_amd operators need HIP-device tensors (there is no CPU fallback)")
    if not return_coord:
        return _HypLcaDist.apply(a, b)
    return _HypLcaCoord.apply(a, b)
