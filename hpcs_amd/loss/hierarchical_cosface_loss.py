"""Mirror of hpcs/loss/hierarchical_cosface_loss.py on HIP kernels.

  hierarchical_loss(probabilities, targets, hierarchy_list)   :9-28   stand-alone op (hpcs_hier_nll_fwd/bwd)
  HierarchicalCosFaceLoss                                      :31-87  fused into hpcs_cosface_fwd/bwd (logits -> softmax
                                                                       -> branch sums -> NLL never leave the kernel)
"""
import torch

from .. import _lib
from .ultrametric_loss import HierarchicalCosFaceLoss, branch_ids  # noqa: F401


class _HierNLLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, targets, bid):
        M, C = prob.shape
        rows = torch.empty(M, device=prob.device, dtype=torch.float32)
        _lib.call("hpcs_hier_nll_fwd", prob, targets, bid, bid.shape[0], M, C, rows)
        mean = torch.empty(1, device=prob.device, dtype=torch.float32)
        _lib.call("hpcs_rowsum", rows, 1, M, 1.0 / M, mean)
        ctx.save_for_backward(prob, targets, bid)
        return mean[0]

    @staticmethod
    def backward(ctx, g):
        prob, targets, bid = ctx.saved_tensors
        M, C = prob.shape
        gp = torch.empty_like(prob)
        _lib.call("hpcs_hier_nll_bwd", prob, targets, bid, bid.shape[0], M, C, g.reshape(1).contiguous().float(), gp)
        return gp, None, None


def hierarchical_loss(probabilities, targets, hierarchy_list):
    """sum over levels of F.nll_loss(log(branch-summed probabilities), targets)   (hierarchical_cosface_loss.py:9-28).

    probabilities [M,C] (softmax output) on the HIP device, targets [M] int, hierarchy_list [level][branch][class ids]."""
    if not probabilities.is_cuda:
        raise _lib.HpcsNativeError("hpcs_amd operators need HIP-device tensors (there is no CPU fallback)")
    if probabilities.dim() != 2:
        raise ValueError("probabilities must be [M, C]")
    C = probabilities.shape[1]
    bid = branch_ids(hierarchy_list, C).to(probabilities.device)
    prob = probabilities.contiguous().float()
    return _HierNLLFn.apply(prob, targets.reshape(-1).to(torch.int64).contiguous(), bid)
