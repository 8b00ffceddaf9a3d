"""Test-time decode on the device: the batch forms of
BaseSimilarityHypHC._decode_linkage (hpcs/models/base_hyp_hc.py:81-86) and get_optimal_k (hpcs/utils/scores.py:141-177).

The reference loops over the clouds of a batch on the host (scipy linkage + fcluster + sklearn per cloud, `test_step`
base_hyp_hc.py:194-200); here one launch sequence handles the whole batch, one workgroup per cloud, and the results stay
on the device until the caller asks for them.  No CPU fallback: without the HIP library every call raises.
"""
import torch

from . import _lib

MAX_POINTS = 2048          # hpcs_complete_linkage / hpcs_optimal_k static LDS plans
_INDEX = {"iou": 0, "ri": 1}


def complete_linkage(leaves, check=True):
    """leaves [B,N,E] (or [N,E]) fp32 device tensor -> Z [B,N-1,4] (or [N-1,4]) fp64 device tensor, row for row what
    scipy.cluster.hierarchy.linkage(leaves[b], method='complete', metric='cosine') returns.
    check=True reads the per-cloud status back (one host sync) and raises ValueError on non-finite distances, as scipy does."""
    single = leaves.dim() == 2
    x = leaves.unsqueeze(0) if single else leaves
    if x.dim() != 3:
        raise ValueError("leaves must be [B, N, E] or [N, E]")
    if not x.is_cuda:
        raise _lib.HpcsNativeError("hpcs_amd needs HIP-device tensors (no CPU fallback)")
    x = x.detach().to(torch.float32).contiguous()
    B, N, E = x.shape
    if N < 2:
        raise ValueError("The number of observations cannot be less than 2")        # scipy's message for n < 2
    Z = torch.empty(B, N - 1, 4, dtype=torch.float64, device=x.device)
    status = torch.empty(B, dtype=torch.int32, device=x.device)
    nbytes = _lib.call("hpcs_complete_linkage_workspace", B, N)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
    _lib.call("hpcs_complete_linkage", x, B, N, E, Z, status, ws, nbytes)
    if check and bool((status != 0).any()):
        raise ValueError("The condensed distance matrix must contain only finite values.")
    return Z[0] if single else Z


def decode_linkage(x_poincare, loss, check=True):
    """Batch _decode_linkage: leaves = project(loss.normalize_embeddings(x)) then complete linkage on cosine."""
    from .distances.poincare import project
    with torch.no_grad():
        leaves = project(loss.normalize_rows(x_poincare.detach()))
    return complete_linkage(leaves, check=check)


def optimal_k(y, Z, index="iou", num_labels=None, check=True):
    """Batch get_optimal_k: y [B,N] int labels, Z [B,N-1,4] fp64 (device) -> (best_pred [B,N] int32, best_k [B] int32,
    best_score [B] fp64), all on the device.  num_labels bounds the label values (default: max + 1, one host sync)."""
    if index not in _INDEX:
        raise ValueError("index must be 'iou' or 'ri'")
    single = Z.dim() == 2
    Zb = (Z.unsqueeze(0) if single else Z).to(torch.float64).contiguous()
    yb = (y.unsqueeze(0) if y.dim() == 1 else y).to(torch.int64).contiguous()
    if not (Zb.is_cuda and yb.is_cuda):
        raise _lib.HpcsNativeError("hpcs_amd needs HIP-device tensors (no CPU fallback)")
    B, M, _ = Zb.shape
    N = M + 1
    if yb.shape != (B, N):
        raise ValueError("y must hold one label per leaf of Z")
    if num_labels is None:
        num_labels = int(yb.max()) + 1
    pred = torch.empty(B, N, dtype=torch.int32, device=Zb.device)
    bk = torch.empty(B, dtype=torch.int32, device=Zb.device)
    bs = torch.empty(B, dtype=torch.float64, device=Zb.device)
    status = torch.empty(B, dtype=torch.int32, device=Zb.device)
    _lib.call("hpcs_optimal_k", Zb, yb, B, N, int(num_labels), _INDEX[index], pred, bk, bs, status)
    if check:
        st = status.cpu()
        if bool((st == 2).any()):
            raise ValueError("labels must lie in [0, num_labels)")
        if bool((st == 3).any()):
            raise _lib.HpcsNativeError("hpcs_optimal_k handles at most 64 distinct parts per cloud")
    return (pred[0], bk[0], bs[0]) if single else (pred, bk, bs)
