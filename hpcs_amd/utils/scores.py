"""Mirror of hpcs/utils/scores.py for the part on the test-time path: get_optimal_k (:141-177) and remap_labels
(:124-139), evaluated by hpcs_optimal_k on the device.  The binary-score helpers of that file (compute_scores,
confusion-matrix utilities) are report formatting around sklearn and stay with the caller."""
import numpy as np
import torch

from .. import decode as _decode


def remap_labels(y_true):
    """Relabels to 0..K-1 in sorted order of the labels present (scores.py:124-139); tensors stay on their device."""
    if isinstance(y_true, torch.Tensor):
        return torch.unique(y_true, return_inverse=True)[1]
    if isinstance(y_true, np.ndarray):
        return np.unique(y_true, return_inverse=True)[1].reshape(y_true.shape)
    raise TypeError("Only allowed type are np.ndarray or torch.Tensor")


def get_optimal_k(y, linkage_matrix, index, device=None):
    """Drop-in for the reference call `get_optimal_k(targets[i].cpu(), Z, 'iou')`: one cloud, host arrays in and out
    -> (best_pred np.int32 [N] or None, best_k int, best_score float).  Batches should call hpcs_amd.decode.optimal_k."""
    dev = device or (linkage_matrix.device if isinstance(linkage_matrix, torch.Tensor) and linkage_matrix.is_cuda else "cuda")
    Z = torch.as_tensor(np.asarray(linkage_matrix) if not isinstance(linkage_matrix, torch.Tensor) else linkage_matrix)
    yy = torch.as_tensor(np.asarray(y) if not isinstance(y, torch.Tensor) else y)
    pred, k, score = _decode.optimal_k(yy.to(dev), Z.to(dev), index=index)
    k = int(k)
    if k == 0:
        return None, 0, 0.0
    return pred.cpu().numpy(), k, float(score)
