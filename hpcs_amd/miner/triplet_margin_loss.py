"""Mirror of hpcs/miner/triplet_margin_loss.py:9-65 (TripletMarginLoss + AvgNonZeroReducer) on HIP kernels."""
import torch

from .. import _lib
from ..distances.cosine import CosineSimilarity


class _TripletMarginFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xn, a, p, n, count, margin):
        sn = torch.empty(2, dtype=torch.float64, device=xn.device)
        T = a.numel()
        _lib.call("hpcs_triplet_margin_fwd", xn, xn.shape[1], a, p, n, count, T, margin, sn)
        ctx.save_for_backward(xn, a, p, n, count, sn)
        ctx.margin = margin
        nnz = sn[1]
        return torch.where(nnz > 0, sn[0] / nnz.clamp_min(1.0), torch.zeros_like(nnz)).float()

    @staticmethod
    def backward(ctx, g):
        xn, a, p, n, count, sn = ctx.saved_tensors
        gxn = torch.zeros_like(xn)
        _lib.call("hpcs_triplet_margin_bwd", xn, xn.shape[1], a, p, n, count, a.numel(), ctx.margin, sn,
                  g.reshape(1).contiguous().float(), gxn)
        return gxn, None, None, None, None, None


class TripletMarginLoss(torch.nn.Module):
    def __init__(self, margin=0.05, swap=False, smooth_loss=False, triplets_per_anchor="all", **kwargs):
        super().__init__()
        if swap or smooth_loss:
            raise NotImplementedError("swap / smooth_loss are never enabled by HPCS (ultrametric_loss.py:40)")
        distance = kwargs.pop("distance", None)          # the one PML base-class argument HPCS passes
        if kwargs:
            raise TypeError(f"unsupported arguments {sorted(kwargs)}")
        self.margin = margin
        self.swap = swap
        self.smooth_loss = smooth_loss
        self.triplets_per_anchor = triplets_per_anchor
        self.distance = CosineSimilarity() if distance is None else distance

    def compute_loss(self, embeddings, labels, indices_tuple, ref_emb=None, ref_labels=None):
        """PML loss-dict contract (triplet_margin_loss.py:34-62): per-triplet relu(an - ap + margin), unreduced.  An
        API-surface method (the training path calls forward(), whose kernels fuse the AvgNonZero reduction); the
        per-triplet values are formed with device gathers of the normalised rows."""
        from ..loss.ultrametric_loss import L2RowsFn
        if ref_emb is not None or ref_labels is not None:
            raise NotImplementedError("ref_emb/ref_labels are never used by HPCS")
        if indices_tuple is None:
            raise ValueError("TripletMarginLoss needs mined triplets (indices_tuple)")
        a, p, n = indices_tuple
        if a.numel() == 0:
            return {"loss": {"losses": 0, "indices": None, "reduction_type": "already_reduced"}}
        xn = L2RowsFn.apply(embeddings)
        ap = self.distance.pairwise_distance(xn[a], xn[p])
        an = self.distance.pairwise_distance(xn[a], xn[n])
        loss = torch.nn.functional.relu(self.distance.margin(ap, an) + self.margin)
        return {"loss": {"losses": loss, "indices": indices_tuple, "reduction_type": "triplet"}}

    def forward(self, embeddings, labels=None, indices_tuple=None, count=None):
        from ..loss.ultrametric_loss import L2RowsFn
        if indices_tuple is None:
            raise ValueError("TripletMarginLoss needs mined triplets (indices_tuple)")
        a, p, n = indices_tuple
        if count is None:
            count = torch.full((1,), a.numel(), dtype=torch.int64, device=embeddings.device)
        if a.numel() == 0:
            return torch.sum(embeddings * 0)
        xn = L2RowsFn.apply(embeddings)
        return _TripletMarginFn.apply(xn, a.contiguous(), p.contiguous(), n.contiguous(), count, float(self.margin))
