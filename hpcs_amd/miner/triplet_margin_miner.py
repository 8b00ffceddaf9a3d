"""Mirror of hpcs/miner/triplet_margin_miner.py:6-38 on HIP kernels (matrix-free margins + stable compaction)."""
import torch

from .. import _lib
from ..distances.cosine import CosineSimilarity, l2_rows
from .loss_and_miner_utils import TripletPlan, sample_with_plan

_KINDS = {"all": 0, "easy": 1, "semihard": 2, "hard": 3}


def filter_triplets(xn, a, p, n, margin, kind):
    """Returns (a', p', n') capacity-T arrays and a device int64 count (no host sync)."""
    T = a.numel()
    dev = xn.device
    ao, po, no = torch.empty_like(a), torch.empty_like(p), torch.empty_like(n)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    nbytes = _lib.call("hpcs_triplet_filter_workspace", T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.call("hpcs_triplet_filter", xn, xn.shape[1], a, p, n, T, margin, _KINDS[kind], ao, po, no, count, ws, nbytes)
    return ao, po, no, count


class RandomTripletMarginMiner:
    def __init__(self, t_per_anchor, fraction, margin=0.2, type_of_triplets="all", distance=None, rng="host", seed=0):
        if type_of_triplets not in _KINDS:
            raise ValueError(f"unknown type_of_triplets {type_of_triplets!r}")
        self.t_per_anchor = t_per_anchor
        self.fraction = fraction
        self.margin = margin
        self.type_of_triplets = type_of_triplets
        self.distance = CosineSimilarity() if distance is None else distance
        if not isinstance(self.distance, CosineSimilarity):
            raise NotImplementedError("HPCS only mines on CosineSimilarity (ultrametric_loss.py:31-39)")
        self.rng = rng
        self.seed = seed

    @torch.no_grad()
    def mine_device(self, embeddings, labels, plan=None, xn=None):
        """Sync-free form: capacity arrays + device count."""
        if plan is None:
            plan = TripletPlan.from_labels(labels, self.t_per_anchor, self.fraction)
        if xn is None:
            xn, _ = l2_rows(embeddings.detach())
        self.seed += 1
        a, p, n = sample_with_plan(labels, plan, rng=self.rng, seed=self.seed)
        return filter_triplets(xn, a, p, n, self.margin, self.type_of_triplets)

    @torch.no_grad()
    def __call__(self, embeddings, labels, ref_emb=None, ref_labels=None):
        """Reference contract: exact-size (a, p, n).  Reads the count back (one host sync)."""
        if ref_emb is not None or ref_labels is not None:
            raise NotImplementedError("ref_emb/ref_labels are never used by HPCS")
        a, p, n, count = self.mine_device(embeddings, labels)
        c = int(count.item())
        return a[:c], p[:c], n[:c]

    mine = __call__
