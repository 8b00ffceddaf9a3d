"""Mirror of hpcs/miner/triplet_margin_miner.py:6-38 on HIP kernels (matrix-free margins + stable compaction)."""
import torch

from .. import _lib
from ..distances.cosine import CosineSimilarity, l2_rows
from .loss_and_miner_utils import TripletPlan, sample_with_plan

_KINDS = {"all": 0, "easy": 1, "semihard": 2, "hard": 3}
_M64 = (1 << 64) - 1


def mix64(z):
    """splitmix64 finaliser on a Python int."""
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


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
    def __init__(self, t_per_anchor, fraction, margin=0.2, type_of_triplets="all", **kwargs):
        distance = kwargs.pop("distance", None)          # the one PML base-class argument HPCS passes
        rng, seed = kwargs.pop("rng", "host"), kwargs.pop("seed", None)      # additive: RNG mode of the sampler
        if kwargs:
            raise TypeError(f"unsupported arguments {sorted(kwargs)}")
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
        # device-RNG stream = (base_seed, call counter).  `seed` is the call counter (tests rewind it to replay a draw);
        # base_seed comes from torch.initial_seed() and the data-parallel rank at first use, so torch.manual_seed
        # reproduces a run and every rank draws its own triplets.  The two are MIXED before they meet the kernels'
        # per-triplet counter: kernels hash (t ^ seed), so consecutive raw seeds would replay one stream permuted.
        self.seed = 0 if seed is None else int(seed)
        self.base_seed = None

    def next_seed(self):
        if self.base_seed is None:
            import torch.distributed as dist
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            self.base_seed = mix64((int(torch.initial_seed()) & _M64) ^ mix64(0xD1B54A32D192ED03 * (rank + 1) & _M64))
        self.seed += 1
        return mix64(self.base_seed ^ mix64(self.seed)) & (2 ** 63 - 1)

    @torch.no_grad()
    def mine_device(self, embeddings, labels, plan=None, xn=None):
        """Sync-free form: capacity arrays + device count."""
        if plan is None:
            plan = TripletPlan.from_labels(labels, self.t_per_anchor, self.fraction)
        if xn is None:
            xn, _ = l2_rows(embeddings.detach())
        a, p, n = sample_with_plan(labels, plan, rng=self.rng, seed=self.next_seed())
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
