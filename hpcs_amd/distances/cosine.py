"""Mirror of hpcs/distances/cosine.py:4-16 (PML DotProductSimilarity with normalize_embeddings=True, inverted).

PML is absent: the base behaviour the reference relies on (L2 row normalisation before compute_mat, margin
and smallest_dist of an inverted metric) is provided here.  The dense matrix is an API-completeness path; the
miner and losses of this package never build it.
"""
import torch

from .. import _lib


def l2_rows(x):
    x = x.contiguous().float()
    xn = torch.empty_like(x)
    inv = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
    _lib.call("hpcs_l2_rows_fwd", x, x.shape[0], x.shape[1], xn, inv)
    return xn, inv


class CosineSimilarity:
    is_inverted = True
    normalize_embeddings = True
    p = 2
    power = 1

    def __init__(self, **kwargs):
        if kwargs:
            raise TypeError(f"unsupported arguments {sorted(kwargs)}")

    def __call__(self, query_emb, ref_emb=None):
        q, _ = l2_rows(query_emb.detach())
        r = q if ref_emb is None else l2_rows(ref_emb.detach())[0]
        return self.compute_mat(q, r)

    def compute_mat(self, query_emb, ref_emb):
        """0.5*(1 + q r^T) for already normalised rows."""
        M, Mp, E = query_emb.shape[0], ref_emb.shape[0], query_emb.shape[1]
        out = torch.empty(M, Mp, device=query_emb.device, dtype=torch.float32)
        _lib.call("hpcs_cosine_mat", query_emb.contiguous().float(), ref_emb.contiguous().float(), M, Mp, E, out)
        return out

    def pairwise_distance(self, query_emb, ref_emb):
        return 0.5 * (1 + torch.sum(query_emb * ref_emb, dim=1))

    def margin(self, x, y):
        return y - x

    def smallest_dist(self, *args):
        return torch.max(*args)
