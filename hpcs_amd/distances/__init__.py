"""Similarity operators of the hot path (mirror of the `hpcs.distances` namespace):
`CosineSimilarity` drives the triplet miner, `hyp_lca` the HypHC loss.  Both run on the HIP library."""
from . import cosine as _cosine
from . import lca as _lca

CosineSimilarity = _cosine.CosineSimilarity
hyp_lca = _lca.hyp_lca

__all__ = ("CosineSimilarity", "hyp_lca")
