__all__ = ["CosineSimilarity", "hyp_lca"]

from .cosine import CosineSimilarity
from .lca import hyp_lca
