__all__ = ["ExpMap", "MLPExpMap"]

from .hyp_embed import ExpMap, MLPExpMap
