"""Poincare-ball embedding heads (mirror of the `hpcs.nn.hyperbolic` namespace)."""
from . import hyp_embed as _hyp_embed

ExpMap = _hyp_embed.ExpMap
MLPExpMap = _hyp_embed.MLPExpMap

__all__ = ("ExpMap", "MLPExpMap")
