"""Mirror of hpcs/loss/hierarchical_cosface_loss.py (HierarchicalCosFaceLoss); kernels: hpcs_cosface_fwd/bwd."""
from .ultrametric_loss import HierarchicalCosFaceLoss  # noqa: F401
