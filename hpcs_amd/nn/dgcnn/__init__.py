"""Mirror of hpcs.nn.dgcnn: only the north-star backbone (vn_dgcnn_partseg) is provided."""
from .vn_dgcnn_partseg import VN_DGCNN_partseg

__all__ = ["VN_DGCNN_partseg"]
