"""Mirror of hpcs/nn/dgcnn/utils/vn_dgcnn_util.py (knn :4-10, get_graph_feature :13-41) on HIP kernels."""
from .... import ops


def knn(x, k):
    """x [B,D,N] -> int64 [B,N,k]: k nearest (self included), nearest first; ties by ascending index."""
    return ops.knn(x, k)


def get_graph_feature(x, k=20, idx=None, x_coord=None):
    """x [B,C,3,N] -> [B,2C,3,N,k] = (neighbour - centre, centre).  idx: cloud-local int64 [B,N,k]."""
    B, N = x.size(0), x.size(3)
    if idx is None:
        idx = knn(x.reshape(B, -1, N), k) if x_coord is None else knn(x_coord, k)
    return ops.EdgeGatherFn.apply(x, idx)
