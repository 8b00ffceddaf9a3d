"""Mirror of hpcs/nn/dgcnn/utils/vn_layers.py on HIP kernels.

Same constructor signatures and state-dict names (map_to_feat / map_to_dir / batchnorm.bn.*); the torch
sub-modules are parameter containers only -- forward runs hpcs_amd.ops (fused GEMM + VN kernels).
"""
import torch
import torch.nn as nn

from .... import ops

EPS = 1e-6


class batch_counters:
    """Collects the BatchNorm `num_batches_tracked` bumps of one forward pass into ONE multi-tensor add instead of a
    tiny kernel per layer (13 per step in VN_DGCNN_partseg).  Outside the context every layer bumps its own counter."""
    _active = None

    def __enter__(self):
        self._outer = batch_counters._active
        batch_counters._active = []
        return self

    def __exit__(self, *exc):
        pending, batch_counters._active = batch_counters._active, self._outer
        if pending:
            torch._foreach_add_(pending, 1)
        return False


def bump(bn):
    if bn.num_batches_tracked is None:
        return
    if batch_counters._active is not None:
        batch_counters._active.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


def _flat(x):
    """[B,C,3,*S] view with all trailing dims kept (kernels see L = prod(S))."""
    if x.dim() < 3 or x.size(2) != 3:
        raise ValueError("vector-neuron features must have shape [B, C, 3, ...]")
    return x


class VNLinear(nn.Module):                                     # vn_layers.py:13-24
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.map_to_feat = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x):
        return ops.chan_mix(_flat(x), self.map_to_feat.weight, sink=(self.map_to_feat.weight, ...))


class VNLeakyReLU(nn.Module):                                  # vn_layers.py:27-45
    def __init__(self, in_channels, share_nonlinearity=False, negative_slope=0.2):
        super().__init__()
        self.map_to_dir = nn.Linear(in_channels, 1 if share_nonlinearity else in_channels, bias=False)
        self.negative_slope = negative_slope
        self.register_buffer("_eye", torch.eye(in_channels), persistent=False)

    def forward(self, x):
        return ops.vn_layer(_flat(x), self._eye, self.map_to_dir.weight, None, None, None, None, False,
                            slope=self.negative_slope, flags=ops.VN_LRELU)


class VNBatchNorm(nn.Module):                                  # vn_layers.py:112-132
    def __init__(self, num_features, dim):
        super().__init__()
        self.dim = dim
        if dim == 3 or dim == 4:
            self.bn = nn.BatchNorm1d(num_features)
        elif dim == 5:
            self.bn = nn.BatchNorm2d(num_features)
        self.register_buffer("_eye", torch.eye(num_features), persistent=False)

    def forward(self, x):
        bn = self.bn
        training = self.training or not bn.track_running_stats
        out = ops.vn_layer(_flat(x), self._eye, None, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                           flags=ops.VN_BN)
        if training:
            bump(bn)
        return out


class VNLinearLeakyReLU(nn.Module):                            # vn_layers.py:48-77
    def __init__(self, in_channels, out_channels, dim=5, share_nonlinearity=False, negative_slope=0.2):
        super().__init__()
        self.dim = dim
        self.negative_slope = negative_slope
        self.map_to_feat = nn.Linear(in_channels, out_channels, bias=False)
        self.batchnorm = VNBatchNorm(out_channels, dim=dim)
        self.map_to_dir = nn.Linear(in_channels, 1 if share_nonlinearity else out_channels, bias=False)

    def forward(self, x, pool_k=0, xconst=None, hub=None):
        """pool_k>0 additionally averages over the trailing dim of size pool_k (fused mean_pool).
        xconst [B,Cc,3]: the last Cc input channels given as per-cloud constants instead of expanded over N."""
        bn = self.batchnorm.bn
        out = ops.vn_layer(_flat(x), self.map_to_feat.weight, self.map_to_dir.weight, bn.weight, bn.bias, bn.running_mean,
                           bn.running_var, self.training, slope=self.negative_slope, pool_k=pool_k, xconst=xconst, hub=hub)
        if self.training:
            bump(bn)
        return out


    def forward_graph(self, feat, idx, pool_k=0):
        """forward(get_graph_feature(feat, idx=idx), pool_k) with the same forward arithmetic and a point-level backward
        (ops.VNGraphLayerFn): no edge-level data-gradient / weight-gradient GEMM, no scatter of an edge-sized gradient."""
        if self.map_to_dir.weight.shape[0] == 1:             # shared direction: the generic layer on the materialised edge tensor
            from .vn_dgcnn_util import get_graph_feature
            return self.forward(get_graph_feature(feat, k=idx.shape[-1], idx=idx), pool_k=pool_k)
        bn = self.batchnorm.bn
        out = ops.vn_graph_layer(feat, idx, self.map_to_feat.weight, self.map_to_dir.weight, bn.weight, bn.bias,
                                 bn.running_mean, bn.running_var, self.training, slope=self.negative_slope, pool_k=pool_k)
        if self.training:
            bump(bn)
        return out

    def forward_edges(self, feat, idx, pool=False):
        """Equivalent to forward(get_graph_feature(feat, idx=idx)) [then mean over k if pool], computed from the
        point features: no [B,2C,3,N,k] edge tensor, and a channel GEMM k times smaller (ops.VNEdgeLayerFn)."""
        if self.map_to_dir.weight.shape[0] == 1:
            from .vn_dgcnn_util import get_graph_feature
            return self.forward(get_graph_feature(feat, k=idx.shape[-1], idx=idx), pool_k=idx.shape[-1] if pool else 0)
        bn = self.batchnorm.bn
        out = ops.vn_edge_layer(feat, idx, self.map_to_feat.weight, self.map_to_dir.weight, bn.weight, bn.bias,
                                bn.running_mean, bn.running_var, self.training, slope=self.negative_slope, pool=pool)
        if self.training:
            bump(bn)
        return out


class VNMaxPool(nn.Module):                                    # vn_layers.py:135-149
    def __init__(self, in_channels):
        super().__init__()
        self.map_to_dir = nn.Linear(in_channels, in_channels, bias=False)

    def forward(self, x):
        d = ops.chan_mix(x.detach(), self.map_to_dir.weight.detach())
        return ops.VNMaxPoolFn.apply(x, d)


def mean_pool(x, dim=-1, keepdim=False):                       # vn_layers.py:152-153
    if dim not in (-1, x.dim() - 1):
        return x.mean(dim=dim, keepdim=keepdim)
    out = ops.RowMeanFn.apply(x)
    return out.unsqueeze(-1) if keepdim else out


class VNStdFeature(nn.Module):                                 # vn_layers.py:156-205
    def __init__(self, in_channels, dim=4, normalize_frame=False, share_nonlinearity=False, negative_slope=0.2):
        super().__init__()
        if dim not in (3, 4, 5):
            raise ValueError("VNStdFeature: dim must be 3, 4 or 5 (vn_layers.py:197-202)")
        self.dim = dim
        self.normalize_frame = normalize_frame
        self.vn1 = VNLinearLeakyReLU(in_channels, in_channels // 2, dim=dim, share_nonlinearity=share_nonlinearity,
                                     negative_slope=negative_slope)
        self.vn2 = VNLinearLeakyReLU(in_channels // 2, in_channels // 4, dim=dim, share_nonlinearity=share_nonlinearity,
                                     negative_slope=negative_slope)
        self.vn_lin = nn.Linear(in_channels // 4, 2 if normalize_frame else 3, bias=False)

    def frame(self, x, xconst=None, hub=None):
        """xconst: the N-constant half of vn1's input (tensor), or "mean": vn1 forms it as the mean of x over the
        positions and the call returns (z0, mean)."""
        h = self.vn1(x, xconst=xconst, hub=hub)            # hub: ops.GradHub for sparse gradients of x
        xmean = None
        if isinstance(xconst, str):
            h, xmean = h
        z = self.vn2(h)
        z = ops.chan_mix(z, self.vn_lin.weight, sink=(self.vn_lin.weight, ...))        # [B, 3 (or 2), 3, *S]
        if self.normalize_frame:
            # Gram-Schmidt of the two predicted vectors + their cross product (vn_layers.py:180-193): a handful of
            # device ops on [B,3,*S] tensors -- off the HPCS path (vn_dgcnn_partseg.py:39 builds normalize_frame=False)
            v1 = z[:, 0]
            u1 = v1 / (torch.sqrt((v1 * v1).sum(1, keepdim=True)) + 1e-6)
            v2 = z[:, 1]
            v2 = v2 - (v2 * u1).sum(1, keepdim=True) * u1
            u2 = v2 / (torch.sqrt((v2 * v2).sum(1, keepdim=True)) + 1e-6)
            u3 = torch.cross(u1, u2, dim=1)
            z0 = torch.stack([u1, u2, u3], dim=1).transpose(1, 2).contiguous()
        else:
            z0 = z.transpose(1, 2).contiguous()
        return z0 if xmean is None else (z0, xmean)

    def forward(self, x):
        z0 = self.frame(x)
        B, C = x.shape[0], x.shape[1]
        # einsum('bij...,bjk...->bik...') for dim 3 / 4 / 5: the positions are one flattened axis for the kernel
        x_std = ops.FrameProjectFn.apply(x.reshape(B, C, 3, -1), z0.reshape(B, 3, 3, -1)).reshape(x.shape)
        return x_std, z0
