"""VN_DGCNN_partseg on gfx950 kernels -- drop-in for hpcs/nn/dgcnn/vn_dgcnn_partseg.py:7-104.

Same constructor, same state-dict keys (conv1.map_to_feat.weight, conv8.0.weight, bn8.* aliases ...), same
forward contract: x [B,3,N] fp32, l [B,num_categories] one-hot -> [B,N,out_features].

Differences in HOW (not what) it computes (DESIGN.md "Extractor"):
  * kNN is the fused score+top-k kernel (no N x N matrix);
  * mean_pool is fused into the last VN layer of each EdgeConv block;
  * std_feature.vn1's 341 mean channels (constant over N, :82-83) enter its GEMM as a per-cloud row bias;
  * conv8's 2110 channels that are constant over N (global max feature + category code) are applied as a
    per-cloud row bias instead of being repeated N times (vn_dgcnn_partseg.py:92-95).
"""
import torch
import torch.nn as nn

from ... import ops
from .utils.vn_layers import VNLinearLeakyReLU, VNMaxPool, VNStdFeature, batch_counters, bump, mean_pool
from .utils.vn_dgcnn_util import get_graph_feature, knn


class VN_DGCNN_partseg(nn.Module):
    def __init__(self, in_channels, out_features, k, dropout, pooling, num_categories):
        super().__init__()
        self.in_channels = in_channels
        self.out_features = out_features
        self.k = k
        self.dropout = dropout
        self._drop_seed, self._drop_calls = None, 0
        self.pooling = pooling
        self.num_categories = num_categories
        c, g = 64 // 3, 1024 // 3

        head_bn = {7: 64, 8: 256, 9: 256, 10: 128, 11: out_features}
        for n, ch in head_bn.items():
            setattr(self, f"bn{n}", nn.BatchNorm1d(ch))

        for i, (cin, cout) in enumerate(((2, c), (c, c), (2 * c, c), (c, c), (2 * c, c)), start=1):
            setattr(self, f"conv{i}", VNLinearLeakyReLU(cin, cout))

        if pooling == "max":
            for i in (1, 2, 3):
                setattr(self, f"pool{i}", VNMaxPool(c))
        elif pooling == "mean":
            self.pool1 = self.pool2 = self.pool3 = mean_pool

        self.conv6 = VNLinearLeakyReLU(3 * c, g, dim=4, share_nonlinearity=True)
        self.std_feature = VNStdFeature(2 * g, dim=4, normalize_frame=False)

        def head(cin, cout, bn, act=True):
            layers = [nn.Conv1d(cin, cout, kernel_size=1, bias=False), bn]
            if act:
                layers.append(nn.LeakyReLU(negative_slope=0.2))
            return nn.Sequential(*layers)

        glob_ch = 2 * g * 3 + 64                     # 2046 invariant global + 64 category = 2110
        self._glob_ch = glob_ch
        self.conv8 = head(glob_ch + 3 * c * 3, 256, self.bn8)          # 2299 -> 256
        self.conv7 = head(num_categories, 64, self.bn7)
        self.dp1 = nn.Dropout(p=dropout)
        self.conv9 = head(256, 256, self.bn9)
        self.dp2 = nn.Dropout(p=dropout)
        self.conv10 = head(256, 128, self.bn10)
        self.conv11 = head(128, out_features, self.bn11, act=False)

    # ---- helpers -----------------------------------------------------------------------------------------
    # "materialised" (default): build the [B,2C,3,N,k] graph feature, then the VN layers on it -- the reference's
    # rounding sequence (difference BEFORE the channel mix).  "gather": point-level GEMM + in-kernel gather
    # (VNLinearLeakyReLU.forward_edges); measured in round 1: same step time (cheaper GEMMs, dearer gathers and a
    # 2x larger scatter) and ~10x more fp32 noise from W_a x_nbr - W_a x_ctr cancellation -> not the default.
    edge_mode = "materialised"
    # backward of the graph layers (conv1/3/5) reduced over the graph BEFORE the channel GEMMs (ops.VNGraphLayerFn); the
    # forward arithmetic is that of "materialised" either way
    point_level_backward = True

    def _edge_block(self, h, convs, pool):
        """kNN graph on h, then the block's VN layers; mean pooling is fused into the last layer."""
        mean = self.pooling == "mean"
        if self.edge_mode == "gather":
            B, N = h.size(0), h.size(3)
            idx = knn(h.reshape(B, -1, N), self.k)
            if len(convs) == 1:
                e = convs[0].forward_edges(h, idx, pool=mean)
                return e if mean else pool(e)
            e = convs[0].forward_edges(h, idx)
            convs = convs[1:]
        elif (self.edge_mode == "materialised" and self.point_level_backward
              and ops.graph_backward_supported(h.size(3), self.k)):
            # (the graph reduction keeps one plane -- or one slab of a plane -- of 20480 edges in LDS; other geometries take
            # the edge-level backward below, which measured faster than the atomic fallback of hpcs_edge_grad_reduce)
            B, N = h.size(0), h.size(3)
            idx = knn(h.reshape(B, -1, N), self.k)
            if len(convs) == 1:
                if mean:
                    return convs[0].forward_graph(h, idx, pool_k=self.k)
                return pool(convs[0].forward_graph(h, idx))
            e = convs[0].forward_graph(h, idx)
            convs = convs[1:]
        else:
            e = get_graph_feature(h, k=self.k)
        for conv in convs[:-1]:
            e = conv(e)
        if mean:
            return convs[-1](e, pool_k=self.k)            # fused mean over the k neighbours
        return pool(convs[-1](e))

    def _head(self, seq, x, rowbias=None, mask=None, drop=None):
        conv, bn = seq[0], seq[1]
        y = ops.chan_mix(x, conv.weight[:, :, 0], rowbias, sink=(conv.weight, (slice(None), slice(None), 0)))
        slope = 0.2 if len(seq) > 2 else 1.0
        out = ops.bn_act(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, slope, mask, drop)
        if self.training:
            bump(bn)
        return out

    def next_drop_seed(self):
        """Seed of the next dropout call's in-kernel counter RNG: (torch.initial_seed() and the data-parallel rank at first
        use, running call counter), mixed -- reproducible under torch.manual_seed, different on every rank,
        statistically equivalent to torch's Philox stream, not bit-identical to it."""
        from ...miner.triplet_margin_miner import mix64
        if self._drop_seed is None:
            import torch.distributed as dist
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            self._drop_seed = mix64((int(torch.initial_seed()) & (2 ** 64 - 1)) ^ mix64(0xA0761D6478BD642F * (rank + 1) & (2 ** 64 - 1)))
        self._drop_calls += 1
        # (kept below 2^63: autograd.Function arguments are inspected as C++ int64 by torch.profiler(record_shapes=True))
        return mix64(self._drop_seed ^ mix64(self._drop_calls)) & (2 ** 63 - 1)

    def _dropout(self, p, given, seed=None):
        """(mask, drop) for ops.bn_act: an explicit multiplier tensor when the caller supplies one (parity tests), else
        (seed, p[, device word holding the seed]) for the in-kernel counter RNG -- the mask of nn.Dropout
        (vn_dgcnn_partseg.py:46,50) is never stored.  `seed` = (value, device tensor) prepared by the step driver."""
        if given is not None:
            return given, None
        if not self.training or p <= 0.0:
            return None, None
        if seed is not None:
            return None, (seed[0], p, seed[1])
        return None, (self.next_drop_seed(), p)

    # ---- forward -----------------------------------------------------------------------------------------
    def forward(self, x, l, drop_masks=None, drop_seeds=None):
        """drop_masks=(m1,m2): optional already-scaled dropout multipliers [B,256,N] (parity tests).
        drop_seeds=((seed1, dev1), (seed2, dev2)): the two dropout seeds drawn ahead by the step driver
        (next_drop_seed) together with the device words that hold them (hpcs_amd/_params.py)."""
        if not x.is_cuda:
            raise RuntimeError("hpcs_amd.VN_DGCNN_partseg runs on a HIP device only (no CPU fallback)")
        with batch_counters():            # one multi-tensor add for all BatchNorm step counters
            return self._forward(x, l, drop_masks, drop_seeds)

    def _forward(self, x, l, drop_masks, drop_seeds=None):
        B, N = x.size(0), x.size(2)
        h = x.unsqueeze(1)

        x1 = self._edge_block(h, (self.conv1, self.conv2), self.pool1)
        x2 = self._edge_block(x1, (self.conv3, self.conv4), self.pool2)
        x3 = self._edge_block(x2, (self.conv5,), self.pool3)
        x123 = torch.cat((x1, x2, x3), dim=1)                                   # [B,63,3,N]

        g6 = self.conv6(x123)                                                   # [B,341,3,N]
        # vn1 sees cat(g6, mean expanded) (:82-83): the N-constant half enters its GEMM as a row bias, and the mean over N
        # [B,341,3] is formed (and differentiated) inside that layer
        hub = ops.GradHub()               # the project+max's sparse gradient of g6 lands in vn1's data gradient
        z0, gbar = self.std_feature.frame(g6, xconst="mean", hub=hub)           # [B,3,3,N], [B,341,3]
        # invariant global feature: max over N of the frame projection of cat(g6, mean) (:84-87), fused -- the
        # [B,682,3,N] concatenation and its projection are never materialised, and the backward is sparse
        glob = ops.ProjectMaxFn.apply(g6, gbar, z0, hub)                        # [B,2046]
        loc = ops.FrameProjectFn.apply(x123, z0).view(B, -1, N)                 # [B,189,N]

        cat = self._head(self.conv7, l.view(B, -1, 1).float())                  # [B,64,1]
        vec = torch.cat((glob, cat.view(B, -1)), dim=1)                         # [B,2110]

        W8 = self.conv8[0].weight[:, :, 0]
        W8p, gc = self.conv8[0].weight, self._glob_ch
        bias8 = ops.row_linear(vec, W8[:, :gc], sink=(W8p, (slice(None), slice(0, gc), 0)))  # [B,256]
        y8 = ops.chan_mix(loc, W8[:, gc:], bias8, sink=(W8p, (slice(None), slice(gc, None), 0)))
        bn = self.bn8
        m1, d1 = self._dropout(self.dropout, None if drop_masks is None else drop_masks[0],
                               None if drop_seeds is None else drop_seeds[0])
        hh = ops.bn_act(y8, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, 0.2, m1, d1)
        if self.training:
            bump(bn)
        m2, d2 = self._dropout(self.dropout, None if drop_masks is None else drop_masks[1],
                               None if drop_seeds is None else drop_seeds[1])
        hh = self._head(self.conv9, hh, mask=m2, drop=d2)
        hh = self._head(self.conv10, hh)
        hh = self._head(self.conv11, hh)
        return hh.transpose(1, 2)
