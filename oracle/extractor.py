"""Oracle (TEST INFRASTRUCTURE) for the VN-DGCNN feature extractor, rows a1-a6.

Torch-CPU restatement, functional style: every layer takes its tensors
explicitly and the whole network takes a ``params`` dict keyed by the
reference's state-dict names (``conv1.map_to_feat.weight`` ...), so a
reference checkpoint drives it unchanged.  Gradients come from torch autograd
on these functions.

Reference lines restated (paths relative to /root/reference):
  knn                      hpcs/nn/dgcnn/utils/vn_dgcnn_util.py:4-10
  get_graph_feature        hpcs/nn/dgcnn/utils/vn_dgcnn_util.py:13-41
  VNLinearLeakyReLU        hpcs/nn/dgcnn/utils/vn_layers.py:48-77
  VNBatchNorm              hpcs/nn/dgcnn/utils/vn_layers.py:112-132
  VNMaxPool / mean_pool    hpcs/nn/dgcnn/utils/vn_layers.py:135-153
  VNStdFeature             hpcs/nn/dgcnn/utils/vn_layers.py:156-205
  VN_DGCNN_partseg.forward hpcs/nn/dgcnn/vn_dgcnn_partseg.py:59-104
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

VN_EPS = 1e-6          # vn_layers.py:10
BN_EPS = 1e-5          # torch.nn.BatchNorm default
BN_MOMENTUM = 0.1      # torch.nn.BatchNorm default
SLOPE = 0.2

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def _clib():
    """The C restatement of knn (oracle/knn_oracle.c), built by __graft_entry__.build()."""
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "libhpcs_oracle.so")
        if not os.path.exists(path):
            from . import build_c
            build_c.build()
        _CLIB = ctypes.CDLL(path)
        _CLIB.hpcs_oracle_knn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _CLIB.hpcs_oracle_knn.restype = ctypes.c_int
    return _CLIB


# --------------------------------------------------------------------------- a1
def knn_formula(x, k):
    """vn_dgcnn_util.py:4-10 with torch ops: negated squared distance, then top-k.

    Tie order follows torch.topk (implementation defined).  Used to check
    ``knn_exact`` and the golden fixtures; not used for bit-exact comparisons.
    """
    gram = torch.matmul(x.transpose(2, 1), x)
    sq = torch.sum(x * x, dim=1, keepdim=True)
    neg_d2 = -sq - (-2 * gram) - sq.transpose(2, 1)
    return neg_d2.topk(k=k, dim=-1)[1]


def knn_exact(x, k, return_dist=False):
    """Bit-defined kNN: the arithmetic the HIP kernel must reproduce exactly.

    * gram[i][j]  = k-ordered fp32 fma chain over d   (== MKL sgemm here, probed; == gfx950 f32 MFMA)
    * sq[j]       = torch's cascade: sequential fp32 sums of 16 squares, chunk sums added
                    sequentially, <16 remainder summed sequentially and added first
    * score[i][j] = ((-sq[j]) - (-2*gram)) - sq[i]        (vn_dgcnn_util.py:7 evaluation order)
    * order       = score descending, index ascending among exact ties (our rule; torch's is unspecified)
    Returns int64 [B,N,k] (and the fp32 scores of the winners if asked).
    """
    x = x.detach().contiguous().to(torch.float32)
    B, D, N = x.shape
    idx = torch.empty(B, N, k, dtype=torch.int64)
    dist = torch.empty(B, N, k, dtype=torch.float32)
    rc = _clib().hpcs_oracle_knn(x.data_ptr(), B, D, N, k, idx.data_ptr(), dist.data_ptr())
    if rc != 0:
        raise ValueError(f"hpcs_oracle_knn failed rc={rc}")
    return (idx, dist) if return_dist else idx


def knn_scores(x):
    """Full [B,N,N] score matrix by the reference formula (torch ops)."""
    gram = torch.matmul(x.transpose(2, 1), x)
    sq = torch.sum(x * x, dim=1, keepdim=True)
    return -sq - (-2 * gram) - sq.transpose(2, 1)


# --------------------------------------------------------------------------- a2
def graph_feature(x, k, idx=None, x_coord=None, knn_fn=knn_formula):
    """[B,C,3,N] -> [B,2C,3,N,k]: (neighbour - centre, centre) per edge."""
    B, C, _, N = x.shape
    flat = x.reshape(B, C * 3, N)
    if idx is None:
        idx = knn_fn(flat if x_coord is None else x_coord, k)
    rows = flat.permute(0, 2, 1)                       # [B,N,C*3]
    nbr = torch.gather(rows.unsqueeze(1).expand(B, N, N, C * 3), 2,
                       idx.unsqueeze(-1).expand(B, N, k, C * 3))   # [B,N,k,C*3]
    ctr = rows.unsqueeze(2).expand(B, N, k, C * 3)
    edge = torch.cat(((nbr - ctr).reshape(B, N, k, C, 3), ctr.reshape(B, N, k, C, 3)), dim=3)
    return edge.permute(0, 3, 4, 1, 2).contiguous()


# --------------------------------------------------------------------------- a3
def vn_batchnorm(p, weight, bias, running_mean, running_var, training):
    """VNBatchNorm: batch-norm the vector norms, rescale the vectors. p [B,C,3,...]"""
    norm = torch.linalg.vector_norm(p, dim=2) + VN_EPS
    norm_bn = F.batch_norm(norm, running_mean, running_var, weight, bias, training, BN_MOMENTUM, BN_EPS)
    return p / norm.unsqueeze(2) * norm_bn.unsqueeze(2)


def vn_leaky(p, d, slope=SLOPE):
    dot = (p * d).sum(2, keepdim=True)
    keep = (dot >= 0).to(p.dtype)
    dsq = (d * d).sum(2, keepdim=True)
    return slope * p + (1 - slope) * (keep * p + (1 - keep) * (p - (dot / (dsq + VN_EPS)) * d))


def _chan_mix(w, x):
    """nn.Linear over the channel axis (dim 1) of [B,C,3,...]."""
    return F.linear(x.transpose(1, -1), w).transpose(1, -1)


def vn_linear_leaky_relu(x, P, prefix, training):
    """VNLinearLeakyReLU.forward.  P holds ``{prefix}.map_to_feat.weight`` etc."""
    p = _chan_mix(P[prefix + ".map_to_feat.weight"], x)
    p = vn_batchnorm(p, P[prefix + ".batchnorm.bn.weight"], P[prefix + ".batchnorm.bn.bias"],
                     P[prefix + ".batchnorm.bn.running_mean"], P[prefix + ".batchnorm.bn.running_var"], training)
    d = _chan_mix(P[prefix + ".map_to_dir.weight"], x)
    return vn_leaky(p, d)


# --------------------------------------------------------------------------- a4
def mean_pool(x, dim=-1, keepdim=False):
    return x.mean(dim=dim, keepdim=keepdim)


def vn_max_pool(x, w_dir):
    """VNMaxPool: pick, per channel, the neighbour with the largest <x, W x>."""
    d = _chan_mix(w_dir, x)
    dot = (x * d).sum(2, keepdim=True)
    arg = dot.max(dim=-1, keepdim=True)[1]
    return torch.gather(x, -1, arg.expand(*x.shape[:-1], 1)).squeeze(-1)


# --------------------------------------------------------------------------- a5
def vn_std_feature(x, P, prefix, training):
    """VNStdFeature (normalize_frame=False, dim=4): returns (x_std, z0)."""
    z = vn_linear_leaky_relu(x, P, prefix + ".vn1", training)
    z = vn_linear_leaky_relu(z, P, prefix + ".vn2", training)
    z = _chan_mix(P[prefix + ".vn_lin.weight"], z)        # [B,3,3,N]
    z0 = z.transpose(1, 2)
    x_std = torch.einsum("bijm,bjkm->bikm", x, z0)
    return x_std, z0


# --------------------------------------------------------------------------- a6
def _conv_bn_act(x, P, conv, training, act=True):
    """Conv1d(k=1,bias=False) + BatchNorm1d (+ LeakyReLU 0.2) of the nn.Sequential heads."""
    w = P[conv + ".0.weight"]
    y = torch.matmul(w[:, :, 0], x)
    y = F.batch_norm(y, P[conv + ".1.running_mean"], P[conv + ".1.running_var"],
                     P[conv + ".1.weight"], P[conv + ".1.bias"], training, BN_MOMENTUM, BN_EPS)
    return F.leaky_relu(y, SLOPE) if act else y


def vn_dgcnn_partseg(x, l, P, k, training=False, pooling="mean", drop_masks=None, knn_fn=knn_formula,
                     return_aux=False):
    """VN_DGCNN_partseg.forward.  x [B,3,N], l [B,num_categories] -> [B,N,E].

    ``P`` is a state-dict-like mapping (running stats are updated in place when
    training).  ``drop_masks`` = (m1, m2): already-scaled dropout multipliers of
    shape [B,256,N] applied after conv8 / conv9 (None = no dropout), so that
    parity tests control the RNG-dependent part (SURVEY section 7 "Dropout").
    """
    B, _, N = x.shape
    aux = {}
    h = x.unsqueeze(1)

    def pool(t, name):
        if pooling == "mean":
            return mean_pool(t)
        return vn_max_pool(t, P[name + ".map_to_dir.weight"])

    e = graph_feature(h, k, knn_fn=knn_fn)
    e = vn_linear_leaky_relu(e, P, "conv1", training)
    e = vn_linear_leaky_relu(e, P, "conv2", training)
    x1 = pool(e, "pool1")
    e = graph_feature(x1, k, knn_fn=knn_fn)
    e = vn_linear_leaky_relu(e, P, "conv3", training)
    e = vn_linear_leaky_relu(e, P, "conv4", training)
    x2 = pool(e, "pool2")
    e = graph_feature(x2, k, knn_fn=knn_fn)
    e = vn_linear_leaky_relu(e, P, "conv5", training)
    x3 = pool(e, "pool3")
    x123 = torch.cat((x1, x2, x3), dim=1)                       # [B,63,3,N]
    aux.update(x1=x1, x2=x2, x3=x3)

    g = vn_linear_leaky_relu(x123, P, "conv6", training)         # [B,341,3,N]
    aux.update(g6=g)
    g = torch.cat((g, g.mean(dim=-1, keepdim=True).expand(g.size())), 1)
    g, z0 = vn_std_feature(g, P, "std_feature", training)
    loc = torch.einsum("bijm,bjkm->bikm", x123, z0).reshape(B, -1, N)   # [B,189,N]
    glob = g.reshape(B, -1, N).max(dim=-1, keepdim=True)[0]              # [B,2046,1]
    aux.update(z0=z0, loc=loc, glob=glob)

    cat = _conv_bn_act(l.reshape(B, -1, 1), P, "conv7", training)        # [B,64,1]
    h = torch.cat((torch.cat((glob, cat), dim=1).repeat(1, 1, N), loc), dim=1)   # [B,2299,N]
    h = _conv_bn_act(h, P, "conv8", training)
    aux.update(h8=h)
    if drop_masks is not None:
        h = h * drop_masks[0]
    h = _conv_bn_act(h, P, "conv9", training)
    if drop_masks is not None:
        h = h * drop_masks[1]
    h = _conv_bn_act(h, P, "conv10", training)
    h = _conv_bn_act(h, P, "conv11", training, act=False)
    out = h.transpose(1, 2)
    return (out, aux) if return_aux else out


# ----------------------------------------------------------------- parameters
def param_shapes(out_features, num_categories, pooling="mean"):
    """Names and shapes of VN_DGCNN_partseg's state dict (probed: SURVEY section 5)."""
    c = 64 // 3
    g = 1024 // 3
    shapes = {}

    def vn(prefix, cin, cout, shared=False):
        shapes[prefix + ".map_to_feat.weight"] = (cout, cin)
        shapes[prefix + ".batchnorm.bn.weight"] = (cout,)
        shapes[prefix + ".batchnorm.bn.bias"] = (cout,)
        shapes[prefix + ".batchnorm.bn.running_mean"] = (cout,)
        shapes[prefix + ".batchnorm.bn.running_var"] = (cout,)
        shapes[prefix + ".batchnorm.bn.num_batches_tracked"] = ()
        shapes[prefix + ".map_to_dir.weight"] = (1 if shared else cout, cin)

    def bn(prefix, ch):
        shapes[prefix + ".weight"] = (ch,)
        shapes[prefix + ".bias"] = (ch,)
        shapes[prefix + ".running_mean"] = (ch,)
        shapes[prefix + ".running_var"] = (ch,)
        shapes[prefix + ".num_batches_tracked"] = ()

    for name, ch in (("bn7", 64), ("bn8", 256), ("bn9", 256), ("bn10", 128), ("bn11", out_features)):
        bn(name, ch)
    vn("conv1", 2, c)
    vn("conv2", c, c)
    vn("conv3", 2 * c, c)
    vn("conv4", c, c)
    vn("conv5", 2 * c, c)
    if pooling == "max":
        for name in ("pool1", "pool2", "pool3"):
            shapes[name + ".map_to_dir.weight"] = (c, c)
    vn("conv6", 3 * c, g, shared=True)
    vn("std_feature.vn1", 2 * g, g)
    vn("std_feature.vn2", g, g // 2)
    shapes["std_feature.vn_lin.weight"] = (3, g // 2)
    for conv, cin, cout in (("conv8", 2299, 256), ("conv7", num_categories, 64), ("conv9", 256, 256),
                            ("conv10", 256, 128), ("conv11", 128, out_features)):
        shapes[conv + ".0.weight"] = (cout, cin, 1)
        bn(conv + ".1", cout)
    return shapes


def resolve_aliases(P):
    """``convN.1.*`` and ``bnN.*`` are the same tensors in the reference (shared module)."""
    for n in (7, 8, 9, 10, 11):
        for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            P[f"bn{n}.{s}"] = P[f"conv{n}.1.{s}"]
    return P


def numpy_cascade_sumsq(x):
    """torch.sum(x**2, dim=1) for [B,D,N] restated in numpy (documents the cascade order)."""
    sq = (x * x).astype(np.float32)
    B, D, N = sq.shape
    acc1 = np.zeros((B, N), np.float32)
    i = 0
    while i + 16 <= D:
        c = np.zeros((B, N), np.float32)
        for j in range(16):
            c = c + sq[:, i + j, :]
        acc1 = acc1 + c
        i += 16
    r = np.zeros((B, N), np.float32)
    while i < D:
        r = r + sq[:, i, :]
        i += 1
    return r + acc1
