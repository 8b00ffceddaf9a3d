"""Mirror of hpcs/distances/poincare.py: Poincare toolbox + HyperbolicDistance (a15)."""
import torch

from .. import _lib
from ..utils.math import arctanh, tanh  # noqa: F401
from ..utils.poincare import (BALL_EPS, MIN_NORM, expmap_1 as expmap, hyp_dist_o, lambda_, mobius_add,  # noqa: F401
                              project)


def _no_pml_kwargs(kwargs):
    """The reference forwards **kwargs to PML's BaseDistance (normalize_embeddings, p, power, is_inverted); the mirror's
    distances fix the values the reference sets itself (poincare.py:140-142, 191-193) and reject anything else."""
    allowed = {"normalize_embeddings": False, "is_inverted": False, "p": 2, "power": 1}
    for k, v in kwargs.items():
        if k not in allowed or v != allowed[k]:
            raise TypeError(f"unsupported argument {k}={v!r}")


def _dot(u, v):
    return (u * v).sum(dim=-1, keepdim=True)


def egrad2rgrad(p, dp):
    """Euclidean -> Riemannian gradient (in place, like the reference): dp / lambda_p^2."""
    return dp.div_(lambda_(p).square())


def inner(x, u, v=None):
    """Riemannian inner product of tangent vectors at x."""
    return lambda_(x).square() * _dot(u, u if v is None else v)


def gyration(u, v, w):
    """gyr[u,v]w on the Poincare ball."""
    uu, vv, uv, uw, vw = _dot(u, u), _dot(v, v), _dot(u, v), _dot(u, w), _dot(v, w)
    cu = vw * (1 + 2 * uv) - uw * vv
    cv = -(vw * uu + uw)
    return w + 2 * (cu * u + cv * v) / (1 + 2 * uv + uu * vv).clamp_min(MIN_NORM)


def ptransp(x, y, u):
    """Parallel transport of u from x to y."""
    return gyration(y, -x, u) * (lambda_(x) / lambda_(y))


def mobius_mul(x, t):
    n = x.norm(dim=-1, p=2, keepdim=True).clamp_min(MIN_NORM)
    return tanh(t * arctanh(n)) * (x / n)


def get_midpoint_o(x):
    return mobius_mul(x, 0.5)


class HyperbolicDistance:
    """exp(-acosh(1 + 2|x-y|^2/((1-|x|^2)(1-|y|^2)))) on projected points (hpcs/distances/poincare.py:139-187)."""
    is_inverted = False
    normalize_embeddings = False
    p = 2

    def __init__(self, **kwargs):
        _no_pml_kwargs(kwargs)

    def __call__(self, query_emb, ref_emb=None):
        return self.compute_mat(query_emb, ref_emb)

    def compute_mat(self, query_emb, ref_emb=None):
        if ref_emb is None:
            ref_emb = query_emb
        if query_emb.dim() == 3:
            return torch.stack([self.compute_mat(q, r) for q, r in zip(query_emb, ref_emb)])
        if query_emb.dim() != 2:
            raise ValueError("Distance implemented only for tensors of rank 2 and 3")
        x, y = query_emb.detach().contiguous().float(), ref_emb.detach().contiguous().float()
        out = torch.empty(x.shape[0], y.shape[0], device=x.device, dtype=torch.float32)
        _lib.call("hpcs_hyperbolic_sim_mat", x, y, x.shape[0], y.shape[0], x.shape[1], out)
        return out

    def pairwise_distance(self, query_emb, ref_emb):
        """Row-aligned pairs [..., E] -> [...] (the same projected-point formula as compute_mat, one thread per pair)."""
        if query_emb.shape != ref_emb.shape:
            raise RuntimeError("pairwise_distance needs tensors of the same shape")
        E = query_emb.shape[-1]
        x = query_emb.detach().reshape(-1, E).contiguous().float()
        y = ref_emb.detach().reshape(-1, E).contiguous().float()
        if not x.is_cuda:
            raise _lib.HpcsNativeError("hpcs_amd operators need HIP-device tensors (there is no CPU fallback)")
        out = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        _lib.call("hpcs_hyperbolic_sim_pairs", x, y, x.shape[0], E, out)
        return out.view(query_emb.shape[:-1])


class HyperbolicLCA:
    """min(d(o,x), d(o,y), d(o, midpoint of the geodesic x-y)) on projected points of the Poincare DISC (E = 2; the
    reference works on torch.view_as_complex) -- hpcs/distances/poincare.py:190-237.  Never instantiated by the
    reference's callers; part of the hpcs.distances API surface."""
    is_inverted = False
    normalize_embeddings = False
    p = 2

    def __init__(self, **kwargs):
        _no_pml_kwargs(kwargs)

    def __call__(self, query_emb, ref_emb=None):
        return self.compute_mat(query_emb, ref_emb)

    @staticmethod
    def _check(t):
        if t.dim() != 2 or t.shape[-1] != 2:
            raise RuntimeError("HyperbolicLCA works on [N, 2] embeddings (view_as_complex needs a last dimension of 2)")
        return t.detach().contiguous().float()

    def compute_mat(self, query_emb, ref_emb=None):
        x = self._check(query_emb)
        y = x if ref_emb is None else self._check(ref_emb)
        out = torch.empty(x.shape[0], y.shape[0], device=x.device, dtype=torch.float32)
        _lib.call("hpcs_hyperbolic_lca", x, y, x.shape[0], y.shape[0], 2, 0, out)
        return out

    def pairwise_distance(self, query_emb, ref_emb):
        """Row-aligned pairs -> [N, 1] like the reference's hyp_dist_o-shaped result.  (The reference's own code takes the
        first minimum with Python's builtin min(), so it only runs for N = 1; this is the element-wise reading.)"""
        x, y = self._check(query_emb), self._check(ref_emb)
        if x.shape[0] != y.shape[0]:
            raise RuntimeError("pairwise_distance needs the same number of rows")
        out = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        _lib.call("hpcs_hyperbolic_lca", x, y, x.shape[0], y.shape[0], 2, 1, out)
        return out.unsqueeze(-1)
