"""Oracle (TEST INFRASTRUCTURE) for the Poincare-ball math, rows a7, a11, a15.

Torch-CPU restatement; op ORDER follows the reference because several of these
expressions are cancellation-sensitive in fp32.

Reference lines restated (relative to /root/reference):
  Artanh / tanh clamp     hpcs/utils/math.py:61-82
  lambda_, expmap_1, mobius_add, project, mobius_mul, hyp_dist_o, gyration, ptransp, inner,
  egrad2rgrad             hpcs/utils/poincare.py:9-136 and hpcs/distances/poincare.py:13-136
  ExpMap / MLPExpMap      hpcs/nn/hyperbolic/hyp_embed.py:6-23
  hyp_lca + helpers       hpcs/distances/lca.py:8-52
  HyperbolicDistance      hpcs/distances/poincare.py:139-187
  HyperbolicLCA           hpcs/distances/poincare.py:190-237
"""
import torch

MIN_NORM = 1e-15
BALL_EPS = {torch.float32: 4e-3, torch.float64: 1e-5}
ARTANH_CLAMP = 1e-5


class _ArtanhFn(torch.autograd.Function):
    """artanh with input clamp to +-(1-1e-5), evaluated in fp64, grad g/(1-x^2) (clamped x)."""

    @staticmethod
    def forward(ctx, x):
        xc = x.clamp(-1 + ARTANH_CLAMP, 1 - ARTANH_CLAMP)
        ctx.save_for_backward(xc)
        xd = xc.double()
        return ((1 + xd).log() - (1 - xd).log()).mul(0.5).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return g / (1 - xc ** 2)


def artanh(x):
    return _ArtanhFn.apply(x)


def tanh_clamped(x):
    return x.clamp(-15, 15).tanh()


def conformal_factor(x):
    """lambda_(x) = 2 / max(1-|x|^2, MIN_NORM), computed on detached data."""
    n2 = torch.sum(x.detach().pow(2), dim=-1, keepdim=True)
    return 2 / (1.0 - n2).clamp_min(MIN_NORM)


def mobius_add(x, y):
    x2 = torch.sum(x * x, dim=-1, keepdim=True)
    y2 = torch.sum(y * y, dim=-1, keepdim=True)
    xy = torch.sum(x * y, dim=-1, keepdim=True)
    num = (1 + 2 * xy + y2) * x + (1 - x2) * y
    den = 1 + 2 * xy + x2 * y2
    return num / den.clamp_min(MIN_NORM)


def expmap(u, p):
    """expmap_1(u, p): exponential map of tangent u at base point p."""
    un = u.norm(dim=-1, p=2, keepdim=True).clamp_min(MIN_NORM)
    step = tanh_clamped(conformal_factor(p) * un / 2) * u / un
    return mobius_add(p, step)


def expmap0(u):
    """ExpMap.forward: expmap_1(u, zeros)."""
    return expmap(u, torch.zeros_like(u))


def mlp_expmap(x, weight, bias=None):
    """MLPExpMap.forward with dropout=0: bias-free Linear then exp-map at the origin."""
    return expmap0(torch.nn.functional.linear(x, weight, bias))


def project(x):
    n = x.norm(dim=-1, p=2, keepdim=True).clamp_min(MIN_NORM)
    maxnorm = 1 - BALL_EPS[x.dtype]
    return torch.where(n > maxnorm, x / n * maxnorm, x)


def mobius_mul(x, t):
    n = x.norm(dim=-1, p=2, keepdim=True).clamp_min(MIN_NORM)
    return tanh_clamped(t * artanh(n)) * x / n


def midpoint_o(x):
    return mobius_mul(x, 0.5)


def hyp_dist_o(x):
    return 2 * artanh(x.norm(dim=-1, p=2, keepdim=True))


def egrad2rgrad(p, dp):
    return dp / conformal_factor(p).pow(2)


def inner(x, u, v=None):
    v = u if v is None else v
    return conformal_factor(x) ** 2 * (u * v).sum(dim=-1, keepdim=True)


def gyration(u, v, w):
    u2 = u.pow(2).sum(dim=-1, keepdim=True)
    v2 = v.pow(2).sum(dim=-1, keepdim=True)
    uv = (u * v).sum(dim=-1, keepdim=True)
    uw = (u * w).sum(dim=-1, keepdim=True)
    vw = (v * w).sum(dim=-1, keepdim=True)
    a = -uw * v2 + vw + 2 * uv * vw
    b = -vw * u2 - uw
    d = 1 + 2 * uv + u2 * v2
    return w + 2 * (a * u + b * v) / d.clamp_min(MIN_NORM)


def ptransp(x, y, u):
    return gyration(y, -x, u) * conformal_factor(x) / conformal_factor(y)


# ------------------------------------------------------------------ a11 hyp_lca
def _inversion(c, x):
    """Circle inversion of x through the circle centred c orthogonal to the unit circle."""
    r2 = torch.sum(c ** 2, dim=-1, keepdim=True) - 1.0
    u = x - c
    return r2 / torch.sum(u ** 2, dim=-1, keepdim=True) * u + c


def _line_reflection(x, a):
    """Reflection of x across the line through the origin and a."""
    xa = torch.sum(x * a, dim=-1, keepdim=True)
    a2 = torch.sum(a ** 2, dim=-1, keepdim=True).clamp_min(MIN_NORM)
    return 2 * (xa * a / a2) - x


def hyp_lca(a, b, return_coord=True):
    """Projection of the origin on the geodesic a-b (or its hyperbolic distance to the origin)."""
    c = a / torch.sum(a ** 2, dim=-1, keepdim=True)
    b_inv = _inversion(c, b)
    o_ref = _inversion(c, _line_reflection(a, b_inv))
    proj = o_ref / (1.0 + torch.sqrt(1 - torch.sum(o_ref ** 2, dim=-1, keepdim=True)))
    return proj if return_coord else hyp_dist_o(proj)


def hyp_lca_dist_f64(a, b):
    """Same quantity in float64 from the closed form (ground truth for conditioning studies).

    For |a|=|b|=s and angle t between them the geodesic's closest point to the origin has
    Euclidean norm  m = (1+s^2 - sqrt((1+s^2)^2 - 4 s^2 cos^2(t/2))) / (2 s cos(t/2));
    the general case is evaluated through the same reflections in fp64.
    """
    return hyp_lca(a.double(), b.double(), return_coord=False)


# ------------------------------------------------------------------ a15
def hyperbolic_similarity_mat(x, y=None):
    """HyperbolicDistance.compute_mat: exp(-acosh(1 + 2|x-y|^2/((1-|x|^2)(1-|y|^2))))."""
    y = x if y is None else y
    x = project(x)
    y = project(y)
    d2 = torch.cdist(x, y, p=2) ** 2
    ax = 1 - torch.norm(x, p=2, dim=-1, keepdim=True) ** 2
    ay = 1 - torch.norm(y, p=2, dim=-1, keepdim=True) ** 2
    if x.dim() == 2:
        den = torch.matmul(ax, ay.T)
    elif x.dim() == 3:
        den = torch.einsum("ikj,ihj->ikh", ax, ay)
    else:
        raise ValueError("Distance implemented only for tensors of rank 2 and 3")
    return torch.exp(-torch.acosh(1 + 2 * (d2 / den)))


def hyperbolic_similarity_pairs(x, y):
    x = project(x)
    y = project(y)
    d2 = torch.nn.functional.pairwise_distance(x, y, p=2) ** 2
    ax = 1 - torch.norm(x, dim=-1, p=2) ** 2
    ay = 1 - torch.norm(y, dim=-1, p=2) ** 2
    return torch.exp(-torch.acosh(1 + 2 * (d2 / (ax * ay))))


def _complex_mobius_to_origin(z, x):
    """mobius_transf (hpcs/distances/poincare.py:81-96), E = 2 read as complex numbers: the disc isometry that sends z to 0,
    applied to x: (x - z) / (1 - conj(z) x).  z [N,2], x [M,2] -> [N,M,2]."""
    z1 = torch.view_as_complex(z.contiguous()).view(-1, 1)
    x1 = torch.view_as_complex(x.contiguous()).view(1, -1)
    return torch.view_as_real((x1 - z1) / (1 - z1.conj() * x1))


def _complex_mobius_from_origin(z, w):
    """inverse_mobius_transf (poincare.py:99-115): (w + z) / (1 + conj(z) w) with w [N,M,2] already paired with z [N,2]."""
    z1 = torch.view_as_complex(z.contiguous()).view(-1, 1)
    w1 = torch.view_as_complex(w.contiguous())
    return torch.view_as_real((w1 + z1) / (1 + z1.conj() * w1))


def hyperbolic_lca_mat(x, y=None):
    """HyperbolicLCA.compute_mat (poincare.py:197-222), E = 2: for every pair the smallest of d(o,x), d(o,y), d(o,m),
    m = hyperbolic midpoint of the geodesic x-y (map x to the origin, halve M(y) there, map back)."""
    y = x if y is None else y
    x = project(x)
    y = project(y)
    dox = hyp_dist_o(x)                                   # [N,1]
    doy = hyp_dist_o(y).T                                 # [1,M]
    mid = midpoint_o(_complex_mobius_to_origin(x, y))     # [N,M,2]
    dom = hyp_dist_o(_complex_mobius_from_origin(x, mid))[..., 0]
    return torch.min(torch.min(dox + torch.zeros_like(dom), doy + torch.zeros_like(dom)), dom)


def hyperbolic_lca_pairs(x, y):
    """HyperbolicLCA.pairwise_distance (poincare.py:225-237) for row-aligned x, y [N,2] -> [N].  (The reference takes the
    first minimum with Python's builtin min(), which only runs for N = 1; this is the element-wise reading.)"""
    full = hyperbolic_lca_mat(x, y)
    return torch.diagonal(full)
