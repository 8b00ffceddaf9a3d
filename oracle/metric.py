"""Oracle (TEST INFRASTRUCTURE) for the miner / loss rows a8-a14 and the RAdam "next" row.

Torch-CPU restatement.  The in-tree parts follow the reference lines below; the
parts whose arithmetic lives in the absent ``pytorch-metric-learning==1.6.3``
(L2 row normalisation before compute_mat, BaseMiner's no_grad wrapper,
CosFace = LargeMarginSoftmax with cos-margin, MeanReducer, AvgNonZeroReducer,
inverted-metric margin an-ap) are restated from that library's published
algorithm -- PARITY UNPINNED for a8/a10/a12/a13/a14.

Reference lines restated (relative to /root/reference):
  CosineSimilarity                       hpcs/distances/cosine.py:4-16
  get_balanced_random_triplet_indices    hpcs/miner/loss_and_miner_utils.py:7-75
  RandomTripletMarginMiner.mine          hpcs/miner/triplet_margin_miner.py:13-38
  TripletMarginLoss.compute_loss         hpcs/miner/triplet_margin_loss.py:34-65
  MetricHyperbolicLoss                   hpcs/loss/ultrametric_loss.py:16-143
  CosFace body (verbatim mirror in-tree) hpcs/loss/hierarchical_cosface_loss.py:57-87
  hierarchical_loss                      hpcs/loss/hierarchical_cosface_loss.py:9-28   (pinned: reference_vectors_r2.npz)
  get_triplets (miner-less fallback)     hpcs/loss/ultrametric_loss.py:42-55           (pinned: reference_vectors_r2.npz)
  RAdam.step                             hpcs/optim/radam.py:59-134
"""
import torch
import torch.nn.functional as F

from . import hyperbolic as H


# --------------------------------------------------------------------------- a8
def l2_rows(x):
    """PML BaseDistance.maybe_normalize: F.normalize(x, p=2, dim=1) (eps 1e-12)."""
    return F.normalize(x, p=2, dim=1)


def cosine_sim_mat(q, r=None):
    """CosineSimilarity.__call__: 0.5*(1 + q_hat r_hat^T)."""
    qn = l2_rows(q)
    rn = qn if r is None else l2_rows(r)
    return 0.5 * (1 + torch.matmul(qn, rn.t()))


def cosine_sim_pairs(q, r):
    return 0.5 * (1 + torch.sum(l2_rows(q) * l2_rows(r), dim=1))


# --------------------------------------------------------------------------- a9
def triplets_per_anchor(counts, t_per_anchor, fraction):
    """k_l = int(t * (max_count / n_l)^fraction) evaluated like the reference: fp32 tensor math."""
    counts = torch.as_tensor(counts)
    mx = torch.max(counts)
    return [int(t_per_anchor * torch.pow((mx / c), fraction)) for c in counts]


def balanced_triplets(labels, t_per_anchor, fraction, generator=None):
    """Class-balanced random triplets, consuming the CPU RNG exactly like the reference.

    Per unique label (ascending): skip if <2 members or no negatives; k from
    ``triplets_per_anchor``; draw n_a*k positives ``randint(0, n_a-1)`` then n_a*k
    negatives ``randint(0, n_neg)``; the positive draw r maps to the r-th member
    of the class with the anchor itself removed, i.e. member r + (r >= own rank).
    """
    dev = labels.device
    labels_c = labels.detach().cpu()
    uniq = torch.unique(labels_c)
    dist = torch.bincount(labels_c)
    mx = torch.max(dist)
    A, Pn, Ng = [], [], []
    for lab in uniq:
        members = torch.where(labels_c == lab)[0]
        others = torch.where(labels_c != lab)[0]
        n_a = members.numel()
        if n_a < 2 or others.numel() < 1:
            continue
        k = int(t_per_anchor * torch.pow((mx / n_a), fraction))
        T = n_a * k
        kw = {} if generator is None else {"generator": generator}
        r_pos = torch.randint(0, n_a - 1, (T,), **kw)
        rank = torch.arange(n_a).view(-1, 1).repeat(1, k).view(T)
        r_neg = torch.randint(0, others.numel(), (T,), **kw)
        A.append(members[rank])
        Pn.append(members[r_pos + (r_pos >= rank).long()])
        Ng.append(others[r_neg])
    if not A:
        e = torch.empty(0, dtype=torch.long, device=dev)
        return e.clone(), e.clone(), e.clone()
    return torch.cat(A).to(dev), torch.cat(Pn).to(dev), torch.cat(Ng).to(dev)


# --------------------------------------------------------------------------- a10
def filter_triplets(mat, a, p, n, margin, kind):
    """RandomTripletMarginMiner's threshold filter on an inverted (similarity) metric."""
    tm = mat[a, p] - mat[a, n]
    if kind == "easy":
        keep = tm > margin
    else:
        keep = tm <= margin
        if kind == "hard":
            keep &= tm <= 0
        elif kind == "semihard":
            keep &= tm > 0
    return a[keep], p[keep], n[keep]


@torch.no_grad()
def mine_triplets(emb, labels, t_per_anchor, fraction, margin, kind, generator=None):
    a, p, n = balanced_triplets(labels, t_per_anchor, fraction, generator)
    return filter_triplets(cosine_sim_mat(emb), a, p, n, margin, kind)


# --------------------------------------------------------------------------- a13
def triplet_margin_loss(emb, triplets, margin):
    """TripletMarginLoss on the cosine similarity with AvgNonZeroReducer."""
    a, p, n = triplets
    if a.numel() == 0:
        return torch.sum(emb * 0)
    mat = cosine_sim_mat(emb)
    viol = (mat[a, n] - mat[a, p]) + margin
    loss = F.relu(viol)
    nz = loss > 0
    if nz.sum() == 0:
        return torch.sum(emb * 0)
    return loss[nz].mean()


# --------------------------------------------------------------------------- a14
def cosface_logits(emb, labels, W, margin=0.35, scale=2.0):
    """W [E,C].  cos(emb, W^T) with the target entry lowered by margin, times scale."""
    cos = torch.matmul(l2_rows(emb), l2_rows(W.t()).t())
    mask = torch.zeros_like(cos)
    mask[torch.arange(emb.shape[0]), labels] = 1
    tgt = cos[mask == 1]
    diff = ((tgt - margin) - tgt).unsqueeze(1)
    return (cos + mask * diff) * scale


def cosface_loss(emb, labels, W, margin=0.35, scale=2.0):
    return F.cross_entropy(cosface_logits(emb, labels, W, margin, scale), labels, reduction="none").mean()


def hierarchical_loss(prob, targets, hierarchy_list):
    """Per level: replace each branch's class probabilities by the branch sum, then NLL."""
    total = 0
    for level in hierarchy_list:
        summed = prob.clone()
        for branch in level:
            s = prob[:, list(branch)].sum(1, keepdim=True) if len(branch) else None
            for ch in branch:
                summed[:, ch:ch + 1] = s
        total = total + F.nll_loss(torch.log(summed), targets)
    return total


def hierarchical_cosface_loss(emb, labels, W, hierarchy_list, margin=0.35, scale=2.0):
    prob = F.softmax(cosface_logits(emb, labels, W, margin, scale), dim=1)
    return hierarchical_loss(prob, labels, hierarchy_list)


# --------------------------------------------------------------------------- a12
def scale_to_diameter(x, scale):
    """MetricHyperbolicLoss.normalize_embeddings: unit rows times clamp(scale, 1e-4, 1)."""
    return F.normalize(x, p=2, dim=1) * torch.clamp(scale, 1e-4, 1)


def hyphc_terms(x_poincare, triplets, scale, temperature):
    """The per-triplet summand of compute_hyp (ultrametric_loss.py:65-91): sum(w) - sum(w * softmax(d_lca / temperature)),
    [T]; similarities as row dots of the normalised rows (= the entries the reference gathers from its M x M matrix)."""
    a, p, n = triplets
    xn = l2_rows(x_poincare)
    wij = 0.5 * (1 + (xn[a] * xn[p]).sum(1))
    wik = 0.5 * (1 + (xn[a] * xn[n]).sum(1))
    wjk = 0.5 * (1 + (xn[p] * xn[n]).sum(1))
    e1 = scale_to_diameter(x_poincare[a], scale)
    e2 = scale_to_diameter(x_poincare[p], scale)
    e3 = scale_to_diameter(x_poincare[n], scale)
    d = torch.cat([H.hyp_lca(e1, e2, False), H.hyp_lca(e1, e3, False), H.hyp_lca(e2, e3, False)], dim=1)  # [T,3]
    w = torch.stack([wij, wik, wjk], dim=1)                                                               # [T,3]
    soft = torch.softmax(d / temperature, dim=-1)
    return w.sum(-1) - (w * soft).sum(-1)


def hyphc_loss(x_poincare, triplets, scale, temperature, mat_mean="matrix"):
    """compute_hyp given already mined triplets.

    mat_mean="matrix": mean of the materialised M x M similarity (the reference);
    mat_mean="closed": 0.5 + 0.5*|sum_i x_hat_i|^2 / M^2 (what the HIP path computes).
    """
    per_triplet = hyphc_terms(x_poincare, triplets, scale, temperature)
    xn = l2_rows(x_poincare)
    if mat_mean == "matrix":
        mm = cosine_sim_mat(x_poincare).mean()
    else:
        M = x_poincare.shape[0]
        s = xn.sum(0)
        mm = 0.5 + 0.5 * (s * s).sum() / (M * M)
    return per_triplet.mean() + mm


def hyphc_loss_gathered(x_poincare, triplets, scale, temperature):
    """compute_hyp exactly as written (ultrametric_loss.py:57-93): gathers from the M x M matrix."""
    a, p, n = triplets
    mat = cosine_sim_mat(x_poincare)
    wij, wik, wjk = mat[a, p], mat[a, n], mat[p, n]
    e1 = scale_to_diameter(x_poincare[a], scale)
    e2 = scale_to_diameter(x_poincare[p], scale)
    e3 = scale_to_diameter(x_poincare[n], scale)
    dij, dik, djk = H.hyp_lca(e1, e2, False), H.hyp_lca(e1, e3, False), H.hyp_lca(e2, e3, False)
    sim = torch.stack([wij, wik, wjk]).T                       # [T,3]
    lca = torch.stack([dij, dik, djk]).permute(2, 1, 0)        # [1,T,3] (what .T does on a 3-D tensor)
    soft = torch.softmax(lca / temperature, dim=-1)
    w_ord = torch.sum(sim * soft, dim=-1, keepdim=True)
    total = torch.sum(sim, dim=-1, keepdim=True) - w_ord
    return torch.mean(total) + mat.mean()


def all_pairs_triplets(n_samples, t_per_anchor, generator=None):
    """MetricHyperbolicLoss.get_triplets (ultrametric_loss.py:42-55), the miner-less fallback: every pair (i<j)
    repeated t_per_anchor times, one uniform third index each (ONE torch.randint of the full length on the CPU
    generator), triplets whose third index hits i or j dropped."""
    n_triplets = t_per_anchor * (n_samples * (n_samples - 1) // 2)
    ij = torch.combinations(torch.arange(n_samples), r=2).repeat_interleave(t_per_anchor, dim=0)
    kw = {} if generator is None else {"generator": generator}
    k = torch.randint(n_samples, (n_triplets,), dtype=torch.long, **kw)
    keep = (ij[:, 0] != k) & (ij[:, 1] != k)
    return ij[keep, 0], ij[keep, 1], k[keep]


def metric_hyperbolic_loss(x_poincare, labels, scale, W=None, *, margin=1.0, t_per_anchor=50, fraction=1.2,
                           temperature=0.05, cosface=True, generator=None, mat_mean="matrix"):
    """MetricHyperbolicLoss.compute_loss with miner=True. Returns (loss_hyp, loss_metric, triplets)."""
    trip = mine_triplets(x_poincare, labels, t_per_anchor, fraction, 0, "easy", generator)
    loss_hyp = hyphc_loss(x_poincare, trip, scale, temperature, mat_mean)
    if cosface:
        loss_metric = cosface_loss(x_poincare, labels.long(), W)
    else:
        trip2 = mine_triplets(x_poincare, labels, t_per_anchor, fraction, margin, "semihard", generator)
        loss_metric = triplet_margin_loss(x_poincare, trip2, margin)
    return loss_hyp, loss_metric, trip


# --------------------------------------------------------------------------- RAdam (next row f1)
def radam_step(params, grads, states, group):
    """One hpcs.optim.RAdam.step over a list of tensors (in place on ``params`` / ``states``).

    Quirks kept: group["step"] is incremented twice per parameter (radam.py:116,133);
    ``project`` (row-norm clip to 1-4e-3) is applied to every tensor (radam.py:127).
    """
    b1, b2 = group["betas"]
    eps, lr, wd, amsgrad = group["eps"], group["lr"], group["weight_decay"], group["amsgrad"]
    with torch.no_grad():
        for p, g, st in zip(params, grads, states):
            if g is None:
                continue
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
                if amsgrad:
                    st["max_exp_avg_sq"] = torch.zeros_like(p)
            g = g + wd * p
            g = H.egrad2rgrad(p, g)
            st["exp_avg"].mul_(b1).add_(g, alpha=1 - b1)
            st["exp_avg_sq"].mul_(b2).add_(H.inner(p, g), alpha=1 - b2)
            if amsgrad:
                torch.max(st["max_exp_avg_sq"], st["exp_avg_sq"], out=st["max_exp_avg_sq"])
                denom = st["max_exp_avg_sq"].sqrt().add_(eps)
            else:
                denom = st["exp_avg_sq"].sqrt().add_(eps)
            group["step"] += 1
            bc1 = 1 - b1 ** group["step"]
            bc2 = 1 - b2 ** group["step"]
            step_size = lr * bc2 ** 0.5 / bc1
            direction = st["exp_avg"] / denom
            new_p = H.project(H.expmap(-step_size * direction, p))
            st["exp_avg"].copy_(H.ptransp(p, new_p, st["exp_avg"]))
            p.copy_(new_p)
            group["step"] += 1
