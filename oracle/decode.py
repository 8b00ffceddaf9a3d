"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's test-time decode.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(hpcs_amd) never does.

Path restated:
  * BaseSimilarityHypHC._decode_linkage          hpcs/models/base_hyp_hc.py:81-86
        leaves = project(normalize_embeddings(x))  (fp32 torch)  ->  .cpu()  ->
        scipy.cluster.hierarchy.linkage(leaves, method='complete', metric='cosine')
  * get_optimal_k(y, linkage_matrix, 'iou')       hpcs/utils/scores.py:141-177   (+ remap_labels :124-139)
        for k in 1..K_true+4: fcluster(Z, k, 'maxclust') - 1, IoU matrix, arg-max remap, one-hot Jaccard; best k (strict >)

The arithmetic lives in third-party code that IS present in this image and on the GPU box: scipy 1.15.3
(linkage / fcluster) and scikit-learn 1.7.2 (jaccard_score).  Two layers therefore:
  * `*_scipy` functions call the real third-party routines exactly as the reference does (the pin),
  * `pdist_cosine`, `nn_chain_complete`, `label_linkage`, `fcluster_maxclust`, `optimal_k` restate the published
    algorithms in numpy / plain Python (scipy/cluster/_hierarchy.pyx: nn_chain, label, cluster_maxclust_monocrit,
    cluster_monocrit; scipy/spatial/src/distance_impl.h: cosine) -- this is what the HIP kernels follow step by step, and
    tests/test_oracle_decode.py holds the restatement bit-equal to the real routines, ties included.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------------------------
# leaves: normalize_embeddings (ultrametric_loss.py:139-143) then project (poincare.py:22-27), both fp32
# ---------------------------------------------------------------------------------------------------------------
def decode_leaves(x, scale):
    """x [N,E] float32 numpy, scale python float -> fp32 leaves exactly as torch computes them on the CPU."""
    import torch
    import torch.nn.functional as F
    xt = torch.as_tensor(np.asarray(x, dtype=np.float32))
    s = torch.tensor([scale], dtype=torch.float32)
    e = F.normalize(xt, p=2, dim=1) * s.clamp_min(1e-4).clamp_max(1.0)
    norm = e.norm(dim=-1, p=2, keepdim=True).clamp_min(1e-15)
    maxnorm = 1 - 4e-3
    proj = e / norm * maxnorm
    return torch.where(norm > maxnorm, proj, e).numpy()


def _dot2(U, V):
    """Row-wise dot products in the rounding sequence of the scipy 1.15.3 wheel in this image: its C loop
    (distance_impl.h dot_product) is auto-vectorised 2-wide, so even and odd feature indices accumulate in separate
    lanes, the lanes are added, and an odd last feature is added after that.  tests/test_oracle_decode.py pins this
    against the real pdist for even, odd and tiny E."""
    E = U.shape[-1]
    a0 = np.zeros(np.broadcast_shapes(U.shape[:-1], V.shape[:-1]))
    a1 = np.zeros_like(a0)
    for e in range(0, E - 1, 2):
        a0 = a0 + U[..., e] * V[..., e]
        a1 = a1 + U[..., e + 1] * V[..., e + 1]
    s = a0 + a1
    if E % 2:
        s = s + U[..., E - 1] * V[..., E - 1]
    return s


def pdist_cosine(X):
    """scipy pdist(X, 'cosine') restated: fp64; norms = sqrt(dot(u,u)); cos = dot(u,v) / (|u| |v|) clipped to [-1, 1];
    1 - cos.  Returns the full symmetric [n, n] matrix (diagonal 0)."""
    X = np.asarray(X, dtype=np.float64)
    nrm = np.sqrt(_dot2(X, X))
    dot = _dot2(X[:, None, :], X[None, :, :])
    with np.errstate(invalid="ignore", divide="ignore"):
        c = dot / (nrm[:, None] * nrm[None, :])
    c = np.where(np.abs(c) > 1.0, np.copysign(1.0, c), c)
    D = 1.0 - c
    np.fill_diagonal(D, 0.0)
    return D


def nn_chain_complete(D):
    """Nearest-neighbour-chain complete linkage on a full distance matrix (scipy _hierarchy.pyx nn_chain with
    new_dist = max).  Returns the UNSORTED, unlabelled merge list [(x, y, dist, size)] with x < y cluster slots."""
    D = np.array(D, dtype=np.float64, copy=True)
    n = D.shape[0]
    size = np.ones(n, dtype=np.int64)
    chain = np.zeros(n, dtype=np.int64)
    clen = 0
    out = np.zeros((n - 1, 4))
    for k in range(n - 1):
        if clen == 0:
            chain[0] = int(np.nonzero(size > 0)[0][0])
            clen = 1
        while True:
            x = int(chain[clen - 1])
            row = np.where(size > 0, D[x], np.inf)
            row[x] = np.inf
            i = int(np.argmin(row))                      # first index of the minimum
            if clen > 1:
                y = int(chain[clen - 2])
                cur = D[x, y]
                if row[i] < cur:                         # strictly smaller only: prefer the previous chain element
                    y, cur = i, row[i]
            else:
                y, cur = i, row[i]
            if clen > 1 and y == chain[clen - 2]:
                break
            chain[clen] = y
            clen += 1
        clen -= 2
        if x > y:
            x, y = y, x
        nx, ny = size[x], size[y]
        out[k] = (x, y, cur, nx + ny)
        size[x] = 0
        size[y] = nx + ny
        act = size > 0
        act[y] = False
        upd = np.maximum(D[act, x], D[act, y])
        D[act, y] = upd
        D[y, act] = upd
    return out


def label_linkage(merges):
    """Stable sort by height, then union-find relabelling (scipy _hierarchy.pyx label): rows become
    (min root, max root, height, size) with new cluster ids n + row."""
    n = merges.shape[0] + 1
    order = np.argsort(merges[:, 2], kind="stable")
    Z = merges[order].copy()
    parent = np.arange(2 * n - 1)
    sz = np.ones(2 * n - 1, dtype=np.int64)
    nxt = n

    def find(a):
        p = a
        while parent[a] != a:
            a = parent[a]
        while parent[p] != a:
            p, parent[p] = parent[p], a
        return a

    for i in range(n - 1):
        xr, yr = find(int(Z[i, 0])), find(int(Z[i, 1]))
        Z[i, 0], Z[i, 1] = (xr, yr) if xr < yr else (yr, xr)
        Z[i, 3] = sz[xr] + sz[yr]
        parent[xr] = parent[yr] = nxt
        sz[nxt] = sz[xr] + sz[yr]
        nxt += 1
    return Z


def complete_linkage(leaves):
    """= scipy linkage(leaves, 'complete', 'cosine'), restated."""
    return label_linkage(nn_chain_complete(pdist_cosine(leaves)))


def complete_linkage_scipy(leaves):
    from scipy.cluster.hierarchy import linkage
    return linkage(leaves, method="complete", metric="cosine")


def decode_linkage_scipy(x, scale):
    """The reference's _decode_linkage on one cloud (base_hyp_hc.py:81-86)."""
    return complete_linkage_scipy(decode_leaves(x, scale))


# ---------------------------------------------------------------------------------------------------------------
# fcluster(Z, k, 'maxclust') restated
# ---------------------------------------------------------------------------------------------------------------
def fcluster_maxclust(Z, k):
    """scipy fcluster(Z, k, criterion='maxclust') for a MONOTONE linkage (complete linkage is): the cut height is the
    smallest merge height leaving <= k clusters; clusters are numbered 1.. in the order cluster_monocrit meets them
    (a flat cluster when its top node is entered, a singleton leaf when its parent node is left; left before right)."""
    Z = np.asarray(Z)
    n = Z.shape[0] + 1
    h = Z[:, 2]
    if not 1 <= k < n:
        raise ValueError("1 <= k < n")         # get_optimal_k asks for at most K_true + 4 clusters
    cutoff = h[n - k - 1]                      # after n-k merges exactly k clusters remain; equal heights merge further
    T = np.zeros(n, dtype=np.int32)
    left = Z[:, 0].astype(np.int64)
    right = Z[:, 1].astype(np.int64)
    ncl = 0
    leader = -1
    stack = [2 * n - 2]
    visited = np.zeros(2 * n - 1, dtype=bool)
    while stack:
        node = stack[-1]
        r = node - n
        lc, rc = left[r], right[r]
        if leader == -1 and h[r] <= cutoff:
            leader = r
            ncl += 1
        if lc >= n and not visited[lc]:
            visited[lc] = True
            stack.append(lc)
            continue
        if rc >= n and not visited[rc]:
            visited[rc] = True
            stack.append(rc)
            continue
        if lc < n:
            if leader == -1:
                ncl += 1
            T[lc] = ncl
        if rc < n:
            if leader == -1:
                ncl += 1
            T[rc] = ncl
        if leader == r:
            leader = -1
        stack.pop()
    return T


def fcluster_maxclust_keys(Z, k):
    """The same numbering without the traversal (the form the HIP kernel evaluates in parallel): an Euler tour over the
    INTERNAL nodes gives each one an enter and an exit time; a flat cluster is created when its top node is entered
    (key 4*enter), a singleton leaf when its parent is left (key 4*exit + 1 left / + 2 right); cluster ids are the ranks
    of the distinct keys.  Because Z is sorted by height, `height <= cutoff` is `row <= rmax`."""
    Z = np.asarray(Z)
    n = Z.shape[0] + 1
    if not 1 <= k < n:
        raise ValueError("1 <= k < n")
    h = Z[:, 2]
    rmax = int(np.nonzero(h <= h[n - k - 1])[0].max())
    left = Z[:, 0].astype(np.int64)
    right = Z[:, 1].astype(np.int64)
    cnt = Z[:, 3].astype(np.int64)
    parent = np.full(2 * n - 1, -1, dtype=np.int64)
    parent[left] = np.arange(n, 2 * n - 1)
    parent[right] = np.arange(n, 2 * n - 1)
    enter = np.zeros(n - 1, dtype=np.int64)
    for r in range(n - 3, -1, -1):                       # parents have larger row numbers than their children
        p = parent[r + n] - n
        off = 1
        if right[p] == r + n and left[p] >= n:
            off += 2 * (cnt[left[p] - n] - 1)
        enter[r] = enter[p] + off
    exit_ = enter + 2 * (cnt - 1) - 1
    key = np.zeros(n, dtype=np.int64)
    for i in range(n):
        node = i
        while parent[node] >= 0 and parent[node] - n <= rmax:
            node = parent[node]
        if node == i:
            p = parent[i] - n
            key[i] = 4 * exit_[p] + (1 if left[p] == i else 2)
        else:
            key[i] = 4 * enter[node - n]
    return (np.searchsorted(np.unique(key), key) + 1).astype(np.int32)


def remap_labels(y):
    y = np.asarray(y)
    u = np.unique(y)
    return np.searchsorted(u, y)


def optimal_k(y, Z, index="iou"):
    """get_optimal_k(y, Z, index) restated on integer counts (scores.py:141-177): (best_pred, best_k, best_score).
    'iou': IoU matrix (float32, as the reference's torch.zeros matrix) -> first arg-max per true part -> remap -> Jaccard of
    the one-hot matrices = m / (2n - m), m = points whose remapped cluster equals their part.
    'ri' : sklearn adjusted_rand_score from the pair confusion matrix (integers until the final division)."""
    y = np.asarray(y).astype(np.int64)
    y_true = remap_labels(y)
    kt = int(y_true.max()) + 1
    n = y.shape[0]
    best_score, best_k, best_pred = 0.0, 0, None
    for k in range(1, kt + 5):
        y_pred = fcluster_maxclust(Z, k) - 1
        kp = len(np.unique(y_pred))
        cont = np.zeros((kt, kp), dtype=np.int64)
        np.add.at(cont, (y_true, y_pred), 1)
        ct = cont.sum(1, keepdims=True)
        cp = cont.sum(0, keepdims=True)
        if index == "iou":
            union = ct + cp - cont
            iou = (cont / union).astype(np.float32)
            ind = np.argmax(iou, axis=1)                                                  # first maximum
            y_remap = np.zeros(n, dtype=np.int64)
            for i in range(kt):
                y_remap[y_pred == ind[i]] = i + 1
            m = int(np.sum(y_remap == y_true + 1))
            score = m / (2 * n - m)
        elif index == "ri":
            ss = int((cont ** 2).sum())
            tp, fp, fn = ss - n, int((cp ** 2).sum()) - ss, int((ct ** 2).sum()) - ss
            tn = n * n - fp - fn - ss
            score = 1.0 if (fn == 0 and fp == 0) else 2.0 * (tp * tn - fn * fp) / ((tp + fn) * (fn + tn) + (tp + fp) * (fp + tn))
        else:
            raise ValueError(index)
        if score > best_score:
            best_score, best_k, best_pred = score, k, y_pred
    return best_pred, best_k, best_score


def optimal_k_thirdparty(y, Z, index="iou"):
    """The same loop written on the real third-party routines the reference calls (scipy fcluster, sklearn jaccard_score /
    adjusted_rand_score, torch.max) -- the live pin for `optimal_k` where /root/reference is absent (GPU box)."""
    import torch
    from scipy.cluster.hierarchy import fcluster
    from sklearn.metrics import jaccard_score
    from sklearn.metrics.cluster import adjusted_rand_score
    y = np.asarray(y)
    y_true = remap_labels(y)
    kt = len(np.unique(y_true))
    best_score, best_k, best_pred = 0.0, 0, None
    for k in range(1, kt + 5):
        y_pred = fcluster(Z, k, criterion="maxclust") - 1
        kp = len(np.unique(y_pred))
        if index == "ri":
            score = adjusted_rand_score(y, y_pred)
        else:
            matrix = torch.zeros(kt, kp)
            for i in range(kt):
                for j in range(kp):
                    matrix[i, j] = jaccard_score(y_true == i, y_pred == j)
            _, ind = torch.max(matrix, dim=1)
            y_remap = np.zeros_like(y_pred)
            for i in range(kt):
                y_remap[y_pred == int(ind[i])] = i + 1
            a = np.eye(kt + 1)[y_true + 1]
            b = np.eye(kt + 1)[y_remap]
            score = np.logical_and(a, b).sum() / np.logical_or(a, b).sum()
        if score > best_score:
            best_score, best_k, best_pred = score, k, y_pred
    return best_pred, best_k, best_score
