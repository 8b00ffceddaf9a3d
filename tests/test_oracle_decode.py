"""CPU: the decode oracle (oracle/decode.py, row f4) against the golden vectors made by the reference side
(tests/golden/make_golden_decode.py) and against the live third-party routines (scipy / scikit-learn are in the image)."""
import os
import sys

import numpy as np
import pytest

from oracle import decode as OD

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "decode.npz"))


def _inputs(name):
    import make_golden_decode as MGD
    return MGD.inputs(name)


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_linkage_matches_reference_side(gold, name):
    x, y, scale = _inputs(name)
    leaves = OD.decode_leaves(x.numpy(), scale)
    assert np.array_equal(leaves, gold[f"{name}_leaves"])                 # fp32 normalise + project, bit for bit on CPU
    Z = OD.complete_linkage(leaves)
    assert np.array_equal(Z, gold[f"{name}_Z"])                           # heights, pairs, sizes, order: identical
    assert np.array_equal(OD.decode_linkage_scipy(x.numpy(), scale), gold[f"{name}_Z"])


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("index", ["iou", "ri"])
def test_optimal_k_matches_reference(gold, name, index):
    x, y, scale = _inputs(name)
    pred, k, score = OD.optimal_k(y.numpy(), gold[f"{name}_Z"], index)
    assert k == int(gold[f"{name}_{index}_k"])
    assert score == float(gold[f"{name}_{index}_score"])                  # same integers, same final division
    assert np.array_equal(pred, gold[f"{name}_{index}_pred"])


@pytest.mark.parametrize("n,E,dup", [(64, 32, 0), (200, 32, 7), (300, 2, 5), (257, 7, 0), (130, 33, 3), (50, 1, 0), (33, 3, 0)])
def test_restatement_equals_scipy(n, E, dup):
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import pdist, squareform
    rng = np.random.default_rng(n * 31 + E)
    x = rng.standard_normal((n, E)).astype(np.float32)
    if dup:
        x[::dup] = x[0]
        x[1::dup] = -x[0]
    leaves = OD.decode_leaves(x, 1e-3)
    assert np.array_equal(OD.pdist_cosine(leaves), squareform(pdist(leaves.astype(np.float64), "cosine")))
    Zs = linkage(leaves, method="complete", metric="cosine")
    assert np.array_equal(OD.complete_linkage(leaves), Zs)
    for k in list(range(1, 12)) + [n // 2, n - 1]:
        want = fcluster(Zs, k, criterion="maxclust")
        assert np.array_equal(OD.fcluster_maxclust(Zs, k), want)
        assert np.array_equal(OD.fcluster_maxclust_keys(Zs, k), want)     # the parallel form the kernel evaluates
    y = rng.integers(0, 5, n) * 3
    for index in ("iou", "ri"):
        a, b = OD.optimal_k(y, Zs, index), OD.optimal_k_thirdparty(y, Zs, index)
        assert a[1] == b[1] and a[2] == b[2] and np.array_equal(a[0], b[0])
