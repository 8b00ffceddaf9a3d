"""Pins the CPU oracle against outputs of the reference itself (tests/golden/reference_vectors.npz).

CPU-only.  Rows a1-a7, a9, a11, Poincare helpers, RAdam.  Rows a8/a10/a12-a14 have no
reference vectors (their arithmetic is in the absent pytorch-metric-learning) and are
covered by analytic identities in test_oracle_identities.py -- "parity unpinned".
"""
import numpy as np
import pytest
import torch

from _det import det_uniform, det_normalish, det_int, det_cloud, det_params
from oracle import extractor as OE, hyperbolic as OH, metric as OM

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach() if torch.is_tensor(a) else torch.as_tensor(a)
    b = T(np.asarray(b)) if not torch.is_tensor(b) else b
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a.float(), b.float(), rtol=rtol, atol=atol)


def knn_tie_aware_equal(idx, ref_idx, scores):
    """Indices equal, or they differ only inside groups of exactly equal reference scores."""
    idx = np.asarray(idx)
    ref_idx = np.asarray(ref_idx)
    if np.array_equal(idx, ref_idx):
        return True
    B, N, k = idx.shape
    for b in range(B):
        for i in range(N):
            if np.array_equal(idx[b, i], ref_idx[b, i]):
                continue
            s_a = scores[b, i, idx[b, i]]
            s_r = scores[b, i, ref_idx[b, i]]
            if not np.array_equal(s_a, s_r):           # same sorted score sequence required
                return False
            # outside the last tie group the SETS per score value must agree
            last = s_r[-1]
            if set(idx[b, i][s_a != last]) != set(ref_idx[b, i][s_r != last]):
                return False
    return True


KNN_CASES = {"d3": (lambda: det_cloud(2, 256, 11), 10), "d63": (lambda: det_normalish((2, 63, 256), 12), 10),
             "dup": (lambda: det_cloud(2, 128, 13, dup_every=5), 8), "d3_1024": (lambda: det_cloud(1, 1024, 14), 20)}


@pytest.mark.parametrize("tag", list(KNN_CASES))
def test_knn_exact_matches_reference(golden, tag):
    make, k = KNN_CASES[tag]
    x = make()
    idx, dist = OE.knn_exact(x, k, return_dist=True)
    # winners' scores are bit-identical to the reference's top-k values
    assert np.array_equal(dist.numpy(), golden[f"knn_{tag}_topv"])
    scores = OE.knn_scores(x).numpy()
    assert knn_tie_aware_equal(idx.numpy(), golden[f"knn_{tag}_idx"].astype(np.int64), scores)
    if tag != "dup":
        assert np.array_equal(idx.numpy(), golden[f"knn_{tag}_idx"].astype(np.int64))


def test_knn_formula_and_cascade(golden):
    x = det_normalish((2, 63, 256), 12)
    assert np.array_equal(OE.knn_formula(x, 10).numpy(), golden["knn_d63_idx"].astype(np.int64))
    assert np.array_equal(OE.numpy_cascade_sumsq(x.numpy()), torch.sum(x ** 2, dim=1).numpy())


@pytest.mark.parametrize("tag,C", [("c1", 1), ("c21", 21)])
def test_graph_feature(golden, tag, C):
    x = det_normalish((2, C, 3, 32), 20 + C)
    close(OE.graph_feature(x, 4), golden[f"ggf_{tag}"], rtol=0, atol=0)
    idx = det_int((2, 32, 4), 23, 32)
    close(OE.graph_feature(x, 4, idx=idx), golden[f"ggf_{tag}_given_idx"], rtol=0, atol=0)


def _vn_params(prefix, cin, cout, seed, shared=False):
    shapes = {"map_to_feat.weight": (cout, cin), "batchnorm.bn.weight": (cout,), "batchnorm.bn.bias": (cout,),
              "batchnorm.bn.running_mean": (cout,), "batchnorm.bn.running_var": (cout,),
              "batchnorm.bn.num_batches_tracked": (), "map_to_dir.weight": (1 if shared else cout, cin)}
    return {prefix + "." + k: v for k, v in det_params(shapes, seed).items()}


def test_vn_linear_leaky_relu_dim5(golden):
    x5 = det_normalish((2, 6, 3, 16, 4), 30)
    P = _vn_params("m", 6, 5, 31)
    for k in ("m.map_to_feat.weight", "m.map_to_dir.weight", "m.batchnorm.bn.weight", "m.batchnorm.bn.bias"):
        P[k].requires_grad_(True)
    xg = x5.clone().requires_grad_(True)
    y = OE.vn_linear_leaky_relu(xg, P, "m", training=True)
    close(y, golden["vnllr5_train_out"])
    (y * det_uniform(tuple(y.shape), 32)).sum().backward()
    close(xg.grad, golden["vnllr5_train_gx"], rtol=1e-4, atol=1e-6)
    for k in ("map_to_feat.weight", "map_to_dir.weight", "batchnorm.bn.weight", "batchnorm.bn.bias"):
        close(P["m." + k].grad, golden["vnllr5_train_g_" + k], rtol=1e-4, atol=1e-5)
    close(P["m.batchnorm.bn.running_mean"], golden["vnllr5_train_rm"])
    close(P["m.batchnorm.bn.running_var"], golden["vnllr5_train_rv"])
    with torch.no_grad():
        close(OE.vn_linear_leaky_relu(x5, P, "m", training=False), golden["vnllr5_eval_out"])


def test_vn_linear_leaky_relu_dim4_shared(golden):
    x4 = det_normalish((2, 7, 3, 24), 33)
    P = _vn_params("m", 7, 9, 34, shared=True)
    for k in ("m.map_to_feat.weight", "m.map_to_dir.weight", "m.batchnorm.bn.weight", "m.batchnorm.bn.bias"):
        P[k].requires_grad_(True)
    xg = x4.clone().requires_grad_(True)
    y = OE.vn_linear_leaky_relu(xg, P, "m", training=True)
    close(y, golden["vnllr4s_train_out"])
    (y * det_uniform(tuple(y.shape), 35)).sum().backward()
    close(xg.grad, golden["vnllr4s_train_gx"], rtol=1e-4, atol=1e-6)
    for k in ("map_to_feat.weight", "map_to_dir.weight", "batchnorm.bn.weight", "batchnorm.bn.bias"):
        close(P["m." + k].grad, golden["vnllr4s_train_g_" + k], rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        close(OE.vn_linear_leaky_relu(x4, P, "m", training=False), golden["vnllr4s_eval_out"])


def test_vn_batchnorm_pool_std(golden):
    x5 = det_normalish((2, 6, 3, 16, 4), 30)
    bn = det_params({"bn.weight": (6,), "bn.bias": (6,), "bn.running_mean": (6,), "bn.running_var": (6,),
                     "bn.num_batches_tracked": ()}, 36)
    y = OE.vn_batchnorm(x5, bn["bn.weight"], bn["bn.bias"], bn["bn.running_mean"], bn["bn.running_var"], True)
    close(y, golden["vnbn5_train_out"])
    close(bn["bn.running_mean"], golden["vnbn5_train_rm"])
    close(bn["bn.running_var"], golden["vnbn5_train_rv"])
    y = OE.vn_batchnorm(x5, bn["bn.weight"], bn["bn.bias"], bn["bn.running_mean"], bn["bn.running_var"], False)
    close(y, golden["vnbn5_eval_out"])
    w = det_params({"map_to_dir.weight": (6, 6)}, 37)["map_to_dir.weight"]
    close(OE.vn_max_pool(x5, w), golden["vnmaxpool_out"], rtol=0, atol=0)
    close(OE.mean_pool(x5), golden["meanpool_out"])

    x8 = det_normalish((2, 8, 3, 24), 38)
    shapes = {}
    for pre, (ci, co) in (("s.vn1", (8, 4)), ("s.vn2", (4, 2))):
        shapes.update({pre + ".map_to_feat.weight": (co, ci), pre + ".batchnorm.bn.weight": (co,),
                       pre + ".batchnorm.bn.bias": (co,), pre + ".batchnorm.bn.running_mean": (co,),
                       pre + ".batchnorm.bn.running_var": (co,), pre + ".batchnorm.bn.num_batches_tracked": (),
                       pre + ".map_to_dir.weight": (co, ci)})
    shapes["s.vn_lin.weight"] = (3, 2)
    # the reference module's own key order decides which seed each tensor got: mirror it
    ref_order = {k[2:]: v for k, v in shapes.items()}
    P = {"s." + k: v for k, v in det_params(ref_order, 39).items()}
    xs, z0 = OE.vn_std_feature(x8, P, "s", training=True)
    close(xs, golden["vnstd_train_xstd"], rtol=1e-4, atol=1e-5)
    close(z0, golden["vnstd_train_z0"], rtol=1e-4, atol=1e-5)
    xs, z0 = OE.vn_std_feature(x8, P, "s", training=False)
    close(xs, golden["vnstd_eval_xstd"], rtol=1e-4, atol=1e-5)
    close(z0, golden["vnstd_eval_z0"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("E", [2, 32])
def test_full_extractor(golden, E):
    P = OE.resolve_aliases(det_params(OE.param_shapes(E, 16), 40 + E))
    x = det_cloud(4, 256, 50 + E)
    lab = torch.eye(16)[det_int((4,), 51, 16)]
    with torch.no_grad():
        y = OE.vn_dgcnn_partseg(x, lab, P, k=10, training=False)
    close(y, golden[f"net_E{E}_eval_out"], rtol=1e-4, atol=1e-5)
    learn = [k for k, v in P.items() if v.is_floating_point() and "running" not in k and not k.startswith("bn")]
    for k in learn:
        P[k].requires_grad_(True)
    OE.resolve_aliases(P)
    xg = x.clone().requires_grad_(True)
    y = OE.vn_dgcnn_partseg(xg, lab, P, k=10, training=True)
    ref_y = T(golden[f"net_E{E}_train_out"])      # 13 train-mode BNs amplify fp32 summation-order noise
    close(y, ref_y, rtol=1e-4, atol=3e-5 * float(ref_y.abs().max()))
    (y * det_uniform(tuple(y.shape), 52)).sum().backward()
    gx_ref = T(golden[f"net_E{E}_train_gx"])
    # d/dx crosses VN-LeakyReLU mask boundaries (dot >= 0): single points may flip, so compare in L2
    assert float((xg.grad - gx_ref).norm() / gx_ref.norm()) < 1e-3
    checked = 0
    for k in learn:
        g = P[k].grad
        ref_names = [k] + ([k.replace("conv", "bn").replace(".1.", ".")] if ".1." in k else [])
        for rn in ref_names:
            if f"net_E{E}_train_g_{rn}" in golden:
                ref = T(golden[f"net_E{E}_train_g_{rn}"])
                close(g, ref, rtol=2e-3, atol=2e-5 * float(ref.abs().max() + 1))
                checked += 1
            elif f"net_E{E}_train_gs_{rn}" in golden:
                ref = T(golden[f"net_E{E}_train_gs_{rn}"])
                close(g.flatten()[::97], ref, rtol=2e-3, atol=2e-5 * float(ref.abs().max() + 1))
                checked += 1
    assert checked >= 40
    for name in ("conv1.batchnorm.bn.running_mean", "conv5.batchnorm.bn.running_var", "bn11.running_var"):
        close(P[name], golden[f"net_E{E}_train_buf_{name}"], rtol=1e-4, atol=1e-6)


def test_embedders(golden):
    u = det_normalish((6, 5, 32), 60)
    u[0, 0] = 0.0
    u[0, 1] *= 1e-9
    u[1, 0] *= 40.0
    ug = u.clone().requires_grad_(True)
    y = OH.expmap0(ug)
    close(y, golden["expmap_out"])
    (y * det_uniform(tuple(y.shape), 61)).sum().backward()
    close(ug.grad, golden["expmap_gu"], rtol=1e-4, atol=1e-6)
    w = (det_uniform((8, 32), 62) * 0.3).requires_grad_(True)
    ug = u.clone().requires_grad_(True)
    y = OH.mlp_expmap(ug, w)
    close(y, golden["mlpexpmap_out"])
    (y * det_uniform(tuple(y.shape), 63)).sum().backward()
    close(ug.grad, golden["mlpexpmap_gu"], rtol=1e-4, atol=1e-6)
    close(w.grad, golden["mlpexpmap_gw"], rtol=1e-4, atol=1e-5)


def test_poincare_helpers(golden):
    a = det_normalish((64, 8), 70) * 0.3
    b = det_normalish((64, 8), 71) * 0.3
    a[0] = 0.0
    big = det_normalish((16, 8), 72) * 3.0
    close(OH.mobius_add(a, b), golden["mobius_add_out"])
    close(OH.project(torch.cat([a, big])), golden["project_out"])
    close(OH.expmap(b, a), golden["expmap_pu_out"])
    close(OH.conformal_factor(a), golden["lambda_out"])
    close(OH.hyp_dist_o(torch.cat([a, big * 0.33])), golden["hyp_dist_o_out"])
    close(OH.mobius_mul(a + 1e-3, 0.5), golden["mobius_mul_out"])
    t = torch.tensor([-2.0, -1.0, -0.999995, -0.5, 0.0, 1e-8, 0.5, 0.99999, 0.999999, 1.0, 3.0])
    tg = t.clone().requires_grad_(True)
    y = OH.artanh(tg)
    close(y, golden["artanh_out"], rtol=1e-6, atol=0)
    y.sum().backward()
    close(tg.grad, golden["artanh_grad"], rtol=1e-6, atol=0)
    w = det_normalish((64, 8), 73) * 0.2
    close(OH.gyration(a, b, w), golden["gyration_out"])
    close(OH.ptransp(a, b, w), golden["ptransp_out"])
    close(OH.inner(a, w), golden["inner_out"])
    close(OH.egrad2rgrad(a, w.clone()), golden["egrad2rgrad_out"])


@pytest.mark.parametrize("si,s", list(enumerate((1e-4, 1e-3, 1e-2, 1e-1, 0.5, 0.9))))
def test_hyp_lca(golden, si, s):
    da = torch.nn.functional.normalize(det_normalish((48, 32), 80 + si), dim=1)
    db = torch.nn.functional.normalize(det_normalish((48, 32), 90 + si), dim=1)
    sc = torch.tensor([s], requires_grad=True)
    ag = da.clone().requires_grad_(True)
    bg = db.clone().requires_grad_(True)
    d = OH.hyp_lca(ag * sc, bg * sc, return_coord=False)
    # same torch ops in the same order on the same machine: bit-identical
    close(d, golden[f"hyp_lca_s{si}_dist"], rtol=0, atol=0)
    close(OH.hyp_lca(da * s, db * s, return_coord=True), golden[f"hyp_lca_s{si}_coord"], rtol=0, atol=0)
    d.sum().backward()
    close(ag.grad, golden[f"hyp_lca_s{si}_ga"], rtol=1e-5, atol=1e-7)
    close(bg.grad, golden[f"hyp_lca_s{si}_gb"], rtol=1e-5, atol=1e-7)
    close(sc.grad, golden[f"hyp_lca_s{si}_gs"], rtol=1e-5, atol=1e-7)


def test_hyp_lca_general(golden):
    pa = det_normalish((48, 4), 100) * 0.35
    pb = det_normalish((48, 4), 101) * 0.35
    close(OH.hyp_lca(pa, pb, return_coord=False), golden["hyp_lca_general_dist"], rtol=0, atol=0)


SAMPLER_CASES = {
    "balanced": lambda: det_int((64,), 110, 4),
    "imbalanced": lambda: torch.cat([torch.zeros(40), torch.ones(14), torch.full((6,), 2)]).long()[
        det_int((60,), 111, 60).argsort()],
    "singleton": lambda: torch.cat([torch.zeros(20), torch.ones(19), torch.full((1,), 5)]).long(),
    "single_class": lambda: torch.zeros(16).long(),
    "gappy": lambda: torch.cat([torch.full((10,), 3), torch.full((12,), 7), torch.full((9,), 1)]).long(),
}


@pytest.mark.parametrize("name", list(SAMPLER_CASES))
@pytest.mark.parametrize("frac", [0.0, 1.2])
def test_balanced_sampler(golden, name, frac):
    lab = SAMPLER_CASES[name]()
    assert np.array_equal(lab.numpy(), golden[f"sampler_{name}_f{int(frac * 10)}_labels"])
    torch.manual_seed(1234)
    a, p, n = OM.balanced_triplets(lab, 5, frac)
    ref = golden[f"sampler_{name}_f{int(frac * 10)}_apn"].astype(np.int64)
    assert np.array_equal(torch.stack([a, p, n]).numpy(), ref)


def test_radam(golden):
    shapes = (((5, 7), 0.3), ((6,), 0.5), ((4, 3, 1), 0.8), ((1,), 1e-3), ((3, 4), 2.0))
    ps = [det_uniform(s, 120 + i) * sc for i, (s, sc) in enumerate(shapes)]
    states = [{} for _ in ps]
    group = {"betas": (0.9, 0.999), "eps": 1e-8, "lr": 0.05, "weight_decay": 0, "amsgrad": False, "step": 0}
    for step in range(3):
        grads = [det_uniform(tuple(p.shape), 130 + 10 * step + i) * 0.7 for i, p in enumerate(ps)]
        OM.radam_step(ps, grads, states, group)
        for i, p in enumerate(ps):
            close(p, golden[f"radam_step{step}_p{i}"], rtol=1e-5, atol=1e-7)
    for i in range(len(ps)):
        close(states[i]["exp_avg"], golden[f"radam_final_m{i}"], rtol=1e-5, atol=1e-7)
        close(states[i]["exp_avg_sq"], golden[f"radam_final_v{i}"], rtol=1e-5, atol=1e-9)


def test_hyperbolic_lca_oracle_matches_reference():
    """a15: HyperbolicLCA (never instantiated by the reference's callers; API surface).  Fixture: tests/golden/
    hyperbolic_lca.npz, produced by tests/golden/make_golden_lca.py from the reference's own class."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hyperbolic_lca.npz"))
    x, y = torch.as_tensor(g["x"]), torch.as_tensor(g["y"])
    assert float((OH.hyperbolic_lca_mat(x, y) - torch.as_tensor(g["mat_xy"])).abs().max()) <= 1e-6
    assert float((OH.hyperbolic_lca_mat(x) - torch.as_tensor(g["mat_xx"])).abs().max()) <= 1e-6
    assert float((OH.hyperbolic_lca_pairs(x[:23], y) - torch.as_tensor(g["pairs"])).abs().max()) <= 1e-6
    m = OH.hyperbolic_lca_mat(x, y)
    assert float(m.min()) >= 0.0 and float(m[0].abs().max()) == 0.0        # LCA with the origin is the origin
