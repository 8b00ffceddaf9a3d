"""GPU parity of the miner / loss / optimizer rows (a8-a14, RAdam) against the oracle and golden vectors."""
import numpy as np
import pytest
import torch

from _det import det_int, det_normalish, det_uniform
from oracle import hyperbolic as OH, metric as OM
from test_oracle_golden import SAMPLER_CASES

pytestmark = pytest.mark.gpu
dev = "cuda"
T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().float()
    b = (T(np.asarray(b)) if not torch.is_tensor(b) else b.detach().cpu()).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def relclose(a, b, tol):
    a = a.detach().cpu().double()
    b = (T(np.asarray(b)) if not torch.is_tensor(b) else b.detach().cpu()).double()
    err = float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert err < tol, err


@pytest.mark.parametrize("name", list(SAMPLER_CASES))
@pytest.mark.parametrize("frac", [0.0, 1.2])
def test_sampler_identical_to_reference(golden, name, frac):
    """rng='host' consumes the CPU generator like the reference -> identical (a,p,n)."""
    from hpcs_amd.miner.loss_and_miner_utils import get_balanced_random_triplet_indices
    lab = SAMPLER_CASES[name]()
    torch.manual_seed(1234)
    a, p, n = get_balanced_random_triplet_indices(lab.to(dev), t_per_anchor=5, fraction=frac)
    ref = golden[f"sampler_{name}_f{int(frac * 10)}_apn"].astype(np.int64)
    assert np.array_equal(torch.stack([a, p, n]).cpu().numpy(), ref)


def test_sampler_device_rng_properties():
    from hpcs_amd.miner.loss_and_miner_utils import get_balanced_random_triplet_indices, TripletPlan
    lab = det_int((4096,), 5, 7)
    a, p, n = get_balanced_random_triplet_indices(lab.to(dev), t_per_anchor=50, fraction=1.2, rng="device", seed=3)
    a, p, n = a.cpu(), p.cpu(), n.cpu()
    plan = TripletPlan.from_labels(lab, 50, 1.2)
    assert a.numel() == plan.T == sum(int(c) * int(k) for c, k in zip(plan.counts, plan.k))
    assert bool((lab[a] == lab[p]).all()) and bool((a != p).all()) and bool((lab[a] != lab[n]).all())
    # every anchor appears exactly k_l times; positives/negatives roughly uniform
    cnt = torch.bincount(a, minlength=4096)
    assert bool((cnt == plan.k[lab].long()).all())
    neg_hist = torch.bincount(n, minlength=4096).float()
    assert float(neg_hist.std() / neg_hist.mean()) < 0.2
    a2, p2, n2 = get_balanced_random_triplet_indices(lab.to(dev), t_per_anchor=50, fraction=1.2, rng="device", seed=4)
    assert not torch.equal(p2.cpu(), p)


@pytest.mark.parametrize("kind,margin", [("easy", 0.0), ("semihard", 0.35), ("hard", 0.1), ("all", 0.2)])
def test_miner_filter(kind, margin):
    from hpcs_amd.miner.triplet_margin_miner import RandomTripletMarginMiner
    M, E = 1500, 32
    x = OH.expmap0(det_normalish((M, E), 7) * 0.6)
    lab = det_int((M,), 8, 5)
    torch.manual_seed(7)
    wa, wp, wn = OM.mine_triplets(x, lab, 20, 1.2, margin, kind)
    torch.manual_seed(7)
    miner = RandomTripletMarginMiner(t_per_anchor=20, fraction=1.2, margin=margin, type_of_triplets=kind)
    a, p, n = miner(x.to(dev), lab.to(dev))
    got = torch.stack([a, p, n]).cpu()
    want = torch.stack([wa, wp, wn])
    if got.shape == want.shape and torch.equal(got, want):
        return
    # borderline margins may flip with the dot-product summation order: allow only such differences
    mat = OM.cosine_sim_mat(x)
    torch.manual_seed(7)
    fa, fp_, fn = OM.balanced_triplets(lab, 20, 1.2)
    tm = (mat[fa, fp_] - mat[fa, fn])
    edge = (tm - margin).abs() < 1e-6
    if kind in ("semihard", "hard"):
        edge |= tm.abs() < 1e-6
    key = lambda t: set(map(tuple, t.t().tolist()))
    diff = key(got) ^ key(want)
    border = key(torch.stack([fa[edge], fp_[edge], fn[edge]]))
    assert diff <= border and len(diff) <= 8


@pytest.mark.parametrize("s", [1e-3, 1e-2, 0.3, 0.9])   # at s=1 the reference's construction is NaN (points on the boundary)
def test_hyphc_loss_and_grads(s):
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    M, E, C = 2000, 32, 6
    x = OH.expmap0(det_normalish((M, E), 11) * 0.5)
    lab = det_int((M,), 12, C)
    scale = torch.nn.Parameter(torch.tensor([s]))
    crit = MetricHyperbolicLoss(margin=0.35, t_per_anchor=10, fraction=1.2, scale=scale, temperature=0.05,
                                num_class=C, embedding_size=E, cosface=True, miner=True).to(dev)
    scale_d = crit.scale            # registered as a parameter of the loss module (as in the reference)
    assert scale_d.is_cuda
    W = det_normalish((E, C), 13)
    crit.loss_cosface.W.data.copy_(W)
    xg = x.to(dev).requires_grad_(True)
    torch.manual_seed(5)
    out = crit.compute_loss(None, xg, lab.to(dev))
    lh, lm = out["loss_hyp"]["losses"], out["loss_metric"]["losses"]
    a, p, n, count = crit.last_triplets
    c = int(count)
    trip = (a[:c].cpu(), p[:c].cpu(), n[:c].cpu())
    # oracle on the SAME triplets, fp64 (ground truth) and fp32 reference-order
    x64 = x.double().requires_grad_(True)
    s64 = torch.tensor([s], dtype=torch.float64, requires_grad=True)
    W64 = W.double().requires_grad_(True)
    lh64 = OM.hyphc_loss(x64, trip, s64, 0.05, mat_mean="closed")
    lm64 = OM.cosface_loss(x64, lab, W64, 0.35, 2.0)
    lh32 = OM.hyphc_loss_gathered(x, trip, torch.tensor([s]), 0.05)
    assert abs(float(lh) - float(lh64)) <= 2e-6 * abs(float(lh64))
    assert abs(float(lm) - float(lm64)) <= 1e-5 * abs(float(lm64))
    # against the fp32 reference formulation (materialised matrix, reflections): north-star 1e-5
    assert abs(float(lh) - float(lh32)) <= 1e-5 * abs(float(lh32))
    (0.1 * lh + lm).backward()
    (0.1 * lh64 + lm64).backward()
    relclose(xg.grad, x64.grad, 2e-4)
    relclose(crit.loss_cosface.W.grad, W64.grad, 2e-4)
    if s < 1.0:
        assert abs(float(scale_d.grad) - float(s64.grad)) <= 1e-3 * abs(float(s64.grad)) + 1e-9
    # mined triplets equal the oracle's miner under the same seed (up to borderline flips)
    torch.manual_seed(5)
    wa, wp, wn = OM.mine_triplets(x, lab, 10, 1.2, 0, "easy")
    assert abs(wa.numel() - c) <= 8


def test_fused_filter_equals_two_pass():
    """Inline margin filter (no compaction) gives the same loss and gradients as filter-then-loss."""
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    M, E, C = 3000, 32, 5
    x = OH.expmap0(det_normalish((M, E), 41) * 0.5)
    lab = det_int((M,), 42, C)
    res = []
    for fused in (False, True):
        scale = torch.nn.Parameter(torch.tensor([0.02]))
        crit = MetricHyperbolicLoss(t_per_anchor=12, fraction=1.2, scale=scale, temperature=0.05, num_class=C,
                                    embedding_size=E, cosface=True, miner=True, fused_filter=fused).to(dev)
        xg = x.to(dev).requires_grad_(True)
        torch.manual_seed(3)
        lh = crit.compute_hyp(xg, lab.to(dev))
        lh.backward()
        res.append((float(lh), xg.grad.cpu(), float(crit.scale.grad)))
    assert abs(res[0][0] - res[1][0]) <= 5e-5 * abs(res[0][0])       # a borderline margin may flip one triplet
    relclose(res[1][1], res[0][1], 1e-3)
    assert abs(res[0][2] - res[1][2]) <= 1e-3 * abs(res[0][2])


def test_triplet_branch_loss_and_grads():
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    M, E, C = 1200, 16, 4
    x = OH.expmap0(det_normalish((M, E), 21) * 0.7)
    lab = det_int((M,), 22, C)
    scale = torch.tensor([0.05], device=dev, requires_grad=True)
    crit = MetricHyperbolicLoss(margin=0.35, t_per_anchor=8, fraction=1.2, scale=scale, temperature=0.05, num_class=C,
                                embedding_size=E, cosface=False, miner=True)
    xg = x.to(dev).requires_grad_(True)
    torch.manual_seed(9)
    out = crit.compute_loss(None, xg, lab.to(dev))
    lm = out["loss_metric"]["losses"]
    torch.manual_seed(9)
    OM.mine_triplets(x, lab, 8, 1.2, 0, "easy")               # consume the hyp miner's draws
    t2 = OM.mine_triplets(x, lab, 8, 1.2, 0.35, "semihard")
    x64 = x.double().requires_grad_(True)
    want = OM.triplet_margin_loss(x64, t2, 0.35)
    assert abs(float(lm) - float(want)) <= 1e-4 * abs(float(want))
    lm.backward()
    want.backward()
    relclose(xg.grad, x64.grad, 2e-3)
    assert not hasattr(crit, "loss_cosface")
    with pytest.raises(ValueError):
        crit.get_logits(xg, lab.to(dev))


def test_get_logits_and_cosine_matrix():
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    from hpcs_amd.distances import CosineSimilarity
    M, E, C = 300, 8, 5
    x = det_normalish((M, E), 31) * 0.4
    lab = det_int((M,), 32, C)
    crit = MetricHyperbolicLoss(num_class=C, embedding_size=E, cosface=True, miner=True).to(dev)
    W = crit.loss_cosface.W.detach().cpu()
    close(crit.get_logits(x.to(dev), lab.to(dev)), OM.cosface_logits(x, lab, W, 0.35, 2.0), rtol=1e-5, atol=1e-6)
    close(CosineSimilarity()(x.to(dev)), OM.cosine_sim_mat(x), rtol=1e-5, atol=1e-6)
    close(crit.normalize_embeddings(x.to(dev)), OM.scale_to_diameter(x, torch.tensor([1e-3])), rtol=1e-5, atol=1e-9)


def test_radam_matches_reference(golden):
    from hpcs_amd.optim import RAdam
    shapes = (((5, 7), 0.3), ((6,), 0.5), ((4, 3, 1), 0.8), ((1,), 1e-3), ((3, 4), 2.0))
    ps = [torch.nn.Parameter((det_uniform(s, 120 + i) * sc).to(dev)) for i, (s, sc) in enumerate(shapes)]
    opt = RAdam(ps, lr=0.05)
    for step in range(3):
        for i, p_ in enumerate(ps):
            p_.grad = (det_uniform(tuple(p_.shape), 130 + 10 * step + i) * 0.7).to(dev)
        opt.step()
        for i, p_ in enumerate(ps):
            close(p_, golden[f"radam_step{step}_p{i}"], rtol=2e-5, atol=1e-7)
    for i, p_ in enumerate(ps):
        close(opt.state[p_]["exp_avg"], golden[f"radam_final_m{i}"], rtol=2e-5, atol=1e-7)
        close(opt.state[p_]["exp_avg_sq"], golden[f"radam_final_v{i}"], rtol=2e-5, atol=1e-9)


def test_hierarchical_cosface_loss_and_grads():
    """Next row f2: HierarchicalMetricHyperbolicLoss's metric term vs the oracle restatement (fp64)."""
    from hpcs_amd.loss.ultrametric_loss import HierarchicalMetricHyperbolicLoss
    M, E, C = 700, 16, 9
    hier = [[[0, 1, 2], [3, 4], [5, 6, 7, 8]], [[0], [1, 2], [3], [4], [5, 6], [7, 8]], [[c] for c in range(C)]]
    x = OH.expmap0(det_normalish((M, E), 51) * 0.6)
    lab = det_int((M,), 52, C)
    crit = HierarchicalMetricHyperbolicLoss(t_per_anchor=6, fraction=1.2, scale=torch.nn.Parameter(torch.tensor([0.05])),
                                            num_class=C, embedding_size=E, miner=True, hierarchy_list=hier).to(dev)
    W = det_normalish((E, C), 53)
    crit.loss_cosface.W.data.copy_(W)
    xg = x.to(dev).requires_grad_(True)
    torch.manual_seed(2)
    out = crit.compute_loss(None, xg, lab.to(dev))
    lm = out["loss_metric"]["losses"]
    x64 = x.double().requires_grad_(True)
    W64 = W.double().requires_grad_(True)
    want = OM.hierarchical_cosface_loss(x64, lab, W64, hier, 0.35, 2.0)
    assert abs(float(lm) - float(want)) <= 1e-5 * abs(float(want))
    lm.backward()
    want.backward()
    relclose(xg.grad, x64.grad, 2e-4)
    relclose(crit.loss_cosface.W.grad, W64.grad, 2e-4)
    # the last level is the flat classification: a single flat level must equal the plain CosFace loss
    from hpcs_amd.loss.ultrametric_loss import CosFaceLoss, HierarchicalCosFaceLoss
    flat = HierarchicalCosFaceLoss(C, E, 0.35, 2, hierarchy_list=[[[c] for c in range(C)]]).to(dev)
    plain = CosFaceLoss(C, E, 0.35, 2).to(dev)
    flat.W.data.copy_(W)
    plain.W.data.copy_(W)
    assert abs(float(flat(x.to(dev), lab.to(dev))) - float(plain(x.to(dev), lab.to(dev)))) < 1e-6


@pytest.mark.parametrize("variant,E", [("4", 32), ("8", 32), ("4", 16), ("4", 64)])
def test_hyphc_sampled_in_kernel(variant, E, tuning):
    """The training-path kernel (triplets generated in-kernel, LDS positive table) against the array-based kernels
    run on the very triplets it generated, plus validity / balance of those triplets.  variant: HPCS_SAMPLED_V -- "4" = the
    r03 kernel (four triplets per group and batch, rows kept in registers; E = 64 falls back to "8"), "8" = the r02 kernel."""
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss, HypHCFn, L2RowsFn
    tuning("HPCS_SAMPLED_V", int(variant))
    M, C = 6000, 7
    x = OH.expmap0(det_normalish((M, E), 61) * 0.5)
    lab = torch.cat([det_int((M - 900,), 62, C - 1), torch.full((900,), C - 1)])        # one large class
    crit = MetricHyperbolicLoss(t_per_anchor=9, fraction=1.2, scale=torch.nn.Parameter(torch.tensor([0.03])),
                                temperature=0.05, num_class=C, embedding_size=E, cosface=True, miner=True, rng="device",
                                fused_filter=True).to(dev)
    crit.debug_triplets = []
    xg = x.to(dev).requires_grad_(True)
    lh = crit.compute_hyp(xg, lab.to(dev))
    lh.backward()
    a, p, n, _live = crit.debug_triplets
    T = a.numel()
    plan = crit.make_plan(lab)
    assert T == plan.T == crit.last_T
    la, lp, ln = lab[a.cpu()], lab[p.cpu()], lab[n.cpu()]
    assert bool((la == lp).all()) and bool((a != p).all()) and bool((la != ln).all())
    cnt = torch.bincount(a.cpu(), minlength=M)
    assert bool((cnt == plan.k[lab].long()).all())                 # every anchor exactly k_l times
    neg_hist = torch.bincount(n.cpu(), minlength=M).float()      # ~Poisson: relative spread ~ 1/sqrt(mean)
    assert float(neg_hist.std() / neg_hist.mean()) < 1.5 / float(neg_hist.mean()) ** 0.5
    # same triplets through the array-based one-pass kernel
    x2 = x.to(dev).requires_grad_(True)
    sc2 = torch.tensor([0.03], device=dev, requires_grad=True)
    cnt_t = torch.full((1,), T, dtype=torch.int64, device=dev)
    l2 = HypHCFn.apply(L2RowsFn.apply(x2), sc2, a, p, n, cnt_t, 0.05, 0.0, "easy")
    l2.backward()
    assert abs(float(lh) - float(l2)) <= 2e-6 * abs(float(l2))
    relclose(xg.grad, x2.grad, 2e-5)
    assert abs(float(crit.scale.grad) - float(sc2.grad)) <= 1e-4 * abs(float(sc2.grad))
    # no-grad call takes the gradient-free instantiation and agrees
    crit.debug_triplets = None
    crit.hyp_miner.seed -= 1
    with torch.no_grad():
        l3 = crit.compute_hyp(x.to(dev), lab.to(dev))
    assert abs(float(l3) - float(lh)) <= 2e-6 * abs(float(lh))


@pytest.mark.parametrize("M,C", [(32768, 50), (1000, 7), (5000, 600), (1, 3), (3001, 512)])
def test_class_index_is_stable_grouping(M, C):
    """order = members grouped by class, ascending index inside a class; offsets = exclusive class counts (both the
    blocked C<=512 kernels and the general path)."""
    from hpcs_amd.miner.loss_and_miner_utils import class_index
    lab = det_int((M,), 4000 + M, C)
    order, offsets = class_index(lab.to(dev), C)
    want_order = torch.argsort(lab, stable=True).to(torch.int32)
    want_off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.bincount(lab, minlength=C).cumsum(0)]).to(torch.int32)
    assert torch.equal(offsets.cpu(), want_off)
    assert torch.equal(order.cpu(), want_order)


def test_hyperbolic_lca_matches_reference_fixture():
    """a15: hpcs_amd.distances.poincare.HyperbolicLCA against the reference's outputs (tests/golden/hyperbolic_lca.npz) and
    the oracle; E != 2 is refused like torch.view_as_complex refuses it."""
    import os
    from hpcs_amd.distances.poincare import HyperbolicLCA
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hyperbolic_lca.npz"))
    x, y = torch.as_tensor(g["x"]), torch.as_tensor(g["y"])
    lca = HyperbolicLCA()
    got = lca.compute_mat(x.to(dev), y.to(dev)).cpu()
    assert float((got - torch.as_tensor(g["mat_xy"])).abs().max()) <= 2e-5
    got = lca(x.to(dev)).cpu()
    assert float((got - torch.as_tensor(g["mat_xx"])).abs().max()) <= 2e-5
    got = lca.pairwise_distance(x[:23].to(dev), y.to(dev)).cpu()
    assert got.shape == (23, 1)
    assert float((got[:, 0] - torch.as_tensor(g["pairs"])).abs().max()) <= 2e-5
    # random disc points incl. near-boundary pairs, where 1 - conj(x) y cancels in fp32: the yardstick is the oracle in
    # fp64, and the HIP result may not be further from it than a few times the reference's own fp32 evaluation is
    big = det_normalish((300, 2), 8200) * 0.6
    ref64 = OH.hyperbolic_lca_mat(OH.project(big).double())
    err_hip = float((lca(big.to(dev)).cpu().double() - ref64).abs().max())
    err_ref32 = float((OH.hyperbolic_lca_mat(big).double() - ref64).abs().max())
    assert err_hip <= max(5e-5, 3.0 * err_ref32), (err_hip, err_ref32)
    with pytest.raises(RuntimeError):
        lca(torch.zeros(4, 3, device=dev))


@pytest.mark.parametrize("M,C", [(5000, 50), (257, 4), (64, 600)])
def test_class_metrics_match_confusion_matrix(M, C):
    """hpcs_class_metrics: micro top-1 accuracy and macro IoU over the classes present in predictions or targets
    (torchmetrics' Accuracy(task='multiclass') / MulticlassJaccardIndex as the reference uses them, restated)."""
    from hpcs_amd.loss.ultrametric_loss import class_metrics
    logits = det_normalish((M, C), 9100 + C)
    logits[:7] = 0.25                                  # ties: the first maximum wins (torch.argmax)
    y = det_int((M,), 9101, max(1, C - 1))             # the last class never occurs as a target
    acc, iou = class_metrics(logits.to(dev), y.to(dev))
    pred = logits.argmax(dim=1)
    assert int(pred[:7].max()) == 0
    conf = torch.zeros(C, C, dtype=torch.float64)
    conf.index_put_((y, pred), torch.ones(M, dtype=torch.float64), accumulate=True)
    tp = conf.diag()
    denom = conf.sum(0) + conf.sum(1) - tp
    present = (conf.sum(0) + conf.sum(1)) > 0
    want_iou = float((tp[present] / denom[present]).mean())
    want_acc = float(tp.sum() / M)
    assert abs(float(acc) - want_acc) < 1e-6 and abs(float(iou) - want_iou) < 1e-6
