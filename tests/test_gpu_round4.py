"""Round 4 GPU parity tests.

1. The kernel the benchmark runs for rows a10 + a12 -- hpcs_hyphc_sampled: class-balanced triplets generated in-kernel,
   the miner's margin rule inline, HypHC loss + gradient in one sweep -- DIRECTLY against the fp64 oracle
   (oracle/metric.py: ultrametric_loss.py:57-93, triplet_margin_miner.py:13-38, loss_and_miner_utils.py:33-61 restated),
   not through another HIP kernel: kept set (tie-aware), loss, gradients, and the distribution of the sampled triplets.
2. The embedding sizes the reference shipped (checkpoints/*/config.yaml: 50, 4, 15) on every loss path.
3. BASELINE configs[1] (B = 32, N = 1024, k = 20, C = 50, 16 categories, flat CosFace) as one full step.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _det import det_cloud, det_int, det_normalish, det_params
from oracle import extractor as OE, hyperbolic as OH, metric as OM

pytestmark = pytest.mark.gpu
dev = "cuda"


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def bench_labels():
    """The benchmark's part labels: Voronoi cells of 50 centroids over 32 clouds of 1024 points (imbalanced on purpose)."""
    import bench
    return bench.synthetic_batch(32, 1024, 50, 16, seed=0)[2].reshape(-1)


def one_big_class(M, C, big, seed):
    return torch.cat([det_int((M - big,), seed, C - 1), torch.full((big,), C - 1)])


def oracle_hyphc(x, kept, scale, temperature, device, chunk=1 << 20):
    """OM.hyphc_loss (fp64, closed-form matrix mean) and its gradients with respect to x and scale, the triplet sum taken in
    chunks.  device="cuda": the oracle's torch code evaluated by torch in fp64 on the device (torch's own kernels, none of
    the library's), for triplet counts the host would need minutes for."""
    x64 = x.detach().to(device=device, dtype=torch.float64).requires_grad_(True)
    s64 = torch.tensor([float(scale)], dtype=torch.float64, device=device, requires_grad=True)
    a, p, n = (t.to(device) for t in kept)
    K = a.numel()
    total = 0.0
    for i in range(0, K, chunk):
        part = OM.hyphc_terms(x64, (a[i:i + chunk], p[i:i + chunk], n[i:i + chunk]), s64, temperature).sum() / K
        part.backward()
        total += float(part.detach())
    M = x64.shape[0]
    xn = OM.l2_rows(x64)
    sm = xn.sum(0)
    mm = 0.5 + 0.5 * (sm * sm).sum() / (M * M)
    mm.backward()
    return total + float(mm), x64.grad.cpu(), float(s64.grad)


def chi2_band(stat, dof, sigmas=6.0):
    return abs(stat - dof) <= sigmas * math.sqrt(2.0 * dof)


def check_sampled_against_oracle(E, M, lab, t_per_anchor, scale, oracle_device, gathered32, variant=None, tuning=None,
                                 chi2=False):
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    if variant is not None:
        tuning("HPCS_SAMPLED_V", int(variant))
    C = int(lab.max()) + 1
    x = OH.expmap0(det_normalish((M, E), 61 + E) * 0.5)
    crit = MetricHyperbolicLoss(t_per_anchor=t_per_anchor, fraction=1.2, scale=torch.nn.Parameter(torch.tensor([scale])),
                                temperature=0.05, num_class=C, embedding_size=E, cosface=True, miner=True, rng="device",
                                fused_filter=True).to(dev)
    crit.debug_triplets = []
    xg = x.to(dev).requires_grad_(True)
    lh = crit.compute_hyp(xg, lab.to(dev))
    lh.backward()
    a, p, n, live = crit.debug_triplets
    T = a.numel()
    plan = crit.make_plan(lab)
    assert T == plan.T == crit.last_T
    assert int(live.sum()) == int(crit.last_kept)                         # the exported flags are the kernel's kept count
    # ---- validity of the sampled triplets (loss_and_miner_utils.py:33-61: same class, not the anchor; another class)
    lab_d = lab.to(dev)
    assert bool((lab_d[a] == lab_d[p]).all()) and bool((a != p).all()) and bool((lab_d[a] != lab_d[n]).all())
    assert bool((torch.bincount(a, minlength=M).cpu() == plan.k[lab].long()).all())      # every anchor exactly k_l times
    # ---- (i) the miner's rule (triplet_margin_miner.py:20-30, 'easy', margin 0) in fp64 on the normalised rows
    xn64 = F.normalize(x.double(), dim=1).to(dev)
    tm = 0.5 * ((xn64[a] * xn64[p]).sum(1) - (xn64[a] * xn64[n]).sum(1))
    odd = (tm > 0) != live
    assert int(odd.sum()) <= max(8, T // 20000), int(odd.sum())
    assert float(tm[odd].abs().max()) <= 2e-6 if bool(odd.any()) else True
    # ---- (ii) + (iii) loss and gradients on the triplets the kernel kept, fp64 oracle
    kept = (a[live], p[live], n[live])
    want, gx64, gs64 = oracle_hyphc(x, kept, scale, 0.05, oracle_device)
    assert abs(float(lh) - want) <= 2e-6 * abs(want), (float(lh), want)
    e_gx = relerr(xg.grad, gx64)
    assert e_gx <= 2e-4, e_gx
    assert abs(float(crit.scale.grad) - gs64) <= 1e-3 * abs(gs64) + 1e-12, (float(crit.scale.grad), gs64)
    if oracle_device != "cpu":
        # anchor torch-on-the-device to torch-on-the-host: the same oracle terms on a 200 k subsample, both in fp64
        sel = torch.arange(0, kept[0].numel(), max(1, kept[0].numel() // 200000), device=dev)
        sub = tuple(t[sel] for t in kept)
        s64 = torch.tensor([scale], dtype=torch.float64)
        t_dev = OM.hyphc_terms(x.double().to(dev), sub, s64.to(dev), 0.05).cpu()
        t_cpu = OM.hyphc_terms(x.double(), tuple(t.cpu() for t in sub), s64, 0.05)
        assert float((t_dev - t_cpu).abs().max()) <= 1e-10            # two libms: measured 1e-12 on O(1) terms
    if gathered32:
        # the reference's own op order in fp32: M x M matrix, gathers, reflections (ultrametric_loss.py:57-93 as written)
        lh32 = OM.hyphc_loss_gathered(x, tuple(t.cpu() for t in kept), torch.tensor([scale]), 0.05)
        assert abs(float(lh) - float(lh32)) <= 1e-5 * abs(float(lh32)), (float(lh), float(lh32))
    # ---- the gradient-free instantiation on the same stream
    crit.debug_triplets = None
    crit.hyp_miner.seed -= 1
    with torch.no_grad():
        l3 = crit.compute_hyp(x.to(dev), lab_d)
    assert abs(float(l3) - want) <= 2e-6 * abs(want)
    if chi2:
        # positives: uniform over the anchor's class minus the anchor => every member of a class is drawn k_l times on
        # average as a positive; negatives: uniform over the other classes' points (loss_and_miner_utils.py:43-62)
        counts = torch.bincount(lab, minlength=C).double()
        kk = plan.k.double()
        pos_hist = torch.bincount(p, minlength=M).cpu().double()
        exp_pos = kk[lab]
        ok = exp_pos > 0
        stat = float((((pos_hist - exp_pos) ** 2)[ok] / exp_pos[ok]).sum())
        assert chi2_band(stat, int(ok.sum()) - int((counts > 1).sum())), ("positives", stat, int(ok.sum()))
        t_c = counts * kk                                                 # triplets per anchor class
        share = t_c / (M - counts)                                       # expected draws per outside point, per class
        exp_neg = share.sum() - share[lab]
        neg_hist = torch.bincount(n, minlength=M).cpu().double()
        stat = float((((neg_hist - exp_neg) ** 2) / exp_neg).sum())
        assert chi2_band(stat, M - C), ("negatives", stat, M)
        # and per anchor class the negatives are spread over the other classes in proportion to their sizes
        big = int(counts.argmax())
        from_big = n[lab_d[a] == big]
        hist_c = torch.bincount(lab_d[from_big], minlength=C).cpu().double()
        assert float(hist_c[big]) == 0.0
        exp_c = counts * from_big.numel() / (M - counts[big])
        exp_c[big] = 0.0
        okc = exp_c > 0
        stat = float((((hist_c - exp_c) ** 2)[okc] / exp_c[okc]).sum())
        assert chi2_band(stat, int(okc.sum()) - 1), ("negatives by class", stat)
    return {"T": T, "kept": int(live.sum()), "near_tie_flips": int(odd.sum()), "loss_rel": abs(float(lh) - want) / abs(want),
            "grad_l2": e_gx}


@pytest.mark.parametrize("variant,E,scale", [("4", 32, 0.03), ("4", 32, 1e-3), ("8", 32, 0.03), ("4", 16, 0.03), ("4", 64, 0.03)])
def test_sampled_hyphc_kernel_against_the_fp64_oracle(variant, E, scale, tuning):
    """M = 6000 with one class of 1500 rows (beyond the LDS positive table at every E: overflow rows and finished anchors
    leave through the ring), oracle on the host."""
    lab = one_big_class(6000, 7, 1500, 62)
    check_sampled_against_oracle(E, 6000, lab, 9, scale, "cpu", gathered32=True, variant=variant, tuning=tuning,
                                 chi2=(variant == "4" and E == 32 and scale == 0.03))


@pytest.mark.parametrize("variant", ["4", "8"])
def test_sampled_hyphc_kernel_at_the_bench_size(variant, tuning):
    """The benchmark's own loss workload: M = 32768, the Voronoi labels of bench.synthetic_batch (C = 50, four classes of
    2600-3800 points), t_per_anchor = 50, fraction 1.2 -> 16.9 M sampled triplets; scale 1e-3 (base_hyp_hc.py:62)."""
    lab = bench_labels()
    r = check_sampled_against_oracle(32, 32768, lab, 50, 1e-3, dev, gathered32=False, variant=variant,
                                     tuning=tuning, chi2=(variant == "4"))
    assert r["T"] > 16_000_000


@pytest.mark.parametrize("E", [4, 15, 50])
def test_sampled_hyphc_kernel_at_the_shipped_embedding_sizes(E):
    """checkpoints/shapenet/config.yaml:65-67 (E = 50), checkpoints/partnet/*/config.yaml (4; Bed 15): the device-RNG path
    runs these on zero-padded rows (ultrametric_loss.sampled_width)."""
    lab = one_big_class(6000, 7, 1500, 62)
    check_sampled_against_oracle(E, 6000, lab, 9, 0.03, "cpu", gathered32=True)


@pytest.mark.parametrize("E", [4, 15, 50])
def test_array_path_losses_at_the_shipped_embedding_sizes(E):
    """Host-RNG miner -> HypHC + CosFace, and the triplet-margin branch, at E = 4 / 15 / 50 against the fp64 oracle
    (values and gradients), like test_gpu_loss.py does at 32 / 16."""
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    M, C = 2000, 6
    x = OH.expmap0(det_normalish((M, E), 11 + E) * 0.5)
    lab = det_int((M,), 12, C)
    s = 0.02
    crit = MetricHyperbolicLoss(margin=0.35, t_per_anchor=10, fraction=1.2, scale=torch.nn.Parameter(torch.tensor([s])),
                                temperature=0.05, num_class=C, embedding_size=E, cosface=True, miner=True).to(dev)
    W = det_normalish((E, C), 13)
    crit.loss_cosface.W.data.copy_(W)
    xg = x.to(dev).requires_grad_(True)
    torch.manual_seed(5)
    out = crit.compute_loss(None, xg, lab.to(dev))
    lh, lm = out["loss_hyp"]["losses"], out["loss_metric"]["losses"]
    a, p, n, count = crit.last_triplets
    c = int(count)
    trip = (a[:c].cpu(), p[:c].cpu(), n[:c].cpu())
    torch.manual_seed(5)
    wa, wp, wn = OM.mine_triplets(x, lab, 10, 1.2, 0, "easy")                # the oracle's miner under the same seed
    assert abs(wa.numel() - c) <= 8
    x64 = x.double().requires_grad_(True)
    s64 = torch.tensor([s], dtype=torch.float64, requires_grad=True)
    W64 = W.double().requires_grad_(True)
    lh64 = OM.hyphc_loss(x64, trip, s64, 0.05, mat_mean="closed")
    lm64 = OM.cosface_loss(x64, lab, W64, 0.35, 2.0)
    assert abs(float(lh) - float(lh64)) <= 2e-6 * abs(float(lh64))
    assert abs(float(lm) - float(lm64)) <= 1e-5 * abs(float(lm64))
    lh32 = OM.hyphc_loss_gathered(x, trip, torch.tensor([s]), 0.05)
    assert abs(float(lh) - float(lh32)) <= 1e-5 * abs(float(lh32))
    (0.1 * lh + lm).backward()
    (0.1 * lh64 + lm64).backward()
    assert relerr(xg.grad, x64.grad) <= 2e-4
    assert relerr(crit.loss_cosface.W.grad, W64.grad) <= 2e-4
    assert abs(float(crit.scale.grad) - float(s64.grad)) <= 1e-3 * abs(float(s64.grad)) + 1e-9
    # triplet-margin branch (cosface=False): semihard miner + TripletMarginLoss with AvgNonZeroReducer
    crit2 = MetricHyperbolicLoss(margin=0.35, t_per_anchor=8, fraction=1.2, scale=torch.tensor([0.05], device=dev),
                                 temperature=0.05, num_class=C, embedding_size=E, cosface=False, miner=True)
    xg2 = x.to(dev).requires_grad_(True)
    torch.manual_seed(9)
    lm2 = crit2.compute_loss(None, xg2, lab.to(dev))["loss_metric"]["losses"]
    torch.manual_seed(9)
    OM.mine_triplets(x, lab, 8, 1.2, 0, "easy")
    t2 = OM.mine_triplets(x, lab, 8, 1.2, 0.35, "semihard")
    x64b = x.double().requires_grad_(True)
    want = OM.triplet_margin_loss(x64b, t2, 0.35)
    assert abs(float(lm2) - float(want)) <= 1e-4 * abs(float(want))
    lm2.backward()
    want.backward()
    assert relerr(xg2.grad, x64b.grad) <= 2e-3


@pytest.mark.parametrize("E", [15, 50])
def test_extractor_head_at_the_shipped_embedding_sizes(E):
    """VN_DGCNN_partseg(out_features = 15 / 50) against the oracle on the same kNN graphs (train mode, dropout masks),
    forward and parameter gradients of the head."""
    from hpcs_amd import ops
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    B, N, k = 4, 256, 10
    net = VN_DGCNN_partseg(3, E, k=k, dropout=0.0, pooling="mean", num_categories=16)
    P = det_params(OE.param_shapes(E, 16), 5)
    net.load_state_dict(P, strict=True)
    net.to(dev).train()
    x = det_cloud(B, N, 6)
    lab = torch.eye(16)[det_int((B,), 7, 16)]
    graphs, real_knn = [], ops.knn

    def recording_knn(xx, kk):
        idx_ = real_knn(xx, kk)
        graphs.append(idx_.cpu())
        return idx_
    ops.knn = recording_knn
    try:
        y = net(x.to(dev), lab.to(dev))
    finally:
        ops.knn = real_knn
    w = det_normalish((B, N, E), 9)
    (y * w.to(dev)).sum().backward()
    Po = OE.resolve_aliases({k_: v.clone().double() if v.is_floating_point() else v.clone() for k_, v in P.items()})
    learn = ["conv11.0.weight", "conv10.0.weight", "conv10.1.weight", "conv10.1.bias"]
    for k_ in learn:
        Po[k_].requires_grad_(True)
    it = iter(graphs)
    want = OE.vn_dgcnn_partseg(x.double(), lab.double(), Po, k=k, training=True, knn_fn=lambda xx, kk: next(it))
    assert y.shape == (B, N, E)
    assert float((y.detach().cpu().double() - want).abs().max() / want.abs().max()) <= 1e-4
    (want * w.double()).sum().backward()
    got = dict(net.named_parameters(remove_duplicate=False))          # conv10.1.* are the same tensors as bn10.*
    for k_ in learn:
        assert relerr(got[k_].grad, Po[k_].grad) <= 2e-3, k_


def test_smoke_step_at_embedding_size_4():
    """The driver's smoke() (one training step against the oracle: embeddings, kept set, both loss terms) at E = 4, the
    size of the shipped PartNet checkpoints."""
    import __graft_entry__ as ge
    ge.smoke(E=4)


def test_config2_full_size_step():
    """BASELINE configs[1] as ONE full step under -m gpu: B = 32, N = 1024, k = 20, E = 32, C = 50, 16 categories, flat
    CosFace, device RNG, SO(3), dropout, side-stream weight gradients, RAdam -- bench.build_step's exact construction.
    Checked: finite losses / parameters over 3 steps; the CosFace term, accuracy and IoU against the fp64 oracle /
    confusion matrix on the step's own embeddings; the HypHC term's kept count inside the plan; RAdam's projection;
    reproducibility of the first step under the same seeds."""
    import argparse
    import bench
    args = argparse.Namespace(batch=32, points=1024, k=20, embedding=32, num_class=50, num_categories=16, hierarchical=False)
    pts, cat, tgt = bench.synthetic_batch(32, 1024, 50, 16, seed=0)
    pts_d, cat_d, tgt_d = pts.to(dev), cat.to(dev), tgt.to(dev)
    firsts = []
    for _ in range(2):
        step = bench.build_step(args, torch.device(dev))
        step.use_graph = False
        emb = []
        hook = step.nn_feat.register_forward_hook(lambda m_, i_, o_: emb.append(o_.detach()))
        W0 = step.metric_hyp_loss.loss_cosface.W.detach().cpu().clone()
        out = step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
        hook.remove()
        first = {k_: float(v) for k_, v in out.items()}
        firsts.append(first)
        assert all(math.isfinite(v) for v in first.values()), first
        # CosFace (a14) + in-step metrics on the step's own embeddings: fp64 oracle on the host (M = 32768, C = 50)
        xp = OH.expmap0(emb[0].cpu().double()).reshape(-1, 32)
        lab = tgt.reshape(-1)
        logits = OM.cosface_logits(xp, lab, W0.double(), 0.35, 2.0)
        lm = F.cross_entropy(logits, lab, reduction="none").mean()
        assert abs(first["loss_metric"] - float(lm)) <= 1e-5 * abs(float(lm)), (first["loss_metric"], float(lm))
        pred = logits.argmax(1)
        assert abs(first["acc"] - float((pred == lab).double().mean())) <= 2e-4            # fp32 / fp64 arg-max near-ties
        plan = step.metric_hyp_loss.make_plan(lab)
        assert 0 < int(step.metric_hyp_loss.last_kept) <= plan.T == step.metric_hyp_loss.last_T
        for _s in range(2):
            out = step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
        assert all(math.isfinite(float(v)) for v in out.values())
        for p_ in step.parameters():
            assert bool(torch.isfinite(p_).all())
            rows = p_.detach().reshape(-1, p_.shape[-1] if p_.dim() else 1)
            assert float(rows.norm(dim=1).max()) <= 1 - 4e-3 + 1e-5
    assert abs(firsts[0]["loss_metric"] - firsts[1]["loss_metric"]) <= 1e-6 * abs(firsts[0]["loss_metric"])
    assert abs(firsts[0]["loss_hyp"] - firsts[1]["loss_hyp"]) <= 1e-5 * abs(firsts[0]["loss_hyp"])


@pytest.mark.parametrize("E,scale", [(2, 0.5), (6, 0.3), (32, 0.05)])
def test_hyp_lca_coordinate_output_is_differentiable(E, scale):
    """hyp_lca(a, b, return_coord=True) (hpcs/distances/lca.py:37-52) returns the projection point WITH its gradient, like
    the reference's; forward and vector-Jacobian product against the oracle's reflections evaluated in fp64."""
    from hpcs_amd.distances import hyp_lca
    a = det_normalish((300, E), 9701) * scale / E ** 0.5
    b = det_normalish((300, E), 9702) * scale / E ** 0.5
    b[0] = a[0] * 0.5                                   # collinear with the origin: the geodesic is a diameter, proj = 0
    w = det_normalish((300, E), 9703)
    ag, bg = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    p = hyp_lca(ag, bg, return_coord=True)
    assert p.requires_grad
    (p * w.to(dev)).sum().backward()
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    live = torch.ones(300, 1, dtype=torch.float64)
    live[0] = 0.0
    p64 = OH.hyp_lca(a64, b64, True)
    (p64 * w.double() * live).sum().backward()
    # the kernel reduces |a|^2, a.b, |b|^2 in fp32 and continues in fp64: rows where a, b and the origin are nearly collinear
    # (sin^2 of the angle below 1e-2; frequent at E = 2) lose digits in the Gram determinant -- held to finiteness only
    sin2 = 1 - (a * b).sum(1).double() ** 2 / ((a * a).sum(1).double() * (b * b).sum(1).double())
    good = sin2 > 1e-2
    good[0] = False
    assert int(good.sum()) > 200
    assert float(p[0].abs().max()) <= 1e-6
    assert relerr(p[good], p64[good]) <= 5e-6, relerr(p[good], p64[good])
    assert relerr(ag.grad[good], a64.grad[good]) <= 5e-5, relerr(ag.grad[good], a64.grad[good])
    assert relerr(bg.grad[good], b64.grad[good]) <= 5e-5
    assert bool(torch.isfinite(ag.grad).all()) and bool(torch.isfinite(bg.grad).all()) and bool(torch.isfinite(p).all())


@pytest.mark.parametrize("B,N,k,pool,training", [(2, 64, 20, True, True), (3, 96, 8, True, True), (2, 64, 20, False, True),
                                                 (2, 52, 40, True, True), (2, 64, 20, True, False), (1, 1024, 20, True, True)])
def test_recompute_edge_layer_equals_the_materialised_layer(B, N, k, pool, training, monkeypatch):
    """csrc/vn_recompute.hip: an edge-level VNLinearLeakyReLU (vn_layers.py:48-77 + VNBatchNorm :112-132 + mean_pool :150-153,
    21 -> 21 channels) whose four passes re-form the pre-activations on the matrix cores, against the streaming kernels over
    a materialised pre-activation tensor: forward output to an ulp (the f32 MFMA is the GEMM's fma chain, the pooled sums keep
    their order), batch statistics and gradients to fp32 summation-order noise."""
    from hpcs_amd import ops
    Cin = Cout = 21
    L = N * k
    x = det_normalish((B, Cin, 3, N, k), 9801)
    Wf = det_normalish((Cout, Cin), 9802) / Cin ** 0.5
    Wd = det_normalish((Cout, Cin), 9803) / Cin ** 0.5
    gamma = 1.0 + 0.1 * det_normalish((Cout,), 9804)
    beta = 0.1 * det_normalish((Cout,), 9805)
    w = det_normalish((B, Cout, 3, N) if pool else (B, Cout, 3, N, k), 9806)
    monkeypatch.setattr(ops, "RC_MIN_L", 0)
    res = []
    for rc in (False, True):
        monkeypatch.setattr(ops, "RC_EDGE", rc)
        t = [v.to(dev).requires_grad_(True) for v in (x, Wf, Wd, gamma, beta)]
        rmean, rvar = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        out = ops.vn_layer(t[0], t[1], t[2], t[3], t[4], rmean, rvar, training, slope=0.2, pool_k=k if pool else 0)
        (out * w.to(dev)).sum().backward()
        torch.cuda.synchronize()
        res.append([out.detach().cpu(), rmean.cpu(), rvar.cpu()] + [v.grad.cpu() for v in t])
    ref, got = res
    assert got[0].shape == ref[0].shape
    # same fma chains and the same summation orders, but two compilations of the point-wise code (the compiler contracts
    # a*b+c into fma where it sees fit): equal to an ulp here and there (measured: max 3e-8 absolute with given statistics)
    assert relerr(got[0], ref[0]) <= (1e-6 if training else 2e-7), relerr(got[0], ref[0])
    assert float((got[0] - ref[0]).abs().max()) <= 1e-6 * float(ref[0].abs().max())
    assert torch.equal(got[1], ref[1]) or relerr(got[1], ref[1]) <= 1e-6        # statistics: another fp32 / fp64 split of the sums
    assert relerr(got[2], ref[2]) <= 1e-6
    errs = {name: relerr(got[3 + i], ref[3 + i]) for i, name in enumerate(("gx", "gWf", "gWd", "ggamma", "gbeta"))}
    assert max(errs.values()) <= 2e-5, errs


@pytest.mark.parametrize("form", ["side", "serial"])
def test_captured_step_holds_no_memset_node_and_replays_at_the_bench_geometry(form):
    """the source scan over csrc/ cannot see what torch's own ops put into a capture.  The step driver walks the
    captured hipGraph (hpcs_debug_graph_census) and refuses one that holds a memset node (their arguments do not survive a
    second graph on this runtime).  Here: BASELINE configs[1] geometry (B = 32, N = 1024, k = 20, C = 50), both
    capture forms -- the census must show kernel nodes only (memcpy nodes are reported, they replay correctly), and 300
    replays across device-wide synchronisations keep losses finite and metrics in range."""
    import argparse
    import bench
    args = argparse.Namespace(batch=32, points=1024, k=20, embedding=32, num_class=50, num_categories=16, hierarchical=False)
    step = bench.build_step(args, torch.device(dev))
    step.use_graph, step.graph_auto = True, False
    step._graph_form = form
    pts, cat, tgt = bench.synthetic_batch(32, 1024, 50, 16, seed=0)
    pts_d, cat_d, tgt_d = pts.to(dev), cat.to(dev), tgt.to(dev)
    first = None
    for i in range(304):
        if i % 50 == 0:
            torch.cuda.synchronize()
        out = step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
        if i == 4:
            assert step._graph is not None and step._graph["form"] == form
            nodes = step.last_capture_nodes["fwd_bwd"]
            assert nodes is not None, "this torch cannot hand out the captured graph (keep_graph): the census did not run"
            assert nodes.get("memset", 0) == 0 and nodes.get("kernel", 0) > 100, nodes
        if i % 25 == 0 or i == 303:
            v = {k_: float(x) for k_, x in out.items()}
            assert all(math.isfinite(x) for x in v.values()) and 0.0 <= v["acc"] <= 1.0 and 0.0 <= v["iou"] <= 1.0, (i, v)
            first = first or v
    assert step.graph_mode() == "graph" and step.use_graph
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(step.last_capture_nodes, open(os.path.join(root, "gpurun_out", f"r4_capture_nodes_{form}.json"), "w"), indent=1)


def test_capture_with_a_memset_node_is_refused(monkeypatch):
    """The refusal itself: a census that reports a memset node makes _capture raise, and the step falls back to eager with a
    warning (the same path as any failed capture)."""
    import argparse
    import warnings
    import bench
    from hpcs_amd import step as S
    args = argparse.Namespace(batch=4, points=256, k=10, embedding=32, num_class=6, num_categories=16, hierarchical=False)
    step = bench.build_step(args, torch.device(dev))
    step.use_graph, step.graph_auto = True, False
    pts, cat, tgt = bench.synthetic_batch(4, 256, 6, 16, seed=0)
    real = S._lib_graph_census

    def fake(graph):
        c = real(graph)
        if c is not None:
            c = dict(c, memset=2)
        return c
    monkeypatch.setattr(S, "_lib_graph_census", fake)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for _ in range(6):
            out = step.training_step(pts.to(dev), cat.to(dev), tgt.to(dev), targets_host=tgt)
    assert real is not None and step.use_graph is False and step._graph is None
    assert any("memset node" in str(w.message) for w in rec), [str(w.message) for w in rec]
    assert all(math.isfinite(float(v)) for v in out.values())


def test_in_launch_split_reduction_is_reproducible_under_traffic():
    """nt_fused_reduce publishes split partials across workgroups with write-through stores, a vmcnt(0) drain and a
    relaxed agent-scope ticket (no fence).  A partial that was not visible to the reducing workgroup would change the sum.
    Stress: 1500 launches of the weight-gradient shapes of the step with REUSED, dirty workspaces (the previous launch's
    partials and noise in L2) while a second stream streams 256 MB copies through the memory system.  The reduction has a
    fixed summation order, so EVERY result must equal the first one bit for bit; and it must agree with hpcs_gemm_nt +
    split_reduce (HPCS_NT_FUSED=0's path: the same partials in another fixed order) to fp32 rounding of the sum."""
    from hpcs_amd import _lib
    call = _lib.call
    shapes = [(42, 21, 3 * 20480, 8), (42, 42, 3 * 1024, 32), (128, 256, 4096, 8), (64, 341, 3 * 1024, 16)]
    side = torch.cuda.Stream()
    noise_a = torch.empty(64 << 20, device=dev)
    noise_b = torch.empty(64 << 20, device=dev)
    for si, (M, Kx, Nc, nb) in enumerate(shapes):
        G = det_normalish((nb, M, Nc), 9900 + si).to(dev)
        X = det_normalish((nb, Kx, Nc), 9950 + si).to(dev)
        nbytes = call("hpcs_gemm_nt_workspace", M, Kx, Nc, nb, 1)
        nt = call("hpcs_gemm_nt_tickets", M, Kx, Nc, nb, 1)
        ws = torch.empty(nbytes // 4 + 4, device=dev).normal_()           # dirty from the start
        ws2 = torch.empty(nbytes // 4 + 4, device=dev)
        tickets = torch.zeros(nt, dtype=torch.int32, device=dev)
        two = torch.zeros(M, Kx, device=dev)
        call("hpcs_gemm_nt", G, Nc, M * Nc, 0, X, Nc, Kx * Nc, 0, M, Kx, Nc, nb, 1, two, Kx, 0, ws2, nbytes)
        first = torch.empty(M, Kx, device=dev)
        call("hpcs_gemm_nt_fused", G, Nc, M * Nc, 0, X, Nc, Kx * Nc, 0, M, Kx, Nc, nb, 1, first, Kx, 0, ws, nbytes, tickets, nt)
        torch.testing.assert_close(first, two, rtol=2e-6, atol=1e-6 * float(two.abs().max()))
        outs = [torch.empty(M, Kx, device=dev) for _ in range(4)]
        differ = torch.zeros((), dtype=torch.int64, device=dev)
        for r in range(375):
            if r % 8 == 0:
                with torch.cuda.stream(side):
                    noise_b.copy_(noise_a, non_blocking=True)
            o = outs[r % 4]
            call("hpcs_gemm_nt_fused", G, Nc, M * Nc, 0, X, Nc, Kx * Nc, 0, M, Kx, Nc, nb, 1, o, Kx, 0, ws, nbytes, tickets, nt)
            differ += (o != first).sum()
        torch.cuda.synchronize()
        assert int(differ) == 0, (M, Kx, Nc, nb, int(differ))
        assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("M,K,Nc,nb,ni,bias,acc", [(682, 341, 1024, 2, 3, True, False), (341, 682, 1024, 2, 3, False, False),
                                                   (340, 341, 3072, 2, 1, False, True), (256, 256, 1000, 3, 1, False, False),
                                                   (65, 7, 130, 2, 1, True, True), (130, 300, 33, 1, 1, False, False)])
def test_bf16x3_gemm_is_as_close_to_fp64_as_the_fma_chain(M, K, Nc, nb, ni, bias, acc, tuning):
    """csrc/gemm_x3.hip: the large dense layers evaluate every fp32 product as six exact bf16-piece products on the bf16
    matrix cores with fp32 accumulation.  Against the fp64 product of the same fp32 operands its error must not exceed the
    fma-chain kernel's (hpcs_gemm_tn with HPCS_GEMM_X3=0: the k-ordered chain that equals the reference's sgemm) by more than
    a small factor, on the step's shapes and on ragged ones (edge tiles, K not a multiple of 16, odd column counts, row bias,
    accumulation, the [B,K,3,L] walk as 3B matrices); also with operands spanning 12 orders of magnitude per row."""
    from hpcs_amd import _lib
    call = _lib.call
    W = det_normalish((M, K), 9601 + M) / K ** 0.5
    X = det_normalish((nb, K, ni, Nc), 9602 + K)
    X[:, : max(1, K // 3)] *= 1e-6                      # wide dynamic range inside every dot product
    X[:, -max(1, K // 5):] *= 1e+3
    rb = det_normalish((nb * ni, M), 9603) if bias else None
    C0 = det_normalish((nb, M, ni, Nc), 9604)
    want = torch.einsum("mk,bkin->bmin", W.double(), X.double())
    if bias:
        want = want + rb.double().view(nb, ni, M).permute(0, 2, 1).unsqueeze(-1)
    if acc:
        want = want + C0.double()
    At = W.t().contiguous().to(dev)
    Xd = X.to(dev)
    errs = {}
    for x3 in (0, 1):
        tuning("HPCS_GEMM_X3", x3)
        out = C0.to(dev).clone()
        call("hpcs_gemm_tn", At, M, Xd, ni * Nc, K * ni * Nc, Nc, out, ni * Nc, M * ni * Nc, Nc, M, Nc, K, nb, ni,
             None if rb is None else rb.to(dev), 1 if acc else 0)
        d = out.cpu().double() - want
        errs[x3] = (float(d.norm() / want.norm()), float(d.abs().max() / want.abs().max()))
    assert errs[1][0] <= 1.25 * errs[0][0] + 1e-9 and errs[1][1] <= 2.0 * errs[0][1] + 1e-9, errs
    assert errs[1][0] <= 1e-6, errs


# ---------------------------------------------------------------------------- vector / scalar forms of the per-channel VN layer
@pytest.mark.parametrize("B,Cin,Cout,N,k,pool", [
    (2, 42, 21, 128, 20, False),      # EdgeConv shape, L = N k % 4 == 0: vector forward apply, vector backward apply
    (2, 42, 21, 128, 20, True),       # ... with the fused mean over k (pool_k % 4 == 0: the four positions share one pooled cell)
    (2, 2, 21, 128, 20, False),       # conv1: two input channels, the pre-activations formed on the fly (K2 instantiations)
    (2, 2, 21, 128, 20, True),
    (3, 6, 5, 37, 10, False),         # L = 370: not a multiple of 4 -> the scalar kernels
    (3, 6, 5, 37, 10, True),          # pool_k = 10: not a multiple of 4 -> scalar backward, generic pooled forward
    (2, 2, 4, 30, 7, True),
    (2, 12, 6, 1, 1024, False),       # dim-4 style plane (vn1 / vn2): one row of 1024 positions
])
def test_per_channel_vn_layer_vector_and_scalar_forms_against_the_fp64_oracle(B, Cin, Cout, N, k, pool):
    """VNLinearLeakyReLU with one direction per feature channel (vn_layers.py:48-77, 112-132; mean_pool :152-153) on shapes
    that take the four-positions-per-thread kernels (vn_apply_fwd_vec / vn_bwd_apply_vec, plain and two-input-channel forms)
    and on shapes that must fall back to the scalar ones: output, input gradient and every parameter gradient against the
    oracle's op sequence evaluated in fp64 (train mode)."""
    from hpcs_amd.nn.dgcnn.utils.vn_layers import VNLinearLeakyReLU
    dev = "cuda"
    x = det_normalish((B, Cin, 3, N, k), 900 + Cin + N)
    m = VNLinearLeakyReLU(Cin, Cout, dim=5)
    sd = m.state_dict()
    m.load_state_dict(det_params({k_: tuple(v.shape) for k_, v in sd.items()}, 901))
    m = m.to(dev).train()
    P = {"c." + k_: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point and "running" not in k_)
         for k_, v in m.state_dict().items()}
    xg = x.to(dev).requires_grad_(True)
    y = m(xg, pool_k=k) if pool else m(xg)
    xr = x.double().requires_grad_(True)
    yr = OE.vn_linear_leaky_relu(xr, P, "c", True)
    if pool:
        yr = yr.mean(dim=-1)
    assert tuple(y.shape) == tuple(yr.shape)
    w = det_normalish(tuple(yr.shape), 902)
    (y * w.to(dev)).sum().backward()
    (yr * w.double()).sum().backward()
    assert relerr(y.detach().cpu().double(), yr.detach()) < 2e-6
    assert relerr(xg.grad.cpu().double(), xr.grad) < 3e-5
    for n_, p_ in m.named_parameters():
        assert relerr(p_.grad.cpu().double(), P["c." + n_].grad) < 8e-5, n_


def test_bench_line_keeps_the_driver_contract():
    """`python bench.py --steps 3 --warmup 1` (one GPU): exactly one JSON line on stdout carrying the keys the driver and the
    judge read -- the BASELINE metric on configs[1], `roofline` of the dominant kernel with frac = achieved / peak and the
    PMC traffic, `cpu_baseline` of the oracle on the host cores -- and a value that is the batch over the measured step time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1"], capture_output=True,
                       text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "N=1024,k=20" in d["metric"] and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-3
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
