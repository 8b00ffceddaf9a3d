"""Round-2 GPU parity: the extractor on the REFERENCE's own kNN graphs (1e-5 bar of the north star, no outlier
allowance), graph flips confined to near-ties, f2 / get_triplets pins, cfg-3 / cfg-5 end to end, RNG stream checks.

Measured error levels are also written to gpurun_out/parity_r02.json (the numbers DESIGN.md section 3 quotes).
"""
import json
import os

import numpy as np
import pytest
import torch

from _det import det_cloud, det_int, det_normalish, det_params, det_uniform
from oracle import extractor as OE, hyperbolic as OH, metric as OM

pytestmark = pytest.mark.gpu
dev = "cuda"
T = torch.from_numpy
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REPORT = {}
FAILS = []


@pytest.fixture(scope="module")
def golden2():
    return dict(np.load(os.path.join(HERE, "golden", "reference_vectors_r2.npz")))


@pytest.fixture(scope="module")
def hier():
    return json.load(open(os.path.join(HERE, "golden", "hierarchy_lists.json")))


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_r02.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def relerr(a, b):
    """(max-abs / max|b|, L2 / |b|) in fp64."""
    a = a.detach().cpu().double()
    b = (T(np.asarray(b)) if not torch.is_tensor(b) else b.detach().cpu()).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    d = a - b
    return float(d.abs().max() / b.abs().max()), float(d.norm() / b.norm())


class _Tap:
    """Runs the HIP net with injected kNN graphs (or records its own) and taps the intermediates the fixture holds."""

    def __init__(self, net, graphs=None):
        self.net, self.graphs = net, graphs
        self.rec_graphs, self.rec_inputs, self.inter = [], [], {}

    def run(self, x, lab):
        from hpcs_amd import ops
        real_knn = ops.knn
        it = iter(self.graphs) if self.graphs is not None else None

        def knn(xx, kk):
            own = real_knn(xx, kk)
            self.rec_graphs.append(own.cpu())
            self.rec_inputs.append(xx.detach().cpu())
            return next(it).to(xx.device) if it is not None else own
        hooks = [self.net.conv2.register_forward_hook(lambda m, i, o: self.inter.__setitem__("x1", o.detach())),
                 self.net.conv4.register_forward_hook(lambda m, i, o: self.inter.__setitem__("x2", o.detach())),
                 self.net.conv5.register_forward_hook(lambda m, i, o: self.inter.__setitem__("x3", o.detach())),
                 self.net.conv6.register_forward_hook(lambda m, i, o: self.inter.__setitem__("g6", o.detach()))]
        real_frame = self.net.std_feature.frame

        def frame(*a, **k):
            out = real_frame(*a, **k)
            self.inter["z0"] = out[0].detach()
            return out
        self.net.std_feature.frame = frame
        ops.knn = knn
        try:
            return self.net(x, lab)
        finally:
            ops.knn = real_knn
            self.net.std_feature.frame = real_frame
            for h in hooks:
                h.remove()


def _net(E):
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    net = VN_DGCNN_partseg(in_channels=3, out_features=E, k=10, dropout=0.0, pooling="mean", num_categories=16)
    sd = det_params(OE.param_shapes(E, 16), 40 + E)
    net.load_state_dict(sd, strict=True)
    return net.to(dev), sd


# ---------------------------------------------------------------------------------------------- a6 at 1e-5
def _oracle_eval(x, lab, sd, graphs, dtype, device, training):
    """The oracle (= the reference's torch op sequence) in `dtype` on `device`, with the given graphs; taps included."""
    P = OE.resolve_aliases({k_: (v.to(dtype).clone().to(device) if v.is_floating_point() else v.clone().to(device))
                            for k_, v in sd.items()})
    it = iter(graphs)
    y, aux = OE.vn_dgcnn_partseg(x.to(dtype).to(device), lab.to(dtype).to(device), P, k=10, training=training,
                                 knn_fn=lambda xx, kk: next(it).to(device), return_aux=True)
    aux = {k_: v.detach().cpu() for k_, v in aux.items()}
    aux["out"] = y.detach().cpu()
    return aux


@pytest.mark.parametrize("E", [2, 32])
def test_extractor_on_reference_graphs_1e5(golden, golden2, E):
    """north_star: "fp32 embeddings within 1e-5 rel".  With the reference's three kNN graphs injected (only arithmetic
    is compared, not the discrete graph; test below covers the graphs) and NO outlier allowance:

      eval : embeddings within 1e-5 * max|ref| of the reference at every element and in L2 (measured 5e-7 / 2e-7).
      train: 13 train-mode batch-norms amplify fp32 rounding ~150x (a 6e-8 relative input perturbation moves the fp64
             output by 7e-6 L2): the REFERENCE's own output sits 2.2e-5 / 3.8e-5 (max, E=2 / 32) from an fp64 evaluation
             of the same graph, and the same torch op sequence executed by PyTorch-ROCm on this GPU sits 2e-5 / 1e-4
             from it.  1e-5 against the reference is therefore not a property any fp32 evaluation order has; the bar is
             stated against the exact value and the two torch realisations, all measured in this test:
                 err(HIP, fp64) <= 2.5 * max(err(reference, fp64), err(torch-on-GPU fp32, fp64))     (max and L2)
             measured r02: HIP 2.5e-5 / 1.9e-4 max, 1.3e-5 / 7.1e-5 L2 (E = 2 / 32).
      per-block intermediates (x1, x2, x3, conv6 output, frame z0; strided samples of the reference's tensors), eval and
      train: err(HIP, fp64) <= 3 * max(err(reference, fp64), err(torch-on-GPU, fp64)) + 1e-5 per tensor (the frames z0 are the ill-conditioned stage: the reference itself is 2e-5 (eval)
      to 2e-4 (train) from fp64 there)."""
    net, sd = _net(E)
    x = det_cloud(4, 256, 50 + E)
    lab = torch.eye(16)[det_int((4,), 51, 16)]
    for mode in ("eval", "train"):
        training = mode == "train"
        graphs = [T(golden2[f"net_E{E}_{mode}_knn{i}"].astype(np.int64)) for i in (1, 2, 3)]
        net.train(training)
        tap = _Tap(net, graphs)
        with torch.no_grad():
            y = tap.run(x.to(dev), lab.to(dev))
        a64 = _oracle_eval(x, lab, sd, graphs, torch.float64, "cpu", training)
        agpu = _oracle_eval(x, lab, sd, graphs, torch.float32, dev, training)
        ref = T(golden[f"net_E{E}_{mode}_out"])
        e_ref, e_64 = relerr(y, ref), relerr(y, a64["out"])
        n_ref, n_gpu = relerr(ref, a64["out"]), relerr(agpu["out"], a64["out"])
        rep = {"out_hip_vs_reference": e_ref, "out_hip_vs_fp64": e_64, "out_reference_vs_fp64": n_ref,
               "out_torch_gpu_fp32_vs_fp64": n_gpu}
        checks = []
        if not training:
            checks.append(("out vs reference", max(e_ref), 1e-5))
        else:
            checks.append(("out vs fp64 (max)", e_64[0], 2.5 * max(n_ref[0], n_gpu[0])))
            checks.append(("out vs fp64 (L2)", e_64[1], 2.5 * max(n_ref[1], n_gpu[1])))
        for nm, t in tap.inter.items():
            stride = int(golden2[f"net_E{E}_{mode}_{nm}_stride"])
            r_s = T(golden2[f"net_E{E}_{mode}_{nm}_samp"]).double()
            amax = float(golden2[f"net_E{E}_{mode}_{nm}_absmax"])
            h_s = t.flatten()[::stride].cpu().double()
            f_s = a64[nm].flatten()[::stride]
            g_s = agpu[nm].flatten()[::stride].double()
            e_h = float((h_s - f_s).abs().max() / amax)
            e_r = float((r_s - f_s).abs().max() / amax)
            e_g = float((g_s - f_s).abs().max() / amax)
            rep[nm] = {"hip_vs_fp64": e_h, "reference_vs_fp64": e_r, "torch_gpu_vs_fp64": e_g,
                       "hip_vs_reference": float((h_s - r_s).abs().max() / amax)}
            checks.append((f"{nm} vs fp64", e_h, 3.0 * max(e_r, e_g) + 1e-5))
        FAILS.extend((f"E{E} {mode} {what}", v, b) for what, v, b in checks if not v <= b)
        REPORT[f"extractor_E{E}_{mode}_pinned_graphs"] = rep
    mine = [f for f in FAILS if f[0].startswith(f"E{E} ")]
    assert not mine, mine


@pytest.mark.parametrize("E", [2, 32])
def test_hip_graphs_differ_from_reference_only_at_near_ties(golden2, E):
    """The HIP path's OWN graphs on its OWN features (nothing injected): layer 1 (coordinates) must be the reference's
    graph exactly; in layers 2-3 a row may differ only by swapping candidates whose reference-formula scores (oracle, on
    the HIP features) lie within 2e-5 relative of the row's k-th score -- i.e. near-ties that fp32 feature noise may
    order either way -- and such rows must be rare."""
    net, _ = _net(E)
    x = det_cloud(4, 256, 50 + E)
    lab = torch.eye(16)[det_int((4,), 51, 16)]
    for mode in ("eval", "train"):
        net.train(mode == "train")
        tap = _Tap(net, None)
        with torch.no_grad():
            tap.run(x.to(dev), lab.to(dev))
        flips = []
        for li in range(3):
            g_ref = T(golden2[f"net_E{E}_{mode}_knn{li + 1}"].astype(np.int64))
            g_hip = tap.rec_graphs[li]
            rows = (g_hip != g_ref).any(-1).nonzero()
            flips.append(len(rows))
            if li == 0:
                assert len(rows) == 0
                continue
            assert len(rows) <= 0.01 * g_ref.shape[0] * g_ref.shape[1], len(rows)
            if len(rows) == 0:
                continue
            scores = OE.knn_scores(tap.rec_inputs[li])                     # [B,N,N] reference formula, fp32
            for b, i in rows.tolist():
                s = scores[b, i].double()
                a_, r_ = set(g_hip[b, i].tolist()), set(g_ref[b, i].tolist())
                kth = float(s[g_ref[b, i, -1]])
                for j in a_ ^ r_:
                    assert abs(float(s[j]) - kth) <= 2e-5 * max(abs(kth), 1e-30), (li, b, i, j, float(s[j]), kth)
        REPORT[f"graph_flips_E{E}_{mode}_rows_of_1024"] = flips


def test_layer_tolerances_tightened(golden):
    """VNStdFeature / VNLinearLeakyReLU against the reference at what the kernels deliver (was rtol 1e-3 in round 1)."""
    from hpcs_amd.nn.dgcnn.utils.vn_layers import VNStdFeature
    x8 = det_normalish((2, 8, 3, 24), 38)
    m = VNStdFeature(8, dim=4, normalize_frame=False)
    m.load_state_dict(det_params({k: tuple(v.shape) for k, v in m.state_dict().items()}, 39))
    m.to(dev)
    rep = {}
    for mode in ("train", "eval"):
        m.train(mode == "train")
        xs, z0 = m(x8.to(dev))
        e1, e2 = relerr(xs, golden[f"vnstd_{mode}_xstd"]), relerr(z0, golden[f"vnstd_{mode}_z0"])
        rep[mode] = {"xstd": e1, "z0": e2}
        # train: batch statistics over only 48 positions and two stacked VN layers; eval: plain arithmetic
        bar = 2e-5 if mode == "train" else 5e-6
        assert max(e1[0], e2[0]) <= bar, (mode, e1, e2)
    REPORT["vnstd_vs_reference"] = rep


# ---------------------------------------------------------------------------------------------- f2
@pytest.mark.parametrize("cat,M", [("Small", 48), ("Chair", 96), ("Table", 96)])
def test_hierarchical_loss_vs_reference(golden2, hier, cat, M):
    """hpcs_amd.loss.hierarchical_cosface_loss.hierarchical_loss (hpcs_hier_nll_fwd/bwd) against the reference's own
    outputs: value and the gradient w.r.t. the logits (through a device softmax)."""
    from hpcs_amd.loss.hierarchical_cosface_loss import hierarchical_loss
    C = hier[cat]["num_class"]
    lst = hier[cat]["hierarchy_list"]
    logits = (det_normalish((M, C), 200 + C) * 2.0).to(dev).requires_grad_(True)
    tgt = det_int((M,), 201 + C, C).to(dev)
    loss = hierarchical_loss(torch.softmax(logits, dim=1), tgt, lst)
    loss.backward()
    want = float(golden2[f"hier_{cat}_loss"])
    assert abs(float(loss) - want) <= 1e-5 * abs(want), (float(loss), want)
    e = relerr(logits.grad, golden2[f"hier_{cat}_glogits"])
    assert e[0] <= 1e-5 and e[1] <= 1e-5, e
    REPORT[f"hierarchical_loss_{cat}"] = {"loss_rel": abs(float(loss) - want) / abs(want), "glogits": e}


@pytest.mark.parametrize("cat", ["Chair", "Table"])
def test_fused_hierarchical_cosface_equals_softmax_plus_hierarchical_loss(hier, cat):
    """The training path fuses logits -> softmax -> branch sums -> NLL into hpcs_cosface_fwd/bwd; it must equal the
    (reference-pinned) stand-alone hierarchical_loss applied to the softmax of the logits the same kernel emits, and
    the fp64 oracle, on the Chair-3 / Table-3 level lists."""
    from hpcs_amd.loss.hierarchical_cosface_loss import HierarchicalCosFaceLoss, hierarchical_loss
    C = hier[cat]["num_class"]
    lst = hier[cat]["hierarchy_list"]
    M, E = 1500, 32
    x = OH.expmap0(det_normalish((M, E), 71) * 0.5)
    lab = det_int((M,), 72, C)
    W = det_normalish((E, C), 73)
    crit = HierarchicalCosFaceLoss(C, E, margin=0.35, scale=2, hierarchy_list=lst).to(dev)
    crit.W.data.copy_(W)
    xg = x.to(dev).requires_grad_(True)
    fused = crit(xg, lab.to(dev))
    two_step = hierarchical_loss(torch.softmax(crit.last_logits, dim=1), lab.to(dev), lst)
    assert abs(float(fused) - float(two_step)) <= 2e-6 * abs(float(two_step))
    x64, W64 = x.double().requires_grad_(True), W.double().requires_grad_(True)
    want = OM.hierarchical_cosface_loss(x64, lab, W64, lst, 0.35, 2.0)
    assert abs(float(fused) - float(want)) <= 1e-5 * abs(float(want))
    fused.backward()
    want.backward()
    ex, ew = relerr(xg.grad, x64.grad), relerr(crit.W.grad, W64.grad)
    assert ex[1] <= 1e-4 and ew[1] <= 1e-4, (ex, ew)
    out = crit.compute_loss(x.to(dev), lab.to(dev))           # PML loss-dict contract
    assert out["loss"]["reduction_type"] == "element" and out["loss"]["losses"].shape == (M,)
    assert abs(float(out["loss"]["losses"].mean()) - float(fused)) <= 1e-6 * abs(float(fused))


# ---------------------------------------------------------------------------------------------- a12 miner-less path
def test_get_triplets_fallback_vs_reference_and_oracle(golden2):
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    crit = MetricHyperbolicLoss(t_per_anchor=3, fraction=1.2, scale=torch.nn.Parameter(torch.tensor([0.05])),
                                num_class=3, embedding_size=8, cosface=True, miner=False).to(dev)
    torch.manual_seed(4321)
    a, p, n = crit.get_triplets(7)                              # the reference's signature and CPU draw order
    assert torch.equal(torch.stack([a, p, n]), T(golden2["get_triplets_n7_t3"].astype(np.int64)))
    M, E = 12, 8
    crit.t_per_anchor = 5
    x = OH.expmap0(det_normalish((M, E), 81) * 0.5)
    lab = det_int((M,), 82, 3)
    xg = x.to(dev).requires_grad_(True)
    torch.manual_seed(4321)
    got = crit.compute_hyp(xg, lab.to(dev))
    trip = crit.last_triplets
    assert torch.equal(torch.stack([t.cpu() for t in trip[:3]]), T(golden2["get_triplets_n12_t5"].astype(np.int64)))
    x64 = x.double().requires_grad_(True)
    sc = torch.tensor([0.05], dtype=torch.float64, requires_grad=True)
    torch.manual_seed(4321)
    want = OM.hyphc_loss(x64, OM.all_pairs_triplets(M, 5), sc, 0.05)
    assert abs(float(got) - float(want)) <= 1e-5 * abs(float(want)), (float(got), float(want))
    got.backward()
    want.backward()
    e = relerr(xg.grad, x64.grad)
    assert e[1] <= 1e-4, e
    assert abs(float(crit.scale.grad) - float(sc.grad)) <= 1e-3 * abs(float(sc.grad))


def test_sampler_options_raise_named_errors():
    from hpcs_amd.miner.loss_and_miner_utils import get_balanced_random_triplet_indices
    lab = det_int((64,), 110, 4).to(dev)
    with pytest.raises(NotImplementedError, match="ref_labels"):
        get_balanced_random_triplet_indices(lab, ref_labels=lab.clone(), t_per_anchor=5, fraction=1.2)
    with pytest.raises(NotImplementedError, match="weighted"):
        get_balanced_random_triplet_indices(lab, t_per_anchor=5, fraction=1.2, weights=torch.ones(64, 64))
    a, p, n = get_balanced_random_triplet_indices(lab, ref_labels=lab, t_per_anchor=5, fraction=1.2)   # same object: fine
    assert a.numel() > 0


# ---------------------------------------------------------------------------------------------- a15
def test_hyperbolic_distance_pairwise_kernel():
    from hpcs_amd.distances.poincare import HyperbolicDistance
    x = det_normalish((200, 8), 91) * 0.3
    y = det_normalish((200, 8), 92) * 0.3
    x[0] *= 10.0                                                 # outside the ball: projected
    d = HyperbolicDistance()
    got = d.pairwise_distance(x.to(dev), y.to(dev))
    want = OH.hyperbolic_similarity_pairs(x, y)
    torch.testing.assert_close(got.cpu(), want.reshape(got.shape), rtol=2e-5, atol=1e-6)
    mat = d.compute_mat(x.to(dev), y.to(dev))
    torch.testing.assert_close(torch.diagonal(mat).cpu(), got.cpu(), rtol=1e-3, atol=1e-5)   # (pairs add F.pairwise_distance's eps)
    with pytest.raises(TypeError):
        HyperbolicDistance(normalize_embeddings=True)


# ---------------------------------------------------------------------------------------------- RNG streams
def test_successive_steps_draw_different_triplets():
    """the in-kernel sampler hashed (t ^ seed) with seed = 1, 2, 3 ..., so step s replayed step 1's stream
    permuted.  Seeds are now mixed (base seed from torch.initial_seed() and the rank, call counter) before they meet
    the counter: successive calls must draw different positives / negatives for the same anchors, torch.manual_seed
    must reproduce the stream, and the sorted multiset of positives must not repeat."""
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    M, E, C = 4096, 32, 6
    x = (det_normalish((M, E), 601) * 0.3).to(dev)
    lab = det_int((M,), 602, C).to(dev)

    def draws(n_calls):
        crit = MetricHyperbolicLoss(t_per_anchor=10, fraction=1.2, scale=torch.nn.Parameter(torch.tensor([1e-3])),
                                    num_class=C, embedding_size=E, cosface=True, miner=True, rng="device",
                                    fused_filter=True).to(dev)
        out = []
        for _ in range(n_calls):
            crit.debug_triplets = []
            with torch.no_grad():
                crit.compute_hyp(x, lab)
            out.append([t.cpu() for t in crit.debug_triplets[:3]])
        return out
    torch.manual_seed(123)
    s1 = draws(3)
    torch.manual_seed(123)
    s2 = draws(1)
    torch.manual_seed(124)
    s3 = draws(1)
    a0, p0, n0 = s1[0]
    assert torch.equal(a0, s1[1][0])                                   # anchors are deterministic (rank t / k)
    for other in (s1[1], s1[2], s3[0]):
        assert float((p0 == other[1]).float().mean()) < 0.05           # ~1/n_class_members by chance
        assert float((n0 == other[2]).float().mean()) < 0.01
        assert not torch.equal(p0.sort().values, other[1].sort().values)
        # a stream that is the first one permuted among neighbouring triplets keeps per-anchor positive multisets
        k0 = int((a0 == a0[0]).sum())
        same_sets = (p0.view(-1, k0).sort(1).values == other[1].view(-1, k0).sort(1).values).all(1) if p0.numel() % k0 == 0 \
            else torch.zeros(1, dtype=torch.bool)
        assert float(same_sets.float().mean()) < 0.01
    assert all(torch.equal(u, v) for u, v in zip(s1[0], s2[0]))        # reproducible under torch.manual_seed
    lab_c = lab.cpu()
    assert bool((lab_c[a0] == lab_c[p0]).all()) and bool((lab_c[a0] != lab_c[n0]).all()) and bool((a0 != p0).all())


def test_negative_sampling_is_exact_for_a_dominant_class():
    """rejection sampling gave up after 64 tries and emitted same-class 'negatives' (3.5 % at a 95 % class).
    Negatives are now drawn by rank among the other classes' points: never same-class, uniform over them."""
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    M, E = 8192, 16
    lab = torch.zeros(M, dtype=torch.int64)
    lab[:164] = 1                                                     # class 0 holds 98 % of the points
    lab = lab[torch.randperm(M, generator=torch.Generator().manual_seed(1))]
    x = (det_normalish((M, E), 611) * 0.3).to(dev)
    crit = MetricHyperbolicLoss(t_per_anchor=4, fraction=0.0, scale=torch.nn.Parameter(torch.tensor([1e-3])),
                                num_class=2, embedding_size=E, cosface=True, miner=True, rng="device",
                                fused_filter=True).to(dev)
    crit.debug_triplets = []
    with torch.no_grad():
        crit.compute_hyp(x, lab.to(dev))
    a, p, n = [t.cpu() for t in crit.debug_triplets[:3]]
    assert bool((lab[a] != lab[n]).all())
    big = lab[a] == 0
    hist = torch.bincount(n[big], minlength=M)[lab == 1].float()     # 164 candidates, ~196 draws each
    assert float(hist.min()) > 0 and float(hist.std() / hist.mean()) < 3.0 / float(hist.mean()) ** 0.5


# ---------------------------------------------------------------------------------------------- cfg-3 / cfg-5
def test_partnet_step_small_vs_oracle(hier):
    """cfg-3's code path at a size the oracle evaluates in seconds: num_categories=1 (PartNet feeds a single category
    code, partnet_hyp_hc.py:72-111), HierarchicalMetricHyperbolicLoss with a reference-parsed 3-level list, host RNG.
    Loss terms against the oracle on the HIP path's own kNN graphs."""
    from hpcs_amd import ops
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from hpcs_amd.nn.hyperbolic import ExpMap
    from hpcs_amd.step import HypHCStep
    lst = hier["Small"]["hierarchy_list"]
    B, N, k, E, C = 2, 128, 8, 8, hier["Small"]["num_class"]
    feat = VN_DGCNN_partseg(3, E, k=k, dropout=0.0, pooling="mean", num_categories=1)
    P = det_params(OE.param_shapes(E, 1), 5)
    feat.load_state_dict(P)
    step = HypHCStep(feat, ExpMap(), hyp_size=E, num_class=C, num_categories=1, lr=0.05, margin=0.35, t_per_anchor=5,
                     fraction=1.2, train_rotation=None, rng="host", hierarchical=True,
                     hierarchy_list=lst).to(dev).train().configure()
    x = det_cloud(B, N, 6)
    pts = x.transpose(1, 2).contiguous()
    cat = torch.zeros(B, dtype=torch.int64)
    tgt = det_int((B, N), 8, C)
    W = step.metric_hyp_loss.loss_cosface.W.detach().cpu().clone()
    graphs, real_knn, emb = [], ops.knn, []

    def recording_knn(xx, kk):
        idx_ = real_knn(xx, kk)
        graphs.append(idx_.cpu())
        return idx_
    ops.knn = recording_knn
    hook = step.nn_feat.register_forward_hook(lambda m, i, o: emb.append(o.detach().cpu()))
    try:
        torch.manual_seed(11)
        losses = step.training_step(pts.to(dev), cat.to(dev), tgt.to(dev), targets_host=tgt)
    finally:
        ops.knn = real_knn
        hook.remove()
    # (1) extractor: oracle on the HIP path's own graphs (train-mode BN over 256 positions: fp32 noise ~1e-5)
    Po = OE.resolve_aliases({k_: v.clone() for k_, v in P.items()})
    it = iter(graphs)
    xe = OE.vn_dgcnn_partseg(x, torch.ones(B, 1), Po, k=k, training=True, knn_fn=lambda xx, kk: next(it))
    e_emb = relerr(emb[0], xe)
    assert e_emb[0] <= 2e-3, e_emb        # measured 4.6e-4: batch statistics over 256 positions (conv7's over 2 values)
    # (2) loss terms: the fp64 oracle on the HIP path's own embeddings (isolates the loss kernels from (1))
    xp = OH.expmap0(emb[0].double()).reshape(-1, E)
    torch.manual_seed(11)
    trip = OM.mine_triplets(xp, tgt.reshape(-1), 5, 1.2, 0, "easy")
    lh = OM.hyphc_loss(xp, trip, torch.tensor([1e-3], dtype=torch.float64), 0.05)
    lm = OM.hierarchical_cosface_loss(xp, tgt.reshape(-1), W.double(), lst, 0.35, 2.0)
    got_m, got_h = float(losses["loss_metric"]), float(losses["loss_hyp"])
    assert abs(got_m - float(lm)) <= 1e-5 * abs(float(lm)), (got_m, float(lm))
    assert abs(got_h - 0.1 * float(lh)) <= 1e-5 * abs(0.1 * float(lh)), (got_h, 0.1 * float(lh))
    REPORT["partnet_small_step"] = {"embeddings_vs_oracle": e_emb, "metric_rel": abs(got_m - float(lm)) / abs(float(lm)),
                                    "hyp_rel": abs(got_h - 0.1 * float(lh)) / abs(0.1 * float(lh))}
    for p_ in step.parameters():
        assert torch.isfinite(p_).all()


@pytest.mark.parametrize("cfg,cat,B,N,k", [("cfg-3", "Chair", 32, 1024, 20), ("cfg-5", "Table", 16, 2048, 40)])
def test_partnet_full_size_end_to_end(hier, cfg, cat, B, N, k):
    """BASELINE configs[2] (PartNet Chair-3: C=39, one category, hierarchical loss, B=32, N=1024, k=20) and the per-GPU
    shape of configs[4] (Table-3: C=51, B=16, N=2048, k=40) through HypHCStep.training_step on the real code path
    (device RNG, SO(3), dropout, side-stream weight gradients, RAdam).  Size-independent properties: finite losses and
    parameters over 3 steps, hierarchical CosFace bounded below by the flat CosFace of the same logits, kNN graph
    properties on the coordinate layer, SO(3) invariance of the eval embeddings, reproducibility under the same seeds."""
    import bench
    from hpcs_amd import ops
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from hpcs_amd.nn.hyperbolic import ExpMap
    from hpcs_amd.step import HypHCStep, random_rotations
    C, lst = hier[cat]["num_class"], hier[cat]["hierarchy_list"]
    E = 32

    def make():
        torch.manual_seed(0)
        feat = VN_DGCNN_partseg(3, E, k=k, dropout=0.5, pooling="mean", num_categories=1)
        return HypHCStep(feat, ExpMap(), hyp_size=E, num_class=C, num_categories=1, lr=0.05, margin=0.35, t_per_anchor=50,
                         fraction=1.2, temperature=0.05, trade_off=0.1, miner=True, train_rotation="so3", rng="device",
                         hierarchical=True, hierarchy_list=lst).to(dev).train().configure()
    pts, _, tgt = bench.synthetic_batch(B, N, C, 1, seed=3)
    cat_ids = torch.zeros(B, dtype=torch.int64)
    pts_d, cat_d, tgt_d = pts.to(dev), cat_ids.to(dev), tgt.to(dev)
    runs = []
    for _ in range(2):
        step = make()
        graphs, real_knn = [], ops.knn

        def recording_knn(xx, kk):
            idx_ = real_knn(xx, kk)
            if len(graphs) < 1:
                graphs.append((xx.detach(), idx_))
            return idx_
        ops.knn = recording_knn
        try:
            hist = []
            for _s in range(3):
                out = step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
                hist.append([float(out["loss_metric"]), float(out["loss_hyp"])])
        finally:
            ops.knn = real_knn
        runs.append((hist, [p_.detach().clone() for p_ in step.parameters()]))
        assert all(np.isfinite(v) for row in hist for v in row), hist
        assert all(bool(torch.isfinite(p_).all()) for p_ in step.parameters())
        # the hierarchical loss adds non-negative level terms to the flat cross entropy (last level = singletons)
        logits = step.metric_hyp_loss.loss_cosface.last_logits
        flat_ce = torch.nn.functional.cross_entropy(logits, tgt_d.reshape(-1))
        assert hist[-1][0] >= float(flat_ce) - 1e-4
    # same seeds -> same first-step losses (float atomics in the HypHC gradient make later steps differ at fp32 noise level)
    assert abs(runs[0][0][0][0] - runs[1][0][0][0]) <= 1e-6 * abs(runs[0][0][0][0])
    assert abs(runs[0][0][0][1] - runs[1][0][0][1]) <= 1e-5 * abs(runs[0][0][0][1])
    # kNN properties on the coordinate layer of the last run
    xx, idx = graphs[0]
    assert idx.shape == (B, N, k) and bool((idx[:, :, 0] == torch.arange(N, device=dev)).all())
    s_, _ = idx.sort(dim=-1)
    assert bool((s_[:, :, 1:] != s_[:, :, :-1]).all())
    # SO(3) invariance of the trained extractor in eval mode at full size
    step.eval()
    lab1 = torch.ones(B, 1, device=dev)
    xcf = pts_d.transpose(1, 2).contiguous()
    R = random_rotations(B, dev)
    with torch.no_grad():
        y1 = step.nn_feat(xcf, lab1)
        y2 = step.nn_feat(torch.einsum("bij,bjn->bin", R, xcf).contiguous(), lab1)
    inv = float((y1 - y2).abs().max() / y1.abs().max())
    assert inv < 2e-3, inv
    REPORT[f"{cfg}_end_to_end"] = {"losses": runs[0][0], "so3_invariance_maxabs_rel": inv}


# ---------------------------------------------------------------------------------------------- hipGraph replay
def test_graph_replay_equals_eager_steps():
    """The steady-state step captured as a hipGraph (hpcs_amd/step.py) against the same steps run eagerly from the
    same seeds: identical losses step by step up to the float-atomics noise of the HypHC gradient, the per-step values
    (seeds, sampling plan, RAdam step sizes) reaching the replayed kernels through the parameter block -- checked by
    changing the labels (hence the plan and T) and the points between replays."""
    import bench
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from hpcs_amd.nn.hyperbolic import ExpMap
    from hpcs_amd.step import HypHCStep
    B, N, k, E, C = 4, 256, 10, 32, 6

    def run(use_graph):
        torch.manual_seed(7)
        feat = VN_DGCNN_partseg(3, E, k=k, dropout=0.5, pooling="mean", num_categories=16)
        step = HypHCStep(feat, ExpMap(), hyp_size=E, num_class=C, lr=1e-3, margin=0.35, t_per_anchor=20, fraction=1.2,
                         temperature=0.05, trade_off=0.1, miner=True, train_rotation="so3", rng="device").to(dev).train()
        step.configure()
        step.use_graph = use_graph
        step.graph_warmup = 2
        hist = []
        for s in range(7):
            # two alternating batches (different points AND different label histograms): the plan changes per step
            pts, cat, tgt = bench.synthetic_batch(B, N, C, 16, seed=s % 2, labels="voronoi" if s % 2 else "invariant")
            out = step.training_step(pts.to(dev), cat.to(dev), tgt.to(dev), targets_host=tgt)
            hist.append([float(out["loss_metric"]), float(out["loss_hyp"]), float(out["acc"])])
        return hist, [p_.detach().clone() for p_ in step.parameters()], step

    h_e, p_e, _ = run(False)
    h_c, p_c, _ = run(False)                               # control: a second eager run from the same seeds
    h_g, p_g, step_g = run(True)
    assert step_g._graph is not None                       # steps 2.. were replays

    def pdist(pa, pb):
        num = sum(float((x - y).double().norm() ** 2) for x, y in zip(pa, pb)) ** 0.5
        return num / sum(float(x.double().norm() ** 2) for x in pa) ** 0.5
    # The float atomics of the HypHC gradient make two runs from the same seeds differ at the 1e-7 level in step 0, and
    # Adam's sign-like first updates amplify that step by step (the eager control run shows by how much): the replayed
    # run must track the eager one as closely as a second eager run does.
    ctrl = max(max(abs(a[i] - b[i]) / abs(a[i]) for a, b in zip(h_e, h_c)) for i in (0, 1))
    for s, (a, b) in enumerate(zip(h_e, h_g)):
        tol = 1e-5 if s < 2 else max(8 * ctrl, 5e-3)    # (the control's own spread varies 3x run to run)
        assert abs(a[0] - b[0]) <= tol * abs(a[0]) and abs(a[1] - b[1]) <= 10 * tol * abs(a[1]), (s, a, b, ctrl)
    d_ctrl, d_graph = pdist(p_e, p_c), pdist(p_e, p_g)
    assert d_graph <= max(8 * d_ctrl, 5e-3), (d_graph, d_ctrl)
    REPORT["graph_vs_eager"] = {"eager": h_e, "eager_control": h_c, "graph": h_g, "param_rel_l2_graph": d_graph,
                                "param_rel_l2_control": d_ctrl}


def test_train_mode_error_is_no_larger_than_torchs_over_an_ensemble():
    """VERDICT r2 weak #1 asked whether the HIP extractor's train-mode error (1.6x / 3.1x the reference's on the two golden
    draws) is systematic.  tests/parity_ensemble.py measures err(X, fp64) on 12 independent parameter / cloud draws with
    identical kNN graphs, for X = HIP, the oracle's torch op sequence (= the reference's, pinned by the golden tests) in fp32
    on the CPU and on this GPU.  The requested bar, per tensor and norm, on EVERY draw:

        err(HIP, fp64) <= 1.5 * max(err(torch CPU fp32, fp64), err(torch GPU fp32, fp64))

    and on average (geometric mean over the draws) HIP must not be further from fp64 than torch's CPU evaluation is.
    Measured r03 (profiles/r03_parity_ensemble.json): worst single-draw ratio 1.41 (frame z0, max-norm), embeddings 1.10 /
    0.85 (max / L2); geometric means 0.76-0.96 of the CPU's error and 0.43-0.74 of the GPU's -- the golden draws' excess was
    the spread of a heavy-tailed quantity, not the kernels."""
    import parity_ensemble as PE
    res = PE.run(seeds=12, E=32, mode="train", verbose=False)
    REPORT["train_mode_ensemble_summary"] = res["summary"]
    bad = []
    for key, v in res["summary"].items():
        if v["worst_ratio_hip_over_max_of_both"] > 1.5:
            bad.append((key, "worst draw", v["worst_ratio_hip_over_max_of_both"]))
        if v["gmean_ratio_hip_over_cpu"] > 1.1:
            bad.append((key, "geometric mean vs torch CPU", v["gmean_ratio_hip_over_cpu"]))
    assert not bad, bad


def test_train_mode_gradient_error_is_torchs_over_an_ensemble():
    """the full-network gradient test holds the HIP path to a fixed 2.5e-2 of an fp64 evaluation -- a
    2 %-wrong kernel passes where the tensor is noise-dominated.  tests/parity_ensemble.py::run_grads measures, on 6 independent
    parameter / cloud draws with identical kNN graphs, the L2-relative distance from the fp64 gradients (input cloud + every
    parameter tensor of the train-mode network) of the HIP path and of the oracle's torch op sequence (= the reference's) in
    fp32 on the CPU and on this GPU.  Measured r03 (profiles/r03_grad_ensemble.json): the quantity is heavy-tailed for EVERY
    realisation -- all-parameter error per draw 1.2e-3 ... 1.1e-1 for HIP, 1.3e-3 ... 4.3e-2 for torch-CPU, 1.7e-3 ... 3.1e-2
    for torch-GPU; geometric mean of HIP / max(both) 0.83, of HIP / torch-GPU 1.10; the median tensor is CLOSER to fp64 on the HIP
    path than on torch-GPU (0.51); 10 of 294 tensors further than twice both.
    r04 (profiles/r04_grad_ensemble.json): 24 draws -- geometric mean HIP / torch-CPU 0.86 (90 % bootstrap interval
    0.64 ... 1.17), HIP / torch-GPU 0.89 (0.68 ... 1.17): the six-draw "1.40x" was the spread of the quantity, not an excess; the 38
    of 1176 tensors beyond twice both torch runs are spread over every layer family (no tensor in more than 2 of 24 draws), i.e.
    there is no kernel to single out.  Bars, tightened to that evidence (six draws here, so a margin over the 24-draw means):

        geometric mean over the draws of  err_all(HIP) / max(err_all(torch CPU), err_all(torch GPU))  <= 1.15
        geometric mean of                 err_all(HIP) / err_all(torch CPU)                             <= 1.5
        no single draw above 3x, at most 10 % of the tensors further than twice both torch realisations.

    (The 10 %: with the dense layers on the bf16x3 GEMMs -- csrc/gemm_x3.hip, round 4 -- the aggregate ratios IMPROVE (24 draws:
    HIP / torch-CPU 0.83, HIP / torch-GPU 0.87, worst tensor 0.79; profiles/r04_grad_ensemble_x3.json) while the count of
    tensors beyond twice both torch runs rises from 3.2 % to 7.4 %: the fma-chain GEMM rounds exactly like the reference's
    sgemm, so its errors were CORRELATED with torch-CPU's and the per-tensor ratio sat near 1; the bf16x3 products are closer
    to the exact ones but independent of torch's, which widens the ratio's distribution in both directions.)"""
    import math
    import parity_ensemble as PE
    res = PE.run_grads(seeds=6, E=32, verbose=False)
    REPORT["train_mode_gradient_ensemble_summary"] = {k_: v for k_, v in res["summary"].items() if k_ != "per_tensor_over_draws"}
    ratios = [r["hip"]["all_l2"] / max(r["torch_cpu_fp32"]["all_l2"], r["torch_gpu_fp32"]["all_l2"]) for r in res["runs"]]
    gm = math.exp(sum(math.log(v) for v in ratios) / len(ratios))
    assert gm <= 1.15, (gm, ratios)
    assert res["summary"]["all_l2"]["gmean_ratio_hip_over_cpu"] <= 1.5, res["summary"]["all_l2"]
    assert max(ratios) <= 3.0, ratios
    far = res["summary"]["tensors_further_than_2x_both"]
    assert far["count"] <= 0.10 * far["of"], far
    assert all(r["tensors"] >= 40 for r in res["runs"])
