"""GPU property tests at BASELINE.json's full sizes (where the CPU oracle would take minutes):
size-independent properties of the kNN, the extractor, the loss and the optimizer."""
import numpy as np
import pytest
import torch

from _det import det_cloud, det_int, det_normalish, det_params
from oracle import extractor as OE

pytestmark = pytest.mark.gpu
dev = "cuda"


@pytest.mark.parametrize("B,N,k", [(32, 1024, 20), (16, 2048, 40)])          # cfg-2/4 and cfg-5 per-GPU shapes
def test_knn_properties_full_size(B, N, k):
    from hpcs_amd import ops
    x = det_cloud(B, N, 501).to(dev)
    idx = ops.knn(x, k)
    assert idx.shape == (B, N, k) and idx.dtype == torch.int64
    assert int(idx.min()) >= 0 and int(idx.max()) < N
    # each row: k distinct indices, the query itself first (distinct points), distances non-decreasing
    assert bool((idx[:, :, 0] == torch.arange(N, device=dev)).all())
    s, _ = idx.sort(dim=-1)
    assert bool((s[:, :, 1:] != s[:, :, :-1]).all())
    xt = x.transpose(1, 2)                                   # [B,N,3]
    nb = torch.gather(xt.unsqueeze(1).expand(B, N, N, 3), 2, idx.unsqueeze(-1).expand(B, N, k, 3))
    d2 = ((nb - xt.unsqueeze(2)) ** 2).sum(-1)
    assert bool((d2[:, :, 1:] >= d2[:, :, :-1] - 1e-6).all())
    # the k-th neighbour distance bounds every non-neighbour (checked on a slab of queries, in fp64)
    q = slice(0, 64)
    full = torch.cdist(xt[:, q].double(), xt.double()) ** 2
    kth = d2[:, q, -1].double().unsqueeze(-1)
    inside = (full < kth - 1e-6).sum(-1)
    assert int(inside.max()) <= k - 1
    # feature-space kNN (D=63): permuting the points permutes the graph
    f = det_normalish((4, 63, N), 502).to(dev)
    i1 = ops.knn(f, k)
    perm = torch.randperm(N, device=dev)
    i2 = ops.knn(f[:, :, perm].contiguous(), k)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(N, device=dev)
    assert torch.equal(perm[i2][:, inv], i1) or float((perm[i2][:, inv] != i1).float().mean()) < 1e-4


def test_extractor_full_size_invariance_and_train_step():
    """cfg-2 geometry: SO(3) invariance in eval mode; one training step keeps everything finite and changes weights."""
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from hpcs_amd.nn.hyperbolic import ExpMap
    from hpcs_amd.step import HypHCStep, random_rotations
    B, N, E, C = 8, 1024, 32, 50
    net = VN_DGCNN_partseg(3, E, k=20, dropout=0.5, pooling="mean", num_categories=16)
    net.load_state_dict(det_params(OE.param_shapes(E, 16), 77))
    step = HypHCStep(net, ExpMap(), hyp_size=E, num_class=C, lr=0.05, margin=0.35, t_per_anchor=50, fraction=1.2,
                     rng="device").to(dev).configure()
    x = det_cloud(B, N, 503).to(dev)
    cat = det_int((B,), 504, 16).to(dev)
    lab = torch.nn.functional.one_hot(cat, 16).float()
    R = random_rotations(B, dev)
    assert float((R @ R.transpose(1, 2) - torch.eye(3, device=dev)).abs().max()) < 1e-5
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
    net.eval()
    with torch.no_grad():
        y1 = net(x, lab)
        y2 = net(torch.matmul(R, x).contiguous(), lab)
    assert y1.shape == (B, N, E)
    assert float((y1 - y2).norm() / y1.norm()) < 2e-3
    net.train()
    tgt = det_int((B, N), 505, C)
    before = [p.detach().clone() for p in step.parameters()]
    out = step.training_step(x.transpose(1, 2).contiguous(), cat, tgt.to(dev), targets_host=tgt)
    assert all(torch.isfinite(v).all() for v in out.values())
    assert all(torch.isfinite(p).all() for p in step.parameters())
    moved = sum(int(not torch.equal(a, b)) for a, b in zip(before, step.parameters()))
    assert moved >= len(before) - 2
    # RAdam's projection: every row of every tensor stays inside the ball of radius 1 - 4e-3
    for p in step.parameters():
        rows = p.detach().reshape(-1, p.shape[-1] if p.dim() else 1)
        assert float(rows.norm(dim=1).max()) <= 1 - 4e-3 + 1e-5


def test_loss_full_size_properties():
    """M = 32768 (the per-GPU size of every BASELINE config): closed-form mean similarity equals the sampled
    matrix mean; the loss is invariant to a global rotation of the embedding and to positive rescaling of rows."""
    from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
    from hpcs_amd.distances import CosineSimilarity
    M, E, C = 32768, 32, 50
    x = (det_normalish((M, E), 601) * 0.3).to(dev)
    lab = det_int((M,), 602, C).to(dev)
    crit = MetricHyperbolicLoss(t_per_anchor=50, fraction=1.2, scale=torch.nn.Parameter(torch.tensor([1e-3])),
                                num_class=C, embedding_size=E, cosface=True, miner=True, rng="device",
                                fused_filter=True).to(dev)
    crit.hyp_miner.seed = 10
    l1 = crit.compute_hyp(x, lab)
    q, _ = torch.linalg.qr(torch.randn(E, E, device=dev))
    crit.hyp_miner.seed = 10
    l2 = crit.compute_hyp((x @ q).contiguous(), lab)
    crit.hyp_miner.seed = 10
    l3 = crit.compute_hyp((x * 3.7).contiguous(), lab)
    assert abs(float(l1) - float(l2)) < 2e-5 * abs(float(l1))
    assert abs(float(l1) - float(l3)) < 2e-5 * abs(float(l1))
    # closed-form mean of the 32768^2 similarity matrix vs the mean of a 2048 x 32768 slab of it
    xn = torch.nn.functional.normalize(x, dim=1)
    closed = 0.5 + 0.5 * float((xn.double().sum(0) ** 2).sum()) / (M * M)
    slab = CosineSimilarity().compute_mat(xn[:2048].contiguous(), xn)
    assert abs(float(slab.double().mean()) - closed) < 2e-3
    # triplet bookkeeping: kept <= sampled, both as planned on the host
    plan = crit.make_plan(lab.cpu())
    assert crit.last_T == plan.T and 0 < int(crit.last_kept) <= plan.T


def test_side_stream_weight_grads_match_inline():
    """The dW GEMMs run on a second stream and add into the flat bucket (ops.deferred_weight_grads); the gradients
    must equal the in-line autograd path (same kernels, 0 + dW accumulation) up to the run-to-run atomics noise."""
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from hpcs_amd.nn.hyperbolic import ExpMap
    from hpcs_amd.step import HypHCStep, random_rotations
    B, N, E, C = 4, 512, 32, 50
    x = det_cloud(B, N, 511).to(dev).transpose(1, 2).contiguous()
    cat = det_int((B,), 512, 16).to(dev)
    tgt = det_int((B, N), 513, C)
    R = random_rotations(B, dev)
    masks = tuple(torch.empty(B, 256, N, device=dev).bernoulli_(0.5).mul_(2.0) for _ in range(2))
    flats = []
    for side in (False, True, True):
        torch.manual_seed(5)                        # the CosFace class weights are drawn at construction
        net = VN_DGCNN_partseg(3, E, k=20, dropout=0.5, pooling="mean", num_categories=16)
        net.load_state_dict(det_params(OE.param_shapes(E, 16), 78))
        step = HypHCStep(net, ExpMap(), hyp_size=E, num_class=C, lr=0.0, margin=0.35, t_per_anchor=50, fraction=1.2,
                         rng="device").to(dev).configure()
        step.side_stream_wgrad = side
        step.training_step(x, cat, tgt.to(dev), targets_host=tgt, rot=R, drop_masks=masks)
        torch.cuda.synchronize()
        flats.append(step.bucket.flat.clone())
    assert torch.isfinite(flats[0]).all() and float(flats[0].abs().sum()) > 0
    # the loss kernel's fp32 atomics make runs differ in the last bits; a missed or raced dW would be O(1)
    noise = float((flats[1] - flats[2]).norm() / flats[1].norm())
    assert noise < 1e-4
    assert float((flats[0] - flats[1]).norm() / flats[0].norm()) < max(1e-5, 4 * noise)


def test_so3_rotate_kernel_matches_torch_and_is_haar_like():
    from hpcs_amd.step import rotate_to_channels_first, random_rotations
    B, N = 64, 300
    pts = det_cloud(B, N, 520).transpose(1, 2).contiguous().to(dev)          # [B,N,3]
    R = random_rotations(B, dev)
    x = rotate_to_channels_first(pts, rot=R)
    want = torch.matmul(pts, R).transpose(1, 2)
    assert torch.allclose(x, want, atol=1e-6)
    x2, R2 = rotate_to_channels_first(pts, seed=7, return_rot=True)
    eye = torch.eye(3, device=dev)
    assert float((R2 @ R2.transpose(1, 2) - eye).abs().max()) < 1e-5            # orthonormal
    assert float((torch.linalg.det(R2) - 1).abs().max()) < 1e-5                # proper rotations
    assert torch.allclose(x2, torch.matmul(pts, R2).transpose(1, 2), atol=1e-6)
    x3, R3 = rotate_to_channels_first(pts, seed=8, return_rot=True)
    assert float((R2 - R3).abs().max()) > 0.1                                   # a new seed gives new rotations
    Rm = rotate_to_channels_first(torch.zeros(4096, 1, 3, device=dev), seed=3, return_rot=True)[1]
    assert float(Rm.mean(0).abs().max()) < 0.05                                 # Haar: E[R] = 0
    assert abs(float(torch.diagonal(Rm, dim1=1, dim2=2).sum(1).var()) - 1.0) < 0.1    # Var(trace R) = 1 on SO(3)


def test_training_reduces_the_loss_on_a_fixed_batch():
    """End-to-end sanity of the whole step (extractor, losses, gradients on two streams, RAdam): 80 steps on one fixed
    synthetic batch lower the total loss and the in-step metrics are finite numbers in [0, 1]."""
    import argparse
    import bench
    args = argparse.Namespace(batch=4, points=256, k=10, embedding=32, num_class=6, num_categories=16, hierarchical=False)
    torch.manual_seed(0)
    step = bench.build_step(args, torch.device(dev))
    for g in step.optimizer.param_groups:
        g["lr"] = 0.005
    pts, cat, tgt = bench.synthetic_batch(4, 256, 6, 16, seed=0)
    pts_d, cat_d, tgt_d = pts.to(dev), cat.to(dev), tgt.to(dev)
    hist = []
    for i in range(81):
        out = step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
        if i % 20 == 0:
            hist.append({k_: float(v) for k_, v in out.items()})
    assert all(np.isfinite(list(h.values())).all() for h in hist)
    assert hist[-1]["total_loss"] < hist[0]["total_loss"] - 0.1, hist
    assert all(0.0 <= h["acc"] <= 1.0 and 0.0 <= h["iou"] <= 1.0 for h in hist)


@pytest.mark.parametrize("forms", [("side", "serial"), ("serial", "serial"), ("serial", "side")])
def test_long_replayed_run_keeps_losses_and_metrics_sane(forms):
    """3000 replayed steps on a learnable batch in round 2's FAILING sequence: a first capture is replayed a few times and
    dropped, the step is captured again (here in every order of the two forms), and the run is interrupted by device-wide
    synchronisations.  Round 2: the second capture's replays returned accuracy 6e12 / IoU exactly 1.0 from the first
    device-wide synchronize on and ended in a GPU memory-access fault -- the runtime loses the arguments of hipMemsetAsync
    NODES once a second graph exists (scripts/graph_lifetime_repro2.py; DESIGN.md "The stray store").  The library issues no
    memset nodes any more; every sampled output must stay finite and in range, the loss must fall, accuracy must reach ~1."""
    import argparse
    import math
    import bench
    args = argparse.Namespace(batch=4, points=256, k=10, embedding=32, num_class=50, num_categories=16, hierarchical=False)
    torch.manual_seed(0)
    step = bench.build_step(args, torch.device(dev))
    step.use_graph, step.graph_auto = True, False
    step._graph_form = forms[0]
    pts, cat, tgt = bench.synthetic_batch(4, 256, 50, 16, seed=0, labels="invariant")
    pts_d, cat_d, tgt_d = pts.to(dev), cat.to(dev), tgt.to(dev)
    first = last = None
    for i in range(3000):
        if i == 8:
            assert step._graph is not None and step._graph["form"] == forms[0]
            step._graph_form = forms[1]                       # this step drops the capture and captures the other form
        if i % 100 == 0:
            torch.cuda.synchronize()                          # device-wide: the trigger of round 2's corruption
        out = step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
        if i % 25 == 0 or i == 2999 or 100 <= i <= 102:
            v = {k_: float(x) for k_, x in out.items()}
            assert all(math.isfinite(x) for x in v.values()) and 0.0 <= v["acc"] <= 1.0 and 0.0 <= v["iou"] <= 1.0, (i, v)
            first = first or v
            last = v
    assert step.graph_mode() == "graph" and step._graph is not None and step._graph["form"] == forms[1]
    assert last["total_loss"] < first["total_loss"] - 0.5 and last["acc"] > 0.9, (first, last)


def test_replayed_library_calls_survive_a_second_graph_and_a_device_sync():
    """The minimal form of round 2's fault (scripts/graph_lifetime_repro2.py): two single-stream graphs made of
    hpcs_class_metrics calls; the second one is replayed 300 times across a device-wide synchronize and every output is
    checked.  With hipMemsetAsync inside the entry point this failed from replay ~20..200 on (stale fill pattern)."""
    from hpcs_amd import _lib
    M, C, nodes = 1024, 50, 40
    g_ = torch.Generator().manual_seed(3)
    logits = torch.randn(M, C, generator=g_).to(dev)
    labels = torch.randint(0, C, (M,), generator=g_).to(dev)
    pred = logits.argmax(1)
    want_acc = float((pred == labels).float().mean())
    iou = []
    for c in range(C):
        p_, y_ = pred == c, labels == c
        if bool((p_ | y_).any()):
            iou.append(float((p_ & y_).sum()) / float((p_ | y_).sum()))
    want_iou = sum(iou) / len(iou)

    def capture():
        outs = []
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(nodes):
                counts = torch.empty(3 * C, dtype=torch.int64, device=dev)
                out = torch.empty(2, dtype=torch.float32, device=dev)
                _lib.call("hpcs_class_metrics", logits, labels, M, C, counts, out)
                outs.append(out)
                del counts
        return g, outs

    gA, oA = capture()
    for _ in range(4):
        gA.replay()
    torch.cuda.synchronize()
    gB, oB = capture()
    for i in range(300):
        if i == 100:
            torch.cuda.synchronize()
        for t in oB:
            t.fill_(-1.0)
        gB.replay()
        got = torch.stack(oB).cpu()
        assert float((got[:, 0] - want_acc).abs().max()) < 1e-6 and float((got[:, 1] - want_iou).abs().max()) < 1e-5, (i, got[:3])
    del gA, oA
