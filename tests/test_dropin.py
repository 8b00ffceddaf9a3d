"""The caller-shaped drop-in: the reference's own import lines resolve through
`hpcs_amd.install_as("hpcs")`, and a restatement of what `BaseSimilarityHypHC` does with those imports --
`__init__` (hpcs/models/base_hyp_hc.py:25-76), `forward` / `compute_losses` (:101-140), `training_step` (:157-171),
`configure_optimizers` (:142-155: RAdam over ALL parameters) and `ShapeNetHypHC._forward` (hpcs/models/shapenet_hyp_hc.py:55-91)
-- trains through the PUBLIC operator API only: plain `total.backward()` through autograd, `RAdam(params).step()`, no
HypHCStep, no flat gradient bucket, no parameter block.  That is the path a maintainer's Lightning module takes.
"""
import math
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def test_reference_import_lines_resolve_to_the_mirror():
    """train.py:12-20 and hpcs/models/base_hyp_hc.py:16-21, verbatim, after install_as -- in a child process so that the
    alias does not leak into the other tests; the reference package is NOT on the path (the GPU box has none)."""
    code = r"""
import sys
sys.path.insert(0, %r)
import hpcs_amd
names = hpcs_amd.install_as("hpcs")
from hpcs.nn.dgcnn import VN_DGCNN_partseg
from hpcs.nn.hyperbolic import ExpMap, MLPExpMap
from hpcs.optim import RAdam
from hpcs.distances.poincare import project
from hpcs.loss.ultrametric_loss import MetricHyperbolicLoss
from hpcs.utils.data import to_categorical
from hpcs.distances import CosineSimilarity, hyp_lca
from hpcs.miner.triplet_margin_miner import RandomTripletMarginMiner
from hpcs.miner.loss_and_miner_utils import get_balanced_random_triplet_indices
from hpcs.loss.hierarchical_cosface_loss import HierarchicalCosFaceLoss, hierarchical_loss
from hpcs.nn.dgcnn.utils.vn_layers import VNLinearLeakyReLU, VNBatchNorm, VNStdFeature, VNMaxPool, mean_pool
from hpcs.nn.dgcnn.utils.vn_dgcnn_util import knn, get_graph_feature
from hpcs.utils.poincare import expmap_1, mobius_add
import hpcs_amd.loss.ultrametric_loss as L, hpcs_amd.nn.dgcnn as D, hpcs_amd.optim as O
assert MetricHyperbolicLoss is L.MetricHyperbolicLoss and VN_DGCNN_partseg is D.VN_DGCNN_partseg and RAdam is O.RAdam
assert sys.modules["hpcs.loss.ultrametric_loss"] is L          # leaf modules are the mirror's own module objects
try:
    from hpcs.nn.dgcnn import DGCNN_partseg                    # out of scope, and no reference here to fall back to
    raise SystemExit("expected ImportError")
except ImportError:
    pass
assert hpcs_amd.install_as("hpcs") == names                     # idempotent
print("ok", len(names))
""" % ROOT
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_out_of_scope_names_fall_back_to_the_reference_package():
    """With the reference importable, everything OUTSIDE the hot path still comes from its own files: the non-VN backbone
    next to VN_DGCNN_partseg in `hpcs.nn.dgcnn`, `pc_normalize` next to `to_categorical`, `hpcs.utils.arrays`."""
    code = r"""
import sys
sys.path.insert(0, %r)
import hpcs_amd
hpcs_amd.install_as("hpcs", reference_root=%r)
from hpcs.nn.dgcnn import VN_DGCNN_partseg, DGCNN_partseg
from hpcs.utils.data import to_categorical, pc_normalize
from hpcs.utils.scores import get_optimal_k, compute_scores
import hpcs.utils.arrays as A
assert VN_DGCNN_partseg.__module__.startswith("hpcs_amd.") and DGCNN_partseg.__module__ == "hpcs.nn.dgcnn.dgcnn_partseg"
assert to_categorical.__module__ == "hpcs_amd.utils.data" and pc_normalize.__module__.startswith("_hpcs_reference_.")
assert get_optimal_k.__module__ == "hpcs_amd.utils.scores" and A.__file__.startswith(%r)
print("ok")
""" % (ROOT, REF, REF)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


# ------------------------------------------------------------------------------------------------------------- GPU
def _caller(hpcs, E, C, k, dropout, lr, t_per_anchor):
    """What BaseSimilarityHypHC.__init__ + configure_optimizers build, with the reference's import lines."""
    from hpcs.nn.dgcnn import VN_DGCNN_partseg
    from hpcs.nn.hyperbolic import ExpMap
    from hpcs.optim import RAdam
    from hpcs.loss.ultrametric_loss import MetricHyperbolicLoss
    from hpcs.utils.data import to_categorical

    class Caller(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.nn_feat = VN_DGCNN_partseg(in_channels=3, out_features=E, k=k, dropout=dropout, pooling="mean",
                                            num_categories=16)                      # train.py:64-74
            self.nn_emb = ExpMap()                                                    # train.py:99-105 (E_in == E_out)
            self.trade_off = 0.1
            self.scale = torch.nn.Parameter(torch.Tensor([1e-3]), requires_grad=True)   # base_hyp_hc.py:62
            self.metric_hyp_loss = MetricHyperbolicLoss(margin=0.35, t_per_anchor=t_per_anchor, fraction=1.2,
                                                        scale=self.scale, temperature=0.05, anneal_factor=0.5,
                                                        num_class=C, embedding_size=E, cosface=True, miner=True)   # :64-73

        def forward(self, points, label, targets):                                    # shapenet_hyp_hc.py:55-91, train_rotation != 'so3'
            points = points.transpose(2, 1)
            decode_vector = to_categorical(label, 16)
            x_euclidean = self.nn_feat(points, decode_vector)
            x_poincare = self.nn_emb(x_euclidean)
            xe = x_euclidean.contiguous().view(-1, x_euclidean.shape[-1])              # base_hyp_hc.py:122-123
            xp = x_poincare.contiguous().view(-1, x_poincare.shape[-1])
            y = targets.view(-1)
            losses = self.metric_hyp_loss.compute_loss(xe, xp, y)                       # :101-109
            out = {"loss_hyp": losses["loss_hyp"]["losses"] * self.trade_off, "loss_metric": losses["loss_metric"]["losses"]}
            if hasattr(self.metric_hyp_loss, "loss_cosface"):                            # :126-130, :88-99
                logits = self.metric_hyp_loss.get_logits(xp, y)
                pred = torch.nn.functional.softmax(logits, dim=1).argmax(dim=1)
                out["acc"] = (pred == y).float().mean()
            return out

    model = Caller()
    return model, RAdam(model.parameters(), lr=lr)                                       # base_hyp_hc.py:143


@pytest.mark.gpu
def test_caller_shaped_training_through_the_public_api_matches_the_oracle():
    """Five optimisation steps at cfg-1 geometry (B=4, N=256, k=10, E=32) driven like Lightning drives the reference:
    optimizer.zero_grad() -> forward -> (loss_hyp + loss_metric).backward() -> optimizer.step(), against the CPU oracle
    from the same start with the same CPU RNG stream for the miner.  Step 0: the loss operators at the per-step parity bar
    (1e-5) against the oracle on the same embeddings; all steps end to end in the bands of tests/test_gpu_trajectory.py
    (Adam's first updates are sign-like)."""
    import hpcs_amd
    import bench
    from _det import det_params, det_uniform
    from oracle import extractor as OE, hyperbolic as OH, metric as OM
    hpcs_amd.install_as("hpcs")
    try:
        import hpcs
        B, N, k, E, C, STEPS, LR, TPA = 4, 256, 10, 32, 4, 5, 1e-3, 10
        pts, cat, tgt = bench.synthetic_batch(B, N, C, 16, seed=5, labels="invariant")
        model, opt = _caller(hpcs, E, C, k, 0.0, LR, TPA)
        sd = det_params(OE.param_shapes(E, 16), 321)
        model.nn_feat.load_state_dict(sd, strict=True)
        W0 = det_uniform((E, C), 322) * 1.5
        model = model.to("cuda").train()
        model.metric_hyp_loss.loss_cosface.W.data.copy_(W0)
        opt = type(opt)(model.parameters(), lr=LR)                   # parameters moved: rebuild like Lightning does after .to()

        P = OE.resolve_aliases({k_: v.clone() for k_, v in sd.items()})
        W, scale = W0.clone(), torch.tensor([1e-3])
        by_name = {}
        for name, _ in model.named_parameters():
            by_name[name] = scale if name == "scale" else (W if name.endswith("loss_cosface.W") else P[name[len("nn_feat."):]])
        params = list(by_name.values())
        for t in params:
            t.requires_grad_(True)
        OE.resolve_aliases(P)
        states = [{} for _ in params]
        group = {"betas": (0.9, 0.999), "eps": 1e-8, "lr": LR, "weight_decay": 0, "amsgrad": False, "step": 0}
        x_cf, lab = pts.transpose(2, 1).contiguous(), torch.eye(16)[cat]
        pts_d, cat_d, tgt_d = pts.cuda(), cat.cuda(), tgt.cuda()
        rel, accs = [], []
        emb = []
        hook = model.nn_emb.register_forward_hook(lambda m_, i_, o_: emb.append(o_.detach().cpu()))
        for s in range(STEPS):
            opt.zero_grad()
            torch.manual_seed(500 + s)
            out = model(pts_d, cat_d, tgt_d)
            if s == 0:
                # the loss operators in isolation: the oracle on the HIP path's OWN step-0 embeddings, same CPU RNG stream.
                # (The two extractors' train-mode embeddings differ by fp32 batch-norm noise; at initialisation
                # all points' embeddings are nearly parallel, so that noise flips the miner's `easy` filter for many
                # triplets -- the end-to-end losses are held to the trajectory bands below, the operators to 1e-5 here.)
                torch.manual_seed(500)
                xp0 = emb[0].reshape(-1, E)
                lh0, lm0, trip0 = OM.metric_hyperbolic_loss(xp0, tgt.reshape(-1), torch.tensor([1e-3]), W0, margin=0.35,
                                                            t_per_anchor=TPA, fraction=1.2, temperature=0.05, cosface=True)
                assert abs(float(out["loss_metric"]) - float(lm0)) <= 1e-5 * abs(float(lm0)), (float(out["loss_metric"]), float(lm0))
                # HypHC on the triplets the HIP miner KEPT (the `easy` filter decides on sim(a,p) - sim(a,n) > 0, and an
                # untrained extractor maps all points to nearly parallel embeddings: thousands of triplets sit within fp32
                # rounding of 0, so two fp32 evaluations of the filter keep different sets -- the reference's own included)
                a_, p_, n_, c_ = model.metric_hyp_loss.last_triplets
                c_ = int(c_)
                kept = (a_[:c_].cpu(), p_[:c_].cpu(), n_[:c_].cpu())
                lh_same = OM.hyphc_loss(xp0.double(), kept, torch.tensor([1e-3], dtype=torch.float64), 0.05)
                assert abs(float(out["loss_hyp"]) - 0.1 * float(lh_same)) <= 1e-5 * abs(0.1 * float(lh_same)), \
                    (float(out["loss_hyp"]), 0.1 * float(lh_same))
                # ... and the two kept sets differ only at such near-ties of the filter quantity (fp64 on these embeddings)
                xn64 = torch.nn.functional.normalize(xp0.double(), dim=1)
                key = lambda t: set(zip(t[0].tolist(), t[1].tolist(), t[2].tolist()))        # noqa: E731
                for (ia, ip, in_) in key(kept) ^ key(trip0):
                    tm = 0.5 * float((xn64[ia] * xn64[ip]).sum() - (xn64[ia] * xn64[in_]).sum())
                    assert abs(tm) <= 2e-6, (ia, ip, in_, tm)
            total = out["loss_hyp"] + out["loss_metric"]
            total.backward()
            opt.step()
            for t in params:
                t.grad = None
            torch.manual_seed(500 + s)
            xe = OE.vn_dgcnn_partseg(x_cf, lab, P, k=k, training=True, knn_fn=OE.knn_exact)
            xp = OH.expmap0(xe).reshape(-1, E)
            lh, lm, _ = OM.metric_hyperbolic_loss(xp, tgt.reshape(-1), scale, W, margin=0.35, t_per_anchor=TPA, fraction=1.2,
                                                  temperature=0.05, cosface=True)
            (lm + 0.1 * lh).backward()
            OM.radam_step(params, [t.grad for t in params], states, group)
            rel.append((abs(float(out["loss_metric"]) - float(lm)) / abs(float(lm)),
                        abs(float(out["loss_hyp"]) - float(0.1 * lh)) / abs(float(0.1 * lh))))
            logits = OM.cosface_logits(xp.detach(), tgt.reshape(-1), W.detach())
            accs.append((float(out["acc"]), float((logits.argmax(1) == tgt.reshape(-1)).float().mean())))
        hook.remove()
        assert max(r[0] for r in rel) <= 3e-3 and max(r[1] for r in rel) <= 6e-3, rel
        assert rel[0][0] <= 1e-5, rel                     # CosFace at step 0: no sampling, no filter
        assert all(abs(a - b) <= 2.0 / (B * N) for a, b in accs), accs
        # every parameter moved, and ended where the oracle's did (to the travel of 5 sign-like Adam steps)
        num = den = 0.0
        for name, p in model.named_parameters():
            if not name.startswith("nn_feat."):
                continue
            a, b = p.detach().cpu().double().reshape(-1), by_name[name].detach().double().reshape(-1)
            s0 = OE.resolve_aliases({k_: v.clone() for k_, v in sd.items()})[name[len("nn_feat."):]].double().reshape(-1)
            num += float((a - b).norm()) ** 2
            den += float((b - s0).norm()) ** 2
        assert den > 0 and (num / den) ** 0.5 <= 0.3, (num, den)
    finally:
        from hpcs_amd import _alias
        _alias.uninstall("hpcs")


@pytest.mark.gpu
def test_caller_shaped_soak_2000_steps():
    """2000 steps on that path (dropout 0.5 from the module's own stream, C = 50 invariant parts, lr 0.05 like bench.py):
    every sampled loss finite, accuracy in [0, 1] and rising, RAdam's double step count as the reference's."""
    import hpcs_amd
    import bench
    hpcs_amd.install_as("hpcs")
    try:
        import hpcs
        B, N, k, E, C = 4, 256, 10, 32, 50
        torch.manual_seed(0)
        model, _ = _caller(hpcs, E, C, k, 0.5, 0.05, 50)
        model = model.to("cuda").train()
        from hpcs.optim import RAdam
        opt = RAdam(model.parameters(), lr=0.05)
        pts, cat, tgt = bench.synthetic_batch(B, N, C, 16, seed=0, labels="invariant")
        pts_d, cat_d, tgt_d = pts.cuda(), cat.cuda(), tgt.cuda()
        first = last = None
        for i in range(2000):
            opt.zero_grad()
            out = model(pts_d, cat_d, tgt_d)
            (out["loss_hyp"] + out["loss_metric"]).backward()
            opt.step()
            if i % 50 == 0 or i == 1999:
                v = {k_: float(x) for k_, x in out.items()}
                assert all(math.isfinite(x) for x in v.values()) and 0.0 <= v["acc"] <= 1.0, (i, v)
                first = first or v
                last = v
        n_params = len(list(model.parameters()))
        assert opt.param_groups[0]["step"] == 2 * n_params * 2000            # radam.py:116,133: two increments per parameter
        assert last["loss_metric"] < first["loss_metric"] - 0.5 and last["acc"] > 0.5, (first, last)
    finally:
        from hpcs_amd import _alias
        _alias.uninstall("hpcs")
