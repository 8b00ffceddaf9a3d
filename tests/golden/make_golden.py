"""Generates tests/golden/*.npz by running the REFERENCE (imported from /root/reference).

Runs only in the build container (no-op when /root/reference is absent).  Inputs and
weights come from tests/_det.py's counter-based generator, so only the reference's
outputs are stored.  Nothing of the reference's source is copied: this script calls it.

Import notes: the extractor files are torch-only but their package
``__init__`` pulls geoopt/PML, so ``hpcs.nn.dgcnn`` is pre-registered as a bare package;
``hpcs.distances.lca`` / ``hpcs.miner.loss_and_miner_utils`` / ``hpcs.optim.radam`` import
``pytorch_metric_learning`` (absent) for base-class NAMES and ``c_f.to_device`` only, so
name-only placeholder modules are registered for it.  No PML behaviour is involved in
any fixture; rows whose arithmetic lives in PML (a8,a10,a12-a14) get NO fixture here.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference"


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import hpcs          # noqa: F401
    import hpcs.nn       # noqa: F401
    pkg = types.ModuleType("hpcs.nn.dgcnn")
    pkg.__path__ = [os.path.join(REF, "hpcs/nn/dgcnn")]
    sys.modules["hpcs.nn.dgcnn"] = pkg

    def ph(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Base(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    cf = ph("pytorch_metric_learning.utils.common_functions",
            to_device=lambda x, tensor=None, device=None, dtype=None: x.to(device=device, dtype=dtype))
    lmu = ph("pytorch_metric_learning.utils.loss_and_miner_utils")
    utils = ph("pytorch_metric_learning.utils", common_functions=cf, loss_and_miner_utils=lmu)
    dist = ph("pytorch_metric_learning.distances", BaseDistance=_Base, DotProductSimilarity=_Base)
    miners = ph("pytorch_metric_learning.miners", TripletMarginMiner=_Base)
    pml = ph("pytorch_metric_learning", utils=utils, distances=dist, miners=miners)
    pml.__path__ = []


def main():
    if not os.path.isdir(REF):
        print("reference absent: nothing to do")
        return
    _import_reference()
    from _det import det_uniform, det_normalish, det_int, det_cloud, det_params
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.extractor import param_shapes

    from hpcs.nn.dgcnn.utils.vn_dgcnn_util import knn, get_graph_feature
    from hpcs.nn.dgcnn.utils import vn_layers as VL
    from hpcs.nn.dgcnn.vn_dgcnn_partseg import VN_DGCNN_partseg
    from hpcs.nn.hyperbolic import ExpMap, MLPExpMap
    from hpcs.utils import poincare as UP
    from hpcs.utils.math import arctanh
    from hpcs.distances.lca import hyp_lca
    from hpcs.distances import poincare as DP
    from hpcs.miner.loss_and_miner_utils import get_balanced_random_triplet_indices
    from hpcs.optim.radam import RAdam

    torch.set_num_threads(4)
    out = {}

    def put(name, t):
        t = t.detach().cpu() if torch.is_tensor(t) else torch.as_tensor(t)
        out[name] = t.numpy()

    # ---- a1 knn -------------------------------------------------------------------------
    for tag, x, k in (("d3", det_cloud(2, 256, 11), 10),
                      ("d63", det_normalish((2, 63, 256), 12), 10),
                      ("dup", det_cloud(2, 128, 13, dup_every=5), 8),
                      ("d3_1024", det_cloud(1, 1024, 14), 20)):
        inner = -2 * torch.matmul(x.transpose(2, 1), x)
        xx = torch.sum(x ** 2, dim=1, keepdim=True)
        pd = -xx - inner - xx.transpose(2, 1)
        put(f"knn_{tag}_idx", knn(x, k).to(torch.int32))
        put(f"knn_{tag}_topv", pd.topk(k=k, dim=-1)[0])

    # ---- a2 get_graph_feature -----------------------------------------------------------
    for tag, C in (("c1", 1), ("c21", 21)):
        x = det_normalish((2, C, 3, 32), 20 + C)
        put(f"ggf_{tag}", get_graph_feature(x, k=4))
        idx = det_int((2, 32, 4), 23, 32)
        put(f"ggf_{tag}_given_idx", get_graph_feature(x, k=4, idx=idx))

    # ---- a3/a4/a5 VN layers -------------------------------------------------------------
    def load(mod, seed):
        sd = mod.state_dict()
        new = det_params({k: tuple(v.shape) for k, v in sd.items()}, seed)
        mod.load_state_dict(new)
        return mod

    x5 = det_normalish((2, 6, 3, 16, 4), 30)
    m = load(VL.VNLinearLeakyReLU(6, 5), 31)
    m.train()
    xg = x5.clone().requires_grad_(True)
    y = m(xg)
    put("vnllr5_train_out", y)
    (y * det_uniform(tuple(y.shape), 32)).sum().backward()
    put("vnllr5_train_gx", xg.grad)
    for n_, p_ in m.named_parameters():
        put("vnllr5_train_g_" + n_, p_.grad)
    put("vnllr5_train_rm", m.batchnorm.bn.running_mean)
    put("vnllr5_train_rv", m.batchnorm.bn.running_var)
    m.eval()
    put("vnllr5_eval_out", m(x5))

    x4 = det_normalish((2, 7, 3, 24), 33)
    m = load(VL.VNLinearLeakyReLU(7, 9, dim=4, share_nonlinearity=True), 34)
    m.train()
    xg = x4.clone().requires_grad_(True)
    y = m(xg)
    put("vnllr4s_train_out", y)
    (y * det_uniform(tuple(y.shape), 35)).sum().backward()
    put("vnllr4s_train_gx", xg.grad)
    for n_, p_ in m.named_parameters():
        put("vnllr4s_train_g_" + n_, p_.grad)
    m.eval()
    put("vnllr4s_eval_out", m(x4))

    m = load(VL.VNBatchNorm(6, dim=5), 36)
    m.train()
    put("vnbn5_train_out", m(x5))
    put("vnbn5_train_rm", m.bn.running_mean)
    put("vnbn5_train_rv", m.bn.running_var)
    m.eval()
    put("vnbn5_eval_out", m(x5))

    m = load(VL.VNMaxPool(6), 37)
    put("vnmaxpool_out", m(x5))
    put("meanpool_out", VL.mean_pool(x5))

    x8 = det_normalish((2, 8, 3, 24), 38)
    m = load(VL.VNStdFeature(8, dim=4, normalize_frame=False), 39)
    m.train()
    xs, z0 = m(x8)
    put("vnstd_train_xstd", xs)
    put("vnstd_train_z0", z0)
    m.eval()
    xs, z0 = m(x8)
    put("vnstd_eval_xstd", xs)
    put("vnstd_eval_z0", z0)

    # ---- a6 full extractor --------------------------------------------------------------
    for E in (2, 32):
        net = VN_DGCNN_partseg(in_channels=3, out_features=E, k=10, dropout=0.0, pooling="mean", num_categories=16)
        shapes = param_shapes(E, 16)
        sd = det_params(shapes, 40 + E)
        missing = net.load_state_dict(sd, strict=True)
        x = det_cloud(4, 256, 50 + E)
        lab = torch.eye(16)[det_int((4,), 51, 16)]
        net.eval()
        put(f"net_E{E}_eval_out", net(x, lab))
        net.train()
        xg = x.clone().requires_grad_(True)
        y = net(xg, lab)
        put(f"net_E{E}_train_out", y)
        (y * det_uniform(tuple(y.shape), 52)).sum().backward()
        put(f"net_E{E}_train_gx", xg.grad)
        for n_, p_ in net.named_parameters():
            g = p_.grad
            if g.numel() <= 4096:
                put(f"net_E{E}_train_g_{n_}", g)
            else:   # big tensors: keep a strided sample + norms (keeps the fixture small)
                put(f"net_E{E}_train_gs_{n_}", g.flatten()[::97])
                put(f"net_E{E}_train_gn_{n_}", torch.stack([g.norm(), g.sum()]))
        for n_, b_ in net.named_buffers():
            if "running" in n_ and b_.numel() <= 512:
                put(f"net_E{E}_train_buf_{n_}", b_)

    # rotation invariance input for the reference's one test (test/test_rotation_invariance.py)
    # is a property, not a value: nothing to store.

    # ---- a7 embedders -------------------------------------------------------------------
    u = det_normalish((6, 5, 32), 60)
    u[0, 0] = 0.0
    u[0, 1] *= 1e-9
    u[1, 0] *= 40.0        # |u| > 15 -> tanh clamp
    ug = u.clone().requires_grad_(True)
    y = ExpMap()(ug)
    put("expmap_out", y)
    (y * det_uniform(tuple(y.shape), 61)).sum().backward()
    put("expmap_gu", ug.grad)
    me = MLPExpMap(32, 8)
    me.load_state_dict({"mlp.0.0.weight": det_uniform((8, 32), 62) * 0.3})
    ug = u.clone().requires_grad_(True)
    y = me(ug)
    put("mlpexpmap_out", y)
    (y * det_uniform(tuple(y.shape), 63)).sum().backward()
    put("mlpexpmap_gu", ug.grad)
    put("mlpexpmap_gw", me.mlp[0][0].weight.grad)

    # ---- Poincare helpers ---------------------------------------------------------------
    a = det_normalish((64, 8), 70) * 0.3
    b = det_normalish((64, 8), 71) * 0.3
    a[0] = 0.0
    big = det_normalish((16, 8), 72) * 3.0
    put("mobius_add_out", UP.mobius_add(a, b))
    put("project_out", UP.project(torch.cat([a, big])))
    put("expmap_pu_out", UP.expmap_1(b, a))
    put("lambda_out", UP.lambda_(a))
    put("hyp_dist_o_out", UP.hyp_dist_o(torch.cat([a, big * 0.33])))
    put("mobius_mul_out", UP.mobius_mul(a + 1e-3, 0.5))
    t = torch.tensor([-2.0, -1.0, -0.999995, -0.5, 0.0, 1e-8, 0.5, 0.99999, 0.999999, 1.0, 3.0])
    tg = t.clone().requires_grad_(True)
    y = arctanh(tg)
    put("artanh_out", y)
    y.sum().backward()
    put("artanh_grad", tg.grad)
    w = det_normalish((64, 8), 73) * 0.2
    put("gyration_out", DP.gyration(a, b, w))
    put("ptransp_out", DP.ptransp(a, b, w))
    put("inner_out", DP.inner(a, w))
    put("egrad2rgrad_out", DP.egrad2rgrad(a, w.clone()))

    # ---- a11 hyp_lca --------------------------------------------------------------------
    for si, s in enumerate((1e-4, 1e-3, 1e-2, 1e-1, 0.5, 0.9)):
        da = torch.nn.functional.normalize(det_normalish((48, 32), 80 + si), dim=1)
        db = torch.nn.functional.normalize(det_normalish((48, 32), 90 + si), dim=1)
        sc = torch.tensor([s], requires_grad=True)
        ag = da.clone().requires_grad_(True)
        bg = db.clone().requires_grad_(True)
        d = hyp_lca(ag * sc, bg * sc, return_coord=False)
        put(f"hyp_lca_s{si}_dist", d)
        put(f"hyp_lca_s{si}_coord", hyp_lca(da * s, db * s, return_coord=True))
        d.sum().backward()
        put(f"hyp_lca_s{si}_ga", ag.grad)
        put(f"hyp_lca_s{si}_gb", bg.grad)
        put(f"hyp_lca_s{si}_gs", sc.grad)
    # general position (different radii)
    pa = det_normalish((48, 4), 100) * 0.35
    pb = det_normalish((48, 4), 101) * 0.35
    put("hyp_lca_general_dist", hyp_lca(pa, pb, return_coord=False))

    # ---- a9 balanced sampler ------------------------------------------------------------
    cases = {
        "balanced": det_int((64,), 110, 4),
        "imbalanced": torch.cat([torch.zeros(40), torch.ones(14), torch.full((6,), 2)]).long()[det_int((60,), 111, 60).argsort()],
        "singleton": torch.cat([torch.zeros(20), torch.ones(19), torch.full((1,), 5)]).long(),
        "single_class": torch.zeros(16).long(),
        "gappy": torch.cat([torch.full((10,), 3), torch.full((12,), 7), torch.full((9,), 1)]).long(),
    }
    for name, lab in cases.items():
        for frac in (0.0, 1.2):
            torch.manual_seed(1234)
            a_, p_, n_ = get_balanced_random_triplet_indices(lab, t_per_anchor=5, fraction=frac)
            put(f"sampler_{name}_f{int(frac * 10)}_labels", lab.to(torch.int32))
            put(f"sampler_{name}_f{int(frac * 10)}_apn", torch.stack([a_, p_, n_]).to(torch.int32))

    # ---- RAdam (next row f1) ------------------------------------------------------------
    ps = [torch.nn.Parameter(det_uniform(s, 120 + i) * sc_) for i, (s, sc_) in
          enumerate((((5, 7), 0.3), ((6,), 0.5), ((4, 3, 1), 0.8), ((1,), 1e-3), ((3, 4), 2.0)))]
    opt = RAdam(ps, lr=0.05)
    for step in range(3):
        for i, p_ in enumerate(ps):
            p_.grad = det_uniform(tuple(p_.shape), 130 + 10 * step + i) * 0.7
        opt.step()
        for i, p_ in enumerate(ps):
            put(f"radam_step{step}_p{i}", p_)
    for i, p_ in enumerate(ps):
        put(f"radam_final_m{i}", opt.state[p_]["exp_avg"])
        put(f"radam_final_v{i}", opt.state[p_]["exp_avg_sq"])

    path = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
