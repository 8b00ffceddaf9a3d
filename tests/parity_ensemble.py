#!/usr/bin/env python3
"""Train-mode embedding error as a DISTRIBUTION (a tool, not a test; lives under tests/ because it drives the oracle).

VERDICT r2 weak #1: on the two golden parameter draws the HIP extractor sat 1.6x / 3.1x further (L2) from an fp64
evaluation than the reference did, and 0.5x / 1.9x of what the same torch ops give on this GPU.  Two draws cannot tell a
systematic excess from the spread of a heavy-tailed quantity (13 train-mode batch-norms amplify rounding ~150x, and the
amplification is dominated by the few channels whose norm distribution is narrow).  This tool measures, over `--seeds`
independent parameter / cloud draws at the golden geometry ([4,3,256], k=10, E=32, train mode, dropout 0), on IDENTICAL
kNN graphs (those of the fp64 evaluation):

    err(X, fp64) for X in { HIP kernels, the oracle's torch op sequence in fp32 on this GPU, the same on the CPU }

per tapped tensor (x1, x2, x3, conv6 output g6, frame z0, embeddings out), max-abs / max|.| and L2 relative.  The oracle's
fp32 CPU run IS the reference's op sequence (pinned to the reference by tests/test_oracle_golden.py), so it stands for
err(reference, fp64) on draws the committed fixtures do not cover.  Output: one JSON document (kept as
profiles/rNN_parity_ensemble.json) with per-seed numbers and per-tensor medians / geometric-mean ratios.
"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _det import det_cloud, det_int, det_params  # noqa: E402
from oracle import extractor as OE  # noqa: E402  (checker only)
import test_gpu_round2 as R2  # noqa: E402  (_Tap, _oracle_eval, relerr)


def run(seeds=12, E=32, mode="train", verbose=True):
    a = argparse.Namespace(seeds=seeds, E=E, mode=mode)
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    dev = "cuda"
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    training = a.mode == "train"
    names = ("x1", "x2", "x3", "g6", "z0", "out")
    runs = []
    for s in range(a.seeds):
        sd = det_params(OE.param_shapes(a.E, 16), 7000 + 17 * s)
        x = det_cloud(4, 256, 7100 + s)
        lab = torch.eye(16)[det_int((4,), 7200 + s, 16)]
        # the fp64 evaluation's own graphs, recorded, then injected everywhere
        rec = []

        def rec_knn(xx, kk):
            g = OE.knn_exact(xx.float(), kk)
            rec.append(g)
            return g
        P64 = OE.resolve_aliases({k_: (v.double().clone() if v.is_floating_point() else v.clone()) for k_, v in sd.items()})
        OE.vn_dgcnn_partseg(x.double(), lab.double(), P64, k=10, training=training, knn_fn=rec_knn)
        graphs = [g.clone() for g in rec]
        a64 = R2._oracle_eval(x, lab, sd, graphs, torch.float64, "cpu", training)
        variants = {"torch_cpu_fp32": R2._oracle_eval(x, lab, sd, graphs, torch.float32, "cpu", training),
                    "torch_gpu_fp32": R2._oracle_eval(x, lab, sd, graphs, torch.float32, dev, training)}
        net = VN_DGCNN_partseg(in_channels=3, out_features=a.E, k=10, dropout=0.0, pooling="mean", num_categories=16)
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).train(training)
        tap = R2._Tap(net, graphs)
        with torch.no_grad():
            y = tap.run(x.to(dev), lab.to(dev))
        hip = {k_: v.detach().cpu() for k_, v in tap.inter.items()}
        hip["out"] = y.detach().cpu()
        variants["hip"] = hip
        row = {"seed": s}
        for vn, tensors in variants.items():
            row[vn] = {nm: R2.relerr(tensors[nm], a64[nm]) for nm in names}
        runs.append(row)
        if verbose:
            print(f"[ensemble] seed {s}: out L2 hip {row['hip']['out'][1]:.2e} cpu {row['torch_cpu_fp32']['out'][1]:.2e} "
                  f"gpu {row['torch_gpu_fp32']['out'][1]:.2e}", file=sys.stderr, flush=True)

    def med(v):
        v = sorted(v)
        n = len(v)
        return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])

    def gmean(v):
        return math.exp(sum(math.log(max(e, 1e-300)) for e in v) / len(v))
    summary = {}
    for nm in names:
        for j, norm in enumerate(("max", "l2")):
            h = [r["hip"][nm][j] for r in runs]
            c = [r["torch_cpu_fp32"][nm][j] for r in runs]
            g = [r["torch_gpu_fp32"][nm][j] for r in runs]
            summary[f"{nm}_{norm}"] = {
                "median_hip": med(h), "median_torch_cpu": med(c), "median_torch_gpu": med(g),
                "gmean_ratio_hip_over_cpu": gmean([x_ / y_ for x_, y_ in zip(h, c)]),
                "gmean_ratio_hip_over_gpu": gmean([x_ / y_ for x_, y_ in zip(h, g)]),
                "gmean_ratio_gpu_over_cpu": gmean([x_ / y_ for x_, y_ in zip(g, c)]),
                "worst_ratio_hip_over_max_of_both": max(x_ / max(y_, z_) for x_, y_, z_ in zip(h, c, g)),
                "min_max_hip": [min(h), max(h)], "min_max_cpu": [min(c), max(c)], "min_max_gpu": [min(g), max(g)]}
    return {"geometry": "[4,3,256], k=10, E=%d, %s mode, dropout 0, fp64 graphs injected" % (a.E, a.mode),
            "seeds": a.seeds, "summary": summary, "runs": runs}


def _oracle_grads(x, lab, sd, graphs, wsum, dtype, device):
    """Gradients of sum(y * wsum) through the oracle's torch op sequence (train mode) in `dtype` on `device`: {name: grad}."""
    P = OE.resolve_aliases({k_: (v.to(dtype).clone().to(device) if v.is_floating_point() else v.clone().to(device))
                            for k_, v in sd.items()})
    learn = [k_ for k_, v in P.items() if v.is_floating_point() and "running" not in k_ and "num_batches" not in k_]
    for k_ in learn:
        P[k_].requires_grad_(True)
    OE.resolve_aliases(P)
    xx = x.detach().to(dtype).to(device).clone().requires_grad_(True)
    it = iter(graphs)
    y = OE.vn_dgcnn_partseg(xx, lab.to(dtype).to(device), P, k=10, training=True, knn_fn=lambda a_, kk: next(it).to(device))
    (y * wsum.to(dtype).to(device)).sum().backward()
    out = {k_: P[k_].grad.detach().cpu().double() for k_ in learn if P[k_].grad is not None}
    out["__x__"] = xx.grad.detach().cpu().double()
    return out


def run_grads(seeds=6, E=32, verbose=True):
    """Train-mode GRADIENTS as a distribution (a fixed 2.5e-2 band lets a 2 %-wrong kernel through where
    the tensor is noise-dominated).  Per draw, on the fp64 evaluation's own kNN graphs: the gradient of sum(y * w) with respect
    to the input cloud and every parameter tensor, from (a) the HIP path, (b) the oracle's torch ops in fp32 on this GPU,
    (c) the same on the CPU, each against the fp64 evaluation: L2-relative error per tensor.  Returned per draw: the
    error of each path aggregated over ALL parameter tensors (root of summed squares over the summed norms) and per tensor."""
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from _det import det_uniform
    dev = "cuda"
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    runs = []
    for s in range(seeds):
        sd = det_params(OE.param_shapes(E, 16), 8000 + 17 * s)
        x = det_cloud(4, 256, 8100 + s)
        lab = torch.eye(16)[det_int((4,), 8200 + s, 16)]
        rec = []

        def rec_knn(xx, kk):
            g = OE.knn_exact(xx.float(), kk)
            rec.append(g)
            return g
        P64 = OE.resolve_aliases({k_: (v.double().clone() if v.is_floating_point() else v.clone()) for k_, v in sd.items()})
        y0 = OE.vn_dgcnn_partseg(x.double(), lab.double(), P64, k=10, training=True, knn_fn=rec_knn)
        graphs = [g.clone() for g in rec]
        wsum = det_uniform(tuple(y0.shape), 8300 + s)
        g64 = _oracle_grads(x, lab, sd, graphs, wsum, torch.float64, "cpu")
        variants = {"torch_cpu_fp32": _oracle_grads(x, lab, sd, graphs, wsum, torch.float32, "cpu"),
                    "torch_gpu_fp32": _oracle_grads(x, lab, sd, graphs, wsum, torch.float32, dev)}
        net = VN_DGCNN_partseg(in_channels=3, out_features=E, k=10, dropout=0.0, pooling="mean", num_categories=16)
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).train()
        tap = R2._Tap(net, graphs)
        xg = x.detach().to(dev).clone().requires_grad_(True)
        y = tap.run(xg, lab.to(dev))
        (y * wsum.to(dev)).sum().backward()
        hip = {n_: p_.grad.detach().cpu().double() for n_, p_ in net.named_parameters() if p_.grad is not None}
        hip["__x__"] = xg.grad.detach().cpu().double()
        variants["hip"] = hip
        names = [n_ for n_ in g64 if n_ in hip and float(g64[n_].norm()) > 0]
        row = {"seed": s, "tensors": len(names)}
        for vn, gr in variants.items():
            per = {n_: float((gr[n_].reshape(g64[n_].shape) - g64[n_]).norm() / g64[n_].norm()) for n_ in names}
            num = sum(float((gr[n_].reshape(g64[n_].shape) - g64[n_]).norm()) ** 2 for n_ in names) ** 0.5
            den = sum(float(g64[n_].norm()) ** 2 for n_ in names) ** 0.5
            row[vn] = {"all_l2": num / den, "x_l2": per["__x__"], "worst_tensor_l2": max(per.values()),
                       "median_tensor_l2": sorted(per.values())[len(per) // 2], "per_tensor": per}
        runs.append(row)
        if verbose:
            print(f"[grad ensemble] seed {s}: all-parameter L2 hip {row['hip']['all_l2']:.2e} cpu "
                  f"{row['torch_cpu_fp32']['all_l2']:.2e} gpu {row['torch_gpu_fp32']['all_l2']:.2e}", file=sys.stderr, flush=True)

    def gmean(v):
        return math.exp(sum(math.log(max(e, 1e-300)) for e in v) / len(v))
    summary = {}
    for key in ("all_l2", "x_l2", "median_tensor_l2", "worst_tensor_l2"):
        h = [r["hip"][key] for r in runs]
        c = [r["torch_cpu_fp32"][key] for r in runs]
        g = [r["torch_gpu_fp32"][key] for r in runs]
        summary[key] = {"gmean_hip": gmean(h), "gmean_torch_cpu": gmean(c), "gmean_torch_gpu": gmean(g),
                        "gmean_ratio_hip_over_cpu": gmean([a_ / b_ for a_, b_ in zip(h, c)]),
                        "gmean_ratio_hip_over_gpu": gmean([a_ / b_ for a_, b_ in zip(h, g)]),
                        "worst_ratio_hip_over_max_of_both": max(a_ / max(b_, c_) for a_, b_, c_ in zip(h, c, g))}
    # per tensor, over all draws: how often the HIP gradient is further from fp64 than BOTH torch realisations by more than 2x
    far = tot = 0
    for r in runs:
        for n_, e in r["hip"]["per_tensor"].items():
            tot += 1
            far += e > 2.0 * max(r["torch_cpu_fp32"]["per_tensor"][n_], r["torch_gpu_fp32"]["per_tensor"][n_]) + 1e-7
    summary["tensors_further_than_2x_both"] = {"count": far, "of": tot}
    # r04: WHICH tensors.  Per parameter tensor over all draws: geometric-mean error of each realisation,
    # the HIP / torch ratios, and in how many draws HIP was further from fp64 than twice both torch runs.
    per_name = {}
    for n_ in runs[0]["hip"]["per_tensor"]:
        h = [r["hip"]["per_tensor"][n_] for r in runs if n_ in r["hip"]["per_tensor"]]
        c = [r["torch_cpu_fp32"]["per_tensor"][n_] for r in runs if n_ in r["hip"]["per_tensor"]]
        g = [r["torch_gpu_fp32"]["per_tensor"][n_] for r in runs if n_ in r["hip"]["per_tensor"]]
        per_name[n_] = {"gmean_hip": gmean(h), "gmean_cpu": gmean(c), "gmean_gpu": gmean(g),
                        "ratio_hip_over_cpu": gmean([a_ / max(b_, 1e-300) for a_, b_ in zip(h, c)]),
                        "ratio_hip_over_gpu": gmean([a_ / max(b_, 1e-300) for a_, b_ in zip(h, g)]),
                        "draws_beyond_2x_both": sum(a_ > 2.0 * max(b_, c_) + 1e-7 for a_, b_, c_ in zip(h, c, g)), "draws": len(h)}
    summary["per_tensor_over_draws"] = per_name

    def family(n_):
        if n_ == "__x__":
            return "input cloud"
        head = n_.split(".")[0]
        kind = "bn" if ("bn" in n_ or n_.endswith(".bias") or ".1." in n_ and "weight" in n_) else \
               "dir" if "map_to_dir" in n_ else "feat" if "map_to_feat" in n_ else "weight"
        return f"{head}:{kind}"
    fam = {}
    for n_, v in per_name.items():
        fam.setdefault(family(n_), []).append(v)
    summary["per_family"] = {f: {"tensors": len(v), "ratio_hip_over_cpu": gmean([t["ratio_hip_over_cpu"] for t in v]),
                                 "ratio_hip_over_gpu": gmean([t["ratio_hip_over_gpu"] for t in v]),
                                 "gmean_hip": gmean([t["gmean_hip"] for t in v]),
                                 "draws_beyond_2x_both": sum(t["draws_beyond_2x_both"] for t in v)} for f, v in sorted(fam.items())}
    summary["outliers"] = sorted(({"tensor": n_, **v} for n_, v in per_name.items() if v["draws_beyond_2x_both"] > 0),
                                 key=lambda d: -d["draws_beyond_2x_both"])
    # 90 % bootstrap interval of the geometric-mean ratio HIP / torch-CPU of the all-parameter error (does it straddle 1?)
    import random
    rng = random.Random(0)
    lr = [math.log(r["hip"]["all_l2"] / r["torch_cpu_fp32"]["all_l2"]) for r in runs]
    boots = sorted(math.exp(sum(rng.choice(lr) for _ in lr) / len(lr)) for _ in range(4000))
    summary["all_l2"]["gmean_ratio_hip_over_cpu_ci90"] = [boots[200], boots[3799]]
    lr = [math.log(r["hip"]["all_l2"] / r["torch_gpu_fp32"]["all_l2"]) for r in runs]
    boots = sorted(math.exp(sum(rng.choice(lr) for _ in lr) / len(lr)) for _ in range(4000))
    summary["all_l2"]["gmean_ratio_hip_over_gpu_ci90"] = [boots[200], boots[3799]]
    for r in runs:                                   # keep the JSON small: per-tensor tables only for the first draw
        if r["seed"] > 0:
            for vn in ("hip", "torch_cpu_fp32", "torch_gpu_fp32"):
                r[vn].pop("per_tensor")
    return {"geometry": "[4,3,256], k=10, E=%d, train mode, dropout 0, fp64 graphs injected, loss = sum(y * w)" % E,
            "seeds": seeds, "summary": summary, "runs": runs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--E", type=int, default=32)
    ap.add_argument("--mode", default="train")
    ap.add_argument("--grads", action="store_true", help="the gradient ensemble instead of the forward one")
    a = ap.parse_args()
    if a.grads:
        print(json.dumps(run_grads(a.seeds, a.E), indent=1))
    else:
        print(json.dumps(run(a.seeds, a.E, a.mode), indent=1))


if __name__ == "__main__":
    main()
