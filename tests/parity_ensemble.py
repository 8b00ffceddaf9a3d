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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--E", type=int, default=32)
    ap.add_argument("--mode", default="train")
    a = ap.parse_args()
    print(json.dumps(run(a.seeds, a.E, a.mode), indent=1))


if __name__ == "__main__":
    main()
