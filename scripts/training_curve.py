#!/usr/bin/env python3
"""Training curves on ONE fixed synthetic batch (round-2 form).

(1) BASELINE configs[1] geometry (B=32, N=1024, k=20, E=32, C=50) for `--steps` optimisation steps with ROTATION-INVARIANT
    part labels (bench.synthetic_batch(labels="invariant"): radial shells x local-density halves -- learnable by a
    rotation-invariant extractor under SO(3) augmentation, unlike the benchmark's Voronoi-of-absolute-position labels, on
    which accuracy stays at chance) and, beside it, with the benchmark's Voronoi labels; SO(3) on, dropout 0.5, device RNG,
    lr 0.05 (README recipe) -- loss and accuracy every `--every` steps.
Prints one JSON object (kept as profiles/rNN_training_curve.json).  The long HIP-vs-oracle trajectory (the oracle is test
infrastructure) lives under tests/: tests/trajectory_curve.py."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run_cfg2(labels, steps, every):
    args = argparse.Namespace(batch=32, points=1024, k=20, embedding=32, num_class=50, num_categories=16, hierarchical=False)
    dev = torch.device("cuda", 0)
    step = bench.build_step(args, dev)
    pts, cat, tgt = bench.synthetic_batch(32, 1024, 50, 16, seed=0, labels=labels)
    pts_d, cat_d, tgt_d = pts.to(dev), cat.to(dev), tgt.to(dev)
    out = []
    for i in range(steps):
        losses = step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
        if i % every == 0 or i == steps - 1:
            out.append({"step": i, "total": round(float(losses["total_loss"]), 5), "metric": round(float(losses["loss_metric"]), 5),
                        "hyp": round(float(losses["loss_hyp"]), 5), "acc": round(float(losses["acc"]), 4),
                        "iou": round(float(losses["iou"]), 4)})
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--every", type=int, default=25)
    a = ap.parse_args()
    res = {"cfg2_invariant_labels": run_cfg2("invariant", a.steps, a.every),
           "cfg2_voronoi_labels": run_cfg2("voronoi", a.steps, a.every),
           "notes": "B=32, N=1024, k=20, E=32, C=50, SO(3) + dropout 0.5, device RNG, lr 0.05, one fixed batch"}
    print(json.dumps(res))
