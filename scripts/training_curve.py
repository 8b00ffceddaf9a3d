#!/usr/bin/env python3
"""Sanity artefact: the BASELINE configs[1] step run for a few hundred optimisation steps on ONE fixed synthetic batch, with
the point-level (transposed-graph) EdgeConv backward and with the edge-level one, same seeds.  Prints one JSON object:
loss every 25 steps for both runs.  The two curves track each other (different summation orders, same mathematics) and the
loss falls -- evidence that the fast backward paths train the same model."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(point_level, steps, every):
    args = argparse.Namespace(batch=32, points=1024, k=20, embedding=32, num_class=50, num_categories=16, hierarchical=False)
    dev = torch.device("cuda", 0)
    step = bench.build_step(args, dev)
    step.nn_feat.point_level_backward = point_level
    pts, cat, tgt = bench.synthetic_batch(32, 1024, 50, 16, seed=0)
    pts_d, cat_d, tgt_d = pts.to(dev), cat.to(dev), tgt.to(dev)
    out = []
    for i in range(steps):
        losses = step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
        if i % every == 0 or i == steps - 1:
            out.append({"step": i, "total": round(float(losses["total_loss"]), 5), "metric": round(float(losses["loss_metric"]), 5),
                        "hyp": round(float(losses["loss_hyp"]), 5), "acc": round(float(losses["acc"]), 4)})
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--every", type=int, default=25)
    a = ap.parse_args()
    print(json.dumps({"workload": "BASELINE configs[1], one fixed synthetic batch, lr 0.05, SO(3) augmentation on",
                      "point_level_backward": run(True, a.steps, a.every),
                      "edge_level_backward": run(False, a.steps, a.every)}))
