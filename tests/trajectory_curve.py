#!/usr/bin/env python3
"""Long form of tests/test_gpu_trajectory.py (a tool, not a test): the SAME `--steps` optimisation steps on the HIP path and
on the CPU oracle from the same start -- cfg-1 geometry (B=4, N=256, k=10), E=32, C=4 rotation-invariant parts, no
rotation, dropout through supplied masks, host RNG (identical CPU draws), RAdam lr 1e-3.  Prints one JSON list
(kept as profiles/rNN_trajectory_curve.json).  Lives under tests/ because it drives the oracle (test infrastructure)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def run_vs_oracle(steps, every, with_oracle=True):
    from _det import det_params, det_uniform
    from oracle import extractor as OE, hyperbolic as OH, metric as OM      # checker only (test infrastructure)
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from hpcs_amd.nn.hyperbolic import ExpMap
    from hpcs_amd.step import HypHCStep
    dev = torch.device("cuda", 0)
    B, N, k, E, C, LR = 4, 256, 10, 32, 4, 1e-3
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    pts, cat, tgt = bench.synthetic_batch(B, N, C, 16, seed=5, labels="invariant")
    sd = det_params(OE.param_shapes(E, 16), 321)
    feat = VN_DGCNN_partseg(3, E, k=k, dropout=0.5, pooling="mean", num_categories=16)
    feat.load_state_dict(sd, strict=True)
    step = HypHCStep(feat, ExpMap(), hyp_size=E, num_class=C, lr=LR, margin=0.35, t_per_anchor=10, fraction=1.2,
                     temperature=0.05, trade_off=0.1, train_rotation=None, rng="host").to(dev).train().configure()
    W0 = det_uniform((E, C), 322) * 1.5
    step.metric_hyp_loss.loss_cosface.W.data.copy_(W0)
    P = OE.resolve_aliases({k_: v.clone() for k_, v in sd.items()})
    W, scale = W0.clone(), torch.tensor([1e-3])
    by_name = {}
    for name, _ in step.named_parameters():
        by_name[name] = scale if name == "scale" else (W if name.endswith("loss_cosface.W") else P[name[len("nn_feat."):]])
    params = list(by_name.values())
    for t in params:
        t.requires_grad_(True)
    OE.resolve_aliases(P)
    states = [{} for _ in params]
    group = {"betas": (0.9, 0.999), "eps": 1e-8, "lr": LR, "weight_decay": 0, "amsgrad": False, "step": 0}
    x_cf, lab = pts.transpose(2, 1).contiguous(), torch.eye(16)[cat]
    pts_d, cat_d, tgt_d = pts.to(dev), cat.to(dev), tgt.to(dev)
    out = []
    for s in range(steps):
        masks = [(det_uniform((B, 256, N), 1000 + 2 * s + i) > 0).float() * 2.0 for i in range(2)]
        torch.manual_seed(500 + s)
        o = step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt, drop_masks=[m.to(dev) for m in masks])
        if not with_oracle:
            if s % every == 0 or s == steps - 1:
                out.append({"step": s, "hip_metric": round(float(o["loss_metric"]), 6), "hip_hyp": round(float(o["loss_hyp"]), 6),
                            "hip_acc": round(float(o["acc"]), 4)})
            continue
        for t in params:
            t.grad = None
        torch.manual_seed(500 + s)
        xe = OE.vn_dgcnn_partseg(x_cf, lab, P, k=k, training=True, drop_masks=masks, knn_fn=OE.knn_exact)
        xp = OH.expmap0(xe).reshape(-1, E)
        lh, lm, _ = OM.metric_hyperbolic_loss(xp, tgt.reshape(-1), scale, W, margin=0.35, t_per_anchor=10, fraction=1.2,
                                              temperature=0.05, cosface=True)
        (lm + 0.1 * lh).backward()
        OM.radam_step(params, [t.grad for t in params], states, group)
        if s % every == 0 or s == steps - 1:
            logits = OM.cosface_logits(xp.detach(), tgt.reshape(-1), W.detach())
            out.append({"step": s, "hip_metric": round(float(o["loss_metric"]), 6), "oracle_metric": round(float(lm), 6),
                        "hip_hyp": round(float(o["loss_hyp"]), 6), "oracle_hyp": round(float(0.1 * lh), 6),
                        "hip_acc": round(float(o["acc"]), 4),
                        "oracle_acc": round(float((logits.argmax(1) == tgt.reshape(-1)).float().mean()), 4)})
            print(f"[curve] oracle/HIP step {s}: {out[-1]}", file=sys.stderr, flush=True)
    return out



if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--every", type=int, default=5)
    ap.add_argument("--hip-repeats", type=int, default=0,
                    help="also run the HIP side alone this many times (float atomics make every run a different sample of "
                         "the same chaotic trajectory: the spread calibrates the HIP-vs-oracle distance)")
    a = ap.parse_args()
    res = {"hip_vs_oracle": run_vs_oracle(a.steps, a.every)}
    if a.hip_repeats:
        res["hip_alone_repeats"] = [run_vs_oracle(a.steps, a.every, with_oracle=False) for _ in range(a.hip_repeats)]
    print(json.dumps(res))
