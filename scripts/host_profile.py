#!/usr/bin/env python3
"""Where the host spends a training step: cProfile over N eager steps of the bench workload (kernels run asynchronously, so
the profile is the ENQUEUE cost: Python, ctypes, torch's allocator and autograd).  usage: python scripts/host_profile.py [steps]"""
import argparse
import cProfile
import io
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
args = argparse.Namespace(batch=32, points=1024, k=20, embedding=32, num_class=50, num_categories=16, hierarchical=False)
dev = torch.device("cuda")
step = bench.build_step(args, dev)
step.use_graph = False
pts, cat, tgt = bench.synthetic_batch(32, 1024, 50, 16, seed=0)
pts_d, cat_d, tgt_d = pts.to(dev), cat.to(dev), tgt.to(dev)
for _ in range(10):
    step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print(f"steps: {steps}")
print(s.getvalue())
