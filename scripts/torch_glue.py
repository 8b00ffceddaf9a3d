#!/usr/bin/env python3
"""Which torch (aten) ops still launch kernels inside one training step, and from which source line: a TorchDispatchMode
around one settled eager step at the bench geometry.   usage: python scripts/torch_glue.py [--batch 32]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HPCS_GRAPH", "0")
import bench  # noqa: E402

NO_KERNEL = ("aten.view", "aten._unsafe_view", "aten.empty", "aten.as_strided", "aten.t.", "aten.transpose", "aten.permute", "aten.reshape",
             "aten.detach", "aten.alias", "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.is_",
             "aten.sym_", "aten._local_scalar", "aten.lift_fresh", "aten.result_type", "aten.size", "aten.stride", "aten.empty_like",
             "aten.empty_strided", "aten.new_empty", "aten.unbind", "aten.split", "aten.narrow", "aten.record_stream", "aten._reshape_alias",
             "aten.view_as", "aten.unfold", "aten.flatten", "aten.contiguous", "aten.set_", "aten.resize_")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(NO_KERNEL):
            on_gpu = any(torch.is_tensor(a) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
            if on_gpu or "zeros" in name or "ones" in name or "full" in name:
                frames = [f for f in traceback.extract_stack()[:-1] if "hpcs_amd" in f.filename or f.filename.endswith("bench.py")]
                where = " <- ".join(f"{os.path.relpath(f.filename, ROOT)}:{f.lineno} {f.line.strip()[:60]}" for f in frames[-2:][::-1])
                self.rows[(name, where or "(autograd engine / other)")] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    args = argparse.Namespace(embedding=32, k=20, num_categories=16, num_class=50, hierarchical=False)
    model = bench.build_step(args, "cuda")
    pts, lab, tgt = bench.synthetic_batch(a.batch, 1024, 50, 16, seed=1234)
    batch = (pts.cuda(), lab.cuda(), tgt.cuda())
    for _ in range(6):
        model.training_step(*batch, targets_host=tgt)
    torch.cuda.synchronize()
    log = Log()
    with log:
        model.training_step(*batch, targets_host=tgt)
    torch.cuda.synchronize()
    for (name, where), n in sorted(log.rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f"{n:3d}  {name:34s} {where}")
    print("total ops:", sum(log.rows.values()))


if __name__ == "__main__":
    main()
