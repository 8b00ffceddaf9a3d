"""Per-step HOST time of the training step with gc.callbacks attached (r03: the 90-130 ms single-step stalls are CPython's
first generation-2 collection).  usage: python scripts/host_stall.py [--eager] [--freeze]"""
import argparse, gc, sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench as B

log = []
def cb(phase, info):
    if phase == "start":
        cb.t0 = time.perf_counter()
    else:
        log.append((info["generation"], (time.perf_counter() - cb.t0) * 1e3, info["collected"]))
gc.callbacks.append(cb)

a = argparse.Namespace(embedding=32, k=20, num_categories=16, num_class=50, hierarchical=False, batch=32, points=1024)
dev = torch.device("cuda", 0)
step = B.build_step(a, dev)
if "--eager" in sys.argv:
    step.use_graph = False
pts, cat, tgt = B.synthetic_batch(32, 1024, 50, 16, seed=0)
pd, cd, td = pts.to(dev), cat.to(dev), tgt.to(dev)
if "--freeze" in sys.argv:
    for _ in range(3):
        step.training_step(pd, cd, td, targets_host=tgt)
    gc.collect(); gc.freeze()
print("gc counts", gc.get_count(), "threshold", gc.get_threshold(), "objects", len(gc.get_objects()))
step.graph_warmup = 1 << 30
torch.cuda.synchronize()
T0 = time.perf_counter()
for i in range(120):
    n0 = len(log)
    t0 = time.perf_counter()
    step.training_step(pd, cd, td, targets_host=tgt)
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 20 or len(log) > n0:
        print(f"step {i}: host {dt:.1f} ms gc {[(g, round(ms, 1), c) for g, ms, c in log[n0:]]}", flush=True)
torch.cuda.synchronize()
print(f"total {(time.perf_counter() - T0) / 120 * 1e3:.2f} ms/step; gen2 collections {[round(ms,1) for g, ms, c in log if g == 2]}")
