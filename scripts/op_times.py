"""Per-entry-point HIP-event timings of one training step (side stream off), grouped by call shape: the table the
kernel audits in DESIGN.md section 5 were made from.  Run on the GPU box: python scripts/op_times.py"""
import sys, os, torch, argparse, collections
sys.path.insert(0, '/root/repo')
import bench
from hpcs_amd import _lib
args = argparse.Namespace(batch=32, points=1024, k=20, embedding=32, num_class=50, num_categories=16, hierarchical=False)
torch.set_num_threads(16)
dev = torch.device('cuda', 0)
step = bench.build_step(args, dev)
step.side_stream_wgrad = False
pts, cat, tgt = bench.synthetic_batch(32, 1024, 50, 16, seed=0)
pts_d, cat_d, tgt_d = pts.to(dev), cat.to(dev), tgt.to(dev)
for _ in range(5):
    step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
torch.cuda.synchronize()
_lib.PROFILE = {n: [] for n in _lib.SIGNATURES if not n.endswith("_workspace") and n != "hpcs_version"}
NS = 4
for _ in range(NS):
    step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)
torch.cuda.synchronize()
prof, _lib.PROFILE = _lib.PROFILE, None
rows = []
for n, evs in prof.items():
    agg = collections.OrderedDict()
    for e0, e1, tag in evs:
        tag = tuple(t for t in tag if abs(t) < (1 << 40))     # drop raw pointers
        agg.setdefault(tag, []).append(e0.elapsed_time(e1))
    for tag, ts in agg.items():
        rows.append((sum(ts) / NS, len(ts) / NS, sum(ts) / len(ts), n, tag))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total {tot:.3f} ms/step over {sum(r[1] for r in rows):.0f} calls")
for t, c, avg, n, tag in rows[:70]:
    print(f"{t*1e3:8.1f} us/step x{c:4.1f} avg {avg*1e3:7.1f}  {n[5:]:24s} {tag}")
