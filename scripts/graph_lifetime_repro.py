"""Does a hipGraph captured AFTER another one was destroyed survive a device-wide synchronize?  (torch ops only, no hpcs
kernels: separates the runtime / torch from the product kernels in the round-2 'stray store' hunt, DESIGN.md.)

    python scripts/graph_lifetime_repro.py --first {none,one,two} --sync {device,stream,none} [--keep]

first: the form of graph A (none: no first graph; one: single stream; two: fork/join over a second stream), replayed a few
times, then destroyed (unless --keep) before graph B is captured.  B writes a 2-float output from a small reduction; every
replay is checked on the host; at replay 100 the chosen synchronisation happens.
"""
import argparse
import sys

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--first", default="two")
ap.add_argument("--sync", default="device")
ap.add_argument("--keep", action="store_true")
ap.add_argument("--replays", type=int, default=300)
a = ap.parse_args()
dev = torch.device("cuda", 0)
side = torch.cuda.Stream()


def capture(two_streams, n):
    x = torch.zeros(n, device=dev)
    step = torch.zeros(1, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        step += 1
        y = x + step
        if two_streams:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                z = (x * 2 + 1).sum()
            torch.cuda.current_stream().wait_stream(side)
            y = y + z * 0
        tmp = torch.empty(300, device=dev)            # freed inside the capture: the block is reused below
        tmp.fill_(7.0)
        del tmp
        out = torch.empty(2, device=dev)
        out[0:1].copy_(y.mean().reshape(1))
        out[1:2].fill_(1.0)
    return g, out


if a.first != "none":
    gA, oA = capture(a.first == "two", 1 << 16)
    for _ in range(4):
        gA.replay()
    torch.cuda.synchronize()
    print("graph A replayed:", oA.tolist(), flush=True)
    if not a.keep:
        del gA, oA
gB, oB = capture(False, 1 << 12)
bad = 0
for i in range(a.replays):
    if i == 100:
        if a.sync == "device":
            torch.cuda.synchronize()
        elif a.sync == "stream":
            torch.cuda.current_stream().synchronize()
    gB.replay()
    v = oB.tolist()
    if v != [float(i + 1), 1.0]:
        bad += 1
        if bad < 4:
            print(f"replay {i}: got {v}, expected {[float(i + 1), 1.0]}", flush=True)
print(f"first={a.first} sync={a.sync} keep={a.keep}: bad {bad}/{a.replays}", flush=True)
sys.exit(0)
