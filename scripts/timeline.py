#!/usr/bin/env python3
"""Critical-path view of one training step from a rocprofv3 --kernel-trace CSV (the per-kernel averages of --stats add
concurrent kernels' times together; with the weight-gradient GEMMs on a side stream that sum exceeds the wall time).

  python3 scripts/timeline.py <kernel_trace.csv> [once-per-step kernel substring, default hyphc_sampled] [last N steps only]

Steps are delimited by successive starts of the once-per-step kernel; reported per step (median over the steps found):
wall, time with >=1 kernel running, time with >=2 running, idle time, and per kernel name the time during which it was the
ONLY kernel running ("exclusive") next to its summed duration."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    mark = sys.argv[2] if len(sys.argv) > 2 else "hyphc_sampled"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    marks = [s for s, e, n, q in rows if mark in n]
    if len(marks) < 3:
        print("not enough steps"); return
    steps = []
    for i in range(len(marks) - 1):
        t0, t1 = marks[i], marks[i + 1]
        ks = [(s, e, n, q) for s, e, n, q in rows if t0 <= s < t1]
        ev = []
        for j, (s, e, n, q) in enumerate(ks):
            ev.append((s, 1, j)); ev.append((min(e, t1), -1, j))
        ev.sort()
        active = set(); last = t0
        busy1 = busy2 = 0
        excl = defaultdict(int); tot = defaultdict(int)
        for s, e, n, q in ks:
            tot[n] += e - s
        for t, d, j in ev:
            dt = t - last
            if dt > 0:
                if len(active) >= 1: busy1 += dt
                if len(active) >= 2: busy2 += dt
                if len(active) == 1:
                    excl[ks[next(iter(active))][2]] += dt
            last = t
            if d == 1: active.add(j)
            else: active.discard(j)
        steps.append({"wall": t1 - t0, "busy1": busy1, "busy2": busy2, "excl": excl, "tot": tot, "n": len(ks),
                      "queues": len({q for _, _, _, q in ks})})
    last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    steps = steps[-last_n:] if last_n else steps[len(steps) // 3:]      # steady-state tail, or just drop the warm-up steps
    med = lambda xs: sorted(xs)[len(xs) // 2]
    wall = med([s["wall"] for s in steps])
    print(f"steps analysed: {len(steps)}; queues: {steps[-1]['queues']}; launches/step: {steps[-1]['n']}")
    print(f"wall {wall/1e3:.1f} us | >=1 kernel {med([s['busy1'] for s in steps])/1e3:.1f} us | >=2 kernels "
          f"{med([s['busy2'] for s in steps])/1e3:.1f} us | idle {med([s['wall']-s['busy1'] for s in steps])/1e3:.1f} us")
    names = set()
    for s in steps: names |= set(s["tot"])
    tab = []
    for n in names:
        tab.append((med([s["excl"].get(n, 0) for s in steps]), med([s["tot"].get(n, 0) for s in steps]), n))
    tab.sort(reverse=True)
    print("| exclusive us | summed us | kernel |\n|---|---|---|")
    for ex, to, n in tab[:28]:
        print(f"| {ex/1e3:.1f} | {to/1e3:.1f} | `{n[:110]}` |")


if __name__ == "__main__":
    main()
