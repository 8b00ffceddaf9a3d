#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace --stats output directory into the markdown table kept under profiles/.

usage: kernel_stats_summary.py <rocprof output dir> <n_steps> "<command that was profiled>" [--last [anchor]]
           > profiles/rNN_kernel_stats_summary.md

Without --last: rocprofv3's own per-kernel totals (kernel_stats.csv) divided by the number of steps in the run.
With --last: the table covers only the LAST n_steps steps of the run, cut out of kernel_trace.csv at the end of the
(n_steps+1)-th launch from last of the once-per-step anchor kernel (default hyphc_sampled).  That is how a run that first
settles to the steady workload (bench.py's default: the miner's kept-triplet count ramps for ~40 steps) is summarised on
the same steps bench.py times, so that the per-kernel average durations agree with the bench line's.
"""
import csv
import glob
import sys
from collections import defaultdict


def table(per, steps, cmd, note):
    tot = sum(t for t, _ in per.values())
    calls = sum(c for _, c in per.values())
    print(f"# rocprofv3 --kernel-trace --stats: `{cmd}` ({note})\n")
    print(f"total kernel time per step: {tot / steps / 1e6:.2f} ms; launches per step: {calls / steps:.0f}\n")
    print("| ms/step | % | launches/step | avg us | kernel |\n|---|---|---|---|---|")
    for name, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:45]:
        print(f"| {t / steps / 1e6:.3f} | {100 * t / tot:.2f} | {c / steps:.1f} | {t / c / 1e3:.1f} | `{name[:110]}` |")


def main():
    d, steps, cmd = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    if len(sys.argv) > 4 and sys.argv[4] == "--last":
        anchor = sys.argv[5] if len(sys.argv) > 5 else "hyphc_sampled"
        f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
        rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
        rows.sort()
        ends = [e for s, e, n in rows if anchor in n]
        n = int(steps)
        if len(ends) < n + 1:
            raise SystemExit(f"only {len(ends)} launches of {anchor}: cannot cut the last {n} steps")
        t0, t1 = ends[-n - 1], ends[-1]
        per = defaultdict(lambda: [0, 0])
        for s, e, name in rows:
            if t0 <= s < t1:
                per[name][0] += e - s
                per[name][1] += 1
        table({k: tuple(v) for k, v in per.items()}, steps, cmd,
              f"the last {n} of {len(ends)} steps, cut from kernel_trace.csv at the end of `{anchor}`; wall span of the cut "
              f"{(t1 - t0) / n / 1e6:.2f} ms per step")
        return
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    table({r["Name"]: (float(r["TotalDurationNs"]), int(r["Calls"])) for r in rows}, steps, cmd, f"{steps:g} steps")


if __name__ == "__main__":
    main()
