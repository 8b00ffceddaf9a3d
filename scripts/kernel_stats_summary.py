#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace --stats output directory into the markdown table kept under profiles/.

usage: kernel_stats_summary.py <rocprof output dir> <n_steps_in_run> "<command that was profiled>" > profiles/rNN_kernel_stats_summary.md
"""
import csv
import glob
import sys


def main():
    d, steps, cmd = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats: `{cmd}` ({steps:g} steps)\n")
    print(f"total kernel time per step: {tot / steps / 1e6:.2f} ms; launches per step: {calls / steps:.0f}\n")
    print("| ms/step | % | launches/step | avg us | kernel |\n|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
        t = float(r["TotalDurationNs"])
        print(f"| {t / steps / 1e6:.3f} | {100 * t / tot:.2f} | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
              f"`{r['Name'][:110]}` |")


if __name__ == "__main__":
    main()
