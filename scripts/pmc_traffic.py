#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; FETCH_SIZE reports exactly 1/2 of the
bytes of wide (16 B/lane) coalesced streaming reads -> doubled for kernels whose loads are float4-coalesced
(listed in WIDE); WRITE_SIZE is exact for 16 B/lane streaming stores and float atomics.  Other access widths are
uncalibrated and are reported uncorrected (flagged).

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [steps] [--last]
  steps alone: the run held that many steps, all of them summarised;  with --last: only the LAST `steps` steps of the run are
  summarised (dispatches after the (steps+1)-th launch from last of the once-per-step sampled HypHC kernel), which is how a
  run that first settles to the steady workload is read on the steps the bench line times.
"""
import collections
import csv
import json
import sys

# Kernels whose GLOBAL loads are contiguous 16 B/lane streaming reads (float4 / 128-bit staging loads; list made by scanning
# csrc/*.hip for 16-byte global loads, r03): FETCH_SIZE counts half of their bytes.  r02 doubled the GEMMs only, which left the
# vectorised element-wise passes under-counted (vn_bwd_apply reported 183 MB fetched where it reads a 330 MB PD + 165 MB g).
WIDE = ("gemm_tn_kernel", "gemm_nt_kernel", "gemm_nt_big_kernel", "gemm_nt_big_x3_kernel", "gemm_nt_edge_kernel", "_vec_kernel", "vn_bwd_apply_kernel",
        "vn_apply_pool_fwd_kernel", "vn_apply_pool_tile_kernel", "k2all_kernel", "edge_gather_bwd_center",
        "edge_segsum_kernel", "edge_gather_bwd_inv_kernel", "edge_grad_reduce_inv_kernel", "edge_grad_reduce_slab_kernel",
        "rowsum_wave_kernel", "rowsum_long_kernel", "vn_rc_stats_kernel", "vn_rc_apply_fwd_kernel", "vn_rc_bwd_reduce_kernel", "vn_rc_bwd_apply_kernel", "hpcs_zero_kernel")
# 16 B/lane row GATHERS (random rows through L2): the x2 rule is calibrated for streaming reads only -> reported uncorrected
GATHER = ("hyphc_sampled", "hyphc_kernel", "knn_mfma", "edge_gather_fwd_kernel")


def per_kernel(path, counter, last_steps=0):
    agg = collections.defaultdict(lambda: [0.0, 0])
    disp = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        disp[(int(r["Dispatch_Id"]), r["Kernel_Name"])] += float(r["Counter_Value"])
    cut = -1
    if last_steps:
        marks = sorted(d for d, name in disp if "hyphc_sampled" in name)
        if len(marks) < last_steps + 1:
            raise SystemExit(f"{path}: only {len(marks)} steps in the run, cannot cut the last {last_steps}")
        cut = marks[-last_steps - 1]
        last = marks[-1]
        disp = {k: v for k, v in disp.items() if cut < k[0] <= last}
    for (_, name), v in disp.items():
        agg[name][0] += v
        agg[name][1] += 1
    return agg


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def main():
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else None
    last = steps if (len(sys.argv) > 5 and sys.argv[5] == "--last") else 0
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE", last)
    write = per_kernel(sys.argv[2], "WRITE_SIZE", last)
    out = {}
    for name in sorted(set(fetch) | set(write), key=lambda n: -(fetch.get(n, [0, 1])[0] + write.get(n, [0, 1])[0])):
        f, nf = fetch.get(name, [0.0, 0])
        w, nw = write.get(name, [0.0, 0])
        wide = any(t in name for t in WIDE)
        fb = f * 1024 * (2 if wide else 1)
        wb = w * 1024
        n = max(nf, nw, 1)
        out[short(name)] = {"launches": n, "fetch_bytes_per_launch": round(fb / n), "write_bytes_per_launch": round(wb / n),
                            "hbm_bytes_per_launch": round((fb + wb) / n), "fetch_x2_correction": wide,
                            "fetch_uncalibrated_gather": any(t in name for t in GATHER)}
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
    out["_total"] = {"hbm_bytes_all_launches": total, "steps_summarised": steps, "cut": "the last steps of a settled run" if last else "the whole run",
                     "hbm_bytes_per_step": round(total / steps) if steps else None,
                     "note": "FETCH_SIZE doubled for the kernels in WIDE (16 B/lane streaming reads); gathers uncorrected"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in [kv for kv in out.items() if kv[0] != "_total"][:14]:
        print(f"{v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch  x{v['launches']:4d}  {k}")


if __name__ == "__main__":
    main()
