#!/usr/bin/env python3
"""One training step's kernel launches in dispatch order from a rocprofv3 --kernel-trace CSV: start offset (us, from the start of
the once-per-step sampled HypHC kernel), duration, queue, grid / workgroup x-size, LDS and VGPR use.  The step printed is the
third from last.   usage: python3 scripts/step_sequence.py <kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                 r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count",""), r.get("Accum_VGPR_Count","")))
rows.sort()
marks = [s for s, e, n, *_ in rows if "hyphc_sampled" in n]
t0, t1 = marks[-3], marks[-2]
# a step = from the end of hyphc... print from hyphc start to next hyphc start
q0 = None
for s, e, n, q, g, w, lds, vg, ag in rows:
    if t0 <= s < t1:
        if q0 is None: q0 = q
        nm = n.replace("(anonymous namespace)::", "").replace("void ", "")
        nm = nm.split("(")[0][:70]
        print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} q{q} grid {g:>9} wg {w:>5} lds {lds:>6} vgpr {vg:>4}+{ag:<4} {nm}")
