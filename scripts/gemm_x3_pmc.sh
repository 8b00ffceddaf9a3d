#!/bin/bash
# SQ counters of gemm_tn_x3_kernel on vn1's forward shape (separate --pmc passes, kernel-trace only):  bash scripts/gemm_x3_pmc.sh
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rm -rf /tmp/pmcx3_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcx3_$i -- python3 /root/repo/scripts/gemm_x3_pmc.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/pmcx3_$i/*/*counter_collection.csv")
if not fs: print("no output for group $i: $grp"); raise SystemExit
agg=collections.defaultdict(lambda:[0.0,0])
for r in csv.DictReader(open(fs[0])):
    if "gemm_tn_x3" in r["Kernel_Name"]:
        agg[r["Counter_Name"]][0]+=float(r["Counter_Value"]); agg[r["Counter_Name"]][1]+=1
for k,(v,n) in agg.items(): print(f"{k:36s} per launch {v/max(n,1):.4g}  (dispatch rows {n})")
PY
done
