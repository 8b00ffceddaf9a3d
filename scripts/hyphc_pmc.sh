#!/bin/bash
# usage: bash scripts/hyphc_pmc.sh [nograd]   -- SQ / TCC / TCP counters of the sampled HypHC kernel (grad or no-grad variant)
cd /tmp && export TMPDIR=/tmp
export PMC_NOGRAD=$1
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc2_s$i -- python3 /root/repo/scripts/hyphc_pmc.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/pmc2_s$i/*/*counter_collection.csv")
if not fs: print("no output for group $i"); raise SystemExit
agg=collections.defaultdict(lambda:[0.0,0])
for r in csv.DictReader(open(fs[0])):
    if "hyphc_sampled" in r["Kernel_Name"]:
        agg[r["Counter_Name"]][0]+=float(r["Counter_Value"]); agg[r["Counter_Name"]][1]+=1
for k,(v,n) in agg.items(): print(f"{k:36s} per launch {v/max(n,1):.4g}  (dispatch rows {n})")
PY
done
