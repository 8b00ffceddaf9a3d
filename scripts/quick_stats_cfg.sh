#!/bin/bash
# Per-kernel averages of a settled eager run of another bench configuration: bash scripts/quick_stats_cfg.sh <tag> <bench args...>
set -e
TAG=$1
shift
cd /tmp && export TMPDIR=/tmp
export HPCS_GRAPH=0
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" > /root/repo/gpurun_out/${TAG}_prof_bench.json 2> /dev/null
python3 /root/repo/scripts/kernel_stats_summary.py /tmp/prof_$TAG 10 "HPCS_GRAPH=0 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $*" --last > /root/repo/gpurun_out/${TAG}_kernel_stats_summary.md
python3 /root/repo/scripts/timeline.py $(ls /tmp/prof_$TAG/*/*kernel_trace.csv | head -1) hyphc_sampled 12 > /root/repo/gpurun_out/${TAG}_timeline.md 2>&1 || true
echo done $TAG
