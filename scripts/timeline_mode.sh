#!/bin/bash
# wall / busy / overlapped / idle per step of the bench workload in a given step mode:
#   bash scripts/timeline_mode.sh <tag> [env assignments...]   -> gpurun_out/<tag>_timeline.md   (on the GPU box, via gpurun)
set -e
TAG=$1
shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > /root/repo/gpurun_out/${TAG}_prof_bench.json 2> /dev/null
python3 /root/repo/scripts/timeline.py $(ls /tmp/prof_$TAG/*/*kernel_trace.csv | head -1) hyphc_sampled 12 > /root/repo/gpurun_out/${TAG}_timeline.md 2>&1 || true
python3 /root/repo/scripts/step_sequence.py $(ls /tmp/prof_$TAG/*/*kernel_trace.csv | head -1) > /root/repo/gpurun_out/${TAG}_sequence.txt 2>&1 || true
head -3 /root/repo/gpurun_out/${TAG}_timeline.md
