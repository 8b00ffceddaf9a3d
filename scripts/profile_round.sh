#!/bin/bash
# Produces the per-round artefacts kept under profiles/ (run on the GPU box through gpurun from the repo root):
#   bash scripts/profile_round.sh rNN  ->  gpurun_out/rNN_bench.json, rNN_kernel_stats{.csv,_summary.md}, rNN_pmc_traffic.json
# Counter passes are separate rocprofv3 runs (one counter each) with --kernel-trace only, as the pool requires.
set -e
cd /root/repo
R=$1
python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
tail -1 gpurun_out/${R}_bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
# profiled runs are EAGER (HPCS_GRAPH=0: a replayed hipGraph shows the same kernels; eager keeps its step count exact).  The
# run settles to the steady workload like the bench line does; the summary is cut to its LAST 10 steps (kernel_trace.csv),
# the whole-run totals (kernel_stats.csv, ramp included) are kept beside it.
export HPCS_GRAPH=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cp $(ls /tmp/prof_stats/*/*kernel_stats.csv | head -1) /root/repo/gpurun_out/${R}_kernel_stats.csv
python3 /root/repo/scripts/kernel_stats_summary.py /tmp/prof_stats 10 "HPCS_GRAPH=0 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing" --last > /root/repo/gpurun_out/${R}_kernel_stats_summary.md
python3 /root/repo/scripts/timeline.py $(ls /tmp/prof_stats/*/*kernel_trace.csv | head -1) hyphc_sampled 12 > /root/repo/gpurun_out/${R}_timeline_cfg2.md 2>&1 || true
# counter passes: the same settled run, summarised on its last 5 steps
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python3 /root/repo/scripts/pmc_traffic.py $(ls /tmp/prof_f/*/*counter_collection.csv | head -1) $(ls /tmp/prof_w/*/*counter_collection.csv | head -1) /root/repo/gpurun_out/${R}_pmc_traffic.json 5 --last
echo done
