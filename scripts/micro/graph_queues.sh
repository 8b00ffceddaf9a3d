for q in 1 2 3 4 8; do
  DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/gq_$q.json 2> gpurun_out/gq_$q.err || { echo "fail $q"; tail -3 gpurun_out/gq_$q.err; continue; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/gq_$q.json').read().strip().splitlines()[-1])
print('queues', $q, d['ms_per_step'], d['config']['step_mode'][:230])
PY
done
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/gq_nopc.json 2> gpurun_out/gq_nopc.err && python - <<PY
import json
d=json.loads(open('gpurun_out/gq_nopc.json').read().strip().splitlines()[-1])
print('no packet capture', d['ms_per_step'], d['config']['step_mode'][:230])
PY
