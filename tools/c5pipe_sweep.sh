#!/bin/bash
# config 5 with frames in flight: bench.py --workload c5pipe over "<queues>:<K>" pairs (GPU box), e.g. tools/c5pipe_sweep.sh 16:8 16:12 24:12
mkdir -p gpurun_out/c5pipe
for qk in "$@"; do
  q=${qk%%:*}; k=${qk##*:}
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --workload c5pipe --steps 600 --warmup 24 --frames-in-flight $k --no-also --no-cpu-baseline > gpurun_out/c5pipe/pipe_${q}_$k.json 2> gpurun_out/c5pipe/pipe_${q}_$k.err || { tail -3 gpurun_out/c5pipe/pipe_${q}_$k.err; exit 5; }
  if grep -q "Memory access fault\|HSA_STATUS_ERROR\|GPU coredump" gpurun_out/c5pipe/pipe_${q}_$k.err; then echo "GPU FAULT: stopping"; exit 3; fi
  python3 - $q $k <<'PY'
import json, sys
q, k = sys.argv[1:3]
d = json.loads(open("gpurun_out/c5pipe/pipe_%s_%s.json" % (q, k)).read().strip().splitlines()[-1]); c = d["config"]
print("queues", q, "K", k, ":", round(c["frames_per_s"], 1), "frames/s", {a: round(b, 1) for a, b in c["submit_to_paths_latency_ms"].items()})
PY
done
