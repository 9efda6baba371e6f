#!/bin/bash
# config-2 batches in flight: bench.py --workload c2pipe with K = $@ planner handles (GPU box)
mkdir -p gpurun_out/c2pipe
for k in ${@:-2 3 4}; do
  timeout -k 10 300 python bench.py --workload c2pipe --frames-in-flight $k --no-cpu-baseline > gpurun_out/c2pipe/c2pipe_$k.json 2> gpurun_out/c2pipe/c2pipe_$k.err || { tail -5 gpurun_out/c2pipe/c2pipe_$k.err; exit 5; }
  if grep -q "Memory access fault\|HSA_STATUS_ERROR\|GPU coredump" gpurun_out/c2pipe/c2pipe_$k.err; then echo "GPU FAULT: stopping"; exit 3; fi
  python3 - $k <<'PY'
import json, sys
k = sys.argv[1]
d = json.loads(open("gpurun_out/c2pipe/c2pipe_%s.json" % k).read().strip().splitlines()[-1]); c = d["config"]
print(k, "in flight:", round(d["value"]), "plans/s, batch period", round(c["batch_period_ms"], 1), "ms, submit-to-paths",
      {a: round(b, 1) for a, b in c["submit_to_paths_latency_ms"].items()}, "wavefronts", c["resident_wavefronts"])
PY
done
