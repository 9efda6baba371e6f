#!/usr/bin/env python3
"""BatchPipeline on config-2 batches, per-batch timings (GPU box)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuxi_planner_amd import synth
from fuxi_planner_amd.replan import BatchPipeline

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
occ = synth.synth_grid(1024, 1024, 1, 0.20)
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
s, g = synth.synth_queries(occ, 1, NQ)
with BatchPipeline(0, K, occ) as pipe:
    for f in [pipe.submit(s, g, 2, 1024) for _ in range(2 * K)]:
        f.result()
    t0 = time.perf_counter()
    sub, done, futs = [], {}, []
    for i in range(12):
        sub.append(time.perf_counter() - t0)
        f = pipe.submit(s, g, 2, 1024)
        f.add_done_callback(lambda _f, _i=i: done.__setitem__(_i, time.perf_counter() - t0))
        futs.append(f)
    [f.result() for f in futs]
    el = time.perf_counter() - t0
    time.sleep(0.05)
    for i in range(12):
        print("batch %2d: submit at %7.1f ms, submitted by %7.1f, done at %7.1f" % (i, sub[i] * 1e3, (sub[i + 1] if i + 1 < 12 else el) * 1e3, done[i] * 1e3))
    print("period %.1f ms -> %.0f plans/s" % (el / 12 * 1e3, 12 * NQ / el))
    print([p.timing()["search_kernel_ms"] for p in pipe.planners])
