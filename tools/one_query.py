#!/usr/bin/env python3
"""Plans a few queries of a benchmark workload on their own (one wavefront each on an otherwise idle chip): the latency
of a single chain of pops, without the batch around it.  Meant to run under `rocprofv3 --pmc ...` as well -- with one
live wavefront the SQ counters are that wavefront's instruction counts and cycles (tools/one_query_report.py).

    python tools/one_query.py [workload=c2] [query ids, comma separated = 9206] [repeats = 3]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth
wname = sys.argv[1] if len(sys.argv) > 1 else "c2"
ids = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "9206").split(",")]
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
wl = json.load(open(os.path.join(ROOT, "fuxi-planner_amd", "workloads.json")))[wname]
p = fx.Planner([0])
occ = synth.synth_grid(wl["W"], wl["H"], wl["grid_seed"], wl["p"])
p.set_grid_occ(occ)
s, g = synth.synth_queries(occ, wl["qseed"], wl["nq"])
for q in ids:
    for r in range(rep):
        off, cells, cost, st = p.plan_batch(s[q:q + 1], g[q:q + 1], wl["hchoice"], wl["max_path_len"])
        tm = p.timing()
    print("%s query %d: kernel %.2f ms, pops %d, pushes %d, path %d cells, %.3f us per pop" % (
        wname, q, tm["search_kernel_ms"], tm["pops"], tm["pushes"], off[1], 1e3 * tm["search_kernel_ms"] / max(tm["pops"], 1)))
