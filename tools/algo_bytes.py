#!/usr/bin/env python3
"""Algorithmic bytes of a benchmark workload:
    B_q = cells_q * 1 B + (pushes_q + pops_q) * 16 B
with the three counts taken on the LITERAL reference algorithm (the oracle's literal mode, whose
counters are pinned to the real jps1.py by tests/golden).  Writes fuxi-planner_amd/workloads.json,
which bench.py reads for roofline.achieved.  Deterministic: depends only on (grid seed, query seed).

    python tools/algo_bytes.py            # config 2 (10 000 queries, ~1 min on 8 cores)
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle

WORKLOADS = {
    "c2_1024_p20_10k": dict(W=1024, H=1024, grid_seed=1, p=0.20, qseed=1, nq=10000, hchoice=2, max_path_len=1024),
}

def main():
    out = {}
    for name, w in WORKLOADS.items():
        occ = oracle.synth_grid(w["W"], w["H"], w["grid_seed"], w["p"])
        s, g = oracle.synth_queries(occ, w["qseed"], w["nq"])
        t = time.time()
        _, ln, _, st = oracle.plan_batch(occ, s, g, w["hchoice"], literal=True, max_len=w["max_path_len"],
                                         nthreads=os.cpu_count(), want_stats=True)
        cells, pushes, pops = int(st["cells"].sum()), int(st["pushes"].sum()), int(st["pops"].sum())
        rec = dict(w)
        rec.update(cells=cells, pushes=pushes, pops=pops, algorithmic_bytes=cells + 16 * (pushes + pops),
                   bytes_per_query=(cells + 16 * (pushes + pops)) / w["nq"], reachable=int((ln > 0).sum()),
                   mean_path_len=float(ln[ln > 0].mean()), max_path_len_seen=int(ln.max()))
        out[name] = rec
        print(name, rec, "oracle literal secs", round(time.time() - t, 1))
    with open(os.path.join(ROOT, "fuxi-planner_amd", "workloads.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)

if __name__ == "__main__":
    main()
