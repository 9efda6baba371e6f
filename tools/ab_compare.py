import os, sys, time, numpy as np
sys.path.insert(0, ".")
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth
p = fx.Planner([0])
occ = synth.synth_grid(1024, 1024, 1, 0.20); p.set_grid_occ(occ)
s, g = synth.synth_queries(occ, 1, 10000)
ks = []
for rep in range(4):
    p.plan_batch(s, g, 2, 1024); ks.append(p.timing()["search_kernel_ms"])
print(os.environ.get("FXJPS_LIB", "current").split("/")[-1], ["%.1f" % k for k in ks], flush=True)
