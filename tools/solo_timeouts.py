import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth
p = fx.Planner([0])
occ = synth.synth_grid(1024, 1024, 1, 0.2); p.set_grid_occ(occ)
s, g = synth.synth_queries(occ, 1, 6000)
for i in range(6):
    p.plan_batch(s, g, 2, 1024); t = p.timing(); print("default", t["search_launches"], t["solo_timeouts"], round(t["search_kernel_ms"], 1))
for env in (dict(FXJPS_SOLO="64"), dict(FXJPS_SOLO="40", FXJPS_SOLO_LIVE="2"), dict(FXJPS_SOLO="92", FXJPS_SOLO_LIVE="4")):
    os.environ.update(env)
    for h in (2, 1):
        p.plan_batch(s, g, h, 1024); t = p.timing(); print(env, h, t["search_launches"], t["solo_timeouts"], round(t["search_kernel_ms"], 1))
