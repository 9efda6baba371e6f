#!/usr/bin/env python3
"""Can torch's HIP runtime and libfxjps.so's (system ROCm) live in one process?  Order: argv[1] in {torch_first, fx_first}."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "librccl" in l or "libhsa-runtime" in l})
if order == "torch_first":
    import torch
    print("torch cuda available:", torch.cuda.is_available(), "count", torch.cuda.device_count(), flush=True)
    t = torch.zeros(4, device="cuda:0"); torch.cuda.synchronize(); print("torch tensor ok", flush=True)
    import fuxi_planner_amd as fx
    try:
        p = fx.Planner([0]); p.set_grid_occ(np.zeros((8, 8), np.uint8)); print("fx after torch ok:", p.plan((0, 0), (7, 7)), flush=True)
        buf = torch.zeros(64, dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
        p.set_grid_device(buf.data_ptr(), 8, 8); print("set_grid_device from a torch buffer ok", flush=True)
    except Exception as e:
        print("fx after torch FAILED:", repr(e), flush=True)
else:
    import fuxi_planner_amd as fx
    p = fx.Planner([0]); p.set_grid_occ(np.zeros((8, 8), np.uint8)); print("fx ok:", p.plan((0, 0), (7, 7)), flush=True)
    import torch
    try:
        print("torch cuda available:", torch.cuda.is_available(), flush=True)
        t = torch.zeros(4, device="cuda:0"); torch.cuda.synchronize(); print("torch after fx ok", flush=True)
    except Exception as e:
        print("torch after fx FAILED:", repr(e)[:300], flush=True)
print("\n".join(maps()))
