#!/usr/bin/env python3
"""bench.py -- start->goal plans/sec of the batched JPS/A* planner on MI355X.

A "step" is one pass of the hot path over one batch: BASELINE config 2, i.e. 10 000 synthetic
(start, goal) queries on the 1024x1024 20 %-obstacle grid (generator: fuxi-planner_amd/synth.py),
planned through the C ABI (fxjps_plan_batch_csr) with the grid already resident in HBM.  Each step
returns every path to host memory.  With N ranks (one process per GPU, launched by
torch.distributed.run) rank r plans queries [r*10000, (r+1)*10000) of the same stream (weak
scaling); the grid is built on rank 0 and broadcast once with RCCL; no other collective is on the
data path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Prints ONE JSON line on rank 0 (see the driver contract); adds `roofline` (HBM bound, algorithmic
bytes / HIP-event kernel time) and, at N=1, `cpu_baseline` (the C oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOAD = "c2_1024_p20_10k"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=10000, help="queries timed on the host cores")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    import fuxi_planner_amd as fx
    from fuxi_planner_amd import synth
    from fuxi_planner_amd.distributed import ShardedPlanner

    with open(os.path.join(ROOT, "fuxi-planner_amd", "workloads.json")) as f:
        wl = json.load(f)[WORKLOAD]
    W, H, nq, hchoice, mpl = wl["W"], wl["H"], wl["nq"], wl["hchoice"], wl["max_path_len"]

    dist = None
    torch = None
    # Bring-up aid for 1-GPU boxes (never set by the driver): FXJPS_BENCH_BACKEND=gloo runs every rank on
    # device 0 with a host-side broadcast, to exercise the multi-rank code path without a second GPU.
    backend = os.environ.get("FXJPS_BENCH_BACKEND", "nccl")
    dev_index = 0 if backend == "gloo" else local_rank
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend, rank=rank, world_size=world)  # "nccl" is RCCL on ROCm

    planner = fx.Planner([dev_index])
    occ = synth.synth_grid(W, H, wl["grid_seed"], wl["p"])  # every rank needs it to draw its queries
    if world > 1:
        sp = ShardedPlanner(planner, device="cpu" if backend == "gloo" else "cuda:%d" % dev_index)
        sp.set_grid(occ if rank == 0 else None)  # one RCCL broadcast of W*H bytes over xGMI
    else:
        planner.set_grid_occ(occ)
    starts, goals = synth.synth_queries(occ, wl["qseed"], nq, first=rank * nq)

    def sync():
        if world > 1:
            torch.cuda.synchronize()
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    def step():
        off, cells, cost, status = planner.plan_batch(starts, goals, hchoice, mpl)  # blocking: results are in host memory
        return status, planner.timing()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    kernel_ms = []
    status = None
    for _ in range(a.steps):
        status, tm = step()
        kernel_ms.append(tm["search_kernel_ms"])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda:%d" % dev_index)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if (status < 0).any():
        raise SystemExit("bench: %d queries failed" % int((status < 0).sum()))

    if rank == 0:
        value = world * nq * a.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        algo = float(wl["algorithmic_bytes"])  # rank 0 plans exactly the committed workload
        achieved = algo / (k_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/hbm_traffic.json)
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                traffic = float(json.load(f)["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "start->goal plans/sec on 1024x1024 grid",
            "value": value,
            "unit": "plans/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: 1024x1024 grid, 20% splitmix64 obstacles (seed 1), "
                                   "10000 uniform free-cell (start,goal) queries per GPU (qseed 1), hchoice=2, "
                                   "all paths returned to host",
                       "queries_per_gpu": nq, "grid": [W, H], "hchoice": hchoice,
                       "reachable": int((status > 0).sum()), "parallelism": "queries sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "fx::k_search<2>", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo,
                         "note": "latency-bound graph search (DESIGN.md section 4); traffic = rocprofv3 FETCH_SIZE+WRITE_SIZE "
                                 "bytes per launch of the profiled build (profiles/hbm_traffic.json)"},
        }
        if world == 1 and not a.no_cpu_baseline:
            from oracle import oracle  # checker used as the CPU baseline ("port"), never by the planner
            ns = min(a.cpu_sample, nq)
            cores = os.cpu_count() or 1
            tc = time.perf_counter()
            _, ol, _, _ = oracle.plan_batch(occ, starts[:ns], goals[:ns], hchoice, literal=False, max_len=mpl, nthreads=cores)
            dt = time.perf_counter() - tc
            assert np.array_equal(ol, status[:ns])
            out["cpu_baseline"] = {"value": ns / dt, "unit": "plans/s", "cores": cores, "kind": "port",
                                   "sample": "first %d queries of the same batch, oracle/jps_oracle.c (-O2), "
                                             "%d pthreads, %.1f s" % (ns, cores, dt)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    planner.close()


if __name__ == "__main__":
    main()
