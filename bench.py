#!/usr/bin/env python3
"""bench.py -- start->goal plans/sec of the batched JPS/A* planner on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input, planned through the C ABI with the grid
already resident in HBM; every step returns every path to host memory.  Workloads (fuxi-planner_amd/workloads.json,
generator fuxi-planner_amd/synth.py):

    c2       BASELINE config 2 (default, the headline): 1024^2, 20 % obstacles, 10 000 queries, hchoice 2
    c2h1     the same with hchoice 1 (octile x10/x14)
    c3       BASELINE config 3: 4096^2, 100 000 queries
    c4shard  one GPU's share of BASELINE config 4: 1024^2, 125 000 queries
    c4       BASELINE config 4 itself: 1 000 000 queries split over the N ranks (strong scaling)
    c5       BASELINE config 5, streaming replan: a step is one frame = toggle 10 % of the cells + rebuild the maps +
             plan the 1 000 persistent queries (SURVEY 8d toggle stream); c5low = the same with 0.1 % toggles;
             c5local = one 64 x 64 window re-observed per frame (the exact-reuse case of fxjps_replan_frame);
             c5pipe = config 5 with eight frames in flight (fuxi_planner_amd.replan.FramePipeline: eight planner handles
             on the GPU take the frames in turn): the sustained frame rate, `--steps 32 --warmup 8` for a steady state

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

With N ranks (one process per GPU, launched by torch.distributed.run) rank r plans its own nq-query slice of the same
query stream (weak scaling; c4: the r-th N-th of the 1 M queries, strong scaling); the grid is built on rank 0 and
broadcast once with RCCL; no other collective is on the data path.  `--inlib` instead drives all N GPUs from ONE
process through the library's own multi-device handle (fxjps_create(n_dev = N): ncclCommInitAll + one ncclBroadcast,
no torch).

Prints ONE JSON line on rank 0 (see the driver contract); adds `roofline` (HBM bound, algorithmic bytes / HIP-event
kernel time) and, at N = 1, `cpu_baseline` (the C oracle on the host cores, bounded sample).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def kernel_src_sha16():
    h = hashlib.sha256()
    for n in ("fxjps_kernels.hip.inc", "fxjps.hip"):
        with open(os.path.join(ROOT, "fuxi-planner_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c2", "c2h1", "c3", "c4shard", "c4", "c5", "c5low", "c5local", "c5pipe"])
    ap.add_argument("--inlib", action="store_true", help="one process, all GPUs through fxjps_create(n_dev = N)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="c5pipe: planner handles taking the frames in turn (0: the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries timed on the host cores (0: the workload's default)")
    a = ap.parse_args()
    if a.workload == "c5pipe":
        # one hardware queue per planner handle, or the persistent search kernels of handles that share a queue run one
        # after the other; read by the HIP runtime when it initialises (nothing has touched the GPU yet)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.inlib:
        if world > 1:
            raise SystemExit("--inlib is a single-process mode: start it without torch.distributed.run")
    elif world != a.gpus:
        raise SystemExit("--gpus %d needs one process per GPU: launch with\n  python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ...\n"
                         "(or add --inlib to drive all GPUs from one process through the library's multi-device handle)"
                         % (a.gpus, a.gpus, a.gpus))

    import fuxi_planner_amd as fx
    from fuxi_planner_amd import synth
    from fuxi_planner_amd.distributed import ShardedPlanner, shard_bounds

    with open(os.path.join(ROOT, "fuxi-planner_amd", "workloads.json")) as f:
        WL = json.load(f)
    wl = WL[a.workload]
    W, H, nq, hchoice, mpl = wl["W"], wl["H"], wl["nq"], wl["hchoice"], wl["max_path_len"]
    streaming = "toggle_frac" in wl
    strong = bool(wl.get("strong"))
    n_units = a.gpus  # GPUs taking part

    torch = dist = None
    # Bring-up aid for 1-GPU boxes (never set by the driver): FXJPS_BENCH_BACKEND=gloo runs every rank on
    # device 0 with a host-side broadcast, to exercise the multi-rank code path without a second GPU.
    backend = os.environ.get("FXJPS_BENCH_BACKEND", "nccl")
    dev_index = 0 if backend == "gloo" else local_rank
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend, rank=rank, world_size=world)  # "nccl" is RCCL on ROCm

    planner = fx.Planner(list(range(a.gpus)) if a.inlib else [dev_index])
    occ = synth.synth_grid(W, H, wl["grid_seed"], wl["p"])  # every rank needs it to draw its queries
    if world > 1:
        sp = ShardedPlanner(planner, device="cpu" if backend == "gloo" else "cuda:%d" % dev_index)
        sp.set_grid(occ if rank == 0 else None)  # one RCCL broadcast of W*H bytes over xGMI
    else:
        planner.set_grid_occ(occ)  # --inlib with N > 1: H2D to the first device + ncclBroadcast inside the library
    if strong:
        lo, hi = shard_bounds(nq, rank, world) if world > 1 else (0, nq)
        starts, goals = synth.synth_queries(occ, wl["qseed"], hi - lo, first=lo)
    elif a.inlib:
        starts, goals = synth.synth_queries(occ, wl["qseed"], nq * a.gpus)  # the library shards them contiguously
    else:
        starts, goals = synth.synth_queries(occ, wl["qseed"], nq, first=rank * nq)
    n_local = len(starts)

    frames = []
    if streaming:  # the toggle stream is input: generated before the timed region
        keep = np.zeros((W, H), dtype=bool)
        keep[starts[:, 0], starts[:, 1]] = True
        keep[goals[:, 0], goals[:, 1]] = True
        g = occ.copy()
        for fr in range(a.warmup + a.steps):
            xy, val = synth.frame_update(g, keep, fr, wl)
            synth.apply_toggles(g, xy, val)
            frames.append((xy, val))

    def sync():
        if world > 1:
            torch.cuda.synchronize()
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    pipe = None
    if streaming and wl.get("frames_in_flight"):
        if world > 1 or a.inlib:
            raise SystemExit("c5pipe is a single-GPU workload")
        if a.frames_in_flight > 0:
            wl["frames_in_flight"] = a.frames_in_flight
        from fuxi_planner_amd.replan import FramePipeline
        pipe = FramePipeline(dev_index, int(wl["frames_in_flight"]), occ, starts, goals, hchoice, mpl)
    elif streaming:
        planner.set_queries(starts, goals, hchoice, mpl)

    def step(i):
        if streaming:  # one call per frame: cell updates + map rebuild + search of the persistent queries
            off, cells, cost, status = planner.replan_frame(*frames[i])
        else:
            off, cells, cost, status = planner.plan_batch(starts, goals, hchoice, mpl)  # blocking: results are in host memory
        return status, planner.timing()

    kernel_ms = []
    status = None
    retried = 0
    reused = 0
    direct = 0
    frame_latency_ms = None
    if pipe is not None:  # frames in flight: submit them all, the pipeline hands each to the next free handle
        for f in [pipe.submit(*frames[i]) for i in range(a.warmup)]:
            f.result()
        t0 = time.perf_counter()
        t_sub, t_done = [], {}
        futs = []
        for i in range(a.steps):
            t_sub.append(time.perf_counter())  # (submit blocks while the handle whose turn it is still plans)
            f = pipe.submit(*frames[a.warmup + i])
            f.add_done_callback(lambda _f, _i=i: t_done.__setitem__(_i, time.perf_counter()))
            futs.append(f)
        res = [f.result() for f in futs]
        elapsed = time.perf_counter() - t0
        if any((r[3] < 0).any() for r in res):
            raise SystemExit("bench: queries failed")
        status = res[-1][3]  # (the last frame's: the CPU baseline below plans on that frame's grid)
        direct = 1
        kernel_ms = [elapsed / a.steps * 1e3]  # (the launches of the frames overlap: the frame period stands in)
        time.sleep(0.01)
        frame_latency_ms = float(np.mean([(t_done[i] - t_sub[i]) * 1e3 for i in range(a.steps) if i in t_done]))
        pipe.close()
    else:
        for i in range(a.warmup):
            step(i)
        sync()
        t0 = time.perf_counter()
        for i in range(a.steps):
            status, tm = step(a.warmup + i)
            kernel_ms.append(tm["search_kernel_ms"])
            retried += tm["retried"]
            reused += tm["reused"]
            direct = tm.get("table_direct", 0)
        sync()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda:%d" % dev_index)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if (status < 0).any():
        raise SystemExit("bench: %d queries failed" % int((status < 0).sum()))

    if rank == 0:
        total_q = nq if strong else nq * n_units  # queries all ranks planned per step
        value = total_q * a.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        # algorithmic bytes of what rank 0 launched (cells + 16 * (pushes + pops), literal reference run)
        if streaming:
            per = wl.get("algorithmic_bytes_per_frame") or []
            fr = [per[a.warmup + i] for i in range(a.steps) if a.warmup + i < len(per)]
            algo = float(np.mean(fr)) if len(fr) == a.steps else None  # frames beyond the committed counts: unknown
        elif wl.get("algorithmic_bytes") is None:
            algo = None
        else:  # weak: rank 0 plans exactly the committed workload; strong: the mean shard
            algo = float(wl["algorithmic_bytes"]) / (n_units if strong else 1)
        achieved = algo / (k_ms * 1e-3) / 1e9 if algo else None
        traffic = None  # HBM bytes per launch from the committed PMC passes, only if they were taken on this very source
        tnote = "no PMC pass of this workload on this build committed"
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                ht = json.load(f).get(a.workload)
            if ht and ht.get("kernel_src_sha16") == kernel_src_sha16():
                traffic = float(ht["hbm_bytes_per_launch"])
                tnote = "rocprofv3 FETCH_SIZE+WRITE_SIZE per launch, calibrated (profiles/hbm_traffic.json)"
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "start->goal plans/sec on %dx%d grid" % (W, H),
            "value": value,
            "unit": "plans/s",
            "n_gpus": n_units,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: %s" % (a.workload, wl["describe"]),
                       "queries_per_step": total_q, "queries_on_rank0": n_local, "grid": [W, H], "hchoice": hchoice,
                       "reachable_rank0": int((status > 0).sum()), "retried_on_large_scratch": int(retried),
                       "parallelism": ("one process, fxjps_create(n_dev=%d)" % n_units) if a.inlib else "queries sharded x%d, one process per GPU" % n_units},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic,
                         # (the instantiation rocprofv3 lists: heuristic, read-set recording, table indexed by the cell)
                         "kernel": "fx::k_search<%d, %s, %s>" % (hchoice, "true" if (streaming and a.workload not in ("c5", "c5pipe")) else "false", "true" if direct else "false"),
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo,
                         "algorithmic_bytes_source": wl.get("algorithmic_bytes_source", "oracle literal mode, all queries (tools/algo_bytes.py)"),
                         "note": "graph search bound by scattered-request rate and instruction issue, not by bytes (DESIGN.md section 4); traffic: " + tnote},
        }
        if streaming:
            out["config"].update({"frames_per_s": a.steps / elapsed, "target_frames_per_s": 60,
                                  "cells_sent_per_frame": int(len(frames[0][1])),
                                  "results_reused_per_frame": reused / a.steps})
            if pipe is not None:
                out["config"].update({"frames_in_flight": int(wl["frames_in_flight"]), "frame_latency_ms": frame_latency_ms,
                                      "parallelism": "%d planner handles on one GPU take the frames in turn" % int(wl["frames_in_flight"])})
                out["roofline"]["note"] = "frames overlap: kernel_ms is the frame period; " + out["roofline"]["note"]
        if world == 1 and not a.inlib and not a.no_cpu_baseline:
            from oracle import oracle  # checker used as the CPU baseline ("port"), never by the planner
            ns = min(a.cpu_sample or wl.get("cpu_sample", nq), n_local)
            cores = min(os.cpu_count() or 1, 256)
            g_now = g if pipe is not None else (planner.get_grid() if streaming else occ)
            tc = time.perf_counter()
            _, ol, _, _ = oracle.plan_batch(g_now, starts[:ns], goals[:ns], hchoice, literal=False, max_len=mpl, nthreads=min(cores, wl.get("cpu_threads", cores)))
            dt = time.perf_counter() - tc
            assert np.array_equal(ol, status[:ns])
            out["cpu_baseline"] = {"value": ns / dt, "unit": "plans/s", "cores": min(cores, wl.get("cpu_threads", cores)), "kind": "port",
                                   "sample": "first %d queries of the same batch, oracle/jps_oracle.c (-O2), "
                                             "%d pthreads, %.1f s" % (ns, min(cores, wl.get("cpu_threads", cores)), dt)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    planner.close()


if __name__ == "__main__":
    main()
