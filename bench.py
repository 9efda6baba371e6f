#!/usr/bin/env python3
"""bench.py -- start->goal plans/sec of the batched JPS/A* planner on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input, planned through the C ABI with the grid
already resident in HBM; every step returns every path to host memory.  Workloads (fuxi-planner_amd/workloads.json,
generator fuxi-planner_amd/synth.py):

    c2       BASELINE config 2 (default, the headline): 1024^2, 20 % obstacles, 10 000 queries, hchoice 2
    c2h1     the same with hchoice 1 (octile x10/x14)
    c2pipe   config-2 batches back to back with two in flight (fuxi_planner_amd.replan.BatchPipeline: two planner
             handles on the GPU take the batches in turn): sustained plans/s and submit-to-paths latency of a batch --
             what the chip does with the headline's batches when it need not wait for each one's slowest query
    c3       BASELINE config 3: 4096^2, 100 000 queries
    c4shard  one GPU's share of BASELINE config 4: 1024^2, 125 000 queries
    c4       BASELINE config 4 itself: 1 000 000 queries split over the N GPUs (strong scaling; N = 1: all of them)
    c5       BASELINE config 5, streaming replan: a step is one frame = toggle 10 % of the cells + rebuild the maps +
             plan the 1 000 persistent queries (SURVEY 8d toggle stream); c5low = the same with 0.1 % toggles;
             c5local = one 64 x 64 window re-observed per frame (the exact-reuse case of fxjps_replan_frame), c5local4k =
             the same on the 4096 x 4096 grid of config 3;
             c5pipe = config 5 with frames in flight (fuxi_planner_amd.replan.FramePipeline: K planner handles on the
             GPU take the frames in turn): sustained frames/s and p50 / p99 submit-to-paths latency over the 600 frames
             SURVEY 8d prescribes (its default --steps)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

N > 1, two ways (contiguous query shards, one broadcast of the grid over xGMI, no other collective), neither of
which imports torch:
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (any launcher that sets RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_ADDR / MASTER_PORT): one process per GPU, rank r plans its own nq-query slice (weak scaling; c4:
    the r-th N-th of the 1 M queries); the ranks meet over a TCP socket (fuxi_planner_amd.ranks), rank 0 hands out the RCCL
    id, the library broadcasts the grid (fxjps_create_rank / fxjps_set_grid_rank);
  * plain `python bench.py --gpus N` (no WORLD_SIZE): ONE process drives all N GPUs through the library's own
    multi-device handle -- fxjps_create(n_dev = N): ncclCommInitAll + one ncclBroadcast.
An N > 1 line verifies itself outside the timed region (`config.verified`): the SHA-256 of the grid every device / rank
holds equals rank 0's, a stratified sample of EVERY shard's results (>= 500 queries each) equals the C oracle's on rank 0 --
lengths, float64 cost bytes, cells --, and the run FAILS (exit code 3) when the RCCL communicator does not span all N
(unless FXJPS_BENCH_ONE_DEVICE marks the run as a one-device rehearsal).

Prints ONE JSON line on rank 0 (the driver contract) with `roofline` (HBM bound, algorithmic bytes / HIP-event kernel
time), at N = 1 `cpu_baseline` (the C oracle on the host cores, bounded sample), and -- default workload only --
`config.also`: the other BASELINE workloads measured right behind the timed region of the headline (c4shard, c1, c3,
c5pipe and c2pipe at N = 1; the 1 M queries of c4 split N ways at N > 1), so that they are in the driver's record too.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
DEFAULT_STEPS = {"c5pipe": (600, 16), "c2pipe": (24, 6)}  # workload -> (steps, warmup); everything else 5 / 2


def kernel_src_sha16():
    h = hashlib.sha256()
    for n in ("fxjps_kernels.hip.inc", "fxjps_maps.hip.inc", "fxjps.hip"):
        with open(os.path.join(ROOT, "fuxi-planner_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


class Ctx(object):
    """What every workload of one bench.py process shares: ranks, the planner handle, the torch plumbing (if any)."""
    pass


def make_frames(synth, occ, starts, goals, wl, n):
    """The toggle stream is input: generated before the timed region.  -> (frames, grid after the last one)"""
    W, H = occ.shape
    keep = np.zeros((W, H), dtype=bool)
    keep[starts[:, 0], starts[:, 1]] = True
    keep[goals[:, 0], goals[:, 1]] = True
    g = occ.copy()
    frames = []
    for fr in range(n):
        xy, val = synth.frame_update(g, keep, fr, wl)
        synth.apply_toggles(g, xy, val)
        frames.append((xy, val))
    return frames, g


def measure(cx, name, steps, warmup, frames_in_flight=0):
    """Run one workload: W untimed steps, then exactly K timed ones bracketed by a barrier + device sync on both sides,
    MAX over ranks.  -> dict (rank 0 fills everything; the others return the timing only)."""
    from fuxi_planner_amd import synth
    from fuxi_planner_amd.distributed import shard_bounds
    wl = dict(cx.WL[name])
    W, H, nq, hchoice, mpl = wl["W"], wl["H"], wl["nq"], wl["hchoice"], wl["max_path_len"]
    streaming = "toggle_frac" in wl
    strong = bool(wl.get("strong"))
    n_units = cx.gpus
    planner = cx.planner
    occ = synth.synth_grid(W, H, wl["grid_seed"], wl["p"])  # every rank needs it to draw its queries
    piped = bool(wl.get("frames_in_flight") or wl.get("batches_in_flight"))  # (the pipeline's own handles hold the grid: K of them, not K + 1)
    if cx.world > 1:
        cx.sp.set_grid(occ if cx.rank == 0 else None)  # one RCCL broadcast of W*H bytes over xGMI
    elif not piped:
        planner.set_grid_occ(occ)  # in-library N > 1: H2D to the first device + ncclBroadcast inside the library
    if strong:
        lo, hi = shard_bounds(nq, cx.rank, cx.world) if cx.world > 1 else (0, nq)
        starts, goals = synth.synth_queries(occ, wl["qseed"], hi - lo, first=lo)
    elif cx.inlib:
        starts, goals = synth.synth_queries(occ, wl["qseed"], nq * cx.gpus)  # the library shards them contiguously
    else:
        starts, goals = synth.synth_queries(occ, wl["qseed"], nq, first=cx.rank * nq)
    n_local = len(starts)

    frames, g_last = [], occ
    pipe = None
    if streaming:
        if cx.world > 1 and wl.get("frames_in_flight"):
            raise SystemExit("c5pipe is a single-GPU workload")
        ends_s, ends_g = starts, goals
        if cx.world > 1:
            # one grid on every rank: the toggle stream avoids the endpoints of ALL ranks' queries, every rank generates the
            # same frames, and rank 0's list is the one that travels (RankPlanner.replan_frame: SURVEY 8e, "broadcast only
            # the per-frame toggle list")
            ends_s, ends_g = synth.synth_queries(occ, wl["qseed"], nq * cx.world)
        frames, g_last = make_frames(synth, occ, ends_s, ends_g, wl, warmup + steps)
        if cx.world > 1:
            cx.sp.set_queries(ends_s, ends_g, hchoice, mpl)  # (this rank keeps [rank * nq, (rank + 1) * nq): the queries it drew above)
        elif wl.get("frames_in_flight"):
            if cx.gpus > 1:
                raise SystemExit("c5pipe is a single-GPU workload")
            if frames_in_flight > 0:
                wl["frames_in_flight"] = frames_in_flight
            from fuxi_planner_amd.replan import FramePipeline
            pipe = FramePipeline(cx.dev_index, int(wl["frames_in_flight"]), occ, starts, goals, hchoice, mpl, schedule=os.environ.get("FXJPS_BENCH_SCHEDULE", "free"))
        else:
            planner.set_queries(starts, goals, hchoice, mpl)

    if wl.get("batches_in_flight"):
        if cx.world > 1 or cx.gpus > 1:
            raise SystemExit("c2pipe is a single-GPU workload")
        if frames_in_flight > 0:
            wl["batches_in_flight"] = frames_in_flight
        from fuxi_planner_amd.replan import BatchPipeline
        pipe = BatchPipeline(cx.dev_index, int(wl["batches_in_flight"]), occ, schedule=os.environ.get("FXJPS_BENCH_SCHEDULE", "free"))

    last = {}

    def step(i):
        last.clear()  # (the previous step's arrays go back to the allocator first: a step that has to fault in 23 MB of fresh pages for its cells is 1 ms slower)
        if streaming and cx.world > 1:  # rank 0's cell updates reach every rank over the rendezvous socket, then as below
            off, cells, cost, status = cx.sp.replan_frame(*(frames[i] if cx.rank == 0 else (None, None)))[2:]
        elif streaming:  # one call per frame: cell updates + map rebuild + search of the persistent queries
            off, cells, cost, status = planner.replan_frame(*frames[i])
        else:
            off, cells, cost, status = planner.plan_batch(starts, goals, hchoice, mpl)  # blocking: results are in host memory
        last["res"] = (off, cells, cost, status)  # (what the CPU legs check, cell for cell: outside the timed region)
        return status, planner.timing()

    kernel_ms, per_ctx = [], None
    head_ms, batch_ms = [], []
    status = None
    retried = reused = direct = 0
    waves = waves_short = 0
    launches = 1
    lat = None
    if pipe is not None:  # frames / batches in flight: submit them all, the pipeline hands each to the next free handle
        def job(i):
            return frames[i] if streaming else (starts, goals, hchoice, mpl)
        for f in [pipe.submit(*job(i)) for i in range(warmup)]:
            f.result()
        t_sub, t_done, futs = [], {}, []
        t0 = time.perf_counter()
        for i in range(steps):
            t_sub.append(time.perf_counter())  # (submit blocks while the handle whose turn it is still plans)
            f = pipe.submit(*job(warmup + i))
            f.add_done_callback(lambda _f, _i=i: t_done.__setitem__(_i, time.perf_counter()))
            futs.append(f)
        res = [f.result() for f in futs]
        elapsed = time.perf_counter() - t0
        if any((r[3] < 0).any() for r in res):
            raise SystemExit("bench: queries failed")
        status = res[-1][3]  # (the last frame's: the CPU baseline below plans on that frame's grid)
        last["res"] = res[-1]
        direct = 1
        kernel_ms = [float("nan")]  # (the launches of the frames overlap: there is no kernel time; see frame_period_ms)
        frame_period_ms = elapsed / steps * 1e3
        time.sleep(0.02)
        lat = np.array([(t_done[i] - t_sub[i]) * 1e3 for i in range(steps) if i in t_done])
        tm = [p.timing() for p in pipe.planners]
        waves = int(sum(t["waves"] for t in tm))
        waves_short = int(any(t["waves_short"] for t in tm))
        pipe.close()
    else:
        for i in range(warmup):
            step(i)
        cx.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            status, tm = step(warmup + i)
            kernel_ms.append(tm["search_kernel_ms"])
            retried += tm["retried"]
            reused += tm["reused"]
            direct = tm.get("table_direct", 0)
            waves, waves_short = int(tm["waves"]), int(tm["waves_short"])
            launches = int(tm.get("search_launches", 1))
            head_ms.append(tm.get("head_launch_ms", 0.0))
            batch_ms.append(tm.get("batch_launch_ms", 0.0))
        cx.sync()
        elapsed = time.perf_counter() - t0
        per_ctx = planner.timing_per_context()  # (of the last step)
    k_ms = float(np.mean(kernel_ms))
    if pipe is not None:
        k_ms = None
    k_all = [k_ms]
    if cx.world > 1:  # (maximum and per-rank kernel times over the rendezvous socket)
        parts = cx.rdv.bcast(cx.rdv.gather([elapsed, k_ms]))
        elapsed = max(float(x[0]) for x in parts)
        k_all = [float(x[1]) for x in parts]
    elif per_ctx is not None and len(per_ctx) > 1:
        k_all = [c["kernel_ms"] for c in per_ctx]
    if (status < 0).any():
        raise SystemExit("bench: %d queries failed" % int((status < 0).sum()))
    verified = verify_multi(cx, wl, occ, g_last if streaming else occ, starts, goals, last.get("res"), strong, nq) if (cx.world > 1 or cx.gpus > 1) and pipe is None else None
    if cx.rank != 0:
        return None

    total_q = nq if strong else nq * n_units  # queries all ranks planned per step
    # algorithmic bytes of what rank 0 launched (cells + 16 * (pushes + pops), literal reference run)
    if streaming:
        per = wl.get("algorithmic_bytes_per_frame") or []
        fr = [per[warmup + i] for i in range(steps) if warmup + i < len(per)]
        algo = float(np.mean(fr)) if len(fr) == steps else None  # frames beyond the committed counts: unknown
    elif wl.get("algorithmic_bytes") is None:
        algo = None
    else:  # weak: each device plans exactly the committed workload; strong: the mean shard
        algo = float(wl["algorithmic_bytes"]) / (n_units if strong else 1)
    k_dev = max(k_all) if (cx.inlib and cx.gpus > 1) else k_ms  # the kernel time the algorithmic bytes of ONE device go against
    achieved = algo / (k_dev * 1e-3) / 1e9 if (algo and k_dev) else None
    out = {
        "workload": name, "W": W, "H": H, "hchoice": hchoice, "streaming": streaming, "strong": strong,
        "describe": wl["describe"], "value": total_q * steps / elapsed, "ms_per_step": elapsed / steps * 1e3,
        "steps": steps, "warmup": warmup, "total_q": total_q, "n_local": n_local, "status": status, "starts": starts, "goals": goals,
        "grid_now": g_last if streaming else None, "occ": occ, "mpl": mpl,
        "kernel_ms": k_dev, "kernel_ms_per_device": k_all, "algo": algo, "achieved": achieved,
        "frac": achieved / HBM_PEAK_GBS if achieved else None, "retried": int(retried), "reused": reused, "direct": direct,
        "waves": waves, "waves_short": waves_short, "launches": launches, "wl": wl, "frames": frames,
        "launch_ms": ({"head": float(np.mean(head_ms)), "batch": float(np.mean(batch_ms))} if launches == 2 and head_ms else None),
        "last": last.get("res"), "verified": verified,
        "kernel": "fx::k_search<%d, %s, %s>" % (hchoice, "true" if (streaming and name not in ("c5", "c5pipe")) else "false", "true" if direct else "false"),
    }
    if streaming:
        out["frames_per_s"] = steps / elapsed
        out["cells_sent_per_frame"] = int(len(frames[0][1]))
        out["results_reused_per_frame"] = reused / steps
    if pipe is not None:
        out["frame_period_ms"] = frame_period_ms
        # (what the frames' overlapping launches move per second: algorithmic bytes per frame / frame period -- a rate of
        # the pipeline, not of a kernel)
        out["pipeline_GBps"] = algo / (frame_period_ms * 1e-3) / 1e9 if algo else None
        out["frames_in_flight"] = int(wl.get("frames_in_flight") or wl["batches_in_flight"])
        out["latency_ms"] = {"mean": float(lat.mean()), "p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                             "max": float(lat.max()), "frames": int(len(lat))}
    return out


def _sample_ids(lo, hi, k):
    """k query positions of the shard [lo, hi), evenly spread, its first and last among them"""
    n = hi - lo
    if n <= k:
        return np.arange(lo, hi)
    return np.unique(np.r_[lo + (np.arange(k, dtype=np.int64) * n) // k, hi - 1])


def verify_multi(cx, wl, occ0, grid_now, starts, goals, res, strong, nq, per_shard=None):
    """What makes an N > 1 line evidence (outside every timed region): (a) the SHA-256 of the grid each
    context / rank holds resident equals the one rank 0 generated; (b) a stratified sample of EVERY shard's results of the
    last step -- `per_shard` queries each, at least 500 -- equals the C oracle's on rank 0: lengths, float64 cost bytes,
    cells.  Raises SystemExit(3) on any difference.  -> the `config.verified` record (rank 0; None elsewhere)."""
    per_shard = per_shard or int(os.environ.get("FXJPS_BENCH_VERIFY_PER_SHARD", "500"))
    want = hashlib.sha256(np.ascontiguousarray(grid_now).tobytes()).hexdigest()
    off, cells, cost, status = res
    hc, mpl = wl["hchoice"], wl["max_path_len"]
    if cx.world > 1:
        hashes = cx.sp.grid_hashes()
        lo, hi = 0, len(starts)  # (this rank's own queries: it drew them itself)
        ids = _sample_ids(lo, hi, per_shard)
        picked = [np.concatenate([cells[off[q]:off[q + 1]] for q in ids] or [np.zeros((0, 2), np.int32)]),
                  np.array([off[q + 1] - off[q] for q in ids], np.int64)]
        parts = cx.rdv.gather((starts[ids], goals[ids], status[ids], cost[ids], picked[0], picked[1]))
        shards = parts if cx.rank == 0 else None
    else:
        hashes = [hashlib.sha256(cx.planner.get_grid(c).tobytes()).hexdigest() for c in range(cx.gpus)]
        shards = []
        for r in range(cx.gpus):  # the library's contiguous shards of the batch (fxjps.hip plan_core: nq * r / nd)
            lo, hi = len(starts) * r // cx.gpus, len(starts) * (r + 1) // cx.gpus
            ids = _sample_ids(lo, hi, per_shard)
            shards.append((starts[ids], goals[ids], status[ids], cost[ids],
                           np.concatenate([cells[off[q]:off[q + 1]] for q in ids] or [np.zeros((0, 2), np.int32)]),
                           np.array([off[q + 1] - off[q] for q in ids], np.int64)))
    ok_hash = all(h == want for h in hashes)
    if cx.rank != 0:
        return None
    if not ok_hash:
        print("bench: the grid differs between devices / ranks: %s (rank 0 generated %s)" % (hashes, want), file=sys.stderr)
        raise SystemExit(3)
    from oracle import oracle  # the checker: never on the planner's path, never inside a timed region
    checked = []
    for r, (s_, g_, st_, co_, ce_, ln_) in enumerate(shards):
        oc, ol, ocost, _ = oracle.plan_batch(grid_now, s_, g_, hc, literal=False, max_len=mpl, nthreads=min(os.cpu_count() or 1, 64))
        keep = np.arange(oc.shape[1])[None, :] < np.maximum(ol, 0)[:, None]
        if not (np.array_equal(ol, st_) and ocost.tobytes() == np.asarray(co_, np.float64).tobytes() and np.array_equal(oc[keep], ce_)
                and np.array_equal(np.maximum(ol, 0), ln_)):
            print("bench: shard %d of %d disagrees with the oracle (lengths / float64 cost bytes / cells)" % (r, len(shards)), file=sys.stderr)
            raise SystemExit(3)
        checked.append(int(len(s_)))
    return {"grid_hash_equal": True, "grid_sha256_16": want[:16], "devices_or_ranks_hashed": len(hashes), "shards_checked": len(shards),
            "queries_checked_per_shard": checked, "against": "oracle/jps_oracle.c on rank 0: lengths, float64 cost bytes, cells of the last step"}


def measure_c1(cx, reps=20):
    """BASELINE config 1 and the node's real call: ONE query per call on the reference's own maps (tests/golden/maps_png.npz:
    the 35 maps/*.png as the loader of global_planner_st.py:176-182 reads them).  The 147 x 112 map pasted into a zero
    256 x 256 canvas with the three queries of SURVEY 8(c), and every map with its first-free -> last-free query.  Every
    answer is checked (canvas: against the paths captured from the real jps1.py; maps: against the C oracle).  Microseconds
    per call of Planner.plan on the resident grid and of the jps1.method drop-in (grid conversion + upload + map build +
    search), beside the C port and the pure-Python restatement of the reference on one host core."""
    import contextlib
    import io
    import fuxi_planner_amd as fx
    from oracle import jps_python, oracle
    z = np.load(os.path.join(ROOT, "tests", "golden", "maps_png.npz"))
    with open(os.path.join(ROOT, "tests", "golden", "maps_png.json")) as f:
        recs = json.load(f)
    p = cx.planner

    def timed(fn, n):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        return (time.perf_counter() - t0) / n * 1e6, r

    name = "-16.20-11.40_out.png"
    occ = np.zeros((256, 256), dtype=np.uint8)
    occ[:147, :112] = np.unpackbits(z[name])[:147 * 112].reshape(147, 112)
    m64 = occ.astype(np.float64)  # what the node passes (st:248)
    canvas = []
    p.set_grid_occ(occ)
    for rec in [r for r in recs if r.get("canvas") and r["map"] == name][:3]:
        s, g = tuple(rec["start"]), tuple(rec["goal"])
        want = [tuple(rec["path"][i:i + 2]) for i in range(0, len(rec["path"]), 2)]
        us, path = timed(lambda: p.plan(s, g, 2), reps)
        assert path == want, "c1: the path differs from the one captured from jps1.py"
        with contextlib.redirect_stdout(io.StringIO()):
            us_m, r = timed(lambda: fx.jps1.method(m64, s, g, 2), reps)  # the map of the last tick again: not uploaded
            assert r[0] == want
            # ... and a map that differs from the last tick's (one cell in a far corner of the canvas, toggled every call):
            # conversion + upload + map build + search
            m_alt = [m64, m64.copy()]
            m_alt[1][255, 255] = 1.0
            tick = [0]

            def new_map():
                tick[0] += 1
                return fx.jps1.method(m_alt[tick[0] & 1], s, g, 2)
            us_new, r = timed(new_map, reps)
        assert r[0] == want
        us_c, rc = timed(lambda: oracle.plan(occ, s, g, 2, literal=False), reps)
        assert rc[0] == want
        t0 = time.perf_counter()
        rp = jps_python.search(m64, s, g, 2)
        ms_py = (time.perf_counter() - t0) * 1e3
        assert rp[0] == want
        canvas.append({"start": list(s), "goal": list(g), "jump_points": len(want), "us_per_call_resident_grid": us,
                       "us_per_call_jps1_method": us_m, "us_per_call_jps1_method_new_map": us_new, "kernel_us": p.timing()["search_kernel_ms"] * 1e3,
                       "c_port_us_one_core": us_c, "python_restatement_ms_one_core": ms_py})
    maps = []
    for nm in z.files:
        rec = [r for r in recs if r["map"] == nm and "canvas" not in r][0]
        W, H = rec["shape"]
        grid = np.unpackbits(z[nm])[:W * H].reshape(W, H).astype(np.uint8)
        free = np.argwhere(grid == 0)
        s, g = tuple(int(v) for v in free[0]), tuple(int(v) for v in free[-1])
        p.set_grid_occ(grid)
        us, path = timed(lambda: p.plan(s, g, 2), 5)
        us_c, rc = timed(lambda: oracle.plan(grid, s, g, 2, literal=False), 5)
        assert (path if path else 0) == rc[0], "c1: %s differs from the oracle" % nm
        maps.append((us, us_c, W * H))
    mu = np.array([m[0] for m in maps])
    mc = np.array([m[1] for m in maps])
    us_mean = float(np.mean([c["us_per_call_resident_grid"] for c in canvas]))
    return {"value": 1e6 / us_mean, "unit": "plans/s (one query per call)", "us_per_call": us_mean,
            "describe": "BASELINE config 1: the 147x112 reference map in a 256x256 canvas, the three SURVEY 8(c) queries, one query per call "
                        "(mean of %d calls each); then the 35 reference maps, first-free -> last-free" % reps,
            "canvas_queries": canvas,
            "maps35_us_per_call": {"mean": float(mu.mean()), "median": float(np.median(mu)), "max": float(mu.max()), "min": float(mu.min())},
            "maps35_c_port_us_one_core": {"mean": float(mc.mean()), "median": float(np.median(mc)), "max": float(mc.max())},
            "maps35_cells": {"min": int(min(m[2] for m in maps)), "max": int(max(m[2] for m in maps))},
            "checked": "every path against jps1.py's (canvas) / the C oracle (maps)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c2h1", "c2pipe", "c3", "c4shard", "c4", "c5", "c5low", "c5local", "c5local4k", "c5pipe"])
    ap.add_argument("--inlib", action="store_true", help="one process, all GPUs through fxjps_create(n_dev = N) (the default when "
                    "bench.py is not started by torch.distributed.run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="default workload: skip the config.also measurements")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="c5pipe / c2pipe: planner handles taking the frames / batches in turn (0: the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries timed on the host cores (0: the workload's default)")
    ap.add_argument("--py-sample", type=int, default=64, help="queries timed with the pure-Python restatement of the reference (0: none)")
    a = ap.parse_args()
    d_steps, d_warm = DEFAULT_STEPS.get(a.workload, (5, 2))
    a.steps = d_steps if a.steps is None else a.steps
    a.warmup = d_warm if a.warmup is None else a.warmup
    if a.workload in ("c5pipe", "c2pipe"):
        # one hardware queue per planner handle, or the persistent search kernels of handles that share a queue run one
        # after the other; read by the HIP runtime when it initialises (nothing has touched the GPU yet)
        from fuxi_planner_amd.replan import configure_hw_queues
        configure_hw_queues()

    cx = Ctx()
    cx.rank = int(os.environ.get("RANK", "0"))
    cx.world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cx.gpus = a.gpus
    # Decided before anything touches a GPU: started by torch.distributed.run (WORLD_SIZE == N) -> one process per GPU;
    # otherwise ONE process drives all N GPUs through the library's multi-device handle (no torch is imported at all).
    cx.inlib = a.inlib or (cx.world == 1 and a.gpus > 1)
    if cx.inlib and cx.world > 1:
        raise SystemExit("--inlib is a single-process mode: start it without torch.distributed.run")
    if not cx.inlib and cx.world != a.gpus:
        raise SystemExit("--gpus %d under torch.distributed.run needs --nproc-per-node %d (WORLD_SIZE is %d)" % (a.gpus, a.gpus, cx.world))

    import fuxi_planner_amd as fx

    with open(os.path.join(ROOT, "fuxi-planner_amd", "workloads.json")) as f:
        cx.WL = json.load(f)
    # (config 2 itself, batch after batch with several in flight: the same grid, the same 10 000 queries per batch)
    cx.WL["c2pipe"] = dict(cx.WL["c2"], batches_in_flight=2, describe=cx.WL["c2"]["describe"].replace(
        "BASELINE config 2:", "BASELINE config 2, batches back to back with 2 in flight (2 planner handles on the GPU take them in turn; "
        "every batch is the same fxjps_plan_batch call):"))

    cx.sp = cx.rdv = None
    # One process per GPU (WORLD_SIZE > 1, e.g. under torch.distributed.run): NO torch in the process -- the ranks meet over a
    # TCP socket at MASTER_ADDR : MASTER_PORT + 1 (fuxi_planner_amd.ranks), rank 0 hands out the RCCL id, the library
    # broadcasts the grid (fxjps_create_rank / fxjps_set_grid_rank).
    # Bring-up aid for 1-GPU boxes (never set by the driver): FXJPS_BENCH_ONE_DEVICE=1 puts every rank (or every context
    # of the in-library handle) on device 0 -- everything but the collective, which RCCL refuses between ranks that share
    # a device: the grid bytes then travel over the rendezvous socket.
    one_dev = bool(os.environ.get("FXJPS_BENCH_ONE_DEVICE"))
    cx.one_dev = one_dev
    cx.dev_index = 0 if one_dev else local_rank

    if cx.world > 1:
        from fuxi_planner_amd.ranks import RankPlanner, Rendezvous
        cx.rdv = Rendezvous.from_env()
        # (FXJPS_BENCH_TRY_RCCL=1 with FXJPS_BENCH_ONE_DEVICE=1: the ranks ask RCCL all the same -- it refuses ranks that share a
        # device --, which rehearses the agreement of RankPlanner on "every rank or none" and its way over the socket)
        cx.sp = RankPlanner(cx.rdv, device=cx.dev_index, host_broadcast=one_dev and not os.environ.get("FXJPS_BENCH_TRY_RCCL"))
        cx.planner = cx.sp.engine
        if one_dev:
            cx.planner.set_memory_share(cx.world)  # (the rehearsal's ranks share device 0: a host that does that says so)
    else:
        if cx.inlib:
            devs = [0] * a.gpus if one_dev else list(range(a.gpus))
        else:
            devs = [cx.dev_index]
        cx.planner = fx.Planner(devs)

    def sync():
        if cx.rdv is not None:
            cx.rdv.barrier()  # (the planner's calls are blocking: results are in host memory when they return)
    cx.sync = sync

    if a.workload == "c1":  # single calls: a latency, reported on its own line
        if cx.world > 1 or a.gpus > 1:
            raise SystemExit("c1 is a single-GPU, single-process workload (one query per call)")
        r = measure_c1(cx)
        print(json.dumps({"metric": "start->goal plans/sec, one query per call, 256x256 canvas of a reference map", "value": r["value"], "unit": "plans/s",
                          "n_gpus": 1, "steps": 20, "warmup": 1, "ms_per_step": r["us_per_call"] / 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64", "data": "the reference's maps/*.png (tests/golden/maps_png.npz)",
                          "config": {"workload": "c1: " + r["describe"], "detail": r}}), flush=True)
        cx.planner.close()
        return
    rc_final = 0
    m = measure(cx, a.workload, a.steps, a.warmup, a.frames_in_flight)

    # ---- the other BASELINE workloads, behind the headline's timed region (default invocation only)
    also = {}
    if a.workload == "c2" and not a.no_also:
        def brief(r, extra=()):
            d = {"value": r["value"], "unit": "plans/s", "ms_per_step": r["ms_per_step"], "steps": r["steps"], "kernel_ms": r["kernel_ms"],
                 "frac": r["frac"], "achieved_GBps": r["achieved"], "queries_per_step": r["total_q"], "scaling": "strong" if r["strong"] else "weak"}
            for k in extra:
                d[k] = r[k]
            return d
        if a.gpus == 1:
            r = measure(cx, "c4shard", 3, 1)
            if r:
                also["c4shard"] = brief(r)
            if cx.world == 1:
                # BASELINE config 4 itself on ONE GPU: the anchor of the strong-scaling curve (at N > 1 the same 1 M queries
                # are split N ways, below)
                r = measure(cx, "c4", 1, 1)
                if r:
                    also["c4"] = brief(r, ("kernel_ms_per_device",))
                also["c1"] = measure_c1(cx)
                r = measure(cx, "c3", 1, 1)  # BASELINE config 3: one warm-up and one timed step (100 000 queries on 4096^2, ~ 10 s each)
                if r:
                    also["c3"] = brief(r, ("retried",))
                    also["c3"]["resident_wavefronts"] = r["waves"]
        else:
            r = measure(cx, "c4", 2, 1)  # the 1 M queries of BASELINE config 4 split N ways: the strong-scaling workload
            if r:
                also["c4"] = brief(r, ("kernel_ms_per_device", "verified"))

    if cx.rank == 0:
        ci = cx.planner.comm_info()
        if cx.world > 1:
            rccl_ranks = ci["rccl_ranks"]
            collective = ("ncclBroadcast of the grid inside the library (fxjps_set_grid_rank), ncclCommCount = %d; rendezvous over TCP, no torch in the process" % rccl_ranks) \
                if rccl_ranks else ("none: RCCL did not come up (%s), the grid bytes travel over the rendezvous socket" % cx.sp.rccl_error if getattr(cx.sp, "rccl_error", None)
                                    else "none: the ranks share one device (rehearsal), the grid bytes travel over the rendezvous socket")
        else:
            rccl_ranks = ci["rccl_ranks"]
            collective = ("ncclBroadcast of the grid inside the library, ncclCommCount = %d" % rccl_ranks) if rccl_ranks else \
                         ("none (one device)" if ci["contexts"] == 1 else "none: %d contexts share %d device(s), device-to-device copies" % (ci["contexts"], ci["devices"]))
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
        wl = m["wl"]
        out = {
            "metric": "start->goal plans/sec on %dx%d grid" % (m["W"], m["H"]),
            "value": m["value"],
            "unit": "plans/s",
            "n_gpus": a.gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": m["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong" if m["strong"] else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: %s" % (a.workload, m["describe"]),
                       "queries_per_step": m["total_q"], "queries_on_rank0": m["n_local"], "grid": [m["W"], m["H"]], "hchoice": m["hchoice"],
                       "reachable_rank0": int((m["status"] > 0).sum()), "retried_on_large_scratch": m["retried"],
                       "parallelism": ("one process, fxjps_create(n_dev=%d)" % a.gpus) if cx.inlib else "queries sharded x%d, one process per GPU%s" % (a.gpus, ", torch-free ranks" if cx.rdv is not None else ""),
                       "verified": m["verified"],
                       "torch_imported": "torch" in sys.modules,
                       "rccl_ranks": rccl_ranks, "collective": collective, "contexts": ci["contexts"] if cx.world == 1 else cx.world,
                       "kernel_ms_per_device": m["kernel_ms_per_device"], "resident_wavefronts": m["waves"], "wavefronts_cut_by_memory": m["waves_short"]},
            "roofline": {"bound": "hbm", "achieved": m["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": m["frac"], "traffic": traffic,
                         # (the instantiation rocprofv3 lists: heuristic, read-set recording, table indexed by the cell)
                         "kernel": m["kernel"],
                         "kernel_ms": m["kernel_ms"], "launches_per_step": m["launches"], "launch_ms": m["launch_ms"],
                         "algorithmic_bytes_per_step": m["algo"],
                         "algorithmic_bytes_source": wl.get("algorithmic_bytes_source", "oracle literal mode, all queries (tools/algo_bytes.py)"),
                         "note": "graph search bound by scattered-request rate and instruction issue, not by bytes (DESIGN.md section 4); traffic: " + tnote},
        }
        if m["launches"] == 2:
            out["roofline"]["note"] = ("a step is TWO overlapping launches of this kernel (the 16 longest queries on CUs of their own, "
                                       "the rest of the batch beside them): kernel_ms is the HIP-event time around both, launch_ms that of "
                                       "each, algorithmic bytes are the step's; " + out["roofline"]["note"])
        if m["streaming"]:
            out["config"].update({"frames_per_s": m["frames_per_s"], "target_frames_per_s": 60,
                                  "cells_sent_per_frame": m["cells_sent_per_frame"],
                                  "results_reused_per_frame": m["results_reused_per_frame"]})
            if "latency_ms" in m:
                out["config"].update({"frames_in_flight": m["frames_in_flight"], "planner_handles": m["frames_in_flight"],
                                      "frame_latency_ms": m["latency_ms"]["mean"], "submit_to_paths_latency_ms": m["latency_ms"],
                                      "parallelism": "%d planner handles on one GPU take the frames in turn" % m["frames_in_flight"]})
                out["config"]["frame_period_ms"] = m["frame_period_ms"]
                out["roofline"].update({"achieved": None, "frac": None, "kernel_ms": None, "pipeline_GBps": m["pipeline_GBps"]})
                out["roofline"]["note"] = ("the launches of the frames in flight overlap: there is no kernel time and no roofline fraction of a "
                                           "kernel; pipeline_GBps = algorithmic bytes per frame / frame period; " + out["roofline"]["note"])
        if "latency_ms" in m and not m["streaming"]:  # batches in flight (c2pipe)
            out["config"].update({"batches_in_flight": m["frames_in_flight"], "planner_handles": m["frames_in_flight"],
                                  "submit_to_paths_latency_ms": m["latency_ms"], "batch_period_ms": m["frame_period_ms"],
                                  "parallelism": "%d planner handles on one GPU take the batches in turn" % m["frames_in_flight"]})
            out["roofline"].update({"achieved": None, "frac": None, "kernel_ms": None, "pipeline_GBps": m["pipeline_GBps"]})
            out["roofline"]["note"] = ("the launches of the batches in flight overlap: there is no kernel time and no roofline fraction of a "
                                       "kernel; pipeline_GBps = algorithmic bytes per batch / batch period; " + out["roofline"]["note"])
        if a.workload == "c2" and not a.no_also and a.gpus == 1 and cx.world == 1:
            # config 5 with frames in flight, in a child process of its own (its handles want their own hardware
            # queues -- an environment variable the HIP runtime reads when it starts -- and their own memory)
            cx.planner.close()
            cx.planner = None
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "c5pipe", "--no-cpu-baseline"],  # (its 600 frames)
                                   capture_output=True, text=True, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and line:
                    j = json.loads(line[-1])
                    also["c5pipe"] = {"value": j["config"]["frames_per_s"], "unit": "frames/s", "target": 60, "steps": j["steps"],
                                      "frame_period_ms": j["config"]["frame_period_ms"], "kernel_ms": None, "frac": None,
                                      "pipeline_GBps": j["roofline"]["pipeline_GBps"],
                                      "planner_handles": j["config"]["planner_handles"],
                                      "submit_to_paths_latency_ms": j["config"]["submit_to_paths_latency_ms"],
                                      "plans_per_s": j["value"]}
                else:
                    also["c5pipe"] = {"error": (r.stderr or r.stdout)[-300:]}
            except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
                also["c5pipe"] = {"error": repr(e)[:300]}
            # ... and config 2 itself, batch after batch with two in flight: what the chip does with the headline's batches
            # when it need not wait for each one's slowest query
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "c2pipe", "--no-cpu-baseline"],
                                   capture_output=True, text=True, timeout=200)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and line:
                    j = json.loads(line[-1])
                    also["c2pipe"] = {"value": j["value"], "unit": "plans/s", "steps": j["steps"], "queries_per_step": j["config"]["queries_per_step"],
                                      "batch_period_ms": j["config"]["batch_period_ms"], "kernel_ms": None, "frac": None,
                                      "pipeline_GBps": j["roofline"]["pipeline_GBps"], "planner_handles": j["config"]["planner_handles"],
                                      "submit_to_paths_latency_ms": j["config"]["submit_to_paths_latency_ms"]}
                else:
                    also["c2pipe"] = {"error": (r.stderr or r.stdout)[-300:]}
            except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
                also["c2pipe"] = {"error": repr(e)[:300]}
        if (cx.world > 1 or cx.gpus > 1) and rccl_ranks != a.gpus:
            if cx.one_dev:
                out["config"]["rehearsal"] = "FXJPS_BENCH_ONE_DEVICE: every rank / context on device 0, no RCCL communicator -- not a multi-GPU measurement"
            else:  # the line is printed (it says what happened), the run fails
                out["config"]["error"] = "the RCCL communicator spans %d of %d ranks" % (rccl_ranks, a.gpus)
                rc_final = 3
        if also:
            out["config"]["also"] = also
        if not a.no_cpu_baseline:  # (at N > 1 too: rank 0's shard, the same leg)
            from oracle import oracle  # checker used as the CPU baseline ("port"), never by the planner
            ns = min(a.cpu_sample or wl.get("cpu_sample", m["wl"]["nq"]), m["n_local"])
            cores = min(os.cpu_count() or 1, 256)
            nth = min(cores, wl.get("cpu_threads", cores))
            g_now = m["grid_now"] if m["grid_now"] is not None else m["occ"]  # (streaming: the grid after the last frame)
            tc = time.perf_counter()
            oc, ol, ocost, _ = oracle.plan_batch(g_now, m["starts"][:ns], m["goals"][:ns], m["hchoice"], literal=False, max_len=m["mpl"], nthreads=nth)
            dt = time.perf_counter() - tc
            # the CPU leg is the checker too: lengths, float64 costs as bytes, every cell of every path
            off_g, cells_g, cost_g, _ = m["last"]
            assert np.array_equal(ol, m["status"][:ns]), "the C port disagrees with the GPU on path lengths"
            assert ocost.tobytes() == cost_g[:ns].tobytes(), "the C port disagrees with the GPU on a cost (float64 bytes)"
            keep = np.arange(oc.shape[1])[None, :] < np.maximum(ol, 0)[:, None]
            assert np.array_equal(oc[keep], cells_g[:off_g[ns]]), "the C port disagrees with the GPU on the cells of a path"
            out["cpu_baseline"] = {"value": ns / dt, "unit": "plans/s", "cores": nth, "kind": "port",
                                   "sample": "first %d queries of the same batch, oracle/jps_oracle.c (-O2), "
                                             "%d pthreads, %.1f s" % (ns, nth, dt)}
            if not m["streaming"] and m["W"] * m["H"] <= (1 << 20):
                # The reference's own cost structure (dicts, heapq, the O(|open|) membership scan of jps1.py:224) in pure
                # Python -- oracle/jps_python.py, pinned bit for bit to the real jps1.py by tests/test_oracle_python.py: the
                # reference file itself cannot travel to this box.  A fixed subset, one process per host core.
                from oracle import jps_python
                npy = min(a.py_sample, m["n_local"])
                if npy > 0:
                    nproc = min(os.cpu_count() or 1, npy)
                    lens, pcost, wall, cpu_s = jps_python.timed_batch(m["occ"], m["starts"][:npy], m["goals"][:npy], m["hchoice"], nproc)
                    assert lens == [max(int(v), 0) for v in m["status"][:npy]], "the Python restatement disagrees with the GPU on path lengths"
                    for q in range(npy):  # (its printed cost, jps1.py:207, is the float64 the GPU returns -- bit for bit)
                        if lens[q] > 0:
                            assert np.float64(pcost[q]).tobytes() == np.float64(cost_g[q]).tobytes(), "the Python restatement disagrees with the GPU on a cost"
                    out["cpu_baseline_python"] = {"value": npy / wall, "unit": "plans/s", "cores": nproc, "kind": "port",
                                                  "per_core_plans_per_s": npy / cpu_s,
                                                  "sample": "first %d queries of the same batch, oracle/jps_python.py (pure Python restatement with the "
                                                            "reference's data structures: dict / heapq / list scan of jps1.py:224), one process per core on "
                                                            "%d of the host's %d hardware threads, %.1f s wall, %.1f s summed over the processes"
                                                            % (npy, nproc, os.cpu_count() or 1, wall, cpu_s)}
        print(json.dumps(out), flush=True)
    if cx.rdv is not None:
        cx.rdv.barrier()
        cx.rdv.close()
    if cx.planner is not None:
        cx.planner.close()
    if rc_final:
        raise SystemExit(rc_final)


if __name__ == "__main__":
    main()
