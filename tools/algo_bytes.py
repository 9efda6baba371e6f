#!/usr/bin/env python3
"""Algorithmic bytes of the benchmark workloads:
    B_q = cells_q * 1 B + (pushes_q + pops_q) * 16 B
with the three counts taken on the LITERAL reference algorithm (the oracle's literal mode, whose counters are pinned
to the real jps1.py by tests/golden).  Writes fuxi-planner_amd/workloads.json, which bench.py reads for
roofline.achieved.  Deterministic: depends only on the seeds.

    python tools/algo_bytes.py [c2 c2h1 c4shard c3 c5 c5low]     # default: all; c4 is derived from c4shard

c3 counts all 100 000 queries in chunks of 2 500 (round 5; until then a 2 000-query sample x 50; FX_C3_SAMPLE=n for a
sample) -- the oracle needs ~0.5 M pops per query there, the whole count takes ~ 25 min of 64 host threads, finished
chunks are kept under gpurun_out/algo_parts/ and a later run goes on behind them; c4 is
8 x the literal count of its first 125 000 queries; c5 holds one literal count per frame for the first 616 frames of
the toggle stream (16 warm-up frames + the 600 frames SURVEY 8d prescribes; c5pipe is the same stream), c5low / c5local
for the first 40.  Meant for a many-core host (the GPU box: ~5 min on 256 threads, c5 alone ~10 min).
"""
import json, os, re, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle
from fuxi_planner_amd import synth

OUT = os.path.join(ROOT, "fuxi-planner_amd", "workloads.json")
C2 = dict(W=1024, H=1024, grid_seed=1, p=0.20, qseed=1, hchoice=2, max_path_len=1024)
WORKLOADS = {
    "c2": dict(C2, nq=10000, describe="BASELINE config 2: 1024x1024 grid, 20% splitmix64 obstacles (seed 1), 10000 uniform free-cell "
                                      "(start,goal) queries per GPU (qseed 1), hchoice=2, all paths returned to host"),
    "c2h1": dict(C2, nq=10000, hchoice=1, describe="config 2 with hchoice=1 (octile x10/x14 keys)"),
    "c4shard": dict(C2, nq=125000, cpu_sample=20000, describe="one GPU's share of BASELINE config 4: 1024x1024 grid (seed 1), 125000 queries "
                                                             "(the first eighth of the 1M-query stream, qseed 1), hchoice=2"),
    "c4": dict(C2, nq=1000000, strong=True, cpu_sample=20000,
               describe="BASELINE config 4: 1024x1024 grid (seed 1), 1000000 queries (qseed 1) split over the GPUs in contiguous shards, hchoice=2"),
    "c3": dict(W=4096, H=4096, grid_seed=2, p=0.20, qseed=2, hchoice=2, max_path_len=4096, nq=100000, sample=int(os.environ.get("FX_C3_SAMPLE", "100000")), chunk=2500, cpu_sample=512, cpu_threads=64,
               describe="BASELINE config 3: 4096x4096 grid, 20% obstacles (seed 2), 100000 queries (qseed 2), hchoice=2"),
    "c5": dict(C2, qseed=5, nq=1000, max_path_len=2048, toggle_frac=0.05, toggle_seed=5, frames=616,  # 16 warm-up + the 600 frames of SURVEY 8d
               describe="BASELINE config 5, streaming replan: 1024x1024 grid (seed 1), per frame floor(0.05*W*H) occupied cells freed and as many "
                        "free cells occupied (10% of the cells toggled, SURVEY 8d splitmix stream, seed 5), 1000 persistent queries (qseed 5) "
                        "replanned every frame, target 60 frames/s"),
    "c5low": dict(C2, qseed=5, nq=1000, max_path_len=2048, toggle_frac=0.0005, toggle_seed=5, frames=40,
                  describe="config 5 with 0.1% of the cells toggled per frame, scattered uniformly over the map (low churn, but no locality)"),
    "c5local": dict(C2, qseed=5, nq=1000, max_path_len=2048, toggle_frac=0.0, toggle_mode="local", window=64, toggle_seed=5, frames=40,
                    describe="config 5 with local churn (the ROS node's ticks: the map changes where the vehicle looks): per frame one 64x64 "
                             "window, moving with the frame index, is re-observed (all its cells sent, fresh 20% values), 1000 persistent queries"),
    "c5local4k": dict(W=4096, H=4096, grid_seed=2, p=0.20, qseed=5, hchoice=2, nq=1000, max_path_len=8192, toggle_frac=0.0, toggle_mode="local",
                      window=64, toggle_seed=5, frames=8, cpu_sample=256, cpu_threads=64,
                      describe="streaming replan at the size of config 3: 4096x4096 grid (seed 2), per frame one 64x64 window re-observed, 1000 persistent "
                               "queries (qseed 5): the read-set instantiation on hashed tables, partial map rebuilds, united component labels"),
}
PIPE_FRAMES = 12  # (round 4: 8; measured 5 .. 16) c5pipe: config 5 with this many frames in flight (same frames, same counts: derived from c5)
NT = min(os.cpu_count() or 8, 256)


def count(occ, s, g, h, mpl, nthreads=NT):
    _, ln, _, st = oracle.plan_batch(occ, s, g, h, literal=True, max_len=mpl, nthreads=nthreads, want_stats=True)
    cells, pushes, pops = int(st["cells"].sum()), int(st["pushes"].sum()), int(st["pops"].sum())
    return dict(cells=cells, pushes=pushes, pops=pops, algorithmic_bytes=cells + 16 * (pushes + pops)), ln


def main():
    which = sys.argv[1:] or ["c2", "c2h1", "c4shard", "c3", "c5", "c5low", "c5local"]
    try:
        with open(OUT) as f:
            out = json.load(f)
    except (OSError, ValueError):
        out = {}
    out.pop("c2_1024_p20_10k", None)
    for name in which:
        w = WORKLOADS[name]
        rec = dict(w)
        t = time.time()
        occ = oracle.synth_grid(w["W"], w["H"], w["grid_seed"], w["p"])
        if "toggle_frac" in w:
            s, g = oracle.synth_queries(occ, w["qseed"], w["nq"])
            keep = np.zeros(occ.shape, dtype=bool)
            keep[s[:, 0], s[:, 1]] = True
            keep[g[:, 0], g[:, 1]] = True
            per, reach = [], []
            for fr in range(w["frames"]):
                xy, val = synth.frame_update(occ, keep, fr, w)
                synth.apply_toggles(occ, xy, val)
                c, ln = count(occ, s, g, w["hchoice"], w["max_path_len"], nthreads=min(NT, w.get("cpu_threads", NT)))
                per.append(c["algorithmic_bytes"])
                reach.append(int((ln > 0).sum()))
                if fr % 20 == 19:
                    print("%s frame %d of %d, %.0f s" % (name, fr + 1, w["frames"], time.time() - t), flush=True)
            rec.update(algorithmic_bytes_per_frame=per, reachable_per_frame=reach, algorithmic_bytes=int(np.mean(per)),
                       algorithmic_bytes_source="oracle literal mode, all %d queries of each of the first %d frames" % (w["nq"], w["frames"]))
        else:
            n = w.get("sample", w["nq"])
            s, g = oracle.synth_queries(occ, w["qseed"], n)
            if w.get("chunk"):
                # a count that takes longer than one session on the box: chunk by chunk, every finished chunk kept under
                # gpurun_out/ (a later run picks up where this one was cut)
                c, lns = dict(cells=0, pushes=0, pops=0, algorithmic_bytes=0), []
                part_dir = os.path.join(ROOT, "gpurun_out", "algo_parts")
                os.makedirs(part_dir, exist_ok=True)
                for lo in range(0, n, w["chunk"]):
                    hi = min(n, lo + w["chunk"])
                    pf = os.path.join(part_dir, "%s_%d_%d.json" % (name, lo, hi))
                    have = [q for q in (pf, os.path.join(ROOT, ".algo_parts", os.path.basename(pf))) if os.path.exists(q)]  # (.algo_parts/: parts of
                    if have:                                                                                   # an earlier session, carried to the box)
                        with open(have[0]) as f:
                            pc = json.load(f)
                    else:
                        cc, ln = count(occ, s[lo:hi], g[lo:hi], w["hchoice"], w["max_path_len"], nthreads=min(NT, w.get("cpu_threads", NT)))
                        pc = dict(cc, lens=[int(v) for v in ln])
                        with open(pf, "w") as f:
                            json.dump(pc, f)
                        print("%s queries %d .. %d counted, %.0f s" % (name, lo, hi, time.time() - t), flush=True)
                    for k in c:
                        c[k] += pc[k]
                    lns += pc["lens"]
                ln = np.array(lns)
            else:
                c, ln = count(occ, s, g, w["hchoice"], w["max_path_len"], nthreads=min(NT, w.get("cpu_threads", NT)))
            scale = w["nq"] / n
            rec.update(cells=int(c["cells"] * scale), pushes=int(c["pushes"] * scale), pops=int(c["pops"] * scale),
                       algorithmic_bytes=int(c["algorithmic_bytes"] * scale), bytes_per_query=c["algorithmic_bytes"] / n,
                       reachable_in_sample=int((ln > 0).sum()), mean_path_len=float(ln[ln > 0].mean()), max_path_len_seen=int(ln.max()),
                       algorithmic_bytes_source=("oracle literal mode, all %d queries" % n) if n == w["nq"] else
                       ("oracle literal mode on the first %d queries, extrapolated x %g" % (n, scale)))
        out[name] = rec
        print(name, {k: v for k, v in rec.items() if k not in ("describe", "algorithmic_bytes_per_frame", "reachable_per_frame")},
              "oracle literal secs", round(time.time() - t, 1), flush=True)
        if name == "c4shard":
            r4 = dict(WORKLOADS["c4"])
            r4.update(algorithmic_bytes=8 * rec["algorithmic_bytes"], bytes_per_query=rec["bytes_per_query"],
                      algorithmic_bytes_source="8 x the literal count of the first 125000 queries (c4shard)")
            out["c4"] = r4
        if name == "c5":  # the same frames with several of them in flight (bench.py --workload c5pipe)
            rp = dict(rec, frames_in_flight=PIPE_FRAMES)
            rp["describe"] = rec["describe"].replace("BASELINE config 5, streaming replan:",
                "BASELINE config 5, streaming replan with %d frames in flight (%d planner handles on the GPU take the frames in "
                "turn; every frame is the same fxjps_replan_frame call):" % (PIPE_FRAMES, PIPE_FRAMES))
            out["c5pipe"] = rp
        txt = json.dumps(out, indent=1, sort_keys=True)  # (number lists on one line each)
        txt = re.sub(r"\[\s+((?:-?\d+(?:\.\d+)?(?:e[+-]?\d+)?,\s+)*-?\d+(?:\.\d+)?(?:e[+-]?\d+)?)\s+\]", lambda m: "[" + re.sub(r"\s+", " ", m.group(1)) + "]", txt)
        with open(OUT, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
