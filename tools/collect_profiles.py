#!/usr/bin/env python3
"""Builds the committed profile set of a round from gpurun_out/<tag>/prof_<workload>/summary.json (tools/gpu_round4.sh
profiles): profiles/<round>_<workload>_kernel_stats.csv, profiles/<round>_<workload>_pmc.csv, profiles/<round>_bench_lines.json
and profiles/hbm_traffic.json (FETCH_SIZE + WRITE_SIZE per step, tagged with the hash of the kernel source it was taken
on: bench.py reports it as roofline.traffic only while that hash matches).
Usage: python tools/collect_profiles.py <tag> <round>      e.g.  r2z r02"""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
h = hashlib.sha256()
for n in ("fxjps_kernels.hip.inc", "fxjps_maps.hip.inc", "fxjps.hip"):
    h.update(open(os.path.join(ROOT, "fuxi-planner_amd", "csrc", n), "rb").read())
sha = h.hexdigest()[:16]
wl = json.load(open(os.path.join(ROOT, "fuxi-planner_amd", "workloads.json")))
calib = json.load(open(os.path.join(ROOT, "profiles", "r02_calib_factors.json")))
traffic, lines, table = {}, {}, []
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "prof_*", "summary.json"))):
    rec = json.loads(open(f).read())
    w = rec["workload"]
    ks = rec.get("kernel_stats", [])
    if ks:
        with open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (rnd, w)), "w", newline="") as o:
            cw = csv.DictWriter(o, fieldnames=list(ks[0].keys()))
            cw.writeheader()
            cw.writerows(ks)
    c = rec.get("counters_per_step", {})
    with open(os.path.join(ROOT, "profiles", "%s_%s_pmc.csv" % (rnd, w)), "w", newline="") as o:
        cw = csv.writer(o)
        cw.writerow(["counter", "value_per_step"])
        for k in sorted(c):
            cw.writerow([k, c[k]])
        for k, v in (rec.get("registers") or {}).items():
            cw.writerow(["reg_" + k, v])
    b = rec.get("bench")
    lines[w] = b
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
        traffic[w] = {"kernel_src_sha16": sha, "FETCH_SIZE_bytes": fetch, "WRITE_SIZE_bytes": write, "hbm_bytes_per_launch": fetch + write,
                      "L2_hit_rate": (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])) if c.get("TCC_HIT_sum") else None,
                      "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum (separate passes) -- python3 bench.py --workload %s --steps 1 --warmup 0 --no-cpu-baseline" % w}
    kern = [r for r in ks if "k_search" in r.get("Name", "")]
    algo = (b or {}).get("roofline", {}).get("algorithmic_bytes_per_step", (b or {}).get("roofline", {}).get("algorithmic_bytes_per_launch"))
    table.append((w, (b or {}).get("value"), (b or {}).get("ms_per_step"), kern[0]["AverageNs"] if kern else None, algo,
                  traffic.get(w, {}).get("hbm_bytes_per_launch"), c.get("SQ_INSTS_VALU"), c.get("SQ_INSTS_SALU")))
traffic["_calibration"] = {"factors_counter_bytes_over_bytes_touched": calib, "source": "tools/calib_scatter.hip, profiles/r02_calib_scatter.txt",
                           "reading": "FETCH_SIZE counts 64 B per read request (a scattered 16-B read is one request: x4.0; a 64-B line read by one lane: x1.08; "
                                      "coalesced 16-B-per-lane streams are 128-B requests tallied at 64 B: x0.5), WRITE_SIZE counts 32-B units (a 4-B or "
                                      "16-B scattered store: 32 B; coalesced streams: x1.0).  k_search's reads are ~85 % scattered line / entry reads and ~15 % "
                                      "coalesced far-tier scans, so its true fetch bytes are about 1.1 x FETCH_SIZE; the figure reported is the uncorrected sum."}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
json.dump(lines, open(os.path.join(ROOT, "profiles", "%s_bench_lines.json" % rnd), "w"), indent=1)
print("kernel source", sha)
for t in table:
    print(t)
