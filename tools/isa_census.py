#!/usr/bin/env python3
"""Census of the search kernel's ISA by phase of the loop (build with -DFXJPS_MARK: every PF(k) leaves a `; FXMARK k`
comment): per region between consecutive marks the instructions, the scalar-spill traffic (v_readlane / v_writelane on
the VGPRs the register allocator uses as spill slots), waits and memory instructions.  The hot path of one loop
iteration is the regions 8 -> 0 -> 3 -> 4 -> 5 -> 1 -> 6 -> 7 -> 2 (first occurrence of each: the likely blocks are laid
out in source order, the rarely taken ones -- refills, splits, the general commit -- behind them).

    tools/isa_census.py [HC TRK DIRECT]          default 2 0 1 (the headline instantiation)
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hc, trk, direct = (sys.argv[1:4] + ["2", "0", "1"])[:3] if len(sys.argv) >= 4 else ("2", "0", "1")
out = os.path.join(tempfile.gettempdir(), "fx_mark.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                       "-DFXJPS_MARK", "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "fuxi-planner_amd", "csrc", "fxjps.hip")],
                      stderr=subprocess.DEVNULL)
name = "_ZN2fx8k_searchILi%sELb%sELb%sE" % (hc, trk, direct)
lines, on = [], False
for l in open(out):
    if l.startswith(name):
        on = True
    elif on and l.startswith(".Lfunc_end"):
        break
    if on:
        lines.append(l.rstrip("\n"))
# the VGPRs that serve as SGPR spill slots: targets of v_writelane
spill = {}
for l in lines:
    m = re.match(r"\s+v_writelane_b32 (v\d+),", l)
    if m:
        spill[m.group(1)] = spill.get(m.group(1), 0) + 1
slots = {v for v, n in spill.items() if n >= 8}
marks = [(i, int(re.search(r"FXMARK (\d+)", l).group(1))) for i, l in enumerate(lines) if "FXMARK" in l]
print("kernel k_search<%s,%s,%s>: %d lines, spill VGPRs %s" % (hc, trk, direct, len(lines), sorted(slots)))
print("%-22s %6s %6s %6s %8s %8s %6s %6s %6s" % ("region (marks, line)", "insts", "valu", "salu", "readlane", "writelane", "vmem", "lds", "waitcnt"))
hot = {"insts": 0, "readlane": 0, "writelane": 0}
seen = set()
order = [8, 0, 3, 4, 5, 1, 6, 7, 2]
for (i, a), (j, b) in zip(marks, marks[1:] + [(len(lines), -1)]):
    seg = [l for l in lines[i:j] if re.match(r"\s+[a-z]", l)]
    rl = sum(1 for l in seg if re.match(r"\s+v_readlane_b32 s\d+, (v\d+)", l) and re.match(r"\s+v_readlane_b32 s\d+, (v\d+)", l).group(1) in slots)
    wl = sum(1 for l in seg if re.match(r"\s+v_writelane_b32 (v\d+)", l) and re.match(r"\s+v_writelane_b32 (v\d+)", l).group(1) in slots)
    row = (len(seg), sum(1 for l in seg if re.match(r"\s+v_", l)), sum(1 for l in seg if re.match(r"\s+s_", l)), rl, wl,
           sum(1 for l in seg if re.match(r"\s+(global|flat|buffer|scratch)_", l)), sum(1 for l in seg if re.match(r"\s+ds_", l)),
           sum(1 for l in seg if "s_waitcnt" in l))
    first = (a, b) not in seen
    seen.add((a, b))
    tag = ""
    if first and a in order and (b in order or b == 12) and j - i < 1700:
        tag = " hot"
        hot["insts"] += row[0]
        hot["readlane"] += rl
        hot["writelane"] += wl
    print("%-22s %6d %6d %6d %8d %8d %6d %6d %6d%s" % ("%d -> %d @%d" % (a, b, i), *row, tag))
print("hot regions: %d instructions, %d spill reloads (v_readlane), %d spill stores (v_writelane)" % (hot["insts"], hot["readlane"], hot["writelane"]))
# Waits for ALL outstanding vector-memory operations on the first pass through the loop body (loop head .. open_insert):
# each is a point where the iteration stands still until the previous batch's table stores have landed.  Expected: one
# behind the lane deal (mark 0: the explicit wait of search_one), the waits of the ray loops (mark 3), one in front of the
# probe evaluation (mark 5) -- none at the loop head (9), behind open_fill (8) or inside r_refill (FXRT 39 - 41).  (FXRT 42
# is followed by the whole rarely taken rest of open_fill in the layout: its count means nothing.)
first = {}
i9 = next(i for i, l in enumerate(lines) if "FXMARK 9" in l)
i12 = next(i for i, l in enumerate(lines) if "FXMARK 12" in l)
cur = None
for l in lines[i9:i12]:
    m = re.search(r"(FXMARK|FXRT) (\d+)", l)
    if m:
        cur = m.group(1) + " " + m.group(2)
    elif "s_waitcnt vmcnt(0)" in l:
        first[cur] = first.get(cur, 0) + 1
print("s_waitcnt vmcnt(0) between the loop head and open_insert, by the mark in front of them: %s" % first)
bad = [k for k in first if k in ("FXMARK 9", "FXMARK 8", "FXRT 39", "FXRT 40", "FXRT 41")]
print("drain-everything waits on the hot path: %s" % (bad if bad else "none"))

