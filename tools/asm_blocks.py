#!/usr/bin/env python3
"""Basic blocks of one kernel's assembly (-S output, ideally a -DFXJPS_MARK build) in layout order: label, loop depth, the
mark region it lies in, instruction mix, how it ends.  Reading aid for tools/isa_census.py's regions: which blocks of a
region are the straight-line hot path and which are rarely taken code laid out inside it.
    tools/asm_blocks.py <kernel.s> [first line] [last line]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
blocks, cur, mark = [], None, "-"
def newblock(name, i, depth):
    global cur
    cur = {"name": name, "line": i, "depth": depth, "mark": mark, "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "wait": 0, "rl": 0, "wl": 0, "n": 0, "end": "", "marks": []}
    blocks.append(cur)
newblock("entry", 0, "")
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l) or re.match(r"^; %bb\.(\d+):\s*(;.*)?$", l)
    if m:
        d = re.search(r"Depth=(\d+)", l)
        h = re.search(r"Header=(BB\d+_\d+)", l)
        newblock(m.group(1), i, (h.group(1) + "/" + d.group(1)) if d and h else "")
        continue
    mm = re.search(r"; (FXMARK|FXRT) (\d+)", l)
    if mm:
        mark = ("M" if mm.group(1) == "FXMARK" else "R") + mm.group(2)
        cur["marks"].append(mark)
        continue
    t = l.strip()
    if not re.match(r"^[a-z]", t) or t.startswith(";"):
        continue
    op = t.split()[0]
    cur["n"] += 1
    if op.startswith("v_readlane") and re.search(r"v11[23]", t): cur["rl"] += 1
    if op.startswith("v_writelane") and re.search(r"v11[23]", t): cur["wl"] += 1
    if op.startswith("v_"): cur["valu"] += 1
    elif op == "s_waitcnt": cur["wait"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): cur["end"] += " " + op.replace("s_cbranch_", "cb_").replace("s_branch", "b") + ">" + t.split()[-1].replace(".LBB", "")
    elif op.startswith("s_"): cur["salu"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.split("_")[0] in ("global", "flat", "buffer", "scratch"): cur["vmem"] += 1
print("%-12s %6s %-12s %-5s %4s %4s %4s %3s %3s %3s %3s %3s  %s" % ("block", "line", "loop/depth", "mark", "n", "valu", "salu", "lds", "vm", "wt", "rl", "wl", "ends / marks inside"))
for b in blocks:
    if b["line"] < lo or b["line"] > hi or b["n"] == 0 and not b["marks"]:
        continue
    print("%-12s %6d %-12s %-5s %4d %4d %4d %3d %3d %3d %3d %3d  %s %s" % (b["name"].replace(".LBB", ""), b["line"], b["depth"], b["mark"], b["n"], b["valu"], b["salu"], b["lds"], b["vmem"], b["wait"], b["rl"], b["wl"], b["end"], ",".join(b["marks"])))
