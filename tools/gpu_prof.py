#!/usr/bin/env python3
"""Per-phase cycle breakdown of the search kernel (diagnostic build libfxjps_prof.so)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--noprof" not in sys.argv:
    os.environ["FXJPS_LIB"] = os.environ.get("FX_PROF_LIB") or os.path.join(ROOT, "fuxi-planner_amd", "libfxjps_prof.so")  # (FX_PROF_LIB: the diagnostic build of a variant)
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth, _lib
W = int(os.environ.get("FX_W", "1024"))
nqs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [128]
p = fx.Planner([0])
L = _lib.load()
L.fxjps_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
occ = synth.synth_grid(W, W, 1 if W == 1024 else 2, 0.20)
C1Q = None
if os.environ.get("FX_C1"):  # BASELINE config 1: the reference map in its 256 x 256 canvas, query FX_C1 (0 .. 2) of SURVEY 8(c) alone
    import json
    z = np.load(os.path.join(ROOT, "tests", "golden", "maps_png.npz"))
    recs = json.load(open(os.path.join(ROOT, "tests", "golden", "maps_png.json")))
    nm = "-16.20-11.40_out.png"
    w_, h_ = [r for r in recs if r["map"] == nm and "canvas" not in r][0]["shape"]
    occ = np.zeros((256, 256), np.uint8)
    occ[:w_, :h_] = np.unpackbits(z[nm])[:w_ * h_].reshape(w_, h_)
    C1Q = [((0, 0), (146, 111)), ((2, 2), (140, 100)), ((5, 100), (140, 5))][int(os.environ["FX_C1"])]
p.set_grid_occ(occ)
names = ["take batch", "c:wait+probe_eval", "c:write+push+merge", "x:dirs+issue", "x:eval+diag", "x:more+bcast", "c:math+nvalid", "c:hazard", "R/M/far refills", "looptop"]
for nq in nqs:
    s, g = synth.synth_queries(occ, 1 if W == 1024 else 2, nq)
    if C1Q is not None:
        s, g = np.array([C1Q[0]], np.int32), np.array([C1Q[1]], np.int32)
        nq = 1
    if os.environ.get("FX_QIDS") and C1Q is None:  # only these queries of the stream (e.g. one heavy query on an idle chip)
        ids = [int(v) for v in os.environ["FX_QIDS"].split(",")]
        s, g = s[ids], g[ids]
        nq = len(ids)
    for rep in range(2):
        t = time.time(); off, cells, cost, st = p.plan_batch(s, g, int(os.environ.get("FX_HC", "2")), 1024 if W == 1024 else 4096); dt = time.time() - t
    tm = p.timing()
    c = (C.c_uint64 * 64)(); L.fxjps_debug_counters(p._h, c); c = list(c)
    print("nq=%d wall %.3fs kernel %.1f ms -> %.0f plans/s | pops %d pushes %d refills %d slow %d retried %d nopath %d" % (
        nq, dt, tm["search_kernel_ms"], nq / dt, c[0], c[1], c[2], c[3], tm["retried"], int((st == 0).sum())), flush=True)
    if c[4]:
        print("   batches %d: popped %.2f / batch, committed %.2f / batch, general form %.2f %%" % (c[4], c[5] / c[4], c[0] / c[4], 100.0 * c[6] / c[4]), flush=True)
    if c[4] and c[18]:
        print("   diagonal rounds %.2f / batch, rays in flight %.1f / round" % (c[18] / c[4], c[19] / c[18]), flush=True)
    if c[4] and c[20]:
        print("   R refills: one per %.1f batches, %.1f entries each, %.2f aiming rounds each, exact sort in %.1f %%" % (c[4] / c[20], c[23] / c[20], c[21] / c[20], 100.0 * c[22] / c[20]), flush=True)
    if c[4]:
        if c[29]: print("   R refill: %.0f cycles each (%d of them): read M + threshold %.0f, select + gather %.0f, sort %.0f, compaction %.0f" % (
            c[28] / c[29], c[29], c[40] / c[29], c[41] / c[29], c[42] / c[29], (c[28] - c[40] - c[41] - c[42]) / c[29]), flush=True)
        if c[26]: print("   far refill passes %d, entries scanned per pass %.0f, passes per refill %.2f, cycles per pass %.0f" % (c[26], c[27] / c[26], c[26] / max(c[2], 1), c[30] / c[26]), flush=True)
    if c[26]:
        print("   far tier: %d re-bandings down, %d up, %.0f entries gone through each; entries scanned per pop: refills %.2f, re-banding %.2f" % (
            c[43], c[44], c[47] / max(c[43] + c[44], 1), c[27] / max(c[0], 1), c[47] / max(c[0], 1)), flush=True)
    if c[4]:
        print("   nodes of a batch not committed: %.2f key rule or shared bucket, %.2f goal/parent/full bucket" % (c[24] / c[4], c[25] / c[4]), flush=True)
    if c[4] and c[33]:
        print("   lanes in use %.1f / batch, nodes without lanes %.2f / batch, behind a shared bucket %.2f / batch, (unused %.1f)" % (
            c[33] / c[4], c[34] / c[4], c[31] / c[4], 100.0 * c[32] / c[4]), flush=True)
    if c[4] and c[35]:
        print("   rays held back by the bucket detector %.3f / batch (in %.1f %% of the batches): %.1f %% really share a bucket, %.1f %% relaxed as the second of a pair; batches cut %.1f %%" % (
            c[35] / c[4], 100.0 * c[39] / c[4], 100.0 * c[36] / c[35], 100.0 * c[37] / c[35], 100.0 * c[38] / c[4]), flush=True)
    if c[4]:
        print("   duplicate nodes %.3f / batch, batches with one %.2f %%" % (c[45] / c[4], 100.0 * c[46] / c[4]), flush=True)
        print("   commit in detail, cycles / batch: stores + counts %.0f, classify %.0f, far append + R merge %.0f, M append %.0f" % (
            c[10] / c[4], c[60] / c[4], c[61] / c[4], c[62] / c[4]), flush=True)
    if c[4] and c[51]:
        print("   second rounds in %.1f %% of the batches: %.2f more nodes each; intruder on a diagonal ray %.1f %%; predictable before the table lines: unique candidate in %.1f %% of the batches, with known cell info %.1f %%, and right in %.1f %% of the second rounds" % (
            100.0 * c[51] / c[4], c[52] / c[51], 100.0 * c[53] / c[51], 100.0 * c[54] / c[4], 100.0 * c[55] / c[4], 100.0 * c[56] / c[51]), flush=True)
    if c[4] and c[50]:
        print("   open list per batch: %.2f pushes merged into R (r_merge ran in %.1f %% of the batches), %.2f appended to M, %.2f to the far tier" % (
            c[32] / c[4], 100.0 * c[50] / c[4], c[58] / c[4], (c[1] - c[32] - c[58]) / c[4]), flush=True)
        print("   loop trips per batch: rays with a key inside the batch %.2f; per second round: cells of s checked %.2f (sieve let %.1f %% of the second rounds through), pushes of s inside the batch %.2f" % (
            c[59] / c[4], c[48] / max(c[51], 1), 100.0 * c[57] / max(c[51], 1), c[49] / max(c[51], 1)), flush=True)
    tot = sum(c[8:18])
    if tot:
        print("   cycles/pop %.0f : " % (tot / max(c[0], 1)) + ", ".join("%s %.0f (%.0f%%)" % (names[k], c[8 + k] / max(c[0], 1), 100.0 * c[8 + k] / tot) for k in range(10)), flush=True)
