#!/usr/bin/env python3
"""Which hardware queue do the search launches of each stream land on, and do launches that share one overlap?
tools/queue_map.py <rocprofv3 --kernel-trace dir> [kernel substring]   (frames in flight)"""
import csv, glob, sys
from collections import defaultdict
pat = sys.argv[2] if len(sys.argv) > 2 else "k_search"
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), int(r["Stream_Id"])))
rows.sort()
if not rows:
    sys.exit("no %s dispatches in %s" % (pat, sys.argv[1]))
q_of = defaultdict(set)
for s, e, q, st in rows:
    q_of[st].add(q)
by_q = defaultdict(list)
for s, e, q, st in rows:
    by_q[q].append((s, e, st))
print("%d dispatches of *%s* on %d streams, %d hardware queues" % (len(rows), pat, len(q_of), len(by_q)))
for q in sorted(by_q):
    sts = sorted({st for _, _, st in by_q[q]})
    # overlap between consecutive dispatches of DIFFERENT streams on this queue
    ov = gaps = 0
    ls = sorted(by_q[q])
    for (s0, e0, a), (s1, e1, b) in zip(ls, ls[1:]):
        if a != b:
            if s1 < e0:
                ov += 1
            else:
                gaps += 1
    print("  queue %2d: streams %s  %4d dispatches; consecutive ones of different streams: %d overlap, %d run one after the other" % (q, sts, len(ls), ov, gaps))
# how many search kernels run at once, time-weighted
ev = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
cur, last, acc, tot = 0, ev[0][0], defaultdict(int), 0
for t, d in ev:
    acc[cur] += t - last
    tot += t - last
    last = t
    cur += d
print("searches in flight (share of the traced span):", {k: round(v / tot, 3) for k, v in sorted(acc.items()) if v / tot >= 0.005})
d = [(e - s) / 1e6 for s, e, _, _ in rows]
d.sort()
print("search kernel ms: median %.1f, p90 %.1f, max %.1f" % (d[len(d) // 2], d[int(len(d) * 0.9)], d[-1]))
