#!/usr/bin/env python3
"""Register / scratch / LDS figures of every k_search instantiation from `make -C fuxi-planner_amd resource-usage`
(-Rpass-analysis=kernel-resource-usage).  Usage: make -C fuxi-planner_amd resource-usage 2>&1 | python tools/resource_usage.py"""
import re, sys
cur, rows = None, {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line) or re.search(r' Name: (\S+)', line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+(.*?): (\S+) \[-Rpass', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
keys = ('TotalSGPRs', 'VGPRs', 'AGPRs', 'SGPRs Spill', 'VGPRs Spill', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]')
print("kernel," + ",".join(keys))
for k, v in sorted(rows.items()):
    if 'k_search' in k:
        m = re.search(r'k_searchILi(\d)ELb(\d)ELb(\d)E', k)
        name = "k_search<%s,%s,%s>" % (m.group(1), "true" if m.group(2) == "1" else "false", "true" if m.group(3) == "1" else "false") if m else k
        print(name + "," + ",".join(v.get(a, "") for a in keys))
