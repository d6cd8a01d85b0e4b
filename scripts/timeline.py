#!/usr/bin/env python3
"""GPU busy fraction and per-kernel gaps from a rocprofv3 --kernel-trace csv (kernel_trace.csv):
    python3 scripts/timeline.py <kernel_trace.csv>
Prints, for the last 60 % of the trace (steady state): wall time, union of kernel intervals, idle time, and the idle
time attributed to the kernel that FOLLOWS each gap."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")) for r in rows)
t0, t1 = ev[0][0], ev[-1][1]
lo = t0 + int(0.4 * (t1 - t0))
ev = [e for e in ev if e[0] >= lo]
wall = ev[-1][1] - ev[0][0]
busy, cur_end, gaps = 0, ev[0][0], defaultdict(lambda: [0, 0])
for s, e, n in ev:
    if s > cur_end:
        gaps[n][0] += s - cur_end
        gaps[n][1] += 1
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
ksum = defaultdict(int)
for s, e, n in ev:
    ksum[n] += e - s
print("wall %.1f ms  busy %.1f ms (%.1f %%)  idle %.1f ms  sum of kernel durations %.1f ms" % (wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, sum(ksum.values()) / 1e6))
for n, (g, c) in sorted(gaps.items(), key=lambda x: -x[1][0])[:12]:
    print("  idle before %-28s %8.1f us in %5d gaps (%.1f us each)" % (n, g / 1e3, c, g / 1e3 / max(c, 1)))
for n, v in sorted(ksum.items(), key=lambda x: -x[1])[:12]:
    print("  kernel %-28s %8.1f ms (%.1f %% of wall)" % (n, v / 1e6, 100.0 * v / wall))
