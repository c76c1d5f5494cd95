#!/usr/bin/env python3
"""Per-kernel totals of the LAST step in a rocprofv3 kernel trace of bench.py.
usage: step_breakdown.py <kernel_trace.csv> [top_n] [marker]   marker: substring of a kernel launched exactly once per step
(default: the forward's markers; for --mode train use matcher_cost_kernel)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[3] if len(sys.argv) > 3 else "resize_pyramid"  # one per forward (the level-resolution mask route)
marks = [r for r in rows if marker in r["Kernel_Name"]]
if len(marks) >= 2:
    t0, t1 = int(marks[-2]["Start_Timestamp"]), int(marks[-1]["Start_Timestamp"])
else:  # no pyramid launch in the trace: every prediction at full resolution -- 10 mask builds per forward close a step
    marks = [r for r in rows if "attn_mask_build" in r["Kernel_Name"]]
    t0, t1 = int(marks[-10]["End_Timestamp"]), int(marks[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and int(r["Start_Timestamp"]) < t1]
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    agg[r["Kernel_Name"]][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"]][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"last step: wall {(t1 - t0) / 1e6:.2f} ms, kernel sum {tot / 1e6:.2f} ms, {len(sel)} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{k[:100]:100s} n={v[1]:4d} ms={v[0] / 1e6:7.3f} avg_us={v[0] / v[1] / 1e3:8.1f}")
