"""Per-kernel time of the LAST train step in a rocprofv3 --kernel-trace CSV (steady state: the warm-up
steps contain MIOpen's find-step kernels).  Usage: python scripts/last_step_breakdown.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "multi_tensor" in n]  # the fused AdamW kernels end a step
ends, prev = [], None
for i in idx:
    if prev is None or i - prev > 50:
        ends.append(i)
    prev = i
a, b = ends[-2], ends[-1]
seg = rows[a:b]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"][:120]][0] += d
    agg[r["Kernel_Name"][:120]][1] += 1
busy = sum(v[0] for v in agg.values()) / 1e6
print("last step: span %.2f ms, busy %.2f ms, %d launches" % (span, busy, len(seg)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%8.2f ms  n=%4d  avg=%7.3f  %s" % (v[0] / 1e6, v[1], v[0] / v[1] / 1e6, k[:100]))
