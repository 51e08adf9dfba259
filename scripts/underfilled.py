"""Launches of the last step of a rocprofv3 kernel trace that cannot fill the chip (fewer workgroups than CUs),
by time: python scripts/underfilled.py <kernel_trace.csv> [min_us]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "multi_tensor" in r["Kernel_Name"]]
last_end = ends[-1]
prev_end = max(i for i in ends if i < last_end - 50)
step = rows[prev_end + 1:last_end + 1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    n = grid // max(wg, 1)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n < 512:
        k = (r["Kernel_Name"][:70], n)
        agg[k][0] += 1; agg[k][1] += d
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
for (k, n), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if d >= thr:
        print("%8.1f us  n=%3d  workgroups=%4d  %s" % (d, c, n, k))
