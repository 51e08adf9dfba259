"""Mean of every counter per kernel from a rocprofv3 --pmc counter_collection.csv.
usage: pmc_summary.py counter_collection.csv [kernel-name substring | all] [characters of the kernel name kept, default 48]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
NAME = int(sys.argv[3]) if len(sys.argv) > 3 else 48
for r in rows:
    # (launches of one kernel with different grids are different problems: keep them apart)
    agg[(r["Kernel_Name"][:NAME] + " grid " + r.get("Grid_Size", "?"), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if "conv3d" in k or "corr" in k or (len(sys.argv) > 2 and (sys.argv[2] == "all" or sys.argv[2] in k)):
        print("%-*s %-32s %14.0f (n=%d)" % (NAME + 14, k, c, sum(v) / len(v), len(v)))
