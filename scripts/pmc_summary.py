"""Mean of every counter per kernel from a rocprofv3 --pmc counter_collection.csv."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    # (launches of one kernel with different grids are different problems: keep them apart)
    agg[(r["Kernel_Name"][:48] + " grid " + r.get("Grid_Size", "?"), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if "conv3d" in k or "corr" in k or (len(sys.argv) > 2 and sys.argv[2] in k):
        print("%-62s %-28s %14.0f (n=%d)" % (k, c, sum(v) / len(v), len(v)))
