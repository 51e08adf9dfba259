"""Mean of every counter per kernel from a rocprofv3 --pmc counter_collection.csv."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if "conv3d" in k:
        print("%-62s %-28s %14.0f (n=%d)" % (k, c, sum(v) / len(v), len(v)))
