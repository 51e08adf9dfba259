# HBM bytes fetched per launch of every weight-gradient kernel at the step's layer shapes (scripts/wrwbench.py under
# rocprofv3 --pmc FETCH_SIZE; x2 on gfx950, KB units) next to the algorithmic bytes
export TMPDIR=/tmp
rm -rf gpurun_out/wfa; mkdir -p gpurun_out/wfa
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/wfa/r -- python3 scripts/wrwbench.py > gpurun_out/wfa/log.txt 2>&1 || { tail -3 gpurun_out/wfa/log.txt; exit 1; }
grep "TFLOP" gpurun_out/wfa/log.txt
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/wfa/r/**/*counter_collection.csv", recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE" or "wrw" not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"][:90], r["Grid_Size"])
    d.setdefault(k, []).append(float(r["Counter_Value"]))
for (k, g), v in d.items():
    print("%-90s grid %-9s n=%2d  fetched %.0f MB per launch" % (k, g, len(v), 2 * sum(v) / len(v) * 1.024 / 1e3))
PY
