# HBM bytes fetched per launch of the F(4,3) weight-gradient kernel (rocprofv3 --pmc FETCH_SIZE; x2 on gfx950, KB units):
# product library vs a previous build (csrc/ablation/libflowsci_hip_prev.so)
export TMPDIR=/tmp
mkdir -p gpurun_out/wf
for lib in new prev; do
  rm -rf gpurun_out/wf/$lib
  if [ $lib = prev ]; then export FLOWSCI_HIP_LIBRARY=$PWD/opticalflowscivis_amd/csrc/ablation/libflowsci_hip_prev.so; else unset FLOWSCI_HIP_LIBRARY; fi
  timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/wf/$lib -- python3 tests/tools/wino_wrw_bench.py > gpurun_out/wf/$lib.log 2>&1 || { tail -3 gpurun_out/wf/$lib.log; exit 1; }
  python3 - "$lib" <<'PY'
import csv, glob, sys
lib = sys.argv[1]
f = glob.glob("gpurun_out/wf/%s/**/*counter_collection.csv" % lib, recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "wrw_wino4" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("%s: conv3d_wrw_wino4_kernel FETCH_SIZE mean %.0f KB over %d launches -> %.0f MB fetched from HBM per launch (x2 correction)" % (lib, sum(v) / len(v), len(v), 2 * sum(v) / len(v) / 1e3 * 1.024))
PY
done
