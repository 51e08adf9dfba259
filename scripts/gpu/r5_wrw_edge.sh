# A/B of the F(4,3) weight-gradient kernel's edge-column reads (dword at the channel pitch vs component of the aligned
# 16-byte slot): previous build (csrc/_prev/libflowsci_hip_prev.so) against the product library, same box.
export TMPDIR=/tmp
set -e
timeout -k 10 300 python -m pytest tests/test_gpu_wino.py -x -q -m gpu 2>&1 | tail -2
for lib in prev new prev new; do
  if [ $lib = prev ]; then export FLOWSCI_HIP_LIBRARY=$PWD/opticalflowscivis_amd/csrc/_prev/libflowsci_hip_prev.so; else unset FLOWSCI_HIP_LIBRARY; fi
  echo "== $lib"
  timeout -k 10 200 python scripts/wrwbench.py 2>&1 | grep -E "convblock|block0"
done
