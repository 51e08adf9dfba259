# iteration loop for the persistent 2-D Winograd kernel: parity tests (optional), stamps, A/B CRCs + times
set -x
mkdir -p gpurun_out
AB=$PWD/opticalflowscivis_amd/csrc/ablation
if [ -n "$W2_TESTS" ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_wino.py -q -m gpu -x > gpurun_out/wino_tests.log 2>&1; rc=$?; tail -3 gpurun_out/wino_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
fi
for d in ${W2_STAMP_DBGS:-0}; do
FLOWSCI_WINO_DBG=$d FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_w2s.so timeout -k 10 100 python scripts/w2_stamps.py 2>&1 | grep -v amdgpu.ids || exit 1
done
timeout -k 10 200 python scripts/wino2d_ab.py 2>&1 | grep -v amdgpu.ids || exit 1
if [ -n "$W2_R3" ]; then
FLOWSCI_WINO2D_R3=1 FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so timeout -k 10 200 python scripts/wino2d_ab.py 2>&1 | grep -v amdgpu.ids || exit 1
fi
