# persistent 2-D Winograd trunk kernel: parity tests, then every fused form on the product library and on the round-3 kernel
# (ablation build, FLOWSCI_WINO2D_R3=1): CRCs must match; then the ablation variants of tests/tools/wino_bench.py
set -x
mkdir -p gpurun_out
AB=$PWD/opticalflowscivis_amd/csrc/ablation/libflowsci_hip_ab.so
timeout -k 10 600 python -m pytest tests/test_gpu_wino.py -q -m gpu -x > gpurun_out/wino_tests.log 2>&1; rc=$?; tail -5 gpurun_out/wino_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python scripts/wino2d_ab.py > gpurun_out/wino2d_ps.txt 2>&1 || { cat gpurun_out/wino2d_ps.txt; exit 1; }
FLOWSCI_WINO2D_R3=1 FLOWSCI_HIP_LIBRARY=$AB timeout -k 10 200 python scripts/wino2d_ab.py > gpurun_out/wino2d_r3.txt 2>&1 || { cat gpurun_out/wino2d_r3.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/wino2d_ps.txt; grep -v amdgpu.ids gpurun_out/wino2d_r3.txt
WINO_DBGS="${WINO_DBGS:-0 1 2 3 4}" bash scripts/gpu/r4_wino_dbg.sh
