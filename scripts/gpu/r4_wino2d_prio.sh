# loader-wave priority A/B on the ablation build (FLOWSCI_WINO2D_AB bit 0), then the product library
AB=$PWD/opticalflowscivis_amd/csrc/ablation/libflowsci_hip_ab.so
for v in 0 1; do
  echo "== FLOWSCI_WINO2D_AB=$v"
  for d in ${WINO_DBGS:-0 1}; do
  FLOWSCI_WINO2D_AB=$v FLOWSCI_WINO_DBG=$d FLOWSCI_HIP_LIBRARY=$AB timeout -k 10 200 python tests/tools/wino_bench.py 2>&1 | grep "ms/launch" | cut -c1-110 || exit 1
  done
done
timeout -k 10 200 python scripts/wino2d_ab.py 2>&1 | grep -v amdgpu.ids
