# profiles/r04_wino2d_stamps.txt: cycle stamps of the persistent 2-D Winograd trunk kernel (full and ablation modes), the
# per-form A/B against the round-3 kernel (times + CRCs), the priority A/B; plus the new bitwise test
set -x
mkdir -p gpurun_out
AB=$PWD/opticalflowscivis_amd/csrc/ablation
timeout -k 10 600 python -m pytest tests/test_gpu_wino.py -q -m gpu -x -k "bit_identical or superseded" > gpurun_out/wino_ab_test.log 2>&1; rc=$?; tail -3 gpurun_out/wino_ab_test.log
if [ $rc -ne 0 ]; then exit $rc; fi
O=gpurun_out/wino2d_stamps.txt
{
echo "# conv3d_wino2d_ps_kernel<0, 16>, 64 -> 64 channels, 2 x 64^3 (scripts/w2_stamps.py on csrc/ablation/libflowsci_hip_w2s.so:"
echo "# s_memtime sums per wave of one workgroup; 8 bricks x 32 periods; the stamps themselves cost ~8 % of the launch)"
for d in 0 1 2 3 4; do
  echo "## FLOWSCI_WINO_DBG=$d  (0 full, 1 no slab copies, 2 no input loads / transforms, 3 neither, 4 loaders alone)"
  FLOWSCI_WINO_DBG=$d FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_w2s.so timeout -k 10 100 python scripts/w2_stamps.py 2>&1 | grep -v amdgpu.ids
done
echo
echo "# every fused form, product library (persistent kernel): ms per launch incl. the weight re-layout launch, CRC-32 of the outputs"
timeout -k 10 200 python scripts/wino2d_ab.py 2>&1 | grep -v amdgpu.ids
echo "# the same on the round-3 kernel (ablation build, FLOWSCI_WINO2D_R3=1), same box"
FLOWSCI_WINO2D_R3=1 FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so timeout -k 10 200 python scripts/wino2d_ab.py 2>&1 | grep -v amdgpu.ids
echo
echo "# wave priorities (ablation build, FLOWSCI_WINO2D_AB: 1 = loader waves raised (product), 0 = none, 2 = matrix waves raised)"
for v in 1 0 2; do
  echo "## FLOWSCI_WINO2D_AB=$v"
  FLOWSCI_WINO2D_AB=$v FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so timeout -k 10 200 python scripts/wino2d_ab.py 2>&1 | grep -E "64\^3 (plain|wmode1 |dprelu )"
done
} > $O 2>&1
tail -5 $O
