# wave priorities in the persistent 2-D Winograd kernel (ablation build, FLOWSCI_WINO2D_AB: bit 0 loader waves raised -- the
# product's setting --, bit 1 matrix waves raised)
AB=$PWD/opticalflowscivis_amd/csrc/ablation/libflowsci_hip_ab.so
for v in 1 0 2 1 0 2; do
  echo "== FLOWSCI_WINO2D_AB=$v"
  FLOWSCI_WINO2D_AB=$v FLOWSCI_HIP_LIBRARY=$AB timeout -k 10 200 python scripts/wino2d_ab.py 2>&1 | grep -E "64\^3 (plain|wmode1 |dprelu )" || exit 1
done
