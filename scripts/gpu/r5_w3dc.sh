# slices per workgroup of the row-cache kernels: repeated same-box runs (ablation build, FLOWSCI_W3_RC_DC)
mkdir -p gpurun_out
AB=$PWD/opticalflowscivis_amd/csrc/ablation
make -C opticalflowscivis_amd/csrc ablation -j16 > gpurun_out/make_ablation.log 2>&1 || { tail -20 gpurun_out/make_ablation.log; exit 1; }
for rep in 1 2; do
for dc in ${W3_DCS:-16 32 64}; do
  echo "== dc $dc (rep $rep)"
  FLOWSCI_W3_RC_DC=$dc FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so timeout -k 10 300 python scripts/w3bench.py 256 ${W3_KIND:-smooth} acc 2>&1 | grep "^fs_"
done
done
