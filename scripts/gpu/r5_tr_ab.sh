# what bounds the loader-wave transposed-convolution kernels: time with the weight-slab / input-brick DMA of all chunks
# past the second switched off (ablation library; wrong results by design)
export TMPDIR=/tmp
set -e
(cd opticalflowscivis_amd/csrc && make -j16 ablation 2>&1 | grep -E "error|warning" || true)
export FLOWSCI_HIP_LIBRARY=$PWD/opticalflowscivis_amd/csrc/ablation/libflowsci_hip_ab.so
for ab in 0 4 8 16 24 28; do
  echo "== FLOWSCI_TR_AB=$ab"
  FLOWSCI_TR_AB=$ab timeout -k 10 200 python scripts/trbench.py 2>&1 | grep -E "cin= 64 cout= 32 in= 64|cin= 32 cout= 1[12] in=128" | sed 's/miopen.*| hip/hip/'
done
