# the split-bf16 transposed-convolution kernel (csrc/convtr_s3.hpp) against the fp32-MFMA class kernel, same box; then what
# bounds it (ablation library; FLOWSCI_TR_AB bits: 4 no epilogue, 32 no conversion, 64 no matrix-wave work, 128 operand reads
# without MFMAs -- wrong results by design)
export TMPDIR=/tmp
set -e
(cd opticalflowscivis_amd/csrc && make -j16 ablation 2>&1 | grep -E "error|warning" || true)
echo "== product"
timeout -k 10 200 python scripts/trbench.py 2>&1 | grep -E "cin= 64 cout= 32|cin= 32 cout= 1[12] in=128" | sed 's/miopen.*| hip/hip/'
export FLOWSCI_HIP_LIBRARY=$PWD/opticalflowscivis_amd/csrc/ablation/libflowsci_hip_ab.so
echo "== FLOWSCI_TR_NO_S3=1 (fp32 MFMA)"
FLOWSCI_TR_NO_S3=1 timeout -k 10 200 python scripts/trbench.py 2>&1 | grep -E "cin= 64 cout= 32|cin= 32 cout= 1[12] in=128" | sed 's/miopen.*| hip/hip/'
for ab in 4 32 64 128 36 100; do
  echo "== FLOWSCI_TR_AB=$ab"
  FLOWSCI_TR_AB=$ab timeout -k 10 200 python scripts/trbench.py 2>&1 | grep -E "cin= 64 cout= 32 in= 64|cin= 32 cout= 12 in=128" | sed 's/miopen.*| hip/hip/'
done
