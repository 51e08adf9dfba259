set -x
mkdir -p gpurun_out
AB=$PWD/opticalflowscivis_amd/csrc/ablation
for S in 192 256 320; do
  timeout -k 10 300 python scripts/w3bench.py $S smooth plain 2>&1 | grep -v amdgpu.ids
  FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_w3r3.so timeout -k 10 300 python scripts/w3bench.py $S smooth plain 2>&1 | grep -v amdgpu.ids
done
timeout -k 10 300 python scripts/w3bench.py 256 zero plain 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python scripts/w3bench.py 256 noise plain 2>&1 | grep -v amdgpu.ids
