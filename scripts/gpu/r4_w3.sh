# warp-family iteration loop: parity tests of the warps, then the micro-benchmark on the product library, on the
# round-3 warp kernels (same process recipe, library swapped) and on the ablation build's switches.
#   W3_VARIANTS="A=1 B=2,C=3 ..."  space-separated env sets (comma = several variables) run on the ablation build
#   W3_ONLY=<substring>            restrict w3bench to entry points containing it;  W3_SKIP_TESTS=1, W3_SKIP_R3=1
set -x
mkdir -p gpurun_out
AB=$PWD/opticalflowscivis_amd/csrc/ablation
if [ -z "$W3_SKIP_TESTS" ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_warps.py tests/test_gpu_resize.py -q -m gpu -x -k "3d or warp3 or upsample or pair" > gpurun_out/w3_tests.log 2>&1
rc=$?; tail -5 gpurun_out/w3_tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
fi
timeout -k 10 300 python scripts/w3bench.py 256 smooth $W3_ONLY > gpurun_out/w3_new.txt 2>&1 || { cat gpurun_out/w3_new.txt; exit 1; }
cat gpurun_out/w3_new.txt
if [ -z "$W3_SKIP_R3" ] && [ -f $AB/libflowsci_hip_w3r3.so ]; then
FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_w3r3.so timeout -k 10 300 python scripts/w3bench.py 256 smooth $W3_ONLY > gpurun_out/w3_r3.txt 2>&1 || exit 1
cat gpurun_out/w3_r3.txt
fi
for v in $W3_VARIANTS; do
  env ${v//,/ } FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so timeout -k 10 300 python scripts/w3bench.py 256 smooth $W3_ONLY > gpurun_out/w3_var.txt 2>&1 || { cat gpurun_out/w3_var.txt; exit 1; }
  echo "== $v"; grep -v "amdgpu.ids" gpurun_out/w3_var.txt
done
