# round 5 warp-family loop: parity tests of the 3-D warps, then scripts/w3bench.py on the product library and, on the
# ablation build, with the round-4 kernels (FLOWSCI_W3_RC=0) and the measurement forms of the row-cache kernels.
#   W3_VARIANTS="A=1 B=2,C=3 ..."  space-separated env sets (comma = several variables) run on the ablation build
#   W3_ONLY=<plain|...>, W3_SKIP_TESTS=1, W3_KIND=smooth|noise|zero|small
set -x
mkdir -p gpurun_out
AB=$PWD/opticalflowscivis_amd/csrc/ablation
make -C opticalflowscivis_amd/csrc ablation -j16 > gpurun_out/make_ablation.log 2>&1 || { tail -20 gpurun_out/make_ablation.log; exit 1; }
KIND=${W3_KIND:-smooth}
if [ -z "$W3_SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_warps.py tests/test_gpu_resize.py -q -m gpu -x -k "3d or warp3 or upsample or pair or row_cache" > gpurun_out/w3_tests.log 2>&1
rc=$?; tail -15 gpurun_out/w3_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
fi
timeout -k 10 300 python scripts/w3bench.py 256 $KIND $W3_ONLY > gpurun_out/w3_new.txt 2>&1 || { cat gpurun_out/w3_new.txt; exit 1; }
cat gpurun_out/w3_new.txt
for v in FLOWSCI_W3_RC=0 $W3_VARIANTS; do
  env ${v//,/ } FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so timeout -k 10 300 python scripts/w3bench.py 256 $KIND $W3_ONLY > gpurun_out/w3_var.txt 2>&1 || { cat gpurun_out/w3_var.txt; exit 1; }
  echo "== $v"; grep -v "amdgpu.ids" gpurun_out/w3_var.txt
done
