# split-bf16 k4 s2 forward kernel: its tests, then the conv bench of the step's k4 layers on the product library vs the
# fp32 kernels (ablation build, FLOWSCI_FWD_NO_S3=1)
mkdir -p gpurun_out
AB=$PWD/opticalflowscivis_amd/csrc/ablation
make -C opticalflowscivis_amd/csrc ablation -j16 > gpurun_out/make_ablation.log 2>&1 || { tail -20 gpurun_out/make_ablation.log; exit 1; }
if [ -z "$S3_SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_losses.py -q -m gpu -x -k "split_bf16 or conv3d_fwd_mfma or conv0_reads or head_fused or conv_prelu_fused" > gpurun_out/s3_tests.log 2>&1
rc=$?; tail -25 gpurun_out/s3_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
fi
timeout -k 10 300 python scripts/s3bench.py > gpurun_out/s3bench_new.txt 2>&1; grep -v amdgpu gpurun_out/s3bench_new.txt
FLOWSCI_FWD_NO_S3=1 FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so timeout -k 10 300 python scripts/s3bench.py > gpurun_out/s3bench_old.txt 2>&1; grep -v amdgpu gpurun_out/s3bench_old.txt
