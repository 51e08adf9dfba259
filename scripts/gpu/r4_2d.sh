# 2-D rows: parity tests of the warps / C3 batch / UPFlow levels, then the C2 / C3 steps with their hot-path kernel table
set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_warps.py tests/test_gpu_c3_batch.py tests/test_gpu_losses.py -q -m gpu -x > gpurun_out/t2d.log 2>&1
rc=$?; tail -5 gpurun_out/t2d.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python tests/tools/bench_configs.py all > gpurun_out/cfg.txt 2>&1
cat gpurun_out/cfg.txt | tail -40
