set -x
mkdir -p gpurun_out
python -c "import sys; sys.argv=['x']; import bench; print('KFD', bench.kfd_gpus())" > gpurun_out/kfd.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_wprep.py tests/test_gpu_c1.py tests/test_gpu_wino.py -q -m gpu -x > gpurun_out/t1a.log 2>&1
rc=$?; tail -5 gpurun_out/t1a.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py -q -m gpu -s -k "drift or 256_vs_reference" > gpurun_out/t1b.log 2>&1
rc=$?; tail -30 gpurun_out/t1b.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > gpurun_out/b1.json 2> gpurun_out/b1.log
rc=$?; tail -3 gpurun_out/b1.log; cut -c1-1500 gpurun_out/b1.json
exit $rc
