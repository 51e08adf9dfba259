set -x
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/full_tests.log 2>&1
rc=$?; tail -15 gpurun_out/full_tests.log; exit $rc
