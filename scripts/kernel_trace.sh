set -x
export TMPDIR=/tmp
O=gpurun_out/kt1
rm -rf $O && mkdir -p $O
python -m pytest tests/test_gpu_wprep.py -x -q > $O/wprep_tests.log 2>&1; tail -2 $O/wprep_tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --no-cpu-baseline --no-configs > $O/under_rocprof.json 2> $O/kt.log
python scripts/last_step_breakdown.py "$(find $O/kt -name '*kernel_trace.csv' | head -1)" 90 > $O/last_step.txt
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
rm -rf $O/kt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_shape.hip -o /tmp/mfma_shape 2>/dev/null && /tmp/mfma_shape > $O/mfma_shape.txt; cat $O/mfma_shape.txt
head -5 $O/last_step.txt
