set -x
cd $GRAFT_REPO_ROOT
python tests/tools/bench_configs.py all > gpurun_out/cfg_standalone.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf gpurun_c2 gpurun_c3
rocprofv3 --kernel-trace --stats -d gpurun_c2 -o c2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/tools/bench_configs.py c2 > $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log 2>&1
find gpurun_c2 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/c2_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats -d gpurun_c3 -o c3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/tools/bench_configs.py c3 > $GRAFT_REPO_ROOT/gpurun_out/prof_c3.log 2>&1
find gpurun_c3 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/c3_kernel_stats.csv \;
cd $GRAFT_REPO_ROOT; cat gpurun_out/cfg_standalone.txt
