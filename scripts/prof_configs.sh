# C2 / C3 standalone timings + rocprofv3 kernel stats (called by scripts/final_profile.sh, or alone on the GPU box).
# The profiled program follows `--` directly (python3 <script>): no wrapper process may sit between rocprofv3 and it.
set -ex
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
cd "$ROOT"
python tests/tools/bench_configs.py all > gpurun_out/cfg_standalone.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf gpurun_c2 gpurun_c3
rocprofv3 --kernel-trace --stats -d gpurun_c2 -o c2 --output-format csv -- python3 "$ROOT/tests/tools/bench_configs.py" c2 > "$ROOT/gpurun_out/prof_c2.log" 2>&1
find gpurun_c2 -name "*kernel_stats.csv" -exec cp {} "$ROOT/gpurun_out/c2_kernel_stats.csv" \;
rocprofv3 --kernel-trace --stats -d gpurun_c3 -o c3 --output-format csv -- python3 "$ROOT/tests/tools/bench_configs.py" c3 > "$ROOT/gpurun_out/prof_c3.log" 2>&1
find gpurun_c3 -name "*kernel_stats.csv" -exec cp {} "$ROOT/gpurun_out/c3_kernel_stats.csv" \;
cd "$ROOT"; cat gpurun_out/cfg_standalone.txt
