#!/bin/bash
# The round's measurement batch on one MI355X (run from the repo root through gpurun):
#   HIP-event bench line -> two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/rNN_pmc_traffic.json
#   -> the bench line again (it reads roofline.traffic from that file, cpu_baseline leg included)
#   -> rocprofv3 --kernel-trace --stats of the same command + the last step's per-kernel breakdown.
# Results land in gpurun_out/final/; copy what is to be judged into profiles/.
set -e
export TMPDIR=/tmp
R=${1:-r03}
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python bench.py --no-cpu-baseline --no-configs > $O/bench_events.json 2> $O/bench_events.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-configs > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-configs > $O/w.log 2>&1
python scripts/pmc_traffic.py "$(find $O/f -name '*counter_collection.csv' | head -1)" \
    "$(find $O/w -name '*counter_collection.csv' | head -1)" $O/bench_events.json > $O/pmc_traffic.json
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
rm -rf $O/f $O/w
python bench.py > $O/bench.json 2> $O/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --no-cpu-baseline --no-configs > $O/under_rocprof.json 2> $O/kt.log
python scripts/last_step_breakdown.py "$(find $O/kt -name '*kernel_trace.csv' | head -1)" 70 > $O/last_step.txt
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
rm -rf $O/kt
tail -1 $O/bench.json | cut -c1-400
head -12 $O/last_step.txt
# the C2 / C3 steps stand-alone + their kernel stats
bash scripts/prof_configs.sh > $O/prof_configs.log 2>&1 || true
