#!/bin/bash
# The round's measurement batch on one MI355X (run from the repo root through gpurun):
#   HIP-event bench line -> two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/rNN_pmc_traffic.json
#   -> the bench line again (it reads roofline.traffic from that file, cpu_baseline leg included)
#   -> rocprofv3 --kernel-trace --stats of the same command + the last step's per-kernel breakdown.
# Results land in gpurun_out/final/; copy what is to be judged into profiles/.
set -e
export TMPDIR=/tmp
R=${1:-r05}
PART=${2:-all}   # 1 = bench line, PMC traffic, kernel trace; 2 = everything after (a gpurun call is limited to 20 minutes)
O=gpurun_out/final
if [ "$PART" != 2 ]; then
rm -rf $O && mkdir -p $O
python bench.py --no-cpu-baseline --no-configs > $O/bench_events.json 2> $O/bench_events.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python bench.py --eager --steps 2 --warmup 2 --no-cpu-baseline --no-configs --no-bench-parity > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python bench.py --eager --steps 2 --warmup 2 --no-cpu-baseline --no-configs --no-bench-parity > $O/w.log 2>&1
python scripts/pmc_traffic.py "$(find $O/f -name '*counter_collection.csv' | head -1)" \
    "$(find $O/w -name '*counter_collection.csv' | head -1)" $O/bench_events.json > $O/pmc_traffic.json
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
rm -rf $O/f $O/w
python bench.py > $O/bench.json 2> $O/bench.log
# the same command as the bench line (N = 1: the step replayed from one HIP graph, then the eager pass behind it); should the
# tracer lose the kernels of a replayed graph, the eager form of the command is traced instead
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --no-cpu-baseline --no-configs --no-bench-parity > $O/under_rocprof.json 2> $O/kt.log \
  || { rm -rf $O/kt; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --eager --no-cpu-baseline --no-configs --no-bench-parity > $O/under_rocprof.json 2> $O/kt.log; }
python scripts/last_step_breakdown.py "$(find $O/kt -name '*kernel_trace.csv' | head -1)" 70 > $O/last_step.txt
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
rm -rf $O/kt
tail -1 $O/bench.json | cut -c1-400
head -12 $O/last_step.txt
fi
if [ "$PART" = 1 ]; then exit 0; fi
mkdir -p $O
# the C2 / C3 steps stand-alone + their kernel stats
bash scripts/prof_configs.sh > $O/prof_configs.log 2>&1 || echo "prof_configs.sh FAILED (see $O/prof_configs.log)"
# train.py at the bench's size (device-generated triplets)
python -m opticalflowscivis_amd.flow3d.train --dataset droplet3d --size 256 --samples 24 --batch_size 2 --epoch 2 --mode train \
    --log_every 4 --log_path /tmp/tl256 > $O/train256.txt 2>&1 || true
# the DDP + RCCL code path rehearsed with ONE rank (process group, DDP hooks, bucket views) next to the plain N = 1 step
{ echo "# plain N = 1:"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-bench-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f  value %.3f  step_driver %s' % (d['ms_per_step'], d['value'], d['step_driver']))";
  echo "# FLOWSCI_BENCH_FORCE_DDP=1 (RCCL process group of one rank, DistributedDataParallel):"; FLOWSCI_BENCH_FORCE_DDP=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-bench-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f  value %.3f  step_driver %s  backend %s' % (d['ms_per_step'], d['value'], d['step_driver'], d['ranks']['backend']))"; } > $O/ddp_one_rank.txt 2>&1 || true
cat $O/ddp_one_rank.txt
# the trilinear-warp family stand-alone (ms, algorithmic GB/s, checksums) and the C2 / C3 hot-path kernel tables
python scripts/w3bench.py 256 smooth 2>&1 | grep -v amdgpu.ids > $O/w3bench.txt || true
python scripts/c3_kernels.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $O/c2_c3_kernels.txt || true
# the Winograd-domain trunk kernels against the direct ones, and the weight gradient's ablation builds: these switches
# exist only in the measurement build of the library (make ablation), loaded through FLOWSCI_HIP_LIBRARY
AB=$PWD/opticalflowscivis_amd/csrc/ablation/libflowsci_hip_ab.so
make -C opticalflowscivis_amd/csrc ablation -j16 > $O/make_ablation.log 2>&1 || tail -5 $O/make_ablation.log  # (not part of build(); does not travel)
if [ -f "$AB" ]; then
export FLOWSCI_HIP_LIBRARY=$AB
{ python tests/tools/wino_bench.py; FLOWSCI_FWD_NO_WINO2D=1 python tests/tools/wino_bench.py; FLOWSCI_FWD_NO_WINO4=1 python tests/tools/wino_bench.py; FLOWSCI_FWD_NO_WINO=1 python tests/tools/wino_bench.py;
  python tests/tools/wino_wrw_bench.py; FLOWSCI_WRW_NO_WINO4=1 python tests/tools/wino_wrw_bench.py; FLOWSCI_WRW_NO_WINO=1 python tests/tools/wino_wrw_bench.py;
  FLOWSCI_WINO_DBG=1 python tests/tools/wino_wrw_bench.py; FLOWSCI_WINO_DBG=2 python tests/tools/wino_wrw_bench.py;
  FLOWSCI_WINO_DBG=3 python tests/tools/wino_bench.py; FLOWSCI_WINO_DBG=4 python tests/tools/wino_bench.py; } 2>&1 \
    | grep -E "wmode|wrw:" > $O/wino_kernels.txt || true
# the direct kernels of the trunk shapes still pass their tests when the Winograd forms are switched off
FLOWSCI_FWD_NO_WINO=1 FLOWSCI_WRW_NO_WINO=1 python -m pytest tests/test_gpu_losses.py tests/test_gpu_scale.py -q -k "(conv or res_unit or head or 256) and not drift and not b2_at" \
    > $O/direct_kernels_tests.log 2>&1 || true
tail -3 $O/direct_kernels_tests.log
unset FLOWSCI_HIP_LIBRARY
fi
