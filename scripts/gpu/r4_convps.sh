# persistent loader-wave convolution kernels: parity tests of the convolution layer, per-brick overheads, the bench line
set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_wino.py tests/test_gpu_scale.py tests/test_gpu_e2e.py tests/test_gpu_det.py -q -m gpu -x > gpurun_out/convps_tests.log 2>&1; rc=$?; tail -4 gpurun_out/convps_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
python scripts/brick_overhead.py 2>&1 | grep -v amdgpu.ids
python bench.py --no-cpu-baseline --no-configs > gpurun_out/bench_ps.json 2> gpurun_out/bench_ps.err || { tail -5 gpurun_out/bench_ps.err; exit 1; }
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_ps.json') if x.startswith('{')][-1]
d=json.loads(l)
print("value %.3f  ms_per_step %.3f  roofline %s frac %.4f avg_ms %s  parity %s" % (d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('avg_ms'), d.get('parity_at_bench_size',{}).get('ok')))
for k in d.get('kernels', [])[:12]: print(k)
PY
