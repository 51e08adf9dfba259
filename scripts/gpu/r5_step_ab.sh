# the 256^3 bench step on the product library vs the ablation build with a switch set, alternating, same box
#   STEP_AB_ENV="FLOWSCI_W3_RC=0"  (variables for the ablation arm)
mkdir -p gpurun_out
AB=$PWD/opticalflowscivis_amd/csrc/ablation
make -C opticalflowscivis_amd/csrc ablation -j16 > gpurun_out/make_ablation.log 2>&1 || { tail -20 gpurun_out/make_ablation.log; exit 1; }
for rep in 1 2; do
  for arm in product ablation; do
    if [ $arm = ablation ]; then E="env FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so ${STEP_AB_ENV}"; else E="env"; fi
    $E python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-configs --no-bench-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print('$arm rep $rep: ms_per_step %.3f' % d['ms_per_step'])
for n in sorted(k):
    if 'warp3d' in n: print('   %-34s %s' % (n, {a:k[n][a] for a in k[n] if a in ('ms_per_step','launches_per_step','ms_per_launch','GBps')}))
"
  done
done
