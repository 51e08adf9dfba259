# the bench step on the product library vs a previous build of it (csrc/ablation/libflowsci_hip_prev.so), alternating, same box
for i in 1 2 3; do
for lib in new prev; do
  if [ $lib = prev ]; then export FLOWSCI_HIP_LIBRARY=$PWD/opticalflowscivis_amd/csrc/ablation/libflowsci_hip_prev.so; else unset FLOWSCI_HIP_LIBRARY; fi
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-configs --no-bench-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k['name']:k for k in d.get('kernels',[])} if isinstance(d.get('kernels'),list) else {}
print('$lib', 'ms_per_step %.3f' % d['ms_per_step'], ' '.join('%s %.2f' % (n, ks[n].get('ms_per_step',0)) for n in ('fs_conv3d_fwd','fs_conv3d_wrw','fs_conv3d_tr') if n in ks))" || exit 1
done
done
