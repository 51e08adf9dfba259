# 2-D Winograd trunk kernel: which loader job costs the overlap?  FLOWSCI_WINO_DBG on the ablation build:
# 0 full, 1 no U slab DMA, 2 no input loads / transforms, 3 neither (matrix waves + epilogue), 4 loaders alone.
# WINO_BENCH_CIN=32 halves the periods per brick: T64 = 8 (32 p + f), T32 = 8 (16 p + f) separates the per-brick fixed cost f.
AB=$PWD/opticalflowscivis_amd/csrc/ablation/libflowsci_hip_ab.so
for d in ${WINO_DBGS:-0 1 2 3 4}; do
  for c in 64 32; do
  echo "== FLOWSCI_WINO_DBG=$d Cin=$c $WINO_ENV"
  env $WINO_ENV WINO_BENCH_CIN=$c FLOWSCI_WINO_DBG=$d FLOWSCI_HIP_LIBRARY=$AB timeout -k 10 200 python tests/tools/wino_bench.py 2>&1 | grep "ms/launch" | cut -c1-110 || exit 1
  done
done
