#!/bin/bash
# PMC view of the HBM-bound hot-path kernels (scripts/pmc_hbm.py): one counter group per run, as gpurun requires.
export TMPDIR=/tmp
O=gpurun_out/pmc_hbm
rm -rf $O && mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  if rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 scripts/pmc_hbm.py > $O/g$i.log 2>&1; then
    python scripts/pmc_summary.py "$(find $O/g$i -name '*counter_collection.csv' | head -1)" all 100 >> $O/summary.txt
  else
    echo "# group '$grp' not collected (see g$i.log)" >> $O/summary.txt
  fi
  rm -rf $O/g$i
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 scripts/pmc_hbm.py > $O/kt.log 2>&1
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv; rm -rf $O/kt
grep -v "wprep\|elementwise\|Fill\|fill\|distribution\|normal\|reduce_kernel" $O/summary.txt
