# PMC view of the plain warp pair kernels: product library vs the round-3 kernels (one counter group per run)
export TMPDIR=/tmp
O=gpurun_out/pmc_w3
rm -rf $O && mkdir -p $O
AB=$PWD/opticalflowscivis_amd/csrc/ablation
for lib in product r3; do
  if [ $lib = r3 ]; then export FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_w3r3.so; else unset FLOWSCI_HIP_LIBRARY; fi
  i=0
  for grp in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
             "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
             "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
    i=$((i+1))
    echo "[$lib] group $i: $grp"
    if timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $O/g -- python3 scripts/w3bench.py 256 smooth plain > $O/$lib.g$i.log 2>&1; then
      python scripts/pmc_summary.py "$(find $O/g -name '*counter_collection.csv' | head -1)" warp3d 60 >> $O/$lib.txt
    else
      echo "# group '$grp' not collected" >> $O/$lib.txt; tail -3 $O/$lib.g$i.log >> $O/$lib.txt
    fi
    rm -rf $O/g
  done
done
cat $O/product.txt; echo ====; cat $O/r3.txt
