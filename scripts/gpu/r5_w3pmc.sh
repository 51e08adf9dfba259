# PMC view of the trilinear warp pair kernels, one launch kind per run (scripts/w3bench.py 256 <flow> fwd|bwd|acc3):
# product library (round-5 row-cache ring kernels) vs the round-4 kernels (ablation build, FLOWSCI_W3_RC=0).
#   W3_KIND=smooth|small  W3_PARTS="fwd bwd acc3"  W3_LIBS="product r4"
export TMPDIR=/tmp
set -x
O=gpurun_out/pmc_w3
rm -rf $O && mkdir -p $O
KIND=${W3_KIND:-smooth}
AB=$PWD/opticalflowscivis_amd/csrc/ablation
make -C opticalflowscivis_amd/csrc ablation -j16 > gpurun_out/make_ablation.log 2>&1 || { tail -20 gpurun_out/make_ablation.log; exit 1; }
for lib in ${W3_LIBS:-product r4}; do
  if [ $lib = r4 ]; then export FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so FLOWSCI_W3_RC=0; else unset FLOWSCI_HIP_LIBRARY FLOWSCI_W3_RC; fi
  for part in ${W3_PARTS:-fwd bwd acc3}; do
    echo "## $lib / $part ($KIND flow)" >> $O/$lib.txt
    i=0
    for grp in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum" "FETCH_SIZE" "WRITE_SIZE" \
               "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
               "TD_TD_BUSY_sum TD_TC_STALL_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
               "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
               "TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1))
      if timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $O/g -- python3 scripts/w3bench.py 256 $KIND $part > $O/$lib.$part.g$i.log 2>&1; then
        python scripts/pmc_summary.py "$(find $O/g -name '*counter_collection.csv' | head -1)" warp3d 64 >> $O/$lib.txt
      else
        echo "# group '$grp' not collected" >> $O/$lib.txt; tail -3 $O/$lib.$part.g$i.log >> $O/$lib.txt
      fi
      rm -rf $O/g
    done
  done
done
cat $O/product.txt; echo ====; cat $O/r4.txt 2>/dev/null
