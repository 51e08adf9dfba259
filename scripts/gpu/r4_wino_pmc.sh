# PMC counters of the Winograd-domain trunk kernels (the persistent 2-D forward kernel, the F(4,3) weight gradient), one small
# group per rocprofv3 run, each under its own timeout -> gpurun_out/pmc_wino/summary.txt
export TMPDIR=/tmp
O=gpurun_out/pmc_wino
rm -rf $O && mkdir -p $O
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  echo "group $i: $grp"
  timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 scripts/pmc_wino.py > $O/g$i.log 2>&1 || { tail -3 $O/g$i.log; exit 1; }
  python3 scripts/pmc_summary.py "$(find $O/g$i -name '*counter_collection.csv' | head -1)" >> $O/summary.txt
  rm -rf $O/g$i
done
grep -v "wprep\|elementwise\|Fill\|fill" $O/summary.txt
