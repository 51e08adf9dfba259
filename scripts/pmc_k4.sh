#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/pmc_k4
rm -rf $O && mkdir -p $O
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python scripts/pmc_k4.py > $O/g$i.log 2>&1
  python scripts/pmc_summary.py "$(find $O/g$i -name '*counter_collection.csv' | head -1)" >> $O/summary.txt
  rm -rf $O/g$i
done
grep -v "wprep\|elementwise\|Fill\|fill\|distribution\|normal" $O/summary.txt
