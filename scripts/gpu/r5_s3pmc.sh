# matrix-pipe utilisation and effective clock of the split-bf16 forward kernel vs the fp32 kernels (one counter group per run)
export TMPDIR=/tmp
O=gpurun_out/pmc_s3
rm -rf $O && mkdir -p $O
AB=$PWD/opticalflowscivis_amd/csrc/ablation
make -C opticalflowscivis_amd/csrc ablation -j16 > gpurun_out/make_ablation.log 2>&1 || { tail -20 gpurun_out/make_ablation.log; exit 1; }
for lib in product fp32; do
  if [ $lib = fp32 ]; then export FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so FLOWSCI_FWD_NO_S3=1; else unset FLOWSCI_HIP_LIBRARY FLOWSCI_FWD_NO_S3; fi
  for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE" "WRITE_SIZE"; do
    if timeout -k 5 200 rocprofv3 --pmc $grp --output-format csv -d $O/g -- python3 scripts/s3bench.py > $O/$lib.log 2>&1; then
      python scripts/pmc_summary.py "$(find $O/g -name '*counter_collection.csv' | head -1)" conv3d_fwd 70 >> $O/$lib.txt
    else
      echo "# group '$grp' not collected" >> $O/$lib.txt; tail -3 $O/$lib.log >> $O/$lib.txt
    fi
    rm -rf $O/g
  done
done
cat $O/product.txt; echo =====; cat $O/fp32.txt
