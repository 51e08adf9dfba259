# matrix-pipe utilisation, LDS activity and HBM bytes of the split-bf16 transposed kernels (product) next to the fp32 class
# kernels of the same layers (ablation build, FLOWSCI_TR_NO_S3=1); one counter group per run of scripts/trbench.py
export TMPDIR=/tmp
O=gpurun_out/pmc_t3
rm -rf $O && mkdir -p $O
AB=$PWD/opticalflowscivis_amd/csrc/ablation
make -C opticalflowscivis_amd/csrc ablation -j16 > gpurun_out/make_ablation.log 2>&1 || { tail -20 gpurun_out/make_ablation.log; exit 1; }
for lib in product fp32; do
  if [ $lib = fp32 ]; then export FLOWSCI_HIP_LIBRARY=$AB/libflowsci_hip_ab.so FLOWSCI_TR_NO_S3=1; else unset FLOWSCI_HIP_LIBRARY FLOWSCI_TR_NO_S3; fi
  for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE" "WRITE_SIZE"; do
    if timeout -k 5 200 rocprofv3 --pmc $grp --output-format csv -d $O/g -- python3 scripts/trbench.py > $O/$lib.log 2>&1; then
      python scripts/pmc_summary.py "$(find $O/g -name '*counter_collection.csv' | head -1)" convtr_ 60 | grep -E "convtr_s3|convtr_mfma" >> $O/$lib.txt
    else
      echo "# group '$grp' not collected" >> $O/$lib.txt; tail -3 $O/$lib.log >> $O/$lib.txt
    fi
    rm -rf $O/g
  done
done
echo "######## product"; cat $O/product.txt; echo "######## fp32"; cat $O/fp32.txt
