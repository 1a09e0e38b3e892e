# round 3, VERDICT r2 item 6 (i): the persistent GEMM on 32x32x16 fragments against the 16x16x32 form -- counters on the c_fc shape
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
for v in 0 1; do
  export VMC_GEMM_MFMA32=$v
  i=0
  for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "WRITE_SIZE" "FETCH_SIZE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/m${v}_$i -- python3 tools/gemm_bench.py --shapes 65792,4096,1024 --act 1 --iters 3 --rounds 1 > $O/m${v}_$i.log 2>&1
  done
  python3 tools/pmc_summary.py gemm8p $O/m${v}_1 $O/m${v}_2 $O/m${v}_3 $O/m${v}_4 $O/m${v}_5 > $O/gemm_mfma32_${v}_pmc.txt
  rm -rf $O/m${v}_1 $O/m${v}_2 $O/m${v}_3 $O/m${v}_4 $O/m${v}_5
done
for v in 0 1; do echo "MFMA32=$v"; cat $O/gemm_mfma32_${v}_pmc.txt; done
