# round-3: PMC passes on the ViT attention kernel (variant 0 = round-2 kernel, 1 = 8 waves + K re-read)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
for v in 0 1; do
  export VMC_ATTN_VARIANT=$v
  i=0
  for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/apmc_${v}_$i -- python3 tools/attn_bench.py > $O/apmc_${v}_$i.log 2>&1
  done
  python3 tools/pmc_summary.py attn_vit_kernel $O/apmc_${v}_1 $O/apmc_${v}_2 > $O/attn_pmc_v$v.txt
  cat $O/attn_pmc_v$v.txt
  rm -rf $O/apmc_${v}_1 $O/apmc_${v}_2
done
