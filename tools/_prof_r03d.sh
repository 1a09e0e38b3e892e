# round-3, VERDICT r2 item 6: (i) MFMA shape vs sustained clock; (ii) tile-walk group width vs HBM traffic of the c_fc GEMM
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
python3 tools/mfma_shape_probe.py 20000 > $O/mfma_shape_probe.log 2>&1
for gc in 4 2 8 16; do
  export VMC_GEMM_GC=$gc
  python3 tools/gemm_bench.py --shapes 65792,4096,1024 65792,3072,1024 --act 1 --iters 10 --rounds 3 > $O/gemm_gc$gc.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/g_$n -- python3 tools/gemm_bench.py --shapes 65792,4096,1024 --act 1 --iters 3 --rounds 1 > $O/g_$n.log 2>&1
  done
  python3 tools/pmc_summary.py gemm8p_kernel $O/g_FETCH_SIZE $O/g_WRITE_SIZE $O/g_TCC_HIT_sum > $O/gemm_gc${gc}_pmc.txt
  rm -rf $O/g_FETCH_SIZE $O/g_WRITE_SIZE $O/g_TCC_HIT_sum
done
cat $O/mfma_shape_probe.log
for gc in 4 2 8 16; do echo "GC=$gc"; grep TFLOP $O/gemm_gc$gc.log; cat $O/gemm_gc${gc}_pmc.txt; done
