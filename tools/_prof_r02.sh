# round-2 evidence: bench line, rocprofv3 kernel stats of the same command, PMC passes (FETCH_SIZE / WRITE_SIZE / L2) on the c_fc GEMM
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/bench_line.json 2> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/ks.log 2>&1
find $O/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rm -rf $O/ks
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 tools/gemm_bench.py --shapes 65792,4096,1024 --act 1 --iters 3 --rounds 1 > $O/pmc_$n.log 2>&1
done
python3 tools/pmc_summary.py gemm8_kernel $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum > $O/gemm8_pmc.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tf -- python3 tools/tfam_chain_run.py 8 100 > $O/tf.log 2>&1
find $O/tf -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_chain_B8_kernel_stats.csv
rm -rf $O/tf
tail -c 400 $O/bench_line.json
cat $O/gemm8_pmc.txt
