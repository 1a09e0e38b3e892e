# round-3 evidence: bench line, rocprofv3 kernel stats of the same command, PMC passes on the dominant GEMM, the secondary legs
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f
mkdir -p $O
timeout -k 10 600 python3 bench.py > $O/bench_line.json 2> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/ks.log 2>&1
find $O/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rm -rf $O/ks
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 tools/gemm_bench.py --shapes 65792,4096,1024 --act 1 --iters 3 --rounds 1 > $O/pmc_$n.log 2>&1
done
python3 tools/pmc_summary.py gemm8p_kernel $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $O/gemm8_pmc.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/es -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/es.log 2>&1
find $O/es -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/encoder_step_kernel_stats.csv
rm -rf $O/es
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tf -- python3 tools/tfam_chain_run.py 8 100 > $O/tf.log 2>&1
find $O/tf -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_chain_B8_kernel_stats.csv
rm -rf $O/tf
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 tools/student_bench.py > $O/st.log 2>&1
find $O/st -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/student_train_kernel_stats.csv
rm -rf $O/st
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tc -- python3 tools/tfam_train_census.py 8 50 > $O/tc.log 2>&1
find $O/tc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_train_B8_fused_kernel_stats.csv
rm -rf $O/tc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t5 -- python3 tools/tfam_bench.py 512 train > $O/t5.log 2>&1
find $O/t5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_train_B512_kernel_stats.csv
rm -rf $O/t5
tail -c 600 $O/bench_line.json
