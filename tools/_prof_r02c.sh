# round-2 evidence, final pass after the attention-backward change: full GPU test suite, bench line, kernel stats of the bench, the student
# step, the TFAM train step (B = 512) and the B = 8 census (the GEMM / encoder files of tools/_prof_r02.sh are not affected by it)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
timeout -k 10 400 python3 bench.py > $O/bench_line.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/ks.log 2>&1
find $O/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rm -rf $O/ks
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 tools/student_bench.py > $O/st.log 2>&1
find $O/st -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/student_train_kernel_stats.csv
rm -rf $O/st
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t5 -- python3 tools/tfam_bench.py 512 train > $O/t5.log 2>&1
find $O/t5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_train_B512_kernel_stats.csv
rm -rf $O/t5
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tc -- python3 tools/tfam_train_census.py 8 20 > $O/tc.log 2>&1
find $O/tc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_train_B8_census_kernel_stats.csv
rm -rf $O/tc
tail -3 $O/gputest.log
tail -c 300 $O/bench_line.json
