set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/st
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 tools/student_bench.py > $O/ks.log 2>&1
find $O/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/student_kernel_stats.csv
rm -rf $O/ks
tail -n 3 $O/ks.log
