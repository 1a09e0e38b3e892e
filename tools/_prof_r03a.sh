# round-3: kernel census of the fused TFAM train step at B = 8 (eager launches; the durations are device time per kernel)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tc -- python3 tools/tfam_train_census.py 8 50 > $O/tc.log 2>&1
find $O/tc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_train_B8_census_kernel_stats.csv
find $O/tc -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/tfam_train_B8_trace.csv
rm -rf $O/tc
python3 tools/kstats.py $O/tfam_train_B8_census_kernel_stats.csv 40
