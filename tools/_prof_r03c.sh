set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tf64 -- python3 tools/tfam_chain_run.py 64 50 > $O/tf64.log 2>&1
find $O/tf64 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_chain_B64_kernel_stats.csv
rm -rf $O/tf64
python3 tools/kstats.py $O/tfam_chain_B64_kernel_stats.csv 30
