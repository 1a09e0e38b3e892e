# encoder-only kernel breakdown (headline step): rocprofv3 kernel stats of bench.py without the secondary legs
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/enc
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/ks.log 2>&1
find $O/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/enc_kernel_stats.csv
rm -rf $O/ks
tail -n 1 $O/ks.log | cut -c1-200
