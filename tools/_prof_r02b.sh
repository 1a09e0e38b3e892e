# round-2 evidence, part 2: HBM traffic counters of the two secondary kernels of the encoder step (add+LayerNorm, ViT attention)
# and kernel stats of the TFAM training step at B = 512 and the TFAM forward at B = 4096
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pa_$c -- python3 tools/attn_bench.py > $O/pa_$c.log 2>&1
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pl_$c -- python3 tools/addln_bench.py > $O/pl_$c.log 2>&1
done
python3 tools/pmc_summary.py attn_vit_kernel $O/pa_FETCH_SIZE $O/pa_WRITE_SIZE > $O/secondary_pmc.txt
python3 tools/pmc_summary.py add_ln_kernel $O/pl_FETCH_SIZE $O/pl_WRITE_SIZE >> $O/secondary_pmc.txt
rm -rf $O/pa_FETCH_SIZE $O/pa_WRITE_SIZE $O/pl_FETCH_SIZE $O/pl_WRITE_SIZE
timeout -k 10 100 python3 tools/attn_bench.py > $O/attn_bench.log 2>&1
timeout -k 10 100 python3 tools/addln_bench.py > $O/addln_bench.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t5 -- python3 tools/tfam_bench.py 512 train > $O/t5.log 2>&1
find $O/t5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_train_B512_kernel_stats.csv
rm -rf $O/t5
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t4 -- python3 tools/tfam_bench.py 4096 > $O/t4.log 2>&1
find $O/t4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tfam_fwd_B4096_kernel_stats.csv
rm -rf $O/t4
cat $O/secondary_pmc.txt $O/attn_bench.log $O/addln_bench.log
