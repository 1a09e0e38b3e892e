set -e
python tools/gemm_bench.py --shapes 65792,4096,1024 65792,3072,1024 --variants 1 0x11 0x21 0x31 --act 1 --rounds 3 --iters 15
python tools/gemm_bench.py --shapes 65792,4096,1024 --variants 1 0x11 --act 0 --rounds 3 --iters 15
