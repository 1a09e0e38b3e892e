set -e
python tools/gemm_bench.py --shapes 65792,4096,1024 65792,3072,1024 65792,1024,1024 --variants 0x21 0x61 0xa1 0xe1 --act 0 --rounds 3 --iters 15
python tools/gemm_bench.py --shapes 65792,1024,4096 65792,1024,1024 --variants 1 0x41 0x81 --res32 --rounds 3 --iters 15
