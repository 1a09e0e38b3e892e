#!/usr/bin/env python3
"""Weight-gradient GEMM: TN kernel on token-major operands vs two transposes + NT split-K kernel.
    python tools/wgrad_bench.py [--shapes M,N,K ...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import ops  # noqa: E402
from vimo_clip_amd._lib import check, lib, ptr, stream  # noqa: E402

DEFAULT = ["25600,768,768", "25600,2304,768", "25600,3072,768", "25600,768,3072", "65792,1024,4096", "65536,512,512", "4096,512,2048"]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=DEFAULT)
    args = ap.parse_args()
    for s in args.shapes:
        M, N, K = map(int, s.split(","))
        dy = torch.randn(M, N, device="cuda").bfloat16()
        x = torch.randn(M, K, device="cuda").bfloat16()
        out = torch.empty(N, K, device="cuda")
        dyt = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
        xt = torch.empty(K, M, device="cuda", dtype=torch.bfloat16)

        def nt():
            check(lib.vmc_transpose16(ptr(dy), ptr(dyt), M, N, N, M, stream()), "t")
            check(lib.vmc_transpose16(ptr(x), ptr(xt), M, K, K, M, stream()), "t")
            ops.linear_wgrad(dyt, xt, out)

        t_tn = timeit(lambda: ops.wgrad_tn(dy, x, out))
        a = out.clone()
        t_nt = timeit(nt) if M % 64 == 0 else float("nan")
        err = (a - out).abs().max().item() if M % 64 == 0 else float("nan")
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K}: TN {t_tn*1e3:7.1f} us {fl/t_tn/1e9:7.1f} TF | transposes+NT {t_nt*1e3:7.1f} us {fl/t_nt/1e9:7.1f} TF | max diff {err:.3g}", flush=True)


if __name__ == "__main__":
    main()
