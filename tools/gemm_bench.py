#!/usr/bin/env python3
"""Micro-benchmark of vmc_linear on the shapes of the ViT-L/14 encoder (and square reference shapes).
    python tools/gemm_bench.py [--shapes M,N,K ...] [--iters 20] [--dtype bf16]
Random (never zero) operands; HIP events on the launch stream; prints TFLOP/s per shape."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import ops  # noqa: E402

DEFAULT = ["65792,1024,1024", "65792,3072,1024", "65792,4096,1024", "65792,1024,4096", "8192,8192,8192", "4096,4096,4096"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=DEFAULT)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--variants", type=lambda x: int(x, 0), nargs="*", default=[1],
                    help="vmc_linear_variant values: 0 two-stage, 1 default, 2 no tail split")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--act", type=int, default=0, help="0 none, 1 QuickGELU (adds a bias too)")
    ap.add_argument("--res32", action="store_true", help="fp32 residual in + fp32 out in place (out_proj / c_proj epilogue)")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands (NOT a throughput figure: shows how much of the gap to "
                    "peak is data-dependent power / clock, cdna_hip_programming.md rule 25)")
    ap.add_argument("--fill", default="randn", choices=["randn", "zeros", "ones", "sparse90", "smallint"],
                    help="operand values: randn (default, the throughput figure), zeros, ones (constant 1.0), sparse90 (randn with 90 %% of "
                         "the elements zeroed), smallint (integers -2..2) -- everything but randn only shows the data dependence of the clock")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    for s in args.shapes:
        M, N, K = map(int, s.split(","))
        a = torch.randn(M, K, device="cuda").to(dt)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        fill = "zeros" if args.zeros else args.fill
        if fill == "zeros":
            a.zero_()
            w.zero_()
        elif fill == "ones":
            a.fill_(1.0)
            w.fill_(1.0)
        elif fill == "sparse90":
            a *= (torch.rand(M, K, device="cuda") < 0.1).to(dt)
            w *= (torch.rand(N, K, device="cuda") < 0.1).to(dt)
        elif fill == "smallint":
            a = torch.randint(-2, 3, (M, K), device="cuda").to(dt)
            w = torch.randint(-2, 3, (N, K), device="cuda").to(dt)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if args.res32 else dt)
        bias = torch.randn(N, device="cuda")          # every encoder linear has one (and the persistent kernel takes bias linears)
        kw = dict(bias=bias, out=out, act=args.act, res=out if args.res32 else None)
        res = {}
        for rnd in range(args.rounds):            # interleaved rounds in ONE process (variants share clocks/thermals)
            for var in args.variants:
                kw["variant"] = var
                for _ in range(2):
                    ops.linear(a, w, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    ops.linear(a, w, **kw)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(var, []).append(e0.elapsed_time(e1) / args.iters)
        for var, ts in res.items():
            ts = sorted(ts)
            ms = ts[len(ts) // 2]
            print(f"M={M} N={N} K={K} {args.dtype} variant {var:#x}: median {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s"
                  f"  (min {ts[0]*1e3:.1f} us)", flush=True)


if __name__ == "__main__":
    main()
