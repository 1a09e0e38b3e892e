#!/usr/bin/env python3
"""Bit-equality screen of the persistent 8-phase GEMM (VMC_GEMM_PERSISTENT) against the default kernel, several shapes and
repeats (a schedule race shows as rare wrong tiles), then an interleaved timing of both.
    python tools/gemm_persist_check.py [--repeats 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import ops  # noqa: E402

CASES = [  # M, N, K, act, res32 (the persistent kernel takes the 16-bit-output ones; the fp32-residual ones stay one-tile)
    (65792, 4096, 1024, 1, False), (65792, 3072, 1024, 0, False), (65792, 1024, 1024, 0, False), (65792, 1024, 4096, 0, False),
    (16384, 4096, 256, 0, False), (8192, 8192, 512, 3, False), (25600, 3072, 768, 0, False), (65536, 256, 384, 1, False),
    (65792, 1024, 1024, 0, True),
]


def run(a, w, bias, out, act, res32, variant):
    return ops.linear(a, w, bias=bias, out=out, act=act, res=out if res32 else None, variant=variant)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    bad = 0
    for dt in (torch.bfloat16, torch.float16):
        for (M, N, K, act, res32) in CASES:
            g = torch.Generator(device="cuda").manual_seed(M + N + K)
            a = torch.randn(M, K, device="cuda", generator=g).to(dt)
            w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
            bias = torch.randn(N, device="cuda", generator=g)
            x0 = torch.randn(M, N, device="cuda", generator=g) if res32 else None
            mk = lambda: x0.clone() if res32 else torch.zeros(M, N, device="cuda", dtype=dt)
            ref = run(a, w, bias, mk(), act, res32, 4)
            for rep in range(args.repeats):
                got = run(a, w, bias, mk(), act, res32, 1)
                if not torch.equal(got, ref):
                    d = (got.float() - ref.float()).abs()
                    rows = (d.amax(1) > 0).nonzero().flatten()
                    print(f"MISMATCH {dt} {M}x{N}x{K} act {act} res32 {res32} rep {rep}: max {d.max().item():.3e}, {rows.numel()} rows, first {rows[:4].tolist()}", flush=True)
                    bad += 1
            print(f"ok  {str(dt):15s} M={M} N={N} K={K} act={act} res32={res32}", flush=True)
    print("mismatches:", bad, flush=True)
    if bad:
        sys.exit(1)
    for (M, N, K, act, res32) in CASES[:4] + CASES[-1:]:
        dt = torch.bfloat16
        a = torch.randn(M, K, device="cuda").to(dt)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        bias = torch.randn(N, device="cuda")
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if res32 else dt)
        res = {}
        for rnd in range(3):
            for var in (4, 1):
                for _ in range(2):
                    run(a, w, bias, out, act, res32, var)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    run(a, w, bias, out, act, res32, var)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(var, []).append(e0.elapsed_time(e1) / args.iters)
        for var, ts in res.items():
            ms = sorted(ts)[1]
            print(f"M={M} N={N} K={K} act={act} res32={res32} variant {var}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
