#!/usr/bin/env python3
"""Measuring stick only (not used by the product): torch.matmul (hipBLASLt) on the encoder GEMM shapes."""
import torch
for M, N, K in [(65792, 1024, 1024), (65792, 3072, 1024), (65792, 4096, 1024), (65792, 1024, 4096), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    for _ in range(3):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.matmul(a, w.t())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"hipBLASLt M={M} N={N} K={K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
