#!/usr/bin/env python3
"""Bare bf16 MFMA loops on pseudo-random register operands, one 256-thread workgroup per CU (vmc_clock_probe): 16x16x32 against
32x32x16 -- sustained clock and TFLOP/s of each on THIS box (the power question of VERDICT r2 item 6(i))."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd._lib import check, lib, ptr, stream  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
stamps = torch.zeros(512, dtype=torch.int64, device="cuda")
for rnd in range(3):
    for shape, name in ((0, "16x16x32"), (1, "32x32x16")):
        for _ in range(2):
            check(lib.vmc_clock_probe(ptr(stamps), 256, iters, shape, stream()), "probe")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.vmc_clock_probe(ptr(stamps), 256, iters, shape, stream()), "probe")
        e1.record()
        torch.cuda.synchronize()
        st = stamps.view(256, 2).cpu().double()
        mhz = (st[:, 0] / st[:, 1] * 100.0).median().item()
        flops = 256 * 4 * iters * 64 * 2.0 * 16 * 16 * 32          # CUs x waves x rounds x MFMAs x FLOPs (both shapes: same per round)
        ms = e0.elapsed_time(e1)
        cyc_per_round = (st[:, 0].median().item()) / iters
        print(f"round {rnd} {name}: {mhz:7.1f} MHz in-kernel, {cyc_per_round:7.1f} cycles per round of 64-equivalent MFMAs, "
              f"{flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
