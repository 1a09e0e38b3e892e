#!/usr/bin/env python3
"""In-process A/B of encoder options on the headline workload (ViT-L/14, 256 u8 frames, bf16): interleaved rounds so both arms share the
box, the clocks and the thermals.    python tools/enc_ab.py defer_attn_add [--rounds 4 --steps 5]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth  # noqa: E402
from vimo_clip_amd.clip_vit import VisionTransformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("attr")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--frames", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    m = VisionTransformer.from_name("ViT-L/14", compute_dtype=torch.bfloat16).to(dev).eval()
    m.load_state_dict(synth.vit_state_dict("ViT-L/14", seed=2), strict=True)
    m.frame_chunk = args.frames
    frames = synth.randint_u8(1, "frames", (args.frames, 3, 224, 224)).to(dev)
    res = {True: [], False: []}
    for rnd in range(args.rounds):
        for val in (True, False):
            setattr(m, args.attr, val)
            for _ in range(2):
                m.encode_frames_u8(frames)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                m.encode_frames_u8(frames)
            torch.cuda.synchronize()
            res[val].append((time.perf_counter() - t0) / args.steps)
    for val, ts in res.items():
        ts = sorted(ts)
        print(f"{args.attr}={val}: median {1e3 * ts[len(ts) // 2]:.3f} ms/step ({args.frames / ts[len(ts) // 2]:.0f} frames/s), min {1e3 * ts[0]:.3f}", flush=True)


if __name__ == "__main__":
    main()
