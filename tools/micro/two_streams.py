#!/usr/bin/env python3
"""Independent attention calls issued on two (or more) streams: does the next kernel's start overlap the previous
kernel's tail?  (Within one stream a kernel starts only when its predecessor has finished completely: last stores
drained, caches written back.)  Each stream has its own output; inputs are shared.

    python tools/micro/two_streams.py [--streams 1 2 3] [--shapes 512 1024 2048]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3])
ap.add_argument("--shapes", type=int, nargs="+", default=[512, 1024, 2048])
ap.add_argument("--steps", type=int, default=240)
ap.add_argument("--preheat", type=float, default=1.0)
args = ap.parse_args()
fa = ge.load_package()
dev = torch.device("cuda:0")
H, D = 16, 128
for L in args.shapes:
    B = 16384 // L
    q, k, v = [torch.empty((B, H, L, D), dtype=torch.float16, device=dev) for _ in range(3)]
    for t, seed in zip((q, k, v), (1234, 2345, 3456)):
        fa.init_half(t, seed)
    flops = 4.0 * B * H * L * L * D
    for rep in range(2):
        for n in args.streams:
            streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
            outs = [torch.empty_like(q) for _ in range(n)]
            torch.cuda.synchronize()

            def burst(count):
                for i in range(count):
                    fa.attention(q, k, v, out=outs[i % n], stream=streams[i % n])

            t0 = time.perf_counter()
            while time.perf_counter() - t0 < args.preheat:
                burst(24)
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            burst(args.steps)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            print(f"B={B:2d} L={L:5d} streams={n}: {dt * 1e6:8.2f} us per call  {flops / dt / 1e12:8.1f} TFLOP/s", flush=True)
