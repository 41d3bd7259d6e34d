#!/usr/bin/env python3
"""Back-to-back launches of the L = 512 README shape: plain stream launches against one HIP graph of the same
launches (does a graph shorten the gap between consecutive kernels?).

    python tools/graph_gap.py [--seq-len 512] [--launches 50]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--seq-len", type=int, default=512)
ap.add_argument("--launches", type=int, default=50)
args = ap.parse_args()
fa = ge.load_package()
dev = torch.device("cuda:0")
L, H, D = args.seq_len, 16, 128
B = max(1, 16384 // L)
q, k, v = [torch.empty((B, H, L, D), dtype=torch.float16, device=dev) for _ in range(3)]
for t, seed in zip((q, k, v), (1234, 2345, 3456)):
    fa.init_half(t, seed)
o = torch.empty_like(q)
n = args.launches


def plain():
    for _ in range(n):
        fa.attention(q, k, v, out=o)


side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    plain()
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(n):
            fa.attention(q, k, v, out=o, stream=side)


def timed(fn, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:  # the clock needs sustained load to settle
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


flops = 4.0 * B * H * L * L * D
for name, fn in (("stream launches", plain), ("one graph", graph.replay), ("stream launches", plain), ("one graph", graph.replay)):
    us = timed(fn, 10)
    print(f"{name:16s} {us:8.2f} us per launch   {flops / us / 1e6:8.1f} TFLOP/s", flush=True)
