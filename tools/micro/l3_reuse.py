#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache serve part of the L = 512 benchmark's reads?

The benchmark launches the same Q, K, V, O (4 x 64 MiB = the cache's size) again and again.  This times the same
launches with the tensors rotated through N independent sets (N x 256 MiB touched between two uses of a line: no
reuse possible) and at half the batch (128 MiB per set), same buffers against rotated ones.

    python tools/micro/l3_reuse.py [--steps 200] [--data pattern|normal]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--data", default="pattern")
ap.add_argument("--preheat", type=float, default=1.0)
ap.add_argument("--libs", nargs="*", default=[], help="experiment builds to time instead of the product (B = 32 only)")
args = ap.parse_args()
fa = ge.load_package()
dev = torch.device("cuda:0")
H, L, D = 16, 512, 128


def make(B, nsets):
    sets = []
    for s in range(nsets):
        q, k, v = [torch.empty((B, H, L, D), dtype=torch.float16, device=dev) for _ in range(3)]
        if args.data == "pattern":
            for t, seed in zip((q, k, v), (1234, 2345, 3456)):
                fa.init_half(t, seed + s)
        else:
            for t in (q, k, v):
                t.copy_(torch.randn(t.shape, device=dev, dtype=torch.float32).to(torch.float16))
        sets.append((q, k, v, torch.empty_like(q)))
    return sets


def run(B, nsets, only_out_rotates=False):
    sets = make(B, nsets)
    calls = []
    for i in range(nsets):
        q, k, v, o = sets[0] if only_out_rotates else sets[i]
        o = sets[i][3]
        calls.append((q, k, v, o))
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < args.preheat:
        for _ in range(10):
            q, k, v, o = calls[i % nsets]
            fa.attention(q, k, v, out=o)
            i += 1
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(args.steps):
        q, k, v, o = calls[j % nsets]
        fa.attention(q, k, v, out=o)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    tf = 4.0 * B * H * L * L * D / ms / 1e9
    what = "inputs fixed, outputs rotate" if only_out_rotates else "all four tensors rotate"
    print(f"B={B:3d} sets={nsets:2d} ({what if nsets > 1 else 'same buffers every launch'}; "
          f"{nsets * 4 * B * H * L * D * 2 / 2**20:.0f} MiB touched per cycle) {ms * 1e3:8.2f} us  {tf:8.1f} TFLOP/s",
          flush=True)
    del sets, calls
    torch.cuda.empty_cache()


if args.libs:  # same buffers against rotated ones, per build (timing builds may give wrong results: only times count)
    for rep in range(2):
        for path in ["product"] + args.libs:
            if path != "product":
                fa._lib = fa.bind(os.path.join(ROOT, path))
            print(path, end=": ")
            run(32, 1)
            print(path, end=": ")
            run(32, 4)
            fa._lib = None
    sys.exit(0)
for rep in range(2):
    run(32, 1)
    run(32, 4)
    run(32, 4, only_out_rotates=True)
    run(16, 1)
    run(16, 8)
    run(8, 1)
    run(8, 16)
