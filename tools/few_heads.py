#!/usr/bin/env python3
"""Few heads, long sequence: B=1, H heads, L keys on one GPU, with the launcher's own key split and without.

    python tools/few_heads.py [--L 16384] [--heads 1 2 3 4 8 16] [--splits auto 0 2 4]

Prints per (H, FA_SPLIT) the time per call (events over back-to-back calls after a preheat) and TFLOP/s; the
H = 16 row is the full long-sequence shape, H = 2 what one of eight ranks holds of it."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=16384)
ap.add_argument("--heads", type=int, nargs="+", default=[1, 2, 3, 4, 8, 16])
ap.add_argument("--splits", nargs="+", default=["auto", "0"])
ap.add_argument("--data", choices=["pattern", "normal"], default="pattern")
ap.add_argument("--preheat", type=float, default=0.7)
args = ap.parse_args()
fa = ge.load_package()
dev = torch.device("cuda:0")
L, D = args.L, 128
for H in args.heads:
    q, k, v = [torch.empty((1, H, L, D), dtype=torch.float16, device=dev) for _ in range(3)]
    if args.data == "pattern":
        for t, seed in zip((q, k, v), (1234, 2345, 3456)):
            fa.init_half(t, seed)
    else:
        g = torch.Generator(device=dev).manual_seed(0)
        for t in (q, k, v):
            t.copy_(torch.randn(t.shape, generator=g, device=dev).to(torch.float16))
    o = torch.empty_like(q)
    for sp in args.splits:
        os.environ.pop("FA_SPLIT", None)
        if sp != "auto":
            os.environ["FA_SPLIT"] = sp
        fa.reload_env()
        ways = fa.lib().fa_split_ways(H, L, L)
        f = lambda: fa.attention(q, k, v, out=o)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.preheat:
            for _ in range(10):
                f()
            torch.cuda.synchronize()
        reps = max(5, int(40 * 16 / H * (16384 / L) ** 2 / 8))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tf = 4.0 * H * L * L * D / ms / 1e9
        print(f"H={H:2d} L={L} FA_SPLIT={sp:>4s} (ways {ways}): {ms * 1e3:9.1f} us  {tf:8.1f} TFLOP/s", flush=True)
os.environ.pop("FA_SPLIT", None)
