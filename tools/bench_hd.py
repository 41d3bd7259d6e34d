#!/usr/bin/env python3
"""Head dims 64 and 256 through fa_forward_general, several builds of the library side by side.

    python tools/bench_hd.py [--libs build_ab/a.so ...] [--dims 64 256] [--shapes 2048 8192]

Per (D, L) with B*L = 16384 rows, H = 16 (pattern data): time per call after a preheat and TFLOP/s (4*B*H*L^2*D)."""
import argparse, ctypes as ct, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ap = argparse.ArgumentParser()
ap.add_argument("--libs", nargs="*", default=[])
ap.add_argument("--dims", type=int, nargs="+", default=[64, 256])
ap.add_argument("--shapes", type=int, nargs="+", default=[2048, 8192])
ap.add_argument("--causal", action="store_true")
ap.add_argument("--check", action="store_true")
ap.add_argument("--lminus", type=int, default=0, help="shorten every L by this much (ragged lengths)")
args = ap.parse_args()
fa = ge.load_package()
libs = [("product", fa.lib())] + [(os.path.basename(p), fa.bind(os.path.join(ROOT, p))) for p in args.libs]
dev = torch.device("cuda:0")
H = 16
for D in args.dims:
    for L in args.shapes:
        B = max(1, 16384 // L)
        L -= args.lminus
        q, k, v = [torch.empty((B, H, L, D), dtype=torch.float16, device=dev) for _ in range(3)]
        for t, seed in zip((q, k, v), (1234, 2345, 3456)):
            fa.init_half(t, seed)
        outs = []
        for name, lib in libs:
            fa._lib = lib
            o = torch.empty_like(q)
            f = lambda: fa.attention_general(q, k, v, causal=args.causal, out=o)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.5:
                for _ in range(10):
                    f()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            flops = (2.0 * B * H * L * (L + 1) * D) if args.causal else 4.0 * B * H * L * L * D
            kern = lib.fa_kernel_name_general(L, L, D, int(args.causal)).decode()
            extra = ""
            outs.append(o)
            if args.check and len(outs) > 1:
                extra = f"  max|diff vs first| {(o.float() - outs[0].float()).abs().max().item():.2e}"
            print(f"D={D:3d} B={B:2d} L={L:5d} {name:16s} {kern:34s} {ms * 1e3:9.1f} us {flops / ms / 1e9:8.1f} TFLOP/s{extra}", flush=True)
fa._lib = libs[0][1]
