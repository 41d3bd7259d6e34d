#!/usr/bin/env python3
"""Shapes with fewer workgroups than the chip holds on the 128-row kernels: the launcher's key split (FA_SPLIT unset)
next to the undivided launch (FA_SPLIT=1).

    python tools/split_mid.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
dev = torch.device("cuda:0")
shapes = [(64, 8, 4096, 4096), (64, 4, 8192, 8192), (64, 16, 2048, 2048), (64, 2, 16384, 16384), (64, 32, 256, 8192),
          (256, 2, 8192, 8192), (256, 1, 16384, 16384), (256, 8, 128, 8192), (256, 4, 2048, 4096),
          (128, 8, 32, 8192), (128, 2, 63, 32768), (128, 32, 8, 4096), (128, 1, 127, 65536)]
for D, BH, Lq, Lk in shapes:
    q = torch.empty((1, BH, Lq, D), dtype=torch.float16, device=dev)
    k = torch.empty((1, BH, Lk, D), dtype=torch.float16, device=dev)
    v = torch.empty_like(k)
    for t, s in zip((q, k, v), (1, 2, 3)):
        fa.init_half(t, s)
    o = torch.empty_like(q)
    res = []
    for split in (None, "1"):
        os.environ.pop("FA_SPLIT", None)
        if split:
            os.environ["FA_SPLIT"] = split
        fa.reload_env()
        f = lambda: fa.attention_general(q, k, v, out=o)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            f()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / 10, fa.lib().fa_split_ways_general(BH, Lq, Lk, D, 0)))
    os.environ.pop("FA_SPLIT", None)
    fa.reload_env()
    fl = 4.0 * BH * Lq * Lk * D
    (a, sa), (b, _) = res
    print(f"D={D:3d} BH={BH:3d} Lq={Lq:6d} Lk={Lk:6d} {fa.lib().fa_kernel_name_general(Lq, Lk, D, 0).decode():32s} x{sa:<2d} {a * 1e3:8.1f} us {fl / a / 1e9:7.1f} TF"
          f"   undivided {b * 1e3:8.1f} us {fl / b / 1e9:7.1f} TF   ratio {b / a:.2f}", flush=True)
