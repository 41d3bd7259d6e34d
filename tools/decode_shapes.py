#!/usr/bin/env python3
"""Few query rows against many keys (the shape of a decoding step) through fa_forward_general: time next to the
HBM floor of reading K and V once (8 TB/s).

    python tools/decode_shapes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
dev = torch.device("cuda:0")
for D in (64, 128, 256):
    for BH, Lq, Lk in ((512, 1, 8192), (512, 16, 8192), (64, 1, 16384), (16, 1, 65536), (512, 64, 2048)):
        q = torch.empty((1, BH, Lq, D), dtype=torch.float16, device=dev)
        k = torch.empty((1, BH, Lk, D), dtype=torch.float16, device=dev)
        v = torch.empty_like(k)
        for t, s in zip((q, k, v), (1, 2, 3)):
            fa.init_half(t, s)
        o = torch.empty_like(q)
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
        ms = e0.elapsed_time(e1) / 10
        bytes_ = 2.0 * BH * Lk * D * 2
        ways = fa.lib().fa_split_ways_general(BH, Lq, Lk, D, 0)
        print(f"D={D:3d} BH={BH:4d} Lq={Lq:3d} Lk={Lk:6d}  {fa.lib().fa_kernel_name_general(Lq, Lk, D, 0).decode():34s} x{ways:<2d} {ms * 1e3:9.1f} us"
              f"   K+V {bytes_ / 1e6:7.1f} MB  {bytes_ / ms / 1e9:6.2f} TB/s  (floor {bytes_ / 8e12 * 1e6:7.1f} us)", flush=True)
