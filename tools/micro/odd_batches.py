#!/usr/bin/env python3
"""Whole-tile self-attention on batches that do not make whole rounds of 256-row items (B*H*ceil(L/256) against 256
compute units): the default kernel (fa_forward) next to the 128-row kernel (reached through fa_forward_varlen with
every k_len = L and FA_VARLEN_W64=0), with what the launcher's models of the two kernels predict.
    python tools/micro/odd_batches.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["FA_VARLEN_W64"] = "0"
import torch
import __graft_entry__ as ge
fa = ge.load_package()
dev = torch.device("cuda:0")
D = 128


def timeit(f, reps=20, preheat=0.5):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < preheat:
        for _ in range(10):
            f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B, H, L in ((1, 40, 2048), (2, 40, 2048), (1, 40, 4096), (1, 32, 2048), (3, 32, 1024), (5, 16, 1024), (17, 16, 512), (20, 16, 512),
                (33, 16, 512), (1, 24, 8192), (1, 20, 4096), (9, 8, 2048), (1, 72, 1024)):
    q, k, v, o = [torch.empty((B, H, L, D), dtype=torch.float16, device=dev) for _ in range(4)]
    for t, seed in zip((q, k, v), (1234, 2345, 3456)):
        fa.init_half(t.view(-1), seed)
    kl = torch.full((B,), L, dtype=torch.int32, device=dev)
    a = timeit(lambda: fa.attention(q, k, v, out=o))
    b = timeit(lambda: fa.attention_varlen(q, k, v, k_len=kl, out=o))
    items = B * H * ((L + 255) // 256)
    rounds = -(-items // 256)
    m_w64 = 8.0 + rounds * (1.31 * (L // 64) + 3.3)
    m_hd = 22.4 + B * H * ((L + 127) // 128) / 512.0 * (0.8535 * (L // 32) + 0.094)
    ways = fa.lib().fa_split_ways(B * H, L, L)
    print(f"B={B:2d} H={H:2d} L={L:5d} items {items:5d} = {items / 256:5.2f} rounds (key split x{ways})  default {a:8.1f} us   128-row {b:8.1f} us"
          f"   default/128-row {a / b:.3f}   models {m_w64:7.1f} / {m_hd:7.1f}", flush=True)
