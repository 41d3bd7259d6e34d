#!/usr/bin/env python3
"""Padded batches (fa_forward_varlen) at D = 128: the two-kernel route for long sequences next to the 128-row
kernel alone and next to one fa_forward_general call per sequence.

    python tools/bench_varlen.py [seed]

Pattern data, 1 s of preheat per line, back-to-back launches between two events.  FLOPs: 4*H*D*sum(q_len*k_len), with the causal mask
the visible triangle 2*H*D*sum(n*(n+1)).
Lines: lengths drawn uniformly from a range (padded to its upper end), then a batch with one long sequence among
short ones, then all sequences at the padded length (what the two-kernel route costs when nothing is ragged)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
fa = ge.load_package()
dev = torch.device("cuda:0")
H, D = 16, 128
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)


def timed(f, reps=20, preheat=1.0):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < preheat:
        for _ in range(5):
            f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def case(label, lens, pad, causal=False):
    B = len(lens)
    q, k, v, o = [torch.empty((B, H, pad, D), dtype=torch.float16, device=dev) for _ in range(4)]
    for t, seed in zip((q, k, v), (1234, 2345, 3456)):
        fa.init_half(t.view(-1), seed)
    ql = torch.tensor(lens, dtype=torch.int32, device=dev)
    flops = (2.0 * H * D * sum(float(n) * (n + 1) for n in lens) if causal else
             4.0 * H * D * sum(float(n) * n for n in lens))
    out = []
    for name, env in (("two kernels", None), ("128-row kernel", "0")):
        os.environ.pop("FA_VARLEN_W64", None)
        if env is not None:
            os.environ["FA_VARLEN_W64"] = env
        fa.reload_env()
        ms = timed(lambda: fa.attention_varlen(q, k, v, ql, ql, causal=causal, out=o))
        out.append(f"{name} {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TFLOP/s")
    os.environ.pop("FA_VARLEN_W64", None)
    fa.reload_env()
    seqs = [[t[b:b + 1, :, :n].contiguous() for t in (q, k, v, o)] for b, n in enumerate(lens) if n > 0]

    def per_sequence():
        for qs, ks, vs, os_ in seqs:
            fa.attention_general(qs, ks, vs, causal=causal, out=os_)
    ms = timed(per_sequence)
    out.append(f"one call per sequence {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TFLOP/s")
    print(f"{label + (', causal' if causal else ''):34s} B={B:2d} pad={pad:5d} | " + " | ".join(out), flush=True)


case("uniform 1100..4096", [int(x) for x in rng.integers(1100, 4097, 8)], 4096)
case("uniform 1100..2048", [int(x) for x in rng.integers(1100, 2049, 16)], 2048)
case("uniform 64..4096", [int(x) for x in rng.integers(64, 4097, 8)], 4096)
case("uniform 64..2048", [int(x) for x in rng.integers(64, 2049, 16)], 2048)
case("one of 4096 among 15 of 200", [4096] + [200] * 15, 4096)
case("all 2048 (nothing ragged)", [2048] * 8, 2048)
case("all 2047", [2047] * 8, 2048)
case("all 1024 (below the threshold)", [1024] * 16, 1024)
case("uniform 1665..4096", [int(x) for x in rng.integers(1665, 4097, 8)], 4096, causal=True)
case("uniform 2305..8192", [int(x) for x in rng.integers(2305, 8193, 4)], 8192, causal=True)
case("uniform 64..4096", [int(x) for x in rng.integers(64, 4097, 8)], 4096, causal=True)
case("all 4096", [4096] * 4, 4096, causal=True)
case("all 4095", [4095] * 4, 4096, causal=True)
