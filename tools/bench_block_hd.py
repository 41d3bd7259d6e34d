#!/usr/bin/env python3
"""Blockwise entry point at head dims 64 and 256: one rank's share of a sequence-parallel ring on one GPU (a query
chunk folded over `--blocks` key chunks of the same length), fa_fwd_hd_state_kernel<D> next to the general
kernel (FA_HD=0).

    python tools/bench_block_hd.py [--rows 2048] [--blocks 4] [--heads 64] [--causal]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2048)
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--heads", type=int, default=64)
ap.add_argument("--causal", action="store_true", help="the chunk is the sequence's last: its own block is the diagonal one")
args = ap.parse_args()
fa = ge.load_package()
dev = torch.device("cuda:0")
R, NB, BH = args.rows, args.blocks, args.heads
for D in (64, 256):
    q = torch.empty((1, BH, R, D), dtype=torch.float16, device=dev)
    ks = [torch.empty_like(q) for _ in range(NB)]
    vs = [torch.empty_like(q) for _ in range(NB)]
    for i, t in enumerate([q] + ks + vs):
        fa.init_half(t, 1000 + i)
    res = {}
    for hd in (None, "0"):
        os.environ.pop("FA_HD", None)
        if hd is not None:
            os.environ["FA_HD"] = hd
        fa.reload_env()
        st = fa.BlockState(BH, R, D, dev)
        def ring():
            st.first = True
            out = None
            for n in range(NB):
                out = fa.attention_block_general(q, ks[n], vs[n], st, last=(n == NB - 1), causal=args.causal,
                                                 q_pos0=(NB - 1) * R, k_pos0=n * R)
            return out
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            ring()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = ring()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        flops = 4.0 * BH * R * R * D * (NB - 0.5 if args.causal else NB)
        name = fa.lib().fa_kernel_name_block_general(R, R, D, int(args.causal), (NB - 1) * R, 0).decode()
        res[hd] = out.float()
        print(f"D={D:3d} rows {R} x {NB} blocks, {BH} heads, causal={args.causal}: {name:32s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
    print(f"        max|diff| between the two: {float((res[None] - res['0']).abs().max()):.2e}")
os.environ.pop("FA_HD", None)
