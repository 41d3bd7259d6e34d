#!/usr/bin/env python3
"""Soak test of fa_forward_general and fa_forward_block_general against the CPU oracle.

    python tests/soak/fuzz_general.py [--iters 200] [--seed 0]

Random B, H, Lq, Lk (1..700), D in {64, 128, 256}, with and without the causal mask; every third case
goes through the blockwise entry point with the keys cut into random blocks folded in random order.
Gate as in the tests (max-abs <= 2e-3, relative L2 <= 1e-3), with the absolute bound scaled by |O| where |O| > 1:
with N(0,1) values and the mask, early rows have outputs of 2 to 5, where neighbouring fp16 numbers are 2e-3 to 4e-3
apart and one rounding step of the output alone would break an absolute 2e-3.

    python tests/soak/fuzz_general.py --tiles

D = 128 and lengths in whole 64-row tiles, the shapes that fa_forward_general and fa_forward_block_general hand to
the tuned kernels: causal sequences in one call, and causal sequences cut into chunks (some of them ragged, which
stay on the general kernel) with every query chunk folded over its key chunks in random order."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
import oracle
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--tiles", action="store_true")
args = ap.parse_args()
fa = ge.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(args.seed)
bad = 0


def within_gate(ref, out16):
    out = out16.astype(np.float32)
    m = oracle.error_metrics(ref, out16)
    ok = (np.isfinite(out).all() and bool((np.abs(out - ref) <= 2e-3 * np.maximum(1.0, np.abs(ref))).all())
          and m["global_rel"] <= 1e-3)
    return ok, m


for it in range(args.iters if args.tiles else 0):
    B, H = int(rng.integers(1, 3)), int(rng.integers(1, 12))
    nchunk = int(rng.integers(1, 5))
    sizes = [64 * int(rng.integers(1, 13)) if rng.integers(0, 2) else int(rng.integers(1, 1200)) for _ in range(nchunk)]
    cuts = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    L = cuts[-1]
    scale = float(rng.uniform(0.3, 1.6))
    Q = (rng.standard_normal((B, H, L, 128), dtype=np.float32) * scale).astype(np.float16)
    K = (rng.standard_normal((B, H, L, 128), dtype=np.float32) * scale).astype(np.float16)
    V = rng.standard_normal((B, H, L, 128), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    if it % 2 == 0:
        out = fa.attention_general(q, k, v, causal=True)
        kernels = {fa.lib().fa_kernel_name_general(L, L, 128, 1)}
    else:
        out = torch.empty_like(q)
        kernels = set()
        for qc in range(nchunk):
            qa, qe = cuts[qc], cuts[qc + 1]
            qq = q[:, :, qa:qe].contiguous()
            st = fa.BlockState(B * H, qe - qa, 128, dev)
            order = rng.permutation(qc + 1)
            for n, kc in enumerate(order):
                ka, ke = cuts[kc], cuts[kc + 1]
                kernels.add(fa.lib().fa_kernel_name_block_general(qe - qa, ke - ka, 128, 1, qa, ka))
                o = fa.attention_block_general(qq, k[:, :, ka:ke].contiguous(), v[:, :, ka:ke].contiguous(), st,
                                               last=(n == len(order) - 1), causal=True, q_pos0=qa, k_pos0=ka)
            out[:, :, qa:qe] = o
    torch.cuda.synchronize()
    ok, m = within_gate(oracle.attention_cpu(Q, K, V, causal=True), out.cpu().numpy())
    if not ok:
        bad += 1
        ref_ = oracle.attention_cpu(Q, K, V, causal=True)
        out_ = out.float().cpu().numpy()
        at = np.unravel_index(np.argmax(np.abs(out_ - ref_)), ref_.shape)
        print(f"iter {it}: B={B} H={H} chunks={sizes} one_call={it % 2 == 0} kernels={sorted(kernels)}: {m}\n"
              f"    worst element {tuple(int(i) for i in at)}: oracle {ref_[at]:.6f}, kernel {out_[at]:.6f}")
    if it % 50 == 49:
        print(f"{it + 1} cases, {bad} failures", flush=True)
if args.tiles:
    print("fuzz_general --tiles:", "FAILED" if bad else "ok", f"({args.iters} cases, {bad} failures)")
    sys.exit(1 if bad else 0)
for it in range(args.iters):
    D = int(rng.choice([64, 128, 256]))
    B, H = int(rng.integers(1, 3)), int(rng.integers(1, 5))
    causal = bool(rng.integers(0, 2))
    Lq = int(rng.integers(1, 700))
    Lk = Lq if causal else int(rng.integers(1, 700))
    scale = float(rng.uniform(0.3, 1.6))
    Q = (rng.standard_normal((B, H, Lq, D), dtype=np.float32) * scale).astype(np.float16)
    K = (rng.standard_normal((B, H, Lk, D), dtype=np.float32) * scale).astype(np.float16)
    V = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    blockwise = it % 3 == 2
    if blockwise:
        ncut = int(rng.integers(0, min(4, Lk)))
        cuts = [0] + sorted(rng.choice(np.arange(1, Lk), size=ncut, replace=False).tolist()) + [Lk] if Lk > 1 else [0, Lk]
        order = rng.permutation(len(cuts) - 1)
        st = fa.BlockState(B * H, Lq, D, dev)
        out = None
        for n, c in enumerate(order):
            a, b = cuts[c], cuts[c + 1]
            out = fa.attention_block_general(q, k[:, :, a:b].contiguous(), v[:, :, a:b].contiguous(), st,
                                             last=(n == len(order) - 1), causal=causal, q_pos0=0, k_pos0=a)
    else:
        out = fa.attention_general(q, k, v, causal=causal)
    torch.cuda.synchronize()
    if causal:
        ref = oracle.attention_cpu(Q, K, V, causal=True)
    else:
        st = oracle.BlockState(B, H, Lq, D)
        oracle.block_update(st, Q, K, V)
        ref = oracle.block_finalize(st)
    ok, m = within_gate(ref, out.cpu().numpy())
    if not ok:
        bad += 1
        print(f"iter {it}: B={B} H={H} Lq={Lq} Lk={Lk} D={D} causal={causal} blockwise={blockwise}: {m}")
    if it % 50 == 49:
        print(f"{it + 1} cases, {bad} failures", flush=True)
print("fuzz_general:", "FAILED" if bad else "ok", f"({args.iters} cases, {bad} failures)")
sys.exit(1 if bad else 0)
