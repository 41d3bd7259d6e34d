#!/usr/bin/env python3
"""Soak test of fa_forward_general and fa_forward_block_general against the CPU oracle.

    python tests/soak/fuzz_general.py [--iters 200] [--seed 0]

Random B, H, Lq, Lk (1..700), D in {64, 128, 256}, with and without the causal mask; every third case
goes through the blockwise entry point with the keys cut into random blocks folded in random order.
Gate as in the tests: max-abs <= 2e-3, relative L2 <= 1e-3."""
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
args = ap.parse_args()
fa = ge.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(args.seed)
bad = 0
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
    m = oracle.error_metrics(ref, out.cpu().numpy())
    ok = np.isfinite(out.float().cpu().numpy()).all() and m["max_abs"] <= 2e-3 and m["global_rel"] <= 1e-3
    if not ok:
        bad += 1
        print(f"iter {it}: B={B} H={H} Lq={Lq} Lk={Lk} D={D} causal={causal} blockwise={blockwise}: {m}")
    if it % 50 == 49:
        print(f"{it + 1} cases, {bad} failures", flush=True)
print("fuzz_general:", "FAILED" if bad else "ok", f"({args.iters} cases, {bad} failures)")
sys.exit(1 if bad else 0)
