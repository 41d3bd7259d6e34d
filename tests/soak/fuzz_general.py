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
stay on the general kernel) with every query chunk folded over its key chunks in random order.

    python tests/soak/fuzz_general.py --varlen

Padded batches through fa_forward_varlen: random per-sequence query and key lengths (empty sequences included), NaN in
the padding and a sentinel in the output, three kinds of shapes: self-attention up to 700 rows (with and without the
causal mask), different query and key lengths (the causal mask counted from each sequence's start), and decoding
batches of 1 to 16 query rows against up to 8000 keys, which the launcher divides into key ranges.  Per sequence
the oracle runs on the truncated tensors; rows behind q_len must keep the sentinel, a sequence without keys gives 0."""
import argparse
import os, sys
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
ap.add_argument("--varlen", action="store_true")
args = ap.parse_args()
os.environ.setdefault("FA_VARLEN_W64", "1")  # (small batches: the tuned kernel's list form wherever it can run)
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


for it in range(args.iters if args.varlen else 0):
    D = int(rng.choice([64, 128, 256]))
    B, H = int(rng.integers(1, 7)), int(rng.integers(1, 4))
    kind = ("self", "cross", "decode", "long")[it % 4]
    causal = bool(rng.integers(0, 2)) and kind not in ("decode", "long")
    if kind == "long":  # D = 128, lengths on both sides of 1100: the tuned kernel's per-item lengths beside the 128-row kernel
        D, H = 128, int(rng.choice([1, 2, 3, 8, 9, 10]))
        B = int(rng.integers(1, max(2, 14 // H + 1)))
        Lq, Lk = int(rng.integers(1100, 2200)), int(rng.integers(1100, 2600))
        if it % 8 == 3:  # with the mask: as many rows as keys, thresholds 1665 / 2305
            causal, Lq = True, int(rng.integers(1665, 2700))
            Lk = Lq
    elif kind == "self":
        Lq = Lk = int(rng.integers(1, 700))
    elif kind == "cross":
        Lq, Lk = int(rng.integers(1, 500)), int(rng.integers(1, 700))
    else:
        Lq, Lk = int(rng.integers(1, 17)), int(rng.integers(1000, 8000))
    low = 900 if kind == "long" else 0
    q_lens = [int(rng.integers(low, Lq + 1)) if rng.integers(0, 4) else Lq for _ in range(B)]
    k_lens = [int(rng.integers(low, Lk + 1)) if rng.integers(0, 4) else Lk for _ in range(B)]
    if kind == "self" or (kind == "long" and causal):
        k_lens = list(q_lens)
        if kind == "long" and B > 1:  # one sequence whose rows and keys differ: never the tuned kernel's
            k_lens[0] = max(1, q_lens[0] - int(rng.integers(1, 100)))
    scale = float(rng.uniform(0.3, 1.6))
    Q = (rng.standard_normal((B, H, Lq, D), dtype=np.float32) * scale).astype(np.float16)
    K = (rng.standard_normal((B, H, Lk, D), dtype=np.float32) * scale).astype(np.float16)
    V = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    Qp, Kp, Vp = Q.copy(), K.copy(), V.copy()
    for b in range(B):
        Qp[b, :, q_lens[b]:] = np.nan
        Kp[b, :, k_lens[b]:] = np.nan
        Vp[b, :, k_lens[b]:] = np.nan
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Qp, Kp, Vp)]
    o = torch.full((B, H, Lq, D), 123.0, dtype=torch.float16, device=dev)
    fa.attention_varlen(q, k, v, torch.tensor(q_lens, dtype=torch.int32, device=dev),
                        torch.tensor(k_lens, dtype=torch.int32, device=dev), causal=causal, out=o)
    torch.cuda.synchronize()
    O = o.cpu().numpy()
    why = []
    for b in range(B):
        nq, nk = q_lens[b], k_lens[b]
        if not (O[b, :, nq:] == 123.0).all():
            why.append(f"sequence {b}: wrote behind its last query row")
        if nq == 0:
            continue
        if nk == 0:
            if not (O[b, :, :nq] == 0).all():
                why.append(f"sequence {b}: no keys must give 0")
            continue
        Qs, Ks, Vs = [np.ascontiguousarray(a[b:b + 1, :, :n]) for a, n in ((Q, nq), (K, nk), (V, nk))]
        st = oracle.BlockState(1, H, nq, D)
        oracle.block_update(st, Qs, Ks, Vs, causal=causal, q_pos0=0, k_pos0=0)
        ok, m = within_gate(oracle.block_finalize(st), O[b:b + 1, :, :nq])
        if not ok:
            why.append(f"sequence {b} (q_len {nq}, k_len {nk}): {m}")
    if why:
        bad += 1
        ways = fa.lib().fa_split_ways_general(B * H, Lq, Lk, D, int(causal))
        print(f"iter {it}: {kind} B={B} H={H} Lq={Lq} Lk={Lk} D={D} causal={causal} split={ways}: " + "; ".join(why))
    if it % 50 == 49:
        print(f"{it + 1} cases, {bad} failures", flush=True)
if args.varlen:
    print("fuzz_general --varlen:", "FAILED" if bad else "ok", f"({args.iters} cases, {bad} failures)")
    sys.exit(1 if bad else 0)
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
