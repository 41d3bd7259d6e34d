#!/usr/bin/env python3
"""The compute side of the sequence-parallel ring, measured on ONE GPU.

    python tests/soak/ring_one_gpu.py [--world 8] [--L 131072] [--rank 3]

Plays rank `--rank` of a `--world`-way ring over a sequence of `--L` rows (B=1, H=16, D=128; the
long-sequence configuration of the multi-GPU plan): its L/world query rows are folded against the
world K/V blocks in ring order with fa_forward_block, exactly what multi_gpu.ring_forward does
between its sends and receives.  Prints the time per step and per rank, and checks a few rows of the
result against the CPU oracle over the full sequence.  (The exchange itself needs a multi-GPU node:
each step moves 2 * 16 * (L/world) * 128 * 2 bytes to the next rank.)

    python tests/soak/ring_one_gpu.py --causal [--general-only]

The same for the zig-zag causal ring (multi_gpu.run_zigzag_steps): the rank owns chunks r and 2P-1-r of 2P
chunks and folds two or three (query chunk, key chunk) pairs per step with fa_forward_block_general, which
sends pairs below the diagonal to the unmasked tuned kernel and diagonal pairs to its causal form
(--general-only keeps everything on the general kernel, FA_BLOCK_TUNED=0, for comparison)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--L", type=int, default=131072)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--check-rows", type=int, default=6)
ap.add_argument("--causal", action="store_true")
ap.add_argument("--general-only", action="store_true")
args = ap.parse_args()
fa = ge.load_package()
import oracle
from importlib import import_module
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
dev = torch.device("cuda:0")
H, D, P, L = 16, 128, args.world, args.L


def causal_mode():
    mg = import_module("multi_gpu")
    if args.general_only:
        os.environ["FA_BLOCK_TUNED"] = "0"
        fa.reload_env()
    Lc = L // (2 * P)
    g = torch.Generator(device=dev).manual_seed(1)
    K, V, Qall = [torch.randn((1, H, L, D), generator=g, device=dev, dtype=torch.float32).to(torch.float16) for _ in range(3)]
    chunk = lambda T, c: T[:, :, c * Lc:(c + 1) * Lc].contiguous()
    mine = mg.zigzag_chunks(args.rank, P)
    q2 = torch.stack([chunk(Qall, c) for c in mine])
    kv = []
    for s in range(P):
        theirs = mg.zigzag_chunks((args.rank - s) % P, P)
        kv.append(torch.stack((torch.stack([chunk(K, c) for c in theirs]), torch.stack([chunk(V, c) for c in theirs]))))
    kernels = set()

    def step(q, k, v, state, last, causal, q_pos0, k_pos0):
        kernels.add(fa.lib().fa_kernel_name_block_general(q.shape[2], k.shape[2], D, int(causal), q_pos0, k_pos0).decode())
        return fa.attention_block_general(q, k, v, state, last, causal=causal, q_pos0=q_pos0, k_pos0=k_pos0)

    run = lambda: mg.run_zigzag_steps(q2, lambda s: kv[s], args.rank, P, Lc, step, fa.BlockState)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        out = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    pairs = sum(len(st) for st in mg.zigzag_schedule(args.rank, P))
    visible = sum(2.0 * H * Lc * (Lc + 1) * D if d else 4.0 * H * Lc * Lc * D for st in mg.zigzag_schedule(args.rank, P) for (_, _, _, _, d) in st)
    print(f"causal zig-zag ring world={P} L={L} rank={args.rank}: {pairs} pairs of {Lc} x {Lc}, rank compute {ms:.2f} ms = "
          f"{ms/P:.3f} ms per step, {visible/ms/1e9:.0f} TFLOP/s of visible work; kernels: {sorted(kernels)}")
    rows = np.unique(np.linspace(0, Lc - 1, args.check_rows).astype(np.int32))
    Qn, Kn, Vn = Qall.cpu().numpy(), K.cpu().numpy(), V.cpu().numpy()
    for slot, c in enumerate(mine):
        pos = (c * Lc + rows).astype(np.int32)
        ref = oracle.attention_cpu_rows(Qn, Kn, Vn, pos, causal=True)
        m = oracle.error_metrics(ref, out[slot].cpu().numpy()[:, :, rows])
        print(f"  chunk {c}: parity on {len(rows)} rows x {H} heads: max_abs {m['max_abs']:.2e} rel_l2 {m['global_rel']:.2e}")
        assert m["max_abs"] <= 2e-3 * max(1.0, float(np.abs(ref).max())) and m["global_rel"] <= 1e-3
    sys.exit(0)


if args.causal:
    causal_mode()
Lb = L // P
g = torch.Generator(device=dev).manual_seed(1)
K = torch.randn((1, H, L, D), generator=g, device=dev, dtype=torch.float32).to(torch.float16)
V = torch.randn((1, H, L, D), generator=g, device=dev, dtype=torch.float32).to(torch.float16)
Q = torch.randn((1, H, Lb, D), generator=g, device=dev, dtype=torch.float32).to(torch.float16)  # this rank's rows
order = [(args.rank - s) % P for s in range(P)]
blocks = [(K[:, :, b * Lb:(b + 1) * Lb].contiguous(), V[:, :, b * Lb:(b + 1) * Lb].contiguous()) for b in range(P)]

def run():
    st = fa.BlockState(H, Lb, D, dev)
    out = None
    for s, b in enumerate(order):
        out = fa.attention_block(Q, blocks[b][0], blocks[b][1], st, last=(s == P - 1))
    return out

for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    run()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    out = run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flops = 4.0 * H * Lb * L * D
msg_mb = 2 * H * Lb * D * 2 / 1e6
print(f"ring world={P} L={L}: rank compute {ms:.2f} ms = {ms/P:.3f} ms per step, {flops/ms/1e9:.0f} TFLOP/s per GPU; "
      f"each step sends {msg_mb:.0f} MB to the next rank ({msg_mb/1e3/(ms/P*1e-3):.0f} GB/s hides it fully)")
# parity on a few rows: the oracle sees the rank's Q rows against the whole K, V
rows = np.unique(np.linspace(0, Lb - 1, args.check_rows).astype(np.int32))
Qn, Kn, Vn = Q.cpu().numpy(), K.cpu().numpy(), V.cpu().numpy()
st = oracle.BlockState(1, H, len(rows), D)
oracle.block_update(st, np.ascontiguousarray(Qn[:, :, rows]), Kn, Vn)
ref = oracle.block_finalize(st)
m = oracle.error_metrics(ref, out.cpu().numpy()[:, :, rows])
print(f"parity on {len(rows)} rows x {H} heads: max_abs {m['max_abs']:.2e} rel_l2 {m['global_rel']:.2e}")
assert m["max_abs"] <= 2e-3 and m["global_rel"] <= 1e-3
