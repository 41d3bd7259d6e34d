#!/usr/bin/env python3
"""The compute side of the sequence-parallel ring, measured on ONE GPU.

    python tests/soak/ring_one_gpu.py [--world 8] [--L 131072] [--rank 3]

Plays rank `--rank` of a `--world`-way ring over a sequence of `--L` rows (B=1, H=16, D=128; the
long-sequence configuration of the multi-GPU plan): its L/world query rows are folded against the
world K/V blocks in ring order with fa_forward_block, exactly what multi_gpu.ring_forward does
between its sends and receives.  Prints the time per step and per rank, and checks a few rows of the
result against the CPU oracle over the full sequence.  (The exchange itself needs a multi-GPU node:
each step moves 2 * 16 * (L/world) * 128 * 2 bytes to the next rank.)"""
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
args = ap.parse_args()
fa = ge.load_package()
import oracle
from importlib import import_module
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
dev = torch.device("cuda:0")
H, D, P, L = 16, 128, args.world, args.L
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
