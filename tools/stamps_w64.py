#!/usr/bin/env python3
"""Diagnostic: where the cycles of a step go around an item boundary in the one-wave-per-SIMD kernel.

    python tools/stamps_w64.py build_ab/NAME.so [L] [pattern|normal] [mask]

NAME is a tools/build_w64_only.sh build with -DFA_STAMPS -DW64_FINE_STAMPS=<first step>: every wave keeps the shader
clock at seven points of each of 18 consecutive steps (from its <first step>-th on) in register lanes and dumps
them at the end (mask = the build's -DW64_FINE_STAMPS_MASK, the points it takes; default all seven).  Printed: per step and point, the median over workgroups of the cycles since the point before,
for wave 0 and for the slowest wave of each workgroup, and the spread between the waves at the step's end.

  0 step start        1 phase 1 starts     2 phase 1 done     3 phase 2 starts (cold block of B, rescale done)
  4 phase 2 done      5 cold block of A, rescale done         6 end-of-step wait done  (then the barrier)
"""
import ctypes as ct
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

fa = ge.load_package()
lib = ct.CDLL(os.path.join(ROOT, sys.argv[1]))
L = int(sys.argv[2]) if len(sys.argv) > 2 else 512
data = sys.argv[3] if len(sys.argv) > 3 else "pattern"
sig = [fa.HalfPtr] * 4 + [ct.c_int] * 5 + [ct.c_void_p, ct.POINTER(ct.c_float)]
lib.fa_forward.argtypes = sig
lib.fa_forward.restype = ct.c_int
lib.fa_debug_set_stamps.argtypes = [ct.c_void_p]
dev = torch.device("cuda:0")
H, D = 16, 128
n = 16384 * H * D
q = torch.empty(n, dtype=torch.float16, device=dev)
k, v, o = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
if data == "pattern":
    fa.init_half(q, 1234), fa.init_half(k, 2345), fa.init_half(v, 3456)
else:
    g = torch.Generator(device=dev).manual_seed(0)
    for t in (q, k, v):
        t.copy_(torch.randn(n, generator=g, device=dev, dtype=torch.float32).to(torch.float16))
WG = 256
buf = torch.zeros(4 * WG + WG * 4 * 64, dtype=torch.int64, device=dev)  # 4 u64 per workgroup, then 128 u32 per wave
p = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), fa.HalfPtr)
B = 16384 // L
import time

t0 = time.time()
while time.time() - t0 < 1.5:
    for _ in range(50):
        assert lib.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None) == 0
    torch.cuda.synchronize()
lib.fa_debug_set_stamps(ct.c_void_p(buf.data_ptr()))
for _ in range(3):
    lib.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
whole = raw[:4 * WG].reshape(WG, 4)
print(f"{sys.argv[1]} L={L} {data}: cycles per workgroup median {np.median(whole[:, 2] - whole[:, 0]):.0f}")
mask = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0x7F
ks = [k_ for k_ in range(7) if mask >> k_ & 1]
st = raw[4 * WG:].view(np.uint32).reshape(WG, 4, 128)[:, :, :126].reshape(WG, 4, 18, 7).astype(np.int64)
st = st[:, :, :, ks]
ok = (st[:, :, :, 0] != 0).all(axis=(1, 2))
st = st[ok]
print(f"workgroups with stamps: {ok.sum()}")
n = st.shape[0]
flat = st.reshape(n, 4, 18 * len(ks)).copy()
for i in range(1, flat.shape[2]):  # a point not taken in a step (builds that stamp inside a cold block) counts 0 cycles
    miss = flat[:, :, i] == 0
    flat[:, :, i][miss] = flat[:, :, i - 1][miss]
st = flat.reshape(n, 4, 18, len(ks))
d = np.diff(flat, axis=2) & 0xFFFFFFFF  # cycles since the stamp before, per wave
d = np.concatenate([np.zeros_like(d[:, :, :1]), d], axis=2).reshape(n, 4, 18, len(ks))
names = ["barrier", "pre-p1", "phase1", "coldB", "phase2", "coldA", "wait"]
print("(each column: cycles from the stamp before to this one; wave 0, median over workgroups)")
print("step " + " ".join(f"->{names[k_]:>7s}" for k_ in ks) + "    total | slowest wave | waves' spread at the step's last stamp")
for s_ in range(18):
    w0 = np.median(d[:, 0, s_, :], axis=0)
    per_wave_total = d[:, :, s_, :].sum(axis=2)
    end = st[:, :, s_, -1]
    spread = (end.max(axis=1) - end.min(axis=1)) & 0xFFFFFFFF
    print(f"{s_:4d} " + " ".join(f"{x:9.0f}" for x in w0) + f" {w0.sum():8.0f} | {np.median(per_wave_total.max(axis=1)):8.0f} | {np.median(spread):6.0f}")
print("per wave, total per step: ")
for wv in range(4):
    print(f"  wave {wv}: " + " ".join(f"{np.median(d[:, wv, s_, :].sum(axis=1)):.0f}" for s_ in range(18)))
