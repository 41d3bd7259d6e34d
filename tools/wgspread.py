#!/usr/bin/env python3
"""Diagnostic: how the workgroups of one launch spread in time (library built with -DFA_STAMPS).

    python tools/wgspread.py build_ab/NAME.so L

Per launch of a back-to-back train: start spread, per-workgroup duration and cycle count, and the same
per XCD (workgroup id mod 8).  On the boxes measured the odd XCDs clock about 2 % lower than the even
ones under this load, so with equal work per workgroup they finish 1.5-2 us later at L=512."""
import ctypes as ct, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
lib = ct.CDLL(os.path.join(ROOT, sys.argv[1]))
L = int(sys.argv[2])
sig = [fa.HalfPtr] * 4 + [ct.c_int] * 5 + [ct.c_void_p, ct.POINTER(ct.c_float)]
lib.fa_forward.argtypes = sig; lib.fa_forward.restype = ct.c_int
lib.fa_debug_set_stamps.argtypes = [ct.c_void_p]
dev = torch.device("cuda:0")
H, D = 16, 128
n = 16384 * H * D
q = torch.empty(n, dtype=torch.float16, device=dev); k = torch.empty_like(q); v = torch.empty_like(q); o = torch.empty_like(q)
fa.init_half(q, 1234); fa.init_half(k, 2345); fa.init_half(v, 3456)
N, WG = 8, 256
stamps = torch.zeros(N * WG * 4, dtype=torch.int64, device=dev)
p = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), fa.HalfPtr)
B = 16384 // L
t0 = time.time()
while time.time() - t0 < 1.5:
    for _ in range(50):
        lib.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None)
    torch.cuda.synchronize()
for i in range(N):
    lib.fa_debug_set_stamps(ct.c_void_p(stamps.data_ptr() + i * WG * 4 * 8))
    lib.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(N, WG, 4)
for i in (3, 6):
    st = (s[i, :, 1] - s[i, :, 1].min()) / 100.0
    en = (s[i, :, 3] - s[i, :, 1].min()) / 100.0
    du = en - st
    cyc = s[i, :, 2] - s[i, :, 0]
    print(f"launch {i}: start spread {st.max():.2f} us; duration min {du.min():.1f} med {np.median(du):.1f} max {du.max():.1f}; end min {en.min():.1f} med {np.median(en):.1f} max {en.max():.1f}; cycles min {cyc.min()} med {int(np.median(cyc))} max {cyc.max()}")
    for x in range(8):
        m = np.arange(WG) % 8 == x
        print(f"   xcd {x}: start med {np.median(st[m]):.2f}  dur med {np.median(du[m]):.1f} max {du[m].max():.1f}  end max {en[m].max():.1f}  clock {np.median(cyc[m]/du[m]):.0f} MHz")
    order = np.argsort(-du)[:8]
    print("   slowest:", [(int(w), round(float(du[w]),1), round(float(st[w]),2)) for w in order])
