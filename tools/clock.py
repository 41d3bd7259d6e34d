#!/usr/bin/env python3
"""Diagnostic: in-kernel shader clock of the forward kernel (library built with -DFA_STAMPS).

    python tools/clock.py build_ab/NAME.so [L] [pattern|normal]

Every workgroup records s_memtime (shader cycles) and s_memrealtime (100 MHz) at entry and exit;
clock = d(memtime) / d(memrealtime) * 100 MHz, median over workgroups, after about two seconds of
back-to-back launches on the same data."""
import ctypes as ct, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
lib = ct.CDLL(os.path.join(ROOT, sys.argv[1]))
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
data = sys.argv[3] if len(sys.argv) > 3 else "pattern"
sig = [fa.HalfPtr] * 4 + [ct.c_int] * 5 + [ct.c_void_p, ct.POINTER(ct.c_float)]
lib.fa_forward.argtypes = sig; lib.fa_forward.restype = ct.c_int
lib.fa_debug_set_stamps.argtypes = [ct.c_void_p]
dev = torch.device("cuda:0")
H, D = 16, 128
n = 16384 * H * D
q = torch.empty(n, dtype=torch.float16, device=dev); k = torch.empty_like(q); v = torch.empty_like(q); o = torch.empty_like(q)
if data == "pattern":
    fa.init_half(q, 1234); fa.init_half(k, 2345); fa.init_half(v, 3456)
else:
    g = torch.Generator(device=dev).manual_seed(0)
    for t in (q, k, v):
        t.copy_(torch.randn(n, generator=g, device=dev, dtype=torch.float32).to(torch.float16))
stamps = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
p = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), fa.HalfPtr)
B = 16384 // L
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        lib.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None)
    torch.cuda.synchronize()
lib.fa_debug_set_stamps(ct.c_void_p(stamps.data_ptr()))
ms = ct.c_float(0)
for _ in range(5):
    lib.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, ct.byref(ms))
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 4)[:256]
s = s[s[:, 3] > 0]
clk = (s[:, 2] - s[:, 0]) / np.maximum(s[:, 3] - s[:, 1], 1) * 100.0
dur = (s[:, 3] - s[:, 1]) / 100.0
span = (s[:, 3].max() - s[:, 1].min()) / 100.0
spread = (s[:, 1].max() - s[:, 1].min()) / 100.0
print(f"{sys.argv[1]} L={L} {data}: launch {ms.value*1e3:.1f} us; in-kernel {np.median(dur):.1f} us; "
      f"shader clock median {np.median(clk):.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f}); "
      f"cycles per workgroup {np.median(s[:,2]-s[:,0]):.0f}; first entry to last exit {span:.1f} us, entries spread over {spread:.1f} us")
