#!/usr/bin/env python3
"""Diagnostic: phase timing of the ping-pong kernel (build with -DFA_STAMPS): s_memtime of wave 0
and wave 4 of workgroup 0 at: step start, end of vector phase, after barrier, end of matrix math,
after tile write (before barrier).  Prints cycle deltas."""
import ctypes as ct, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
L_ = ct.CDLL(os.path.join(ROOT, "build_ab", "pp_stamps.so"))
sig = [fa.HalfPtr] * 4 + [ct.c_int] * 5 + [ct.c_void_p, ct.POINTER(ct.c_float)]
L_.fa_forward.argtypes = sig; L_.fa_forward.restype = ct.c_int
L_.fa_debug_set_stamps.argtypes = [ct.c_void_p]
dev = torch.device("cuda:0")
H, D = 16, 128
n = 16384 * H * D
q = torch.empty(n, dtype=torch.float16, device=dev); k = torch.empty_like(q); v = torch.empty_like(q); o = torch.empty_like(q)
fa.init_half(q, 1234); fa.init_half(k, 2345); fa.init_half(v, 3456)
stamps = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
p = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), fa.HalfPtr)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
B = 16384 // L
for _ in range(20):
    L_.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None)
torch.cuda.synchronize()
L_.fa_debug_set_stamps(ct.c_void_p(stamps.data_ptr()))
ms = ct.c_float(0)
tot = 0.0
for _ in range(10):
    L_.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, ct.byref(ms))
    tot += ms.value
torch.cuda.synchronize()
steps = (L // 64) * max(1, (B * H * (L // 256) + 255) // 256)
print(f"stamped launch: {tot / 10 * 1e3:.1f} us, {steps} steps per workgroup -> {tot / 10 * 1e3 / steps:.3f} us per step")
s = stamps.cpu().numpy()
names = ["step start", "Y done", "after barrier", "X math done", "tile written"]
for w, off in ((0, 0), (4, 64)):
    a = s[off:off + 60]
    a = a[a > 0]
    print(f"wave {w}: {len(a)} stamps")
    for i in range(1, min(len(a), 31)):
        print(f"   {names[i % 5]:14s} +{a[i] - a[i-1]:6d} cycles   (t={a[i]-s[0]:7d})")
