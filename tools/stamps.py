#!/usr/bin/env python3
"""Diagnostic: per-workgroup time stamps (s_memrealtime, 100 MHz) from a -DFA_STAMPS build.
Prints, per README shape, the median over workgroups of each interval in microseconds."""
import ctypes as ct, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
L_ = ct.CDLL(os.path.join(ROOT, "build_ab", os.environ.get("FA_STAMPS_LIB", "stamps.so")))
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
for L in [int(x) for x in sys.argv[1:]] or [512, 1024, 2048]:
    B = 16384 // L
    L_.fa_debug_set_stamps(None)
    for _ in range(10):
        L_.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None)
    torch.cuda.synchronize()
    stamps.zero_(); torch.cuda.synchronize()
    L_.fa_debug_set_stamps(ct.c_void_p(stamps.data_ptr()))
    L_.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None)
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy().reshape(256, 64).astype(np.float64)
    s = raw / 100.0  # us
    nst = int((s[0, :60] > 0).sum())
    t0 = s[:, 0].min()
    rel = s[:, :nst] - t0
    print(f"--- L={L}: {nst} stamps per workgroup; kernel span {rel.max():.1f} us; start spread {rel[:,0].max():.2f} us")
    names = [str(i) for i in range(64)]
    names_old = ["start"] + sum([[f"it{i}:tiles_done", f"it{i}:last_pv_done", f"it{i}:staged+qfetch", f"it{i}:out_issued", f"it{i}:next_ready"] for i in range(8)], [])
    prev = rel[:, 0]
    for j in range(1, nst):
        d = rel[:, j] - rel[:, j - 1]
        print(f"   {names[j]:18s} +{np.median(d):7.2f} us (min {d.min():6.2f} max {d.max():6.2f})   at {np.median(rel[:, j]):7.2f}")
