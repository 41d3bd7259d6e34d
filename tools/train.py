#!/usr/bin/env python3
"""Diagnostic: where the time between back-to-back launches goes (library built with -DFA_STAMPS).

    python tools/train.py build_ab/NAME.so [L] [pattern|normal] [launches] [warm-up seconds]

Each launch of a train of back-to-back launches writes its workgroups' entry/exit stamps
(s_memtime, s_memrealtime) to its own buffer.  s_memrealtime is one 100 MHz counter for the whole
chip, so consecutive launches can be laid on one time axis: per launch the tool prints the
period (first entry to the next launch's first entry), the busy span (first entry to last exit),
the gap to the next launch, the median in-kernel time of a workgroup and the shader clock."""
import ctypes as ct, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
lib = ct.CDLL(os.path.join(ROOT, sys.argv[1]))
L = int(sys.argv[2]) if len(sys.argv) > 2 else 512
data = sys.argv[3] if len(sys.argv) > 3 else "pattern"
N = int(sys.argv[4]) if len(sys.argv) > 4 else 24
WARM_S = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
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
WG = 256
stamps = torch.zeros(N * WG * 4, dtype=torch.int64, device=dev)
p = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), fa.HalfPtr)
B = 16384 // L
t0 = time.time()
while time.time() - t0 < WARM_S:
    for _ in range(50):
        lib.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None)
    torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(N):
    lib.fa_debug_set_stamps(ct.c_void_p(stamps.data_ptr() + i * WG * 4 * 8))
    lib.fa_forward(p(q), p(k), p(v), p(o), B, H, L, D, 64, None, None)
e1.record()
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(N, WG, 4)
first = s[:, :, 1].min(axis=1) / 100.0
last = s[:, :, 3].max(axis=1) / 100.0
dur = np.median((s[:, :, 3] - s[:, :, 1]) / 100.0, axis=1)
clk = np.median((s[:, :, 2] - s[:, :, 0]) / np.maximum(s[:, :, 3] - s[:, :, 1], 1) * 100.0, axis=1)
print(f"{sys.argv[1]} L={L} {data}: {N} launches, {e0.elapsed_time(e1)*1e3/N:.1f} us per launch by events")
for i in range(N):
    period = first[i + 1] - first[i] if i + 1 < N else float('nan')
    gap = first[i + 1] - last[i] if i + 1 < N else float('nan')
    print(f"  launch {i:2d}: period {period:6.1f} us  busy {last[i]-first[i]:6.1f}  gap {gap:5.1f}  workgroup median {dur[i]:6.1f}  clock {clk[i]:5.0f} MHz")
per = np.diff(first)
print(f"  median period {np.median(per):.1f} us, busy {np.median(last-first):.1f}, gap {np.median(first[1:]-last[:-1]):.1f}, clock {np.median(clk):.0f} MHz")
