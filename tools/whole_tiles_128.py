#!/usr/bin/env python3
"""D = 128, whole tiles, no mask: the tuned kernels (fa_forward) next to fa_fwd_hd_masked_kernel<128> (reached through
fa_forward_varlen with every k_len = L), B*L = 16384 rows, H = 16, pattern data.

    python tools/whole_tiles_128.py [L | LqxLk ...]"""
import os, sys, time
os.environ["FA_VARLEN_W64"] = "0"  # (padded batches stay on the 128-row kernel: that is the comparison)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
dev = torch.device("cuda:0")
H, D = 16, 128
n = 16384 * H * D
buf = [torch.empty(n, dtype=torch.float16, device=dev) for _ in range(4)]
for t, seed in zip(buf[:3], (1234, 2345, 3456)):
    fa.init_half(t, seed)


def timeit(f, reps=20, pre=0.4):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < pre:
        for _ in range(10):
            f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for arg in sys.argv[1:] or [str(x) for x in range(64, 1089, 64)]:
    Lq, Lk = [int(x) for x in arg.split("x")] if "x" in arg else (int(arg), int(arg))
    B = max(1, 16384 // max(Lq, Lk))
    q, o = [t[:B * H * Lq * D].view(B, H, Lq, D) for t in (buf[0], buf[3])]
    k, v = [t[:B * H * Lk * D].view(B, H, Lk, D) for t in buf[1:3]]
    kl = torch.full((B,), Lk, dtype=torch.int32, device=dev)
    a = timeit(lambda: fa.attention_general(q, k, v, out=o))
    o1 = o.clone()
    b = timeit(lambda: fa.attention_varlen(q, k, v, k_len=kl, out=o))
    d = float((o.float() - o1.float()).abs().max())
    fl = 4.0 * B * H * Lq * Lk * D
    print(f"Lq={Lq:5d} Lk={Lk:5d} B={B:3d} last={(Lq - 1) % 256 + 1:3d}  {fa.lib().fa_kernel_name_general(Lq, Lk, D, 0).decode():28s} {fl / a / 1e9:7.1f} TF"
          f"   hd_masked<128> {fl / b / 1e9:7.1f} TF   ratio {a / b:.3f}  max|diff| {d:.1e}", flush=True)
