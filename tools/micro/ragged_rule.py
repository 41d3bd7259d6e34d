#!/usr/bin/env python3
"""Where the ragged tuned kernel beats the 128-row kernel on one unmasked sequence length, by the fill of the last
256-row block: L = base + r for bases of whole blocks and r rows in the last one; FA_MASKED_W64=1 against 0
(B*L ~ 16384, H = 16, pattern data, 0.4 s preheat per line).  The launcher's rule (masked_on_w64) is fitted to this.
    python tools/micro/ragged_rule.py [--causal] [--heads H]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
dev = torch.device("cuda:0")
H, D = 16, 128
if "--heads" in sys.argv:  # (fewer heads: batches of one to two rounds of 256-row items)
    H = int(sys.argv[sys.argv.index("--heads") + 1])
causal = "--causal" in sys.argv
n = 16384 * H * D + 16384 * H * D
buf = [torch.empty(n, dtype=torch.float16, device=dev) for _ in range(4)]
for t, seed in zip(buf[:3], (1234, 2345, 3456)):
    fa.init_half(t, seed)


def run(L, tuned):
    """tuned: True / False = FA_MASKED_W64=1 / 0, None = the library's own choice"""
    os.environ.pop("FA_MASKED_W64", None)
    if tuned is not None:
        os.environ["FA_MASKED_W64"] = "1" if tuned else "0"
    fa.reload_env()
    B = max(1, 16384 // L)
    m = B * H * L * D
    q, k, v, o = [t[:m].view(B, H, L, D) for t in buf]
    f = lambda: fa.attention_general(q, k, v, causal=causal, out=o)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(10):
            f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


bases = (0, 256, 512, 768, 1024, 1280) if not causal else (0, 256, 512, 768, 1024, 1280, 1536, 1792, 2048, 2304, 2560)
for base in bases:
    for r in ((1, 65, 129, 193, 255) if not causal else (1, 64, 129, 192, 255, 256)):  # (with the mask whole tiles count too)
        L = base + r
        if L < 192:
            continue
        a, b, c = run(L, True), run(L, False), run(L, None)
        name = fa.lib().fa_kernel_name_general_bh(max(1, 16384 // L) * H, L, L, D, 1 if causal else 0).decode()
        print(f"L={L:5d} (last block {r:3d} rows)  tuned {a:8.1f} us   128-row {b:8.1f} us   tuned/128-row {a / b:.3f}"
              f"   library's choice {c:8.1f} us ({name}; best/choice {min(a, b) / c:.3f})", flush=True)
os.environ.pop("FA_MASKED_W64", None)
