#!/usr/bin/env python3
"""Throughput of fa_forward_general (ragged lengths, causal mask) next to the tuned kernels.

    python tools/bench_general.py

Pattern data of the reference's throughput driver, 1 s of preheat, back-to-back launches between two
events.  FLOPs: 4*B*H*Lq*Lk*D without mask, half of that (the visible triangle, diagonal included:
2*B*H*L*(L+1)*D) with the causal mask.  --causal-sweep / --ragged-sweep: the masked tuned kernel (FA_MASKED_W64=1) and the 128-row kernel
that takes its place (FA_MASKED_W64=0: labelled "general kernel"; fa_fwd_hd_masked_kernel<128>, or
fa_fwd_general_kernel<128> with FA_HD=1 in the environment) side by side over L; --only-sweep skips the table before it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
dev = torch.device("cuda:0")
H, D = 16, 128
n = 16384 * H * D + 4096
buf = [torch.empty(n, dtype=torch.float16, device=dev) for _ in range(4)]
for t, seed in zip(buf[:3], (1234, 2345, 3456)):
    fa.init_half(t, seed)

PREHEAT = 0.3 if "--fine" in sys.argv else 1.0


def run(label, B, L, causal, general=True, reps=20, D=128, tuned_causal=True):
    """tuned_causal: None = the library's own choice, True / False = FA_MASKED_W64=1 / 0"""
    os.environ.pop("FA_MASKED_W64", None)
    if tuned_causal is not None:
        os.environ["FA_MASKED_W64"] = "1" if tuned_causal else "0"
    fa.reload_env()  # (the switches are read once per process)
    m = B * H * L * D
    q, k, v, o = [t[:m].view(B, H, L, D) for t in buf]
    f = (lambda: fa.attention_general(q, k, v, causal=causal, out=o)) if general else (lambda: fa.attention(q, k, v, out=o))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < PREHEAT:
        for _ in range(10):
            f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * B * H * L * (L + 1) * D if causal else 4.0 * B * H * L * L * D
    print(f"{label:44s} B={B:2d} L={L:5d} D={D:3d}  {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s")

for B, L in () if "--only-sweep" in sys.argv else ((32, 512), (8, 2048), (1, 16384)):
    run("tuned kernel, no mask", B, L, False, general=False)
    run("masked tuned kernel, no mask, L-1 (ragged)", B, L - 1, False)
    run("general kernel, no mask, L-1 (ragged)", B, L - 1, False, tuned_causal=False)
    run("masked tuned kernel, causal", B, L, True)
    run("general kernel, causal", B, L, True, tuned_causal=False)
    run("masked tuned kernel, causal, L-1", B, L - 1, True)
    run("general kernel, causal, L-1", B, L - 1, True, tuned_causal=False)
if "--ragged-sweep" in sys.argv:
    Ls = [L for L0 in range(192, 2305, 64) for L in (L0, L0 + 1)] if "--fine" in sys.argv else (193, 257, 321, 385, 449, 511, 577, 767, 1023)
    if "--lengths" in sys.argv:  # --lengths 2305,2561,...
        Ls = [int(x) for x in sys.argv[sys.argv.index("--lengths") + 1].split(",")]
    for L in Ls:
        B = max(1, 16384 // L)
        for causal in (False, True):
            if not causal and L % 64 == 0:
                continue
            run(f"masked tuned kernel, causal={causal}", B, L, causal)
            run(f"general kernel, causal={causal}", B, L, causal, tuned_causal=False)
    sys.exit(0)
if "--causal-sweep" in sys.argv:
    for L in (256, 320, 384, 512, 768, 1024, 2048, 4096, 8192, 16384):
        B = max(1, 16384 // L)
        run("tuned kernel, causal", B, L, True)
        run("general kernel, causal", B, L, True, tuned_causal=False)
    sys.exit(0)
for D_, shapes in ((64, ((8, 2048), (1, 8192))), (256, ((4, 2048), (1, 8192)))):
    for B, L in shapes:
        run("general kernel, no mask", B, L, False, D=D_)
        run("general kernel, causal", B, L, True, D=D_)
