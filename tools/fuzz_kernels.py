#!/usr/bin/env python3
"""Soak test of the three kernels that serve 256-row query blocks.

    python tools/fuzz_kernels.py [--iters 300] [--seed 0]

Random shapes (B, H, L = 64k), random input scales, interleaved with launches on other data so that
registers and LDS hold something else each time.  Every kernel must repeat its own result bit for
bit; the 8-wave ring kernel (FA_W64=0) and the register-staged kernel (FA_W64=0 FA_RING=0) run the
same arithmetic in the same order per row and must agree bit for bit with each other; the
one-wave-per-SIMD kernel adds a row's exponentials in a different order, so it may differ from them
by the rounding of the result (one fp16 ulp).  Anything else is a scheduling, synchronisation or
initialisation fault.  The launcher's key split (few heads on a long sequence: FA_SPLIT unset) folds
partial sums of key ranges, another arithmetic again: it is run beside the undivided launch
(FA_SPLIT=1) and held to a bound relative to the row's largest output.

    python tools/fuzz_kernels.py --causal [--iters 300]

The same for the masked forms of the one-wave-per-SIMD kernel (FA_MASKED_W64=1: wherever they can run)
against the general kernel (FA_MASKED_W64=0 FA_HD=1) and the masked form of the pipelined 128-row kernel
(FA_MASKED_W64=0): with and without the causal mask, L from 192 up (half of the cases not a multiple of 64)
and B*H up to 400 so that workgroups walk several items of different lengths.  Each must repeat itself bit
for bit; the first two may differ by the rounding of the result; the third walks the keys in tiles of 32,
so its rows rescale at other points and it is held to the bound relative to the row's largest output."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--causal", action="store_true")
args = ap.parse_args()
fa = ge.load_package()
if not args.causal:  # the register-staged kernel (FA_RING=0) lives in the comparison build only
    fa._lib = fa.bind(os.path.join(ROOT, "flash-attention_amd", "csrc", "libflash_attention_ab.so"))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(args.seed)
cpu = torch.Generator().manual_seed(args.seed)
# (FA_W64=1: from three tiles on; FA_SPLIT=1: never divide the keys; the last one is the launcher's own choice)
envs = [{"FA_W64": "1", "FA_SPLIT": "1"}, {"FA_W64": "0"}, {"FA_W64": "0", "FA_RING": "0"}, {"FA_W64": "1"}]

if args.causal:
    envs = [{"FA_MASKED_W64": "1"}, {"FA_MASKED_W64": "0", "FA_HD": "1"}, {"FA_MASKED_W64": "0"}]

def run(q, k, v, env):
    for a, b in env.items():
        os.environ[a] = b
    fa.reload_env()  # (the switches are read once per process)
    try:
        if args.causal and env:
            return fa.attention_general(q, k, v, causal=mask)
        return fa.attention(q, k, v)
    finally:
        for a in env:
            os.environ.pop(a, None)
        fa.reload_env()

bad = 0
noise = [torch.randn((2, 8, 1024, 128), generator=g, device=dev).to(torch.float16) * 30 for _ in range(3)]
for it in range(args.iters):
    B = int(torch.randint(1, 5, (1,), generator=cpu))
    H = int(torch.randint(1, 21, (1,), generator=cpu))
    L = 64 * int(torch.randint(1, 97, (1,), generator=cpu))  # (a partly filled last 256-row block in 3 of 4 cases)
    mask = True
    if args.causal:
        L = 64 * int(torch.randint(3, 49, (1,), generator=cpu))
        H = int(torch.randint(1, 101, (1,), generator=cpu)) if L <= 1024 else H
        mask = bool(torch.randint(0, 2, (1,), generator=cpu))
        if not mask or bool(torch.randint(0, 2, (1,), generator=cpu)):
            L += int(torch.randint(1, 64, (1,), generator=cpu))  # (without the mask only ragged lengths go there)
    scale = float(torch.rand(1, generator=cpu)) * 2.5 + 0.2
    q, k, v = [(torch.randn((B, H, L, 128), generator=g, device=dev) * scale + 0.3 * (i == 0)).to(torch.float16) for i in range(3)]
    outs = []
    for env in envs:
        run(*noise, {})  # something else passes through the CUs in between
        outs.append(run(q, k, v, env))
        outs.append(run(q, k, v, env))
    torch.cuda.synchronize()
    a, b = outs[0].float(), outs[2].float()
    mag = torch.maximum(a.abs(), b.abs())
    if args.causal and L % 64 != 0:
        # Ragged lengths: the masked kernel cuts the keys into tiles differently (its last tile ends at the last
        # key) and groups the last query rows differently, so a row's deferred-rescale reference point, and with
        # it the fp16 rounding of its probabilities (2^-11 relative each), differs from the general kernel's.
        # That noise scales with the row's values, not with the single output element: the bound is relative
        # to the row's largest output.  (Both kernels sit at the same distance from the CPU oracle:
        # tests/soak/diag_ragged.py.)
        mag = mag.amax(dim=-1, keepdim=True)
    close = bool(((a - b).abs() <= 2.0 ** -9 * mag + 1e-6).all())
    if args.causal:
        c = outs[4].float()
        row = torch.maximum(a.abs(), c.abs()).amax(dim=-1, keepdim=True)
        close = close and bool(((a - c).abs() <= 2.0 ** -9 * row + 1e-6).all())
        same = all(torch.equal(outs[i], outs[i + 1]) for i in (0, 2, 4))
    else:
        same = torch.equal(outs[0], outs[1]) and all(torch.equal(outs[2], o) for o in outs[3:6]) and torch.equal(outs[6], outs[7])
        c = outs[6].float()  # (the key split, where the launcher chooses it)
        row = torch.maximum(a.abs(), c.abs()).amax(dim=-1, keepdim=True)
        close = close and bool(((a - c).abs() <= 2.0 ** -9 * row + 1e-6).all())
    ok = same and close and bool(torch.isfinite(a).all())
    if not ok:
        bad += 1
        d = [float((outs[0].float() - o.float()).abs().max()) for o in outs]
        print(f"iter {it}: B={B} H={H} L={L} mask={mask} scale={scale:.2f}: MISMATCH max|diff| per run {d}")
    if it % 50 == 49:
        print(f"{it + 1} shapes, {bad} mismatches", flush=True)
print("fuzz:", "FAILED" if bad else "ok", f"({args.iters} shapes, {bad} mismatches)")
sys.exit(1 if bad else 0)
