#!/usr/bin/env python3
"""Interleaved A/B timing of several builds of libflash_attention.so in ONE process.

    python tools/ab.py --libs build_ab/a.so build_ab/b.so [--env FA_W64=0 ...] --shapes 512 2048 16384

(FA_NWAVES and FA_RING select the register-staged kernels, which only builds with -DFA_AB_KERNELS hold:
flash-attention_amd/csrc/libflash_attention_ab.so, or tools/build_variant.sh NAME -DFA_AB_KERNELS.)

Each (lib, env) pair is a variant; per shape the variants are timed round-robin
(rounds x reps launches each) on the reference driver's input pattern, and the
median / min per-launch time and TFLOP/s are printed.  With --check every variant's
output is compared with the first variant's.
"""
import argparse
import ctypes as ct
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--env", nargs="*", default=[], help="extra variants: KEY=VAL[,KEY=VAL] applied to every lib")
    ap.add_argument("--shapes", nargs="+", type=int, default=[512, 1024, 2048, 4096, 8192, 16384])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--preheat", type=float, default=1.0,
                    help="seconds of untimed launches per shape before timing (the chip's clock needs sustained load to settle)")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--causal", action="store_true", help="fa_forward_general with the causal mask instead of fa_forward")
    ap.add_argument("--lminus", type=int, default=0, help="shorten every L by this much (ragged lengths, fa_forward_general)")
    ap.add_argument("--data", choices=["pattern", "normal"], default="pattern",
                    help="pattern = the reference driver's init_half; normal = N(0,1) rounded to fp16")
    args = ap.parse_args()

    import torch

    import __graft_entry__ as ge

    fa = ge.load_package()  # pulls in torch's HIP runtime first
    HalfPtr = fa.HalfPtr
    sig = [HalfPtr] * 4 + [ct.c_int] * 5 + [ct.c_void_p, ct.POINTER(ct.c_float)]
    envs = [dict()] + [dict(kv.split("=") for kv in e.split(",")) for e in args.env]
    variants = []
    for path in args.libs:
        L = ct.CDLL(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
        L.fa_forward.argtypes = sig
        L.fa_forward.restype = ct.c_int
        if args.causal or args.lminus:
            L.fa_forward_general.argtypes = [HalfPtr] * 4 + [ct.c_int] * 6 + [ct.c_void_p, ct.POINTER(ct.c_float)]
            L.fa_forward_general.restype = ct.c_int
        for e in envs:
            name = os.path.basename(path) + ("" if not e else "[" + ",".join(f"{k}={v}" for k, v in e.items()) + "]")
            variants.append((name, L, e))

    dev = torch.device("cuda:0")
    H, D = 16, 128
    n = 16384 * H * D
    q = torch.empty(n, dtype=torch.float16, device=dev)
    k, v = torch.empty_like(q), torch.empty_like(q)
    if args.data == "pattern":
        fa.init_half(q, 1234), fa.init_half(k, 2345), fa.init_half(v, 3456)
    else:
        g = torch.Generator(device=dev).manual_seed(0)
        for t in (q, k, v):
            t.copy_(torch.randn(n, generator=g, device=dev, dtype=torch.float32).to(torch.float16))
    # random-looking magnitudes matter for clocks; the pattern above is what the reference driver uses
    outs = [torch.empty_like(q) for _ in variants]
    stream = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), HalfPtr)

    def launch(i, B, L):
        name, lib, e = variants[i]
        for kk, vv in e.items():
            os.environ[kk] = vv
        if e:
            lib.fa_debug_reload_env()  # (the switches are read once per process)
        if args.causal or args.lminus:
            rc = lib.fa_forward_general(p(q), p(k), p(v), p(outs[i]), B, H, L, L, D, int(args.causal), stream, None)
        else:
            rc = lib.fa_forward(p(q), p(k), p(v), p(outs[i]), B, H, L, D, 64, stream, None)
        for kk in e:
            os.environ.pop(kk, None)
        if e:
            lib.fa_debug_reload_env()
        assert rc == 0, (name, rc)

    for L in args.shapes:
        B = max(1, 16384 // L)
        L -= args.lminus
        flops = 2.0 * B * H * L * (L + 1) * D if args.causal else 4.0 * B * H * L * L * D
        times = [[] for _ in variants]
        for i in range(len(variants)):
            for _ in range(5):
                launch(i, B, L)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.preheat:
            for i in range(len(variants)):
                for _ in range(25):
                    launch(i, B, L)
            torch.cuda.synchronize()
        for _ in range(args.rounds):
            for i in range(len(variants)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    launch(i, B, L)
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / args.reps)
        print(f"--- B={B} H={H} L={L}")
        for i, (name, _, _) in enumerate(variants):
            med, mn = statistics.median(times[i]), min(times[i])
            line = f"  {name:48s} median {med*1e3:9.1f} us  {flops/med/1e9:8.1f} TF   min {mn*1e3:9.1f} us  {flops/mn/1e9:8.1f} TF"
            if args.check and i > 0:
                d = (outs[i].float() - outs[0].float()).abs().max().item()
                line += f"   max|diff vs first| {d:.2e}"
            print(line)


if __name__ == "__main__":
    main()
