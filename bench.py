#!/usr/bin/env python3
"""Throughput benchmark of the forward-attention hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one forward pass of the hot path over one batch of synthetic input,
called through the C ABI (device-pointer entry point) with inputs already in
HBM.  The default workload is README row 1 of the reference (B=32,H=16,L=512,
D=128; flash_attention_tensor_cores/benchmark_tensor_flash_attention.cu:49-51),
filled with that driver's input pattern (:34-41, seeds 1234/2345/3456).
FLOPs = 4*B*H*L^2*D (:218).

With N>1 (launched by torch.distributed.run, one rank per GPU) every rank runs
the same workload on its own GPU: the path shards over independent (batch,
head) pairs, so there is no data-path collective; the only collectives are the
barrier and the max-over-ranks of the timing.  --scaling strong instead splits
the B*H heads of ONE workload across the ranks.

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      MFMA roofline of the attention kernel from HIP events over the timed region
  cpu_baseline  the CPU oracle (a port of the reference's attention_cpu) timed on this host
  sweep         the README shapes on this GPU (back-to-back launches between two events)
  causal_sweep  the same shapes with the causal mask (fa_forward_general), TFLOP/s of the visible triangle
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

README_SHAPES = [(32, 512), (16, 1024), (8, 2048), (4, 4096), (2, 8192), (1, 16384)]  # (B, L), H=16, D=128
H_DEFAULT, D = 16, 128
# README.md:27-32 of the reference (RTX 3090), TFLOP/s of its kernel per L
PUBLISHED_TFLOPS = {512: 9.41, 1024: 9.79, 2048: 10.91, 4096: 10.94, 8192: 10.95, 16384: 10.96}
PEAK_TFLOPS = 2500.0  # dense fp16 MFMA, MI355X
PEAK_HBM_GBPS = 8000.0  # HBM3E


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seq-len", type=int, default=512, help="L of the README shape to time (B = 16384 / L)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--preheat-seconds", type=float, default=1.5,
                    help="untimed back-to-back launches before the warm-up steps, so that the chip's clock governor "
                         "has settled (0 disables)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


def preheat(fa, torch, q, k, v, o, seconds, call=None):
    """The chip raises its clock only under sustained load: measured on L=512, launches that follow
    a few milliseconds of work run at about 1.95 GHz (75 us per launch), launches that follow a second
    of work at about 2.25 GHz (66 us).  Steady-state throughput is what the metric means, so the timed
    region is preceded by this much untimed work, on top of the --warmup steps."""
    call = call or fa.attention
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            call(q, k, v, out=o)
        torch.cuda.synchronize()


def time_steps(fa, torch, q, k, v, o, steps, warmup, barrier, preheat_s=0.0, call=None):
    call = call or fa.attention
    preheat(fa, torch, q, k, v, o, preheat_s, call)
    for _ in range(warmup):
        call(q, k, v, out=o)
    torch.cuda.synchronize()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        call(q, k, v, out=o)
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier()
    return wall, ev0.elapsed_time(ev1) / steps  # seconds for all steps, ms per launch (events)


def roofline(fa, L, n_local, flops_local, ms_launch):
    """The roof that binds the attention kernel for this shape.  Algorithmic work per launch: 4*L^2*D
    FLOP and 4*L*D*2 bytes per (batch, head) pair (Q, K, V read once, O written once), i.e. L/2 FLOP
    per byte.  The machine balance is 2500 TFLOP/s / 8 TB/s = 312 FLOP/B: L=512 (256 FLOP/B) lies
    below the ridge, so its floor is the HBM one (33.6 us against 27.5 us on the matrix pipe); from
    L=1024 up the matrix pipe binds.  Both fractions are reported either way."""
    algo_bytes = 4 * n_local * 2
    sec = ms_launch * 1e-3
    tflops = flops_local / sec / 1e12
    gbps = algo_bytes / sec / 1e9
    hbm_bound = algo_bytes / (PEAK_HBM_GBPS * 1e9) > flops_local / (PEAK_TFLOPS * 1e12)
    out = {
        "bound": "hbm" if hbm_bound else "mfma",
        "achieved": gbps if hbm_bound else tflops,
        "peak": PEAK_HBM_GBPS if hbm_bound else PEAK_TFLOPS,
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": (gbps / PEAK_HBM_GBPS) if hbm_bound else (tflops / PEAK_TFLOPS),
        "traffic": measured_traffic(L),
        "kernel": fa.lib().fa_kernel_name(L, L).decode(),
        "kernel_ms": ms_launch,
        "algorithmic_bytes": algo_bytes,
        "algorithmic_flops": flops_local,
        "mfma_tflops": tflops,
        "mfma_frac": tflops / PEAK_TFLOPS,
        "hbm_GBps": gbps,
        "hbm_frac": gbps / PEAK_HBM_GBPS,
    }
    return out


def cpu_baseline(B, H, L, target_s):
    """Oracle (port of attention_cpu, testing.cu:41-132) on this host's cores, on a bounded
    sample of the workload: whole (batch, head) slices with the same L and D."""
    import numpy as np

    import oracle

    oracle.build()
    threads = oracle.max_threads()
    per_head = 4.0 * L * L * D
    probe = np.stack([oracle.init_half(L * D, s).reshape(1, 1, L, D) for s in (1234, 2345, 3456)])
    t = time.perf_counter()
    oracle.attention_cpu(probe[0], probe[1], probe[2], nthreads=0)
    t_probe = max(time.perf_counter() - t, 1e-4)
    # threads are spread over rows of one head at a time, so one head is a fair probe
    n_heads = int(max(1, min(B * H, target_s / t_probe)))
    n = n_heads * L * D
    Q, K, V = [oracle.init_half(n, s).reshape(1, n_heads, L, D) for s in (1234, 2345, 3456)]
    # when even the whole workload is short on this host, repeat it until ~target_s has been spent
    reps = int(max(1, min(50, target_s / max(t_probe * n_heads, 1e-3))))
    t = time.perf_counter()
    for _ in range(reps):
        oracle.attention_cpu(Q, K, V, nthreads=0)
    dt = (time.perf_counter() - t) / reps
    # the literal single-threaded loop nest of the reference, on one head of at most 512 rows
    Ls = min(L, 512)
    t = time.perf_counter()
    oracle.attention_cpu(np.ascontiguousarray(Q[:, :1, :Ls]), np.ascontiguousarray(K[:, :1, :Ls]),
                         np.ascontiguousarray(V[:, :1, :Ls]), literal=True)
    dt_lit = time.perf_counter() - t
    return {
        "value": n_heads * per_head / dt / 1e12,
        "unit": "TFLOP/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n_heads} of {B * H} (batch,head) slices of the workload (L={L}, D={D}), "
                  f"{n_heads * per_head / 1e9:.1f} GFLOP in {dt:.3f} s on {threads} OpenMP threads, mean of {reps} passes",
        "host_cpus": os.cpu_count(),
        "literal_single_thread_gflops": 4.0 * Ls * Ls * D / dt_lit / 1e9,
    }


def measured_traffic(L):
    """HBM bytes per launch of this shape from the newest committed rocprofv3 counter summary
    (profiles/*_L<L>.json, written by tools/summarize_profile.py from separate FETCH_SIZE and
    WRITE_SIZE passes); None when this shape has not been profiled."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_L{L}.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))["hbm_bytes"])
    except Exception:
        return None


def main():
    args = parse_args()
    import torch

    import __graft_entry__ as ge

    fa = ge.load_package()
    fa.lib()  # fail loudly if the HIP library is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # (FA_BENCH_FORCE_DIST=1 takes the multi-rank code path with a single rank: RCCL init, barrier and
    # the max-over-ranks reduction can then be exercised on a one-GPU box)
    if world > 1 or os.environ.get("FA_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    L = args.seq_len
    if L % 64 or L <= 0:
        raise SystemExit("--seq-len must be a positive multiple of 64")
    B, H = max(1, 16384 // L), H_DEFAULT
    BH_total = B * H
    if args.scaling == "strong" and world > 1:
        if BH_total % world:
            raise SystemExit("strong scaling needs B*H divisible by the number of GPUs")
        bh_local, head0 = BH_total // world, rank * (BH_total // world)
    else:
        bh_local, head0 = BH_total, 0

    # inputs: the driver's pattern depends only on the flat element index; a strong-scaling shard
    # gets its own window of the global tensors
    n_local = bh_local * L * D
    full = [torch.empty((BH_total, L, D), dtype=torch.float16, device=dev) for _ in range(3)]
    for t, seed in zip(full, (1234, 2345, 3456)):
        fa.init_half(t, seed)
    q, k, v = [t[head0:head0 + bh_local].reshape(1, bh_local, L, D) for t in full]  # contiguous slabs
    o = torch.empty_like(q)
    torch.cuda.synchronize()

    # The first collective sets the communicator up, which takes long enough for the chip to fall back to
    # its idle clock: it must not sit between the preheat and the timed steps.
    barrier()
    wall, ms_launch = time_steps(fa, torch, q, k, v, o, args.steps, args.warmup, barrier, args.preheat_seconds)
    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall_max = float(wall_t.item())

    flops_local = 4.0 * bh_local * L * L * D
    flops_job = flops_local * (world if args.scaling == "weak" or world == 1 else world)
    value = flops_job * args.steps / wall_max / 1e12

    result = {
        "metric": "fwd attention TFLOP/s & % FP16 MFMA peak, D=128 L=512..16384, 1/2/4/8 GPUs",
        "value": value,
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": (value / PUBLISHED_TFLOPS[L]) if (world == 1 and L in PUBLISHED_TFLOPS) else None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {
            "workload": f"B={B} H={H} L={L} D={D} fp16 in / fp32 accumulate, non-causal forward (README shape)",
            "per_gpu_heads": bh_local,
            "global_heads": bh_local * world if args.scaling == "weak" else BH_total,
            "parallelism": f"BxH x{world}" if world > 1 else "single GPU",
            "flops_formula": "4*B*H*L^2*D",
            "preheat_s": args.preheat_seconds,
            "vs_baseline_source": "reference README.md:27-32, its own kernel on an RTX 3090",
        },
        "pct_of_fp16_peak": 100.0 * (value / world) / PEAK_TFLOPS,
        "roofline": roofline(fa, L, n_local, flops_local, ms_launch),
    }

    if rank == 0:
        info = fa.device_info()
        result["device"] = {"compute_units": info["compute_units"], "clock_mhz": info["clock_khz"] / 1e3,
                            "peak_from_device_tflops": info["fp16_dense_peak_flops"] / 1e12}

    if rank == 0 and world == 1 and not args.no_sweep:
        sweep = []
        for Bs, Ls in README_SHAPES:
            n = Bs * H * Ls * D
            qs, ks, vs = q.reshape(-1)[:n].view(Bs, H, Ls, D), k.reshape(-1)[:n].view(Bs, H, Ls, D), v.reshape(-1)[:n].view(Bs, H, Ls, D)
            os_ = o.reshape(-1)[:n].view(Bs, H, Ls, D)
            reps = max(5, min(100, int(20.0 * 512 / Ls)))
            _, ms = time_steps(fa, torch, qs, ks, vs, os_, reps, 3, lambda: None, min(args.preheat_seconds, 0.5))
            tf = 4.0 * Bs * H * Ls * Ls * D / (ms * 1e-3) / 1e12
            sweep.append({"B": Bs, "H": H, "L": Ls, "ms": ms, "tflops": tf, "pct_peak": 100 * tf / PEAK_TFLOPS})
        result["sweep"] = sweep
        # the same shapes with the causal mask of the reference's CPU baseline (testing.cu:74-77) through
        # fa_forward_general; FLOPs of the visible triangle, diagonal included
        causal = []
        for Bs, Ls in README_SHAPES:
            n = Bs * H * Ls * D
            qs, ks, vs, os_ = [t.reshape(-1)[:n].view(Bs, H, Ls, D) for t in (q, k, v, o)]
            reps = max(5, min(100, int(20.0 * 512 / Ls)))
            _, ms = time_steps(fa, torch, qs, ks, vs, os_, reps, 3, lambda: None, min(args.preheat_seconds, 0.5),
                               call=lambda a, b, c, out: fa.attention_general(a, b, c, causal=True, out=out))
            tf = 2.0 * Bs * H * Ls * (Ls + 1) * D / (ms * 1e-3) / 1e12
            causal.append({"B": Bs, "H": H, "L": Ls, "ms": ms, "tflops_visible": tf,
                           "kernel": fa.lib().fa_kernel_name_general(Ls, Ls, D, fa.FLAG_CAUSAL).decode()})
        result["causal_sweep"] = causal

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(B, H, L, args.cpu_seconds)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
