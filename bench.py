#!/usr/bin/env python3
"""Throughput benchmark of the forward-attention hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--seq-len L] [--scaling weak|strong] [--ring]

A step is one forward pass of the hot path over one batch of synthetic input, called through the C ABI
(device-pointer entry point) with inputs already in HBM.  The default workload is README row 1 of the
reference (B=32,H=16,L=512,D=128; flash_attention_tensor_cores/benchmark_tensor_flash_attention.cu:49-51),
filled with that driver's input pattern (:34-41, seeds 1234/2345/3456).  FLOPs = 4*B*H*L^2*D (:218).

Ranks.  One process per GPU.  Started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) this process is one rank.  Started plainly with --gpus N > 1 it is the launcher: it starts N
fresh rank processes itself, before anything here has touched a GPU, relays rank 0's JSON line and exits
non-zero if the ranks cannot be started or do not finish.

Modes.
  (default)         every rank runs the same README shape on its own GPU ("weak"): the path shards over
                    independent (batch, head) pairs, so there is no data-path collective; the only
                    collectives are the barrier and the max-over-ranks of the timing.  With more than one
                    rank the line also carries `strong_c4`: B=1, H=16, L=16384 with its heads split over
                    the ranks, timed right after (--no-strong-c4 skips it).
  --scaling strong  the B*H heads of ONE shape are split across the ranks (with --seq-len 16384: B=1, H=16
                    over 1/2/4/8 GPUs); still no data-path collective.
  --ring            sequence-parallel ring (default --seq-len 131072, B=1, H=16): rank r owns L/N rows of
                    Q, K, V, O; N steps fold the visiting K/V block while RCCL send/recv passes it on.  The
                    line also carries the ring's compute-only and exchange-only times per step.

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      roofline of the attention kernel from HIP events over the timed region
  cpu_baseline  the CPU oracle (a port of the reference's attention_cpu) timed on this host (N = 1 only)
  sweep         the README shapes on this GPU (back-to-back launches between two events)
  causal_sweep  the same shapes with the causal mask (fa_forward_general), TFLOP/s of the visible triangle
  normal_data_tflops, reference_protocol_tflops
                the timed shape on N(0,1) inputs (compare_impls.py:84-87), and with the reference driver's
                own protocol (warm-up 3, five launches, one event pair each, no preheat; :148-215)
  cold_inputs_tflops
                the timed shape with its four tensors rotated through four independent sets, so that no read is
                served from the 256 MiB last-level cache (the timed steps, like the reference driver's, launch
                the same 4 x 64 MiB every time)
  clock_ghz_effective, clock_ghz_effective_normal_data
                the shader clock under the kernel's own load on either data set (in-kernel s_memtime /
                s_memrealtime stamps of the library's diagnostic twin): the two data sets run the same
                cycles at different clocks
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
METRIC = "fwd attention TFLOP/s & % FP16 MFMA peak, D=128 L=512..16384, 1/2/4/8 GPUs"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200; 5 with --ring)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 20; 2 with --ring)")
    ap.add_argument("--seq-len", type=int, default=None,
                    help="L of the shape to time (default 512, B = 16384 / L; 131072 with --ring)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--ring", action="store_true", help="sequence-parallel ring over the ranks (B=1, H=16)")
    ap.add_argument("--reserved-cus", type=int, default=None,
                    help="compute units the attention kernels leave free (default: 0; with --ring on more than one "
                         "rank 8, for the exchange's transport kernels)")
    ap.add_argument("--preheat-seconds", type=float, default=1.5,
                    help="untimed back-to-back launches before the warm-up steps, so that the chip's clock governor "
                         "has settled (0 disables)")
    ap.add_argument("--data", choices=["pattern", "normal"], default="pattern",
                    help="inputs of the timed steps: the reference driver's pattern (default; what `value` is quoted "
                         "on) or N(0,1) rounded to fp16 (compare_impls.py:84-87; for profiles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip normal_data_tflops / reference_protocol_tflops")
    ap.add_argument("--no-strong-c4", action="store_true",
                    help="multi-rank default mode: skip the strong-scaling point (B=1, H=16, L=16384 split by heads)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="launcher: seconds the ranks may take")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="plumbing test without a GPU: gloo ranks on the CPU time a stand-in step that sleeps; no "
                         "kernel runs, and the line says so in `data`")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 5 if args.ring else 200
    if args.warmup is None:
        args.warmup = 2 if args.ring else 20
    if args.seq_len is None:
        args.seq_len = 131072 if args.ring else 512
    if args.ring:
        args.scaling = "strong"
    return args


# ----------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a rank environment
# ----------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def count_gpus(sysfs_root="/sys/class/kfd/kfd/topology/nodes", env=None):
    """GPUs of this node without loading a GPU runtime: the kernel driver's topology lists every agent as a node,
    and a GPU is a node with SIMDs (`simd_count` > 0 in its `properties`; CPUs have 0).  A visibility list in the
    environment (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES) caps the count the way the
    runtime would."""
    env = os.environ if env is None else env
    n = 0
    try:
        nodes = sorted(os.listdir(sysfs_root))
    except OSError:
        nodes = []
    for node in nodes:
        try:
            with open(os.path.join(sysfs_root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue  # (a node this user may not read is not a GPU this user can run on)
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if env.get(var, "").strip():
            n = min(n, len([x for x in env[var].split(",") if x.strip()]))
    return n


def launch_ranks(args, argv):
    """Start args.gpus rank processes of this script and relay rank 0's JSON line.  Nothing here loads a GPU
    runtime (the GPUs are counted from sysfs, torch is not imported), and the ranks are fresh children (never a
    re-exec of this process)."""
    n = args.gpus
    if not args.selftest_cpu:
        have = count_gpus()
        if have < n:
            print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr)
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    deadline = time.monotonic() + args.launch_timeout
    out0, failed = "", None
    try:
        out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.monotonic()))
        for r, p in enumerate(procs):
            rc = p.wait(timeout=max(1.0, deadline - time.monotonic()))
            if rc != 0 and failed is None:
                failed = (r, rc)
    except subprocess.TimeoutExpired:
        failed = ("timeout", 124)
    finally:
        for p in procs:  # exactly the children started above
            if p.poll() is None:
                p.kill()
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if failed is not None or len(lines) != 1:
        sys.stderr.write(out0)
        print(f"bench.py: ranks failed ({failed}) or rank 0 printed {len(lines)} JSON lines", file=sys.stderr)
        return 1
    if json.loads(lines[0]).get("n_gpus") != n:
        print(f"bench.py: rank 0 reports n_gpus != {n}", file=sys.stderr)
        return 1
    print(lines[0])
    return 0


# ----------------------------------------------------------------------------------------------------------
# timing helpers
# ----------------------------------------------------------------------------------------------------------
def preheat(torch, call, seconds):
    """The chip raises its clock only under sustained load: measured on L=512, launches that follow
    a few milliseconds of work run at about 1.95 GHz (75 us per launch), launches that follow a second
    of work at about 2.25 GHz (66 us).  Steady-state throughput is what the metric means, so the timed
    region is preceded by this much untimed work, on top of the --warmup steps."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            call()
        torch.cuda.synchronize()


def time_steps(torch, call, steps, warmup, barrier, preheat_s=0.0):
    """(seconds for all steps by the host clock, ms per step by events on the launch stream)."""
    preheat(torch, call, preheat_s)
    for _ in range(warmup):
        call()
    torch.cuda.synchronize()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        call()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier()
    return wall, ev0.elapsed_time(ev1) / steps


def reference_protocol_ms(torch, call, warmup=3, reps=5):
    """benchmark_tensor_flash_attention.cu:148-215: warm-up 3, then 5 launches each between its own event pair and
    waited for; the mean.  No preheat."""
    for _ in range(warmup):
        call()
    torch.cuda.synchronize()
    total = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        e1.synchronize()
        total += e0.elapsed_time(e1)
    return total / reps


def roofline(fa, L, bh_local, ms_launch):
    """The roof that binds the attention kernel for this shape.  Algorithmic work per launch: 4*L^2*D
    FLOP and 4*L*D*2 bytes per (batch, head) pair (Q, K, V read once, O written once), i.e. L/2 FLOP
    per byte.  The machine balance is 2500 TFLOP/s / 8 TB/s = 312 FLOP/B: L=512 (256 FLOP/B) lies
    below the ridge, so its floor is the HBM one (33.6 us against 27.5 us on the matrix pipe); from
    L=1024 up the matrix pipe binds.  Both fractions are reported either way."""
    flops = 4.0 * bh_local * L * L * D
    algo_bytes = 4 * bh_local * L * D * 2
    sec = ms_launch * 1e-3
    tflops = flops / sec / 1e12
    gbps = algo_bytes / sec / 1e9
    hbm_bound = algo_bytes / (PEAK_HBM_GBPS * 1e9) > flops / (PEAK_TFLOPS * 1e12)
    ways = fa.lib().fa_split_ways(bh_local, L, L)
    traffic, src = measured_traffic(L) if bh_local * L == 16384 * H_DEFAULT else (None, None)
    return {
        "bound": "hbm" if hbm_bound else "mfma",
        "achieved": gbps if hbm_bound else tflops,
        "peak": PEAK_HBM_GBPS if hbm_bound else PEAK_TFLOPS,
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": (gbps / PEAK_HBM_GBPS) if hbm_bound else (tflops / PEAK_TFLOPS),
        "traffic": traffic,
        "traffic_source": (f"{src} (committed rocprofv3 summary of this shape: separate FETCH_SIZE and WRITE_SIZE "
                           "passes, not measured in this run)") if traffic is not None else None,
        "kernel": fa.lib().fa_kernel_name(L, L).decode() if ways <= 1 else "fa::fa_fwd_w64_split_kernel",
        "key_split_ways": ways,
        "kernel_ms": ms_launch,
        "algorithmic_bytes": algo_bytes,
        "algorithmic_flops": flops,
        "mfma_tflops": tflops,
        "mfma_frac": tflops / PEAK_TFLOPS,
        "hbm_GBps": gbps,
        "hbm_frac": gbps / PEAK_HBM_GBPS,
    }


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
    # when even the whole workload is short on this host, repeat it until ~target_s has been spent (the probe
    # above also paid for starting the threads, so one pass of the sample is timed before deciding)
    t = time.perf_counter()
    oracle.attention_cpu(Q, K, V, nthreads=0)
    t_pass = max(time.perf_counter() - t, 1e-3)
    reps = int(max(1, min(200, target_s / t_pass)))
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
    WRITE_SIZE passes) and that file's name; (None, None) when this shape has not been profiled."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_L{L}.json")))
    if not files:
        return None, None
    try:
        return float(json.load(open(files[-1]))["hbm_bytes"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def clock_probe(fa, torch, q, k, v, o, seconds=0.4):
    """Effective shader clock under THIS kernel's load, in GHz: the library's diagnostic twin
    (libflash_attention_stamps.so: the default kernel alone, every workgroup recording s_memtime, shader cycles, and
    s_memrealtime, 100 MHz, at entry and exit) runs the same problem back to back for `seconds`, and the clock is
    cycles / time of the last launch's workgroups (median).  None when the twin is not built or does not serve
    the shape."""
    import ctypes as ct

    import numpy as np

    path = os.path.join(ROOT, "flash-attention_amd", "csrc", "libflash_attention_stamps.so")
    B, H, L, Dh = q.shape
    if not os.path.exists(path) or L % 64 or L < 320 or Dh != D:
        return None
    twin = ct.CDLL(path)
    twin.fa_forward.argtypes = [fa.HalfPtr] * 4 + [ct.c_int] * 5 + [ct.c_void_p, ct.POINTER(ct.c_float)]
    twin.fa_forward.restype = ct.c_int
    twin.fa_debug_set_stamps.argtypes = [ct.c_void_p]
    ptr = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), fa.HalfPtr)
    stream = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: twin.fa_forward(ptr(q), ptr(k), ptr(v), ptr(o), B, H, L, Dh, 64, stream, None)
    stamps = torch.zeros(4 * 1024, dtype=torch.int64, device=q.device)
    twin.fa_debug_set_stamps(None)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            if call() != 0:
                return None
        torch.cuda.synchronize()
    twin.fa_debug_set_stamps(ct.c_void_p(stamps.data_ptr()))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    twin.fa_debug_set_stamps(None)
    s = stamps.cpu().numpy().reshape(-1, 4)
    s = s[s[:, 3] > 0]
    if len(s) == 0:
        return None
    ghz = (s[:, 2] - s[:, 0]) / np.maximum(s[:, 3] - s[:, 1], 1) * 0.1
    return {"ghz": float(np.median(ghz)), "cycles_per_workgroup": float(np.median(s[:, 2] - s[:, 0])),
            "workgroups": int(len(s))}


# ----------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------
class Ranks:
    """The process group of this run (or none): barrier, max and gather over ranks."""

    def __init__(self, torch, backend, world, rank, dev):
        self.torch, self.world, self.rank, self.dev, self.dist = torch, world, rank, dev, None
        # (FA_BENCH_FORCE_DIST=1 takes the multi-rank code path with a single rank: RCCL init, barrier and
        # the max-over-ranks reduction can then be exercised on a one-GPU box)
        if world > 1 or os.environ.get("FA_BENCH_FORCE_DIST") == "1":
            import torch.distributed as dist

            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group("gloo")
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        if self.dist is None:
            return [float(x)]
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


class JsonLine:
    """Keeps the process's stdout for the one JSON line.  Libraries under this script write to file descriptor 1
    themselves (RCCL prints a version banner there when the first communicator is set up), so everything else
    that goes to descriptor 1 while the ranks run is sent to stderr."""

    def __init__(self):
        sys.stdout.flush()
        self.fd = os.dup(1)
        os.dup2(2, 1)

    def emit(self, result):
        sys.stdout.flush()
        os.write(self.fd, (json.dumps(result) + "\n").encode())


def base_result(args, world, value, wall_max):
    return {
        "metric": METRIC,
        "value": value,
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
    }


C4_L, C4_HEADS = 16384, 16  # the long-sequence shape of the metric's multi-GPU curve: B=1, H=16, L=16384


def strong_c4(args, fa, torch, ranks, dev):
    """The strong-scaling point of the metric's multi-GPU curve beside the default (weak) line: the 16 heads of
    B=1, H=16, L=16384 split over the ranks (independent (batch, head) pairs: tensor_flash_attention.cu:30-31), no
    data-path collective.  Also times all 16 heads on this rank's GPU alone: efficiency_vs_N1_rate = the job's
    rate per GPU over that one-GPU rate, both measured in this run."""
    world, rank = ranks.world, ranks.rank
    if C4_HEADS % world:
        return {"error": f"{C4_HEADS} heads do not divide over {world} ranks"}
    L, bh_local = C4_L, C4_HEADS // world
    full = [torch.empty((C4_HEADS, L, D), dtype=torch.float16, device=dev) for _ in range(3)]
    for t, seed in zip(full, (1234, 2345, 3456)):
        fa.init_half(t, seed)
    q, k, v = [t[rank * bh_local:(rank + 1) * bh_local].reshape(1, bh_local, L, D) for t in full]
    o = torch.empty_like(q)
    steps = max(3, min(20, args.steps))
    wall, ms = time_steps(torch, lambda: fa.attention(q, k, v, out=o), steps, 2, ranks.barrier,
                          min(args.preheat_seconds, 0.5))
    wall_max = ranks.max(wall)
    per_rank_ms = ranks.gather(ms)
    flops_job = 4.0 * C4_HEADS * float(L) * float(L) * D
    tflops_total = flops_job * steps / wall_max / 1e12
    # the one-GPU rate on this run's own hardware: all 16 heads on every rank at the same time
    qa, ka, va = [t.reshape(1, C4_HEADS, L, D) for t in full]
    oa = torch.empty_like(qa)
    _, ms_all = time_steps(torch, lambda: fa.attention(qa, ka, va, out=oa), max(3, steps // 2), 1, ranks.barrier,
                           min(args.preheat_seconds, 0.3))
    rate_n1 = flops_job / (ranks.max(ms_all) * 1e-3) / 1e12
    return {
        "workload": f"B=1 H={C4_HEADS} L={L} D={D}, heads split over the ranks",
        "per_gpu_heads": bh_local,
        "steps": steps,
        "ms_max_over_ranks": wall_max / steps * 1e3,
        "per_rank_kernel_ms": per_rank_ms,
        "tflops_total": tflops_total,
        "one_gpu_16_heads_tflops": rate_n1,
        "efficiency_vs_N1_rate": tflops_total / world / rate_n1,
        "key_split_ways": int(fa.lib().fa_split_ways(bh_local, L, L)),
    }


def run_selftest(args, world, rank, line):
    """No GPU, no kernel: the launcher, the process group and the line's assembly with a stand-in step."""
    import torch

    ranks = Ranks(torch, "gloo", world, rank, torch.device("cpu"))
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    wall_max = ranks.max(time.perf_counter() - t0)
    per_rank = ranks.gather(1.0 + rank)
    result = base_result(args, world, 0.0, wall_max)
    result.update({"data": "selftest: no kernel ran (stand-in step sleeps 1 ms)", "value": None,
                   "config": {"workload": "none (launcher / process-group plumbing test)"},
                   "per_rank_kernel_ms": per_rank})
    if world > 1 and args.scaling == "weak" and not args.no_strong_c4:
        # the keys of the strong-scaling point, with the stand-in step in the kernel's place
        t0 = time.perf_counter()
        time.sleep(0.001)
        w = ranks.max(time.perf_counter() - t0)
        result["strong_c4"] = {"workload": "none (stand-in)", "per_gpu_heads": C4_HEADS // world if C4_HEADS % world == 0 else None,
                               "steps": 1, "ms_max_over_ranks": w * 1e3, "per_rank_kernel_ms": ranks.gather(1.0 + rank),
                               "tflops_total": None, "one_gpu_16_heads_tflops": None, "efficiency_vs_N1_rate": None,
                               "key_split_ways": None}
    ranks.close()
    if rank == 0:
        line.emit(result)


def run_heads(args, fa, torch, ranks, dev):
    """Default and --scaling strong: the README shape, whole on every rank or its heads split over the ranks."""
    world, rank = ranks.world, ranks.rank
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
    full = [torch.empty((BH_total, L, D), dtype=torch.float16, device=dev) for _ in range(3)]
    for t, seed in zip(full, (1234, 2345, 3456)):
        if args.data == "pattern":
            fa.init_half(t, seed)
        else:
            t.copy_(torch.randn(t.shape, generator=torch.Generator(device=dev).manual_seed(seed), device=dev,
                                dtype=torch.float32).to(torch.float16))
    q, k, v = [t[head0:head0 + bh_local].reshape(1, bh_local, L, D) for t in full]  # contiguous slabs
    o = torch.empty_like(q)
    torch.cuda.synchronize()
    call = lambda: fa.attention(q, k, v, out=o)

    # The first collective sets the communicator up, which takes long enough for the chip to fall back to
    # its idle clock: it must not sit between the preheat and the timed steps.
    ranks.barrier()
    wall, ms_launch = time_steps(torch, call, args.steps, args.warmup, ranks.barrier, args.preheat_seconds)
    wall_max = ranks.max(wall)
    per_rank_ms = ranks.gather(ms_launch)

    flops_local = 4.0 * bh_local * L * L * D
    value = flops_local * world * args.steps / wall_max / 1e12

    result = base_result(args, world, value, wall_max)
    result["vs_baseline"] = (value / PUBLISHED_TFLOPS[L]) if (world == 1 and L in PUBLISHED_TFLOPS) else None
    if args.data == "normal":
        result["data"] = "synthetic, N(0,1) rounded to fp16"
    result["config"] = {
        "workload": f"B={B} H={H} L={L} D={D} fp16 in / fp32 accumulate, non-causal forward (README shape)",
        "per_gpu_heads": bh_local,
        "global_heads": bh_local * world,
        "parallelism": f"BxH x{world}" if world > 1 else "single GPU",
        "flops_formula": "4*B*H*L^2*D",
        "preheat_s": args.preheat_seconds,
        "vs_baseline_source": "reference README.md:27-32, its own kernel on an RTX 3090",
    }
    result["pct_of_fp16_peak"] = 100.0 * (value / world) / PEAK_TFLOPS
    result["per_rank_kernel_ms"] = per_rank_ms
    result["roofline"] = roofline(fa, L, bh_local, ms_launch)

    if rank == 0:
        info = fa.device_info()
        result["device"] = {"compute_units": info["compute_units"], "clock_mhz": info["clock_khz"] / 1e3,
                            "peak_from_device_tflops": info["fp16_dense_peak_flops"] / 1e12}

    if rank == 0 and world == 1 and not args.no_extras:
        # the same shape with the reference driver's own timing protocol, and on N(0,1) inputs
        ms_ref = reference_protocol_ms(torch, call)
        result["reference_protocol_tflops"] = flops_local / (ms_ref * 1e-3) / 1e12
        g = torch.Generator(device=dev).manual_seed(0)
        qn, kn, vn = [torch.randn(q.shape, generator=g, device=dev, dtype=torch.float32).to(torch.float16) for _ in range(3)]
        _, ms_n = time_steps(torch, lambda: fa.attention(qn, kn, vn, out=o), max(5, args.steps // 2), 3, lambda: None,
                             min(args.preheat_seconds, 1.0))
        result["normal_data_tflops"] = flops_local / (ms_n * 1e-3) / 1e12
        # the effective shader clock under the kernel's load, per data set (same cycle count, different clock)
        cn = clock_probe(fa, torch, qn, kn, vn, o)
        cp = clock_probe(fa, torch, q, k, v, o)
        result["clock_ghz_effective"] = cp["ghz"] if cp else None
        result["clock_ghz_effective_normal_data"] = cn["ghz"] if cn else None
        result["cycles_per_workgroup"] = cp["cycles_per_workgroup"] if cp else None
        result["clock_source"] = ("s_memtime / s_memrealtime stamps of every workgroup of the default kernel's diagnostic "
                                  "twin (libflash_attention_stamps.so) under back-to-back launches of this shape"
                                  if cp else "diagnostic twin not built or shape not served")
        del qn, kn, vn
        # The timed steps launch the same four tensors again and again (as the reference driver does,
        # benchmark_tensor_flash_attention.cu:148-215); at 4 x 64 MiB they are as large as the chip's last-level
        # cache, which then serves part of the input reads.  The same launches with Q, K, V and O rotated through
        # four independent sets (1 GiB touched between two uses of a line: every read comes from HBM):
        try:
            sets = [(q, k, v, o)]
            for s_ in range(1, 4):
                extra = [torch.empty_like(q) for _ in range(4)]
                for t, seed in zip(extra[:3], (1234, 2345, 3456)):
                    fa.init_half(t, seed + 7 * s_)
                sets.append(tuple(extra))
            turn = [0]

            def rotating():
                qq, kk, vv, oo = sets[turn[0] & 3]
                turn[0] += 1
                fa.attention(qq, kk, vv, out=oo)

            _, ms_c = time_steps(torch, rotating, max(8, args.steps // 2), 4, lambda: None,
                                 min(args.preheat_seconds, 1.0))
            result["cold_inputs_tflops"] = flops_local / (ms_c * 1e-3) / 1e12
            del sets
        except RuntimeError as e:  # not enough free memory for the extra sets
            result["cold_inputs_tflops"] = None
            result["cold_inputs_note"] = f"{type(e).__name__}: {str(e)[:120]}"

    if rank == 0 and world == 1 and not args.no_sweep:
        n_all = 16384 * H * D
        sq, sk, sv = [torch.empty(n_all, dtype=torch.float16, device=dev) for _ in range(3)]
        for t, seed in zip((sq, sk, sv), (1234, 2345, 3456)):
            fa.init_half(t, seed)
        so = torch.empty_like(sq)
        sweep, causal = [], []
        for Bs, Ls in README_SHAPES:
            qs, ks, vs, os_ = [t.view(Bs, H, Ls, D) for t in (sq, sk, sv, so)]
            reps = max(5, min(100, int(20.0 * 512 / Ls)))
            _, ms = time_steps(torch, lambda: fa.attention(qs, ks, vs, out=os_), reps, 3, lambda: None,
                               min(args.preheat_seconds, 0.5))
            tf = 4.0 * Bs * H * Ls * Ls * D / (ms * 1e-3) / 1e12
            sweep.append({"B": Bs, "H": H, "L": Ls, "ms": ms, "tflops": tf, "pct_peak": 100 * tf / PEAK_TFLOPS})
            # the same shape with the causal mask of the reference's CPU baseline (testing.cu:74-77) through
            # fa_forward_general; FLOPs of the visible triangle, diagonal included
            _, ms = time_steps(torch, lambda: fa.attention_general(qs, ks, vs, causal=True, out=os_), reps, 3,
                               lambda: None, min(args.preheat_seconds, 0.5))
            tf = 2.0 * Bs * H * Ls * (Ls + 1) * D / (ms * 1e-3) / 1e12
            causal.append({"B": Bs, "H": H, "L": Ls, "ms": ms, "tflops_visible": tf,
                           "kernel": fa.lib().fa_kernel_name_general(Ls, Ls, D, fa.FLAG_CAUSAL).decode()})
        result["sweep"] = sweep
        result["causal_sweep"] = causal

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(B, H, L, args.cpu_seconds)

    # More than one rank and no mode flag: the same line also carries the strong-scaling point of the metric's
    # curve.  (All ranks take part; a failure is recorded in the line instead of losing it.)
    if ranks.dist is not None and args.scaling == "weak" and not args.no_strong_c4:
        try:
            result["strong_c4"] = strong_c4(args, fa, torch, ranks, dev)
        except Exception as e:  # noqa: BLE001
            result["strong_c4"] = {"error": f"{type(e).__name__}: {e}"}
    return result


def run_ring(args, fa, torch, ranks, dev):
    """--ring: B=1, H=16, L rows split over the ranks; multi_gpu.ring_forward is the step.  Beside the whole step
    (compute with the exchange running behind it) the compute alone and the exchange alone are timed."""
    from flash_attention_amd import multi_gpu as mg

    world, rank = ranks.world, ranks.rank
    L, H = args.seq_len, H_DEFAULT
    if L % (64 * world) or L <= 0:
        raise SystemExit("--ring needs --seq-len divisible by 64 * (number of GPUs)")
    Lb = L // world
    reserved = args.reserved_cus if args.reserved_cus is not None else (8 if world > 1 else 0)
    fa.set_reserved_cus(reserved)
    q, k, v = [torch.empty((1, H, Lb, D), dtype=torch.float16, device=dev) for _ in range(3)]
    for t, seed in zip((q, k, v), (1234, 2345, 3456)):
        fa.init_half(t, seed + 7919 * rank)
    torch.cuda.synchronize()
    ranks.barrier()

    call = lambda: mg.ring_forward(q, k, v)
    # a step is tens of milliseconds: the warm-up steps are the preheat
    wall, ms_step = time_steps(torch, call, args.steps, args.warmup, ranks.barrier, 0.0)
    wall_max = ranks.max(wall)
    per_rank_ms = ranks.gather(ms_step)

    def compute_only():  # the rank's kernels of one ring pass, every block read from the local K/V
        st = fa.BlockState(H, Lb, D, dev)
        for s in range(world):
            fa.attention_block_general(q, k, v, st, last=(s == world - 1))

    _, ms_compute = time_steps(torch, compute_only, max(2, args.steps // 2), 1, ranks.barrier, 0.0)
    ms_exchange = None
    if world > 1:
        dist = ranks.dist
        cur, spare = torch.stack((k, v)).contiguous(), torch.empty((2,) + tuple(k.shape), dtype=k.dtype, device=dev)

        def exchange_only():  # the ring pass's world - 1 hand-offs and nothing else
            a, b = cur, spare
            for _ in range(world - 1):
                reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, a, (rank + 1) % world),
                                               dist.P2POp(dist.irecv, b, (rank - 1) % world)])
                for r in reqs:
                    r.wait()
                a, b = b, a

        _, ms_x = time_steps(torch, exchange_only, max(2, args.steps // 2), 1, ranks.barrier, 0.0)
        ms_exchange = ms_x / (world - 1)

    flops_job = 4.0 * H * float(L) * float(L) * D
    value = flops_job * args.steps / wall_max / 1e12
    result = base_result(args, world, value, wall_max)
    result["config"] = {
        "workload": f"B=1 H={H} L={L} D={D} fp16 in / fp32 accumulate, non-causal forward, sequence-parallel ring",
        "rows_per_gpu": Lb,
        "parallelism": f"ring x{world}" if world > 1 else "single GPU (one block)",
        "flops_formula": "4*B*H*L^2*D",
        "reserved_cus": reserved,
        "exchange_bytes_per_step_per_rank": 2 * H * Lb * D * 2 if world > 1 else 0,
    }
    result["pct_of_fp16_peak"] = 100.0 * (value / world) / PEAK_TFLOPS
    result["per_rank_kernel_ms"] = per_rank_ms
    result["ring"] = {
        "steps_per_pass": world,
        "pass_ms_max_over_ranks": wall_max / args.steps * 1e3,
        "compute_only_ms_per_ring_step": ranks.max(ms_compute) / world,
        "exchange_only_ms_per_ring_step": ranks.max(ms_exchange) if ms_exchange is not None else None,
        "whole_ms_per_ring_step": wall_max / args.steps * 1e3 / world,
    }
    flops_local = flops_job / world
    tflops = flops_local / (ranks.max(ms_compute) * 1e-3) / 1e12
    result["roofline"] = {
        "bound": "mfma", "achieved": tflops, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_TFLOPS,
        "traffic": None, "kernel": fa.lib().fa_kernel_name_block_general(Lb, Lb, D, 0, 0, 0).decode(),
        "kernel_ms": ranks.max(ms_compute) / world, "algorithmic_flops": flops_local / world,
        "algorithmic_bytes": (2 * H * Lb * D * 2) + (H * Lb * D * 2 if world == 1 else 2 * H * Lb * D * 4),
    }
    return result


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, argv))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    line = JsonLine()
    if args.selftest_cpu:
        return run_selftest(args, world, rank, line)

    import torch

    import __graft_entry__ as ge

    fa = ge.load_package()
    fa.lib()  # fail loudly if the HIP library is missing
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = Ranks(torch, "nccl", world, rank, dev)
    if args.reserved_cus is not None:
        fa.set_reserved_cus(args.reserved_cus)
    result = run_ring(args, fa, torch, ranks, dev) if args.ring else run_heads(args, fa, torch, ranks, dev)
    ranks.close()
    if rank == 0:
        line.emit(result)


if __name__ == "__main__":
    main()
