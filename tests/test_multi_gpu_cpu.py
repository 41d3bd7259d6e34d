"""CPU tier for the N>1 paths: world_size-2 (and 4) gloo runs of the head-sharded and the
ring drivers, with the GPU compute step replaced by the CPU oracle so that only the
partitioning and the communication schedule are under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, normal_fp16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _load():
    sys.path.insert(0, ROOT)
    import importlib

    import __graft_entry__ as ge
    import oracle

    ge.load_package()
    return importlib.import_module("flash_attention_amd.multi_gpu"), oracle


class _CpuState:
    def __init__(self, BH, Lq, D, device):
        self.inner = None


def _make_step(oracle, B, H):
    def step(q, kb, vb, state, last):
        Q, K, V = [np.ascontiguousarray(t.numpy()) for t in (q, kb, vb)]
        if state.inner is None:
            state.inner = oracle.BlockState(B, H, Q.shape[2], Q.shape[3])
        oracle.block_update(state.inner, Q, K, V, nthreads=1)
        return torch.from_numpy(oracle.block_finalize(state.inner)) if last else None

    return step


def _ring_worker(rank, world, port, out_dir):
    mg, oracle = _load()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        B, H, L, D = 1, 2, 64 * 2 * world, 128
        Q, K, V = normal_fp16(100, (B, H, L, D))
        n = L // world
        sl = slice(rank * n, (rank + 1) * n)
        q, k, v = [torch.from_numpy(np.ascontiguousarray(a[:, :, sl])) for a in (Q, K, V)]
        out = mg.ring_forward(q, k, v, step_fn=_make_step(oracle, B, H), state_factory=_CpuState)
        ref = oracle.attention_cpu(Q, K, V, nthreads=1)[:, :, sl]
        m = oracle.error_metrics(ref, out.numpy())
        ok = m["max_abs"] < 3e-4 and m["global_rel"] < 3e-4
        # inputs must come back untouched (the ring only reads them)
        ok = ok and np.array_equal(k.numpy(), K[:, :, sl]) and np.array_equal(v.numpy(), V[:, :, sl])
        open(os.path.join(out_dir, f"ring_{rank}.txt"), "w").write(f"{int(ok)} {m['max_abs']:.3e} {m['global_rel']:.3e}")
    finally:
        dist.destroy_process_group()


def _uneven_ring_worker(rank, world, port, out_dir):
    """Shards of different lengths (a sequence that does not divide by the ranks): rows_per_rank."""
    mg, oracle = _load()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        rows = [100, 37, 192][:world]
        B, H, L, D = 1, 2, sum(rows), 128
        Q, K, V = normal_fp16(103, (B, H, L, D))
        a = sum(rows[:rank])
        sl = slice(a, a + rows[rank])
        q, k, v = [torch.from_numpy(np.ascontiguousarray(t[:, :, sl])) for t in (Q, K, V)]
        out = mg.ring_forward(q, k, v, step_fn=_make_step(oracle, B, H), state_factory=_CpuState, rows_per_rank=rows)
        st = oracle.BlockState(B, H, rows[rank], D)
        oracle.block_update(st, np.ascontiguousarray(Q[:, :, sl]), K, V, nthreads=1)
        m = oracle.error_metrics(oracle.block_finalize(st), out.numpy())
        ok = m["max_abs"] < 3e-4 and m["global_rel"] < 3e-4 and tuple(out.shape) == (B, H, rows[rank], D)
        bad_list = False
        try:
            mg.ring_forward(q, k, v, step_fn=_make_step(oracle, B, H), state_factory=_CpuState, rows_per_rank=rows[:-1])
        except ValueError:
            bad_list = True
        open(os.path.join(out_dir, f"uring_{rank}.txt"), "w").write(f"{int(ok and bad_list)} {m['max_abs']:.3e} {m['global_rel']:.3e}")
    finally:
        dist.destroy_process_group()


def _shard_worker(rank, world, port, out_dir):
    mg, oracle = _load()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        B, H, L, D = 3, 3, 128, 128   # 9 heads over 2 ranks: uneven split
        Q, K, V = normal_fp16(101, (B, H, L, D))
        attn = lambda q, k, v: torch.from_numpy(oracle.attention_cpu(*[np.ascontiguousarray(t.numpy()) for t in (q, k, v)], nthreads=1))
        start, stop, o = mg.sharded_forward(*[torch.from_numpy(a) for a in (Q, K, V)], rank, world, attention_fn=attn)
        ref = oracle.attention_cpu(Q, K, V, nthreads=1).reshape(1, B * H, L, D)[:, start:stop]
        ok = np.array_equal(ref.view(np.uint32), o.numpy().view(np.uint32))
        # the timing reduction bench.py uses: max over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        open(os.path.join(out_dir, f"shard_{rank}.txt"), "w").write(f"{int(ok)} {start} {stop}")
    finally:
        dist.destroy_process_group()


def _make_causal_step(oracle, B, H):
    def step(q, kb, vb, state, last, causal, q_pos0, k_pos0):
        Q, K, V = [np.ascontiguousarray(t.numpy()) for t in (q, kb, vb)]
        if state.inner is None:
            state.inner = oracle.BlockState(B, H, Q.shape[2], Q.shape[3])
        oracle.block_update(state.inner, Q, K, V, causal=causal, q_pos0=q_pos0, k_pos0=k_pos0, nthreads=1)
        return torch.from_numpy(oracle.block_finalize(state.inner)) if last else None

    return step


def _causal_ring_worker(rank, world, port, out_dir):
    mg, oracle = _load()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        B, H, Lc, D = 1, 2, 64, 128
        L = 2 * world * Lc
        Q, K, V = normal_fp16(102, (B, H, L, D))
        lo, hi = mg.zigzag_chunks(rank, world)
        pick = lambda a: torch.from_numpy(np.stack([a[:, :, c * Lc:(c + 1) * Lc] for c in (lo, hi)]))
        out = mg.ring_forward_causal(pick(Q), pick(K), pick(V), step_fn=_make_causal_step(oracle, B, H),
                                     state_factory=_CpuState)
        ref = oracle.attention_cpu(Q, K, V, causal=True, nthreads=1)
        ref2 = np.stack([ref[:, :, c * Lc:(c + 1) * Lc] for c in (lo, hi)])
        m = oracle.error_metrics(ref2, out.numpy())
        ok = m["max_abs"] < 3e-4 and m["global_rel"] < 3e-4
        open(os.path.join(out_dir, f"cring_{rank}.txt"), "w").write(f"{int(ok)} {m['max_abs']:.3e} {m['global_rel']:.3e}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_causal_zigzag_ring_gloo(tmp_path, world):
    mp.spawn(_causal_ring_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, *rest = open(tmp_path / f"cring_{r}.txt").read().split()
        assert ok == "1", (r, rest)


def test_zigzag_schedule_properties():
    """Every (query chunk, key chunk) pair at or below the diagonal is computed exactly once, by the
    owner of the query chunk, and every rank has the same amount of work in every step."""
    mg, _ = _load()
    for world in (1, 2, 3, 8):
        seen = set()
        for rank in range(world):
            for pairs in mg.zigzag_schedule(rank, world):
                work = sum(0.5 if diag else 1.0 for *_, diag in pairs)
                assert work == 2.0, (world, rank, pairs)
                for qs, ks, qc, kc, diag in pairs:
                    assert qc == mg.zigzag_chunks(rank, world)[qs] and kc <= qc and diag == (kc == qc)
                    assert (qc, kc) not in seen
                    seen.add((qc, kc))
        n = 2 * world
        assert seen == {(a, b) for a in range(n) for b in range(a + 1)}


@pytest.mark.parametrize("world", [2, 4])
def test_ring_forward_gloo(tmp_path, world):
    mp.spawn(_ring_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, *rest = open(tmp_path / f"ring_{r}.txt").read().split()
        assert ok == "1", (r, rest)


def test_ring_forward_uneven_shards_gloo(tmp_path):
    world = 3
    mp.spawn(_uneven_ring_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, *rest = open(tmp_path / f"uring_{r}.txt").read().split()
        assert ok == "1", (r, rest)


def test_head_sharded_forward_gloo(tmp_path):
    world = 2
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [open(tmp_path / f"shard_{r}.txt").read().split() for r in range(world)]
    assert [g[0] for g in got] == ["1", "1"]
    assert (got[0][1], got[0][2], got[1][1], got[1][2]) == ("0", "5", "5", "9")


def test_head_shard_partition_properties():
    mg, _ = _load()
    for total in (1, 7, 16, 512):
        for world in (1, 2, 3, 4, 8):
            ranges = [mg.head_shard(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    # C4: 16 heads over 1/2/4/8 GPUs
    assert [mg.head_shard(16, w, w - 1) for w in (1, 2, 4, 8)] == [(0, 16), (8, 16), (12, 16), (14, 16)]
    with pytest.raises(ValueError):
        mg.head_shard(4, 2, 2)


def test_ring_block_order():
    mg, _ = _load()
    assert mg.ring_block_order(0, 4) == [0, 3, 2, 1]
    assert mg.ring_block_order(2, 4) == [2, 1, 0, 3]
    for w in (2, 8):
        for r in range(w):
            assert sorted(mg.ring_block_order(r, w)) == list(range(w))


def _run_bench(*argv, env=None):
    import subprocess
    import sys

    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                          timeout=280, cwd=ROOT, env=e)


def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no rank environment starts two rank processes itself (here: gloo ranks on
    the CPU with a stand-in step, --selftest-cpu) and prints exactly one line, rank 0's, with n_gpus = 2."""
    import json

    r = _run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-cpu")
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    assert j["per_rank_kernel_ms"] == [1.0, 2.0]  # gathered in rank order
    assert "selftest" in j["data"] and j["value"] is None  # no kernel ran and the line says so
    # with more than one rank and no mode flag the line also carries the strong-scaling point's keys
    c4 = j["strong_c4"]
    for key in ("per_gpu_heads", "ms_max_over_ranks", "tflops_total", "efficiency_vs_N1_rate", "key_split_ways"):
        assert key in c4, key
    assert c4["per_gpu_heads"] == 8 and c4["per_rank_kernel_ms"] == [1.0, 2.0]


def test_bench_launcher_counts_gpus_without_a_gpu_runtime(tmp_path):
    """The launcher counts GPUs from the kernel driver's topology in sysfs (nodes with SIMDs) and imports neither
    torch nor any HIP library before it starts the ranks: a parent that had initialised HIP would then fork and
    exec children, which the GPU boxes refuse."""
    import subprocess
    import sys

    nodes = tmp_path / "nodes"
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):  # two CPU sockets, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "assert bench.count_gpus(%r, env={}) == 3\n"
            "assert bench.count_gpus(%r, env={'HIP_VISIBLE_DEVICES': '0,2'}) == 2\n"
            "assert bench.count_gpus(%r + '/missing', env={}) == 0\n"
            "import inspect; src = inspect.getsource(bench.launch_ranks)\n"
            "assert 'import torch' not in src and 'count_gpus()' in src\n"
            "assert not [m for m in sys.modules if m == 'torch' or m.startswith('torch.')], 'torch imported'\n"
            % (ROOT, str(nodes), str(nodes), str(nodes)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr


def test_bench_launcher_fails_loudly_without_the_gpus():
    """Asked for more GPUs than the node has, bench.py exits non-zero and prints no JSON line (it used to run one
    rank and report n_gpus = 1)."""
    sys.path.insert(0, ROOT)
    import bench

    n = bench.count_gpus()
    r = _run_bench("--gpus", str(n + 2), "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "GPU" in r.stderr


def test_bench_rank_rejects_a_world_that_is_not_gpus():
    r = _run_bench("--gpus", "1", "--selftest-cpu", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_stdout_carries_the_json_line_only():
    """What a library writes to file descriptor 1 while the ranks run (RCCL prints a version banner there when the
    first communicator is set up) goes to stderr; stdout holds the one JSON line."""
    import subprocess
    import sys

    code = ("import os, sys; sys.path.insert(0, %r); import bench; line = bench.JsonLine(); "
            "os.write(1, b'RCCL version : banner\\n'); print('a stray print'); line.emit({'n_gpus': 1})" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"n_gpus": 1}\n'
    assert "banner" in r.stderr and "a stray print" in r.stderr


def test_ring_reservation_is_released_early_and_puts_back_what_it_found():
    """multi_gpu._reserved_cus: takes the units only when nobody holds any, release() (the ring's last step, which
    exchanges nothing) and the block's end put back what was found, and a reservation somebody else made in
    between stays."""
    mg, _ = _load()
    import flash_attention_amd as fa

    get = fa.lib().fa_get_reserved_cus
    fa.set_reserved_cus(0)
    try:
        with mg._reserved_cus(8) as r:
            assert get() == 8
            r.release()
            assert get() == 0
            r.release()  # (idempotent)
        assert get() == 0
        fa.set_reserved_cus(4)  # a caller's own reservation: left alone
        with mg._reserved_cus(8):
            assert get() == 4
        assert get() == 4
        fa.set_reserved_cus(0)
        with mg._reserved_cus(8):
            fa.set_reserved_cus(12)  # changed inside the block: theirs stays
        assert get() == 12
    finally:
        fa.set_reserved_cus(0)
