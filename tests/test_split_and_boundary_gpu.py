"""GPU tier: launches with few heads (keys split over workgroups and folded by a second kernel), the drop-in
device-pointer symbol itself, concurrent callers, rows that have seen no key, and the long-sequence ring
configuration played on one GPU.  Gate as everywhere: max-abs <= 2e-3, relative L2 (testing.cu:236) <= 1e-3
against the CPU oracle on identical fp16 inputs."""
import ctypes as ct
import threading

import numpy as np
import pytest

from conftest import MAX_ABS_TOL, REL_L2_TOL, normal_fp16

pytestmark = pytest.mark.gpu


def gate(oracle, ref32, out16, what=""):
    m = oracle.error_metrics(ref32, out16)
    assert np.isfinite(out16.astype(np.float32)).all(), what
    assert m["max_abs"] <= MAX_ABS_TOL and m["global_rel"] <= REL_L2_TOL, (what, m)
    return m


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def to_dev(dev, *arrays):
    import torch

    return [torch.from_numpy(a).to(dev) for a in arrays]


# ------------------------------------------------------------------------------------------------
# key split
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H", [1, 2, 3, 4])
def test_few_heads_long_sequence(fa, oracle, dev, H, fa_env):
    """B=1, H in 1..4, L=16384: what one rank holds of the 16-head long-sequence configuration on 4, 8 (and more)
    GPUs.  64*H items for 256 compute units: the launcher splits the keys (H = 1, 2, 3) or not (H = 4); either
    way 96 rows spread over the sequence, all heads, against the oracle, and against the unsplit launch."""
    import torch

    fa_env(FA_SPLIT=None)
    L = 16384
    Q, K, V = normal_fp16(70 + H, (1, H, L, 128))
    q, k, v = to_dev(dev, Q, K, V)
    o = fa.attention(q, k, v)
    fa_env(FA_SPLIT="0")
    o_unsplit = fa.attention(q, k, v)
    torch.cuda.synchronize()
    O = o.cpu().numpy()
    assert np.isfinite(O.astype(np.float32)).all()
    rows = np.unique(np.concatenate([np.arange(0, L, 173), [L - 1, L - 64, 255, 256]])).astype(np.int32)[:96]
    gate(oracle, oracle.attention_cpu_rows(Q, K, V, rows), O[:, :, rows, :], f"H={H}")
    # the two launches add the same products in different groupings: equal within fp16 rounding of O
    d = (o.float() - o_unsplit.float()).abs().max().item()
    assert d <= 1e-3, d


@pytest.mark.parametrize("shape,S", [((1, 2, 1536, 128), 2), ((1, 2, 1536, 128), 8), ((1, 1, 2048, 128), 4),
                                     ((2, 3, 768, 128), 4), ((1, 5, 1280, 128), 2), ((1, 1, 576, 128), 3),
                                     ((1, 2, 960, 128), 5)])
def test_forced_split_full_tensor(fa, oracle, dev, shape, S, fa_env):
    """FA_SPLIT=S forces S key ranges per head (the smallest range here is three tiles, the kernel's minimum);
    whole tensors against the oracle, partly filled last query blocks (L % 256 != 0) included; results repeat
    bit for bit, and the bytes behind the output stay untouched."""
    import torch

    fa_env(FA_SPLIT=str(S))
    Q, K, V = normal_fp16(300 + S, shape)
    q, k, v = to_dev(dev, Q, K, V)
    n = Q.size
    o_flat = torch.full((n + 8192,), 7.0, dtype=torch.float16, device=dev)
    o = o_flat[:n].view(*shape)
    fa.attention(q, k, v, out=o)
    first = o.clone()
    fa.attention(q, k, v, out=o)
    torch.cuda.synchronize()
    assert torch.equal(first, o)
    assert bool((o_flat[n:] == 7.0).all()), "wrote past the last row"
    gate(oracle, oracle.attention_cpu(Q, K, V), o.cpu().numpy(), f"{shape} S={S}")


def test_split_with_large_score_spread(fa, oracle, dev, fa_env):
    """Key ranges whose row maxima differ by a lot: the merge has to weigh them by exp(m_s - m)."""
    fa_env(FA_SPLIT="4")
    rng = np.random.default_rng(5)
    shape = (1, 2, 1024, 128)
    Q = (rng.standard_normal(shape) * 2).astype(np.float16)
    K = (rng.standard_normal(shape) * 2).astype(np.float16)
    V = rng.standard_normal(shape).astype(np.float16)
    for r in range(4):  # one dominating key per range, stronger from range to range
        for row in (3, 100, 700, 1023):
            K[:, :, 256 * r + (row % 256), :] = (Q[:, :, row, :].astype(np.float32) * (0.1 + 0.1 * r)).astype(np.float16)
    q, k, v = to_dev(dev, Q, K, V)
    import torch

    o = fa.attention(q, k, v)
    torch.cuda.synchronize()
    gate(oracle, oracle.attention_cpu(Q, K, V), o.cpu().numpy(), "spread")


def test_split_inside_general_entry_point_and_timed(fa, oracle, dev, fa_env):
    """Lq != Lk in whole tiles goes through fa_forward_general to the same launcher; elapsed_ms covers both kernels."""
    import torch

    fa_env(FA_SPLIT="2")
    rng = np.random.default_rng(6)
    Q = rng.standard_normal((1, 2, 512, 128), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((1, 2, 1024, 128), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((1, 2, 1024, 128), dtype=np.float32).astype(np.float16)
    q, k, v = to_dev(dev, Q, K, V)
    o, ms = fa.attention_general(q, k, v, timed=True)
    assert 0.0 < ms < 100.0
    st = oracle.BlockState(1, 2, 512, 128)
    oracle.block_update(st, Q, K, V)
    gate(oracle, oracle.block_finalize(st), o.cpu().numpy(), "Lq != Lk, split")


# ------------------------------------------------------------------------------------------------
# the drop-in symbol itself
# ------------------------------------------------------------------------------------------------
def test_run_tensor_flash_attention_device_pointers(fa, oracle, dev):
    """wrapper.cuh:8-19 as compare_impls.py:34-40 would bind it: device pointers, a stream handle, elapsed_ms NULL
    (asynchronous on the stream) and non-NULL (waits, returns the kernel's milliseconds)."""
    import torch

    shape = (2, 4, 512, 128)
    Q, K, V = normal_fp16(81, shape)
    q, k, v = to_dev(dev, Q, K, V)
    ref = oracle.attention_cpu(Q, K, V)
    fn = fa.lib().run_tensor_flash_attention
    ptr = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), fa.HalfPtr)
    s = torch.cuda.Stream(device=dev)
    o1 = torch.zeros_like(q)
    fn(ptr(q), ptr(k), ptr(v), ptr(o1), 2, 4, 512, 128, 64, ct.c_void_p(s.cuda_stream), None)
    s.synchronize()
    gate(oracle, ref, o1.cpu().numpy(), "stream, elapsed NULL")
    o2 = torch.zeros_like(q)
    ms = ct.c_float(-1.0)
    fn(ptr(q), ptr(k), ptr(v), ptr(o2), 2, 4, 512, 128, 64, ct.c_void_p(s.cuda_stream), ct.byref(ms))
    assert 0.0 < ms.value < 100.0
    assert torch.equal(o1, o2)  # no synchronisation needed: the call has waited
    o3 = torch.zeros_like(q)
    fn(ptr(q), ptr(k), ptr(v), ptr(o3), 2, 4, 512, 128, 64, None, None)  # the NULL stream, as the reference's callers pass
    torch.cuda.synchronize()
    assert torch.equal(o1, o3)


def test_two_host_threads_two_streams(fa, oracle, dev):
    """The entry points keep no per-call global state: two host threads, each with a stream of its own and its own
    problem (one of them a split launch with a workspace from its stream's pool), give the single-threaded bits."""
    import torch

    shapes = [(2, 8, 1024, 128), (1, 2, 4096, 128)]
    data = [to_dev(dev, *normal_fp16(90 + i, s)) for i, s in enumerate(shapes)]
    expect = [fa.attention(*d).clone() for d in data]
    torch.cuda.synchronize()
    outs = [[None] * 8, [None] * 8]
    errors = []

    def work(i):
        try:
            s = torch.cuda.Stream(device=dev)
            for j in range(8):
                outs[i][j] = fa.attention(*data[i], stream=s)
            s.synchronize()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for i in range(2):
        for o in outs[i]:
            assert torch.equal(o, expect[i])


# ------------------------------------------------------------------------------------------------
# rows that have seen no key
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [64, 128, 256])
def test_last_on_rows_without_keys_gives_zero(fa, dev, D):
    """fa_forward_block_general with `last` on a state that has seen no key for some rows (a block wholly above the
    diagonal; a first block that starts after the first queries): status OK, those rows 0, nothing NaN."""
    import torch

    g = torch.Generator(device=dev).manual_seed(3)
    q, k, v = [torch.randn((1, 2, 192, D), generator=g, device=dev).to(torch.float16) for _ in range(3)]
    st = fa.BlockState(2, 192, D, dev)
    out = fa.attention_block_general(q, k, v, st, last=True, causal=True, q_pos0=0, k_pos0=1000)  # nothing visible
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and bool((out == 0).all())
    st = fa.BlockState(2, 192, D, dev)
    out = fa.attention_block_general(q, k, v, st, last=True, causal=True, q_pos0=0, k_pos0=100)  # rows 0..99 see nothing
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert bool((out[:, :, :100] == 0).all()) and bool((out[:, :, 100:].abs().sum(-1) > 0).all())
    # and a state carried over such a block is still usable
    st = fa.BlockState(2, 192, D, dev)
    fa.attention_block_general(q, k, v, st, last=False, causal=True, q_pos0=0, k_pos0=1000)
    out = fa.attention_block_general(q, k, v, st, last=True, causal=True, q_pos0=0, k_pos0=0)
    ref = fa.attention_general(q, k, v, causal=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref.float()).abs().max().item() <= 2e-3


# ------------------------------------------------------------------------------------------------
# B=1, H=16, L=131072: one rank of the 8-way sequence-parallel ring, on one GPU
# ------------------------------------------------------------------------------------------------
RING_WORLD, RING_L, RING_H, RING_D = 8, 131072, 16, 128


@pytest.fixture(scope="module")
def long_sequence(dev):
    """K and V of the whole 131072-row sequence (2^28 elements each: byte offsets pass 2^29) and every rank's Q."""
    import torch

    g = torch.Generator(device=dev).manual_seed(1)
    K, V, Q = [torch.randn((1, RING_H, RING_L, RING_D), generator=g, device=dev, dtype=torch.float32).to(torch.float16)
               for _ in range(3)]
    torch.cuda.synchronize()
    return Q, K, V


@pytest.mark.parametrize("rank", [3, 7])
def test_ring_rank_compute_full_length(fa, oracle, dev, long_sequence, rank):
    """The compute of rank `rank` in multi_gpu.ring_forward at the long-sequence configuration: its 16384 query
    rows against the 8 K/V blocks of 16384 keys in ring order, carried state in between; sampled rows against the
    oracle over the WHOLE sequence."""
    import torch

    from flash_attention_amd import multi_gpu as mg

    Q, K, V = long_sequence
    P, Lb = RING_WORLD, RING_L // RING_WORLD
    q = Q[:, :, rank * Lb:(rank + 1) * Lb].contiguous()
    st = fa.BlockState(RING_H, Lb, RING_D, dev)
    out = None
    for s, b in enumerate(mg.ring_block_order(rank, P)):
        kb = K[:, :, b * Lb:(b + 1) * Lb].contiguous()
        vb = V[:, :, b * Lb:(b + 1) * Lb].contiguous()
        out = fa.attention_block_general(q, kb, vb, st, last=(s == P - 1))
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    rows = np.unique(np.linspace(0, Lb - 1, 6).astype(np.int32))
    Qn = q[:, :, rows].cpu().numpy()
    ost = oracle.BlockState(1, RING_H, len(rows), RING_D)
    oracle.block_update(ost, np.ascontiguousarray(Qn), K.cpu().numpy(), V.cpu().numpy())
    gate(oracle, oracle.block_finalize(ost), out[:, :, rows].cpu().numpy(), f"ring rank {rank}")


def test_whole_tensor_offsets_beyond_32_bits(fa, oracle, dev, long_sequence):
    """One call on the whole B=1, H=16, L=131072 tensors (2^28 elements per tensor, the last head starts at byte
    offset 2^29 - 2^25): sampled rows of the first and the last head against the oracle.  The reference's 32-bit
    element offsets (tensor_flash_attention.cu:38-41) would still hold here, its byte arithmetic would not."""
    import torch

    Q, K, V = long_sequence
    o = fa.attention(Q, K, V)
    torch.cuda.synchronize()
    rows = np.array([0, 65535, 131071], dtype=np.int32)
    for h in (0, RING_H - 1):
        Qn, Kn, Vn = [t[:, h:h + 1].contiguous().cpu().numpy() for t in (Q, K, V)]
        ref = oracle.attention_cpu_rows(Qn, Kn, Vn, rows)
        gate(oracle, ref, o[:, h:h + 1][:, :, rows].cpu().numpy(), f"head {h}")


def test_zigzag_ring_rank_compute_full_length(fa, oracle, dev, long_sequence):
    """The compute of one rank of the 8-way zig-zag causal ring over 131072 rows (16 chunks of 8192; 17 (query chunk,
    key chunk) pairs, the two diagonal ones with the mask on absolute positions): sampled rows of both chunks against
    attention_cpu(causal) over the whole sequence."""
    import torch

    from flash_attention_amd import multi_gpu as mg

    Q, K, V = long_sequence
    P, rank = RING_WORLD, 2
    Lc = RING_L // (2 * P)
    chunk = lambda T, c: T[:, :, c * Lc:(c + 1) * Lc].contiguous()
    mine = mg.zigzag_chunks(rank, P)
    q2 = torch.stack([chunk(Q, c) for c in mine])

    def kv_of_step(s):
        theirs = mg.zigzag_chunks((rank - s) % P, P)
        return torch.stack((torch.stack([chunk(K, c) for c in theirs]), torch.stack([chunk(V, c) for c in theirs])))

    def step(q, k, v, state, last, causal, q_pos0, k_pos0):
        return fa.attention_block_general(q, k, v, state, last, causal=causal, q_pos0=q_pos0, k_pos0=k_pos0)

    out = mg.run_zigzag_steps(q2, kv_of_step, rank, P, Lc, step, fa.BlockState)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    rows = np.unique(np.linspace(0, Lc - 1, 5).astype(np.int32))
    Qn, Kn, Vn = Q.cpu().numpy(), K.cpu().numpy(), V.cpu().numpy()
    for slot, c in enumerate(mine):
        pos = (c * Lc + rows).astype(np.int32)
        ref = oracle.attention_cpu_rows(Qn, Kn, Vn, pos, causal=True)
        m = oracle.error_metrics(ref, out[slot].cpu().numpy()[:, :, rows])
        # (early rows of a causal sequence average few values: |O| reaches 2 to 5, where fp16 neighbours lie
        # 2e-3 to 4e-3 apart, so the absolute bound scales with |O| above 1)
        assert m["max_abs"] <= MAX_ABS_TOL * max(1.0, float(np.abs(ref).max())) and m["global_rel"] <= REL_L2_TOL, (c, m)


# ------------------------------------------------------------------------------------------------
# another stream's kernel beside the persistent attention kernels
# ------------------------------------------------------------------------------------------------
def test_side_stream_kernel_runs_inside_the_attention_window(fa, dev):
    """The ring hands K/V on while it computes, through transport kernels of a collective library on another
    stream.  The attention kernels are persistent (one workgroup per compute unit, each with the unit's whole LDS
    and register file), so such a kernel finds no room until an attention kernel ends, unless units are left free:
    with fa_set_reserved_cus(8) a kernel shaped like a transport kernel (8 workgroups x 512 threads, 32 KiB of LDS
    each, copying one ring step's 134 MB) that becomes ready when the attention pass starts must finish before
    the pass's FIRST kernel does, i.e. it ran beside it."""
    import ctypes as ct_
    import os

    import torch

    from conftest import ROOT

    path = os.path.join(ROOT, "tests", "cpp", "libside_copy.so")
    assert os.path.exists(path), "build it with make -C tests/cpp libside_copy.so"
    side = ct_.CDLL(path)
    side.side_copy_launch.argtypes = [ct_.c_void_p, ct_.c_void_p, ct_.c_size_t, ct_.c_int, ct_.c_int, ct_.c_int, ct_.c_int,
                                      ct_.c_void_p]
    H, Lb, D, steps = 16, 16384, 128, 4
    q, k, v = [torch.empty((1, H, Lb, D), dtype=torch.float16, device=dev) for _ in range(3)]
    for t, seed in zip((q, k, v), (1234, 2345, 3456)):
        fa.init_half(t, seed)
    src = torch.randn(2 * H * Lb * D, device=dev).to(torch.float16)
    dst = torch.zeros_like(src)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    st = fa.BlockState(H, Lb, D, dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    # first launch of the helper outside the timed window (it loads the helper's code object)
    assert side.side_copy_launch(src.data_ptr(), dst.data_ptr(), src.numel() * 2, 8, 512, 32768, 1, sb.cuda_stream) == 0
    torch.cuda.synchronize()
    dst.zero_()

    def run(reserved):
        fa.set_reserved_cus(reserved)
        st.first = True
        for s in range(steps):  # warm-up pass
            fa.attention_block(q, k, v, st, last=(s == steps - 1), stream=sa)
        torch.cuda.synchronize()
        a0, a_first, a1, b1 = ev(), ev(), ev(), ev()
        a0.record(sa)
        st.first = True
        fa.attention_block(q, k, v, st, last=False, stream=sa)
        a_first.record(sa)
        for s in range(1, steps):
            fa.attention_block(q, k, v, st, last=(s == steps - 1), stream=sa)
        a1.record(sa)
        sb.wait_event(a0)
        assert side.side_copy_launch(src.data_ptr(), dst.data_ptr(), src.numel() * 2, 8, 512, 32768, 1, sb.cuda_stream) == 0
        b1.record(sb)
        torch.cuda.synchronize()
        return a0.elapsed_time(b1), a0.elapsed_time(a_first), a0.elapsed_time(a1)

    try:
        side_done, first_done, all_done = run(8)
        assert torch.equal(src, dst)
        assert side_done < first_done < all_done, (side_done, first_done, all_done)
        side0, first0, all0 = run(0)  # for the record: without free units the side kernel waits for a kernel boundary
        print(f"reserved 8: side kernel done at {side_done:.3f} ms, first attention kernel at {first_done:.3f} ms, pass "
              f"{all_done:.3f} ms; reserved 0: {side0:.3f} / {first0:.3f} / {all0:.3f} ms")
    finally:
        fa.set_reserved_cus(0)
