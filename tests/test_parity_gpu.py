"""GPU tier: the HIP path, called through the C ABI, against the CPU oracle on
identical fp16 inputs.  Gate: max-abs <= 2e-3 and relative L2 (testing.cu:236)
<= 1e-3 on O."""
import ctypes as ct
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import MAX_ABS_TOL, REL_L2_TOL, ROOT, load_golden, normal_fp16

pytestmark = pytest.mark.gpu


def gate(oracle, ref32, out16, what=""):
    m = oracle.error_metrics(ref32, out16)
    assert np.isfinite(out16.astype(np.float32)).all(), what
    assert m["max_abs"] <= MAX_ABS_TOL and m["global_rel"] <= REL_L2_TOL, (what, m)
    return m


@pytest.fixture(scope="module")
def torch_dev():
    import torch

    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def run_dev(fa, dev, Q, K, V):
    import torch

    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o = fa.attention(q, k, v)
    torch.cuda.synchronize()
    return o.cpu().numpy()


@pytest.mark.parametrize("name", ["f1_testing_l64.npz", "f2_c1_l512.npz"])
def test_golden_fixtures_host_pointer_path(fa, oracle, name):
    """testing.cu:188-191 / compare_impls.py:108-110: host pointers in, host pointers out."""
    g = load_golden(name)
    O, ms = fa.attention_host_half(g["Q"], g["K"], g["V"])
    assert ms > 0.0
    gate(oracle, g["O_oracle"], O, name)
    # the reference's own NumPy formulation (un-rounded P) is inside the same gate
    gate(oracle, g["O_ref_numpy"], O, name + " vs reference numpy")


@pytest.mark.parametrize("shape", [
    (1, 1, 64, 128), (1, 1, 128, 128), (2, 3, 192, 128), (1, 2, 256, 128), (1, 1, 320, 128),
    (3, 1, 512, 128), (1, 8, 1024, 128), (2, 8, 768, 128), (1, 5, 2048, 128),
])
def test_device_pointer_path_shapes(fa, oracle, torch_dev, shape):
    """Runtime B, H and every L % 64 == 0 are honoured (the reference only ran
    seven fixed shapes, wrapper.cu:89-185)."""
    Q, K, V = normal_fp16(11, shape)
    O = run_dev(fa, torch_dev, Q, K, V)
    gate(oracle, oracle.attention_cpu(Q, K, V), O, str(shape))


@pytest.fixture
def ab_build(fa, monkeypatch):
    """The comparison build of the library (make ab: -DFA_AB_KERNELS), which also holds the two register-staged
    kernels that the product build leaves out.  The front end is pointed at it for one test."""
    path = os.path.join(ROOT, "flash-attention_amd", "csrc", "libflash_attention_ab.so")
    assert os.path.exists(path), "build it with make -C flash-attention_amd/csrc ab"
    monkeypatch.setattr(fa, "_lib", fa.bind(path))
    return fa


@pytest.mark.parametrize("nwaves", ["4", "8"])
def test_both_workgroup_sizes(ab_build, oracle, torch_dev, nwaves, fa_env):
    fa = ab_build
    fa_env(FA_NWAVES=nwaves)
    assert fa.lib().fa_kernel_name(512, 512) == (b"fa::fa_fwd_kernel<4>" if nwaves == "4" else b"fa::fa_fwd_w64_kernel")
    for shape in [(1, 2, 64, 128), (1, 3, 448, 128), (2, 8, 512, 128)]:
        Q, K, V = normal_fp16(12, shape)
        O = run_dev(fa, torch_dev, Q, K, V)
        gate(oracle, oracle.attention_cpu(Q, K, V), O, f"{shape} nwaves={nwaves}")


@pytest.mark.parametrize("env,kernel", [
    ({"FA_W64": "1"}, b"fa::fa_fwd_w64_kernel"),  # (forced from three tiles on; by default from five)
    ({"FA_W64": "0"}, b"fa::fa_fwd_ring_kernel<8>"),
    ({"FA_W64": "0", "FA_RING": "0"}, b"fa::fa_fwd_kernel<8>"),  # (comparison build only)
])
def test_every_kernel_variant(fa, oracle, torch_dev, env, kernel, fa_env, request):
    """The launcher's three kernels for 256-row query blocks, each on: few tiles per item, several
    items per workgroup (more items than CUs), rows whose maximum jumps from tile to tile, and
    bit-identical results across the variants' common shapes."""
    if "FA_RING" in env:
        fa = request.getfixturevalue("ab_build")
    fa_env(**env)
    assert fa.lib().fa_kernel_name(512, 512) == kernel
    assert fa.lib().fa_kernel_name(256, 128) != b"fa::fa_fwd_w64_kernel"  # needs three tiles per item
    assert fa.lib().fa_kernel_name(320, 320) == kernel  # a last block of 64 rows: three waves without rows
    for shape in [(1, 1, 256, 128), (2, 3, 768, 128), (1, 9, 1280, 128), (2, 3, 320, 128), (1, 5, 448, 128),
                  (1, 3, 960, 128)]:
        Q, K, V = normal_fp16(21, shape)
        O = run_dev(fa, torch_dev, Q, K, V)
        gate(oracle, oracle.attention_cpu(Q, K, V), O, f"{shape} {kernel}")
    # 40*16 heads x 1 block = 640 items on 256 CUs: two or three items per workgroup, 4 tiles each
    shape = (40, 16, 256, 128)
    Q, K, V = normal_fp16(22, shape)
    O = run_dev(fa, torch_dev, Q, K, V)
    for b, h in [(0, 0), (17, 5), (39, 15)]:
        ref = oracle.attention_cpu(Q[b:b + 1, h:h + 1], K[b:b + 1, h:h + 1], V[b:b + 1, h:h + 1])
        gate(oracle, ref, O[b:b + 1, h:h + 1], f"{shape} slice {(b, h)} {kernel}")
    assert np.isfinite(O.astype(np.float32)).all()
    # a dominating key in every tile, growing from tile to tile: a rescale per tile in both sub-blocks
    rng = np.random.default_rng(23)
    shape = (1, 2, 512, 128)
    Q = (rng.standard_normal(shape) * 2).astype(np.float16)
    K = (rng.standard_normal(shape) * 2).astype(np.float16)
    V = rng.standard_normal(shape).astype(np.float16)
    for t in range(8):
        for row in (3, 40, 70, 200, 255, 300, 511):
            K[:, :, 64 * t + (row % 64), :] = (Q[:, :, row, :].astype(np.float32) * (0.15 + 0.08 * t)).astype(np.float16)
    O = run_dev(fa, torch_dev, Q, K, V)
    gate(oracle, oracle.attention_cpu(Q, K, V), O, f"spiked {kernel}")


def test_default_kernel_choice(fa):
    name = fa.lib().fa_kernel_name
    assert name(512, 512) == name(16384, 16384) == name(320, 384) == name(448, 448) == name(1984, 1984) == b"fa::fa_fwd_w64_kernel"
    assert name(64, 64) == name(256, 256) == name(512, 256) == b"fa::fa_fwd_ring_kernel<8>"
    # (self-attention at 320 and 384: the second 256-row block would be at most half full)
    assert name(320, 320) == name(384, 384) == b"fa::fa_fwd_hd_masked_kernel<128>"


def test_result_does_not_depend_on_what_ran_before(fa, oracle, torch_dev):
    """Registers and LDS keep the previous launch's contents: a kernel that reads anything it has
    not written itself gives results that depend on the launch before.  The same inputs are run
    after a launch on large random values and after a launch on infinities and NaNs."""
    import torch

    shape = (16, 16, 1024, 128)  # four items per workgroup on 256 CUs
    n = int(np.prod(shape))
    q = torch.empty(shape, dtype=torch.float16, device=torch_dev)
    k, v = torch.empty_like(q), torch.empty_like(q)
    fa.init_half(q, 1234), fa.init_half(k, 2345), fa.init_half(v, 3456)
    g = torch.Generator(device=torch_dev).manual_seed(5)
    noise = [torch.randn(n, generator=g, device=torch_dev).mul(50).to(torch.float16).view(shape) for _ in range(3)]
    poison = [torch.full(shape, float("nan"), dtype=torch.float16, device=torch_dev) for _ in range(3)]
    first = fa.attention(q, k, v).clone()
    fa.attention(*noise)
    second = fa.attention(q, k, v).clone()
    fa.attention(*poison)
    third = fa.attention(q, k, v).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(first.float()).all()
    assert torch.equal(first, second), "result changed after a launch on other data"
    assert torch.equal(first, third), "result changed after a launch on NaN data"
    O = first[3:4, 5:6].cpu().numpy()
    ref = oracle.attention_cpu(q[3:4, 5:6].cpu().numpy(), k[3:4, 5:6].cpu().numpy(), v[3:4, 5:6].cpu().numpy())
    gate(oracle, ref, O, "slice (3,5) of the pattern workload")


def test_testing_recipe_bias_and_large_scores(fa, oracle, torch_dev):
    """Inputs with a mean (testing.cu:160) and with large dot products: the running max has to
    move a lot between tiles."""
    rng = np.random.default_rng(13)
    shape = (1, 2, 512, 128)
    Q = (rng.standard_normal(shape) * 2 + 0.5).astype(np.float16)
    K = (rng.standard_normal(shape) * 2 + 0.5).astype(np.float16)
    V = (rng.standard_normal(shape) + 0.5).astype(np.float16)
    # one key per 64-key tile is made to dominate, later tiles more so: forces a rescale per tile
    for t in range(8):
        K[:, :, 64 * t + 5, :] = (Q[:, :, 17, :].astype(np.float32) * (0.2 + 0.1 * t)).astype(np.float16)
    O = run_dev(fa, torch_dev, Q, K, V)
    gate(oracle, oracle.attention_cpu(Q, K, V), O, "spiked")


def test_compare_impls_shape_head_slices(fa, oracle, torch_dev):
    """C2 = B=32,H=16,L=512 with the compare_impls.py:84-87 recipe; a few (b,h) slices are
    checked in full against the oracle, every slice for finiteness."""
    B, H, L, D = 32, 16, 512, 128
    Q, K, V = [a.reshape(B, H, L, D) for a in normal_fp16(0, B * H * L * D)]
    O = run_dev(fa, torch_dev, Q, K, V)
    assert np.isfinite(O.astype(np.float32)).all()
    for b, h in [(0, 0), (7, 3), (31, 15), (16, 8)]:
        sl = (slice(b, b + 1), slice(h, h + 1))
        ref = oracle.attention_cpu(np.ascontiguousarray(Q[sl]), np.ascontiguousarray(K[sl]), np.ascontiguousarray(V[sl]))
        gate(oracle, ref, O[sl], f"C2 slice {b},{h}")
    # MAE / MRE as compare_impls.py:66-74 reports them, vs the fp32 softmax expression (:100-102)
    sl = (slice(0, 1), slice(0, 2))
    dense = oracle.attention_dense_fp64(Q[sl], K[sl], V[sl]).astype(np.float32)
    mae, mre = oracle.mae_mre(dense, O[sl])
    assert mae < 1e-4 and mre < 1.5e-2, (mae, mre)   # the reference kernel recorded 6.05e-5 / 7.6e-3


def test_mid_shape_head_slices(fa, oracle, torch_dev):
    """C3 = B=8,H=16,L=2048."""
    B, H, L, D = 8, 16, 2048, 128
    Q, K, V = normal_fp16(21, (B, H, L, D))
    O = run_dev(fa, torch_dev, Q, K, V)
    assert np.isfinite(O.astype(np.float32)).all()
    for b, h in [(0, 0), (7, 15), (3, 9)]:
        sl = (slice(b, b + 1), slice(h, h + 1))
        ref = oracle.attention_cpu(np.ascontiguousarray(Q[sl]), np.ascontiguousarray(K[sl]), np.ascontiguousarray(V[sl]))
        gate(oracle, ref, O[sl], f"C3 slice {b},{h}")


def test_long_sequence_row_subset(fa, oracle, torch_dev):
    """C4 = B=1,H=16,L=16384: 96 rows spread over the sequence, all heads, against the oracle."""
    B, H, L, D = 1, 16, 16384, 128
    Q, K, V = normal_fp16(31, (B, H, L, D))
    O = run_dev(fa, torch_dev, Q, K, V)
    assert np.isfinite(O.astype(np.float32)).all()
    rows = np.unique(np.concatenate([np.arange(0, L, 257), [L - 1, L - 64, 255, 256]])).astype(np.int32)[:96]
    ref = oracle.attention_cpu_rows(Q, K, V, rows)
    gate(oracle, ref, O[:, :, rows, :], "C4 rows")


def test_bench_pattern_inputs_and_linearity(fa, oracle, torch_dev):
    """The throughput driver's inputs (benchmark_tensor_flash_attention.cu:34-41) generated on the
    device equal the CPU pattern bit for bit; attention on them passes the gate; and, as a
    size-independent property at full size, scaling V by 2 scales O by 2 exactly (power of two)."""
    import torch

    B, H, L, D = 4, 16, 4096, 128
    n = B * H * L * D
    q = torch.empty((B, H, L, D), dtype=torch.float16, device=torch_dev)
    k = torch.empty_like(q)
    v = torch.empty_like(q)
    fa.init_half(q, 1234)
    fa.init_half(k, 2345)
    fa.init_half(v, 3456)
    torch.cuda.synchronize()
    head = 1 << 20
    for t, seed in ((q, 1234), (k, 2345), (v, 3456)):
        got = t.reshape(-1)[:head].cpu().numpy()
        assert np.array_equal(got.view(np.uint16), oracle.init_half(head, seed).view(np.uint16))
        tail = t.reshape(-1)[n - 4096:].cpu().numpy()
        assert np.array_equal(tail.view(np.uint16), oracle.init_half(4096, seed, idx0=n - 4096).view(np.uint16))
    o1 = fa.attention(q, k, v)
    o2 = fa.attention(q, k, v * 2)
    torch.cuda.synchronize()
    assert torch.equal(o1 * 2, o2)
    rows = np.arange(0, L, 131, dtype=np.int32)
    Qn, Kn, Vn = [t[:1, :2].contiguous().cpu().numpy() for t in (q, k, v)]
    ref = oracle.attention_cpu_rows(Qn, Kn, Vn, rows)
    gate(oracle, ref, o1[:1, :2].cpu().numpy()[:, :, rows, :], "bench pattern")


def test_repeat_runs_are_bit_identical(fa, torch_dev):
    import torch

    Q, K, V = normal_fp16(41, (2, 16, 1024, 128))
    q, k, v = [torch.from_numpy(a).to(torch_dev) for a in (Q, K, V)]
    outs = [fa.attention(q, k, v).clone() for _ in range(4)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_elapsed_ms_and_stream_argument(fa, torch_dev):
    import torch

    Q, K, V = normal_fp16(42, (1, 16, 1024, 128))
    q, k, v = [torch.from_numpy(a).to(torch_dev) for a in (Q, K, V)]
    o0 = fa.attention(q, k, v)
    s = torch.cuda.Stream(device=torch_dev)
    torch.cuda.synchronize()
    o1, ms = fa.attention(q, k, v, stream=s, timed=True)   # synchronous when elapsed_ms is given
    assert 0.0 < ms < 100.0
    assert torch.equal(o0, o1)


@pytest.mark.parametrize("nblocks", [1, 2, 4])
def test_blockwise_entry_point(fa, oracle, torch_dev, nblocks):
    """fa_forward_block: K/V folded block by block with carried (O,m,l) equals one pass."""
    import torch

    B, H, L, D = 1, 4, 512, 128
    Q, K, V = normal_fp16(51, (B, H, L, D))
    q, k, v = [torch.from_numpy(a).to(torch_dev) for a in (Q, K, V)]
    st = fa.BlockState(B * H, L, D, torch_dev)
    step = L // nblocks
    out = None
    for c in range(nblocks):
        kb = k[:, :, c * step:(c + 1) * step].contiguous()
        vb = v[:, :, c * step:(c + 1) * step].contiguous()
        out = fa.attention_block(q, kb, vb, st, last=(c == nblocks - 1))
    torch.cuda.synchronize()
    ref = oracle.attention_cpu(Q, K, V)
    gate(oracle, ref, out.cpu().numpy(), f"blockwise {nblocks}")
    one = fa.attention(q, k, v)
    m = oracle.error_metrics(one.float().cpu().numpy(), out.cpu().numpy())
    assert m["max_abs"] <= 1e-3, m
    if nblocks > 1:
        # the carried state is the oracle's per-row state after the same blocks
        ost = oracle.BlockState(B, H, L, D)
        for c in range(nblocks - 1):
            oracle.block_update(ost, Q, np.ascontiguousarray(K[:, :, c * step:(c + 1) * step]),
                                np.ascontiguousarray(V[:, :, c * step:(c + 1) * step]))
        assert np.allclose(st.row_max.cpu().numpy().reshape(B, H, L), ost.mx, atol=1e-4)
        assert np.allclose(st.row_sum.cpu().numpy().reshape(B, H, L), ost.sm, rtol=2e-3)


@pytest.mark.parametrize("Lk", [192, 256])
def test_blockwise_shortest_items_many_per_workgroup(fa, oracle, torch_dev, Lk, fa_env):
    """The one-wave-per-SIMD kernel at its minimum of three (and four) K/V tiles per item, with more
    items than workgroups (300 heads x one 256-row block on 256 CUs): every step is an item's first,
    last-but-one or last step, and some workgroups carry state across two items."""
    import torch

    fa_env(FA_W64="1")  # (below five tiles the launcher would pick the ring kernel)
    BH, Lq, D, nblk = 300, 256, 128, 2
    assert fa.lib().fa_kernel_name(Lq, Lk) == b"fa::fa_fwd_w64_kernel"
    rng = np.random.default_rng(Lk)
    Q = rng.standard_normal((1, BH, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((1, BH, nblk * Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((1, BH, nblk * Lk, D), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(torch_dev) for a in (Q, K, V)]
    st = fa.BlockState(BH, Lq, D, torch_dev)
    out = None
    for c in range(nblk):
        out = fa.attention_block(q, k[:, :, c * Lk:(c + 1) * Lk].contiguous(), v[:, :, c * Lk:(c + 1) * Lk].contiguous(),
                                 st, last=(c == nblk - 1))
    torch.cuda.synchronize()
    ost = oracle.BlockState(1, BH, Lq, D)
    oracle.block_update(ost, Q, K, V)
    gate(oracle, oracle.block_finalize(ost), out.cpu().numpy(), f"shortest items Lk={Lk}")


@pytest.mark.parametrize("L", [64, 192, 320, 448, 960])
def test_partly_filled_last_block_stays_inside_its_tensors(fa, oracle, torch_dev, L):
    """L % 256 != 0 on 256-row workgroups: inputs are followed by NaNs and the output by a sentinel, so a
    read past the last head would poison the result and a write past it would show."""
    import torch

    shape = (2, 3, L, 128)
    Q, K, V = normal_fp16(30 + L, shape)
    n = Q.size

    def padded(a, fill):
        flat = torch.full((n + 65536,), fill, dtype=torch.float16, device=torch_dev)
        flat[:n] = torch.from_numpy(a).to(torch_dev).reshape(-1)
        return flat, flat[:n].view(*shape)

    (_, q), (_, k), (_, v) = padded(Q, float("nan")), padded(K, float("nan")), padded(V, float("nan"))
    o_flat, o = padded(np.zeros(shape, np.float16), 7.0)
    fa.attention(q, k, v, out=o)
    torch.cuda.synchronize()
    assert bool((o_flat[n:] == 7.0).all()), "wrote past the last row"
    gate(oracle, oracle.attention_cpu(Q, K, V), o.cpu().numpy(), f"L={L}")


def test_blockwise_partly_filled_last_query_block(fa, oracle, torch_dev):
    """Lq = 320: the second 256-row block holds 64 rows, three of its waves have none.  Output, carried
    accumulator and row statistics sit in buffers with a sentinel tail that must stay untouched."""
    import torch

    BH, Lq, Lk, D, nblk = 6, 320, 384, 128, 2
    assert fa.lib().fa_kernel_name(Lq, Lk) == b"fa::fa_fwd_w64_kernel"
    rng = np.random.default_rng(9)
    Q = rng.standard_normal((1, BH, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((1, BH, nblk * Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((1, BH, nblk * Lk, D), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(torch_dev) for a in (Q, K, V)]

    def padded(n, dtype, fill):
        flat = torch.full((n + 8192,), fill, dtype=dtype, device=torch_dev)
        return flat, flat[:n]

    st = fa.BlockState(BH, Lq, D, torch_dev)
    acc_flat, acc = padded(BH * Lq * D, torch.float32, 7.0)
    mx_flat, mx = padded(BH * Lq, torch.float32, 7.0)
    sm_flat, sm = padded(BH * Lq, torch.float32, 7.0)
    st.o_acc, st.row_max, st.row_sum = acc.view(BH, Lq, D), mx.view(BH, Lq), sm.view(BH, Lq)
    o_flat, o = padded(BH * Lq * D, torch.float16, 7.0)
    out = o.view(1, BH, Lq, D)
    for c in range(nblk):
        fa.attention_block(q, k[:, :, c * Lk:(c + 1) * Lk].contiguous(), v[:, :, c * Lk:(c + 1) * Lk].contiguous(), st,
                           last=(c == nblk - 1), out=out)
    torch.cuda.synchronize()
    for flat, n in ((acc_flat, BH * Lq * D), (mx_flat, BH * Lq), (sm_flat, BH * Lq), (o_flat, BH * Lq * D)):
        assert bool((flat[n:] == 7.0).all()), "wrote past the last row"
    ost = oracle.BlockState(1, BH, Lq, D)
    oracle.block_update(ost, Q, K, V)
    gate(oracle, oracle.block_finalize(ost), out.cpu().numpy(), "partly filled last block, blockwise")


def test_blockwise_rectangular(fa, oracle, torch_dev):
    """Lq != Lk, blocks visited in rotated (ring) order."""
    import torch

    B, H, Lq, Lk, D = 1, 2, 128, 512, 128
    rng = np.random.default_rng(52)
    Q = rng.standard_normal((B, H, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(torch_dev) for a in (Q, K, V)]
    st = fa.BlockState(B * H, Lq, D, torch_dev)
    order = [2, 3, 0, 1]
    out = None
    for i, c in enumerate(order):
        out = fa.attention_block(q, k[:, :, c * 128:(c + 1) * 128].contiguous(),
                                 v[:, :, c * 128:(c + 1) * 128].contiguous(), st, last=(i == 3))
    torch.cuda.synchronize()
    ost = oracle.BlockState(B, H, Lq, D)
    for c in range(4):
        oracle.block_update(ost, Q, np.ascontiguousarray(K[:, :, c * 128:(c + 1) * 128]),
                            np.ascontiguousarray(V[:, :, c * 128:(c + 1) * 128]))
    gate(oracle, oracle.block_finalize(ost), out.cpu().numpy(), "rectangular ring order")


def test_error_convention_of_drop_in_symbols():
    """wrapper.cu:56: a bad tile raises std::runtime_error through the C boundary; under ctypes
    that is std::terminate.  Run in a child process and look at how it died."""
    code = (
        "import sys; sys.path.insert(0, %r); import __graft_entry__ as ge, numpy as np; fa = ge.load_package();"
        "a = np.zeros((1,1,64,128), dtype=np.float16); p = a.ctypes.data_as(fa.HalfPtr);"
        "fa.lib().run_tensor_flash_attention_host_half(p,p,p,p,1,1,64,128,32,None,None)" % ROOT
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "tile must be 64" in r.stderr


def test_device_pointer_entry_points_can_be_captured_in_a_graph(fa, torch_dev):
    """With elapsed_ms == NULL the device-pointer entry points neither allocate nor synchronise, so a stream
    capture may hold them: tuned, masked and general kernels captured once and replayed on new inputs give the
    bits of the eager calls."""
    import torch

    def make(seed, L):
        g = torch.Generator(device=torch_dev).manual_seed(seed)
        return [torch.randn((2, 4, L, 128), generator=g, device=torch_dev).to(torch.float16) for _ in range(3)]

    q, k, v = make(1, 512)
    qr, kr, vr = make(3, 500)  # a ragged length: the masked form of the tuned kernel
    outs = [torch.empty_like(q), torch.empty_like(q), torch.empty_like(qr)]
    fa.attention(q, k, v, out=outs[0])  # first call outside the capture (it caches the device's CU count)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fa.attention(q, k, v, out=outs[0])
        fa.attention_general(q, k, v, causal=True, out=outs[1])
        fa.attention_general(qr, kr, vr, out=outs[2])
    for t, new in zip((q, k, v, qr, kr, vr), make(2, 512) + make(4, 500)):
        t.copy_(new)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(outs[0], fa.attention(q, k, v))
    assert torch.equal(outs[1], fa.attention_general(q, k, v, causal=True))
    assert torch.equal(outs[2], fa.attention_general(qr, kr, vr))


def test_cpp_correctness_driver():
    """The `testing` binary (counterpart of testing.cu) exits 0 and prints the five metrics."""
    exe = os.path.join(ROOT, "tests", "cpp", "testing")
    assert os.path.exists(exe), "build it with make -C flash-attention_amd/csrc"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Global Relative Error" in r.stdout and "PASS" in r.stdout
    r = subprocess.run([exe, "2", "3", "320"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    # the causal mask of attention_cpu and a length that is not a multiple of 64 (the general entry point)
    for extra in (["2", "3", "512", "1"], ["1", "2", "1100", "1"], ["1", "2", "700", "0"]):
        r = subprocess.run([exe] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (extra, r.stdout + r.stderr)


def test_compare_impls_counterpart():
    """The ctypes caller flow of benchmarks/compare_impls.py (same argtypes, same recipe) against
    the torch fp32 softmax reference: the reference kernel recorded MAE 6.05e-5 / MRE 7.6e-3."""
    script = os.path.join(ROOT, "flash-attention_amd", "benchmarks", "compare_impls.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("Tensor Cores: MAE=")][0]
    mae = float(line.split("MAE=")[1].split()[0])
    mre = float(line.split("MRE=")[1])
    assert mae < 1e-4 and mre < 1.5e-2, line
    # the sibling libraries' rows (compare_impls.py:133-136) come from the same kernels here
    for name in ("CUDA Cores", "Naive"):
        row = [l for l in r.stdout.splitlines() if l.startswith(name + ": MAE=")][0]
        assert float(row.split("MAE=")[1].split()[0]) < 1e-4, row
    assert "Speed (ms)" in r.stdout
