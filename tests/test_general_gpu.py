"""GPU tier: fa_forward_general (sequence lengths that are not a multiple of 64, Lq != Lk, the causal
mask of attention_cpu, testing.cu:74-77) through the C ABI against the CPU oracle on identical fp16
inputs.  Same gate as the other parity tests: max-abs <= 2e-3, relative L2 <= 1e-3."""
import os

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


def run_general(fa, dev, Q, K, V, causal=False):
    """Inputs and output sit in buffers with a NaN / sentinel tail right behind them: a read past the
    last key row would poison the result, a write past the last query row would change the tail."""
    import torch

    def padded(a, fill):
        flat = torch.full((a.size + 4096,), fill, dtype=torch.float16, device=dev)
        flat[: a.size] = torch.from_numpy(a).to(dev).reshape(-1)
        return flat, flat[: a.size].view(*a.shape)

    _, q = padded(Q, float("nan"))
    _, k = padded(K, float("nan"))
    _, v = padded(V, float("nan"))
    o_flat, o = padded(np.zeros(Q.shape, np.float16), 123.0)
    fa.attention_general(q, k, v, causal=causal, out=o)
    torch.cuda.synchronize()
    assert bool((o_flat[Q.size:] == 123.0).all()), "wrote past the last query row"
    return o.cpu().numpy()


@pytest.mark.parametrize("L", [1, 2, 31, 63, 65, 100, 127, 129, 200, 333, 777])
def test_sequence_length_not_a_multiple_of_64(fa, oracle, dev, L):
    Q, K, V = normal_fp16(100 + L, (2, 3, L, 128))
    O = run_general(fa, dev, Q, K, V)
    gate(oracle, oracle.attention_cpu(Q, K, V), O, f"L={L}")


@pytest.mark.parametrize("L", [1, 5, 64, 65, 128, 200, 512, 1000])
def test_causal_mask(fa, oracle, dev, L):
    Q, K, V = normal_fp16(200 + L, (1, 3, L, 128))
    O = run_general(fa, dev, Q, K, V, causal=True)
    gate(oracle, oracle.attention_cpu(Q, K, V, causal=True), O, f"causal L={L}")


def test_causal_first_row_is_first_value_row(fa, dev):
    """Row 0 sees key 0 only: its output is V[0] exactly (p = 1, l = 1)."""
    Q, K, V = normal_fp16(7, (1, 2, 130, 128))
    O = run_general(fa, dev, Q, K, V, causal=True)
    assert np.array_equal(O[:, :, 0], V[:, :, 0])


@pytest.mark.parametrize("Lq,Lk", [(1, 500), (100, 64), (130, 1000), (512, 77), (64, 192), (257, 3)])
def test_query_and_key_lengths_differ(fa, oracle, dev, Lq, Lk):
    rng = np.random.default_rng(Lq * 1000 + Lk)
    Q = rng.standard_normal((2, 2, Lq, 128), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((2, 2, Lk, 128), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((2, 2, Lk, 128), dtype=np.float32).astype(np.float16)
    O = run_general(fa, dev, Q, K, V)
    st = oracle.BlockState(2, 2, Lq, 128)
    oracle.block_update(st, Q, K, V)
    gate(oracle, oracle.block_finalize(st), O, f"Lq={Lq} Lk={Lk}")


def test_large_scores_with_tail_and_mask(fa, oracle, dev):
    """Scores of a few hundred (biased inputs) with a ragged tail and the causal mask: the masked
    entries must stay out of the row maximum and the rescale must fire."""
    rng = np.random.default_rng(3)
    shape = (1, 2, 333, 128)
    Q, K = [(rng.standard_normal(shape, dtype=np.float32) * 2.0 + 0.5).astype(np.float16) for _ in range(2)]
    V = rng.standard_normal(shape, dtype=np.float32).astype(np.float16)  # |O| stays where half an fp16 ulp < 2e-3
    for causal in (False, True):
        O = run_general(fa, dev, Q, K, V, causal=causal)
        gate(oracle, oracle.attention_cpu(Q, K, V, causal=causal), O, f"large scores causal={causal}")


def test_long_causal_row_subset(fa, oracle, dev):
    L = 4096 + 40
    Q, K, V = normal_fp16(9, (1, 2, L, 128))
    O = run_general(fa, dev, Q, K, V, causal=True)
    rows = np.unique(np.concatenate([np.arange(0, L, 97), [63, 64, 65, 127, 128, L - 2, L - 1]])).astype(np.int32)
    ref = oracle.attention_cpu_rows(Q, K, V, rows, causal=True)
    gate(oracle, ref, O[:, :, rows], "long causal rows")


CAUSAL_TUNED = b"fa::fa_fwd_w64_causal_kernel"


def masked_tuned(Lq, Lk, causal):
    """Name of the one-wave-per-SIMD kernel's masked instantiation for a shape."""
    ragged = Lq % 64 != 0 or Lk % 64 != 0
    if causal:
        return b"fa::fa_fwd_w64_ragged_causal_kernel" if ragged else CAUSAL_TUNED
    assert ragged
    return b"fa::fa_fwd_w64_ragged_kernel"


def test_masked_kernel_choice(fa, fa_env):
    """D = 128 with the causal mask or ragged lengths: the masked forms of the one-wave-per-SIMD kernel where they
    were measured faster than 128-row workgroups (long sequences, or a last 256-row block more than half full), else
    the masked form of the pipelined 128-row kernel (`general` below; FA_HD=0 / 1 would put fa_fwd_general_kernel<128>
    in its place); FA_MASKED_W64=1 extends the former to every shape they can run (Lq >= 64, three tiles of keys)."""
    name = fa.lib().fa_kernel_name_general
    general = b"fa::fa_fwd_hd_masked_kernel<128>"
    fa_env(FA_HD=None)
    fa_env(FA_MASKED_W64=None)
    for L in (512, 960, 1000, 1023, 1024, 1216, 1535, 1700, 1920, 2048, 2305, 16383, 16384):
        assert name(L, L, 128, 1) == masked_tuned(L, L, True)
    for L in (64, 128, 192, 193, 256, 320, 384, 448, 500, 576, 640, 700, 768, 959, 1025, 1088, 1344, 1537, 1793, 2049):
        assert name(L, L, 128, 1) == general
    for Lq, Lk in ((2047, 2047), (1100, 320), (1153, 1153), (16383, 16383)):
        assert name(Lq, Lk, 128, 0) == masked_tuned(Lq, Lk, False)
    for Lq, Lk in ((63, 1000), (511, 191), (1, 5000), (257, 512), (100, 4000), (321, 321), (600, 320), (513, 513), (511, 511),
                   (193, 1000), (1025, 1025)):
        assert name(Lq, Lk, 128, 0) == general
    fa_env(FA_HD="1")  # (against the general kernel the tuned forms win earlier)
    for L in (192, 256, 448, 500, 700, 1088):
        assert name(L, L, 128, 1) == masked_tuned(L, L, True)
    assert name(600, 320, 128, 0) == masked_tuned(600, 320, False) and name(511, 511, 128, 0) == masked_tuned(511, 511, False)
    fa_env(FA_HD=None)
    assert name(512, 512, 64, 1) == b"fa::fa_fwd_hd_masked_kernel<64>" and name(511, 511, 256, 0) == b"fa::fa_fwd_hd_masked_kernel<256>"
    assert name(512, 512, 128, 0) == fa.lib().fa_kernel_name(512, 512)
    fa_env(FA_MASKED_W64="1")
    for L in (320, 200, 577):
        assert name(L, L, 128, 1) == masked_tuned(L, L, True)
    assert name(64, 193, 128, 0) == masked_tuned(64, 193, False)
    assert name(128, 128, 128, 1) == general and name(63, 500, 128, 0) == general and name(100, 191, 128, 0) == general
    fa_env(FA_MASKED_W64="0")
    assert name(512, 512, 128, 1) == general and name(511, 511, 128, 0) == general
    fa_env(FA_HD="1")
    assert name(512, 512, 128, 1) == b"fa::fa_fwd_general_kernel<128>" == name(63, 1000, 128, 0)


@pytest.mark.parametrize("B,H,L", [(2, 2, 192), (1, 3, 256), (2, 3, 320), (1, 5, 448), (2, 4, 512), (1, 3, 576),
                                   (1, 2, 1024), (1, 3, 1088)])
def test_causal_mask_on_the_tuned_kernel(fa, oracle, dev, B, H, L, fa_env):
    """Whole tensors against attention_cpu(causal); the shortest sequence the kernel takes (three tiles) and
    lengths with a partly filled last query block included (320, 448, 576, 1088: waves without rows of their
    own)."""
    fa_env(FA_MASKED_W64="1")
    assert fa.lib().fa_kernel_name_general(L, L, 128, 1) == CAUSAL_TUNED
    Q, K, V = normal_fp16(300 + L, (B, H, L, 128))
    O = run_general(fa, dev, Q, K, V, causal=True)
    gate(oracle, oracle.attention_cpu(Q, K, V, causal=True), O, f"causal tuned L={L}")
    assert np.array_equal(O[:, :, 0], V[:, :, 0])  # row 0 sees key 0 only


@pytest.mark.parametrize("B,H,L", [(8, 16, 1024), (7, 21, 512), (5, 16, 2560), (3, 5, 1280)])
def test_causal_tuned_kernel_item_order(fa, oracle, dev, B, H, L, fa_env):
    """More items than workgroups: two whole rounds (the second dealt in reverse), a partial second round with
    B*H not a multiple of 8, two whole rounds plus a partial one, and fewer items than workgroups.  Every row of
    every head against the general kernel (a head or block computed twice, not at all or with another block's
    tile count shows there), sampled rows against the oracle."""
    import torch

    assert fa.lib().fa_kernel_name_general(L, L, 128, 1) == CAUSAL_TUNED
    Q, K, V = normal_fp16(400 + L, (B, H, L, 128))
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o = torch.full_like(q, float("nan"))
    fa.attention_general(q, k, v, causal=True, out=o)
    fa_env(FA_MASKED_W64="0", FA_HD="1")
    assert fa.lib().fa_kernel_name_general(L, L, 128, 1) == b"fa::fa_fwd_general_kernel<128>"
    o_ref = fa.attention_general(q, k, v, causal=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o.float()).all())
    assert float((o.float() - o_ref.float()).abs().max()) <= 1e-3
    rows = np.unique(np.concatenate([np.arange(0, L, 61), [255, 256, 257, L - 65, L - 64, L - 1]])).astype(np.int32)
    ref = oracle.attention_cpu_rows(Q, K, V, rows, causal=True)
    gate(oracle, ref, o.cpu().numpy()[:, :, rows], f"causal tuned item order {B}x{H}x{L}")


@pytest.mark.parametrize("Lq,Lk,causal", [(64, 193, False), (65, 200, False), (100, 321, False), (511, 511, False),
                                          (513, 513, False), (300, 1000, False), (777, 640, False), (640, 777, False),
                                          (255, 255, True), (257, 257, True), (200, 200, True), (511, 511, True),
                                          (513, 513, True), (1000, 1000, True), (1087, 1087, True)])
def test_ragged_lengths_on_the_tuned_kernel(fa, oracle, dev, Lq, Lk, causal, fa_env):
    """Lengths that are not whole tiles on the masked kernel: the last key tile is read shifted down to the last
    key (its repeated rows masked), a wave whose rows would pass the last query row takes the last 64 rows and
    stores only its own.  NaN-padded inputs and a sentinel-padded output catch reads and writes past the end."""
    fa_env(FA_MASKED_W64="1")
    assert fa.lib().fa_kernel_name_general(Lq, Lk, 128, int(causal)) == masked_tuned(Lq, Lk, causal)
    rng = np.random.default_rng(Lq * 7 + Lk)
    B, H = 2, 3
    Q = rng.standard_normal((B, H, Lq, 128), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((B, H, Lk, 128), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((B, H, Lk, 128), dtype=np.float32).astype(np.float16)
    O = run_general(fa, dev, Q, K, V, causal=causal)
    if causal:
        ref = oracle.attention_cpu(Q, K, V, causal=True)
    else:
        st = oracle.BlockState(B, H, Lq, 128)
        oracle.block_update(st, Q, K, V)
        ref = oracle.block_finalize(st)
    gate(oracle, ref, O, f"ragged tuned Lq={Lq} Lk={Lk} causal={causal}")
    O2 = run_general(fa, dev, Q, K, V, causal=causal)
    assert np.array_equal(O, O2)  # rows computed by two waves are stored by one: repeats are bit-exact


def test_unmasked_ragged_lengths_are_routed_by_the_batch(fa, oracle, dev, fa_env):
    """D = 128, no mask, lengths that are not whole tiles: with the batch known the launcher compares a cost model of
    the ragged tuned kernel (whole rounds of 256-row items) with one of the 128-row kernel.  32 x 16 heads of 511 rows
    are four whole rounds and go to the tuned kernel, 513 rows (a third block of one row) to the 128-row kernel;
    a batch that does not fill the chip keeps the length threshold (1100).  Either way the oracle's result."""
    import torch

    fa_env(FA_MASKED_W64=None, FA_HD=None)
    name = fa.lib().fa_kernel_name_general_bh
    assert name(512, 511, 511, 128, 0) == b"fa::fa_fwd_w64_ragged_kernel"
    assert name(512, 513, 513, 128, 0) == b"fa::fa_fwd_hd_masked_kernel<128>"
    assert name(512, 1023, 1023, 128, 0) == b"fa::fa_fwd_w64_ragged_kernel"
    assert name(6, 511, 511, 128, 0) == b"fa::fa_fwd_hd_masked_kernel<128>" == fa.lib().fa_kernel_name_general(511, 511, 128, 0)
    assert name(6, 1100, 1100, 128, 0) == b"fa::fa_fwd_w64_ragged_kernel"
    B, H, L = 32, 16, 511
    g = torch.Generator(device=dev).manual_seed(49)
    q, k, v = [torch.randn((B, H, L, 128), generator=g, device=dev).to(torch.float16) for _ in range(3)]
    o = fa.attention_general(q, k, v)
    fa_env(FA_MASKED_W64="0")
    o_hd = fa.attention_general(q, k, v)
    fa_env(FA_MASKED_W64=None)
    torch.cuda.synchronize()
    d = (o.float() - o_hd.float()).abs().max().item()
    assert 0 < d <= 2e-3, d  # (two kernels)
    sl = slice(5, 7)
    Q, K, V = [np.ascontiguousarray(t[sl].cpu().numpy()) for t in (q, k, v)]
    st = oracle.BlockState(2, H, L, 128)
    oracle.block_update(st, Q, K, V)
    gate(oracle, oracle.block_finalize(st), o[sl].cpu().numpy(), "511 rows on the ragged tuned kernel, routed by the batch")


@pytest.mark.parametrize("B,H,L,causal", [(9, 16, 1000, False), (7, 21, 705, True), (4, 16, 2100, True)])
def test_ragged_tuned_kernel_many_items(fa, oracle, dev, B, H, L, causal, fa_env):
    """Several items per workgroup with ragged lengths: every row against the general kernel, sampled rows
    against the oracle."""
    import torch

    fa_env(FA_MASKED_W64="1")
    assert fa.lib().fa_kernel_name_general(L, L, 128, int(causal)) == masked_tuned(L, L, causal)
    Q, K, V = normal_fp16(500 + L, (B, H, L, 128))
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o = torch.full_like(q, float("nan"))
    fa.attention_general(q, k, v, causal=causal, out=o)
    fa_env(FA_MASKED_W64="0")
    o_ref = fa.attention_general(q, k, v, causal=causal)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o.float()).all())
    assert float((o.float() - o_ref.float()).abs().max()) <= 1e-3
    rows = np.unique(np.concatenate([np.arange(0, L, 61), [255, 256, 257, L - 65, L - 64, L - 2, L - 1]])).astype(np.int32)
    if causal:
        ref = oracle.attention_cpu_rows(Q, K, V, rows, causal=True)
    else:
        ref = oracle.attention_cpu_rows(Q, K, V, rows)
    gate(oracle, ref, o.cpu().numpy()[:, :, rows], f"ragged tuned, many items {B}x{H}x{L} causal={causal}")


def test_causal_tuned_kernel_large_scores(fa, oracle, dev):
    """Scores of a few hundred with the mask on the tuned kernel: masked entries (often the largest of their
    tile) must stay out of the row maximum, and the rescale must fire on the visible ones."""
    rng = np.random.default_rng(11)
    shape = (1, 4, 512, 128)
    Q, K = [(rng.standard_normal(shape, dtype=np.float32) * 2.0 + 0.5).astype(np.float16) for _ in range(2)]
    V = rng.standard_normal(shape, dtype=np.float32).astype(np.float16)
    # rising scores along the keys: every later (masked) key would dominate the row if it leaked in
    K = (K.astype(np.float32) * np.linspace(0.5, 1.5, 512, dtype=np.float32)[None, None, :, None]).astype(np.float16)
    O = run_general(fa, dev, Q, K, V, causal=True)
    gate(oracle, oracle.attention_cpu(Q, K, V, causal=True), O, "causal tuned, large scores")


def test_tuned_shapes_are_passed_on(fa, dev):
    """Lq == Lk, a multiple of 64, no mask: the same kernel as fa_forward, so the same bits."""
    import torch

    Q, K, V = normal_fp16(5, (2, 4, 512, 128))
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    a = fa.attention(q, k, v)
    b = fa.attention_general(q, k, v)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_argument_errors(fa, dev):
    import torch

    q = torch.zeros((1, 1, 64, 128), dtype=torch.float16, device=dev)
    k = torch.zeros((1, 1, 128, 128), dtype=torch.float16, device=dev)
    with pytest.raises(fa.FlashAttentionError, match="causal mask needs Lq == Lk"):
        fa.attention_general(q, k, k, causal=True)
    q96 = torch.zeros((1, 1, 64, 96), dtype=torch.float16, device=dev)
    with pytest.raises(fa.FlashAttentionError, match="D must be 64, 128 or 256"):
        fa.attention_general(q96, q96, q96)


@pytest.mark.parametrize("D", [64, 256])
@pytest.mark.parametrize("L,causal", [(64, False), (100, False), (512, False), (777, False), (200, True), (640, True)])
def test_other_head_dims(fa, oracle, dev, D, L, causal):
    """D = 64 and 256 (the reference fixes 128, cuda_utils.cuh:11; the CPU baseline takes any D)."""
    Q, K, V = normal_fp16(300 + D + L, (2, 2, L, D))
    O = run_general(fa, dev, Q, K, V, causal=causal)
    gate(oracle, oracle.attention_cpu(Q, K, V, causal=causal), O, f"D={D} L={L} causal={causal}")


@pytest.mark.parametrize("D", [64, 256])
def test_other_head_dims_rectangular(fa, oracle, dev, D):
    rng = np.random.default_rng(D)
    Lq, Lk = 130, 333
    Q = rng.standard_normal((1, 3, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((1, 3, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((1, 3, Lk, D), dtype=np.float32).astype(np.float16)
    O = run_general(fa, dev, Q, K, V)
    st = oracle.BlockState(1, 3, Lq, D)
    oracle.block_update(st, Q, K, V)
    gate(oracle, oracle.block_finalize(st), O, f"D={D} rectangular")


def test_host_pointer_form(fa, oracle):
    Q, K, V = normal_fp16(31, (1, 2, 100, 128))
    O, ms = fa.attention_general_host_half(Q, K, V, causal=True)
    assert ms > 0.0
    gate(oracle, oracle.attention_cpu(Q, K, V, causal=True), O, "host pointers, causal, L=100")


def test_sibling_library_symbols(fa, oracle):
    """The reference's three-way caller (benchmarks/compare_impls.py:42-63, 112-141) calls these two with
    (..., B, H, L, D, 32, 4, None, byref(ms)) and (..., B, H, L, D, 0, None, byref(ms))."""
    import ctypes as ct

    B, H, L, D = 2, 2, 256, 128
    Q, K, V = normal_fp16(32, (B, H, L, D))
    ref = oracle.attention_cpu(Q, K, V)
    hp = lambda a: a.ctypes.data_as(fa.HalfPtr)
    for call in ("cuda_cores", "naive", "naive_causal"):
        O = np.zeros_like(Q)
        ms = ct.c_float(0.0)
        if call == "cuda_cores":
            fa.lib().run_cuda_cores_flash_attention_host_half(hp(Q), hp(K), hp(V), hp(O), B, H, L, D, 32, 4, None, ct.byref(ms))
        else:
            fa.lib().run_naive_attention_host_half(hp(Q), hp(K), hp(V), hp(O), B, H, L, D, int(call == "naive_causal"),
                                                   None, ct.byref(ms))
        assert ms.value > 0.0
        gate(oracle, oracle.attention_cpu(Q, K, V, causal=True) if call == "naive_causal" else ref, O, call)


@pytest.mark.parametrize("D", [64, 128, 256])
def test_blockwise_general_blocks_in_any_order(fa, oracle, dev, D):
    """fa_forward_block_general: a causal sequence folded from K/V blocks of unequal, ragged sizes in
    shuffled order, absolute positions given per block; rows that see nothing of a block keep their
    state.  Reference: attention_cpu with causal=True over the whole sequence."""
    import torch

    B, H, L = 1, 3, 333
    Q, K, V = normal_fp16(400 + D, (B, H, L, D))
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    cuts = [0, 70, 71, 200, 264, 333]
    order = [3, 0, 4, 2, 1]  # starts with a block most rows cannot see at all
    st = fa.BlockState(B * H, L, D, dev)
    out = None
    for n, c in enumerate(order):
        a, b = cuts[c], cuts[c + 1]
        out = fa.attention_block_general(q, k[:, :, a:b].contiguous(), v[:, :, a:b].contiguous(), st,
                                         last=(n == len(order) - 1), causal=True, q_pos0=0, k_pos0=a)
    torch.cuda.synchronize()
    gate(oracle, oracle.attention_cpu(Q, K, V, causal=True), out.cpu().numpy(), f"any order D={D}")


def test_blockwise_general_hands_tuned_shapes_to_the_tuned_kernels(fa, oracle, dev, fa_env):
    """fa_forward_block_general sends a block to the tuned kernels when it needs no mask and has their shape, and
    the diagonal block of a causal sequence to the causal tuned kernel; the kernels carry the same state, so a
    sequence may mix them: three chunks of 512 rows and a ragged tail of 100, each query chunk folded over its
    key chunks in shuffled order, against attention_cpu(causal) over the whole sequence."""
    import torch

    fa_env(FA_MASKED_W64=None)
    fa_env(FA_BLOCK_TUNED=None)
    name = fa.lib().fa_kernel_name_block_general
    general, tuned = b"fa::fa_fwd_hd_state_kernel<128>", b"fa::fa_fwd_w64_kernel"
    assert name(512, 512, 128, 1, 512, 512) == CAUSAL_TUNED     # the diagonal block
    assert name(512, 512, 128, 1, 1024, 512) == tuned           # wholly below the diagonal: no mask needed
    assert name(512, 512, 128, 1, 512, 1024) == general         # wholly above it: nothing to see
    assert name(512, 512, 128, 1, 512, 256) == general          # crossed by the diagonal elsewhere
    assert name(512, 512, 128, 0, 0, 0) == tuned
    assert name(512, 100, 128, 0, 0, 0) == general and name(100, 512, 128, 1, 1536, 0) == general
    assert name(512, 512, 64, 0, 0, 0) == b"fa::fa_fwd_hd_state_kernel<64>"

    B, H, L = 1, 5, 3 * 512 + 100
    Q, K, V = normal_fp16(77, (B, H, L, 128))
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    cuts = [0, 512, 1024, 1536, L]
    rng = np.random.default_rng(5)
    out = torch.empty_like(q)
    for qc in range(4):
        qa, qe = cuts[qc], cuts[qc + 1]
        qq = q[:, :, qa:qe].contiguous()
        st = fa.BlockState(B * H, qe - qa, 128, dev)
        order = list(rng.permutation(qc + 1))
        for n, kc in enumerate(order):
            ka, ke = cuts[kc], cuts[kc + 1]
            o = fa.attention_block_general(qq, k[:, :, ka:ke].contiguous(), v[:, :, ka:ke].contiguous(), st,
                                           last=(n == len(order) - 1), causal=True, q_pos0=qa, k_pos0=ka)
        out[:, :, qa:qe] = o
    torch.cuda.synchronize()
    ref = oracle.attention_cpu(Q, K, V, causal=True)
    gate(oracle, ref, out.cpu().numpy(), "mixed kernels, blockwise causal")

    # the same sequence with everything on the general kernel: same gate (and close to the mixed run)
    fa_env(FA_BLOCK_TUNED="0")
    assert name(512, 512, 128, 1, 512, 512) == general
    st = fa.BlockState(B * H, L, 128, dev)
    out_g = fa.attention_block_general(q, k, v, st, last=True, causal=True)
    torch.cuda.synchronize()
    gate(oracle, ref, out_g.cpu().numpy(), "general kernel only")
    assert float((out.float() - out_g.float()).abs().max()) <= 1e-3


def test_blockwise_general_ragged_chunks_on_the_tuned_kernels(fa, oracle, dev, fa_env):
    """Chunks that are not whole tiles: unmasked pairs on the ragged form of the tuned kernel, diagonal pairs on
    the ragged causal form, carrying state (rows that two waves compute are stored by their owner only)."""
    import torch

    fa_env(FA_MASKED_W64=None)
    fa_env(FA_BLOCK_TUNED=None)
    name = fa.lib().fa_kernel_name_block_general
    assert name(511, 700, 128, 1, 700, 0) == b"fa::fa_fwd_w64_ragged_kernel"           # below the diagonal
    assert name(700, 700, 128, 1, 0, 0) == b"fa::fa_fwd_w64_ragged_causal_kernel"     # diagonal pair
    assert name(600, 600, 128, 1, 0, 0) == b"fa::fa_fwd_hd_state_kernel<128>"          # (last block a third full)
    assert name(600, 600, 128, 0, 0, 0) == b"fa::fa_fwd_w64_ragged_kernel"
    assert name(50, 600, 128, 0, 0, 0) == b"fa::fa_fwd_hd_state_kernel<128>"            # fewer than 64 query rows
    B, H = 1, 4
    cuts = [0, 700, 1211, 1600, 1650]
    L = cuts[-1]
    Q, K, V = normal_fp16(91, (B, H, L, 128))
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    rng = np.random.default_rng(8)
    out = torch.full_like(q, float("nan"))
    for qc in range(len(cuts) - 1):
        qa, qe = cuts[qc], cuts[qc + 1]
        qq = q[:, :, qa:qe].contiguous()
        st = fa.BlockState(B * H, qe - qa, 128, dev)
        order = list(rng.permutation(qc + 1))
        for n, kc in enumerate(order):
            ka, ke = cuts[kc], cuts[kc + 1]
            o = fa.attention_block_general(qq, k[:, :, ka:ke].contiguous(), v[:, :, ka:ke].contiguous(), st,
                                           last=(n == len(order) - 1), causal=True, q_pos0=qa, k_pos0=ka)
        out[:, :, qa:qe] = o
    torch.cuda.synchronize()
    gate(oracle, oracle.attention_cpu(Q, K, V, causal=True), out.cpu().numpy(), "ragged chunks, blockwise causal")


def test_query_and_key_lengths_in_whole_tiles_go_to_the_tuned_kernels(fa, oracle, dev):
    assert fa.lib().fa_kernel_name_general(256, 1024, 128, 0) == b"fa::fa_fwd_w64_kernel"
    rng = np.random.default_rng(21)
    Q = rng.standard_normal((2, 3, 256, 128), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((2, 3, 1024, 128), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((2, 3, 1024, 128), dtype=np.float32).astype(np.float16)
    O = run_general(fa, dev, Q, K, V)
    st = oracle.BlockState(2, 3, 256, 128)
    oracle.block_update(st, Q, K, V)
    gate(oracle, oracle.block_finalize(st), O, "Lq=256 Lk=1024 on the tuned kernel")


def test_blockwise_general_without_mask_matches_one_pass(fa, oracle, dev):
    import torch

    B, H, Lq, Lk, D = 2, 2, 100, 500, 128
    rng = np.random.default_rng(77)
    Q = rng.standard_normal((B, H, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    st = fa.BlockState(B * H, Lq, D, dev)
    out = None
    for n, (a, b) in enumerate([(0, 130), (130, 131), (131, 500)]):
        out = fa.attention_block_general(q, k[:, :, a:b].contiguous(), v[:, :, a:b].contiguous(), st, last=(n == 2))
    torch.cuda.synchronize()
    ost = oracle.BlockState(B, H, Lq, D)
    oracle.block_update(ost, Q, K, V)
    gate(oracle, oracle.block_finalize(ost), out.cpu().numpy(), "blockwise general, no mask")


def test_causal_zigzag_ring_on_one_gpu(fa, oracle, dev):
    """The compute schedule of multi_gpu.ring_forward_causal, every rank played in turn on one GPU
    (the visiting K/V chunks are taken from the full tensors instead of a neighbour)."""
    import importlib

    import torch

    mg = importlib.import_module("flash_attention_amd.multi_gpu")
    world, Lc, B, H, D = 4, 96, 1, 2, 128
    L = 2 * world * Lc
    Q, K, V = normal_fp16(500, (B, H, L, D))
    chunk = lambda a, c: torch.from_numpy(np.ascontiguousarray(a[:, :, c * Lc:(c + 1) * Lc])).to(dev)

    def step(q, kk, vv, state, last, causal, q_pos0, k_pos0):
        return fa.attention_block_general(q, kk, vv, state, last, causal=causal, q_pos0=q_pos0, k_pos0=k_pos0)

    O = np.zeros((B, H, L, D), np.float16)
    for rank in range(world):
        lo, hi = mg.zigzag_chunks(rank, world)
        q2 = torch.stack((chunk(Q, lo), chunk(Q, hi)))

        def kv_of_step(s, rank=rank):
            a, b = mg.zigzag_chunks((rank - s) % world, world)
            return torch.stack((torch.stack((chunk(K, a), chunk(K, b))), torch.stack((chunk(V, a), chunk(V, b)))))

        out = mg.run_zigzag_steps(q2, kv_of_step, rank, world, Lc, step, fa.BlockState)
        torch.cuda.synchronize()
        O[:, :, lo * Lc:(lo + 1) * Lc] = out[0].cpu().numpy()
        O[:, :, hi * Lc:(hi + 1) * Lc] = out[1].cpu().numpy()
    gate(oracle, oracle.attention_cpu(Q, K, V, causal=True), O, "zig-zag ring on one GPU")


def test_ring_forward_default_step_takes_any_length(fa, oracle, dev, tmp_path, fa_env):
    """multi_gpu.ring_forward with its default step (fa_forward_block_general) on one rank: rows per rank that are
    not a multiple of 64 go through; the result is, bit for bit, that of the one-call entry point on the same kernel
    (FA_MASKED_W64=1: at this length the one call would otherwise take the 128-row kernel), and within the gate
    of the one call's own choice."""
    import sys
    import torch
    import torch.distributed as dist

    sys.path.insert(0, fa.__path__[0])
    import multi_gpu as mg

    sys.modules.setdefault("flash_attention_amd", fa)
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/rendezvous", rank=0, world_size=1)
    try:
        Q, K, V = normal_fp16(33, (1, 4, 700, 128))
        q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
        fa_env(FA_MASKED_W64="1")
        out = mg.ring_forward(q, k, v)
        torch.cuda.synchronize()
        assert torch.equal(out, fa.attention_general(q, k, v))
        fa_env(FA_MASKED_W64=None)
        gate(oracle, fa.attention_general(q, k, v).cpu().numpy().astype(np.float32), out.cpu().numpy(),
             "ring step against the one call")
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# padded batches (fa_forward_varlen)
# ------------------------------------------------------------------------------------------------
def _varlen_case(fa, oracle, dev, B, H, Lq, Lk, D, q_lens, k_lens, causal, seed):
    """Sequences of mixed lengths in one padded batch; the padding holds NaN (keys, values, queries) and the
    output a sentinel.  Per sequence: the oracle on the truncated tensors; behind q_len the sentinel stays."""
    import torch

    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((B, H, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    Qp, Kp, Vp = Q.copy(), K.copy(), V.copy()
    for b in range(B):
        Qp[b, :, q_lens[b]:] = np.nan
        Kp[b, :, k_lens[b]:] = np.nan
        Vp[b, :, k_lens[b]:] = np.nan
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Qp, Kp, Vp)]
    o = torch.full((B, H, Lq, D), 123.0, dtype=torch.float16, device=dev)
    ql = torch.tensor(q_lens, dtype=torch.int32, device=dev)
    kl = torch.tensor(k_lens, dtype=torch.int32, device=dev)
    fa.attention_varlen(q, k, v, ql, kl, causal=causal, out=o)
    torch.cuda.synchronize()
    O = o.cpu().numpy()
    for b in range(B):
        nq, nk = q_lens[b], k_lens[b]
        assert (O[b, :, nq:] == 123.0).all(), f"sequence {b}: wrote behind its last query row"
        if nq == 0:
            continue
        if nk == 0:
            assert (O[b, :, :nq] == 0).all(), f"sequence {b}: no keys must give 0"
            continue
        Qs, Ks, Vs = [np.ascontiguousarray(a[b:b + 1, :, :n]) for a, n in ((Q, nq), (K, nk), (V, nk))]
        if nq == nk:
            ref = oracle.attention_cpu(Qs, Ks, Vs, causal=causal)
        else:  # attention_cpu wants one length: fold the keys into a state of the nq rows (no mask)
            assert not causal
            st = oracle.BlockState(1, H, nq, D)
            oracle.block_update(st, Qs, Ks, Vs)
            ref = oracle.block_finalize(st)
        gate(oracle, ref, O[b:b + 1, :, :nq], f"sequence {b} (q_len {nq}, k_len {nk}, causal {causal})")


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("D", [64, 128, 256])
def test_padded_batch_mixed_lengths(fa, oracle, dev, D, causal):
    """Lengths below 192, not multiples of 64, a full-length one, a single row, and an empty sequence in one batch."""
    lens = [700, 1, 191, 64, 0, 513, 640, 100]
    _varlen_case(fa, oracle, dev, 8, 3, 700, 700, D, lens, lens, causal, seed=900 + D)


def test_padded_batch_different_query_and_key_lengths(fa, oracle, dev):
    q_lens = [300, 17, 256, 1]
    k_lens = [511, 200, 0, 65]
    _varlen_case(fa, oracle, dev, 4, 2, 300, 512, 128, q_lens, k_lens, False, seed=77)


def test_padded_batch_only_key_lengths_given(fa, oracle, dev):
    """k_len alone (q_len NULL): every query row of the padded tensor is computed against its sequence's keys."""
    import torch

    B, H, L, D = 3, 2, 320, 128
    Q, K, V = normal_fp16(55, (B, H, L, D))
    k_lens = [320, 130, 1]
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o = fa.attention_varlen(q, k, v, None, torch.tensor(k_lens, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    for b in range(B):
        st = oracle.BlockState(1, H, L, D)
        oracle.block_update(st, np.ascontiguousarray(Q[b:b + 1]), np.ascontiguousarray(K[b:b + 1, :, :k_lens[b]]),
                            np.ascontiguousarray(V[b:b + 1, :, :k_lens[b]]))
        gate(oracle, oracle.block_finalize(st), o[b:b + 1].cpu().numpy(), f"sequence {b}")
    # no lengths at all: the unpadded entry point's result
    o2 = fa.attention_varlen(q, k, v)
    torch.cuda.synchronize()
    assert torch.equal(o2, fa.attention_general(q, k, v))


def _on_tuned_kernel(q_lens, k_lens, causal):
    """Which sequences of a padded D = 128 batch the library hands to the tuned kernel's list form (the rule of
    fa_varlen_plan_kernel and fa::varlen_on_tuned): without the mask those that cost less there, with it those over
    the thresholds; and when the others that the list form can take at all hold at most a quarter of the batch's
    rows x keys, those as well."""
    def over(ql, kl, tq, tk):
        if not causal:
            if tq != -1:
                return ql >= tq and kl >= tk
            if ql < 64 or kl < 192:  # (by cost: 256-row items against 128-row workgroups, two to a compute unit)
                return False
            f32 = np.float32
            tuned = f32(-(-ql // 256)) * (f32(1.33) * f32(-(-kl // 64)) + f32(3.6))
            rows128 = f32(0.5) * f32(-(-ql // 128)) * (f32(0.8535) * f32(-(-kl // 32)) + f32(0.094))
            return bool(tuned < rows128)
        if ql != kl:
            return False
        last = (ql - 1) % 256 + 1
        if tq == 1665 and last >= 192 and ql >= 960:
            return True
        return ql >= (tq if last > 128 else tk)

    tq, tk = (1665, 2305) if causal else (-1, -1)
    mq, mk = (192, 192) if causal else (64, 192)
    w_long = sum(float(a) * b for a, b in zip(q_lens, k_lens) if over(a, b, tq, tk))
    w_mid = sum(float(a) * b for a, b in zip(q_lens, k_lens) if not over(a, b, tq, tk) and over(a, b, mq, mk))
    if w_long > 0 and w_mid <= 0.25 * (w_long + w_mid):
        tq, tk = mq, mk
    return [over(a, b, tq, tk) for a, b in zip(q_lens, k_lens)]


@pytest.mark.parametrize("H", [9, 2])
def test_padded_batch_long_and_short_sequences(fa, oracle, dev, H, fa_env):
    """A padded batch at D = 128 with sequences on both sides of the rule by which a sequence runs on the tuned kernel's
    ragged form with its own lengths (what it costs there against the 128-row kernel): lengths that are and are not
    whole tiles, a sequence with few rows and many keys (300 x 1400: the 128-row kernel's), many rows and 1099 keys
    (a tie that the 128-row kernel gets), an empty
    one, and 9 heads (groups of eight and one) / 2.  The two kernels write disjoint sequences; together the
    batch; same bits as the 128-row kernel alone on the sequences it keeps.  In the 2-head batch the one short
    sequence (64 rows) holds 2 % of the work and joins the list instead of costing a launch of its own."""
    import torch

    q_lens = [1400, 1100, 1099, 1250, 0, 1400, 300, 1153]
    k_lens = [1400, 1234, 1400, 1100, 700, 1099, 1400, 1153]
    if H == 2:  # longer sequences, different padded lengths
        q_lens, k_lens = [2600, 1101, 64, 2305], [1100, 3000, 3000, 2111]
    B = len(q_lens)
    fa_env(FA_VARLEN_W64="1")  # (these batches are smaller than the library's own rule for the route asks)
    _varlen_case(fa, oracle, dev, B, H, max(q_lens), max(k_lens), 128, q_lens, k_lens, False, seed=4100 + H)
    # against the 128-row kernel alone: the short sequences' bits are the same, the long ones agree to the gate
    rng = np.random.default_rng(4200 + H)
    Q, K, V = [torch.from_numpy(rng.standard_normal((B, H, n, 128), dtype=np.float32).astype(np.float16)).to(dev)
               for n in (max(q_lens), max(k_lens), max(k_lens))]
    ql = torch.tensor(q_lens, dtype=torch.int32, device=dev)
    kl = torch.tensor(k_lens, dtype=torch.int32, device=dev)
    both = fa.attention_varlen(Q, K, V, ql, kl)
    fa_env(FA_VARLEN_W64="0")
    one = fa.attention_varlen(Q, K, V, ql, kl)
    fa_env(FA_VARLEN_W64=None)
    auto = fa.attention_varlen(Q, K, V, ql, kl)
    torch.cuda.synchronize()
    assert torch.equal(auto, one)  # (left to itself the library keeps a batch this small on the 128-row kernel)
    tuned = _on_tuned_kernel(q_lens, k_lens, False)
    assert tuned == ([True, False, True, True, False, False, False, True] if H == 9 else [True] * 4)
    for b in range(B):
        if not tuned[b]:
            assert torch.equal(both[b], one[b]), f"sequence {b}"
        else:
            d = (both[b, :, :q_lens[b]].float() - one[b, :, :q_lens[b]].float()).abs().max().item()
            assert 0 < d <= 2e-3, f"sequence {b}: {d}"  # (another kernel: some last bits differ)


@pytest.mark.parametrize("H", [8, 3])
def test_padded_batch_long_and_short_sequences_causal(fa, oracle, dev, H, fa_env):
    """The same under the causal mask: a sequence runs on the tuned kernel's list form when it has as many rows as
    keys and at least 1665 of them with a last 256-row block more than half full, 2305 otherwise; lengths on both
    sides of each rule, an empty and a short sequence, and (second part) sequences whose rows and keys differ,
    which stay on the 128-row kernel whatever their length."""
    import torch

    lens = [2305, 1665, 1664, 2100, 0, 2304, 700] if H == 3 else [1665, 2400, 1664, 1921]
    B = len(lens)
    fa_env(FA_VARLEN_W64="1")
    _varlen_case(fa, oracle, dev, B, H, max(lens), max(lens), 128, lens, lens, True, seed=4300 + H)
    rng = np.random.default_rng(4400 + H)
    L = max(lens)
    Q, K, V = [torch.from_numpy(rng.standard_normal((B, H, L, 128), dtype=np.float32).astype(np.float16)).to(dev)
               for _ in range(3)]
    k_lens = list(lens)
    k_lens[0], k_lens[1] = lens[0] - 1, min(L, lens[1] + 300)  # rows != keys: not the tuned kernel's
    ql = torch.tensor(lens, dtype=torch.int32, device=dev)
    kl = torch.tensor(k_lens, dtype=torch.int32, device=dev)
    both = fa.attention_varlen(Q, K, V, ql, kl, causal=True)
    fa_env(FA_VARLEN_W64="0")
    one = fa.attention_varlen(Q, K, V, ql, kl, causal=True)
    fa_env(FA_VARLEN_W64=None)
    torch.cuda.synchronize()
    tuned_all = _on_tuned_kernel(lens, k_lens, True)
    assert tuned_all == ([False, False, False, False, False, True, False] if H == 3 else [False, True, True, True])
    for b in range(B):
        n = lens[b]
        tuned = tuned_all[b]
        if not tuned:
            assert torch.equal(both[b], one[b]), f"sequence {b}"
        else:
            d = (both[b, :, :n].float() - one[b, :, :n].float()).abs().max().item()
            assert 0 < d <= 2e-3, f"sequence {b}: {d}"  # (another kernel: some last bits differ)


def test_padded_batch_of_more_sequences_than_the_planner_sorts(fa, oracle, dev, fa_env):
    """520 sequences (the list's planner sorts up to 512 and scans 256 groups at a time): without the mask the list is
    built in the sequences' own order; against the 128-row kernel alone on every sequence and against the oracle on a
    few.  With the mask such a batch stays on the 128-row kernel (same bits)."""
    import torch

    B, H, L, D = 520, 1, 1216, 128
    rng = np.random.default_rng(4500)
    lens = [int(x) for x in rng.integers(1000, L + 1, B)]
    lens[7], lens[300], lens[519] = L, 1100, 1099
    Q, K, V = [rng.standard_normal((B, H, L, D), dtype=np.float32).astype(np.float16) for _ in range(3)]
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    ql = torch.tensor(lens, dtype=torch.int32, device=dev)
    fa_env(FA_VARLEN_W64="1")  # (the batch is near the library's own limit for the route: not what this test is about)
    both = fa.attention_varlen(q, k, v, ql, ql)
    assert torch.equal(both, fa.attention_varlen(q, k, v, ql, ql))  # (fixed item lists: the same bits every time)
    both_causal = fa.attention_varlen(q, k, v, ql, ql, causal=True)
    fa_env(FA_VARLEN_W64="0")
    one = fa.attention_varlen(q, k, v, ql, ql)
    one_causal = fa.attention_varlen(q, k, v, ql, ql, causal=True)
    fa_env(FA_VARLEN_W64=None)
    torch.cuda.synchronize()
    assert torch.equal(both_causal, one_causal)
    tuned = _on_tuned_kernel(lens, lens, False)
    assert sum(tuned) > 100 and not all(tuned)
    worst = 0.0
    for b in range(B):
        n = lens[b]
        assert (both[b, :, n:] == 0).all()  # (attention_varlen's own output starts as zeros)
        if tuned[b]:
            worst = max(worst, (both[b, :, :n].float() - one[b, :, :n].float()).abs().max().item())
        else:
            assert torch.equal(both[b], one[b]), f"sequence {b}"
    assert 0 < worst <= 2e-3
    for b in (0, 7, 300, 519):
        n = lens[b]
        Qs, Ks, Vs = [np.ascontiguousarray(a[b:b + 1, :, :n]) for a in (Q, K, V)]
        gate(oracle, oracle.attention_cpu(Qs, Ks, Vs), both[b:b + 1, :, :n].cpu().numpy(), f"sequence {b} ({n} rows)")


def test_padded_batch_masked_with_more_block_indices_than_one_scan(fa, dev, fa_env):
    """One masked sequence of 66 001 rows (258 block indices: the planner's groups, scanned 256 at a time) beside a
    short one: per sequence the bits of fa_forward_general on that sequence alone (the same kernel forms, the items
    in another order)."""
    import torch

    H, D, L = 1, 128, 66001
    lens = [L, 3000]
    g = torch.Generator(device=dev).manual_seed(46)
    q, k, v = [torch.randn((2, H, L, D), generator=g, device=dev).to(torch.float16) for _ in range(3)]
    ql = torch.tensor(lens, dtype=torch.int32, device=dev)
    fa_env(FA_VARLEN_W64=None)
    o = fa.attention_varlen(q, k, v, ql, ql, causal=True)
    torch.cuda.synchronize()
    assert _on_tuned_kernel(lens, lens, True) == [True, True]
    for b, n in enumerate(lens):
        ref = fa.attention_general(q[b:b + 1, :, :n].contiguous(), k[b:b + 1, :, :n].contiguous(),
                                   v[b:b + 1, :, :n].contiguous(), causal=True)
        d = (o[b:b + 1, :, :n].float() - ref.float()).abs().max().item()
        assert d <= 1e-3, f"sequence {b}: {d}"
        assert (o[b, :, n:] == 0).all()


def test_padded_batch_under_stream_capture_keeps_the_128_row_kernel(fa, dev):
    """The list of the tuned route comes from the stream's memory pool; a capturing stream gets the 128-row kernel
    instead, and the replayed graph gives the bits of FA_VARLEN_W64=0."""
    import torch

    B, H, L, D = 3, 2, 1300, 128
    g = torch.Generator(device=dev).manual_seed(47)
    q, k, v = [torch.randn((B, H, L, D), generator=g, device=dev).to(torch.float16) for _ in range(3)]
    ql = torch.tensor([1300, 1200, 400], dtype=torch.int32, device=dev)
    out = torch.zeros_like(q)
    os.environ["FA_VARLEN_W64"] = "1"  # (the route wherever it can run: the batch is small)
    fa.reload_env()
    try:
        fa.attention_varlen(q, k, v, ql, ql, out=out)  # (first call outside the capture)
        torch.cuda.synchronize()
        eager = out.clone()
        out.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fa.attention_varlen(q, k, v, ql, ql, out=out)
        graph.replay()
        torch.cuda.synchronize()
        os.environ["FA_VARLEN_W64"] = "0"
        fa.reload_env()
        plain = fa.attention_varlen(q, k, v, ql, ql)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("FA_VARLEN_W64", None)
        fa.reload_env()
    assert torch.equal(out, plain)
    d = (eager.float() - plain.float()).abs().max().item()
    assert 0 < d <= 2e-3  # (outside the capture the tuned kernel ran: some last bits differ)


def test_padded_batch_with_too_little_work_for_the_list_form(fa, dev, fa_env):
    """Padded lengths that would fill the chip (8 x 16 heads x 4096 rows) but sequences that do not: the planner finds
    less than its minimum of rows x keys in the sequences the tuned kernel would take and leaves the whole batch to
    the 128-row kernel (same bits as with the route switched off); forced, the tuned kernel runs."""
    import torch

    B, H, L, D = 8, 16, 4096, 128
    lens = [1200, 1150, 300, 64, 1100, 90, 7, 1300]
    g = torch.Generator(device=dev).manual_seed(48)
    q, k, v = [torch.randn((B, H, L, D), generator=g, device=dev).to(torch.float16) for _ in range(3)]
    ql = torch.tensor(lens, dtype=torch.int32, device=dev)
    fa_env(FA_VARLEN_W64=None)
    auto = fa.attention_varlen(q, k, v, ql, ql)
    fa_env(FA_VARLEN_W64="0")
    one = fa.attention_varlen(q, k, v, ql, ql)
    fa_env(FA_VARLEN_W64="1")
    forced = fa.attention_varlen(q, k, v, ql, ql)
    fa_env(FA_VARLEN_W64=None)
    torch.cuda.synchronize()
    assert torch.equal(auto, one)
    d = (forced.float() - one.float()).abs().max().item()
    assert 0 < d <= 2e-3


def test_padded_batch_long_sequences_without_query_lengths(fa, oracle, dev):
    """k_len alone on long padded lengths: every query row of the padded tensor against its sequence's keys
    (sequences with at least 1100 keys on the tuned kernel, the others on the 128-row kernel)."""
    import torch

    B, H, L, D = 3, 2, 1300, 128
    Q, K, V = normal_fp16(56, (B, H, L, D))
    k_lens = [1300, 1100, 37]
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o = fa.attention_varlen(q, k, v, None, torch.tensor(k_lens, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    for b in range(B):
        st = oracle.BlockState(1, H, L, D)
        oracle.block_update(st, np.ascontiguousarray(Q[b:b + 1]), np.ascontiguousarray(K[b:b + 1, :, :k_lens[b]]),
                            np.ascontiguousarray(V[b:b + 1, :, :k_lens[b]]))
        gate(oracle, oracle.block_finalize(st), o[b:b + 1].cpu().numpy(), f"sequence {b}")


@pytest.mark.parametrize("Lq,Lk", [(200, 328), (320, 192)])
def test_causal_mask_with_different_lengths_and_no_length_arrays(fa, oracle, dev, Lq, Lk):
    """fa_forward_varlen under the causal mask with Lq != Lk: accepted with both length arrays NULL exactly as with
    arrays that hold the padded lengths (same kernel, same bits), and equal to the oracle's blockwise form with the
    mask on positions from 0 (key j is visible to query i iff j <= i, testing.cu:74-77)."""
    import torch

    B, H, D = 2, 3, 128
    rng = np.random.default_rng(31)
    Q = rng.standard_normal((B, H, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o_null = fa.attention_varlen(q, k, v, None, None, causal=True)
    ql = torch.full((B,), Lq, dtype=torch.int32, device=dev)
    kl = torch.full((B,), Lk, dtype=torch.int32, device=dev)
    o_lens = fa.attention_varlen(q, k, v, ql, kl, causal=True)
    torch.cuda.synchronize()
    assert torch.equal(o_null, o_lens)
    st = oracle.BlockState(B, H, Lq, D)
    oracle.block_update(st, Q, K, V, causal=True)
    gate(oracle, oracle.block_finalize(st), o_null.cpu().numpy(), f"causal Lq={Lq} Lk={Lk}")


# ------------------------------------------------------------------------------------------------
# head dims 64 and 256 in whole tiles: fa_fwd_hd_kernel
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,Lq,Lk", [
    (256, 64, 64), (256, 128, 192), (256, 192, 64), (256, 320, 448), (256, 1024, 1024), (256, 512, 2048),
    (64, 64, 128), (64, 128, 256), (64, 192, 384), (64, 320, 128), (64, 1024, 1024), (64, 512, 2048),
])
def test_other_head_dims_whole_tiles(fa, oracle, dev, D, Lq, Lk, fa_env):
    """D = 64 / 256 with whole tiles run on fa_fwd_hd_kernel<D> (one to many K/V tiles per item, partly filled
    last query blocks, Lq != Lk), inputs followed by NaN and the output by a sentinel; whole tensors against the
    oracle and, within rounding, against the general kernel."""
    import torch

    fa_env(FA_HD=None)
    assert fa.lib().fa_kernel_name_general(Lq, Lk, D, 0) == f"fa::fa_fwd_hd_kernel<{D}>".encode()
    rng = np.random.default_rng(D + Lq + Lk)
    Q = rng.standard_normal((2, 3, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((2, 3, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((2, 3, Lk, D), dtype=np.float32).astype(np.float16)
    O = run_general(fa, dev, Q, K, V)
    st = oracle.BlockState(2, 3, Lq, D)
    oracle.block_update(st, Q, K, V)
    gate(oracle, oracle.block_finalize(st), O, f"D={D} Lq={Lq} Lk={Lk}")
    fa_env(FA_HD="0")
    assert fa.lib().fa_kernel_name_general(Lq, Lk, D, 0) == f"fa::fa_fwd_general_kernel<{D}>".encode()
    O2 = run_general(fa, dev, Q, K, V)
    assert np.abs(O.astype(np.float32) - O2.astype(np.float32)).max() <= 1e-3


@pytest.mark.parametrize("D", [64, 256])
def test_other_head_dims_rescale_and_repeat(fa, oracle, dev, D, fa_env):
    """A dominating key in every 128-key stretch, growing along the sequence: the running reference moves (the
    accumulator rescale runs) on most tiles; several heads so that many workgroups run; results repeat bit for bit."""
    import torch

    fa_env(FA_HD=None)
    rng = np.random.default_rng(D)
    shape = (1, 20, 1024, D)
    Q = (rng.standard_normal(shape) * 1.5).astype(np.float16)
    K = (rng.standard_normal(shape) * 1.5).astype(np.float16)
    V = (rng.standard_normal(shape) * 0.5).astype(np.float16)  # (|O| stays below 2, where fp16 neighbours are 1e-3 apart)
    for t in range(8):
        for row in (3, 40, 70, 200, 255, 300, 511, 1000):
            K[:, :, 128 * t + (row % 128), :] = (Q[:, :, row, :].astype(np.float32) * (0.1 + 0.06 * t)).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o1 = fa.attention_general(q, k, v).clone()
    o2 = fa.attention_general(q, k, v)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    gate(oracle, oracle.attention_cpu(Q, K, V), o1.cpu().numpy(), f"D={D} spiked")


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("Lq,Lk,causal", [
    (1, 1, False), (1, 1, True), (5, 5, True), (33, 33, True), (64, 64, True), (65, 65, True), (128, 128, True),
    (200, 200, True), (513, 513, True), (1000, 1000, True), (1, 500, False), (100, 64, False), (130, 1000, False),
    (512, 77, False), (257, 3, False), (300, 31, False), (64, 129, False), (129, 2048, False),
])
def test_other_head_dims_masked_form(fa, oracle, dev, D, Lq, Lk, causal, fa_env):
    """D = 64 / 128 / 256 with the causal mask, lengths that are not whole tiles (down to one row, fewer keys than a tile) and
    Lq != Lk run on fa_fwd_hd_masked_kernel<D>: whole tensors against the oracle, NaN behind the inputs and a sentinel
    behind the output (run_general), and within rounding the general kernel's result."""
    fa_env(FA_HD=None, FA_MASKED_W64="0")  # (D = 128: the one-wave-per-SIMD kernel's masked forms would take some)
    flags = 1 if causal else 0
    assert fa.lib().fa_kernel_name_general(Lq, Lk, D, flags) == f"fa::fa_fwd_hd_masked_kernel<{D}>".encode()
    rng = np.random.default_rng(7 * D + Lq + 3 * Lk)
    Q = rng.standard_normal((2, 3, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((2, 3, Lk, D), dtype=np.float32).astype(np.float16)
    V = (rng.standard_normal((2, 3, Lk, D), dtype=np.float32) * 0.5).astype(np.float16)
    O = run_general(fa, dev, Q, K, V, causal=causal)
    if Lq == Lk:
        ref = oracle.attention_cpu(Q, K, V, causal=causal)
    else:
        st = oracle.BlockState(2, 3, Lq, D)
        oracle.block_update(st, Q, K, V)
        ref = oracle.block_finalize(st)
    gate(oracle, ref, O, f"D={D} Lq={Lq} Lk={Lk} causal={causal}")
    fa_env(FA_HD="1")  # (1: only the unmasked whole-tile shapes leave the general kernel)
    assert fa.lib().fa_kernel_name_general(Lq, Lk, D, flags) == f"fa::fa_fwd_general_kernel<{D}>".encode()
    O2 = run_general(fa, dev, Q, K, V, causal=causal)
    assert np.abs(O.astype(np.float32) - O2.astype(np.float32)).max() <= 1e-3


@pytest.mark.parametrize("D", [64, 128, 256])
def test_other_head_dims_causal_many_blocks(fa, oracle, dev, D, fa_env):
    """Causal, several heads and query blocks (the long blocks are handed out first), a ragged length on top:
    sampled rows against the oracle over the whole sequence."""
    import torch

    fa_env(FA_HD=None, FA_MASKED_W64="0")
    L = 2100
    Q, K, V = normal_fp16(500 + D, (2, 5, L, D))
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o = fa.attention_general(q, k, v, causal=True)
    torch.cuda.synchronize()
    rows = np.unique(np.concatenate([np.arange(0, L, 97), [L - 1, 127, 128, 129, 2047, 2048]])).astype(np.int32)
    ref = oracle.attention_cpu_rows(Q, K, V, rows, causal=True)
    m = oracle.error_metrics(ref, o.cpu().numpy()[:, :, rows])
    # (the first rows of a causal sequence average few values: |O| reaches 2 to 4 there, where fp16 neighbours lie
    # 2e-3 apart, so the absolute bound scales with |O| above 1)
    assert m["max_abs"] <= MAX_ABS_TOL * max(1.0, float(np.abs(ref).max())) and m["global_rel"] <= REL_L2_TOL, m


@pytest.mark.parametrize("L", [320, 384])
def test_short_self_attention_takes_the_128_row_kernel(fa, oracle, dev, L, fa_env):
    """Lq = Lk = 320 / 384 in whole tiles: the drop-in entry point's launcher hands these two lengths to the 128-row
    kernel (the second 256-row block would be at most half full); FA_HD=1 and FA_W64 keep them on the 256-row kernels,
    as do calls with Lq != Lk."""
    import torch

    fa_env(FA_HD=None, FA_W64=None)
    name = fa.lib().fa_kernel_name
    assert name(L, L) == b"fa::fa_fwd_hd_masked_kernel<128>" == fa.lib().fa_kernel_name_general(L, L, 128, 0)
    assert name(L, L + 64) == b"fa::fa_fwd_w64_kernel" and name(448, 448) == b"fa::fa_fwd_w64_kernel"
    assert fa.lib().fa_split_ways(2, L, L) == 1
    Q, K, V = normal_fp16(7000 + L, (2, 3, L, 128))
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o = torch.full_like(q, float("nan"))
    fa.attention(q, k, v, out=o)
    torch.cuda.synchronize()
    ref = oracle.attention_cpu(Q, K, V)
    gate(oracle, ref, o.cpu().numpy(), f"L={L} on the 128-row kernel")
    fa_env(FA_HD="1")
    assert name(L, L) == b"fa::fa_fwd_w64_kernel"
    o2 = fa.attention(q, k, v)
    torch.cuda.synchronize()
    gate(oracle, ref, o2.cpu().numpy(), f"L={L} on the 256-row kernel")
    fa_env(FA_HD=None, FA_W64="1")
    assert name(L, L) == b"fa::fa_fwd_w64_kernel"


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("D", [64, 256])
def test_blockwise_other_head_dims_on_the_pipelined_kernel(fa, oracle, dev, D, causal, fa_env):
    """fa_forward_block_general at D = 64 / 256 runs on fa_fwd_hd_state_kernel<D> (FA_HD=0: the general kernel): a
    query chunk that starts in the middle of a sequence (q_pos0 = 500) folded over four key chunks of several
    tiles each in shuffled order: one wholly before it, one that its diagonal crosses, one whose first keys only
    the last rows see, one wholly behind it (with the mask: the carried state passes through unchanged).  The
    output against the oracle over the whole sequence, on this kernel and on the general kernel."""
    import torch

    fa_env(FA_HD=None)
    BH, L = 3, 1200
    Q, K, V = normal_fp16(5000 + D, (1, BH, L, D))
    q0, nq = 500, 300
    cuts = [0, 448, 700, 1000, 1200]
    order = [3, 1, 0, 2] if causal else [2, 0, 3, 1]
    qc = torch.from_numpy(np.ascontiguousarray(Q[:, :, q0:q0 + nq])).to(dev)
    name = fa.lib().fa_kernel_name_block_general
    assert name(nq, 252, D, int(causal), q0, 448) == f"fa::fa_fwd_hd_state_kernel<{D}>".encode()
    outs = {}
    for hd in (None, "0"):
        fa_env(FA_HD=hd)
        st = fa.BlockState(BH, nq, D, dev)
        out = None
        for n, c in enumerate(order):
            a, b = cuts[c], cuts[c + 1]
            kc, vc = [torch.from_numpy(np.ascontiguousarray(x[:, :, a:b])).to(dev) for x in (K, V)]
            out = fa.attention_block_general(qc, kc, vc, st, last=(n == len(order) - 1), causal=causal, q_pos0=q0, k_pos0=a)
        torch.cuda.synchronize()
        outs[hd] = out.cpu().numpy()
    if causal:
        ref = oracle.attention_cpu(Q, K, V, causal=True)[:, :, q0:q0 + nq]
    else:
        ref_st = oracle.BlockState(1, BH, nq, D)
        oracle.block_update(ref_st, np.ascontiguousarray(Q[:, :, q0:q0 + nq]), K, V)
        ref = oracle.block_finalize(ref_st)
    gate(oracle, ref, outs[None], f"blockwise D={D} causal={causal} on the pipelined kernel")
    gate(oracle, ref, outs["0"], f"blockwise D={D} causal={causal} on the general kernel")


@pytest.mark.parametrize("D", [64, 256])
def test_blockwise_other_head_dims_state_values(fa, oracle, dev, D):
    """The carried state itself after a first block (no mask): row_max is the true row maximum in natural-log units,
    row_sum and o_acc the sums relative to it, as the oracle's block_update leaves them."""
    import torch

    BH, Lq, Lk = 2, 200, 333
    Q = normal_fp16(6000 + D, (1, BH, Lq, D))[0]
    K, V = normal_fp16(6100 + D, (1, BH, Lk, D))[:2]
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    st = fa.BlockState(BH, Lq, D, dev)
    fa.attention_block_general(q, k, v, st, last=False)
    torch.cuda.synchronize()
    ref = oracle.BlockState(1, BH, Lq, D)
    oracle.block_update(ref, Q, K, V)
    m_ref, l_ref, o_ref = [np.asarray(x, dtype=np.float64).reshape(BH * Lq, -1) for x in (ref.mx, ref.sm, ref.acc)]
    m, l, o = [x.cpu().numpy().astype(np.float64).reshape(BH * Lq, -1) for x in (st.row_max, st.row_sum, st.o_acc)]
    assert np.abs(m - m_ref).max() <= 1e-5 * max(1.0, np.abs(m_ref).max())
    assert np.abs(l / l_ref - 1).max() <= 2e-3
    assert np.abs(o - o_ref).max() <= 2e-3 * max(1.0, np.abs(o_ref).max())


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("BH,Lq,Lk", [(4, 1, 5000), (2, 16, 8192), (3, 100, 4097), (16, 1, 20000)])
def test_few_rows_against_many_keys_divide_the_keys(fa, oracle, dev, D, BH, Lq, Lk, fa_env):
    """A handful of query rows against a long sequence (a decoding step): the launcher divides every head's keys into
    ranges of whole tiles (the last one shorter), one workgroup each, and folds the partial states.  Against the
    oracle, and against the undivided launch (FA_SPLIT=1) under the row-relative bound."""
    import torch

    fa_env(FA_HD=None, FA_SPLIT=None)
    assert fa.lib().fa_split_ways_general(BH, Lq, Lk, D, 0) > 1
    assert fa.lib().fa_split_ways_general(2048, Lq, Lk, D, 0) == 1  # (enough heads to fill the chip)
    rng = np.random.default_rng(8000 + D + Lk)
    Q = rng.standard_normal((1, BH, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((1, BH, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((1, BH, Lk, D), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o = torch.full_like(q, float("nan"))
    fa.attention_general(q, k, v, out=o)
    torch.cuda.synchronize()
    st = oracle.BlockState(1, BH, Lq, D)
    oracle.block_update(st, Q, K, V)
    ref = oracle.block_finalize(st)
    gate(oracle, ref, o.cpu().numpy(), f"divided keys D={D} {BH}x{Lq}x{Lk}")
    fa_env(FA_SPLIT="1")
    o1 = fa.attention_general(q, k, v)
    torch.cuda.synchronize()
    gate(oracle, ref, o1.cpu().numpy(), f"undivided D={D} {BH}x{Lq}x{Lk}")
    a, b = o.float(), o1.float()
    row = torch.maximum(a.abs(), b.abs()).amax(dim=-1, keepdim=True)
    assert bool(((a - b).abs() <= 2.0 ** -9 * row + 1e-6).all())
    # a forced count of ranges that does not divide the tiles evenly
    fa_env(FA_SPLIT="7")
    o7 = fa.attention_general(q, k, v)
    torch.cuda.synchronize()
    gate(oracle, ref, o7.cpu().numpy(), f"seven ranges D={D} {BH}x{Lq}x{Lk}")


@pytest.mark.parametrize("D", [64, 128, 256])
def test_padded_batch_of_few_rows_divides_the_keys(fa, oracle, dev, D, fa_env):
    """A padded decoding batch (fa_forward_varlen: one to sixteen query rows per sequence, key lengths from 0 to the
    padded 6000) takes the key split as well: ranges behind a sequence's last key hold the empty state, rows behind
    its last query row are neither merged nor written."""
    fa_env(FA_HD=None, FA_SPLIT=None)
    B, H, Lq, Lk = 4, 2, 16, 6000
    assert fa.lib().fa_split_ways_general(B * H, Lq, Lk, D, 0) > 1
    _varlen_case(fa, oracle, dev, B, H, Lq, Lk, D, [16, 3, 0, 9], [6000, 100, 5000, 0], False, seed=9100 + D)
    _varlen_case(fa, oracle, dev, B, H, 1, Lk, D, [1, 1, 1, 1], [1, 2049, 5999, 6000], False, seed=9200 + D)


def test_divided_keys_under_stream_capture(fa, oracle, dev, fa_env):
    """fa_forward_general on a shape that divides its keys, captured in a graph: nothing is allocated inside a capture,
    so the launch is the undivided one there; it replays to the undivided launch's bits and within the gate."""
    import torch

    fa_env(FA_HD=None, FA_SPLIT=None)
    BH, Lq, Lk, D = 4, 8, 6000, 128
    assert fa.lib().fa_split_ways_general(BH, Lq, Lk, D, 0) > 1
    rng = np.random.default_rng(31)
    Q = rng.standard_normal((1, BH, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((1, BH, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((1, BH, Lk, D), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    eager = fa.attention_general(q, k, v)
    torch.cuda.synchronize()
    o = torch.zeros_like(q)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fa.attention_general(q, k, v, out=o)  # (on the capturing stream)
    graph.replay()
    torch.cuda.synchronize()
    fa_env(FA_SPLIT="1")
    undivided = fa.attention_general(q, k, v)
    torch.cuda.synchronize()
    assert torch.equal(o, undivided)
    st = oracle.BlockState(1, BH, Lq, D)
    oracle.block_update(st, Q, K, V)
    ref = oracle.block_finalize(st)
    gate(oracle, ref, o.cpu().numpy(), "captured")
    gate(oracle, ref, eager.cpu().numpy(), "eager, divided")
