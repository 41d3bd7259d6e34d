"""A second, independent reference for everything the reference's own fixtures do not reach.

The reference holds outputs only for non-causal D = 128 attention (tests/golden, from its notebook's NumPy function).
The causal mask (testing.cu:74-77, normal_attention.cu:50: key j is hidden from query i when j > i), other head
dims, ragged lengths and Lq != Lk are checked elsewhere against this repo's restatement of attention_cpu.  Here the
same cases are checked against an implementation nobody in this repo wrote: PyTorch's
scaled_dot_product_attention in fp32 (its causal mask is the same top-left rule).

CPU tier: the oracle against PyTorch on the CPU.  GPU tier: the HIP library against PyTorch's math backend on the
GPU (fp32 inputs converted from the same fp16 values), at sizes the CPU oracle would take minutes for.

Tolerance: attention_cpu rounds P to fp16 before P.V (testing.cu:107) and PyTorch's fp32 path does not, so the
two differ at the fp16-P level (a few 1e-4 on N(0,1) data); the bound is 1e-3 absolute, rel-L2 5e-4: half the
parity gate the HIP path is held to against the oracle.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import MAX_ABS_TOL, REL_L2_TOL, normal_fp16


def sdpa_fp32(Q, K, V, causal, device="cpu"):
    q, k, v = [torch.from_numpy(a.astype(np.float32)).to(device) for a in (Q, K, V)]
    with torch.no_grad():
        if device != "cpu":
            from torch.nn.attention import SDPBackend, sdpa_kernel

            with sdpa_kernel([SDPBackend.MATH]):
                o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        else:
            o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
    return o.cpu().numpy()


CPU_CASES = [
    # (B, H, Lq, Lk, D, causal)
    (1, 2, 192, 192, 128, True),
    (1, 2, 200, 200, 128, True),    # ragged tail under the mask
    (2, 1, 100, 100, 64, True),
    (1, 2, 130, 130, 256, True),
    (1, 2, 64, 300, 128, False),    # Lq != Lk
    (1, 1, 300, 77, 64, False),
    (1, 2, 96, 160, 128, True),     # the mask with more keys than rows: top-left rule, keys beyond Lq never seen
    (1, 1, 1, 513, 128, False),     # one query row
    (1, 2, 257, 257, 256, False),
]


@pytest.mark.parametrize("B,H,Lq,Lk,D,causal", CPU_CASES)
def test_oracle_against_pytorch_fp32(oracle, B, H, Lq, Lk, D, causal):
    rng = np.random.default_rng(1000 + Lq + Lk + D)
    Q = rng.standard_normal((B, H, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    if Lq == Lk:
        ours = oracle.attention_cpu(Q, K, V, causal=causal)
    else:  # the oracle's blockwise form takes any pair of lengths: one block, folded and finalised
        st = oracle.BlockState(B, H, Lq, D)
        oracle.block_update(st, Q, K, V, causal=causal)
        ours = oracle.block_finalize(st)
    ref = sdpa_fp32(Q, K, V, causal)
    m = oracle.error_metrics(ref, ours)
    assert m["max_abs"] < 1e-3 and m["global_rel"] < 5e-4, m


def test_oracle_causal_first_rows_against_pytorch(oracle):
    """Row 0 sees key 0 only, row 1 keys 0-1: where an off-by-one in the mask would show first."""
    Q, K, V = normal_fp16(77, (1, 1, 128, 128))
    ours = oracle.attention_cpu(Q, K, V, causal=True)
    ref = sdpa_fp32(Q, K, V, True)
    assert np.allclose(ours[0, 0, 0], V[0, 0, 0].astype(np.float32), atol=1e-6)
    assert np.abs(ours[0, 0, :4] - ref[0, 0, :4]).max() < 1e-3


GPU_CASES = [
    # (B, H, Lq, Lk, D, causal): every kernel family behind fa_forward_general
    (2, 16, 2048, 2048, 128, True),    # causal tuned kernel
    (2, 16, 2047, 2047, 128, True),    # ragged causal tuned kernel
    (4, 16, 1023, 1023, 128, False),   # ragged tuned kernel or the 128-row kernel, by the batch
    (2, 8, 700, 700, 128, True),       # 128-row masked kernel
    (2, 8, 512, 2048, 128, False),     # Lq != Lk in whole tiles
    (2, 8, 333, 1500, 128, False),     # Lq != Lk, ragged
    (2, 8, 2048, 2048, 64, True),
    (2, 8, 2047, 2047, 64, False),
    (2, 8, 2048, 2048, 256, True),
    (2, 8, 1001, 1001, 256, False),
    (16, 1, 1, 16384, 128, False),     # decoding shape, keys divided over the chip
    (8, 2, 5, 8192, 64, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,Lq,Lk,D,causal", GPU_CASES)
def test_hip_path_against_pytorch_math_backend(fa, oracle, B, H, Lq, Lk, D, causal):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2000 + Lq + Lk + D)
    Q = rng.standard_normal((B, H, Lq, D), dtype=np.float32).astype(np.float16)
    K = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    V = rng.standard_normal((B, H, Lk, D), dtype=np.float32).astype(np.float16)
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    o = fa.attention_general(q, k, v, causal=causal)
    torch.cuda.synchronize()
    ref = sdpa_fp32(Q, K, V, causal, device=dev)
    m = oracle.error_metrics(ref, o.cpu().numpy())
    # the HIP path rounds P and O to fp16 (as attention_cpu's callers do); the parity gate itself applies
    assert m["max_abs"] <= MAX_ABS_TOL and m["global_rel"] <= REL_L2_TOL, m


@pytest.mark.gpu
@pytest.mark.parametrize("causal", [False, True])
def test_padded_batch_against_pytorch_per_sequence(fa, oracle, causal):
    """fa_forward_varlen against PyTorch run on each sequence's valid part by itself."""
    dev = torch.device("cuda:0")
    B, H, L, D = 6, 4, 1536, 128
    lens = [1536, 1100, 777, 64, 1, 1301]
    rng = np.random.default_rng(31)
    Q, K, V = [rng.standard_normal((B, H, L, D), dtype=np.float32).astype(np.float16) for _ in range(3)]
    q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    o = fa.attention_varlen(q, k, v, q_len=ln, k_len=ln, causal=causal)
    torch.cuda.synchronize()
    got = o.cpu().numpy()
    for b, n in enumerate(lens):
        ref = sdpa_fp32(Q[b:b + 1, :, :n], K[b:b + 1, :, :n], V[b:b + 1, :, :n], causal, device=dev)
        m = oracle.error_metrics(ref, got[b:b + 1, :, :n])
        assert m["max_abs"] <= MAX_ABS_TOL and m["global_rel"] <= REL_L2_TOL, (b, n, m)


# ------------------------------------------------------------------------------------------------
# Properties that hold at any size: checked at the benchmark's full shapes, where the CPU oracle only reaches
# slices.
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("B,L", [(32, 512), (1, 16384)])
def test_keys_may_come_in_any_order(fa, B, L):
    """softmax(Q K^T) V does not depend on the order of the (key, value) pairs: the same permutation applied to the
    rows of K and V of every head changes the order of summation only."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    q, k, v = [torch.randn((B, 16, L, 128), generator=g, device=dev, dtype=torch.float32).to(torch.float16) for _ in range(3)]
    perm = torch.randperm(L, generator=g, device=dev)
    o1 = fa.attention(q, k, v)
    o2 = fa.attention(q, k[:, :, perm].contiguous(), v[:, :, perm].contiguous())
    torch.cuda.synchronize()
    d = (o1.float() - o2.float()).abs()
    # two fp16 roundings of nearly the same fp32 value: an ulp of the output at most, far inside the parity gate
    assert d.max().item() <= 2e-3 and d.mean().item() < 2e-5, (d.max().item(), d.mean().item())


@pytest.mark.gpu
@pytest.mark.parametrize("B,L", [(8, 2048), (1, 16384)])
def test_causal_rows_do_not_see_later_keys(fa, B, L):
    """With the mask of testing.cu:74-77 a query row's output is a function of the keys up to its own position:
    overwriting every later key and value with large values leaves those rows as they were.  (Finite
    values: a masked key's weight is exactly 0, and 0 x NaN is NaN in attention_cpu's P.V loop as well,
    testing.cu:114-122.)"""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(6)
    q, k, v = [torch.randn((B, 16, L, 128), generator=g, device=dev, dtype=torch.float32).to(torch.float16) for _ in range(3)]
    o1 = fa.attention_general(q, k, v, causal=True)
    for cut in (1, 63, 64, L // 2 + 17, L - 1):
        k2, v2 = k.clone(), v.clone()
        k2[:, :, cut:] = 30.0
        v2[:, :, cut:] = 1000.0
        o2 = fa.attention_general(q, k2, v2, causal=True)
        torch.cuda.synchronize()
        # Bit for bit up to the last 32-row sub-block that lies wholly before the cut.  Rows that share a sub-block
        # with rows behind the cut keep their values within an fp16 ulp: the kernel moves a row's reference point
        # (the deferred rescale) together with its sub-block's, and P is rounded to fp16 relative to it.
        whole = cut // 32 * 32
        assert torch.equal(o1[:, :, :whole], o2[:, :, :whole]), cut
        d = (o1[:, :, whole:cut].float() - o2[:, :, whole:cut].float()).abs()
        assert d.numel() == 0 or d.max().item() <= 4e-3, (cut, d.max().item())
        assert torch.isfinite(o2[:, :, :cut].float()).all()
