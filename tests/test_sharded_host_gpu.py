"""GPU tier: fa_forward_host_half_sharded, the host-pointer entry point over several GPUs from one process.

A one-GPU box exercises everything but the second device: a GPU may be named more than once, so ranges of the
flattened B*H axis run side by side on streams and host threads of their own, with uneven and empty ranges.  The
results must have the single-device entry point's bits (the kernels see B*H only, and a range is a contiguous
slab of every tensor)."""
import ctypes as ct

import numpy as np
import pytest
import torch

from conftest import MAX_ABS_TOL, REL_L2_TOL, normal_fp16

pytestmark = pytest.mark.gpu


def test_one_range_has_the_single_device_bits(fa, oracle):
    Q, K, V = normal_fp16(41, (2, 3, 256, 128))
    want, _ = fa.attention_host_half(Q, K, V)
    got, ms = fa.attention_host_half_sharded(Q, K, V, devices=[0])
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    assert len(ms) == 1 and ms[0] > 0
    m = oracle.error_metrics(oracle.attention_cpu(Q, K, V), got)
    assert m["max_abs"] <= MAX_ABS_TOL and m["global_rel"] <= REL_L2_TOL, m


@pytest.mark.parametrize("B,H,L,ranges", [(1, 5, 256, 3), (2, 8, 512, 4), (1, 7, 320, 2), (3, 3, 128, 9)])
def test_uneven_ranges_on_one_gpu(fa, oracle, B, H, L, ranges):
    """5 heads in 3 ranges (2 + 2 + 1), 16 in 4, 7 in 2, 9 in 9: every range is a launch of its own."""
    Q, K, V = normal_fp16(100 + B * H + L, (B, H, L, 128))
    want, _ = fa.attention_host_half(Q, K, V)
    got, ms = fa.attention_host_half_sharded(Q, K, V, devices=[0] * ranges)
    assert len(ms) == ranges and all(x > 0 for x in ms)
    m = oracle.error_metrics(oracle.attention_cpu(Q, K, V), got)
    assert m["max_abs"] <= MAX_ABS_TOL and m["global_rel"] <= REL_L2_TOL, m
    # the same kernel on the same rows of every head: the bits do not depend on how the heads were grouped
    # (shapes here stay on one kernel whatever the head count)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_more_ranges_than_heads(fa):
    Q, K, V = normal_fp16(43, (1, 2, 128, 128))
    want, _ = fa.attention_host_half(Q, K, V)
    got, ms = fa.attention_host_half_sharded(Q, K, V, devices=[0, 0, 0, 0])
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    assert ms[0] > 0 and ms[1] > 0 and ms[2] == 0.0 and ms[3] == 0.0


def test_every_visible_gpu_by_default(fa):
    Q, K, V = normal_fp16(44, (2, 4, 192, 128))
    want, _ = fa.attention_host_half(Q, K, V)
    got, ms = fa.attention_host_half_sharded(Q, K, V)  # devices=None: GPUs 0 .. count-1
    assert len(ms) == torch.cuda.device_count()
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    got2, ms2 = fa.attention_host_half_sharded(Q, K, V, devices=1)  # an int: GPUs 0 .. n-1
    assert len(ms2) == 1 and np.array_equal(got2.view(np.uint16), want.view(np.uint16))


def test_current_device_and_output_outside_the_shape_are_left_alone(fa):
    Q, K, V = normal_fp16(45, (1, 3, 128, 128))
    n = Q.size
    O = np.full(n + 1024, 7.0, dtype=np.float16)
    before = torch.cuda.current_device()
    L_ = fa.lib()
    ms = (ct.c_float * 2)()
    rc = L_.fa_forward_host_half_sharded(fa._np_half_ptr(Q), fa._np_half_ptr(K), fa._np_half_ptr(V),
                                         O.ctypes.data_as(fa.HalfPtr), 1, 3, 128, 128, 64, (ct.c_int * 2)(0, 0), 2, ms)
    assert rc == 0, L_.fa_last_error()
    assert torch.cuda.current_device() == before
    assert np.all(O[n:] == 7.0)
    want, _ = fa.attention_host_half(Q, K, V)
    assert np.array_equal(O[:n].view(np.uint16), want.reshape(-1).view(np.uint16))


def test_argument_errors(fa):
    Q, K, V = normal_fp16(46, (1, 2, 128, 128))
    with pytest.raises(fa.FlashAttentionError, match="not one of the"):
        fa.attention_host_half_sharded(Q, K, V, devices=[0, torch.cuda.device_count()])
    with pytest.raises(fa.FlashAttentionError, match="not one of the"):
        fa.attention_host_half_sharded(Q, K, V, devices=[-1])
    with pytest.raises(fa.FlashAttentionError, match="divisible by tile"):
        fa.attention_host_half_sharded(Q[:, :, :100].copy(), K[:, :, :100].copy(), V[:, :, :100].copy(), devices=[0])
    L_ = fa.lib()
    rc = L_.fa_forward_host_half_sharded(fa._np_half_ptr(Q), fa._np_half_ptr(K), fa._np_half_ptr(V), None, 1, 2, 128, 128,
                                         64, None, 1, None)
    assert rc == 5 and b"null pointer" in L_.fa_last_error()
    rc = L_.fa_forward_host_half_sharded(fa._np_half_ptr(Q), fa._np_half_ptr(K), fa._np_half_ptr(V), fa._np_half_ptr(Q.copy()),
                                         1, 2, 128, 128, 64, (ct.c_int * 1)(0), 0, None)
    assert rc == 4 and b"n_devices > 0" in L_.fa_last_error()
