"""CPU tier: pins the oracle (C restatement of attention_cpu, testing.cu:41-132)
against the golden fixtures, against the reference notebook's NumPy function
(outputs committed in tests/golden), and against itself in its other forms."""
import json
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, normal_fp16


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_half_to_float_exhaustive(oracle):
    h = np.arange(65536, dtype=np.uint16)
    out = np.empty(65536, dtype=np.float32)
    oracle.lib().oracle_halfs_to_floats(
        h.ctypes.data_as(oracle._U16P), out.ctypes.data_as(oracle._F32P), h.size)
    want = h.view(np.float16).astype(np.float32)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(out), nan)
    assert np.array_equal(bits(out)[~nan], bits(want)[~nan])


def test_float_to_half_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    f = np.concatenate([
        rng.standard_normal(200000).astype(np.float32),
        (rng.standard_normal(200000) * 1e-5).astype(np.float32),   # subnormal halves
        (rng.standard_normal(100000) * 3e4).astype(np.float32),    # near overflow
        rng.integers(0, 2**32, 300000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e9, -1e9, np.inf, -np.inf,
                  2.0**-24, 2.0**-25, 2.0**-25 * 1.0000001, 2.0**-14, 0.1, 1.0 + 2.0**-11, 1.0 + 3 * 2.0**-11],
                 dtype=np.float32),
    ])
    out = np.empty(f.size, dtype=np.uint16)
    oracle.lib().oracle_floats_to_halfs(f.ctypes.data_as(oracle._F32P), out.ctypes.data_as(oracle._U16P), f.size)
    with np.errstate(over="ignore", invalid="ignore"):
        want = f.astype(np.float16).view(np.uint16)
    nan = np.isnan(f)
    assert np.array_equal(out[~nan], want[~nan])
    assert np.all((out[nan] & 0x7C00) == 0x7C00) and np.all((out[nan] & 0x3FF) != 0)


@pytest.mark.parametrize("name", ["f1_testing_l64.npz", "f2_c1_l512.npz"])
def test_oracle_reproduces_golden_bits(oracle, name):
    g = load_golden(name)
    O = oracle.attention_cpu(g["Q"], g["K"], g["V"], nthreads=1)
    assert np.array_equal(bits(O), bits(g["O_oracle"]))
    O_mt = oracle.attention_cpu(g["Q"], g["K"], g["V"], nthreads=0)
    assert np.array_equal(bits(O_mt), bits(O))


@pytest.mark.parametrize("name", ["f1_testing_l64.npz", "f2_c1_l512.npz"])
def test_oracle_against_reference_numpy_outputs(oracle, name):
    """O_ref_numpy was produced by the reference's own flash_attention_numpy
    (proof_correctness.ipynb cell 2100755c), which keeps P un-rounded; the
    restated attention_cpu rounds P to fp16 (testing.cu:107), so agreement is at
    the fp16-P level: a few 1e-4 absolute, far inside the 2e-3 / 1e-3 parity gate."""
    g = load_golden(name)
    O = oracle.attention_cpu(g["Q"], g["K"], g["V"])
    m = oracle.error_metrics(g["O_ref_numpy"], O)
    assert m["max_abs"] < 4e-4 and m["global_rel"] < 4e-4, m
    m = oracle.error_metrics(g["O_dense64"], O)
    assert m["max_abs"] < 4e-4 and m["global_rel"] < 4e-4, m
    # and the reference's two formulations agree with each other to fp32 round-off
    m = oracle.error_metrics(g["O_dense64"], g["O_ref_numpy"])
    assert m["max_abs"] < 1e-5, m


def test_f2_inputs_follow_compare_impls_recipe():
    """compare_impls.py:84-87: default_rng(0), standard_normal float32 -> float16, Q then K then V."""
    g = load_golden("f2_c1_l512.npz")
    Q, K, V = normal_fp16(0, 512 * 128)
    assert np.array_equal(Q.view(np.uint16), g["Q"].reshape(-1).view(np.uint16))
    assert np.array_equal(K.view(np.uint16), g["K"].reshape(-1).view(np.uint16))
    assert np.array_equal(V.view(np.uint16), g["V"].reshape(-1).view(np.uint16))


def test_f1_inputs_look_like_testing_recipe():
    """testing.cu:158-165: N(0,1)+0.5 rounded to fp16; 3*8192 draws."""
    g = load_golden("f1_testing_l64.npz")
    for k in ("Q", "K", "V"):
        a = g[k].astype(np.float64)
        assert a.shape == (1, 1, 64, 128)
        assert abs(a.mean() - 0.5) < 0.05 and abs(a.std() - 1.0) < 0.05


@pytest.mark.parametrize("causal", [False, True])
def test_literal_and_fast_forms_are_bit_identical(oracle, causal):
    Q, K, V = normal_fp16(3, (2, 2, 192, 128))
    a = oracle.attention_cpu(Q, K, V, causal=causal, literal=True)
    b = oracle.attention_cpu(Q, K, V, causal=causal, nthreads=3)
    assert np.array_equal(bits(a), bits(b))


def test_literal_form_handles_other_head_dims_and_ragged_tail(oracle):
    # L not a multiple of 64 exercises the t_end = min(t+64, L) tail (testing.cu:66)
    Q, K, V = normal_fp16(4, (1, 2, 100, 64))
    a = oracle.attention_cpu(Q, K, V, literal=True)
    b = oracle.attention_cpu(Q, K, V)
    assert np.array_equal(bits(a), bits(b))
    m = oracle.error_metrics(oracle.attention_dense_fp64(Q, K, V).astype(np.float32), a)
    assert m["max_abs"] < 5e-4


@pytest.mark.parametrize("causal", [False, True])
def test_numpy_restatement_agrees(oracle, causal):
    Q, K, V = normal_fp16(5, (1, 2, 256, 128))
    a = oracle.attention_cpu(Q, K, V, causal=causal)
    b = oracle.attention_numpy(Q, K, V, causal=causal)
    m = oracle.error_metrics(a, b)
    # same recipe, different summation association; an occasional 1-ulp flip of a fp16 P
    # (causal rows with one or two visible keys feel such a flip most)
    assert m["max_abs"] < (1e-4 if causal else 2e-5), m
    d = oracle.attention_dense_fp64(Q, K, V, causal=causal).astype(np.float32)
    assert oracle.error_metrics(d, a)["max_abs"] < 5e-4


@pytest.mark.parametrize("nchunks", [2, 4, 8])
def test_blockwise_merge_equals_full_pass(oracle, nchunks):
    """F4: folding K/V blocks in order with carried (acc,max,sum) is the same
    arithmetic as one pass (testing.cu:91-96 applied across blocks)."""
    g = load_golden("f2_c1_l512.npz")
    Q, K, V = g["Q"], g["K"], g["V"]
    L = Q.shape[2]
    st = oracle.BlockState(1, 1, L, 128)
    step = L // nchunks
    for c in range(nchunks):
        oracle.block_update(st, Q, np.ascontiguousarray(K[:, :, c * step:(c + 1) * step]),
                            np.ascontiguousarray(V[:, :, c * step:(c + 1) * step]), k_pos0=c * step)
    O = oracle.block_finalize(st)
    assert np.array_equal(bits(O), bits(g["O_oracle"]))


def test_blockwise_merge_any_order_is_close(oracle):
    """A ring visits blocks in rotated order: not bit-identical, but the same value to fp32 round-off
    plus the fp16 rounding of P relative to a different running max."""
    Q, K, V = normal_fp16(6, (1, 2, 256, 128))
    full = oracle.attention_cpu(Q, K, V)
    st = oracle.BlockState(1, 2, 256, 128)
    for c in (2, 3, 0, 1):
        oracle.block_update(st, Q, np.ascontiguousarray(K[:, :, c * 64:(c + 1) * 64]),
                            np.ascontiguousarray(V[:, :, c * 64:(c + 1) * 64]))
    m = oracle.error_metrics(full, oracle.block_finalize(st))
    assert m["max_abs"] < 3e-4 and m["global_rel"] < 3e-4, m


def test_row_subset_equals_full_rows(oracle):
    Q, K, V = normal_fp16(7, (2, 3, 256, 128))
    rows = [0, 1, 63, 64, 130, 255]
    full = oracle.attention_cpu(Q, K, V)
    sub = oracle.attention_cpu_rows(Q, K, V, rows)
    assert np.array_equal(bits(sub), bits(full[:, :, rows, :]))


def test_init_half_pattern(oracle):
    """benchmark_tensor_flash_attention.cu:34-41: multiples of 1/1024 in [-0.5,0.5), exact in fp16."""
    ref = json.load(open(os.path.join(GOLDEN, "f3_init_half.json")))
    for seed in (1234, 2345, 3456):
        a = oracle.init_half(65536, seed)
        assert zlib.crc32(a.tobytes()) == ref["crc32_first_65536"][str(seed)]
        idx = np.arange(65536, dtype=np.uint64)
        want = (((idx * np.uint64(16807) + np.uint64(seed)) & np.uint64(0xFFFF)) % np.uint64(1024)).astype(np.float64) / 1024 - 0.5
        assert np.array_equal(a.astype(np.float64), want)
    # a window of a larger buffer equals the matching slice
    assert np.array_equal(oracle.init_half(100, 1234, idx0=5000), oracle.init_half(5100, 1234)[5000:])


def test_error_metrics_definitions(oracle):
    """testing.cu:221-236 and compare_impls.py:66-74."""
    ref = np.array([1.0, -2.0, 0.0, 4.0], dtype=np.float32)
    tst = np.array([1.5, -2.0, 0.25, 3.0], dtype=np.float32)
    m = oracle.error_metrics(ref, tst)
    d = np.abs(ref - tst)
    assert m["max_abs"] == pytest.approx(1.0)
    assert m["mean_abs"] == pytest.approx(d.mean())
    assert m["max_rel"] == pytest.approx((d / (np.abs(ref) + 1e-6)).max(), rel=1e-5)
    assert m["global_rel"] == pytest.approx(np.sqrt((d ** 2).sum()) / (np.sqrt((ref ** 2).sum()) + 1e-6), rel=1e-5)
    mae, mre = oracle.mae_mre(ref, tst)
    assert mae == pytest.approx(d.mean())
    assert mre == pytest.approx(np.mean(d[[0, 1, 3]] / np.abs(ref[[0, 1, 3]])))
