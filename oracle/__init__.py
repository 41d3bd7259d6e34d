"""CPU oracle for the forward-attention hot path.  TEST INFRASTRUCTURE ONLY.

ctypes front end of ``oracle/attention_cpu.c`` (a C restatement of the
reference's ``attention_cpu``, flash_attention_tensor_cores/testing.cu:41-132)
plus a NumPy restatement used as an independent cross-check.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` import
this package, and only as the checker / the timed CPU baseline.  Nothing under
the product package imports it.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_attention.so")
_lib = None

_U16P = ct.POINTER(ct.c_uint16)
_F32P = ct.POINTER(ct.c_float)
_I32P = ct.POINTER(ct.c_int)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (no GPU toolchain involved)."""
    src = os.path.join(_HERE, "attention_cpu.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> ct.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ct.CDLL(_LIB_PATH)
        L.oracle_half_to_float.argtypes = [ct.c_uint16]
        L.oracle_half_to_float.restype = ct.c_float
        L.oracle_float_to_half.argtypes = [ct.c_float]
        L.oracle_float_to_half.restype = ct.c_uint16
        L.oracle_halfs_to_floats.argtypes = [_U16P, _F32P, ct.c_size_t]
        L.oracle_halfs_to_floats.restype = None
        L.oracle_floats_to_halfs.argtypes = [_F32P, _U16P, ct.c_size_t]
        L.oracle_floats_to_halfs.restype = None
        L.oracle_attention_cpu_literal.argtypes = [_U16P, _U16P, _U16P, _F32P] + [ct.c_int] * 5
        L.oracle_attention_cpu_literal.restype = None
        L.oracle_attention_cpu.argtypes = [_U16P, _U16P, _U16P, _F32P] + [ct.c_int] * 6
        L.oracle_attention_cpu.restype = ct.c_int
        L.oracle_attention_cpu_rows.argtypes = (
            [_U16P, _U16P, _U16P, _F32P] + [ct.c_int] * 5 + [_I32P, ct.c_int, ct.c_int]
        )
        L.oracle_attention_cpu_rows.restype = ct.c_int
        L.oracle_attention_block_update.argtypes = (
            [_U16P, _U16P, _U16P, _F32P, _F32P, _F32P] + [ct.c_int] * 6 + [ct.c_long, ct.c_long, ct.c_int, ct.c_int]
        )
        L.oracle_attention_block_update.restype = ct.c_int
        L.oracle_attention_block_finalize.argtypes = [_F32P, _F32P, _F32P, ct.c_size_t, ct.c_int]
        L.oracle_attention_block_finalize.restype = None
        L.oracle_init_half.argtypes = [_U16P, ct.c_size_t, ct.c_size_t, ct.c_ulonglong]
        L.oracle_init_half.restype = None
        L.oracle_error_metrics.argtypes = [_F32P, _F32P, ct.c_size_t, ct.POINTER(ct.c_double)]
        L.oracle_error_metrics.restype = None
        L.oracle_max_threads.argtypes = []
        L.oracle_max_threads.restype = ct.c_int
        _lib = L
    return _lib


def _h(a: np.ndarray):
    assert a.dtype == np.float16 and a.flags["C_CONTIGUOUS"], "fp16 C-contiguous array required"
    return a.ctypes.data_as(_U16P)


def _f(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], "fp32 C-contiguous array required"
    return a.ctypes.data_as(_F32P)


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def attention_cpu(Q, K, V, causal: bool = False, nthreads: int = 0, literal: bool = False) -> np.ndarray:
    """fp32 O = attention_cpu(Q,K,V) for fp16 [B,H,L,D] inputs (testing.cu:41-132).

    ``literal=True`` runs the reference's exact loop nest (slow, single thread);
    otherwise the bit-identical reordered form on ``nthreads`` threads (0 = all).
    """
    B, H, L, D = Q.shape
    assert K.shape == Q.shape and V.shape == Q.shape
    O = np.empty((B, H, L, D), dtype=np.float32)
    if literal:
        lib().oracle_attention_cpu_literal(_h(Q), _h(K), _h(V), _f(O), B, H, L, D, int(causal))
    else:
        rc = lib().oracle_attention_cpu(_h(Q), _h(K), _h(V), _f(O), B, H, L, D, int(causal), int(nthreads))
        if rc != 0:
            raise ValueError("oracle_attention_cpu: bad arguments")
    return O


def attention_cpu_rows(Q, K, V, rows, causal: bool = False, nthreads: int = 0) -> np.ndarray:
    """fp32 O[:, :, rows, :] only: cost 4*len(rows)*L*D per head instead of 4*L*L*D."""
    B, H, L, D = Q.shape
    rows = np.ascontiguousarray(np.asarray(rows, dtype=np.int32))
    assert rows.ndim == 1 and rows.min() >= 0 and rows.max() < L
    O = np.empty((B, H, len(rows), D), dtype=np.float32)
    rc = lib().oracle_attention_cpu_rows(
        _h(Q), _h(K), _h(V), _f(O), B, H, L, D, int(causal), rows.ctypes.data_as(_I32P), len(rows), int(nthreads)
    )
    if rc != 0:
        raise ValueError("oracle_attention_cpu_rows: bad arguments")
    return O


class BlockState:
    """Carried (acc, max, sum) of the blockwise form; the state a sequence-parallel
    ring hands from step to step (same rescale rule as testing.cu:91-96)."""

    def __init__(self, B, H, Lq, D):
        self.acc = np.zeros((B, H, Lq, D), dtype=np.float32)
        self.mx = np.full((B, H, Lq), -1e30, dtype=np.float32)
        self.sm = np.zeros((B, H, Lq), dtype=np.float32)
        self.first = True


def block_update(state: BlockState, Q, Kb, Vb, causal=False, q_pos0=0, k_pos0=0, nthreads=0) -> None:
    B, H, Lq, D = Q.shape
    Lk = Kb.shape[2]
    assert Kb.shape == (B, H, Lk, D) and Vb.shape == Kb.shape
    rc = lib().oracle_attention_block_update(
        _h(Q), _h(Kb), _h(Vb), _f(state.acc), _f(state.mx), _f(state.sm),
        B, H, Lq, Lk, D, int(causal), int(q_pos0), int(k_pos0), int(state.first), int(nthreads),
    )
    if rc != 0:
        raise ValueError("oracle_attention_block_update: bad arguments")
    state.first = False


def block_finalize(state: BlockState) -> np.ndarray:
    O = np.empty_like(state.acc)
    D = state.acc.shape[-1]
    lib().oracle_attention_block_finalize(_f(state.acc), _f(state.sm), _f(O), state.sm.size, D)
    return O


def init_half(n: int, seed: int, idx0: int = 0) -> np.ndarray:
    """The benchmark driver's input pattern (benchmark_tensor_flash_attention.cu:34-41)."""
    out = np.empty(n, dtype=np.float16)
    lib().oracle_init_half(_h(out), idx0, n, seed)
    return out


def error_metrics(ref32: np.ndarray, test: np.ndarray) -> dict:
    """The five metrics of testing.cu:214-236 (ref fp32, test converted to fp32)."""
    a = np.ascontiguousarray(ref32, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(test.astype(np.float32)).reshape(-1)
    out = (ct.c_double * 5)()
    lib().oracle_error_metrics(_f(a), _f(b), a.size, out)
    return {
        "mean_abs": out[0], "max_abs": out[1], "mean_rel": out[2], "max_rel": out[3], "global_rel": out[4],
    }


def mae_mre(ref32: np.ndarray, test: np.ndarray):
    """MAE and MRE over non-zero reference entries (benchmarks/compare_impls.py:66-74)."""
    a = ref32.astype(np.float32).reshape(-1)
    b = test.astype(np.float32).reshape(-1)
    diff = a - b
    ad = np.abs(diff)
    nz = np.abs(a) > 0
    mre = float(np.mean(ad[nz] / np.abs(a[nz]))) if np.any(nz) else float("nan")
    return float(np.mean(ad)), mre


# ---------------------------------------------------------------------------
# NumPy restatement (independent of the C file; vectorised over query rows, so
# its sums associate differently and it agrees to fp32 round-off, not bitwise).
# ---------------------------------------------------------------------------

def attention_numpy(Q, K, V, causal: bool = False, tile: int = 64) -> np.ndarray:
    """Same recipe as attention_cpu in NumPy fp32: 64-key tiles, running max/sum,
    P rounded to fp16 before P.V, row sum from the un-rounded exponentials,
    fp32 output (testing.cu:52-128)."""
    B, H, L, D = Q.shape
    q = Q.astype(np.float32)
    k = K.astype(np.float32)
    v = V.astype(np.float32)
    scale = np.float32(1.0) / np.sqrt(np.float32(D))
    acc = np.zeros((B, H, L, D), dtype=np.float32)
    mx = np.full((B, H, L), -1e30, dtype=np.float32)
    sm = np.zeros((B, H, L), dtype=np.float32)
    rows = np.arange(L)[:, None]
    for t in range(0, L, tile):
        te = min(t + tile, L)
        s = np.matmul(q, k[:, :, t:te, :].transpose(0, 1, 3, 2)) * scale
        if causal:
            masked = np.arange(t, te)[None, :] > rows
            s = np.where(masked, np.float32(-1e30), s)
            local = np.where(masked, np.float32(-1e30), s).max(axis=-1)
        else:
            local = s.max(axis=-1)
        new_mx = np.maximum(mx, local)
        resc = np.exp(mx - new_mx, dtype=np.float32)
        acc *= resc[..., None]
        sm *= resc
        ex = np.exp(s - new_mx[..., None], dtype=np.float32)
        sm += ex.sum(axis=-1, dtype=np.float32)
        p = ex.astype(np.float16).astype(np.float32)
        acc += np.matmul(p, v[:, :, t:te, :])
        mx = new_mx
    return acc / sm[..., None]


def attention_dense_fp64(Q, K, V, causal: bool = False) -> np.ndarray:
    """Plain softmax(QK^T/sqrt(D))V in float64 (the expression of compare_impls.py:100-102)."""
    B, H, L, D = Q.shape
    q = Q.astype(np.float64)
    k = K.astype(np.float64)
    v = V.astype(np.float64)
    s = np.matmul(q, k.transpose(0, 1, 3, 2)) / np.sqrt(float(D))
    if causal:
        s = np.where(np.arange(L)[None, :] > np.arange(L)[:, None], -np.inf, s)
    s -= s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=-1, keepdims=True)
    return np.matmul(p, v)
