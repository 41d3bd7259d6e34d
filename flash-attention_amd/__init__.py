"""ctypes front end of ``csrc/libflash_attention.so`` (MI355X / gfx950).

The directory name carries a hyphen, so load this package by path (see
``load_package`` in ``__graft_entry__.py``); its import name is
``flash_attention_amd``.

The argtypes below are the ones the reference's caller declares
(benchmarks/compare_impls.py:30-40): four ``POINTER(c_uint16)``, five ``c_int``,
``c_void_p`` stream, ``POINTER(c_float)`` elapsed.  There is no CPU fallback:
if the shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libflash_attention.so")

HalfPtr = ct.POINTER(ct.c_uint16)
FloatPtr = ct.POINTER(ct.c_float)

EXPORTED_SYMBOLS = (
    "run_tensor_flash_attention",
    "run_tensor_flash_attention_host_half",
    "fa_forward",
    "fa_forward_host_half",
    "fa_forward_host_half_sharded",
    "fa_forward_noalloc",
    "fa_forward_block",
    "fa_forward_general",
    "fa_forward_varlen",
    "fa_forward_block_general",
    "fa_forward_general_host_half",
    "run_naive_attention_host_half",
    "run_cuda_cores_flash_attention_host_half",
    "fa_init_half",
    "fa_device_info",
    "fa_last_error",
    "fa_version",
    "fa_kernel_name",
    "fa_kernel_name_general",
    "fa_kernel_name_general_bh",
    "fa_kernel_name_block_general",
    "fa_split_ways",
    "fa_split_ways_general",
    "fa_set_reserved_cus",
    "fa_get_reserved_cus",
    "fa_debug_reload_env",
)

_lib = None


class FlashAttentionError(RuntimeError):
    pass


def lib() -> ct.CDLL:
    """Load the HIP library; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        _lib = bind(LIB_PATH)
    return _lib


def bind(path: str) -> ct.CDLL:
    """dlopen a build of the library and declare its entry points' argument types."""
    if not os.path.exists(path):
        raise FlashAttentionError(
            f"{path} not found: build it with flash-attention_amd/csrc/shared_compile.sh "
            "(there is no CPU fallback)"
        )
    # PyTorch ships a HIP runtime of its own (same SONAME as /opt/rocm's).  Two HIP runtimes in
    # one process do not work, so when torch is installed its copy must be mapped first; the
    # dynamic linker then binds this library to that copy as well.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ct.CDLL(path)
    sig = [HalfPtr, HalfPtr, HalfPtr, HalfPtr] + [ct.c_int] * 5 + [ct.c_void_p, FloatPtr]
    for name in ("run_tensor_flash_attention", "run_tensor_flash_attention_host_half"):
        getattr(L, name).argtypes = sig
        getattr(L, name).restype = None
    for name in ("fa_forward", "fa_forward_host_half"):
        getattr(L, name).argtypes = sig
        getattr(L, name).restype = ct.c_int
    L.fa_forward_host_half_sharded.argtypes = sig[:-2] + [ct.POINTER(ct.c_int), ct.c_int, FloatPtr]
    L.fa_forward_host_half_sharded.restype = ct.c_int
    L.fa_forward_noalloc.argtypes = sig[:-1]
    L.fa_forward_noalloc.restype = ct.c_int
    L.fa_forward_block.argtypes = (
        [HalfPtr, HalfPtr, HalfPtr, HalfPtr, FloatPtr, FloatPtr, FloatPtr] + [ct.c_int] * 6 + [ct.c_void_p]
    )
    L.fa_forward_block.restype = ct.c_int
    L.fa_forward_general.argtypes = [HalfPtr, HalfPtr, HalfPtr, HalfPtr] + [ct.c_int] * 6 + [ct.c_void_p, FloatPtr]
    L.fa_forward_general.restype = ct.c_int
    L.fa_forward_varlen.argtypes = ([HalfPtr, HalfPtr, HalfPtr, HalfPtr] + [ct.c_int] * 6 +
                                    [ct.POINTER(ct.c_int), ct.POINTER(ct.c_int), ct.c_void_p, FloatPtr])
    L.fa_forward_varlen.restype = ct.c_int
    L.fa_forward_block_general.argtypes = (
        [HalfPtr, HalfPtr, HalfPtr, HalfPtr, FloatPtr, FloatPtr, FloatPtr] + [ct.c_int] * 9 + [ct.c_void_p]
    )
    L.fa_forward_block_general.restype = ct.c_int
    L.fa_forward_general_host_half.argtypes = L.fa_forward_general.argtypes
    L.fa_forward_general_host_half.restype = ct.c_int
    # the sibling libraries' symbols, with the argtypes the reference's caller declares
    # (benchmarks/compare_impls.py:42-58)
    L.run_naive_attention_host_half.argtypes = [HalfPtr] * 4 + [ct.c_int] * 5 + [ct.c_void_p, FloatPtr]
    L.run_naive_attention_host_half.restype = None
    L.run_cuda_cores_flash_attention_host_half.argtypes = [HalfPtr] * 4 + [ct.c_int] * 6 + [ct.c_void_p, FloatPtr]
    L.run_cuda_cores_flash_attention_host_half.restype = None
    L.fa_init_half.argtypes = [HalfPtr, ct.c_size_t, ct.c_ulonglong, ct.c_void_p]
    L.fa_init_half.restype = ct.c_int
    L.fa_device_info.argtypes = [ct.POINTER(ct.c_int), ct.POINTER(ct.c_int), ct.POINTER(ct.c_double)]
    L.fa_device_info.restype = ct.c_int
    L.fa_last_error.argtypes = []
    L.fa_last_error.restype = ct.c_char_p
    L.fa_version.argtypes = []
    L.fa_version.restype = ct.c_char_p
    L.fa_kernel_name.argtypes = [ct.c_int, ct.c_int]
    L.fa_kernel_name.restype = ct.c_char_p
    L.fa_kernel_name_general.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int]
    L.fa_kernel_name_general.restype = ct.c_char_p
    L.fa_kernel_name_general_bh.argtypes = [ct.c_int] * 5
    L.fa_kernel_name_general_bh.restype = ct.c_char_p
    L.fa_kernel_name_block_general.argtypes = [ct.c_int] * 6
    L.fa_kernel_name_block_general.restype = ct.c_char_p
    L.fa_split_ways.argtypes = [ct.c_int] * 3
    L.fa_split_ways.restype = ct.c_int
    L.fa_split_ways_general.argtypes = [ct.c_int] * 5
    L.fa_split_ways_general.restype = ct.c_int
    L.fa_set_reserved_cus.argtypes = [ct.c_int]
    L.fa_set_reserved_cus.restype = ct.c_int
    L.fa_get_reserved_cus.argtypes = []
    L.fa_get_reserved_cus.restype = ct.c_int
    L.fa_debug_reload_env.argtypes = []
    L.fa_debug_reload_env.restype = None
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise FlashAttentionError(f"libflash_attention error {rc}: {lib().fa_last_error().decode()}")


def _np_half_ptr(a: np.ndarray):
    if a.dtype != np.float16 or not a.flags["C_CONTIGUOUS"]:
        raise ValueError("expected a C-contiguous float16 array")
    return a.ctypes.data_as(HalfPtr)


def _dev_half_ptr(t):
    return ct.cast(ct.c_void_p(t.data_ptr()), HalfPtr)


def _dev_float_ptr(t):
    return ct.cast(ct.c_void_p(t.data_ptr()), FloatPtr)


def _stream_handle(stream) -> ct.c_void_p:
    if stream is None:
        return ct.c_void_p(0)
    if isinstance(stream, int):
        return ct.c_void_p(stream)
    return ct.c_void_p(stream.cuda_stream)  # torch.cuda.Stream


def attention_host_half(Q: np.ndarray, K: np.ndarray, V: np.ndarray, tile: int = 64, want_ms: bool = True):
    """Host-pointer path (``run_tensor_flash_attention_host_half``, wrapper.cu:9-35).

    fp16 NumPy arrays of shape [B,H,L,D] in, (fp16 O, kernel-only ms) out."""
    B, H, L, D = Q.shape
    if K.shape != Q.shape or V.shape != Q.shape:
        raise ValueError("Q, K, V must share one [B,H,L,D] shape")
    O = np.empty_like(Q)
    ms = ct.c_float(0.0)
    rc = lib().fa_forward_host_half(
        _np_half_ptr(Q), _np_half_ptr(K), _np_half_ptr(V), _np_half_ptr(O), B, H, L, D, tile, None,
        ct.byref(ms) if want_ms else None,
    )
    _check(rc)
    return O, float(ms.value)


def attention_host_half_sharded(Q: np.ndarray, K: np.ndarray, V: np.ndarray, devices=None, tile: int = 64):
    """The host-pointer path over several GPUs of this node from one process (``fa_forward_host_half_sharded``):
    the flattened B*H axis in contiguous ranges, one per entry of ``devices`` (a list of GPU indices, an int n for
    GPUs 0..n-1, or None for every visible GPU).  Returns (fp16 O, [kernel-only ms of each range])."""
    B, H, L, D = Q.shape
    if K.shape != Q.shape or V.shape != Q.shape:
        raise ValueError("Q, K, V must share one [B,H,L,D] shape")
    O = np.empty_like(Q)
    if devices is None:
        dev_arr, n = None, 0
        n_out = max(1, device_count())
    elif isinstance(devices, int):
        dev_arr, n = None, devices
        n_out = max(1, devices)
    else:
        n = n_out = len(devices)
        dev_arr = (ct.c_int * max(1, n))(*devices)
    ms = (ct.c_float * n_out)()
    rc = lib().fa_forward_host_half_sharded(
        _np_half_ptr(Q), _np_half_ptr(K), _np_half_ptr(V), _np_half_ptr(O), B, H, L, D, tile, dev_arr, n, ms)
    _check(rc)
    return O, [float(x) for x in ms]


def device_count() -> int:
    """GPUs visible to this process (asks torch, which this front end maps first)."""
    import torch

    return torch.cuda.device_count()


def attention(q, k, v, out=None, tile: int = 64, stream=None, timed: bool = False):
    """Device-pointer path (``run_tensor_flash_attention``, wrapper.cu:39-203) on torch
    tensors resident on the GPU.  q,k,v: contiguous fp16 [B,H,L,D].  Returns ``out`` (and
    the kernel-only milliseconds when ``timed``)."""
    import torch

    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise FlashAttentionError("attention() needs GPU tensors; there is no CPU fallback")
    for t in (q, k, v):
        if t.dtype != torch.float16 or not t.is_contiguous():
            raise ValueError("q, k, v must be contiguous float16 tensors")
    B, H, L, D = q.shape
    if k.shape != q.shape or v.shape != q.shape:
        raise ValueError("q, k, v must share one [B,H,L,D] shape")
    if out is None:
        out = torch.empty_like(q)
    if stream is None:
        stream = torch.cuda.current_stream(q.device)
    ms = ct.c_float(0.0)
    with torch.cuda.device(q.device):
        rc = lib().fa_forward(
            _dev_half_ptr(q), _dev_half_ptr(k), _dev_half_ptr(v), _dev_half_ptr(out), B, H, L, D, tile,
            _stream_handle(stream), ct.byref(ms) if timed else None,
        )
    _check(rc)
    return (out, float(ms.value)) if timed else out


FLAG_CAUSAL = 1  # FA_FLAG_CAUSAL


def attention_general_host_half(Q: np.ndarray, K: np.ndarray, V: np.ndarray, causal: bool = False):
    """``fa_forward_general_host_half``: fp16 NumPy arrays Q [B,H,Lq,D], K/V [B,H,Lk,D] in, (fp16 O,
    kernel-only ms) out."""
    B, H, Lq, D = Q.shape
    Lk = K.shape[2]
    if K.shape != (B, H, Lk, D) or V.shape != K.shape:
        raise ValueError("K, V must be [B,H,Lk,D] with Q's B, H, D")
    O = np.empty_like(Q)
    ms = ct.c_float(0.0)
    rc = lib().fa_forward_general_host_half(
        _np_half_ptr(Q), _np_half_ptr(K), _np_half_ptr(V), _np_half_ptr(O), B, H, Lq, Lk, D,
        FLAG_CAUSAL if causal else 0, None, ct.byref(ms),
    )
    _check(rc)
    return O, float(ms.value)


def attention_general(q, k, v, causal: bool = False, out=None, stream=None, timed: bool = False):
    """``fa_forward_general``: q [B,H,Lq,D] against k, v [B,H,Lk,D], any Lq, Lk >= 1; ``causal``
    applies the mask of the reference's CPU baseline (testing.cu:74-77; needs Lq == Lk).
    Contiguous fp16 GPU tensors.  Returns ``out`` (and the kernel-only milliseconds when ``timed``)."""
    import torch

    for t in (q, k, v):
        if not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
            raise ValueError("q, k, v must be contiguous float16 GPU tensors; there is no CPU fallback")
    B, H, Lq, D = q.shape
    Lk = k.shape[2]
    if k.shape != (B, H, Lk, D) or v.shape != k.shape:
        raise ValueError("k, v must be [B,H,Lk,D] with q's B, H, D")
    if out is None:
        out = torch.empty_like(q)
    if stream is None:
        stream = torch.cuda.current_stream(q.device)
    ms = ct.c_float(0.0)
    with torch.cuda.device(q.device):
        rc = lib().fa_forward_general(
            _dev_half_ptr(q), _dev_half_ptr(k), _dev_half_ptr(v), _dev_half_ptr(out), B, H, Lq, Lk, D,
            FLAG_CAUSAL if causal else 0, _stream_handle(stream), ct.byref(ms) if timed else None,
        )
    _check(rc)
    return (out, float(ms.value)) if timed else out


def attention_varlen(q, k, v, q_len=None, k_len=None, causal: bool = False, out=None, stream=None, timed: bool = False):
    """``fa_forward_varlen``: padded batches.  q [B,H,Lq,D], k, v [B,H,Lk,D] contiguous fp16 GPU tensors; ``q_len`` /
    ``k_len`` int32 GPU tensors of B valid lengths (or None).  Rows of ``out`` from ``q_len[b]`` on are left as
    they are.  Returns ``out`` (and the kernel-only milliseconds when ``timed``)."""
    import torch

    for t in (q, k, v):
        if not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
            raise ValueError("q, k, v must be contiguous float16 GPU tensors; there is no CPU fallback")
    B, H, Lq, D = q.shape
    Lk = k.shape[2]
    if k.shape != (B, H, Lk, D) or v.shape != k.shape:
        raise ValueError("k, v must be [B,H,Lk,D] with q's B, H, D")
    for t in (q_len, k_len):
        if t is not None and (not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous() or t.numel() != B):
            raise ValueError("q_len / k_len must be contiguous int32 GPU tensors of B entries")
    if out is None:
        out = torch.zeros_like(q)
    if stream is None:
        stream = torch.cuda.current_stream(q.device)
    ip = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), ct.POINTER(ct.c_int)) if t is not None else None
    ms = ct.c_float(0.0)
    with torch.cuda.device(q.device):
        rc = lib().fa_forward_varlen(
            _dev_half_ptr(q), _dev_half_ptr(k), _dev_half_ptr(v), _dev_half_ptr(out), B, H, Lq, Lk, D,
            FLAG_CAUSAL if causal else 0, ip(q_len), ip(k_len), _stream_handle(stream), ct.byref(ms) if timed else None,
        )
    _check(rc)
    return (out, float(ms.value)) if timed else out


class BlockState:
    """Carried (o_acc, row_max, row_sum) of the blockwise forward, on the GPU."""

    def __init__(self, BH: int, Lq: int, D: int, device):
        import torch

        self.o_acc = torch.empty((BH, Lq, D), dtype=torch.float32, device=device)
        self.row_max = torch.empty((BH, Lq), dtype=torch.float32, device=device)
        self.row_sum = torch.empty((BH, Lq), dtype=torch.float32, device=device)
        self.first = True


def attention_block(q, k_blk, v_blk, state: BlockState, last: bool, out=None, stream=None):
    """One step of the blockwise forward (``fa_forward_block``): fold the K/V block into
    ``state``; on the ``last`` step write the normalised fp16 output into ``out``."""
    import torch

    B, H, Lq, D = q.shape
    Lk = k_blk.shape[2]
    if k_blk.shape != (B, H, Lk, D) or v_blk.shape != k_blk.shape:
        raise ValueError("k_blk / v_blk must be [B,H,Lk,D]")
    for t in (q, k_blk, v_blk):
        if not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
            raise ValueError("q, k_blk, v_blk must be contiguous float16 GPU tensors")
    if last and out is None:
        out = torch.empty_like(q)
    if stream is None:
        stream = torch.cuda.current_stream(q.device)
    with torch.cuda.device(q.device):
        rc = lib().fa_forward_block(
            _dev_half_ptr(q), _dev_half_ptr(k_blk), _dev_half_ptr(v_blk),
            _dev_half_ptr(out) if out is not None else None,
            _dev_float_ptr(state.o_acc), _dev_float_ptr(state.row_max), _dev_float_ptr(state.row_sum),
            B * H, Lq, Lk, D, int(state.first), int(last), _stream_handle(stream),
        )
    _check(rc)
    state.first = False
    return out


def attention_block_general(q, k_blk, v_blk, state: BlockState, last: bool, causal: bool = False, q_pos0: int = 0,
                            k_pos0: int = 0, out=None, stream=None):
    """One step of the blockwise forward through ``fa_forward_block_general``: any Lq, Lk >= 1, D of
    64, 128 or 256; with ``causal`` key ``k_pos0 + j`` is visible to query ``q_pos0 + i`` iff it is not
    ahead of it, so blocks may be folded in any order."""
    import torch

    B, H, Lq, D = q.shape
    Lk = k_blk.shape[2]
    if k_blk.shape != (B, H, Lk, D) or v_blk.shape != k_blk.shape:
        raise ValueError("k_blk / v_blk must be [B,H,Lk,D]")
    for t in (q, k_blk, v_blk):
        if not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
            raise ValueError("q, k_blk, v_blk must be contiguous float16 GPU tensors")
    if last and out is None:
        out = torch.empty_like(q)
    if stream is None:
        stream = torch.cuda.current_stream(q.device)
    with torch.cuda.device(q.device):
        rc = lib().fa_forward_block_general(
            _dev_half_ptr(q), _dev_half_ptr(k_blk), _dev_half_ptr(v_blk),
            _dev_half_ptr(out) if out is not None else None,
            _dev_float_ptr(state.o_acc), _dev_float_ptr(state.row_max), _dev_float_ptr(state.row_sum),
            B * H, Lq, Lk, D, FLAG_CAUSAL if causal else 0, int(q_pos0), int(k_pos0), int(state.first), int(last),
            _stream_handle(stream),
        )
    _check(rc)
    state.first = False
    return out


def init_half(t, seed: int, stream=None) -> None:
    """Fill a GPU fp16 tensor with the throughput driver's pattern
    (benchmark_tensor_flash_attention.cu:34-41)."""
    import torch

    if not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
        raise ValueError("init_half needs a contiguous float16 GPU tensor")
    if stream is None:
        stream = torch.cuda.current_stream(t.device)
    with torch.cuda.device(t.device):
        _check(lib().fa_init_half(_dev_half_ptr(t), t.numel(), seed, _stream_handle(stream)))


def set_reserved_cus(k: int) -> None:
    """``fa_set_reserved_cus``: persistent launches leave ``k`` compute units free for other streams' kernels
    (the transport kernels of a ring's send/recv); 0 restores one workgroup per unit."""
    _check(lib().fa_set_reserved_cus(int(k)))


def reload_env() -> None:
    """Re-read the FA_* environment switches (they are otherwise read once per process).  For tests and A/B runs."""
    lib().fa_debug_reload_env()


def device_info() -> dict:
    cus, khz, peak = ct.c_int(0), ct.c_int(0), ct.c_double(0.0)
    _check(lib().fa_device_info(ct.byref(cus), ct.byref(khz), ct.byref(peak)))
    return {"compute_units": cus.value, "clock_khz": khz.value, "fp16_dense_peak_flops": peak.value}


def mae_mre(ref32: np.ndarray, test16: np.ndarray):
    """Mean absolute error and mean relative error over the reference's non-zero entries, the two figures
    benchmarks/compare_impls.py:66-74 prints."""
    a = np.asarray(ref32, dtype=np.float32).reshape(-1)
    d = np.abs(a - np.asarray(test16).astype(np.float32).reshape(-1))
    nz = a != 0
    return float(d.mean()), (float((d[nz] / np.abs(a[nz])).mean()) if nz.any() else float("nan"))


def attention_flops(B: int, H: int, L: int, D: int) -> float:
    """4*B*H*L^2*D (benchmark_tensor_flash_attention.cu:218)."""
    return 4.0 * B * H * float(L) * float(L) * D
