#!/usr/bin/env python3
"""Caller-side counterpart of the reference's benchmarks/compare_impls.py.

Same flow: load the shared library by path with ctypes and the reference's argtypes
(compare_impls.py:30-40), draw Q, K, V with default_rng(0) as float32 -> float16
(:84-87) at B=32,H=16,L=512,D=128, build the fp32 softmax reference with torch when a
GPU is visible (:94-103), call the host-pointer entry point (:107-110), print MAE / MRE
(:66-74,129-138) and the kernel-only milliseconds (:140-145).  The reference's two
optional sibling libraries (CUDA-cores and naive kernels, :17-28) are single symbols of
the same library here (served by the same kernels), so the three-way table has three
rows of one implementation; they are called with the reference's arguments (:112-122).
"""
import ctypes as ct
import os

import numpy as np

try:
    import torch  # also maps PyTorch's HIP runtime before the library's own dependency resolves

    HAS_TORCH = True
except Exception:
    HAS_TORCH = False

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "csrc", "libflash_attention.so")

HalfPtr = ct.POINTER(ct.c_uint16)
FloatPtr = ct.POINTER(ct.c_float)


def load():
    lib = ct.CDLL(LIB)
    lib.run_tensor_flash_attention_host_half.argtypes = [
        HalfPtr, HalfPtr, HalfPtr, HalfPtr, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, FloatPtr,
    ]
    lib.run_tensor_flash_attention_host_half.restype = None
    lib.run_cuda_cores_flash_attention_host_half.argtypes = [
        HalfPtr, HalfPtr, HalfPtr, HalfPtr, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p,
        FloatPtr,
    ]
    lib.run_cuda_cores_flash_attention_host_half.restype = None
    lib.run_naive_attention_host_half.argtypes = [
        HalfPtr, HalfPtr, HalfPtr, HalfPtr, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, FloatPtr,
    ]
    lib.run_naive_attention_host_half.restype = None
    return lib


def _front_end():
    """The package's ctypes front end (flash-attention_amd/__init__.py), loaded by path: its fp16 pointer helper and
    the oracle-side MAE / MRE are what this script uses, rather than restating them."""
    import importlib.util
    import sys

    if "flash_attention_amd" in sys.modules:
        return sys.modules["flash_attention_amd"]
    spec = importlib.util.spec_from_file_location("flash_attention_amd", os.path.join(ROOT, "__init__.py"),
                                                  submodule_search_locations=[ROOT])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["flash_attention_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def main(B=32, H=16, L=512, D=128, tile=64, KQVO_block_y=32, warps=4):
    lib = load()
    fe = _front_end()
    as_half_ptr, metrics = fe._np_half_ptr, fe.mae_mre
    size = B * H * L * D
    rng = np.random.default_rng(0)
    Q = rng.standard_normal(size, dtype=np.float32).astype(np.float16)
    K = rng.standard_normal(size, dtype=np.float32).astype(np.float16)
    V = rng.standard_normal(size, dtype=np.float32).astype(np.float16)
    O_tc, O_cc, O_nv = np.zeros_like(Q), np.zeros_like(Q), np.zeros_like(Q)

    O_ref = None
    if HAS_TORCH and torch.cuda.is_available():
        dev = torch.device("cuda")
        Qt, Kt, Vt = [torch.from_numpy(a.reshape(B, H, L, D)).to(dev) for a in (Q, K, V)]
        with torch.no_grad():
            scores = torch.matmul(Qt.float(), Kt.transpose(-1, -2).float()) * (1.0 / np.sqrt(D))
            O_ref = torch.matmul(torch.softmax(scores, dim=-1), Vt.float()).to(torch.float16).cpu().numpy().reshape(-1)
    else:
        print("[warn] Torch GPU not available; metrics will be relative to the kernel's own output.")

    elapsed = ct.c_float(0.0)
    lib.run_tensor_flash_attention_host_half(
        as_half_ptr(Q), as_half_ptr(K), as_half_ptr(V), as_half_ptr(O_tc), B, H, L, D, tile, None, ct.byref(elapsed))
    elapsed_cc = ct.c_float(0.0)
    lib.run_cuda_cores_flash_attention_host_half(
        as_half_ptr(Q), as_half_ptr(K), as_half_ptr(V), as_half_ptr(O_cc), B, H, L, D, KQVO_block_y, warps, None,
        ct.byref(elapsed_cc))
    elapsed_nv = ct.c_float(0.0)
    lib.run_naive_attention_host_half(
        as_half_ptr(Q), as_half_ptr(K), as_half_ptr(V), as_half_ptr(O_nv), B, H, L, D, 0, None, ct.byref(elapsed_nv))

    ref = O_tc if O_ref is None else O_ref
    print("\nPrecision metrics vs", "Torch ref" if O_ref is not None else "Tensor Cores")
    mae, mre = metrics(ref.astype(np.float32), O_tc)
    print(f"Tensor Cores: MAE={mae:.6f} MRE={mre:.6e}")
    for name, out in (("CUDA Cores", O_cc), ("Naive", O_nv)):
        a, r = metrics(ref.astype(np.float32), out)
        print(f"{name}: MAE={a:.6f} MRE={r:.6e}")
    print("\nSpeed (ms)")
    print(f"Tensor Cores: {elapsed.value:.3f} ms")
    print(f"CUDA Cores:   {elapsed_cc.value:.3f} ms")
    print(f"Naive:        {elapsed_nv.value:.3f} ms")
    return mae, mre, elapsed.value


if __name__ == "__main__":
    main()
