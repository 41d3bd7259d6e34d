"""CPU tier for the C ABI: the library loads without a GPU, exports every symbol
include/flash_attention.h declares, and rejects bad arguments before touching
the device (validation of wrapper.cu:53-61 plus the D=128 requirement)."""
import ctypes as ct
import os
import re

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "flash_attention.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(run_[a-z_0-9]+|fa_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_drop_in_symbols():
    names = declared_functions()
    assert "run_tensor_flash_attention" in names
    assert "run_tensor_flash_attention_host_half" in names
    assert len(names) >= 8


def test_library_exports_every_declared_symbol(fa):
    L = fa.lib()
    for name in declared_functions():
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert sorted(fa.EXPORTED_SYMBOLS) == declared_functions()


def test_version_and_last_error(fa):
    assert b"gfx950" in fa.lib().fa_version()
    assert isinstance(fa.lib().fa_last_error(), bytes)


def _call(fa, fn, B=1, H=1, L=64, D=128, tile=64, null=False):
    buf = np.zeros(16, dtype=np.float16)
    p = None if null else buf.ctypes.data_as(fa.HalfPtr)
    return getattr(fa.lib(), fn)(p, p, p, p, B, H, L, D, tile, None, None)


@pytest.mark.parametrize("fn", ["fa_forward", "fa_forward_host_half"])
def test_argument_validation_happens_before_any_device_work(fa, fn):
    L = fa.lib()
    assert _call(fa, fn, tile=32) == 1 and b"tile must be 64" in L.fa_last_error()
    assert _call(fa, fn, L=100) == 2 and b"divisible by tile" in L.fa_last_error()
    assert _call(fa, fn, L=0) == 2
    assert _call(fa, fn, D=64) == 3 and b"D must be 128" in L.fa_last_error()
    assert _call(fa, fn, B=0) == 4
    assert _call(fa, fn, H=-1) == 4
    assert _call(fa, fn, null=True) == 5


def test_block_entry_point_validation(fa):
    L = fa.lib()
    buf = np.zeros(16, dtype=np.float16)
    f32 = np.zeros(16, dtype=np.float32)
    h = buf.ctypes.data_as(fa.HalfPtr)
    f = f32.ctypes.data_as(fa.FloatPtr)
    assert L.fa_forward_block(h, h, h, h, f, f, f, 1, 64, 64, 64, 1, 1, None) == 3
    assert L.fa_forward_block(h, h, h, h, f, f, f, 1, 96, 64, 128, 1, 1, None) == 2
    assert L.fa_forward_block(h, h, h, h, f, f, f, 1, 64, 0, 128, 1, 1, None) == 2
    assert L.fa_forward_block(h, h, h, h, f, f, f, 0, 64, 64, 128, 1, 1, None) == 4
    assert L.fa_forward_block(None, h, h, h, f, f, f, 1, 64, 64, 128, 1, 1, None) == 5
    assert L.fa_forward_block(h, h, h, None, f, f, f, 1, 64, 64, 128, 1, 1, None) == 5   # last needs dO
    assert L.fa_forward_block(h, h, h, h, None, f, f, 1, 64, 64, 128, 1, 0, None) == 5   # state needed
    assert L.fa_init_half(None, 10, 1, None) == 5


def test_general_entry_points_validation(fa):
    """fa_forward_general and its blockwise / host forms refuse bad arguments before any device work."""
    L = fa.lib()
    buf = np.zeros(16, dtype=np.float16)
    f32 = np.zeros(16, dtype=np.float32)
    h = buf.ctypes.data_as(fa.HalfPtr)
    f = f32.ctypes.data_as(fa.FloatPtr)
    gen = lambda B=1, H=1, Lq=10, Lk=10, D=128, flags=0, q=h: L.fa_forward_general(q, h, h, h, B, H, Lq, Lk, D, flags, None, None)
    assert gen(D=96) == 3 and b"64, 128 or 256" in L.fa_last_error()
    assert gen(Lq=0) == 2 and gen(Lk=-5) == 2
    assert gen(flags=2) == 4 and b"unknown flag" in L.fa_last_error()
    assert gen(Lq=10, Lk=11, flags=1) == 2 and b"causal mask needs Lq == Lk" in L.fa_last_error()
    assert gen(B=0) == 4 and gen(H=0) == 4
    assert gen(q=None) == 5
    blk = lambda Lq=10, Lk=10, D=128, flags=0, qp=0, kp=0, first=1, last=1, o=h, acc=f: L.fa_forward_block_general(
        h, h, h, o, acc, f, f, 1, Lq, Lk, D, flags, qp, kp, first, last, None)
    assert blk(D=32) == 3
    assert blk(Lk=0) == 2
    assert blk(flags=4) == 4
    assert blk(qp=-1) == 4 and b"positions" in L.fa_last_error()
    assert blk(o=None) == 5            # last needs dO
    assert blk(acc=None, last=0) == 5  # state needed unless first and last
    assert L.fa_forward_general_host_half(h, h, h, h, 1, 1, 0, 4, 128, 0, None, None) == 2  # as the device form
    assert L.fa_forward_general_host_half(None, h, h, h, 1, 1, 4, 4, 128, 0, None, None) == 5


def test_python_front_end_refuses_cpu_tensors(fa):
    import torch

    q = torch.zeros((1, 1, 64, 128), dtype=torch.float16)
    with pytest.raises(fa.FlashAttentionError):
        fa.attention(q, q, q)


def test_flop_formula(fa):
    # benchmark_tensor_flash_attention.cu:218
    assert fa.attention_flops(32, 16, 512, 128) == 4.0 * 512 * 512 * 128 * 16 * 32
    assert round(fa.attention_flops(1, 16, 16384, 128) / 1e12, 3) == 2.199


def test_varlen_and_noalloc_validation(fa):
    """Bad arguments of the padded-batch and the no-allocation entry points are refused before any device work."""
    L = fa.lib()
    buf = np.zeros(16, dtype=np.float16)
    p = buf.ctypes.data_as(fa.HalfPtr)
    assert L.fa_forward_varlen(p, p, p, p, 1, 1, 64, 64, 96, 0, None, None, None, None) == 3
    assert L.fa_forward_varlen(p, p, p, p, 1, 1, 0, 64, 128, 0, None, None, None, None) == 2
    assert L.fa_forward_varlen(p, p, p, p, 0, 1, 64, 64, 128, 0, None, None, None, None) == 4
    assert L.fa_forward_varlen(p, p, p, p, 1, 1, 64, 64, 128, 2, None, None, None, None) == 4
    assert L.fa_forward_varlen(p, p, p, None, 1, 1, 64, 64, 128, 0, None, None, None, None) == 5
    assert L.fa_forward_noalloc(p, p, p, p, 1, 1, 64, 128, 32, None) == 1 and b"tile must be 64" in L.fa_last_error()
    assert L.fa_forward_noalloc(p, p, p, p, 1, 1, 100, 128, 64, None) == 2
    assert L.fa_forward_noalloc(p, p, p, None, 1, 1, 64, 128, 64, None) == 5
    assert L.fa_set_reserved_cus(-1) == 4


def test_sharded_host_entry_point_validation(fa):
    """Shape and pointer errors of the several-GPU host entry point come before any device work; on a machine
    without a GPU a well-formed call reports that (and does not crash)."""
    import torch

    L = fa.lib()
    buf = np.zeros(64 * 128, dtype=np.float16)
    p = buf.ctypes.data_as(fa.HalfPtr)
    sh = lambda B=1, H=1, Ls=64, D=128, tile=64, o=p: L.fa_forward_host_half_sharded(p, p, p, o, B, H, Ls, D, tile, None, 1, None)
    assert sh(tile=32) == 1 and sh(Ls=100) == 2 and sh(D=64) == 3 and sh(B=0) == 4 and sh(o=None) == 5
    if not torch.cuda.is_available():
        assert sh() == 6 and b"no GPU visible" in L.fa_last_error()
