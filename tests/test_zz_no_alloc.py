"""GPU tier: fa_forward_noalloc (the entry point that allocates nothing; its split launches take their workspace
from a buffer inside the library's device image) against fa_forward, through the C ABI."""
import ctypes as ct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _call(fa, fn_name, q, k, v, o, stream=None):
    import torch

    B, H, L, D = q.shape
    ptr = lambda t: ct.cast(ct.c_void_p(t.data_ptr()), fa.HalfPtr)
    s = ct.c_void_p((stream or torch.cuda.current_stream(q.device)).cuda_stream)
    fn = getattr(fa.lib(), fn_name)
    if fn_name == "fa_forward":
        return fn(ptr(q), ptr(k), ptr(v), ptr(o), B, H, L, D, 64, s, None)
    return fn(ptr(q), ptr(k), ptr(v), ptr(o), B, H, L, D, 64, s)


def _inputs(dev, seed, shape):
    import torch

    g = torch.Generator(device=dev).manual_seed(seed)
    return [torch.randn(shape, generator=g, device=dev, dtype=torch.float32).to(torch.float16) for _ in range(3)]


def test_noalloc_matches_fa_forward_on_two_calls_in_a_row(fa, oracle, dev, fa_env):
    """Two calls in a row on the shared scratch buffer, the second on a smaller input: a large split launch (B=1,
    H=2, L=16384 in two key ranges: the whole buffer), then a smaller one (H=1, L=4096 in eight ranges), then one
    that needs no scratch.  Each gives the bits of fa_forward on the same inputs; the first also passes the
    parity gate against the oracle on sampled rows."""
    import torch

    fa_env(FA_SPLIT=None)
    shapes = [(1, 2, 16384, 128), (1, 1, 4096, 128), (2, 16, 1024, 128)]
    ways = [fa.lib().fa_split_ways(s[0] * s[1], s[2], s[2]) for s in shapes]
    assert ways[0] == 2 and ways[1] > 1 and ways[2] == 1, ways
    data = [_inputs(dev, 7 + i, s) for i, s in enumerate(shapes)]
    expect = []
    for q, k, v in data:
        o = torch.empty_like(q)
        assert _call(fa, "fa_forward", q, k, v, o) == 0, fa.lib().fa_last_error()
        expect.append(o)
    torch.cuda.synchronize()
    got = [torch.full_like(d[0], float("nan")) for d in data]
    for (q, k, v), o in zip(data, got):  # back to back on one stream, no synchronisation in between
        assert _call(fa, "fa_forward_noalloc", q, k, v, o) == 0, fa.lib().fa_last_error()
    torch.cuda.synchronize()
    for e, g_, s in zip(expect, got, shapes):
        assert torch.isfinite(g_.float()).all(), s
        assert torch.equal(e, g_), s
    # and once more in the other order (small after large was above; now large after small)
    for i in (1, 0):
        q, k, v = data[i]
        o = torch.full_like(q, float("nan"))
        assert _call(fa, "fa_forward_noalloc", q, k, v, o) == 0
        torch.cuda.synchronize()
        assert torch.equal(expect[i], o), shapes[i]
    q, k, v = data[0]
    rows = np.unique(np.linspace(0, 16383, 24).astype(np.int32))
    ref = oracle.attention_cpu_rows(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), rows)
    m = oracle.error_metrics(ref, got[0][:, :, rows].cpu().numpy())
    assert m["max_abs"] <= 2e-3 and m["global_rel"] <= 1e-3, m


def test_noalloc_rejects_what_does_not_fit_and_bad_arguments(fa, dev, fa_env):
    """B=1, H=3, L=16384 forced into four key ranges needs 12 * 16384 * 520 bytes: more than the buffer holds.  The
    call returns FA_ERR_WORKSPACE with the sizes in the error text and launches nothing; fa_forward takes the shape."""
    import torch

    fa_env(FA_SPLIT="4")
    q, k, v = _inputs(dev, 3, (1, 3, 16384, 128))
    o = torch.full_like(q, 5.0)
    rc = _call(fa, "fa_forward_noalloc", q, k, v, o)
    torch.cuda.synchronize()
    assert rc == 7, rc
    msg = fa.lib().fa_last_error().decode()
    assert "102236160" in msg and "34078720" in msg, msg
    assert bool((o == 5.0).all())
    assert _call(fa, "fa_forward", q, k, v, o) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    fa_env(FA_SPLIT=None)
    L_ = fa.lib()
    z = ct.cast(ct.c_void_p(q.data_ptr()), fa.HalfPtr)
    assert L_.fa_forward_noalloc(z, z, z, z, 1, 1, 100, 128, 64, None) == 2   # L % 64
    assert L_.fa_forward_noalloc(z, z, z, z, 1, 1, 64, 64, 64, None) == 3     # D
    assert L_.fa_forward_noalloc(None, z, z, z, 1, 1, 64, 128, 64, None) == 5  # NULL pointer


def test_noalloc_takes_the_best_split_that_fits_or_none(fa, oracle, dev, fa_env):
    """Shapes whose cost-model split would not fit the static buffer (B=1, H=1, L=7744: eleven ranges of 85 184
    rows; B=1, H=3, L=24576: three ranges of 221 184 rows) are no error: the entry point takes the best split that
    fits, or launches undivided, which needs no scratch at all.  Checked against the oracle on sampled rows."""
    import torch

    fa_env(FA_SPLIT=None)
    for shape, seed in (((1, 1, 7744, 128), 21), ((1, 3, 24576, 128), 22)):
        q, k, v = _inputs(dev, seed, shape)
        o = torch.full_like(q, float("nan"))
        rc = _call(fa, "fa_forward_noalloc", q, k, v, o)
        assert rc == 0, (shape, rc, fa.lib().fa_last_error())
        torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all(), shape
        rows = np.unique(np.linspace(0, shape[2] - 1, 16).astype(np.int32))
        ref = oracle.attention_cpu_rows(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), rows)
        m = oracle.error_metrics(ref, o[:, :, rows].cpu().numpy())
        assert m["max_abs"] <= 2e-3 and m["global_rel"] <= 1e-3, (shape, m)


def test_noalloc_allocates_nothing(fa, dev, fa_env):
    """The stream-ordered pool's high-water mark of used memory, reset before the call: fa_forward on a shape that
    divides its keys raises it (its workspace comes from that pool; this shows the probe sees an allocation),
    fa_forward_noalloc on the same shape leaves it at zero."""
    import torch

    fa_env(FA_SPLIT=None)
    hip = ct.CDLL("libamdhip64.so")
    pool = ct.c_void_p()
    assert hip.hipDeviceGetDefaultMemPool(ct.byref(pool), 0) == 0
    kUsedMemHigh = 8  # hipMemPoolAttrUsedMemHigh

    def reset():
        zero = ct.c_uint64(0)
        assert hip.hipMemPoolSetAttribute(pool, kUsedMemHigh, ct.byref(zero)) == 0

    def high():
        val = ct.c_uint64(0)
        assert hip.hipMemPoolGetAttribute(pool, kUsedMemHigh, ct.byref(val)) == 0
        return val.value

    q, k, v = _inputs(dev, 5, (1, 2, 8192, 128))
    assert fa.lib().fa_split_ways(2, 8192, 8192) > 1
    o = torch.empty_like(q)
    assert _call(fa, "fa_forward_noalloc", q, k, v, o) == 0  # (first call on the device done)
    torch.cuda.synchronize()
    reset()
    assert _call(fa, "fa_forward", q, k, v, o) == 0
    torch.cuda.synchronize()
    assert high() > 0, "the probe does not see fa_forward's workspace"
    reset()
    for _ in range(3):
        assert _call(fa, "fa_forward_noalloc", q, k, v, o) == 0
    torch.cuda.synchronize()
    assert high() == 0, high()


def test_split_shapes_under_stream_capture(fa, dev, fa_env):
    """A shape that divides its keys, captured in a graph: fa_forward launches it unsplit there (no allocation inside a
    capture), fa_forward_noalloc launches the split form on the static scratch buffer; both replay to the eager bits."""
    import torch

    fa_env(FA_SPLIT=None)
    q, k, v = _inputs(dev, 11, (1, 2, 8192, 128))
    assert fa.lib().fa_split_ways(2, 8192, 8192) > 1
    eager = torch.empty_like(q)
    assert _call(fa, "fa_forward", q, k, v, eager) == 0
    warm = torch.empty_like(q)
    assert _call(fa, "fa_forward_noalloc", q, k, v, warm) == 0  # first call on the device: outside any capture
    torch.cuda.synchronize()
    o1, o2 = torch.zeros_like(q), torch.zeros_like(q)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        assert _call(fa, "fa_forward", q, k, v, o1) == 0
        assert _call(fa, "fa_forward_noalloc", q, k, v, o2) == 0
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(o2, eager)                       # the same kernels as the eager split launch
    assert (o1.float() - eager.float()).abs().max().item() <= 1e-3  # unsplit: other summation order
