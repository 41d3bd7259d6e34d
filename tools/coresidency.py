#!/usr/bin/env python3
"""Can another stream's kernel run while the persistent attention kernels hold every compute unit?

    python tools/coresidency.py [--reserved 0 8 16 32] [--blocks 16] [--repeat 1]

Stream A runs one rank's compute of the 8-way ring at L = 131072 (8 blockwise calls of 16 heads x 16384 rows x
16384 keys, about 15 ms).  Stream B, once A's first kernel is running, launches a kernel shaped like a collective
library's transport kernel (tests/cpp/side_copy.hip: `--blocks` workgroups of 512 threads with 32 KiB of LDS each,
copying 134 MB device-to-device, the bytes of one ring step's K/V hand-off).  Printed per setting of
fa_set_reserved_cus: when the side kernel finished relative to A's start, A's own duration, and both alone."""
import argparse, ctypes as ct, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ap = argparse.ArgumentParser()
ap.add_argument("--reserved", type=int, nargs="+", default=[0, 8, 16, 32])
ap.add_argument("--blocks", type=int, nargs="+", default=[16])
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--steps", type=int, default=8)
args = ap.parse_args()
fa = ge.load_package()
side = ct.CDLL(os.path.join(ROOT, "tests", "cpp", "libside_copy.so"))
side.side_copy_launch.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p]
dev = torch.device("cuda:0")
H, Lb, D = 16, 16384, 128
q, k, v = [torch.empty((1, H, Lb, D), dtype=torch.float16, device=dev) for _ in range(3)]
for t, seed in zip((q, k, v), (1234, 2345, 3456)):
    fa.init_half(t, seed)
src = torch.empty(2 * H * Lb * D, dtype=torch.float16, device=dev).normal_()
dst = torch.empty_like(src)
nbytes = src.numel() * 2
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
st = fa.BlockState(H, Lb, D, dev)


def attention_pass():
    st.first = True
    for s in range(args.steps):
        fa.attention_block(q, k, v, st, last=(s == args.steps - 1), stream=sa)


def side_kernel(blocks):
    rc = side.side_copy_launch(src.data_ptr(), dst.data_ptr(), nbytes, blocks, 512, 32768, args.repeat, sb.cuda_stream)
    assert rc == 0


def ev():
    return torch.cuda.Event(enable_timing=True)


for _ in range(3):
    attention_pass()
torch.cuda.synchronize()
for blocks in args.blocks:
    e0, e1 = ev(), ev()
    e0.record(sb); side_kernel(blocks); e1.record(sb); torch.cuda.synchronize()
    e0.record(sb); side_kernel(blocks); e1.record(sb); torch.cuda.synchronize()
    print(f"side kernel alone ({blocks} workgroups x 512 threads, 32 KiB LDS, {nbytes / 1e6:.0f} MB x {args.repeat}): "
          f"{e0.elapsed_time(e1):.3f} ms")
    for kres in args.reserved:
        fa.set_reserved_cus(kres)
        for _ in range(2):
            attention_pass()
        torch.cuda.synchronize()
        a0, a1 = ev(), ev()
        a0.record(sa); attention_pass(); a1.record(sa); torch.cuda.synchronize()
        alone = a0.elapsed_time(a1)
        a0, a1, a_first, b1 = ev(), ev(), ev(), ev()
        a0.record(sa)
        st.first = True
        fa.attention_block(q, k, v, st, last=False, stream=sa)
        a_first.record(sa)
        for s in range(1, args.steps):
            fa.attention_block(q, k, v, st, last=(s == args.steps - 1), stream=sa)
        a1.record(sa)
        sb.wait_event(a0)  # the side kernel becomes ready when the attention pass starts
        side_kernel(blocks)
        b1.record(sb)
        torch.cuda.synchronize()
        print(f"  reserved_cus={kres:3d}: attention alone {alone:7.3f} ms; with the side kernel {a0.elapsed_time(a1):7.3f} ms; "
              f"side kernel done at {a0.elapsed_time(b1):7.3f} ms (first attention kernel done at {a0.elapsed_time(a_first):6.3f} ms)",
              flush=True)
fa.set_reserved_cus(0)
