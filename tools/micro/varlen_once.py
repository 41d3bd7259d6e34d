#!/usr/bin/env python3
"""One padded batch through fa_forward_varlen a few times (for rocprofv3 --kernel-trace: what each of its kernels takes).
    python tools/micro/varlen_once.py LOW HIGH B [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
fa = ge.load_package()
dev = torch.device("cuda:0")
lo, hi, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(int(sys.argv[4]) if len(sys.argv) > 4 else 0)
lens = [int(x) for x in rng.integers(lo, hi + 1, B)]
H, D = 16, 128
q, k, v, o = [torch.empty((B, H, hi, D), dtype=torch.float16, device=dev) for _ in range(4)]
for t, seed in zip((q, k, v), (1234, 2345, 3456)):
    fa.init_half(t.view(-1), seed)
ql = torch.tensor(lens, dtype=torch.int32, device=dev)
for _ in range(300):
    fa.attention_varlen(q, k, v, ql, ql, out=o)
torch.cuda.synchronize()
print(lens)
