#!/usr/bin/env python3
"""One causal shape through fa_forward_general a few dozen times (for a rocprofv3 counter pass: HBM bytes of the masked
tuned kernel, whose item order puts no two workgroups on the same K/V at the same time).
    python tools/micro/causal_once.py L [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
fa = ge.load_package()
dev = torch.device("cuda:0")
L = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, H, D = max(1, 16384 // L), 16, 128
q, k, v, o = [torch.empty((B, H, L, D), dtype=torch.float16, device=dev) for _ in range(4)]
for t, seed in zip((q, k, v), (1234, 2345, 3456)):
    fa.init_half(t.view(-1), seed)
for _ in range(reps):
    fa.attention_general(q, k, v, causal=True, out=o)
torch.cuda.synchronize()
print("done", B, H, L)
