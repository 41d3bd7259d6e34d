#!/usr/bin/env python3
"""One shape through fa_forward_general with a forced key split (for a kernel trace): python tools/split_one.py D BH Lq Lk S"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
D, BH, Lq, Lk, S = [int(x) for x in sys.argv[1:6]]
os.environ["FA_SPLIT"] = str(S)
fa = ge.load_package()
dev = torch.device("cuda:0")
q = torch.empty((1, BH, Lq, D), dtype=torch.float16, device=dev)
k = torch.empty((1, BH, Lk, D), dtype=torch.float16, device=dev)
v = torch.empty_like(k)
for t, s in zip((q, k, v), (1, 2, 3)):
    fa.init_half(t, s)
o = torch.empty_like(q)
for _ in range(200):
    fa.attention_general(q, k, v, out=o)
torch.cuda.synchronize()
