#!/usr/bin/env python3
"""Diagnostic: masked tuned kernel and general kernel against the oracle on one ragged shape; where they differ most."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
import oracle
fa = ge.load_package()
dev = torch.device("cuda:0")
B, H, L, causal, scale = [(1, 8, 843, True, 1.34), (2, 4, 1410, False, 2.53), (1, 8, 843, False, 1.0)][int(sys.argv[1]) if len(sys.argv) > 1 else 0]
rng = np.random.default_rng(0)
Q = (rng.standard_normal((B, H, L, 128), dtype=np.float32) * scale + 0.3).astype(np.float16)
K = (rng.standard_normal((B, H, L, 128), dtype=np.float32) * scale).astype(np.float16)
V = (rng.standard_normal((B, H, L, 128), dtype=np.float32) * scale).astype(np.float16)
q, k, v = [torch.from_numpy(a).to(dev) for a in (Q, K, V)]
outs = {}
for name, env in (("masked", "1"), ("general", "0")):
    os.environ["FA_MASKED_W64"] = env
    fa.reload_env()
    print(name, fa.lib().fa_kernel_name_general(L, L, 128, int(causal)).decode())
    outs[name] = fa.attention_general(q, k, v, causal=causal).float().cpu().numpy()
ref = oracle.attention_cpu(Q, K, V, causal=causal)
for name, o in outs.items():
    e = np.abs(o - ref)
    at = np.unravel_index(np.argmax(e), e.shape)
    print(f"{name}: max|err| {e.max():.3e} at {tuple(int(i) for i in at)} (oracle {ref[at]:.5f}), rel L2 {np.linalg.norm(o-ref)/np.linalg.norm(ref):.3e}")
d = np.abs(outs["masked"] - outs["general"])
at = np.unravel_index(np.argmax(d), d.shape)
print(f"masked vs general: max {d.max():.3e} at {tuple(int(i) for i in at)}: masked {outs['masked'][at]:.5f} general {outs['general'][at]:.5f} oracle {ref[at]:.5f}")
rows = np.argwhere(d.max(axis=3) > 2.0 ** -9 * np.maximum(np.abs(outs['masked']), np.abs(outs['general'])).max(axis=3))
print("rows differing by more than 2^-9 of the row's largest output:", len(rows), "of", B * H * L)
print("  row positions (first 40):", sorted(set(int(r[2]) for r in rows))[:40])
rel = d / (np.maximum(np.abs(outs['masked']), np.abs(outs['general'])) + 1e-6)
at = np.unravel_index(np.argmax(rel * (d > 5e-4)), rel.shape)
print(f"largest relative difference among |diff| > 5e-4: {rel[at]:.3e} at {tuple(int(i) for i in at)}: masked {outs['masked'][at]:.5f} general {outs['general'][at]:.5f} oracle {ref[at]:.5f}")
