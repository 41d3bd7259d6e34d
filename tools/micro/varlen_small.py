#!/usr/bin/env python3
"""Padded batches of ONE to four long sequences (H = 16): where the tuned kernel's list form has fewer items than the
chip has compute units.  Same columns as tools/bench_varlen.py, whose helpers this runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "tools", "bench_varlen.py")).read().split('case("uniform 1100..4096"')[0]
src = src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(ROOT))
sys.argv = sys.argv[:1]
exec(src)
case("one of 4096", [4096], 4096)
case("one of 2048", [2048], 2048)
case("one of 1500", [1500], 1536)
case("two of 1500", [1500, 1400], 1536)
case("four of 1500", [1500, 1400, 1300, 1200], 1536)
case("one of 8192", [8192], 8192)
case("one of 4096", [4096], 4096, causal=True)
case("two of 4096", [4096, 3000], 4096, causal=True)
