#!/usr/bin/env bash
# Runs on the GPU box: the measurement tools behind the secondary numbers of the design notes, one after the other,
# each under its own time limit; outputs under gpurun_out/evidence/ (copied into profiles/ by hand).
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
mkdir -p gpurun_out/evidence
run() { name=$1; shift; timeout -k 10 "$@" > gpurun_out/evidence/$name.txt 2>&1; rc=$?; echo "$name rc=$rc"; return $rc; }
run bench_hd 300 python3 -u tools/bench_hd.py && \
run bench_general 400 python3 -u tools/bench_general.py && \
run decode_shapes 300 python3 -u tools/decode_shapes.py && \
run few_heads 300 python3 -u tools/few_heads.py && \
run coresidency 300 python3 -u tools/coresidency.py && \
run ring_one_gpu 400 python3 -u tests/soak/ring_one_gpu.py
