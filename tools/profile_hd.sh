#!/usr/bin/env bash
# Runs on the GPU box: rocprofv3 kernel trace over tools/bench_hd.py (head dims 64 and 256, L = 512, 2048 and 8192,
# the whole-tile kernel and, with FA_HD=0, the general kernel; then the causal mask (TFLOP/s of the visible triangle)
# and ragged lengths on the masked form (D = 128 included: what the tuned kernels leave) and on the general kernel).
# Results: gpurun_out/prof_hd/.
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_hd
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/bench_hd.py --shapes 512 2048 8192 > "$OUT/bench_hd.txt" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/bench_hd.txt"; exit 1; }
rm -f "$OUT"/trace/*/*_kernel_trace.csv
FA_HD=0 timeout -k 10 200 python3 tools/bench_hd.py --shapes 512 2048 8192 > "$OUT/bench_general.txt" 2>&1
for hd in 2 0; do
    FA_HD=$hd timeout -k 10 200 python3 tools/bench_hd.py --dims 64 128 256 --shapes 512 2048 8192 --causal >> "$OUT/bench_masked.txt" 2>&1
    FA_HD=$hd timeout -k 10 200 python3 tools/bench_hd.py --dims 64 128 256 --shapes 320 576 832 2048 8192 --lminus 1 >> "$OUT/bench_masked.txt" 2>&1
done
grep -h "TFLOP" "$OUT/bench_hd.txt" "$OUT/bench_general.txt" "$OUT/bench_masked.txt"
cat "$OUT"/trace/*/*_kernel_stats.csv
