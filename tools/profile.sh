#!/usr/bin/env bash
# Runs on the GPU box.  rocprofv3 passes over `python bench.py` for one README shape:
#   1. kernel trace + stats          (per-kernel durations)
#   2. SQ counters                   (MFMA busy, LDS bank conflicts, wave cycles)
#   3. FETCH_SIZE                    (HBM read side; 3 TCC slots)
#   4. WRITE_SIZE                    (HBM write side; 2 TCC slots)
# Counter passes never share a run with tracing.  Results land in gpurun_out/prof_L<L>/.
set -uo pipefail
L=${1:-512}
STEPS=${2:-20}
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_L$L
rm -rf "$OUT"; mkdir -p "$OUT"
# the trace pass keeps bench.py's preheat (steady-state clock, as in the reported number); the counter
# passes serialise every launch and read counters after each, so they run without it
TRACE_CMD=(python3 bench.py --seq-len "$L" --steps "$STEPS" --warmup 5 --no-cpu-baseline --no-sweep)
CMD=("${TRACE_CMD[@]}" --preheat-seconds 0)

timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "${TRACE_CMD[@]}" > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- "${CMD[@]}" > "$OUT/pmc_sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- "${CMD[@]}" > "$OUT/pmc_fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- "${CMD[@]}" > "$OUT/pmc_write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
find "$OUT" -name "*.csv" | head -20
echo "profile passes done for L=$L"
