#!/usr/bin/env bash
# Runs on the GPU box.  rocprofv3 passes over `python bench.py` for one README shape:
#   1. kernel trace + stats          (per-kernel durations)
#   2. SQ counters                   (MFMA busy, LDS bank conflicts, wave cycles)
#   3. FETCH_SIZE                    (HBM read side; 3 TCC slots)
#   4. WRITE_SIZE                    (HBM write side; 2 TCC slots)
# Counter passes never share a run with tracing.  Every pass runs the SAME command, preheat included, so that
# the counters and the durations describe launches at the same clock.  Results land in gpurun_out/prof_<TAG>/.
#   [PREHEAT=seconds] tools/profile.sh L [STEPS] [TAG] [extra bench.py arguments, e.g. --data normal]
set -uo pipefail
L=${1:-512}
STEPS=${2:-20}
TAG=${3:-L$L}
shift; shift || true; shift || true
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD=(python3 bench.py --seq-len "$L" --steps "$STEPS" --warmup 5 --preheat-seconds "${PREHEAT:-0.5}" --no-cpu-baseline --no-sweep --no-extras "$@")
echo "${CMD[*]}" > "$OUT/command.txt"

timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "${CMD[@]}" > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- "${CMD[@]}" > "$OUT/pmc_sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- "${CMD[@]}" > "$OUT/pmc_fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- "${CMD[@]}" > "$OUT/pmc_write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/pmc_tcc" -- "${CMD[@]}" > "$OUT/pmc_tcc.log" 2>&1 || { echo "tcc pass failed (optional)"; tail -3 "$OUT/pmc_tcc.log"; }
# the counter CSVs hold one row per counter and launch, preheat launches included: keep the attention kernel's
# last 200 launches only (what gpurun copies back is limited)
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/pmc_*/*/*_counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "fa_fwd" in r["Kernel_Name"] or "fa_split" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in keep})[-200:]
    keep = [r for r in keep if int(r["Dispatch_Id"]) in set(ids)]
    w = csv.DictWriter(open(f, "w", newline=""), fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
    w.writeheader(); w.writerows(keep)
PY
# the timed region by itself: the attention kernel's last STEPS launches in the trace (what the bench timed; the
# stats file averages over the preheat as well, where the clock is still coming up), and the line the bench
# printed in the same run
python3 - "$OUT" "$STEPS" <<'PY'
import csv, glob, json, sys
out, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + "/trace/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "fa_fwd_w64" in r["Kernel_Name"] or "fa_fwd_ring" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-steps:]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / len(last)
line = [l for l in open(out + "/trace.log") if l.startswith("{")]
bench = json.loads(line[-1]) if line else {}
json.dump({"timed_launches": len(last), "timed_kernel_us_avg": sum(dur) / len(dur) / 1e3, "timed_kernel_us_min": min(dur) / 1e3,
           "timed_period_us": span / 1e3, "bench_line_ms_per_step": bench.get("ms_per_step"),
           "bench_line_kernel_ms": (bench.get("roofline") or {}).get("kernel_ms"), "bench_line_value": bench.get("value")},
          open(out + "/trace_timed.json", "w"), indent=1)
print(open(out + "/trace_timed.json").read())
PY
rm -f "$OUT"/trace/*/*_kernel_trace.csv
find "$OUT" -name "*.csv" | head -20
echo "profile passes done for $TAG"
