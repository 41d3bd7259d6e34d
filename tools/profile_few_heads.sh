#!/usr/bin/env bash
# Runs on the GPU box: rocprofv3 kernel trace + MFMA counters over tools/few_heads.py for B=1, H heads, L=16384
# (what one of 16/H ranks holds of the long-sequence shape).  Results: gpurun_out/prof_H<H>/.
#   tools/profile_few_heads.sh [H]
set -uo pipefail
H=${1:-2}
cd "$(dirname "${BASH_SOURCE[0]}")/.."
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_H$H
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/few_heads.py --heads "$H" 16 --splits auto > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
rm -f "$OUT"/trace/*/*_kernel_trace.csv
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -- python3 tools/few_heads.py --heads "$H" --splits auto --preheat 0.3 > "$OUT/pmc.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
f = glob.glob(out + "/pmc/*/*_counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
acc, dur, waves = collections.defaultdict(list), collections.defaultdict(dict), {}
for r in rows[-6000:]:
    name = r["Kernel_Name"].split("(")[0]
    acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    dur[name][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    waves[name] = int(r["Grid_Size"]) // 64
summary = {}
for (name, c), v in acc.items():
    summary.setdefault(name, {})[c] = sum(v) / len(v)
for name, s in summary.items():
    s["counter_pass_kernel_us"] = sum(dur[name].values()) / len(dur[name]) / 1e3
    s["waves"] = waves[name]
    if "SQ_WAVE_CYCLES" in s:
        s["kernel_cycles"] = 4.0 * s["SQ_WAVE_CYCLES"] / waves[name]
        s["mfma_busy_frac"] = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * s["kernel_cycles"])
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
os.remove(f)
PY
tail -3 "$OUT/trace.log"
cat "$OUT"/trace/*/*_kernel_stats.csv
