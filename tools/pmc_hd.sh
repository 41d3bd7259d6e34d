#!/usr/bin/env bash
# pmc_hd.sh D "COUNTERS...": one rocprofv3 counter pass over tools/bench_hd.py for one head dim (L = 8192)
set -uo pipefail
D=$1; shift
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_hd_$$
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 200 rocprofv3 --pmc $@ --output-format csv -d "$OUT" -- python3 tools/bench_hd.py --dims "$D" --shapes 8192 > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(list); dur = {}
for r in csv.DictReader(open(f)):
    if "fa_fwd" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        name, grid = r["Kernel_Name"], r["Grid_Size"]
print(name, "grid", grid, "avg us", sum(dur.values()) / len(dur) / 1e3)
for k, v in sorted(acc.items()):
    print(f"  {k:30s} {sum(v)/len(v):.6g}")
import shutil; shutil.rmtree(sys.argv[1])
PY
