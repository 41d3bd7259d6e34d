#!/usr/bin/env bash
# pmc_pass.sh L "COUNTERS..." : one rocprofv3 counter pass over bench.py for one README shape
set -uo pipefail
L=$1; shift
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$$
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 240 rocprofv3 --pmc $@ --output-format csv -d "$OUT" -- python3 bench.py --seq-len "$L" --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --preheat-seconds 0 > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "fa_fwd" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"  {k:30s} {sum(v)/len(v):.4g}")
PY
