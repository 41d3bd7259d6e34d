#!/usr/bin/env bash
# causal_pmc.sh L: FETCH_SIZE / WRITE_SIZE passes over tools/micro/causal_once.py (HBM bytes per launch of the masked kernel)
set -uo pipefail
L=$1
cd "$(dirname "${BASH_SOURCE[0]}")/../.."
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=gpurun_out/cpmc_${C}_$L
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$OUT" -- python3 tools/micro/causal_once.py $L > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
  python3 - "$OUT" $C $L <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "fa_fwd" in r["Kernel_Name"]]
print(f"L={sys.argv[3]} {sys.argv[2]}: {sum(v)/len(v):.5g} per launch over {len(v)} launches (KiB per the counter's unit; the guide's gfx950 rule doubles FETCH_SIZE)")
PY
done
