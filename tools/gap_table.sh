#!/usr/bin/env bash
# Runs on the GPU box.  Where inside a phase the cycles go: builds gs_p<P>_g<G> (tools/build_w64_only.sh with -DFA_STAMPS
# -DW64_FINE_STAMPS=40 -DW64_GAP_STAMP=<G> -DW64_GAP_STAMP_PHASE=<P> and the mask that keeps the phase's start stamp
# only; gs_p<P>_end keeps start and end) each give the cycles from the phase's start to gap G in steady-state steps
# of L = 16384; the differences between them are the cycles of gaps 0-7, 8-15, 16-23, 24-30 and the tail.
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
L=${1:-16384}
for P in 1 2; do
  if [ $P = 1 ]; then MASK=0x06; else MASK=0x18; fi
  for V in g8 g16 g24 g31 end; do
    f=build_ab/gs_p${P}_$V.so
    [ -f $f ] || continue
    timeout -k 10 120 python3 -u tools/stamps_w64.py $f $L pattern $MASK > gpurun_out/gap_p${P}_$V.txt 2>&1 || { echo "failed: $f"; tail -3 gpurun_out/gap_p${P}_$V.txt; exit 1; }
    python3 - gpurun_out/gap_p${P}_$V.txt $P $V <<'PY'
import sys, statistics
rows = []
for l in open(sys.argv[1]):
    t = l.split()
    if len(t) >= 4 and t[0].isdigit() and "|" in l:
        rows.append(float(t[2]))   # second column of the mask: cycles from the phase's start stamp to this one
print(f"phase {sys.argv[2]} {sys.argv[3]:>4s}: median over {len(rows)} steps {statistics.median(rows):7.0f} cycles from the phase's start (min {min(rows):.0f}, max {max(rows):.0f})")
PY
  done
done
