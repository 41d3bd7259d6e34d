#!/usr/bin/env python3
"""Condenses gpurun_out/prof_L<L>/ (written by tools/profile.sh) into profiles/<tag>_L<L>.json
and copies the rocprofv3 kernel-stats CSV next to it.

HBM bytes follow the gfx950 rules for rocprofv3: FETCH_SIZE and WRITE_SIZE are in KiB, collected
in separate passes, and FETCH_SIZE reports half the bytes of wide (16 B per lane) coalesced reads,
so the read side is doubled.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs * kernel cycles)
with kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    acc = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(path)):
        if "fa_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                      "Accum_VGPR_Count", "SGPR_Count", "Grid_Size", "Scratch_Size")}
    return {k: sum(v) / len(v) for k, v in acc.items()}, meta


def newest(pattern):
    """gpurun merges every run's files into the same directory: take the latest run's."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def main():
    L = int(sys.argv[1])
    tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_L{L}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(dst, f"{tag}_L{L}_kernel_stats.csv"))
    kern = None
    for r in csv.DictReader(open(stats)):
        if "fa_fwd" in r["Name"]:
            kern = r
    sq, meta = counters(newest(os.path.join(src, "pmc_sq", "*", "*_counter_collection.csv")))
    fetch, _ = counters(newest(os.path.join(src, "pmc_fetch", "*", "*_counter_collection.csv")))
    write, _ = counters(newest(os.path.join(src, "pmc_write", "*", "*_counter_collection.csv")))
    B, H, D = 16384 // L, 16, 128
    flops = 4.0 * B * H * L * L * D
    avg_ns = float(kern["AverageNs"])
    cycles = sq["GRBM_GUI_ACTIVE"] / 8.0
    read_b = 2.0 * fetch["FETCH_SIZE"] * 1024.0
    write_b = write["WRITE_SIZE"] * 1024.0
    out = {
        "command": f"tools/profile.sh {L} STEPS: rocprofv3 --kernel-trace --stats -- python3 bench.py --seq-len {L} --steps STEPS "
                   "--warmup 5 --no-cpu-baseline --no-sweep; counter passes (--pmc, one per run) add --preheat-seconds 0",
        "workload": f"B={B} H={H} L={L} D={D}",
        "kernel": kern["Name"],
        "launches": int(kern["Calls"]),
        "avg_kernel_us": avg_ns / 1e3,
        "min_kernel_us": float(kern["MinNs"]) / 1e3,
        "tflops_at_avg": flops / avg_ns / 1e3,
        "frac_of_2500_tflops": flops / avg_ns / 1e3 / 2500.0,
        "resources": meta,
        "counters_per_launch": {**sq, **fetch, **write},
        "kernel_cycles": cycles,
        "effective_clock_ghz_during_counter_pass": cycles / (avg_ns),
        "mfma_busy_frac": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles),
        "lds_bank_conflict_cycles": sq["SQ_LDS_BANK_CONFLICT"],
        "lds_active_frac": sq["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles),
        "wave_wait_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
        "hbm_read_bytes": read_b,
        "hbm_write_bytes": write_b,
        "hbm_bytes": read_b + write_b,
        "algorithmic_bytes": 4.0 * B * H * L * D * 2,
        "hbm_GBps_at_avg": (read_b + write_b) / avg_ns,
    }
    path = os.path.join(dst, f"{tag}_L{L}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("workload", "avg_kernel_us", "tflops_at_avg", "mfma_busy_frac",
                                          "lds_bank_conflict_cycles", "hbm_bytes", "algorithmic_bytes",
                                          "hbm_GBps_at_avg", "wave_wait_frac")}, indent=1))


if __name__ == "__main__":
    main()
