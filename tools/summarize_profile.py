#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<TAG>/ (written by tools/profile.sh) into profiles/<round>_<TAG>.json
and copies the rocprofv3 kernel-stats CSV next to it.

    python tools/summarize_profile.py L [round] [TAG]

HBM bytes follow the gfx950 rules for rocprofv3: FETCH_SIZE and WRITE_SIZE are in KiB, collected
in separate passes, and FETCH_SIZE reports half the bytes of wide (16 B per lane) coalesced reads,
so the read side is doubled.

Cycles.  Every pass runs the same command (preheat included), and the counter pass carries its own launch
timestamps, so nothing is mixed between passes: the kernel's cycles are SQ_WAVE_CYCLES (counted in units of four
cycles, summed over waves) per wave -- every wave of this persistent kernel lives from launch to exit -- and the
clock is those cycles over the SAME pass's launch duration.  (GRBM_GUI_ACTIVE / 8 is also reported; its window
is wider than the kernel's, it includes the dispatch and the end-of-kernel write-back.)
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 1024


def counters(path):
    acc = collections.defaultdict(list)
    dur = {}
    meta = {}
    for r in csv.DictReader(open(path)):
        if "fa_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            meta = {k: r[k] for k in ("Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                      "Accum_VGPR_Count", "SGPR_Count", "Grid_Size", "Scratch_Size")}
    return {k: sum(v) / len(v) for k, v in acc.items()}, meta, (sum(dur.values()) / max(1, len(dur)))


def newest(pattern):
    """gpurun merges every run's files into the same directory: take the latest run's."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def main():
    L = int(sys.argv[1])
    rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
    tag = sys.argv[3] if len(sys.argv) > 3 else f"L{L}"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv"))
    kern = None
    for r in csv.DictReader(open(stats)):
        if "fa_fwd" in r["Name"] and kern is None:
            kern = r
    sq, meta, sq_ns = counters(newest(os.path.join(src, "pmc_sq", "*", "*_counter_collection.csv")))
    fetch, _, _ = counters(newest(os.path.join(src, "pmc_fetch", "*", "*_counter_collection.csv")))
    write, _, _ = counters(newest(os.path.join(src, "pmc_write", "*", "*_counter_collection.csv")))
    tcc = {}
    try:
        tcc, _, _ = counters(newest(os.path.join(src, "pmc_tcc", "*", "*_counter_collection.csv")))
    except Exception:
        pass
    B, H, D = max(1, 16384 // L), 16, 128
    flops = 4.0 * B * H * L * L * D
    avg_ns = float(kern["AverageNs"])
    n_waves = int(meta["Grid_Size"]) // 64
    cycles = 4.0 * sq["SQ_WAVE_CYCLES"] / n_waves
    read_b = 2.0 * fetch["FETCH_SIZE"] * 1024.0
    write_b = write["WRITE_SIZE"] * 1024.0
    out = {
        "command": "tools/profile.sh: rocprofv3 --kernel-trace --stats / --pmc <one counter set per run> -- "
                   + open(os.path.join(src, "command.txt")).read().strip(),
        "workload": f"B={B} H={H} L={L} D={D}",
        "kernel": kern["Name"],
        "launches": int(kern["Calls"]),
        "avg_kernel_us": avg_ns / 1e3,
        "min_kernel_us": float(kern["MinNs"]) / 1e3,
        "tflops_at_avg": flops / avg_ns / 1e3,
        "frac_of_2500_tflops": flops / avg_ns / 1e3 / 2500.0,
        "resources": meta,
        "counters_per_launch": {**sq, **fetch, **write, **tcc},
        "counter_pass_kernel_us": sq_ns / 1e3,
        "kernel_cycles": cycles,
        "kernel_cycles_source": "4 * SQ_WAVE_CYCLES / waves (every wave lives from launch to exit)",
        "gui_active_cycles_per_xcd": sq["GRBM_GUI_ACTIVE"] / 8.0,
        "effective_clock_ghz": cycles / sq_ns,
        "mfma_busy_frac": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cycles),
        "mfma_busy_frac_of_gui_active": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * sq["GRBM_GUI_ACTIVE"] / 8.0),
        "mfma_busy_cycles_expected_from_flops": flops / 32768.0 * 32.0,
        "lds_bank_conflict_cycles": sq["SQ_LDS_BANK_CONFLICT"],
        "lds_bank_conflict_frac_of_lds_active": sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"],
        "lds_active_frac": sq["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles),
        "wave_wait_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
        "hbm_read_bytes": read_b,
        "hbm_write_bytes": write_b,
        "hbm_bytes": read_b + write_b,
        "algorithmic_bytes": 4.0 * B * H * L * D * 2,
        "hbm_GBps_at_avg": (read_b + write_b) / avg_ns,
    }
    timed = os.path.join(src, "trace_timed.json")
    if os.path.exists(timed):
        # the bench's timed region inside the trace pass: its launches' own durations, their period (launch to
        # launch) and the line the bench printed in that run (wall clock per step, HIP-event time per launch)
        out["timed_region_of_the_trace_pass"] = json.load(open(timed))
    if "TCC_HIT_sum" in tcc and "TCC_MISS_sum" in tcc:
        out["l2_hit_rate"] = tcc["TCC_HIT_sum"] / max(1.0, tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"])
    path = os.path.join(dst, f"{rnd}_{tag}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("workload", "avg_kernel_us", "counter_pass_kernel_us", "tflops_at_avg",
                                          "effective_clock_ghz", "mfma_busy_frac", "lds_bank_conflict_cycles", "hbm_bytes",
                                          "algorithmic_bytes", "hbm_GBps_at_avg", "wave_wait_frac")}, indent=1))


if __name__ == "__main__":
    main()
