"""GPU tier: bench.py prints ONE JSON line with the fields its callers rely on."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_line():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--preheat-seconds", "0.2",
           "--no-sweep", "--cpu-seconds", "1"]  # (keeps normal_data_tflops / reference_protocol_tflops)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["unit"] == "TFLOP/s" and j["value"] > 0 and j["ms_per_step"] > 0
    assert j["dtype"] == "f16" and j["data"] == "synthetic" and j["scaling"] == "weak"
    assert "workload" in j["config"] and "model" not in j["config"]
    # value and ms_per_step describe the same steps: FLOPs of the L=512 README shape
    assert abs(j["value"] - 4 * 32 * 16 * 512 * 512 * 128 / (j["ms_per_step"] * 1e-3) / 1e12) < 1e-6 * j["value"]
    rf = j["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    assert rf["kernel"] == "fa::fa_fwd_w64_kernel" and rf["kernel_ms"] > 0
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert j["normal_data_tflops"] > 0 and j["reference_protocol_tflops"] > 0 and j["pct_of_fp16_peak"] > 0
    assert "cold_inputs_tflops" in j and (j["cold_inputs_tflops"] is None or j["cold_inputs_tflops"] > 0)
    assert j["per_rank_kernel_ms"] == [rf["kernel_ms"]] and rf["key_split_ways"] == 1
    # provenance: where the traffic figure comes from, and the clock the kernel ran at on either data set
    assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
    if rf["traffic"] is not None:
        assert rf["traffic_source"].startswith("profiles/") and "not measured in this run" in rf["traffic_source"]
    assert 1.0 < j["clock_ghz_effective"] < 2.6 and 1.0 < j["clock_ghz_effective_normal_data"] < 2.6
    assert j["cycles_per_workgroup"] > 5e4 and "s_memtime" in j["clock_source"]


def _bench(*argv, env=None, timeout=280):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + list(argv)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_strong_scaling_shape_on_one_gpu():
    """--scaling strong --seq-len 16384: B=1, H=16 (the heads would be split over the ranks; one rank holds all)."""
    j = _bench("--seq-len", "16384", "--scaling", "strong", "--steps", "3", "--warmup", "1", "--preheat-seconds", "0.2",
               "--no-sweep", "--no-cpu-baseline", "--no-extras")
    assert j["scaling"] == "strong" and j["n_gpus"] == 1 and j["config"]["per_gpu_heads"] == 16
    assert "L=16384" in j["config"]["workload"] and j["roofline"]["bound"] == "mfma"
    assert abs(j["value"] - 4 * 16 * 16384 * 16384 * 128 / (j["ms_per_step"] * 1e-3) / 1e12) < 1e-6 * j["value"]


def test_bench_ring_mode_on_one_gpu():
    """--ring on one rank: the whole sequence is one block; the line carries the ring's timing fields."""
    j = _bench("--ring", "--seq-len", "8192", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert j["scaling"] == "strong" and j["n_gpus"] == 1 and "ring" in j["config"]["workload"]
    assert j["ring"]["steps_per_pass"] == 1 and j["ring"]["exchange_only_ms_per_ring_step"] is None
    assert j["ring"]["compute_only_ms_per_ring_step"] > 0 and j["value"] > 0


def test_bench_multi_rank_code_path_with_one_rank():
    """The process-group path of bench.py (RCCL init, barriers, max-over-ranks) on a one-GPU box."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FA_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3",
           "--preheat-seconds", "0.3", "--no-sweep", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["scaling"] == "weak"
    # the multi-rank default line also carries the strong-scaling point: here one rank holds all 16 heads
    c4 = j["strong_c4"]
    assert "error" not in c4, c4
    assert c4["per_gpu_heads"] == 16 and c4["key_split_ways"] == 1 and c4["tflops_total"] > 0
    assert 0.8 < c4["efficiency_vs_N1_rate"] < 1.25  # the same 16 heads timed twice
