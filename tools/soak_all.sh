#!/usr/bin/env bash
# Runs on the GPU box: the GPU test tier, every soak run, smoke() and the default bench line, each under its own time
# limit, stopping at the first failure; logs under gpurun_out/soak/.
set -uo pipefail
cd /root/repo
mkdir -p gpurun_out/soak
run() { name=$1; shift; timeout -k 10 "$@" > gpurun_out/soak/$name.log 2>&1; rc=$?; echo "$name rc=$rc: $(tail -1 gpurun_out/soak/$name.log)"; return $rc; }
run tests 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread && \
run fuzz_varlen 280 python -u tests/soak/fuzz_general.py --varlen --iters 240 --seed 11 && \
run fuzz_general 200 python -u tests/soak/fuzz_general.py --iters 200 --seed 12 && \
run fuzz_tiles 200 python -u tests/soak/fuzz_general.py --tiles --iters 100 --seed 13 && \
run fuzz_kernels 200 python -u tools/fuzz_kernels.py && \
run smoke 120 python -u -c "import __graft_entry__ as g; g.smoke()" && \
run bench 300 python -u bench.py && \
FA_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 run bench_dist 300 python -u bench.py --gpus 1 --no-cpu-baseline --no-sweep
