#!/usr/bin/env bash
# build_w64_only.sh NAME [extra hipcc flags...] -> build_ab/NAME.so (+ NAME.s): the default kernel alone behind
# fa_forward (tools/micro/w64_only.hip), for quick experiments; the register audit runs on that kernel
# (FA_AUDIT_ALLOW_FAIL=1: build anyway, for diagnostic builds whose stamps spill).
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
name=$1; shift
mkdir -p build_ab
FA_AUDIT_SRC=tools/micro/w64_only.hip FA_AUDIT_KERNELS=_ZN2fa17fa_fwd_w64_kernelENS_9FwdParamsE \
    FA_AUDIT_KEEP=build_ab/$name.s python3 tools/check_owned_regs.py -DFA_W64_ONLY "$@" ||
    { [ -n "${FA_AUDIT_ALLOW_FAIL:-}" ] && echo "(audit failed: diagnostic build, allowed)"; }
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -fPIC -shared -Wall -Wno-unused-result \
    -DFA_W64_ONLY "$@" -o build_ab/$name.so tools/micro/w64_only.hip
echo "built build_ab/$name.so"
