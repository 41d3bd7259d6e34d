#!/usr/bin/env bash
# Builds libflash_attention.so for MI355X.  Counterpart of the reference's
# flash_attention_tensor_cores/shared_compile.sh (nvcc -arch=sm_86 -Xcompiler
# -fPIC -shared): same artefact name, same OUT_LIB override, gfx950 only.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT_LIB=${OUT_LIB:-libflash_attention.so}
"$HIPCC" -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -fPIC -shared \
    -Wall -Wno-unused-result \
    -o "$OUT_LIB" flash_attention_capi.hip
echo "built $OUT_LIB"
python3 ../../tools/check_owned_regs.py
