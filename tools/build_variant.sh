#!/usr/bin/env bash
# build_variant.sh NAME [extra hipcc flags...] -> build_ab/NAME.so (an experiment build of the library)
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
name=$1; shift
mkdir -p build_ab
python3 tools/check_owned_regs.py "$@"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -fPIC -shared -Wall -Wno-unused-result "$@" \
    -o build_ab/$name.so flash-attention_amd/csrc/flash_attention_capi.hip
echo "built build_ab/$name.so"
