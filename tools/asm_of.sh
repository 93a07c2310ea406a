#!/bin/bash
# Device assembly of one translation unit of the library with the library's flags:
#   tools/asm_of.sh draw.hip [extra -D flags]  ->  /tmp/<name>.s ; prints VGPR / scratch use per kernel
cd "$(dirname "$0")/../cora_amd/csrc" || exit 1
src=$1; shift
out=/tmp/$(basename "$src" .hip).s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result \
      -S --cuda-device-only "$@" "$src" -o "$out" 2>&1 | grep -v "hip-link" | head -30
grep -E "^\s*\.set .*\.(num_vgpr|private_seg_size)," "$out" | sed -e 's/^\s*\.set //' | paste - - | awk '{print $1, $2, $4}'
