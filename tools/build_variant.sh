#!/bin/bash
# Diagnostic: build an A/B variant of the library into gpurun_out/variants/<name>.so (select it with EU_LIB_PATH).
# Usage: tools/build_variant.sh <name> [extra hipcc flags...]
set -e
name="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/variants"
cd "$root/euclider_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function -shared \
  -o "$root/variants/$name.so" -x hip renderer.hip -x hip capi.cpp -x hip scene_host.cpp -x hip camera_host.cpp -x hip jit.cpp "$@" -Wl,-Bsymbolic -L/opt/rocm/lib -lhiprtc -ldl 2>&1 | grep -E "error" || true
ls -la "$root/variants/$name.so"
