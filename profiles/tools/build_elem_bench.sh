#!/bin/bash
# builds unet-studio_amd/csrc/build/elem_bench against the in-tree libunet_hip.so (run after csrc/build.sh)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
hipcc -O2 --offload-arch=gfx950 -std=c++17 -x hip "$R/profiles/tools/elem_bench.cpp" -o "$R/unet-studio_amd/csrc/build/elem_bench" \
    -L"$R/unet-studio_amd" -l:libunet_hip.so -Wl,-rpath,'$ORIGIN/../..'
echo "built $R/unet-studio_amd/csrc/build/elem_bench"
