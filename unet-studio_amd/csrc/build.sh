#!/bin/bash
# Builds libunet_hip.so for gfx950 in-tree (next to the Python package).  No GPU needed: hipcc cross-compiles.
set -e
cd "$(dirname "$0")"
SRCS="graph.cpp engine.cpp kernels_direct.hip kernels_elem.hip kernels_mfma.hip"
hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -Wno-unused-result -o ../libunet_hip.so $SRCS
echo "built $(cd .. && pwd)/libunet_hip.so"
