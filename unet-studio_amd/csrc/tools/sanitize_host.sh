#!/bin/bash
# Host-code sanitizers (the reference runs none).  CPU box only -- never on the GPU box (GPU ASan / XNACK
# runs are refused there): builds the HOST translation units of the engine (graph.cpp, engine.cpp, comm.cpp: DSL parser, graph
# lowering, plan layout, C ABI argument handling, profiler, communicator plumbing) and of the C++ host (unet_host.cpp, nz_io.cpp)
# with -fsanitize=address,undefined (g++; no device code is instrumented -- the kernel objects are linked as built), then runs
# the CPU test files that exercise them under the sanitizer runtimes:
#     tests/test_host.py  tests/test_augment_host.py  tests/test_nz.py  (+ the C++ .nz round trip binary)
# Usage: bash unet-studio_amd/csrc/tools/sanitize_host.sh          (from anywhere; exit code != 0 on any report)
set -e
cd "$(dirname "$0")/.."
bash build.sh > /dev/null
mkdir -p build/asan
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g -O1"
HIPINC="-I/opt/rocm/include -D__HIP_PLATFORM_AMD__"
for f in graph.cpp engine.cpp comm.cpp; do
    g++ -std=c++17 -fPIC $SAN $HIPINC -Wno-unused-result -c $f -o build/asan/${f%.*}.o
done
OBJS="build/asan/graph.o build/asan/engine.o build/asan/comm.o"
for f in kernels_direct kernels_elem kernels_mfma_conv kernels_mfma_conv_z16 kernels_mfma_wgrad kernels_mfma_wgrad_z kernels_augment kernels_mfma_f32; do OBJS="$OBJS build/$f.o"; done
g++ -shared -fPIC $SAN -o build/asan/libunet_hip.so $OBJS -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -ldl
T=$(python3 -c 'import torch, os; print(os.path.dirname(torch.__file__))')
ABI=$(python3 -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')
INC="-I$T/include -I$T/include/torch/csrc/api/include -I/opt/rocm/include -I../../include"
DEF="-D_GLIBCXX_USE_CXX11_ABI=$ABI -D__HIP_PLATFORM_AMD__ -DUSE_ROCM"
LIBS="-L$T/lib -Wl,-rpath,$T/lib -ltorch -ltorch_cpu -ltorch_hip -lc10 -lc10_hip -Lbuild/asan -Wl,-rpath,$PWD/build/asan -lunet_hip -lamdhip64 -lpthread -lz"
g++ -std=c++17 -O1 -g $SAN $DEF $INC unet_host.cpp nz_io.cpp ../../tests/cpp/test_nz_io.cpp -o build/asan/test_nz_io $LIBS
ASAN_RT=$(g++ -print-file-name=libasan.so)
UBSAN_RT=$(g++ -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
cd ../..
echo "== pytest under ASan + UBSan (host code of libunet_hip.so) =="
# CPython and PyTorch's OpenMP runtime are not instrumented: preload the runtimes, keep the interposed allocator quiet about them
LD_PRELOAD="$ASAN_RT $UBSAN_RT" UNET_HIP_LIBRARY=$PWD/unet-studio_amd/csrc/build/asan/libunet_hip.so \
    python3 -m pytest tests/test_host.py tests/test_augment_host.py tests/test_nz.py -x -q -p no:cacheprovider
echo "== C++ host .nz round trip under ASan + UBSan =="
TMP=$(mktemp -d)
python3 - "$TMP" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd() + "/tests")
sys.path.insert(0, os.getcwd())
import test_nz
from unet_studio_amd import nz
assert nz.save_to_file(test_nz._model(), sys.argv[1] + "/py.nz")
PY
LD_LIBRARY_PATH=$T/lib:$LD_LIBRARY_PATH unet-studio_amd/csrc/build/asan/test_nz_io $TMP/py.nz $TMP/cpp.nz | tail -1
echo "== damaged files through the C++ reader under ASan + UBSan (must fail with a message: no exception, no leak of the gzFile) =="
python3 - "$TMP" <<'PY'
import gzip, struct, sys
d = sys.argv[1]
raw = gzip.open(d + "/py.nz", "rb").read()
gzip.open(d + "/truncated.nz", "wb").write(raw[: len(raw) // 2])
gzip.open(d + "/huge.nz", "wb").write(struct.pack("<5i", 0, 2 ** 31 - 1, 2 ** 31 - 1, 0, 9) + b"channels\0" + b"\0" * 64)
open(d + "/notgzip.nz", "wb").write(b"this is not a gzip stream")
PY
for f in truncated huge notgzip; do
    if LD_LIBRARY_PATH=$T/lib:$LD_LIBRARY_PATH unet-studio_amd/csrc/build/asan/test_nz_io $TMP/$f.nz $TMP/out.nz > $TMP/$f.log 2>&1; then echo "$f.nz was accepted"; exit 1; fi
    grep -q "LOAD FAILED" $TMP/$f.log || { echo "$f.nz: no clean failure"; cat $TMP/$f.log; exit 1; }
done
echo "damaged files refused cleanly"
rm -rf "$TMP"
echo "sanitize_host: clean"
