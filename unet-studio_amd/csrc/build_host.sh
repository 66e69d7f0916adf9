#!/bin/bash
# Builds the C++ host side (include/unet.hpp + unet_host.cpp over libtorch) and its parity test binary, in-tree.
# libtorch comes from the image's PyTorch-ROCm wheel; the CXX11 ABI flag must equal torch._C._GLIBCXX_USE_CXX11_ABI.
set -e
cd "$(dirname "$0")"
T=$(python3 -c 'import torch, os; print(os.path.dirname(torch.__file__))')
ABI=$(python3 -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')
INC="-I$T/include -I$T/include/torch/csrc/api/include -I/opt/rocm/include -I../../include"
DEF="-D_GLIBCXX_USE_CXX11_ABI=$ABI -D__HIP_PLATFORM_AMD__ -DUSE_ROCM"
LIBS="-L$T/lib -Wl,-rpath,$T/lib -ltorch -ltorch_cpu -ltorch_hip -lc10 -lc10_hip -L.. -Wl,-rpath,\$ORIGIN -lunet_hip -lpthread"
if [ ! -f ../libunet_host.so ] || [ unet_host.cpp -nt ../libunet_host.so ] || [ nz_io.cpp -nt ../libunet_host.so ] || [ ../../include/unet.hpp -nt ../libunet_host.so ] || [ ../../include/unet_hip.h -nt ../libunet_host.so ]; then
    g++ -std=c++17 -O1 -fPIC -shared $DEF $INC unet_host.cpp nz_io.cpp -o ../libunet_host.so $LIBS -lz
fi
if [ ! -f ../test_unet_hpp ] || [ ../../tests/cpp/test_unet_hpp.cpp -nt ../test_unet_hpp ] || [ ../libunet_host.so -nt ../test_unet_hpp ]; then
    g++ -std=c++17 -O1 $DEF $INC ../../tests/cpp/test_unet_hpp.cpp -o ../test_unet_hpp -L.. -lunet_host $LIBS
fi
if [ ! -f ../test_nz_io ] || [ ../../tests/cpp/test_nz_io.cpp -nt ../test_nz_io ] || [ ../libunet_host.so -nt ../test_nz_io ]; then
    g++ -std=c++17 -O1 $DEF $INC ../../tests/cpp/test_nz_io.cpp -o ../test_nz_io -L.. -lunet_host $LIBS
fi
if [ ! -f ../bench_host ] || [ bench_host.cpp -nt ../bench_host ] || [ ../libunet_host.so -nt ../bench_host ]; then
    g++ -std=c++17 -O1 $DEF $INC bench_host.cpp -o ../bench_host -L.. -lunet_host $LIBS
fi
echo "built $(cd .. && pwd)/libunet_host.so and test_unet_hpp"
