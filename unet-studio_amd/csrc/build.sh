#!/bin/bash
# Builds libunet_hip.so for gfx950 in-tree (next to the Python package).  No GPU needed: hipcc cross-compiles.
# Objects are compiled in parallel (one hipcc per source) and only when the source or a header is newer.
set -e
cd "$(dirname "$0")"
SRCS="graph.cpp engine.cpp kernels_direct.hip kernels_elem.hip kernels_mfma_conv.hip kernels_mfma_wgrad.hip kernels_mfma_wgrad_z.hip kernels_augment.hip kernels_mfma_f32.hip"
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result"
mkdir -p build
newest_hdr=$(ls -t *.h *.hpp ../../include/*.h | head -1)
todo=""
for f in $SRCS; do
    o=build/${f%.*}.o
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then todo="$todo $f"; fi
done
[ -f ../conv_z_check.json ] || todo="$todo conv_z_check"
todo_src=$(echo $todo | tr ' ' '\n' | grep -v '^conv_z_check$' | tr '\n' ' ')
if [ -n "$(echo $todo_src | tr -d ' ')" ]; then
    echo $todo_src | tr ' ' '\n' | xargs -P 6 -I{} sh -c 'f={}; hipcc '"$FLAGS"' -c "$f" -o build/${f%.*}.o'
fi
# k_mfma_conv_z waits for inline-asm loads with hand-counted vmcnt values: check the emitted code whenever its file was rebuilt
# (tools/check_conv_z.py: no scratch, the expected memory operations, no instruction touching a load's registers in flight)
case " $todo " in *" kernels_mfma_conv.hip "*|*" conv_z_check "*)
    hipcc $FLAGS --cuda-device-only -S kernels_mfma_conv.hip -o build/kernels_mfma_conv.s 2>/dev/null
    python3 tools/check_conv_z.py build/kernels_mfma_conv.s ../conv_z_check.json || { rm -f build/kernels_mfma_conv.o; exit 1; }
    ;;
esac
OBJS=""
for f in $SRCS; do OBJS="$OBJS build/${f%.*}.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libunet_hip.so $OBJS
echo "built $(cd .. && pwd)/libunet_hip.so"
