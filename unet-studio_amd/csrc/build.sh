#!/bin/bash
# Builds libunet_hip.so for gfx950 in-tree (next to the Python package).  No GPU needed: hipcc cross-compiles.
# Objects are compiled in parallel (one hipcc per source) and only when the source or a header is newer.
set -e
cd "$(dirname "$0")"
SRCS="graph.cpp engine.cpp comm.cpp kernels_direct.hip kernels_elem.hip kernels_mfma_conv.hip kernels_mfma_deep.hip kernels_mfma_conv_z16.hip kernels_mfma_s2.hip kernels_mfma_s2_wgrad.hip kernels_mfma_wgrad.hip kernels_mfma_wgrad_z.hip kernels_mfma_wgrad_zd.hip kernels_augment.hip kernels_mfma_f32.hip"
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result"
mkdir -p build
newest_hdr=$(ls -t *.h *.hpp ../../include/*.h | head -1)
todo=""
for f in $SRCS; do
    o=build/${f%.*}.o
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then todo="$todo $f"; fi
done
[ -f ../asm_loads_check_conv_z.json ] || todo="$todo check:kernels_mfma_conv.hip"
[ -f ../asm_loads_check_wgrad_z.json ] || todo="$todo check:kernels_mfma_wgrad_z.hip"
[ -f ../asm_loads_check_conv_zdma.json ] || todo="$todo check:kernels_mfma_conv_z16.hip"
[ -f ../asm_loads_check_s2dma.json ] || todo="$todo check:kernels_mfma_s2.hip"
[ -f ../asm_loads_check_wgrad_zd.json ] || todo="$todo check:kernels_mfma_wgrad_zd.hip"
todo_src=$(echo $todo | tr ' ' '\n' | grep -v '^check:' | tr '\n' ' ')
if [ -n "$(echo $todo_src | tr -d ' ')" ]; then
    echo $todo_src | tr ' ' '\n' | xargs -P 6 -I{} sh -c 'f={}; hipcc '"$FLAGS"' -c "$f" -o build/${f%.*}.o'
fi
# k_mfma_conv_z and k_mfma_wgrad_z wait for inline-asm loads with hand-counted vmcnt values: check the emitted code whenever
# their file was rebuilt (tools/check_asm_loads.py: no scratch, the expected memory operations, no instruction touching a load's
# registers while it is in flight); the result and the toolchain it was validated with are recorded next to the library
for pair in conv_z:kernels_mfma_conv wgrad_z:kernels_mfma_wgrad_z conv_zdma:kernels_mfma_conv_z16 s2dma:kernels_mfma_s2 wgrad_zd:kernels_mfma_wgrad_zd; do
    which=${pair%%:*}; f=${pair##*:}
    case " $todo " in *" $f.hip "*|*" check:$f.hip "*)
        hipcc $FLAGS --cuda-device-only -S $f.hip -o build/$f.s 2>/dev/null
        python3 tools/check_asm_loads.py $which build/$f.s ../asm_loads_check_$which.json || { rm -f build/$f.o; exit 1; }
        ;;
    esac
done
OBJS=""
for f in $SRCS; do OBJS="$OBJS build/${f%.*}.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libunet_hip.so $OBJS -ldl
echo "built $(cd .. && pwd)/libunet_hip.so"
