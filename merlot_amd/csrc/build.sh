#!/bin/bash
# Build libmerlot_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [-j]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libmerlot_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p build
pids=()
for f in gemm attention layernorm elementwise index probe conv image; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ ../../include/merlot_hip.h -nt build/$f.o ]; then
    ( $HIPCC $FLAGS -c $f.hip -o build/$f.o ) &
    pids+=($!)
  fi
done
if [ ! -f build/capi.o ] || [ capi.cpp -nt build/capi.o ] || [ ../../include/merlot_hip.h -nt build/capi.o ]; then
  ( g++ -O2 -fPIC -std=c++17 -c capi.cpp -o build/capi.o ) &
  pids+=($!)
fi
if [ ! -f build/hostio.o ] || [ hostio.cpp -nt build/hostio.o ] || [ ../../include/merlot_hip.h -nt build/hostio.o ]; then
  ( g++ -O2 -fPIC -std=c++17 -c hostio.cpp -o build/hostio.o ) &
  pids+=($!)
fi
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT build/gemm.o build/attention.o build/layernorm.o build/elementwise.o build/index.o build/probe.o build/conv.o build/image.o build/capi.o build/hostio.o
echo "built $(realpath $OUT)"
