#!/bin/bash
# Build the HIP libraries for gfx950 (cross-compiles without a GPU).
#   build.sh        libmerlot_hip.so   (product: no experiment switches, no probes)
#                   libmerlot_probe.so (hardware probes used by tests/ and scripts/; never loaded by the product)
#   build.sh exp    additionally libmerlot_hip_exp.so = the product sources with -DMERLOT_EXPERIMENTS (scripts/ only)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
SRCS="gemm attention layernorm elementwise index conv conv_gemm image fp8 jpeg"
HDRS="common.h gemm_ring.h gemm_p8.inc gemm_q8.inc attention_res.inc attention_fb.inc attention_pp.inc ../../include/merlot_hip.h"

newer() {  # newer <target> <deps...>: true when target is missing or older than a dependency
  local t=$1; shift
  [ ! -f "$t" ] && return 0
  for d in "$@"; do [ "$d" -nt "$t" ] && return 0; done
  return 1
}

build_lib() {  # build_lib <objdir> <out> <extra flags>
  local dir=$1 out=$2 extra=$3
  mkdir -p $dir
  local pids=()
  : > $dir/BUILD_MANIFEST                      # one line per object: "compiled <obj>" or "reused <obj>" (what build() really did)
  for f in $SRCS; do
    if newer $dir/$f.o $f.hip $HDRS; then ( $HIPCC $FLAGS $extra -c $f.hip -o $dir/$f.o ) & pids+=($!); echo "compiled $f.o" >> $dir/BUILD_MANIFEST
    else echo "reused $f.o" >> $dir/BUILD_MANIFEST; fi
  done
  for f in capi hostio jpeg_host; do
    if newer $dir/$f.o $f.cpp ../../include/merlot_hip.h; then ( g++ -O2 -fPIC -std=c++17 $extra -c $f.cpp -o $dir/$f.o ) & pids+=($!); echo "compiled $f.o" >> $dir/BUILD_MANIFEST
    else echo "reused $f.o" >> $dir/BUILD_MANIFEST; fi
  done
  for p in "${pids[@]}"; do wait $p || { echo "build.sh: a compile job failed" >&2; exit 1; }; done
  local objs=""
  for f in $SRCS capi hostio jpeg_host; do objs="$objs $dir/$f.o"; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $out $objs
  echo "built $(realpath $out)"
}

build_lib build ../libmerlot_hip.so ""
if newer ../libmerlot_probe.so probe.hip capi.cpp common.h ../../include/merlot_probe.h; then
  $HIPCC $FLAGS -c probe.hip -o build/probe.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libmerlot_probe.so build/probe.o build/capi.o
  echo "built $(realpath ../libmerlot_probe.so)"
fi
if [ "$1" = "exp" ]; then
  build_lib build_exp ../libmerlot_hip_exp.so "-DMERLOT_EXPERIMENTS"
fi
