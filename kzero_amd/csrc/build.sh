#!/bin/bash
# Builds kzero_amd/libkzhip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${KZ_OUT:-../libkzhip.so}   # diagnostic builds: KZ_OUT=../libkzhip_diag.so KZ_BUILD_DIR=build_diag KZ_EXTRA_FLAGS=-DKZ_BC_STAMPS
B=${KZ_BUILD_DIR:-build}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mcode-object-version=5 -Wall -Wno-unused-result ${KZ_EXTRA_FLAGS:-}"
mkdir -p $B
pids=()
for src in kz_kernels.hip kz_tower.hip kz_tower4.hip kz_tower_f32.hip kz_tower_split.hip kz_board_conv.hip kz_board_conv2.hip kz_engine.hip; do
  obj=$B/${src%.hip}.o
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ kz_kernels.hpp -nt "$obj" ] || [ kz_model.hpp -nt "$obj" ] || [ ../../include/kz_hip.h -nt "$obj" ]; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for src in kz_model.cpp kz_onnx.cpp; do
  obj=$B/${src%.cpp}.o
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ kz_model.hpp -nt "$obj" ]; then
    g++ -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" $B/kz_kernels.o $B/kz_tower.o $B/kz_tower4.o $B/kz_tower_f32.o $B/kz_tower_split.o $B/kz_board_conv.o $B/kz_board_conv2.o $B/kz_engine.o $B/kz_model.o $B/kz_onnx.o \
  -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
echo "built $(realpath $OUT)"
