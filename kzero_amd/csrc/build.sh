#!/bin/bash
# Builds kzero_amd/libkzhip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libkzhip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mcode-object-version=5 -Wall -Wno-unused-result"
mkdir -p build
pids=()
for src in kz_kernels.hip kz_tower.hip kz_board_conv.hip kz_engine.hip; do
  obj=build/${src%.hip}.o
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ kz_kernels.hpp -nt "$obj" ] || [ kz_model.hpp -nt "$obj" ] || [ ../../include/kz_hip.h -nt "$obj" ]; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for src in kz_model.cpp kz_onnx.cpp; do
  obj=build/${src%.cpp}.o
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ kz_model.hpp -nt "$obj" ]; then
    g++ -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" build/kz_kernels.o build/kz_tower.o build/kz_board_conv.o build/kz_engine.o build/kz_model.o build/kz_onnx.o \
  -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
echo "built $(realpath $OUT)"
