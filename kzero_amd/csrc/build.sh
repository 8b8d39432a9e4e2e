#!/bin/bash
# Builds kzero_amd/libkzhip.so for gfx950 (cross-compiles without a GPU): the PRODUCT library.
#
#   KZ_EXPERIMENTS=1 build.sh   (= experiments/build.sh) builds experiments/libkzhip_exp.so instead: the same sources with
#                               -DKZ_EXPERIMENTS plus the measured-and-rejected kernel organisations of experiments/csrc/
#                               (kz_tower4.hip, kz_board_conv2.hip, the 32x32x16 variants, hipGraph replay) and their
#                               environment switches.  Only tests/test_gpu_experiments.py and experiments/tools_closed/ load
#                               it; nothing of it is compiled into the product, and none of its sources is in this directory.
#   diagnostic builds:          KZ_OUT=../libkzhip_diag.so KZ_BUILD_DIR=build_diag KZ_EXTRA_FLAGS=-DKZ_BC_STAMPS build.sh
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
EXP=${KZ_EXPERIMENTS:-0}
if [ "$EXP" = "1" ]; then
  X=../../experiments
  OUT=${KZ_OUT:-$X/libkzhip_exp.so}
  B=${KZ_BUILD_DIR:-$X/build_exp}
  DEFS="-DKZ_EXPERIMENTS -I. -I$X/csrc"
  HIP_SRCS="kz_kernels.hip kz_tower.hip $X/csrc/kz_tower4.hip kz_tower_f32.hip kz_tower_split.hip kz_tower_f16g.hip kz_tower_pairs_pack.hip kz_conv1x1_split.hip kz_board_conv.hip $X/csrc/kz_board_conv2.hip kz_att_tower.hip kz_att_tower_mfma.hip kz_att_heads.hip kz_dense_network.hip kz_engine.hip"
else
  OUT=${KZ_OUT:-../libkzhip.so}
  B=${KZ_BUILD_DIR:-build}
  DEFS=""
  HIP_SRCS="kz_kernels.hip kz_tower.hip kz_tower_f32.hip kz_tower_split.hip kz_tower_f16g.hip kz_tower_pairs_pack.hip kz_conv1x1_split.hip kz_board_conv.hip kz_att_tower.hip kz_att_tower_mfma.hip kz_att_heads.hip kz_dense_network.hip kz_engine.hip"
fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mcode-object-version=5 -Wall -Wno-unused-result $DEFS ${KZ_EXTRA_FLAGS:-}"
mkdir -p $B
# a change of flags rebuilds everything
if [ "$(cat $B/.flags 2>/dev/null)" != "$FLAGS" ]; then rm -f $B/*.o; echo "$FLAGS" > $B/.flags; fi
pids=()
objs=()
for src in $HIP_SRCS; do
  obj=$B/$(basename ${src%.hip}).o
  objs+=("$obj")
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ kz_kernels.hpp -nt "$obj" ] || [ kz_conv_heads.hpp -nt "$obj" ] || [ kz_decode_dev.hpp -nt "$obj" ] || [ kz_tower_pairs.hpp -nt "$obj" ] || [ kz_tower_pairs_shapes.hpp -nt "$obj" ] || [ ../../experiments/csrc/kz_tower_pairs_exp32.hpp -nt "$obj" ] || [ kz_model.hpp -nt "$obj" ] || [ ../../include/kz_hip.h -nt "$obj" ] || { [ "$src" = kz_engine.hip ] && { [ kz_plan.hpp -nt "$obj" ] || [ kz_device_weights.hpp -nt "$obj" ] || [ kz_engine_util.hpp -nt "$obj" ] || [ kz_engine_state.hpp -nt "$obj" ] || [ kz_engine_forward.hpp -nt "$obj" ]; }; }; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for src in kz_model.cpp kz_onnx.cpp; do
  obj=$B/${src%.cpp}.o
  objs+=("$obj")
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ kz_model.hpp -nt "$obj" ] || [ kz_onnx_match.hpp -nt "$obj" ]; then
    g++ -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
[ $rc = 0 ] || { echo "compile failed" >&2; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}" -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
echo "built $(realpath $OUT)"
