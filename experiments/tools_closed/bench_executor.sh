#!/bin/bash
# NN evals/s through the drop-in seam (tests/cpp/bench_executor.cpp) on the GPU box: builds the program, writes a
# synthetic chess 20x256 attention model and runs blocking (depth 1) and pipelined (depth 2) executor threads.
#   tools/bench_executor.sh [seconds]
set -e
cd "$(dirname "$0")/../.."
SEC=${1:-4}
mkdir -p tests/cpp/build gpurun_out
g++ -std=c++17 -O2 -pthread tests/cpp/bench_executor.cpp -o tests/cpp/build/bench_executor -Lkzero_amd -lkzhip -Wl,-rpath,$PWD/kzero_amd
python3 -c "
from kzero_amd.synth import random_model
open('/tmp/chess20x256.kzm','wb').write(random_model('chess', 20, 256, 'attention'))"
# gpu_threads, pipeline depth, decode_output on the device
for cfg in "1 1 0" "2 1 0" "3 1 0" "4 1 0" "1 2 0" "2 2 0" "3 2 0" "1 2 1" "2 2 1" "2 1 1" "4 1 1"; do
  set -- $cfg
  tests/cpp/build/bench_executor /tmp/chess20x256.kzm $SEC $1 ${GENERATORS:-6} 256 16 f16 $2 $3
done
