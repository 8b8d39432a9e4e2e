#!/bin/bash
# Round 4: the host work the Rust shim will actually run, timed on the executor thread (tests/cpp/bench_executor.cpp, work =
# real / pre / packed) at f16 and at the parity default, for 1..4 executor threads per device.
#   tools/bench_executor_r4.sh [seconds] > gpurun_out/bench_executor_r4.jsonl
set -e
cd "$(dirname "$0")/.."
SEC=${1:-3}
mkdir -p tests/cpp/build gpurun_out
g++ -std=c++17 -O2 -pthread tests/cpp/bench_executor.cpp -o tests/cpp/build/bench_executor -Lkzero_amd -lkzhip -Wl,-rpath,$PWD/kzero_amd
python3 -c "
from kzero_amd.synth import random_model
open('/tmp/chess20x256.kzm','wb').write(random_model('chess', 20, 256, 'attention'))"
GEN=${GENERATORS:-6}
for dtype in f16 f32split16; do
  # work, gpu_threads, pipeline depth, device_decode
  for cfg in "packed 1 3 0" "real 1 3 0" "real 2 2 0" "real 3 2 0" "real 4 2 0" "real 1 3 1" "real 1 2 1" "real 2 2 1" "pre 1 3 1" "pre 1 2 1" "pre 2 2 1"; do
    set -- $cfg
    tests/cpp/build/bench_executor /tmp/chess20x256.kzm $SEC $2 $GEN 256 8 $dtype $3 $4 0 $1 2>/dev/null
  done
done
