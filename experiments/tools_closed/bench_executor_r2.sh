#!/bin/bash
# round 2: the seam bench with the four-slot zero-copy engine: blocking threads and ONE pipelined thread at depth 2, 3, 4
set -e
cd "$(dirname "$0")/../.."
SEC=${1:-3}
mkdir -p tests/cpp/build gpurun_out
g++ -std=c++17 -O2 -pthread tests/cpp/bench_executor.cpp -o tests/cpp/build/bench_executor -Lkzero_amd -lkzhip -Wl,-rpath,$PWD/kzero_amd
python3 -c "
from kzero_amd.synth import random_model
open('/tmp/chess20x256.kzm','wb').write(random_model('chess', 20, 256, 'attention'))"
for cfg in "1 1 0" "2 1 0" "4 1 0" "1 2 0" "1 3 0" "1 4 0" "2 2 0" "2 4 0" "1 4 1"; do
  set -- $cfg
  tests/cpp/build/bench_executor /tmp/chess20x256.kzm $SEC $1 ${GENERATORS:-8} 256 16 f16 $2 $3
done
