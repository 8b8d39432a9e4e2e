#!/bin/bash
# Go-19 40x256 B=512: effective CU clock, launch-to-launch gap and matrix-pipe share of kz_board_conv_f16 from in-kernel
# stamps.  Needs kzero_amd/libkzhip_clock.so, built HERE before the gpurun call:
#   (cd kzero_amd/csrc && KZ_OUT=../libkzhip_clock.so KZ_BUILD_DIR=build_clock KZ_EXTRA_FLAGS="-DKZ_BC_STAMPS -DKZ_BC_REALTIME" bash build.sh)
mkdir -p gpurun_out/goclock
KZ_BC_STAMP_LAUNCH=20 KZ_LIB_PATH=$PWD/kzero_amd/libkzhip_clock.so KZ_BC_STAMP_FILE=$PWD/gpurun_out/goclock/stamps.bin python bench.py --repeats 1 --workload go19-40x256 --dtype f16 --steps 5 --warmup 2 --prewarm 0 --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/goclock/bench.json 2> gpurun_out/goclock/bench.err
python tools/board_conv_clock.py gpurun_out/goclock/stamps.bin gpurun_out/goclock/bench.json | tee gpurun_out/goclock/clock.txt
