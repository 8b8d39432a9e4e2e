#!/bin/bash
# In-kernel phase stamps of kz_board_conv_f16 on Go-19 40x256 B=512 (a -DKZ_BC_STAMPS build as kzero_amd/libkzhip_diag.so):
# launch 20 of the forward pass (second convolution of a block: with the residual) and launch 21 (first: without).
mkdir -p gpurun_out/gostamps
for launch in 20 21; do
KZ_BC_STAMP_LAUNCH=$launch KZ_LIB_PATH=$PWD/kzero_amd/libkzhip_diag.so KZ_BC_STAMP_FILE=$PWD/gpurun_out/gostamps/stamps$launch.bin python bench.py --repeats 1 --workload go19-40x256 --dtype f16 --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-others --no-seam --boundary resident --no-host-io > gpurun_out/gostamps/bench.json 2> gpurun_out/gostamps/bench.err
echo "== launch $launch"
python tools/board_conv_stamps.py gpurun_out/gostamps/stamps$launch.bin
done
