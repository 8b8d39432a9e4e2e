#!/bin/bash
# In-kernel phase stamps of the exact-f32 tower launch on A1 at a full chip (batch 512, one engine), with the heads
# inside the launch and without.  Needs kzero_amd/libkzhip_stamp.so (a -DKZ_T32_STAMPS build).  Usage (GPU box): bash tools/a1_stamps.sh
mkdir -p gpurun_out/a1s
for nf in 0 1; do
  KZ_NO_FUSED_HEADS=$nf KZ_LIB_PATH=$PWD/kzero_amd/libkzhip_stamp.so KZ_T32_STAMP_FILE=$PWD/gpurun_out/a1s/stamps_nf$nf.bin \
    python bench.py --workload ataxx-8x128 --dtype f32 --engines 1 --batch 512 --steps 60 --warmup 30 --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/a1s/nf$nf.json 2> gpurun_out/a1s/nf$nf.err
  echo "== KZ_NO_FUSED_HEADS=$nf"; python tools/tower32_stamps.py gpurun_out/a1s/stamps_nf$nf.bin 8
done
