#!/bin/bash
# In-load clock, MFMA-busy share, SQ wait / issue, LDS bank conflicts and L2 hit rate of EVERY dominant kernel of the bench
# line (tools/pmc_workload.sh per workload x arithmetic; separate --pmc passes, one engine) -> gpurun_out/r4/mfma_busy.txt
mkdir -p gpurun_out/r4
: > gpurun_out/r4/mfma_busy.txt
for w in "chess-20x256 f16 30" "chess-20x256 f32split16 20" "chess-20x256 f32 6" "ataxx-8x128 f32 60" "ataxx-8x128 f32split16 60" "ataxx-8x128 f16 60" \
         "go19-40x256 f16 3" "go19-40x256 f32split16 2" "go9-16x128 f16 20" "go9-16x128 f32split16 10"; do
  set -- $w
  bash tools/pmc_workload.sh $1 $2 $3 2>&1 | grep -v "^$" >> gpurun_out/r4/mfma_busy.txt
done
cat gpurun_out/r4/mfma_busy.txt
