#!/bin/bash
# Per-kernel register / spill / LDS figures of a build (default kzero_amd/csrc/build), from the AMDGPU metadata note of the
# gfx950 code object embedded in every object file:   tools/kernel_resources.sh [build dir] [name regex]
set -e
DIR=$(realpath ${1:-$(dirname "$0")/../kzero_amd/csrc/build})
FILTER=${2:-.}
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
trap 'rm -rf $TMP' EXIT
cd $TMP
for obj in $DIR/*.o; do
  $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin $obj fat.bin 2>/dev/null || continue
  [ -s fat.bin ] || continue
  T=$($LLVM/clang-offload-bundler --list --type=o --input=fat.bin | grep gfx950 | head -1)
  [ -n "$T" ] || continue
  $LLVM/clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=$T --output=co.elf
  $LLVM/llvm-readelf --notes co.elf | python3 -c '
import re, subprocess, sys
txt, flt = sys.stdin.read(), sys.argv[1]
for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    dem = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*\)$", "", dem.replace("kz::(anonymous namespace)::", "")).replace("void ", "")
    if not re.search(flt, dem): continue
    print("%-62s vgpr %4s (agpr %4s) sgpr %4s spilled v %3s s %3s scratch %5s lds %6s" % (dem, g("vgpr_count"), g("agpr_count"),
          g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
' "$FILTER"
done
