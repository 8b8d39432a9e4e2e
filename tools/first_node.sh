#!/bin/bash
# First contact with a multi-GPU box: everything of this repository that has only ever run on ONE device, in the order a
# failure is cheapest to read, each step's output kept under gpurun_out/first_node/.  Run on the box from the repo root:
#
#   tools/first_node.sh              two-device parity test, bench.py --gpus 1 / 2 / 4 / 8 (as many as the box has), summary
#   REHEARSE=2 tools/first_node.sh   on a ONE-GPU box: the REAL multi-rank path (gloo control plane, real engines, per-rank lines)
#                                    with 2 ranks sharing the GPU (KZ_BENCH_ALLOW_SHARED_GPU=1); the lines say shared_gpu
#   FAKE=1 tools/first_node.sh       CPU dry run of the same script (tests/test_first_node.py): bench.py --fake-step on
#                                    KZ_FAKE_NDEV=8 fake GPUs, optionally over a fake sysfs tree (KZ_FAKE_SYSFS) for the NUMA
#                                    binding; no pytest -m gpu
#
# The topology it exercises is the reference's: one job channel + executor threads per device (rust/kz-selfplay/src/server/
# server.rs:323-331) — as one process per GPU (bench.py's ranks, the driver's launch) and as ONE process over all devices
# (the `seam_one_process` record of the N > 1 bench lines, tests/cpp/test_two_devices.cpp).
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/first_node}
mkdir -p "$OUT"
FAKE=${FAKE:-0}
PY=${PYTHON:-python3}
fail=0

if [ "$FAKE" = "1" ]; then
  NDEV=${KZ_FAKE_NDEV:-8}
  export KZ_FAKE_NDEV=$NDEV
  BENCH_ARGS="--fake-step ${FAKE_STEP_MS:-2} --steps ${STEPS:-5} --warmup 1 --repeats 3"
else
  NDEV=$($PY -c "from kzero_amd import capi; print(capi.device_count())" 2>"$OUT/devices.err") || { echo "first_node: no HIP device / library (see $OUT/devices.err)"; exit 2; }
  $PY -c "from kzero_amd import capi; [print(d, capi.device_pci_bus_id(d)) for d in range(capi.device_count())]" | tee "$OUT/devices.txt"
  BENCH_ARGS="${BENCH_ARGS:---steps 2000 --warmup 50}"
  # 1. two devices in one process: bitwise what device 0 gives alone (f16 one-launch, split16, f32, board paths)
  if [ "$NDEV" -ge 2 ]; then
    $PY -m pytest tests -q -m gpu -k "two_devices" > "$OUT/pytest_two_devices.log" 2>&1 || { echo "first_node: FAILED pytest -k two_devices (see $OUT/pytest_two_devices.log)"; fail=1; }
    tail -3 "$OUT/pytest_two_devices.log"
  else
    echo "first_node: one GPU visible: nothing here has not run before (the N = 1 line follows)"
  fi
fi
if [ "${REHEARSE:-0}" -gt "$NDEV" ] && [ "$FAKE" != "1" ]; then
  export KZ_BENCH_ALLOW_SHARED_GPU=1
  echo "first_node: REHEARSAL — up to $REHEARSE ranks on $NDEV GPU(s)"
  NDEV=$REHEARSE
fi
echo "first_node: $NDEV device(s)"

# 2. one process per GPU, as the driver launches it; N = 1 first (the scaling baseline), then every power of two the box has
for n in 1 2 4 8; do
  [ "$n" -le "$NDEV" ] || continue
  if $PY bench.py --gpus $n $BENCH_ARGS > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"; then
    :
  else
    echo "first_node: FAILED bench.py --gpus $n (exit $?; see $OUT/bench_$n.err)"; tail -5 "$OUT/bench_$n.err"; fail=1
  fi
done

# 3. what to read: per-N aggregate, efficiency against N = 1, every rank's own rate, distinct PCI bus ids, NUMA binding,
#    and the one-process seam over all devices
$PY tools/first_node_summary.py "$OUT" || fail=1
exit $fail
