#!/usr/bin/env python3
"""bench.py — self-play NN evals/sec, Chess 20x256 ResNet, executor batch 256 (BASELINE.json metric).

A "step" is one executor batch (256 synthetic packed boards, already resident in HBM) through the whole hot path:
board encode -> ResNet tower -> scalar + attention-policy heads, outputs left in HBM.  evals/s = sum of batch lengths of
completed evaluations / wall time — the reference's own `real` counter (rust/kz-selfplay/src/server/
server_alphazero.rs:113-115, collector.rs:172-191).  Steps are issued round-robin over `--engines` executor engines per
GPU (the reference's gpu_threads_per_device, rust/Readme.md:51), each with its own HIP stream.

Multi-GPU: the path shards by game -> device with no collective (each device has its own job channel, server.rs:325-331):
one process per GPU, every rank runs K steps on its own boards ("weak" scaling), value = all ranks' evals / max time.
torch.distributed is only the control plane (barrier + max of the elapsed time) and runs on gloo: there is no tensor
to exchange, so RCCL/xGMI stay idle by design.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps; default: about 5 s of the workload (SURVEY.md §8(d)): chess 10000, ataxx 10000, go 400")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=None,
                    help="executor batch (gpu_batch_size); default: BASELINE.json's for the workload (256; Go 512)")
    ap.add_argument("--engines", type=int, default=None,
                    help="executor engines (streams) per GPU = gpu_threads_per_device; default: what fills the chip "
                         "for the workload (chess 2: half-chip launches; ataxx 3: covers its separate head kernels; "
                         "go 1: a launch per layer already fills the chip)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32", "f32split16"],
                    help="f32split16: f32 tensors and <=1e-4 parity, tower products as three f16 MFMAs on (hi, lo) pairs")
    ap.add_argument("--workload", default="chess-20x256", choices=["chess-20x256", "ataxx-8x128", "go19-40x256"])
    ap.add_argument("--prewarm", type=float, default=0.25,
                    help="seconds of untimed conditioning steps before the --warmup steps (0 to disable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-io", action="store_true",
                    help="diagnostic: feed host buffers through kz_engine_submit_packed/kz_engine_wait (PCIe-inclusive, "
                         "two slots per engine); never the configuration `value` is quoted on")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time for the cpu_baseline sample")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 400 if args.workload == "go19-40x256" else 10000
    if args.batch is None:
        args.batch = 512 if args.workload == "go19-40x256" else 256
    if args.engines is None:
        args.engines = {"chess-20x256": 2, "ataxx-8x128": 3, "go19-40x256": 1}[args.workload]
    return args


WORKLOADS = {
    "chess-20x256": dict(game="chess", depth=20, channels=256, head="attention"),
    "ataxx-8x128": dict(game="ataxx-7", depth=8, channels=128, head="ataxx_conv"),
    "go19-40x256": dict(game="go-19", depth=40, channels=256, head="conv"),
}


def usable_cores():
    """Host threads this process can actually run on: the affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the machine, not the container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(blob, bits, scalars_in, target_seconds):
    """The CPU restatement (oracle, kind 'port') timed on this box's host cores, on a bounded sample of the same
    boards.  Reported beside the GPU number; never the thing measured as `value`."""
    import numpy as np
    from tests import oracle_lib as O
    net = O.OracleNet(blob)
    cores = usable_cores()
    dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    t0 = time.perf_counter()
    net.forward(dense[:1], threads=1)
    one = time.perf_counter() - t0
    n = int(max(cores, min(len(dense), cores * max(1.0, target_seconds / max(one, 1e-3)) // 1)))
    n = max(cores, (n // cores) * cores)
    n = min(n, len(dense))
    t0 = time.perf_counter()
    net.forward(dense[:n], threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"{n} boards of the same synthetic batch, oracle/kz_oracle.c f32 NCHW direct conv, "
                      f"OpenMP over boards on {cores} threads (of {os.cpu_count()} the machine reports), {dt:.1f} s; "
                      f"single-thread {1.0 / one:.3f} evals/s"}


def main():
    args = parse_args()
    from kzero_amd import benchlib
    rank, local_rank, world, distributed = benchlib.rank_info()
    torch = None
    if distributed:
        import torch  # noqa: F811  (control plane only; imported before the HIP library on purpose)
    dist = benchlib.init_control_plane()

    import numpy as np
    from kzero_amd import capi, synth

    ndev = capi.device_count()
    device = local_rank % max(ndev, 1)
    dtype = {"f16": capi.KZ_DTYPE_F16, "f32": capi.KZ_DTYPE_F32, "f32split16": capi.KZ_DTYPE_F32_SPLIT16}[args.dtype]
    wl = WORKLOADS[args.workload]

    blob = synth.random_model(wl["game"], wl["depth"], wl["channels"], wl["head"], seed=0)
    model = capi.Model(blob=blob)
    info = model.info
    B = args.batch
    bits, scalars_in = synth.random_boards(wl["game"], B, seed=benchlib.board_seed(rank))

    engines = [capi.Engine(model, device, B, dtype) for _ in range(args.engines)]
    d_bits = capi.DeviceBuffer.from_host(device, bits)
    d_sin = capi.DeviceBuffer.from_host(device, scalars_in)
    outs = [(capi.DeviceBuffer(device, B * 5 * 4), capi.DeviceBuffer(device, B * info.policy_len * 4))
            for _ in engines]
    stride = bits.shape[1]

    inflight = {}

    def step_host(i):
        # round-robin over (engine, slot); wait for the slot's previous batch before reusing it
        e, slot = i % len(engines), (i // len(engines)) % capi.KZ_ENGINE_SLOTS
        if (e, slot) in inflight:
            engines[e].wait_view(slot, inflight.pop((e, slot)))  # zero-copy: the results stay in pinned staging
        inflight[(e, slot)] = engines[e].submit_packed(slot, bits, scalars_in)

    def step(i):
        if args.host_io:
            return step_host(i)
        e = i % len(engines)
        engines[e].enqueue_packed_device(d_bits, stride, d_sin, B, outs[e][0], outs[e][1])

    def sync_all():
        for (e, slot), n in list(inflight.items()):
            engines[e].wait_view(slot, n)
        inflight.clear()
        for e in engines:
            e.synchronize()
        capi.check(capi.load().kz_device_synchronize(device))
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    def start_profiling():
        for e in engines:
            e.set_profiling(True)

    # Conditioning, not measurement: a quarter of a second of the same steps takes the device out of its idle power
    # state and through first-launch set-up before the W warm-up steps and the K timed steps of the contract, so that a
    # short --steps/--warmup run measures the same steady state as a long one.
    t_end = time.perf_counter() + args.prewarm
    i = 0
    while time.perf_counter() < t_end:
        step(i)
        i += 1
    sync_all()
    elapsed = benchlib.run_timed(step, sync_all, args.steps, args.warmup, dist, on_timed_start=start_profiling)

    # dominant kernel, timed with HIP events on the engines' own streams over the timed region
    tower_path = engines[0].tower_path
    kname = {"tower_resident_f16+heads": "kz_tower_resident_f16", "tower_resident_f16": "kz_tower_resident_f16",
             "tower_resident_f32": "kz_tower_resident_f32", "tower_resident_split16": "kz_tower_resident_split", "tower_resident_f16g": "kz_tower_resident_f16g",
             "board_conv_f16": "kz_board_conv_f16", "conv_igemm_f16": "kz_conv_igemm_f16",
             "conv_igemm_f32": "kz_conv_igemm_f32"}[tower_path]
    k_ms, k_n = 0.0, 0
    for e in engines:
        ms, n = e.kernel_time(kname)
        k_ms += ms
        k_n += n
        e.set_profiling(False)

    # sanity: outputs are finite numbers
    if not args.host_io:
        s_host = outs[0][0].to_host(np.float32, (B, 5))
        assert np.isfinite(s_host).all(), "non-finite network output"

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = benchlib.whole_job_value(args.steps, B, world, elapsed)
    hw = info.board_h * info.board_w
    C = info.tower_channels
    tower_flops = 2.0 * hw * 9 * C * (info.input_channels + 2 * info.tower_depth * C)  # per board, direct conv
    if tower_path == "tower_resident_f16+heads":
        flops_per_launch = info.flops_per_eval * B  # one launch = tower + heads for one batch
    elif tower_path in ("tower_resident_f16", "tower_resident_f32", "tower_resident_split16", "tower_resident_f16g"):
        flops_per_launch = tower_flops * B  # one launch = the whole tower for one batch
    elif tower_path == "board_conv_f16":
        # one launch per 3x3 tower convolution except the stem (which has too few input channels for this kernel)
        flops_per_launch = 2.0 * hw * 9 * C * C * B
    else:
        # per-layer launches (tower + the 1x1 head convolutions that share the kernel): average over the step
        head_flops = info.flops_per_eval - tower_flops
        launches_per_step = k_n / max(args.steps, 1)
        flops_per_launch = (tower_flops + head_flops) * B / max(launches_per_step, 1)
    # f32split16: algorithmic FLOP (one multiply-add per product) against the f16 matrix cores that execute three
    peak = 157.3 if args.dtype == "f32" else 2500.0
    avg_ms = k_ms / max(k_n, 1)
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if k_n else 0.0
    # `achieved`/`frac` follow the contract literally: algorithmic FLOP of ONE launch / its average duration.  A resident
    # launch covers ceil(batch / boards_per_workgroup) of the 256 CUs and `engines` launches run side by side, so the
    # chip-level figure is `chip_frac` (all engines' FLOP / wall time), not `frac`.
    nb = int(os.environ.get("KZ_TOWER_NB", "2")) if tower_path.startswith("tower_resident_f16") else None
    wgs = -(-B // (1 if nb == 1 else 2)) if nb else None
    # HBM-side bytes per launch of this kernel at this shape, from the separate rocprofv3 --pmc passes committed under
    # profiles/r1_pmc_final/ (FETCH_SIZE 188,179 KiB x2 per the gfx950 wide-read correction + WRITE_SIZE 1,888 KiB):
    # each of the 8 non-coherent XCD L2s pulls the 49 MB weight stream once.  None for shapes that were not profiled.
    traffic = None
    if tower_path == "tower_resident_f16+heads" and B == 256 and args.workload == "chess-20x256":
        traffic = (2 * 188179.35 + 1888.0) * 1024
    roofline = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 5),
                "launches": k_n, "flop_per_launch": flops_per_launch,
                "workgroups_per_launch": wgs, "concurrent_launches": args.engines,
                "chip_frac": round(value / world * info.flops_per_eval / 1e12 / peak, 4)}

    out = {
        "metric": "self-play NN evals/sec (node), Chess 20x256 ResNet b=256, 1/2/4/8 GPU",
        "value": round(value, 1), "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.workload} {wl['head']} head, executor batch {B}, " +
                               ("host buffers over PCIe (diagnostic)" if args.host_io else "packed boards resident in HBM"),
                   "engines_per_gpu": args.engines, "conditioning_s": args.prewarm, "tower_path": tower_path, "parallelism": f"dp{world} (no collective)",
                   "flop_per_eval": info.flops_per_eval},
        "roofline": roofline,
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(blob, bits, scalars_in, args.cpu_seconds)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
