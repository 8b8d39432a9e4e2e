#!/usr/bin/env python3
"""bench.py — self-play NN evals/sec, Chess 20x256 ResNet, executor batch 256 (BASELINE.json metric).

A "step" is one executor batch (256 synthetic packed boards) through the whole hot path: board encode -> ResNet tower ->
scalar + attention-policy heads.  evals/s = sum of batch lengths of completed evaluations / wall time — the reference's
own `real` counter (rust/kz-selfplay/src/server/server_alphazero.rs:113-115, collector.rs:172-191).  Steps are issued
round-robin over `--engines` executor engines per GPU (the reference's gpu_threads_per_device, rust/Readme.md:51), each
with its own HIP stream(s).

ONE JSON line of at most 4 KB on rank 0 (kzero_amd/benchline.py: fixed key set; round 5's 27 KB line was not parsed by the
driver) and the full record of everything measured in bench_full.json beside it (tools/show_bench.py prints it):
  * `value`: the K timed steps THROUGH THE BOUNDARY A CALLER HAS — host boards + move lists in, decoded values + legal-move
    probabilities out (kz_engine_submit_packed_decoded -> kz_engine_wait_decoded: what kzero_amd/rust/hip.rs runs by
    default, replacing cudnn.rs:55-87 + common.rs:16-100), H2D of 284 B and D2H of 160 B per chess evaluation inside the
    timed region.  The K-step region is timed `--repeats` times (default 7), each bracketed by barrier + device sync on both
    sides; `value` is the MEDIAN region, `value_min` / `value_max` the slowest and fastest, `ms_per_step` x `steps` = the
    median region;
  * `value_device_resident`: the same K steps with the packed boards already in HBM and the raw outputs left in HBM — what
    `value` was in rounds 1-5 (1.00-1.015 x `value`; no caller of the Network trait can obtain it); `--boundary resident` makes
    it `value` for kernel A/B runs.  `value_host_boundary_raw`: host boards in, the raw 7.5 KB of policy rows out;
  * `roofline` for the dominant kernel, following `value`: HIP events on the engines' own streams over the timed region;
  * `others` (N=1 only): the other single-GPU BASELINE configs as `[workload, dtype, evals/s, frac]` tuples from ~1 s
    sub-records through the same boundary — A1 Ataxx 8x128 f32 B=256 (and at parity), the G8 network Go-19 40x256 B=512 in
    f16 and at parity, and the chess network through the <=1e-4-parity path (f32split16) = `value_parity_default`;
  * `cpu_baseline` (N=1 only): the oracle on this box's host cores, bounded sample.

Multi-GPU: the path shards by game -> device with no collective (each device has its own job channel,
server.rs:323-331): one process per GPU.  `--gpus N` without a RANK in the environment starts the N rank processes
itself (fresh children, before this process touches the GPU); under `torch.distributed.run` the ranks already exist.
Every rank runs K steps on its own boards ("weak" scaling), value = all ranks' evals / max time.  torch.distributed is
only the control plane (barrier, max of the elapsed time, gather of the PCI bus ids) and runs on gloo: there is no
tensor to exchange, so RCCL/xGMI stay idle by design.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
FULL_AFFINITY = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set()

WORKLOADS = {
    "chess-20x256": dict(game="chess", depth=20, channels=256, head="attention", batch=256, steps=2000,
                         engines={"f16": 2, "f32": 2, "f32split16": 2},
                         label="Chess 20x256 ResNet b=256"),
    "ataxx-8x128": dict(game="ataxx-7", depth=8, channels=128, head="ataxx_conv", batch=256, steps=10000,
                        engines={"f16": 3, "f32": 3, "f32split16": 3},  # (f32: 517k with three, 507k with two)
                        label="Ataxx 7x7 8x128 ResNet b=256"),
    "go19-40x256": dict(game="go-19", depth=40, channels=256, head="conv", batch=512, steps=400,
                        engines={"f16": 1, "f32": 1, "f32split16": 1},
                        label="Go 19x19 40x256 ResNet b=512"),
    # the reference's own shipping configuration (python/main/loop_main_alpha.py:16-30,68-76): Go 9x9, 16 blocks x 128
    # channels, ConvPolicyHead(extra_moves=1), gpu_batch_size 2048 (one launch of 2048 workgroups fills the chip eight times)
    "go9-16x128": dict(game="go-9", depth=16, channels=128, head="conv", batch=2048, steps=1000,
                       engines={"f16": 2, "f32": 2, "f32split16": 2},
                       label="Go 9x9 16x128 ResNet b=2048 (loop_main_alpha.py)"),
    # diagnostic workloads (tools/pmc_workload.sh): chess towers off the flagship width; not part of the default line
    "chess-20x192": dict(game="chess", depth=20, channels=192, head="attention", batch=256, steps=2000,
                         engines={"f16": 3, "f32": 2, "f32split16": 2}, label="Chess 20x192 ResNet b=256"),
    "chess-20x384": dict(game="chess", depth=20, channels=384, head="attention", batch=256, steps=1000,
                         engines={"f16": 2, "f32": 2, "f32split16": 2}, label="Chess 20x384 ResNet b=256"),
    "chess-20x64": dict(game="chess", depth=20, channels=64, head="attention", batch=256, steps=4000,
                        engines={"f16": 3, "f32": 2, "f32split16": 3}, label="Chess 20x64 ResNet b=256"),
    "chess-20x128": dict(game="chess", depth=20, channels=128, head="attention", batch=1024, steps=1000,
                         engines={"f16": 2, "f32": 2, "f32split16": 2}, label="Chess 20x128 ResNet b=1024"),
    "chess-20x320": dict(game="chess", depth=20, channels=320, head="attention", batch=256, steps=1000,
                         engines={"f16": 2, "f32": 2, "f32split16": 2}, label="Chess 20x320 ResNet b=256"),
    "chess-20x512": dict(game="chess", depth=20, channels=512, head="attention", batch=256, steps=500,
                         engines={"f16": 2, "f32": 2, "f32split16": 2}, label="Chess 20x512 ResNet b=256"),
    "go9-20x256": dict(game="go-9", depth=20, channels=256, head="conv", batch=256, steps=1000,
                       engines={"f16": 2, "f32": 2, "f32split16": 2}, label="Go 9x9 20x256 ResNet b=256"),
    # the network python/main/supervised_main_alpha.py:69-77 trains on chess: AttentionTower(8, 21, depth 16, d_model 256,
    # 8 heads, d_k = d_v = 16, d_ff 256) under the ScalarHead and the AttentionPolicyHead (0.594 GFLOP per eval)
    "chess-att16x256": dict(game="chess", depth=16, channels=256, head="attention", batch=256, steps=1000,
                            engines={"f16": 2, "f32": 2}, model_kw=dict(attention=(8, 16, 16, 256)),
                            label="Chess AttentionTower 16x256 (8 heads) b=256 (supervised_main_alpha.py)"),
    "go13-20x128": dict(game="go-13", depth=20, channels=128, head="conv", batch=256, steps=1000,
                        engines={"f16": 2, "f32": 2, "f32split16": 2}, label="Go 13x13 20x128 ResNet b=256"),
}
# the other single-GPU BASELINE configs, reported as sub-records of the default line
# the headline network with other weight statistics (sub-records `weights`): the f16 matrix cores are power-limited and the
# power they draw depends on the operands' bits (the same launch: 0.63 of peak on uniform random weights, 0.79 on all-zero
# ones), so the headline's uniform PyTorch-default init comes with a measured band
WEIGHT_VARIANTS = [
    ("uniform (PyTorch default init: the headline's)", dict()),
    ("kaiming_normal (N(0, sqrt(2 / fan_in)) convolution weights)", dict(init="kaiming_normal")),
    ("trained-like scale (block_gain 3: the residual stream grows to ~170, logits to ~1.8e3)", dict(block_gain=3.0)),
]
# BASELINE.json configs[1] (A1, f32), configs[2] at the <= 1e-4 default arithmetic, configs[4] (G8) in both arithmetics — and
# nothing else: AttentionTower / DenseNetwork (SURVEY.md:145, out of scope) and go9-16x128 stay selectable with --workload
OTHERS = [("ataxx-8x128", "f32"), ("ataxx-8x128", "f32split16"), ("go19-40x256", "f16"), ("chess-20x256", "f32split16"),
          ("go19-40x256", "f32split16")]

KERNEL_OF_PATH = {
    "tower_resident_f16+heads": "kz_tower_resident_f16", "tower_resident_f16": "kz_tower_resident_f16",
    "tower_resident_f32": "kz_tower_resident_f32", "tower_resident_f32+heads": "kz_tower_resident_f32",
    "tower_resident_split16": "kz_tower_resident_split", "tower_resident_split16+heads": "kz_tower_resident_split",
    "tower_resident_f16g": "kz_tower_resident_f16g", "tower_resident_f16g+heads": "kz_tower_resident_f16g",
    "board_conv_f16": "kz_board_conv_f16",
    "attention_tower_f16": "kz_att_tower_f16", "attention_tower_f32": "kz_att_tower_f32", "attention_tower_f32_valu": "kz_att_tower_f32_valu",
    "board_conv_split16": "kz_board_conv_split16",
    "conv_igemm_f16": "kz_conv_igemm_f16", "conv_igemm_f32": "kz_conv_igemm_f32",
}
# source file of each dominant kernel: the committed PMC traffic figure is only reported while this file is unchanged
KERNEL_SOURCE = {
    "kz_tower_resident_f16": "kz_tower.hip", "kz_tower_resident_f32": "kz_tower_f32.hip",
    "kz_tower_resident_split": "kz_tower_split.hip", "kz_tower_resident_f16g": "kz_tower_f16g.hip",
    "kz_board_conv_f16": "kz_board_conv.hip", "kz_board_conv_split16": "kz_board_conv.hip", "kz_conv_igemm_f16": "kz_kernels.hip", "kz_conv_igemm_f32": "kz_kernels.hip",
    "kz_att_tower_f16": "kz_att_tower_mfma.hip", "kz_att_tower_f32": "kz_att_tower_mfma.hip", "kz_att_tower_f32_valu": "kz_att_tower.hip",
}
# device code a kernel source pulls in (hashed with it: a traffic record goes stale when either changes).  The (hi, lo) tower
# and its plain-f16 sibling are one template (kz_tower_pairs.hpp) instantiated by a translation unit each: an edit to one
# family's instances leaves the other's record alone, an edit to the shared body stales both — as it should
_PAIRS = ["kz_tower_pairs.hpp", "kz_tower_pairs_shapes.hpp", "kz_conv_heads.hpp", "kz_decode_dev.hpp"]
KERNEL_DEVICE_HEADERS = {"kz_tower.hip": ["kz_decode_dev.hpp"], "kz_tower_f32.hip": ["kz_conv_heads.hpp", "kz_decode_dev.hpp"],
                         "kz_tower_split.hip": _PAIRS, "kz_tower_f16g.hip": _PAIRS}
TRAFFIC_FILE = os.path.join(REPO, "profiles", "hbm_traffic.json")
FULL_RECORD = os.environ.get("KZ_BENCH_FULL_RECORD", os.path.join(REPO, "bench_full.json"))


def kernel_source_hash(kernel: str) -> str:
    h = hashlib.sha256()
    src = KERNEL_SOURCE[kernel]
    for name in [src] + KERNEL_DEVICE_HEADERS.get(src, []):
        h.update(open(os.path.join(REPO, "kzero_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps per region; default: about 1 s of the workload (chess 2000, ataxx 10000, go 400)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=7,
                    help="timed regions of --steps steps each (every one bracketed by barrier + sync); value = the median region")
    ap.add_argument("--batch", type=int, default=None,
                    help="executor batch (gpu_batch_size); default: BASELINE.json's for the workload (256; Go 512)")
    ap.add_argument("--engines", type=int, default=None,
                    help="executor engines (streams) per GPU = gpu_threads_per_device; default: what fills the chip "
                         "for the workload (chess 2: half-chip launches; ataxx 3; go 1: a launch per layer fills the chip)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32", "f32split16"],
                    help="f32split16: f32 tensors and <=1e-4 parity, tower products as three f16 MFMAs on (hi, lo) pairs")
    ap.add_argument("--workload", default="chess-20x256", choices=sorted(WORKLOADS))
    ap.add_argument("--prewarm", type=float, default=0.25,
                    help="seconds of untimed conditioning steps before the --warmup steps (0 to disable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the sub-records of the other BASELINE configs")
    ap.add_argument("--weight-variants", action="store_true",
                    help="also run the headline network with other weight statistics (full record only: others[].weights)")
    ap.add_argument("--no-seam", action="store_true", help="skip the whole-seam sub-record (generators -> executor loop); --no-others skips it too")
    ap.add_argument("--seam-seconds", type=float, default=2.0)
    ap.add_argument("--boundary", default="decoded", choices=["decoded", "resident"],
                    help="what `value` times.  decoded (default): host boards + move lists in, decoded values + legal-move "
                         "probabilities out — what a caller of the Network trait gets.  resident: packed boards already in HBM, raw "
                         "outputs left there (kernel A/B runs; what `value` was up to round 5)")
    ap.add_argument("--no-host-io", action="store_true", help="skip the secondary timed regions (device-resident and raw host boundary)")
    ap.add_argument("--other-seconds", type=float, default=1.0, help="timed seconds per sub-record")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time for the cpu_baseline sample")
    ap.add_argument("--fake-step", type=float, default=None, metavar="MS",
                    help="TEST ONLY (tests/test_bench_launcher.py): no GPU, no library — a step is a sleep of MS "
                         "milliseconds; exercises the rank launcher and the aggregation; the line says data=fake")
    args = ap.parse_args(argv)
    wl = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = wl["steps"]
    if args.batch is None:
        args.batch = wl["batch"]
    if args.engines is None:
        args.engines = wl["engines"][args.dtype]
    args.is_default_line = args.workload == "chess-20x256" and args.dtype == "f16" and args.batch == 256
    return args


# ------------------------------------------------------------------------------------------------------------------
# rank launcher: `python bench.py --gpus N` on its own starts N fresh rank processes (one per GPU)
# ------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    """Starts N children with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, BEFORE this process has made any HIP call (a
    process that has initialised the GPU must not fork+exec on this pool).  Rank 0 prints the one JSON line; the exit
    code is the worst child's."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = None
    while procs:
        for p in list(procs):
            r = p.poll()
            if r is None:
                continue
            procs.remove(p)
            if r != 0:
                rc = rc or r
                deadline = deadline or time.time() + 20.0  # a failed rank: give the others a moment, then stop them
        if deadline and time.time() > deadline:
            for p in procs:
                p.kill()  # exact PIDs we started
            for p in procs:
                p.wait()
            break
        time.sleep(0.05)
    return rc


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
    from tests import oracle_lib as O
    if hasattr(os, "sched_setaffinity") and FULL_AFFINITY:
        try:
            os.sched_setaffinity(0, FULL_AFFINITY)  # (this process bound itself to its GPU's NUMA node for the timed regions)
        except OSError:
            pass
    net = O.OracleNet(blob)
    cores = usable_cores()
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
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
    a0 = None
    try:  # BASELINE configs[0]: Ataxx 7x7, 4 blocks x 64 channels, the CPU executor on ONE thread (the generator's own)
        from kzero_amd import synth
        a0_blob = synth.random_model("ataxx-7", 4, 64, "ataxx_conv", seed=0)
        a0_net = O.OracleNet(a0_blob)
        a0_bits, a0_sin = synth.random_boards("ataxx-7", 256, seed=1000)
        a0_dense = O.encode_input_full(a0_bits, a0_sin, a0_net.n_scalar, a0_net.n_bool, a0_net.h, a0_net.w)
        a0_net.forward(a0_dense[:4], threads=1)
        t0 = time.perf_counter()
        done = 0
        while time.perf_counter() - t0 < 2.0:
            a0_net.forward(a0_dense[done % 256:done % 256 + 16], threads=1)
            done += 16
        a0 = {"config": "BASELINE configs[0]: Ataxx 7x7 4x64, CPU executor, 1 thread", "value": round(done / (time.perf_counter() - t0), 2),
              "unit": "evals/s", "cores": 1, "kind": "port", "sample": f"{done} boards in batches of 16, oracle/kz_oracle.c, one thread"}
    except Exception as ex:  # noqa: BLE001
        a0 = {"error": f"{type(ex).__name__}: {ex}"[:200]}
    return {"value": round(n / dt, 3), "unit": "evals/s", "cores": cores, "kind": "port", "a0": a0,
            "sample": f"{n} boards of the same batch, oracle/kz_oracle.c (f32 direct conv), OpenMP on {cores} threads, {dt:.1f} s; "
                      f"1 thread: {1.0 / one:.2f} evals/s",
            "sample_detail": f"f32 NCHW direct conv (zero-padded planes, nine taps and two output channels per pass); {cores} threads = the "
                             f"container's CPU quota (cgroup cpu.max; the affinity mask allows {affinity}, the machine reports "
                             f"{os.cpu_count()}: a whole socket is not this process's to use)"}


def committed_traffic(kernel: str, workload: str, batch: int):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (profiles/hbm_traffic.json, written by
    tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950
    correction) — reported only while the kernel's source file still has the hash it had when the passes ran."""
    try:
        table = json.load(open(TRAFFIC_FILE))
        sha = kernel_source_hash(kernel)
    except (OSError, ValueError, KeyError):
        return None, None
    for rec in table.get("records", []):
        if rec["kernel"] == kernel and rec["workload"] == workload and rec["batch"] == batch:
            if rec["source_sha256_16"] != sha:
                return None, f"stale: {KERNEL_SOURCE[kernel]} changed since {rec['profile']}"
            return rec["traffic_bytes_per_launch"], rec["profile"]
    return None, None


class Workload:
    """One (network, dtype, batch) on one device: engines, resident inputs, the two kinds of step."""

    def __init__(self, capi, synth, name, dtype_name, batch, n_engines, device, seed, model_kw=None):
        import numpy as np
        from kzero_amd import benchlib
        self.np, self.capi, self.benchlib = np, capi, benchlib
        self.name, self.dtype_name, self.B, self.device = name, dtype_name, batch, device
        wl = WORKLOADS[name]
        self.wl = wl
        dtype = {"f16": capi.KZ_DTYPE_F16, "f32": capi.KZ_DTYPE_F32, "f32split16": capi.KZ_DTYPE_F32_SPLIT16}[dtype_name]
        self.blob = synth.random_model(wl["game"], wl["depth"], wl["channels"], wl["head"], seed=0,
                                       **{**wl.get("model_kw", {}), **(model_kw or {})})
        self.model = capi.Model(blob=self.blob)
        self.info = self.model.info
        self.bits, self.scalars_in = synth.random_boards(wl["game"], batch, seed=seed)
        self.engines = [capi.Engine(self.model, device, batch, dtype) for _ in range(n_engines)]
        self.d_bits = capi.DeviceBuffer.from_host(device, self.bits)
        self.d_sin = capi.DeviceBuffer.from_host(device, self.scalars_in)
        self.outs = [(capi.DeviceBuffer(device, batch * 5 * 4), capi.DeviceBuffer(device, batch * self.info.policy_len * 4))
                     for _ in self.engines]
        self.stride = self.bits.shape[1]
        self.inflight = {}
        self.inflight_dec = set()
        self.moves = None
        self.tower_path = self.engines[0].tower_path
        self.kernel = KERNEL_OF_PATH[self.tower_path]
        # host-pointer loop: on the fused path an engine's two slots own a stream each, so ONE executor thread keeps
        # both half-chip launches in flight; more launches than the chip holds at once (engines x slots x 128 workgroups
        # > 256 CUs) only queue, and on a cold start that queueing has been seen to stall a stream for ~8 ms
        self.host_engines = self.engines[:1] if self.tower_path.endswith("+heads") else self.engines

    def step_resident(self, i):
        e = i % len(self.engines)
        self.engines[e].enqueue_packed_device(self.d_bits, self.stride, self.d_sin, self.B, self.outs[e][0], self.outs[e][1])

    def step_host(self, i):
        # round-robin over (engine, slot); wait for the slot's previous batch before reusing it.  Results are read in
        # place through kz_engine_wait_view (the lifetime of the reference executor's `&[DTensor]`, cudnn.rs:73-82)
        n, S = len(self.host_engines), self.capi.KZ_ENGINE_SLOTS
        e, slot = i % n, (i // n) % S
        if (e, slot) in self.inflight:
            self.host_engines[e].wait_view(slot, self.inflight.pop((e, slot)))
        self.inflight[(e, slot)] = self.host_engines[e].submit_packed(slot, self.bits, self.scalars_in)

    def step_host_decoded(self, i):
        # the entry points the Rust shim uses by default: packed boards + the CSR lists of move_to_index of the available moves
        # in, decoded values + one probability per move out (kz_engine_submit_packed_decoded / kz_engine_wait_decoded)
        if self.moves is None:
            rng = self.np.random.default_rng(7)
            counts = rng.integers(20, 51, size=self.B)  # ~35 legal moves per position
            self.moves = (self.np.concatenate([[0], self.np.cumsum(counts)]).astype(self.np.int64),
                          self.np.ascontiguousarray(self.np.concatenate([rng.permutation(self.info.policy_len)[:c] for c in counts]).astype(self.np.int32)))
        n, S = len(self.host_engines), self.capi.KZ_ENGINE_SLOTS
        e, slot = i % n, (i // n) % S
        if (e, slot) in self.inflight_dec:
            self.host_engines[e].wait_decoded_view(slot)
            self.inflight_dec.discard((e, slot))
        self.host_engines[e].submit_packed_decoded_csr(slot, self.bits, self.scalars_in, self.moves[0], self.moves[1])
        self.inflight_dec.add((e, slot))

    def sync(self):
        for (e, slot), n in list(self.inflight.items()):
            self.host_engines[e].wait_view(slot, n)
        self.inflight.clear()
        for (e, slot) in list(self.inflight_dec):
            self.host_engines[e].wait_decoded_view(slot)
        self.inflight_dec.clear()
        for e in self.engines:
            e.synchronize()
        self.capi.check(self.capi.load().kz_device_synchronize(self.device))

    def profiling(self, on):
        if on and os.environ.get("KZ_BENCH_NO_KERNEL_TIMING") == "1":
            return  # (A/B runs of KZ_HIP_GRAPH=1: a forward pass bracketed by events is not replayed from a graph)
        for e in self.engines:
            e.set_profiling(on)

    @property
    def per_layer(self):
        """One launch per layer (85 per Go batch): HIP events around every launch cost such a path 1.7 % of its rate
        (Go-19: 36.2k against 36.8k evals/s), so its K timed steps run without them and the dominant kernel's average
        launch duration comes from an instrumented pass of the same steps right behind the timed region."""
        return not self.tower_path.startswith(("tower_resident", "attention_tower"))

    def instrumented_pass(self, step, steps):
        self.profiling(True)
        for i in range(steps):
            step(i)
        self.sync()
        k = self.kernel_time()
        self.small = self.small_kernels()
        self.profiling(False)
        return k

    def small_kernels(self):
        """The HBM-bound kernels around a per-layer tower (board encode in front, head kernels behind — the BN / ReLU tails
        themselves are fused into the convolutions' epilogues): average launch time and achieved GB/s against the 8 TB/s HBM
        peak, from the same HIP events as the dominant kernel.  Algorithmic bytes: what the kernel must read and write."""
        info = self.info
        byt = self.benchlib.side_kernel_bytes(self.tower_path, self.dtype_name, self.B, info.board_h * info.board_w,
                                              info.tower_channels, info.input_channels, info.input_scalar_channels,
                                              self.stride, info.policy_len)
        out = []
        for name, nbytes in byt.items():
            ms = n = 0
            for e in self.engines:
                m, k = e.kernel_time(name)
                ms, n = ms + m, n + k
            if n:
                out.append(self.benchlib.bandwidth_record(name, n, ms / n * 1e3, nbytes))
        return out

    def kernel_time(self):
        ms, n = 0.0, 0
        for e in self.engines:
            m, k = e.kernel_time(self.kernel)
            ms += m
            n += k
        return ms, n

    def condition(self, step, seconds):
        """Conditioning, not measurement: takes the device out of its idle power state and through first-launch set-up,
        so that a short --steps/--warmup run measures the same steady state as a long one."""
        t_end = time.perf_counter() + seconds
        i = 0
        burst = 4 * len(self.engines)
        while time.perf_counter() < t_end:
            step(i)
            i += 1
            if i % burst == 0:
                # (back-pressure: launches are enqueued far faster than a slow network executes them — a 7.6 ms launch
                # enqueued for a quarter of a second is a minute of queued work that the sync below would then wait for)
                self.sync()
        self.sync()
        return i

    def check_finite(self):
        s = self.outs[0][0].to_host(self.np.float32, (self.B, 5))
        assert self.np.isfinite(s).all(), "non-finite network output"

    def check_finite_decoded(self):
        """One more batch through the decoded boundary, read back: finite values, every board's probabilities sum to 1."""
        np = self.np
        e = self.host_engines[0]
        e.submit_packed_decoded_csr(0, self.bits, self.scalars_in, self.moves[0], self.moves[1])
        v, probs = e.wait_decoded(0, self.moves[0])
        assert np.isfinite(v).all(), "non-finite decoded values"
        sums = np.array([p.sum() for p in probs])
        assert np.abs(sums - 1.0).max() < 1e-3, f"decoded probabilities do not sum to 1: {sums.min()}..{sums.max()}"

    def flops_per_launch(self, launches_per_step):
        info, B = self.info, self.B
        hw, C = info.board_h * info.board_w, info.tower_channels
        tower = 2.0 * hw * 9 * C * (info.input_channels + 2 * info.tower_depth * C)  # per board, direct conv
        p = self.tower_path
        if p.endswith("+heads"):
            return info.flops_per_eval * B  # one launch = tower + heads for one batch
        if p.startswith("attention_tower"):  # one launch = the AttentionTower of one batch: all but the heads' FLOP
            heads = 2.0 * (hw * 4 * C + 32 * 4 * hw + 5 * 32)  # ScalarHead(board, C, 4, 32)
            return (info.flops_per_eval - heads) * B if info.policy_kind != 2 else \
                (info.flops_per_eval - heads - 2.0 * (64 * 2 * C * C + 8 * 3 * C * C + 64 * 88 * C)) * B
        if p.startswith("tower_resident"):
            return tower * B  # one launch = the whole tower for one batch
        if p in ("board_conv_f16", "board_conv_split16"):
            return 2.0 * hw * 9 * C * C * B  # one launch per 3x3 tower convolution (split16: algorithmic FLOP, a third of the executed)
        return info.flops_per_eval * B / max(launches_per_step, 1)  # per-layer launches: average over the step

    def algorithmic_bytes_per_launch(self):
        """What one launch of the dominant kernel must move at least: its weights once, its inputs and its outputs
        (DESIGN.md §8).  One-launch paths: the weight stream + packed boards in + (scalars, policy) out.  Per-layer board
        paths: the activation tensor in and out, the residual on every other layer, one layer's weights."""
        info, B = self.info, self.B
        hw, C = info.board_h * info.board_w, info.tower_channels
        wbytes = {"f16": 2, "f32": 4, "f32split16": 4}[self.dtype_name]  # (split16: a (hi, lo) f16 pair per weight)
        p = self.tower_path
        io = B * (self.stride + 4 * info.input_scalar_channels + 4 * (5 + info.policy_len))
        if p.endswith("+heads"):
            return int(info.param_count * wbytes + io)
        if p.startswith("attention_tower"):  # its weights (all but the heads'), the encoded planes in, the tower output out
            head_params = 4 * C + 4 + 32 * 4 * hw + 32 + 5 * 32 + 5 + (2 * C * C + 2 * C + 3 * C * C + 3 * C if info.policy_kind == 2 else 0)
            act = B * hw * (wbytes if self.dtype_name == "f16" else 4)
            return int((info.param_count - head_params) * wbytes + act * (-(-info.input_channels // 32) * 32 + C))
        tower_params = 9 * C * (info.input_channels + 2 * info.tower_depth * C)
        if p.startswith("tower_resident"):
            act = B * hw * C * (4 if self.dtype_name != "f16" else 2)  # the tower output written for the head kernels
            return int(tower_params * wbytes + B * (self.stride + 4 * info.input_scalar_channels) + act)
        if p in ("board_conv_f16", "board_conv_split16"):
            act = B * hw * C * (2 if p == "board_conv_f16" else 4)
            return int(2.5 * act + 9 * C * C * wbytes)
        return None

    def roofline(self, k_ms, k_n, steps, evals_per_s_per_gpu, concurrent=None):
        peak = 157.3 if self.dtype_name == "f32" else 2500.0  # f32split16: algorithmic FLOP against the f16 matrix cores
        avg_ms = k_ms / max(k_n, 1)
        fpl = self.flops_per_launch(k_n / max(steps, 1))
        achieved = fpl / (avg_ms * 1e-3) / 1e12 if k_n else 0.0
        wgs, per = self.engines[0].launch_geometry(self.B)
        traffic, source = committed_traffic(self.kernel, self.name, self.B)
        alg_bytes = self.algorithmic_bytes_per_launch()
        # `achieved` / `frac` are the CHIP's: the algorithmic FLOP of all launches running side by side / wall time /
        # peak — a resident launch covers `workgroups_per_launch` of the 256 CUs and `concurrent_launches` of them run
        # at once, so one launch's own rate (`launch_achieved` = its FLOP / its average duration, `launch_frac`) is a half
        # or a third of what the chip does.  `traffic` = HBM-side bytes of one launch from the committed PMC passes,
        # `traffic_ratio` = traffic / the launch's algorithmic bytes (weights once + what it must read and write).
        chip = evals_per_s_per_gpu * self.info.flops_per_eval / 1e12
        if concurrent is None:  # the host boundary drives ONE engine of a one-launch path: its two slot streams run side by side
            concurrent = 2 if self.tower_path.endswith("+heads") else len(self.host_engines)
        # a fraction above 1 means the FLOP model or the timing is wrong: say so instead of printing it as an achievement
        bad = [k for k, v in (("frac", chip / peak), ("launch_frac", achieved / peak)) if not 0.0 <= v <= 1.0]
        return {**({"error": f"{', '.join(bad)} outside [0, 1]: the FLOP model or the launch timing is wrong"} if bad else {}),
                "bound": "mfma", "kernel": self.kernel, "achieved": round(chip, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(chip / peak, 4), "launch_achieved": round(achieved, 2), "launch_frac": round(achieved / peak, 4),
                "traffic": traffic, "traffic_source": source, "algorithmic_bytes_per_launch": alg_bytes,
                "traffic_ratio": round(traffic / alg_bytes, 2) if traffic and alg_bytes else None,
                "avg_launch_ms": round(avg_ms, 5), "launches": k_n, "flop_per_launch": fpl,
                "launch_timing": ("HIP events around every launch of an instrumented pass behind the timed steps (per-layer path: "
                                  "events around each of its ~85 launches per batch cost 1.7 % of the rate)") if self.per_layer
                else "HIP events around every launch of the timed steps",
                "workgroups_per_launch": wgs, "boards_per_workgroup": per or None,
                "concurrent_launches": concurrent,
                "chip_frac": round(chip / peak, 4)}

    def close(self):
        self.sync()
        for e in self.engines:
            e.close()
        for a, b in self.outs:
            a.free()
            b.free()
        self.d_bits.free()
        self.d_sin.free()
        self.model.close()


def sub_record(capi, synth, name, dtype_name, device, seconds, prewarm, model_kw=None):
    """~`seconds` of timed steps of another BASELINE config on this GPU (N=1 only), through the same decoded host boundary
    as `value`."""
    wl = WORKLOADS[name]
    w = Workload(capi, synth, name, dtype_name, wl["batch"], wl["engines"][dtype_name], device, seed=1000, model_kw=model_kw)
    try:
        step = w.step_host_decoded
        w.condition(step, prewarm)
        t0 = time.perf_counter()
        probe = max(4, 2 * len(w.engines))
        for i in range(probe):
            step(i)
        w.sync()
        per = (time.perf_counter() - t0) / probe
        steps = max(probe, int(seconds / max(per, 1e-6)))
        if not w.per_layer:
            w.profiling(True)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        w.sync()
        dt = time.perf_counter() - t0
        if w.per_layer:
            k_ms, k_n = w.instrumented_pass(step, min(steps, 50))
        else:
            k_ms, k_n = w.kernel_time()
            w.profiling(False)
        w.check_finite_decoded()
        value = steps * w.B / dt
        return {"metric": f"self-play NN evals/sec, {wl['label']}, 1 GPU", "workload": name, "dtype": dtype_name,
                "value": round(value, 1), "unit": "evals/s", "batch": w.B, "steps": steps, "boundary": "decoded host boundary",
                "ms_per_step": round(dt / steps * 1e3, 4), "engines_per_gpu": len(w.host_engines), "tower_path": w.tower_path,
                "flop_per_eval": w.info.flops_per_eval,
                "roofline": w.roofline(k_ms, k_n, min(steps, 50) if w.per_layer else steps, value),
                **({"hbm_bound_kernels": w.small} if getattr(w, "small", None) else {})}
    finally:
        w.close()


# The seam configurations of the default line.  `work` = what the executor thread does besides submit / wait
# (tests/cpp/bench_executor.cpp): "real" = what kzero_amd/rust/hip.rs does — ChessStdMapper::encode_input at submit and,
# per board, a fresh move generation and a SipHash map lookup per move (chess.rs:202-210), at submit with the softmax on
# the GPU (device_decode 1) or at decode with the softmax on the thread (0: the reference's decode_output);
# "pre" = the generators computed the policy indices (move generation + lookups on their threads) and the GPU decodes;
# "packed" = pre-packed boards and move lists: the channel and PCIe alone (what rounds 1-3 reported as `seam`).
SEAM_CONFIGS = [
    dict(name="hip.rs default (KZ_HIP_DECODE=device, KZ_HIP_PREP_THREADS=0): encode_input and move lists at submit on the executor "
              "thread; gather + softmax inside the network's launch; one executor thread",
         work="real", gpu_threads=1, depth=3, device_decode=1, helpers=0),
    dict(name="the same with one prep helper thread (KZ_HIP_PREP_THREADS=1: the batch's host work cut in two; the C++ mirror's "
              "persistent helper — the Rust shim's scoped threads spawn per batch and are not what this times)",
         work="real", gpu_threads=1, depth=3, device_decode=1, helpers=1),
    dict(name="KZ_HIP_DECODE=host (the reference's decode_output on the executor thread), one executor thread", work="real",
         gpu_threads=1, depth=3, device_decode=0, helpers=0),
    dict(name="KZ_HIP_DECODE=host, gpu_threads_per_device = 4", work="real", gpu_threads=4, depth=2, device_decode=0, helpers=0),
    dict(name="policy indices computed by the generators and carried in the job, decode on the GPU, one executor thread",
         work="pre", gpu_threads=1, depth=3, device_decode=1, helpers=0),
    dict(name="channel and PCIe alone (pre-packed boards and move lists: what rounds 1-3 reported)", work="packed",
         gpu_threads=1, depth=3, device_decode=0, helpers=0),
]


def bench_executor_exe():
    exe = os.path.join(REPO, "tests", "cpp", "build", "bench_executor")
    srcs = [os.path.join(REPO, "tests", "cpp", n) for n in ("bench_executor.cpp", "bench_chess.hpp")]
    host = os.path.join(REPO, "kzero_amd", "csrc", "host")
    srcs += [os.path.join(host, n) for n in sorted(os.listdir(host)) if n.endswith(".hpp")]  # (rebuilt when the mirror changes)
    lib_dir = os.path.join(REPO, "kzero_amd")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(x) for x in srcs):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", srcs[0], "-o", exe, f"-L{lib_dir}", "-lkzhip",
                               f"-Wl,-rpath,{lib_dir}"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    return exe


def seam_run(exe, model_path, seconds, cfg, dtype, devices):
    env = dict(os.environ)
    if len(devices) > 1:
        env["KZ_BENCH_FULL_AFFINITY"] = "1"  # (this process bound itself to ONE GPU's NUMA node; the child drives several)
    out = subprocess.run([exe, model_path, str(seconds), str(cfg["gpu_threads"]), "6", "256", "8", dtype, str(cfg["depth"]),
                          str(cfg["device_decode"]), ",".join(str(d) for d in devices), cfg["work"], str(cfg.get("helpers", 0))],
                         capture_output=True, text=True, timeout=120, env=env)
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    if not rec["evals_per_s"] > 0:
        raise RuntimeError("the seam answered no evaluation in the timed window (stalled)")
    return {"config": cfg["name"], "work": cfg["work"], "value": rec["evals_per_s"], "unit": "evals/s", "fill": rec["fill"],
            "executor_threads": rec["gpu_threads"], "pipeline_depth": rec["pipeline_depth"], "device_decode": rec["device_decode"],
            "generator_threads": rec["generator_threads"], "concurrent_games": rec["concurrent_games"],
            "gpu_batch": rec["gpu_batch"], "search_batch": rec["search_batch"], "seconds": rec["seconds"],
            # executor_work_util = CPU time outside kz_engine_wait* / wall (the HIP runtime polls inside the wait: an executor
            # thread shows ~100 % CPU however little work it does)
            "executor_work_util": rec["executor_work_util"], "executor_work_util_max": rec["executor_work_util_max"],
            "executor_cpu_util": rec["executor_cpu_util"], "generator_cpu_util": rec["generator_cpu_util"],
            "prep_helpers": rec.get("prep_helpers"), "helper_cpu_util": rec.get("helper_cpu_util"),
            "helper_cpu_s_per_Meval": rec.get("helper_cpu_s_per_Meval"),
            "host_cpu_s_per_Meval": rec["host_cpu_s_per_Meval"], "executor_work_s_per_Meval": rec["executor_work_s_per_Meval"],
            "executor_cpu_s_per_Meval": rec["executor_cpu_s_per_Meval"], "generator_cpu_s_per_Meval": rec["generator_cpu_s_per_Meval"],
            "projection_8gpu": rec["projection_8gpu"], "devices": rec.get("devices"),
            "per_device_evals_per_s": rec.get("per_device_evals_per_s")}


def seam_record(blob, seconds, devices=(0,), dtype="f16", configs=None):
    """BASELINE configs[2] as the reference runs it: generator threads -> job channel -> executor loop
    (`RunCondition::JobCount`, sizing of server_alphazero.rs:47-55) -> `HipNetwork` (host encode, the engine over PCIe,
    decode_output) -> replies, counted like the collector's `real` evals/s, with the host work of the Rust shim ON the
    threads that will do it (SEAM_CONFIGS) and every thread's CPU utilisation in the record.  The host side is the C++
    mirror of the Rust seam (tests/cpp/bench_executor.cpp over kzero_amd/csrc/host/), search_batch 8 (the record carries
    the number of concurrent games the sizing rule gives).  The first configuration is the record's own `value`, the
    others sit in `variants`.  Built with g++ on first use; any failure here is reported in the record, it never fails
    the bench."""
    try:
        exe = bench_executor_exe()
        path = os.path.join("/tmp", f"kz_bench_seam_{os.getpid()}.kzm")
        with open(path, "wb") as f:
            f.write(blob)
        try:
            runs = []
            for cfg in (configs or SEAM_CONFIGS):
                try:
                    runs.append(seam_run(exe, path, seconds, cfg, dtype, devices))
                except Exception as ex:  # noqa: BLE001
                    runs.append({"config": cfg["name"], "error": f"{type(ex).__name__}: {ex}"[:300]})
        finally:
            os.unlink(path)
        head = dict(runs[0])
        head["metric"] = ("NN evals/sec through generators -> job channel -> executor loop -> PCIe -> replies, host work of the "
                          f"Rust shim on the executor thread, {dtype}")
        head["dtype"] = dtype
        head["variants"] = runs[1:]
        head["host"] = "C++ mirror of the Rust seam (kzero_amd/csrc/host), tests/cpp/bench_executor.cpp"
        return head
    except Exception as ex:  # noqa: BLE001 (diagnostic record only)
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}


def select_device(benchlib, dist, rank, local_rank, world, ndev, bus_id_of):
    """The device this rank drives and every rank's PCI bus id, or (None, message): LOCAL_RANK among the visible GPUs, or
    the single visible GPU of a rank whose launcher masked the others; the ranks must report `world` distinct bus ids."""
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if os.environ.get("KZ_BENCH_ALLOW_SHARED_GPU") == "1" and ndev < local_world:
        # REHEARSAL ONLY (tools/first_node.sh --rehearse on a one-GPU box): the real multi-rank path — gloo control plane, real
        # engines, per-rank lines, the one-process seam — with several ranks on ONE GPU.  The line says shared_gpu and is no
        # scaling measurement.
        device = local_rank % ndev
        seen = benchlib.gather_strings(dist, bus_id_of(device))
        return device, seen, None
    device, why = benchlib.pick_device(ndev, local_rank, local_world)
    # every rank takes part in the gather, also one that has no device: nobody may be left waiting in a collective
    seen = benchlib.gather_strings(dist, bus_id_of(device) if device is not None else f"none:{rank}")
    if device is None:
        return None, seen, f"bench.py: {why}"
    bad = benchlib.check_distinct(seen, world)
    if any(x.startswith("none:") for x in seen):
        bad = bad or f"a rank has no GPU: {seen}"
    return (None, seen, f"bench.py: {bad}") if bad else (device, seen, None)


def fake_main(args, benchlib, rank, local_rank, world, dist):
    """--fake-step: the launcher, the device selection and the aggregation without a GPU (tests/test_bench_launcher.py).
    KZ_FAKE_NDEV = GPUs every rank "sees" (default: one per rank); a fake GPU's bus id is its index behind
    HIP_VISIBLE_DEVICES, so a per-rank mask gives distinct ids and a shared one collides."""
    ndev = int(os.environ.get("KZ_FAKE_NDEV", os.environ.get("LOCAL_WORLD_SIZE", world)))
    device, seen, err = select_device(benchlib, dist, rank, local_rank, world, ndev,
                                      lambda d: f"fake:{benchlib.visible_index(d)}")
    if err:
        print(err, file=sys.stderr)
        return 3

    # KZ_FAKE_SYSFS=<dir> (with kfd/, pci/, node/, dri/ below it): the NUMA lookup a real rank does before its first HIP
    # call, on a fake topology (never applied: the test process keeps its affinity)
    numa = {"bus_id": None, "numa_node": None, "cpus": None, "bound": False}
    fake_sysfs = os.environ.get("KZ_FAKE_SYSFS")
    if fake_sysfs:
        numa = benchlib.bind_to_gpu_numa(device, kfd_root=os.path.join(fake_sysfs, "kfd"), pci_root=os.path.join(fake_sysfs, "pci"),
                                         node_root=os.path.join(fake_sysfs, "node"), dri_root=os.path.join(fake_sysfs, "dri"), apply=False)

    def step(i):
        time.sleep(args.fake_step * 1e-3 * (1 + rank))  # (rank r is r+1 times slower: the per-rank lines must show it)
    own = []
    regions = benchlib.run_timed_regions(step, lambda: None, args.steps, args.warmup, args.repeats, dist, owns=own)
    elapsed = benchlib.median_region(regions)
    per_rank = benchlib.gather_objects(dist, {"rank": rank, "device": device, "bus_id": numa["bus_id"] or seen[rank],
                                              "numa_node": numa["numa_node"], "numa_bound": numa["bound"], "host_cpus": numa["cpus"],
                                              "evals_s": round(args.steps / benchlib.median_region(own), 3)})
    if rank == 0:
        from kzero_amd import benchline
        print(benchline.emit({"metric": "fake steps/sec (launcher test)", "value": round(args.steps * world / elapsed, 3),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "regions": len(regions), "value_min": round(args.steps * world / max(regions), 3),
                          "value_max": round(args.steps * world / min(regions), 3),
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "fake", "devices_seen": sorted(set(seen)),
                          "per_rank": per_rank,
                          "config": {"workload": "sleep", "parallelism": f"dp{world} (no collective)"}},
                             os.environ.get("KZ_BENCH_FULL_RECORD")), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    from kzero_amd import benchlib, benchline
    rank, local_rank, world, distributed = benchlib.rank_info()
    if args.gpus > 1 and not distributed:
        return spawn_ranks(args)  # nothing in this process has touched the GPU
    if distributed and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if distributed:
        import torch  # noqa: F401  (control plane only; imported before the HIP library on purpose)
    dist = benchlib.init_control_plane()
    if args.fake_step is not None:
        return fake_main(args, benchlib, rank, local_rank, world, dist)

    # NUMA: before the first HIP call, pin this rank to the CPUs next to its GPU (bus id from the KFD topology in sysfs):
    # the executor threads and the zero-copy pinned staging (first touched by the allocating thread) then sit on the
    # GPU's node.  A rank whose launcher masked the other GPUs drives ordinal 0 of its mask.
    masked = len((os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or "").split(",")) == 1 \
        and bool(os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES"))
    numa = benchlib.bind_to_gpu_numa(0 if masked else local_rank)

    from kzero_amd import capi, synth
    ndev = capi.device_count()
    # never fold several ranks onto one GPU: a scaling curve measured that way would be wrong without any error
    device, devices_seen, err = select_device(benchlib, dist, rank, local_rank, world, ndev, capi.device_pci_bus_id)
    if err:
        print(f"{err} (rank {rank}, {ndev} GPU(s) visible)", file=sys.stderr)
        return 3
    if numa["bus_id"] is None or numa["bus_id"].lower() != devices_seen[rank].lower():
        # the sysfs guess was not available (or wrong): bind now, from the bus id HIP reports — later than intended
        numa = dict(benchlib.bind_to_gpu_numa(device, bus_id=devices_seen[rank]), bound_before_first_hip_call=False)
    else:
        numa["bound_before_first_hip_call"] = numa["bound"]

    w = Workload(capi, synth, args.workload, args.dtype, args.batch, args.engines, device, benchlib.board_seed(rank))
    B = args.batch

    def timed(step, own, profile=True, prewarm=args.prewarm):
        """`--repeats` regions of K steps of `step`, each bracketed by barrier + device sync; with `profile`, HIP events
        around every launch of the timed steps (per-layer paths: an instrumented pass of the same steps behind them)."""
        w.condition(step, prewarm)
        regs = benchlib.run_timed_regions(step, w.sync, args.steps, args.warmup, args.repeats, dist,
                                          on_timed_start=(lambda: w.profiling(True)) if profile and not w.per_layer else None, owns=own)
        k = (0.0, 0)
        if profile:
            if w.per_layer:
                k = w.instrumented_pass(step, min(args.steps, 50))
            else:
                k = w.kernel_time()
                w.profiling(False)
        vals = [benchlib.whole_job_value(args.steps, B, world, t) for t in regs]
        med = benchlib.median_region(regs)
        return {"regions": regs, "elapsed": med, "value": benchlib.whole_job_value(args.steps, B, world, med),
                "values": vals, "k_ms": k[0], "k_n": k[1]}

    # ---- timed regions 1 (`value`): the K steps through the boundary a caller has — host boards and move lists in, decoded
    # values and legal-move probabilities out (kz_engine_submit_packed_decoded -> kz_engine_wait_decoded: what hip.rs runs
    # by default, replacing cudnn.rs:55-87 + common.rs:16-100); H2D and D2H inside the timed region ----
    own, r_own, h_own = [], [], []
    decoded = args.boundary == "decoded"
    main = timed(w.step_host_decoded if decoded else w.step_resident, own)
    if decoded:
        w.check_finite_decoded()
    else:
        w.check_finite()
    regions, elapsed, value, region_values, k_ms, k_n = (main[k] for k in ("regions", "elapsed", "value", "values", "k_ms", "k_n"))
    info = w.info
    peak = 157.3 if args.dtype == "f32" else 2500.0

    # ---- timed regions 2 (`value_device_resident`): packed boards already in HBM, raw outputs left in HBM ----
    res = main
    if decoded and not args.no_host_io:
        res = timed(w.step_resident, r_own, prewarm=min(args.prewarm, 0.1))
        w.check_finite()

    # ---- timed regions 3 (`value_host_boundary_raw`): host boards in, the raw (scalars, policy) rows out
    # (kz_engine_submit_packed -> kz_engine_wait_view: the reference's evaluate_batch alone, 7.5 KB D2H per chess eval) ----
    host = None
    if decoded and not args.no_host_io:
        raw = timed(w.step_host, h_own, prewarm=min(args.prewarm, 0.1))
        host = {"value": round(raw["value"], 1), "value_min": round(min(raw["values"]), 1), "value_max": round(max(raw["values"]), 1),
                "unit": "evals/s", "steps": args.steps, "regions": len(raw["regions"]),
                "ms_per_step": round(raw["elapsed"] / args.steps * 1e3, 4), "of_value": round(raw["value"] / value, 4),
                "engines_per_gpu": len(w.host_engines),
                "entry_points": "kz_engine_submit_packed -> kz_engine_wait_view (pinned staging, "
                                f"{capi.KZ_ENGINE_SLOTS} slots per engine)",
                "h2d_bytes_per_eval": int(w.stride + 4 * info.input_scalar_channels),
                "d2h_bytes_per_eval": int(4 * (5 + info.policy_len)),
                "kernel_avg_launch_ms": round(raw["k_ms"] / max(raw["k_n"], 1), 5),
                "chip_frac": round(raw["value"] / world * info.flops_per_eval / 1e12 / peak, 4)}
    resident = {"value": round(res["value"], 1), "value_min": round(min(res["values"]), 1), "value_max": round(max(res["values"]), 1),
                "regions": len(res["regions"]), "ms_per_step": round(res["elapsed"] / args.steps * 1e3, 4),
                "of_value": round(res["value"] / value, 4), "engines_per_gpu": len(w.engines),
                "entry_points": "kz_engine_enqueue_packed_device (no PCIe in the timed region; no caller of the Network trait can obtain it)",
                "kernel_avg_launch_ms": round(res["k_ms"] / max(res["k_n"], 1), 5),
                "chip_frac": round(res["value"] / world * info.flops_per_eval / 1e12 / peak, 4)}

    # every rank's own line (a slow rank — NUMA, thermals, a bad link — must be visible next to the aggregate)
    per_rank = benchlib.gather_objects(dist, {
        "rank": rank, "device": device, "bus_id": devices_seen[rank], "numa_node": numa["numa_node"],
        "numa_bound": numa["bound"], "numa_bound_before_first_hip_call": numa.get("bound_before_first_hip_call", False),
        "host_cpus": numa["cpus"], "evals_s": round(args.steps * B / benchlib.median_region(own), 1),
        "avg_launch_ms": round(k_ms / max(k_n, 1), 5),
        "device_resident_evals_s": round(args.steps * B / benchlib.median_region(r_own), 1) if r_own else None})
    if rank != 0:
        w.close()
        if dist is not None:
            dist.barrier()  # (rank 0's one-process seam run starts once every rank has released its GPU)
            dist.destroy_process_group()
        return 0

    wl = WORKLOADS[args.workload]
    metric = ("self-play NN evals/sec (node), Chess 20x256 ResNet b=256, 1/2/4/8 GPU" if args.is_default_line else
              f"self-play NN evals/sec (node), {wl['label'].rsplit(' b=', 1)[0]} b={B}, {world} GPU")
    roof = w.roofline(k_ms, k_n, min(args.steps, 50) if w.per_layer else args.steps * len(regions), value / world,
                      concurrent=None if decoded else len(w.engines))
    roof["device_resident_frac"] = resident["chip_frac"]
    n_moves = int(w.moves[0][-1]) / B if w.moves is not None else 0
    out = {
        "metric": metric,
        # value = the MEDIAN of `regions` timed regions of `steps` steps each through the decoded host boundary (every region
        # bracketed by barrier + device sync); ms_per_step x steps = that region
        "value": round(value, 1), "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "regions": len(regions), "value_min": round(min(region_values), 1), "value_max": round(max(region_values), 1),
        "region_values": [round(v, 1) for v in region_values],
        # the same K steps with the packed boards already in HBM and the raw outputs left there (what `value` was up to round 5)
        "value_device_resident": resident["value"] if decoded and not args.no_host_io else None,
        # ... and through the raw host boundary (the reference's evaluate_batch alone: 7.5 KB of policy rows back per chess eval)
        "value_host_boundary_raw": host["value"] if host else None,
        "schema": "r6: value = the decoded host boundary (hip.rs's default entry points), median region; value_device_resident "
                  "= r1-r5's value; roofline follows value: achieved/frac = the chip's (all concurrent launches), launch_frac = one launch's",
        "config": {"workload": f"{args.workload} {wl['head']} head, executor batch {B}",
                   "boundary": ("host boards + move lists in, decoded values + legal-move probabilities out "
                                "(kz_engine_submit_packed_decoded/_wait_decoded), PCIe inside the timed region") if decoded else
                   "device-resident (--boundary resident): packed boards in HBM, raw outputs left in HBM",
                   "engines_per_gpu": len(w.host_engines) if decoded else len(w.engines), "conditioning_s": args.prewarm, "tower_path": w.tower_path,
                   "parallelism": f"dp{world} (no collective)", "flop_per_eval": w.info.flops_per_eval,
                   "weights": "PyTorch-default uniform init, seed 0 (others[].weights: the same network with other statistics)",
                   "legal_moves_per_board": round(n_moves, 1),
                   "h2d_bytes_per_eval": int(w.stride + 4 * info.input_scalar_channels + 8 + 4 * n_moves),
                   "d2h_bytes_per_eval": int(20 + 4 * n_moves),
                   "device_resident_evals_s": resident["value"],
                   "host_boundary_raw_evals_s": host["value"] if host else None},
        "devices_seen": sorted(set(devices_seen)),
        "per_rank": per_rank,
        "roofline": roof,
        "device_resident": resident,
    }
    if len(set(devices_seen)) < world:  # (only reachable with KZ_BENCH_ALLOW_SHARED_GPU=1)
        out["shared_gpu"] = True
        out["data"] = "synthetic; REHEARSAL: %d ranks share %d GPU(s) — not a scaling measurement" % (world, len(set(devices_seen)))
    if host:
        out["host_boundary_raw"] = host
    if getattr(w, "small", None):
        out["hbm_bound_kernels"] = w.small
    blob, bits, scalars_in = w.blob, w.bits, w.scalars_in
    w.close()
    if world == 1 and args.is_default_line and not args.no_others:
        out["others"] = []
        for n, d in OTHERS:  # (a sub-record that fails says so in its place: it must not take the headline line with it)
            try:
                out["others"].append(sub_record(capi, synth, n, d, device, args.other_seconds, args.prewarm))
            except Exception as ex:  # noqa: BLE001
                out["others"].append({"workload": n, "dtype": d, "error": f"{type(ex).__name__}: {ex}"[:300]})
        # the headline network with other weight statistics, same box, same run (the matrix cores' power depends on the data)
        for label, kw in (WEIGHT_VARIANTS if args.weight_variants else []):
            try:
                rec = sub_record(capi, synth, "chess-20x256", "f16", device, args.other_seconds, args.prewarm, model_kw=kw)
                rec["weights"] = label
                out["others"].append(rec)
            except Exception as ex:  # noqa: BLE001
                out["others"].append({"workload": "chess-20x256", "dtype": "f16", "weights": label,
                                      "error": f"{type(ex).__name__}: {ex}"[:300]})
    if "others" in out:  # the two numbers a reader needs side by side: the f16 headline and the <= 1e-4 path of the same network
        par = [o for o in out["others"] if o.get("workload") == "chess-20x256" and o.get("dtype") == "f32split16" and "value" in o]
        if par:
            out["value_parity_default"] = par[0]["value"]
            out["config"]["parity_default_dtype"] = "f32split16 (<= 1e-4 vs the oracle; hip.rs's default unless KZ_HIP_DTYPE=f16)"
            out["config"]["parity_default"] = ("KZ_DTYPE_F32_SPLIT16 (<= 1e-4 against the oracle; what the Rust binding runs unless "
                                               f"KZ_HIP_DTYPE=f16): {par[0]['value']} evals/s, frac {par[0]['roofline']['frac']}")
    if world == 1 and args.is_default_line and not args.no_seam and not args.no_others:
        out["seam"] = seam_record(blob, args.seam_seconds)
        # ... and at the arithmetic the Rust binding defaults to (KZ_HIP_DTYPE=parity -> split16 for this network)
        out["seam_parity"] = seam_record(blob, args.seam_seconds, dtype="f32split16", configs=SEAM_CONFIGS[:3] + SEAM_CONFIGS[4:5])
    if world > 1:
        dist.barrier()  # every rank has closed its engines
        if args.is_default_line and not args.no_seam and not args.no_others:
            # the topology the reference and the Rust drop-in use: ONE process, a thread set per device
            # (rust/kz-selfplay/src/server/server.rs:323-331) over exactly the GPUs the ranks of this run drove
            drove = [r["device"] for r in per_rank]
            if out.get("shared_gpu"):
                out["seam_one_process"] = {"skipped": "rehearsal on a shared GPU: one process would drive the same device twice"}
            elif ndev >= world and len(set(drove)) == world:
                rec = seam_record(blob, args.seam_seconds, devices=drove, configs=SEAM_CONFIGS[:1])
                rec["metric"] = rec.get("metric", "seam") + f", one process over devices {drove}"
                out["seam_one_process"] = rec
            else:
                out["seam_one_process"] = {"skipped": f"ranks see {ndev} GPU(s) each (masked visibility): rank 0 cannot drive "
                                                      f"the other ranks' devices from one process"}
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(blob, bits, scalars_in, args.cpu_seconds)
        except Exception as ex:  # noqa: BLE001 (the oracle library is test infrastructure: its absence must not cost the line)
            out["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": None, "kind": "port",
                                   "error": f"{type(ex).__name__}: {ex}"[:300]}
    # ONE line of at most 4 KB on stdout (kzero_amd/benchline.py); everything measured goes to bench_full.json beside it
    print(benchline.emit(out, FULL_RECORD), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
