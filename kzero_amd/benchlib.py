"""Timing and multi-rank aggregation for bench.py, separated so that the N>1 path can be tested on CPU (gloo).

The hot path shards by game -> device with no exchange step (rust/kz-selfplay/src/server/server.rs:325-331), so ranks
never exchange tensors: torch.distributed only carries the barrier and the max of the elapsed time.
"""
import os
import time
from typing import Callable, Optional, Tuple


def rank_info() -> Tuple[int, int, int, bool]:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    return rank, local_rank, world, distributed


def init_control_plane():
    """Returns the torch.distributed module with an initialised gloo group, or None when not launched by torchrun."""
    rank, _, world, distributed = rank_info()
    if not distributed:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        # gloo's C++ side announces its connections on stdout; stdout carries exactly one JSON line, so the
        # announcement goes to stderr
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return dist


def gather_strings(dist, s: str):
    """Every rank's string, in rank order, on every rank (e.g. the PCI bus id of the GPU a rank drives)."""
    if dist is None:
        return [s]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, s)
    return out


def board_seed(rank: int) -> int:
    """Every rank evaluates its own boards (independent units: weak scaling)."""
    return 1000 + rank


def gather_objects(dist, obj):
    """Every rank's (picklable) record, in rank order, on every rank — the per-rank lines of an N > 1 run."""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


# ------------------------------------------------------------------------------------------------------------------
# which GPU a rank drives, and where its host threads and pinned staging should live
# ------------------------------------------------------------------------------------------------------------------
def pick_device(ndev: int, local_rank: int, local_world: int):
    """(device ordinal, None) or (None, message).  Every rank normally sees all GPUs of the node and takes ordinal
    LOCAL_RANK.  A launcher that masks visibility per rank (HIP_VISIBLE_DEVICES=$LOCAL_RANK, common under torchrun
    wrappers) leaves every rank with ONE visible GPU: take ordinal 0 and let the distinct-PCI-bus-id check across ranks
    (check_distinct) decide whether the ranks really sit on different devices.  Anything else would fold several ranks
    onto one GPU — a scaling curve measured that way is wrong without any error — and is refused."""
    if ndev >= local_world and local_rank < ndev:
        return local_rank, None
    if ndev == 1:
        return 0, None
    return None, (f"rank with LOCAL_RANK {local_rank} of {local_world} on this node sees {ndev} GPU(s): neither one per "
                  f"rank nor a single masked device")


def check_distinct(bus_ids, world: int):
    """None when the `world` ranks reported `world` distinct PCI bus ids, else the message to fail with."""
    if len(set(bus_ids)) == world:
        return None
    return f"{world} ranks on {len(set(bus_ids))} distinct GPU(s): {list(bus_ids)}"


def kfd_gpu_nodes(root: str = "/sys/class/kfd/kfd/topology/nodes", dri_root: str = "/dev/dri"):
    """The GPU nodes of the KFD topology in its order — the order HIP enumerates devices in before HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES are applied — as PCI bus ids ("0000:c1:00.0"), read from sysfs so that a rank can find its GPU's
    NUMA node BEFORE its first HIP call.  A container may see the whole node's topology but only the GPUs it was given:
    a node whose properties cannot be read, or whose /dev/dri/renderD<drm_render_minor> this process cannot open, is a
    GPU HIP will not enumerate here; it keeps its place in the list as None (a device mask may count it)."""
    out = []
    try:
        nodes = sorted((int(n) for n in os.listdir(root) if n.isdigit()))
    except OSError:
        return out
    for n in nodes:
        props = {}
        try:
            for line in open(os.path.join(root, str(n), "properties")):
                k, _, v = line.strip().partition(" ")
                props[k] = v
        except OSError:
            out.append(None)  # not ours to look at: another tenant's GPU
            continue
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue  # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        except (KeyError, ValueError):
            continue
        minor = props.get("drm_render_minor")
        if minor is not None and os.path.isdir(dri_root) and not os.access(os.path.join(dri_root, f"renderD{minor}"), os.R_OK | os.W_OK):
            out.append(None)
            continue
        out.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}")
    return out


def kfd_gpu_bus_ids(root: str = "/sys/class/kfd/kfd/topology/nodes", dri_root: str = "/dev/dri"):
    """Bus ids of the GPUs this process can use, in KFD order."""
    return [b for b in kfd_gpu_nodes(root, dri_root) if b is not None]


def visible_index(ordinal: int, env=None):
    """KFD-order index of HIP ordinal `ordinal` under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (integer lists only;
    None when a mask is present that this cannot interpret, e.g. UUIDs or both masks at once)."""
    env = os.environ if env is None else env
    hip, rocr = env.get("HIP_VISIBLE_DEVICES", env.get("CUDA_VISIBLE_DEVICES")), env.get("ROCR_VISIBLE_DEVICES")
    if hip and rocr:
        return None
    mask = hip or rocr
    if not mask:
        return ordinal
    try:
        ids = [int(x) for x in mask.split(",") if x.strip() != ""]
    except ValueError:
        return None
    return ids[ordinal] if 0 <= ordinal < len(ids) else None


def numa_node_of(bus_id: str, root: str = "/sys/bus/pci/devices"):
    try:
        n = int(open(os.path.join(root, bus_id.lower(), "numa_node")).read().strip())
    except (OSError, ValueError):
        return None
    return n if n >= 0 else None


def node_cpus(node: int, root: str = "/sys/devices/system/node"):
    """CPUs of a NUMA node from its cpulist ("0-63,128-191")."""
    try:
        txt = open(os.path.join(root, f"node{node}", "cpulist")).read().strip()
    except OSError:
        return set()
    cpus = set()
    for part in txt.split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa(ordinal: int, bus_id: str = None, kfd_root: str = "/sys/class/kfd/kfd/topology/nodes",
                     pci_root: str = "/sys/bus/pci/devices", node_root: str = "/sys/devices/system/node", env=None,
                     apply: bool = True, dri_root: str = "/dev/dri"):
    """Pins this process to the CPUs of the NUMA node its GPU hangs off, so that the executor threads run and the
    zero-copy pinned staging (hipHostMalloc, first touched by the allocating thread) lands next to the device.  Call it
    before the first HIP call with `bus_id=None` (the bus id is then taken from the KFD topology); call it again with
    the bus id HIP reports to verify/correct.  Returns a record for the bench line; never raises."""
    rec = {"bus_id": bus_id, "numa_node": None, "cpus": None, "bound": False}
    try:
        if bus_id is None:
            # the mask (if any) may count the node's GPUs or only the ones this container was given: try both readings,
            # and when exactly one GPU is usable at all it is that one
            idx = visible_index(ordinal, env)
            every, usable = kfd_gpu_nodes(kfd_root, dri_root), kfd_gpu_bus_ids(kfd_root, dri_root)
            guess = None
            if idx is not None and idx < len(every) and every[idx] is not None:
                guess = every[idx]
            elif idx is not None and idx < len(usable):
                guess = usable[idx]
            elif len(usable) == 1:
                guess = usable[0]
            if guess is None:
                return rec
            bus_id = rec["bus_id"] = guess
        node = numa_node_of(bus_id, pci_root)
        rec["numa_node"] = node
        if node is None:
            return rec
        cpus = node_cpus(node, node_root)
        allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set()
        target = cpus & allowed
        rec["cpus"] = len(target)
        if target and apply and hasattr(os, "sched_setaffinity"):
            os.sched_setaffinity(0, target)
            rec["bound"] = True
    except (OSError, ValueError):
        pass
    return rec


def run_timed(step: Callable[[int], None], sync: Callable[[], None], steps: int, warmup: int, dist=None,
              on_timed_start: Optional[Callable[[], None]] = None, own: Optional[list] = None) -> float:
    """W untimed warm-up steps, then exactly `steps` steps bracketed by barrier + device sync on both sides.
    Returns the MAX over ranks of the elapsed seconds; `own`, when given, receives this rank's own time from the start
    barrier to the end of its last step (before the closing barrier): what tells a slow rank from a fast one."""
    for i in range(warmup):
        step(i)
    sync()
    if on_timed_start is not None:
        on_timed_start()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    if own is not None:
        own.append(time.perf_counter() - t0)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def run_timed_regions(step: Callable[[int], None], sync: Callable[[], None], steps: int, warmup: int, repeats: int, dist=None,
                      on_timed_start: Optional[Callable[[], None]] = None, owns: Optional[list] = None) -> list:
    """W untimed warm-up steps, then `repeats` timed regions of exactly `steps` steps, EACH bracketed by barrier + device
    sync on both sides like run_timed's one region.  Returns the regions' elapsed seconds (each the MAX over ranks);
    `owns` receives this rank's own time of every region.  A region of 20 chess batches is 10 ms: one slow launch moves
    it by 5 %, so the line reports the median region and the spread."""
    out = []
    for r in range(max(1, repeats)):
        own = []
        out.append(run_timed(step, sync, steps, warmup if r == 0 else 0, dist, on_timed_start if r == 0 else None, own))
        if owns is not None:
            owns.append(own[0])
    return out


def median_region(regions: list) -> float:
    """The median of the regions' elapsed times (the upper one of an even count: never better than measured)."""
    return sorted(regions)[len(regions) // 2]


def whole_job_value(steps: int, batch: int, world: int, elapsed_max: float) -> float:
    """evals/s of the whole job: the units all ranks processed / the slowest rank's time."""
    return steps * batch * world / elapsed_max


# ------------------------------------------------------------------------------------------------------------------
# byte model of the HBM-bound side kernels (bench.py's `hbm_bound_kernels`; tests/test_bench_line.py)
# ------------------------------------------------------------------------------------------------------------------
HBM_PEAK_BPS = 8e12  # /opt/skills/guides/MI355X_MICROARCH.md


def side_kernel_bytes(tower_path: str, dtype_name: str, batch: int, hw: int, channels: int, input_channels: int,
                      input_scalar_channels: int, bits_stride: int, policy_len: int) -> dict:
    """Algorithmic bytes per launch — what the kernel must read plus what it must write — of the HBM-bound kernels around a
    per-layer tower (board encode in front, head kernels behind; the BN / ReLU tails themselves are fused into the
    convolutions' epilogues).  Channel counts are the padded row widths the engine allocates (kz_engine.hip, kz_engine_create)."""
    esz = 2 if dtype_name == "f16" else 4
    C = -(-channels // 32) * 32
    # encoded input rows: 64 channels when the f16 stem runs through the board-tile kernel (kz_device_weights.hpp `stem64`)
    cin_rows = 64 if tower_path == "board_conv_f16" and input_channels <= 64 else -(-input_channels // 32) * 32
    act = batch * hw * C * esz
    # kz_split_rows converts f32 rows to (hi, lo) f16 pairs: 4 B in + 4 B out per element.  When the stem itself runs in
    # split arithmetic (board_conv_split16 and <= 32 input planes: kz_engine.hip `wts->stem_split`) it converts the ENCODED
    # INPUT rows (one 32-channel chunk), otherwise the stem's f32 output at the tower width
    split_c = 32 if tower_path == "board_conv_split16" and input_channels <= 32 else C
    return {"kz_encode_packed": batch * (bits_stride + 4 * input_scalar_channels) + batch * hw * cin_rows * esz,
            "kz_scalar_head": act + 4 * (4 * C + 32 * 4 * hw + 5 * 32) + batch * 5 * 4,
            "kz_conv1x1_split": act + C * C * 2 + batch * policy_len * 4,
            "kz_policy_conv": act + batch * policy_len * 4,
            "kz_policy_extra": act + batch * 4,
            "kz_split_rows": batch * hw * split_c * (4 + 4)}


def bandwidth_record(kernel: str, launches: int, avg_launch_us: float, nbytes: int) -> dict:
    """One `hbm_bound_kernels` row.  A fraction of the HBM peak above 1 means the byte model is wrong (round 5 printed
    kz_split_rows at 4.4 x the peak because it charged the tower width for a 32-channel conversion): it is flagged."""
    bps = nbytes / (avg_launch_us * 1e-6)
    rec = {"kernel": kernel, "launches": launches, "avg_launch_us": round(avg_launch_us, 2), "algorithmic_bytes": int(nbytes),
           "achieved_GBps": round(bps / 1e9, 1), "frac_of_hbm_peak": round(bps / HBM_PEAK_BPS, 4)}
    if bps > HBM_PEAK_BPS:
        rec["error"] = "above the HBM peak: the byte model is wrong (or the launch was served from L2)"
    return rec
