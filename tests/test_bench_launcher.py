"""bench.py's own rank launcher on CPU: `python bench.py --gpus N` must start N rank processes (one per GPU, the
reference iterates devices the same way: rust/kz-selfplay/src/server/server.rs:323-331), print ONE JSON line from rank 0
with n_gpus = N, and fail loudly — never fold ranks onto one GPU — when fewer than N GPUs are visible."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(180)
def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--fake-step", "10"],
                       env=_clean_env(), capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["warmup"] == 2
    assert rec["devices_seen"] == ["fake:0", "fake:1"]      # two distinct ranks reported in
    # every rank's own line: rank 1 (twice as slow in the fake) must be visible next to the aggregate
    pr = rec["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and [r["device"] for r in pr] == [0, 1]
    assert pr[0]["evals_s"] > 1.5 * pr[1]["evals_s"]
    assert rec["data"] == "fake" and rec["scaling"] == "weak"
    # whole-job value: both ranks' steps over the slowest rank's time
    assert rec["value"] == pytest.approx(5 * 2 / (rec["ms_per_step"] * 5e-3), rel=1e-3)
    assert rec["ms_per_step"] >= 10.0


@pytest.mark.timeout(180)
def test_torchrun_style_env_is_honoured():
    """The driver's launch: the ranks already exist (RANK/WORLD_SIZE/MASTER_* set) — bench.py must not spawn again."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(_clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1",
                                       "--fake-step", "5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=170) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert json.loads(outs[0][0])["n_gpus"] == 2
    assert outs[1][0].strip() == ""                          # only rank 0 prints


@pytest.mark.timeout(180)
def test_world_size_mismatch_is_an_error():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--fake-step", "1"], env=env, capture_output=True,
                       text=True, timeout=170)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


@pytest.mark.timeout(300)
def test_more_ranks_than_gpus_fails_loudly():
    """On a box with fewer than 2 GPUs (this container has none; a 1-GPU box likewise) `--gpus 2` must exit non-zero
    with a message and print no JSON line."""
    from kzero_amd import capi
    try:
        ndev = capi.device_count()
    except capi.KzError:
        ndev = 0
    if ndev >= 2:
        pytest.skip("2+ GPUs visible: the real multi-GPU path runs instead")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-others"], env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "GPU" in p.stderr or "hip" in p.stderr.lower()


@pytest.mark.timeout(180)
def test_masked_visibility_one_gpu_per_rank_is_accepted():
    """A launcher that masks the GPUs per rank (HIP_VISIBLE_DEVICES=$LOCAL_RANK, common under torchrun wrappers) leaves
    every rank with ONE visible device: ordinal 0 is used and the distinct-PCI-bus-id check decides (bench.py used to
    exit 3 here although the ranks sat on different GPUs)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(_clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIP_VISIBLE_DEVICES=str(r), KZ_FAKE_NDEV="1")
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1",
                                       "--fake-step", "5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=170) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    rec = json.loads(outs[0][0])
    assert rec["devices_seen"] == ["fake:0", "fake:1"]
    assert [r["device"] for r in rec["per_rank"]] == [0, 0]   # both drive ordinal 0 of their own mask


@pytest.mark.timeout(180)
def test_masked_visibility_onto_the_same_gpu_is_refused():
    """Two ranks masked onto the SAME device report the same bus id: exit 3, no JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(_clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIP_VISIBLE_DEVICES="0", KZ_FAKE_NDEV="1")
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1",
                                       "--fake-step", "5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=170) for p in procs]
    assert [p.returncode for p in procs] == [3, 3], outs
    assert outs[0][0].strip() == "" and "distinct GPU" in outs[0][1]


def test_numa_binding_from_sysfs(tmp_path):
    """bind_to_gpu_numa on a fake sysfs tree: KFD topology -> PCI bus id of the rank's GPU (behind HIP_VISIBLE_DEVICES)
    -> numa_node -> that node's CPUs.  (apply=False: the test process keeps its affinity.)"""
    from kzero_amd import benchlib
    kfd, pci, node = tmp_path / "kfd", tmp_path / "pci", tmp_path / "node"
    # node 0 is the CPU; GPUs at 0000:05:00.0 (NUMA 0) and 0000:c5:00.0 (NUMA 1)
    for i, props in enumerate(["simd_count 0\nlocation_id 0\ndomain 0\n",
                               f"simd_count 1024\nlocation_id {0x05 << 8}\ndomain 0\n",
                               f"simd_count 1024\nlocation_id {0xc5 << 8}\ndomain 0\n"]):
        (kfd / str(i)).mkdir(parents=True)
        (kfd / str(i) / "properties").write_text(props)
    for bdf, n in (("0000:05:00.0", 0), ("0000:c5:00.0", 1)):
        (pci / bdf).mkdir(parents=True)
        (pci / bdf / "numa_node").write_text(f"{n}\n")
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    for n, cpus in ((0, allowed[:half]), (1, allowed[half:] or allowed[:half])):
        (node / f"node{n}").mkdir(parents=True)
        (node / f"node{n}" / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    dri = tmp_path / "dri"
    dri.mkdir()
    roots = dict(kfd_root=str(kfd), pci_root=str(pci), node_root=str(node), apply=False, dri_root=str(dri))
    assert benchlib.kfd_gpu_bus_ids(str(kfd), str(dri)) == ["0000:05:00.0", "0000:c5:00.0"]
    r1 = benchlib.bind_to_gpu_numa(1, env={}, **roots)
    assert (r1["bus_id"], r1["numa_node"]) == ("0000:c5:00.0", 1) and r1["cpus"] >= 1 and not r1["bound"]
    # a rank whose launcher masked the GPUs drives ordinal 0 of its mask
    r_masked = benchlib.bind_to_gpu_numa(0, env={"HIP_VISIBLE_DEVICES": "1"}, **roots)
    assert r_masked["bus_id"] == "0000:c5:00.0" and r_masked["numa_node"] == 1
    # verification path: the bus id HIP reported
    assert benchlib.bind_to_gpu_numa(0, bus_id="0000:05:00.0", **roots)["numa_node"] == 0
    # unreadable topology / UUID masks: no binding, no exception
    assert benchlib.bind_to_gpu_numa(0, env={"HIP_VISIBLE_DEVICES": "GPU-abc"}, **roots)["numa_node"] is None
    assert benchlib.bind_to_gpu_numa(0, kfd_root=str(tmp_path / "missing"), apply=False)["bound"] is False
    assert benchlib.node_cpus(0, str(node)) == set(allowed[:half])
    # a container that sees the whole node's topology but was given one GPU's render node: only that GPU is enumerated
    for i, minor in ((1, 128), (2, 129)):
        with open(kfd / str(i) / "properties", "a") as f:
            f.write(f"drm_render_minor {minor}\n")
    (dri / "renderD129").write_text("")
    assert benchlib.kfd_gpu_bus_ids(str(kfd), str(dri)) == ["0000:c5:00.0"]
    assert benchlib.bind_to_gpu_numa(0, env={}, **roots)["bus_id"] == "0000:c5:00.0"
    # ... also when a device mask counts the node's GPUs (ROCR_VISIBLE_DEVICES=1 = the second GPU of the node) or names
    # an index beyond what the container shows (one usable GPU: it is that one)
    assert benchlib.kfd_gpu_nodes(str(kfd), str(dri)) == [None, "0000:c5:00.0"]
    assert benchlib.bind_to_gpu_numa(0, env={"ROCR_VISIBLE_DEVICES": "1"}, **roots)["bus_id"] == "0000:c5:00.0"
    assert benchlib.bind_to_gpu_numa(0, env={"ROCR_VISIBLE_DEVICES": "5"}, **roots)["bus_id"] == "0000:c5:00.0"
    # pick_device: one per rank, a single masked device, or refusal
    assert benchlib.pick_device(8, 3, 8) == (3, None) and benchlib.pick_device(1, 3, 8) == (0, None)
    assert benchlib.pick_device(2, 3, 8)[0] is None
    assert benchlib.check_distinct(["a", "b"], 2) is None and "1 distinct" in benchlib.check_distinct(["a", "a"], 2)


@pytest.mark.timeout(180)
def test_shared_gpu_rehearsal_needs_the_explicit_override():
    """Two ranks that see ONE (fake) GPU: refused (exit 3) — unless KZ_BENCH_ALLOW_SHARED_GPU=1 asks for the rehearsal of the
    multi-rank path on a one-GPU box (tools/first_node.sh REHEARSE=2; the real line then says shared_gpu)."""
    import socket

    def run(extra):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(2):
            env = dict(_clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2",
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KZ_FAKE_NDEV="1", **extra)
            procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2",
                                           "--fake-step", "5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=170) for p in procs]
        return [p.returncode for p in procs], outs

    rcs, outs = run({})
    assert rcs == [3, 3] and "distinct GPU" in outs[0][1]
    rcs, outs = run({"KZ_BENCH_ALLOW_SHARED_GPU": "1"})
    assert rcs == [0, 0], outs
    rec = json.loads(outs[0][0])
    assert rec["n_gpus"] == 2 and [r["device"] for r in rec["per_rank"]] == [0, 0] and rec["regions"] == 2
