"""tools/first_node.sh — the script for the first box with more than one GPU — end to end on CPU: FAKE=1 runs the same script
with `bench.py --fake-step` over KZ_FAKE_NDEV=8 fake GPUs and a fake sysfs tree (8 GPUs on 2 NUMA nodes), so that the first
real run has nothing new in it but the hardware: rank launcher, per-rank lines, PCI-bus-id de-duplication, the NUMA lookup and
the summary's checks.  Topology: one thread set per device, rust/kz-selfplay/src/server/server.rs:323-331."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(REPO, "tools", "first_node.sh")


def fake_sysfs(root, n_gpus=8):
    """KFD topology of a 2-socket box: node 0/1 = CPUs, then n_gpus GPUs, the first half on NUMA node 0."""
    kfd, pci, node, dri = (root / d for d in ("kfd", "pci", "node", "dri"))
    dri.mkdir(parents=True)
    props = ["simd_count 0\nlocation_id 0\ndomain 0\n"] * 2
    bus_ids = []
    for g in range(n_gpus):
        bus = 0x05 + 0x10 * g
        props.append(f"simd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        bus_ids.append(f"0000:{bus:02x}:00.0")
    for i, p in enumerate(props):
        (kfd / str(i)).mkdir(parents=True)
        (kfd / str(i) / "properties").write_text(p)
    for g, bdf in enumerate(bus_ids):
        (pci / bdf).mkdir(parents=True)
        (pci / bdf / "numa_node").write_text(f"{0 if g < n_gpus // 2 else 1}\n")
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    for n, cpus in ((0, allowed[:half]), (1, allowed[half:] or allowed[:half])):
        (node / f"node{n}").mkdir(parents=True)
        (node / f"node{n}" / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    return bus_ids


def _env(tmp_path, **extra):
    env = dict(os.environ, FAKE="1", OUT=str(tmp_path / "out"), PYTHON=sys.executable, **extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(600)
def test_first_node_script_runs_end_to_end_on_eight_fake_gpus(tmp_path):
    bus_ids = fake_sysfs(tmp_path / "sys")
    p = subprocess.run(["bash", SCRIPT], env=_env(tmp_path, KZ_FAKE_NDEV="8", KZ_FAKE_SYSFS=str(tmp_path / "sys")),
                       capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "first_node: 8 device(s)" in p.stdout and "first_node: ok" in p.stdout
    for n in (1, 2, 4, 8):
        rec = json.loads(open(tmp_path / "out" / f"bench_{n}.json").read())
        assert rec["n_gpus"] == n and len(rec["per_rank"]) == n and rec["regions"] == 3
        assert [r["rank"] for r in rec["per_rank"]] == list(range(n))
        assert [r["device"] for r in rec["per_rank"]] == list(range(n))
        # every rank found ITS GPU in the topology and that GPU's NUMA node: the first four on node 0, the others on node 1
        assert [r["bus_id"] for r in rec["per_rank"]] == bus_ids[:n]
        assert [r["numa_node"] for r in rec["per_rank"]] == [0 if g < 4 else 1 for g in range(n)]
        assert all(r["host_cpus"] >= 1 for r in rec["per_rank"])
        assert len(set(rec["devices_seen"])) == n
    # the summary table carries one row per N with every rank's own line
    rows = [ln for ln in p.stdout.splitlines() if ln[:2].strip() in ("1", "2", "4", "8") and "numa" in ln]
    assert len(rows) == 4 and rows[3].count("|") == 7


@pytest.mark.timeout(300)
def test_first_node_summary_flags_what_a_wrong_run_gets_wrong(tmp_path):
    """Two ranks on one GPU, a missing rank line, a rank that knows its NUMA node but is not bound to it, a straggler."""
    out = tmp_path / "out"
    out.mkdir()
    good = {"value": 1000.0, "n_gpus": 1, "data": "synthetic", "devices_seen": ["0000:05:00.0"],
            "per_rank": [{"rank": 0, "device": 0, "bus_id": "0000:05:00.0", "numa_node": 0, "numa_bound": True, "evals_s": 1000.0}]}
    bad = {"value": 1500.0, "n_gpus": 2, "data": "synthetic", "devices_seen": ["0000:05:00.0", "0000:05:00.0"],
           "per_rank": [{"rank": 0, "device": 0, "bus_id": "0000:05:00.0", "numa_node": 0, "numa_bound": True, "evals_s": 1000.0},
                        {"rank": 1, "device": 0, "bus_id": "0000:05:00.0", "numa_node": 1, "numa_bound": False, "evals_s": 500.0}]}
    (out / "bench_1.json").write_text(json.dumps(good) + "\n")
    (out / "bench_2.json").write_text(json.dumps(bad) + "\n")
    p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "first_node_summary.py"), str(out)], capture_output=True, text=True)
    assert p.returncode == 1
    assert "1 distinct PCI bus ids for 2 ranks" in p.stdout
    assert "did not bind" in p.stdout and "slowest rank at 0.50" in p.stdout
    assert "0.750" in p.stdout  # the efficiency column: 1500 / (2 x 1000)


@pytest.mark.timeout(300)
def test_first_node_reports_a_failing_step(tmp_path):
    """More ranks than (fake) GPUs that collide on a bus id: bench.py refuses, the script says which step failed and exits 1."""
    p = subprocess.run(["bash", SCRIPT], env=_env(tmp_path, KZ_FAKE_NDEV="2", HIP_VISIBLE_DEVICES="0,0"),
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 1, p.stdout + p.stderr
    assert "FAILED bench.py --gpus 2" in p.stdout
