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
