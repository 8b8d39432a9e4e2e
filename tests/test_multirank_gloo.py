"""The N>1 path of bench.py on CPU: two processes, gloo, 127.0.0.1.  No data-path collective exists (the path shards
by game -> device); what is covered is the control plane: barrier bracketing, max-over-ranks timing, whole-job value,
and per-rank board shards."""
import os
import socket
import time

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from kzero_amd import benchlib, synth
    r, lr, w, distributed = benchlib.rank_info()
    assert (r, lr, w, distributed) == (rank, rank, world, True)
    dist = benchlib.init_control_plane()
    calls = []

    def step(i):
        calls.append(i)
        time.sleep(0.01 * (rank + 1))  # rank 1 is twice as slow

    elapsed = benchlib.run_timed(step, lambda: None, steps=5, warmup=2, dist=dist)
    bits, _ = synth.random_boards("ataxx-7", 4, seed=benchlib.board_seed(rank))
    q.put((rank, elapsed, len(calls), bits.tobytes()))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, e0, n0, b0), (r1, e1, n1, b1) = results
    assert (n0, n1) == (7, 7)          # warm-up + exactly K timed steps on every rank
    assert e0 == e1                    # both ranks hold the MAX over ranks
    assert e0 >= 5 * 0.02 * 0.9        # ... which is the slow rank's time
    assert b0 != b1                    # every rank evaluates its own boards
    from kzero_amd import benchlib
    assert benchlib.whole_job_value(5, 256, 2, e0) == pytest.approx(5 * 256 * 2 / e0)


def test_single_process_defaults(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR"):
        monkeypatch.delenv(k, raising=False)
    from kzero_amd import benchlib
    assert benchlib.rank_info() == (0, 0, 1, False)
    assert benchlib.init_control_plane() is None
    n = []
    elapsed = benchlib.run_timed(lambda i: n.append(i), lambda: None, steps=3, warmup=1, dist=None)
    assert len(n) == 4 and elapsed >= 0
