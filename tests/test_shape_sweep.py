"""Shape-generality sweep on the GPU against the oracle (tests/sweep_cases.py): tower channels off the benchmark lattice
(32 .. 512, multiples of 64 or not), 10 .. 60 input planes (GoStdMapper without territory, ChessStdMapper,
ChessHistoryMapper lengths 1..3: rust/kz-core/src/mapping/chess.rs:32-95), Ataxx 4..8 / Go 9, 13, 19 / chess boards, all
four policy heads (python/lib/model/post_act.py:26-141; dense with and without hidden layers), depths 1 and 3, ragged batches.

Every case runs through KZ_DTYPE_F32 (<= 1e-4), KZ_DTYPE_F16 (the stated f16 tolerance of tests/test_gpu_parity.py) and
the parity default of the Rust binding (split16 where `kz_model_supports_dtype` says so, else f32: <= 1e-4), and asserts the
tower path `kz_engine_create` chose against tests/golden/path_table.json — the table DESIGN §5.0 prints and
tests/test_path_table.py checks without a GPU."""
import json
import os

import numpy as np
import pytest

from kzero_amd import capi, synth
from tests import oracle_lib as O
from tests import sweep_cases
from tests.test_gpu_parity import assert_f16, assert_f32

pytestmark = pytest.mark.gpu

PATHS = json.load(open(os.path.join(O.GOLDEN, "path_table.json")))["paths"]

_cache = {}


def reference(case):
    """(blob, bits, scalars, oracle scalars, oracle policy), once per case."""
    if case.id not in _cache:
        _cache.clear()  # one network at a time: the dense heads are tens of MB
        blob = synth.random_model(case.game, case.depth, case.channels, case.head, seed=11, **case.kw)
        bits, sin = synth.random_boards(case.game, case.boards, seed=5)
        net = O.OracleNet(blob)
        x = O.encode_input_full(bits, sin, net.n_scalar, net.n_bool, net.h, net.w)
        s, p = net.forward(x, threads=min(16, case.boards))
        _cache[case.id] = (blob, bits, sin, s, p)
    return _cache[case.id]


@pytest.mark.parametrize("arith", ["f32", "f16", "parity"])
@pytest.mark.parametrize("case", sweep_cases.CASES, ids=lambda c: c.id)
def test_shape_sweep_vs_oracle(case, arith):
    blob, bits, sin, s_ref, p_ref = reference(case)
    model = capi.Model(blob=blob)
    name = sweep_cases.parity_dtype_name(model, capi) if arith == "parity" else arith
    dtype = {"f32": capi.KZ_DTYPE_F32, "f16": capi.KZ_DTYPE_F16, "f32split16": capi.KZ_DTYPE_F32_SPLIT16}[name]
    # max_batch 256: what the executor creates (a small max_batch may choose another path: the per-layer board-tile
    # kernel wants enough workgroups to fill the chip)
    eng = capi.Engine(model, 0, 256, dtype)
    assert eng.tower_path == PATHS[case.id][arith], f"{case.id} {arith}: path {eng.tower_path}"
    s, p = eng.eval_packed(bits, sin)
    if name == "f16":
        assert_f16(s, s_ref, f"{case.id} scalars")
        assert_f16(p, p_ref, f"{case.id} policy")
    else:
        assert_f32(s, s_ref, f"{case.id} {name} scalars")
        assert_f32(p, p_ref, f"{case.id} {name} policy")
    # ragged batches: one board alone gives the same rows (boards are independent; a launch that packs several boards
    # into a workgroup must not let them see each other)
    s1, p1 = eng.eval_packed(bits[-1:], sin[-1:])
    tol = 1e-5 if name != "f16" else 2e-3 * max(1.0, float(np.abs(p_ref).max()))
    assert np.abs(s1 - s[-1:]).max() <= tol and np.abs(p1 - p[-1:]).max() <= tol


def test_reference_loop_config_go9_16x128_at_batch_2048():
    """The reference's own shipping configuration (python/main/loop_main_alpha.py:16-30,68-76): Go 9x9, 16 x 128,
    ConvPolicyHead(extra_moves=1), gpu_batch_size 2048 — at the parity default and in f16, one evaluation of 2048 boards;
    64 boards spread over the batch against the oracle, every board against a small-batch evaluation of itself."""
    cfg = sweep_cases.REFERENCE_LOOP
    blob = synth.random_model(cfg["game"], cfg["depth"], cfg["channels"], cfg["head"], seed=31)
    bits, sin = synth.random_boards(cfg["game"], cfg["batch"], seed=32)
    model = capi.Model(blob=blob)
    net = O.OracleNet(blob)
    pick = np.arange(0, cfg["batch"], cfg["batch"] // 64)[:64]
    x = O.encode_input_full(bits[pick], sin[pick], net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(x, threads=16)
    for name, dtype in (("parity", capi.KZ_DTYPE_F32_SPLIT16), ("f16", capi.KZ_DTYPE_F16)):
        eng = capi.Engine(model, 0, cfg["batch"], dtype)
        s, p = eng.eval_packed(bits, sin)
        if name == "f16":
            assert_f16(s[pick], s_ref, "scalars")
            assert_f16(p[pick], p_ref, "policy")
        else:
            assert_f32(s[pick], s_ref, "scalars")
            assert_f32(p[pick], p_ref, "policy")
        # the other 1984 boards: the same engine on 5-board batches cut out of the big one
        for lo in (0, 777, 2043):
            s5, p5 = eng.eval_packed(bits[lo:lo + 5], sin[lo:lo + 5])
            assert np.abs(s5 - s[lo:lo + 5]).max() <= 1e-5 and np.abs(p5 - p[lo:lo + 5]).max() <= 1e-5
