"""Pins the CPU oracle (oracle/kz_oracle.c) to the golden vectors generated from the reference's own PyTorch
network definition (oracle/gen_golden.py) and to the reference's in-tree known answers."""
import ctypes as C
import os

import numpy as np
import pytest

from kzero_amd.model_file import read_model
from tests import oracle_lib as O

# f32 direct convolution vs PyTorch/oneDNN: different summation order only
TOL = dict(rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", O.GOLDEN_NETS)
@pytest.mark.parametrize("kind", ["planes", "randn"])
def test_forward_matches_reference_pytorch(name, kind):
    net = O.OracleNet(O.load_blob(name))
    x, scalars_ref, policy_ref = O.read_io(name, kind, net.c_in, net.h, net.w, net.policy_len)
    scalars, policy = net.forward(x)
    np.testing.assert_allclose(scalars, scalars_ref, **TOL)
    np.testing.assert_allclose(policy, policy_ref, **TOL)


def test_forward_threads_identical():
    net = O.OracleNet(O.load_blob("ataxx7_2x16"))
    x, _, _ = O.read_io("ataxx7_2x16", "planes", net.c_in, net.h, net.w, net.policy_len)
    s1, p1 = net.forward(x, threads=1)
    s4, p4 = net.forward(x, threads=4)
    assert np.array_equal(s1, s4) and np.array_equal(p1, p4)


@pytest.mark.parametrize("name", ["ataxx7_2x16", "chess_att2x64"])
def test_per_layer_activations(name):
    net = O.OracleNet(O.load_blob(name))
    x, _, _ = O.read_io(name, "planes", net.c_in, net.h, net.w, net.policy_len)
    _, _, acts = net.forward_trace(x)
    _, ref = read_model(open(os.path.join(O.GOLDEN, f"{name}.layers.kzm"), "rb").read())
    assert set(ref) == set(acts)
    for k, v in ref.items():
        np.testing.assert_allclose(acts[k].reshape(v.shape), v, err_msg=k, **TOL)


@pytest.mark.parametrize("name", O.GOLDEN_NETS)
def test_encode_input_full_matches_planes(name):
    """bits+scalars -> dense planes: scalar planes first, then bools (mapping/mod.rs:40-63); the packed fixture is
    np.packbits(bitorder='little'), the inverse of the reference reader (python/lib/data/position.py:95-97)."""
    net = O.OracleNet(O.load_blob(name))
    x, _, _ = O.read_io(name, "planes", net.c_in, net.h, net.w, net.policy_len)
    bits, scalars = O.read_packed(name, net.n_bool, net.n_scalar, net.h, net.w)
    dense = O.encode_input_full(bits, scalars, net.n_scalar, net.n_bool, net.h, net.w)
    assert np.array_equal(dense, x)


def _bitbuffer(capacity):
    n = O.lib().kzo_bits_storage_len(capacity)
    return np.zeros(n, np.uint8), C.c_size_t(0)


def test_bitbuffer_known_answers():
    """rust/kz-core/src/mapping/bit_buffer.rs:112-164"""
    L = O.lib()
    # short (:117-124)
    buf, n = _bitbuffer(8)
    for b in (1, 0, 1):
        assert L.kzo_bits_push(buf.ctypes.data, 8, C.byref(n), b) == 0
    assert buf.tolist() == [0b101]
    # edge_length (:127-135)
    buf, n = _bitbuffer(8)
    for _ in range(8):
        L.kzo_bits_push(buf.ctypes.data, 8, C.byref(n), 1)
    assert buf.tolist() == [0xFF]
    buf, n = _bitbuffer(9)
    for _ in range(9):
        L.kzo_bits_push(buf.ctypes.data, 9, C.byref(n), 1)
    assert buf.tolist() == [0xFF, 0b1]
    # longer (:138-144)
    buf, n = _bitbuffer(16)
    for i in range(16):
        L.kzo_bits_push(buf.ctypes.data, 16, C.byref(n), int(i in (1, 5, 12)))
    assert buf.tolist() == [0b0010_0010, 0b1_0000]
    # overflow (:147-153)
    buf, n = _bitbuffer(32)
    for _ in range(32):
        assert L.kzo_bits_push(buf.ctypes.data, 32, C.byref(n), 0) == 0
    assert L.kzo_bits_push(buf.ctypes.data, 32, C.byref(n), 0) != 0
    # block (:156-163)
    buf, n = _bitbuffer(64)
    assert L.kzo_bits_push_block(buf.ctypes.data, 64, C.byref(n), 0b1_0000_0001) == 0
    assert buf.tolist() == [1, 1, 0, 0, 0, 0, 0, 0] and n.value == 64
    # unaligned block is rejected (:45-51)
    buf, n = _bitbuffer(128)
    L.kzo_bits_push(buf.ctypes.data, 128, C.byref(n), 1)
    assert L.kzo_bits_push_block(buf.ctypes.data, 128, C.byref(n), 1) != 0


def test_decode_output_known_answers():
    """tanh / softmax-of-gathered-logits (network/common.rs:60-86) vs torch-computed answers."""
    meta, t = read_model(open(os.path.join(O.GOLDEN, "decode_kat.kzm"), "rb").read())
    b = meta["batch"]
    moves = [t[f"indices.{i}"].astype(np.int32) for i in range(b)]
    values, pols = O.decode_output(t["scalars"], t["logits"], moves)
    np.testing.assert_allclose(values[:, 0], t["value"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(values[:, 1:4], t["wdl"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(values[:, 4], t["moves_left"])
    for i in range(b):
        assert len(pols[i]) == len(moves[i])
        np.testing.assert_allclose(pols[i], t[f"policy.{i}"], rtol=1e-5, atol=1e-7)
        if len(moves[i]):
            assert abs(pols[i].sum() - 1) < 1e-5


def test_softmax_rejects_non_positive_sum():
    """common.rs:110: assert!(sum > 0.0) — NaN logits trip it."""
    v = np.array([np.nan, 1.0], np.float32)
    assert O.lib().kzo_softmax_in_place(v.ctypes.data, 2) != 0


# ---- chess policy indexing (chess.rs:180-507): what decode_output calls for every available move ----
def _chess_tables():
    g = O.GOLDEN
    rows = [[int(x) for x in line.strip().strip(",").split(",")] for line in open(os.path.join(g, "chess_flat_to_move_input.txt"))]
    t = np.array(rows)
    promo = np.where(t[:, 3] == 1, 1, np.where(t[:, 4] == 1, 2, np.where(t[:, 5] == 1, 3, np.where(t[:, 6] == 1, 4, 0))))
    moves = np.stack([t[:, 0], t[:, 1], promo], axis=1)
    conv = np.loadtxt(os.path.join(g, "chess_flat_to_conv.txt"), dtype=np.int64)
    att = np.loadtxt(os.path.join(g, "chess_flat_to_att.txt"), dtype=np.int64)
    return moves, conv, att


def test_chess_flat_moves_match_the_reference_tables():
    """generate_all_flat_moves_pov against the tables the reference's write_chess_mapping.rs wrote from it
    (python/lib/mapping/chess_flat_to_{move_input,conv,att}.txt, copied as data): all 1880 moves, their conv-policy
    index (ClassifiedPovMove) and their attention index; plus flat_gen's count / no-duplicates check."""
    L = O.lib()
    L.kzo_chess_flat_moves.argtypes = [C.c_void_p]
    out = (C.c_int * (1880 * 3))()
    assert L.kzo_chess_flat_moves(out) == 1880  # tests/mapper/chess/mod.rs:6-17
    got = np.array(out).reshape(1880, 3)
    assert len({tuple(r) for r in got.tolist()}) == 1880
    moves, conv, att = _chess_tables()
    assert (got == moves).all()
    for i, (f, t, p) in enumerate(got.tolist()):
        assert L.kzo_chess_conv_index(1, f, t, p) == conv[i]
        assert L.kzo_chess_att_index(f, t, p) == att[i]
        assert L.kzo_chess_move_to_index(1, f, t, p) == i
        # black: the same POV move, ranks flipped (square_pov)
        fb, tb = (7 - f // 8) * 8 + f % 8, (7 - t // 8) * 8 + t % 8
        assert L.kzo_chess_move_to_index(0, fb, tb, p) == i
        assert L.kzo_chess_conv_index(0, fb, tb, p) == conv[i]
    assert L.kzo_chess_move_to_index(1, 0, 0, 0) == -1
    # the attention gather table of the golden chess model is that same table
    _, tensors = read_model(open(os.path.join(O.GOLDEN, "chess_2x32_att.kzm"), "rb").read())
    assert (tensors["policy_head.FLAT_TO_ATT"].astype(np.int64) == att).all()


def test_chess_conv_index_known_answers_of_the_reference_tests():
    """The (side to move, move) <-> index vectors of rust/kz-core/tests/mapper/chess/pairs.rs:16-356."""
    import json
    L = O.lib()
    rows = json.load(open(os.path.join(O.GOLDEN, "chess_conv_pairs.json")))
    assert len(rows) == 84 and {r["case"] for r in rows} >= {"castles", "en_passant", "black_potential_promotions"}
    for r in rows:
        assert L.kzo_chess_conv_index(int(r["white_to_move"]), r["from"], r["to"], r["promotion"]) == r["conv_index"], r


# ---- property tests of the checker itself (hypothesis): the oracle against independent numpy formulations ----
from hypothesis import given, settings, strategies as st


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 5), st.integers(0, 9), st.integers(1, 9), st.integers(1, 9), st.integers(1, 4), st.integers(0, 2 ** 31))
def test_encode_input_full_matches_numpy_unpackbits(n_scalar, n_bool, h, w, batch, seed):
    """encode_input_full (mapping/mod.rs:40-63) for arbitrary shapes: scalar planes are broadcasts, bool planes are
    np.unpackbits(bitorder="little") of the BitBuffer storage — the formulation of the reference's own reader
    (python/lib/data/position.py:95-97)."""
    if n_scalar + n_bool == 0:
        return
    rng = np.random.default_rng(seed)
    nbytes = max((n_bool * h * w + 7) // 8, 1)
    bits = rng.integers(0, 256, size=(batch, nbytes), dtype=np.uint8)
    scalars = rng.normal(size=(batch, max(n_scalar, 1))).astype(np.float32)[:, :n_scalar]
    got = O.encode_input_full(bits, scalars, n_scalar, n_bool, h, w)
    want = np.empty((batch, n_scalar + n_bool, h, w), np.float32)
    want[:, :n_scalar] = scalars[:, :, None, None]
    planes = np.unpackbits(bits, axis=1, bitorder="little")[:, :n_bool * h * w].reshape(batch, n_bool, h, w)
    want[:, n_scalar:] = planes
    assert np.array_equal(got, want)


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 6), st.integers(2, 60), st.integers(0, 2 ** 31))
def test_decode_output_matches_numpy(batch, policy_len, seed):
    """decode_output (network/common.rs:16-100) against a direct numpy softmax / tanh over random legal-move lists."""
    rng = np.random.default_rng(seed)
    scalars = rng.normal(size=(batch, 5)).astype(np.float32)
    logits = (3 * rng.normal(size=(batch, policy_len))).astype(np.float32)
    moves = [rng.permutation(policy_len)[:rng.integers(0, policy_len + 1)].astype(np.int32) for _ in range(batch)]
    values, probs = O.decode_output(scalars, logits, moves)
    for b in range(batch):
        wdl = np.exp(scalars[b, 1:4] - scalars[b, 1:4].max())
        wdl /= wdl.sum()
        np.testing.assert_allclose(values[b], [np.tanh(scalars[b, 0]), *wdl, scalars[b, 4]], rtol=2e-6, atol=1e-7)
        if len(moves[b]):
            g = logits[b, moves[b]].astype(np.float64)
            e = np.exp(g - g.max())
            np.testing.assert_allclose(probs[b], e / e.sum(), rtol=5e-6, atol=1e-8)
            assert abs(float(probs[b].sum()) - 1.0) < 1e-5
        else:
            assert probs[b].size == 0
