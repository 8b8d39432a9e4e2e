"""N4: the packed position record format (rust/kz-selfplay/src/binary_output.rs:210-289)."""
import os
import sys

import numpy as np
import pytest

from kzero_amd import synth
from kzero_amd.position_file import SCALAR_NAMES, PositionFile, PositionFileWriter, PositionRecord
from tests import oracle_lib as O

REF = "/root/reference/python"


def _games(rng, game="ataxx-7", n_games=3):
    g = synth.GAMES[game]
    size, policy_len = g["size"], g["policy_len"]
    games = []
    for gi in range(n_games):
        length = 3 + gi
        bits, scalars = synth.random_boards(game, length + 1, seed=50 + gi, n_bool=3)
        recs = []
        for pi in range(length + 1):
            terminal = pi == length
            mv = 0 if terminal else int(rng.integers(1, 12))
            idx = rng.permutation(policy_len)[:mv].astype(np.uint32)
            vals = rng.uniform(0.1, 1, size=mv).astype(np.float32)
            vals = vals / vals.sum() if mv else vals
            sc = {n: 0.0 for n in SCALAR_NAMES}
            sc.update(game_id=float(gi), pos_index=float(pi), game_length=float(length), zero_visits=100.0,
                      is_full_search=1.0, is_final_position=float(terminal), is_terminal=float(terminal),
                      available_mv_count=float(mv), played_mv=float(idx[0]) if mv else -1.0, kdl_policy=0.25,
                      final_v=1.0, final_wdl_w=1.0, zero_v=0.5, zero_wdl_w=0.6, zero_wdl_d=0.3, zero_wdl_l=0.1,
                      net_v=0.1, net_wdl_w=0.4, net_wdl_d=0.3, net_wdl_l=0.3, final_moves_left=float(length - pi),
                      zero_moves_left=3.0, net_moves_left=2.0)
            recs.append(PositionRecord(sc, bits[pi], scalars[pi], idx, vals))
        games.append(recs)
    return games, (3, size, size), 1, (policy_len,)


def _write(path, games, shape, n_scalar, policy_shape, game="ataxx-7"):
    w = PositionFileWriter(path, game, shape, n_scalar, policy_shape)
    for g in games:
        w.append_game(g)
    w.finish()


def test_outcome_metadata_is_real_or_nan(tmp_path):
    """root_wdl / hit_move_limit of the metadata are averages over the games as in the reference
    (binary_output.rs:144-145,276-277) when the caller supplies the outcomes, and NaN — not an invented value — when not."""
    import json
    import math
    games, shape, ns, pshape = _games(np.random.default_rng(7))
    path = str(tmp_path / "games_7")
    w = PositionFileWriter(path, "ataxx-7", shape, ns, pshape)
    w.append_game(games[0], root_wdl=(1, 0, 0), hit_move_limit=False)
    w.append_game(games[1], root_wdl=(0, 0, 1), hit_move_limit=False)
    w.append_game(games[2], root_wdl=(0, 1, 0), hit_move_limit=True)
    w.finish()
    meta = json.load(open(path + ".json"))
    assert meta["root_wdl"] == pytest.approx([1 / 3, 1 / 3, 1 / 3]) and meta["hit_move_limit"] == pytest.approx(1 / 3)
    # f32 averages the way serde_json prints them (shortest decimal of the f32), not 6 or 17 significant digits
    assert '"hit_move_limit": 0.33333334' in open(path + ".json").read()
    path2 = str(tmp_path / "games_8")
    _write(path2, games, shape, ns, pshape)
    # strict JSON: serde_json writes an f32 NaN as null, and a bare NaN token would be rejected by it (and by this parser)
    meta2 = json.loads(open(path2 + ".json").read(), parse_constant=lambda c: pytest.fail(f"non-JSON constant {c}"))
    assert meta2["root_wdl"] == [None, None, None] and meta2["hit_move_limit"] is None
    back = PositionFile(path2).meta
    assert all(math.isnan(v) for v in back.root_wdl) and math.isnan(back.hit_move_limit)
    assert len(PositionFile(path2)) == sum(len(g) for g in games)


def test_round_trip(tmp_path):
    games, shape, ns, pshape = _games(np.random.default_rng(1))
    path = str(tmp_path / "games_0")
    _write(path, games, shape, ns, pshape)
    assert not os.path.exists(path + ".json.tmp")
    f = PositionFile(path)
    flat = [r for g in games for r in g]
    assert len(f) == len(flat) and f.meta.game_count == 3
    assert f.game_starts.tolist() == [0, 4, 9]
    assert (f.meta.min_game_length, f.meta.max_game_length) == (3, 5)
    for i, r in enumerate(flat):
        q = f.position(i)
        assert q.scalars == pytest.approx(r.scalars)
        assert np.array_equal(q.bits, r.bits) and np.array_equal(q.input_scalars, r.input_scalars)
        assert np.array_equal(q.policy_indices, r.policy_indices) and np.array_equal(q.policy_values, r.policy_values)
    bits, scalars, moves = f.read_boards([0, 5])
    assert bits.shape == (2, 19) and scalars.shape == (2, 1) and len(moves[1]) == int(flat[5].scalars["available_mv_count"])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's own reader only exists in the build container")
def test_reference_reader_accepts_our_files(tmp_path):
    """Pins the writer against the reference's own consumer of this format: python/lib/data/file.py + position.py."""
    games, shape, ns, pshape = _games(np.random.default_rng(2))
    path = str(tmp_path / "games_7")
    _write(path, games, shape, ns, pshape)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        from lib.data.file import DataFile
        from lib.games import Game
        df = DataFile.open(Game.find("ataxx-7"), path)
        flat = [r for g in games for r in g]
        assert len(df.positions) == len(flat) and len(df.simulations) == 3
        for i, r in enumerate(flat):
            p = df.positions[i]
            assert p.move_index == int(r.scalars["pos_index"]) and p.available_mv_count == len(r.policy_indices)
            unpacked = np.unpackbits(r.bits, bitorder="little")[:3 * 49].reshape(3, 7, 7)
            assert np.array_equal(p.input_bools, unpacked)
            assert np.array_equal(p.input_scalars, r.input_scalars)
            assert np.array_equal(p.policy_indices, r.policy_indices.astype(np.int32))
            assert np.array_equal(p.policy_values, r.policy_values)
            assert p.zero_v == pytest.approx(0.5) and p.is_final == bool(r.scalars["is_final_position"])
        sim = df.simulations[1]
        assert sim.start_file_pi == 4 and sim.move_count == 4
    finally:
        sys.path.remove(REF)


def test_recorded_boards_are_engine_inputs(tmp_path):
    """The board part of a record is the packed engine input: expanding it gives the planes of encode_input_full."""
    games, shape, ns, pshape = _games(np.random.default_rng(3))
    path = str(tmp_path / "games_1")
    _write(path, games, shape, ns, pshape)
    bits, scalars, _ = PositionFile(path).read_boards()
    dense = O.encode_input_full(bits, scalars, 1, 3, 7, 7)
    assert dense.shape == (15, 4, 7, 7)
    assert np.array_equal(dense[:, 0, 0, 0], scalars[:, 0])
    assert np.array_equal(dense[3, 1:].reshape(-1).astype(np.uint8), np.unpackbits(bits[3], bitorder="little")[:147])


# ---- the C++ writer/reader of the host mirror (kzero_amd/csrc/host/position_file.hpp) against the Python one ----
def _fnv(parts):
    h = 1469598103934665603
    for p in parts:
        for b in p:
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _checksum_lines(f: PositionFile):
    out = []
    for i in range(len(f)):
        r = f.position(i)
        parts = [np.array([r.scalars[n] for n in SCALAR_NAMES], dtype="<f4").tobytes(), r.bits.tobytes(),
                 np.asarray(r.input_scalars, dtype="<f4").tobytes(), np.asarray(r.policy_indices, dtype="<u4").tobytes(),
                 np.asarray(r.policy_values, dtype="<f4").tobytes()]
        out.append(f"pos {i} mv {len(r.policy_values)} sum {_fnv(parts):016x}")
    return out


def _tool():
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = os.path.join(repo, "tests", "cpp", "build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "position_file_tool")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", os.path.join(repo, "tests", "cpp", "position_file_tool.cpp"),
                           "-o", exe])
    return exe


def test_cpp_writer_is_read_by_the_python_reader_and_back(tmp_path):
    import subprocess
    exe = _tool()
    # C++ writes, Python reads: every record byte for byte, offsets, game starts, metadata
    path = str(tmp_path / "games_7")
    out = subprocess.run([exe, "write", path, "11", "5"], capture_output=True, text=True, check=True).stdout.split("\n")
    written = [line for line in out if line.startswith("pos ")]
    f = PositionFile(path)
    assert f.meta.game == "chess" and list(f.meta.input_bool_shape) == [13, 8, 8] and f.meta.game_count == 5
    assert not os.path.exists(path + ".json.tmp")
    assert _checksum_lines(f) == written
    starts = f.game_starts.tolist()
    assert starts[0] == 0 and starts == sorted(starts) and len(starts) == 5
    # ... and the C++ reader reads its own file the same way
    dumped = subprocess.run([exe, "dump", path], capture_output=True, text=True, check=True).stdout.split("\n")
    assert [line for line in dumped if line.startswith("pos ")] == written
    assert dumped[0].startswith(f"meta game chess positions {len(written)} games 5 bits_bytes 104 scalars 8 starts 0")
    # Python writes, C++ reads
    games, shape, ns, pshape = _games(np.random.default_rng(3))
    path2 = str(tmp_path / "games_8")
    _write(path2, games, shape, ns, pshape)
    dumped2 = subprocess.run([exe, "dump", path2], capture_output=True, text=True, check=True).stdout.split("\n")
    assert [line for line in dumped2 if line.startswith("pos ")] == _checksum_lines(PositionFile(path2))
    assert dumped2[0].startswith("meta game ataxx-7 positions 15 games 3 bits_bytes 19 scalars 1 starts 0 4 9")
    # a truncated .bin is refused with a message
    with open(path2 + ".bin", "r+b") as fh:
        fh.truncate(os.path.getsize(path2 + ".bin") - 3)
    bad = subprocess.run([exe, "dump", path2], capture_output=True, text=True)
    assert bad.returncode == 1 and "error:" in bad.stderr


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's own reader only exists in the build container")
def test_reference_reader_accepts_the_cpp_writers_files(tmp_path):
    """The reference's DataFile (python/lib/data/file.py) opens what the C++ writer wrote: chess-shaped records."""
    import subprocess
    exe = _tool()
    path = str(tmp_path / "games_9")
    subprocess.run([exe, "write", path, "5", "4"], capture_output=True, text=True, check=True)
    ours = PositionFile(path)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        from lib.data.file import DataFile
        from lib.games import Game
        df = DataFile.open(Game.find("chess"), path)
        assert len(df.positions) == len(ours) and len(df.simulations) == 4
        for i in range(len(ours)):
            r, p = ours.position(i), df.positions[i]
            assert p.available_mv_count == len(r.policy_indices)
            assert np.array_equal(p.input_bools, np.unpackbits(r.bits, bitorder="little")[:13 * 64].reshape(13, 8, 8))
            assert np.array_equal(p.input_scalars, r.input_scalars)
            assert np.array_equal(p.policy_indices, r.policy_indices.astype(np.int32))
            assert np.array_equal(p.policy_values, r.policy_values)
    finally:
        sys.path.remove(REF)
