"""Packed position records: `games_N.{bin,off,json}` as written by the self-play server
(rust/kz-selfplay/src/binary_output.rs:88-373) and read by the trainer (python/lib/data/file.py:68-135,
python/lib/data/position.py:34-104).

Per position, appended to .bin (binary_output.rs:210-256): 26 f32 scalars in `SCALAR_NAMES` order (:321-349) |
ceil(input_bool_len/8) bytes of BitBuffer storage | input_scalar_count f32 | available_mv_count u32 policy indices |
available_mv_count f32 policy values.  .off: one u64-LE byte offset per position, then one u64 start-position index per
game (:239,281).  .json: metadata, written as .json.tmp and renamed (:287-289).

The board part of a record is exactly the packed input of `kz_engine_eval_packed` (same BitBuffer layout, same scalars),
so recorded self-play positions can be replayed through the engine (`read_boards`).
"""
import json
import os
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

SCALAR_NAMES = [  # binary_output.rs:321-349
    "game_id", "pos_index", "game_length", "zero_visits", "is_full_search", "is_final_position", "is_terminal",
    "hit_move_limit", "available_mv_count", "played_mv", "kdl_policy",
    "final_v", "final_wdl_w", "final_wdl_d", "final_wdl_l", "final_moves_left",
    "zero_v", "zero_wdl_w", "zero_wdl_d", "zero_wdl_l", "zero_moves_left",
    "net_v", "net_wdl_w", "net_wdl_d", "net_wdl_l", "net_moves_left",
]


def _f32_json(v):
    """An f32 the way serde_json writes it (binary_output.rs:276-277 are f32): the shortest decimal that reads back as the
    same f32, and null for NaN — strict JSON, byte-identical to the C++ writer (host/position_file.hpp)."""
    v = np.float32(v)
    return None if np.isnan(v) else float(str(v))


@dataclass
class PositionRecord:
    scalars: dict                      # name -> float, all 26
    bits: np.ndarray                   # u8 [ceil(bool_len/8)], BitBuffer storage
    input_scalars: np.ndarray          # f32 [input_scalar_count]
    policy_indices: np.ndarray         # u32 [available_mv_count]
    policy_values: np.ndarray          # f32 [available_mv_count]


@dataclass
class PositionFileMeta:
    game: str
    input_bool_shape: Sequence[int]
    input_scalar_count: int
    policy_shape: Sequence[int]
    game_count: int = 0
    position_count: int = 0
    max_game_length: int = -1
    min_game_length: int = -1
    root_wdl: Sequence[float] = (float("nan"),) * 3
    hit_move_limit: float = float("nan")
    scalar_names: List[str] = field(default_factory=lambda: list(SCALAR_NAMES))

    @property
    def bits_bytes(self) -> int:
        return (int(np.prod(self.input_bool_shape)) + 7) // 8


class PositionFileWriter:
    """BinaryOutput (binary_output.rs:88-289) without the game-playing part: positions are appended game by game."""

    def __init__(self, path: str, game: str, input_bool_shape, input_scalar_count: int, policy_shape):
        assert os.path.splitext(path)[1] == "", "path must not have an extension"  # :91-95
        self.path = path
        self.meta = PositionFileMeta(game, list(input_bool_shape), input_scalar_count, list(policy_shape))
        self._bin = open(path + ".bin", "wb")
        self._off = open(path + ".off", "wb")
        self._next_offset = 0
        self._game_starts: List[int] = []
        self._lengths: List[int] = []
        self._finished = False
        self._root_wdl_sum = np.zeros(3, np.float64)  # total_root_wdl, :58,144
        self._hit_move_limit_count = 0.0              # hit_move_limit_count, :59,145
        self._outcomes_known = True

    def append_game(self, records: Sequence[PositionRecord], root_wdl=None, hit_move_limit=None):
        """All positions of one game, the terminal position last (includes_terminal_positions = true).
        root_wdl: the game's outcome as (win, draw, loss) from the point of view of the player to move in the start
        position (binary_output.rs:144); hit_move_limit: the game stopped without an outcome (:145).  finish() writes
        their averages over the games like the reference (:276-277); when a game comes without them the metadata says NaN
        instead of inventing a value."""
        if root_wdl is None or hit_move_limit is None:
            self._outcomes_known = False
        else:
            self._root_wdl_sum += np.asarray(root_wdl, np.float64)
            self._hit_move_limit_count += float(bool(hit_move_limit))
        self._game_starts.append(self.meta.position_count)
        self._lengths.append(len(records) - 1)
        for r in records:
            self._append_position(r)
        self.meta.game_count += 1

    def _append_position(self, r: PositionRecord):  # :210-256
        bits = np.ascontiguousarray(r.bits, dtype=np.uint8)
        assert bits.size == self.meta.bits_bytes
        assert len(r.input_scalars) == self.meta.input_scalar_count
        assert len(r.policy_indices) == len(r.policy_values)
        if len(r.policy_values):
            s = float(np.sum(r.policy_values))
            assert np.isnan(s) or abs(1.0 - s) < 0.001  # assert_normalized_or_nan, :317-319
        self._off.write(struct.pack("<Q", self._next_offset))
        parts = [
            np.array([r.scalars[n] for n in SCALAR_NAMES], dtype="<f4").tobytes(),
            bits.tobytes(),
            np.asarray(r.input_scalars, dtype="<f4").tobytes(),
            np.asarray(r.policy_indices, dtype="<u4").tobytes(),
            np.asarray(r.policy_values, dtype="<f4").tobytes(),
        ]
        for p in parts:
            self._bin.write(p)
            self._next_offset += len(p)
        self.meta.position_count += 1

    def finish(self):  # :258-290
        assert not self._finished, "This output is already finished"
        self._finished = True
        if self._lengths:
            self.meta.max_game_length, self.meta.min_game_length = max(self._lengths), min(self._lengths)
        if self._outcomes_known and self.meta.game_count:
            self.meta.root_wdl = [float(v) for v in self._root_wdl_sum / self.meta.game_count]
            self.meta.hit_move_limit = self._hit_move_limit_count / self.meta.game_count
        meta = {
            "game": self.meta.game, "input_bool_shape": list(self.meta.input_bool_shape),
            "input_scalar_count": self.meta.input_scalar_count, "policy_shape": list(self.meta.policy_shape),
            "game_count": self.meta.game_count, "position_count": self.meta.position_count,
            "includes_terminal_positions": True, "includes_game_start_indices": True,
            "max_game_length": self.meta.max_game_length, "min_game_length": self.meta.min_game_length,
            "root_wdl": [_f32_json(v) for v in self.meta.root_wdl], "hit_move_limit": _f32_json(self.meta.hit_move_limit),
            "scalar_names": list(SCALAR_NAMES),
        }
        self._off.write(np.asarray(self._game_starts, dtype="<u8").tobytes())
        self._bin.close()
        self._off.close()
        with open(self.path + ".json.tmp", "w") as f:
            json.dump(meta, f, indent=2)
        os.rename(self.path + ".json.tmp", self.path + ".json")  # atomic publish, :287-289


class PositionFile:
    """Reader (python/lib/data/file.py:68-135 + position.py:34-104)."""

    def __init__(self, path: str):
        with open(path + ".json") as f:
            meta = json.load(f)
        self.meta = PositionFileMeta(
            game=meta["game"], input_bool_shape=meta["input_bool_shape"], input_scalar_count=meta["input_scalar_count"],
            policy_shape=meta["policy_shape"], game_count=meta["game_count"], position_count=meta["position_count"],
            max_game_length=meta["max_game_length"], min_game_length=meta["min_game_length"],
            root_wdl=[float("nan") if v is None else v for v in (meta.get("root_wdl") or [None] * 3)],  # null = f32 NaN (serde_json)
            hit_move_limit=float("nan") if meta.get("hit_move_limit") is None else meta["hit_move_limit"],
            scalar_names=meta["scalar_names"])
        self._bin = np.fromfile(path + ".bin", dtype=np.uint8)
        off = np.fromfile(path + ".off", dtype="<u8")
        n = self.meta.position_count
        has_starts = meta.get("includes_game_start_indices", False)
        assert len(off) == n + (self.meta.game_count if has_starts else 0), "Mismatch in offset size"  # file.py:97-102
        self.offsets = off[:n]
        self.game_starts = off[n:] if has_starts else None

    def __len__(self):
        return self.meta.position_count

    def position(self, pi: int) -> PositionRecord:
        start = int(self.offsets[pi])
        end = int(self.offsets[pi + 1]) if pi + 1 < len(self.offsets) else len(self._bin)  # file.py:116-124
        data = self._bin[start:end].tobytes()
        names = self.meta.scalar_names
        pos = 4 * len(names)
        scalars = dict(zip(names, np.frombuffer(data[:pos], dtype="<f4").tolist()))
        nb = self.meta.bits_bytes
        bits = np.frombuffer(data[pos:pos + nb], dtype=np.uint8)
        pos += nb
        ns = self.meta.input_scalar_count
        input_scalars = np.frombuffer(data[pos:pos + 4 * ns], dtype="<f4")
        pos += 4 * ns
        mv = int(scalars["available_mv_count"])
        indices = np.frombuffer(data[pos:pos + 4 * mv], dtype="<u4")
        pos += 4 * mv
        values = np.frombuffer(data[pos:pos + 4 * mv], dtype="<f4")
        pos += 4 * mv
        assert pos == len(data), "Leftover bytes in position record"  # Taker.finish(), position.py:104
        return PositionRecord(scalars, bits, input_scalars, indices, values)

    def read_boards(self, indices: Optional[Sequence[int]] = None) -> Tuple[np.ndarray, np.ndarray, List[np.ndarray]]:
        """Packed engine inputs of the given positions: (bits [n, bits_bytes], scalars [n, S], move index lists)."""
        indices = range(len(self)) if indices is None else indices
        recs = [self.position(i) for i in indices]
        bits = np.stack([r.bits for r in recs]) if recs else np.zeros((0, self.meta.bits_bytes), np.uint8)
        scalars = np.stack([r.input_scalars for r in recs]) if recs else np.zeros((0, self.meta.input_scalar_count),
                                                                                   np.float32)
        return np.ascontiguousarray(bits), np.ascontiguousarray(scalars, dtype=np.float32), \
            [r.policy_indices.astype(np.int32) for r in recs]
