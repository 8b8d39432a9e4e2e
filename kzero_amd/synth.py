"""Synthetic models and boards (no network for checkpoints or datasets).

`random_model` builds a KZMODEL1 container with the tensor names and shapes of the reference's
`PredictionHeads(ResTower(d, C_in, C), ScalarHead(S, C, 4, 32), <policy head>)` state_dict
(python/lib/model/post_act.py:187-211), initialised like PyTorch's defaults
(U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for conv/linear weights and biases) with mildly non-trivial BatchNorm
statistics so that Conv+BN folding is exercised.

`random_boards` builds packed boards: BitBuffer-layout bool planes (LSB-first, rust/kz-core/src/mapping/
bit_buffer.rs:73-75) plus the per-board scalars, with densities that resemble real positions.
"""
import os
from typing import Dict, Tuple

import numpy as np

from .model_file import read_model, write_model

GAMES = {
    # name: (board, scalar planes, bool planes, policy_len fn)
    "chess": dict(size=8, n_scalar=8, n_bool=13, policy_len=1880, p_bool=0.04),      # chess.rs:125-171
    "ataxx-7": dict(size=7, n_scalar=1, n_bool=3, policy_len=17 * 49 + 1, p_bool=0.3),  # ataxx.rs:93-116
    "ataxx-5": dict(size=5, n_scalar=1, n_bool=3, policy_len=17 * 25 + 1, p_bool=0.3),  # (AtaxxBoard sizes 2..8)
    "ataxx-6": dict(size=6, n_scalar=1, n_bool=3, policy_len=17 * 36 + 1, p_bool=0.3),
    "go-19": dict(size=19, n_scalar=6, n_bool=7, policy_len=1 + 361, p_bool=0.25),   # go.rs:46-113 (territory on)
    "go-9": dict(size=9, n_scalar=6, n_bool=7, policy_len=1 + 81, p_bool=0.25),
    # the other games the server dispatches (rust/kz-selfplay/src/server/server.rs:114-185)
    "ttt": dict(size=3, n_scalar=0, n_bool=2, policy_len=9, p_bool=0.3),              # ttt.rs:13-24
    "sttt": dict(size=9, n_scalar=0, n_bool=3, policy_len=81, p_bool=0.3),            # sttt.rs:11-26
    "arimaa-split": dict(size=8, n_scalar=12, n_bool=26, policy_len=1 + 6 + 256, p_bool=0.04),  # arimaa.rs:15-83
}


def game_spec(game: str) -> dict:
    """The table above, plus every other size / mapper the reference's server accepts (rust/kz-selfplay/src/server/
    server.rs:170-200): `ataxx-N` (N 2..8), `go-N`, `go-N-noterr` (GoStdMapper without the territory planes: 4 bool + 6
    scalar = 10 input planes), `chess-hist-L` (ChessHistoryMapper, chess.rs:32-39: 8 + L scalars, 1 + 12 (L + 1) bools)."""
    if game in GAMES:
        return GAMES[game]
    if game.startswith("chess-hist-"):
        length = int(game[len("chess-hist-"):])
        return dict(size=8, n_scalar=7 + (length + 1), n_bool=1 + 12 * (length + 1), policy_len=1880, p_bool=0.04)
    if game.startswith("ataxx-"):
        n = int(game[6:])
        return dict(size=n, n_scalar=1, n_bool=3, policy_len=17 * n * n + 1, p_bool=0.3)
    if game.startswith("go-"):
        noterr = game.endswith("-noterr")
        n = int(game[3:-7] if noterr else game[3:])
        return dict(size=n, n_scalar=6, n_bool=4 if noterr else 7, policy_len=1 + n * n, p_bool=0.25)
    raise KeyError(game)


_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def chess_flat_to_att() -> np.ndarray:
    """The 1880-entry gather table of AttentionPolicyHead (a data table of the reference,
    python/lib/mapping/chess_flat_to_att.txt), taken from the committed golden chess model."""
    _, tensors = read_model(open(os.path.join(_GOLDEN, "chess_2x32_att.kzm"), "rb").read())
    return tensors["policy_head.FLAT_TO_ATT"].astype(np.int64)


def _uniform(rng, shape, fan_in):
    bound = 1.0 / np.sqrt(fan_in)
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


_INIT = "uniform"  # what _conv draws convolution weights from (set by random_model for the duration of one call)


def _conv(rng, t, prefix, cout, cin, k):
    if _INIT == "kaiming_normal":  # nn.init.kaiming_normal_(w, nonlinearity="relu"): N(0, sqrt(2 / fan_in))
        t[prefix + ".weight"] = rng.normal(0.0, np.sqrt(2.0 / (cin * k * k)), size=(cout, cin, k, k)).astype(np.float32)
    else:
        t[prefix + ".weight"] = _uniform(rng, (cout, cin, k, k), cin * k * k)
    t[prefix + ".bias"] = _uniform(rng, (cout,), cin * k * k)


def _linear(rng, t, prefix, out, inp):
    t[prefix + ".weight"] = _uniform(rng, (out, inp), inp)
    t[prefix + ".bias"] = _uniform(rng, (out,), inp)


def _bn(rng, t, prefix, c):
    t[prefix + ".weight"] = rng.uniform(0.8, 1.2, size=c).astype(np.float32)
    t[prefix + ".bias"] = rng.normal(0.0, 0.1, size=c).astype(np.float32)
    t[prefix + ".running_mean"] = rng.normal(0.0, 0.1, size=c).astype(np.float32)
    t[prefix + ".running_var"] = rng.uniform(0.8, 1.25, size=c).astype(np.float32)


def random_model(game: str, depth: int, channels: int, head: str, seed: int = 0, query_channels: int = None,
                 n_bool: int = None, scalar_hidden_size: int = 32, block_gain: float = 1.0,
                 dense_hidden_channels: int = None, dense_hidden_size: int = None, scalar_hidden_channels: int = 4,
                 final_affine: bool = True, init: str = "uniform", arimaa_hidden_channels: int = 2,
                 arimaa_hidden_size: int = 32, attention: tuple = None, dense_network: bool = None) -> bytes:
    """`block_gain` > 1 scales every block's second BatchNorm weight: the residual stream then grows from block to block
    the way a trained network's does (a random-init tower keeps it within a few tens).
    `init`: "uniform" = PyTorch's default for Conv2d / Linear (U(+-1/sqrt(fan_in))); "kaiming_normal" = convolution weights
    from N(0, sqrt(2 / fan_in)) — a bell-shaped weight distribution with 2.4 x the standard deviation (the matrix cores'
    power draw, and with it the clock they sustain, depends on the data: DESIGN.md, bench.py `weights`).
    `attention` = (heads, d_k, d_v, d_ff): AttentionTower(board_size, input_channels, depth, channels, heads, d_k, d_v, d_ff)
    (python/lib/model/attention.py:8-30) in place of the ResTower, initialised as the reference does (:84-95: Xavier-normal,
    gain (8 depth)^(-1/4) for v / project_out / ff, gain 1 for q / k; embedding N(0, 1); expand nn.Linear's default).
    `dense_network` = res (True / False), with head "none": DenseNetwork(game, depth, size = channels, res)
    (python/lib/model/simple.py:7-52) — the whole network, no tower and no heads."""
    global _INIT
    if init not in ("uniform", "kaiming_normal"):
        raise ValueError(f"unknown init '{init}'")
    _INIT = init
    try:
        return _random_model(game, depth, channels, head, seed, query_channels, n_bool, scalar_hidden_size, block_gain,
                             dense_hidden_channels, dense_hidden_size, scalar_hidden_channels, final_affine,
                             arimaa_hidden_channels, arimaa_hidden_size, attention, dense_network)
    finally:
        _INIT = "uniform"


def _random_model(game, depth, channels, head, seed, query_channels, n_bool, scalar_hidden_size, block_gain,
                  dense_hidden_channels, dense_hidden_size, scalar_hidden_channels, final_affine,
                  arimaa_hidden_channels, arimaa_hidden_size, attention=None, dense_network=None) -> bytes:
    g = game_spec(game)
    size, n_scalar = g["size"], g["n_scalar"]
    n_bool = g["n_bool"] if n_bool is None else n_bool
    hw, C = size * size, channels
    rng = np.random.default_rng(seed)
    t: Dict[str, np.ndarray] = {}
    meta = {
        "game": game, "board_h": size, "board_w": size,
        "input_scalar_channels": n_scalar, "input_bool_channels": n_bool,
        "tower_depth": depth, "tower_channels": C, "tower_final_affine": 1 if final_affine else 0,
        "scalar_hidden_channels": scalar_hidden_channels, "scalar_hidden_size": scalar_hidden_size,
        "policy_kind": head, "policy_len": g["policy_len"], "bn_eps": 1e-5,
    }
    if dense_network is not None:
        if head != "none":
            raise ValueError("a DenseNetwork has no heads: head='none'")
        meta.update({"tower_kind": "dense_network", "dn_res": 1 if dense_network else 0})
        for k in ("tower_final_affine", "scalar_hidden_channels", "scalar_hidden_size"):
            del meta[k]
        _linear(rng, t, "seq.1", C, (n_scalar + n_bool) * hw)
        if game == "chess":  # raw counters (chess.rs:153-154): planes 6 and 7 of the channel-major flatten
            t["seq.1.weight"][:, 6 * hw:7 * hw] *= 0.5
            t["seq.1.weight"][:, 7 * hw:8 * hw] *= 0.01
        for i in range(depth):
            _bn(rng, t, f"seq.{2 + i}.seq.0", C)
            _linear(rng, t, f"seq.{2 + i}.seq.2", C, C)
            _bn(rng, t, f"seq.{2 + i}.seq.3", C)
            _linear(rng, t, f"seq.{2 + i}.seq.5", C, C)
        _bn(rng, t, f"seq.{2 + depth}", C)
        _linear(rng, t, f"seq.{4 + depth}", 5 + g["policy_len"], C)
        return write_model(meta, t)
    if attention is not None:
        _attention_tower(rng, t, meta, game, hw, n_scalar + n_bool, depth, C, *attention)
    else:
        _res_tower(rng, t, game, n_scalar, n_bool, depth, C, block_gain, final_affine)
    _heads(rng, t, meta, g, head, hw, C, query_channels, scalar_hidden_channels, scalar_hidden_size, dense_hidden_channels,
           dense_hidden_size, arimaa_hidden_channels, arimaa_hidden_size)
    return write_model(meta, t)


def _xavier_normal(rng, rows, cols, gain):
    return rng.normal(0.0, gain * np.sqrt(2.0 / (rows + cols)), size=(rows, cols)).astype(np.float32)


def _attention_tower(rng, t, meta, game, hw, c_in, depth, C, heads, d_k, d_v, d_ff):
    meta.update({"tower_kind": "attention", "att_heads": heads, "att_d_k": d_k, "att_d_v": d_v, "att_d_ff": d_ff,
                 "att_alpha": float((2 * depth) ** 0.25), "ln_eps": 1e-5})
    del meta["tower_final_affine"]
    beta = (8 * depth) ** -0.25
    t["common.expand.weight"] = _uniform(rng, (C, c_in), c_in)
    if game == "chess":  # raw counters (chess.rs:153-154), as for the ResTower's stem below
        t["common.expand.weight"][:, 6] *= 0.5
        t["common.expand.weight"][:, 7] *= 0.01
    t["common.embedding"] = rng.normal(0.0, 1.0, size=(hw, C)).astype(np.float32)
    for i in range(depth):
        p = f"common.encoders.{i}."
        # (the reference initialises the q / k / v row blocks of project_qkv.weight separately, attention.py:87-95)
        dkt = heads * d_k
        w = np.concatenate([_xavier_normal(rng, dkt, C, 1.0), _xavier_normal(rng, dkt, C, 1.0),
                            _xavier_normal(rng, heads * (d_k + d_v) - dkt, C, beta)])
        t[p + "project_qkv.weight"] = w
        t[p + "project_out.weight"] = _xavier_normal(rng, C, heads * d_v, beta)
        t[p + "ff.0.weight"] = _xavier_normal(rng, d_ff, C, beta)
        t[p + "ff.2.weight"] = _xavier_normal(rng, C, d_ff, beta)


def _res_tower(rng, t, game, n_scalar, n_bool, depth, C, block_gain, final_affine):
    _conv(rng, t, "common.tower.0", C, n_scalar + n_bool, 3)
    if game == "chess":
        # the mapper feeds raw counters (repetitions 0..2, 50-move counter 0..99; chess.rs:153-154); a trained stem
        # scales them down, an untrained one would let them dominate every activation
        t["common.tower.0.weight"][:, 6] *= 0.5
        t["common.tower.0.weight"][:, 7] *= 0.01
    elif game.startswith("chess-hist-"):
        t["common.tower.0.weight"][:, 6] *= 0.01  # the 50-move counter (chess.rs:54)
        t["common.tower.0.weight"][:, 7:n_scalar] *= 0.5  # 1 + repetitions per board (chess.rs:86-87)
    elif game == "arimaa-split":
        t["common.tower.0.weight"][:, 10] *= 0.25  # history_len, move_number: raw counters (arimaa.rs:81-82)
        t["common.tower.0.weight"][:, 11] *= 0.02
    for i in range(1, depth + 1):
        _conv(rng, t, f"common.tower.{i}.seq.0", C, C, 3)
        _bn(rng, t, f"common.tower.{i}.seq.1", C)
        _conv(rng, t, f"common.tower.{i}.seq.3", C, C, 3)
        _bn(rng, t, f"common.tower.{i}.seq.4", C)
        if block_gain != 1.0:
            t[f"common.tower.{i}.seq.4.weight"] *= np.float32(block_gain)
    _bn(rng, t, f"common.tower.{depth + 1}", C)
    if not final_affine:  # ResTower(..., final_affine=False) (post_act.py:207; the MuZero towers of loop_main_mu.py:78): no weight / bias
        del t[f"common.tower.{depth + 1}.weight"], t[f"common.tower.{depth + 1}.bias"]


def _heads(rng, t, meta, g, head, hw, C, query_channels, scalar_hidden_channels, scalar_hidden_size, dense_hidden_channels,
           dense_hidden_size, arimaa_hidden_channels, arimaa_hidden_size):
    _conv(rng, t, "scalar_head.seq.0", scalar_hidden_channels, C, 1)
    _linear(rng, t, "scalar_head.seq.3", scalar_hidden_size, scalar_hidden_channels * hw)
    _linear(rng, t, "scalar_head.seq.5", 5, scalar_hidden_size)
    if head == "ataxx_conv":
        meta["policy_conv_channels"] = 17
        _conv(rng, t, "policy_head.seq.0", C, C, 1)
        _conv(rng, t, "policy_head.seq.2", 17, C, 1)
    elif head == "conv":
        meta["policy_conv_channels"] = 1
        meta["policy_extra_moves"] = 1
        _conv(rng, t, "policy_head.seq.0", C, C, 1)
        _conv(rng, t, "policy_head.seq.2", 1, C, 1)
        _conv(rng, t, "policy_head.seq_extra.0", 1, C, 1)
        _linear(rng, t, "policy_head.seq_extra.2", 1, hw)
    elif head == "attention":
        q = query_channels or C  # supervised_main_alpha.py:76: AttentionPolicyHead(game, channels, channels)
        meta["policy_query_channels"] = q
        _conv(rng, t, "policy_head.conv_bulk", 2 * q, C, 1)
        _conv(rng, t, "policy_head.conv_under", 3 * q, C, 1)
        t["policy_head.FLAT_TO_ATT"] = chess_flat_to_att()
    elif head == "dense":
        # DensePolicyHead(game, channels, hidden_channels, hidden_size) (post_act.py:26-51): nn.Sequential indices shift
        # with the optional layers — [Conv2d, ReLU,] Flatten, [Linear, ReLU,] Linear
        idx, ch = 0, C
        if dense_hidden_channels:
            meta["policy_dense_hidden_channels"] = dense_hidden_channels
            _conv(rng, t, "policy_head.seq.0", dense_hidden_channels, C, 1)
            ch, idx = dense_hidden_channels, 2
        idx += 1
        size_in = ch * hw
        if dense_hidden_size:
            meta["policy_dense_hidden_size"] = dense_hidden_size
            _linear(rng, t, f"policy_head.seq.{idx}", dense_hidden_size, size_in)
            size_in, idx = dense_hidden_size, idx + 2
        _linear(rng, t, f"policy_head.seq.{idx}", g["policy_len"], size_in)
    elif head == "arimaa":
        # ArimaaPolicyHead(game, channels, hidden_channels, hidden_size) (post_act.py:144-173)
        if g["policy_len"] != 7 + 4 * hw:
            raise ValueError("the arimaa head needs a game whose policy is 1 + 6 + 4 * squares (arimaa-split)")
        meta["policy_arimaa_hidden_channels"] = arimaa_hidden_channels
        meta["policy_arimaa_hidden_size"] = arimaa_hidden_size
        _conv(rng, t, "policy_head.bulk.0", C, C, 1)
        _conv(rng, t, "policy_head.bulk.2", 4, C, 1)
        _conv(rng, t, "policy_head.scalar.0", arimaa_hidden_channels, C, 1)
        _linear(rng, t, "policy_head.scalar.3", arimaa_hidden_size, arimaa_hidden_channels * hw)
        _linear(rng, t, "policy_head.scalar.5", 7, arimaa_hidden_size)
    else:
        raise ValueError(f"unsupported synthetic head '{head}'")


def random_boards(game: str, batch: int, seed: int = 0, n_bool: int = None) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (bits [batch, ceil(n_bool*hw/8)] u8, scalars [batch, n_scalar] f32)."""
    g = game_spec(game)
    size, n_scalar = g["size"], g["n_scalar"]
    n_bool = g["n_bool"] if n_bool is None else n_bool
    rng = np.random.default_rng(seed)
    bools = rng.uniform(size=(batch, n_bool * size * size)) < g["p_bool"]
    bits = np.packbits(bools.astype(np.uint8), axis=1, bitorder="little")
    if game == "chess":
        # [pov==White, pov==Black, K/Q castle us, K/Q castle them, repetitions, 50-move counter] (chess.rs:136-155)
        white = rng.integers(0, 2, size=batch)
        scalars = np.stack([white, 1 - white, *(rng.integers(0, 2, size=batch) for _ in range(4)),
                            rng.integers(0, 3, size=batch), rng.integers(0, 100, size=batch)], axis=1)
    elif game.startswith("chess-hist-"):
        # [pov==White, pov==Black, castling x4, 50-move counter, then 1 + repetitions per board] (chess.rs:41-75)
        white = rng.integers(0, 2, size=batch)
        scalars = np.stack([white, 1 - white, *(rng.integers(0, 2, size=batch) for _ in range(4)),
                            rng.integers(0, 100, size=batch),
                            *(1 + rng.integers(0, 3, size=batch) for _ in range(n_scalar - 7))], axis=1)
    elif game.startswith("ataxx"):
        scalars = rng.uniform(0, 1, size=(batch, 1))  # moves_since_last_copy / MAX (ataxx.rs:107-109)
    elif n_scalar == 0:
        scalars = np.zeros((batch, 0))  # ttt / sttt: no scalar planes (ttt.rs:17-19)
    elif game == "arimaa-split":
        # [place, play, pull, push, one-hot steps_taken x4, next == A, next == B, history_len, move_number] (arimaa.rs:63-82)
        play = rng.integers(0, 2, size=batch)
        steps = np.eye(4)[rng.integers(0, 4, size=batch)]
        gold = rng.integers(0, 2, size=batch)
        scalars = np.concatenate([np.stack([1 - play, play, np.zeros(batch), np.zeros(batch)], axis=1), steps,
                                  np.stack([gold, 1 - gold, rng.integers(0, 4, size=batch), rng.integers(0, 60, size=batch)], axis=1)], axis=1)
    else:
        black = rng.integers(0, 2, size=batch)
        scalars = np.stack([black, 1 - black, np.zeros(batch), np.zeros(batch), np.full(batch, 7.5 / 15.0),
                            np.zeros(batch)], axis=1)  # go.rs:105-112
    return np.ascontiguousarray(bits), np.ascontiguousarray(scalars.astype(np.float32))
