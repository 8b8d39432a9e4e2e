#!/usr/bin/env python3
"""Golden-vector generator.  TEST INFRASTRUCTURE — runs ONLY in the build container.

Imports the reference's own PyTorch definition of the network
(/root/reference/python/lib/model/post_act.py:10-239, lib/games.py) and writes
small committed fixtures under tests/golden/:

  <net>.kzm               model container (kzero_amd/model_file.py) holding the
                          unfolded state_dict of PredictionHeads(ResTower, ScalarHead, head)
  <net>.<kind>.io.bin     the reference's own check format (python/lib/save_onnx.py:95-102):
                          u8 batch | input f32 [B,C,H,W] | scalars f32 [B,5] | policy f32 [B,P]
                          kind = "planes" (0/1 bool planes + broadcast scalars, what
                          encode_input_full produces, rust/kz-core/src/mapping/mod.rs:40-63)
                          or "randn" (what save_onnx itself feeds, save_onnx.py:86-89)
  <net>.planes.packed.bin u8 batch | bits u8 [B, ceil(nbool/8)] (np.packbits little, the inverse of
                          python/lib/data/position.py:95-97) | scalars f32 [B,S]
  <net>.layers.kzm        per-layer intermediate activations for one small net
  decode_kat.kzm          tanh / softmax known answers for decode_output
                          (rust/kz-core/src/network/common.rs:60-86), computed with torch

Nothing from /root/reference is copied: only tensors (data) leave this script.
The reference cannot travel to the GPU box, the fixtures do.
"""
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference/python"
sys.path.insert(0, REF)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch
from torch import nn

from lib.games import Game  # noqa: E402  (reference)
from lib.model.post_act import (  # noqa: E402  (reference)
    PredictionHeads, ResTower, ScalarHead, AtaxxConvPolicyHead, AttentionPolicyHead, ConvPolicyHead,
    DensePolicyHead, ResBlock, ArimaaPolicyHead,
)
from lib.model.attention import AttentionTower  # noqa: E402  (reference)
from lib.model.simple import DenseNetwork  # noqa: E402  (reference)

from kzero_amd.model_file import write_model  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def build(game_name, depth, channels, head_kind, input_bool_channels=None, attention=None, **head_args):
    game = Game.find(game_name)
    n_scalar = game.input_scalar_channels
    n_bool = game.input_bool_channels if input_bool_channels is None else input_bool_channels
    c_in = n_scalar + n_bool
    scalar_head = ScalarHead(game.board_size, channels, 4, 32)
    meta = {
        "game": game.name,
        "board_h": game.board_size, "board_w": game.board_size,
        "input_scalar_channels": n_scalar, "input_bool_channels": n_bool,
        "tower_depth": depth, "tower_channels": channels, "tower_final_affine": 1,
        "scalar_hidden_channels": 4, "scalar_hidden_size": 32,
        "policy_kind": head_kind, "policy_len": game.policy_size,
        "bn_eps": 1e-5,
    }
    if attention is None:
        tower = ResTower(depth, c_in, channels)
    else:
        # the tower python/main/supervised_main_alpha.py:72 trains: AttentionTower(board_size, input_channels, depth,
        # d_model, heads, d_k, d_v, d_ff, dropout) (python/lib/model/attention.py:8-45); channels = d_model
        heads, d_k, d_v, d_ff = attention
        tower = AttentionTower(game.board_size, c_in, depth, channels, heads, d_k, d_v, d_ff, 0.1)
        meta.update({"tower_kind": "attention", "att_heads": heads, "att_d_k": d_k, "att_d_v": d_v, "att_d_ff": d_ff,
                     "att_alpha": float((2 * depth) ** (1 / 4)), "ln_eps": 1e-5})
        del meta["tower_final_affine"]
    if head_kind == "ataxx_conv":
        head = AtaxxConvPolicyHead(game, channels)
        meta["policy_conv_channels"] = game.policy_conv_channels
    elif head_kind == "conv":
        head = ConvPolicyHead(game, channels, **head_args)
        meta["policy_conv_channels"] = game.policy_conv_channels
        meta["policy_extra_moves"] = head_args.get("extra_moves", 0)
    elif head_kind == "attention":
        head = AttentionPolicyHead(game, channels, **head_args)
        meta["policy_query_channels"] = head_args["query_channels"]
    elif head_kind == "dense":
        head = DensePolicyHead(game, channels, **head_args)
        meta["policy_dense_hidden_channels"] = head_args["hidden_channels"] or 0
        meta["policy_dense_hidden_size"] = head_args["hidden_size"] or 0
    elif head_kind == "arimaa":
        head = ArimaaPolicyHead(game, channels, **head_args)
        meta["policy_arimaa_hidden_channels"] = head_args["hidden_channels"]
        meta["policy_arimaa_hidden_size"] = head_args["hidden_size"]
    else:
        raise ValueError(head_kind)
    net = PredictionHeads(tower, scalar_head, head)
    return game, net, meta, (n_scalar, n_bool)


def make_planes_input(rng, batch, n_scalar, n_bool, size, p_bool):
    """What encode_input_full produces: scalar planes first (each broadcast), then 0/1 planes."""
    scalars = rng.uniform(0.0, 1.0, size=(batch, n_scalar)).astype(np.float32)
    # make some scalars exact small integers / flags like the real mappers do
    scalars[:, ::2] = rng.integers(0, 3, size=scalars[:, ::2].shape).astype(np.float32)
    bools = (rng.uniform(size=(batch, n_bool, size, size)) < p_bool)
    dense = np.concatenate([
        np.broadcast_to(scalars[:, :, None, None], (batch, n_scalar, size, size)),
        bools.astype(np.float32),
    ], axis=1).astype(np.float32)
    flat = bools.reshape(batch, -1).astype(np.uint8)
    bits = np.packbits(flat, axis=1, bitorder="little")
    return np.ascontiguousarray(dense), bits, scalars


def randomize_bn(net, gen):
    """Give every BN non-trivial affine parameters and running statistics
    (cf. python/main/write_test_networks.py:24-37 which trains for the same reason)."""
    for m in net.modules():
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            with torch.no_grad():
                m.weight.copy_(torch.empty_like(m.weight).uniform_(0.6, 1.4, generator=gen))
                m.bias.copy_(torch.empty_like(m.bias).normal_(0.0, 0.2, generator=gen))
                m.running_mean.copy_(torch.empty_like(m.running_mean).normal_(0.0, 0.3, generator=gen))
                m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 2.0, generator=gen))


def write_io(path, batch, inputs, outputs):
    with open(path, "wb") as f:
        f.write(batch.to_bytes(1, byteorder="little", signed=False))
        f.write(np.ascontiguousarray(inputs, dtype=np.float32).tobytes())
        for o in outputs:
            f.write(np.ascontiguousarray(o.detach().numpy(), dtype=np.float32).tobytes())


def export_onnx(net, path, shape):
    """The reference's own export call (python/lib/save_onnx.py:107-119: opset 10, names input/scalars/policy, dynamic
    batch axis).  The `onnx` package is absent here; torch only needs it for an onnxscript post-pass, stubbed out."""
    import warnings
    import torch.onnx._internal.torchscript_exporter.onnx_proto_utils as opu
    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    batch_axis = {0: "batch_size"}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model=net, args=(torch.randn(2, *shape),), f=path, input_names=["input"],
                          output_names=["scalars", "policy"],
                          dynamic_axes={k: batch_axis for k in ["input", "scalars", "policy"]}, opset_version=10,
                          dynamo=False)


def build_dense_network(game_name, depth, size, res):
    """DenseNetwork(game, depth, size, res) (python/lib/model/simple.py:7-33): no tower, no heads — Flatten, Linear, `depth`
    DenseBlocks (BatchNorm1d, ReLU, Linear, BatchNorm1d, ReLU, Linear, residual optional), BatchNorm1d, ReLU, Linear to
    5 + policy_size; the reference's own test networks (python/main/write_test_networks.py:14-18)."""
    game = Game.find(game_name)
    net = DenseNetwork(game, depth, size, res)
    meta = {
        "game": game.name, "board_h": game.board_size, "board_w": game.board_size,
        "input_scalar_channels": game.input_scalar_channels, "input_bool_channels": game.input_bool_channels,
        "tower_kind": "dense_network", "tower_depth": depth, "tower_channels": size, "dn_res": 1 if res else 0,
        "policy_kind": "none", "policy_len": game.policy_size, "bn_eps": 1e-5,
    }
    return game, net, meta, (game.input_scalar_channels, game.input_bool_channels)


def gen_net(name, seed, batch, p_bool, layers=False, onnx=False, dense_network=None, **kw):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    rng = np.random.default_rng(seed + 2)
    game, net, meta, (n_scalar, n_bool) = build_dense_network(*dense_network) if dense_network else build(**kw)
    randomize_bn(net, gen)
    net.eval()  # save_onnx.py:82

    tensors = {}
    for k, v in net.state_dict().items():
        if k.endswith("num_batches_tracked"):
            continue
        a = v.detach().numpy()
        tensors[k] = a.astype(np.int64) if a.dtype == np.int64 else a.astype(np.float32)
    with open(os.path.join(OUT, f"{name}.kzm"), "wb") as f:
        f.write(write_model(meta, tensors))

    size = game.board_size
    if onnx:
        path = os.path.join(OUT, f"{name}.onnx")
        if os.path.exists(path):
            os.remove(path)
        export_onnx(net, path, (n_scalar + n_bool, size, size))
    dense, bits, scalars_in = make_planes_input(rng, batch, n_scalar, n_bool, size, p_bool)
    with torch.no_grad():
        out_planes = net(torch.from_numpy(dense))
    write_io(os.path.join(OUT, f"{name}.planes.io.bin"), batch, dense, out_planes)
    with open(os.path.join(OUT, f"{name}.planes.packed.bin"), "wb") as f:
        f.write(batch.to_bytes(1, "little"))
        f.write(bits.tobytes())
        f.write(scalars_in.tobytes())

    randn = torch.randn(batch, n_scalar + n_bool, size, size, generator=gen)
    with torch.no_grad():
        out_randn = net(randn)
    write_io(os.path.join(OUT, f"{name}.randn.io.bin"), batch, randn.numpy(), out_randn)

    if layers:
        acts = {}

        def hook(label):
            def f(_m, _i, o):
                acts[label] = o.detach().numpy().astype(np.float32)
            return f

        handles = []
        for i, m in enumerate(getattr(net.common, "encoders", [])):  # (n, b, d_model) -> [b][n][d_model]
            handles.append(m.register_forward_hook(
                lambda _m, _i, o, label=f"encoder.{i}": acts.__setitem__(label, o.detach().permute(1, 0, 2).contiguous().numpy())))
        for i, m in enumerate(getattr(net.common, "tower", [])):
            handles.append(m.register_forward_hook(hook(f"tower.{i}")))
            if isinstance(m, ResBlock):
                handles.append(m.seq[2].register_forward_hook(hook(f"tower.{i}.mid")))
        handles.append(net.scalar_head.seq[1].register_forward_hook(hook("scalar_head.conv_relu")))
        handles.append(net.scalar_head.seq[4].register_forward_hook(hook("scalar_head.fc0_relu")))
        with torch.no_grad():
            net(torch.from_numpy(dense))
        for h in handles:
            h.remove()
        with open(os.path.join(OUT, f"{name}.layers.kzm"), "wb") as f:
            f.write(write_model({"input": "planes"}, acts))

    nparam = sum(int(np.prod(t.shape)) for t in tensors.values())
    print(f"{name}: {nparam} values, scalars {tuple(out_planes[0].shape)}, policy {tuple(out_planes[1].shape)}")


def gen_decode_kat():
    gen = torch.Generator().manual_seed(77)
    scalars = torch.randn(6, 5, generator=gen) * 2
    logits = torch.randn(6, 40, generator=gen) * 3
    counts = [0, 1, 2, 7, 33, 40]
    tensors = {
        "scalars": scalars.numpy(),
        "value": torch.tanh(scalars[:, 0]).numpy(),
        "wdl": torch.softmax(scalars[:, 1:4], dim=1).numpy(),
        "moves_left": scalars[:, 4].numpy().copy(),
        "logits": logits.numpy(),
    }
    for bi, n in enumerate(counts):
        idx = torch.randperm(40, generator=gen)[:n]
        tensors[f"indices.{bi}"] = idx.numpy().astype(np.int64)
        tensors[f"policy.{bi}"] = torch.softmax(logits[bi, idx], dim=0).numpy() if n else np.zeros(0, np.float32)
    with open(os.path.join(OUT, "decode_kat.kzm"), "wb") as f:
        f.write(write_model({"batch": 6, "policy_len": 40}, tensors))


def gen_ataxx_symmetry():
    """D4 move/plane symmetry tables of Ataxx, sizes 2..8: a data file of the reference
    (python/lib/mapping/ataxx_symmetry.json, written by rust/kz-misc/src/bin/write_ataxx_mapping.rs:62-88 from
    board-game's D4Symmetry) re-encoded as text: one line per (size, symmetry):
    size index transpose flip_x flip_y n map_mv[0..n)."""
    import json
    data = json.load(open(os.path.join(REF, "lib", "mapping", "ataxx_symmetry.json")))
    with open(os.path.join(OUT, "ataxx_symmetry.txt"), "w") as f:
        for si, syms in enumerate(data):
            for i, e in enumerate(syms):
                mv = e["map_mv"]
                f.write(f"{si + 2} {i} {int(e['transpose'])} {int(e['flip_x'])} {int(e['flip_y'])} {len(mv)} "
                        + " ".join(str(v) for v in mv) + "\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_ataxx_symmetry()
    gen_net("ataxx7_2x16", 1, 4, 0.3, layers=True, onnx=True,
            game_name="ataxx-7", depth=2, channels=16, head_kind="ataxx_conv")
    gen_net("ataxx7_4x64", 2, 2, 0.3,
            game_name="ataxx-7", depth=4, channels=64, head_kind="ataxx_conv")
    gen_net("chess_2x32_att", 3, 3, 0.05, onnx=True,
            game_name="chess", depth=2, channels=32, head_kind="attention", query_channels=16)
    gen_net("chess_2x32_dense_h", 4, 2, 0.05, onnx=True,
            game_name="chess", depth=2, channels=32, head_kind="dense", hidden_channels=2, hidden_size=24)
    gen_net("chess_1x32_dense", 5, 2, 0.05,
            game_name="chess", depth=1, channels=32, head_kind="dense", hidden_channels=1, hidden_size=None)
    # python go-N declares 4 bool planes (lib/games.py:185); the server's mapper uses 7
    # (rust/kz-selfplay/src/server/server.rs:193, kz-core/src/mapping/go.rs:46-55): cover both.
    gen_net("go9_2x16_conv", 6, 2, 0.25,
            game_name="go-9", depth=2, channels=16, head_kind="conv", extra_moves=1)
    gen_net("go9_2x16_conv_terr", 7, 2, 0.25, onnx=True,
            game_name="go-9", depth=2, channels=16, head_kind="conv", extra_moves=1, input_bool_channels=7)
    gen_decode_kat()


def main_round5():
    """The other games the server dispatches (rust/kz-selfplay/src/server/server.rs:114-185), added in round 5 without
    touching the fixtures above: arimaa-split through ArimaaPolicyHead (post_act.py:144-173; ArimaaSplitMapper: 26 bool +
    12 scalar planes, policy 1 + 6 + 4*64), ttt (3x3) and sttt (9x9) through DensePolicyHead — their policy_shape is
    (1, n, n), which ConvPolicyHead's assert (post_act.py:60) refuses."""
    os.makedirs(OUT, exist_ok=True)
    gen_net("arimaa_2x32", 11, 3, 0.05, onnx=True,
            game_name="arimaa-split", depth=2, channels=32, head_kind="arimaa", hidden_channels=2, hidden_size=16)
    gen_net("ttt_2x16_dense", 12, 4, 0.3, onnx=True,
            game_name="ttt", depth=2, channels=16, head_kind="dense", hidden_channels=None, hidden_size=None)
    gen_net("sttt_2x16_dense_h", 13, 3, 0.3, onnx=True,
            game_name="sttt", depth=2, channels=16, head_kind="dense", hidden_channels=2, hidden_size=24)


def main_attention():
    """PredictionHeads(AttentionTower, ScalarHead, head): the network python/main/supervised_main_alpha.py:69-77 builds
    (there: depth 16, d_model 256, 8 heads, d_k = d_v = 16, d_ff 256 on chess).  Small instances, one with a sequence
    length that is not a multiple of 16 (Ataxx 7x7: 49 squares) and d_k != d_v."""
    os.makedirs(OUT, exist_ok=True)
    gen_net("chess_att2x64", 21, 3, 0.05, layers=True, onnx=True,
            game_name="chess", depth=2, channels=64, head_kind="attention", query_channels=16, attention=(4, 16, 16, 96))
    gen_net("ataxx7_att2x32", 22, 3, 0.3, onnx=True,
            game_name="ataxx-7", depth=2, channels=32, head_kind="ataxx_conv", attention=(2, 8, 12, 48))
    gen_net("chess_att3x256", 23, 2, 0.05,
            game_name="chess", depth=3, channels=256, head_kind="attention", query_channels=32, attention=(8, 16, 16, 256))


def main_dense_network():
    """The reference's own test networks (python/main/write_test_networks.py:14-18: DenseNetwork(sttt, 1, 64, False / True))
    and a deeper one on chess."""
    os.makedirs(OUT, exist_ok=True)
    gen_net("sttt_dn1x64", 31, 3, 0.3, onnx=True, dense_network=("sttt", 1, 64, False))
    gen_net("sttt_dn1x64_res", 32, 3, 0.3, onnx=True, dense_network=("sttt", 1, 64, True))
    gen_net("chess_dn3x96_res", 33, 2, 0.05, dense_network=("chess", 3, 96, True))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dense_network":
        main_dense_network()
    elif len(sys.argv) > 1 and sys.argv[1] == "round5":
        main_round5()
    elif len(sys.argv) > 1 and sys.argv[1] == "attention":
        main_attention()
    else:
        main()
