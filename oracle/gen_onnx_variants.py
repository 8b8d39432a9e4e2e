#!/usr/bin/env python3
"""ONNX variants of the golden networks.  TEST INFRASTRUCTURE — runs ONLY in the build container (imports the reference).

The reference loads any graph generically (`load_graph_from_onnx_path` + `optimize_graph`,
rust/kz-selfplay/src/server/server_alphazero.rs:126-128) and its trainer pins only `torch>=1.9.0`
(python/requirements.txt:2), so the files it meets come from many exporter versions.  This script writes, for networks
whose `.kzm` and `.io.bin` goldens are already committed (oracle/gen_golden.py), the SAME network as other exporters
would: the weights are loaded from the committed container into the reference's own modules
(python/lib/model/post_act.py) and exported with other settings of `torch.onnx.export` (python/lib/save_onnx.py:107-119:
opset, constant folding, training mode, initializers as inputs), and the committed opset-10 files are rewritten node by
node (oracle/onnx_wire.py: Flatten -> Reshape, Gemm -> MatMul + Add, Gemm with transB = 0, Identity insertion, Constant
initializers, Dropout in front of the heads).  Every variant must give the golden outputs (tests/test_gpu_parity.py::
test_onnx_variants_match_golden, tests/test_onnx_variants.py on the CPU).

Output: tests/golden/onnx_variants/<net>.<variant>.onnx and a manifest.json.
"""
import json
import os
import struct
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/python")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import gen_golden  # noqa: E402  (build(): the reference modules)
import onnx_wire as W  # noqa: E402
from kzero_amd.model_file import read_model  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")
OUT = os.path.join(GOLDEN, "onnx_variants")

NETS = {  # name -> (build kwargs as in gen_golden.main, scalar planes)
    "ataxx7_2x16": (dict(game_name="ataxx-7", depth=2, channels=16, head_kind="ataxx_conv"), 1),
    "chess_2x32_att": (dict(game_name="chess", depth=2, channels=32, head_kind="attention", query_channels=16), 8),
    "chess_2x32_dense_h": (dict(game_name="chess", depth=2, channels=32, head_kind="dense", hidden_channels=2, hidden_size=24), 8),
    "go9_2x16_conv_terr": (dict(game_name="go-9", depth=2, channels=16, head_kind="conv", extra_moves=1, input_bool_channels=7), 6),
}


def load_net(name):
    kw, _ = NETS[name]
    game, net, meta, (n_scalar, n_bool) = gen_golden.build(**kw)
    _, tensors = read_model(open(os.path.join(GOLDEN, f"{name}.kzm"), "rb").read())
    state = net.state_dict()
    for k in state:
        if k.endswith("num_batches_tracked"):
            continue
        state[k] = torch.from_numpy(np.array(tensors[k]))
    net.load_state_dict(state)
    net.eval()
    return net, (n_scalar + n_bool, game.board_size, game.board_size)


def export(net, shape, path, **kw):
    import warnings
    import torch.onnx._internal.torchscript_exporter.onnx_proto_utils as opu
    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    batch_axis = {0: "batch_size"}
    args = dict(model=net, args=(torch.randn(2, *shape),), f=path, input_names=["input"], output_names=["scalars", "policy"],
                dynamic_axes={k: batch_axis for k in ["input", "scalars", "policy"]}, opset_version=10, dynamo=False)
    args.update(kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(**args)


# ---- rewrites of a committed opset-10 file ----
def rw_flatten_to_reshape(m):
    """Flatten(axis=1) -> Reshape(x, [0, -1]) (what exporters emit for `.view(batch, -1)` / older `flatten`)."""
    out, k = [], 0
    for n in m.nodes:
        if n.op == "Flatten" and n.attr_int("axis", 1) == 1:
            shape = f"rw_flat_shape_{k}"
            k += 1
            m.initializers.append(W.tensor(shape, [2], 7, struct.pack("<2q", 0, -1)))
            out.append(W.make_node("Reshape", [n.inputs[0], shape], n.outputs))
        else:
            out.append(n)
    m.nodes = out


def rw_flatten_to_dynamic_reshape(m):
    """Flatten(axis=1) -> Reshape(x, Concat(Unsqueeze(Gather(Shape(x), 0)), [-1])): `x.view(x.shape[0], -1)` traced with a
    dynamic batch axis (the shape arithmetic stays in the graph)."""
    out, k = [], 0
    for n in m.nodes:
        if n.op == "Flatten" and n.attr_int("axis", 1) == 1:
            p = f"rw_dyn_{k}"
            k += 1
            m.initializers.append(W.tensor(p + "_zero", [], 7, struct.pack("<q", 0)))
            m.initializers.append(W.tensor(p + "_minus1", [1], 7, struct.pack("<q", -1)))
            out.append(W.make_node("Shape", [n.inputs[0]], [p + "_shape"]))
            out.append(W.make_node("Gather", [p + "_shape", p + "_zero"], [p + "_batch"], [W.attr_int("axis", 0)]))
            out.append(W.make_node("Unsqueeze", [p + "_batch"], [p + "_batch1"], [W.attr_ints("axes", [0])]))
            out.append(W.make_node("Concat", [p + "_batch1", p + "_minus1"], [p + "_target"], [W.attr_int("axis", 0)]))
            out.append(W.make_node("Reshape", [n.inputs[0], p + "_target"], n.outputs))
        else:
            out.append(n)
    m.nodes = out


def rw_gemm_to_matmul_add(m):
    """Gemm(x, W, b, transB=1) -> MatMul(x, W^T) + Add(b) (what `nn.Linear` becomes on batched inputs / other exporters)."""
    inits = {W.parse_tensor(t)[0]: t for t in m.initializers}
    out, k = [], 0
    for n in m.nodes:
        if n.op == "Gemm" and n.attr_int("transB", 0) == 1 and n.inputs[1] in inits:
            name, dims, dt, raw = W.parse_tensor(inits[n.inputs[1]])
            wt = np.frombuffer(raw, np.float32).reshape(dims).T.copy()
            tname = f"rw_matmul_w_{k}"
            mid = f"rw_matmul_out_{k}"
            k += 1
            m.initializers.append(W.tensor(tname, list(wt.shape), 1, wt.tobytes()))
            out.append(W.make_node("MatMul", [n.inputs[0], tname], [mid]))
            out.append(W.make_node("Add", [mid, n.inputs[2]], n.outputs))
        else:
            out.append(n)
    m.nodes = out


def rw_gemm_transb0(m):
    """Gemm with transB = 0 and the weight stored [in, out]."""
    inits = {W.parse_tensor(t)[0]: i for i, t in enumerate(m.initializers)}
    k = 0
    for n in m.nodes:
        if n.op == "Gemm" and n.attr_int("transB", 0) == 1 and n.inputs[1] in inits:
            name, dims, dt, raw = W.parse_tensor(m.initializers[inits[n.inputs[1]]])
            wt = np.frombuffer(raw, np.float32).reshape(dims).T.copy()
            tname = f"rw_gemm_w_{k}"
            k += 1
            m.initializers.append(W.tensor(tname, list(wt.shape), 1, wt.tobytes()))
            n.inputs[1] = tname
            n.attrs = [a for a in n.attrs if not any(f == 1 and v == b"transB" for f, _, v in a)]


def rw_identity(m):
    """An Identity behind every Relu and in front of every Conv weight (graph passes of other tools leave them)."""
    out, k = [], 0
    for n in m.nodes:
        if n.op == "Conv":
            alias = f"rw_id_w_{k}"
            k += 1
            out.append(W.make_node("Identity", [n.inputs[1]], [alias]))
            n.inputs[1] = alias
            out.append(n)
        elif n.op == "Relu":
            real = n.outputs[0]
            tmp = f"rw_id_relu_{k}"
            k += 1
            n.outputs[0] = tmp
            out.append(n)
            out.append(W.make_node("Identity", [tmp], [real]))
        else:
            out.append(n)
    m.nodes = out


def rw_constants(m):
    """Every initializer as a Constant node instead (exporters without an initializer list)."""
    consts = []
    for t in m.initializers:
        name = W.parse_tensor(t)[0]
        consts.append(W.make_node("Constant", [], [name], [[(1, 2, b"value"), (5, 2, t), (20, 0, 4)]]))
    m.initializers = []
    m.nodes = consts + m.nodes


def rw_dropout(m):
    """Dropout (inference: identity) between the tower and the heads, and Cast-to-float of the input (no-ops exporters of
    training-mode graphs leave behind)."""
    # the tower output = the tensor at least three nodes read
    reads = {}
    for n in m.nodes:
        for i in n.inputs:
            reads.setdefault(i, []).append(n)
    bn = [n for n in m.nodes if n.op == "BatchNormalization"][-1]
    t = bn.outputs[0]
    new = "rw_dropout_out"
    for n in reads.get(t, []):
        n.inputs = [new if i == t else i for i in n.inputs]
    idx = m.nodes.index(bn)
    m.nodes.insert(idx + 1, W.make_node("Dropout", [t], [new]))
    # Cast(input -> FLOAT)
    stem = [n for n in m.nodes if n.op == "Conv" and n.inputs[0] == "input"][0]
    stem.inputs[0] = "rw_cast_input"
    m.nodes.insert(m.nodes.index(stem), W.make_node("Cast", ["input"], ["rw_cast_input"], [W.attr_int("to", 1)]))


REWRITES = {"reshape": rw_flatten_to_reshape, "dynamic_reshape": rw_flatten_to_dynamic_reshape, "matmul_add": rw_gemm_to_matmul_add, "gemm_transb0": rw_gemm_transb0,
            "identity": rw_identity, "constants": rw_constants, "dropout_cast": rw_dropout}

EXPORTS = {
    "opset9": dict(opset_version=9),
    "opset11": dict(opset_version=11),
    "opset13": dict(opset_version=13),
    "nofold": dict(do_constant_folding=False),
    "preserve": dict(training=torch.onnx.TrainingMode.PRESERVE),  # on an eval-mode net: BatchNormalization stays unfolded
    "init_inputs": dict(keep_initializers_as_inputs=True),
    "opset13_nofold_preserve": dict(opset_version=13, do_constant_folding=False, training=torch.onnx.TrainingMode.PRESERVE),
}

PLAN = {  # which variants of which net (every variant on the two small nets, the combined ones on the large heads)
    "ataxx7_2x16": ["opset9", "opset11", "opset13", "nofold", "preserve", "init_inputs", "reshape", "matmul_add",
                    "gemm_transb0", "identity", "constants", "dropout_cast", "dynamic_reshape", "legacy3"],
    "go9_2x16_conv_terr": ["opset13_nofold_preserve", "reshape+matmul_add+identity", "constants", "dynamic_reshape+matmul_add",
                           "legacy3"],
    "chess_2x32_att": ["opset13_nofold_preserve", "opset9", "reshape+matmul_add+identity"],
    "chess_2x32_dense_h": ["opset13_nofold_preserve", "reshape+matmul_add+identity", "gemm_transb0"],
}


class LegacyOutputs(torch.nn.Module):
    """The older output form `check_graph_shapes` / `decode_output` still accept (rust/kz-core/src/network/common.rs:42-49,
    186-190): (value [B], wdl [B, 3], policy) instead of (scalars [B, 5], policy).  No module of the reference writes it any
    more, so the stand-in is the reference network with its scalars cut up: value = scalars[:, 0], wdl = scalars[:, 1:4]."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x):
        scalars, policy = self.net(x)
        return scalars[:, 0], scalars[:, 1:4], policy


def export_legacy(net, shape, path):
    import warnings
    import torch.onnx._internal.torchscript_exporter.onnx_proto_utils as opu
    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    batch_axis = {0: "batch_size"}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model=LegacyOutputs(net).eval(), args=(torch.randn(2, *shape),), f=path, input_names=["input"],
                          output_names=["value", "wdl", "policy"],
                          dynamic_axes={k: batch_axis for k in ["input", "value", "wdl", "policy"]}, opset_version=10, dynamo=False)


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = []
    for name, variants in PLAN.items():
        net, shape = load_net(name)
        base = open(os.path.join(GOLDEN, f"{name}.onnx"), "rb").read()
        for v in variants:
            path = os.path.join(OUT, f"{name}.{v.replace('+', '_')}.onnx")
            if v == "legacy3":
                export_legacy(net, shape, path)
            elif v in EXPORTS:
                export(net, shape, path, **EXPORTS[v])
            else:
                m = W.Model(base)
                for step in v.split("+"):
                    REWRITES[step](m)
                open(path, "wb").write(m.serialize())
            m = W.Model(open(path, "rb").read())
            ops = sorted(set(m.ops()))
            manifest.append({"net": name, "variant": v, "file": os.path.basename(path), "scalar_planes": NETS[name][1],
                             "opset": m.opset(), "nodes": len(m.nodes), "ops": ops})
            print(name, v, m.opset(), len(m.nodes), " ".join(ops))
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
