"""N1 beyond one exporter: every file of tests/golden/onnx_variants/ — the golden networks as other exporter versions and
settings write them (opsets 9 / 11 / 13, no constant folding, training-mode export of an eval net with its BatchNorms
un-folded, initializers as graph inputs, Reshape for Flatten with a constant or a computed target, MatMul + Add or Gemm
with transB = 0 for Linear, Constant nodes for initializers, Identity / Dropout / Cast no-ops; oracle/gen_onnx_variants.py)
and the legacy (value, wdl, policy) output form (rust/kz-core/src/network/common.rs:42-49,186-190) — parses to the SAME
model as the network's KZMODEL1 container: architecture descriptor and every folded tensor (tests/cpp/compare_models.cpp,
built with AddressSanitizer + UBSan).  CPU only; the GPU side is tests/test_gpu_parity.py::test_onnx_variants_match_golden."""
import json
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "kzero_amd", "csrc")
BUILD = os.path.join(REPO, "tests", "cpp", "build")
GOLDEN = os.path.join(REPO, "tests", "golden")
VARIANTS = os.path.join(GOLDEN, "onnx_variants")
MANIFEST = json.load(open(os.path.join(VARIANTS, "manifest.json")))


@pytest.fixture(scope="module")
def compare_exe():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "compare_models")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           os.path.join(REPO, "tests", "cpp", "compare_models.cpp"), os.path.join(CSRC, "kz_model.cpp"),
                           os.path.join(CSRC, "kz_onnx.cpp"), "-o", exe])
    return exe


def test_the_variant_set_is_wide_enough():
    assert len(MANIFEST) >= 12
    kinds = {m["variant"] for m in MANIFEST}
    for needed in ("opset9", "opset11", "opset13", "nofold", "preserve", "init_inputs", "reshape", "dynamic_reshape", "matmul_add",
                   "gemm_transb0", "identity", "constants", "dropout_cast", "legacy3"):
        assert needed in kinds, needed
    ops = set().union(*(set(m["ops"]) for m in MANIFEST))
    assert {"Reshape", "MatMul", "Identity", "Dropout", "Cast", "Slice", "Shape"} <= ops
    assert {m["opset"] for m in MANIFEST} >= {9, 10, 11, 13}
    for m in MANIFEST:
        assert os.path.exists(os.path.join(VARIANTS, m["file"]))


@pytest.mark.parametrize("entry", MANIFEST, ids=lambda m: f"{m['net']}.{m['variant']}")
def test_variant_parses_to_the_container_model(compare_exe, entry):
    args = [compare_exe]
    if entry["variant"] == "legacy3":
        args.append("--legacy")
    args += [os.path.join(VARIANTS, entry["file"]), str(entry["scalar_planes"]), os.path.join(GOLDEN, f"{entry['net']}.kzm")]
    out = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "models equal" in out.stdout, out.stdout + out.stderr[-2000:]


@pytest.mark.parametrize("name,n_scalar", [("arimaa_2x32", 12), ("ttt_2x16_dense", 0), ("sttt_2x16_dense_h", 0),
                                           ("chess_att2x64", 8), ("ataxx7_att2x32", 1), ("sttt_dn1x64", 0), ("sttt_dn1x64_res", 0)])
def test_round5_games_exported_by_the_reference_parse_to_the_container_model(compare_exe, name, n_scalar):
    """The other games the server dispatches (server.rs:114-185): arimaa-split's ArimaaPolicyHead (a fifth head kind) and
    ttt / sttt through DensePolicyHead with its trailing .view(-1, 1, n, n) — the reference exporter's opset-10 files.
    And PredictionHeads(AttentionTower, ...) (python/lib/model/attention.py; supervised_main_alpha.py:69-77): 258 nodes of
    MatMul / Reshape / Transpose / Slice / Softmax / spelled-out LayerNorm for two encoder layers.  And DenseNetwork
    (python/lib/model/simple.py; write_test_networks.py:14-18): no tower, no heads, the outputs two slices of one Linear."""
    out = subprocess.run([compare_exe, os.path.join(GOLDEN, f"{name}.onnx"), str(n_scalar), os.path.join(GOLDEN, f"{name}.kzm")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "models equal" in out.stdout, out.stdout + out.stderr[-2000:]


def test_a_different_architecture_is_still_rejected_with_a_message(compare_exe, tmp_path):
    """The normalising pass accepts other spellings, not other networks: a graph whose residual Add is removed fails."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import onnx_wire as W
    m = W.Model(open(os.path.join(GOLDEN, "ataxx7_2x16.onnx"), "rb").read())
    add = [n for n in m.nodes if n.op == "Add"][0]
    add.op = "Mul"
    bad = tmp_path / "mul.onnx"
    bad.write_bytes(m.serialize())
    out = subprocess.run([compare_exe, str(bad), "1", os.path.join(GOLDEN, "ataxx7_2x16.kzm")], capture_output=True, text=True)
    assert out.returncode == 1 and "ONNX rejected" in out.stdout, out.stdout + out.stderr
