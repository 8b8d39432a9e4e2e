"""Mutation fuzz of the model parsers (KZMODEL1 and ONNX) under AddressSanitizer + UBSan, on the CPU: a mutated file may
be rejected with a message or accepted, the parser must never crash or read out of bounds."""
import glob
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "kzero_amd", "csrc")
BUILD = os.path.join(REPO, "tests", "cpp", "build")
GOLDEN = os.path.join(REPO, "tests", "golden")


@pytest.mark.timeout(600)
def test_parsers_survive_mutated_files():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "fuzz_parsers")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", os.path.join(REPO, "tests", "cpp", "fuzz_parsers.cpp"),
                           os.path.join(CSRC, "kz_model.cpp"), os.path.join(CSRC, "kz_onnx.cpp"), "-o", exe])
    # (+ the graphs other exporters write: every rewrite of the normalising pass meets mutated input too)
    variants = [os.path.join(GOLDEN, "onnx_variants", n) for n in
                ("ataxx7_2x16.opset13.onnx", "ataxx7_2x16.dynamic_reshape.onnx", "ataxx7_2x16.legacy3.onnx",
                 "ataxx7_2x16.constants.onnx", "go9_2x16_conv_terr.reshape_matmul_add_identity.onnx",
                 "ataxx7_2x16.dropout_cast.onnx", "chess_2x32_att.opset9.onnx")]
    files = sorted(glob.glob(os.path.join(GOLDEN, "*.onnx"))) + variants + [os.path.join(GOLDEN, n) for n in
                                                                  ("ataxx7_2x16.kzm", "chess_2x32_att.kzm",
                                                                   "go9_2x16_conv.kzm", "chess_1x32_dense.kzm")]
    for seed in ("7", "10"):  # (10: the seed that found an int overflow in the descriptor before it was range-checked)
        out = subprocess.run([exe, "400", seed] + files, capture_output=True, text=True, timeout=250,
                             env={**os.environ, "UBSAN_OPTIONS": "print_stacktrace=1"})
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        assert "parser fuzz ok" in out.stdout
