"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/kz_hip.h declares,
parses models on the host, and fails loudly (no fallback) where a GPU is needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from kzero_amd import capi, synth
from kzero_amd.model_file import read_model, write_model
from tests import oracle_lib as O

HEADER = os.path.join(O.REPO, "include", "kz_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in kz_hip.h but not exported"
    assert set(names) == set(capi.SIGNATURES), "capi.py binding out of sync with kz_hip.h"


@pytest.mark.parametrize("name", O.GOLDEN_NETS)
def test_model_info_matches_oracle(name):
    blob = O.load_blob(name)
    model = capi.Model(blob=blob)
    net = O.OracleNet(blob)
    i = model.info
    assert (i.input_channels, i.board_h, i.board_w) == (net.c_in, net.h, net.w)
    assert (i.input_scalar_channels, i.input_bool_channels) == (net.n_scalar, net.n_bool)
    assert (i.policy_len, i.tower_depth, i.tower_channels) == (net.policy_len, net.depth, net.channels)
    assert i.bits_bytes == (net.n_bool * net.h * net.w + 7) // 8
    _, tensors = read_model(blob)
    assert i.param_count == sum(t.size for k, t in tensors.items() if t.dtype == np.float32)


def test_flops_match_baseline_md():
    """FLOP/eval counted by hooking the reference modules."""
    chess = capi.Model(blob=synth.random_model("chess", 20, 256, "attention"))
    assert abs(chess.info.flops_per_eval / 3.049e9 - 1) < 2e-3
    ataxx = capi.Model(blob=synth.random_model("ataxx-7", 8, 128, "ataxx_conv"))
    assert abs(ataxx.info.flops_per_eval / 233.5e6 - 1) < 2e-3
    a0 = capi.Model(blob=synth.random_model("ataxx-7", 4, 64, "ataxx_conv"))
    assert abs(a0.info.flops_per_eval / 29.67e6 - 1) < 2e-3
    go = capi.Model(blob=synth.random_model("go-19", 40, 256, "conv"))
    assert abs(go.info.flops_per_eval / 34.13e9 - 1) < 3e-3


def test_which_models_the_parity_arithmetic_takes():
    """kz_model_supports_dtype needs no GPU: it is what the Rust binding's default (`KZ_HIP_DTYPE=parity`, hip.rs) asks before
    it creates an engine.  KZ_DTYPE_F32_SPLIT16: the one-launch shapes, since round 3 Go-size boards per layer, and since
    round 4 towers of any channel count up to 512 (widened to the next multiple of 64 by zero filters); not: a tower
    without blocks, or a channel count beyond 512 that is no multiple of 64."""
    S = capi.KZ_DTYPE_F32_SPLIT16
    for game, depth, ch, head, want in [("chess", 20, 256, "attention", True), ("ataxx-7", 8, 128, "ataxx_conv", True),
                                        ("go-9", 4, 128, "conv", True), ("go-19", 40, 256, "conv", True),
                                        ("go-19", 2, 64, "conv", True), ("go-9", 2, 256, "conv", True),
                                        ("chess", 2, 32, "attention", True), ("go-19", 2, 96, "conv", True),
                                        ("chess", 2, 192, "attention", True), ("chess-hist-2", 2, 256, "attention", True),
                                        ("chess", 0, 64, "attention", False), ("go-19", 1, 520, "conv", False)]:
        model = capi.Model(blob=synth.random_model(game, depth, ch, head))
        assert model.supports_dtype(capi.KZ_DTYPE_F32) and model.supports_dtype(capi.KZ_DTYPE_F16)
        assert model.supports_dtype(S) == want, (game, depth, ch)


def test_environment_switches_are_exactly_the_documented_ones():
    """include/kz_hip.h promises a complete list of the environment switches libkzhip.so reads: compare it with the
    KZ_* strings of the built library.  Experiments and ablation knobs live in libkzhip_exp.so only."""
    import subprocess
    text = open(HEADER).read()
    block = text[text.index("Environment switches read by kz_engine_create"):text.index("Name of the path the engine chose")]
    documented = set(re.findall(r"^ \*   (KZ_[A-Z0-9_]+)=", block, flags=re.M))
    out = subprocess.run(["strings", "-n", "4", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    constants = {"KZ_DTYPE_F32", "KZ_DTYPE_F16", "KZ_DTYPE_F32_SPLIT16"}  # named in error messages, not read from the environment
    in_library = set(re.findall(r"\bKZ_[A-Z0-9_]+\b", out)) - constants
    assert documented == in_library, (sorted(documented), sorted(in_library))
    assert len(documented) <= 6
    for needle in ("ABLATE", "LDS_MIN", "KZ_HIP_GRAPH", "KZ_BOARD_CONV2", "MFMA32", "launch_tower_resident4", "kz_board_conv2"):
        assert needle not in out, needle


def test_model_load_from_path(tmp_path):
    p = tmp_path / "m.kzm"
    p.write_bytes(O.load_blob("ataxx7_2x16"))
    assert capi.Model(path=str(p)).info.policy_len == 834
    with pytest.raises(capi.KzError, match="cannot open"):
        capi.Model(path=str(tmp_path / "missing.kzm"))


def test_bad_models_are_rejected_with_a_message():
    with pytest.raises(capi.KzError, match="KZMODEL1"):
        capi.Model(blob=b"not a model at all")
    blob = O.load_blob("ataxx7_2x16")
    with pytest.raises(capi.KzError, match="truncated"):
        capi.Model(blob=blob[:len(blob) // 2])
    meta, tensors = read_model(blob)
    tensors = dict(tensors)
    del tensors["common.tower.1.seq.1.running_var"]
    with pytest.raises(capi.KzError, match="running_var"):
        capi.Model(blob=write_model(meta, tensors))
    meta2 = dict(meta)
    meta2["policy_len"] = 100
    with pytest.raises(capi.KzError, match="policy_len"):
        capi.Model(blob=write_model(meta2, dict(read_model(blob)[1])))


def test_no_cpp_exception_crosses_the_c_abi():
    """The host on the other side of the C ABI is Rust (kzero_amd/rust/hip.rs): an exception that unwinds through `extern "C"`
    is undefined behaviour there.  Absurd sizes — from a corrupted settings value or a hostile model file — must come back
    as a non-zero return with a message (the shim turns that into the panic rust/kz-core/src/network/cudnn.rs:29-43 raises),
    and the process must survive.  Runs without a GPU: the checks sit in front of the first HIP call."""
    import struct
    model = capi.Model(blob=O.load_blob("ataxx7_2x16"))
    for mb in (2**31 - 1, 2**30, 50_000_000):
        with pytest.raises(capi.KzError, match="max_batch .* too large"):
            capi.Engine(model, 0, mb, capi.KZ_DTYPE_F32)
    with pytest.raises(capi.KzError, match="positive"):
        capi.Engine(model, 0, 0, capi.KZ_DTYPE_F32)
    # a container whose declared counts overflow every allocation: n_meta / n_tensors = 0xFFFFFFFF, a tensor of 2^62 dims
    blob = O.load_blob("ataxx7_2x16")
    for bad in (blob[:8] + struct.pack("<I", 0xFFFFFFFF) + blob[12:],
                blob[:8] + struct.pack("<I", 0) + struct.pack("<I", 0xFFFFFFFF) + b"\x00" * 64,
                blob[:8] + struct.pack("<I", 0) + struct.pack("<I", 1) + struct.pack("<H", 1) + b"t" + struct.pack("<BI", 0, 0xFFFFFFFF) + b"\x00" * 64,
                blob[:8] + struct.pack("<I", 1) + struct.pack("<H", 0xFFFF) + b"k" * 10):
        with pytest.raises(capi.KzError) as ei:
            capi.Model(blob=bad)
        assert str(ei.value)                       # a message, not an abort
    # the ONNX reader's side of the same rule: a length-delimited field that claims 2^63 bytes
    with pytest.raises(capi.KzError):
        capi.Model(blob=b"\x08\x07\x3a" + b"\xff" * 9 + b"\x7f" + b"\x00" * 32)
    # ... and the library still works afterwards
    assert capi.Model(blob=blob).info.policy_len == 17 * 49 + 1


def test_the_guard_turns_every_kind_of_exception_into_a_return_code():
    """The guard every entry point runs through (kz_engine_util.hpp `guarded`), exercised for real: the experiment build
    throws from inside kz_device_count on request — std::length_error, std::bad_alloc, a non-std exception."""
    import subprocess, sys
    exp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments", "libkzhip_exp.so")
    if not os.path.exists(exp):
        pytest.skip("libkzhip_exp.so not built")
    for mode, needle in (("1", "C++ exception: vector::_M_default_append"), ("2", "out of host memory"), ("3", "unknown C++ exception")):
        code = ("from kzero_amd import capi\n"
                "try:\n    capi.device_count(); print('NO ERROR')\n"
                "except capi.KzError as e:\n    print('KzError:', e)\n")
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KZ_LIB_PATH=exp, KZ_TEST_THROW=mode),
                           capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=120)
        assert p.returncode == 0 and "KzError: kz_device_count: " + needle in p.stdout, (p.stdout, p.stderr)


def test_no_gpu_means_loud_failure_not_fallback():
    """The product path must fail when it cannot run on a GPU; it never computes on the CPU."""
    try:
        n = capi.device_count()
    except capi.KzError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is visible")
    model = capi.Model(blob=O.load_blob("ataxx7_2x16"))
    with pytest.raises(capi.KzError):
        capi.Engine(model, 0, 4, capi.KZ_DTYPE_F32)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under kzero_amd/ may import, link or load it."""
    pkg = os.path.join(O.REPO, "kzero_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".sh")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "kzoracle" not in text and "oracle_lib" not in text and "kz_oracle" not in text, f
    out = os.popen(f"readelf -d {capi.LIB_PATH}").read()
    assert "oracle" not in out


ONNX_NETS = {"ataxx7_2x16": 1, "chess_2x32_att": 8, "chess_2x32_dense_h": 8, "go9_2x16_conv_terr": 6}  # name: scalar planes


@pytest.mark.parametrize("name", sorted(ONNX_NETS))
def test_onnx_reader_recovers_the_architecture(name):
    """N1: the ONNX file the reference's trainer emits (export call of python/lib/save_onnx.py:107-119, run on the
    reference's modules by oracle/gen_golden.py) parses into the same architecture as the KZMODEL1 container of the
    same network."""
    kzm = capi.Model(blob=O.load_blob(name)).info
    onnx = capi.Model(path=os.path.join(O.GOLDEN, f"{name}.onnx"), onnx_scalar_channels=ONNX_NETS[name]).info
    for field in ("input_channels", "board_h", "board_w", "input_scalar_channels", "input_bool_channels", "policy_len",
                  "tower_depth", "tower_channels", "policy_kind", "bits_bytes"):
        assert getattr(onnx, field) == getattr(kzm, field), field
    assert onnx.flops_per_eval == kzm.flops_per_eval
    # generic entry point: ONNX is recognised, but the scalar/bool split is unknown
    auto = capi.Model(path=os.path.join(O.GOLDEN, f"{name}.onnx")).info
    assert auto.input_scalar_channels == -1 and auto.bits_bytes == -1 and auto.policy_len == kzm.policy_len


def test_onnx_reader_rejects_what_it_does_not_understand():
    blob = open(os.path.join(O.GOLDEN, "ataxx7_2x16.onnx"), "rb").read()
    with pytest.raises(capi.KzError, match="ONNX"):
        capi.Model(blob=blob[:len(blob) // 3], onnx_scalar_channels=1)
    with pytest.raises(capi.KzError, match="exceeds the input channels"):
        capi.Model(blob=blob, onnx_scalar_channels=9)
    # a graph whose op type was renamed is not silently accepted
    bad = blob.replace(b"BatchNormalization", b"BatchNormalizatioX")
    with pytest.raises(capi.KzError, match="unsupported ONNX graph"):
        capi.Model(blob=bad, onnx_scalar_channels=1)


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/kz_hip.h is the boundary any FFI binds: it must compile as C99 with nothing but the standard headers, and
    examples/eval_packed.c (model load -> engine -> submit / zero-copy wait) must link against libkzhip.so alone."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "kz_hip.h"\nint main(void) { return KZ_ENGINE_SLOTS == 4 && KZ_DTYPE_F32_SPLIT16 == 2 ? 0 : 1; }\n')
    REPO = O.REPO
    inc = os.path.join(REPO, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(tmp_path / "abi")])
    assert subprocess.run([str(tmp_path / "abi")]).returncode == 0
    lib = os.path.join(REPO, "kzero_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc,
                           os.path.join(REPO, "examples", "eval_packed.c"), "-L", lib, "-lkzhip", f"-Wl,-rpath,{lib}",
                           "-o", str(tmp_path / "eval_packed")])
