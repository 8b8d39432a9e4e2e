"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the golden vectors.

Tolerances:
  f32 path: max |delta| <= 1e-4 on policy logits and the 5 scalars (BASELINE.json north_star / BASELINE.md §4).
  f16 path: 1e-4 is not attainable by construction (f16 storage of weights and activations, f32 accumulate: every
            layer rounds the residual stream to 11 bits).  Stated tolerance, per board and per output tensor:
            max |delta| <= F16_REL * max(1, max |ref|)   with F16_REL = 3.5e-3 (1.5 x the largest value any test measures: 2.3e-3)
            rms |delta| <= F16_RMS * max(1, max |ref|)   with F16_RMS = 6e-4   (1.5 x the largest measured on networks of up to
                                                                                41 convolutions, 4.1e-4: rms |dlogit| 1.1e-3 at logit
                                                                                scale 3.3, chess 20x256; G8's 81 convolutions:
                                                                                F16_RMS_DEEP = 1.05e-3 = 1.5 x its 6.9e-4)
            and, on the three BASELINE configs at full size (A1, C1, G8), post-softmax max |delta p| <= F16_SOFTMAX_ATOL = 1e-3.
            The bounds sit 1.5 x above what is measured (round 5 had them 2 x above: a 2 x numerical regression passed);
            a mis-scaled layer or a wrong weight fragment moves the rms by far more than that.
            Two f16 paths of this library against each other (same operands and rounding points, different summation
            order) are bounded relative to the output scale as well: F16_PATHS_REL for 2-block nets, _DEEP for 40+ layers.
"""
import os

import numpy as np
import pytest

from kzero_amd import capi, synth
from kzero_amd.model_file import read_model
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

F32_ATOL = 1e-4
F16_REL = 3.5e-3
F16_RMS = 6e-4
F16_SOFTMAX_ATOL = 1e-3
# the one network twice as deep as the others (G8: 81 convolutions against <= 41): rms 6.9e-4 on the five scalars, 4.1e-4 on
# the policy, measured against this library's exact-f32 path on 64 of the 512 boards — again 1.5 x the measured
F16_RMS_DEEP = 1.05e-3
# two f16 paths of this library against each other (measured <= 2e-4 on 2-block nets, <= 4e-3 after 41 convolutions at
# logit scale 3.3): absolute on the shallow nets (outputs of order 1), relative to the output scale on the deep ones
F16_PATHS_ATOL = 2e-3
F16_PATHS_REL_DEEP = 3e-3


def assert_f16_paths_deep(a, b, what):
    """Two f16 paths after 40+ layers: max |delta| relative to the tensor's own scale."""
    scale = max(1.0, float(np.abs(b).max()))
    d = float(np.abs(a - b).max()) / scale
    print(f"[f16 paths] {what}: max |delta| / scale = {d:.3e} (scale {scale:.2f})")
    assert d <= F16_PATHS_REL_DEEP, f"{what}: {d:.3e} > {F16_PATHS_REL_DEEP}"


def assert_f32(actual, ref, what):
    err = np.abs(actual - ref).max() if actual.size else 0.0
    assert err <= F32_ATOL, f"{what}: max |delta| = {err:.3e} > 1e-4"


def assert_f16(actual, ref, what, rms_tol=None):
    rms_tol = F16_RMS if rms_tol is None else rms_tol
    scale = np.maximum(1.0, np.abs(ref).max(axis=-1, keepdims=True))
    rel = (np.abs(actual - ref) / scale).max()
    rms = float(np.sqrt(np.mean(((actual - ref) / scale) ** 2)))
    print(f"[f16] {what}: max |delta| / scale = {rel:.3e}, rms = {rms:.3e}")
    assert rel <= F16_REL, f"{what}: max |delta| / scale = {rel:.3e} > {F16_REL}"
    assert rms <= rms_tol, f"{what}: rms |delta| / scale = {rms:.3e} > {rms_tol}"
    return rel


def softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


@pytest.fixture(scope="module")
def dev():
    assert capi.device_count() >= 1
    return 0


@pytest.mark.parametrize("name", O.GOLDEN_NETS)
@pytest.mark.parametrize("kind", ["planes", "randn"])
def test_f32_dense_matches_golden_and_oracle(dev, name, kind):
    blob = O.load_blob(name)
    net = O.OracleNet(blob)
    x, s_gold, p_gold = O.read_io(name, kind, net.c_in, net.h, net.w, net.policy_len)
    s_or, p_or = net.forward(x)
    eng = capi.Engine(capi.Model(blob=blob), dev, 8, capi.KZ_DTYPE_F32)
    s, p = eng.eval_dense(x)
    assert_f32(s, s_gold, "scalars vs golden")
    assert_f32(p, p_gold, "policy vs golden")
    assert_f32(s, s_or, "scalars vs oracle")
    assert_f32(p, p_or, "policy vs oracle")


@pytest.mark.parametrize("name", O.GOLDEN_NETS)
@pytest.mark.parametrize("dtype", [capi.KZ_DTYPE_F32, capi.KZ_DTYPE_F16])
def test_packed_input(dev, name, dtype):
    """bits + scalars in, the GPU does encode_input_full (F0)."""
    blob = O.load_blob(name)
    net = O.OracleNet(blob)
    _, s_gold, p_gold = O.read_io(name, "planes", net.c_in, net.h, net.w, net.policy_len)
    bits, scalars_in = O.read_packed(name, net.n_bool, net.n_scalar, net.h, net.w)
    eng = capi.Engine(capi.Model(blob=blob), dev, 4, dtype)
    s, p = eng.eval_packed(bits, scalars_in)
    if dtype == capi.KZ_DTYPE_F32:
        assert_f32(s, s_gold, "scalars")
        assert_f32(p, p_gold, "policy")
    else:
        assert_f16(s, s_gold, "scalars")
        assert_f16(p, p_gold, "policy")
        assert np.abs(softmax(p) - softmax(p_gold)).max() < 5e-3


import json as _json

ONNX_VARIANTS = _json.load(open(os.path.join(O.GOLDEN, "onnx_variants", "manifest.json")))


@pytest.mark.parametrize("entry", ONNX_VARIANTS, ids=lambda m: f"{m['net']}.{m['variant']}")
def test_onnx_variants_match_golden(dev, entry):
    """N1 beyond one exporter (oracle/gen_onnx_variants.py: other opsets and export settings, rewritten graphs): every
    variant of a golden network gives the reference PyTorch outputs (<= 1e-4, f32).  The legacy (value, wdl, policy) output
    form (rust/kz-core/src/network/common.rs:42-49) comes back as scalars [B, 5] with moves_left = NaN — what decode_output
    makes of such a graph — here and through the device-side decode."""
    name = entry["net"]
    net = O.OracleNet(O.load_blob(name))
    _, s_gold, p_gold = O.read_io(name, "planes", net.c_in, net.h, net.w, net.policy_len)
    bits, scalars_in = O.read_packed(name, net.n_bool, net.n_scalar, net.h, net.w)
    path = os.path.join(O.GOLDEN, "onnx_variants", entry["file"])
    eng = capi.Engine(capi.Model(path=path, onnx_scalar_channels=entry["scalar_planes"]), dev, 4, capi.KZ_DTYPE_F32)
    s, p = eng.eval_packed(bits, scalars_in)
    assert_f32(p, p_gold, "policy")
    if entry["variant"] != "legacy3":
        assert_f32(s, s_gold, "scalars")
        return
    assert_f32(s[:, :4], s_gold[:, :4], "value and wdl logits")
    assert np.isnan(s[:, 4]).all(), "legacy graphs have no moves_left: NaN (common.rs:44)"
    moves = [list(range(0, net.policy_len, 7)) for _ in range(len(bits))]
    values, probs = eng.eval_packed_decoded(bits, scalars_in, moves)
    v_ref, p_ref = O.decode_output(s_gold, p_gold, moves)
    assert np.abs(values[:, :4] - v_ref[:, :4]).max() < 1e-5 and np.isnan(values[:, 4]).all()
    for a, b in zip(probs, p_ref):
        assert np.abs(a - b).max() < 1e-5


@pytest.mark.parametrize("name,n_scalar", [("ataxx7_2x16", 1), ("chess_2x32_att", 8), ("chess_2x32_dense_h", 8),
                                           ("go9_2x16_conv_terr", 6), ("arimaa_2x32", 12), ("ttt_2x16_dense", 0),
                                           ("sttt_2x16_dense_h", 0), ("chess_att2x64", 8), ("ataxx7_att2x32", 1),
                                           ("sttt_dn1x64", 0), ("sttt_dn1x64_res", 0)])
def test_onnx_models_match_golden(dev, name, n_scalar):
    """N1: the engine fed with the trainer's ONNX file gives the reference PyTorch outputs (<= 1e-4, f32) and agrees
    with the same network loaded from the KZMODEL1 container."""
    net = O.OracleNet(O.load_blob(name))
    x, s_gold, p_gold = O.read_io(name, "planes", net.c_in, net.h, net.w, net.policy_len)
    bits, scalars_in = O.read_packed(name, net.n_bool, net.n_scalar, net.h, net.w)
    onnx_path = os.path.join(O.GOLDEN, f"{name}.onnx")
    eng = capi.Engine(capi.Model(path=onnx_path, onnx_scalar_channels=n_scalar), dev, 4, capi.KZ_DTYPE_F32)
    s, p = eng.eval_packed(bits, scalars_in)
    assert_f32(s, s_gold, "scalars")
    assert_f32(p, p_gold, "policy")
    s2, p2 = capi.Engine(capi.Model(blob=O.load_blob(name)), dev, 4, capi.KZ_DTYPE_F32).eval_packed(bits, scalars_in)
    assert np.abs(s - s2).max() < 1e-5 and np.abs(p - p2).max() < 1e-5
    # loaded without the plane split: dense input works, packed input fails loudly
    blind = capi.Engine(capi.Model(path=onnx_path), dev, 4, capi.KZ_DTYPE_F32)
    s3, p3 = blind.eval_dense(x)
    assert_f32(p3, p_gold, "policy (dense)")
    with pytest.raises(capi.KzError, match="plane split"):
        blind.eval_packed(bits, scalars_in)


def test_device_side_decode_output(dev):
    """N2: decode_output on the GPU (tanh, wdl softmax, legal-move gather + softmax) against the oracle's restatement
    of rust/kz-core/src/network/common.rs:16-100 applied to the engine's own logits."""
    name = "chess_2x32_att"
    blob = O.load_blob(name)
    net = O.OracleNet(blob)
    bits, scalars_in = O.read_packed(name, net.n_bool, net.n_scalar, net.h, net.w)
    rng = np.random.default_rng(9)
    moves = [rng.permutation(net.policy_len)[:n].astype(np.int32) for n in (37, 0, 300)]  # 0: a finished game
    eng = capi.Engine(capi.Model(blob=blob), dev, 4, capi.KZ_DTYPE_F32)
    s, p = eng.eval_packed(bits, scalars_in)
    v_ref, probs_ref = O.decode_output(s, p, moves)
    v, probs = eng.eval_packed_decoded(bits, scalars_in, moves)
    np.testing.assert_allclose(v, v_ref, rtol=1e-5, atol=1e-6)
    for a, b in zip(probs, probs_ref):
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-7)
    assert abs(probs[0].sum() - 1) < 1e-5 and abs(probs[2].sum() - 1) < 1e-5 and probs[1].size == 0
    # where the reference asserts (sum > 0.0, common.rs:110) the call fails: an out-of-range index poisons the sum
    with pytest.raises(capi.KzError, match="strictly positive"):
        eng.eval_packed_decoded(bits, scalars_in, [np.array([5, net.policy_len], np.int32), moves[1], moves[2]])
    v2, _ = eng.eval_packed_decoded(bits, scalars_in, moves)  # still usable afterwards
    assert np.array_equal(v, v2)
    # the asynchronous pair with the decode on the device: two batches in flight, each with its own move lists
    moves_b = [m[::-1].copy() for m in moves]
    off_a = eng.submit_packed_decoded(0, bits, scalars_in, moves)
    off_b = eng.submit_packed_decoded(1, bits, scalars_in, moves_b)
    with pytest.raises(capi.KzError, match="in flight"):
        eng.submit_packed_decoded(0, bits, scalars_in, moves)
    with pytest.raises(capi.KzError, match="nothing submitted"):
        eng.wait(1, len(bits))  # a decoded submission is collected with wait_decoded
    v_b, probs_b = eng.wait_decoded(1, off_b)
    v_a, probs_a = eng.wait_decoded(0, off_a)
    assert np.array_equal(v_a, v) and np.array_equal(v_b, v)
    for a, b, c in zip(probs_a, probs, probs_b):
        assert np.array_equal(a, b)
        np.testing.assert_allclose(c[::-1], b, rtol=1e-6, atol=1e-8)
    with pytest.raises(capi.KzError, match="nothing submitted"):
        eng.wait_decoded(0, off_a)
    off_e = eng.submit_packed_decoded(0, bits, scalars_in, [np.array([net.policy_len], np.int32), moves[1], moves[2]])
    with pytest.raises(capi.KzError, match="strictly positive"):
        eng.wait_decoded(0, off_e)
    s3, p3 = eng.eval_packed(bits, scalars_in)  # the slot is free again
    assert np.array_equal(s3, s) and np.array_equal(p3, p)


def test_replay_recorded_positions(dev, tmp_path):
    """N4: positions recorded in the self-play output format (binary_output.rs:210-256) replay through the engine; the
    recorded move lists drive the device-side decode."""
    from kzero_amd.position_file import PositionFile
    from tests.test_position_file import _games, _write
    games, shape, ns, pshape = _games(np.random.default_rng(4))
    path = str(tmp_path / "games_3")
    _write(path, games, shape, ns, pshape)
    bits, scalars_in, moves = PositionFile(path).read_boards()
    blob = O.load_blob("ataxx7_2x16")
    net = O.OracleNet(blob)
    s_ref, p_ref = net.forward(O.encode_input_full(bits, scalars_in, 1, 3, 7, 7))
    v_ref, probs_ref = O.decode_output(s_ref, p_ref, moves)
    eng = capi.Engine(capi.Model(blob=blob), dev, 16, capi.KZ_DTYPE_F32)
    v, probs = eng.eval_packed_decoded(bits, scalars_in, moves)
    np.testing.assert_allclose(v, v_ref, rtol=1e-4, atol=1e-5)
    for a, b in zip(probs, probs_ref):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6)


def test_per_layer_activations_f32(dev, monkeypatch):
    monkeypatch.setenv("KZ_FORCE_GENERIC", "1")
    monkeypatch.setenv("KZ_KEEP_ACTIVATIONS", "1")
    name = "ataxx7_2x16"
    blob = O.load_blob(name)
    net = O.OracleNet(blob)
    x, _, _ = O.read_io(name, "planes", net.c_in, net.h, net.w, net.policy_len)
    _, ref = read_model(open(os.path.join(O.GOLDEN, f"{name}.layers.kzm"), "rb").read())
    eng = capi.Engine(capi.Model(blob=blob), dev, 4, capi.KZ_DTYPE_F32)
    eng.eval_dense(x)
    # the last block's pre-BN output is never materialised: the final BN is fused into its epilogue
    for key in ["tower.0", "tower.1.mid", "tower.1", "tower.2.mid", "tower.3"]:
        assert_f32(eng.read_activation(key, x.shape[0]), ref[key], key)


def test_batch_edges(dev):
    """empty, single, ragged and oversize batches; only `batch` rows are written (cudnn.rs:58,65,75-82)."""
    name = "chess_2x32_att"
    blob = O.load_blob(name)
    net = O.OracleNet(blob)
    x, s_gold, p_gold = O.read_io(name, "planes", net.c_in, net.h, net.w, net.policy_len)
    eng = capi.Engine(capi.Model(blob=blob), dev, 3, capi.KZ_DTYPE_F32)
    assert eng.max_batch == 3
    s0, p0 = eng.eval_dense(x[:0])
    assert s0.shape == (0, 5) and p0.shape == (0, net.policy_len)
    s1, p1 = eng.eval_dense(x[:1])
    assert_f32(s1, s_gold[:1], "single scalars")
    assert_f32(p1, p_gold[:1], "single policy")
    s3, p3 = eng.eval_dense(x)
    # per-sample independence: a board's result does not depend on its batch (eval-mode BN)
    assert np.array_equal(s3[:1], s1) and np.array_equal(p3[:1], p1)
    with pytest.raises(capi.KzError, match="exceeds max_batch"):
        eng.eval_dense(np.concatenate([x, x]))
    # the engine is still usable after an error
    s3b, _ = eng.eval_dense(x)
    assert np.array_equal(s3, s3b)


def test_async_slots_and_shared_weights(dev):
    name = "ataxx7_4x64"
    blob = O.load_blob(name)
    net = O.OracleNet(blob)
    _, s_gold, p_gold = O.read_io(name, "planes", net.c_in, net.h, net.w, net.policy_len)
    bits, scalars_in = O.read_packed(name, net.n_bool, net.n_scalar, net.h, net.w)
    model = capi.Model(blob=blob)
    e1 = capi.Engine(model, dev, 2, capi.KZ_DTYPE_F32)
    e2 = capi.Engine(model, dev, 2, capi.KZ_DTYPE_F32)  # shares the uploaded weights with e1
    n0 = e1.submit_packed(0, bits, scalars_in)
    n1 = e1.submit_packed(1, bits[::-1], scalars_in[::-1])
    with pytest.raises(capi.KzError, match="in flight"):
        e1.submit_packed(0, bits, scalars_in)
    s_b, p_b = e1.wait(1, n1)
    s_a, p_a = e1.wait(0, n0)
    assert_f32(s_a, s_gold, "slot 0")
    assert_f32(p_a, p_gold, "slot 0")
    assert np.array_equal(s_b[::-1], s_a) and np.array_equal(p_b[::-1], p_a)
    s_c, p_c = e2.eval_packed(bits, scalars_in)
    assert np.array_equal(s_c, s_a) and np.array_equal(p_c, p_a)
    with pytest.raises(capi.KzError, match="nothing submitted"):
        e1.wait(0, n0)
    # the zero-copy wait hands out the slot's pinned staging: same numbers, valid until the slot is submitted again
    n2 = e1.submit_packed(1, bits, scalars_in)
    s_v, p_v = e1.wait_view(1, n2)
    assert np.array_equal(s_v, s_a) and np.array_equal(p_v, p_a)
    with pytest.raises(capi.KzError, match="nothing submitted"):
        e1.wait_view(1, n2)
    e1.close()
    s_d, _ = e2.eval_packed(bits, scalars_in)  # weights stay alive while any engine uses them
    assert np.array_equal(s_d, s_a)


def test_device_resident_entry_points(dev):
    name = "go9_2x16_conv_terr"
    blob = O.load_blob(name)
    net = O.OracleNet(blob)
    x, s_gold, p_gold = O.read_io(name, "planes", net.c_in, net.h, net.w, net.policy_len)
    bits, scalars_in = O.read_packed(name, net.n_bool, net.n_scalar, net.h, net.w)
    b = x.shape[0]
    eng = capi.Engine(capi.Model(blob=blob), dev, b, capi.KZ_DTYPE_F32)
    d_s = capi.DeviceBuffer(dev, b * 5 * 4)
    d_p = capi.DeviceBuffer(dev, b * net.policy_len * 4)
    eng.enqueue_packed_device(capi.DeviceBuffer.from_host(dev, bits), bits.shape[1],
                              capi.DeviceBuffer.from_host(dev, scalars_in), b, d_s, d_p)
    eng.synchronize()
    assert_f32(d_s.to_host(np.float32, (b, 5)), s_gold, "packed device scalars")
    assert_f32(d_p.to_host(np.float32, (b, net.policy_len)), p_gold, "packed device policy")
    eng.enqueue_dense_device(capi.DeviceBuffer.from_host(dev, x), b, d_s, d_p)
    eng.synchronize()
    assert_f32(d_p.to_host(np.float32, (b, net.policy_len)), p_gold, "dense device policy")


def test_config_a1_ataxx_8x128_f32_batch256(dev):
    """BASELINE.json configs[1]: Ataxx 7x7, 8-block x 128ch ResNet fp32, executor batch 256 — vs the oracle."""
    blob = synth.random_model("ataxx-7", 8, 128, "ataxx_conv", seed=11)
    bits, scalars_in = synth.random_boards("ataxx-7", 256, seed=12)
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    eng = capi.Engine(capi.Model(blob=blob), dev, 256, capi.KZ_DTYPE_F32)
    assert eng.tower_path == "tower_resident_f32+heads"  # the whole network in one launch (128 workgroups of two boards)
    assert eng.launch_geometry(256) == (128, 2)
    s, p = eng.eval_packed(bits, scalars_in)
    assert_f32(s, s_ref, "scalars")
    assert_f32(p, p_ref, "policy")
    # the same through f16, with the stated f16 tolerance
    eng16 = capi.Engine(capi.Model(blob=blob), dev, 256, capi.KZ_DTYPE_F16)
    s16, p16 = eng16.eval_packed(bits, scalars_in)
    assert_f16(s16, s_ref, "f16 scalars")
    assert_f16(p16, p_ref, "f16 policy")
    sm = float(np.abs(softmax(p16) - softmax(p_ref)).max())
    print(f"ataxx 8x128 f16 vs oracle, 256 boards: max |dsoftmax| {sm:.2e}")
    assert sm <= F16_SOFTMAX_ATOL, f"A1 f16: post-softmax max |delta p| = {sm:.3e} > {F16_SOFTMAX_ATOL}"
    # and through the arithmetic the Rust binding defaults to (KZ_HIP_DTYPE=parity): ONE launch per batch since round 3 —
    # the same 1e-4, on every entry point (the asynchronous pair writes straight into the slot's pinned staging)
    split = capi.Engine(capi.Model(blob=blob), dev, 256, capi.KZ_DTYPE_F32_SPLIT16)
    assert split.tower_path == "tower_resident_split16+heads" and split.launch_geometry(256) == (128, 2)
    ss, sp = split.eval_packed(bits, scalars_in)
    assert_f32(ss, s_ref, "split16 scalars")
    assert_f32(sp, p_ref, "split16 policy")
    n = split.submit_packed(0, bits[:77], scalars_in[:77])  # ragged last workgroup
    sv, pv = split.wait_view(0, n)
    assert np.array_equal(sv, ss[:77]) and np.array_equal(pv, sp[:77])
    d_s, d_p = capi.DeviceBuffer(dev, 256 * 5 * 4), capi.DeviceBuffer(dev, 256 * net.policy_len * 4)
    split.enqueue_packed_device(capi.DeviceBuffer.from_host(dev, bits), bits.shape[1],
                                capi.DeviceBuffer.from_host(dev, scalars_in), 256, d_s, d_p)
    split.synchronize()
    assert np.array_equal(d_s.to_host(np.float32, (256, 5)), ss)
    assert np.array_equal(d_p.to_host(np.float32, (256, net.policy_len)), sp)


@pytest.fixture(scope="module")
def chess_full():
    blob = synth.random_model("chess", 20, 256, "attention", seed=21)
    bits, scalars_in = synth.random_boards("chess", 256, seed=22)
    return blob, bits, scalars_in


C1_PICK = np.arange(256)  # every board of the batch (rounds 2-3: 32 of them)


@pytest.fixture(scope="module")
def chess_full_oracle(chess_full):
    """The REAL oracle on ALL 256 boards of the full configuration (one board takes it about 0.3 s on one core; all host
    threads are used: ~5 s on the GPU box's 16).  Shared by the f16, exact-f32 and split-f16 tests."""
    blob, bits, scalars_in = chess_full
    assert len(set(C1_PICK.tolist())) == 256
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits[C1_PICK], scalars_in[C1_PICK], net.n_scalar, net.n_bool, net.h, net.w)
    return net.forward(dense, threads=os.cpu_count() or 1)


def test_config_c1_chess_20x256_f16_vs_oracle_sample(dev, chess_full, chess_full_oracle):
    """BASELINE.json configs[2] at full size against the oracle on all 256 boards."""
    blob, bits, scalars_in = chess_full
    pick = C1_PICK
    s_ref, p_ref = chess_full_oracle
    eng = capi.Engine(capi.Model(blob=blob), dev, 256, capi.KZ_DTYPE_F16)
    s, p = eng.eval_packed(bits, scalars_in)
    rs = assert_f16(s[pick], s_ref, "scalars")
    rp = assert_f16(p[pick], p_ref, "policy")
    print(f"tower path: {eng.tower_path}; logit scale {np.abs(p_ref).max():.2f}; max |dlogit| "
          f"{np.abs(p[pick] - p_ref).max():.3e} (rel {rp:.2e}); max |dscalar| {np.abs(s[pick] - s_ref).max():.3e} "
          f"(rel {rs:.2e}); max |dsoftmax| {np.abs(softmax(p[pick]) - softmax(p_ref)).max():.3e}")


def test_config_c1_f32_vs_oracle_sample(dev, chess_full, chess_full_oracle):
    """The two <= 1e-4 paths at the full configuration against the oracle on all 256 boards."""
    blob, bits, scalars_in = chess_full
    pick = C1_PICK
    s_ref, p_ref = chess_full_oracle
    model = capi.Model(blob=blob)
    eng = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F32)
    s, p = eng.eval_packed(bits, scalars_in)
    assert_f32(s[pick], s_ref, "scalars")
    assert_f32(p[pick], p_ref, "policy")
    # the same configuration through the split-f16 tower (three f16 MFMAs per product): the same 1e-4, all 256 boards
    # against the exact-f32 launch and the sample against the oracle
    split = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F32_SPLIT16)
    assert split.tower_path.startswith("tower_resident_split16")
    s2, p2 = split.eval_packed(bits, scalars_in)
    print(f"split16 20x256: vs oracle max |d| {max(np.abs(s2[pick] - s_ref).max(), np.abs(p2[pick] - p_ref).max()):.2e}; "
          f"vs exact f32 {max(np.abs(s2 - s).max(), np.abs(p2 - p).max()):.2e}")
    assert_f32(s2[pick], s_ref, "split16 scalars")
    assert_f32(p2[pick], p_ref, "split16 policy")
    assert_f32(s2, s, "split16 vs exact f32 scalars")
    assert_f32(p2, p, "split16 vs exact f32 policy")


def test_config_c1_f16_against_the_f32_accurate_launch_on_the_whole_batch(dev, chess_full):
    """BASELINE.json configs[2] at full size, ALL 256 boards: the oracle needs a second per board, but the split-f16
    launch is pinned to it at 1e-4 (test_config_c1_f32_vs_oracle_sample) and evaluates the whole batch in milliseconds,
    so it serves as the f32-accurate reference for every board of the f16 path."""
    blob, bits, scalars_in = chess_full
    model = capi.Model(blob=blob)
    ref = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F32_SPLIT16)
    s_ref, p_ref = ref.eval_packed(bits, scalars_in)
    eng = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F16)
    assert eng.tower_path == "tower_resident_f16+heads"
    s, p = eng.eval_packed(bits, scalars_in)
    rs = assert_f16(s, s_ref, "scalars, 256 boards")
    rp = assert_f16(p, p_ref, "policy, 256 boards")
    sm = np.abs(softmax(p) - softmax(p_ref)).max()
    print(f"chess 20x256 f16 vs f32-accurate, 256 boards: scalars rel {rs:.2e}, policy rel {rp:.2e}, max |dsoftmax| {sm:.2e}, "
          f"rms |dlogit| {np.sqrt(np.mean((p - p_ref) ** 2)):.2e}")
    assert sm <= F16_SOFTMAX_ATOL


# what the Rust shim runs by default (KZ_HIP_DECODE=device, kzero_amd/rust/hip.rs): legal-move probabilities, f16
# F16_PROB_ATOL: |delta p| of a softmax over 20-60 legal moves whose logits differ by <= F16_REL * scale (measured 6e-4)
F16_PROB_ATOL = 2e-3
F16_VALUE_ATOL = 5e-3


def _c1_move_lists(policy_len, batch, seed, finished=()):
    """Per-board legal-move lists of the size chess positions have (20-60 moves, distinct indices, available_moves()
    order = arbitrary order); `finished` boards have none (common.rs:77 `map_or(vec![], ..)`)."""
    rng = np.random.default_rng(seed)
    lists = [rng.permutation(policy_len)[:int(n)].astype(np.int32) for n in rng.integers(20, 61, size=batch)]
    for b in finished:
        lists[b] = np.zeros(0, np.int32)
    return lists


@pytest.mark.parametrize("dtype", [capi.KZ_DTYPE_F16, capi.KZ_DTYPE_F32_SPLIT16], ids=["f16", "split16"])
def test_config_c1_default_shim_entry_decoded_on_all_slots(dev, chess_full, chess_full_oracle, dtype):
    """The entry points `HipNetwork` uses by default — kz_engine_submit_packed_decoded on all four slots, then
    kz_engine_wait_decoded — on the flagship launches at BASELINE configs[2]'s size (chess 20x256, 256 boards, 20-60
    legal moves per board, one finished game per batch), against
      (a) the oracle's decode_output (common.rs:16-100) of the ORACLE's logits on all 256 boards, and
      (b) this library's eval_packed + the oracle's decode_output on the host, on all 256 boards;
    plus an out-of-range move index (the reference would panic on the slice index; here the wait fails) and a batch
    that does not fill the last workgroup."""
    blob, bits, scalars_in = chess_full
    s_ora, p_ora = chess_full_oracle
    net = O.OracleNet(blob)
    eng = capi.Engine(capi.Model(blob=blob), dev, 256, dtype)
    assert eng.tower_path == ("tower_resident_f16+heads" if dtype == capi.KZ_DTYPE_F16 else "tower_resident_split16+heads")
    s_own, p_own = eng.eval_packed(bits, scalars_in)
    # four different batches in flight: slot k evaluates the boards rolled by 64 k with its own move lists
    slots = []
    for k in range(capi.KZ_ENGINE_SLOTS):
        order = np.roll(np.arange(256), -64 * k)
        moves = _c1_move_lists(net.policy_len, 256, seed=100 + k, finished=(17 + k,))
        slots.append((order, moves, eng.submit_packed_decoded(k, bits[order], scalars_in[order], moves)))
    worst_p = worst_v = worst_own = 0.0
    for k in (2, 0, 3, 1):  # (any order of waits)
        order, moves, off = slots[k]
        v, probs = eng.wait_decoded(k, off)
        v_ora, probs_ora = O.decode_output(s_ora[order], p_ora[order], moves)
        v_own, probs_own = O.decode_output(s_own[order], p_own[order], moves)
        assert probs[17 + k].size == 0
        for b in range(256):
            assert probs[b].shape == moves[b].shape
            if moves[b].size:
                assert abs(float(probs[b].sum()) - 1.0) < 1e-5
                worst_p = max(worst_p, float(np.abs(probs[b] - probs_ora[b]).max()))
                worst_own = max(worst_own, float(np.abs(probs[b] - probs_own[b]).max()))
        worst_v = max(worst_v, float(np.abs(v[:, :4] - v_ora[:, :4]).max()))
        # moves_left is a raw network output (common.rs:61): the logit tolerance applies
        if dtype == capi.KZ_DTYPE_F16:
            assert_f16(v[:, 4:], v_ora[:, 4:], f"slot {k} moves_left")
        else:
            assert_f32(v[:, 4:], v_ora[:, 4:], f"slot {k} moves_left")
        worst_own = max(worst_own, float(np.abs(v - v_own).max()))
    print(f"[decoded {eng.tower_path}] vs oracle: max |dprob| {worst_p:.2e}, max |dvalue, dwdl| {worst_v:.2e}; "
          f"vs own logits decoded on the host: {worst_own:.2e}")
    if dtype == capi.KZ_DTYPE_F16:
        assert worst_p <= F16_PROB_ATOL and worst_v <= F16_VALUE_ATOL
    else:
        assert worst_p <= F32_ATOL and worst_v <= F32_ATOL
    assert worst_own <= 2e-6  # same logits, expf / tanhf on the device against the host's
    # a ragged batch: 37 boards (the last workgroup holds one board), every board its own list
    moves = _c1_move_lists(net.policy_len, 37, seed=7)
    v, probs = eng.wait_decoded(1, eng.submit_packed_decoded(1, bits[:37], scalars_in[:37], moves))
    v_own, probs_own = O.decode_output(s_own[:37], p_own[:37], moves)
    np.testing.assert_allclose(v, v_own, rtol=0, atol=2e-6)
    for a, b in zip(probs, probs_own):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)
    # an index outside the policy: the wait that returns the batch fails, the slot and the engine stay usable
    bad = [m.copy() for m in moves]
    bad[20][3] = net.policy_len
    off = eng.submit_packed_decoded(2, bits[:37], scalars_in[:37], bad)
    with pytest.raises(capi.KzError, match="strictly positive"):
        eng.wait_decoded(2, off)
    bad[20][3] = -1
    off = eng.submit_packed_decoded(2, bits[:37], scalars_in[:37], bad)
    with pytest.raises(capi.KzError, match="strictly positive"):
        eng.wait_decoded(2, off)
    v2, probs2 = eng.wait_decoded(2, eng.submit_packed_decoded(2, bits[:37], scalars_in[:37], moves))
    assert np.array_equal(v2, v) and all(np.array_equal(a, b) for a, b in zip(probs2, probs))
    # the undecoded entry points still serve the same engine
    s3, p3 = eng.wait_view(3, eng.submit_packed(3, bits, scalars_in))
    assert np.array_equal(s3, s_own) and np.array_equal(p3, p_own)


@pytest.mark.parametrize("game,depth,channels,head,dtype,path", [
    ("ataxx-7", 2, 128, "ataxx_conv", capi.KZ_DTYPE_F32, "tower_resident_f32+heads"),
    ("ataxx-7", 2, 128, "ataxx_conv", capi.KZ_DTYPE_F32_SPLIT16, "tower_resident_split16+heads"),
    ("ataxx-7", 2, 128, "ataxx_conv", capi.KZ_DTYPE_F16, "tower_resident_f16g+heads"),
    ("go-9", 2, 128, "conv", capi.KZ_DTYPE_F16, "tower_resident_f16g+heads"),
    ("go-9", 2, 128, "conv", capi.KZ_DTYPE_F32_SPLIT16, "tower_resident_split16+heads"),
    ("go-9", 2, 256, "conv", capi.KZ_DTYPE_F32_SPLIT16, "board_conv_split16"),
    ("ataxx-5", 2, 128, "ataxx_conv", capi.KZ_DTYPE_F16, "tower_resident_f16g+heads"),
    ("chess", 2, 256, "attention", capi.KZ_DTYPE_F16, "tower_resident_f16+heads"),
    ("chess", 2, 256, "attention", capi.KZ_DTYPE_F32_SPLIT16, "tower_resident_split16+heads"),
    ("go-19", 1, 64, "conv", capi.KZ_DTYPE_F16, "board_conv_f16"),           # (heads are separate launches: the stand-alone decode kernel)
    ("chess", 2, 128, "dense", capi.KZ_DTYPE_F16, "tower_resident_f16g"),
])
def test_decode_inside_the_launch_on_every_one_launch_path(dev, game, depth, channels, head, dtype, path):
    """decode_output (common.rs:16-100) as the last step of every "...+heads" launch (and, for the paths whose heads are
    separate launches, through kz_decode_output): submit_packed_decoded on all four slots against the oracle's decode_output
    of the engine's own logits — ragged batches (the last workgroup partly empty), move lists from none (a finished game)
    to every index of the policy twice over (longer than a wave: the 64-lane loops wrap many times), boards in another
    order on every slot."""
    blob = synth.random_model(game, depth, channels, head, seed=51)
    net = O.OracleNet(blob)
    n = 23
    bits, scalars_in = synth.random_boards(game, n, seed=52)
    eng = capi.Engine(capi.Model(blob=blob), dev, 256 if path == "board_conv_f16" else 32, dtype)
    assert eng.tower_path == path
    s_own, p_own = eng.eval_packed(bits, scalars_in)
    rng = np.random.default_rng(53)
    P = net.policy_len
    slots = []
    for k in range(capi.KZ_ENGINE_SLOTS):
        order = rng.permutation(n)[:n - 2 * k]  # 23, 21, 19, 17 boards
        moves = [rng.permutation(P)[:int(c)].astype(np.int32) for c in rng.integers(1, min(P, 90), size=len(order))]
        moves[1] = np.zeros(0, np.int32)                                   # a finished game
        moves[2] = np.concatenate([rng.permutation(P), rng.permutation(P)]).astype(np.int32)  # 2 P moves, every index twice
        moves[-1] = np.arange(P, dtype=np.int32)                           # the last board of a ragged batch
        slots.append((order, moves, eng.submit_packed_decoded(k, bits[order], scalars_in[order], moves)))
    for k in (1, 3, 0, 2):
        order, moves, off = slots[k]
        v, probs = eng.wait_decoded(k, off)
        v_ref, probs_ref = O.decode_output(s_own[order], p_own[order], moves)
        np.testing.assert_allclose(v, v_ref, rtol=0, atol=3e-6)
        for b in range(len(order)):
            assert probs[b].shape == moves[b].shape
            np.testing.assert_allclose(probs[b], probs_ref[b], rtol=2e-5, atol=1e-7)
    # an index out of range on the LAST board of a batch, then the engine again
    order, moves, _ = slots[0]
    bad = [m.copy() for m in moves]
    bad[-1][5] = P + 3
    off = eng.submit_packed_decoded(1, bits[order], scalars_in[order], bad)
    with pytest.raises(capi.KzError, match="strictly positive"):
        eng.wait_decoded(1, off)
    v, probs = eng.wait_decoded(1, eng.submit_packed_decoded(1, bits[order], scalars_in[order], moves))
    v_ref, _ = O.decode_output(s_own[order], p_own[order], moves)
    np.testing.assert_allclose(v, v_ref, rtol=0, atol=3e-6)
    s2, p2 = eng.eval_packed(bits, scalars_in)
    assert np.array_equal(s2, s_own) and np.array_equal(p2, p_own)


G8_PICK = np.arange(0, 512, 32)  # 16 boards spread over the batch (rounds 2-3: 8)


@pytest.fixture(scope="module")
def go19_full():
    return synth.random_model("go-19", 40, 256, "conv", seed=33), *synth.random_boards("go-19", 512, seed=34)


@pytest.fixture(scope="module")
def go19_full_oracle(go19_full):
    """The REAL oracle on 16 of the 512 boards of the G8 network (~10 s of one core per board; the boards run on separate
    host threads).  Shared by the f16 and the split16 test."""
    blob, bits, scalars_in = go19_full
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits[G8_PICK], scalars_in[G8_PICK], net.n_scalar, net.n_bool, net.h, net.w)
    return net.forward(dense, threads=os.cpu_count() or 1)


def test_config_g8_go19_40x256_at_executor_batch_512(dev, go19_full, go19_full_oracle):
    """BASELINE.json configs[4] at its stated size on one GPU: Go 19x19, 40 blocks x 256 channels, f16, executor batch
    512 through kz_board_conv_f16.  Size-independent properties on all 512 boards (determinism, permutation
    equivariance over the batch, batch-size invariance), the exact-f32 path of this library on a 64-board sample, and
    the real oracle on 16 boards."""
    blob, bits, scalars_in = go19_full
    model = capi.Model(blob=blob)
    f16 = capi.Engine(model, dev, 512, capi.KZ_DTYPE_F16)
    assert f16.tower_path == "board_conv_f16" and f16.max_batch == 512
    s, p = f16.eval_packed(bits, scalars_in)
    assert s.shape == (512, 5) and p.shape == (512, 362)
    assert np.isfinite(s).all() and np.isfinite(p).all()
    s2, p2 = f16.eval_packed(bits, scalars_in)
    assert np.array_equal(s, s2) and np.array_equal(p, p2), "not deterministic"
    perm = np.random.default_rng(6).permutation(512)
    sp, pp = f16.eval_packed(bits[perm], scalars_in[perm])
    assert np.array_equal(sp, s[perm]) and np.array_equal(pp, p[perm]), "not permutation equivariant"
    for lo, n in ((0, 7), (300, 64), (511, 1)):
        sn, pn = f16.eval_packed(bits[lo:lo + n], scalars_in[lo:lo + n])
        assert np.array_equal(sn, s[lo:lo + n]) and np.array_equal(pn, p[lo:lo + n]), "result depends on the batch size"
    sample = np.arange(0, 512, 8)
    f32 = capi.Engine(model, dev, 64, capi.KZ_DTYPE_F32)
    s_ref, p_ref = f32.eval_packed(bits[sample], scalars_in[sample])
    rs = assert_f16(s[sample], s_ref, "scalars, 64 of 512 boards vs exact f32", rms_tol=F16_RMS_DEEP)
    rp = assert_f16(p[sample], p_ref, "policy, 64 of 512 boards vs exact f32", rms_tol=F16_RMS_DEEP)
    sm = float(np.abs(softmax(p[sample]) - softmax(p_ref)).max())
    print(f"go-19 40x256 B=512 f16 vs exact f32 on 64 boards: scalars rel {rs:.2e}, policy rel {rp:.2e}, max |dsoftmax| {sm:.2e}")
    assert sm <= F16_SOFTMAX_ATOL, f"G8 f16: post-softmax max |delta p| = {sm:.3e} > {F16_SOFTMAX_ATOL}"
    so, po = go19_full_oracle
    assert np.array_equal(sample[::4], G8_PICK)  # boards 0, 32, ..., 480
    assert_f32(s_ref[::4], so, "exact f32 vs oracle, scalars")
    assert_f32(p_ref[::4], po, "exact f32 vs oracle, policy")
    assert_f16(s[G8_PICK], so, "f16 vs oracle, scalars", rms_tol=F16_RMS_DEEP)
    assert_f16(p[G8_PICK], po, "f16 vs oracle, policy", rms_tol=F16_RMS_DEEP)


def test_config_c1_trained_like_activation_scale(dev):
    """Every other full-size test runs PyTorch-default random weights, whose residual stream stays within a few tens.  A
    trained tower's grows: here every block's second BatchNorm weight is scaled by 3, so the stream reaches ~170 by block 20
    and the logits ~1.8e3 (oracle trace, build container).  The f16 path must stay inside its stated tolerance RELATIVE
    to the output scale, the split-f16 path inside 1e-4 relative, and the +-65504 range check must stay silent."""
    blob = synth.random_model("chess", 20, 256, "attention", seed=21, block_gain=3.0)
    bits, scalars_in = synth.random_boards("chess", 256, seed=22)
    pick = np.array([0, 1, 62, 63, 64, 127, 200, 255])
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits[pick], scalars_in[pick], net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    scale_p, scale_s = float(np.abs(p_ref).max()), float(np.abs(s_ref).max())
    assert scale_p > 300.0, f"the fixture no longer stresses the range: logit scale {scale_p}"
    model = capi.Model(blob=blob)
    split = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F32_SPLIT16)
    s2, p2 = split.eval_packed(bits, scalars_in)  # (a non-finite activation would raise here)
    rel_p = float(np.abs(p2[pick] - p_ref).max()) / scale_p
    rel_s = float(np.abs(s2[pick] - s_ref).max()) / max(1.0, scale_s)
    f16 = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F16)
    s, p = f16.eval_packed(bits, scalars_in)
    assert np.isfinite(s).all() and np.isfinite(p).all()
    r16s = assert_f16(s[pick], s_ref, "trained-like scale, f16 scalars vs oracle")
    r16p = assert_f16(p[pick], p_ref, "trained-like scale, f16 policy vs oracle")
    assert_f16(s, s2, "trained-like scale, f16 scalars vs split16, 256 boards")
    assert_f16(p, p2, "trained-like scale, f16 policy vs split16, 256 boards")
    print(f"trained-like 20x256: logit scale {scale_p:.0f}, scalar scale {scale_s:.1f}; split16 rel {rel_p:.2e} / {rel_s:.2e}; "
          f"f16 rel {r16p:.2e} / {r16s:.2e}")
    assert rel_p <= F32_ATOL and rel_s <= F32_ATOL, (rel_p, rel_s)


@pytest.mark.parametrize("game,hs,nb", [("ataxx-5", 72, 4), ("ataxx-6", 96, 3)])
def test_fused_f32_heads_with_more_hidden_units_than_threads(dev, game, hs, nb):
    """The fused exact-f32 heads with boards-per-workgroup x scalar_hidden_size > 256 threads (three 6x6 boards with 96
    hidden units, four 5x5 boards with 72): every hidden unit of every board of a workgroup must be reduced."""
    blob = synth.random_model(game, 2, 128, "ataxx_conv", seed=91, scalar_hidden_size=hs)
    bits, scalars_in = synth.random_boards(game, 4 * nb + 1, seed=92)
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    eng = capi.Engine(capi.Model(blob=blob), dev, 64, capi.KZ_DTYPE_F32)
    assert eng.tower_path == "tower_resident_f32+heads" and eng.launch_geometry(4 * nb + 1) == (5, nb)
    assert nb * hs > 256
    s, p = eng.eval_packed(bits, scalars_in)
    assert_f32(s, s_ref, "scalars")
    assert_f32(p, p_ref, "policy")


def _scaled_stem(blob, factor):
    """The same network with its stem convolution scaled: the residual stream grows by `factor`."""
    from kzero_amd.model_file import write_model
    meta, tensors = read_model(blob)
    tensors = dict(tensors)
    tensors["common.tower.0.weight"] = tensors["common.tower.0.weight"] * np.float32(factor)
    tensors["common.tower.0.bias"] = tensors["common.tower.0.bias"] * np.float32(factor)
    return write_model(meta, tensors)


@pytest.mark.parametrize("game,depth,channels,head,dtype,path,env", [
    ("chess", 2, 256, "attention", "f16", "tower_resident_f16+heads", {}),
    ("chess", 2, 256, "attention", "f16", "tower_resident_f16", {"KZ_NO_FUSED_HEADS": "1"}),
    ("chess", 2, 256, "attention", "split16", "tower_resident_split16+heads", {}),
    ("chess", 2, 256, "attention", "split16", "tower_resident_split16", {"KZ_NO_FUSED_HEADS": "1"}),
    ("ataxx-7", 2, 128, "ataxx_conv", "f16", "tower_resident_f16g+heads", {}),
    ("ataxx-7", 2, 128, "ataxx_conv", "f16", "tower_resident_f16g", {"KZ_NO_FUSED_HEADS": "1"}),
    ("go-9", 2, 128, "conv", "f16", "board_conv_f16", {"KZ_NO_RESIDENT_F16G": "1"}),
    ("go-9", 2, 128, "conv", "f16", "conv_igemm_f16", {"KZ_NO_RESIDENT_F16G": "1", "KZ_NO_BOARD_CONV": "1"}),
    ("go-19", 2, 128, "conv", "split16", "board_conv_split16", {}),
])
def test_f16_range_overflow_is_reported(dev, game, depth, channels, head, dtype, path, env):
    """f16 storage overflows beyond +-65504 (include/kz_hip.h, KZ_DTYPE_F16): a network whose residual stream leaves that
    range must not come back as silent inf/NaN with rc 0.  Every f16 / split-f16 path reports it on the call that
    returns the batch; the same network through KZ_DTYPE_F32 evaluates normally; the engine stays usable."""
    base = synth.random_model(game, depth, channels, head, seed=17)
    big = _scaled_stem(base, 3.0e5)  # stem outputs of order 1e5 > 65504
    bits, scalars_in = synth.random_boards(game, 9, seed=18)
    code = capi.KZ_DTYPE_F16 if dtype == "f16" else capi.KZ_DTYPE_F32_SPLIT16
    os.environ.update(env)
    try:
        eng_big = capi.Engine(capi.Model(blob=big), dev, 512, code)
        eng_ok = capi.Engine(capi.Model(blob=base), dev, 512, code)
    finally:
        for k in env:
            del os.environ[k]
    assert eng_big.tower_path == path
    s_ok, p_ok = eng_ok.eval_packed(bits, scalars_in)  # in range: no error
    assert np.isfinite(s_ok).all() and np.isfinite(p_ok).all()
    with pytest.raises(capi.KzError, match="non-finite activation"):
        eng_big.eval_packed(bits, scalars_in)
    with pytest.raises(capi.KzError, match="non-finite activation"):  # the asynchronous pair reports it at the wait
        eng_big.wait_view(0, eng_big.submit_packed(0, bits, scalars_in))
    n = eng_ok.submit_packed(1, bits, scalars_in)  # other engines and later batches are unaffected
    s_again, _ = eng_ok.wait_view(1, n)
    assert np.array_equal(s_again, s_ok)
    # device-resident entry points: kz_engine_synchronize reports it, once
    d_s, d_p = capi.DeviceBuffer(dev, 9 * 5 * 4), capi.DeviceBuffer(dev, 9 * eng_big.model.info.policy_len * 4)
    eng_big.enqueue_packed_device(capi.DeviceBuffer.from_host(dev, bits), bits.shape[1],
                                  capi.DeviceBuffer.from_host(dev, scalars_in), 9, d_s, d_p)
    with pytest.raises(capi.KzError, match="non-finite activation"):
        eng_big.synchronize()
    eng_big.synchronize()
    # the exact-f32 path has no such limit
    f32 = capi.Engine(capi.Model(blob=big), dev, 16, capi.KZ_DTYPE_F32)
    s32, p32 = f32.eval_packed(bits, scalars_in)
    assert np.isfinite(s32).all() and np.isfinite(p32).all()


def test_full_size_properties(dev, chess_full):
    """Size-independent properties at the full configuration: permutation equivariance over the batch,
    batch-size invariance, determinism, and agreement of the board-resident tower with the generic per-layer path."""
    blob, bits, scalars_in = chess_full
    model = capi.Model(blob=blob)
    eng = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F16)
    s, p = eng.eval_packed(bits, scalars_in)
    assert np.isfinite(s).all() and np.isfinite(p).all()
    s2, p2 = eng.eval_packed(bits, scalars_in)
    assert np.array_equal(s, s2) and np.array_equal(p, p2), "not deterministic"
    perm = np.random.default_rng(5).permutation(256)
    sp, pp = eng.eval_packed(bits[perm], scalars_in[perm])
    assert np.array_equal(sp, s[perm]) and np.array_equal(pp, p[perm]), "not permutation equivariant"
    s7, p7 = eng.eval_packed(bits[:7], scalars_in[:7])
    assert np.array_equal(s7, s[:7]) and np.array_equal(p7, p[:7]), "result depends on the batch size"
    if eng.tower_path.startswith("tower_resident_f16"):
        os.environ["KZ_FORCE_GENERIC"] = "1"
        try:
            gen = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F16)
        finally:
            del os.environ["KZ_FORCE_GENERIC"]
        assert gen.tower_path in ("board_conv_f16", "conv_igemm_f16")
        sg, pg = gen.eval_packed(bits, scalars_in)
        # both are f16-storage/f32-accumulate; they differ only in summation order
        assert_f16_paths_deep(sg, s, "resident vs per-layer f16, scalars")
        assert_f16_paths_deep(pg, p, "resident vs per-layer f16, policy")


def test_fused_heads_match_separate_head_kernels(dev, chess_full):
    """The heads fused behind the resident tower (one launch per batch) vs the same tower followed by the generic
    head kernels: both read the same f16 tower output, so they agree to f16-rounding of the 1x1 convolution outputs."""
    blob, bits, scalars_in = chess_full
    model = capi.Model(blob=blob)
    fused = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F16)
    os.environ["KZ_NO_FUSED_HEADS"] = "1"
    try:
        split = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F16)
    finally:
        del os.environ["KZ_NO_FUSED_HEADS"]
    for n in (256, 255, 1):  # odd batches exercise the half-empty last workgroup
        s1, p1 = fused.eval_packed(bits[:n], scalars_in[:n])
        s2, p2 = split.eval_packed(bits[:n], scalars_in[:n])
        assert np.abs(s1 - s2).max() < 2e-3, np.abs(s1 - s2).max()
        assert np.abs(p1 - p2).max() < 1e-2, np.abs(p1 - p2).max()


def test_dense_and_packed_inputs_agree_on_the_resident_path(dev):
    """kz_engine_eval_dense (bit-compatible with CudaExecutor::evaluate: f32 NCHW planes) and kz_engine_eval_packed
    (bits + scalars, encode fused into the launch) must give identical results: same planes, same kernel."""
    blob = synth.random_model("chess", 2, 256, "attention", seed=51)
    bits, scalars_in = synth.random_boards("chess", 37, seed=52)
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    eng = capi.Engine(capi.Model(blob=blob), dev, 64, capi.KZ_DTYPE_F16)
    assert eng.tower_path == "tower_resident_f16+heads"
    s1, p1 = eng.eval_packed(bits, scalars_in)
    s2, p2 = eng.eval_dense(dense)
    assert np.array_equal(s1, s2) and np.array_equal(p1, p2)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    assert_f16(s1, s_ref, "scalars")
    assert_f16(p1, p_ref, "policy")


def test_large_batches_and_engine_churn(dev):
    """Maximum sizes: a 2048-board executor batch (1024 resident workgroups, 4 per CU queued) gives, row for row, the
    results of small batches; engines can be created and destroyed repeatedly without disturbing others."""
    blob = synth.random_model("chess", 2, 256, "attention", seed=61)
    bits, scalars_in = synth.random_boards("chess", 2048, seed=62)
    model = capi.Model(blob=blob)
    big = capi.Engine(model, dev, 2048, capi.KZ_DTYPE_F16)
    s, p = big.eval_packed(bits, scalars_in)
    assert np.isfinite(p).all() and np.isfinite(s).all()
    small = capi.Engine(model, dev, 64, capi.KZ_DTYPE_F16)
    for lo in (0, 1000, 1984):
        s2, p2 = small.eval_packed(bits[lo:lo + 64], scalars_in[lo:lo + 64])
        assert np.array_equal(s2, s[lo:lo + 64]) and np.array_equal(p2, p[lo:lo + 64])
    for i in range(5):
        tmp = capi.Engine(model, dev, 32 + i, capi.KZ_DTYPE_F16 if i % 2 else capi.KZ_DTYPE_F32)
        tmp.eval_packed(bits[:3], scalars_in[:3])
        tmp.close()
    s3, _ = small.eval_packed(bits[:64], scalars_in[:64])
    assert np.array_equal(s3, s[:64])


def test_go19_generic_path_vs_oracle(dev):
    """Large board (19x19, 13 input planes, conv head + pass move): the per-layer implicit-GEMM path."""
    blob = synth.random_model("go-19", 3, 64, "conv", seed=31)
    bits, scalars_in = synth.random_boards("go-19", 5, seed=32)
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    eng = capi.Engine(capi.Model(blob=blob), dev, 8, capi.KZ_DTYPE_F32)
    s, p = eng.eval_packed(bits, scalars_in)
    assert_f32(s, s_ref, "scalars")
    assert_f32(p, p_ref, "policy")


@pytest.mark.parametrize("game,depth,channels,batch", [("go-19", 2, 128, 5), ("go-19", 3, 64, 9), ("go-19", 1, 256, 3),
                                                       ("go-9", 2, 256, 11)])  # 256 channels on 81 squares: four boards per workgroup
def test_board_conv_split16_vs_oracle(dev, game, depth, channels, batch):
    """KZ_DTYPE_F32_SPLIT16 on a board the one-launch split tower cannot hold: one kz_board_conv_split16 launch per layer
    ((hi, lo) f16 rows in HBM, three MFMAs per product), stem in exact f32, f32 heads.  The north_star's 1e-4 against the
    oracle, and against the exact-f32 path of this library; odd batches leave the last workgroups ragged."""
    blob = synth.random_model(game, depth, channels, "conv", seed=51)
    bits, scalars_in = synth.random_boards(game, batch, seed=52)
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    model = capi.Model(blob=blob)
    assert model.supports_dtype(capi.KZ_DTYPE_F32_SPLIT16)
    eng = capi.Engine(model, dev, 16, capi.KZ_DTYPE_F32_SPLIT16)
    assert eng.tower_path == "board_conv_split16"
    s, p = eng.eval_packed(bits, scalars_in)
    print(f"board conv split16 {game} {depth}x{channels}: max |d scalars| {np.abs(s - s_ref).max():.2e}, "
          f"max |d policy| {np.abs(p - p_ref).max():.2e}")
    assert_f32(s, s_ref, "scalars vs oracle")
    assert_f32(p, p_ref, "policy vs oracle")
    f32 = capi.Engine(model, dev, 16, capi.KZ_DTYPE_F32)
    s32, p32 = f32.eval_packed(bits, scalars_in)
    assert_f32(s, s32, "scalars vs exact f32")
    assert_f32(p, p32, "policy vs exact f32")
    s1, p1 = eng.eval_packed(bits[:1], scalars_in[:1])
    assert np.array_equal(s1, s[:1]) and np.array_equal(p1, p[:1]), "result depends on the batch size"


def test_board_conv_split16_trained_like_scale(dev):
    """The split board-tile path on a residual stream that grows like a trained tower's (every block's second BatchNorm
    weight x 6, eight blocks: the stream grows by orders of magnitude, the logits reach ~10 behind the final BatchNorm): 1e-4
    RELATIVE to the output scale against the oracle, the range check silent."""
    blob = synth.random_model("go-19", 8, 64, "conv", seed=61, block_gain=6.0)
    bits, scalars_in = synth.random_boards("go-19", 4, seed=62)
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    scale_p, scale_s = max(1.0, float(np.abs(p_ref).max())), max(1.0, float(np.abs(s_ref).max()))
    assert scale_p > 5.0, f"the fixture no longer stresses the range: logit scale {scale_p}"
    eng = capi.Engine(capi.Model(blob=blob), dev, 8, capi.KZ_DTYPE_F32_SPLIT16)
    assert eng.tower_path == "board_conv_split16"
    s, p = eng.eval_packed(bits, scalars_in)  # (a non-finite activation would raise here)
    rel_p, rel_s = float(np.abs(p - p_ref).max()) / scale_p, float(np.abs(s - s_ref).max()) / scale_s
    print(f"go-19 8x64 gain 6: logit scale {scale_p:.0f}, split16 rel err policy {rel_p:.2e}, scalars {rel_s:.2e}")
    assert rel_p <= 1e-4 and rel_s <= 1e-4


def test_go19_40x256_split16_at_executor_batch_512(dev, go19_full, go19_full_oracle):
    """The G8 network at its executor batch in the arithmetic the Rust binding defaults to (KZ_HIP_DTYPE=parity):
    83 per-layer launches in split arithmetic (the stem included since round 4).  All 512 boards: finite, deterministic,
    batch-size invariant; 1e-4 against the exact-f32 path of this library on a 32-board sample and against the real oracle
    on 16 boards."""
    blob, bits, scalars_in = go19_full
    model = capi.Model(blob=blob)
    eng = capi.Engine(model, dev, 512, capi.KZ_DTYPE_F32_SPLIT16)
    assert eng.tower_path == "board_conv_split16" and eng.max_batch == 512
    s, p = eng.eval_packed(bits, scalars_in)
    assert s.shape == (512, 5) and p.shape == (512, 362)
    assert np.isfinite(s).all() and np.isfinite(p).all()
    s2, p2 = eng.eval_packed(bits, scalars_in)
    assert np.array_equal(s, s2) and np.array_equal(p, p2), "not deterministic"
    for lo, n in ((0, 7), (300, 64), (511, 1)):
        sn, pn = eng.eval_packed(bits[lo:lo + n], scalars_in[lo:lo + n])
        assert np.array_equal(sn, s[lo:lo + n]) and np.array_equal(pn, p[lo:lo + n]), "result depends on the batch size"
    sample = np.arange(0, 512, 16)
    f32 = capi.Engine(model, dev, 32, capi.KZ_DTYPE_F32)
    s_ref, p_ref = f32.eval_packed(bits[sample], scalars_in[sample])
    print(f"go-19 40x256 B=512 split16 vs exact f32 on 32 boards: max |d scalars| {np.abs(s[sample] - s_ref).max():.2e}, "
          f"max |d policy| {np.abs(p[sample] - p_ref).max():.2e}")
    assert_f32(s[sample], s_ref, "scalars, 32 of 512 boards vs exact f32")
    assert_f32(p[sample], p_ref, "policy, 32 of 512 boards vs exact f32")
    so, po = go19_full_oracle
    assert np.array_equal(sample[::2], G8_PICK)
    assert_f32(s[G8_PICK], so, "split16 vs oracle, scalars")
    assert_f32(p[G8_PICK], po, "split16 vs oracle, policy")


@pytest.mark.parametrize("game,depth,batch", [("go-19", 2, 5), ("go-9", 2, 11), ("ataxx-7", 2, 13)])
def test_board_conv_path_vs_oracle(dev, game, depth, batch):
    """f16 per-layer path with whole boards as LDS-resident spatial tiles (kz_board_conv_f16): 128-channel towers on
    boards of 1 (go-19), 4 (go-9) and 6 (ataxx-7) boards per workgroup, ragged last workgroup included; and the same
    engine with that kernel disabled (generic implicit GEMM) must agree with it."""
    head = "ataxx_conv" if game.startswith("ataxx") else "conv"
    blob = synth.random_model(game, depth, 128, head, seed=41)
    bits, scalars_in = synth.random_boards(game, batch, seed=42)
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    model = capi.Model(blob=blob)
    os.environ["KZ_NO_RESIDENT_F16G"] = "1"  # (small boards at 128 channels otherwise take the one-launch f16 tower)
    try:
        eng = capi.Engine(model, dev, 512, capi.KZ_DTYPE_F16)
    finally:
        del os.environ["KZ_NO_RESIDENT_F16G"]
    assert eng.tower_path == "board_conv_f16"
    s, p = eng.eval_packed(bits, scalars_in)
    assert_f16(s, s_ref, "scalars")
    assert_f16(p, p_ref, "policy")
    os.environ["KZ_NO_BOARD_CONV"] = "1"
    os.environ["KZ_NO_RESIDENT_F16G"] = "1"
    try:
        gen = capi.Engine(model, dev, 512, capi.KZ_DTYPE_F16)
    finally:
        del os.environ["KZ_NO_BOARD_CONV"], os.environ["KZ_NO_RESIDENT_F16G"]
    assert gen.tower_path == "conv_igemm_f16"
    sg, pg = gen.eval_packed(bits, scalars_in)
    # same arithmetic (f16 operands, f32 accumulate, f16 activations), different summation order: a far tighter bound
    # than the f16-vs-oracle tolerance, tight enough to catch a wrong weight fragment (which the oracle bound does not
    # on a 2-block net)
    ds, dp = np.abs(sg - s).max(), np.abs(pg - p).max()
    print(f"board conv vs implicit GEMM f16: max |d scalars| {ds:.2e}, max |d policy| {dp:.2e}")
    assert ds < F16_PATHS_ATOL and dp < F16_PATHS_ATOL


@pytest.mark.parametrize("game,depth,channels,head,batches", [
    ("ataxx-7", 3, 128, "ataxx_conv", (1, 2, 13)),   # two boards per workgroup, ragged last workgroup
    ("chess", 2, 256, "attention", (1, 5)),          # one board per workgroup, 64 rows
    ("chess", 2, 128, "attention", (3,)),            # 128 channels on an 8x8 board: one board per workgroup
    ("go-9", 2, 128, "conv", (3,)),                  # 81 pixels: six tiles
])
def test_resident_f32_tower_vs_oracle(dev, game, depth, channels, head, batches):
    """Exact-f32 board-resident launch (kz_tower_resident_f32): <= 1e-4 against the oracle for every supported
    shape, and the per-layer implicit-GEMM path of the same engine (KZ_FORCE_GENERIC) agrees with it."""
    blob = synth.random_model(game, depth, channels, head, seed=71)
    net = O.OracleNet(blob)
    model = capi.Model(blob=blob)
    eng = capi.Engine(model, dev, 16, capi.KZ_DTYPE_F32)
    # conv policy heads (and the scalar head) ride in the same launch; the attention head keeps its own kernels
    assert eng.tower_path == ("tower_resident_f32+heads" if head in ("ataxx_conv", "conv") else "tower_resident_f32")
    os.environ["KZ_NO_FUSED_HEADS"] = "1"
    try:
        unfused = capi.Engine(model, dev, 16, capi.KZ_DTYPE_F32)
    finally:
        del os.environ["KZ_NO_FUSED_HEADS"]
    assert unfused.tower_path == "tower_resident_f32"
    os.environ["KZ_FORCE_GENERIC"] = "1"
    try:
        gen = capi.Engine(model, dev, 16, capi.KZ_DTYPE_F32)
    finally:
        del os.environ["KZ_FORCE_GENERIC"]
    assert gen.tower_path == "conv_igemm_f32"
    for batch in batches:
        bits, scalars_in = synth.random_boards(game, batch, seed=72 + batch)
        dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
        s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
        s, p = eng.eval_packed(bits, scalars_in)
        assert_f32(s, s_ref, f"scalars b={batch}")
        assert_f32(p, p_ref, f"policy b={batch}")
        sd, pd = eng.eval_dense(dense)
        assert np.array_equal(sd, s) and np.array_equal(pd, p)  # dense and packed entry points: same launch
        sg, pg = gen.eval_packed(bits, scalars_in)
        assert_f32(sg, s, f"generic vs resident scalars b={batch}")
        assert_f32(pg, p, f"generic vs resident policy b={batch}")
        su, pu = unfused.eval_packed(bits, scalars_in)
        assert_f32(su, s, f"separate head launches vs one launch, scalars b={batch}")
        assert_f32(pu, p, f"separate head launches vs one launch, policy b={batch}")
        # the async pair (on the one-launch path: zero-copy staging, slots alternating over two streams)
        for slot in range(capi.KZ_ENGINE_SLOTS):
            eng.submit_packed(slot, bits, scalars_in)
        for slot in range(capi.KZ_ENGINE_SLOTS):
            sa, pa = eng.wait(slot, batch)
            assert np.array_equal(sa, s) and np.array_equal(pa, p), f"slot {slot} b={batch}"


def test_one_launch_f32_network_decode_and_range_check(dev):
    """The exact-f32 launch with the heads inside (tower_resident_f32+heads, BASELINE configs[1]'s network): the
    device-side decode_output entry points (rust/kz-core/src/network/common.rs:16-100) against the oracle, and the range
    check — a non-finite input plane reaches every output, which must come back as an error on the call that returns
    the batch, on the synchronous, the asynchronous and the device-resident entry points; the engine stays usable."""
    game = "ataxx-7"
    blob = synth.random_model(game, 8, 128, "ataxx_conv", seed=91)
    net = O.OracleNet(blob)
    eng = capi.Engine(capi.Model(blob=blob), dev, 64, capi.KZ_DTYPE_F32)
    assert eng.tower_path == "tower_resident_f32+heads"
    bits, scalars_in = synth.random_boards(game, 37, seed=92)  # 19 workgroups, the last one with one board
    s_ref, p_ref = net.forward(O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w),
                               threads=os.cpu_count() or 1)
    rng = np.random.default_rng(93)
    moves = [rng.permutation(net.policy_len)[:int(n)].astype(np.int32) for n in rng.integers(0, 60, size=37)]
    v_ref, probs_ref = O.decode_output(s_ref, p_ref, moves)
    v, probs = eng.eval_packed_decoded(bits, scalars_in, moves)
    np.testing.assert_allclose(v, v_ref, rtol=1e-4, atol=1e-5)
    for a, b in zip(probs, probs_ref):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6)
    offs = [eng.submit_packed_decoded(slot, bits, scalars_in, moves) for slot in range(capi.KZ_ENGINE_SLOTS)]
    for slot, off in enumerate(offs):
        v2, probs2 = eng.wait_decoded(slot, off)
        assert np.array_equal(v2, v) and all(np.array_equal(a, b) for a, b in zip(probs2, probs))
    s_ok, p_ok = eng.eval_packed(bits, scalars_in)
    assert_f32(s_ok, s_ref, "scalars")
    assert_f32(p_ok, p_ref, "policy")
    if net.n_scalar == 0:
        return
    bad = scalars_in.copy()
    bad[5, 0] = np.inf  # one board of the batch
    with pytest.raises(capi.KzError, match="non-finite activation"):
        eng.eval_packed(bits, bad)
    with pytest.raises(capi.KzError, match="non-finite activation"):
        eng.wait_view(0, eng.submit_packed(0, bits, bad))
    d_s, d_p = capi.DeviceBuffer(dev, 37 * 5 * 4), capi.DeviceBuffer(dev, 37 * net.policy_len * 4)
    eng.enqueue_packed_device(capi.DeviceBuffer.from_host(dev, bits), bits.shape[1],
                              capi.DeviceBuffer.from_host(dev, bad), 37, d_s, d_p)
    with pytest.raises(capi.KzError, match="non-finite activation"):
        eng.synchronize()
    eng.synchronize()
    s_again, p_again = eng.eval_packed(bits, scalars_in)
    assert np.array_equal(s_again, s_ok) and np.array_equal(p_again, p_ok)


@pytest.mark.parametrize("game,depth,channels,head,batches", [
    ("chess", 1, 256, "attention", (1, 3)),           # one board per workgroup
    ("chess", 3, 256, "attention", (5, 17)),
    ("ataxx-7", 8, 128, "ataxx_conv", (1, 2, 13)),    # BASELINE configs[1]'s network: two boards per workgroup, ragged
    ("chess", 2, 128, "attention", (3,)),             # 128 channels on an 8x8 board
    ("go-9", 2, 128, "conv", (3,)),                   # 81 pixels: six tiles; conv head with the extra (pass) move
    ("go-9", 3, 128, "conv", (1, 4)),
    ("ataxx-5", 2, 128, "ataxx_conv", (7,)),          # 25 pixels: four boards per workgroup
    ("ataxx-7", 4, 64, "ataxx_conv", (5,)),           # BASELINE configs[0]'s network (64 channels)
    ("chess", 2, 32, "dense", (4,)),                  # 32 channels: widened to 64 by zero filters (round 4)
    ("chess", 0, 64, "attention", (1,)),              # no block: not a shape of the kernel
])
def test_split16_tower_vs_oracle(dev, game, depth, channels, head, batches):
    """KZ_DTYPE_F32_SPLIT16 (kz_tower_resident_split: (hi, lo) f16 pairs, three MFMAs per product): the SAME <= 1e-4
    against the oracle as the exact-f32 path, which it also agrees with; shapes it does not take are refused."""
    if depth == 0:
        with pytest.raises(capi.KzError, match="SPLIT16 needs"):
            capi.Engine(capi.Model(blob=synth.random_model(game, 0, channels, head)), dev, 4, capi.KZ_DTYPE_F32_SPLIT16)
        return
    if head == "dense":  # the golden chess net with the dense head, 32 channels: runs as a 64-channel tower
        blob = O.load_blob("chess_2x32_dense_h")
        net = O.OracleNet(blob)
        eng = capi.Engine(capi.Model(blob=blob), dev, 4, capi.KZ_DTYPE_F32_SPLIT16)
        assert eng.tower_path == "tower_resident_split16"
        _, s_gold, p_gold = O.read_io("chess_2x32_dense_h", "planes", net.c_in, net.h, net.w, net.policy_len)
        bits, scalars_in = O.read_packed("chess_2x32_dense_h", net.n_bool, net.n_scalar, net.h, net.w)
        s, p = eng.eval_packed(bits, scalars_in)
        assert_f32(s, s_gold, "scalars vs the reference's golden outputs")
        assert_f32(p, p_gold, "policy vs the reference's golden outputs")
        return
    blob = synth.random_model(game, depth, channels, head, seed=81 + depth)
    net = O.OracleNet(blob)
    model = capi.Model(blob=blob)
    eng = capi.Engine(model, dev, 32, capi.KZ_DTYPE_F32_SPLIT16)
    # ONE launch (tower + scalar head + policy head): the chess attention network at 256 channels in split arithmetic, and
    # since round 3 the conv-policy networks at 128 / 256 channels (the policy head's hidden layer as one more pass of the
    # split stream, then the exact-f32 launch's tail on f32 copies of the images)
    fused = (game == "chess" and channels == 256 and head == "attention") or (head in ("ataxx_conv", "conv") and channels >= 128)
    assert eng.tower_path == ("tower_resident_split16+heads" if fused else "tower_resident_split16")
    unfused = None
    if fused:  # the same tower launch followed by the separate f32 head kernels
        os.environ["KZ_NO_FUSED_HEADS"] = "1"
        try:
            unfused = capi.Engine(model, dev, 32, capi.KZ_DTYPE_F32_SPLIT16)
        finally:
            del os.environ["KZ_NO_FUSED_HEADS"]
        assert unfused.tower_path == "tower_resident_split16"
    exact = capi.Engine(model, dev, 32, capi.KZ_DTYPE_F32)
    assert exact.tower_path == ("conv_igemm_f32" if channels < 128 else
                                "tower_resident_f32+heads" if head in ("ataxx_conv", "conv") else "tower_resident_f32")
    for batch in batches:
        bits, scalars_in = synth.random_boards(game, batch, seed=82 + batch)
        dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
        s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
        s, p = eng.eval_packed(bits, scalars_in)
        print(f"split16 {game} {depth}x{channels} b={batch}: max |d scalars| {np.abs(s - s_ref).max():.2e}, "
              f"max |d policy| {np.abs(p - p_ref).max():.2e}")
        assert_f32(s, s_ref, f"scalars b={batch}")
        assert_f32(p, p_ref, f"policy b={batch}")
        se, pe = exact.eval_packed(bits, scalars_in)
        assert_f32(se, s, f"exact f32 vs split16 scalars b={batch}")
        assert_f32(pe, p, f"exact f32 vs split16 policy b={batch}")
        if unfused is not None:
            su, pu = unfused.eval_packed(bits, scalars_in)
            assert_f32(su, s_ref, f"split16 with separate heads, scalars b={batch}")
            assert_f32(pu, p_ref, f"split16 with separate heads, policy b={batch}")
            dense_in = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
            sd, pd = eng.eval_dense(dense_in)  # the dense entry point goes through the same fused launch
            assert_f32(sd, s_ref, f"split16 fused, dense input, scalars b={batch}")
            assert_f32(pd, p_ref, f"split16 fused, dense input, policy b={batch}")


@pytest.mark.parametrize("game,depth,channels,head,batches,max_batch", [
    ("ataxx-7", 8, 128, "ataxx_conv", (1, 2, 13, 256), 256),  # BASELINE configs[1]'s network in f16: two boards per workgroup
    ("chess", 2, 128, "attention", (3, 40), 64),              # 128 channels on an 8x8 board: one board per workgroup ...
    ("chess", 2, 128, "attention", (1, 3, 40, 256), 256),     # ... two (eight tiles) when that still makes 128 workgroups
    ("go-9", 2, 128, "conv", (3, 11), 64),                    # 81 pixels: six tiles, heads in the launch
    ("go-9", 2, 128, "conv", (1, 2, 3, 11, 256), 256),        # two 9x9 boards in eleven tiles
    ("go-9", 2, 128, "conv", (2, 300, 1537, 2048), 2048),     # round 5: THREE 9x9 boards in sixteen tiles (683 workgroups at 2048)
    ("chess", 2, 128, "attention", (5, 1024, 2048), 2048),    # ... four 8x8 boards in sixteen tiles
    ("ataxx-7", 2, 128, "ataxx_conv", (1, 3, 4, 5, 9, 512), 512),  # four 7x7 boards in thirteen tiles
    ("ataxx-5", 2, 128, "ataxx_conv", (1, 7, 8, 9, 17, 1024), 1024),  # four 5x5 boards in seven tiles with the heads inside
                                                                      # (eight in thirteen would leave the heads outside: round 5)
    ("chess", 2, 192, "attention", (1, 3, 40, 256), 256),     # 192 channels: two 8x8 boards per workgroup as well
    ("go-9", 2, 192, "conv", (1, 2, 3, 256), 256),            # ... two 9x9 boards in eleven tiles
    ("ataxx-7", 2, 192, "ataxx_conv", (1, 4, 5, 384), 384),   # ... three 7x7 boards in ten tiles
    ("go-9", 2, 256, "conv", (5, 64), 256),                   # 256 channels on 81 squares: six tiles, one board per workgroup
    ("ataxx-7", 4, 64, "ataxx_conv", (7, 256), 256),          # BASELINE configs[0]'s network (64 channels)
    ("go-19", 2, 128, "conv", (2,), 256),                     # 361 squares: not a shape of the launch
])
def test_resident_f16g_tower(dev, game, depth, channels, head, batches, max_batch):
    """The one-launch f16 tower for the shapes the chess launch does not take (kz_tower_resident_split without its lo
    halves): against the oracle at the f16 tolerance, and against the per-layer implicit GEMM of the same engine — same
    operands and rounding points — at the far tighter summation-order bound."""
    blob = synth.random_model(game, depth, channels, head, seed=91)
    net = O.OracleNet(blob)
    model = capi.Model(blob=blob)
    eng = capi.Engine(model, dev, max_batch, capi.KZ_DTYPE_F16)
    if game == "go-19":
        assert not eng.tower_path.startswith("tower_resident_f16g")
        return
    # at 128 channels an engine whose max_batch still gives 128 workgroups takes twice the boards per workgroup (round 4)
    wide_boards = {("chess", 128, 256): 2, ("go-9", 128, 256): 2, ("ataxx-7", 128, 512): 4, ("go-9", 128, 2048): 3, ("chess", 128, 2048): 4,
                   ("chess", 192, 256): 2, ("go-9", 192, 256): 2, ("ataxx-7", 192, 384): 3}.get((game, channels, max_batch))
    if channels in (128, 192):
        narrow = {"chess": 1, "go-9": 1, "ataxx-7": 2, "ataxx-5": 4}[game]
        per = wide_boards or narrow
        assert eng.launch_geometry(max_batch) == ((max_batch + per - 1) // per, per)
        if wide_boards:  # a batch too small for 128 wide workgroups is launched with the narrow tiles (same weights)
            assert eng.launch_geometry(9) == ((9 + narrow - 1) // narrow, narrow)
        if max_batch == 2048:  # ... and one too small for 512 sixteen-tile workgroups with the two-board level
            assert eng.launch_geometry(300) == (150, 2) and eng.launch_geometry(1024) == (512, 2)
    # conv-policy networks at 128 channels (256 on <= 64 squares) carry their heads in the launch since round 3: the tail of
    # the exact-f32 launch with its two small convolutions as f16 MFMAs on the f16 images (round 4; any number of tiles)
    # (at most four boards per workgroup in the tail: where the wide tiles would hold more — eight 5x5 boards — the selector
    # keeps the narrow tiles and the heads inside, kz_plan.hpp)
    fused = head in ("ataxx_conv", "conv") and channels == 128 and (wide_boards or 1) <= 4
    assert eng.tower_path == ("tower_resident_f16g+heads" if fused else "tower_resident_f16g")
    tower_only = eng
    if fused:
        os.environ["KZ_NO_FUSED_HEADS"] = "1"
        try:
            tower_only = capi.Engine(model, dev, max_batch, capi.KZ_DTYPE_F16)
        finally:
            del os.environ["KZ_NO_FUSED_HEADS"]
        assert tower_only.tower_path == "tower_resident_f16g"
    os.environ["KZ_FORCE_GENERIC"] = "1"
    os.environ["KZ_NO_BOARD_CONV"] = "1"
    try:
        gen = capi.Engine(model, dev, max_batch, capi.KZ_DTYPE_F16)
    finally:
        del os.environ["KZ_FORCE_GENERIC"], os.environ["KZ_NO_BOARD_CONV"]
    assert gen.tower_path == "conv_igemm_f16"
    for batch in batches:
        bits, scalars_in = synth.random_boards(game, batch, seed=92 + batch)
        s, p = eng.eval_packed(bits, scalars_in)
        n = min(batch, 16)  # the oracle on a sample
        dense = O.encode_input_full(bits[:n], scalars_in[:n], net.n_scalar, net.n_bool, net.h, net.w)
        s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
        assert_f16(s[:n], s_ref, f"scalars b={batch}")
        assert_f16(p[:n], p_ref, f"policy b={batch}")
        if fused:  # the same tower with the separate f16 head kernels: inside the tolerance as well, and it is what the
            s, p = tower_only.eval_packed(bits, scalars_in)  # implicit-GEMM path below shares its rounding points with
            assert_f16(s[:n], s_ref, f"separate heads, scalars b={batch}")
            assert_f16(p[:n], p_ref, f"separate heads, policy b={batch}")
        sg, pg = gen.eval_packed(bits, scalars_in)
        ds, dp = np.abs(sg - s).max(), np.abs(pg - p).max()
        print(f"f16g {game} {depth}x{channels} b={batch} vs implicit GEMM f16: max |d scalars| {ds:.2e}, max |d policy| {dp:.2e}")
        if depth <= 2:
            assert ds < F16_PATHS_ATOL and dp < F16_PATHS_ATOL
        else:
            assert_f16_paths_deep(sg, s, "scalars")
            assert_f16_paths_deep(pg, p, "policy")


@pytest.mark.parametrize("game,channels,head,batch", [
    ("chess", 256, "attention", 40), ("chess", 128, "attention", 40), ("go-19", 256, "conv", 5),
    ("go-9", 256, "conv", 13), ("ataxx-7", 256, "ataxx_conv", 13), ("ataxx-7", 64, "ataxx_conv", 200),
])
def test_board_conv_agrees_with_implicit_gemm(dev, game, channels, head, batch):
    """kz_board_conv_f16 against kz_conv_igemm<f16> of the same engine on every board geometry and 64/128/256 channels
    (1, 2 and 4 output-channel quarters per board group): identical operands, so they agree to summation order."""
    blob = synth.random_model(game, 2, channels, head, seed=5)
    bits, scalars_in = synth.random_boards(game, batch, seed=6)
    model = capi.Model(blob=blob)
    os.environ["KZ_FORCE_GENERIC"] = "1"
    try:
        eng = capi.Engine(model, dev, 1024, capi.KZ_DTYPE_F16)  # (the kernel is chosen when the grid fills the chip)
        os.environ["KZ_NO_BOARD_CONV"] = "1"
        gen = capi.Engine(model, dev, 1024, capi.KZ_DTYPE_F16)
    finally:
        os.environ.pop("KZ_NO_BOARD_CONV", None)
        del os.environ["KZ_FORCE_GENERIC"]
    assert eng.tower_path == "board_conv_f16" and gen.tower_path == "conv_igemm_f16"
    s, p = eng.eval_packed(bits, scalars_in)
    sg, pg = gen.eval_packed(bits, scalars_in)
    assert np.abs(sg - s).max() < F16_PATHS_ATOL and np.abs(pg - p).max() < F16_PATHS_ATOL


@pytest.mark.parametrize("nb", ["1", "2"])
def test_resident_f16_agrees_with_per_layer_paths(dev, nb):
    """The one-launch chess kernel (1 and 2 boards per workgroup) against the per-layer implicit-GEMM path of the same
    engine on a 2-block net: same operands and rounding points, so they agree to summation order — a bound 50x tighter
    than the f16-vs-oracle tolerance."""
    blob = synth.random_model("chess", 2, 256, "attention", seed=81)
    bits, scalars_in = synth.random_boards("chess", 37, seed=82)
    model = capi.Model(blob=blob)
    os.environ["KZ_TOWER_NB"] = nb
    try:
        eng = capi.Engine(model, dev, 64, capi.KZ_DTYPE_F16)
        s, p = eng.eval_packed(bits, scalars_in)
    finally:
        del os.environ["KZ_TOWER_NB"]
    assert eng.tower_path == "tower_resident_f16+heads"
    os.environ["KZ_FORCE_GENERIC"] = "1"
    os.environ["KZ_NO_BOARD_CONV"] = "1"
    try:
        gen = capi.Engine(model, dev, 64, capi.KZ_DTYPE_F16)
    finally:
        del os.environ["KZ_FORCE_GENERIC"], os.environ["KZ_NO_BOARD_CONV"]
    assert gen.tower_path == "conv_igemm_f16"
    sg, pg = gen.eval_packed(bits, scalars_in)
    ds, dp = np.abs(sg - s).max(), np.abs(pg - p).max()
    print(f"resident (NB={nb}) vs implicit GEMM f16: max |d scalars| {ds:.2e}, max |d policy| {dp:.2e}")
    assert ds < F16_PATHS_ATOL and dp < F16_PATHS_ATOL


def test_profiling_reports_kernel_time(dev):
    blob = O.load_blob("ataxx7_4x64")
    net = O.OracleNet(blob)
    bits, scalars_in = O.read_packed("ataxx7_4x64", net.n_bool, net.n_scalar, net.h, net.w)
    eng = capi.Engine(capi.Model(blob=blob), dev, 2, capi.KZ_DTYPE_F32)
    eng.set_profiling(True)
    eng.eval_packed(bits, scalars_in)
    total, n = eng.kernel_time("kz_conv_igemm")
    assert n == 1 + 2 * 4 + 1 and total > 0  # stem + 2 per block + the policy head's 1x1
    eng.set_profiling(False)


def test_abi_helpers_on_the_device(dev):
    """kz_device_pci_bus_id, kz_engine_launch_geometry, kz_model_supports_dtype, "tower.out"."""
    bus = capi.device_pci_bus_id(dev)
    assert len(bus.split(":")) == 3 and "." in bus, bus
    blob = O.load_blob("chess_2x32_att")
    model = capi.Model(blob=blob)
    assert model.supports_dtype(capi.KZ_DTYPE_F32) and model.supports_dtype(capi.KZ_DTYPE_F16)
    assert model.supports_dtype(capi.KZ_DTYPE_F32_SPLIT16)              # 32 channels: widened to 64 by zero filters
    assert not capi.Model(blob=synth.random_model("chess", 0, 64, "attention")).supports_dtype(capi.KZ_DTYPE_F32_SPLIT16)  # no block
    big = capi.Model(blob=synth.random_model("chess", 1, 256, "attention", seed=1))
    assert big.supports_dtype(capi.KZ_DTYPE_F32_SPLIT16)
    fused = capi.Engine(big, dev, 256, capi.KZ_DTYPE_F16)
    assert fused.launch_geometry(256) == (128, 2) and fused.launch_geometry(5) == (3, 2)
    with pytest.raises(capi.KzError, match="tower.out"):
        fused.read_activation("tower.out", 1)                              # the fused launch never writes the tower output
    with pytest.raises(capi.KzError, match="tower.out"):
        capi.Engine(big, dev, 256, capi.KZ_DTYPE_F32_SPLIT16).read_activation("tower.out", 1)  # fused heads as well
    os.environ["KZ_NO_FUSED_HEADS"] = "1"
    try:
        split = capi.Engine(big, dev, 256, capi.KZ_DTYPE_F32_SPLIT16)
    finally:
        del os.environ["KZ_NO_FUSED_HEADS"]
    assert split.launch_geometry(256) == (256, 1)
    bits, scalars_in = synth.random_boards("chess", 3, seed=2)
    split.eval_packed(bits, scalars_in)
    assert split.read_activation("tower.out", 3).shape == (3, 256, 8, 8)


# ---- AttentionTower networks (python/lib/model/attention.py; the tower python/main/supervised_main_alpha.py:72 trains) ----
ATT_CASES = [
    # game, depth, d_model, (heads, d_k, d_v, d_ff), head, f32 path, f16 path
    ("chess", 4, 256, (8, 16, 16, 256), "attention", "attention_tower_f32", "attention_tower_f16"),   # the reference's shape (there: depth 16)
    ("chess", 2, 256, (8, 16, 16, 512), "attention", "attention_tower_f32_valu", "attention_tower_f16"),  # (two f32 images of d_ff 512 exceed the LDS)
    ("chess", 3, 128, (8, 16, 16, 128), "dense", "attention_tower_f32", "attention_tower_f16"),
    ("chess", 2, 128, (8, 16, 16, 256), "attention", "attention_tower_f32", "attention_tower_f16"),
    # shapes only the vector-ALU kernel takes (f16 engines: f16 rows around f32 arithmetic)
    ("chess", 2, 192, (6, 32, 16, 320), "attention", "attention_tower_f32_valu", "attention_tower_f32_valu"),
    ("ataxx-7", 3, 96, (4, 12, 20, 100), "ataxx_conv", "attention_tower_f32_valu", "attention_tower_f32_valu"),
    ("go-9", 2, 64, (4, 16, 16, 128), "conv", "attention_tower_f32_valu", "attention_tower_f32_valu"),
    # odd sizes everywhere: project_out's and ff.2's rows (12 and 8 values) start at byte offsets that are no multiple of 16
    ("ataxx-7", 2, 25, (3, 3, 4, 8), "ataxx_conv", "attention_tower_f32_valu", "attention_tower_f32_valu"),
    ("chess-hist-2", 2, 256, (8, 16, 16, 256), "attention", "attention_tower_f32", "attention_tower_f16"),  # 47 input planes: more expand k-steps
]


@pytest.mark.parametrize("game,depth,d_model,att,head,f32_path,f16_path", ATT_CASES,
                         ids=[f"{c[0]}-{c[1]}x{c[2]}-h{c[3][0]}k{c[3][1]}v{c[3][2]}f{c[3][3]}" for c in ATT_CASES])
def test_attention_tower_against_the_oracle(dev, game, depth, d_model, att, head, f32_path, f16_path):
    """Exact f32 at <= 1e-4 (on v_mfma_f32_16x16x4_f32 where the shape allows, else on the vector ALUs) and f16 at the f16
    tolerance; ragged batches, and — batch 200 of an engine of 256 — the f16 launch's two boards per workgroup with an odd
    board out; through the packed-input entry point."""
    kw = dict(dense_hidden_channels=2, dense_hidden_size=32) if head == "dense" else {}
    blob = synth.random_model(game, depth, d_model, head, seed=17, attention=att, **kw)
    net = O.OracleNet(blob)
    model = capi.Model(blob=blob)
    assert not model.supports_dtype(capi.KZ_DTYPE_F32_SPLIT16)
    batch = 37
    bits, scalars_in = synth.random_boards(game, batch, seed=23)
    x = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    s_or, p_or = net.forward(x, threads=8)
    e32 = capi.Engine(model, dev, 64, capi.KZ_DTYPE_F32)
    assert e32.tower_path == f32_path
    s, p = e32.eval_packed(bits, scalars_in)
    assert_f32(s, s_or, "scalars f32")
    assert_f32(p, p_or, "policy f32")
    s1, p1 = e32.eval_packed(bits[:5], scalars_in[:5])
    assert np.array_equal(s1, s[:5]) and np.array_equal(p1, p[:5])
    e16 = capi.Engine(model, dev, 64, capi.KZ_DTYPE_F16)
    assert e16.tower_path == f16_path
    s, p = e16.eval_packed(bits, scalars_in)
    assert_f16(s, s_or, "scalars f16")
    assert_f16(p, p_or, "policy f16")
    assert np.abs(softmax(p) - softmax(p_or)).max() < 5e-3
    s1, p1 = e16.eval_packed(bits[30:], scalars_in[30:])
    assert np.array_equal(s1, s[30:]) and np.array_equal(p1, p[30:])
    if f16_path == "attention_tower_f16" and att[3] <= 256:
        # two boards per workgroup from batch 192 on: 201 boards = 100 pairs and one board alone
        big = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F16)
        assert big.launch_geometry(201) == (101, 2) and big.launch_geometry(37) == (37, 1)
        bits2, sc2 = synth.random_boards(game, 201, seed=29)
        s2, p2 = big.eval_packed(bits2, sc2)
        pick = [0, 1, 100, 199, 200]
        so, po = net.forward(O.encode_input_full(bits2[pick], sc2[pick], net.n_scalar, net.n_bool, net.h, net.w), threads=8)
        assert_f16(s2[pick], so, "scalars f16, two boards per workgroup")
        assert_f16(p2[pick], po, "policy f16, two boards per workgroup")
        s3, p3 = big.eval_packed(bits2[:37], sc2[:37])  # the same boards one per workgroup: the same arithmetic per board
        if head == "dense":  # (the dense head's GEMM tiles its rows by the batch: another summation order)
            assert np.abs(s3 - s2[:37]).max() <= F16_PATHS_ATOL and np.abs(p3 - p2[:37]).max() <= F16_PATHS_ATOL
        else:
            assert np.array_equal(s3, s2[:37]) and np.array_equal(p3, p2[:37])


def test_attention_tower_layernorm_with_a_large_common_offset(dev):
    """The exact-f32 AttentionTower is held to 1e-4.  Its first LayerNorm normalises expand(x) + embedding unnormalised; a
    trained embedding may carry a common offset far above the tokens' spread, where a one-pass variance (E[x^2] - mean^2 in
    f32) cancels: offset 60 against a spread of ~1 loses four digits.  The kernels combine per-wave (sum, M2) pairs instead
    (advisor, round 5); this case makes the offset real — on the matrix-core tower and on the vector-ALU one."""
    from kzero_amd.model_file import write_model
    for att, path in (((8, 16, 16, 256), "attention_tower_f32"), ((8, 16, 16, 512), "attention_tower_f32_valu")):
        blob = synth.random_model("chess", 2, 256, "attention", seed=31, attention=att)
        meta, tensors = read_model(blob)
        tensors = {k: np.array(v) for k, v in tensors.items()}
        tensors["common.embedding"] = (tensors["common.embedding"] + np.float32(60.0)).astype(np.float32)
        blob = write_model(meta, tensors)
        net = O.OracleNet(blob)
        bits, scalars_in = synth.random_boards("chess", 9, seed=37)
        x = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
        s_or, p_or = net.forward(x, threads=8)
        eng = capi.Engine(capi.Model(blob=blob), dev, 16, capi.KZ_DTYPE_F32)
        assert eng.tower_path == path
        s, p = eng.eval_packed(bits, scalars_in)
        assert_f32(s, s_or, f"{path}: scalars with an embedding offset of 60")
        assert_f32(p, p_or, f"{path}: policy with an embedding offset of 60")


@pytest.mark.parametrize("dtype", [capi.KZ_DTYPE_F16, capi.KZ_DTYPE_F32])
def test_attention_network_full_size_default_shim_entry(dev, dtype):
    """The network python/main/supervised_main_alpha.py:69-77 builds — AttentionTower(8, 21, 16, 256, 8, 16, 16, 256) under the
    ScalarHead and AttentionPolicyHead(game, 256, 256) — at full depth on 256 boards through the entry points `HipNetwork`
    uses by default (submit_packed_decoded on all four slots, wait_decoded): against the oracle's decode_output of the ORACLE's
    logits on every board (f16: the stated probability / value tolerances; exact f32: 1e-4) and against this library's own
    logits decoded on the host."""
    blob = synth.random_model("chess", 16, 256, "attention", seed=51, attention=(8, 16, 16, 256))
    net = O.OracleNet(blob)
    bits, scalars_in = synth.random_boards("chess", 256, seed=52)
    s_ora, p_ora = net.forward(O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, 8, 8), threads=16)
    eng = capi.Engine(capi.Model(blob=blob), dev, 256, dtype)
    assert eng.tower_path == ("attention_tower_f16" if dtype == capi.KZ_DTYPE_F16 else "attention_tower_f32")
    s_own, p_own = eng.eval_packed(bits, scalars_in)
    if dtype == capi.KZ_DTYPE_F16:
        assert_f16(s_own, s_ora, "scalars")
        assert_f16(p_own, p_ora, "policy")
    else:
        assert_f32(s_own, s_ora, "scalars")
        assert_f32(p_own, p_ora, "policy")
    slots = []
    for k in range(capi.KZ_ENGINE_SLOTS):
        order = np.roll(np.arange(256), -64 * k)
        moves = _c1_move_lists(net.policy_len, 256, seed=200 + k, finished=(9 + k,))
        slots.append((order, moves, eng.submit_packed_decoded(k, bits[order], scalars_in[order], moves)))
    worst_p = worst_v = worst_own = 0.0
    for k in (1, 3, 0, 2):
        order, moves, off = slots[k]
        v, probs = eng.wait_decoded(k, off)
        v_ora, probs_ora = O.decode_output(s_ora[order], p_ora[order], moves)
        v_own, probs_own = O.decode_output(s_own[order], p_own[order], moves)
        assert probs[9 + k].size == 0
        for b in range(256):
            if moves[b].size:
                worst_p = max(worst_p, float(np.abs(probs[b] - probs_ora[b]).max()))
                worst_own = max(worst_own, float(np.abs(probs[b] - probs_own[b]).max()))
        worst_v = max(worst_v, float(np.abs(v[:, :4] - v_ora[:, :4]).max()))
        worst_own = max(worst_own, float(np.abs(v - v_own).max()))
    print(f"[decoded {eng.tower_path}] vs oracle: max |dprob| {worst_p:.2e}, max |dvalue, dwdl| {worst_v:.2e}; vs own logits: {worst_own:.2e}")
    if dtype == capi.KZ_DTYPE_F16:
        assert worst_p <= F16_PROB_ATOL and worst_v <= F16_VALUE_ATOL
    else:
        assert worst_p <= F32_ATOL and worst_v <= F32_ATOL
    assert worst_own <= 2e-6
