"""Parity cases of the EXPERIMENT build (experiments/libkzhip_exp.so: `experiments/build.sh`): the kernel
organisations that were built, measured and not adopted — four boards per workgroup (kz_tower4.hip), two Go boards per
workgroup (kz_board_conv2.hip), the 32x32x16 MFMA variants, hipGraph replay, three Ataxx boards per workgroup of the
exact-f32 launch — each against the product kernels compiled into the same library and against the oracle; and, the other
way round, the four head launches that the product's one-launch ScalarHead + AttentionPolicyHead kernel replaced
(KZ_NO_ATT_HEADS=1).  None of the experiment code is in libkzhip.so.

Not collected by `pytest tests/` (the file name does not match test_*.py): tests/test_gpu_experiments.py runs it in a
child pytest process with KZ_LIB_PATH pointing at the experiment library.
"""
import os

import numpy as np
import pytest

from kzero_amd import capi, synth
from tests import oracle_lib as O
from tests.test_gpu_parity import (F16_PATHS_ATOL, _scaled_stem, assert_f16, assert_f16_paths_deep, assert_f32)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert capi.LIB_PATH.endswith("libkzhip_exp.so"), "run through tests/test_gpu_experiments.py (KZ_LIB_PATH)"
    assert capi.device_count() >= 1
    return 0


@pytest.fixture(scope="module")
def chess_full():
    blob = synth.random_model("chess", 20, 256, "attention", seed=21)
    bits, scalars_in = synth.random_boards("chess", 256, seed=22)
    return blob, bits, scalars_in


def test_four_board_tower_agrees_with_the_two_board_launch(dev, chess_full):
    """kz_tower4.hip (KZ_TOWER_NB=4: four boards per workgroup, in-place LDS image, residual slab, asm-pinned
    accumulators) against the product launch (two boards per workgroup) on the full 20x256 network, all 256 boards plus a
    ragged batch: the tower outputs differ only by where the bias enters the f32 sum (after the products instead of
    before), i.e. by single f16 roundings; and both against the oracle on a sample at the f16 tolerance."""
    blob, bits, scalars_in = chess_full
    model = capi.Model(blob=blob)
    os.environ["KZ_NO_FUSED_HEADS"] = "1"
    try:
        two = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F16)
        os.environ["KZ_TOWER_NB"] = "4"
        four = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F16)
    finally:
        os.environ.pop("KZ_TOWER_NB", None)
        del os.environ["KZ_NO_FUSED_HEADS"]
    assert two.tower_path == four.tower_path == "tower_resident_f16"
    assert two.launch_geometry(256) == (128, 2) and four.launch_geometry(256) == (64, 4)
    for n in (256, 7, 1):
        s2, p2 = two.eval_packed(bits[:n], scalars_in[:n])
        t2 = two.read_activation("tower.out", n)
        s4, p4 = four.eval_packed(bits[:n], scalars_in[:n])
        t4 = four.read_activation("tower.out", n)
        assert_f16_paths_deep(t4, t2, f"tower output, {n} boards")
        assert_f16_paths_deep(p4, p2, f"policy, {n} boards")
        assert_f16_paths_deep(s4, s2, f"scalars, {n} boards")
    net = O.OracleNet(blob)
    pick = np.array([0, 1, 2, 3, 252, 253, 254, 255])
    dense = O.encode_input_full(bits[pick], scalars_in[pick], net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    s4, p4 = four.eval_packed(bits, scalars_in)
    assert_f16(s4[pick], s_ref, "four-board launch vs oracle, scalars")
    assert_f16(p4[pick], p_ref, "four-board launch vs oracle, policy")



def test_board_conv2_experiment_agrees_with_the_product_kernel(dev):
    """kz_board_conv2.hip (opt-in KZ_BOARD_CONV2=1: two Go boards per workgroup, staging under the MFMAs) against
    kz_board_conv.hip on Go-19 256-channel layers, odd batch included: identical operands and rounding points, so they
    agree to summation order; and against the oracle on one board."""
    blob = synth.random_model("go-19", 2, 256, "conv", seed=5)
    model = capi.Model(blob=blob)
    ref = capi.Engine(model, dev, 512, capi.KZ_DTYPE_F16)
    os.environ["KZ_BOARD_CONV2"] = "1"
    try:
        exp = capi.Engine(model, dev, 512, capi.KZ_DTYPE_F16)
    finally:
        del os.environ["KZ_BOARD_CONV2"]
    assert ref.tower_path == exp.tower_path == "board_conv_f16"
    assert ref.launch_geometry(512) == (2048, 0) and exp.launch_geometry(512) == (1024, 0)
    for n in (7, 64):
        bits, scalars_in = synth.random_boards("go-19", n, seed=6 + n)
        s0, p0 = ref.eval_packed(bits, scalars_in)
        t0 = ref.read_activation("tower.out", n)
        s1, p1 = exp.eval_packed(bits, scalars_in)
        t1 = exp.read_activation("tower.out", n)
        assert np.abs(t1 - t0).max() < F16_PATHS_ATOL and np.abs(p1 - p0).max() < F16_PATHS_ATOL and np.abs(s1 - s0).max() < F16_PATHS_ATOL
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits[:1], scalars_in[:1], net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    assert_f16(s1[:1], s_ref, "board_conv2 vs oracle, scalars")
    assert_f16(p1[:1], p_ref, "board_conv2 vs oracle, policy")


@pytest.mark.parametrize("dtype_name,env", [("split16", "KZ_SPLIT_MFMA32"), ("f16", "KZ_F16G_MFMA32")])
def test_mfma_32x32_experiment_agrees_with_the_product_kernel(dev, tmp_path, dtype_name, env):
    """kz_tower_resident_split32 (opt-in: the 256-channel, 64-row launch on v_mfma_f32_32x32x16_f16; the switch is read
    once per process, hence the child process): the same sums as the 16x16x32 launch in another order — split16 within the
    1e-4 of the oracle, plain f16 within the f16 path tolerance of the product launch."""
    import subprocess
    import sys
    game, depth, head = ("chess", 3, "attention") if dtype_name == "split16" else ("ataxx-7", 3, "ataxx_conv")
    out = str(tmp_path / "out.npz")
    code = f"""
import numpy as np
from kzero_amd import capi, synth
blob = synth.random_model({game!r}, {depth}, 256, {head!r}, seed=44)
eng = capi.Engine(capi.Model(blob=blob), {dev}, 16, capi.KZ_DTYPE_F32_SPLIT16 if {dtype_name!r} == "split16" else capi.KZ_DTYPE_F16)
bits, sc = synth.random_boards({game!r}, 11, seed=45)
s, p = eng.eval_packed(bits, sc)
np.savez({out!r}, s=s, p=p, path=eng.tower_path)
"""
    # (the 32x32x16 launch has no fused heads: both sides run the tower launch + the separate head kernels)
    child_env = dict(os.environ, **{env: "1", "KZ_NO_FUSED_HEADS": "1"})
    r = subprocess.run([sys.executable, "-c", code], env=child_env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr
    got = np.load(out)
    blob = synth.random_model(game, depth, 256, head, seed=44)
    bits, sc = synth.random_boards(game, 11, seed=45)
    code_dtype = capi.KZ_DTYPE_F32_SPLIT16 if dtype_name == "split16" else capi.KZ_DTYPE_F16
    os.environ["KZ_NO_FUSED_HEADS"] = "1"
    try:
        eng = capi.Engine(capi.Model(blob=blob), dev, 16, code_dtype)
    finally:
        del os.environ["KZ_NO_FUSED_HEADS"]
    assert str(got["path"]) == eng.tower_path == ("tower_resident_split16" if dtype_name == "split16" else "tower_resident_f16g")
    s, p = eng.eval_packed(bits, sc)
    if dtype_name == "split16":
        net = O.OracleNet(blob)
        s_ref, p_ref = net.forward(O.encode_input_full(bits, sc, net.n_scalar, net.n_bool, net.h, net.w),
                                   threads=os.cpu_count() or 1)
        assert_f32(got["s"], s_ref, "32x32x16 split launch vs oracle, scalars")
        assert_f32(got["p"], p_ref, "32x32x16 split launch vs oracle, policy")
        assert_f32(got["s"], s, "32x32x16 vs 16x16x32 split launch, scalars")
        assert_f32(got["p"], p, "32x32x16 vs 16x16x32 split launch, policy")
        assert not (np.array_equal(got["p"], p) and np.array_equal(got["s"], s)), "the child did not run the other kernel"
    else:
        assert np.abs(got["p"] - p).max() < F16_PATHS_ATOL and np.abs(got["s"] - s).max() < F16_PATHS_ATOL


@pytest.mark.parametrize("game,depth,channels,head,dtype,path,env", [
    ("go-9", 2, 128, "conv", "f16", "board_conv_f16", {"KZ_NO_RESIDENT_F16G": "1"}),
    ("go-9", 2, 128, "conv", "f16", "conv_igemm_f16", {"KZ_NO_RESIDENT_F16G": "1", "KZ_NO_BOARD_CONV": "1"}),
    ("ataxx-7", 2, 128, "ataxx_conv", "f16", "tower_resident_f16g", {"KZ_NO_FUSED_HEADS": "1"}),
    ("chess", 2, 256, "attention", "split16", "tower_resident_split16", {"KZ_NO_FUSED_HEADS": "1"}),
    ("chess", 2, 64, "attention", "f32", "conv_igemm_f32", {}),
])
def test_hip_graph_replay_is_the_same_forward_pass(dev, game, depth, channels, head, dtype, path, env):
    """KZ_HIP_GRAPH=1 (the multi-launch paths' forward pass captured once per (entry point, batch,
    buffers) and replayed with one hipGraphLaunch): bitwise the outputs of the eager engine on the asynchronous pair and on
    the device-resident entry point, for several batch sizes and repeated replays; the range check still reports."""
    blob = synth.random_model(game, depth, channels, head, seed=61)
    code = {"f16": capi.KZ_DTYPE_F16, "f32": capi.KZ_DTYPE_F32, "split16": capi.KZ_DTYPE_F32_SPLIT16}[dtype]
    os.environ.update(env)
    try:
        eager = capi.Engine(capi.Model(blob=blob), dev, 512, code)
        os.environ["KZ_HIP_GRAPH"] = "1"
        graph = capi.Engine(capi.Model(blob=blob), dev, 512, code)
        big = capi.Engine(capi.Model(blob=_scaled_stem(blob, 3.0e5)), dev, 512, code) if dtype != "f32" else None
    finally:
        for k in list(env) + ["KZ_HIP_GRAPH"]:
            os.environ.pop(k, None)
    assert eager.tower_path == graph.tower_path == path
    for batch in (7, 32, 7):  # the second 7 replays the first 7's graphs
        bits, sc = synth.random_boards(game, batch, seed=62 + batch)
        s_ref, p_ref = eager.eval_packed(bits, sc)
        for rep in range(3):  # first pass eager (warm-up), second captures, third replays
            for slot in range(capi.KZ_ENGINE_SLOTS):
                graph.submit_packed(slot, bits, sc)
            for slot in range(capi.KZ_ENGINE_SLOTS):
                s, p = graph.wait(slot, batch)
                assert np.array_equal(s, s_ref) and np.array_equal(p, p_ref), f"slot {slot} batch {batch} rep {rep}"
        d_bits, d_sc = capi.DeviceBuffer.from_host(dev, bits), capi.DeviceBuffer.from_host(dev, sc)
        d_s, d_p = capi.DeviceBuffer(dev, batch * 5 * 4), capi.DeviceBuffer(dev, batch * graph.model.info.policy_len * 4)
        for rep in range(3):
            graph.enqueue_packed_device(d_bits, bits.shape[1], d_sc, batch, d_s, d_p)
            graph.synchronize()
            assert np.array_equal(d_s.to_host(np.float32, (batch, 5)), s_ref), f"device-resident batch {batch} rep {rep}"
            assert np.array_equal(d_p.to_host(np.float32, p_ref.shape), p_ref)
    if big is not None:
        bits, sc = synth.random_boards(game, 9, seed=70)
        for rep in range(3):  # eager, captured, replayed: all report
            with pytest.raises(capi.KzError, match="non-finite activation"):
                big.wait_view(0, big.submit_packed(0, bits, sc))
        d_s, d_p = capi.DeviceBuffer(dev, 9 * 5 * 4), capi.DeviceBuffer(dev, 9 * big.model.info.policy_len * 4)
        d_bits, d_sc = capi.DeviceBuffer.from_host(dev, bits), capi.DeviceBuffer.from_host(dev, sc)
        for rep in range(3):
            big.enqueue_packed_device(d_bits, bits.shape[1], d_sc, 9, d_s, d_p)
            with pytest.raises(capi.KzError, match="non-finite activation"):
                big.synchronize()
            big.synchronize()  # reported once


def test_three_board_f32_launch_agrees_with_the_oracle(dev):
    """KZ_T32_BOARDS=3: the exact-f32 one-launch network with THREE 7x7 boards per workgroup (ten tiles, images that hold
    the boards' 147 rows and nothing behind them, three row tiles per wave in the heads' small convolutions) on BASELINE
    configs[1] against the oracle and the product's two-board launch, ragged last workgroups included.  Measured (round 4,
    tools/ab_a1_f32.sh): 7 % faster per board and workgroup, but a batch of 256 is 86 workgroups — 458k evals/s with three
    engines, 517k with four, against 520k for two boards."""
    blob = synth.random_model("ataxx-7", 8, 128, "ataxx_conv", seed=11)
    bits, scalars_in = synth.random_boards("ataxx-7", 256, seed=12)
    net = O.OracleNet(blob)
    dense = O.encode_input_full(bits, scalars_in, net.n_scalar, net.n_bool, net.h, net.w)
    s_ref, p_ref = net.forward(dense, threads=os.cpu_count() or 1)
    model = capi.Model(blob=blob)
    two = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F32)
    os.environ["KZ_T32_BOARDS"] = "3"
    try:
        three = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F32)
        os.environ["KZ_NO_FUSED_HEADS"] = "1"
        three_tower = capi.Engine(model, dev, 256, capi.KZ_DTYPE_F32)
    finally:
        os.environ.pop("KZ_T32_BOARDS", None)
        os.environ.pop("KZ_NO_FUSED_HEADS", None)
    assert two.launch_geometry(256) == (128, 2)
    assert three.tower_path == "tower_resident_f32+heads" and three.launch_geometry(256) == (86, 3)
    assert three_tower.tower_path == "tower_resident_f32" and three_tower.launch_geometry(5) == (2, 3)
    for n in (256, 7, 5, 4, 3, 2, 1):
        s3, p3 = three.eval_packed(bits[:n], scalars_in[:n])
        assert_f32(s3, s_ref[:n], f"scalars, {n} boards")
        assert_f32(p3, p_ref[:n], f"policy, {n} boards")
        s2, p2 = two.eval_packed(bits[:n], scalars_in[:n])
        assert_f32(s3, s2, f"scalars vs two boards, {n} boards")
        assert_f32(p3, p2, f"policy vs two boards, {n} boards")
        st, pt = three_tower.eval_packed(bits[:n], scalars_in[:n])
        assert_f32(pt, p_ref[:n], f"policy, separate heads, {n} boards")
        assert_f32(st, s_ref[:n], f"scalars, separate heads, {n} boards")


@pytest.mark.parametrize("game,depth,channels,kw", [
    ("chess", 3, 128, {}),                                   # tower_resident_f16g, Q = C = 128
    ("chess", 2, 256, dict(query_channels=64)),              # the chess launch without its heads (Q != 256)
    ("chess", 2, 256, dict(attention=(8, 16, 16, 256))),     # the AttentionTower network
    ("chess-hist-1", 2, 192, {}),
])
def test_one_launch_attention_heads_agree_with_the_four_launches(dev, game, depth, channels, kw):
    """kz_att_heads_f16 (ScalarHead + AttentionPolicyHead of a board in one launch) against the four launches it replaces
    (KZ_NO_ATT_HEADS=1, experiment build only: kz_scalar_head, two kz_conv1x1_split, kz_attention_mfma) — same operands and
    rounding points (the 1x1 convolutions' outputs rounded to f16), other summation orders."""
    blob = synth.random_model(game, depth, channels, "attention", seed=41, **kw)
    model = capi.Model(blob=blob)
    bits, scalars_in = synth.random_boards(game, 77, seed=42)
    one = capi.Engine(model, dev, 128, capi.KZ_DTYPE_F16)
    n_one = model.plan(128, capi.KZ_DTYPE_F16)[1]
    os.environ["KZ_NO_ATT_HEADS"] = "1"
    try:
        four = capi.Engine(model, dev, 128, capi.KZ_DTYPE_F16)
        n_four = model.plan(128, capi.KZ_DTYPE_F16)[1]
    finally:
        del os.environ["KZ_NO_ATT_HEADS"]
    assert n_four == n_one + 3, (n_one, n_four)
    s1, p1 = one.eval_packed(bits, scalars_in)
    s4, p4 = four.eval_packed(bits, scalars_in)
    assert np.abs(s1 - s4).max() <= F16_PATHS_ATOL and np.abs(p1 - p4).max() <= F16_PATHS_ATOL * max(1.0, float(np.abs(p4).max()))
