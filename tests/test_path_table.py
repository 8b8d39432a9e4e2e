"""Which path a network takes: the committed table equals what the built library's selector chooses.
`kz_model_plan` is the host logic `kz_engine_create` runs (kzero_amd/csrc/kz_plan.hpp: plan_path) — no GPU needed."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

import gen_path_table  # noqa: E402


def test_committed_path_table_is_what_the_selector_chooses():
    committed = json.load(open(gen_path_table.OUT))
    fresh = json.loads(json.dumps(gen_path_table.build()))
    assert committed["max_batch"] == fresh["max_batch"]
    for case_id, row in fresh["paths"].items():
        assert committed["paths"].get(case_id) == row, f"{case_id}: regenerate with tools/gen_path_table.py"
    assert set(committed["paths"]) == set(fresh["paths"])
    assert committed["lattice"] == fresh["lattice"], "lattice differs: regenerate with tools/gen_path_table.py"


def test_every_sweep_case_has_a_one_launch_or_board_tile_path_at_128_channels_and_up():
    """No network with >= 128 tower channels in a multiple of 64 falls back to the generic implicit GEMM in f16 at the
    executor batch: it takes a one-launch tower or the per-layer board-tile kernel."""
    table = json.load(open(gen_path_table.OUT))
    for row in table["lattice"]:
        if row["channels"] >= 128 and row["channels"] % 64 == 0:
            assert not row["f16"]["path"].startswith("conv_igemm"), row


def test_selector_choices_the_round4_advisor_asked_for():
    """(1) An engine whose max_batch would take the wide tiles keeps the fused conv heads where they fit the narrow tiles only
    (128 channels on 5x5: eight boards per wide workgroup against the tail's four) — one launch per batch is what the
    zero-copy slots and the in-launch decode need.  (2) A tower is widened to a multiple of 64 channels only when that buys
    another kernel: Go 19x19 x 96 at max_batch 8 stays a 96-channel implicit GEMM (same path name, one launch more for the
    head that the 96-channel 1x1 kernel does not fuse)."""
    from kzero_amd import capi, synth
    m = capi.Model(blob=synth.random_model("ataxx-5", 2, 128, "ataxx_conv", seed=1))
    assert m.plan(2048, capi.KZ_DTYPE_F16) == ("tower_resident_f16g+heads", 1)
    assert m.plan(256, capi.KZ_DTYPE_F16) == ("tower_resident_f16g+heads", 1)
    g = capi.Model(blob=synth.random_model("go-19", 2, 96, "conv", seed=1))
    small, big = g.plan(8, capi.KZ_DTYPE_F16), g.plan(512, capi.KZ_DTYPE_F16)
    assert small[0] == "conv_igemm_f16" and big[0] == "board_conv_f16"  # (512 boards: widened to 128 channels, the board-tile kernel)
    # split arithmetic needs a multiple of 64: still widened, still accepted
    assert g.supports_dtype(capi.KZ_DTYPE_F32_SPLIT16) and g.plan(8, capi.KZ_DTYPE_F32_SPLIT16)[0] == "board_conv_split16"
