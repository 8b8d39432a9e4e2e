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
