#!/usr/bin/env python3
"""Which path a network takes: the table of DESIGN.md §5.0, generated from the library's own selector.

`kz_model_plan` (include/kz_hip.h) is the host logic `kz_engine_create` runs — no GPU needed — so this script runs in the
build container.  It writes tests/golden/path_table.json:

  "paths":   {sweep case id: {"f32" | "f16" | "parity": tower path}}   for tests/test_shape_sweep.py (-m gpu)
  "lattice": rows (game, squares, c_in, channels, head) x (f32, f16, parity) -> tower path, launches per batch
             at max_batch 256, for the DESIGN table

tests/test_path_table.py recomputes both from the built library and fails on any difference: a change of a support
predicate has to come with a regenerated table (python tools/gen_path_table.py) and shows up in the diff.

    python tools/gen_path_table.py            # rewrite the JSON
    python tools/gen_path_table.py --md       # print the DESIGN table (with measured rates from a sweep record, if given:
                                              #   --rates profiles/r4/shape_sweep.json)
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from kzero_amd import capi, synth  # noqa: E402
from tests import sweep_cases  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "path_table.json")
MAX_BATCH = 256
DTYPES = {"f32": capi.KZ_DTYPE_F32, "f16": capi.KZ_DTYPE_F16, "f32split16": capi.KZ_DTYPE_F32_SPLIT16}
CHANNELS = (32, 48, 64, 96, 128, 160, 192, 256, 320, 384, 512)
LATTICE_GAMES = [  # (game, head): every board size the reference's server accepts a mapper for, with its usual head
    ("ataxx-4", "ataxx_conv"), ("ataxx-5", "ataxx_conv"), ("ataxx-6", "ataxx_conv"), ("ataxx-7", "ataxx_conv"),
    ("ataxx-8", "ataxx_conv"), ("chess", "attention"), ("chess-hist-1", "attention"), ("chess-hist-3", "attention"),
    ("chess", "dense"), ("go-9", "conv"), ("go-9-noterr", "conv"), ("go-13", "conv"), ("go-19", "conv"),
]


def plans(model):
    row = {}
    for arith in ("f32", "f16", "parity"):
        name = sweep_cases.parity_dtype_name(model, capi) if arith == "parity" else arith
        path, launches = model.plan(MAX_BATCH, DTYPES[name])
        row[arith] = {"dtype": name, "path": path, "launches": launches}
    return row


def build():
    paths = {}
    for case in sweep_cases.CASES:
        model = capi.Model(blob=synth.random_model(case.game, case.depth, case.channels, case.head, seed=11, **case.kw))
        paths[case.id] = {k: v["path"] for k, v in plans(model).items()}
    lattice = []
    for game, head in LATTICE_GAMES:
        g = synth.game_spec(game)
        for ch in CHANNELS:
            kw = dict(dense_hidden_channels=8) if head == "dense" else {}
            model = capi.Model(blob=synth.random_model(game, 2, ch, head, seed=1, **kw))
            lattice.append({"game": game, "squares": g["size"] ** 2, "c_in": g["n_scalar"] + g["n_bool"], "channels": ch,
                            "head": head, **plans(model)})
    return {"max_batch": MAX_BATCH, "paths": paths, "lattice": lattice}


def markdown(table, rates, games=None):
    by = {}
    for r in rates:
        by.setdefault((r["squares"], r["c_in"], r["channels"], r["head"]), {})[r["arith"]] = r
    print("| board (squares) | input planes | channels | head | f32 | f16 | parity default |")
    print("|---|---|---|---|---|---|---|")
    for row in table["lattice"]:
        if games and row["game"] not in games:
            continue
        cells = []
        for arith in ("f32", "f16", "parity"):
            p = row[arith]
            txt = f"`{p['path']}` ({p['launches']})"
            m = by.get((row["squares"], row["c_in"], row["channels"], row["head"]), {}).get(arith)
            if m and "evals_per_s" in m and m.get("rate_path") == p["path"]:
                txt += f" **{m['evals_per_s'] / 1e3:,.0f}k** {m['frac_of_peak']:.2f} ({m.get('engines', 2)}e)"
            cells.append(txt)
        print(f"| {row['game']} ({row['squares']}) | {row['c_in']} | {row['channels']} | {row['head']} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", action="store_true")
    ap.add_argument("--rates", default=None)
    ap.add_argument("--games", default=None, help="comma-separated subset of the lattice's games for --md")
    args = ap.parse_args()
    if args.md:
        rates = json.load(open(args.rates))["rows"] if args.rates else []
        markdown(json.load(open(OUT)), rates, args.games.split(",") if args.games else None)
    else:
        json.dump(build(), open(OUT, "w"), indent=1, sort_keys=True)
        print("wrote", OUT)
