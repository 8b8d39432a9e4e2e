#!/usr/bin/env python3
"""Path x evals/s over the shape-generality sweep (tests/sweep_cases.py) — the table of DESIGN §5.0.

For every case and every arithmetic (f32, f16, the parity default): the tower path kz_engine_create chose, the largest
deviation from the oracle on the case's boards, and the device-resident evals/s at batch 256 (Go 19x19: 128) over
`--seconds` per point with `--engines` engines round-robin (the best of the listed counts).  GPU box only:

    python tools/shape_sweep.py --out gpurun_out/shape_sweep.json [--filter chess_3x192] [--no-oracle]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

from kzero_amd import capi, synth  # noqa: E402
from tests import sweep_cases  # noqa: E402


def rate(model, dtype, game, batch, seconds, engines, n_bool=None):
    bits, sin = synth.random_boards(game, batch, seed=3, n_bool=n_bool)
    engs = [capi.Engine(model, 0, batch, dtype) for _ in range(engines)]
    info = model.info
    d_bits, d_sin = capi.DeviceBuffer.from_host(0, bits), capi.DeviceBuffer.from_host(0, sin)
    outs = [(capi.DeviceBuffer(0, batch * 5 * 4), capi.DeviceBuffer(0, batch * info.policy_len * 4)) for _ in engs]

    def run(n):
        for i in range(n):
            e = i % len(engs)
            engs[e].enqueue_packed_device(d_bits, bits.shape[1], d_sin, batch, outs[e][0], outs[e][1])
        for e in engs:
            e.synchronize()

    run(4)
    t0 = time.perf_counter()
    run(8)
    per = (time.perf_counter() - t0) / 8
    n = max(8, int(seconds / per))
    run(n // 4)
    t0 = time.perf_counter()
    run(n)
    dt = time.perf_counter() - t0
    return n * batch / dt, engs[0].launch_geometry(batch), engs[0].tower_path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--filter", default="", help="comma-separated substrings of case ids")
    ap.add_argument("--seconds", type=float, default=0.3)
    ap.add_argument("--engines", default="2,3",
                    help="engines (streams) per point, comma separated: the best rate is recorded with its engine count (the "
                         "executor's gpu_threads_per_device; 128-channel towers want three, 256-channel ones two)")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--dtypes", default="f32,f16,parity")
    ap.add_argument("--rate-depth", type=int, default=20,
                    help="tower depth of the network the rate is measured on (same shape otherwise): at the sweep's own "
                         "depths of 1 and 3 the heads and the launch overhead dominate a rate")
    ap.add_argument("--batch", type=int, default=0)
    args = ap.parse_args()
    if not args.no_oracle:
        from tests import oracle_lib as O
    rows = []
    for case in sweep_cases.CASES:
        if args.filter and not any(f in case.id for f in args.filter.split(",")):
            continue
        blob = synth.random_model(case.game, case.depth, case.channels, case.head, seed=11, **case.kw)
        model = capi.Model(blob=blob)
        info = model.info
        bits, sin = synth.random_boards(case.game, case.boards, seed=5)
        ref = None
        if not args.no_oracle:
            net = O.OracleNet(blob)
            x = O.encode_input_full(bits, sin, net.n_scalar, net.n_bool, net.h, net.w)
            ref = net.forward(x, threads=min(16, case.boards))
        batch = args.batch or (128 if info.board_h * info.board_w > 200 else 256)
        deep = capi.Model(blob=synth.random_model(case.game, args.rate_depth, case.channels, case.head, seed=11, **case.kw))
        for name in args.dtypes.split(","):
            dn = sweep_cases.parity_dtype_name(model, capi) if name == "parity" else name
            dtype = {"f32": capi.KZ_DTYPE_F32, "f16": capi.KZ_DTYPE_F16, "f32split16": capi.KZ_DTYPE_F32_SPLIT16}[dn]
            rec = {"case": case.id, "arith": name, "dtype": dn, "c_in": info.input_channels,
                   "squares": info.board_h * info.board_w, "channels": case.channels, "depth": case.depth,
                   "head": case.head, "gflop_per_eval": round(info.flops_per_eval / 1e9, 4), "batch": batch}
            try:
                eng = capi.Engine(model, 0, max(case.boards, 8), dtype)
                rec["path"] = eng.tower_path
                if ref is not None:
                    s, p = eng.eval_packed(bits, sin)
                    scale = max(1.0, float(np.abs(ref[1]).max()))
                    rec["max_abs_err"] = float(max(np.abs(s - ref[0]).max(), np.abs(p - ref[1]).max()))
                    rec["logit_scale"] = scale
                del eng
                best = None
                for n_eng in [int(x) for x in str(args.engines).split(",")]:
                    got = rate(deep, dtype, case.game, batch, args.seconds, n_eng)
                    if best is None or got[0] > best[0][0]:
                        best = (got, n_eng)
                (r, (wgs, per), rec["rate_path"]), rec["engines"] = best
                rec["rate_depth"] = args.rate_depth
                rec["evals_per_s"] = round(r, 1)
                peak = 157.3e12 if dn == "f32" else 2.5e15
                rec["rate_gflop_per_eval"] = round(deep.info.flops_per_eval / 1e9, 4)
                rec["frac_of_peak"] = round(r * deep.info.flops_per_eval / peak, 4)
                rec["workgroups"], rec["boards_per_workgroup"] = wgs, per
            except capi.KzError as e:
                rec["error"] = str(e)[:200]
            rows.append(rec)
            print(json.dumps(rec), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump({"rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
