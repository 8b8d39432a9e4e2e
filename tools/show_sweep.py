#!/usr/bin/env python3
"""Prints a shape-sweep record file (tools/shape_sweep.py --out) as a table; --md for the DESIGN table."""
import json
import sys

rows = json.load(open(sys.argv[1]))["rows"]
md = "--md" in sys.argv
only = [a for a in sys.argv[2:] if not a.startswith("--")]
if md:
    print("| case | c_in | squares | channels | head | arithmetic | path (batch 256; Go-19: 128) | evals/s at depth %s | of the peak | max err vs oracle |" % rows[0].get("rate_depth", "?"))
    print("|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    if only and not any(o in r["case"] for o in only):
        continue
    err = r.get("max_abs_err")
    if md:
        print(f"| {r['case']} | {r['c_in']} | {r['squares']} | {r['channels']} | {r['head']} | {r['arith']} ({r['dtype']}) | "
              f"`{r.get('rate_path', r.get('path', 'ERR'))}` | {r.get('evals_per_s', 0):,.0f} ({r.get('engines', '?')} engines) | {r.get('frac_of_peak', 0):.3f} | "
              f"{'' if err is None else format(err, '.1e')} |")
    else:
        print(f"{r['case']:50s} {r['arith']:7s} {r.get('rate_path', r.get('path', 'ERR')):30s} "
              f"err={-1 if err is None else err:.2e} {r.get('evals_per_s', 0):>10.0f} {r.get('frac_of_peak', 0):.3f} "
              f"wg={r.get('workgroups')} {r.get('error', '')[:80]}")
