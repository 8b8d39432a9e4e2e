#!/usr/bin/env python3
"""Summary of a tools/first_node.sh run: reads <dir>/bench_<N>.json and prints, per N, the aggregate rate, the scaling
efficiency against N = 1 (the driver computes its own; this is for the person at the box), every rank's own line, and checks
what a wrong multi-GPU run gets wrong without any error: fewer distinct PCI bus ids than ranks, a missing rank line, a rank
that did not bind to its GPU's NUMA node, a rank far slower than the others.  Exit code 1 when a check fails."""
import json
import os
import sys


def load(path):
    try:
        lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def main(out_dir):
    recs = {n: load(os.path.join(out_dir, f"bench_{n}.json")) for n in (1, 2, 4, 8)}
    recs = {n: r for n, r in recs.items() if r is not None}
    if not recs:
        print("first_node: no bench line was written")
        return 1
    bad = []
    base = recs.get(1, {}).get("value")
    print(f"{'N':>2} {'value':>12} {'per GPU':>12} {'eff':>6}  ranks (device bus numa evals/s)")
    for n, r in sorted(recs.items()):
        per = r.get("per_rank", [])
        eff = f"{r['value'] / (n * base):.3f}" if base else "-"
        ranks = " | ".join(f"{x['device']} {x.get('bus_id')} numa {x.get('numa_node')}{'' if x.get('numa_bound', True) else ' (unbound)'} "
                           f"{x['evals_s']:,.0f}" for x in per)
        print(f"{n:>2} {r['value']:>12,.1f} {r['value'] / n:>12,.1f} {eff:>6}  {ranks}")
        if r.get("n_gpus") != n:
            bad.append(f"N={n}: the line says n_gpus = {r.get('n_gpus')}")
        if len(per) != n:
            bad.append(f"N={n}: {len(per)} rank lines")
        seen = r.get("devices_seen", [])
        if r.get("shared_gpu"):
            print(f"   N={n}: REHEARSAL on a shared GPU ({sorted(set(seen))}): the multi-rank code path ran, the rate is no scaling figure")
        elif len(set(seen)) != n:
            bad.append(f"N={n}: {len(set(seen))} distinct PCI bus ids for {n} ranks: {seen}")
        rates = [x["evals_s"] for x in per]
        if rates and r.get("data") != "fake" and not r.get("shared_gpu") and min(rates) < 0.9 * max(rates):
            bad.append(f"N={n}: slowest rank at {min(rates) / max(rates):.2f} of the fastest (NUMA? thermals? a shared link?)")
        unbound = [x["rank"] for x in per if x.get("numa_node") is not None and x.get("numa_bound") is False]
        if unbound and r.get("data") != "fake":
            bad.append(f"N={n}: ranks {unbound} know their GPU's NUMA node but did not bind to it")
        one = r.get("seam_one_process")
        if one:
            if "value" in one:
                print(f"   one process over {one.get('devices')}: {one['value']:,.0f} evals/s, per device {one.get('per_device_evals_per_s')}, "
                      f"executor work util {one.get('executor_work_util')}")
            else:
                print(f"   one process: {one}")
    for b in bad:
        print("CHECK FAILED:", b)
    print("first_node: " + ("ok" if not bad else f"{len(bad)} check(s) failed"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/first_node"))
