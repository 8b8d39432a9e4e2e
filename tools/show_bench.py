#!/usr/bin/env python3
"""Prints the key figures of a bench.py JSON line (file argument or stdin)."""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
r = json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])


def roof(x):
    f = x["roofline"]
    return (f"frac {f['frac']} (launch_frac {f.get('launch_frac')}, {f.get('concurrent_launches')} at once) launch_ms {f['avg_launch_ms']} "
            f"traffic {f['traffic']} ratio {f.get('traffic_ratio')}")


print("value", r["value"], f"[{r.get('value_min')} .. {r.get('value_max')}] over {r.get('regions')} regions of {r['steps']} steps;",
      "host boundary", r.get("value_host_boundary"), f"(frac {r['roofline'].get('host_boundary_frac')});", roof(r))
if "pcie_inclusive" in r:
    h = r["pcie_inclusive"]; print("pcie_inclusive", h["value"], f"[{h.get('value_min')} .. {h.get('value_max')}]", "of_resident", h["of_resident"], "engines", h.get("engines_per_gpu"), "| decoded entry:", h.get("decoded"))
if "value_parity_default" in r: print("parity default (split16, <= 1e-4):", r["value_parity_default"])
for o in r.get("others", []):
    if "error" in o:
        print(" ", o["workload"], o["dtype"], "ERROR", o["error"])
        continue
    print(" ", o["workload"], o["dtype"], o["value"], o["tower_path"], roof(o), ("| weights: " + o["weights"]) if "weights" in o else "")
for key in ("seam", "seam_parity", "seam_one_process"):
    s = r.get(key)
    if not s:
        continue
    if "error" in s or "skipped" in s:
        print(key, s)
        continue
    for run in [s] + s.get("variants", []):
        if "error" in run:
            print(" ", key, run["config"], "ERROR", run["error"])
            continue
        p = run["projection_8gpu"]
        print(f"  {key} [{run['config']}] {run['value']:.0f} evals/s fill {run['fill']} executor work util {run['executor_work_util']} (cpu {run['executor_cpu_util']}) helpers {run.get('helper_cpu_util')} "
              f"generators {run['generator_cpu_util']} host cpu s/Meval {run['host_cpu_s_per_Meval']} | x8: {p['cores_needed']} cores "
              f"of {p['cores_per_numa_node']} per node, {p['pcie_GBps']} GB/s PCIe")
if "cpu_baseline" in r: print("cpu", r["cpu_baseline"]["value"], "cores", r["cpu_baseline"]["cores"], r["cpu_baseline"].get("error", ""), "| a0:", r["cpu_baseline"].get("a0"))
