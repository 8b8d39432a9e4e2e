#!/usr/bin/env python3
"""Prints the key figures of a bench.py FULL record (bench_full.json, written beside the one-line output; default: ./bench_full.json)."""
import json, sys
path = sys.argv[1] if len(sys.argv) > 1 else "bench_full.json"
r = json.load(open(path))


def roof(x):
    f = x["roofline"]
    return (f"frac {f['frac']} (launch_frac {f.get('launch_frac')}, {f.get('concurrent_launches')} at once) launch_ms {f['avg_launch_ms']} "
            f"traffic {f['traffic']} ratio {f.get('traffic_ratio')}" + (f" ERROR {f['error']}" if "error" in f else ""))


print("value", r["value"], f"[{r.get('value_min')} .. {r.get('value_max')}] over {r.get('regions')} regions of {r['steps']} steps ({r['config'].get('boundary', '')[:60]});", roof(r))
print("  device-resident", r.get("value_device_resident"), "| raw host boundary", r.get("value_host_boundary_raw"), "| parity default (split16, <= 1e-4)", r.get("value_parity_default"))
for k in r.get("hbm_bound_kernels", []):
    print("   ", k)
for o in r.get("others", []):
    if "error" in o:
        print(" ", o["workload"], o["dtype"], "ERROR", o["error"])
        continue
    print(" ", o["workload"], o["dtype"], o["value"], o["tower_path"], roof(o), ("| weights: " + o["weights"]) if "weights" in o else "")
    for k in o.get("hbm_bound_kernels", []):
        print("     ", k["kernel"], k["avg_launch_us"], "us", k["achieved_GBps"], "GB/s", k["frac_of_hbm_peak"], k.get("error", ""))
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
        print(f"  {key} [{run['config'][:70]}] {run['value']:.0f} evals/s fill {run['fill']} executor work util {run['executor_work_util']} (cpu {run['executor_cpu_util']}) helpers {run.get('helper_cpu_util')} "
              f"host cpu s/Meval {run['host_cpu_s_per_Meval']} | x8: {p['cores_needed']} cores, {p['pcie_GBps']} GB/s PCIe")
if "cpu_baseline" in r: print("cpu", r["cpu_baseline"]["value"], "cores", r["cpu_baseline"]["cores"], r["cpu_baseline"].get("error", ""), "| a0:", r["cpu_baseline"].get("a0"))
