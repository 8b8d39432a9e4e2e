#!/usr/bin/env python3
"""Prints the key figures of a bench.py JSON line (file argument or stdin)."""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
r = json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])
print("value", r["value"], "chip_frac", r["roofline"]["chip_frac"], "launch_ms", r["roofline"]["avg_launch_ms"], "traffic", r["roofline"]["traffic"])
if "pcie_inclusive" in r:
    h = r["pcie_inclusive"]; print("pcie_inclusive", h["value"], "of_resident", h["of_resident"], "engines", h.get("engines_per_gpu"))
for o in r.get("others", []):
    if "error" in o:
        print(" ", o["workload"], o["dtype"], "ERROR", o["error"])
        continue
    print(" ", o["workload"], o["dtype"], o["value"], "chip_frac", o["roofline"]["chip_frac"], "launch_ms", o["roofline"]["avg_launch_ms"], o["tower_path"])
if "cpu_baseline" in r: print("cpu", r["cpu_baseline"]["value"], "cores", r["cpu_baseline"]["cores"], r["cpu_baseline"].get("error", ""))
