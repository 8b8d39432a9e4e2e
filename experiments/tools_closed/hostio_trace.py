#!/usr/bin/env python3
"""Diagnostic: host-side time of every step of the PCIe-inclusive loop right after a device sync (engines x slots in
flight), to see where a short timed region loses time.  tools/hostio_trace.py [engines] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kzero_amd import capi, synth
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
w = bench.Workload(capi, synth, "chess-20x256", "f16", 256, E, 0, 1000)
w.condition(w.step_host, 0.25)
for rep in range(3):
    w.sync()
    t0 = time.perf_counter(); ts = []
    for i in range(K):
        w.step_host(i); ts.append(time.perf_counter() - t0)
    w.sync(); end = time.perf_counter() - t0
    print(f"rep {rep} engines {E}: total {end*1e3:.2f} ms = {K*256/end:.0f} evals/s; step returns (ms):",
          " ".join(f"{t*1e3:.2f}" for t in ts))
