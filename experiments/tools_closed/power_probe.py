#!/usr/bin/env python3
"""Is the chess launch's rate at a full chip set by the DATA the matrix cores multiply?  Same kernel, same launch
geometry, same instruction stream, three weight sets: the bench's uniform random weights; the same weights with their f16
mantissas cleared (signs and exponents kept: powers of two); all tower weights zero.  Device-resident steps, two engines,
batch 256, a few thousand steps each, interleaved rounds.   tools/power_probe.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kzero_amd import capi, synth
from kzero_amd.model_file import read_model, write_model

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
base = synth.random_model("chess", 20, 256, "attention", seed=0)
meta, t = read_model(base)

def variant(kind):
    tt = dict(t)
    for k in list(tt):
        if k.startswith("common.tower.") and k.endswith(".weight") and tt[k].ndim == 4:
            w = tt[k].astype(np.float32)
            if kind == "zero":
                w = np.zeros_like(w)
            elif kind == "pow2":  # keep sign and exponent, clear the mantissa (as f16 would see it)
                h = w.astype(np.float16).view(np.uint16) & np.uint16(0xFC00)
                w = h.view(np.float16).astype(np.float32)
            tt[k] = w
    return write_model(meta, tt)

blobs = {"random": base, "pow2": variant("pow2"), "zero": variant("zero")}
bits, sc = synth.random_boards("chess", 256, seed=1000)
B = 256
setups = {}
for name, blob in blobs.items():
    model = capi.Model(blob=blob)
    engines = [capi.Engine(model, 0, B, capi.KZ_DTYPE_F16) for _ in range(2)]
    d_bits = capi.DeviceBuffer.from_host(0, bits); d_sc = capi.DeviceBuffer.from_host(0, sc)
    outs = [(capi.DeviceBuffer(0, B * 5 * 4), capi.DeviceBuffer(0, B * 1880 * 4)) for _ in engines]
    setups[name] = (model, engines, d_bits, d_sc, outs)

def run(name, steps):
    model, engines, d_bits, d_sc, outs = setups[name]
    def sync():
        for e in engines:
            try: e.synchronize()
            except capi.KzError: pass   # (zero weights are fine; nothing should overflow here)
    for i in range(200):
        engines[i % 2].enqueue_packed_device(d_bits, bits.shape[1], d_sc, B, outs[i % 2][0], outs[i % 2][1])
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        engines[i % 2].enqueue_packed_device(d_bits, bits.shape[1], d_sc, B, outs[i % 2][0], outs[i % 2][1])
    sync()
    return steps * B / (time.perf_counter() - t0)

for r in range(rounds):
    for name in ("random", "pow2", "zero"):
        print(f"round {r} weights {name:7s} {run(name, 3000):10.0f} evals/s", flush=True)
