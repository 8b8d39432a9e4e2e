#!/usr/bin/env python3
"""Reads the s_memtime stamps of a -DKZ_T32_STAMPS build of kz_tower_f32.hip (KZ_T32_STAMP_FILE) and prints where a launch's
time goes: per phase, the median over waves of the stamp-to-stamp interval in 100 MHz ticks (10 ns).
Usage: tower32_stamps.py stamps.bin [depth]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 64).astype(np.int64)
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a = a[a[:, 0] > 0]
print("waves", len(a), "tick = 10 ns")
t0 = a[:, 0].min()
def med(x): return float(np.median(x))
print("start skew (max-min of stamp 0)", a[:, 0].max() - t0)
print("staging            ", med(a[:, 1] - a[:, 0]))
print("stem               ", med(a[:, 2] - a[:, 1]))
n_layers = 2 * depth + (1 if a[:, 3 * (2 * depth + 1)].max() > 0 else 0)
loop, epi, bar = [], [], []
prev = a[:, 2]
for l in range(1, n_layers + 1):
    s0, s1, s2 = a[:, 3 * l], a[:, 3 * l + 1], a[:, 3 * l + 2]
    loop.append(med(s0 - prev)); epi.append(med(s1 - s0)); bar.append(med(s2 - s1))
    prev = s2
print("layer loops        ", [round(x) for x in loop])
print("epilogues          ", [round(x) for x in epi])
print("barrier waits      ", [round(x) for x in bar])
print("sum loops", sum(loop), "epilogues", sum(epi), "barriers", sum(bar))
if a[:, 60].max() > 0:
    if a[:, 56].max() > 0:  # finer stamps of kz_conv_heads.hpp
        print("tail: preloads issued            ", med(a[:, 56] - prev))
        print("tail: small conv over x (scalar) ", med(a[:, 57] - a[:, 56]))
        print("tail: small conv over hidden     ", med(a[:, 60] - a[:, 57]))
        print("tail: barrier + Linear partials  ", med(a[:, 58] - a[:, 60]))
        print("tail: extra moves, barrier, hidden, barrier", med(a[:, 59] - a[:, 58]))
        print("tail: last Linear                ", med(a[:, 61] - a[:, 59]))
    print("tail: small convs  ", med(a[:, 60] - prev))
    print("tail: rest         ", med(a[:, 61] - a[:, 60]))
print("whole (median wave)", med(a[:, 62] - a[:, 0]), " launch span", a[:, 62].max() - t0)
