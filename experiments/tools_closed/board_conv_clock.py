"""The clock the CUs really run at during kz_board_conv_f16, and the idle time between two launches, from a
-DKZ_BC_STAMPS -DKZ_BC_REALTIME build's stamps of four consecutive launches (tools/go_clock.sh).

Every wave stamps s_memtime (shader cycles) and s_memrealtime (a constant-rate counter shared by the chip) at its start and
end: the ratio of the two spans of one wave is cycles per tick; the real-time stamps of different CUs are comparable (the
cycle counters are not), so the first start and the last end of a launch give its execution window and the gap to the next
launch; the tick's length follows from the host-timed step (a step is 81 such periods plus the small kernels)."""
import json, sys, collections
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(4, -1, 4, 32).astype(np.int64)
bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
S = [d[q][:, :, 25].min() for q in range(4)]
E = [d[q][:, :, 26].max() for q in range(4)]
period = (S[3] - S[0]) / 3.0
small_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25  # head, encode and decode kernels of a step
tick_ns = (bench["ms_per_step"] - small_ms) / 81.0 * 1e6 / period
print(f"period between launches {period:.0f} ticks; tick = {tick_ns:.2f} ns (step {bench['ms_per_step']:.3f} ms = 81 periods + {small_ms} ms)")
for q in range(4):
    gap = f", then {S[q + 1] - E[q]} ticks ({(S[q + 1] - E[q]) * tick_ns / 1e3:.2f} us) until the next launch's first workgroup" if q < 3 else ""
    print(f"launch {q}: first start -> last end {E[q] - S[q]} ticks ({(E[q] - S[q]) * tick_ns / 1e3:.1f} us){gap}")
w = d[0]
cyc = (w[:, :, 20] - w[:, :, 0]).astype(np.float64)
tic = (w[:, :, 26] - w[:, :, 25]).astype(np.float64)
cpt = cyc.sum() / tic.sum()
print(f"shader cycles per tick (all waves of launch 0): {cpt:.2f} -> clock {cpt / tick_ns:.3f} GHz")
starts = np.sort(w[:, :, 25].min(axis=1) - S[0])
ends = np.sort(w[:, :, 26].max(axis=1) - S[0])
n = w.shape[0]
print(f"workgroup {n // 4} of {n} starts at {starts[n // 4 - 1] * tick_ns / 1e3:.2f} us; the last workgroup of each of the {n // 4} slots ends between "
      f"{ends[-(n // 4)] * tick_ns / 1e3:.1f} and {ends[-1] * tick_ns / 1e3:.1f} us (mean {ends[-(n // 4):].mean() * tick_ns / 1e3:.1f})")
hw = w[:, 0, 29]
key = ((((w[:, 0, 30] & 0xF) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xF))
span = collections.defaultdict(lambda: [1 << 62, 0, 0])
for k, s, e in zip(key, w[:, :, 0].min(axis=1), w[:, :, 20].max(axis=1)):
    v = span[k]
    v[0] = min(v[0], s); v[1] = max(v[1], e); v[2] += 1
spans = np.array([v[1] - v[0] for v in span.values()], dtype=np.float64)
items = np.array([v[2] for v in span.values()], dtype=np.float64)
mfma = float(sys.argv[4]) if len(sys.argv) > 4 else 27648.0  # MFMA cycles of one workgroup per SIMD: 72 k-steps x 24 x 16
print(f"CUs {len(spans)}: span of a CU's workgroups {spans.mean():.0f} cycles (min {spans.min():.0f}, max {spans.max():.0f}), "
      f"{items.mean():.1f} workgroups each -> the matrix pipe is issued to for {np.mean(items * mfma / spans) * 100:.1f} % of the span")
