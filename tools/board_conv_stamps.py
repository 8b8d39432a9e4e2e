import sys, numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 32).astype(np.int64)
ok = d[:, 0, 0] > 0
d = d[ok]
print("workgroups with stamps:", d.shape[0])
t0 = d[:, :, 0].min()
# stamp slots (kz_board_conv.hip): 0 start, 1 set-up done, 2 first barrier, 3 chunk 0 staged; per chunk c: 4+4c staged data
# visible (k-loop starts), 5+4c k-loop done, 6+4c (c < 3) barrier behind the k-loop, 7+4c next chunk's pieces written and
# the ring re-issued; 18 barrier in front of the epilogue, 19 output tile complete, 20 stores issued
names = {1: "prologue", 2: "first barrier", 3: "chunk0 staging"}
for c in range(4):
    names[4 + 4 * c] = f"c{c}.barrier"
    names[5 + 4 * c] = f"c{c}.kloop"
    if c < 3:
        names[6 + 4 * c] = f"c{c}.end barrier"
        names[7 + 4 * c] = f"c{c}.pieces->LDS"
names[18] = "epi.barrier"
names[19] = "epi.math"
names[20] = "epi.store"
order = sorted(names)
prev = d[:, :, 0]
print("kernel span (cycles):", (d[:, :, 20].max() - t0))
life = d[:, :, 20] - d[:, :, 0]
print("wave lifetime mean/min/max:", life.mean(), life.min(), life.max())
tot = {}
prev_slot = 0
for i in order:
    seg = d[:, :, i] - d[:, :, prev_slot]
    prev_slot = i
    print(f"{names[i]:18s} mean {seg.mean():9.0f}  p10 {np.percentile(seg,10):9.0f}  p90 {np.percentile(seg,90):9.0f}  share {seg.mean()/life.mean()*100:5.1f}%")
# start times: how WGs are spread
st = np.sort(d[:, 0, 0] - t0)
print("start-time quantiles:", [int(np.percentile(st, q)) for q in (0, 25, 50, 75, 100)])

# ---- dispatch gaps: consecutive workgroups on the same (XCC, SE, CU, LDS slot) ----
hw = d[:, 0, 29]
xcc = d[:, 0, 30] & 0xF
ldsb = d[:, 0, 31] & 0xFF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
slot = (ldsb != 0).astype(np.int64)
import collections
groups = collections.defaultdict(list)
for i in range(d.shape[0]):
    groups[(int(key[i]), int(slot[i]))].append((int(d[i, :, 0].min()), int(d[i, :, 20].max())))
gaps, per = [], []
for k, v in groups.items():
    v.sort()
    per.append(len(v))
    for (s0, e0), (s1, e1) in zip(v, v[1:]):
        gaps.append(s1 - e0)
if gaps:
    gaps = np.array(gaps)
    print("CU slots seen:", len(groups), "workgroups per slot mean:", np.mean(per))
    print("gap between consecutive workgroups on a CU slot: mean %.0f p10 %.0f p50 %.0f p90 %.0f" %
          (gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 50), np.percentile(gaps, 90)))

# ---- finer set-up stamps, when present ----
if (d[:, :, 21] > 0).all():
    seq = [(0, "start"), (25, "kernel arguments, priority, ids"), (26, "12 slots located, chunk-0 loads issued"), (21, "weight ring issued"),
           (22, "halo cleared"), (24, "fragment rows"), (1, "accumulators initialised")]
    seq = [(k, n) for k, n in seq if (d[:, :, k] > 0).all()]
    for (k0, _), (k1, n1) in zip(seq, seq[1:]):
        seg = d[:, :, k1] - d[:, :, k0]
        print(f"  set-up: {n1:42s} mean {seg.mean():7.0f}  p10 {np.percentile(seg, 10):7.0f}  p90 {np.percentile(seg, 90):7.0f}")

# ---- finer epilogue stamps, when present (layers with a residual only have slot 23) ----
if (d[:, :, 27] > 0).all() and (d[:, :, 28] > 0).all():
    res = d[:, 0, 23] > 0
    for name, sel in (("with residual", res), ("without residual", ~res)):
        if not sel.any():
            continue
        e = d[sel]
        parts = [("residual pieces -> O tile (waits for the loads)", e[:, :, 23] - e[:, :, 18])] if name == "with residual" else []
        parts += [("barrier", e[:, :, 27] - (e[:, :, 23] if name == "with residual" else e[:, :, 18])),
                  ("relu / + residual / -> f16 into O", e[:, :, 28] - e[:, :, 27]), ("barrier", e[:, :, 19] - e[:, :, 28]),
                  ("stores issued", e[:, :, 20] - e[:, :, 19])]
        print(f"  epilogue, {name} ({int(sel.sum())} workgroups):")
        for n, seg in parts:
            print(f"    {n:52s} mean {seg.mean():7.0f}  p10 {np.percentile(seg, 10):7.0f}  p90 {np.percentile(seg, 90):7.0f}")
