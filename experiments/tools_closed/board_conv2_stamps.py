import sys, numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 32).astype(np.int64)
d = d[d[:, 0, 0] > 0]
print("workgroups with stamps:", d.shape[0])
names = {1: "set-up (tables, halo, ring)", 2: "first half-chunk staged"}
for h in range(8):
    names[3 + 2 * h] = f"half-chunk {h}: k-loop"
    names[4 + 2 * h] = f"half-chunk {h}: stage write + barrier"
names.update({19: "epilogue: res loads + tile writes", 20: "epilogue: barrier", 21: "epilogue: stores issued"})
order = [0, 1, 2] + [3 + i for i in range(16)] + [19, 20, 21]
life = d[:, :, 21] - d[:, :, 0]
print("wave lifetime mean/min/max:", int(life.mean()), life.min(), life.max())
for a, b in zip(order, order[1:]):
    seg = d[:, :, b] - d[:, :, a]
    print(f"{names[b]:40s} mean {seg.mean():9.0f}  p10 {np.percentile(seg,10):9.0f}  p90 {np.percentile(seg,90):9.0f}  share {seg.mean()/life.mean()*100:5.1f}%")
kl = sum((d[:, :, 3 + 2 * h] - d[:, :, 2 + 2 * h]).mean() for h in range(8))
print("k-loops total %.0f cycles (MFMA time 55296): %.1f %% efficient" % (kl, 55296 / kl * 100))
