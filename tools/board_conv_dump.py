"""Diagnostic: dump the per-layer activations of the per-layer f16 path (KZ_KEEP_ACTIVATIONS) to an .npz, or compare two
dumps made with different builds (KZ_LIB_PATH)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "dump":
    os.environ["KZ_FORCE_GENERIC"] = "1"
    os.environ["KZ_KEEP_ACTIVATIONS"] = "1"
    from kzero_amd import capi, synth
    game, ch, batch = sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    blob = synth.random_model(game, 1, ch, "conv" if game.startswith("go") else "attention", seed=5)
    bits, sc = synth.random_boards(game, batch, seed=6)
    e = capi.Engine(capi.Model(blob=blob), 0, 512, capi.KZ_DTYPE_F16)
    e.eval_packed(bits, sc)
    out = {k: e.read_activation(k, batch) for k in ["tower.0", "tower.1.mid", "tower.2"]}
    np.savez(sys.argv[2], path=e.tower_path, **out)
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    print(a["path"], b["path"])
    for k in ["tower.0", "tower.1.mid", "tower.2"]:
        d = np.abs(a[k].astype(np.float32) - b[k].astype(np.float32))
        print(k, a[k].shape, "max", d.max())
        if d.max() > 1e-3:
            idx = np.argwhere(d > 1e-3)
            print("  differing entries:", len(idx), "of", d.size)
            for ax in range(idx.shape[1]):
                vals, cnt = np.unique(idx[:, ax], return_counts=True)
                print("  axis", ax, "values:", vals[:40].tolist(), "counts:", cnt[:40].tolist())
    k = "tower.1.mid"
    d = np.abs(a[k].astype(np.float32) - b[k].astype(np.float32))
    per_ch = (d > 1e-3).mean(axis=(0, 2, 3))
    print("fraction of differing entries per channel:")
    print(np.round(per_ch, 2).tolist())
    dd = d[0].max(axis=0)
    print("board 0: max diff per pixel (rows = y):")
    for y in range(dd.shape[0]):
        print(" ".join(f"{v:4.2f}" for v in dd[y]))
