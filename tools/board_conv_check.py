import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kzero_amd import capi, synth
def run(game, depth, ch, head, batch):
    blob = synth.random_model(game, depth, ch, head, seed=5)
    bits, sc = synth.random_boards(game, batch, seed=6)
    m = capi.Model(blob=blob)
    os.environ["KZ_FORCE_GENERIC"] = "1"
    e = capi.Engine(m, 0, max(batch, 512), capi.KZ_DTYPE_F16)
    path = e.tower_path
    s, p = e.eval_packed(bits, sc)
    os.environ["KZ_NO_BOARD_CONV"] = "1"
    g = capi.Engine(m, 0, max(batch, 512), capi.KZ_DTYPE_F16)
    del os.environ["KZ_NO_BOARD_CONV"]; del os.environ["KZ_FORCE_GENERIC"]
    sg, pg = g.eval_packed(bits, sc)
    print(f"{game:8s} d{depth} c{ch} b{batch}: {path} vs {g.tower_path}: ds={np.abs(s-sg).max():.4f} dp={np.abs(p-pg).max():.4f}  worst boards s: {np.argsort(-np.abs(s-sg).max(1))[:6].tolist()}")
run("chess", 2, 256, "attention", 40)
run("chess", 2, 128, "attention", 40)
run("chess", 1, 64, "attention", 40)
run("go-19", 2, 256, "conv", 5)
run("go-9", 2, 256, "conv", 13)
run("ataxx-7", 2, 256, "ataxx_conv", 13)
