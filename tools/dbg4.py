import sys, os, numpy as np
sys.path.insert(0, '.')
from kzero_amd import capi, synth
from kzero_amd.model_file import read_model, write_model
os.environ["KZ_NO_FUSED_HEADS"] = "1"
mode = sys.argv[1]; nb = sys.argv[2]; out = sys.argv[3]
depth, n = 1, 8
blob = synth.random_model("chess", depth, 256, "attention", seed=81)
meta, t = read_model(blob); t = dict(t)
for key in ("common.tower.1.seq.0.weight", "common.tower.1.seq.3.weight"):
    w = t[key].copy()
    if mode == "zero": w[:] = 0
    elif mode.startswith("tap"):          # keep only tap (ky,kx)
        ky, kx = int(mode[3]), int(mode[4])
        keep = w[:, :, ky, kx].copy(); w[:] = 0; w[:, :, ky, kx] = keep
    t[key] = w
blob = write_model(meta, t)
bits, sc = synth.random_boards("chess", n, seed=82)
os.environ["KZ_TOWER_NB"] = nb
e = capi.Engine(capi.Model(blob=blob), 0, 256, capi.KZ_DTYPE_F16)
try:
    s, p = e.eval_packed(bits, sc)
except capi.KzError as ex:
    print('ERR', str(ex)[:60])
np.save(out, e.read_activation('tower.out', n))
