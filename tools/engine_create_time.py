import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kzero_amd import capi, synth
capi.device_count()
for game, depth, ch, head, batch, dts in [("chess", 20, 256, "attention", 256, ("f16", "split16", "f32")), ("go-19", 40, 256, "conv", 512, ("f16", "split16")), ("ataxx-7", 8, 128, "ataxx_conv", 256, ("f32", "split16"))]:
    blob = synth.random_model(game, depth, ch, head, seed=1)
    t0 = time.time(); model = capi.Model(blob=blob); t1 = time.time()
    print(f"{game} {depth}x{ch}: model parse {t1 - t0:.2f} s")
    for dt in dts:
        code = {"f16": capi.KZ_DTYPE_F16, "f32": capi.KZ_DTYPE_F32, "split16": capi.KZ_DTYPE_F32_SPLIT16}[dt]
        t0 = time.time(); e = capi.Engine(model, 0, batch, code); t1 = time.time()
        e2 = capi.Engine(model, 0, batch, code); t2 = time.time()
        print(f"   {dt:8s} first engine {t1 - t0:.2f} s ({e.tower_path}), second engine on the same weights {t2 - t1:.3f} s")
