import sys, time, numpy as np
sys.path.insert(0, '.')
from kzero_amd import capi, synth
for game, depth, ch, head in [("chess", 10, 128, "attention"), ("chess", 20, 128, "attention"), ("go-9", 10, 128, "conv"), ("ataxx-7", 4, 64, "ataxx_conv")]:
    blob = synth.random_model(game, depth, ch, head, seed=1)
    model = capi.Model(blob=blob)
    B = 256
    bits, sc = synth.random_boards(game, B, seed=2)
    for dtype, name in [(capi.KZ_DTYPE_F16, "f16"), (capi.KZ_DTYPE_F32_SPLIT16, "f32split16"), (capi.KZ_DTYPE_F32, "f32")]:
        engines = [capi.Engine(model, 0, B, dtype) for _ in range(3)]
        d_bits = capi.DeviceBuffer.from_host(0, bits); d_sin = capi.DeviceBuffer.from_host(0, sc)
        outs = [(capi.DeviceBuffer(0, B * 5 * 4), capi.DeviceBuffer(0, B * model.info.policy_len * 4)) for _ in engines]
        def run(n):
            for i in range(n):
                e = i % 3
                engines[e].enqueue_packed_device(d_bits, bits.shape[1], d_sin, B, outs[e][0], outs[e][1])
            for e in engines: e.synchronize()
        run(300)
        t = time.perf_counter(); n = 3000; run(n); dt = time.perf_counter() - t
        print(f"{game} {depth}x{ch} {name}: {n * B / dt:,.0f} evals/s  path={engines[0].tower_path}")
