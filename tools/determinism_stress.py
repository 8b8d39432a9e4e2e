"""Race screen: every kernel path, many evaluations of the same batch, bitwise-identical outputs expected."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kzero_amd import capi, synth
CASES = [("chess", 20, 256, "attention", 256, capi.KZ_DTYPE_F16, 200),
         ("chess", 4, 256, "attention", 256, capi.KZ_DTYPE_F32, 50),
         ("ataxx-7", 8, 128, "ataxx_conv", 256, capi.KZ_DTYPE_F32, 200),
         ("go-19", 4, 256, "conv", 512, capi.KZ_DTYPE_F16, 50),
         ("go-9", 4, 128, "conv", 512, capi.KZ_DTYPE_F16, 100),
         ("chess", 4, 256, "attention", 255, capi.KZ_DTYPE_F32_SPLIT16, 100),
         ("ataxx-7", 8, 128, "ataxx_conv", 255, capi.KZ_DTYPE_F32_SPLIT16, 200),
         ("chess", 3, 256, "attention", 255, capi.KZ_DTYPE_F16, 200),
         ("ataxx-7", 8, 128, "ataxx_conv", 255, capi.KZ_DTYPE_F32, 300),   # one-launch f32 network, ragged last workgroup
         ("go-9", 6, 128, "conv", 200, capi.KZ_DTYPE_F32, 200),            # the same with the extra-move head, one board per workgroup
         ("go-19", 6, 256, "conv", 509, capi.KZ_DTYPE_F16, 150),           # round 3 board-tile kernel: ring-register prefetch, odd batch
         ("chess", 6, 256, "attention", 251, capi.KZ_DTYPE_F32_SPLIT16, 150),  # round 3: split launch with the heads inside
         ("go-19", 3, 128, "conv", 203, capi.KZ_DTYPE_F32_SPLIT16, 100),       # round 3: per-layer split board-tile kernel
         # round 4: wide stems, the new channel counts, widened towers, the attention head on the matrix cores (f16 / f32),
         # two / four boards per workgroup of the plain-f16 launch, the branch-free heads tail with a wide scalar head
         ("chess-hist-3", 4, 256, "attention", 255, capi.KZ_DTYPE_F16, 150),
         ("chess-hist-2", 4, 256, "attention", 251, capi.KZ_DTYPE_F32_SPLIT16, 100),
         ("chess", 4, 192, "attention", 255, capi.KZ_DTYPE_F16, 150),
         ("chess", 4, 192, "attention", 255, capi.KZ_DTYPE_F32_SPLIT16, 100),
         ("chess", 4, 384, "attention", 255, capi.KZ_DTYPE_F16, 100),
         ("chess", 4, 96, "attention", 255, capi.KZ_DTYPE_F16, 150),
         ("chess", 4, 128, "attention", 301, capi.KZ_DTYPE_F16, 200),
         ("chess", 4, 128, "attention", 255, capi.KZ_DTYPE_F32, 100),
         ("go-9", 4, 128, "conv", 1025, capi.KZ_DTYPE_F16, 150),
         ("ataxx-7", 12, 128, "ataxx_conv", 1021, capi.KZ_DTYPE_F16, 100),
         ("go-19", 4, 256, "conv", 131, capi.KZ_DTYPE_F32_SPLIT16, 60),
         ("go-9", 4, 128, "conv", 1023, capi.KZ_DTYPE_F16, 150),     # two boards per workgroup, heads inside (f16 tail)
         ("go-9", 4, 128, "conv", 77, capi.KZ_DTYPE_F16, 150),       # the same engine shape, narrow launch
         ("go-13", 4, 128, "conv", 255, capi.KZ_DTYPE_F16, 100),     # one 13x13 board in eleven tiles
         ("go-13", 2, 192, "conv", 255, capi.KZ_DTYPE_F16, 100),
         ("go-9", 4, 128, "conv", 2047, capi.KZ_DTYPE_F16, 100),     # round 5: three boards per workgroup (sixteen tiles), ragged
         ("chess", 4, 128, "attention", 2045, capi.KZ_DTYPE_F16, 100),  # ... four 8x8 boards
         # round 5: AttentionTower networks — the matrix-core launch in f16 (two boards per workgroup, an odd board out; one per
         # workgroup) and exact f32, the vector-ALU kernel, the one-launch ScalarHead + AttentionPolicyHead
         ("chess", 6, 256, "attention", 253, capi.KZ_DTYPE_F16, 150),
         ("chess", 6, 256, "attention", 97, capi.KZ_DTYPE_F16, 150),
         ("chess", 4, 256, "attention", 249, capi.KZ_DTYPE_F32, 60),
         ("ataxx-7", 3, 64, "ataxx_conv", 254, capi.KZ_DTYPE_F32, 60)]
KW = {("go-9", 1025): dict(scalar_hidden_channels=8, scalar_hidden_size=128),
      ("chess", 253): dict(attention=(8, 16, 16, 256)), ("chess", 97): dict(attention=(8, 16, 16, 256)),
      ("chess", 249): dict(attention=(8, 16, 16, 256)), ("ataxx-7", 254): dict(attention=(4, 8, 8, 96))}
bad = 0
for game, depth, ch, head, batch, dtype, reps in CASES:
    blob = synth.random_model(game, depth, ch, head, seed=9, **KW.get((game, batch), {}))
    bits, sc = synth.random_boards(game, batch, seed=10)
    engines = [capi.Engine(capi.Model(blob=blob), 0, batch, dtype) for _ in range(2)]
    s0, p0 = engines[0].eval_packed(bits, sc)
    n_bad = 0
    for r in range(reps):
        s, p = engines[r % 2].eval_packed(bits, sc)
        if not (np.array_equal(s, s0) and np.array_equal(p, p0)):
            n_bad += 1
    print(f"{game} {depth}x{ch} { {capi.KZ_DTYPE_F16: 'f16', capi.KZ_DTYPE_F32: 'f32', capi.KZ_DTYPE_F32_SPLIT16: 'f32split16'}[dtype] } path={engines[0].tower_path}: {n_bad} of {reps} differ")
    bad += n_bad
    # round 5: the decoded entry points on all four slots at once (decode_output inside the launch on the one-launch paths,
    # the stand-alone kernel on pinned staging otherwise): the same values and probabilities from every slot, every time
    rng = np.random.default_rng(11)
    plen = engines[0].model.info.policy_len
    moves = [rng.permutation(plen)[:int(c)].astype(np.int32) for c in rng.integers(0, min(plen, 70), size=batch)]
    ref = None
    d_bad = 0
    for r in range(max(4, reps // 10)):
        eng = engines[r % 2]
        offs = [eng.submit_packed_decoded(k, bits, sc, moves) for k in range(capi.KZ_ENGINE_SLOTS)]
        for k in range(capi.KZ_ENGINE_SLOTS):
            v, probs = eng.wait_decoded(k, offs[k])
            flat = np.concatenate([v.ravel()] + [q for q in probs])
            if ref is None:
                ref = flat
            elif not np.array_equal(flat, ref, equal_nan=True):
                d_bad += 1
    print(f"    decoded entry, 4 slots x {max(4, reps // 10)} rounds: {d_bad} differ")
    bad += d_bad
sys.exit(1 if bad else 0)
