"""The shape-generality sweep: networks OUTSIDE the five benchmark shapes, shared by tests/test_shape_sweep.py (parity
against the oracle, `-m gpu`) and tools/shape_sweep.py (path x evals/s table of DESIGN §5.0).

The reference builds `ResTower(depth, C_in, channels)` for any channel count (python/lib/model/post_act.py:201-211), its
server accepts Ataxx 2..8, Go of any size with or without territory planes, chess and chess with 1..n history boards
(rust/kz-core/src/mapping/chess.rs:32-95: 21 + 13 L input planes), and any of the four policy heads (post_act.py:26-141).

A case = (id, game, depth, channels, head, kwargs of synth.random_model, oracle boards).  `boards` is ragged on purpose
(never a multiple of a launch's boards per workgroup).
"""
from collections import namedtuple

Case = namedtuple("Case", "id game depth channels head kw boards")


def _c(game, depth, channels, head, boards=5, **kw):
    tag = "".join(f"_{k[6:] if k.startswith('dense_') and k != 'dense_network' else k}{v}" for k, v in sorted(kw.items()) if k != "attention")
    if "attention" in kw:  # (heads, d_k, d_v, d_ff) of an AttentionTower in place of the ResTower
        tag += "_att%dh%dk%dv%df" % kw["attention"]
    return Case(f"{game}_{depth}x{channels}_{head}{tag}", game, depth, channels, head, kw, boards)


CASES = [
    # --- tower channels off the lattice, chess, attention head with query_channels = channels
    #     (python/main/supervised_main_alpha.py:76) ---
    *[_c("chess", 3, c, "attention") for c in (32, 48, 96, 160, 192, 320, 384, 512)],
    _c("chess", 1, 192, "attention", boards=3),
    _c("chess", 1, 256, "attention", query_channels=64, boards=7),
    # --- input planes: GoStdMapper without territory (10), ChessStd (21), ChessHistoryMapper lengths 1..3 (34, 47, 60) ---
    _c("go-9-noterr", 3, 128, "conv"),
    _c("chess-hist-1", 3, 256, "attention"),
    _c("chess-hist-2", 1, 256, "attention", boards=3),
    _c("chess-hist-3", 3, 256, "attention", boards=7),
    _c("chess-hist-1", 3, 128, "dense", dense_hidden_channels=8),
    _c("chess-hist-3", 1, 64, "attention"),
    _c("chess-hist-2", 3, 192, "attention"),
    # --- boards ---
    *[_c(f"ataxx-{n}", 3, 128, "ataxx_conv", boards=7) for n in (4, 5, 6, 7, 8)],
    _c("ataxx-7", 1, 64, "ataxx_conv", boards=3),
    _c("ataxx-7", 3, 192, "ataxx_conv"),
    _c("ataxx-8", 3, 256, "ataxx_conv"),
    _c("go-9", 3, 256, "conv"),
    _c("go-9", 1, 96, "conv"),
    _c("go-9", 3, 192, "conv"),
    _c("go-13", 3, 128, "conv", boards=3),
    _c("go-13", 3, 256, "conv", boards=3),
    _c("go-13", 1, 192, "conv", boards=3),
    _c("go-19", 3, 128, "conv", boards=2),
    _c("go-19", 1, 192, "conv", boards=3),
    _c("go-19", 3, 320, "conv", boards=2),
    _c("go-19-noterr", 3, 64, "conv", boards=2),
    # --- dense policy heads (post_act.py:26-51) with and without hidden layers, behind the resident towers ---
    _c("chess", 3, 128, "dense"),
    _c("chess", 3, 256, "dense"),
    _c("chess", 3, 256, "dense", dense_hidden_channels=16),
    _c("chess", 1, 128, "dense", dense_hidden_channels=8, dense_hidden_size=256),
    _c("chess", 3, 256, "dense", dense_hidden_size=128),
    _c("ataxx-7", 3, 128, "dense", dense_hidden_channels=4),
    _c("go-9", 3, 128, "dense", dense_hidden_channels=2, dense_hidden_size=64),
    _c("chess", 3, 192, "dense", dense_hidden_channels=8),
    # --- other head shapes on the lattice towers ---
    _c("chess", 3, 128, "attention"),
    _c("chess", 3, 64, "attention"),
    _c("chess", 3, 256, "attention", scalar_hidden_size=64),
    # the other head sizes the reference's main scripts build: ScalarHead(board, channels, 8, 128) with
    # DensePolicyHead(game, channels, 32, None) (python/main/supervised_main_mu.py:80-81) and a tower whose final BatchNorm
    # has no affine (ResTower(..., final_affine=False), python/main/loop_main_mu.py:78)
    _c("chess", 3, 256, "dense", dense_hidden_channels=32, scalar_hidden_channels=8, scalar_hidden_size=128),
    _c("go-9", 3, 128, "conv", scalar_hidden_channels=8, scalar_hidden_size=128),
    _c("ataxx-7", 3, 128, "ataxx_conv", final_affine=False),
    # --- round 5: the other games the server dispatches (rust/kz-selfplay/src/server/server.rs:114-185): arimaa-split
    #     (ArimaaSplitMapper: 38 input planes, ArimaaPolicyHead, post_act.py:144-173), ttt (3x3) and sttt (9x9) through
    #     DensePolicyHead ---
    _c("arimaa-split", 3, 256, "arimaa"),
    _c("arimaa-split", 3, 128, "arimaa", arimaa_hidden_channels=4, arimaa_hidden_size=64, boards=7),
    _c("arimaa-split", 1, 96, "arimaa", boards=3),
    _c("ttt", 3, 64, "dense", boards=7),
    _c("ttt", 1, 32, "dense", dense_hidden_size=16, boards=3),
    _c("sttt", 3, 128, "dense", dense_hidden_channels=2),
    _c("sttt", 3, 64, "dense", dense_hidden_channels=4, dense_hidden_size=32, boards=7),
    # --- round 5: PredictionHeads(AttentionTower, ...) (python/lib/model/attention.py:8-136; the tower
    #     python/main/supervised_main_alpha.py:72 trains: d_model 256, 8 heads, d_k = d_v = 16, d_ff 256): the matrix-core
    #     launch's shapes in both arithmetics, and shapes only the vector-ALU kernel takes (other head counts and sizes,
    #     boards that are not 8x8) ---
    _c("chess", 3, 256, "attention", attention=(8, 16, 16, 256), boards=7),
    _c("chess", 1, 128, "dense", attention=(8, 16, 16, 128), dense_hidden_channels=8, boards=3),
    _c("chess-hist-1", 1, 256, "attention", attention=(8, 16, 16, 256), boards=3),
    _c("chess", 3, 256, "attention", attention=(8, 16, 16, 512), boards=3),
    _c("chess", 3, 64, "attention", attention=(4, 16, 16, 96)),
    _c("ataxx-7", 3, 64, "ataxx_conv", attention=(4, 8, 8, 96), boards=7),
    _c("go-9", 1, 128, "conv", attention=(8, 16, 16, 128), boards=3),
    _c("sttt", 1, 64, "dense", attention=(2, 16, 24, 64), dense_hidden_channels=2, boards=3),
    # --- DenseNetwork(game, depth, size, res) (python/lib/model/simple.py:7-52): the whole network an MLP ---
    _c("sttt", 1, 64, "none", dense_network=False, boards=7),
    _c("chess", 3, 256, "none", dense_network=True, boards=3),
    _c("go-9", 2, 100, "none", dense_network=True),
]

# the reference's own shipping configuration (python/main/loop_main_alpha.py:16-30,68-76): Go 9x9, 16 blocks x 128
# channels, ConvPolicyHead(extra_moves=1), gpu_batch_size 2048
REFERENCE_LOOP = dict(game="go-9", depth=16, channels=128, head="conv", batch=2048)


def parity_dtype_name(model, capi):
    """What `KZ_HIP_DTYPE=parity` (the Rust binding's default, kzero_amd/rust/hip.rs) creates for this model."""
    return "f32split16" if model.supports_dtype(capi.KZ_DTYPE_F32_SPLIT16) else "f32"
