// kz_model.hpp — host-side model: KZMODEL1 parse + Conv/BN folding.
// Replaces `load_graph_from_onnx_path` + `optimize_graph` (rust/kz-selfplay/src/server/server_alphazero.rs:126-128)
// for the PredictionHeads(ResTower, ScalarHead, <policy head>) family of python/lib/model/post_act.py.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace kz {

struct Conv {  // nn.Conv2d with square kernel k, weights OIHW, BN (if any) already folded in
    int cout = 0, cin = 0, k = 1;
    std::vector<float> w, b;
};

struct Linear {  // nn.Linear, weights [out][in]
    int out = 0, in = 0;
    std::vector<float> w, b;
};

enum TowerKind { TOWER_RES = 0, TOWER_ATTENTION = 1, TOWER_DENSE_NET = 2 };

// One EncoderLayer (python/lib/model/attention.py:48-136) of the AttentionTower: four bias-free Linear layers, two LayerNorms
// without parameters
struct AttLayer {
    std::vector<float> qkv;  // project_qkv.weight [heads * (2 d_k + d_v)][d_model]: per head q | k | v rows
    std::vector<float> out;  // project_out.weight [d_model][heads * d_v]
    std::vector<float> ff0;  // ff.0.weight [d_ff][d_model]
    std::vector<float> ff1;  // ff.2.weight [d_model][d_ff]
};

enum PolicyKind { POLICY_ATAXX_CONV = 0, POLICY_CONV = 1, POLICY_ATTENTION = 2, POLICY_DENSE = 3, POLICY_ARIMAA = 4,
                  POLICY_NONE = 5 };  // NONE: a DenseNetwork (no PredictionHeads: one Linear yields scalars and policy)

struct Model {
    // architecture descriptor
    std::string game;
    int h = 0, w = 0, n_scalar = 0, n_bool = 0, c_in = 0;
    int depth = 0, channels = 0;
    int policy_len = 0;
    PolicyKind policy_kind = POLICY_CONV;
    int policy_conv_channels = 0, policy_extra_moves = 0, policy_query_channels = 0;
    int dense_hidden_channels = 0, dense_hidden_size = 0;
    int arimaa_hidden_channels = 0, arimaa_hidden_size = 0;  // ArimaaPolicyHead's scalar branch (post_act.py:155-162)

    // ResTower (post_act.py:201-211): tower[0] = stem; tower[2i-1], tower[2i] = block i conv A / conv B,
    // each with its BatchNorm folded (W' = s*W, b' = s*b + t).  The tower's final BatchNorm stays a per-channel
    // affine (final_scale, final_shift): it is applied after the last residual add, so it cannot be folded
    // backwards; the executor fuses it into the last conv's epilogue.
    std::vector<Conv> tower;
    std::vector<float> final_scale, final_shift;

    // AttentionTower (python/lib/model/attention.py:8-45; the tower python/main/supervised_main_alpha.py:72 trains) instead of
    // the ResTower: every square a token of `channels` = d_model features, `depth` encoder layers.  expand [d_model][c_in]
    // (bias-free Linear over the input planes), embedding [h*w][d_model] added per square; DeepNorm residuals
    // LayerNorm(x * alpha + f(x)) with alpha = (2 depth)^(1/4) (:22, :126, :129).  final_scale / final_shift stay (1, 0).
    TowerKind tower_kind = TOWER_RES;
    int att_heads = 0, att_dk = 0, att_dv = 0, att_dff = 0;
    float att_alpha = 1.0f, ln_eps = 1e-5f;
    std::vector<float> att_expand, att_embedding;
    std::vector<AttLayer> att_layers;

    // ScalarHead (post_act.py:10-23)
    Conv sh_conv;
    Linear sh_fc0, sh_fc1;

    // policy heads
    Conv p_conv0, p_conv1;       // ataxx_conv / conv: seq.0, seq.2     dense: seq.0 (optional)
    Conv p_extra_conv;           // conv: seq_extra.0
    Linear p_extra_fc;           // conv: seq_extra.2
    Conv p_bulk, p_under;        // attention
    std::vector<int32_t> flat_to_att;
    Linear p_fc0, p_fc1;         // dense: optional hidden Linear, final Linear
    // arimaa (ArimaaPolicyHead, post_act.py:144-173): bulk = p_conv0 (C -> C) + ReLU + p_conv1 (C -> 4), flattened channel-major
    // BEHIND the scalar branch's 1 + 6 outputs: pa_conv (C -> hc) + ReLU, Flatten, pa_fc0 (hc*hw -> hs) + ReLU, pa_fc1 (hs -> 7)
    Conv pa_conv;
    Linear pa_fc0, pa_fc1;

    // DenseNetwork(game, depth, size, res) (python/lib/model/simple.py:7-52; the reference's own test networks,
    // python/main/write_test_networks.py:14-18) — the WHOLE network, no tower / heads: Flatten (channel-major), Linear dn_in,
    // `depth` DenseBlocks (BatchNorm1d as y = s x + t, ReLU, Linear, BatchNorm1d, ReLU, Linear; x + y when dn_res), BatchNorm1d,
    // ReLU, Linear dn_out to 5 + policy_len.  tower_kind == TOWER_DENSE_NET, channels = size, policy_kind = POLICY_NONE.
    struct DnBlock {
        std::vector<float> sa, ta, sb, tb;
        Linear la, lb;
    };
    bool dn_res = false;
    Linear dn_in, dn_out;
    std::vector<DnBlock> dn_blocks;
    std::vector<float> dn_sf, dn_tf;

    int64_t param_count = 0;
    double flops_per_eval = 0;
};

// Returns nullptr and sets `err` on failure.
Model *parse_model(const void *blob, size_t len, std::string &err);

// The same network with its tower widened to `cpad` channels by all-zero filters (same outputs: the new channels carry
// zeros from the stem to the heads).  param_count and flops_per_eval stay the original network's.  The engine widens a
// tower whose channel count is not a multiple of 64 (48, 96, 160 ...) to the next one, so that it runs on the one-launch
// and board-tile kernels instead of the generic implicit GEMM.  nullptr when cpad <= channels.
Model *pad_channels(const Model &m, int cpad);

// ONNX as exported by the trainer (python/lib/save_onnx.py:60-122), kz_onnx.cpp.  n_scalar = how many of the input
// planes are broadcast scalars (InputMapper::input_scalar_count, rust/kz-core/src/mapping/mod.rs:21) — the graph does
// not carry that split; pass -1 when unknown (then only the dense-input entry points work).
bool looks_like_onnx(const void *blob, size_t len);
Model *parse_onnx(const void *blob, size_t len, int n_scalar, std::string &err);

}  // namespace kz
