// kz_plan.hpp — which kernels run a network.  Part of kz_engine.hip's translation unit (included inside its anonymous
// namespace); exported through kz_model_plan and kz_model_supports_dtype.
#pragma once

// ------------------------------------------------------------------------------------------------
// Which kernels run a network: pure host logic over the kernels' support predicates (no HIP call), shared by
// kz_engine_create and kz_model_plan — DESIGN.md §5.0 prints its table from it and tests/test_path_table.py holds it to
// tests/golden/path_table.json without a GPU.
//
//   dtype f16:         8x8, 256 channels, <= 224 planes          -> tower_resident_f16 [+heads: attention head, Q = 256]
//                      else a shape of kz_tower_split.hip        -> tower_resident_f16g [+heads: conv heads at 128 / 256]
//                      else channels % 64 == 0, >= 160 workgroups -> board_conv_f16          (one launch per layer)
//                      else                                       -> conv_igemm_f16          (one launch per layer)
//   dtype f32:         128 / 256 channels on a small board        -> tower_resident_f32 [+heads: conv heads]
//                      else                                       -> conv_igemm_f32
//   dtype f32split16:  a shape of kz_tower_split.hip (split)      -> tower_resident_split16 [+heads]
//                      else channels % 64 == 0                    -> board_conv_split16      (one launch per layer)
//                      else                                       -> refused (kz_model_supports_dtype says 0)
// ------------------------------------------------------------------------------------------------
//   AttentionTower (attention.py) instead of the ResTower:
//     8x8, 8 heads of d_k = d_v = 16, d_model 128 / 256: dtype f16 -> attention_tower_f16, f32 -> attention_tower_f32 (the same
//                                                       launch on v_mfma_f32_16x16x4_f32; d_ff <= 256)
//     any other shape (f32, or f16 rows around f32 arithmetic)   -> attention_tower_f32_valu (one launch, vector ALUs)
//     f32split16                                                 -> refused
//   DenseNetwork (simple.py: no tower, no heads), f32 / f16         -> dense_network_f32 (one launch, f32 arithmetic)
struct PathPlan {
    bool dense_net = false;  // Model::tower_kind == TOWER_DENSE_NET: kz_dense_network.hip runs the whole network
    bool att_tower = false;  // Model::tower_kind == TOWER_ATTENTION: kz_att_tower.hip
    bool att_f16 = false;    // ... on the matrix cores (f16 or exact f32): kz_att_tower_mfma.hip
    bool resident = false, fused_heads = false, resident32 = false, split16 = false, bsplit = false, pairs16 = false;
    bool fused32 = false, fused_split = false, fused_pairs = false, board_conv = false, keep = false;
    bool wide = false;  // tower_resident_f16g with twice the boards per workgroup (kz::tower_split_wide_supported)
    std::string path;
    int launches = 0;  // kernel launches per batch through the packed-input entry points
};

bool env_on(const char *name) {
    const char *v = getenv(name);
    return v && v[0] == '1';
}

// the f16 engines' one-launch ScalarHead + AttentionPolicyHead (kz_att_heads.hip)
bool att_heads_one_launch(const Model &m, int dtype, bool split16) {
#ifdef KZ_EXPERIMENTS
    if (env_on("KZ_NO_ATT_HEADS")) return false;  // (A/B against the four separate launches)
#endif
    return m.policy_kind == kz::POLICY_ATTENTION && dtype == KZ_DTYPE_F16 && !split16 &&
           kz::att_heads_supported(dtype, m.h, m.w, m.channels, m.policy_query_channels, m.sh_conv.cout, m.sh_fc0.out, m.policy_len);
}

// launches of run_heads for a network whose tower output is materialised (head convolutions with cout_p = round_up(cout, 32))
int head_launches(const Model &m, int dtype, bool split16, int cp) {
    if (att_heads_one_launch(m, dtype, split16)) return 1;
    int n = 1;  // kz_scalar_head
    const bool f16_heads = split16 || dtype == KZ_DTYPE_F16;  // 1x1 head convolutions through kz_conv1x1_split where it fits
    switch (m.policy_kind) {
        case kz::POLICY_ATAXX_CONV:
        case kz::POLICY_CONV: {
            const int c0_in = round_up(m.p_conv0.cin, 32), c0_out = round_up(m.p_conv0.cout, 32);
            const bool one = m.policy_kind == kz::POLICY_CONV && f16_heads && kz::conv1x1_split_supported(c0_in, c0_out) && cp >= c0_in &&
                             kz::conv1x1_policy_epilogue_supported(c0_in, c0_out, m.p_conv0.cout, m.policy_conv_channels);
            n += one ? 1 : 2;
            if (m.policy_kind == kz::POLICY_CONV && m.policy_extra_moves) {
                const bool in_scalar_head = m.sh_conv.cout == 4 && m.p_extra_conv.cout == 1 && m.p_extra_conv.cin == m.sh_conv.cin &&
                                            kz::scalar_head_takes_extra(dtype == KZ_DTYPE_F32 || split16 ? 0 : 1, cp, m.sh_conv.cout);
                n += in_scalar_head ? 0 : 1;
            }
            break;
        }
        case kz::POLICY_ATTENTION: n += 3; break;
        case kz::POLICY_ARIMAA: n += 3; break;  // bulk: two 1x1 convolutions; the scalar branch through kz_scalar_head
        case kz::POLICY_DENSE: n += (m.dense_hidden_channels ? 1 : 0) + (m.dense_hidden_size ? 2 : 1); break;
        case kz::POLICY_NONE: return 0;
    }
    return n;
}

// dtype_in: KZ_DTYPE_F32 / KZ_DTYPE_F16 / KZ_DTYPE_F32_SPLIT16.  false + why: kz_engine_create refuses.
bool plan_path(const Model &m, int max_batch, int dtype_in, PathPlan &p, std::string &why) {
    const bool split16 = dtype_in == KZ_DTYPE_F32_SPLIT16;
    const int dtype = split16 ? KZ_DTYPE_F32 : dtype_in;  // KZ_DTYPE_F32_SPLIT16 is the f32 engine with one kernel exchanged
    const int cp = round_up(m.channels, 32);
    const bool force = env_on("KZ_FORCE_GENERIC"), nofuse = env_on("KZ_NO_FUSED_HEADS"), noboard = env_on("KZ_NO_BOARD_CONV");
    p = PathPlan();
    if (m.tower_kind == kz::TOWER_DENSE_NET) {  // DenseNetwork (simple.py): the whole network is one launch behind the encode
        if (split16) {
            why = "KZ_DTYPE_F32_SPLIT16 has no DenseNetwork kernel: such a network runs in f32 arithmetic (KZ_DTYPE_F32, or KZ_DTYPE_F16 rows around it)";
            return false;
        }
        if (!kz::dense_network_supported(m.h, m.w, m.c_in, m.channels, m.depth, m.policy_len)) {
            why = "dense network: the board's input vector and the hidden vectors do not fit the LDS of one workgroup";
            return false;
        }
        p.dense_net = true;
        p.path = "dense_network_f32";
        p.launches = 2;
        return true;
    }
    if (m.tower_kind == kz::TOWER_ATTENTION) {
        if (split16) {
            why = "KZ_DTYPE_F32_SPLIT16 has no attention-tower kernel: an AttentionTower network runs as KZ_DTYPE_F32 (exact) or KZ_DTYPE_F16";
            return false;
        }
        if (!kz::att_tower_supported(m.h, m.w, m.c_in, m.channels, m.att_heads, m.att_dk, m.att_dv, m.att_dff, m.depth)) {
            why = "attention tower: the token matrix of one board (squares x d_model) with one head's q, k, v and all heads' "
                  "outputs beside it does not fit the 160 KB of LDS of one workgroup, or the board has more than 384 squares";
            return false;
        }
        p.att_tower = true;
        // (att_f16: the matrix-core launch, in the engine's arithmetic — f16, or exact f32)
        p.att_f16 = !force && kz::att_tower16_supported(m.h, m.w, m.c_in, m.channels, m.att_heads, m.att_dk, m.att_dv, m.att_dff, m.depth,
                                                        dtype == KZ_DTYPE_F32);
        p.path = !p.att_f16 ? "attention_tower_f32_valu" : dtype == KZ_DTYPE_F32 ? "attention_tower_f32" : "attention_tower_f16";
        p.launches = (p.att_f16 ? 1 : 2) + head_launches(m, dtype, false, cp);  // (encode,) the tower, the heads
        return true;
    }
    p.resident = kz::tower_resident_supported(dtype, m.h, m.w, m.channels, m.depth, m.c_in) && !force;
    p.fused_heads = p.resident && !nofuse &&
                    kz::tower_heads_supported((int)m.policy_kind, m.policy_query_channels, m.policy_len, m.sh_conv.cout, m.sh_fc0.out);
    // the board-tile kernel needs enough workgroups to fill the chip (two per CU when it is busy)
    const bool board_conv_ok = !p.resident && !noboard && m.depth >= 1 &&
                               kz::board_conv_supported(dtype, m.h, m.w, m.channels, m.channels) &&
                               kz::board_conv_workgroups(max_batch, m.h, m.w, m.channels) >= 160 &&
                               (size_t)max_batch * m.h * m.w * m.channels * 2 < ((size_t)1 << 31);  // 32-bit buffer offsets
    p.keep = env_on("KZ_KEEP_ACTIVATIONS") && !p.resident;
    // exact-f32 resident launch (the per-layer activation taps of KZ_KEEP_ACTIVATIONS need the per-layer path)
    p.resident32 = kz::tower32_supported(dtype, m.h, m.w, m.channels, m.depth) && !force && !p.keep;
    // split arithmetic per layer for boards the resident split launch cannot hold (Go 19x19)
    const bool split_resident = kz::tower_split_supported(m.h, m.w, m.channels, m.depth, m.c_in, true);
    const bool board_split_ok = split16 && !split_resident && m.depth >= 1 && !p.keep &&
                                kz::board_conv_split_supported(m.h, m.w, m.channels, m.channels) && m.channels % 32 == 0 &&
                                (size_t)max_batch * m.h * m.w * m.channels * 4 < ((size_t)1 << 31);
    if (board_split_ok) {
        p.split16 = p.bsplit = true;
        p.resident32 = false;  // (a shape the exact-f32 launch takes too stays per layer here)
    } else if (split16) {
        if (!split_resident) {
            why = "KZ_DTYPE_F32_SPLIT16 needs a shape of the one-launch split tower (256 tower channels on a board of at most 64 "
                  "squares, 192 on at most 64, 64 / 128 channels on at most 96 squares, and no more input planes than tower "
                  "channels); or, per layer, tower channels a multiple of 64, at least one block and max_batch * squares * "
                  "channels * 4 bytes < 2 GiB";
            return false;
        }
        p.split16 = p.resident32 = true;  // same tensors in and out as the exact-f32 resident launch
    }
    // plain-f16 board-resident tower for the shapes the chess launch (kz_tower.hip) does not take: the split kernel
    // without its lo halves
    p.pairs16 = dtype == KZ_DTYPE_F16 && !p.resident && !force && !p.keep && !env_on("KZ_NO_RESIDENT_F16G") &&
                kz::tower_split_supported(m.h, m.w, m.channels, m.depth, m.c_in, false);
    p.board_conv = board_conv_ok && !p.pairs16 && !p.bsplit;
    p.wide = p.pairs16 && kz::tower_split_wide_supported(m.h, m.w, m.channels, max_batch);
    p.fused_pairs = p.pairs16 && !nofuse &&
                    kz::tower_split_conv_heads_supported((int)m.policy_kind, m.policy_extra_moves, m.policy_conv_channels, m.h, m.w,
                                                         m.channels, m.sh_conv.cout, m.sh_fc0.out, false, p.wide ? max_batch : 0);
    // the wide tiles hold more boards per workgroup than the fused conv heads' tail takes (four): where the heads fit the
    // narrow tiles only (128 channels on 5x5: eight boards against four), one launch per batch — zero-copy slots, the decode
    // inside — is worth more than the wide tiles' smaller weight traffic
    if (p.wide && !p.fused_pairs && !nofuse &&
        kz::tower_split_conv_heads_supported((int)m.policy_kind, m.policy_extra_moves, m.policy_conv_channels, m.h, m.w, m.channels,
                                             m.sh_conv.cout, m.sh_fc0.out, false, 0)) {
        p.wide = false;
        p.fused_pairs = true;
    }
    p.fused32 = p.resident32 && !p.split16 && !nofuse &&
                kz::tower32_heads_supported((int)m.policy_kind, m.policy_extra_moves, m.policy_conv_channels, m.h, m.w, m.channels,
                                            m.sh_conv.cout, m.sh_fc0.out);
    p.fused_split = p.split16 && !p.bsplit && !nofuse &&
                    (kz::tower_split_heads_supported((int)m.policy_kind, m.policy_query_channels, m.policy_len, m.h, m.w, m.channels,
                                                     m.sh_conv.cout, m.sh_fc0.out) ||
                     kz::tower_split_conv_heads_supported((int)m.policy_kind, m.policy_extra_moves, m.policy_conv_channels, m.h, m.w,
                                                          m.channels, m.sh_conv.cout, m.sh_fc0.out, true));
    p.path = p.fused_heads   ? "tower_resident_f16+heads"
             : p.resident    ? "tower_resident_f16"
             : p.bsplit      ? "board_conv_split16"
             : p.fused_split ? "tower_resident_split16+heads"
             : p.split16     ? "tower_resident_split16"
             : p.fused32     ? "tower_resident_f32+heads"
             : p.resident32  ? "tower_resident_f32"
             : p.fused_pairs ? "tower_resident_f16g+heads"
             : p.pairs16     ? "tower_resident_f16g"
             : p.board_conv  ? "board_conv_f16"
                             : (dtype == KZ_DTYPE_F32 ? "conv_igemm_f32" : "conv_igemm_f16");
    const bool fused = p.fused_heads || p.fused32 || p.fused_split || p.fused_pairs;
    const bool one_launch_tower = p.resident || p.resident32 || p.pairs16;  // (board encode inside)
    p.launches = fused ? 1
                 : (one_launch_tower ? 1 : p.bsplit ? 3 + 2 * m.depth : 2 + 2 * m.depth) + head_launches(m, dtype, p.split16, cp);
    return true;
}
