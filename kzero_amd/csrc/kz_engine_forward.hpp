// kz_engine_forward.hpp — the forward pass of an engine: which kernels run for a network on its path (PathPlan, kz_plan.hpp),
// in which order, on the engine's current stream.  Out-of-class definitions of the members kz_engine_state.hpp declares;
// included ONCE, by kz_engine.hip.
#pragma once

inline int kz_engine::conv(const DevConv &w, const void *x, int ldx, void *y, int ldy, int M, int relu, const void *res, bool post,
         int h, int wd, int group, int src_group, int src_off, float *y32, int ldy32) {
    if (w.bws) {  // whole boards as LDS-resident spatial tiles, split arithmetic: (hi, lo) rows of 2 C halves
        kz::BoardConvArgs b{};
        b.x = x; b.ldx = 2 * ldx; b.weights = w.bws; b.bias = w.b; b.res = res; b.y = y; b.ldy = 2 * ldy;
        b.y32 = y32; b.ldy32 = ldy32;
        b.post_scale = post ? wts->post_scale : nullptr;
        b.post_shift = post ? wts->post_shift : nullptr;
        b.boards = M / (h * wd); b.h = h; b.w = wd; b.cin = w.cin_p; b.cout = w.cout; b.relu = relu;
        prof.begin("kz_board_conv_split16", stream);
        kz::launch_board_conv_split(b, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (w.bw) {  // whole boards as LDS-resident spatial tiles
        kz::BoardConvArgs b{};
        b.x = x; b.ldx = ldx; b.weights = w.bw; b.bias = w.b; b.res = res; b.y = y; b.ldy = ldy;
        b.post_scale = post ? wts->post_scale : nullptr;
        b.post_shift = post ? wts->post_shift : nullptr;
        b.boards = M / (h * wd); b.h = h; b.w = wd; b.cin = w.cin_p; b.cout = w.cout; b.relu = relu;
        b.rowmap = wts->bc_rowmap; b.halo = wts->bc_halo; b.n_halo = wts->bc_n_halo;
        prof.begin("kz_board_conv_f16", stream);
#ifdef KZ_EXPERIMENTS
        if (w.bw2) kz::launch_board_conv2(b, stream);
        else
#endif
        kz::launch_board_conv(b, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (w.sw && y && !res && !post && !y32 && ldx >= w.cin_p) {  // 1x1 head convolution: split16 or any f16 path
        kz::Conv1x1SplitArgs c{};
        c.split = split16;
        c.x = x; c.ldx = ldx; c.weights = w.sw; c.bias = w.b; c.y = y; c.ldy = ldy;
        c.M = M; c.cin_p = w.cin_p; c.cout_p = w.cout_p; c.relu = relu;
        c.group = group; c.src_group = src_group; c.src_off = src_off;
        prof.begin("kz_conv1x1_split", stream);
        kz::launch_conv1x1_split(c, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    kz::ConvArgs a{};
    a.x = x; a.ldx = ldx; a.w = w.w; a.bias = w.b; a.res = res; a.ldres = ldy;
    a.post_scale = post ? wts->post_scale : nullptr;
    a.post_shift = post ? wts->post_shift : nullptr;
    a.y = y; a.y32 = y32; a.ldy = ldy; a.ldy32 = ldy32;
    a.M = M; a.h = h; a.w_ = wd; a.group = group; a.src_group = src_group; a.src_off = src_off;
    a.cin_p = w.cin_p; a.cout_p = w.cout_p; a.cout = w.cout; a.k = w.k; a.relu = relu;
    prof.begin(kz::conv_kernel_name(dtype), stream);
    kz::launch_conv(dtype, a, stream);
    prof.end(stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

// packed != nullptr (resident path only): the launch encodes the boards itself.  The one-launch networks ("...+heads") can
// end in decode_output (kz_decode_dev.hpp): with `dec` nothing but the decoded values and the available moves' probabilities
// leave the launch
inline int kz_engine::run_tower(int batch, float *d_scalars, float *d_policy, const PackedIn *packed,
              const kz::DecodeArgs *dec) {
    const Model &m = *model;
    const int hw = m.h * m.w, M = batch * hw;
    if (dense_net) {  // DenseNetwork: encoded planes in x_in -> scalars and policy, one launch
        kz::DenseNetArgs t{};
        t.x0 = x_in; t.in_f16 = dtype == KZ_DTYPE_F16; t.batch = batch; t.hw = hw; t.cin_p = cin_p; t.size = m.channels;
        t.depth = m.depth; t.res = m.dn_res ? 1 : 0; t.policy_len = m.policy_len;
        t.w_in = wts->dn_w_in; t.b_in = wts->dn_b_in; t.blocks = wts->dn_blocks; t.sf = wts->dn_sf; t.tf = wts->dn_tf;
        t.w_out = wts->dn_w_out; t.b_out = wts->dn_b_out; t.scalars = d_scalars; t.policy = d_policy;
        t.nonfinite_flag = nf_flag; t.epoch = nf_epoch;
        prof.begin("kz_dense_network", stream);
        kz::launch_dense_network(t, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        tower_out = 0;
        return 0;
    }
    if (att_f16) {  // AttentionTower on the matrix cores, in the engine's arithmetic
        kz::AttTower16Args t{};
        t.f32 = dtype == KZ_DTYPE_F32;
        t.x0 = x_in; t.cin_p = cin_p; t.w_expand = wts->att16_expand; t.embedding = wts->att_embedding;
        t.w_layers = wts->att16_layers; t.y = act[0]; t.batch = batch; t.depth = m.depth; t.d_model = m.channels;
        t.d_ff = m.att_dff; t.alpha = m.att_alpha; t.eps = m.ln_eps;
        if (packed) {  // fused board encode
            t.bits = (const uint8_t *)packed->bits;
            t.bits_stride = packed->stride;
            t.scalars_in = (const float *)packed->scalars;
            t.n_scalar = m.n_scalar;
            t.n_bool = m.n_bool;
        }
        prof.begin(t.f32 ? "kz_att_tower_f32" : "kz_att_tower_f16", stream);
        kz::launch_att_tower16(t, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        tower_out = 0;
        return 0;
    }
    if (att_tower) {  // AttentionTower: encoded planes in x_in -> tower output rows in act[0], one launch
        kz::AttTowerArgs t{};
        t.x0 = x_in; t.ldx0 = cin_p; t.in_f16 = dtype == KZ_DTYPE_F16; t.c_in = m.c_in;
        t.expand = wts->att_expand; t.embedding = wts->att_embedding; t.layers = wts->att_layers;
        t.y = act[0]; t.ldy = cp; t.out_f16 = dtype == KZ_DTYPE_F16;
        t.batch = batch; t.h = m.h; t.w = m.w; t.depth = m.depth; t.d_model = m.channels; t.heads = m.att_heads;
        t.d_k = m.att_dk; t.d_v = m.att_dv; t.d_ff = m.att_dff; t.alpha = m.att_alpha; t.eps = m.ln_eps;
        prof.begin("kz_att_tower_f32_valu", stream);
        kz::launch_att_tower(t, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        tower_out = 0;
        return 0;
    }
    if (resident) {
        kz::TowerArgs t{};
        if (packed) {
            t.bits = (const uint8_t *)packed->bits;
            t.bits_stride = packed->stride;
            t.scalars_in = (const float *)packed->scalars;
            t.n_scalar = m.n_scalar;
            t.n_bool = m.n_bool;
        }
        t.x0 = x_in; t.cin_p = cin_p; t.w_stem = wts->res_w_stem; t.w_tower = wts->res_w_tower;
        t.bias = wts->res_bias; t.post_scale = wts->post_scale; t.post_shift = wts->post_shift;
        t.y = act[0]; t.batch = batch; t.h = m.h; t.w = m.w; t.depth = m.depth;
        t.fused_heads = fused_heads;
        t.sh_w0 = wts->sh_w0; t.sh_b0 = wts->sh_b0; t.sh_w1 = wts->sh_w1; t.sh_b1 = wts->sh_b1;
        t.sh_w2 = wts->sh_w2; t.sh_b2 = wts->sh_b2; t.att_idx = wts->att_idx;
        t.scalars = d_scalars; t.policy = d_policy;
        t.nonfinite_flag = nf_flag; t.epoch = nf_epoch;
        if (dec && fused_heads) t.decode = *dec;
        prof.begin("kz_tower_resident_f16", stream);
#ifdef KZ_EXPERIMENTS
        if (nb4) kz::launch_tower_resident4(t, xres, stream);
        else
#endif
        kz::launch_tower_resident(t, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        tower_out = 0;
        return 0;
    }
    if (resident32 || pairs16) {
        kz::Tower32Args t{};
        t.x0 = (const float *)x_in; t.ldx0 = cin_p; t.c_in = m.c_in; t.weights = wts->res32_w; t.bias = wts->res_bias;
        t.post_scale = wts->post_scale; t.post_shift = wts->post_shift;
        t.y = (float *)act[0]; t.ldy = cp; t.batch = batch; t.h = m.h; t.w = m.w; t.channels = m.channels;
        t.depth = m.depth;
        if (packed) {  // fused board encode
            t.bits = (const uint8_t *)packed->bits;
            t.bits_stride = packed->stride;
            t.scalars_in = (const float *)packed->scalars;
            t.n_scalar = m.n_scalar;
            t.n_bool = m.n_bool;
        }
        if (fused_split && m.policy_kind == kz::POLICY_ATTENTION) {
            kz::Tower32Args::Heads &hd = t.heads;
            hd.on = true;
            hd.sh_w0 = wts->sh_w0; hd.sh_b0 = wts->sh_b0; hd.sh_w1 = wts->sh_w1; hd.sh_b1 = wts->sh_b1;
            hd.sh_w2 = wts->sh_w2; hd.sh_b2 = wts->sh_b2; hd.att_idx = wts->att_idx;
            hd.policy_len = m.policy_len;
            hd.scalars = d_scalars; hd.policy = d_policy;
            hd.nonfinite_flag = nf_flag; hd.epoch = nf_epoch;
        }
        if (fused32 || fused_pairs || (fused_split && m.policy_kind != kz::POLICY_ATTENTION)) {  // conv policy heads: the f32 tail
            kz::Tower32Args::Heads &hd = t.heads;
            hd.on = true;
            hd.hc = m.sh_conv.cout; hd.hs = m.sh_fc0.out;
            hd.small_w = wts->h32_small; hd.sh_b0 = wts->sh_b0; hd.sh_w1t = wts->sh_w1t; hd.sh_b1 = wts->sh_b1;
            hd.sh_w2 = wts->sh_w2; hd.sh_b2 = wts->sh_b2;
            hd.pc = m.policy_conv_channels; hd.p_b1 = wts->p_b1;
            hd.policy_len = m.policy_len; hd.zero_tail = m.policy_kind == kz::POLICY_ATAXX_CONV ? 1 : 0;
            if (m.policy_kind == kz::POLICY_CONV && m.policy_extra_moves) {
                hd.extra = m.policy_extra_moves;
                hd.pe_bc = wts->pe_bc; hd.pe_wl = wts->pe_wl; hd.pe_bl = wts->pe_bl;
            }
            hd.scalars = d_scalars; hd.policy = d_policy;
            hd.nonfinite_flag = nf_flag; hd.epoch = nf_epoch;
        }
        if (dec && t.heads.on) t.heads.decode = *dec;
        t.dense3 = t32_dense3;
        t.wide = wide;
        prof.begin(split16 ? "kz_tower_resident_split" : pairs16 ? "kz_tower_resident_f16g" : "kz_tower_resident_f32", stream);
        if (split16) kz::launch_tower_split(t, stream);
        else if (pairs16) kz::launch_tower_pairs(t, false, stream);  // f16 tensors behind the same pointers
        else kz::launch_tower32(t, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        tower_out = 0;
        return 0;
    }
    if (bsplit) {
        // the stem in exact f32 (its inputs are f32 planes), its output split into (hi, lo) halves — an f32 tensor and a
        // (hi, lo) tensor of the same shape have the same size, so the three activation buffers serve both —, the
        // 2·depth tower convolutions in split arithmetic, the last one writing f32 for the heads
        if (wts->stem_split) {  // encoded f32 planes [M][32] -> (hi, lo) rows -> the board-tile kernel, one chunk
            prof.begin("kz_split_rows", stream);
            kz::launch_split_rows((const float *)x_in, act[2], (size_t)M, cin_p, stream);
            prof.end(stream);
            if (conv(wts->tower[0], act[2], cin_p, act[0], cp, M, 0, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
        } else {
            if (conv(wts->tower[0], x_in, cin_p, act[2], cp, M, 0, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
            prof.begin("kz_split_rows", stream);
            kz::launch_split_rows((const float *)act[2], act[0], (size_t)M, cp, stream);
            prof.end(stream);
        }
        int cur = 0;
        for (int i = 1; i <= m.depth; i++) {
            const int mid = (cur + 1) % 3, nxt = (cur + 2) % 3;
            const bool last = i == m.depth;
            if (conv(wts->tower[2 * i - 1], act[cur], cp, act[mid], cp, M, 1, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
            if (conv(wts->tower[2 * i], act[mid], cp, last ? nullptr : act[nxt], cp, M, 1, act[cur], last, m.h, m.w, hw, hw, 0,
                     last ? (float *)act[nxt] : nullptr, cp))
                return 1;
            cur = nxt;
        }
        tower_out = cur;
        return 0;
    }
    // stem: conv + bias, no activation (post_act.py:205)
    if (conv(wts->tower[0], x_in, cin_p, act[0], cp, M, 0, nullptr, m.depth == 0, m.h, m.w, hw, hw, 0)) return 1;
    if (stash(m.depth == 0 ? "tower.1" : "tower.0", act[0], batch)) return 1;
    int cur = 0;
    for (int i = 1; i <= m.depth; i++) {
        const int mid = (cur + 1) % 3, nxt = (cur + 2) % 3;
        const bool last = i == m.depth;
        if (conv(wts->tower[2 * i - 1], act[cur], cp, act[mid], cp, M, 1, nullptr, false, m.h, m.w, hw, hw, 0))
            return 1;
        if (stash("tower." + std::to_string(i) + ".mid", act[mid], batch)) return 1;
        // x + relu(bn(conv(mid))) (post_act.py:227-228); the tower's final BN rides on the last block
        if (conv(wts->tower[2 * i], act[mid], cp, act[nxt], cp, M, 1, act[cur], last, m.h, m.w, hw, hw, 0))
            return 1;
        if (stash("tower." + std::to_string(last ? i + 1 : i), act[nxt], batch)) return 1;
        cur = nxt;
    }
    tower_out = cur;
    return 0;
}

inline bool kz_engine::extra_in_scalar_head() const {
    const Model &m = *model;
    return m.policy_kind == kz::POLICY_CONV && m.policy_extra_moves > 0 && wts->sh_w0x &&
           kz::scalar_head_takes_extra(dtype == KZ_DTYPE_F32 || split16 ? 0 : 1, cp, m.sh_conv.cout);
}

inline int kz_engine::run_heads(int batch, float *d_scalars, float *d_policy) {
    if (fused_heads || fused32 || fused_split || fused_pairs || dense_net) return 0;  // written by the tower launch
    const Model &m = *model;
    const int hw = m.h * m.w, M = batch * hw;
    const void *x = act[tower_out];
    if (wts->att_heads) {  // ScalarHead + AttentionPolicyHead in one launch
        kz::AttHeadsArgs a{};
        a.x = x; a.ldx = cp; a.batch = batch; a.channels = m.channels; a.q = m.policy_query_channels;
        a.hc = m.sh_conv.cout; a.hs = m.sh_fc0.out; a.policy_len = m.policy_len;
        a.weights = wts->ah_w; a.bias = wts->ah_bias;
        a.w1 = wts->sh_w1; a.b1 = wts->sh_b1; a.w2 = wts->sh_w2; a.b2 = wts->sh_b2;
        a.flat_to_att = wts->flat_to_att; a.scalars = d_scalars; a.policy = d_policy;
        a.nonfinite_flag = nf_flag; a.epoch = nf_epoch;
        prof.begin("kz_att_heads_f16", stream);
        kz::launch_att_heads(a, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    {
        kz::ScalarHeadArgs a{x, cp, batch, hw, m.channels, m.sh_conv.cout, m.sh_fc0.out,
                             wts->sh_w0, wts->sh_b0, wts->sh_w1, wts->sh_b1, wts->sh_w2, wts->sh_b2, d_scalars,
                             nf_flag, nf_epoch, wts->sh_w1t};
        // ConvPolicyHead's extra moves read the same tower output: one pass for both (post_act.py:63-67)
        if (extra_in_scalar_head()) {
            a.extra = m.policy_extra_moves;
            a.w0x = wts->sh_w0x; a.pe_bc = wts->pe_bc; a.pe_wl = wts->pe_wl; a.pe_bl = wts->pe_bl;
            a.policy = d_policy; a.policy_len = m.policy_len; a.policy_offset = m.policy_conv_channels * hw;
        }
        prof.begin("kz_scalar_head", stream);
        kz::launch_scalar_head(dtype, a, stream);
        prof.end(stream);
    }
    switch (m.policy_kind) {
        case kz::POLICY_ATAXX_CONV:
        case kz::POLICY_CONV: {
            const int pc = m.policy_conv_channels;
            const DevConv &c0 = wts->p_conv0;
            if (m.policy_kind == kz::POLICY_CONV && c0.sw && cp >= c0.cin_p &&
                kz::conv1x1_policy_epilogue_supported(c0.cin_p, c0.cout_p, c0.cout, pc)) {
                // Conv1x1 C->C + ReLU + Conv1x1 C->1 in one launch: the hidden layer never goes to memory
                kz::Conv1x1SplitArgs c{};
                c.split = split16;
                c.x = x; c.ldx = cp; c.weights = c0.sw; c.bias = c0.b; c.y = nullptr; c.ldy = 0;
                c.M = M; c.cin_p = c0.cin_p; c.cout_p = c0.cout_p; c.relu = 1;
                c.group = hw; c.src_group = hw; c.src_off = 0;
                c.pw1 = wts->p_w1; c.pb1 = wts->p_b1; c.policy = d_policy; c.policy_len = m.policy_len; c.hw = hw;
                prof.begin("kz_conv1x1_split", stream);
                kz::launch_conv1x1_split(c, stream);
                prof.end(stream);
            } else {
                if (conv(c0, x, cp, head0, c0.cout_p, M, 1, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
                kz::PolicyConvArgs a{head0, c0.cout_p, batch, hw, m.channels, pc, wts->p_w1, wts->p_b1,
                                     d_policy, m.policy_len, m.policy_kind == kz::POLICY_ATAXX_CONV ? 1 : 0};
                prof.begin("kz_policy_conv", stream);
                kz::launch_policy_conv(dtype, a, stream);
                prof.end(stream);
            }
            if (m.policy_kind == kz::POLICY_CONV && m.policy_extra_moves && !extra_in_scalar_head()) {
                kz::PolicyExtraArgs e{x, cp, batch, hw, m.channels, m.policy_extra_moves, wts->pe_wc, wts->pe_bc,
                                      wts->pe_wl, wts->pe_bl, d_policy, m.policy_len, pc * hw};
                prof.begin("kz_policy_extra", stream);
                kz::launch_policy_extra(dtype, e, stream);
                prof.end(stream);
            }
            break;
        }
        case kz::POLICY_ARIMAA: {
            // ArimaaPolicyHead (post_act.py:144-173): policy = concat(scalar(common) [1 + 6], flatten(bulk(common)) [4 * hw])
            const DevConv &c0 = wts->p_conv0;
            if (conv(c0, x, cp, head0, c0.cout_p, M, 1, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
            kz::PolicyConvArgs a{head0, c0.cout_p, batch, hw, m.channels, 4, wts->p_w1, wts->p_b1,
                                 d_policy + 7, m.policy_len, 0};  // (the four planes start behind the seven scalars)
            prof.begin("kz_policy_conv", stream);
            kz::launch_policy_conv(dtype, a, stream);
            prof.end(stream);
            // the scalar branch has the ScalarHead's shape: the same kernel, seven outputs into the policy rows
            kz::ScalarHeadArgs sa{x, cp, batch, hw, m.channels, m.arimaa_hidden_channels, m.arimaa_hidden_size,
                                  wts->pa_w0, wts->pa_b0, wts->pa_w1, wts->pa_b1, wts->pa_w2, wts->pa_b2, d_policy,
                                  nullptr, 0, wts->pa_w1t};
            sa.n_out = 7;
            sa.out_ld = m.policy_len;
            prof.begin("kz_scalar_head", stream);
            kz::launch_scalar_head(dtype, sa, stream);
            prof.end(stream);
            break;
        }
        case kz::POLICY_ATTENTION: {
            // bulk = conv_bulk(common) on all 64 squares; under = conv_under(common[:, :, 7, None, :]) on the
            // 8 squares of rank index 7 (post_act.py:128-129): source rows 56..63 of each board
            if (conv(wts->p_bulk, x, cp, head0, wts->p_bulk.cout_p, M, 0, nullptr, false, m.h, m.w, hw, hw, 0))
                return 1;
            if (conv(wts->p_under, x, cp, head1, wts->p_under.cout_p, batch * 8, 0, nullptr, false, 1, 8, 8, hw, 56))
                return 1;
            kz::AttentionArgs a{head0, head1, wts->p_bulk.cout_p, wts->p_under.cout_p, batch,
                                m.policy_query_channels, wts->flat_to_att, d_policy, m.policy_len};
            prof.begin("kz_attention_gather", stream);
            kz::launch_attention(dtype, a, stream);
            prof.end(stream);
            break;
        }
        case kz::POLICY_NONE: break;
        case kz::POLICY_DENSE: {
            const void *flat = x;
            int flat_ld = hw * cp;
            if (m.dense_hidden_channels) {
                if (conv(wts->p_conv0, x, cp, head0, wts->p_conv0.cout_p, M, 1, nullptr, false, m.h, m.w, hw, hw, 0))
                    return 1;
                flat = head0;
                flat_ld = hw * wts->p_conv0.cout_p;
            }
            // Flatten + Linear: one GEMM row per board
            if (m.dense_hidden_size) {
                if (conv(wts->p_fc0, flat, flat_ld, head1, wts->p_fc0.cout_p, batch, 1, nullptr, false, 1, 1, 1, 1, 0))
                    return 1;
                if (conv(wts->p_fc1, head1, wts->p_fc0.cout_p, nullptr, 0, batch, 0, nullptr, false, 1, 1, 1, 1, 0,
                         d_policy, m.policy_len))
                    return 1;
            } else {
                if (conv(wts->p_fc1, flat, flat_ld, nullptr, 0, batch, 0, nullptr, false, 1, 1, 1, 1, 0, d_policy,
                         m.policy_len))
                    return 1;
            }
            break;
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// dec (decode_in_launch() engines only): the launch ends in decode_output and writes dec->values / dec->probs
inline int kz_engine::forward_packed(const void *d_bits, size_t stride, const void *d_sin, int batch, void *d_sout, void *d_pol,
                   const kz::DecodeArgs *dec) {
    const Model &m = *model;
    if (resident || resident32 || pairs16 || att_f16) {  // encode is fused into the tower launch
        const PackedIn in{d_bits, stride, d_sin};
        if (run_tower(batch, (float *)d_sout, (float *)d_pol, &in, dec)) return 1;
        return run_heads(batch, (float *)d_sout, (float *)d_pol);
    }
    prof.begin("kz_encode_packed", stream);
    kz::launch_encode_packed(dtype, (const uint8_t *)d_bits, stride, (const float *)d_sin, batch, m.n_scalar,
                             m.n_bool, m.h * m.w, x_in, cin_p, stream);
    prof.end(stream);
    HIP_TRY(hipGetLastError());
    if (run_tower(batch, (float *)d_sout, (float *)d_pol)) return 1;
    return run_heads(batch, (float *)d_sout, (float *)d_pol);
}

inline int kz_engine::forward_dense(const void *d_nchw, int batch, void *d_sout, void *d_pol) {
    const Model &m = *model;
    prof.begin("kz_encode_dense", stream);
    kz::launch_encode_dense(dtype, (const float *)d_nchw, batch, m.c_in, m.h * m.w, x_in, cin_p, stream);
    prof.end(stream);
    HIP_TRY(hipGetLastError());
    if (run_tower(batch, (float *)d_sout, (float *)d_pol)) return 1;
    return run_heads(batch, (float *)d_sout, (float *)d_pol);
}
