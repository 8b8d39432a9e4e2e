// kz_device_weights.hpp — device weights, shared by all engines of one (model, device, dtype): the packed weight streams of
// whichever kernels kz_plan.hpp chose for the network, uploaded once and reference-counted.  Part of kz_engine.hip's
// translation unit (included inside its anonymous namespace, after kz_engine_util.hpp).
#pragma once

// ------------------------------------------------------------------------------------------------
// Device weights, shared by all engines of one (model, device, dtype)
// ------------------------------------------------------------------------------------------------
struct DevConv {  // packed for kz_conv_igemm: [k*k][cout_p][cin_p] in T, bias f32 [cout_p]
    void *w = nullptr;
    float *b = nullptr;
    int cin_p = 0, cout_p = 0, cout = 0, k = 1;
    void *bw = nullptr;  // instead of w: packed for kz_board_conv_f16 (3x3, f16, channels % 64 == 0)
    bool bw2 = false;    // ... packed for kz_board_conv2_f16 (two Go-size boards per workgroup)
    void *bws = nullptr; // instead of w: (hi, lo) pairs packed for kz_board_conv_split16 (3x3, split arithmetic)
    void *sw = nullptr;  // in addition to w: (hi, lo) f16 pairs for kz_conv1x1_split (1x1 head convolutions, split16)
};

struct DeviceWeights {
    int device = 0, dtype = 0;
    std::vector<void *> allocs;

    std::vector<DevConv> tower;  // generic path
    float *post_scale = nullptr, *post_shift = nullptr;

    // resident tower
    bool resident = false, fused_heads = false, resident32 = false, split16 = false, pairs16 = false;
    bool fused_split = false;  // the split launch carries the heads (set before build)
    bool fused_pairs = false;  // the plain-f16 generic launch carries the conv policy heads (set before build)
    float *h32_small = nullptr;  // the fused f32 heads' small 1x1 convolutions (tower32_pack_small_weights)
    void *res32_w = nullptr;  // f32 resident launch (exact f32, or split f16 pairs): one packed weight stream
    void *res_w_stem = nullptr, *res_w_tower = nullptr;
    float *res_bias = nullptr;
    int32_t *att_idx = nullptr;

    // scalar head
    float *sh_w0 = nullptr, *sh_b0 = nullptr, *sh_w1 = nullptr, *sh_b1 = nullptr, *sh_w2 = nullptr, *sh_b2 = nullptr;
    float *sh_w1t = nullptr;  // sh_w1 transposed: [inputs][outputs]
    float *sh_w0x = nullptr;  // [hc + 1][C]: the scalar head's 1x1 filters followed by ConvPolicyHead's extra-move filter
    // ArimaaPolicyHead's scalar branch (post_act.py:155-162), laid out like the scalar head's
    float *pa_w0 = nullptr, *pa_b0 = nullptr, *pa_w1 = nullptr, *pa_b1 = nullptr, *pa_w2 = nullptr, *pa_b2 = nullptr, *pa_w1t = nullptr;
    // policy
    DevConv p_conv0;                                   // conv / ataxx_conv / dense hidden conv
    float *p_w1 = nullptr, *p_b1 = nullptr;            // last 1x1 conv of the conv heads
    float *pe_wc = nullptr, *pe_bc = nullptr, *pe_wl = nullptr, *pe_bl = nullptr;  // seq_extra
    DevConv p_bulk, p_under;
    int32_t *flat_to_att = nullptr;
    DevConv p_fc0, p_fc1;

    ~DeviceWeights() {
        (void)hipSetDevice(device);
        for (void *p : allocs) (void)hipFree(p);
    }

    int upload(const void *src, size_t bytes, void **dst) {
        HIP_TRY(hipMalloc(dst, bytes ? bytes : 16));
        allocs.push_back(*dst);
        if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    int upload_f32(const std::vector<float> &v, float **dst) { return upload(v.data(), v.size() * 4, (void **)dst); }

    // values [rows][cols] f32 -> T [rows_p][cols_p], zero padded
    int upload_matrix(const std::vector<float> &v, int rows, int cols, int rows_p, int cols_p, void **dst) {
        if (dtype == KZ_DTYPE_F32) {
            std::vector<float> p((size_t)rows_p * cols_p, 0.0f);
            for (int r = 0; r < rows; r++)
                for (int c = 0; c < cols; c++) p[(size_t)r * cols_p + c] = v[(size_t)r * cols + c];
            return upload(p.data(), p.size() * 4, dst);
        }
        std::vector<uint16_t> p((size_t)rows_p * cols_p, 0);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) p[(size_t)r * cols_p + c] = f32_to_f16_bits(v[(size_t)r * cols + c]);
        return upload(p.data(), p.size() * 2, dst);
    }

    // 3x3 tower conv for the board-tile kernel (kz_board_conv.hip)
    // (the stem's few input planes are padded with zero weights to the kernel's 64-channel chunk: cin_pad)
    int upload_board_conv(const Conv &cv, DevConv &d, int cin_pad = 0) {
        d.k = 3;
        d.cout = d.cout_p = cv.cout;
        const int cin = cin_pad ? cin_pad : cv.cin;
        d.cin_p = cin;
        std::vector<float> padded;
        const float *w = cv.w.data();
        if (cin != cv.cin) {
            padded.assign((size_t)cv.cout * cin * 9, 0.0f);
            for (int o = 0; o < cv.cout; o++)
                for (int i = 0; i < cv.cin; i++)
                    for (int t = 0; t < 9; t++) padded[((size_t)o * cin + i) * 9 + t] = cv.w[((size_t)o * cv.cin + i) * 9 + t];
            w = padded.data();
        }
        std::vector<uint16_t> packed(kz::board_conv_weight_elems(cin, cv.cout));
#ifdef KZ_EXPERIMENTS
        d.bw2 = conv2;
        if (conv2) kz::board_conv2_pack_weights(w, cv.cout, cin, packed.data());
        else
#endif
        kz::board_conv_pack_weights(w, cv.cout, cin, packed.data());
        if (upload(packed.data(), packed.size() * 2, &d.bw)) return 1;
        return upload_f32(cv.b, &d.b);
    }

    // the same convolution for the board-tile kernel in split arithmetic
    // (cin_pad: the stem's few input planes padded with zero weights to the kernel's 32-channel chunk)
    int upload_board_conv_split(const Conv &cv, DevConv &d, int cin_pad = 0) {
        d.k = 3;
        d.cout = d.cout_p = cv.cout;
        const int cin = cin_pad ? cin_pad : cv.cin;
        d.cin_p = cin;
        std::vector<float> padded;
        const float *w = cv.w.data();
        if (cin != cv.cin) {
            padded.assign((size_t)cv.cout * cin * 9, 0.0f);
            for (int o = 0; o < cv.cout; o++)
                for (int i = 0; i < cv.cin; i++)
                    for (int t = 0; t < 9; t++) padded[((size_t)o * cin + i) * 9 + t] = cv.w[((size_t)o * cv.cin + i) * 9 + t];
            w = padded.data();
        }
        std::vector<uint16_t> packed(kz::board_conv_split_weight_elems(cin, cv.cout));
        kz::board_conv_split_pack_weights(w, cv.cout, cin, packed.data());
        if (upload(packed.data(), packed.size() * 2, &d.bws)) return 1;
        return upload_f32(cv.b, &d.b);
    }

    // OIHW conv -> [tap][cout_p][cin_p]; tap = ky*k + kx
    int upload_conv(const Conv &cv, DevConv &d) {
        d.k = cv.k;
        d.cout = cv.cout;
        d.cin_p = round_up(cv.cin, 32);
        d.cout_p = round_up(cv.cout, 32);
        const int taps = cv.k * cv.k;
        std::vector<float> flat((size_t)taps * d.cout_p * d.cin_p, 0.0f);
        for (int o = 0; o < cv.cout; o++)
            for (int i = 0; i < cv.cin; i++)
                for (int t = 0; t < taps; t++)
                    flat[((size_t)t * d.cout_p + o) * d.cin_p + i] = cv.w[((size_t)o * cv.cin + i) * taps + t];
        if (upload_matrix(flat, taps * d.cout_p, d.cin_p, taps * d.cout_p, d.cin_p, &d.w)) return 1;
        // 1x1 head convolutions: the tiled GEMM of kz_tower_split.hip in split arithmetic (split16) or plain f16
        if ((split16 || dtype == KZ_DTYPE_F16) && cv.k == 1 && kz::conv1x1_split_supported(d.cin_p, d.cout_p)) {
            std::vector<uint16_t> packed(kz::conv1x1_split_weight_elems(d.cin_p, d.cout_p, split16));
            kz::conv1x1_split_pack_weights(cv.w.data(), cv.cout, cv.cin, d.cout_p, d.cin_p, split16, packed.data());
            if (upload(packed.data(), packed.size() * 2, &d.sw)) return 1;
        }
        std::vector<float> b(d.cout_p, 0.0f);
        for (int o = 0; o < cv.cout; o++) b[o] = cv.b[o];
        return upload_f32(b, &d.b);
    }

    // nn.Linear over a channel-major flatten of [ch][hw] (index c*hw + p) applied to NHWC rows [hw][ch_p]:
    // re-index the input dimension to p*ch_p + c and run it as a 1x1 "convolution" over one row per board.
    int upload_flat_linear(const Linear &l, int ch, int hw, int ch_p, DevConv &d) {
        d.k = 1;
        d.cout = l.out;
        d.cout_p = round_up(l.out, 32);
        d.cin_p = hw * ch_p;
        std::vector<float> flat((size_t)d.cout_p * d.cin_p, 0.0f);
        for (int o = 0; o < l.out; o++)
            for (int c = 0; c < ch; c++)
                for (int p = 0; p < hw; p++)
                    flat[(size_t)o * d.cin_p + (size_t)p * ch_p + c] = l.w[(size_t)o * l.in + (size_t)c * hw + p];
        if (upload_matrix(flat, d.cout_p, d.cin_p, d.cout_p, d.cin_p, &d.w)) return 1;
        std::vector<float> b(d.cout_p, 0.0f);
        for (int o = 0; o < l.out; o++) b[o] = l.b[o];
        return upload_f32(b, &d.b);
    }

    int upload_linear(const Linear &l, DevConv &d) {
        d.k = 1;
        d.cout = l.out;
        d.cout_p = round_up(l.out, 32);
        d.cin_p = round_up(l.in, 32);
        std::vector<float> flat((size_t)d.cout_p * d.cin_p, 0.0f);
        for (int o = 0; o < l.out; o++)
            for (int i = 0; i < l.in; i++) flat[(size_t)o * d.cin_p + i] = l.w[(size_t)o * l.in + i];
        if (upload_matrix(flat, d.cout_p, d.cin_p, d.cout_p, d.cin_p, &d.w)) return 1;
        std::vector<float> b(d.cout_p, 0.0f);
        for (int o = 0; o < l.out; o++) b[o] = l.b[o];
        return upload_f32(b, &d.b);
    }

    bool use_board_conv = false;
    bool use_board_split = false;  // split16 on a board too large for the resident launch: per-layer board-tile kernel
    int stem_cin_p = 0;  // != 0: the stem goes through the board-tile kernel and wants encoded rows of this many channels
    bool stem_split = false;  // split16 per layer: the stem too goes through kz_board_conv_split16 (<= 32 input planes)
    bool conv2 = false;  // the board-tile layers go through kz_board_conv2_f16
    // AttentionTower (kz_att_tower.hip): the model's own matrices in f32
    float *att_expand = nullptr, *att_embedding = nullptr, *att_layers = nullptr;
    // DenseNetwork (kz_dense_network.hip)
    float *dn_w_in = nullptr, *dn_b_in = nullptr, *dn_blocks = nullptr, *dn_sf = nullptr, *dn_tf = nullptr, *dn_w_out = nullptr,
          *dn_b_out = nullptr;
    bool att_heads = false;  // (set before build) ScalarHead + AttentionPolicyHead in one f16 launch (kz_att_heads.hip)
    void *ah_w = nullptr;
    float *ah_bias = nullptr;
    bool att_f16 = false;  // (set before build) the f16 launch's fragment streams instead of att_expand / att_layers
    void *att16_expand = nullptr, *att16_layers = nullptr;
    int *bc_rowmap = nullptr;  // (experiment build: kz_board_conv2_f16's tile-row map and halo-row list)
    unsigned short *bc_halo = nullptr;
    int bc_n_halo = 0;
    int build(const Model &m, bool want_resident, bool want_fused, bool want_resident32, bool want_split16,
              bool want_pairs16) {
        resident32 = want_resident32;
        split16 = want_split16;
        pairs16 = want_pairs16;  // kz_tower_resident_split without the lo halves: plain f16, generic shapes
        const int C = m.channels, cp = round_up(C, 32), hw = m.h * m.w;
        HIP_TRY(hipSetDevice(device));
        resident = want_resident;
        fused_heads = want_resident && want_fused;

        if (m.tower_kind == kz::TOWER_DENSE_NET) {
            // dn_in's columns from the channel-major flatten (c * hw + p) to the encoded rows' order (p * cin_p + c)
            const int cin_p = round_up(m.c_in, 32), n_in = hw * cin_p;
            std::vector<float> w_in((size_t)C * n_in, 0.0f);
            for (int o = 0; o < C; o++)
                for (int c = 0; c < m.c_in; c++)
                    for (int p = 0; p < hw; p++) w_in[(size_t)o * n_in + (size_t)p * cin_p + c] = m.dn_in.w[(size_t)o * m.dn_in.in + (size_t)c * hw + p];
            std::vector<float> blocks;
            blocks.reserve(kz::dense_network_block_elems(C) * m.dn_blocks.size());
            for (auto &b : m.dn_blocks)
                for (const std::vector<float> *v : {&b.sa, &b.ta, &b.la.w, &b.la.b, &b.sb, &b.tb, &b.lb.w, &b.lb.b}) blocks.insert(blocks.end(), v->begin(), v->end());
            if (upload_f32(w_in, &dn_w_in) || upload_f32(m.dn_in.b, &dn_b_in) || upload_f32(blocks, &dn_blocks) || upload_f32(m.dn_sf, &dn_sf) ||
                upload_f32(m.dn_tf, &dn_tf) || upload_f32(m.dn_out.w, &dn_w_out) || upload_f32(m.dn_out.b, &dn_b_out))
                return 1;
            return 0;
        }
        std::vector<float> ps(cp, 1.0f), pt(cp, 0.0f);
        for (int i = 0; i < C; i++) {
            ps[i] = m.final_scale[i];
            pt[i] = m.final_shift[i];
        }
        if (upload_f32(ps, &post_scale) || upload_f32(pt, &post_shift)) return 1;

        if (m.tower_kind == kz::TOWER_ATTENTION && att_f16) {
            const int cin_p = round_up(m.c_in, 32);
            const bool f32 = dtype == KZ_DTYPE_F32;
            const size_t esz = f32 ? 4 : 2;
            std::vector<char> ex(kz::att_tower16_expand_elems(C, cin_p) * esz);
            kz::att_tower16_pack_expand(m.att_expand.data(), C, m.c_in, cin_p, f32, ex.data());
            const size_t per = kz::att_tower16_layer_elems(C, m.att_dff) * esz;
            std::vector<char> all(per * m.att_layers.size());
            for (size_t l = 0; l < m.att_layers.size(); l++)
                kz::att_tower16_pack_layer(m.att_layers[l].qkv.data(), m.att_layers[l].out.data(), m.att_layers[l].ff0.data(),
                                           m.att_layers[l].ff1.data(), C, m.att_dff, m.att_alpha, f32, all.data() + per * l);
            if (upload(ex.data(), ex.size(), &att16_expand) || upload(all.data(), all.size(), &att16_layers) ||
                upload_f32(m.att_embedding, &att_embedding))
                return 1;
        } else if (m.tower_kind == kz::TOWER_ATTENTION) {
            std::vector<float> all;
            all.reserve(kz::att_tower_layer_elems(C, m.att_heads, m.att_dk, m.att_dv, m.att_dff) * m.att_layers.size());
            for (auto &l : m.att_layers) {
                all.insert(all.end(), l.qkv.begin(), l.qkv.end());
                all.insert(all.end(), l.out.begin(), l.out.end());
                all.insert(all.end(), l.ff0.begin(), l.ff0.end());
                all.insert(all.end(), l.ff1.begin(), l.ff1.end());
            }
            if (upload_f32(m.att_expand, &att_expand) || upload_f32(m.att_embedding, &att_embedding) || upload_f32(all, &att_layers)) return 1;
        } else if ((split16 && !use_board_split) || pairs16) {
            // f16 fragments — (hi, lo) pairs for split16 — in fragment order: 9 * ceil(c_in / 32) stem k-steps, then 9*C/32 per convolution
            // (+ the attention heads' five passes and bias rows when the split launch carries the heads)
            const bool conv_heads = (split16 && fused_split && m.policy_kind != kz::POLICY_ATTENTION) || (pairs16 && fused_pairs);  // (Ataxx, Go 9x9)
            const bool heads = split16 && fused_split && !conv_heads;
            const size_t tower_elems = kz::tower_split_weight_elems(C, m.depth, m.c_in, split16);
            std::vector<uint16_t> packed(tower_elems + (heads ? kz::tower_split_heads_weight_elems() : 0) +
                                         (conv_heads ? kz::tower_split_conv_heads_weight_elems(C, split16) : 0));
            const size_t step_elems = (size_t)(split16 ? 2 : 1) * C * 32, stem_elems = kz::tower_split_stem_elems(C, m.c_in, split16),
                         layer_elems = (size_t)9 * (C / 32) * step_elems;
            kz::tower_split_pack_weights(m.tower[0].w.data(), C, m.c_in, hw, true, split16, packed.data());
            for (int l = 0; l < 2 * m.depth; l++)
                kz::tower_split_pack_weights(m.tower[1 + l].w.data(), C, C, hw, false, split16,
                                             packed.data() + stem_elems + layer_elems * l);
            std::vector<float> bias((size_t)(1 + 2 * m.depth + (heads ? 5 : conv_heads ? 1 : 0)) * C);
            for (int l = 0; l < 1 + 2 * m.depth; l++)
                for (int o = 0; o < C; o++) bias[(size_t)l * C + o] = m.tower[l].b[o];
            if (conv_heads) {  // the policy head's hidden layer as one more pass; the small convolutions as for the f32 launch
                kz::tower_split_pack_conv_heads(m.p_conv0.w.data(), C, split16, packed.data() + tower_elems);
                for (int o = 0; o < C; o++) bias[(size_t)(1 + 2 * m.depth) * C + o] = m.p_conv0.b[o];
                if (pairs16) {  // the plain-f16 launch runs the two small convolutions as f16 MFMAs
                    std::vector<uint16_t> small(kz::tower_split_small_weight16_elems(C));
                    kz::tower_split_pack_small_weights16(m.sh_conv.w.data(), m.sh_conv.cout,
                                                         m.policy_extra_moves ? m.p_extra_conv.w.data() : nullptr,
                                                         m.p_conv1.w.data(), m.policy_conv_channels, C, small.data());
                    if (upload(small.data(), small.size() * 2, (void **)&h32_small)) return 1;
                } else {
                    std::vector<float> small(kz::tower32_small_weight_elems(C));
                    kz::tower32_pack_small_weights(m.sh_conv.w.data(), m.sh_conv.cout,
                                                   m.policy_extra_moves ? m.p_extra_conv.w.data() : nullptr, m.p_conv1.w.data(),
                                                   m.policy_conv_channels, C, small.data());
                    if (upload_f32(small, &h32_small)) return 1;
                }
            }
            if (heads) {
                kz::tower_split_pack_heads(m.p_bulk.w.data(), m.p_bulk.b.data(), m.p_under.w.data(), m.p_under.b.data(),
                                           packed.data() + tower_elems, bias.data() + (size_t)(1 + 2 * m.depth) * C);
                std::vector<int32_t> idx(m.flat_to_att.size());
                for (size_t i = 0; i < idx.size(); i++) idx[i] = (m.flat_to_att[i] / 88) * 96 + m.flat_to_att[i] % 88;
                if (upload(idx.data(), idx.size() * 4, (void **)&att_idx)) return 1;
            }
            if (upload(packed.data(), packed.size() * 2, &res32_w)) return 1;
            if (upload_f32(bias, &res_bias)) return 1;
        } else if (resident32) {
            // (the conv policy head's first 1x1 conv rides at the end of the stream whenever the launch can fuse the
            // heads; a launch without heads never reads it)
            const bool heads32 = kz::tower32_heads_supported((int)m.policy_kind, m.policy_extra_moves,
                                                             m.policy_conv_channels, m.h, m.w, C, m.sh_conv.cout, m.sh_fc0.out);
            const size_t tower_elems = kz::tower32_weight_elems(m.c_in, C, m.depth);
            std::vector<float> packed(tower_elems + (heads32 ? kz::tower32_heads_weight_elems(C) : 0) +
                                      kz::tower32_weight_pad_elems(C), 0.0f);
            const size_t stem_elems = (size_t)9 * ((m.c_in + 15) / 16) * 16 * C, layer_elems = (size_t)9 * C * C;
            kz::tower32_pack_weights(m.tower[0].w.data(), C, m.c_in, true, packed.data());
            for (int l = 0; l < 2 * m.depth; l++)
                kz::tower32_pack_weights(m.tower[1 + l].w.data(), C, C, false, packed.data() + stem_elems + layer_elems * l);
            std::vector<float> bias((size_t)(1 + 2 * m.depth + (heads32 ? 1 : 0)) * C);
            for (int l = 0; l < 1 + 2 * m.depth; l++)
                for (int o = 0; o < C; o++) bias[(size_t)l * C + o] = m.tower[l].b[o];
            if (heads32) {
                kz::tower32_pack_head_weights(m.p_conv0.w.data(), C, packed.data() + tower_elems);
                for (int o = 0; o < C; o++) bias[(size_t)(1 + 2 * m.depth) * C + o] = m.p_conv0.b[o];
                std::vector<float> small(kz::tower32_small_weight_elems(C));
                kz::tower32_pack_small_weights(m.sh_conv.w.data(), m.sh_conv.cout,
                                               m.policy_extra_moves ? m.p_extra_conv.w.data() : nullptr, m.p_conv1.w.data(),
                                               m.policy_conv_channels, C, small.data());
                if (upload_f32(small, &h32_small)) return 1;
            }
            if (upload(packed.data(), packed.size() * 4, &res32_w)) return 1;
            if (upload_f32(bias, &res_bias)) return 1;
        } else if (resident) {
            const int cin_p = round_up(m.c_in, 32);
            const size_t stem_elems = (size_t)9 * 256 * cin_p, layer_elems = (size_t)9 * 256 * 256;
            const size_t head_elems = fused_heads ? kz::tower_heads_weight_elems() : 0;
            std::vector<uint16_t> stem(stem_elems), rest(layer_elems * 2 * m.depth + head_elems);
            kz::tower_pack_weights(m.tower[0].w.data(), C, m.c_in, cin_p, stem.data());
            for (int l = 0; l < 2 * m.depth; l++)
                kz::tower_pack_weights(m.tower[1 + l].w.data(), C, C, 256, rest.data() + layer_elems * l);
            std::vector<float> bias((size_t)(1 + 2 * m.depth + (fused_heads ? 5 : 0)) * 256);
            for (int l = 0; l < 1 + 2 * m.depth; l++)
                for (int o = 0; o < 256; o++) bias[(size_t)l * 256 + o] = m.tower[l].b[o];
            if (fused_heads) {
                kz::tower_pack_heads(m.p_bulk.w.data(), m.p_bulk.b.data(), m.p_under.w.data(), m.p_under.b.data(),
                                     rest.data() + layer_elems * 2 * m.depth, bias.data() + (size_t)(1 + 2 * m.depth) * 256);
                std::vector<int32_t> idx(m.flat_to_att.size());
                for (size_t i = 0; i < idx.size(); i++) idx[i] = (m.flat_to_att[i] / 88) * 96 + m.flat_to_att[i] % 88;
                if (upload(idx.data(), idx.size() * 4, (void **)&att_idx)) return 1;
            }
            if (upload(stem.data(), stem.size() * 2, &res_w_stem)) return 1;
            if (upload(rest.data(), rest.size() * 2, &res_w_tower)) return 1;
            if (upload_f32(bias, &res_bias)) return 1;
        } else {
            tower.resize(m.tower.size());
            const char *noboard = getenv("KZ_NO_BOARD_CONV");
#ifdef KZ_EXPERIMENTS
            if (use_board_conv && !(noboard && noboard[0] == '1')) {
                const char *c2 = getenv("KZ_BOARD_CONV2");
                // (experiment, opt-in: the second organisation measured 26.2k against 33.5k evals/s on Go-19 40x256)
                conv2 = c2 && c2[0] == '1' && kz::board_conv2_supported(dtype, m.h, m.w, m.channels, m.channels);
                if (conv2) {  // its tile-row map and halo-row list (the product kernel needs no tables)
                    std::vector<int> rowmap;
                    std::vector<unsigned short> halo;
                    kz::board_conv2_tables(m.h, m.w, rowmap, halo);
                    bc_n_halo = (int)halo.size();
                    if (upload(rowmap.data(), rowmap.size() * sizeof(int), (void **)&bc_rowmap)) return 1;
                    if (upload(halo.data(), halo.size() * sizeof(unsigned short), (void **)&bc_halo)) return 1;
                }
            }
#endif
            for (size_t i = 0; i < m.tower.size(); i++) {
                const bool on = use_board_conv && !(noboard && noboard[0] == '1');
                // the stem joins the board-tile family with its input planes padded to one 64-channel chunk (the encode
                // kernel then writes 64-channel rows): a quarter of a tower layer's work instead of an implicit GEMM
                const bool stem64 = on && i == 0 && !conv2 && m.tower[0].cin <= 64 && m.tower[0].k == 3 &&
                                    kz::board_conv_supported(dtype, m.h, m.w, 64, m.tower[0].cout);
                const bool board = on && kz::board_conv_supported(dtype, m.h, m.w, m.tower[i].cin, m.tower[i].cout);
                if (use_board_split && i == 0 && m.tower[0].cin <= 32 && m.tower[0].k == 3) {
                    // the stem through the same kernel: its input planes as one 32-channel chunk of (hi, lo) rows — an
                    // eighth of a 256-channel layer's work instead of an exact-f32 implicit GEMM and a splitting pass
                    stem_split = true;
                    if (upload_board_conv_split(m.tower[0], tower[0], 32)) return 1;
                } else if (use_board_split && i >= 1) {  // (a stem of more than 32 planes stays an exact-f32 implicit GEMM)
                    if (upload_board_conv_split(m.tower[i], tower[i])) return 1;
                } else if (stem64) {
                    stem_cin_p = 64;
                    if (upload_board_conv(m.tower[0], tower[0], 64)) return 1;
                } else if (board ? upload_board_conv(m.tower[i], tower[i]) : upload_conv(m.tower[i], tower[i])) {
                    return 1;
                }
            }
        }

        // scalar head: w0 [hc][C] is the OIHW 1x1 conv as is; w1 keeps the channel-major flatten order
        if (upload_f32(m.sh_conv.w, &sh_w0) || upload_f32(m.sh_conv.b, &sh_b0) || upload_f32(m.sh_fc0.w, &sh_w1) ||
            upload_f32(m.sh_fc0.b, &sh_b1) || upload_f32(m.sh_fc1.w, &sh_w2) || upload_f32(m.sh_fc1.b, &sh_b2))
            return 1;
        {
            std::vector<float> wt((size_t)m.sh_fc0.in * m.sh_fc0.out);
            for (int o = 0; o < m.sh_fc0.out; o++)
                for (int i = 0; i < m.sh_fc0.in; i++) wt[(size_t)i * m.sh_fc0.out + o] = m.sh_fc0.w[(size_t)o * m.sh_fc0.in + i];
            if (upload_f32(wt, &sh_w1t)) return 1;
        }

        if (fused_heads) return 0;  // the policy head lives in the tower's weight stream
        switch (m.policy_kind) {
            case kz::POLICY_ATAXX_CONV:
            case kz::POLICY_CONV:
                if (upload_conv(m.p_conv0, p_conv0)) return 1;
                if (upload_f32(m.p_conv1.w, &p_w1) || upload_f32(m.p_conv1.b, &p_b1)) return 1;
                if (m.policy_extra_moves) {
                    if (upload_f32(m.p_extra_conv.w, &pe_wc) || upload_f32(m.p_extra_conv.b, &pe_bc) ||
                        upload_f32(m.p_extra_fc.w, &pe_wl) || upload_f32(m.p_extra_fc.b, &pe_bl))
                        return 1;
                    if (m.sh_conv.cout == 4 && m.p_extra_conv.cout == 1 && m.p_extra_conv.cin == m.sh_conv.cin) {
                        std::vector<float> w0x(m.sh_conv.w);  // [4][C] ...
                        w0x.insert(w0x.end(), m.p_extra_conv.w.begin(), m.p_extra_conv.w.end());  // ... + [1][C]
                        if (upload_f32(w0x, &sh_w0x)) return 1;
                    }
                }
                break;
            case kz::POLICY_ARIMAA: {
                // bulk: the conv policy heads' pair of 1x1 convolutions with four planes; the scalar branch: a second ScalarHead
                if (upload_conv(m.p_conv0, p_conv0)) return 1;
                if (upload_f32(m.p_conv1.w, &p_w1) || upload_f32(m.p_conv1.b, &p_b1)) return 1;
                if (upload_f32(m.pa_conv.w, &pa_w0) || upload_f32(m.pa_conv.b, &pa_b0) || upload_f32(m.pa_fc0.w, &pa_w1) ||
                    upload_f32(m.pa_fc0.b, &pa_b1) || upload_f32(m.pa_fc1.w, &pa_w2) || upload_f32(m.pa_fc1.b, &pa_b2))
                    return 1;
                std::vector<float> wt((size_t)m.pa_fc0.in * m.pa_fc0.out);
                for (int o = 0; o < m.pa_fc0.out; o++)
                    for (int i = 0; i < m.pa_fc0.in; i++) wt[(size_t)i * m.pa_fc0.out + o] = m.pa_fc0.w[(size_t)o * m.pa_fc0.in + i];
                if (upload_f32(wt, &pa_w1t)) return 1;
                break;
            }
            case kz::POLICY_ATTENTION:
                if (att_heads) {
                    const int Q = m.policy_query_channels;
                    std::vector<uint16_t> packed(kz::att_heads_weight_elems(C, Q));
                    std::vector<float> bias((size_t)5 * Q + 16);
                    kz::att_heads_pack(m.p_bulk.w.data(), m.p_bulk.b.data(), m.p_under.w.data(), m.p_under.b.data(), m.sh_conv.w.data(),
                                       m.sh_conv.b.data(), C, Q, m.sh_conv.cout, packed.data(), bias.data());
                    if (upload(packed.data(), packed.size() * 2, &ah_w) || upload_f32(bias, &ah_bias)) return 1;
                } else if (upload_conv(m.p_bulk, p_bulk) || upload_conv(m.p_under, p_under)) {
                    return 1;
                }
                if (upload(m.flat_to_att.data(), m.flat_to_att.size() * 4, (void **)&flat_to_att)) return 1;
                break;
            case kz::POLICY_NONE: break;
            case kz::POLICY_DENSE: {
                int ch = C, ch_p = cp;
                if (m.dense_hidden_channels) {
                    if (upload_conv(m.p_conv0, p_conv0)) return 1;
                    ch = m.dense_hidden_channels;
                    ch_p = p_conv0.cout_p;
                }
                if (m.dense_hidden_size) {
                    if (upload_flat_linear(m.p_fc0, ch, hw, ch_p, p_fc0)) return 1;
                    if (upload_linear(m.p_fc1, p_fc1)) return 1;
                } else {
                    if (upload_flat_linear(m.p_fc1, ch, hw, ch_p, p_fc1)) return 1;
                }
                break;
            }
        }
        return 0;
    }
};

std::mutex g_cache_mutex;
std::map<std::tuple<const Model *, int, int, bool, bool, bool>, std::weak_ptr<DeviceWeights>> g_cache;
