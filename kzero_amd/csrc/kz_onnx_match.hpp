// kz_onnx_match.hpp — the architecture matcher of the ONNX reader: walks the normalised graph (kz_onnx.cpp: wire reader and
// normalising pass) and fills a kz::Model — Conv / Gemm extraction with BatchNorm folding, the AttentionTower
// (python/lib/model/attention.py) and DenseNetwork (python/lib/model/simple.py) matchers.  Part of kz_onnx.cpp's translation
// unit (included inside its anonymous namespace, behind OGraph / ONode / OTensor / fail).
#pragma once

struct Matcher {
    const OGraph &g;

    const ONode *producer(const std::string &name, const char *op = nullptr) const {
        auto it = g.producer.find(name);
        if (it == g.producer.end()) return nullptr;
        const ONode &n = g.nodes[it->second];
        return (!op || n.op == op) ? &n : nullptr;
    }
    const ONode &expect_producer(const std::string &name, const char *op) const {
        const ONode *n = producer(name, op);
        if (!n) fail(std::string("expected a ") + op + " producing '" + name + "'");
        return *n;
    }
    std::vector<const ONode *> consumers(const std::string &name, const char *op = nullptr) const {
        std::vector<const ONode *> r;
        auto it = g.consumers.find(name);
        if (it != g.consumers.end())
            for (int i : it->second)
                if (!op || g.nodes[i].op == op) r.push_back(&g.nodes[i]);
        return r;
    }
    // initializer, Identity of an initializer (the exporter de-duplicates equal tensors), or Constant node
    const OTensor &constant(const std::string &name) const {
        auto it = g.init.find(name);
        if (it != g.init.end()) return it->second;
        if (const ONode *n = producer(name)) {
            if (n->op == "Identity") return constant(n->in[0]);
            if (n->op == "Constant") {
                auto a = n->attr.find("value");
                if (a != n->attr.end() && a->second.t) return *a->second.t;
            }
        }
        fail("'" + name + "' is not a constant");
    }
    const std::vector<float> &floats(const std::string &name, size_t expect) const {
        const OTensor &t = constant(name);
        if (t.dtype != 1 || t.f.size() != expect) fail("tensor '" + name + "' has the wrong type or size");
        return t.f;
    }

    Conv conv(const ONode &n, int k) const {
        if (n.op != "Conv" || n.in.size() != 3) fail("expected Conv with bias");
        const OTensor &w = constant(n.in[1]);
        if (w.dtype != 1 || w.dims.size() != 4 || w.dims[2] != k || w.dims[3] != k)
            fail("Conv '" + n.out[0] + "': expected a " + std::to_string(k) + "x" + std::to_string(k) + " kernel");
        for (int64_t p : n.attr_ints("pads")) if (p != k / 2) fail("Conv: padding must be k/2");
        for (int64_t s : n.attr_ints("strides")) if (s != 1) fail("Conv: stride must be 1");
        for (int64_t d : n.attr_ints("dilations")) if (d != 1) fail("Conv: dilation must be 1");
        if (n.attr_i("group", 1) != 1) fail("Conv: group must be 1");
        Conv c;
        c.cout = (int)w.dims[0];
        c.cin = (int)w.dims[1];
        c.k = k;
        c.w = w.f;
        c.b = floats(n.in[2], (size_t)c.cout);
        return c;
    }
    // y = s*x + t of a BatchNormalization node (inputs: X, scale, B, mean, var)
    void bn_affine(const ONode &n, int ch, std::vector<float> &s, std::vector<float> &t) const {
        if (n.op != "BatchNormalization" || n.in.size() != 5) fail("malformed BatchNormalization");
        const auto &gamma = floats(n.in[1], ch), &beta = floats(n.in[2], ch), &mean = floats(n.in[3], ch), &var = floats(n.in[4], ch);
        const double eps = n.attr_f("epsilon", 1e-5f);
        s.resize(ch);
        t.resize(ch);
        for (int i = 0; i < ch; i++) {
            const double sd = (double)gamma[i] / std::sqrt((double)var[i] + eps);
            s[i] = (float)sd;
            t[i] = (float)((double)beta[i] - sd * (double)mean[i]);
        }
    }
    // if `name` feeds exactly one BatchNormalization, fold it into the conv and return the BN's output
    std::string fold_optional_bn(Conv &c, const std::string &name) const {
        auto bns = consumers(name, "BatchNormalization");
        if (bns.size() != 1 || consumers(name).size() != 1) return name;
        std::vector<float> s, t;
        bn_affine(*bns[0], c.cout, s, t);
        const size_t per = (size_t)c.cin * c.k * c.k;
        for (int o = 0; o < c.cout; o++) {
            for (size_t i = 0; i < per; i++) c.w[o * per + i] *= s[o];
            c.b[o] = s[o] * c.b[o] + t[o];
        }
        return bns[0]->out[0];
    }
    const ONode &sole_consumer(const std::string &name, const char *op) const {
        auto c = consumers(name);
        if (c.size() != 1 || c[0]->op != op) fail(std::string("expected '") + name + "' to feed exactly one " + op);
        return *c[0];
    }
    Linear gemm(const ONode &n) const {
        if (n.op != "Gemm" || n.in.size() != 3) fail("expected Gemm with bias");
        if (n.attr_i("transA", 0) != 0 || n.attr_f("alpha", 1.f) != 1.f || n.attr_f("beta", 1.f) != 1.f)
            fail("Gemm must be x * W^T + b (or x * W + b)");
        const OTensor &w = constant(n.in[1]);
        if (w.dtype != 1 || w.dims.size() != 2) fail("Gemm weight must be 2-D");
        Linear l;
        if (n.attr_i("transB", 0) == 1) {  // nn.Linear's own layout [out, in]
            l.out = (int)w.dims[0];
            l.in = (int)w.dims[1];
            l.w = w.f;
        } else {  // [in, out]: MatMul + Add, or an exporter that transposes the weight
            l.in = (int)w.dims[0];
            l.out = (int)w.dims[1];
            l.w.resize(w.f.size());
            for (int o = 0; o < l.out; o++)
                for (int i = 0; i < l.in; i++) l.w[(size_t)o * l.in + i] = w.f[(size_t)i * l.out + o];
        }
        l.b = floats(n.in[2], (size_t)l.out);
        return l;
    }
    // ---- the AttentionTower network (python/lib/model/attention.py:8-136) as torch's exporter writes it: MatMul against constant [in, out]
    // matrices (bias-free Linear layers), Reshape / Transpose / Slice around them, LayerNorm spelled out ----
    std::vector<const ONode *> data_consumers(const std::string &name) const {  // (Shape nodes read only the dimensions)
        std::vector<const ONode *> r;
        for (auto *c : consumers(name))
            if (c->op != "Shape") r.push_back(c);
        return r;
    }
    // the node of type `op` that reads `name` directly or through a chain of Reshape nodes; nullptr if there is none
    const ONode *via_reshape(const std::string &name, const char *op, int depth = 0) const {
        for (auto *c : data_consumers(name)) {
            if (c->op == op && c->in[0] == name) return c;
            if (c->op == "Reshape" && c->in[0] == name && depth < 4)
                if (const ONode *n = via_reshape(c->out[0], op, depth + 1)) return n;
        }
        return nullptr;
    }
    // the producer of `name`, looking through Reshape nodes
    const ONode *producer_via_reshape(std::string name, const char *op) const {
        for (int hop = 0; hop < 5; hop++) {
            const ONode *n = producer(name);
            if (!n) return nullptr;
            if (n->op == op) return n;
            if (n->op != "Reshape") return nullptr;
            name = n->in[0];
        }
        return nullptr;
    }
    // MatMul(x, W) with W a constant [in, out] matrix -> nn.Linear's [out][in] rows
    void matmul_weight(const ONode &n, int &in, int &out, std::vector<float> &w) const {
        if (n.op != "MatMul" || n.in.size() != 2) fail("expected MatMul");
        const OTensor &t = constant(n.in[1]);
        if (t.dtype != 1 || t.dims.size() != 2 || t.dims[0] <= 0 || t.dims[1] <= 0 || t.dims[0] > (1 << 20) || t.dims[1] > (1 << 20) ||
            t.f.size() != (size_t)(t.dims[0] * t.dims[1])) fail("MatMul '" + n.out[0] + "': the weight must be a constant 2-D matrix");
        in = (int)t.dims[0];
        out = (int)t.dims[1];
        w.resize(t.f.size());
        for (int o = 0; o < out; o++)
            for (int i = 0; i < in; i++) w[(size_t)o * in + i] = t.f[(size_t)i * out + o];
    }
    bool scalar_const(const std::string &name, float &v) const {
        auto it = g.init.find(name);
        if (it == g.init.end() || it->second.dtype != 1 || it->second.f.size() != 1) return false;
        v = it->second.f[0];
        return true;
    }
    std::vector<int64_t> int_param(const ONode &n, const char *attr, size_t input) const {
        auto a = n.attr.find(attr);
        if (a != n.attr.end()) return a->second.ints;
        auto it = g.init.find(n.in[input]);
        if (it == g.init.end() || it->second.dtype != 7) return {};
        return it->second.i;
    }
    void expect_perm(const ONode &n, std::initializer_list<int64_t> perm) const {
        if (n.op != "Transpose" || n.attr_ints("perm") != std::vector<int64_t>(perm)) fail("attention tower: unexpected Transpose at '" + n.out[0] + "'");
    }
    // z -> LayerNorm over the last axis without parameters (attention.py:80-81): ReduceMean, Sub, Pow 2, ReduceMean, Add eps,
    // Sqrt, Div (opset < 17), or one LayerNormalization node whose scale / bias are absent or trivial.  Returns the output.
    std::string layernorm(const std::string &z, float &eps) const {
        for (auto *c : data_consumers(z))
            if (c->op == "LayerNormalization") {
                if (c->attr_i("axis", -1) != -1) fail("LayerNormalization must run over the last axis");
                for (size_t k = 1; k < c->in.size(); k++) {
                    if (c->in[k].empty()) continue;
                    const OTensor &p = constant(c->in[k]);
                    for (float v : p.f) if (v != (k == 1 ? 1.0f : 0.0f)) fail("LayerNormalization with parameters (the reference's has none)");
                }
                eps = c->attr_f("epsilon", 1e-5f);
                return c->out[0];
            }
        const ONode *mean = nullptr, *sub = nullptr;
        for (auto *c : data_consumers(z)) {
            if (c->op == "ReduceMean") mean = c;
            else if (c->op == "Sub" && c->in[0] == z) sub = c;
        }
        if (!mean || !sub || sub->in[1] != mean->out[0] || data_consumers(z).size() != 2) fail("attention tower: expected a LayerNorm behind '" + z + "'");
        auto last_axis = [&](const ONode &r) {
            const std::vector<int64_t> ax = int_param(r, "axes", 1);
            if (ax.size() != 1 || (ax[0] != -1 && ax[0] != 2) || r.attr_i("keepdims", 1) != 1) fail("LayerNorm: ReduceMean must keep the last axis");
        };
        last_axis(*mean);
        const ONode *pow = nullptr, *div = nullptr;
        for (auto *c : data_consumers(sub->out[0])) {
            if (c->op == "Pow") pow = c;
            else if (c->op == "Div" && c->in[0] == sub->out[0]) div = c;
        }
        float two = 0;
        if (!pow || !div || !scalar_const(pow->in[1], two) || two != 2.0f) fail("LayerNorm: expected (x - mean)^2 and the division");
        const ONode &var = sole_consumer(pow->out[0], "ReduceMean");
        last_axis(var);
        const ONode &add = sole_consumer(var.out[0], "Add");
        if (!scalar_const(add.in[1], eps) && !scalar_const(add.in[0], eps)) fail("LayerNorm: epsilon must be a constant");
        const ONode &sq = sole_consumer(add.out[0], "Sqrt");
        if (div->in[1] != sq.out[0]) fail("LayerNorm: division by sqrt(var + eps)");
        return div->out[0];
    }
    // x * alpha + f(x): the Add that joins Mul(x, alpha) with `fx` (possibly through Reshape nodes); returns its output
    std::string deepnorm_add(const std::string &x, const ONode &f_last, float &alpha) const {
        const ONode *mul = nullptr;
        for (auto *c : data_consumers(x))
            if (c->op == "Mul" && (scalar_const(c->in[1], alpha) || scalar_const(c->in[0], alpha))) mul = c;
        if (!mul) fail("attention tower: expected x * alpha beside '" + x + "'");
        const ONode &add = sole_consumer(mul->out[0], "Add");
        const std::string &other = add.in[0] == mul->out[0] ? add.in[1] : add.in[0];
        const ONode *src = producer_via_reshape(other, f_last.op.c_str());
        if (src != &f_last) fail("attention tower: the residual Add does not join x * alpha with the sub-layer's output");
        return add.out[0];
    }
    // Returns the name of the tower output [B, d_model, h, w]; fills the model's attention-tower fields.
    std::string attention_tower(Model &m, const ONode &first) const {
        const int n = m.h * m.w;
        expect_perm(first, {2, 3, 0, 1});  // "b c h w -> (h w) b c" (attention.py:36)
        const ONode *ex = via_reshape(first.out[0], "MatMul");
        if (!ex) fail("attention tower: no expand MatMul behind the input's Transpose");
        int in = 0, D = 0;
        matmul_weight(*ex, in, D, m.att_expand);
        if (in != m.c_in) fail("attention tower: expand reads " + std::to_string(in) + " input channels");
        m.channels = D;
        const ONode *emb = via_reshape(ex->out[0], "Add");
        if (!emb) fail("attention tower: no embedding Add");
        const OTensor &e = constant(emb->in[1]);  // [n, 1, d_model] (embedding.unsqueeze(1), attention.py:40)
        if (e.dtype != 1 || e.f.size() != (size_t)n * D) fail("attention tower: embedding size");
        m.att_embedding = e.f;
        std::string x = emb->out[0];
        m.tower_kind = TOWER_ATTENTION;
        for (;;) {
            const ONode *qkv = via_reshape(x, "MatMul");
            if (!qkv) break;  // the "(h w) b c -> b c h w" Reshape + Transpose follow
            AttLayer L;
            int din = 0, nqkv = 0, dout = 0, kout = 0, dff = 0, d2 = 0;
            matmul_weight(*qkv, din, nqkv, L.qkv);
            if (din != D) fail("attention tower: project_qkv input size");
            // .view(n, b * heads, d_kqv) and the three slices of its last axis (attention.py:106-111)
            const ONode *sl[3] = {nullptr, nullptr, nullptr};
            int64_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            int ns = 0;
            std::function<void(const std::string &, int)> find_slices = [&](const std::string &name, int depth) {
                for (auto *c : data_consumers(name)) {
                    if (c->op == "Slice" && ns < 3) {
                        const std::vector<int64_t> st = int_param(*c, "starts", 1), en = int_param(*c, "ends", 2), ax = int_param(*c, "axes", 3),
                                                   sp = int_param(*c, "steps", 4);
                        if (st.size() != 1 || en.size() != 1 || ax.size() != 1 || (ax[0] != 2 && ax[0] != -1) || (!sp.empty() && (sp.size() != 1 || sp[0] != 1)))
                            fail("attention tower: q / k / v must be slices of the last axis");
                        sl[ns] = c; lo[ns] = st[0]; hi[ns] = en[0]; ns++;
                    } else if (c->op == "Reshape" && depth < 3) {
                        find_slices(c->out[0], depth + 1);
                    }
                }
            };
            find_slices(qkv->out[0], 0);
            if (ns != 3) fail("attention tower: expected three slices (q, k, v) of project_qkv's output");
            for (int a = 0; a < 3; a++)  // order by start
                for (int b = a + 1; b < 3; b++)
                    if (lo[b] < lo[a]) { std::swap(lo[a], lo[b]); std::swap(hi[a], hi[b]); std::swap(sl[a], sl[b]); }
            const int64_t dk = hi[0];
            if (lo[0] != 0 || dk <= 0 || dk > (1 << 16) || lo[1] != dk || hi[1] != 2 * dk || lo[2] != 2 * dk || hi[2] <= lo[2])
                fail("attention tower: q | k | v slice ranges");
            // logits = bmm(q^T, k^T^T), softmax over the keys, bmm with v (attention.py:117-122; no scale factor)
            const ONode &tq = sole_consumer(sl[0]->out[0], "Transpose"), &tk = sole_consumer(sl[1]->out[0], "Transpose"),
                        &tv = sole_consumer(sl[2]->out[0], "Transpose");
            expect_perm(tq, {1, 0, 2});
            expect_perm(tk, {1, 2, 0});
            expect_perm(tv, {1, 0, 2});
            const ONode &logits = sole_consumer(tq.out[0], "MatMul");
            if (logits.in[0] != tq.out[0] || logits.in[1] != tk.out[0]) fail("attention tower: logits must be q k^T");
            const ONode &sm = sole_consumer(logits.out[0], "Softmax");
            const int64_t sax = sm.attr_i("axis", -1);
            if (sax != 2 && sax != -1) fail("attention tower: softmax must run over the keys");
            const ONode &av = sole_consumer(sm.out[0], "MatMul");
            if (av.in[0] != sm.out[0] || av.in[1] != tv.out[0]) fail("attention tower: attention output must be weights v");
            const ONode &tb = sole_consumer(av.out[0], "Transpose");
            expect_perm(tb, {1, 0, 2});
            const ONode *po = via_reshape(tb.out[0], "MatMul");
            if (!po) fail("attention tower: no project_out MatMul");
            matmul_weight(*po, kout, dout, L.out);
            if (dout != D) fail("attention tower: project_out output size");
            // heads and d_v from the two matrices: heads (2 d_k + d_v) = project_qkv's rows, heads d_v = project_out's columns
            if ((nqkv - kout) <= 0 || (nqkv - kout) % (2 * dk)) fail("attention tower: head count does not divide project_qkv");
            const int H = (int)((nqkv - kout) / (2 * dk));
            if (kout % H) fail("attention tower: d_v");
            const int dv = kout / H;
            if (hi[2] < 2 * dk + dv) fail("attention tower: the v slice is shorter than d_v");
            float alpha = 1.0f, alpha2 = 1.0f, eps = 1e-5f, eps2 = 1e-5f;
            const std::string y = layernorm(deepnorm_add(x, *po, alpha), eps);
            const ONode *f0 = via_reshape(y, "MatMul");
            if (!f0) fail("attention tower: no feed-forward MatMul");
            matmul_weight(*f0, din, dff, L.ff0);
            if (din != D) fail("attention tower: ff.0 input size");
            const ONode &relu = sole_consumer(f0->out[0], "Relu");
            const ONode &f1 = sole_consumer(relu.out[0], "MatMul");
            matmul_weight(f1, din, d2, L.ff1);
            if (din != dff || d2 != D) fail("attention tower: ff.2 sizes");
            const std::string next = layernorm(deepnorm_add(y, f1, alpha2), eps2);
            for (float v : {alpha, alpha2})
                if (!(std::isfinite(v) && v > 0.0f)) fail("attention tower: the residual scale (alpha) must be finite and > 0");
            for (float v : {eps, eps2})
                if (!(std::isfinite(v) && v >= 0.0f)) fail("attention tower: LayerNorm epsilon must be finite and >= 0");
            if (m.att_layers.empty()) {
                m.att_heads = H; m.att_dk = (int)dk; m.att_dv = dv; m.att_dff = dff; m.att_alpha = alpha; m.ln_eps = eps;
            }
            if (H != m.att_heads || dk != m.att_dk || dv != m.att_dv || dff != m.att_dff || alpha != m.att_alpha || alpha2 != m.att_alpha ||
                eps != m.ln_eps || eps2 != m.ln_eps)
                fail("attention tower: the encoder layers differ in shape, alpha or epsilon");
            m.att_layers.push_back(std::move(L));
            x = next;
            if (m.att_layers.size() > 1024) fail("attention tower: too many layers");
        }
        if (m.att_layers.empty()) fail("attention tower: no encoder layer");
        m.depth = (int)m.att_layers.size();
        const ONode *back = via_reshape(x, "Transpose");  // "(h w) b c -> b c h w" (attention.py:43-44)
        if (!back) fail("attention tower: no Transpose back to [b, c, h, w]");
        expect_perm(*back, {2, 3, 0, 1});
        m.final_scale.assign(D, 1.0f);
        m.final_shift.assign(D, 0.0f);
        return back->out[0];
    }

    // ---- DenseNetwork with its DenseBlocks (python/lib/model/simple.py:7-52): Flatten, Gemm, blocks of BatchNormalization, Relu, Gemm,
    // BatchNormalization, Relu, Gemm (+ Add with the block's input), BatchNormalization, Relu, Gemm; scalars = Slice [:, :5],
    // policy = Slice [:, 5:] (reshaped to the game's policy shape) ----
    void dense_network(Model &m, const ONode &flatten) const {
        if (flatten.attr_i("axis", 1) != 1) fail("dense network: Flatten must keep the batch axis");
        const ONode &gin = sole_consumer(flatten.out[0], "Gemm");
        m.dn_in = gemm(gin);
        const int size = m.dn_in.out;
        if (m.dn_in.in != m.c_in * m.h * m.w) fail("dense network: the first Linear does not read the whole input");
        m.tower_kind = TOWER_DENSE_NET;
        m.policy_kind = POLICY_NONE;
        m.channels = size;
        std::string cur = gin.out[0];
        bool first = true;
        for (;;) {
            auto bns = consumers(cur, "BatchNormalization");
            if (bns.size() != 1) fail("dense network: expected one BatchNormalization behind '" + cur + "'");
            std::vector<float> sa, ta;
            bn_affine(*bns[0], size, sa, ta);
            const ONode &r1 = sole_consumer(bns[0]->out[0], "Relu");
            const ONode &g1 = sole_consumer(r1.out[0], "Gemm");
            Linear l1 = gemm(g1);
            if (l1.in != size) fail("dense network: Linear input size");
            if (!consumers(g1.out[0], "Slice").empty()) {  // the last Linear: scalars | policy
                if (consumers(cur).size() != 1) fail("dense network: the last block's output feeds something else too");
                m.dn_sf = sa;
                m.dn_tf = ta;
                m.dn_out = std::move(l1);
                cur = g1.out[0];
                break;
            }
            Model::DnBlock b;
            b.sa = sa;
            b.ta = ta;
            b.la = std::move(l1);
            const ONode &bn2 = sole_consumer(g1.out[0], "BatchNormalization");
            bn_affine(bn2, size, b.sb, b.tb);
            const ONode &r2 = sole_consumer(bn2.out[0], "Relu");
            const ONode &g2 = sole_consumer(r2.out[0], "Gemm");
            b.lb = gemm(g2);
            if (b.la.out != size || b.lb.in != size || b.lb.out != size) fail("dense network: block sizes");
            auto adds = consumers(g2.out[0], "Add");
            bool res = false;
            std::string next = g2.out[0];
            if (adds.size() == 1 && consumers(g2.out[0]).size() == 1) {
                const ONode &add = *adds[0];
                if (!((add.in[0] == cur && add.in[1] == g2.out[0]) || (add.in[1] == cur && add.in[0] == g2.out[0])))
                    fail("dense network: the residual Add does not join the block's input with its output");
                res = true;
                next = add.out[0];
            } else if (consumers(cur).size() != 1) {
                fail("dense network: a block input with two consumers but no residual Add");
            }
            if (!first && res != m.dn_res) fail("dense network: blocks with and without residual");
            m.dn_res = res;
            first = false;
            m.dn_blocks.push_back(std::move(b));
            cur = next;
            if (m.dn_blocks.size() > 4096) fail("dense network: too many blocks");
        }
        m.depth = (int)m.dn_blocks.size();
        // scalars = output[:, :5]; policy = output[:, 5:].view(-1, *policy_shape) (simple.py:29-33)
        const int outs = m.dn_out.out;
        if (outs <= 5) fail("dense network: the last Linear must yield 5 scalars and the policy");
        auto range_of = [&](const ONode &sl, int64_t &lo, int64_t &hi) {
            const std::vector<int64_t> st = int_param(sl, "starts", 1), en = int_param(sl, "ends", 2), ax = int_param(sl, "axes", 3),
                                       sp = int_param(sl, "steps", 4);
            if (st.size() != 1 || en.size() != 1 || ax.size() != 1 || ax[0] != 1 || (!sp.empty() && (sp.size() != 1 || sp[0] != 1)))
                fail("dense network: the outputs must be slices of axis 1");
            lo = st[0];
            hi = std::min<int64_t>(en[0], outs);
        };
        const ONode &ss = expect_producer("scalars", "Slice");
        const ONode *ps = producer("policy");
        if (ps && ps->op == "Reshape") ps = producer(ps->in[0]);
        if (!ps || ps->op != "Slice" || ss.in[0] != cur || ps->in[0] != cur) fail("dense network: scalars / policy are not slices of the last Linear");
        int64_t lo = 0, hi = 0;
        range_of(ss, lo, hi);
        if (lo != 0 || hi != 5) fail("dense network: scalars must be columns 0..5");
        range_of(*ps, lo, hi);
        if (lo != 5 || hi != outs) fail("dense network: the policy must be columns 5..");
        m.policy_len = outs - 5;
        m.final_scale.assign(size, 1.0f);
        m.final_shift.assign(size, 0.0f);
    }

    // name <- Relu <- Conv1x1 <- src ?  returns the conv
    bool relu_conv1x1(const std::string &name, const std::string &src, Conv &out) const {
        const ONode *r = producer(name, "Relu");
        if (!r) return false;
        const ONode *c = producer(r->in[0], "Conv");
        if (!c || c->in[0] != src) return false;
        out = conv(*c, 1);
        return true;
    }
};
