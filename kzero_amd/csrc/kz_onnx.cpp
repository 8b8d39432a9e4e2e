// kz_onnx.cpp — ONNX reader for the networks kZero's trainer exports (python/lib/save_onnx.py:60-122: opset 10,
// input "input", outputs "scalars" and "policy", dynamic batch axis), replacing
// `load_graph_from_onnx_path` + `optimize_graph` (rust/kz-selfplay/src/server/server_alphazero.rs:126-128) for the
// PredictionHeads(ResTower, ScalarHead, <policy head>) family (python/lib/model/post_act.py:187-211).
//
// No protobuf library: the wire format is read directly (varint / length-delimited / fixed32 / fixed64), only the
// ModelProto / GraphProto / NodeProto / AttributeProto / TensorProto / ValueInfoProto fields the exporter emits.
// The graph is not interpreted generically: it is pattern-matched back into the architecture the engine implements
// (stem conv, [conv (BN) relu conv (BN) relu add]*, final BatchNormalization, scalar head, one of the four policy
// heads); anything else is rejected with a message.  The exporter has already folded the in-block Conv+BN pairs
// (eval mode); a BatchNormalization that does follow a Conv is folded here.
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <set>

#include "kz_model.hpp"

namespace kz {
namespace {

struct Wire {
    const uint8_t *p, *end;
    bool bad = false;
    bool more() const { return !bad && p < end; }
    uint64_t varint() {
        uint64_t r = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) { bad = true; return 0; }
            const uint8_t c = *p++;
            r |= (uint64_t)(c & 0x7f) << shift;
            if (!(c & 0x80)) return r;
        }
        bad = true;
        return 0;
    }
    // reads the next field; for wire type 2 `sub` spans the payload
    bool field(int &num, int &type, uint64_t &value, Wire &sub) {
        const uint64_t key = varint();
        if (bad) return false;
        num = (int)(key >> 3);
        type = (int)(key & 7);
        value = 0;
        switch (type) {
            case 0: value = varint(); break;
            case 1: if (end - p < 8) { bad = true; return false; } memcpy(&value, p, 8); p += 8; break;
            case 5: { if (end - p < 4) { bad = true; return false; } uint32_t v; memcpy(&v, p, 4); value = v; p += 4; break; }
            case 2: {
                const uint64_t n = varint();
                if (bad || (uint64_t)(end - p) < n) { bad = true; return false; }
                sub = Wire{p, p + n};
                p += n;
                break;
            }
            default: bad = true; return false;
        }
        return !bad;
    }
    std::string str() const { return std::string(reinterpret_cast<const char *>(p), end - p); }
};

struct OTensor {
    std::string name;
    std::vector<int64_t> dims;
    int dtype = 0;  // 1 = FLOAT, 7 = INT64
    std::vector<float> f;
    std::vector<int64_t> i;
    size_t count() const { size_t n = 1; for (auto d : dims) n *= (size_t)d; return n; }
};

struct OAttr {
    float f = 0;
    int64_t i = 0;
    std::vector<int64_t> ints;
    std::shared_ptr<OTensor> t;
};

// node input / output names: indexing past the end yields the empty name ("absent", as ONNX writes an omitted optional
// input), which no tensor or node answers to — so a malformed node fails the pattern match with a message instead of
// reading out of bounds
struct Names : std::vector<std::string> {
    const std::string &operator[](size_t i) const {
        static const std::string none;
        return i < size() ? std::vector<std::string>::operator[](i) : none;
    }
};

struct ONode {
    std::string op;
    Names in, out;
    std::map<std::string, OAttr> attr;
    int64_t attr_i(const char *k, int64_t dflt) const { auto it = attr.find(k); return it == attr.end() ? dflt : it->second.i; }
    float attr_f(const char *k, float dflt) const { auto it = attr.find(k); return it == attr.end() ? dflt : it->second.f; }
    std::vector<int64_t> attr_ints(const char *k) const { auto it = attr.find(k); return it == attr.end() ? std::vector<int64_t>{} : it->second.ints; }
};

struct OGraph {
    std::vector<ONode> nodes;
    std::map<std::string, OTensor> init;
    std::vector<std::pair<std::string, std::vector<int64_t>>> inputs;  // dims: -1 = symbolic
    std::vector<std::string> outputs;
    std::map<std::string, int> producer;
    std::map<std::string, std::vector<int>> consumers;
};

struct Fail {
    std::string msg;
};
[[noreturn]] void fail(const std::string &m) { throw Fail{"unsupported ONNX graph: " + m}; }

void read_packed_ints(Wire w, std::vector<int64_t> &out) {
    while (w.more()) out.push_back((int64_t)w.varint());
}

OTensor read_tensor(Wire w) {
    OTensor t;
    std::string raw;
    int num, type;
    uint64_t v;
    Wire sub{nullptr, nullptr};
    while (w.more() && w.field(num, type, v, sub)) {
        if (num == 1) { if (type == 0) t.dims.push_back((int64_t)v); else read_packed_ints(sub, t.dims); }
        else if (num == 2) t.dtype = (int)v;
        else if (num == 8) t.name = sub.str();
        else if (num == 9) raw = sub.str();
        else if (num == 4) {  // float_data
            if (type == 5) { float f; uint32_t u = (uint32_t)v; memcpy(&f, &u, 4); t.f.push_back(f); }
            else for (const uint8_t *q = sub.p; q + 4 <= sub.end; q += 4) { float f; memcpy(&f, q, 4); t.f.push_back(f); }
        } else if (num == 7) {  // int64_data
            if (type == 0) t.i.push_back((int64_t)v); else read_packed_ints(sub, t.i);
        }
    }
    if (w.bad) fail("corrupt TensorProto");
    const size_t n = t.count();
    if (t.dtype == 1) {
        if (!raw.empty()) { if (raw.size() != n * 4) fail("tensor '" + t.name + "': raw_data size"); t.f.resize(n); memcpy(t.f.data(), raw.data(), n * 4); }
        if (t.f.size() != n) fail("tensor '" + t.name + "': element count");
    } else if (t.dtype == 7) {
        if (!raw.empty()) { if (raw.size() != n * 8) fail("tensor '" + t.name + "': raw_data size"); t.i.resize(n); memcpy(t.i.data(), raw.data(), n * 8); }
        if (t.i.size() != n) fail("tensor '" + t.name + "': element count");
    } else {
        fail("tensor '" + t.name + "': data type " + std::to_string(t.dtype) + " (only FLOAT and INT64)");
    }
    return t;
}

ONode read_node(Wire w) {
    ONode n;
    int num, type;
    uint64_t v;
    Wire sub{nullptr, nullptr};
    while (w.more() && w.field(num, type, v, sub)) {
        if (num == 1) n.in.push_back(sub.str());
        else if (num == 2) n.out.push_back(sub.str());
        else if (num == 4) n.op = sub.str();
        else if (num == 5) {
            std::string name;
            OAttr a;
            Wire aw = sub;
            int an, at;
            uint64_t av;
            Wire as{nullptr, nullptr};
            while (aw.more() && aw.field(an, at, av, as)) {
                if (an == 1) name = as.str();
                else if (an == 2) { uint32_t u = (uint32_t)av; memcpy(&a.f, &u, 4); }
                else if (an == 3) a.i = (int64_t)av;
                else if (an == 5) a.t = std::make_shared<OTensor>(read_tensor(as));
                else if (an == 8) { if (at == 0) a.ints.push_back((int64_t)av); else read_packed_ints(as, a.ints); }
            }
            if (aw.bad) fail("corrupt AttributeProto");
            n.attr[name] = a;
        }
    }
    if (w.bad) fail("corrupt NodeProto");
    return n;
}

std::pair<std::string, std::vector<int64_t>> read_value_info(Wire w) {
    std::string name;
    std::vector<int64_t> dims;
    int num, type;
    uint64_t v;
    Wire sub{nullptr, nullptr};
    while (w.more() && w.field(num, type, v, sub)) {
        if (num == 1) name = sub.str();
        else if (num == 2) {  // TypeProto
            Wire tw = sub; int a, b; uint64_t c; Wire ts{nullptr, nullptr};
            while (tw.more() && tw.field(a, b, c, ts)) {
                if (a != 1) continue;  // tensor_type
                Wire tt = ts; Wire sh{nullptr, nullptr};
                while (tt.more() && tt.field(a, b, c, sh)) {
                    if (a != 2) continue;  // shape
                    Wire sw = sh; Wire dm{nullptr, nullptr};
                    while (sw.more() && sw.field(a, b, c, dm)) {
                        if (a != 1) continue;  // dim
                        int64_t value = -1;
                        Wire dw = dm; Wire ds{nullptr, nullptr};
                        while (dw.more() && dw.field(a, b, c, ds))
                            if (a == 1) value = (int64_t)c;
                        dims.push_back(value);
                    }
                }
            }
        }
    }
    return {name, dims};
}

OGraph read_model(const void *blob, size_t len) {
    OGraph g;
    Wire w{static_cast<const uint8_t *>(blob), static_cast<const uint8_t *>(blob) + len};
    int num, type;
    uint64_t v;
    Wire sub{nullptr, nullptr};
    bool have_graph = false;
    while (w.more() && w.field(num, type, v, sub)) {
        if (num != 7 || type != 2) continue;  // ModelProto.graph
        have_graph = true;
        Wire gw = sub;
        Wire gs{nullptr, nullptr};
        while (gw.more() && gw.field(num, type, v, gs)) {
            if (num == 1) g.nodes.push_back(read_node(gs));
            else if (num == 5) { OTensor t = read_tensor(gs); g.init[t.name] = std::move(t); }
            else if (num == 11) g.inputs.push_back(read_value_info(gs));
            else if (num == 12) g.outputs.push_back(read_value_info(gs).first);
        }
        if (gw.bad) fail("corrupt GraphProto");
    }
    if (w.bad || !have_graph) fail("not an ONNX ModelProto");
    for (size_t i = 0; i < g.nodes.size(); i++) {
        for (auto &o : g.nodes[i].out) g.producer[o] = (int)i;
        for (auto &in : g.nodes[i].in) g.consumers[in].push_back((int)i);
    }
    return g;
}

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
        if (n.attr_i("transB", 0) != 1 || n.attr_i("transA", 0) != 0 || n.attr_f("alpha", 1.f) != 1.f || n.attr_f("beta", 1.f) != 1.f)
            fail("Gemm must be x * W^T + b");
        const OTensor &w = constant(n.in[1]);
        if (w.dtype != 1 || w.dims.size() != 2) fail("Gemm weight must be 2-D");
        Linear l;
        l.out = (int)w.dims[0];
        l.in = (int)w.dims[1];
        l.w = w.f;
        l.b = floats(n.in[2], (size_t)l.out);
        return l;
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

}  // namespace

void finalize_model(Model &m);  // kz_model.cpp

bool looks_like_onnx(const void *blob, size_t len) {
    // ModelProto starts with field 1 (ir_version, varint): key byte 0x08
    return len > 16 && static_cast<const uint8_t *>(blob)[0] == 0x08 && memcmp(blob, "KZMODEL1", 8) != 0;
}

Model *parse_onnx(const void *blob, size_t len, int n_scalar, std::string &err) {
    try {
        OGraph g = read_model(blob, len);
        Matcher M{g};
        std::unique_ptr<Model> m(new Model());

        // input [batch, C, H, W] (check_graph_shapes, rust/kz-core/src/network/common.rs:165-175)
        const std::vector<int64_t> *in_dims = nullptr;
        for (auto &in : g.inputs)
            if (in.first == "input") in_dims = &in.second;
        if (!in_dims || in_dims->size() != 4) fail("no 4-D graph input named 'input'");
        m->c_in = (int)(*in_dims)[1];
        m->h = (int)(*in_dims)[2];
        m->w = (int)(*in_dims)[3];
        if (m->c_in <= 0 || m->h <= 0 || m->w <= 0) fail("input shape must be [batch, C, H, W] with fixed C, H, W");
        if (n_scalar > m->c_in) fail("input_scalar_channels exceeds the input channels");
        m->n_scalar = n_scalar;  // < 0: unknown (the split is the mapper's knowledge, not the graph's)
        m->n_bool = n_scalar < 0 ? -1 : m->c_in - n_scalar;
        bool has_scalars = false, has_policy = false;
        for (auto &o : g.outputs) { has_scalars |= o == "scalars"; has_policy |= o == "policy"; }
        if (g.outputs.size() != 2 || !has_scalars || !has_policy) fail("outputs must be 'scalars' and 'policy' (legacy (value, wdl, policy) graphs, network/common.rs:36-45, "
                                                                        "are not taken: re-export with save_onnx.py)");
        const int hw = m->h * m->w;

        // ---- ResTower (post_act.py:201-228) ----
        auto stems = M.consumers("input", "Conv");
        if (stems.size() != 1 || M.consumers("input").size() != 1) fail("'input' must feed exactly one Conv");
        m->tower.push_back(M.conv(*stems[0], 3));
        if (m->tower[0].cin != m->c_in) fail("stem input channels");
        const int C = m->channels = m->tower[0].cout;
        std::string cur = M.fold_optional_bn(m->tower[0], stems[0]->out[0]);
        for (;;) {
            auto adds = M.consumers(cur, "Add");
            auto convs = M.consumers(cur, "Conv");
            if (adds.size() != 1) break;
            const ONode *ca = nullptr;
            for (auto *c : convs) {
                const OTensor &w = M.constant(c->in[1]);
                if (w.dims.size() == 4 && w.dims[2] == 3) ca = c;
            }
            if (!ca) fail("residual block without a 3x3 convolution");
            Conv a = M.conv(*ca, 3);
            std::string x = M.fold_optional_bn(a, ca->out[0]);
            const ONode &r1 = M.sole_consumer(x, "Relu");
            const ONode &cbn = M.sole_consumer(r1.out[0], "Conv");
            Conv b = M.conv(cbn, 3);
            x = M.fold_optional_bn(b, cbn.out[0]);
            const ONode &r2 = M.sole_consumer(x, "Relu");
            const ONode &add = *adds[0];
            // input + seq(input): the residual is added AFTER the ReLU (post_act.py:227-228)
            if (!((add.in[0] == cur && add.in[1] == r2.out[0]) || (add.in[1] == cur && add.in[0] == r2.out[0])))
                fail("residual Add does not join the block input with the block's last ReLU");
            if (a.cin != C || a.cout != C || b.cin != C || b.cout != C) fail("block channel count changes");
            m->tower.push_back(std::move(a));
            m->tower.push_back(std::move(b));
            cur = add.out[0];
        }
        m->depth = (int)(m->tower.size() - 1) / 2;
        std::string t = cur;  // tower output seen by the heads
        auto final_bn = M.consumers(cur, "BatchNormalization");
        if (final_bn.size() == 1 && M.consumers(cur).size() == 1) {
            M.bn_affine(*final_bn[0], C, m->final_scale, m->final_shift);  // post_act.py:207
            t = final_bn[0]->out[0];
        } else {
            m->final_scale.assign(C, 1.0f);
            m->final_shift.assign(C, 0.0f);
        }

        // ---- ScalarHead (post_act.py:10-23): scalars <- Gemm <- Relu <- Gemm <- Flatten <- Relu <- Conv1x1 <- t ----
        {
            const ONode &g2 = M.expect_producer("scalars", "Gemm");
            const ONode &r = M.expect_producer(g2.in[0], "Relu");
            const ONode &g1 = M.expect_producer(r.in[0], "Gemm");
            const ONode &fl = M.expect_producer(g1.in[0], "Flatten");
            if (!M.relu_conv1x1(fl.in[0], t, m->sh_conv)) fail("scalar head: expected Conv1x1 + ReLU on the tower output");
            m->sh_fc0 = M.gemm(g1);
            m->sh_fc1 = M.gemm(g2);
            if (m->sh_fc0.in != m->sh_conv.cout * hw || m->sh_fc1.in != m->sh_fc0.out || m->sh_fc1.out != 5)
                fail("scalar head shapes");
        }

        // ---- policy head ----
        const ONode *p = M.producer("policy");
        if (!p) fail("nothing produces 'policy'");
        if (p->op == "Reshape") p = M.producer(p->in[0]);  // DensePolicyHead's .view(-1, *policy_shape) (post_act.py:51)
        if (!p) fail("policy: dangling Reshape");
        auto conv_stack = [&](const std::string &flat_in, Conv &c0, Conv &c1) {  // Flatten <- Conv1x1 <- Relu <- Conv1x1 <- t
            const ONode &fl = M.expect_producer(flat_in, "Flatten");
            const ONode &c2 = M.expect_producer(fl.in[0], "Conv");
            if (!M.relu_conv1x1(c2.in[0], t, c0)) fail("policy head: expected Conv1x1 + ReLU on the tower output");
            c1 = M.conv(c2, 1);
        };
        if (p->op == "Gather" && p->attr_i("axis", 0) == 1) {
            // AttentionPolicyHead (post_act.py:115-141)
            m->policy_kind = POLICY_ATTENTION;
            const OTensor &idx = M.constant(p->in[1]);
            if (idx.dtype != 7) fail("attention head: gather indices must be INT64");
            m->policy_len = (int)idx.i.size();
            for (int64_t v : idx.i) {
                if (v < 0 || v >= 64 * 88) fail("attention head: gather index out of range");
                m->flat_to_att.push_back((int32_t)v);
            }
            const ONode &fl = M.expect_producer(p->in[0], "Flatten");
            const ONode &dv = M.expect_producer(fl.in[0], "Div");
            M.expect_producer(dv.in[0], "MatMul");
            // the two 1x1 convolutions behind the MatMul: conv_bulk reads the tower output, conv_under its rank-7 row
            std::set<int> seen;
            std::vector<std::string> stack{dv.in[0]};
            const ONode *bulk = nullptr, *under = nullptr;
            while (!stack.empty()) {
                std::string name = stack.back();
                stack.pop_back();
                auto it = g.producer.find(name);
                if (it == g.producer.end() || !seen.insert(it->second).second) continue;
                const ONode &n = g.nodes[it->second];
                if (n.op == "Conv") {
                    if (n.in[0] == t) { if (bulk) fail("attention head: two convolutions on the tower output"); bulk = &n; }
                    else { if (under) fail("attention head: unexpected convolution"); under = &n; }
                    continue;
                }
                for (auto &in : n.in) stack.push_back(in);
            }
            if (!bulk || !under) fail("attention head: conv_bulk / conv_under not found");
            const ONode &uq = M.expect_producer(under->in[0], "Unsqueeze");
            const ONode &ga = M.expect_producer(uq.in[0], "Gather");
            const OTensor &row = M.constant(ga.in[1]);
            if (ga.in[0] != t || ga.attr_i("axis", 0) != 2 || row.i.size() != 1 || row.i[0] != 7)
                fail("attention head: conv_under must read common[:, :, 7, None, :]");
            m->p_bulk = M.conv(*bulk, 1);
            m->p_under = M.conv(*under, 1);
            const int Q = m->policy_query_channels = m->p_bulk.cout / 2;
            if (m->p_bulk.cout != 2 * Q || m->p_under.cout != 3 * Q || m->h != 8 || m->w != 8) fail("attention head shapes");
            const OTensor &scale = M.constant(dv.in[1]);
            if (scale.f.size() != 1 || std::fabs(scale.f[0] - std::sqrt((float)Q)) > 1e-4f * std::sqrt((float)Q))
                fail("attention head: logits must be divided by sqrt(query_channels)");
        } else if (p->op == "Concat" && p->attr_i("axis", 0) == 1 && p->in.size() == 2) {
            const ONode *tail = M.producer(p->in[1]);
            if (!tail) fail("policy: dangling Concat input");
            conv_stack(p->in[0], m->p_conv0, m->p_conv1);
            m->policy_conv_channels = m->p_conv1.cout;
            if (tail->op == "ConstantOfShape") {
                // AtaxxConvPolicyHead (post_act.py:91-112): one constant-zero pass logit
                m->policy_kind = POLICY_ATAXX_CONV;
                auto a = tail->attr.find("value");
                if (a != tail->attr.end() && a->second.t && !a->second.t->f.empty() && a->second.t->f[0] != 0.0f)
                    fail("ataxx head: the appended column must be zero");
                m->policy_len = m->policy_conv_channels * hw + 1;
            } else if (tail->op == "Gemm") {
                // ConvPolicyHead with extra moves (post_act.py:54-88): seq_extra = Conv1x1(C->1), Flatten, Linear
                m->policy_kind = POLICY_CONV;
                m->p_extra_fc = M.gemm(*tail);
                const ONode &fl = M.expect_producer(tail->in[0], "Flatten");
                const ONode &ce = M.expect_producer(fl.in[0], "Conv");
                if (ce.in[0] != t) fail("conv head: seq_extra must read the tower output");
                m->p_extra_conv = M.conv(ce, 1);
                m->policy_extra_moves = m->p_extra_fc.out;
                if (m->p_extra_conv.cout != 1 || m->p_extra_fc.in != hw) fail("conv head: seq_extra shapes");
                m->policy_len = m->policy_conv_channels * hw + m->policy_extra_moves;
            } else {
                fail("policy: unknown Concat tail '" + tail->op + "'");
            }
        } else if (p->op == "Gemm") {
            // DensePolicyHead (post_act.py:26-51): [Conv1x1 + ReLU] -> Flatten -> [Linear + ReLU] -> Linear
            m->policy_kind = POLICY_DENSE;
            m->p_fc1 = M.gemm(*p);
            m->policy_len = m->p_fc1.out;
            std::string x = p->in[0];
            if (const ONode *r = M.producer(x, "Relu")) {
                const ONode &g0 = M.expect_producer(r->in[0], "Gemm");
                m->p_fc0 = M.gemm(g0);
                m->dense_hidden_size = m->p_fc0.out;
                x = g0.in[0];
            }
            const ONode &fl = M.expect_producer(x, "Flatten");
            int ch = C;
            if (fl.in[0] != t) {
                if (!M.relu_conv1x1(fl.in[0], t, m->p_conv0)) fail("dense head: expected [Conv1x1 + ReLU] on the tower output");
                ch = m->dense_hidden_channels = m->p_conv0.cout;
            }
            const Linear &first = m->dense_hidden_size ? m->p_fc0 : m->p_fc1;
            if (first.in != ch * hw) fail("dense head: Linear input size");
        } else {
            fail("policy is produced by '" + p->op + "': not one of the known heads");
        }
        finalize_model(*m);
        return m.release();
    } catch (const Fail &f) {
        err = f.msg;
        return nullptr;
    }
}

}  // namespace kz
