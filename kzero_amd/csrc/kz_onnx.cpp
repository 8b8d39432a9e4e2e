// kz_onnx.cpp — ONNX reader for the networks kZero's trainer exports (python/lib/save_onnx.py:60-122: opset 10,
// input "input", outputs "scalars" and "policy", dynamic batch axis), replacing
// `load_graph_from_onnx_path` + `optimize_graph` (rust/kz-selfplay/src/server/server_alphazero.rs:126-128) for the
// PredictionHeads(ResTower, ScalarHead, <policy head>) family (python/lib/model/post_act.py:187-211) and, since round 5, for
// the reference's other network definitions: PredictionHeads over an AttentionTower (python/lib/model/attention.py) and
// DenseNetwork (python/lib/model/simple.py).  This file: the wire reader, the normalising pass, parse_onnx;
// kz_onnx_match.hpp: the architecture matcher.
//
// No protobuf library: the wire format is read directly (varint / length-delimited / fixed32 / fixed64), only the
// ModelProto / GraphProto / NodeProto / AttributeProto / TensorProto / ValueInfoProto fields the exporter emits.
// The graph is not interpreted generically: it is pattern-matched back into the architecture the engine implements
// (stem conv, [conv (BN) relu conv (BN) relu add]*, final BatchNormalization, scalar head, one of the five policy
// heads; or the two other families); anything else is rejected with a message.  The exporter has already folded the in-block Conv+BN pairs
// (eval mode); a BatchNormalization that does follow a Conv is folded here.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <functional>
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

// ---------------------------------------------------------------------------------------------------------------------
// Normalising pass in front of the architecture matcher.  Exporters of other versions and settings write the same
// network with other nodes: `Reshape(x, [0, -1])` or `Reshape(x, Concat(Unsqueeze(Gather(Shape(x), 0)), [-1]))` for
// Flatten, `MatMul` + `Add` for Gemm, Gemm with transB = 0, `Constant` nodes instead of initializers (or both: initializers
// listed as graph inputs), `Identity` / `Dropout` / `Cast` no-ops, un-folded Conv -> BatchNormalization pairs (folded by
// the matcher), `Unsqueeze` / `Squeeze` / `Slice` with their parameters as attributes (opset < 10 / 13) or as inputs.
// The pass brings all of them to ONE form — the one python/lib/save_onnx.py:107-119 gives with torch 2.x at opset 10 —
// by (1) turning Constant nodes into initializers, (2) bypassing no-ops, (3) folding the integer "shape arithmetic"
// (Shape / Gather / Unsqueeze / Squeeze / Concat / Slice / Cast chains) with a small shape inference in which the batch
// axis stays symbolic, (4) rewriting Reshape-as-flatten to Flatten, identity reshapes to nothing and MatMul + Add to
// Gemm.  What the matcher then rejects is a different architecture, not a different spelling.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int64_t BATCH = INT64_MIN + 1;  // the symbolic batch size inside folded integer tensors

struct Normalizer {
    OGraph &g;
    std::map<std::string, std::vector<int64_t>> shape;  // data tensors: -1 = the batch (or another dynamic) axis
    std::map<std::string, std::vector<int64_t>> ints;   // folded INT64 tensors (1-D or scalar): BATCH = symbolic
    std::set<std::string> int_scalar;                   // ... of rank 0

    bool is_output(const std::string &name) const {
        for (auto &o : g.outputs) if (o == name) return true;
        return false;
    }
    // every use of `from` (node inputs) now reads `to`
    void replace_uses(const std::string &from, const std::string &to) {
        for (auto &n : g.nodes)
            for (auto &in : n.in) if (in == from) in = to;
    }
    // node i computes out[0] = in[0]: remove it
    void bypass(size_t i) {
        const std::string out = g.nodes[i].out[0], in = g.nodes[i].in[0];
        g.nodes.erase(g.nodes.begin() + (long)i);
        if (!is_output(out)) {
            replace_uses(out, in);
            return;
        }
        // a graph output: its name must survive — the producer of `in` writes `out` instead
        bool renamed = false;
        for (auto &n : g.nodes)
            for (auto &o : n.out) if (o == in) { o = out; renamed = true; }
        if (!renamed) fail("graph output '" + out + "' is a constant or the graph input");
        replace_uses(in, out);
    }
    const OTensor *init(const std::string &name) const {
        auto it = g.init.find(name);
        return it == g.init.end() ? nullptr : &it->second;
    }
    // INT64 initializer or folded tensor -> values
    bool get_ints(const std::string &name, std::vector<int64_t> &out) const {
        auto it = ints.find(name);
        if (it != ints.end()) { out = it->second; return true; }
        if (const OTensor *t = init(name)) {
            if (t->dtype == 7) { out = t->i; return true; }
        }
        return false;
    }
    bool is_scalar(const std::string &name) const {
        if (int_scalar.count(name)) return true;
        const OTensor *t = init(name);
        return t && t->dims.empty();
    }
    const std::vector<int64_t> *get_shape(const std::string &name) const {
        auto it = shape.find(name);
        return it == shape.end() ? nullptr : &it->second;
    }
    // parameters that are attributes in old opsets and inputs in new ones
    bool param_ints(const ONode &n, const char *attr, size_t input, std::vector<int64_t> &out) const {
        auto a = n.attr.find(attr);
        if (a != n.attr.end()) { out = a->second.ints; return true; }
        return input < n.in.size() && !n.in[input].empty() && get_ints(n.in[input], out);
    }

    void absorb_constants() {
        for (size_t i = 0; i < g.nodes.size();) {
            ONode &n = g.nodes[i];
            if (n.op != "Constant" || n.out.empty()) { i++; continue; }
            OTensor t;
            auto a = n.attr.find("value");
            if (a != n.attr.end() && a->second.t) t = *a->second.t;
            else if ((a = n.attr.find("value_int")) != n.attr.end()) { t.dtype = 7; t.i = {a->second.i}; }
            else if ((a = n.attr.find("value_ints")) != n.attr.end()) { t.dtype = 7; t.i = a->second.ints; t.dims = {(int64_t)t.i.size()}; }
            else if ((a = n.attr.find("value_float")) != n.attr.end()) { t.dtype = 1; t.f = {a->second.f}; }
            else fail("Constant node without a value");
            t.name = n.out[0];
            g.init[t.name] = std::move(t);
            g.nodes.erase(g.nodes.begin() + (long)i);
        }
    }

    void bypass_noops() {
        for (size_t i = 0; i < g.nodes.size();) {
            ONode &n = g.nodes[i];
            const bool noop = n.op == "Identity" || n.op == "Dropout" || n.op == "Cast";
            if (!noop || n.in.empty() || n.out.empty()) { i++; continue; }
            if (const OTensor *t = init(n.in[0])) {  // of a constant: an alias (Cast: converted)
                OTensor c = *t;
                if (n.op == "Cast") {
                    const int64_t to = n.attr_i("to", 1);
                    if (to == 1 && c.dtype == 7) { c.f.assign(c.i.begin(), c.i.end()); c.i.clear(); c.dtype = 1; }
                    else if (to == 7 && c.dtype == 1) { c.i.clear(); for (float f : c.f) c.i.push_back((int64_t)f); c.f.clear(); c.dtype = 7; }
                    else if (!((to == 1 && c.dtype == 1) || (to == 7 && c.dtype == 7))) fail("Cast of a constant to data type " + std::to_string(to));
                }
                c.name = n.out[0];
                if (is_output(c.name)) fail("graph output '" + c.name + "' is a constant");
                g.init[c.name] = std::move(c);
                g.nodes.erase(g.nodes.begin() + (long)i);
                continue;
            }
            if (n.op == "Cast") {
                const int64_t to = n.attr_i("to", 1);
                if (to == 7 || to == 6) { i++; continue; }  // integer arithmetic on shapes: folded below
                if (to != 1) fail("Cast of an activation to data type " + std::to_string(to) + " (only FLOAT)");
            }
            bypass(i);
        }
    }

    static int64_t prod(const std::vector<int64_t> &d, size_t lo, size_t hi) {
        int64_t p = 1;
        for (size_t k = lo; k < hi; k++) {
            if (d[k] < 0) return -1;
            p *= d[k];
        }
        return p;
    }

    // one pass over the (topologically ordered) nodes: shapes of data tensors and values of integer tensors, where known
    void infer() {
        shape.clear();
        ints.clear();
        int_scalar.clear();
        for (auto &in : g.inputs)
            if (!g.init.count(in.first)) shape[in.first] = in.second;
        for (auto &n : g.nodes) {
            if (n.out.empty()) continue;
            const std::string &o = n.out[0];
            const std::vector<int64_t> *a = n.in.empty() ? nullptr : get_shape(n.in[0]);
            std::vector<int64_t> v, w;
            if (n.op == "Shape") {
                if (a) {
                    v = *a;
                    for (auto &d : v) if (d < 0) d = BATCH;
                    ints[o] = v;
                }
            } else if (n.op == "Gather") {
                const int64_t axis = n.attr_i("axis", 0);
                if (get_ints(n.in[0], v) && get_ints(n.in[1], w) && axis == 0) {  // integer tensor, constant indices
                    std::vector<int64_t> r;
                    for (int64_t k : w) {
                        if (k < 0) k += (int64_t)v.size();
                        if (k < 0 || k >= (int64_t)v.size()) fail("Gather index out of range in shape arithmetic");
                        r.push_back(v[(size_t)k]);
                    }
                    ints[o] = r;
                    if (is_scalar(n.in[1])) int_scalar.insert(o);
                } else if (a && get_ints(n.in[1], w)) {  // data tensor
                    std::vector<int64_t> r = *a;
                    const int64_t ax = axis < 0 ? axis + (int64_t)r.size() : axis;
                    if (ax >= 0 && ax < (int64_t)r.size()) {
                        if (is_scalar(n.in[1])) r.erase(r.begin() + ax);
                        else r[(size_t)ax] = (int64_t)w.size();
                        shape[o] = r;
                    }
                }
            } else if (n.op == "Unsqueeze" || n.op == "Squeeze") {
                if (get_ints(n.in[0], v)) {  // integer tensors stay lists here (rank 0 <-> rank 1)
                    ints[o] = v;
                    if (n.op == "Squeeze" && v.size() == 1) int_scalar.insert(o);
                } else if (a && param_ints(n, "axes", 1, w)) {
                    std::vector<int64_t> r = *a;
                    if (n.op == "Unsqueeze") {
                        std::vector<int64_t> ax = w;
                        for (auto &x : ax) if (x < 0) x += (int64_t)(r.size() + ax.size());
                        std::sort(ax.begin(), ax.end());
                        for (int64_t x : ax) if (x >= 0 && x <= (int64_t)r.size()) r.insert(r.begin() + x, 1);
                    } else {
                        std::vector<int64_t> ax = w;
                        for (auto &x : ax) if (x < 0) x += (int64_t)r.size();
                        std::sort(ax.rbegin(), ax.rend());
                        for (int64_t x : ax) if (x >= 0 && x < (int64_t)r.size()) r.erase(r.begin() + x);
                    }
                    shape[o] = r;
                }
            } else if (n.op == "Cast") {
                if (get_ints(n.in[0], v)) { ints[o] = v; if (is_scalar(n.in[0])) int_scalar.insert(o); }
            } else if (n.op == "Concat") {
                bool all_ints = !n.in.empty();
                std::vector<int64_t> r;
                for (auto &in : n.in) {
                    if (!get_ints(in, v)) { all_ints = false; break; }
                    r.insert(r.end(), v.begin(), v.end());
                }
                if (all_ints) ints[o] = r;
                else if (a) {
                    std::vector<int64_t> s = *a;
                    int64_t axis = n.attr_i("axis", 0);
                    if (axis < 0) axis += (int64_t)s.size();
                    bool ok = axis >= 0 && axis < (int64_t)s.size();
                    for (size_t k = 1; k < n.in.size() && ok; k++) {
                        const std::vector<int64_t> *b = get_shape(n.in[k]);
                        if (!b || b->size() != s.size()) { ok = false; break; }
                        s[(size_t)axis] = s[(size_t)axis] < 0 || (*b)[(size_t)axis] < 0 ? -1 : s[(size_t)axis] + (*b)[(size_t)axis];
                    }
                    if (ok) shape[o] = s;
                }
            } else if (n.op == "Slice") {
                std::vector<int64_t> starts, ends, axes, steps;
                bool have = param_ints(n, "starts", 1, starts) && param_ints(n, "ends", 2, ends);
                const bool have_axes = param_ints(n, "axes", 3, axes);
                param_ints(n, "steps", 4, steps);
                // a file controls these lengths: a node whose parameter vectors disagree is left unfolded (the matcher then
                // rejects the graph with a message) instead of being indexed
                if (have && (starts.empty() || ends.size() != starts.size() || (have_axes && axes.size() != starts.size()) ||
                             (!steps.empty() && steps.size() != starts.size())))
                    have = false;
                if (have && get_ints(n.in[0], v)) {  // a slice of an integer tensor (1-D)
                    if (starts.size() == 1 && (!have_axes || (axes.size() == 1 && axes[0] == 0)) && (steps.empty() || steps[0] == 1)) {
                        const int64_t len = (int64_t)v.size();
                        int64_t lo = starts[0] < 0 ? starts[0] + len : starts[0], hi = ends[0] < 0 ? ends[0] + len : ends[0];
                        lo = std::max<int64_t>(0, std::min(lo, len));
                        hi = std::max<int64_t>(0, std::min(hi, len));
                        ints[o] = std::vector<int64_t>(v.begin() + lo, v.begin() + std::max(lo, hi));
                    }
                } else if (have && a) {
                    std::vector<int64_t> r = *a;
                    bool ok = true;
                    for (size_t k = 0; k < starts.size() && ok; k++) {
                        int64_t ax = have_axes ? axes[k] : (int64_t)k;
                        if (ax < 0) ax += (int64_t)r.size();
                        const int64_t step = k < steps.size() ? steps[k] : 1;
                        if (ax < 0 || ax >= (int64_t)r.size() || step != 1) { ok = false; break; }
                        const int64_t len = r[(size_t)ax];
                        if (len < 0) continue;  // a slice of the batch axis stays symbolic
                        int64_t lo = starts[k] < 0 ? starts[k] + len : starts[k], hi = ends[k] < 0 ? ends[k] + len : ends[k];
                        lo = std::max<int64_t>(0, std::min(lo, len));
                        hi = std::max<int64_t>(0, std::min(hi, len));
                        r[(size_t)ax] = std::max<int64_t>(0, hi - lo);
                    }
                    if (ok) shape[o] = r;
                }
            } else if (n.op == "Conv") {
                const OTensor *wt = n.in.size() > 1 ? init(n.in[1]) : nullptr;
                if (a && a->size() == 4 && wt && wt->dims.size() == 4) shape[o] = {(*a)[0], wt->dims[0], (*a)[2], (*a)[3]};
            } else if (n.op == "Relu" || n.op == "BatchNormalization" || n.op == "Div" || n.op == "Mul" || n.op == "Sub" ||
                       n.op == "Sigmoid" || n.op == "Tanh" || n.op == "Softmax" || n.op == "Identity" || n.op == "Dropout") {
                if (a) shape[o] = *a;
            } else if (n.op == "Add") {
                const std::vector<int64_t> *b = n.in.size() > 1 ? get_shape(n.in[1]) : nullptr;
                if (a && (!b || a->size() >= b->size())) shape[o] = *a;
                else if (b) shape[o] = *b;
            } else if (n.op == "Flatten") {
                if (a) {
                    int64_t axis = n.attr_i("axis", 1);
                    if (axis < 0) axis += (int64_t)a->size();
                    if (axis >= 0 && axis <= (int64_t)a->size()) shape[o] = {prod(*a, 0, (size_t)axis), prod(*a, (size_t)axis, a->size())};
                }
            } else if (n.op == "Gemm") {
                const OTensor *wt = n.in.size() > 1 ? init(n.in[1]) : nullptr;
                if (a && a->size() == 2 && wt && wt->dims.size() == 2) shape[o] = {(*a)[0], wt->dims[n.attr_i("transB", 0) ? 0 : 1]};
            } else if (n.op == "MatMul") {
                const OTensor *wt = n.in.size() > 1 ? init(n.in[1]) : nullptr;
                const std::vector<int64_t> *b = n.in.size() > 1 ? get_shape(n.in[1]) : nullptr;
                if (a && !a->empty() && wt && wt->dims.size() == 2) { v = *a; v.back() = wt->dims[1]; shape[o] = v; }
                else if (a && b && a->size() >= 2 && b->size() >= 2) { v = *a; v.back() = b->back(); shape[o] = v; }
            } else if (n.op == "Transpose") {
                std::vector<int64_t> perm = n.attr_ints("perm");
                if (a && perm.size() == a->size()) {
                    for (int64_t pz : perm) v.push_back(pz >= 0 && pz < (int64_t)a->size() ? (*a)[(size_t)pz] : -1);
                    shape[o] = v;
                }
            } else if (n.op == "Reshape") {
                if (n.in.size() > 1 && get_ints(n.in[1], w)) {
                    std::vector<int64_t> r;
                    int64_t known = 1;
                    int infer_at = -1;
                    for (size_t k = 0; k < w.size(); k++) {
                        int64_t d = w[k];
                        if (d == 0) d = a && k < a->size() ? (*a)[k] : -1;
                        else if (d == BATCH) d = -1;
                        else if (d == -1) { infer_at = (int)k; d = -2; }
                        r.push_back(d);
                        if (d > 0) known *= d;
                    }
                    if (infer_at >= 0) {
                        const bool batch_in = a && std::count(a->begin(), a->end(), (int64_t)-1) == 1;
                        const bool batch_out = std::count(r.begin(), r.end(), (int64_t)-1) == 1;
                        const int64_t total = a ? prod(*a, 0, a->size()) : -1;
                        if (a && total > 0 && !batch_out) r[(size_t)infer_at] = total / known;
                        else if (a && batch_in && batch_out) {  // both carry the batch: the rest must match
                            int64_t rest = 1;
                            for (int64_t d : *a) if (d > 0) rest *= d;
                            r[(size_t)infer_at] = rest / known;
                        } else r[(size_t)infer_at] = -1;  // the inferred axis is the batch (or unknown)
                    }
                    shape[o] = r;
                }
            }
        }
    }

    // Reshape that flattens everything behind the batch axis -> Flatten(axis = 1); a reshape (or flatten) of a tensor
    // that has that shape already -> nothing.  Returns true when the graph changed.
    bool rewrite_reshapes() {
        for (size_t i = 0; i < g.nodes.size(); i++) {
            ONode &n = g.nodes[i];
            if (n.op != "Reshape" && n.op != "Flatten") continue;
            const std::vector<int64_t> *a = get_shape(n.in[0]);
            if (n.op == "Flatten") {
                if (a && a->size() == 2 && n.attr_i("axis", 1) == 1) { bypass(i); return true; }
                continue;
            }
            std::vector<int64_t> t;
            if (n.in.size() < 2 || !get_ints(n.in[1], t)) continue;
            const std::vector<int64_t> *o = get_shape(n.out[0]);
            if (a && o && *a == *o) { bypass(i); return true; }  // e.g. DensePolicyHead's .view(-1, *policy_shape), post_act.py:51
            if (t.size() != 2) continue;
            const bool batch_first = t[0] == BATCH || t[0] == 0 || (t[0] == -1 && t[1] > 0);
            if (!batch_first || (a && (a->size() < 2 || (*a)[0] >= 0))) continue;
            if (t[1] != -1 && a) {  // an explicit size must be the product of the flattened axes
                const int64_t rest = prod(*a, 1, a->size());
                if (rest > 0 && rest != t[1]) continue;
            }
            n.op = "Flatten";
            n.in.resize(1);
            n.attr.clear();
            OAttr axis;
            axis.i = 1;
            n.attr["axis"] = axis;
            return true;
        }
        return false;
    }

    // MatMul(x, W [in, out] constant) -> Add(., b [out] constant)  =>  Gemm(x, W, b), transB = 0
    bool rewrite_matmul_add() {
        for (size_t i = 0; i < g.nodes.size(); i++) {
            ONode &mm = g.nodes[i];
            if (mm.op != "MatMul" || mm.in.size() != 2) continue;
            const OTensor *w = init(mm.in[1]);
            if (!w || w->dtype != 1 || w->dims.size() != 2) continue;
            int users = 0, add_at = -1;
            for (size_t k = 0; k < g.nodes.size(); k++)
                for (auto &in : g.nodes[k].in)
                    if (in == mm.out[0]) { users++; if (g.nodes[k].op == "Add") add_at = (int)k; }
            if (users != 1 || add_at < 0 || is_output(mm.out[0])) continue;
            ONode &add = g.nodes[(size_t)add_at];
            const std::string &other = add.in[0] == mm.out[0] ? add.in[1] : add.in[0];
            const OTensor *b = init(other);
            if (!b || b->dtype != 1 || (int64_t)b->f.size() != w->dims[1]) continue;
            ONode gemm;
            gemm.op = "Gemm";
            gemm.in.push_back(mm.in[0]);
            gemm.in.push_back(mm.in[1]);
            gemm.in.push_back(other);
            gemm.out = add.out;
            g.nodes[(size_t)add_at] = gemm;  // (transB absent = 0; alpha = beta = 1)
            g.nodes.erase(g.nodes.begin() + (long)i);
            return true;
        }
        return false;
    }

    // The older output form (value [B], wdl [B, 3], policy) that check_graph_shapes and decode_output still accept
    // (rust/kz-core/src/network/common.rs:42-49, 186-190) -> the (scalars [B, 5], policy) form: the rows of the scalar
    // head's last Linear that value and wdl select become rows 0..3 of a [5, hidden] Linear whose fifth row yields NaN —
    // decode_output's `moves_left = NaN` for such graphs (common.rs:44).  Outputs are taken by position, like the
    // reference does.
    void legacy_outputs() {
        if (g.outputs.size() != 3) return;
        auto producer_of = [&](const std::string &name) -> ONode * {
            for (auto &n : g.nodes)
                for (auto &o : n.out) if (o == name) return &n;
            return nullptr;
        };
        // rows of a Gemm's output a name selects: walks through shape-only nodes to Gather / Slice on axis 1
        struct Pick { ONode *gemm; std::vector<int64_t> rows; };
        auto pick = [&](std::string name, size_t want) -> Pick {
            std::vector<int64_t> rows;
            bool selected = false;
            for (int hop = 0; hop < 8; hop++) {
                ONode *n = producer_of(name);
                if (!n) break;
                if (n->op == "Squeeze" || n->op == "Flatten" || n->op == "Reshape" || n->op == "Unsqueeze") { name = n->in[0]; continue; }
                if (n->op == "Gather" && !selected) {
                    std::vector<int64_t> idx;
                    if (n->attr_i("axis", 0) != 1 || !get_ints(n->in[1], idx)) break;
                    rows = idx;
                    selected = true;
                    name = n->in[0];
                    continue;
                }
                if (n->op == "Slice" && !selected) {
                    std::vector<int64_t> starts, ends, axes, steps;
                    if (!param_ints(*n, "starts", 1, starts) || !param_ints(*n, "ends", 2, ends) || starts.size() != 1 || ends.size() != 1) break;
                    if (param_ints(*n, "axes", 3, axes) && (axes.size() != 1 || axes[0] != 1)) break;
                    if (param_ints(*n, "steps", 4, steps) && (steps.size() != 1 || steps[0] != 1)) break;
                    if (axes.empty()) break;  // (without axes a one-element slice would cut the batch axis)
                    for (int64_t r = starts[0]; r < std::min<int64_t>(ends[0], starts[0] + 64); r++) rows.push_back(r);
                    selected = true;
                    name = n->in[0];
                    continue;
                }
                if (n->op == "Gemm") {
                    const OTensor *w = n->in.size() > 1 ? init(n->in[1]) : nullptr;
                    if (!w || w->dims.size() != 2) break;
                    const int64_t outs = w->dims[n->attr_i("transB", 0) ? 0 : 1];
                    if (!selected) for (int64_t r = 0; r < outs; r++) rows.push_back(r);
                    for (auto &r : rows) if (r < 0) r += outs;
                    for (int64_t r : rows) if (r < 0 || r >= outs) fail("legacy outputs: selected row out of range");
                    if (rows.size() != want) break;
                    return Pick{n, rows};
                }
                break;
            }
            fail("legacy (value, wdl, policy) outputs: '" + name + "' is not " + std::to_string(want) + " row(s) of the scalar head's last Linear");
        };
        const Pick value = pick(g.outputs[0], 1), wdl = pick(g.outputs[1], 3);
        if (value.gemm->in[0] != wdl.gemm->in[0]) fail("legacy outputs: value and wdl do not read the same hidden layer");
        // [5, hidden] rows: value, wdl x 3, moves_left = NaN
        Linear rows[2];
        int hidden = -1;
        OTensor w5, b5;
        w5.dtype = b5.dtype = 1;
        for (const Pick *pk : {&value, &wdl}) {
            const OTensor *w = init(pk->gemm->in[1]), *b = pk->gemm->in.size() > 2 ? init(pk->gemm->in[2]) : nullptr;
            if (!w || !b || w->dtype != 1 || b->dtype != 1) fail("legacy outputs: Gemm without constant weights and bias");
            const bool tb = pk->gemm->attr_i("transB", 0) == 1;
            const int64_t outs = w->dims[tb ? 0 : 1], in = w->dims[tb ? 1 : 0];
            if (hidden < 0) hidden = (int)in;
            if (in != hidden || (int64_t)b->f.size() != outs) fail("legacy outputs: Linear shapes");
            for (int64_t r : pk->rows) {
                for (int64_t i = 0; i < in; i++) w5.f.push_back(tb ? w->f[(size_t)(r * in + i)] : w->f[(size_t)(i * outs + r)]);
                b5.f.push_back(b->f[(size_t)r]);
            }
        }
        w5.f.insert(w5.f.end(), (size_t)hidden, 0.0f);
        b5.f.push_back(std::nanf(""));
        w5.dims = {5, hidden};
        b5.dims = {5};
        w5.name = "kz.legacy.scalars.weight";
        b5.name = "kz.legacy.scalars.bias";
        ONode gemm;
        gemm.op = "Gemm";
        gemm.in.push_back(value.gemm->in[0]);
        gemm.in.push_back(w5.name);
        gemm.in.push_back(b5.name);
        gemm.out.push_back("scalars");
        OAttr one;
        one.i = 1;
        gemm.attr["transB"] = one;
        g.init[w5.name] = std::move(w5);
        g.init[b5.name] = std::move(b5);
        g.nodes.push_back(gemm);
        const std::string policy = g.outputs[2];
        if (policy != "policy") {
            ONode *pp = producer_of(policy);
            if (!pp) fail("legacy outputs: nothing produces the policy");
            for (auto &o : pp->out) if (o == policy) o = "policy";
            replace_uses(policy, "policy");
        }
        g.outputs = {"scalars", "policy"};
    }

    void rebuild_index() {
        g.producer.clear();
        g.consumers.clear();
        for (size_t i = 0; i < g.nodes.size(); i++) {
            for (auto &o : g.nodes[i].out) g.producer[o] = (int)i;
            for (auto &in : g.nodes[i].in) g.consumers[in].push_back((int)i);
        }
    }

    void run() {
        absorb_constants();
        bypass_noops();
        for (int guard = 0; guard < 10000; guard++) {
            infer();
            if (!rewrite_reshapes() && !rewrite_matmul_add()) break;
        }
        legacy_outputs();
        // the folded integer tensors are constants from here on (e.g. the shape ConstantOfShape reads)
        for (auto &kv : ints) {
            if (g.init.count(kv.first)) continue;
            bool symbolic = false;
            for (int64_t v : kv.second) symbolic |= v == BATCH;
            if (symbolic) continue;
            OTensor t;
            t.name = kv.first;
            t.dtype = 7;
            t.i = kv.second;
            if (!int_scalar.count(kv.first)) t.dims = {(int64_t)t.i.size()};
            g.init[t.name] = std::move(t);
        }
        rebuild_index();
    }
};

#include "kz_onnx_match.hpp"

}  // namespace

void finalize_model(Model &m);  // kz_model.cpp

bool looks_like_onnx(const void *blob, size_t len) {
    // ModelProto starts with field 1 (ir_version, varint): key byte 0x08
    return len > 16 && static_cast<const uint8_t *>(blob)[0] == 0x08 && memcmp(blob, "KZMODEL1", 8) != 0;
}

Model *parse_onnx(const void *blob, size_t len, int n_scalar, std::string &err) {
    try {
        OGraph g = read_model(blob, len);
        Normalizer{g}.run();
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
        if (g.outputs.size() != 2 || !has_scalars || !has_policy)
            fail("outputs must be ('scalars', 'policy'), or three outputs (value, wdl, policy) in that order (network/common.rs:36-49)");
        const int hw = m->h * m->w;

        auto flat_in = M.consumers("input", "Flatten");
        if (flat_in.size() == 1 && M.consumers("input", "Conv").empty() && M.consumers("input", "Transpose").empty()) {
            // ---- DenseNetwork (simple.py:7-33): no tower, no heads ----
            M.dense_network(*m, *flat_in[0]);
            finalize_model(*m);
            return m.release();
        }
        std::string t;  // tower output seen by the heads
        int C = 0;
        auto to_tokens = M.consumers("input", "Transpose");
        if (to_tokens.size() == 1 && M.consumers("input", "Conv").empty()) {
            // ---- AttentionTower (attention.py:8-45) ----
            t = M.attention_tower(*m, *to_tokens[0]);
            C = m->channels;
        } else {
        // ---- the tower (post_act.py:201-228: ResTower, ResBlock) ----
        auto stems = M.consumers("input", "Conv");
        if (stems.size() != 1 || M.consumers("input").size() != 1) fail("'input' must feed exactly one Conv");
        m->tower.push_back(M.conv(*stems[0], 3));
        if (m->tower[0].cin != m->c_in) fail("stem input channels");
        C = m->channels = m->tower[0].cout;
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
        t = cur;
        auto final_bn = M.consumers(cur, "BatchNormalization");
        if (final_bn.size() == 1 && M.consumers(cur).size() == 1) {
            M.bn_affine(*final_bn[0], C, m->final_scale, m->final_shift);  // post_act.py:207
            t = final_bn[0]->out[0];
        } else {
            m->final_scale.assign(C, 1.0f);
            m->final_shift.assign(C, 0.0f);
        }
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
        } else if (p->op == "Concat" && p->attr_i("axis", 0) == 1 && p->in.size() == 2 && M.producer(p->in[0], "Gemm") &&
                   M.producer(p->in[1], "Flatten")) {
            // ArimaaPolicyHead (post_act.py:144-173): concat([scalar(common), flatten(bulk(common), 1)])
            //   scalar = Conv1x1(C->hc), ReLU, Flatten, Linear(hc*hw -> hs), ReLU, Linear(hs -> 1 + 6);  bulk = Conv1x1(C->C), ReLU, Conv1x1(C->4)
            m->policy_kind = POLICY_ARIMAA;
            const ONode &g2 = M.expect_producer(p->in[0], "Gemm");
            const ONode &r = M.expect_producer(g2.in[0], "Relu");
            const ONode &g1 = M.expect_producer(r.in[0], "Gemm");
            const ONode &fl = M.expect_producer(g1.in[0], "Flatten");
            if (!M.relu_conv1x1(fl.in[0], t, m->pa_conv)) fail("arimaa head: expected Conv1x1 + ReLU on the tower output");
            m->pa_fc0 = M.gemm(g1);
            m->pa_fc1 = M.gemm(g2);
            conv_stack(p->in[1], m->p_conv0, m->p_conv1);
            m->policy_conv_channels = m->p_conv1.cout;
            m->arimaa_hidden_channels = m->pa_conv.cout;
            m->arimaa_hidden_size = m->pa_fc0.out;
            if (m->pa_fc0.in != m->pa_conv.cout * hw || m->pa_fc1.in != m->pa_fc0.out || m->pa_fc1.out != 7 || m->p_conv1.cout != 4 ||
                m->p_conv0.cout != C)
                fail("arimaa head shapes");
            m->policy_len = 7 + 4 * hw;
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
    } catch (const std::exception &e) {  // (length_error / bad_alloc from a file-controlled size must not cross the C ABI)
        err = std::string("ONNX: ") + e.what();
        return nullptr;
    }
}

}  // namespace kz
