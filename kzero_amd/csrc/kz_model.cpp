// kz_model.cpp — KZMODEL1 container parse (format: kzero_amd/model_file.py) and Conv+BN folding.
#include "kz_model.hpp"

#include <cmath>
#include <cstring>
#include <memory>

namespace kz {
namespace {

struct RawTensor {
    int dtype = 0;
    std::vector<uint64_t> dims;
    const uint8_t *data = nullptr;
    uint64_t count = 0;
};

struct Container {
    std::map<std::string, int64_t> ints;
    std::map<std::string, double> floats;
    std::map<std::string, std::string> strings;
    std::map<std::string, RawTensor> tensors;
};

struct Reader {
    const uint8_t *p;
    size_t left;
    bool bad = false;
    template <class T>
    T get() {
        T v{};
        if (left < sizeof(T)) {
            bad = true;
            return v;
        }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        left -= sizeof(T);
        return v;
    }
    std::string str(size_t n) {
        if (left < n) {
            bad = true;
            return {};
        }
        std::string s(reinterpret_cast<const char *>(p), n);
        p += n;
        left -= n;
        return s;
    }
};

bool parse_container(const void *blob, size_t len, Container &c, std::string &err) {
    Reader r{static_cast<const uint8_t *>(blob), len};
    if (r.str(8) != "KZMODEL1") {
        err = "not a KZMODEL1 container";
        return false;
    }
    uint32_t n_meta = r.get<uint32_t>();
    for (uint32_t i = 0; i < n_meta && !r.bad; i++) {
        std::string key = r.str(r.get<uint16_t>());
        uint8_t kind = r.get<uint8_t>();
        if (kind == 0) c.ints[key] = r.get<int64_t>();
        else if (kind == 1) c.floats[key] = r.get<double>();
        else if (kind == 2) c.strings[key] = r.str(r.get<uint32_t>());
        else r.bad = true;
    }
    uint32_t n_tensors = r.get<uint32_t>();
    std::vector<std::pair<std::string, std::pair<uint64_t, uint64_t>>> spans;
    for (uint32_t i = 0; i < n_tensors && !r.bad; i++) {
        std::string name = r.str(r.get<uint16_t>());
        RawTensor t;
        t.dtype = r.get<uint8_t>();
        uint32_t ndim = r.get<uint32_t>();
        if (ndim > 8) {
            r.bad = true;
            break;
        }
        t.count = 1;
        for (uint32_t d = 0; d < ndim; d++) {
            t.dims.push_back(r.get<uint64_t>());
            if (__builtin_mul_overflow(t.count, t.dims.back(), &t.count)) r.bad = true;  // (a hostile file)
        }
        uint64_t off = r.get<uint64_t>(), nbytes = r.get<uint64_t>();
        spans.push_back({name, {off, nbytes}});
        c.tensors[name] = t;
    }
    uint64_t data_len = r.get<uint64_t>();
    if (r.bad || r.left < data_len) {
        err = "truncated KZMODEL1 container";
        return false;
    }
    for (auto &s : spans) {
        RawTensor &t = c.tensors[s.first];
        uint64_t esz = t.dtype == 0 ? 4 : 8, end = 0, want = 0;
        if (t.dtype > 1 || __builtin_add_overflow(s.second.first, s.second.second, &end) || end > data_len ||
            __builtin_mul_overflow(t.count, esz, &want) || s.second.second != want) {
            err = "tensor '" + s.first + "' out of range";
            return false;
        }
        t.data = r.p + s.second.first;
    }
    return true;
}

struct Loader {
    const Container &c;
    std::string &err;
    bool ok = true;
    int64_t params = 0;

    int64_t geti(const char *key, int64_t dflt, bool required = false) {
        auto it = c.ints.find(key);
        if (it != c.ints.end()) return it->second;
        auto jt = c.floats.find(key);
        if (jt != c.floats.end()) return (int64_t)jt->second;
        if (required && ok) {
            ok = false;
            err = std::string("missing descriptor key '") + key + "'";
        }
        return dflt;
    }

    // (a failure unwinds to parse_model: nothing downstream ever sees a tensor of the wrong size, and a descriptor that
    //  asks for an absurd size never turns into an allocation)
    struct Failed {};
    // a descriptor integer that must lie in [lo, hi] (sizes are products of these: bounded before anything is allocated
    // or looped over)
    int geti_in(const char *key, int64_t dflt, int64_t lo, int64_t hi, bool required = false) {
        const int64_t v = geti(key, dflt, required);
        if (v < lo || v > hi) {
            if (ok) {
                ok = false;
                err = std::string("descriptor key '") + key + "' out of range";
            }
            return (int)lo;
        }
        return (int)v;
    }

    std::vector<float> f32(const std::string &name, uint64_t expect) {
        auto it = c.tensors.find(name);
        if (it == c.tensors.end() || it->second.dtype != 0 || it->second.count != expect) {
            if (ok) {
                ok = false;
                err = "missing or mis-shaped tensor '" + name + "' (want " + std::to_string(expect) + " f32 values)";
            }
            throw Failed{};
        }
        std::vector<float> v(expect);
        memcpy(v.data(), it->second.data, expect * 4);
        params += (int64_t)expect;
        return v;
    }

    Conv conv(const std::string &prefix, int cout, int cin, int k) {
        Conv cv;
        cv.cout = cout;
        cv.cin = cin;
        cv.k = k;
        cv.w = f32(prefix + ".weight", (uint64_t)cout * cin * k * k);
        cv.b = f32(prefix + ".bias", (uint64_t)cout);
        return cv;
    }

    Linear linear(const std::string &prefix, int out, int in) {
        Linear l;
        l.out = out;
        l.in = in;
        l.w = f32(prefix + ".weight", (uint64_t)out * in);
        l.b = f32(prefix + ".bias", (uint64_t)out);
        return l;
    }

    // nn.BatchNorm2d in eval mode as y = s*x + t with s = gamma / sqrt(var + eps), t = beta - s*mean
    void bn_affine(const std::string &prefix, int ch, bool affine, float eps, std::vector<float> &s,
                   std::vector<float> &t) {
        std::vector<float> gamma(ch, 1.0f), beta(ch, 0.0f);
        if (affine) {
            gamma = f32(prefix + ".weight", ch);
            beta = f32(prefix + ".bias", ch);
        }
        std::vector<float> mean = f32(prefix + ".running_mean", ch);
        std::vector<float> var = f32(prefix + ".running_var", ch);
        s.resize(ch);
        t.resize(ch);
        for (int i = 0; i < ch; i++) {
            // computed in double then rounded once, so the fold itself adds no error beyond one f32 rounding
            double sd = (double)gamma[i] / std::sqrt((double)var[i] + (double)eps);
            s[i] = (float)sd;
            t[i] = (float)((double)beta[i] - sd * (double)mean[i]);
        }
    }

    // conv followed by BN: W' = s*W, b' = s*b + t  (what optimize_graph does to Conv+BN pairs)
    void fold(Conv &cv, const std::vector<float> &s, const std::vector<float> &t) {
        size_t per = (size_t)cv.cin * cv.k * cv.k;
        for (int o = 0; o < cv.cout; o++) {
            for (size_t i = 0; i < per; i++) cv.w[o * per + i] *= s[o];
            cv.b[o] = s[o] * cv.b[o] + t[o];
        }
    }
};

}  // namespace

// parameter count and direct-convolution FLOPs (2 per MAC, heads included: BASELINE.md §2) from the folded model;
// used for models that do not come from a KZMODEL1 container
// multiply-adds of AttentionTower.forward for one board: expand, and per layer the four Linear layers and the two batched
// matrix products of the attention (attention.py:106-129)
double attention_tower_macs(const Model &m) {
    const double n = (double)m.h * m.w, D = m.channels, H = m.att_heads, dk = m.att_dk, dv = m.att_dv, dff = m.att_dff;
    return n * m.c_in * D + m.depth * (n * D * H * (2 * dk + dv) + H * n * n * (dk + dv) + n * H * dv * D + 2 * n * D * dff);
}

void finalize_model(Model &m) {
    const int C = m.channels, hw = m.h * m.w;
    double macs = 0;
    int64_t params = 0;
    auto conv = [&](const Conv &c, int pixels) {
        if (!c.cout) return;
        macs += (double)pixels * c.cout * c.cin * c.k * c.k;
        params += (int64_t)c.w.size() + (int64_t)c.b.size();
    };
    auto lin = [&](const Linear &l) {
        if (!l.out) return;
        macs += (double)l.out * l.in;
        params += (int64_t)l.w.size() + (int64_t)l.b.size();
    };
    for (auto &c : m.tower) conv(c, hw);
    if (m.tower_kind == TOWER_DENSE_NET) {
        lin(m.dn_in);
        for (auto &b : m.dn_blocks) {
            lin(b.la);
            lin(b.lb);
            params += 4 * C;  // (the two BatchNorm1d as affines; the container counts their four vectors each: parse_model sets it)
        }
        lin(m.dn_out);
        params += 2 * C;
        m.flops_per_eval = 2.0 * macs;
        if (m.param_count == 0) m.param_count = params;
        return;
    }
    if (m.tower_kind == TOWER_ATTENTION) {
        macs += attention_tower_macs(m);
        params += (int64_t)m.att_expand.size() + (int64_t)m.att_embedding.size();
        for (auto &l : m.att_layers) params += (int64_t)(l.qkv.size() + l.out.size() + l.ff0.size() + l.ff1.size());
    } else {
        params += 2 * C;  // final BN as an affine
    }
    conv(m.sh_conv, hw);
    lin(m.sh_fc0);
    lin(m.sh_fc1);
    switch (m.policy_kind) {
        case POLICY_ATAXX_CONV:
        case POLICY_CONV:
            conv(m.p_conv0, hw);
            conv(m.p_conv1, hw);
            conv(m.p_extra_conv, hw);
            lin(m.p_extra_fc);
            break;
        case POLICY_ATTENTION:
            conv(m.p_bulk, hw);
            conv(m.p_under, 8);
            macs += 64.0 * 88 * m.policy_query_channels;
            break;
        case POLICY_DENSE:
            conv(m.p_conv0, hw);
            lin(m.p_fc0);
            lin(m.p_fc1);
            break;
        case POLICY_ARIMAA:
            conv(m.p_conv0, hw);
            conv(m.p_conv1, hw);
            conv(m.pa_conv, hw);
            lin(m.pa_fc0);
            lin(m.pa_fc1);
            break;
        case POLICY_NONE: break;
    }
    m.flops_per_eval = 2.0 * macs;
    if (m.param_count == 0) m.param_count = params;
}

static Model *parse_model_checked(const void *blob, size_t len, std::string &err);

Model *parse_model(const void *blob, size_t len, std::string &err) {
    try {
        return parse_model_checked(blob, len, err);
    } catch (const Loader::Failed &) {
        if (err.empty()) err = "malformed KZMODEL1 container";
        return nullptr;
    } catch (const std::exception &e) {  // bad_alloc / length_error on a hostile file
        err = std::string("KZMODEL1 container: ") + e.what();
        return nullptr;
    }
}

static Model *parse_model_checked(const void *blob, size_t len, std::string &err) {
    Container c;
    if (!parse_container(blob, len, c, err)) return nullptr;
    Loader L{c, err};
    std::unique_ptr<Model> m(new Model());

    m->h = L.geti_in("board_h", 0, 1, 64, true);
    m->w = L.geti_in("board_w", 0, 1, 64, true);
    m->n_scalar = L.geti_in("input_scalar_channels", 0, 0, 4096, true);
    m->n_bool = L.geti_in("input_bool_channels", 0, 0, 4096, true);
    m->c_in = m->n_scalar + m->n_bool;
    m->depth = L.geti_in("tower_depth", 0, 0, 1024, true);
    m->channels = L.geti_in("tower_channels", 0, 1, 8192, true);
    m->policy_len = L.geti_in("policy_len", 0, 1, 1 << 24, true);
    bool final_affine = L.geti("tower_final_affine", 1) != 0;
    int sh_c = L.geti_in("scalar_hidden_channels", 4, 1, 4096);
    int sh_s = L.geti_in("scalar_hidden_size", 32, 1, 1 << 20);
    float eps = 1e-5f;
    if (c.floats.count("bn_eps")) eps = (float)c.floats.at("bn_eps");
    if (c.strings.count("game")) m->game = c.strings.at("game");
    if (!L.ok) return nullptr;
    if (m->c_in <= 0) {
        err = "bad architecture descriptor";
        return nullptr;
    }
    auto kind = c.strings.find("policy_kind");
    if (kind == c.strings.end()) {
        err = "missing descriptor key 'policy_kind'";
        return nullptr;
    }
    {
        auto tkd = c.strings.find("tower_kind");
        if (tkd != c.strings.end() && tkd->second == "dense_network") {
            // DenseNetwork with its DenseBlocks (python/lib/model/simple.py:7-52) under its state_dict names: seq.1 Linear; seq.{2+i}.seq.{0,2,3,5} per
            // block; seq.{2+depth} BatchNorm1d; seq.{4+depth} Linear
            if (kind->second != "none") {
                err = "a dense_network has policy_kind 'none'";
                return nullptr;
            }
            m->tower_kind = TOWER_DENSE_NET;
            m->policy_kind = POLICY_NONE;
            m->dn_res = L.geti("dn_res", 0) != 0;
            const int size = m->channels, in = m->c_in * m->h * m->w;
            if ((int64_t)m->policy_len + 5 > (1 << 24)) {
                err = "bad architecture descriptor";
                return nullptr;
            }
            std::vector<float> s, t;
            m->dn_in = L.linear("seq.1", size, in);
            for (int i = 0; i < m->depth; i++) {
                const std::string p = "seq." + std::to_string(2 + i) + ".seq.";
                Model::DnBlock b;
                L.bn_affine(p + "0", size, true, eps, b.sa, b.ta);
                b.la = L.linear(p + "2", size, size);
                L.bn_affine(p + "3", size, true, eps, b.sb, b.tb);
                b.lb = L.linear(p + "5", size, size);
                m->dn_blocks.push_back(std::move(b));
            }
            L.bn_affine("seq." + std::to_string(2 + m->depth), size, true, eps, m->dn_sf, m->dn_tf);
            m->dn_out = L.linear("seq." + std::to_string(4 + m->depth), 5 + m->policy_len, size);
            m->final_scale.assign(size, 1.0f);
            m->final_shift.assign(size, 0.0f);
            if (!L.ok) return nullptr;
            m->param_count = L.params;
            m->flops_per_eval = 2.0 * ((double)size * in + 2.0 * m->depth * size * size + (double)(5 + m->policy_len) * size);
            return m.release();
        }
    }
    if (kind->second == "ataxx_conv") m->policy_kind = POLICY_ATAXX_CONV;
    else if (kind->second == "conv") m->policy_kind = POLICY_CONV;
    else if (kind->second == "attention") m->policy_kind = POLICY_ATTENTION;
    else if (kind->second == "dense") m->policy_kind = POLICY_DENSE;
    else if (kind->second == "arimaa") m->policy_kind = POLICY_ARIMAA;
    else {
        err = "unknown policy_kind '" + kind->second + "'";
        return nullptr;
    }

    const int C = m->channels, hw = m->h * m->w;
    double macs = 0;

    auto tk = c.strings.find("tower_kind");
    if (tk != c.strings.end() && tk->second == "attention") {
        // AttentionTower (python/lib/model/attention.py:8-45) under its state_dict names
        m->tower_kind = TOWER_ATTENTION;
        const int H = m->att_heads = L.geti_in("att_heads", 0, 1, 256, true);
        const int dk = m->att_dk = L.geti_in("att_d_k", 0, 1, 4096, true);
        const int dv = m->att_dv = L.geti_in("att_d_v", 0, 1, 4096, true);
        const int dff = m->att_dff = L.geti_in("att_d_ff", 0, 1, 1 << 16, true);
        if (!L.ok) return nullptr;
        if ((int64_t)H * (2 * dk + dv) > (1 << 16) || m->depth < 1) {
            err = "bad attention tower descriptor";
            return nullptr;
        }
        m->att_alpha = c.floats.count("att_alpha") ? (float)c.floats.at("att_alpha") : (float)std::pow(2.0 * m->depth, 0.25);
        if (c.floats.count("ln_eps")) m->ln_eps = (float)c.floats.at("ln_eps");
        // LayerNorm(x * alpha + f(x)): the matrix-core tower folds 1 / alpha into the weights and uses eps / alpha^2
        // (att_tower16_pack_layer), which is the same function only for a finite alpha > 0 and a finite eps >= 0
        if (!(std::isfinite(m->att_alpha) && m->att_alpha > 0.0f) || !(std::isfinite(m->ln_eps) && m->ln_eps >= 0.0f)) {
            err = "attention tower: att_alpha must be finite and > 0, ln_eps finite and >= 0";
            return nullptr;
        }
        m->att_expand = L.f32("common.expand.weight", (uint64_t)C * m->c_in);
        m->att_embedding = L.f32("common.embedding", (uint64_t)hw * C);
        for (int i = 0; i < m->depth; i++) {
            const std::string p = "common.encoders." + std::to_string(i) + ".";
            AttLayer l;
            l.qkv = L.f32(p + "project_qkv.weight", (uint64_t)H * (2 * dk + dv) * C);
            l.out = L.f32(p + "project_out.weight", (uint64_t)C * H * dv);
            l.ff0 = L.f32(p + "ff.0.weight", (uint64_t)dff * C);
            l.ff1 = L.f32(p + "ff.2.weight", (uint64_t)C * dff);
            m->att_layers.push_back(std::move(l));
        }
        m->final_scale.assign(C, 1.0f);
        m->final_shift.assign(C, 0.0f);
        macs += attention_tower_macs(*m);
    } else if (tk != c.strings.end() && tk->second != "res") {
        err = "unknown tower_kind '" + tk->second + "'";
        return nullptr;
    } else {
    // ResTower: stem conv (no BN, no ReLU, post_act.py:205), ResBlocks (:214-228), final BN (:207)
    m->tower.push_back(L.conv("common.tower.0", C, m->c_in, 3));
    macs += (double)hw * C * m->c_in * 9;
    for (int i = 1; i <= m->depth; i++) {
        std::string p = "common.tower." + std::to_string(i) + ".seq.";
        std::vector<float> s, t;
        Conv a = L.conv(p + "0", C, C, 3);
        L.bn_affine(p + "1", C, true, eps, s, t);
        L.fold(a, s, t);
        Conv b = L.conv(p + "3", C, C, 3);
        L.bn_affine(p + "4", C, true, eps, s, t);
        L.fold(b, s, t);
        m->tower.push_back(std::move(a));
        m->tower.push_back(std::move(b));
        macs += 2.0 * hw * C * C * 9;
    }
    L.bn_affine("common.tower." + std::to_string(m->depth + 1), C, final_affine, eps, m->final_scale, m->final_shift);
    }

    // ScalarHead (post_act.py:10-23)
    m->sh_conv = L.conv("scalar_head.seq.0", sh_c, C, 1);
    m->sh_fc0 = L.linear("scalar_head.seq.3", sh_s, sh_c * hw);
    m->sh_fc1 = L.linear("scalar_head.seq.5", 5, sh_s);
    macs += (double)hw * sh_c * C + (double)sh_s * sh_c * hw + 5.0 * sh_s;

    switch (m->policy_kind) {
        case POLICY_ATAXX_CONV: {
            int pc = m->policy_conv_channels = L.geti_in("policy_conv_channels", 0, 1, 65536, true);
            if (L.ok && pc * hw + 1 != m->policy_len) {
                err = "ataxx_conv head: policy_len != policy_conv_channels*h*w + 1";
                return nullptr;
            }
            m->p_conv0 = L.conv("policy_head.seq.0", C, C, 1);
            m->p_conv1 = L.conv("policy_head.seq.2", pc, C, 1);
            macs += (double)hw * C * C + (double)hw * pc * C;
            break;
        }
        case POLICY_CONV: {
            int pc = m->policy_conv_channels = L.geti_in("policy_conv_channels", 0, 1, 65536, true);
            int ex = m->policy_extra_moves = L.geti_in("policy_extra_moves", 0, 0, 65536);
            if (L.ok && pc * hw + ex != m->policy_len) {
                err = "conv head: policy_len != policy_conv_channels*h*w + extra_moves";
                return nullptr;
            }
            m->p_conv0 = L.conv("policy_head.seq.0", C, C, 1);
            m->p_conv1 = L.conv("policy_head.seq.2", pc, C, 1);
            macs += (double)hw * C * C + (double)hw * pc * C;
            if (ex) {
                m->p_extra_conv = L.conv("policy_head.seq_extra.0", 1, C, 1);
                m->p_extra_fc = L.linear("policy_head.seq_extra.2", ex, hw);
                macs += (double)hw * C + (double)ex * hw;
            }
            break;
        }
        case POLICY_ATTENTION: {
            int Q = m->policy_query_channels = L.geti_in("policy_query_channels", 0, 1, 8192, true);
            if (L.ok && (m->h != 8 || m->w != 8)) {
                err = "attention head needs an 8x8 board";
                return nullptr;
            }
            if (!L.ok) return nullptr;
            m->p_bulk = L.conv("policy_head.conv_bulk", 2 * Q, C, 1);
            m->p_under = L.conv("policy_head.conv_under", 3 * Q, C, 1);
            auto it = c.tensors.find("policy_head.FLAT_TO_ATT");
            if (it == c.tensors.end() || it->second.dtype != 1 || (int)it->second.count != m->policy_len) {
                err = "missing or mis-shaped tensor 'policy_head.FLAT_TO_ATT'";
                return nullptr;
            }
            m->flat_to_att.resize(m->policy_len);
            for (int i = 0; i < m->policy_len; i++) {
                int64_t v;
                memcpy(&v, it->second.data + 8 * (size_t)i, 8);
                if (v < 0 || v >= 64 * 88) {
                    err = "FLAT_TO_ATT entry out of range";
                    return nullptr;
                }
                m->flat_to_att[i] = (int32_t)v;
            }
            macs += 64.0 * 2 * Q * C + 8.0 * 3 * Q * C + 64.0 * 88 * Q;
            break;
        }
        case POLICY_DENSE: {
            int hc = m->dense_hidden_channels = L.geti_in("policy_dense_hidden_channels", 0, 0, 8192);
            int hs = m->dense_hidden_size = L.geti_in("policy_dense_hidden_size", 0, 0, 1 << 20);
            if (!L.ok) return nullptr;
            int idx = 0, ch = C;
            if (hc) {
                m->p_conv0 = L.conv("policy_head.seq.0", hc, C, 1);
                macs += (double)hw * hc * C;
                ch = hc;
                idx = 2;
            }
            idx += 1;  // Flatten
            int size = ch * hw;
            if (hs) {
                m->p_fc0 = L.linear("policy_head.seq." + std::to_string(idx), hs, size);
                macs += (double)hs * size;
                size = hs;
                idx += 2;
            }
            m->p_fc1 = L.linear("policy_head.seq." + std::to_string(idx), m->policy_len, size);
            macs += (double)m->policy_len * size;
            break;
        }
        case POLICY_ARIMAA: {  // ArimaaPolicyHead(game, channels, hidden_channels, hidden_size) (post_act.py:144-173)
            int hc = m->arimaa_hidden_channels = L.geti_in("policy_arimaa_hidden_channels", 0, 1, 4096, true);
            int hs = m->arimaa_hidden_size = L.geti_in("policy_arimaa_hidden_size", 0, 1, 1 << 20, true);
            if (L.ok && m->policy_len != 1 + 6 + 4 * hw) {
                err = "arimaa head: policy_len != 1 + 6 + 4*h*w";
                return nullptr;
            }
            if (!L.ok) return nullptr;
            m->policy_conv_channels = 4;
            m->p_conv0 = L.conv("policy_head.bulk.0", C, C, 1);
            m->p_conv1 = L.conv("policy_head.bulk.2", 4, C, 1);
            m->pa_conv = L.conv("policy_head.scalar.0", hc, C, 1);
            m->pa_fc0 = L.linear("policy_head.scalar.3", hs, hc * hw);
            m->pa_fc1 = L.linear("policy_head.scalar.5", 7, hs);
            macs += (double)hw * C * C + (double)hw * 4 * C + (double)hw * hc * C + (double)hs * hc * hw + 7.0 * hs;
            break;
        }
        case POLICY_NONE: break;  // (a dense_network returned above)
    }
    if (!L.ok) return nullptr;
    m->param_count = L.params;
    m->flops_per_eval = 2.0 * macs;
    return m.release();
}

// ---- the same network with its tower widened to `cpad` channels by all-zero filters ----
namespace {
// [cout][cin][k][k] -> [cout_p][cin_p][k][k], new rows / columns zero (bias too)
Conv widen(const Conv &c, int cout_p, int cin_p) {
    Conv o;
    o.cout = cout_p;
    o.cin = cin_p;
    o.k = c.k;
    const int taps = c.k * c.k;
    o.w.assign((size_t)cout_p * cin_p * taps, 0.0f);
    o.b.assign((size_t)cout_p, 0.0f);
    for (int oc = 0; oc < c.cout; oc++) {
        for (int ic = 0; ic < c.cin; ic++)
            for (int t = 0; t < taps; t++) o.w[((size_t)oc * cin_p + ic) * taps + t] = c.w[((size_t)oc * c.cin + ic) * taps + t];
        o.b[oc] = c.b[oc];
    }
    return o;
}
// nn.Linear over a channel-major flatten of [ch][hw]: the new channels' inputs are appended (index c * hw + p)
Linear widen_flat(const Linear &l, int in_p) {
    Linear o;
    o.out = l.out;
    o.in = in_p;
    o.w.assign((size_t)l.out * in_p, 0.0f);
    o.b = l.b;
    for (int r = 0; r < l.out; r++) memcpy(&o.w[(size_t)r * in_p], &l.w[(size_t)r * l.in], (size_t)l.in * 4);
    return o;
}
}  // namespace

Model *pad_channels(const Model &m, int cpad) {
    if (cpad <= m.channels || m.tower_kind != TOWER_RES) return nullptr;  // (LayerNorm runs over d_model: it cannot be widened)
    std::unique_ptr<Model> o(new Model(m));
    const int C = m.channels, hw = m.h * m.w;
    o->channels = cpad;
    // a widened channel carries 0 through the whole tower: zero filters and bias in the stem, relu(0) in conv A,
    // relu(0) + 0 in conv B, and 1 * 0 + 0 through the final BatchNorm
    o->tower[0] = widen(m.tower[0], cpad, m.tower[0].cin);
    for (size_t i = 1; i < m.tower.size(); i++) o->tower[i] = widen(m.tower[i], cpad, cpad);
    o->final_scale.resize(cpad, 1.0f);
    o->final_shift.resize(cpad, 0.0f);
    o->sh_conv = widen(m.sh_conv, m.sh_conv.cout, cpad);
    switch (m.policy_kind) {
        case POLICY_ATAXX_CONV:
        case POLICY_CONV:
            o->p_conv0 = widen(m.p_conv0, cpad, cpad);  // (its hidden layer has the tower's width: relu(0) = 0 in the new channels)
            o->p_conv1 = widen(m.p_conv1, m.p_conv1.cout, cpad);
            if (m.policy_extra_moves) o->p_extra_conv = widen(m.p_extra_conv, m.p_extra_conv.cout, cpad);
            break;
        case POLICY_ATTENTION:
            o->p_bulk = widen(m.p_bulk, m.p_bulk.cout, cpad);
            o->p_under = widen(m.p_under, m.p_under.cout, cpad);
            break;
        case POLICY_DENSE:
            if (m.dense_hidden_channels) o->p_conv0 = widen(m.p_conv0, m.p_conv0.cout, cpad);
            else if (m.dense_hidden_size) o->p_fc0 = widen_flat(m.p_fc0, cpad * hw);
            else o->p_fc1 = widen_flat(m.p_fc1, cpad * hw);
            break;
        case POLICY_ARIMAA:
            o->p_conv0 = widen(m.p_conv0, cpad, cpad);
            o->p_conv1 = widen(m.p_conv1, m.p_conv1.cout, cpad);
            o->pa_conv = widen(m.pa_conv, m.pa_conv.cout, cpad);
            break;
        case POLICY_NONE: break;
    }
    (void)C;
    return o.release();
}

}  // namespace kz
