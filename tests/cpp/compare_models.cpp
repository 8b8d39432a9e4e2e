// compare_models.cpp — the model an ONNX file parses to (kz_onnx.cpp: normalising pass + architecture matcher) against
// the model the KZMODEL1 container of the same network parses to (kz_model.cpp): same architecture descriptor, every
// folded tensor equal within 1e-6 relative.  CPU only (the parsers have no HIP dependency); tests/test_onnx_variants.py
// runs it over every file of tests/golden/onnx_variants/ under AddressSanitizer.
//   compare_models <file.onnx> <n_scalar> <file.kzm>        exit 0 = equal; prints the first difference otherwise
//   compare_models --legacy <file.onnx> <n_scalar> <file.kzm>   the (value, wdl, policy) output form: the scalar head's last
//                                                            Linear must be rows 0..3 of the container's, row 4 = NaN
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../kzero_amd/csrc/kz_model.hpp"

static std::string slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static int bad(const std::string &what) {
    std::printf("DIFFERENT: %s\n", what.c_str());
    return 1;
}

static bool close(const std::vector<float> &a, const std::vector<float> &b, const char *what, std::string &msg) {
    if (a.size() != b.size()) {
        msg = std::string(what) + ": size " + std::to_string(a.size()) + " vs " + std::to_string(b.size());
        return false;
    }
    for (size_t i = 0; i < a.size(); i++)
        if (!(std::fabs(a[i] - b[i]) <= 1e-6f * std::fmax(1.0f, std::fabs(b[i])))) {
            msg = std::string(what) + "[" + std::to_string(i) + "]: " + std::to_string(a[i]) + " vs " + std::to_string(b[i]);
            return false;
        }
    return true;
}
static bool same(const kz::Conv &a, const kz::Conv &b, const char *what, std::string &msg) {
    if (a.cout != b.cout || a.cin != b.cin || a.k != b.k) { msg = std::string(what) + ": shape"; return false; }
    return close(a.w, b.w, what, msg) && close(a.b, b.b, what, msg);
}
static bool same(const kz::Linear &a, const kz::Linear &b, const char *what, std::string &msg) {
    if (a.out != b.out || a.in != b.in) { msg = std::string(what) + ": shape"; return false; }
    return close(a.w, b.w, what, msg) && close(a.b, b.b, what, msg);
}

int main(int argc, char **argv) {
    const bool legacy = argc > 1 && !strcmp(argv[1], "--legacy");
    if (argc != (legacy ? 5 : 4)) {
        std::fprintf(stderr, "usage: %s [--legacy] file.onnx n_scalar file.kzm\n", argv[0]);
        return 2;
    }
    const int base = legacy ? 2 : 1;
    const std::string ob = slurp(argv[base]), kb = slurp(argv[base + 2]);
    std::string err;
    std::unique_ptr<kz::Model> o(kz::parse_onnx(ob.data(), ob.size(), atoi(argv[base + 1]), err));
    if (!o) return bad("ONNX rejected: " + err);
    std::unique_ptr<kz::Model> k(kz::parse_model(kb.data(), kb.size(), err));
    if (!k) return bad("container rejected: " + err);
    if (o->h != k->h || o->w != k->w || o->c_in != k->c_in || o->n_scalar != k->n_scalar || o->n_bool != k->n_bool ||
        o->depth != k->depth || o->channels != k->channels || o->policy_len != k->policy_len || o->policy_kind != k->policy_kind ||
        o->policy_conv_channels != k->policy_conv_channels || o->policy_extra_moves != k->policy_extra_moves ||
        o->policy_query_channels != k->policy_query_channels || o->dense_hidden_channels != k->dense_hidden_channels ||
        o->dense_hidden_size != k->dense_hidden_size || o->arimaa_hidden_channels != k->arimaa_hidden_channels ||
        o->arimaa_hidden_size != k->arimaa_hidden_size || o->tower.size() != k->tower.size())
        return bad("architecture descriptor");
    std::string msg;
    for (size_t i = 0; i < o->tower.size(); i++)
        if (!same(o->tower[i], k->tower[i], ("tower." + std::to_string(i)).c_str(), msg)) return bad(msg);
    if (!close(o->final_scale, k->final_scale, "final_scale", msg) || !close(o->final_shift, k->final_shift, "final_shift", msg)) return bad(msg);
    if (!same(o->sh_conv, k->sh_conv, "sh_conv", msg) || !same(o->sh_fc0, k->sh_fc0, "sh_fc0", msg)) return bad(msg);
    if (legacy) {
        if (o->sh_fc1.out != 5 || o->sh_fc1.in != k->sh_fc1.in) return bad("legacy scalar head shape");
        for (int r = 0; r < 4; r++) {
            for (int i = 0; i < o->sh_fc1.in; i++)
                if (o->sh_fc1.w[(size_t)r * o->sh_fc1.in + i] != k->sh_fc1.w[(size_t)r * k->sh_fc1.in + i]) return bad("legacy scalar head row");
            if (o->sh_fc1.b[r] != k->sh_fc1.b[r]) return bad("legacy scalar head bias");
        }
        if (!std::isnan(o->sh_fc1.b[4])) return bad("legacy moves_left must be NaN");
    } else if (!same(o->sh_fc1, k->sh_fc1, "sh_fc1", msg)) return bad(msg);
    if (!same(o->p_conv0, k->p_conv0, "p_conv0", msg) || !same(o->p_conv1, k->p_conv1, "p_conv1", msg) ||
        !same(o->p_extra_conv, k->p_extra_conv, "p_extra_conv", msg) || !same(o->p_extra_fc, k->p_extra_fc, "p_extra_fc", msg) ||
        !same(o->p_bulk, k->p_bulk, "p_bulk", msg) || !same(o->p_under, k->p_under, "p_under", msg) ||
        !same(o->p_fc0, k->p_fc0, "p_fc0", msg) || !same(o->p_fc1, k->p_fc1, "p_fc1", msg) ||
        !same(o->pa_conv, k->pa_conv, "pa_conv", msg) || !same(o->pa_fc0, k->pa_fc0, "pa_fc0", msg) || !same(o->pa_fc1, k->pa_fc1, "pa_fc1", msg))
        return bad(msg);
    if (o->flat_to_att != k->flat_to_att) return bad("flat_to_att");
    // DenseNetwork (python/lib/model/simple.py)
    if (o->dn_res != k->dn_res || o->dn_blocks.size() != k->dn_blocks.size()) return bad("dense network descriptor");
    if (!same(o->dn_in, k->dn_in, "dn_in", msg) || !same(o->dn_out, k->dn_out, "dn_out", msg) || !close(o->dn_sf, k->dn_sf, "dn_sf", msg) ||
        !close(o->dn_tf, k->dn_tf, "dn_tf", msg))
        return bad(msg);
    for (size_t i = 0; i < o->dn_blocks.size(); i++) {
        const auto &a = o->dn_blocks[i], &b = k->dn_blocks[i];
        if (!close(a.sa, b.sa, "dn.sa", msg) || !close(a.ta, b.ta, "dn.ta", msg) || !close(a.sb, b.sb, "dn.sb", msg) || !close(a.tb, b.tb, "dn.tb", msg) ||
            !same(a.la, b.la, "dn.la", msg) || !same(a.lb, b.lb, "dn.lb", msg))
            return bad(msg);
    }
    // AttentionTower (python/lib/model/attention.py): the descriptor and every matrix
    if (o->tower_kind != k->tower_kind || o->att_heads != k->att_heads || o->att_dk != k->att_dk || o->att_dv != k->att_dv ||
        o->att_dff != k->att_dff || o->att_layers.size() != k->att_layers.size() || std::fabs(o->att_alpha - k->att_alpha) > 1e-6f ||
        std::fabs(o->ln_eps - k->ln_eps) > 1e-12f)
        return bad("attention tower descriptor");
    if (!close(o->att_expand, k->att_expand, "att_expand", msg) || !close(o->att_embedding, k->att_embedding, "att_embedding", msg)) return bad(msg);
    for (size_t i = 0; i < o->att_layers.size(); i++) {
        const std::string p = "encoder." + std::to_string(i);
        if (!close(o->att_layers[i].qkv, k->att_layers[i].qkv, (p + ".qkv").c_str(), msg) ||
            !close(o->att_layers[i].out, k->att_layers[i].out, (p + ".out").c_str(), msg) ||
            !close(o->att_layers[i].ff0, k->att_layers[i].ff0, (p + ".ff0").c_str(), msg) ||
            !close(o->att_layers[i].ff1, k->att_layers[i].ff1, (p + ".ff1").c_str(), msg))
            return bad(msg);
    }
    // (an AttentionTower has no BatchNorm to fold: the ONNX file holds the container's parameters one for one)
    if (o->tower_kind == kz::TOWER_ATTENTION &&
        (o->param_count != k->param_count || std::fabs(o->flops_per_eval - k->flops_per_eval) > 1e-6 * k->flops_per_eval))
        return bad("param_count / flops_per_eval");
    std::printf("models equal\n");
    return 0;
}
