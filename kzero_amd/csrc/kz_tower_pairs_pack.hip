// kz_tower_pairs_pack.hip — host side of the one-launch (hi, lo) / plain-f16 towers (kz_tower_pairs.hpp): which shapes they
// take, the weight streams in MFMA fragment order, the heads' weights.  No kernel in this file.
#include <cstdlib>
#include <vector>

#include "kz_kernels.hpp"

namespace kz {

namespace {
#include "kz_tower_pairs_shapes.hpp"
}  // namespace


// (the stem takes the input planes in chunks of 32; beyond one chunk they are staged in the Y image: rows of 64 B per
// chunk must fit a row of 2 C + 16 B)
bool tower_split_supported(int h, int w, int channels, int depth, int c_in, bool split) {
    const int nt = split_tiles_for(h * w, channels, split);
#ifdef KZ_EXPERIMENTS
    if (c_in > 32 && split_uses_32x32(channels, nt, split)) return false;  // (the 32x32x16 variant has the one-chunk stem)
#endif
    return depth >= 1 && c_in >= 1 && (c_in + 31) / 32 <= channels / 32 && h >= 2 && w >= 2 && w <= 32 && nt != 0;
}

int tower_split_boards_per_workgroup(int h, int w, int channels, bool split, int wide_batch) {
    int nt = !split && wide_batch ? split_wide_tiles_for(h * w, channels, wide_batch) : 0;
    if (!nt) nt = split_tiles_for(h * w, channels, split);
    return nt ? nt * 16 / (h * w) : 0;
}

// whether a plain-f16 engine of this shape takes the wide tiles: at least 128 workgroups at max_batch
bool tower_split_wide_supported(int h, int w, int channels, int max_batch) {
    return split_tiles_for(h * w, channels, false) != 0 && split_wide_tiles_for(h * w, channels, max_batch) != 0;
}

size_t tower_split_stem_elems(int channels, int c_in, bool split) {  // f16 elements of the stem's k-steps
    return (size_t)9 * ((c_in + 31) / 32) * (split ? 2 : 1) * channels * 32;
}

size_t tower_split_weight_elems(int channels, int depth, int c_in, bool split) {  // f16 elements
    const size_t step = (size_t)(split ? 2 : 1) * channels * 32;  // [hi | lo][channels][32]
    return tower_split_stem_elems(channels, c_in, split) + (size_t)2 * depth * 9 * (channels / 32) * step;
}

// OIHW f32 (BN folded) -> k-steps of [hi | lo][wave 4][ot C/64][lane 64][8] f16; element j of lane (fr, kq) of (wave, ot)
// is W[oc = 16*(wave*C/64 + ot) + fr][channel][tap], channel = 8*chunk + {0, C/2, C/4, 3C/4}[kq] + j for a tower layer (one
// k-step per tap and chunk of 32 channels) and 32*chunk + 8*kq + j for the stem (one k-step per tap and chunk of 32 padded
// input channels).
void tower_split_pack_weights(const float *oihw, int cout, int cin, int hw, bool stem, bool split, uint16_t *dst) {
    const int kq_base[4] = {0, cout / 2, cout / 4, 3 * cout / 4};  // tower layers: cin == cout
    const int nchunk = stem ? (cin + 31) / 32 : cin / 32, ot_n = cout / 64;
    const size_t part = (size_t)cout * 32;  // f16 elements of the hi (or lo) half of a k-step
    (void)hw;
#ifdef KZ_EXPERIMENTS
    if (split_uses_32x32(cout, split_tiles_for(hw, cout, split), split)) {
        // kz_tower_resident_split32: [hi | lo][wave 4][f 4][lane 64][8]; fragment f = 2 o + half, lane (n, kg): output
        // channel 64 wave + 32 o + n, piece q = 2 half + kg of the k-step: channel 8 chunk + kq_base[q] + j (stem: 8 q + j)
        for (int tap = 0; tap < 9; tap++)
            for (int chunk = 0; chunk < nchunk; chunk++) {
                uint16_t *step = dst + ((size_t)tap * nchunk + chunk) * (split ? 2 : 1) * part;
                for (int wave = 0; wave < 4; wave++)
                    for (int f = 0; f < 4; f++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int j = 0; j < 8; j++) {
                                const int oc = 64 * wave + 32 * (f >> 1) + (lane & 31);
                                const int q = 2 * (f & 1) + (lane >> 5);
                                const int ch = stem ? 8 * q + j : 8 * chunk + kq_base[q] + j;
                                float v = 0.0f;
                                if (oc < cout && ch < cin) v = oihw[((size_t)oc * cin + ch) * 9 + tap];
                                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                                uint16_t hb, lb;
                                __builtin_memcpy(&hb, &hi, 2);
                                __builtin_memcpy(&lb, &lo, 2);
                                const size_t e = (((size_t)wave * 4 + f) * 64 + lane) * 8 + j;
                                step[e] = hb;
                                if (split) step[part + e] = lb;
                            }
            }
        return;
    }
#endif
    for (int tap = 0; tap < 9; tap++)
        for (int chunk = 0; chunk < nchunk; chunk++) {
            uint16_t *step = dst + ((size_t)tap * nchunk + chunk) * (split ? 2 : 1) * part;
            for (int wave = 0; wave < 4; wave++)
                for (int ot = 0; ot < ot_n; ot++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int oc = 16 * (wave * ot_n + ot) + (lane & 15);
                            const int kq = lane >> 4;
                            const int ch = stem ? 32 * chunk + 8 * kq + j : 8 * chunk + kq_base[kq] + j;
                            float v = 0.0f;
                            if (oc < cout && ch < cin) v = oihw[((size_t)oc * cin + ch) * 9 + tap];
                            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                            uint16_t hb, lb;
                            __builtin_memcpy(&hb, &hi, 2);
                            __builtin_memcpy(&lb, &lo, 2);
                            const size_t e = (((size_t)wave * ot_n + ot) * 64 + lane) * 8 + j;
                            step[e] = hb;
                            if (split) step[part + e] = lb;
                        }
        }
}

// The chess attention network's heads inside the split launch: 256 tower channels = query channels on 8x8, the
// reference's ScalarHead(8, C, 4, 32) (post_act.py:10-23, :115-141).
bool tower_split_heads_supported(int policy_kind, int query_channels, int policy_len, int h, int w, int channels, int sh_channels,
                                 int sh_size) {
    return policy_kind == 2 && channels == 256 && query_channels == 256 && policy_len == POLICY && h == 8 && w == 8 &&
           sh_channels == 4 && sh_size == 32;
}

size_t tower_split_heads_weight_elems() { return (size_t)HEAD_PASSES * 8 * 2 * 256 * 32; }  // f16 elements: 5 passes of 8 k-steps

// One 1x1 convolution [256 out][256 in] as a pass of 8 k-steps in the tower layers' (hi, lo) fragment order
static void pack_1x1_split(const float *w, uint16_t *dst) {
    const int kq_base[4] = {0, 128, 64, 192};
    const size_t part = (size_t)256 * 32;
    for (int chunk = 0; chunk < 8; chunk++) {
        uint16_t *step = dst + (size_t)chunk * 2 * part;
        for (int wave = 0; wave < 4; wave++)
            for (int ot = 0; ot < 4; ot++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int oc = 16 * (wave * 4 + ot) + (lane & 15);
                        const int ch = 8 * chunk + kq_base[lane >> 4] + j;
                        const float v = w[(size_t)oc * 256 + ch];
                        const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                        uint16_t hb, lb;
                        __builtin_memcpy(&hb, &hi, 2);
                        __builtin_memcpy(&lb, &lo, 2);
                        const size_t e = (((size_t)wave * 4 + ot) * 64 + lane) * 8 + j;
                        step[e] = hb;
                        step[part + e] = lb;
                    }
    }
}

// conv_bulk [512][256] and conv_under [768][256] (post_act.py:122-123) -> the five passes behind the tower's k-steps, in
// the order the launch runs them: conv_bulk[0:Q) (q_from), conv_under's channels 3 q + s as three s-major passes
// (under.reshape(Q, 24)[q][8 s + x], post_act.py:134), conv_bulk[Q:2Q) (the board squares of q_to); bias5 [5][256] alike
void tower_split_pack_heads(const float *w_bulk, const float *b_bulk, const float *w_under, const float *b_under, uint16_t *dst,
                            float *bias5) {
    const size_t pass = (size_t)8 * 2 * 256 * 32;
    std::vector<float> tmp((size_t)256 * 256);
    pack_1x1_split(w_bulk, dst);
    for (int q = 0; q < 256; q++) bias5[q] = b_bulk[q];
    for (int sp = 0; sp < 3; sp++) {
        for (int q = 0; q < 256; q++) {
            for (int c = 0; c < 256; c++) tmp[(size_t)q * 256 + c] = w_under[(size_t)(3 * q + sp) * 256 + c];
            bias5[(1 + sp) * 256 + q] = b_under[3 * q + sp];
        }
        pack_1x1_split(tmp.data(), dst + pass * (1 + sp));
    }
    pack_1x1_split(w_bulk + (size_t)256 * 256, dst + pass * 4);
    for (int q = 0; q < 256; q++) bias5[4 * 256 + q] = b_bulk[256 + q];
}

// ---- conv policy heads in the split launch: the shapes of the exact-f32 launch's fused heads at 128 / 256 channels ----
bool tower_split_conv_heads_supported(int policy_kind, int extra_moves, int pc, int h, int w, int channels, int hc, int hs,
                                      bool split, int wide_batch) {
    if (channels != 128 && channels != 256) return false;
    // (wide_batch: the engine's max_batch when it takes the wide tiles — a launch with a smaller batch takes a narrower level,
    // whose fewer boards need less of the tail's scratch)
    int nt = !split && wide_batch ? split_wide_tiles_for(h * w, channels, wide_batch) : 0;
    if (!nt) nt = split_tiles_for(h * w, channels, split);
    // instances: <256, 4> and <128, 4 / 6 / 7> (plain f16: <128, 8 / 11 / 13> too).  Split arithmetic: the f32 row images of the
    // tail must fit the LDS next to nothing else; plain f16: the tail's scratch behind the launch's own images.
    if (nt == 0 || (channels == 256 && nt != 4)) return false;
    if (split && (size_t)(16 + 2 * nt * 16) * (channels * 4 + 16) > (size_t)160 * 1024) return false;
    return conv_heads_fit(nt, policy_kind, extra_moves, pc, h, w, channels, hc, hs,
                          split ? 0 : (size_t)pairs_f16_tail_scratch_bytes(channels, nt));
}

size_t tower_split_conv_heads_weight_elems(int channels, bool split) { return (size_t)(channels / 32) * (split ? 2 : 1) * channels * 32; }  // one pass

// The policy head's first 1x1 convolution [C out][C in] as one pass of C/32 k-steps in the tower layers' (hi, lo) fragment
// order (lane group kq takes channels 8 chunk + {0, C/2, C/4, 3C/4}[kq] + j)
void tower_split_pack_conv_heads(const float *w, int channels, bool split, uint16_t *dst) {
    const int C = channels, ot_n = C / 64;
    const size_t part = (size_t)C * 32;
    for (int chunk = 0; chunk < C / 32; chunk++) {
        uint16_t *step = dst + (size_t)chunk * (split ? 2 : 1) * part;
        for (int wave = 0; wave < 4; wave++)
            for (int ot = 0; ot < ot_n; ot++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int kq = lane >> 4;
                        const int oc = 16 * (wave * ot_n + ot) + (lane & 15);
                        const int ch = 8 * chunk + (C / 2) * (kq & 1) + (C / 4) * (kq >> 1) + j;
                        const float v = w[(size_t)oc * C + ch];
                        const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                        uint16_t hb, lb;
                        __builtin_memcpy(&hb, &hi, 2);
                        __builtin_memcpy(&lb, &lo, 2);
                        const size_t e = (((size_t)wave * ot_n + ot) * 64 + lane) * 8 + j;
                        step[e] = hb;
                        if (split) step[part + e] = lb;
                    }
    }
}

size_t tower_split_small_weight16_elems(int channels) { return (size_t)2 * (channels / 32) * 2 * 64 * 8; }

// The plain-f16 launch's two small convolutions ([hc (+ 1 extra-move)][C] over the tower output, [pc][C] over the policy
// head's hidden layer; at most 32 output channels each, zero-padded) as f16 MFMA row-operand fragments in the tower layers'
// channel assignment: [conv 2][k-step C/32][tile 2][lane 64][8]
void tower_split_pack_small_weights16(const float *sh_w0, int hc, const float *pe_wc, const float *p_w1, int pc, int channels,
                                      uint16_t *dst) {
    const int C = channels;
    size_t o = 0;
    for (int conv = 0; conv < 2; conv++)
        for (int chunk = 0; chunk < C / 32; chunk++)
            for (int mt = 0; mt < 2; mt++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int kq = lane >> 4, oc = 16 * mt + (lane & 15);
                        const int ch = 8 * chunk + (C / 2) * (kq & 1) + (C / 4) * (kq >> 1) + j;
                        float v = 0.0f;
                        if (conv == 0) {
                            if (oc < hc) v = sh_w0[(size_t)oc * C + ch];
                            else if (oc == hc && pe_wc) v = pe_wc[ch];
                        } else if (oc < pc) {
                            v = p_w1[(size_t)oc * C + ch];
                        }
                        const _Float16 h = (_Float16)v;
                        uint16_t hb;
                        __builtin_memcpy(&hb, &h, 2);
                        dst[o++] = hb;
                    }
}

}  // namespace kz
