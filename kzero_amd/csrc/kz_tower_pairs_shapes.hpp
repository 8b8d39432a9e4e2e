// kz_tower_pairs_shapes.hpp — which instance of the one-launch (hi, lo) / plain-f16 tower (kz_tower_pairs.hpp) a shape
// takes: host logic only, shared by the two kernel translation units (kz_tower_split.hip, kz_tower_f16g.hip) and the host-side
// packing and support predicates (kz_tower_pairs_pack.hip).  Included INSIDE `namespace kz { namespace {`.
#pragma once

constexpr int HEAD_PASSES = 5, POLICY = 1880, LOGIT_LD = 96;

#ifdef KZ_EXPERIMENTS
// KZ_SPLIT_MFMA32=1 / KZ_F16G_MFMA32=1: the 256-channel, 64-row launch (nt == 4) through the 32x32x16 kernel (same-box
// A/B); read once — the weight packing and the launch must agree, so both ask with the same (channels, nt)
bool split_uses_32x32(int channels, int nt, bool split) {
    static const bool split_on = [] {
        const char *e = getenv("KZ_SPLIT_MFMA32");
        return e && e[0] == '1';
    }();
    static const bool plain_on = [] {
        const char *e = getenv("KZ_F16G_MFMA32");
        return e && e[0] == '1';
    }();
    return channels == 256 && nt == 4 && (split ? split_on : plain_on);
}
#endif

// Tiles of 16 pixel rows per workgroup (0: no instance): as many whole boards as the LDS images and the accumulators
// hold.  The weight stream is read once per workgroup and layer, so more boards per workgroup = fewer L2 bytes per board.
int split_tiles_for(int hw, int channels, bool split) {
    // (the plain-f16 launch has half the LDS footprint: 256 / 320 channels fit up to 96 squares — Go 9x9)
    if (channels == 256 || channels == 320) return channels == 320 && split ? 0 : hw <= 64 ? 4 : (!split && hw <= 96) ? 6 : 0;
    if (channels == 384 || channels == 512) return !split && hw <= 64 ? 4 : 0;
    if (channels == 192) return split ? (hw <= 64 ? 4 : 0) : hw * 2 <= 112 ? 7 : hw <= 64 ? 4 : hw <= 96 ? 6 : hw <= 176 ? 11 : 0;
    // (twice the boards per workgroup at 128 channels in plain f16: split_wide_tiles_for)
    // (128 channels in plain f16: the eleven- and thirteen-tile instances of the wide tiles also take ONE board of up to 208
    // squares — Go 13x13 — where the per-layer kernel was the only f16 path)
    if (channels == 128 && !split && hw > 96) return hw <= 176 ? 11 : hw <= 208 ? 13 : 0;
    if (channels == 128 || channels == 64) return hw * 2 <= 112 ? 7 : hw <= 64 ? 4 : hw <= 96 ? 6 : 0;
    return 0;
}

// Twice the boards per workgroup for the plain-f16 launch at 128 (and 192) channels: two 8x8 boards in 8 tiles, two 9x9 boards in 11
// (162 of 176 rows are boards; one board in six tiles: 81 of 96), four 7x7 or eight 5x5 boards in 13.  The weight stream
// is read once per workgroup, so this halves the bytes a workgroup pulls from L2 per board.  While these launches waited
// for their weights (four ring stages, rounds 1-3) that was measured SLOWER (half as many workgroups); since the deeper
// ring (round 4) they are bound by the chip's power like the 256-channel launches, and less data moved per MFMA is more
// MFMAs per watt: Go 9x9 16x128 at batch 2048 1.40M -> 1.62M evals/s, chess x 128 1.59M -> 1.69M (batch 256) / 1.57M ->
// 1.70M (1024), Ataxx 7x7 x 128 (20 blocks) at batch 1024 2.04M -> 2.27M — but only with enough workgroups to fill the chip (Ataxx at
// batch 256: 64 workgroups, 2.02M -> 1.69M), and not at 64 channels (latency-bound: 4.44M -> 3.97M at batch 256) or 192
// (no difference).  No fused conv heads at these sizes (the tail's f32 row images do not fit the LDS).
// Round 5: a third level at 128 channels — THREE 9x9 or FOUR 8x8 boards in sixteen tiles (243 / 256 of 256 rows are boards,
// 152 KB of LDS) when the batch still gives 128 such workgroups.  The widest level that fills the chip at `batch` is taken.
int split_wide_tiles_for(int hw, int channels, int batch) {
    // (192 channels: measured late in round 4, chess x 192 at batch 256 / 1024 711k -> 805k / 653k -> 807k evals/s with two boards
    // — its counters read 0.52 busy at 2.06 GHz with one board: neither the matrix cores' limit nor a full clock; three 7x7
    // boards in ten tiles there, four do not fit the LDS)
    int levels[2] = {0, 0};
    if (channels == 192) levels[0] = hw == 64 ? 8 : hw == 81 ? 11 : hw == 49 ? 10 : 0;
    else if (channels == 128) {
        if (hw == 64 || hw == 81) {
            levels[0] = 16;
            levels[1] = hw == 64 ? 8 : 11;
        } else {
            levels[0] = (hw == 49 || hw == 25) ? 13 : 0;
        }
    }
    for (int nt : levels) {
        if (!nt) continue;
        const int per = nt * 16 / hw;
        // enough workgroups to fill the chip: 128 for the two-board levels (round 4's A/Bs); the sixteen-tile level wants two
        // per CU — same-box, Go 9x9 16x128 at batch 2048: 1.554M -> 1.653M evals/s with 683 workgroups of three boards, but
        // chess x 128 at batch 1024: 1.640M -> 1.590M with 256 workgroups of four (profiles/r5/ab_tiles16.txt)
        if ((batch + per - 1) / per >= (nt == 16 ? 512 : 128)) return nt;
    }
    return 0;
}

// LDS bytes of the launch's own images (Geo<C, NT, SPLIT>::LDS_BYTES, restated for the host: kz_tower_pairs.hpp asserts the two
// agree) and what is left of a CU's 160 KB behind them for the conv heads' tail of the plain-f16 launch
constexpr int pairs_own_lds_bytes(int channels, int nt, bool split) {
    const bool two = channels % 256 != 0;
    const int rs = two ? channels + 16 : channels * 2 + 16, img = nt * 16 * rs;
    const int rounded = (2 * img + 16 * rs + 255) / 256 * 256;
    return (split ? 2 : 1) * (two ? 2 * rounded : rounded);
}
constexpr int pairs_f16_tail_scratch_bytes(int channels, int nt) {
    const int room = 160 * 1024 - pairs_own_lds_bytes(channels, nt, false);
    return room < (int)F16_TAIL_SCRATCH_BYTES ? room : (int)F16_TAIL_SCRATCH_BYTES;
}
