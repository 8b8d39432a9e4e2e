// kz_tower_f16g.hip — the SPLIT = false instances of kz_tower_pairs.hpp: the same one-launch tower in plain f16 (one image per
// activation, one MFMA per product, f16 tensors in and out) for the shapes kz_tower.hip (chess: 8x8, 256 channels, attention
// heads) does not take — KZ_DTYPE_F16's "tower_resident_f16g[+heads]": 64 .. 512 channels on small boards, with the wide tiles
// (twice the boards per workgroup) at 128 / 192 channels.  The kernel's body is in kz_tower_pairs.hpp; this file holds this
// family's instances and the choice among them.
#include "kz_tower_pairs.hpp"

namespace kz {

void launch_tower_pairs(const Tower32Args &t, bool split, hipStream_t stream) {
    if (split) {  // (the (hi, lo) instances live in kz_tower_split.hip)
        launch_tower_split(t, stream);
        return;
    }
    int nt = 0, grid = 0;
    const SplitDev d = make_split_dev(t, false, nt, grid);
    if (t.heads.on && t.heads.small_w) {  // "tower_resident_f16g+heads": conv policy heads (tower_split_conv_heads_supported)
        if (t.channels == 256) launch<256, 4, false, 2>(d, grid, stream);
        else if (nt == 4) launch<128, 4, false, 2>(d, grid, stream);
        else if (nt == 7) launch<128, 7, false, 2>(d, grid, stream);
        else if (nt == 8) launch<128, 8, false, 2>(d, grid, stream);
        else if (nt == 11) launch<128, 11, false, 2>(d, grid, stream);
        else if (nt == 13) launch<128, 13, false, 2>(d, grid, stream);
        else if (nt == 16) launch<128, 16, false, 2>(d, grid, stream);
        else launch<128, 6, false, 2>(d, grid, stream);
        return;
    }
#ifdef KZ_EXPERIMENTS
    if (split_uses_32x32(t.channels, nt, false)) launch32<false>(d, grid, stream);
    else
#endif
    if (t.channels == 512) launch<512, 4, false>(d, grid, stream);
    else if (t.channels == 384) launch<384, 4, false>(d, grid, stream);
    else if (t.channels == 320 && nt == 6) launch<320, 6, false>(d, grid, stream);
    else if (t.channels == 320) launch<320, 4, false>(d, grid, stream);
    else if (t.channels == 256 && nt == 6) launch<256, 6, false>(d, grid, stream);
    else if (t.channels == 256) launch<256, 4, false>(d, grid, stream);
    else if (t.channels == 192 && nt == 4) launch<192, 4, false>(d, grid, stream);
    else if (t.channels == 192 && nt == 7) launch<192, 7, false>(d, grid, stream);
    else if (t.channels == 192 && nt == 11) launch<192, 11, false>(d, grid, stream);
    else if (t.channels == 192 && nt == 8) launch<192, 8, false>(d, grid, stream);
    else if (t.channels == 192 && nt == 10) launch<192, 10, false>(d, grid, stream);
    else if (t.channels == 192) launch<192, 6, false>(d, grid, stream);
    else if (t.channels == 128 && nt == 4) launch<128, 4, false>(d, grid, stream);
    else if (t.channels == 128 && nt == 7) launch<128, 7, false>(d, grid, stream);
    else if (t.channels == 128 && nt == 8) launch<128, 8, false>(d, grid, stream);
    else if (t.channels == 128 && nt == 11) launch<128, 11, false>(d, grid, stream);
    else if (t.channels == 128 && nt == 13) launch<128, 13, false>(d, grid, stream);
    else if (t.channels == 128 && nt == 16) launch<128, 16, false>(d, grid, stream);
    else if (t.channels == 128) launch<128, 6, false>(d, grid, stream);
    else if (nt == 4) launch<64, 4, false>(d, grid, stream);
    else if (nt == 7) launch<64, 7, false>(d, grid, stream);
    else launch<64, 6, false>(d, grid, stream);
}

}  // namespace kz
