// kz_tower_split.hip — the SPLIT = true instances of kz_tower_pairs.hpp: the board-resident ResTower with f32-EQUIVALENT results
// on the f16 matrix cores (every activation and weight a (hi, lo) f16 pair, three MFMAs per product, f32 accumulation),
// KZ_DTYPE_F32_SPLIT16's "tower_resident_split16[+heads]".  The kernel's body, its LDS geometry and the arithmetic are in
// kz_tower_pairs.hpp; this file holds what is this family's own: which instances exist and which one a launch takes.
// (Python: python/lib/model/post_act.py:201-239; the fused heads :10-23, :54-141.)
#include "kz_tower_pairs.hpp"

namespace kz {

void launch_tower_split(const Tower32Args &t, hipStream_t stream) {
    int nt = 0, grid = 0;
    const SplitDev d = make_split_dev(t, true, nt, grid);
    if (t.heads.on) {  // (the engine asked tower_split_[conv_]heads_supported)
        if (t.heads.small_w) {  // conv policy heads (Ataxx, Go 9x9) at 128 / 256 channels
            if (t.channels == 256) launch<256, 4, true, 2>(d, grid, stream);
            else if (nt == 4) launch<128, 4, true, 2>(d, grid, stream);
            else if (nt == 7) launch<128, 7, true, 2>(d, grid, stream);
            else launch<128, 6, true, 2>(d, grid, stream);
            return;
        }
        launch<256, 4, true, 1>(d, grid, stream);  // the chess attention network
        return;
    }
#ifdef KZ_EXPERIMENTS
    if (split_uses_32x32(t.channels, nt, true)) launch32<true>(d, grid, stream);
    else
#endif
    if (t.channels == 256) launch<256, 4, true>(d, grid, stream);
    else if (t.channels == 192) launch<192, 4, true>(d, grid, stream);
    else if (t.channels == 128 && nt == 4) launch<128, 4, true>(d, grid, stream);
    else if (t.channels == 128 && nt == 7) launch<128, 7, true>(d, grid, stream);
    else if (t.channels == 128) launch<128, 6, true>(d, grid, stream);
    else if (nt == 4) launch<64, 4, true>(d, grid, stream);
    else if (nt == 7) launch<64, 7, true>(d, grid, stream);
    else launch<64, 6, true>(d, grid, stream);
}

}  // namespace kz
