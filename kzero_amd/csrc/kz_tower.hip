// kz_tower.hip — board-resident ResTower kernel (placeholder until the kernel lands; the generic path runs).
#include "kz_kernels.hpp"

namespace kz {

bool tower_resident_supported(int, int, int, int, int) { return false; }
size_t tower_packed_weight_elems(int, int) { return 0; }
void tower_pack_weights(const float *, int, int, int, uint16_t *) {}
void launch_tower_resident(const TowerArgs &, hipStream_t) {}

}  // namespace kz
