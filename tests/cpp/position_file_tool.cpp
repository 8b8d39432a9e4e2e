// position_file_tool.cpp — cross-check helper of tests/test_position_file.py for kzero_amd/csrc/host/position_file.hpp:
//   write <path> <seed> <games>   : writes a deterministic synthetic file (chess-shaped records), prints a checksum line per position
//   dump <path>                   : reads a file (written by anyone) and prints the same checksum lines
// A checksum line = "pos <i> mv <count> sum <fnv1a of the record's bytes in file order>".
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../kzero_amd/csrc/host/position_file.hpp"

using namespace kz::host;

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
static void print_record(size_t i, const PositionRecord &r) {
    uint64_t h = 1469598103934665603ull;
    h = fnv(h, r.scalars, sizeof(r.scalars));
    h = fnv(h, r.bits.data(), r.bits.size());
    h = fnv(h, r.input_scalars.data(), r.input_scalars.size() * 4);
    h = fnv(h, r.policy_indices.data(), r.policy_indices.size() * 4);
    h = fnv(h, r.policy_values.data(), r.policy_values.size() * 4);
    std::printf("pos %zu mv %zu sum %016llx\n", i, r.policy_values.size(), (unsigned long long)h);
}

int main(int argc, char **argv) {
    try {
        if (argc >= 5 && std::string(argv[1]) == "write") {
            std::mt19937 rng((unsigned)std::atoi(argv[3]));
            const int games = std::atoi(argv[4]);
            PositionFileWriter w(argv[2], "chess", {13, 8, 8}, 8, {1880});
            size_t index = 0;
            for (int g = 0; g < games; g++) {
                std::vector<PositionRecord> recs(2 + rng() % 5);
                for (size_t k = 0; k < recs.size(); k++) {
                    PositionRecord &r = recs[k];
                    const bool terminal = k + 1 == recs.size();
                    const size_t mv = terminal ? 0 : 1 + rng() % 40;
                    for (size_t s = 0; s < POSITION_SCALAR_COUNT; s++) r.scalars[s] = (float)(rng() % 1000) / 8.0f;
                    r.scalars[0] = (float)g;
                    r.scalars[1] = (float)k;
                    r.scalars[2] = (float)(recs.size() - 1);
                    r.scalars[POSITION_SCALAR_AVAILABLE_MV_COUNT] = (float)mv;
                    r.bits.resize(w.meta().bits_bytes());
                    for (auto &b : r.bits) b = (uint8_t)rng();
                    r.input_scalars.resize(8);
                    for (auto &s : r.input_scalars) s = (float)(rng() % 100) / 4.0f;
                    r.policy_indices.resize(mv);
                    r.policy_values.assign(mv, mv ? 1.0f / (float)mv : 0.0f);
                    for (auto &p : r.policy_indices) p = rng() % 1880;
                }
                w.append_game(recs);
                for (const auto &r : recs) print_record(index++, r);
            }
            w.finish();
            return 0;
        }
        if (argc >= 3 && std::string(argv[1]) == "dump") {
            PositionFile f(argv[2]);
            std::printf("meta game %s positions %zu games %lld bits_bytes %zu scalars %lld starts", f.meta().game.c_str(), f.size(),
                        (long long)f.meta().game_count, f.meta().bits_bytes(), (long long)f.meta().input_scalar_count);
            for (auto s : f.game_starts()) std::printf(" %llu", (unsigned long long)s);
            std::printf("\n");
            for (size_t i = 0; i < f.size(); i++) print_record(i, f.position(i));
            return 0;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    std::fprintf(stderr, "usage: %s write <path> <seed> <games> | dump <path>\n", argv[0]);
    return 2;
}
