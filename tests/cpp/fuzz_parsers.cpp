// fuzz_parsers.cpp — mutation fuzz of the two model parsers (KZMODEL1: kz_model.cpp, ONNX: kz_onnx.cpp) under
// AddressSanitizer/UBSan: truncations, byte flips, length-field edits of the golden files.  A parser may reject a
// mutated file (nullptr + message) or accept it; it must never crash, read out of bounds or loop forever.
// Built and run by tests/test_parser_fuzz.py (CPU only: the parsers have no HIP dependency).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../kzero_amd/csrc/kz_model.hpp"

static std::vector<unsigned char> read_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string s = ss.str();
    return std::vector<unsigned char>(s.begin(), s.end());
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <iterations> <seed> <file>...\n", argv[0]);
        return 2;
    }
    const int iters = std::atoi(argv[1]);
    std::mt19937 rng((unsigned)std::atoi(argv[2]));
    long accepted = 0, rejected = 0;
    for (int fi = 3; fi < argc; fi++) {
        const std::vector<unsigned char> base = read_file(argv[fi]);
        if (base.empty()) {
            std::fprintf(stderr, "cannot read %s\n", argv[fi]);
            return 2;
        }
        const bool onnx = kz::looks_like_onnx(base.data(), base.size());
        auto parse = [&](const std::vector<unsigned char> &b) {
            std::string err;
            // heap copy of the exact size so that ASan sees every over-read
            std::unique_ptr<unsigned char[]> copy(new unsigned char[b.size() ? b.size() : 1]);
            if (!b.empty()) std::memcpy(copy.get(), b.data(), b.size());
            std::unique_ptr<kz::Model> m(onnx ? kz::parse_onnx(copy.get(), b.size(), 1, err)
                                               : kz::parse_model(copy.get(), b.size(), err));
            if (m) accepted++;
            else {
                rejected++;
                if (err.empty()) {
                    std::fprintf(stderr, "rejected without a message\n");
                    std::exit(1);
                }
            }
        };
        parse(base);  // the unmodified file parses
        for (int it = 0; it < iters; it++) {
            std::vector<unsigned char> b = base;
            switch (rng() % 5) {
                case 0:  // truncate
                    b.resize(rng() % (b.size() + 1));
                    break;
                case 1:  // flip a few bytes, biased to the structured head of the file
                    for (int k = 0, n = 1 + (int)(rng() % 4); k < n; k++) {
                        const size_t lim = (rng() & 1) ? std::min<size_t>(b.size(), 4096) : b.size();
                        b[rng() % lim] ^= (unsigned char)(1u << (rng() % 8));
                    }
                    break;
                case 2:  // overwrite 4 bytes with an extreme value (length fields, counts, dims)
                {
                    const size_t lim = std::min<size_t>(b.size() - 4, 8192);
                    const size_t at = rng() % lim;
                    const unsigned v[] = {0u, 0xffffffffu, 0x7fffffffu, 0x80000000u, 1u << 20};
                    const unsigned x = v[rng() % 5];
                    std::memcpy(&b[at], &x, 4);
                    break;
                }
                case 3:  // delete a span
                {
                    const size_t at = rng() % b.size(), n = std::min<size_t>(b.size() - at, 1 + rng() % 64);
                    b.erase(b.begin() + (long)at, b.begin() + (long)(at + n));
                    break;
                }
                default:  // insert garbage
                {
                    const size_t at = rng() % b.size();
                    std::vector<unsigned char> junk(1 + rng() % 32);
                    for (auto &c : junk) c = (unsigned char)rng();
                    b.insert(b.begin() + (long)at, junk.begin(), junk.end());
                    break;
                }
            }
            parse(b);
        }
    }
    std::printf("parser fuzz ok: %ld accepted, %ld rejected\n", accepted, rejected);
    return 0;
}
