// symmetry.hpp — `RandomSymmetryNetwork` (rust/kz-core/src/network/symmetry.rs:18-68,126-148) and the D4 symmetry of
// Ataxx positions and moves (board-game's D4Symmetry: transpose first, then the flips; pinned against the reference's
// python/lib/mapping/ataxx_symmetry.json in tests).
//
// Symmetric-board concept:  static bool symmetry_is_unit();  static int symmetry_count();
//                           B map(int sym) const;  Move map_move(int sym, const Move&) const;
#pragma once
#include <algorithm>
#include <random>

#include "mapping.hpp"
#include "network.hpp"

namespace kz::host {

// index = 4*transpose + 2*flip_x + flip_y (the order of D4Symmetry::all() / ataxx_symmetry.json)
struct D4 {
    bool transpose, flip_x, flip_y;
    static D4 from_index(int i) { return {(i & 4) != 0, (i & 2) != 0, (i & 1) != 0}; }
    void map_xy(int size, int &x, int &y) const {
        if (transpose) std::swap(x, y);
        if (flip_x) x = size - 1 - x;
        if (flip_y) y = size - 1 - y;
    }
};

inline AtaxxMove ataxx_map_move(int size, int sym, AtaxxMove mv) {
    const D4 d = D4::from_index(sym);
    if (mv.kind == AtaxxMove::Pass) return mv;
    d.map_xy(size, mv.to_x, mv.to_y);
    if (mv.kind == AtaxxMove::Jump) d.map_xy(size, mv.from_x, mv.from_y);
    return mv;
}

inline uint64_t ataxx_map_tiles(int size, int sym, uint64_t tiles) {
    const D4 d = D4::from_index(sym);
    uint64_t out = 0;
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
            if ((tiles >> (y * size + x)) & 1) {
                int nx = x, ny = y;
                d.map_xy(size, nx, ny);
                out |= 1ull << (ny * size + nx);
            }
    return out;
}

// AtaxxPosition with the symmetric-board interface
struct AtaxxSymBoard : AtaxxPosition {
    static bool symmetry_is_unit() { return false; }
    static int symmetry_count() { return 8; }
    AtaxxSymBoard map(int sym) const {
        AtaxxSymBoard r = *this;
        r.tiles_next = ataxx_map_tiles(size, sym, tiles_next);
        r.tiles_other = ataxx_map_tiles(size, sym, tiles_other);
        r.gaps = ataxx_map_tiles(size, sym, gaps);
        if (moves) {
            // a real board regenerates its moves; the order after mapping is whatever its move generator yields.
            // Here: the mapped moves sorted by policy index, so that un-mapping is exercised against a different order.
            std::vector<AtaxxMove> mapped;
            for (const auto &mv : *moves) mapped.push_back(ataxx_map_move(size, sym, mv));
            AtaxxStdMapper mapper(size);
            std::sort(mapped.begin(), mapped.end(),
                      [&](const AtaxxMove &a, const AtaxxMove &b) { return mapper.move_to_index(a) < mapper.move_to_index(b); });
            r.moves = std::move(mapped);
        }
        return r;
    }
    AtaxxMove map_move(int sym, const AtaxxMove &mv) const { return ataxx_map_move(size, sym, mv); }
};

// symmetry.rs:126-148
template <class B>
ZeroEvaluation unmap_eval(const B &board, int sym, const B &mapped_board, const ZeroEvaluation &mapped_eval) {
    auto mapped_moves = mapped_board.available_moves();
    ZeroEvaluation out;
    out.values = mapped_eval.values;
    auto moves = board.available_moves();
    if (moves) {
        for (const auto &mv : *moves) {
            const auto mapped_mv = board.map_move(sym, mv);
            auto it = std::find(mapped_moves->begin(), mapped_moves->end(), mapped_mv);
            if (it == mapped_moves->end()) throw std::logic_error("mapped move not available on the mapped board");
            out.policy.push_back(mapped_eval.policy[(size_t)(it - mapped_moves->begin())]);
        }
    }
    return out;
}

// symmetry.rs:18-68
template <class B, class N, class R = std::mt19937_64>
class RandomSymmetryNetwork : public Network<B> {
    N inner_;
    R rng_;
    bool enabled_;

  public:
    RandomSymmetryNetwork(N inner, R rng, bool enabled) : inner_(std::move(inner)), rng_(rng), enabled_(enabled) {}
    size_t max_batch_size() const override { return inner_.max_batch_size(); }
    N &inner() { return inner_; }
    std::vector<ZeroEvaluation> evaluate_batch(const B *boards, size_t n) override {
        if (!enabled_ || B::symmetry_is_unit()) return inner_.evaluate_batch(boards, n);  // :42-45
        std::vector<int> syms(n);
        std::vector<B> mapped;
        mapped.reserve(n);
        std::uniform_int_distribution<int> dist(0, B::symmetry_count() - 1);
        for (size_t i = 0; i < n; i++) {
            syms[i] = dist(rng_);
            mapped.push_back(boards[i].map(syms[i]));
        }
        auto mapped_evals = inner_.evaluate_batch(mapped.data(), n);
        std::vector<ZeroEvaluation> out;
        out.reserve(n);
        for (size_t i = 0; i < n; i++) out.push_back(unmap_eval(boards[i], syms[i], mapped[i], mapped_evals[i]));
        return out;
    }
};

// symmetry.rs:150-184: values summed and divided by the symmetry count; every available move's probability averaged over
// the symmetries (the probability the mapped board's evaluation gives the mapped move)
template <class B>
ZeroEvaluation average_evals(const B &board, const B *mapped_boards, const ZeroEvaluation *mapped_evals, int n_sym) {
    ZeroEvaluation out;
    const float n = (float)n_sym;
    for (int k = 0; k < n_sym; k++) {  // fold(default, a + b) / n (:156-160)
        const ZeroValuesPov &v = mapped_evals[k].values;
        out.values.value += v.value;
        out.values.wdl.win += v.wdl.win;
        out.values.wdl.draw += v.wdl.draw;
        out.values.wdl.loss += v.wdl.loss;
        out.values.moves_left += v.moves_left;
    }
    out.values.value /= n;
    out.values.wdl.win /= n;
    out.values.wdl.draw /= n;
    out.values.wdl.loss /= n;
    out.values.moves_left /= n;
    const size_t policy_len = n_sym ? mapped_evals[0].policy.size() : 0;
    out.policy.assign(policy_len, 0.0f);
    if (policy_len > 0) {
        const auto board_moves = board.available_moves();
        for (int k = 0; k < n_sym; k++) {
            const auto mapped_moves = mapped_boards[k].available_moves();
            for (size_t i = 0; i < board_moves->size(); i++) {
                const auto mapped_mv = board.map_move(k, (*board_moves)[i]);
                auto it = std::find(mapped_moves->begin(), mapped_moves->end(), mapped_mv);
                if (it == mapped_moves->end()) throw std::logic_error("mapped move not available on the mapped board");
                out.policy[i] += mapped_evals[k].policy[(size_t)(it - mapped_moves->begin())] / n;
            }
        }
    }
    return out;
}

// symmetry.rs:70-124: averages values and policy over ALL symmetries of every board; re-batches to the inner network's
// max_batch_size, so its own is unbounded
template <class B, class N>
class AverageSymmetryNetwork : public Network<B> {
    N inner_;

  public:
    explicit AverageSymmetryNetwork(N inner) : inner_(std::move(inner)) {}
    size_t max_batch_size() const override { return (size_t)-1; }  // usize::MAX (:88-92)
    N &inner() { return inner_; }
    std::vector<ZeroEvaluation> evaluate_batch(const B *boards, size_t n) override {
        if (B::symmetry_is_unit()) return inner_.evaluate_batch(boards, n);  // :95-98
        const int n_sym = B::symmetry_count();
        std::vector<B> mapped;
        mapped.reserve(n * (size_t)n_sym);
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < n_sym; k++) mapped.push_back(boards[i].map(k));
        std::vector<ZeroEvaluation> mapped_evals;
        mapped_evals.reserve(mapped.size());
        const size_t chunk = inner_.max_batch_size();
        for (size_t lo = 0; lo < mapped.size(); lo += chunk) {  // mapped_boards.chunks(inner.max_batch_size()) (:108-113)
            auto y = inner_.evaluate_batch(mapped.data() + lo, std::min(chunk, mapped.size() - lo));
            for (auto &e : y) mapped_evals.push_back(std::move(e));
        }
        std::vector<ZeroEvaluation> out;
        out.reserve(n);
        for (size_t i = 0; i < n; i++)
            out.push_back(average_evals(boards[i], mapped.data() + i * n_sym, mapped_evals.data() + i * n_sym, n_sym));
        return out;
    }
};

}  // namespace kz::host
