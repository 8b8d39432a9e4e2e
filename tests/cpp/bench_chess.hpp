// bench_chess.hpp — the host work of one chess evaluation as the Rust shim will do it, for tests/cpp/bench_executor.cpp:
//
//   * positions: random but plausible ChessPositions (a subset of the initial material on random squares);
//   * `available_moves()` GENERATES the moves at every call, like `ChessBoard::available_moves()` does in the reference
//     (a new `MoveGen` per call: board-game crate, ext).  The generator here is a pseudo-legal one over the bitboards
//     (knight / king tables, ray scans with blockers, pawn pushes, captures and the four promotions; no castling, no en
//     passant, no pin / check filtering): a stand-in with the same kind and amount of work (~25-35 moves per position),
//     NOT a rules engine — every move it emits is one of the 1880 flat POV moves, which is all the decode needs;
//   * `HashedChessMapper::move_to_index` looks the POV move up in a hash map like the reference
//     (rust/kz-core/src/mapping/chess.rs:202-210: `FLAT_MOVES_POV.mv_to_index.get(&mv_pov)`, a std `HashMap<ChessMove, usize>`,
//     i.e. SipHash-1-3 over the derived `Hash` of the move) instead of the mirror's direct table (host/mapping.hpp);
//   * encode = `ChessStdMapper::encode_input` over the bitboards (chess.rs:136-170).
#pragma once
#include <cstdint>
#include <optional>
#include <random>
#include <unordered_map>
#include <vector>

#include "../../kzero_amd/csrc/host/mapping.hpp"

namespace kz::bench {

using kz::host::ChessMove;
using kz::host::ChessPosition;

// ---- SipHash-1-3 with a zero key over a short message (what Rust's DefaultHasher runs per lookup) ----
inline uint64_t rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
inline uint64_t siphash13(const uint8_t *msg, size_t len) {
    uint64_t v0 = 0x736f6d6570736575ull, v1 = 0x646f72616e646f6dull, v2 = 0x6c7967656e657261ull, v3 = 0x7465646279746573ull;
    auto round = [&] {
        v0 += v1; v1 = rotl64(v1, 13); v1 ^= v0; v0 = rotl64(v0, 32);
        v2 += v3; v3 = rotl64(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl64(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl64(v1, 17); v1 ^= v2; v2 = rotl64(v2, 32);
    };
    size_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t m = 0;
        for (int k = 0; k < 8; k++) m |= (uint64_t)msg[i + k] << (8 * k);
        v3 ^= m;
        round();
        v0 ^= m;
    }
    uint64_t b = (uint64_t)len << 56;
    for (int k = 0; i + k < len; k++) b |= (uint64_t)msg[i + k] << (8 * k);
    v3 ^= b;
    round();
    v0 ^= b;
    v2 ^= 0xff;
    round();
    round();
    round();
    return v0 ^ v1 ^ v2 ^ v3;
}

struct MoveHash {
    size_t operator()(const ChessMove &m) const {
        // derive(Hash) of chess::ChessMove { source: Square(u8), dest: Square(u8), promotion: Option<Piece> }: two bytes, the
        // Option's discriminant as an isize, the piece's discriminant as an isize when present
        uint8_t msg[18] = {m.from, m.to};
        size_t len = 2;
        const uint64_t disc = m.promotion ? 1 : 0;
        for (int k = 0; k < 8; k++) msg[len++] = (uint8_t)(disc >> (8 * k));
        if (m.promotion) {
            const uint64_t piece = (uint64_t)m.promotion;
            for (int k = 0; k < 8; k++) msg[len++] = (uint8_t)(piece >> (8 * k));
        }
        return (size_t)siphash13(msg, len);
    }
};

struct FlatMoveMap {
    std::unordered_map<ChessMove, int32_t, MoveHash> mv_to_index;
    FlatMoveMap() {
        const auto &flat = kz::host::ChessFlatMoves::get();
        mv_to_index.reserve(flat.index_to_mv.size() * 2);
        for (size_t i = 0; i < flat.index_to_mv.size(); i++) mv_to_index.emplace(flat.index_to_mv[i], (int32_t)i);
    }
    static const FlatMoveMap &get() {
        static const FlatMoveMap m;
        return m;
    }
};

// ---- pseudo-legal move generation over the bitboards ----
struct AttackTables {
    uint64_t knight[64], king[64];
    AttackTables() {
        for (int sq = 0; sq < 64; sq++) {
            knight[sq] = king[sq] = 0;
            const int r = sq / 8, f = sq % 8;
            static const int kn[8][2] = {{2, 1}, {1, 2}, {-1, 2}, {-2, 1}, {-2, -1}, {-1, -2}, {1, -2}, {2, -1}};
            for (auto &d : kn)
                if (r + d[0] >= 0 && r + d[0] < 8 && f + d[1] >= 0 && f + d[1] < 8) knight[sq] |= 1ull << ((r + d[0]) * 8 + f + d[1]);
            for (int dr = -1; dr <= 1; dr++)
                for (int df = -1; df <= 1; df++)
                    if ((dr || df) && r + dr >= 0 && r + dr < 8 && f + df >= 0 && f + df < 8) king[sq] |= 1ull << ((r + dr) * 8 + f + df);
        }
    }
    static const AttackTables &get() {
        static const AttackTables t;
        return t;
    }
};

inline void pseudo_legal_moves(const ChessPosition &p, std::vector<ChessMove> &out) {
    const AttackTables &T = AttackTables::get();
    const int us = p.white_to_move ? 0 : 1, them = 1 - us;
    uint64_t own = 0, opp = 0;
    for (int k = 0; k < 6; k++) {
        own |= p.pieces[us][k];
        opp |= p.pieces[them][k];
    }
    const uint64_t occ = own | opp;
    auto emit_set = [&](int from, uint64_t targets) {
        while (targets) {
            const int to = __builtin_ctzll(targets);
            targets &= targets - 1;
            out.push_back(ChessMove{(uint8_t)from, (uint8_t)to, ChessMove::None});
        }
    };
    auto slide = [&](int from, const int (*dirs)[2], int ndirs) {
        const int r0 = from / 8, f0 = from % 8;
        for (int d = 0; d < ndirs; d++) {
            int r = r0 + dirs[d][0], f = f0 + dirs[d][1];
            while (r >= 0 && r < 8 && f >= 0 && f < 8) {
                const uint64_t bit = 1ull << (r * 8 + f);
                if (own & bit) break;
                out.push_back(ChessMove{(uint8_t)from, (uint8_t)(r * 8 + f), ChessMove::None});
                if (opp & bit) break;
                r += dirs[d][0];
                f += dirs[d][1];
            }
        }
    };
    static const int rook_dirs[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}}, bishop_dirs[4][2] = {{1, 1}, {1, -1}, {-1, 1}, {-1, -1}};
    static const int queen_dirs[8][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {1, -1}, {-1, 1}, {-1, -1}};
    // pawns: pushes, double pushes from the second rank, captures, promotions to the four pieces
    {
        uint64_t pawns = p.pieces[us][0];
        const int fwd = us == 0 ? 8 : -8, start_rank = us == 0 ? 1 : 6, last_rank = us == 0 ? 7 : 0;
        while (pawns) {
            const int from = __builtin_ctzll(pawns);
            pawns &= pawns - 1;
            auto emit_pawn = [&](int to) {
                if (to / 8 == last_rank)
                    for (int promo = ChessMove::Queen; promo <= ChessMove::Knight; promo++)
                        out.push_back(ChessMove{(uint8_t)from, (uint8_t)to, (int8_t)promo});
                else
                    out.push_back(ChessMove{(uint8_t)from, (uint8_t)to, ChessMove::None});
            };
            const int one = from + fwd;
            if (one >= 0 && one < 64 && !(occ >> one & 1)) {
                emit_pawn(one);
                const int two = one + fwd;
                if (from / 8 == start_rank && !(occ >> two & 1)) emit_pawn(two);
            }
            for (int df : {-1, 1}) {
                const int f = from % 8 + df, to = from + fwd + df;
                if (f >= 0 && f < 8 && to >= 0 && to < 64 && (opp >> to & 1)) emit_pawn(to);
            }
        }
    }
    for (uint64_t b = p.pieces[us][1]; b; b &= b - 1) emit_set(__builtin_ctzll(b), T.knight[__builtin_ctzll(b)] & ~own);
    for (uint64_t b = p.pieces[us][2]; b; b &= b - 1) slide(__builtin_ctzll(b), bishop_dirs, 4);
    for (uint64_t b = p.pieces[us][3]; b; b &= b - 1) slide(__builtin_ctzll(b), rook_dirs, 4);
    for (uint64_t b = p.pieces[us][4]; b; b &= b - 1) slide(__builtin_ctzll(b), queen_dirs, 8);
    for (uint64_t b = p.pieces[us][5]; b; b &= b - 1) emit_set(__builtin_ctzll(b), T.king[__builtin_ctzll(b)] & ~own);
}

// a plausible middle-game position: each side keeps its king and a random subset of its other fifteen pieces
template <class Rng>
ChessPosition random_position(Rng &rng) {
    ChessPosition p;
    p.white_to_move = rng() & 1;
    uint64_t occ = 0;
    auto place = [&](int color, int piece) {
        for (;;) {
            const int sq = (int)(rng() % 64);
            if (occ >> sq & 1) continue;
            if (piece == 0 && (sq / 8 == 0 || sq / 8 == 7)) continue;  // no pawns on the back ranks
            occ |= 1ull << sq;
            p.pieces[color][piece] |= 1ull << sq;
            return;
        }
    };
    static const int counts[6] = {8, 2, 2, 2, 1, 1};
    for (int color = 0; color < 2; color++)
        for (int piece = 0; piece < 6; piece++)
            for (int k = 0; k < counts[piece]; k++)
                if (piece == 5 || rng() % 100 < 50) place(color, piece);
    for (int color = 0; color < 2; color++) {
        p.castle_kingside[color] = rng() & 1;
        p.castle_queenside[color] = rng() & 1;
    }
    p.repetitions = (int)(rng() % 3);
    p.non_pawn_or_capture_moves = (int)(rng() % 100);
    return p;
}

// The board the generators send: `available_moves()` runs the move generator (as the reference's does at every call).
// `indices`, when a generator filled it in (move_to_index of every available move, in order), lets the executor thread
// skip both the move generation and the lookups: kz::host::HipNetwork::build_move_lists takes it as it is.
struct BenchChessBoard {
    using Move = ChessMove;
    ChessPosition pos;
    std::vector<int32_t> indices;
    bool has_indices = false;
    std::optional<std::vector<ChessMove>> available_moves() const {
        std::vector<ChessMove> mv;
        mv.reserve(48);
        pseudo_legal_moves(pos, mv);
        return mv;
    }
    const std::vector<int32_t> *policy_indices() const { return has_indices ? &indices : nullptr; }
};

struct HashedChessMapper {
    kz::host::ChessStdMapper inner;
    std::array<size_t, 3> input_bool_shape() const { return inner.input_bool_shape(); }
    size_t input_scalar_count() const { return inner.input_scalar_count(); }
    size_t policy_len() const { return inner.policy_len(); }
    void encode_input(kz::host::BitBuffer &bools, std::vector<float> &scalars, const BenchChessBoard &b) const {
        inner.encode_input(bools, scalars, b.pos);
    }
    size_t move_to_index(const BenchChessBoard &b, ChessMove mv) const {  // chess.rs:202-210
        const ChessMove pov = kz::host::chess_move_pov(b.pos.white_to_move, mv);
        const auto &map = FlatMoveMap::get().mv_to_index;
        auto it = map.find(pov);
        if (it == map.end()) throw std::invalid_argument("chess move not found in flat moves");
        return (size_t)it->second;
    }
};

}  // namespace kz::bench
