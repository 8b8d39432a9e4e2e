// mapping.hpp — board -> tensor encoding (rust/kz-core/src/mapping/{mod.rs, bit_buffer.rs, chess.rs, ataxx.rs, go.rs, ttt.rs, sttt.rs}).
//
// Game rules live in the ext `board-game` crate on the Rust side; this mirror defines the mappers over plain position
// records that carry exactly what the mappers read (bitboards, counters, the list of available moves).
#pragma once
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <vector>

namespace kz::host {

// bit_buffer.rs:4-96: LSB-first bit packing
class BitBuffer {
    std::vector<uint8_t> storage_;
    size_t capacity_, len_ = 0;

  public:
    explicit BitBuffer(size_t capacity) : storage_((capacity - 1) / 8 + 1, 0), capacity_(capacity) {}  // :8-14
    void push(bool b) {                                                                                  // :16-36
        if (len_ >= capacity_) throw std::out_of_range("BitBuffer: not enough space left, pushing 1");
        const size_t index = len_ / 8, bit = len_ % 8;
        len_++;
        if (b) storage_[index] |= (uint8_t)(1u << bit);
        else storage_[index] &= (uint8_t)~(1u << bit);
    }
    void push_block(uint64_t v) {  // :38-57: 8 little-endian bytes, byte-aligned only
        if (len_ + 64 > capacity_) throw std::out_of_range("BitBuffer: not enough space left, pushing 64");
        if (len_ % 8 != 0) throw std::logic_error("BitBuffer: can only push aligned blocks of bits");
        for (int i = 0; i < 8; i++) storage_[len_ / 8 + i] = (uint8_t)(v >> (8 * i));
        len_ += 64;
    }
    void clear() {  // :59-62
        std::fill(storage_.begin(), storage_.end(), 0);
        len_ = 0;
    }
    size_t len() const { return len_; }
    const std::vector<uint8_t> &storage() const { return storage_; }
    bool operator[](size_t i) const { return (storage_[i / 8] >> (i % 8)) & 1; }  // :80-90
};

// mapping/mod.rs:19-38 — InputMapper concept:
//     std::array<size_t,3> input_bool_shape() const;  size_t input_scalar_count() const;
//     void encode_input(BitBuffer&, std::vector<float>& scalars, const B&) const;
template <class M>
std::array<size_t, 3> input_full_shape(const M &m) {  // :23-27
    auto s = m.input_bool_shape();
    return {s[0] + m.input_scalar_count(), s[1], s[2]};
}
template <class M>
size_t input_bool_len(const M &m) {  // :29-31
    auto s = m.input_bool_shape();
    return s[0] * s[1] * s[2];
}

// mapping/mod.rs:40-63: scalar planes first (each scalar broadcast over w*h), then the bool planes as 0/1
template <class M, class B>
void encode_input_full(const M &m, std::vector<float> &result, const B &board) {
    const size_t bool_count = input_bool_len(m);
    auto shape = m.input_bool_shape();
    BitBuffer bools(bool_count);
    std::vector<float> scalars;
    m.encode_input(bools, scalars, board);
    if (bools.len() != bool_count || scalars.size() != m.input_scalar_count())
        throw std::logic_error("mapper wrote the wrong number of bools/scalars");
    for (float s : scalars) result.insert(result.end(), shape[1] * shape[2], s);
    for (size_t i = 0; i < bool_count; i++) result.push_back(bools[i] ? 1.0f : 0.0f);
}

// --------------------------------------------------------------------------------------------------------------
// Chess (chess.rs:125-178).  The record holds what ChessStdMapper reads from board-game's ChessBoard.
// --------------------------------------------------------------------------------------------------------------
// A chess move as the mappers see it (the ext `chess` crate's ChessMove): squares are rank * 8 + file, A1 = 0.
struct ChessMove {
    enum Promotion : int8_t { None = 0, Queen = 1, Rook = 2, Bishop = 3, Knight = 4 };  // order of chess.rs:489
    uint8_t from = 0, to = 0;
    int8_t promotion = None;
    bool operator==(const ChessMove &o) const { return from == o.from && to == o.to && promotion == o.promotion; }
};

// square_pov (chess.rs:397-406) and move_pov (chess.rs:409-416): black sees the board with the ranks flipped; its own inverse
inline uint8_t chess_square_pov(bool white_pov, uint8_t sq) { return white_pov ? sq : (uint8_t)((7 - sq / 8) * 8 + sq % 8); }
inline ChessMove chess_move_pov(bool white_pov, ChessMove mv) {
    return ChessMove{chess_square_pov(white_pov, mv.from), chess_square_pov(white_pov, mv.to), mv.promotion};
}

// generate_all_flat_moves_pov (chess.rs:439-481) and its inverse, the FLAT_MOVES_POV table (chess.rs:180-195): the 1880 moves from
// the point of view of the player making them — queen-like moves for every (from, to) in square order, the knight
// moves, then the promotions from the seventh to the eighth rank, piece-major.
struct ChessFlatMoves {
    static constexpr size_t COUNT = 1880;  // FLAT_MOVE_COUNT
    std::vector<ChessMove> index_to_mv;
    int16_t mv_to_index[64][64][5];
    ChessFlatMoves() {
        auto push = [&](int from, int to, int promo) {
            mv_to_index[from][to][promo] = (int16_t)index_to_mv.size();
            index_to_mv.push_back(ChessMove{(uint8_t)from, (uint8_t)to, (int8_t)promo});
        };
        for (auto &a : mv_to_index)
            for (auto &b : a)
                for (auto &c : b) c = -1;
        for (int from = 0; from < 64; from++)
            for (int to = 0; to < 64; to++) {
                const int df = from % 8 - to % 8, dr = from / 8 - to / 8;
                if (((df == 0) != (dr == 0)) || (df != 0 && std::abs(df) == std::abs(dr))) push(from, to, 0);
            }
        for (int from = 0; from < 64; from++)
            for (int to = 0; to < 64; to++) {
                const int df = std::abs(from % 8 - to % 8), dr = std::abs(from / 8 - to / 8);
                if ((df == 1 && dr == 2) || (df == 2 && dr == 1)) push(from, to, 0);
            }
        for (int piece = ChessMove::Queen; piece <= ChessMove::Knight; piece++)
            for (int from_f = 0; from_f < 8; from_f++)
                for (int to_f = 0; to_f < 8; to_f++)
                    if (std::abs(from_f - to_f) <= 1) push(6 * 8 + from_f, 7 * 8 + to_f, piece);
        if (index_to_mv.size() != COUNT) throw std::logic_error("flat chess move count");  // assert_eq!, :505
    }
    static const ChessFlatMoves &get() {
        static const ChessFlatMoves table;
        return table;
    }
};

// ClassifiedPovMove::{from_move, to_channel} (chess.rs:299-346) for a POV move: 8 directions x 7 distances (clockwise
// from N), 8 knight directions (clockwise from NNE), 3 x 3 under-promotions (direction-major; rook, bishop, knight).
// A queen promotion classifies as the queen-like move it is.  -1: "Could not find move type" (the reference panics).
inline int chess_conv_channel(ChessMove mv) {
    static const int queen_dirs[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
    static const int knight_deltas[8][2] = {{2, 1}, {1, 2}, {-1, 2}, {-2, 1}, {-2, -1}, {-1, -2}, {1, -2}, {2, -1}};
    const int dr = mv.to / 8 - mv.from / 8, df = mv.to % 8 - mv.from % 8;
    const int sr = (dr > 0) - (dr < 0), sf = (df > 0) - (df < 0);
    if (mv.promotion >= ChessMove::Rook) return 56 + 8 + (sf + 1) * 3 + (mv.promotion - ChessMove::Rook);
    for (int d = 0; d < 8; d++)
        if (queen_dirs[d][0] == sr && queen_dirs[d][1] == sf) {
            const int dist = std::max(std::abs(dr), std::abs(df));
            if (dr == sr * dist && df == sf * dist) return d * 7 + dist - 1;
        }
    for (int d = 0; d < 8; d++)
        if (knight_deltas[d][0] == dr && knight_deltas[d][1] == df) return 56 + d;
    return -1;
}

struct ChessPosition {
    using Move = ChessMove;  // move generation is ext (board-game crate): the caller supplies the available moves
    bool white_to_move = true;
    uint64_t pieces[2][6] = {};  // [color: 0 white, 1 black][pawn, knight, bishop, rook, queen, king], A1 = bit 0
    uint64_t en_passant = 0;     // bitboard with the en-passant square, or 0
    bool castle_kingside[2] = {}, castle_queenside[2] = {};  // [white, black]
    int repetitions = 0, non_pawn_or_capture_moves = 0;
    std::optional<std::vector<Move>> moves;
    std::optional<std::vector<Move>> available_moves() const { return moves; }
    // earlier positions of the game, oldest first (ChessBoard::history()), for ChessHistoryMapper: absolute colours
    struct Past {
        uint64_t pieces[2][6] = {};
        int repetitions = 0;  // repetitions_for(that board)
    };
    std::vector<Past> history;
};

// chess.rs:173-178: black sees the board with ranks flipped (BitBoard::reverse_colors = byte swap)
inline uint64_t pov_ranks(uint64_t board, bool white_pov) { return white_pov ? board : __builtin_bswap64(board); }

struct ChessStdMapper {
    std::array<size_t, 3> input_bool_shape() const { return {13, 8, 8}; }  // :126-129
    size_t input_scalar_count() const { return 8; }                        // :131-134
    size_t policy_len() const { return 1880; }
    // chess.rs:202-217: the index of the POV move in the flat list; an unknown move panics in the reference
    size_t move_to_index(const ChessPosition &b, ChessMove mv) const {
        const ChessMove pov = chess_move_pov(b.white_to_move, mv);
        const int index = pov.promotion >= 0 && pov.promotion <= ChessMove::Knight && pov.from < 64 && pov.to < 64
                              ? ChessFlatMoves::get().mv_to_index[pov.from][pov.to][pov.promotion]
                              : -1;
        if (index < 0) throw std::invalid_argument("chess move not found in flat moves");
        return (size_t)index;
    }
    ChessMove index_to_move(const ChessPosition &b, size_t index) const {
        return chess_move_pov(b.white_to_move, ChessFlatMoves::get().index_to_mv.at(index));
    }
    void encode_input(BitBuffer &bools, std::vector<float> &scalars, const ChessPosition &b) const {  // :136-170
        const int pov = b.white_to_move ? 0 : 1, other = 1 - pov;
        scalars.push_back(pov == 0 ? 1.0f : 0.0f);  // absolute colour of the side to move (:144-146)
        scalars.push_back(pov == 1 ? 1.0f : 0.0f);
        for (int color : {pov, other}) {  // castling rights, us then them (:149-153)
            scalars.push_back(b.castle_kingside[color] ? 1.0f : 0.0f);
            scalars.push_back(b.castle_queenside[color] ? 1.0f : 0.0f);
        }
        scalars.push_back((float)b.repetitions);  // counters (:156-157)
        scalars.push_back((float)b.non_pawn_or_capture_moves);
        for (int color : {pov, other})  // pieces, us then them (:160-165)
            for (int piece = 0; piece < 6; piece++) bools.push_block(pov_ranks(b.pieces[color][piece], b.white_to_move));
        bools.push_block(pov_ranks(b.en_passant, b.white_to_move));  // :168-169
    }
};

// ChessHistoryMapper (chess.rs:26-124; `Game::ChessHist { length }`, kz-selfplay/src/server/server.rs:162-173): the
// current board and the last `length` boards of the game.  Bool planes: en passant, then 12 piece planes per board
// (current first, then history newest to oldest, empty boards as zeros); scalars: side to move (white, black), castling
// rights (us K/Q, them K/Q), the 50-move counter, then 1 + repetitions per board (0 for an empty one).  Same policy as
// ChessStdMapper.
struct ChessHistoryMapper {
    size_t length;
    explicit ChessHistoryMapper(size_t length) : length(length) {}
    std::array<size_t, 3> input_bool_shape() const { return {1 + (length + 1) * 12, 8, 8}; }  // :33-36
    size_t input_scalar_count() const { return 7 + (length + 1); }                            // :38-40
    size_t policy_len() const { return 1880; }
    size_t move_to_index(const ChessPosition &b, ChessMove mv) const { return ChessStdMapper().move_to_index(b, mv); }
    ChessMove index_to_move(const ChessPosition &b, size_t index) const { return ChessStdMapper().index_to_move(b, index); }
    void encode_input(BitBuffer &bools, std::vector<float> &scalars, const ChessPosition &b) const {  // :42-81
        const int pov = b.white_to_move ? 0 : 1, other = 1 - pov;
        scalars.push_back(pov == 0 ? 1.0f : 0.0f);
        scalars.push_back(pov == 1 ? 1.0f : 0.0f);
        for (int color : {pov, other}) {
            scalars.push_back(b.castle_kingside[color] ? 1.0f : 0.0f);
            scalars.push_back(b.castle_queenside[color] ? 1.0f : 0.0f);
        }
        scalars.push_back((float)b.non_pawn_or_capture_moves);
        bools.push_block(pov_ranks(b.en_passant, b.white_to_move));
        auto append_board = [&](const uint64_t (&pieces)[2][6], int repetitions) {  // :83-97
            for (int color : {pov, other})
                for (int piece = 0; piece < 6; piece++) bools.push_block(pov_ranks(pieces[color][piece], b.white_to_move));
            scalars.push_back(1.0f + (float)repetitions);  // one more than the count: tells a real board from padding
        };
        append_board(b.pieces, b.repetitions);
        const size_t have = b.history.size(), used = std::min(length, have);
        for (size_t k = 0; k < used; k++) {  // newest first
            const ChessPosition::Past &h = b.history[have - 1 - k];
            append_board(h.pieces, h.repetitions);
        }
        for (size_t k = used; k < length; k++) {
            for (int i = 0; i < 12; i++) bools.push_block(0);
            scalars.push_back(0.0f);
        }
    }
};

// --------------------------------------------------------------------------------------------------------------
// Ataxx (ataxx.rs:8-116).  Tiles are indexed densely: i = y*size + x.
// --------------------------------------------------------------------------------------------------------------
// ChessLegacyConvPolicyMapper (chess.rs:219-247): policy [73, 8, 8], index = channel * 64 + POV source square
struct ChessLegacyConvPolicyMapper {
    size_t policy_len() const { return 73 * 64; }
    size_t move_to_index(const ChessPosition &b, ChessMove mv) const {
        const ChessMove pov = chess_move_pov(b.white_to_move, mv);
        const int channel = chess_conv_channel(pov);
        if (channel < 0) throw std::invalid_argument("could not find move type");
        return (size_t)channel * 64 + pov.from;
    }
};

struct AtaxxMove {
    enum Kind { Pass, Copy, Jump } kind = Pass;
    int from_x = 0, from_y = 0, to_x = 0, to_y = 0;
    bool operator==(const AtaxxMove &o) const {
        if (kind != o.kind) return false;
        if (kind == Pass) return true;
        if (to_x != o.to_x || to_y != o.to_y) return false;
        return kind == Copy || (from_x == o.from_x && from_y == o.from_y);
    }
};

struct AtaxxPosition {
    using Move = AtaxxMove;
    int size = 7;
    uint64_t tiles_next = 0, tiles_other = 0, gaps = 0;  // dense bit i = y*size + x; tiles_pov() order
    int moves_since_last_copy = 0;
    std::optional<std::vector<AtaxxMove>> moves;
    std::optional<std::vector<AtaxxMove>> available_moves() const { return moves; }
};

constexpr int ATAXX_MAX_MOVES_SINCE_LAST_COPY = 100;  // board_game::games::ataxx::MAX_MOVES_SINCE_LAST_COPY

// ataxx.rs:135-152
constexpr int ATAXX_FROM_DX_DY[16][2] = {{-2, -2}, {-1, -2}, {0, -2}, {1, -2}, {2, -2}, {-2, -1}, {2, -1}, {-2, 0},
                                         {2, 0},   {-2, 1},  {2, 1},  {-2, 2}, {-1, 2}, {0, 2},   {1, 2},  {2, 2}};

struct AtaxxStdMapper {
    int size;
    explicit AtaxxStdMapper(int size) : size(size) {}
    std::array<size_t, 3> input_bool_shape() const { return {3, (size_t)size, (size_t)size}; }  // :94-96
    size_t input_scalar_count() const { return 1; }                                             // :98-104
    size_t policy_len() const { return 17 * (size_t)size * size + 1; }                          // :23
    void encode_input(BitBuffer &bools, std::vector<float> &scalars, const AtaxxPosition &b) const {  // :106-115
        scalars.push_back((float)b.moves_since_last_copy / (float)ATAXX_MAX_MOVES_SINCE_LAST_COPY);
        const int area = size * size;
        for (int i = 0; i < area; i++) bools.push((b.tiles_next >> i) & 1);
        for (int i = 0; i < area; i++) bools.push((b.tiles_other >> i) & 1);
        for (int i = 0; i < area; i++) bools.push((b.gaps >> i) & 1);
    }
    // ataxx.rs:60-81
    size_t move_to_index(const AtaxxMove &mv) const {
        const size_t area = (size_t)size * size;
        switch (mv.kind) {
            case AtaxxMove::Pass: return 17 * area;
            case AtaxxMove::Copy: return (size_t)mv.to_y * size + mv.to_x;
            case AtaxxMove::Jump: {
                const int dx = mv.from_x - mv.to_x, dy = mv.from_y - mv.to_y;
                for (int i = 0; i < 16; i++)
                    if (ATAXX_FROM_DX_DY[i][0] == dx && ATAXX_FROM_DX_DY[i][1] == dy)
                        return (1 + (size_t)i) * area + (size_t)mv.to_y * size + mv.to_x;
                throw std::invalid_argument("not a jump of distance 2");
            }
        }
        return 0;
    }
    size_t move_to_index(const AtaxxPosition &, const AtaxxMove &mv) const { return move_to_index(mv); }
    // ataxx.rs:35-58
    std::optional<AtaxxMove> index_to_move(size_t index) const {
        const size_t area = (size_t)size * size;
        if (index >= policy_len()) throw std::out_of_range("policy index");
        const int to = (int)(index % area), tx = to % size, ty = to / size;
        if (index == policy_len() - 1) return AtaxxMove{AtaxxMove::Pass};
        if (index < area) return AtaxxMove{AtaxxMove::Copy, 0, 0, tx, ty};
        const size_t from_index = index / area - 1;
        const int fx = tx + ATAXX_FROM_DX_DY[from_index][0], fy = ty + ATAXX_FROM_DX_DY[from_index][1];
        if (fx < 0 || fx >= size || fy < 0 || fy >= size) return std::nullopt;
        return AtaxxMove{AtaxxMove::Jump, fx, fy, tx, ty};
    }
};

// --------------------------------------------------------------------------------------------------------------
// Tic-tac-toe and super tic-tac-toe (ttt.rs:9-55, sttt.rs:7-54): the server's `Game::TTT` / `Game::STTT`
// (rust/kz-selfplay/src/server/server.rs:114-137).  Tiles in the board-game crate's own coordinate order (`Coord3::all()`
// / `Coord::all()`: index 0..8 / o 0..80), which is also the policy index of a move.
// (`Game::ArimaaSplit`'s mapper, arimaa.rs:15-139, is not mirrored: its plane and policy order is that of `Piece::ALL`,
// `Direction::ALL` and `Square::index` of the ext crate arimaa_engine_step, which is not in the reference tree; the
// NETWORK side — 38 input planes, ArimaaPolicyHead — is complete: KZ_POLICY_ARIMAA.)
// --------------------------------------------------------------------------------------------------------------
template <int N>
struct TilesPosition {  // N = 9 (ttt) or 81 (sttt)
    using Move = int;   // Coord3::index() / Coord::o()
    std::array<uint8_t, N> next{}, other{};  // tile == Some(next_player()) / Some(next_player().other())
    std::array<uint8_t, N> available{};      // sttt only: is_available_move(c).unwrap_or(false) (all zero on a done board)
    std::optional<std::vector<int>> moves;
    std::optional<std::vector<int>> available_moves() const { return moves; }
};
using TTTPosition = TilesPosition<9>;
using STTTPosition = TilesPosition<81>;

struct TTTStdMapper {
    std::array<size_t, 3> input_bool_shape() const { return {2, 3, 3}; }  // ttt.rs:13-15
    size_t input_scalar_count() const { return 0; }                       // :17-19
    size_t policy_len() const { return 9; }                               // :28-30 ([1, 3, 3])
    void encode_input(BitBuffer &bools, std::vector<float> &, const TTTPosition &b) const {  // :21-24
        for (int i = 0; i < 9; i++) bools.push(b.next[i] != 0);
        for (int i = 0; i < 9; i++) bools.push(b.other[i] != 0);
    }
    size_t move_to_index(const TTTPosition &, int mv) const { return (size_t)mv; }  // :32-34
    std::optional<int> index_to_move(const TTTPosition &, size_t index) const {    // :36-39
        if (index >= 9) throw std::out_of_range("policy index");
        return (int)index;
    }
};

struct STTTStdMapper {
    std::array<size_t, 3> input_bool_shape() const { return {3, 9, 9}; }  // sttt.rs:11-13
    size_t input_scalar_count() const { return 0; }                       // :15-17
    size_t policy_len() const { return 81; }                              // :28-30 ([1, 9, 9])
    void encode_input(BitBuffer &bools, std::vector<float> &, const STTTPosition &b) const {  // :19-24
        for (int i = 0; i < 81; i++) bools.push(b.next[i] != 0);
        for (int i = 0; i < 81; i++) bools.push(b.other[i] != 0);
        for (int i = 0; i < 81; i++) bools.push(b.available[i] != 0);  // a done board: no available moves
    }
    size_t move_to_index(const STTTPosition &, int mv) const { return (size_t)mv; }  // :32-34
    std::optional<int> index_to_move(const STTTPosition &, size_t index) const {     // :36-39 (asserts index < 256: a u8)
        if (index >= 256) throw std::out_of_range("policy index");
        return (int)index;
    }
};

// --------------------------------------------------------------------------------------------------------------
// Go (go.rs:8-113).  Planes are over the max_size x max_size grid; tile i = y*max_size + x.
// --------------------------------------------------------------------------------------------------------------
struct GoPosition {
    using Move = int32_t;  // policy index: 0 = pass, 1 + tile (go.rs:26-31)
    int size = 19;
    std::vector<uint8_t> stones_next, stones_other;  // [max_area] 0/1
    std::vector<uint8_t> ko_illegal;                 // empty tiles that are not available (:73-77)
    std::vector<int8_t> territory;                   // [max_area]: 0 next player, 1 neither, 2 other (:79-87)
    bool next_is_black = true, pass_1 = false, pass_2 = false, multi_stone_suicide = false;
    float komi_pov = 7.5f;  // already from the next player's point of view (:91-94)
    std::optional<std::vector<Move>> moves;
    std::optional<std::vector<Move>> available_moves() const { return moves; }
};

struct GoStdMapper {
    int max_size;
    bool territory;
    GoStdMapper(int max_size, bool territory) : max_size(max_size), territory(territory) {}
    std::array<size_t, 3> input_bool_shape() const {  // :46-55
        return {(size_t)(4 + (territory ? 3 : 0)), (size_t)max_size, (size_t)max_size};
    }
    size_t input_scalar_count() const { return 6; }  // :57-62
    size_t policy_len() const { return 1 + (size_t)max_size * max_size; }
    size_t move_to_index(const GoPosition &, int32_t mv) const { return (size_t)mv; }
    // go.rs:26-43: pass = 0, Place(tile) = 1 + tile.to_flat(max_size) = 1 + y * max_size + x, and back
    struct Move {
        bool pass;
        int x, y;
    };
    size_t move_to_index(Move mv) const {
        if (mv.pass) return 0;
        if (mv.x < 0 || mv.y < 0 || mv.x >= max_size || mv.y >= max_size) throw std::invalid_argument("tile outside max_size");
        return 1 + (size_t)mv.y * max_size + mv.x;
    }
    Move index_to_move(size_t index) const {
        if (index == 0) return Move{true, 0, 0};
        const size_t tile = index - 1;
        if (tile >= (size_t)max_size * max_size) throw std::invalid_argument("tile_index < max_area");  // assert!, :39
        return Move{false, (int)(tile % max_size), (int)(tile / max_size)};
    }
    void encode_input(BitBuffer &bools, std::vector<float> &scalars, const GoPosition &b) const {  // :64-113
        const int area = max_size * max_size;
        auto exists = [&](int i) { return i % max_size < b.size && i / max_size < b.size; };
        for (int i = 0; i < area; i++) bools.push(exists(i) && b.stones_next[i]);
        for (int i = 0; i < area; i++) bools.push(exists(i) && b.stones_other[i]);
        for (int i = 0; i < area; i++) bools.push(exists(i));
        for (int i = 0; i < area; i++) bools.push(exists(i) && b.ko_illegal[i]);
        if (territory)
            for (int owner = 0; owner < 3; owner++)
                for (int i = 0; i < area; i++) bools.push(exists(i) && b.territory[i] == owner);
        scalars.push_back(b.next_is_black ? 1.0f : 0.0f);  // :105-110
        scalars.push_back(b.next_is_black ? 0.0f : 1.0f);
        scalars.push_back(b.pass_1 ? 1.0f : 0.0f);
        scalars.push_back(b.pass_2 ? 1.0f : 0.0f);
        scalars.push_back(b.komi_pov / 15.0f);
        scalars.push_back(b.multi_stone_suicide ? 1.0f : 0.0f);
    }
};

// --------------------------------------------------------------------------------------------------------------
// Already-packed boards: what crosses the FFI boundary (BitBuffer storage + scalars + available-move indices).
// --------------------------------------------------------------------------------------------------------------
struct PackedBoard {
    using Move = int32_t;
    std::vector<uint8_t> bits;
    std::vector<float> scalars;
    std::optional<std::vector<Move>> moves;
    std::optional<std::vector<Move>> available_moves() const { return moves; }
};

struct PackedMapper {
    size_t planes, h, w, n_scalar, n_policy;
    std::array<size_t, 3> input_bool_shape() const { return {planes, h, w}; }
    size_t input_scalar_count() const { return n_scalar; }
    size_t policy_len() const { return n_policy; }
    size_t move_to_index(const PackedBoard &, int32_t mv) const { return (size_t)mv; }
    void encode_input(BitBuffer &bools, std::vector<float> &scalars, const PackedBoard &b) const {
        const size_t n = planes * h * w;
        size_t i = 0;
        for (; i + 64 <= n; i += 64) {  // whole 64-bit blocks, like the chess mapper's push_block (chess.rs:160-169)
            uint64_t v = 0;
            for (int k = 0; k < 8; k++) v |= (uint64_t)b.bits[i / 8 + k] << (8 * k);
            bools.push_block(v);
        }
        for (; i < n; i++) bools.push((b.bits[i / 8] >> (i % 8)) & 1);
        scalars.insert(scalars.end(), b.scalars.begin(), b.scalars.end());
    }
};

}  // namespace kz::host
