// test_host.cpp — CPU tests of the host-side mirror (kzero_amd/csrc/host): BitBuffer, mappers, decode_output,
// job channel, batched_executor_loop, symmetry.  Built and run by tests/test_host_cpp.py.
//
// Modelled on the reference's own tests: rust/kz-core/src/mapping/bit_buffer.rs:112-164 (BitBuffer known answers),
// rust/kz-core/tests/mapper/mod.rs:13-82 (mapper shape validity, move <-> index round trip and uniqueness),
// rust/kz-core/tests/tree.rs (a fake uniform network drives the machinery).
#include <atomic>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <cstring>
#include <set>
#include <sstream>
#include <thread>
#include <tuple>

#include "../../kzero_amd/csrc/host/device_threads.hpp"
#include "../../kzero_amd/csrc/host/executor.hpp"
#include "../../kzero_amd/csrc/host/mapping.hpp"
#include "../../kzero_amd/csrc/host/network.hpp"
#include "../../kzero_amd/csrc/host/symmetry.hpp"

using namespace kz::host;

static int g_failed = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_failed++;                                                          \
        }                                                                        \
    } while (0)
template <class F>
static bool throws(F f) {
    try {
        f();
    } catch (...) {
        return true;
    }
    return false;
}

// ---- bit_buffer.rs:112-164 ----
static void test_bitbuffer() {
    {  // short
        BitBuffer b(8);
        b.push(true); b.push(false); b.push(true);
        CHECK(b.storage() == std::vector<uint8_t>{0b101});
    }
    {  // edge_length
        BitBuffer b(8);
        for (int i = 0; i < 8; i++) b.push(true);
        CHECK(b.storage() == std::vector<uint8_t>{0xFF});
        BitBuffer c(9);
        for (int i = 0; i < 9; i++) c.push(true);
        CHECK((c.storage() == std::vector<uint8_t>{0xFF, 0b1}));
    }
    {  // longer
        BitBuffer b(16);
        for (int i = 0; i < 16; i++) b.push(i == 1 || i == 5 || i == 12);
        CHECK((b.storage() == std::vector<uint8_t>{0b00100010, 0b10000}));
    }
    {  // overflow
        BitBuffer b(32);
        CHECK(throws([&] { for (int i = 0; i < 33; i++) b.push(false); }));
    }
    {  // block
        BitBuffer b(64);
        b.push_block(0b100000001);
        CHECK(b.storage()[0] == 1 && b.storage()[1] == 1 && b.len() == 64);
        for (int i = 2; i < 8; i++) CHECK(b.storage()[i] == 0);
        BitBuffer c(128);
        c.push(true);
        CHECK(throws([&] { c.push_block(1); }));  // only aligned blocks
    }
}

// ---- tests/mapper/mod.rs:13-82: shapes, scalars-first layout, move <-> index round trip ----
static void test_mappers() {
    {
        ChessStdMapper m;
        ChessPosition p;  // start position
        p.pieces[0][0] = 0xFF00ull; p.pieces[0][1] = 0x42; p.pieces[0][2] = 0x24; p.pieces[0][3] = 0x81;
        p.pieces[0][4] = 0x08; p.pieces[0][5] = 0x10;
        for (int i = 0; i < 6; i++) p.pieces[1][i] = __builtin_bswap64(p.pieces[0][i]);
        for (int c = 0; c < 2; c++) p.castle_kingside[c] = p.castle_queenside[c] = true;
        BitBuffer bools(input_bool_len(m));
        std::vector<float> scalars;
        m.encode_input(bools, scalars, p);
        CHECK(bools.len() == input_bool_len(m) && input_bool_len(m) == 13 * 64);  // mod.rs:21-27
        CHECK(scalars.size() == m.input_scalar_count());
        CHECK((scalars == std::vector<float>{1, 0, 1, 1, 1, 1, 0, 0}));
        CHECK(bools[0 * 64 + 8] && !bools[0 * 64 + 0]);   // our pawns on rank 2
        CHECK(bools[5 * 64 + 4]);                          // our king on e1
        CHECK(bools[6 * 64 + 48] && bools[11 * 64 + 60]);  // their pawns on rank 7, their king on e8
        // black to move sees the same picture from its side (pov_ranks, chess.rs:173-178)
        ChessPosition q = p;
        q.white_to_move = false;
        BitBuffer bools2(input_bool_len(m));
        std::vector<float> scalars2;
        m.encode_input(bools2, scalars2, q);
        CHECK(bools2.storage() == bools.storage());
        CHECK(scalars2[0] == 0 && scalars2[1] == 1);
        // encode_input_full: scalar planes first, then bools (mapping/mod.rs:54-59)
        std::vector<float> full;
        encode_input_full(m, full, p);
        CHECK(full.size() == 21 * 64);
        CHECK(full[0] == 1 && full[63] == 1 && full[64] == 0 && full[2 * 64 + 5] == 1);
        CHECK(full[8 * 64 + 8] == 1 && full[8 * 64 + 0] == 0);
    }
    {
        // Known-answer planes for ChessStdMapper::encode_input (chess.rs:125-171), derived BY HAND from three FENs: plane
        // order = us P N B R Q K, them P N B R Q K, en passant; bit = square index (A1 = 0, file fastest); black to move
        // sees every bitboard with the ranks flipped (pov_ranks, chess.rs:173-178); scalars = [pov white, pov black,
        // castle us K, us Q, them K, them Q, repetitions, non-pawn-or-capture moves] (:136-155).
        auto from_fen = [](const char *fen) {  // piece placement, side, castling and half-move clock of a FEN
            ChessPosition p;
            int rank = 7, file = 0;
            const char *c = fen;
            for (; *c != ' '; c++) {
                if (*c == '/') { rank--; file = 0; continue; }
                if (*c >= '1' && *c <= '8') { file += *c - '0'; continue; }
                const char *kinds = "pnbrqk";
                const int color = (*c >= 'a') ? 1 : 0;
                const int piece = (int)(strchr(kinds, (char)tolower(*c)) - kinds);
                p.pieces[color][piece] |= 1ull << (rank * 8 + file);
                file++;
            }
            c++;
            p.white_to_move = *c == 'w';
            c += 2;
            for (; *c != ' '; c++) {
                if (*c == 'K') p.castle_kingside[0] = true;
                if (*c == 'Q') p.castle_queenside[0] = true;
                if (*c == 'k') p.castle_kingside[1] = true;
                if (*c == 'q') p.castle_queenside[1] = true;
            }
            c++;
            while (*c != ' ') c++;  // the en-passant field: set by the caller (what `inner.en_passant()` reports)
            p.non_pawn_or_capture_moves = atoi(c + 1);
            return p;
        };
        auto planes_of = [](const ChessPosition &p, std::vector<float> &scalars) {
            ChessStdMapper m;
            BitBuffer bools(input_bool_len(m));
            m.encode_input(bools, scalars, p);
            std::vector<std::vector<int>> planes(13);
            for (int pl = 0; pl < 13; pl++)
                for (int sq = 0; sq < 64; sq++)
                    if (bools[pl * 64 + sq]) planes[pl].push_back(sq);
            return planes;
        };
        using P = std::vector<std::vector<int>>;
        {   // Ruy Lopez after 3...a6, white to move
            auto p = from_fen("r1bqkbnr/1ppp1ppp/p1n5/1B2p3/4P3/5N2/PPPP1PPP/RNBQK2R w KQkq - 0 4");
            std::vector<float> sc;
            const P got = planes_of(p, sc);
            const P want = {{8, 9, 10, 11, 13, 14, 15, 28}, {1, 21}, {2, 33}, {0, 7}, {3}, {4},
                            {36, 40, 49, 50, 51, 53, 54, 55}, {42, 62}, {58, 61}, {56, 63}, {59}, {60}, {}};
            CHECK(got == want);
            CHECK((sc == std::vector<float>{1, 0, 1, 1, 1, 1, 0, 0}));
        }
        {   // French advance after 2...d5: the chess crate reports the capturable PAWN's square (d5 = 35) as en passant
            auto p = from_fen("rnbqkbnr/ppp2ppp/4p3/3pP3/8/8/PPPP1PPP/RNBQKBNR w KQkq d6 0 3");
            p.en_passant = 1ull << 35;
            std::vector<float> sc;
            const P got = planes_of(p, sc);
            const P want = {{8, 9, 10, 11, 13, 14, 15, 36}, {1, 6}, {2, 5}, {0, 7}, {3}, {4},
                            {35, 44, 48, 49, 50, 53, 54, 55}, {57, 62}, {58, 61}, {56, 63}, {59}, {60}, {35}};
            CHECK(got == want);
            CHECK((sc == std::vector<float>{1, 0, 1, 1, 1, 1, 0, 0}));
        }
        {   // 1. e4 e5 2. Ke2: BLACK to move, white has lost both castling rights, one reversible half-move
            auto p = from_fen("rnbqkbnr/pppp1ppp/8/4p3/4P3/8/PPPPKPPP/RNBQ1BNR b kq - 1 2");
            p.repetitions = 0;
            std::vector<float> sc;
            const P got = planes_of(p, sc);
            // "us" = black, seen with the ranks flipped: rank 8 -> rank 1, e5 (36) -> e4 (28), e4 (28) -> e5 (36), e2 -> e7
            const P want = {{8, 9, 10, 11, 13, 14, 15, 28}, {1, 6}, {2, 5}, {0, 7}, {3}, {4},
                            {36, 48, 49, 50, 51, 53, 54, 55}, {57, 62}, {58, 61}, {56, 63}, {59}, {52}, {}};
            CHECK(got == want);
            CHECK((sc == std::vector<float>{0, 1, 1, 1, 0, 0, 0, 1}));
        }
    }
    {
        // Known answers for AtaxxStdMapper::encode_input (ataxx.rs:106-116) and GoStdMapper::encode_input (go.rs:64-113),
        // derived by hand from position strings.  Ataxx 7x7 start "x5o/7/7/7/7/7/o5x x" with one gap at d4: the board-game
        // crate's rank 1 is y = 0, so x (to move) owns a7 = (0,6) and g1 = (6,0), o owns g7 = (6,6) and a1 = (0,0); planes =
        // next player's tiles, other player's tiles, gaps, each over full_mask() in y-major order; scalar = 30 / 100.
        AtaxxStdMapper am(7);
        AtaxxPosition ap;
        ap.size = 7;
        ap.tiles_next = (1ull << (6 * 7 + 0)) | (1ull << (0 * 7 + 6));
        ap.tiles_other = (1ull << (6 * 7 + 6)) | (1ull << (0 * 7 + 0));
        ap.gaps = 1ull << (3 * 7 + 3);
        ap.moves_since_last_copy = 30;
        BitBuffer ab(input_bool_len(am));
        std::vector<float> as;
        am.encode_input(ab, as, ap);
        std::vector<int> on;
        for (size_t i = 0; i < ab.len(); i++) if (ab[i]) on.push_back((int)i);
        CHECK((on == std::vector<int>{6, 42, 49 + 0, 49 + 48, 98 + 24}));
        CHECK(as.size() == 1 && std::fabs(as[0] - 0.3f) < 1e-7f);
        // TTTStdMapper / STTTStdMapper (ttt.rs:21-24, sttt.rs:19-24): plane order (mover, other[, available]), tiles in the
        // crate's coordinate order; a move's policy index is its coordinate index; what test_valid_mapping
        // (rust/kz-core/tests/mapper/mod.rs:13-30) checks: lengths, and every available move round-trips to itself with no
        // two moves on one index
        {
            TTTStdMapper tm;
            TTTPosition tp;
            tp.next[0] = tp.next[4] = 1;
            tp.other[8] = 1;
            tp.moves = std::vector<int>{1, 2, 3, 5, 6, 7};
            BitBuffer tb(input_bool_len(tm));
            std::vector<float> ts;
            tm.encode_input(tb, ts, tp);
            CHECK(tb.len() == 18 && ts.empty() && input_full_shape(tm) == (std::array<size_t, 3>{2, 3, 3}));
            std::vector<int> on;
            for (size_t i = 0; i < tb.len(); i++) if (tb[i]) on.push_back((int)i);
            CHECK((on == std::vector<int>{0, 4, 9 + 8}));
            std::set<size_t> seen;
            for (int mv : *tp.moves) {
                const size_t idx = tm.move_to_index(tp, mv);
                CHECK(idx < tm.policy_len() && seen.insert(idx).second && tm.index_to_move(tp, idx) == mv);
            }
            STTTStdMapper sm;
            STTTPosition sp;
            sp.next[40] = 1;
            sp.other[0] = sp.other[80] = 1;
            for (int i = 36; i < 45; i++) if (i != 40) sp.available[i] = 1;  // the centre macro board
            sp.moves = std::vector<int>{36, 37, 38, 39, 41, 42, 43, 44};
            BitBuffer sb(input_bool_len(sm));
            sm.encode_input(sb, ts, sp);
            CHECK(sb.len() == 243 && ts.empty() && sm.policy_len() == 81);
            on.clear();
            for (size_t i = 0; i < sb.len(); i++) if (sb[i]) on.push_back((int)i);
            CHECK((on == std::vector<int>{40, 81, 81 + 80, 162 + 36, 162 + 37, 162 + 38, 162 + 39, 162 + 41, 162 + 42, 162 + 43, 162 + 44}));
            seen.clear();
            for (int mv : *sp.moves) {
                const size_t idx = sm.move_to_index(sp, mv);
                CHECK(idx < sm.policy_len() && seen.insert(idx).second && sm.index_to_move(sp, idx) == mv);
            }
            STTTPosition done;  // a finished game: no available plane bits, no moves, an empty policy
            BitBuffer db(input_bool_len(sm));
            sm.encode_input(db, ts, done);
            bool any = false;
            for (size_t i = 162; i < 243; i++) any |= db[i];
            CHECK(!any && !done.available_moves().has_value());
        }
        // Go 5x5 inside 9x9 planes, white (Player::B) to move after a black pass, komi 6.5 for black: planes = stones of
        // the player to move, stones of the other, in-board, ko/illegal, territory (mover, neither, other); scalars =
        // [black to move, white to move, pass_1, pass_2, komi from the mover's side / 15, multi-stone suicide] (go.rs:89-112)
        GoStdMapper gm(9, true);
        GoPosition gp;
        gp.size = 5;
        gp.stones_next.assign(81, 0); gp.stones_other.assign(81, 0); gp.ko_illegal.assign(81, 0); gp.territory.assign(81, 1);
        gp.stones_next[1 * 9 + 1] = 1;            // white stone at (1,1)
        gp.stones_other[2 * 9 + 2] = 1;           // black stone at (2,2)
        gp.ko_illegal[1 * 9 + 2] = 1;             // an empty point white may not play
        gp.territory[0] = 0; gp.territory[4 * 9 + 4] = 2;
        gp.next_is_black = false; gp.pass_1 = true; gp.komi_pov = -6.5f;
        BitBuffer gb(input_bool_len(gm));
        std::vector<float> gs;
        gm.encode_input(gb, gs, gp);
        CHECK(gb[0 * 81 + 10] && gb[1 * 81 + 20] && gb[3 * 81 + 11]);
        int in_board = 0, own_next = 0, own_none = 0, own_other = 0;
        for (int i = 0; i < 81; i++) {
            in_board += gb[2 * 81 + i]; own_next += gb[4 * 81 + i]; own_none += gb[5 * 81 + i]; own_other += gb[6 * 81 + i];
            CHECK(gb[2 * 81 + i] == (i % 9 < 5 && i / 9 < 5));
        }
        CHECK(in_board == 25 && own_next == 1 && own_other == 1 && gb[4 * 81 + 0] && gb[6 * 81 + 40]);
        CHECK(gs.size() == 6 && gs[0] == 0 && gs[1] == 1 && gs[2] == 1 && gs[3] == 0 && gs[5] == 0);
        CHECK(std::fabs(gs[4] - (-6.5f / 15.0f)) < 1e-7f);
    }
    {  // ChessHistoryMapper (chess.rs:26-124): shapes for every length (tests/mapper/chess/pairs.rs:366-368), contents
        ChessPosition p;
        p.pieces[0][0] = 0xFF00ull; p.pieces[0][5] = 0x10;
        p.pieces[1][0] = 0xFFull << 48; p.pieces[1][5] = 0x10ull << 56;
        p.repetitions = 1;
        p.non_pawn_or_capture_moves = 3;
        p.en_passant = 1ull << 20;
        ChessPosition::Past older, newer;
        older.pieces[0][0] = 0x1; older.repetitions = 2;
        newer.pieces[0][0] = 0x2; newer.repetitions = 0;
        p.history = {older, newer};
        for (size_t length : {0, 1, 2, 8}) {
            ChessHistoryMapper m(length);
            BitBuffer bools(input_bool_len(m));
            std::vector<float> scalars;
            m.encode_input(bools, scalars, p);
            CHECK(bools.len() == input_bool_len(m) && input_bool_len(m) == (1 + (length + 1) * 12) * 64);
            CHECK(scalars.size() == m.input_scalar_count() && scalars.size() == 7 + length + 1);
            CHECK((std::vector<float>(scalars.begin(), scalars.begin() + 8) == std::vector<float>{1, 0, 0, 0, 0, 0, 3, 2}));
            CHECK(bools[20] && !bools[21]);                  // plane 0: en passant
            CHECK(bools[64 + 8] && bools[64 + 5 * 64 + 4]);  // current board: our pawns, our king
            if (length >= 1) CHECK(bools[13 * 64 + 1] && !bools[13 * 64 + 0] && scalars[8] == 1.0f);      // newest past board first
            if (length >= 2) CHECK(bools[25 * 64 + 0] && scalars[9] == 3.0f);                             // then the older one
            if (length == 8) CHECK(!bools[37 * 64 + 0] && scalars[10] == 0.0f && scalars.back() == 0.0f);  // padding
            CHECK(m.move_to_index(p, ChessMove{12, 28, 0}) == ChessStdMapper().move_to_index(p, ChessMove{12, 28, 0}));
        }
        // black to move: every board, past ones included, is seen with the ranks flipped
        ChessPosition q = p;
        q.white_to_move = false;
        ChessHistoryMapper m(1);
        BitBuffer bools(input_bool_len(m));
        std::vector<float> scalars;
        m.encode_input(bools, scalars, q);
        CHECK(scalars[0] == 0 && scalars[1] == 1);
        CHECK(bools[64 + 0 * 64 + 8] && !bools[64 + 0 * 64 + 48]);  // black's pawns (rank 7) appear as "our" pawns on rank 2
        CHECK(bools[64 + 6 * 64 + 48]);                              // white's pawns (rank 2) as "their" pawns on rank 7
        CHECK(bools[13 * 64 + 6 * 64 + 57]);  // the newer past board's white pawn on b1 is "theirs", seen on b8
    }
    for (int size = 2; size <= 8; size++) {  // ataxx: every index maps to a unique move and back (mod.rs:37-72)
        AtaxxStdMapper m(size);
        std::set<size_t> seen;
        size_t valid = 0;
        for (size_t i = 0; i < m.policy_len(); i++) {
            auto mv = m.index_to_move(i);
            if (!mv) continue;
            valid++;
            CHECK(m.move_to_index(*mv) == i);
            seen.insert(i);
        }
        CHECK(seen.size() == valid);
        // python/lib/mapping/mapping.py:52: number of valid moves per size
        static const size_t expect[] = {5, 42, 113, 218, 357, 530, 737};
        CHECK(valid == expect[size - 2]);
        AtaxxPosition p;
        p.size = size;
        p.tiles_next = 1;
        p.tiles_other = 1ull << (size * size - 1);
        p.moves_since_last_copy = 50;
        BitBuffer bools(input_bool_len(m));
        std::vector<float> scalars;
        m.encode_input(bools, scalars, p);
        CHECK(bools.len() == (size_t)3 * size * size && scalars.size() == 1 && scalars[0] == 0.5f);
        CHECK(bools[0] && bools[(size_t)2 * size * size - 1] && !bools[1]);
    }
    {
        GoStdMapper m(9, true);
        GoPosition p;
        p.size = 7;  // smaller board inside the 9x9 planes
        p.stones_next.assign(81, 0); p.stones_other.assign(81, 0); p.ko_illegal.assign(81, 0); p.territory.assign(81, 1);
        p.stones_next[3] = 1;
        p.stones_next[8] = 1;  // x = 8 is outside a size-7 board: masked by `exists`
        BitBuffer bools(input_bool_len(m));
        std::vector<float> scalars;
        m.encode_input(bools, scalars, p);
        CHECK(bools.len() == 7 * 81 && scalars.size() == 6);
        CHECK(bools[3] && !bools[8]);
        CHECK(bools[2 * 81 + 0] && !bools[2 * 81 + 7] && !bools[2 * 81 + 7 * 9]);  // in-board plane
        CHECK(std::fabs(scalars[4] - 0.5f) < 1e-6f);
        CHECK(GoStdMapper(19, true).policy_len() == 362 && input_full_shape(GoStdMapper(19, true))[0] == 13);
        // policy round trip (tests/mapper/mod.rs:37-72): pass = 0, tiles row-major after it
        GoStdMapper g19(19, true);
        for (size_t i = 0; i < g19.policy_len(); i++) CHECK(g19.move_to_index(g19.index_to_move(i)) == i);
        CHECK(g19.index_to_move(0).pass && g19.move_to_index({false, 3, 2}) == 1 + 2 * 19 + 3);
        CHECK(throws([&] { g19.index_to_move(362); }));
    }
}

// ---- the other adapters behind the Network contract (dummy.rs, multibatch.rs) ----
struct CountingNet : Network<PackedBoard> {  // value = batch size it was called with, policy = 1, 2, 3, ... per move
    size_t cap;
    int *calls;
    CountingNet(size_t cap, int *calls) : cap(cap), calls(calls) {}
    size_t max_batch_size() const override { return cap; }
    std::vector<ZeroEvaluation> evaluate_batch(const PackedBoard *boards, size_t n) override {
        if (calls) (*calls)++;
        std::vector<ZeroEvaluation> out(n);
        for (size_t i = 0; i < n; i++) {
            out[i].values = ZeroValuesPov{(float)cap, WDL{1, 0, 0}, 7.0f};
            const size_t moves = boards[i].moves ? boards[i].moves->size() : 0;
            for (size_t k = 0; k < moves; k++) out[i].policy.push_back((float)(k + 1));
        }
        return out;
    }
};
struct WrappedBoard {  // MaxMovesBoard<B>: a board with a move counter around an inner board
    PackedBoard b;
    int moves_played = 0;
    const PackedBoard &inner() const { return b; }
};

static void test_adapters() {
    PackedBoard three, none;
    three.moves = std::vector<int32_t>{4, 5, 6};
    std::vector<PackedBoard> boards{three, none, three};
    {
        DummyValueNetwork<PackedBoard, CountingNet> dv(CountingNet(8, nullptr));
        auto y = dv.evaluate_batch(boards.data(), boards.size());
        CHECK(y.size() == 3 && y[0].values.value == 0.0f && std::fabs(y[0].values.wdl.win - 1.0f / 3) < 1e-7f);
        CHECK((y[0].policy == std::vector<float>{1, 2, 3}) && y[1].policy.empty());
        DummyPolicyNetwork<PackedBoard, CountingNet> dp(CountingNet(8, nullptr));
        auto z = dp.evaluate_batch(boards.data(), boards.size());
        CHECK(z[0].values.value == 8.0f && z[0].values.moves_left == 7.0f);
        CHECK(z[0].policy.size() == 3 && std::fabs(z[0].policy[1] - 1.0f / 3) < 1e-7f && z[1].policy.empty());
        CHECK(dp.max_batch_size() == 8);
    }
    {
        std::vector<WrappedBoard> wrapped{{three, 5}, {none, 9}};
        MaxMovesNetwork<WrappedBoard, PackedBoard, CountingNet> mm(CountingNet(4, nullptr));
        auto y = mm.evaluate_batch(wrapped.data(), wrapped.size());
        CHECK(y.size() == 2 && y[0].policy.size() == 3 && y[1].policy.empty() && mm.max_batch_size() == 4);
    }
    {
        using E = EitherNetwork<PackedBoard, CountingNet, DummyNetwork<PackedBoard>>;
        E left(CountingNet(8, nullptr));
        E right(E::RightTag{}, DummyNetwork<PackedBoard>{});
        CHECK(left.max_batch_size() == 8 && left.evaluate_batch(boards.data(), 3)[0].values.value == 8.0f);
        auto y = right.evaluate_batch(boards.data(), 3);  // "UseDummyNetwork": uniform everything
        CHECK(y[0].values.value == 0.0f && std::fabs(y[0].policy[2] - 1.0f / 3) < 1e-7f);
        CHECK(right.max_batch_size() == std::numeric_limits<size_t>::max());
    }
    {
        int calls[3] = {0, 0, 0};
        int which = 0;
        auto mb = MultiBatchNetwork<PackedBoard, CountingNet>::build_sizes({16, 2, 4}, [&](size_t s) { return CountingNet(s, &calls[which++]); });
        CHECK(mb.max_batch_size() == 16);
        CHECK(mb.used_batch_size(1) == 2 && mb.used_batch_size(2) == 2 && mb.used_batch_size(3) == 4 && mb.used_batch_size(5) == 16);
        CHECK(mb.evaluate_batch(boards.data(), 3)[0].values.value == 4.0f);  // the smallest instance that fits 3 boards
        CHECK(calls[0] == 0 && calls[1] == 0 && calls[2] == 1);
        CHECK(throws([&] { mb.used_network_index(17); }));  // "No network for batch size"
    }
}

// ---- chess policy indexing: chess.rs:180-507, pinned to the reference's own data ----
//  * python/lib/mapping/chess_flat_to_move_input.txt / chess_flat_to_conv.txt / chess_flat_to_att.txt (written by
//    rust/kz-misc/src/bin/write_chess_mapping.rs from generate_all_flat_moves_pov): every one of the 1880 flat moves,
//    its conv-policy index and its attention index;
//  * the (side, move) <-> conv index vectors of rust/kz-core/tests/mapper/chess/pairs.rs (oracle/gen_chess_pairs.py);
//  * flat_gen (tests/mapper/chess/mod.rs:6-17): 1880 moves, no duplicates;
//  * test_valid_policy_mapping and the two checks it runs (tests/mapper/mod.rs:29-82): index -> move -> index round trip for both colours.
static void test_chess_policy(const std::string &golden) {
    const auto &flat = ChessFlatMoves::get();
    CHECK(flat.index_to_mv.size() == 1880);
    std::set<std::tuple<int, int, int>> unique;
    for (auto &m : flat.index_to_mv) unique.insert({m.from, m.to, m.promotion});
    CHECK(unique.size() == 1880);

    std::ifstream fin(golden + "/chess_flat_to_move_input.txt"), fconv(golden + "/chess_flat_to_conv.txt"),
        fatt(golden + "/chess_flat_to_att.txt");
    CHECK(fin.good() && fconv.good() && fatt.good());
    ChessStdMapper std_mapper;
    ChessLegacyConvPolicyMapper conv_mapper;
    ChessPosition white, black;
    black.white_to_move = false;
    std::string line;
    size_t rows = 0;
    for (size_t i = 0; i < 1880 && std::getline(fin, line); i++, rows++) {
        int v[8];
        std::stringstream ss(line);
        for (int k = 0; k < 8; k++) {
            std::string cell;
            std::getline(ss, cell, ',');
            v[k] = std::atoi(cell.c_str());
        }
        const int promo = v[3] ? ChessMove::Queen : v[4] ? ChessMove::Rook : v[5] ? ChessMove::Bishop : v[6] ? ChessMove::Knight : ChessMove::None;
        const ChessMove mv = flat.index_to_mv[i];
        CHECK(mv.from == v[0] && mv.to == v[1] && mv.promotion == promo && (v[7] == 1) == (promo == ChessMove::None));
        int conv = -1, att = -1;
        fconv >> conv;
        fatt >> att;
        CHECK((int)conv_mapper.move_to_index(white, mv) == conv);
        // write_chess_mapping.rs:51-66
        const int att_to = mv.promotion == ChessMove::None ? mv.to : 64 + (mv.to % 8) * 3 + (mv.promotion - 1);
        CHECK(mv.from * 88 + att_to == att);
        // round trips: white sees the list as is, black with the ranks flipped
        CHECK(std_mapper.move_to_index(white, mv) == i && std_mapper.index_to_move(white, i) == mv);
        const ChessMove abs_black = std_mapper.index_to_move(black, i);
        CHECK(abs_black.from == (7 - mv.from / 8) * 8 + mv.from % 8 && abs_black.to == (7 - mv.to / 8) * 8 + mv.to % 8);
        CHECK(std_mapper.move_to_index(black, abs_black) == i);
    }
    CHECK(rows == 1880);
    CHECK(throws([&] { std_mapper.move_to_index(white, ChessMove{0, 0, 0}); }));                  // a1a1: not a move
    CHECK(throws([&] { std_mapper.move_to_index(white, ChessMove{8, 16, ChessMove::Queen}); }));  // a2a3=Q: not one either

    std::ifstream pairs(golden + "/chess_conv_pairs.txt");
    CHECK(pairs.good());
    int side, from, to, promo, index, n_pairs = 0;
    while (pairs >> side >> from >> to >> promo >> index) {
        ChessPosition pos;
        pos.white_to_move = side != 0;
        CHECK((int)conv_mapper.move_to_index(pos, ChessMove{(uint8_t)from, (uint8_t)to, (int8_t)promo}) == index);
        n_pairs++;
    }
    CHECK(n_pairs == 84);

    // decode_output gathers the logits of the available moves through move_to_index (common.rs:77-86)
    ChessPosition p;
    p.moves = std::vector<ChessMove>{{12, 28, 0}, {6, 21, 0}};  // e2e4, g1f3
    std::vector<float> scalars(5, 0.0f), logits(1880, 0.0f);
    logits[std_mapper.move_to_index(p, {12, 28, 0})] = std::log(3.0f);
    auto ev = decode_output(std_mapper, &p, 1, scalars.data(), logits.data());
    CHECK(ev.size() == 1 && ev[0].policy.size() == 2);
    CHECK(std::fabs(ev[0].policy[0] - 0.75f) < 1e-6f && std::fabs(ev[0].policy[1] - 0.25f) < 1e-6f);
}

// ---- network/common.rs:16-114 ----
static void test_decode() {
    PackedMapper m{1, 2, 2, 0, 6};
    PackedBoard boards[3];
    boards[0].moves = std::vector<int32_t>{4, 0, 5};
    boards[1].moves = std::vector<int32_t>{2};  // a single legal move gets probability 1
    boards[2].moves = std::nullopt;  // finished game: empty policy (map_or(vec![], ..))
    const float scalars[15] = {0.5f, 1, 2, 3, 7, -2, 0, 0, 0, 1.5f, 0, 5, 5, 5, -1};
    const float logits[18] = {1, 9, 9, 9, 2, 3, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6};
    auto ev = decode_output(m, boards, 3, scalars, logits);
    CHECK(ev.size() == 3);
    CHECK(std::fabs(ev[0].values.value - std::tanh(0.5f)) < 1e-7f && ev[0].values.moves_left == 7);
    const float e1 = std::exp(-2.f), e2 = std::exp(-1.f), es = e1 + e2 + 1;
    CHECK(std::fabs(ev[0].values.wdl.win - e1 / es) < 1e-6f && std::fabs(ev[0].values.wdl.loss - 1 / es) < 1e-6f);
    // policy: logits gathered in available_moves order {4, 0, 5} -> {2, 1, 3}, then softmax
    const float p0 = std::exp(2.f - 3), p1 = std::exp(1.f - 3), p2 = 1, ps = p0 + p1 + p2;
    CHECK(ev[0].policy.size() == 3 && std::fabs(ev[0].policy[0] - p0 / ps) < 1e-6f &&
          std::fabs(ev[0].policy[1] - p1 / ps) < 1e-6f && std::fabs(ev[0].policy[2] - p2 / ps) < 1e-6f);
    CHECK(ev[1].policy.size() == 1 && ev[1].policy[0] == 1.0f && ev[2].policy.empty());
    PackedBoard none_left;  // Some(no moves) trips the reference's assert!(sum > 0.0) as well (common.rs:110)
    none_left.moves = std::vector<int32_t>{};
    CHECK(throws([&] { decode_output(m, &none_left, 1, scalars, logits); }));
    CHECK(std::fabs(ev[2].values.wdl.win - 1.f / 3) < 1e-6f);
    float bad[2] = {NAN, 1};
    CHECK(throws([&] { softmax_in_place(bad, 2); }));  // assert!(sum > 0.0), common.rs:110
    DummyNetwork<PackedBoard> dummy;  // dummy.rs:44-60
    auto d = dummy.evaluate_batch(boards, 3);
    CHECK(d[0].policy.size() == 3 && std::fabs(d[0].policy[1] - 1.f / 3) < 1e-7f && d[2].policy.empty());
}

// ---- job_channel.rs ----
static void test_job_channel() {
    auto [client, server] = job_pair<int, int>(2);
    CHECK(client.map_blocking({}).empty());  // empty request short-circuits, nothing reaches the server (:37-40)
    TryRecvError err;
    CHECK(!server.receiver().try_recv(err) && err == TryRecvError::Empty);
    std::thread worker([srv = server]() mutable {
        auto rx = srv.into_receiver();
        while (auto job = rx.recv()) {
            std::vector<int> y;
            for (int v : job->x) y.push_back(v * 2);
            job->sender.send(std::move(y));
        }
    });
    CHECK((client.map_blocking({1, 2, 3}) == std::vector<int>{2, 4, 6}));
    auto fut = client.map_async({5});
    auto single = client.map_async_single(7);
    CHECK(fut.get() == std::vector<int>{10} && single.get() == 14);
    // map_into (not in the reference): several requests answered into ONE reply channel, in the order they are served
    {
        auto [reply_tx, reply_rx] = bounded<std::vector<int>>(4);
        client.map_into({1}, reply_tx);
        client.map_into({}, reply_tx);  // the empty request short-circuits here too
        client.map_into({2, 3}, reply_tx);
        std::multiset<size_t> sizes;
        int sum = 0;
        for (int k = 0; k < 3; k++) {
            auto y = reply_rx.recv();
            CHECK(y.has_value());
            sizes.insert(y->size());
            for (int v : *y) sum += v;
        }
        CHECK((sizes == std::multiset<size_t>{0, 1, 2}) && sum == 12);
    }
    client = JobClient<int, int>();  // drop the last sender -> the worker sees Disconnected
    server = JobServer<int, int>();
    worker.join();
    // bounded: a full channel blocks the sender until the receiver takes an item
    auto [tx, rx] = bounded<int>(1);
    CHECK(tx.send(1));
    std::atomic<bool> sent{false};
    std::thread t([&, tx2 = tx] { tx2.send(2); sent = true; });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CHECK(!sent);
    CHECK(*rx.recv() == 1 && *rx.recv() == 2);
    t.join();
    rx = Receiver<int>();
    CHECK(!tx.send(3));  // SendError once every receiver is gone
}

// ---- executor.rs:176-302 ----
static void test_executor_state() {
    ExecutorState<int, int> st;
    std::vector<Receiver<std::vector<int>>> replies;
    auto push = [&](std::vector<int> x) {
        auto [tx, rx] = bounded<std::vector<int>>(1);
        replies.push_back(rx);
        st.push_job(Job<int, int>{std::move(x), std::move(tx)});
    };
    CHECK(!st.should_eval(RunCondition::any(), 4));
    push({});  // empty job answered immediately, never queued (:229-231)
    CHECK(replies[0].recv()->empty() && st.sender_count() == 0);
    push({1, 2, 3});
    CHECK(st.should_eval(RunCondition::any(), 4) && !st.should_eval(RunCondition::full_batch(), 4));
    CHECK(!st.should_eval(RunCondition::job_count(2), 4) && st.should_eval(RunCondition::job_count(1), 4));
    push({4});
    push({5, 6, 7, 8, 9});
    CHECK(st.should_eval(RunCondition::full_batch(), 4) && st.items_to_eval() == 9);
    auto eval = [&](size_t max) {
        auto [data, n] = st.get_batch(max);
        std::vector<int> y(data, data + n);
        for (auto &v : y) v *= 10;
        st.respond_batch(std::move(y));
    };
    TryRecvError err;
    eval(4);  // {1,2,3,4}: completes jobs 1 and 2
    CHECK((*replies[1].recv() == std::vector<int>{10, 20, 30}) && (*replies[2].recv() == std::vector<int>{40}));
    CHECK(!replies[3].try_recv(err));
    eval(4);  // {5,6,7,8}: job 3 still incomplete, results wait in leftover_y
    CHECK(!replies[3].try_recv(err) && st.items_waiting_for_send() == 4 && st.items_to_eval() == 1);
    eval(4);  // {9}
    CHECK((*replies[3].recv() == std::vector<int>{50, 60, 70, 80, 90}));
    CHECK(st.items_to_eval() == 0 && st.items_to_send() == 0);
    push({1, 2});
    eval(4);  // shortcut path: the whole batch belongs to one sender (:285-287)
    CHECK((*replies[4].recv() == std::vector<int>{10, 20}));
}

struct FakeNet {
    int id;
    static std::atomic<int> alive;
    explicit FakeNet(int id) : id(id) { alive++; }
    FakeNet(FakeNet &&o) noexcept : id(o.id) { alive++; }
    FakeNet(const FakeNet &) = delete;
    ~FakeNet() { alive--; }
};
std::atomic<int> FakeNet::alive{0};

struct Trace : ExecutorEvents {
    std::mutex m;
    std::vector<std::string> log;
    std::vector<size_t> batches;
    int max_alive = 0;
    void on_drop_network() override { std::lock_guard<std::mutex> g(m); log.push_back("drop"); }
    void on_load_network() override {
        std::lock_guard<std::mutex> g(m);
        log.push_back("load");
        max_alive = std::max(max_alive, FakeNet::alive.load());
    }
    void on_eval(size_t n) override { std::lock_guard<std::mutex> g(m); batches.push_back(n); }
};

// ---- executor.rs:27-146 ----
static void test_executor_loop() {
    auto [client, server] = job_pair<int, int>(8);
    auto [gtx, grx] = bounded<std::optional<int>>(1);
    Trace trace;
    std::thread exec([&, srv = std::move(server), rx = std::move(grx)]() mutable {
        batched_executor_loop<int, FakeNet, int, int>(
            4, RunCondition::any(), std::move(rx), std::move(srv), [](int g) { return FakeNet(g); },
            [](FakeNet &n, const int *x, size_t len) {
                std::vector<int> y(x, x + len);
                for (auto &v : y) v = v * 100 + n.id;
                return y;
            },
            &trace);
    });
    // jobs sent before any network exists wait in the channel (the loop only listens for jobs once it has a network)
    auto early = client.map({1, 2});
    auto early2 = client.map({3});
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
    TryRecvError err;
    CHECK(!early.try_recv(err));
    gtx.send(7);
    CHECK((*early.recv() == std::vector<int>{107, 207}) && (*early2.recv() == std::vector<int>{307}));
    // many clients, order preserved per job, each job answered exactly once
    std::vector<std::thread> gens;
    std::atomic<int> wrong{0}, finished{0};
    for (int t = 0; t < 4; t++)
        gens.emplace_back([&, t, c = client] {
            for (int i = 0; i < 50; i++) {
                std::vector<int> x;
                for (int k = 0; k <= (i + t) % 3; k++) x.push_back(t * 1000 + i * 4 + k);
                auto y = c.map_blocking(x);
                if (y.size() != x.size()) wrong++;
                for (size_t k = 0; k < y.size(); k++)
                    if (y[k] != x[k] * 100 + 7) wrong++;
            }
            finished++;
        });
    // The loop evaluates at most ONE batch per message (executor.rs:93-96), so the tail of a job that was split over
    // two batches waits for the next message.  The server never splits (search_batch_size divides gpu_batch_size);
    // these job sizes do, so a ticker keeps messages coming until the blocking clients are done.
    while (finished < 4) {
        CHECK((client.map_blocking({0}) == std::vector<int>{7}));
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    for (auto &g : gens) g.join();
    CHECK(wrong == 0);
    for (size_t n : trace.batches) CHECK(n >= 1 && n <= 4);  // never above max_batch_size
    // hot swap: None = "wait for new network" drops the current one; the next graph is loaded after the drop
    gtx.send(std::nullopt);
    gtx.send(9);
    CHECK((client.map_blocking({5, 6}) == std::vector<int>{509, 609}));
    client = JobClient<int, int>();
    gtx = Sender<std::optional<int>>();
    exec.join();
    CHECK((trace.log == std::vector<std::string>{"load", "drop", "load"}));
    CHECK(trace.max_alive <= 1 && FakeNet::alive == 0);  // the old network is dropped BEFORE the new one is built (:326-341)
}

// RunCondition::JobCount(n): a batch runs once n jobs (or max_batch items) are queued; what is left when the job
// channel disconnects is evaluated before the loop exits (executor.rs:101-115)
static void test_executor_job_count() {
    auto [client, server] = job_pair<int, int>(8);
    auto [gtx, grx] = bounded<std::optional<int>>(1);
    Trace trace;
    std::thread exec([&, srv = std::move(server), rx = std::move(grx)]() mutable {
        batched_executor_loop<int, FakeNet, int, int>(
            8, RunCondition::job_count(2), std::move(rx), std::move(srv), [](int g) { return FakeNet(g); },
            [](FakeNet &, const int *x, size_t len) { return std::vector<int>(x, x + len); }, &trace);
    });
    gtx.send(1);
    TryRecvError err;
    auto a = client.map({1});
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CHECK(!a.try_recv(err));  // one job < JobCount(2): pending
    auto b = client.map({2, 3});
    CHECK((*a.recv() == std::vector<int>{1}) && (*b.recv() == std::vector<int>{2, 3}));
    auto last = client.map({8});
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CHECK(!last.try_recv(err));
    client = JobClient<int, int>();  // disconnect: the leftover is evaluated, then the loop exits
    exec.join();
    CHECK((*last.recv() == std::vector<int>{8}));
    CHECK((trace.batches == std::vector<size_t>{3, 1}));
}

// executor exits when the graph channel closes before any network arrived and the job channel closes too (:119-143)
static void test_executor_exit_without_network() {
    auto [client, server] = job_pair<int, int>(1);
    auto [gtx, grx] = bounded<std::optional<int>>(1);
    std::thread exec([srv = std::move(server), rx = std::move(grx)]() mutable {
        batched_executor_loop<int, FakeNet, int, int>(
            4, RunCondition::any(), std::move(rx), std::move(srv), [](int g) { return FakeNet(g); },
            [](FakeNet &, const int *x, size_t len) { return std::vector<int>(x, x + len); });
    });
    gtx = Sender<std::optional<int>>();
    client = JobClient<int, int>();
    exec.join();
    CHECK(FakeNet::alive == 0);
}

// ---- pipelined_executor_loop: the same channel semantics with batches in flight ----
// A fake asynchronous network: submit() queues the batch, a "device" delay passes, wait() hands back the OLDEST batch.
struct FakeAsyncNet {
    int id;
    std::deque<std::pair<std::vector<int>, std::chrono::steady_clock::time_point>> q;
    static std::atomic<int> alive, max_in_flight;
    explicit FakeAsyncNet(int id) : id(id) { alive++; }
    FakeAsyncNet(FakeAsyncNet &&o) noexcept : id(o.id), q(std::move(o.q)) { alive++; }
    FakeAsyncNet(const FakeAsyncNet &) = delete;
    ~FakeAsyncNet() {
        if (!q.empty()) std::fprintf(stderr, "FAIL: network dropped with %zu batches in flight\n", q.size()), g_failed++;
        alive--;
    }
    void submit(const int *x, size_t n) {
        q.emplace_back(std::vector<int>(x, x + n), std::chrono::steady_clock::now() + std::chrono::milliseconds(2));
        int cur = (int)q.size(), seen = max_in_flight.load();
        while (cur > seen && !max_in_flight.compare_exchange_weak(seen, cur)) {}
    }
    std::vector<int> wait() {
        auto [y, ready] = std::move(q.front());
        q.pop_front();
        std::this_thread::sleep_until(ready);
        for (auto &v : y) v = v * 100 + id;
        return y;
    }
};
std::atomic<int> FakeAsyncNet::alive{0}, FakeAsyncNet::max_in_flight{0};

static void test_pipelined_loop() {
    auto [client, server] = job_pair<int, int>(16);
    auto [gtx, grx] = bounded<std::optional<int>>(1);
    Trace trace;
    std::thread exec([&, srv = std::move(server), rx = std::move(grx)]() mutable {
        pipelined_executor_loop<int, FakeAsyncNet, int, int>(
            4, 2, RunCondition::any(), std::move(rx), std::move(srv), [](int g) { return FakeAsyncNet(g); },
            [](FakeAsyncNet &n, int *x, size_t len) { n.submit(x, len); }, [](FakeAsyncNet &n) { return n.wait(); },
            &trace);
    });
    // no network yet: jobs wait in the channel
    auto early = client.map({1, 2});
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    TryRecvError err;
    CHECK(!early.try_recv(err));
    gtx.send(7);
    CHECK((*early.recv() == std::vector<int>{107, 207}));
    // many clients with several requests outstanding each: replies in job order, each job answered exactly once, and
    // the loop really keeps two batches in flight
    std::vector<std::thread> gens;
    std::atomic<int> wrong{0};
    for (int t = 0; t < 4; t++)
        gens.emplace_back([&, t, c = client] {
            std::deque<std::pair<std::vector<int>, Receiver<std::vector<int>>>> out;
            for (int i = 0; i < 200; i++) {
                std::vector<int> x;
                for (int k = 0; k <= (i + t) % 3; k++) x.push_back(t * 10000 + i * 4 + k);
                out.emplace_back(x, c.map(x));
                if (out.size() == 3 || i == 199) {
                    while (!out.empty() && (out.size() == 3 || i == 199)) {
                        auto y = out.front().second.recv();
                        const auto &xs = out.front().first;
                        if (!y || y->size() != xs.size()) wrong++;
                        else
                            for (size_t k = 0; k < xs.size(); k++)
                                if ((*y)[k] != xs[k] * 100 + 7) wrong++;
                        out.pop_front();
                    }
                }
            }
        });
    for (auto &g : gens) g.join();
    CHECK(wrong == 0);
    CHECK(FakeAsyncNet::max_in_flight == 2);
    for (size_t n : trace.batches) CHECK(n >= 1 && n <= 4);
    // hot swap with work in flight: the old network answers what it was given, is dropped, then the new one is built
    auto a = client.map({1, 2, 3, 4});
    auto b = client.map({5, 6, 7, 8});
    gtx.send(std::nullopt);
    gtx.send(9);
    auto ya = *a.recv(), yb = *b.recv();
    CHECK(ya.size() == 4 && yb.size() == 4);
    for (int v : ya) CHECK(v % 100 == 7 || v % 100 == 9);  // whichever network had it, one network per batch
    CHECK((client.map_blocking({5, 6}) == std::vector<int>{509, 609}));
    // disconnect with requests outstanding: everything queued is answered before the loop returns
    auto c1 = client.map({11}), c2 = client.map({12, 13}), c3 = client.map({14});
    client = JobClient<int, int>();
    gtx = Sender<std::optional<int>>();
    exec.join();
    CHECK((*c1.recv() == std::vector<int>{1109}) && (*c2.recv() == std::vector<int>{1209, 1309}) &&
          (*c3.recv() == std::vector<int>{1409}));
    CHECK((trace.log == std::vector<std::string>{"load", "drop", "load"}));
    CHECK(FakeAsyncNet::alive == 0);
}

// RunCondition::JobCount counts the jobs NOT yet handed to the network: with one batch in flight the next one still
// waits for its own `count` jobs
static void test_pipelined_job_count() {
    auto [client, server] = job_pair<int, int>(16);
    auto [gtx, grx] = bounded<std::optional<int>>(1);
    Trace trace;
    std::thread exec([&, srv = std::move(server), rx = std::move(grx)]() mutable {
        pipelined_executor_loop<int, FakeAsyncNet, int, int>(
            8, 2, RunCondition::job_count(2), std::move(rx), std::move(srv), [](int g) { return FakeAsyncNet(g); },
            [](FakeAsyncNet &n, int *x, size_t len) { n.submit(x, len); }, [](FakeAsyncNet &n) { return n.wait(); },
            &trace);
    });
    gtx.send(1);
    TryRecvError err;
    auto a = client.map({1});
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
    CHECK(!a.try_recv(err));  // one job < JobCount(2)
    auto b = client.map({2, 3});
    CHECK((*a.recv() == std::vector<int>{101}) && (*b.recv() == std::vector<int>{201, 301}));
    auto c = client.map({4});
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
    CHECK(!c.try_recv(err));  // again only one job pending
    client = JobClient<int, int>();
    exec.join();
    CHECK((*c.recv() == std::vector<int>{401}));
    CHECK((trace.batches == std::vector<size_t>{3, 1}));
}

// ---- symmetry: D4 tables of the reference + RandomSymmetryNetwork un-mapping ----
struct CoordNet : Network<AtaxxSymBoard> {  // policy weight of a move depends on where it lands on the evaluated board
    size_t max_batch_size() const override { return 64; }
    std::vector<ZeroEvaluation> evaluate_batch(const AtaxxSymBoard *boards, size_t n) override {
        std::vector<ZeroEvaluation> out(n);
        for (size_t i = 0; i < n; i++) {
            out[i].values.value = (float)__builtin_popcountll(boards[i].tiles_next);
            if (!boards[i].moves) continue;  // a finished game: an empty policy
            for (const auto &mv : *boards[i].moves)
                out[i].policy.push_back((float)AtaxxStdMapper(boards[i].size).move_to_index(mv));
        }
        return out;
    }
};

static void test_symmetry(const std::string &golden_dir) {
    std::ifstream f(golden_dir + "/ataxx_symmetry.txt");
    CHECK(f.good());
    std::string line;
    int rows = 0;
    while (std::getline(f, line)) {
        std::istringstream is(line);
        int size, index, tr, fx, fy;
        size_t n;
        is >> size >> index >> tr >> fx >> fy >> n;
        const D4 d = D4::from_index(index);
        CHECK(d.transpose == (bool)tr && d.flip_x == (bool)fx && d.flip_y == (bool)fy);
        AtaxxStdMapper m(size);
        CHECK(n == m.policy_len());
        for (size_t i = 0; i < n; i++) {
            long expect;
            is >> expect;
            auto mv = m.index_to_move(i);
            if (!mv) {
                CHECK(expect == -1);
                continue;
            }
            CHECK((long)m.move_to_index(ataxx_map_move(size, index, *mv)) == expect);
        }
        rows++;
    }
    CHECK(rows == 7 * 8);

    AtaxxSymBoard b;
    b.size = 7;
    b.tiles_next = 0b1000001;
    b.tiles_other = 1ull << 48;
    b.moves = std::vector<AtaxxMove>{{AtaxxMove::Copy, 0, 0, 1, 0}, {AtaxxMove::Jump, 0, 0, 2, 1}, {AtaxxMove::Copy, 0, 0, 5, 1},
                                     {AtaxxMove::Jump, 6, 0, 4, 2}};
    // plane mapping agrees with the move mapping: the tile at (x,y) lands where a copy to (x,y) lands
    for (int sym = 0; sym < 8; sym++) {
        AtaxxMove to{AtaxxMove::Copy, 0, 0, 6, 0};
        auto mapped = ataxx_map_move(7, sym, to);
        CHECK((ataxx_map_tiles(7, sym, 1ull << 6) >> (mapped.to_y * 7 + mapped.to_x)) & 1);
    }
    RandomSymmetryNetwork<AtaxxSymBoard, CoordNet> net(CoordNet{}, std::mt19937_64(3), true);
    for (int rep = 0; rep < 20; rep++) {
        auto ev = net.evaluate(b);
        CHECK(ev.values.value == 2 && ev.policy.size() == 4);
        // whatever symmetry was drawn, entry i belongs to the image of move i (symmetry.rs:126-148)
        bool found = false;
        for (int sym = 0; sym < 8 && !found; sym++) {
            bool all = true;
            for (size_t i = 0; i < 4; i++)
                all &= ev.policy[i] == (float)AtaxxStdMapper(7).move_to_index(ataxx_map_move(7, sym, (*b.moves)[i]));
            found = all;
        }
        CHECK(found);
    }
    RandomSymmetryNetwork<AtaxxSymBoard, CoordNet> off(CoordNet{}, std::mt19937_64(3), false);  // disabled: passthrough
    auto ev = off.evaluate(b);
    CHECK(ev.policy[0] == (float)AtaxxStdMapper(7).move_to_index((*b.moves)[0]));

    // AverageSymmetryNetwork (symmetry.rs:70-124, average_evals :150-184): entry i = the mean over all eight symmetries of
    // what the inner network gives the image of move i on the mapped board; values averaged; more mapped boards (5 x 8)
    // than the inner network's batch (a CoordNet of 16) are re-batched
    struct SmallNet : CoordNet {
        size_t calls = 0;
        size_t max_batch_size() const override { return 16; }
        std::vector<ZeroEvaluation> evaluate_batch(const AtaxxSymBoard *boards, size_t n) override {
            calls++;
            if (n > 16) throw std::invalid_argument("batch above max_batch_size");
            return CoordNet::evaluate_batch(boards, n);
        }
    };
    AverageSymmetryNetwork<AtaxxSymBoard, SmallNet> avg{SmallNet{}};
    CHECK(avg.max_batch_size() == (size_t)-1);
    std::vector<AtaxxSymBoard> five(5, b);
    five[3].moves = std::nullopt;  // a finished game: an empty policy
    five[3].tiles_next = 0b111;
    auto evs = avg.evaluate_batch(five.data(), five.size());
    CHECK(evs.size() == 5 && avg.inner().calls == 3);  // 40 mapped boards in chunks of 16
    for (size_t k = 0; k < 5; k++) {
        if (k == 3) {
            CHECK(evs[k].policy.empty() && evs[k].values.value == 3);
            continue;
        }
        CHECK(evs[k].values.value == 2 && evs[k].policy.size() == 4);
        for (size_t i = 0; i < 4; i++) {
            float want = 0;
            for (int sym = 0; sym < 8; sym++) want += (float)AtaxxStdMapper(7).move_to_index(ataxx_map_move(7, sym, (*b.moves)[i])) / 8.0f;
            CHECK(std::fabs(evs[k].policy[i] - want) <= 1e-4f * want);
        }
    }
}

// ---- per-device spawn (server.rs:323-331, server_alphazero.rs:89-121) over several devices, without a GPU: a fake network
// that remembers which device it was built for and answers with it ----
struct FakeGraph {
    int generation;
};
static std::mutex g_fake_mu;
static std::vector<std::tuple<int, int, std::thread::id>> g_fake_built;  // (device, generation, thread)
struct FakeDeviceNet {
    PackedMapper mapper;
    std::shared_ptr<const FakeGraph> graph;
    size_t max_batch;
    int device;
    std::vector<std::vector<PackedBoard>> pending;
    FakeDeviceNet(PackedMapper m, std::shared_ptr<const FakeGraph> g, size_t max_batch, int device, int /*dtype*/)
        : mapper(m), graph(std::move(g)), max_batch(max_batch), device(device) {
        std::lock_guard<std::mutex> lock(g_fake_mu);
        g_fake_built.emplace_back(device, graph->generation, std::this_thread::get_id());
    }
    void set_device_decode(bool) {}
    static constexpr size_t max_in_flight() { return 2; }
    ZeroEvaluation answer(const PackedBoard &b) const {
        ZeroEvaluation e;
        e.values.value = (float)device;                           // which device answered
        e.values.moves_left = (float)graph->generation;           // with which network
        e.values.wdl = {b.scalars.empty() ? 0.f : b.scalars[0], 0.f, 0.f};  // and for which board
        return e;
    }
    std::vector<ZeroEvaluation> evaluate_batch(const PackedBoard *x, size_t n) {
        if (n > max_batch) throw std::logic_error("batch too large");
        std::vector<ZeroEvaluation> y;
        for (size_t i = 0; i < n; i++) y.push_back(answer(x[i]));
        return y;
    }
    void submit_batch(PackedBoard *x, size_t n) { pending.emplace_back(x, x + n); }
    std::vector<ZeroEvaluation> wait_batch() {
        auto boards = std::move(pending.front());
        pending.erase(pending.begin());
        return evaluate_batch(boards.data(), boards.size());
    }
};

static void test_spawn_all_devices() {
    for (size_t depth : {1, 2}) {
        {
            std::lock_guard<std::mutex> lock(g_fake_mu);
            g_fake_built.clear();
        }
        StartupSettings st;
        st.gpu_threads_per_device = 2;
        st.gpu_batch_size = 8;
        st.search_batch_size = 8;  // one job = one batch (RunCondition::JobCount(1)): with several jobs per batch the last
                                   // ones of a finite test would strand on two executor threads, as in the reference
        st.pipeline_depth = depth;
        EvalCounters counters;
        PackedMapper mapper{1, 2, 4, 1, 3};
        const std::vector<int> devices = {0, 1, 5};  // (device ordinals need not be contiguous: `--device 0 1 5`)
        auto all = spawn_all_devices<PackedBoard, PackedMapper, FakeDeviceNet, std::shared_ptr<const FakeGraph>>(
            devices, st, mapper, 0, &counters);
        CHECK(all.size() == 3);
        CHECK(throws([&] {
            spawn_all_devices<PackedBoard, PackedMapper, FakeDeviceNet, std::shared_ptr<const FakeGraph>>({}, st, mapper, 0, nullptr);
        }));
        auto graph1 = std::make_shared<const FakeGraph>(FakeGraph{1});
        for (auto &dev : all) dev->send_graph(graph1);
        // every device has its own job channel: a job sent to device d is answered by a network built for device d
        auto run = [&](int generation) {
            std::vector<std::thread> gens;
            std::atomic<int> wrong{0};
            for (size_t dg = 0; dg < all.size() * 2; dg++)
                gens.emplace_back([&, d = dg / 2] {
                    auto client = all[d]->eval_client;
                    for (int round = 0; round < 40; round++) {
                        std::vector<PackedBoard> x(st.search_batch_size);
                        for (size_t k = 0; k < x.size(); k++) {
                            x[k].bits.assign(1, 0);
                            x[k].scalars = {(float)(round * 10 + (int)k)};
                        }
                        auto y = client.map_blocking(std::move(x));
                        if (y.size() != st.search_batch_size) { wrong++; continue; }
                        for (size_t k = 0; k < y.size(); k++)
                            if (y[k].values.value != (float)devices[d] || y[k].values.moves_left != (float)generation ||
                                y[k].values.wdl.win != (float)(round * 10 + (int)k))
                                wrong++;
                    }
                });
            for (auto &g : gens) g.join();
            return wrong.load();
        };
        CHECK(run(1) == 0);
        CHECK(counters.real == 3 * 2 * 40 * st.search_batch_size);
        // NewNetwork reaches every executor of every device (commander.rs:36-45)
        auto graph2 = std::make_shared<const FakeGraph>(FakeGraph{2});
        for (auto &dev : all) dev->send_graph(graph2);
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        CHECK(run(2) == 0);
        for (auto &dev : all) dev->join();
        std::lock_guard<std::mutex> lock(g_fake_mu);
        for (int device : devices)
            for (int generation : {1, 2}) {
                std::set<std::thread::id> threads;
                for (auto &[d, g, t] : g_fake_built)
                    if (d == device && g == generation) threads.insert(t);
                CHECK(threads.size() == st.gpu_threads_per_device);  // one network per executor thread, built ON that thread
            }
        CHECK(g_fake_built.size() == devices.size() * 2 * st.gpu_threads_per_device);
    }
}

// The load generator of tests/cpp/bench_executor.cpp: threads that each stand for many games and take the replies in the
// order they arrive.  With the reference's sizing rule (server_alphazero.rs:47) this keeps several executor threads fed
// for ever; taking the replies in SUBMISSION order instead does not (a thread can wait for a request that sits in an
// executor's partial batch while its answered games are not replaced), which is why `map_into` exists.
static void test_multiplexed_generators() {
    for (size_t depth : {1, 2}) {
        StartupSettings st;
        st.gpu_threads_per_device = 4;
        st.gpu_batch_size = 32;
        st.search_batch_size = 4;  // RunCondition::JobCount(8)
        st.pipeline_depth = depth;
        EvalCounters counters;
        PackedMapper mapper{1, 2, 4, 1, 3};
        auto all = spawn_all_devices<PackedBoard, PackedMapper, FakeDeviceNet, std::shared_ptr<const FakeGraph>>({0}, st, mapper, 0,
                                                                                                              &counters);
        all[0]->send_graph(std::make_shared<const FakeGraph>(FakeGraph{1}));
        const DeviceSizing sizing(st);
        const size_t n_threads = 3, games_per_thread = ceil_div(sizing.concurrent_games, n_threads), target = 30000;
        std::atomic<size_t> answered{0};
        std::atomic<bool> stalled{false};
        std::vector<std::thread> gens;
        for (size_t t = 0; t < n_threads; t++)
            gens.emplace_back([&, client = all[0]->eval_client] {
                auto request = [&] {
                    std::vector<PackedBoard> x(st.search_batch_size);
                    for (auto &b : x) b.bits.assign(1, 0), b.scalars = {1.f};
                    return x;
                };
                auto [reply_tx, reply_rx] = bounded<std::vector<ZeroEvaluation>>(games_per_thread);
                for (size_t g = 0; g < games_per_thread; g++) client.map_into(request(), reply_tx);
                while (answered < target && !stalled) {
                    auto y = reply_rx.recv();
                    if (!y || y->size() != st.search_batch_size) break;
                    answered += y->size();
                    client.map_into(request(), reply_tx);
                }
            });
        // watchdog: progress every 5 s or the test fails instead of hanging
        std::thread watchdog([&] {
            size_t last = 0;
            for (int quiet = 0; answered < target && quiet < 50;) {
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
                const size_t now = answered;
                quiet = now == last ? quiet + 1 : 0;
                last = now;
                if (quiet >= 50) stalled = true;
            }
        });
        watchdog.join();
        CHECK(!stalled && answered >= target);
        if (stalled) std::_Exit(1);  // (generators are blocked in recv(): nothing to join)
        for (auto &g : gens) g.join();
        all[0]->join();
        CHECK(counters.real >= target);
    }
}

// ---- PrepHelper (hip_network.hpp): the helper thread that shares a batch's host work with the executor thread ----
static void test_prep_helper() {
    {
        PrepHelper h;  // destroyed idle, never used
    }
    PrepHelper h;
    std::vector<int> out(1000, 0);
    for (int round = 0; round < 200; round++) {  // the owner fills one half while the helper fills the other
        h.start([&out, round] {
            for (size_t i = 500; i < 1000; i++) out[i] = round + (int)i;
        });
        for (size_t i = 0; i < 500; i++) out[i] = round + (int)i;
        h.finish();
        bool ok = true;
        for (size_t i = 0; i < 1000; i++) ok &= out[i] == round + (int)i;
        CHECK(ok);
    }
    h.finish();  // nothing in flight: returns at once
    // what the job throws comes out of finish(), once; the helper stays usable
    h.start([] { throw std::out_of_range("move_to_index out of range"); });
    bool caught = false;
    try {
        h.finish();
    } catch (const std::out_of_range &e) {
        caught = std::string(e.what()) == "move_to_index out of range";
    }
    CHECK(caught);
    h.finish();
    int x = 0;
    h.start([&x] { x = 7; });
    h.finish();
    CHECK(x == 7);
    CHECK(h.cpu_ns.load() > 0);
}

int main(int argc, char **argv) {
    const std::string golden = argc > 1 ? argv[1] : "tests/golden";
    std::fputs("prep helper\n", stderr); test_prep_helper();
    std::fputs("bitbuffer\n", stderr); test_bitbuffer();
    std::fputs("mappers\n", stderr); test_mappers();
    std::fputs("decode\n", stderr); test_decode();
    std::fputs("chess policy\n", stderr); test_chess_policy(golden);
    std::fputs("adapters\n", stderr); test_adapters();
    std::fputs("job_channel\n", stderr); test_job_channel();
    std::fputs("state\n", stderr); test_executor_state();
    std::fputs("loop\n", stderr); test_executor_loop();
    std::fputs("job_count\n", stderr); test_executor_job_count();
    std::fputs("exit\n", stderr); test_executor_exit_without_network();
    std::fputs("pipelined\n", stderr); test_pipelined_loop();
    std::fputs("pipelined job_count\n", stderr); test_pipelined_job_count();
    std::fputs("symmetry\n", stderr); test_symmetry(golden);
    std::fputs("devices\n", stderr); test_spawn_all_devices();
    std::fputs("multiplexed generators\n", stderr); test_multiplexed_generators();
    if (g_failed) {
        std::fprintf(stderr, "%d check(s) failed\n", g_failed);
        return 1;
    }
    std::puts("host tests ok");
    return 0;
}
