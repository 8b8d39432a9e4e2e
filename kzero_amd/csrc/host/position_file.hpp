// position_file.hpp — packed position records `games_N.{bin,off,json}`: the writer the self-play
// server uses (rust/kz-selfplay/src/binary_output.rs:88-373, `BinaryOutput`) and the reader the trainer uses
// (python/lib/data/file.py:68-135, python/lib/data/position.py:34-104), in C++ next to the rest of the host mirror.
//
// Per position, appended to .bin (binary_output.rs:210-256): 26 f32 scalars in `scalar_names()` order (:321-349) |
// ceil(input_bool_len / 8) bytes of BitBuffer storage | input_scalar_count f32 | available_mv_count u32 policy indices |
// available_mv_count f32 policy values.  .off: one u64-LE byte offset per position, then one u64 start-position index
// per game (:239, :281).  .json: metadata, written as .json.tmp and renamed (:287-289).
//
// The board part of a record is exactly the packed input of kz_engine_eval_packed (same BitBuffer layout, same scalars),
// so recorded self-play positions replay through the engine unchanged.  kzero_amd/position_file.py is the same format
// in Python; tests/test_position_file.py checks that each reads what the other writes.
#pragma once
#include <charconv>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace kz::host {

inline const std::vector<std::string> &position_scalar_names() {  // binary_output.rs:321-349
    static const std::vector<std::string> names = {
        "game_id", "pos_index", "game_length", "zero_visits", "is_full_search", "is_final_position", "is_terminal",
        "hit_move_limit", "available_mv_count", "played_mv", "kdl_policy",
        "final_v", "final_wdl_w", "final_wdl_d", "final_wdl_l", "final_moves_left",
        "zero_v", "zero_wdl_w", "zero_wdl_d", "zero_wdl_l", "zero_moves_left",
        "net_v", "net_wdl_w", "net_wdl_d", "net_wdl_l", "net_moves_left",
    };
    return names;
}
constexpr size_t POSITION_SCALAR_COUNT = 26;
constexpr size_t POSITION_SCALAR_AVAILABLE_MV_COUNT = 8;  // index of "available_mv_count"

struct PositionRecord {
    float scalars[POSITION_SCALAR_COUNT] = {};  // in position_scalar_names() order
    std::vector<uint8_t> bits;                  // BitBuffer storage, ceil(bool_len / 8) bytes
    std::vector<float> input_scalars;           // input_scalar_count
    std::vector<uint32_t> policy_indices;       // available_mv_count
    std::vector<float> policy_values;           // available_mv_count
};

struct PositionFileMeta {
    std::string game;
    std::vector<int64_t> input_bool_shape, policy_shape;
    int64_t input_scalar_count = 0, game_count = 0, position_count = 0;
    int64_t max_game_length = -1, min_game_length = -1;
    bool includes_game_start_indices = true;
    size_t bits_bytes() const {
        int64_t n = 1;
        for (auto d : input_bool_shape) n *= d;
        return (size_t)((n + 7) / 8);
    }
};

// BinaryOutput (binary_output.rs:88-289) without the game-playing part: positions are appended game by game.
class PositionFileWriter {
    std::string path_;
    PositionFileMeta meta_;
    std::ofstream bin_, off_;
    uint64_t next_offset_ = 0;
    std::vector<uint64_t> game_starts_;
    std::vector<int64_t> lengths_;
    bool finished_ = false;
    bool outcomes_known_ = true;
    double root_wdl_sum_[3] = {0, 0, 0}, hit_move_limit_count_ = 0;

    static void put_u64(std::ofstream &f, uint64_t v) {
        unsigned char b[8];
        for (int i = 0; i < 8; i++) b[i] = (unsigned char)(v >> (8 * i));  // little endian whatever the host is
        f.write(reinterpret_cast<const char *>(b), 8);
    }
    template <class T>
    void put_bin(const T *p, size_t n) {
        static_assert(sizeof(T) == 4 || sizeof(T) == 1, "f32 / u32 / u8");
        if (n) bin_.write(reinterpret_cast<const char *>(p), (std::streamsize)(n * sizeof(T)));  // (little-endian hosts)
        next_offset_ += n * sizeof(T);
    }

  public:
    PositionFileWriter(const std::string &path, const std::string &game, std::vector<int64_t> input_bool_shape,
                       int64_t input_scalar_count, std::vector<int64_t> policy_shape)
        : path_(path) {
        if (path.find('.', path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/')) != std::string::npos)
            throw std::invalid_argument("path must not have an extension");  // binary_output.rs:91-95
        meta_.game = game;
        meta_.input_bool_shape = std::move(input_bool_shape);
        meta_.input_scalar_count = input_scalar_count;
        meta_.policy_shape = std::move(policy_shape);
        bin_.open(path + ".bin", std::ios::binary | std::ios::trunc);
        off_.open(path + ".off", std::ios::binary | std::ios::trunc);
        if (!bin_ || !off_) throw std::runtime_error("cannot create " + path + ".bin/.off");
    }
    const PositionFileMeta &meta() const { return meta_; }

    // all positions of one game, the terminal position last (includes_terminal_positions = true)
    // root_wdl: the game's outcome as (win, draw, loss) for the player to move in the start position
    // (binary_output.rs:144), hit_move_limit: 1 when the game stopped without an outcome (:145), -1 / nullptr: unknown —
    // the metadata then says NaN instead of an invented value
    void append_game(const std::vector<PositionRecord> &records, const float *root_wdl = nullptr, int hit_move_limit = -1) {
        if (finished_) throw std::logic_error("This output is already finished");
        if (!root_wdl || hit_move_limit < 0) {
            outcomes_known_ = false;
        } else {
            for (int i = 0; i < 3; i++) root_wdl_sum_[i] += root_wdl[i];
            hit_move_limit_count_ += hit_move_limit ? 1.0 : 0.0;
        }
        game_starts_.push_back((uint64_t)meta_.position_count);
        lengths_.push_back((int64_t)records.size() - 1);
        for (const auto &r : records) append_position(r);
        meta_.game_count++;
    }

    void append_position(const PositionRecord &r) {  // binary_output.rs:210-256
        if (r.bits.size() != meta_.bits_bytes()) throw std::invalid_argument("bits: wrong length");
        if ((int64_t)r.input_scalars.size() != meta_.input_scalar_count) throw std::invalid_argument("input scalars: wrong length");
        if (r.policy_indices.size() != r.policy_values.size()) throw std::invalid_argument("policy: indices and values differ in length");
        if ((size_t)r.scalars[POSITION_SCALAR_AVAILABLE_MV_COUNT] != r.policy_values.size())
            throw std::invalid_argument("available_mv_count does not match the policy length");
        if (!r.policy_values.empty()) {  // assert_normalized_or_nan (:317-319)
            double s = 0;
            for (float v : r.policy_values) s += v;
            if (!std::isnan(s) && std::fabs(1.0 - s) >= 0.001) throw std::invalid_argument("policy is not normalised");
        }
        put_u64(off_, next_offset_);
        put_bin(r.scalars, POSITION_SCALAR_COUNT);
        put_bin(r.bits.data(), r.bits.size());
        put_bin(r.input_scalars.data(), r.input_scalars.size());
        put_bin(r.policy_indices.data(), r.policy_indices.size());
        put_bin(r.policy_values.data(), r.policy_values.size());
        meta_.position_count++;
    }

    void finish() {  // binary_output.rs:258-290
        if (finished_) throw std::logic_error("This output is already finished");
        finished_ = true;
        if (!lengths_.empty()) {
            meta_.max_game_length = *std::max_element(lengths_.begin(), lengths_.end());
            meta_.min_game_length = *std::min_element(lengths_.begin(), lengths_.end());
        }
        for (uint64_t s : game_starts_) put_u64(off_, s);
        bin_.close();
        off_.close();
        auto list = [](const std::vector<int64_t> &v) {
            std::string s = "[";
            for (size_t i = 0; i < v.size(); i++) s += (i ? ", " : "") + std::to_string(v[i]);
            return s + "]";
        };
        const bool known = outcomes_known_ && meta_.game_count > 0;
        // averages over the games (:276-277) as serde_json writes an f32: the shortest decimal that reads back as the same
        // f32, and `null` for NaN (a game that came without them; 0 games) — strict JSON, byte-identical to the Python writer
        auto outcome_number = [&](double sum) -> std::string {
            if (!known) return "null";
            char buf[32];
            const auto r = std::to_chars(buf, buf + sizeof buf, (float)(sum / (double)meta_.game_count));
            std::string out(buf, r.ptr);
            if (out.find_first_of(".en") == std::string::npos) out += ".0";  // serde_json prints 1.0, not 1
            return out;
        };
        auto outcome_list = [&] {
            return "[" + outcome_number(root_wdl_sum_[0]) + ", " + outcome_number(root_wdl_sum_[1]) + ", " +
                   outcome_number(root_wdl_sum_[2]) + "]";
        };
        std::ostringstream js;
        js << "{\n  \"game\": \"" << meta_.game << "\",\n  \"input_bool_shape\": " << list(meta_.input_bool_shape)
           << ",\n  \"input_scalar_count\": " << meta_.input_scalar_count << ",\n  \"policy_shape\": " << list(meta_.policy_shape)
           << ",\n  \"game_count\": " << meta_.game_count << ",\n  \"position_count\": " << meta_.position_count
           << ",\n  \"includes_terminal_positions\": true,\n  \"includes_game_start_indices\": true"
           << ",\n  \"max_game_length\": " << meta_.max_game_length << ",\n  \"min_game_length\": " << meta_.min_game_length
           << ",\n  \"root_wdl\": " << outcome_list() << ",\n  \"hit_move_limit\": " << outcome_number(hit_move_limit_count_)
           << ",\n  \"scalar_names\": [";
        const auto &names = position_scalar_names();
        for (size_t i = 0; i < names.size(); i++) js << (i ? ", " : "") << '"' << names[i] << '"';
        js << "]\n}\n";
        {
            std::ofstream f(path_ + ".json.tmp", std::ios::trunc);
            f << js.str();
            if (!f) throw std::runtime_error("cannot write " + path_ + ".json.tmp");
        }
        if (std::rename((path_ + ".json.tmp").c_str(), (path_ + ".json").c_str()) != 0)  // atomic publish (:287-289)
            throw std::runtime_error("cannot rename " + path_ + ".json.tmp");
    }
};

// Reader (python/lib/data/file.py:68-135 + position.py:34-104)
class PositionFile {
    std::vector<uint8_t> bin_;
    std::vector<uint64_t> offsets_, game_starts_;
    PositionFileMeta meta_;

    static std::string slurp(const std::string &p, bool binary) {
        std::ifstream f(p, binary ? std::ios::binary : std::ios::in);
        if (!f) throw std::runtime_error("cannot open " + p);
        std::stringstream ss;
        ss << f.rdbuf();
        return ss.str();
    }
    // the value text of a top-level "key": a number, true/false, a quoted string, or a [...] list (metadata only:
    // binary_output.rs:22-41 has no nesting)
    static std::string json_value(const std::string &js, const std::string &key) {
        const std::string needle = '"' + key + '"';
        size_t k = js.find(needle);
        if (k == std::string::npos) return {};
        size_t p = js.find(':', k + needle.size());
        if (p == std::string::npos) return {};
        p++;
        while (p < js.size() && std::isspace((unsigned char)js[p])) p++;
        size_t e = p;
        if (p < js.size() && js[p] == '[') e = js.find(']', p) + 1;
        else if (p < js.size() && js[p] == '"') e = js.find('"', p + 1) + 1;
        else
            while (e < js.size() && js[e] != ',' && js[e] != '}' && js[e] != '\n') e++;
        return js.substr(p, e - p);
    }
    static std::vector<int64_t> json_ints(const std::string &v) {
        std::vector<int64_t> out;
        for (size_t i = 0; i < v.size();) {
            if (std::isdigit((unsigned char)v[i]) || v[i] == '-') {
                size_t j = i + 1;
                while (j < v.size() && std::isdigit((unsigned char)v[j])) j++;
                out.push_back(std::stoll(v.substr(i, j - i)));
                i = j;
            } else {
                i++;
            }
        }
        return out;
    }

  public:
    explicit PositionFile(const std::string &path) {
        const std::string js = slurp(path + ".json", false);
        auto need = [&](const char *k) {
            std::string v = json_value(js, k);
            if (v.empty()) throw std::runtime_error(std::string("metadata: missing '") + k + "'");
            return v;
        };
        std::string game = need("game");
        meta_.game = game.size() >= 2 ? game.substr(1, game.size() - 2) : game;
        meta_.input_bool_shape = json_ints(need("input_bool_shape"));
        meta_.policy_shape = json_ints(need("policy_shape"));
        meta_.input_scalar_count = json_ints(need("input_scalar_count")).at(0);
        meta_.game_count = json_ints(need("game_count")).at(0);
        meta_.position_count = json_ints(need("position_count")).at(0);
        meta_.max_game_length = json_ints(need("max_game_length")).at(0);
        meta_.min_game_length = json_ints(need("min_game_length")).at(0);
        meta_.includes_game_start_indices = json_value(js, "includes_game_start_indices").rfind("true", 0) == 0;
        // scalar layout must be the one this reader knows (position.py reads by name; the order is fixed in practice)
        {
            const std::string names = need("scalar_names");
            size_t pos = 0;
            for (const auto &n : position_scalar_names()) {
                size_t at = names.find('"' + n + '"', pos);
                if (at == std::string::npos) throw std::runtime_error("metadata: unexpected scalar_names");
                pos = at + 1;
            }
        }
        const std::string b = slurp(path + ".bin", true), o = slurp(path + ".off", true);
        bin_.assign(b.begin(), b.end());
        if (o.size() % 8) throw std::runtime_error("offset file: size is not a multiple of 8");
        std::vector<uint64_t> off(o.size() / 8);
        for (size_t i = 0; i < off.size(); i++) {
            uint64_t v = 0;
            for (int k = 0; k < 8; k++) v |= (uint64_t)(unsigned char)o[i * 8 + k] << (8 * k);
            off[i] = v;
        }
        const size_t n = (size_t)meta_.position_count;
        const size_t expect = n + (meta_.includes_game_start_indices ? (size_t)meta_.game_count : 0);
        if (off.size() != expect) throw std::runtime_error("Mismatch in offset size");  // file.py:97-102
        offsets_.assign(off.begin(), off.begin() + (long)n);
        game_starts_.assign(off.begin() + (long)n, off.end());
    }

    const PositionFileMeta &meta() const { return meta_; }
    size_t size() const { return (size_t)meta_.position_count; }
    const std::vector<uint64_t> &game_starts() const { return game_starts_; }

    PositionRecord position(size_t pi) const {
        if (pi >= size()) throw std::out_of_range("position index");
        const uint64_t start = offsets_[pi], end = pi + 1 < offsets_.size() ? offsets_[pi + 1] : bin_.size();  // file.py:116-124
        if (start > end || end > bin_.size()) throw std::runtime_error("offset out of range");
        const uint8_t *p = bin_.data() + start;
        size_t left = (size_t)(end - start);
        auto take = [&](void *dst, size_t bytes) {
            if (left < bytes) throw std::runtime_error("position record too short");
            if (bytes) std::memcpy(dst, p, bytes);  // (an empty vector's data() may be null)
            p += bytes;
            left -= bytes;
        };
        PositionRecord r;
        take(r.scalars, sizeof(r.scalars));
        r.bits.resize(meta_.bits_bytes());
        take(r.bits.data(), r.bits.size());
        r.input_scalars.resize((size_t)meta_.input_scalar_count);
        take(r.input_scalars.data(), r.input_scalars.size() * 4);
        const float mvf = r.scalars[POSITION_SCALAR_AVAILABLE_MV_COUNT];
        if (!(mvf >= 0) || mvf > 1e7f) throw std::runtime_error("bad available_mv_count");
        const size_t mv = (size_t)mvf;
        r.policy_indices.resize(mv);
        take(r.policy_indices.data(), mv * 4);
        r.policy_values.resize(mv);
        take(r.policy_values.data(), mv * 4);
        if (left != 0) throw std::runtime_error("Leftover bytes in position record");  // Taker.finish(), position.py:104
        return r;
    }
};

}  // namespace kz::host
