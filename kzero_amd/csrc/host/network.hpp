// network.hpp — the `Network` contract and output decoding (rust/kz-core/src/network/mod.rs:26-73,
// rust/kz-core/src/network/common.rs:16-114, rust/kz-core/src/network/dummy.rs:16-60).
//
// Board concept (what the ext `board-game` crate provides on the Rust side):
//     using Move = ...;                                              // equality comparable
//     std::optional<std::vector<Move>> available_moves() const;      // nullopt once the game is done
// PolicyMapper concept (rust/kz-core/src/mapping/mod.rs:67-80):
//     size_t policy_len() const;  size_t move_to_index(const B&, const Move&) const;
#pragma once
#include <algorithm>
#include <string>
#include <cmath>
#include <cstddef>
#include <limits>
#include <optional>
#include <stdexcept>
#include <vector>

namespace kz::host {

struct WDL {
    float win = 0, draw = 0, loss = 0;
};

// rust/kz-core/src/zero/values.rs:13-18
struct ZeroValuesPov {
    float value = 0;
    WDL wdl;
    float moves_left = 0;
};

// network/mod.rs:26-31: the policy holds one entry per AVAILABLE move, in available_moves() order
struct ZeroEvaluation {
    ZeroValuesPov values;
    std::vector<float> policy;
};

// network/mod.rs:52-63
template <class B>
struct Network {
    virtual ~Network() = default;
    virtual size_t max_batch_size() const = 0;
    // returns one evaluation per board, in order
    virtual std::vector<ZeroEvaluation> evaluate_batch(const B *boards, size_t n) = 0;
    ZeroEvaluation evaluate(const B &board) {  // :58-62
        auto r = evaluate_batch(&board, 1);
        if (r.size() != 1) throw std::logic_error("evaluate_batch returned the wrong number of results");
        return std::move(r[0]);
    }
};

// common.rs:102-114; throws where the reference asserts `sum > 0.0`
inline void softmax_in_place(float *v, size_t n) {
    float max = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < n; i++) max = v[i] > max ? v[i] : max;
    float sum = 0.0f;
    for (size_t i = 0; i < n; i++) {
        v[i] = std::exp(v[i] - max);
        sum += v[i];
    }
    if (!(sum > 0.0f)) throw std::runtime_error("Softmax input sum must be strictly positive");
    for (size_t i = 0; i < n; i++) v[i] /= sum;
}

// common.rs:200-215
inline ZeroValuesPov zero_values_from_scalars(const float *s) {
    float wdl[3] = {s[1], s[2], s[3]};
    softmax_in_place(wdl, 3);
    return ZeroValuesPov{std::tanh(s[0]), WDL{wdl[0], wdl[1], wdl[2]}, s[4]};
}

// common.rs:16-100 for the (scalars [n,5], policy [n,policy_len]) output form (:31-42)
template <class B, class P>
std::vector<ZeroEvaluation> decode_output(const P &policy_mapper, const B *boards, size_t n, const float *scalars,
                                          const float *policy_logits) {
    const size_t policy_len = policy_mapper.policy_len();
    std::vector<ZeroEvaluation> out;
    out.reserve(n);
    for (size_t bi = 0; bi < n; bi++) {
        ZeroEvaluation ev;
        ev.values = zero_values_from_scalars(scalars + bi * 5);  // :60-74
        auto moves = boards[bi].available_moves();               // :77: map_or(vec![], ...)
        if (moves) {
            ev.policy.reserve(moves->size());
            for (const auto &mv : *moves) {
                const size_t index = policy_mapper.move_to_index(boards[bi], mv);
                if (index >= policy_len) throw std::out_of_range("move_to_index out of range");
                ev.policy.push_back(policy_logits[bi * policy_len + index]);
            }
            softmax_in_place(ev.policy.data(), ev.policy.size());
        }
        out.push_back(std::move(ev));
    }
    return out;
}

// dummy.rs:16,44-60 and :130-148: uniform wdl and policy — the fake backend of the reference's tree tests
template <class B>
struct DummyNetwork : Network<B> {
    size_t max_batch_size() const override { return std::numeric_limits<size_t>::max(); }
    std::vector<ZeroEvaluation> evaluate_batch(const B *boards, size_t n) override {
        std::vector<ZeroEvaluation> out(n);
        for (size_t i = 0; i < n; i++) {
            out[i].values = ZeroValuesPov{0.0f, WDL{1.0f / 3, 1.0f / 3, 1.0f / 3}, 0.0f};
            auto moves = boards[i].available_moves();
            const size_t count = moves ? moves->size() : 0;
            out[i].policy.assign(count, count ? 1.0f / count : 0.0f);
        }
        return out;
    }
};

// ---- the other adapters behind the same trait (dummy.rs:18-148, multibatch.rs:10-49) ----

inline ZeroValuesPov uniform_values() { return ZeroValuesPov{0.0f, WDL{1.0f / 3, 1.0f / 3, 1.0f / 3}, 0.0f}; }  // dummy.rs:98-108
inline std::vector<float> uniform_policy(size_t available_moves) {                                                // :110-112
    return std::vector<float>(available_moves, available_moves ? 1.0f / (float)available_moves : 0.0f);
}

// uniform wdl, the inner network's policy (dummy.rs:18-23, 62-78)
template <class B, class N>
struct DummyValueNetwork : Network<B> {
    N inner;
    explicit DummyValueNetwork(N n) : inner(std::move(n)) {}
    size_t max_batch_size() const override { return inner.max_batch_size(); }
    std::vector<ZeroEvaluation> evaluate_batch(const B *boards, size_t n) override {
        auto out = inner.evaluate_batch(boards, n);
        for (auto &e : out) e.values = uniform_values();
        return out;
    }
};

// the inner network's values, uniform policy (dummy.rs:25-30, 80-96)
template <class B, class N>
struct DummyPolicyNetwork : Network<B> {
    N inner;
    explicit DummyPolicyNetwork(N n) : inner(std::move(n)) {}
    size_t max_batch_size() const override { return inner.max_batch_size(); }
    std::vector<ZeroEvaluation> evaluate_batch(const B *boards, size_t n) override {
        auto out = inner.evaluate_batch(boards, n);
        for (auto &e : out) e.policy = uniform_policy(e.policy.size());
        return out;
    }
};

// accepts a board wrapper W with `inner()` (MaxMovesBoard<B>) and passes the inner board along (dummy.rs:114-131)
template <class W, class B, class N>
struct MaxMovesNetwork : Network<W> {
    N inner;
    explicit MaxMovesNetwork(N n) : inner(std::move(n)) {}
    size_t max_batch_size() const override { return inner.max_batch_size(); }
    std::vector<ZeroEvaluation> evaluate_batch(const W *boards, size_t n) override {
        std::vector<B> in;
        in.reserve(n);
        for (size_t i = 0; i < n; i++) in.push_back(boards[i].inner());
        return inner.evaluate_batch(in.data(), n);
    }
};

// NetworkOrDummy = Either<N, DummyNetwork> (dummy.rs:133-148): what the executor holds after "UseDummyNetwork"
template <class B, class L, class R>
struct EitherNetwork : Network<B> {
    std::optional<L> left;
    std::optional<R> right;
    explicit EitherNetwork(L l) : left(std::move(l)) {}
    struct RightTag {};
    EitherNetwork(RightTag, R r) : right(std::move(r)) {}
    size_t max_batch_size() const override { return left ? left->max_batch_size() : right->max_batch_size(); }
    std::vector<ZeroEvaluation> evaluate_batch(const B *boards, size_t n) override {
        return left ? left->evaluate_batch(boards, n) : right->evaluate_batch(boards, n);
    }
};

// several instances of one network built for different batch sizes: a batch goes to the smallest that fits
// (multibatch.rs:10-49; with Kyanite every instance is planned for a fixed batch — kz_engine takes any batch up to its
// max_batch, so this is only needed by callers that already hold such a set)
template <class B, class I>
class MultiBatchNetwork : public Network<B> {
    std::vector<std::pair<size_t, I>> networks_;

  public:
    explicit MultiBatchNetwork(std::vector<std::pair<size_t, I>> networks) : networks_(std::move(networks)) {}
    template <class F>
    static MultiBatchNetwork build_sizes(const std::vector<size_t> &sizes, F f) {  // :19-22
        std::vector<std::pair<size_t, I>> nets;
        for (size_t s : sizes) nets.emplace_back(s, f(s));
        return MultiBatchNetwork(std::move(nets));
    }
    size_t used_network_index(size_t batch_size) const {  // :26-31
        size_t best = networks_.size();
        for (size_t i = 0; i < networks_.size(); i++)
            if (networks_[i].first >= batch_size && (best == networks_.size() || networks_[i].first < networks_[best].first)) best = i;
        if (best == networks_.size()) throw std::invalid_argument("No network for batch size " + std::to_string(batch_size));
        return best;
    }
    size_t used_batch_size(size_t batch_size) const { return networks_[used_network_index(batch_size)].first; }  // :33-35
    size_t max_batch_size() const override {                                                                      // :39-41
        size_t m = 0;
        for (auto &n : networks_) m = std::max(m, n.first);
        return m;
    }
    std::vector<ZeroEvaluation> evaluate_batch(const B *boards, size_t n) override {  // :43-48
        return networks_[used_network_index(n)].second.evaluate_batch(boards, n);
    }
};

}  // namespace kz::host
