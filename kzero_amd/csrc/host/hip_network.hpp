// hip_network.hpp — `HipNetwork<B, M>`: the executor behind the `Network` contract, C++ mirror of the Rust shim in
// kzero_amd/rust/hip.rs.  Same shape as `CudaNetwork` (rust/kz-core/src/network/cudnn.rs:18-88): encode every board
// with the mapper, run the engine, decode the outputs.  Talks to the product only through the C ABI.
#pragma once
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <iterator>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>

#include <time.h>

#include "../../../include/kz_hip.h"
#include "mapping.hpp"
#include "network.hpp"

namespace kz::host {

// A board type may carry the policy indices of its available moves, computed where the board was made (a generator
// thread): `const std::vector<int32_t> *policy_indices() const`, null when it does not.  The executor thread then
// neither generates the moves nor looks them up for the device-side decode.
template <class T, class = void>
struct has_policy_indices : std::false_type {};
template <class T>
struct has_policy_indices<T, std::void_t<decltype(std::declval<const T &>().policy_indices())>> : std::true_type {};

// CPU time of the calling thread (measurement: how much of an executor thread's time is work, how much is the HIP
// runtime spinning in kz_engine_wait*)
inline uint64_t thread_cpu_ns() {
    timespec ts{};
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

inline void kz_check(int rc) {
    if (rc != 0) throw std::runtime_error(std::string("kzhip: ") + kz_last_error());  // the reference panics
}

// One persistent helper thread that runs a job handed over by its owner while the owner does its own share of the same
// work (HipNetwork::prepare: half of a batch's board encoding and move-list building each).  start() hands over, finish()
// waits and re-throws what the job threw.  Not part of the reference: its executor thread does all of a batch's host work
// alone (cudnn.rs:55-87), which at 500k evals/s per GPU is most of a core.
class PrepHelper {
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false;
    std::exception_ptr err_;
    std::thread th_;

    void run() {
        std::unique_lock<std::mutex> l(m_);
        for (;;) {
            cv_.wait(l, [&] { return quit_ || job_; });
            if (quit_) return;
            std::function<void()> j = std::move(job_);
            job_ = nullptr;
            l.unlock();
            std::exception_ptr err;
            try {
                j();
            } catch (...) {
                err = std::current_exception();
            }
            cpu_ns = thread_cpu_ns();
            l.lock();
            err_ = err;
            busy_ = false;
            cv_.notify_all();
        }
    }

  public:
    std::atomic<uint64_t> cpu_ns{0};  // CPU time of the helper thread so far (measurement)
    PrepHelper() : th_([this] { run(); }) {}
    PrepHelper(const PrepHelper &) = delete;
    ~PrepHelper() {
        {
            std::lock_guard<std::mutex> l(m_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void start(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = std::move(job);
            busy_ = true;
        }
        cv_.notify_all();
    }
    void finish() {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !busy_; });
        if (err_) {
            std::exception_ptr e = err_;
            err_ = nullptr;
            std::rethrow_exception(e);
        }
    }
};

// the `Arc<Graph>` of the reference: immutable, shared by every executor
class HipModel {
    kz_model *ptr_ = nullptr;

  public:
    kz_model_info info{};
    explicit HipModel(const std::string &path) {
        kz_check(kz_model_load(path.c_str(), &ptr_));
        kz_check(kz_model_get_info(ptr_, &info));
    }
    HipModel(const void *blob, size_t len) {
        kz_check(kz_model_load_memory(blob, len, &ptr_));
        kz_check(kz_model_get_info(ptr_, &info));
    }
    HipModel(const HipModel &) = delete;
    HipModel &operator=(const HipModel &) = delete;
    ~HipModel() { kz_model_free(ptr_); }
    const kz_model *get() const { return ptr_; }
};

template <class B, class M>
class HipNetwork : public Network<B> {
    M mapper_;
    std::shared_ptr<const HipModel> model_;
    kz_engine *engine_ = nullptr;
    size_t max_batch_size_;
    std::vector<uint8_t> bits_;
    std::vector<float> scalars_in_;
    // asynchronous pair: the boards of every batch in flight (decode_output needs their legal moves), oldest first
    std::vector<B> pending_boards_[KZ_ENGINE_SLOTS];
    int next_slot_ = 0, oldest_slot_ = 0, in_flight_ = 0;
    // decode_output on the device (kz_engine_submit_packed_decoded / kz_engine_wait_decoded): the CSR move lists of
    // the batches in flight
    bool device_decode_ = false;
    std::vector<int64_t> move_offsets_[KZ_ENGINE_SLOTS];
    std::vector<int32_t> move_indices_;

    // ---- a batch's host work before the launch: encode_input of every board (cudnn.rs:61-64) and, with the decode on the
    // device, move_to_index of every available move as CSR (common.rs:77-86 up to the gather).  The batch is cut into
    // 1 + helpers contiguous ranges; this thread takes the first, a PrepHelper each of the others; the ranges meet in the
    // staging vectors in board order.
    struct PrepRange {
        std::vector<float> scalars;
        std::vector<int64_t> counts;   // moves per board
        std::vector<int32_t> indices;
    };
    std::vector<PrepRange> ranges_ = std::vector<PrepRange>(1);
    std::vector<std::unique_ptr<PrepHelper>> helpers_;

    void prep_range(const B *boards, size_t lo, size_t hi, bool moves, PrepRange &out) {
        const size_t bool_count = input_bool_len(mapper_), bits_bytes = (bool_count + 7) / 8, policy_len = mapper_.policy_len();
        out.scalars.clear();
        out.counts.clear();
        out.indices.clear();
        BitBuffer buffer(bool_count);
        for (size_t bi = lo; bi < hi; bi++) {
            buffer.clear();
            mapper_.encode_input(buffer, out.scalars, boards[bi]);
            if (buffer.len() != bool_count) throw std::logic_error("mapper wrote the wrong number of bools");
            std::copy(buffer.storage().begin(), buffer.storage().begin() + bits_bytes, bits_.begin() + bi * bits_bytes);
            if (!moves) continue;
            const size_t before = out.indices.size();
            bool carried = false;
            if constexpr (has_policy_indices<B>::value) {
                if (const std::vector<int32_t> *idx = boards[bi].policy_indices()) {
                    for (int32_t index : *idx)
                        if (index < 0 || (size_t)index >= policy_len) throw std::out_of_range("policy index out of range");
                    out.indices.insert(out.indices.end(), idx->begin(), idx->end());
                    carried = true;
                }
            }
            if (!carried) {
                auto available = boards[bi].available_moves();
                if (available)
                    for (const auto &mv : *available) {
                        const size_t index = mapper_.move_to_index(boards[bi], mv);
                        if (index >= policy_len) throw std::out_of_range("move_to_index out of range");
                        out.indices.push_back((int32_t)index);
                    }
            }
            out.counts.push_back((int64_t)(out.indices.size() - before));
        }
    }

    // returns bits_bytes; fills bits_, scalars_in_ and (moves) offsets + move_indices_
    size_t prepare(const B *boards, size_t n, bool moves, std::vector<int64_t> *offsets) {
        const size_t parts = std::min(ranges_.size(), std::max<size_t>(1, n / 16));  // (a handful of boards: not worth a hand-over)
        auto cut = [&](size_t k) { return n * k / parts; };
        for (size_t k = 1; k < parts; k++)
            helpers_[k - 1]->start([this, boards, moves, k, lo = cut(k), hi = cut(k + 1)] { prep_range(boards, lo, hi, moves, ranges_[k]); });
        std::exception_ptr first;
        const uint64_t t0 = thread_cpu_ns();
        try {
            prep_range(boards, 0, cut(1), moves, ranges_[0]);
        } catch (...) {
            first = std::current_exception();
        }
        prep_own_cpu_ns += thread_cpu_ns() - t0;
        for (size_t k = 1; k < parts; k++) {  // every helper is waited for, whatever happened: the boards are the caller's
            try {
                helpers_[k - 1]->finish();
            } catch (...) {
                if (!first) first = std::current_exception();
            }
        }
        if (first) std::rethrow_exception(first);
        const uint64_t t1 = thread_cpu_ns();
        scalars_in_.clear();
        if (moves) {
            offsets->assign(1, 0);
            move_indices_.clear();
        }
        for (size_t k = 0; k < parts; k++) {
            const PrepRange &r = ranges_[k];
            scalars_in_.insert(scalars_in_.end(), r.scalars.begin(), r.scalars.end());
            if (!moves) continue;
            move_indices_.insert(move_indices_.end(), r.indices.begin(), r.indices.end());
            for (int64_t c : r.counts) offsets->push_back(offsets->back() + c);
        }
        merge_cpu_ns += thread_cpu_ns() - t1;
        return (input_bool_len(mapper_) + 7) / 8;
    }

    // values [n,5] (already tanh / softmax) + probabilities parallel to the move lists -> evaluations
    static std::vector<ZeroEvaluation> assemble_decoded(size_t n, const std::vector<int64_t> &offsets, const float *values,
                                                        const float *probs) {
        std::vector<ZeroEvaluation> out(n);
        for (size_t bi = 0; bi < n; bi++) {
            const float *v = values + bi * 5;
            out[bi].values = ZeroValuesPov{v[0], WDL{v[1], v[2], v[3]}, v[4]};
            out[bi].policy.assign(probs + offsets[bi], probs + offsets[bi + 1]);
        }
        return out;
    }

  public:
    // CPU time this thread has spent inside kz_engine_wait* so far: the HIP runtime polls the event, so an executor
    // thread shows ~100 % CPU however little work it does; (thread CPU - this) is its work
    uint64_t wait_cpu_ns = 0;
    // ... and where the rest goes (measurement, tests/cpp/bench_executor.cpp): this thread's share of a batch's preparation, the
    // merge of the ranges, the engine's submit call (copies into pinned staging + the launch), building the evaluations
    uint64_t prep_own_cpu_ns = 0, merge_cpu_ns = 0, submit_cpu_ns = 0, assemble_cpu_ns = 0;
    // cudnn.rs:29-43 (check_graph_shapes: common.rs:165-198)
    HipNetwork(M mapper, std::shared_ptr<const HipModel> model, size_t max_batch_size, int device, int dtype)
        : mapper_(mapper), model_(std::move(model)), max_batch_size_(max_batch_size) {
        const kz_model_info &info = model_->info;
        auto shape = input_full_shape(mapper_);
        if ((size_t)info.input_channels != shape[0] || (size_t)info.board_h != shape[1] || (size_t)info.board_w != shape[2])
            throw std::invalid_argument("Input shape mismatch between model and mapper");
        if ((size_t)info.input_scalar_channels != mapper_.input_scalar_count())
            throw std::invalid_argument("Scalar plane count mismatch between model and mapper");
        if ((size_t)info.policy_len != mapper_.policy_len()) throw std::invalid_argument("Wrong policy shape");
        kz_check(kz_engine_create(model_->get(), device, (int)max_batch_size, dtype, &engine_));
        bits_.resize(max_batch_size * (size_t)info.bits_bytes);
    }
    HipNetwork(HipNetwork &&o) noexcept
        : mapper_(o.mapper_), model_(std::move(o.model_)), engine_(o.engine_), max_batch_size_(o.max_batch_size_),
          bits_(std::move(o.bits_)), scalars_in_(std::move(o.scalars_in_)), next_slot_(o.next_slot_),
          oldest_slot_(o.oldest_slot_), in_flight_(o.in_flight_), device_decode_(o.device_decode_), ranges_(std::move(o.ranges_)),
          helpers_(std::move(o.helpers_)), wait_cpu_ns(o.wait_cpu_ns) {
        for (int i = 0; i < KZ_ENGINE_SLOTS; i++) {
            pending_boards_[i] = std::move(o.pending_boards_[i]);
            move_offsets_[i] = std::move(o.move_offsets_[i]);
        }
        o.engine_ = nullptr;
    }
    HipNetwork(const HipNetwork &) = delete;
    ~HipNetwork() override { kz_engine_destroy(engine_); }

    size_t max_batch_size() const override { return max_batch_size_; }

    // true: decode_output runs on the GPU (legal-move gather + softmax, tanh, wdl); 0.2 KB instead of 7.5 KB per chess
    // evaluation cross PCIe and this thread does no softmax.  Same results to f32 rounding (device expf/tanhf).
    void set_device_decode(bool on) {
        if (in_flight_ != 0) throw std::logic_error("set_device_decode while batches are in flight");
        device_decode_ = on;
    }

    // Helper threads for a batch's host work (encode_input, move lists): 0 = all of it on the calling executor thread like
    // the reference; 1 (hip.rs's default) halves what the executor thread does per batch.  Results are identical.
    void set_prep_helpers(size_t n) {
        if (in_flight_ != 0) throw std::logic_error("set_prep_helpers while batches are in flight");
        helpers_.clear();
        for (size_t i = 0; i < n; i++) helpers_.push_back(std::make_unique<PrepHelper>());
        ranges_.assign(n + 1, PrepRange{});
    }
    // CPU time the helper threads have used so far (measurement)
    uint64_t helper_cpu_ns() const {
        uint64_t t = 0;
        for (const auto &h : helpers_) t += h->cpu_ns.load();
        return t;
    }

    // cudnn.rs:55-87
    std::vector<ZeroEvaluation> evaluate_batch(const B *boards, size_t n) override {
        if (n > max_batch_size_) throw std::invalid_argument("batch_size <= max_batch_size");  // assert!, :58
        if (n == 0) return {};
        if (in_flight_ != 0) throw std::logic_error("evaluate_batch while submitted batches are in flight");
        const size_t bits_bytes = prepare(boards, n, device_decode_, &move_offsets_[0]);
        if (device_decode_) {
            const float *values = nullptr, *probs = nullptr;
            kz_check(kz_engine_submit_packed_decoded(engine_, 0, bits_.data(), bits_bytes, scalars_in_.data(), (int)n,
                                                     move_offsets_[0].data(), move_indices_.data()));
            const uint64_t w0 = thread_cpu_ns();
            kz_check(kz_engine_wait_decoded(engine_, 0, &values, &probs));
            wait_cpu_ns += thread_cpu_ns() - w0;
            return assemble_decoded(n, move_offsets_[0], values, probs);
        }
        // kz_engine_eval_packed without its copy into caller buffers: decode reads the pinned staging directly
        const float *scalars = nullptr, *policy = nullptr;
        kz_check(kz_engine_submit_packed(engine_, 0, bits_.data(), bits_bytes, scalars_in_.data(), (int)n));
        const uint64_t w0 = thread_cpu_ns();
        kz_check(kz_engine_wait_view(engine_, 0, &scalars, &policy));
        wait_cpu_ns += thread_cpu_ns() - w0;
        return decode_output(mapper_, boards, n, scalars, policy);
    }

    // ---- asynchronous pair for pipelined_executor_loop (kz_engine_submit_packed / kz_engine_wait) ----
    static constexpr size_t max_in_flight() { return KZ_ENGINE_SLOTS; }
    size_t batches_in_flight() const { return (size_t)in_flight_; }

    // encode + hand the batch to the engine; returns while the GPU works.  The boards are MOVED out of `boards`
    // (decode_output needs their available moves when the results come back); the pointer itself is not retained.
    void submit_batch(B *boards, size_t n) {
        if (n == 0 || n > max_batch_size_) throw std::invalid_argument("0 < batch_size <= max_batch_size");
        if (in_flight_ == KZ_ENGINE_SLOTS) throw std::logic_error("every engine slot is in flight");
        const size_t bits_bytes = prepare(boards, n, device_decode_, &move_offsets_[next_slot_]);
        // the engine copies its inputs to pinned staging before submit returns (include/kz_hip.h)
        if (device_decode_) {
            const uint64_t s0 = thread_cpu_ns();
            kz_check(kz_engine_submit_packed_decoded(engine_, next_slot_, bits_.data(), bits_bytes, scalars_in_.data(),
                                                     (int)n, move_offsets_[next_slot_].data(), move_indices_.data()));
            submit_cpu_ns += thread_cpu_ns() - s0;
            pending_boards_[next_slot_].clear();  // the move lists are all the decode needs
        } else {
            kz_check(kz_engine_submit_packed(engine_, next_slot_, bits_.data(), bits_bytes, scalars_in_.data(), (int)n));
            pending_boards_[next_slot_].assign(std::make_move_iterator(boards), std::make_move_iterator(boards + n));
        }
        next_slot_ = (next_slot_ + 1) % KZ_ENGINE_SLOTS;
        in_flight_++;
    }

    // results of the OLDEST submitted batch
    std::vector<ZeroEvaluation> wait_batch() {
        if (in_flight_ == 0) throw std::logic_error("wait_batch with nothing in flight");
        const int slot = oldest_slot_;
        oldest_slot_ = (oldest_slot_ + 1) % KZ_ENGINE_SLOTS;
        in_flight_--;
        if (device_decode_) {
            const float *values = nullptr, *probs = nullptr;
            const uint64_t w0 = thread_cpu_ns();
            kz_check(kz_engine_wait_decoded(engine_, slot, &values, &probs));
            const uint64_t w1 = thread_cpu_ns();
            wait_cpu_ns += w1 - w0;
            auto out = assemble_decoded(move_offsets_[slot].size() - 1, move_offsets_[slot], values, probs);
            assemble_cpu_ns += thread_cpu_ns() - w1;
            return out;
        }
        // decode straight from the engine's pinned staging (valid until the next submit on this slot)
        const float *scalars = nullptr, *policy = nullptr;
        const uint64_t w0 = thread_cpu_ns();
        kz_check(kz_engine_wait_view(engine_, slot, &scalars, &policy));
        wait_cpu_ns += thread_cpu_ns() - w0;
        const std::vector<B> &boards = pending_boards_[slot];
        return decode_output(mapper_, boards.data(), boards.size(), scalars, policy);
    }
};

}  // namespace kz::host
