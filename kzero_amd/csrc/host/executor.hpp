// executor.hpp — the batching executor loop (rust/kz-selfplay/src/server/executor.rs:21-342), restated in C++ with the
// same generic shape: <G, N, X, Y> with closures `load_network: G -> N` and `evaluate_batch: (N&, X*, n) -> vector<Y>`.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <optional>
#include <string>
#include <vector>

#include "job_channel.hpp"

#define KZ_HOST_ASSERT(cond, msg)                                                             \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            std::fprintf(stderr, "%s:%d: assertion failed: %s (%s)\n", __FILE__, __LINE__, #cond, msg); \
            std::abort(); /* the reference panics */                                          \
        }                                                                                     \
    } while (0)

namespace kz::host {

// executor.rs:21-25
struct RunCondition {
    enum Kind { FullBatch, JobCount, Any } kind;
    size_t count = 0;
    static RunCondition full_batch() { return {FullBatch, 0}; }
    static RunCondition job_count(size_t n) { return {JobCount, n}; }
    static RunCondition any() { return {Any, 0}; }
};

// executor.rs:176-302
template <class X, class Y>
class ExecutorState {
    std::vector<X> x_;  // pending inputs, x_[head_..] (a VecDeque in the reference)
    size_t head_ = 0;
    std::deque<std::pair<size_t, Sender<std::vector<Y>>>> senders_;
    std::deque<Y> leftover_y_;
    // pipelined loop only: inputs handed to the network whose results are still outstanding (x_[head_, head_ +
    // submitted_)), and for every queued job the running index of its last input (to count the jobs not yet submitted)
    size_t submitted_ = 0;
    unsigned long long pushed_total_ = 0, submitted_total_ = 0;
    std::deque<unsigned long long> job_ends_;

  public:
    size_t items_to_eval() const { return x_.size() - head_ - submitted_; }  // :210-212 (minus what is in flight)
    size_t items_in_flight() const { return submitted_; }
    size_t items_waiting_for_send() const { return leftover_y_.size(); }  // :214-216
    size_t items_to_send() const {                                        // :218-220
        size_t n = 0;
        for (auto &s : senders_) n += s.first;
        return n;
    }
    size_t sender_count() const { return senders_.size(); }

    void check_invariants() const {  // :202-208
        KZ_HOST_ASSERT(!can_fill_next_sender(), "a reply that could have been sent is still queued");
        KZ_HOST_ASSERT(items_to_eval() + items_in_flight() + items_waiting_for_send() == items_to_send(), "item accounting");
    }

    void push_job(Job<X, Y> job) {  // :222-238
        if (job.x.empty()) {
            // never queue an empty sender (:229-231)
            job.sender.send({});
        } else {
            senders_.emplace_back(job.x.size(), std::move(job.sender));
            pushed_total_ += job.x.size();
            job_ends_.push_back(pushed_total_);
            for (auto &v : job.x) x_.push_back(std::move(v));
        }
        check_invariants();
    }

    bool should_eval(RunCondition cond, size_t max_batch_size) const {  // :240-253
        if (items_to_eval() == 0) return false;
        if (items_to_eval() >= max_batch_size) return true;
        switch (cond.kind) {
            case RunCondition::FullBatch: return false;
            case RunCondition::JobCount: return job_ends_.size() >= cond.count;  // == senders_.len() with nothing in flight
            case RunCondition::Any: return items_to_eval() > 0;
        }
        return false;
    }

    // the first <= max_batch_size pending inputs, contiguous (:255-264)
    std::pair<const X *, size_t> get_batch(size_t max_batch_size) const {
        const size_t n = std::min(items_to_eval(), max_batch_size);
        KZ_HOST_ASSERT(n != 0, "empty batch");
        return {x_.data() + head_ + submitted_, n};
    }

    // pipelined loop: the same batch, mutable — the network may move the inputs out (they are only counted from
    // here on: respond_batch never reads x_)
    std::pair<X *, size_t> take_batch(size_t max_batch_size) {
        auto [data, n] = get_batch(max_batch_size);
        return {const_cast<X *>(data), n};
    }

    // pipelined loop: the batch get_batch returned has been handed to the network; its results arrive later, in
    // submission order, through respond_batch
    void mark_submitted(size_t n) {
        KZ_HOST_ASSERT(n <= items_to_eval(), "submitted more than was pending");
        submitted_ += n;
        note_covered(n);
    }

    bool can_fill_next_sender() const {  // :266-274
        if (senders_.empty()) {
            KZ_HOST_ASSERT(leftover_y_.empty(), "results without a receiver");
            return false;
        }
        return leftover_y_.size() >= senders_.front().first;
    }

    // distribute the results over the senders in order; a failed send is ignored (:276-301)
    void respond_batch(std::vector<Y> batch_y) {
        const size_t batch_size = batch_y.size();
        if (submitted_ > 0) {  // results of the oldest batch in flight
            KZ_HOST_ASSERT(batch_size <= submitted_, "more results than were in flight");
            submitted_ -= batch_size;
        } else {  // the synchronous loop answers straight from the pending inputs
            KZ_HOST_ASSERT(batch_size <= items_to_eval(), "more results than inputs");
            note_covered(batch_size);
        }
        head_ += batch_size;
        if (head_ == x_.size()) {
            x_.clear();
            head_ = 0;
        } else if (head_ > 4096 && head_ * 2 > x_.size()) {
            x_.erase(x_.begin(), x_.begin() + head_);
            head_ = 0;
        }
        if (leftover_y_.empty() && !senders_.empty() && senders_.front().first == batch_size) {
            // shortcut: the whole vector goes to one sender (:285-287)
            senders_.front().second.send(std::move(batch_y));
            senders_.pop_front();
        } else {
            for (auto &y : batch_y) leftover_y_.push_back(std::move(y));
            while (can_fill_next_sender()) {
                auto [count, sender] = std::move(senders_.front());
                senders_.pop_front();
                std::vector<Y> block;
                block.reserve(count);
                for (size_t i = 0; i < count; i++) {
                    block.push_back(std::move(leftover_y_.front()));
                    leftover_y_.pop_front();
                }
                sender.send(std::move(block));
            }
        }
        check_invariants();
    }

  private:
    // n more inputs have gone to the network: jobs whose last input is among them no longer count for JobCount
    void note_covered(size_t n) {
        submitted_total_ += n;
        while (!job_ends_.empty() && job_ends_.front() <= submitted_total_) job_ends_.pop_front();
    }
};

// Hooks for tests/tracing; the reference brackets these points with superluminal events ("wait"/"run"/"reply"/
// "load"/"drop", executor.rs:63-65,310-317,328-339).
struct ExecutorEvents {
    virtual ~ExecutorEvents() = default;
    virtual void on_drop_network() {}
    virtual void on_load_network() {}
    virtual void on_eval(size_t /*batch*/) {}
};

template <class X, class Y, class N, class Eval>
void run_eval(ExecutorState<X, Y> &state, N &network, Eval &evaluate_batch, size_t max_batch_size,
              ExecutorEvents *events) {  // :304-318
    auto [data, n] = state.get_batch(max_batch_size);
    if (events) events->on_eval(n);
    std::vector<Y> batch_y = evaluate_batch(network, data, n);
    KZ_HOST_ASSERT(batch_y.size() == n, "evaluate_batch must return one result per input");
    state.respond_batch(std::move(batch_y));
}

// executor.rs:27-146.  graph_receiver carries Option<G>: nullopt = "wait for a new network" (drop the current one).
template <class G, class N, class X, class Y, class Load, class Eval>
void batched_executor_loop(size_t max_batch_size, RunCondition run_condition, Receiver<std::optional<G>> graph_receiver,
                           JobServer<X, Y> server, Load load_network, Eval evaluate_batch,
                           ExecutorEvents *events = nullptr) {
    KZ_HOST_ASSERT(max_batch_size != 0, "got batch size 0");
    Receiver<Job<X, Y>> job_receiver = server.into_receiver();
    ExecutorState<X, Y> state;
    std::optional<N> network;
    // a separate flag so that the disconnection event itself is handled exactly once (:43-45)
    bool graph_disconnected = false;

    auto handle_new_graph = [&](std::optional<G> graph) {  // :320-342
        if (network) {  // drop the previous network first, to save GPU memory
            if (events) events->on_drop_network();
            network.reset();
        }
        if (graph) {
            if (events) events->on_load_network();
            network.emplace(load_network(std::move(*graph)));
        }
    };

    for (;;) {
        KZ_HOST_ASSERT(network.has_value() || !graph_disconnected, "nothing left to wait on");
        // wait for graphs only while that channel is open; for jobs only once there is a network (:52-60)
        const int which = select2(graph_disconnected ? nullptr : &graph_receiver, network ? &job_receiver : nullptr);

        if (which == 0) {
            TryRecvError err = TryRecvError::Empty;
            auto msg = graph_receiver.try_recv(err);
            if (msg) {
                handle_new_graph(std::move(*msg));
                continue;
            }
            if (err == TryRecvError::Empty) continue;  // somebody else took it
            // Message::Graph(Err(Disconnected)) (:119-143)
            if (network) {
                graph_disconnected = true;  // keep evaluating with the final network
                continue;
            }
            KZ_HOST_ASSERT(state.items_to_eval() == 0, "graph disconnected but items are still pending");
            auto extra = job_receiver.recv();  // wait for the job channel to disconnect as well
            KZ_HOST_ASSERT(!extra.has_value(), "got a new job after graph disconnection");
            return;
        }

        // Message::Job
        TryRecvError err = TryRecvError::Empty;
        auto job = job_receiver.try_recv(err);
        if (job) {
            state.push_job(std::move(*job));
            // greedily take what is already queued, up to a full batch (:83-92)
            while (state.items_to_eval() < max_batch_size) {
                auto more = job_receiver.try_recv(err);
                if (!more) break;  // Empty: done; Disconnected: the next select handles it
                state.push_job(std::move(*more));
            }
            if (state.should_eval(run_condition, max_batch_size))
                run_eval(state, *network, evaluate_batch, max_batch_size, events);
            continue;
        }
        if (err == TryRecvError::Empty) continue;
        // the job channel has disconnected: evaluate what is left, then exit (:101-115)
        KZ_HOST_ASSERT(state.items_to_eval() < max_batch_size, "a full batch was left unevaluated");
        if (state.items_to_eval() > 0) run_eval(state, *network, evaluate_batch, max_batch_size, events);
        KZ_HOST_ASSERT(state.items_to_eval() == 0 && state.items_to_send() == 0, "leftovers at exit");
        return;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the same loop with up to `depth` batches in flight on ONE executor thread, for a network with an
// asynchronous pair `submit(N&, X*, n)` (may move the inputs out) / `wait(N&) -> vector<Y>` (results of the OLDEST
// submitted batch).  Channel
// semantics are those of batched_executor_loop: the same RunCondition, replies in job order, a new graph first drains
// what is in flight on the old network, a disconnected job channel evaluates the remainder and returns.  What changes
// is only when the thread blocks: while the GPU works on a batch the thread keeps collecting jobs and encodes and
// submits the next batch; it blocks on the channels only when nothing is in flight, and on the GPU only when it
// cannot submit (nothing ready, or `depth` batches out).  This replaces `gpu_threads_per_device` blocking threads
// (rust/Readme.md:51; server_alphazero.rs:89-121) by one thread per device.
template <class G, class N, class X, class Y, class Load, class Submit, class Wait>
void pipelined_executor_loop(size_t max_batch_size, size_t depth, RunCondition run_condition,
                             Receiver<std::optional<G>> graph_receiver, JobServer<X, Y> server, Load load_network,
                             Submit submit_batch, Wait wait_batch, ExecutorEvents *events = nullptr) {
    KZ_HOST_ASSERT(max_batch_size != 0, "got batch size 0");
    KZ_HOST_ASSERT(depth != 0, "got pipeline depth 0");
    Receiver<Job<X, Y>> job_receiver = server.into_receiver();
    ExecutorState<X, Y> state;
    std::optional<N> network;
    bool graph_disconnected = false, jobs_disconnected = false;
    std::deque<size_t> in_flight;  // batch sizes, oldest first

    auto submit_one = [&]() {
        auto [data, n] = state.take_batch(max_batch_size);
        if (events) events->on_eval(n);
        submit_batch(*network, data, n);
        state.mark_submitted(n);
        in_flight.push_back(n);
    };
    auto finish_oldest = [&]() {
        std::vector<Y> batch_y = wait_batch(*network);
        KZ_HOST_ASSERT(batch_y.size() == in_flight.front(), "wait_batch must return one result per submitted input");
        in_flight.pop_front();
        state.respond_batch(std::move(batch_y));
    };
    auto handle_new_graph = [&](std::optional<G> graph) {
        while (!in_flight.empty()) finish_oldest();  // the old network answers what it was given
        if (network) {
            if (events) events->on_drop_network();
            network.reset();
        }
        if (graph) {
            if (events) events->on_load_network();
            network.emplace(load_network(std::move(*graph)));
        }
    };
    // non-blocking: everything already queued, up to a full batch beyond what is in flight (:83-92)
    auto drain_jobs = [&]() {
        TryRecvError err = TryRecvError::Empty;
        while (state.items_to_eval() < max_batch_size) {
            auto more = job_receiver.try_recv(err);
            if (!more) {
                if (err == TryRecvError::Disconnected) jobs_disconnected = true;
                break;
            }
            state.push_job(std::move(*more));
        }
    };

    for (;;) {
        KZ_HOST_ASSERT(network.has_value() || !graph_disconnected, "nothing left to wait on");

        if (!in_flight.empty()) {
            // The GPU is busy: never block on the channels.  A new graph takes effect between batches, as in the
            // synchronous loop (it is seen once the running evaluate_batch returns).
            if (!graph_disconnected) {
                TryRecvError err = TryRecvError::Empty;
                auto msg = graph_receiver.try_recv(err);
                if (msg) {
                    handle_new_graph(std::move(*msg));
                    continue;
                }
                if (err == TryRecvError::Disconnected) graph_disconnected = true;  // keep the final network (:119-143)
            }
            if (!jobs_disconnected) drain_jobs();
            const bool ready = jobs_disconnected ? state.items_to_eval() > 0 : state.should_eval(run_condition, max_batch_size);
            if (ready && in_flight.size() < depth) submit_one();
            else finish_oldest();
            continue;
        }

        if (jobs_disconnected) {
            // the job channel has disconnected: evaluate what is left, then exit (:101-115)
            while (state.items_to_eval() > 0) {
                submit_one();
                finish_oldest();
            }
            KZ_HOST_ASSERT(state.items_to_eval() == 0 && state.items_to_send() == 0, "leftovers at exit");
            return;
        }

        // nothing in flight: block exactly like the synchronous loop (:52-60)
        const int which = select2(graph_disconnected ? nullptr : &graph_receiver, network ? &job_receiver : nullptr);
        if (which == 0) {
            TryRecvError err = TryRecvError::Empty;
            auto msg = graph_receiver.try_recv(err);
            if (msg) {
                handle_new_graph(std::move(*msg));
                continue;
            }
            if (err == TryRecvError::Empty) continue;
            if (network) {
                graph_disconnected = true;
                continue;
            }
            KZ_HOST_ASSERT(state.items_to_eval() == 0, "graph disconnected but items are still pending");
            auto extra = job_receiver.recv();
            KZ_HOST_ASSERT(!extra.has_value(), "got a new job after graph disconnection");
            return;
        }
        drain_jobs();
        if (!jobs_disconnected && state.should_eval(run_condition, max_batch_size)) submit_one();
    }
}

}  // namespace kz::host
