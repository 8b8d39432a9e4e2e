// job_channel.hpp — the generator -> executor job channel (rust/kz-core/src/network/job_channel.rs:9-88).
#pragma once
#include <future>
#include <stdexcept>
#include <vector>

#include "channel.hpp"

namespace kz::host {

// job_channel.rs:19-23
template <class X, class Y>
struct Job {
    std::vector<X> x;
    Sender<std::vector<Y>> sender;
};

template <class X, class Y>
class JobServer {
    Receiver<Job<X, Y>> receiver_;

  public:
    JobServer() = default;
    explicit JobServer(Receiver<Job<X, Y>> r) : receiver_(std::move(r)) {}
    const Receiver<Job<X, Y>> &receiver() const { return receiver_; }               // job_channel.rs:60-62
    Receiver<Job<X, Y>> into_receiver() { return std::move(receiver_); }            // job_channel.rs:64-66
};

template <class X, class Y>
class JobClient {
    Sender<Job<X, Y>> sender_;

  public:
    JobClient() = default;
    explicit JobClient(Sender<Job<X, Y>> s) : sender_(std::move(s)) {}

    // job_channel.rs:34-46: a bounded(1) reply channel per request; an empty request short-circuits (:37-40)
    Receiver<std::vector<Y>> map(std::vector<X> x) const {
        auto [sender, receiver] = bounded<std::vector<Y>>(1);
        if (x.empty()) {
            sender.send({});
        } else {
            if (!sender_.send(Job<X, Y>{std::move(x), std::move(sender)}))
                throw std::runtime_error("job channel: executor is gone");  // `.unwrap()` on SendError
        }
        return receiver;
    }

    // not in the reference: the reply goes to a channel the caller owns, so that one thread can stand for many games and
    // take their replies in the order they arrive (the reference's games are independent futures; a thread that waited
    // for its requests in submission order could starve the executors' partial batches: tests/cpp/bench_executor.cpp)
    void map_into(std::vector<X> x, Sender<std::vector<Y>> reply) const {
        if (x.empty()) {
            reply.send({});
        } else if (!sender_.send(Job<X, Y>{std::move(x), std::move(reply)})) {
            throw std::runtime_error("job channel: executor is gone");
        }
    }

    // job_channel.rs:48-50
    std::vector<Y> map_blocking(std::vector<X> x) const {
        auto r = map(std::move(x)).recv();
        if (!r) throw std::runtime_error("job channel: reply sender dropped");
        return std::move(*r);
    }

    // job_channel.rs:52-54 — the Rust version is a Future polled by the generator's thread pool; the C++ mirror
    // hands back a deferred std::future (the wait happens in get())
    std::future<std::vector<Y>> map_async(std::vector<X> x) const {
        auto receiver = map(std::move(x));
        return std::async(std::launch::deferred, [receiver]() {
            auto r = receiver.recv();
            if (!r) throw std::runtime_error("job channel: reply sender dropped");
            return std::move(*r);
        });
    }

    // job_channel.rs:56-59
    std::future<Y> map_async_single(X x) const {
        std::vector<X> v;
        v.push_back(std::move(x));
        auto fut = std::make_shared<std::future<std::vector<Y>>>(map_async(std::move(v)));
        return std::async(std::launch::deferred, [fut]() {
            auto y = fut->get();
            if (y.size() != 1) throw std::runtime_error("expected a single result");
            return std::move(y[0]);
        });
    }
};

// job_channel.rs:25-32
template <class X, class Y>
std::pair<JobClient<X, Y>, JobServer<X, Y>> job_pair(size_t cap) {
    auto [s, r] = bounded<Job<X, Y>>(cap);
    return {JobClient<X, Y>(std::move(s)), JobServer<X, Y>(std::move(r))};
}

}  // namespace kz::host
