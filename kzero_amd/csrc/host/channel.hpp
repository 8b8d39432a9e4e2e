// channel.hpp — bounded MPMC channel with flume's semantics, as far as the executor path uses them
// (rust/kz-core/src/network/job_channel.rs and rust/kz-selfplay/src/server/executor.rs use `flume::bounded`,
// `Sender::send`, `Receiver::{recv, try_recv}`, clone of both ends, disconnection on last drop, and `Selector`).
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <optional>
#include <vector>

namespace kz::host {

enum class RecvError { Disconnected };
enum class TryRecvError { Empty, Disconnected };

// Wakes a thread blocked in select2 on several channels.
struct Notifier {
    std::mutex m;
    std::condition_variable cv;
    unsigned long epoch = 0;
    void notify() {
        {
            std::lock_guard<std::mutex> g(m);
            epoch++;
        }
        cv.notify_all();
    }
};

template <class T>
struct ChannelCore {
    std::mutex m;
    std::condition_variable not_empty, not_full;
    std::deque<T> q;
    size_t cap;
    int senders = 0, receivers = 0;
    std::vector<Notifier *> watchers;
    explicit ChannelCore(size_t cap) : cap(cap) {}
    void wake_watchers() {
        for (Notifier *n : watchers) n->notify();
    }
};

template <class T>
class Sender {
    std::shared_ptr<ChannelCore<T>> c;

  public:
    Sender() = default;
    explicit Sender(std::shared_ptr<ChannelCore<T>> core) : c(std::move(core)) {
        std::lock_guard<std::mutex> g(c->m);
        c->senders++;
    }
    Sender(const Sender &o) : c(o.c) {
        if (c) {
            std::lock_guard<std::mutex> g(c->m);
            c->senders++;
        }
    }
    Sender(Sender &&o) noexcept : c(std::move(o.c)) {}
    Sender &operator=(Sender o) {
        std::swap(c, o.c);
        return *this;
    }
    ~Sender() { drop(); }
    void drop() {
        if (!c) return;
        bool last;
        {
            std::lock_guard<std::mutex> g(c->m);
            last = --c->senders == 0;
            if (last) c->wake_watchers();
        }
        if (last) c->not_empty.notify_all();
        c.reset();
    }
    // Blocks while the channel is full.  Returns false (SendError) when every receiver is gone.
    bool send(T v) const {
        std::unique_lock<std::mutex> g(c->m);
        c->not_full.wait(g, [&] { return c->q.size() < c->cap || c->receivers == 0; });
        if (c->receivers == 0) return false;
        c->q.push_back(std::move(v));
        c->wake_watchers();
        g.unlock();
        c->not_empty.notify_one();
        return true;
    }
    explicit operator bool() const { return (bool)c; }
};

template <class T>
class Receiver {
    std::shared_ptr<ChannelCore<T>> c;

  public:
    Receiver() = default;
    explicit Receiver(std::shared_ptr<ChannelCore<T>> core) : c(std::move(core)) {
        std::lock_guard<std::mutex> g(c->m);
        c->receivers++;
    }
    Receiver(const Receiver &o) : c(o.c) {
        if (c) {
            std::lock_guard<std::mutex> g(c->m);
            c->receivers++;
        }
    }
    Receiver(Receiver &&o) noexcept : c(std::move(o.c)) {}
    Receiver &operator=(Receiver o) {
        std::swap(c, o.c);
        return *this;
    }
    ~Receiver() { drop(); }
    void drop() {
        if (!c) return;
        bool last;
        {
            std::lock_guard<std::mutex> g(c->m);
            last = --c->receivers == 0;
        }
        if (last) c->not_full.notify_all();
        c.reset();
    }
    // Ok(value) or Err(Disconnected) once the queue is empty and every sender is gone.
    std::optional<T> recv() const {
        std::unique_lock<std::mutex> g(c->m);
        c->not_empty.wait(g, [&] { return !c->q.empty() || c->senders == 0; });
        if (c->q.empty()) return std::nullopt;
        T v = std::move(c->q.front());
        c->q.pop_front();
        g.unlock();
        c->not_full.notify_one();
        return v;
    }
    // value, or Empty / Disconnected in `err`
    std::optional<T> try_recv(TryRecvError &err) const {
        std::unique_lock<std::mutex> g(c->m);
        if (c->q.empty()) {
            err = c->senders == 0 ? TryRecvError::Disconnected : TryRecvError::Empty;
            return std::nullopt;
        }
        T v = std::move(c->q.front());
        c->q.pop_front();
        g.unlock();
        c->not_full.notify_one();
        return v;
    }
    // for select2
    bool ready_or_disconnected() const {
        std::lock_guard<std::mutex> g(c->m);
        return !c->q.empty() || c->senders == 0;
    }
    void watch(Notifier *n) const {
        std::lock_guard<std::mutex> g(c->m);
        c->watchers.push_back(n);
    }
    void unwatch(Notifier *n) const {
        std::lock_guard<std::mutex> g(c->m);
        for (auto it = c->watchers.begin(); it != c->watchers.end(); ++it)
            if (*it == n) {
                c->watchers.erase(it);
                break;
            }
    }
    explicit operator bool() const { return (bool)c; }
};

template <class T>
std::pair<Sender<T>, Receiver<T>> bounded(size_t cap) {
    auto core = std::make_shared<ChannelCore<T>>(cap ? cap : 1);
    return {Sender<T>(core), Receiver<T>(core)};
}

// flume::Selector over up to two receivers: blocks until one of the enabled receivers has a message or is
// disconnected and returns its index (0 or 1).  A disabled receiver (nullptr) is never selected.
template <class A, class B>
int select2(const Receiver<A> *a, const Receiver<B> *b) {
    Notifier n;
    if (a) a->watch(&n);
    if (b) b->watch(&n);
    int which = -1;
    {
        std::unique_lock<std::mutex> g(n.m);
        for (;;) {
            const unsigned long seen = n.epoch;
            g.unlock();
            if (a && a->ready_or_disconnected()) which = 0;
            else if (b && b->ready_or_disconnected()) which = 1;
            g.lock();
            if (which >= 0) break;
            n.cv.wait(g, [&] { return n.epoch != seen; });
        }
    }
    if (a) a->unwatch(&n);
    if (b) b->unwatch(&n);
    return which;
}

}  // namespace kz::host
