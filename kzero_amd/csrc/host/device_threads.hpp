// device_threads.hpp — per-device construction of the executor side
// (rust/kz-selfplay/src/server/server_alphazero.rs:32-124, `AlphaZeroSpecialization::spawn_device_threads`):
// one job channel per device, `gpu_threads` OS threads each running batched_executor_loop with its own network
// (engines of one device share the uploaded weights), every executed batch reported as ExpandEvals(real, potential).
// The generator side (MCTS, futures thread pool) stays the caller's.
#pragma once
#include <atomic>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include <time.h>

#include "executor.hpp"
#include "hip_network.hpp"

namespace kz::host {

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }  // kz_util::math::ceil_div

// rust/kz-selfplay/src/server/protocol.rs:10-29 (the fields this layer reads)
struct StartupSettings {
    size_t cpu_threads_per_device = 4;
    size_t gpu_threads_per_device = 2;
    size_t gpu_batch_size = 256;
    size_t search_batch_size = 16;
    // not in the reference: batches each executor thread keeps in flight (1 = batched_executor_loop as it is;
    // 2 = pipelined_executor_loop over the engine's two async slots)
    size_t pipeline_depth = 1;
    // not in the reference: decode_output on the GPU instead of on the executor thread
    bool device_decode = false;
    // not in the reference: helper threads per executor thread that share a batch's host work (encode_input, move lists)
    // with it (HipNetwork::set_prep_helpers; hip.rs: KZ_HIP_PREP_THREADS, default 0)
    size_t prep_helpers = 0;
};

struct DeviceSizing {
    size_t concurrent_games, eval_job_count, job_buffer_size;
    // a pipelined executor thread stands for `pipeline_depth` blocking ones
    static size_t lanes(const StartupSettings &s) { return s.gpu_threads_per_device * (s.pipeline_depth ? s.pipeline_depth : 1); }
    explicit DeviceSizing(const StartupSettings &s)
        : concurrent_games(ceil_div((lanes(s) + 1) * s.gpu_batch_size, s.search_batch_size)),  // :47
          eval_job_count(s.gpu_batch_size / s.search_batch_size),                              // :48
          job_buffer_size(ceil_div(lanes(s) * s.gpu_batch_size, s.search_batch_size)) {}       // :55
};

// Evals::new(real, potential, cached) (protocol.rs:52-72); `real`/s is the north-star metric (collector.rs:172-191)
struct EvalCounters {
    std::atomic<uint64_t> real{0}, potential{0};
    // measurement only: CPU time (CLOCK_THREAD_CPUTIME_ID, ns) each executor thread of the device has used so far, read
    // after every batch — what tells whether one executor thread keeps up with its GPU
    static constexpr size_t MAX_EXECUTORS = 16;
    std::atomic<uint64_t> executor_cpu_ns[MAX_EXECUTORS] = {};
    // ... of which inside the engine's wait calls (the HIP runtime polls: spinning, not work)
    std::atomic<uint64_t> executor_wait_ns[MAX_EXECUTORS] = {};
    // CPU time of the executor thread's prep helpers (StartupSettings::prep_helpers)
    std::atomic<uint64_t> helper_cpu_ns[MAX_EXECUTORS] = {};
    // executor thread 0's work by phase: own share of the preparation, merge, the engine's submit call, building the evaluations
    std::atomic<uint64_t> phase_ns[4] = {};
};

// (has `wait_cpu_ns`: HipNetwork; the tests' fake networks do not)
template <class T, class = void>
struct has_wait_cpu_ns : std::false_type {};
template <class T>
struct has_wait_cpu_ns<T, std::void_t<decltype(std::declval<const T &>().wait_cpu_ns)>> : std::true_type {};
template <class T, class = void>
struct has_prep_helpers : std::false_type {};
template <class T>
struct has_prep_helpers<T, std::void_t<decltype(std::declval<T &>().set_prep_helpers(size_t(0)))>> : std::true_type {};

// Net: the network an executor thread builds from a graph, `Net(mapper, graph, max_batch, device, dtype)` with
// evaluate_batch / submit_batch / wait_batch / set_device_decode / max_in_flight (HipNetwork<B, M>; the tests put a fake
// in its place to run several "devices" without a GPU).  GraphT: what the commander sends (`Arc<G>`).
template <class B, class M, class Net = HipNetwork<B, M>, class GraphT = std::shared_ptr<const HipModel>>
struct DeviceExecutors {
    using Graph = GraphT;
    int device = 0;
    JobClient<B, ZeroEvaluation> eval_client;
    std::vector<Sender<std::optional<Graph>>> graph_senders;  // one per executor thread (commander.rs:19-25)
    std::vector<std::thread> threads;

    void send_graph(std::optional<Graph> g) {
        for (auto &s : graph_senders) s.send(g);
    }
    // drop the channel ends and wait for the executors to drain and exit
    void join() {
        eval_client = JobClient<B, ZeroEvaluation>();
        graph_senders.clear();
        for (auto &t : threads) t.join();
        threads.clear();
    }
};

// server_alphazero.rs:89-121
template <class B, class M, class Net = HipNetwork<B, M>, class GraphT = std::shared_ptr<const HipModel>>
std::unique_ptr<DeviceExecutors<B, M, Net, GraphT>> spawn_device_executors(int device, const StartupSettings &startup,
                                                                           M mapper, int dtype, EvalCounters *counters) {
    const DeviceSizing sizing(startup);
    auto dev = std::make_unique<DeviceExecutors<B, M, Net, GraphT>>();
    dev->device = device;
    auto [client, server] = job_pair<B, ZeroEvaluation>(sizing.job_buffer_size);
    dev->eval_client = client;
    const size_t gpu_batch_size = startup.gpu_batch_size;
    for (size_t local_id = 0; local_id < startup.gpu_threads_per_device; local_id++) {
        auto [gtx, grx] = bounded<std::optional<GraphT>>(1);  // :90
        dev->graph_senders.push_back(gtx);
        dev->threads.emplace_back([=, srv = server, rx = std::move(grx)]() mutable {
            using Graph = GraphT;
            const bool device_decode = startup.device_decode;
            auto load = [=](Graph g) {
                Net net(mapper, std::move(g), gpu_batch_size, device, dtype);
                net.set_device_decode(device_decode);
                if constexpr (has_prep_helpers<Net>::value) net.set_prep_helpers(startup.prep_helpers);
                return net;
            };
            auto count = [=](size_t n, const Net &net) {
                if (counters) {
                    counters->real += n;  // ExpandEvals(real = x.len(), potential = gpu_batch_size) (:113-115)
                    counters->potential += gpu_batch_size;
                    if (local_id < EvalCounters::MAX_EXECUTORS) {
                        counters->executor_cpu_ns[local_id] = thread_cpu_ns();
                        if constexpr (has_wait_cpu_ns<Net>::value) counters->executor_wait_ns[local_id] = net.wait_cpu_ns;
                        if constexpr (has_prep_helpers<Net>::value) {
                            counters->helper_cpu_ns[local_id] = net.helper_cpu_ns();
                            if (local_id == 0) {
                                counters->phase_ns[0] = net.prep_own_cpu_ns;
                                counters->phase_ns[1] = net.merge_cpu_ns;
                                counters->phase_ns[2] = net.submit_cpu_ns;
                                counters->phase_ns[3] = net.assemble_cpu_ns;
                            }
                        }
                    }
                }
            };
            const RunCondition cond = RunCondition::job_count(sizing.eval_job_count);
            if (startup.pipeline_depth <= 1) {
                batched_executor_loop<Graph, Net, B, ZeroEvaluation>(
                    gpu_batch_size, cond, std::move(rx), std::move(srv), load, [=](Net &net, const B *x, size_t n) {
                        auto y = net.evaluate_batch(x, n);
                        count(n, net);
                        return y;
                    });
            } else {
                pipelined_executor_loop<Graph, Net, B, ZeroEvaluation>(
                    gpu_batch_size, std::min(startup.pipeline_depth, Net::max_in_flight()), cond, std::move(rx),
                    std::move(srv), load, [](Net &net, B *x, size_t n) { net.submit_batch(x, n); },
                    [=](Net &net) {
                        auto y = net.wait_batch();
                        count(y.size(), net);
                        return y;
                    });
            }
        });
    }
    return dev;
}

// The per-device loop of selfplay_start (rust/kz-selfplay/src/server/server.rs:323-331): one spawn_device_threads per
// device, each with its own job channel, executors and (the caller's) generators; nothing is shared between devices but
// the immutable graph and the counters.
template <class B, class M, class Net = HipNetwork<B, M>, class GraphT = std::shared_ptr<const HipModel>>
std::vector<std::unique_ptr<DeviceExecutors<B, M, Net, GraphT>>> spawn_all_devices(const std::vector<int> &devices,
                                                                                     const StartupSettings &startup, M mapper,
                                                                                     int dtype, EvalCounters *counters,
                                                                                     EvalCounters *per_device = nullptr) {
    // per_device: optional array of devices.size() counters, one per device, instead of the shared one (measurement only:
    // the reference's collector sums all devices, collector.rs:172-191)
    if (devices.empty()) throw std::invalid_argument("No devices found");  // server.rs:53, :314
    std::vector<std::unique_ptr<DeviceExecutors<B, M, Net, GraphT>>> all;
    for (size_t i = 0; i < devices.size(); i++)
        all.push_back(spawn_device_executors<B, M, Net, GraphT>(devices[i], startup, mapper, dtype,
                                                                per_device ? &per_device[i] : counters));
    return all;
}

}  // namespace kz::host
