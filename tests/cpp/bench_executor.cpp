// bench_executor.cpp — NN evals/s THROUGH the drop-in seam: generator threads -> job channel -> batched_executor_loop ->
// HipNetwork::evaluate_batch (host encode_input, kz_engine_eval_packed over PCIe, host decode_output) -> replies.
// Sized like the self-play server (rust/kz-selfplay/src/server/server_alphazero.rs:47-55) and counted like its
// collector (`real` evals/s, collector.rs:172-191).  Not a test: a measurement of the host side next to bench.py.
//
//   bench_executor <model.kzm|onnx> <seconds> <gpu_threads> <generator_threads> [gpu_batch] [search_batch] [dtype] [depth]
//                  [device_decode] [devices] [work] [prep_helpers]
//   prep_helpers = helper threads per executor thread sharing a batch's encode_input + move lists with it (hip.rs default 1)
//   depth = batches each executor thread keeps in flight (1 = batched_executor_loop, 2 = pipelined_executor_loop)
//   devices = comma-separated device ordinals (default "0"): ONE process, a thread set (executors + generators) per device —
//             the topology of selfplay_start (rust/kz-selfplay/src/server/server.rs:323-331); gpu_threads and
//             generator_threads are per device
//   work = what the executor thread does besides submit / wait (chess models only, except "packed"):
//     packed  pre-packed boards with pre-made move lists through a pass-through mapper: the channel and PCIe alone
//     real    what kzero_amd/rust/hip.rs does: ChessStdMapper::encode_input over bitboards at submit; at decode, per board,
//             a fresh move generation (`available_moves()`), a SipHash map lookup per move (chess.rs:202-210) and the
//             softmax (device_decode = 0), or move generation + lookups at submit and the softmax on the GPU (= 1)
//     pre     the generators compute the policy indices when they make the request (move generation + lookups on THEIR
//             threads) and carry them in the job; the executor encodes, submits, waits (device_decode = 1)
//
// The record carries, besides evals/s: CPU utilisation of every executor thread (CLOCK_THREAD_CPUTIME_ID over the timed
// span), of the generator threads, the whole process's CPU seconds per million evaluations, and a projection of the
// host budget to eight GPUs.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>

#include <sys/resource.h>

#include "../../kzero_amd/csrc/host/device_threads.hpp"
#include "bench_chess.hpp"

using namespace kz::host;

namespace {

struct Args {
    std::string model, dtype_name = "f16", work = "packed";
    double seconds = 2;
    int generators = 6;
    std::vector<int> devices;
    StartupSettings st;
};

double process_cpu_s() {
    rusage ru{};
    getrusage(RUSAGE_SELF, &ru);
    return ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec);
}

// CPUs per NUMA node of this machine (sysfs), for the projection
std::vector<int> numa_node_cpus() {
    std::vector<int> out;
    for (int node = 0; node < 64; node++) {
        std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
        if (!f) break;
        std::string txt;
        std::getline(f, txt);
        int n = 0;
        size_t pos = 0;
        while (pos < txt.size()) {
            size_t comma = txt.find(',', pos);
            std::string part = txt.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
            size_t dash = part.find('-');
            if (!part.empty()) {
                const int lo = atoi(part.c_str()), hi = dash == std::string::npos ? lo : atoi(part.c_str() + dash + 1);
                n += hi - lo + 1;
            }
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
        out.push_back(n);
    }
    return out;
}

template <class B, class M, class MakeRequest>
int run(const Args &a, std::shared_ptr<const HipModel> model, M mapper, MakeRequest make_request) {
    const StartupSettings &st = a.st;
    const int dtype = a.dtype_name == "f32" ? KZ_DTYPE_F32 : a.dtype_name == "f32split16" ? KZ_DTYPE_F32_SPLIT16 : KZ_DTYPE_F16;
    const kz_model_info info = model->info;
    std::unique_ptr<EvalCounters[]> per_device(new EvalCounters[a.devices.size()]);
    auto devs = spawn_all_devices<B, M>(a.devices, st, mapper, dtype, nullptr, per_device.get());
    for (auto &dev : devs) dev->send_graph(model);
    const DeviceSizing sizing(st);
    std::atomic<bool> stop{false};
    const size_t n_gen = (size_t)a.generators * devs.size();
    std::unique_ptr<std::atomic<uint64_t>[]> gen_cpu(new std::atomic<uint64_t>[n_gen]);
    for (size_t i = 0; i < n_gen; i++) gen_cpu[i] = 0;
    // each generator thread stands for concurrent_games / generators games, each with one request in flight
    const size_t games_per_thread = ceil_div(sizing.concurrent_games, (size_t)a.generators);
    std::vector<std::thread> gens;
    size_t gi = 0;
    for (auto &dev : devs)
        for (int t = 0; t < a.generators; t++, gi++)
            gens.emplace_back([&, gi, client = dev->eval_client, seed = 100 + t + 1000 * dev->device] {
                std::mt19937 r(seed);
                // one reply channel for all of this thread's games: replies are taken in the order they ARRIVE, like the
                // reference's independent game futures.  (Waiting for them in submission order deadlocks now and then
                // with several executor threads: every generator waits for a request that sits in some executor's
                // partial batch while the answered ones are not replaced — seen once as a 0 evals/s record.)
                auto [reply_tx, reply_rx] = bounded<std::vector<ZeroEvaluation>>(games_per_thread);
                for (size_t gme = 0; gme < games_per_thread; gme++) client.map_into(make_request(r, st.search_batch_size), reply_tx);
                while (!stop) {
                    auto y = reply_rx.recv();
                    if (!y) break;
                    client.map_into(make_request(r, st.search_batch_size), reply_tx);
                    gen_cpu[gi] = thread_cpu_ns();
                }
            });
    // warm-up: until every device has answered its first batches (engine creation — weight packing and upload — runs on
    // the executor thread and must not be counted as its work), then half a second more
    for (int spin = 0; spin < 600; spin++) {
        bool all = true;
        for (size_t i = 0; i < a.devices.size(); i++) all &= per_device[i].real.load() > 4 * st.gpu_batch_size;
        if (all) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    const size_t nd = a.devices.size(), ne = std::min(st.gpu_threads_per_device, EvalCounters::MAX_EXECUTORS);
    std::vector<uint64_t> r0(nd), p0(nd), g0(n_gen);
    std::vector<std::vector<uint64_t>> e0(nd, std::vector<uint64_t>(ne)), w0(nd, std::vector<uint64_t>(ne)), h0(nd, std::vector<uint64_t>(ne));
    for (size_t i = 0; i < nd; i++) {
        r0[i] = per_device[i].real, p0[i] = per_device[i].potential;
        for (size_t k = 0; k < ne; k++)
            e0[i][k] = per_device[i].executor_cpu_ns[k], w0[i][k] = per_device[i].executor_wait_ns[k], h0[i][k] = per_device[i].helper_cpu_ns[k];
    }
    for (size_t i = 0; i < n_gen; i++) g0[i] = gen_cpu[i];
    uint64_t ph0[4];
    for (int k = 0; k < 4; k++) ph0[k] = per_device[0].phase_ns[k];
    const double cpu0 = process_cpu_s();
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(a.seconds));
    uint64_t real = 0, potential = 0;
    std::vector<uint64_t> r1(nd);
    for (size_t i = 0; i < nd; i++) {
        r1[i] = per_device[i].real;
        real += r1[i] - r0[i];
        potential += per_device[i].potential - p0[i];
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double cpu_s = process_cpu_s() - cpu0;
    std::string exec_util, exec_work, gen_util, per, devs_json, helper_util;
    double exec_cpu_total = 0, exec_max = 0, gen_cpu_total = 0, exec_work_total = 0, exec_work_max = 0, helper_total = 0;
    for (size_t i = 0; i < nd; i++)
        for (size_t k = 0; k < ne; k++) {
            const double u = (per_device[i].executor_cpu_ns[k] - e0[i][k]) * 1e-9 / dt;
            // work = CPU time outside kz_engine_wait* (where the HIP runtime polls the event)
            const double wk = u - (per_device[i].executor_wait_ns[k] - w0[i][k]) * 1e-9 / dt;
            exec_cpu_total += u * dt;
            exec_work_total += wk * dt;
            exec_max = std::max(exec_max, u);
            exec_work_max = std::max(exec_work_max, wk);
            char buf[32];
            std::snprintf(buf, sizeof buf, "%s%.3f", exec_util.empty() ? "" : ", ", u);
            exec_util += buf;
            std::snprintf(buf, sizeof buf, "%s%.3f", exec_work.empty() ? "" : ", ", wk);
            exec_work += buf;
            const double hu = (per_device[i].helper_cpu_ns[k] - h0[i][k]) * 1e-9 / dt;  // all helpers of this executor thread
            helper_total += hu * dt;
            std::snprintf(buf, sizeof buf, "%s%.3f", helper_util.empty() ? "" : ", ", hu);
            helper_util += buf;
        }
    for (size_t i = 0; i < n_gen; i++) {
        const double u = (gen_cpu[i] - g0[i]) * 1e-9 / dt;
        gen_cpu_total += u * dt;
        char buf[32];
        std::snprintf(buf, sizeof buf, "%s%.3f", i ? ", " : "", u);
        gen_util += buf;
    }
    stop = true;
    for (size_t i = 0; i < nd; i++) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "%s%.1f", i ? ", " : "", (r1[i] - r0[i]) / dt);
        per += buf;
        std::snprintf(buf, sizeof buf, "%s%d", i ? ", " : "", a.devices[i]);
        devs_json += buf;
    }
    const double evals_s = real / dt, mevals = (real ? real : 1) / 1e6;  // (a stalled run prints 0 evals/s, not inf)
    {  // device 0, executor thread 0: microseconds of work per batch by phase (the rest of its work: job collection, reply fan-out)
        const double batches = (double)(r1[0] - r0[0]) / (double)st.gpu_batch_size / (double)ne;
        std::fprintf(stderr, "executor 0 per batch: prepare(own) %.1f us, merge %.1f us, submit %.1f us, assemble %.1f us\n",
                     (per_device[0].phase_ns[0] - ph0[0]) * 1e-3 / batches, (per_device[0].phase_ns[1] - ph0[1]) * 1e-3 / batches,
                     (per_device[0].phase_ns[2] - ph0[2]) * 1e-3 / batches, (per_device[0].phase_ns[3] - ph0[3]) * 1e-3 / batches);
    }
    // ---- projection to one node of eight GPUs at this per-device rate: host cores (the whole process's CPU seconds per
    // second: executors, generators' request handling, the HIP runtime's own threads) and PCIe bytes ----
    const double per_dev_rate = evals_s / nd, cores_per_device = cpu_s / dt / nd;
    const double in_bytes = info.bits_bytes + 4.0 * info.input_scalar_channels;
    const double out_bytes = st.device_decode ? 20.0 + 8.0 + 30.0 * 8.0 : 4.0 * (5 + info.policy_len);  // (~30 moves: index in, probability out)
    std::string numa;
    for (int n : numa_node_cpus()) numa += (numa.empty() ? "" : ", ") + std::to_string(n);
    std::printf("{\"evals_per_s\": %.1f, \"fill\": %.3f, \"gpu_threads\": %zu, \"generator_threads\": %d, "
                "\"concurrent_games\": %zu, \"gpu_batch\": %zu, \"search_batch\": %zu, \"pipeline_depth\": %zu, \"device_decode\": %d, "
                "\"seconds\": %.2f, \"devices\": [%s], \"per_device_evals_per_s\": [%s], \"work\": \"%s\", \"dtype\": \"%s\", "
                "\"executor_cpu_util\": [%s], \"executor_cpu_util_max\": %.3f, \"executor_work_util\": [%s], "
                "\"executor_work_util_max\": %.3f, \"prep_helpers\": %zu, \"helper_cpu_util\": [%s], \"helper_cpu_s_per_Meval\": %.3f, "
                "\"generator_cpu_util\": [%s], "
                "\"host_cpu_s_per_Meval\": %.3f, \"executor_cpu_s_per_Meval\": %.3f, \"executor_work_s_per_Meval\": %.3f, "
                "\"generator_cpu_s_per_Meval\": %.3f, "
                "\"projection_8gpu\": {\"evals_per_s\": %.0f, \"cores_needed\": %.1f, \"cores_per_numa_node\": [%s], "
                "\"pcie_GBps\": %.3f, \"pcie_GBps_per_gpu\": %.3f}}\n",
                evals_s, (double)real / (double)(potential ? potential : 1), st.gpu_threads_per_device, a.generators,
                sizing.concurrent_games, st.gpu_batch_size, st.search_batch_size, st.pipeline_depth, (int)st.device_decode, dt,
                devs_json.c_str(), per.c_str(), a.work.c_str(), a.dtype_name.c_str(), exec_util.c_str(), exec_max, exec_work.c_str(),
                exec_work_max, st.prep_helpers, helper_util.c_str(), helper_total / mevals, gen_util.c_str(), cpu_s / mevals, exec_cpu_total / mevals, exec_work_total / mevals, gen_cpu_total / mevals,
                8 * per_dev_rate, 8 * cores_per_device, numa.c_str(),
                8 * per_dev_rate * (in_bytes + out_bytes) / 1e9, per_dev_rate * (in_bytes + out_bytes) / 1e9);
    std::fflush(stdout);
    if (const char *clean = getenv("KZ_BENCH_CLEAN_EXIT"); clean && clean[0] == '1') {
        // under a profiler (tools/profile_r5.sh: rocprofv3 writes its kernel trace when the process ends NORMALLY): every
        // generator leaves its loop at its next reply, the job channels disconnect, the executors evaluate what is left and
        // exit, the engines are destroyed
        for (auto &g : gens) g.join();
        for (auto &dev : devs) dev->join();
        return 0;
    }
    // generators block in recv(); the process exit tears everything down (the reference server has no clean stop
    // either: commander.rs:63-64)
    std::_Exit(0);
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s model seconds gpu_threads generator_threads [gpu_batch] [search_batch] [f16|f32|f32split16] "
                             "[depth] [device_decode] [devices] [packed|real|pre] [prep_helpers]\n", argv[0]);
        return 2;
    }
    Args a;
    a.model = argv[1];
    a.seconds = atof(argv[2]);
    a.st.gpu_threads_per_device = (size_t)atoi(argv[3]);
    a.generators = atoi(argv[4]);
    if (argc > 5) a.st.gpu_batch_size = (size_t)atoi(argv[5]);
    if (argc > 6) a.st.search_batch_size = (size_t)atoi(argv[6]);
    if (argc > 7) a.dtype_name = argv[7];
    if (argc > 8) a.st.pipeline_depth = (size_t)atoi(argv[8]);
    if (argc > 9) a.st.device_decode = atoi(argv[9]) != 0;
    {
        const std::string list = argc > 10 ? argv[10] : "0";
        size_t pos = 0;
        while (pos <= list.size()) {
            const size_t comma = list.find(',', pos);
            const std::string item = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
            if (!item.empty()) a.devices.push_back(atoi(item.c_str()));
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
        if (a.devices.empty()) a.devices.push_back(0);
    }
    if (argc > 11) a.work = argv[11];
    if (argc > 12) a.st.prep_helpers = (size_t)atoi(argv[12]);
    // (the executor and generator threads of a device belong next to it: a launcher that pinned this process to one NUMA
    // node for its own GPU passes KZ_BENCH_FULL_AFFINITY=1 when the child drives several devices)
    if (const char *full = getenv("KZ_BENCH_FULL_AFFINITY"); full && full[0] == '1') {
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &set);
        (void)sched_setaffinity(0, sizeof set, &set);
    }

    auto model = std::make_shared<const HipModel>(a.model);
    const kz_model_info info = model->info;
    if (a.work == "packed") {
        PackedMapper mapper{(size_t)info.input_bool_channels, (size_t)info.board_h, (size_t)info.board_w,
                            (size_t)info.input_scalar_channels, (size_t)info.policy_len};
        // a pool of synthetic positions with ~30 legal moves each
        std::mt19937 rng(1);
        auto pool = std::make_shared<std::vector<PackedBoard>>(1024);
        for (auto &b : *pool) {
            b.bits.resize((size_t)info.bits_bytes);
            for (auto &byte : b.bits) byte = (uint8_t)(rng() & rng() & rng() & 0xff);
            b.scalars.assign((size_t)info.input_scalar_channels, 0.0f);
            if (!b.scalars.empty()) b.scalars[0] = 1.0f;
            std::vector<int32_t> moves(20 + rng() % 21);
            for (auto &m : moves) m = (int32_t)(rng() % info.policy_len);
            b.moves = moves;
        }
        return run<PackedBoard, PackedMapper>(a, model, mapper, [pool](std::mt19937 &r, size_t n) {
            std::vector<PackedBoard> x;
            for (size_t k = 0; k < n; k++) x.push_back((*pool)[r() % pool->size()]);
            return x;
        });
    }
    if (a.work != "real" && a.work != "pre") {
        std::fprintf(stderr, "unknown work '%s'\n", a.work.c_str());
        return 2;
    }
    if (info.policy_len != 1880 || info.input_channels != 21) {
        std::fprintf(stderr, "work=%s needs a chess model (ChessStdMapper: 21 planes, 1880 moves)\n", a.work.c_str());
        return 2;
    }
    using kz::bench::BenchChessBoard;
    using kz::bench::HashedChessMapper;
    std::mt19937 rng(1);
    auto pool = std::make_shared<std::vector<kz::host::ChessPosition>>(4096);
    double moves_total = 0;
    for (auto &p : *pool) {
        p = kz::bench::random_position(rng);
        std::vector<kz::host::ChessMove> mv;
        kz::bench::pseudo_legal_moves(p, mv);
        moves_total += mv.size();
    }
    std::fprintf(stderr, "positions: %.1f pseudo-legal moves on average\n", moves_total / pool->size());
    const bool pre = a.work == "pre";
    if (pre) a.st.device_decode = true;
    HashedChessMapper mapper;
    return run<BenchChessBoard, HashedChessMapper>(a, model, mapper, [pool, pre, mapper](std::mt19937 &r, size_t n) {
        std::vector<BenchChessBoard> x(n);
        for (auto &b : x) {
            b.pos = (*pool)[r() % pool->size()];
            if (pre) {  // the generator's share: move generation + a hash lookup per move, carried in the job
                auto moves = b.available_moves();
                b.indices.reserve(moves->size());
                for (const auto &mv : *moves) b.indices.push_back((int32_t)mapper.move_to_index(b, mv));
                b.has_indices = true;
            }
        }
        return x;
    });
}
