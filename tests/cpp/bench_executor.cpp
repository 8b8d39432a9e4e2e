// bench_executor.cpp — NN evals/s THROUGH the drop-in seam: generator threads -> job channel -> batched_executor_loop ->
// HipNetwork::evaluate_batch (host encode_input, kz_engine_eval_packed over PCIe, host decode_output) -> replies.
// Sized like the self-play server (rust/kz-selfplay/src/server/server_alphazero.rs:47-55) and counted like its
// collector (`real` evals/s, collector.rs:172-191).  Not a test: a measurement of the host side next to bench.py.
//
//   bench_executor <model.kzm|onnx> <seconds> <gpu_threads> <generator_threads> [gpu_batch] [search_batch] [dtype] [depth] [device_decode] [devices]
//   depth = batches each executor thread keeps in flight (1 = batched_executor_loop, 2 = pipelined_executor_loop)
//   devices = comma-separated device ordinals (default "0"): ONE process, a thread set (executors + generators) per device —
//             the topology of selfplay_start (rust/kz-selfplay/src/server/server.rs:323-331); gpu_threads and
//             generator_threads are per device
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "../../kzero_amd/csrc/host/device_threads.hpp"

using namespace kz::host;

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s model seconds gpu_threads generator_threads [gpu_batch] [search_batch] [f16|f32|f32split16]\n", argv[0]);
        return 2;
    }
    const double seconds = atof(argv[2]);
    StartupSettings st;
    st.gpu_threads_per_device = (size_t)atoi(argv[3]);
    const int generators = atoi(argv[4]);
    if (argc > 5) st.gpu_batch_size = (size_t)atoi(argv[5]);
    if (argc > 6) st.search_batch_size = (size_t)atoi(argv[6]);
    const std::string dtype_name = argc > 7 ? argv[7] : "f16";
    const int dtype = dtype_name == "f32" ? KZ_DTYPE_F32 : dtype_name == "f32split16" ? KZ_DTYPE_F32_SPLIT16 : KZ_DTYPE_F16;
    if (argc > 8) st.pipeline_depth = (size_t)atoi(argv[8]);
    if (argc > 9) st.device_decode = atoi(argv[9]) != 0;
    std::vector<int> devices;
    {
        const std::string list = argc > 10 ? argv[10] : "0";
        size_t pos = 0;
        while (pos <= list.size()) {
            const size_t comma = list.find(',', pos);
            const std::string item = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
            if (!item.empty()) devices.push_back(atoi(item.c_str()));
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
        if (devices.empty()) devices.push_back(0);
    }

    auto model = std::make_shared<const HipModel>(argv[1]);
    const kz_model_info info = model->info;
    PackedMapper mapper{(size_t)info.input_bool_channels, (size_t)info.board_h, (size_t)info.board_w,
                        (size_t)info.input_scalar_channels, (size_t)info.policy_len};
    // a pool of synthetic positions with ~30 legal moves each
    std::mt19937 rng(1);
    std::vector<PackedBoard> pool(1024);
    for (auto &b : pool) {
        b.bits.resize((size_t)info.bits_bytes);
        for (auto &byte : b.bits) byte = (uint8_t)(rng() & rng() & rng() & 0xff);
        b.scalars.assign((size_t)info.input_scalar_channels, 0.0f);
        if (!b.scalars.empty()) b.scalars[0] = 1.0f;
        std::vector<int32_t> moves(20 + rng() % 21);
        for (auto &m : moves) m = (int32_t)(rng() % info.policy_len);
        b.moves = moves;
    }

    // one process, one thread set per device (server.rs:323-331), one counter per device
    std::unique_ptr<EvalCounters[]> per_device(new EvalCounters[devices.size()]);
    auto devs = spawn_all_devices<PackedBoard, PackedMapper>(devices, st, mapper, dtype, nullptr, per_device.get());
    for (auto &dev : devs) dev->send_graph(model);
    const DeviceSizing sizing(st);
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> replies{0};
    // each generator thread stands for concurrent_games / generators games, each with one request in flight
    const size_t games_per_thread = ceil_div(sizing.concurrent_games, (size_t)generators);
    std::vector<std::thread> gens;
    for (auto &dev : devs)
    for (int t = 0; t < generators; t++)
        gens.emplace_back([&, t, client = dev->eval_client, seed = 100 + t + 1000 * dev->device] {
            std::mt19937 r(seed);
            std::vector<Receiver<std::vector<ZeroEvaluation>>> inflight;
            auto request = [&] {
                std::vector<PackedBoard> x;
                for (size_t k = 0; k < st.search_batch_size; k++) x.push_back(pool[r() % pool.size()]);
                return client.map(std::move(x));
            };
            for (size_t gme = 0; gme < games_per_thread; gme++) inflight.push_back(request());
            size_t next = 0;
            while (!stop) {
                auto y = inflight[next].recv();
                if (!y) break;
                replies += y->size();
                inflight[next] = request();
                next = (next + 1) % inflight.size();
            }
        });
    std::this_thread::sleep_for(std::chrono::milliseconds(500));  // warm-up
    std::vector<uint64_t> r0(devices.size()), p0(devices.size());
    for (size_t i = 0; i < devices.size(); i++) r0[i] = per_device[i].real, p0[i] = per_device[i].potential;
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    uint64_t real = 0, potential = 0;
    std::string per, devs_json;
    std::vector<uint64_t> r1(devices.size());
    for (size_t i = 0; i < devices.size(); i++) {
        r1[i] = per_device[i].real;
        real += r1[i] - r0[i];
        potential += per_device[i].potential - p0[i];
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop = true;
    for (size_t i = 0; i < devices.size(); i++) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "%s%.1f", i ? ", " : "", (r1[i] - r0[i]) / dt);
        per += buf;
        std::snprintf(buf, sizeof buf, "%s%d", i ? ", " : "", devices[i]);
        devs_json += buf;
    }
    std::printf("{\"evals_per_s\": %.1f, \"fill\": %.3f, \"gpu_threads\": %zu, \"generator_threads\": %d, "
                "\"concurrent_games\": %zu, \"gpu_batch\": %zu, \"search_batch\": %zu, \"pipeline_depth\": %zu, \"device_decode\": %d, "
                "\"seconds\": %.2f, \"devices\": [%s], \"per_device_evals_per_s\": [%s]}\n",
                real / dt, (double)real / (double)(potential ? potential : 1), st.gpu_threads_per_device, generators,
                sizing.concurrent_games, st.gpu_batch_size, st.search_batch_size, st.pipeline_depth, (int)st.device_decode, dt,
                devs_json.c_str(), per.c_str());
    std::fflush(stdout);
    // generators block in recv(); the process exit tears everything down (the reference server has no clean stop
    // either: commander.rs:63-64)
    std::_Exit(0);
}
