// test_two_devices.cpp — the topology the reference and the Rust drop-in use on a multi-GPU node: ONE process, one set of
// executor threads per device (rust/kz-selfplay/src/server/server.rs:323-331 -> server_alphazero.rs:89-121), every
// thread building its own network on its own device from the one shared graph.  Runs spawn_all_devices on devices
// {0} and then on {0, 1} and requires, bitwise, the same evaluations from both devices as from device 0 alone — what
// would catch a weight cache keyed without the device, pinned staging or a stream bound to the wrong device, or a
// hipSetDevice missing on an executor thread.  Exit code 77 = fewer than two GPUs visible (the pytest wrapper skips).
//
//   test_two_devices <model.kzm> [dtype f16|f32|f32split16] [gpu_batch]
#include <cstdio>
#include <chrono>
#include <cstring>
#include <optional>
#include <random>
#include <string>

#include "../../kzero_amd/csrc/host/device_threads.hpp"

using namespace kz::host;

static int g_failed = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_failed++;                                                          \
        }                                                                        \
    } while (0)

struct Flat {
    std::vector<float> values;  // per evaluation: value, wdl x3, moves_left, then the policy over its moves
};

// `requests` requests of `per` boards each through the device's job channel, in order; the replies flattened
static Flat run_requests(DeviceExecutors<PackedBoard, PackedMapper> &dev, const std::vector<PackedBoard> &pool, size_t requests,
                         size_t per) {
    std::vector<Receiver<std::vector<ZeroEvaluation>>> pending;
    for (size_t r = 0; r < requests; r++) {
        std::vector<PackedBoard> x;
        for (size_t k = 0; k < per; k++) x.push_back(pool[(r * per + k) % pool.size()]);
        pending.push_back(dev.eval_client.map(std::move(x)));
    }
    Flat out;
    // Two executor threads share the device's job channel and each evaluates when it holds JobCount(gpu_batch / search_batch)
    // jobs (server_alphazero.rs:48): a FINITE set of requests can end with both holding a partial batch — in the server the
    // generators never stop, here filler requests keep arriving until every counted reply is in (the first version of this
    // test waited without them and deadlocked now and then: found by the one-GPU rehearsal of round 5)
    std::vector<Receiver<std::vector<ZeroEvaluation>>> fillers;
    for (auto &p : pending) {
        std::optional<std::vector<ZeroEvaluation>> y;
        for (;;) {
            TryRecvError err = TryRecvError::Empty;
            y = p.try_recv(err);
            if (y || err == TryRecvError::Disconnected) break;
            std::vector<PackedBoard> x(pool.begin(), pool.begin() + (long)per);
            fillers.push_back(dev.eval_client.map(std::move(x)));
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        if (!y) {
            CHECK(!"reply channel closed");
            break;
        }
        for (const ZeroEvaluation &e : *y) {
            out.values.push_back(e.values.value);
            out.values.push_back(e.values.wdl.win);
            out.values.push_back(e.values.wdl.draw);
            out.values.push_back(e.values.wdl.loss);
            out.values.push_back(e.values.moves_left);
            out.values.insert(out.values.end(), e.policy.begin(), e.policy.end());
        }
    }
    return out;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s model.kzm [dtype] [gpu_batch]\n", argv[0]);
        return 2;
    }
    int ndev = 0;
    // KZ_TWO_DEVICES_REHEARSE=1 on a one-GPU box: both thread sets on device 0 — everything of the topology but the second
    // piece of hardware (two job channels, 2 x 2 executor threads with their own engines, one shared graph, concurrent drive)
    const char *rehearse_env = getenv("KZ_TWO_DEVICES_REHEARSE");
    const bool rehearse = rehearse_env && rehearse_env[0] == '1';
    if (kz_device_count(&ndev) != 0 || ndev < (rehearse ? 1 : 2)) {
        std::printf("skip: %d GPU(s) visible, the two-device topology needs 2\n", ndev);
        return 77;
    }
    const int second = ndev >= 2 ? 1 : 0;
    const std::string dtype_name = argc > 2 ? argv[2] : "f16";
    const int dtype = dtype_name == "f32" ? KZ_DTYPE_F32 : dtype_name == "f32split16" ? KZ_DTYPE_F32_SPLIT16 : KZ_DTYPE_F16;
    StartupSettings st;
    st.gpu_threads_per_device = 2;
    st.gpu_batch_size = argc > 3 ? (size_t)atoi(argv[3]) : 64;
    st.search_batch_size = 8;
    st.pipeline_depth = 2;

    auto model = std::make_shared<const HipModel>(argv[1]);
    const kz_model_info info = model->info;
    PackedMapper mapper{(size_t)info.input_bool_channels, (size_t)info.board_h, (size_t)info.board_w,
                        (size_t)info.input_scalar_channels, (size_t)info.policy_len};
    std::mt19937 rng(7);
    std::vector<PackedBoard> pool(256);
    for (auto &b : pool) {
        b.bits.resize((size_t)info.bits_bytes);
        for (auto &byte : b.bits) byte = (uint8_t)(rng() & rng() & 0xff);
        b.scalars.assign((size_t)info.input_scalar_channels, 0.0f);
        for (auto &s : b.scalars) s = (float)(rng() % 3);
        std::vector<int32_t> moves(1 + rng() % 30);
        for (auto &m : moves) m = (int32_t)(rng() % info.policy_len);
        b.moves = moves;
    }
    const size_t requests = 64, per = st.search_batch_size;

    // device 0 alone
    Flat alone;
    {
        EvalCounters c;
        auto one = spawn_all_devices<PackedBoard, PackedMapper>({0}, st, mapper, dtype, &c);
        one[0]->send_graph(model);
        alone = run_requests(*one[0], pool, requests, per);
        CHECK(c.real >= requests * per);  // (+ the filler requests)
        for (auto &d : one) d->join();
    }
    CHECK(alone.values.size() > requests * per * 5);

    // devices 0 and 1, one process: both driven at the same time from two generator threads
    {
        std::unique_ptr<EvalCounters[]> per_device(new EvalCounters[2]);
        auto both = spawn_all_devices<PackedBoard, PackedMapper>({0, second}, st, mapper, dtype, nullptr, per_device.get());
        CHECK(both.size() == 2 && both[0]->device == 0 && both[1]->device == second);
        for (auto &d : both) d->send_graph(model);
        Flat got[2];
        std::thread t0([&] { got[0] = run_requests(*both[0], pool, requests, per); });
        std::thread t1([&] { got[1] = run_requests(*both[1], pool, requests, per); });
        t0.join();
        t1.join();
        for (int d = 0; d < 2; d++) {
            CHECK(per_device[d].real >= requests * per);
            CHECK(got[d].values.size() == alone.values.size());
            CHECK(got[d].values.size() == alone.values.size() &&
                  std::memcmp(got[d].values.data(), alone.values.data(), alone.values.size() * sizeof(float)) == 0);
        }
        // a new graph reaches every executor of every device (commander.rs:36-45) and the answers stay the same
        for (auto &d : both) d->send_graph(model);
        std::thread t2([&] { got[0] = run_requests(*both[0], pool, requests, per); });
        std::thread t3([&] { got[1] = run_requests(*both[1], pool, requests, per); });
        t2.join();
        t3.join();
        for (int d = 0; d < 2; d++)
            CHECK(got[d].values.size() == alone.values.size() &&
                  std::memcmp(got[d].values.data(), alone.values.data(), alone.values.size() * sizeof(float)) == 0);
        for (auto &d : both) d->join();
    }
    if (g_failed) {
        std::fprintf(stderr, "%d check(s) failed\n", g_failed);
        return 1;
    }
    std::printf("two-device tests ok (%s%s)\n", dtype_name.c_str(), second ? "" : "; REHEARSAL: both thread sets on device 0");
    return 0;
}
