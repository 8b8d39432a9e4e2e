// test_hip_network.cpp — GPU test of the host-side mirror on the real engine: HipNetwork behind the Network contract,
// driven by batched_executor_loop from several generator threads, with a network hot-swap (the reference's own GPU
// check of this layer is kz-misc/src/bin/test_concurrent.rs:32-145: many threads, one executor each, results must not
// depend on the thread).  Built against libkzhip.so and run by tests/test_host_cpp.py (-m gpu).
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <thread>

#include "../../kzero_amd/csrc/host/executor.hpp"
#include "../../kzero_amd/csrc/host/hip_network.hpp"
#include "../../kzero_amd/csrc/host/symmetry.hpp"

using namespace kz::host;

static int g_failed = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_failed++;                                                          \
        }                                                                        \
    } while (0)

static std::vector<uint8_t> read_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}

// the committed packed fixture of a golden net: u8 batch | bits [b][bits_bytes] | scalars f32 [b][n_scalar]
static std::vector<PackedBoard> load_boards(const std::string &golden, const std::string &name, const kz_model_info &info,
                                            std::mt19937 &rng) {
    auto raw = read_file(golden + "/" + name + ".planes.packed.bin");
    const size_t b = raw[0], nb = info.bits_bytes, ns = info.input_scalar_channels;
    std::vector<PackedBoard> boards(b);
    for (size_t i = 0; i < b; i++) {
        boards[i].bits.assign(raw.begin() + 1 + i * nb, raw.begin() + 1 + (i + 1) * nb);
        boards[i].scalars.resize(ns);
        std::memcpy(boards[i].scalars.data(), raw.data() + 1 + b * nb + i * ns * 4, ns * 4);
        std::vector<int32_t> moves;  // a random set of "available moves"
        const int count = 1 + (int)(rng() % 20);
        for (int k = 0; k < count; k++) moves.push_back((int32_t)(rng() % info.policy_len));
        boards[i].moves = moves;
    }
    return boards;
}

// golden outputs in the reference's check format: u8 batch | input | scalars [b,5] | policy [b,P]
static void load_expected(const std::string &golden, const std::string &name, const kz_model_info &info, size_t b,
                          std::vector<float> &scalars, std::vector<float> &policy) {
    auto raw = read_file(golden + "/" + name + ".planes.io.bin");
    const size_t n_in = b * info.input_channels * info.board_h * info.board_w;
    const float *f = reinterpret_cast<const float *>(raw.data() + 1);
    std::vector<float> all((raw.size() - 1) / 4);
    std::memcpy(all.data(), raw.data() + 1, all.size() * 4);
    (void)f;
    scalars.assign(all.begin() + n_in, all.begin() + n_in + b * 5);
    policy.assign(all.begin() + n_in + b * 5, all.end());
}

static bool close_eval(const ZeroEvaluation &a, const ZeroEvaluation &b, float tol) {
    if (a.policy.size() != b.policy.size()) return false;
    if (std::fabs(a.values.value - b.values.value) > tol || std::fabs(a.values.wdl.win - b.values.wdl.win) > tol) return false;
    for (size_t i = 0; i < a.policy.size(); i++)
        if (std::fabs(a.policy[i] - b.policy[i]) > tol) return false;
    return true;
}

int main(int argc, char **argv) {
    const std::string golden = argc > 1 ? argv[1] : "tests/golden";
    std::mt19937 rng(5);
    using Net = HipNetwork<PackedBoard, PackedMapper>;
    auto model_a = std::make_shared<const HipModel>(golden + "/ataxx7_2x16.kzm");
    auto model_b = std::make_shared<const HipModel>(golden + "/ataxx7_4x64.kzm");
    const kz_model_info info = model_a->info;
    PackedMapper mapper{(size_t)info.input_bool_channels, (size_t)info.board_h, (size_t)info.board_w,
                        (size_t)info.input_scalar_channels, (size_t)info.policy_len};
    auto boards = load_boards(golden, "ataxx7_2x16", info, rng);
    auto boards_b = load_boards(golden, "ataxx7_4x64", model_b->info, rng);

    // 1. HipNetwork == golden logits through decode_output (f32 path, 1e-4)
    std::vector<float> s_gold, p_gold;
    load_expected(golden, "ataxx7_2x16", info, boards.size(), s_gold, p_gold);
    auto expect = decode_output(mapper, boards.data(), boards.size(), s_gold.data(), p_gold.data());
    Net direct(mapper, model_a, 8, 0, KZ_DTYPE_F32);
    CHECK(direct.max_batch_size() == 8);
    auto got = direct.evaluate_batch(boards.data(), boards.size());
    CHECK(got.size() == boards.size());
    for (size_t i = 0; i < got.size(); i++) CHECK(close_eval(got[i], expect[i], 1e-4f));
    auto single = direct.evaluate(boards[2]);
    CHECK(close_eval(single, expect[2], 1e-4f));
    CHECK(direct.evaluate_batch(boards.data(), 0).empty());
    {
        std::vector<PackedBoard> many(9, boards[0]);
        bool threw = false;
        try { direct.evaluate_batch(many.data(), many.size()); } catch (const std::invalid_argument &) { threw = true; }
        CHECK(threw);  // assert!(batch_size <= max_batch_size), cudnn.rs:58
    }
    {  // shape check like check_graph_shapes (common.rs:165-198)
        PackedMapper wrong = mapper;
        wrong.n_policy += 1;
        bool threw = false;
        try { Net bad(wrong, model_a, 8, 0, KZ_DTYPE_F32); } catch (const std::invalid_argument &) { threw = true; }
        CHECK(threw);
    }
    auto expect_b = Net(mapper, model_b, 8, 0, KZ_DTYPE_F32).evaluate_batch(boards_b.data(), boards_b.size());

    // 2. two executor threads on one device (gpu_threads_per_device = 2, server_alphazero.rs:89-121), four generator
    //    threads, then a network hot swap to model B (commander.rs:36-45 -> executor.rs:320-342)
    auto [client, server] = job_pair<PackedBoard, ZeroEvaluation>(8);
    std::vector<Sender<std::optional<std::shared_ptr<const HipModel>>>> graph_senders;
    std::vector<std::thread> executors;
    std::atomic<long> evals{0};
    for (int t = 0; t < 2; t++) {
        auto [gtx, grx] = bounded<std::optional<std::shared_ptr<const HipModel>>>(1);
        graph_senders.push_back(gtx);
        executors.emplace_back([&, srv = server, rx = std::move(grx)]() mutable {
            batched_executor_loop<std::shared_ptr<const HipModel>, Net, PackedBoard, ZeroEvaluation>(
                // 16 >= the 4 x 3 boards that can be outstanding: a job is never split over two batches (the loop
                // evaluates one batch per message, so a split job's tail would wait for the next message)
                16, RunCondition::any(), std::move(rx), std::move(srv),
                [&](std::shared_ptr<const HipModel> m) { return Net(mapper, std::move(m), 16, 0, KZ_DTYPE_F32); },
                [&](Net &net, const PackedBoard *x, size_t n) {
                    evals += (long)n;  // the `real` evals counter (server_alphazero.rs:113-115)
                    return net.evaluate_batch(x, n);
                });
        });
    }
    server = JobServer<PackedBoard, ZeroEvaluation>();
    for (auto &g : graph_senders) g.send(model_a);
    auto run_generators = [&](const std::vector<PackedBoard> &bs, const std::vector<ZeroEvaluation> &want) {
        std::atomic<int> wrong{0};
        std::vector<std::thread> gens;
        for (int t = 0; t < 4; t++)
            gens.emplace_back([&, t, c = client] {
                std::mt19937 r(100 + t);
                for (int i = 0; i < 40; i++) {
                    std::vector<PackedBoard> x;
                    std::vector<size_t> idx;
                    const int count = 1 + (int)(r() % 3);
                    for (int k = 0; k < count; k++) {
                        idx.push_back(r() % bs.size());
                        x.push_back(bs[idx.back()]);
                    }
                    auto y = c.map_blocking(std::move(x));
                    if (y.size() != idx.size()) { wrong++; continue; }
                    for (size_t k = 0; k < y.size(); k++)
                        if (!close_eval(y[k], want[idx[k]], 1e-4f)) wrong++;
                }
            });
        for (auto &g : gens) g.join();
        return wrong.load();
    };
    CHECK(run_generators(boards, expect) == 0);
    for (auto &g : graph_senders) g.send(model_b);  // NewNetwork
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    CHECK(run_generators(boards_b, expect_b) == 0);
    CHECK(evals > 0);
    client = JobClient<PackedBoard, ZeroEvaluation>();
    graph_senders.clear();
    for (auto &e : executors) e.join();

    // 2b. ONE executor thread with two batches in flight (pipelined_executor_loop over kz_engine_submit_packed /
    //     kz_engine_wait): same answers, replies in job order, hot swap with work in flight
    {
        // the asynchronous pair alone: two submits, two waits, oldest first
        Net net(mapper, model_a, 8, 0, KZ_DTYPE_F32);
        auto copy = boards;  // submit_batch moves the boards out
        net.submit_batch(copy.data(), 3);
        net.submit_batch(copy.data() + 3, copy.size() - 3);
        CHECK(net.batches_in_flight() == 2);
        // fill the remaining engine slots, then one more must be refused
        std::vector<std::vector<PackedBoard>> extra;
        while (net.batches_in_flight() < (size_t)KZ_ENGINE_SLOTS) {
            extra.push_back(boards);
            net.submit_batch(extra.back().data(), 1);
        }
        bool threw = false;
        auto one = boards;
        try { net.submit_batch(one.data(), 1); } catch (const std::logic_error &) { threw = true; }
        CHECK(threw);  // every engine slot is out
        auto y0 = net.wait_batch(), y1 = net.wait_batch();
        for (size_t i = 0; i < extra.size(); i++) {  // results come back oldest first
            auto ye = net.wait_batch();
            CHECK(ye.size() == 1 && close_eval(ye[0], expect[0], 1e-4f));
        }
        CHECK(net.batches_in_flight() == 0);
        CHECK(y0.size() == 3 && y1.size() == boards.size() - 3);
        for (size_t i = 0; i < y0.size(); i++) CHECK(close_eval(y0[i], expect[i], 1e-4f));
        for (size_t i = 0; i < y1.size(); i++) CHECK(close_eval(y1[i], expect[3 + i], 1e-4f));
    }
    {
        // decode_output on the device: the same evaluations (device expf/tanhf: 1e-5), blocking and pipelined
        Net net(mapper, model_a, 8, 0, KZ_DTYPE_F32);
        net.set_device_decode(true);
        auto y = net.evaluate_batch(boards.data(), boards.size());
        CHECK(y.size() == boards.size());
        for (size_t i = 0; i < y.size(); i++) CHECK(close_eval(y[i], expect[i], 1e-4f));
        auto copy = boards;
        net.submit_batch(copy.data(), 2);
        net.submit_batch(copy.data() + 2, copy.size() - 2);
        auto y0 = net.wait_batch(), y1 = net.wait_batch();
        CHECK(y0.size() == 2 && y1.size() == boards.size() - 2);
        for (size_t i = 0; i < y0.size(); i++) CHECK(close_eval(y0[i], expect[i], 1e-4f));
        for (size_t i = 0; i < y1.size(); i++) CHECK(close_eval(y1[i], expect[2 + i], 1e-4f));
    }
    {
        // prep helpers (set_prep_helpers: a batch's encode_input and move lists shared with helper threads): the same
        // evaluations, bit for bit, with 0, 1 and 3 helpers, host and device decode, blocking and pipelined; 70 boards so
        // that every range holds several
        std::vector<PackedBoard> many;
        for (size_t i = 0; i < 70; i++) many.push_back(boards[(i * 5 + i / 3) % boards.size()]);
        auto same = [](const std::vector<ZeroEvaluation> &a, const std::vector<ZeroEvaluation> &b) {
            if (a.size() != b.size()) return false;
            for (size_t i = 0; i < a.size(); i++)
                if (!close_eval(a[i], b[i], 0.0f)) return false;
            return true;
        };
        for (bool dev : {false, true}) {
            Net ref(mapper, model_a, 128, 0, KZ_DTYPE_F32);
            ref.set_device_decode(dev);
            const auto want = ref.evaluate_batch(many.data(), many.size());
            CHECK(want.size() == many.size());
            for (size_t helpers : {size_t(1), size_t(3)}) {
                Net net(mapper, model_a, 128, 0, KZ_DTYPE_F32);
                net.set_device_decode(dev);
                net.set_prep_helpers(helpers);
                CHECK(same(net.evaluate_batch(many.data(), many.size()), want));
                auto copy = many;
                net.submit_batch(copy.data(), 40);
                net.submit_batch(copy.data() + 40, copy.size() - 40);
                auto y0 = net.wait_batch(), y1 = net.wait_batch();
                y0.insert(y0.end(), y1.begin(), y1.end());
                CHECK(same(y0, want));
                CHECK(net.helper_cpu_ns() > 0);
            }
        }
    }
    {
        auto [pclient, pserver] = job_pair<PackedBoard, ZeroEvaluation>(16);
        auto [gtx, grx] = bounded<std::optional<std::shared_ptr<const HipModel>>>(1);
        std::atomic<long> pevals{0};
        std::thread exec([&, srv = std::move(pserver), rx = std::move(grx)]() mutable {
            pipelined_executor_loop<std::shared_ptr<const HipModel>, Net, PackedBoard, ZeroEvaluation>(
                16, 2, RunCondition::any(), std::move(rx), std::move(srv),
                [&](std::shared_ptr<const HipModel> m) { return Net(mapper, std::move(m), 16, 0, KZ_DTYPE_F32); },
                [](Net &net, PackedBoard *x, size_t n) { net.submit_batch(x, n); },
                [&](Net &net) {
                    auto y = net.wait_batch();
                    pevals += (long)y.size();
                    return y;
                });
        });
        gtx.send(model_a);
        auto saved = client;
        client = pclient;  // run_generators sends through `client`
        CHECK(run_generators(boards, expect) == 0);
        gtx.send(model_b);
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        CHECK(run_generators(boards_b, expect_b) == 0);
        CHECK(pevals > 0);
        client = saved;
        pclient = JobClient<PackedBoard, ZeroEvaluation>();
        gtx = Sender<std::optional<std::shared_ptr<const HipModel>>>();
        exec.join();
    }

    // 3. RandomSymmetryNetwork over HipNetwork on Ataxx positions: values are symmetric-invariant only for a trained
    //    net; what must hold exactly is that the wrapper un-maps the policy it got for the mapped board
    {
        using SymNet = HipNetwork<AtaxxSymBoard, AtaxxStdMapper>;
        AtaxxSymBoard b;
        b.size = 7;
        b.tiles_next = 0b1000001;
        b.tiles_other = 1ull << 48;
        b.moves_since_last_copy = 3;
        b.moves = std::vector<AtaxxMove>{{AtaxxMove::Copy, 0, 0, 1, 0}, {AtaxxMove::Jump, 0, 0, 2, 1}, {AtaxxMove::Copy, 0, 0, 5, 1}};
        SymNet plain(AtaxxStdMapper(7), model_a, 4, 0, KZ_DTYPE_F32);
        RandomSymmetryNetwork<AtaxxSymBoard, SymNet> wrapped(SymNet(AtaxxStdMapper(7), model_a, 4, 0, KZ_DTYPE_F32),
                                                             std::mt19937_64(1), true);
        for (int rep = 0; rep < 8; rep++) {
            auto ev = wrapped.evaluate(b);
            bool found = false;
            for (int sym = 0; sym < 8 && !found; sym++) {
                auto mb = b.map(sym);
                auto direct_ev = plain.evaluate(mb);
                found = close_eval(ev, unmap_eval(b, sym, mb, direct_ev), 1e-6f);
            }
            CHECK(found);
            float sum = 0;
            for (float p : ev.policy) sum += p;
            CHECK(std::fabs(sum - 1.0f) < 1e-5f);
        }
    }
    if (g_failed) {
        std::fprintf(stderr, "%d check(s) failed\n", g_failed);
        return 1;
    }
    std::puts("hip network tests ok");
    return 0;
}
