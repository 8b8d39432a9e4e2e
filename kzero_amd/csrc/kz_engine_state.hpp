// kz_engine_state.hpp — what an engine IS: the per-launch profiler, the model handle, and `struct kz_engine` with its
// streams, slots (pinned staging + device buffers), range-check epochs and allocation bookkeeping.  The forward pass over
// that state — which kernels run, in which order — is declared here and defined in kz_engine_forward.hpp; the `extern "C"`
// entry points that drive both are kz_engine.hip.  Included ONCE, by kz_engine.hip, inside its anonymous namespace (the
// first part) — the same arrangement as kz_engine_util.hpp / kz_device_weights.hpp / kz_plan.hpp.
#pragma once

struct Prof {
    struct Rec {
        std::string name;
        hipEvent_t a, b;
    };
    bool on = false;
    std::vector<Rec> recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    void clear() {
        for (auto &r : recs) pool.push_back({r.a, r.b});
        recs.clear();
    }
    void destroy() {
        clear();
        for (auto &p : pool) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        pool.clear();
    }
    void begin(const char *name, hipStream_t s) {
        if (!on) return;
        Rec r;
        r.name = name;
        if (!pool.empty()) {
            r.a = pool.back().first;
            r.b = pool.back().second;
            pool.pop_back();
        } else {
            (void)hipEventCreate(&r.a);
            (void)hipEventCreate(&r.b);
        }
        (void)hipEventRecord(r.a, s);
        recs.push_back(r);
    }
    void end(hipStream_t s) {
        if (!on) return;
        (void)hipEventRecord(recs.back().b, s);
    }
};


#include "kz_plan.hpp"  // PathPlan, plan_path: which kernels run a network

}  // namespace

struct kz_model {
    std::shared_ptr<Model> m;
    // the tower widened to a multiple of 64 channels by zero filters (kz::pad_channels), built at first use; null when the
    // channel count is one already
    mutable std::mutex widened_mutex;
    mutable std::shared_ptr<Model> widened;
    mutable bool widened_tried = false;
    explicit kz_model(std::shared_ptr<Model> model) : m(std::move(model)) {}
};

namespace {
// The network the kernels of `dtype` run: the model itself, or — f16 / split arithmetic, a tower of 48, 96, 160 ...
// channels — the same network widened to the next multiple of 64 channels: zero filters cost (Cpad / C)^2 of the
// multiply-adds and buy the one-launch and board-tile kernels instead of the generic implicit GEMM (chess x 96 channels,
// f16: 0.53M -> 1.0M evals/s; x 160: 0.24M -> 0.6M).  Exact f32 keeps its implicit GEMM (the f32 one-launch tower exists
// for 128 / 256 channels only and the f32 matrix rate makes the zero work expensive).
std::shared_ptr<Model> effective_model(const kz_model *model, int dtype_in, int max_batch) {
    const Model &m = *model->m;
    if (dtype_in == KZ_DTYPE_F32 || m.tower_kind != kz::TOWER_RES || m.channels % 64 == 0 || m.channels > 512 || m.depth < 1 || env_on("KZ_FORCE_GENERIC") ||
        env_on("KZ_KEEP_ACTIVATIONS"))
        return model->m;
    std::shared_ptr<Model> wide;
    {
        std::lock_guard<std::mutex> lock(model->widened_mutex);
        if (!model->widened_tried) {
            model->widened_tried = true;
            model->widened.reset(kz::pad_channels(m, round_up(m.channels, 64)));
        }
        wide = model->widened;
    }
    if (!wide) return model->m;
    // The zero filters only pay when they buy another kernel: a widened tower that still takes the generic implicit GEMM
    // (Go 19x19 x 96 channels at max_batch 8: too few workgroups for the board-tile kernel) would run (Cpad / C)^2 of the
    // multiply-adds through the same kernel.  Keep the network as it is then — unless it is refused as it is (split16).
    PathPlan pw, po;
    std::string why;
    if (!plan_path(*wide, max_batch, dtype_in, pw, why)) return model->m;
    if (pw.path.compare(0, 10, "conv_igemm") == 0 && plan_path(m, max_batch, dtype_in, po, why)) return model->m;
    return wide;
}
}  // namespace

struct kz_engine {
    std::shared_ptr<Model> model;
    std::shared_ptr<DeviceWeights> wts;
    int device = 0, dtype = 0, max_batch = 0;
    int out_channels = 0;  // the network's own tower channels (model->channels may be widened: effective_model)
    size_t esz = 4;
    hipStream_t stream = nullptr;          // the stream the forward pass is currently enqueued on
    // [0] = the main stream.  On the fused path (one launch per batch, which touches nothing but its slot's buffers) slots
    // alternate over TWO streams — a batch of 256 is half a chip of workgroups, so two launches run side by side and the
    // next launch of a stream starts the moment the previous one ends — and the launch reads the packed boards from and
    // writes the results to the slot's pinned host staging directly (zero copy): no H2D/D2H operation sits between two
    // launches of a stream.  Otherwise all slots share the main stream and staging is copied.
    hipStream_t slot_stream[KZ_ENGINE_SLOTS] = {};
    bool zero_copy = false;
    int sync_all() {
        for (auto st : slot_stream)
            if (st) HIP_TRY(hipStreamSynchronize(st));
        return 0;
    }
    std::vector<void *> allocs, pinned;
    bool dense_net = false;  // DenseNetwork: kz_dense_network.hip runs the whole network
    bool att_tower = false;  // AttentionTower network: kz_att_tower.hip runs the tower
    bool att_f16 = false;    // ... kz_att_tower_f16.hip does
    bool resident = false, fused_heads = false, resident32 = false, split16 = false, pairs16 = false;
    bool bsplit = false;  // split16 per layer through kz_board_conv_split16 (Go-size boards)
    bool wide = false;    // the plain-f16 one-launch tower with twice the boards per workgroup (PathPlan::wide)
    bool fused32 = false;  // the exact-f32 resident launch with the conv policy head and the scalar head inside
    bool fused_split = false;  // the split-f16 launch with the scalar head and the policy head inside
    bool fused_pairs = false;  // the plain-f16 generic launch with the conv policy head and the scalar head inside
    bool nb4 = false;        // resident chess tower with four boards per workgroup (KZ_TOWER_NB=4)
    bool t32_dense3 = false;  // exact-f32 launch with three 7x7 boards per workgroup (experiment build: KZ_T32_BOARDS=3)
    void *xres = nullptr;    // its residual scratch
    std::string path;

    // activations
    int cin_p = 0, cp = 0;
    void *x_in = nullptr;
    void *act[3] = {nullptr, nullptr, nullptr};
    void *head0 = nullptr, *head1 = nullptr;  // head temporaries
    int tower_out = 0;

    // host-pointer entry points: per-slot device io + pinned staging
    struct Slot {
        uint8_t *d_bits = nullptr, *h_bits = nullptr;
        float *d_sin = nullptr, *h_sin = nullptr;
        // d_sout / h_sout start with a 16-byte header: [0] = the range-check flag (kz::ScalarHeadArgs::nonfinite_flag),
        // so that it crosses PCIe in the same copy as the scalars
        float *d_sout = nullptr, *h_sout = nullptr;
        float *d_pol = nullptr, *h_pol = nullptr;
        hipEvent_t done = nullptr;
        int batch = -1;
        int epoch = 0;  // what the flag reads when this submission saw a non-finite activation
        // device-side decode (N2): CSR move lists, decoded values, probabilities, error flag; grown on demand
        bool decoded = false;  // what is in flight was submitted with a move list
        bool in_launch = false;  // ... and decoded by the network's own launch (the range check reports in h_sout's header)
        size_t move_cap = 0, moves = 0;
        int64_t *h_moff = nullptr;
        int32_t *h_midx = nullptr;
        float *h_values = nullptr, *h_probs = nullptr;
        int *h_err = nullptr;  // [0] softmax sum / move index, [1] range check (kz_kernels.hpp: launch_decode_output)
    } slots[KZ_ENGINE_SLOTS];
    float *d_dense = nullptr, *h_dense = nullptr;
    static constexpr int SOUT_HDR = 4;  // floats in front of the scalars
    // range check (see kz::ScalarHeadArgs): every submission gets a new epoch; a kernel that meets a non-finite
    // activation raises the flag it was given to that epoch.  No reset between batches is needed.
    // Epochs run 1 .. GRAPH_EPOCH-1 and start over (0 is the cleared word, GRAPH_EPOCH the replayed passes' constant):
    // at ~2k submissions/s an int would overflow after 12 days of self-play.
    int epoch = 0;
    int *nf_flag = nullptr;  // what the running forward pass writes to
    int nf_epoch = 0;
    int *d_devflag = nullptr;  // flag of the device-resident entry points, checked by kz_engine_synchronize
    int dev_epoch_enqueued = 0;  // epoch of the last device-resident enqueue (slot submissions do not touch d_devflag)
    int dev_epoch_checked = 0;
    int next_epoch() {
        if (epoch >= GRAPH_EPOCH - 1) {  // start over: settle the device-resident flag first (slot flags compare for equality)
            (void)sync_all();
            check_devflag_pending();
            if (d_devflag) (void)hipMemset(d_devflag, 0, 4);
            // the slots' own flag words too: a slot that once recorded a non-finite batch at epoch X keeps X in its header,
            // and X is about to be issued again (everything is idle here: sync_all above)
            for (auto &s : slots) {
                if (s.batch >= 0) continue;  // (a finished batch nobody has waited for yet keeps its verdict)
                if (s.d_sout) (void)hipMemset(s.d_sout, 0, 4);
                if (s.h_sout) *reinterpret_cast<int *>(s.h_sout) = 0;
                s.epoch = 0;
            }
            epoch = dev_epoch_enqueued = dev_epoch_checked = 0;
        }
        return ++epoch;
    }
    bool wrap_nonfinite_pending = false;  // a non-finite batch seen while starting the epochs over: reported by the next synchronize
    void check_devflag_pending() {
        if (!d_devflag || dev_epoch_checked == dev_epoch_enqueued) return;
        int v = 0;
        if (hipMemcpy(&v, d_devflag, 4, hipMemcpyDeviceToHost) == hipSuccess && v != GRAPH_EPOCH && v > dev_epoch_checked)
            wrap_nonfinite_pending = true;
    }
    void arm_device() {  // the forward pass enqueued next reports into the device-resident flag
        nf_flag = d_devflag;
        nf_epoch = dev_epoch_enqueued = next_epoch();
    }

    // hipGraph replay of the forward pass (KZ_HIP_GRAPH=1; multi-launch paths only — the one-launch paths have nothing to
    // replay): the launches of one (entry point, batch size, buffers) are captured once from the engine's own stream and
    // replayed with one hipGraphLaunch.  A captured kernel argument cannot change, so the range check of a replayed pass
    // reports a CONSTANT epoch: per slot the flag word is cleared by a captured memset, for the device-resident entry
    // points kz_engine_synchronize clears it after reporting.
    static constexpr int GRAPH_EPOCH = 0x7fffffff;
#ifndef KZ_EXPERIMENTS
    static constexpr bool graph_mode() { return false; }  // the replay is an experiment build's switch (no gain measured)
#else
    bool use_graph = false, graph_warm = false;
    struct GraphEntry {
        int kind, batch;  // kind: slot index, or -1 for the device-resident entry point
        const void *bits;
        size_t stride;
        const void *sin;
        void *sout, *pol;
        hipGraphExec_t exec;
    };
    std::vector<GraphEntry> graphs;
    bool graph_mode() const { return use_graph && graph_warm && !prof.on && !keep; }
    // runs `body` (which enqueues on `stream`) through the graph of this key: captured at first sight
    template <class Body>
    int replay(int kind, int batch, const void *bits, size_t stride, const void *sin, void *sout, void *pol, Body body) {
        for (const GraphEntry &g : graphs)
            if (g.kind == kind && g.batch == batch && g.bits == bits && g.stride == stride && g.sin == sin && g.sout == sout &&
                g.pol == pol) {
                HIP_TRY(hipGraphLaunch(g.exec, stream));
                return 0;
            }
        if (graphs.size() >= 32) return body();  // (a caller cycling through many shapes: eager)
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        const int rc = body();
        const hipError_t end = hipStreamEndCapture(stream, &graph);
        if (rc || end != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc ? rc : fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(end));
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (inst != hipSuccess) return fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(inst));
        graphs.push_back({kind, batch, bits, stride, sin, sout, pol, exec});
        HIP_TRY(hipGraphLaunch(exec, stream));
        return 0;
    }
#endif
    void arm(Slot &s) {  // the forward pass enqueued next reports into this slot's header
        s.epoch = next_epoch();
        nf_flag = reinterpret_cast<int *>(s.d_sout);
        nf_epoch = s.epoch;
    }
    static bool slot_nonfinite(const Slot &s) { return *reinterpret_cast<const int *>(s.h_sout) == s.epoch; }
    int check_devflag() {
        if (wrap_nonfinite_pending) {
            wrap_nonfinite_pending = false;
            return fail(nonfinite_message("kz_engine_synchronize"));
        }
        // nothing device-resident enqueued since the last check: no blocking copy (slot submissions report per slot)
        if (!d_devflag || dev_epoch_checked == dev_epoch_enqueued) return 0;
        int v = 0;
        HIP_TRY(hipMemcpy(&v, d_devflag, 4, hipMemcpyDeviceToHost));
        const int since = dev_epoch_checked;
        dev_epoch_checked = dev_epoch_enqueued;
        if (v == GRAPH_EPOCH) HIP_TRY(hipMemset(d_devflag, 0, 4));  // (a replayed pass cannot carry a fresh epoch)
        if (v > since) return fail(nonfinite_message("kz_engine_synchronize"));
        return 0;
    }
    static std::string nonfinite_message(const char *fn) {
        return std::string(fn) + ": non-finite activation in the network output of this batch (beyond +-65504 the f16 "
               "and split-f16 paths overflow: evaluate this network with KZ_DTYPE_F32)";
    }

    // debugging
    bool keep = false;
    std::map<std::string, void *> kept;

    Prof prof;

    int dmalloc(void **p, size_t bytes) {
        HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
        allocs.push_back(*p);
        return 0;
    }
    int hmalloc(void **p, size_t bytes) {
        HIP_TRY(hipHostMalloc(p, bytes ? bytes : 16, hipHostMallocDefault));
        pinned.push_back(*p);
        return 0;
    }

    int stash(const std::string &name, const void *src, int batch) {
        if (!keep) return 0;
        const size_t bytes = (size_t)batch * model->h * model->w * cp * esz;
        auto it = kept.find(name);
        if (it == kept.end()) {
            void *p = nullptr;
            if (dmalloc(&p, (size_t)max_batch * model->h * model->w * cp * esz)) return 1;
            it = kept.emplace(name, p).first;
        }
        HIP_TRY(hipMemcpyAsync(it->second, src, bytes, hipMemcpyDeviceToDevice, stream));
        return 0;
    }

    // ---- the forward pass (kz_engine_forward.hpp) ----
    struct PackedIn {  // packed boards still to be encoded (the one-launch towers encode inside the launch)
        const void *bits;
        size_t stride;
        const void *scalars;
    };
    int conv(const DevConv &w, const void *x, int ldx, void *y, int ldy, int M, int relu, const void *res, bool post,
             int h, int wd, int group, int src_group, int src_off, float *y32 = nullptr, int ldy32 = 0);
    bool decode_in_launch() const { return fused_heads || fused32 || fused_split || fused_pairs; }
    int run_tower(int batch, float *d_scalars, float *d_policy, const PackedIn *packed = nullptr,
                  const kz::DecodeArgs *dec = nullptr);
    bool extra_in_scalar_head() const;
    int run_heads(int batch, float *d_scalars, float *d_policy);
    int forward_packed(const void *d_bits, size_t stride, const void *d_sin, int batch, void *d_sout, void *d_pol,
                       const kz::DecodeArgs *dec = nullptr);
    int forward_dense(const void *d_nchw, int batch, void *d_sout, void *d_pol);
};
