// kz_engine.hip — the executor behind the C ABI of include/kz_hip.h.
//
// One kz_engine = one `CudaNetwork` of the reference (rust/kz-core/src/network/cudnn.rs:18-88): a private HIP stream,
// private activation buffers sized for max_batch, pinned staging for the host-pointer entry points, and a shared,
// reference-counted copy of the device weights per (model, device, dtype).
// Forward schedule per batch: encode (F0) -> tower (one board-resident launch, or stem + 2*depth fused conv launches
// on the generic path) -> ScalarHead -> policy head.  Only `batch` rows are ever touched.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/kz_hip.h"
#include "kz_kernels.hpp"
#include "kz_model.hpp"

namespace {

#include "kz_engine_util.hpp"     // g_err, fail, HIP_TRY
#include "kz_device_weights.hpp"  // DevConv, DeviceWeights, the per-(model, device, dtype) cache

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

    int conv(const DevConv &w, const void *x, int ldx, void *y, int ldy, int M, int relu, const void *res, bool post,
             int h, int wd, int group, int src_group, int src_off, float *y32 = nullptr, int ldy32 = 0) {
        if (w.bws) {  // whole boards as LDS-resident spatial tiles, split arithmetic: (hi, lo) rows of 2 C halves
            kz::BoardConvArgs b{};
            b.x = x; b.ldx = 2 * ldx; b.weights = w.bws; b.bias = w.b; b.res = res; b.y = y; b.ldy = 2 * ldy;
            b.y32 = y32; b.ldy32 = ldy32;
            b.post_scale = post ? wts->post_scale : nullptr;
            b.post_shift = post ? wts->post_shift : nullptr;
            b.boards = M / (h * wd); b.h = h; b.w = wd; b.cin = w.cin_p; b.cout = w.cout; b.relu = relu;
            prof.begin("kz_board_conv_split16", stream);
            kz::launch_board_conv_split(b, stream);
            prof.end(stream);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        if (w.bw) {  // whole boards as LDS-resident spatial tiles
            kz::BoardConvArgs b{};
            b.x = x; b.ldx = ldx; b.weights = w.bw; b.bias = w.b; b.res = res; b.y = y; b.ldy = ldy;
            b.post_scale = post ? wts->post_scale : nullptr;
            b.post_shift = post ? wts->post_shift : nullptr;
            b.boards = M / (h * wd); b.h = h; b.w = wd; b.cin = w.cin_p; b.cout = w.cout; b.relu = relu;
            b.rowmap = wts->bc_rowmap; b.halo = wts->bc_halo; b.n_halo = wts->bc_n_halo;
            prof.begin("kz_board_conv_f16", stream);
#ifdef KZ_EXPERIMENTS
            if (w.bw2) kz::launch_board_conv2(b, stream);
            else
#endif
            kz::launch_board_conv(b, stream);
            prof.end(stream);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        if (w.sw && y && !res && !post && !y32 && ldx >= w.cin_p) {  // 1x1 head convolution: split16 or any f16 path
            kz::Conv1x1SplitArgs c{};
            c.split = split16;
            c.x = x; c.ldx = ldx; c.weights = w.sw; c.bias = w.b; c.y = y; c.ldy = ldy;
            c.M = M; c.cin_p = w.cin_p; c.cout_p = w.cout_p; c.relu = relu;
            c.group = group; c.src_group = src_group; c.src_off = src_off;
            prof.begin("kz_conv1x1_split", stream);
            kz::launch_conv1x1_split(c, stream);
            prof.end(stream);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        kz::ConvArgs a{};
        a.x = x; a.ldx = ldx; a.w = w.w; a.bias = w.b; a.res = res; a.ldres = ldy;
        a.post_scale = post ? wts->post_scale : nullptr;
        a.post_shift = post ? wts->post_shift : nullptr;
        a.y = y; a.y32 = y32; a.ldy = ldy; a.ldy32 = ldy32;
        a.M = M; a.h = h; a.w_ = wd; a.group = group; a.src_group = src_group; a.src_off = src_off;
        a.cin_p = w.cin_p; a.cout_p = w.cout_p; a.cout = w.cout; a.k = w.k; a.relu = relu;
        prof.begin(kz::conv_kernel_name(dtype), stream);
        kz::launch_conv(dtype, a, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // packed != nullptr (resident path only): the launch encodes the boards itself
    struct PackedIn {
        const void *bits;
        size_t stride;
        const void *scalars;
    };
    // the one-launch networks ("...+heads") can end in decode_output (kz_decode_dev.hpp): no decode launch, nothing but the
    // decoded values and the available moves' probabilities leave the launch
    bool decode_in_launch() const { return fused_heads || fused32 || fused_split || fused_pairs; }
    int run_tower(int batch, float *d_scalars, float *d_policy, const PackedIn *packed = nullptr,
                  const kz::DecodeArgs *dec = nullptr) {
        const Model &m = *model;
        const int hw = m.h * m.w, M = batch * hw;
        if (dense_net) {  // DenseNetwork: encoded planes in x_in -> scalars and policy, one launch
            kz::DenseNetArgs t{};
            t.x0 = x_in; t.in_f16 = dtype == KZ_DTYPE_F16; t.batch = batch; t.hw = hw; t.cin_p = cin_p; t.size = m.channels;
            t.depth = m.depth; t.res = m.dn_res ? 1 : 0; t.policy_len = m.policy_len;
            t.w_in = wts->dn_w_in; t.b_in = wts->dn_b_in; t.blocks = wts->dn_blocks; t.sf = wts->dn_sf; t.tf = wts->dn_tf;
            t.w_out = wts->dn_w_out; t.b_out = wts->dn_b_out; t.scalars = d_scalars; t.policy = d_policy;
            t.nonfinite_flag = nf_flag; t.epoch = nf_epoch;
            prof.begin("kz_dense_network", stream);
            kz::launch_dense_network(t, stream);
            prof.end(stream);
            HIP_TRY(hipGetLastError());
            tower_out = 0;
            return 0;
        }
        if (att_f16) {  // AttentionTower on the matrix cores, in the engine's arithmetic
            kz::AttTower16Args t{};
            t.f32 = dtype == KZ_DTYPE_F32;
            t.x0 = x_in; t.cin_p = cin_p; t.w_expand = wts->att16_expand; t.embedding = wts->att_embedding;
            t.w_layers = wts->att16_layers; t.y = act[0]; t.batch = batch; t.depth = m.depth; t.d_model = m.channels;
            t.d_ff = m.att_dff; t.alpha = m.att_alpha; t.eps = m.ln_eps;
            if (packed) {  // fused board encode
                t.bits = (const uint8_t *)packed->bits;
                t.bits_stride = packed->stride;
                t.scalars_in = (const float *)packed->scalars;
                t.n_scalar = m.n_scalar;
                t.n_bool = m.n_bool;
            }
            prof.begin(t.f32 ? "kz_att_tower_f32" : "kz_att_tower_f16", stream);
            kz::launch_att_tower16(t, stream);
            prof.end(stream);
            HIP_TRY(hipGetLastError());
            tower_out = 0;
            return 0;
        }
        if (att_tower) {  // AttentionTower: encoded planes in x_in -> tower output rows in act[0], one launch
            kz::AttTowerArgs t{};
            t.x0 = x_in; t.ldx0 = cin_p; t.in_f16 = dtype == KZ_DTYPE_F16; t.c_in = m.c_in;
            t.expand = wts->att_expand; t.embedding = wts->att_embedding; t.layers = wts->att_layers;
            t.y = act[0]; t.ldy = cp; t.out_f16 = dtype == KZ_DTYPE_F16;
            t.batch = batch; t.h = m.h; t.w = m.w; t.depth = m.depth; t.d_model = m.channels; t.heads = m.att_heads;
            t.d_k = m.att_dk; t.d_v = m.att_dv; t.d_ff = m.att_dff; t.alpha = m.att_alpha; t.eps = m.ln_eps;
            prof.begin("kz_att_tower_f32_valu", stream);
            kz::launch_att_tower(t, stream);
            prof.end(stream);
            HIP_TRY(hipGetLastError());
            tower_out = 0;
            return 0;
        }
        if (resident) {
            kz::TowerArgs t{};
            if (packed) {
                t.bits = (const uint8_t *)packed->bits;
                t.bits_stride = packed->stride;
                t.scalars_in = (const float *)packed->scalars;
                t.n_scalar = m.n_scalar;
                t.n_bool = m.n_bool;
            }
            t.x0 = x_in; t.cin_p = cin_p; t.w_stem = wts->res_w_stem; t.w_tower = wts->res_w_tower;
            t.bias = wts->res_bias; t.post_scale = wts->post_scale; t.post_shift = wts->post_shift;
            t.y = act[0]; t.batch = batch; t.h = m.h; t.w = m.w; t.depth = m.depth;
            t.fused_heads = fused_heads;
            t.sh_w0 = wts->sh_w0; t.sh_b0 = wts->sh_b0; t.sh_w1 = wts->sh_w1; t.sh_b1 = wts->sh_b1;
            t.sh_w2 = wts->sh_w2; t.sh_b2 = wts->sh_b2; t.att_idx = wts->att_idx;
            t.scalars = d_scalars; t.policy = d_policy;
            t.nonfinite_flag = nf_flag; t.epoch = nf_epoch;
            if (dec && fused_heads) t.decode = *dec;
            prof.begin("kz_tower_resident_f16", stream);
#ifdef KZ_EXPERIMENTS
            if (nb4) kz::launch_tower_resident4(t, xres, stream);
            else
#endif
            kz::launch_tower_resident(t, stream);
            prof.end(stream);
            HIP_TRY(hipGetLastError());
            tower_out = 0;
            return 0;
        }
        if (resident32 || pairs16) {
            kz::Tower32Args t{};
            t.x0 = (const float *)x_in; t.ldx0 = cin_p; t.c_in = m.c_in; t.weights = wts->res32_w; t.bias = wts->res_bias;
            t.post_scale = wts->post_scale; t.post_shift = wts->post_shift;
            t.y = (float *)act[0]; t.ldy = cp; t.batch = batch; t.h = m.h; t.w = m.w; t.channels = m.channels;
            t.depth = m.depth;
            if (packed) {  // fused board encode
                t.bits = (const uint8_t *)packed->bits;
                t.bits_stride = packed->stride;
                t.scalars_in = (const float *)packed->scalars;
                t.n_scalar = m.n_scalar;
                t.n_bool = m.n_bool;
            }
            if (fused_split && m.policy_kind == kz::POLICY_ATTENTION) {
                kz::Tower32Args::Heads &hd = t.heads;
                hd.on = true;
                hd.sh_w0 = wts->sh_w0; hd.sh_b0 = wts->sh_b0; hd.sh_w1 = wts->sh_w1; hd.sh_b1 = wts->sh_b1;
                hd.sh_w2 = wts->sh_w2; hd.sh_b2 = wts->sh_b2; hd.att_idx = wts->att_idx;
                hd.policy_len = m.policy_len;
                hd.scalars = d_scalars; hd.policy = d_policy;
                hd.nonfinite_flag = nf_flag; hd.epoch = nf_epoch;
            }
            if (fused32 || fused_pairs || (fused_split && m.policy_kind != kz::POLICY_ATTENTION)) {  // conv policy heads: the f32 tail
                kz::Tower32Args::Heads &hd = t.heads;
                hd.on = true;
                hd.hc = m.sh_conv.cout; hd.hs = m.sh_fc0.out;
                hd.small_w = wts->h32_small; hd.sh_b0 = wts->sh_b0; hd.sh_w1t = wts->sh_w1t; hd.sh_b1 = wts->sh_b1;
                hd.sh_w2 = wts->sh_w2; hd.sh_b2 = wts->sh_b2;
                hd.pc = m.policy_conv_channels; hd.p_b1 = wts->p_b1;
                hd.policy_len = m.policy_len; hd.zero_tail = m.policy_kind == kz::POLICY_ATAXX_CONV ? 1 : 0;
                if (m.policy_kind == kz::POLICY_CONV && m.policy_extra_moves) {
                    hd.extra = m.policy_extra_moves;
                    hd.pe_bc = wts->pe_bc; hd.pe_wl = wts->pe_wl; hd.pe_bl = wts->pe_bl;
                }
                hd.scalars = d_scalars; hd.policy = d_policy;
                hd.nonfinite_flag = nf_flag; hd.epoch = nf_epoch;
            }
            if (dec && t.heads.on) t.heads.decode = *dec;
            t.dense3 = t32_dense3;
            t.wide = wide;
            prof.begin(split16 ? "kz_tower_resident_split" : pairs16 ? "kz_tower_resident_f16g" : "kz_tower_resident_f32", stream);
            if (split16) kz::launch_tower_split(t, stream);
            else if (pairs16) kz::launch_tower_pairs(t, false, stream);  // f16 tensors behind the same pointers
            else kz::launch_tower32(t, stream);
            prof.end(stream);
            HIP_TRY(hipGetLastError());
            tower_out = 0;
            return 0;
        }
        if (bsplit) {
            // the stem in exact f32 (its inputs are f32 planes), its output split into (hi, lo) halves — an f32 tensor and a
            // (hi, lo) tensor of the same shape have the same size, so the three activation buffers serve both —, the
            // 2·depth tower convolutions in split arithmetic, the last one writing f32 for the heads
            if (wts->stem_split) {  // encoded f32 planes [M][32] -> (hi, lo) rows -> the board-tile kernel, one chunk
                prof.begin("kz_split_rows", stream);
                kz::launch_split_rows((const float *)x_in, act[2], (size_t)M, cin_p, stream);
                prof.end(stream);
                if (conv(wts->tower[0], act[2], cin_p, act[0], cp, M, 0, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
            } else {
                if (conv(wts->tower[0], x_in, cin_p, act[2], cp, M, 0, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
                prof.begin("kz_split_rows", stream);
                kz::launch_split_rows((const float *)act[2], act[0], (size_t)M, cp, stream);
                prof.end(stream);
            }
            int cur = 0;
            for (int i = 1; i <= m.depth; i++) {
                const int mid = (cur + 1) % 3, nxt = (cur + 2) % 3;
                const bool last = i == m.depth;
                if (conv(wts->tower[2 * i - 1], act[cur], cp, act[mid], cp, M, 1, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
                if (conv(wts->tower[2 * i], act[mid], cp, last ? nullptr : act[nxt], cp, M, 1, act[cur], last, m.h, m.w, hw, hw, 0,
                         last ? (float *)act[nxt] : nullptr, cp))
                    return 1;
                cur = nxt;
            }
            tower_out = cur;
            return 0;
        }
        // stem: conv + bias, no activation (post_act.py:205)
        if (conv(wts->tower[0], x_in, cin_p, act[0], cp, M, 0, nullptr, m.depth == 0, m.h, m.w, hw, hw, 0)) return 1;
        if (stash(m.depth == 0 ? "tower.1" : "tower.0", act[0], batch)) return 1;
        int cur = 0;
        for (int i = 1; i <= m.depth; i++) {
            const int mid = (cur + 1) % 3, nxt = (cur + 2) % 3;
            const bool last = i == m.depth;
            if (conv(wts->tower[2 * i - 1], act[cur], cp, act[mid], cp, M, 1, nullptr, false, m.h, m.w, hw, hw, 0))
                return 1;
            if (stash("tower." + std::to_string(i) + ".mid", act[mid], batch)) return 1;
            // x + relu(bn(conv(mid))) (post_act.py:227-228); the tower's final BN rides on the last block
            if (conv(wts->tower[2 * i], act[mid], cp, act[nxt], cp, M, 1, act[cur], last, m.h, m.w, hw, hw, 0))
                return 1;
            if (stash("tower." + std::to_string(last ? i + 1 : i), act[nxt], batch)) return 1;
            cur = nxt;
        }
        tower_out = cur;
        return 0;
    }

    bool extra_in_scalar_head() const {
        const Model &m = *model;
        return m.policy_kind == kz::POLICY_CONV && m.policy_extra_moves > 0 && wts->sh_w0x &&
               kz::scalar_head_takes_extra(dtype == KZ_DTYPE_F32 || split16 ? 0 : 1, cp, m.sh_conv.cout);
    }

    int run_heads(int batch, float *d_scalars, float *d_policy) {
        if (fused_heads || fused32 || fused_split || fused_pairs || dense_net) return 0;  // written by the tower launch
        const Model &m = *model;
        const int hw = m.h * m.w, M = batch * hw;
        const void *x = act[tower_out];
        if (wts->att_heads) {  // ScalarHead + AttentionPolicyHead in one launch
            kz::AttHeadsArgs a{};
            a.x = x; a.ldx = cp; a.batch = batch; a.channels = m.channels; a.q = m.policy_query_channels;
            a.hc = m.sh_conv.cout; a.hs = m.sh_fc0.out; a.policy_len = m.policy_len;
            a.weights = wts->ah_w; a.bias = wts->ah_bias;
            a.w1 = wts->sh_w1; a.b1 = wts->sh_b1; a.w2 = wts->sh_w2; a.b2 = wts->sh_b2;
            a.flat_to_att = wts->flat_to_att; a.scalars = d_scalars; a.policy = d_policy;
            a.nonfinite_flag = nf_flag; a.epoch = nf_epoch;
            prof.begin("kz_att_heads_f16", stream);
            kz::launch_att_heads(a, stream);
            prof.end(stream);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        {
            kz::ScalarHeadArgs a{x, cp, batch, hw, m.channels, m.sh_conv.cout, m.sh_fc0.out,
                                 wts->sh_w0, wts->sh_b0, wts->sh_w1, wts->sh_b1, wts->sh_w2, wts->sh_b2, d_scalars,
                                 nf_flag, nf_epoch, wts->sh_w1t};
            // ConvPolicyHead's extra moves read the same tower output: one pass for both (post_act.py:63-67)
            if (extra_in_scalar_head()) {
                a.extra = m.policy_extra_moves;
                a.w0x = wts->sh_w0x; a.pe_bc = wts->pe_bc; a.pe_wl = wts->pe_wl; a.pe_bl = wts->pe_bl;
                a.policy = d_policy; a.policy_len = m.policy_len; a.policy_offset = m.policy_conv_channels * hw;
            }
            prof.begin("kz_scalar_head", stream);
            kz::launch_scalar_head(dtype, a, stream);
            prof.end(stream);
        }
        switch (m.policy_kind) {
            case kz::POLICY_ATAXX_CONV:
            case kz::POLICY_CONV: {
                const int pc = m.policy_conv_channels;
                const DevConv &c0 = wts->p_conv0;
                if (m.policy_kind == kz::POLICY_CONV && c0.sw && cp >= c0.cin_p &&
                    kz::conv1x1_policy_epilogue_supported(c0.cin_p, c0.cout_p, c0.cout, pc)) {
                    // Conv1x1 C->C + ReLU + Conv1x1 C->1 in one launch: the hidden layer never goes to memory
                    kz::Conv1x1SplitArgs c{};
                    c.split = split16;
                    c.x = x; c.ldx = cp; c.weights = c0.sw; c.bias = c0.b; c.y = nullptr; c.ldy = 0;
                    c.M = M; c.cin_p = c0.cin_p; c.cout_p = c0.cout_p; c.relu = 1;
                    c.group = hw; c.src_group = hw; c.src_off = 0;
                    c.pw1 = wts->p_w1; c.pb1 = wts->p_b1; c.policy = d_policy; c.policy_len = m.policy_len; c.hw = hw;
                    prof.begin("kz_conv1x1_split", stream);
                    kz::launch_conv1x1_split(c, stream);
                    prof.end(stream);
                } else {
                    if (conv(c0, x, cp, head0, c0.cout_p, M, 1, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
                    kz::PolicyConvArgs a{head0, c0.cout_p, batch, hw, m.channels, pc, wts->p_w1, wts->p_b1,
                                         d_policy, m.policy_len, m.policy_kind == kz::POLICY_ATAXX_CONV ? 1 : 0};
                    prof.begin("kz_policy_conv", stream);
                    kz::launch_policy_conv(dtype, a, stream);
                    prof.end(stream);
                }
                if (m.policy_kind == kz::POLICY_CONV && m.policy_extra_moves && !extra_in_scalar_head()) {
                    kz::PolicyExtraArgs e{x, cp, batch, hw, m.channels, m.policy_extra_moves, wts->pe_wc, wts->pe_bc,
                                          wts->pe_wl, wts->pe_bl, d_policy, m.policy_len, pc * hw};
                    prof.begin("kz_policy_extra", stream);
                    kz::launch_policy_extra(dtype, e, stream);
                    prof.end(stream);
                }
                break;
            }
            case kz::POLICY_ARIMAA: {
                // ArimaaPolicyHead (post_act.py:144-173): policy = concat(scalar(common) [1 + 6], flatten(bulk(common)) [4 * hw])
                const DevConv &c0 = wts->p_conv0;
                if (conv(c0, x, cp, head0, c0.cout_p, M, 1, nullptr, false, m.h, m.w, hw, hw, 0)) return 1;
                kz::PolicyConvArgs a{head0, c0.cout_p, batch, hw, m.channels, 4, wts->p_w1, wts->p_b1,
                                     d_policy + 7, m.policy_len, 0};  // (the four planes start behind the seven scalars)
                prof.begin("kz_policy_conv", stream);
                kz::launch_policy_conv(dtype, a, stream);
                prof.end(stream);
                // the scalar branch has the ScalarHead's shape: the same kernel, seven outputs into the policy rows
                kz::ScalarHeadArgs sa{x, cp, batch, hw, m.channels, m.arimaa_hidden_channels, m.arimaa_hidden_size,
                                      wts->pa_w0, wts->pa_b0, wts->pa_w1, wts->pa_b1, wts->pa_w2, wts->pa_b2, d_policy,
                                      nullptr, 0, wts->pa_w1t};
                sa.n_out = 7;
                sa.out_ld = m.policy_len;
                prof.begin("kz_scalar_head", stream);
                kz::launch_scalar_head(dtype, sa, stream);
                prof.end(stream);
                break;
            }
            case kz::POLICY_ATTENTION: {
                // bulk = conv_bulk(common) on all 64 squares; under = conv_under(common[:, :, 7, None, :]) on the
                // 8 squares of rank index 7 (post_act.py:128-129): source rows 56..63 of each board
                if (conv(wts->p_bulk, x, cp, head0, wts->p_bulk.cout_p, M, 0, nullptr, false, m.h, m.w, hw, hw, 0))
                    return 1;
                if (conv(wts->p_under, x, cp, head1, wts->p_under.cout_p, batch * 8, 0, nullptr, false, 1, 8, 8, hw, 56))
                    return 1;
                kz::AttentionArgs a{head0, head1, wts->p_bulk.cout_p, wts->p_under.cout_p, batch,
                                    m.policy_query_channels, wts->flat_to_att, d_policy, m.policy_len};
                prof.begin("kz_attention_gather", stream);
                kz::launch_attention(dtype, a, stream);
                prof.end(stream);
                break;
            }
            case kz::POLICY_NONE: break;
            case kz::POLICY_DENSE: {
                const void *flat = x;
                int flat_ld = hw * cp;
                if (m.dense_hidden_channels) {
                    if (conv(wts->p_conv0, x, cp, head0, wts->p_conv0.cout_p, M, 1, nullptr, false, m.h, m.w, hw, hw, 0))
                        return 1;
                    flat = head0;
                    flat_ld = hw * wts->p_conv0.cout_p;
                }
                // Flatten + Linear: one GEMM row per board
                if (m.dense_hidden_size) {
                    if (conv(wts->p_fc0, flat, flat_ld, head1, wts->p_fc0.cout_p, batch, 1, nullptr, false, 1, 1, 1, 1, 0))
                        return 1;
                    if (conv(wts->p_fc1, head1, wts->p_fc0.cout_p, nullptr, 0, batch, 0, nullptr, false, 1, 1, 1, 1, 0,
                             d_policy, m.policy_len))
                        return 1;
                } else {
                    if (conv(wts->p_fc1, flat, flat_ld, nullptr, 0, batch, 0, nullptr, false, 1, 1, 1, 1, 0, d_policy,
                             m.policy_len))
                        return 1;
                }
                break;
            }
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // dec (decode_in_launch() engines only): the launch ends in decode_output and writes dec->values / dec->probs
    int forward_packed(const void *d_bits, size_t stride, const void *d_sin, int batch, void *d_sout, void *d_pol,
                       const kz::DecodeArgs *dec = nullptr) {
        const Model &m = *model;
        if (resident || resident32 || pairs16 || att_f16) {  // encode is fused into the tower launch
            const PackedIn in{d_bits, stride, d_sin};
            if (run_tower(batch, (float *)d_sout, (float *)d_pol, &in, dec)) return 1;
            return run_heads(batch, (float *)d_sout, (float *)d_pol);
        }
        prof.begin("kz_encode_packed", stream);
        kz::launch_encode_packed(dtype, (const uint8_t *)d_bits, stride, (const float *)d_sin, batch, m.n_scalar,
                                 m.n_bool, m.h * m.w, x_in, cin_p, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        if (run_tower(batch, (float *)d_sout, (float *)d_pol)) return 1;
        return run_heads(batch, (float *)d_sout, (float *)d_pol);
    }

    int forward_dense(const void *d_nchw, int batch, void *d_sout, void *d_pol) {
        const Model &m = *model;
        prof.begin("kz_encode_dense", stream);
        kz::launch_encode_dense(dtype, (const float *)d_nchw, batch, m.c_in, m.h * m.w, x_in, cin_p, stream);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        if (run_tower(batch, (float *)d_sout, (float *)d_pol)) return 1;
        return run_heads(batch, (float *)d_sout, (float *)d_pol);
    }
};

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

#define KZ_API __attribute__((visibility("default")))

KZ_API const char *kz_last_error(void) { return g_err.c_str(); }

KZ_API int kz_device_count(int *count) {
    if (!count) return fail("kz_device_count: null argument");
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}

KZ_API int kz_device_pci_bus_id(int device, char *buf, size_t len) {
    if (!buf || len < 16) return fail("kz_device_pci_bus_id: buffer of at least 16 bytes needed");
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int)len, device));
    return 0;
}

KZ_API int kz_model_load_onnx_memory(const void *blob, size_t len, int input_scalar_channels, kz_model **out) {
    if (!blob || !out) return fail("kz_model_load_onnx: null argument");
    std::string err;
    Model *m = kz::parse_onnx(blob, len, input_scalar_channels, err);
    if (!m) return fail("kz_model_load_onnx: " + err);
    *out = new kz_model(std::shared_ptr<Model>(m));
    return 0;
}

KZ_API int kz_model_load_memory(const void *blob, size_t len, kz_model **out) {
    if (!blob || !out) return fail("kz_model_load_memory: null argument");
    if (kz::looks_like_onnx(blob, len)) return kz_model_load_onnx_memory(blob, len, -1, out);
    std::string err;
    Model *m = kz::parse_model(blob, len, err);
    if (!m) return fail("kz_model_load: " + err);
    *out = new kz_model(std::shared_ptr<Model>(m));
    return 0;
}

static int read_whole_file(const char *path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail(std::string("kz_model_load: cannot open '") + path + "'");
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return 0;
}

KZ_API int kz_model_load(const char *path, kz_model **out) {
    if (!path || !out) return fail("kz_model_load: null argument");
    std::vector<uint8_t> buf;
    if (read_whole_file(path, buf)) return 1;
    return kz_model_load_memory(buf.data(), buf.size(), out);
}

KZ_API int kz_model_load_onnx(const char *path, int input_scalar_channels, kz_model **out) {
    if (!path || !out) return fail("kz_model_load_onnx: null argument");
    std::vector<uint8_t> buf;
    if (read_whole_file(path, buf)) return 1;
    return kz_model_load_onnx_memory(buf.data(), buf.size(), input_scalar_channels, out);
}

KZ_API void kz_model_free(kz_model *model) { delete model; }

KZ_API int kz_model_get_info(const kz_model *model, kz_model_info *out) {
    if (!model || !out) return fail("kz_model_get_info: null argument");
    const Model &m = *model->m;
    out->input_channels = m.c_in;
    out->board_h = m.h;
    out->board_w = m.w;
    out->input_scalar_channels = m.n_scalar;
    out->input_bool_channels = m.n_bool;
    out->policy_len = m.policy_len;
    out->tower_depth = m.depth;
    out->tower_channels = m.channels;
    out->policy_kind = (int)m.policy_kind;
    out->bits_bytes = m.n_bool < 0 ? -1 : (m.n_bool * m.h * m.w + 7) / 8;
    out->param_count = m.param_count;
    out->flops_per_eval = m.flops_per_eval;
    return 0;
}

KZ_API void kz_engine_destroy(kz_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (auto st : e->slot_stream)
        if (st) (void)hipStreamSynchronize(st);
    e->prof.destroy();
#ifdef KZ_EXPERIMENTS
    for (auto &g : e->graphs) (void)hipGraphExecDestroy(g.exec);
#endif
    for (auto &s : e->slots)
        if (s.done) (void)hipEventDestroy(s.done);
    for (void *p : e->allocs) (void)hipFree(p);
    for (void *p : e->pinned) (void)hipHostFree(p);
    for (int i = 0; i < KZ_ENGINE_SLOTS; i++)
        if (e->slot_stream[i] && (i < 2 || e->slot_stream[i] != e->slot_stream[i & 1])) (void)hipStreamDestroy(e->slot_stream[i]);
    e->wts.reset();
    delete e;
}

KZ_API int kz_engine_create(const kz_model *model, int device, int max_batch, int dtype, kz_engine **out) {
    if (!model || !out) return fail("kz_engine_create: null argument");
    if (max_batch <= 0) return fail("kz_engine_create: max_batch must be positive");
    if (dtype != KZ_DTYPE_F32 && dtype != KZ_DTYPE_F16 && dtype != KZ_DTYPE_F32_SPLIT16)
        return fail("kz_engine_create: unknown dtype");
    // KZ_DTYPE_F32_SPLIT16 is the f32 engine with one kernel exchanged: everything below sees KZ_DTYPE_F32
    const bool split16 = dtype == KZ_DTYPE_F32_SPLIT16;
    if (split16) dtype = KZ_DTYPE_F32;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail("kz_engine_create: device " + std::to_string(device) + " out of range (" + std::to_string(ndev) +
                    " visible)");
    HIP_TRY(hipSetDevice(device));

    std::unique_ptr<kz_engine, void (*)(kz_engine *)> e(new kz_engine(), kz_engine_destroy);
    e->model = effective_model(model, split16 ? KZ_DTYPE_F32_SPLIT16 : dtype, max_batch);
    e->out_channels = model->m->channels;
    const Model &m = *e->model;
    e->device = device;
    e->dtype = dtype;
    e->max_batch = max_batch;
    e->esz = dtype == KZ_DTYPE_F32 ? 4 : 2;
    e->cin_p = round_up(m.c_in, 32);
    e->cp = round_up(m.channels, 32);
    // which kernels run this network: plan_path (above) — the table of DESIGN.md §5.0 is printed from it
    PathPlan plan;
    {
        std::string why;
        if (!plan_path(m, max_batch, split16 ? KZ_DTYPE_F32_SPLIT16 : dtype, plan, why)) return fail("kz_engine_create: " + why);
    }
    e->dense_net = plan.dense_net;
    e->att_tower = plan.att_tower;
    e->att_f16 = plan.att_f16;
    e->resident = plan.resident;
    e->fused_heads = plan.fused_heads;
    e->keep = plan.keep;
    e->resident32 = plan.resident32;
    e->split16 = plan.split16;
    e->bsplit = plan.bsplit;
    e->pairs16 = plan.pairs16;
    e->wide = plan.wide;
    e->fused_pairs = plan.fused_pairs;
    e->fused32 = plan.fused32;
    e->fused_split = plan.fused_split;
    e->path = plan.path;
    const bool board_conv = plan.board_conv;
#ifdef KZ_EXPERIMENTS
    const char *notower = getenv("KZ_NO_TOWER_F16");  // (chess f16 through the generic one-launch f16 tower)
    if (notower && notower[0] == '1' && e->resident) {
        e->resident = e->fused_heads = false;
        e->pairs16 = kz::tower_split_supported(m.h, m.w, m.channels, m.depth, m.c_in, false);
        e->wide = false;
        e->path = e->pairs16 ? "tower_resident_f16g" : "conv_igemm_f16";
    }
    const char *nb_env = getenv("KZ_TOWER_NB");
    e->nb4 = e->resident && nb_env && atoi(nb_env) == 4 && e->cin_p == 32;
    if (e->nb4) {  // (the four-board launch has no fused heads yet)
        e->fused_heads = false;
        e->path = "tower_resident_f16";
    }
    const char *t32b = getenv("KZ_T32_BOARDS");
    e->t32_dense3 = e->resident32 && !e->split16 && !e->pairs16 && t32b && atoi(t32b) == 3 &&
                    kz::tower32_dense3_supported((int)m.policy_kind, m.policy_extra_moves, m.policy_conv_channels, m.h, m.w,
                                                 m.channels, m.sh_conv.cout, m.sh_fc0.out, e->fused32);
#endif

    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        int variant = 0;
#ifdef KZ_EXPERIMENTS
        const char *c2 = getenv("KZ_BOARD_CONV2");  // (the opt-in board-conv organisation has its own weight packing)
        variant = c2 && c2[0] == '1' ? 400 : 0;
#endif
        const bool att_heads = !e->fused_heads && att_heads_one_launch(m, dtype, e->split16);
        auto key = std::make_tuple(e->model.get(), device,
                                   dtype + (e->split16 ? 100 : 0) + (e->pairs16 ? 200 : 0) + (e->att_f16 ? 800 : 0) + (att_heads ? 1600 : 0) + variant,
                                   e->resident || e->resident32,
                                   e->fused_heads || e->fused_split || e->fused_pairs, board_conv);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) e->wts = it->second.lock();
        if (!e->wts) {
            auto w = std::make_shared<DeviceWeights>();
            w->device = device;
            w->dtype = dtype;
            w->use_board_conv = board_conv;
            w->use_board_split = e->bsplit;
            w->fused_split = e->fused_split;
            w->fused_pairs = e->fused_pairs;
            w->att_f16 = e->att_f16;
            w->att_heads = att_heads;
            if (w->build(m, e->resident, e->fused_heads, e->resident32, e->split16, e->pairs16)) return 1;
            g_cache[key] = w;
            e->wts = w;
        }
    }

#ifdef KZ_EXPERIMENTS
    {
        const char *hg = getenv("KZ_HIP_GRAPH");
        e->use_graph = hg && hg[0] == '1' && !e->fused_heads && !e->fused32 && !e->fused_split && !e->fused_pairs;
    }
#endif
    HIP_TRY(hipStreamCreateWithFlags(&e->slot_stream[0], hipStreamNonBlocking));
    e->stream = e->slot_stream[0];
    if (e->fused_heads || e->fused32 || e->fused_split || e->fused_pairs) {  // one launch per batch that touches nothing but its slot's buffers
        HIP_TRY(hipStreamCreateWithFlags(&e->slot_stream[1], hipStreamNonBlocking));
        for (int i = 2; i < KZ_ENGINE_SLOTS; i++) e->slot_stream[i] = e->slot_stream[i & 1];
        e->zero_copy = true;
#ifdef KZ_EXPERIMENTS
        const char *nzc = getenv("KZ_NO_ZERO_COPY");  // (the staged-copy variant of the one-launch paths, for A/B timing)
        if (nzc && nzc[0] == '1') e->zero_copy = false;
#endif
    }
    if (e->wts->stem_cin_p) e->cin_p = e->wts->stem_cin_p;
    const size_t hw = (size_t)m.h * m.w, rows = (size_t)max_batch * hw;
    if (e->dmalloc(&e->x_in, rows * e->cin_p * e->esz)) return 1;
#ifdef KZ_EXPERIMENTS
    if (e->nb4 && e->dmalloc(&e->xres, kz::tower4_scratch_bytes(max_batch))) return 1;
#endif
    const int nact = (e->resident || e->resident32 || e->pairs16 || e->att_tower || e->dense_net) ? 1 : 3;
    for (int i = 0; i < nact; i++)
        if (e->dmalloc(&e->act[i], rows * e->cp * e->esz)) return 1;
    // head temporaries
    size_t h0 = 0, h1 = 0;
    const DeviceWeights &w = *e->wts;
    if (!e->fused_heads) switch (m.policy_kind) {
        case kz::POLICY_ATAXX_CONV:
        case kz::POLICY_ARIMAA:
        case kz::POLICY_CONV: h0 = rows * w.p_conv0.cout_p; break;
        case kz::POLICY_ATTENTION:
            h0 = rows * w.p_bulk.cout_p;
            h1 = (size_t)max_batch * 8 * w.p_under.cout_p;
            break;
        case kz::POLICY_NONE: break;
        case kz::POLICY_DENSE:
            if (m.dense_hidden_channels) h0 = rows * w.p_conv0.cout_p;
            if (m.dense_hidden_size) h1 = (size_t)max_batch * w.p_fc0.cout_p;
            break;
    }
    if (e->dmalloc(&e->head0, h0 * e->esz) || e->dmalloc(&e->head1, h1 * e->esz)) return 1;

    const int nb_planes = m.n_bool < 0 ? 0 : m.n_bool, ns_planes = m.n_scalar < 0 ? 0 : m.n_scalar;
    const size_t bits_bytes = (size_t)(nb_planes * hw + 7) / 8;
    for (auto &s : e->slots) {
        if (e->dmalloc((void **)&s.d_bits, max_batch * bits_bytes) ||
            e->dmalloc((void **)&s.d_sin, (size_t)max_batch * ns_planes * 4) ||
            e->dmalloc((void **)&s.d_sout, ((size_t)max_batch * 5 + kz_engine::SOUT_HDR) * 4) ||
            e->dmalloc((void **)&s.d_pol, (size_t)max_batch * m.policy_len * 4))
            return 1;
        if (e->hmalloc((void **)&s.h_bits, max_batch * bits_bytes) ||
            e->hmalloc((void **)&s.h_sin, (size_t)max_batch * ns_planes * 4) ||
            e->hmalloc((void **)&s.h_sout, ((size_t)max_batch * 5 + kz_engine::SOUT_HDR) * 4) ||
            e->hmalloc((void **)&s.h_pol, (size_t)max_batch * m.policy_len * 4))
            return 1;
        HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HIP_TRY(hipMemset(s.d_sout, 0, kz_engine::SOUT_HDR * 4));
        memset(s.h_sout, 0, kz_engine::SOUT_HDR * 4);
    }
    if (e->dmalloc((void **)&e->d_devflag, 16)) return 1;
    HIP_TRY(hipMemset(e->d_devflag, 0, 16));
    *out = e.release();
    return 0;
}

KZ_API int kz_model_supports_dtype(const kz_model *model, int dtype) {
    if (!model) return -1;
    if (dtype != KZ_DTYPE_F32 && dtype != KZ_DTYPE_F16 && dtype != KZ_DTYPE_F32_SPLIT16) return -1;
    // (larger boards in split arithmetic run per layer: the engine additionally needs max_batch * h * w * channels * 4 bytes
    // < 2 GiB, asked here for one board)
    PathPlan plan;
    std::string why;
    return plan_path(*effective_model(model, dtype, 1), 1, dtype, plan, why) ? 1 : 0;
}

KZ_API int kz_model_plan(const kz_model *model, int max_batch, int dtype, kz_path_plan *out) {
    if (!model || !out) return fail("kz_model_plan: null argument");
    if (max_batch <= 0) return fail("kz_model_plan: max_batch must be positive");
    if (dtype != KZ_DTYPE_F32 && dtype != KZ_DTYPE_F16 && dtype != KZ_DTYPE_F32_SPLIT16) return fail("kz_model_plan: unknown dtype");
    PathPlan plan;
    std::string why;
    if (!plan_path(*effective_model(model, dtype, max_batch), max_batch, dtype, plan, why)) return fail("kz_model_plan: " + why);
    memset(out, 0, sizeof *out);
    snprintf(out->tower_path, sizeof out->tower_path, "%s", plan.path.c_str());
    out->launches_per_batch = plan.launches;
    return 0;
}

KZ_API int kz_engine_max_batch(const kz_engine *e) { return e ? e->max_batch : 0; }

KZ_API const char *kz_engine_tower_path(const kz_engine *e) { return e ? e->path.c_str() : ""; }

KZ_API int kz_engine_launch_geometry(const kz_engine *e, int batch, int *workgroups, int *boards_per_workgroup) {
    if (!e || !workgroups || !boards_per_workgroup) return fail("kz_engine_launch_geometry: null argument");
    if ((batch < 0 || batch > e->max_batch ? fail("kz_engine_launch_geometry: batch out of range") : 0)) return 1;
    const Model &m = *e->model;
    int per = 0, wgs = 0;
    if (e->dense_net) per = 1;
    else if (e->att_f16) per = kz::att_tower16_boards_per_workgroup(m.channels, m.att_dff, batch, e->dtype == KZ_DTYPE_F32);
    else if (e->att_tower) per = 1;  // a workgroup is a board
    else if (e->resident) per = e->nb4 ? 4 : e->cin_p > 32 ? 2 : kz::tower_resident_boards_per_workgroup();
    else if ((e->split16 && !e->bsplit) || e->pairs16) per = kz::tower_split_boards_per_workgroup(m.h, m.w, m.channels, e->split16,
                                                         e->wide ? batch : 0);  // (per launch: the widest level this batch fills the chip with)
    else if (e->resident32) per = e->t32_dense3 ? 3 : kz::tower32_boards_per_workgroup(m.h, m.w, m.channels);
    if (per) wgs = (batch + per - 1) / per;
    else if (e->path == "board_conv_split16") wgs = kz::board_conv_workgroups(batch, m.h, m.w, m.channels);
    else if (e->path == "board_conv_f16")
#ifdef KZ_EXPERIMENTS
        wgs = e->wts->conv2 ? kz::board_conv2_workgroups(batch, m.channels) : kz::board_conv_workgroups(batch, m.h, m.w, m.channels);
#else
        wgs = kz::board_conv_workgroups(batch, m.h, m.w, m.channels);
#endif
    else wgs = kz::conv_workgroups(e->dtype, batch * m.h * m.w, e->cp);
    *workgroups = wgs;
    *boards_per_workgroup = per;
    return 0;
}

static int check_packed(const kz_engine *e, const char *fn) {
    if (e && e->model->n_scalar < 0)
        return fail(std::string(fn) + ": the model was loaded from ONNX without the scalar/bool plane split; load it "
                                      "with kz_model_load_onnx(path, input_scalar_channels) to use packed inputs");
    return 0;
}

static int check_batch(const kz_engine *e, int batch, const char *fn) {
    if (!e) return fail(std::string(fn) + ": null engine");
    if (batch < 0 || batch > e->max_batch)  // assert!(batch_size <= max_batch_size), cudnn.rs:58
        return fail(std::string(fn) + ": batch " + std::to_string(batch) + " exceeds max_batch " +
                    std::to_string(e->max_batch));
    return 0;
}

KZ_API int kz_engine_submit_packed(kz_engine *e, int slot, const uint8_t *bits, size_t bits_stride,
                                   const float *scalars_in, int batch) {
    if (check_batch(e, batch, "kz_engine_submit_packed") || check_packed(e, "kz_engine_submit_packed")) return 1;
    if (slot < 0 || slot >= KZ_ENGINE_SLOTS) return fail("kz_engine_submit_packed: bad slot");
    kz_engine::Slot &s = e->slots[slot];
    if (s.batch >= 0) return fail("kz_engine_submit_packed: slot still in flight (call kz_engine_wait first)");
    const Model &m = *e->model;
    const size_t bits_bytes = (size_t)(m.n_bool * m.h * m.w + 7) / 8;
    if (batch > 0 && (!bits || (m.n_scalar && !scalars_in))) return fail("kz_engine_submit_packed: null input");
    if (batch > 0 && bits_stride < bits_bytes) return fail("kz_engine_submit_packed: bits_stride too small");
    HIP_TRY(hipSetDevice(e->device));
    if (batch == 0) {
        s.batch = 0;
        return 0;
    }
    for (int b = 0; b < batch; b++) memcpy(s.h_bits + b * bits_bytes, bits + b * bits_stride, bits_bytes);
    if (m.n_scalar) memcpy(s.h_sin, scalars_in, (size_t)batch * m.n_scalar * 4);
    // on the fused path every slot has its own stream, so two submitted batches run side by side (each resident
    // launch covers half of the CUs at batch 256); otherwise the slots share the activation buffers and one stream
    struct StreamSwap {
        kz_engine *e;
        hipStream_t saved;
        ~StreamSwap() { e->stream = saved; }
    } swap{e, e->stream};
    if (e->slot_stream[slot]) e->stream = e->slot_stream[slot];
    if (e->zero_copy) {
        // the one launch reads 136 B per board from pinned host memory and writes its 7.5 KB per board there
        e->arm(s);
        e->nf_flag = reinterpret_cast<int *>(s.h_sout);
        if (e->forward_packed(s.h_bits, bits_bytes, s.h_sin, batch, s.h_sout + kz_engine::SOUT_HDR, s.h_pol)) return 1;
        HIP_TRY(hipEventRecord(s.done, e->stream));
        s.batch = batch;
        return 0;
    }
    HIP_TRY(hipMemcpyAsync(s.d_bits, s.h_bits, batch * bits_bytes, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(s.d_sin, s.h_sin, (size_t)batch * m.n_scalar * 4, hipMemcpyHostToDevice, e->stream));
#ifdef KZ_EXPERIMENTS
    if (e->graph_mode()) {
        s.epoch = kz_engine::GRAPH_EPOCH;
        e->nf_flag = reinterpret_cast<int *>(s.d_sout);
        e->nf_epoch = s.epoch;
        if (e->replay(slot, batch, s.d_bits, bits_bytes, s.d_sin, s.d_sout, s.d_pol, [&]() -> int {
                HIP_TRY(hipMemsetAsync(s.d_sout, 0, 4, e->stream));
                return e->forward_packed(s.d_bits, bits_bytes, s.d_sin, batch, s.d_sout + kz_engine::SOUT_HDR, s.d_pol);
            }))
            return 1;
    } else
#endif
    {
        e->arm(s);
        if (e->forward_packed(s.d_bits, bits_bytes, s.d_sin, batch, s.d_sout + kz_engine::SOUT_HDR, s.d_pol)) return 1;
#ifdef KZ_EXPERIMENTS
        e->graph_warm = true;  // (the first pass runs eagerly: lazy per-kernel set-up must not land in a capture)
#endif
    }
    HIP_TRY(hipMemcpyAsync(s.h_sout, s.d_sout, ((size_t)batch * 5 + kz_engine::SOUT_HDR) * 4, hipMemcpyDeviceToHost,
                           e->stream));
    HIP_TRY(hipMemcpyAsync(s.h_pol, s.d_pol, (size_t)batch * m.policy_len * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipEventRecord(s.done, e->stream));
    s.batch = batch;  // in flight only once the event is recorded: a failed submit leaves the slot free
    return 0;
}

KZ_API int kz_engine_wait(kz_engine *e, int slot, float *scalars_out, float *policy_out) {
    if (!e) return fail("kz_engine_wait: null engine");
    if (slot < 0 || slot >= KZ_ENGINE_SLOTS) return fail("kz_engine_wait: bad slot");
    kz_engine::Slot &s = e->slots[slot];
    if (s.batch < 0 || s.decoded) return fail("kz_engine_wait: nothing submitted on this slot");
    const int batch = s.batch;
    s.batch = -1;
    if (batch == 0) return 0;
    if (!scalars_out || !policy_out) return fail("kz_engine_wait: null output");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventSynchronize(s.done));
    memcpy(scalars_out, s.h_sout + kz_engine::SOUT_HDR, (size_t)batch * 5 * 4);
    memcpy(policy_out, s.h_pol, (size_t)batch * e->model->policy_len * 4);
    if (kz_engine::slot_nonfinite(s)) return fail(kz_engine::nonfinite_message("kz_engine_wait"));
    return 0;
}

KZ_API int kz_engine_wait_view(kz_engine *e, int slot, const float **scalars_out, const float **policy_out) {
    if (!e) return fail("kz_engine_wait_view: null engine");
    if (slot < 0 || slot >= KZ_ENGINE_SLOTS) return fail("kz_engine_wait_view: bad slot");
    if (!scalars_out || !policy_out) return fail("kz_engine_wait_view: null output");
    kz_engine::Slot &s = e->slots[slot];
    if (s.batch < 0 || s.decoded) return fail("kz_engine_wait_view: nothing submitted on this slot");
    const int batch = s.batch;
    s.batch = -1;
    *scalars_out = s.h_sout + kz_engine::SOUT_HDR;
    *policy_out = s.h_pol;
    if (batch == 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventSynchronize(s.done));
    if (kz_engine::slot_nonfinite(s)) return fail(kz_engine::nonfinite_message("kz_engine_wait_view"));
    return 0;
}

KZ_API int kz_engine_eval_packed(kz_engine *e, const uint8_t *bits, size_t bits_stride, const float *scalars_in,
                                 int batch, float *scalars_out, float *policy_out) {
    if (kz_engine_submit_packed(e, 0, bits, bits_stride, scalars_in, batch)) return 1;
    return kz_engine_wait(e, 0, scalars_out, policy_out);
}

KZ_API int kz_engine_submit_packed_decoded(kz_engine *e, int slot, const uint8_t *bits, size_t bits_stride,
                                           const float *scalars_in, int batch, const int64_t *move_offsets,
                                           const int32_t *move_indices) {
    const char *fn = "kz_engine_submit_packed_decoded";
    if (check_batch(e, batch, fn) || check_packed(e, fn)) return 1;
    if (slot < 0 || slot >= KZ_ENGINE_SLOTS) return fail(std::string(fn) + ": bad slot");
    kz_engine::Slot &s = e->slots[slot];
    if (s.batch >= 0) return fail(std::string(fn) + ": slot still in flight (call kz_engine_wait_decoded first)");
    if (batch == 0) {
        s.batch = 0;
        s.decoded = true;
        s.moves = 0;
        return 0;
    }
    if (!bits || !move_offsets) return fail(std::string(fn) + ": null argument");
    const Model &m = *e->model;
    const size_t bits_bytes = (size_t)(m.n_bool * m.h * m.w + 7) / 8;
    if (bits_stride < bits_bytes) return fail(std::string(fn) + ": bits_stride too small");
    if (m.n_scalar && !scalars_in) return fail(std::string(fn) + ": null scalars");
    if (move_offsets[0] != 0) return fail(std::string(fn) + ": move_offsets[0] must be 0");
    for (int b = 0; b < batch; b++)
        if (move_offsets[b + 1] < move_offsets[b]) return fail(std::string(fn) + ": move_offsets must be non-decreasing");
    const size_t total = (size_t)move_offsets[batch];
    if (total && !move_indices) return fail(std::string(fn) + ": null move list");
    HIP_TRY(hipSetDevice(e->device));
    if (!s.h_moff) {  // (pinned only: the decode reads and writes the host staging directly, on every path)
        if (e->hmalloc((void **)&s.h_moff, (size_t)(e->max_batch + 1) * 8) || e->hmalloc((void **)&s.h_values, (size_t)e->max_batch * 20) ||
            e->hmalloc((void **)&s.h_err, 16))
            return 1;
    }
    if (total > s.move_cap) {  // the old (smaller) buffers stay on the engine's free list until it is destroyed
        const size_t cap = std::max(total, std::max(s.move_cap * 2, (size_t)e->max_batch * 64));
        if (e->hmalloc((void **)&s.h_midx, cap * 4) || e->hmalloc((void **)&s.h_probs, cap * 4)) return 1;
        s.move_cap = cap;
    }
    for (int b = 0; b < batch; b++) memcpy(s.h_bits + b * bits_bytes, bits + b * bits_stride, bits_bytes);
    if (m.n_scalar) memcpy(s.h_sin, scalars_in, (size_t)batch * m.n_scalar * 4);
    memcpy(s.h_moff, move_offsets, (size_t)(batch + 1) * 8);
    if (total) memcpy(s.h_midx, move_indices, total * 4);
    s.h_err[0] = s.h_err[1] = 0;
    struct StreamSwap {
        kz_engine *e;
        hipStream_t saved;
        ~StreamSwap() { e->stream = saved; }
    } swap{e, e->stream};
    if (e->slot_stream[slot]) e->stream = e->slot_stream[slot];
    if (e->zero_copy && e->decode_in_launch()) {
        // ONE launch and no copy operation: it reads the packed boards and the move lists from the slot's pinned staging
        // and writes the decoded values and the available moves' probabilities there (0.2 KB per chess evaluation cross
        // PCIe); decode_output (common.rs:16-100) is the launch's last step.  The conv policy heads keep their logits in
        // device memory (s.d_pol) for the gather; the attention network's never leave LDS.
        e->arm(s);
        e->nf_flag = reinterpret_cast<int *>(s.h_sout);
        const kz::DecodeArgs dec{s.h_moff, s.h_midx, s.h_values, s.h_probs, s.h_err};
        if (e->forward_packed(s.h_bits, bits_bytes, s.h_sin, batch, s.d_sout + kz_engine::SOUT_HDR, s.d_pol, &dec)) return 1;
        HIP_TRY(hipEventRecord(s.done, e->stream));
        s.batch = batch;
        s.decoded = s.in_launch = true;
        s.moves = total;
        return 0;
    }
    // heads in launches of their own: the network leaves scalars and logits in device memory, the stand-alone decode kernel
    // reads the move lists from and writes values / probabilities / flags to the slot's pinned staging directly (every word
    // once): the two input copies are the only copy operations of the batch
    HIP_TRY(hipMemcpyAsync(s.d_bits, s.h_bits, batch * bits_bytes, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(s.d_sin, s.h_sin, (size_t)batch * m.n_scalar * 4, hipMemcpyHostToDevice, e->stream));
    e->arm(s);
    if (e->forward_packed(s.d_bits, bits_bytes, s.d_sin, batch, s.d_sout + kz_engine::SOUT_HDR, s.d_pol)) return 1;
    e->prof.begin("kz_decode_output", e->stream);
    kz::launch_decode_output(s.d_sout + kz_engine::SOUT_HDR, s.d_pol, batch, m.policy_len, s.h_moff, s.h_midx, s.h_values,
                             s.h_probs, s.h_err, reinterpret_cast<const int *>(s.d_sout), s.epoch, e->stream);
    e->prof.end(e->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s.done, e->stream));
    s.batch = batch;
    s.decoded = true;
    s.in_launch = false;
    s.moves = total;
    return 0;
}

KZ_API int kz_engine_wait_decoded(kz_engine *e, int slot, const float **values_out, const float **probs_out) {
    if (!e) return fail("kz_engine_wait_decoded: null engine");
    if (slot < 0 || slot >= KZ_ENGINE_SLOTS) return fail("kz_engine_wait_decoded: bad slot");
    if (!values_out || !probs_out) return fail("kz_engine_wait_decoded: null output");
    kz_engine::Slot &s = e->slots[slot];
    if (s.batch < 0 || !s.decoded) return fail("kz_engine_wait_decoded: nothing submitted with a move list on this slot");
    const int batch = s.batch;
    s.batch = -1;
    s.decoded = false;
    *values_out = s.h_values;
    *probs_out = s.h_probs;
    if (batch == 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventSynchronize(s.done));
    if (s.in_launch && kz_engine::slot_nonfinite(s)) return fail(kz_engine::nonfinite_message("kz_engine_wait_decoded"));
    if (s.h_err[1]) return fail(kz_engine::nonfinite_message("kz_engine_wait_decoded"));
    if (s.h_err[0]) return fail("kz_engine_wait_decoded: Softmax input sum must be strictly positive (or a move index is out of range)");
    return 0;
}

KZ_API int kz_engine_eval_packed_decoded(kz_engine *e, const uint8_t *bits, size_t bits_stride, const float *scalars_in,
                                         int batch, const int64_t *move_offsets, const int32_t *move_indices,
                                         float *values_out, float *probs_out) {
    if (check_batch(e, batch, "kz_engine_eval_packed_decoded") || check_packed(e, "kz_engine_eval_packed_decoded")) return 1;
    if (batch == 0) return 0;
    if (!values_out) return fail("kz_engine_eval_packed_decoded: null argument");
    if (move_offsets && batch > 0 && move_offsets[batch] > 0 && !probs_out) return fail("kz_engine_eval_packed_decoded: null move list");
    if (kz_engine_submit_packed_decoded(e, 0, bits, bits_stride, scalars_in, batch, move_offsets, move_indices)) return 1;
    const float *values = nullptr, *probs = nullptr;
    const size_t total = e->slots[0].moves;
    if (kz_engine_wait_decoded(e, 0, &values, &probs)) return 1;
    memcpy(values_out, values, (size_t)batch * 20);
    if (total) memcpy(probs_out, probs, total * 4);
    return 0;
}

KZ_API int kz_engine_eval_dense(kz_engine *e, const float *input_nchw, int batch, float *scalars_out,
                                float *policy_out) {
    if (check_batch(e, batch, "kz_engine_eval_dense")) return 1;
    if (batch == 0) return 0;
    if (!input_nchw || !scalars_out || !policy_out) return fail("kz_engine_eval_dense: null argument");
    kz_engine::Slot &s = e->slots[0];
    if (s.batch >= 0) return fail("kz_engine_eval_dense: slot 0 still in flight");
    const Model &m = *e->model;
    HIP_TRY(hipSetDevice(e->device));
    const size_t per = (size_t)m.c_in * m.h * m.w * 4;
    if (!e->d_dense) {
        if (e->dmalloc((void **)&e->d_dense, e->max_batch * per) || e->hmalloc((void **)&e->h_dense, e->max_batch * per))
            return 1;
    }
    memcpy(e->h_dense, input_nchw, batch * per);
    HIP_TRY(hipMemcpyAsync(e->d_dense, e->h_dense, batch * per, hipMemcpyHostToDevice, e->stream));
    e->arm(s);
    if (e->forward_dense(e->d_dense, batch, s.d_sout + kz_engine::SOUT_HDR, s.d_pol)) return 1;
    HIP_TRY(hipMemcpyAsync(s.h_sout, s.d_sout, ((size_t)batch * 5 + kz_engine::SOUT_HDR) * 4, hipMemcpyDeviceToHost,
                           e->stream));
    HIP_TRY(hipMemcpyAsync(s.h_pol, s.d_pol, (size_t)batch * m.policy_len * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(scalars_out, s.h_sout + kz_engine::SOUT_HDR, (size_t)batch * 5 * 4);
    memcpy(policy_out, s.h_pol, (size_t)batch * m.policy_len * 4);
    if (kz_engine::slot_nonfinite(s)) return fail(kz_engine::nonfinite_message("kz_engine_eval_dense"));
    return 0;
}

KZ_API int kz_engine_enqueue_packed_device(kz_engine *e, const void *d_bits, size_t bits_stride,
                                           const void *d_scalars_in, int batch, void *d_scalars_out,
                                           void *d_policy_out) {
    if (check_batch(e, batch, "kz_engine_enqueue_packed_device") || check_packed(e, "kz_engine_enqueue_packed_device"))
        return 1;
    if (batch == 0) return 0;
    if (!d_bits || !d_scalars_out || !d_policy_out) return fail("kz_engine_enqueue_packed_device: null argument");
    const Model &m = *e->model;
    if (bits_stride < (size_t)(m.n_bool * m.h * m.w + 7) / 8)
        return fail("kz_engine_enqueue_packed_device: bits_stride too small");
    HIP_TRY(hipSetDevice(e->device));
    e->arm_device();
#ifdef KZ_EXPERIMENTS
    if (e->graph_mode()) {
        e->nf_epoch = kz_engine::GRAPH_EPOCH;
        return e->replay(-1, batch, d_bits, bits_stride, d_scalars_in, d_scalars_out, d_policy_out, [&]() -> int {
            return e->forward_packed(d_bits, bits_stride, d_scalars_in, batch, d_scalars_out, d_policy_out);
        });
    }
    e->graph_warm = true;
#endif
    return e->forward_packed(d_bits, bits_stride, d_scalars_in, batch, d_scalars_out, d_policy_out);
}

KZ_API int kz_engine_enqueue_dense_device(kz_engine *e, const void *d_input_nchw, int batch, void *d_scalars_out,
                                          void *d_policy_out) {
    if (check_batch(e, batch, "kz_engine_enqueue_dense_device")) return 1;
    if (batch == 0) return 0;
    if (!d_input_nchw || !d_scalars_out || !d_policy_out) return fail("kz_engine_enqueue_dense_device: null argument");
    HIP_TRY(hipSetDevice(e->device));
    e->arm_device();
    return e->forward_dense(d_input_nchw, batch, d_scalars_out, d_policy_out);
}

KZ_API int kz_engine_synchronize(kz_engine *e) {
    if (!e) return fail("kz_engine_synchronize: null engine");
    HIP_TRY(hipSetDevice(e->device));
    if (e->sync_all()) return 1;
    return e->check_devflag();
}

KZ_API int kz_device_malloc(int device, size_t bytes, void **out) {
    if (!out) return fail("kz_device_malloc: null argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
    return 0;
}

KZ_API int kz_device_free(int device, void *ptr) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(ptr));
    return 0;
}

KZ_API int kz_memcpy_h2d(int device, void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

KZ_API int kz_memcpy_d2h(int device, void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

KZ_API int kz_device_synchronize(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

KZ_API int kz_engine_set_profiling(kz_engine *e, int enable) {
    if (!e) return fail("kz_engine_set_profiling: null engine");
    HIP_TRY(hipSetDevice(e->device));
    if (e->sync_all()) return 1;
    e->prof.clear();
    e->prof.on = enable != 0;
    return 0;
}

KZ_API int kz_engine_kernel_time(kz_engine *e, const char *prefix, double *total_ms, int64_t *launches) {
    if (!e || !prefix || !total_ms || !launches) return fail("kz_engine_kernel_time: null argument");
    HIP_TRY(hipSetDevice(e->device));
    if (e->sync_all()) return 1;
    double total = 0;
    int64_t n = 0;
    const size_t plen = strlen(prefix);
    for (auto &r : e->prof.recs) {
        if (r.name.compare(0, plen, prefix) != 0) continue;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        total += ms;
        n++;
    }
    *total_ms = total;
    *launches = n;
    return 0;
}

KZ_API int kz_engine_read_activation(kz_engine *e, const char *name, int batch, float *out_nchw) {
    if (!e || !name || !out_nchw) return fail("kz_engine_read_activation: null argument");
    // "tower.out": the tower output of the last evaluation, on every path that materialises it (all but the fused-heads
    // launch)
    const bool tower_out = std::string(name) == "tower.out" && !e->fused_heads && !e->fused32 && !e->fused_split && !e->fused_pairs;
    if (!e->keep && !tower_out)
        return fail("kz_engine_read_activation: engine keeps no activations (create it with KZ_FORCE_GENERIC=1 and "
                    "KZ_KEEP_ACTIVATIONS=1; \"tower.out\" is available on every path without fused heads)");
    auto it = e->kept.find(name);
    if (!tower_out && it == e->kept.end())
        return fail(std::string("kz_engine_read_activation: no activation named '") + name + "'");
    const void *src_act = tower_out ? e->act[e->tower_out] : it->second;
    if (check_batch(e, batch, "kz_engine_read_activation")) return 1;
    const Model &m = *e->model;
    const int hw = m.h * m.w, C = e->out_channels, cp = e->cp;
    HIP_TRY(hipSetDevice(e->device));
    if (e->sync_all()) return 1;
    std::vector<uint8_t> raw((size_t)batch * hw * cp * e->esz);
    HIP_TRY(hipMemcpy(raw.data(), src_act, raw.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < batch; b++)
        for (int c = 0; c < C; c++)
            for (int p = 0; p < hw; p++) {
                const size_t src = ((size_t)b * hw + p) * cp + c;
                float v;
                if (e->dtype == KZ_DTYPE_F32) {
                    memcpy(&v, raw.data() + src * 4, 4);
                } else {
                    _Float16 h;
                    memcpy(&h, raw.data() + src * 2, 2);
                    v = (float)h;
                }
                out_nchw[((size_t)b * C + c) * hw + p] = v;
            }
    return 0;
}

}  // extern "C"
