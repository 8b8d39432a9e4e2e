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
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/kz_hip.h"
#include "kz_kernels.hpp"
#include "kz_model.hpp"

namespace {

#include "kz_engine_util.hpp"     // g_err, fail, guarded, HIP_TRY
#include "kz_device_weights.hpp"  // DevConv, DeviceWeights, the per-(model, device, dtype) cache

#include "kz_engine_state.hpp"    // Prof, kz_model, effective_model, struct kz_engine: streams, slots, staging (closes the namespace itself)
#include "kz_engine_forward.hpp"  // kz_engine::run_tower / run_heads / forward_*: the launches of a forward pass


// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

#define KZ_API __attribute__((visibility("default")))

KZ_API const char *kz_last_error(void) { return g_err.c_str(); }

KZ_API int kz_device_count(int *count) {
    return guarded("kz_device_count", [&]() -> int {
        if (!count) return fail("kz_device_count: null argument");
#ifdef KZ_EXPERIMENTS  // (tests/test_abi.py: the guard itself, on the experiment build only — the product reads no such switch)
        if (const char *t = getenv("KZ_TEST_THROW")) {
            if (t[0] == '1') throw std::length_error("vector::_M_default_append (KZ_TEST_THROW)");
            if (t[0] == '2') throw std::bad_alloc();
            if (t[0] == '3') throw 42;
        }
#endif
        HIP_TRY(hipGetDeviceCount(count));
        return 0;
    });
}

KZ_API int kz_device_pci_bus_id(int device, char *buf, size_t len) {
    return guarded("kz_device_pci_bus_id", [&]() -> int {
        if (!buf || len < 16) return fail("kz_device_pci_bus_id: buffer of at least 16 bytes needed");
        HIP_TRY(hipDeviceGetPCIBusId(buf, (int)len, device));
        return 0;
    });
}

KZ_API int kz_model_load_onnx_memory(const void *blob, size_t len, int input_scalar_channels, kz_model **out) {
    return guarded("kz_model_load_onnx_memory", [&]() -> int {
        if (!blob || !out) return fail("kz_model_load_onnx: null argument");
        std::string err;
        Model *m = kz::parse_onnx(blob, len, input_scalar_channels, err);
        if (!m) return fail("kz_model_load_onnx: " + err);
        *out = new kz_model(std::shared_ptr<Model>(m));
        return 0;
    });
}

KZ_API int kz_model_load_memory(const void *blob, size_t len, kz_model **out) {
    return guarded("kz_model_load_memory", [&]() -> int {
        if (!blob || !out) return fail("kz_model_load_memory: null argument");
        if (kz::looks_like_onnx(blob, len)) return kz_model_load_onnx_memory(blob, len, -1, out);
        std::string err;
        Model *m = kz::parse_model(blob, len, err);
        if (!m) return fail("kz_model_load: " + err);
        *out = new kz_model(std::shared_ptr<Model>(m));
        return 0;
    });
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
    return guarded("kz_model_load", [&]() -> int {
        if (!path || !out) return fail("kz_model_load: null argument");
        std::vector<uint8_t> buf;
        if (read_whole_file(path, buf)) return 1;
        return kz_model_load_memory(buf.data(), buf.size(), out);
    });
}

KZ_API int kz_model_load_onnx(const char *path, int input_scalar_channels, kz_model **out) {
    return guarded("kz_model_load_onnx", [&]() -> int {
        if (!path || !out) return fail("kz_model_load_onnx: null argument");
        std::vector<uint8_t> buf;
        if (read_whole_file(path, buf)) return 1;
        return kz_model_load_onnx_memory(buf.data(), buf.size(), input_scalar_channels, out);
    });
}

KZ_API void kz_model_free(kz_model *model) {
    try {
        delete model;
    } catch (...) {
    }
}

KZ_API int kz_model_get_info(const kz_model *model, kz_model_info *out) {
    return guarded("kz_model_get_info", [&]() -> int {
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
    });
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
    return guarded("kz_engine_create", [&]() -> int {
        if (!model || !out) return fail("kz_engine_create: null argument");
        if (max_batch <= 0) return fail("kz_engine_create: max_batch must be positive");
        {
            // every tensor of an engine is addressed with 32-bit byte offsets (buffer descriptors, int row indices): the
            // largest one — an activation row set in f32, or the policy rows — must stay below 2 GiB.  This also keeps an
            // absurd max_batch (a corrupted settings value) from reaching the allocator.
            const Model &mm = *model->m;
            const size_t per_board = (size_t)4 * std::max<size_t>((size_t)mm.h * mm.w * round_up(std::max(mm.channels, mm.c_in), 64),
                                                                   (size_t)std::max(mm.policy_len, 1));
            const size_t limit = (((size_t)1 << 31) - 1) / per_board;
            if ((size_t)max_batch > limit)
                return fail("kz_engine_create: max_batch " + std::to_string(max_batch) + " too large for this network: at most " +
                            std::to_string(limit) + " boards (every engine tensor must stay below 2 GiB)");
        }
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
    });
}

KZ_API int kz_model_supports_dtype(const kz_model *model, int dtype) {
    return guarded("kz_model_supports_dtype", [&]() -> int {
        if (!model) return -1;
        if (dtype != KZ_DTYPE_F32 && dtype != KZ_DTYPE_F16 && dtype != KZ_DTYPE_F32_SPLIT16) return -1;
        // (larger boards in split arithmetic run per layer: the engine additionally needs max_batch * h * w * channels * 4 bytes
        // < 2 GiB, asked here for one board)
        PathPlan plan;
        std::string why;
        return plan_path(*effective_model(model, dtype, 1), 1, dtype, plan, why) ? 1 : 0;
    }, -1);
}

KZ_API int kz_model_plan(const kz_model *model, int max_batch, int dtype, kz_path_plan *out) {
    return guarded("kz_model_plan", [&]() -> int {
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
    });
}

KZ_API int kz_engine_max_batch(const kz_engine *e) { return e ? e->max_batch : 0; }

KZ_API const char *kz_engine_tower_path(const kz_engine *e) { return e ? e->path.c_str() : ""; }

KZ_API int kz_engine_launch_geometry(const kz_engine *e, int batch, int *workgroups, int *boards_per_workgroup) {
    return guarded("kz_engine_launch_geometry", [&]() -> int {
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
    });
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
    return guarded("kz_engine_submit_packed", [&]() -> int {
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
    });
}

KZ_API int kz_engine_wait(kz_engine *e, int slot, float *scalars_out, float *policy_out) {
    return guarded("kz_engine_wait", [&]() -> int {
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
    });
}

KZ_API int kz_engine_wait_view(kz_engine *e, int slot, const float **scalars_out, const float **policy_out) {
    return guarded("kz_engine_wait_view", [&]() -> int {
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
    });
}

KZ_API int kz_engine_eval_packed(kz_engine *e, const uint8_t *bits, size_t bits_stride, const float *scalars_in,
                                 int batch, float *scalars_out, float *policy_out) {
    return guarded("kz_engine_eval_packed", [&]() -> int {
        if (kz_engine_submit_packed(e, 0, bits, bits_stride, scalars_in, batch)) return 1;
        return kz_engine_wait(e, 0, scalars_out, policy_out);
    });
}

KZ_API int kz_engine_submit_packed_decoded(kz_engine *e, int slot, const uint8_t *bits, size_t bits_stride,
                                           const float *scalars_in, int batch, const int64_t *move_offsets,
                                           const int32_t *move_indices) {
    return guarded("kz_engine_submit_packed_decoded", [&]() -> int {
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
    });
}

KZ_API int kz_engine_wait_decoded(kz_engine *e, int slot, const float **values_out, const float **probs_out) {
    return guarded("kz_engine_wait_decoded", [&]() -> int {
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
    });
}

KZ_API int kz_engine_eval_packed_decoded(kz_engine *e, const uint8_t *bits, size_t bits_stride, const float *scalars_in,
                                         int batch, const int64_t *move_offsets, const int32_t *move_indices,
                                         float *values_out, float *probs_out) {
    return guarded("kz_engine_eval_packed_decoded", [&]() -> int {
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
    });
}

KZ_API int kz_engine_eval_dense(kz_engine *e, const float *input_nchw, int batch, float *scalars_out,
                                float *policy_out) {
    return guarded("kz_engine_eval_dense", [&]() -> int {
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
    });
}

KZ_API int kz_engine_enqueue_packed_device(kz_engine *e, const void *d_bits, size_t bits_stride,
                                           const void *d_scalars_in, int batch, void *d_scalars_out,
                                           void *d_policy_out) {
    return guarded("kz_engine_enqueue_packed_device", [&]() -> int {
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
    });
}

KZ_API int kz_engine_enqueue_dense_device(kz_engine *e, const void *d_input_nchw, int batch, void *d_scalars_out,
                                          void *d_policy_out) {
    return guarded("kz_engine_enqueue_dense_device", [&]() -> int {
        if (check_batch(e, batch, "kz_engine_enqueue_dense_device")) return 1;
        if (batch == 0) return 0;
        if (!d_input_nchw || !d_scalars_out || !d_policy_out) return fail("kz_engine_enqueue_dense_device: null argument");
        HIP_TRY(hipSetDevice(e->device));
        e->arm_device();
        return e->forward_dense(d_input_nchw, batch, d_scalars_out, d_policy_out);
    });
}

KZ_API int kz_engine_synchronize(kz_engine *e) {
    return guarded("kz_engine_synchronize", [&]() -> int {
        if (!e) return fail("kz_engine_synchronize: null engine");
        HIP_TRY(hipSetDevice(e->device));
        if (e->sync_all()) return 1;
        return e->check_devflag();
    });
}

KZ_API int kz_device_malloc(int device, size_t bytes, void **out) {
    return guarded("kz_device_malloc", [&]() -> int {
        if (!out) return fail("kz_device_malloc: null argument");
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
        return 0;
    });
}

KZ_API int kz_device_free(int device, void *ptr) {
    return guarded("kz_device_free", [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipFree(ptr));
        return 0;
    });
}

KZ_API int kz_memcpy_h2d(int device, void *dst, const void *src, size_t bytes) {
    return guarded("kz_memcpy_h2d", [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    });
}

KZ_API int kz_memcpy_d2h(int device, void *dst, const void *src, size_t bytes) {
    return guarded("kz_memcpy_d2h", [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
        return 0;
    });
}

KZ_API int kz_device_synchronize(int device) {
    return guarded("kz_device_synchronize", [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipDeviceSynchronize());
        return 0;
    });
}

KZ_API int kz_engine_set_profiling(kz_engine *e, int enable) {
    return guarded("kz_engine_set_profiling", [&]() -> int {
        if (!e) return fail("kz_engine_set_profiling: null engine");
        HIP_TRY(hipSetDevice(e->device));
        if (e->sync_all()) return 1;
        e->prof.clear();
        e->prof.on = enable != 0;
        return 0;
    });
}

KZ_API int kz_engine_kernel_time(kz_engine *e, const char *prefix, double *total_ms, int64_t *launches) {
    return guarded("kz_engine_kernel_time", [&]() -> int {
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
    });
}

KZ_API int kz_engine_read_activation(kz_engine *e, const char *name, int batch, float *out_nchw) {
    return guarded("kz_engine_read_activation", [&]() -> int {
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
    });
}

}  // extern "C"
