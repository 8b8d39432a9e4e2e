/*
 * kz_hip.h — C ABI of the MI355X-native self-play NN executor for kZero.
 *
 * This is the drop-in boundary: the entry points a Rust `HipNetwork<B, M>: Network<B>` binds over FFI in place
 * of the kn-cuda-eval `CudaExecutor` used by `CudaNetwork` (rust/kz-core/src/network/cudnn.rs:18-88).  Plain
 * pointers and sizes only.  Every function returns 0 on success and non-zero on error; `kz_last_error()` gives
 * the message (the reference panics on every error: `unwrap()` cudnn.rs:70,78, `assert!` :58 — the Rust shim
 * turns a non-zero return into a panic to keep that behaviour).
 *
 * Threading contract (mirrors the reference): `kz_model` is immutable and may be shared by any number of threads
 * and devices (it is the `Arc<Graph>` sent to every executor, rust/kz-selfplay/src/server/commander.rs:36-45).
 * `kz_engine` is NOT thread-safe: one per executor thread (`evaluate_batch(&mut self)`, kz-core/src/network/mod.rs:56),
 * created on that thread like `CudaNetwork::new` in `handle_new_graph` (kz-selfplay/src/server/executor.rs:320-342).
 * Engines of one model on one device share a single uploaded copy of the weights.
 */
#ifndef KZ_HIP_H
#define KZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KZ_DTYPE_F32 0 /* f32 storage, exact-f32 MFMA: the <=1e-4 parity path (the reference is f32 only, cudnn.rs:73) */
#define KZ_DTYPE_F16 1 /* f16 storage, f32 accumulate: the throughput path.  RANGE: activations are stored as f16, so a
                          residual stream beyond +-65504 overflows; the overflow is DETECTED, not saturated: the call that
                          returns the batch (kz_engine_eval_*, kz_engine_wait*, or kz_engine_synchronize after the
                          device-resident entry points) fails with a "non-finite activation" message and the caller
                          should evaluate that network with KZ_DTYPE_F32.  The same limit and the same check apply to
                          KZ_DTYPE_F32_SPLIT16 (its (hi, lo) pairs are f16 too) */
#define KZ_DTYPE_F32_SPLIT16 2 /* f32 tensors and the same <=1e-4 parity as KZ_DTYPE_F32, but the tower's products run on
                                  the f16 matrix cores: every activation and weight as a (hi, lo) f16 pair, three MFMAs per
                                  product, f32 accumulate.  One launch per batch for 192 / 256 tower channels on <= 64
                                  squares or 64 / 128 channels on <= 96 squares, with no more input planes than tower
                                  channels; one launch per layer otherwise (Go 19x19, wider towers: channels a multiple
                                  of 64, max_batch * squares * channels * 4 bytes < 2 GiB).  A tower whose channel count
                                  is no multiple of 64 (48, 96, 160 ... up to 512) runs widened to the next one by
                                  all-zero filters (same outputs).  kz_engine_create fails for the rest — a tower without
                                  blocks, more than 512 channels in no multiple of 64 — and kz_model_supports_dtype
                                  tells.  Everything outside the tower is the KZ_DTYPE_F32 path */

#define KZ_POLICY_ATAXX_CONV 0 /* AtaxxConvPolicyHead, python/lib/model/post_act.py:91-112 */
#define KZ_POLICY_CONV 1       /* ConvPolicyHead,      post_act.py:54-88 */
#define KZ_POLICY_ATTENTION 2  /* AttentionPolicyHead, post_act.py:115-141 */
#define KZ_POLICY_DENSE 3      /* DensePolicyHead,     post_act.py:26-51 */
#define KZ_POLICY_ARIMAA 4     /* ArimaaPolicyHead,    post_act.py:144-173 (the server's arimaa-split game, server.rs:174) */
#define KZ_POLICY_NONE 5       /* no PredictionHeads: a DenseNetwork (python/lib/model/simple.py:7-33), one Linear yields scalars and policy */

typedef struct kz_model kz_model;
typedef struct kz_engine kz_engine;

/* What `check_graph_shapes` compares against the mapper (rust/kz-core/src/network/common.rs:165-198):
 * input [BATCH, input_channels, board_h, board_w] with input_channels = scalar + bool planes (scalars first,
 * kz-core/src/mapping/mod.rs:40-63); outputs [BATCH, 5] and [BATCH, policy_len]. */
typedef struct kz_model_info {
    int32_t input_channels;
    int32_t board_h;
    int32_t board_w;
    int32_t input_scalar_channels; /* -1: unknown (ONNX loaded without the split) */
    int32_t input_bool_channels;   /* -1: unknown */
    int32_t policy_len;
    int32_t tower_depth;
    int32_t tower_channels;
    int32_t policy_kind;
    int32_t bits_bytes;     /* ceil(input_bool_channels * h * w / 8): BitBuffer storage per board (bit_buffer.rs:8-14); -1: unknown */
    int64_t param_count;
    double flops_per_eval;  /* direct-convolution FLOPs (2 per MAC), heads included: the roofline numerator */
} kz_model_info;

/* Thread-local message of the last failing call on this thread. */
const char *kz_last_error(void);

/* Replaces CudaDevice::all() (rust/kz-selfplay/src/server/server.rs:48-52). */
int kz_device_count(int *count);
/* PCI bus id of `device` as a NUL-terminated string ("0000:c1:00.0"): lets a launcher that starts one process per GPU
 * prove that its ranks sit on distinct devices (bench.py reports the set). */
int kz_device_pci_bus_id(int device, char *buf, size_t len);

/* ---- model: replaces load_graph_from_onnx_path + optimize_graph (server_alphazero.rs:126-128) ----
 * Accepts the KZMODEL1 container (kzero_amd/model_file.py); Conv+BN folding happens here. */
int kz_model_load(const char *path, kz_model **out);
int kz_model_load_memory(const void *blob, size_t len, kz_model **out);
/* The ONNX file the trainer already writes (python/lib/save_onnx.py:60-122: opset 10, input "input", outputs "scalars"
 * and "policy"), i.e. what `Command::NewNetwork(path)` carries (kz-selfplay/src/server/protocol.rs:36).  The graph does
 * not say how many of its input planes are broadcast scalars: pass the mapper's `input_scalar_count()`
 * (kz-core/src/mapping/mod.rs:21).  kz_model_load/_memory also accept ONNX, with the split unknown: such a model
 * serves kz_engine_eval_dense only and the packed entry points fail with a message. */
int kz_model_load_onnx(const char *path, int input_scalar_channels, kz_model **out);
int kz_model_load_onnx_memory(const void *blob, size_t len, int input_scalar_channels, kz_model **out);
void kz_model_free(kz_model *model);
int kz_model_get_info(const kz_model *model, kz_model_info *out);

/* ---- engine: replaces CudaNetwork::new(mapper, &graph, max_batch_size, device) (cudnn.rs:29-43) ---- */
int kz_engine_create(const kz_model *model, int device, int max_batch, int dtype, kz_engine **out);
void kz_engine_destroy(kz_engine *engine);
/* 1 when kz_engine_create would accept `dtype` for this model, 0 when not (KZ_DTYPE_F32_SPLIT16 has shape limits,
 * see above), negative on a null/unknown argument.  Lets a host pick "the fastest path with <= 1e-4 parity":
 * KZ_DTYPE_F32_SPLIT16 where supported, else KZ_DTYPE_F32. */
int kz_model_supports_dtype(const kz_model *model, int dtype);
/* What kz_engine_create(model, any device, max_batch, dtype) WOULD choose, without touching a GPU: the tower path (the
 * names kz_engine_tower_path documents) and the kernel launches one packed-input batch takes.  Pure host logic over the
 * kernels' support predicates — DESIGN.md §5.0 prints its path table from it and a CPU test holds it to a committed
 * copy.  Fails (non-zero, kz_last_error) exactly when kz_engine_create would refuse the dtype for this model. */
typedef struct kz_path_plan {
    char tower_path[48];
    int32_t launches_per_batch;
    int32_t reserved[3];
} kz_path_plan;
int kz_model_plan(const kz_model *model, int max_batch, int dtype, kz_path_plan *out);
int kz_engine_max_batch(const kz_engine *engine); /* Network::max_batch_size, network/mod.rs:53 */

/* ---- synchronous evaluation: replaces CudaNetwork::evaluate_batch's encode + executor.evaluate (cudnn.rs:55-82) ----
 * Only `batch` rows are computed and written (the reference NaN-pads to max_batch and discards, cudnn.rs:65,75-82).
 * batch must be in [0, max_batch]; batch == 0 is a no-op.  Caller owns all buffers; nothing is retained. */

/* Bit-compatible with CudaExecutor::evaluate: dense f32 NCHW [batch, C, H, W] as built by encode_input_full. */
int kz_engine_eval_dense(kz_engine *engine, const float *input_nchw, int batch, float *scalars_out /* [batch,5] */,
                         float *policy_out /* [batch,policy_len] */);

/* Packed input, the GPU does encode_input_full: `bits` = BitBuffer storage per board (LSB-first,
 * bit_buffer.rs:73-75), bits_stride bytes apart; `scalars_in` [batch, input_scalar_channels] as appended by
 * InputMapper::encode_input (mapping/mod.rs:37). */
int kz_engine_eval_packed(kz_engine *engine, const uint8_t *bits, size_t bits_stride, const float *scalars_in,
                          int batch, float *scalars_out, float *policy_out);

/* Packed input AND decoded output: decode_output (rust/kz-core/src/network/common.rs:16-100) runs on the GPU, so only
 * the decoded values and the probabilities of the available moves cross PCIe (~0.2 KB instead of 7.5 KB per chess eval).
 * move_offsets [batch+1] (CSR; move_offsets[0] == 0) and move_indices [move_offsets[batch]] list, per board and in
 * available_moves() order, `PolicyMapper::move_to_index(board, mv)` (kz-core/src/mapping/mod.rs:74) of every available
 * move; a finished board has an empty range (common.rs:77 `map_or(vec![], ..)`).
 * values_out [batch,5] = value (tanh), win, draw, loss (softmax), moves_left; probs_out parallel to move_indices.
 * Fails — where the reference asserts `sum > 0.0` (common.rs:110) — when a softmax sum is not strictly positive. */
int kz_engine_eval_packed_decoded(kz_engine *engine, const uint8_t *bits, size_t bits_stride, const float *scalars_in,
                                  int batch, const int64_t *move_offsets, const int32_t *move_indices,
                                  float *values_out, float *probs_out);

/* ---- asynchronous pair: several batches in flight per executor thread (replaces gpu_threads_per_device blocking
 * threads, rust/Readme.md:51).  slot in [0, KZ_ENGINE_SLOTS).  Inputs are copied to the slot's pinned staging before
 * submit returns; outputs are written to the caller's buffers by kz_engine_wait.  On the one-launch chess path the slots
 * alternate over two streams (a batch of 256 is half a chip of workgroups: two launches run side by side, the next launch
 * of a stream starts when the previous one ends) and the launch reads and writes the pinned staging directly, so no copy
 * operation sits between launches: keep all four slots submitted to keep both halves of the chip busy. */
#define KZ_ENGINE_SLOTS 4
int kz_engine_submit_packed(kz_engine *engine, int slot, const uint8_t *bits, size_t bits_stride,
                            const float *scalars_in, int batch);
int kz_engine_wait(kz_engine *engine, int slot, float *scalars_out, float *policy_out);
/* Same wait without the copy: *scalars_out [batch*5] and *policy_out [batch*policy_len] point into the slot's pinned
 * staging (library-owned) and stay valid until the next kz_engine_submit_packed on that slot or kz_engine_destroy —
 * the lifetime of the `&[DTensor]` the reference's executor hands out until its next call (cudnn.rs:73-82).  Saves a
 * 1.9 MB host copy per chess batch of 256 on the executor thread. */
int kz_engine_wait_view(kz_engine *engine, int slot, const float **scalars_out, const float **policy_out);
/* The asynchronous pair with decode_output on the device (kz_engine_eval_packed_decoded split in two): submit takes
 * the CSR move lists of the batch (move_offsets [batch+1], move_indices = move_to_index of every available move),
 * wait hands out views of the slot's pinned staging — values [batch*5] (value, win, draw, loss, moves_left) and the
 * probabilities parallel to move_indices — valid until the next submit on that slot.  0.2 KB instead of 7.5 KB per
 * chess evaluation cross PCIe and the executor thread does no softmax. */
int kz_engine_submit_packed_decoded(kz_engine *engine, int slot, const uint8_t *bits, size_t bits_stride,
                                    const float *scalars_in, int batch, const int64_t *move_offsets,
                                    const int32_t *move_indices);
int kz_engine_wait_decoded(kz_engine *engine, int slot, const float **values_out, const float **probs_out);

/* ---- device-resident evaluation (inputs and outputs already in HBM; used by bench.py and the parity tests) ----
 * Pointers are device pointers on the engine's device (kz_device_malloc).  Enqueues on the engine's stream and
 * returns; kz_engine_synchronize waits. */
int kz_engine_enqueue_packed_device(kz_engine *engine, const void *d_bits, size_t bits_stride,
                                    const void *d_scalars_in, int batch, void *d_scalars_out, void *d_policy_out);
int kz_engine_enqueue_dense_device(kz_engine *engine, const void *d_input_nchw, int batch, void *d_scalars_out,
                                   void *d_policy_out);
int kz_engine_synchronize(kz_engine *engine);

/* ---- device memory helpers (plain hipMalloc/hipMemcpy on `device`) ---- */
int kz_device_malloc(int device, size_t bytes, void **out);
int kz_device_free(int device, void *ptr);
int kz_memcpy_h2d(int device, void *dst, const void *src, size_t bytes);
int kz_memcpy_d2h(int device, void *dst, const void *src, size_t bytes);
int kz_device_synchronize(int device);

/* ---- measurement ----
 * With profiling on, every kernel launch of the forward pass is bracketed by HIP events on the engine's stream.
 * kz_engine_kernel_time sums the elapsed time of the launches whose kernel name starts with `prefix` since
 * profiling was last enabled (it synchronizes the stream first). */
int kz_engine_set_profiling(kz_engine *engine, int enable);
int kz_engine_kernel_time(kz_engine *engine, const char *prefix, double *total_ms, int64_t *launches);
/* Environment switches read by kz_engine_create.  This is the COMPLETE list for libkzhip.so (tests/test_abi.py compares
 * it with the strings in the built library); each selects between product paths that are parity-tested against the
 * oracle, none changes results beyond summation order, all are off by default:
 *   KZ_FORCE_GENERIC=1      no one-launch tower: one launch per layer ("board_conv_f16" / "conv_igemm_*"); an AttentionTower
 *                           network: the vector-ALU kernel ("attention_tower_f32_valu") instead of the matrix-core launch
 *   KZ_NO_BOARD_CONV=1      per-layer f16 convolutions through the implicit-GEMM kernel instead of the board-tile kernel
 *   KZ_NO_RESIDENT_F16G=1   no "tower_resident_f16g" launch (f16 shapes other than the chess network go per layer)
 *   KZ_NO_FUSED_HEADS=1     the "...+heads" launches without their heads: tower launch + separate head kernels
 *   KZ_TOWER_NB=1|2         boards per workgroup of the chess f16 launch (default 2; 1 = twice the workgroups: the better
 *                           choice for ONE engine at batch <= 256)
 *   KZ_KEEP_ACTIVATIONS=1   with KZ_FORCE_GENERIC=1: keep every layer's output for kz_engine_read_activation
 * The kernel organisations that were measured and rejected (four boards per workgroup, two Go boards per workgroup,
 * 32x32x16 tiles, hipGraph replay, ablation knobs) are NOT in this library nor in its source directory: `experiments/build.sh`
 * builds them (experiments/csrc/) into a separate experiments/libkzhip_exp.so for tests/test_gpu_experiments.py.
 *
 * Name of the path the engine chose; DESIGN.md 5.0 has the table (shape x arithmetic -> path, launches per batch, measured
 * rate), printed from kz_model_plan by tools/gen_path_table.py and held to the built library by tests/test_path_table.py.
 * One launch for the whole tower: "tower_resident_f16+heads" (chess attention
 * network — ChessStdMapper or ChessHistoryMapper input planes —, heads included), "tower_resident_f16",
 * "tower_resident_f16g" (other board-resident f16 shapes: 64 .. 512 channels on small boards),
 * "tower_resident_f16g+heads" (the same with the conv policy head and the scalar head inside: 128 / 256 channels),
 * "tower_resident_f32+heads" (exact f32, conv policy heads: decode, tower and heads in one launch), "tower_resident_f32"
 * (exact f32, other heads), "tower_resident_split16+heads" (KZ_DTYPE_F32_SPLIT16: the chess attention network at 256
 * channels and the conv-policy networks at 128 / 256 channels: encode, tower, scalar head and policy head in one launch),
 * "tower_resident_split16" (the other shapes of
 * KZ_DTYPE_F32_SPLIT16: tower launch + f32 head kernels).  One launch per layer:
 * "board_conv_f16" (whole boards as LDS tiles, Go-size boards), "board_conv_split16" (the same per-layer kernel in split
 * arithmetic: KZ_DTYPE_F32_SPLIT16 on boards the one-launch split tower cannot hold), "conv_igemm_f16", "conv_igemm_f32".
 * Networks whose tower is the reference's AttentionTower (python/lib/model/attention.py:8-45) instead of the ResTower — one
 * launch for the tower: "attention_tower_f16" / "attention_tower_f32" (8x8 boards, 8 heads of d_k = d_v = 16, d_model 128 / 256:
 * f16 and exact f32 on the matrix cores), "attention_tower_f32_valu" (every other shape: exact f32 on the vector ALUs; an f16
 * engine reads and writes f16 rows around it).  KZ_DTYPE_F32_SPLIT16 has no AttentionTower kernel: kz_model_supports_dtype = 0.
 * "dense_network_f32": a DenseNetwork (python/lib/model/simple.py), the whole network in one launch, f32 arithmetic. */
const char *kz_engine_tower_path(const kz_engine *engine);
/* How the dominant launch of that path covers the chip for a batch of `batch` boards: workgroups per launch and boards
 * per workgroup (per-layer paths: boards_per_workgroup = 0 when a workgroup holds a tile, not whole boards). */
int kz_engine_launch_geometry(const kz_engine *engine, int batch, int *workgroups, int *boards_per_workgroup);

/* ---- debugging / parity: copy an intermediate activation of the last evaluation to the host as f32 NCHW.
 * name: "tower.<i>" as in python/lib/model/post_act.py's nn.Sequential indices (0 = stem, 1..d = blocks,
 * d+1 = final BN): only available when the engine was created with the generic per-layer path (set KZ_FORCE_GENERIC=1
 * and KZ_KEEP_ACTIVATIONS=1 in the environment before kz_engine_create); or "tower.out", the tower's output (after the
 * final BN), on every path that writes it to memory (all but the "...+heads" paths; KZ_NO_FUSED_HEADS=1 in the
 * environment before kz_engine_create gives the separate head launches back). */
int kz_engine_read_activation(kz_engine *engine, const char *name, int batch, float *out_nchw);

#ifdef __cplusplus
}
#endif
#endif /* KZ_HIP_H */
