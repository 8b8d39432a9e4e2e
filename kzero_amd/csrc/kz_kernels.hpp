// kz_kernels.hpp — launch interface of the gfx950 kernels (definitions: kz_kernels.hip, kz_tower.hip).
// All activations are NHWC with the channel dimension padded to a multiple of 32 ("Cp"): row = one board
// square (pixel), columns = channels.  T is float (KZ_DTYPE_F32) or _Float16 (KZ_DTYPE_F16).
#pragma once
#include <vector>
#include <hip/hip_runtime.h>

#include <cstdint>

namespace kz {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// F0 — board encode (rust/kz-core/src/mapping/mod.rs:40-63) on the GPU.
// packed: bits [batch][bits_stride] u8 (BitBuffer layout) + scalars [batch][n_scalar] f32 -> x [batch*hw][ldx]
void launch_encode_packed(int dtype, const uint8_t *bits, size_t bits_stride, const float *scalars, int batch,
                          int n_scalar, int n_bool, int hw, void *x, int ldx, hipStream_t stream);
// dense: NCHW f32 [batch][c][hw] -> x [batch*hw][ldx] (channels >= c zero-filled)
void launch_encode_dense(int dtype, const float *nchw, int batch, int c, int hw, void *x, int ldx,
                         hipStream_t stream);

// Implicit-GEMM convolution, kernel k in {1,3}, pad k/2, fused epilogue:
//   v = acc + bias[oc];  if (relu) v = max(v, 0);  if (res) v += res[row][oc];
//   if (post_scale) v = v * post_scale[oc] + post_shift[oc];  y[row][oc] = v
// (conv A of a ResBlock: relu; conv B: relu then residual add — post_act.py:227-228 adds AFTER the ReLU;
//  the tower's final BatchNorm rides on the last conv as post_scale/post_shift.)
struct ConvArgs {
    const void *x;       // [rows_src][ldx]
    int ldx;
    const void *w;       // packed [k*k][cout_p][cin_p]
    const float *bias;   // [cout_p]
    const void *res;     // optional [M][ldres]
    int ldres;
    const float *post_scale, *post_shift;  // optional [cout_p]
    void *y;             // [M][ldy] in T, or nullptr
    float *y32;          // optional f32 output [M][ldy32], columns < cout only
    int ldy, ldy32;
    int M;               // output rows
    int h, w_;           // board dims (tap validity); rows of one board are contiguous: row = b*group + y*w + x
    int group;           // rows per board in M space (h*w)
    int src_group;       // rows per board in the source buffer
    int src_off;         // first source row of the group inside its board
    int cin_p, cout_p, cout;
    int k;
    int relu;
};
void launch_conv(int dtype, const ConvArgs &a, hipStream_t stream);
const char *conv_kernel_name(int dtype);
int conv_workgroups(int dtype, int M, int cout_p);

// ScalarHead (post_act.py:10-23) from the tower output x [batch*hw][ldx] -> scalars [batch][5] f32
struct ScalarHeadArgs {
    const void *x;
    int ldx, batch, hw, c, hc, hs;
    const float *w0, *b0;  // [hc][c], [hc]
    const float *w1, *b1;  // [hs][hc*hw] (channel-major flatten index c*hw + p), [hs]
    const float *w2, *b2;  // [5][hs], [5]
    float *out;
    // range check: when a pre-activation sum of the head's 1x1 convolution (it reads every value of the tower output)
    // is not finite, *nonfinite_flag = epoch (a plain store: the flag may live in pinned host memory; every batch in
    // flight has a flag of its own or, on the device-resident entry points, a larger epoch than what was checked last).
    // An f16 overflow anywhere in the residual stream persists to the tower output (x + relu(..) never removes an inf/NaN), so this is where every path checks it.
    int *nonfinite_flag = nullptr;
    int epoch = 0;
    const float *w1t = nullptr;  // optional: w1 transposed to [hc*hw][hs] (used when hs == 32: every input's 32 weights are
                                 // 128 contiguous bytes, so the first Linear reads coalesced and without dependent loops)
    // optional: ConvPolicyHead's extra moves (post_act.py:63-67,84-85: Conv1x1 C->1 on the same tower output, Flatten,
    // Linear hw -> extra) in the same pass over the tower output: w0x = [hc + 1][c], the scalar head's filters followed by
    // the extra-move filter (hc == 4 only); the logits go to policy[b * policy_len + policy_offset + j]
    int extra = 0;
    const float *w0x = nullptr, *pe_bc = nullptr, *pe_wl = nullptr, *pe_bl = nullptr;
    float *policy = nullptr;
    int policy_len = 0, policy_offset = 0;
    // the last Linear's width and the output's row stride: 5 and 5 for the ScalarHead; ArimaaPolicyHead's scalar branch
    // (post_act.py:155-162: the same Conv1x1 + ReLU, Flatten, Linear + ReLU, Linear chain with 1 + 6 outputs) runs through
    // the same kernel with n_out = 7 and out = the policy rows (out_ld = policy_len)
    int n_out = 5, out_ld = 5;
};
bool scalar_head_takes_extra(int dtype, int ldx, int hc);  // the launch below can carry ScalarHeadArgs::extra
void launch_scalar_head(int dtype, const ScalarHeadArgs &a, hipStream_t stream);

// Last 1x1 conv of the conv policy heads: y [batch*hw][ldy] (after conv1x1+ReLU) -> policy[b][oc*hw + p]
// (channel-major flatten, post_act.py:83,108) and optional trailing columns:
//   zero_tail columns of 0.0 (AtaxxConvPolicyHead's pass logit, post_act.py:106-110)
struct PolicyConvArgs {
    const void *y;
    int ldy, batch, hw, c, pc;
    const float *w, *b;  // [pc][c], [pc]
    float *policy;
    int policy_len, zero_tail;
};
void launch_policy_conv(int dtype, const PolicyConvArgs &a, hipStream_t stream);

// ConvPolicyHead.seq_extra (post_act.py:64-68): conv1x1 C->1, Flatten, Linear(hw -> extra) -> policy[b][offset + j]
struct PolicyExtraArgs {
    const void *x;
    int ldx, batch, hw, c, extra;
    const float *wc, *bc;  // [c], [1]
    const float *wl, *bl;  // [extra][hw], [extra]
    float *policy;
    int policy_len, offset;
};
void launch_policy_extra(int dtype, const PolicyExtraArgs &a, hipStream_t stream);

// AttentionPolicyHead tail (post_act.py:127-141): bulk [batch*64][ld_bulk] (2Q channels), under [batch*8][ld_under]
// (3Q channels) -> policy[b][k] = (q_from^T q_to)[flat_to_att[k]] / sqrt(Q)
struct AttentionArgs {
    const void *bulk, *under;
    int ld_bulk, ld_under, batch, q;
    const int32_t *flat_to_att;
    float *policy;
    int policy_len;
};
void launch_attention(int dtype, const AttentionArgs &a, hipStream_t stream);

// F7 — decode_output (rust/kz-core/src/network/common.rs:16-100) on the device: values [batch][5] = tanh / wdl softmax /
// moves_left; probs = per-board softmax over the logits at the available-move indices (CSR lists).
// error_flag: TWO words — [0] = 1: a softmax sum is not strictly positive (or a move index is out of range); [1] = 1:
// *nonfinite_flag == epoch (see ScalarHeadArgs).  The move lists, values, probs and error_flag may be pinned host memory
void launch_decode_output(const float *scalars, const float *logits, int batch, int policy_len,
                          const int64_t *move_offsets, const int32_t *move_indices, float *values, float *probs,
                          int *error_flag, const int *nonfinite_flag, int epoch, hipStream_t stream);

// The same decode as the LAST STEP OF A LAUNCH that has the heads inside (kz_decode_dev.hpp): with move_offsets set, a
// "...+heads" launch writes values [batch][5] and probs (parallel to move_indices) instead of the raw scalars and logits —
// no decode launch, and all five pointers may be the slot's pinned host staging (every word read or written once).
// error_flag: set to 1 where kz_decode_output raises bit 0; the range check keeps its own flag (ScalarHeadArgs).
struct DecodeArgs {
    const int64_t *move_offsets = nullptr;  // [batch + 1] CSR; nullptr: no decode
    const int32_t *move_indices = nullptr;
    float *values = nullptr, *probs = nullptr;
    int *error_flag = nullptr;
};

// ---- per-layer 3x3 convolution with the board as an LDS-resident spatial tile (kz_board_conv.hip): f16, cin and cout
// multiples of 64, h*w <= 384.  Same epilogue contract as ConvArgs. ----
struct BoardConvArgs {
    const void *x;   // [boards*h*w][ldx] f16
    int ldx;
    const void *weights;  // board_conv_pack_weights layout
    const float *bias, *post_scale, *post_shift;
    const void *res; // optional, [boards*h*w][ldy]
    void *y;
    int ldy, boards, h, w, cin, cout, relu;
    // (kz_board_conv2.hip, the experiment build's second organisation, only: its tile-row map and halo-row list; the
    // product kernel computes both from the thread id)
    const int *rowmap = nullptr;
    const unsigned short *halo = nullptr;
    int n_halo = 0;
    // launch_board_conv_split only: y32 != nullptr writes the result as f32 [boards*h*w][ldy32] instead of (hi, lo) halves
    float *y32 = nullptr;
    int ldy32 = 0;
};
bool board_conv_supported(int dtype, int h, int w, int cin, int cout);
int board_conv_workgroups(int boards, int h, int w, int cout);  // grid size: 64 output channels per workgroup
size_t board_conv_weight_elems(int cin, int cout);
void board_conv_pack_weights(const float *oihw, int cout, int cin, uint16_t *dst);
void launch_board_conv(const BoardConvArgs &a, hipStream_t stream);
// the same convolution in split arithmetic (f32-equivalent results on the f16 matrix cores, kz_tower_split.hip's): x, res
// and y are [boards*h*w][C / 32][hi 32 | lo 32] f16 rows (ldx = ldy = 2 C), cin == cout a multiple of 64
bool board_conv_split_supported(int h, int w, int cin, int cout);
size_t board_conv_split_weight_elems(int cin, int cout);
void board_conv_split_pack_weights(const float *oihw, int cout, int cin, uint16_t *dst);
void launch_board_conv_split(const BoardConvArgs &a, hipStream_t stream);
// f32 [rows][c] -> [rows][c / 32][hi 32 | lo 32] f16 (hi = f16(x), lo = f16(x - hi)): the stem's output enters the split layers
void launch_split_rows(const float *x, void *y, size_t rows, int c, hipStream_t stream);
// second organisation for boards of 193..384 squares (Go 19x19): two boards per workgroup, one workgroup per CU, staging
// under the MFMAs (kz_board_conv2.hip).  Same BoardConvArgs, with its own weight packing and tables.
bool board_conv2_supported(int dtype, int h, int w, int cin, int cout);
int board_conv2_workgroups(int boards, int cout);
void board_conv2_pack_weights(const float *oihw, int cout, int cin, uint16_t *dst);
void board_conv2_tables(int h, int w, std::vector<int> &rowmap, std::vector<unsigned short> &halo);
void launch_board_conv2(const BoardConvArgs &a, hipStream_t stream);

// ---- the AttentionTower network (python/lib/model/attention.py:8-136) in exact f32, one workgroup per board, one launch per batch
// (kz_att_tower.hip).  Reads the encoded planes, writes the tower output rows the head kernels read. ----
struct AttTowerArgs {
    const void *x0;       // encoded input [batch*h*w][ldx0], f32 or f16 (channels beyond c_in are zero)
    int ldx0, in_f16, c_in;
    const float *expand, *embedding;  // expand.weight [d_model][c_in], embedding [h*w][d_model]
    const float *layers;  // per encoder layer: project_qkv | project_out | ff.0 | ff.2 weights as the model stores them
    void *y;              // [batch*h*w][ldy], f32 or f16
    int ldy, out_f16;
    int batch, h, w, depth, d_model, heads, d_k, d_v, d_ff;
    float alpha, eps;
};
bool att_tower_supported(int h, int w, int c_in, int d_model, int heads, int d_k, int d_v, int d_ff, int depth);
size_t att_tower_layer_elems(int d_model, int heads, int d_k, int d_v, int d_ff);
void launch_att_tower(const AttTowerArgs &t, hipStream_t stream);

// ---- the same tower on the matrix cores (kz_att_tower_mfma.hip), in f16 or (f32 = true) exact f32: 8x8 boards, 8 heads of
// d_k = d_v = 16, the d_model / d_ff pairs att_tower16_supported names; rows of the arithmetic's element type in and out ----
struct AttTower16Args {
    bool f32 = false;
    const void *x0;        // encoded input [batch*64][cin_p]
    int cin_p;
    const void *w_expand;  // att_tower16_pack_expand
    const float *embedding;
    const void *w_layers;  // att_tower16_pack_layer, layer after layer
    void *y;               // [batch*64][d_model]
    // fused board encode (F0): packed boards straight into the launch (bits == nullptr: read x0)
    const uint8_t *bits = nullptr;
    size_t bits_stride = 0;
    const float *scalars_in = nullptr;
    int n_scalar = 0, n_bool = 0;
    int batch, depth, d_model, d_ff;
    float alpha, eps;
};
bool att_tower16_supported(int h, int w, int c_in, int d_model, int heads, int d_k, int d_v, int d_ff, int depth, bool f32);
int att_tower16_boards_per_workgroup(int d_model, int d_ff, int batch, bool f32);  // of a launch of `batch` boards
size_t att_tower16_expand_elems(int d_model, int cin_p);
size_t att_tower16_layer_elems(int d_model, int d_ff);
void att_tower16_pack_expand(const float *expand, int d_model, int c_in, int cin_p, bool f32, void *dst);
void att_tower16_pack_layer(const float *qkv, const float *out, const float *ff0, const float *ff1, int d_model, int d_ff, float alpha,
                            bool f32, void *dst);
void launch_att_tower16(const AttTower16Args &t, hipStream_t stream);

// ---- ScalarHead + AttentionPolicyHead of 8x8 boards in one launch, f16 (kz_att_heads.hip): for the f16 engines whose tower
// launch does not carry these heads itself ----
struct AttHeadsArgs {
    const void *x;         // tower output rows [batch*64][ldx] f16
    int ldx, batch, channels, q, hc, hs, policy_len;
    const void *weights;   // att_heads_pack
    const float *bias;
    const float *w1, *b1, *w2, *b2;  // the scalar head's Linear layers as the model stores them ([hs][hc*64], [5][hs])
    const int32_t *flat_to_att;
    float *scalars, *policy;
    int *nonfinite_flag = nullptr;   // the range check of ScalarHeadArgs
    int epoch = 0;
};
bool att_heads_supported(int dtype, int h, int w, int channels, int q, int hc, int hs, int policy_len);
size_t att_heads_weight_elems(int channels, int q);
void att_heads_pack(const float *w_bulk, const float *b_bulk, const float *w_under, const float *b_under, const float *w_sc, const float *b_sc,
                    int channels, int q, int hc, uint16_t *dst, float *bias);
void launch_att_heads(const AttHeadsArgs &t, hipStream_t stream);

// ---- DenseNetwork with its DenseBlocks (python/lib/model/simple.py:7-52) in one launch, f32 arithmetic (kz_dense_network.hip) ----
struct DenseNetArgs {
    const void *x0;        // encoded input rows [batch*hw][cin_p], f32 or f16
    int in_f16, batch, hw, cin_p, size, depth, res, policy_len;
    const float *w_in, *b_in;    // [size][hw*cin_p] (columns in the rows' order: square-major, channels padded), [size]
    const float *blocks;         // per block: sa | ta | wa [size][size] | ba | sb | tb | wb | bb
    const float *sf, *tf, *w_out, *b_out;  // final BatchNorm1d as an affine; [5 + policy_len][size], [5 + policy_len]
    float *scalars, *policy;
    int *nonfinite_flag;
    int epoch;
};
bool dense_network_supported(int h, int w, int c_in, int size, int depth, int policy_len);
size_t dense_network_block_elems(int size);
void launch_dense_network(const DenseNetArgs &a, hipStream_t stream);

// ---- board-resident tower in exact f32 (kz_tower_f32.hip): stem + 2*depth 3x3 convolutions in ONE launch ----
// Requirements: f32, channels 256 with h*w <= 64, or channels 128 with h*w <= 96; depth >= 1.
struct Tower32Args {
    const float *x0;      // encoded input [batch*hw][ldx0] f32 (channels beyond c_in are zero)
    int ldx0, c_in;
    const void *weights;  // tower32_pack_weights: stem, then the 2*depth tower convolutions
    const float *bias;    // [1 + 2*depth][channels]
    const float *post_scale, *post_shift;  // final BN [channels]
    float *y;             // tower output [batch*hw][ldy] f32
    int ldy, batch, h, w, channels, depth;
    // launch_tower_split / launch_tower_pairs only: fused board encode (F0) — packed boards straight into the launch
    // (bits == nullptr: read x0)
    const uint8_t *bits = nullptr;
    size_t bits_stride = 0;
    const float *scalars_in = nullptr;
    int n_scalar = 0, n_bool = 0;
    bool wide = false;    // launch_tower_pairs: twice the boards per workgroup (tower_split_wide_supported) — the engine's
                          // choice at max_batch; a launch whose own batch is too small for it takes the narrow tiles
    bool dense3 = false;  // launch_tower32, experiment build: three 7x7 boards per workgroup (tower32_dense3_supported)
    // launch_tower32 only: the conv policy head (Conv1x1 C->C + ReLU in the weight stream as one more centre-tap layer,
    // then Conv1x1 C->pc, post_act.py:75-110) and the scalar head (post_act.py:8-31) in the same launch — the tower
    // output never leaves LDS and `y` is not written.  tower32_heads_supported says for which models.
    struct Heads {
        bool on = false;
        // (the split launch's attention heads, kz_tower_split.hip: the scalar head's matrices as they are, the gather table)
        const float *sh_w0 = nullptr, *sh_w1 = nullptr;
        const int32_t *att_idx = nullptr;  // [1880]: (flat_to_att / 88) * 96 + flat_to_att % 88
        int hc = 0, hs = 0;  // scalar head: 1x1 conv channels, hidden size
        const float *small_w = nullptr;  // tower32_pack_small_weights: scalar-head conv (+ extra-move conv), policy conv
        const float *sh_b0 = nullptr, *sh_w1t = nullptr, *sh_b1 = nullptr, *sh_w2 = nullptr, *sh_b2 = nullptr;
        int pc = 0;          // policy: output channels of the second 1x1 conv
        const float *p_b1 = nullptr;
        int policy_len = 0, zero_tail = 0;
        int extra = 0;       // ConvPolicyHead's extra moves (post_act.py:86-96): Conv1x1 C->1 on the tower output, Linear hw->extra
        const float *pe_bc = nullptr, *pe_wl = nullptr, *pe_bl = nullptr;
        float *scalars = nullptr, *policy = nullptr;  // [batch][5], [batch][policy_len]
        int *nonfinite_flag = nullptr;                // range check, see ScalarHeadArgs
        int epoch = 0;
        DecodeArgs decode;  // set: decode_output inside the launch (the conv policy heads still write `policy`, which
                            // must then be device memory: the decode gathers from it)
    } heads;
};
bool tower32_supported(int dtype, int h, int w, int channels, int depth);
// (experiment build only; false in the product) Tower32Args::dense3 may be set for this network
bool tower32_dense3_supported(int policy_kind, int extra_moves, int pc, int h, int w, int channels, int hc, int hs, bool heads);
// conv / ataxx_conv policy head + a scalar head whose activations fit the launch's spare LDS
bool tower32_heads_supported(int policy_kind, int extra_moves, int pc, int h, int w, int channels, int hc, int hs);
// the same question for any launch that ends in kz_conv_heads.hpp with nt tiles of 16 pixel rows per workgroup
bool conv_heads_fit(int nt, int policy_kind, int extra_moves, int pc, int h, int w, int channels, int hc, int hs,
                    size_t scratch_bytes = 0);
// the plain-f16 launch keeps the tail's scratch behind its own images: this much
constexpr size_t F16_TAIL_SCRATCH_BYTES = 16 * 1024;
size_t tower32_small_weight_elems(int channels);
void tower32_pack_small_weights(const float *sh_w0, int hc, const float *pe_wc /* or null */, const float *p_w1, int pc,
                                int channels, float *dst);
size_t tower32_heads_weight_elems(int channels);           // the C->C 1x1 conv as channels/16 more steps
void tower32_pack_head_weights(const float *oi, int channels, float *dst);  // [C][C] 1x1 conv -> fragment order
int tower32_boards_per_workgroup(int h, int w, int channels);
size_t tower32_weight_elems(int c_in, int channels, int depth);
size_t tower32_weight_pad_elems(int channels);  // zeros the stream must end with (the launch's weight ring reads ahead)
void tower32_pack_weights(const float *oihw, int cout, int cin, bool stem, float *dst);
void launch_tower32(const Tower32Args &a, hipStream_t stream);

// ---- the same tower with f32-equivalent results on the f16 matrix cores (kz_tower_split.hip): activations and weights
// as (hi, lo) f16 pairs, three MFMAs per product.  Shapes of the exact-f32 launch with c_in <= channels; same Tower32Args
// (f32 in/out), `weights` = tower_split_pack_weights stream: 9 * ceil(c_in / 32) stem k-steps, then 9*C/32 per tower
// convolution ----
bool tower_split_supported(int h, int w, int channels, int depth, int c_in, bool split);
int tower_split_boards_per_workgroup(int h, int w, int channels, bool split, int wide_batch = 0);  // wide_batch: the launch's batch when the engine takes the wide tiles
// plain f16, 128 channels: twice the boards per workgroup (Tower32Args::wide) when max_batch still gives >= 128 workgroups
bool tower_split_wide_supported(int h, int w, int channels, int max_batch);
size_t tower_split_stem_elems(int channels, int c_in, bool split);              // f16 elements of the 9 * ceil(c_in / 32) stem k-steps
size_t tower_split_weight_elems(int channels, int depth, int c_in, bool split);  // f16 elements: stem + 2 * depth layers
void tower_split_pack_weights(const float *oihw, int cout, int cin, int hw, bool stem, bool split, uint16_t *dst);
void launch_tower_split(const Tower32Args &a, hipStream_t stream);
// the chess attention network's heads inside that launch (a.heads.on; scalars and policy are then its only output)
bool tower_split_heads_supported(int policy_kind, int query_channels, int policy_len, int h, int w, int channels, int sh_channels,
                                 int sh_size);
size_t tower_split_heads_weight_elems();  // f16 elements behind the tower's k-steps
void tower_split_pack_heads(const float *w_bulk, const float *b_bulk, const float *w_under, const float *b_under, uint16_t *dst,
                            float *bias5 /* [5][256] behind the tower's bias rows */);
// ... and for the conv policy heads (Ataxx, Go 9x9) at 128 / 256 channels: the policy head's Conv1x1 C->C rides behind the
// tower as one more pass (its bias as one more bias row), the rest is the exact-f32 launch's tail (Tower32Args::Heads with
// small_w set, as for launch_tower32) on f32 copies of the LDS images
// (split = false: the plain-f16 launch, launch_tower_pairs(t, false): "tower_resident_f16g+heads" — the same tail with its
// two small convolutions as f16 MFMAs on the f16 images: Heads::small_w = tower_split_pack_small_weights16; wide: with
// Tower32Args::wide)
bool tower_split_conv_heads_supported(int policy_kind, int extra_moves, int pc, int h, int w, int channels, int hc, int hs,
                                      bool split, int wide_batch = 0);
size_t tower_split_small_weight16_elems(int channels);  // f16 elements
void tower_split_pack_small_weights16(const float *sh_w0, int hc, const float *pe_wc /* or null */, const float *p_w1, int pc,
                                      int channels, uint16_t *dst);
size_t tower_split_conv_heads_weight_elems(int channels, bool split);
void tower_split_pack_conv_heads(const float *w /* [C][C] */, int channels, bool split, uint16_t *dst);
// the same launch without the lo halves (split = false): plain f16 arithmetic, x0 and y are f16 tensors behind the
// float pointers of Tower32Args — the board-resident f16 tower for the shapes kz_tower.hip does not take
void launch_tower_pairs(const Tower32Args &a, bool split, hipStream_t stream);

// ---- 1x1 convolution in the same split arithmetic (the head convolutions behind the split tower), f32 in and out:
// y[r][0..cout_p) = [relu](bias + W x[row(r)]), row(r) = (r / group) * src_group + src_off + r % group ----
struct Conv1x1SplitArgs {
    const void *x;        // f32 (split) or f16
    int ldx;
    const void *weights;  // conv1x1_split_pack_weights
    const float *bias;    // [cout_p]
    void *y;              // f32 (split) or f16
    int ldy, M, cin_p, cout_p, relu, group, src_group, src_off;
    bool split;           // false: plain f16 arithmetic and tensors (behind the one-launch f16 tower)
    // optional epilogue (conv1x1_policy_epilogue_supported; relu must be set, y is not written): the conv policy head's
    // second 1x1 convolution with ONE output channel — policy[(r / hw) * policy_len + r % hw] = pb1 + pw1 . relu(hidden[r])
    const float *pw1 = nullptr, *pb1 = nullptr;
    float *policy = nullptr;
    int policy_len = 0, hw = 0;
};
bool conv1x1_split_supported(int cin_p, int cout_p);
bool conv1x1_policy_epilogue_supported(int cin_p, int cout_p, int cout, int policy_channels);
size_t conv1x1_split_weight_elems(int cin_p, int cout_p, bool split);
void conv1x1_split_pack_weights(const float *w, int cout, int cin, int cout_p, int cin_p, bool split, uint16_t *dst);
void launch_conv1x1_split(const Conv1x1SplitArgs &a, hipStream_t stream);

// ---- board-resident tower (kz_tower.hip): the whole ResTower in ONE launch, activations never leave LDS ----
// Requirements: f16, 8x8, channels == 256 (cp), any depth >= 1, at most 224 input planes.
struct TowerArgs {
    const void *x0;       // encoded input [batch*hw][cin_p] f16 (used when bits == nullptr)
    int cin_p;            // input planes padded to a multiple of 32 (<= 224)
    // fused board encode: packed boards straight into the launch (bits == nullptr: read x0 instead)
    const uint8_t *bits;
    size_t bits_stride;
    const float *scalars_in;
    int n_scalar, n_bool;
    const void *w_stem;   // fragment-packed stem weights
    const void *w_tower;  // fragment-packed weights of the 2*depth 3x3 convs
    const float *bias;    // [1 + 2*depth][256]
    const float *post_scale, *post_shift;  // final BN [256]
    void *y;              // tower output [batch*hw][256] f16 (unused with fused heads)
    int batch, h, w, depth;
    // fused chess heads (ScalarHead + AttentionPolicyHead): the launch writes scalars/policy instead of y
    bool fused_heads;
    const float *sh_w0, *sh_b0, *sh_w1, *sh_b1, *sh_w2, *sh_b2;
    const int32_t *att_idx;  // [1880]: (flat_to_att / 88) * 96 + flat_to_att % 88
    float *scalars, *policy;
    int *nonfinite_flag;  // fused heads only: see ScalarHeadArgs
    int epoch;
    DecodeArgs decode;    // fused heads only; set: decode_output inside the launch, scalars / policy are not written
};
bool tower_resident_supported(int dtype, int h, int w, int channels, int depth, int c_in);
int tower_resident_boards_per_workgroup();  // 2 (1 with KZ_TOWER_NB=1)
bool tower_heads_supported(int policy_kind, int query_channels, int policy_len, int sh_channels, int sh_size);
size_t tower_packed_weight_elems(int cin_p, int depth);
size_t tower_heads_weight_elems();
// heads part of the weight stream (appended after the tower layers) and its 5 x 256 bias rows
void tower_pack_heads(const float *w_bulk, const float *b_bulk, const float *w_under, const float *b_under,
                      uint16_t *dst, float *bias5);
// host-side packing: OIHW f32 (BN folded) -> MFMA A-fragment order f16; dst index for conv layer l (0 = stem)
void tower_pack_weights(const float *oihw, int cout, int cin, int cin_p, uint16_t *dst);
void launch_tower_resident(const TowerArgs &a, hipStream_t stream);
// the same tower (no fused heads yet) with four boards per workgroup (kz_tower4.hip): single in-place LDS image, the
// residual stream in a private global scratch slab of tower4_scratch_bytes(max batch)
size_t tower4_scratch_bytes(int batch);
void launch_tower_resident4(const TowerArgs &a, void *xres, hipStream_t stream);

}  // namespace kz
