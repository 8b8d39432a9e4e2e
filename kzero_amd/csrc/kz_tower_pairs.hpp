// kz_tower_pairs.hpp — the device code of the one-launch tower on (hi, lo) f16 pairs and of its plain-f16 sibling: ONE template,
// `kz_tower_resident_split<C, NT, SPLIT, HEADS>`, instantiated by two translation units so that each kernel family has a source
// file of its own (bench.py hashes a kernel's sources to tell whether its committed HBM-traffic record still belongs to it):
//   kz_tower_split.hip  SPLIT = true   "tower_resident_split16[+heads]"  (KZ_DTYPE_F32_SPLIT16)
//   kz_tower_f16g.hip   SPLIT = false  "tower_resident_f16g[+heads]"      (KZ_DTYPE_F16 on the shapes kz_tower.hip does not take)
// Host-side weight packing and the support predicates: kz_tower_pairs_pack.hip; instance selection: kz_tower_pairs_shapes.hpp.
//
// every weight is carried as a pair of f16 values (hi = f16(v), lo = f16(v - hi): 22 significant bits) and every product
// is three MFMAs, hi*hi + hi*lo + lo*hi, accumulated in f32 (the lo*lo term is below 2^-22 of the product).  Same
// organisation as the resident launches (kz_tower.hip, kz_tower_f32.hip): the residual stream X and the mid activation
// Y live in LDS for the whole tower — as two images each, hi and lo — and the weights stream from L2 straight into MFMA
// A-fragment registers, 32 KB per k-step (hi fragments, then lo fragments).
//
// Why: the exact-f32 launch (kz_tower_f32.hip, v_mfma_f32_16x16x4_f32) is bound by the f32 MFMA rate, 157 TFLOP/s; three
// f16 MFMAs per product run at 2500 / 3 = 833 TFLOP/s.  The results agree with the CPU oracle within the same 1e-4 as
// the exact-f32 path (tests/test_gpu_parity.py), which the plain f16 path cannot (it rounds the residual stream to 11
// bits per layer).  Shapes of the exact-f32 launch (256 channels on <= 64 squares, 128 channels on <= 96); input and
// output are the f32 tensors of the f32 engine path (or packed boards in: the board encode, F0, is fused like in
// kz_tower.hip), the head kernels behind are the f32 engine's, with kz_conv1x1_split (below) for their 1x1 convolutions.
//
// The same template with SPLIT = false is the launch in plain f16 — one image per activation, one MFMA per product, f16
// tensors — i.e. the one-launch f16 tower for the shapes kz_tower.hip (chess: 8x8, 256 channels, fused heads) does not
// cover: "tower_resident_f16g".
//
// Arithmetic follows python/lib/model/post_act.py:201-239 with Conv+BN folded on the host (kz_model.cpp).
#pragma once
#include <cstdlib>
#include <vector>

#include "kz_kernels.hpp"

namespace kz {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

#include "kz_tower_pairs_shapes.hpp"  // HEAD_PASSES, POLICY, LOGIT_LD, split_tiles_for, split_wide_tiles_for

#include "kz_decode_dev.hpp"  // DecodeDev, decode_board_wave: decode_output as the last step of a launch with the heads inside
#include "kz_conv_heads.hpp"  // conv_heads_f32: the conv policy heads and the scalar head on f32 row images in LDS

constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100;

// Shapes: C tower channels (256 or 128), NT tiles of 16 pixel rows per workgroup = floor(16 NT / hw) whole boards packed
// densely (row r = board r / hw, pixel r % hw), as in kz_tower_f32.hip.  Chess: <256, 4> (one board); Ataxx 7x7 8x128:
// <128, 7> (two boards); Go 9x9 at 128 channels: <128, 6>.
// SPLIT = false is the same launch in plain f16 (one image per activation, one MFMA per product, f16 tensors in and out):
// the board-resident f16 tower for the shapes kz_tower.hip does not cover (128 channels; boards other than 8x8).
template <int C, int NT, bool SPLIT>
struct Geo {
    static constexpr int ROWS = NT * 16;
    // Bank conflicts of the fragment reads.  A ds_read_b128 is served in four groups of sixteen lanes, and a group mixes
    // eight rows of lane group kq = 0 (or 2) with the OTHER eight rows of kq = 1 (or 3): its sixteen 16-byte pieces fall
    // on sixteen different slots of the 256-byte bank row only if the pieces of kq and kq + 1 of the same row are a
    // multiple of 256 B apart (and rows advance by an odd number of slots).  With one row of 2 C bytes per pixel and the
    // second lane group at + C bytes that holds for C = 256 and 512 only: the 64 / 128 / 192 / 320 / 384-channel
    // instances of rounds 3-4 read with two-way conflicts on every fragment (SQ_LDS_BANK_CONFLICT = half of
    // SQ_LDS_IDX_ACTIVE on Go 9x9 16x128, `tools/pmc_workload.sh`).  For those an image is TWO planes — channels
    // [0, C/2) and [C/2, C) of every row, rows of C + 16 bytes, the planes a multiple of 256 B apart — and lane group kq
    // reads plane kq & 1 at (kq >> 1) * C/2 bytes: the same channel assignment {0, C/2, C/4, 3C/4}[kq] as before, so the
    // weight packing does not change.
    static constexpr bool TWO = C % 256 != 0;
    static constexpr int RS = TWO ? C + 16 : C * 2 + 16;  // LDS bytes per pixel row (of a plane): an odd number of 16-byte slots
    static constexpr int IMG = ROWS * RS;
    // hi block = [X][Y][16 all-zero rows] (TWO: once per plane), lo block = the same DELTA bytes later: one address array
    // serves both images of a pair (lo = hi + DELTA, zero rows included); DELTA and PLANE are multiples of 256 B so the
    // bank pattern is the same
    static constexpr int XH = 0, YH = IMG, ZH = 2 * IMG;
    static constexpr int PLANE = TWO ? (2 * IMG + 16 * RS + 255) / 256 * 256 : 0;
    static constexpr int DELTA = TWO ? 2 * PLANE : (2 * IMG + 16 * RS + 255) / 256 * 256;
    // byte offset, within a pixel row's address, of the channel at byte `cb` of the logical row (channel index * 2)
    __device__ static constexpr int chan_off(int cb) { return TWO ? (cb >= C ? PLANE + cb - C : cb) : cb; }
    static constexpr int PARTS = SPLIT ? 2 : 1;
    // (the stem input — rows of 64 B per chunk of 32 input planes — is staged in the Y image, which nothing else touches
    // before the first block's epilogue)
    static constexpr int LDS_BYTES = PARTS * DELTA;
    static constexpr int SH = PARTS * DELTA, SL = SH + ROWS * 64, LDS_BYTES_OWN_STEM = SH + PARTS * ROWS * 64;  // (the experiment build's 32x32x16 variant keeps its own stem rows)
    static constexpr int OT = C / 64;   // 16-channel output tiles per wave
    static constexpr int G = C / 32;    // k-steps per tap
    static constexpr int STEP = PARTS * 4 * OT * 64;  // uint4 per k-step: [hi | lo][wave 4][ot][lane 64]
    static_assert(C % 64 == 0 && LDS_BYTES <= 160 * 1024, "LDS budget");
    // Weight ring depth in k-steps.  A weight fragment is requested PF k-steps before its MFMAs, and an L2 round trip under
    // this load is ~2,000 cycles: a k-step of OT * NT MFMAs (16 cycles each, twice that wall time with two workgroups per
    // CU) must be shorter than latency / PF or every k-step waits for its weights.  At 256 channels and two boards (kz_tower.hip:
    // 32 MFMAs = 512 cycles, PF = 4) that holds; at 128 channels and one board a k-step is 8-12 MFMAs and four stages
    // cover 800 cycles — the round-4 counters of Go 9x9 16x128 (tools/pmc_workload.sh) show the matrix pipe busy 38 % of
    // the time AT FULL CLOCK, each k-step taking ~500 cycles = latency / 4.  So the ring is as deep as the registers allow:
    // the tap loop is unrolled U taps at a time (all nine for <= 128 channels, three otherwise) and PF divides U * G, so
    // that a k-step's stage is still a compile-time constant.  (Split arithmetic: three MFMAs per product, k-steps three
    // times as long, two register sets per stage: six stages; the chess network's fused attention heads run passes of
    // G = 8 k-steps, so its ring stays at four.)  Same-box A/Bs at 128 channels (Go 9x9 16x128 b=2048 / Ataxx 8x128 b=256,
    // evals/s): plain f16 with 4 / 9 / 12 / 18 stages 1.006M / 1.402M / 1.384M / 1.384M and 3.46M / 4.59M / 4.47M / 4.52M;
    // split with 4 / 6 / 9 / 12 stages 547k / 596k / 595k / 513k (12 stages spill) and 1.795M / 1.930M / 1.941M / 1.737M.  From 192 channels up a k-step is long enough for a ring that divides G
    // (3, 4 or 5 stages), and a deeper one was measured SLOWER there in same-box A/Bs (192: 663k -> 638k evals/s with nine
    // stages; 320: 262k -> 242k with six; 256 on Go 9x9: 330k -> 295k with eight; split 192: 285k -> 263k with six): those
    // launches are bound by the matrix cores' power, not by latency, and the extra registers and bytes in flight only cost.
    static constexpr int PF = G <= 4 ? (SPLIT ? 6 : G == 2 ? 18 : 9)
                                     : G % 4 == 0 ? 4 : G % 3 == 0 ? 3 : (G % 5 == 0 && NT < 6) ? 5 : 2;
    static constexpr int U = G % PF == 0 ? 1 : G <= 4 ? 9 : 3;  // (a ring that divides G keeps the one-tap loop body)
    static_assert((U * G) % PF == 0 && (9 * G) % PF == 0, "ring stage of a k-step must be a compile-time constant");
};

struct SplitDev {
    const void *x0;     // encoded input [batch*hw][ldx0]: f32 (SPLIT) or f16
    const uint4 *w;     // k-steps of [hi | lo][wave 4][ot][lane 64] x 16 B: 9 stem k-steps, then 2*depth*9*C/32
    const float *bias;  // [1 + 2*depth][C]
    const float *post_scale, *post_shift;
    void *y;            // tower output [batch*hw][ldy]: f32 (SPLIT) or f16
    int ldx0, ldy, batch, depth, h, w_, hw, nb;
    int stem_chunks;    // 32-channel chunks of the (padded) input planes: 9 * stem_chunks stem k-steps
    unsigned inv_w, inv_hw;  // ceil(65536 / w), ceil(65536 / hw): exact quotients for values < 512
    // fused encode (F0): packed boards; when bits == nullptr the stem input comes from x0
    const uint8_t *bits;
    size_t bits_stride;
    const float *scalars_in;
    int n_scalar, n_bool;
    // fused heads (HEADS: chess attention network, 256 channels = query channels): ScalarHead + AttentionPolicyHead on the
    // LDS-resident tower output.  The weight stream carries 5 more passes of 8 k-steps (conv_bulk[0:Q), conv_under as
    // three s-major passes, conv_bulk[Q:2Q)), the bias table 5 more rows in the same order.
    const float *sh_w0, *sh_b0, *sh_w1, *sh_b1, *sh_w2, *sh_b2;
    const int32_t *att_idx;  // [1880]: (flat_to_att / 88) * 96 + flat_to_att % 88
    // fused heads (HEADS == 2: conv policy heads — Ataxx, Go 9x9): the policy head's Conv1x1 C->C + ReLU as one more pass of
    // the weight stream (+ one bias row), then kz_conv_heads.hpp on f32 copies of the two images (its members:)
    int hc, hs, pc, policy_len, zero_tail, extra;
    const float *sh_w1t, *p_b1, *pe_bc, *pe_wl, *pe_bl;
    const f32x4 *small_w;
    const uint4 *small_w16;  // plain f16: the two small convolutions as f16 fragments (tower_split_pack_small_weights16)
    float *scalars, *policy;
    int *nonfinite_flag;
    int epoch;
    DecodeDev dec;  // dec.move_offsets set (fused heads only): decode_output inside the launch
};

__device__ __forceinline__ void split4(f32x4 v, h16x4 &hi, h16x4 &lo) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        hi[j] = (h16)v[j];
        lo[j] = (h16)(v[j] - (float)hi[j]);
    }
}

// HEADS: 0 = the tower alone; 1 = + the chess attention network's heads; 2 = + conv policy heads and the scalar head
template <int C, int NT, bool SPLIT, int HEADS = 0>
__global__ __launch_bounds__(256, 1) void kz_tower_resident_split(SplitDev a) {
    static_assert(HEADS != 1 || (C == 256 && NT == 4 && SPLIT), "attention heads: the chess network in split arithmetic");
    static_assert(HEADS != 2 || C == 256 || C == 128, "conv heads: a channel count of kz_tower_f32.hip");
    using L = Geo<C, NT, SPLIT>;
    constexpr int PARTS = L::PARTS, PF = L::PF;
    constexpr int RS = L::RS, OT = L::OT, G = L::G, DELTA = L::DELTA, XH = L::XH, YH = L::YH, ZH = L::ZH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int board0 = blockIdx.x * a.nb;
    const int boards = min(a.nb, a.batch - board0);
    const int rows_valid = boards * a.hw;
    const int layers = 2 * a.depth;
    // of the ring: the 9 * stem_chunks stem k-steps in front of them are read directly; the heads' passes follow the tower's
    const int total_ksteps = layers * 9 * G + (HEADS == 1 ? HEAD_PASSES * G : HEADS == 2 ? G : 0);
    const int bias_rows = layers + (HEADS == 1 ? HEAD_PASSES : HEADS == 2 ? 1 : 0);

    // ---- weight stream: prime PF stages (stage s = k-step g % PF) ----
    const uint4 *wp_stem = a.w + wave * OT * 64 + lane;
    const int sc = a.stem_chunks;
    const uint4 *wp = wp_stem + (size_t)9 * sc * L::STEP;
    auto wload = [&](int gk, int part, int ot) __attribute__((always_inline)) {
        return wp[(size_t)gk * L::STEP + part * (4 * OT * 64) + ot * 64];
    };
    uint4 wreg[PF][PARTS][OT];
#pragma unroll
    for (int s = 0; s < PF; s++)
#pragma unroll
        for (int part = 0; part < PARTS; part++)
#pragma unroll
            for (int ot = 0; ot < OT; ot++) wreg[s][part][ot] = wload(s < total_ksteps ? s : total_ksteps - 1, part, ot);
    int g = 0;
    auto ring_take = [&](int stage, h16x8 (&ah)[OT], h16x8 (&al)[OT]) __attribute__((always_inline)) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            ah[ot] = *reinterpret_cast<const h16x8 *>(&wreg[stage][0][ot]);
            if constexpr (SPLIT) al[ot] = *reinterpret_cast<const h16x8 *>(&wreg[stage][PARTS - 1][ot]);
        }
        const int gn = g + PF < total_ksteps ? g + PF : total_ksteps - 1;
#pragma unroll
        for (int part = 0; part < PARTS; part++)
#pragma unroll
            for (int ot = 0; ot < OT; ot++) wreg[stage][part][ot] = wload(gn, part, ot);
    };

    // ---- zero rows and the stem input (f32 -> hi/lo, 32 sc channels per square; rows beyond the batch are zero), staged
    // in the Y image: rows of 64 B per chunk of 32 input planes (ChessStdMapper 21 planes: one chunk; ChessHistoryMapper,
    // chess.rs:32-39, 34 / 47 / 60 planes: two) ----
    // (two planes: chunk k sits in the Y rows of plane k & 1, at (k >> 1) * 64 of a row of 64 ceil(sc / 2) bytes)
    constexpr int stem_h = YH, stem_l = YH + DELTA;
    const int srow = L::TWO ? 64 * ((sc + 1) >> 1) : 64 * sc, spieces = 8 * sc;
    auto stem_at = [&](int row, int chunk) __attribute__((always_inline)) {
        return L::TWO ? (chunk & 1) * L::PLANE + row * srow + (chunk >> 1) * 64 : row * srow + chunk * 64;
    };
    for (int id = tid; id < 16 * RS / 16; id += 256) {
#pragma unroll
        for (int plane = 0; plane < (L::TWO ? 2 : 1); plane++) {
            *reinterpret_cast<uint4 *>(lds + ZH + plane * L::PLANE + id * 16) = make_uint4(0, 0, 0, 0);
            if constexpr (SPLIT) *reinterpret_cast<uint4 *>(lds + ZH + plane * L::PLANE + DELTA + id * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    for (int id = tid; id < L::ROWS * spieces; id += 256) {  // (row, 4-channel piece)
        const int row = id / spieces, c4 = id - row * spieces;
        const bool have = row < rows_valid && c4 * 4 < a.ldx0;
        if (a.bits) {
            // encode_input_full (rust/kz-core/src/mapping/mod.rs:40-63) for 4 channels of one square: scalar planes first,
            // then the bool planes; bool i = bit i%8 of byte i/8 (bit_buffer.rs:73-75)
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < rows_valid) {
                const int b = (int)(((unsigned)row * a.inv_hw) >> 16), q = row - b * a.hw;
                const uint8_t *bb = a.bits + (size_t)(board0 + b) * a.bits_stride;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int ch = c4 * 4 + j;
                    if (ch < a.n_scalar) {
                        v[j] = a.scalars_in[(size_t)(board0 + b) * a.n_scalar + ch];
                    } else if (ch < a.n_scalar + a.n_bool) {
                        const unsigned bit = (unsigned)(ch - a.n_scalar) * a.hw + q;
                        v[j] = (float)((bb[bit >> 3] >> (bit & 7)) & 1);
                    }
                }
            }
            h16x4 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<h16x4 *>(lds + stem_h + stem_at(row, c4 >> 3) + (c4 & 7) * 8) = hi;
            if constexpr (SPLIT) *reinterpret_cast<h16x4 *>(lds + stem_l + stem_at(row, c4 >> 3) + (c4 & 7) * 8) = lo;
        } else if constexpr (SPLIT) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (have) v = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(a.x0) + ((size_t)board0 * a.hw + row) * a.ldx0 + c4 * 4);
            h16x4 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<h16x4 *>(lds + stem_h + stem_at(row, c4 >> 3) + (c4 & 7) * 8) = hi;
            *reinterpret_cast<h16x4 *>(lds + stem_l + stem_at(row, c4 >> 3) + (c4 & 7) * 8) = lo;
        } else {
            h16x4 v = h16x4{};
            if (have) v = *reinterpret_cast<const h16x4 *>(static_cast<const h16 *>(a.x0) + ((size_t)board0 * a.hw + row) * a.ldx0 + c4 * 4);
            *reinterpret_cast<h16x4 *>(lds + stem_h + stem_at(row, c4 >> 3) + (c4 & 7) * 8) = v;
        }
    }

    // Validity of (tile row, tap) as bitmasks: bit nt of okmask[tap] says that for this lane's row of tile nt the tap
    // lands on the same board
    unsigned okmask[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int r = nt * 16 + fr;
        const int b = (int)(((unsigned)r * a.inv_hw) >> 16), q = r - b * a.hw;
        const unsigned valid = r < rows_valid;
        const int yy = (int)(((unsigned)q * a.inv_w) >> 16), xx = q - yy * a.w_;
        const unsigned ym[3] = {(unsigned)(yy >= 1), 1u, (unsigned)(yy <= a.h - 2)};
        const unsigned xm[3] = {(unsigned)(xx >= 1), 1u, (unsigned)(xx <= a.w_ - 2)};
#pragma unroll
        for (int tap = 0; tap < 9; tap++) okmask[tap] |= (valid & ym[tap / 3] & xm[tap % 3]) << nt;
    }
    __syncthreads();

    f32x4 acc[OT][NT];
    f32x4 bias_next[OT];
    auto fetch_bias = [&](int row) __attribute__((always_inline)) {
        const int l = row <= bias_rows ? row : bias_rows;
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
            bias_next[ot] = *reinterpret_cast<const f32x4 *>(a.bias + l * C + (wave * OT + ot) * 16 + kq * 4);
    };
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[ot][nt] = bias_next[ot];
    };

    // three MFMAs per (output tile, pixel tile): hi*hi + hi*lo + lo*hi
    auto mfma3 = [&](const h16x8 (&ah)[OT], const h16x8 (&al)[OT], const h16x8 (&bh)[NT], const h16x8 (&bl)[NT])
                     __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int ot = 0; ot < OT; ot++) {
                if constexpr (SPLIT) {
                    acc[ot][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ot], bh[nt], acc[ot][nt], 0, 0, 0);
                    acc[ot][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ot], bl[nt], acc[ot][nt], 0, 0, 0);
                }
                acc[ot][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ot], bh[nt], acc[ot][nt], 0, 0, 0);
            }
    };
    auto ok_of = [&](int tap) __attribute__((always_inline)) {
        return tap == 0   ? okmask[0] : tap == 1 ? okmask[1] : tap == 2 ? okmask[2] : tap == 3 ? okmask[3]
               : tap == 4 ? okmask[4] : tap == 5 ? okmask[5] : tap == 6 ? okmask[6] : tap == 7 ? okmask[7] : okmask[8];
    };

    // ---- stem: 9 sc k-steps over the 32 sc (padded) input channels; conv + bias, no activation (post_act.py:205) ----
    fetch_bias(0);
    init_acc();
    fetch_bias(1);
#pragma nounroll
    for (int ks = 0; ks < 9 * sc; ks++) {
        const int tap = sc == 1 ? ks : ks / sc, chunk = ks - tap * sc;
        const int shift = (tap / 3 - 1) * a.w_ + (tap % 3 - 1);
        const unsigned ok = ok_of(tap);
        h16x8 ah[OT], al[OT], bh[NT], bl[NT];
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            const uint4 th = wp_stem[(size_t)ks * L::STEP + ot * 64];
            ah[ot] = *reinterpret_cast<const h16x8 *>(&th);
            if constexpr (SPLIT) {
                const uint4 tl = wp_stem[(size_t)ks * L::STEP + 4 * OT * 64 + ot * 64];
                al[ot] = *reinterpret_cast<const h16x8 *>(&tl);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int off = stem_at(nt * 16 + fr + shift, chunk) + kq * 16;  // stem: natural k (channel = 32 chunk + 8 kq + j)
            const bool valid = (ok >> nt) & 1;
            bh[nt] = valid ? *reinterpret_cast<const h16x8 *>(lds + stem_h + off) : h16x8{};
            if constexpr (SPLIT) bl[nt] = valid ? *reinterpret_cast<const h16x8 *>(lds + stem_l + off) : h16x8{};
        }
        mfma3(ah, al, bh, bl);
    }

    const int lane_row = fr * RS;
    const int epi_base = lane_row + ((wave * OT) * 16 + kq * 4) * 2;  // (one plane: C = 256, 512)
    // this lane's four channels of output tile ot, pixel row of tile nt: byte offset within an image
    auto epi_off = [&](int ot, int nt) __attribute__((always_inline)) {
        if constexpr (L::TWO) return lane_row + nt * 16 * RS + L::chan_off(((wave * OT + ot) * 16 + kq * 4) * 2);
        else return epi_base + nt * 16 * RS + ot * 32;
    };
    // epilogue: [relu]; [+ residual X]; -> (hi, lo) -> the image pair at dst_h
    auto epilogue = [&](int dst_h, bool relu, bool residual) __attribute__((always_inline)) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int off = epi_off(ot, nt);
                f32x4 v = acc[ot][nt];
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                }
                if (residual) {  // added in f32, AFTER the ReLU (post_act.py:227-228)
                    const h16x4 rh = *reinterpret_cast<const h16x4 *>(lds + XH + off);
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] += (float)rh[j];
                    if constexpr (SPLIT) {
                        const h16x4 rl = *reinterpret_cast<const h16x4 *>(lds + XH + DELTA + off);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] += (float)rl[j];
                    }
                }
                if constexpr (SPLIT) {
                    h16x4 hi, lo;
                    split4(v, hi, lo);
                    *reinterpret_cast<h16x4 *>(lds + dst_h + off) = hi;
                    *reinterpret_cast<h16x4 *>(lds + dst_h + DELTA + off) = lo;
                } else {
                    *reinterpret_cast<h16x4 *>(lds + dst_h + off) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                }
            }
    };
    epilogue(XH, false, false);
    __syncthreads();

    // ---- convolution passes over the LDS images.  Channel assignment of a k-step (as in kz_tower.hip): lane group kq
    // reads the 16-byte piece at kq_off + 16 ch of the row, i.e. channels 8 ch + {0, C/2, C/4, 3C/4}[kq] + j; the weights
    // are packed with the same assignment ----
    const int kq_off = L::TWO ? L::PLANE * (kq & 1) + (C / 2) * (kq >> 1) : C * (kq & 1) + (C / 2) * (kq >> 1);
    const int frag_base = lane_row + kq_off;
    // T[nt] = LDS address, in the hi block, of this lane's fragment row (pixel shifted by the tap) or of a zero row
    auto tap_rows = [&](int tap, int src_h, int (&T)[NT]) __attribute__((always_inline)) {
        const int shift = (tap / 3 - 1) * a.w_ + (tap % 3 - 1);
        const unsigned ok = ok_of(tap);
        const int shifted = src_h + frag_base + shift * RS;
        const int zrow = ZH + ((fr + shift) & 15) * RS + kq_off;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) T[nt] = ((ok >> nt) & 1) ? shifted + nt * 16 * RS : zrow;
    };
    auto conv_3x3 = [&](int src_h) __attribute__((always_inline)) {
        int T[NT], Tn[NT];
        h16x8 bh[2][NT], bl[2][NT];
        tap_rows(0, src_h, T);
        auto rd = [&](int t, int extra) __attribute__((always_inline)) { return *reinterpret_cast<const h16x8 *>(lds + t + extra); };
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            bh[0][nt] = rd(T[nt], 0);
            if constexpr (SPLIT) bl[0][nt] = rd(T[nt], DELTA);
        }
#pragma nounroll
        for (int tb = 0; tb < 9; tb += L::U)
#pragma unroll
        for (int tu = 0; tu < L::U; tu++) {
            const int tap = tb + tu;
            tap_rows(tap + 1 < 9 ? tap + 1 : tap, src_h, Tn);
#pragma unroll
            for (int ch = 0; ch < G; ch++) {
                const int stage = (tu * G + ch) % PF, cur = ch & 1, nxt = cur ^ 1;
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    bh[nxt][nt] = ch < G - 1 ? rd(T[nt], (ch + 1) * 16) : rd(Tn[nt], 0);
                    if constexpr (SPLIT) bl[nxt][nt] = ch < G - 1 ? rd(T[nt], DELTA + (ch + 1) * 16) : rd(Tn[nt], DELTA);
                }
                h16x8 ah[OT], al[OT];
                ring_take(stage, ah, al);
                mfma3(ah, al, bh[cur], bl[cur]);
                // every memory instruction in the shadow of an MFMA: the ring refills, the fragment reads, then the
                // remaining MFMAs back to back
                constexpr int NMF = (SPLIT ? 3 : 1) * OT * NT, NVM = PARTS * OT, NDS = PARTS * NT;
                constexpr int PAIRED = NVM + NDS < NMF ? NVM + NDS : NMF;  // memory instructions with an MFMA in front
#pragma unroll
                for (int i = 0; i < NVM; i++) {
                    if (i < PAIRED) __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < NDS; i++) {
                    if (NVM + i < PAIRED) __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 1, 0);
                }
                if constexpr (NMF > PAIRED) __builtin_amdgcn_sched_group_barrier(SG_MFMA, NMF - PAIRED, 0);
                __builtin_amdgcn_sched_barrier(0);
                g++;
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) T[nt] = Tn[nt];
        }
    };

    // ---- the 2*depth 3x3 convolutions ----
    for (int layer = 1; layer <= layers; layer++) {
        const bool is_b = (layer & 1) == 0;  // conv A: X -> Y; conv B: Y -> X (+ residual)
        init_acc();
        fetch_bias(layer + 1);
        conv_3x3(is_b ? YH : XH);
        if (!is_b) {
            epilogue(YH, true, false);
        } else if (layer != layers) {
            epilogue(XH, true, true);
        } else {
            // last layer: ReLU, residual, final BN -> f32 rows of the tower output in global memory
#pragma unroll
            for (int ot = 0; ot < OT; ot++) {
                const int oc = (wave * OT + ot) * 16 + kq * 4;
                const f32x4 ps = *reinterpret_cast<const f32x4 *>(a.post_scale + oc);
                const f32x4 pt = *reinterpret_cast<const f32x4 *>(a.post_shift + oc);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const int off = epi_off(ot, nt);
                    f32x4 v = acc[ot][nt];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                    const h16x4 rh = *reinterpret_cast<const h16x4 *>(lds + XH + off);
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] += (float)rh[j];
                    if constexpr (SPLIT) {
                        const h16x4 rl = *reinterpret_cast<const h16x4 *>(lds + XH + DELTA + off);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] += (float)rl[j];
                    }
                    v = v * ps + pt;
                    if constexpr (HEADS != 0) {  // the heads read the tower output from X (in place: this lane owns the slot)
                        if constexpr (SPLIT) {
                            h16x4 hi, lo;
                            split4(v, hi, lo);
                            *reinterpret_cast<h16x4 *>(lds + XH + off) = hi;
                            *reinterpret_cast<h16x4 *>(lds + XH + DELTA + off) = lo;
                        } else {
                            *reinterpret_cast<h16x4 *>(lds + XH + off) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                        }
                        continue;
                    }
                    const int r = nt * 16 + fr;
                    const size_t o = ((size_t)board0 * a.hw + r) * a.ldy + oc;
                    if (r < rows_valid) {
                        if constexpr (SPLIT) *reinterpret_cast<f32x4 *>(static_cast<float *>(a.y) + o) = v;
                        else *reinterpret_cast<h16x4 *>(static_cast<h16 *>(a.y) + o) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                    }
                }
            }
        }
        __syncthreads();
    }

    if constexpr (HEADS == 2) {
        // ---- conv policy heads (post_act.py:75-110) and scalar head (post_act.py:8-31) ----
        // The policy head's hidden layer, Conv1x1 C->C + ReLU, is one more pass of the weight stream over X (centre tap only,
        // three MFMAs per product) into Y.
        init_acc();
#pragma unroll
        for (int ch = 0; ch < G; ch++) {
            h16x8 ah[OT], al[OT], bh[NT], bl[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int t = XH + frag_base + nt * 16 * RS + ch * 16;
                bh[nt] = *reinterpret_cast<const h16x8 *>(lds + t);
                if constexpr (SPLIT) bl[nt] = *reinterpret_cast<const h16x8 *>(lds + t + DELTA);
            }
            ring_take(ch % PF, ah, al);
            mfma3(ah, al, bh, bl);
            g++;
        }
        epilogue(YH, true, false);
        __syncthreads();
        if constexpr (!SPLIT) {
            // Plain f16: the two small convolutions run as f16 MFMAs straight on the two f16 images (the tower's own fragment
            // reads: lane group kq's 16-byte piece of k-step g at kq_off + 16 g of the row; the weights packed to match),
            // row tiles split over the waves like in the f32 provider; the tail's scratch is F16_TAIL_SCRATCH_BYTES BEHIND the
            // launch's own LDS (the launcher asks for them), so no f32 copies of the images and any number of tiles.
            constexpr int TW = (NT + 3) / 4;
            auto small_conv = [&](int which, auto emit) {
                const int img = which ? YH : XH;
                const uint4 *wfrag = a.small_w16 + which * (G * 2 * 64);  // [G][2][64]
                f32x4 sa[2][TW];
                int base[TW];
#pragma unroll
                for (int t = 0; t < TW; t++) {
                    sa[0][t] = sa[1][t] = f32x4{0, 0, 0, 0};
                    const int row = (wave + 4 * t) * 16 + fr;
                    base[t] = img + (row < L::ROWS ? row : L::ROWS - 1) * RS + kq_off;
                }
#pragma unroll
                for (int gs = 0; gs < G; gs++) {
                    const uint4 w0 = wfrag[(gs * 2 + 0) * 64 + lane], w1 = wfrag[(gs * 2 + 1) * 64 + lane];
                    const h16x8 a0 = *reinterpret_cast<const h16x8 *>(&w0), a1 = *reinterpret_cast<const h16x8 *>(&w1);
#pragma unroll
                    for (int t = 0; t < TW; t++) {
                        const h16x8 b = *reinterpret_cast<const h16x8 *>(lds + base[t] + gs * 16);
                        sa[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b, sa[0][t], 0, 0, 0);
                        sa[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, sa[1][t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < TW; t++) {
                    const int row = (wave + 4 * t) * 16 + fr;
                    if (wave + 4 * t < NT && row < rows_valid) {
#pragma unroll
                        for (int mt = 0; mt < 2; mt++)
#pragma unroll
                            for (int q = 0; q < 4; q++) emit(mt, q, row, sa[mt][t][q]);
                    }
                }
            };
            conv_heads_tail<C, NT>(a, lds, L::LDS_BYTES, board0, boards, rows_valid, small_conv, XH, L::IMG);
        } else {
            // Both images, (hi, lo) — or plain f16 — -> f32 rows in the layout of kz_tower_f32.hip ([16 scratch rows][X][Y], row stride 4 C + 16),
            // through registers (everything else in LDS is dead); then the exact-f32 launch's own tail.
            constexpr int RS32 = C * 4 + 16, X32 = 16 * RS32, Y32 = X32 + L::ROWS * RS32;
            static_assert(Y32 + L::ROWS * RS32 <= 160 * 1024, "the f32 images fit a CU's LDS (the launcher asks for them)");
            constexpr int PIECES = L::ROWS * (C / 4), PER = (PIECES + 255) / 256;
            f32x4 vx[PER], vy[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int id = tid + k * 256;
                if (id < PIECES) {
                    const int r = id / (C / 4), p4 = id - r * (C / 4), off = r * RS + L::chan_off(p4 * 8);
                    const h16x4 xh = *reinterpret_cast<const h16x4 *>(lds + XH + off), yh = *reinterpret_cast<const h16x4 *>(lds + YH + off);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        vx[k][j] = (float)xh[j];
                        vy[k][j] = (float)yh[j];
                    }
                    if constexpr (SPLIT) {
                        const h16x4 xl = *reinterpret_cast<const h16x4 *>(lds + XH + DELTA + off), yl = *reinterpret_cast<const h16x4 *>(lds + YH + DELTA + off);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            vx[k][j] += (float)xl[j];
                            vy[k][j] += (float)yl[j];
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int id = tid + k * 256;
                if (id < PIECES) {
                    const int r = id / (C / 4), p4 = id - r * (C / 4);
                    *reinterpret_cast<f32x4 *>(lds + X32 + r * RS32 + p4 * 16) = vx[k];
                    *reinterpret_cast<f32x4 *>(lds + Y32 + r * RS32 + p4 * 16) = vy[k];
                }
            }
            __syncthreads();
            conv_heads_f32<C, NT>(a, lds, 0, X32, Y32, board0, boards, rows_valid);
        }
    }

    if constexpr (HEADS == 1) {
        // =====================================================================================================
        // Heads on the LDS-resident tower output X (hi, lo; already through the final BN), in the same split arithmetic:
        // every 1x1 convolution is one more pass of the weight stream (three MFMAs per product, f32 accumulators), every
        // intermediate is stored as a (hi, lo) pair, the attention product runs on the same three MFMAs, the scalar
        // head's small Linears in f32.  The zero rows and the stem input are dead: the under image of one pass lives in
        // the hi zero rows (UA), the scalar head's activations behind the lo zero rows (ACT).
        // =====================================================================================================
        constexpr int UA = ZH, UDELTA = 8 * RS;            // under block s: 8 rows (x) x 256 q, hi then lo
        static_assert(2 * UDELTA <= 16 * RS, "one under pass fits the hi zero rows");
        constexpr int ACT = ZH + DELTA, HID = ACT + 256 * 4;  // act [4*64] f32 (channel-major flatten), hid [32] f32
        static_assert(HID + 32 * 4 <= L::LDS_BYTES, "scalar head scratch");
        constexpr int LOG = YH;                            // logits [64][96] f32 once q_from is consumed
        static_assert(64 * LOGIT_LD * 4 <= 2 * L::IMG, "logits fit the hi block's X and Y");
        auto conv_1x1 = [&](int src_h, int tiles) __attribute__((always_inline)) {
            // centre tap only; `tiles` = 4: all 64 rows, 1: the tile of rows 48..63 (the far rank is rows 56..63)
#pragma unroll
            for (int ch = 0; ch < G; ch++) {
                h16x8 ah[OT], al[OT], bh[NT], bl[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const int t = src_h + frag_base + (tiles == 1 ? 3 : nt) * 16 * RS + ch * 16;
                    bh[nt] = *reinterpret_cast<const h16x8 *>(lds + t);
                    bl[nt] = *reinterpret_cast<const h16x8 *>(lds + t + DELTA);
                }
                ring_take(ch % PF, ah, al);
                if (tiles == 1) {
#pragma unroll
                    for (int ot = 0; ot < OT; ot++) {
                        acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ot], bh[0], acc[ot][0], 0, 0, 0);
                        acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ot], bl[0], acc[ot][0], 0, 0, 0);
                        acc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ot], bh[0], acc[ot][0], 0, 0, 0);
                    }
                } else {
                    mfma3(ah, al, bh, bl);
                }
                g++;
            }
        };

        // ---- H1: ScalarHead conv1x1 C->4 + ReLU (post_act.py:14-15), channel-major flatten (:16) -> act[c*64 + p];
        // the range check: a non-finite value anywhere in the residual stream persists to the tower output
        {
            float *act = reinterpret_cast<float *>(lds + ACT);
            const int c4 = tid >> 6, p = tid & 63;
            const unsigned char *row = lds + XH + p * RS;
            const float *w = a.sh_w0 + c4 * C;
            float sum = a.sh_b0[c4];
#pragma unroll 4
            for (int i = 0; i < C; i += 8) {
                const h16x8 xh = *reinterpret_cast<const h16x8 *>(row + i * 2), xl = *reinterpret_cast<const h16x8 *>(row + DELTA + i * 2);
                const f32x4 w0 = *reinterpret_cast<const f32x4 *>(w + i), w1 = *reinterpret_cast<const f32x4 *>(w + i + 4);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    sum += ((float)xh[j] + (float)xl[j]) * w0[j] + ((float)xh[4 + j] + (float)xl[4 + j]) * w1[j];
            }
            if (!(fabsf(sum) <= 3.0e38f) && a.nonfinite_flag)
                *reinterpret_cast<volatile int *>(a.nonfinite_flag) = a.epoch;  // (plain store: the flag may be in pinned host memory)
            act[tid] = fmaxf(sum, 0.0f);
        }

        // ---- H2: conv_bulk channels [0, Q) = q_from (post_act.py:127,131): X -> Y
        init_acc();
        fetch_bias(layers + 2);
        conv_1x1(XH, 4);
        epilogue(YH, false, false);
        __syncthreads();  // q_from and act are complete

        // ---- H3: conv_under on the 8 squares of rank index 7 (post_act.py:129) as 3 passes of 256 output channels: the
        // host permutes its output channels to s-major (oc' = 256 s + q for the original channel 3 q + s), so that
        // under.reshape(Q, 24)[q][8 s + x] (post_act.py:134) is row x of pass s.  Each pass goes straight into its 8
        // columns of the logits: L[i][64 + 8 s + x] = sum_q q_from[q][i] * under_s[x][q] — q_to rows as the A operand,
        // q_from rows as the B operand, this wave's 16 squares i
        f32x4 lu[3];
#pragma unroll
        for (int sp = 0; sp < 3; sp++) {
#pragma unroll
            for (int ot = 0; ot < OT; ot++) acc[ot][0] = bias_next[ot];
            fetch_bias(layers + 3 + sp);
            conv_1x1(XH, 1);
            if (sp > 0) __syncthreads();  // the previous pass's under image has been multiplied
            if (fr >= 8) {
#pragma unroll
                for (int ot = 0; ot < OT; ot++) {
                    h16x4 hi, lo;
                    split4(acc[ot][0], hi, lo);
                    const int off = UA + (fr - 8) * RS + ((wave * OT + ot) * 16 + kq * 4) * 2;
                    *reinterpret_cast<h16x4 *>(lds + off) = hi;
                    *reinterpret_cast<h16x4 *>(lds + off + UDELTA) = lo;
                }
            }
            __syncthreads();
            f32x4 l = f32x4{0.f, 0.f, 0.f, 0.f};
            const int ua = UA + (fr & 7) * RS + kq_off, qf = YH + (wave * 16 + fr) * RS + kq_off;
#pragma unroll
            for (int ch = 0; ch < G; ch++) {
                const h16x8 uh = *reinterpret_cast<const h16x8 *>(lds + ua + ch * 16), ul = *reinterpret_cast<const h16x8 *>(lds + ua + UDELTA + ch * 16);
                const h16x8 qh = *reinterpret_cast<const h16x8 *>(lds + qf + ch * 16), ql = *reinterpret_cast<const h16x8 *>(lds + qf + DELTA + ch * 16);
                l = __builtin_amdgcn_mfma_f32_16x16x32_f16(ul, qh, l, 0, 0, 0);
                l = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh, ql, l, 0, 0, 0);
                l = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh, qh, l, 0, 0, 0);
            }
            lu[sp] = l;  // lane (fr = i, kq < 2): columns 64 + 8 sp + 4 kq + 0..3
        }

        // ---- H4: conv_bulk channels [Q, 2Q) = the 64 board squares of q_to: X -> X in place (every wave reads all of X
        // in its k-loop, so the writes wait for a barrier)
        init_acc();
        conv_1x1(XH, 4);
        __syncthreads();
        epilogue(XH, false, false);

        // ---- H5: ScalarHead Linear(256 -> 32) + ReLU (post_act.py:17-18): 4 lanes per output, 64 inputs each
        {
            const float *act = reinterpret_cast<const float *>(lds + ACT);
            float *hid = reinterpret_cast<float *>(lds + HID);
            if (tid < 128) {
                const int j = tid >> 2, part = tid & 3;
                const float *w = a.sh_w1 + j * 256 + part * 64;
                const float *x = act + part * 64;
                float sum = 0.0f;
#pragma unroll 4
                for (int i = 0; i < 64; i += 4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + i), xv = *reinterpret_cast<const f32x4 *>(x + i);
                    sum += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
                }
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                if (part == 0) hid[j] = fmaxf(sum + a.sh_b1[j], 0.0f);
            }
        }
        __syncthreads();  // q_to (X) and hid are complete

        // ---- H6: the 64 x 64 board block of the logits (post_act.py:138): L[i][j] = sum_q q_from[q][i] * q_to[q][j]
        f32x4 la[4];
        {
            const int qf = YH + (wave * 16 + fr) * RS + kq_off;
#pragma unroll
            for (int jt = 0; jt < 4; jt++) la[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < G; ch++) {
                const h16x8 qh = *reinterpret_cast<const h16x8 *>(lds + qf + ch * 16), ql = *reinterpret_cast<const h16x8 *>(lds + qf + DELTA + ch * 16);
#pragma unroll
                for (int jt = 0; jt < 4; jt++) {
                    const int tj = XH + (jt * 16 + fr) * RS + kq_off + ch * 16;
                    const h16x8 th = *reinterpret_cast<const h16x8 *>(lds + tj), tl = *reinterpret_cast<const h16x8 *>(lds + tj + DELTA);
                    la[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl, qh, la[jt], 0, 0, 0);
                    la[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th, ql, la[jt], 0, 0, 0);
                    la[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(th, qh, la[jt], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // every wave is done reading q_from (Y) and q_to (X)
        {
            const float inv = 1.0f / sqrtf((float)C);  // / sqrt(query_channels) (post_act.py:138)
            float *lg = reinterpret_cast<float *>(lds + LOG) + (wave * 16 + fr) * LOGIT_LD;
#pragma unroll
            for (int jt = 0; jt < 4; jt++) *reinterpret_cast<f32x4 *>(lg + jt * 16 + kq * 4) = la[jt] * inv;
            if (kq < 2) {
#pragma unroll
                for (int sp = 0; sp < 3; sp++) *reinterpret_cast<f32x4 *>(lg + 64 + 8 * sp + kq * 4) = lu[sp] * inv;
            }
        }
        // ---- H7: ScalarHead Linear(32 -> 5) (post_act.py:19)
        const bool decode = a.dec.move_offsets != nullptr;
        float *raw = reinterpret_cast<float *>(lds + ACT);  // (act is dead since H5: the decode's five scalars)
        if (tid < 5) {
            const float *hid = reinterpret_cast<const float *>(lds + HID);
            float sum = a.sh_b2[tid];
            for (int i = 0; i < 32; i++) sum += a.sh_w2[tid * 32 + i] * hid[i];
            if (decode) raw[tid] = sum;
            else a.scalars[(size_t)board0 * 5 + tid] = sum;
        }
        __syncthreads();
        if (decode) {
            // ---- H8': decode_output (common.rs:16-100) on the LDS-resident logits: the gather of post_act.py:140 per
            // available move, softmax, tanh / wdl — one wave; q_to's hi image (X) is dead and holds its staging
            if (wave == 0) {
                const float *lg = reinterpret_cast<const float *>(lds + LOG);
                decode_board_wave(a.dec, board0, lane, raw, reinterpret_cast<float *>(lds + XH), L::IMG / 4,
                                  [&](int idx) { return lg[a.att_idx[idx]]; });
            }
        } else {
            // ---- H8: policy.flatten(1)[:, FLAT_TO_ATT] (post_act.py:140): coalesced 1880-float rows
            const float *lg = reinterpret_cast<const float *>(lds + LOG);
            float *pol = a.policy + (size_t)board0 * POLICY;
            for (int k = tid; k < POLICY; k += 256) pol[k] = lg[a.att_idx[k]];
        }
    }
}

#ifdef KZ_EXPERIMENTS
#include "kz_tower_pairs_exp32.hpp"
#endif

template <int C, int NT, bool SPLIT, int HEADS = 0>
void launch(const SplitDev &d, int grid, hipStream_t stream) {
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    // (the conv heads' tail: the split launch wants the two images as f32 rows; the plain-f16 launch the tail's scratch,
    // F16_TAIL_SCRATCH_BYTES, behind its own images)
    constexpr int F32_IMAGES = (16 + 2 * NT * 16) * (C * 4 + 16), OWN = Geo<C, NT, SPLIT>::LDS_BYTES;
    static_assert(OWN == pairs_own_lds_bytes(C, NT, SPLIT), "the host's restatement of the LDS geometry");
    constexpr int LDS = HEADS != 2 ? OWN : !SPLIT ? OWN + pairs_f16_tail_scratch_bytes(C, NT) : F32_IMAGES > OWN ? F32_IMAGES : OWN;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_tower_resident_split<C, NT, SPLIT, HEADS>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        done_mask |= 1ull << (dev & 63);
    }
    kz_tower_resident_split<C, NT, SPLIT, HEADS><<<grid, 256, LDS, stream>>>(d);
}


// The launch arguments of a Tower32Args: what both translation units fill in before they pick an instance.
// nt: tiles of 16 pixel rows per workgroup of the instance to launch; grid: workgroups.
inline SplitDev make_split_dev(const Tower32Args &t, bool split, int &nt, int &grid) {
    SplitDev d{};
    d.bits = t.bits;
    d.bits_stride = t.bits_stride;
    d.scalars_in = t.scalars_in;
    d.n_scalar = t.n_scalar;
    d.n_bool = t.n_bool;
    d.x0 = t.x0;  // (f16 tensors behind the same pointers when !split)
    d.ldx0 = t.ldx0;
    d.w = static_cast<const uint4 *>(t.weights);
    d.bias = t.bias;
    d.post_scale = t.post_scale;
    d.post_shift = t.post_shift;
    d.y = t.y;
    d.ldy = t.ldy;
    d.batch = t.batch;
    d.depth = t.depth;
    d.h = t.h;
    d.w_ = t.w;
    d.hw = t.h * t.w;
    d.stem_chunks = (t.c_in + 31) / 32;
    // (an engine that takes the wide tiles still launches the narrow ones for a batch too small to fill 128 wide workgroups:
    // the same weight stream, twice the workgroups, half the time per workgroup)
    nt = !split && t.wide ? split_wide_tiles_for(d.hw, t.channels, t.batch) : 0;  // (the widest level this batch fills the chip with)
    if (!nt) nt = split_tiles_for(d.hw, t.channels, split);
    d.nb = nt * 16 / d.hw;
    d.inv_w = (65536u + (unsigned)t.w - 1) / (unsigned)t.w;
    d.inv_hw = (65536u + (unsigned)d.hw - 1) / (unsigned)d.hw;
    grid = (t.batch + d.nb - 1) / d.nb;
    if (t.heads.on && (split || t.heads.small_w)) {  // (the engine asked tower_split_[conv_]heads_supported)
        const Tower32Args::Heads &hd = t.heads;
        d.sh_w0 = hd.sh_w0; d.sh_b0 = hd.sh_b0; d.sh_w1 = hd.sh_w1; d.sh_b1 = hd.sh_b1; d.sh_w2 = hd.sh_w2; d.sh_b2 = hd.sh_b2;
        d.att_idx = hd.att_idx;
        d.scalars = hd.scalars; d.policy = hd.policy;
        d.nonfinite_flag = hd.nonfinite_flag; d.epoch = hd.epoch;
        d.dec = DecodeDev{hd.decode.move_offsets, hd.decode.move_indices, hd.decode.values, hd.decode.probs, hd.decode.error_flag,
                          hd.small_w ? hd.policy_len : POLICY};
        if (hd.small_w) {  // conv policy heads (tower_split_conv_heads_supported)
            d.hc = hd.hc; d.hs = hd.hs; d.pc = hd.pc; d.policy_len = hd.policy_len; d.zero_tail = hd.zero_tail; d.extra = hd.extra;
            d.sh_w1t = hd.sh_w1t; d.p_b1 = hd.p_b1; d.pe_bc = hd.pe_bc; d.pe_wl = hd.pe_wl; d.pe_bl = hd.pe_bl;
            d.small_w = reinterpret_cast<const f32x4 *>(hd.small_w);
            d.small_w16 = reinterpret_cast<const uint4 *>(hd.small_w);  // (one pointer: f32 fragments for the split launch, f16 for the plain one)
        }
    }
    return d;
}

}  // namespace

}  // namespace kz
