// kz_tower.hip — board-resident ResTower (+ fused chess heads): the whole tower (stem + 2*depth fused 3x3
// convolutions + final BN) and, for the attention-policy chess network, the ScalarHead and AttentionPolicyHead, in ONE
// launch.  Replaces the ~45 cuDNN/NVRTC launches of the reference's GPU path (docs/conv_bn_sm_flow.svg; SURVEY.md
// §2.2 F1-F6) for 8x8 boards with 256 channels in f16.
//
// Why this shape (MI355X-first):
//  * Boards are independent and an 8x8x256 f16 board is 32 KB, so a workgroup keeps NB boards' residual stream (X) and
//    mid activation (Y) in LDS for the whole tower: activations never touch HBM between layers, there is no launch
//    boundary between layers and no inter-workgroup communication at all.
//  * The only stream is the weights (1.18 MB per layer).  They are read from L2 straight into MFMA A-fragment
//    registers: the host packs them in fragment order, so a wave-instruction is one coalesced 1 KiB global_load_dwordx4
//    and weights never pass through LDS.  Each wave owns 64 output channels (no weight byte is loaded twice per CU).
//    Loads run PF k-steps ahead of the MFMAs, across layer boundaries and into the heads.
//  * GEMM orientation D[oc][pixel] = W[oc][k] * X[k][pixel]: weights are the MFMA A operand, activations the B
//    operand, so a lane ends up with 4 consecutive channels of one pixel and writes them to the NHWC LDS image with one
//    ds_write_b64.  im2col exists only as LDS addressing: a tap outside the board reads an all-zero row.
//  * LDS image: row = pixel, 512 B of channels + 16 B pad (row stride 528 B), and a k-step's channel assignment is
//    permuted so that every ds_read_b128 fragment read is bank-conflict-free (see frag_base below).
//  * Two boards per workgroup: an MFMA pixel tile is line y of board 0 + line y of board 1, so the taps that fall off
//    the top / bottom edge hit a WHOLE tile (dy = -1: tile 0, dy = +1: tile 7) and those MFMAs are not issued: 8.3 % of
//    a direct convolution's multiply-adds are multiplications by the zero padding.
//  * Heads (post_act.py:10-23, 115-141) run on the LDS-resident tower output: conv_under and conv_bulk are more passes
//    of the same weight stream, q_from^T q_to is MFMA on LDS operands, the 1880-entry gather and the scalar head write
//    the only HBM output of the launch (7.5 KB per board).
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kz_kernels.hpp"

namespace kz {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

#include "kz_decode_dev.hpp"  // DecodeDev, decode_board_wave: decode_output as the last step of the launch

constexpr int C = 256;          // tower channels (= attention query channels in the fused-heads variant)
constexpr int RS = C * 2 + 16;  // LDS bytes per pixel row: 512 B of channels + 16 B pad, so that the 16 rows of a
                                // fragment read fall on 16 different 16-byte slots of the 256-byte bank row
constexpr int URS = 3 * C * 2 + 16;  // row stride of the conv_under image (768 channels)
constexpr int KSTEPS = 72;      // 9 taps x 8 chunks of 32 channels
constexpr int HEAD_KSTEPS = 5 * 8;  // conv_under as 3 passes of 256 channels, conv_bulk as 2: 8 k-steps each
constexpr int POLICY = 1880, LOGIT_LD = 96;  // 64 x 88 attention logits, rows padded to 96

struct TowerDev {
    const h16 *x0;
    const uint4 *w_stem, *w_tower;
    const float *bias, *post_scale, *post_shift;
    h16 *y;
    int cin_p, batch, depth;
    // fused encode (F0): packed boards; when bits == nullptr the stem input comes from x0
    const uint8_t *bits;
    size_t bits_stride;
    const float *scalars_in;
    int n_scalar, n_bool;
    // fused heads
    const float *sh_w0, *sh_b0, *sh_w1, *sh_b1, *sh_w2, *sh_b2;
    const int32_t *att_idx;  // [1880]: (flat_to_att / 88) * 96 + flat_to_att % 88
    float *scalars, *policy;
    int *nonfinite_flag;  // range check (fused heads), see ScalarHeadArgs in kz_kernels.hpp
    int epoch;
    DecodeDev dec;        // dec.move_offsets set: values / probabilities of the available moves instead of scalars / policy
};

template <int NB>
struct Layout {
    static constexpr int M = NB * 64;
    static constexpr int MT = M / 16;
    // Tile -> pixel rows.  NB == 1: tile mt = board lines 2mt, 2mt+1.  NB == 2: tile mt = line mt of board 0 (lanes
    // fr < 8) and line mt of board 1 (fr >= 8): a tap with dy = -1 then reads ONLY padding in tile 0 and dy = +1 only
    // padding in tile 7, and those MFMAs are simply not issued — 6 of the 72 (tap, tile) pairs, 8.3 % of the direct
    // convolution's multiply-adds are multiplications by the zero padding.
    static constexpr bool LINE_TILES = NB == 2;
    static constexpr int BPAD = LINE_TILES ? 128 : 0;   // board 1's image starts 8 sixteen-byte slots later, so that the
                                                        // 16 rows of a line tile still fall on 16 different slots
    static constexpr int BOARD = 64 * RS + BPAD;        // bytes from a board's image to the next one's
    static constexpr int TS = LINE_TILES ? 8 * RS : 16 * RS;  // bytes from a tile's rows to the next tile's
    static constexpr int IMG = NB * BOARD;
    static constexpr int X_OFF = 0;
    static constexpr int Y_OFF = IMG;
    static constexpr int Z_OFF = 2 * IMG;           // 16 all-zero rows (what a tap outside the board reads)
    static_assert(Z_OFF % 256 == 0, "zero rows keep the slot pattern of the image rows");
    static constexpr int S_OFF = Z_OFF + 16 * RS;   // stem input, rows of 64 B (32 channels)
    static constexpr int TOWER_BYTES = S_OFF + M * 64;
    // heads phase (the zero rows and the stem input are dead by then)
    static constexpr int U_OFF = Z_OFF;             // conv_under image: 16 rows x URS
    static constexpr int ACT_OFF = U_OFF + 16 * URS;  // scalar head: act [NB][256] f32, hid [NB][32] f32
    static constexpr int HID_OFF = ACT_OFF + NB * 256 * 4;
    static constexpr int HEADS_BYTES = HID_OFF + NB * 32 * 4;
    static constexpr int LOG_OFF = Y_OFF;           // attention logits [NB][64][96] f32 (after q_from is consumed)
    static constexpr int BYTES = TOWER_BYTES > HEADS_BYTES ? TOWER_BYTES : HEADS_BYTES;
    static_assert(NB * 64 * LOGIT_LD * 4 <= IMG, "logits fit the Y region");
    // byte offset of pixel row p of board b within an image
    static constexpr int row_off(int b, int p) { return b * BOARD + p * RS; }
    static_assert(BYTES <= 160 * 1024, "LDS budget");
};

// LLVM SchedGroupMask bits for __builtin_amdgcn_sched_group_barrier
constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100;

// PF = weight prefetch distance in k-steps (register ring stages, a power of two <= 8); PF * 16 KB are in flight per
// CU.  Measured at one board per workgroup: PF = 8 is SLOWER than PF = 4 (0.665 vs 0.611 ms per batch) — that
// configuration is bound by L2->CU bandwidth (~80 GB/s per CU), not by latency.
// WIDE: more than 32 input planes (the instance for up to 32 keeps its compile-time stem: a same-box A/B of a run-time
// chunk count in the benchmark's instance cost 0.3 %)
template <int NB, bool HEADS, int PF, bool WIDE = false>
__global__ __launch_bounds__(256, 1) void kz_tower_resident(TowerDev a) {
    using L = Layout<NB>;
    constexpr int M = L::M, MT = L::MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int board0 = blockIdx.x * NB;
    const int layers = 2 * a.depth;
    const int total_ksteps = layers * KSTEPS + (HEADS ? HEAD_KSTEPS : 0);
    const int bias_rows = layers + (HEADS ? 5 : 0);  // last valid row of the bias table

    // ---- weight stream: per k-step 16 KB = [wave 4][nt 4][lane 64] x 16 B; prime PF stages before anything else ----
    const uint4 *wp = a.w_tower + wave * 256 + lane;
    auto wload = [&](int gk, int nt) __attribute__((always_inline)) { return wp[(size_t)gk * 1024 + nt * 64]; };
    uint4 wreg[PF][4];
#pragma unroll
    for (int s = 0; s < PF; s++) {
        const int g = s < total_ksteps ? s : total_ksteps - 1;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wreg[s][nt] = wload(g, nt);
    }
    int g = 0;  // global k-step index into the weight stream
    // take this k-step's fragments out of ring stage `stage` and refill the stage with k-step g + PF (clamped at the
    // end of the stream; the surplus loads are never used)
    auto ring_take = [&](int stage, h16x8 (&af)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) af[nt] = *reinterpret_cast<const h16x8 *>(&wreg[stage][nt]);
#ifndef KZ_TW_NO_WLOAD  // (timing experiments: a build without the weight stream)
        const int gn = g + PF < total_ksteps ? g + PF : total_ksteps - 1;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wreg[stage][nt] = wload(gn, nt);
#endif
    };

    // ---- zero rows and stem input.  Up to 32 input planes the stem input has its own rows of 64 B behind the zero rows;
    // more (ChessHistoryMapper, chess.rs:32-39: 34 / 47 / 60 planes; sc chunks of 32) are staged in the Y image, which
    // nothing touches before the first block's epilogue ----
    const int sc = WIDE ? a.cin_p >> 5 : 1;
    const int stem_off = WIDE ? L::Y_OFF : L::S_OFF, srow = 64 * sc, spieces = 4 * sc;
    for (int id = tid; id < 16 * RS / 16; id += 256)
        *reinterpret_cast<uint4 *>(lds + L::Z_OFF + id * 16) = make_uint4(0, 0, 0, 0);
    for (int id = tid; id < M * spieces; id += 256) {
        const int row = WIDE ? id / spieces : id >> 2, c = id - row * spieces;
        const int board = board0 + (row >> 6);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (board < a.batch) {
            if (a.bits) {
                // encode_input_full (rust/kz-core/src/mapping/mod.rs:40-63) for 8 channels of one square: scalar planes
                // first, then the bool planes; bool i = bit i%8 of byte i/8 (bit_buffer.rs:73-75)
                const uint8_t *bb = a.bits + (size_t)board * a.bits_stride;
                const int p = row & 63;
                h16x8 e;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int ch = c * 8 + j;
                    float f = 0.0f;
                    if (ch < a.n_scalar) {
                        f = a.scalars_in[(size_t)board * a.n_scalar + ch];
                    } else if (ch < a.n_scalar + a.n_bool) {
                        const unsigned bit = (unsigned)(ch - a.n_scalar) * 64 + p;
                        f = (float)((bb[bit >> 3] >> (bit & 7)) & 1);
                    }
                    e[j] = (h16)f;
                }
                v = *reinterpret_cast<const uint4 *>(&e);
            } else {
                v = *reinterpret_cast<const uint4 *>(a.x0 + ((size_t)board0 * 64 + row) * a.cin_p + c * 8);
            }
        }
        *reinterpret_cast<uint4 *>(lds + stem_off + row * srow + c * 16) = v;
    }
    __syncthreads();

    // accumulators start at the bias: D = bias + W*X, so the epilogue has no add.  The bias of layer l+1 is fetched
    // while layer l computes: a load that is consumed right away would make the compiler wait for vmcnt(0) and drain
    // the weight prefetch ring once per layer.
    f32x4 acc[4][MT];
    f32x4 bias_next[4];
    auto fetch_bias = [&](int row) __attribute__((always_inline)) {
        const int l = row <= bias_rows ? row : bias_rows;
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
            bias_next[nt] = *reinterpret_cast<const f32x4 *>(a.bias + l * C + wave * 64 + nt * 16 + kq * 4);
    };
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[nt][mt] = bias_next[nt];
    };
    f32x4 post_s[4], post_t[4];  // final BN, fetched at the start of the last layer
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        post_s[nt] = f32x4{1.f, 1.f, 1.f, 1.f};
        post_t[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ReLU on the bit pattern: negative floats are negative integers (no canonicalising v_max inserted)
    auto relu4 = [](f32x4 v) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 b = __builtin_bit_cast(i32x4, v);
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = b[j] > 0 ? b[j] : 0;
        return __builtin_bit_cast(f32x4, b);
    };
    auto to_h4 = [](f32x4 v) { return h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]}; };

    // this lane's pixel row of tile 0 (byte offset in an image; tile mt is mt * L::TS further)
    const int lane_row = L::LINE_TILES ? (fr >> 3) * L::BOARD + (fr & 7) * RS : fr * RS;
    // this lane's slice of the LDS image: its pixel row of tile 0, channels [64*wave + 4*kq, +4) of oc-tile 0
    const int epi_base = lane_row + (wave * 64 + kq * 4) * 2;
    // epilogue: [relu]; [+ residual X]; [final BN]; -> f16 -> LDS image at dst_off.  Flags are compile-time.
    auto epilogue = [&](int dst_off, auto relu, auto residual, auto post) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            h16x4 rx[MT];
            if constexpr (decltype(residual)::value) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    rx[mt] = *reinterpret_cast<const h16x4 *>(lds + L::X_OFF + epi_base + mt * L::TS + nt * 32);
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int off = epi_base + mt * L::TS + nt * 32;
                f32x4 v = acc[nt][mt];
                if constexpr (decltype(relu)::value) v = relu4(v);
                if constexpr (decltype(residual)::value) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] += (float)rx[mt][j];
                }
                if constexpr (decltype(post)::value) v = v * post_s[nt] + post_t[nt];
                *reinterpret_cast<h16x4 *>(lds + dst_off + off) = to_h4(v);
            }
        }
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};

    // tap validity per lane.  NB == 1: the pixel of tile row fr is (y = 2*(mt&3) + (fr>>3), x = fr&7); these four lane
    // masks combined with the (wave-uniform) tap give the invalid lanes without any per-lane arithmetic.  Line tiles
    // (NB == 2): (y = mt, x = fr&7): only the x masks are per lane, a tile outside the board in y is skipped whole.
    const bool x_is0 = (lane & 7) == 0, x_is7 = (lane & 7) == 7, yo_is0 = (lane & 8) == 0, yo_is1 = !yo_is0;
    auto tap_ok = [&](int mt, int dy, int dx) __attribute__((always_inline)) {
        const bool kill_x = (dx < 0 && x_is0) || (dx > 0 && x_is7);
        if constexpr (L::LINE_TILES) return !(kill_x || (dy < 0 && mt == 0) || (dy > 0 && mt == MT - 1));
        else return !(kill_x || ((mt & 3) == 0 && dy < 0 && yo_is0) || ((mt & 3) == 3 && dy > 0 && yo_is1));
    };

    // ---- stem: 9 sc k-steps over the 32 sc (padded) input channels; conv + bias, no activation (post_act.py:205) ----
    fetch_bias(0);
    init_acc();
    fetch_bias(1);
    for (int ks = 0; ks < 9 * sc; ks++) {
        const int tap = WIDE ? ks / sc : ks, chunk = ks - tap * sc;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        h16x8 af[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const uint4 t = a.w_stem[((ks * 4 + wave) * 4 + nt) * 64 + lane];
            af[nt] = *reinterpret_cast<const h16x8 *>(&t);
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const bool ok = tap_ok(mt, dy, dx);
            const int p = L::LINE_TILES ? (fr >> 3) * 64 + mt * 8 + (fr & 7) : mt * 16 + fr;  // row of the stem input
            const int off = ok ? stem_off + (p + dy * 8 + dx) * srow + chunk * 64 + kq * 16 : L::Z_OFF + kq * 16;  // stem: natural k
            const h16x8 bf = *reinterpret_cast<const h16x8 *>(lds + off);
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf, acc[nt][mt], 0, 0, 0);
        }
    }
    epilogue(L::X_OFF, NO, NO, NO);
    __syncthreads();

    // ---- convolution passes over the LDS image ----
    // T[mt] = LDS address of this lane's fragment row (pixel shifted by the tap) or of a zero row.
    // Channel assignment of a k-step: lane group kq reads the 16-byte chunk at 256*(kq&1) + 128*(kq>>1) + 16*ch of the
    // row, i.e. k-step ch covers channels {8ch..8ch+7} + {0, 128, 64, 192}[kq].  The two kq groups that share a
    // ds_read_b128 bank group are then exactly 256 B (one bank row) apart, and with the 16-byte row pad the 16 pixel
    // rows of a fragment read hit 16 different slots: conflict-free for every tap.  (The k order of a sum is free as
    // long as the weights are packed with the same assignment: tower_pack_weights.)  Lanes whose tap falls outside the
    // board read zero row (q & 15), which keeps the same slot pattern.
    const int kq_off = 256 * (kq & 1) + 128 * (kq >> 1);
    const int frag_base = lane_row + kq_off;
    auto tap_rows = [&](int tap, int src_off, int (&T)[MT]) __attribute__((always_inline)) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int shifted = frag_base + src_off + (dy * 8 + dx) * RS;
        // zero row with the slot of the row it stands in for: consecutive line tiles are 8 slots apart, so even and odd
        // tiles differ by 8 rows
        const int zrow = L::Z_OFF + ((fr + dy * 8 + dx) & 15) * RS + kq_off;
        const int zrow_odd = L::LINE_TILES ? L::Z_OFF + ((fr + dy * 8 + dx + 8) & 15) * RS + kq_off : zrow;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) T[mt] = tap_ok(mt, dy, dx) ? shifted + mt * L::TS : ((mt & 1) ? zrow_odd : zrow);
    };

    // One line of taps: acc += sum over taps (DY, dx in [dx_lo, dx_hi)) and all 256 input channels of W * image(src_off),
    // 8 k-steps per tap.  On entry bf[0] holds the fragments of the first k-step; the last k-step prefetches the first
    // fragments of tap `tap_after` (what runs next) for ALL tiles.  With line tiles the tiles that DY puts outside the
    // board are left out: no fragment read, no MFMA.
    h16x8 bf[2][MT];  // activation fragments, double buffered one k-step ahead
    auto conv_line = [&](auto dy_tag, int src_off, int dx_lo, int dx_hi, int tap_after) __attribute__((always_inline)) {
        constexpr int DY = decltype(dy_tag)::value;
        constexpr int LO = L::LINE_TILES && DY < 0 ? 1 : 0, HI = L::LINE_TILES && DY > 0 ? MT - 1 : MT, NT_ = HI - LO;
        int T[MT], Tn[MT];
        tap_rows((DY + 1) * 3 + dx_lo + 1, src_off, T);
#pragma nounroll
        for (int dx = dx_lo; dx < dx_hi; dx++) {
            tap_rows(dx + 1 < dx_hi ? (DY + 1) * 3 + dx + 2 : tap_after, src_off, Tn);
#pragma unroll
            for (int ch = 0; ch < 8; ch++) {
                const int stage = ch & (PF - 1), cur = ch & 1, nxt = cur ^ 1;
                // next k-step's activation fragments: LDS -> registers
#ifndef KZ_TW_NO_DSREAD  // (timing experiments: a build without the fragment reads)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    if (ch < 7) {
                        if (mt >= LO && mt < HI) bf[nxt][mt] = *reinterpret_cast<const h16x8 *>(lds + T[mt] + (ch + 1) * 16);
                    } else {
                        bf[nxt][mt] = *reinterpret_cast<const h16x8 *>(lds + Tn[mt]);
                    }
                }
#endif
                // this k-step's weight fragments were loaded PF k-steps ago
                h16x8 af[4];
                ring_take(stage, af);
                // 4 x NT_ MFMAs on independent accumulators
                // (output-channel tile outermost: the weight fragment is held for all pixel tiles and the activation
                //  fragment changes — same sums, +0.4 % at a full chip against the other order in a same-box A/B)
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
#pragma unroll
                    for (int mt = LO; mt < HI; mt++)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf[cur][mt], acc[nt][mt], 0, 0, 0);
                // issue order: every memory instruction rides in the shadow of one MFMA — first the 4 ring refills,
                // then the LDS fragment reads, then the remaining MFMAs back to back.  (Same-box A/B: +0.5 % over a
                // 2:1 interleave with the LDS reads first, +3 % over the compiler's own order.)
                constexpr int NR_LAST = MT;
                const int nr = ch < 7 ? NT_ : NR_LAST;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    if (i < nr) {
                        __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 1, 0);
                    }
                }
                if (ch < 7) __builtin_amdgcn_sched_group_barrier(SG_MFMA, 4 * NT_ - (NT_ + 4), 0);
                else __builtin_amdgcn_sched_group_barrier(SG_MFMA, 4 * NT_ - (NR_LAST + 4), 0);
                __builtin_amdgcn_sched_barrier(0);
                g++;
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) T[mt] = Tn[mt];
        }
    };
    constexpr std::integral_constant<int, -1> DY_UP{};
    constexpr std::integral_constant<int, 0> DY_MID{};
    constexpr std::integral_constant<int, 1> DY_DOWN{};
    // first fragments of a pass that starts at `tap`
    auto conv_prime = [&](int src_off, int tap) __attribute__((always_inline)) {
        int T[MT];
        tap_rows(tap, src_off, T);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bf[0][mt] = *reinterpret_cast<const h16x8 *>(lds + T[mt]);
    };
    // all nine taps
    auto conv_3x3 = [&](int src_off) __attribute__((always_inline)) {
        conv_prime(src_off, 0);
        conv_line(DY_UP, src_off, -1, 2, 3);
        conv_line(DY_MID, src_off, -1, 2, 6);
        conv_line(DY_DOWN, src_off, -1, 2, 8);
    };
    // the centre tap only (1x1 convolutions of the heads)
    auto conv_1x1 = [&](int src_off) __attribute__((always_inline)) {
        conv_prime(src_off, 4);
        conv_line(DY_MID, src_off, 0, 1, 4);
    };

    // ---- the 2*depth 3x3 convolutions ----
    for (int layer = 1; layer <= layers; layer++) {
        const bool is_b = (layer & 1) == 0;  // conv A: X -> Y; conv B: Y -> X (+ residual)
        init_acc();
        fetch_bias(layer + 1);
        if (layer == layers) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const int oc = wave * 64 + nt * 16 + kq * 4;
                post_s[nt] = *reinterpret_cast<const f32x4 *>(a.post_scale + oc);
                post_t[nt] = *reinterpret_cast<const f32x4 *>(a.post_shift + oc);
            }
        }
        conv_3x3(is_b ? L::Y_OFF : L::X_OFF);
        if (!is_b) epilogue(L::Y_OFF, YES, NO, NO);
        else if (layer != layers) epilogue(L::X_OFF, YES, YES, NO);
        else epilogue(L::X_OFF, YES, YES, YES);
        __syncthreads();
    }

    if constexpr (!HEADS) {
        // ---- write the tower output: coalesced 16-byte stores ----
        for (int id = tid; id < M * 32; id += 256) {
            const int p = id >> 5, c16 = id & 31;
            const int board = board0 + (p >> 6);
            if (board < a.batch) {
                const uint4 v = *reinterpret_cast<const uint4 *>(lds + L::X_OFF + L::row_off(p >> 6, p & 63) + c16 * 16);
                *reinterpret_cast<uint4 *>(a.y + ((size_t)board0 * 64 + p) * C + c16 * 8) = v;
            }
        }
    } else {
        // =====================================================================================================
        // Heads on the LDS-resident tower output X (already through the final BN).
        // =====================================================================================================
        // ---- H1: ScalarHead conv1x1 C->4 + ReLU (post_act.py:14-15), channel-major flatten (:16) -> act[b][c*64+p]
        {
            float *act = reinterpret_cast<float *>(lds + L::ACT_OFF);
            for (int o = tid; o < NB * 256; o += 256) {
                const int b = o >> 8, c4 = (o >> 6) & 3, p = o & 63;
                const unsigned char *row = lds + L::X_OFF + L::row_off(b, p);
                const float *w = a.sh_w0 + c4 * C;
                float s = a.sh_b0[c4];
#pragma unroll 4
                for (int i = 0; i < C; i += 8) {
                    const h16x8 xv = *reinterpret_cast<const h16x8 *>(row + i * 2);
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(w + i), w1 = *reinterpret_cast<const f32x4 *>(w + i + 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) s += (float)xv[j] * w0[j] + (float)xv[4 + j] * w1[j];
                }
                // range check: this sum runs over every channel of one square of the tower output, and an f16 overflow
                // anywhere in the residual stream persists to the tower output (x + relu(..) never removes an inf/NaN)
                if (!(fabsf(s) <= 3.0e38f) && a.nonfinite_flag && board0 + b < a.batch)
                    *reinterpret_cast<volatile int *>(a.nonfinite_flag) = a.epoch;  // (plain store: the flag may be in pinned host memory)
                act[o] = fmaxf(s, 0.0f);
            }
        }

        // ---- H2: conv_under on the 8 squares of rank index 7 (post_act.py:129), as 3 passes of 256 output channels.
        // The host permutes its output channels to s-major (oc' = 256*s + q for original channel 3q + s) so that
        // under.reshape(Q, 24)[q][8s + x] (post_act.py:134) is contiguous in q: row (b*8+x), bytes [512 s, 512 s + 512).
        {
            const int tb = L::X_OFF + L::row_off(NB == 2 ? fr >> 3 : 0, 56 + (fr & 7)) + kq_off;
            for (int s = 0; s < 3; s++) {
                f32x4 ua[4];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) ua[nt] = bias_next[nt];
                fetch_bias(layers + 2 + s);
#pragma unroll
                for (int ch = 0; ch < 8; ch++) {
                    const h16x8 bf = *reinterpret_cast<const h16x8 *>(lds + tb + ch * 16);
                    h16x8 af[4];
                    ring_take(ch & (PF - 1), af);
#pragma unroll
                    for (int nt = 0; nt < 4; nt++)
                        ua[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf, ua[nt], 0, 0, 0);
                    g++;
                }
                if (NB == 2 || fr < 8) {
#pragma unroll
                    for (int nt = 0; nt < 4; nt++)
                        *reinterpret_cast<h16x4 *>(lds + L::U_OFF + fr * URS + (256 * s + wave * 64 + nt * 16 + kq * 4) * 2) =
                            to_h4(ua[nt]);
                }
            }
        }

        // ---- H3: conv_bulk channels [0, Q) = q_from: X -> Y
        init_acc();
        fetch_bias(layers + 5);
        conv_1x1(L::X_OFF);
        epilogue(L::Y_OFF, NO, NO, NO);

        // ---- H4: conv_bulk channels [Q, 2Q) = the 64 board squares of q_to: X -> X in place (every wave reads all of X
        // in its k-loop, so the writes wait for a barrier)
        init_acc();
        conv_1x1(L::X_OFF);
        __syncthreads();
        epilogue(L::X_OFF, NO, NO, NO);
        __syncthreads();

        // ---- H5: ScalarHead Linear(256 -> 32) + ReLU (post_act.py:17-18): 4 lanes per output, 64 inputs each
        {
            const float *act = reinterpret_cast<const float *>(lds + L::ACT_OFF);
            float *hid = reinterpret_cast<float *>(lds + L::HID_OFF);
            if (tid < NB * 128) {
                const int pair = tid >> 2, part = tid & 3, b = pair >> 5, j = pair & 31;
                const float *w = a.sh_w1 + j * 256 + part * 64;
                const float *x = act + b * 256 + part * 64;
                float s = 0.0f;
#pragma unroll 4
                for (int i = 0; i < 64; i += 4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + i), xv = *reinterpret_cast<const f32x4 *>(x + i);
                    s += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
                }
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if (part == 0) hid[b * 32 + j] = fmaxf(s + a.sh_b1[j], 0.0f);
            }
        }

        // ---- H6: attention logits (post_act.py:138): L[i][j] = sum_q q_from[q][i] * q_to[q][j] / sqrt(Q).
        // MFMA with q_to rows as the A operand and q_from rows as the B operand: a lane ends up with 4 consecutive j
        // of one i.  q_to row j: j < 64 -> X row j (conv_bulk upper half); j = 64 + 8s + x -> under image row x, block s.
        constexpr int IT = NB == 2 ? 2 : 1;  // i-tiles per wave
        const int hb = NB == 2 ? wave >> 1 : 0;
        const int it0 = NB == 2 ? (wave & 1) * 2 : wave;
        f32x4 la[IT][6];
        {
            int ja[6], ia[IT];
#pragma unroll
            for (int jt = 0; jt < 4; jt++) ja[jt] = L::X_OFF + L::row_off(hb, jt * 16 + fr) + kq_off;
            ja[4] = L::U_OFF + (hb * 8 + (fr & 7)) * URS + (fr >> 3) * 512 + kq_off;  // t = fr: s = fr>>3 in {0,1}
            ja[5] = L::U_OFF + (hb * 8 + (fr & 7)) * URS + 2 * 512 + kq_off;          // t = 16 + fr: s = 2 (fr >= 8: pad)
#pragma unroll
            for (int ii = 0; ii < IT; ii++) ia[ii] = L::Y_OFF + L::row_off(hb, (it0 + ii) * 16 + fr) + kq_off;
#pragma unroll
            for (int ii = 0; ii < IT; ii++)
#pragma unroll
                for (int jt = 0; jt < 6; jt++) la[ii][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < 8; ch++) {
                h16x8 qa[6], qb[IT];
#pragma unroll
                for (int jt = 0; jt < 6; jt++) qa[jt] = *reinterpret_cast<const h16x8 *>(lds + ja[jt] + ch * 16);
#pragma unroll
                for (int ii = 0; ii < IT; ii++) qb[ii] = *reinterpret_cast<const h16x8 *>(lds + ia[ii] + ch * 16);
#pragma unroll
                for (int ii = 0; ii < IT; ii++)
#pragma unroll
                    for (int jt = 0; jt < 6; jt++)
                        la[ii][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[jt], qb[ii], la[ii][jt], 0, 0, 0);
            }
        }
        __syncthreads();  // every wave is done reading q_from (Y), q_to (X, U) and has written hid
        {
            const float inv = 1.0f / sqrtf((float)C);
#pragma unroll
            for (int ii = 0; ii < IT; ii++)
#pragma unroll
                for (int jt = 0; jt < 6; jt++) {
                    const int i = (it0 + ii) * 16 + fr, j = jt * 16 + kq * 4;
                    *reinterpret_cast<f32x4 *>(lds + L::LOG_OFF + ((hb * 64 + i) * LOGIT_LD + j) * 4) = la[ii][jt] * inv;
                }
        }
        // ---- H7: ScalarHead Linear(32 -> 5) (post_act.py:19)
        const bool decode = a.dec.move_offsets != nullptr;
        float *raw = reinterpret_cast<float *>(lds + L::ACT_OFF);  // (act is dead since H5: the decode's five scalars per board)
        if (tid < NB * 5) {
            const int b = tid / 5, k = tid % 5;
            const float *hid = reinterpret_cast<const float *>(lds + L::HID_OFF) + b * 32;
            float s = a.sh_b2[k];
            for (int i = 0; i < 32; i++) s += a.sh_w2[k * 32 + i] * hid[i];
            if (decode) raw[b * 8 + k] = s;
            else if (board0 + b < a.batch) a.scalars[(size_t)(board0 + b) * 5 + k] = s;
        }
        __syncthreads();
        if (decode) {
            // ---- H8': decode_output (common.rs:16-100) on the LDS-resident logits: wave b decodes board b.  The gather
            // policy.flatten(1)[:, FLAT_TO_ATT] (post_act.py:140) happens per available move; q_to's image (X) is dead and
            // holds the wave's staging
            if (wave < NB && board0 + wave < a.batch) {
                const float *lg = reinterpret_cast<const float *>(lds + L::LOG_OFF) + wave * 64 * LOGIT_LD;
                float *stage = reinterpret_cast<float *>(lds + L::X_OFF + L::row_off(wave, 0));
                decode_board_wave(a.dec, board0 + wave, lane, raw + wave * 8, stage, 64 * RS / 4,
                                  [&](int idx) { return lg[a.att_idx[idx]]; });
            }
        } else {
            // ---- H8: policy.flatten(1)[:, FLAT_TO_ATT] (post_act.py:140): coalesced 1880-float rows
            for (int b = 0; b < NB; b++) {
                if (board0 + b >= a.batch) break;
                const float *lg = reinterpret_cast<const float *>(lds + L::LOG_OFF) + b * 64 * LOGIT_LD;
                float *pol = a.policy + (size_t)(board0 + b) * POLICY;
                for (int k = tid; k < POLICY; k += 256) pol[k] = lg[a.att_idx[k]];
            }
        }
    }
}

#ifndef KZ_PF_NB1
#define KZ_PF_NB1 4
#endif
#ifndef KZ_PF_NB2
#define KZ_PF_NB2 4
#endif
constexpr int PF_NB1 = KZ_PF_NB1, PF_NB2 = KZ_PF_NB2;

int boards_per_wg() {
    static int nb = [] {
        const char *e = getenv("KZ_TOWER_NB");
        int v = e ? atoi(e) : 2;
        return v == 1 ? 1 : 2;
    }();
    return nb;
}

// [256 out][256 in] row-major f32 -> one 8-k-step pass of the weight stream (same layout as a tap of a 3x3 layer)
void pack_1x1(const float *w, uint16_t *dst) {
    static const int kq_base[4] = {0, 128, 64, 192};
    for (int chunk = 0; chunk < 8; chunk++)
        for (int wave = 0; wave < 4; wave++)
            for (int nt = 0; nt < 4; nt++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int oc = 64 * wave + 16 * nt + (lane & 15);
                        const int ch = 8 * chunk + kq_base[lane >> 4] + j;
                        const _Float16 hv = (_Float16)w[(size_t)oc * 256 + ch];
                        uint16_t bits;
                        __builtin_memcpy(&bits, &hv, 2);
                        dst[((((size_t)chunk * 4 + wave) * 4 + nt) * 64 + lane) * 8 + j] = bits;
                    }
}

}  // namespace

int tower_resident_boards_per_workgroup() { return boards_per_wg(); }

// (input planes in chunks of 32; beyond one chunk they are staged in the Y image: 64 B per chunk and row; 7 chunks at
// most — tower_pack_weights tells a tower layer from a stem by cin_p == 256)
bool tower_resident_supported(int dtype, int h, int w, int channels, int depth, int c_in) {
    return dtype == 1 && h == 8 && w == 8 && channels == C && depth >= 1 && c_in >= 1 && c_in <= 224;
}

bool tower_heads_supported(int policy_kind, int query_channels, int policy_len, int sh_channels, int sh_size) {
    return policy_kind == 2 && query_channels == C && policy_len == POLICY && sh_channels == 4 && sh_size == 32;
}

size_t tower_packed_weight_elems(int cin_p, int depth) {
    return (size_t)9 * C * cin_p + (size_t)2 * depth * 9 * C * C;
}

size_t tower_heads_weight_elems() { return (size_t)HEAD_KSTEPS * 16 * 1024 / 2; }

// OIHW f32 -> [tap 9][chunk cin_p/32][wave 4][nt 4][lane 64][8] f16: element j of lane (fr, kq) of (wave, nt) is
// W[oc = 64*wave + 16*nt + fr][channel][tap] — the A fragment of v_mfma_f32_16x16x32_f16 — where the k-step's channel
// assignment is the kernel's: cin_p == 256: channel = 8*chunk + {0,128,64,192}[kq] + j (bank-conflict-free LDS reads);
// stem (cin_p a multiple of 32 below 256: ceil(c_in / 32) chunks): channel = 32*chunk + 8*kq + j.
void tower_pack_weights(const float *oihw, int cout, int cin, int cin_p, uint16_t *dst) {
    const int nchunk = cin_p / 32;
    static const int kq_base[4] = {0, 128, 64, 192};
    for (int tap = 0; tap < 9; tap++)
        for (int chunk = 0; chunk < nchunk; chunk++)
            for (int wave = 0; wave < 4; wave++)
                for (int nt = 0; nt < 4; nt++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int oc = 64 * wave + 16 * nt + (lane & 15);
                            const int kq = lane >> 4;
                            const int ch = cin_p == 256 ? 8 * chunk + kq_base[kq] + j : 32 * chunk + 8 * kq + j;
                            float v = 0.0f;
                            if (oc < cout && ch < cin) v = oihw[((size_t)oc * cin + ch) * 9 + tap];
                            const _Float16 hv = (_Float16)v;
                            uint16_t bits;
                            __builtin_memcpy(&bits, &hv, 2);
                            dst[(((((size_t)tap * nchunk + chunk) * 4 + wave) * 4 + nt) * 64 + lane) * 8 + j] = bits;
                        }
}

// Heads part of the weight stream, in execution order: conv_under pass s = 0,1,2 (output channel 3q + s -> row q),
// conv_bulk rows [0,256) (q_from), conv_bulk rows [256,512) (q_to).  bias5: the matching 5 x 256 bias rows.
void tower_pack_heads(const float *w_bulk /*[512][256]*/, const float *b_bulk, const float *w_under /*[768][256]*/,
                      const float *b_under, uint16_t *dst, float *bias5) {
    const size_t pass = (size_t)8 * 16 * 1024 / 2;
    std::vector<float> tmp((size_t)256 * 256);
    for (int s = 0; s < 3; s++) {
        for (int q = 0; q < 256; q++) {
            for (int c = 0; c < 256; c++) tmp[(size_t)q * 256 + c] = w_under[(size_t)(3 * q + s) * 256 + c];
            bias5[s * 256 + q] = b_under[3 * q + s];
        }
        pack_1x1(tmp.data(), dst + pass * s);
    }
    pack_1x1(w_bulk, dst + pass * 3);
    pack_1x1(w_bulk + (size_t)256 * 256, dst + pass * 4);
    for (int q = 0; q < 256; q++) {
        bias5[3 * 256 + q] = b_bulk[q];
        bias5[4 * 256 + q] = b_bulk[256 + q];
    }
}

void launch_tower_resident(const TowerArgs &t, hipStream_t stream) {
    TowerDev d{};
    d.x0 = static_cast<const h16 *>(t.x0);
    d.w_stem = static_cast<const uint4 *>(t.w_stem);
    d.w_tower = static_cast<const uint4 *>(t.w_tower);
    d.bias = t.bias;
    d.post_scale = t.post_scale;
    d.post_shift = t.post_shift;
    d.y = static_cast<h16 *>(t.y);
    d.cin_p = t.cin_p;
    d.batch = t.batch;
    d.depth = t.depth;
    d.bits = t.bits; d.bits_stride = t.bits_stride; d.scalars_in = t.scalars_in; d.n_scalar = t.n_scalar;
    d.n_bool = t.n_bool;
    d.sh_w0 = t.sh_w0; d.sh_b0 = t.sh_b0; d.sh_w1 = t.sh_w1; d.sh_b1 = t.sh_b1; d.sh_w2 = t.sh_w2; d.sh_b2 = t.sh_b2;
    d.att_idx = t.att_idx;
    d.scalars = t.scalars;
    d.policy = t.policy;
    d.nonfinite_flag = t.nonfinite_flag;
    d.epoch = t.epoch;
    d.dec = DecodeDev{t.fused_heads ? t.decode.move_offsets : nullptr, t.decode.move_indices, t.decode.values, t.decode.probs,
                      t.decode.error_flag, POLICY};
    const bool heads = t.fused_heads;
    // (the dynamic-LDS attribute is per device: set it on every launch's current device, it is a cheap host call,
    //  but only once per kernel and device)
    auto launch = [&](auto kernel, int grid, int bytes) {
        static thread_local unsigned long long done_mask = 0;  // per instantiation (the lambda is generic)
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!((done_mask >> (dev & 63)) & 1)) {
            (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            done_mask |= 1ull << (dev & 63);
        }
        kernel<<<grid, 256, bytes, stream>>>(d);
    };
    if (t.cin_p > 32) {  // (ChessHistoryMapper: 34 / 47 / 60 planes; always two boards per workgroup)
        const int grid = (t.batch + 1) / 2;
        if (heads) launch(kz_tower_resident<2, true, PF_NB2, true>, grid, Layout<2>::BYTES);
        else launch(kz_tower_resident<2, false, PF_NB2, true>, grid, Layout<2>::BYTES);
    } else if (boards_per_wg() == 1) {
        if (heads) launch(kz_tower_resident<1, true, PF_NB1>, t.batch, Layout<1>::BYTES);
        else launch(kz_tower_resident<1, false, PF_NB1>, t.batch, Layout<1>::BYTES);
    } else {
        const int grid = (t.batch + 1) / 2;
        if (heads) launch(kz_tower_resident<2, true, PF_NB2>, grid, Layout<2>::BYTES);
        else launch(kz_tower_resident<2, false, PF_NB2>, grid, Layout<2>::BYTES);
    }
}

}  // namespace kz
