// kz_tower_f32.hip — board-resident ResTower in exact f32 (the <= 1e-4 parity precision): ONE launch for the stem and all
// 2*depth 3x3 convolutions, activations never leave LDS.  Same organisation as the f16 launch (kz_tower.hip), sized for
// v_mfma_f32_16x16x4_f32 (64 FLOP/clk/SIMD: 157 TFLOP/s per MI355X):
//
//   workgroup = NT tiles of 16 pixel rows = floor(16 NT / hw) whole boards, packed densely (row r = board r / hw,
//               pixel r % hw); 256 threads = 4 waves; wave w owns output channels [w C/4, (w+1) C/4) of ALL rows, so an
//               activation fragment is read from LDS once per wave and the weights exactly once per workgroup
//   LDS       = 16 zero rows + two row images (in / out, swapped per layer) of 16 NT rows x (4 C + 16) B.
//               chess 20x256 (NT 4): 146 KB; Ataxx 8x128 (NT 7, two boards): 124 KB
//   k order   = a step is 16 input channels of one tap = four MFMA k-steps: lane (fr, kq) holds 4 CONSECUTIVE
//               channels of pixel row fr — one ds_read_b128 — at byte plane(kq) + 16 g of the row, where the planes of
//               the two lane groups that share a ds_read_b128 bank group are a multiple of 256 B apart and the row
//               stride is an odd number of 16-byte slots: conflict-free for every tap.  MFMA k-step s pairs element s of
//               the activation fragment with element s of the weight fragment, which the host packed in the same order.
//   weights   = streamed L2 -> registers in fragment order (16 B per lane per step and 16-channel output tile) through
//               a four-step ring that runs on across layer boundaries; a step is 4 * (C/64) * NT MFMAs of 32 cycles
//   epilogue  = bias-initialised accumulators (the next layer's bias is fetched a layer ahead); [ReLU]; [+ residual, in
//               place in the out image]; [final BN]; 16-byte stores (4 consecutive output channels of a pixel row per lane)
//   encode    = packed boards (bit planes + scalar planes) are decoded while the stem input is staged
//   heads     = (HEADS launches: conv / ataxx_conv policy head + scalar head) the policy head's Conv1x1 C->C + ReLU is
//               one more layer of the centre tap only, at the end of the same weight stream; the 1x1 convolutions with
//               few output channels (scalar head, extra moves, policy planes) are two small MFMA passes over the two
//               images in LDS, the Linears a few hundred FMAs per thread — the tower output never reaches HBM and a
//               batch is ONE launch
//
// Arithmetic follows python/lib/model/post_act.py:201-239 (tower), :8-31 (scalar head), :75-110 (conv policy heads) with
// Conv+BN folded on the host (kz_model.cpp).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kz_kernels.hpp"

namespace kz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// Diagnostic build only (-DKZ_T32_STAMPS): s_memtime stamps at the phase boundaries of every wave, dumped by the
// launcher to $KZ_T32_STAMP_FILE after the 20th launch (tools/tower32_stamps.py reads them).  No stamp executes in the
// real kernel.
#ifdef KZ_T32_STAMPS
#define KZ_STAMP(slot)                                                                         \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (lane == 0 && (slot) < 64) a.stamps[((size_t)blockIdx.x * 4 + wave) * 64 + (slot)] = t_; \
    } while (0)
#else
#define KZ_STAMP(slot) do { } while (0)
#endif
#define KZ_HEADS_STAMP(slot) KZ_STAMP(slot)
#include "kz_decode_dev.hpp"  // DecodeDev, decode_board_wave: decode_output as the last step of the launch
#include "kz_conv_heads.hpp"  // plane_of, conv_heads_f32 (inside this namespace)

struct TowerF32Dev {
    const float *x0;    // encoded input [batch*hw][ldx0]
    const f32x4 *w;     // fragment-packed: stem steps, then 2*depth layers of 9*C/16 steps (tower32_pack_weights)
    const float *bias;  // [1 + 2*depth][C]
    const float *post_scale, *post_shift;  // final BN [C]
    float *y;           // tower output [batch*hw][ldy]
    int ldx0, ldy, batch, h, w_, hw, depth, stem_groups, nb;
    unsigned inv_w, inv_hw;  // ceil(65536 / w), ceil(65536 / hw): exact quotients for values < 512
    // fused board encode (bits != nullptr: x0 is not read)
    const uint8_t *bits;
    size_t bits_stride;
    const float *scalars_in;
    int n_scalar, n_bool;
    // fused heads (HEADS launches)
    int hc, hs, pc, policy_len, zero_tail, extra, epoch;
    const float *sh_b0, *sh_w1t, *sh_b1, *sh_w2, *sh_b2, *p_b1, *pe_bc, *pe_wl, *pe_bl;
    const f32x4 *small_w;  // tower32_pack_small_weights: [over x: scalar-head conv rows, extra-move conv row][over hidden: policy conv]
    float *scalars, *policy;
    int *nonfinite_flag;
    DecodeDev dec;  // dec.move_offsets set: decode_output inside the launch (policy must be device memory then)
    unsigned long long *stamps;  // (diagnostic build)
};


// DENSE: an image holds the workgroup's nb * hw pixel rows and nothing behind them (the rows that fill the last tile up
// to 16 are not stored: as inputs they are zero rows, as outputs they are not written) — what makes room for THREE 7x7
// boards (147 rows in ten tiles, 160 KB of LDS to the byte) where the padded image holds two (98 rows in seven tiles).
template <int C, int NT, bool HEADS, bool DENSE = false>
__global__ __launch_bounds__(256, 1) void kz_tower_resident_f32(TowerF32Dev a) {
    constexpr int OT = C / 64;           // 16-channel output tiles per wave
    constexpr int G = C / 16;            // steps per tap in a tower layer
    constexpr int ROWS = NT * 16;
    constexpr int RS = C * 4 + 16;       // row stride: odd number of 16-byte slots
    constexpr int ZERO = 0, IMG0 = 16 * RS;
    constexpr int STEP = 4 * OT * 64;    // f32x4 per step: [wave][ot][lane]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int board0 = blockIdx.x * a.nb;
    const int boards = min(a.nb, a.batch - board0);
    const int rows_valid = boards * a.hw;
    const int rows_img = DENSE ? a.nb * a.hw : ROWS;  // rows an image holds
    const int IMG1 = IMG0 + rows_img * RS;
    KZ_STAMP(0);

    // zero rows; padding rows of image 0 (their outputs are never stored, but keep them finite)
    for (int i = tid; i < 16 * RS / 16; i += 256) *reinterpret_cast<f32x4 *>(lds + ZERO + i * 16) = f32x4{0, 0, 0, 0};
    // stage the encoded input: stem_groups * 16 channels per row
    {
        const int pieces = a.stem_groups * 4;  // 16-byte pieces per row
        for (int i = tid; i < ROWS * pieces; i += 256) {
            const int r = i / pieces, p = i - r * pieces;
            if (DENSE && r >= rows_img) continue;
            f32x4 v = f32x4{0, 0, 0, 0};
            if (r < rows_valid) {
                if (a.bits) {  // F0 (rust/kz-core/src/mapping/mod.rs:40-63, bit order bit_buffer.rs:73-75): scalar planes, then bit planes
                    const int bb = (int)(((unsigned)r * a.inv_hw) >> 16), q = r - bb * a.hw;
                    const uint8_t *bits = a.bits + (size_t)(board0 + bb) * a.bits_stride;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int c = p * 4 + j;
                        float f = 0.0f;
                        if (c < a.n_scalar) f = a.scalars_in[(size_t)(board0 + bb) * a.n_scalar + c];
                        else if (c < a.n_scalar + a.n_bool) {
                            const unsigned bit = (unsigned)(c - a.n_scalar) * a.hw + q;
                            f = (float)((bits[bit >> 3] >> (bit & 7)) & 1);
                        }
                        v[j] = f;
                    }
                } else {
                    v = *reinterpret_cast<const f32x4 *>(a.x0 + ((size_t)board0 * a.hw + r) * a.ldx0 + p * 4);
                }
            }
            *reinterpret_cast<f32x4 *>(lds + IMG0 + r * RS + p * 16) = v;
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
    KZ_STAMP(1);

    const f32x4 *wl = a.w + wave * OT * 64 + lane;  // this lane's fragment of step 0
    int in = IMG0, out = IMG1;
    f32x4 acc[OT][NT];
    int T[NT];

    auto init_acc = [&](int layer) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + (size_t)layer * C + (wave * OT + ot) * 16 + kq * 4);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[ot][nt] = b;
        }
    };
    // LDS address of this lane's fragment row per tile for one tap (the shifted pixel row, or a zero row)
    auto tap_rows = [&](int tap, int koff) {
        const int shift = (tap / 3 - 1) * a.w_ + (tap % 3 - 1);
        const unsigned ok = tap == 0   ? okmask[0] : tap == 1 ? okmask[1] : tap == 2 ? okmask[2] : tap == 3 ? okmask[3]
                            : tap == 4 ? okmask[4] : tap == 5 ? okmask[5] : tap == 6 ? okmask[6] : tap == 7 ? okmask[7]
                                                                                                           : okmask[8];
        const int shifted = in + (fr + shift) * RS + koff;
        const int zero = ZERO + ((fr + shift) & 15) * RS + koff;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) T[nt] = ((ok >> nt) & 1) ? shifted + nt * 16 * RS : zero;
    };
    auto mfma_step = [&](const f32x4 (&wf)[OT], const f32x4 (&bf)[NT]) {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int ot = 0; ot < OT; ot++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc[ot][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ot][s], bf[nt][s], acc[ot][nt], 0, 0, 0);
    };
    // rows this lane may store (bit nt: row fr of tile nt is part of the image)
    unsigned wmask = 0;
#pragma unroll
    for (int nt = 0; nt < NT; nt++) wmask |= (unsigned)(nt * 16 + fr < rows_img) << nt;
    // [relu]; [+ residual from the out image, in place]; [final BN]; 16-byte stores into the out image
    auto epilogue = [&](bool relu, bool residual, bool post) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            const int oc = (wave * OT + ot) * 16 + kq * 4;
            f32x4 ps = f32x4{1, 1, 1, 1}, pt = f32x4{0, 0, 0, 0};
            if (post) {
                ps = *reinterpret_cast<const f32x4 *>(a.post_scale + oc);
                pt = *reinterpret_cast<const f32x4 *>(a.post_shift + oc);
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                if (DENSE && !((wmask >> nt) & 1)) continue;
                f32x4 *slot = reinterpret_cast<f32x4 *>(lds + out + (nt * 16 + fr) * RS + oc * 4);
                f32x4 v = acc[ot][nt];
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                }
                if (residual) v += *slot;  // x + relu(bn(conv(.))) (post_act.py:227-228)
                if (post) v = v * ps + pt;
                *slot = v;
            }
        }
    };

    // ---- stem: conv + bias, no activation (post_act.py:205); stem_groups steps of 16 channels per tap ----
    {
        init_acc(0);
        const int koff = 16 * kq;  // input rows are short: channels 16 g + 4 kq + s at byte 64 g + 16 kq
        for (int tap = 0; tap < 9; tap++) {
            tap_rows(tap, koff);
            for (int g = 0; g < a.stem_groups; g++) {
                f32x4 wf[OT], bf[NT];
                const f32x4 *p = wl + (size_t)(tap * a.stem_groups + g) * STEP;
#pragma unroll
                for (int ot = 0; ot < OT; ot++) wf[ot] = p[ot * 64];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) bf[nt] = *reinterpret_cast<const f32x4 *>(lds + T[nt] + g * 64);
                mfma_step(wf, bf);
            }
        }
        epilogue(false, false, a.depth == 0);
        __syncthreads();
        in = IMG1;
        out = IMG0;
    }
    KZ_STAMP(2);

    // ---- 2*depth tower convolutions: 9 G steps each.  Weights run FOUR steps ahead through a register ring (a step is
    // 4*OT*NT MFMAs of 32 cycles = 0.75 us at A1: two steps of cover were less than an L2 round trip under load), the
    // activation fragments one step ahead in the other of two buffers; inside a step every fragment read and ring refill
    // is placed in the shadow of MFMAs (24 free issue cycles per f32 MFMA) instead of in a burst between two blocks of
    // MFMAs.  Round 2, A1 at a full chip: MFMA-busy 83.7 % -> see DESIGN.md §5.3. ----
    const int koff = plane_of<C>(kq);
    constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100;
    static_assert(G % 4 == 0 && G % 2 == 0, "the ring stage and the fragment buffer of a step are compile-time constants");
    // The ring runs across layer boundaries (the stream is contiguous, and the host pads it with four steps so the ring
    // may run past the end).  HEADS: the policy head's Conv1x1 C->C + ReLU rides as one more layer of the centre tap
    // only (G steps).
    // A tap (G steps) is unrolled: the ring stage, the fragment buffer and the channel offset of every read are
    // compile-time, the weight address is a uniform base advanced once per step plus a constant lane offset — every
    // instruction that is not an MFMA costs ~6 cycles of the (single wave's) matrix pipe, measured with in-kernel
    // stamps: 27 of them per step were 8 % of the loop.
    const int n_layers = 2 * a.depth + (HEADS ? 1 : 0);
#ifndef KZ_T32_RING
#define KZ_T32_RING 4
#endif
    constexpr int RING = KZ_T32_RING;
    static_assert(G % RING == 0, "the ring stage of a step is a compile-time constant");
    f32x4 wring[RING][OT];
    const int wlane = wave * OT * 64 + lane;
    const f32x4 *wnext = a.w + (size_t)9 * a.stem_groups * STEP;  // uniform
    auto issue_w = [&](f32x4(&wf)[OT]) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) wf[ot] = wnext[wlane + ot * 64];
        wnext += STEP;
    };
#pragma unroll
    for (int u = 0; u < RING; u++) issue_w(wring[u]);
    // the bias of the next layer is fetched while this one computes
    f32x4 bias_next[OT];
    auto fetch_bias = [&](int layer) {
        const int l = layer <= n_layers ? layer : n_layers;
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
            bias_next[ot] = *reinterpret_cast<const f32x4 *>(a.bias + (size_t)l * C + (wave * OT + ot) * 16 + kq * 4);
    };
    fetch_bias(1);
    for (int layer = 1; layer <= n_layers; layer++) {
        const bool head = HEADS && layer > 2 * a.depth;
        const int tap_first = head ? 4 : 0, tap_last = head ? 4 : 8;
        f32x4 bA[NT], bB[NT];
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[ot][nt] = bias_next[ot];
        fetch_bias(layer + 1);
        tap_rows(tap_first, koff);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) bA[nt] = *reinterpret_cast<const f32x4 *>(lds + T[nt]);
#pragma unroll 1
        for (int tap = tap_first; tap <= tap_last; tap++) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                f32x4(&cur)[NT] = (g & 1) ? bB : bA;
                f32x4(&nxt)[NT] = (g & 1) ? bA : bB;
                // this step's fragments were requested early in the previous step: ONE wait here instead of one in front
                // of every tile's first MFMA (lgkmcnt(0), vmcnt/expcnt untouched)
                __builtin_amdgcn_s_waitcnt(0xC07F);
                if (g + 1 < G) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) nxt[nt] = *reinterpret_cast<const f32x4 *>(lds + T[nt] + (g + 1) * 16);
                } else {  // first step of the next tap (after the last tap: a read nobody uses)
                    tap_rows(tap < tap_last ? tap + 1 : tap_last, koff);
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) nxt[nt] = *reinterpret_cast<const f32x4 *>(lds + T[nt]);
                }
                mfma_step(wring[g % RING], cur);
                issue_w(wring[g % RING]);  // (this stage's fragments have been issued to the MFMAs)
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 4 * OT * NT - 4 * NT - 4 * OT, 0);
#pragma unroll
                for (int i = 0; i < OT; i++) {
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        KZ_STAMP(3 * layer);
        const bool second = (layer & 1) == 0;  // the block's second conv: residual add, and the final BN on the last
        epilogue(true, second, second && layer == 2 * a.depth);
        KZ_STAMP(3 * layer + 1);
        __syncthreads();
        KZ_STAMP(3 * layer + 2);
        const int tmp = in;
        in = out;
        out = tmp;
    }

    if constexpr (!HEADS) {
        // ---- the tower's output (the last `out`, now `in`) -> global, 16 bytes per lane, whole rows ----
        for (int i = tid; i < rows_valid * (C / 4); i += 256) {
            const int r = i / (C / 4), p = i - r * (C / 4);
            *reinterpret_cast<f32x4 *>(a.y + ((size_t)board0 * a.hw + r) * a.ldy + p * 4) =
                *reinterpret_cast<const f32x4 *>(lds + in + r * RS + p * 16);
        }
    } else {
        // ---- heads on the images in LDS: `out` holds the tower output, `in` the policy head's hidden layer; the zero rows
        // are dead by now: scratch for the scalar head (kz_conv_heads.hpp) ----
        conv_heads_f32<C, NT>(a, lds, ZERO, out, in, board0, boards, rows_valid, rows_img);
        KZ_STAMP(61);
    }
    KZ_STAMP(62);
}

int tiles_for(int hw, int channels) {
    if (channels == 256) return hw <= 64 ? 4 : 0;
    if (channels == 128) return hw * 2 <= 112 ? 7 : hw <= 64 ? 4 : hw <= 96 ? 6 : 0;
    return 0;
}

template <int C, int NT, bool HEADS, bool DENSE = false>
void launch1(const TowerF32Dev &d, int grid, hipStream_t stream) {
    const int LDS_BYTES = (16 + 2 * (DENSE ? d.nb * d.hw : NT * 16)) * (C * 4 + 16);
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_tower_resident_f32<C, NT, HEADS, DENSE>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        done_mask |= 1ull << (dev & 63);
    }
#ifdef KZ_T32_STAMPS
    static unsigned long long *stamp_buf = nullptr;
    static int launches = 0;
    const size_t stamp_bytes = (size_t)grid * 4 * 64 * sizeof(unsigned long long);
    if (!stamp_buf) (void)hipMalloc((void **)&stamp_buf, (size_t)4096 * 4 * 64 * 8);
    TowerF32Dev ds = d;
    ds.stamps = stamp_buf;
    if (launches == 20) (void)hipMemsetAsync(stamp_buf, 0, stamp_bytes, stream);
    kz_tower_resident_f32<C, NT, HEADS, DENSE><<<grid, 256, LDS_BYTES, stream>>>(ds);
    if (launches++ == 20 && getenv("KZ_T32_STAMP_FILE")) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> host(stamp_bytes / 8);
        (void)hipMemcpy(host.data(), stamp_buf, stamp_bytes, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(getenv("KZ_T32_STAMP_FILE"), "wb")) {
            fwrite(host.data(), 1, stamp_bytes, f);
            fclose(f);
        }
    }
#else
    kz_tower_resident_f32<C, NT, HEADS, DENSE><<<grid, 256, LDS_BYTES, stream>>>(d);
#endif
}

template <int C, int NT, bool DENSE = false>
void launch(const TowerF32Dev &d, bool heads, int grid, hipStream_t stream) {
    if (heads) launch1<C, NT, true, DENSE>(d, grid, stream);
    else launch1<C, NT, false, DENSE>(d, grid, stream);
}

}  // namespace

bool tower32_supported(int dtype, int h, int w, int channels, int depth) {
    return dtype == 0 && depth >= 1 && h >= 2 && w >= 2 && w <= 32 && tiles_for(h * w, channels) != 0;
}

bool tower32_heads_supported(int policy_kind, int extra_moves, int pc, int h, int w, int channels, int hc, int hs) {
    return conv_heads_fit(tiles_for(h * w, channels), policy_kind, extra_moves, pc, h, w, channels, hc, hs);
}

// kz_conv_heads.hpp on a launch of nt tiles of 16 rows: at most 16 tiles (four per wave) and four boards
bool conv_heads_fit(int nt, int policy_kind, int extra_moves, int pc, int h, int w, int channels, int hc, int hs, size_t scratch_bytes) {
    const int hw = h * w;
    if (!nt || nt > 16 || (policy_kind != 0 && policy_kind != 1) || extra_moves < 0 || hc < 1 || hs < 1 || hs > 256) return false;
    if (policy_kind == 0 && extra_moves) return false;
    if (pc < 1 || pc > 32 || hc + (extra_moves ? 1 : 0) > 32) return false;  // two 16-channel tiles per small conv
    const int nb = nt * 16 / hw, nseg = 256 / hs;
    if (nb > 4) return false;
    // the in-launch decode keeps the five raw scalars of board b at sred[8 b ..]: the partial-sum region (nseg * nb * hs floats)
    // must hold nb * 8 (always true for hs <= 256 — nseg * hs > 128 — and checked where the assumption is made)
    if (nseg * hs < 8) return false;
    // the zero rows' LDS: conv activations, extra-move plane, hidden, last Linear's weights, partial sums
    const size_t floats = (size_t)nb * hc * hw + (size_t)nb * hw + (size_t)nb * hs + (size_t)5 * hs + (size_t)nseg * nb * hs;
    // (scratch_bytes = 0: the zero rows of the f32 images, what the exact-f32 and the split launches have)
    return floats * 4 <= (scratch_bytes ? scratch_bytes : (size_t)16 * (channels * 4 + 16));
}

size_t tower32_heads_weight_elems(int channels) { return (size_t)channels * channels; }

// [C][C] 1x1 conv -> the fragment order of one tap of a tower layer
void tower32_pack_head_weights(const float *oi, int channels, float *dst) {
    const int groups = channels / 16, ot_n = channels / 64;
    size_t o = 0;
    for (int g = 0; g < groups; g++)
        for (int wave = 0; wave < 4; wave++)
            for (int ot = 0; ot < ot_n; ot++)
                for (int lane = 0; lane < 64; lane++)
                    for (int s = 0; s < 4; s++) {
                        const int fr = lane & 15, kq = lane >> 4;
                        const int oc = 16 * (wave * ot_n + ot) + fr;
                        const int plane = channels == 256 ? 256 * kq : 256 * (kq & 1) + 128 * (kq >> 1);
                        dst[o++] = oi[(size_t)oc * channels + plane / 4 + 4 * g + s];
                    }
}

size_t tower32_small_weight_elems(int channels) { return (size_t)2 * (channels / 16) * 2 * 64 * 4; }

// The two small 1x1 convolutions of the heads, each as [g][tile 2][lane 64][4] (32 output rows, zero-padded): first the
// one over the tower output (rows: the scalar head's hc conv filters, then the extra moves' single filter if any), then
// the policy conv over the hidden layer (pc rows).  Fragment element order of a tower layer's tap.
void tower32_pack_small_weights(const float *sh_w0, int hc, const float *pe_wc, const float *p_w1, int pc, int channels,
                                float *dst) {
    const int groups = channels / 16;
    size_t o = 0;
    for (int conv = 0; conv < 2; conv++)
        for (int g = 0; g < groups; g++)
            for (int mt = 0; mt < 2; mt++)
                for (int lane = 0; lane < 64; lane++)
                    for (int s = 0; s < 4; s++) {
                        const int fr = lane & 15, kq = lane >> 4, oc = 16 * mt + fr;
                        const int plane = channels == 256 ? 256 * kq : 256 * (kq & 1) + 128 * (kq >> 1);
                        const int ch = plane / 4 + 4 * g + s;
                        float v = 0.0f;
                        if (conv == 0) {
                            if (oc < hc) v = sh_w0[(size_t)oc * channels + ch];
                            else if (oc == hc && pe_wc) v = pe_wc[ch];
                        } else if (oc < pc) {
                            v = p_w1[(size_t)oc * channels + ch];
                        }
                        dst[o++] = v;
                    }
}

// (experiment build) three 7x7 boards per workgroup in ten tiles with DENSE images: 147 of 160 tile rows are boards
// instead of 98 of 112, and a workgroup is 7 % faster per board — but a batch of 256 is 86 workgroups, and with the two to
// four launches an executor keeps in flight the chip's 256 CUs are not filled: 458k evals/s with three engines and 517k
// with four against 520k / 520k for two boards (tools/ab_a1_f32.sh).  It would pay from batch 768 on.
bool tower32_dense3_supported(int policy_kind, int extra_moves, int pc, int h, int w, int channels, int hc, int hs, bool heads) {
#ifdef KZ_EXPERIMENTS
    return channels == 128 && h * w == 49 && (!heads || conv_heads_fit(10, policy_kind, extra_moves, pc, h, w, channels, hc, hs));
#else
    return false;
#endif
}

int tower32_boards_per_workgroup(int h, int w, int channels) {
    const int nt = tiles_for(h * w, channels);
    return nt ? nt * 16 / (h * w) : 0;
}

size_t tower32_weight_pad_elems(int channels) { return (size_t)8 * channels * 16; }  // the ring's depth past the end of the stream

size_t tower32_weight_elems(int c_in, int channels, int depth) {
    const int stem_groups = (c_in + 15) / 16;
    return (size_t)9 * stem_groups * 16 * channels + (size_t)2 * depth * 9 * channels * channels;
}

// OIHW f32 (BN folded) -> [tap][group][wave 4][ot C/64][lane 64][4]: element s of lane (fr, kq) is
// W[oc = 16 * (wave * C/64 + ot) + fr][channel][tap], channel = 16 g + 4 kq + s for the stem (short input rows) and
// plane(kq)/4 + 4 g + s for a tower layer (plane: the byte offsets of plane_of<C>)
void tower32_pack_weights(const float *oihw, int cout, int cin, bool stem, float *dst) {
    const int groups = stem ? (cin + 15) / 16 : cin / 16, ot_n = cout / 64;
    size_t o = 0;
    for (int tap = 0; tap < 9; tap++)
        for (int g = 0; g < groups; g++)
            for (int wave = 0; wave < 4; wave++)
                for (int ot = 0; ot < ot_n; ot++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int s = 0; s < 4; s++) {
                            const int fr = lane & 15, kq = lane >> 4;
                            const int oc = 16 * (wave * ot_n + ot) + fr;
                            int ch;
                            if (stem) ch = 16 * g + 4 * kq + s;
                            else {
                                const int plane = cout == 256 ? 256 * kq : 256 * (kq & 1) + 128 * (kq >> 1);
                                ch = plane / 4 + 4 * g + s;
                            }
                            dst[o++] = ch < cin ? oihw[((size_t)oc * cin + ch) * 9 + tap] : 0.0f;
                        }
}

void launch_tower32(const Tower32Args &t, hipStream_t stream) {
    TowerF32Dev d;
    d.x0 = t.x0;
    d.w = static_cast<const f32x4 *>(t.weights);
    d.bias = t.bias;
    d.post_scale = t.post_scale;
    d.post_shift = t.post_shift;
    d.y = t.y;
    d.ldx0 = t.ldx0;
    d.ldy = t.ldy;
    d.batch = t.batch;
    d.h = t.h;
    d.w_ = t.w;
    d.hw = t.h * t.w;
    d.depth = t.depth;
    d.stem_groups = (t.c_in + 15) / 16;
    const int nt = t.dense3 ? 10 : tiles_for(d.hw, t.channels);
    d.nb = nt * 16 / d.hw;
    d.inv_w = (65536u + (unsigned)t.w - 1) / (unsigned)t.w;
    d.inv_hw = (65536u + (unsigned)d.hw - 1) / (unsigned)d.hw;
    d.bits = t.bits;
    d.bits_stride = t.bits_stride;
    d.scalars_in = t.scalars_in;
    d.n_scalar = t.n_scalar;
    d.n_bool = t.n_bool;
    const Tower32Args::Heads &hd = t.heads;
    d.hc = hd.hc; d.hs = hd.hs; d.pc = hd.pc; d.policy_len = hd.policy_len; d.zero_tail = hd.zero_tail; d.epoch = hd.epoch;
    d.sh_b0 = hd.sh_b0; d.sh_w1t = hd.sh_w1t; d.sh_b1 = hd.sh_b1; d.sh_w2 = hd.sh_w2; d.sh_b2 = hd.sh_b2;
    d.p_b1 = hd.p_b1;
    d.extra = hd.extra; d.pe_bc = hd.pe_bc; d.pe_wl = hd.pe_wl; d.pe_bl = hd.pe_bl;
    d.small_w = static_cast<const f32x4 *>(static_cast<const void *>(hd.small_w));
    d.stamps = nullptr;
    d.scalars = hd.scalars; d.policy = hd.policy; d.nonfinite_flag = hd.nonfinite_flag;
    d.dec = DecodeDev{hd.on ? hd.decode.move_offsets : nullptr, hd.decode.move_indices, hd.decode.values, hd.decode.probs,
                      hd.decode.error_flag, hd.policy_len};
    const int grid = (t.batch + d.nb - 1) / d.nb;
    if (t.channels == 256) launch<256, 4>(d, hd.on, grid, stream);
#ifdef KZ_EXPERIMENTS
    else if (nt == 10) launch<128, 10, true>(d, hd.on, grid, stream);
#endif
    else if (nt == 7) launch<128, 7>(d, hd.on, grid, stream);
    else if (nt == 6) launch<128, 6>(d, hd.on, grid, stream);
    else launch<128, 4>(d, hd.on, grid, stream);
}

}  // namespace kz
