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
//   epilogue  = bias-initialised accumulators (the next layer's bias is fetched a layer ahead); [ReLU]; [+ residual, in place in the out image]; [final BN]; 16-byte
//               stores (4 consecutive output channels of a pixel row per lane)
//
// Arithmetic follows python/lib/model/post_act.py:201-239 with Conv+BN folded on the host (kz_model.cpp).
#include "kz_kernels.hpp"

namespace kz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct TowerF32Dev {
    const float *x0;    // encoded input [batch*hw][ldx0]
    const f32x4 *w;     // fragment-packed: stem steps, then 2*depth layers of 9*C/16 steps (tower32_pack_weights)
    const float *bias;  // [1 + 2*depth][C]
    const float *post_scale, *post_shift;  // final BN [C]
    float *y;           // tower output [batch*hw][ldy]
    int ldx0, ldy, batch, h, w_, hw, depth, stem_groups, nb;
    unsigned inv_w, inv_hw;  // ceil(65536 / w), ceil(65536 / hw): exact quotients for values < 512
};

template <int C>
__device__ __forceinline__ int plane_of(int kq) {  // byte offset of lane group kq's 16-byte piece within a step
    return C == 256 ? 256 * kq : 256 * (kq & 1) + 128 * (kq >> 1);
}

template <int C, int NT>
__global__ __launch_bounds__(256, 1) void kz_tower_resident_f32(TowerF32Dev a) {
    constexpr int OT = C / 64;           // 16-channel output tiles per wave
    constexpr int G = C / 16;            // steps per tap in a tower layer
    constexpr int ROWS = NT * 16;
    constexpr int RS = C * 4 + 16;       // row stride: odd number of 16-byte slots
    constexpr int ZERO = 0, IMG0 = 16 * RS, IMG1 = IMG0 + ROWS * RS;
    constexpr int STEP = 4 * OT * 64;    // f32x4 per step: [wave][ot][lane]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int board0 = blockIdx.x * a.nb;
    const int boards = min(a.nb, a.batch - board0);
    const int rows_valid = boards * a.hw;

    // zero rows; padding rows of image 0 (their outputs are never stored, but keep them finite)
    for (int i = tid; i < 16 * RS / 16; i += 256) *reinterpret_cast<f32x4 *>(lds + ZERO + i * 16) = f32x4{0, 0, 0, 0};
    // stage the encoded input: stem_groups * 16 channels per row
    {
        const int pieces = a.stem_groups * 4;  // 16-byte pieces per row
        for (int i = tid; i < ROWS * pieces; i += 256) {
            const int r = i / pieces, p = i - r * pieces;
            f32x4 v = f32x4{0, 0, 0, 0};
            if (r < rows_valid) v = *reinterpret_cast<const f32x4 *>(a.x0 + ((size_t)board0 * a.hw + r) * a.ldx0 + p * 4);
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
        wl += (size_t)9 * a.stem_groups * STEP;
        epilogue(false, false, a.depth == 0);
        __syncthreads();
        in = IMG1;
        out = IMG0;
    }

    // ---- 2*depth tower convolutions: 9 G steps each.  Weights run FOUR steps ahead through a register ring (a step is
    // 4*OT*NT MFMAs of 32 cycles = 0.75 us at A1: two steps of cover were less than an L2 round trip under load), the
    // activation fragments one step ahead in the other of two buffers; inside a step every fragment read and ring refill
    // is placed in the shadow of MFMAs (24 free issue cycles per f32 MFMA) instead of in a burst between two blocks of
    // MFMAs.  Round 2, A1 at a full chip: MFMA-busy 83.7 % -> see DESIGN.md §5.2a. ----
    const int koff = plane_of<C>(kq);
    constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100;
    constexpr int S = 9 * G;
    static_assert(S % 4 == 0, "the ring stage of a step is a compile-time constant");
    // the ring runs across layer boundaries (the stream is contiguous): step index ts counts from the first tower layer
    const int total_steps = 2 * a.depth * S;
    int ts = 0;
    f32x4 wring[4][OT];
    auto issue_w = [&](f32x4(&wf)[OT], int t) {  // t: step of the whole tower, clamped at the end of the stream
        const f32x4 *p = wl + (size_t)(t < total_steps ? t : total_steps - 1) * STEP;
#pragma unroll
        for (int ot = 0; ot < OT; ot++) wf[ot] = p[ot * 64];
    };
#pragma unroll
    for (int u = 0; u < 4; u++) issue_w(wring[u], u);
    // the bias of the next layer is fetched while this one computes
    f32x4 bias_next[OT];
    auto fetch_bias = [&](int layer) {
        const int l = layer <= 2 * a.depth ? layer : 2 * a.depth;
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
            bias_next[ot] = *reinterpret_cast<const f32x4 *>(a.bias + (size_t)l * C + (wave * OT + ot) * 16 + kq * 4);
    };
    fetch_bias(1);
    for (int layer = 1; layer <= 2 * a.depth; layer++) {
        f32x4 bA[NT], bB[NT];
        auto issue_b = [&](f32x4(&bf)[NT], int t) {
            const int tap = t / G, g = t % G;
            if (g == 0) tap_rows(tap, koff);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) bf[nt] = *reinterpret_cast<const f32x4 *>(lds + T[nt] + g * 16);
        };
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[ot][nt] = bias_next[ot];
        fetch_bias(layer + 1);
        issue_b(bA, 0);
#pragma unroll 1
        for (int t = 0; t < S; t += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int tn = t + u + 1 < S ? t + u + 1 : S - 1, tw = ts + 4;
                if (u & 1) {
                    issue_b(bA, tn);
                    mfma_step(wring[u], bB);
                } else {
                    issue_b(bB, tn);
                    mfma_step(wring[u], bA);
                }
                issue_w(wring[u], tw);  // (this stage's fragments have been issued to the MFMAs)
                ts++;
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
        const bool second = (layer & 1) == 0;  // the block's second conv: residual add, and the final BN on the last
        epilogue(true, second, second && layer == 2 * a.depth);
        __syncthreads();
        const int tmp = in;
        in = out;
        out = tmp;
    }

    // ---- the tower's output (the last `out`, now `in`) -> global, 16 bytes per lane, whole rows ----
    for (int i = tid; i < rows_valid * (C / 4); i += 256) {
        const int r = i / (C / 4), p = i - r * (C / 4);
        *reinterpret_cast<f32x4 *>(a.y + ((size_t)board0 * a.hw + r) * a.ldy + p * 4) =
            *reinterpret_cast<const f32x4 *>(lds + in + r * RS + p * 16);
    }
}

int tiles_for(int hw, int channels) {
    if (channels == 256) return hw <= 64 ? 4 : 0;
    if (channels == 128) return hw * 2 <= 112 ? 7 : hw <= 64 ? 4 : hw <= 96 ? 6 : 0;
    return 0;
}

template <int C, int NT>
void launch(const TowerF32Dev &d, int grid, hipStream_t stream) {
    constexpr int LDS_BYTES = (16 + 2 * NT * 16) * (C * 4 + 16);
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_tower_resident_f32<C, NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES);
        done_mask |= 1ull << (dev & 63);
    }
    kz_tower_resident_f32<C, NT><<<grid, 256, LDS_BYTES, stream>>>(d);
}

}  // namespace

bool tower32_supported(int dtype, int h, int w, int channels, int depth) {
    return dtype == 0 && depth >= 1 && h >= 2 && w >= 2 && w <= 32 && tiles_for(h * w, channels) != 0;
}

int tower32_boards_per_workgroup(int h, int w, int channels) {
    const int nt = tiles_for(h * w, channels);
    return nt ? nt * 16 / (h * w) : 0;
}

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
    const int nt = tiles_for(d.hw, t.channels);
    d.nb = nt * 16 / d.hw;
    d.inv_w = (65536u + (unsigned)t.w - 1) / (unsigned)t.w;
    d.inv_hw = (65536u + (unsigned)d.hw - 1) / (unsigned)d.hw;
    const int grid = (t.batch + d.nb - 1) / d.nb;
    if (t.channels == 256) launch<256, 4>(d, grid, stream);
    else if (nt == 7) launch<128, 7>(d, grid, stream);
    else if (nt == 6) launch<128, 6>(d, grid, stream);
    else launch<128, 4>(d, grid, stream);
}

}  // namespace kz
