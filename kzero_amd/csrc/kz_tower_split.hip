// kz_tower_split.hip — board-resident ResTower with f32-EQUIVALENT results on the f16 matrix cores: every activation and
// every weight is carried as a pair of f16 values (hi = f16(v), lo = f16(v - hi): 22 significant bits) and every product
// is three MFMAs, hi*hi + hi*lo + lo*hi, accumulated in f32 (the lo*lo term is below 2^-22 of the product).  Same
// organisation as the f16 launch (kz_tower.hip) at one board per workgroup: the residual stream X and the mid activation
// Y live in LDS for the whole tower — as two images each, hi and lo — and the weights stream from L2 straight into MFMA
// A-fragment registers, 32 KB per k-step (hi fragments, then lo fragments).
//
// Why: the exact-f32 launch (kz_tower_f32.hip, v_mfma_f32_16x16x4_f32) is bound by the f32 MFMA rate, 157 TFLOP/s; three
// f16 MFMAs per product run at 2500 / 3 = 833 TFLOP/s.  The results agree with the CPU oracle within the same 1e-4 as
// the exact-f32 path (tests/test_gpu_parity.py), which the plain f16 path cannot (it rounds the residual stream to 11
// bits per layer).  8x8 boards, 256 channels; input and output are the f32 tensors of the f32 engine path, so the
// encode kernel in front and the generic f32 head kernels behind are unchanged.
//
// Arithmetic follows python/lib/model/post_act.py:201-239 with Conv+BN folded on the host (kz_model.cpp).
#include <vector>

#include "kz_kernels.hpp"

namespace kz {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int C = 256;
constexpr int RS = C * 2 + 16;  // LDS bytes per pixel row (see kz_tower.hip)
constexpr int KSTEPS = 72;      // 9 taps x 8 chunks of 32 channels
constexpr int MT = 4;           // 16-pixel tiles of the board
constexpr int PF = 4;           // weight ring depth in k-steps (a k-step is 48 MFMAs = 768 cycles)
constexpr int IMG = 64 * RS;
// hi block = [X][Y][16 all-zero rows], lo block = the same DELTA bytes later: one address array serves both images of a
// pair (lo = hi + DELTA, zero rows included), and DELTA is a multiple of 256 B so the bank pattern is the same
constexpr int XH = 0, YH = IMG, ZH = 2 * IMG, DELTA = 2 * IMG + 16 * RS;
constexpr int XL = XH + DELTA, ZL = ZH + DELTA;
static_assert(DELTA % 256 == 0 && ZH % 256 == 0, "slot pattern");
constexpr int SH = 2 * DELTA, SL = SH + 64 * 64;  // stem input, rows of 64 B (32 channels)
constexpr int LDS_BYTES = SL + 64 * 64;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100;

struct SplitDev {
    const float *x0;    // encoded input [batch*64][ldx0] f32
    const uint4 *w;     // k-steps of [hi | lo][wave 4][nt 4][lane 64] x 16 B: 9 stem k-steps, then 2*depth*72
    const float *bias;  // [1 + 2*depth][256]
    const float *post_scale, *post_shift;
    float *y;           // tower output [batch*64][ldy] f32
    int ldx0, c_in, ldy, batch, depth;
};

__device__ __forceinline__ void split4(f32x4 v, h16x4 &hi, h16x4 &lo) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        hi[j] = (h16)v[j];
        lo[j] = (h16)(v[j] - (float)hi[j]);
    }
}

__global__ __launch_bounds__(256, 1) void kz_tower_resident_split(SplitDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int board = blockIdx.x;
    const int layers = 2 * a.depth;
    const int total_ksteps = layers * KSTEPS;  // of the ring: the 9 stem k-steps in front of them are read directly

    // ---- weight stream: prime PF stages (stage s = k-step g % PF) ----
    const uint4 *wp_stem = a.w + wave * 256 + lane;
    const uint4 *wp = wp_stem + (size_t)9 * 2048;
    auto wload = [&](int gk, int part, int nt) __attribute__((always_inline)) {
        return wp[(size_t)gk * 2048 + part * 1024 + nt * 64];
    };
    uint4 wreg[PF][2][4];
#pragma unroll
    for (int s = 0; s < PF; s++)
#pragma unroll
        for (int part = 0; part < 2; part++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) wreg[s][part][nt] = wload(s < total_ksteps ? s : total_ksteps - 1, part, nt);
    int g = 0;
    auto ring_take = [&](int stage, h16x8 (&ah)[4], h16x8 (&al)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            ah[nt] = *reinterpret_cast<const h16x8 *>(&wreg[stage][0][nt]);
            al[nt] = *reinterpret_cast<const h16x8 *>(&wreg[stage][1][nt]);
        }
        const int gn = g + PF < total_ksteps ? g + PF : total_ksteps - 1;
#pragma unroll
        for (int part = 0; part < 2; part++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) wreg[stage][part][nt] = wload(gn, part, nt);
    };

    // ---- zero rows and the stem input (f32 -> hi/lo, 32 channels per square) ----
    for (int id = tid; id < 16 * RS / 16; id += 256) {
        *reinterpret_cast<uint4 *>(lds + ZH + id * 16) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(lds + ZL + id * 16) = make_uint4(0, 0, 0, 0);
    }
    for (int id = tid; id < 64 * 8; id += 256) {  // (square, 4-channel piece)
        const int row = id >> 3, c4 = id & 7;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 * 4 < a.ldx0) v = *reinterpret_cast<const f32x4 *>(a.x0 + ((size_t)board * 64 + row) * a.ldx0 + c4 * 4);
        h16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<h16x4 *>(lds + SH + row * 64 + c4 * 8) = hi;
        *reinterpret_cast<h16x4 *>(lds + SL + row * 64 + c4 * 8) = lo;
    }
    __syncthreads();

    f32x4 acc[4][MT];
    f32x4 bias_next[4];
    auto fetch_bias = [&](int row) __attribute__((always_inline)) {
        const int l = row <= layers ? row : layers;
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
    f32x4 post_s[4], post_t[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        post_s[nt] = f32x4{1.f, 1.f, 1.f, 1.f};
        post_t[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // three MFMAs per (output tile, pixel tile): hi*hi + hi*lo + lo*hi
    auto mfma3 = [&](const h16x8 (&ah)[4], const h16x8 (&al)[4], const h16x8 (&bh)[MT], const h16x8 (&bl)[MT])
                     __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[nt], bh[mt], acc[nt][mt], 0, 0, 0);
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nt], bl[mt], acc[nt][mt], 0, 0, 0);
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nt], bh[mt], acc[nt][mt], 0, 0, 0);
            }
    };

    // tap validity per lane: the pixel of tile row fr is (y = 2*mt + (fr>>3), x = fr&7)
    const bool x_is0 = (lane & 7) == 0, x_is7 = (lane & 7) == 7, yo_is0 = (lane & 8) == 0, yo_is1 = !yo_is0;
    auto tap_ok = [&](int mt, int dy, int dx) __attribute__((always_inline)) {
        const bool kill_x = (dx < 0 && x_is0) || (dx > 0 && x_is7);
        return !(kill_x || (mt == 0 && dy < 0 && yo_is0) || (mt == 3 && dy > 0 && yo_is1));
    };

    // ---- stem: 9 k-steps over the 32 (padded) input channels; conv + bias, no activation (post_act.py:205) ----
    fetch_bias(0);
    init_acc();
    fetch_bias(1);
#pragma nounroll
    for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        h16x8 ah[4], al[4], bh[MT], bl[MT];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const uint4 th = wp_stem[(size_t)tap * 2048 + nt * 64], tl = wp_stem[(size_t)tap * 2048 + 1024 + nt * 64];
            ah[nt] = *reinterpret_cast<const h16x8 *>(&th);
            al[nt] = *reinterpret_cast<const h16x8 *>(&tl);
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int p = mt * 16 + fr;
            const bool ok = tap_ok(mt, dy, dx);
            const int off = ok ? (p + dy * 8 + dx) * 64 + kq * 16 : -1;  // stem: natural k (channel = 8 kq + j)
            bh[mt] = off >= 0 ? *reinterpret_cast<const h16x8 *>(lds + SH + off) : h16x8{};
            bl[mt] = off >= 0 ? *reinterpret_cast<const h16x8 *>(lds + SL + off) : h16x8{};
        }
        mfma3(ah, al, bh, bl);
    }

    const int lane_row = fr * RS;
    const int epi_base = lane_row + (wave * 64 + kq * 4) * 2;
    // epilogue: [relu]; [+ residual X]; -> (hi, lo) -> the two LDS images at dst
    auto epilogue = [&](int dst_h, bool relu, bool residual) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int off = epi_base + mt * 16 * RS + nt * 32;
                f32x4 v = acc[nt][mt];
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                }
                if (residual) {  // added in f32, AFTER the ReLU (post_act.py:227-228)
                    const h16x4 rh = *reinterpret_cast<const h16x4 *>(lds + XH + off);
                    const h16x4 rl = *reinterpret_cast<const h16x4 *>(lds + XL + off);
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] += (float)rh[j] + (float)rl[j];
                }
                h16x4 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<h16x4 *>(lds + dst_h + off) = hi;
                *reinterpret_cast<h16x4 *>(lds + dst_h + DELTA + off) = lo;
            }
    };
    epilogue(XH, false, false);
    __syncthreads();

    // ---- convolution passes over the LDS images (channel assignment of a k-step as in kz_tower.hip) ----
    const int kq_off = 256 * (kq & 1) + 128 * (kq >> 1);
    const int frag_base = lane_row + kq_off;
    // T[mt] = LDS address, in the hi block, of this lane's fragment row (pixel shifted by the tap) or of a zero row
    auto tap_rows = [&](int tap, int src_h, int (&T)[MT]) __attribute__((always_inline)) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int shifted = src_h + frag_base + (dy * 8 + dx) * RS;
        const int zrow = ZH + ((fr + dy * 8 + dx) & 15) * RS + kq_off;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) T[mt] = tap_ok(mt, dy, dx) ? shifted + mt * 16 * RS : zrow;
    };
    auto conv_3x3 = [&](int src_h) __attribute__((always_inline)) {
        int T[MT], Tn[MT];
        h16x8 bh[2][MT], bl[2][MT];
        tap_rows(0, src_h, T);
        auto rd = [&](int t, int extra) __attribute__((always_inline)) { return *reinterpret_cast<const h16x8 *>(lds + t + extra); };
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            bh[0][mt] = rd(T[mt], 0);
            bl[0][mt] = rd(T[mt], DELTA);
        }
#pragma nounroll
        for (int tap = 0; tap < 9; tap++) {
            tap_rows(tap + 1 < 9 ? tap + 1 : tap, src_h, Tn);
#pragma unroll
            for (int ch = 0; ch < 8; ch++) {
                const int stage = ch & (PF - 1), cur = ch & 1, nxt = cur ^ 1;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    bh[nxt][mt] = ch < 7 ? rd(T[mt], (ch + 1) * 16) : rd(Tn[mt], 0);
                    bl[nxt][mt] = ch < 7 ? rd(T[mt], DELTA + (ch + 1) * 16) : rd(Tn[mt], DELTA);
                }
                h16x8 ah[4], al[4];
                ring_take(stage, ah, al);
                mfma3(ah, al, bh[cur], bl[cur]);
                // every memory instruction in the shadow of an MFMA: the 8 ring refills, the 8 fragment reads, then
                // the remaining MFMAs back to back
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 2 * MT; i++) {
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 3 * 4 * MT - 8 - 2 * MT, 0);
                __builtin_amdgcn_sched_barrier(0);
                g++;
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) T[mt] = Tn[mt];
        }
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
        conv_3x3(is_b ? YH : XH);
        if (!is_b) {
            epilogue(YH, true, false);
        } else if (layer != layers) {
            epilogue(XH, true, true);
        } else {
            // last layer: ReLU, residual, final BN -> f32 rows of the tower output in global memory
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const int off = epi_base + mt * 16 * RS + nt * 32;
                    f32x4 v = acc[nt][mt];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                    const h16x4 rh = *reinterpret_cast<const h16x4 *>(lds + XH + off);
                    const h16x4 rl = *reinterpret_cast<const h16x4 *>(lds + XL + off);
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] += (float)rh[j] + (float)rl[j];
                    v = v * post_s[nt] + post_t[nt];
                    *reinterpret_cast<f32x4 *>(a.y + ((size_t)board * 64 + mt * 16 + fr) * a.ldy + wave * 64 + nt * 16 + kq * 4) = v;
                }
        }
        __syncthreads();
    }
}

}  // namespace

bool tower_split_supported(int h, int w, int channels, int depth, int c_in) {
    return h == 8 && w == 8 && channels == C && depth >= 1 && c_in <= 32;
}

size_t tower_split_weight_elems(int depth) { return ((size_t)9 + (size_t)2 * depth * KSTEPS) * 2 * 8192; }  // f16 elements

// OIHW f32 (BN folded) -> k-steps of [hi | lo][wave 4][nt 4][lane 64][8] f16; element j of lane (fr, kq) of (wave, nt) is
// W[oc = 64*wave + 16*nt + fr][channel][tap], channel = 8*chunk + {0,128,64,192}[kq] + j for a tower layer (one k-step
// per tap and chunk) and 8*kq + j for the stem (one k-step per tap, 32 padded input channels).
void tower_split_pack_weights(const float *oihw, int cout, int cin, bool stem, uint16_t *dst) {
    static const int kq_base[4] = {0, 128, 64, 192};
    const int nchunk = stem ? 1 : 8;
    for (int tap = 0; tap < 9; tap++)
        for (int chunk = 0; chunk < nchunk; chunk++) {
            uint16_t *step = dst + ((size_t)tap * nchunk + chunk) * 2 * 8192;
            for (int wave = 0; wave < 4; wave++)
                for (int nt = 0; nt < 4; nt++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int oc = 64 * wave + 16 * nt + (lane & 15);
                            const int kq = lane >> 4;
                            const int ch = stem ? 8 * kq + j : 8 * chunk + kq_base[kq] + j;
                            float v = 0.0f;
                            if (oc < cout && ch < cin) v = oihw[((size_t)oc * cin + ch) * 9 + tap];
                            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                            uint16_t hb, lb;
                            __builtin_memcpy(&hb, &hi, 2);
                            __builtin_memcpy(&lb, &lo, 2);
                            const size_t e = (((size_t)wave * 4 + nt) * 64 + lane) * 8 + j;
                            step[e] = hb;
                            step[8192 + e] = lb;
                        }
        }
}

void launch_tower_split(const Tower32Args &t, hipStream_t stream) {
    SplitDev d{};
    d.x0 = t.x0;
    d.ldx0 = t.ldx0;
    d.c_in = t.c_in;
    d.w = static_cast<const uint4 *>(t.weights);
    d.bias = t.bias;
    d.post_scale = t.post_scale;
    d.post_shift = t.post_shift;
    d.y = t.y;
    d.ldy = t.ldy;
    d.batch = t.batch;
    d.depth = t.depth;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_tower_resident_split, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        done_mask |= 1ull << (dev & 63);
    }
    kz_tower_resident_split<<<t.batch, 256, LDS_BYTES, stream>>>(d);
}

}  // namespace kz
