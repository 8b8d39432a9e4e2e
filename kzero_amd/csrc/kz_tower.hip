// kz_tower.hip — board-resident ResTower: the whole tower (stem + 2*depth fused 3x3 convolutions + final BN) in ONE
// launch.  Replaces the ~41 cuDNN `cudnnConvolutionBiasActivationForward` launches of the reference's GPU path
// (docs/conv_bn_sm_flow.svg) for 8x8 boards with 256 channels in f16.
//
// Why this shape (MI355X-first):
//  * Boards are independent and an 8x8x256 f16 board is 32 KB, so a workgroup keeps NB boards' residual stream (X) and
//    mid activation (Y) in LDS for the whole tower: activations never touch HBM between layers, there is no launch
//    boundary between layers and no inter-workgroup communication at all.
//  * The only stream is the weights (1.18 MB per layer).  They are read from L2 straight into MFMA A-fragment
//    registers: the host packs them in fragment order, so a wave-instruction is one coalesced 1 KiB global_load_dwordx4
//    and weights never pass through LDS.  Each wave owns 64 output channels (no weight byte is loaded twice per CU).
//    Loads run PF k-steps ahead of the MFMAs, across layer boundaries.
//  * GEMM orientation D[oc][pixel] = W[oc][k] * X[k][pixel]: weights are the MFMA A operand, activations the B
//    operand, so a lane ends up with 4 consecutive channels of one pixel and writes them to the NHWC LDS image with one
//    ds_write_b64.  im2col exists only as LDS addressing: a tap outside the board reads a shared all-zero row.
//  * LDS image: row = pixel (512 B = 256 f16), 16-byte chunk c of row p stored at chunk position c ^ (p & 15), which
//    makes the ds_read_b128 fragment reads of 16 different pixels conflict-free.
#include "kz_kernels.hpp"

namespace kz {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int C = 256;        // tower channels
constexpr int ROW = C * 2;    // bytes per pixel row in LDS
constexpr int KSTEPS = 72;    // 9 taps x 8 chunks of 32 channels
constexpr int PF = 4;         // weight prefetch distance in k-steps (register stages)

struct TowerDev {
    const h16 *x0;
    const uint4 *w_stem, *w_tower;
    const float *bias, *post_scale, *post_shift;
    h16 *y;
    int cin_p, batch, depth;
};

template <int NB>
struct Layout {
    static constexpr int M = NB * 64;
    static constexpr int MT = M / 16;
    static constexpr int X_OFF = 0;
    static constexpr int Y_OFF = M * ROW;
    static constexpr int Z_OFF = 2 * M * ROW;       // 512 zero bytes
    static constexpr int S_OFF = Z_OFF + ROW;       // stem input, rows of 64 B (32 channels)
    static constexpr int BYTES = S_OFF + M * 64;
};

// LDS byte offset of 16-byte chunk c16 of pixel row p
__device__ __forceinline__ int lds_chunk(int p, int c16) { return p * ROW + ((c16 ^ (p & 15)) << 4); }

template <int NB>
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
    const int total_ksteps = layers * KSTEPS;

    // ---- weight stream: per k-step 16 KB = [wave 4][nt 4][lane 64] x 16 B; prime PF stages before anything else ----
    const uint4 *wp = a.w_tower + wave * 256 + lane;
    uint4 wreg[PF][4];
#pragma unroll
    for (int s = 0; s < PF; s++) {
        const int g = s < total_ksteps ? s : total_ksteps - 1;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wreg[s][nt] = wp[(size_t)g * 1024 + nt * 64];
    }

    // ---- zero row and stem input ----
    if (tid < 32) *reinterpret_cast<uint4 *>(lds + L::Z_OFF + tid * 16) = make_uint4(0, 0, 0, 0);
    for (int id = tid; id < M * 4; id += 256) {
        const int row = id >> 2, c = id & 3;
        const int board = board0 + (row >> 6);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (board < a.batch)
            v = *reinterpret_cast<const uint4 *>(a.x0 + ((size_t)board0 * 64 + row) * a.cin_p + c * 8);
        *reinterpret_cast<uint4 *>(lds + L::S_OFF + row * 64 + c * 16) = v;
    }
    __syncthreads();

    f32x4 acc[4][MT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // epilogue: v = acc + bias; [relu]; [+ residual X]; [final BN]; -> f16 -> LDS image at dst_off
    auto epilogue = [&](int layer, int dst_off, bool relu, bool residual, bool post) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int oc = wave * 64 + nt * 16 + kq * 4;
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.bias + layer * C + oc);
            f32x4 ps = f32x4{1.f, 1.f, 1.f, 1.f}, pt = f32x4{0.f, 0.f, 0.f, 0.f};
            if (post) {
                ps = *reinterpret_cast<const f32x4 *>(a.post_scale + oc);
                pt = *reinterpret_cast<const f32x4 *>(a.post_shift + oc);
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int p = mt * 16 + fr;
                const int off = lds_chunk(p, oc >> 3) + (kq & 1) * 8;
                f32x4 v = acc[nt][mt] + bias;
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = fmaxf(v[j], 0.0f);
                }
                if (residual) {
                    const h16x4 rx = *reinterpret_cast<const h16x4 *>(lds + L::X_OFF + off);
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] += (float)rx[j];
                }
                if (post) v = v * ps + pt;
                *reinterpret_cast<h16x4 *>(lds + dst_off + off) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
            }
        }
    };

    // ---- stem: 9 k-steps over the 32 (padded) input channels; conv + bias, no activation (post_act.py:205) ----
    zero_acc();
    for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        h16x8 af[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const uint4 t = a.w_stem[((tap * 4 + wave) * 4 + nt) * 64 + lane];
            af[nt] = *reinterpret_cast<const h16x8 *>(&t);
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int p = mt * 16 + fr;
            const int yy = ((p >> 3) & 7) + dy, xx = (p & 7) + dx;
            const bool ok = (unsigned)yy < 8u && (unsigned)xx < 8u;
            const int off = ok ? L::S_OFF + (p + dy * 8 + dx) * 64 + kq * 16 : L::Z_OFF + kq * 16;
            const h16x8 bf = *reinterpret_cast<const h16x8 *>(lds + off);
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf, acc[nt][mt], 0, 0, 0);
        }
    }
    epilogue(0, L::X_OFF, false, false, a.depth == 0);
    __syncthreads();

    // ---- the 2*depth 3x3 convolutions ----
    // per pixel-tile LDS address of this lane's fragment row for one tap; chunk bits are XORed in per k-step:
    // chunk position (ch*4 + kq) ^ (q & 15) = ((ch ^ (q>>2 & 3)) << 2) | (kq ^ (q & 3))
    auto tap_rows = [&](int tap, int src_off, int (&T)[MT]) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int p = mt * 16 + fr;
            const int yy = ((p >> 3) & 7) + dy, xx = (p & 7) + dx;
            const bool ok = (unsigned)yy < 8u && (unsigned)xx < 8u;
            const int q = p + dy * 8 + dx;
            const int t_ok = src_off + q * ROW + (((q >> 2) & 3) << 6) + ((kq ^ (q & 3)) << 4);
            T[mt] = ok ? t_ok : L::Z_OFF + (kq << 4);
        }
    };

    int g = 0;  // global k-step index into the weight stream
    for (int layer = 1; layer <= layers; layer++) {
        const bool is_b = (layer & 1) == 0;  // conv A: X -> Y; conv B: Y -> X (+ residual)
        const int src_off = is_b ? L::Y_OFF : L::X_OFF;
        zero_acc();
        int T[MT], Tn[MT];
        h16x8 bf[2][MT];  // activation fragments, double buffered one k-step ahead
        tap_rows(0, src_off, T);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bf[0][mt] = *reinterpret_cast<const h16x8 *>(lds + T[mt]);
        for (int tap = 0; tap < 9; tap++) {
            tap_rows(tap < 8 ? tap + 1 : 8, src_off, Tn);
#pragma unroll
            for (int ch = 0; ch < 8; ch++) {
                const int stage = ch & (PF - 1), cur = ch & 1, nxt = cur ^ 1;
                // (1) next k-step's activation fragments: LDS -> registers
                if (ch < 7) {
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
                        bf[nxt][mt] = *reinterpret_cast<const h16x8 *>(lds + (T[mt] ^ ((ch + 1) << 6)));
                } else if (tap < 8) {
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) bf[nxt][mt] = *reinterpret_cast<const h16x8 *>(lds + Tn[mt]);
                }
                // (2) this k-step's weight fragments were loaded PF k-steps ago; refill the stage with k-step g + PF
                //     (clamped at the end of the stream; the surplus loads are never used)
                h16x8 af[4];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) af[nt] = *reinterpret_cast<const h16x8 *>(&wreg[stage][nt]);
                {
                    const int gn = g + PF < total_ksteps ? g + PF : total_ksteps - 1;
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) wreg[stage][nt] = wp[(size_t)gn * 1024 + nt * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                // (3) 4 x MT MFMAs on independent accumulators
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < 4; nt++)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf[cur][mt], acc[nt][mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                g++;
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) T[mt] = Tn[mt];
        }
        epilogue(layer, is_b ? L::X_OFF : L::Y_OFF, true, is_b, layer == layers);
        __syncthreads();
    }

    // ---- write the tower output: un-swizzle, coalesced 16-byte stores ----
    for (int id = tid; id < M * 32; id += 256) {
        const int p = id >> 5, c16 = id & 31;
        const int board = board0 + (p >> 6);
        if (board < a.batch) {
            const uint4 v = *reinterpret_cast<const uint4 *>(lds + L::X_OFF + lds_chunk(p, c16));
            *reinterpret_cast<uint4 *>(a.y + ((size_t)board0 * 64 + p) * C + c16 * 8) = v;
        }
    }
}

int boards_per_wg() {
    static int nb = [] {
        const char *e = getenv("KZ_TOWER_NB");
        int v = e ? atoi(e) : 2;
        return v == 1 ? 1 : 2;
    }();
    return nb;
}

}  // namespace

bool tower_resident_supported(int dtype, int h, int w, int channels, int depth) {
    return dtype == 1 && h == 8 && w == 8 && channels == C && depth >= 1;
}

size_t tower_packed_weight_elems(int cin_p, int depth) {
    return (size_t)9 * C * cin_p + (size_t)2 * depth * 9 * C * C;
}

// OIHW f32 -> [tap 9][chunk cin_p/32][wave 4][nt 4][lane 64][8] f16: element j of lane (fr, kq) of (wave, nt) is
// W[oc = 64*wave + 16*nt + fr][channel = 32*chunk + 8*kq + j][tap] — the A fragment of v_mfma_f32_16x16x32_f16.
void tower_pack_weights(const float *oihw, int cout, int cin, int cin_p, uint16_t *dst) {
    const int nchunk = cin_p / 32;
    for (int tap = 0; tap < 9; tap++)
        for (int chunk = 0; chunk < nchunk; chunk++)
            for (int wave = 0; wave < 4; wave++)
                for (int nt = 0; nt < 4; nt++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int oc = 64 * wave + 16 * nt + (lane & 15);
                            const int ch = 32 * chunk + 8 * (lane >> 4) + j;
                            float v = 0.0f;
                            if (oc < cout && ch < cin) v = oihw[((size_t)oc * cin + ch) * 9 + tap];
                            const _Float16 hv = (_Float16)v;
                            uint16_t bits;
                            __builtin_memcpy(&bits, &hv, 2);
                            dst[(((((size_t)tap * nchunk + chunk) * 4 + wave) * 4 + nt) * 64 + lane) * 8 + j] = bits;
                        }
}

void launch_tower_resident(const TowerArgs &t, hipStream_t stream) {
    TowerDev d;
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
    static bool attr_done = [] {
        (void)hipFuncSetAttribute((const void *)kz_tower_resident<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  Layout<1>::BYTES);
        (void)hipFuncSetAttribute((const void *)kz_tower_resident<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  Layout<2>::BYTES);
        return true;
    }();
    (void)attr_done;
    if (boards_per_wg() == 1) {
        kz_tower_resident<1><<<t.batch, 256, Layout<1>::BYTES, stream>>>(d);
    } else {
        kz_tower_resident<2><<<(t.batch + 1) / 2, 256, Layout<2>::BYTES, stream>>>(d);
    }
}

}  // namespace kz
