// kz_board_conv.hip — per-layer 3x3 convolution for boards whose whole tower does not fit in LDS (Go 19x19: a
// 361 x 256 f16 board is 185 KB), f16, channels a multiple of 128.
//
// Same inner loop as the resident tower (kz_tower.hip): a workgroup keeps whole boards' pixels in LDS and the 9 taps
// are 9 shifted views of that image (a tap outside the board reads a zero row), so the activations are fetched from
// L2/HBM ONCE per layer instead of once per tap; the weights stream from L2 straight into MFMA A-fragment registers in
// host-packed fragment order.  What differs from the tower: one launch per layer (activations round-trip through HBM:
// 2 x 94.6 MB per layer at Go B=512 — 0.4 TB/s at the kernel's MFMA-bound pace), the board is staged in two
// 128-channel chunks (a 384-row x 128-channel image is 110 KB), and a workgroup owns 128 of the output channels.
//
//   grid  = (ceil(boards / bpw), C / 128);  256 threads = 4 waves, wave w owns output channels [32w, 32w + 32)
//   rows  = 24 tiles of 16 pixel rows: bpw boards, each padded to tpb = ceil(h*w / 16) tiles (Go: 23 tiles, bpw = 1)
//   LDS   = two planes (channels [0,64) and [64,128) of the chunk) of 384 rows x 144 B: the two lane groups that share
//           a ds_read_b128 bank group read the two planes at the same row offset, the planes are a multiple of 256 B
//           apart and rows advance by 9 sixteen-byte slots -> conflict-free fragment reads for every tap.
#include <type_traits>

#include "kz_kernels.hpp"

namespace kz {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MT = 24;                 // 16-row tiles per workgroup
constexpr int ROWS = MT * 16;          // 384
constexpr int PRS = 64 * 2 + 16;       // plane row stride: 64 channels + 16 B pad = 144 B
constexpr int PLANE = (ROWS + 16) * PRS;  // 400 rows (384 + 16 zero rows) = 57,600 B = 225 * 256
constexpr int ZROW = ROWS;             // first zero row
constexpr int LDS_BYTES = 2 * PLANE;   // 115,200 B
constexpr int PF = 4;                  // weight ring depth in k-steps (a k-step is 48 MFMAs = 768 cycles)
static_assert(PLANE % 256 == 0, "planes must be a whole number of bank rows apart");

constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100;

struct BoardConvDev {
    const h16 *x;       // [boards*hw][ldx]
    const uint4 *w;     // fragment-packed: [n_half][k-step][wave 4][nt 2][lane 64] x 16 B
    const float *bias, *post_scale, *post_shift;  // [cout]
    const h16 *res;     // optional residual [boards*hw][ldy]
    h16 *y;             // [boards*hw][ldy]
    int ldx, ldy, boards, h, w_, hw, tpb, bpw, cin, relu;
    unsigned inv_w;  // ceil(65536 / w): q / w == (q * inv_w) >> 16 for q < 512, w <= 32
};

__global__ __launch_bounds__(256, 1) void kz_board_conv_f16(BoardConvDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int board0 = blockIdx.x * a.bpw;
    const int nhalf = blockIdx.y;
    const int chunks = a.cin / 128;
    const int total_ksteps = chunks * 36;  // 9 taps x 4 k-steps of 32 channels per chunk

    // weight ring: k-step g of this (layer, n-half): 8 KB = [wave][nt 2][lane] x 16 B
    const uint4 *wp = a.w + (size_t)nhalf * total_ksteps * 512 + wave * 128 + lane;
    uint4 wreg[PF][2];
#pragma unroll
    for (int s = 0; s < PF; s++) {
        const int g = s < total_ksteps ? s : total_ksteps - 1;
        wreg[s][0] = wp[(size_t)g * 512];
        wreg[s][1] = wp[(size_t)g * 512 + 64];
    }
    int g = 0;

    // zero rows of both planes
    for (int id = tid; id < 2 * 16 * PRS / 16; id += 256) {
        const int plane = id / (16 * PRS / 16), off = id % (16 * PRS / 16);
        *reinterpret_cast<uint4 *>(lds + plane * PLANE + ZROW * PRS + off * 16) = make_uint4(0, 0, 0, 0);
    }

    // Validity of (tile row, tap) as bitmasks: bit mt of okmask[tap] says that, for this lane's row of tile mt, the tap
    // lands on the board; bit mt of rowmask says the row is a real pixel of a board of this batch.  Computed once,
    // so a tap costs 3 VALU per tile in the k-loop instead of a dozen.
    unsigned rowmask = 0;
    unsigned okmask[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int b = mt / a.tpb, t = mt - b * a.tpb, q = t * 16 + fr;
        const unsigned valid = b < a.bpw && board0 + b < a.boards && q < a.hw;
        const int yy = (int)(((unsigned)q * a.inv_w) >> 16), xx = q - yy * a.w_;
        const unsigned ym[3] = {(unsigned)(yy >= 1), 1u, (unsigned)(yy <= a.h - 2)};
        const unsigned xm[3] = {(unsigned)(xx >= 1), 1u, (unsigned)(xx <= a.w_ - 2)};
        rowmask |= valid << mt;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) okmask[tap] |= (valid & ym[tap / 3] & xm[tap % 3]) << mt;
    }

    f32x4 acc[2][MT];
    {
        const int oc = nhalf * 128 + wave * 32 + kq * 4;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + oc + nt * 16);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[nt][mt] = b;
        }
    }

    // fragment address pieces that do not depend on the tap: plane (kq & 1), 16-byte piece (kq >> 1) of the k-step
    const int lane_off = (kq & 1) * PLANE + (kq >> 1) * 16;

    // LDS address of this lane's fragment row per tile for one tap (pixel shifted by the tap, or a zero row)
    auto tap_rows = [&](int tap, int lo, int hi, int (&T)[MT]) {
        const int shift = (tap / 3 - 1) * a.w_ + (tap % 3 - 1);
        const unsigned ok = tap == 0   ? okmask[0] : tap == 1 ? okmask[1] : tap == 2 ? okmask[2] : tap == 3 ? okmask[3]
                            : tap == 4 ? okmask[4] : tap == 5 ? okmask[5] : tap == 6 ? okmask[6] : tap == 7 ? okmask[7]
                                                                                                           : okmask[8];
        const int shifted = (fr + shift) * PRS + lane_off;                       // + mt * 16 * PRS per tile
        const int zero = (ZROW + ((fr + shift) & 15)) * PRS + lane_off;          // same 16-byte slot pattern
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            if (mt < lo || mt >= hi) continue;
            T[mt] = ((ok >> mt) & 1) ? shifted + mt * 16 * PRS : zero;
        }
    };

    for (int chunk = 0; chunk < chunks; chunk++) {
        // ---- stage channels [128*chunk, +128) of this workgroup's boards: 16 pieces of 16 B per pixel row ----
        __syncthreads();  // everyone is done reading the previous chunk
        // 24 pieces per thread, in two batches of 12 loads in flight (a load-wait-store per piece would serialise
        // 24 HBM/L2 round trips; the fragment registers are dead here, so the batch is free)
#pragma unroll 1
        for (int part = 0; part < 2; part++) {
            uint4 v[12];
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const int id = tid + (part * 12 + i) * 256;
                const int row = id >> 4, piece = id & 15;
                const int mt = row >> 4, b = mt / a.tpb, q = (mt - b * a.tpb) * 16 + (row & 15);
                // unconditional load from a clamped address + select: a branch around the load would make the compiler
                // wait for vmcnt(0) per element
                const bool ok = b < a.bpw && board0 + b < a.boards && q < a.hw;
                const size_t src_row = ok ? (size_t)(board0 + b) * a.hw + q : 0;
                const uint4 ld = *reinterpret_cast<const uint4 *>(a.x + src_row * a.ldx + chunk * 128 + piece * 8);
                v[i] = ok ? ld : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const int id = tid + (part * 12 + i) * 256;
                const int row = id >> 4, piece = id & 15;
                *reinterpret_cast<uint4 *>(lds + (piece >> 3) * PLANE + row * PRS + (piece & 7) * 16) = v[i];
            }
        }
        __syncthreads();

        // Two half-steps per k-step: the MFMAs of tiles 0..11 run while the fragments of tiles 12..23 are read, and
        // vice versa (the fragments of the NEXT k-step's first half), so every LDS read has ~400 cycles of cover.
        int T[MT];
        h16x8 bfA[MT / 2], bfB[MT / 2];
        tap_rows(0, 0, MT, T);
#pragma unroll
        for (int i = 0; i < MT / 2; i++) bfA[i] = *reinterpret_cast<const h16x8 *>(lds + T[i]);
        for (int tap = 0; tap < 9; tap++) {
            const int next_tap = tap < 8 ? tap + 1 : 8;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const int stage = ks & (PF - 1);  // 36 k-steps per chunk, 4 per tap: g % 4 == ks
                // ---- half 1 ----
#pragma unroll
                for (int i = 0; i < MT / 2; i++)
                    bfB[i] = *reinterpret_cast<const h16x8 *>(lds + T[MT / 2 + i] + ks * 32);
                h16x8 af[2];
                af[0] = *reinterpret_cast<const h16x8 *>(&wreg[stage][0]);
                af[1] = *reinterpret_cast<const h16x8 *>(&wreg[stage][1]);
                {
                    const int gn = g + PF < total_ksteps ? g + PF : total_ksteps - 1;
                    wreg[stage][0] = wp[(size_t)gn * 512];
                    wreg[stage][1] = wp[(size_t)gn * 512 + 64];
                }
#pragma unroll
                for (int i = 0; i < MT / 2; i++) {
                    acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bfA[i], acc[0][i], 0, 0, 0);
                    acc[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bfA[i], acc[1][i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
                __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
                __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
#pragma unroll
                for (int i = 0; i < MT / 2 - 2; i++) {
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
                // ---- half 2 ----
                // T is updated in place for the next tap: rows 0..11 are dead after the last half-2 read of this tap,
                // rows 12..23 after the half-1 read above
                if (ks == 3) tap_rows(next_tap, 0, MT / 2, T);
#pragma unroll
                for (int i = 0; i < MT / 2; i++)
                    bfA[i] = *reinterpret_cast<const h16x8 *>(lds + T[i] + (ks < 3 ? (ks + 1) * 32 : 0));
                if (ks == 3) tap_rows(next_tap, MT / 2, MT, T);
#pragma unroll
                for (int i = 0; i < MT / 2; i++) {
                    acc[0][MT / 2 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bfB[i], acc[0][MT / 2 + i], 0, 0, 0);
                    acc[1][MT / 2 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bfB[i], acc[1][MT / 2 + i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < MT / 2; i++) {
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                g++;
            }
        }
    }

    // ---- epilogue: [relu]; [+ residual]; [final BN]; -> f16 -> NHWC rows in global memory ----
    // Staged through LDS (the image is dead now) so that HBM sees whole 256-byte rows (this workgroup's 128 output
    // channels of a pixel) instead of 8-byte pieces: O[row][128 oc] f16, row stride 272 B.
    constexpr int ORS = 128 * 2 + 16;
    static_assert(ROWS * ORS <= LDS_BYTES, "output tile fits the image buffer");
    auto row_pixel = [&](int row, bool &ok) {  // global pixel row of tile row `row`
        const int mt = row >> 4, b = mt / a.tpb, q = (mt - b * a.tpb) * 16 + (row & 15);
        ok = b < a.bpw && board0 + b < a.boards && q < a.hw;
        return ok ? (size_t)(board0 + b) * a.hw + q : (size_t)0;
    };
    __syncthreads();  // every wave is done with the last chunk's fragments
    if (a.res) {      // residual tile, coalesced: 24 sixteen-byte pieces per thread in two batches
#pragma unroll 1
        for (int part = 0; part < 2; part++) {
            uint4 v[12];
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const int id = tid + (part * 12 + i) * 256;
                bool ok;
                const size_t gr = row_pixel(id >> 4, ok);
                v[i] = *reinterpret_cast<const uint4 *>(a.res + gr * a.ldy + nhalf * 128 + (id & 15) * 8);
            }
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const int id = tid + (part * 12 + i) * 256;
                *reinterpret_cast<uint4 *>(lds + (id >> 4) * ORS + (id & 15) * 16) = v[i];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int ocl = wave * 32 + nt * 16 + kq * 4;  // within this workgroup's 128 channels
        f32x4 ps = f32x4{1.f, 1.f, 1.f, 1.f}, pt = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.post_scale) {
            ps = *reinterpret_cast<const f32x4 *>(a.post_scale + nhalf * 128 + ocl);
            pt = *reinterpret_cast<const f32x4 *>(a.post_shift + nhalf * 128 + ocl);
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            unsigned char *slot = lds + (mt * 16 + fr) * ORS + ocl * 2;  // owned by exactly this lane
            f32x4 v = acc[nt][mt];
            if (a.relu) {
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
            }
            if (a.res) {
                const h16x4 r = *reinterpret_cast<const h16x4 *>(slot);
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] += (float)r[j];  // added in f32, AFTER the ReLU (post_act.py:227-228)
            }
            if (a.post_scale) v = v * ps + pt;
            *reinterpret_cast<h16x4 *>(slot) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 24; i++) {
        const int id = tid + i * 256;
        bool ok;
        const size_t gr = row_pixel(id >> 4, ok);
        if (ok)
            *reinterpret_cast<uint4 *>(a.y + gr * a.ldy + nhalf * 128 + (id & 15) * 8) =
                *reinterpret_cast<const uint4 *>(lds + (id >> 4) * ORS + (id & 15) * 16);
    }
}

}  // namespace

bool board_conv_supported(int dtype, int h, int w, int cin, int cout) {
    const int hw = h * w, tpb = (hw + 15) / 16;
    return dtype == 1 && cin % 128 == 0 && cout % 128 == 0 && tpb <= MT && w <= 32 && h <= 32;
}

size_t board_conv_weight_elems(int cin, int cout) { return (size_t)9 * cin * cout; }

// OIHW f32 (BN folded) -> [n_half][chunk][tap][ks 4][wave 4][nt 2][lane 64][8] f16: element j of lane (fr, kq) is
// W[oc = 128*n_half + 32*wave + 16*nt + fr][channel = 128*chunk + 64*(kq&1) + 16*ks + 8*(kq>>1) + j][tap]
void board_conv_pack_weights(const float *oihw, int cout, int cin, uint16_t *dst) {
    const int chunks = cin / 128, halves = cout / 128;
    size_t o = 0;
    for (int nh = 0; nh < halves; nh++)
        for (int chunk = 0; chunk < chunks; chunk++)
            for (int tap = 0; tap < 9; tap++)
                for (int ks = 0; ks < 4; ks++)
                    for (int wave = 0; wave < 4; wave++)
                        for (int nt = 0; nt < 2; nt++)
                            for (int lane = 0; lane < 64; lane++)
                                for (int j = 0; j < 8; j++) {
                                    const int kq = lane >> 4;
                                    const int oc = 128 * nh + 32 * wave + 16 * nt + (lane & 15);
                                    const int ch = 128 * chunk + 64 * (kq & 1) + 16 * ks + 8 * (kq >> 1) + j;
                                    const _Float16 hv = (_Float16)oihw[((size_t)oc * cin + ch) * 9 + tap];
                                    uint16_t bits;
                                    __builtin_memcpy(&bits, &hv, 2);
                                    dst[o++] = bits;
                                }
}

void launch_board_conv(const BoardConvArgs &t, hipStream_t stream) {
    BoardConvDev d;
    d.x = static_cast<const h16 *>(t.x);
    d.w = static_cast<const uint4 *>(t.weights);
    d.bias = t.bias;
    d.post_scale = t.post_scale;
    d.post_shift = t.post_shift;
    d.res = static_cast<const h16 *>(t.res);
    d.y = static_cast<h16 *>(t.y);
    d.ldx = t.ldx;
    d.ldy = t.ldy;
    d.boards = t.boards;
    d.h = t.h;
    d.w_ = t.w;
    d.hw = t.h * t.w;
    d.tpb = (d.hw + 15) / 16;
    d.bpw = MT / d.tpb;
    d.cin = t.cin;
    d.relu = t.relu;
    d.inv_w = (65536u + (unsigned)t.w - 1) / (unsigned)t.w;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_board_conv_f16, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        done_mask |= 1ull << (dev & 63);
    }
    dim3 grid((t.boards + d.bpw - 1) / d.bpw, t.cout / 128);
    kz_board_conv_f16<<<grid, 256, LDS_BYTES, stream>>>(d);
}

}  // namespace kz
