// kz_board_conv2.hip — per-layer 3x3 convolution for Go-size boards, second organisation (kz_board_conv.hip is the
// first): ONE workgroup per CU holding TWO boards, with the staging of the next 32-channel half-chunk running under the
// current one's MFMAs.
//
// EXPERIMENT, opt-in (KZ_BOARD_CONV2=1), parity-tested, SLOWER than the first organisation: Go-19 40x256 B=512 26.2k
// against 33.5k evals/s (0.234 against 0.181 ms per layer; without staging loads and output traffic 0.190 ms).  With
// one workgroup per CU nothing covers a workgroup's prologue (tile-row map, halo clear, ring fill, first half-chunk:
// ~5 us) and epilogue (~4 us) around 26 us of MFMA work, and the k-loop alone runs at ~70 % — the two co-resident
// workgroups of kz_board_conv.hip cover each other's phases better than this single one pipelines its own.  What it
// would take is a persistent workgroup that prefetches the next item under the current epilogue.
// In-kernel stamps (-DKZ_BC2_STAMPS, tools/board_conv2_stamps.py), cycles per workgroup of 119.5k: set-up 5.1k, first
// half-chunk 4.3k, eight k-loops 78.7k (9.3k each against 6.9k of MFMA time: a single wave per SIMD pays ~265 cycles of
// issue per k-step for its 12 fragment reads, 12 address adds and 4 loads — what the co-resident wave hides in
// kz_board_conv.hip), stage writes + barriers 8k, epilogue 24k (tile writes 10.4k, stores 13.5k).
//
// Why: in kz_board_conv.hip a wave owns 64 output channels x 6 pixel tiles and two workgroups share a CU, so the CU's
// eight waves pull the same 4 KB of weight fragments per k-step through L1 eight times: 53 of the 64 B/clk L1 delivers
// (DESIGN.md §5.4), and the two workgroups' staging / epilogue phases overlap each other's MFMAs only by chance.  Here a
// wave owns 64 output channels x TWELVE tiles (192 accumulators; one wave per SIMD, 512 registers), so a weight fragment
// feeds 12 MFMAs and the CU's L1 weight traffic is 21 B/clk; the image lives in LDS as two 32-channel buffers of
// rows x 80 B with a zero halo (a tap is a constant row offset, as before), and while the k-loop reads one buffer each
// thread's 12 pieces of the next half-chunk are in flight from L2 and are written to the other buffer at the end of the
// k-loop: one barrier per half-chunk.
//
//   workgroup = 2 boards x 24 tile slots of 16 pixel rows (Go: 23 used) x 64 output channels; 4 waves, wave w = tiles
//               [12 w, 12 w + 12): waves 0,1 board 0, waves 2,3 board 1
//   k order   = (chunk of 64 channels, half ks, tap): a k-step is 32 channels of one tap = 48 MFMAs per wave; lane group
//               kq holds channels [8 kq, 8 kq + 8) of the half: row byte 16 kq (2-way bank conflicts on part of the
//               fragment reads, LDS is not what bounds this kernel: 12 reads per 48 MFMAs)
//   epilogue  = as kz_board_conv.hip (residual in the accumulators' layout, result through LDS so that HBM sees whole
//               128-byte lines), one board at a time through the dead image buffers
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kz_kernels.hpp"

namespace kz {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BPW = 2;             // boards per workgroup
constexpr int TSB = 24;            // tile slots per board
constexpr int MTW = 12;            // tiles per wave
constexpr int ROWS = BPW * TSB * 16;  // 768
constexpr int OCW = 64;            // output channels per workgroup
constexpr int PRS = 32 * 2 + 16;   // row stride of a 32-channel buffer: 64 B + 16 B pad
constexpr int ORS = OCW * 2 + 16;  // row stride of the epilogue's output tile
constexpr int PF = 3;              // weight ring depth in k-steps (a k-step is 48 MFMAs = 768 cycles per wave)
constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100;

__device__ __forceinline__ h16x8 lds_frag(int addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const h16x8 __attribute__((address_space(3))) * lds_cptr;
    return *(lds_cptr)(unsigned)addr;
#else
    (void)addr;
    return h16x8{};
#endif
}

// Diagnostic build only (-DKZ_BC2_STAMPS): s_memtime stamps at the phase boundaries of every wave, dumped by the
// launcher to $KZ_BC_STAMP_FILE after the 20th launch.  No stamp executes in the real kernel.
#ifdef KZ_BC2_STAMPS
#define KZ_STAMP2(slot)                                                                        \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wave) * 32 + (slot)] = t_;           \
    } while (0)
#else
#define KZ_STAMP2(slot) do { } while (0)
#endif

struct BoardConv2Dev {
    const h16 *x;
    const uint4 *w;     // [n_quarter][k-step = (chunk, ks, tap)][nt 4][lane 64] x 16 B
    const float *bias, *post_scale, *post_shift;
    const h16 *res;
    h16 *y;
    int bytes, ld, boards, hw, cin, relu, groups, nq;
    const int *rowmap;             // [768] tile row -> board << 20 | pixel << 10 | image row, or -1
    const unsigned short *halo;   // image rows that are halo
    int n_halo, pitch, buf_bytes;
    unsigned long long *stamps;  // diagnostic build only
    int ablate;  // timing experiments only (KZ_BC_ABLATE): 1 no staging loads, 2 no residual/output traffic, 4 no k-loop barrier
};

__global__ __launch_bounds__(256, 1) void kz_board_conv2_f16(BoardConv2Dev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    KZ_STAMP2(0);
    // XCD-aware order as kz_board_conv.hip: the nq channel quarters of a board pair take consecutive slots of one XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nquarter = slot % a.nq, group = (slot / a.nq) * 8 + xcd;
    if (group >= a.groups) return;
    const int board0 = group * BPW;
    const int halves = a.cin / 32;          // half-chunks of 32 input channels
    const int total_ksteps = halves * 9;

    // the 12 (tile row, piece) slots this thread stages: row (tid >> 2) + 64 i, 16-byte piece tid & 3 of its 64 bytes
    int emap[12], emapT[MTW];
#pragma unroll
    for (int i = 0; i < 12; i++) emap[i] = a.rowmap[(tid >> 2) + i * 64];
#pragma unroll
    for (int i = 0; i < MTW; i++) emapT[i] = a.rowmap[(wave * MTW + i) * 16 + fr];

    // weight ring
    const uint4 *wp = a.w + (size_t)nquarter * total_ksteps * 256 + lane;
    uint4 wreg[PF][4];
#pragma unroll
    for (int s = 0; s < PF; s++) {
        const int gs = s < total_ksteps ? s : total_ksteps - 1;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wreg[s][nt] = wp[(size_t)gs * 256 + nt * 64];
    }
    int g = 0;

    // zero the halo rows of both buffers (5 sixteen-byte pieces per row); pixel rows are overwritten by every half-chunk
    for (int id = tid; id < a.n_halo * 10; id += 256) {
        const int k = (int)(((unsigned)id * 6554u) >> 16), pc = id - k * 10;  // id / 10 for id < 16384
        *reinterpret_cast<uint4 *>(lds + (pc >= 5 ? a.buf_bytes + (pc - 5) * 16 : pc * 16) + a.halo[k] * PRS) = make_uint4(0, 0, 0, 0);
    }

    const int piece = tid & 3;
    int po[12], ls[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int e = emap[i];
        const int b = e >> 20, q = (e >> 10) & 1023;
        const bool ok = e >= 0 && board0 + b < a.boards;
        po[i] = ok ? (((board0 + b) * a.hw + q) * a.ld + piece * 8) * 2 : -1;
        ls[i] = (e & 1023) * PRS + piece * 16;
    }
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16 *>(a.x), 0, a.bytes, 0x00020000);

    // centre-tap LDS address of this lane's fragment row for each of the wave's 12 tiles (within a buffer)
    int T0[MTW];
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        const int e = emapT[i];
        const bool valid = e >= 0 && board0 + (e >> 20) < a.boards;
        const int row = valid ? (e & 1023) : a.pitch + 1;
        T0[i] = row * PRS + kq * 16;
    }

    f32x4 acc[4][MTW];
    {
        const int oc = nquarter * OCW + kq * 4;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + oc + nt * 16);
#pragma unroll
            for (int i = 0; i < MTW; i++) acc[nt][i] = b;
        }
    }

    KZ_STAMP2(1);
    // ---- first half-chunk into buffer 0 ----
    u32x4 stage[12];
#pragma unroll
    for (int i = 0; i < 12; i++) stage[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, po[i], 0, 0);
#pragma unroll
    for (int i = 0; i < 12; i++)
        if (po[i] >= 0) *reinterpret_cast<u32x4 *>(lds + ls[i]) = stage[i];  // never into the halo
    __syncthreads();
    KZ_STAMP2(2);

    constexpr int HT = MTW / 2;
    for (int hc = 0; hc < halves; hc++) {
        const int buf = (hc & 1) * a.buf_bytes;
        const bool more = hc + 1 < halves;
        // the next half-chunk's 12 pieces: in flight from here to the end of this k-loop
#pragma unroll
        for (int i = 0; i < 12; i++)
            stage[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, more && !(a.ablate & 1) ? po[i] : -1, (hc + 1) * 64, 0);

        int pitch_prs = a.pitch * PRS;
        asm volatile("" : "+s"(pitch_prs));  // (keeps the 9 x 12 tap rows from being hoisted out of the loop and spilled)
        h16x8 bfA[HT], bfB[HT];
        {
            const int off0 = buf - pitch_prs - PRS;  // tap 0
#pragma unroll
            for (int i = 0; i < HT; i++) bfA[i] = lds_frag(T0[i] + off0);
        }
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int stage_w = tap % PF;  // 9 % PF == 0: the ring stage of a k-step does not depend on the half-chunk
            const int off = buf + (tap / 3 - 1) * pitch_prs + (tap % 3 - 1) * PRS;
            const int offn = tap < 8 ? buf + ((tap + 1) / 3 - 1) * pitch_prs + ((tap + 1) % 3 - 1) * PRS : off;
            h16x8 af[4];
#pragma unroll
            for (int nt = 0; nt < 4; nt++) af[nt] = *reinterpret_cast<const h16x8 *>(&wreg[stage_w][nt]);
            // ---- half 1: tiles 0..5 multiply while the fragments of tiles 6..11 are read ----
#pragma unroll
            for (int i = 0; i < HT; i++) bfB[i] = lds_frag(T0[HT + i] + off);
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int i = 0; i < HT; i++)
                    acc[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bfA[i], acc[nt][i], 0, 0, 0);
            // all six reads first (they are for the OTHER half: consumed 24 MFMAs = 384 cycles later), then the MFMAs — a
            // finer interleave lets the scheduler satisfy a DS_READ group with the read an MFMA is about to wait for
            __builtin_amdgcn_sched_group_barrier(SG_DS_READ, HT, 0);
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, HT * 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- half 2: tiles 6..11 multiply while the next tap's fragments of tiles 0..5 are read ----
#pragma unroll
            for (int i = 0; i < HT; i++) bfA[i] = lds_frag(T0[i] + offn);
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int i = 0; i < HT; i++)
                    acc[nt][HT + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bfB[i], acc[nt][HT + i], 0, 0, 0);
            {   // this stage's fragments have been issued to the MFMAs: refill it for k-step g + PF
                const int gn = g + PF < total_ksteps ? g + PF : total_ksteps - 1;
#pragma unroll
                for (int nt = 0; nt < 4; nt++) wreg[stage_w][nt] = wp[(size_t)gn * 256 + nt * 64];
            }
            __builtin_amdgcn_sched_group_barrier(SG_DS_READ, HT, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, HT * 4 - 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            g++;
        }
        KZ_STAMP2(3 + 2 * (hc & 7));
        // the staged pieces -> the other buffer (last read during the previous half-chunk, which ended with a barrier)
        if (more) {
            const int nbuf = ((hc + 1) & 1) * a.buf_bytes;
#pragma unroll
            for (int i = 0; i < 12; i++)
                if (po[i] >= 0) *reinterpret_cast<u32x4 *>(lds + nbuf + ls[i]) = stage[i];
        }
        __syncthreads();
        KZ_STAMP2(4 + 2 * (hc & 7));
    }

    // ---- epilogue: [relu]; [+ residual]; [final BN]; -> f16 -> NHWC rows in global memory, one board at a time ----
    const bool with_res = a.res != nullptr && !(a.ablate & 2);
    u32x2 resv[4][MTW];
    if (with_res) {
        const auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16 *>(a.res), 0, a.bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            const int e = emapT[i];
            const bool valid = e >= 0 && board0 + (e >> 20) < a.boards;
            const int off = valid ? (((board0 + (e >> 20)) * a.hw + ((e >> 10) & 1023)) * a.ld + kq * 4) * 2 : -1;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) resv[nt][i] = __builtin_amdgcn_raw_buffer_load_b64(rrsrc, off, (nquarter * OCW + nt * 16) * 2, 0);
        }
    }
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.bytes, 0x00020000);
    // output rows this thread stores for one board: tile rows (tid >> 3) + 32 i of that board, piece tid & 7 of its 128 B
    const int opiece = tid & 7;
#pragma unroll
    for (int b = 0; b < BPW; b++) {
        // board b's two waves put their tiles into the out tile (buffer b: both image buffers are dead now)
        unsigned char *tile = lds + b * a.buf_bytes;
        if ((wave >> 1) == b) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const int ocl = nt * 16 + kq * 4;
                f32x4 ps = f32x4{1.f, 1.f, 1.f, 1.f}, pt = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.post_scale) {
                    ps = *reinterpret_cast<const f32x4 *>(a.post_scale + nquarter * OCW + ocl);
                    pt = *reinterpret_cast<const f32x4 *>(a.post_shift + nquarter * OCW + ocl);
                }
#pragma unroll
                for (int i = 0; i < MTW; i++) {
                    unsigned char *slot_p = tile + (((wave & 1) * MTW + i) * 16 + fr) * ORS + ocl * 2;
                    f32x4 v = acc[nt][i];
                    if (a.relu) {
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                    }
                    if (with_res) {
                        const h16x4 r = __builtin_bit_cast(h16x4, resv[nt][i]);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] += (float)r[j];  // in f32, AFTER the ReLU (post_act.py:227-228)
                    }
                    if (a.post_scale) v = v * ps + pt;
                    *reinterpret_cast<h16x4 *>(slot_p) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                }
            }
        }
    }
    KZ_STAMP2(19);
    __syncthreads();
    KZ_STAMP2(20);
#pragma unroll
    for (int b = 0; b < BPW; b++) {
        const unsigned char *tile = lds + b * a.buf_bytes;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int trow = (tid >> 3) + 32 * i;              // row within the board's 384 tile rows
            const int e = a.rowmap[b * TSB * 16 + trow];
            const bool ok = e >= 0 && board0 + b < a.boards;
            const int off = ok && !(a.ablate & 2) ? (((board0 + b) * a.hw + ((e >> 10) & 1023)) * a.ld + opiece * 8) * 2 : -1;
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(tile + trow * ORS + opiece * 16), yrsrc, off,
                                                   nquarter * OCW * 2, 0);  // a padding row's store is out of range: dropped
        }
    }
    KZ_STAMP2(21);
}

struct Geometry2 {
    int tpb, pitch, rpb, buf_bytes;
};
Geometry2 geometry2(int h, int w) {
    Geometry2 g{};
    g.tpb = (h * w + 15) / 16;
    g.pitch = w + 1;
    g.rpb = (h + 2) * g.pitch + 1;
    g.buf_bytes = (BPW * g.rpb * PRS + 255) / 256 * 256;
    return g;
}

}  // namespace

bool board_conv2_supported(int dtype, int h, int w, int cin, int cout) {
    if (!(dtype == 1 && cin % 64 == 0 && cout % OCW == 0 && w <= 32 && h <= 32 && w >= 2 && h >= 2)) return false;
    const Geometry2 g = geometry2(h, w);
    // boards of 13..24 tiles (193..384 squares): two of them are a workgroup; the image rows must fit the 10-bit field
    return g.tpb > 12 && g.tpb <= TSB && BPW * g.rpb < 1024 && 2 * g.buf_bytes <= 160 * 1024 && TSB * 16 * ORS <= g.buf_bytes;
}

int board_conv2_workgroups(int boards, int cout) { return ((boards + BPW - 1) / BPW) * (cout / OCW); }

// OIHW f32 (BN folded) -> [n_quarter][chunk][half ks][tap][nt 4][lane 64][8] f16: element j of lane (fr, kq) is
// W[oc = 64*n_quarter + 16*nt + fr][channel = 64*chunk + 32*ks + 8*kq + j][tap]
void board_conv2_pack_weights(const float *oihw, int cout, int cin, uint16_t *dst) {
    const int chunks = cin / 64, quarters = cout / OCW;
    size_t o = 0;
    for (int nq = 0; nq < quarters; nq++)
        for (int chunk = 0; chunk < chunks; chunk++)
            for (int ks = 0; ks < 2; ks++)
                for (int tap = 0; tap < 9; tap++)
                    for (int nt = 0; nt < 4; nt++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int j = 0; j < 8; j++) {
                                const int kq = lane >> 4;
                                const int oc = OCW * nq + 16 * nt + (lane & 15);
                                const int ch = 64 * chunk + 32 * ks + 8 * kq + j;
                                const _Float16 hv = (_Float16)oihw[((size_t)oc * cin + ch) * 9 + tap];
                                uint16_t bits;
                                __builtin_memcpy(&bits, &hv, 2);
                                dst[o++] = bits;
                            }
}

void board_conv2_tables(int h, int w, std::vector<int> &rowmap, std::vector<unsigned short> &halo) {
    const Geometry2 g = geometry2(h, w);
    rowmap.assign(ROWS, -1);
    for (int r = 0; r < ROWS; r++) {
        const int t = r / 16, b = t / TSB, q = (t % TSB) * 16 + r % 16;
        if (t % TSB < g.tpb && q < h * w) rowmap[r] = b << 20 | q << 10 | (b * g.rpb + (q / w + 1) * g.pitch + q % w + 1);
    }
    halo.clear();
    for (int b = 0; b < BPW; b++)
        for (int idx = 0; idx < g.rpb; idx++)
            if (idx < g.pitch || idx >= (h + 1) * g.pitch || idx % g.pitch == 0) halo.push_back((unsigned short)(b * g.rpb + idx));
}

void launch_board_conv2(const BoardConvArgs &t, hipStream_t stream) {
    BoardConv2Dev d;
    d.x = static_cast<const h16 *>(t.x);
    d.w = static_cast<const uint4 *>(t.weights);
    d.bias = t.bias; d.post_scale = t.post_scale; d.post_shift = t.post_shift;
    d.res = static_cast<const h16 *>(t.res);
    d.y = static_cast<h16 *>(t.y);
    d.bytes = (int)((size_t)t.boards * t.h * t.w * t.ldx * 2);
    d.ld = t.ldx;
    d.boards = t.boards;
    d.hw = t.h * t.w;
    d.cin = t.cin;
    d.relu = t.relu;
    const Geometry2 geo = geometry2(t.h, t.w);
    d.pitch = geo.pitch;
    d.buf_bytes = geo.buf_bytes;
    d.groups = (t.boards + BPW - 1) / BPW;
    d.nq = t.cout / OCW;
    d.rowmap = t.rowmap; d.halo = t.halo; d.n_halo = t.n_halo;
    static const int ablate = getenv("KZ_BC_ABLATE") ? atoi(getenv("KZ_BC_ABLATE")) : 0;
    d.ablate = ablate;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_board_conv2_f16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done_mask |= 1ull << (dev & 63);
    }
    const int grid = ((d.groups + 7) / 8) * 8 * d.nq;
#ifdef KZ_BC2_STAMPS
    static unsigned long long *stamp_buf = nullptr;
    static int launches = 0;
    const size_t stamp_bytes = (size_t)grid * 4 * 32 * sizeof(unsigned long long);
    if (!stamp_buf) (void)hipMalloc((void **)&stamp_buf, (size_t)8192 * 4 * 32 * 8);
    d.stamps = stamp_buf;
    if (launches == 20) (void)hipMemsetAsync(stamp_buf, 0, stamp_bytes, stream);
#else
    d.stamps = nullptr;
#endif
    kz_board_conv2_f16<<<grid, 256, 2 * geo.buf_bytes, stream>>>(d);
#ifdef KZ_BC2_STAMPS
    if (launches++ == 20 && getenv("KZ_BC_STAMP_FILE")) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> host(stamp_bytes / 8);
        (void)hipMemcpy(host.data(), stamp_buf, stamp_bytes, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(getenv("KZ_BC_STAMP_FILE"), "wb")) {
            fwrite(host.data(), 1, stamp_bytes, f);
            fclose(f);
        }
    }
#endif
}

}  // namespace kz
