// kz_tower4.hip — the board-resident chess tower with FOUR boards per workgroup (kz_tower.hip keeps two).
//
// Why: at a full chip the two-board launch is held down by the weight stream, not by the matrix cores — every CU pulls
// the whole 49 MB stream from its XCD's L2 (25 B/clk per CU, 70 % of what a CU can take from L2; the chip's clock
// drops from 2.28 to 1.98 GHz with it).  Four boards per workgroup halve the bytes per evaluation.
//
// What changes against kz_tower.hip (same weight stream, same MFMA orientation, same fragment/bank layout):
//  * LDS holds ONE image of the four boards and every layer runs IN PLACE (two images of four boards do not fit in
//    160 KB): k-loop over the image -> barrier -> epilogue overwrites it -> barrier.
//  * The residual stream X, which the in-place conv A overwrites, lives in a private scratch slab in global memory
//    (L2-resident: 128 KB per workgroup, written by the epilogue that produces X and re-read — prefetched under the last
//    taps of conv B — by the epilogue that adds it; every thread reads back exactly the bytes it wrote).
//  * The image has a zero pixel behind every board line (9 rows per line), so a tap that leaves the board in x reads
//    zeros without any per-lane test, and every fragment address is `base(tap, board pair) + immediate`.  Tiles are
//    board lines of two boards (as in kz_tower.hip), so the tiles a dy = +-1 tap puts outside the board are skipped
//    whole (8.3 % of the multiply-adds).
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

constexpr int C = 256;
constexpr int RS = C * 2 + 16;   // LDS bytes per pixel row (512 B of channels + 16 B pad: 16 rows -> 16 different slots)
constexpr int KSTEPS = 72;       // 9 taps x 8 chunks of 32 channels
constexpr int NB = 4, MT = 16;   // boards and 16-pixel tiles per workgroup; tile t = (pair t>>3, line t&7)

struct L4 {
    static constexpr int LINE = 9 * RS;        // a board line: 8 pixel rows + 1 all-zero row
    static constexpr int BOARD = 8 * LINE;     // 38016 = 148.5 x 256: the second board of a tile sits 8 slots further
    static constexpr int PAIR = 2 * BOARD;
    static constexpr int IMG0 = RS;            // one zero row in front (x = -1 of the first line)
    static constexpr int BYTES = RS + NB * BOARD;  // 152,592
    static_assert(BOARD % 256 == 128, "the two boards of a line tile fall on different slots");
    static_assert(BYTES <= 160 * 1024, "LDS budget");
    static_assert(7 * LINE + 7 * 16 + 3 * 32 < 65536, "tile and chunk offsets fit the ds immediate");
    static constexpr int row(int b, int y, int x) { return IMG0 + b * BOARD + (y * 9 + x) * RS; }
};

constexpr int XCHUNKS = 32;                       // 16-byte pieces of the residual per thread: (nt 4) x (tile pair 8)
constexpr size_t XRES_BYTES = (size_t)XCHUNKS * 256 * 16;  // per workgroup: 128 KB

struct Tower4Dev {
    const h16 *x0;
    const uint4 *w_stem, *w_tower;
    const float *bias, *post_scale, *post_shift;
    h16 *y;
    uint4 *xres;  // [grid][XCHUNKS][256] x 16 B
    int cin_p, batch, depth;
    const uint8_t *bits;
    size_t bits_stride;
    const float *scalars_in;
    int n_scalar, n_bool;
};

// The 256 accumulator registers of a wave (64 output channels x 256 pixels) are the WHOLE accumulator file.  Left to the
// builtin, hipcc's allocator rotates accumulators through spare registers it does not have (MFMAs with dst != src C,
// 200-480 v_accvgpr moves and scratch spills per tap).  As an asm statement with a read-write "a" operand the MFMA is
// pinned to D = C in the accumulator file.  Everything else (fragment reads, the weight ring, waits) stays the
// compiler's; statement order is pinned with sched_barrier(0) where it matters.  Hazards (cdna_hip_programming.md §5.7
// item 2): the operands of these MFMAs come straight from ds_read / global_load destinations (no VALU in between, the
// build is audited for v_mov in the loop bodies); an accumulator is next touched 64 MFMAs later or, by the epilogue's
// reads, after s_nop padding and a workgroup barrier.
#define KZ_MFMA(c, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
// An accumulator's life starts HERE, as the output of an asm statement in the accumulator file (D = 0 * 0 + 0 on the
// cheapest matrix instruction, 2 passes), and every later touch until the epilogue's read is a KZ_MFMA: no compiler copy
// has a reason to exist.  (Initialising with `acc = bias` instead lets hipcc keep the shared bias value in VGPRs and
// copy it into an accumulator register right in front of the first MFMA, or park whole tiles in VGPRs and shuttle them
// through a spare accumulator around each statement — inside the MFMA's wait states.)  The bias is added by the epilogue.
#define KZ_ACC_ZERO(c, z) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %1, 0" : "=a"(c) : "v"(z))

template <int PF, int R>
__global__ __launch_bounds__(256, 1) void kz_tower_resident4(Tower4Dev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int board0 = blockIdx.x * NB;
    const int layers = 2 * a.depth;
    // operand of KZ_ACC_ZERO: materialised here, far from its first use (a VALU write right in front of an asm MFMA that
    // reads it is a hazard hipcc does not pad), and opaque, so that it stays one register pair for the whole kernel
    h16x4 zero4 = {(h16)0.f, (h16)0.f, (h16)0.f, (h16)0.f};
    asm volatile("" : "+v"(zero4));
    const int total_ksteps = layers * KSTEPS;

    // ---- weight stream (as kz_tower.hip): per k-step 16 KB = [wave 4][nt 4][lane 64] x 16 B, PF k-steps ahead ----
    const uint4 *wp = a.w_tower + wave * 256 + lane;
    auto wload = [&](int gk, int nt) __attribute__((always_inline)) { return wp[(size_t)gk * 1024 + nt * 64]; };
    uint4 wreg[PF][4];
#pragma unroll
    for (int s = 0; s < PF; s++) {
        const int gs = s < total_ksteps ? s : total_ksteps - 1;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wreg[s][nt] = wload(gs, nt);
    }
    int g = 0;
    // ---- zero the image (the pad rows stay zero for the whole launch), then the stem input: 32 channels = the first
    // 64 bytes of every pixel row ----
    for (int id = tid; id < L4::BYTES / 16; id += 256) *reinterpret_cast<uint4 *>(lds + id * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int id = tid; id < NB * 64 * 4; id += 256) {
        const int row = id >> 2, c = id & 3, b = row >> 6, p = row & 63;
        const int board = board0 + b;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (board < a.batch) {
            if (a.bits) {  // encode_input_full (rust/kz-core/src/mapping/mod.rs:40-63), 8 channels of one square
                const uint8_t *bb = a.bits + (size_t)board * a.bits_stride;
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
                v = *reinterpret_cast<const uint4 *>(a.x0 + ((size_t)board * 64 + p) * a.cin_p + c * 8);
            }
        }
        *reinterpret_cast<uint4 *>(lds + L4::row(b, p >> 3, p & 7) + c * 16) = v;
    }
    __syncthreads();

    f32x4 acc[4][MT];
    f32x4 bias_cur[4];  // bias of the layer being computed, fetched at its start, added by its epilogue
    auto fetch_bias = [&](int row) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
            bias_cur[nt] = *reinterpret_cast<const f32x4 *>(a.bias + row * C + wave * 64 + nt * 16 + kq * 4);
    };
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) KZ_ACC_ZERO(acc[nt][mt], zero4);
    };
    auto relu4 = [](f32x4 v) {  // on the bit pattern: negative floats are negative integers
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 b = __builtin_bit_cast(i32x4, v);
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = b[j] > 0 ? b[j] : 0;
        return __builtin_bit_cast(f32x4, b);
    };
    auto to_h4 = [](f32x4 v) { return h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]}; };

    // this lane's pixel row of tile (pair 0, line 0); tile t is (t >> 3) * PAIR + (t & 7) * LINE further
    const int lane_base = L4::IMG0 + (fr >> 3) * L4::BOARD + (fr & 7) * RS;
    const int kq_off = 256 * (kq & 1) + 128 * (kq >> 1);  // channel assignment of a k-step, as kz_tower.hip
    const int frag_base = lane_base + kq_off;
    const int epi_base = lane_base + (wave * 64 + kq * 4) * 2;
    uint4 *xres = a.xres + (size_t)blockIdx.x * XCHUNKS * 256 + tid;
    // this workgroup's slab as a buffer: chunk c of thread tid at byte c * 4096 + tid * 16
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(a.xres + (size_t)blockIdx.x * XCHUNKS * 256, 0, (int)XRES_BYTES, 0x00020000);

    // ---- epilogues.  Chunk c = nt * 8 + tp of the residual slab holds this thread's 4 channels of tiles 2tp, 2tp+1. ----
    // store: v -> f16 -> image (in place) [and -> residual slab]
    // The residual slab is written by the NEXT layer's k-loop (copy_x below) and read back in four groups of 8 chunks
    // (one per nt): the first group is prefetched into xpre before the last tap line of conv B, every later group is
    // requested when the epilogue starts on the group before it.  (Stored and re-read inside the epilogues, all
    // workgroups of a launch burst the same 32 MB through L2 at the same time with the matrix cores idle: 11 % of the
    // launch.)
    uint4 xpre[8];
    auto prefetch_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
#ifdef KZ_T4_NO_XLOAD  // (timing experiments: a build that does not read the residual back — wrong results)
            xpre[c] = make_uint4(0, 0, 0, 0);
#else
            xpre[c] = xres[(size_t)c * 256];
#endif
        }
    };
    auto epilogue = [&](auto relu, auto residual, auto post) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu)::value, RES = decltype(residual)::value, POST = decltype(post)::value;
        h16x4 dep = zero4;
        uint4 xr[8], xnext[8];
        if constexpr (RES) {
#pragma unroll
            for (int tp = 0; tp < 8; tp++) xr[tp] = xpre[tp];
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            if constexpr (RES) {
                if (nt < 3) {
                    // (tied to the previous group's last result like the accumulator reads: otherwise all three groups'
                    //  loads are hoisted to the top and spill)
                    const uint4 *xp = xres;
                    asm volatile("" : "+v"(xp) : "v"(dep));
#pragma unroll
                    for (int tp = 0; tp < 8; tp++) {
#ifdef KZ_T4_NO_XLOAD
                        xnext[tp] = make_uint4(0, 0, 0, 0);
#else
                        xnext[tp] = xp[(size_t)((nt + 1) * 8 + tp) * 256];
#endif
                    }
                }
            }
            f32x4 ps, pt;
            if constexpr (POST) {
                const int oc = wave * 64 + nt * 16 + kq * 4;
                ps = *reinterpret_cast<const f32x4 *>(a.post_scale + oc);
                pt = *reinterpret_cast<const f32x4 *>(a.post_shift + oc);
            }
#pragma unroll
            for (int tp = 0; tp < 8; tp++) {
                // The accumulator reads are register-only instructions: left alone, hipcc hoists all 256 of them to the top
                // of the epilogue and spills.  This statement makes the next two tiles' accumulators depend on the last
                // value the previous two produced, so the epilogue walks the tiles in order with bounded pressure.
                asm volatile("" : "+a"(acc[nt][2 * tp]), "+a"(acc[nt][2 * tp + 1]) : "v"(dep));
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int t = 2 * tp + e;
                    f32x4 v = acc[nt][t] + bias_cur[nt];
                    if constexpr (RELU) v = relu4(v);
                    if constexpr (RES) {
                        const h16x4 rx = *reinterpret_cast<const h16x4 *>(reinterpret_cast<const unsigned *>(&xr[tp]) + 2 * e);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] += (float)rx[j];
                    }
                    if constexpr (POST) v = v * ps + pt;
                    const h16x4 hv = to_h4(v);
                    *reinterpret_cast<h16x4 *>(lds + epi_base + (t >> 3) * L4::PAIR + (t & 7) * L4::LINE + nt * 32) = hv;
                    dep = hv;
                }
            }
            if constexpr (RES) {
#pragma unroll
                for (int tp = 0; tp < 8; tp++) xr[tp] = xnext[tp];
            }
        }
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};

    // An MFMA's result may be read 12 wait states after its issue at the earliest, and hipcc pads nothing around an asm
    // MFMA: the epilogue's accumulator reads are register-only instructions, free to be scheduled right behind the MFMA
    // that produced their operand — also behind one several taps back, for the tiles the last taps skip.  So after a
    // k-loop: pad the wait states, then make EVERY accumulator opaque (empty volatile statements keep their order
    // against the MFMA and nop statements, and every reader of an accumulator is ordered behind its statement).
    auto settle = [&]() __attribute__((always_inline)) {
        asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int t = 0; t < MT; t += 4)
                asm volatile("" : "+a"(acc[nt][t]), "+a"(acc[nt][t + 1]), "+a"(acc[nt][t + 2]), "+a"(acc[nt][t + 3]));
    };

    // ---- stem: 9 k-steps over the 32 input channels in the first 64 bytes of every row; no activation ----
    fetch_bias(0);
    init_acc();
    // (compile-time DY like the tower's k-loop: no asm MFMA sits behind a branch, where hipcc would park its accumulator
    //  in a VGPR and shuttle it through a spare accumulator register around the statement — reading it back inside the
    //  MFMA's 12 wait states)
    auto stem_line = [&](auto dy_tag) __attribute__((always_inline)) {
        constexpr int DY = decltype(dy_tag)::value;
        auto &accr = acc;  // (an asm operand alone does not make a generic lambda capture the variable)
#pragma nounroll
        for (int dx = -1; dx <= 1; dx++) {
            const int tap = (DY + 1) * 3 + dx + 1;
            h16x8 af[4];
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const uint4 t = a.w_stem[((tap * 4 + wave) * 4 + nt) * 64 + lane];
                af[nt] = *reinterpret_cast<const h16x8 *>(&t);
            }
            const int sb = lane_base + kq * 16 + (DY * 9 + dx) * RS;  // stem: natural k order, 8 channels per lane group
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const int y = t & 7;
                if ((DY < 0 && y == 0) || (DY > 0 && y == 7)) continue;  // (compile time: the whole tile is padding)
                const h16x8 bf = *reinterpret_cast<const h16x8 *>(lds + sb + (t >> 3) * L4::PAIR + y * L4::LINE);
#pragma unroll
                for (int nt = 0; nt < 4; nt++) KZ_MFMA(accr[nt][t], af[nt], bf);
            }
        }
    };
    stem_line(std::integral_constant<int, -1>{});
    stem_line(std::integral_constant<int, 0>{});
    stem_line(std::integral_constant<int, 1>{});
    settle();
    __syncthreads();
    epilogue(NO, NO, NO);
    __syncthreads();

    // ---- one line of taps (DY, dx = -1..1) over all 256 input channels of the image: 3 x 8 k-steps ----
    // Activation fragments run R tile-steps ahead of their MFMAs through a ring of R register sets.  Tile order within
    // a k-step: first the 12 tiles no tap ever skips (lines 1..6 of both pairs), then the edge lines this DY keeps — so
    // the first R tile-steps of EVERY tap are the same tiles, and the last R reads of a tap can fetch the next tap's
    // first fragments whatever line it belongs to.
    h16x8 bf[R];
    static_assert(R <= 12, "the ring is primed from the tiles common to all taps");
    auto common_tile = [](int i) constexpr { return i < 6 ? i + 1 : i + 3; };  // 1..6, 9..14
    auto prime = [&](int b0) __attribute__((always_inline)) {  // first R tile-steps of a tap whose pair-0 base is b0
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int t = common_tile(s);
            bf[s] = *reinterpret_cast<const h16x8 *>(lds + b0 + (t >> 3) * L4::PAIR + (t & 7) * L4::LINE);
        }
    };
    // copy_x (conv A): this thread's share of the image — the residual stream X, which conv A's epilogue will overwrite —
    // goes to the residual slab during the k-loop: 4 chunks per tap over the first 8 taps, read from LDS at one k-step
    // and stored at the next, in the shadow of the MFMAs.
    auto conv_line = [&](auto dy_tag, bool copy_x) __attribute__((always_inline)) {
        constexpr int DY = decltype(dy_tag)::value;
        constexpr int NTL = DY == 0 ? 16 : 14;     // tiles per k-step
        constexpr int STEPS = 8 * NTL;             // tile-steps per tap
        auto tile_of = [&](int i) constexpr {
            if (i < 12) return common_tile(i);
            if (DY < 0) return i == 12 ? 7 : 15;                      // line 0 of both pairs is skipped
            if (DY > 0) return i == 12 ? 0 : 8;                       // line 7 is skipped
            return i == 12 ? 0 : i == 13 ? 8 : i == 14 ? 7 : 15;
        };
#pragma nounroll
        for (int dx = -1; dx <= 1; dx++) {
            const int b0 = frag_base + (DY * 9 + dx) * RS, b1 = b0 + L4::PAIR;
            // the tap that runs next: (DY, dx + 1), or the first tap of the next line; after the last tap of the layer
            // the reads are surplus (the next layer primes its own ring after the epilogue)
            const int nb0 = dx < 1 ? b0 + RS : (DY < 1 ? b0 + 7 * RS : b0), nb1 = nb0 + L4::PAIR;
            const int tap = (DY + 1) * 3 + dx + 1;
            // no branch inside the loop body (hipcc un-pins the accumulators of asm MFMAs behind branches): when nothing is
            // to be copied the store goes out of the slab's range, where a buffer store is dropped
            const unsigned copy_off = copy_x && tap < 8 ? (unsigned)tid * 16u : 0xfffffff0u;
            uint2 cx[2];
#pragma unroll
            for (int ch = 0; ch < 8; ch++) {
                h16x8 af[4];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) af[nt] = *reinterpret_cast<const h16x8 *>(&wreg[ch & (PF - 1)][nt]);
                const int gn = g + PF < total_ksteps ? g + PF : total_ksteps - 1;  // the k-step this stage is refilled with
#pragma unroll
                for (int i = 0; i < NTL; i++) {
                    const int s = ch * NTL + i, t = tile_of(i);
                    const h16x8 b = bf[s % R];
                    // issue order, pinned: MFMA, [weight-ring refill], fragment read for tile-step s + R, 3 MFMAs — every
                    // memory instruction in the shadow of an MFMA
                    KZ_MFMA(acc[0][t], af[0], b);
                    __builtin_amdgcn_sched_barrier(0);
#ifndef KZ_T4_NO_WLOAD  // (timing experiments: a build without the weight stream — wrong results)
                    if (i < 4) wreg[ch & (PF - 1)][i] = wload(gn, i);
#endif
                    if (s + R < STEPS) {
                        const int s2 = s + R, t2 = tile_of(s2 % NTL);
                        bf[s % R] = *reinterpret_cast<const h16x8 *>(lds + ((t2 >> 3) ? b1 : b0) + (t2 & 7) * L4::LINE + (s2 / NTL) * 16);
                    } else {
                        const int t2 = common_tile(s + R - STEPS);
                        bf[s % R] = *reinterpret_cast<const h16x8 *>(lds + ((t2 >> 3) ? nb1 : nb0) + (t2 & 7) * L4::LINE);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    KZ_MFMA(acc[1][t], af[1], b);
                    KZ_MFMA(acc[2][t], af[2], b);
                    KZ_MFMA(acc[3][t], af[3], b);
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == 8) {  // even k-steps read chunk 4*tap + ch/2 from the image, odd ones store it
#ifndef KZ_T4_NO_XSTORE  // (timing experiments)
                        const int c = 4 * (tap & 7) + (ch >> 1), cnt = c >> 3, ctp = c & 7;
                        if ((ch & 1) == 0) {
                            const int off = epi_base + (ctp >> 2) * L4::PAIR + 2 * (ctp & 3) * L4::LINE + cnt * 32;
                            cx[0] = *reinterpret_cast<const uint2 *>(lds + off);
                            cx[1] = *reinterpret_cast<const uint2 *>(lds + off + L4::LINE);
                        } else {
                            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{cx[0].x, cx[0].y, cx[1].x, cx[1].y}, xrsrc, copy_off,
                                                                   c * 4096, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#endif
                    }
                }
                g++;
            }
        }
    };
    constexpr std::integral_constant<int, -1> DY_UP{};
    constexpr std::integral_constant<int, 0> DY_MID{};
    constexpr std::integral_constant<int, 1> DY_DOWN{};

    for (int layer = 1; layer <= layers; layer++) {
        const bool is_b = (layer & 1) == 0;  // conv A: X -> Y (in place); conv B: Y -> X + residual from the slab
        fetch_bias(layer);
        init_acc();
        prime(frag_base + (-9 - 1) * RS);
        conv_line(DY_UP, !is_b);
        conv_line(DY_MID, !is_b);
        prefetch_x();  // (conv B's epilogue uses it; unconditional, so that xpre is not a value carried around the layer loop)
        conv_line(DY_DOWN, !is_b);
        settle();
        __syncthreads();  // every wave is done reading the image
        if (!is_b) epilogue(YES, NO, NO);
        else if (layer != layers) epilogue(YES, YES, NO);
        else epilogue(YES, YES, YES);
        __syncthreads();
    }

    // ---- tower output: coalesced 16-byte stores of the real pixel rows ----
    for (int id = tid; id < NB * 64 * 32; id += 256) {
        const int p = id >> 5, c16 = id & 31, b = p >> 6, q = p & 63;
        if (board0 + b < a.batch) {
            const uint4 v = *reinterpret_cast<const uint4 *>(lds + L4::row(b, q >> 3, q & 7) + c16 * 16);
            *reinterpret_cast<uint4 *>(a.y + ((size_t)(board0 + b) * 64 + q) * C + c16 * 8) = v;
        }
    }
}

}  // namespace

size_t tower4_scratch_bytes(int batch) { return (size_t)((batch + NB - 1) / NB) * XRES_BYTES; }

void launch_tower_resident4(const TowerArgs &t, void *xres, hipStream_t stream) {
    Tower4Dev d{};
    d.x0 = static_cast<const h16 *>(t.x0);
    d.w_stem = static_cast<const uint4 *>(t.w_stem);
    d.w_tower = static_cast<const uint4 *>(t.w_tower);
    d.bias = t.bias; d.post_scale = t.post_scale; d.post_shift = t.post_shift;
    d.y = static_cast<h16 *>(t.y);
    d.xres = static_cast<uint4 *>(xres);
    d.cin_p = t.cin_p; d.batch = t.batch; d.depth = t.depth;
    d.bits = t.bits; d.bits_stride = t.bits_stride; d.scalars_in = t.scalars_in; d.n_scalar = t.n_scalar; d.n_bool = t.n_bool;
    auto kernel = kz_tower_resident4<4, 8>;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, L4::BYTES);
        done_mask |= 1ull << (dev & 63);
    }
    kernel<<<(t.batch + NB - 1) / NB, 256, L4::BYTES, stream>>>(d);
}

}  // namespace kz
