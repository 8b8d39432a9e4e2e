// kz_board_conv.hip — per-layer 3x3 convolution for boards whose whole tower does not fit in LDS (Go 19x19: a
// 361 x 256 f16 board is 185 KB), f16, input channels a multiple of 32, output channels a multiple of 64.
//
// A workgroup keeps whole boards' pixels in LDS and the 9 taps are 9 shifted views of that image (a tap outside the board
// reads a zero row), so the activations are fetched ONCE per layer and workgroup instead of once per tap.  One launch per
// layer (activations round-trip through HBM), shaped for TWO workgroups per CU — one stages or stores while the other
// multiplies:
//
//   workgroup = 24 tiles of 16 pixel rows (bpw boards, each padded to tpb = ceil(h*w/16) tiles; Go: 23 tiles, bpw = 1)
//               x 64 output channels; 256 threads = 4 waves; wave = (row quarter: 6 tiles) x (all 64 output channels)
//   chunk     = 32 input channels = one k-step (24 MFMAs per wave) per tap, 9 k-steps per chunk.  BOTH operands of a
//               chunk live in LDS: the image chunk WITH a zero halo (board b, pixel (y, x) is image row
//               b*rpb + (y+1)*(w+1) + x+1, the right halo of a line is the left halo of the next, so a tap is a constant
//               row offset; rows of 64 B + 16 B pad: conflict-free fragment reads) and the chunk's 36 KB of weights in
//               fragment order [tap][nt 4][lane 64] x 16 B.  The four waves of a workgroup multiply the same weights:
//               loading them once per workgroup into LDS instead of once per wave into registers takes three quarters
//               of the weight bytes off the CU's one L1 / texture path, which every other load and store of the CU
//               queues behind.  The k-loop of a chunk issues no global load of its own operands
//               and meets no barrier: it only carries the PREFETCH of the next chunk (6 image pieces + 9 weight pieces
//               of 16 B per thread, two per tap, into 60 registers), written to LDS between two barriers at the chunk's
//               end; in the last chunk the same registers fetch the residual.
//   registers = 96 accumulators + 32 weight fragments (this tap's and the next one's) + 24 activation fragments + 60
//               staged pieces: two waves per SIMD.
//   grid      = 1-D, XCD-aware: the cout/64 workgroups of one board group run on ONE XCD back to back, so the
//               image is read from HBM once and from that XCD's L2 by the others.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

#include "kz_kernels.hpp"

namespace kz {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MT = 24;                 // 16-row tiles per workgroup
constexpr int NTW = 4;                 // 16-channel output tiles per wave: all 64 channels of the workgroup
constexpr int MTW = MT / NTW;          // tiles per wave: a row quarter
constexpr int ROWS = MT * 16;          // 384
constexpr int OCW = 64;                // output channels per workgroup
constexpr int CH = 32;                 // input channels per staged chunk
constexpr int PRS = CH * 2 + 16;       // image row stride: 32 channels + 16 B pad = 80 B
constexpr int LDS_MAX = 80 * 1024;     // two workgroups per CU
constexpr int ORS = OCW * 2 + 16;      // row stride of the epilogue's output tile
constexpr int KPC = 9;                 // k-steps per chunk: one per tap
constexpr int WSTEP = NTW * 64 * 16;   // weight bytes of a k-step: 4 KB
constexpr int WCHUNK = KPC * WSTEP;    // ... of a chunk: 36 KB
constexpr int NIMG = 6, NWGT = KPC, NSTG = NIMG + NWGT;  // 16-byte pieces a thread stages per chunk: image, weights
static_assert(ROWS * (CH * 2 / 16) == NIMG * 256 && WCHUNK == NWGT * 256 * 16, "a chunk's pieces divide over 256 threads");

// fragment read from an integer LDS byte address (the dynamic LDS block starts at 0; going through the `lds` symbol
// costs a v_add per read)
__device__ __forceinline__ h16x8 lds_frag(int addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const h16x8 __attribute__((address_space(3))) * lds_cptr;
    return *(lds_cptr)(unsigned)addr;
#else
    (void)addr;
    return h16x8{};
#endif
}

constexpr int SG_MFMA = 0x8, SG_DS_READ = 0x100;
#ifndef KZ_BC_STORE_AUX
#define KZ_BC_STORE_AUX 0  // cache policy bits of the output stores (diagnostic builds: 2 = nt)
#endif

// Diagnostic build only (-DKZ_BC_STAMPS): s_memtime stamps at the phase boundaries of every wave, dumped by the
// launcher to $KZ_BC_STAMP_FILE after the 20th launch.  No stamp executes in the real kernel.
#ifdef KZ_BC_STAMPS
#define KZ_STAMP(slot)                                                                         \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wave) * 32 + (slot)] = t_;           \
    } while (0)
#else
#define KZ_STAMP(slot) do { } while (0)
#endif

struct BoardConvDev {
    const h16 *x;       // [boards*hw][ldx]
    const uint4 *w;     // fragment-packed: [n_quarter][chunk][tap][nt 4][lane 64] x 16 B
    const float *bias, *post_scale, *post_shift;  // [cout]
    const h16 *res;     // optional residual [boards*hw][ld]
    h16 *y;             // [boards*hw][ld]
    int bytes;          // size of y (and of the residual): boards * hw * ld * 2 (< 2^31)
    int bytes_x, ldx;   // the input's (the stem's input tensor is narrower than its output)
    int bytes_w;        // size of the packed weights of this layer
    int ld, boards, h, w_, hw, tpb, bpw, cin, relu, groups, nq;
    unsigned inv_tpb, inv_w, inv_nhb;  // ceil(65536 / tpb), / w, / (halo rows per board): exact quotients for the small
                                       // values they meet
    int pitch, rpb;     // halo image: w + 1 rows per line, (h + 2) * pitch + 1 rows per board
    int w_off;          // LDS offset of the chunk's weights, behind the image
    int rm_off;         // LDS offset of the 384 image-row indices (u16), behind the weights / the epilogue's output tile
    unsigned long long *stamps;  // diagnostic build only
};

__global__ __launch_bounds__(256, 2) void kz_board_conv_f16(BoardConvDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave;  // wave = row quarter (6 tiles) x all 64 output channels
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    KZ_STAMP(0);
#ifdef KZ_BC_STAMPS
    if (lane == 0) {  // where this workgroup ran: HW_ID, XCC_ID, LDS_ALLOC
        a.stamps[((size_t)blockIdx.x * 4 + wave) * 32 + 29] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
        a.stamps[((size_t)blockIdx.x * 4 + wave) * 32 + 30] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
        a.stamps[((size_t)blockIdx.x * 4 + wave) * 32 + 31] = __builtin_amdgcn_s_getreg((8 - 1) << 11 | 0 << 6 | 6);
    }
#endif
    // Two workgroups share a CU.  Left alone they run in lockstep — same work, MFMA pipe shared 50/50, so both stage,
    // both multiply, both store at the same time and nothing overlaps.  Giving ONE of them issue priority (the one
    // whose LDS allocation starts at 0) lets it run its k-loops at full rate and reach its staging/epilogue phases
    // while the other multiplies: the pair falls into complementary phases.
    {
        const unsigned lds_base = __builtin_amdgcn_s_getreg((8 - 1) << 11 | 0 << 6 | 6);  // HW_REG_LDS_ALLOC.LDS_BASE
        if (lds_base == 0) __builtin_amdgcn_s_setprio(3);
    }
    // XCD-aware order: consecutive workgroup ids go to consecutive XCDs, so id = (slot, xcd); the nq channel quarters of
    // a board group take consecutive slots of one XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nquarter = slot % a.nq, group = (slot / a.nq) * 8 + xcd;
    if (group >= a.groups) return;
    const int board0 = group * a.bpw;
    const int chunks = a.cin / CH;

    // ---- set-up, ordered by latency: (1) the first chunk's pieces are requested FIRST — everything about a tile row
    // (board, pixel, image row) is arithmetic on the thread id, no table is read — (2) then, under those loads, the halo
    // clear, the fragment rows and the accumulators.
    // Output slot i of a thread = tile row (tid >> 3) + 32 i, 16-byte piece tid & 7 of its 64 output channels: the same 12
    // slots serve the residual and the output stores.  An image chunk is half as wide (4 pieces per row): a thread stages
    // the 6 of its 12 rows with i % 2 == piece / 4, piece % 4 of each.
    const int piece = tid & 7, piece4 = piece & 3, ihalf = piece >> 2;
    // tile row r -> board b (of this workgroup), pixel q, image row; false for a padding row or a board beyond the batch
    // (every factor is below 2^24: v_mul_u32_u24 / v_mad_u32_u24 run at full rate, a 32-bit v_mul_lo_u32 at a quarter)
    auto locate = [&](int r, int &b, int &q, int &irow) __attribute__((always_inline)) {
        const int mt = r >> 4;
        b = (int)(__umul24((unsigned)mt, a.inv_tpb) >> 16);           // mt / tpb (exact: mt < 24)
        q = (mt - (int)__umul24((unsigned)b, (unsigned)a.tpb)) * 16 + (r & 15);
        const int yy = (int)(__umul24((unsigned)q, a.inv_w) >> 16);   // q / w (exact: q < 512, w <= 32)
        irow = (int)(__umul24((unsigned)b, (unsigned)a.rpb) + __umul24((unsigned)(yy + 1), (unsigned)a.pitch)) +
               (q - (int)__umul24((unsigned)yy, (unsigned)a.w_)) + 1;
        return b < a.bpw && q < a.hw && board0 + b < a.boards;
    };
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16 *>(a.x), 0, a.bytes_x, 0x00020000);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(a.w), 0, a.bytes_w, 0x00020000);
    const auto nrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(a.w), 0, 0, 0x00020000);  // no records: loads return zeros
    const int wvoff = tid * 16;                       // this thread's piece of every 4 KB of weights
    const int wbase = nquarter * chunks * WCHUNK;     // this quarter's weights: + chunk * WCHUNK + j * 4096
    u32x4 stg[NSTG];  // the pieces in flight: [0, NIMG) image, [NIMG, NSTG) weights; in the last chunk [0, 12) the residual
    bool g0ok[NIMG];  // (prologue only)
    int lx0[NIMG];
#pragma unroll
    for (int k = 0; k < NIMG; k++) {
        int b, q, irow;
        g0ok[k] = locate((tid >> 3) + (2 * k + ihalf) * 32, b, q, irow);
        const unsigned pix = __umul24((unsigned)(board0 + b), (unsigned)a.hw) + (unsigned)q;  // < boards * hw < 2^24
        lx0[k] = irow * PRS + piece4 * 16;
        stg[k] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, g0ok[k] ? (int)((__umul24(pix, (unsigned)a.ldx) + (unsigned)piece4 * 8) * 2) : -1, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NWGT; j++) stg[NIMG + j] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wbase + j * 4096, 0);
    // Byte offsets of a thread's slots are NOT kept in registers (the k-loop needs every one of its 256): they are a dozen
    // full-rate VALU instructions each, recomputed where they are used — under the MFMAs of the k-loop for the prefetch.
    // out_off(i) = output slot i in the [pixels][ld] output tensor (and the residual), image_off(k) = image slot k in the
    // [pixels][ldx] input tensor; -1 for a padding row (32-bit offsets on a uniform base)
    // (`t` = the thread id behind an optimisation barrier of the chunk loop: the offsets are loop-invariant, and the compiler
    // would otherwise hoist all eighteen out of the loop and spill them)
    auto out_off = [&](int t, int i) __attribute__((always_inline)) {
        int b, q, irow;
        const bool ok = locate((t >> 3) + i * 32, b, q, irow);
        const unsigned pix = __umul24((unsigned)(board0 + b), (unsigned)a.hw) + (unsigned)q;
        return ok ? (int)((__umul24(pix, (unsigned)a.ld) + (unsigned)piece * 8) * 2) : -1;
    };
    auto image_off = [&](int t, int k) __attribute__((always_inline)) {
        int b, q, irow;
        const bool ok = locate((t >> 3) + (2 * k + ihalf) * 32, b, q, irow);
        const unsigned pix = __umul24((unsigned)(board0 + b), (unsigned)a.hw) + (unsigned)q;
        return ok ? (int)((__umul24(pix, (unsigned)a.ldx) + (unsigned)piece4 * 8) * 2) : -1;
    };
    // (where an image piece goes in LDS — image row * PRS — or -1 for a padding row: 384 entries behind the weights, read
    // back at the chunk boundaries)
#pragma unroll
    for (int i = 0; i < 12; i++) {
        int b, q, irow;
        const bool ok = locate((tid >> 3) + i * 32, b, q, irow);
        if (piece == 0) *reinterpret_cast<int *>(lds + a.rm_off + ((tid >> 3) + i * 32) * 4) = ok ? irow * PRS : -1;
    }

    KZ_STAMP(21);
    // zero the halo rows once (5 sixteen-byte pieces per row); they are never written again, and the pixel rows are
    // overwritten by every chunk.  Halo row k of a board: the line above the board (k < pitch), the left neighbour of
    // every line (the right neighbour of the line before), the line below plus one.
    {
        const int nhb = 2 * a.pitch + a.h + 1;  // halo rows per board
        for (int id = tid; id < a.bpw * nhb * 5; id += 256) {
            const int k = (int)(((unsigned)id * 13108u) >> 16), pc = id - k * 5;  // id / 5 for id < 8192
            const int b = (int)(((unsigned)k * a.inv_nhb) >> 16), kk = k - b * nhb;
            const int row = b * a.rpb + (kk < a.pitch ? kk : kk < a.pitch + a.h ? (kk - a.pitch + 1) * a.pitch : (a.h + 1) * a.pitch + (kk - a.pitch - a.h));
            *reinterpret_cast<uint4 *>(lds + pc * 16 + row * PRS) = make_uint4(0, 0, 0, 0);
        }
    }

    KZ_STAMP(22);
    // Centre-tap LDS address of this lane's fragment row for each of the wave's 6 tiles (16-byte piece kq of the row's
    // 32 channels); a lane without a pixel (padding row, missing board) reads pixel (0, 0) of board 0 — its outputs are
    // never stored.
    // T[i] walks the nine taps of a chunk by constant steps and returns to the first one at the chunk's end.
    int T[MTW];
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        int b, q, irow;
        const bool valid = locate((wr * MTW + i) * 16 + fr, b, q, irow);
        T[i] = (valid ? irow : a.pitch + 1) * PRS + kq * 16 - a.pitch * PRS - PRS;  // tap 0: one line up, one pixel left
    }

    KZ_STAMP(24);
    f32x4 acc[NTW][MTW];
    {
        const int oc = nquarter * OCW + kq * 4;
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + oc + nt * 16);
#pragma unroll
            for (int i = 0; i < MTW; i++) acc[nt][i] = b;
        }
    }

    // LDS address of this lane's fragment row per tile for one tap: a constant row offset in the halo image.
    // (pitch_prs is a.pitch * PRS behind an optimisation barrier inside the chunk loop: the rows are the same for every
    // chunk, and the compiler would otherwise hoist all 9 x 6 of them out of the loop and spill them)
    // (from tap - 1 to tap: one pixel right, or — at the start of a tap line — one line down and two pixels left)
    auto tap_rows = [&](int tap, int lo, int hi, int pitch_prs) {
        const int step = tap % 3 ? PRS : pitch_prs - 2 * PRS;
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            if (i < lo || i >= hi) continue;
            T[i] += step;
        }
    };
    // where this thread's staged pieces go: the image slots (never into the halo) and its piece of every 4 KB of weights
    auto weights_to_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NWGT; j++) *reinterpret_cast<u32x4 *>(lds + a.w_off + j * 4096 + wvoff) = stg[NIMG + j];
    };

    KZ_STAMP(1);
#ifdef KZ_BC_NO_SKIP  // (diagnostic builds: the A/B of the padding-tile skip)
    const bool skip_last_tile = false;
#else
    // (wave-uniform) this wave's sixth tile lies beyond the workgroup's last pixel tile
    const bool skip_last_tile = (wr * MTW + MTW - 1) >= a.bpw * a.tpb && (wr * MTW + MTW - 2) < a.bpw * a.tpb;
#endif
    const bool with_res = a.res != nullptr;
    const auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16 *>(with_res ? a.res : a.x), 0, with_res ? a.bytes : 0, 0x00020000);
    // ---- chunk 0 into LDS (the halo clear of other threads touches other rows: no barrier in front of these writes) ----
    KZ_STAMP(2);
#pragma unroll
    for (int k = 0; k < NIMG; k++)
        if (g0ok[k]) *reinterpret_cast<u32x4 *>(lds + lx0[k]) = stg[k];  // never into the halo
    weights_to_lds();
    KZ_STAMP(3);
    const int wa_lds = a.w_off + lane * 16;  // fragment nt of tap t: + t * 4096 + nt * 1024
    for (int chunk = 0; chunk < chunks; chunk++) {
        __syncthreads();  // the chunk is staged
        KZ_STAMP(4 + (chunk & 3) * 4);
        // What the k-loop prefetches, two pieces per tap (no branch in the loop: descriptors, offsets and voffsets are
        // selected here): the next chunk's image pieces [0, NIMG) and weight pieces [NIMG, NSTG); in the last chunk the
        // residual's 12 pieces (without a residual its descriptor has no records: the loads return zeros without traffic)
        const bool last_chunk = chunk + 1 == chunks;
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const auto irsrc = last_chunk ? rrsrc : xrsrc;               // pieces [0, NIMG)
        const auto qrsrc = last_chunk ? rrsrc : wrsrc;               // pieces [NIMG, 12)
        const auto zrsrc = last_chunk ? nrsrc : wrsrc;               // pieces [12, NSTG): weights, or nothing
        const int isoff = last_chunk ? nquarter * OCW * 2 : (chunk + 1) * CH * 2;
        const int wsoff = wbase + (chunk + 1) * WCHUNK;              // (a weight piece beyond the last chunk is out of range)

        // Two half-steps per k-step: the MFMAs of tiles 0..2 run while the fragments of tiles 3..5 and the NEXT tap's
        // weight fragments are read, the MFMAs of tiles 3..5 while the next tap's fragments of tiles 0..2 are read; the
        // other wave of the SIMD (the CU's second workgroup) fills whatever latency is left.
        // A wave whose sixth tile is pure padding (Go's 361 pixels are 22.6 tiles, so tile 23 — wave 3's sixth — holds
        // none) does not issue that tile's fragment read and 4 MFMAs: 4 % of the launch's MFMA work, and of its power.
        constexpr int HT = MTW / 2;
        h16x8 bfA[HT], bfB[HT] = {};
        h16x8 wa[NTW];  // the current tap's weight fragments; fragment nt is re-read for the next tap behind its last MFMAs
        const int pitch_prs = a.pitch * PRS;
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) wa[nt] = lds_frag(wa_lds + nt * 1024);
#pragma unroll
        for (int i = 0; i < HT; i++) bfA[i] = lds_frag(T[i]);
#pragma unroll
        for (int tap = 0; tap < KPC; tap++) {
            const int next_tap = tap < KPC - 1 ? tap + 1 : KPC - 1;
            // ---- half 1: tiles 0..2, weight fragment outermost ----
#pragma unroll
            for (int i = 0; i < HT - 1; i++) bfB[i] = lds_frag(T[HT + i]);
            if (!skip_last_tile) bfB[HT - 1] = lds_frag(T[MTW - 1]);
#pragma unroll
            for (int nt = 0; nt < NTW; nt++)
#pragma unroll
                for (int i = 0; i < HT; i++)
                    acc[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[nt], bfA[i], acc[nt][i], 0, 0, 0);
            // the reads first: a fragment is then consumed >= 12 MFMAs (192 cycles) after its read was issued
            __builtin_amdgcn_sched_group_barrier(SG_DS_READ, HT - 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, HT * NTW, 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- half 2: tile 5 (if it holds pixels), then tiles 3 and 4 with the weight fragment outermost; fragment nt is
            // dead behind its two MFMAs and is read again for the next tap: 6 + nt MFMAs (100-150 cycles) ahead of its use.
            // T is updated in place for the next tap: rows 0..2 are dead after the half-2 read of the previous tap, rows 3..5
            // after the half-1 read above
            if (tap < KPC - 1) {
                tap_rows(next_tap, 0, HT, pitch_prs);
#pragma unroll
                for (int i = 0; i < HT; i++) bfA[i] = lds_frag(T[i]);
                tap_rows(next_tap, HT, MTW, pitch_prs);
            }
            if (!skip_last_tile) {
#pragma unroll
                for (int nt = 0; nt < NTW; nt++)
                    acc[nt][MTW - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[nt], bfB[HT - 1], acc[nt][MTW - 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) {
#pragma unroll
                for (int i = 0; i < HT - 1; i++)
                    acc[nt][HT + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[nt], bfB[i], acc[nt][HT + i], 0, 0, 0);
                if (tap < KPC - 1) wa[nt] = lds_frag(wa_lds + (tap + 1) * WSTEP + nt * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- the prefetch: pieces 2 tap and 2 tap + 1 ----
#pragma unroll
            for (int p = 2 * tap; p < 2 * tap + 2 && p < NSTG; p++) {
                if (p < NIMG) {
                    stg[p] = __builtin_amdgcn_raw_buffer_load_b128(irsrc, last_chunk ? out_off(tv, p) : image_off(tv, p), isoff, 0);
                } else if (p < 12) {
                    stg[p] = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, last_chunk ? out_off(tv, p) : wvoff, last_chunk ? isoff : wsoff + (p - NIMG) * 4096, 0);
                } else {
                    stg[p] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, wvoff, wsoff + (p - NIMG) * 4096, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        KZ_STAMP(5 + (chunk & 3) * 4);
#pragma unroll
        for (int i = 0; i < MTW; i++) T[i] -= 2 * pitch_prs + 2 * PRS;  // back to tap 0
        if (!last_chunk) {
            // ---- the next chunk's pieces sit in registers: into LDS once every wave is done with this chunk's fragments ----
            int erow[NIMG];
#pragma unroll
            for (int k = 0; k < NIMG; k++)
                erow[k] = *reinterpret_cast<const int *>(lds + a.rm_off + ((tid >> 3) + (2 * k + ihalf) * 32) * 4);
            __syncthreads();
            KZ_STAMP(6 + (chunk & 3) * 4);
#pragma unroll
            for (int k = 0; k < NIMG; k++)
                if (erow[k] >= 0) *reinterpret_cast<u32x4 *>(lds + erow[k] + piece4 * 16) = stg[k];  // never into the halo
            weights_to_lds();
            KZ_STAMP(7 + (chunk & 3) * 4);
        }
    }

    // ---- epilogue: [relu]; [+ residual]; [final BN]; -> f16 -> NHWC rows in global memory ----
    // The result is staged through LDS (image and weights are dead now) so that HBM sees whole 128-byte lines (this
    // workgroup's 64 output channels of a pixel) instead of 8-byte pieces: O[row][64 oc] f16, row stride 144 B.  The
    // residual arrived in the staging registers as the same coalesced 16-byte pieces during the last chunk: it goes into O
    // first, and each lane then replaces the 8 bytes it owns (its 4 channels of a pixel row) by relu(acc) + residual, added
    // in f32.
    __syncthreads();  // every wave is done with the last chunk's fragments
    KZ_STAMP(18);
    const int out_lds = (tid >> 3) * ORS + piece * 16;  // + i * 32 * ORS
    if (with_res) {
#pragma unroll
        for (int i = 0; i < 12; i++) *reinterpret_cast<u32x4 *>(lds + out_lds + i * 32 * ORS) = stg[i];
        __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
        const int ocl = nt * 16 + kq * 4;  // within this workgroup's 64 channels
        f32x4 ps = f32x4{1.f, 1.f, 1.f, 1.f}, pt = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.post_scale) {
            ps = *reinterpret_cast<const f32x4 *>(a.post_scale + nquarter * OCW + ocl);
            pt = *reinterpret_cast<const f32x4 *>(a.post_shift + nquarter * OCW + ocl);
        }
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            unsigned char *slot = lds + ((wr * MTW + i) * 16 + fr) * ORS + ocl * 2;  // owned by exactly this lane
            f32x4 v = acc[nt][i];
            if (a.relu) {
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
            }
            if (with_res) {
                const h16x4 r = *reinterpret_cast<const h16x4 *>(slot);
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] += (float)r[j];  // added in f32, AFTER the ReLU (post_act.py:227-228)
            }
            if (a.post_scale) v = v * ps + pt;
            *reinterpret_cast<h16x4 *>(slot) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
        }
    }
    __syncthreads();
    KZ_STAMP(19);
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 12; i++)  // a padding row's store is out of range and dropped
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(lds + out_lds + i * 32 * ORS), yrsrc, out_off(tid, i),
                                               nquarter * OCW * 2, KZ_BC_STORE_AUX);
    KZ_STAMP(20);
}

}  // namespace

namespace {
// halo image geometry: boards per workgroup limited by the 24 tiles and by the LDS left beside a chunk's weights (two
// workgroups per CU)
struct Geometry {
    int tpb, bpw, pitch, rpb, plane, rm_off, lds_bytes;
};
constexpr int RM_BYTES = ROWS * 4;
Geometry geometry(int h, int w) {
    Geometry g{};
    g.tpb = (h * w + 15) / 16;
    g.pitch = w + 1;
    g.rpb = (h + 2) * g.pitch + 1;
    const int by_tiles = g.tpb <= MT ? MT / g.tpb : 0, by_lds = ((LDS_MAX - RM_BYTES - WCHUNK) / PRS) / g.rpb;
    g.bpw = by_tiles < by_lds ? by_tiles : by_lds;
    g.plane = (g.bpw * g.rpb * PRS + 255) / 256 * 256;  // the image; the chunk's weights follow
    g.rm_off = g.plane + WCHUNK > ROWS * ORS ? g.plane + WCHUNK : ROWS * ORS;  // the epilogue reuses both for the output tile
    g.lds_bytes = g.rm_off + RM_BYTES;
    return g;
}
}  // namespace

int board_conv_cin_granule() { return CH; }

bool board_conv_supported(int dtype, int h, int w, int cin, int cout) {
    return dtype == 1 && cin % CH == 0 && cout % OCW == 0 && w <= 32 && h <= 32 && w >= 2 && h >= 2 &&
           geometry(h, w).bpw >= 1;
}

int board_conv_workgroups(int boards, int h, int w, int cout) {
    const int bpw = geometry(h, w).bpw;
    return bpw ? ((boards + bpw - 1) / bpw) * (cout / OCW) : 0;
}

size_t board_conv_weight_elems(int cin, int cout) { return (size_t)9 * cin * cout; }

// OIHW f32 (BN folded) -> [n_quarter][chunk][tap][nt 4][lane 64][8] f16: element j of lane (fr, kq) is
// W[oc = 64*n_quarter + 16*nt + fr][channel = 32*chunk + 8*kq + j][tap]
void board_conv_pack_weights(const float *oihw, int cout, int cin, uint16_t *dst) {
    const int chunks = cin / CH, quarters = cout / OCW;
    size_t o = 0;
    for (int nq = 0; nq < quarters; nq++)
        for (int chunk = 0; chunk < chunks; chunk++)
            for (int tap = 0; tap < 9; tap++)
                for (int nt = 0; nt < OCW / 16; nt++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int kq = lane >> 4;
                            const int oc = OCW * nq + 16 * nt + (lane & 15);
                            const int ch = CH * chunk + 8 * kq + j;
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
    d.bytes = (int)((size_t)t.boards * t.h * t.w * t.ldy * 2);
    d.bytes_x = (int)((size_t)t.boards * t.h * t.w * t.ldx * 2);
    d.bytes_w = (int)((size_t)9 * t.cin * t.cout * 2);
    d.ld = t.ldy;
    d.ldx = t.ldx;
    d.boards = t.boards;
    d.h = t.h;
    d.w_ = t.w;
    d.hw = t.h * t.w;
    const Geometry geo = geometry(t.h, t.w);
    d.tpb = geo.tpb;
    d.bpw = geo.bpw;
    d.pitch = geo.pitch;
    d.rpb = geo.rpb;
    d.w_off = geo.plane;
    d.rm_off = geo.rm_off;
    d.cin = t.cin;
    d.relu = t.relu;
    d.groups = (t.boards + d.bpw - 1) / d.bpw;
    d.nq = t.cout / OCW;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    d.inv_tpb = (65536u + (unsigned)geo.tpb - 1) / (unsigned)geo.tpb;
    d.inv_w = (65536u + (unsigned)t.w - 1) / (unsigned)t.w;
    const unsigned nhb = 2 * (unsigned)geo.pitch + (unsigned)t.h + 1;
    d.inv_nhb = (65536u + nhb - 1) / nhb;
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_board_conv_f16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done_mask |= 1ull << (dev & 63);
    }
    const int grid = ((d.groups + 7) / 8) * 8 * d.nq;
#ifdef KZ_BC_STAMPS
    static unsigned long long *stamp_buf = nullptr;
    static int launches = 0;
    const size_t stamp_bytes = (size_t)grid * 4 * 32 * sizeof(unsigned long long);
    if (!stamp_buf) (void)hipMalloc((void **)&stamp_buf, (size_t)8192 * 4 * 32 * 8);
    d.stamps = stamp_buf;
    if (launches == 20) (void)hipMemsetAsync(stamp_buf, 0, stamp_bytes, stream);
#else
    d.stamps = nullptr;
#endif
    kz_board_conv_f16<<<grid, 256, geo.lds_bytes, stream>>>(d);
#ifdef KZ_BC_STAMPS
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
