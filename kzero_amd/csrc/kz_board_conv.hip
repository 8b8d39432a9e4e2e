// kz_board_conv.hip — per-layer 3x3 convolution for boards whose whole tower does not fit in LDS (Go 19x19: a
// 361 x 256 f16 board is 185 KB), f16, channels a multiple of 64.
//
// Same inner loop as the resident tower (kz_tower.hip): a workgroup keeps whole boards' pixels in LDS and the 9 taps
// are 9 shifted views of that image (a tap outside the board reads a zero row), so the activations are fetched from
// L2 ONCE per layer and workgroup instead of once per tap; the weights stream from L2 straight into MFMA A-fragment
// registers in host-packed fragment order.  What differs from the tower: one launch per layer (activations round-trip
// through HBM: 2-3 x 94.6 MB per layer at Go B=512, the same order of time as the MFMAs), so the kernel is shaped
// for TWO workgroups per CU — one stages or stores while the other multiplies:
//
//   workgroup = 24 tiles of 16 pixel rows (bpw boards, each padded to tpb = ceil(h*w/16) tiles; Go: 23 tiles, bpw = 1)
//               x 64 output channels; 256 threads = 4 waves; wave = (row quarter: 6 tiles) x (all 64 output channels)
//   LDS       = one 64-channel chunk of the image, WITH a zero halo: board b, pixel (y, x) sits at 16-byte slot
//               b*board16 + (y+1)*line16 + (x+1)*5 of a plane (a pixel row is 80 B = 5 slots: 64 B of channels + 16 B pad),
//               so a tap is a constant byte offset and needs no validity test in the k-loop — per tap and tile ONE
//               v_add, against a compare/select chain per tile that made the loop issue-bound.  Two planes (channels
//               [0,32) and [32,64); split arithmetic: channel pieces {0,2} and {1,3} of the chunk, hi then lo) a multiple
//               of 256 B apart: a ds_read_b128 is served in four groups of 16 lanes, each the sixteen pixel rows of a
//               tile spread over lane groups kq and kq + 1, which read the two planes at the same row offset — a group
//               is conflict-free when the sixteen rows fall on sixteen different 16-byte slots of the 256-byte bank row.
//               Rows of a line advance by 5 slots (odd: 16 consecutive rows are 16 slots); a tile of 16 pixels that
//               crosses a line end — 80 % of Go's tiles, 16 pixels on 19-wide lines — must see the same advance across
//               the gap: line16 = 5*(w+1) + 11 slots (left halo row, w pixel rows, an 11-slot gap whose first 5 slots
//               are the right halo row), so that line16 = 5*w (mod 16).  (Rounds 1-4 had line16 = 5*(w+1), the right
//               halo of a line being the left halo of the next: pixel (y+1, 0) then sat 10 slots behind (y, w-1), rows
//               r and r+16 of such a tile shared a slot and every group of the read took two cycles: SQ_LDS_BANK_CONFLICT
//               was 44 % of SQ_LDS_IDX_ACTIVE on Go 19x19, 48 % in split arithmetic, whose four channel pieces in ONE
//               plane conflicted on top of that.)  The gap costs 21 % more LDS per board: a board size that would lose a
//               board per workgroup to it keeps the old line (Geometry::skew16 = 0).  Go: 75 KB, two workgroups per CU.
//   registers = 96 accumulators + 24 fragment + 48 weight ring: < 256, two waves per SIMD.  The ring's 48 registers are
//               also the staging buffer: during the last PF k-steps of a chunk a ring stage that has fed its MFMAs is not
//               refilled with weights but with this thread's 12 pieces of the NEXT chunk's image (or, in the last chunk,
//               of the residual), so those loads fly under the MFMAs instead of behind a barrier.
//   grid      = 1-D, XCD-aware: the cout/64 workgroups of one board group run on ONE XCD back to back, so the
//               image is read from HBM once and from that XCD's L2 by the others.
//
// Two instances of one body: kz_board_conv_f16, and kz_board_conv_split16 — the same loop in the split arithmetic of
// kz_tower_split.hip ((hi, lo) f16 pairs, three MFMAs per product) for KZ_DTYPE_F32_SPLIT16 on Go-size boards: chunks of 32
// channels whose hi and lo halves fill the two planes, 128-byte [hi 32 | lo 32] groups in HBM, the epilogue in two passes
// through the same output tile.
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
// Wave shape of the f16 instance (template parameters of the body; the split instance is <4, 3>):
//   NTW = 16-channel output tiles per wave: 4 = all 64 channels of the workgroup x a row quarter (6 tiles: a weight fragment
//         feeds 6 MFMAs, the four waves load the SAME 4 KB of weights per k-step); 2 = 32 channels x a row half (12 tiles:
//         a weight fragment feeds 12 MFMAs, a wave loads 2 KB per k-step — half the traffic through the CU's one L1 path —
//         and every image tile is read from LDS by two waves instead of one)
//   PF  = weight ring depth in k-steps; PF x NTW = 12 registers of 16 bytes, which double as the staging buffer
#ifndef KZ_BC_NTW
#define KZ_BC_NTW 4
#endif
constexpr int ROWS = MT * 16;          // 384
constexpr int OCW = 64;                // output channels per workgroup
constexpr int CH = 64;                 // input channels per staged chunk
constexpr int PRS = 32 * 2 + 16;       // plane row stride: 32 channels + 16 B pad = 80 B
constexpr int LDS_MAX = 80 * 1024;     // two workgroups per CU
constexpr int ORS = OCW * 2 + 16;      // row stride of the epilogue's output tile
constexpr int KPC = 18;                // k-steps (32 channels of one tap) per chunk: 9 taps x 2
// f(integral_constant<0>), ..., f(integral_constant<N - 1>), in order
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

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
#ifndef KZ_BC_DIAG
#define KZ_BC_DIAG 0  // diagnostic builds only (wrong results): 1 = no output-tile writes in the epilogue, 2 = no read-back of the
#endif                // output tile, 4 = no halo clear, 8 = no residual reads — to attribute SQ_LDS_BANK_CONFLICT by phase
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
// (-DKZ_BC_REALTIME on top: slots 25 / 26 hold s_memrealtime at a wave's start / end instead of two set-up stamps, and the
// launcher keeps the stamps of four consecutive launches — tools/go_clock.sh derives the clock the CUs really run at and
// the idle time between two launches from them)
#define KZ_RSTAMP(slot)                                                                        \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wave) * 32 + (slot)] = t_;           \
    } while (0)
#else
#define KZ_STAMP(slot) do { } while (0)
#endif

struct BoardConvDev {
    const h16 *x;       // [boards*hw][ld]
    const uint4 *w;     // fragment-packed: [n_quarter][k-step][nt 4][lane 64] x 16 B
    const float *bias, *post_scale, *post_shift;  // [cout]
    const h16 *res;     // optional residual [boards*hw][ld]
    h16 *y;             // [boards*hw][ld]
    int bytes;          // size of y (and of the residual): boards * hw * ld * 2 (< 2^31)
    int bytes_x, ldx;   // the input's: ldx == ld unless the convolution has a single chunk (the stem: 64 input channels)
    int ld, boards, h, w_, hw, tpb, bpw, cin, relu, groups, nq;
    unsigned inv_tpb, inv_w, inv_nhb;  // ceil(65536 / tpb), / w: exact quotients for the small values they meet;
                                       // ceil(2^32 / halo slots per board and plane)
    int pitch, plane;       // halo image: w + 1 rows per line (+ the gap), bytes per plane
    int line16, board16;    // 16-byte slots per line (5 * pitch + skew) and per board ((h + 2) * line16 + 5)
    unsigned inv_10;
    int rm_off;             // LDS offset of the 384 image-row indices (u16), behind the image / the epilogue's output tile
    // split arithmetic (kz_board_conv_split16) only: a tensor row is [hi 32 | lo 32] f16 per group of 32 channels (ld = 2 C);
    // y32 != nullptr: the result as f32 [pixels][ld32 = C] instead (the tower's last layer, for the f32 heads)
    int ld32, bytes32;
    float *y32;
    unsigned long long *stamps;  // diagnostic build only
};

// SPLIT: every activation and weight is a (hi, lo) f16 pair, x = hi + lo to 22 bits, and a product is three MFMAs
// (hi*hi + lo*hi + hi*lo, f32 accumulation) — the arithmetic of kz_tower_split.hip, per layer: a chunk is 32 channels, its
// hi halves in plane 0 and lo halves in plane 1 of the same image, a tap takes a hi and a lo weight step from the ring.  In
// HBM a pixel row is [hi 32 | lo 32] per group of 32 channels: a chunk, and a pass of the epilogue, are whole 128-byte lines.
template <bool SPLIT, int NTW, int PF>
__device__ __forceinline__ void board_conv_body(const BoardConvDev &a) {
    constexpr int MTW = MT / NTW;  // tiles per wave (there are NTW row groups)
    static_assert(KPC % PF == 0, "ring stage of a k-step must not depend on the chunk");
    static_assert(NTW * PF == 12, "12 staging pieces = PF ring stages x NTW registers");
    static_assert(!SPLIT || NTW == 4, "the split epilogue's two passes are written for 64 channels per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave = (output-channel group wo) x (row group wr): an activation fragment feeds NTW MFMAs, a weight fragment MTW
    const int wo = wave % (4 / NTW), wr = wave / (4 / NTW);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    // XCD-aware order: consecutive workgroup ids go to consecutive XCDs, so id = (slot, xcd); the nq channel quarters of
    // a board group take consecutive slots of one XCD
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
#ifndef KZ_BC_PRIO_MODE
#define KZ_BC_PRIO_MODE 0  // diagnostic builds (tools/ab_go.sh): 1 = memory phases high / k-loops low for BOTH workgroups,
#endif                     // 2 = the opposite, 3 = no priorities at all
        if (KZ_BC_PRIO_MODE == 0 && lds_base == 0) __builtin_amdgcn_s_setprio(3);
        if (KZ_BC_PRIO_MODE == 1) __builtin_amdgcn_s_setprio(3);
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nquarter = slot % a.nq, group = (slot / a.nq) * 8 + xcd;
    if (group >= a.groups) return;
    const int board0 = group * a.bpw;
    constexpr int CHS = SPLIT ? 32 : CH;  // input channels per staged chunk
    const int chunks = a.cin / CHS;
    const int total_ksteps = chunks * KPC;

    // ---- set-up, ordered by latency: (1) this thread's 12 image pieces of chunk 0 are requested FIRST — everything about
    // a tile row (board, pixel, image row) is arithmetic on the thread id, no table is read — (2) then the weight ring,
    // (3) then, under those loads, the halo clear, the fragment rows and the accumulators.
    // Slot i of a thread = tile row trow + 32 i, 16-byte piece `piece` of its 64 channels; the same 12 slots serve
    // the staging of every chunk, the residual and the output stores.
    // f16: a ds_write_b128 is served in eight groups of 8 contiguous lanes, banks (a / 4) mod 32.  With a group = the eight
    // pieces of ONE row (rounds 1-5: piece = tid & 7, row = tid >> 3), pieces 0..3 went to plane 0 and 4..7 to plane 1 at
    // the same row offset — the planes are a multiple of 256 B apart for the fragment reads — and every group took two
    // cycles: ~3e6 of the 5.8e6 SQ_LDS_BANK_CONFLICT cycles of a Go launch.  A group is now pieces 0..3 of rows r and r + 4
    // (4 x 80 B = 64 mod 128: the other half of the bank row; 4 x 144 B likewise in the epilogue's output tile), the next
    // group the same rows' pieces 4..7: lane bits (0,1) = piece low bits, 2 = row bit 2, (3,4) = row bits (0,1), 5 = the
    // plane, the wave = row bits (3,4).  (The plane on bit 5, not 3: the four 16-lane groups of the epilogue's ds_read_b128
    // read-back then meet half as many occupied slots of the output tile.)  A wave's load or store still covers the same
    // eight whole 128-byte rows.
    const int piece = SPLIT ? (tid & 7) : ((tid & 3) | ((tid >> 3) & 4));
    const int trow = SPLIT ? (tid >> 3) : (((tid >> 3) & 3) | (tid & 4) | ((tid >> 6) << 3));
    // f16: pieces 0..3 = channels [0, 32) -> plane 0, 4..7 -> plane 1.  SPLIT: pieces 0..3 = the hi halves of channel
    // pieces c = 0..3, 4..7 their lo halves: channel piece c lives in plane c & 1 at 16 (c >> 1), lo 32 bytes behind hi —
    // lane groups kq and kq + 1 then read the two planes at the same row offset, like the f16 instance
    const int ls_piece = SPLIT ? (piece & 1) * a.plane + ((piece >> 1) & 1) * 16 + (piece >> 2) * 32
                               : (piece >> 2) * a.plane + (piece & 3) * 16;
    // tile row r -> board b (of this workgroup), pixel q, image row; false for a padding row or a board beyond the batch
    // (every factor is below 2^24: v_mul_u32_u24 / v_mad_u32_u24 run at full rate, a 32-bit v_mul_lo_u32 at a quarter)
    auto locate = [&](int r, int &b, int &q, int &irow) __attribute__((always_inline)) {
        b = (int)(__umul24((unsigned)(r >> 4), a.inv_tpb) >> 16);     // tile / tpb (exact: tile < 24)
        q = r - (int)__umul24((unsigned)b, (unsigned)a.tpb * 16u);
        const int yy = (int)(__umul24((unsigned)q, a.inv_w) >> 16);   // q / w (exact: q < 512, w <= 32)
        // 16-byte slot of the pixel row: (yy + 1) * line16 + (q - yy * w + 1) * 5
        irow = (int)__umul24((unsigned)b, (unsigned)a.board16) + (int)__umul24((unsigned)(yy + 1), (unsigned)a.line16) +
               (q - (int)__umul24((unsigned)yy, (unsigned)a.w_) + 1) * 5;
        return b < a.bpw && q < a.hw && board0 + b < a.boards;
    };
#ifdef KZ_BC_REALTIME
    KZ_RSTAMP(25);
#else
    KZ_STAMP(25);
#endif
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16 *>(a.x), 0, a.bytes_x, 0x00020000);
    // po[i] = byte offset of slot i in the [pixels][ld] OUTPUT tensor (the residual and, for a convolution of more than
    // one chunk, the input have the same ld), or -1 for a padding row: 32-bit offsets on a uniform base keep the twelve
    // addresses in twelve registers.  The image rows (where a piece goes in LDS) are not kept across the k-loops — 12
    // registers the loop needs: they sit in LDS behind the image, 768 bytes, and are read back at every chunk boundary.
    // (SPLIT: the twelve offsets do not fit the registers next to 96 accumulators, the ring and three MFMA steps' worth of
    // fragments — the round-3 build spilled them and reloaded four of them from scratch inside every chunk's last k-steps,
    // each reload behind an s_waitcnt vmcnt(0) that drained the weight ring.  They sit in LDS instead, one row offset per
    // tile row behind the image-row table, read back with the LDS counter where they are needed.)
    int po[SPLIT ? 1 : 12];
#ifdef KZ_BCS_PO_REGS
    int po_dbg[12];
#endif
    auto po_of = [&](int i) __attribute__((always_inline)) -> int {
#ifdef KZ_BCS_PO_REGS
        if constexpr (SPLIT) return po_dbg[i];
#endif
        if constexpr (SPLIT) {
            const int base = *reinterpret_cast<const int *>(lds + a.rm_off + ROWS * 2 + (trow + i * 32) * 4);
            return base < 0 ? -1 : base + piece * 16;
        } else {
            return po[i];
        }
    };
    u32x4 v0[12];  // chunk 0
    int irow0[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        int b, q, irow;
        const bool ok = locate(trow + i * 32, b, q, irow);
        irow0[i] = irow;
#ifdef KZ_BC_L2_ABLATE
        // DIAGNOSTIC BUILD ONLY (tools/ab_go_l2.sh; results are wrong): every workgroup of XCD x stages from, adds and stores to
        // board x — the same instruction stream with the activation traffic served by the XCD's L2 instead of HBM
        const unsigned pix = __umul24((unsigned)((board0 + b) & 7), (unsigned)a.hw) + (unsigned)q;
#else
        const unsigned pix = __umul24((unsigned)(board0 + b), (unsigned)a.hw) + (unsigned)q;  // < boards * hw < 2^24
#endif
        // (SPLIT: a row is [hi 32 | lo 32] per group of 32 channels — a chunk is again 128 contiguous bytes, pieces 0..3 its
        // hi halves, 4..7 its lo halves, and the same offsets serve input, residual and output)
        const int po_i = ok ? (int)((__umul24(pix, (unsigned)a.ld) + (unsigned)piece * 8) * 2) : -1;
        if constexpr (!SPLIT) po[i] = po_i;
#ifdef KZ_BCS_PO_REGS
        po_dbg[i] = po_i;
#endif
        v0[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (int)((__umul24(pix, (unsigned)a.ldx) + (unsigned)piece * 8) * 2) : -1, 0, 0);
        if (piece == 0) {
            // (SPLIT: a padding row is marked in the table itself — there is no po[i] to ask)
            *reinterpret_cast<unsigned short *>(lds + a.rm_off + (trow + i * 32) * 2) = SPLIT && !ok ? (unsigned short)0xffff : (unsigned short)irow;
            if constexpr (SPLIT) *reinterpret_cast<int *>(lds + a.rm_off + ROWS * 2 + (trow + i * 32) * 4) = po_i;  // (piece 0: the row's own offset)
        }
        if constexpr (SPLIT) {
            if (!ok) irow0[i] = -1;
        }
    }

#ifndef KZ_BC_REALTIME
    KZ_STAMP(26);
#endif
    // weight ring: k-step g of this (layer, quarter): 4 KB = [nt 4][lane] x 16 B, the same for the four waves
    const uint4 *wp = a.w + (size_t)nquarter * total_ksteps * 256 + lane;
    uint4 wreg[PF][NTW];
#pragma unroll
    for (int s = 0; s < PF; s++) {
        const int g = s < total_ksteps ? s : total_ksteps - 1;
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) wreg[s][nt] = wp[(size_t)g * 256 + (wo * NTW + nt) * 64];
    }
    int g = 0;  // k-step counter; at a chunk boundary the ring holds k-steps g .. g + PF - 1

    KZ_STAMP(21);
    // zero the halo once (it is never written again; the pixel rows are overwritten by every chunk): per board and plane
    // the line above the board (line16 slots), the line below (5 * (pitch + 1) slots) and, per pixel line, the five slots of
    // its left halo row and the five of its right one (with skew 0 the next line's left halo)
    {
        const int nhb = a.line16 + 5 * (a.pitch + 1) + 10 * a.h;  // halo slots per board and plane
        for (int id = tid; id < ((KZ_BC_DIAG & 4) ? 0 : 2 * a.bpw * nhb); id += 256) {
            const int pb = (int)__umulhi((unsigned)id, a.inv_nhb), k = id - pb * nhb;  // id / nhb (plane-major, then board)
            const int pl = pb >= a.bpw ? 1 : 0, b = pb - pl * a.bpw;
            int slot16;
            if (k < a.line16) {
                slot16 = k;
            } else if (k < a.line16 + 5 * (a.pitch + 1)) {
                slot16 = (a.h + 1) * a.line16 + (k - a.line16);
            } else {
                const int kk = k - a.line16 - 5 * (a.pitch + 1);
                const int y = (int)(((unsigned)kk * a.inv_10) >> 16), pc = kk - y * 10;
                slot16 = (y + 1) * a.line16 + (pc < 5 ? pc : a.pitch * 5 + pc - 5);
            }
            *reinterpret_cast<uint4 *>(lds + pl * a.plane + (b * a.board16 + slot16) * 16) = make_uint4(0, 0, 0, 0);
        }
    }

    KZ_STAMP(22);
    // Centre-tap LDS address of this lane's fragment row for each of the wave's 6 tiles (plane kq & 1, 16-byte piece
    // kq >> 1 of the k-step); a lane without a pixel (padding row, missing board) reads pixel (0, 0) of board 0 — its
    // outputs are never stored.
    int T0[MTW];
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        int b, q, irow;
        const bool valid = locate((wr * MTW + i) * 16 + fr, b, q, irow);
        T0[i] = (valid ? irow : a.line16 + 5) * 16 + (kq & 1) * a.plane + (kq >> 1) * 16;
    }

    KZ_STAMP(24);
    f32x4 acc[NTW][MTW];
    {
        const int oc = nquarter * OCW + wo * NTW * 16 + kq * 4;
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + oc + nt * 16);
#pragma unroll
            for (int i = 0; i < MTW; i++) acc[nt][i] = b;
        }
    }

    // LDS address of this lane's fragment row per tile for one tap: a constant row offset in the halo image.
    // (pitch_prs is the line's bytes behind an optimisation barrier inside the chunk loop: the rows are the same for every
    // chunk, and the compiler would otherwise hoist all 9 x 12 of them out of the loop and spill them)
    auto tap_rows = [&](int tap, int lo, int hi, int pitch_prs, int (&T)[MT / NTW]) {
        const int off = (tap / 3 - 1) * pitch_prs + (tap % 3 - 1) * PRS;
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            if (i < lo || i >= hi) continue;
            T[i] = T0[i] + off;
        }
    };

    KZ_STAMP(1);
    // (wave-uniform) this wave's sixth tile lies beyond the workgroup's last pixel tile
#ifdef KZ_BC_NO_SKIP  // (diagnostic builds: the A/B of the padding-tile skip)
    const bool skip_last_tile = false;
#else
    const bool skip_last_tile = (wr * MTW + MTW - 1) >= a.bpw * a.tpb && (wr * MTW + MTW - 2) < a.bpw * a.tpb;
#endif
    const bool with_res = a.res != nullptr;
    const auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16 *>(with_res ? a.res : a.x), 0, with_res ? a.bytes : 0, 0x00020000);
    // ---- chunk 0 into the image: channels [0, 64) of this workgroup's boards, 8 pieces of 16 B per pixel row (never
    // into the halo: the halo clear of other threads needs no barrier in front of these writes) ----
    KZ_STAMP(2);
#pragma unroll
    for (int i = 0; i < 12; i++)
        if (SPLIT ? irow0[i] >= 0 : po[SPLIT ? 0 : i] >= 0) *reinterpret_cast<u32x4 *>(lds + irow0[i] * 16 + ls_piece) = v0[i];
    KZ_STAMP(3);
    for (int chunk = 0; chunk < chunks; chunk++) {
        __syncthreads();  // the chunk is staged
        if (KZ_BC_PRIO_MODE == 1) __builtin_amdgcn_s_setprio(0);
        if (KZ_BC_PRIO_MODE == 2) __builtin_amdgcn_s_setprio(3);
        KZ_STAMP(4 + (chunk & 3) * 4);
        // what the ring's dying stages fetch during the last PF k-steps of this chunk: the next chunk's image pieces, or
        // (last chunk) the residual's, or nothing
        // (no branch in the k-loop: one descriptor and one scalar offset, selected here; without a residual the last
        // chunk's descriptor has no records, so its tail loads return zeros without touching memory)
        const bool last_chunk = chunk + 1 == chunks;
        const auto trsrc = last_chunk ? rrsrc : xrsrc;
        const int tsoff = last_chunk ? nquarter * OCW * (SPLIT ? 4 : 2) : (chunk + 1) * 128;  // (a chunk is 128 bytes of a row either way)

        // A k-step in NP parts of three tiles: the MFMAs of one part run while the fragments of the next are read (the last
        // part reads the NEXT k-step's first); the other wave of the SIMD (the CU's second workgroup) fills whatever latency
        // is left.  (NTW = 4: two parts — the half-steps of rounds 1-5; NTW = 2: four.)
        // A wave whose last tile is pure padding (Go's 361 pixels are 22.6 tiles, so tile 23 holds none) does not issue that
        // tile's fragment read and NTW MFMAs: 4 % of the launch's MFMA work, and of its power.
        constexpr int HT = 3, NP = MTW / HT;
        static_assert(MTW % HT == 0 && NP % 2 == 0, "parts alternate between two fragment buffers; part 0 is always buffer 0");
        int T[MTW];
        h16x8 bf[2][HT] = {};
#ifndef KZ_BC_EARLY_W
#define KZ_BC_EARLY_W 1  // (0: the A/B build without the early weight step)
#endif
        // The ring's registers carry the next chunk's image pieces over the chunk boundary, so the next chunk's FIRST weight
        // step could only be requested behind the boundary's barriers and the first k-step then waited a whole L2 round trip
        // for it.  f16 instance (16 spare registers; the split instance has none): that one step is requested two k-steps
        // before the boundary into registers of its own and handed to the ring at the boundary.
        constexpr bool EARLY = !SPLIT && KZ_BC_EARLY_W;
        uint4 early[EARLY ? NTW : 1];
        int pitch_prs = a.line16 * 16;  // bytes from a line to the next
        asm volatile("" : "+s"(pitch_prs));
        tap_rows(0, 0, MTW, pitch_prs, T);
#pragma unroll
        for (int i = 0; i < HT; i++) bf[0][i] = lds_frag(T[i]);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int next_tap = tap < 8 ? tap + 1 : 8;
            // MFMA steps of a tap: f16 — two, each with its own weight step (channels [0,32) and [32,64) of the chunk);
            // SPLIT — three on the chunk's 32 channels: hi x W_hi, lo x W_hi (same weight step), hi x W_lo
            constexpr int KS = SPLIT ? 3 : 2;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const int wstep = SPLIT ? tap * 2 + (ks == 2 ? 1 : 0) : tap * 2 + ks;  // weight step within the chunk
                const int stage = wstep % PF;
                const bool stage_done = SPLIT ? ks != 0 : true;  // this MFMA step is the last one that reads the stage
                const int ao = SPLIT ? (ks == 1 ? 32 : 0) : ks * 32;  // where this step's fragments sit in a row (SPLIT: lo behind hi)
                const int ao_next = ks == KS - 1 ? 0 : SPLIT ? (ks + 1 == 1 ? 32 : 0) : (ks + 1) * 32;
                h16x8 af[NTW];  // aliases of the ring stage (no copy: the stage is reloaded after this k-step's MFMAs)
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) af[nt] = *reinterpret_cast<const h16x8 *>(&wreg[stage][nt]);
                static_for<NP>([&](auto P) __attribute__((always_inline)) {
                    constexpr int p = decltype(P)::value;  // (a constant expression: the scheduling barriers take immediates)
                    constexpr int cur = p & 1, nxt = cur ^ 1;
                    constexpr bool last_part = p == NP - 1;
                    // ---- the next part's fragments: tiles (p + 1) HT ..; behind the last part, the next k-step's first ----
                    // T is updated in place for the next tap as soon as a part's rows have been read for the last time
                    if (!last_part) {
#pragma unroll
                        for (int i = 0; i < HT; i++) {
                            const int t = (p + 1) * HT + i;
                            if (t != MTW - 1) bf[nxt][i] = lds_frag(T[t] + ao);
                        }
                        if (p + 1 == NP - 1 && !skip_last_tile) bf[nxt][HT - 1] = lds_frag(T[MTW - 1] + ao);
                        if (ks == KS - 1) tap_rows(next_tap, (p + 1) * HT, (p + 2) * HT, pitch_prs, T);
                    } else {
                        if (ks == KS - 1) tap_rows(next_tap, 0, HT, pitch_prs, T);
#pragma unroll
                        for (int i = 0; i < HT; i++) bf[nxt][i] = lds_frag(T[i] + ao_next);
                    }
                    // ---- this part's MFMAs (the wave's last tile, when it is padding, behind a wave-uniform branch) ----
#pragma unroll
                    for (int i = 0; i < HT; i++) {
                        const int t = p * HT + i;
                        if (t == MTW - 1) continue;
#pragma unroll
                        for (int nt = 0; nt < NTW; nt++)
                            acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf[cur][i], acc[nt][t], 0, 0, 0);
                    }
                    // the reads first: a fragment is then consumed >= 12 MFMAs (192 cycles; NTW = 2: 6) after its read was issued
                    __builtin_amdgcn_sched_group_barrier(SG_DS_READ, p + 1 == NP - 1 ? HT - 1 : HT, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, (last_part ? HT - 1 : HT) * NTW, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (last_part && !skip_last_tile) {
#pragma unroll
                        for (int nt = 0; nt < NTW; nt++)
                            acc[nt][MTW - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf[cur][HT - 1], acc[nt][MTW - 1], 0, 0, 0);
                    }
                    if (last_part) __builtin_amdgcn_sched_barrier(0);
                });
                // this stage's fragments have been issued to the MFMAs: refill it
                if (!stage_done) {
                } else if (wstep < KPC - PF) {  // (compile time) with the weights of k-step g + PF
#pragma unroll
                    for (int nt = 0; nt < NTW; nt++) wreg[stage][nt] = wp[(size_t)(g + PF) * 256 + (wo * NTW + nt) * 64];
                } else {  // last PF k-steps: pieces NTW jj .. NTW jj + NTW - 1 of the next image chunk / the residual
                    const int jj = wstep - (KPC - PF);
#pragma unroll
                    for (int nt = 0; nt < NTW; nt++)
                        wreg[stage][nt] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(trsrc, po_of(jj * NTW + nt), tsoff, 0));
                }
                if constexpr (EARLY) {
                    if (wstep == KPC - 2) {  // (compile time) k-step g + 2 = the next chunk's first (the last chunk: a harmless reload)
                        const int ge = last_chunk ? g : g + 2;
#pragma unroll
                        for (int nt = 0; nt < NTW; nt++) early[nt] = wp[(size_t)ge * 256 + (wo * NTW + nt) * 64];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (stage_done) g++;
            }
        }
        if (KZ_BC_PRIO_MODE == 1) __builtin_amdgcn_s_setprio(3);
        if (KZ_BC_PRIO_MODE == 2) __builtin_amdgcn_s_setprio(0);
        KZ_STAMP(5 + (chunk & 3) * 4);
        if (!last_chunk) {
            // ---- the next chunk's 12 pieces sit in the ring registers (piece 4 j + nt in stage j): into the image once
            // every wave is done with this chunk's fragments; then the ring takes the next chunk's first PF k-steps ----
            int erow[12];
#pragma unroll
            for (int i = 0; i < 12; i++) erow[i] = *reinterpret_cast<const unsigned short *>(lds + a.rm_off + (trow + i * 32) * 2);
            __syncthreads();
            KZ_STAMP(6 + (chunk & 3) * 4);
#pragma unroll
            for (int i = 0; i < 12; i++)
                if (SPLIT ? erow[i] != 0xffff : po[SPLIT ? 0 : i] >= 0) *reinterpret_cast<uint4 *>(lds + erow[i] * 16 + ls_piece) = wreg[i / NTW][i % NTW];  // never into the halo
#pragma unroll
            for (int st = 0; st < PF; st++)
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) {
                    if (EARLY && st == 0) wreg[0][nt] = early[nt];
                    else wreg[st][nt] = wp[(size_t)(g + st) * 256 + (wo * NTW + nt) * 64];
                }
            KZ_STAMP(7 + (chunk & 3) * 4);
        }
    }

    if constexpr (SPLIT) {
        // ---- epilogue, split arithmetic: [relu]; [+ residual (hi + lo)]; [final BN]; -> the (hi, lo) f16 halves of the
        // output row, or f32 for the heads.  Through the same output tile in LDS as the f16 epilogue below, in two passes of
        // 32 output channels: a tile row is [hi 64 B | lo 64 B] (or 32 f32), so its eight 16-byte pieces are again what a
        // thread's slot moves, coalesced, between registers and HBM.  The residual of pass 0 arrived in the ring registers
        // during the last three weight steps; that of pass 1 is requested as soon as those registers are free.
        KZ_STAMP(18);
        const auto yrsrc = a.y32 ? __builtin_amdgcn_make_buffer_rsrc(a.y32, 0, a.bytes32, 0x00020000)
                                 : __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.bytes, 0x00020000);
        const int out_lds = trow * ORS + piece * 16;  // + i * 32 * ORS
        const float floor_ = a.relu ? 0.0f : -__builtin_inff();
        constexpr int NH = NTW / 2;
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            __syncthreads();  // pass 0: every wave is done with the last chunk's fragments; pass 1: with pass 0's tile
            if (with_res) {
#pragma unroll
                for (int i = 0; i < 12; i++) *reinterpret_cast<uint4 *>(lds + out_lds + i * 32 * ORS) = wreg[i / NTW][i % NTW];
                if (pass == 0) {
#pragma unroll
                    for (int i = 0; i < 12; i++)
                        wreg[i / NTW][i % NTW] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, po_of(i), nquarter * OCW * 4 + 128, 0));
                }
                __syncthreads();
            }
            // One tile row at a time: the residual pieces of BOTH 16-channel tiles of the row are read before the row is
            // rewritten.  (An f32 result row — the tower's last layer, a.y32 — covers the bytes the other tile's (hi, lo)
            // residual occupies, so all of a row's residual reads must come first; reading all six rows' pieces up front, as
            // the round-3 build did, took 48 registers next to the accumulators, the ring — which holds pass 1's residual
            // by now — and the slot offsets, and that build spilled.  Six dependent LDS round trips per pass instead of
            // one: ~1 us of a 480 us launch.)
            f32x4 ps[NH], pt[NH];
#pragma unroll
            for (int n2 = 0; n2 < NH; n2++) {
                const int oc = nquarter * OCW + (pass * NH + n2) * 16 + kq * 4;
                ps[n2] = f32x4{1.f, 1.f, 1.f, 1.f};
                pt[n2] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.post_scale) {
                    ps[n2] = *reinterpret_cast<const f32x4 *>(a.post_scale + oc);
                    pt[n2] = *reinterpret_cast<const f32x4 *>(a.post_shift + oc);
                }
            }
#pragma unroll
            for (int i = 0; i < MTW; i++) {
                unsigned char *row = lds + ((wr * MTW + i) * 16 + fr) * ORS;  // this lane's 4 channels per tile of it: owned by the lane
                u32x2 rh[NH], rl[NH];
                if (with_res) {
#pragma unroll
                    for (int n2 = 0; n2 < NH; n2++) {
                        rh[n2] = *reinterpret_cast<const u32x2 *>(row + (n2 * 16 + kq * 4) * 2);
                        rl[n2] = *reinterpret_cast<const u32x2 *>(row + 64 + (n2 * 16 + kq * 4) * 2);
                    }
                }
#pragma unroll
                for (int n2 = 0; n2 < NH; n2++) {
                    f32x4 v = acc[pass * NH + n2][i];
#pragma unroll
                    for (int j = 0; j < 4; j++) asm("v_max_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(floor_));  // [relu]
                    if (with_res) {
                        const h16x4 h = __builtin_bit_cast(h16x4, rh[n2]), l = __builtin_bit_cast(h16x4, rl[n2]);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] += (float)h[j] + (float)l[j];  // hi + lo is exact in f32; AFTER the ReLU (post_act.py:227-228)
                    }
                    v = v * ps[n2] + pt[n2];
                    if (a.y32) {
                        *reinterpret_cast<f32x4 *>(row + (n2 * 16 + kq * 4) * 4) = v;
                    } else {
                        h16x4 hi, lo;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            hi[j] = (h16)v[j];
                            lo[j] = (h16)(v[j] - (float)hi[j]);
                        }
                        *reinterpret_cast<h16x4 *>(row + (n2 * 16 + kq * 4) * 2) = hi;
                        *reinterpret_cast<h16x4 *>(row + 64 + (n2 * 16 + kq * 4) * 2) = lo;
                    }
                }
            }
            __syncthreads();
            // a padding row's store is out of range and dropped.  The pass's 32 channels are 128 contiguous bytes of the output
            // row in either form — [hi 32 | lo 32] f16 or 32 f32, and an f32 row is as long as a (hi, lo) row — so the slot
            // offsets po[i] serve both
#pragma unroll
            for (int i = 0; i < 12; i++)
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(lds + out_lds + i * 32 * ORS), yrsrc, po_of(i),
                                                       nquarter * OCW * 4 + pass * 128, 0);
        }
        KZ_STAMP(19);
        KZ_STAMP(20);
        return;
    }
    // ---- epilogue: [relu]; [+ residual]; [final BN]; -> f16 -> NHWC rows in global memory ----
    // The result is staged through LDS (the image is dead now) so that HBM sees whole 128-byte lines (this workgroup's
    // 64 output channels of a pixel) instead of 8-byte pieces: O[row][64 oc] f16, row stride 144 B.  The residual arrived
    // in the ring registers as the same coalesced 16-byte pieces during the last three k-steps: it goes into O first, and
    // each lane then replaces the 8 bytes it owns (its 4 channels of a pixel row) by relu(acc) + residual, added in f32.
    __syncthreads();  // every wave is done with the last chunk's fragments
    KZ_STAMP(18);
    const int out_lds = trow * ORS + piece * 16;  // + i * 32 * ORS
    // The output tile keeps the two 8-byte halves of every 16-byte piece SWAPPED in the rows whose bit 3 is set: a lane's four
    // channels are 8 bytes, a ds_write_b64 / ds_read_b64 group is the 16 rows of a tile at one channel offset, and at a
    // 144-byte row stride rows r and r + 8 share a bank — with the swap they take the two halves of the slot instead (the
    // 2-way conflicts of the epilogue's 24 writes and 24 residual reads per lane are gone).  For the lanes of the MFMA layout
    // that is kq ^ (fr >> 3), a constant; for the coalesced 16-byte side (residual in, rows out) the row's bit 3 is the
    // wave's bit 0: waves 1 and 3 swap the halves of what they stage and of what they store.
    const bool swap_halves = wave & 1;
    auto halves = [&](uint4 v) __attribute__((always_inline)) { return swap_halves ? make_uint4(v.z, v.w, v.x, v.y) : v; };
    if (with_res) {
#pragma unroll
        for (int i = 0; i < 12; i++) *reinterpret_cast<uint4 *>(lds + out_lds + i * 32 * ORS) = halves(wreg[i / NTW][i % NTW]);
        KZ_STAMP(23);
        __syncthreads();
    }
    KZ_STAMP(27);
    // All 24 residual slots are read before the first one is rewritten — one LDS round trip instead of 24 in a row (the
    // compiler cannot move a slot's read past the previous slot's write by itself) — and the launch's wave-uniform
    // options pick one of four straight-line bodies instead of branching at every slot.
    auto finish = [&](auto RES, auto POST) __attribute__((always_inline)) {
        constexpr bool with_residual = decltype(RES)::value, post = decltype(POST)::value;
        const float floor_ = a.relu ? 0.0f : -__builtin_inff();
        const int kqs = kq ^ (fr >> 3);  // where this lane's four channels sit in the output tile (halves swapped in rows 8..15 of a tile)
        h16x4 rres[NTW][MTW];
        if constexpr (with_residual) {
#pragma unroll
            for (int nt = 0; nt < NTW; nt++)
#pragma unroll
                for (int i = 0; i < MTW; i++)
                    rres[nt][i] = (KZ_BC_DIAG & 8) ? h16x4{} : *reinterpret_cast<const h16x4 *>(lds + ((wr * MTW + i) * 16 + fr) * ORS + ((wo * NTW + nt) * 16 + kqs * 4) * 2);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            const int ocl = (wo * NTW + nt) * 16 + kq * 4;  // within this workgroup's 64 channels
            f32x4 ps = f32x4{1.f, 1.f, 1.f, 1.f}, pt = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (post) {
                ps = *reinterpret_cast<const f32x4 *>(a.post_scale + nquarter * OCW + ocl);
                pt = *reinterpret_cast<const f32x4 *>(a.post_shift + nquarter * OCW + ocl);
            }
#pragma unroll
            for (int i = 0; i < MTW; i++) {
                unsigned char *slot = lds + ((wr * MTW + i) * 16 + fr) * ORS + ((wo * NTW + nt) * 16 + kqs * 4) * 2;  // owned by exactly this lane
                f32x4 v = acc[nt][i];
#pragma unroll
                for (int j = 0; j < 4; j++) asm("v_max_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(floor_));  // [relu] (one instruction: no compare mask, no canonicalising copy)
                if constexpr (with_residual && !post) {
                    // relu(acc) + residual, added in f32 AFTER the ReLU (post_act.py:227-228) and rounded to f16 once:
                    // v_fma_mixlo/mixhi_f16 take the f16 residual and the f32 sum in one instruction each
                    unsigned lo01, lo23;  // (mixlo keeps the destination's high half: whatever it was, mixhi overwrites it)
                    const unsigned r01 = reinterpret_cast<const unsigned *>(&rres[nt][i])[0], r23 = reinterpret_cast<const unsigned *>(&rres[nt][i])[1];
                    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo01) : "v"(r01), "v"(v[0]));
                    asm("v_fma_mixhi_f16 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo01) : "v"(r01), "v"(v[1]));
                    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo23) : "v"(r23), "v"(v[2]));
                    asm("v_fma_mixhi_f16 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo23) : "v"(r23), "v"(v[3]));
                    if (!(KZ_BC_DIAG & 1)) *reinterpret_cast<uint2 *>(slot) = make_uint2(lo01, lo23);
                } else {
                    if constexpr (with_residual) {
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] += (float)rres[nt][i][j];  // added in f32, AFTER the ReLU
                    }
                    if constexpr (post) v = v * ps + pt;
                    if (!(KZ_BC_DIAG & 1)) *reinterpret_cast<h16x4 *>(slot) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                    else asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
                }
            }
        }
    };
    if (with_res) {
        if (a.post_scale) finish(std::true_type{}, std::true_type{});
        else finish(std::true_type{}, std::false_type{});
    } else {
        if (a.post_scale) finish(std::false_type{}, std::true_type{});
        else finish(std::false_type{}, std::false_type{});
    }
    KZ_STAMP(28);
    __syncthreads();
    KZ_STAMP(19);
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 12; i++)  // a padding row's store is out of range and dropped
        __builtin_amdgcn_raw_buffer_store_b128((KZ_BC_DIAG & 2) ? u32x4{} : __builtin_bit_cast(u32x4, halves(*reinterpret_cast<const uint4 *>(lds + out_lds + i * 32 * ORS))), yrsrc, po[i],
                                               nquarter * OCW * 2, KZ_BC_STORE_AUX);
    KZ_STAMP(20);
#ifdef KZ_BC_REALTIME
    KZ_RSTAMP(26);
#endif
}

__global__ __launch_bounds__(256, 2) void kz_board_conv_f16(BoardConvDev a) { board_conv_body<false, KZ_BC_NTW, 12 / KZ_BC_NTW>(a); }
__global__ __launch_bounds__(256, 2) void kz_board_conv_split16(BoardConvDev a) { board_conv_body<true, 4, 3>(a); }

}  // namespace

namespace {
// halo image geometry: boards per workgroup limited by the 24 tiles and by 40 KB per plane (two workgroups per CU)
struct Geometry {
    int tpb, bpw, pitch, skew16, line16, board16, plane, rm_off, lds_bytes;
};
constexpr int RM_BYTES = ROWS * 2 + ROWS * 4;  // image-row table (u16) + the split instance's row-offset table (i32)
int boards_per_workgroup(int h, int w, int skew16) {
    const int tpb = (h * w + 15) / 16, line16 = 5 * (w + 1) + skew16, board_bytes = ((h + 2) * line16 + 5) * 16;
    const int by_tiles = tpb <= MT ? MT / tpb : 0, by_lds = ((LDS_MAX - RM_BYTES) / 2) / board_bytes;
    return by_tiles < by_lds ? by_tiles : by_lds;
}
Geometry geometry(int h, int w) {
    Geometry g{};
    g.tpb = (h * w + 15) / 16;
    g.pitch = w + 1;
    // the 11-slot gap per line that keeps tiles across a line end conflict-free (top of the file) — unless it costs this
    // board size a board per workgroup
    g.skew16 = boards_per_workgroup(h, w, 11) == boards_per_workgroup(h, w, 0) ? 11 : 0;
    g.line16 = 5 * g.pitch + g.skew16;
    g.board16 = (h + 2) * g.line16 + 5;
    g.bpw = boards_per_workgroup(h, w, g.skew16);
    g.plane = (g.bpw * g.board16 * 16 + 255) / 256 * 256;
    g.rm_off = 2 * g.plane > ROWS * ORS ? 2 * g.plane : ROWS * ORS;  // the epilogue reuses the image for the output tile
    g.lds_bytes = g.rm_off + RM_BYTES;
    return g;
}
}  // namespace

bool board_conv_supported(int dtype, int h, int w, int cin, int cout) {
    return dtype == 1 && cin % CH == 0 && cout % OCW == 0 && w <= 32 && h <= 32 && w >= 2 && h >= 2 &&
           geometry(h, w).bpw >= 1;
}

int board_conv_workgroups(int boards, int h, int w, int cout) {
    const int bpw = geometry(h, w).bpw;
    return bpw ? ((boards + bpw - 1) / bpw) * (cout / OCW) : 0;
}

size_t board_conv_weight_elems(int cin, int cout) { return (size_t)9 * cin * cout; }

// OIHW f32 (BN folded) -> [n_quarter][chunk][tap][ks 2][nt 4][lane 64][8] f16: element j of lane (fr, kq) is
// W[oc = 64*n_quarter + 16*nt + fr][channel = 64*chunk + 32*(kq&1) + 16*ks + 8*(kq>>1) + j][tap]
void board_conv_pack_weights(const float *oihw, int cout, int cin, uint16_t *dst) {
    const int chunks = cin / CH, quarters = cout / OCW;
    size_t o = 0;
    for (int nq = 0; nq < quarters; nq++)
        for (int chunk = 0; chunk < chunks; chunk++)
            for (int tap = 0; tap < 9; tap++)
                for (int ks = 0; ks < 2; ks++)
                    for (int nt = 0; nt < OCW / 16; nt++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int j = 0; j < 8; j++) {
                                const int kq = lane >> 4;
                                const int oc = OCW * nq + 16 * nt + (lane & 15);
                                const int ch = CH * chunk + 32 * (kq & 1) + 16 * ks + 8 * (kq >> 1) + j;
                                const _Float16 hv = (_Float16)oihw[((size_t)oc * cin + ch) * 9 + tap];
                                uint16_t bits;
                                __builtin_memcpy(&bits, &hv, 2);
                                dst[o++] = bits;
                            }
}

// ---- split arithmetic: tensors of [pixels][C / 32][hi 32 | lo 32] f16 rows, 32-channel chunks ----
bool board_conv_split_supported(int h, int w, int cin, int cout) {
    return cin == cout && cout % OCW == 0 && w <= 32 && h <= 32 && w >= 2 && h >= 2 && geometry(h, w).bpw >= 1;
}

size_t board_conv_split_weight_elems(int cin, int cout) { return (size_t)2 * 9 * cin * cout; }

// OIHW f32 (BN folded) -> [n_quarter][chunk][tap][hi, lo][nt 4][lane 64][8] f16: element j of lane (fr, kq) is the hi
// (lo) half of W[oc = 64*n_quarter + 16*nt + fr][channel = 32*chunk + 8*kq + j][tap]; hi = f16(w), lo = f16(w - hi)
void board_conv_split_pack_weights(const float *oihw, int cout, int cin, uint16_t *dst) {
    const int chunks = cin / 32, quarters = cout / OCW;
    size_t o = 0;
    for (int nq = 0; nq < quarters; nq++)
        for (int chunk = 0; chunk < chunks; chunk++)
            for (int tap = 0; tap < 9; tap++)
                for (int half = 0; half < 2; half++)
                    for (int nt = 0; nt < OCW / 16; nt++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int j = 0; j < 8; j++) {
                                const int oc = OCW * nq + 16 * nt + (lane & 15);
                                const int ch = 32 * chunk + 8 * (lane >> 4) + j;
                                const float wv = oihw[((size_t)oc * cin + ch) * 9 + tap];
                                const _Float16 hi = (_Float16)wv;
                                const _Float16 hv = half ? (_Float16)(wv - (float)hi) : hi;
                                uint16_t bits;
                                __builtin_memcpy(&bits, &hv, 2);
                                dst[o++] = bits;
                            }
}

namespace {
void launch_board_conv_any(const BoardConvArgs &t, bool split, hipStream_t stream) {
    BoardConvDev d;
    d.y32 = t.y32;
    d.ld32 = t.ldy32;
    d.bytes32 = (int)((size_t)t.boards * t.h * t.w * t.ldy32 * 4);
    d.x = static_cast<const h16 *>(t.x);
    d.w = static_cast<const uint4 *>(t.weights);
    d.bias = t.bias;
    d.post_scale = t.post_scale;
    d.post_shift = t.post_shift;
    d.res = static_cast<const h16 *>(t.res);
    d.y = static_cast<h16 *>(t.y);
    d.bytes = (int)((size_t)t.boards * t.h * t.w * t.ldy * 2);
    d.bytes_x = (int)((size_t)t.boards * t.h * t.w * t.ldx * 2);
    d.ld = t.ldy;
    d.ldx = t.ldx;  // (ldx != ldy only for a single-chunk convolution: the stem)
    d.boards = t.boards;
    d.h = t.h;
    d.w_ = t.w;
    d.hw = t.h * t.w;
    const Geometry geo = geometry(t.h, t.w);
    d.tpb = geo.tpb;
    d.bpw = geo.bpw;
    d.pitch = geo.pitch;
    d.line16 = geo.line16;
    d.board16 = geo.board16;
    d.plane = geo.plane;
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
    const unsigned nhb = (unsigned)geo.line16 + 5 * ((unsigned)geo.pitch + 1) + 10 * (unsigned)t.h;  // halo slots per board and plane
    d.inv_nhb = (unsigned)(((1ull << 32) + nhb - 1) / nhb);  // __umulhi(id, inv) == id / nhb for id * nhb < 2^32
    d.inv_10 = (65536u + 9) / 10;
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_board_conv_f16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)kz_board_conv_split16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done_mask |= 1ull << (dev & 63);
    }
    const int grid = ((d.groups + 7) / 8) * 8 * d.nq;
#ifdef KZ_BC_STAMPS
#ifdef KZ_BC_REALTIME
    constexpr int KEPT = 4;  // the stamps of launches stamp_launch .. stamp_launch + 3, one after the other
#else
    constexpr int KEPT = 1;
#endif
    static unsigned long long *stamp_buf = nullptr;
    static int launches = 0;
    const size_t stamp_bytes = (size_t)grid * 4 * 32 * sizeof(unsigned long long);
    if (!stamp_buf) (void)hipMalloc((void **)&stamp_buf, (size_t)8192 * 4 * 32 * 8 * KEPT);
    static const int stamp_launch = getenv("KZ_BC_STAMP_LAUNCH") ? atoi(getenv("KZ_BC_STAMP_LAUNCH")) : 20;  // 20: a layer with a residual
    d.stamps = stamp_buf + (size_t)((launches - stamp_launch) & (KEPT - 1)) * (stamp_bytes / 8);
    if (launches == stamp_launch) (void)hipMemsetAsync(stamp_buf, 0, stamp_bytes * KEPT, stream);
#else
    d.stamps = nullptr;
#endif
    if (split) kz_board_conv_split16<<<grid, 256, geo.lds_bytes, stream>>>(d);
    else kz_board_conv_f16<<<grid, 256, geo.lds_bytes, stream>>>(d);
#ifdef KZ_BC_STAMPS
    if (launches++ == stamp_launch + KEPT - 1 && getenv("KZ_BC_STAMP_FILE")) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> host(stamp_bytes / 8 * KEPT);
        (void)hipMemcpy(host.data(), stamp_buf, stamp_bytes * KEPT, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(getenv("KZ_BC_STAMP_FILE"), "wb")) {
            fwrite(host.data(), 1, stamp_bytes * KEPT, f);
            fclose(f);
        }
    }
#endif
}
}  // namespace

void launch_board_conv(const BoardConvArgs &t, hipStream_t stream) { launch_board_conv_any(t, false, stream); }
void launch_board_conv_split(const BoardConvArgs &t, hipStream_t stream) { launch_board_conv_any(t, true, stream); }

}  // namespace kz
