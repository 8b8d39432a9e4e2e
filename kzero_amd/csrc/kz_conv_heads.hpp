// kz_conv_heads.hpp — the scalar head and the conv policy heads (ConvPolicyHead, AtaxxConvPolicyHead) on two images (the tower
// output and the policy head's hidden layer) that already sit in LDS: the tail of the one-launch networks.  Shared by the
// exact-f32 launch (kz_tower_f32.hip), the split-f16 launch (kz_tower_split.hip, which converts its (hi, lo) images to f32
// rows first) and the plain-f16 launch (which runs the two small convolutions on its f16 images).  Arithmetic follows
// python/lib/model/post_act.py:8-31 (scalar head) and :75-110 (conv policy heads).
//
// Device code only; included INSIDE `namespace kz { namespace {` of a .hip file, after f32x4 is defined.  `Dev` is the
// launch's argument struct: it must have the members hc, hs, pc, policy_len, zero_tail, extra, epoch, hw, nb, inv_hw,
// sh_b0, sh_w1t, sh_b1, sh_w2, sh_b2, p_b1, pe_bc, pe_wl, pe_bl, small_w (tower32_pack_small_weights), scalars, policy,
// nonfinite_flag, dec (DecodeDev, kz_decode_dev.hpp — included before this file).
#pragma once

#ifndef KZ_HEADS_STAMP
#define KZ_HEADS_STAMP(slot) do { } while (0)
#endif

template <int N>
struct IntC {
    static constexpr int value = N;
};

template <int C>
__device__ __forceinline__ int plane_of(int kq) {  // byte offset of lane group kq's 16-byte piece within a step
    return C == 256 ? 256 * kq : 256 * (kq & 1) + 128 * (kq >> 1);
}

// The tail behind the two small convolutions' operands: small_conv(which, emit) runs the 1x1 convolution with at most 32
// output channels over the tower output (which = 0: the scalar head's filters and the extra-move filter) or over the policy
// head's hidden layer (which = 1: the policy planes) on the matrix cores and calls emit(mt, q, row, value) for this lane's
// results — output channel mt * 16 + kq * 4 + q of pixel row `row` (the accumulator layout of a 16x16 MFMA with the weights as
// the row operand), rows of this workgroup's boards only.  Two providers: conv_heads_f32 below (f32 row images, exact-f32
// MFMAs: the exact-f32 and the split launches) and the plain-f16 launch's own (kz_tower_split.hip: f16 MFMAs straight on its
// f16 images).
// stage / stage_bytes: an LDS region that is dead once both small convolutions have run (the tower output's image), for
// the in-launch decode_output's staging (a.dec.move_offsets set; 0 bytes: it re-reads instead).
template <int C, int NT, typename Dev, typename SmallConv>
__device__ __forceinline__ void conv_heads_tail(const Dev &a, unsigned char *lds, int scratch, int board0, int boards,
                                                int rows_valid, SmallConv small_conv, int stage = 0, int stage_bytes = 0) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kq = lane >> 4;
    // the zero rows are dead by now: scratch for the scalar head
    const int n_in = a.hc * a.hw, nseg = 256 / a.hs;
    float *sact = reinterpret_cast<float *>(lds + scratch);  // [nb][hc*hw] channel-major like nn.Flatten on NCHW (post_act.py:16)
    float *sext = sact + a.nb * n_in;                     // [nb][hw] extra-move plane
    float *shid = sext + a.nb * a.hw;                     // [nb][hs]
    float *sw2 = shid + a.nb * a.hs;                      // [5][hs]
    float *sred = sw2 + 5 * a.hs;                         // [nseg][nb*hs]
    // Everything the tail reads from global memory is requested up front — a dependent L2 round trip per use costs
    // ~1 us with one wave per SIMD, and there were forty of them in a row: the biases of the two small convolutions
    // for this lane's output channels, the first 32 of this thread's Linear weights, the hidden layer's bias
    float sb[2][4], pb[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int oc = mt * 16 + kq * 4 + q;
            sb[mt][q] = oc < a.hc ? a.sh_b0[oc] : (oc == a.hc && a.extra) ? a.pe_bc[0] : 0.0f;
            pb[mt][q] = oc < a.pc ? a.p_b1[oc] : 0.0f;
        }
    const int fseg = tid / a.hs, fj = tid - fseg * a.hs;
    constexpr int WPRE = 32;
    float wpre[WPRE];
#pragma unroll
    for (int k = 0; k < WPRE; k++) {
        const int i = fseg + k * nseg, ic = i < n_in ? i : n_in - 1;  // (fseg >= nseg: a thread without a segment reads row ic too)
        const float v = a.sh_w1t[(size_t)ic * a.hs + fj];
        wpre[k] = i < n_in ? v : 0.0f;
    }
    const float b1v = tid < a.nb * a.hs ? a.sh_b1[tid % a.hs] : 0.0f;
    const float b2v = tid < boards * 5 ? a.sh_b2[tid % 5] : 0.0f;
    for (int i = tid; i < 5 * a.hs; i += 256) sw2[i] = a.sh_w2[i];
    KZ_HEADS_STAMP(56);
    // scalar head Conv1x1 C->hc + ReLU and the extra moves' Conv1x1 C->1 (post_act.py:8-31, :86-96): one small conv over x
    bool bad = false;  // a non-finite sum = a non-finite value somewhere in this board's tower output
    small_conv(0, [&](int mt, int q, int row, float v) {
        const int oc = mt * 16 + kq * 4 + q;
        const int bb = (int)(((unsigned)row * a.inv_hw) >> 16), p = row - bb * a.hw;
        if (oc < a.hc) {
            bad |= !(fabsf(v) <= 3.0e38f);
            sact[bb * n_in + oc * a.hw + p] = fmaxf(v + sb[mt][q], 0.0f);
        } else if (oc == a.hc && a.extra) {
            sext[row] = v + sb[mt][q];
        }
    });
    KZ_HEADS_STAMP(57);
    if (bad && a.nonfinite_flag) *reinterpret_cast<volatile int *>(a.nonfinite_flag) = a.epoch;  // (plain store: the flag may live in pinned host memory)
    // policy (post_act.py:75-110): Conv1x1 C->pc on the hidden layer, channel-major flatten
    small_conv(1, [&](int mt, int q, int row, float v) {
        const int oc = mt * 16 + kq * 4 + q;
        const int bb = (int)(((unsigned)row * a.inv_hw) >> 16), p = row - bb * a.hw;
        if (oc < a.pc) a.policy[(size_t)(board0 + bb) * a.policy_len + oc * a.hw + p] = v + pb[mt][q];
    });
    // AtaxxConvPolicyHead appends a constant-zero pass logit (post_act.py:106-110)
    if (a.zero_tail && tid < boards) a.policy[(size_t)(board0 + tid) * a.policy_len + a.pc * a.hw] = 0.0f;
    KZ_HEADS_STAMP(60);
    __syncthreads();
    // Linear(n_in -> hs): thread (segment, output) walks every nseg-th input of the transposed matrix (the hs weights of
    // one input are contiguous) for all boards of the workgroup at once; the segments meet through LDS
    {
        const int seg = fseg, j = fj;
        // branch free: an input past the end has weight 0 and reads the last activation; a board past a.nb repeats the last
        // board and is not stored (with a branch per term the LDS reads waited for each other: 350 cycles per input)
        auto narrow = [&](auto nbc) {
            constexpr int NBB = decltype(nbc)::value;
            float part[NBB];
            int boff[NBB];
#pragma unroll
            for (int bb = 0; bb < NBB; bb++) {
                part[bb] = 0.0f;
                boff[bb] = (bb < a.nb ? bb : a.nb - 1) * n_in;
            }
#pragma unroll
            for (int k = 0; k < WPRE; k++) {
                const int i = seg + k * nseg, ic = i < n_in ? i : n_in - 1;
#pragma unroll
                for (int bb = 0; bb < NBB; bb++) part[bb] += wpre[k] * sact[boff[bb] + ic];
                if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // (eight inputs' LDS reads in flight, not all of them)
            }
            // the rest in batches of WB loads, the next batch requested before this one is used: a thread of a wide head
            // (ScalarHead(8, 128) on a 9x9 board: 324 inputs per thread) waited for ~40 L2 round trips in a row when the
            // loop asked for eight weights at a time
            constexpr int WB = 32;
            auto fetch = [&](float (&wv)[WB], int i0) {
#pragma unroll
                for (int k = 0; k < WB; k++) {
                    const int i = i0 + k * nseg, ic = i < n_in ? i : n_in - 1;
                    const float v = a.sh_w1t[(size_t)ic * a.hs + j];
                    wv[k] = i < n_in ? v : 0.0f;
                }
            };
            int i0 = seg + WPRE * nseg;
            if (i0 < n_in) {
                float wcur[WB];
                fetch(wcur, i0);
                for (; i0 < n_in; i0 += WB * nseg) {
                    float wnext[WB];
                    fetch(wnext, i0 + WB * nseg);  // (zeros past the end)
#pragma unroll
                    for (int k = 0; k < WB; k++) {
                        const int i = i0 + k * nseg, ic = i < n_in ? i : n_in - 1;
#pragma unroll
                        for (int bb = 0; bb < NBB; bb++) part[bb] += wcur[k] * sact[boff[bb] + ic];
                        if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int k = 0; k < WB; k++) wcur[k] = wnext[k];
                }
            }
#pragma unroll
            for (int bb = 0; bb < NBB; bb++)
                if (bb < a.nb) sred[(seg * a.nb + bb) * a.hs + j] = part[bb];
        };
        if (seg < nseg) {
            if (a.nb <= 2) narrow(IntC<2>{});
            else narrow(IntC<4>{});
        }
    }
    KZ_HEADS_STAMP(58);
    if (a.extra) {  // Linear(hw -> extra) behind the policy planes
        for (int o = tid >> 2; o < boards * a.extra; o += 64) {
            const int bb = o / a.extra, j = o - bb * a.extra, seg = tid & 3;
            float s = 0.0f;
#pragma unroll 8
            for (int p = seg; p < a.hw; p += 4) s += a.pe_wl[(size_t)j * a.hw + p] * sext[bb * a.hw + p];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            if (seg == 0) a.policy[(size_t)(board0 + bb) * a.policy_len + a.pc * a.hw + j] = s + a.pe_bl[j];
        }
    }
    __syncthreads();
    // (nb * hs may exceed the 256 threads — e.g. three 6x6 boards with a hidden size of 96: every hidden unit of every
    // board gets a turn; the first turn's bias was fetched before the small convolutions)
    for (int o = tid; o < a.nb * a.hs; o += 256) {
        float s = o == tid ? b1v : a.sh_b1[o % a.hs];
        for (int seg = 0; seg < nseg; seg++) s += sred[seg * a.nb * a.hs + o];
        shid[o] = fmaxf(s, 0.0f);
    }
    __syncthreads();
    KZ_HEADS_STAMP(59);
    const bool decode = a.dec.move_offsets != nullptr;
    if (tid < boards * 5) {
        const int bb = tid / 5, j = tid - bb * 5;
        float s = b2v;
        for (int i = 0; i < a.hs; i++) s += sw2[j * a.hs + i] * shid[bb * a.hs + i];
        if (decode) sred[bb * 8 + j] = s;  // (sred is dead: the hidden layer has been reduced)
        else a.scalars[(size_t)(board0 + bb) * 5 + j] = s;
    }
    if (decode) {
        // decode_output (common.rs:16-100) as the launch's last step: the logits were written to a.policy (device memory) by
        // this workgroup above; wave w gathers and normalises the available moves of boards w, w + 4, ...
        __threadfence();
        __syncthreads();
        // (four waves share the staging region, a quarter each, and take boards w, w + 4, ...: every instantiation of this tail
        // is a 256-thread workgroup — a wider one would write past the region and decode boards twice)
        if (blockDim.x != 256) __builtin_trap();
        const int wv = tid >> 6, cap = stage_bytes / 16;  // floats per wave
        float *st = reinterpret_cast<float *>(lds + stage) + wv * cap;
        for (int bb = wv; bb < boards; bb += 4) {
            const float *lg = a.policy + (size_t)(board0 + bb) * a.policy_len;
            decode_board_wave(a.dec, board0 + bb, lane, sred + bb * 8, st, cap, [&](int idx) {
                // (a coherent load: the line may sit in this CU's vector cache from before the stores above)
                return __hip_atomic_load(lg + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
        }
    }
}

// Images: rows of C f32 (natural channel order), row stride RS = 4 C + 16 bytes, NT tiles of 16 rows; `xin` = LDS offset of
// the tower output (after the final BN), `hin` = of the policy head's hidden layer (Conv1x1 C->C + ReLU), `scratch` = of
// 16 * RS bytes the tail may overwrite (the f32 launch's zero rows).  board0 / boards / rows_valid: this workgroup's
// boards; rows_img: rows an image really holds (a tile row behind them is read as the last row and never emitted).
// Every thread of the 256 calls it; the caller has synchronised the workgroup behind the images' last writes.
template <int C, int NT, typename Dev>
__device__ __forceinline__ void conv_heads_f32(const Dev &a, unsigned char *lds, int scratch, int xin, int hin, int board0,
                                               int boards, int rows_valid, int rows_img = NT * 16) {
    constexpr int G = C / 16, RS = C * 4 + 16;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int koff = plane_of<C>(kq);
    // A 1x1 convolution with at most 32 output channels (two 16-channel tiles, zero-padded by the host) on the MFMAs:
    // the row tiles are split over the waves (wave w: tiles w, w + 4 and w + 8), every wave streams the whole (small)
    // weight fragment set: 16 * C/16 MFMAs per wave and tile.
    constexpr int TW = (NT + 3) / 4;  // tiles per wave
    auto small_conv = [&](int which, auto emit) {
        const int img = which ? hin : xin;
        const f32x4 *wfrag = a.small_w + which * (G * 2 * 64);  // [G][2][64]
        f32x4 sa[2][TW];
        int base[TW];
#pragma unroll
        for (int t = 0; t < TW; t++) {
            sa[0][t] = sa[1][t] = f32x4{0, 0, 0, 0};
            const int row = (wave + 4 * t) * 16 + fr;
            base[t] = img + (row < rows_img ? row : rows_img - 1) * RS + koff;
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const f32x4 w0 = wfrag[(g * 2 + 0) * 64 + lane], w1 = wfrag[(g * 2 + 1) * 64 + lane];
            f32x4 bt[TW];
#pragma unroll
            for (int t = 0; t < TW; t++) bt[t] = *reinterpret_cast<const f32x4 *>(lds + base[t] + g * 16);
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int t = 0; t < TW; t++) {
                    sa[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[q], bt[t][q], sa[0][t], 0, 0, 0);
                    sa[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[q], bt[t][q], sa[1][t], 0, 0, 0);
                }
        }
#pragma unroll
        for (int t = 0; t < TW; t++) {
            const int row = (wave + 4 * t) * 16 + fr;
            if (wave + 4 * t < NT && row < rows_valid) {
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int q = 0; q < 4; q++) emit(mt, q, row, sa[mt][t][q]);
            }
        }
    };
    conv_heads_tail<C, NT>(a, lds, scratch, board0, boards, rows_valid, small_conv, xin, rows_img * RS);
}
