// kz_conv1x1_split.hip — 1x1 convolution (a GEMM over pixel rows) in the split arithmetic of kz_tower_pairs.hpp (SPLIT) or in plain
// f16: the head convolutions behind the one-launch towers whose heads are separate launches.
#include <cstdlib>
#include <vector>

#include "kz_kernels.hpp"

namespace kz {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {


__device__ __forceinline__ void split4(f32x4 v, h16x4 &hi, h16x4 &lo) {  // hi = f16(v), lo = f16(v - hi)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        hi[j] = (h16)v[j];
        lo[j] = (h16)(v[j] - (float)hi[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 1x1 convolution (a GEMM over pixel rows) in the same split arithmetic, for the head convolutions behind the split
// tower: y[r][oc] = bias[oc] + sum_c W[oc][c] * x[row(r)][c], f32 in and out.  A workgroup stages 64 rows of the f32
// input as (hi, lo) images in LDS and runs passes of 64 * OT output channels over them (wave w: OT 16-channel tiles x the
// four row tiles), the weights read from L2 in fragment order one 32-channel chunk ahead.
struct Conv1x1SplitDev {
    const void *x;      // f32 (SPLIT) or f16
    const uint4 *w;     // [pass][chunk cin/32][hi | lo][wave 4][ot OT][lane 64] x 16 B
    const float *bias;  // [cout_p]
    void *y;            // f32 (SPLIT) or f16
    int ldx, ldy, M, cin, cout_p, relu, group, src_group, src_off;
    // PEPI: the conv policy head's second 1x1 convolution (one output channel: Go's ConvPolicyHead, post_act.py:70-73)
    // as the epilogue of its first — the hidden layer never goes to memory.  policy[(r / hw) * policy_len + r % hw]
    const float *pw1, *pb1;  // [cout], [1]
    float *policy;
    int policy_len, hw;
};

template <int OT, bool SPLIT, bool PEPI = false>
__global__ __launch_bounds__(256) void kz_conv1x1_split(Conv1x1SplitDev a) {
    constexpr int PARTS = SPLIT ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * 64;
    const int RS = a.cin * 2 + 16, LO = 64 * RS;  // hi image, then lo image
    const int chunks = a.cin / 32;

    // stage 64 rows: f32 -> (hi, lo); rows beyond M are zero
    const int pieces = a.cin / 4;
    for (int id = tid; id < 64 * pieces; id += 256) {
        const int r = id / pieces, c4 = id - r * pieces;
        const int orow = row0 + r;
        size_t src = 0;
        if (orow < a.M) {
            const int b = orow / a.group, q = orow - b * a.group;
            src = ((size_t)b * a.src_group + a.src_off + q) * a.ldx + c4 * 4;
        }
        if constexpr (SPLIT) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (orow < a.M) v = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(a.x) + src);
            h16x4 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<h16x4 *>(lds + r * RS + c4 * 8) = hi;
            *reinterpret_cast<h16x4 *>(lds + LO + r * RS + c4 * 8) = lo;
        } else {
            h16x4 v = h16x4{};
            if (orow < a.M) v = *reinterpret_cast<const h16x4 *>(static_cast<const h16 *>(a.x) + src);
            *reinterpret_cast<h16x4 *>(lds + r * RS + c4 * 8) = v;
        }
    }
    __syncthreads();

    const int frag = fr * RS + kq * 16;  // natural k: chunk c covers channels [32 c, 32 c + 32), 8 per lane group
    const int passes = a.cout_p / (64 * OT);
    const size_t step = (size_t)PARTS * 4 * OT * 64;  // uint4 per (pass, chunk)
    for (int pass = 0; pass < passes; pass++) {
        const uint4 *wp = a.w + (size_t)pass * chunks * step + (wave * OT) * 64 + lane;
        const int oc0 = pass * 64 * OT + wave * OT * 16 + kq * 4;
        f32x4 acc[OT][4];
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + oc0 + ot * 16);
#pragma unroll
            for (int mt = 0; mt < 4; mt++) acc[ot][mt] = b;
        }
        uint4 wh[2][OT], wl[2][OT];
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            wh[0][ot] = wp[ot * 64];
            if constexpr (SPLIT) wl[0][ot] = wp[4 * OT * 64 + ot * 64];
        }
#pragma nounroll
        for (int c = 0; c < chunks; c += 2) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int cc = c + half;
                if (cc < chunks) {
                    const int cn = cc + 1 < chunks ? cc + 1 : cc;
#pragma unroll
                    for (int ot = 0; ot < OT; ot++) {
                        wh[half ^ 1][ot] = wp[(size_t)cn * step + ot * 64];
                        if constexpr (SPLIT) wl[half ^ 1][ot] = wp[(size_t)cn * step + 4 * OT * 64 + ot * 64];
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; mt++) {
                        const h16x8 bh = *reinterpret_cast<const h16x8 *>(lds + frag + mt * 16 * RS + cc * 64);
                        h16x8 bl = h16x8{};
                        if constexpr (SPLIT) bl = *reinterpret_cast<const h16x8 *>(lds + LO + frag + mt * 16 * RS + cc * 64);
#pragma unroll
                        for (int ot = 0; ot < OT; ot++) {
                            const h16x8 ah = *reinterpret_cast<const h16x8 *>(&wh[half][ot]);
                            if constexpr (SPLIT) {
                                const h16x8 al = *reinterpret_cast<const h16x8 *>(&wl[half][ot]);
                                acc[ot][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[ot][mt], 0, 0, 0);
                                acc[ot][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[ot][mt], 0, 0, 0);
                            }
                            acc[ot][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[ot][mt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if constexpr (PEPI) {
            // policy logit of a row = b1 + sum over all output channels of w1[oc] * relu(hidden[oc]), the hidden value
            // rounded to the tensor type first (f16 unless SPLIT) as the separate launches did: lanes add their 4 x OT
            // channels, the four lane groups of a row meet by butterfly, the four waves through LDS (one pass: cout_p ==
            // 64 * OT, checked by the launcher)
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ot = 0; ot < OT; ot++) {
                const f32x4 w1 = *reinterpret_cast<const f32x4 *>(a.pw1 + oc0 + ot * 16);
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float h = acc[ot][mt][j];
                        h = h > 0.0f ? h : 0.0f;
                        if constexpr (!SPLIT) h = (float)(h16)h;
                        part[mt] += w1[j] * h;
                    }
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                part[mt] += __shfl_xor(part[mt], 16, 64);
                part[mt] += __shfl_xor(part[mt], 32, 64);
            }
            __syncthreads();  // every wave is done reading the staged rows: their LDS is free
            float *red = reinterpret_cast<float *>(lds);  // [wave 4][row 64]
            if (kq == 0) {
#pragma unroll
                for (int mt = 0; mt < 4; mt++) red[wave * 64 + mt * 16 + fr] = part[mt];
            }
            __syncthreads();
            if (tid < 64) {
                const int r = row0 + tid;
                if (r < a.M) {
                    const int b = r / a.hw, q = r - b * a.hw;
                    a.policy[(size_t)b * a.policy_len + q] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid] + a.pb1[0];
                }
            }
            return;
        }
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const int r = row0 + mt * 16 + fr;
                f32x4 v = acc[ot][mt];
                if (a.relu) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                }
                if (r < a.M) {
                    const size_t o = (size_t)r * a.ldy + oc0 + ot * 16;
                    if constexpr (SPLIT) *reinterpret_cast<f32x4 *>(static_cast<float *>(a.y) + o) = v;
                    else *reinterpret_cast<h16x4 *>(static_cast<h16 *>(a.y) + o) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                }
            }
    }
}

}  // namespace

// ---- 1x1 convolution in split arithmetic (head convolutions behind the split tower) ----
static int conv1x1_split_ot(int cout_p) { return cout_p % 256 == 0 ? 4 : cout_p % 128 == 0 ? 2 : cout_p % 64 == 0 ? 1 : 0; }

// the one-filter second convolution of a conv policy head as the first one's epilogue: a single pass over the output
// channels (every wave holds all of them for its rows), rows gathered as they are (no source remapping)
bool conv1x1_policy_epilogue_supported(int cin_p, int cout_p, int cout, int policy_channels) {
    const int ot = conv1x1_split_ot(cout_p);
    return policy_channels == 1 && cin_p % 32 == 0 && cin_p >= 32 && cin_p <= 512 && ot != 0 && cout_p == 64 * ot && cout == cout_p;
}

bool conv1x1_split_supported(int cin_p, int cout_p) {
    return cin_p % 32 == 0 && cin_p >= 32 && cin_p <= 512 && conv1x1_split_ot(cout_p) != 0;
}

size_t conv1x1_split_weight_elems(int cin_p, int cout_p, bool split) { return (size_t)(split ? 2 : 1) * cin_p * cout_p; }  // f16 elements

// [cout_p][cin_p] f32 (zero padded) -> [pass][chunk][hi | lo][wave 4][ot][lane 64][8] f16: element j of lane (fr, kq) is
// W[oc = 64*OT*pass + 16*(wave*OT + ot) + fr][channel = 32*chunk + 8*kq + j]
void conv1x1_split_pack_weights(const float *w, int cout, int cin, int cout_p, int cin_p, bool split, uint16_t *dst) {
    const int ot_n = conv1x1_split_ot(cout_p), passes = cout_p / (64 * ot_n), chunks = cin_p / 32;
    const size_t part = (size_t)4 * ot_n * 64 * 8;
    for (int pass = 0; pass < passes; pass++)
        for (int chunk = 0; chunk < chunks; chunk++) {
            uint16_t *step = dst + ((size_t)pass * chunks + chunk) * (split ? 2 : 1) * part;
            for (int wave = 0; wave < 4; wave++)
                for (int ot = 0; ot < ot_n; ot++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int oc = 64 * ot_n * pass + 16 * (wave * ot_n + ot) + (lane & 15);
                            const int ch = 32 * chunk + 8 * (lane >> 4) + j;
                            float v = 0.0f;
                            if (oc < cout && ch < cin) v = w[(size_t)oc * cin + ch];
                            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                            uint16_t hb, lb;
                            __builtin_memcpy(&hb, &hi, 2);
                            __builtin_memcpy(&lb, &lo, 2);
                            const size_t e = (((size_t)wave * ot_n + ot) * 64 + lane) * 8 + j;
                            step[e] = hb;
                            if (split) step[part + e] = lb;
                        }
        }
}

void launch_conv1x1_split(const Conv1x1SplitArgs &t, hipStream_t stream) {
    Conv1x1SplitDev d{};
    d.x = t.x;
    d.w = static_cast<const uint4 *>(t.weights);
    d.bias = t.bias;
    d.y = t.y;
    d.ldx = t.ldx;
    d.ldy = t.ldy;
    d.M = t.M;
    d.cin = t.cin_p;
    d.cout_p = t.cout_p;
    d.relu = t.relu;
    d.group = t.group;
    d.src_group = t.src_group;
    d.src_off = t.src_off;
    const int lds_bytes = (t.split ? 2 : 1) * 64 * (t.cin_p * 2 + 16);
    const int grid = (t.M + 63) / 64;
    const int ot = conv1x1_split_ot(t.cout_p);
    auto go = [&](auto kernel) {
        static thread_local unsigned long long done_mask = 0;  // per instantiation
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!((done_mask >> (dev & 63)) & 1)) {
            (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done_mask |= 1ull << (dev & 63);
        }
        kernel<<<grid, 256, lds_bytes, stream>>>(d);
    };
    if (t.policy) {  // (conv1x1_policy_epilogue_supported)
        d.pw1 = t.pw1; d.pb1 = t.pb1; d.policy = t.policy; d.policy_len = t.policy_len; d.hw = t.hw;
        if (t.split) {
            if (ot == 4) go(kz_conv1x1_split<4, true, true>);
            else if (ot == 2) go(kz_conv1x1_split<2, true, true>);
            else go(kz_conv1x1_split<1, true, true>);
        } else {
            if (ot == 4) go(kz_conv1x1_split<4, false, true>);
            else if (ot == 2) go(kz_conv1x1_split<2, false, true>);
            else go(kz_conv1x1_split<1, false, true>);
        }
        return;
    }
    if (t.split) {
        if (ot == 4) go(kz_conv1x1_split<4, true>);
        else if (ot == 2) go(kz_conv1x1_split<2, true>);
        else go(kz_conv1x1_split<1, true>);
    } else {
        if (ot == 4) go(kz_conv1x1_split<4, false>);
        else if (ot == 2) go(kz_conv1x1_split<2, false>);
        else go(kz_conv1x1_split<1, false>);
    }
}

}  // namespace kz
