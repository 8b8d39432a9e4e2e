// kz_att_heads.hip — ScalarHead (python/lib/model/post_act.py:10-23) and AttentionPolicyHead (post_act.py:115-141) of one 8x8
// board in ONE workgroup, ONE launch per batch, f16: for every f16 engine whose tower launch does not carry these heads
// itself (the chess towers off the flagship width, the AttentionTower networks) in place of four launches (kz_scalar_head,
// two 1x1 convolutions through kz_conv1x1_split, kz_attention_mfma).
//
//   bulk  = conv_bulk(common)                 [2 Q][64]   q_from = bulk[:Q], k_to = bulk[Q:]            (:127, :130-132)
//   under = conv_under(common[:, :, 7, None]) [3 Q][8]    reshape(Q, 24): k_under[q][8 r + w] = under[3 q + r][w]   (:128, :133)
//   policy = (q_from^T [k_to | k_under]) / sqrt(Q)  [64][88], flattened and gathered by FLAT_TO_ATT     (:137-140)
//
// The tower output's 64 rows sit in LDS; the three 1x1 convolutions run features x tokens (weights = the MFMA's A operand,
// a 16-feature tile per wave at a time with the next tile's fragments in flight), so a lane writes four consecutive features
// of one token into the q_from / k_to rows; the logits are a second GEMM over those rows (24 tile pairs, three per wave).
// conv_under's rows are stored r-major (row r Q + q = the reference's 3 q + r) so that a tile is 16 q of one r.
#include "kz_kernels.hpp"

namespace kz {
namespace {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AH_THREADS = 512, AH_WAVES = 8, AH_PAD = 16, AH_KSMAX = 16, AH_TO = 96, AH_LDL = 100;

struct AttHeadsDev {
    const h16 *x;
    int ldx, batch, cp, q, hc, hs, policy_len;
    const uint4 *w_bulk, *w_under, *w_sc;
    const float *b_bulk, *b_under, *b_sc;
    const float *w1, *b1, *w2, *b2;
    const int32_t *flat_to_att;
    float *scalars, *policy;
    int *nonfinite_flag;
    int epoch;
    float inv_sqrt_q;
};

struct AhGeo {
    int ldx, ldq;                                    // row strides (f16 values) of the tower rows and of the q_from / k_to rows
    size_t off_qf, off_kt, off_log, off_sc, off_hid, bytes;  // byte offsets
};
__host__ __device__ inline AhGeo ah_geo(int cp, int q, int hc, int hs) {
    AhGeo g;
    g.ldx = cp + AH_PAD;
    g.ldq = q + AH_PAD;
    g.off_qf = (size_t)64 * g.ldx * 2;
    g.off_kt = g.off_qf + (size_t)64 * g.ldq * 2;
    g.off_log = g.off_kt + (size_t)AH_TO * g.ldq * 2;
    g.off_sc = g.off_log + (size_t)64 * AH_LDL * 4;
    g.off_hid = g.off_sc + (size_t)hc * 64 * 4;
    g.bytes = g.off_hid + (size_t)hs * 4;
    return g;
}

__device__ __forceinline__ float ah_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(AH_THREADS) void kz_att_heads_f16(AttHeadsDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const AhGeo g = ah_geo(a.cp, a.q, a.hc, a.hs);
    h16 *XS = reinterpret_cast<h16 *>(lds), *QF = reinterpret_cast<h16 *>(lds + g.off_qf), *KT = reinterpret_cast<h16 *>(lds + g.off_kt);
    float *LOG = reinterpret_cast<float *>(lds + g.off_log), *SC = reinterpret_cast<float *>(lds + g.off_sc),
          *HID = reinterpret_cast<float *>(lds + g.off_hid);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, kq = lane >> 4;
    const int KS = a.cp / 32, Q = a.q, n_bulk = 2 * Q / 16, n_under = 3 * Q / 16, n_tiles = n_bulk + n_under + 1;
    const int board = blockIdx.x;

    // ---- the board's 64 tower rows -> LDS; the 8 rows of k_to beyond the 88 "to" squares are zero ----
    {
        const int per_row = a.cp / 8;
        for (int i = tid; i < 64 * per_row; i += AH_THREADS) {
            const int r = i / per_row, c = i - r * per_row;
            *reinterpret_cast<uint4 *>(XS + r * g.ldx + c * 8) = *reinterpret_cast<const uint4 *>(a.x + ((size_t)board * 64 + r) * a.ldx + c * 8);
        }
        for (int i = tid; i < 8 * (Q / 8); i += AH_THREADS) {
            const int r = 88 + i / (Q / 8), c = i % (Q / 8);
            *reinterpret_cast<uint4 *>(KT + r * g.ldq + c * 8) = uint4{0, 0, 0, 0};
        }
    }
    __syncthreads();

    // ---- the three 1x1 convolutions: 16-feature tiles, round-robin over the waves ----
    auto tile_ptr = [&](int t) -> const uint4 * {
        if (t < n_bulk) return a.w_bulk + (size_t)t * KS * 64 + lane;
        if (t < n_bulk + n_under) return a.w_under + (size_t)(t - n_bulk) * KS * 64 + lane;
        return a.w_sc + lane;
    };
    uint4 f0[AH_KSMAX], f1[AH_KSMAX];  // the tile in work and the wave's next one, in flight
    auto load_tile = [&](uint4 (&f)[AH_KSMAX], int t) {
        const uint4 *p = tile_ptr(t);
#pragma unroll
        for (int ks = 0; ks < AH_KSMAX; ks++)
            if (ks < KS) f[ks] = p[(size_t)ks * 64];
    };
    bool bad = false;
    auto compute = [&](const uint4 (&f)[AH_KSMAX], int t) {
        const bool is_under = t >= n_bulk && t < n_bulk + n_under;
        f32x4 acc[4];
#pragma unroll
        for (int tt = 0; tt < 4; tt++) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const h16 *brow = XS + fr * g.ldx + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < AH_KSMAX; ks++) {
            if (ks < KS) {
                const h16x8 av = *reinterpret_cast<const h16x8 *>(&f[ks]);
                if (is_under) {  // rank 7 only: tokens 56 .. 63 = the upper half of token tile 3
                    const h16x8 b = *reinterpret_cast<const h16x8 *>(brow + 48 * g.ldx + ks * 32);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b, acc[3], 0, 0, 0);
                } else {
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) {
                        const h16x8 b = *reinterpret_cast<const h16x8 *>(brow + tt * 16 * g.ldx + ks * 32);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b, acc[tt], 0, 0, 0);
                    }
                }
            }
        }
        // lane: features 4 kq + j of the tile, token 16 tt + fr
        if (t < n_bulk) {
            const int feat = 16 * t + 4 * kq;
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.b_bulk + feat);
            h16 *dst = feat < Q ? QF + feat : KT + (feat - Q);
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                h16x4 o;
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = (h16)(acc[tt][j] + bias[j]);
                *reinterpret_cast<h16x4 *>(dst + (16 * tt + fr) * g.ldq) = o;
            }
        } else if (is_under) {
            const int u0 = 16 * (t - n_bulk) + 4 * kq, r = u0 / Q, q0 = u0 - r * Q;  // row r Q + q of the r-major conv_under
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.b_under + u0);
            if (fr >= 8) {
                h16x4 o;
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = (h16)(acc[3][j] + bias[j]);
                *reinterpret_cast<h16x4 *>(KT + (64 + 8 * r + (fr - 8)) * g.ldq + q0) = o;
            }
        } else if (kq * 4 < a.hc) {  // the scalar head's conv: features 0 .. hc of one zero-padded tile; ReLU; channel-major flatten
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.b_sc + 4 * kq);
#pragma unroll
            for (int tt = 0; tt < 4; tt++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float v = acc[tt][j] + bias[j];
                    if (4 * kq + j < a.hc) {
                        bad |= !(fabsf(v) <= 3.0e38f);  // the range check: this convolution reads every value of the tower output
                        SC[(4 * kq + j) * 64 + 16 * tt + fr] = fmaxf(v, 0.0f);
                    }
                }
        }
    };
    {
        int t = wave;
        if (t < n_tiles) load_tile(f0, t);
        while (t < n_tiles) {
            if (t + AH_WAVES < n_tiles) load_tile(f1, t + AH_WAVES);
            compute(f0, t);
            t += AH_WAVES;
            if (t >= n_tiles) break;
            if (t + AH_WAVES < n_tiles) load_tile(f0, t + AH_WAVES);
            compute(f1, t);
            t += AH_WAVES;
        }
    }
    if (bad && a.nonfinite_flag) *reinterpret_cast<volatile int *>(a.nonfinite_flag) = a.epoch;  // (plain store: the flag may live in pinned host memory)
    __syncthreads();

    // ---- logits^T [to][from] = k_to q_from^T / sqrt(Q): 6 x 4 tile pairs ----
    for (int p = wave; p < 24; p += AH_WAVES) {
        const int tt = p >> 2, ft = p & 3;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const h16 *arow = KT + (16 * tt + fr) * g.ldq + 8 * kq, *brow = QF + (16 * ft + fr) * g.ldq + 8 * kq;
        for (int ks = 0; ks < Q / 32; ks++)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h16x8 *>(arow + ks * 32),
                                                         *reinterpret_cast<const h16x8 *>(brow + ks * 32), acc, 0, 0, 0);
        // lane: to = 16 tt + 4 kq + j, from = 16 ft + fr
        *reinterpret_cast<f32x4 *>(LOG + (16 * ft + fr) * AH_LDL + 16 * tt + 4 * kq) = acc * a.inv_sqrt_q;
    }
    // ---- ScalarHead's first Linear + ReLU (post_act.py:19-20): an output per wave at a time, the inputs across the lanes ----
    {
        const int in = a.hc * 64;
        for (int o = wave; o < a.hs; o += AH_WAVES) {
            float s = 0.0f;
            for (int i = lane; i < in; i += 64) s = fmaf(SC[i], a.w1[(size_t)o * in + i], s);
            s = ah_wave_sum(s);
            if (lane == 0) HID[o] = fmaxf(s + a.b1[o], 0.0f);
        }
    }
    __syncthreads();
    if (wave == 0 && lane < 5) {  // the last Linear (:21)
        float s = a.b2[lane];
        for (int h = 0; h < a.hs; h++) s = fmaf(HID[h], a.w2[lane * a.hs + h], s);
        a.scalars[(size_t)board * 5 + lane] = s;
    }
    // ---- policy.flatten(1)[:, FLAT_TO_ATT] (:140) ----
    for (int i = tid; i < a.policy_len; i += AH_THREADS) {
        const int idx = a.flat_to_att[i];
        a.policy[(size_t)board * a.policy_len + i] = LOG[(idx / 88) * AH_LDL + idx % 88];
    }
}

// W [N][K_src] f32 -> A-operand fragments [ceil(N / 16)][K / 32][64 lanes][8] f16 (rows >= N and columns >= K_src zero);
// row_of(r) = the source row stored at fragment row r
template <class RowOf>
void ah_pack(const float *w, int N, int K_src, int K, uint16_t *dst, RowOf row_of) {
    const int nt = (N + 15) / 16;
    for (int t = 0; t < nt; t++)
        for (int ks = 0; ks < K / 32; ks++)
            for (int lane = 0; lane < 64; lane++)
                for (int i = 0; i < 8; i++) {
                    const int r = 16 * t + (lane & 15), col = 32 * ks + 8 * (lane >> 4) + i;
                    const _Float16 h = (_Float16)(r < N && col < K_src ? w[(size_t)row_of(r) * K_src + col] : 0.0f);
                    __builtin_memcpy(dst + (((size_t)t * (K / 32) + ks) * 64 + lane) * 8 + i, &h, 2);
                }
}

}  // namespace

bool att_heads_supported(int dtype, int h, int w, int channels, int q, int hc, int hs, int policy_len) {
    const int cp = (channels + 31) / 32 * 32;
    if (dtype != 1 || h != 8 || w != 8 || q < 32 || q % 32 || hc < 1 || hc > 16 || hs < 1 || hs > 4096 || policy_len < 1) return false;
    if (cp / 32 > AH_KSMAX) return false;
    return ah_geo(cp, q, hc, hs).bytes <= (size_t)160 * 1024;
}

size_t att_heads_weight_elems(int channels, int q) {
    const int cp = (channels + 31) / 32 * 32;
    return (size_t)(2 * q + 3 * q + 16) * cp;
}

// dst: bulk fragments | under fragments (r-major rows) | scalar conv fragments (one zero-padded tile); bias: [2 Q] | [3 Q] r-major | [16]
void att_heads_pack(const float *w_bulk, const float *b_bulk, const float *w_under, const float *b_under, const float *w_sc, const float *b_sc,
                    int channels, int q, int hc, uint16_t *dst, float *bias) {
    const int cp = (channels + 31) / 32 * 32;
    ah_pack(w_bulk, 2 * q, channels, cp, dst, [](int r) { return r; });
    ah_pack(w_under, 3 * q, channels, cp, dst + (size_t)2 * q * cp, [q](int r) { return 3 * (r % q) + r / q; });
    ah_pack(w_sc, hc, channels, cp, dst + (size_t)5 * q * cp, [](int r) { return r; });
    for (int i = 0; i < 2 * q; i++) bias[i] = b_bulk[i];
    for (int r = 0; r < 3 * q; r++) bias[2 * q + r] = b_under[3 * (r % q) + r / q];
    for (int i = 0; i < 16; i++) bias[5 * q + i] = i < hc ? b_sc[i] : 0.0f;
}

void launch_att_heads(const AttHeadsArgs &t, hipStream_t stream) {
    if (t.batch <= 0) return;
    const int cp = (t.channels + 31) / 32 * 32;
    AttHeadsDev d{};
    d.x = static_cast<const h16 *>(t.x); d.ldx = t.ldx; d.batch = t.batch; d.cp = cp; d.q = t.q; d.hc = t.hc; d.hs = t.hs;
    d.policy_len = t.policy_len;
    const uint4 *w = static_cast<const uint4 *>(t.weights);
    d.w_bulk = w;
    d.w_under = w + (size_t)2 * t.q * cp / 8;
    d.w_sc = w + (size_t)5 * t.q * cp / 8;
    d.b_bulk = t.bias; d.b_under = t.bias + 2 * t.q; d.b_sc = t.bias + 5 * t.q;
    d.w1 = t.w1; d.b1 = t.b1; d.w2 = t.w2; d.b2 = t.b2;
    d.flat_to_att = t.flat_to_att; d.scalars = t.scalars; d.policy = t.policy;
    d.nonfinite_flag = t.nonfinite_flag; d.epoch = t.epoch;
    d.inv_sqrt_q = 1.0f / sqrtf((float)t.q);
    const size_t bytes = ah_geo(cp, t.q, t.hc, t.hs).bytes;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_att_heads_f16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done_mask |= 1ull << (dev & 63);
    }
    kz_att_heads_f16<<<t.batch, AH_THREADS, bytes, stream>>>(d);
}

}  // namespace kz
