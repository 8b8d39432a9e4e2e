// kz_att_tower_f16.hip — AttentionTower (python/lib/model/attention.py:8-136) on the f16 matrix cores: the whole tower of one
// 8x8 board in ONE workgroup of eight waves, ONE launch per batch.  Shapes: 64 squares, 8 heads of d_k = d_v = 16 (what
// python/main/supervised_main_alpha.py:72 builds), d_model / d_ff of the instances at the end of this file; every other
// AttentionTower runs through kz_att_tower.hip (exact f32).
//
// Orientation.  Every Linear layer is computed TRANSPOSED, features x tokens: the weights are the A operand of
// v_mfma_f32_16x16x32_f16 (streamed from global memory in fragment order, one 1 KB load per wave and fragment), the tokens'
// f16 rows in LDS the B operand, so a lane's four accumulator values are four CONSECUTIVE FEATURES OF ONE TOKEN:
//   * the residual stream X (f32) never leaves the registers — each wave owns d_model / 8 features of all 64 tokens, and the
//     DeepNorm residual x * alpha + f(x) (attention.py:126,129) is the accumulator's initial value;
//   * LayerNorm's sums over the features are in-lane adds, two butterflies and one exchange between the waves through LDS;
//   * the f16 copy of X the next layer multiplies is written as 8-byte stores.
// Attention without LDS.  Wave h computes q, k and v of head h (rows 48 h .. 48 h + 47 of project_qkv: the
// view(n, b * heads, d_kqv) of attention.py:106) and the head's whole attention in registers: q and k tiles in the
// accumulator layout ARE operands of v_mfma_f32_16x16x16_f16 (token = lane & 15, four features per lane group), so
// logits^T = k q^T needs no data movement; its accumulator layout (query = lane & 15, four keys per lane group) is the A
// operand of weights x v; and the v tiles are computed with the MFMA's operands exchanged (tokens x features), which is the
// B operand of that product.  Softmax over the keys (attention.py:119, no scale factor :117): in-lane over 16 values, two
// butterflies.
#include "kz_kernels.hpp"

namespace kz {
namespace {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS rows of f16 are padded by 16 values: a row stride of 32 bytes modulo 64 makes the 16-byte fragment reads of a wave
// (lane = 16 kq + fr reads row fr at byte 16 kq: ds_read_b128 serves lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... together)
// hit 16 distinct 16-byte slots; with 8 (stride 16 modulo 32) rows fr = 12 and fr = 11 of neighbouring lane groups collide
// (measured: SQ_LDS_BANK_CONFLICT 44 % of the LDS-active cycles)
constexpr int A16_PAD = 16;
constexpr int A16_THREADS = 512, A16_WAVES = 8, A16_TOKENS = 64, A16_HEADS = 8, A16_LDA = A16_HEADS * 16 + A16_PAD;
constexpr int A16_MAX_CIN = 224;

struct AttTower16Dev {
    const h16 *x0;          // encoded planes [batch * 64][cin_p] f16 (channels >= c_in zero)
    int cin_p;
    const uint4 *w_expand;  // fragments of expand.weight [D][cin_p]
    const float *embedding; // [64][D] f32
    const uint4 *w_layers;  // per layer: project_qkv | project_out | ff.0 | ff.2 fragments
    h16 *y;                 // [batch * 64][D] f16
    // fused board encode (F0, rust/kz-core/src/mapping/mod.rs:40-63): packed boards straight into the launch (bits == nullptr: x0)
    const uint8_t *bits;
    size_t bits_stride;
    const float *scalars_in;
    int n_scalar, n_bool;
    int batch, depth;
    float alpha, eps;
};

template <int D, int DFF>
struct A16Shape {
    static constexpr int LDX = D + A16_PAD, LDH = DFF + A16_PAD;
    // the second region holds, one after the other: the encoded planes (<= 224 columns + padding), all heads' attention output
    // (128), the feed-forward hidden layer (DFF)
    static constexpr int LDR = LDH > A16_MAX_CIN + A16_PAD ? LDH : A16_MAX_CIN + A16_PAD;
    static constexpr int X_ELEMS = A16_TOKENS * LDX, R_ELEMS = A16_TOKENS * LDR;
    static constexpr int RED_FLOATS = 2 * A16_TOKENS * A16_WAVES;
    static constexpr size_t LDS_BYTES = (size_t)(X_ELEMS + R_ELEMS) * 2 + RED_FLOATS * 4;
    static constexpr int KSD = D / 32, KSF = DFF / 32, KSA = A16_HEADS * 16 / 32;
    static constexpr int NTD = D / 128, NTF = DFF / 128;  // 16-feature tiles per wave
    static constexpr size_t LAYER_FRAGS = (size_t)64 * (24 * KSD + (D / 16) * KSA + (DFF / 16) * KSD + (D / 16) * KSF);
};

__device__ __forceinline__ h16x8 as_h8(const uint4 &v) { return *reinterpret_cast<const h16x8 *>(&v); }

// The first k-steps' weight fragments of a GEMM, loaded ahead of it (before the barrier and the LayerNorm / attention / store
// that precede it, whose time then hides the loads' latency).
template <int NTW, int KS>
struct Ring {
    static constexpr int PF = KS < 4 ? KS : 4;
    uint4 r[PF][NTW];
};
template <int NTW, int KS>
__device__ __forceinline__ void ring_preload(Ring<NTW, KS> &g, const uint4 *__restrict__ wf) {
#pragma unroll
    for (int p = 0; p < Ring<NTW, KS>::PF; p++)
#pragma unroll
        for (int t = 0; t < NTW; t++) g.r[p][t] = wf[(size_t)(t * KS + p) * 64];
}

// acc[t][tt] (+)= W tile t (16 features) x tokens tile tt over KS k-steps of 32.  wf: this lane's slot of the wave's first
// tile (tile stride KS * 64 fragments, k-step stride 64), its first k-steps already in `g`; act: LDS rows of LDB f16.  VT: the
// last tile is computed with the operands exchanged (tokens x features).
template <int NTW, int KS, int LDB, bool VT>
__device__ __forceinline__ void gemm16(Ring<NTW, KS> &g, const uint4 *__restrict__ wf, const h16 *act, int fr, int kq,
                                       f32x4 (&acc)[NTW][4]) {
    constexpr int PF = Ring<NTW, KS>::PF;
    const h16 *brow = act + fr * LDB + 8 * kq;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        h16x8 b[4];
#pragma unroll
        for (int tt = 0; tt < 4; tt++) b[tt] = *reinterpret_cast<const h16x8 *>(brow + tt * 16 * LDB + ks * 32);
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const h16x8 a = as_h8(g.r[ks % PF][t]);
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                if (VT && t == NTW - 1) acc[t][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[tt], a, acc[t][tt], 0, 0, 0);
                else acc[t][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[tt], acc[t][tt], 0, 0, 0);
            }
        }
        if (ks + PF < KS) {
#pragma unroll
            for (int t = 0; t < NTW; t++) g.r[ks % PF][t] = wf[(size_t)(t * KS + ks + PF) * 64];
        }
    }
}

// the same with a run-time number of k-steps (the expand layer: one to seven)
template <int NTW>
__device__ __forceinline__ void gemm16_rt(const uint4 *__restrict__ wf, int ks_n, const h16 *act, int ldb, int fr, int kq,
                                          f32x4 (&acc)[NTW][4]) {
    const h16 *brow = act + fr * ldb + 8 * kq;
    for (int ks = 0; ks < ks_n; ks++) {
        h16x8 b[4];
#pragma unroll
        for (int tt = 0; tt < 4; tt++) b[tt] = *reinterpret_cast<const h16x8 *>(brow + tt * 16 * ldb + ks * 32);
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const h16x8 a = as_h8(wf[(size_t)(t * ks_n + ks) * 64]);
#pragma unroll
            for (int tt = 0; tt < 4; tt++) acc[t][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[tt], acc[t][tt], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ float group_sum(float v) {  // over the four lane groups (lanes fr, fr + 16, fr + 32, fr + 48)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// LayerNorm(D) without parameters over the features of every token (attention.py:80-81), X in registers: this wave's NTD
// feature tiles of all 64 tokens.  One pass (sum and sum of squares in f32), one exchange between the waves: ln_partial writes
// this wave's sums, the caller synchronises, ln_finish normalises X and writes its f16 copy into X16 (NOT synchronised).
template <int NTD>
__device__ __forceinline__ void ln_partial(const f32x4 (&X)[NTD][4], float *red, int wave, int fr, int kq) {
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
        float s = 0.0f, q = 0.0f;
#pragma unroll
        for (int t = 0; t < NTD; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                s += X[t][tt][j];
                q = fmaf(X[t][tt][j], X[t][tt][j], q);
            }
        s = group_sum(s);
        q = group_sum(q);
        if (kq == 0) {
            red[(16 * tt + fr) * A16_WAVES + wave] = s;
            red[A16_TOKENS * A16_WAVES + (16 * tt + fr) * A16_WAVES + wave] = q;
        }
    }
}
template <int D, int NTD, int LDX>
__device__ __forceinline__ void ln_finish(f32x4 (&X)[NTD][4], const float *red, h16 *X16, int wave, int fr, int kq, float eps) {
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
        const f32x4 *ps = reinterpret_cast<const f32x4 *>(red + (16 * tt + fr) * A16_WAVES);
        const f32x4 *pq = reinterpret_cast<const f32x4 *>(red + A16_TOKENS * A16_WAVES + (16 * tt + fr) * A16_WAVES);
        const f32x4 s0 = ps[0], s1 = ps[1], q0 = pq[0], q1 = pq[1];
        const float sum = ((s0[0] + s0[1]) + (s0[2] + s0[3])) + ((s1[0] + s1[1]) + (s1[2] + s1[3]));
        const float sq = ((q0[0] + q0[1]) + (q0[2] + q0[3])) + ((q1[0] + q1[1]) + (q1[2] + q1[3]));
        const float mean = sum * (1.0f / D);
        const float var = fmaxf(sq * (1.0f / D) - mean * mean, 0.0f);
        const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int t = 0; t < NTD; t++) {
            h16x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                X[t][tt][j] = (X[t][tt][j] - mean) * inv;
                o[j] = (h16)X[t][tt][j];
            }
            *reinterpret_cast<h16x4 *>(X16 + (16 * tt + fr) * LDX + (wave * NTD + t) * 16 + 4 * kq) = o;
        }
    }
}

template <int D, int DFF>
__global__ __launch_bounds__(A16_THREADS) void kz_att_tower_f16(AttTower16Dev a) {
    using S = A16Shape<D, DFF>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    h16 *X16 = reinterpret_cast<h16 *>(lds_raw);
    h16 *R = X16 + S::X_ELEMS;  // IN16 / ATT16 / H16
    float *red = reinterpret_cast<float *>(R + S::R_ELEMS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, kq = lane >> 4;
    const int ks_in = a.cin_p / 32, ldi = a.cin_p + A16_PAD;

    for (int board = blockIdx.x; board < a.batch; board += gridDim.x) {
        // ---- the board's encoded planes -> LDS ----
        __syncthreads();
        if (a.bits) {
            // scalar planes first, each broadcast over the board, then the bool planes: bool i of a board = bit i % 8 of byte
            // i / 8 (bit_buffer.rs:73-75), i = plane * 64 + square
            const int per_row = a.cin_p / 8;
            const uint8_t *bb = a.bits + (size_t)board * a.bits_stride;
            const float *sc = a.scalars_in + (size_t)board * a.n_scalar;
            for (int i = tid; i < A16_TOKENS * per_row; i += A16_THREADS) {
                const int c = i / A16_TOKENS, r = i - c * A16_TOKENS;
                h16x8 v;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int ch = c * 8 + j;
                    float f = 0.0f;
                    if (ch < a.n_scalar) {
                        f = sc[ch];
                    } else if (ch < a.n_scalar + a.n_bool) {
                        const unsigned bit = (unsigned)(ch - a.n_scalar) * A16_TOKENS + r;
                        f = (float)((bb[bit >> 3] >> (bit & 7)) & 1);
                    }
                    v[j] = (h16)f;
                }
                *reinterpret_cast<h16x8 *>(R + r * ldi + c * 8) = v;
            }
        } else {
            const int per_row = a.cin_p / 8;
            const uint4 *src = reinterpret_cast<const uint4 *>(a.x0 + (size_t)board * A16_TOKENS * a.cin_p);
            for (int i = tid; i < A16_TOKENS * per_row; i += A16_THREADS) {
                const int r = i / per_row, c = i - r * per_row;
                *reinterpret_cast<uint4 *>(R + r * ldi + c * 8) = src[i];
            }
        }
        __syncthreads();
        // ---- expand + embedding (attention.py:39-40): X[feature][token] ----
        f32x4 X[S::NTD][4];
#pragma unroll
        for (int t = 0; t < S::NTD; t++)
#pragma unroll
            for (int tt = 0; tt < 4; tt++)
                X[t][tt] = *reinterpret_cast<const f32x4 *>(a.embedding + (size_t)(16 * tt + fr) * D + (wave * S::NTD + t) * 16 + 4 * kq);
        gemm16_rt<S::NTD>(a.w_expand + (size_t)wave * S::NTD * ks_in * 64 + lane, ks_in, R, ldi, fr, kq, X);
#pragma unroll
        for (int t = 0; t < S::NTD; t++)
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                h16x4 o;
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = (h16)X[t][tt][j];
                *reinterpret_cast<h16x4 *>(X16 + (16 * tt + fr) * S::LDX + (wave * S::NTD + t) * 16 + 4 * kq) = o;
            }

        // a GEMM's first weight fragments are requested one phase ahead: q | k | v's behind the previous layer's last GEMM
        // (here: behind the expand layer), project_out's ahead of the attention, ff.0's and ff.2's ahead of the LayerNorms
        Ring<3, S::KSD> g_qkv;
        const uint4 *wl = a.w_layers;
        ring_preload(g_qkv, wl + (size_t)wave * 3 * S::KSD * 64 + lane);
        __syncthreads();
        for (int l = 0; l < a.depth; l++) {
            const uint4 *wqkv = wl + (size_t)wave * 3 * S::KSD * 64 + lane;
            const uint4 *wout = wl + (size_t)64 * 24 * S::KSD + (size_t)wave * S::NTD * S::KSA * 64 + lane;
            const uint4 *wf0 = wl + (size_t)64 * (24 * S::KSD + (D / 16) * S::KSA) + (size_t)wave * S::NTF * S::KSD * 64 + lane;
            const uint4 *wf1 = wl + (size_t)64 * (24 * S::KSD + (D / 16) * S::KSA + (DFF / 16) * S::KSD) + (size_t)wave * S::NTD * S::KSF * 64 + lane;
            wl += S::LAYER_FRAGS;
            Ring<S::NTD, S::KSA> g_out;
            // ---- q, k, v of head `wave` and its attention, in registers ----
            {
                f32x4 qkv[3][4];
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) qkv[t][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm16<3, S::KSD, S::LDX, true>(g_qkv, wqkv, X16, fr, kq, qkv);
                ring_preload(g_out, wout);
                h16x4 qf[4], kf[4], vf[4];
#pragma unroll
                for (int tt = 0; tt < 4; tt++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        qf[tt][j] = (h16)qkv[0][tt][j];  // q[token 16 tt + fr][feature 4 kq + j]
                        kf[tt][j] = (h16)qkv[1][tt][j];
                        vf[tt][j] = (h16)qkv[2][tt][j];  // v[token 16 tt + 4 kq + j][feature fr]
                    }
#pragma unroll
                for (int qt = 0; qt < 4; qt++) {
                    // logits^T [key][query] = k q^T: lane holds query 16 qt + fr, keys 16 kt + 4 kq + j
                    f32x4 sc[4];
                    float mx = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 4; kt++) {
                        sc[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[kt], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 4; j++) mx = fmaxf(mx, sc[kt][j]);
                    }
                    mx = group_max(mx);
                    float sum = 0.0f;
#pragma unroll
                    for (int kt = 0; kt < 4; kt++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            sc[kt][j] = __expf(sc[kt][j] - mx);
                            sum += sc[kt][j];
                        }
                    const float inv = 1.0f / group_sum(sum);
                    // att[query][feature] = weights v: A = weights (query fr, keys 4 kq ..), B = v (keys 4 kq .., feature fr)
                    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < 4; kt++) {
                        h16x4 p;
#pragma unroll
                        for (int j = 0; j < 4; j++) p[j] = (h16)(sc[kt][j] * inv);
                        o = __builtin_amdgcn_mfma_f32_16x16x16f16(p, vf[kt], o, 0, 0, 0);
                    }
                    // o: query 16 qt + 4 kq + j, feature fr of this head
#pragma unroll
                    for (int j = 0; j < 4; j++) R[(16 * qt + 4 * kq + j) * A16_LDA + wave * 16 + fr] = (h16)o[j];
                }
            }
            __syncthreads();
            // ---- att_result = norm_att(x * alpha + project_out(att)) (:125-126) ----
#pragma unroll
            for (int t = 0; t < S::NTD; t++)
#pragma unroll
                for (int tt = 0; tt < 4; tt++) X[t][tt] *= a.alpha;
            gemm16<S::NTD, S::KSA, A16_LDA, false>(g_out, wout, R, fr, kq, X);
            // (the hidden layer goes FG feature tiles per wave at a time: at d_ff 512 four tiles' accumulators and fragments
            //  would not fit the registers beside X)
            constexpr int FG = S::NTF > 2 ? 2 : S::NTF;
            Ring<FG, S::KSD> g_f0;
            ring_preload(g_f0, wf0);
            ln_partial<S::NTD>(X, red, wave, fr, kq);
            __syncthreads();
            ln_finish<D, S::NTD, S::LDX>(X, red, X16, wave, fr, kq, a.eps);
            __syncthreads();
            // ---- ff_result = norm_ff(att_result * alpha + ff(att_result)) (:128-129) ----
            Ring<S::NTD, S::KSF> g_f1;
#pragma unroll
            for (int fg = 0; fg < S::NTF; fg += FG) {
                f32x4 hid[FG][4];
#pragma unroll
                for (int t = 0; t < FG; t++)
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) hid[t][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm16<FG, S::KSD, S::LDX, false>(g_f0, wf0 + (size_t)fg * S::KSD * 64, X16, fr, kq, hid);
                if (fg + FG < S::NTF) ring_preload(g_f0, wf0 + (size_t)(fg + FG) * S::KSD * 64);
                else ring_preload(g_f1, wf1);
#pragma unroll
                for (int t = 0; t < FG; t++)
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) {
                        h16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; j++) o[j] = (h16)fmaxf(hid[t][tt][j], 0.0f);
                        *reinterpret_cast<h16x4 *>(R + (16 * tt + fr) * S::LDH + (wave * S::NTF + fg + t) * 16 + 4 * kq) = o;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < S::NTD; t++)
#pragma unroll
                for (int tt = 0; tt < 4; tt++) X[t][tt] *= a.alpha;
            gemm16<S::NTD, S::KSF, S::LDH, false>(g_f1, wf1, R, fr, kq, X);
            if (l + 1 < a.depth) ring_preload(g_qkv, wl + (size_t)wave * 3 * S::KSD * 64 + lane);
            ln_partial<S::NTD>(X, red, wave, fr, kq);
            __syncthreads();
            ln_finish<D, S::NTD, S::LDX>(X, red, X16, wave, fr, kq, a.eps);
            __syncthreads();
        }
        // ---- "(h w) b c -> b c h w" (:43-44) as the NHWC rows the head kernels read ----
        {
            constexpr int per_row = D / 8;
            uint4 *dst = reinterpret_cast<uint4 *>(a.y + (size_t)board * A16_TOKENS * D);
            for (int i = tid; i < A16_TOKENS * per_row; i += A16_THREADS) {
                const int r = i / per_row, c = i - r * per_row;
                dst[i] = *reinterpret_cast<const uint4 *>(X16 + r * S::LDX + c * 8);
            }
        }
    }
}

template <int D, int DFF>
void launch1(const AttTower16Dev &d, hipStream_t stream) {
    using S = A16Shape<D, DFF>;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_att_tower_f16<D, DFF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
        done_mask |= 1ull << (dev & 63);
    }
    kz_att_tower_f16<D, DFF><<<d.batch, A16_THREADS, S::LDS_BYTES, stream>>>(d);
}

bool shape_ok(int d_model, int d_ff) {
    return (d_model == 128 && (d_ff == 128 || d_ff == 256)) || (d_model == 256 && (d_ff == 256 || d_ff == 512));
}

// W [N][K] f32 (nn.Linear's weight; rows >= K_src columns zero) -> fragments [N / 16][K / 32][64 lanes][8] f16
void pack_linear(const float *w, int N, int K_src, int K, uint16_t *dst) {
    for (int nt = 0; nt < N / 16; nt++)
        for (int ks = 0; ks < K / 32; ks++)
            for (int lane = 0; lane < 64; lane++)
                for (int i = 0; i < 8; i++) {
                    const int row = 16 * nt + (lane & 15), col = 32 * ks + 8 * (lane >> 4) + i;
                    const _Float16 h = (_Float16)(col < K_src ? w[(size_t)row * K_src + col] : 0.0f);
                    uint16_t bits;
                    __builtin_memcpy(&bits, &h, 2);
                    dst[(((size_t)nt * (K / 32) + ks) * 64 + lane) * 8 + i] = bits;
                }
}

}  // namespace

bool att_tower16_supported(int h, int w, int c_in, int d_model, int heads, int d_k, int d_v, int d_ff, int depth) {
    return h * w == A16_TOKENS && heads == A16_HEADS && d_k == 16 && d_v == 16 && depth >= 1 && c_in >= 1 &&
           (c_in + 31) / 32 * 32 <= A16_MAX_CIN && shape_ok(d_model, d_ff);
}

size_t att_tower16_expand_elems(int d_model, int cin_p) { return (size_t)d_model * cin_p; }
size_t att_tower16_layer_elems(int d_model, int d_ff) {
    return (size_t)8 * 64 * (24 * (d_model / 32) + (d_model / 16) * 4 + (d_ff / 16) * (d_model / 32) + (d_model / 16) * (d_ff / 32));
}

void att_tower16_pack_expand(const float *expand, int d_model, int c_in, int cin_p, uint16_t *dst) {
    pack_linear(expand, d_model, c_in, cin_p, dst);
}

void att_tower16_pack_layer(const float *qkv, const float *out, const float *ff0, const float *ff1, int d_model, int d_ff, uint16_t *dst) {
    pack_linear(qkv, A16_HEADS * 48, d_model, d_model, dst);
    dst += (size_t)A16_HEADS * 48 * d_model;
    pack_linear(out, d_model, A16_HEADS * 16, A16_HEADS * 16, dst);
    dst += (size_t)d_model * A16_HEADS * 16;
    pack_linear(ff0, d_ff, d_model, d_model, dst);
    dst += (size_t)d_ff * d_model;
    pack_linear(ff1, d_model, d_ff, d_ff, dst);
}

void launch_att_tower16(const AttTower16Args &t, hipStream_t stream) {
    if (t.batch <= 0) return;
    AttTower16Dev d{};
    d.x0 = static_cast<const h16 *>(t.x0); d.cin_p = t.cin_p;
    d.w_expand = static_cast<const uint4 *>(t.w_expand); d.embedding = t.embedding;
    d.w_layers = static_cast<const uint4 *>(t.w_layers);
    d.bits = t.bits; d.bits_stride = t.bits_stride; d.scalars_in = t.scalars_in; d.n_scalar = t.n_scalar; d.n_bool = t.n_bool;
    d.y = static_cast<h16 *>(t.y); d.batch = t.batch; d.depth = t.depth; d.alpha = t.alpha; d.eps = t.eps;
    if (t.d_model == 128 && t.d_ff == 128) launch1<128, 128>(d, stream);
    else if (t.d_model == 128 && t.d_ff == 256) launch1<128, 256>(d, stream);
    else if (t.d_model == 256 && t.d_ff == 256) launch1<256, 256>(d, stream);
    else if (t.d_model == 256 && t.d_ff == 512) launch1<256, 512>(d, stream);
}

}  // namespace kz
