// kz_att_tower_mfma.hip — the AttentionTower network (python/lib/model/attention.py:8-136) on the matrix cores, in f16 (v_mfma_f32_16x16x32_f16)
// and, same kernel, second instance, in exact f32 (v_mfma_f32_16x16x4_f32: the <= 1e-4 path): the whole tower of one or two
// 8x8 boards in ONE workgroup of eight waves, ONE launch per batch.  Shapes: 64 squares, 8 heads of d_k = d_v = 16 (what
// python/main/supervised_main_alpha.py:72 builds), d_model / d_ff of the instances at the end of this file; every other
// AttentionTower runs through kz_att_tower.hip (exact f32 on the vector ALUs).
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
// (f32 rows: by 8 values, the same 32 bytes.)
constexpr int A16_THREADS = 512, A16_WAVES = 8, A16_TOKENS = 64, A16_HEADS = 8;
constexpr int A16_MAX_CIN = 224;

// The arithmetic of an instance.  A "fragment" is the 16 bytes a lane holds of an MFMA operand row: 8 f16 values = one
// v_mfma_f32_16x16x32_f16 step of 32 k, or 4 f32 values = FOUR v_mfma_f32_16x16x4_f32 steps of 4 k — step j multiplies the
// lanes' j-th values (lane group kq then stands for k = 4 kq + j: the order of a sum's terms is free), so the f32 instance
// reads its operands with the same 16-byte loads and holds its tiles in the same registers.
struct OpsF16 {
    typedef h16 E;       // LDS / tensor element
    typedef h16x8 Frag;  // a lane's 16 bytes of an operand row
    typedef h16x4 E4;    // a lane's four accumulator values as an operand of the attention's small products
    static constexpr int KB = 32, FE = 8, PAD = 16;
    static __device__ __forceinline__ f32x4 mma(const Frag &a, const Frag &b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mma16(const E4 &a, const E4 &b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ E4 cvt4(const f32x4 &v) { return E4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]}; }
};
struct OpsF32 {
    typedef float E;
    typedef f32x4 Frag;
    typedef f32x4 E4;
    static constexpr int KB = 16, FE = 4, PAD = 8;
    static __device__ __forceinline__ f32x4 mma(const Frag &a, const Frag &b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 4; j++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ f32x4 mma16(const E4 &a, const E4 &b, f32x4 c) { return mma(a, b, c); }
    static __device__ __forceinline__ E4 cvt4(const f32x4 &v) { return v; }
};

struct AttTower16Dev {
    const void *x0;         // encoded planes [batch * 64][cin_p] in the instance's element type (channels >= c_in zero)
    int cin_p;
    const uint4 *w_expand;  // fragments of expand.weight [D][cin_p]
    const float *embedding; // [64][D] f32
    const uint4 *w_layers;  // per layer: project_qkv | project_out | ff.0 | ff.2 fragments
    void *y;                // [batch * 64][D]
    // fused board encode (F0, rust/kz-core/src/mapping/mod.rs:40-63): packed boards straight into the launch (bits == nullptr: x0)
    const uint8_t *bits;
    size_t bits_stride;
    const float *scalars_in;
    int n_scalar, n_bool;
    int batch, depth;
    float alpha, eps;
};

int att_tower16_boards_per_workgroup_impl(int d_ff, int batch);

template <class O, int D, int DFF, int NB>
struct A16Shape {
    static constexpr int ROWS = NB * A16_TOKENS, TT = 4 * NB;  // tokens and 16-token tiles of a workgroup's NB boards
    static constexpr int LDX = D + O::PAD, LDH = DFF + O::PAD, LDA = A16_HEADS * 16 + O::PAD;
    // the second region holds, one after the other: the encoded planes (<= 224 columns + padding), all heads' attention output
    // (128), the feed-forward hidden layer (DFF)
    static constexpr int LDR = LDH > A16_MAX_CIN + O::PAD ? LDH : A16_MAX_CIN + O::PAD;
    static constexpr int X_ELEMS = ROWS * LDX, R_ELEMS = ROWS * LDR;
    static constexpr int RED_FLOATS = 2 * ROWS * A16_WAVES;
    static constexpr size_t LDS_BYTES = (size_t)(X_ELEMS + R_ELEMS) * sizeof(typename O::E) + RED_FLOATS * 4;
    static constexpr int KSD = D / O::KB, KSF = DFF / O::KB, KSA = A16_HEADS * 16 / O::KB;
    static constexpr int NTD = D / 128, NTF = DFF / 128;  // 16-feature tiles per wave
    static constexpr size_t LAYER_FRAGS = (size_t)64 * (24 * KSD + (D / 16) * KSA + (DFF / 16) * KSD + (D / 16) * KSF);
};

template <class F>
__device__ __forceinline__ F as_frag(const uint4 &v) { return *reinterpret_cast<const F *>(&v); }

// The first k-steps' weight fragments of a GEMM, loaded ahead of it (before the barrier and the LayerNorm / attention / store
// that precede it, whose time then hides the loads' latency).
// DEPTH k-steps ahead: four with one board per workgroup, two with two (a k-step then holds twice the MFMAs).
template <int NTW, int KS, int DEPTH>
struct Ring {
    static constexpr int PF = KS < DEPTH ? KS : DEPTH;
    uint4 r[PF][NTW];
};
template <int NTW, int KS, int DEPTH>
__device__ __forceinline__ void ring_preload(Ring<NTW, KS, DEPTH> &g, const uint4 *__restrict__ wf) {
#pragma unroll
    for (int p = 0; p < Ring<NTW, KS, DEPTH>::PF; p++)
#pragma unroll
        for (int t = 0; t < NTW; t++) g.r[p][t] = wf[(size_t)(t * KS + p) * 64];
}

// acc[t][tt] (+)= W tile t (16 features) x token tile tt (TT tiles of 16 tokens) over KS k-steps of 32.  wf: this lane's slot
// of the wave's first tile (tile stride KS * 64 fragments, k-step stride 64), its first k-steps already in `g`; act: LDS rows of
// LDB f16.  VT: the operands exchanged (tokens x features: the accumulator then holds four TOKENS of one feature per lane).
template <class O, int NTW, int KS, int LDB, int TT, bool VT, int DEPTH>
__device__ __forceinline__ void gemm16(Ring<NTW, KS, DEPTH> &g, const uint4 *__restrict__ wf, const typename O::E *act, int fr, int kq,
                                       f32x4 (&acc)[NTW][TT]) {
    typedef typename O::Frag Frag;
    constexpr int PF = Ring<NTW, KS, DEPTH>::PF;
    const typename O::E *brow = act + fr * LDB + O::FE * kq;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
#pragma unroll
        for (int t4 = 0; t4 < TT; t4 += 4) {  // (four token tiles' fragments at a time: sixteen registers)
            Frag b[4];
#pragma unroll
            for (int tt = 0; tt < 4; tt++) b[tt] = *reinterpret_cast<const Frag *>(brow + (t4 + tt) * 16 * LDB + ks * O::KB);
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                const Frag a = as_frag<Frag>(g.r[ks % PF][t]);
#pragma unroll
                for (int tt = 0; tt < 4; tt++) {
                    if (VT) acc[t][t4 + tt] = O::mma(b[tt], a, acc[t][t4 + tt]);
                    else acc[t][t4 + tt] = O::mma(a, b[tt], acc[t][t4 + tt]);
                }
            }
        }
        if (ks + PF < KS) {
#pragma unroll
            for (int t = 0; t < NTW; t++) g.r[ks % PF][t] = wf[(size_t)(t * KS + ks + PF) * 64];
        }
    }
}

// the same with a run-time number of k-steps (the expand layer)
template <class O, int NTW, int TT>
__device__ __forceinline__ void gemm16_rt(const uint4 *__restrict__ wf, int ks_n, const typename O::E *act, int ldb, int fr, int kq,
                                          f32x4 (&acc)[NTW][TT]) {
    typedef typename O::Frag Frag;
    const typename O::E *brow = act + fr * ldb + O::FE * kq;
    for (int ks = 0; ks < ks_n; ks++) {
#pragma unroll
        for (int t4 = 0; t4 < TT; t4 += 4) {
            Frag b[4];
#pragma unroll
            for (int tt = 0; tt < 4; tt++) b[tt] = *reinterpret_cast<const Frag *>(brow + (t4 + tt) * 16 * ldb + ks * O::KB);
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                const Frag a = as_frag<Frag>(wf[(size_t)(t * ks_n + ks) * 64]);
#pragma unroll
                for (int tt = 0; tt < 4; tt++) acc[t][t4 + tt] = O::mma(a, b[tt], acc[t][t4 + tt]);
            }
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
// feature tiles of all tokens.  One exchange between the waves, and no cancellation: a wave reduces its OWN D / 8 features
// of a token in two passes over its registers (sum, then the squared deviations from its own mean), the waves exchange
// (sum, M2) and ln_finish combines them exactly (Chan et al.: M2 = sum_w M2_w + n sum_w (mean_w - mean)^2).  (Rounds 5's
// E[x^2] - mean^2 in one pass lost digits when a token's mean was large against its spread — a trained embedding with a
// common offset in front of the first layer — while the exact-f32 instance is held to 1e-4.)  ln_partial writes this wave's
// pair, the caller synchronises, ln_finish normalises X and writes its f16 copy into X16 (NOT synchronised).
template <int NTD, int TT>
__device__ __forceinline__ void ln_partial(const f32x4 (&X)[NTD][TT], float *red, int wave, int fr, int kq) {
#pragma unroll
    for (int tt = 0; tt < TT; tt++) {
        float s = 0.0f, q = 0.0f;
#pragma unroll
        for (int t = 0; t < NTD; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) s += X[t][tt][j];
        s = group_sum(s);
        const float mw = __fmul_rn(s, 1.0f / (NTD * 16));  // this wave's mean over its NTD * 16 features of the token
#pragma unroll
        for (int t = 0; t < NTD; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float d = __fsub_rn(X[t][tt][j], mw);
                q = fmaf(d, d, q);
            }
        q = group_sum(q);
        if (kq == 0) {
            red[(16 * tt + fr) * A16_WAVES + wave] = s;
            red[TT * 16 * A16_WAVES + (16 * tt + fr) * A16_WAVES + wave] = q;
        }
    }
}
template <class O, int D, int NTD, int TT, int LDX>
__device__ __forceinline__ void ln_finish(f32x4 (&X)[NTD][TT], const float *red, typename O::E *X16, int wave, int fr, int kq, float eps) {
    static_assert(A16_WAVES == 8 && NTD * 16 * A16_WAVES == D, "eight waves share a token's D features evenly");
#pragma unroll
    for (int tt = 0; tt < TT; tt++) {
        const f32x4 *ps = reinterpret_cast<const f32x4 *>(red + (16 * tt + fr) * A16_WAVES);
        const f32x4 *pq = reinterpret_cast<const f32x4 *>(red + TT * 16 * A16_WAVES + (16 * tt + fr) * A16_WAVES);
        const f32x4 s0 = ps[0], s1 = ps[1], q0 = pq[0], q1 = pq[1];
        const float sum = ((s0[0] + s0[1]) + (s0[2] + s0[3])) + ((s1[0] + s1[1]) + (s1[2] + s1[3]));
        const float mean = __fmul_rn(sum, 1.0f / D);
        float m2 = ((q0[0] + q0[1]) + (q0[2] + q0[3])) + ((q1[0] + q1[1]) + (q1[2] + q1[3]));
        float between = 0.0f;  // sum over the waves of (mean_w - mean)^2
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float d0 = fmaf(s0[w], 1.0f / (NTD * 16), -mean), d1 = fmaf(s1[w], 1.0f / (NTD * 16), -mean);
            between = fmaf(d0, d0, between);
            between = fmaf(d1, d1, between);
        }
        m2 = fmaf(between, (float)(NTD * 16), m2);
        const float inv = __builtin_amdgcn_rsqf(fmaf(m2, 1.0f / D, eps));  // biased variance M2 / D
        const float shift = __fmul_rn(-mean, inv);
#pragma unroll
        for (int t = 0; t < NTD; t++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                X[t][tt][j] = fmaf(X[t][tt][j], inv, shift);
                // the f16 copy is the rounding of THIS f32 value in every instance: without the barrier the compiler may fold
                // fma + conversion into one v_fma_mix (a single rounding) in one instance and not in another
                asm volatile("" : "+v"(X[t][tt][j]));
            }
            *reinterpret_cast<typename O::E4 *>(X16 + (16 * tt + fr) * LDX + (wave * NTD + t) * 16 + 4 * kq) = O::cvt4(X[t][tt]);
        }
    }
}

// NB boards per workgroup: every weight fragment a wave loads multiplies NB * 64 tokens.  The launch is bound by what a CU
// can pull from its L2 (8.4 MB of fragments per workgroup at d_model = d_ff = 256, 16 layers: 38 GB/s per CU with one board —
// the chip's L2 serves ~70), so two boards per workgroup halve the bytes per board.
template <class O, int D, int DFF, int NB>
__global__ __launch_bounds__(A16_THREADS) void kz_att_tower_mfma(AttTower16Dev a) {
    using S = A16Shape<O, D, DFF, NB>;
    typedef typename O::E E;
    typedef typename O::E4 E4;
    constexpr int TT = S::TT, RD = NB == 1 ? 4 : 2, FE = O::FE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    E *X16 = reinterpret_cast<E *>(lds_raw);  // the tokens' rows the next layer multiplies (f16 / f32 copy of the residual stream)
    E *R = X16 + S::X_ELEMS;                  // encoded planes / attention output / hidden layer
    float *red = reinterpret_cast<float *>(R + S::R_ELEMS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, kq = lane >> 4;
    const int ks_in = a.cin_p / O::KB, ldi = a.cin_p + O::PAD;
    const float eps_scaled = a.eps / (a.alpha * a.alpha);

    for (int board0 = blockIdx.x * NB; board0 < a.batch; board0 += gridDim.x * NB) {
        // ---- the boards' encoded planes -> LDS (a board past the batch's end repeats the last one and is not stored) ----
        __syncthreads();
        const int per_row = a.cin_p / FE;
        if (a.bits) {
            // scalar planes first, each broadcast over the board, then the bool planes: bool i of a board = bit i % 8 of byte
            // i / 8 (bit_buffer.rs:73-75), i = plane * 64 + square
            for (int i = tid; i < S::ROWS * per_row; i += A16_THREADS) {
                const int c = i / S::ROWS, r = i - c * S::ROWS, board = min(board0 + r / A16_TOKENS, a.batch - 1), sq = r % A16_TOKENS;
                const uint8_t *bb = a.bits + (size_t)board * a.bits_stride;
                const float *sc = a.scalars_in + (size_t)board * a.n_scalar;
                typename O::Frag v;
#pragma unroll
                for (int j = 0; j < FE; j++) {
                    const int ch = c * FE + j;
                    float f = 0.0f;
                    if (ch < a.n_scalar) {
                        f = sc[ch];
                    } else if (ch < a.n_scalar + a.n_bool) {
                        const unsigned bit = (unsigned)(ch - a.n_scalar) * A16_TOKENS + sq;
                        f = (float)((bb[bit >> 3] >> (bit & 7)) & 1);
                    }
                    v[j] = (E)f;
                }
                *reinterpret_cast<typename O::Frag *>(R + r * ldi + c * FE) = v;
            }
        } else {
            for (int i = tid; i < S::ROWS * per_row; i += A16_THREADS) {
                const int r = i / per_row, c = i - r * per_row, board = min(board0 + r / A16_TOKENS, a.batch - 1);
                *reinterpret_cast<uint4 *>(R + r * ldi + c * FE) =
                    *reinterpret_cast<const uint4 *>(static_cast<const E *>(a.x0) + ((size_t)board * A16_TOKENS + r % A16_TOKENS) * a.cin_p + c * FE);
            }
        }
        __syncthreads();
        // ---- expand + embedding (attention.py:39-40): X[feature][token] ----
        f32x4 X[S::NTD][TT];
#pragma unroll
        for (int t = 0; t < S::NTD; t++)
#pragma unroll
            for (int tt = 0; tt < TT; tt++)
                X[t][tt] = *reinterpret_cast<const f32x4 *>(a.embedding + (size_t)(16 * (tt & 3) + fr) * D + (wave * S::NTD + t) * 16 + 4 * kq);
        gemm16_rt<O, S::NTD, TT>(a.w_expand + (size_t)wave * S::NTD * ks_in * 64 + lane, ks_in, R, ldi, fr, kq, X);
#pragma unroll
        for (int t = 0; t < S::NTD; t++)
#pragma unroll
            for (int tt = 0; tt < TT; tt++)
                *reinterpret_cast<E4 *>(X16 + (16 * tt + fr) * S::LDX + (wave * S::NTD + t) * 16 + 4 * kq) = O::cvt4(X[t][tt]);

        // a GEMM's first weight fragments are requested one phase ahead: q | k's behind the previous layer's last GEMM (here:
        // behind the expand layer), v's behind q | k's GEMM, project_out's ahead of the attention, ff.0's and ff.2's ahead of
        // the LayerNorms
        Ring<2, S::KSD, RD> g_qk;
        const uint4 *wl = a.w_layers;
        ring_preload(g_qk, wl + (size_t)wave * 3 * S::KSD * 64 + lane);
        __syncthreads();
        for (int l = 0; l < a.depth; l++) {
            const uint4 *wqkv = wl + (size_t)wave * 3 * S::KSD * 64 + lane;
            const uint4 *wout = wl + (size_t)64 * 24 * S::KSD + (size_t)wave * S::NTD * S::KSA * 64 + lane;
            const uint4 *wf0 = wl + (size_t)64 * (24 * S::KSD + (D / 16) * S::KSA) + (size_t)wave * S::NTF * S::KSD * 64 + lane;
            const uint4 *wf1 = wl + (size_t)64 * (24 * S::KSD + (D / 16) * S::KSA + (DFF / 16) * S::KSD) + (size_t)wave * S::NTD * S::KSF * 64 + lane;
            wl += S::LAYER_FRAGS;
            Ring<S::NTD, S::KSA, RD> g_out;
            // ---- q, k, v of head `wave` and its attention, in registers ----
            {
                E4 qf[TT], kf[TT], vf[TT];
                Ring<1, S::KSD, RD> g_v;
                {
                    f32x4 qk[2][TT];
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int tt = 0; tt < TT; tt++) qk[t][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    gemm16<O, 2, S::KSD, S::LDX, TT, false, RD>(g_qk, wqkv, X16, fr, kq, qk);
                    ring_preload(g_v, wqkv + (size_t)2 * S::KSD * 64);
#pragma unroll
                    for (int tt = 0; tt < TT; tt++) {
                        qf[tt] = O::cvt4(qk[0][tt]);  // q[token 16 tt + fr][feature 4 kq + j]
                        kf[tt] = O::cvt4(qk[1][tt]);
                    }
                }
                {
                    f32x4 v[1][TT];
#pragma unroll
                    for (int tt = 0; tt < TT; tt++) v[0][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    gemm16<O, 1, S::KSD, S::LDX, TT, true, RD>(g_v, wqkv + (size_t)2 * S::KSD * 64, X16, fr, kq, v);
                    ring_preload(g_out, wout);
#pragma unroll
                    for (int tt = 0; tt < TT; tt++) vf[tt] = O::cvt4(v[0][tt]);  // v[token 16 tt + 4 kq + j][feature fr]
                }
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
#pragma unroll
                    for (int qt = 0; qt < 4; qt++) {
                        // logits^T [key][query] = k q^T: lane holds query 16 qt + fr, keys 16 kt + 4 kq + j
                        f32x4 sc[4];
                        float mx = -INFINITY;
#pragma unroll
                        for (int kt = 0; kt < 4; kt++) {
                            sc[kt] = O::mma16(kf[4 * nb + kt], qf[4 * nb + qt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                            for (int j = 0; j < 4; j++) mx = fmaxf(mx, sc[kt][j]);
                        }
                        mx = group_max(mx);
                        float sum = 0.0f;
#pragma unroll
                        for (int kt = 0; kt < 4; kt++)
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                sc[kt][j] = __builtin_amdgcn_exp2f(sc[kt][j] - mx);  // (q carries log2(e): att_tower16_pack_layer)
                                sum += sc[kt][j];
                            }
                        const float inv = __builtin_amdgcn_rcpf(group_sum(sum));
                        // att[query][feature] = weights v: A = weights (query fr, keys 4 kq ..), B = v (keys 4 kq .., feature fr)
                        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kt = 0; kt < 4; kt++) {
                            o = O::mma16(O::cvt4(sc[kt] * inv), vf[4 * nb + kt], o);
                        }
                        // o: query 16 qt + 4 kq + j, feature fr of this head
#pragma unroll
                        for (int j = 0; j < 4; j++) R[(64 * nb + 16 * qt + 4 * kq + j) * S::LDA + wave * 16 + fr] = (E)o[j];
                    }
            }
            __syncthreads();
            // ---- att_result = norm_att(x * alpha + project_out(att)) (:125-126) = LayerNorm with eps / alpha^2 of
            // x + project_out(att) / alpha: project_out and ff.2 carry the 1 / alpha (att_tower16_pack_layer) ----
            gemm16<O, S::NTD, S::KSA, S::LDA, TT, false, RD>(g_out, wout, R, fr, kq, X);
            // (the hidden layer goes FG feature tiles per wave at a time: at d_ff 512 four tiles' accumulators and fragments
            //  would not fit the registers beside X, with two boards per workgroup two tiles' would not)
            constexpr int FG = NB == 2 ? 1 : S::NTF > 2 ? 2 : S::NTF;
            Ring<FG, S::KSD, RD> g_f0;
            ring_preload(g_f0, wf0);
            ln_partial<S::NTD, TT>(X, red, wave, fr, kq);
            __syncthreads();
            ln_finish<O, D, S::NTD, TT, S::LDX>(X, red, X16, wave, fr, kq, eps_scaled);
            __syncthreads();
            // ---- ff_result = norm_ff(att_result * alpha + ff(att_result)) (:128-129) ----
            Ring<S::NTD, S::KSF, RD> g_f1;
#pragma unroll
            for (int fg = 0; fg < S::NTF; fg += FG) {
                f32x4 hid[FG][TT];
#pragma unroll
                for (int t = 0; t < FG; t++)
#pragma unroll
                    for (int tt = 0; tt < TT; tt++) hid[t][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm16<O, FG, S::KSD, S::LDX, TT, false, RD>(g_f0, wf0 + (size_t)fg * S::KSD * 64, X16, fr, kq, hid);
                if (fg + FG < S::NTF) ring_preload(g_f0, wf0 + (size_t)(fg + FG) * S::KSD * 64);
                else ring_preload(g_f1, wf1);
#pragma unroll
                for (int t = 0; t < FG; t++)
#pragma unroll
                    for (int tt = 0; tt < TT; tt++) {
                        f32x4 o;
#pragma unroll
                        for (int j = 0; j < 4; j++) o[j] = fmaxf(hid[t][tt][j], 0.0f);
                        *reinterpret_cast<E4 *>(R + (16 * tt + fr) * S::LDH + (wave * S::NTF + fg + t) * 16 + 4 * kq) = O::cvt4(o);
                    }
            }
            __syncthreads();
            gemm16<O, S::NTD, S::KSF, S::LDH, TT, false, RD>(g_f1, wf1, R, fr, kq, X);
            if (l + 1 < a.depth) ring_preload(g_qk, wl + (size_t)wave * 3 * S::KSD * 64 + lane);
            ln_partial<S::NTD, TT>(X, red, wave, fr, kq);
            __syncthreads();
            ln_finish<O, D, S::NTD, TT, S::LDX>(X, red, X16, wave, fr, kq, eps_scaled);
            __syncthreads();
        }
        // ---- "(h w) b c -> b c h w" (:43-44) as the NHWC rows the head kernels read ----
        {
            constexpr int per_out = D / FE;
            for (int i = tid; i < S::ROWS * per_out; i += A16_THREADS) {
                const int r = i / per_out, c = i - r * per_out, board = board0 + r / A16_TOKENS;
                if (board < a.batch)
                    *reinterpret_cast<uint4 *>(static_cast<E *>(a.y) + ((size_t)board * A16_TOKENS + r % A16_TOKENS) * D + c * FE) =
                        *reinterpret_cast<const uint4 *>(X16 + r * S::LDX + c * FE);
            }
        }
    }
}

template <class O, int D, int DFF, int NB>
void launch1(const AttTower16Dev &d, hipStream_t stream) {
    using S = A16Shape<O, D, DFF, NB>;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_att_tower_mfma<O, D, DFF, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
        done_mask |= 1ull << (dev & 63);
    }
    kz_att_tower_mfma<O, D, DFF, NB><<<(d.batch + NB - 1) / NB, A16_THREADS, S::LDS_BYTES, stream>>>(d);
}

// two boards per workgroup where their LDS images fit (f16, d_ff <= 256) and the batch still gives the chip's 256 CUs a
// workgroup each between the two launches an engine pair keeps in flight
template <class O, int D, int DFF>
void launch_nb(const AttTower16Dev &d, hipStream_t stream) {
    if constexpr (A16Shape<O, D, DFF, 2>::LDS_BYTES <= 160 * 1024) {
        if (att_tower16_boards_per_workgroup_impl(DFF, d.batch) == 2) {
            launch1<O, D, DFF, 2>(d, stream);
            return;
        }
    }
    launch1<O, D, DFF, 1>(d, stream);
}

int att_tower16_boards_per_workgroup_impl(int d_ff, int batch) {
#ifdef KZ_ATT16_ONE_BOARD  // (diagnostic builds: tools/ab_att.sh)
    return 1;
#else
    return d_ff <= 256 && batch >= 192 ? 2 : 1;
#endif
}

// the instances: f16 — (128, 128), (128, 256), (256, 256), (256, 512); exact f32 (one board's two f32 images fill the LDS) —
// (128, 128), (128, 256), (256, 256)
bool shape_ok(int d_model, int d_ff, bool f32) {
    return (d_model == 128 && (d_ff == 128 || d_ff == 256)) || (d_model == 256 && (d_ff == 256 || (d_ff == 512 && !f32)));
}

// W [N][K_src] f32 (nn.Linear's weight; columns K_src .. K zero) -> fragments [N / 16][K / KB][64 lanes][FE] in the instance's
// element type (KB = 32, FE = 8 in f16; KB = 16, FE = 4 in f32): lane 16 kq + fr holds W[16 nt + fr][KB ks + FE kq ..].  Row r
// scaled by `scale` — with q_rows, only the rows r % 48 < 16 (the q rows of project_qkv).
void pack_linear(const float *w, int N, int K_src, int K, bool f32, void *dst_v, float scale = 1.0f, bool q_rows = false) {
    const int KB = f32 ? 16 : 32, FE = f32 ? 4 : 8;
    for (int nt = 0; nt < N / 16; nt++)
        for (int ks = 0; ks < K / KB; ks++)
            for (int lane = 0; lane < 64; lane++)
                for (int i = 0; i < FE; i++) {
                    const int row = 16 * nt + (lane & 15), col = KB * ks + FE * (lane >> 4) + i;
                    const float sc = !q_rows || row % 48 < 16 ? scale : 1.0f;
                    const float v = col < K_src ? w[(size_t)row * K_src + col] * sc : 0.0f;
                    const size_t at = (((size_t)nt * (K / KB) + ks) * 64 + lane) * FE + i;
                    if (f32) {
                        static_cast<float *>(dst_v)[at] = v;
                    } else {
                        const _Float16 h = (_Float16)v;
                        __builtin_memcpy(static_cast<uint16_t *>(dst_v) + at, &h, 2);
                    }
                }
}

}  // namespace

bool att_tower16_supported(int h, int w, int c_in, int d_model, int heads, int d_k, int d_v, int d_ff, int depth, bool f32) {
    return h * w == A16_TOKENS && heads == A16_HEADS && d_k == 16 && d_v == 16 && depth >= 1 && c_in >= 1 &&
           (c_in + 31) / 32 * 32 <= A16_MAX_CIN && shape_ok(d_model, d_ff, f32);
}

int att_tower16_boards_per_workgroup(int d_model, int d_ff, int batch, bool f32) {
    return !shape_ok(d_model, d_ff, f32) ? 0 : f32 ? 1 : att_tower16_boards_per_workgroup_impl(d_ff, batch);
}

size_t att_tower16_expand_elems(int d_model, int cin_p) { return (size_t)d_model * cin_p; }
size_t att_tower16_layer_elems(int d_model, int d_ff) {
    return (size_t)A16_HEADS * 48 * d_model + (size_t)d_model * A16_HEADS * 16 + (size_t)2 * d_ff * d_model;
}

void att_tower16_pack_expand(const float *expand, int d_model, int c_in, int cin_p, bool f32, void *dst) {
    pack_linear(expand, d_model, c_in, cin_p, f32, dst);
}

// q rows carry log2(e) (the softmax is then exp2 of the logits' differences), project_out and ff.2 carry 1 / alpha (the DeepNorm
// residual x * alpha + f(x) under a LayerNorm = x + f(x) / alpha under the LayerNorm with eps / alpha^2)
void att_tower16_pack_layer(const float *qkv, const float *out, const float *ff0, const float *ff1, int d_model, int d_ff, float alpha,
                            bool f32, void *dst_v) {
    const size_t esz = f32 ? 4 : 2;
    char *dst = static_cast<char *>(dst_v);
    pack_linear(qkv, A16_HEADS * 48, d_model, d_model, f32, dst, 1.4426950408889634f, true);
    dst += (size_t)A16_HEADS * 48 * d_model * esz;
    pack_linear(out, d_model, A16_HEADS * 16, A16_HEADS * 16, f32, dst, 1.0f / alpha);
    dst += (size_t)d_model * A16_HEADS * 16 * esz;
    pack_linear(ff0, d_ff, d_model, d_model, f32, dst);
    dst += (size_t)d_ff * d_model * esz;
    pack_linear(ff1, d_model, d_ff, d_ff, f32, dst, 1.0f / alpha);
}

void launch_att_tower16(const AttTower16Args &t, hipStream_t stream) {
    if (t.batch <= 0) return;
    AttTower16Dev d{};
    d.x0 = t.x0; d.cin_p = t.cin_p;
    d.w_expand = static_cast<const uint4 *>(t.w_expand); d.embedding = t.embedding;
    d.w_layers = static_cast<const uint4 *>(t.w_layers);
    d.bits = t.bits; d.bits_stride = t.bits_stride; d.scalars_in = t.scalars_in; d.n_scalar = t.n_scalar; d.n_bool = t.n_bool;
    d.y = t.y; d.batch = t.batch; d.depth = t.depth; d.alpha = t.alpha; d.eps = t.eps;
    if (t.f32) {
        if (t.d_model == 128 && t.d_ff == 128) launch1<OpsF32, 128, 128, 1>(d, stream);
        else if (t.d_model == 128 && t.d_ff == 256) launch1<OpsF32, 128, 256, 1>(d, stream);
        else if (t.d_model == 256 && t.d_ff == 256) launch1<OpsF32, 256, 256, 1>(d, stream);
        return;
    }
    if (t.d_model == 128 && t.d_ff == 128) launch_nb<OpsF16, 128, 128>(d, stream);
    else if (t.d_model == 128 && t.d_ff == 256) launch_nb<OpsF16, 128, 256>(d, stream);
    else if (t.d_model == 256 && t.d_ff == 256) launch_nb<OpsF16, 256, 256>(d, stream);
    else if (t.d_model == 256 && t.d_ff == 512) launch_nb<OpsF16, 256, 512>(d, stream);
}

}  // namespace kz
