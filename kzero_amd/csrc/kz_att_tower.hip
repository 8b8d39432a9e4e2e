// kz_att_tower.hip — the AttentionTower network (python/lib/model/attention.py:8-136, the tower python/main/supervised_main_alpha.py:72
// trains in place of the ResTower) in exact f32: the whole tower of one board in ONE workgroup, ONE launch per batch.
//
// Every square of the board is a token of d_model features.  The token matrix X [n][d_model] stays in LDS from the expand
// layer to the last encoder layer; the weights are read from global memory as they are stored in the model ([out][in] rows
// of nn.Linear, attention.py:69-78), a 64 x 32 tile at a time through LDS.  Per encoder layer (forward_with_weights,
// attention.py:97-133), head after head: qkv of ONE head (a [n][2 d_k + d_v] GEMM), logits q k^T without scale factor (:117),
// softmax over the keys (:119) and weights v (:122) a query row per wave; then project_out, the DeepNorm residual
// LayerNorm(x * alpha + f(x)) (:126), the feed-forward pair of Linear layers (:128) and the second LayerNorm (:129).
// All arithmetic is f32 FMA (this is the <= 1e-4 path, and the path of every shape the matrix-core kernel kz_att_tower_mfma.hip does not
// take); the rows it reads and writes are f32 or f16 as the engine's other kernels expect them.
#include "kz_kernels.hpp"

namespace kz {
namespace {

typedef _Float16 h16;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AT_THREADS = 256, AT_WAVES = AT_THREADS / 64;
constexpr int AT_WT_LD = 36;   // the staged weight tile [64 columns][32 k]: rows 36 floats apart (16-byte reads without bank conflicts)
constexpr int AT_KMAX = 6;     // keys per lane in the softmax: boards of up to 384 squares (Go 19x19: 361)

struct AttGeo {
    int n, D, H, dk, dv, dff, c_in;
    int ldx, ldq, lda, ldt, ldi, np;  // row strides (floats) of X, QKV, ATT, TMP, IN; np = n rounded up to 64 (a wave's weights row)
    int off_r, off_w;                 // float offsets of the overlaid region and of the weight tile
    int ff_rows;                      // tokens per pass of the feed-forward pair (all n where the hidden layer fits the region)
    size_t lds_floats;
};

constexpr size_t AT_MAX_LDS = 160 * 1024;

__host__ __device__ inline int at_round_up(int v, int m) { return (v + m - 1) / m * m; }

__host__ __device__ inline AttGeo att_geo(int n, int c_in, int D, int H, int dk, int dv, int dff) {
    AttGeo g;
    g.n = n; g.D = D; g.H = H; g.dk = dk; g.dv = dv; g.dff = dff; g.c_in = c_in;
    g.ldx = at_round_up(D, 4) + 4;
    g.ldq = at_round_up(2 * dk + dv, 4) + 4;
    g.lda = at_round_up(H * dv, 4) + 4;
    g.ldt = at_round_up(dff, 4) + 4;
    g.ldi = at_round_up(c_in, 4) + 4;
    g.np = at_round_up(n, 64);
    const int attention = n * g.ldq + n * g.lda + AT_WAVES * g.np, in = n * g.ldi;
    g.off_r = n * g.ldx;
    // the feed-forward pair is token-wise: where [n][d_ff] does not fit beside X, it runs over as many tokens at a time as do
    const long budget = (long)(AT_MAX_LDS / 4) - g.off_r - 64 * AT_WT_LD - 4;
    g.ff_rows = budget / g.ldt >= n ? n : (int)(budget / g.ldt > 0 ? budget / g.ldt : 0);
    const int ff = g.ff_rows * g.ldt;
    const int r = attention > ff ? (attention > in ? attention : in) : (ff > in ? ff : in);
    g.off_w = g.off_r + at_round_up(r, 4);
    g.lds_floats = (size_t)g.off_w + 64 * AT_WT_LD;
    return g;
}

struct AttTowerDev {
    const void *x0;   // encoded planes [batch * n][ldx0], f32 or f16 (channels >= c_in zero)
    int ldx0, in_f16;
    const float *expand, *embedding;  // [D][c_in], [n][D]
    const float *layers;              // per layer: qkv [H (2 dk + dv)][D] | out [D][H dv] | ff0 [dff][D] | ff1 [D][dff]
    void *y;          // tower output rows [batch * n][ldy], f32 or f16; columns D .. ldy zero
    int ldy, out_f16;
    int batch, depth;
    float alpha, eps;
    AttGeo g;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// C[r][c] = sum_k A[r][k] * W[c][k] for r < n, c < N: A in LDS (rows lda floats apart, lda % 4 == 0), W in global memory as
// nn.Linear stores it ([N][K]).  16 x 16 threads, each 4 rows x 4 columns of a 64 x 64 tile; W a [64][32] tile at a time
// through `wst`.  epi(r, c, value) for every element; the caller synchronises before anybody reads what epi wrote.
template <class Epi>
__device__ __forceinline__ void gemm_rows(const float *A, int lda, int n, int K, const float *__restrict__ W, int N, float *wst,
                                          Epi epi) {
    const int tid = threadIdx.x, cg = tid & 15, rg = tid >> 4;
    const bool vec_w = (K & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;  // (16-byte loads of W's rows)
    for (int c0 = 0; c0 < N; c0 += 64) {
        for (int r0 = 0; r0 < n; r0 += 64) {
            float acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = 0.0f;
            int rows[4];
#pragma unroll
            for (int i = 0; i < 4; i++) rows[i] = min(r0 + rg + 16 * i, n - 1);  // (rows beyond n read row n - 1 and are dropped)
            for (int k0 = 0; k0 < K; k0 += 32) {
                const int kt = min(32, K - k0);
                __syncthreads();  // the previous tile's readers are done
                {
                    const int c = tid >> 2, kk = (tid & 3) * 8, col = c0 + c;
                    float v[8];
                    if (col < N && vec_w && kk + 8 <= kt) {
                        const f32x4 *src = reinterpret_cast<const f32x4 *>(W + (size_t)col * K + k0 + kk);
                        const f32x4 lo = src[0], hi = src[1];
                        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
                        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) v[j] = (col < N && kk + j < kt) ? W[(size_t)col * K + k0 + kk + j] : 0.0f;
                    }
                    float *dst = wst + c * AT_WT_LD + kk;
                    *reinterpret_cast<f32x4 *>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
                __syncthreads();
                if (kt == 32) {
#pragma unroll 2
                    for (int k4 = 0; k4 < 8; k4++) {
                        f32x4 a[4], w[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) a[i] = *reinterpret_cast<const f32x4 *>(A + (size_t)rows[i] * lda + k0 + 4 * k4);
#pragma unroll
                        for (int j = 0; j < 4; j++) w[j] = *reinterpret_cast<const f32x4 *>(wst + (cg + 16 * j) * AT_WT_LD + 4 * k4);
#pragma unroll
                        for (int i = 0; i < 4; i++)
#pragma unroll
                            for (int j = 0; j < 4; j++)
#pragma unroll
                                for (int e = 0; e < 4; e++) acc[i][j] = fmaf(a[i][e], w[j][e], acc[i][j]);
                    }
                } else {
                    for (int k = 0; k < kt; k++) {
                        float a[4], w[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) a[i] = A[(size_t)rows[i] * lda + k0 + k];
#pragma unroll
                        for (int j = 0; j < 4; j++) w[j] = wst[(cg + 16 * j) * AT_WT_LD + k];
#pragma unroll
                        for (int i = 0; i < 4; i++)
#pragma unroll
                            for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int r = r0 + rg + 16 * i, c = c0 + cg + 16 * j;
                    if (r < n && c < N) epi(r, c, acc[i][j]);
                }
        }
    }
}

// nn.LayerNorm(D, elementwise_affine=False) over every row of X (biased variance, two passes), a row per wave at a time
__device__ __forceinline__ void layernorm_rows(float *X, int ldx, int n, int D, float eps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < n; r += AT_WAVES) {
        float *row = X + (size_t)r * ldx;
        float s = 0.0f;
        for (int c = lane; c < D; c += 64) s += row[c];
        const float mean = wave_sum(s) / (float)D;
        float q = 0.0f;
        for (int c = lane; c < D; c += 64) {
            const float d = row[c] - mean;
            q = fmaf(d, d, q);
        }
        const float inv = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
        for (int c = lane; c < D; c += 64) row[c] = (row[c] - mean) * inv;
    }
}

__global__ __launch_bounds__(AT_THREADS) void kz_att_tower_f32(AttTowerDev a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const AttGeo &g = a.g;
    const int n = g.n, D = g.D, H = g.H, dk = g.dk, dv = g.dv, dff = g.dff, dkqv = 2 * dk + dv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float *X = lds, *R = lds + g.off_r, *wst = lds + g.off_w;
    float *QKV = R, *ATT = R + n * g.ldq, *P = ATT + n * g.lda + wave * g.np;  // attention phase
    float *TMP = R;                                                            // feed-forward phase
    float *IN = R;                                                             // expand

    for (int board = blockIdx.x; board < a.batch; board += gridDim.x) {
        // ---- "b c h w -> (h w) b c", expand + embedding (attention.py:35-40) ----
        __syncthreads();
        for (int i = tid; i < n * g.c_in; i += AT_THREADS) {
            const int p = i / g.c_in, c = i - p * g.c_in;
            const size_t src = ((size_t)board * n + p) * a.ldx0 + c;
            IN[p * g.ldi + c] = a.in_f16 ? (float)reinterpret_cast<const h16 *>(a.x0)[src] : reinterpret_cast<const float *>(a.x0)[src];
        }
        __syncthreads();
        gemm_rows(IN, g.ldi, n, g.c_in, a.expand, D, wst,
                  [&](int r, int c, float v) { X[r * g.ldx + c] = v + a.embedding[(size_t)r * D + c]; });
        __syncthreads();

        const float *w = a.layers;
        for (int l = 0; l < a.depth; l++) {
            const float *wqkv = w, *wout = wqkv + (size_t)H * dkqv * D, *wf0 = wout + (size_t)D * H * dv, *wf1 = wf0 + (size_t)dff * D;
            w = wf1 + (size_t)D * dff;
            for (int h = 0; h < H; h++) {
                // qkv of head h: rows h * dkqv .. of project_qkv (the view(n, b * heads, d_kqv) of attention.py:106)
                gemm_rows(X, g.ldx, n, D, wqkv + (size_t)h * dkqv * D, dkqv, wst, [&](int r, int c, float v) { QKV[r * g.ldq + c] = v; });
                __syncthreads();
                for (int q = wave; q < n; q += AT_WAVES) {
                    const float *qrow = QKV + q * g.ldq;
                    float lg[AT_KMAX], mx = -INFINITY;
#pragma unroll
                    for (int t = 0; t < AT_KMAX; t++) {
                        const int key = lane + 64 * t;
                        lg[t] = -INFINITY;
                        if (key < n) {
                            const float *krow = QKV + key * g.ldq + dk;
                            float acc = 0.0f;
                            for (int d = 0; d < dk; d++) acc = fmaf(qrow[d], krow[d], acc);
                            lg[t] = acc;
                            mx = fmaxf(mx, acc);
                        }
                    }
                    mx = wave_max(mx);
                    float sum = 0.0f;
#pragma unroll
                    for (int t = 0; t < AT_KMAX; t++) {
                        lg[t] = lane + 64 * t < n ? expf(lg[t] - mx) : 0.0f;
                        sum += lg[t];
                    }
                    sum = wave_sum(sum);
#pragma unroll
                    for (int t = 0; t < AT_KMAX; t++)
                        if (lane + 64 * t < n) P[lane + 64 * t] = lg[t] / sum;
                    __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations complete in order: its own row of P)
                    for (int j = lane; j < dv; j += 64) {
                        float acc = 0.0f;
                        for (int key = 0; key < n; key++) acc = fmaf(P[key], QKV[key * g.ldq + 2 * dk + j], acc);
                        ATT[q * g.lda + h * dv + j] = acc;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                __syncthreads();
            }
            // att_result = norm_att(input * alpha + project_out(att)) (:125-126)
            gemm_rows(ATT, g.lda, n, H * dv, wout, D, wst, [&](int r, int c, float v) { X[r * g.ldx + c] = fmaf(X[r * g.ldx + c], a.alpha, v); });
            __syncthreads();
            layernorm_rows(X, g.ldx, n, D, a.eps);
            __syncthreads();
            // ff_result = norm_ff(att_result * alpha + ff(att_result)) (:128-129); ff = Linear, ReLU, Linear (:74-78)
            for (int r0 = 0; r0 < n; r0 += g.ff_rows) {
                const int nr = min(g.ff_rows, n - r0);
                float *Xr = X + r0 * g.ldx;
                gemm_rows(Xr, g.ldx, nr, D, wf0, dff, wst, [&](int r, int c, float v) { TMP[r * g.ldt + c] = fmaxf(v, 0.0f); });
                __syncthreads();
                gemm_rows(TMP, g.ldt, nr, dff, wf1, D, wst, [&](int r, int c, float v) { Xr[r * g.ldx + c] = fmaf(Xr[r * g.ldx + c], a.alpha, v); });
                __syncthreads();
            }
            layernorm_rows(X, g.ldx, n, D, a.eps);
            __syncthreads();
        }
        // ---- "(h w) b c -> b c h w" (:43-44): as the NHWC rows the head kernels read ----
        for (int i = tid; i < n * a.ldy; i += AT_THREADS) {
            const int p = i / a.ldy, c = i - p * a.ldy;
            const float v = c < D ? X[p * g.ldx + c] : 0.0f;
            const size_t dst = ((size_t)board * n + p) * a.ldy + c;
            if (a.out_f16) reinterpret_cast<h16 *>(a.y)[dst] = (h16)v;
            else reinterpret_cast<float *>(a.y)[dst] = v;
        }
    }
}

}  // namespace

bool att_tower_supported(int h, int w, int c_in, int d_model, int heads, int d_k, int d_v, int d_ff, int depth) {
    const int n = h * w;
    if (n < 1 || n > 64 * AT_KMAX || depth < 1 || c_in < 1 || d_model < 1 || heads < 1 || d_k < 1 || d_v < 1 || d_ff < 1) return false;
    if ((size_t)n * (d_model + d_ff + heads * (2 * d_k + d_v)) > ((size_t)1 << 24)) return false;  // (before any product can overflow)
    const AttGeo g = att_geo(n, c_in, d_model, heads, d_k, d_v, d_ff);
    return g.lds_floats * 4 <= AT_MAX_LDS && g.ff_rows >= (n < 16 ? n : 16);
}

size_t att_tower_layer_elems(int d_model, int heads, int d_k, int d_v, int d_ff) {
    return (size_t)heads * (2 * d_k + d_v) * d_model + (size_t)d_model * heads * d_v + (size_t)2 * d_ff * d_model;
}

void launch_att_tower(const AttTowerArgs &t, hipStream_t stream) {
    AttTowerDev d{};
    d.x0 = t.x0; d.ldx0 = t.ldx0; d.in_f16 = t.in_f16;
    d.expand = t.expand; d.embedding = t.embedding; d.layers = t.layers;
    d.y = t.y; d.ldy = t.ldy; d.out_f16 = t.out_f16;
    d.batch = t.batch; d.depth = t.depth; d.alpha = t.alpha; d.eps = t.eps;
    d.g = att_geo(t.h * t.w, t.c_in, t.d_model, t.heads, t.d_k, t.d_v, t.d_ff);
    const size_t lds = d.g.lds_floats * 4;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_att_tower_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AT_MAX_LDS);
        done_mask |= 1ull << (dev & 63);
    }
    if (t.batch <= 0) return;
    kz_att_tower_f32<<<t.batch, AT_THREADS, lds, stream>>>(d);
}

}  // namespace kz
