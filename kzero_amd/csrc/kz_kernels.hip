// kz_kernels.hip — gfx950 kernels of the generic (any board size, any channel count) path.
//
//   kz_encode_*        F0  board encode, HBM-bound
//   kz_conv_igemm      F1-F3 implicit-GEMM 3x3 / 1x1 convolution on MFMA with the fused bias/ReLU/residual/BN-tail
//                      epilogue; f32 uses v_mfma_f32_16x16x4_f32 (exact f32), f16 uses v_mfma_f32_16x16x32_f16
//   kz_scalar_head     F5  ScalarHead
//   kz_policy_*        F6  policy head tails
//
// The 8x8/256-channel f16 tower has its own board-resident kernel in kz_tower.hip; this file is the path for
// everything else (f32 parity path, Ataxx, Go 19x19, 1x1 head convolutions, dense head GEMMs).
#include "kz_kernels.hpp"

namespace kz {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int EPL = 4;  // elements per 16-byte chunk
};
template <>
struct Elem<h16> {
    static constexpr int EPL = 8;
};

// ---------------------------------------------------------------------------------------------------------
// F0: encode.  One thread per (board, square, 8-channel group): 8 contiguous channels of one NHWC row.
// Channel order = NCHW channel order of encode_input_full: scalar planes first, then bool planes
// (rust/kz-core/src/mapping/mod.rs:54-59); bool i of a board = bit i%8 of byte i/8 (bit_buffer.rs:73-75).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void store8(T *dst, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<float>(float *dst, const float (&v)[8]) {
    reinterpret_cast<f32x4 *>(dst)[0] = f32x4{v[0], v[1], v[2], v[3]};
    reinterpret_cast<f32x4 *>(dst)[1] = f32x4{v[4], v[5], v[6], v[7]};
}
template <>
__device__ __forceinline__ void store8<h16>(h16 *dst, const float (&v)[8]) {
    h16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (h16)v[j];
    *reinterpret_cast<h16x8 *>(dst) = o;
}

template <typename T>
__global__ void kz_encode_packed(const uint8_t *__restrict__ bits, size_t bits_stride,
                                 const float *__restrict__ scalars, int batch, int n_scalar, int n_bool, int hw,
                                 T *__restrict__ x, int ldx) {
    const int groups = ldx / 8;
    const long total = (long)batch * hw * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups);
        const long bp = idx / groups;
        const int p = (int)(bp % hw);
        const int b = (int)(bp / hw);
        const uint8_t *bb = bits + (size_t)b * bits_stride;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = g * 8 + j;
            float f = 0.0f;
            if (c < n_scalar) {
                f = scalars[(size_t)b * n_scalar + c];
            } else if (c < n_scalar + n_bool) {
                const unsigned bit = (unsigned)(c - n_scalar) * hw + p;
                f = (float)((bb[bit >> 3] >> (bit & 7)) & 1);
            }
            v[j] = f;
        }
        store8<T>(x + bp * ldx + g * 8, v);
    }
}

template <typename T>
__global__ void kz_encode_dense(const float *__restrict__ nchw, int batch, int c, int hw, T *__restrict__ x, int ldx) {
    const int groups = ldx / 8;
    const long total = (long)batch * hw * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        // pixel fastest so that the strided NCHW reads of a wave are contiguous per channel
        const int p = (int)(idx % hw);
        const long r = idx / hw;
        const int g = (int)(r % groups);
        const int b = (int)(r / groups);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int ch = g * 8 + j;
            v[j] = ch < c ? nchw[((size_t)b * c + ch) * hw + p] : 0.0f;
        }
        store8<T>(x + ((long)b * hw + p) * ldx + g * 8, v);
    }
}

static int grid_for(long total, int block) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

void launch_encode_packed(int dtype, const uint8_t *bits, size_t bits_stride, const float *scalars, int batch,
                          int n_scalar, int n_bool, int hw, void *x, int ldx, hipStream_t stream) {
    long total = (long)batch * hw * (ldx / 8);
    if (dtype == 0)
        kz_encode_packed<float><<<grid_for(total, 256), 256, 0, stream>>>(bits, bits_stride, scalars, batch, n_scalar,
                                                                         n_bool, hw, (float *)x, ldx);
    else
        kz_encode_packed<h16><<<grid_for(total, 256), 256, 0, stream>>>(bits, bits_stride, scalars, batch, n_scalar,
                                                                       n_bool, hw, (h16 *)x, ldx);
}

void launch_encode_dense(int dtype, const float *nchw, int batch, int c, int hw, void *x, int ldx,
                         hipStream_t stream) {
    long total = (long)batch * hw * (ldx / 8);
    if (dtype == 0)
        kz_encode_dense<float><<<grid_for(total, 256), 256, 0, stream>>>(nchw, batch, c, hw, (float *)x, ldx);
    else
        kz_encode_dense<h16><<<grid_for(total, 256), 256, 0, stream>>>(nchw, batch, c, hw, (h16 *)x, ldx);
}

// f32 [rows][c] -> [rows][c / 32][hi 32 | lo 32] f16, hi = f16(x), lo = f16(x - hi): the (hi, lo) tensors of the per-layer split
// convolution (kz_board_conv.hip).  HBM-bound: 4 B read + 4 B written per element; a thread takes 4 consecutive channels.
__global__ void kz_split_rows(const float *__restrict__ x, h16 *__restrict__ y, long rows, int c) {
    const long total = rows * (c / 4);
    for (long id = blockIdx.x * (long)blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        const long r = id / (c / 4);
        const int g = (int)(id - r * (c / 4));
        const float4 v = *reinterpret_cast<const float4 *>(x + r * c + g * 4);
        const float f[4] = {v.x, v.y, v.z, v.w};
        h16 hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            hi[j] = (h16)f[j];
            lo[j] = (h16)(f[j] - (float)hi[j]);
        }
        h16 *dst = y + r * 2 * c + ((g * 4) >> 5) * 64 + ((g * 4) & 31);  // group of 32 channels: [hi 32 | lo 32]
        *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(hi);
        *reinterpret_cast<uint2 *>(dst + 32) = *reinterpret_cast<const uint2 *>(lo);
    }
}

void launch_split_rows(const float *x, void *y, size_t rows, int c, hipStream_t stream) {
    kz_split_rows<<<grid_for((long)rows * (c / 4), 256), 256, 0, stream>>>(x, (h16 *)y, (long)rows, c);
}

// ---------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution.
//
// GEMM view: D[oc][row] = sum_{tap, c} W[tap][oc][c] * X[row shifted by tap][c]; the weights are the MFMA A operand
// and the activations the B operand, so each lane ends up with 4 CONSECUTIVE output channels of one row and the
// epilogue stores 8 (f16) / 16 (f32) contiguous bytes of the NHWC output.  im2col happens only in addressing: a tap
// outside the board contributes a zero row.
//
// Tile: TM rows x TN output channels per 256-thread workgroup, K consumed in chunks of 32 channels of one tap.
// Staging: global -> registers (next chunk, issued before the MFMAs of the current one) -> LDS -> fragments.
// Fragment order: lane (r = lane&15, kq = lane>>4) reads elements [8*kq, 8*kq+8) of tile row r for both operands;
//   f16: one v_mfma_f32_16x16x32_f16 per (oc-tile, row-tile) (that IS the instruction's k order: k = 8*kq + j);
//   f32: eight v_mfma_f32_16x16x4_f32, MFMA i pairing element i of both operands (its k index kq <-> channel
//        8*kq + i; every channel of the chunk is used exactly once, the order of a sum is free).
// ---------------------------------------------------------------------------------------------------------
struct ConvDev {
    const void *x, *w, *res;
    const float *bias, *post_scale, *post_shift;
    void *y;
    float *y32;
    int ldx, ldres, ldy, ldy32;
    int M, h, w_, group, src_group, src_off;
    int cin_p, cout_p, cout, k, relu;
};

template <typename T>
struct Frag;
template <>
struct Frag<h16> {
    h16x8 v;
    __device__ __forceinline__ void load(const h16 *p) { v = *reinterpret_cast<const h16x8 *>(p); }
};
template <>
struct Frag<float> {
    f32x4 lo, hi;
    __device__ __forceinline__ void load(const float *p) {
        lo = reinterpret_cast<const f32x4 *>(p)[0];
        hi = reinterpret_cast<const f32x4 *>(p)[1];
    }
};

__device__ __forceinline__ f32x4 mma(const Frag<h16> &a, const Frag<h16> &b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const Frag<float> &a, const Frag<float> &b, f32x4 c) {
#pragma unroll
    for (int i = 0; i < 4; i++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[i], b.lo[i], c, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[i], b.hi[i], c, 0, 0, 0);
    return c;
}

template <typename T>
__device__ __forceinline__ void load4(const T *p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float *p, float (&v)[4]) {
    f32x4 t = *reinterpret_cast<const f32x4 *>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <>
__device__ __forceinline__ void load4<h16>(const h16 *p, float (&v)[4]) {
    h16x4 t = *reinterpret_cast<const h16x4 *>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <typename T>
__device__ __forceinline__ void store4(T *p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(float *p, const float (&v)[4]) {
    *reinterpret_cast<f32x4 *>(p) = f32x4{v[0], v[1], v[2], v[3]};
}
template <>
__device__ __forceinline__ void store4<h16>(h16 *p, const float (&v)[4]) {
    *reinterpret_cast<h16x4 *>(p) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
}

template <typename T, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void kz_conv_igemm(ConvDev a) {
    constexpr int KC = 32;
    constexpr int EPL = Elem<T>::EPL;
    constexpr int CPR = KC / EPL;      // 16-byte chunks per tile row
    constexpr int KCP = KC + EPL;      // LDS row stride in elements (+16 B pad)
    constexpr int NX = (TM * CPR + 255) / 256;
    constexpr int NW = (TN * CPR + 255) / 256;
    constexpr int TMW = TM / WM, TNW = TN / WN;
    constexpr int MT = TMW / 16, NT = TNW / 16;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(TMW % 16 == 0 && TNW % 16 == 0, "tile");

    __shared__ __attribute__((aligned(16))) T Xs[TM * KCP];
    __shared__ __attribute__((aligned(16))) T Ws[TN * KCP];

    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    const T *__restrict__ xg = static_cast<const T *>(a.x);
    const T *__restrict__ wg = static_cast<const T *>(a.w);

    // per-thread staging assignments (fixed for the whole K loop)
    int x_off[NX], x_yx[NX];
    bool x_ok[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int id = tid + i * 256;
        const int row = id / CPR, cc = id % CPR;
        const int m = m0 + row;
        const bool ok = (id < TM * CPR) && (m < a.M);
        const int g = ok ? m / a.group : 0, r = ok ? m % a.group : 0;
        x_ok[i] = ok;
        x_yx[i] = ((r / a.w_) << 16) | (r % a.w_);
        x_off[i] = (g * a.src_group + a.src_off + r) * a.ldx + cc * EPL;
    }
    int w_off[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const int id = tid + i * 256;
        const int row = id / CPR, cc = id % CPR;
        w_off[i] = (n0 + row) * a.cin_p + cc * EPL;
    }

    const int nchunks = a.cin_p / KC;
    const int iters = a.k * a.k * nchunks;
    uint4 xr[NX], wr[NW];

    auto load_global = [&](int it) {
        const int tap = it / nchunks, c0 = (it - tap * nchunks) * KC;
        const int dy = a.k == 3 ? tap / 3 - 1 : 0, dx = a.k == 3 ? tap % 3 - 1 : 0;
#pragma unroll
        for (int i = 0; i < NX; i++) {
            const int yy = (x_yx[i] >> 16) + dy, xx = (x_yx[i] & 0xffff) + dx;
            const bool inb = x_ok[i] && (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.w_;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (inb) v = *reinterpret_cast<const uint4 *>(xg + (x_off[i] + (dy * a.w_ + dx) * a.ldx + c0));
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NW; i++) {
            if (tid + i * 256 < TN * CPR)
                wr[i] = *reinterpret_cast<const uint4 *>(wg + ((size_t)tap * a.cout_p * a.cin_p + w_off[i] + c0));
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < NX; i++) {
            const int id = tid + i * 256;
            if (id < TM * CPR) *reinterpret_cast<uint4 *>(&Xs[(id / CPR) * KCP + (id % CPR) * EPL]) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int id = tid + i * 256;
            if (id < TN * CPR) *reinterpret_cast<uint4 *>(&Ws[(id / CPR) * KCP + (id % CPR) * EPL]) = wr[i];
        }
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, kq = lane >> 4;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_global(0);
    for (int it = 0; it < iters; it++) {
        __syncthreads();  // everyone is done reading the previous chunk
        store_lds();
        __syncthreads();
        if (it + 1 < iters) load_global(it + 1);  // in flight during the MFMAs below

        Frag<T> bf[MT], af[NT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bf[mt].load(&Xs[(wm * TMW + mt * 16 + fr) * KCP + kq * 8]);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) af[nt].load(&Ws[(wn * TNW + nt * 16 + fr) * KCP + kq * 8]);
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[nt][mt] = mma(af[nt], bf[mt], acc[nt][mt]);
    }

    // epilogue: lane holds oc = base + 4*kq + {0..3} for row fr of each 16x16 tile
    const T *__restrict__ resg = static_cast<const T *>(a.res);
    T *__restrict__ yg = static_cast<T *>(a.y);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int oc = n0 + wn * TNW + nt * 16 + kq * 4;
        float bias[4], ps[4], pt[4];
        load4<float>(a.bias + oc, bias);
        if (a.post_scale) {
            load4<float>(a.post_scale + oc, ps);
            load4<float>(a.post_shift + oc, pt);
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int m = m0 + wm * TMW + mt * 16 + fr;
            if (m >= a.M) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[j] = acc[nt][mt][j] + bias[j];
                if (a.relu) v[j] = fmaxf(v[j], 0.0f);
            }
            if (resg) {
                float rv[4];
                load4<T>(resg + (size_t)m * a.ldres + oc, rv);
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] += rv[j];
            }
            if (a.post_scale) {
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = v[j] * ps[j] + pt[j];
            }
            if (yg) store4<T>(yg + (size_t)m * a.ldy + oc, v);
            if (a.y32) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (oc + j < a.cout) a.y32[(size_t)m * a.ldy32 + oc + j] = v[j];
            }
        }
    }
}

template <typename T>
static void launch_conv_t(const ConvArgs &a, hipStream_t stream) {
    ConvDev d;
    d.x = a.x; d.w = a.w; d.res = a.res; d.bias = a.bias; d.post_scale = a.post_scale; d.post_shift = a.post_shift;
    d.y = a.y; d.y32 = a.y32; d.ldx = a.ldx; d.ldres = a.ldres; d.ldy = a.ldy; d.ldy32 = a.ldy32;
    d.M = a.M; d.h = a.h; d.w_ = a.w_; d.group = a.group; d.src_group = a.src_group; d.src_off = a.src_off;
    d.cin_p = a.cin_p; d.cout_p = a.cout_p; d.cout = a.cout; d.k = a.k; d.relu = a.relu;
    if (a.cout_p % 128 == 0 && a.M >= 128 * 128) {
        dim3 grid((a.M + 127) / 128, a.cout_p / 128);
        kz_conv_igemm<T, 128, 128, 2, 2><<<grid, 256, 0, stream>>>(d);
    } else if (a.cout_p % 64 == 0) {
        dim3 grid((a.M + 63) / 64, a.cout_p / 64);
        kz_conv_igemm<T, 64, 64, 2, 2><<<grid, 256, 0, stream>>>(d);
    } else {
        dim3 grid((a.M + 63) / 64, a.cout_p / 32);
        kz_conv_igemm<T, 64, 32, 4, 1><<<grid, 256, 0, stream>>>(d);
    }
}

void launch_conv(int dtype, const ConvArgs &a, hipStream_t stream) {
    if (dtype == 0) launch_conv_t<float>(a, stream);
    else launch_conv_t<h16>(a, stream);
}

// grid size launch_conv picks for a 3x3 tower convolution of M rows and cout_p channels (same dispatch as above)
int conv_workgroups(int dtype, int M, int cout_p) {
    (void)dtype;
    if (cout_p % 128 == 0 && M >= 128 * 128) return ((M + 127) / 128) * (cout_p / 128);
    if (cout_p % 64 == 0) return ((M + 63) / 64) * (cout_p / 64);
    return ((M + 63) / 64) * (cout_p / 32);
}

const char *conv_kernel_name(int dtype) { return dtype == 0 ? "kz_conv_igemm_f32" : "kz_conv_igemm_f16"; }

// ---------------------------------------------------------------------------------------------------------
// Heads (small, VALU, f32 math).  One workgroup per board.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename T>
__device__ __forceinline__ float dot_row(const T *__restrict__ x, const float *__restrict__ w, int c) {
    // c is a multiple of 4 whenever x is a padded NHWC row; the tail loop covers anything else
    float acc = 0.0f;
    int i = 0;
    for (; i + 4 <= c; i += 4) {
        float xv[4];
        load4<T>(x + i, xv);
        acc += xv[0] * w[i] + xv[1] * w[i + 1] + xv[2] * w[i + 2] + xv[3] * w[i + 3];
    }
    for (; i < c; i++) acc += (float)x[i] * w[i];
    return acc;
}

// NF 1x1-conv filters over all pixel rows of one board, coalesced: a wave-instruction reads 64/G whole pixel rows
// (G = 16-byte pieces per row, a power of two <= 64; lane = (row, piece)), every lane multiplies its piece with its
// slice of the filters and the G lanes of a row add up by butterfly; U row groups in flight per wave.  emit(p, sums) runs
// on one lane per pixel row.  (One thread walking a 512-byte row on its own — dot_row per (filter, pixel) — costs 10x
// the time at Go size: each lane of a wave then reads a different row.)  Call with all 256 threads of the workgroup.
template <typename T>
__device__ __forceinline__ bool rows_coalescable(int ld) {
    const int G = ld / Elem<T>::EPL;
    return G <= 64 && (G & (G - 1)) == 0;
}
template <typename T, int NF, typename Emit>
__device__ __forceinline__ void rows_dot(const T *__restrict__ xb, int ld, int hw, int c, const float *__restrict__ w,
                                         Emit emit) {
    constexpr int EPL = Elem<T>::EPL, U = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = ld / EPL, R = 64 / G, r = lane / G, piece = lane % G;
    float wf[NF][EPL];
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            const int ch = piece * EPL + j;
            wf[f][j] = ch < c ? w[(size_t)f * c + ch] : 0.0f;
        }
    for (int p0 = wave * R; p0 < hw; p0 += 4 * R * U) {
        float xv[U][EPL];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int p = p0 + u * 4 * R + r;
            const T *src = xb + (size_t)(p < hw ? p : 0) * ld + piece * EPL;
#pragma unroll
            for (int h = 0; h < EPL / 4; h++) {
                float t[4];
                load4<T>(src + h * 4, t);
#pragma unroll
                for (int j = 0; j < 4; j++) xv[u][h * 4 + j] = t[j];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int p = p0 + u * 4 * R + r;
            float sum[NF];
#pragma unroll
            for (int f = 0; f < NF; f++) {
                float t = 0.0f;
#pragma unroll
                for (int j = 0; j < EPL; j++) t += xv[u][j] * wf[f][j];
                sum[f] = t;
            }
            for (int off = G >> 1; off > 0; off >>= 1)
#pragma unroll
                for (int f = 0; f < NF; f++) sum[f] += __shfl_xor(sum[f], off, 64);
            if (piece == 0 && p < hw) emit(p, sum);
        }
    }
}

struct ScalarHeadDev {
    const void *x;
    int ldx, batch, hw, c, hc, hs;
    const float *w0, *b0, *w1, *b1, *w2, *b2;
    float *out;
    int *nonfinite_flag;
    int epoch;
    const float *w1t;
    int extra;
    const float *w0x, *pe_bc, *pe_wl, *pe_bl;
    float *policy;
    int policy_len, policy_offset;
    int n_out, out_ld;
};

template <typename T>
__global__ __launch_bounds__(256) void kz_scalar_head(ScalarHeadDev a) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float *act = sh;                 // [hc*hw], channel-major like nn.Flatten on NCHW (post_act.py:16)
    float *hid = sh + a.hc * a.hw;   // [hs]
    float *sext = hid + a.hs;        // [hw]: the extra-move plane, when this launch carries ConvPolicyHead.seq_extra
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const T *xb = static_cast<const T *>(a.x) + (size_t)b * a.hw * a.ldx;

    bool bad = false;  // a non-finite sum = a non-finite value somewhere in this board's tower output
    if (a.extra && a.hc == 4 && rows_coalescable<T>(a.ldx)) {
        // the scalar head's four filters and the extra-move filter in ONE pass over the tower output
        const float bias[5] = {a.b0[0], a.b0[1], a.b0[2], a.b0[3], a.pe_bc[0]};
        rows_dot<T, 5>(xb, a.ldx, a.hw, a.c, a.w0x, [&](int p, const float(&sum)[5]) {
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                bad |= !(fabsf(sum[ch]) <= 3.0e38f);
                act[ch * a.hw + p] = fmaxf(sum[ch] + bias[ch], 0.0f);
            }
            sext[p] = sum[4] + bias[4];
        });
    } else if (a.hc == 4 && rows_coalescable<T>(a.ldx)) {
        const float bias[4] = {a.b0[0], a.b0[1], a.b0[2], a.b0[3]};
        rows_dot<T, 4>(xb, a.ldx, a.hw, a.c, a.w0, [&](int p, const float(&sum)[4]) {
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                bad |= !(fabsf(sum[ch]) <= 3.0e38f);
                act[ch * a.hw + p] = fmaxf(sum[ch] + bias[ch], 0.0f);
            }
        });
    } else {
        for (int o = tid; o < a.hc * a.hw; o += 256) {
            const int ch = o / a.hw, p = o % a.hw;
            const float v = dot_row<T>(xb + (size_t)p * a.ldx, a.w0 + (size_t)ch * a.c, a.c) + a.b0[ch];
            bad |= !(fabsf(v) <= 3.0e38f);
            act[o] = fmaxf(v, 0.0f);
        }
    }
    if (bad && a.nonfinite_flag) *reinterpret_cast<volatile int *>(a.nonfinite_flag) = a.epoch;  // (plain store: the flag may live in pinned host memory)
    __syncthreads();
    const int n_in = a.hc * a.hw;
    if (a.w1t && a.hs == 32) {
        // Linear(n_in -> 32): every thread takes inputs tid, tid + 256, ... and all 32 outputs (the 32 weights of an input
        // are 128 contiguous bytes of the transposed matrix: independent coalesced loads, no dependent chain), then the
        // partial sums meet by wave butterfly and through LDS.  (One wave per output walking its 1444-long row — the
        // loop below — is a chain of dependent L2 round trips: 589 us per call on Go 19x19 at batch 512.)
        float part[32];
#pragma unroll
        for (int j = 0; j < 32; j++) part[j] = 0.0f;
        for (int i = tid; i < n_in; i += 256) {
            const float x = act[i];
            const f32x4 *w = reinterpret_cast<const f32x4 *>(a.w1t + (size_t)i * 32);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x4 wv = w[q];
#pragma unroll
                for (int e = 0; e < 4; e++) part[q * 4 + e] += wv[e] * x;
            }
        }
        __shared__ float red[4][32];
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const float s = wave_sum(part[j]);
            if (lane == 0) red[wave][j] = s;
        }
        __syncthreads();
        if (tid < 32) hid[tid] = fmaxf(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + a.b1[tid], 0.0f);
    } else {
        for (int j = wave; j < a.hs; j += 4) {
            float acc = 0.0f;
            for (int i = lane; i < n_in; i += 64) acc += a.w1[(size_t)j * n_in + i] * act[i];
            acc = wave_sum(acc);
            if (lane == 0) hid[j] = fmaxf(acc + a.b1[j], 0.0f);
        }
    }
    __syncthreads();
    for (int j = wave; j < a.n_out; j += 4) {
        float acc = 0.0f;
        for (int i = lane; i < a.hs; i += 64) acc += a.w2[j * a.hs + i] * hid[i];
        acc = wave_sum(acc);
        if (lane == 0) a.out[(size_t)b * a.out_ld + j] = acc + a.b2[j];
    }
    if (a.extra) {  // Linear(hw -> extra) behind the policy planes (sext is complete since the first barrier)
        for (int j = wave; j < a.extra; j += 4) {
            float acc = 0.0f;
            for (int p = lane; p < a.hw; p += 64) acc += a.pe_wl[(size_t)j * a.hw + p] * sext[p];
            acc = wave_sum(acc);
            if (lane == 0) a.policy[(size_t)b * a.policy_len + a.policy_offset + j] = acc + a.pe_bl[j];
        }
    }
}

// whether kz_scalar_head can carry ConvPolicyHead's extra-move head in its pass over the tower output: four scalar-head
// filters and whole 16-byte pieces per row in a power-of-two count (rows_dot)
bool scalar_head_takes_extra(int dtype, int ldx, int hc) {
    const int epl = dtype == 0 ? 4 : 8, g = ldx / epl;
    return hc == 4 && ldx % epl == 0 && g >= 1 && g <= 64 && (g & (g - 1)) == 0;
}

void launch_scalar_head(int dtype, const ScalarHeadArgs &a, hipStream_t stream) {
    const bool extra = a.extra > 0 && a.w0x && scalar_head_takes_extra(dtype, a.ldx, a.hc);
    ScalarHeadDev d{a.x, a.ldx, a.batch, a.hw, a.c, a.hc, a.hs, a.w0, a.b0, a.w1, a.b1, a.w2, a.b2, a.out,
                    a.nonfinite_flag, a.epoch, a.w1t, extra ? a.extra : 0, a.w0x, a.pe_bc, a.pe_wl, a.pe_bl, a.policy,
                    a.policy_len, a.policy_offset, a.n_out, a.out_ld};
    size_t shmem = sizeof(float) * ((size_t)a.hc * a.hw + a.hs + (extra ? a.hw : 0));
    if (dtype == 0) kz_scalar_head<float><<<a.batch, 256, shmem, stream>>>(d);
    else kz_scalar_head<h16><<<a.batch, 256, shmem, stream>>>(d);
}

struct PolicyConvDev {
    const void *y;
    int ldy, batch, hw, c, pc;
    const float *w, *b;
    float *policy;
    int policy_len, zero_tail;
};

template <typename T>
__global__ __launch_bounds__(256) void kz_policy_conv(PolicyConvDev a) {
    const int b = blockIdx.x;
    const T *yb = static_cast<const T *>(a.y) + (size_t)b * a.hw * a.ldy;
    float *pol = a.policy + (size_t)b * a.policy_len;
    const int n_out = a.pc * a.hw;
    if (a.pc == 1 && gridDim.y == 1 && rows_coalescable<T>(a.ldy)) {  // one filter (the Go conv head): coalesced rows
        const float bias = a.b[0];
        rows_dot<T, 1>(yb, a.ldy, a.hw, a.c, a.w, [&](int p, const float(&sum)[1]) { pol[p] = sum[0] + bias; });
        for (int o = n_out + threadIdx.x; o < n_out + a.zero_tail; o += 256) pol[o] = 0.0f;
        return;
    }
    for (int o = blockIdx.y * 256 + threadIdx.x; o < n_out + a.zero_tail; o += gridDim.y * 256) {
        if (o >= n_out) {
            pol[o] = 0.0f;  // AtaxxConvPolicyHead: the pass logit is a constant zero column (post_act.py:106-110)
            continue;
        }
        const int oc = o / a.hw, p = o % a.hw;  // channel-major flatten (post_act.py:83,108)
        pol[o] = dot_row<T>(yb + (size_t)p * a.ldy, a.w + (size_t)oc * a.c, a.c) + a.b[oc];
    }
}

void launch_policy_conv(int dtype, const PolicyConvArgs &a, hipStream_t stream) {
    PolicyConvDev d{a.y, a.ldy, a.batch, a.hw, a.c, a.pc, a.w, a.b, a.policy, a.policy_len, a.zero_tail};
    int gy = (a.pc * a.hw + a.zero_tail + 255) / 256;
    if (gy > 8) gy = 8;
    if (a.pc == 1) gy = 1;  // one workgroup per board walks the rows together
    dim3 grid(a.batch, gy);
    if (dtype == 0) kz_policy_conv<float><<<grid, 256, 0, stream>>>(d);
    else kz_policy_conv<h16><<<grid, 256, 0, stream>>>(d);
}

struct PolicyExtraDev {
    const void *x;
    int ldx, batch, hw, c, extra;
    const float *wc, *bc, *wl, *bl;
    float *policy;
    int policy_len, offset;
};

template <typename T>
__global__ __launch_bounds__(256) void kz_policy_extra(PolicyExtraDev a) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [hw]
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const T *xb = static_cast<const T *>(a.x) + (size_t)b * a.hw * a.ldx;
    if (rows_coalescable<T>(a.ldx)) {
        const float bias = a.bc[0];
        rows_dot<T, 1>(xb, a.ldx, a.hw, a.c, a.wc, [&](int p, const float(&sum)[1]) { sh[p] = sum[0] + bias; });
    } else {
        for (int p = tid; p < a.hw; p += 256) sh[p] = dot_row<T>(xb + (size_t)p * a.ldx, a.wc, a.c) + a.bc[0];
    }
    __syncthreads();
    for (int j = wave; j < a.extra; j += 4) {
        float acc = 0.0f;
        for (int p = lane; p < a.hw; p += 64) acc += a.wl[(size_t)j * a.hw + p] * sh[p];
        acc = wave_sum(acc);
        if (lane == 0) a.policy[(size_t)b * a.policy_len + a.offset + j] = acc + a.bl[j];
    }
}

void launch_policy_extra(int dtype, const PolicyExtraArgs &a, hipStream_t stream) {
    PolicyExtraDev d{a.x, a.ldx, a.batch, a.hw, a.c, a.extra, a.wc, a.bc, a.wl, a.bl, a.policy, a.policy_len, a.offset};
    size_t shmem = sizeof(float) * a.hw;
    if (dtype == 0) kz_policy_extra<float><<<a.batch, 256, shmem, stream>>>(d);
    else kz_policy_extra<h16><<<a.batch, 256, shmem, stream>>>(d);
}

struct AttentionDev {
    const void *bulk, *under;
    int ld_bulk, ld_under, batch, q;
    const int32_t *flat_to_att;
    float *policy;
    int policy_len;
};

// policy[b][k] = sum_q q_from[q][i] * q_to[q][j] / sqrt(Q), (i, j) = divmod(flat_to_att[k], 88)   (post_act.py:127-141)
//   q_from[q][i]      = bulk[square i][q]
//   q_to[q][j < 64]   = bulk[square j][Q + q]
//   q_to[q][64 + t]   = under.reshape(Q, 24)[q][t] = under[channel 3q + t/8][file t%8]
template <typename T>
__global__ __launch_bounds__(256) void kz_attention_gather(AttentionDev a) {
    const int b = blockIdx.x;
    const T *bulk = static_cast<const T *>(a.bulk) + (size_t)b * 64 * a.ld_bulk;
    const T *under = static_cast<const T *>(a.under) + (size_t)b * 8 * a.ld_under;
    const float inv = 1.0f / sqrtf((float)a.q);
    for (int k = blockIdx.y * 256 + threadIdx.x; k < a.policy_len; k += gridDim.y * 256) {
        const int idx = a.flat_to_att[k];
        const int i = idx / 88, j = idx % 88;
        const T *qf = bulk + (size_t)i * a.ld_bulk;
        float acc = 0.0f;
        if (j < 64 && (a.q & 3) == 0) {
            const T *qt = bulk + (size_t)j * a.ld_bulk + a.q;
            for (int q = 0; q < a.q; q += 4) {
                float f[4], t[4];
                load4<T>(qf + q, f);
                load4<T>(qt + q, t);
                acc += f[0] * t[0] + f[1] * t[1] + f[2] * t[2] + f[3] * t[3];
            }
        } else if (j < 64) {
            const T *qt = bulk + (size_t)j * a.ld_bulk + a.q;
            for (int q = 0; q < a.q; q++) acc += (float)qf[q] * (float)qt[q];
        } else {
            const int t = j - 64;
            const T *qt = under + (size_t)(t % 8) * a.ld_under + t / 8;
            for (int q = 0; q < a.q; q++) acc += (float)qf[q] * (float)qt[3 * q];
        }
        // the reference divides by sqrt(Q) (post_act.py:138); acc / s and acc * (1/s) differ by <= 1 ulp
        a.policy[(size_t)b * a.policy_len + k] = acc * inv;
    }
}

// The same on the matrix cores (Q a multiple of 32 in f16, of 16 in f32): one workgroup per board computes the WHOLE
// 64 x 88 logit matrix — wave w the from-squares 16w .. 16w + 15 against six tiles of 16 to-columns (64 squares, 24
// underpromotion columns, 8 columns of zero) — with its fragments read straight from the two tensors (a lane's piece is 16
// contiguous bytes of a row; only the underpromotion columns' stride-3 channels are gathered), parks the matrix in LDS and
// picks the policy_len moves from there.  The gather kernel above reads 2 Q values from the caches for every one of
// 1880 logits (0.96 MB per board at Q = 128): 48.7 us per batch of 256 next to a 20 x 128 tower of 163 us.
template <typename T>
__global__ __launch_bounds__(256) void kz_attention_mfma(AttentionDev a) {
    constexpr int EPL = Elem<T>::EPL;  // fragment = 16 bytes: 8 f16 = a quarter of one 16x16x32 step; 4 f32 = four 16x16x4 steps
    constexpr int LD = 96;
    __shared__ float logits[64 * LD];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, fr = lane & 15, kq = lane >> 4;
    const T *bulk = static_cast<const T *>(a.bulk) + (size_t)b * 64 * a.ld_bulk;
    const T *under = static_cast<const T *>(a.under) + (size_t)b * 8 * a.ld_under;
    const T *pa = bulk + (size_t)(16 * wave + fr) * a.ld_bulk + EPL * kq;  // q_from of square 16 wave + fr
    const T *pb = bulk + (size_t)fr * a.ld_bulk + a.q + EPL * kq;          // q_to of squares fr, 16 + fr, 32 + fr, 48 + fr
    // underpromotion columns t = fr and 16 + fr (t < 24): channel 3 q + t / 8 of file t % 8
    const bool u1 = fr < 8;
    const T *pu0 = under + (size_t)(fr & 7) * a.ld_under + (fr >> 3) + 3 * EPL * kq;
    const T *pu1 = under + (size_t)(fr & 7) * a.ld_under + 2 + 3 * EPL * kq;
    typedef T frag_t __attribute__((ext_vector_type(EPL)));
    f32x4 acc[6];
#pragma unroll
    for (int n = 0; n < 6; n++) acc[n] = f32x4{0, 0, 0, 0};
    const int steps = a.q / (4 * EPL);
    for (int ks = 0; ks < steps; ks++) {
        const int k0 = 4 * EPL * ks;
        const frag_t fa = *reinterpret_cast<const frag_t *>(pa + k0);
        frag_t fb[6];
#pragma unroll
        for (int n = 0; n < 4; n++) fb[n] = *reinterpret_cast<const frag_t *>(pb + (size_t)16 * n * a.ld_bulk + k0);
#pragma unroll
        for (int r = 0; r < EPL; r++) {
            fb[4][r] = pu0[3 * (k0 + r)];
            fb[5][r] = u1 ? pu1[3 * (k0 + r)] : (T)0;
        }
#pragma unroll
        for (int n = 0; n < 6; n++) {
            if constexpr (EPL == 8) {
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[n], acc[n], 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r], fb[n][r], acc[n], 0, 0, 0);
            }
        }
    }
    // the reference divides by sqrt(Q) (post_act.py:138); acc / s and acc * (1/s) differ by <= 1 ulp
    const float inv = 1.0f / sqrtf((float)a.q);
#pragma unroll
    for (int n = 0; n < 6; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) logits[(16 * wave + 4 * kq + r) * LD + 16 * n + fr] = acc[n][r] * inv;
    __syncthreads();
    for (int k = tid; k < a.policy_len; k += 256) {
        const int idx = a.flat_to_att[k];
        const int i = idx / 88, j = idx - 88 * i;
        a.policy[(size_t)b * a.policy_len + k] = logits[i * LD + j];
    }
}

// ---------------------------------------------------------------------------------------------------------
// F7: decode_output on the device (rust/kz-core/src/network/common.rs:16-100) as a launch of its own, for the paths whose
// heads are separate launches (the one-launch networks end in the same code: kz_decode_dev.hpp).  One WAVE per board, four
// boards per workgroup:
//   values = [tanh(s0), softmax(s1..s3), s4]  (:60-74)
//   probs  = softmax over the logits gathered at the board's available-move indices, in the given order (:77-86)
// error_flag[0] = 1: a softmax sum is not strictly positive (where the reference asserts, :110) or a move index is outside
// the policy; error_flag[1] = 1: *nonfinite_flag == epoch (the range check of ScalarHeadArgs).  Move lists, values,
// probabilities and the flag words may be pinned host memory: every word is read or written once, flags by plain stores.
// ---------------------------------------------------------------------------------------------------------
namespace {
#include "kz_decode_dev.hpp"
constexpr int DECODE_STAGE = 1024;  // floats of LDS staging per wave (a longer move list re-reads its tail)
}  // namespace

__global__ __launch_bounds__(256) void kz_decode_output(const float *__restrict__ scalars, const float *__restrict__ logits,
                                                        int batch, DecodeDev d, const int *__restrict__ nonfinite_flag, int epoch) {
    __shared__ float stage[4][DECODE_STAGE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nonfinite_flag && *nonfinite_flag == epoch)
        *reinterpret_cast<volatile int *>(d.error_flag + 1) = 1;
    if (b >= batch) return;
    const float *lg = logits + (size_t)b * d.policy_len;
    decode_board_wave(d, b, lane, scalars + (size_t)b * 5, stage[wave], DECODE_STAGE, [&](int idx) { return lg[idx]; });
}

void launch_decode_output(const float *scalars, const float *logits, int batch, int policy_len,
                          const int64_t *move_offsets, const int32_t *move_indices, float *values, float *probs,
                          int *error_flag, const int *nonfinite_flag, int epoch, hipStream_t stream) {
    const DecodeDev d{move_offsets, move_indices, values, probs, error_flag, policy_len};
    kz_decode_output<<<(batch + 3) / 4, 256, 0, stream>>>(scalars, logits, batch, d, nonfinite_flag, epoch);
}

void launch_attention(int dtype, const AttentionArgs &a, hipStream_t stream) {
    AttentionDev d{a.bulk, a.under, a.ld_bulk, a.ld_under, a.batch, a.q, a.flat_to_att, a.policy, a.policy_len};
    // (16-byte fragments: rows of both tensors start on 16 bytes when their strides are multiples of a fragment)
    const int epl = dtype == 0 ? 4 : 8;
    if (a.q % (4 * epl) == 0 && a.ld_bulk % epl == 0) {
        if (dtype == 0) kz_attention_mfma<float><<<a.batch, 256, 0, stream>>>(d);
        else kz_attention_mfma<h16><<<a.batch, 256, 0, stream>>>(d);
        return;
    }
    int gy = (a.policy_len + 255) / 256;
    dim3 grid(a.batch, gy);
    if (dtype == 0) kz_attention_gather<float><<<grid, 256, 0, stream>>>(d);
    else kz_attention_gather<h16><<<grid, 256, 0, stream>>>(d);
}

}  // namespace kz
