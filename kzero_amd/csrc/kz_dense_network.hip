// kz_dense_network.hip — DenseNetwork with its DenseBlocks (python/lib/model/simple.py:7-52: the reference's own test networks,
// python/main/write_test_networks.py:14-18) in one launch: Flatten, Linear, `depth` DenseBlocks (BatchNorm1d, ReLU, Linear,
// BatchNorm1d, ReLU, Linear; x + y when res), BatchNorm1d, ReLU, Linear to 5 + policy_len; scalars = output[:5],
// policy = output[5:] (:29-33).  A workgroup per board, the vectors in LDS, every Linear a row per wave at a time with the
// inputs across the lanes (f32 FMA, coalesced weight rows).  These are networks of a few ten thousand parameters: the kernel
// is here so that every network the reference's Python can export runs, not for its rate.
#include "kz_kernels.hpp"

namespace kz {
namespace {

typedef _Float16 h16;

constexpr int DN_THREADS = 256, DN_WAVES = 4;

__device__ __forceinline__ float dn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// out[o] = b[o] + sum_i W[o][i] * in[i]
__device__ __forceinline__ void dn_linear(const float *__restrict__ W, const float *__restrict__ b, const float *in, int n_in, float *out,
                                          int n_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int o = wave; o < n_out; o += DN_WAVES) {
        const float *row = W + (size_t)o * n_in;
        float s = 0.0f;
        for (int i = lane; i < n_in; i += 64) s = fmaf(row[i], in[i], s);
        s = dn_wave_sum(s);
        if (lane == 0) out[o] = s + b[o];
    }
}

__global__ __launch_bounds__(DN_THREADS) void kz_dense_network(DenseNetArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n_in = a.hw * a.cin_p, size = a.size, tid = threadIdx.x, board = blockIdx.x;
    float *IN = lds, *cur = IN + n_in, *act = cur + size, *mid = act + size, *out = mid + size;
    // the board's encoded rows [hw][cin_p] as they lie (dn_in's columns are stored in that order)
    for (int i = tid; i < n_in; i += DN_THREADS) {
        const size_t src = (size_t)board * n_in + i;
        IN[i] = a.in_f16 ? (float)static_cast<const h16 *>(a.x0)[src] : static_cast<const float *>(a.x0)[src];
    }
    __syncthreads();
    dn_linear(a.w_in, a.b_in, IN, n_in, cur, size);
    __syncthreads();
    const float *blk = a.blocks;
    for (int l = 0; l < a.depth; l++) {
        const float *sa = blk, *ta = sa + size, *wa = ta + size, *ba = wa + (size_t)size * size, *sb = ba + size, *tb = sb + size,
                    *wb = tb + size, *bb = wb + (size_t)size * size;
        blk = bb + size;
        for (int i = tid; i < size; i += DN_THREADS) act[i] = fmaxf(fmaf(cur[i], sa[i], ta[i]), 0.0f);
        __syncthreads();
        dn_linear(wa, ba, act, size, mid, size);
        __syncthreads();
        for (int i = tid; i < size; i += DN_THREADS) mid[i] = fmaxf(fmaf(mid[i], sb[i], tb[i]), 0.0f);
        __syncthreads();
        dn_linear(wb, bb, mid, size, act, size);
        __syncthreads();
        for (int i = tid; i < size; i += DN_THREADS) cur[i] = a.res ? cur[i] + act[i] : act[i];
        __syncthreads();
    }
    for (int i = tid; i < size; i += DN_THREADS) act[i] = fmaxf(fmaf(cur[i], a.sf[i], a.tf[i]), 0.0f);
    __syncthreads();
    dn_linear(a.w_out, a.b_out, act, size, out, 5 + a.policy_len);
    __syncthreads();
    bool bad = false;
    for (int i = tid; i < 5 + a.policy_len; i += DN_THREADS) {
        const float v = out[i];
        bad |= !(fabsf(v) <= 3.0e38f);
        if (i < 5) a.scalars[(size_t)board * 5 + i] = v;
        else a.policy[(size_t)board * a.policy_len + (i - 5)] = v;
    }
    if (bad && a.nonfinite_flag) *reinterpret_cast<volatile int *>(a.nonfinite_flag) = a.epoch;  // (plain store: may be pinned host memory)
}

size_t dn_lds_bytes(int hw, int cin_p, int size, int policy_len) { return ((size_t)hw * cin_p + 3 * (size_t)size + 5 + policy_len) * 4; }

}  // namespace

bool dense_network_supported(int h, int w, int c_in, int size, int depth, int policy_len) {
    if (h < 1 || w < 1 || c_in < 1 || size < 1 || depth < 0 || policy_len < 1 || size > (1 << 16) || policy_len > (1 << 16)) return false;
    const int cin_p = (c_in + 31) / 32 * 32;
    return dn_lds_bytes(h * w, cin_p, size, policy_len) <= (size_t)160 * 1024;
}

size_t dense_network_block_elems(int size) { return (size_t)2 * size * size + 6 * (size_t)size; }

void launch_dense_network(const DenseNetArgs &a, hipStream_t stream) {
    if (a.batch <= 0) return;
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_dense_network, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done_mask |= 1ull << (dev & 63);
    }
    kz_dense_network<<<a.batch, DN_THREADS, dn_lds_bytes(a.hw, a.cin_p, a.size, a.policy_len), stream>>>(a);
}

}  // namespace kz
