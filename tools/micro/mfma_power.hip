// Micro-benchmark: does it matter for the matrix cores' power draw (= the clock the chip sustains) WHICH
// MFMA operand carries the post-ReLU activations (half of them exactly zero) and which the weights?  One wave per SIMD,
// register-resident operands, 24 MFMAs per slice as in kz_tower_resident; several seconds per variant so that DVFS
// settles.  Prints wall-clock TFLOP/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: A = weights, B = activations; 1: A = activations, B = weights
template <int MODE, int HOLD>
__global__ __launch_bounds__(256, 1) void k(const h16x8 *wsrc, const h16x8 *xsrc, int iters, float *sink) {
    const int tid = threadIdx.x;
    h16x8 w[4], x[6];
    for (int i = 0; i < 4; i++) w[i] = wsrc[(blockIdx.x % 64 * 4 + i) * 256 + tid];
    for (int i = 0; i < 6; i++) x[i] = xsrc[(blockIdx.x % 64 * 6 + i) * 256 + tid];
    f32x4 acc[4][6];
    for (int o = 0; o < 4; o++) for (int n = 0; n < 6; n++) acc[o][n] = f32x4{0, 0, 0, 0};
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (HOLD == 0) {  // weight fragment held, activation fragment changes every MFMA
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int n = 0; n < 6; n++)
                    acc[o][n] = MODE == 0 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(w[o], x[n], acc[o][n], 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_16x16x32_f16(x[n], w[o], acc[o][n], 0, 0, 0);
        } else {  // activation fragment held
#pragma unroll
            for (int n = 0; n < 6; n++)
#pragma unroll
                for (int o = 0; o < 4; o++)
                    acc[o][n] = MODE == 0 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(w[o], x[n], acc[o][n], 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_16x16x32_f16(x[n], w[o], acc[o][n], 0, 0, 0);
        }
    }
    float s = 0;
    for (int o = 0; o < 4; o++) for (int n = 0; n < 6; n++) s += acc[o][n][0] + acc[o][n][1] + acc[o][n][2] + acc[o][n][3];
    if (s == 123.456f) sink[tid] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the same operands through v_mfma_f32_32x32x16_f16: twice the FLOP per fragment pair read from the register file
template <int HOLD>
__global__ __launch_bounds__(256, 1) void k32(const h16x8 *wsrc, const h16x8 *xsrc, int iters, float *sink) {
    const int tid = threadIdx.x;
    h16x8 w[4], x[3];
    for (int i = 0; i < 4; i++) w[i] = wsrc[(blockIdx.x % 64 * 4 + i) * 256 + tid];
    for (int i = 0; i < 3; i++) x[i] = xsrc[(blockIdx.x % 64 * 6 + i) * 256 + tid];
    f32x16 acc[4][3];
    for (int o = 0; o < 4; o++) for (int n = 0; n < 3; n++) for (int e = 0; e < 16; e++) acc[o][n][e] = 0;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (HOLD == 0) {
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int n = 0; n < 3; n++) acc[o][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[o], x[n], acc[o][n], 0, 0, 0);
        } else {
#pragma unroll
            for (int n = 0; n < 3; n++)
#pragma unroll
                for (int o = 0; o < 4; o++) acc[o][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[o], x[n], acc[o][n], 0, 0, 0);
        }
    }
    float s = 0;
    for (int o = 0; o < 4; o++) for (int n = 0; n < 3; n++) for (int e = 0; e < 16; e++) s += acc[o][n][e];
    if (s == 123.456f) sink[tid] = s;
}

template <int HOLD>
void run32(const char *name, const h16x8 *w, const h16x8 *x, float *sink) {
    const int iters = 400000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 12; rep++) {
        (void)hipEventRecord(e0, 0);
        k32<HOLD><<<256, 256>>>(w, x, iters, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 4 && ms < best) best = ms;
    }
    const double tflops = 256.0 * 4 * iters * 12 * 32768 / (best * 1e-3) / 1e12;
    printf("%-64s %8.2f ms  %6.0f TFLOP/s (%4.1f %% of 2500)\n", name, best, tflops, tflops / 25.0);
}

template <int MODE, int HOLD>
void run(const char *name, const h16x8 *w, const h16x8 *x, float *sink) {
    const int iters = 400000;  // ~64 ms per launch at the MFMA peak
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 12; rep++) {
        (void)hipEventRecord(e0, 0);
        k<MODE, HOLD><<<256, 256>>>(w, x, iters, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 4 && ms < best) best = ms;  // the clock has settled after the first launches
    }
    const double tflops = 256.0 * 4 * iters * 24 * 16384 / (best * 1e-3) / 1e12;
    printf("%-64s %8.2f ms  %6.0f TFLOP/s (%4.1f %% of 2500)\n", name, best, tflops, tflops / 25.0);
}

static float gauss() {
    float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = (rand() + 1.0f) / (RAND_MAX + 2.0f);
    return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v);
}

int main() {
    const size_t nw = 64 * 4 * 256 * 8, nx = 64 * 6 * 256 * 8;
    std::vector<h16> hw(nw), hx(nx), hz(nx, (h16)0.0f), hd(nx);
    srand(1);
    for (auto &v : hw) v = (h16)(0.03f * gauss());
    for (auto &v : hx) { float g = gauss(); v = (h16)(g > 0 ? g : 0.0f); }  // post-ReLU: half zeros
    for (auto &v : hd) v = (h16)gauss();
    h16x8 *w, *x, *z, *d; float *sink;
    (void)hipMalloc((void **)&w, nw * 2); (void)hipMalloc((void **)&x, nx * 2); (void)hipMalloc((void **)&z, nx * 2); (void)hipMalloc((void **)&d, nx * 2);
    (void)hipMalloc((void **)&sink, 4096);
    (void)hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(z, hz.data(), nx * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(d, hd.data(), nx * 2, hipMemcpyHostToDevice);
    run<0, 0>("A = weights (held), B = post-ReLU activations", w, x, sink);
    run<1, 0>("A = post-ReLU activations, B = weights (held)", w, x, sink);
    run<0, 1>("A = weights, B = post-ReLU activations (held)", w, x, sink);
    run<1, 1>("A = post-ReLU activations (held), B = weights", w, x, sink);
    run<0, 0>("A = weights (held), B = dense N(0,1) activations", w, d, sink);
    run<1, 0>("A = dense N(0,1) activations, B = weights (held)", w, d, sink);
    run32<0>("32x32x16: A = weights (held), B = post-ReLU activations", w, x, sink);
    run32<1>("32x32x16: A = weights, B = post-ReLU activations (held)", w, x, sink);
    run32<0>("32x32x16: A = weights (held), B = dense N(0,1)", w, d, sink);
    run<0, 0>("A = weights (held), B = zeros", w, z, sink);
    run<1, 0>("A = zeros, B = weights (held)", w, z, sink);
    return 0;
}
