// Micro-benchmark for the Go-size convolution's inner loop (kz_board_conv.hip): per K = 32 slice a wave multiplies a
// 96-row x 64-channel block, fetching 6 activation fragments from LDS (ds_read_b128) and 4 weight fragments from L2
// (global_load_dwordx4, a ring three slices deep).  Variant 16: 24 x v_mfma_f32_16x16x32_f16 (16 cycles each, 8 of them
// issue); variant 32: 12 x v_mfma_f32_32x32x16_f16 (32 cycles each, 8 of them issue) — same fragments, same FLOPs, three
// times the free issue cycles per fragment.  Prints cycles per slice (ideal 384) with one and with two workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SG_MFMA = 0x8, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100;

template <int V, int NVM, int NDS>
__global__ __launch_bounds__(256, 2) void k(const h16x8 *w, unsigned long long *out, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) reinterpret_cast<float *>(lds)[i] = 0.0f;
    __syncthreads();
    f32x4 a16[4][6];
    f32x16 a32[2][3];
    for (int o = 0; o < 4; o++) for (int n = 0; n < 6; n++) a16[o][n] = f32x4{0, 0, 0, 0};
    for (int o = 0; o < 2; o++) for (int n = 0; n < 3; n++) for (int e = 0; e < 16; e++) a32[o][n][e] = 0;
    h16x8 bA[6], bB[6], wring[3][4];
    int T[6];
    for (int n = 0; n < 6; n++) { T[n] = ((wave * 96 + n * 16 + (lane & 15)) * 80 + (lane >> 4) * 16) & 65535; bA[n] = bB[n] = *reinterpret_cast<const h16x8 *>(lds + T[n]); }
    const h16x8 *wp = w + lane;
    for (int u = 0; u < 3; u++) { for (int o = 0; o < 4; o++) wring[u][o] = wp[o * 64]; wp += 256; }
    unsigned long long t0, t1;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int it = 0; it < iters; it += 6) {
#pragma unroll
        for (int h = 0; h < 6; h++) {
            h16x8(&cur)[6] = (h & 1) ? bB : bA;
            h16x8(&nxt)[6] = (h & 1) ? bA : bB;
#pragma unroll
            for (int n = 0; n < NDS; n++) nxt[n] = *reinterpret_cast<const h16x8 *>(lds + T[n] + ((h + 1) & 3) * 5120);
            if (V == 16) {
#pragma unroll
                for (int o = 0; o < 4; o++)
#pragma unroll
                    for (int n = 0; n < 6; n++)
                        a16[o][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wring[h % 3][o], cur[n], a16[o][n], 0, 0, 0);
            } else {
                // the same 4 + 6 fragments as two K = 16 halves of 32-row / 32-channel tiles
#pragma unroll
                for (int half = 0; half < 2; half++)
#pragma unroll
                    for (int o = 0; o < 2; o++)
#pragma unroll
                        for (int n = 0; n < 3; n++)
                            a32[o][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wring[h % 3][half * 2 + o], cur[half * 3 + n], a32[o][n], 0, 0, 0);
            }
#pragma unroll
            for (int o = 0; o < NVM; o++) wring[h % 3][o] = wp[o * 64];
            wp += 256;
            constexpr int NM = V == 16 ? 24 : 12, PER = V == 16 ? 2 : 1;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, PER, 0);
                if (i < NDS) __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, PER, 0);
                if (i < NVM) __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - 10 * PER, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    float s = 0;
    for (int o = 0; o < 4; o++) for (int n = 0; n < 6; n++) s += a16[o][n][0] + a16[o][n][3];
    for (int o = 0; o < 2; o++) for (int n = 0; n < 3; n++) for (int e = 0; e < 16; e++) s += a32[o][n][e];
    if (s == 123.456f) sink[tid] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int V, int NVM, int NDS>
void run(const char *name, int wgs_per_cu, const h16x8 *w, unsigned long long *out, float *sink) {
    const int iters = 3000, lds_bytes = wgs_per_cu == 1 ? 131072 : 65536, grid = 256 * wgs_per_cu;
    (void)hipFuncSetAttribute((const void *)k<V, NVM, NDS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0, 0);
        k<V, NVM, NDS><<<grid, 256, lds_bytes>>>(w, out, iters, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    // wall-clock rate: every wave does iters slices of 24 x 16384 (or 12 x 32768) FLOP
    const double tflops = (double)grid * 4 * iters * 24 * 16384 / (ms * 1e-3) / 1e12;
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double per = sum / h.size() / iters;
    printf("%-44s %d WG/CU: %7.1f s_memtime ticks/slice per wave; wall clock %.3f ms = %.0f TFLOP/s (%4.1f %% of 2500)\n", name,
           wgs_per_cu, per, ms, tflops, tflops / 25.0);
}

int main() {
    h16x8 *w; unsigned long long *out; float *sink;
    (void)hipMalloc((void **)&w, (size_t)(3100 * 256 + 4096) * 16);
    (void)hipMemset(w, 0, (size_t)(3100 * 256 + 4096) * 16);
    (void)hipMalloc((void **)&out, 2048 * 8);
    (void)hipMalloc((void **)&sink, 1024 * 4);
    for (int wg = 1; wg <= 2; wg++) {
        run<16, 0, 0>("16x16x32: 24 MFMA", wg, w, out, sink);
        run<16, 0, 6>("16x16x32: 24 MFMA + 6 ds_read", wg, w, out, sink);
        run<16, 4, 0>("16x16x32: 24 MFMA + 4 global_load", wg, w, out, sink);
        run<16, 4, 6>("16x16x32: 24 MFMA + 6 ds_read + 4 gload", wg, w, out, sink);
        run<32, 0, 0>("32x32x16: 12 MFMA", wg, w, out, sink);
        run<32, 0, 6>("32x32x16: 12 MFMA + 6 ds_read", wg, w, out, sink);
        run<32, 4, 0>("32x32x16: 12 MFMA + 4 global_load", wg, w, out, sink);
        run<32, 4, 6>("32x32x16: 12 MFMA + 6 ds_read + 4 gload", wg, w, out, sink);
    }
    return 0;
}
