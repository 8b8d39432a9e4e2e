// Micro-benchmark: what does an instruction that is not an MFMA cost a single wave per SIMD that is otherwise issuing
// back-to-back v_mfma_f32_16x16x4_f32 (32 cycles each)?  One workgroup of 4 waves per CU (128 KB of LDS), a loop of
// 56 MFMAs on 14 independent accumulators per iteration with, per variant, LDS fragment reads / global loads / VALU /
// waits interleaved the way kz_tower_resident_f32 does.  Prints cycles per iteration (ideal: 56 * 32 = 1792).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue tools/micro/mfma_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SG_MFMA = 0x8, SG_VALU = 0x2, SG_VMEM_READ = 0x20, SG_DS_READ = 0x100;

template <int NDS, int NVM, int NVALU, int DSW>
__global__ __launch_bounds__(256, 1) void k(const f32x4 *w, unsigned long long *out, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32768; i += 256) reinterpret_cast<float *>(lds)[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[2][7];
    for (int o = 0; o < 2; o++) for (int n = 0; n < 7; n++) acc[o][n] = f32x4{0, 0, 0, 0};
    f32x4 bA[7], bB[7], wring[4][2];
    int T[7];
    for (int n = 0; n < 7; n++) { T[n] = ((n * 16 + (lane & 15)) * 528 + (lane >> 4) * 128); bA[n] = bB[n] = *reinterpret_cast<const f32x4 *>(lds + T[n]); }
    const f32x4 *wp = w + wave * 128 + lane;
    for (int u = 0; u < 4; u++) { wring[u][0] = wp[0]; wring[u][1] = wp[64]; wp += 512; }
    int vdummy = lane;
    unsigned long long t0, t1;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int h = 0; h < 4; h++) {
            f32x4(&cur)[7] = (h & 1) ? bB : bA;
            f32x4(&nxt)[7] = (h & 1) ? bA : bB;
            if (NDS) __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int n = 0; n < NDS; n++) {
                if (DSW == 16) nxt[n] = *reinterpret_cast<const f32x4 *>(lds + T[n] + (h + 1) * 16);
                else {
                    const float2 a = *reinterpret_cast<const float2 *>(lds + T[n] + (h + 1) * 16);
                    nxt[n][0] = a.x; nxt[n][1] = a.y;
                }
            }
#pragma unroll
            for (int v = 0; v < NVALU; v++) vdummy = vdummy * 3 + v;
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int o = 0; o < 2; o++)
#pragma unroll
                    for (int n = 0; n < 7; n++)
                        acc[o][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wring[h][o][s], cur[n][s], acc[o][n], 0, 0, 0);
            if (NVM) { wring[h][0] = wp[0]; if (NVM > 1) wring[h][1] = wp[64]; wp += 512; }
#pragma unroll
            for (int i = 0; i < 7; i++) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 4, 0);
                if (i < NDS) __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 1, 0);
                if (i < NVALU) __builtin_amdgcn_sched_group_barrier(SG_VALU, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 20, 0);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 4, 0);
                if (i < NVM) __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    f32x4 s = f32x4{0, 0, 0, 0};
    for (int o = 0; o < 2; o++) for (int n = 0; n < 7; n++) s += acc[o][n];
    if (s[0] == 123.456f || vdummy == -12345) sink[tid] = s[1] + s[2] + s[3];
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NDS, int NVM, int NVALU, int DSW>
void run(const char *name, const f32x4 *w, unsigned long long *out, float *sink) {
    const int iters = 2000, lds_bytes = 131072;
    (void)hipFuncSetAttribute((const void *)k<NDS, NVM, NVALU, DSW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    for (int rep = 0; rep < 2; rep++) {
        k<NDS, NVM, NVALU, DSW><<<256, 256, lds_bytes>>>(w, out, iters, sink);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    printf("%-46s %8.1f cycles/iteration (ideal 1792)\n", name, sum / 1024 / iters);
}

int main() {
    f32x4 *w; unsigned long long *out; float *sink;
    (void)hipMalloc((void **)&w, (size_t)(2100 * 512 + 4096) * 16);
    (void)hipMemset(w, 0, (size_t)(2100 * 512 + 4096) * 16);
    (void)hipMalloc((void **)&out, 1024 * 8);
    (void)hipMalloc((void **)&sink, 1024 * 4);
    run<0, 0, 0, 16>("56 MFMA", w, out, sink);
    run<7, 0, 0, 16>("56 MFMA + 7 ds_read_b128", w, out, sink);
    run<7, 0, 0, 8>("56 MFMA + 7 ds_read_b64", w, out, sink);
    run<3, 0, 0, 16>("56 MFMA + 3 ds_read_b128", w, out, sink);
    run<0, 2, 0, 16>("56 MFMA + 2 global_load_dwordx4", w, out, sink);
    run<0, 1, 0, 16>("56 MFMA + 1 global_load_dwordx4", w, out, sink);
    run<0, 0, 7, 16>("56 MFMA + 7 VALU", w, out, sink);
    run<7, 2, 0, 16>("56 MFMA + 7 ds_read_b128 + 2 global_load", w, out, sink);
    run<7, 2, 7, 16>("56 MFMA + 7 ds_read_b128 + 2 gload + 7 VALU", w, out, sink);
    return 0;
}
