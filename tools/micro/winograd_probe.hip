// Micro-probe: what would ONE Winograd F(2x2, 3x3) chess layer cost at four boards per workgroup?
//
// Not a kernel of the product and not a correct convolution (there is no board-edge handling and the result is never
// compared): it reproduces the INNER LOOP such a layer would have, instruction for instruction, on realistic data, on a
// full chip (256 workgroups of 256 threads, one per CU, one wave per SIMD), so that its cost is measured instead of argued:
//
//   * LDS holds one f16 image of four 8x8 boards, 256 channels (row = pixel, 512 B + 16 B pad: 135 KB).
//   * In the Winograd domain a layer is 16 independent GEMMs (one per position xi = (i, j) of the 4x4 input patch):
//     [256 oc] x [K = 256] x [N = 64 output tiles (4 boards x 16)].  A wave owns 64 oc: per position 4 oc-tiles x 4 N-tiles x
//     8 k-steps = 128 v_mfma_f32_16x16x32_f16 into 64 accumulators; 16 positions = 2048 per wave against 4608 for the
//     direct convolution (2.25 x fewer).
//   * The transformed weights U = G g G^T stream from L2 in fragment order: 16 positions x 128 KB = 2.1 MB per layer and
//     workgroup (direct: 1.18 MB), 40 distinct layers (84 MB) cycled so that they come from L2 / the Infinity Cache the way
//     a tower's do; a register ring two k-steps deep.
//   * The transformed input V = B^T d B cannot be kept (16 positions x 64 tiles x 256 channels = 524 KB): each B fragment
//     is rebuilt from the image when it is needed — 4 ds_read_b128 (the 2 x 2 pixels of the patch that position xi
//     combines) + 3 v_pk_fma_f16 per register (12 per fragment) — and feeds 4 MFMAs.
//   * The output transform A^T m A adds each position's 64 accumulators into the 4 x 64 spatial accumulators of the 2x2
//     output pixels it contributes to (+-1 coefficients; 9 of 16 (position, pixel) pairs on average).
//
// Prints microseconds per four-board layer (HIP events around a launch that runs `layers` layers).  The product's direct
// convolution takes 46.9 us per four-board layer at a full chip (kz_tower4: 1921.5 us / 41 layers).  Clock and MFMA-busy
// share: run under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES` (tools/winograd_probe.sh).
//
//   hipcc --offload-arch=gfx950 -O3 -o winograd_probe tools/micro/winograd_probe.hip && ./winograd_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h16;
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 256, RS = C * 2 + 16;     // LDS bytes per pixel row
constexpr int ROWS = 4 * 64;                // four boards
constexpr int IMG = ROWS * RS;              // 135,168 B
constexpr int ZROWS = 16;                   // what a patch pixel outside the board would read
constexpr int LDS_BYTES = IMG + ZROWS * RS;
constexpr int KSTEPS = 8;                   // 256 input channels / 32
constexpr int POS_UINT4 = KSTEPS * 4 * 4 * 64;  // uint4 per position: [k-step][wave][oc tile][lane] = 128 KB

// B^T rows of F(2x2, 3x3): out[i] = d[P[i]] + S[i] * d[Q[i]]
__device__ constexpr int BT_P[4] = {0, 1, 2, 1}, BT_Q[4] = {2, 2, 1, 3};
__device__ constexpr float BT_S[4] = {-1.f, 1.f, -1.f, -1.f};
// A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]: coefficient of position i in output pixel u
__device__ constexpr float AT[2][4] = {{1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, -1.f}};

template <bool MFMA_ON, bool TRANSFORM_ON, bool WLOAD_ON, bool IDEAL_LAYOUT>
__global__ __launch_bounds__(256, 1) void winograd_layer(const uint4 *__restrict__ weights, const h16 *__restrict__ image, int layers,
                                                         int distinct_layers, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int fr = lane & 15, kq = lane >> 4;
    // the image: post-ReLU activations (half of them exact zeros), the same for every workgroup (a different slice start)
    for (int id = tid; id < LDS_BYTES / 16; id += 256) {
        const int row = (id * 16) / RS;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < ROWS) v = reinterpret_cast<const uint4 *>(image)[(id + blockIdx.x * 977) % (IMG / 16)];
        reinterpret_cast<uint4 *>(lds)[id] = v;
    }
    __syncthreads();

    // N tile n (16 output tiles) = board n, output tile fr = (ty, tx) of its 4 x 4: patch origin pixel (2 ty - 1, 2 tx - 1);
    // this probe clamps the origin into the board instead of reading zero rows for the edge (same instruction stream)
    int base[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const int ty = fr >> 2, tx = fr & 3;
        const int oy = ty == 0 ? 0 : 2 * ty - 1, ox = tx == 0 ? 0 : 2 * tx - 1;
        // k-step channel assignment as in kz_tower.hip: lane group kq reads the 16-byte piece at 256 (kq & 1) + 128 (kq >> 1).
        // IDEAL_LAYOUT: the 16 lanes of a fragment read 16 CONSECUTIVE rows, as the direct convolution's fragment reads do
        // (conflict-free with this row stride) — the best case for a layout a real kernel would have to find; the natural
        // patch origins (rows 2 ty x 8 + 2 tx apart) collide four ways on the bank slots with this row stride.
        base[n] = (IDEAL_LAYOUT ? (n * 48 + fr) : (n * 64 + oy * 8 + ox)) * RS + 256 * (kq & 1) + 128 * (kq >> 1);
    }
    f32x4 spatial[4][4][4];  // [output pixel 2u+v][oc tile][N tile]: 256 registers
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int o = 0; o < 4; o++)
#pragma unroll
            for (int n = 0; n < 4; n++) spatial[p][o][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int layer = 0; layer < layers; layer++) {
        const uint4 *wl = weights + (size_t)(layer % distinct_layers) * 16 * POS_UINT4 + (wave * 4) * 64 + lane;
#pragma unroll
        for (int xi = 0; xi < 16; xi++) {
            const int pi = xi >> 2, pj = xi & 3;
            // the four pixels of the patch this position combines, as byte offsets from the patch origin
            const int o11 = (BT_P[pi] * 8 + BT_P[pj]) * RS, o12 = (BT_P[pi] * 8 + BT_Q[pj]) * RS;
            const int o21 = (BT_Q[pi] * 8 + BT_P[pj]) * RS, o22 = (BT_Q[pi] * 8 + BT_Q[pj]) * RS;
            const h16x2 sa = h16x2{(h16)BT_S[pi], (h16)BT_S[pi]}, sb = h16x2{(h16)BT_S[pj], (h16)BT_S[pj]};
            const uint4 *wp = wl + (size_t)xi * POS_UINT4;
            f32x4 acc[4][4];
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[o][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            uint4 wreg[2][4];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                wreg[0][o] = WLOAD_ON ? wp[o * 64] : make_uint4(0x2c002c00u + lane, 0x2c00ac00u, 0x2800a800u, 0x30003000u + o);
                wreg[1][o] = WLOAD_ON ? wp[1024 + o * 64] : wreg[0][o];
            }
            // 32 units per position: unit u = (k-step u / 4, N tile u % 4) = one B fragment (4 reads + 12 packed FMAs) and its
            // 4 MFMAs.  The reads run DEPTH units ahead of their use (a ring of DEPTH + 1 source sets, 16 registers each),
            // so that with one wave per SIMD their latency hides behind the MFMAs and FMAs of the units in between.
            constexpr int UNITS = KSTEPS * 4, DEPTH = 2;
            h16x8 src[DEPTH + 1][4];
            auto issue = [&](int u) __attribute__((always_inline)) {
                const int b = base[u & 3] + (u >> 2) * 16;
                src[u % (DEPTH + 1)][0] = *reinterpret_cast<const h16x8 *>(lds + b + o11);
                if (TRANSFORM_ON) {
                    src[u % (DEPTH + 1)][1] = *reinterpret_cast<const h16x8 *>(lds + b + o12);
                    src[u % (DEPTH + 1)][2] = *reinterpret_cast<const h16x8 *>(lds + b + o21);
                    src[u % (DEPTH + 1)][3] = *reinterpret_cast<const h16x8 *>(lds + b + o22);
                }
            };
#pragma unroll
            for (int u = 0; u < DEPTH; u++) issue(u);
#pragma unroll
            for (int u = 0; u < UNITS; u++) {
                const int ks = u >> 2, n = u & 3;
                if (u + DEPTH < UNITS) issue(u + DEPTH);
                h16x8 af[4];
#pragma unroll
                for (int o = 0; o < 4; o++) af[o] = __builtin_bit_cast(h16x8, wreg[ks & 1][o]);
                h16x8 bf;
                const h16x8 *d = src[u % (DEPTH + 1)];
                if (TRANSFORM_ON) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {  // V = (d11 + sb d12) + sa (d21 + sb d22): three v_pk_fma_f16 per register
                        const h16x2 a = h16x2{d[0][2 * r], d[0][2 * r + 1]}, bq = h16x2{d[1][2 * r], d[1][2 * r + 1]};
                        const h16x2 c = h16x2{d[2][2 * r], d[2][2 * r + 1]}, e = h16x2{d[3][2 * r], d[3][2 * r + 1]};
                        const h16x2 t1 = bq * sb + a, t2 = e * sb + c, v = t2 * sa + t1;
                        bf[2 * r] = v[0];
                        bf[2 * r + 1] = v[1];
                    }
                } else {
                    bf = d[0];
                }
                if (MFMA_ON) {
#pragma unroll
                    for (int o = 0; o < 4; o++) acc[o][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[o], bf, acc[o][n], 0, 0, 0);
                } else {
#pragma unroll
                    for (int o = 0; o < 4; o++) acc[o][n][0] += (float)bf[o] + (float)af[o][0];
                }
                if (n == 3 && WLOAD_ON && ks + 2 < KSTEPS) {
#pragma unroll
                    for (int o = 0; o < 4; o++) wreg[ks & 1][o] = wp[(size_t)(ks + 2) * 1024 + o * 64];
                }
                __builtin_amdgcn_sched_barrier(0);  // (one unit = one scheduling region: the reads stay DEPTH units ahead)
            }
            // output transform: pixel (u, v) of the 2 x 2 output tile += AT[u][i] * AT[v][j] * m
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int v = 0; v < 2; v++) {
                    const float coef = AT[u][pi] * AT[v][pj];
                    if (coef == 0.0f) continue;  // (compile time)
#pragma unroll
                    for (int o = 0; o < 4; o++)
#pragma unroll
                        for (int n = 0; n < 4; n++)
                            spatial[2 * u + v][o][n] = coef > 0.0f ? spatial[2 * u + v][o][n] + acc[o][n] : spatial[2 * u + v][o][n] - acc[o][n];
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (a real layer would now add bias / ReLU / residual and write the four pixel sets back into the image: 256
        // conversions and 64 ds_write_b64 per lane and layer, a few percent of the loop above; the probe keeps summing)
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int o = 0; o < 4; o++)
#pragma unroll
            for (int n = 0; n < 4; n++) s += spatial[p][o][n][0] + spatial[p][o][n][1] + spatial[p][o][n][2] + spatial[p][o][n][3];
    if (s == 123.456f) sink[tid] = s;
}

static float gauss() {
    float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = (rand() + 1.0f) / (RAND_MAX + 2.0f);
    return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v);
}

template <bool MFMA_ON, bool TRANSFORM_ON, bool WLOAD_ON, bool IDEAL_LAYOUT = false>
static void run(const char *name, const uint4 *w, const h16 *img, int distinct, float *sink) {
    const int layers = 400;  // ~15 ms per launch: long enough for the clock to settle over the repetitions
    (void)hipFuncSetAttribute((const void *)winograd_layer<MFMA_ON, TRANSFORM_ON, WLOAD_ON, IDEAL_LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 10; rep++) {
        (void)hipEventRecord(e0, 0);
        winograd_layer<MFMA_ON, TRANSFORM_ON, WLOAD_ON, IDEAL_LAYOUT><<<256, 256, LDS_BYTES>>>(w, img, layers, distinct, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 4 && ms < best) best = ms;
    }
    const double us = best * 1e3 / layers;
    // direct-convolution FLOP of the layer on 1024 boards (the roofline numerator does not change with the algorithm)
    const double flop = 1024.0 * 64 * 9 * 256 * 256 * 2;
    printf("%-78s %7.2f us per 4-board layer  (%5.0f TFLOP/s of direct-conv work = %4.1f %% of 2500; direct today: 46.9 us)\n", name, us,
           flop / (us * 1e-6) / 1e12, flop / (us * 1e-6) / 1e12 / 25.0);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  HIP error: %s\n", hipGetErrorString(err));
}

int main() {
    const int distinct = 40;
    const size_t wn = (size_t)distinct * 16 * POS_UINT4;  // uint4: 84 MB
    std::vector<h16> hw(wn * 8), himg(IMG / 2);
    srand(1);
    // U = G g G^T of N(0, 0.03) weights: the transformed taps are sums of up to 9 taps with coefficients 1, 1/2, 1/4 —
    // variance between 1/16 and 1 of the taps'; N(0, 0.03 * 0.6) stands for the mix
    for (auto &v : hw) v = (h16)(0.018f * gauss());
    for (auto &v : himg) { const float g = gauss(); v = (h16)(g > 0 ? g : 0.0f); }  // post-ReLU: half zeros
    uint4 *w; h16 *img; float *sink;
    (void)hipMalloc((void **)&w, wn * 16);
    (void)hipMalloc((void **)&img, IMG);
    (void)hipMalloc((void **)&sink, 4096);
    (void)hipMemcpy(w, hw.data(), wn * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(img, himg.data(), IMG, hipMemcpyHostToDevice);
    run<true, true, true>("Winograd F(2x2,3x3) inner loop: weight stream + input transform + MFMA + output transform", w, img, distinct, sink);
    run<true, true, false>("  the same without the weight stream (A operands constant)", w, img, distinct, sink);
    run<true, false, true>("  the same with ONE fragment read and no transform per B fragment (LDS / VALU share)", w, img, distinct, sink);
    run<false, true, true>("  the same without the MFMAs (what the feeding alone costs)", w, img, distinct, sink);
    run<true, true, true, true>("BEST CASE: every fragment read conflict-free (16 consecutive rows per read)", w, img, distinct, sink);
    run<false, true, true, true>("  best case without the MFMAs (feeding alone)", w, img, distinct, sink);
    return 0;
}
