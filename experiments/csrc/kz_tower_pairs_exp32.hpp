// kz_tower_pairs_exp32.hpp — EXPERIMENT build only (libkzhip_exp.so): the 256-channel launch on v_mfma_f32_32x32x16_f16.
// Included by kz_tower_pairs.hpp under KZ_EXPERIMENTS, inside its anonymous namespace.
#pragma once
// ---------------------------------------------------------------------------------------------------------------------
// EXPERIMENT (libkzhip_exp.so only; opt-in: KZ_SPLIT_MFMA32=1 for KZ_DTYPE_F32_SPLIT16, KZ_F16G_MFMA32=1 for the plain-f16 launch; read once,
// the weight packing and the launch must agree): the 256-channel launch on 64 pixel rows on v_mfma_f32_32x32x16_f16.
// The f16 matrix cores are power-limited on this data, and in a loop of nothing but MFMAs on
// register-resident operands a 32x32x16 stream sustains 1.79 PFLOP/s against 1.63 for 16x16x32
// (tools/micro/mfma_power.hip).  In THIS kernel it does not carry over: same cycles and MFMA-busy share (2.766 M, 80 %) but
// the chip settles at 1.71 GHz instead of 1.88 (tools/pmc_split.sh), 143-152k against 157-165k evals/s; parity-tested
// (tests/test_gpu_parity.py), kept as the record of that measurement.
// Same LDS images, same weight stream size, same sums as kz_tower_resident_split<256, 4>:
//   64 pixel rows = two tiles of 32, a wave's 64 output channels = two tiles of 32; a k-step (32 input channels of one tap)
//   is two halves of 16 channels; fragment lane (n = lane % 32, kg = lane / 32) holds 8 consecutive channels of pixel row
//   n (B operand) or of output channel n (A operand): piece q = 2 * half + kg of the k-step, i.e. channels
//   8 ch + {0, C/2, C/4, 3C/4}[q] + j — the byte offsets of the 16x16x32 launch, so the fragment reads stay conflict-free
//   (the four 16-lane groups of a ds_read_b128 see 16 consecutive-modulo-16 rows each).
//   Accumulator register v of tile (o, t): output channel 32 o + 8 (v / 4) + 4 kg + v % 4 of pixel row 32 t + n.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool SPLIT>
__global__ __launch_bounds__(256, 1) void kz_tower_resident_split32(SplitDev a) {
    constexpr int C = 256, NT = 4;
    using L = Geo<C, NT, SPLIT>;
    constexpr int PARTS = L::PARTS, PF = L::PF;
    constexpr int RS = L::RS, G = L::G, DELTA = L::DELTA, XH = L::XH, YH = L::YH, ZH = L::ZH;
    constexpr int NF = 4;  // weight fragments per wave, k-step and part: (o, half) = f / 2, f % 2
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n = lane & 31, kg = lane >> 5;
    const int board0 = blockIdx.x * a.nb;
    const int boards = min(a.nb, a.batch - board0);
    const int rows_valid = boards * a.hw;
    const int layers = 2 * a.depth;
    const int total_ksteps = layers * 9 * G;

    // ---- weight stream: prime PF stages (stage s = k-step g % PF) ----
    const uint4 *wp_stem = a.w + wave * NF * 64 + lane;
    const uint4 *wp = wp_stem + (size_t)9 * L::STEP;
    auto wload = [&](int gk, int part, int f) __attribute__((always_inline)) {
        return wp[(size_t)gk * L::STEP + part * (4 * NF * 64) + f * 64];
    };
    uint4 wreg[PF][PARTS][NF];
#pragma unroll
    for (int s = 0; s < PF; s++)
#pragma unroll
        for (int part = 0; part < PARTS; part++)
#pragma unroll
            for (int f = 0; f < NF; f++) wreg[s][part][f] = wload(s < total_ksteps ? s : total_ksteps - 1, part, f);
    int g = 0;
    auto ring_take = [&](int stage, h16x8 (&ah)[NF], h16x8 (&al)[NF]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NF; f++) {
            ah[f] = *reinterpret_cast<const h16x8 *>(&wreg[stage][0][f]);
            if constexpr (SPLIT) al[f] = *reinterpret_cast<const h16x8 *>(&wreg[stage][PARTS - 1][f]);
        }
        const int gn = g + PF < total_ksteps ? g + PF : total_ksteps - 1;
#pragma unroll
        for (int part = 0; part < PARTS; part++)
#pragma unroll
            for (int f = 0; f < NF; f++) wreg[stage][part][f] = wload(gn, part, f);
    };

    // ---- zero rows and the stem input (f32 -> hi/lo, 32 channels per square; rows beyond the batch are zero) ----
    for (int id = tid; id < 16 * RS / 16; id += 256) {
        *reinterpret_cast<uint4 *>(lds + ZH + id * 16) = make_uint4(0, 0, 0, 0);
        if constexpr (SPLIT) *reinterpret_cast<uint4 *>(lds + ZH + DELTA + id * 16) = make_uint4(0, 0, 0, 0);
    }
    for (int id = tid; id < L::ROWS * 8; id += 256) {  // (row, 4-channel piece)
        const int row = id >> 3, c4 = id & 7;
        const bool have = row < rows_valid && c4 * 4 < a.ldx0;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bits) {
            // encode_input_full (rust/kz-core/src/mapping/mod.rs:40-63): scalar planes first, then the bool planes
            if (row < rows_valid) {
                const int b = (int)(((unsigned)row * a.inv_hw) >> 16), q = row - b * a.hw;
                const uint8_t *bb = a.bits + (size_t)(board0 + b) * a.bits_stride;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int ch = c4 * 4 + j;
                    if (ch < a.n_scalar) {
                        v[j] = a.scalars_in[(size_t)(board0 + b) * a.n_scalar + ch];
                    } else if (ch < a.n_scalar + a.n_bool) {
                        const unsigned bit = (unsigned)(ch - a.n_scalar) * a.hw + q;
                        v[j] = (float)((bb[bit >> 3] >> (bit & 7)) & 1);
                    }
                }
            }
        } else if constexpr (SPLIT) {
            if (have) v = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(a.x0) + ((size_t)board0 * a.hw + row) * a.ldx0 + c4 * 4);
        } else {
            if (have) {
                const h16x4 t = *reinterpret_cast<const h16x4 *>(static_cast<const h16 *>(a.x0) + ((size_t)board0 * a.hw + row) * a.ldx0 + c4 * 4);
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = (float)t[j];
            }
        }
        h16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<h16x4 *>(lds + L::SH + row * 64 + c4 * 8) = hi;
        if constexpr (SPLIT) *reinterpret_cast<h16x4 *>(lds + L::SL + row * 64 + c4 * 8) = lo;
    }

    // bit 2 tap + t of okbits: for this lane's row of 32-row tile t the tap lands on the same board (one register: an
    // array indexed by the tap went to scratch memory)
    unsigned okbits = 0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int r = t * 32 + n;
        const int b = (int)(((unsigned)r * a.inv_hw) >> 16), q = r - b * a.hw;
        const unsigned valid = r < rows_valid;
        const int yy = (int)(((unsigned)q * a.inv_w) >> 16), xx = q - yy * a.w_;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const unsigned ym = tap / 3 == 0 ? (unsigned)(yy >= 1) : tap / 3 == 2 ? (unsigned)(yy <= a.h - 2) : 1u;
            const unsigned xm = tap % 3 == 0 ? (unsigned)(xx >= 1) : tap % 3 == 2 ? (unsigned)(xx <= a.w_ - 2) : 1u;
            okbits |= (valid & ym & xm) << (2 * tap + t);
        }
    }
    __syncthreads();

    f32x16 acc[2][2];       // [o][t]
    f32x4 bias_next[2][4];  // [o][v / 4]
    auto oc_of = [&](int o, int g4) __attribute__((always_inline)) { return wave * 64 + o * 32 + 8 * g4 + 4 * kg; };
    auto fetch_bias = [&](int row) __attribute__((always_inline)) {
        const int l = row <= layers ? row : layers;
#pragma unroll
        for (int o = 0; o < 2; o++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) bias_next[o][g4] = *reinterpret_cast<const f32x4 *>(a.bias + l * C + oc_of(o, g4));
    };
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < 2; o++)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int v = 0; v < 16; v++) acc[o][t][v] = bias_next[o][v >> 2][v & 3];
    };
    auto quad = [&](const f32x16 &c, int g4) __attribute__((always_inline)) {
        return f32x4{c[4 * g4], c[4 * g4 + 1], c[4 * g4 + 2], c[4 * g4 + 3]};
    };

    // one half (16 input channels) of a k-step: three MFMAs per (output tile, pixel tile): hi*hi + hi*lo + lo*hi; the
    // weight fragment is held across the pixel tiles
    auto mfma3 = [&](int half, const h16x8 (&ah)[NF], const h16x8 (&al)[NF], const h16x8 (&bh)[4], const h16x8 (&bl)[4])
                     __attribute__((always_inline)) {
        // (term outermost: the three MFMAs of a tile are four MFMAs apart — back to back they wait for each other's result)
#pragma unroll
        for (int term = SPLIT ? 0 : 2; term < 3; term++)
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int f = o * 2 + half, bi = t * 2 + half;
                    acc[o][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al[f] : ah[f], term == 1 ? bl[bi] : bh[bi],
                                                                       acc[o][t], 0, 0, 0);
                }
    };
    auto ok_of = [&](int tap) __attribute__((always_inline)) { return (okbits >> (2 * tap)) & 3u; };

    // ---- stem: 9 k-steps over the 32 (padded) input channels; conv + bias, no activation (post_act.py:205) ----
    fetch_bias(0);
    init_acc();
    fetch_bias(1);
#pragma nounroll
    for (int tap = 0; tap < 9; tap++) {
        const int shift = (tap / 3 - 1) * a.w_ + (tap % 3 - 1);
        const unsigned ok = ok_of(tap);
        h16x8 ah[NF], al[NF], bh[4], bl[4];
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const uint4 th = wp_stem[(size_t)tap * L::STEP + f * 64];
            ah[f] = *reinterpret_cast<const h16x8 *>(&th);
            if constexpr (SPLIT) {
                const uint4 tl = wp_stem[(size_t)tap * L::STEP + 4 * NF * 64 + f * 64];
                al[f] = *reinterpret_cast<const h16x8 *>(&tl);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int off = (t * 32 + n + shift) * 64 + (half * 2 + kg) * 16;  // stem: natural k (channel = 8 q + j)
                const bool valid = (ok >> t) & 1;
                bh[t * 2 + half] = valid ? *reinterpret_cast<const h16x8 *>(lds + L::SH + off) : h16x8{};
                if constexpr (SPLIT) bl[t * 2 + half] = valid ? *reinterpret_cast<const h16x8 *>(lds + L::SL + off) : h16x8{};
            }
        mfma3(0, ah, al, bh, bl);
        mfma3(1, ah, al, bh, bl);
    }

    // this lane's slice of an image: pixel row n of tile 0, channel oc_of(0, 0)
    const int epi_base = n * RS + (wave * 64 + 4 * kg) * 2;
    auto epi_off = [&](int o, int t, int g4) __attribute__((always_inline)) { return epi_base + t * 32 * RS + (o * 32 + 8 * g4) * 2; };
    // epilogue: [relu]; [+ residual X]; -> (hi, lo) -> the image pair at dst_h
    auto epilogue = [&](int dst_h, bool relu, bool residual) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < 2; o++)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int off = epi_off(o, t, g4);
                    f32x4 v = quad(acc[o][t], g4);
                    if (relu) {
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                    }
                    if (residual) {  // added in f32, AFTER the ReLU (post_act.py:227-228)
                        const h16x4 rh = *reinterpret_cast<const h16x4 *>(lds + XH + off);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] += (float)rh[j];
                        if constexpr (SPLIT) {
                            const h16x4 rl = *reinterpret_cast<const h16x4 *>(lds + XH + DELTA + off);
#pragma unroll
                            for (int j = 0; j < 4; j++) v[j] += (float)rl[j];
                        }
                    }
                    if constexpr (SPLIT) {
                        h16x4 hi, lo;
                        split4(v, hi, lo);
                        *reinterpret_cast<h16x4 *>(lds + dst_h + off) = hi;
                        *reinterpret_cast<h16x4 *>(lds + dst_h + DELTA + off) = lo;
                    } else {
                        *reinterpret_cast<h16x4 *>(lds + dst_h + off) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                    }
                }
    };
    epilogue(XH, false, false);
    __syncthreads();

    // ---- convolution passes over the LDS images: piece q = 2 half + kg of a k-step sits at C kg + (C/2) half + 16 ch ----
    const int frag_base = n * RS + C * kg;
    // T[t] = LDS address, in the hi block, of this lane's fragment row (pixel shifted by the tap) or of a zero row
    auto tap_rows = [&](int tap, int src_h, int (&T)[2]) __attribute__((always_inline)) {
        const int shift = (tap / 3 - 1) * a.w_ + (tap % 3 - 1);
        const unsigned ok = ok_of(tap);
        const int shifted = src_h + frag_base + shift * RS;
        const int zrow = ZH + ((n + shift) & 15) * RS + C * kg;
#pragma unroll
        for (int t = 0; t < 2; t++) T[t] = ((ok >> t) & 1) ? shifted + t * 32 * RS : zrow;
    };
    auto conv_3x3 = [&](int src_h) __attribute__((always_inline)) {
        int T[2], Tn[2];
        h16x8 bh[2][4], bl[2][4];  // [buffer][2 t + half]
        tap_rows(0, src_h, T);
        auto rd = [&](int t, int extra) __attribute__((always_inline)) { return *reinterpret_cast<const h16x8 *>(lds + t + extra); };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            bh[0][i] = rd(T[i >> 1], (i & 1) * (C / 2));
            if constexpr (SPLIT) bl[0][i] = rd(T[i >> 1], DELTA + (i & 1) * (C / 2));
        }
#pragma nounroll
        for (int tap = 0; tap < 9; tap++) {
            tap_rows(tap + 1 < 9 ? tap + 1 : tap, src_h, Tn);
#pragma unroll
            for (int ch = 0; ch < G; ch++) {
                const int stage = ch & (PF - 1), cur = ch & 1, nxt = cur ^ 1;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int half_off = (i & 1) * (C / 2);
                    bh[nxt][i] = ch < G - 1 ? rd(T[i >> 1], half_off + (ch + 1) * 16) : rd(Tn[i >> 1], half_off);
                    if constexpr (SPLIT)
                        bl[nxt][i] = ch < G - 1 ? rd(T[i >> 1], DELTA + half_off + (ch + 1) * 16) : rd(Tn[i >> 1], DELTA + half_off);
                }
                h16x8 ah[NF], al[NF];
                ring_take(stage, ah, al);
                mfma3(0, ah, al, bh[cur], bl[cur]);
                mfma3(1, ah, al, bh[cur], bl[cur]);
                // every memory instruction in the shadow of an MFMA (24 free issue cycles each): the ring refills, the
                // fragment reads, then the remaining MFMAs back to back
                constexpr int NMF = (SPLIT ? 3 : 1) * 8, NVM = PARTS * NF, NDS = PARTS * 4;
                constexpr int PAIRED = NVM + NDS < NMF ? NVM + NDS : NMF;
#pragma unroll
                for (int i = 0; i < NVM; i++) {
                    if (i < PAIRED) __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_VMEM_READ, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < NDS; i++) {
                    if (NVM + i < PAIRED) __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_DS_READ, 1, 0);
                }
                if constexpr (NMF > PAIRED) __builtin_amdgcn_sched_group_barrier(SG_MFMA, NMF - PAIRED, 0);
                __builtin_amdgcn_sched_barrier(0);
                g++;
            }
#pragma unroll
            for (int t = 0; t < 2; t++) T[t] = Tn[t];
        }
    };

    // ---- the 2*depth 3x3 convolutions ----
    for (int layer = 1; layer <= layers; layer++) {
        const bool is_b = (layer & 1) == 0;  // conv A: X -> Y; conv B: Y -> X (+ residual)
        init_acc();
        fetch_bias(layer + 1);
        conv_3x3(is_b ? YH : XH);
        if (!is_b) {
            epilogue(YH, true, false);
        } else if (layer != layers) {
            epilogue(XH, true, true);
        } else {
            // last layer: ReLU, residual, final BN -> rows of the tower output in global memory
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int oc = oc_of(o, g4);
                    const f32x4 ps = *reinterpret_cast<const f32x4 *>(a.post_scale + oc);
                    const f32x4 pt = *reinterpret_cast<const f32x4 *>(a.post_shift + oc);
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const int off = epi_off(o, t, g4);
                        f32x4 v = quad(acc[o][t], g4);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                        const h16x4 rh = *reinterpret_cast<const h16x4 *>(lds + XH + off);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] += (float)rh[j];
                        if constexpr (SPLIT) {
                            const h16x4 rl = *reinterpret_cast<const h16x4 *>(lds + XH + DELTA + off);
#pragma unroll
                            for (int j = 0; j < 4; j++) v[j] += (float)rl[j];
                        }
                        v = v * ps + pt;
                        const int r = t * 32 + n;
                        const size_t idx = ((size_t)board0 * a.hw + r) * a.ldy + oc;
                        if (r < rows_valid) {
                            if constexpr (SPLIT) *reinterpret_cast<f32x4 *>(static_cast<float *>(a.y) + idx) = v;
                            else *reinterpret_cast<h16x4 *>(static_cast<h16 *>(a.y) + idx) = h16x4{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                        }
                    }
                }
        }
        __syncthreads();
    }
}

template <bool SPLIT>
void launch32(const SplitDev &d, int grid, hipStream_t stream) {
    static thread_local unsigned long long done_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((done_mask >> (dev & 63)) & 1)) {
        (void)hipFuncSetAttribute((const void *)kz_tower_resident_split32<SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  Geo<256, 4, SPLIT>::LDS_BYTES_OWN_STEM);
        done_mask |= 1ull << (dev & 63);
    }
    kz_tower_resident_split32<SPLIT><<<grid, 256, Geo<256, 4, SPLIT>::LDS_BYTES_OWN_STEM, stream>>>(d);
}
