// kz_decode_dev.hpp — decode_output (rust/kz-core/src/network/common.rs:16-100) as the last step of a launch that has the
// network's heads inside: the launch then writes what `ZeroEvaluation` holds — values [5] and one probability per available
// move — instead of the raw scalars and the policy_len logits per board.  Device code only; included INSIDE
// `namespace kz { namespace {` of a .hip file.
//
// Same arithmetic as the stand-alone kz_decode_output kernel (kz_kernels.hip), which stays for the paths whose heads are
// separate launches:
//   values = [tanh(s0), softmax(s1..s3), s4]                                   (common.rs:60-74)
//   probs  = softmax over the logits at the board's available-move indices, in the order given (common.rs:77-86);
//            a finished board has an empty range and gets nothing               (:77 `map_or(vec![], ..)`)
//   a softmax sum that is not strictly positive — the reference's assert (:110) — or a move index outside the policy
//   (the reference would panic on the slice index) raises *error_flag; the host fails the call that returns the batch.
// All five pointers may be pinned host memory (the zero-copy slots): every word of the move list is read once, every
// output word written once, the flag is a plain store.
#pragma once

struct DecodeDev {
    const int64_t *move_offsets;  // [batch + 1]; nullptr: the launch writes raw scalars and logits
    const int32_t *move_indices;
    float *values;                // [batch][5]
    float *probs;                 // parallel to move_indices
    int *error_flag;
    int policy_len;
};

__device__ __forceinline__ float decode_wave_max_nan(float v) {  // max over the wave; a NaN anywhere gives NaN
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = (v != v || o != o) ? NAN : fmaxf(v, o);
    }
    return v;
}

__device__ __forceinline__ float decode_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ONE WAVE decodes ONE board.  raw: the board's five raw scalars (LDS or registers' spill — any generic pointer);
// stage: `cap` floats of LDS owned by this wave for the duration of the call (logits, then exponentials: a move list is
// read once even when it sits in host memory; a list longer than cap re-reads its tail);
// logit_at(idx): the board's logit at policy index idx (0 <= idx < policy_len guaranteed by the caller below).
template <class LogitAt>
__device__ __forceinline__ void decode_board_wave(const DecodeDev &d, int board, int lane, const float *raw, float *stage,
                                                  int cap, LogitAt logit_at) {
    if (lane == 0) {
        const float s0 = raw[0], s1 = raw[1], s2 = raw[2], s3 = raw[3], s4 = raw[4];
        const float m = fmaxf(s1, fmaxf(s2, s3));
        const float e0 = expf(s1 - m), e1 = expf(s2 - m), e2 = expf(s3 - m), sum = e0 + e1 + e2;
        float *v = d.values + (size_t)board * 5;
        v[0] = tanhf(s0);
        v[1] = e0 / sum;
        v[2] = e1 / sum;
        v[3] = e2 / sum;
        v[4] = s4;
        if (!(sum > 0.0f)) *reinterpret_cast<volatile int *>(d.error_flag) = 1;
    }
    const int64_t lo = d.move_offsets[board];
    const int n = (int)(d.move_offsets[board + 1] - lo);
    if (n <= 0) return;
    auto logit_of = [&](int i) {
        const int idx = d.move_indices[lo + i];
        return (idx >= 0 && idx < d.policy_len) ? logit_at(idx) : NAN;  // a bad index poisons the sum -> error flag
    };
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) {
        const float v = logit_of(i);
        if (i < cap) stage[i] = v;
        mx = (mx != mx || v != v) ? NAN : fmaxf(mx, v);
    }
    mx = decode_wave_max_nan(mx);
    float sum = 0.0f;
    for (int i = lane; i < n; i += 64) {
        const float e = expf((i < cap ? stage[i] : logit_of(i)) - mx);
        if (i < cap) stage[i] = e;
        sum += e;
    }
    sum = decode_wave_sum(sum);
    if (lane == 0 && !(sum > 0.0f)) *reinterpret_cast<volatile int *>(d.error_flag) = 1;
    for (int i = lane; i < n; i += 64) d.probs[lo + i] = (i < cap ? stage[i] : expf(logit_of(i) - mx)) / sum;
}
