/*
 * kz_oracle.c — CPU restatement of the kZero self-play NN-eval path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the checker /
 * reported CPU baseline.  The product path (kzero_amd/csrc) never links or calls it.
 *
 * Parity status: the arithmetic of this path lives in the un-vendored crates.io
 * dependency kn-graph 0.7.3 (rust/Cargo.toml:47-51; call site
 * rust/kz-core/src/network/cpu.rs:50 `cpu_eval_graph_exec`) and no reference test pins a
 * numeric network output.  This oracle is therefore pinned against
 * golden vectors produced HERE by the reference's own PyTorch network definition
 * (python/lib/model/post_act.py, the module the ONNX is exported from) with the
 * committed generator oracle/gen_golden.py — see tests/test_oracle_golden.py — plus the
 * BitBuffer known answers of rust/kz-core/src/mapping/bit_buffer.rs:112-164.
 *
 * Every function cites the reference lines it restates.  Layout is NCHW f32 and the
 * convolution is the direct 7-loop form, like the reference CPU executor.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define KZO_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* KZMODEL1 container (format: kzero_amd/model_file.py)                       */
/* ------------------------------------------------------------------------- */

typedef struct {
    char name[96];
    int dtype; /* 0 f32, 1 i64 */
    int ndim;
    uint64_t dims[6];
    const void *data;
    uint64_t count;
} kzo_tensor;

typedef struct {
    char key[64];
    int kind; /* 0 int, 1 float, 2 string */
    int64_t i;
    double f;
    char s[64];
} kzo_meta;

typedef struct kzo_net {
    uint8_t *blob;
    int n_meta;
    kzo_meta *meta;
    int n_tensors;
    kzo_tensor *tensors;

    int h, w, n_scalar, n_bool, c_in;
    int depth, channels, final_affine;
    int tower_kind; /* 0 ResTower (post_act.py:201-211), 1 AttentionTower (attention.py:8-45), 2 DenseNetwork (simple.py:7-33: the whole net) */
    int dn_res;
    int att_heads, att_dk, att_dv, att_dff;
    float att_alpha, ln_eps;
    int sh_channels, sh_size;
    int policy_len;
    int policy_kind; /* 0 ataxx_conv, 1 conv, 2 attention, 3 dense, 4 arimaa */
    int arimaa_hidden_channels, arimaa_hidden_size;
    int policy_conv_channels, policy_extra_moves, policy_query_channels;
    int dense_hidden_channels, dense_hidden_size;
    float bn_eps;
} kzo_net;

static __thread char kzo_err[256];

KZO_EXPORT const char *kzo_last_error(void) { return kzo_err; }

static int fail(const char *msg) {
    snprintf(kzo_err, sizeof kzo_err, "%s", msg);
    return -1;
}

typedef struct {
    const uint8_t *p;
    size_t left;
    int bad;
} reader;

static void rd(reader *r, void *dst, size_t n) {
    if (r->left < n) {
        r->bad = 1;
        memset(dst, 0, n);
        return;
    }
    memcpy(dst, r->p, n);
    r->p += n;
    r->left -= n;
}

static const kzo_meta *find_meta(const kzo_net *net, const char *key) {
    for (int i = 0; i < net->n_meta; i++)
        if (!strcmp(net->meta[i].key, key)) return &net->meta[i];
    return NULL;
}

static int64_t meta_int(const kzo_net *net, const char *key, int64_t dflt) {
    const kzo_meta *m = find_meta(net, key);
    if (!m) return dflt;
    return m->kind == 1 ? (int64_t)m->f : m->i;
}

static const kzo_tensor *find_tensor(const kzo_net *net, const char *name) {
    for (int i = 0; i < net->n_tensors; i++)
        if (!strcmp(net->tensors[i].name, name)) return &net->tensors[i];
    return NULL;
}

static const float *tensor_f32(const kzo_net *net, const char *name, uint64_t expect) {
    const kzo_tensor *t = find_tensor(net, name);
    if (!t || t->dtype != 0 || t->count != expect) {
        snprintf(kzo_err, sizeof kzo_err, "missing or mis-shaped tensor '%s' (want %llu values)", name,
                 (unsigned long long)expect);
        return NULL;
    }
    return (const float *)t->data;
}

KZO_EXPORT void kzo_free(kzo_net *net) {
    if (!net) return;
    free(net->blob);
    free(net->meta);
    free(net->tensors);
    free(net);
}

KZO_EXPORT int kzo_load(const void *blob, size_t len, kzo_net **out) {
    kzo_net *net = calloc(1, sizeof *net);
    net->blob = malloc(len);
    memcpy(net->blob, blob, len);
    reader r = {net->blob, len, 0};

    char magic[8];
    rd(&r, magic, 8);
    if (r.bad || memcmp(magic, "KZMODEL1", 8)) {
        kzo_free(net);
        return fail("not a KZMODEL1 container");
    }
    uint32_t n_meta;
    rd(&r, &n_meta, 4);
    net->n_meta = (int)n_meta;
    net->meta = calloc(n_meta ? n_meta : 1, sizeof(kzo_meta));
    for (uint32_t i = 0; i < n_meta && !r.bad; i++) {
        kzo_meta *m = &net->meta[i];
        uint16_t klen;
        rd(&r, &klen, 2);
        if (klen >= sizeof m->key) { r.bad = 1; break; }
        rd(&r, m->key, klen);
        uint8_t kind;
        rd(&r, &kind, 1);
        m->kind = kind;
        if (kind == 0) rd(&r, &m->i, 8);
        else if (kind == 1) rd(&r, &m->f, 8);
        else if (kind == 2) {
            uint32_t vlen;
            rd(&r, &vlen, 4);
            if (vlen >= sizeof m->s) { r.bad = 1; break; }
            rd(&r, m->s, vlen);
        } else r.bad = 1;
    }
    uint32_t n_tensors;
    rd(&r, &n_tensors, 4);
    net->n_tensors = (int)n_tensors;
    net->tensors = calloc(n_tensors ? n_tensors : 1, sizeof(kzo_tensor));
    uint64_t *offsets = calloc(n_tensors ? n_tensors : 1, sizeof(uint64_t));
    uint64_t *nbytes = calloc(n_tensors ? n_tensors : 1, sizeof(uint64_t));
    for (uint32_t i = 0; i < n_tensors && !r.bad; i++) {
        kzo_tensor *t = &net->tensors[i];
        uint16_t nlen;
        rd(&r, &nlen, 2);
        if (nlen >= sizeof t->name) { r.bad = 1; break; }
        rd(&r, t->name, nlen);
        uint8_t dtype;
        uint32_t ndim;
        rd(&r, &dtype, 1);
        rd(&r, &ndim, 4);
        if (ndim > 6) { r.bad = 1; break; }
        t->dtype = dtype;
        t->ndim = (int)ndim;
        t->count = 1;
        for (uint32_t d = 0; d < ndim; d++) {
            rd(&r, &t->dims[d], 8);
            t->count *= t->dims[d];
        }
        rd(&r, &offsets[i], 8);
        rd(&r, &nbytes[i], 8);
    }
    uint64_t data_len;
    rd(&r, &data_len, 8);
    if (r.bad || r.left < data_len) {
        free(offsets);
        free(nbytes);
        kzo_free(net);
        return fail("truncated KZMODEL1 container");
    }
    for (uint32_t i = 0; i < n_tensors; i++) {
        kzo_tensor *t = &net->tensors[i];
        uint64_t esz = t->dtype == 0 ? 4 : 8;
        if (offsets[i] + nbytes[i] > data_len || nbytes[i] != t->count * esz) {
            free(offsets);
            free(nbytes);
            kzo_free(net);
            return fail("tensor table out of range");
        }
        t->data = r.p + offsets[i];
    }
    free(offsets);
    free(nbytes);

    net->h = (int)meta_int(net, "board_h", 0);
    net->w = (int)meta_int(net, "board_w", 0);
    net->n_scalar = (int)meta_int(net, "input_scalar_channels", 0);
    net->n_bool = (int)meta_int(net, "input_bool_channels", 0);
    net->c_in = net->n_scalar + net->n_bool;
    net->depth = (int)meta_int(net, "tower_depth", -1);
    net->channels = (int)meta_int(net, "tower_channels", 0);
    net->final_affine = (int)meta_int(net, "tower_final_affine", 1);
    net->sh_channels = (int)meta_int(net, "scalar_hidden_channels", 4);
    net->sh_size = (int)meta_int(net, "scalar_hidden_size", 32);
    net->policy_len = (int)meta_int(net, "policy_len", 0);
    net->policy_conv_channels = (int)meta_int(net, "policy_conv_channels", 0);
    net->policy_extra_moves = (int)meta_int(net, "policy_extra_moves", 0);
    net->policy_query_channels = (int)meta_int(net, "policy_query_channels", 0);
    net->dense_hidden_channels = (int)meta_int(net, "policy_dense_hidden_channels", 0);
    net->dense_hidden_size = (int)meta_int(net, "policy_dense_hidden_size", 0);
    net->arimaa_hidden_channels = (int)meta_int(net, "policy_arimaa_hidden_channels", 0);
    net->arimaa_hidden_size = (int)meta_int(net, "policy_arimaa_hidden_size", 0);
    const kzo_meta *eps = find_meta(net, "bn_eps");
    net->bn_eps = eps ? (float)(eps->kind == 1 ? eps->f : (double)eps->i) : 1e-5f;
    const kzo_meta *tk = find_meta(net, "tower_kind");
    net->tower_kind = 0;
    if (tk && tk->kind == 2 && !strcmp(tk->s, "attention")) {
        net->tower_kind = 1;
        net->att_heads = (int)meta_int(net, "att_heads", 0);
        net->att_dk = (int)meta_int(net, "att_d_k", 0);
        net->att_dv = (int)meta_int(net, "att_d_v", 0);
        net->att_dff = (int)meta_int(net, "att_d_ff", 0);
        const kzo_meta *al = find_meta(net, "att_alpha"), *le = find_meta(net, "ln_eps");
        net->att_alpha = al ? (float)(al->kind == 1 ? al->f : (double)al->i) : 1.0f;
        net->ln_eps = le && le->kind == 1 ? (float)le->f : 1e-5f;
        if (net->att_heads <= 0 || net->att_dk <= 0 || net->att_dv <= 0 || net->att_dff <= 0) {
            kzo_free(net);
            return fail("bad attention tower descriptor");
        }
    } else if (tk && tk->kind == 2 && !strcmp(tk->s, "dense_network")) {
        net->tower_kind = 2;
        net->dn_res = (int)meta_int(net, "dn_res", 0);
    } else if (tk && !(tk->kind == 2 && !strcmp(tk->s, "res"))) {
        kzo_free(net);
        return fail("unknown tower_kind");
    }
    const kzo_meta *kind = find_meta(net, "policy_kind");
    if (!kind || kind->kind != 2) {
        kzo_free(net);
        return fail("missing policy_kind");
    }
    if (net->tower_kind == 2 && !strcmp(kind->s, "none")) net->policy_kind = -1;
    else if (!strcmp(kind->s, "ataxx_conv")) net->policy_kind = 0;
    else if (!strcmp(kind->s, "conv")) net->policy_kind = 1;
    else if (!strcmp(kind->s, "attention")) net->policy_kind = 2;
    else if (!strcmp(kind->s, "dense")) net->policy_kind = 3;
    else if (!strcmp(kind->s, "arimaa")) net->policy_kind = 4;
    else {
        kzo_free(net);
        return fail("unknown policy_kind");
    }
    if (net->h <= 0 || net->w <= 0 || net->c_in <= 0 || net->depth < 0 || net->channels <= 0 ||
        net->policy_len <= 0) {
        kzo_free(net);
        return fail("bad architecture descriptor");
    }
    *out = net;
    return 0;
}

/* out[0..6] = c_in, h, w, n_scalar, n_bool, policy_len, depth, channels */
KZO_EXPORT int kzo_info(const kzo_net *net, int *out8) {
    out8[0] = net->c_in; out8[1] = net->h; out8[2] = net->w; out8[3] = net->n_scalar;
    out8[4] = net->n_bool; out8[5] = net->policy_len; out8[6] = net->depth; out8[7] = net->channels;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* BitBuffer — rust/kz-core/src/mapping/bit_buffer.rs:4-96                     */
/* ------------------------------------------------------------------------- */

/* bit_buffer.rs:19-36 `push`: bit i lives in byte i/8 at position i%8 (LSB first). */
KZO_EXPORT int kzo_bits_push(uint8_t *storage, size_t capacity, size_t *len, int b) {
    if (*len >= capacity) return fail("BitBuffer: not enough space left");
    size_t index = *len / 8, bit = *len % 8;
    *len += 1;
    if (b) storage[index] |= (uint8_t)(1u << bit);
    else storage[index] &= (uint8_t)~(1u << bit);
    return 0;
}

/* bit_buffer.rs:38-57 `push_block`: 64 bits as 8 little-endian bytes, byte aligned only. */
KZO_EXPORT int kzo_bits_push_block(uint8_t *storage, size_t capacity, size_t *len, uint64_t block) {
    if (*len + 64 > capacity) return fail("BitBuffer: not enough space left for block");
    if (*len % 8 != 0) return fail("BitBuffer: can only push aligned blocks");
    size_t index = *len / 8;
    for (int i = 0; i < 8; i++) storage[index + i] = (uint8_t)(block >> (8 * i));
    *len += 64;
    return 0;
}

/* bit_buffer.rs:80-90 `index`. */
KZO_EXPORT int kzo_bits_get(const uint8_t *storage, size_t i) { return (storage[i / 8] >> (i % 8)) & 1; }

/* bit_buffer.rs:8-14: storage bytes for a capacity. */
KZO_EXPORT size_t kzo_bits_storage_len(size_t capacity) { return (capacity - 1) / 8 + 1; }

/* ------------------------------------------------------------------------- */
/* encode_input_full — rust/kz-core/src/mapping/mod.rs:40-63                   */
/* scalar planes first (each scalar broadcast over H*W, :54-56), then the     */
/* bool planes as 0.0/1.0 (:57-59); bool index = c*H*W + y*W + x.             */
/* ------------------------------------------------------------------------- */
KZO_EXPORT void kzo_encode_input_full(const uint8_t *bits, size_t bits_stride, const float *scalars, int batch,
                                      int n_scalar, int n_bool, int h, int w, float *out) {
    size_t hw = (size_t)h * w;
    size_t bool_count = (size_t)n_bool * hw;
    size_t full = (n_scalar + n_bool) * hw;
    for (int b = 0; b < batch; b++) {
        float *dst = out + (size_t)b * full;
        for (int s = 0; s < n_scalar; s++)
            for (size_t i = 0; i < hw; i++) *dst++ = scalars[(size_t)b * n_scalar + s];
        const uint8_t *bb = bits + (size_t)b * bits_stride;
        for (size_t i = 0; i < bool_count; i++) *dst++ = (float)kzo_bits_get(bb, i);
    }
}

/* ------------------------------------------------------------------------- */
/* Network arithmetic — python/lib/model/post_act.py                           */
/* ------------------------------------------------------------------------- */

/* conv2d(): post_act.py:231-239 — nn.Conv2d, square odd kernel k, padding k/2, bias. x [cin,h,w] -> y [cout,h,w] */
static void conv2d(const float *x, int cin, int h, int w, const float *wt, const float *bias, int cout, int k,
                   float *y) {
    /* Same sums in the same order as the plain loop nest (per output: bias, then input channel by input channel, tap by
     * tap), but over a ZERO-PADDED copy of the input so that the innermost loop runs over a whole plane at once — h * (w + 2)
     * contiguous floats instead of one board line (8 on a chess board), which the compiler vectorises.  A tap outside the
     * board adds w * 0.0f where the plain nest added nothing: the same value.  (Round 5: the CPU baseline of bench.py is
     * this code; the plain nest ran at ~10 GFLOP/s per core.) */
    const int pad = k / 2, w2 = w + 2 * pad, h2 = h + 2 * pad;
    const size_t plane = (size_t)h2 * w2, span = (size_t)(h - 1) * w2 + w;
    float *xp = calloc((size_t)cin * plane, sizeof(float));
    float *yp = malloc(sizeof(float) * (size_t)h * w2);
    for (int ic = 0; ic < cin; ic++)
        for (int yy = 0; yy < h; yy++)
            memcpy(xp + (size_t)ic * plane + (size_t)(yy + pad) * w2 + pad, x + ((size_t)ic * h + yy) * w, sizeof(float) * (size_t)w);
    int oc = 0;
    if (k == 3) {
        /* two output channels per pass share the nine loads of the input plane (each output's own sum is unchanged) */
        float *yq = malloc(sizeof(float) * (size_t)h * w2);
        for (; oc + 1 < cout; oc += 2) {
            for (size_t i = 0; i < span; i++) {
                yp[i] = bias[oc];
                yq[i] = bias[oc + 1];
            }
            for (int ic = 0; ic < cin; ic++) {
                const float *r0 = xp + (size_t)ic * plane, *r1 = r0 + w2, *r2 = r0 + 2 * (size_t)w2;
                const float *a = wt + ((size_t)oc * cin + ic) * 9, *b = wt + ((size_t)(oc + 1) * cin + ic) * 9;
                const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6], a7 = a[7], a8 = a[8];
                const float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4], b5 = b[5], b6 = b[6], b7 = b[7], b8 = b[8];
                for (size_t i = 0; i < span; i++) {
                    const float x0 = r0[i], x1 = r0[i + 1], x2 = r0[i + 2], x3 = r1[i], x4 = r1[i + 1], x5 = r1[i + 2],
                                x6 = r2[i], x7 = r2[i + 1], x8 = r2[i + 2];
                    float p = yp[i], q = yq[i];
                    p += a0 * x0; q += b0 * x0;
                    p += a1 * x1; q += b1 * x1;
                    p += a2 * x2; q += b2 * x2;
                    p += a3 * x3; q += b3 * x3;
                    p += a4 * x4; q += b4 * x4;
                    p += a5 * x5; q += b5 * x5;
                    p += a6 * x6; q += b6 * x6;
                    p += a7 * x7; q += b7 * x7;
                    p += a8 * x8; q += b8 * x8;
                    yp[i] = p;
                    yq[i] = q;
                }
            }
            for (int yy = 0; yy < h; yy++) {
                memcpy(y + ((size_t)oc * h + yy) * w, yp + (size_t)yy * w2, sizeof(float) * (size_t)w);
                memcpy(y + ((size_t)(oc + 1) * h + yy) * w, yq + (size_t)yy * w2, sizeof(float) * (size_t)w);
            }
        }
        free(yq);
    }
    for (; oc < cout; oc++) {
        for (size_t i = 0; i < span; i++) yp[i] = bias[oc];
        for (int ic = 0; ic < cin; ic++) {
            const float *xi = xp + (size_t)ic * plane;
            const float *wk = wt + ((size_t)oc * cin + ic) * k * k;
            if (k == 3) {
                /* the nine taps of one input channel in one pass over the plane: the running sum stays in a register from tap
                 * to tap (same additions in the same order, one load and one store of the sum instead of nine) */
                const float w0 = wk[0], w1 = wk[1], w2_ = wk[2], w3 = wk[3], w4 = wk[4], w5 = wk[5], w6 = wk[6], w7 = wk[7], w8 = wk[8];
                const float *r0 = xi, *r1 = xi + w2, *r2 = xi + 2 * (size_t)w2;
                for (size_t i = 0; i < span; i++) {
                    float acc = yp[i];
                    acc += w0 * r0[i];
                    acc += w1 * r0[i + 1];
                    acc += w2_ * r0[i + 2];
                    acc += w3 * r1[i];
                    acc += w4 * r1[i + 1];
                    acc += w5 * r1[i + 2];
                    acc += w6 * r2[i];
                    acc += w7 * r2[i + 1];
                    acc += w8 * r2[i + 2];
                    yp[i] = acc;
                }
                continue;
            }
            for (int ky = 0; ky < k; ky++)
                for (int kx = 0; kx < k; kx++) {
                    const float wv = wk[ky * k + kx];
                    const float *xs = xi + (size_t)ky * w2 + kx;
                    for (size_t i = 0; i < span; i++) yp[i] += wv * xs[i];
                }
        }
        float *yo = y + (size_t)oc * h * w;
        for (int yy = 0; yy < h; yy++) memcpy(yo + (size_t)yy * w, yp + (size_t)yy * w2, sizeof(float) * (size_t)w);
    }
    free(xp);
    free(yp);
}

/* nn.BatchNorm2d in eval mode (network.eval(), python/lib/save_onnx.py:82): running stats, eps 1e-5. In place. */
static void batchnorm_eval(float *x, int c, int hw, const float *weight, const float *bias, const float *mean,
                           const float *var, float eps) {
    for (int ch = 0; ch < c; ch++) {
        float inv = 1.0f / sqrtf(var[ch] + eps);
        float g = weight ? weight[ch] : 1.0f, b = bias ? bias[ch] : 0.0f;
        float *p = x + (size_t)ch * hw;
        for (int i = 0; i < hw; i++) p[i] = (p[i] - mean[ch]) * inv * g + b;
    }
}

static void relu(float *x, size_t n) {
    for (size_t i = 0; i < n; i++) x[i] = x[i] > 0.0f ? x[i] : 0.0f;
}

/* nn.Linear: y = W x + b, W [out, in] */
static void linear(const float *x, int in, const float *wt, const float *bias, int out, float *y) {
    for (int o = 0; o < out; o++) {
        float acc = bias[o];
        const float *wr = wt + (size_t)o * in;
        for (int i = 0; i < in; i++) acc += wr[i] * x[i];
        y[o] = acc;
    }
}

typedef void (*kzo_trace_fn)(const char *name, const float *data, int count, int board, void *user);

typedef struct {
    const float *w, *b;
} convp;

typedef struct {
    const float *weight, *bias, *mean, *var;
} bnp;

static int get_conv(const kzo_net *net, const char *prefix, int cout, int cin, int k, convp *c) {
    char name[128];
    snprintf(name, sizeof name, "%s.weight", prefix);
    c->w = tensor_f32(net, name, (uint64_t)cout * cin * k * k);
    snprintf(name, sizeof name, "%s.bias", prefix);
    c->b = c->w ? tensor_f32(net, name, (uint64_t)cout) : NULL;
    return (c->w && c->b) ? 0 : -1;
}

static int get_bn(const kzo_net *net, const char *prefix, int c, int affine, bnp *bn) {
    char name[128];
    bn->weight = bn->bias = NULL;
    if (affine) {
        snprintf(name, sizeof name, "%s.weight", prefix);
        bn->weight = tensor_f32(net, name, (uint64_t)c);
        snprintf(name, sizeof name, "%s.bias", prefix);
        bn->bias = tensor_f32(net, name, (uint64_t)c);
        if (!bn->weight || !bn->bias) return -1;
    }
    snprintf(name, sizeof name, "%s.running_mean", prefix);
    bn->mean = tensor_f32(net, name, (uint64_t)c);
    snprintf(name, sizeof name, "%s.running_var", prefix);
    bn->var = tensor_f32(net, name, (uint64_t)c);
    return (bn->mean && bn->var) ? 0 : -1;
}

/* y[r][o] = sum_i x[r][i] * wt[o][i]: nn.Linear(bias=False) on `rows` rows */
static void linear_rows(const float *x, int rows, int in, const float *wt, int out, float *y) {
    for (int r = 0; r < rows; r++)
        for (int o = 0; o < out; o++) {
            float acc = 0.0f;
            for (int i = 0; i < in; i++) acc += x[(size_t)r * in + i] * wt[(size_t)o * in + i];
            y[(size_t)r * out + o] = acc;
        }
}

/* nn.LayerNorm(d, elementwise_affine=False) over each row: (x - mean) / sqrt(biased var + eps) */
static void layernorm_rows(float *x, int rows, int d, float eps) {
    for (int r = 0; r < rows; r++) {
        float *v = x + (size_t)r * d;
        float mean = 0.0f, var = 0.0f;
        for (int i = 0; i < d; i++) mean += v[i];
        mean /= (float)d;
        for (int i = 0; i < d; i++) var += (v[i] - mean) * (v[i] - mean);
        var /= (float)d;
        const float inv = 1.0f / sqrtf(var + eps);
        for (int i = 0; i < d; i++) v[i] = (v[i] - mean) * inv;
    }
}

/* AttentionTower.forward (python/lib/model/attention.py:32-45) for one board: input [c_in][hw] -> x_out [d_model][hw].
 * Every square is a token; the encoder layers are EncoderLayer.forward_with_weights (:97-133) in eval mode (dropout = id). */
static int attention_tower(const kzo_net *net, const float *input, float *x_out, int board, kzo_trace_fn trace, void *user) {
    const int n = net->h * net->w, D = net->channels, H = net->att_heads, dk = net->att_dk, dv = net->att_dv;
    const int dff = net->att_dff, dkqv = 2 * dk + dv, cin = net->c_in;
    const float alpha = net->att_alpha;
    char name[128];
    int rc = -1;
    const float *expand = tensor_f32(net, "common.expand.weight", (uint64_t)D * cin);
    const float *embedding = tensor_f32(net, "common.embedding", (uint64_t)n * D);
    if (!expand || !embedding) return -1;
    float *cur = malloc(sizeof(float) * (size_t)n * D);
    float *qkv = malloc(sizeof(float) * (size_t)n * H * dkqv);
    float *att = malloc(sizeof(float) * (size_t)n * H * dv);
    float *tmp = malloc(sizeof(float) * (size_t)n * (D > dff ? D : dff));
    float *mid = malloc(sizeof(float) * (size_t)n * D);
    float *wrow = malloc(sizeof(float) * (size_t)n);

    /* "b c h w -> (h w) b c", expand (Linear, no bias) + embedding (:35-40) */
    for (int p = 0; p < n; p++)
        for (int c = 0; c < D; c++) {
            float acc = 0.0f;
            for (int i = 0; i < cin; i++) acc += input[(size_t)i * n + p] * expand[(size_t)c * cin + i];
            cur[(size_t)p * D + c] = acc + embedding[(size_t)p * D + c];
        }

    for (int l = 0; l < net->depth; l++) {
        snprintf(name, sizeof name, "common.encoders.%d.project_qkv.weight", l);
        const float *wqkv = tensor_f32(net, name, (uint64_t)H * dkqv * D);
        snprintf(name, sizeof name, "common.encoders.%d.project_out.weight", l);
        const float *wout = tensor_f32(net, name, (uint64_t)D * H * dv);
        snprintf(name, sizeof name, "common.encoders.%d.ff.0.weight", l);
        const float *wf0 = tensor_f32(net, name, (uint64_t)dff * D);
        snprintf(name, sizeof name, "common.encoders.%d.ff.2.weight", l);
        const float *wf2 = tensor_f32(net, name, (uint64_t)D * dff);
        if (!wqkv || !wout || !wf0 || !wf2) goto done;

        /* qkv = project_qkv(input).view(n, b * heads, d_kqv) (:106): token p's row holds, head after head, q | k | v */
        linear_rows(cur, n, D, wqkv, H * dkqv, qkv);
        for (int hh = 0; hh < H; hh++) {
            for (int pq = 0; pq < n; pq++) {
                const float *q = qkv + (size_t)pq * H * dkqv + (size_t)hh * dkqv;
                /* logits = q k^T with no scale factor (:117-118), softmax over the keys (:119) */
                float mx = -INFINITY;
                for (int pk = 0; pk < n; pk++) {
                    const float *k = qkv + (size_t)pk * H * dkqv + (size_t)hh * dkqv + dk;
                    float acc = 0.0f;
                    for (int j = 0; j < dk; j++) acc += q[j] * k[j];
                    wrow[pk] = acc;
                    if (acc > mx) mx = acc;
                }
                float sum = 0.0f;
                for (int pk = 0; pk < n; pk++) {
                    wrow[pk] = expf(wrow[pk] - mx);
                    sum += wrow[pk];
                }
                /* att_raw = weights v (:122), heads side by side in the token's row (:124) */
                for (int j = 0; j < dv; j++) {
                    float acc = 0.0f;
                    for (int pk = 0; pk < n; pk++)
                        acc += (wrow[pk] / sum) * qkv[(size_t)pk * H * dkqv + (size_t)hh * dkqv + 2 * dk + j];
                    att[(size_t)pq * H * dv + (size_t)hh * dv + j] = acc;
                }
            }
        }
        /* att_result = norm_att(input * alpha + project_out(att)) (:125-126) */
        linear_rows(att, n, H * dv, wout, D, tmp);
        for (int i = 0; i < n * D; i++) mid[i] = cur[i] * alpha + tmp[i];
        layernorm_rows(mid, n, D, net->ln_eps);
        /* ff_result = norm_ff(att_result * alpha + ff(att_result)) (:128-129), ff = Linear, ReLU, Linear without biases (:74-78) */
        linear_rows(mid, n, D, wf0, dff, tmp);
        relu(tmp, (size_t)n * dff);
        linear_rows(tmp, n, dff, wf2, D, cur);
        for (int i = 0; i < n * D; i++) cur[i] = mid[i] * alpha + cur[i];
        layernorm_rows(cur, n, D, net->ln_eps);
        if (trace) {
            snprintf(name, sizeof name, "encoder.%d", l);
            trace(name, cur, n * D, board, user);
        }
    }
    /* "(h w) b c -> b c h w" (:43-44) */
    for (int p = 0; p < n; p++)
        for (int c = 0; c < D; c++) x_out[(size_t)c * n + p] = cur[(size_t)p * D + c];
    rc = 0;
done:
    free(cur); free(qkv); free(att); free(tmp); free(mid); free(wrow);
    return rc;
}

/* nn.BatchNorm1d in eval mode followed by ReLU, in place on a vector */
static int bn1d_relu(const kzo_net *net, const char *prefix, float *v, int n) {
    bnp bn;
    if (get_bn(net, prefix, n, 1, &bn)) return -1;
    for (int i = 0; i < n; i++) {
        const float y = (v[i] - bn.mean[i]) / sqrtf(bn.var[i] + net->bn_eps) * bn.weight[i] + bn.bias[i];
        v[i] = y > 0.0f ? y : 0.0f;
    }
    return 0;
}

static int get_linear(const kzo_net *net, const char *prefix, int out, int in, const float **w, const float **b) {
    char name[128];
    snprintf(name, sizeof name, "%s.weight", prefix);
    *w = tensor_f32(net, name, (uint64_t)out * in);
    snprintf(name, sizeof name, "%s.bias", prefix);
    *b = tensor_f32(net, name, (uint64_t)out);
    return (*w && *b) ? 0 : -1;
}

/* DenseNetwork.forward (python/lib/model/simple.py:26-33) for one board: Flatten (channel-major, as the NCHW input lies),
 * Linear, `depth` DenseBlocks (:36-52: BatchNorm1d, ReLU, Linear, BatchNorm1d, ReLU, Linear; x + y when res), BatchNorm1d, ReLU,
 * Linear; scalars = output[:5], policy = output[5:]. */
static int dense_network(const kzo_net *net, const float *input, float *scalars_out, float *policy_out) {
    const int in = net->c_in * net->h * net->w, size = net->channels, outs = 5 + net->policy_len;
    char name[128];
    const float *w, *b;
    int rc = -1;
    float *cur = malloc(sizeof(float) * (size_t)size), *a = malloc(sizeof(float) * (size_t)size), *hmid = malloc(sizeof(float) * (size_t)size);
    float *out = malloc(sizeof(float) * (size_t)outs);
    if (get_linear(net, "seq.1", size, in, &w, &b)) goto done;
    linear(input, in, w, b, size, cur);
    for (int i = 0; i < net->depth; i++) {
        memcpy(a, cur, sizeof(float) * (size_t)size);
        snprintf(name, sizeof name, "seq.%d.seq.0", 2 + i);
        if (bn1d_relu(net, name, a, size)) goto done;
        snprintf(name, sizeof name, "seq.%d.seq.2", 2 + i);
        if (get_linear(net, name, size, size, &w, &b)) goto done;
        linear(a, size, w, b, size, hmid);
        snprintf(name, sizeof name, "seq.%d.seq.3", 2 + i);
        if (bn1d_relu(net, name, hmid, size)) goto done;
        snprintf(name, sizeof name, "seq.%d.seq.5", 2 + i);
        if (get_linear(net, name, size, size, &w, &b)) goto done;
        linear(hmid, size, w, b, size, a);
        for (int j = 0; j < size; j++) cur[j] = net->dn_res ? cur[j] + a[j] : a[j];
    }
    snprintf(name, sizeof name, "seq.%d", 2 + net->depth);
    if (bn1d_relu(net, name, cur, size)) goto done;
    snprintf(name, sizeof name, "seq.%d", 4 + net->depth);
    if (get_linear(net, name, outs, size, &w, &b)) goto done;
    linear(cur, size, w, b, outs, out);
    memcpy(scalars_out, out, sizeof(float) * 5);
    memcpy(policy_out, out + 5, sizeof(float) * (size_t)net->policy_len);
    rc = 0;
done:
    free(cur); free(a); free(hmid); free(out);
    return rc;
}

/* One board through PredictionHeads.forward (post_act.py:194-198). Returns 0 or -1. */
static int forward_board(const kzo_net *net, const float *input, float *scalars_out, float *policy_out, int board,
                         kzo_trace_fn trace, void *user) {
    const int h = net->h, w = net->w, hw = h * w, C = net->channels;
    char name[128];
    int rc = -1;
    if (net->tower_kind == 2) return dense_network(net, input, scalars_out, policy_out);
    float *x = malloc(sizeof(float) * (size_t)C * hw);
    float *t0 = malloc(sizeof(float) * (size_t)C * hw);
    float *t1 = malloc(sizeof(float) * (size_t)C * hw);
    float *head = NULL;

    if (net->tower_kind == 1) {
        if (attention_tower(net, input, x, board, trace, user)) goto done;
        goto heads;
    }
    /* ResTower (post_act.py:201-211): stem conv with no BN and no ReLU (:205) */
    convp stem;
    if (get_conv(net, "common.tower.0", C, net->c_in, 3, &stem)) goto done;
    conv2d(input, net->c_in, h, w, stem.w, stem.b, C, 3, x);
    if (trace) trace("tower.0", x, C * hw, board, user);

    /* ResBlock (post_act.py:214-228): input + relu(bn(conv(relu(bn(conv(input)))))) */
    for (int i = 1; i <= net->depth; i++) {
        convp ca, cb;
        bnp ba, bb;
        snprintf(name, sizeof name, "common.tower.%d.seq.0", i);
        if (get_conv(net, name, C, C, 3, &ca)) goto done;
        snprintf(name, sizeof name, "common.tower.%d.seq.1", i);
        if (get_bn(net, name, C, 1, &ba)) goto done;
        snprintf(name, sizeof name, "common.tower.%d.seq.3", i);
        if (get_conv(net, name, C, C, 3, &cb)) goto done;
        snprintf(name, sizeof name, "common.tower.%d.seq.4", i);
        if (get_bn(net, name, C, 1, &bb)) goto done;

        conv2d(x, C, h, w, ca.w, ca.b, C, 3, t0);
        batchnorm_eval(t0, C, hw, ba.weight, ba.bias, ba.mean, ba.var, net->bn_eps);
        relu(t0, (size_t)C * hw);
        if (trace) {
            snprintf(name, sizeof name, "tower.%d.mid", i);
            trace(name, t0, C * hw, board, user);
        }
        conv2d(t0, C, h, w, cb.w, cb.b, C, 3, t1);
        batchnorm_eval(t1, C, hw, bb.weight, bb.bias, bb.mean, bb.var, net->bn_eps);
        relu(t1, (size_t)C * hw);
        for (int j = 0; j < C * hw; j++) x[j] = x[j] + t1[j]; /* residual AFTER the ReLU (:227-228) */
        if (trace) {
            snprintf(name, sizeof name, "tower.%d", i);
            trace(name, x, C * hw, board, user);
        }
    }

    /* final BatchNorm2d(channels, affine=final_affine) (post_act.py:207) */
    {
        bnp bf;
        snprintf(name, sizeof name, "common.tower.%d", net->depth + 1);
        if (get_bn(net, name, C, net->final_affine, &bf)) goto done;
        batchnorm_eval(x, C, hw, bf.weight, bf.bias, bf.mean, bf.var, net->bn_eps);
        if (trace) {
            snprintf(name, sizeof name, "tower.%d", net->depth + 1);
            trace(name, x, C * hw, board, user);
        }
    }

heads:
    /* ScalarHead (post_act.py:10-23): conv1x1 -> ReLU -> Flatten (channel-major) -> Linear -> ReLU -> Linear(5) */
    {
        const int hc = net->sh_channels, hs = net->sh_size;
        convp c0;
        if (get_conv(net, "scalar_head.seq.0", hc, C, 1, &c0)) goto done;
        const float *w3 = tensor_f32(net, "scalar_head.seq.3.weight", (uint64_t)hs * hc * hw);
        const float *b3 = tensor_f32(net, "scalar_head.seq.3.bias", (uint64_t)hs);
        const float *w5 = tensor_f32(net, "scalar_head.seq.5.weight", (uint64_t)5 * hs);
        const float *b5 = tensor_f32(net, "scalar_head.seq.5.bias", 5);
        if (!w3 || !b3 || !w5 || !b5) goto done;
        float *a = malloc(sizeof(float) * (size_t)hc * hw);
        float *hid = malloc(sizeof(float) * (size_t)hs);
        conv2d(x, C, h, w, c0.w, c0.b, hc, 1, a);
        relu(a, (size_t)hc * hw);
        if (trace) trace("scalar_head.conv_relu", a, hc * hw, board, user);
        linear(a, hc * hw, w3, b3, hs, hid);
        relu(hid, (size_t)hs);
        if (trace) trace("scalar_head.fc0_relu", hid, hs, board, user);
        linear(hid, hs, w5, b5, 5, scalars_out);
        free(a);
        free(hid);
    }

    /* policy heads */
    if (net->policy_kind == 0) {
        /* AtaxxConvPolicyHead (post_act.py:91-112): conv1x1 C->C, ReLU, conv1x1 C->17; flatten(1) ‖ one zero column */
        const int pc = net->policy_conv_channels;
        convp c0, c2;
        if (pc * hw + 1 != net->policy_len) { fail("ataxx head: policy_len mismatch"); goto done; }
        if (get_conv(net, "policy_head.seq.0", C, C, 1, &c0)) goto done;
        if (get_conv(net, "policy_head.seq.2", pc, C, 1, &c2)) goto done;
        conv2d(x, C, h, w, c0.w, c0.b, C, 1, t0);
        relu(t0, (size_t)C * hw);
        conv2d(t0, C, h, w, c2.w, c2.b, pc, 1, policy_out);
        policy_out[pc * hw] = 0.0f;
    } else if (net->policy_kind == 1) {
        /* ConvPolicyHead (post_act.py:54-88), non-chess branch: concat([seq(common).flatten(1), seq_extra(common)]) */
        const int pc = net->policy_conv_channels, extra = net->policy_extra_moves;
        convp c0, c2;
        if (pc * hw + extra != net->policy_len) { fail("conv head: policy_len mismatch"); goto done; }
        if (get_conv(net, "policy_head.seq.0", C, C, 1, &c0)) goto done;
        if (get_conv(net, "policy_head.seq.2", pc, C, 1, &c2)) goto done;
        conv2d(x, C, h, w, c0.w, c0.b, C, 1, t0);
        relu(t0, (size_t)C * hw);
        conv2d(t0, C, h, w, c2.w, c2.b, pc, 1, policy_out);
        if (extra != 0) {
            convp ce;
            if (get_conv(net, "policy_head.seq_extra.0", 1, C, 1, &ce)) goto done;
            const float *we = tensor_f32(net, "policy_head.seq_extra.2.weight", (uint64_t)extra * hw);
            const float *be = tensor_f32(net, "policy_head.seq_extra.2.bias", (uint64_t)extra);
            if (!we || !be) goto done;
            conv2d(x, C, h, w, ce.w, ce.b, 1, 1, t1);
            linear(t1, hw, we, be, extra, policy_out + pc * hw);
        }
    } else if (net->policy_kind == 2) {
        /* AttentionPolicyHead (post_act.py:115-141) */
        const int Q = net->policy_query_channels;
        if (h != 8 || w != 8) { fail("attention head needs an 8x8 board"); goto done; }
        convp cbulk, cunder;
        if (get_conv(net, "policy_head.conv_bulk", 2 * Q, C, 1, &cbulk)) goto done;
        if (get_conv(net, "policy_head.conv_under", 3 * Q, C, 1, &cunder)) goto done;
        const kzo_tensor *tab = find_tensor(net, "policy_head.FLAT_TO_ATT");
        if (!tab || tab->dtype != 1 || (int)tab->count != net->policy_len) { fail("missing FLAT_TO_ATT"); goto done; }
        const int64_t *flat_to_att = tab->data;
        head = malloc(sizeof(float) * ((size_t)2 * Q * 64 + (size_t)3 * Q * 8 + (size_t)C * 8 + (size_t)Q * 88 + 64 * 88));
        float *bulk = head;                 /* [2Q, 8, 8] */
        float *under = bulk + 2 * Q * 64;   /* [3Q, 1, 8] */
        float *row7 = under + 3 * Q * 8;    /* common[:, :, 7, None, :] -> [C, 1, 8] */
        float *q_to = row7 + C * 8;         /* [Q, 88] */
        float *att = q_to + Q * 88;         /* [64, 88] */
        conv2d(x, C, 8, 8, cbulk.w, cbulk.b, 2 * Q, 1, bulk);
        for (int c = 0; c < C; c++)
            for (int xx = 0; xx < 8; xx++) row7[c * 8 + xx] = x[(size_t)c * 64 + 7 * 8 + xx];
        conv2d(row7, C, 1, 8, cunder.w, cunder.b, 3 * Q, 1, under);
        /* q_from = bulk[:, :Q].flatten(2) [Q,64]; q_to = cat(bulk[:, Q:].flatten(2) [Q,64], under.reshape(Q, 24)) */
        const float *q_from = bulk;
        for (int q = 0; q < Q; q++) {
            for (int i = 0; i < 64; i++) q_to[q * 88 + i] = bulk[(size_t)(Q + q) * 64 + i];
            /* under is [3Q,1,8] contiguous; reshape(-1, Q, 24): row q takes flat elements q*24 .. q*24+23 */
            for (int i = 0; i < 24; i++) q_to[q * 88 + 64 + i] = under[(size_t)q * 24 + i];
        }
        /* policy = bmm(q_from^T [64,Q], q_to [Q,88]) / sqrt(Q) */
        float scale = (float)pow((double)Q, 0.5);
        for (int i = 0; i < 64; i++)
            for (int j = 0; j < 88; j++) {
                float acc = 0.0f;
                for (int q = 0; q < Q; q++) acc += q_from[q * 64 + i] * q_to[q * 88 + j];
                att[i * 88 + j] = acc / scale;
            }
        for (int i = 0; i < net->policy_len; i++) policy_out[i] = att[flat_to_att[i]];
    } else if (net->policy_kind == 4) {
        /* ArimaaPolicyHead (post_act.py:144-173): policy = concat([scalar(common) [1 + 6], flatten(bulk(common), 1) [4*hw]])
         * bulk = conv1x1 C->C, ReLU, conv1x1 C->4;  scalar = conv1x1 C->hc, ReLU, Flatten, Linear(hc*hw -> hs), ReLU, Linear(hs -> 7) */
        const int hc = net->arimaa_hidden_channels, hs = net->arimaa_hidden_size;
        convp b0, b2, s0;
        if (hc <= 0 || hs <= 0 || 7 + 4 * hw != net->policy_len) { fail("arimaa head: bad descriptor"); goto done; }
        if (get_conv(net, "policy_head.bulk.0", C, C, 1, &b0)) goto done;
        if (get_conv(net, "policy_head.bulk.2", 4, C, 1, &b2)) goto done;
        if (get_conv(net, "policy_head.scalar.0", hc, C, 1, &s0)) goto done;
        const float *w3 = tensor_f32(net, "policy_head.scalar.3.weight", (uint64_t)hs * hc * hw);
        const float *b3 = tensor_f32(net, "policy_head.scalar.3.bias", (uint64_t)hs);
        const float *w5 = tensor_f32(net, "policy_head.scalar.5.weight", (uint64_t)7 * hs);
        const float *b5 = tensor_f32(net, "policy_head.scalar.5.bias", 7);
        if (!w3 || !b3 || !w5 || !b5) goto done;
        conv2d(x, C, h, w, b0.w, b0.b, C, 1, t0);
        relu(t0, (size_t)C * hw);
        conv2d(t0, C, h, w, b2.w, b2.b, 4, 1, policy_out + 7);
        head = malloc(sizeof(float) * ((size_t)hc * hw + (size_t)hs));
        float *a = head, *hid = head + (size_t)hc * hw;
        conv2d(x, C, h, w, s0.w, s0.b, hc, 1, a);
        relu(a, (size_t)hc * hw);
        linear(a, hc * hw, w3, b3, hs, hid);
        relu(hid, (size_t)hs);
        linear(hid, hs, w5, b5, 7, policy_out);
    } else {
        /* DensePolicyHead (post_act.py:26-51): [conv1x1 + ReLU] -> Flatten -> [Linear + ReLU] -> Linear(policy_size) */
        const int hc = net->dense_hidden_channels, hs = net->dense_hidden_size;
        const float *cur = x;
        int cur_c = C, idx = 0;
        if (hc) {
            convp c0;
            if (get_conv(net, "policy_head.seq.0", hc, C, 1, &c0)) goto done;
            conv2d(x, C, h, w, c0.w, c0.b, hc, 1, t0);
            relu(t0, (size_t)hc * hw);
            cur = t0;
            cur_c = hc;
            idx = 2; /* conv, relu */
        }
        idx += 1; /* flatten */
        int size = cur_c * hw;
        if (hs) {
            snprintf(name, sizeof name, "policy_head.seq.%d.weight", idx);
            const float *w0 = tensor_f32(net, name, (uint64_t)hs * size);
            snprintf(name, sizeof name, "policy_head.seq.%d.bias", idx);
            const float *b0 = tensor_f32(net, name, (uint64_t)hs);
            if (!w0 || !b0) goto done;
            linear(cur, size, w0, b0, hs, t1);
            relu(t1, (size_t)hs);
            cur = t1;
            size = hs;
            idx += 2;
        }
        snprintf(name, sizeof name, "policy_head.seq.%d.weight", idx);
        const float *w1 = tensor_f32(net, name, (uint64_t)net->policy_len * size);
        snprintf(name, sizeof name, "policy_head.seq.%d.bias", idx);
        const float *b1 = tensor_f32(net, name, (uint64_t)net->policy_len);
        if (!w1 || !b1) goto done;
        linear(cur, size, w1, b1, net->policy_len, policy_out);
    }
    rc = 0;
done:
    free(x);
    free(t0);
    free(t1);
    free(head);
    return rc;
}

/*
 * The CPU path: restates CPUNetwork::evaluate_batch_exec's call
 * cpu_eval_graph_exec(&graph, batch, &[input], keep_all) (rust/kz-core/src/network/cpu.rs:33-51)
 * for the PredictionHeads graph.  input NCHW f32 [batch, c_in, h, w]; scalars [batch,5]; policy [batch,P].
 * threads <= 1: serial; otherwise OpenMP over boards (the boards are independent in eval mode).
 */
KZO_EXPORT int kzo_forward(const kzo_net *net, const float *input, int batch, float *scalars, float *policy,
                           int threads) {
    size_t in_stride = (size_t)net->c_in * net->h * net->w;
    int bad = 0;
#ifdef _OPENMP
    int nt = threads > 1 ? threads : 1;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
#endif
    for (int b = 0; b < batch; b++) {
        if (forward_board(net, input + b * in_stride, scalars + (size_t)b * 5, policy + (size_t)b * net->policy_len, b,
                          NULL, NULL))
            bad = 1;
    }
    (void)threads;
    return bad ? -1 : 0;
}

KZO_EXPORT int kzo_forward_trace(const kzo_net *net, const float *input, int batch, float *scalars, float *policy,
                                 kzo_trace_fn trace, void *user) {
    size_t in_stride = (size_t)net->c_in * net->h * net->w;
    for (int b = 0; b < batch; b++)
        if (forward_board(net, input + b * in_stride, scalars + (size_t)b * 5, policy + (size_t)b * net->policy_len, b,
                          trace, user))
            return -1;
    return 0;
}

KZO_EXPORT int kzo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* decode_output — rust/kz-core/src/network/common.rs:16-114                    */
/* ------------------------------------------------------------------------- */

/* softmax_in_place (common.rs:102-114). Returns -1 where the reference asserts (sum must be > 0). */
KZO_EXPORT int kzo_softmax_in_place(float *v, int n) {
    float max = -INFINITY;
    for (int i = 0; i < n; i++) max = v[i] > max ? v[i] : max; /* fold(NEG_INFINITY, max) */
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        v[i] = expf(v[i] - max);
        sum += v[i];
    }
    if (!(sum > 0.0f)) return fail("Softmax input sum must be strictly positive");
    for (int i = 0; i < n; i++) v[i] /= sum;
    return 0;
}

/*
 * decode_output for the 2-output form (common.rs:31-42, :52-98): per board value = tanh(s0) (:60),
 * wdl = softmax(s1..s3) (:64-69), moves_left = s4 (:61); policy = softmax over the logits gathered at
 * move_to_index(mv) for each available move, in available_moves() order (:77-86).  The caller supplies the
 * per-board index lists (CSR: move_offsets[batch+1], move_indices) because move generation lives in the ext
 * board-game crate.  A board with no available moves yields an empty policy (`map_or(vec![], ..)`, :77).
 * values_out [batch,5] = value, win, draw, loss, moves_left; policy_out parallel to move_indices.
 */
KZO_EXPORT int kzo_decode_output(const float *scalars, const float *policy_logits, int batch, int policy_len,
                                 const int64_t *move_offsets, const int32_t *move_indices, float *values_out,
                                 float *policy_out) {
    for (int b = 0; b < batch; b++) {
        const float *s = scalars + (size_t)b * 5;
        float wdl[3] = {s[1], s[2], s[3]};
        if (kzo_softmax_in_place(wdl, 3)) return -1;
        float *vo = values_out + (size_t)b * 5;
        vo[0] = tanhf(s[0]);
        vo[1] = wdl[0];
        vo[2] = wdl[1];
        vo[3] = wdl[2];
        vo[4] = s[4];
        int64_t lo = move_offsets[b], hi = move_offsets[b + 1];
        if (hi == lo) continue;
        for (int64_t i = lo; i < hi; i++) {
            int32_t idx = move_indices[i];
            if (idx < 0 || idx >= policy_len) return fail("policy index out of range");
            policy_out[i] = policy_logits[(size_t)b * policy_len + idx];
        }
        if (kzo_softmax_in_place(policy_out + lo, (int)(hi - lo))) return -1;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Chess policy indexing (rust/kz-core/src/mapping/chess.rs:180-507): what decode_output calls for every available move.
 * A move is (from, to, promotion): squares are rank * 8 + file with A1 = 0; promotion 0 = none, 1 = queen, 2 = rook,
 * 3 = bishop, 4 = knight (the order of the flat list's promotion block, chess.rs:489-500).
 * ------------------------------------------------------------------------------------------------------------------- */

/* generate_all_flat_moves_pov (chess.rs:439-481): queen-like moves for every (from, to) in square order, then the
 * knight moves, then the promotions from the seventh to the eighth rank, piece-major.  out[i] = {from, to, promotion}. */
KZO_EXPORT int kzo_chess_flat_moves(int *out /* [1880][3] */) {
    int n = 0;
    for (int from = 0; from < 64; from++)
        for (int to = 0; to < 64; to++) {
            int df = from % 8 - to % 8, dr = from / 8 - to / 8;
            if (((df == 0) ^ (dr == 0)) || (df != 0 && abs(df) == abs(dr))) {
                out[n * 3] = from, out[n * 3 + 1] = to, out[n * 3 + 2] = 0;
                n++;
            }
        }
    for (int from = 0; from < 64; from++)
        for (int to = 0; to < 64; to++) {
            int df = abs(from % 8 - to % 8), dr = abs(from / 8 - to / 8);
            if ((df == 1 && dr == 2) || (df == 2 && dr == 1)) {
                out[n * 3] = from, out[n * 3 + 1] = to, out[n * 3 + 2] = 0;
                n++;
            }
        }
    for (int piece = 1; piece <= 4; piece++)
        for (int from_f = 0; from_f < 8; from_f++)
            for (int to_f = 0; to_f < 8; to_f++)
                if (abs(from_f - to_f) <= 1) {
                    out[n * 3] = 6 * 8 + from_f, out[n * 3 + 1] = 7 * 8 + to_f, out[n * 3 + 2] = piece;
                    n++;
                }
    return n; /* FLAT_MOVE_COUNT = 1880 (chess.rs:185, asserted :505) */
}

/* square_pov (chess.rs:397-406) and move_pov (chess.rs:409-416): black sees the board with the ranks flipped */
static int kzo_square_pov(int white_to_move, int sq) { return white_to_move ? sq : (7 - sq / 8) * 8 + sq % 8; }

/* ChessStdMapper::move_to_index (chess.rs:202-210): the position of the POV move in the flat list, or -1 (the
 * reference panics) when it is not one of the 1880 */
KZO_EXPORT int kzo_chess_move_to_index(int white_to_move, int from, int to, int promotion) {
    static int moves[1880 * 3], ready = 0;
    if (!ready) {
        kzo_chess_flat_moves(moves);
        ready = 1;
    }
    const int f = kzo_square_pov(white_to_move, from), t = kzo_square_pov(white_to_move, to);
    for (int i = 0; i < 1880; i++)
        if (moves[i * 3] == f && moves[i * 3 + 1] == t && moves[i * 3 + 2] == promotion) return i;
    return -1;
}

/* ChessLegacyConvPolicyMapper::move_to_index (chess.rs:224-236) over ClassifiedPovMove::{from_move, to_channel}
 * (chess.rs:299-346): channel * 64 + from, channels = 8 directions x 7 distances, 8 knight directions, 3 x 3
 * under-promotions (direction-major; pieces rook, bishop, knight).  A queen promotion is a plain queen-like move. */
KZO_EXPORT int kzo_chess_conv_index(int white_to_move, int from, int to, int promotion) {
    static const int queen_dirs[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
    static const int knight_deltas[8][2] = {{2, 1}, {1, 2}, {-1, 2}, {-2, 1}, {-2, -1}, {-1, -2}, {1, -2}, {2, -1}};
    const int f = kzo_square_pov(white_to_move, from), t = kzo_square_pov(white_to_move, to);
    const int dr = t / 8 - f / 8, df = t % 8 - f % 8;
    const int sr = (dr > 0) - (dr < 0), sf = (df > 0) - (df < 0);
    int channel = -1;
    if (promotion >= 2) { /* rook, bishop, knight */
        channel = 56 + 8 + (sf + 1) * 3 + (promotion - 2);
    } else {
        for (int d = 0; d < 8 && channel < 0; d++)
            if (queen_dirs[d][0] == sr && queen_dirs[d][1] == sf) {
                const int dist = abs(dr) > abs(df) ? abs(dr) : abs(df);
                if (dr == sr * dist && df == sf * dist) channel = d * 7 + dist - 1;
            }
        for (int d = 0; d < 8 && channel < 0; d++)
            if (knight_deltas[d][0] == dr && knight_deltas[d][1] == df) channel = 56 + d;
    }
    if (channel < 0) return -1; /* "Could not find move type" (chess.rs:331) */
    return channel * 64 + f;
}

/* The attention head's index of a POV move (rust/kz-misc/src/bin/write_chess_mapping.rs:51-66, the generator of
 * python/lib/mapping/chess_flat_to_att.txt): from * 88 + (to, or 64 + 3 * file(to) + piece for a promotion with
 * piece queen 0, rook 1, bishop 2, knight 3 ... as written there: file * 3 + p_i with p_i in 0..3). */
KZO_EXPORT int kzo_chess_att_index(int from_pov, int to_pov, int promotion) {
    const int att_to = promotion == 0 ? to_pov : 64 + (to_pov % 8) * 3 + (promotion - 1);
    return from_pov * (64 + 8 * 3) + att_to;
}
