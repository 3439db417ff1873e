/*
 * gten_oracle.c -- CPU restatement of tinyllama.cpp's gten forward path.
 *
 * TEST INFRASTRUCTURE ONLY (see gten_oracle.h).  Plain C99 + OpenMP; no SIMD
 * intrinsics: where the reference's AVX/SSE build keeps partial sums in
 * vector lanes, the same lane structure is written out with small scalar
 * arrays, so results are bit-identical to that build as long as the compiler
 * neither contracts mul+add into FMA nor re-associates (build with
 * -ffp-contract=off, no -ffast-math; see oracle/Makefile).
 *
 * Citations are file:line in the upstream repository (gten/... , tinyllama.cpp,
 * tinyllama_to_gten.py).
 */
#include "gten_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define QBLK 32 /* gten/quants.h:12-15: both block sizes are 32 */

#pragma pack(push, 1)
typedef struct { uint16_t d; int8_t  q[QBLK];     } q8blk; /* gten/quants.h:17-23, 34 B */
typedef struct { uint16_t d; uint8_t p[QBLK / 2]; } q4blk; /* gten/quants.h:25-31, 18 B */
#pragma pack(pop)

typedef char q8blk_is_34_bytes[(sizeof(q8blk) == 34) ? 1 : -1];
typedef char q4blk_is_18_bytes[(sizeof(q4blk) == 18) ? 1 : -1];

static int g_avx_order = 1;
void orc_set_simd(int avx_order) { g_avx_order = avx_order ? 1 : 0; }
int  orc_get_simd(void) { return g_avx_order; }

/* ------------------------------------------------------------------ fp16 */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* IEEE binary32 -> binary16, round to nearest even, overflow -> inf,
 * every NaN -> sign|0x7E00 (gten/gten_types.h:99-119). */
uint16_t orc_fp32_to_fp16(float f)
{
    const uint32_t x = f2u(f);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    uint32_t a = x & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);      /* >= 65520 rounds to inf */
    if (a < 0x38800000u) {                                        /* below 2^-14: half subnormal */
        const float t = u2f(a) + 0.5f;                            /* RNE lands in the low mantissa bits */
        return (uint16_t)(sign | (f2u(t) - 0x3f000000u));
    }
    const uint32_t odd = (a >> 13) & 1u;
    a += 0xc8000fffu + odd;                                       /* rebias 127->15, round half to even */
    return (uint16_t)(sign | (a >> 13));
}

/* gten/gten_types.h:79-97 (the LUT at 122-143 holds the same values) */
static float g_h2f[65536];
static int   g_h2f_ready = 0;

static float half_bits_to_float(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        const float v = (float)m * 5.9604644775390625e-8f;       /* m * 2^-24, exact */
        return u2f(sign | f2u(v));
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

static void h2f_init(void)
{
    if (g_h2f_ready) return;
    for (uint32_t i = 0; i < 65536; i++) g_h2f[i] = half_bits_to_float((uint16_t)i);
    g_h2f_ready = 1;
}

__attribute__((constructor)) static void orc_ctor(void) { h2f_init(); }

float orc_fp16_to_fp32(uint16_t h) { return g_h2f[h]; }
#define H2F(h) (g_h2f[(uint16_t)(h)])

/* ------------------------------------------------------------ row codecs */

size_t orc_row_bytes(int dtype, int cols)
{
    switch (dtype) {
    case ORC_I32: case ORC_F32: return (size_t)cols * 4;
    case ORC_F16: return (size_t)cols * 2;
    case ORC_Q8:  return (size_t)((cols + QBLK - 1) / QBLK) * sizeof(q8blk); /* gten/tensor.cpp:37-48 */
    case ORC_Q4:  return (size_t)(cols / QBLK) * sizeof(q4blk);              /* gten/tensor.cpp:49-55 */
    }
    return 0;
}

/* gten/quants.h:52-66: absmax/127 in f32, stored as fp16, but the scale used
 * for rounding is 1/delta of the UNROUNDED f32 delta; roundf = half away from 0. */
static void q8_quantize_block(const float* x, q8blk* out, int len)
{
    float amax = 0.0f;
    for (int j = 0; j < len; j++) {
        const float a = fabsf(x[j]);
        if (a > amax) amax = a;             /* std::max(absmax, |x|): NaN never replaces */
    }
    const float delta = amax / 127.0f;
    out->d = orc_fp32_to_fp16(delta);
    const float scale = (delta != 0.0f) ? 1.0f / delta : 0.0f;
    for (int j = 0; j < len; j++) out->q[j] = (int8_t)roundf(x[j] * scale);
}

/* gten/quants.h:92-110 (full blocks then the partial tail) */
void orc_q8_quantize_row(const float* x, void* out, int n)
{
    q8blk* o = (q8blk*)out;
    const int nb = n / QBLK, rem = n % QBLK;
    for (int b = 0; b < nb; b++) q8_quantize_block(x + b * QBLK, o + b, QBLK);
    if (rem) q8_quantize_block(x + nb * QBLK, o + nb, rem);
}

/* gten/quants.h:69-76, 118-133 */
void orc_q8_dequantize_row(const void* in, float* out, int n)
{
    const q8blk* b = (const q8blk*)in;
    for (int i = 0; i < n; i++) {
        const q8blk* blk = b + i / QBLK;
        out[i] = (float)blk->q[i % QBLK] * H2F(blk->d);
    }
}

/* gten/quants.h:78-90, 135-143: element i<16 = high nibble of byte i, element
 * i+16 = low nibble of byte i, both minus 7. */
void orc_q4_dequantize_row(const void* in, float* out, int n)
{
    const q4blk* b = (const q4blk*)in;
    for (int k = 0; k < n / QBLK; k++) {
        const float d = H2F(b[k].d);
        for (int i = 0; i < QBLK / 2; i++) {
            const int hi = (int)(b[k].p[i] >> 4) - 7;
            const int lo = (int)(b[k].p[i] & 0x0f) - 7;
            out[k * QBLK + i] = (float)hi * d;
            out[k * QBLK + i + QBLK / 2] = (float)lo * d;
        }
    }
}

/* gten/ops.h:40-70 */
void orc_read_row(const void* in, int dtype, float* out, int n)
{
    switch (dtype) {
    case ORC_Q4: orc_q4_dequantize_row(in, out, n); break;
    case ORC_Q8: orc_q8_dequantize_row(in, out, n); break;
    case ORC_F16: { const uint16_t* h = (const uint16_t*)in; for (int i = 0; i < n; i++) out[i] = H2F(h[i]); } break;
    case ORC_F32: memcpy(out, in, (size_t)n * 4); break;
    default: fprintf(stderr, "orc_read_row: bad dtype %d\n", dtype); abort();
    }
}

/* gten/ops.h:73-96 (there is no Q4 writer) */
void orc_write_row(const float* in, void* out, int dtype, int n)
{
    switch (dtype) {
    case ORC_Q8: orc_q8_quantize_row(in, out, n); break;
    case ORC_F16: { uint16_t* h = (uint16_t*)out; for (int i = 0; i < n; i++) h[i] = orc_fp32_to_fp16(in[i]); } break;
    case ORC_F32: memcpy(out, in, (size_t)n * 4); break;
    default: fprintf(stderr, "orc_write_row: bad dtype %d\n", dtype); abort();
    }
}

/* ------------------------------------------- offline weight quantizers */

void orc_weight_to_f16(const float* w, size_t numel, void* out)
{
    uint16_t* h = (uint16_t*)out;               /* tinyllama_to_gten.py:105-110: torch .to(float16) = RNE */
    for (size_t i = 0; i < numel; i++) h[i] = orc_fp32_to_fp16(w[i]);
}

/* tinyllama_to_gten.py:24-51: delta = absmax/127 (f32), scale = 1/delta where
 * delta != 0, q = torch.round(x*scale) -> HALF TO EVEN (unlike roundf above),
 * stored delta = fp16(delta).  Blocks enumerate rows, then 32-column groups. */
void orc_weight_quantize_q8(const float* w, int rows, int cols, void* out)
{
    const size_t nblk = (size_t)rows * (size_t)(cols / QBLK);
    q8blk* o = (q8blk*)out;
    #pragma omp parallel for schedule(static)
    for (size_t b = 0; b < nblk; b++) {
        const float* x = w + b * QBLK;
        float amax = 0.0f;
        for (int j = 0; j < QBLK; j++) { const float a = fabsf(x[j]); if (a > amax) amax = a; }
        const float delta = amax / 127.0f;
        const float scale = (delta != 0.0f) ? 1.0f / delta : 0.0f;
        o[b].d = orc_fp32_to_fp16(delta);
        for (int j = 0; j < QBLK; j++) o[b].q[j] = (int8_t)nearbyintf(x[j] * scale);
    }
}

/* tinyllama_to_gten.py:54-91: delta = absmax/7; q = round_half_even(x*scale)+7
 * in 0..14; first 16 elements go to the high nibbles, last 16 to the low. */
void orc_weight_quantize_q4(const float* w, int rows, int cols, void* out)
{
    const size_t nblk = (size_t)rows * (size_t)(cols / QBLK);
    q4blk* o = (q4blk*)out;
    #pragma omp parallel for schedule(static)
    for (size_t b = 0; b < nblk; b++) {
        const float* x = w + b * QBLK;
        float amax = 0.0f;
        for (int j = 0; j < QBLK; j++) { const float a = fabsf(x[j]); if (a > amax) amax = a; }
        const float delta = amax / 7.0f;
        const float scale = (delta != 0.0f) ? 1.0f / delta : 0.0f;
        o[b].d = orc_fp32_to_fp16(delta);
        for (int j = 0; j < QBLK / 2; j++) {
            const int hi = (int)nearbyintf(x[j] * scale) + 7;
            const int lo = (int)nearbyintf(x[j + QBLK / 2] * scale) + 7;
            o[b].p[j] = (uint8_t)((hi << 4) | (lo & 0x0f));
        }
    }
}

/* ---------------------------------------------------------- dot products */

/* horizontal sum of the 8 float lanes, left to right (gten/simd_ops.h:63-66) */
static inline float hsum8(const float* v)
{
    return v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
}

/* gten/ops.h:140-174.  AVX order: lane l accumulates elements l, l+8, ...
 * with a separate multiply and add (gten/simd_ops.h:59-61), lanes summed
 * left to right, then the scalar tail.  Scalar order: one running sum. */
static float dot_f16(const uint16_t* a, const uint16_t* b, int n)
{
    if (g_avx_order) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int nv = (n / 8) * 8;
        for (int i = 0; i < nv; i += 8)
            for (int l = 0; l < 8; l++) {
                const float p = H2F(a[i + l]) * H2F(b[i + l]);
                acc[l] = p + acc[l];
            }
        float s = hsum8(acc);
        for (int i = nv; i < n; i++) s += H2F(a[i]) * H2F(b[i]);
        return s;
    }
    float s = 0.0f;
    for (int i = 0; i < n; i++) s += H2F(a[i]) * H2F(b[i]);
    return s;
}

/* gten/ops.h:177-221.  The scalar build's 8-way unrolled loop is still one
 * running sum, so it is the plain sequential order. */
static float dot_f32(const float* a, const float* b, int n)
{
    if (g_avx_order) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int nv = (n / 8) * 8;
        for (int i = 0; i < nv; i += 8)
            for (int l = 0; l < 8; l++) {
                const float p = a[i + l] * b[i + l];
                acc[l] = p + acc[l];
            }
        float s = hsum8(acc);
        for (int i = nv; i < n; i++) s += a[i] * b[i];
        return s;
    }
    float s = 0.0f;
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* Shared tail of the two integer dot products.  SSE build (gten/ops.h:236-292,
 * 329-391): element e of a block feeds int lane (e%8)/2 (pairs from madd, the
 * four 8-element groups added lane-wise); each lane is converted to float,
 * multiplied by (da*db) and added to a 4-lane float accumulator; result is
 * (l0+l1)+(l2+l3) from the two hadd steps.  Scalar build (296-312, 454-475):
 * dot += isum * da * db, left to right. */
static inline void lane_split(const int* prod, int* lane)
{
    lane[0] = lane[1] = lane[2] = lane[3] = 0;
    for (int e = 0; e < QBLK; e++) lane[(e & 7) >> 1] += prod[e];
}

static float dot_q8_q8(const q8blk* a, const q8blk* b, int n)
{
    const int nb = n / QBLK;
    if (g_avx_order) {
        float acc[4] = {0, 0, 0, 0};
        for (int i = 0; i < nb; i++) {
            int prod[QBLK], lane[4];
            for (int e = 0; e < QBLK; e++) prod[e] = (int)a[i].q[e] * (int)b[i].q[e];
            lane_split(prod, lane);
            const float dd = H2F(a[i].d) * H2F(b[i].d);
            for (int l = 0; l < 4; l++) acc[l] = acc[l] + (float)lane[l] * dd;
        }
        return (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    float s = 0.0f;
    for (int i = 0; i < nb; i++) {
        int isum = 0;
        for (int e = 0; e < QBLK; e++) isum += (int)a[i].q[e] * (int)b[i].q[e];
        s += (float)isum * H2F(a[i].d) * H2F(b[i].d);
    }
    return s;
}

static float dot_q8_q4(const q8blk* a, const q4blk* b, int n)
{
    const int nb = n / QBLK;
    float acc[4] = {0, 0, 0, 0};
    float s = 0.0f;
    for (int i = 0; i < nb; i++) {
        int prod[QBLK];
        for (int j = 0; j < QBLK / 2; j++) {
            prod[j] = (int)a[i].q[j] * ((int)(b[i].p[j] >> 4) - 7);
            prod[j + QBLK / 2] = (int)a[i].q[j + QBLK / 2] * ((int)(b[i].p[j] & 0x0f) - 7);
        }
        if (g_avx_order) {
            int lane[4];
            lane_split(prod, lane);
            const float dd = H2F(a[i].d) * H2F(b[i].d);
            for (int l = 0; l < 4; l++) acc[l] = acc[l] + (float)lane[l] * dd;
        } else {
            int isum = 0;
            for (int e = 0; e < QBLK; e++) isum += prod[e];
            s += (float)isum * H2F(a[i].d) * H2F(b[i].d);
        }
    }
    return g_avx_order ? (acc[0] + acc[1]) + (acc[2] + acc[3]) : s;
}

/* gten/ops.h:482-512 */
float orc_vec_dot(const void* a, int a_dtype, const void* b, int b_dtype, int n)
{
    if (a_dtype == ORC_Q8 && b_dtype == ORC_Q4) return dot_q8_q4((const q8blk*)a, (const q4blk*)b, n);
    if (a_dtype == ORC_Q8 && b_dtype == ORC_Q8) return dot_q8_q8((const q8blk*)a, (const q8blk*)b, n);
    if (a_dtype == ORC_F16 && b_dtype == ORC_F16) return dot_f16((const uint16_t*)a, (const uint16_t*)b, n);
    if (a_dtype == ORC_F32 && b_dtype == ORC_F32) return dot_f32((const float*)a, (const float*)b, n);
    fprintf(stderr, "orc_vec_dot: unsupported dtype pair (%d,%d)\n", a_dtype, b_dtype);
    abort();
}

/* ------------------------------------------------------------- operators */

static float* xalloc_f32(size_t n)
{
    float* p = (float*)malloc((n ? n : 1) * sizeof(float));
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

/* gten/ops.h:514-564: row gather.  f16 and Q8 tables are copied verbatim; a
 * Q4 table row is dequantized and re-quantized to Q8 (ops.h:522-528). */
void orc_token_embed(const void* w, int w_dtype, size_t w_pitch, const int32_t* tokens,
                     void* out, int out_dtype, size_t out_pitch, int n, int d, int start_pos)
{
    float* buf = xalloc_f32((size_t)d);
    for (int i = start_pos; i < n; i++) {
        const char* src = (const char*)w + (size_t)tokens[i] * w_pitch;
        char* dst = (char*)out + (size_t)i * out_pitch;
        if (w_dtype == ORC_Q4) {
            orc_read_row(src, ORC_Q4, buf, d);
            orc_write_row(buf, dst, out_dtype, d);
        } else {
            memcpy(dst, src, orc_row_bytes(w_dtype, d));
        }
    }
    free(buf);
}

/* gten/ops.h:613-670: out[r][c] = dot(x[r,:], w[c,:]); the f32 row is then
 * written in the output dtype.  One thread computes one output feature
 * wholly, so the result does not depend on the thread count. */
void orc_matmul_2d(const void* x, int x_dtype, size_t x_pitch,
                   const void* w, int w_dtype, size_t w_pitch,
                   void* out, int out_dtype, size_t out_pitch,
                   int n, int d_in, int d_out, int start_pos)
{
    float* row = xalloc_f32((size_t)d_out);
    for (int r = start_pos; r < n; r++) {
        const char* xr = (const char*)x + (size_t)r * x_pitch;
        #pragma omp parallel for schedule(static)
        for (int c = 0; c < d_out; c++)
            row[c] = orc_vec_dot(xr, x_dtype, (const char*)w + (size_t)c * w_pitch, w_dtype, d_in);
        orc_write_row(row, (char*)out + (size_t)r * out_pitch, out_dtype, d_out);
    }
    free(row);
}

/* gten/ops.h:762-814: ss = sum x^2 sequentially, rms = sqrt(ss/N),
 * out = x / (rms + 1e-6) * w, eps OUTSIDE the sqrt, divide before multiply. */
void orc_rms_norm(const void* x, int dtype, size_t x_pitch, const uint16_t* w_f16,
                  void* out, size_t out_pitch, int n, int d, int start_pos)
{
    float* in = xalloc_f32((size_t)d * 2);
    float* o = in + d;
    for (int r = start_pos; r < n; r++) {
        orc_read_row((const char*)x + (size_t)r * x_pitch, dtype, in, d);
        float ss = 0.0f;
        for (int i = 0; i < d; i++) ss += in[i] * in[i];
        const float rms = sqrtf(ss / (float)d);
        for (int i = 0; i < d; i++) o[i] = in[i] / (rms + 1e-6f) * H2F(w_f16[i]);
        orc_write_row(o, (char*)out + (size_t)r * out_pitch, dtype, d);
    }
    free(in);
}

/* gten/ops.h:714-760: in place; position m = row index; pairs (j, j+d_head/2);
 * theta = m * powf(10000, -(2j/d_head)); all in f32 libm. */
void orc_rotary_emb(void* x, int dtype, size_t pitch, int n, int d, int d_head, int start_pos)
{
    float* buf = xalloc_f32((size_t)d);
    const int n_head = d / d_head, half = d_head / 2;
    const float dh = (float)d_head;
    for (int r = start_pos; r < n; r++) {
        char* row = (char*)x + (size_t)r * pitch;
        orc_read_row(row, dtype, buf, d);
        const float m = (float)r;
        for (int h = 0; h < n_head; h++) {
            float* v = buf + h * d_head;
            for (int j = 0; j < half; j++) {
                const float x0 = v[j], x1 = v[j + half];
                const float th = m * powf(10000.0f, -(2.0f * j / dh));
                const float c = cosf(th), s = sinf(th);
                v[j] = x0 * c - x1 * s;
                v[j + half] = x0 * s + x1 * c;
            }
        }
        orc_write_row(buf, row, dtype, d);
    }
    free(buf);
}

/* gten/ops.h:673-711 */
void orc_silu(const void* x, void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    float* buf = xalloc_f32((size_t)d);
    for (int r = start_pos; r < n; r++) {
        orc_read_row((const char*)x + (size_t)r * pitch, dtype, buf, d);
        for (int i = 0; i < d; i++) buf[i] = buf[i] / (1.0f + expf(-buf[i]));
        orc_write_row(buf, (char*)out + (size_t)r * pitch, dtype, d);
    }
    free(buf);
}

/* gten/ops.h:816-867 */
void orc_mul(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    float* fa = xalloc_f32((size_t)d * 2);
    float* fb = fa + d;
    for (int r = start_pos; r < n; r++) {
        orc_read_row((const char*)a + (size_t)r * pitch, dtype, fa, d);
        orc_read_row((const char*)b + (size_t)r * pitch, dtype, fb, d);
        for (int i = 0; i < d; i++) fa[i] = fa[i] * fb[i];
        orc_write_row(fa, (char*)out + (size_t)r * pitch, dtype, d);
    }
    free(fa);
}

/* gten/ops.h:870-910 */
void orc_add(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    float* fa = xalloc_f32((size_t)d * 2);
    float* fb = fa + d;
    for (int r = start_pos; r < n; r++) {
        orc_read_row((const char*)a + (size_t)r * pitch, dtype, fa, d);
        orc_read_row((const char*)b + (size_t)r * pitch, dtype, fb, d);
        for (int i = 0; i < d; i++) fa[i] = fa[i] + fb[i];
        orc_write_row(fa, (char*)out + (size_t)r * pitch, dtype, d);
    }
    free(fa);
}

/* gten/ops.h:930-1133.  For every new row r and head h:
 *   s_c = dot(q[r,h,:], k[c,h/grp,:]) * 1/sqrt(d_head)  for c <= r, else -inf
 *   p   = softmax over all n columns (three passes, divide by the sum)
 *   p is rounded to the activation dtype as a row of n values (fp16 RNE, or
 *   Q8 blocks of 32 along c with a partial tail block) and read back to f32
 *   out[h*d_head+e] = f32 dot over c of p with the dequantized, transposed V
 * then the whole output row is written in the activation dtype.  The
 * reference dequantizes+transposes the ENTIRE V cache per call (1003-1044). */
void orc_qkv_attn(const void* q, const void* k, const void* v, void* out, int dtype,
                  size_t q_pitch, size_t kv_pitch, size_t out_pitch,
                  int n, int n_heads, int n_kv_heads, int d_head, int start_pos)
{
    const int grp = n_heads / n_kv_heads;
    const int kv_dim = n_kv_heads * d_head;
    const int d = n_heads * d_head;
    const float scale = 1.0f / sqrtf((float)d_head);
    const size_t head_bytes = orc_row_bytes(dtype, d_head); /* bstride of the head axis, gten/tensor.h:97-117 */

    float* vt = xalloc_f32((size_t)kv_dim * n);             /* [kv_dim][n] */
    float* vrow = xalloc_f32((size_t)kv_dim);
    for (int c = 0; c < n; c++) {
        orc_read_row((const char*)v + (size_t)c * kv_pitch, dtype, vrow, kv_dim);
        for (int e = 0; e < kv_dim; e++) vt[(size_t)e * n + c] = vrow[e];
    }
    free(vrow);

    float* p = xalloc_f32((size_t)n);
    float* orow = xalloc_f32((size_t)d);
    void* pq = malloc(orc_row_bytes(dtype == ORC_Q8 ? ORC_Q8 : ORC_F16, n) + 64);

    for (int r = start_pos; r < n; r++) {
        for (int h = 0; h < n_heads; h++) {
            const char* qv = (const char*)q + (size_t)r * q_pitch + (size_t)h * head_bytes;
            for (int c = 0; c <= r; c++) {
                const char* kvp = (const char*)k + (size_t)c * kv_pitch + (size_t)(h / grp) * head_bytes;
                p[c] = orc_vec_dot(qv, dtype, kvp, dtype, d_head) * scale;
            }
            for (int c = r + 1; c < n; c++) p[c] = -INFINITY;
            float mx = -INFINITY;
            for (int c = 0; c < n; c++) if (p[c] > mx) mx = p[c];
            float sum = 0.0f;
            for (int c = 0; c < n; c++) { p[c] = expf(p[c] - mx); sum += p[c]; }
            for (int c = 0; c < n; c++) p[c] = p[c] / sum;
            orc_write_row(p, pq, dtype, n);
            orc_read_row(pq, dtype, p, n);
            for (int e = 0; e < d_head; e++)
                orow[h * d_head + e] = dot_f32(p, vt + (size_t)((h / grp) * d_head + e) * n, n);
        }
        orc_write_row(orow, (char*)out + (size_t)r * out_pitch, dtype, d);
    }
    free(pq); free(orow); free(p); free(vt);
}

/* ------------------------------------------------------------ whole model */

typedef struct {
    void *wq, *wk, *wv, *wo, *wgate, *wup, *wdown;      /* storage layout, wdtype */
    uint16_t *attn_norm, *ffn_norm;                     /* fp16, gten/modules.cpp:84 */
    void *kcache, *vcache;                              /* [max_ctx][kv_dim] adtype = Linear::acv of key/value */
} orc_layer;

struct orc_model {
    orc_config c;
    void* embed; uint16_t* norm; void* lm_head;
    orc_layer* L;
    /* activations shared by all layers: only the K/V caches must persist */
    void *x, *xn, *qb, *att, *proj, *h, *gate, *up, *down, *x2;
};

static void* xcalloc(size_t n)
{
    void* p = calloc(n ? n : 1, 1);
    if (!p) { fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n); abort(); }
    return p;
}

static size_t wbytes(const orc_model* m, int rows, int cols, int dtype)
{
    (void)m;
    return (size_t)rows * orc_row_bytes(dtype, cols);
}

orc_model* orc_model_create(const orc_config* cfg)
{
    orc_model* m = (orc_model*)xcalloc(sizeof(*m));
    m->c = *cfg;
    const orc_config* c = &m->c;
    const int E = c->n_embd, F = c->n_ffn, V = c->n_vocab, T = c->max_ctx;
    const int dh = E / c->n_heads, KV = dh * c->n_kv_heads;
    m->embed = xcalloc(wbytes(m, V, E, c->wdtype));
    m->lm_head = xcalloc(wbytes(m, V, E, c->wdtype));
    m->norm = (uint16_t*)xcalloc((size_t)E * 2);
    m->L = (orc_layer*)xcalloc(sizeof(orc_layer) * (size_t)c->n_layers);
    for (int l = 0; l < c->n_layers; l++) {
        orc_layer* y = &m->L[l];
        y->wq = xcalloc(wbytes(m, E, E, c->wdtype));
        y->wk = xcalloc(wbytes(m, KV, E, c->wdtype));
        y->wv = xcalloc(wbytes(m, KV, E, c->wdtype));
        y->wo = xcalloc(wbytes(m, E, E, c->wdtype));
        y->wgate = xcalloc(wbytes(m, F, E, c->wdtype));
        y->wup = xcalloc(wbytes(m, F, E, c->wdtype));
        y->wdown = xcalloc(wbytes(m, E, F, c->wdtype));
        y->attn_norm = (uint16_t*)xcalloc((size_t)E * 2);
        y->ffn_norm = (uint16_t*)xcalloc((size_t)E * 2);
        y->kcache = xcalloc((size_t)T * orc_row_bytes(c->adtype, KV));
        y->vcache = xcalloc((size_t)T * orc_row_bytes(c->adtype, KV));
    }
    const size_t rowE = orc_row_bytes(c->adtype, E), rowF = orc_row_bytes(c->adtype, F);
    m->x = xcalloc((size_t)T * rowE);   m->xn = xcalloc((size_t)T * rowE);
    m->qb = xcalloc((size_t)T * rowE);  m->att = xcalloc((size_t)T * rowE);
    m->proj = xcalloc((size_t)T * rowE); m->h = xcalloc((size_t)T * rowE);
    m->gate = xcalloc((size_t)T * rowF); m->up = xcalloc((size_t)T * rowF);
    m->down = xcalloc((size_t)T * rowE); m->x2 = xcalloc((size_t)T * rowE);
    return m;
}

void orc_model_free(orc_model* m)
{
    if (!m) return;
    for (int l = 0; l < m->c.n_layers; l++) {
        orc_layer* y = &m->L[l];
        free(y->wq); free(y->wk); free(y->wv); free(y->wo); free(y->wgate); free(y->wup); free(y->wdown);
        free(y->attn_norm); free(y->ffn_norm); free(y->kcache); free(y->vcache);
    }
    free(m->L); free(m->embed); free(m->norm); free(m->lm_head);
    free(m->x); free(m->xn); free(m->qb); free(m->att); free(m->proj); free(m->h);
    free(m->gate); free(m->up); free(m->down); free(m->x2);
    free(m);
}

int orc_model_n_weights(const orc_model* m) { return 1 + 9 * m->c.n_layers + 2; }

/* .gten order, tinyllama.cpp:345-391: embed; per layer q,k,v,o,gate,up,down,
 * input_layernorm, post_attention_layernorm; model.norm; lm_head. */
static void* weight_slot(const orc_model* m, int idx, int* rows, int* cols, int* dtype)
{
    const orc_config* c = &m->c;
    const int E = c->n_embd, F = c->n_ffn, V = c->n_vocab;
    const int KV = (E / c->n_heads) * c->n_kv_heads;
    const int last = orc_model_n_weights(m) - 1;
    if (idx == 0)        { *rows = V; *cols = E; *dtype = c->wdtype; return m->embed; }
    if (idx == last)     { *rows = V; *cols = E; *dtype = c->wdtype; return m->lm_head; }
    if (idx == last - 1) { *rows = 1; *cols = E; *dtype = ORC_F16;   return m->norm; }
    const orc_layer* y = &m->L[(idx - 1) / 9];
    *dtype = c->wdtype;
    switch ((idx - 1) % 9) {
    case 0: *rows = E;  *cols = E; return y->wq;
    case 1: *rows = KV; *cols = E; return y->wk;
    case 2: *rows = KV; *cols = E; return y->wv;
    case 3: *rows = E;  *cols = E; return y->wo;
    case 4: *rows = F;  *cols = E; return y->wgate;
    case 5: *rows = F;  *cols = E; return y->wup;
    case 6: *rows = E;  *cols = F; return y->wdown;
    case 7: *rows = 1;  *cols = E; *dtype = ORC_F16; return y->attn_norm;
    default: *rows = 1; *cols = E; *dtype = ORC_F16; return y->ffn_norm;
    }
}

void orc_model_weight_shape(const orc_model* m, int idx, int* rows, int* cols, int* dtype)
{
    (void)weight_slot(m, idx, rows, cols, dtype);
}

size_t orc_model_weight_bytes(const orc_model* m, int idx)
{
    int r, c, d;
    (void)weight_slot(m, idx, &r, &c, &d);
    return (size_t)r * orc_row_bytes(d, c);
}

void orc_model_set_weight(orc_model* m, int idx, const void* bytes, size_t nbytes)
{
    int r, c, d;
    void* dst = weight_slot(m, idx, &r, &c, &d);
    if (nbytes != (size_t)r * orc_row_bytes(d, c)) {
        fprintf(stderr, "orc_model_set_weight: weight %d expects %zu bytes, got %zu\n",
                idx, (size_t)r * orc_row_bytes(d, c), nbytes);
        abort();
    }
    memcpy(dst, bytes, nbytes);
}

/* tinyllama.cpp:301-392: magic, then per tensor [i32 len][name][i32 len][name][i32 nbytes][payload];
 * names are skipped, order and payload size are what is checked. */
int orc_model_load_gten(orc_model* m, const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    int64_t magic = 0;
    if (fread(&magic, 8, 1, f) != 1 || magic != 0x454c49464e455447LL) { fclose(f); return -2; }
    const int nw = orc_model_n_weights(m);
    for (int i = 0; i < nw; i++) {
        for (int rep = 0; rep < 2; rep++) {
            int32_t len = 0;
            if (fread(&len, 4, 1, f) != 1 || len < 0 || fseek(f, len, SEEK_CUR) != 0) { fclose(f); return -3; }
        }
        int32_t nbytes = 0;
        if (fread(&nbytes, 4, 1, f) != 1) { fclose(f); return -3; }
        int r, c, d;
        void* dst = weight_slot(m, i, &r, &c, &d);
        if ((size_t)nbytes != (size_t)r * orc_row_bytes(d, c)) { fclose(f); return -4; }
        if (fread(dst, 1, (size_t)nbytes, f) != (size_t)nbytes) { fclose(f); return -5; }
    }
    fclose(f);
    return 0;
}

/* TinyLlama::logits, tinyllama.cpp:45-61; block order gten/modules.cpp:193-254;
 * lm_head only on the last row into f32 (gten/modules.cpp:70-81). */
void orc_model_logits(orc_model* m, const int32_t* tokens, int n, int sp, float* out)
{
    const orc_config* c = &m->c;
    const int E = c->n_embd, F = c->n_ffn, V = c->n_vocab, A = c->adtype, W = c->wdtype;
    const int dh = E / c->n_heads, KV = dh * c->n_kv_heads;
    const size_t rowE = orc_row_bytes(A, E), rowF = orc_row_bytes(A, F), rowKV = orc_row_bytes(A, KV);
    const size_t wE = orc_row_bytes(W, E), wF = orc_row_bytes(W, F);

    orc_token_embed(m->embed, W, wE, tokens, m->x, A, rowE, n, E, sp);
    void* x = m->x;
    for (int l = 0; l < c->n_layers; l++) {
        void* nxt = (x == m->x) ? m->x2 : m->x;
        orc_layer* y = &m->L[l];
        orc_rms_norm(x, A, rowE, y->attn_norm, m->xn, rowE, n, E, sp);
        orc_matmul_2d(m->xn, A, rowE, y->wq, W, wE, m->qb, A, rowE, n, E, E, sp);
        orc_matmul_2d(m->xn, A, rowE, y->wk, W, wE, y->kcache, A, rowKV, n, E, KV, sp);
        orc_rotary_emb(m->qb, A, rowE, n, E, dh, sp);
        orc_rotary_emb(y->kcache, A, rowKV, n, KV, dh, sp);
        orc_matmul_2d(m->xn, A, rowE, y->wv, W, wE, y->vcache, A, rowKV, n, E, KV, sp);
        orc_qkv_attn(m->qb, y->kcache, y->vcache, m->att, A, rowE, rowKV, rowE,
                     n, c->n_heads, c->n_kv_heads, dh, sp);
        orc_matmul_2d(m->att, A, rowE, y->wo, W, wE, m->proj, A, rowE, n, E, E, sp);
        orc_add(x, m->proj, m->h, A, rowE, n, E, sp);
        orc_rms_norm(m->h, A, rowE, y->ffn_norm, m->xn, rowE, n, E, sp);
        orc_matmul_2d(m->xn, A, rowE, y->wgate, W, wE, m->gate, A, rowF, n, E, F, sp);
        orc_matmul_2d(m->xn, A, rowE, y->wup, W, wE, m->up, A, rowF, n, E, F, sp);
        orc_silu(m->gate, m->gate, A, rowF, n, F, sp);
        orc_mul(m->gate, m->up, m->gate, A, rowF, n, F, sp);
        orc_matmul_2d(m->gate, A, rowF, y->wdown, W, wF, m->down, A, rowE, n, F, E, sp);
        orc_add(m->h, m->down, nxt, A, rowE, n, E, sp);
        x = nxt;
    }
    orc_rms_norm(x, A, rowE, m->norm, m->xn, rowE, n, E, sp);
    /* last row only; a one-row matmul whose f32 row lands in out */
    orc_matmul_2d((const char*)m->xn + (size_t)(n - 1) * rowE, A, rowE, m->lm_head, W, wE,
                  out, ORC_F32, (size_t)V * 4, 1, E, V, 0);
}
