/* TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the reference's block-quantised arithmetic (SURVEY App. B rule 1, quantised half):
 *
 *   weights      ggml blocks of 32 (layouts W/ggml-quants.h:10-47): q4_0 q4_1 q5_0 q5_1 q8_0
 *   activations  a mul_mat whose weight operand is quantised first turns every f32 activation row into
 *                q8_0 / q8_1 blocks (W/ggml.c:9841-9857, vec_dot_type table W/ggml.c:397-583) and then
 *                takes an INTEGER dot per block, scaled by d_w * d_a (+ m_w * s_a for the *_1 types)
 *                (W/ggml-quants.c:2442-3560).
 *
 * The compiled reference of this repository (oracle/Makefile: -mavx2 -mfma -mf16c) runs the AVX2 bodies, so
 * that is the arithmetic restated here, in scalar form:
 *
 *   row quantiser (W/ggml-quants.c:720-776, 958-1016): amax over the block; d = amax / 127; id = 127 / amax
 *     (NOT 1 / d); q = round-to-nearest-even(x * id); q8_1: s = d * (float) sum(q); q8_0: d is stored as f16.
 *   dot (…:2640-2662, 2826-2857, 3056-3082, 3308-3335, 3512-3530): the 32 products of a block fall into 8
 *     groups of 4 consecutive elements (one 32-bit lane of the 256-bit vector each); per block every group
 *     does acc[g] = fma((float) isum_g, d_w * d_a, acc[g]); the *_1 types also summs += m_w * s_a (plain
 *     multiply, plain add: the file is C11, i.e. -ffp-contract=off — this file is compiled the same way);
 *     result = ((acc4 + acc0) + (acc6 + acc2)) + ((acc5 + acc1) + (acc7 + acc3)) [+ summs]
 *     (hsum_float_8, …:67-73).
 *   get_rows on a quantised token embedding (W/ggml.c:10701-10860 -> dequantize_row_*, …:1090-1200):
 *     x = q * d (+ m) in f32.
 *
 * Pinned bit for bit against the reference's own exported functions (quantize_row_q8_0/1,
 * ggml_vec_dot_*_q8_*, dequantize_row_*) by tests/test_oracle_quants.py in the build container.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <immintrin.h>

enum { T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8 };

static inline float h2f(uint16_t h) { return _cvtsh_ss(h); }
static inline uint16_t f2h(float f) { return _cvtss_sh(f, 0); }

int port_q_block_bytes(int type) {
    switch (type) { case T_Q4_0: return 18; case T_Q4_1: return 20; case T_Q5_0: return 22; case T_Q5_1: return 24; case T_Q8_0: return 34; default: return 0; }
}
/* 1 when the activation side of this weight type is q8_1 (carries s), 0 for q8_0 */
int port_q_act_has_sum(int type) { return type == T_Q4_1 || type == T_Q5_1; }

/* one block of weights -> 32 signed integers in element order, plus its scale d and offset m (0 for the symmetric types) */
static void unpack_block(int type, const uint8_t * b, int w[32], float * d, float * m) {
    uint16_t dh, mh = 0; uint32_t qh = 0; const uint8_t * qs;
    memcpy(&dh, b, 2);
    switch (type) {
        case T_Q4_0: qs = b + 2; break;
        case T_Q4_1: memcpy(&mh, b + 2, 2); qs = b + 4; break;
        case T_Q5_0: memcpy(&qh, b + 2, 4); qs = b + 6; break;
        case T_Q5_1: memcpy(&mh, b + 2, 2); memcpy(&qh, b + 4, 4); qs = b + 8; break;
        default:     qs = b + 2; break;
    }
    *d = h2f(dh); *m = (type == T_Q4_1 || type == T_Q5_1) ? h2f(mh) : 0.0f;
    if (type == T_Q8_0) { for (int j = 0; j < 32; ++j) w[j] = (int8_t) qs[j]; return; }
    for (int j = 0; j < 16; ++j) {
        int lo = qs[j] & 0x0F, hi = qs[j] >> 4;
        if (type == T_Q5_0 || type == T_Q5_1) { lo |= ((qh >> j) & 1) << 4; hi |= ((qh >> (j + 16)) & 1) << 4; }
        if (type == T_Q4_0) { lo -= 8; hi -= 8; }
        if (type == T_Q5_0) { lo -= 16; hi -= 16; }
        w[j] = lo; w[j + 16] = hi;
    }
}

/* f32 row -> q8 blocks.  qs [k] int8, d [k/32], s [k/32] (s only written when with_sum).  d_f16: q8_0 keeps d as f16. */
void port_quantize_row_q8(const float * x, int k, int with_sum, int8_t * qs, float * d, float * s) {
    for (int i = 0; i < k / 32; ++i) {
        const float * xb = x + 32 * i;
        float amax = 0.0f;
        for (int j = 0; j < 32; ++j) { const float a = fabsf(xb[j]); if (a > amax) amax = a; }
        const float dd = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        int sum = 0;
        for (int j = 0; j < 32; ++j) {
            const float v = xb[j] * id;
            const int q = (int) nearbyintf(v);            /* default rounding mode: to nearest, ties to even */
            qs[32 * i + j] = (int8_t) q; sum += q;
        }
        if (with_sum) { d[i] = dd; s[i] = dd * (float) sum; }
        else          { d[i] = h2f(f2h(dd)); }
    }
}

/* dot of one weight row (ggml blocks of `type`) with one quantised activation row */
float port_vec_dot_q(int type, int k, const uint8_t * wrow, const int8_t * qs, const float * d, const float * s) {
    const int bb = port_q_block_bytes(type), has_m = port_q_act_has_sum(type);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, summs = 0.0f;
    for (int i = 0; i < k / 32; ++i) {
        int w[32]; float dw, mw;
        unpack_block(type, wrow + (size_t) i * bb, w, &dw, &mw);
        if (has_m) { const float t = mw * s[i]; summs = summs + t; }
        const float scale = dw * d[i];
        for (int g = 0; g < 8; ++g) {
            int isum = 0;
            for (int t = 0; t < 4; ++t) isum += w[4 * g + t] * (int) qs[32 * i + 4 * g + t];
            acc[g] = fmaf((float) isum, scale, acc[g]);
        }
    }
    const float r = ((acc[4] + acc[0]) + (acc[6] + acc[2])) + ((acc[5] + acc[1]) + (acc[7] + acc[3]));
    return has_m ? r + summs : r;
}

/* the scalar form of the same dot (W/ggml-quants.c:2707-2728, 3436-3454): one integer sum per block.  Not what the compiled
 * reference runs; kept as the order the GPU kernels use (integer sums exact, f32 combination per block), so that the
 * distance between the two f32 orders can be measured on the CPU (tests/test_oracle_quants.py). */
float port_vec_dot_q_blockwise(int type, int k, const uint8_t * wrow, const int8_t * qs, const float * d, const float * s) {
    const int bb = port_q_block_bytes(type), has_m = port_q_act_has_sum(type);
    float sumf = 0.0f, summs = 0.0f;
    for (int i = 0; i < k / 32; ++i) {
        int w[32]; float dw, mw;
        unpack_block(type, wrow + (size_t) i * bb, w, &dw, &mw);
        int isum = 0;
        for (int j = 0; j < 32; ++j) isum += w[j] * (int) qs[32 * i + j];
        sumf = fmaf((float) isum, dw * d[i], sumf);
        if (has_m) summs = fmaf(mw, s[i], summs);
    }
    return sumf + summs;
}

void port_dequantize_row(int type, const uint8_t * wrow, float * y, int k) {
    const int bb = port_q_block_bytes(type), has_m = port_q_act_has_sum(type);
    for (int i = 0; i < k / 32; ++i) {
        int w[32]; float dw, mw;
        unpack_block(type, wrow + (size_t) i * bb, w, &dw, &mw);
        for (int j = 0; j < 32; ++j) {
            const float t = (float) w[j] * dw;
            y[32 * i + j] = has_m ? t + mw : t;
        }
    }
}
