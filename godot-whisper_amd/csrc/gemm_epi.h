// Epilogues of the f16 MFMA GEMMs (k_gemm.hip, k_gemm8.hip): everything the reference runs as separate graph nodes behind a mul_mat
// (bias add, GELU through the f16 table, residual add, q/k scaling, f32 -> f16 copies into the K / V layouts; W/whisper.cpp:1660-2074)
// applied to the accumulator fragments of v_mfma_f32_16x16x32_f16, in both fragment orientations.
#pragma once
#include "kernels.h"
#include "wave_ops.h"

namespace wmi { namespace k { namespace gemm_detail {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float    floatx4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ float round_f16(float x) { return __half2float(f2h(x)); }

// GELU exactly as the reference evaluates it: input rounded to f16, tanh form in f32, result rounded
// to f16 (its 65536-entry table is this function tabulated; W/ggml.c:1400-1423, 2229-2231)
__device__ __forceinline__ float gelu16(float x) {
    const float xh = round_f16(x);
    const float g  = 0.5f * xh * (1.0f + tanhf(0.79788456080286535587989211986876f * xh * (1.0f + 0.044715f * xh * xh)));
    return round_f16(g);
}

// Epilogue variant for the encoder GEMMs (24.6 M evaluations per batched mlp.0 launch): tanh through the hardware
// exponential, tanh(u) = 1 - 2 / (exp(2u) + 1) — ~10 VALU instructions instead of libm's tanhf (which, like a
// 65 536-entry table gather, costs as much as the whole GEMM: 39 -> 72 us measured).  v_exp_f32 is good to ~2 ulp
// of f32; after the two f16 roundings the result equals gelu16's except for a 1-ulp(f16) flip on a few per mille of
// the inputs (same trade as exp16_fast in the encoder attention).  The decoder's one-row kernels keep tanhf.
__device__ __forceinline__ float gelu16_fast(float x) {
    const float xh = round_f16(x);
    const float u  = 0.79788456080286535587989211986876f * xh * (1.0f + 0.044715f * xh * xh);
    const float t  = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * u) + 1.0f);      // v_rcp_f32: 1 ulp, no division sequence
    return round_f16(0.5f * xh * (1.0f + t));
}

// Two GELUs at once for the transposed epilogue (a lane holds adjacent columns): gelu(x) = x / (1 + exp(-2u)), u = k0 x (1 + k1 x^2),
// which is 0.5 x (1 + tanh u) without the 1 + tanh cancellation — packed f32 multiplies / FMAs, one v_exp_f32 and one v_rcp_f32 per
// element: ~8 issue slots against ~18 for gelu16_fast (at M = 12 000 the GELU arithmetic was most of mlp.0's 5 us epilogue, itself
// 39 % of a workgroup's life: profiles/r03b_gemm_phase_probe.txt).  Input and result rounded to f16 like the reference's table;
// over all 63 488 finite f16 inputs this form differs from the table on 271 entries (<= 2 ulp, all in the negative tail or at
// |x| < 0.36), gelu16_fast on 422 (<= 5 ulp) — numpy emulation, hardware exp / rcp add their own 1-2 ulp of f32.
typedef float    float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v  __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2v gelu16_pair(float2v x) {
    const half2v xh = __builtin_convertvector(x, half2v);
    const float2v xf = {(float) xh[0], (float) xh[1]};
    constexpr float C0 = -2.0f * 0.79788456080286535587989211986876f * 1.44269504088896340736f;
    constexpr float C1 = C0 * 0.044715f;
    const float2v c0 = {C0, C0}, c1 = {C1, C1}, one = {1.0f, 1.0f};
    const float2v w = __builtin_elementwise_fma(xf * xf, c1, c0) * xf;          // -2u log2(e)
    const float2v e = {__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    const float2v d = one + e;
    const float2v r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const float2v g = xf * r;
    return __builtin_convertvector(g, half2v);
}

// EPI_CONV2 over several lock-step chunks in one launch (GemmArgs::rows_per_chunk = P > 0): the implicit-GEMM rows of chunk c are
// c P .. c P + P - 1 (P = T + 4: the chunk's T output frames and the rows that straddle its guard rows), output row c T + t, positional
// row t; rows t >= T are not stored.  P = 0: one chunk, row m is both.
struct Conv2Row { int out, pe; bool valid; };
__device__ __forceinline__ Conv2Row conv2_row(const GemmArgs & a, int m) {
    if (a.rows_per_chunk <= 0) return Conv2Row{m, m, true};
    const int P = a.rows_per_chunk, c = m / P, t = m - c * P;
    return Conv2Row{c * (P - 4) + t, t < P - 4 ? t : 0, t < P - 4};
}

// first orientation, mfma(A rows, W rows): fragment (i, j) holds rows m = mb + i*16 + fq*4 + r (r = 0..3) of column n = nb + j*16 + frow.
// Interior tiles take the instantiation without bounds checks: a per-element `if (m < M)` makes every store its own basic block, and
// hipcc then waits vmcnt(0) before each one (vmcnt also counts stores on gfx9-family parts), i.e. the 64 stores of a lane complete one
// after the other (profiles/: -10..30 % kernel time on the M = 12 000 GEMMs).
template <int FM, int FN, bool GUARD>
__device__ __forceinline__ void epilogue_vt_wide(const GemmArgs & a, floatx4 (&acc)[FM][FN], const int mb, const int nb, const int lane);
__device__ __forceinline__ bool vt_wide_ok(const GemmArgs & a);
template <int EPI, int FM, int FN, bool GUARD>
__device__ __forceinline__ void epilogue_rows(const GemmArgs & a, floatx4 (&acc)[FM][FN], const int mb, const int nb, const int lane) {
    if constexpr (EPI == EPI_QKV_ENC) {
        // the V^T third (a wave tile never straddles the thirds: 64 | S) with whole-line stores where the chunks' 16-step blocks line up
        if (__builtin_amdgcn_readfirstlane(nb) >= 2 * a.S && vt_wide_ok(a)) { epilogue_vt_wide<FM, FN, GUARD>(a, acc, mb, nb, lane); return; }
    }
    const int frow = lane & 15, fq = lane >> 4;
    float biasv[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = nb + j * 16 + frow;
        biasv[j] = (a.bias && (!GUARD || n < a.N)) ? a.bias[n] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = nb + j * 16 + frow;
        if (GUARD && n >= a.N) continue;
        const float bias = biasv[j];
        // residual / positional operand of this fragment column: all FM x 4 values requested before the first store (the output
        // is updated in place, so element by element every load had to wait for the previous store: one round trip per element)
        float rpre[FM][4];
        if constexpr (EPI == EPI_F32_BIAS_RESID || EPI == EPI_CONV2) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mb + i * 16 + fq * 4 + r;
                    const int mc = (GUARD && m >= a.M) ? a.M - 1 : m;
                    rpre[i][r] = a.resid[(size_t) (EPI == EPI_CONV2 ? conv2_row(a, mc).pe : mc) * a.ldr + n];
                }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int mrow = mb + i * 16 + fq * 4;
            if constexpr (EPI == EPI_QKV_ENC) {
                const int seg = n / a.S, c = n - seg * a.S;
                if (seg == 2) {            // V^T: four consecutive time steps per lane -> one 8-byte store
                    // batched encode: row m = chunk * rows_per_chunk + t, V^T is [chunk][S][Tpad]
                    const int rpc = a.rows_per_chunk > 0 ? a.rows_per_chunk : a.M;
                    const int cb = mrow / rpc, t0 = mrow - cb * rpc;
                    __half * vt = (__half *) a.aux2 + (size_t) cb * a.chunk_stride_aux2 + (size_t) c * a.ldaux2;
                    if ((!GUARD || mrow + 3 < a.M) && t0 + 3 < rpc && ((t0 & 3) == 0)) {
                        half4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (_Float16) pin_f32(acc[i][j][r] + bias);
                        *(half4 *) (vt + vt_pos(t0)) = v;
                    } else {
                        for (int r = 0; r < 4; ++r) {
                            const int m = mrow + r;
                            if (m >= a.M) continue;
                            const int cb2 = m / rpc, t = m - cb2 * rpc;
                            ((__half *) a.aux2)[(size_t) cb2 * a.chunk_stride_aux2 + (size_t) c * a.ldaux2 + vt_pos(t)] = f2h(acc[i][j][r] + bias);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow + r;
                    if (GUARD && m >= a.M) continue;
                    const float v = acc[i][j][r] + bias;
                    if (seg == 0) ((__half *) a.C)[(size_t) m * a.ldc + c] = f2h(v);
                    else          ((__half *) a.aux)[(size_t) m * a.ldaux + c] = f2h(v);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow + r;
                if (GUARD && m >= a.M) continue;
                const float v = acc[i][j][r];
                if constexpr (EPI == EPI_F16_BIAS) {
                    ((__half *) a.C)[(size_t) m * a.ldc + n] = f2h(v + bias);
                } else if constexpr (EPI == EPI_F16_BIAS_GELU) {
                    ((__half *) a.C)[(size_t) m * a.ldc + n] = f2h(gelu16_fast(v + bias));
                } else if constexpr (EPI == EPI_F32_BIAS_RESID) {
                    ((float *) a.C)[(size_t) m * a.ldc + n] = (v + bias) + rpre[i][r];
                } else if constexpr (EPI == EPI_CONV2) {
                    const float g = gelu16_fast(v + bias);
                    const Conv2Row cr = conv2_row(a, m);
                    if (cr.valid) {
                        if (a.aux) ((float *) a.aux)[(size_t) cr.out * a.ldaux + n] = g;
                        ((float *) a.C)[(size_t) cr.out * a.ldc + n] = rpre[i][r] + g;
                    }
                } else if constexpr (EPI == EPI_QKV_DEC) {
                    // The q | k | v segment is decided per 16-column fragment on a WAVE-UNIFORM value
                    // (S is a multiple of 16, so a fragment never straddles a segment).  A per-lane
                    // three-way `if` here is miscompiled by hipcc 7.2 for gfx950 (the third arm's
                    // pointer select is dropped by the control-flow structurizer: v lands in the k
                    // cache) — see DESIGN.md "toolchain hazards"; wmi_selftest_proj / tests/test_gpu_parity.py::test_decoder_projection_paths_agree pins it.
                    const int seg = __builtin_amdgcn_readfirstlane((nb + j * 16) / a.S);
                    const int c = n - seg * a.S;
                    __half * dst; float val;
                    if (seg == 0)      { dst = (__half *) a.C    + (size_t) m * a.ldc;    val = (v + bias) * a.scale; }
                    else if (seg == 1) { dst = (__half *) a.aux  + (size_t) m * a.ldaux;  val = v * a.scale; }
                    else               { dst = (__half *) a.aux2 + (size_t) m * a.ldaux2; val = v + bias; }
                    dst[c] = f2h(val);
                } else if constexpr (EPI == EPI_CROSS_KV) {
                    const int il = n / (2 * a.S), c = n - il * 2 * a.S;
                    if (c < a.S) ((__half *) a.C)[il * a.layer_stride + (size_t) m * a.ldc + c] = f2h(v * a.scale);
                    else         ((__half *) a.aux)[il * a.layer_stride + (size_t) m * a.ldaux + (c - a.S)] = f2h(v + bias);
                } else if constexpr (EPI == EPI_Q_SCALED) {
                    ((__half *) a.C)[(size_t) m * a.ldc + n] = f2h((v + bias) * a.scale);
                }
            }
        }
    }
}

// transposed orientation, mfma(W rows, A rows): fragment (i, j) holds row m = mb + i*16 + frow, columns n = nb + j*16 + fq*4 + r (r = 0..3):
// one 8-byte (f16) or 16-byte (f32) store per fragment and lane.  n0 = first column of the workgroup tile (segment decisions).
template <int EPI, int FM, int FN, bool GUARD>
__device__ __forceinline__ void epilogue_cols(const GemmArgs & a, floatx4 (&acc)[FM][FN], const int mb, const int nb, const int n0, const int lane) {
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = nb + j * 16 + fq * 4;
        if (GUARD && n >= a.N) continue;                       // N is a multiple of 4 on every caller of these epilogues
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) b4 = *(const float4 *) (a.bias + n);
        const float bias[4] = {b4.x, b4.y, b4.z, b4.w};
        float4 rpre[FM];
        if constexpr (EPI == EPI_F32_BIAS_RESID || EPI == EPI_CONV2) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = mb + i * 16 + frow;
                const int mc = (GUARD && m >= a.M) ? a.M - 1 : m;
                rpre[i] = *(const float4 *) (a.resid + (size_t) (EPI == EPI_CONV2 ? conv2_row(a, mc).pe : mc) * a.ldr + n);
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mb + i * 16 + frow;
            if (GUARD && m >= a.M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
            auto put16 = [&](__half * dst, const float (&x)[4]) {
                half4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (_Float16) pin_f32(x[r]);
                *(half4 *) dst = h;
            };
            if constexpr (EPI == EPI_F16_BIAS) {
                const float x[4] = {v[0] + bias[0], v[1] + bias[1], v[2] + bias[2], v[3] + bias[3]};
                put16((__half *) a.C + (size_t) m * a.ldc + n, x);
            } else if constexpr (EPI == EPI_F16_BIAS_GELU) {
                const float2v x0 = {v[0] + bias[0], v[1] + bias[1]}, x1 = {v[2] + bias[2], v[3] + bias[3]};
                const half2v g0 = gelu16_pair(x0), g1 = gelu16_pair(x1);
                half4 h; h[0] = g0[0]; h[1] = g0[1]; h[2] = g1[0]; h[3] = g1[1];
                *(half4 *) ((__half *) a.C + (size_t) m * a.ldc + n) = h;
            } else if constexpr (EPI == EPI_Q_SCALED) {
                const float x[4] = {(v[0] + bias[0]) * a.scale, (v[1] + bias[1]) * a.scale, (v[2] + bias[2]) * a.scale, (v[3] + bias[3]) * a.scale};
                put16((__half *) a.C + (size_t) m * a.ldc + n, x);
            } else if constexpr (EPI == EPI_F32_BIAS_RESID) {
                float4 o;
                o.x = (v[0] + bias[0]) + rpre[i].x; o.y = (v[1] + bias[1]) + rpre[i].y; o.z = (v[2] + bias[2]) + rpre[i].z; o.w = (v[3] + bias[3]) + rpre[i].w;
                *(float4 *) ((float *) a.C + (size_t) m * a.ldc + n) = o;
            } else if constexpr (EPI == EPI_CONV2) {
                float4 g, o;
                const float2v x0 = {v[0] + bias[0], v[1] + bias[1]}, x1 = {v[2] + bias[2], v[3] + bias[3]};
                const half2v g0 = gelu16_pair(x0), g1 = gelu16_pair(x1);
                g.x = (float) g0[0]; g.y = (float) g0[1]; g.z = (float) g1[0]; g.w = (float) g1[1];
                const Conv2Row cr = conv2_row(a, m);
                if (cr.valid) {
                    if (a.aux) *(float4 *) ((float *) a.aux + (size_t) cr.out * a.ldaux + n) = g;
                    o.x = rpre[i].x + g.x; o.y = rpre[i].y + g.y; o.z = rpre[i].z + g.z; o.w = rpre[i].w + g.w;
                    *(float4 *) ((float *) a.C + (size_t) cr.out * a.ldc + n) = o;
                }
            } else if constexpr (EPI == EPI_QKV_ENC) {
                // q (bias, no scale here: the encoder scales the scores) | k (no bias); the V^T third runs in the first orientation
                const int seg = __builtin_amdgcn_readfirstlane(n0 / a.S);       // tile-uniform: BN | S
                const int c = n - seg * a.S;
                const float x[4] = {v[0] + bias[0], v[1] + bias[1], v[2] + bias[2], v[3] + bias[3]};
                if (seg == 0) put16((__half *) a.C + (size_t) m * a.ldc + c, x);
                else          put16((__half *) a.aux + (size_t) m * a.ldaux + c, x);
            } else if constexpr (EPI == EPI_CROSS_KV) {
                // columns [il][K: S | V: S]; four consecutive columns never straddle a boundary (4 | S)
                const int il = n / (2 * a.S), c = n - il * 2 * a.S;
                if (c < a.S) {
                    const float x[4] = {v[0] * a.scale, v[1] * a.scale, v[2] * a.scale, v[3] * a.scale};
                    put16((__half *) a.C + il * a.layer_stride + (size_t) m * a.ldc + c, x);
                } else {
                    const float x[4] = {v[0] + bias[0], v[1] + bias[1], v[2] + bias[2], v[3] + bias[3]};
                    put16((__half *) a.aux + il * a.layer_stride + (size_t) m * a.ldaux + (c - a.S), x);
                }
            }
        }
    }
}

// ---- transposed orientation with FULL-LINE stores.  The plain form above stores 8 bytes per lane: a wave instruction covers 16 rows x 32
// bytes, i.e. 16 partial-line requests for 512 bytes, and the epilogue of a big tile is bound by exactly that request count (measured on
// mlp.0 at M = 12 000, 192 x 256 tiles: 4.5 us of epilogue per tile with a bias-add-and-store epilogue, the same with every store aimed at
// one L2-resident patch — not HBM bandwidth, not the GELU arithmetic: scratch/lab/gemm8_lab.hip).  Here the fragments of one 16-row band
// are exchanged between lanes until a lane holds 16 consecutive bytes and a wave instruction covers 8 rows x 128 bytes (f16: two
// v_permlane16_swap per fragment pair put columns fq*4 .. of fragments j, j + 1 next to each other; then the lane halves frow < 8 / >= 8
// of every 16-lane row trade one 16-byte register through DPP row_ror:8): a quarter of the requests, pure data movement — every stored
// value is the plain form's.  Needs an even number of fragment columns in the wave tile (FN = 2, 4) and, for the segmented epilogues,
// segments that are multiples of the wave tile's 16 FN columns.
__device__ __forceinline__ uint4 ror8_u4(uint4 t) {
    t.x = (uint32_t) xor_lane<8>((int) t.x); t.y = (uint32_t) xor_lane<8>((int) t.y);
    t.z = (uint32_t) xor_lane<8>((int) t.z); t.w = (uint32_t) xor_lane<8>((int) t.w);
    return t;
}
// P, Q: this lane's 16-byte pieces of two column blocks of row frow.  Afterwards P belongs to row (frow & 7), Q to row (frow & 7) + 8,
// both in column block (frow >> 3): lanes frow < 8 keep P and get the partner's P, lanes frow >= 8 keep Q and get the partner's Q.
__device__ __forceinline__ void trade_rows8(uint4 & P, uint4 & Q, const bool lo) {
    uint4 t = lo ? Q : P;
    t = ror8_u4(t);
    if (lo) Q = t; else P = t;
}

// ---- the V^T third of the encoder's q|k|v in the first orientation, with WHOLE-LINE stores.  A lane of fragment (i, j) holds time steps
// t = tb + 4 fq + r of column c = frow (tb = the fragment's 16-step block), i.e. the 8-byte piece at V^T position tb + 4 fq' of row c, where
// fq' = fq with its two bits swapped (vt_pos): lanes fq = 0 / 2 hold the adjacent pieces 0 / 1 of the block, lanes 1 / 3 the pieces 2 / 3.
// Plain form: 64 lanes x 8 B = 16 rows x 32 B per wave instruction, and the epilogue of a big tile is bound by that request count (what the
// q|k|v launch cost over a plain epilogue at M = 12 000: 41 us against 29.5, profiles/r05b_*).  Here:
//   pair  (fragments i, i + 1):  v_permlane32_swap per dword — lanes 0-31 end up with BOTH pieces of their half of block i, lanes 32-63
//         with both pieces of their half of block i + 1: 16 B per lane, 16 rows x 64 B per instruction, half the instructions;
//   quad  (fragments i .. i + 3): two pairs, then the lane halves frow < 8 / >= 8 trade one 16-byte register through DPP row_ror:8
//         (trade_rows8): an instruction covers 8 rows x 128 B.
// Pure data movement: every stored value is the plain form's.  Needs 16-step blocks that do not straddle chunks (rows per chunk a
// multiple of 16, or one chunk) — the caller checks; a block that lies beyond M or beyond the row's Tpad is not stored, rows t >= T of a
// block that straddles the end of a chunk's valid steps land in the row's padding (finite values: clamped A rows; never read as keys).
struct VtDest { __half * p; bool ok; };
template <bool GUARD>
__device__ __forceinline__ VtDest vt_block_dest(const GemmArgs & a, const int m_blk, const int c, const int pos8) {
    const int rpc = a.rows_per_chunk > 0 ? a.rows_per_chunk : a.M;
    const int cb = m_blk / rpc, tb = m_blk - cb * rpc;
    VtDest d;
    d.ok = (!GUARD || m_blk < a.M) && tb + 16 <= a.ldaux2;
    d.p = (__half *) a.aux2 + (size_t) cb * a.chunk_stride_aux2 + (size_t) c * a.ldaux2 + tb + pos8;
    return d;
}
template <int FM, int FN, bool GUARD>
__device__ __forceinline__ void epilogue_vt_wide(const GemmArgs & a, floatx4 (&acc)[FM][FN], const int mb, const int nb, const int lane) {
    const int frow = lane & 15, fq = lane >> 4;
    const bool lo = frow < 8;
    const int pos8 = (fq & 1) << 3, up = fq >> 1;            // after a pair swap: this lane's 8 steps inside its block, and which block of the pair
    auto pack = [&](const floatx4 & v, const float bias, uint32_t (&w)[2]) {
        half4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (_Float16) pin_f32(v[r] + bias);
        const uint2 u = *(const uint2 *) &h;
        w[0] = u.x; w[1] = u.y;
    };
    auto pair16 = [&](const floatx4 & v0, const floatx4 & v1, const float bias) -> uint4 {
        uint32_t w0[2], w1[2];
        pack(v0, bias, w0); pack(v1, bias, w1);
        const auto s0 = __builtin_amdgcn_permlane32_swap(w0[0], w1[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(w0[1], w1[1], false, false);
        return make_uint4(s0[0], s1[0], s0[1], s1[1]);       // [first piece | second piece] of block i (lanes 0-31) / i + 1 (lanes 32-63)
    };
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = nb + j * 16 + frow;                    // (N is a multiple of 16 on every caller: no column guard)
        const int c = n - 2 * a.S;
        const float bias = a.bias ? a.bias[n] : 0.0f;
        int i = 0;
#pragma unroll
        for (; i + 3 < FM; i += 4) {
            uint4 P = pair16(acc[i][j], acc[i + 1][j], bias), Q = pair16(acc[i + 2][j], acc[i + 3][j], bias);
            trade_rows8(P, Q, lo);
            // lanes frow < 8 now hold blocks i / i + 1 of columns (frow & 7) [P] and (frow & 7) + 8 [Q], lanes frow >= 8 blocks i + 2 / i + 3
            const int blk = mb + (i + (lo ? 0 : 2) + up) * 16, cc = c - frow + (frow & 7);
            const VtDest d0 = vt_block_dest<GUARD>(a, blk, cc, pos8), d1 = vt_block_dest<GUARD>(a, blk, cc + 8, pos8);
            if (d0.ok) *(uint4 *) d0.p = P;
            if (d1.ok) *(uint4 *) d1.p = Q;
        }
#pragma unroll
        for (; i + 1 < FM; i += 2) {
            const uint4 P = pair16(acc[i][j], acc[i + 1][j], bias);
            const VtDest d = vt_block_dest<GUARD>(a, mb + (i + up) * 16, c, pos8);
            if (d.ok) *(uint4 *) d.p = P;
        }
#pragma unroll
        for (; i < FM; ++i) {                                // a last single fragment row: the 8-byte form
            uint32_t w[2];
            pack(acc[i][j], bias, w);
            const VtDest d = vt_block_dest<GUARD>(a, mb + i * 16, c, 0);
            if (d.ok) *(uint2 *) (d.p + ((fq & 1) << 3) + ((fq >> 1) << 2)) = make_uint2(w[0], w[1]);
        }
    }
}
// wave-uniform: may the V^T third of this launch take the whole-line form?
__device__ __forceinline__ bool vt_wide_ok(const GemmArgs & a) {
    const int rpc = a.rows_per_chunk > 0 ? a.rows_per_chunk : a.M;
    return !(a.no_glds & 16) && (rpc >= a.M || (rpc & 15) == 0) && (a.ldaux2 & 7) == 0 && (a.chunk_stride_aux2 & 7) == 0;
}

template <int EPI, int FM, int FN, bool GUARD>
__device__ __forceinline__ void epilogue_cols_wide(const GemmArgs & a, floatx4 (&acc)[FM][FN], const int mb, const int nb, const int n0, const int lane) {
    static_assert(FN % 2 == 0, "wide stores pair fragment columns");
    static_assert(EPI == EPI_F16_BIAS || EPI == EPI_F16_BIAS_GELU || EPI == EPI_Q_SCALED || EPI == EPI_F32_BIAS_RESID || EPI == EPI_CROSS_KV || EPI == EPI_QKV_ENC, "epilogue");
    if constexpr (GUARD) {                                 // a wave tile that crosses the right edge of the matrix takes the plain form (wave-uniform)
        if (nb + 16 * FN > a.N) { epilogue_cols<EPI, FM, FN, true>(a, acc, mb, nb, n0, lane); return; }
    }
    const int frow = lane & 15, fq = lane >> 4;
    const bool lo = frow < 8;
    const int r8 = frow & 7, hb = frow >> 3;
    float bias[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) b4 = *(const float4 *) (a.bias + nb + j * 16 + fq * 4);
        bias[j][0] = b4.x; bias[j][1] = b4.y; bias[j][2] = b4.z; bias[j][3] = b4.w;
    }
    if constexpr (EPI == EPI_F32_BIAS_RESID) {
        // f32 rows: a lane already holds 16 bytes (4 columns); fragments j, j + 1 are 128 consecutive bytes of a row
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mb + i * 16 + frow;
            float4 rp[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) rp[j] = *(const float4 *) (a.resid + (size_t) ((GUARD && m >= a.M) ? a.M - 1 : m) * a.ldr + nb + j * 16 + fq * 4);
#pragma unroll
            for (int j = 0; j < FN; j += 2) {
                uint4 P, Q;
                {
                    float4 o;
                    o.x = (acc[i][j][0] + bias[j][0]) + rp[j].x; o.y = (acc[i][j][1] + bias[j][1]) + rp[j].y;
                    o.z = (acc[i][j][2] + bias[j][2]) + rp[j].z; o.w = (acc[i][j][3] + bias[j][3]) + rp[j].w;
                    P = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w));
                    o.x = (acc[i][j + 1][0] + bias[j + 1][0]) + rp[j + 1].x; o.y = (acc[i][j + 1][1] + bias[j + 1][1]) + rp[j + 1].y;
                    o.z = (acc[i][j + 1][2] + bias[j + 1][2]) + rp[j + 1].z; o.w = (acc[i][j + 1][3] + bias[j + 1][3]) + rp[j + 1].w;
                    Q = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w));
                }
                trade_rows8(P, Q, lo);
                const int mr = mb + i * 16 + r8, n = nb + (j + hb) * 16 + fq * 4;
                if (!GUARD || mr < a.M)     *(uint4 *) ((float *) a.C + (size_t) mr * a.ldc + n) = P;
                if (!GUARD || mr + 8 < a.M) *(uint4 *) ((float *) a.C + (size_t) (mr + 8) * a.ldc + n) = Q;
            }
        }
        return;
    } else {
    // f16 rows.  Destination of the wave tile's 16 FN columns: one segment (wave-uniform)
    __half * dst; int ldd; int c0; float scale = 1.0f; bool use_bias = true;
    if constexpr (EPI == EPI_QKV_ENC) {
        const int seg = __builtin_amdgcn_readfirstlane(nb / a.S);
        c0 = nb - seg * a.S;
        dst = seg == 0 ? (__half *) a.C : (__half *) a.aux; ldd = seg == 0 ? a.ldc : a.ldaux;
    } else if constexpr (EPI == EPI_CROSS_KV) {
        const int il = __builtin_amdgcn_readfirstlane(nb / (2 * a.S)), c = nb - il * 2 * a.S;
        const bool isk = c < a.S;
        dst = (isk ? (__half *) a.C : (__half *) a.aux) + il * a.layer_stride; ldd = isk ? a.ldc : a.ldaux; c0 = isk ? c : c - a.S;
        scale = isk ? a.scale : 1.0f; use_bias = !isk;
    } else {
        dst = (__half *) a.C; ldd = a.ldc; c0 = nb;
        if constexpr (EPI == EPI_Q_SCALED) scale = a.scale;
    }
    const int cb = ((fq & 1) << 4) | ((fq >> 1) << 3);         // this lane's 8 columns inside a 32-column fragment pair after the swaps
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        uint4 blk[FN / 2];
#pragma unroll
        for (int j = 0; j < FN; j += 2) {
            uint32_t w[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const floatx4 v = acc[i][j + jj];
                half4 h;
                if constexpr (EPI == EPI_F16_BIAS_GELU) {
                    const float2v x0 = {v[0] + bias[j + jj][0], v[1] + bias[j + jj][1]}, x1 = {v[2] + bias[j + jj][2], v[3] + bias[j + jj][3]};
                    const half2v g0 = gelu16_pair(x0), g1 = gelu16_pair(x1);
                    h[0] = g0[0]; h[1] = g0[1]; h[2] = g1[0]; h[3] = g1[1];
                } else if constexpr (EPI == EPI_CROSS_KV) {
                    // K: acc * scale (no bias) | V: acc + bias — the plain form's two expressions, chosen per wave
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (_Float16) pin_f32(use_bias ? v[r] + bias[j + jj][r] : v[r] * scale);
                } else if constexpr (EPI == EPI_Q_SCALED) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (_Float16) pin_f32((v[r] + bias[j + jj][r]) * scale);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (_Float16) pin_f32(v[r] + bias[j + jj][r]);
                }
                const uint2 u = *(const uint2 *) &h;
                w[jj][0] = u.x; w[jj][1] = u.y;
            }
            // rows (16 lanes) 1 / 3 of fragment j <-> rows 0 / 2 of fragment j + 1
            const auto s0 = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
            blk[j / 2] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
#pragma unroll
        for (int b = 0; b < FN / 2; b += 2) {
            if constexpr (FN / 2 >= 2) {
                uint4 P = blk[b], Q = blk[b + 1];
                trade_rows8(P, Q, lo);
                const int mr = mb + i * 16 + r8, c = c0 + (b + hb) * 32 + cb;
                if (!GUARD || mr < a.M)     *(uint4 *) (dst + (size_t) mr * ldd + c) = P;
                if (!GUARD || mr + 8 < a.M) *(uint4 *) (dst + (size_t) (mr + 8) * ldd + c) = Q;
            } else {
                const int m = mb + i * 16 + frow, c = c0 + b * 32 + cb;
                if (!GUARD || m < a.M) *(uint4 *) (dst + (size_t) m * ldd + c) = blk[b];
            }
        }
    }
    }
}

// ---- the f16 results of epilogue_cols_wide LEFT IN REGISTERS (k_gemm8's deferred stores).  out[i][0] / out[i][1] are the two 16-byte pieces
// a lane stores for fragment row i — rows mb + i*16 + (frow & 7) and + 8 —, ptr the address of out[0][0]'s destination, ldd the row stride
// of the destination in elements: piece (i, h) belongs at ptr + (i*16 + h*8) * ldd.  Same expressions, same lane exchanges as the storing
// form above: the bytes that leave later are the bytes it would have stored.  Interior tiles only (no bounds checks), FN = 4.
template <int EPI, int FM, int FN>
__device__ __forceinline__ void pack_cols_wide(const GemmArgs & a, floatx4 (&acc)[FM][FN], const int mb, const int nb, const int n0, const int lane,
                                               uint4 (&out)[FM][2], __half * & ptr, int & ldd_out) {
    static_assert(FN == 4, "pack_cols_wide: wave tiles of four fragment columns");
    static_assert(EPI == EPI_F16_BIAS || EPI == EPI_F16_BIAS_GELU || EPI == EPI_Q_SCALED || EPI == EPI_CROSS_KV || EPI == EPI_QKV_ENC, "epilogue");
    const int frow = lane & 15, fq = lane >> 4;
    const bool lo = frow < 8;
    const int r8 = frow & 7, hb = frow >> 3;
    float bias[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) b4 = *(const float4 *) (a.bias + nb + j * 16 + fq * 4);
        bias[j][0] = b4.x; bias[j][1] = b4.y; bias[j][2] = b4.z; bias[j][3] = b4.w;
    }
    __half * dst; int ldd; int c0; float scale = 1.0f; bool use_bias = true;
    if constexpr (EPI == EPI_QKV_ENC) {
        const int seg = __builtin_amdgcn_readfirstlane(nb / a.S);
        c0 = nb - seg * a.S;
        dst = seg == 0 ? (__half *) a.C : (__half *) a.aux; ldd = seg == 0 ? a.ldc : a.ldaux;
    } else if constexpr (EPI == EPI_CROSS_KV) {
        const int il = __builtin_amdgcn_readfirstlane(nb / (2 * a.S)), c = nb - il * 2 * a.S;
        const bool isk = c < a.S;
        dst = (isk ? (__half *) a.C : (__half *) a.aux) + il * a.layer_stride; ldd = isk ? a.ldc : a.ldaux; c0 = isk ? c : c - a.S;
        scale = isk ? a.scale : 1.0f; use_bias = !isk;
    } else {
        dst = (__half *) a.C; ldd = a.ldc; c0 = nb;
        if constexpr (EPI == EPI_Q_SCALED) scale = a.scale;
    }
    const int cb = ((fq & 1) << 4) | ((fq >> 1) << 3);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        uint4 blk[2];
#pragma unroll
        for (int j = 0; j < FN; j += 2) {
            uint32_t w[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const floatx4 v = acc[i][j + jj];
                half4 h;
                if constexpr (EPI == EPI_F16_BIAS_GELU) {
                    const float2v x0 = {v[0] + bias[j + jj][0], v[1] + bias[j + jj][1]}, x1 = {v[2] + bias[j + jj][2], v[3] + bias[j + jj][3]};
                    const half2v g0 = gelu16_pair(x0), g1 = gelu16_pair(x1);
                    h[0] = g0[0]; h[1] = g0[1]; h[2] = g1[0]; h[3] = g1[1];
                } else if constexpr (EPI == EPI_CROSS_KV) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (_Float16) pin_f32(use_bias ? v[r] + bias[j + jj][r] : v[r] * scale);
                } else if constexpr (EPI == EPI_Q_SCALED) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (_Float16) pin_f32((v[r] + bias[j + jj][r]) * scale);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (_Float16) pin_f32(v[r] + bias[j + jj][r]);
                }
                const uint2 u = *(const uint2 *) &h;
                w[jj][0] = u.x; w[jj][1] = u.y;
            }
            const auto s0 = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
            blk[j / 2] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
        uint4 P = blk[0], Q = blk[1];
        trade_rows8(P, Q, lo);
        out[i][0] = P; out[i][1] = Q;
    }
    ptr = dst + (size_t) (mb + r8) * ldd + (c0 + hb * 32 + cb);
    ldd_out = ldd;
}

}}} // namespace wmi::k::gemm_detail
