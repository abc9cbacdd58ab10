// Block-quantised weights on gfx950 (BASELINE configs[4]: large-v3 q5_1).
//
// The reference's arithmetic for a mul_mat whose weight operand is q4_0 / q4_1 / q5_0 / q5_1 / q8_0
// (W/ggml.c:9841-9857 + type table :397-583; W/ggml-quants.c:837-870 row quantiser, :2442-3560 dots):
//
//   activations  every f32 row -> q8 blocks of 32: d = amax / 127, q = rne(x * (127 / amax)), q8_1 also s = d * sum(q)
//                (the q8_0 kinds store d as f16)
//   dot          per block  isum = sum_k w_k * q_k  (integers),  out += isum * (d_w * d_a)  [+ m_w * s_a]
//
// Kept here: the weights stay in their 18..34-byte blocks in HBM (2.7x fewer bytes per decoded token than the f16
// expansion of round 1), the activation rows are quantised with the reference's own formula, and the block dot runs on
// v_mfma_i32_32x32x32_i8 — its K is exactly one block, so isum is exact; what is left in floating point is the same
// per-block f32 scale-and-add the reference does (its AVX2 body keeps eight partial sums per output and adds them at
// the end; here there is one — an f32 rounding-order difference of ~1e-6 relative, measured in
// tests/test_oracle_quants.py).
//
// Kernels:
//   k_q8_rows   rows (f32, LayerNorm(f32) or f16) -> q8 blocks in global memory (the GEMM's A operand)      HBM-bound
//   k_qgemm     C = A_q8 . W_q^T, 128 x 128 x 64 tiles, A via global_load_lds, W tiles unpacked to int8 in LDS   VALU-bound:
//               each 32 x 32 x 32 MFMA (8 passes) is followed by 16 x (cvt, mul, fma) per lane to apply d_w * d_a;
//               the m_w * s_a terms of a K step go through one v_mfma_f32_32x32x2_f32 per output tile
//   k_qrows     <= 32 activation rows (a decode step, a beam, lock-step chunks): weight tiles streamed once from HBM,
//               rows quantised in the prologue, K split over the wavefronts of a workgroup                  HBM-bound
//   k_qembed    token embedding gather with dequantisation

#include "kernels.h"
#include <map>
#include <mutex>
#include "wave_ops.h"
#include "xattn_tail.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace wmi { namespace k {

QGeom q_geom(int qtype) {
    switch (qtype) {
        case QT_Q4_0: return {16, 4, 18, false};
        case QT_Q4_1: return {16, 4, 20, true};
        case QT_Q5_0: return {16, 8, 22, false};
        case QT_Q5_1: return {16, 8, 24, true};
        case QT_Q8_0: return {32, 4, 34, false};
        default:      return {0, 0, 0, false};
    }
}

// The fifth bits of a q5 block in the order the unpacker wants them: the file keeps bit e for element e; the tiles keep the bit of element
// 4 i + j (i, j < 4: the low-nibble half, dword i / byte j of the unpacked block) at position 8 j + i and that of element 16 + 4 i + j at
// 8 j + 4 + i, so that "bit 4 of the four bytes of dword i" is one shift and one mask on the device ((w << (4 - i)) & 0x10101010,
// (w >> i) & 0x10101010) instead of shift, mask, multiply, mask, shift per dword — the unpack is what bounds the row kernels' tile phase
// (VALU issue).  Same 32 bits, same bytes in HBM.
static uint32_t q5_bits_tile_order(const uint8_t * qh_file) {
    uint32_t q; memcpy(&q, qh_file, 4);
    uint32_t out = 0;
    for (int e = 0; e < 32; ++e) {
        const int half = e >> 4, i = (e & 15) >> 2, j = e & 3;
        out |= ((q >> e) & 1u) << (8 * j + 4 * half + i);
    }
    return out;
}

void q_repack_host(int qtype, const uint8_t * src, int64_t N, int64_t K, uint8_t * dst) {
    const QGeom g = q_geom(qtype);
    const int64_t nb = K / 32, np = K / 64, ntn = (N + 31) / 32;
    const size_t tile = q_tile_bytes(qtype);
    memset(dst, 0, (size_t) ntn * np * tile);
    for (int64_t n = 0; n < N; ++n) {
        const int64_t tn = n / 32, nl = n % 32;
        for (int64_t b = 0; b < nb; ++b) {
            const uint8_t * s = src + ((size_t) n * nb + b) * g.file_bytes;
            const int64_t tp = b / 2, lane = nl + 32 * (b % 2);
            uint8_t * t = dst + ((size_t) tn * np + tp) * tile;
            uint8_t * qs = t + (size_t) lane * g.qb, * hd = t + (size_t) 64 * g.qb + (size_t) lane * g.hb;
            switch (qtype) {
                case QT_Q4_0: memcpy(hd, s, 2);                         memcpy(qs, s + 2, 16); break;
                case QT_Q4_1: memcpy(hd, s, 4);                         memcpy(qs, s + 4, 16); break;
                case QT_Q5_0: memcpy(hd, s, 2); { const uint32_t p5 = q5_bits_tile_order(s + 2); memcpy(hd + 4, &p5, 4); } memcpy(qs, s + 6, 16); break;
                case QT_Q5_1: memcpy(hd, s, 4); { const uint32_t p5 = q5_bits_tile_order(s + 4); memcpy(hd + 4, &p5, 4); } memcpy(qs, s + 8, 16); break;
                case QT_Q8_0: memcpy(hd, s, 2);                         memcpy(qs, s + 2, 32); break;
                default: break;
            }
        }
    }
}

namespace {

typedef int      intx4  __attribute__((ext_vector_type(4)));
typedef int      intx16 __attribute__((ext_vector_type(16)));
typedef float    floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float round_f16(float x) { return __half2float(f2h(x)); }
__device__ __forceinline__ float gelu16(float x) {
    const float xh = round_f16(x);
    const float g  = 0.5f * xh * (1.0f + tanhf(0.79788456080286535587989211986876f * xh * (1.0f + 0.044715f * xh * xh)));
    return round_f16(g);
}
__device__ __forceinline__ float gelu16_fast(float x) {          // k_gemm.hip: the encoder GEMMs' form
    const float xh = round_f16(x);
    const float u  = 0.79788456080286535587989211986876f * xh * (1.0f + 0.044715f * xh * xh);
    const float t  = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * u) + 1.0f);
    return round_f16(0.5f * xh * (1.0f + t));
}

// ------------------------------------------------------------------------------------------------ block unpacking
template <int QT> struct Geo;
template <> struct Geo<QT_Q4_0> { static constexpr int QW = 4, HW = 1; static constexpr bool M = false, F16D = true;  };
template <> struct Geo<QT_Q4_1> { static constexpr int QW = 4, HW = 1; static constexpr bool M = true,  F16D = false; };
template <> struct Geo<QT_Q5_0> { static constexpr int QW = 4, HW = 2; static constexpr bool M = false, F16D = true;  };
template <> struct Geo<QT_Q5_1> { static constexpr int QW = 4, HW = 2; static constexpr bool M = true,  F16D = false; };
template <> struct Geo<QT_Q8_0> { static constexpr int QW = 8, HW = 1; static constexpr bool M = false, F16D = true;  };
template <int QT> constexpr int tile_bytes() { return 64 * 4 * (Geo<QT>::QW + Geo<QT>::HW); }


// one block -> 32 signed 8-bit integers: lo = elements 0..15, hi = elements 16..31 (four per dword, element order);
// d, m as f32 (m = 0 for the symmetric kinds)
template <int QT>
__device__ __forceinline__ void unpack(const uint32_t (&qs)[Geo<QT>::QW], const uint32_t (&hd)[Geo<QT>::HW],
                                       uint32_t (&lo)[4], uint32_t (&hi)[4], float & d, float & m) {
    const __half2 dm = *(const __half2 *) &hd[0];
    d = __low2float(dm); m = Geo<QT>::M ? __high2float(dm) : 0.0f;
    if constexpr (QT == QT_Q8_0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = qs[i]; hi[i] = qs[4 + i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t l = qs[i] & 0x0F0F0F0Fu, h = (qs[i] >> 4) & 0x0F0F0F0Fu;
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) {
                const uint32_t qh = hd[1];                   // fifth bits in tile order (q5_bits_tile_order): element 4 i + j at bit 8 j + i, 16 + 4 i + j at 8 j + 4 + i
                l |= (qh << (4 - i)) & 0x10101010u; h |= (qh >> i) & 0x10101010u;
            }
            if constexpr (QT == QT_Q4_0) {           // x - 8: flip bit 3 (offset binary -> two's complement), sign-extend the nibble
                l ^= 0x08080808u; h ^= 0x08080808u;
                const uint32_t sl = l & 0x08080808u, sh = h & 0x08080808u;
                l |= (sl << 1) | (sl << 2) | (sl << 3) | (sl << 4); h |= (sh << 1) | (sh << 2) | (sh << 3) | (sh << 4);
            }
            if constexpr (QT == QT_Q5_0) {           // x - 16: flip bit 4, sign-extend the 5-bit value
                l ^= 0x10101010u; h ^= 0x10101010u;
                const uint32_t sl = l & 0x10101010u, sh = h & 0x10101010u;
                l |= (sl << 1) | (sl << 2) | (sl << 3); h |= (sh << 1) | (sh << 2) | (sh << 3);
            }
            lo[i] = l; hi[i] = h;
        }
    }
}

// ------------------------------------------------------------------------------------------------ row quantiser
// Four consecutive values of a row per lane; a q8 block = 8 consecutive lanes.  Returns the packed quants, d (as the
// dot will use it) and s.  The reference's AVX2 body: d = amax / 127, id = 127 / amax, round to nearest even.
template <bool F16D>
__device__ __forceinline__ uint32_t quant4(float y0, float y1, float y2, float y3, float & d_out, float & s_out) {
    float amax = fmaxf(fmaxf(fabsf(y0), fabsf(y1)), fmaxf(fabsf(y2), fabsf(y3)));
    amax = fmaxf(amax, WMI_SHX(amax, 1)); amax = fmaxf(amax, WMI_SHX(amax, 2)); amax = fmaxf(amax, WMI_SHX(amax, 4));
    const float d  = amax / 127.0f;
    const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
    const int q0 = (int) rintf(y0 * id), q1 = (int) rintf(y1 * id), q2 = (int) rintf(y2 * id), q3 = (int) rintf(y3 * id);
    int sum = (q0 + q1) + (q2 + q3);
    sum += WMI_SHX(sum, 1); sum += WMI_SHX(sum, 2); sum += WMI_SHX(sum, 4);
    if (F16D) { d_out = round_f16(d); s_out = 0.0f; }
    else      { d_out = d; s_out = d * (float) sum; }
    return (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
}

// four quants of a block back as f16(d * q): the A operand of the f16 form of the GEMM (qgemm below).  One f32 product, pinned, one
// rounding: the value does not depend on which slot of an unrolled group computed it
__device__ __forceinline__ void put_deq(__half * dst, uint32_t q, float d) {
    const float x0 = (float) (int) (int8_t) (q & 0xFF) * d, x1 = (float) (int) (int8_t) ((q >> 8) & 0xFF) * d;
    const float x2 = (float) (int) (int8_t) ((q >> 16) & 0xFF) * d, x3 = (float) (int) (int8_t) (q >> 24) * d;
    const __half2 h01 = __floats2half2_rn(pin_f32(x0), pin_f32(x1)), h23 = __floats2half2_rn(pin_f32(x2), pin_f32(x3));
    uint2 pk; pk.x = *(const uint32_t *) &h01; pk.y = *(const uint32_t *) &h23;
    *(uint2 *) dst = pk;
}

// LayerNorm of a row held as MAXV float4 per lane (columns (i * 64 + lane) * 4), k_norm.hip's arithmetic
template <int MAXV>
__device__ __forceinline__ void ln_inplace(float4 (&v)[MAXV], const float4 (&gg)[MAXV], const float4 (&bb)[MAXV], int S, float eps, int lane) {
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < S) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sum += WMI_SHX(sum, o);
    const float mean = sum / (float) S;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < S) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sq += WMI_SHX(sq, o);
    const float scale = 1.0f / sqrtf(sq / (float) S + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < S) {
            v[i].x = __fadd_rn(__fmul_rn(v[i].x * scale, gg[i].x), bb[i].x);
            v[i].y = __fadd_rn(__fmul_rn(v[i].y * scale, gg[i].y), bb[i].y);
            v[i].z = __fadd_rn(__fmul_rn(v[i].z * scale, gg[i].z), bb[i].z);
            v[i].w = __fadd_rn(__fmul_rn(v[i].w * scale, gg[i].w), bb[i].w);
        }
    }
}

// SRC: 0 = f32 rows, 1 = LayerNorm of f32 rows (K <= 256 * MAXV), 2 = f16 rows
template <int MAXV, int SRC, bool F16D>
__device__ __forceinline__ void q8_rows_body(int blk, const float * __restrict__ x32, const __half * __restrict__ x16, int M, int K,
                                             const float * __restrict__ g, const float * __restrict__ b, float eps,
                                             int8_t * __restrict__ qs, float * __restrict__ dT, float * __restrict__ sT, int ldm,
                                             float * __restrict__ out32, __half * __restrict__ out16, __half * __restrict__ deq) {
    const int lane = threadIdx.x & 63;
    const int row = blk * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    if constexpr (SRC == 1) {
        const float * xr = x32 + (size_t) row * K;
        float4 v[MAXV], gg[MAXV], bb[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * 4, cc = c < K ? c : 0;
            v[i] = *(const float4 *) (xr + cc); gg[i] = *(const float4 *) (g + cc); bb[i] = *(const float4 *) (b + cc);
        }
        ln_inplace<MAXV>(v, gg, bb, K, eps, lane);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * 4;
            float d, s;
            const uint32_t q = quant4<F16D>(v[i].x, v[i].y, v[i].z, v[i].w, d, s);        // all lanes shuffle; columns past K hold zeros
            if (c < K) {
                *(uint32_t *) (qs + (size_t) row * K + c) = q;
                if ((lane & 7) == 0) { dT[(size_t) (c >> 5) * ldm + row] = d; sT[(size_t) (c >> 5) * ldm + row] = s; }
                if (deq) put_deq(deq + (size_t) row * K + c, q, d);
                if (out32) *(float4 *) (out32 + (size_t) row * K + c) = v[i];
                if (out16) {
                    __half2 h01 = __floats2half2_rn(pin_f32(v[i].x), pin_f32(v[i].y)), h23 = __floats2half2_rn(pin_f32(v[i].z), pin_f32(v[i].w));
                    uint2 pk; pk.x = *(uint32_t *) &h01; pk.y = *(uint32_t *) &h23;
                    *(uint2 *) (out16 + (size_t) row * K + c) = pk;
                }
            }
        }
    } else {
        for (int c0 = 0; c0 < K; c0 += 256 * MAXV) {
            float4 v[MAXV];
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = c0 + (i * 64 + lane) * 4, cc = c < K ? c : 0;
                if constexpr (SRC == 0) v[i] = *(const float4 *) (x32 + (size_t) row * K + cc);
                else {
                    const uint2 u = *(const uint2 *) (x16 + (size_t) row * K + cc);
                    const float2 a = __half22float2(*(const __half2 *) &u.x), bq = __half22float2(*(const __half2 *) &u.y);
                    v[i] = make_float4(a.x, a.y, bq.x, bq.y);
                }
                if (c >= K) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = c0 + (i * 64 + lane) * 4;
                float d, s;
                const uint32_t q = quant4<F16D>(v[i].x, v[i].y, v[i].z, v[i].w, d, s);
                if (c < K) {
                    *(uint32_t *) (qs + (size_t) row * K + c) = q;
                    if ((lane & 7) == 0) { dT[(size_t) (c >> 5) * ldm + row] = d; sT[(size_t) (c >> 5) * ldm + row] = s; }
                    if (deq) put_deq(deq + (size_t) row * K + c, q, d);
                }
            }
        }
    }
}

template <int MAXV, int SRC, bool F16D>
__global__ __launch_bounds__(256) void k_q8_rows(const float * __restrict__ x32, const __half * __restrict__ x16, int M, int K,
                                                 const float * __restrict__ g, const float * __restrict__ b, float eps,
                                                 int8_t * __restrict__ qs, float * __restrict__ dT, float * __restrict__ sT, int ldm,
                                                 float * __restrict__ out32, __half * __restrict__ out16, __half * __restrict__ deq) {
    q8_rows_body<MAXV, SRC, F16D>((int) blockIdx.x, x32, x16, M, K, g, b, eps, qs, dT, sT, ldm, out32, out16, deq);
}

// ------------------------------------------------------------------------------------------------ GEMM
// LDS image of an operand tile: [rows][64 bytes] int8, the 16-byte chunk c of row r stored at slot c ^ ((r >> 2) & 3):
// the MFMA fragment read (lane = (row, k half): 16 bytes) is then conflict-free per 16-lane group of ds_read_b128.
__device__ __forceinline__ uint32_t q_lds_off(int row, int chunk) { return (uint32_t) (row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4)); }

// shared epilogue of k_qgemm: element (m, n) of the accumulator, by kind
template <int EPI, bool GUARD>
__device__ __forceinline__ void q_store(const GemmArgs & a, int m, int n, float v, float bias, float rpre) {
    if (GUARD && m >= a.M) return;
    if constexpr (EPI == EPI_F16_BIAS)            ((__half *) a.C)[(size_t) m * a.ldc + n] = f2h(v + bias);
    else if constexpr (EPI == EPI_F16_BIAS_GELU)  ((__half *) a.C)[(size_t) m * a.ldc + n] = f2h(gelu16_fast(v + bias));
    else if constexpr (EPI == EPI_F32_BIAS_RESID) ((float *) a.C)[(size_t) m * a.ldc + n] = (v + bias) + rpre;
    else if constexpr (EPI == EPI_Q_SCALED)       ((__half *) a.C)[(size_t) m * a.ldc + n] = f2h((v + bias) * a.scale);
    else if constexpr (EPI == EPI_CROSS_KV) {
        const int il = n / (2 * a.S), c = n - il * 2 * a.S;
        if (c < a.S) ((__half *) a.C)[il * a.layer_stride + (size_t) m * a.ldc + c] = f2h(v * a.scale);
        else         ((__half *) a.aux)[il * a.layer_stride + (size_t) m * a.ldaux + (c - a.S)] = f2h(v + bias);
    }
}

// glds_asm / lds_addr (LDS-DMA issued from inline assembly): wave_ops.h

// All operands arrive by global_load_lds into an NST-deep ring (NST - 1 K steps in flight, counted vmcnt waits, one raw
// s_barrier per K step: an LDS-DMA in flight makes __syncthreads() drain the queue): one wavefront per row group fetches its
// RAW quantised tile, unpacks it to int8 when its K step comes up and stores it into one of two B images; the A quants (q8 rows)
// land in fragment order directly, the A scales come from the block-major arrays.  With one tile in flight (round 2a) the
// kernel waited a full memory round trip per K step: 70 - 108 us for the large-v3 projections at one chunk.
template <int QT, int BM, int EPI, int NST>
__global__ __launch_bounds__(256, 2) void k_qgemm(const GemmArgs a, const int8_t * __restrict__ Aq, const float * __restrict__ AdT,
                                                  const float * __restrict__ AsT, int ldm, const uint8_t * __restrict__ Wt) {
    constexpr int BN = 128;
    constexpr int FM = BM / 64, FN = 2;                    // 32 x 32 fragments per wavefront (2 x 2 wavefronts)
    constexpr int QW = Geo<QT>::QW, HW = Geo<QT>::HW;
    constexpr bool HAS_M = Geo<QT>::M;
    // ring stage: A int8 [BM][64] | raw W tiles of the 4 row groups: quants [4][64][QW] dwords, headers [4][HW][64] dwords |
    //             A scales d [2][BM], s [2][BM]
    constexpr int R_WQ = BM * 64, R_WH = R_WQ + 4 * 64 * QW * 4, R_AD = R_WH + 4 * HW * 64 * 4, R_AS = R_AD + 2 * BM * 4, RSTAGE = R_AS + 2 * BM * 4;
    // unpacked B image (two of them): int8 [BN][64] | d [2][BN] | m [2][BN]
    constexpr int B_D = BN * 64, B_M = B_D + 2 * BN * 4, BSTAGE = B_M + 2 * BN * 4;
    constexpr int OFF_BIMG = NST * RSTAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN, nwg = ntm * ntn;
    int wg = blockIdx.x;
    {   // XCD-aware tile order (k_gemm.hip): an XCD's L2 sees a contiguous run of tiles sharing A panels
        const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = wg / ntn, tn = wg % ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int np = a.K >> 6;

    // ---- per-lane source addresses of one stage
    constexpr int PA = BM / 64;                             // A pieces (16 rows x 64 B) per wavefront
    const int8_t * qA[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int lrow = (wave * PA + p) * 16 + (lane >> 2), ch = (lane & 3) ^ ((lrow >> 2) & 3);
        int r = m0 + lrow; if (r > a.M - 1) r = a.M - 1;
        qA[p] = Aq + (size_t) r * a.K + ch * 16;
    }
    const uint8_t * gW = Wt + ((size_t) (n0 / 32 + wave) * np) * tile_bytes<QT>();      // this wavefront's row group
    // A scales: wavefront w fetches (block w & 1, d or s by w >> 1) for rows m0 + [0, BM): BM / 64 loads of 64 floats
    const float * gSc; { int r = m0 + lane; if (r > a.M - 1) r = a.M - 1; gSc = ((wave >> 1) ? AsT : AdT) + r; }
    int sc_r1 = m0 + 64 + lane; if (sc_r1 > a.M - 1) sc_r1 = a.M - 1;
    const float * gSc1 = ((wave >> 1) ? AsT : AdT) + sc_r1;
    constexpr int LPT = PA + 1 + HW + PA;                   // loads per wavefront and stage

    const uint32_t lds0 = lds_addr(smem);
    auto issue = [&](int kt, int slot) {
        const uint32_t st = lds0 + slot * RSTAGE;
#pragma unroll
        for (int p = 0; p < PA; ++p) glds_asm<16>(qA[p] + kt * 64, st + (wave * PA + p) * 1024);
        const uint8_t * t = gW + (size_t) kt * tile_bytes<QT>();
        if constexpr (QW == 4) glds_asm<16>(t + lane * 16, st + R_WQ + wave * 1024);
        else {   // q8_0: 32 payload bytes per block: two 16-byte halves, each its own lane-linear image
            glds_asm<16>(t + lane * 32, st + R_WQ + wave * 2048);
            glds_asm<16>(t + lane * 32 + 16, st + R_WQ + wave * 2048 + 1024);
        }
#pragma unroll
        for (int h = 0; h < HW; ++h) glds_asm<4>(t + 64 * QW * 4 + lane * HW * 4 + h * 4, st + R_WH + (wave * HW + h) * 256);
        const size_t boff = (size_t) (2 * kt + (wave & 1)) * ldm;
        const int soff = (wave >> 1) ? R_AS : R_AD;
        glds_asm<4>(gSc + boff, st + soff + ((wave & 1) * BM) * 4);
        if constexpr (BM == 128) glds_asm<4>(gSc1 + boff, st + soff + ((wave & 1) * BM + 64) * 4);
    };
    constexpr int LPT_REAL = PA + (QW == 4 ? 1 : 2) + HW + PA;
    static_assert(LPT_REAL >= LPT, "load count");

    // this wavefront's raw tile of ring slot -> int8 rows + scales in B image `img`
    auto unpack_b = [&](int slot, int img) {
        const unsigned char * st = smem + slot * RSTAGE;
        unsigned char * bi = smem + OFF_BIMG + img * BSTAGE;
        uint32_t rq[QW], rh[HW];
        if constexpr (QW == 4) { const uint4 u = *(const uint4 *) (st + R_WQ + wave * 1024 + lane * 16); rq[0] = u.x; rq[1] = u.y; rq[2] = u.z; rq[3] = u.w; }
        else { const uint4 u = *(const uint4 *) (st + R_WQ + wave * 2048 + lane * 16), w2 = *(const uint4 *) (st + R_WQ + wave * 2048 + 1024 + lane * 16);
               rq[0] = u.x; rq[1] = u.y; rq[2] = u.z; rq[3] = u.w; rq[4] = w2.x; rq[5] = w2.y; rq[6] = w2.z; rq[7] = w2.w; }
#pragma unroll
        for (int h = 0; h < HW; ++h) rh[h] = *(const uint32_t *) (st + R_WH + (wave * HW + h) * 256 + lane * 4);
        uint32_t lo[4], hi[4]; float d, m;
        unpack<QT>(rq, rh, lo, hi, d, m);
        const int n = wave * 32 + (lane & 31), g = lane >> 5;                       // tile row, block of the pair
        *(uint4 *) (bi + q_lds_off(n, 2 * g))     = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *(uint4 *) (bi + q_lds_off(n, 2 * g + 1)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        ((float *) (bi + B_D))[g * BN + n] = d;
        if (HAS_M) ((float *) (bi + B_M))[g * BN + n] = m;
    };

    floatx16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int frow = lane & 31, fk = lane >> 5;
    auto compute = [&](int slot, int img) {
        const unsigned char * st = smem + slot * RSTAGE;
        const unsigned char * bi = smem + OFF_BIMG + img * BSTAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            intx4 fb[FN]; float dw[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                fb[j] = *(const intx4 *) (bi + q_lds_off(wn * 64 + j * 32 + frow, 2 * kk + fk));
                dw[j] = ((const float *) (bi + B_D))[kk * BN + wn * 64 + j * 32 + frow];
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const intx4 fa = *(const intx4 *) (st + q_lds_off(wm * (BM / 2) + i * 32 + frow, 2 * kk + fk));
                // d_a of this lane's 16 rows: rows (e & 3) + 8 (e >> 2) + 4 fk of the fragment
                float da[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = *(const float4 *) ((const float *) (st + R_AD) + kk * BM + wm * (BM / 2) + i * 32 + 8 * q + 4 * fk);
                    da[4 * q] = t.x; da[4 * q + 1] = t.y; da[4 * q + 2] = t.z; da[4 * q + 3] = t.w;
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    intx16 z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0;
                    const intx16 ia = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb[j], z, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf((float) ia[e], da[e] * dw[j], acc[i][j][e]);
                }
            }
        }
        if constexpr (HAS_M) {
            // sum over the two blocks of m_w * s_a: an f32 MFMA with K = 2 (exact f32 FMAs in block order) into the same accumulators
            float sa[FM], mw[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) sa[i] = ((const float *) (st + R_AS))[fk * BM + wm * (BM / 2) + i * 32 + frow];
#pragma unroll
            for (int j = 0; j < FN; ++j) mw[j] = ((const float *) (bi + B_M))[fk * BN + wn * 64 + j * 32 + frow];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[i], mw[j], acc[i][j], 0, 0, 0);
        }
    };

#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0) if (s0 < np) issue(s0, s0);
    for (int kt = 0; kt < np; ++kt) {
        // stage kt has landed when at most (NST - 2) stages' worth of this wavefront's loads are still outstanding
        if (np - 1 - kt >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * LPT_REAL) : "memory");
        else                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unpack_b(kt % NST, kt & 1);                          // own DMA data: ordered by the wait above
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // everyone's stage kt (A, scales) and B image kt & 1 are visible; slot (kt - 1) % NST is free
        if (kt + NST - 1 < np) issue(kt + NST - 1, (kt + NST - 1) % NST);
        compute(kt % NST, kt & 1);
    }

    // ------------------------------------------------------------------ epilogue
    // fragment (i, j): column n = nb0 + j * 32 + frow, rows m = mb + i * 32 + (e & 3) + 8 (e >> 2) + 4 fk
    const int mb = m0 + wm * (BM / 2), nbase = n0 + wn * 64;
    auto epilogue = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = nbase + j * 32 + frow;
            const float bias = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float rpre[16];
                if constexpr (EPI == EPI_F32_BIAS_RESID) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fk;
                        rpre[e] = a.resid[(size_t) ((GUARD && m >= a.M) ? a.M - 1 : m) * a.ldr + n];
                    }
                }
                if constexpr (EPI == EPI_QKV_ENC) {
                    const int seg = __builtin_amdgcn_readfirstlane((nbase + j * 32) / a.S);   // wave-uniform: 32 | S
                    const int c = n - seg * a.S;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int mrow = mb + i * 32 + 8 * q + 4 * fk;
                        if (seg == 2) {                  // V^T [chunk][S][Tpad]: four consecutive time steps -> one 8-byte store
                            const int rpc = a.rows_per_chunk > 0 ? a.rows_per_chunk : a.M;
                            const int cb = mrow / rpc, t0 = mrow - cb * rpc;
                            __half * vt = (__half *) a.aux2 + (size_t) cb * a.chunk_stride_aux2 + (size_t) c * a.ldaux2;
                            if ((!GUARD || mrow + 3 < a.M) && t0 + 3 < rpc && ((t0 & 3) == 0)) {
                                half4 v;
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = (_Float16) pin_f32(acc[i][j][4 * q + r] + bias);
                                *(half4 *) (vt + vt_pos(t0)) = v;
                            } else {
                                for (int r = 0; r < 4; ++r) {
                                    const int m = mrow + r;
                                    if (m >= a.M) continue;
                                    const int cb2 = m / rpc, t = m - cb2 * rpc;
                                    ((__half *) a.aux2)[(size_t) cb2 * a.chunk_stride_aux2 + (size_t) c * a.ldaux2 + vt_pos(t)] = f2h(acc[i][j][4 * q + r] + bias);
                                }
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int m = mrow + r;
                                if (GUARD && m >= a.M) continue;
                                const float v = acc[i][j][4 * q + r] + bias;
                                if (seg == 0) ((__half *) a.C)[(size_t) m * a.ldc + c] = f2h(v);
                                else          ((__half *) a.aux)[(size_t) m * a.ldaux + c] = f2h(v);
                            }
                        }
                    }
                } else if constexpr (EPI == EPI_QKV_DEC) {
                    const int seg = __builtin_amdgcn_readfirstlane((nbase + j * 32) / a.S);
                    const int c = n - seg * a.S;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fk;
                        if (GUARD && m >= a.M) continue;
                        const float v = acc[i][j][e];
                        __half * dst; float val;
                        if (seg == 0)      { dst = (__half *) a.C    + (size_t) m * a.ldc;    val = (v + bias) * a.scale; }
                        else if (seg == 1) { dst = (__half *) a.aux  + (size_t) m * a.ldaux;  val = v * a.scale; }
                        else               { dst = (__half *) a.aux2 + (size_t) m * a.ldaux2; val = v + bias; }
                        dst[c] = f2h(val);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fk;
                        q_store<EPI, GUARD>(a, m, n, acc[i][j][e], bias, EPI == EPI_F32_BIAS_RESID ? rpre[e] : 0.0f);
                    }
                }
            }
        }
    };
    if (m0 + BM <= a.M) epilogue(std::false_type{}); else epilogue(std::true_type{});
}

template <int QT, int BM, int EPI, int NST>
void launch_qgemm(const GemmArgs & a, Q8Rows A, const uint8_t * Wt, hipStream_t st) {
    constexpr int BN = 128;
    constexpr int QW = Geo<QT>::QW, HW = Geo<QT>::HW;
    constexpr size_t rstage = (size_t) BM * 64 + 4 * 64 * QW * 4 + 4 * HW * 64 * 4 + 4 * BM * 4;
    constexpr size_t bstage = (size_t) BN * 64 + 4 * BN * 4;
    const size_t smem = NST * rstage + 2 * bstage;
    static std::atomic<uint64_t> lds_ok{0};
    allow_full_lds((const void *) k_qgemm<QT, BM, EPI, NST>, lds_ok);
    const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
    hipLaunchKernelGGL((k_qgemm<QT, BM, EPI, NST>), dim3(ntm * ntn), dim3(256), smem, st, a, A.qs, A.d, A.s, A.ldm, Wt);
}

template <int QT, int EPI>
void qgemm_tile(const GemmArgs & a, Q8Rows A, const uint8_t * Wt, hipStream_t st) {
    // 64-row tiles: 158 VGPRs, three workgroups per CU with a 3-deep ring (53 KB of LDS each): the per-block scaling is a chain
    // MFMA -> 48 VALU per fragment, so a SIMD wants several wavefronts to interleave; 128-row tiles (256 VGPRs, two per CU) only
    // where the grid is large enough to keep them busy anyway
    static const int force_bm = getenv("WMI_QGEMM_BM") ? atoi(getenv("WMI_QGEMM_BM")) : 0;       // A/B knobs
    static const int nst64 = getenv("WMI_QGEMM_NST") ? atoi(getenv("WMI_QGEMM_NST")) : 3;
    const long t128 = (long) ((a.M + 127) / 128) * (a.N / 128);
    const bool big = force_bm ? force_bm == 128 : t128 >= 1536;
    if (big) launch_qgemm<QT, 128, EPI, 3>(a, A, Wt, st);
    else if (nst64 == 4) launch_qgemm<QT, 64, EPI, 4>(a, A, Wt, st);
    else launch_qgemm<QT, 64, EPI, 3>(a, A, Wt, st);
}

template <int QT>
void qgemm_epi(int epi, const GemmArgs & a, Q8Rows A, const uint8_t * Wt, hipStream_t st) {
    switch (epi) {
        case EPI_F16_BIAS:       qgemm_tile<QT, EPI_F16_BIAS>(a, A, Wt, st); break;
        case EPI_F16_BIAS_GELU:  qgemm_tile<QT, EPI_F16_BIAS_GELU>(a, A, Wt, st); break;
        case EPI_F32_BIAS_RESID: qgemm_tile<QT, EPI_F32_BIAS_RESID>(a, A, Wt, st); break;
        case EPI_QKV_ENC:        qgemm_tile<QT, EPI_QKV_ENC>(a, A, Wt, st); break;
        case EPI_QKV_DEC:        qgemm_tile<QT, EPI_QKV_DEC>(a, A, Wt, st); break;
        case EPI_CROSS_KV:       qgemm_tile<QT, EPI_CROSS_KV>(a, A, Wt, st); break;
        case EPI_Q_SCALED:       qgemm_tile<QT, EPI_Q_SCALED>(a, A, Wt, st); break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------------ rows (decode)
// <= 8 NR4 activation rows against a quantised matrix.  The activations are the MFMA's row operand (rows >= n read row 0 and
// are never stored), the 32 weight rows of a row group its columns: lane (col = lane % 32) keeps out[r] for rows
// r = (e & 3) + 8 (e >> 2) + 4 (lane / 32), e < 4 NR4.  A workgroup of NW wavefronts owns one row group at a time; K is split
// over its wavefronts (wavefront w: tile pairs w, w + NW, ...: at most CH of them, all requested before the prologue), the
// partial sums meet in LDS and are added in wavefront order.
// SRC: 0 f32 rows, 1 LayerNorm of f32 rows (K <= 1536), 2 f16 rows, 3 the combined partials of the split cross-attention
// (GemvArgs::comb_*: o / l per head, f32 — what attn_cross_combine writes with out32).
constexpr int PF_WG = 64;                                   // prefetch workgroups appended to a k_qrows grid (multiple of 8: XCD affinity)

template <int QT, int NR4, int SRC, int NW, int CHX = 0>
// (8 wavefronts, <= 8 rows: held to 128 VGPRs = two workgroups per CU — the vocabulary projection's grid is sized for that)
__global__ __launch_bounds__(NW * 64, (NW == 8 && NR4 == 1) ? 4 : 1) void k_qrows(const GemvArgs a, const float * __restrict__ a32, const uint8_t * __restrict__ Wt) {
    constexpr int QW = Geo<QT>::QW, HW = Geo<QT>::HW;
    constexpr bool HAS_M = Geo<QT>::M, F16D = Geo<QT>::F16D;
    constexpr int R8 = NR4 * 8;                             // row slots
    constexpr int CH = CHX ? CHX : NW == 16 ? 5 : NW == 8 ? 3 : 5;      // weight tiles in flight per wavefront (K = 5120 / 1280: everything at once; CHX: a K part of 40 tiles on 16 wavefronts)
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // The arguments on the path to the first vector loads, in one batch of scalar loads (kernels.h: WMI_ARG_NOW).  In-kernel stamps
    // (profiles/r05c_large_v3_q5_1_kqrows_preamble.txt) had found the prologue reached 1.7 us after the wavefront's start — behind six
    // DEPENDENT kernarg round trips, a vmcnt(0) on residual + pending partial and four integer divisions — and the activation row
    // arriving at +2.4..3.0 us of a 5.8 us launch.
    WMI_ARG_NOW(a.x32); WMI_ARG_NOW(a.ln_g); WMI_ARG_NOW(a.ln_b); WMI_ARG_NOW(a.a16); WMI_ARG_NOW(a.n); WMI_ARG_NOW(a.K); WMI_ARG_NOW(a.N);
    WMI_ARG_NOW(a.bias); WMI_ARG_NOW(a.epi); WMI_ARG_NOW(a.resid); WMI_ARG_NOW(a.ldr); WMI_ARG_NOW(a.rows); WMI_ARG_NOW(a.row_off);
    WMI_ARG_NOW(a.comb_o); WMI_ARG_NOW(a.comb_l); WMI_ARG_NOW(a.comb_ns); WMI_ARG_NOW(a.comb_m); WMI_ARG_NOW(a.lanes); WMI_ARG_NOW(a.step_stride);
    WMI_ARG_NOW(a.pf_ptr); WMI_ARG_NOW(a.stamps); WMI_ARG_NOW(gridDim.x); WMI_ARG_NOW(a.ksplit); WMI_ARG_NOW(a.pend); WMI_ARG_NOW(a32); WMI_ARG_NOW(Wt);
    const unsigned long long ts0 = stamp_t0(a.stamps);     // probe (wmi_step_stamps): entry, activation rows quantised, tiles multiplied, end
    unsigned long long tm1 = 0, tm2 = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, n = a.n, nb = K >> 5, np = K >> 6;
    const int lda = K + 16;                                  // bytes per quantised row in LDS (+16: the fragment reads of 16 rows spread over the banks)
    int8_t * sq = (int8_t *) smem;                           // [n][lda]
    float  * sd = (float *) (smem + (((size_t) n * lda + 15) & ~(size_t) 15));     // [nb][R8]
    float  * ss = sd + (size_t) nb * R8;                     // [nb][R8]
    float  * red = ss + (size_t) nb * R8;                    // [NW][32][R8]

    const int ngroups = (a.N + 31) >> 5;
    // The next launch's weights.  A dependent chain of launches pays the HBM first-byte latency (~2 us) in every one of them — a step
    // streams 846 MB of large-v3 q5_1, nothing stays cached from the previous token.  PF_WG extra workgroups (the grids here are
    // 40-160 workgroups on 256 CUs: they land on idle CUs) touch one dword per 128-byte line of the next matrix and leave; the
    // lines are in L2 / Infinity Cache when the next launch asks for them.  Extra workgroup j takes the row groups g = j (mod
    // PF_WG): with grids that are multiples of 8 it shares its XCD, i.e. its L2, with the workgroup that will stream group g.
    // (Issued from the streaming workgroups themselves the requests sat in front of the prologue's loads in the in-order
    // vmcnt queue: +1 us per launch instead of -1.)
    const int nmain = a.pf_ptr ? (int) gridDim.x - PF_WG : (int) gridDim.x;
    // K split over workgroups (GemvArgs::ksplit): workgroup = (row group, K part); tiles [tp0, tp0 + npq) of every row group
    // (ksplit is a power of two: shifts, not divisions — every instruction in front of the first activation load is latency the whole
    //  chain pays)
    const int ks = a.ksplit > 1 ? a.ksplit : 1, ksh = 31 - __builtin_clz((unsigned) ks), kq = (int) blockIdx.x & (ks - 1);
    const int npq = np >> ksh, tp0 = kq * npq, rstride = nmain >> ksh;
    if ((int) blockIdx.x >= nmain) {
        // every line of this workgroup's share is requested before the first one is waited for: as `acc ^= load` in a loop hipcc waited
        // vmcnt(0) per iteration (ISA dump), i.e. one HBM round trip per line and thread — the five groups per workgroup of an N = 4 S
        // matrix took ~10 us, twice the launch that was supposed to hide them.  The destination register is never read.
        const uint8_t * pf = (const uint8_t *) a.pf_ptr;
        uint32_t junk = 0;
        for (uint32_t g = blockIdx.x - nmain; g < a.pf_groups; g += PF_WG) {
            const uint8_t * gp = pf + (size_t) g * a.pf_group_bytes;
            for (uint32_t off = (uint32_t) tid * 128u; off < a.pf_group_bytes; off += (uint32_t) NT * 128u)
                asm volatile("global_load_dword %0, %1, off" : "+v"(junk) : "v"(gp + off) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(junk) :: "memory");
        return;
    }
    int rg = (int) blockIdx.x >> ksh;
    const int rg_first = rg;

    // ---- epilogue operands of this workgroup's first row group (bias, residual, cache slot): they depend on nothing this launch
    // computes, so they are requested before everything else — loaded inside the epilogue they were one more dependent round trip
    // (~1 us of a ~5 us launch, 8 launches per decoder layer of the one-row step)
    float bias_pre = 0.0f, resid_pre = 0.0f; int ro_pre = 0;
    {
        const int nl = tid & 31, r = tid >> 5, nf = rg * 32 + nl;
        if (tid < 32 * R8 && r < n && nf < a.N && rg < ngroups && kq == 0) {
            if (a.bias) bias_pre = a.bias[nf];
            if (a.epi == EPI_F32_BIAS_RESID) {
                // (straight-line: the pending partial is read from a valid address either way and selected afterwards)
                // (the sum with a pending K-split partial is a vmcnt(0) right here — a memory round trip in front of the tile loads: only
                //  the one launch per layer that has a partial pending takes that branch; as "load both, select" every residual launch paid it)
                if (a.pend) resid_pre = a.resid[(size_t) r * a.ldr + nf] + a.pend[(size_t) r * a.ldr + nf];
                else        resid_pre = a.resid[(size_t) r * a.ldr + nf];
            }
            if (a.row_off) ro_pre = a.lanes ? a.row_off[r * a.step_stride] : *a.row_off;
        }
    }

    // ---- first weight tiles of this wavefront (independent of the activations)
    uint32_t wq[CH][QW], wh[CH][HW];
    auto load_tiles = [&](int g, int c0) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            int tp = wave + NW * (c0 + u); if (tp > npq - 1) tp = npq - 1;
            tp += tp0;
            const uint8_t * t = Wt + ((size_t) g * np + tp) * tile_bytes<QT>();
            if constexpr (QW == 4) { const uint4 v = *(const uint4 *) (t + lane * 16); wq[u][0] = v.x; wq[u][1] = v.y; wq[u][2] = v.z; wq[u][3] = v.w; }
            else { const uint4 v = *(const uint4 *) (t + lane * 32), w = *(const uint4 *) (t + lane * 32 + 16);
                   wq[u][0] = v.x; wq[u][1] = v.y; wq[u][2] = v.z; wq[u][3] = v.w; wq[u][4] = w.x; wq[u][5] = w.y; wq[u][6] = w.z; wq[u][7] = w.w; }
            if constexpr (HW == 2) { const uint2 h = *(const uint2 *) (t + 64 * QW * 4 + lane * 8); wh[u][0] = h.x; wh[u][1] = h.y; }
            else wh[u][0] = *(const uint32_t *) (t + 64 * QW * 4 + lane * 4);
        }
    };
    if (rg < ngroups) load_tiles(rg, 0);
    // ---- prologue: the activation rows as q8 blocks in LDS
    if constexpr (SRC == 1) {
        // LayerNorm needs the statistics of the whole row, the quantiser only a 256-column slice: (row, slice) tasks are spread over
        // the wavefronts, each task recomputes the row's mean / variance from L2 (ln_inplace's arithmetic and order) and
        // normalises + quantises its own slice — one wavefront walking all slices of a row cost ~1 us more per launch, three
        // launches per decoder layer
        constexpr int MAXV = 6;                              // K <= 1536
        const int nsl = (K + 255) >> 8;
        if (n * nsl > 2 * NW) {
            // many rows (lock-step chunks, beams): a row per wavefront (rows w, w + NW, ...), statistics AND all of its slices from the one
            // copy of the row in registers; the gain / bias vectors arrive two slices at a time.  (Before: statistics per row, a barrier,
            // then (row, slice) tasks that loaded their 256 columns of x, gain and bias again — five dependent L2 round trips per wavefront
            // at 8 rows x 1280 columns, ~1.5 us of every LayerNorm launch of a lock-step / beam step.)  Same arithmetic, same order.
            for (int r = wave; r < n; r += NW) {
                const int src = a.rows ? a.rows[r] : r;
                const float * xr = a.x32 + (size_t) src * K;
                float4 v[MAXV];
#pragma unroll
                for (int i = 0; i < MAXV; ++i) { const int c = (i * 64 + lane) * 4; v[i] = *(const float4 *) (xr + (c < K ? c : 0)); }
                {   // pending K-split partial of the row (GemvArgs::pend): x = x + p, all loads in flight together
                    const float * pr = (a.pend ? a.pend : a.x32) + (size_t) src * K;
                    float4 pv[MAXV];
#pragma unroll
                    for (int i = 0; i < MAXV; ++i) { const int c = (i * 64 + lane) * 4; pv[i] = *(const float4 *) (pr + (c < K ? c : 0)); }
                    if (a.pend) {
#pragma unroll
                        for (int i = 0; i < MAXV; ++i) { v[i].x += pv[i].x; v[i].y += pv[i].y; v[i].z += pv[i].z; v[i].w += pv[i].w; }
                    }
                }
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int c = (i * 64 + lane) * 4;
                    if (c < K) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
                }
                _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sum += WMI_SHX(sum, o);
                const float mean = sum / (float) K;
                float sqs = 0.0f;
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int c = (i * 64 + lane) * 4;
                    if (c < K) {
                        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                        sqs += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
                    }
                }
                _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sqs += WMI_SHX(sqs, o);
                const float scale = 1.0f / sqrtf(sqs / (float) K + a.eps);
#pragma unroll
                for (int i0 = 0; i0 < MAXV; i0 += 2) {
                    if (i0 < nsl) {                              // wave-uniform
                        float4 gg[2], bb[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int cs = ((i0 + u) * 64 + lane) * 4, ccs = cs < K ? cs : 0;
                            gg[u] = *(const float4 *) (a.ln_g + ccs); bb[u] = *(const float4 *) (a.ln_b + ccs);
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            if (i0 + u < nsl) {                  // wave-uniform
                                const int cs = ((i0 + u) * 64 + lane) * 4;
                                float4 y = v[i0 + u];
                                y.x = __fadd_rn(__fmul_rn(y.x * scale, gg[u].x), bb[u].x); y.y = __fadd_rn(__fmul_rn(y.y * scale, gg[u].y), bb[u].y);
                                y.z = __fadd_rn(__fmul_rn(y.z * scale, gg[u].z), bb[u].z); y.w = __fadd_rn(__fmul_rn(y.w * scale, gg[u].w), bb[u].w);
                                if (cs >= K) y = make_float4(0.f, 0.f, 0.f, 0.f);
                                float d, sv;
                                const uint32_t q = quant4<F16D>(y.x, y.y, y.z, y.w, d, sv);
                                if (cs < K) {
                                    *(uint32_t *) (sq + (size_t) r * lda + cs) = q;
                                    if ((lane & 7) == 0) { sd[(cs >> 5) * R8 + r] = d; ss[(cs >> 5) * R8 + r] = sv; }
                                }
                            }
                        }
                    }
                }
            }
        } else                                               // few rows: a (row, slice) task per wavefront, the row's statistics recomputed by each
        for (int task = wave; task < n * nsl; task += NW) {
#ifdef WMI_QROWS_PROBE
            if (a.stamps && !tm1) tm1 = wall_clock64();          // (probe build) mark 1 = the prologue's code has been reached
#endif
            const int r = n == 1 ? 0 : task / nsl, sl = task - r * nsl;
            const int src = a.rows ? a.rows[r] : r;
            const float * xr = a.x32 + (size_t) src * K;
            float4 v[MAXV];
#pragma unroll
            for (int i = 0; i < MAXV; ++i) { const int c = (i * 64 + lane) * 4; v[i] = *(const float4 *) (xr + (c < K ? c : 0)); }
            const float * pr = (a.pend ? a.pend : a.x32) + (size_t) src * K;       // pending K-split partial of the row (GemvArgs::pend)
            float4 pv[MAXV];
#pragma unroll
            for (int i = 0; i < MAXV; ++i) { const int c = (i * 64 + lane) * 4; pv[i] = *(const float4 *) (pr + (c < K ? c : 0)); }
            const int cs = (sl * 64 + lane) * 4, ccs = cs < K ? cs : 0;
            const float4 gg = *(const float4 *) (a.ln_g + ccs), bb = *(const float4 *) (a.ln_b + ccs);
            if (a.pend) {
#pragma unroll
                for (int i = 0; i < MAXV; ++i) { v[i].x += pv[i].x; v[i].y += pv[i].y; v[i].z += pv[i].z; v[i].w += pv[i].w; }
            }
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < K) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
                else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sum += WMI_SHX(sum, o);
#ifdef WMI_QROWS_PROBE
            if (a.stamps && !tm2) { asm volatile("" :: "v"(sum)); tm2 = wall_clock64(); }      // (probe build) mark 2 = the row has arrived
#endif
            const float mean = sum / (float) K;
            float sqs = 0.0f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < K) {
                    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                    sqs += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
                }
            }
            _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sqs += WMI_SHX(sqs, o);
            const float scale = 1.0f / sqrtf(sqs / (float) K + a.eps);
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < MAXV; ++i) if (i == sl) y = v[i];                      // wave-uniform select: the array stays in registers
            y.x = __fadd_rn(__fmul_rn(y.x * scale, gg.x), bb.x); y.y = __fadd_rn(__fmul_rn(y.y * scale, gg.y), bb.y);
            y.z = __fadd_rn(__fmul_rn(y.z * scale, gg.z), bb.z); y.w = __fadd_rn(__fmul_rn(y.w * scale, gg.w), bb.w);
            if (cs >= K) y = make_float4(0.f, 0.f, 0.f, 0.f);
            float d, sv;
            const uint32_t q = quant4<F16D>(y.x, y.y, y.z, y.w, d, sv);
            if (cs < K) {
                *(uint32_t *) (sq + (size_t) r * lda + cs) = q;
                if ((lane & 7) == 0) { sd[(cs >> 5) * R8 + r] = d; ss[(cs >> 5) * R8 + r] = sv; }
            }
        }
    } else {
        // blocks quantise independently: (row, 256-column slice) pairs spread over all wavefronts
        const int nsl = ((K + 255) >> 8) >> ksh, sl0 = kq * nsl;      // (a K part quantises its own slices only)
        for (int sl = wave; sl < n * nsl; sl += NW) {
            const int r = n == 1 ? 0 : sl / nsl, c = (sl0 + sl - r * nsl) * 256 + lane * 4, cc = c < K ? c : 0;
            const int src = a.rows ? a.rows[r] : r;
            float4 v;
            if constexpr (SRC == 0) v = *(const float4 *) (a32 + (size_t) src * K + cc);
            else if constexpr (SRC == 2) {
                const uint2 u = *(const uint2 *) (a.a16 + (size_t) src * K + cc);
                const float2 p = __half22float2(*(const __half2 *) &u.x), q2 = __half22float2(*(const __half2 *) &u.y);
                v = make_float4(p.x, p.y, q2.x, q2.y);
            } else {                                         // o / l of head cc / 64 over the key slices (k_xattn_combine's arithmetic)
                const int H = K >> 6, h = cc >> 6, dd = cc & 63, ns = a.comb_ns;
                const size_t row = (size_t) src * H + h;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f); double l = 0.0;
                if (ns == 8) {
                    float4 po[8]; float pl[8], pm[8];
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2) {
                        po[s2] = *(const float4 *) (a.comb_o + (row * 8 + s2) * 64 + dd); pl[s2] = a.comb_l[row * 8 + s2];
                        pm[s2] = a.comb_m ? a.comb_m[row * 8 + s2] : 0.0f;
                    }
                    float M = -INFINITY;
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2) M = fmaxf(M, pm[s2]);
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2) {
                        const float w = !a.comb_m ? 1.0f : pm[s2] > -INFINITY ? __expf(pm[s2] - M) : 0.0f;
                        o.x += po[s2].x * w; o.y += po[s2].y * w; o.z += po[s2].z * w; o.w += po[s2].w * w; l += (double) pl[s2] * (double) w;
                    }
                } else {
                    float M = -INFINITY;
                    if (a.comb_m) for (int s2 = 0; s2 < ns; ++s2) M = fmaxf(M, a.comb_m[row * ns + s2]);
                    for (int s2 = 0; s2 < ns; ++s2) {
                        const float ms = a.comb_m ? a.comb_m[row * ns + s2] : 0.0f;
                        const float w = !a.comb_m ? 1.0f : ms > -INFINITY ? __expf(ms - M) : 0.0f;
                        const float4 t = *(const float4 *) (a.comb_o + (row * ns + s2) * 64 + dd);
                        o.x += t.x * w; o.y += t.y * w; o.z += t.z * w; o.w += t.w * w; l += (double) a.comb_l[row * ns + s2] * (double) w;
                    }
                }
                const float inv = (float) (1.0 / l);
                v = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
            }
            if (c >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float d, s;
            const uint32_t q = quant4<F16D>(v.x, v.y, v.z, v.w, d, s);
            if (c < K) {
                *(uint32_t *) (sq + (size_t) r * lda + c) = q;
                if ((lane & 7) == 0) { sd[(c >> 5) * R8 + r] = d; ss[(c >> 5) * R8 + r] = s; }
            }
        }
    }
    // scale slots of absent rows: finite zeros (they are multiplied, never stored)
    if (n < R8) for (int e = tid; e < nb * R8; e += NT) { if ((e % R8) >= n) { sd[e] = 0.0f; ss[e] = 0.0f; } }
    __syncthreads();
#ifndef WMI_QROWS_PROBE
    if (a.stamps) tm1 = wall_clock64();
#endif

    const int arow = (lane & 31) < n ? (lane & 31) : 0;     // activation row this lane feeds the MFMA with
    const int fk = lane >> 5;
    const int8_t * afrag = sq + (size_t) arow * lda + fk * 16;

    for (; rg < ngroups; rg += rstride) {
        float out[4 * NR4];
#pragma unroll
        for (int e = 0; e < 4 * NR4; ++e) out[e] = 0.0f;
        for (int c0 = 0; wave + NW * c0 < npq; c0 += CH) {
            if (c0 != 0) load_tiles(rg, c0);                 // (the first tiles of a group: requested before the prologue, resp. before the previous group's reduction)
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int tpl = wave + NW * (c0 + u), tp = tp0 + tpl;
                if (tpl < npq) {                             // wave-uniform
                    uint32_t lo[4], hi[4]; float d, m;
                    unpack<QT>(wq[u], wh[u], lo, hi, d, m);
                    // lane (n, g) unpacked block 2 tp + g; the MFMA of block 2 tp wants elements 0..15 of it on lanes < 32 and 16..31 on
                    // lanes >= 32: exchange the upper half of the even block with the lower half of the odd one
                    float d0 = d, d1 = d, m0 = m, m1 = m;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const auto r2 = __builtin_amdgcn_permlane32_swap(lo[i], hi[i], false, false);
                        lo[i] = r2[0]; hi[i] = r2[1];
                    }
                    { const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d1), false, false); d0 = __uint_as_float(r2[0]); d1 = __uint_as_float(r2[1]); }
                    if (HAS_M) { const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m1), false, false); m0 = __uint_as_float(r2[0]); m1 = __uint_as_float(r2[1]); }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int b = 2 * tp + h;
                        const intx4 fa = *(const intx4 *) (afrag + b * 32);
                        intx4 fb; if (h == 0) { fb[0] = lo[0]; fb[1] = lo[1]; fb[2] = lo[2]; fb[3] = lo[3]; } else { fb[0] = hi[0]; fb[1] = hi[1]; fb[2] = hi[2]; fb[3] = hi[3]; }
                        intx16 z;
#pragma unroll
                        for (int e = 0; e < 16; ++e) z[e] = 0;
                        const intx16 ia = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, z, 0, 0, 0);
                        const float dwv = h == 0 ? d0 : d1, mwv = h == 0 ? m0 : m1;
#pragma unroll
                        for (int q = 0; q < NR4; ++q) {
                            const float4 da = *(const float4 *) (sd + (size_t) b * R8 + 8 * q + 4 * fk);
                            const float dav[4] = {da.x, da.y, da.z, da.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) out[4 * q + r] = fmaf((float) ia[4 * q + r], dav[r] * dwv, out[4 * q + r]);
                            if (HAS_M) {
                                const float4 sa = *(const float4 *) (ss + (size_t) b * R8 + 8 * q + 4 * fk);
                                const float sav[4] = {sa.x, sa.y, sa.z, sa.w};
#pragma unroll
                                for (int r = 0; r < 4; ++r) out[4 * q + r] = fmaf(mwv, sav[r], out[4 * q + r]);
                            }
                        }
                    }
                }
            }
        }
        if (a.stamps && !tm2) { asm volatile("" :: "v"(out[0])); tm2 = wall_clock64(); }
        // the next row group's first tiles go out before this one is reduced (the vocabulary projection walks ~1.6 groups per workgroup)
        const int rgn = rg + rstride;
        const bool more = rgn < ngroups;
        if (more) load_tiles(rgn, 0);
        // ---- K-split partials of the wavefronts, added in wavefront order
        if (rg != rg_first) __syncthreads();                    // red is reused
#pragma unroll
        for (int q = 0; q < NR4; ++q)
            *(float4 *) (red + ((size_t) (wave * 32 + (lane & 31)) * R8 + 8 * q + 4 * fk)) = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
        __syncthreads();
        for (int e = tid; e < 32 * R8; e += NT) {
            const int nl = e & 31, r = e >> 5;
            if (r >= n) continue;
            const int nf = rg * 32 + nl;
            if (nf >= a.N) continue;
            float v = red[(size_t) nl * R8 + r];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += red[(size_t) (w * 32 + nl) * R8 + r];
            if (kq > 0) { a.kpart[(size_t) r * a.N + nf] = v; continue; }         // upper K part: the raw sums (GemvArgs::ksplit)
            const bool pre = rg == rg_first && e == tid;                  // this thread's prefetched element
            const float bias = pre ? bias_pre : (a.bias ? a.bias[nf] : 0.0f);
            switch (a.epi) {
                case EPI_F16_BIAS:       ((__half *) a.C)[(size_t) r * a.ldc + nf] = f2h(v + bias); break;
                case EPI_F16_BIAS_GELU:  ((__half *) a.C)[(size_t) r * a.ldc + nf] = f2h(gelu16(v + bias)); break;
                case EPI_F32_BIAS_RESID: ((float *) a.C)[(size_t) r * a.ldc + nf] = (v + bias) + (pre ? resid_pre : a.pend ? a.resid[(size_t) r * a.ldr + nf] + a.pend[(size_t) r * a.ldr + nf] : a.resid[(size_t) r * a.ldr + nf]); break;
                case EPI_Q_SCALED:       ((__half *) a.C)[(size_t) r * a.ldc + nf] = f2h((v + bias) * a.scale); break;
                case EPI_QKV_DEC: {
                    const int seg = nf / a.S, c = nf - seg * a.S;
                    const int ro = pre ? ro_pre : (a.row_off ? (a.lanes ? a.row_off[r * a.step_stride] : *a.row_off) : 0);
                    const int64_t crow = a.lanes ? (int64_t) r * a.cache_row_stride : 0;
                    const int slot = a.lanes ? ro : r + ro;
                    if (seg == 0)      ((__half *) a.C)[(size_t) r * a.ldc + c] = f2h((v + bias) * a.scale);
                    else if (seg == 1) ((__half *) a.aux)[crow + (size_t) slot * a.ldaux + c] = f2h(v * a.scale);
                    else               ((__half *) a.aux2)[crow + (size_t) slot * a.ldaux2 + c] = f2h(v + bias);
                } break;
                case EPI_LOGITS:         ((float *) a.C)[(size_t) r * a.ldc + nf] = v; break;
                default: break;
            }
        }
    }
    stamp_end(a.stamps, a.stamp_slot, (int) blockIdx.x * NW + wave, ts0, tm1, tm2);
}

// ------------------------------------------------------------------------------------------------ cross-attention, query projected inside
// One launch for "LayerNorm + cross query" and "scores, soft-max numerators, P.V of a key slice" of a block-quantised model (the f16 models'
// k_xattn_fused<NC, true>, k_attn.hip): a (row, head, slice) workgroup projects ITS head's 64 query values — two row groups of W_cq — and
// goes on with its key slice (xattn_tail.h).  The eight slices of a head repeat the projection: 8 x the head's 50 KB of tiles from L2 /
// Infinity Cache against one launch (body + boundary, 6.4 us of the ~49 us decoder layer of large-v3 q5_1) less: ~3 us per layer net
// (the step 1 540 -> 1 450 us, base.en q4_0 240 -> 222, medium q4_1 1 055 -> 965).
// Per query value the operations and their order are k_qrows': LayerNorm and q8 blocks per 256-column slice (the "few rows" form), tile
// pairs w, w + NV, ... on wavefront w in order, partial sums of the NV wavefronts added in their order — NV is the wavefront count
// qrows_nw() picks for this K (4 up to 8 tile pairs, else 8).  The key-slice part runs on wavefronts 0..3 (its sums are laid out for
// four); with NV = 8 the other four only keep the barriers company.  Same bits as the two-launch form (tests/test_gpu_variants.py),
// whose switch is WMI_Q_XATTN_TWO_LAUNCHES.
// pf_ptr: the next weight-streaming launch's matrix (see k_qrows): a dword per 128-byte line, requested behind everything this workgroup
// needs; group g goes to the workgroups with linear id = g (mod pf_groups): a multiple of 8 groups keeps the XCD of the launch that
// streams the group.
template <int QT, int NV>
__global__ __launch_bounds__(64 * NV) void k_xattn_fused_q(const float * __restrict__ x32, const float * __restrict__ ln_g, const float * __restrict__ ln_b,
                                                           float eps, const uint8_t * __restrict__ Wt, const float * __restrict__ bq, float qscale, int S,
                                                           const __half * __restrict__ kc, const __half * __restrict__ vc, int T, int ks, int ns,
                                                           float * __restrict__ pmax, float * __restrict__ part_o, float * __restrict__ part_l,
                                                           int64_t kv_row_stride, int head_major,
                                                           const uint8_t * __restrict__ pf_ptr, uint32_t pf_groups, uint32_t pf_group_bytes, const Stamp sp) {
    constexpr int QW = Geo<QT>::QW, HW = Geo<QT>::HW;
    constexpr bool HAS_M = Geo<QT>::M, F16D = Geo<QT>::F16D;
    constexpr int CHV = NV == 4 ? 2 : 3;                    // tile pairs per wavefront and row group (<= 8, resp. <= 24 pairs per row)
    constexpr int MAXV = 6;                                 // K <= 1536
    constexpr uint32_t NT = 64 * NV;
    __shared__ __attribute__((aligned(16))) int8_t sq[1536 + 16];
    __shared__ float sd[48], ss[48];
    __shared__ float redq[NV][2][32];
    __shared__ float qs[64];
    __shared__ float red[4], lred[4];
    __shared__ float ored[4][64];
    // every argument in one batch of scalar loads (kernels.h: WMI_ARG_NOW): taken where they are first used, the row's loads left behind five
    // dependent round trips to the kernarg segment — rows quantised 0.35 us later
    {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        asm volatile("" :: "s"(x32), "s"(ln_g), "s"(ln_b), "s"(eps), "s"(Wt), "s"(bq), "s"(qscale), "s"(S), "s"(kc), "s"(vc), "s"(T), "s"(ks), "s"(ns),
                     "s"(pmax), "s"(part_o), "s"(part_l), "s"(kv_row_stride), "s"(head_major), "s"(pf_ptr), "s"(pf_groups), "s"(pf_group_bytes),
                     "s"(sp.base), "s"(sp.slot), "s"(gx), "s"(gy));
    }
    const unsigned long long ts0 = stamp_t0(sp.base);
    unsigned long long tm1 = 0, tm2 = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: the tile and key addresses below are scalar base + lane offset)
    const int slice = head_major ? blockIdx.y : blockIdx.x, head = head_major ? blockIdx.x : blockIdx.y, i = blockIdx.z;
    const int H = head_major ? gridDim.x : gridDim.y;
    const size_t row = (size_t) i * H + head;
    kc += (int64_t) i * kv_row_stride; vc += (int64_t) i * kv_row_stride;
    const int K = S, np = K >> 6, nsl = (K + 255) >> 8;
    const bool tail_wave = wave < 4;                         // wave-uniform
    const XaKeys keys = xa_keys(slice, ks, T, S, head, wave & 3, lane);

    // ---- loads, in the order they are needed (vmcnt retires in order): the residual row (the previous launch wrote it), gain / bias of
    // this wavefront's slice, the weight tiles; K, V and the next launch's lines follow behind the quantiser
    const float * xr = x32 + (size_t) i * K;
    float4 v[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) { const int c = (u * 64 + lane) * 4; v[u] = *(const float4 *) (xr + (c < K ? c : 0)); }
    const int cs = (wave * 64 + lane) * 4, ccs = cs < K ? cs : 0;       // this wavefront's slice (wave < nsl), this lane's four columns
    const float4 gg = *(const float4 *) (ln_g + ccs), bb = *(const float4 *) (ln_b + ccs);
    __builtin_amdgcn_sched_barrier(0);
    const float bias = (bq && tid < 64) ? bq[head * 64 + tid] : 0.0f;      // (cold: first in the queue)
    __builtin_amdgcn_sched_barrier(0);
    uint32_t wq[2][CHV][QW], wh[2][CHV][HW];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int u = 0; u < CHV; ++u) {
            int tp = wave + NV * u; if (tp > np - 1) tp = np - 1;
            const uint8_t * t = Wt + ((size_t) (2 * head + g2) * np + tp) * tile_bytes<QT>();       // uniform
            const uint32_t l16 = (uint32_t) lane * 16u, l32 = (uint32_t) lane * 32u, l8 = (uint32_t) lane * 8u, l4 = (uint32_t) lane * 4u;
            uint32_t (&q)[QW] = wq[g2][u]; uint32_t (&h)[HW] = wh[g2][u];
            if constexpr (QW == 4) { const uint4 a4 = *(const uint4 *) (t + l16); q[0] = a4.x; q[1] = a4.y; q[2] = a4.z; q[3] = a4.w; }
            else { const uint4 a4 = *(const uint4 *) (t + l32), b4 = *(const uint4 *) (t + 16 + l32);
                   q[0] = a4.x; q[1] = a4.y; q[2] = a4.z; q[3] = a4.w; q[4] = b4.x; q[5] = b4.y; q[6] = b4.z; q[7] = b4.w; }
            if constexpr (HW == 2) { const uint2 h2 = *(const uint2 *) (t + 64 * QW * 4 + l8); h[0] = h2.x; h[1] = h2.y; }
            else h[0] = *(const uint32_t *) (t + 64 * QW * 4 + l4);
        }
    __builtin_amdgcn_sched_barrier(0);

    // ---- LayerNorm statistics of the row (every wavefront), this wavefront's slice as q8 blocks in LDS (k_qrows, SRC = 1, few rows)
    float sum = 0.0f;
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int c = (u * 64 + lane) * 4;
        if (c < K) sum += (v[u].x + v[u].y) + (v[u].z + v[u].w);
        else v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sum += WMI_SHX(sum, o);
    const float mean = sum / (float) K;
    float sqs = 0.0f;
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int c = (u * 64 + lane) * 4;
        if (c < K) {
            v[u].x -= mean; v[u].y -= mean; v[u].z -= mean; v[u].w -= mean;
            sqs += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
        }
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sqs += WMI_SHX(sqs, o);
    const float scale = 1.0f / sqrtf(sqs / (float) K + eps);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < nsl) {                                        // wave-uniform
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < MAXV; ++w) if (w == wave) y = v[w];
        y.x = __fadd_rn(__fmul_rn(y.x * scale, gg.x), bb.x); y.y = __fadd_rn(__fmul_rn(y.y * scale, gg.y), bb.y);
        y.z = __fadd_rn(__fmul_rn(y.z * scale, gg.z), bb.z); y.w = __fadd_rn(__fmul_rn(y.w * scale, gg.w), bb.w);
        if (cs >= K) y = make_float4(0.f, 0.f, 0.f, 0.f);
        float d, sv;
        const uint32_t q = quant4<F16D>(y.x, y.y, y.z, y.w, d, sv);
        if (cs < K) {
            *(uint32_t *) (sq + cs) = q;
            if ((lane & 7) == 0) { sd[cs >> 5] = d; ss[cs >> 5] = sv; }
        }
    }
    // K, V (wavefronts 0..3) and the next launch's lines go out behind the quantiser.  Measured on large-v3 q5_1 (in-kernel stamps, step
    // chain): requested at the top 1 439 - 1 459 us per step, behind the LayerNorm statistics 1 447 - 1 465, here 1 454 (two launches:
    // 1 535 - 1 545) — the rows are quantised at + 3.8 .. 4.7 us of the launch in every order (k_qrows: + 2.9), the query is ready at
    // + 5.6 .. 6.2, the launch ends at + 7.4 .. 8.0.  All of them UNCONDITIONAL plain loads (the wavefronts without a key slice, the threads
    // without a line to prefetch read byte 0): behind loads in a branch hipcc cannot count what is in flight and waits for everything (ISA
    // dump: vmcnt(0) in front of the quantiser), loads issued from inline assembly are invisible to its count (every later wait one load
    // too strict), volatile ones drain the queue.
    __builtin_amdgcn_sched_barrier(0);
    uint4 kk[XA_KPASS], vv[XA_KPASS];
#pragma unroll
    for (int p = 0; p < XA_KPASS; ++p) kk[p] = *(const uint4 *) ((const char *) kc + (tail_wave ? keys.off[p] : 0u));
#pragma unroll
    for (int p = 0; p < XA_KPASS; ++p) vv[p] = *(const uint4 *) ((const char *) vc + (tail_wave ? keys.off[p] : 0u));
    uint32_t junk0, junk1;
    {
        const uint32_t lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nwg = gridDim.x * gridDim.y * gridDim.z;
        const uint32_t pg = pf_ptr ? pf_groups : 1u, pb = pf_ptr ? pf_group_bytes : 0u;
        const uint32_t g = lin % pg, share = lin / pg, nshare = (nwg + pg - 1) / pg;
        const uint8_t * gp = (pf_ptr ? pf_ptr : Wt) + (size_t) g * pb;
        const uint32_t off0 = (share * NT + (uint32_t) tid) * 128u, off1 = off0 + nshare * NT * 128u;      // at most two lines per thread
        junk0 = __builtin_nontemporal_load((const uint32_t *) (gp + (off0 < pb ? off0 : 0u)));
        junk1 = __builtin_nontemporal_load((const uint32_t *) (gp + (off1 < pb ? off1 : 0u)));
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (sp.base) tm1 = wall_clock64();

    // ---- the block dots of this head's two row groups (k_qrows' tile phase, one activation row)
    const int fk = lane >> 5;
    const int8_t * afrag = sq + fk * 16;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        float out = 0.0f;
#pragma unroll
        for (int u = 0; u < CHV; ++u) {
            const int tp = wave + NV * u;
            if (tp < np) {                                   // wave-uniform
                uint32_t lo[4], hi[4]; float d, m;
                unpack<QT>(wq[g2][u], wh[g2][u], lo, hi, d, m);
                float d0 = d, d1 = d, m0 = m, m1 = m;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto r2 = __builtin_amdgcn_permlane32_swap(lo[e], hi[e], false, false);
                    lo[e] = r2[0]; hi[e] = r2[1];
                }
                { const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d1), false, false); d0 = __uint_as_float(r2[0]); d1 = __uint_as_float(r2[1]); }
                if (HAS_M) { const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m1), false, false); m0 = __uint_as_float(r2[0]); m1 = __uint_as_float(r2[1]); }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int b = 2 * tp + h;
                    const intx4 fa = *(const intx4 *) (afrag + b * 32);
                    intx4 fb; if (h == 0) { fb[0] = lo[0]; fb[1] = lo[1]; fb[2] = lo[2]; fb[3] = lo[3]; } else { fb[0] = hi[0]; fb[1] = hi[1]; fb[2] = hi[2]; fb[3] = hi[3]; }
                    intx16 z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0;
                    const intx16 ia = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, z, 0, 0, 0);
                    const float dwv = h == 0 ? d0 : d1, mwv = h == 0 ? m0 : m1;
                    out = fmaf((float) ia[0], sd[b] * dwv, out);
                    if (HAS_M) out = fmaf(mwv, ss[b], out);
                }
            }
        }
        if (lane < 32) redq[wave][g2][lane] = out;
    }
    __syncthreads();
    if (tid < 64) {
        const int g2 = tid >> 5, nl = tid & 31;
        float a = redq[0][g2][nl];
#pragma unroll
        for (int w = 1; w < NV; ++w) a += redq[w][g2][nl];
        qs[tid] = xa_round_f16((a + bias) * qscale);
    }
    __syncthreads();
    if (sp.base) tm2 = wall_clock64();
    if (tail_wave) xa_slice_tail(qs, kk, vv, keys.ok, red, lred, ored, row, ns, slice, pmax, part_o, part_l);
    else { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }      // (the tail's two barriers; these wavefronts touch nothing it shares)
    asm volatile("" :: "v"(junk0), "v"(junk1));              // (the prefetched dwords are "used" here: their wait sits at the end)
    stamp_end(sp.base, sp.slot, (((int) blockIdx.z * (int) gridDim.y + (int) blockIdx.y) * (int) gridDim.x + (int) blockIdx.x) * NV + wave, ts0, tm1, tm2);
}

template <int QT, int NV>
static void launch_xattn_fused_q(const XattnPlan & P, const float * x32, const float * ln_g, const float * ln_b, float eps, const uint8_t * Wt,
                                 const float * bq, float qscale, int n, int S, int H, const __half * kc, const __half * vc, int T, hipStream_t st,
                                 int64_t kv_row_stride, const uint8_t * pf_ptr, uint32_t pf_groups, uint32_t pf_group_bytes) {
    const dim3 grid = P.head_major ? dim3(H, P.ns, n) : dim3(P.ns, H, n);
    hipLaunchKernelGGL((k_xattn_fused_q<QT, NV>), grid, dim3(64 * NV), 0, st, x32, ln_g, ln_b, eps, Wt, bq, qscale, S, kc, vc, T, P.ks, P.ns,
                       P.pmax, P.part_o, P.part_l, kv_row_stride, P.head_major ? 1 : 0, pf_ptr, pf_groups, pf_group_bytes, stamp_next());
}
template <int QT>
static void xattn_fused_q_nv(const XattnPlan & P, const float * x32, const float * ln_g, const float * ln_b, float eps, const uint8_t * Wt,
                             const float * bq, float qscale, int n, int S, int H, const __half * kc, const __half * vc, int T, hipStream_t st,
                             int64_t kv_row_stride, const uint8_t * pf_ptr, uint32_t pf_groups, uint32_t pf_group_bytes) {
    if (S / 64 <= 8) launch_xattn_fused_q<QT, 4>(P, x32, ln_g, ln_b, eps, Wt, bq, qscale, n, S, H, kc, vc, T, st, kv_row_stride, pf_ptr, pf_groups, pf_group_bytes);
    else             launch_xattn_fused_q<QT, 8>(P, x32, ln_g, ln_b, eps, Wt, bq, qscale, n, S, H, kc, vc, T, st, kv_row_stride, pf_ptr, pf_groups, pf_group_bytes);
}

template <int QT, int NR4, int SRC, int NW, int CHX = 0>
void launch_qrows(const GemvArgs & a, const float * a32, const uint8_t * Wt, hipStream_t st) {
    const int nb = a.K / 32, R8 = NR4 * 8;
    const size_t smem = (((size_t) a.n * (a.K + 16) + 15) & ~(size_t) 15) + (size_t) 2 * nb * R8 * 4 + (size_t) NW * 32 * R8 * 4;
    const int ngroups = (a.N + 31) / 32;
    // The vocabulary projection (1 621 row groups): as many workgroups as are resident at once (124 VGPRs x 8 wavefronts: two per CU; 164 x 4:
    // three), each walking its groups with the next group's tiles in flight — with 1 024 workgroups the second round paid the prologue
    // (LayerNorm + quantiser, ~4 us) again behind the first.
    static const int cap_env = getenv("WMI_QROWS_BLOCKS") ? atoi(getenv("WMI_QROWS_BLOCKS")) : 0;      // A/B knob
    const int cap = cap_env > 0 ? cap_env : NW == 4 ? 768 : NW == 8 ? 512 : 256;
    int blocks = ngroups; if (blocks > cap) blocks = cap;
    if (a.ksplit > 1) blocks = ngroups * a.ksplit;            // (qrows() grants the split only where every (row group, part) gets its own workgroup)
    if (a.pf_ptr) blocks += PF_WG;                          // the prefetch workgroups (see the kernel)
    static std::atomic<uint64_t> lds_ok{0};
    if (smem > 48 * 1024) allow_full_lds((const void *) k_qrows<QT, NR4, SRC, NW, CHX>, lds_ok);
    hipLaunchKernelGGL((k_qrows<QT, NR4, SRC, NW, CHX>), dim3(blocks), dim3(NW * 64), smem, st, a, a32, Wt);
}

template <int QT, int NR4, int SRC>
void qrows_nw(const GemvArgs & a, const float * a32, const uint8_t * Wt, hipStream_t st) {
    // K split: every wavefront should find all of its tiles in one round of loads (<= 5, resp. 3 with 8 wavefronts)
    const int np = a.K / 64 / (a.ksplit > 1 ? a.ksplit : 1);
    if constexpr (SRC == 1) { if (np <= 8) launch_qrows<QT, NR4, SRC, 4>(a, a32, Wt, st); else launch_qrows<QT, NR4, SRC, 8>(a, a32, Wt, st); }
    else {
        if (np <= 8)       launch_qrows<QT, NR4, SRC, 4>(a, a32, Wt, st);
        else if (np <= 24) launch_qrows<QT, NR4, SRC, 8>(a, a32, Wt, st);
        else {
            // a K part of <= 48 tiles on 16 wavefronts: three tiles per wavefront in flight — 18 tile registers instead of 30, no spills
            // under the 128 VGPRs of a 16-wavefront workgroup (the five-deep form spills 21 registers at <= 8 rows)
            if constexpr (SRC == 2) { if (a.ksplit > 1 && np <= 48) { launch_qrows<QT, NR4, SRC, 16, 3>(a, a32, Wt, st); return; } }
            launch_qrows<QT, NR4, SRC, 16>(a, a32, Wt, st);
        }
    }
}

template <int QT, int NR4>
void qrows_src(const GemvArgs & a, const float * a32, const uint8_t * Wt, hipStream_t st) {
    if (a.ln_g)        qrows_nw<QT, NR4, 1>(a, a32, Wt, st);
    else if (a.comb_o) qrows_nw<QT, NR4, 3>(a, a32, Wt, st);
    else if (a32)      qrows_nw<QT, NR4, 0>(a, a32, Wt, st);
    else               qrows_nw<QT, NR4, 2>(a, a32, Wt, st);
}

template <int QT>
void qrows_t(const GemvArgs & a, const float * a32, const uint8_t * Wt, hipStream_t st) {
    if (a.n <= 8)       qrows_src<QT, 1>(a, a32, Wt, st);
    else if (a.n <= 16) qrows_src<QT, 2>(a, a32, Wt, st);
    else                qrows_src<QT, 4>(a, a32, Wt, st);
}

// ------------------------------------------------------------------------------------------------ embedding
template <int QT>
__global__ void k_qembed(const int32_t * __restrict__ tokens, const int32_t * __restrict__ pos, const DecStep * __restrict__ host_step,
                         DecStep * __restrict__ dev_step, int S, const uint8_t * __restrict__ Wt, const float * __restrict__ pe, float * __restrict__ x) {
    constexpr int QW = Geo<QT>::QW, HW = Geo<QT>::HW;
    __shared__ DecStep st;
    const int i = blockIdx.x;
    int tok, ps;
    if (host_step) {                                          // graph-replay form: the step record comes from pinned host memory
        if (threadIdx.x < sizeof(DecStep) / 4) ((int32_t *) &st)[threadIdx.x] = ((const volatile int32_t *) (host_step + i))[threadIdx.x];
        __syncthreads();
        if (threadIdx.x < sizeof(DecStep) / 4) ((int32_t *) (dev_step + i))[threadIdx.x] = ((const int32_t *) &st)[threadIdx.x];
        tok = st.token; ps = st.pos;
    } else { tok = tokens[i]; ps = pos[i]; }
    const int np = S >> 6, tn = tok >> 5, nl = tok & 31;
    // thread -> (block b, dword i4 of its 8 dwords of 4 elements)
    for (int e = threadIdx.x; e < (S >> 5) * 8; e += blockDim.x) {
        const int b = e >> 3, part = e & 7;                   // part: 0..3 -> elements 4 part .. (lo), 4..7 -> 16 + 4 (part - 4) .. (hi)
        const uint8_t * t = Wt + ((size_t) tn * np + (b >> 1)) * tile_bytes<QT>();
        const int ln = nl + 32 * (b & 1);
        uint32_t qs[QW], hd[HW];
#pragma unroll
        for (int w = 0; w < QW; ++w) qs[w] = *(const uint32_t *) (t + ln * QW * 4 + w * 4);
#pragma unroll
        for (int w = 0; w < HW; ++w) hd[w] = *(const uint32_t *) (t + 64 * QW * 4 + ln * HW * 4 + w * 4);
        uint32_t lo[4], hi[4]; float d, m;
        unpack<QT>(qs, hd, lo, hi, d, m);
        uint32_t pk = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (part == w) pk = lo[w]; if (part == 4 + w) pk = hi[w]; }
        const int c = b * 32 + (part & 3) * 4 + (part >> 2) * 16;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int wv = (int) (int8_t) (pk >> (8 * u));
            const float y = Geo<QT>::M ? __fadd_rn(__fmul_rn((float) wv, d), m) : __fmul_rn((float) wv, d);
            x[(size_t) i * S + c + u] = y + pe[(size_t) ps * S + c + u];
        }
    }
}

// ------------------------------------------------------------------------------------------------ f16 form of the GEMM
// From ~256 activation rows on, the block dots above are VALU-bound (48 scale / convert instructions per 32-cycle MFMA).
// The large-M projections therefore run as f16 x f16 -> f32 products on k_gemm's tiles: the activation rows are the SAME q8 quants and
// scales the reference computes, handed over as f16(d_a * q_a) (k_q8_rows, `deq`), the weight blocks as f16(d_w * q_w + m_w) in a scratch
// image written by k_qdequant right before the GEMM (the weights stay in their blocks in HBM; the image holds one matrix at a time).
// sum_k (d_w q_w + m_w)(d_a q_a) is the reference's sum_blocks (d_w d_a) isum + m_w s_a with the two factors of every term rounded to f16
// (2^-11 relative each) before the f32 accumulation — the operand rounding the f16 models have, an order of magnitude below what the
// reference's own outputs move by when a quant flips (tests/test_gpu_parity.py, the yardstick of the quantised models).
//
// k_qdequant: one wavefront per 32 x 64 tile (lane = (row, block of the pair): 32 values = 64 bytes), four K-consecutive tiles per
// workgroup, turned through LDS so that a row's 512 bytes leave as one run.
template <int QT>
__device__ __forceinline__ void qdequant_body(int bx, int by, const uint8_t * __restrict__ Wt, int np, int K, __half * __restrict__ out) {
    constexpr int QW = Geo<QT>::QW, HW = Geo<QT>::HW, ROWB = 528;
    __shared__ __attribute__((aligned(16))) unsigned char sm[32 * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tp = bx * 4 + wave, tn = by;
    if (tp < np) {
        const uint8_t * t = Wt + ((size_t) tn * np + tp) * tile_bytes<QT>();
        uint32_t rq[QW], rh[HW];
        { const uint4 u = *(const uint4 *) (t + (size_t) lane * QW * 4); rq[0] = u.x; rq[1] = u.y; rq[2] = u.z; rq[3] = u.w; }
        if constexpr (QW == 8) { const uint4 u = *(const uint4 *) (t + (size_t) lane * 32 + 16); rq[4] = u.x; rq[5] = u.y; rq[6] = u.z; rq[7] = u.w; }
#pragma unroll
        for (int h = 0; h < HW; ++h) rh[h] = *(const uint32_t *) (t + 64 * QW * 4 + lane * HW * 4 + h * 4);
        uint32_t lo[4], hi[4]; float d, m;
        unpack<QT>(rq, rh, lo, hi, d, m);
        unsigned char * dst = sm + (lane & 31) * ROWB + wave * 128 + (lane >> 5) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = i < 2 ? lo[2 * i] : hi[2 * (i - 2)], w2 = i < 2 ? lo[2 * i + 1] : hi[2 * (i - 2) + 1];
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t src = e < 2 ? w : w2; const int sh = (e & 1) * 16;
                const float x0 = (float) (int) (int8_t) ((src >> sh) & 0xFF) * d + m, x1 = (float) (int) (int8_t) ((src >> (sh + 8)) & 0xFF) * d + m;
                const __half2 h = __floats2half2_rn(pin_f32(x0), pin_f32(x1));
                pk[e] = *(const uint32_t *) &h;
            }
            *(uint4 *) (dst + i * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
    __syncthreads();
    const int k0 = bx * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + tid, row = idx >> 5, ch = idx & 31;
        if (k0 + ch * 8 < K) *(uint4 *) (out + ((size_t) tn * 32 + row) * K + k0 + ch * 8) = *(const uint4 *) (sm + row * ROWB + ch * 16);
    }
}
template <int QT>
__global__ __launch_bounds__(256) void k_qdequant(const uint8_t * __restrict__ Wt, int np, int K, __half * __restrict__ out) {
    qdequant_body<QT>((int) blockIdx.x, (int) blockIdx.y, Wt, np, K, out);
}
// The row quantiser of a projection and the dequantisation of that projection's weights in ONE launch (they are independent, both are
// short streaming jobs, and as two launches each paid its own boundary and ramp): workgroups [0, nrb) quantise four rows each, the
// others are k_qdequant's (bx, by) = ((blockIdx.x - nrb) % nbx, (blockIdx.x - nrb) / nbx).
template <int MAXV, int SRC, int QT>
__global__ __launch_bounds__(256) void k_q8_rows_wdeq(const float * __restrict__ x32, const __half * __restrict__ x16, int M, int K,
                                                      const float * __restrict__ g, const float * __restrict__ b, float eps,
                                                      int8_t * __restrict__ qs, float * __restrict__ dT, float * __restrict__ sT, int ldm,
                                                      float * __restrict__ out32, __half * __restrict__ out16, __half * __restrict__ deq,
                                                      int nrb, const uint8_t * __restrict__ Wt, int np, int Kw, int nbx, __half * __restrict__ wout) {
    if ((int) blockIdx.x < nrb) { q8_rows_body<MAXV, SRC, !Geo<QT>::M>((int) blockIdx.x, x32, x16, M, K, g, b, eps, qs, dT, sT, ldm, out32, out16, deq); return; }
    const int e = (int) blockIdx.x - nrb;
    qdequant_body<QT>(e % nbx, e / nbx, Wt, np, Kw, wout);
}

template <int QT> static void qdequant_launch(const uint8_t * Wt, int64_t row0, int64_t rows, int K, __half * out, hipStream_t st) {
    const int np = K / 64;
    hipLaunchKernelGGL((k_qdequant<QT>), dim3((np + 3) / 4, (unsigned) ((rows + 31) / 32)), dim3(256), 0, st,
                       Wt + (size_t) (row0 / 32) * np * tile_bytes<QT>(), np, K, out);
}

} // namespace

// ---- resident f16 images of the matrices that the f16 form multiplies with (round 6, OPT-IN: WMI_QENC_F16_CACHE=1).  The scratch image above
// is rewritten for every projection of every encode (large-v3 q5_1: 1.26 GB of f16(d q + m) written and re-read per 30 s chunk); with the
// images kept beside the blocks (large-v3: + 1.7 GB of HBM for the encoder's matrices and the cross K | V projections; the decoder's
// matrices — the bytes a token streams — stay quantised only) that pass disappears.  Measured on large-v3 q5_1: encoder 6.70 -> 6.51 ms
// (the pass was already fused into the row quantiser's launch and mostly hidden behind it) — 0.5 % of a chunk for 1.7 GB, so the default
// stays "quantised blocks only".  The image is the one k_qdequant writes: the products are the same bits with and without it.
namespace {
struct F16Image { __half * p = nullptr; size_t bytes = 0; };
std::mutex g_f16img_mu;
std::map<const void *, F16Image> g_f16img;               // key: first tile of the cached row range (device address: unique per arena)
std::atomic<size_t> g_f16img_bytes{0};
}
size_t qweights_f16_cached_bytes() { return g_f16img_bytes.load(std::memory_order_relaxed); }
void qweights_f16_release(const void * lo, const void * hi) {
    std::lock_guard<std::mutex> lk(g_f16img_mu);
    for (auto it = g_f16img.begin(); it != g_f16img.end();) {
        if (it->first >= lo && it->first < hi) { (void) hipFree(it->second.p); g_f16img_bytes.fetch_sub(it->second.bytes, std::memory_order_relaxed); it = g_f16img.erase(it); }
        else ++it;
    }
}
// rows [row0, row0 + rows) of W as a resident f16 image [rows][K]; nullptr: not cached (switched off, no memory to spare, allocation failed)
static const __half * qweights_f16_get(QMat W, int64_t row0, int64_t rows, int K, hipStream_t st) {
    static const int mode = getenv("WMI_QENC_F16_CACHE") ? atoi(getenv("WMI_QENC_F16_CACHE")) : 0;
    if (mode == 0 || !W.tiles || (row0 % 32) != 0) return nullptr;
    const size_t tile_b = (size_t) q_tile_bytes(W.qtype);
    const void * key = W.tiles + (size_t) (row0 / 32) * (size_t) (K / 64) * tile_b;
    std::lock_guard<std::mutex> lk(g_f16img_mu);
    auto it = g_f16img.find(key);
    if (it != g_f16img.end()) return it->second.bytes == (size_t) rows * K * sizeof(__half) ? it->second.p : nullptr;
    const size_t bytes = (size_t) rows * K * sizeof(__half);
    { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < bytes + ((size_t) 16 << 30)) return nullptr; }      // (never the last 16 GB)
    F16Image img;
    if (hipMalloc((void **) &img.p, bytes) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    img.bytes = bytes;
    qdequant(W, row0, rows, K, img.p, st);
    (void) hipStreamSynchronize(st);                      // (once per matrix: other states' streams may use the image from here on)
    g_f16img[key] = img; g_f16img_bytes.fetch_add(bytes, std::memory_order_relaxed);
    return img.p;
}
bool qweights_f16_resident(const QMat & W) {
    if (!W.tiles) return false;
    std::lock_guard<std::mutex> lk(g_f16img_mu);
    return g_f16img.find((const void *) W.tiles) != g_f16img.end();
}

static int qgemm_f16_rows() {
    // f16 form (see k_qdequant) from WMI_QGEMM_F16_ROWS activation rows on (0 = never: the block-dot kernel for every M)
    static const int v = getenv("WMI_QGEMM_F16_ROWS") ? atoi(getenv("WMI_QGEMM_F16_ROWS")) : 256;
    return v;
}

template <int QT>
static void q8_rows_wdeq_launch(const float * x32, const __half * x16, int M, int K, const float * ln_g, const float * ln_b, float eps,
                                Q8Rows out, float * out32, __half * out16, const uint8_t * Wt, int Nw, int Kw, hipStream_t st) {
    const int nrb = (M + 3) / 4, np = Kw / 64, nbx = (np + 3) / 4, nby = (Nw + 31) / 32;
    const dim3 grid(nrb + nbx * nby), block(256);
#define WMI_Q8W(MAXV, SRC) hipLaunchKernelGGL((k_q8_rows_wdeq<MAXV, SRC, QT>), grid, block, 0, st, x32, x16, M, K, ln_g, ln_b, eps, out.qs, out.d, out.s, out.ldm, \
                                              out32, out16, out.deq, nrb, Wt, np, Kw, nbx, out.wdeq)
    if (ln_g) {
        const int nv = (K + 255) / 256;
        if (nv <= 2) WMI_Q8W(2, 1); else if (nv <= 4) WMI_Q8W(4, 1); else WMI_Q8W(6, 1);
    } else if (x32) WMI_Q8W(4, 0);
    else            WMI_Q8W(4, 2);
#undef WMI_Q8W
}

bool quantize_rows(const float * x32, const __half * x16, int M, int K, const float * ln_g, const float * ln_b, float eps,
                   int qtype, Q8Rows out, float * out32, __half * out16, hipStream_t st, const QMat * W_next, int N_next) {
    if (M <= 0) return false;
    static const bool fuse = getenv("WMI_QGEMM_NO_FUSED_DEQ") == nullptr;    // A/B knob
    // (a projection whose matrix may be a resident f16 image gets it from qgemm — made on first use — not from this launch)
    static const bool resident_off = !(getenv("WMI_QENC_F16_CACHE") && atoi(getenv("WMI_QENC_F16_CACHE")) != 0);
    if (fuse && W_next && (resident_off || !out.w_resident_ok) && W_next->tiles && W_next->qtype == qtype && out.deq && out.wdeq && qgemm_f16_rows() > 0 && M >= qgemm_f16_rows() &&
        (K % 64) == 0 && (N_next % 32) == 0 && (size_t) N_next * K <= out.wdeq_elems) {
        // the projection that follows takes the f16 form: its weight image is written by this launch (qgemm is told through Q8Rows::wdeq_ready)
        switch (qtype) {
            case QT_Q4_0: q8_rows_wdeq_launch<QT_Q4_0>(x32, x16, M, K, ln_g, ln_b, eps, out, out32, out16, W_next->tiles, N_next, K, st); break;
            case QT_Q4_1: q8_rows_wdeq_launch<QT_Q4_1>(x32, x16, M, K, ln_g, ln_b, eps, out, out32, out16, W_next->tiles, N_next, K, st); break;
            case QT_Q5_0: q8_rows_wdeq_launch<QT_Q5_0>(x32, x16, M, K, ln_g, ln_b, eps, out, out32, out16, W_next->tiles, N_next, K, st); break;
            case QT_Q5_1: q8_rows_wdeq_launch<QT_Q5_1>(x32, x16, M, K, ln_g, ln_b, eps, out, out32, out16, W_next->tiles, N_next, K, st); break;
            case QT_Q8_0: q8_rows_wdeq_launch<QT_Q8_0>(x32, x16, M, K, ln_g, ln_b, eps, out, out32, out16, W_next->tiles, N_next, K, st); break;
            default: return false;
        }
        return true;
    }
    const dim3 grid((M + 3) / 4), block(256);
    const bool f16d = !q_geom(qtype).has_m;
#define WMI_Q8(MAXV, SRC) do { if (f16d) hipLaunchKernelGGL((k_q8_rows<MAXV, SRC, true>), grid, block, 0, st, x32, x16, M, K, ln_g, ln_b, eps, out.qs, out.d, out.s, out.ldm, out32, out16, out.deq); \
                               else      hipLaunchKernelGGL((k_q8_rows<MAXV, SRC, false>), grid, block, 0, st, x32, x16, M, K, ln_g, ln_b, eps, out.qs, out.d, out.s, out.ldm, out32, out16, out.deq); } while (0)
    if (ln_g) {
        const int nv = (K + 255) / 256;
        if (nv <= 2) WMI_Q8(2, 1); else if (nv <= 4) WMI_Q8(4, 1); else WMI_Q8(6, 1);
    } else if (x32) WMI_Q8(4, 0);
    else            WMI_Q8(4, 2);
#undef WMI_Q8
    return false;
}

void qdequant(QMat W, int64_t row0, int64_t rows, int K, __half * out, hipStream_t st) {
    switch (W.qtype) {
        case QT_Q4_0: qdequant_launch<QT_Q4_0>(W.tiles, row0, rows, K, out, st); break;
        case QT_Q4_1: qdequant_launch<QT_Q4_1>(W.tiles, row0, rows, K, out, st); break;
        case QT_Q5_0: qdequant_launch<QT_Q5_0>(W.tiles, row0, rows, K, out, st); break;
        case QT_Q5_1: qdequant_launch<QT_Q5_1>(W.tiles, row0, rows, K, out, st); break;
        case QT_Q8_0: qdequant_launch<QT_Q8_0>(W.tiles, row0, rows, K, out, st); break;
        default: break;
    }
}

void qgemm(int epi, const GemmArgs & a, Q8Rows A, QMat W, hipStream_t st) {
    const int f16_rows = qgemm_f16_rows();
    if (f16_rows > 0 && a.M >= f16_rows && A.deq && A.wdeq && (a.K % 64) == 0 && (a.N % 32) == 0) {
        const int64_t cap = (int64_t) (A.wdeq_elems / (size_t) a.K) / 32 * 32;               // weight rows the image holds
        GemmArgs g = a; g.A = A.deq; g.lda = a.K; g.W = A.wdeq; g.ldw = a.K;
        if (!(A.wdeq_ready && A.wdeq_of == (const void *) W.tiles)) {
            // the whole matrix as a resident image (made on first use; see qweights_f16_get): no dequantisation pass, no layer groups
            if (A.w_resident_ok) if (const __half * img = qweights_f16_get(W, 0, a.N, a.K, st)) { g.W = img; gemm(epi, g, st); return; }
        }
        if (epi == EPI_CROSS_KV && cap >= 2 * a.S) {
            // the decoder layers' K | V projections, as many layers at a time as the image holds
            const int layers = a.N / (2 * a.S), per = (int) std::min<int64_t>(layers, cap / (2 * a.S));
            for (int l0 = 0; l0 < layers; l0 += per) {
                const int nl = std::min(per, layers - l0);
                qdequant(W, (int64_t) l0 * 2 * a.S, (int64_t) nl * 2 * a.S, a.K, A.wdeq, st);
                g.N = nl * 2 * a.S; g.bias = a.bias ? a.bias + (size_t) l0 * 2 * a.S : nullptr;
                g.C = (__half *) a.C + (size_t) l0 * a.layer_stride; g.aux = (__half *) a.aux + (size_t) l0 * a.layer_stride;
                gemm(epi, g, st);
            }
            return;
        }
        if (epi != EPI_CROSS_KV && a.N <= cap) {
            if (!(A.wdeq_ready && A.wdeq_of == (const void *) W.tiles)) qdequant(W, 0, a.N, a.K, A.wdeq, st);      // (a stale image of another matrix is never trusted)
            gemm(epi, g, st);
            return;
        }
    }
    switch (W.qtype) {
        case QT_Q4_0: qgemm_epi<QT_Q4_0>(epi, a, A, W.tiles, st); break;
        case QT_Q4_1: qgemm_epi<QT_Q4_1>(epi, a, A, W.tiles, st); break;
        case QT_Q5_0: qgemm_epi<QT_Q5_0>(epi, a, A, W.tiles, st); break;
        case QT_Q5_1: qgemm_epi<QT_Q5_1>(epi, a, A, W.tiles, st); break;
        case QT_Q8_0: qgemm_epi<QT_Q8_0>(epi, a, A, W.tiles, st); break;
        default: break;
    }
}

bool qrows_ksplit_ok(const GemvArgs & a, int parts) {
    // a16 rows (mlp.2), whole 256-column slices and tile pairs per part, one workgroup per (row group, part)
    return parts == 2 && a.a16 && !a.ln_g && !a.comb_o && !a.rows && a.epi == EPI_F32_BIAS_RESID && a.n <= 32 &&
           (a.K % (256 * parts)) == 0 && ((a.N + 31) / 32) * parts <= 256;
}

void qrows(const GemvArgs & a_in, const float * a32, QMat W, hipStream_t st) {
    GemvArgs a = a_in;
    if (a.ksplit > 1 && (!a.kpart || a32 || !qrows_ksplit_ok(a, a.ksplit))) { a.ksplit = 0; a.kpart = nullptr; }
    { const Stamp sp = stamp_next(); a.stamps = sp.base; a.stamp_slot = sp.slot; }
    switch (W.qtype) {
        case QT_Q4_0: qrows_t<QT_Q4_0>(a, a32, W.tiles, st); break;
        case QT_Q4_1: qrows_t<QT_Q4_1>(a, a32, W.tiles, st); break;
        case QT_Q5_0: qrows_t<QT_Q5_0>(a, a32, W.tiles, st); break;
        case QT_Q5_1: qrows_t<QT_Q5_1>(a, a32, W.tiles, st); break;
        case QT_Q8_0: qrows_t<QT_Q8_0>(a, a32, W.tiles, st); break;
        default: break;
    }
}

bool qattn_cross_qsplit_partials(const float * x32, const float * ln_g, const float * ln_b, float eps, QMat Wcq, const float * bq, float qscale,
                                 int n, int S, int H, const __half * kc, const __half * vc, int T, float * scratch,
                                 const float ** po, const float ** pl, const float ** pm, int * pns, hipStream_t st, int64_t kv_row_stride,
                                 QMat pfW, int pfN, int pfK) {
    static const bool off = getenv("WMI_Q_XATTN_TWO_LAUNCHES") != nullptr;          // A/B knob: cross query as its own k_qrows launch
    // One row only (the greedy step): every (row, head, slice) workgroup unpacks the head's tiles for ITS row, where the k_qrows launch
    // unpacks a tile once for up to eight rows — large-v3 q5_1, beam 5: 44.3 -> 49.3 ms per chunk, 8 lock-step chunks 78.5 -> 87.4 ms per
    // call with every row count through this launch (greedy chunk 35.3 -> 33.9).  WMI_Q_XATTN_ROWS raises the limit (A/B).
    static const int max_rows = getenv("WMI_Q_XATTN_ROWS") ? atoi(getenv("WMI_Q_XATTN_ROWS")) : 1;
    if (n > max_rows) return false;
    const XattnPlan P = attn_cross_plan(n, H, T, scratch);
    if (off || !P.fused || !Wcq.tiles || !ln_g || S != H * 64 || S > 1536 || (S % 64) != 0 || S / 64 > 24) return false;
    const uint8_t * pf_ptr = nullptr; uint32_t pf_groups = 0, pf_group_bytes = 0;
    if (pfW.tiles && pfN > 0 && pfN <= 8192) {
        pf_ptr = pfW.tiles; pf_groups = (uint32_t) ((pfN + 31) / 32); pf_group_bytes = (uint32_t) ((size_t) (pfK / 64) * q_tile_bytes(pfW.qtype));
    }
    switch (Wcq.qtype) {
        case QT_Q4_0: xattn_fused_q_nv<QT_Q4_0>(P, x32, ln_g, ln_b, eps, Wcq.tiles, bq, qscale, n, S, H, kc, vc, T, st, kv_row_stride, pf_ptr, pf_groups, pf_group_bytes); break;
        case QT_Q4_1: xattn_fused_q_nv<QT_Q4_1>(P, x32, ln_g, ln_b, eps, Wcq.tiles, bq, qscale, n, S, H, kc, vc, T, st, kv_row_stride, pf_ptr, pf_groups, pf_group_bytes); break;
        case QT_Q5_0: xattn_fused_q_nv<QT_Q5_0>(P, x32, ln_g, ln_b, eps, Wcq.tiles, bq, qscale, n, S, H, kc, vc, T, st, kv_row_stride, pf_ptr, pf_groups, pf_group_bytes); break;
        case QT_Q5_1: xattn_fused_q_nv<QT_Q5_1>(P, x32, ln_g, ln_b, eps, Wcq.tiles, bq, qscale, n, S, H, kc, vc, T, st, kv_row_stride, pf_ptr, pf_groups, pf_group_bytes); break;
        case QT_Q8_0: xattn_fused_q_nv<QT_Q8_0>(P, x32, ln_g, ln_b, eps, Wcq.tiles, bq, qscale, n, S, H, kc, vc, T, st, kv_row_stride, pf_ptr, pf_groups, pf_group_bytes); break;
        default: return false;
    }
    *po = P.part_o; *pl = P.part_l; *pm = P.pmax; *pns = P.ns;
    return true;
}

template <int QT> static void qembed_launch(const int32_t * tokens, const int32_t * pos, const DecStep * hs, DecStep * ds, int n, int S,
                                            const uint8_t * Wt, const float * pe, float * x, hipStream_t st) {
    hipLaunchKernelGGL((k_qembed<QT>), dim3(n), dim3(256), 0, st, tokens, pos, hs, ds, S, Wt, pe, x);
}
static void qembed_any(const int32_t * tokens, const int32_t * pos, const DecStep * hs, DecStep * ds, int n, int S, QMat te,
                       const float * pe, float * x, hipStream_t st) {
    switch (te.qtype) {
        case QT_Q4_0: qembed_launch<QT_Q4_0>(tokens, pos, hs, ds, n, S, te.tiles, pe, x, st); break;
        case QT_Q4_1: qembed_launch<QT_Q4_1>(tokens, pos, hs, ds, n, S, te.tiles, pe, x, st); break;
        case QT_Q5_0: qembed_launch<QT_Q5_0>(tokens, pos, hs, ds, n, S, te.tiles, pe, x, st); break;
        case QT_Q5_1: qembed_launch<QT_Q5_1>(tokens, pos, hs, ds, n, S, te.tiles, pe, x, st); break;
        case QT_Q8_0: qembed_launch<QT_Q8_0>(tokens, pos, hs, ds, n, S, te.tiles, pe, x, st); break;
        default: break;
    }
}
void qdec_embed(const int32_t * tokens, const int32_t * pos, int n, int S, QMat te, const float * pe, float * x, hipStream_t st) {
    qembed_any(tokens, pos, nullptr, nullptr, n, S, te, pe, x, st);
}
void qdec_embed_step(const DecStep * host_step, DecStep * dev_step, int S, QMat te, const float * pe, float * x, hipStream_t st, int n_rows) {
    qembed_any(nullptr, nullptr, host_step, dev_step, n_rows, S, te, pe, x, st);
}

}} // namespace wmi::k
