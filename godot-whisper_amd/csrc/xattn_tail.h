// The key-slice part of the decoder's cross-attention for one (row, head, slice) workgroup of four wavefronts: scores of <= 192 keys
// against the query in LDS, the slice maximum, e = f16(exp(f16(s - m))), l = sum e, o = e . V, written as partials (m, l, o[64]) that
// the consumer rescales (GemvArgs::comb_m).  This is k_xattn_fused's tail (k_attn.hip) — the same lanes, the same order of every sum,
// the same roundings — as a function, for the form whose query comes from a block-quantised projection (k_quant.hip: k_xattn_fused_q);
// tests/test_gpu_variants.py holds the two forms to the same bits.
// Lane = (key g = lane / 8, 16-byte octet o = lane % 8) for K and V alike; a wavefront owns KPASS passes x 8 keys.
#pragma once
#include "kernels.h"
#include "wave_ops.h"

namespace wmi { namespace k {

constexpr int XA_KPASS = 6;                                 // 4 wavefronts x 6 passes x 8 keys = 192 keys per slice

// which keys a lane reads: `ok` = the key exists and is this wavefront's; `off` = byte offset of its 16 bytes from the head of the K (or V)
// cache of this row — 32-bit, so that the loads take the scalar-base form (the cache of a row is far below 4 GB).  Keys that do not
// exist read the last row instead (finite values, multiplied by a weight of exactly 0 / replaced by -inf).
struct XaKeys { bool ok[XA_KPASS]; uint32_t off[XA_KPASS]; };

// slice `slice` of `ks` keys, wavefront `wave` (uniform), T keys in all, rows of S halves, this head's 64 columns
__device__ __forceinline__ XaKeys xa_keys(int slice, int ks, int T, int S, int head, int wave, int lane) {
    XaKeys r;
    const int g = lane >> 3, o = lane & 7;
    const int kpw = (((ks + 3) >> 2) + 7) & ~7;             // keys per wavefront, whole passes
    const int t0 = wave * kpw;
    const uint32_t col = (uint32_t) (head * 64 + o * 8) * 2u, rowb = (uint32_t) S * 2u;
#pragma unroll
    for (int p = 0; p < XA_KPASS; ++p) {
        const int t = t0 + 8 * p + g, j = slice * ks + t;
        r.ok[p] = 8 * p < kpw && t < ks && j < T;
        r.off[p] = __umul24((uint32_t) (j < T ? j : T - 1), rowb) + col;
    }
    return r;
}

__device__ __forceinline__ float xa_round_f16(float x) { return __half2float(f2h(x)); }
__device__ __forceinline__ float xa_exp16(float d) { return xa_round_f16(expf(xa_round_f16(d))); }

// qs[64]: the query (f16 values as f32) — written and fenced (__syncthreads) by the caller.  red[4], lred[4], ored[4][64]: LDS scratch.
__device__ __forceinline__ void xa_slice_tail(const float * qs, const uint4 (&kk)[XA_KPASS], const uint4 (&vv)[XA_KPASS], const bool (&ok)[XA_KPASS],
                                              float * red, float * lred, float (*ored)[64], size_t row, int ns, int slice,
                                              float * __restrict__ pmax, float * __restrict__ part_o, float * __restrict__ part_l) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 3, o = lane & 7;
    float qo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qo[e] = qs[o * 8 + e];
    float sv[XA_KPASS];
    float lmax = -INFINITY;
#pragma unroll
    for (int p = 0; p < XA_KPASS; ++p) {
        const __half2 * h = (const __half2 *) &kk[p];
        float dot = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float2 f = __half22float2(h[e]);
            dot = fmaf(f.x, qo[2 * e], dot);
            dot = fmaf(f.y, qo[2 * e + 1], dot);
        }
        dot += WMI_SHX(dot, 1); dot += WMI_SHX(dot, 2); dot += WMI_SHX(dot, 4);     // the 8 octets of a key
        sv[p] = ok[p] ? dot : -INFINITY;
        lmax = fmaxf(lmax, sv[p]);
    }
    _Pragma("unroll") for (int x = 32; x > 4; x >>= 1) lmax = fmaxf(lmax, WMI_SHX(lmax, x));       // the octet lanes of a key agree already
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float acc[8], lsum = 0.0f;
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.0f;
#pragma unroll
    for (int p = 0; p < XA_KPASS; ++p) {
        const float e = ok[p] ? xa_exp16(sv[p] - m) : 0.0f;
        if (o == 0) lsum += e;
        const __half2 * h = (const __half2 *) &vv[p];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 f = __half22float2(h[q]);
            acc[2 * q]     = fmaf(e, f.x, acc[2 * q]);
            acc[2 * q + 1] = fmaf(e, f.y, acc[2 * q + 1]);
        }
    }
    _Pragma("unroll") for (int x = 32; x > 0; x >>= 1) lsum += WMI_SHX(lsum, x);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float v = acc[d];
        v += WMI_SHX(v, 8); v += WMI_SHX(v, 16); v += WMI_SHX(v, 32);
        acc[d] = v;
    }
    if (g == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) ored[wave][o * 8 + d] = acc[d];
    }
    if (lane == 0) lred[wave] = lsum;
    __syncthreads();
    if (tid < 64) part_o[(row * ns + slice) * 64 + tid] = (ored[0][tid] + ored[1][tid]) + (ored[2][tid] + ored[3][tid]);
    if (tid == 0) { part_l[row * ns + slice] = (lred[0] + lred[1]) + (lred[2] + lred[3]); pmax[row * ns + slice] = m; }
}

} } // namespace wmi::k
