// LayerNorm wave kernel (reference W/ggml.c:9301-9352 + the mul/add that follow it
// in every block, W/whisper.cpp:1821-1826).  One 64-lane wavefront owns one row: the row lives in
// registers, mean and variance are two DPP/shuffle reductions, and gamma/beta are applied in the same
// pass — the reference's norm, mul and add nodes fused.  The f16 output is what the following GEMM
// consumes (the reference rounds that operand to f16 anyway, SURVEY App. B rule 1).
// HBM-bound: reads 4 B/element, writes 2 (or 6) B/element.

#include "kernels.h"
#include "wave_ops.h"

namespace wmi { namespace k {

namespace {

template <int MAXV>   // MAXV = ceil(S / 256) float4 groups per lane
__global__ __launch_bounds__(256) void k_layernorm(const float * __restrict__ x, int rows, int S,
                                                   const float * __restrict__ g, const float * __restrict__ b, float eps,
                                                   __half * __restrict__ out16, float * __restrict__ out32, int rpc_in, int rpc_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float * xr = x + (size_t) row * S;
    // lock-step chunks: input rows chunk * rpc_in + t, output rows chunk * rpc_out + t (the q|k|v GEMM wants every chunk to start on a
    // 16-row boundary, see batch.cpp); rpc_in = 0: the same row
    const int orow = rpc_in > 0 ? (row / rpc_in) * rpc_out + row % rpc_in : row;
    // all loads of the row (x, gain, bias) first, from clamped columns: with the load inside the summation loop hipcc waited
    // for every 16 bytes before requesting the next (MAXV + 2 MAXV dependent round trips per row)
    float4 v[MAXV], gg[MAXV], bb[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4, cc = c < S ? c : 0;
        v[i] = *(const float4 *) (xr + cc); gg[i] = *(const float4 *) (g + cc); bb[i] = *(const float4 *) (b + cc);
    }
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
            float4 y;
            y.x = __fadd_rn(__fmul_rn(v[i].x * scale, gg[i].x), bb[i].x);
            y.y = __fadd_rn(__fmul_rn(v[i].y * scale, gg[i].y), bb[i].y);
            y.z = __fadd_rn(__fmul_rn(v[i].z * scale, gg[i].z), bb[i].z);
            y.w = __fadd_rn(__fmul_rn(v[i].w * scale, gg[i].w), bb[i].w);
            if (out32) *(float4 *) (out32 + (size_t) orow * S + c) = y;
            if (out16) {
                __half2 h01 = __floats2half2_rn(pin_f32(y.x), pin_f32(y.y)), h23 = __floats2half2_rn(pin_f32(y.z), pin_f32(y.w));
                uint2 pk; pk.x = *(uint32_t *) &h01; pk.y = *(uint32_t *) &h23;
                *(uint2 *) (out16 + (size_t) orow * S + c) = pk;
            }
        }
    }
}

} // namespace

void layernorm(const float * x, int rows, int S, const float * g, const float * b, float eps,
               __half * out16, float * out32, hipStream_t st, int rows_per_chunk_in, int rows_per_chunk_out) {
    if (rows <= 0) return;
    const dim3 grid((rows + 3) / 4), block(256);
    const int nv = (S + 255) / 256;
    if (nv <= 2)      hipLaunchKernelGGL((k_layernorm<2>), grid, block, 0, st, x, rows, S, g, b, eps, out16, out32, rows_per_chunk_in, rows_per_chunk_out);
    else if (nv <= 4) hipLaunchKernelGGL((k_layernorm<4>), grid, block, 0, st, x, rows, S, g, b, eps, out16, out32, rows_per_chunk_in, rows_per_chunk_out);
    else              hipLaunchKernelGGL((k_layernorm<8>), grid, block, 0, st, x, rows, S, g, b, eps, out16, out32, rows_per_chunk_in, rows_per_chunk_out);
}

}} // namespace wmi::k
