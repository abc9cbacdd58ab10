// Small-batch decoder kernels for gfx950: token/position embedding gather and
// the weight-streaming "skinny GEMM" used when a decode step carries <= 8 tokens (greedy: 1, beam: <= 8).
//
// A decode step at batch 1 touches every decoder weight exactly once (115.6 MB for base.en incl. the
// 53 MB token-embedding matrix used for the logits) and does 2 FLOP per weight: it is HBM-bound, so
// the kernel is organised around the weight stream, not around MFMA:
//   * each wavefront owns whole output rows; a lane loads 16 contiguous bytes of the row (8 f16
//     weights), so a wave reads 1 KiB per instruction, rows are read front to back exactly once;
//   * four rows are in flight per wavefront (independent 16-byte loads before the first use);
//   * the <= 8 activation rows are staged once per workgroup in LDS as f16 (the reference rounds the
//     activation operand of every mul_mat to f16, SURVEY App. B rule 1); an optional fused LayerNorm
//     prologue produces them straight from the f32 residual stream (saves one launch per sub-block);
//   * f32 accumulation, 64-lane butterfly reduction, then the same fused epilogues as the big GEMM.

#include "kernels.h"

namespace wmi { namespace k {

namespace {

__device__ __forceinline__ float round_f16(float x) { return __half2float(__float2half_rn(x)); }
__device__ __forceinline__ float gelu16(float x) {
    const float xh = round_f16(x);
    const float g  = 0.5f * xh * (1.0f + tanhf(0.79788456080286535587989211986876f * xh * (1.0f + 0.044715f * xh * xh)));
    return round_f16(g);
}

__global__ void k_dec_embed(const int32_t * __restrict__ tokens, const int32_t * __restrict__ pos, int S,
                            const __half * __restrict__ te, const float * __restrict__ pe, float * __restrict__ x) {
    const int i = blockIdx.x;
    const __half * t = te + (size_t) tokens[i] * S;
    const float *  p = pe + (size_t) pos[i] * S;
    for (int c = threadIdx.x; c < S; c += blockDim.x) x[(size_t) i * S + c] = __half2float(t[c]) + p[c];
}

constexpr int ROWS_IN_FLIGHT = 4;

template <int R>
__global__ __launch_bounds__(256) void k_gemv(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half * act = (__half *) smem;                         // [R][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K;
    const int nwaves = gridDim.x * 4;
    const int gw = blockIdx.x * 4 + wave;

    // ---- weight prefetch: the first 16 bytes of this wavefront's first rows do not depend on the activations,
    // so their HBM/MALL latency is overlapped with the prologue below
    uint4 wpre[ROWS_IN_FLIGHT];
    const bool have_pre = gw * ROWS_IN_FLIGHT < a.N && lane * 8 < K;
    if (have_pre) {
#pragma unroll
        for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
            int o = gw * ROWS_IN_FLIGHT + u; if (o > a.N - 1) o = a.N - 1;
            wpre[u] = *(const uint4 *) (a.W + (size_t) o * K + lane * 8);
        }
    }

    // ---- prologue: stage the activation rows as f16
    if (a.ln_g) {                                           // fused LayerNorm of the f32 residual stream
        constexpr int XV = 20;                              // row kept in registers: K <= 64 * 20 = 1280 (every Whisper size)
        for (int r = wave; r < R; r += 4) {
            const int src = a.rows ? a.rows[r] : r;
            const float * xr = a.x32 + (size_t) src * K;
            float xv[XV]; float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < XV; ++j) { const int c = lane + 64 * j; xv[j] = c < K ? xr[c] : 0.0f; sum += xv[j]; }
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum / (float) K;
            float sq = 0.0f;
#pragma unroll
            for (int j = 0; j < XV; ++j) { const int c = lane + 64 * j; if (c < K) { xv[j] -= mean; sq += xv[j] * xv[j]; } }
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            const float sc = 1.0f / sqrtf(sq / (float) K + a.eps);
#pragma unroll
            for (int j = 0; j < XV; ++j) {
                const int c = lane + 64 * j;
                if (c < K) act[r * K + c] = __float2half_rn(__fadd_rn(__fmul_rn(xv[j] * sc, a.ln_g[c]), a.ln_b[c]));
            }
        }
    } else if (a.comb_o) {                                  // fused combine of the split cross-attention partials
        const int H = K / 64, ns = a.comb_ns;
        for (int e = tid; e < R * K; e += 256) {
            const int r = e / K, c = e - r * K, h = c >> 6, dd = c & 63;
            const size_t row = (size_t) r * H + h;
            float o = 0.0f; double l = 0.0;
            for (int s2 = 0; s2 < ns; ++s2) { o += a.comb_o[(row * ns + s2) * 64 + dd]; l += (double) a.comb_l[row * ns + s2]; }
            act[e] = __float2half_rn(o * (float) (1.0 / l));
        }
    } else {
        for (int r = 0; r < R; ++r) {
            const int src = a.rows ? a.rows[r] : r;
            const uint4 * s4 = (const uint4 *) (a.a16 + (size_t) src * K);
            uint4 * d4 = (uint4 *) (act + r * K);
            for (int c = tid; c < K / 8; c += 256) d4[c] = s4[c];
        }
    }
    __syncthreads();

    bool first = have_pre;
    for (int o0 = gw * ROWS_IN_FLIGHT; o0 < a.N; o0 += nwaves * ROWS_IN_FLIGHT) {
        float acc[ROWS_IN_FLIGHT][R];
#pragma unroll
        for (int u = 0; u < ROWS_IN_FLIGHT; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[u][r] = 0.0f;

        for (int c = lane * 8; c < K; c += 512) {
            uint4 w[ROWS_IN_FLIGHT];
            if (first) {
#pragma unroll
                for (int u = 0; u < ROWS_IN_FLIGHT; ++u) w[u] = wpre[u];
                first = false;
            } else {
#pragma unroll
                for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
                    int o = o0 + u; if (o > a.N - 1) o = a.N - 1;
                    w[u] = *(const uint4 *) (a.W + (size_t) o * K + c);
                }
            }
            float av[R][8];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint4 u4 = *(const uint4 *) (act + r * K + c);
                const __half2 * h = (const __half2 *) &u4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); av[r][2 * e] = f.x; av[r][2 * e + 1] = f.y; }
            }
#pragma unroll
            for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
                const __half2 * h = (const __half2 *) &w[u];
                float wf[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); wf[2 * e] = f.x; wf[2 * e + 1] = f.y; }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[u][r] = fmaf(wf[e], av[r][e], acc[u][r]);
            }
        }
#pragma unroll
        for (int u = 0; u < ROWS_IN_FLIGHT; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float v = acc[u][r];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                acc[u][r] = v;
            }
        // epilogue: lane (u * R + r) writes element (row r, column o0 + u)
        if (lane < ROWS_IN_FLIGHT * R) {
            const int u = lane / R, r = lane - u * R;
            const int n = o0 + u;
            float v = 0.0f;
#pragma unroll
            for (int uu = 0; uu < ROWS_IN_FLIGHT; ++uu)
#pragma unroll
                for (int rr = 0; rr < R; ++rr) if (uu == u && rr == r) v = acc[uu][rr];
            if (n < a.N) {
                const float bias = a.bias ? a.bias[n] : 0.0f;
                switch (a.epi) {
                    case EPI_F16_BIAS:       ((__half *) a.C)[(size_t) r * a.ldc + n] = __float2half_rn(v + bias); break;
                    case EPI_F16_BIAS_GELU:  ((__half *) a.C)[(size_t) r * a.ldc + n] = __float2half_rn(gelu16(v + bias)); break;
                    case EPI_F32_BIAS_RESID: ((float *) a.C)[(size_t) r * a.ldc + n] = (v + bias) + a.resid[(size_t) r * a.ldr + n]; break;
                    case EPI_Q_SCALED:       ((__half *) a.C)[(size_t) r * a.ldc + n] = __float2half_rn((v + bias) * a.scale); break;
                    case EPI_QKV_DEC: {
                        // segment decided on a wave-uniform value (the 4 rows of this wave iteration never straddle a
                        // q|k|v boundary: S % 4 == 0) — same precaution as in k_gemm.hip, see DESIGN.md §7
                        const int seg = __builtin_amdgcn_readfirstlane(o0 / a.S);
                        const int c = n - seg * a.S;
                        const int ro = a.row_off ? *a.row_off : 0;          // KV-cache head (device scalar under graph replay)
                        __half * dst; float val;
                        if (seg == 0)      { dst = (__half *) a.C    + (size_t) r * a.ldc;           val = (v + bias) * a.scale; }
                        else if (seg == 1) { dst = (__half *) a.aux  + (size_t) (r + ro) * a.ldaux;  val = v * a.scale; }
                        else               { dst = (__half *) a.aux2 + (size_t) (r + ro) * a.ldaux2; val = v + bias; }
                        dst[c] = __float2half_rn(val);
                    } break;
                    case EPI_LOGITS:         ((float *) a.C)[(size_t) r * a.ldc + n] = v; break;
                    default: break;
                }
            }
        }
    }
}

template <int R>
void launch_gemv(const GemvArgs & a, hipStream_t st) {
    const size_t smem = (size_t) R * a.K * sizeof(__half);
    int blocks = (a.N + 4 * ROWS_IN_FLIGHT - 1) / (4 * ROWS_IN_FLIGHT);
    if (blocks > 2048) blocks = 2048;
    static size_t attr_bytes = 0;
    if (smem > 48 * 1024 && smem > attr_bytes) {
        (void) hipFuncSetAttribute((const void *) k_gemv<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
        attr_bytes = smem;
    }
    hipLaunchKernelGGL((k_gemv<R>), dim3(blocks), dim3(256), smem, st, a);
}

} // namespace

void dec_embed(const int32_t * tokens, const int32_t * pos, int n, int S, const __half * te, const float * pe,
               float * x, hipStream_t st) {
    hipLaunchKernelGGL(k_dec_embed, dim3(n), dim3(256), 0, st, tokens, pos, S, te, pe, x);
}

void gemv(const GemvArgs & a, hipStream_t st) {
    switch (a.n) {
        case 1: launch_gemv<1>(a, st); break;
        case 2: launch_gemv<2>(a, st); break;
        case 3: launch_gemv<3>(a, st); break;
        case 4: launch_gemv<4>(a, st); break;
        case 5: launch_gemv<5>(a, st); break;
        case 6: launch_gemv<6>(a, st); break;
        case 7: launch_gemv<7>(a, st); break;
        case 8: launch_gemv<8>(a, st); break;
        default: break;
    }
}

}} // namespace wmi::k
