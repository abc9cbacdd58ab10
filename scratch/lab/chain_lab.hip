// Lab (round 3): what does one launch of the decode step's dependent chain cost on the DEVICE, and what moves it?
//   1. boundary: hipGraph-replayed chains of empty kernels (grid 1 .. 768 workgroups, 64 / 256 threads, small / 300-byte kernarg)
//   2. a base.en-shaped decoder layer chain (LN+qkv -> out -> LN+cq -> co -> LN+fc1 -> fc2, 6 layers, distinct weights, optional
//      53 MB vocabulary sweep per step) as graph replays, in variants:
//        struct kernarg (the product's GemvArgs style) vs scalar arguments with kernarg preload (-mllvm -amdgpu-kernarg-preload-count)
//        256-thread vs 64-thread workgroups, 4 / 2 / 1 weight rows per wavefront, sc1 (write-through) activation stores
//      with in-kernel wall_clock64 stamps (first instruction / x arrived / last store issued) for the body-vs-boundary split
//   3. the pick kernel's hand-over to the host: record + __threadfence_system + seq + fence  vs  two self-tagged 16-byte halves
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 chain_lab.hip -o chain_lab && ./chain_lab
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>
#include "../../godot-whisper_amd/csrc/wave_ops.h"
using wmi::k::xor_lane;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------- 1. empty kernels
struct Fat { const void * p[8]; int v[60]; };            // ~304 bytes by value
__global__ void k_empty_s(int * p) { if (p == (int *) 1) p[0] = 0; }
__global__ void k_empty_f(Fat f) { if (f.p[0] == (void *) 1) ((int *) f.p[1])[0] = f.v[59]; }

// ---------------------------------------------------------------------------------------------- 2. GEMV chain
struct GA { const __half * W; const void * xin; void * out; const float * resid; const float * g; const float * b; int N; unsigned long long * stamps; int pad[40]; };

template <bool DPP, int M> __device__ __forceinline__ float sx(float v) { if constexpr (DPP) return xor_lane<M>(v); else return __shfl_xor(v, M); }
template <bool DPP> __device__ __forceinline__ float wsum(float v) {
    v += sx<DPP, 32>(v); v += sx<DPP, 16>(v); v += sx<DPP, 8>(v); v += sx<DPP, 4>(v); v += sx<DPP, 2>(v); v += sx<DPP, 1>(v); return v;
}
__global__ void k_selftest(const float * in, int * bad) {
    const float x = in[threadIdx.x];
    int b = 0;
    b += __float_as_int(xor_lane<1>(x)) != __float_as_int(__shfl_xor(x, 1));
    b += (__float_as_int(xor_lane<2>(x)) != __float_as_int(__shfl_xor(x, 2))) << 4;
    b += (__float_as_int(xor_lane<4>(x)) != __float_as_int(__shfl_xor(x, 4))) << 8;
    b += (__float_as_int(xor_lane<8>(x)) != __float_as_int(__shfl_xor(x, 8))) << 12;
    b += (__float_as_int(xor_lane<16>(x)) != __float_as_int(__shfl_xor(x, 16))) << 16;
    b += (__float_as_int(xor_lane<32>(x)) != __float_as_int(__shfl_xor(x, 32))) << 20;
    bad[threadIdx.x] = b;
}
__device__ __forceinline__ float round_f16(float x) { return __half2float(__float2half_rn(x)); }

template <int RIF, int NCH, bool LN, int EPI, bool SC1, bool DPP = false>
__device__ __forceinline__ void gv_body(const __half * __restrict__ W, const void * __restrict__ xin, void * __restrict__ out,
                                        const float * __restrict__ resid, const float * __restrict__ g, const float * __restrict__ b,
                                        int N, unsigned long long * stamps, int wpb) {
    const unsigned long long t0 = stamps ? wall_clock64() : 0ull;
    constexpr int K = 512 * NCH;
    const int lane = threadIdx.x & 63, gw = blockIdx.x * wpb + (threadIdx.x >> 6);
    const int r0 = gw * RIF;
    if (r0 >= N) return;
    uint4 w[NCH][RIF];
#pragma unroll
    for (int t = 0; t < NCH; ++t)
#pragma unroll
        for (int u = 0; u < RIF; ++u) w[t][u] = *(const uint4 *) (W + (size_t) (r0 + u) * K + 512 * t + lane * 8);
    constexpr int LPR = 64 / RIF;
    const int wrow = lane / LPR; const bool writer = (lane % LPR) == 0;
    float resid_pre = 0.f;
    if (EPI == 1 && writer) resid_pre = resid[r0 + wrow];
    float av[NCH][8];
    if (LN) {
        const float * x = (const float *) xin;
        float xv[8], gv[8], bv[8];
        { const float4 a0 = *(const float4 *) (x + lane * 8), a1 = *(const float4 *) (x + lane * 8 + 4);
          xv[0] = a0.x; xv[1] = a0.y; xv[2] = a0.z; xv[3] = a0.w; xv[4] = a1.x; xv[5] = a1.y; xv[6] = a1.z; xv[7] = a1.w; }
        { const float4 a0 = *(const float4 *) (g + lane * 8), a1 = *(const float4 *) (g + lane * 8 + 4);
          gv[0] = a0.x; gv[1] = a0.y; gv[2] = a0.z; gv[3] = a0.w; gv[4] = a1.x; gv[5] = a1.y; gv[6] = a1.z; gv[7] = a1.w; }
        { const float4 a0 = *(const float4 *) (b + lane * 8), a1 = *(const float4 *) (b + lane * 8 + 4);
          bv[0] = a0.x; bv[1] = a0.y; bv[2] = a0.z; bv[3] = a0.w; bv[4] = a1.x; bv[5] = a1.y; bv[6] = a1.z; bv[7] = a1.w; }
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += xv[e];
        sum = wsum<DPP>(sum);
        const float mean = sum / 512.f;
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[e] -= mean; sq += xv[e] * xv[e]; }
        sq = wsum<DPP>(sq);
        const float sc = 1.0f / sqrtf(sq / 512.f + 1e-5f);
#pragma unroll
        for (int e = 0; e < 8; ++e) av[0][e] = round_f16(xv[e] * sc * gv[e] + bv[e]);
    } else {
        const __half * x = (const __half *) xin;
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const uint4 u4 = *(const uint4 *) (x + 512 * t + lane * 8);
            const __half2 * h = (const __half2 *) &u4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); av[t][2 * e] = f.x; av[t][2 * e + 1] = f.y; }
        }
    }
    const unsigned long long t1 = stamps ? wall_clock64() : 0ull;
    float acc[RIF];
#pragma unroll
    for (int u = 0; u < RIF; ++u) acc[u] = 0.f;
#pragma unroll
    for (int t = 0; t < NCH; ++t)
#pragma unroll
        for (int u = 0; u < RIF; ++u) {
            const __half2 * h = (const __half2 *) &w[t][u];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); acc[u] = fmaf(f.x, av[t][2 * e], acc[u]); acc[u] = fmaf(f.y, av[t][2 * e + 1], acc[u]); }
        }
    float v;
    if (RIF == 4) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { const bool hi = lane & 32; const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2]; acc[u] = keep + sx<DPP, 32>(send); }
        { const bool hi = lane & 16; const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1]; v = keep + sx<DPP, 16>(send); }
        v += sx<DPP, 8>(v); v += sx<DPP, 4>(v); v += sx<DPP, 2>(v); v += sx<DPP, 1>(v);
    } else if (RIF == 2) {
        { const bool hi = lane & 32; const float keep = hi ? acc[RIF - 1] : acc[0], send = hi ? acc[0] : acc[RIF - 1]; v = keep + sx<DPP, 32>(send); }
        v += sx<DPP, 16>(v); v += sx<DPP, 8>(v); v += sx<DPP, 4>(v); v += sx<DPP, 2>(v); v += sx<DPP, 1>(v);
    } else {
        v = wsum<DPP>(acc[0]);
    }
    if (writer) {
        const int n = r0 + wrow;
        if (EPI == 1) {
            const float r = v * 0.01f + resid_pre;
            if (SC1) __hip_atomic_store((float *) out + n, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ((float *) out)[n] = r;
        } else {
            const __half hv = __float2half_rn(v * 0.05f);
            if (SC1) { unsigned short us = __half_as_ushort(hv); __hip_atomic_store((unsigned short *) out + n, us, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else ((__half *) out)[n] = hv;
        }
    }
    if (stamps && lane == 0) {
        const unsigned long long t2 = wall_clock64();
        unsigned long long * s = stamps + (size_t) gw * 3;
        s[0] = t0; s[1] = t1; s[2] = t2;
    }
}

template <int TPB, int RIF, int NCH, bool LN, int EPI, bool SC1>
__global__ __launch_bounds__(TPB) void k_gv_struct(const GA a) {
    gv_body<RIF, NCH, LN, EPI, SC1>(a.W, a.xin, a.out, a.resid, a.g, a.b, a.N, a.stamps, TPB / 64);
}
template <int TPB, int RIF, int NCH, bool LN, int EPI, bool SC1, bool DPP = false>
__global__ __launch_bounds__(TPB) void k_gv_scalar(const __half * __restrict__ W, const void * __restrict__ xin, void * __restrict__ out,
                                                   const float * __restrict__ resid, const float * __restrict__ g, const float * __restrict__ b,
                                                   int N, unsigned long long * stamps) {
    gv_body<RIF, NCH, LN, EPI, SC1, DPP>(W, xin, out, resid, g, b, N, stamps, TPB / 64);
}

// 53 MB sweep standing in for the vocabulary projection (evicts the layer weights from L2 like the real step)
__global__ __launch_bounds__(256) void k_sweep(const uint4 * __restrict__ p, size_t n16, float * __restrict__ out) {
    unsigned acc = 0;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = 1.f;
}

// ---------------------------------------------------------------------------------------------- 3. host hand-over
struct Rec { int id, tid; float p, plog, pt, ptsum; int forced, seq; };
__global__ void k_rec_fence(Rec * host, const int * seqp, float * x) {
    if (threadIdx.x == 0) {
        Rec r; r.id = 5; r.tid = 6; r.p = 0.5f; r.plog = -1.f; r.pt = 0.1f; r.ptsum = 0.2f; r.forced = 0; r.seq = host->seq;
        *host = r;
        __threadfence_system();
        *(volatile int *) &host->seq = *seqp;
        __threadfence_system();
    }
    x[threadIdx.x] = 1.0f;
}
__global__ void k_rec_tagged(Rec * host, const int * seqp, float * x) {
    if (threadIdx.x == 0) {
        const int s = *seqp;
        int4 h0, h1;
        h0.x = 5; h0.y = 6; h0.z = __float_as_int(0.5f); h0.w = s;
        h1.x = __float_as_int(-1.f); h1.y = __float_as_int(0.1f); h1.z = __float_as_int(0.2f); h1.w = s;
        ((int4 *) host)[0] = h0; ((int4 *) host)[1] = h1;
    }
    x[threadIdx.x] = 1.0f;
}
__global__ void k_rec_none(Rec * host, const int * seqp, float * x) { x[threadIdx.x] = 1.0f + (float) *seqp; }

// ---------------------------------------------------------------------------------------------- host
static double replay_us(hipGraphExec_t ex, hipStream_t s, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ex, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ex, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return (double) ms * 1000.0 / reps;
}
template <typename F> static hipGraphExec_t capture(hipStream_t s, F && f) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    f();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    return ex;
}

struct Layer { __half * wqkv, * wo, * wcq, * wco, * wfc1, * wfc2; };
struct Bufs { float * x; __half * q, * c, * h; float * g, * b; unsigned long long * stamps; };

// one launch of the chain; variant selects the kernel flavour
enum Variant { V_STRUCT256 = 0, V_SCALAR256, V_SCALAR64, V_SCALAR64_R2, V_SCALAR64_R1, V_SCALAR256_SC1, V_SCALAR64_R2_SC1, V_SCALAR256_R2, V_SCALAR256_R4_DPP, V_SCALAR256_R2_DPP, V_SCALAR64_R2_DPP, V_COUNT };
static const char * vname[V_COUNT] = { "struct-kernarg 256thr RIF4 (product today)", "scalar+preload 256thr RIF4", "scalar+preload 64thr RIF4",
                                       "scalar+preload 64thr RIF2", "scalar+preload 64thr RIF1", "scalar+preload 256thr RIF4 sc1-stores",
                                       "scalar+preload 64thr RIF2 sc1-stores", "scalar+preload 256thr RIF2",
                                       "scalar+preload 256thr RIF4 DPP-reductions", "scalar+preload 256thr RIF2 DPP-reductions", "scalar+preload 64thr RIF2 DPP-reductions" };

template <int NCH, bool LN, int EPI>
static void launch_gv(int variant, hipStream_t s, const __half * W, const void * xin, void * out, const float * resid, const float * g, const float * b,
                      int N, unsigned long long * stamps) {
    auto blocks = [&](int tpb, int rif) { const int rows_per_block = (tpb / 64) * rif; return (N + rows_per_block - 1) / rows_per_block; };
    switch (variant) {
        case V_STRUCT256: { GA a{}; a.W = W; a.xin = xin; a.out = out; a.resid = resid; a.g = g; a.b = b; a.N = N; a.stamps = stamps;
            hipLaunchKernelGGL((k_gv_struct<256, 4, NCH, LN, EPI, false>), dim3(blocks(256, 4)), dim3(256), 0, s, a); } break;
        case V_SCALAR256:     hipLaunchKernelGGL((k_gv_scalar<256, 4, NCH, LN, EPI, false>), dim3(blocks(256, 4)), dim3(256), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        case V_SCALAR64:      hipLaunchKernelGGL((k_gv_scalar<64, 4, NCH, LN, EPI, false>), dim3(blocks(64, 4)), dim3(64), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        case V_SCALAR64_R2:   hipLaunchKernelGGL((k_gv_scalar<64, 2, NCH, LN, EPI, false>), dim3(blocks(64, 2)), dim3(64), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        case V_SCALAR64_R1:   hipLaunchKernelGGL((k_gv_scalar<64, 1, NCH, LN, EPI, false>), dim3(blocks(64, 1)), dim3(64), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        case V_SCALAR256_SC1: hipLaunchKernelGGL((k_gv_scalar<256, 4, NCH, LN, EPI, true>), dim3(blocks(256, 4)), dim3(256), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        case V_SCALAR64_R2_SC1: hipLaunchKernelGGL((k_gv_scalar<64, 2, NCH, LN, EPI, true>), dim3(blocks(64, 2)), dim3(64), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        case V_SCALAR256_R2:  hipLaunchKernelGGL((k_gv_scalar<256, 2, NCH, LN, EPI, false>), dim3(blocks(256, 2)), dim3(256), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        case V_SCALAR256_R4_DPP: hipLaunchKernelGGL((k_gv_scalar<256, 4, NCH, LN, EPI, false, true>), dim3(blocks(256, 4)), dim3(256), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        case V_SCALAR256_R2_DPP: hipLaunchKernelGGL((k_gv_scalar<256, 2, NCH, LN, EPI, false, true>), dim3(blocks(256, 2)), dim3(256), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        case V_SCALAR64_R2_DPP:  hipLaunchKernelGGL((k_gv_scalar<64, 2, NCH, LN, EPI, false, true>), dim3(blocks(64, 2)), dim3(64), 0, s, W, xin, out, resid, g, b, N, stamps); break;
        default: break;
    }
}

constexpr int S = 512, NSTAMP_WAVES = 2048;
static int L = 6;                 // decoder layers of the chain: 6 = base.en (44 MB of weights); argv[3] scales the footprint
static void enqueue_step(int variant, hipStream_t s, const std::vector<Layer> & ly, const Bufs & B, bool stamp, const uint4 * sweep, size_t sweep16, int kind_mask = 63) {
    int li = 0;
    auto st = [&]() -> unsigned long long * { return stamp ? B.stamps + (size_t) (li++) * NSTAMP_WAVES * 3 : nullptr; };
    for (int l = 0; l < L; ++l) {
        if (kind_mask & 1)  launch_gv<1, true, 0>(variant, s, ly[l].wqkv, B.x, B.q, nullptr, B.g, B.b, 3 * S, st());       // LN + q|k|v
        if (kind_mask & 2)  launch_gv<1, false, 1>(variant, s, ly[l].wo, B.q, B.x, B.x, nullptr, nullptr, S, st());        // (attention) + out + resid
        if (kind_mask & 4)  launch_gv<1, true, 0>(variant, s, ly[l].wcq, B.x, B.c, nullptr, B.g, B.b, S, st());            // LN + cross query
        if (kind_mask & 8)  launch_gv<1, false, 1>(variant, s, ly[l].wco, B.c, B.x, B.x, nullptr, nullptr, S, st());       // cross out + resid
        if (kind_mask & 16) launch_gv<1, true, 0>(variant, s, ly[l].wfc1, B.x, B.h, nullptr, B.g, B.b, 4 * S, st());       // LN + mlp.0
        if (kind_mask & 32) launch_gv<4, false, 1>(variant, s, ly[l].wfc2, B.h, B.x, B.x, nullptr, nullptr, S, st());      // mlp.2 + resid
    }
    if (sweep) hipLaunchKernelGGL(k_sweep, dim3(768), dim3(256), 0, s, sweep, sweep16, B.x + 4096);
}

int main(int argc, char ** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz wall_clock_rate %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, 0);
    int wall_khz = 100000; (void) hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate %d kHz\n", wall_khz);

    {   // xor_lane<M> == __shfl_xor(., M) ?
        float * din; int * dbad; CK(hipMalloc(&din, 256)); CK(hipMalloc(&dbad, 256));
        float hin[64]; for (int i = 0; i < 64; ++i) hin[i] = 1.0f + i * 0.37f;
        CK(hipMemcpy(din, hin, 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, s, din, dbad); CK(hipStreamSynchronize(s));
        int hb[64]; CK(hipMemcpy(hb, dbad, 256, hipMemcpyDeviceToHost));
        int any = 0; for (int i = 0; i < 64; ++i) any |= hb[i];
        printf("xor_lane self-test: mismatch mask 0x%06x (0 = all six exchanges equal __shfl_xor)\n", any);
    }
    const bool quick = argc > 2;
    if (argc > 3) L = atoi(argv[3]);
    const bool arena = argc > 4;            // all weights in ONE allocation (the product's arena) instead of one hipMalloc per matrix
    const int NL = 6 * L;
    // ---- 1. boundaries
    int * dp; CK(hipMalloc(&dp, 4096)); CK(hipMemset(dp, 0, 4096));
    {
        const int grids[] = {1, 32, 128, 256, 768};
        const int tpbs[] = {64, 256};
        if (!quick) for (int tpb : tpbs) for (int g : grids) {
            hipGraphExec_t ex = capture(s, [&]() { for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty_s, dim3(g), dim3(tpb), 0, s, dp); });
            const double us = replay_us(ex, s, 20) / 200.0;
            Fat f{}; f.p[0] = dp; f.p[1] = dp;
            hipGraphExec_t ex2 = capture(s, [&]() { for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty_f, dim3(g), dim3(tpb), 0, s, f); });
            const double us2 = replay_us(ex2, s, 20) / 200.0;
            printf("boundary: empty kernel chain (graph)  grid %4d x %3d thr : %.3f us/launch  (300-byte kernarg: %.3f)\n", g, tpb, us, us2);
            CK(hipGraphExecDestroy(ex)); CK(hipGraphExecDestroy(ex2));
        }
        // eager for comparison
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty_s, dim3(32), dim3(256), 0, s, dp);
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty_s, dim3(32), dim3(256), 0, s, dp);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("boundary: empty kernel chain (eager, host-paced) grid 32 x 256: %.3f us/launch\n", ms * 1000.0 / 2000);
    }

    // ---- 2. layer chain
    std::vector<Layer> ly(L);
    char * arena_base = nullptr; size_t arena_off = 0;
    if (arena) CK(hipMalloc(&arena_base, (size_t) L * 14 * S * S * 2 + 4096));
    auto walloc = [&](size_t rows, size_t cols) {
        __half * p;
        if (arena) { p = (__half *) (arena_base + arena_off); arena_off += rows * cols * 2; }
        else CK(hipMalloc(&p, rows * cols * 2));
        std::vector<__half> h(rows * cols);
        for (size_t i = 0; i < h.size(); ++i) h[i] = __float2half((float) ((int) ((i * 2654435761u) >> 20 & 255) - 128) / 2048.0f);
        CK(hipMemcpy(p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        return p;
    };
    for (int l = 0; l < L; ++l) { ly[l].wqkv = walloc(3 * S, S); ly[l].wo = walloc(S, S); ly[l].wcq = walloc(S, S); ly[l].wco = walloc(S, S); ly[l].wfc1 = walloc(4 * S, S); ly[l].wfc2 = walloc(S, 4 * S); }
    Bufs B{};
    CK(hipMalloc(&B.x, 8192 * 4)); CK(hipMalloc(&B.q, 4096 * 2)); CK(hipMalloc(&B.c, 4096 * 2)); CK(hipMalloc(&B.h, 4096 * 2));
    CK(hipMalloc(&B.g, S * 4)); CK(hipMalloc(&B.b, S * 4));
    CK(hipMalloc(&B.stamps, (size_t) NL * NSTAMP_WAVES * 3 * 8));
    {
        std::vector<float> one(S, 1.0f), zero(S, 0.0f), x0(8192);
        for (int i = 0; i < 8192; ++i) x0[i] = (float) ((i * 37) % 101) / 101.0f - 0.5f;
        CK(hipMemcpy(B.g, one.data(), S * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B.b, zero.data(), S * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.x, x0.data(), 8192 * 4, hipMemcpyHostToDevice));
        CK(hipMemset(B.q, 0, 8192)); CK(hipMemset(B.c, 0, 8192)); CK(hipMemset(B.h, 0, 8192));
    }
    const size_t sweep_bytes = (size_t) 51864 * 512 * 2;
    uint4 * sweep; CK(hipMalloc(&sweep, sweep_bytes)); CK(hipMemset(sweep, 1, sweep_bytes));

    std::vector<float> xref;
    for (int v = 0; v < V_COUNT; ++v) {
        if (quick && !(v == 0 || v == 1 || v >= V_SCALAR256_R2)) continue;
        // reset x so that every variant computes the same thing
        std::vector<float> x0(8192);
        for (int i = 0; i < 8192; ++i) x0[i] = (float) ((i * 37) % 101) / 101.0f - 0.5f;
        CK(hipMemcpy(B.x, x0.data(), 8192 * 4, hipMemcpyHostToDevice));
        enqueue_step(v, s, ly, B, false, nullptr, 0); CK(hipStreamSynchronize(s));
        std::vector<float> xo(S); CK(hipMemcpy(xo.data(), B.x, S * 4, hipMemcpyDeviceToHost));
        if (v == 0) xref = xo;
        double md = 0; for (int i = 0; i < S; ++i) md = std::max(md, (double) fabsf(xo[i] - xref[i]));
        hipGraphExec_t ex = capture(s, [&]() { enqueue_step(v, s, ly, B, false, nullptr, 0); });
        hipGraphExec_t exs = capture(s, [&]() { enqueue_step(v, s, ly, B, false, sweep, sweep_bytes / 16); });
        hipGraphExec_t exw = capture(s, [&]() { hipLaunchKernelGGL(k_sweep, dim3(768), dim3(256), 0, s, sweep, sweep_bytes / 16, B.x + 4096); });
        const double us = replay_us(ex, s, reps), uss = replay_us(exs, s, reps), usw = replay_us(exw, s, reps);
        printf("chain[%d] %-46s: %d launches %.1f us = %.3f us/launch ; with 53 MB sweep per step %.1f us (sweep alone %.1f) -> %.3f us/launch   (max|dx| vs variant 0: %.2e)\n",
               v, vname[v], NL, us, us / NL, uss, usw, (uss - usw) / NL, md);
        // per kernel kind (own chain of 6 launches x 6 kinds is not a dependent chain of that kind alone, but the shares are indicative)
        for (int kind = 0; kind < 6; ++kind) {
            hipGraphExec_t exk = capture(s, [&]() { for (int r = 0; r < 6; ++r) enqueue_step(v, s, ly, B, false, nullptr, 0, 1 << kind); });
            const double usk = replay_us(exk, s, reps / 2 + 1) / (6.0 * L);
            static const char * kn[6] = {"LN+qkv 1536x512", "out 512x512+res", "LN+cq 512x512", "co 512x512+res", "LN+fc1 2048x512", "fc2 512x2048+res"};
            printf("    kind %-18s %.3f us/launch\n", kn[kind], usk);
            CK(hipGraphExecDestroy(exk));
        }
        // stamps: one replay of the stamped graph after a warm replay
        CK(hipMemset(B.stamps, 0, (size_t) NL * NSTAMP_WAVES * 3 * 8));
        hipGraphExec_t ext = capture(s, [&]() { enqueue_step(v, s, ly, B, true, nullptr, 0); });
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ext, s));
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> st((size_t) NL * NSTAMP_WAVES * 3);
        CK(hipMemcpy(st.data(), B.stamps, st.size() * 8, hipMemcpyDeviceToHost));
        double body = 0, gap = 0, tox = 0, firstlast = 0; unsigned long long prev_end = 0; int ngap = 0;
        for (int li = 0; li < NL; ++li) {
            unsigned long long mn = ~0ull, mx = 0, mnx = ~0ull, mxs = 0;
            for (int w = 0; w < NSTAMP_WAVES; ++w) {
                const unsigned long long * p = &st[((size_t) li * NSTAMP_WAVES + w) * 3];
                if (p[0] == 0) continue;
                mn = std::min(mn, p[0]); mx = std::max(mx, p[2]); mnx = std::min(mnx, p[1]); mxs = std::max(mxs, p[0]);
            }
            if (mx == 0) continue;
            body += (double) (mx - mn); tox += (double) (mnx - mn); firstlast += (double) (mxs - mn);
            if (prev_end) { gap += (double) ((long long) mn - (long long) prev_end); ngap++; }
            prev_end = mx;
        }
        const double tick_us = 1000.0 / (double) wall_khz;
        printf("    stamps: body (first wave start -> last wave end) %.3f us avg ; start -> x arrived %.3f ; first -> last wave start %.3f ; boundary (end -> next start) %.3f us avg\n",
               body / NL * tick_us, tox / NL * tick_us, firstlast / NL * tick_us, ngap ? gap / ngap * tick_us : 0.0);
        CK(hipGraphExecDestroy(ex)); CK(hipGraphExecDestroy(exs)); CK(hipGraphExecDestroy(exw)); CK(hipGraphExecDestroy(ext));
    }

    // ---- 3. host hand-over
    {
        Rec * hrec; CK(hipHostMalloc((void **) &hrec, 64, hipHostMallocDefault)); memset(hrec, 0, 64);
        int * dseq; CK(hipMalloc(&dseq, 4)); CK(hipMemset(dseq, 0, 4));
        float * dx; CK(hipMalloc(&dx, 4096));
        hipGraphExec_t a = capture(s, [&]() { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_rec_fence, dim3(1), dim3(64), 0, s, hrec, dseq, dx); });
        hipGraphExec_t b = capture(s, [&]() { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_rec_tagged, dim3(1), dim3(64), 0, s, hrec, dseq, dx); });
        hipGraphExec_t c = capture(s, [&]() { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_rec_none, dim3(1), dim3(64), 0, s, hrec, dseq, dx); });
        printf("host hand-over: record + 2 system fences %.3f us/launch ; two tagged 16-byte halves, no fence %.3f ; no host write %.3f\n",
               replay_us(a, s, 20) / 100, replay_us(b, s, 20) / 100, replay_us(c, s, 20) / 100);
    }
    return 0;
}
