// Lab: the "front" of a base.en decoder layer of the one-row step — LayerNorm + q|k|v projection (1536 x 512),
// self-attention over a short cache, out projection (512 x 512) + residual — as
//   (A) the product's two launches: K1 = LN + q|k|v on 96 workgroups, K2 = 32 workgroups that each recompute the attention of all eight
//       heads (one per wavefront) and take 16 rows of the out projection;
//   (B) ONE launch on G = 48 (or 96) workgroups of 8 wavefronts (on one XCD / two / four XCDs / spread): phase 1 = LN + q|k|v (32 or 16 rows per workgroup),
//       hand-off of a head's q, k, v (96 data-tagged 8-byte granules {f16 pair, tag}) to ONE wavefront per head, which attends once and
//       hands the head's 64 values on (32 granules), phase 3 = the out projection on 32 of the workgroups (weights requested at the start),
//       each sweeping the 256 granules of the attention row once.
// A chain of NB such blocks (the block's output row is the next block's input), microseconds per block, and for (B) per-edge stamps of the
// last block (s_memrealtime, 100 MHz).  Arithmetic is the same in both forms (same lanes, same order): the rows must agree bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off narrow_phases.hip -o narrow_phases && ./narrow_phases
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int S = 512, H = 8, NKV = 16;                     // NKV cached keys + the new one
typedef unsigned long long u64;

__device__ __forceinline__ u64 wall() { u64 t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// LayerNorm of the row held as 8 values per lane (columns lane * 8 ..)
__device__ __forceinline__ void ln8(const float * x, float (&a)[8], int lane) {
    const float4 p = *(const float4 *) (x + lane * 8), q = *(const float4 *) (x + lane * 8 + 4);
    float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    const float mean = wave_sum(s) / S;
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] -= mean; sq += v[e] * v[e]; }
    const float sc = 1.0f / sqrtf(wave_sum(sq) / S + 1e-5f);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = v[e] * sc;
}
__device__ __forceinline__ float dot8(const uint4 w, const float (&x)[8]) {
    const __half2 * h = (const __half2 *) &w;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float2 f = __half22float2(h[j]); acc = fmaf(f.x, x[2 * j], acc); acc = fmaf(f.y, x[2 * j + 1], acc); }
    return acc;
}

// the attention of one head on one wavefront: q, k_new, v_new as 64 floats each in LDS (qkv[0..63], [64..127], [128..191]);
// cached keys / values f16 [NKV][S]; result: lane d holds output d
__device__ __forceinline__ float attend(const float * qkv, const uint4 (&kc)[8], const __half (&vcol)[NKV], int lane) {
    // lane j < NKV: score of cached key j (its 64 dims in kc); lane NKV: the new key
    float s = -INFINITY;
    if (lane <= NKV) {
        float d = 0.f;
        if (lane < NKV) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const __half2 * h = (const __half2 *) &kc[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); d = fmaf(f.x, qkv[c * 8 + 2 * e], d); d = fmaf(f.y, qkv[c * 8 + 2 * e + 1], d); }
            }
        } else {
            for (int c = 0; c < 64; ++c) d = fmaf(qkv[64 + c], qkv[c], d);
        }
        s = d * 0.125f;
    }
    const float m = wave_max(s);
    const float e = lane <= NKV ? __expf(s - m) : 0.f;
    const float l = wave_sum(e);
    const float p = e / l;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NKV; ++j) acc = fmaf(__shfl(p, j), __half2float(vcol[j]), acc);
    acc = fmaf(__shfl(p, NKV), qkv[128 + lane], acc);
    return acc;
}

// ---------------------------------------------------------------------------------------------- (A) two launches
__global__ __launch_bounds__(256) void k1_qkv(const float * __restrict__ x, const __half * __restrict__ W1, __half * __restrict__ qkv) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint4 w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = *(const uint4 *) (W1 + (size_t) (gw * 4 + u) * S + lane * 8);
    float a[8]; ln8(x, a, lane);
#pragma unroll
    for (int u = 0; u < 4; ++u) { const float r = wave_sum(dot8(w[u], a)); if (lane == 0) qkv[gw * 4 + u] = __float2half(r); }
}
__global__ __launch_bounds__(512) void k2_attn_out(const float * __restrict__ x, const __half * __restrict__ qkv, const __half * __restrict__ kcache,
                                                   const __half * __restrict__ vcache, const __half * __restrict__ W2, float * __restrict__ xout) {
    __shared__ float att[S];
    __shared__ float hq[H][192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint4 w[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) w[u] = *(const uint4 *) (W2 + (size_t) (blockIdx.x * 16 + wave * 2 + u) * S + lane * 8);
    const float res0 = x[blockIdx.x * 16 + wave * 2], res1 = x[blockIdx.x * 16 + wave * 2 + 1];
    const int h = wave;
    uint4 kc[8]; __half vcol[NKV];
    {
        const int j = lane < NKV ? lane : 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) kc[c] = *(const uint4 *) (kcache + (size_t) j * S + h * 64 + c * 8);
#pragma unroll
        for (int j2 = 0; j2 < NKV; ++j2) vcol[j2] = vcache[(size_t) j2 * S + h * 64 + lane];
    }
    hq[h][lane] = __half2float(qkv[h * 64 + lane]); hq[h][64 + lane] = __half2float(qkv[S + h * 64 + lane]); hq[h][128 + lane] = __half2float(qkv[2 * S + h * 64 + lane]);
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // (the wavefront's own LDS writes)
    att[h * 64 + lane] = __half2float(__float2half(attend(hq[h], kc, vcol, lane)));
    __syncthreads();
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = att[lane * 8 + e];
    const float r0 = wave_sum(dot8(w[0], a)), r1 = wave_sum(dot8(w[1], a));
    if (lane == 0) { xout[blockIdx.x * 16 + wave * 2] = r0 + res0; xout[blockIdx.x * 16 + wave * 2 + 1] = r1 + res1; }
}

// ---------------------------------------------------------------------------------------------- (B) one narrow launch
// granules: gq [768] (q|k|v as f16 pairs), ga [256] (attention row as f16 pairs); {pair, tag}
struct Stamps { u64 t0, p1, a_in, a_out, o_in, t1; };
template <int RPW>
__global__ __launch_bounds__(512) void k_front_narrow(const float * __restrict__ x, const __half * __restrict__ W1, const __half * __restrict__ kcache,
                                                      const __half * __restrict__ vcache, const __half * __restrict__ W2, float * __restrict__ xout,
                                                      u64 * __restrict__ gq, u64 * __restrict__ ga, unsigned tag, int stride, int phase3_all,
                                                      Stamps * __restrict__ st, int * __restrict__ err) {
    if (blockIdx.x % stride != 0) return;
    __shared__ float att[S];
    __shared__ float hq[192];
    const int wg = blockIdx.x / stride, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u64 t0 = wall();
    // loads: the row first, then this workgroup's rows of W1, of W2 (phase 3), the cached K / V of its head (phase 2)
    const bool p3 = wg < 32, p2 = wg < H && wave == 0;
    float a[8];
    uint4 w[RPW], w2[2];
#pragma unroll
    for (int u = 0; u < RPW; ++u) w[u] = *(const uint4 *) (W1 + (size_t) (wg * 8 * RPW + wave * RPW + u) * S + lane * 8);
    const int orow = (p3 ? wg : 0) * 16 + wave * 2;
#pragma unroll
    for (int u = 0; u < 2; ++u) w2[u] = *(const uint4 *) (W2 + (size_t) (orow + u) * S + lane * 8);
    const float res0 = x[orow], res1 = x[orow + 1];
    uint4 kc[8]; __half vcol[NKV];
    {
        const int h = p2 ? wg : 0, j = lane < NKV ? lane : 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) kc[c] = *(const uint4 *) (kcache + (size_t) j * S + h * 64 + c * 8);
#pragma unroll
        for (int j2 = 0; j2 < NKV; ++j2) vcol[j2] = vcache[(size_t) j2 * S + h * 64 + lane];
    }
    ln8(x, a, lane);
    // ---- phase 1: rows (wg * 8 + wave) * RPW .. + RPW -> RPW / 2 granules
    {
        float r[RPW];
#pragma unroll
        for (int u = 0; u < RPW; ++u) r[u] = wave_sum(dot8(w[u], a));
        if (lane < RPW / 2) {
            const __half lo = __float2half(RPW == 2 || lane == 0 ? r[0] : r[RPW - 2]), hi = __float2half(RPW == 2 || lane == 0 ? r[1] : r[RPW - 1]);
            const u64 g = ((u64) tag << 32) | (u64) ((unsigned) __half_as_ushort(lo) | ((unsigned) __half_as_ushort(hi) << 16));
            __hip_atomic_store(gq + (wg * 8 + wave) * (RPW / 2) + lane, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const u64 tp1 = wall();
    u64 ta_in = 0, ta_out = 0, to_in = 0;
    // ---- phase 2: one wavefront per head
    if (p2) {
        const int h = wg;
        // q_h: granules 32 h .. + 32, k_h: 256 + 32 h .., v_h: 512 + 32 h ..: lane l < 48 takes two granules
        const int part = lane / 16, gidx = part * 256 + 32 * h + (lane % 16) * 2;
        unsigned spins = 0; bool ok;
        uint4 g;
        do {
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(gq + (lane < 48 ? gidx : 32 * h)) : "memory");
            ok = lane >= 48 || (g.y == tag && g.w == tag);
            if (++spins > 2000000u) { if (lane == 0) atomicExch(err, 1); break; }
        } while (!__all(ok));
        ta_in = wall();
        if (lane < 48) {
            const __half2 p0 = *(const __half2 *) &g.x, p1 = *(const __half2 *) &g.z;
            float * d = hq + part * 64 + (lane % 16) * 4;
            d[0] = __low2float(p0); d[1] = __high2float(p0); d[2] = __low2float(p1); d[3] = __high2float(p1);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const float o = attend(hq, kc, vcol, lane);
        const __half oh = __float2half(o);
        const unsigned mine = __half_as_ushort(oh), other = (unsigned) __shfl_down((int) mine, 1);
        if (!(lane & 1)) {
            const u64 gg = ((u64) tag << 32) | (u64) (mine | (other << 16));
            __hip_atomic_store(ga + h * 32 + (lane >> 1), gg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ta_out = wall();
    }
    if (!p3 && !phase3_all) { if (st && lane == 0 && wave == 0) st[wg] = Stamps{t0, tp1, ta_in, ta_out, 0, wall()}; return; }
    // ---- phase 3: the attention row, swept once per workgroup (threads 0..127: two granules each), then 2 rows per wavefront
    if (p3) {
        if (tid < 128) {
            unsigned spins = 0; bool ok; uint4 g;
            do {
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(ga + tid * 2) : "memory");
                ok = g.y == tag && g.w == tag;
                if (++spins > 2000000u) { if (lane == 0) atomicExch(err, 2); break; }
            } while (!__all(ok));
            const __half2 p0 = *(const __half2 *) &g.x, p1 = *(const __half2 *) &g.z;
            att[tid * 4] = __low2float(p0); att[tid * 4 + 1] = __high2float(p0); att[tid * 4 + 2] = __low2float(p1); att[tid * 4 + 3] = __high2float(p1);
        }
        __syncthreads();
        to_in = wall();
        float av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = att[lane * 8 + e];
        const float r0 = wave_sum(dot8(w2[0], av)), r1 = wave_sum(dot8(w2[1], av));
        if (lane == 0) { xout[orow] = r0 + res0; xout[orow + 1] = r1 + res1; }
    }
    if (st && lane == 0 && wave == 0) st[wg] = Stamps{t0, tp1, ta_in, ta_out, to_in, wall()};
}

int main(int argc, char ** argv) {
    const int NB = 96;
    srand(3);
    std::vector<__half> hW1((size_t) 3 * S * S), hW2((size_t) S * S), hk((size_t) NKV * S), hv((size_t) NKV * S);
    for (auto & h : hW1) h = __float2half(((rand() % 2001) - 1000) / 1000.0f * 0.06f);
    for (auto & h : hW2) h = __float2half(((rand() % 2001) - 1000) / 1000.0f * 0.03f);
    for (auto & h : hk) h = __float2half(((rand() % 2001) - 1000) / 1000.0f);
    for (auto & h : hv) h = __float2half(((rand() % 2001) - 1000) / 1000.0f);
    __half * W1, * W2, * kc, * vc, * qkv; float * xa, * xb; u64 * gq, * ga; Stamps * st; int * err;
    CK(hipMalloc(&W1, hW1.size() * 2)); CK(hipMalloc(&W2, hW2.size() * 2)); CK(hipMalloc(&kc, hk.size() * 2)); CK(hipMalloc(&vc, hv.size() * 2));
    CK(hipMalloc(&qkv, 3 * S * 2)); CK(hipMalloc(&xa, S * 4)); CK(hipMalloc(&xb, S * 4)); CK(hipMalloc(&gq, 768 * 8)); CK(hipMalloc(&ga, 256 * 8));
    CK(hipMalloc(&st, 128 * sizeof(Stamps))); CK(hipMalloc(&err, 4));
    CK(hipMemcpy(W1, hW1.data(), hW1.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W2, hW2.data(), hW2.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(kc, hk.data(), hk.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vc, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> x0(S); for (int i = 0; i < S; ++i) x0[i] = sinf(0.1f * i);
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref(S), got(S);

    // (A)
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemcpy(xa, x0.data(), S * 4, hipMemcpyHostToDevice));
        float * a = xa, * b = xb;
        CK(hipEventRecord(e0, s));
        for (int blk = 0; blk < NB; ++blk) {
            hipLaunchKernelGGL(k1_qkv, dim3(96), dim3(256), 0, s, a, W1, qkv);
            hipLaunchKernelGGL(k2_attn_out, dim3(32), dim3(512), 0, s, a, qkv, kc, vc, W2, b);
            std::swap(a, b);
        }
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(ref.data(), a, S * 4, hipMemcpyDeviceToHost));
        printf("(A) two launches per block              : %6.2f us per block   (x[0] = %g)\n", ms * 1000 / NB, ref[0]);
    }
    // (B)
    auto runB = [&](auto kern, int nwg, int grid, int stride, int p3all, const char * what) {
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemcpy(xa, x0.data(), S * 4, hipMemcpyHostToDevice));
            CK(hipMemset(gq, 0, 768 * 8)); CK(hipMemset(ga, 0, 256 * 8)); CK(hipMemset(err, 0, 4)); CK(hipMemset(st, 0, 128 * sizeof(Stamps)));
            float * a = xa, * b = xb;
            CK(hipEventRecord(e0, s));
            for (int blk = 0; blk < NB; ++blk) {
                hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, s, a, W1, kc, vc, W2, b, gq, ga, (unsigned) (blk + 1), stride, p3all,
                                   blk == NB - 1 ? st : (Stamps *) nullptr, err);
                std::swap(a, b);
            }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(got.data(), a, S * 4, hipMemcpyDeviceToHost));
            int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            double md = 0; for (int i = 0; i < S; ++i) md = fmax(md, fabs(got[i] - ref[i]));
            std::vector<Stamps> hs(128); CK(hipMemcpy(hs.data(), st, 128 * sizeof(Stamps), hipMemcpyDeviceToHost));
            u64 t0 = ~0ull, p1 = 0, ain = 0, aout = 0, oin = 0, t1 = 0;
            for (int w = 0; w < nwg; ++w) { if (!hs[w].t0) continue; t0 = std::min(t0, hs[w].t0); p1 = std::max(p1, hs[w].p1); ain = std::max(ain, hs[w].a_in); aout = std::max(aout, hs[w].a_out);
                                           oin = std::max(oin, hs[w].o_in); t1 = std::max(t1, hs[w].t1); }
            auto us = [&](u64 t) { return t ? (double) (t - t0) / 100.0 : -1.0; };
            printf("(B) %-36s: %6.2f us per block   max |diff| %.3g err %d | last block: q|k|v published +%.2f, heads have q k v +%.2f, attention published +%.2f, "
                   "row gathered +%.2f, end +%.2f\n", what, ms * 1000 / NB, md, herr, us(p1), us(ain), us(aout), us(oin), us(t1));
        }
    };
    runB(k_front_narrow<4>, 48, 48 * 8, 8, 0, "one launch, 48 WGs on ONE XCD");
    runB(k_front_narrow<4>, 48, 48 * 4, 4, 0, "one launch, 48 WGs on TWO XCDs");
    runB(k_front_narrow<4>, 48, 48, 1, 0, "one launch, 48 WGs spread");
    runB(k_front_narrow<2>, 96, 96 * 4, 4, 0, "one launch, 96 WGs on TWO XCDs");
    runB(k_front_narrow<2>, 96, 96 * 2, 2, 0, "one launch, 96 WGs on FOUR XCDs");
    runB(k_front_narrow<2>, 96, 96, 1, 0, "one launch, 96 WGs spread");
    return 0;
}
