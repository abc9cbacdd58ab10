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

#include <atomic>
#include "kernels.h"
#include "wave_ops.h"
#include <type_traits>

#include <cstdlib>
#include <algorithm>

namespace wmi { namespace k {

namespace {

__device__ __forceinline__ float round_f16(float x) { return __half2float(f2h(x)); }
// floats of the score region behind the f16 rows in the LDS of the projection kernels with a self-attention prologue: [H][cap] for the
// long-cache routine, and at least the wavefront-private scratch of self_attn_wave (4 wavefronts x 64 x 5 floats)
__host__ __device__ __forceinline__ size_t sa_score_floats(int K, int cap) { const size_t n = (size_t) (K / 64) * cap; return n < 1280 ? 1280 : n; }
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

__global__ void k_step_mirror(const int32_t * __restrict__ host_step, int32_t * __restrict__ dev_step) {
    if (threadIdx.x < sizeof(DecStep) / 4) dev_step[threadIdx.x] = ((const volatile int32_t *) host_step)[threadIdx.x];
}

// Graph-replay variant: the step parameters are read straight from pinned HOST memory (one PCIe read, no memcpy
// node in the graph) and mirrored into device memory for the kernels that follow.
__global__ void k_dec_embed_step(const DecStep * __restrict__ host_step, DecStep * __restrict__ dev_step, int S,
                                 const __half * __restrict__ te, const float * __restrict__ pe, float * __restrict__ x, const Stamp sp) {
    __shared__ DecStep st;
    const unsigned long long ts0 = stamp_t0(sp.base);
    host_step += blockIdx.x; dev_step += blockIdx.x; x += (size_t) blockIdx.x * S;      // one workgroup per lock-step chunk
    if (threadIdx.x < sizeof(DecStep) / 4) ((int32_t *) &st)[threadIdx.x] = ((const volatile int32_t *) host_step)[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < sizeof(DecStep) / 4) ((int32_t *) dev_step)[threadIdx.x] = ((const int32_t *) &st)[threadIdx.x];
    const __half * t = te + (size_t) st.token * S;
    const float *  p = pe + (size_t) st.pos * S;
    for (int c = threadIdx.x; c < S; c += blockDim.x) x[c] = __half2float(t[c]) + p[c];
    stamp_end(sp.base, sp.slot, blockIdx.x * 4 + (threadIdx.x >> 6), ts0);
}


// Single-token self-attention of one row over its KV cache, all heads, by one 256-thread workgroup.
// Numerics as k_attn_dec: scores f16.f16 -> f32, exp through f16, probabilities rounded to f16 before P.V, P.V
// accumulated in key order (SURVEY App. B rules 1, 4, 5).  sc: [H][cap] floats, qf: [K] floats (LDS); out: [K] f16
// (LDS or global).  kpre: this thread's first K row if it was requested early (use_pre).
// Self-attention of one decoder row for n_kv <= 64: NU heads (hs[u] < H; others skipped) by ONE wavefront, no LDS and no
// workgroup barrier.  The barrier-separated routine below measured 7 us inside the out-projection (q, K, soft-max and V
// phases each waiting on the slowest wavefront, and a 128-byte K row per lane); here every global load that does not
// depend on n_kv goes out before n_kv is read, in shapes the L1 likes:
//   scores   lane = (key group g = lane / 8, dimension octet o = lane % 8); pass t covers keys 8 t + g: one 16-byte load per
//            lane and pass, 8 fmaf in dimension order, then the three-step butterfly over the octets of the key
//   soft-max maximum and sum over passes (in pass order) and over the key groups (xor 8, 16, 32); f16 roundings of the reference
//   P.V      same layout: lane (g, o) accumulates the 8 columns of its octet over its keys in pass order, then the butterfly
//            over the key groups; the V rows of keys [0, 32) are requested up front like the K rows
// This fixes the summation order of the path (octet-wise, then butterflies); every caller — the one-row prologue of
// k_gemv1, the lock-step row kernels — goes through this routine, which keeps them bit-identical to each other.
// Returns false, with nothing written, when n_kv > 64 (callers then take self_attn_row).
// weight of a cross-attention slice partial in the combine: 1 when the partials are relative to the row's global maximum
// (comb_m == null), else exp(m_slice - M) (k_xattn_fused, k_attn.hip).  v_exp_f32 (__expf, ~1e-6 relative on arguments of a few
// units) in EVERY consumer of the partials — k_gemv1, k_gemv, k_rows_mfma, k_qrows, k_xattn_combine — so that they stay
// bit-identical to each other; libm's expf was 16 calls per thread of the out projection's prologue (+0.6 us per launch)
__device__ __forceinline__ float comb_weight(const float * comb_m, size_t idx, float M) {
    if (!comb_m) return 1.0f;
    const float ms = comb_m[idx];
    return ms > -INFINITY ? __expf(ms - M) : 0.0f;
}

// after_loads(): called once every load of the routine is issued (the caller's colder loads go there: vmcnt retires in order, so
// whatever is requested BEFORE the keys delays the attention by its own latency — the out projection's weight rows come from
// HBM / Infinity Cache, q / K / V of a step from L2).
//
// Instruction count is what this routine costs (one wavefront per SIMD, every VALU instruction ~4 cycles on the critical path of a
// launch whose operands are all in L2): round 2's form kept every (head, pass) score on all 8 octet lanes of a key — 8 butterflies of
// 3 steps, 8 libm expf and 8 x 3 x 3 exchange steps for P.V per head pair, ~1000 instructions, 3.5 us.  Here the octet butterflies
// and the key-group butterflies are HALVING exchanges (at each step a lane keeps half of its values and receives the partner's
// partial for those — the same pairs in the same order, bit-identical sums, 7 exchanges instead of 24), after which a lane owns ONE
// (head, pass) score of its key group: one exponential per lane instead of eight, the soft-max statistics as masked 64-lane
// reductions, and the probabilities go back to the (key group, octet) layout of P.V through 256 R bytes of wavefront-private LDS
// (no barrier: one wavefront).  P.V's reduction over the key groups is a halving exchange too (v_permlane16/32_swap: swap + add),
// leaving each lane one output column.
struct NoAfterLoads { __device__ __forceinline__ void operator()() const {} };
#if defined(WMI_SA_STAMPS)
__shared__ unsigned long long wmi_dbg_t[8];
#endif

// halving exchange over lane bit M of n values (n even): v[j] <- (bit ? v[j + n/2] : v[j]) + partner's partial of the same
template <int M, int N>
__device__ __forceinline__ void halve_sum(float (&v)[N], int n, int lane) {
    const bool hi = lane & M;
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
        if (j < n / 2) {
            const float a = v[j], b = v[j + n / 2];
            if constexpr (M == 16) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                v[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            } else if constexpr (M == 32) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                v[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            } else {
                const float keep = hi ? b : a, send = hi ? a : b;
                v[j] = keep + xor_lane<M>(send);
            }
        }
    }
}

template <int NU, int NP>
__device__ __forceinline__ void self_attn_body(const uint4 (&qv)[NU], const uint4 (&kv)[NU][8], const uint4 (&vv)[NU][8], int n_kv,
                                               const int (&hs)[NU], int H, int lane, __half * out, float * out32, float * wscr) {
    constexpr int NV = NU * NP, NV8 = (NV + 7) & ~7, R = NV8 / 8;
    const int g = lane >> 3, o = lane & 7;
    // ---- scores: value i = u * NP + t is q[u] . K[u][8 t + g] over this lane's octet
    float d[NV8];
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
        d[i] = 0.0f;
        if (i < NV) {
            const int u = i / NP, t = i % NP;
            const __half2 * qh = (const __half2 *) &qv[u];
            const __half2 * kh = (const __half2 *) &kv[u][t];
            float dot = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 q2 = __half22float2(qh[e]), k2 = __half22float2(kh[e]);
                dot = fmaf(k2.x, q2.x, dot);
                dot = fmaf(k2.y, q2.y, dot);
            }
            d[i] = dot;
        }
    }
    halve_sum<1, NV8>(d, NV8, lane); halve_sum<2, NV8>(d, NV8 / 2, lane); halve_sum<4, NV8>(d, NV8 / 4, lane);
#if defined(WMI_SA_STAMPS)
    if (threadIdx.x == 0) wmi_dbg_t[WMI_SA_STAMPS] = wall_clock64();
#endif
    // slot r of this lane now holds value i = r + R * code(o) for key group g, summed over the 8 octets in the order 1, 2, 4
    const int code = ((o >> 2) & 1) + 2 * ((o >> 1) & 1) + 4 * (o & 1);
    int ui[R]; bool ok[R]; float sc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = r + R * code, t = i % NP;
        ui[r] = i / NP;
        ok[r] = i < NV && 8 * t + g < n_kv;
        sc[r] = ok[r] ? d[r] : -INFINITY;
    }
    // ---- soft-max per head: maximum, e = f16(exp(f16(s - m))), l = sum e, P = f16(e / l)   (SURVEY App. B rules 4, 5)
    // A head's values sit on the lanes whose high code bits equal its number whenever R divides NP: then ONE butterfly that skips
    // those octet bits reduces every head at once.  The skipped steps of the general (masked, per head) form only ever add 0 /
    // compare with -inf, so both forms give the same bits — callers with different NU stay bit-identical to each other.
    float mr[R], lr[R], er[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { mr[r] = 0.0f; lr[r] = 1.0f; }
    constexpr bool FAST = (NP % R) == 0 && NV8 * NP >= 8 * R;      // q = NP / R values of code per head, q in {1, 2, 4, 8}
    constexpr int QH = FAST ? NP / R : 8;
    auto reduce_heads = [&](float x, auto op) {
        // lanes of one head differ in g (xor 8, 16, 32) and in the code bits below QH: code bit weight 1 <-> xor 4, 2 <-> xor 2, 4 <-> xor 1
        x = op(x, xor_lane<32>(x)); x = op(x, xor_lane<16>(x)); x = op(x, xor_lane<8>(x));
        if constexpr (QH > 1) x = op(x, xor_lane<4>(x));
        if constexpr (QH > 2) x = op(x, xor_lane<2>(x));
        if constexpr (QH > 4) x = op(x, xor_lane<1>(x));
        return x;
    };
    if constexpr (FAST) {
        float x = -INFINITY;
#pragma unroll
        for (int r = 0; r < R; ++r) x = fmaxf(x, sc[r]);
        x = reduce_heads(x, [](float a, float b) { return fmaxf(a, b); });
#pragma unroll
        for (int r = 0; r < R; ++r) mr[r] = x;
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float x = -INFINITY;
#pragma unroll
            for (int r = 0; r < R; ++r) x = fmaxf(x, ui[r] == u ? sc[r] : -INFINITY);
            x = wave_max_desc(x);
#pragma unroll
            for (int r = 0; r < R; ++r) if (ui[r] == u) mr[r] = x;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) er[r] = ok[r] ? round_f16(expf(round_f16(sc[r] - mr[r]))) : 0.0f;
    if constexpr (FAST) {
        float x = 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) x += er[r];
        x = reduce_heads(x, [](float a, float b) { return a + b; });
#pragma unroll
        for (int r = 0; r < R; ++r) lr[r] = x;
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float x = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) x += ui[r] == u ? er[r] : 0.0f;
            x = wave_sum_desc(x);
#pragma unroll
            for (int r = 0; r < R; ++r) if (ui[r] == u) lr[r] = x;
        }
    }
#if defined(WMI_SA_STAMPS)
    if (threadIdx.x == 0) wmi_dbg_t[WMI_SA_STAMPS + 1] = wall_clock64();
#endif
    // ---- probabilities back to the (key group, octet) layout: wscr[g][i], every lane of group g reads the group's NV8 values
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float pr = ok[r] ? round_f16(er[r] * (float) (1.0 / (double) lr[r])) : 0.0f;
        wscr[g * NV8 + r + R * code] = pr;
    }
    float pv[NV8];
#pragma unroll
    for (int i4 = 0; i4 < NV8; i4 += 4) {
        const float4 f = *(const float4 *) (wscr + g * NV8 + i4);
        pv[i4] = f.x; pv[i4 + 1] = f.y; pv[i4 + 2] = f.z; pv[i4 + 3] = f.w;
    }
    // ---- P.V: lane (g, o) accumulates the 8 columns of its octet over its keys 8 t + g in pass order, the key groups are summed by
    // halving exchanges in the order 8, 16, 32 (the old butterflies' pairs); the lane ends with column e = 4 b3 + 2 b4 + b5
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            if (8 * t + g < n_kv) {                          // (rows past n_kv hold whatever the cache held: never multiplied)
                const __half2 * vh = (const __half2 *) &vv[u][t];
                const float w = pv[u * NP + t];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 f = __half22float2(vh[e]);
                    acc[2 * e]     = fmaf(w, f.x, acc[2 * e]);
                    acc[2 * e + 1] = fmaf(w, f.y, acc[2 * e + 1]);
                }
            }
        }
        halve_sum<8, 8>(acc, 8, lane); halve_sum<16, 8>(acc, 4, lane); halve_sum<32, 8>(acc, 2, lane);
        const int col = o * 8 + 4 * ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1);
        if (hs[u] < H) {
            if (out32) out32[hs[u] * 64 + col] = acc[0];      // block-quantised out-projection: the f32 result is quantised as is
            else       out[hs[u] * 64 + col] = f2h(acc[0]);
        }
    }
}

// wscr: 64 * ceil(NU * NP / 8) floats of LDS private to this wavefront (NP = 8 passes when n_kv > 32)
template <int NU, typename AfterLoads = NoAfterLoads>
__device__ __forceinline__ bool self_attn_wave(const __half * __restrict__ sq, const __half * __restrict__ sk,
                                               const __half * __restrict__ sv, const int32_t * __restrict__ n_kv_p, int K, int cap,
                                               const int (&hs)[NU], int H, int lane, __half * out, float * out32, float * wscr,
                                               AfterLoads after_loads = AfterLoads()) {
    const int g = lane >> 3, o = lane & 7;
    int hh[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) hh[u] = hs[u] < H ? hs[u] : hs[0];
    uint4 qv[NU], kv[NU][8], vv[NU][8];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        qv[u] = *(const uint4 *) (sq + hh[u] * 64 + o * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) {                    // keys [0, 32): the decode loop rarely holds more (rows past n_kv: finite cache garbage)
            const int j = 8 * t + g;
            kv[u][t] = *(const uint4 *) (sk + (size_t) (j < cap ? j : 0) * K + hh[u] * 64 + o * 8);
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = 8 * t + g;
            vv[u][t] = *(const uint4 *) (sv + (size_t) (j < cap ? j : 0) * K + hh[u] * 64 + o * 8);
        }
    // n_kv through a lane offset the compiler cannot fold: as a uniform load it became load -> vmcnt(0) -> readfirstlane on the spot,
    // i.e. a wait for everything issued before it; here the wait sits at the first use and counts only the loads above
    int zl = 0; asm volatile("" : "+v"(zl));
    const int n_kv_v = n_kv_p[zl];
    __builtin_amdgcn_sched_barrier(0);
    after_loads();
    __builtin_amdgcn_sched_barrier(0);
    const int n_kv = __builtin_amdgcn_readfirstlane(n_kv_v);
    if (n_kv > 64) return false;
    if (n_kv > 32) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int t = 4; t < 8; ++t) {
                const int j = 8 * t + g, jc = j < n_kv ? j : 0;
                kv[u][t] = *(const uint4 *) (sk + (size_t) jc * K + hh[u] * 64 + o * 8);
                vv[u][t] = *(const uint4 *) (sv + (size_t) jc * K + hh[u] * 64 + o * 8);
            }
        self_attn_body<NU, 8>(qv, kv, vv, n_kv, hs, H, lane, out, out32, wscr);
    } else {
        self_attn_body<NU, 4>(qv, kv, vv, n_kv, hs, H, lane, out, out32, wscr);
    }
    return true;
}

__device__ __forceinline__ void self_attn_row(const __half * __restrict__ sq, const __half * __restrict__ sk,
                                              const __half * __restrict__ sv, int n_kv, int K, int cap,
                                              float * sc, float * qf, __half * out, const uint4 (&kpre)[8], bool use_pre) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NTH = (int) blockDim.x;      // (256 or 512 threads)
    const int H = K / 64;
    {   // q -> f32 in LDS.  All loads of a thread first: as a plain strided loop hipcc emitted one load + vmcnt(0) per
        // element for short trip counts (K = 512: two dependent round trips before the first score)
        constexpr int QU = 5;                           // K <= 1280
        __half qv[QU];
#pragma unroll
        for (int u = 0; u < QU; ++u) { const int c = tid + u * NTH; qv[u] = sq[c < K ? c : 0]; }
#pragma unroll
        for (int u = 0; u < QU; ++u) { const int c = tid + u * NTH; if (c < K) qf[c] = __half2float(qv[u]); }
    }
    __syncthreads();
    for (int p = tid; p < H * n_kv; p += NTH) {
        const int j = p / H, h = p - j * H;
        const uint4 * kp = (const uint4 *) (sk + (size_t) j * K + h * 64);
        float dot = 0.0f;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const uint4 u = (p == tid && use_pre) ? kpre[c8] : kp[c8];
            const __half2 * hh = (const __half2 *) &u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(hh[e]);
                dot = fmaf(f.x, qf[h * 64 + c8 * 8 + e * 2], dot);
                dot = fmaf(f.y, qf[h * 64 + c8 * 8 + e * 2 + 1], dot);
            }
        }
        sc[(size_t) h * cap + j] = dot;
    }
    __syncthreads();
    for (int h = wave; h < H; h += (NTH >> 6)) {                 // soft-max of one head per wavefront
        float * row = sc + (size_t) h * cap;
        float m = -INFINITY;
        for (int j = lane; j < n_kv; j += 64) m = fmaxf(m, row[j]);
        _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, WMI_SHX(m, o));
        float l = 0.0f;
        for (int j = lane; j < n_kv; j += 64) { const float e = round_f16(expf(round_f16(row[j] - m))); row[j] = e; l += e; }
        _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) l += WMI_SHX(l, o);
        const float inv = (float) (1.0 / (double) l);
        for (int j = lane; j < n_kv; j += 64) row[j] = round_f16(row[j] * inv);
    }
    __syncthreads();
    for (int c = tid; c < K; c += NTH) {
        const float * row = sc + (size_t) (c >> 6) * cap;
        const __half * vp = sv + c;
        float acc = 0.0f;
        for (int j = 0; j < n_kv; j += 8) {         // loads issued 8 at a time (the last group predicated: a scalar tail
            __half vv[8];                           // loop would be up to 7 dependent round trips), accumulated in key order
#pragma unroll
            for (int t = 0; t < 8; ++t) vv[t] = vp[(size_t) (j + t < n_kv ? j + t : n_kv - 1) * K];
#pragma unroll
            for (int t = 0; t < 8; ++t) if (j + t < n_kv) acc = fmaf(row[j + t], __half2float(vv[t]), acc);
        }
        out[c] = f2h(acc);
    }
}

// lock-step chunks: one wavefront per (row, head), each row against its own chunk's cache; out [R][K] f16 (global).
// Per (key, head) dot product, per-head soft-max and per-column P.V are evaluated exactly as in self_attn_row (same
// operand order), so the result is bit-identical to the one-row fused prologue; 8 x H workgroups instead of 8.
__global__ __launch_bounds__(64) void k_self_attn_rows(const __half * __restrict__ q, const __half * __restrict__ kc,
                                                       const __half * __restrict__ vc, int64_t cache_row_stride,
                                                       const int32_t * __restrict__ n_kv_p, int step_stride, int K, int cap,
                                                       __half * __restrict__ out, float * __restrict__ out32,
                                                       const int32_t * __restrict__ mirror_src, int32_t * __restrict__ mirror_dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float * row = (float *) smem;                       // [cap] scores -> probabilities of this head
    float * qf  = row + cap;                            // [64]
    const int r = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    if (h == K / 64) {
        // chained lock-step steps (batch.cpp): one extra workgroup per row mirrors the host's step record — filter flags, sequence
        // number, temperature: words 4.. of DecStep — into the device-side record; token / position / cache head (words 0..3) are
        // what the previous step's pick kernel left there.  The PCIe read rides beside the attention instead of in front of the step.
        constexpr int W = (int) (sizeof(DecStep) / 4);
        if (lane >= 4 && lane < W) mirror_dst[r * W + lane] = ((const volatile int32_t *) mirror_src)[r * W + lane];
        return;
    }
    const __half * sk = kc + (int64_t) r * cache_row_stride, * sv = vc + (int64_t) r * cache_row_stride;
    {   // n_kv <= 64 (the decode loop): the wave-level routine shared with the one-row prologue
        const int hs[1] = { h };
        if (self_attn_wave<1>(q + (size_t) r * K, sk, sv, n_kv_p + r * step_stride, K, cap, hs, K / 64, lane, out + (size_t) r * K,
                              out32 ? out32 + (size_t) r * K : nullptr, row)) return;       // (row: >= 64 floats, free until the long form)
    }
    const int n_kv = n_kv_p[r * step_stride];
    qf[lane] = __half2float(q[(size_t) r * K + h * 64 + lane]);
    __syncthreads();
    for (int j = lane; j < n_kv; j += 64) {
        const uint4 * kp = (const uint4 *) (sk + (size_t) j * K + h * 64);
        uint4 u[8];
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) u[c8] = kp[c8];
        float dot = 0.0f;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const __half2 * hh = (const __half2 *) &u[c8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(hh[e]);
                dot = fmaf(f.x, qf[c8 * 8 + e * 2], dot);
                dot = fmaf(f.y, qf[c8 * 8 + e * 2 + 1], dot);
            }
        }
        row[j] = dot;
    }
    __syncthreads();
    {
        float m = -INFINITY;
        for (int j = lane; j < n_kv; j += 64) m = fmaxf(m, row[j]);
        _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, WMI_SHX(m, o));
        float l = 0.0f;
        for (int j = lane; j < n_kv; j += 64) { const float e = round_f16(expf(round_f16(row[j] - m))); row[j] = e; l += e; }
        _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) l += WMI_SHX(l, o);
        const float inv = (float) (1.0 / (double) l);
        for (int j = lane; j < n_kv; j += 64) row[j] = round_f16(row[j] * inv);
    }
    __syncthreads();
    {
        const int c = h * 64 + lane;
        const __half * vp = sv + c;
        float acc = 0.0f;
        for (int j = 0; j < n_kv; j += 8) {
            __half vv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) vv[t] = vp[(size_t) (j + t < n_kv ? j + t : n_kv - 1) * K];
#pragma unroll
            for (int t = 0; t < 8; ++t) if (j + t < n_kv) acc = fmaf(row[j + t], __half2float(vv[t]), acc);
        }
        if (out32) out32[(size_t) r * K + c] = acc; else
        out[(size_t) r * K + c] = f2h(acc);
    }
}

// Long caches (the host knows that some row has more than 64 cells: uncapped transcriptions, context prompts): four wavefronts
// per (row, head).  Scores one key per thread, exact maximum, e = exp16(s - m), l in f32, P = f16(e / l) — the steps of
// k_self_attn_rows' barrier form — and P.V with the keys cut into four contiguous quarters, one per wavefront (lane = column,
// keys in ascending order inside a quarter), the quarters added in order.  Rows of <= 64 cells take the same arithmetic as
// any other here (they do not occur in the callers that pick this kernel more than transiently).
__global__ __launch_bounds__(256) void k_self_attn_rows_long(const __half * __restrict__ q, const __half * __restrict__ kc,
                                                            const __half * __restrict__ vc, int64_t cache_row_stride,
                                                            const int32_t * __restrict__ n_kv_p, int step_stride, int K, int cap,
                                                            __half * __restrict__ out, float * __restrict__ out32) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float * row = (float *) smem;                       // [cap] scores -> probabilities of this head
    float * qf  = row + cap;                            // [64]
    float * part = qf + 64;                             // [4][64] partial outputs
    float * red = part + 256;                           // [4]
    const int r = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __half * sk = kc + (int64_t) r * cache_row_stride, * sv = vc + (int64_t) r * cache_row_stride;
    const int n_kv = n_kv_p[r * step_stride];
    if (tid < 64) qf[tid] = __half2float(q[(size_t) r * K + h * 64 + tid]);
    __syncthreads();
    float m = -INFINITY;
    for (int j = tid; j < n_kv; j += 256) {
        const uint4 * kp = (const uint4 *) (sk + (size_t) j * K + h * 64);
        uint4 u[8];
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) u[c8] = kp[c8];
        float dot = 0.0f;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const __half2 * hh = (const __half2 *) &u[c8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(hh[e]);
                dot = fmaf(f.x, qf[c8 * 8 + e * 2], dot);
                dot = fmaf(f.y, qf[c8 * 8 + e * 2 + 1], dot);
            }
        }
        row[j] = dot;
        m = fmaxf(m, dot);
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, WMI_SHX(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();                                    // red is reused for the sums
    float l = 0.0f;
    for (int j = tid; j < n_kv; j += 256) { const float e = round_f16(expf(round_f16(row[j] - m))); row[j] = e; l += e; }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) l += WMI_SHX(l, o);
    if (lane == 0) red[wave] = l;
    __syncthreads();
    const float inv = (float) (1.0 / (double) ((red[0] + red[1]) + (red[2] + red[3])));
    for (int j = tid; j < n_kv; j += 256) row[j] = round_f16(row[j] * inv);
    __syncthreads();
    {
        const int per = (n_kv + 3) >> 2, j0 = wave * per, j1 = min(n_kv, j0 + per);
        const __half * vp = sv + h * 64 + lane;
        float acc = 0.0f;
        for (int j = j0; j < j1; j += 8) {
            __half vv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) vv[t] = vp[(size_t) (j + t < j1 ? j + t : j1 - 1) * K];
#pragma unroll
            for (int t = 0; t < 8; ++t) if (j + t < j1) acc = fmaf(row[j + t], __half2float(vv[t]), acc);
        }
        part[wave * 64 + lane] = acc;
    }
    __syncthreads();
    if (tid < 64) {
        const float acc = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
        const int c = h * 64 + tid;
        if (out32) out32[(size_t) r * K + c] = acc; else out[(size_t) r * K + c] = f2h(acc);
    }
}

// LayerNorm of one row by one wavefront, lane L holding the slices x[512 t + 8 L .. + 8) (the slices its dot products
// need).  y = f16((x - mean) * rstd * g + b) as separate mul / add (SURVEY App. B rule 6); sums in f32: per lane over
// (t, e) in order, then the 64-lane butterfly.  MAXCH chunks of 512 columns; av[t][e] = 0 outside the row.
// the loads of ln_row_regs: MAXCH chunks of one f32 vector (x, gain or bias), zeros outside the row
template <int MAXCH>
__device__ __forceinline__ void ln_row_load(const float * __restrict__ p, int K, int lane, float (&v)[MAXCH][8]) {
#pragma unroll
    for (int t = 0; t < MAXCH; ++t) {
        // unconditional from a clamped column, masked afterwards: a load inside `if (c < K)` is its own basic block, and the
        // s_waitcnt pass cannot count loads it is not sure were issued — every wait after such a block became vmcnt(0 or 1),
        // i.e. "everything", instead of "the row" 
        // (the mask is ln_row_mask, called where the values are first needed: a select here is a use, i.e. a wait in front of
        // whatever the caller requests next)
        const int c = lane * 8 + 512 * t, cc = c < K ? c : 0;
        const float4 x0 = *(const float4 *) (p + cc), x1 = *(const float4 *) (p + cc + 4);
        v[t][0] = x0.x; v[t][1] = x0.y; v[t][2] = x0.z; v[t][3] = x0.w; v[t][4] = x1.x; v[t][5] = x1.y; v[t][6] = x1.z; v[t][7] = x1.w;
    }
}
template <int MAXCH>
__device__ __forceinline__ void ln_row_mask(float (&v)[MAXCH][8], int K, int lane) {
#pragma unroll
    for (int t = 0; t < MAXCH; ++t) {
        const bool on = lane * 8 + 512 * t < K;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[t][e] = on ? v[t][e] : 0.0f;
    }
}
// the arithmetic of ln_row_regs on loaded values (xv is consumed)
template <int MAXCH>
__device__ __forceinline__ void ln_row_compute(float (&xv)[MAXCH][8], const float (&gv)[MAXCH][8], const float (&bv)[MAXCH][8],
                                               int K, float eps, int lane, float (&av)[MAXCH][8]) {
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < MAXCH; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += xv[t][e];
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sum += WMI_SHX(sum, o);
    const float mean = sum / (float) K;
    float sq = 0.0f;
#pragma unroll
    for (int t = 0; t < MAXCH; ++t) {
        const bool on = lane * 8 + 512 * t < K;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (on) { xv[t][e] -= mean; sq += xv[t][e] * xv[t][e]; }
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sq += WMI_SHX(sq, o);
    const float sc = 1.0f / sqrtf(sq / (float) K + eps);
#pragma unroll
    for (int t = 0; t < MAXCH; ++t) {
        const bool on = lane * 8 + 512 * t < K;
#pragma unroll
        for (int e = 0; e < 8; ++e) av[t][e] = on ? round_f16(__fadd_rn(__fmul_rn(xv[t][e] * sc, gv[t][e]), bv[t][e])) : 0.0f;
    }
}
// ln_row_compute for RW rows at once, in place (xv becomes the normalised row): every stage runs across the rows, so the
// two 6-step butterflies of a row overlap with the other rows' instead of forming RW dependent chains (lock-step q|k|v at
// 16 rows: 5.3 -> ~2.5 us of LayerNorm per launch).  Per row the operations and their order are ln_row_compute's.
template <int RW, int MAXCH>
__device__ __forceinline__ void ln_rows_compute(float (&xv)[RW][MAXCH][8], const float (&gv)[MAXCH][8], const float (&bv)[MAXCH][8],
                                                int K, float eps, int lane) {
    float sum[RW], sq[RW], sc[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        sum[q] = 0.0f;
#pragma unroll
        for (int t = 0; t < MAXCH; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[q] += xv[q][t][e];
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < RW; ++q) sum[q] += WMI_SHX(sum[q], o);
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        const float mean = sum[q] / (float) K;
        sq[q] = 0.0f;
#pragma unroll
        for (int t = 0; t < MAXCH; ++t) {
            const bool on = lane * 8 + 512 * t < K;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (on) { xv[q][t][e] -= mean; sq[q] += xv[q][t][e] * xv[q][t][e]; }
        }
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < RW; ++q) sq[q] += WMI_SHX(sq[q], o);
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        sc[q] = 1.0f / sqrtf(sq[q] / (float) K + eps);
#pragma unroll
        for (int t = 0; t < MAXCH; ++t) {
            const bool on = lane * 8 + 512 * t < K;
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[q][t][e] = on ? round_f16(__fadd_rn(__fmul_rn(xv[q][t][e] * sc[q], gv[t][e]), bv[t][e])) : 0.0f;
        }
    }
}
template <int MAXCH>
__device__ __forceinline__ void ln_row_regs(const float * __restrict__ xr, const float * __restrict__ g, const float * __restrict__ b,
                                            int K, float eps, int lane, float (&av)[MAXCH][8]) {
    float xv[MAXCH][8], gv[MAXCH][8], bv[MAXCH][8];
    ln_row_load<MAXCH>(xr, K, lane, xv);
    ln_row_load<MAXCH>(g, K, lane, gv);
    ln_row_load<MAXCH>(b, K, lane, bv);
    ln_row_mask<MAXCH>(xv, K, lane); ln_row_mask<MAXCH>(gv, K, lane); ln_row_mask<MAXCH>(bv, K, lane);
    ln_row_compute<MAXCH>(xv, gv, bv, K, eps, lane, av);
}

template <bool NT> __device__ __forceinline__ uint4 ldw(const __half * p) {
    if (NT) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_nontemporal_load((const u4 *) p);
        return make_uint4(v[0], v[1], v[2], v[3]);
    }
    return *(const uint4 *) p;
}

template <int R, int ROWS_IN_FLIGHT, bool NT = false>
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
            wpre[u] = ldw<NT>(a.W + (size_t) o * K + lane * 8);
        }
    }

    // epilogue operands of the first tile (bias, residual, KV-cache head): independent of everything, fetched now so
    // that the epilogue does not add a dependent memory round trip (~1.5 us each on freshly written lines)
    float bias_pre = 0.0f, resid_pre = 0.0f; int ro_pre = 0;
    {
        const int u = lane / R, r = lane - u * R, n = gw * ROWS_IN_FLIGHT + u;
        if (lane < ROWS_IN_FLIGHT * R && n < a.N) {
            if (a.bias) bias_pre = a.bias[n];
            if (a.resid) resid_pre = a.resid[(size_t) r * a.ldr + n];
        }
        if (a.row_off) ro_pre = a.lanes ? a.row_off[(lane < ROWS_IN_FLIGHT * R ? r : 0) * a.step_stride] : *a.row_off;
    }

    // ---- prologue: stage the activation rows as f16
    if (a.ln_g) {                                           // fused LayerNorm of the f32 residual stream (K <= 1536)
        for (int r = wave; r < R; r += 4) {
            const int src = a.rows ? a.rows[r] : r;
            float av[3][8];
            ln_row_regs<3>(a.x32 + (size_t) src * K, a.ln_g, a.ln_b, K, a.eps, lane, av);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int c = lane * 8 + 512 * t;
                if (c < K) {
                    __half2 h[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = __floats2half2_rn(av[t][2 * e], av[t][2 * e + 1]);
                    *(uint4 *) (act + r * K + c) = *(const uint4 *) h;
                }
            }
        }
    } else if (a.sa_q) {                                    // fused single-token self-attention over the KV cache
        // Every workgroup recomputes the (tiny) attention of all heads: n_kv x H dot products of 64 — cheaper than
        // a separate launch on the critical path of a decode step at one row.  (Lock-step chunks use the separate
        // k_self_attn_rows launch below: R rows recomputed by every workgroup would cost more than the launch.)
        const int H = K / 64;
        float * sc = (float *) (smem + (((size_t) R * K * sizeof(__half) + 15) & ~(size_t) 15));   // [H][sa_cap]
        float * qf = sc + sa_score_floats(K, a.sa_cap);                                            // [K]
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            const __half * rq = a.sa_q + (size_t) r * K, * rk = a.sa_k + (int64_t) r * a.cache_row_stride, * rv = a.sa_v + (int64_t) r * a.cache_row_stride;
            const int32_t * rn = a.sa_nkv + r * a.step_stride;
            bool done = true;                               // n_kv <= 64: the wave-level routine of the one-row kernel (bit-identical)
            for (int h0 = tid >> 6; h0 < H; h0 += 8) {
                const int hs[2] = { h0, h0 + 4 };
                done = self_attn_wave<2>(rq, rk, rv, rn, K, a.sa_cap, hs, H, tid & 63, act + (size_t) r * K, nullptr, sc + (tid >> 6) * 128) && done;
            }
            if ((tid >> 6) >= H) done = rn[0] <= 64;
            if (!done) {
                const uint4 kpre[8] = {};
                self_attn_row(rq, rk, rv, rn[0], K, a.sa_cap, sc, qf, act + (size_t) r * K, kpre, false);
                if (R > 1) __syncthreads();                     // sc / qf are reused by the next row
            }
        }
    } else if (a.comb_o) {                                  // fused combine of the split cross-attention partials
        const int H = K / 64, ns = a.comb_ns;
        for (int e = tid; e < R * K; e += 256) {
            const int r = e / K, c = e - r * K, h = c >> 6, dd = c & 63;
            const size_t row = (size_t) r * H + h;
            float o = 0.0f; double l = 0.0, M = -INFINITY;
            if (a.comb_m) for (int s2 = 0; s2 < ns; ++s2) M = fmax(M, (double) a.comb_m[row * ns + s2]);
            for (int s2 = 0; s2 < ns; ++s2) {
                const float w = comb_weight(a.comb_m, row * ns + s2, (float) M);
                o += a.comb_o[(row * ns + s2) * 64 + dd] * w; l += (double) a.comb_l[row * ns + s2] * (double) w;
            }
            act[e] = f2h(o * (float) (1.0 / l));
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
            if (first && c == lane * 8) {
#pragma unroll
                for (int u = 0; u < ROWS_IN_FLIGHT; ++u) w[u] = wpre[u];
                first = false;
                // software pipeline over the row tiles: the first chunk of the NEXT tile is requested before this
                // tile is multiplied and reduced (the logits matrix is 51 864 rows: ~6 tiles per wavefront)
                const int on = o0 + nwaves * ROWS_IN_FLIGHT;
                if (on < a.N && lane * 8 < K) {
#pragma unroll
                    for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
                        int o = on + u; if (o > a.N - 1) o = a.N - 1;
                        wpre[u] = ldw<NT>(a.W + (size_t) o * K + lane * 8);
                    }
                    first = true;                       // consumed by the next o0 iteration's first chunk
                }
            } else {
#pragma unroll
                for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
                    int o = o0 + u; if (o > a.N - 1) o = a.N - 1;
                    w[u] = ldw<NT>(a.W + (size_t) o * K + c);
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
                _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) v += WMI_SHX(v, o);
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
                const bool pre = o0 == gw * ROWS_IN_FLIGHT;
                const float bias = pre ? bias_pre : (a.bias ? a.bias[n] : 0.0f);
                const float resid = a.resid ? (pre ? resid_pre : a.resid[(size_t) r * a.ldr + n]) : 0.0f;
                switch (a.epi) {
                    case EPI_F16_BIAS:       ((__half *) a.C)[(size_t) r * a.ldc + n] = f2h(v + bias); break;
                    case EPI_F16_BIAS_GELU:  ((__half *) a.C)[(size_t) r * a.ldc + n] = f2h(gelu16(v + bias)); break;
                    case EPI_F32_BIAS_RESID: ((float *) a.C)[(size_t) r * a.ldc + n] = (v + bias) + resid; break;
                    case EPI_Q_SCALED:       ((__half *) a.C)[(size_t) r * a.ldc + n] = f2h((v + bias) * a.scale); break;
                    case EPI_QKV_DEC: {
                        // segment decided on a wave-uniform value (the 4 rows of this wave iteration never straddle a
                        // q|k|v boundary: S % 4 == 0) — same precaution as in k_gemm.hip, see DESIGN.md §7
                        const int seg = __builtin_amdgcn_readfirstlane(o0 / a.S);
                        const int c = n - seg * a.S;
                        const int ro = ro_pre;                              // KV-cache head (device scalar under graph replay)
                        __half * dst; float val;
                        // cache row: consecutive slots of one sequence batch, or (lock-step chunks) slot ro of chunk r's own cache
                        const int64_t crow = a.lanes ? (int64_t) r * a.cache_row_stride : 0;
                        const int slot = a.lanes ? ro : r + ro;
                        if (seg == 0)      { dst = (__half *) a.C    + (size_t) r * a.ldc;                  val = (v + bias) * a.scale; }
                        else if (seg == 1) { dst = (__half *) a.aux  + crow + (size_t) slot * a.ldaux;      val = v * a.scale; }
                        else               { dst = (__half *) a.aux2 + crow + (size_t) slot * a.ldaux2;     val = v + bias; }
                        dst[c] = f2h(val);
                    } break;
                    case EPI_LOGITS:         ((float *) a.C)[(size_t) r * a.ldc + n] = v; break;
                    default: break;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One activation row (the greedy decode step): the activations never go through LDS.  Every wavefront keeps the row
// in registers — lane L holds columns 512 t + 8 L .. + 8, exactly the columns its 16-byte weight loads multiply —
// and, for the fused-LN variant, normalises it itself (four redundant 64-lane LayerNorms cost less than one LDS
// round trip + barrier on the critical path of a ~5 us kernel).  All NCH x RIF weight loads of a row tile are issued
// together (K = 2048: 16 loads in flight per lane instead of four dependent rounds of four).  The fused attention
// prologues (sa_*, comb_*) still stage through LDS, once.  K <= 512 NCH.
// PRO / EPI >= 0 fix the prologue kind (0 f16 row, 1 LayerNorm, 2 self-attention, 3 cross-attention combine) and the epilogue at
// compile time: the decode step's six hot combinations get kernels without the other kinds' code and without the dispatch on
// kernel arguments (a launch on the step's critical path pays for every instruction and scalar load in front of its first
// memory request); -1 keeps the run-time dispatch.
template <int RIF, int NCH, bool NT, int PRO = -1, int EPI = -1, int HPW = 0, int WPB = 4, bool FS = false>
__global__ __launch_bounds__(64 * WPB) void k_gemv1(const GemvArgs a_in) {
    // Lock-step chunk rows (a_in.lanes != 0, grid.y = rows): row y IS a one-row problem — its own activation row, its own self cache
    // and step record, its own slice of the cross-attention partials — so the arguments are shifted to row y and everything below is
    // the one-row kernel, bit for bit.  The weight rows are read once per row instead of once: grid.x is a multiple of 8, so the
    // workgroups of ALL rows that stream a given weight tile share one XCD (workgroup id % 8 = tile % 8) and the tile comes out of
    // that XCD's L2 for rows 1..n-1 (7.3 MB per decoder layer, 4 MB of L2 per XCD).
    GemvArgs a = a_in;
    if (a_in.lanes) {
        const int y = blockIdx.y;
        const int epi_ = EPI < 0 ? a.epi : EPI;
        if (a.x32) a.x32 += (size_t) y * a.K;
        if (a.a16) a.a16 += (size_t) y * a.K;
        if (a.resid) a.resid += (size_t) y * a.ldr;
        const bool c32 = epi_ == EPI_F32_BIAS_RESID || epi_ == EPI_LOGITS;
        a.C = c32 ? (void *) ((float *) a.C + (size_t) y * a.ldc) : (void *) ((__half *) a.C + (size_t) y * a.ldc);
        if (a.aux)  a.aux  = (__half *) a.aux  + (int64_t) y * a.cache_row_stride;      // (QKV_DEC: the row's self caches)
        if (a.aux2) a.aux2 = (__half *) a.aux2 + (int64_t) y * a.cache_row_stride;
        if (a.row_off) a.row_off += (size_t) y * a.step_stride;
        if (a.sa_q) { a.sa_q += (size_t) y * a.K; a.sa_k += (int64_t) y * a.cache_row_stride; a.sa_v += (int64_t) y * a.cache_row_stride; a.sa_nkv += (size_t) y * a.step_stride; }
        if (a.comb_o) {
            const size_t hs = (size_t) y * (a.K >> 6) * a.comb_ns;                       // partials of row y: [head][slice]
            a.comb_o += hs * 64; a.comb_l += hs; if (a.comb_m) a.comb_m += hs;
        }
        a.lanes = 0; a.n = 1;
    }
    const bool pro_ln = PRO < 0 ? a.ln_g != nullptr : PRO == 1;
    const bool pro_sa = PRO < 0 ? a.sa_q != nullptr : PRO == 2;
    const bool pro_comb = PRO < 0 ? a.comb_o != nullptr : PRO == 3;
    const int epi = EPI < 0 ? a.epi : EPI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half * act = (__half *) smem;                         // [K], attention prologues only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long ts0 = stamp_t0(a.stamps);
    const int K = a.K;
    int nblk = gridDim.x;
    if constexpr (PRO <= 0 && (EPI < 0 || EPI == EPI_F32_BIAS_RESID)) {
        // chained greedy steps (device.cpp): the launch carries ONE extra workgroup that mirrors the host's step record into device
        // memory for the filter kernels — a PCIe read that costs this small-grid launch nothing (32 workgroups on 256 CUs)
        if (a.step_copy_src) {
            nblk -= 1;
            if ((int) blockIdx.x == nblk) {
                if (tid < (int) (sizeof(DecStep) / 4)) ((int32_t *) a.step_copy_dst)[tid] = ((const volatile int32_t *) a.step_copy_src)[tid];
                return;
            }
        }
    }
    const int nwaves = nblk * WPB;                          // WPB wavefronts per workgroup (1: plain rows only — no LDS, no barrier)
    const int gw = blockIdx.x * WPB + wave;

    // Load order.  vmcnt retires in order: a wavefront cannot look at a value before everything requested ahead of it has arrived.
    // The weight rows are the coldest thing this kernel reads (HBM / Infinity Cache, ~1.2 us); the activation row, q / K / V of
    // the step and the cross-attention partials were written by the previous launch and sit in L2 (~0.4 us).  So the prologue's
    // own operands go out FIRST, the weight rows (ALL chunks of the first row tile, plus bias / residual / cache head of the
    // epilogue) right behind them — still before any wait — and the prologue's arithmetic (LayerNorm statistics, the
    // self-attention, the combine) runs while the weights are in flight.  Weights first (round 2) made every prologue start
    // at the weights' latency: LN + q|k|v row ready at +1.8 us of a 2.4 us kernel, self-attention + out at +3.7 of 4.3.
    // Columns past K read column 0 instead: the activation there is exactly 0.
    uint4 wpre[NCH][RIF];
    const bool have_pre = gw * RIF < a.N;
    // after the halving reduction below, row u of a tile ends up on the lanes with (lane / LPR) % RIF == u; lane u * LPR writes it
    constexpr int LPR = 64 / RIF;                           // 16 (RIF 4) or 8 (RIF 8)
    const int wrow = lane / LPR;                            // row this lane would write
    const bool writer = (lane % LPR) == 0;
    float bias_pre = 0.0f, resid_pre = 0.0f; int ro_pre = 0;
    // fused filter statistics (FS): online soft-max partial of the rows this lane writes
    int4 fs_s1 = make_int4(0, 0, 0, 0), fs_s2 = fs_s1; float fs_temp = 0.0f; unsigned char ban_cur = 0, ban_nxt = 0;
    float fs_sum = 0.0f, fs_sum_ts = 0.0f;
    FsMaxIdx fs_all = {-INFINITY, 0x7fffffff}, fs_txt = fs_all, fs_ts = fs_all;
    int zl = 0; asm volatile("" : "+v"(zl));                // a zero the compiler cannot fold (see self_attn_wave: n_kv)
    bool w_issued = false;
    auto issue_weights = [&]() {
        if (w_issued) return;
        w_issued = true;
        __builtin_amdgcn_sched_barrier(0);
        // straight-line: rows clamped to N - 1 (a wavefront past the matrix re-reads its last row), absent vectors read the
        // weights' first bytes and are masked — no load sits in a conditional block (see ln_row_load)
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const int c = lane * 8 + 512 * t, cc = c < K ? c : 0;
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                int o = gw * RIF + u; if (o > a.N - 1) o = a.N - 1;
                wpre[t][u] = ldw<NT>(a.W + (size_t) o * K + cc);
            }
        }
        {
            int n = gw * RIF + wrow; if (n > a.N - 1) n = a.N - 1;
            const float * bp = a.bias ? a.bias + n : (const float *) a.W, * rp = a.resid ? a.resid + n : (const float *) a.W;
            const int32_t * op = a.row_off ? a.row_off : (const int32_t *) a.W;
            // raw values: "absent" is resolved at the use in the epilogue (a select here is a use, i.e. a wait for the weights);
            // ro_pre stays a VGPR until then — as a uniform load it became load -> vmcnt(0) -> readfirstlane on the spot
            bias_pre = *bp; resid_pre = *rp; ro_pre = op[zl];
            if constexpr (FS) {                                  // the row's step record (VGPR copies, see ro_pre) and the first tile's ban bytes
                const int32_t * sp = (const int32_t *) a.fs_step;
                fs_s1 = *(const int4 *) (sp + 4 + zl); fs_s2 = *(const int4 *) (sp + 8 + zl); fs_temp = __int_as_float(sp[12 + zl]);
                ban_cur = a.fs_ban[n];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#if defined(WMI_WEIGHTS_FIRST)
    issue_weights();                                        // A/B build: round 2's order
#endif

    float av[NCH][8];
    // (the specialised instantiations are only launched without a row gather: a.rows[0] as a vector load put a vmcnt(0) in front of
    // every later load group through its result register)
    const int src = PRO >= 0 ? 0 : (a.rows ? a.rows[0] : 0);
    if (pro_ln) {
        // same arithmetic as k_gemv<R>'s prologue: the lock-step VALU path must stay bit-identical to this kernel
        // (ln_row_regs<3> there; with K <= 512 LC the chunks past LC only add exact zeros to the sums: the same bits, a third of the
        // loads and adds for base.en)
        constexpr int LC = NCH < 3 ? NCH : 3;
        float xv[LC][8], gv[LC][8], bv[LC][8], avl[LC][8];
        const float * xr = a.x32 + (size_t) src * K;
        ln_row_load<LC>(xr, K, lane, xv);
        ln_row_load<LC>(a.ln_g, K, lane, gv);
        ln_row_load<LC>(a.ln_b, K, lane, bv);
        issue_weights();
        ln_row_mask<LC>(xv, K, lane); ln_row_mask<LC>(gv, K, lane); ln_row_mask<LC>(bv, K, lane);
        ln_row_compute<LC>(xv, gv, bv, K, a.eps, lane, avl);
#pragma unroll
        for (int t = 0; t < NCH; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) av[t][e] = t < LC ? avl[t < LC ? t : 0][e] : 0.0f;
    } else {
        const __half * arow = a.a16 + (size_t) src * K;
        if (pro_sa) {
            const int H = K / 64;
            float * sc = (float *) (smem + (((size_t) K * sizeof(__half) + 15) & ~(size_t) 15));   // [H][sa_cap]
            float * qf = sc + sa_score_floats(K, a.sa_cap);                                      // [K]
            // the decode loop's case (n_kv <= 64): per wavefront, barrier-free — heads wave, wave + 4 (and wave + 8, wave + 12, ...)
            // all heads of a wavefront in ONE call (ceil(H / 4) of them: 2 for base, 3 small, 4 medium, 5 large): a second call
            // would be a second round trip of loads (small, H = 12: 9.6 -> 6.5 us per launch)
            // (only in the self-attention instantiations: 3-5 heads per wavefront need ~400 VGPRs, which must not leak into the
            // generic kernel that also serves the vocabulary projection at two workgroups per CU)
            bool done = true;
            auto al = [&]() { issue_weights(); };
            if constexpr (PRO == 2) {
                // HPW heads per wavefront (wave, wave + 4, ...), ONE call, no loop: the weight request rides inside the call, and
                // a wait behind a block that MAY have issued loads is vmcnt(0) — the s_waitcnt pass counts only loads it is sure
                // of.  One instantiation per HPW: with all of them in one kernel the code object was 89 KB (instruction cache: 64 KB)
                static_assert(HPW >= 1 && HPW <= 5, "heads per wavefront");
                if (wave < H) {
                    int hs[HPW];
#pragma unroll
                    for (int u = 0; u < HPW; ++u) hs[u] = wave + WPB * u;
                    done = self_attn_wave<HPW>(a.sa_q, a.sa_k, a.sa_v, a.sa_nkv, K, a.sa_cap, hs, H, lane, act, nullptr, sc + wave * 64 * HPW, al);
                }
            } else {
                for (int h0 = wave; h0 < H; h0 += 8) {
                    const int hs[2] = { h0, h0 + 4 };
                    done = self_attn_wave<2>(a.sa_q, a.sa_k, a.sa_v, a.sa_nkv, K, a.sa_cap, hs, H, lane, act, nullptr, sc + wave * 128, al) && done;
                }
            }
            issue_weights();                                 // (a wavefront without a head)
            if (wave >= H) done = a.sa_nkv[0] <= 64;         // a wavefront without a head (H < 4) still has to agree on the branch
            if (!done) {                                     // wave-uniform and the same in every wavefront: it only depends on n_kv
                const uint4 kpre[8] = {};
                self_attn_row(a.sa_q, a.sa_k, a.sa_v, a.sa_nkv[0], K, a.sa_cap, sc, qf, act, kpre, false);
            }
            __syncthreads();
            arow = act;
        } else if (pro_comb) {
            const int H = K / 64, ns = a.comb_ns;
            if (ns == 8) {
                // T = 1500: the 2 x 16 loads of a thread's two elements go out before the first add (element by element this
                // was one round trip per element; a plain loop over the slices one per slice)
                for (int e0 = tid; e0 < K; e0 += 512) {
                    float po[2][8], pl[2][8], pm[2][8];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int e = e0 + u * 256 < K ? e0 + u * 256 : e0, h = e >> 6, dd = e & 63;
#pragma unroll
                        for (int s2 = 0; s2 < 8; ++s2) {
                            po[u][s2] = a.comb_o[((size_t) h * 8 + s2) * 64 + dd]; pl[u][s2] = a.comb_l[(size_t) h * 8 + s2];
                            pm[u][s2] = (a.comb_m ? a.comb_m : a.comb_l)[(size_t) h * 8 + s2];      // (absent: unused below — no select on a loaded value here)
                        }
                    }
                    issue_weights();
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float o = 0.0f, M = -INFINITY; double l = 0.0;
#pragma unroll
                        for (int s2 = 0; s2 < 8; ++s2) M = fmaxf(M, pm[u][s2]);
#pragma unroll
                        for (int s2 = 0; s2 < 8; ++s2) {
                            const float w = !a.comb_m ? 1.0f : pm[u][s2] > -INFINITY ? __expf(pm[u][s2] - M) : 0.0f;
                            o += po[u][s2] * w; l += (double) pl[u][s2] * (double) w;
                        }
                        if (e0 + u * 256 < K) act[e0 + u * 256] = f2h(o * (float) (1.0 / l));
                    }
                }
                issue_weights();
            } else {
            issue_weights();
            for (int e = tid; e < K; e += 256) {
                const int h = e >> 6, dd = e & 63;
                float o = 0.0f, M = -INFINITY; double l = 0.0;
                if (a.comb_m) for (int s2 = 0; s2 < ns; ++s2) M = fmaxf(M, a.comb_m[(size_t) h * ns + s2]);
                for (int s2 = 0; s2 < ns; ++s2) {
                    const float w = comb_weight(a.comb_m, (size_t) h * ns + s2, M);
                    o += a.comb_o[((size_t) h * ns + s2) * 64 + dd] * w; l += (double) a.comb_l[(size_t) h * ns + s2] * (double) w;
                }
                act[e] = f2h(o * (float) (1.0 / l));
            }
            }
            (void) H;
            __syncthreads();
            arow = act;
        }
        uint4 u4[NCH];                                       // unconditional clamped loads, masked afterwards
#pragma unroll
        for (int t = 0; t < NCH; ++t) { const int c = lane * 8 + 512 * t; u4[t] = *(const uint4 *) (arow + (c < K ? c : 0)); }
        issue_weights();                                     // (plain f16 rows: the row first, it is in L2)
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const int c = lane * 8 + 512 * t;
            if (c >= K) u4[t] = make_uint4(0u, 0u, 0u, 0u);
            const __half2 * h = (const __half2 *) &u4[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); av[t][2 * e] = f.x; av[t][2 * e + 1] = f.y; }
        }
    }

    unsigned long long tm1 = stamp_t0(a.stamps);     // activation row ready (LayerNorm / attention prologue done)
    unsigned long long tm2 = 0;
    bool first = have_pre;
    for (int o0 = gw * RIF; o0 < a.N; o0 += nwaves * RIF) {
        uint4 w[NCH][RIF];
#pragma unroll
        for (int t = 0; t < NCH; ++t)
#pragma unroll
            for (int u = 0; u < RIF; ++u) w[t][u] = wpre[t][u];
        (void) first;
        {   // software pipeline over the row tiles: the NEXT tile is requested before this one is reduced
            const int on = o0 + nwaves * RIF;
            if (on < a.N) {
#pragma unroll
                for (int t = 0; t < NCH; ++t) {
                    const int c = lane * 8 + 512 * t, cc = c < K ? c : 0;
#pragma unroll
                    for (int u = 0; u < RIF; ++u) {
                        int o = on + u; if (o > a.N - 1) o = a.N - 1;
                        wpre[t][u] = ldw<NT>(a.W + (size_t) o * K + cc);
                    }
                }
                if constexpr (FS) { int nn = on + wrow; if (nn > a.N - 1) nn = a.N - 1; ban_nxt = a.fs_ban[nn]; }
            }
        }
        float acc[RIF];
#pragma unroll
        for (int u = 0; u < RIF; ++u) acc[u] = 0.0f;
#pragma unroll
        for (int t = 0; t < NCH; ++t)
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                const __half2 * h = (const __half2 *) &w[t][u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 f = __half22float2(h[e]);
                    acc[u] = fmaf(f.x, av[t][2 * e], acc[u]);
                    acc[u] = fmaf(f.y, av[t][2 * e + 1], acc[u]);
                }
            }
        // 64-lane sums of RIF values by halving exchange: at mask m a lane keeps the half of the rows selected by its
        // bit m and receives the partner's partial for those rows — the same pairs in the same order as RIF separate
        // butterflies (bit-identical totals), with RIF - 1 + log2(64 / RIF) shuffles instead of 6 RIF
        float v;
        if (RIF == 8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const bool hi = lane & 32; const float keep = hi ? acc[u + 4] : acc[u], send = hi ? acc[u] : acc[u + 4]; acc[u] = keep + WMI_SHX(send, 32); }
#pragma unroll
            for (int u = 0; u < 2; ++u) { const bool hi = lane & 16; const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2]; acc[u] = keep + WMI_SHX(send, 16); }
            { const bool hi = lane & 8; const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1]; v = keep + WMI_SHX(send, 8); }
            v += WMI_SHX(v, 4); v += WMI_SHX(v, 2); v += WMI_SHX(v, 1);
        } else if (RIF == 2) {
            { const bool hi = lane & 32; const float keep = hi ? acc[RIF - 1] : acc[0], send = hi ? acc[0] : acc[RIF - 1]; v = keep + WMI_SHX(send, 32); }
            v += WMI_SHX(v, 16); v += WMI_SHX(v, 8); v += WMI_SHX(v, 4); v += WMI_SHX(v, 2); v += WMI_SHX(v, 1);
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) { const bool hi = lane & 32; const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2]; acc[u] = keep + WMI_SHX(send, 32); }
            { const bool hi = lane & 16; const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1]; v = keep + WMI_SHX(send, 16); }
            v += WMI_SHX(v, 8); v += WMI_SHX(v, 4); v += WMI_SHX(v, 2); v += WMI_SHX(v, 1);
        }
        if (a.stamps && !tm2) { asm volatile("" :: "v"(v)); tm2 = wall_clock64(); }       // first tile reduced
        if (writer) {
            const int n = o0 + wrow;
            if (n < a.N) {
                const bool pre = o0 == gw * RIF;
                const float bias = a.bias ? (pre ? bias_pre : a.bias[n]) : 0.0f;
                const float resid = a.resid ? (pre ? resid_pre : a.resid[n]) : 0.0f;
                switch (epi) {
                    case EPI_F16_BIAS:       ((__half *) a.C)[n] = f2h(v + bias); break;
                    case EPI_F16_BIAS_GELU:  ((__half *) a.C)[n] = f2h(gelu16(v + bias)); break;
                    case EPI_F32_BIAS_RESID: ((float *) a.C)[n] = (v + bias) + resid; break;
                    case EPI_Q_SCALED:       ((__half *) a.C)[n] = f2h((v + bias) * a.scale); break;
                    case EPI_QKV_DEC: {
                        const int seg = __builtin_amdgcn_readfirstlane(o0 / a.S);      // wave-uniform, see DESIGN.md §7
                        const int c = n - seg * a.S;
                        __half * dst; float val;
                        const int ro = a.row_off ? ro_pre : 0;
                        if (seg == 0)      { dst = (__half *) a.C;                            val = (v + bias) * a.scale; }
                        else if (seg == 1) { dst = (__half *) a.aux  + (size_t) ro * a.ldaux;  val = v * a.scale; }
                        else               { dst = (__half *) a.aux2 + (size_t) ro * a.ldaux2; val = v + bias; }
                        dst[c] = f2h(val);
                    } break;
                    case EPI_LOGITS:         ((float *) a.C)[n] = v; break;
                    default: break;
                }
                if constexpr (FS) {
                    // k_filter_stats' predicate and statistics for vocabulary entry n (W/whisper.cpp:4532-4635), folded into an online
                    // soft-max relative to the running maximum fs_all.v
                    const int flags = fs_s1.x, space_id = fs_s1.y, eot = fs_s1.z, beg = fs_s1.w, ts_floor_end = fs_s2.y, ts_initial_start = fs_s2.z;
                    bool al = !ban_cur;
                    if ((flags & 1) && (n == eot || n == space_id)) al = false;
                    if (flags & 2) { if (flags & 4) { if (n >= beg) al = false; } else { if (n < eot) al = false; } }
                    if (n >= ts_initial_start) al = false;
                    if (n >= beg && n < ts_floor_end) al = false;
                    if (al) {
                        const float lv = fs_temp > 0.0f ? v / fs_temp : v;
                        const float mo = fs_all.v, mn = fmaxf(mo, lv);
                        // (v_exp_f32: ~1e-6 relative on the terms that carry weight; two libm expf per row cost the projection 2.6 us)
                        const float s1 = mo > -INFINITY ? __expf(mo - mn) : 0.0f, e = __expf(lv - mn);
                        fs_sum = fs_sum * s1 + e; fs_sum_ts = fs_sum_ts * s1 + (n >= beg ? e : 0.0f);
                        const FsMaxIdx c = {lv, n};
                        if (c.v > fs_all.v || (c.v == fs_all.v && c.i < fs_all.i)) fs_all = c;
                        if (n < beg) { if (c.v > fs_txt.v || (c.v == fs_txt.v && c.i < fs_txt.i)) fs_txt = c; }
                        else         { if (c.v > fs_ts.v  || (c.v == fs_ts.v  && c.i < fs_ts.i))  fs_ts = c; }
                    }
                }
            }
        }
        if constexpr (FS) ban_cur = ban_nxt;
    }
    if constexpr (FS) {
        // the wavefront's writer lanes (lane % LPR == 0), then the workgroup's wavefronts, merged pairwise: maxima by value then
        // index, sums rescaled to the pair's maximum; one FsPartial per workgroup
        auto merge = [](FsMaxIdx & all, FsMaxIdx & txt, FsMaxIdx & ts, float & sum, float & sum_ts,
                        FsMaxIdx oall, FsMaxIdx otxt, FsMaxIdx ots, float osum, float osum_ts) {
            const float mn = fmaxf(all.v, oall.v);
            const float sa = all.v > -INFINITY ? __expf(all.v - mn) : 0.0f, sb = oall.v > -INFINITY ? __expf(oall.v - mn) : 0.0f;
            sum = sum * sa + osum * sb; sum_ts = sum_ts * sa + osum_ts * sb;
            auto bt = [](FsMaxIdx x, FsMaxIdx y) { return (y.v > x.v || (y.v == x.v && y.i < x.i)) ? y : x; };
            all = bt(all, oall); txt = bt(txt, otxt); ts = bt(ts, ots);
        };
        if (!writer) { fs_all = FsMaxIdx{-INFINITY, 0x7fffffff}; fs_txt = fs_all; fs_ts = fs_all; fs_sum = 0.0f; fs_sum_ts = 0.0f; }
#pragma unroll
        for (int m = LPR; m < 64; m <<= 1) {
            FsMaxIdx oa, ot, oz;
            oa.v = WMI_SHX(fs_all.v, m); oa.i = WMI_SHX(fs_all.i, m); ot.v = WMI_SHX(fs_txt.v, m); ot.i = WMI_SHX(fs_txt.i, m);
            oz.v = WMI_SHX(fs_ts.v, m);  oz.i = WMI_SHX(fs_ts.i, m);
            const float os = WMI_SHX(fs_sum, m), ost = WMI_SHX(fs_sum_ts, m);
            merge(fs_all, fs_txt, fs_ts, fs_sum, fs_sum_ts, oa, ot, oz, os, ost);
        }
        __shared__ FsPartial fs_w[WPB];
        if (lane == 0) { FsPartial pw; pw.all = fs_all; pw.txt = fs_txt; pw.ts = fs_ts; pw.sum = fs_sum; pw.sum_ts = fs_sum_ts; pw.pad[0] = pw.pad[1] = 0.0f; fs_w[wave] = pw; }
        __syncthreads();
        if (tid == 0) {
            FsPartial pw = fs_w[0];
#pragma unroll
            for (int w2 = 1; w2 < WPB; ++w2) merge(pw.all, pw.txt, pw.ts, pw.sum, pw.sum_ts, fs_w[w2].all, fs_w[w2].txt, fs_w[w2].ts, fs_w[w2].sum, fs_w[w2].sum_ts);
            a.fs_part[blockIdx.x] = pw;
        }
    }
#if defined(WMI_SA_STAMPS)
    if (pro_sa && a.stamps) { tm1 = wmi_dbg_t[WMI_SA_STAMPS]; tm2 = wmi_dbg_t[WMI_SA_STAMPS + 1]; }
#endif
    stamp_end(a.stamps, a.stamp_slot, gw, ts0, tm1, tm2);
}

template <int RIF, int NCH, bool NT = false, int PRO = -1, int EPI = -1, int HPW = 0, int WPB = 4, bool FS = false>
void launch_gemv1(const GemvArgs & a, hipStream_t st, int max_blocks = 512) {
    static_assert(WPB == 4 || PRO == 0 || (PRO == 2 && WPB == 8), "one-wavefront workgroups: plain f16 rows only; eight wavefronts: the self-attention prologue");
    size_t smem = 0;
    if (a.sa_q)        smem = ((((size_t) a.K * sizeof(__half)) + 15) & ~(size_t) 15) + (sa_score_floats(a.K, a.sa_cap) + a.K) * sizeof(float);
    else if (a.comb_o) smem = (size_t) a.K * sizeof(__half);
    int blocks = (a.N + WPB * RIF - 1) / (WPB * RIF);
    if (blocks > max_blocks) blocks = max_blocks;
    static std::atomic<uint64_t> lds_ok{0};
    if (smem > 48 * 1024) allow_full_lds((const void *) k_gemv1<RIF, NCH, NT, PRO, EPI, HPW, WPB, FS>, lds_ok);
    if (PRO <= 0 && (EPI < 0 || EPI == EPI_F32_BIAS_RESID) && a.step_copy_src) blocks += 1;       // the step-record mirror (see the kernel)
    int rows = 1;
    if (a.lanes) { rows = a.n; blocks = (blocks + 7) & ~7; }      // lock-step rows: grid.y = row, grid.x a multiple of 8 (see the kernel)
    hipLaunchKernelGGL((k_gemv1<RIF, NCH, NT, PRO, EPI, HPW, WPB, FS>), dim3(blocks, rows), dim3(64 * WPB), smem, st, a);
}

static int logits_blocks_cap() {
    static const int lb = getenv("WMI_LOGITS_BLOCKS") ? std::min(atoi(getenv("WMI_LOGITS_BLOCKS")), FS_MAX_PARTS) : 768;
    return lb;
}
// the decode step's hot (prologue, epilogue) combinations at one row; false = no specialised kernel for these arguments
static bool launch_gemv1_special(const GemvArgs & a, int nch, hipStream_t st) {
    static const bool off = getenv("WMI_GEMV1_GENERIC") != nullptr;       // debug / A-B
    if (off || a.rows) return false;
    const int pro = a.ln_g ? 1 : a.sa_q ? 2 : a.comb_o ? 3 : 0;
    if (a.N >= 16384) {                                      // vocabulary projection: its own lean instantiation (the generic kernel carries
        // the attention prologues: 251 VGPRs, 2 workgroups per CU; this one 143).  768 workgroups = 3 per CU: 9.55 us = 5.56 TB/s
        // (512: 10.6, 1024: 10.8, the generic kernel at 512: 11.4)
        const int lb = logits_blocks_cap();
        if (pro == 1 && a.epi == EPI_LOGITS && a.fs_part) {      // + the logit filters' statistics in the epilogue (greedy step)
            if (nch == 1) { launch_gemv1<8, 1, false, 1, EPI_LOGITS, 0, 4, true>(a, st, lb); return true; }
            if (nch == 2) { launch_gemv1<8, 2, false, 1, EPI_LOGITS, 0, 4, true>(a, st, lb); return true; }
            if (nch == 3) { launch_gemv1<8, 3, false, 1, EPI_LOGITS, 0, 4, true>(a, st, lb); return true; }
        }
        if (pro == 1 && a.epi == EPI_LOGITS && nch == 1) { launch_gemv1<8, 1, false, 1, EPI_LOGITS>(a, st, lb); return true; }
        return false;
    }
    if (nch == 1) {
        if (pro == 1 && a.epi == EPI_QKV_DEC)        { launch_gemv1<4, 1, false, 1, EPI_QKV_DEC>(a, st); return true; }
        if (pro == 1 && a.epi == EPI_F16_BIAS_GELU)  { launch_gemv1<4, 1, false, 1, EPI_F16_BIAS_GELU>(a, st); return true; }
    }
    // the wider f16 models (small / medium: two chunks of 512 columns, large: three) — round 5: the same lean instantiations as base.en's
    // (WMI_GEMV1_WIDE_GENERIC=1: the run-time-dispatch kernel, as before)
    const bool wide_generic = knobs().gemv1_wide_generic;
    if (!wide_generic && !a.lanes && (nch == 2 || nch == 3)) {
        if (pro == 1 && a.epi == EPI_QKV_DEC)       { if (nch == 2) launch_gemv1<4, 2, false, 1, EPI_QKV_DEC>(a, st); else launch_gemv1<4, 3, false, 1, EPI_QKV_DEC>(a, st); return true; }
        if (pro == 1 && a.epi == EPI_F16_BIAS_GELU) { if (nch == 2) launch_gemv1<4, 2, false, 1, EPI_F16_BIAS_GELU>(a, st); else launch_gemv1<4, 3, false, 1, EPI_F16_BIAS_GELU>(a, st); return true; }
        if (pro == 3 && a.epi == EPI_F32_BIAS_RESID) { if (nch == 2) launch_gemv1<4, 2, false, 3, EPI_F32_BIAS_RESID>(a, st); else launch_gemv1<4, 3, false, 3, EPI_F32_BIAS_RESID>(a, st); return true; }
    }
    if (pro == 2 && a.epi == EPI_F32_BIAS_RESID && (a.K % 64) == 0) {
        // self-attention + out projection: (row chunks, heads per wavefront) — tiny 1/2, base 1/2, small 2/3, medium 2/4, large 3/5
        const int hpw = (a.K / 64 + 3) / 4;
        if (nch == 1 && hpw == 1) { launch_gemv1<4, 1, false, 2, EPI_F32_BIAS_RESID, 1>(a, st); return true; }
        // eight heads (base): eight wavefronts with ONE head each and two weight rows per wavefront — the attention is a dependent chain of
        // ~1000 VALU instructions per head pair, i.e. most of this launch's body (row ready at +2.6 .. 2.9 us of 3.2); WMI_SA_WPB=4: the round-3 form
        const int sa_wpb = knobs().sa_wpb;
        if (nch == 1 && hpw == 2 && sa_wpb == 8 && a.K == 512 && !a.lanes) { launch_gemv1<2, 1, false, 2, EPI_F32_BIAS_RESID, 1, 8>(a, st); return true; }
        if (nch == 1 && hpw == 2) { launch_gemv1<4, 1, false, 2, EPI_F32_BIAS_RESID, 2>(a, st); return true; }
        if (nch == 2 && hpw == 3) { launch_gemv1<4, 2, false, 2, EPI_F32_BIAS_RESID, 3>(a, st); return true; }
        if (nch == 2 && hpw == 4) { launch_gemv1<4, 2, false, 2, EPI_F32_BIAS_RESID, 4>(a, st); return true; }
        if (nch == 3 && hpw == 5) { launch_gemv1<4, 3, false, 2, EPI_F32_BIAS_RESID, 5>(a, st); return true; }
        if (nch == 3 && hpw == 4) { launch_gemv1<4, 3, false, 2, EPI_F32_BIAS_RESID, 4>(a, st); return true; }
    }
    if (nch == 1) {
        if (pro == 3 && a.epi == EPI_F32_BIAS_RESID) { launch_gemv1<4, 1, false, 3, EPI_F32_BIAS_RESID>(a, st); return true; }
    }
    if (pro == 0 && a.epi == EPI_F32_BIAS_RESID) {
        if (nch == 3) { launch_gemv1<4, 3, false, 0, EPI_F32_BIAS_RESID>(a, st); return true; }
        if (nch == 4) {
            // mlp.2 of base.en (512 x 2048): 2 MB through 32 four-wavefront workgroups is 64 KB per CU — the CU's own load path is the
            // limit (~60 GB/s per CU: 1 us from "row ready" to "tile reduced"); one-wavefront workgroups put the same rows on 4x the CUs
            // (measured body: 256 threads x 4 rows per wavefront 2.93 us, 64 x 4 rows 2.40, 64 x 2 rows 2.22)
            static const int shape = getenv("WMI_FC2_SHAPE") ? atoi(getenv("WMI_FC2_SHAPE")) : 2;      // A/B knob
            if (shape == 2) launch_gemv1<2, 4, false, 0, EPI_F32_BIAS_RESID, 0, 1>(a, st, 4096);
            else if (shape == 1) launch_gemv1<4, 4, false, 0, EPI_F32_BIAS_RESID, 0, 1>(a, st, 4096);
            else launch_gemv1<4, 4, false, 0, EPI_F32_BIAS_RESID>(a, st);
            return true;
        }
        // mlp.2 of the wider models (K = 4 S = 3072 / 4096 / 5120): register budget = activation row + two row tiles of weights
        if (nch == 5 || nch == 6)  { launch_gemv1<4, 6, false, 0, EPI_F32_BIAS_RESID>(a, st); return true; }
        if (nch == 7 || nch == 8)  { launch_gemv1<2, 8, false, 0, EPI_F32_BIAS_RESID>(a, st); return true; }
        if (nch == 9 || nch == 10) { launch_gemv1<2, 10, false, 0, EPI_F32_BIAS_RESID>(a, st); return true; }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// Lock-step chunk rows (2..16 activation rows, one per chunk) on the matrix cores.  With R rows the VALU
// kernel above spends R x the multiply-adds plus a 64-lane butterfly per (weight row, activation row) pair and
// stops being a pure weight stream (profiles/: 42 us for the vocabulary projection at R = 8 against 15 us at
// R = 1).  Here a wavefront owns a tile of 16 weight rows: A = W[16][32] straight from HBM (16 B per lane, the
// rows are still read front to back exactly once), B = the activation rows from LDS as the 16 MFMA columns
// (columns >= n are zero), C[feature][chunk] accumulates in the MFMA f32 registers — no cross-lane reduction.
//   KSPLIT = false: every wavefront streams whole tiles (grid-stride over tiles) — the vocabulary projection;
//   KSPLIT = true : one tile per workgroup, the four wavefronts take a quarter of K each and are summed through
//                   LDS in a fixed order — the N = S .. 4S projections, where tiles are few and latency matters.
// Same prologues (fused LayerNorm / plain f16 rows) and epilogues as k_gemv.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float    floatx4 __attribute__((ext_vector_type(4)));

// EPI_T: the epilogue at compile time (-1: run-time switch).  With it the four features a lane holds leave as ONE 8- or 16-byte store
// and bias / residual arrive as one 16-byte load each; the run-time form puts every 2-byte store behind its own switch + bounds check,
// i.e. its own basic block with a vmcnt(0) in front (DESIGN.md toolchain hazard 3).
template <bool KSPLIT, int EPI_T = -1>
__global__ __launch_bounds__(256) void k_rows_mfma(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, n = a.n;
    const int lda = K + 8;                                  // LDS row stride in halves: +16 B skews the banks
    __half * act = (__half *) smem;                         // [n][lda]
    const int col = lane & 15, kq = lane >> 4;
    const int ntiles = (a.N + 15) >> 4;
    constexpr int MAXF = 16;                                // A fragments (32 k each) held in registers per pass

    const int kbeg = KSPLIT ? wave * (K >> 2) : 0;
    const int kend = KSPLIT ? kbeg + (K >> 2) : K;
    int nblk = (int) gridDim.x;
    if constexpr (!KSPLIT) {
        // chained lock-step steps (batch.cpp): one extra workgroup of the vocabulary projection mirrors words 4.. of the rows' step
        // records (filter flags, sequence number, temperature) from pinned host memory into the device records the filter kernels
        // read next — the PCIe round trip (~5 us) rides beside the longest launch of the step instead of in front of it
        if (a.rows_mirror_src) {
            nblk -= 1;
            if ((int) blockIdx.x == nblk) {
                constexpr int W = (int) (sizeof(DecStep) / 4);
                for (int i = tid; i < n * W; i += 256)
                    if ((i % W) >= 4) ((int32_t *) a.rows_mirror_dst)[i] = ((const volatile int32_t *) a.rows_mirror_src)[i];
                return;
            }
        }
    }
    int tile = KSPLIT ? (int) blockIdx.x : (int) (blockIdx.x * 4 + wave);
    const int tstride = KSPLIT ? nblk : nblk * 4;

    // ---- weight prefetch of the first pass (independent of the activations)
    uint4 wf[MAXF];
    const int nf0 = min(MAXF, (kend - kbeg) >> 5);
    if (tile < ntiles) {
        int row = tile * 16 + col; if (row > a.N - 1) row = a.N - 1;
        const __half * wp = a.W + (size_t) row * K + kbeg + kq * 8;
#pragma unroll
        for (int f = 0; f < MAXF; ++f) if (f < nf0) wf[f] = *(const uint4 *) (wp + f * 32);
    }
    // f16 activation rows without a LayerNorm in front (out projections, mlp.2): the B fragments of the first pass are requested
    // straight from global memory next to the weights — no LDS copy, no barrier before the first MFMA (the staged copy measured
    // 4 us of a 7.5 us launch at 8 rows x 2048)
    const bool direct_b = KSPLIT && !a.ln_g && !a.rows && !a.comb_o;      // (the vocabulary projection always has its LayerNorm: no registers for bf there)
    uint4 bf[MAXF];
    const __half * brow = a.a16 + (size_t) (col < n ? col : 0) * K + kq * 8;
    if (direct_b && tile < ntiles) {
#pragma unroll
        for (int f = 0; f < MAXF; ++f) if (f < nf0) bf[f] = *(const uint4 *) (brow + kbeg + f * 32);
    }
    int ro_pre = 0;
    if (a.row_off && col < n) ro_pre = a.lanes ? a.row_off[col * a.step_stride] : *a.row_off;

    // epilogue operands of the first tile (bias, residual): independent of the prologue, requested now
    float bias_pre[4] = {0.f, 0.f, 0.f, 0.f}, resid_pre[4] = {0.f, 0.f, 0.f, 0.f};
    const int tile0 = tile;
    constexpr bool VEC = EPI_T >= 0;                         // host side guarantees 16 | N, 4 | ldc / ldr, 16-byte aligned operands
    if (tile < ntiles && col < n && (!KSPLIT || wave == 0)) {
        if constexpr (VEC && EPI_T != EPI_LOGITS) {
            const int nf0_ = tile * 16 + kq * 4;
            // straight-line loads (an absent operand reads the weights' first bytes and is ignored at its use)
            const float4 b4 = *(const float4 *) (a.bias ? (const void *) (a.bias + nf0_) : (const void *) a.W);
            bias_pre[0] = b4.x; bias_pre[1] = b4.y; bias_pre[2] = b4.z; bias_pre[3] = b4.w;
            if constexpr (EPI_T == EPI_F32_BIAS_RESID) {
                const float4 r4 = *(const float4 *) (a.resid + (size_t) col * a.ldr + nf0_);
                resid_pre[0] = r4.x; resid_pre[1] = r4.y; resid_pre[2] = r4.z; resid_pre[3] = r4.w;
            }
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nf_ = tile * 16 + kq * 4 + r;
            if (nf_ < a.N) {
                if (a.bias) bias_pre[r] = a.bias[nf_];
                if (a.resid) resid_pre[r] = a.resid[(size_t) col * a.ldr + nf_];
            }
        }
        }
    }

    // ---- prologue: activation rows as f16 in LDS
    if (a.ln_g) {                                           // K <= 1536; same arithmetic as k_gemv / k_gemv1 (ln_row_regs)
        // rows wave, wave + 4, ... of this wavefront: the x vectors of all of them and gain / bias go out together — row after
        // row, each LayerNorm waited for its own loads.  Two instantiations so that n <= 8 does not issue loads for rows it lacks.
        if constexpr (!KSPLIT) {
            // vocabulary projection: 53 MB streamed by several workgroups per CU — the register budget decides the occupancy, so
            // the rows are normalised one after the other with the loads inside (measured: 19.8 us at 8 rows against 30 us with
            // the all-rows-at-once form below)
            auto ln_seq = [&](auto nc_tag) {
                constexpr int NC = decltype(nc_tag)::value;      // 512-column chunks of a row: registers (and loads) only for the chunks the model has
                if constexpr (NC <= 2) {
                    // two rows at a time (rows r and r + 4 of this wavefront): both rows' loads go out together — row after row each LayerNorm
                    // waited its own ~2 us round trip for a row the previous launch wrote on another XCD (2 round trips at 8 rows, 4 at 16).
                    // 16 * NC more registers than one row: still two workgroups per CU.  Per row the arithmetic is ln_row_compute's.
                    for (int r0 = wave; r0 < n; r0 += 8) {
                        const int r1 = r0 + 4 < n ? r0 + 4 : r0;
                        const int s0 = a.rows ? a.rows[r0] : r0, s1 = a.rows ? a.rows[r1] : r1;
                        float xv[2][NC][8], gv[NC][8], bv[NC][8];
                        ln_row_load<NC>(a.x32 + (size_t) s0 * K, K, lane, xv[0]);
                        ln_row_load<NC>(a.x32 + (size_t) s1 * K, K, lane, xv[1]);
                        ln_row_load<NC>(a.ln_g, K, lane, gv);
                        ln_row_load<NC>(a.ln_b, K, lane, bv);
                        __builtin_amdgcn_sched_barrier(0);
                        ln_row_mask<NC>(xv[0], K, lane); ln_row_mask<NC>(xv[1], K, lane); ln_row_mask<NC>(gv, K, lane); ln_row_mask<NC>(bv, K, lane);
                        ln_rows_compute<2, NC>(xv, gv, bv, K, a.eps, lane);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int r = q == 0 ? r0 : r0 + 4;
                            if (r < n) {
#pragma unroll
                                for (int t = 0; t < NC; ++t) {
                                    const int c = lane * 8 + 512 * t;
                                    if (c < K) {
                                        __half2 h[4];
#pragma unroll
                                        for (int e = 0; e < 4; ++e) h[e] = __floats2half2_rn(xv[q][t][2 * e], xv[q][t][2 * e + 1]);
                                        *(uint4 *) (act + r * lda + c) = *(const uint4 *) h;
                                    }
                                }
                            }
                        }
                    }
                } else
                for (int r = wave; r < n; r += 4) {
                    const int src = a.rows ? a.rows[r] : r;
                    float av[NC][8];
                    ln_row_regs<NC>(a.x32 + (size_t) src * K, a.ln_g, a.ln_b, K, a.eps, lane, av);
#pragma unroll
                    for (int t = 0; t < NC; ++t) {
                        const int c = lane * 8 + 512 * t;
                        if (c < K) {
                            __half2 h[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = __floats2half2_rn(av[t][2 * e], av[t][2 * e + 1]);
                            *(uint4 *) (act + r * lda + c) = *(const uint4 *) h;
                        }
                    }
                }
            };
            if (K <= 512) ln_seq(std::integral_constant<int, 1>{});
            else if (K <= 1024) ln_seq(std::integral_constant<int, 2>{});
            else ln_seq(std::integral_constant<int, 3>{});
        } else {
        auto ln_rows = [&](auto rw_tag) {
            constexpr int RW = decltype(rw_tag)::value;
            float xv[RW][3][8], gv[3][8], bv[3][8];
            int src[RW];
#pragma unroll
            for (int q = 0; q < RW; ++q) { const int r = wave + 4 * q, rc = r < n ? r : (n - 1); src[q] = a.rows ? a.rows[rc] : rc; }
#pragma unroll
            for (int q = 0; q < RW; ++q) ln_row_load<3>(a.x32 + (size_t) src[q] * K, K, lane, xv[q]);
            ln_row_load<3>(a.ln_g, K, lane, gv);
            ln_row_load<3>(a.ln_b, K, lane, bv);
            __builtin_amdgcn_sched_barrier(0);              // keep the loads together: the scheduler sinks each to its first use
#pragma unroll
            for (int q = 0; q < RW; ++q) ln_row_mask<3>(xv[q], K, lane);
            ln_row_mask<3>(gv, K, lane); ln_row_mask<3>(bv, K, lane);
            // K-split launches (one tile per workgroup, a CU to itself): the rows' LayerNorms interleaved
            ln_rows_compute<RW, 3>(xv, gv, bv, K, a.eps, lane);
#pragma unroll
            for (int q = 0; q < RW; ++q) {
                const int r = wave + 4 * q;
                if (r < n) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int c = lane * 8 + 512 * t;
                        if (c < K) {
                            __half2 h[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = __floats2half2_rn(xv[q][t][2 * e], xv[q][t][2 * e + 1]);
                            *(uint4 *) (act + r * lda + c) = *(const uint4 *) h;
                        }
                    }
                }
            }
        };
        if (n <= 4) ln_rows(std::integral_constant<int, 1>{});
        else if (n <= 8) ln_rows(std::integral_constant<int, 2>{});
        else ln_rows(std::integral_constant<int, 4>{});
        }
    } else if (a.comb_o) {
        // B = the combined cross-attention partials of the rows (k_xattn_combine's arithmetic, f16 like its output): 16-byte pieces
        // (row, 8 dims of a head) flattened over the workgroup, the ns slices of a piece requested together — one round trip
        // instead of a combine launch per layer of the lock-step step
        const int cpr = K >> 3, total = n * cpr, H = K >> 6, ns = a.comb_ns;
        for (int e0 = 0; e0 < total; e0 += 256) {
            const int e = e0 + tid;
            if (e < total) {
                const int r = e / cpr, c = e - r * cpr, kk = c * 8, h = kk >> 6, dd = kk & 63;
                const size_t row = (size_t) r * H + h;
                float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; double l = 0.0;
                if (ns == 8) {
                    float4 p0[8], p1[8]; float pl[8], pm[8];
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2) {
                        p0[s2] = *(const float4 *) (a.comb_o + (row * 8 + s2) * 64 + dd); p1[s2] = *(const float4 *) (a.comb_o + (row * 8 + s2) * 64 + dd + 4);
                        pl[s2] = a.comb_l[row * 8 + s2]; pm[s2] = a.comb_m ? a.comb_m[row * 8 + s2] : 0.0f;
                    }
                    float M = -INFINITY;
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2) M = fmaxf(M, pm[s2]);
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2) {
                        const float w = !a.comb_m ? 1.0f : pm[s2] > -INFINITY ? __expf(pm[s2] - M) : 0.0f;
                        o[0] += p0[s2].x * w; o[1] += p0[s2].y * w; o[2] += p0[s2].z * w; o[3] += p0[s2].w * w;
                        o[4] += p1[s2].x * w; o[5] += p1[s2].y * w; o[6] += p1[s2].z * w; o[7] += p1[s2].w * w;
                        l += (double) pl[s2] * (double) w;
                    }
                } else {
                    float M = -INFINITY;
                    if (a.comb_m) for (int s2 = 0; s2 < ns; ++s2) M = fmaxf(M, a.comb_m[row * ns + s2]);
                    for (int s2 = 0; s2 < ns; ++s2) {
                        const float w = comb_weight(a.comb_m, row * ns + s2, M);
                        const float4 q0 = *(const float4 *) (a.comb_o + (row * ns + s2) * 64 + dd), q1 = *(const float4 *) (a.comb_o + (row * ns + s2) * 64 + dd + 4);
                        o[0] += q0.x * w; o[1] += q0.y * w; o[2] += q0.z * w; o[3] += q0.w * w;
                        o[4] += q1.x * w; o[5] += q1.y * w; o[6] += q1.z * w; o[7] += q1.w * w;
                        l += (double) a.comb_l[row * ns + s2] * (double) w;
                    }
                }
                const float inv = (float) (1.0 / l);
                __half hv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) hv[q] = f2h(o[q] * inv);
                *(uint4 *) (act + r * lda + kk) = *(const uint4 *) hv;
            }
        }
    } else if (!direct_b) {
        // n rows of K / 8 16-byte pieces, flattened over the workgroup; the loads of a group of pieces before their LDS stores
        // (as a row-by-row copy loop this was one dependent round trip per row); group size by the amount of work
        const int cpr = K >> 3, total = n * cpr;
        auto copy_rows = [&](auto cp_tag) {
            constexpr int CP = decltype(cp_tag)::value;
            for (int e0 = 0; e0 < total; e0 += 256 * CP) {
                // (as HIP's uint4 struct the array was left in SCRATCH memory — 144 bytes per lane in every k_rows_mfma instantiation, round 5)
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4 tmp[CP];
                int src[CP], cc[CP], dst[CP];                // dst: LDS offset in halves, -1 = no piece
#pragma unroll
                for (int q = 0; q < CP; ++q) {
                    const int e = e0 + tid + 256 * q, ec = e < total ? e : 0;
                    const int r = ec / cpr;
                    cc[q] = ec - r * cpr; dst[q] = e < total ? r * lda + cc[q] * 8 : -1;
                    src[q] = a.rows ? a.rows[r] : r;
                }
#pragma unroll
                for (int q = 0; q < CP; ++q) tmp[q] = ((const u32x4 *) (a.a16 + (size_t) src[q] * K))[cc[q]];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < CP; ++q) if (dst[q] >= 0) *(u32x4 *) (act + dst[q]) = tmp[q];
            }
        };
        if (total <= 512) copy_rows(std::integral_constant<int, 2>{});
        else if (total <= 1024) copy_rows(std::integral_constant<int, 4>{});
        else copy_rows(std::integral_constant<int, 8>{});
    }
    __syncthreads();

    float * red = (float *) (smem + (((size_t) n * lda * sizeof(__half) + 15) & ~(size_t) 15));   // KSPLIT: [4][64][4]
    bool first = true;
    for (; tile < ntiles; tile += tstride) {
        floatx4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        int row = tile * 16 + col; if (row > a.N - 1) row = a.N - 1;
        const __half * wrow = a.W + (size_t) row * K + kq * 8;
        for (int k0 = kbeg; k0 < kend; k0 += MAXF * 32) {
            const int nf = min(MAXF, (kend - k0) >> 5);
            if (!(first && k0 == kbeg)) {
#pragma unroll
                for (int f = 0; f < MAXF; ++f) if (f < nf) wf[f] = *(const uint4 *) (wrow + k0 + f * 32);
                if (direct_b) {
#pragma unroll
                    for (int f = 0; f < MAXF; ++f) if (f < nf) bf[f] = *(const uint4 *) (brow + k0 + f * 32);
                }
            }
#pragma unroll
            for (int f = 0; f < MAXF; ++f) {
                if (f < nf) {
                    uint4 bu = make_uint4(0u, 0u, 0u, 0u);
                    if (direct_b) { if (col < n) bu = bf[f]; }
                    else if (col < n) bu = *(const uint4 *) (act + col * lda + k0 + f * 32 + kq * 8);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const half8 *) &wf[f], *(const half8 *) &bu, acc, 0, 0, 0);
                }
            }
        }
        first = false;
        if (!KSPLIT && (kend - kbeg) <= MAXF * 32 && tile + tstride < ntiles) {
            // grid-stride over tiles (the vocabulary projection): the next tile's weights are requested before this tile's epilogue,
            // so a wavefront's stream never waits a whole round trip between tiles
            int nrow = (tile + tstride) * 16 + col; if (nrow > a.N - 1) nrow = a.N - 1;
            const __half * wp = a.W + (size_t) nrow * K + kbeg + kq * 8;
#pragma unroll
            for (int f = 0; f < MAXF; ++f) if (f < nf0) wf[f] = *(const uint4 *) (wp + f * 32);
            first = true;
        }
        if (KSPLIT) {
            if (tile != (int) blockIdx.x) __syncthreads();          // red reused across tiles (grid-stride)
            *(floatx4 *) (red + ((size_t) wave * 64 + lane) * 4) = acc;
            __syncthreads();
            if (wave != 0) continue;
            const floatx4 p1 = *(const floatx4 *) (red + ((size_t) 64 + lane) * 4);
            const floatx4 p2 = *(const floatx4 *) (red + ((size_t) 128 + lane) * 4);
            const floatx4 p3 = *(const floatx4 *) (red + ((size_t) 192 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = (acc[r] + p1[r]) + (p2[r] + p3[r]);
        }
        if (col >= n) continue;
        // epilogue: this lane holds C[feature = tile*16 + kq*4 + r][chunk row = col]
        const int seg = __builtin_amdgcn_readfirstlane((tile * 16) / (a.S > 0 ? a.S : 1));   // wave-uniform (16 | S), see DESIGN.md §7
        if (VEC && (EPI_T != EPI_LOGITS || tile * 16 + 16 <= a.N)) {     // (the vocabulary's last tile is ragged: 51 864 = 16 x 3 241 + 8)
            const int nf0_ = tile * 16 + kq * 4;
            const bool pre = tile == tile0;
            float b[4];
            if (pre) { b[0] = bias_pre[0]; b[1] = bias_pre[1]; b[2] = bias_pre[2]; b[3] = bias_pre[3]; }
            else if (a.bias) { const float4 b4 = *(const float4 *) (a.bias + nf0_); b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w; }
            else { b[0] = b[1] = b[2] = b[3] = 0.0f; }
            if (!a.bias) { b[0] = b[1] = b[2] = b[3] = 0.0f; }
            typedef _Float16 half4v __attribute__((ext_vector_type(4)));
            if constexpr (EPI_T == EPI_F32_BIAS_RESID) {
                float rr[4];
                if (pre) { rr[0] = resid_pre[0]; rr[1] = resid_pre[1]; rr[2] = resid_pre[2]; rr[3] = resid_pre[3]; }
                else { const float4 r4 = *(const float4 *) (a.resid + (size_t) col * a.ldr + nf0_); rr[0] = r4.x; rr[1] = r4.y; rr[2] = r4.z; rr[3] = r4.w; }
                float4 o; o.x = (acc[0] + b[0]) + rr[0]; o.y = (acc[1] + b[1]) + rr[1]; o.z = (acc[2] + b[2]) + rr[2]; o.w = (acc[3] + b[3]) + rr[3];
                *(float4 *) ((float *) a.C + (size_t) col * a.ldc + nf0_) = o;
            } else if constexpr (EPI_T == EPI_F16_BIAS_GELU) {
                half4v h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (_Float16) __half2float(f2h(gelu16(acc[r] + b[r])));
                *(half4v *) ((__half *) a.C + (size_t) col * a.ldc + nf0_) = h;
            } else if constexpr (EPI_T == EPI_LOGITS) {
                float4 o; o.x = acc[0]; o.y = acc[1]; o.z = acc[2]; o.w = acc[3];
                *(float4 *) ((float *) a.C + (size_t) col * a.ldc + nf0_) = o;
            } else if constexpr (EPI_T == EPI_QKV_DEC) {
                const int c = nf0_ - seg * a.S;
                const int64_t crow = a.lanes ? (int64_t) col * a.cache_row_stride : 0;
                const int slot = a.lanes ? ro_pre : col + ro_pre;
                __half * dst; float val[4];
                if (seg == 0)      { dst = (__half *) a.C    + (size_t) col * a.ldc;            for (int r = 0; r < 4; ++r) val[r] = (acc[r] + b[r]) * a.scale; }
                else if (seg == 1) { dst = (__half *) a.aux  + crow + (size_t) slot * a.ldaux;  for (int r = 0; r < 4; ++r) val[r] = acc[r] * a.scale; }
                else               { dst = (__half *) a.aux2 + crow + (size_t) slot * a.ldaux2; for (int r = 0; r < 4; ++r) val[r] = acc[r] + b[r]; }
                half4v h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (_Float16) pin_f32(val[r]);
                *(half4v *) (dst + c) = h;
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nf_ = tile * 16 + kq * 4 + r;
            if (nf_ >= a.N) continue;
            const float v = acc[r];
            const bool pre = tile == tile0;
            const float bias = pre ? bias_pre[r] : (a.bias ? a.bias[nf_] : 0.0f);
            switch (a.epi) {
                case EPI_F16_BIAS:       ((__half *) a.C)[(size_t) col * a.ldc + nf_] = f2h(v + bias); break;
                case EPI_F16_BIAS_GELU:  ((__half *) a.C)[(size_t) col * a.ldc + nf_] = f2h(gelu16(v + bias)); break;
                case EPI_F32_BIAS_RESID: ((float *) a.C)[(size_t) col * a.ldc + nf_] = (v + bias) + (pre ? resid_pre[r] : a.resid[(size_t) col * a.ldr + nf_]); break;
                case EPI_Q_SCALED:       ((__half *) a.C)[(size_t) col * a.ldc + nf_] = f2h((v + bias) * a.scale); break;
                case EPI_QKV_DEC: {
                    const int c = nf_ - seg * a.S;
                    const int64_t crow = a.lanes ? (int64_t) col * a.cache_row_stride : 0;
                    const int slot = a.lanes ? ro_pre : col + ro_pre;
                    __half * dst; float val;
                    if (seg == 0)      { dst = (__half *) a.C    + (size_t) col * a.ldc;              val = (v + bias) * a.scale; }
                    else if (seg == 1) { dst = (__half *) a.aux  + crow + (size_t) slot * a.ldaux;    val = v * a.scale; }
                    else               { dst = (__half *) a.aux2 + crow + (size_t) slot * a.ldaux2;   val = v + bias; }
                    dst[c] = f2h(val);
                } break;
                case EPI_LOGITS:         ((float *) a.C)[(size_t) col * a.ldc + nf_] = v; break;
                default: break;
            }
        }
    }
}

template <bool KSPLIT, int EPI_T = -1>
void launch_rows_mfma(const GemvArgs & a, hipStream_t st) {
    size_t smem = (((size_t) a.n * (a.K + 8) * sizeof(__half) + 15) & ~(size_t) 15) + (KSPLIT ? 4 * 64 * 4 * sizeof(float) : 0);
    const int ntiles = (a.N + 15) / 16;
    int blocks = KSPLIT ? ntiles : (ntiles + 3) / 4;
    // vocabulary projection: 2 workgroups per CU, all resident at once, each wavefront walking ~1.6 tiles with the next tile's
    // weights in flight (811 workgroups, one tile per wavefront: 19.9 us at 8 rows; 512: 15.9; 576 and more: 20.5)
    static const int cap = getenv("WMI_ROWS_BLOCKS") ? atoi(getenv("WMI_ROWS_BLOCKS")) : 512;        // A/B knob
    if (blocks > 1024) blocks = 1024;
    const bool mirror = !KSPLIT && a.rows_mirror_src;        // the step-record mirror (see the kernel): one more workgroup, inside the resident-sized grid
    if (!KSPLIT && blocks > cap - (mirror ? 1 : 0)) blocks = cap - (mirror ? 1 : 0);
    static std::atomic<uint64_t> lds_ok{0};
    if (smem > 48 * 1024) allow_full_lds((const void *) k_rows_mfma<KSPLIT, EPI_T>, lds_ok);
    if (mirror) blocks += 1;
    hipLaunchKernelGGL((k_rows_mfma<KSPLIT, EPI_T>), dim3(blocks), dim3(256), smem, st, a);
}

template <int R, int RIF, bool NT = false>
void launch_gemv_t(const GemvArgs & a, hipStream_t st, int max_blocks = 512) {
    size_t smem = (size_t) R * a.K * sizeof(__half);
    if (a.sa_q) smem = ((smem + 15) & ~(size_t) 15) + (sa_score_floats(a.K, a.sa_cap) + a.K) * sizeof(float);
    int blocks = (a.N + 4 * RIF - 1) / (4 * RIF);
    if (blocks > max_blocks) blocks = max_blocks;       // 2 workgroups per CU; longer rows-per-wave loops are software-pipelined
    static std::atomic<uint64_t> lds_ok{0};
    if (smem > 48 * 1024) allow_full_lds((const void *) k_gemv<R, RIF, NT>, lds_ok);
    hipLaunchKernelGGL((k_gemv<R, RIF, NT>), dim3(blocks), dim3(256), smem, st, a);
}

template <int R>
void launch_gemv(const GemvArgs & a, hipStream_t st) {
    // the vocabulary projection streams 53 MB: keep 8 rows (8 KB) per wavefront in flight; small matrices use 4
    if (R == 1 && a.N >= 16384) {
        static const int rif = getenv("WMI_LOGITS_RIF") ? atoi(getenv("WMI_LOGITS_RIF")) : 8;          // A/B knobs
        static const int mb = getenv("WMI_LOGITS_BLOCKS") ? atoi(getenv("WMI_LOGITS_BLOCKS")) : 512;
        static const bool nt = getenv("WMI_LOGITS_NT") != nullptr;
        if (rif == 16)     { if (nt) launch_gemv_t<1, 16, true>(a, st, mb); else launch_gemv_t<1, 16>(a, st, mb); }
        else if (rif == 4) { if (nt) launch_gemv_t<1, 4, true>(a, st, mb);  else launch_gemv_t<1, 4>(a, st, mb); }
        else               { if (nt) launch_gemv_t<1, 8, true>(a, st, mb);  else launch_gemv_t<1, 8>(a, st, mb); }
    }
    else launch_gemv_t<R, 4>(a, st);
}

} // namespace

void dec_embed(const int32_t * tokens, const int32_t * pos, int n, int S, const __half * te, const float * pe,
               float * x, hipStream_t st) {
    hipLaunchKernelGGL(k_dec_embed, dim3(n), dim3(256), 0, st, tokens, pos, S, te, pe, x);
}

void dec_embed_step(const DecStep * host_step, DecStep * dev_step, int S, const __half * te, const float * pe, float * x,
                    hipStream_t st, int n_rows) {
    hipLaunchKernelGGL(k_dec_embed_step, dim3(n_rows), dim3(256), 0, st, host_step, dev_step, S, te, pe, x, stamp_next());
}

void self_attn_rows(const __half * q, int n, int K, const __half * kc, const __half * vc, int64_t cache_row_stride,
                    const int32_t * n_kv, int step_stride, int cap, __half * out, hipStream_t st, float * out32, bool long_cache,
                    const void * mirror_src, void * mirror_dst) {
    if (long_cache) {
        const size_t smem = ((size_t) cap + 64 + 256 + 4) * sizeof(float);
        hipLaunchKernelGGL(k_self_attn_rows_long, dim3(n, K / 64), dim3(256), smem, st, q, kc, vc, cache_row_stride, n_kv, step_stride, K, cap, out, out32);
        return;
    }
    const size_t smem = ((size_t) cap + 64) * sizeof(float);
    hipLaunchKernelGGL(k_self_attn_rows, dim3(n, K / 64 + (mirror_src ? 1 : 0)), dim3(64), smem, st, q, kc, vc, cache_row_stride, n_kv, step_stride, K, cap, out, out32,
                       (const int32_t *) mirror_src, (int32_t *) mirror_dst);
}

static bool g_rows_valu = false;
static std::atomic<int> g_mode_epoch{0};
int  mode_epoch() { return g_mode_epoch.load(std::memory_order_relaxed); }
void bump_mode_epoch() { g_mode_epoch.fetch_add(1, std::memory_order_relaxed); }
void set_rows_valu(bool on) { if (on != g_rows_valu) bump_mode_epoch(); g_rows_valu = on; }
bool rows_valu_enabled() { static const bool env = getenv("WMI_ROWS_VALU") != nullptr; return env || g_rows_valu; }

static void gemv_(const GemvArgs & a, hipStream_t st);
int gemv_fused_parts(const GemvArgs & a) {
    // mirrors the dispatch below: one row, LayerNorm prologue, EPI_LOGITS, rows of <= 1536 columns, no row gather
    static const bool off = getenv("WMI_GEMV1_GENERIC") != nullptr || getenv("WMI_GEMV1_OFF") != nullptr || getenv("WMI_GEMV1_MASK") != nullptr ||
                            getenv("WMI_NO_FUSED_STATS") != nullptr;
    if (off || !a.fs_part || a.n != 1 || a.lanes || a.rows || !a.ln_g || a.epi != EPI_LOGITS || a.N < 16384 || a.K > 1536 || (a.K % 8) != 0) return 0;
    const int blocks = (a.N + 31) / 32;
    return std::min(blocks, logits_blocks_cap());
}
// ------------------------------------------------------------------------------------------------ one MLP, one launch
// Phase 1 = k_gemv1<4, 1, false, 1, EPI_F16_BIAS_GELU> (LayerNorm in every wavefront, four weight rows per wavefront, the same halving
// reduction), phase 2 = the mlp.2 projection with ONE row per wavefront (4 S rows of phase 1 on S wavefronts -> S rows of phase 2 on the same
// S wavefronts).  Per output the operations and their order are the two-launch form's (k_gemv1: fmaf chain over (chunk, element), 64-lane
// sum with the xor order 32, 16, .., 1 — addition commutes, so halving and plain butterflies give the same bits), so the step stays
// bit-identical to the other forms (tests/test_gpu_variants.py).
//
// The hand-off (scratch/lab/persist_chain.hip measured it: 1.8 us per dependent 512 x 512 phase inside one launch against 2.8 us per launch
// of a captured chain): a hidden value pair leaves its producer as ONE 8-byte store {f16 pair, tag} that bypasses the caches (sc1); a
// consumer workgroup sweeps the 2 S granules once — two 16-byte sc0 sc1 loads per thread — until every tag is this launch's, stages the
// payloads in LDS, one barrier.  No flag, no fence: a granule is valid exactly when its tag matches; the tag counts this kernel's launches
// (see `epoch` in the kernel).  Measured and dropped: tags from a device-scope counter — bumped when a workgroup leaves it kept the launch
// open ~1.2 us past its last store; as a ticket per wavefront at the start, 512 same-address atomics took ~6 us and the row's loads
// retire behind them.
// All workgroups are resident at once (checked against the runtime's occupancy in mlp_pair()): the spin cannot starve a producer.  The caller
// uses this kernel only while its transcription is alone in the process (device.cpp: BusyScope): beside other contexts' launches the
// workgroups start at different times and the early ones poll (profiles/r05g_* §9).
// NCH1 / NCH2: 512-column chunks of a row of W1 (S columns) / of W2 (4 S columns): (1, 3) tiny, (1, 4) base, (2, 6) small, (2, 8) medium, (3, 10) large
template <int NCH1, int NCH2, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_mlp_pair(const MlpPairArgs a, float * __restrict__ xio, int G, const Stamp sp) {
    __shared__ __attribute__((aligned(16))) uint32_t hrow[NCH2 * 256];       // the hidden row as f16 pairs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long ts0 = stamp_t0(sp.base);
    if ((int) blockIdx.x == G) {                            // chained greedy steps: the host's step record -> device (see k_gemv1)
        if (tid < (int) (sizeof(DecStep) / 4)) ((int32_t *) a.step_copy_dst)[tid] = ((const volatile int32_t *) a.step_copy_src)[tid];
        return;
    }
    const int S = a.S, K2 = 4 * S;
    constexpr int NT = 64 * WPB;
    const int gw = blockIdx.x * WPB + wave;                 // phase 1: rows 4 gw .. + 4 of W1; phase 2: row gw of W2
    // this launch's tag: epoch[par] + 1, where `par` alternates from one launch of this kernel to the next (a.par: the layer's parity) and
    // THIS launch leaves epoch[par ^ 1] = tag for the next one.  No workgroup of a launch reads the word the launch writes, so a workgroup that
    // starts late sees the same value as the first one; nothing is ever reset, stale granules carry smaller tags.
    // (the two words are requested behind the weight rows and looked at where the granules are built: see tag_v below)
    constexpr int RIF = 4, LPR = 16;
    const int wrow = lane / LPR; const bool writer = (lane % LPR) == 0;
    // ---- phase 1 loads: the row first (L2), gain / bias, then the weight rows of both phases (HBM / Infinity Cache)
    float xv[NCH1][8], gv[NCH1][8], bv[NCH1][8], av1[NCH1][8];
    ln_row_load<NCH1>(xio, S, lane, xv);
    ln_row_load<NCH1>(a.ln_g, S, lane, gv);
    ln_row_load<NCH1>(a.ln_b, S, lane, bv);
    __builtin_amdgcn_sched_barrier(0);
    uint4 w1[NCH1][RIF], w2[NCH2];
#pragma unroll
    for (int t = 0; t < NCH1; ++t) {
        const int c = lane * 8 + 512 * t, c1 = c < S ? c : 0;    // (columns past S: the activation there is exactly 0)
#pragma unroll
        for (int u = 0; u < RIF; ++u) w1[t][u] = *(const uint4 *) (a.W1 + (size_t) (gw * RIF + u) * S + c1);
    }
    const float bias1 = a.b1 ? a.b1[gw * RIF + wrow] : 0.0f;
#pragma unroll
    for (int t = 0; t < NCH2; ++t) { const int c = lane * 8 + 512 * t; w2[t] = *(const uint4 *) (a.W2 + (size_t) gw * K2 + (c < K2 ? c : 0)); }
    const float bias2 = a.b2 ? a.b2[gw] : 0.0f;
    const float resid2 = xio[gw];
    // the tag words LAST and through a lane offset the compiler cannot fold: as a uniform load at the kernel's top the tag was
    // load -> vmcnt(0) -> readfirstlane, a memory round trip in front of the row's loads (round 6, found in k_front: LayerNorm + 0.2 - 0.9 us)
    int zl = 0; asm volatile("" : "+v"(zl));
    const uint32_t tag_v = a.epoch[a.par + zl] + 1u, other_v = a.epoch[(a.par ^ 1) + zl];
    __builtin_amdgcn_sched_barrier(0);
    ln_row_mask<NCH1>(xv, S, lane); ln_row_mask<NCH1>(gv, S, lane); ln_row_mask<NCH1>(bv, S, lane);
    ln_row_compute<NCH1>(xv, gv, bv, S, a.eps, lane, av1);
    const unsigned long long tm1 = stamp_t0(sp.base);
    {
        float acc[RIF];
#pragma unroll
        for (int u = 0; u < RIF; ++u) acc[u] = 0.0f;
#pragma unroll
        for (int t = 0; t < NCH1; ++t)                        // (k_gemv1's order: chunk, then row, then element)
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                const __half2 * h = (const __half2 *) &w1[t][u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 f = __half22float2(h[e]);
                    acc[u] = fmaf(f.x, av1[t][2 * e], acc[u]);
                    acc[u] = fmaf(f.y, av1[t][2 * e + 1], acc[u]);
                }
            }
        float v;
#pragma unroll
        for (int u = 0; u < 2; ++u) { const bool hi = lane & 32; const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2]; acc[u] = keep + WMI_SHX(send, 32); }
        { const bool hi = lane & 16; const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1]; v = keep + WMI_SHX(send, 16); }
        v += WMI_SHX(v, 8); v += WMI_SHX(v, 4); v += WMI_SHX(v, 2); v += WMI_SHX(v, 1);
        // lanes 0 / 16 / 32 / 48 hold rows 4 gw + 0 / 1 / 2 / 3: pairs (0, 1) and (2, 3) leave as one granule each
        const uint32_t tag = __builtin_amdgcn_readfirstlane(tag_v);
        if (blockIdx.x == 0 && tid == 0) {
            // the other word holds tag - 2 (or 0 before the first launch) when the launches alternated; tag or more means this parity ran twice in
            // a row — stale granules would pass the tag test below: say so (the step is run again by the host in the two-launch form)
            if (other_v >= tag) __hip_atomic_fetch_or(a.fault, PAIR_FAULT_PARITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.epoch[a.par ^ 1] = tag;
        }
        const __half hv = f2h(gelu16(v + bias1));
        const uint32_t mine = (uint32_t) __half_as_ushort(hv);
        const uint32_t other = (uint32_t) WMI_SHX((int) mine, 16);
        if (a.withhold < 0 && gw == -a.withhold - 1) for (int i = 0; i < 60; ++i) __builtin_amdgcn_s_sleep(127);      // tests: one late producer (~0.2 ms)
        if (writer && !(wrow & 1) && gw + 1 != a.withhold) {
            const unsigned long long g = ((unsigned long long) tag << 32) | (unsigned long long) (mine | (other << 16));
            __hip_atomic_store((unsigned long long *) a.hand + (gw * 2 + (wrow >> 1)), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.h_out && writer) a.h_out[gw * RIF + wrow] = hv;
        // ---- the sweep: thread t takes granules 2 t, 2 t + 1 of every block of 2 NT granules (one 16-byte load each; 2 S granules in all)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        constexpr int NL = WPB == 8 ? NCH1 : 2 * NCH1;       // 16-byte loads per thread and sweep (2 S <= 1024 NCH1 granules, 2 NT per block)
        const int ngran = 2 * S;
        const unsigned long long * src[NL];
#pragma unroll
        for (int b2 = 0; b2 < NL; ++b2) { const int g = b2 * 2 * NT + tid * 2; src[b2] = (const unsigned long long *) a.hand + (g < ngran ? g : 0); }      // (fewer granules than slots: re-read granule 0)
        // (two sweeps in flight, alternating, so that a sweep that leaves just before the granules land does not cost a whole round trip:
        //  measured slower — the hand-off 2.15 -> 2.6 us, the step +6 us: the polling traffic of 128 workgroups doubles)
        // bounded (~1 s): a launch that cannot make progress must not hang the queue.  A consumer that gives up computes its rows from stale
        // granules — and sets PAIR_FAULT_TIMEOUT in the step's status word, which the pick kernel hands to the host in the result's tags: the host
        // runs the step again in the two-launch form (device.cpp: decode_greedy_step).  A sweep that needed many polls only sets PAIR_SLOW.
        const uint32_t spin_cap = a.spin_cap ? a.spin_cap : (1u << 20);
        uint32_t spins = 0; bool landed = false;
        for (; spins < spin_cap; ++spins) {
            u32x4 q[NL];                                     // all requests in flight, one wait (the wait names the registers: nothing reads them before it)
#pragma unroll
            for (int b2 = 0; b2 < NL; ++b2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(q[b2]) : "v"(src[b2]) : "memory");
            if constexpr (NL == 1)      asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]) :: "memory");
            else if constexpr (NL == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]) :: "memory");
            else if constexpr (NL == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]) :: "memory");
            else if constexpr (NL == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) :: "memory");
            else                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[NL - 1]) :: "memory");
            static_assert(NL <= 6, "sweep width");
            bool ok = true;
#pragma unroll
            for (int b2 = 0; b2 < NL; ++b2) {
                const int g = b2 * 2 * NT + tid * 2;
                if (g < ngran) { ok = ok && q[b2][1] == tag && q[b2][3] == tag; hrow[g] = q[b2][0]; hrow[g + 1] = q[b2][2]; }
            }
            if (__all(ok)) { landed = true; break; }
        }
        if (lane == 0 && (!landed || spins > PAIR_SLOW_POLLS))      // (rare: nothing on the ordinary path but the two compares)
            __hip_atomic_fetch_or(a.fault, landed ? PAIR_SLOW : PAIR_FAULT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    const unsigned long long tm2 = stamp_t0(sp.base);
    // ---- phase 2: row gw of W2 against the hidden row
    {
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < NCH2; ++t) {
            const int c = lane * 8 + 512 * t;
            uint4 u4 = *(const uint4 *) (hrow + ((c < K2 ? c : 0) >> 1));
            if (c >= K2) u4 = make_uint4(0u, 0u, 0u, 0u);
            const __half2 * hh = (const __half2 *) &u4; const __half2 * wh = (const __half2 *) &w2[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(wh[e]), x2 = __half22float2(hh[e]);
                acc = fmaf(f.x, x2.x, acc);
                acc = fmaf(f.y, x2.y, acc);
            }
        }
        float v = acc;
        v += WMI_SHX(v, 32); v += WMI_SHX(v, 16); v += WMI_SHX(v, 8); v += WMI_SHX(v, 4); v += WMI_SHX(v, 2); v += WMI_SHX(v, 1);
        if (lane == 0) xio[gw] = (v + bias2) + resid2;
    }
    stamp_end(sp.base, sp.slot, gw, ts0, tm1, tm2);
}

// ------------------------------------------------------------------------------------------------ the front of a decoder layer, one launch
// LayerNorm + q|k|v, the self-attention over the cache and the out projection + residual of the one-row step as ONE launch with two
// hand-offs (profiles/r06o_*: the lab measured 6.7 - 6.9 us per layer against 8.1 - 8.3 for the two launches below it replaces):
//   phase 1 = k_gemv1<4, 1, false, 1, EPI_QKV_DEC> (3 S rows on 3 S / 32 workgroups of eight wavefronts, four rows per wavefront): q to
//             a.q16, k / v into the cache at the step's slot — and every f16 pair as a granule {pair, tag} (k_mlp_pair's form);
//   phase 2 = ONE wavefront per head (wavefront 0 of workgroup h) gathers its head's q, k, v (96 granules), attends over the cache with
//             the new key's k / v taken from the granules (self_attn_body: the routine of the out projection's prologue) and publishes the
//             head's 64 values (32 granules) — once, where the two-launch form recomputes the attention in each of its 32 workgroups;
//   phase 3 = k_gemv1<2, 1, false, 2, EPI_F32_BIAS_RESID, 1, 8>'s projection on S / 16 of the workgroups: the attention row swept once
//             per workgroup (S / 2 granules), weights / bias / residual requested at the start of the launch.
// Per value the operations and their order are the two launches': bit-identical (tests/test_gpu_variants.py).  Caches of <= 64 cells
// (the caller keeps the two launches beyond), S <= 1024 (NCH chunks of 512 columns per row), one head per 64 columns.  Tags, bounded spins and the status word: k_mlp_pair's
// (MlpPairArgs); the phase-3 store overwrites x only after every phase-1 wavefront has consumed it (it cannot gather its row before).
// WPB wavefronts per workgroup: 8 (3 S / 32 workgroups, phase 3 with two rows per wavefront) or 4 (3 S / 16 workgroups, four rows per
// wavefront: k_gemv1<4, ...>'s halving sum, the same bits) — S / 16 workgroups sweep the attention row either way
template <int NCH, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_front(const FrontArgs a_in, const Stamp sp) {
    // lock-step rows: row y is a one-row problem of its own (k_gemv1's convention): shift the per-row operands, everything below is the one-row kernel
    FrontArgs a = a_in;
    {
        const int y = blockIdx.y;
        a.x += (size_t) y * a.S; a.xout += (size_t) y * a.S; a.q16 += (size_t) y * a.S;
        a.ck += (int64_t) y * a.cache_row_stride; a.cv += (int64_t) y * a.cache_row_stride;
        a.kv_head += (size_t) y * a.step_stride; a.n_kv += (size_t) y * a.step_stride;
        a.gq += (size_t) y * 2 * a.S; a.ga += (size_t) y * 2 * a.S;
    }
    __shared__ __attribute__((aligned(16))) __half act[512 * NCH];          // the attention row (phase 3)
    __shared__ __attribute__((aligned(16))) __half hq[3 * 64];        // q, k, v of this workgroup's head (phase 2)
    __shared__ __attribute__((aligned(16))) __half hatt[64];
    __shared__ __attribute__((aligned(16))) float wscr[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wg = blockIdx.x;
    const unsigned long long ts0 = stamp_t0(sp.base);
    const int S = a.S, K = S, H = S >> 6, N1 = 3 * S;
    const int gw = wg * WPB + wave;
    constexpr int R3 = WPB == 8 ? 2 : 4;                    // rows of the out projection per wavefront
    const bool head_wave = wg < H && wave == 0, p3 = wg < (S >> 4);
    constexpr int LPR1 = 16, LPR3 = 64 / R3;
    const int wrow1 = lane / LPR1, wrow3 = lane / LPR3;
    int zl = 0; asm volatile("" : "+v"(zl));

    // ---- every load of the launch that depends on nothing it computes, at the top (DESIGN hazards 23, 35): the head wavefronts' cached
    // keys / values, then in one straight line the row, gain, bias; phase 1's weight rows and epilogue operands; phase 3's; the tag words
    // (the head wavefronts' cached keys / values FIRST and in a branch: a conditional load is harmless to hipcc's wait counts only in front
    //  of — older than — the loads that are waited for early (hazard 23); as unconditional dummy loads on all 384 wavefronts they stood in
    //  the CUs' address pipes in front of the late wavefronts' rows: LayerNorm done 0.3 - 0.5 us later)
    const int g = lane >> 3, o8 = lane & 7;
    uint4 kv[1][8], vv[1][8];
    if (head_wave) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {                        // keys [0, 32) (self_attn_wave's first batch)
            const int j = 8 * t + g, jc = j < a.cap ? j : 0;
            kv[0][t] = *(const uint4 *) (a.ck + (size_t) jc * K + wg * 64 + o8 * 8);
            vv[0][t] = *(const uint4 *) (a.cv + (size_t) jc * K + wg * 64 + o8 * 8);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) { kv[0][t] = make_uint4(0u, 0u, 0u, 0u); vv[0][t] = kv[0][t]; }
    }
    float xv[NCH][8], gv[NCH][8], bv[NCH][8], av[NCH][8];
    ln_row_load<NCH>(a.x, K, lane, xv);
    ln_row_load<NCH>(a.ln_g, K, lane, gv);
    ln_row_load<NCH>(a.ln_b, K, lane, bv);
    __builtin_amdgcn_sched_barrier(0);
    uint4 w1[NCH][4], w3[NCH][R3];
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        const int c = lane * 8 + 512 * t, cc = c < K ? c : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { int o = gw * 4 + u; if (o > N1 - 1) o = N1 - 1; w1[t][u] = *(const uint4 *) (a.Wqkv + (size_t) o * K + cc); }
    }
    float bias1, bias3, resid3; int ro_pre, nkv_pre;
    {
        int n = gw * 4 + wrow1; if (n > N1 - 1) n = N1 - 1;
        bias1 = *(a.bqkv ? a.bqkv + n : (const float *) a.Wqkv);
        ro_pre = a.kv_head[zl]; nkv_pre = a.n_kv[zl];
    }
    const int orow = (p3 ? gw : 0) * R3;
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        const int c = lane * 8 + 512 * t, cc = c < K ? c : 0;
#pragma unroll
        for (int u = 0; u < R3; ++u) w3[t][u] = *(const uint4 *) (a.Wo + (size_t) (orow + u) * K + cc);
    }
    bias3 = *(a.bo ? a.bo + orow + wrow3 : (const float *) a.Wo);
    resid3 = a.x[orow + wrow3];
    // the launch's tag (k_mlp_pair's scheme: epoch[par] + 1, this launch leaves it in epoch[par ^ 1]) — requested LAST and through a lane
    // offset the compiler cannot fold: as a uniform load at the kernel's top it was load -> vmcnt(0) -> readfirstlane, a memory round trip
    // in front of the row's loads (LayerNorm done at + 2.4 us instead of + 1.5)
    const uint32_t tag_v = a.epoch[a.par + zl] + 1u, other_v = a.epoch[(a.par ^ 1) + zl];
    __builtin_amdgcn_sched_barrier(0);

    // ---- phase 1
    ln_row_mask<NCH>(xv, K, lane); ln_row_mask<NCH>(gv, K, lane); ln_row_mask<NCH>(bv, K, lane);
    ln_row_compute<NCH>(xv, gv, bv, K, a.eps, lane, av);
    const unsigned long long tm1 = stamp_t0(sp.base);
    uint32_t tag;
    {
        float acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = 0.0f;
#pragma unroll
        for (int t = 0; t < NCH; ++t)                          // (k_gemv1's order: chunk, then row, then element)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const __half2 * h = (const __half2 *) &w1[t][u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 f = __half22float2(h[e]);
                    acc[u] = fmaf(f.x, av[t][2 * e], acc[u]);
                    acc[u] = fmaf(f.y, av[t][2 * e + 1], acc[u]);
                }
            }
        float v;
#pragma unroll
        for (int u = 0; u < 2; ++u) { const bool hi = lane & 32; const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2]; acc[u] = keep + WMI_SHX(send, 32); }
        { const bool hi = lane & 16; const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1]; v = keep + WMI_SHX(send, 16); }
        v += WMI_SHX(v, 8); v += WMI_SHX(v, 4); v += WMI_SHX(v, 2); v += WMI_SHX(v, 1);
        // lanes 0 / 16 / 32 / 48 hold rows 4 gw + 0 .. 3
        tag = __builtin_amdgcn_readfirstlane(tag_v);
        if (wg == 0 && blockIdx.y == 0 && tid == 0) {
            if (other_v >= tag) __hip_atomic_fetch_or(a.fault, PAIR_FAULT_PARITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.epoch[a.par ^ 1] = tag;
        }
        const int o0 = gw * 4, n = o0 + wrow1;
        const int seg = __builtin_amdgcn_readfirstlane(o0 / S);
        const int c = n - seg * S;
        const float bias = a.bqkv ? bias1 : 0.0f;
        const float val = seg == 0 ? (v + bias) * a.scale : seg == 1 ? v * a.scale : v + bias;
        const __half hv = f2h(val);
        const bool writer = (lane % LPR1) == 0 && n < N1;
        if (writer) {
            __half * dst = seg == 0 ? a.q16 : seg == 1 ? a.ck + (size_t) ro_pre * K : a.cv + (size_t) ro_pre * K;
            dst[c] = hv;
        }
        const uint32_t mine = (uint32_t) __half_as_ushort(hv);
        const uint32_t other = (uint32_t) WMI_SHX((int) mine, 16);
        if (writer && !(wrow1 & 1) && gw + 1 != a.withhold) {
            const unsigned long long gr = ((unsigned long long) tag << 32) | (unsigned long long) (mine | (other << 16));
            __hip_atomic_store(a.gq + (gw * 2 + (wrow1 >> 1)), gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const uint32_t spin_cap = a.spin_cap ? a.spin_cap : (1u << 20);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    unsigned long long tm2 = 0;
    // ---- phase 2: this workgroup's head, on wavefront 0
    if (head_wave) {
        const int h = wg;
        // q_h: granules 32 h .. + 32 of segment 0, k_h / v_h: the same of segments 1 / 2 (S / 2 granules each); lane l < 48 takes two
        const int part = lane >> 4;
        const unsigned long long * src = a.gq + (lane < 48 ? part * (S >> 1) + 32 * h + (lane & 15) * 2 : 32 * h);
        uint32_t spins = 0; bool landed = false; u32x4 q;
        for (; spins < spin_cap; ++spins) {
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(q) : "v"(src) : "memory");
            const bool ok = lane >= 48 || (q[1] == tag && q[3] == tag);
            if (__all(ok)) { landed = true; break; }
        }
        if (lane == 0 && (!landed || spins > PAIR_SLOW_POLLS))
            __hip_atomic_fetch_or(a.fault, landed ? PAIR_SLOW : PAIR_FAULT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < 48) *(uint2 *) (hq + part * 64 + (lane & 15) * 4) = make_uint2(q[0], q[2]);
        const int n_kv = __builtin_amdgcn_readfirstlane(nkv_pre), slot = __builtin_amdgcn_readfirstlane(ro_pre);
        uint4 qv[1];
        qv[0] = *(const uint4 *) (hq + o8 * 8);
        const uint4 knew = *(const uint4 *) (hq + 64 + o8 * 8), vnew = *(const uint4 *) (hq + 128 + o8 * 8);
        const int hs[1] = { h };
        if (n_kv > 32) {
#pragma unroll
            for (int t = 4; t < 8; ++t) {
                const int j = 8 * t + g, jc = j < n_kv ? j : 0;
                kv[0][t] = *(const uint4 *) (a.ck + (size_t) jc * K + h * 64 + o8 * 8);
                vv[0][t] = *(const uint4 *) (a.cv + (size_t) jc * K + h * 64 + o8 * 8);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) if (8 * t + g == slot) { kv[0][t] = knew; vv[0][t] = vnew; }
            self_attn_body<1, 8>(qv, kv, vv, n_kv, hs, H, lane, hatt - h * 64, nullptr, wscr);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) if (8 * t + g == slot) { kv[0][t] = knew; vv[0][t] = vnew; }
#pragma unroll
            for (int t = 4; t < 8; ++t) { kv[0][t] = make_uint4(0u, 0u, 0u, 0u); vv[0][t] = kv[0][t]; }
            self_attn_body<1, 4>(qv, kv, vv, n_kv, hs, H, lane, hatt - h * 64, nullptr, wscr);
        }
        if (lane < 32) {
            const uint32_t pr = *(const uint32_t *) (hatt + 2 * lane);
            const unsigned long long gr = ((unsigned long long) tag << 32) | (unsigned long long) pr;
            __hip_atomic_store(a.ga + (h * 32 + lane), gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        tm2 = stamp_t0(sp.base);
    }
    // ---- phase 3: the attention row, swept once per workgroup; two rows per wavefront
    if (p3) {
        if (tid < (S >> 2)) {
            const unsigned long long * src = a.ga + tid * 2;
            uint32_t spins = 0; bool landed = false; u32x4 q;
            for (; spins < spin_cap; ++spins) {
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(q) : "v"(src) : "memory");
                const bool ok = q[1] == tag && q[3] == tag;
                if (__all(ok)) { landed = true; break; }
            }
            if (lane == 0 && (!landed || spins > PAIR_SLOW_POLLS))
                __hip_atomic_fetch_or(a.fault, landed ? PAIR_SLOW : PAIR_FAULT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *(uint2 *) (act + tid * 4) = make_uint2(q[0], q[2]);
        }
        __syncthreads();
        float a3[NCH][8];
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const int c = lane * 8 + 512 * t;
            uint4 u4 = *(const uint4 *) (act + (c < K ? c : 0));
            if (c >= K) u4 = make_uint4(0u, 0u, 0u, 0u);
            const __half2 * h = (const __half2 *) &u4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); a3[t][2 * e] = f.x; a3[t][2 * e + 1] = f.y; }
        }
        float acc[R3];
#pragma unroll
        for (int u = 0; u < R3; ++u) acc[u] = 0.0f;
#pragma unroll
        for (int t = 0; t < NCH; ++t)
#pragma unroll
            for (int u = 0; u < R3; ++u) {
                const __half2 * h = (const __half2 *) &w3[t][u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 f = __half22float2(h[e]);
                    acc[u] = fmaf(f.x, a3[t][2 * e], acc[u]);
                    acc[u] = fmaf(f.y, a3[t][2 * e + 1], acc[u]);
                }
            }
        float v;
        if constexpr (R3 == 2) {
            { const bool hi = lane & 32; const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1]; v = keep + WMI_SHX(send, 32); }
            v += WMI_SHX(v, 16); v += WMI_SHX(v, 8); v += WMI_SHX(v, 4); v += WMI_SHX(v, 2); v += WMI_SHX(v, 1);
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) { const bool hi = lane & 32; const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2]; acc[u] = keep + WMI_SHX(send, 32); }
            { const bool hi = lane & 16; const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1]; v = keep + WMI_SHX(send, 16); }
            v += WMI_SHX(v, 8); v += WMI_SHX(v, 4); v += WMI_SHX(v, 2); v += WMI_SHX(v, 1);
        }
        if ((lane % LPR3) == 0) {
            const float bias = a.bo ? bias3 : 0.0f;
            a.xout[orow + wrow3] = (v + bias) + resid3;
        }
    }
    stamp_end(sp.base, sp.slot, (int) blockIdx.y * (int) gridDim.x * WPB + gw, ts0, tm1, tm2);
}

// -- the A/B switches of the launch paths: one read of the environment per process (an embedding application may call setenv on its own threads)
static Knobs read_knobs() {
    Knobs kn{};
    kn.no_mlp_pair = getenv("WMI_NO_MLP_PAIR") != nullptr;
    kn.pair_wpb = getenv("WMI_PAIR_WPB") ? atoi(getenv("WMI_PAIR_WPB")) : 4;
    kn.sa_wpb = getenv("WMI_SA_WPB") ? atoi(getenv("WMI_SA_WPB")) : 8;
    kn.gemv1_wide_generic = getenv("WMI_GEMV1_WIDE_GENERIC") != nullptr;
    kn.host_draws = getenv("WMI_HOST_DRAWS") != nullptr;
    kn.debug_sync = getenv("WMI_DEBUG_SYNC") != nullptr;
    kn.pair_withhold = getenv("WMI_PAIR_WITHHOLD") ? atoi(getenv("WMI_PAIR_WITHHOLD")) : 0;        // tests: see MlpPairArgs::withhold
    kn.pair_spin_cap = getenv("WMI_PAIR_SPIN_CAP") ? (uint32_t) strtoul(getenv("WMI_PAIR_SPIN_CAP"), nullptr, 0) : 0u;
    kn.no_front = getenv("WMI_NO_FRONT") != nullptr;           // LN + q|k|v, self-attention + out as two launches (k_front off)
    kn.front_withhold = getenv("WMI_FRONT_WITHHOLD") ? atoi(getenv("WMI_FRONT_WITHHOLD")) : 0;      // tests: see FrontArgs::withhold
    kn.front_wpb = getenv("WMI_FRONT_WPB") ? atoi(getenv("WMI_FRONT_WPB")) : 4;      // wavefronts per workgroup of the one-row k_front (8: A/B)
    kn.no_xback = getenv("WMI_NO_XBACK") != nullptr;           // cross-attention, combine + out projection as two launches (k_xback off)
    kn.xback_withhold = getenv("WMI_XBACK_WITHHOLD") ? atoi(getenv("WMI_XBACK_WITHHOLD")) : 0;      // tests: see XbackArgs::withhold
    return kn;
}
static std::atomic<const Knobs *> g_knobs{nullptr};
const Knobs & knobs() {
    const Knobs * kn = g_knobs.load(std::memory_order_acquire);
    if (kn) return *kn;
    const Knobs * fresh = new Knobs(read_knobs());
    const Knobs * expect = nullptr;
    if (g_knobs.compare_exchange_strong(expect, fresh, std::memory_order_acq_rel)) return *fresh;
    delete fresh;
    return *expect;
}
void reload_knobs() { g_knobs.store(new Knobs(read_knobs()), std::memory_order_release); }      // (the old block is left in place: a reader may still hold it)

// (S, 8-wavefront workgroups?) -> instantiation; the two functions below walk the same table
#define WMI_PAIR_TABLE(S, w8, nch2, X) do { \
        if (w8)             { if ((nch2) <= 3) X(1, 3, 8); else X(1, 4, 8); } \
        else if ((S) <= 512) { if ((nch2) <= 3) X(1, 3, 4); else X(1, 4, 4); } \
        else if ((S) <= 768) X(2, 6, 4); \
        else                X(2, 8, 4); } while (0)

template <int N1, int N2, int W>
static int pair_fit() {
    // workgroups of this instantiation that are resident at once, per device ordinal (an in-process pool holds a context per GPU, and a
    // partitioned device has fewer CUs than the first one seen)
    static std::atomic<int> cache[64];
    int dev = 0;
    (void) hipGetDevice(&dev);
    std::atomic<int> & c = cache[dev & 63];
    int fit = c.load(std::memory_order_relaxed);
    if (fit > 0) return fit;
    int nb = 0, ncu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *) k_mlp_pair<N1, N2, W>, 64 * W, 0) != hipSuccess) return 0;
    (void) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    // (the occupancy query over-reports by one workgroup per CU for SGPR-heavy kernels on this stack: one CU's worth of margin; the
    //  launches here are at most S / 4 + 1 = 257 workgroups against thousands)
    fit = nb > 1 ? (nb - 1) * ncu : nb * ncu;
    c.store(fit, std::memory_order_relaxed);
    return fit;
}

bool mlp_pair_usable(int S, bool with_mirror) {
    // measured, step chain on the GPU with the MLP as one launch / two launches (scratch/pair_ab.py, one process): tiny.en 111.7 / 115.5 us,
    // base.en 155.4 / 158.9, small 384.8 / 406.5 (-5.3 %), medium 873.8 / 890.3, large-v3 (f16) 1563 / 1541 — at S = 1280 the two 13 MB
    // matrices are a bandwidth matter and one row per wavefront streams them worse than the two-launch tiling: two launches there
    if (S > 1024 || (S % 64) != 0) return false;
    const int wpb = knobs().pair_wpb;
    const bool w8 = wpb == 8 && (S % 8) == 0 && S <= 512;
    const int blocks = S / (w8 ? 8 : 4) + (with_mirror ? 1 : 0);
    const int nch2 = (4 * S + 511) / 512;
    int fit = 0;
#define WMI_PAIR_FIT(N1, N2, W) fit = pair_fit<N1, N2, W>()
    WMI_PAIR_TABLE(S, w8, nch2, WMI_PAIR_FIT);
#undef WMI_PAIR_FIT
    return blocks <= fit;
}

void mlp_pair(const MlpPairArgs & a, float * x_inout, hipStream_t st) {
    const int S = a.S;
    // 4 S rows of W1, four per wavefront, 4-wavefront workgroups.  (WMI_PAIR_WPB=8: eight — half as many sweeping workgroups, one 16-byte
    // load per thread and sweep instead of two: measured SLOWER, step chain 157.5 against 155.4 us; two launches 158.9, same process)
    const bool w8 = knobs().pair_wpb == 8 && (S % 8) == 0 && S <= 512;
    const int G = S / (w8 ? 8 : 4);
    const int blocks = G + (a.step_copy_src ? 1 : 0);
    const int nch2 = (4 * S + 511) / 512;
    const Stamp sp = stamp_next();
#define WMI_PAIR_GO(N1, N2, W) hipLaunchKernelGGL((k_mlp_pair<N1, N2, W>), dim3(blocks), dim3(64 * W), 0, st, a, x_inout, G, sp)
    WMI_PAIR_TABLE(S, w8, nch2, WMI_PAIR_GO);
#undef WMI_PAIR_GO
}
#undef WMI_PAIR_TABLE

// wavefronts per workgroup: four for one row (eight-wavefront workgroups get going ~0.7 us later: LayerNorm done + 2.1 - 2.3 us against
// + 1.9 - 2.0, decode 141.2 -> 137.5 us per token), eight for lock-step rows (half as many workgroups to keep resident; measured with eight)
static int front_wpb(int rows) { return rows > 1 ? 8 : (knobs().front_wpb == 8 ? 8 : 4); }
bool front_usable(int S, int rows) {
    // <= two 512-column chunks per row, one head per 64 columns, every workgroup of the launch resident at once (they wait for each other)
    if (S > 1024 || (S % 64) != 0 || S < 128) return false;
    static std::atomic<int> cache[64][2][2];
    int dev = 0; (void) hipGetDevice(&dev);
    const int wide = S > 512 ? 1 : 0, wpb = front_wpb(rows), w8 = wpb == 8 ? 1 : 0;
    int v = cache[dev & 63][wide][w8].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0, nb = 0;
        (void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const void * fn = wpb == 4 ? (wide ? (const void *) k_front<2, 4> : (const void *) k_front<1, 4>) : (wide ? (const void *) k_front<2, 8> : (const void *) k_front<1, 8>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * wpb, 0) != hipSuccess) nb = 0;
        v = 1 + std::max(0, (cus - 1) * std::min(nb, 2));       // workgroups resident at once, one CU left to others (two per CU at most counted)
        cache[dev & 63][wide][w8].store(v, std::memory_order_relaxed);
    }
    return (3 * S / (wpb == 4 ? 16 : 32)) * std::max(rows, 1) <= v - 1;
}
void front(const FrontArgs & a, hipStream_t st) {
    const int rows = a.rows > 1 ? a.rows : 1;
    if (front_wpb(rows) == 4) {
        const dim3 grid(3 * a.S / 16, rows);
        if (a.S <= 512) hipLaunchKernelGGL((k_front<1, 4>), grid, dim3(256), 0, st, a, stamp_next());
        else            hipLaunchKernelGGL((k_front<2, 4>), grid, dim3(256), 0, st, a, stamp_next());
    } else {
        const dim3 grid(3 * a.S / 32, rows);
        if (a.S <= 512) hipLaunchKernelGGL((k_front<1, 8>), grid, dim3(512), 0, st, a, stamp_next());
        else            hipLaunchKernelGGL((k_front<2, 8>), grid, dim3(512), 0, st, a, stamp_next());
    }
}

void gemv(const GemvArgs & a, hipStream_t st) {
    const Stamp sp = stamp_next();
    if (sp.base) { GemvArgs b = a; b.stamps = sp.base; b.stamp_slot = sp.slot; gemv_(b, st); return; }
    gemv_(a, st);
}
static bool rows_on_mfma(const GemvArgs & a) {
    static const bool rows_valu_env = getenv("WMI_ROWS_VALU") != nullptr;
    const bool rows_valu = rows_valu_env || g_rows_valu;
    return a.lanes && a.n >= 2 && a.n <= 16 && !a.sa_q && (!a.comb_o || a.N < 8192) && (a.K % 128) == 0 &&
           (a.epi != EPI_QKV_DEC || (a.S % 16) == 0) && (!rows_valu || a.n > 8);
}
bool gemv_rows_carries_mirror(const GemvArgs & a) { return rows_on_mfma(a) && a.N >= 8192; }
bool gemv_rows_take_self_attention(const GemvArgs & a) {
    static const int rows_y = getenv("WMI_ROWS_Y") ? atoi(getenv("WMI_ROWS_Y")) : 1;
    static const bool off = getenv("WMI_GEMV1_GENERIC") != nullptr;
    const int nch = (a.K + 511) / 512, hpw = (a.K / 64 + 3) / 4;
    const bool inst = (nch == 1 && (hpw == 1 || hpw == 2)) || (nch == 2 && (hpw == 3 || hpw == 4)) || (nch == 3 && (hpw == 5 || hpw == 4));
    return rows_y && !off && a.lanes && a.n >= 2 && a.n <= 16 && a.N < 8192 && a.K <= 2048 && (a.K % 64) == 0 && a.epi == EPI_F32_BIAS_RESID && inst;
}
static void gemv_(const GemvArgs & a, hipStream_t st) {
    // lock-step chunk rows go to the matrix cores (WMI_ROWS_VALU=1 keeps them on the VALU kernel, whose per-row
    // arithmetic is bit-identical to the single-row path: used by the parity tests to pin the control flow)
    // ... except the small projections (everything but the vocabulary): as n one-row problems, row = grid.y of the one-row kernels
    // (k_gemv1).  The matrix-core rows kernel reads every weight once but normalises / gathers all n rows in each of its workgroups and
    // runs 6.3-6.6 us per launch at 8 rows where the one-row kernels take 4.2-4.8; with the row dimension on grid.y the extra weight
    // reads come from the XCD's own L2, and the out projection takes the self-attention in its prologue like the one-row step (one
    // launch fewer per layer).  Per row the arithmetic IS the one-row path's.  WMI_ROWS_Y=0: off (A/B).
    static const int rows_y = getenv("WMI_ROWS_Y") ? atoi(getenv("WMI_ROWS_Y")) : 1;
    if (rows_y && a.lanes && a.n >= 2 && a.n <= 16 && a.N < 8192 && !a.rows && a.K <= 2048 && (a.K % 8) == 0 && (!a.ln_g || a.K <= 1536) && !a.step_copy_src) {
        if (launch_gemv1_special(a, (a.K + 511) / 512, st)) return;
    }
    const bool mfma_ok = rows_on_mfma(a) && !a.sa_q;
    if (mfma_ok) {
        static const bool generic = getenv("WMI_ROWS_GENERIC_EPI") != nullptr;       // A/B knob
        const bool vec = !generic && (a.N % 16) == 0 && (a.ldc % 4) == 0 && (!a.resid || (a.ldr % 4) == 0) &&
                         (a.epi != EPI_QKV_DEC || ((a.ldaux % 4) == 0 && (a.ldaux2 % 4) == 0));
        // vocabulary projection: whole tiles leave as one 16-byte store per lane; the ragged last tile keeps the element-wise form
        if (a.N >= 8192) { if (!generic && a.epi == EPI_LOGITS && (a.ldc % 4) == 0 && !a.bias) launch_rows_mfma<false, EPI_LOGITS>(a, st); else launch_rows_mfma<false>(a, st); }
        else if (vec && a.epi == EPI_QKV_DEC)        launch_rows_mfma<true, EPI_QKV_DEC>(a, st);
        else if (vec && a.epi == EPI_F32_BIAS_RESID) launch_rows_mfma<true, EPI_F32_BIAS_RESID>(a, st);
        else if (vec && a.epi == EPI_F16_BIAS_GELU)  launch_rows_mfma<true, EPI_F16_BIAS_GELU>(a, st);
        else launch_rows_mfma<true>(a, st);
        return;
    }
    static const bool gemv1_off = getenv("WMI_GEMV1_OFF") != nullptr;       // debug / A-B: LDS-staged one-row path
    static const int gemv1_mask = getenv("WMI_GEMV1_MASK") ? atoi(getenv("WMI_GEMV1_MASK")) : 0;   // debug: per-prologue opt-out
    const int kind = a.ln_g ? 1 : a.sa_q ? 2 : a.comb_o ? 4 : 8;
    if (a.n == 1 && !a.lanes && a.K > 2048 && a.K <= 5120 && (a.K % 8) == 0 && !gemv1_off && !(gemv1_mask & kind) && kind == 8 &&
        launch_gemv1_special(a, (a.K + 511) / 512, st)) return;
    if (a.n == 1 && !a.lanes && a.K <= 2048 && (a.K % 8) == 0 && !gemv1_off && !(gemv1_mask & kind) && (!a.ln_g || a.K <= 1536)) {
        const int nch = (a.K + 511) / 512;
        if (launch_gemv1_special(a, nch, st)) return;
        if (a.N >= 16384) {                       // vocabulary projection: 8 rows (8 KB) per wavefront in flight
            if (nch == 1) launch_gemv1<8, 1>(a, st); else if (nch == 2) launch_gemv1<8, 2>(a, st); else if (nch == 3) launch_gemv1<8, 3>(a, st); else launch_gemv1<4, 4>(a, st);
        } else {
            if (nch == 1) launch_gemv1<4, 1>(a, st); else if (nch == 2) launch_gemv1<4, 2>(a, st); else if (nch == 3) launch_gemv1<4, 3>(a, st); else launch_gemv1<4, 4>(a, st);
        }
        return;
    }
    // (the LDS-staged kernels below do not carry the step-record mirror of the chained greedy step: its own tiny launch then)
    if (a.step_copy_src) hipLaunchKernelGGL(k_step_mirror, dim3(1), dim3(64), 0, st, (const int32_t *) a.step_copy_src, (int32_t *) a.step_copy_dst);
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
