// Cross-lane exchanges of a 64-lane wavefront without the LDS crossbar (gfx950).
//
// hipcc lowers every __shfl_xor to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): ~100 cycles of dependent latency per step, 18 steps
// in a LayerNorm + dot-product kernel of the decode step (two 6-step butterflies for mean / variance, one for the dot products)
// — about a third of that kernel's body (scratch/lab/chain_lab.hip).  The forms below move the same lanes with DPP modifiers and
// the gfx950 v_permlane{16,32}_swap instructions: plain VALU issue, no LDS, no lgkmcnt wait.
//
// xor_lane<M>(x) returns x of lane (L ^ M): exactly __shfl_xor(x, M), so reductions written with it add the same pairs in the
// same order and give bit-identical results — kernels that must agree bit for bit (the lock-step "exact" mode against the
// one-row path) can switch independently.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace wmi { namespace k {

#if defined(__HIPCC__)
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_mov(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, ROW_MASK, BANK_MASK, false); }

template <int M> __device__ __forceinline__ int xor_lane_i(int x) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "xor_lane: power of two below 64");
    if constexpr (M == 1) return dpp_mov<0xB1>(x, x);                         // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_mov<0x4E>(x, x);                    // quad_perm [2,3,0,1]
    else if constexpr (M == 4) {
        // banks (4 lanes each) 0 and 2 of a row read 4 lanes up (row_shl:4), banks 1 and 3 read 4 lanes down (row_shr:4)
        int r = dpp_mov<0x104, 0xf, 0x5>(x, x);
        return dpp_mov<0x114, 0xf, 0xA>(r, x);
    }
    else if constexpr (M == 8) return dpp_mov<0x128>(x, x);                   // row_ror:8: lane i of a row reads lane (i - 8) mod 16 = i ^ 8
    else if constexpr (M == 16) {
        // v_permlane16_swap: rows 1 / 3 of the first operand <-> rows 0 / 2 of the second.  Both operands = x:
        // first = [r0 r0 r2 r2], second = [r1 r1 r3 r3]; lane of an even row wants the odd row's value (second), odd row the first
        const auto s = __builtin_amdgcn_permlane16_swap((unsigned) x, (unsigned) x, false, false);
        return (__lane_id() & 16) ? (int) s[0] : (int) s[1];
    } else {
        // v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second.  first = [lo lo], second = [hi hi]
        const auto s = __builtin_amdgcn_permlane32_swap((unsigned) x, (unsigned) x, false, false);
        return (__lane_id() & 32) ? (int) s[0] : (int) s[1];
    }
}
template <int M> __device__ __forceinline__ float xor_lane(float x) { return __int_as_float(xor_lane_i<M>(__float_as_int(x))); }
template <int M> __device__ __forceinline__ int   xor_lane(int x)   { return xor_lane_i<M>(x); }

// 64-lane butterfly sum in the order 32, 16, 8, 4, 2, 1 (the order of `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o)`)
__device__ __forceinline__ float wave_sum_desc(float v) {
    v += xor_lane<32>(v); v += xor_lane<16>(v); v += xor_lane<8>(v); v += xor_lane<4>(v); v += xor_lane<2>(v); v += xor_lane<1>(v);
    return v;
}
__device__ __forceinline__ float wave_max_desc(float v) {
    v = fmaxf(v, xor_lane<32>(v)); v = fmaxf(v, xor_lane<16>(v)); v = fmaxf(v, xor_lane<8>(v));
    v = fmaxf(v, xor_lane<4>(v)); v = fmaxf(v, xor_lane<2>(v)); v = fmaxf(v, xor_lane<1>(v));
    return v;
}

// LDS-DMA issued from inline assembly: hipcc (ROCm 7.2) tracks a __builtin_amdgcn_global_load_lds as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read it cannot prove disjoint — here in front of the fragment reads of EVERY K
// step, i.e. the whole ring was drained right after it had been refilled (ISA dump: vmcnt(0) at the head of the compute block;
// 56 % of the wave cycles in SQ_WAIT_ANY).  An asm statement is invisible to that bookkeeping; the counted waits below are the
// only ones.  M0 = LDS byte address of the wavefront's destination (lane L lands at M0 + L * size); saved and restored because
// the compiler owns M0 (cdna_hip_programming.md §5.7).
template <int BYTES>
__device__ __forceinline__ void glds_asm(const void * gsrc, uint32_t lds_dst) {
    uint32_t keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
    if constexpr (BYTES == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// A run of N consecutive 1 KiB pieces by LDS-DMA from ONE asm statement: source = a wave-uniform 64-bit base (SGPR pair) + a 32-bit
// byte offset per lane and piece, destination = lds_dst, lds_dst + 1 KiB, ...  Against N x glds_asm: no 64-bit VALU address add per
// piece, M0 saved / restored once, three instructions per piece (the DMA, s_add_u32 m0, s_nop) — the issue cost of a tile's pieces
// sits on the LOAD side's critical path in k_gemm8.  N in {1, 2, 3, 4, 6, 8}.  (s_add_u32 writes SCC: declared, or a compare of the surrounding loop is lost.)
#define WMI_GL_(i) "global_load_lds_dwordx4 %[v" #i "], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
#define WMI_GL_HEAD "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
#define WMI_GL_TAIL "s_mov_b32 m0, %[k]"
template <int N>
__device__ __forceinline__ void glds_run(const uint32_t (&v)[N], const void * sbase, uint32_t lds_dst) {
    static_assert(N == 1 || N == 2 || N == 3 || N == 4 || N == 6 || N == 8, "glds_run: piece count");
    uint32_t keep;
    const uint32_t d = __builtin_amdgcn_readfirstlane(lds_dst);
    const uint64_t b64 = (uint64_t) (uintptr_t) sbase;
    const uint64_t b = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (b64 >> 32)) << 32) | (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) b64);
    if constexpr (N == 1)
        asm volatile(WMI_GL_HEAD WMI_GL_(0) WMI_GL_TAIL : [k] "=&s"(keep) : [v0] "v"(v[0]), [d] "s"(d), [b] "s"(b) : "memory", "scc");
    else if constexpr (N == 2)
        asm volatile(WMI_GL_HEAD WMI_GL_(0) WMI_GL_(1) WMI_GL_TAIL : [k] "=&s"(keep) : [v0] "v"(v[0]), [v1] "v"(v[1]), [d] "s"(d), [b] "s"(b) : "memory", "scc");
    else if constexpr (N == 3)
        asm volatile(WMI_GL_HEAD WMI_GL_(0) WMI_GL_(1) WMI_GL_(2) WMI_GL_TAIL : [k] "=&s"(keep) : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [d] "s"(d), [b] "s"(b) : "memory", "scc");
    else if constexpr (N == 4)
        asm volatile(WMI_GL_HEAD WMI_GL_(0) WMI_GL_(1) WMI_GL_(2) WMI_GL_(3) WMI_GL_TAIL : [k] "=&s"(keep)
                     : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [d] "s"(d), [b] "s"(b) : "memory", "scc");
    else if constexpr (N == 6)
        asm volatile(WMI_GL_HEAD WMI_GL_(0) WMI_GL_(1) WMI_GL_(2) WMI_GL_(3) WMI_GL_(4) WMI_GL_(5) WMI_GL_TAIL : [k] "=&s"(keep)
                     : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [d] "s"(d), [b] "s"(b) : "memory", "scc");
    else
        asm volatile(WMI_GL_HEAD WMI_GL_(0) WMI_GL_(1) WMI_GL_(2) WMI_GL_(3) WMI_GL_(4) WMI_GL_(5) WMI_GL_(6) WMI_GL_(7) WMI_GL_TAIL : [k] "=&s"(keep)
                     : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]),
                       [d] "s"(d), [b] "s"(b) : "memory", "scc");
}
#undef WMI_GL_
#undef WMI_GL_HEAD
#undef WMI_GL_TAIL

// The same run with ONE offset register: piece p reads at v0 + p * stride (8 rows further down the same columns), clamped to vclamp
// (CLAMP: the lane's offset in the last valid row — the bottom edge of a matrix whose row count is not a multiple of the tile).  The
// piece offsets are formed in a scratch register right in front of each request (v_add_u32 [+ v_min_u32]): two or three VGPRs per
// operand instead of one per piece — k_gemm8's deferred stores need the registers.  N in 1..8.
#define WMI_GLA_(CL) "v_add_u32 %[t], %[v0], %[acc]\n\t" CL "global_load_lds_dwordx4 %[t], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[acc], %[acc], %[st]\n\t"
#define WMI_GLA_CL "v_min_u32 %[t], %[t], %[vc]\n\t"
#define WMI_GLA_REP1(CL) WMI_GLA_(CL)
#define WMI_GLA_REP2(CL) WMI_GLA_(CL) WMI_GLA_(CL)
#define WMI_GLA_REP3(CL) WMI_GLA_REP2(CL) WMI_GLA_(CL)
#define WMI_GLA_REP4(CL) WMI_GLA_REP2(CL) WMI_GLA_REP2(CL)
#define WMI_GLA_REP5(CL) WMI_GLA_REP4(CL) WMI_GLA_(CL)
#define WMI_GLA_REP6(CL) WMI_GLA_REP4(CL) WMI_GLA_REP2(CL)
#define WMI_GLA_REP7(CL) WMI_GLA_REP4(CL) WMI_GLA_REP3(CL)
#define WMI_GLA_REP8(CL) WMI_GLA_REP4(CL) WMI_GLA_REP4(CL)
#define WMI_GLA_HEAD "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_mov_b32 %[acc], 0\n\t"
#define WMI_GLA_TAIL "s_mov_b32 m0, %[k]"
#define WMI_GLA_CASE(NN)                                                                                                            \
    if constexpr (N == NN) {                                                                                                          \
        if constexpr (CLAMP) asm volatile(WMI_GLA_HEAD WMI_GLA_REP##NN(WMI_GLA_CL) WMI_GLA_TAIL : [k] "=&s"(keep), [acc] "=&s"(acc), [t] "=&v"(t)   \
                                          : [v0] "v"(v0), [vc] "v"(vclamp), [st] "s"(st), [d] "s"(d), [b] "s"(b) : "memory", "scc");  \
        else                 asm volatile(WMI_GLA_HEAD WMI_GLA_REP##NN("") WMI_GLA_TAIL : [k] "=&s"(keep), [acc] "=&s"(acc), [t] "=&v"(t)           \
                                          : [v0] "v"(v0), [st] "s"(st), [d] "s"(d), [b] "s"(b) : "memory", "scc");                    \
    }
template <int N, bool CLAMP>
__device__ __forceinline__ void glds_run_affine(uint32_t v0, uint32_t stride, uint32_t vclamp, const void * sbase, uint32_t lds_dst) {
    static_assert(N >= 1 && N <= 8, "glds_run_affine: piece count");
    uint32_t keep, acc, t;
    const uint32_t d = __builtin_amdgcn_readfirstlane(lds_dst), st = __builtin_amdgcn_readfirstlane(stride);
    const uint64_t b64 = (uint64_t) (uintptr_t) sbase;
    const uint64_t b = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (b64 >> 32)) << 32) | (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) b64);
    WMI_GLA_CASE(1) WMI_GLA_CASE(2) WMI_GLA_CASE(3) WMI_GLA_CASE(4) WMI_GLA_CASE(5) WMI_GLA_CASE(6) WMI_GLA_CASE(7) WMI_GLA_CASE(8)
    (void) vclamp;
}
#undef WMI_GLA_CASE
#undef WMI_GLA_
#undef WMI_GLA_CL
#undef WMI_GLA_HEAD
#undef WMI_GLA_TAIL
__device__ __forceinline__ uint32_t lds_addr(const void * p) { return (uint32_t) (uintptr_t) (__attribute__((address_space(3))) const void *) p; }

// xor_lane with the mask as a value: inside a fully unrolled `for (o = 32; o > 0; o >>= 1)` the switch folds to the one DPP form
// (if the loop is not unrolled the switch stays a scalar branch: still correct).  WMI_NO_DPP keeps __shfl_xor (A/B builds).
#if defined(WMI_NO_DPP)
#define WMI_SHX(x, m) __shfl_xor((x), (m))
#else
template <typename T> __device__ __forceinline__ T xor_lane_dyn32(T x, int m) {
    switch (m) {
        case 1:  return xor_lane<1>(x);
        case 2:  return xor_lane<2>(x);
        case 4:  return xor_lane<4>(x);
        case 8:  return xor_lane<8>(x);
        case 16: return xor_lane<16>(x);
        case 32: return xor_lane<32>(x);
        default: return __shfl_xor(x, m);
    }
}
__device__ __forceinline__ float  xor_lane_dyn(float x, int m) { return xor_lane_dyn32<float>(x, m); }
__device__ __forceinline__ int    xor_lane_dyn(int x, int m)   { return xor_lane_dyn32<int>(x, m); }
__device__ __forceinline__ double xor_lane_dyn(double x, int m) {
    const long long b = __double_as_longlong(x);
    const int lo = xor_lane_dyn32<int>((int) (b & 0xffffffffll), m), hi = xor_lane_dyn32<int>((int) (b >> 32), m);
    return __longlong_as_double(((long long) hi << 32) | (long long) (unsigned) lo);
}
#define WMI_SHX(x, m) ::wmi::k::xor_lane_dyn((x), (m))
#endif
#endif

}} // namespace wmi::k
