// f16 x f16 -> f32 MFMA GEMM with fused epilogues for gfx950.
//
//   C[M][N] = A[M][K] . W[N][K]^T       A: activations (f16, row stride lda)
//                                        W: weights     (f16, row stride ldw)  — both K-contiguous
//
// Numerics follow the reference's mul_mat contract (SURVEY App. B rule 1): both operands are IEEE
// f16, products are exact, accumulation is f32 — v_mfma_f32_16x16x32_f16.  Everything the reference
// does as separate graph nodes after a mul_mat (bias add, GELU through the f16 table, residual add,
// q/k scaling, f32->f16 copies into K/V layouts) happens in the epilogue on the accumulator registers.
//
// Tiling: BM x BN x 64 per workgroup, 256 threads = 4 wavefronts (64 lanes) in a 2 x 2 grid, each
// wavefront owning a (BM/2) x (BN/2) sub-tile as (BM/32) x (BN/32) MFMA fragments.  Operand tiles are
// staged global -> VGPR -> LDS (16 B per lane, coalesced 128 B rows) into a double-buffered,
// XOR-swizzled LDS image so that the ds_read_b128 fragment reads are conflict-free; the next tile's
// global loads are in flight while the current tile is multiplied (one barrier per K step).
// The conv front-end reuses this kernel as an implicit GEMM: a token-major activation buffer with
// lda < K makes consecutive A rows overlap, which is exactly im2col for a k=3 convolution.

#include "kernels.h"
#include "wave_ops.h"

#include <cstdlib>
#include <type_traits>

namespace wmi { namespace k {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float    floatx4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;                 // K extent of one LDS tile (two MFMA k-steps)

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

__device__ __forceinline__ uint32_t lds_off(int row, int chunk) {      // byte offset inside a [rows][64] f16 tile
    return (uint32_t) (row * 128 + ((chunk ^ (row & 7)) << 4));
}

// NST = depth of the LDS ring of the global_load_lds path: 2 where several workgroups share a CU and hide each other's loads
// (the ring costs LDS, i.e. occupancy: mlp.0 at one chunk, 768 tiles, is 20 % slower with 4), 4 where a workgroup is alone
// NW wavefronts as a 2 x (NW / 2) grid: 4 (2 x 2, 256 threads) everywhere but the 128 x 256 tile of the big grids (2 x 4, 512 threads:
// one workgroup per CU with a three-deep 48 KB ring = 96 KB in flight per CU against 2 x 32 KB for two 128 x 128 workgroups, and
// 25 % fewer operand bytes through L2 -> LDS)
template <int BM, int BN, int EPI, int NST = 2, int NW = 4>
__global__ __launch_bounds__(NW * 64) void k_gemm(const GemmArgs a) {
    constexpr int WN_ = NW / 2;                        // wavefronts along N (2 along M)
    constexpr int FM = BM / 32, FN = BN / (WN_ * 16);  // fragments per wavefront
    constexpr int LA = BM / (NW * 8), LB = BN / (NW * 8);   // 16-byte loads per thread per tile (register-staged path)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sA = [&](int buf) -> unsigned char * { return smem + buf * ((BM + BN) * 128); };
    auto sB = [&](int buf) -> unsigned char * { return smem + buf * ((BM + BN) * 128) + BM * 128; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN_, wn = wave % WN_;
    const unsigned long long pt0 = a.probe ? wall_clock64() : 0ull;
    unsigned long long pt1 = 0ull, pt2 = 0ull;

    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so give each
    // XCD a contiguous run of tiles that share A panels in its private L2.
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN, nwg = ntm * ntn;
    int wg = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = wg / ntn, tn = wg % ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    // per-thread staging coordinates: pass p covers rows p*(NW*8) + tid/8, chunk tid%8
    const int srow = tid >> 3, schunk = tid & 7;
    const __half * gA[LA]; const __half * gB[LB];
#pragma unroll
    for (int p = 0; p < LA; ++p) {
        int r = m0 + p * (NW * 8) + srow; if (r > a.M - 1) r = a.M - 1;
        gA[p] = a.A + (size_t) r * a.lda + schunk * 8;
    }
#pragma unroll
    for (int p = 0; p < LB; ++p) {
        int r = n0 + p * (NW * 8) + srow; if (r > a.N - 1) r = a.N - 1;
        gB[p] = a.W + (size_t) r * a.ldw + schunk * 8;
    }

    floatx4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = (a.K + BK - 1) / BK;
    const int ktail = a.K - (nk - 1) * BK;             // 32 or 64 (K is a multiple of 32)
    uint4 ra[LA], rb[LB];

    auto load_tile = [&](int kt) {
        // a tile whose upper 32 columns lie beyond K (K % 64 == 32) must not be read past the weight row
        const bool half_only = (kt == nk - 1) && (ktail < BK) && (schunk >= 4);
#pragma unroll
        for (int p = 0; p < LA; ++p) ra[p] = half_only ? uint4{0, 0, 0, 0} : *(const uint4 *) (gA[p] + kt * BK);
#pragma unroll
        for (int p = 0; p < LB; ++p) rb[p] = half_only ? uint4{0, 0, 0, 0} : *(const uint4 *) (gB[p] + kt * BK);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int p = 0; p < LA; ++p) *(uint4 *) (sA(buf) + lds_off(p * (NW * 8) + srow, schunk)) = ra[p];
#pragma unroll
        for (int p = 0; p < LB; ++p) *(uint4 *) (sB(buf) + lds_off(p * (NW * 8) + srow, schunk)) = rb[p];
    };

    const int frow = lane & 15, fq = lane >> 4;
    // Orientation of the accumulator fragments.  mfma(A rows, W rows) leaves a lane with four consecutive ROWS of one output
    // column: row-major f16 results leave as 64 two-byte stores per lane, and those stores — not the K loop — were the larger part
    // of a K = 512 tile (per-tile time fitted over K: ~7 us fixed against 0.78 us per K step at M = 12 000).  mfma(W rows, A rows)
    // computes the same dot products (same operands, same k order) into the transposed fragment: four consecutive COLUMNS of one
    // row per lane = one 8-byte (f16) or 16-byte (f32) store.  The V^T third of the encoder's q|k|v wants consecutive rows (time
    // steps) and keeps the first orientation; a tile never straddles the segments (BN | S).
#ifdef WMI_GEMM_NO_SWAP
    constexpr bool SWAP = false;
#else
    constexpr bool SWAP = EPI != EPI_QKV_DEC;
#endif
    // Measured per epilogue and grid (profiles/r03b_gemm_orientation_per_kernel.txt, rocprof averages, both orientations on the same
    // ring code): the transposed form wins at one chunk everywhere (q|k|v 9.9 against 11.0 us, GELU 10.0 / 10.9, cross K/V 25 / 33,
    // conv 13.1 / 14.3) and at M = 12 000 for GELU (44.9 / 49.6) and cross K/V (142 / 150), but LOSES there for q|k|v (50.3 against
    // 43.7 us) and the residual epilogues (44.6 / 43.2 on 128 x 128 tiles, 23.9 / 21.6 on 64 x 64): those keep the first orientation
    // on the big grids.  Workgroup-uniform, both epilogues are compiled.
    const bool big = a.M >= 4096;
    const bool swap = SWAP && !(EPI == EPI_QKV_ENC && (n0 >= 2 * a.S || big)) && !(EPI == EPI_F32_BIAS_RESID && big);
    auto compute = [&](int buf, auto sw_tag) {
        constexpr bool SWF = decltype(sw_tag)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
                fa[i] = *(const half8 *) (sA(buf) + lds_off(wm * (BM / 2) + i * 16 + frow, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < FN; ++j)
                fb[j] = *(const half8 *) (sB(buf) + lds_off(wn * (BN / WN_) + j * 16 + frow, kk * 4 + fq));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (SWF) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    else               acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    // The orientation is decided OUTSIDE the K loop (two copies of the loop where it is a run-time choice): with the choice inside,
    // hipcc kept both MFMA forms and both epilogues' state live through the loop — the residual GEMM at M = 12 000 went from 44 to
    // 85 us, q|k|v from 44 to 50.
    auto k_loops = [&](auto sw_tag) {
    // 64x64 tiles take the same path while the grid is small (one chunk: latency-bound, -11..15 % per GEMM); with
    // thousands of small tiles the register-staged loop is the faster one (measured, scratch/lab/gemm_lab.hip)
    if ((BM >= 96 || nwg <= 1024) && (a.K % BK) == 0 && !(a.no_glds & 1)) {
        // Large tiles (batched encoder, cross K/V): operands go global -> LDS directly (global_load_lds, 16 B per lane,
        // 1 KiB per wave instruction, no staging VGPRs or ds_write pass).  LDS is written lane-linearly, so the XOR
        // swizzle is applied to each lane's GLOBAL address instead: position p = row*8 + (chunk ^ (row & 7)) of a
        // piece of 8 rows is fetched by lane p.  One barrier per K step: tile kt+1 is in flight while kt is multiplied.
        // (profiles/: +38 % on the M = 12 000 encoder GEMMs over the register-staged loop)
        constexpr int PA = BM / (NW * 8), PB = BN / (NW * 8); // 1 KiB pieces per wavefront per operand
        const int prow = lane >> 3;
        const __half * qA[PA]; const __half * qB[PB];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int lrow = (wave * PA + p) * 8 + prow, pch = (lane & 7) ^ (lrow & 7);
            int r = m0 + lrow; if (r > a.M - 1) r = a.M - 1;
            qA[p] = a.A + (size_t) r * a.lda + pch * 8;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int lrow = (wave * PB + p) * 8 + prow, pch = (lane & 7) ^ (lrow & 7);
            int r = n0 + lrow; if (r > a.N - 1) r = a.N - 1;
            qB[p] = a.W + (size_t) r * a.ldw + pch * 8;
        }
        // DMA from inline asm + a raw s_barrier: hipcc tracks __builtin_amdgcn_global_load_lds as a memory operation and puts
        // `s_waitcnt vmcnt(0)` in front of every __syncthreads() — the ring was drained at each K step whatever its depth (ISA dump;
        // "a 3-deep ring measured the same as 2").  The counted waits below are now the only ones.
        const uint32_t lds0 = lds_addr(smem);
        auto issue = [&](int kt, int buf) {
#pragma unroll
            for (int p = 0; p < PA; ++p)
                glds_asm<16>(qA[p] + kt * BK, lds0 + buf * ((BM + BN) * 128) + (wave * PA + p) * 1024);
#pragma unroll
            for (int p = 0; p < PB; ++p)
                glds_asm<16>(qB[p] + kt * BK, lds0 + buf * ((BM + BN) * 128) + BM * 128 + (wave * PB + p) * 1024);
        };
        // NST-deep ring: NST - 1 tiles are in flight while one is multiplied.  At one chunk a workgroup has its CU (almost) to
        // itself and nothing else hides the ~0.5 us a tile takes to arrive: with a distance of one every K step cost a full
        // load latency (mlp.2, 32 steps: 18 us).  The wait is counted: tile kt has landed when at most (NST - 2) tiles' worth of
        // this wavefront's loads are still outstanding.
        constexpr int LPT = PA + PB;                          // load instructions per wavefront and tile
#pragma unroll
        for (int s0 = 0; s0 < NST - 1; ++s0) if (s0 < nk) issue(s0, s0);
        for (int kt = 0; kt < nk; ++kt) {
            if (nk - 1 - kt >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * LPT) : "memory");
            else                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // tile kt landed for everyone; the buffer multiplied last step is free
            asm volatile("" ::: "memory");
            if (a.probe && kt == 0) pt1 = wall_clock64();
            if (kt + NST - 1 < nk) issue(kt + NST - 1, (kt + NST - 1) % NST);
            compute(kt % NST, sw_tag);
        }
        if (a.probe) { asm volatile("s_nop 0" ::: "memory"); pt2 = wall_clock64(); }
    } else {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_tile(kt + 1);
            compute(buf, sw_tag);
            if (kt + 1 < nk) store_tile(buf ^ 1);
            __syncthreads();
        }
    }
    };
    constexpr bool RT_ORIENT = SWAP && (EPI == EPI_QKV_ENC || EPI == EPI_F32_BIAS_RESID);    // epilogues whose orientation depends on the tile / grid
    if constexpr (!SWAP) k_loops(std::false_type{});
    else if constexpr (!RT_ORIENT) k_loops(std::true_type{});
    else { if (swap) k_loops(std::true_type{}); else k_loops(std::false_type{}); }

    // ------------------------------------------------------------------ epilogue
    // fragment (i, j): rows m = mb + i*16 + fq*4 + r (r = 0..3), column n = nb + j*16 + frow.
    // Interior tiles take an instantiation without bounds checks: a per-element `if (m < M)` makes every store its own
    // basic block, and hipcc then waits vmcnt(0) before each one (vmcnt also counts stores on gfx9-family parts), i.e.
    // the 64 stores of a lane complete one after the other (profiles/: -10..30 % kernel time on the M = 12 000 GEMMs).
    const int mb = m0 + wm * (BM / 2), nb = n0 + wn * (BN / WN_);
    auto epilogue = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
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
                        rpre[i][r] = a.resid[(size_t) ((GUARD && m >= a.M) ? a.M - 1 : m) * a.ldr + n];
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
                        if (a.aux) ((float *) a.aux)[(size_t) m * a.ldaux + n] = g;
                        ((float *) a.C)[(size_t) m * a.ldc + n] = rpre[i][r] + g;
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
    };
    // swapped orientation: fragment (i, j) holds row m = mb + i*16 + frow, columns n = nb + j*16 + fq*4 + r (r = 0..3)
    auto epilogue_sw = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
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
                    rpre[i] = *(const float4 *) (a.resid + (size_t) ((GUARD && m >= a.M) ? a.M - 1 : m) * a.ldr + n);
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
                    if (a.aux) *(float4 *) ((float *) a.aux + (size_t) m * a.ldaux + n) = g;
                    o.x = rpre[i].x + g.x; o.y = rpre[i].y + g.y; o.z = rpre[i].z + g.z; o.w = rpre[i].w + g.w;
                    *(float4 *) ((float *) a.C + (size_t) m * a.ldc + n) = o;
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
    };
    const bool interior = m0 + BM <= a.M && n0 + BN <= a.N && !(a.no_glds & 2);
    if (SWAP && swap) { if (interior) epilogue_sw(std::false_type{}); else epilogue_sw(std::true_type{}); }
    else              { if (interior) epilogue(std::false_type{});    else epilogue(std::true_type{}); }
    if (a.probe && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the probe's "done" includes the stores leaving the wavefront
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned long long * o = a.probe + (size_t) blockIdx.x * 5;
        o[0] = pt0; o[1] = pt1; o[2] = pt2; o[3] = wall_clock64(); o[4] = hwid;
    }
}

template <int BM, int BN, int EPI, int NST, int NW = 4>
void launch_n(const GemmArgs & a, hipStream_t st) {
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    const size_t smem = NST * (size_t) (BM + BN) * 128;
    static std::atomic<uint64_t> lds_ok{0};
    allow_full_lds((const void *) k_gemm<BM, BN, EPI, NST, NW>, lds_ok);
    hipLaunchKernelGGL((k_gemm<BM, BN, EPI, NST, NW>), dim3(ntm * ntn), dim3(NW * 64), smem, st, a);
}
template <int BM, int BN, int EPI>
void launch(const GemmArgs & a, hipStream_t st) {
    static const bool shallow = getenv("WMI_GEMM_RING2") != nullptr;         // debug / A-B
    const long nwg = (long) ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    // (a 3-deep ring for the grids in between — q|k|v 576, mlp.0 768 tiles — measured the same as 2)
    if (BM < 128 && nwg <= 400 && !shallow) launch_n<BM, BN, EPI, 4>(a, st);
    else                                    launch_n<BM, BN, EPI, 2>(a, st);
}

template <int EPI>
void dispatch(const GemmArgs & a, hipStream_t st) {
    // 256 CUs: prefer the 128x128 tile only when it still yields >= ~1.5 waves of workgroups
    const long t128 = (long) ((a.M + 127) / 128) * ((a.N + 127) / 128);
    static const bool no_narrow = getenv("WMI_GEMM_NO_NARROW") != nullptr;  // debug / A-B
    const long t64 = (long) ((a.M + 63) / 64) * ((a.N + 63) / 64);
    // WMI_GEMM_WIDE=1 (A/B knob, off): 128 x 256 tiles on eight wavefronts with a three-deep ring for the big grids (lock-step encoder,
    // M = chunks x T).  Measured (profiles/r03b_gemm_wide_tile_and_gelu.txt): mlp.0 x 8 45.8 us against 43.5 us for two co-resident
    // 128 x 128 workgroups per CU — its K loop takes 7.9 us per 128 x 256 tile against 7.3 us per PAIR of 128 x 128 tiles, so the loop is
    // not waiting for operands to arrive (twice the bytes in flight changed nothing); it is bound by LDS traffic per MFMA (16 fragment
    // reads per 32 MFMAs of a wavefront + the DMA writes: ~900 LDS cycles per 1024 MFMA cycles of a SIMD).
    static const int wide = getenv("WMI_GEMM_WIDE") ? atoi(getenv("WMI_GEMM_WIDE")) : 0;
    const long t256 = (long) ((a.M + 127) / 128) * (a.N / 256);
    if constexpr (EPI == EPI_F16_BIAS_GELU || EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV || EPI == EPI_F16_BIAS) {
        if (wide && (a.N % 256) == 0 && t256 >= 384 && (a.K % BK) == 0 && !(a.no_glds & 1)) { launch_n<128, 256, EPI, 3, 8>(a, st); return; }
    }
    // WMI_GEMM_TALL (A/B knob, off): wave tiles of 128 x 64 instead of 64 x 64 on the big grids.  Per k the fragment reads of a wavefront are
    // (TM + TN) x 2 bytes for 2 TM TN flops: 64 x 64 wave tiles read 1 KB of LDS per 16-cycle MFMA quartet, which together with the DMA writes
    // is ~1.5x the LDS cycles of the MFMA cycles they feed; 128 x 64 cuts the reads per flop by a quarter.
    //   1: 256 x 128 tiles, four wavefronts, two-deep ring (96 KB)   2: the same, three-deep (144 KB)   3: 256 x 256, eight wavefronts (128 KB)
    static const int tall = getenv("WMI_GEMM_TALL") ? atoi(getenv("WMI_GEMM_TALL")) : 0;
    if constexpr (EPI == EPI_F16_BIAS_GELU) {
        if (tall && a.M >= 4096 && (a.K % BK) == 0 && (a.N % 256) == 0 && !(a.no_glds & 1)) {
            if (tall == 1) { launch_n<256, 128, EPI, 2, 4>(a, st); return; }
            if (tall == 2) { launch_n<256, 128, EPI, 3, 4>(a, st); return; }
            if (tall == 3) { launch_n<256, 256, EPI, 2, 8>(a, st); return; }
        }
    }
    static const long t128_min = getenv("WMI_GEMM_T128") ? atol(getenv("WMI_GEMM_T128")) : 320;        // A/B knob; 376 tiles (out projection at M = 12 000): 18.6 us against 23.0 us as 1 504 tiles of 64 x 64
    if (t128 >= t128_min || (t128 >= 256 && a.K >= 1024)) {
        // Round quantisation on the big grids: two workgroups per CU = 512 resident tiles; q|k|v at M = 12 000 is 1 128 tiles of 128 rows
        // (2.2 rounds: the third runs 20 % full), the N = S projections 376 (one round, 73 % full).  Tiles of 96 rows make that 1 500 and
        // 500: whole rounds.  Taken when they fill the rounds better by more than their ~5 % lower operand reuse costs.
        if constexpr (EPI == EPI_QKV_ENC || EPI == EPI_F32_BIAS_RESID) {
            static const bool no96 = getenv("WMI_GEMM_NO_96") != nullptr;          // A/B knob
            const long t96 = (long) ((a.M + 95) / 96) * ((a.N + 127) / 128);
            auto fill = [](long t) { return (double) t / (double) (((t + 511) / 512) * 512); };
            if (!no96 && a.M >= 4096 && (a.K % BK) == 0 && fill(t96) > fill(t128) + 0.08) { launch_n<96, 128, EPI, 2>(a, st); return; }
        }
        launch<128, 128, EPI>(a, st);
    }
    else if constexpr (EPI == EPI_F32_BIAS_RESID) {
        // The N = S projections of the widest models at one chunk (large-v3: 1500 x 1280, K = 1280 / 5120): 240 tiles of 64 x 128 on a four-deep ring
        // instead of 480 of 64 x 64 — a quarter fewer LDS fragment reads per flop.  Measured on the large-v3 q5_1 encoder (64 of these GEMMs):
        // 6.85 -> 6.70 ms; 128 x 64 tiles 6.83, 64 x 128 on a two-deep ring 6.99 (profiles/r03c_gemm_ns_tile.txt).  WMI_GEMM_NS_TILE=0: off.
        static const int ns_tile = getenv("WMI_GEMM_NS_TILE") ? atoi(getenv("WMI_GEMM_NS_TILE")) : 1;
        if (ns_tile && a.K >= 1024 && (a.N % 128) == 0 && (a.K % BK) == 0 && t64 >= 400 && !(a.no_glds & 1)) {
            if (ns_tile == 2) { launch_n<128, 64, EPI, 4>(a, st); return; }
            launch_n<64, 128, EPI, 4>(a, st); return;
        }
        // one chunk, N = S: 64x64 tiles give fewer workgroups than CUs (192 for base.en) and each walks K alone with nothing to
        // overlap its loads; 64x32 tiles double the workgroups
        if (t64 < 256 && !no_narrow) launch<64, 32, EPI>(a, st); else launch<64, 64, EPI>(a, st);
    }
    else             launch<64, 64, EPI>(a, st);
}

} // namespace

void gemm(int epi, const GemmArgs & a_in, hipStream_t st) {
    static const bool no_glds = getenv("WMI_GEMM_NO_GLDS") != nullptr;       // debug / A-B: register-staged loop for every tile size
    static const bool guard_all = getenv("WMI_GEMM_GUARD_ALL") != nullptr;   // debug / A-B: bounds-checked epilogue for every tile
    GemmArgs a = a_in; a.no_glds = (no_glds ? 1 : 0) | (guard_all ? 2 : 0);
    switch (epi) {
        case EPI_F16_BIAS:       dispatch<EPI_F16_BIAS>(a, st); break;
        case EPI_F16_BIAS_GELU:  dispatch<EPI_F16_BIAS_GELU>(a, st); break;
        case EPI_F32_BIAS_RESID: dispatch<EPI_F32_BIAS_RESID>(a, st); break;
        case EPI_CONV2:          dispatch<EPI_CONV2>(a, st); break;
        case EPI_QKV_ENC:        dispatch<EPI_QKV_ENC>(a, st); break;
        case EPI_QKV_DEC:        dispatch<EPI_QKV_DEC>(a, st); break;
        case EPI_CROSS_KV:       dispatch<EPI_CROSS_KV>(a, st); break;
        case EPI_Q_SCALED:       dispatch<EPI_Q_SCALED>(a, st); break;
        default: break;
    }
}

}} // namespace wmi::k
