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
#include "gemm_epi.h"

#include <cstdlib>
#include <type_traits>

namespace wmi { namespace k {

namespace {

using namespace gemm_detail;

constexpr int BK = 64;                 // K extent of one LDS tile (two MFMA k-steps)

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
    // (round 5, with the full-line stores of epilogue_cols_wide: the q and k thirds in the transposed form on the big grids measure the
    //  same as the first orientation — 41.3-41.8 against 40.9-41.2 us, profiles/r05b_gemm_lab_qkv.txt: the V^T third's scattered 8-byte
    //  stores are what the q|k|v epilogue costs over a plain one (34.5 us) — so the round-3 choice stays; a.no_glds bit 8 =
    //  WMI_GEMM_QKV_SWAP is the A/B knob)
    const bool swap = SWAP && !(EPI == EPI_QKV_ENC && (n0 >= 2 * a.S || (big && !(a.no_glds & 8)))) && !(EPI == EPI_F32_BIAS_RESID && big && !(a.no_glds & 32));
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

    // ------------------------------------------------------------------ epilogue (gemm_epi.h)
    const int mb = m0 + wm * (BM / 2), nb = n0 + wn * (BN / WN_);
    const bool interior = m0 + BM <= a.M && n0 + BN <= a.N && !(a.no_glds & 2);
    // full-line stores (gemm_epi.h: epilogue_cols_wide) wherever the wave tile has an even number of fragment columns and the epilogue is one
    // of the row-major ones; a.no_glds bit 2 (WMI_GEMM_NARROW_STORES, A/B) keeps the 8-byte form
    constexpr bool WIDE_OK = FN % 2 == 0 && (EPI == EPI_F16_BIAS || EPI == EPI_F16_BIAS_GELU || EPI == EPI_Q_SCALED || EPI == EPI_F32_BIAS_RESID ||
                                             EPI == EPI_CROSS_KV || EPI == EPI_QKV_ENC);
    if (SWAP && swap) {
        if constexpr (WIDE_OK) {
            if (!(a.no_glds & 4)) {
                if (interior) epilogue_cols_wide<EPI, FM, FN, false>(a, acc, mb, nb, n0, lane); else epilogue_cols_wide<EPI, FM, FN, true>(a, acc, mb, nb, n0, lane);
            } else {
                if (interior) epilogue_cols<EPI, FM, FN, false>(a, acc, mb, nb, n0, lane); else epilogue_cols<EPI, FM, FN, true>(a, acc, mb, nb, n0, lane);
            }
        } else {
            if (interior) epilogue_cols<EPI, FM, FN, false>(a, acc, mb, nb, n0, lane); else epilogue_cols<EPI, FM, FN, true>(a, acc, mb, nb, n0, lane);
        }
    }
    else              { if (interior) epilogue_rows<EPI, FM, FN, false>(a, acc, mb, nb, lane);     else epilogue_rows<EPI, FM, FN, true>(a, acc, mb, nb, lane); }
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
    // Round 6: the N = S projections of the big grids (out, mlp.2 at M = chunks x 1500) on 192 x 128 tiles, ONE workgroup of eight wavefronts per CU
    // on a three-deep ring: 252 tiles at 8 chunks = one round, 76.8 flop per operand byte against 54.9 for two co-resident 96 x 128 workgroups.
    // Measured at 8 chunks (profiles/r06e_*): mlp.2 40.3 -> 34.2 us, out 15.5 -> 13.8; 16 chunks mlp.2 75.6 -> 72.2; 4 chunks mlp.2 34.3 -> 28.2 but
    // out 9.2 -> 10.6 (128 tiles: half the chip) — so the short-K projection takes it only with >= 200 tiles.  Ring depth: two 40.0, three 35.8, four
    // (all 160 KB of LDS) 33.4 - 34.8 us for mlp.2; four wavefronts 42.1; the conv front-end on this tile 33.3 - 33.9 against 33.4 - 34.7 us (stays).
    //   WMI_GEMM_NS192: 0 = off   1 = eight wavefronts, three-deep   2 = two-deep   3 = four wavefronts   4 = four-deep (default)
    static const int ns192 = getenv("WMI_GEMM_NS192") ? atoi(getenv("WMI_GEMM_NS192")) : 4;
    if constexpr (EPI == EPI_F32_BIAS_RESID) {
        const long t192 = (long) ((a.M + 191) / 192) * (a.N / 128);
        if (ns192 && a.M >= 4096 && a.N <= 1024 && (a.N % 128) == 0 && (a.K % BK) == 0 && !(a.no_glds & 1) && (a.K >= 1024 || t192 >= 200)) {
            if (ns192 == 1) { launch_n<192, 128, EPI, 3, 8>(a, st); return; }
            if (ns192 == 2) { launch_n<192, 128, EPI, 2, 8>(a, st); return; }
            if (ns192 == 3) { launch_n<192, 128, EPI, 3, 4>(a, st); return; }
            if (ns192 == 4) { launch_n<192, 128, EPI, 4, 8>(a, st); return; }
        }
    }
    static const long t128_min = getenv("WMI_GEMM_T128") ? atol(getenv("WMI_GEMM_T128")) : 320;        // A/B knob; 376 tiles (out projection at M = 12 000): 18.6 us against 23.0 us as 1 504 tiles of 64 x 64
    if (t128 >= t128_min || (t128 >= 256 && a.K >= 1024)) {
        // Round quantisation on the big grids: two workgroups per CU = 512 resident tiles; q|k|v at M = 12 000 is 1 128 tiles of 128 rows
        // (2.2 rounds: the third runs 20 % full), the N = S projections 376 (one round, 73 % full).  Tiles of 96 rows make that 1 500 and
        // 500: whole rounds.  Taken when they fill the rounds better by more than their ~5 % lower operand reuse costs.
        // (round 6: the conv front-end's big grids on 96-row tiles too — conv2 x8 is 376 tiles of 128 rows = 0.73 of the 512 resident slots, 504 of 96
        //  rows fill them: 38.5 -> 33.4 us, conv1 18.6 -> 17.6; WMI_GEMM_CONV96=0: off)
        static const bool conv96 = getenv("WMI_GEMM_CONV96") ? atoi(getenv("WMI_GEMM_CONV96")) != 0 : true;
        if constexpr (EPI == EPI_QKV_ENC || EPI == EPI_F32_BIAS_RESID || EPI == EPI_CONV2 || EPI == EPI_F16_BIAS_GELU) {
            if (!conv96 && (EPI == EPI_CONV2 || EPI == EPI_F16_BIAS_GELU)) { launch<128, 128, EPI>(a, st); return; }
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

static thread_local GemmLog * tl_gemm_log = nullptr;
void gemm_log_install(GemmLog * log) { tl_gemm_log = log; }

void gemm(int epi, const GemmArgs & a_in, hipStream_t st) {
    static const bool no_glds = getenv("WMI_GEMM_NO_GLDS") != nullptr;       // debug / A-B: register-staged loop for every tile size
    static const bool guard_all = getenv("WMI_GEMM_GUARD_ALL") != nullptr;   // debug / A-B: bounds-checked epilogue for every tile
    static const bool narrow = getenv("WMI_GEMM_NARROW_STORES") != nullptr;  // debug / A-B: 8-byte epilogue stores
    static const bool qkv_swap = getenv("WMI_GEMM_QKV_SWAP") != nullptr;    // debug / A-B: q and k thirds of the big q|k|v grids in the transposed orientation
    static const bool resid_swap = getenv("WMI_GEMM_RESID_SWAP") != nullptr;   // debug / A-B: the residual epilogues of the big grids in the transposed orientation (16-byte f32 stores)
    static const bool vt_narrow = getenv("WMI_GEMM_VT_NARROW") != nullptr;  // debug / A-B: the V^T third of the encoder's q|k|v as 8-byte stores
    GemmArgs a = a_in; a.no_glds = (no_glds ? 1 : 0) | (guard_all ? 2 : 0) | (narrow ? 4 : 0) | (qkv_swap ? 8 : 0) | (vt_narrow ? 16 : 0) | (resid_swap ? 32 : 0);
    if (GemmLog * lg = tl_gemm_log) {
        const size_t wgs = (size_t) ((a.M + 63) / 64) * (size_t) ((a.N + 31) / 32);     // the smallest tile any dispatch below uses is 64 x 32
        if (!a.probe && lg->used + wgs * 5 <= lg->cap_words) {
            a.probe = lg->buf + lg->used;
            lg->entries.push_back(GemmLogEntry{epi, a.M, a.N, a.K, lg->used, (int) wgs});
            lg->used += wgs * 5;
        }
    }
    // The big grids (lock-step encoder: M = chunks x 1500) whose output is wide enough for several 192 x 256 tiles per CU go to the
    // persistent ping-pong kernel (k_gemm8.hip): mlp.0 (N = 4 S) and the cross K / V of all decoder layers (N = 2 L S).  Measured at
    // M = 12 000 (profiles/r04a_gemm8_lab_*): mlp.0 41 -> 35.7 us, cross K/V 132 -> 105 us, every output element identical.  The N = S
    // projections (one or two column tiles) and q|k|v (its V^T third wants the other fragment orientation) stay here.
    static const int g8 = getenv("WMI_GEMM8") ? atoi(getenv("WMI_GEMM8")) : 1;         // A/B knob: 0 = off
    if (g8 && !no_glds && (epi == EPI_F16_BIAS_GELU || epi == EPI_CROSS_KV) && a.M >= 4096 && a.N >= 1024 && (a.N % 256) == 0 && (a.K % 64) == 0 &&
        (epi != EPI_CROSS_KV || (a.S % 64) == 0)) {
        const long t192 = (long) ((a.M + 191) / 192) * (a.N / 256);
        // (round 6: 288-row tiles for these two measured slower — cross K/V 90.4 against 85.0 us, mlp.0 43.5 against 32.7: profiles/r06c_*)
        if (t192 >= 384) { GemmArgs b = a; b.no_glds = a.no_glds & 16; if (gemm8(epi, 192, true, b, st)) return; }
    }
    // (q|k|v stays below: on 288-row tiles — 42 x 6 = 252, ONE round at M = 12 000 — the persistent kernel measures 41.1 us against 41.0 us
    //  here, in situ 37.5 against 37.4: the V^T third's epilogue decides, not the tiling; WMI_GEMM8_QKV=1 routes it there for A/B)
    // (round 6: with the V^T third leaving in whole lines — gemm_epi.h: epilogue_vt_wide, chunks on 16-row boundaries — the tiling decides again:
    //  288-row tiles in whole rounds of the persistent kernel; WMI_GEMM8_QKV=0: off)
    static const bool g8_qkv = getenv("WMI_GEMM8_QKV") ? atoi(getenv("WMI_GEMM8_QKV")) != 0 : true;
    if (g8_qkv && g8 && !no_glds && epi == EPI_QKV_ENC && a.M >= 4096 && (a.N % 256) == 0 && (a.K % 64) == 0 && a.S > 0 && (a.S % 128) == 0 && a.N == 3 * a.S) {
        const long t288 = (long) ((a.M + 287) / 288) * (a.N / 256);
        const long n_cu = cu_count_x8(), rounds = (t288 + n_cu - 1) / n_cu;
        if (t288 * 5 >= rounds * n_cu * 4) { GemmArgs b = a; b.no_glds = a.no_glds & 16; if (gemm8(epi, 288, true, b, st, 32)) return; }
    }
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
