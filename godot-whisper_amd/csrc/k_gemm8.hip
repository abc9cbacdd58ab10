// f16 x f16 -> f32 MFMA GEMM for the BIG grids of the encoder (lock-step chunks: M = chunks x 1500 rows), eight wavefronts per
// workgroup in two groups that alternate between the LDS / DMA side and the matrix pipe ("ping-pong").
// SURVEY §8 rows a3, a4; arithmetic contract W/ggml.c:9737-9948 (f16 operands, f32 accumulation), graph W/whisper.cpp:1756-2074.
//
//   C[M][N] = A[M][K] . W[N][K]^T        same operands, same k order and the same v_mfma_f32_16x16x32_f16 as k_gemm.hip:
//                                        every output element is bit-identical to k_gemm's, whatever the tile.
//
// Why a second kernel.  k_gemm's 128 x 128 tile moves 256 operand bytes per k for 2 x 128 x 128 flops = 64 flop/B; a CU's vector
// memory path delivers 64 B/clk and its four matrix pipes retire 4096 flop/clk, i.e. at 64 flop/B the path is as busy as the pipes —
// and two co-resident workgroups share neither operands nor phases: measured 0.8 us per K step whatever the ring depth, MFMA busy
// 22-25 % (profiles/r03b_*).  Here ONE workgroup owns the CU with a BM x 256 tile (BM = 192: 110 flop/B, the path 58 % busy at
// the MFMA rate), and the overlap two co-resident workgroups gave by accident is made explicit:
//
//   waves 0-3 (group 0) compute the top BM/2 rows, waves 4-7 (group 1) the bottom BM/2; wave w and w + 4 share a SIMD.
//   A K step of 64 is two sub-steps of 32; every sub-step is a LOAD slot (fragments LDS -> registers, DMA issue for a later tile)
//   followed by an MFMA slot (FM x 4 MFMAs, nothing else), with one s_barrier between slots.  Group 1 runs one slot behind group 0,
//   so on every SIMD one wave is in its MFMA slot while its partner reads fragments and issues DMA: the matrix pipe is never asked
//   for by both, and never idle while the other wave waits for LDS.
//
//     slot      0        1        2        3        4     ...
//     group 0   LOAD00   MFMA00   LOAD01   MFMA01   LOAD10 ...
//     group 1   (idle)   LOAD00   MFMA00   LOAD01   MFMA01 ...
//
// Operand tiles go global -> LDS by DMA (global_load_lds_dwordx4 from inline asm, counted vmcnt, raw s_barrier: wave_ops.h) into an
// NST-deep ring of XOR-swizzled [rows][64] f16 images (swizzle applied to the lane's GLOBAL address, LDS written linearly), as in
// k_gemm.  Tile t + NST - 1 is issued at the start of K step t, each wave waits for its own pieces of tile t + 1 before the barrier
// that ends K step t.  Hazards: a ring slot is re-filled only after the barrier behind its last fragment read (those reads are
// retired with lgkmcnt(0) before that barrier); a tile is read only after the barrier behind every wave's vmcnt wait for it.
#include "kernels.h"
#include "wave_ops.h"
#include "gemm_epi.h"

#include <cstdlib>
#include <type_traits>

namespace wmi { namespace k {

namespace {

using namespace gemm_detail;

__device__ __forceinline__ uint32_t lds_off8(int row, int chunk) {     // byte offset inside a [rows][64] f16 image
    return (uint32_t) (row * 128 + ((chunk ^ (row & 7)) << 4));
}

template <int BM, int EPI, int NST, bool SWAPPED>
__global__ __launch_bounds__(512) void k_gemm8(const GemmArgs a) {
    constexpr int BN = 256, FM = BM / 32, FN = 4;
    constexpr int STAGE = (BM + BN) * 128;                 // bytes of one ring slot: A image, then W image
    constexpr int NA = BM / 8;                             // 1 KiB DMA pieces (8 rows) of the A image; the W image has 32
    constexpr int PA1 = NA / 8, PA0 = (NA - 4 * PA1) / 4;  // A pieces per wave of group 1 / group 0 (BM = 96: 1 / 2)
    static_assert(BM % 32 == 0 && 4 * PA0 + 4 * PA1 == NA, "tile height");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const unsigned long long pt0 = a.probe ? wall_clock64() : 0ull;
    unsigned long long pt1 = 0ull, pt2 = 0ull;

    // tile order: every XCD (workgroup id % 8) gets a contiguous run of the list; the list walks column groups of <= 8 tiles, inside
    // a group row by row: the 32 workgroups an XCD runs at a time share 4 A panels and <= 8 W panels (<= 2 MB + 4 x BM x K x 2 B in its 4 MB L2)
    const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN, nwg = ntm * ntn;
    int wg = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GN = 8;
    const int ng = wg / (ntm * GN), rem = wg - ng * (ntm * GN);
    const int gcur = ntn - ng * GN < GN ? ntn - ng * GN : GN;
    const int tm = rem / gcur, tn = ng * GN + rem % gcur;
    const int m0 = tm * BM, n0 = tn * BN;

    floatx4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / 64;
    const uint32_t lds0 = lds_addr(smem);
    const int frow = lane & 15, fq = lane >> 4;
    // fragment addresses inside a ring slot (kk = 0; kk = 1 flips chunk bit 2: + or - 64 bytes, resolved per row below)
    uint32_t offA[FM], offB[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) offA[i] = lds_off8(grp * (BM / 2) + i * 16 + frow, fq);
#pragma unroll
    for (int j = 0; j < FN; ++j) offB[j] = BM * 128 + lds_off8(wn * 64 + j * 16 + frow, fq);

    auto body = [&](auto grp_tag) {
        constexpr int G = decltype(grp_tag)::value;
        constexpr int PA = G ? PA1 : PA0;
        constexpr int LPT = PA + 4;                       // DMA instructions per wave and tile
        const __half * qA[PA > 0 ? PA : 1]; const __half * qB[4];
        uint32_t dA[PA > 0 ? PA : 1], dB[4];
        const int prow = lane >> 3;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int piece = G ? 4 * PA0 + wn * PA1 + p : wn * PA0 + p;
            const int lrow = piece * 8 + prow, pch = (lane & 7) ^ (lrow & 7);
            int r = m0 + lrow; if (r > a.M - 1) r = a.M - 1;
            qA[p] = a.A + (size_t) r * a.lda + pch * 8;
            dA[p] = (uint32_t) piece * 1024u;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = wave * 4 + p;
            const int lrow = piece * 8 + prow, pch = (lane & 7) ^ (lrow & 7);
            int r = n0 + lrow; if (r > a.N - 1) r = a.N - 1;
            qB[p] = a.W + (size_t) r * a.ldw + pch * 8;
            dB[p] = (uint32_t) (BM * 128) + (uint32_t) piece * 1024u;
        }
        auto issue = [&](int kt) {
            const uint32_t base = lds0 + (uint32_t) (kt % NST) * STAGE;
#pragma unroll
            for (int p = 0; p < PA; ++p) glds_asm<16>(qA[p] + kt * 64, base + dA[p]);
#pragma unroll
            for (int p = 0; p < 4; ++p) glds_asm<16>(qB[p] + kt * 64, base + dB[p]);
        };
        // this wave's pieces of tile kt have landed when at most the tiles issued after it are outstanding
        auto wait_tile = [&](int kt) {
            int later = kt + NST - 2 < nk - 1 ? NST - 2 : nk - 1 - kt;      // tiles issued behind kt so far (issue point: start of K step kt - 1)
            if (later < 0) later = 0;
            if constexpr (NST >= 3) {
                if (later >= 1) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory"); return; }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };

#pragma unroll
        for (int s0 = 0; s0 < NST - 1; ++s0) if (s0 < nk) issue(s0);
        // tile 0: outstanding behind it are the other prologue tiles
        if constexpr (NST >= 3) { if (nk >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (a.probe) pt1 = wall_clock64();
        if constexpr (G == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one slot behind

        for (int kt = 0; kt < nk; ++kt) {
            if (kt + NST - 1 < nk) issue(kt + NST - 1);
            const unsigned char * st = smem + (size_t) (kt % NST) * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // ---- LOAD slot
                half8 fa[FM], fb[FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) fb[j] = *(const half8 *) (st + (offB[j] ^ (uint32_t) (kk << 6)));
#pragma unroll
                for (int i = 0; i < FM; ++i) fa[i] = *(const half8 *) (st + (offA[i] ^ (uint32_t) (kk << 6)));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (G == 1) { if (kk == 1 && kt + 1 < nk) wait_tile(kt + 1); }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA slot
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        if constexpr (SWAPPED) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                        else                   acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_s_setprio(0);
                if constexpr (G == 0) { if (kk == 1 && kt + 1 < nk) wait_tile(kt + 1); }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (G == 0) __builtin_amdgcn_s_barrier();        // the barrier group 1 spent idle at the start
    };
    if (grp == 0) body(std::integral_constant<int, 0>{}); else body(std::integral_constant<int, 1>{});
    if (a.probe) { asm volatile("s_nop 0" ::: "memory"); pt2 = wall_clock64(); }

    // ------------------------------------------------------------------ epilogue (gemm_epi.h)
    const int mb = m0 + grp * (BM / 2), nb = n0 + wn * 64;
    const bool interior = m0 + BM <= a.M;                  // N is a multiple of 256 here
    if constexpr (SWAPPED) { if (interior) epilogue_cols<EPI, FM, FN, false>(a, acc, mb, nb, n0, lane); else epilogue_cols<EPI, FM, FN, true>(a, acc, mb, nb, n0, lane); }
    else                   { if (interior) epilogue_rows<EPI, FM, FN, false>(a, acc, mb, nb, lane);     else epilogue_rows<EPI, FM, FN, true>(a, acc, mb, nb, lane); }
    if (a.probe && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned long long * o = a.probe + (size_t) blockIdx.x * 5;
        o[0] = pt0; o[1] = pt1; o[2] = pt2; o[3] = wall_clock64(); o[4] = hwid;
    }
}

template <int BM, int EPI, int NST, bool SWAPPED>
void launch8(const GemmArgs & a, hipStream_t st) {
    const int ntm = (a.M + BM - 1) / BM, ntn = a.N / 256;
    const size_t smem = NST * (size_t) (BM + 256) * 128;
    static std::atomic<uint64_t> lds_ok{0};
    allow_full_lds((const void *) k_gemm8<BM, EPI, NST, SWAPPED>, lds_ok);
    hipLaunchKernelGGL((k_gemm8<BM, EPI, NST, SWAPPED>), dim3(ntm * ntn), dim3(512), smem, st, a);
}

} // namespace

// bm: 96 / 128 / 192 / 256 rows per tile.  false = this epilogue / shape is not served here (the caller keeps k_gemm).
bool gemm8(int epi, int bm, bool swapped, const GemmArgs & a, hipStream_t st) {
    if ((a.N % 256) != 0 || (a.K % 64) != 0 || a.M < 1) return false;
#define WMI_G8(E)                                                                                       \
    case E:                                                                                             \
        if (swapped) {                                                                                  \
            if (bm == 96) launch8<96, E, 3, true>(a, st); else if (bm == 128) launch8<128, E, 3, true>(a, st);  \
            else if (bm == 192) launch8<192, E, 2, true>(a, st); else if (bm == 256) launch8<256, E, 2, true>(a, st); else return false; \
        } else {                                                                                        \
            if (bm == 96) launch8<96, E, 3, false>(a, st); else if (bm == 128) launch8<128, E, 3, false>(a, st); \
            else if (bm == 192) launch8<192, E, 2, false>(a, st); else if (bm == 256) launch8<256, E, 2, false>(a, st); else return false; \
        }                                                                                               \
        return true;
    switch (epi) {
        WMI_G8(EPI_F16_BIAS)
        WMI_G8(EPI_F16_BIAS_GELU)
        WMI_G8(EPI_F32_BIAS_RESID)
        WMI_G8(EPI_CROSS_KV)
        WMI_G8(EPI_QKV_ENC)
        default: return false;
    }
#undef WMI_G8
}

}} // namespace wmi::k
