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

// Ring depths NSA (A image) / NSW (W image) and who requests what:
//   3 / 3  both groups request their share of step t + 2 behind their own fragment reads (BM <= 160: the ring fits 160 KB three deep)
//   2 / 3  BM = 192: three whole stages are 8 KB too many, so the smaller image gets two — group 0 requests A of step t + 1 at the START
//          of its LOAD slot (the stage was freed by the barrier in front of it: two slots of flight), group 1 requests W of step t + 2
//          behind its reads (four slots)
//   2 / 2  BM = 256: group 0 requests both images of step t + 1 at the start of its LOAD slot
template <int BM, int EPI, int NSA, int NSW, bool SWAPPED, int KS>
__global__ __launch_bounds__(512) void k_gemm8(const GemmArgs a) {
    constexpr int BN = 256, FM = BM / 32, FN = 4;
    constexpr int SZA = BM * 128, SZW = BN * 128;          // bytes of one stage of each ring
    constexpr int RING_W = NSA * SZA;                      // the W ring starts behind the A ring
    constexpr int NA = BM / 8, NW = BN / 8;                // 1 KiB DMA pieces (8 rows) per image
    constexpr bool DEEP = NSA == 3 && NSW == 3;
    // pieces per wave of group 0 / group 1
    constexpr int PA1 = DEEP ? NA / 8 : (NSA == 3 ? NA / 4 : 0), PA0 = DEEP ? (NA - 4 * PA1) / 4 : (NSA == 3 ? 0 : NA / 4);
    constexpr int PW1 = DEEP ? NW / 8 : (NSW == 3 ? NW / 4 : 0), PW0 = DEEP ? NW / 8 : (NSW == 3 ? 0 : NW / 4);
    static_assert(BM % 32 == 0 && 4 * PA0 + 4 * PA1 == NA && 4 * PW0 + 4 * PW1 == NW, "tile");
    static_assert((NSA == 2 || NSA == 3) && (NSW == 2 || NSW == 3) && !(NSA == 3 && NSW == 2), "ring depths");
    constexpr int NKK = KS / 32;                           // MFMA k-steps per slot; 64 / KS slots per K step
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef WMI_G8_LAB                                           // scratch/lab/gemm8_lab.hip: ablations
    const int lab = a.no_glds;
#else
    constexpr int lab = 0;
#endif
    const int grp = wave >> 2, wn = wave & 3;
    const unsigned long long pt0 = a.probe ? wall_clock64() : 0ull;
    unsigned long long pt1 = 0ull, pt2 = 0ull, pte = 0ull;

    // Persistent workgroups (one per CU): the tile list walks column groups of <= 8 tiles, inside a group row by row; XCD x
    // (workgroup id % 8) owns the contiguous run [x * per, (x + 1) * per) of it and its workgroups take the run's entries in turn, so the
    // 32 tiles an XCD works on at a time share 4 A panels and <= 8 W panels (<= 2 MB + 4 x BM x K x 2 B in its 4 MB L2).
    const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN, ntiles = ntm * ntn;
    const int nx = (int) gridDim.x >> 3;                    // workgroups per XCD (grid is a multiple of 8)
    const int xcd = (int) blockIdx.x & 7, jx = (int) blockIdx.x >> 3;
    const int per = (ntiles + 7) >> 3;
    const int run = ntiles - xcd * per < per ? ntiles - xcd * per : per;     // entries of this XCD's run (may be <= 0)
    const int n_my = run > jx ? (run - jx + nx - 1) / nx : 0;
    constexpr int GN = 8;
    auto tile_origin = [&](int i, int & m0, int & n0) {
        const int idx = xcd * per + jx + i * nx;
        const int ng = idx / (ntm * GN), rem = idx - ng * (ntm * GN);
        const int gcur = ntn - ng * GN < GN ? ntn - ng * GN : GN;
        m0 = (rem / gcur) * BM; n0 = (ng * GN + rem % gcur) * BN;
    };

    floatx4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / 64;
    const int total = n_my * nk;                           // K steps of this workgroup, over all of its tiles
    const uint32_t lds0 = lds_addr(smem);
    const int frow = lane & 15, fq = lane >> 4;
    // fragment addresses inside a stage for k-step 0; k-step q flips bit 2 of the chunk index: chunk = (4 q + fq) ^ (row & 7).  Fragment
    // row i / column j is 16 rows = 2 KiB further on and leaves row & 7 alone: ONE address register per operand, the rest is the
    // ds_read's immediate offset (ten registers less than an address per fragment: the deferred stores need them)
    const uint32_t offA0 = lds_off8(grp * (BM / 2) + frow, fq);
    const uint32_t offB0 = RING_W + lds_off8(wn * 64 + frow, fq);

    auto body = [&](auto grp_tag) {
        constexpr int G = decltype(grp_tag)::value;
        // Who requests what.  A DMA instruction holds its wave for ~110 cycles wherever it is issued (the CU's vector memory path accepts
        // 1 KiB per ~27 cycles: ~38 B/clk), so requests belong in LOAD slots, where the wave has nothing else to do — between the MFMAs of
        // an MFMA slot each one cost the matrix pipe 115-130 cycles (measured: profiles/r04_gemm8_lab.txt).
        //   3 / 3  both groups request their share of step t + 2
        //   2 / 3  group 0 requests A of step t + 1 (its LOAD slot starts behind the barrier that freed the stage: two slots of flight),
        //          group 1 requests W of step t + 2
        //   2 / 2  group 0 requests both images of step t + 1
        constexpr int PA = G ? PA1 : PA0, PW = G ? PW1 : PW0;
        constexpr int LATE = (NSA == 3 ? PA : 0) + (NSW == 3 ? PW : 0);      // this wave's DMA instructions per step that run TWO steps ahead
        const int prow = lane >> 3;
        struct Stream { int i = 0, k = 0, n = 0; const __half * base = nullptr; };
        Stream sa, sw;
        // One byte offset per operand: piece p of a wave's run is 8 rows further down the same (swizzled) 16-byte column — + p * 16 ld
        // bytes, formed in front of each request (glds_run_affine).  A rows past the bottom edge of the matrix read the last valid row.
        uint32_t vA0 = 0, vAclamp = 0, vW0 = 0;
        const int pa0 = DEEP ? (G ? 4 * PA0 + wn * PA1 : wn * PA0) : wn * PA;     // this wave's first piece of each image
        const int pw0 = DEEP ? wave * PW : wn * PW;
        const int pchunk = (lane & 7) ^ (prow & 7);          // (piece rows are multiples of 8: row & 7 = prow)
        if constexpr (PW > 0) vW0 = (uint32_t) ((pw0 * 8 + prow) * a.ldw + pchunk * 8) * 2u;     // W rows never leave the matrix (N is a multiple of 256)
        bool a_edge = false;
        auto adv = [&](Stream & t) { ++t.n; if (++t.k == nk) { t.k = 0; ++t.i; } };
        auto request_a = [&]() {
            if constexpr (PA > 0) {
                if (sa.k == 0) {                               // first step of a tile: source rows (clamped at the bottom edge of the matrix)
                    int m0, n0; tile_origin(sa.i, m0, n0);
                    sa.base = a.A + (size_t) m0 * a.lda;
                    const int rmax = a.M - 1 - m0;
                    a_edge = rmax < BM - 1;                    // (workgroup-uniform: the tile hangs over the bottom edge)
                    int lrow = pa0 * 8 + prow; if (lrow > rmax) lrow = rmax;
                    vA0 = (uint32_t) (lrow * a.lda + pchunk * 8) * 2u;
                    vAclamp = (uint32_t) (rmax * a.lda + pchunk * 8) * 2u;
                }
                const uint32_t dstA = lds0 + (uint32_t) (sa.n % NSA) * SZA + (uint32_t) pa0 * 1024u;
                constexpr int PA_1 = PA > 8 ? 8 : PA, PA_2 = PA - PA_1;       // (a run is at most eight pieces; 288-row tiles have nine per wave)
                const uint32_t strideA = (uint32_t) a.lda * 16u;
                if (a_edge) glds_run_affine<PA_1, true>(vA0, strideA, vAclamp, sa.base + sa.k * 64, dstA);
                else        glds_run_affine<PA_1, false>(vA0, strideA, 0u, sa.base + sa.k * 64, dstA);
                if constexpr (PA_2 > 0) {
                    if (a_edge) glds_run_affine<PA_2, true>(vA0 + 8u * strideA, strideA, vAclamp, sa.base + sa.k * 64, dstA + 8192u);
                    else        glds_run_affine<PA_2, false>(vA0 + 8u * strideA, strideA, 0u, sa.base + sa.k * 64, dstA + 8192u);
                }
            }
            adv(sa);
        };
        auto request_w = [&]() {
            if constexpr (PW > 0) {
                if (sw.k == 0) { int m0, n0; tile_origin(sw.i, m0, n0); sw.base = a.W + (size_t) n0 * a.ldw; }
                glds_run_affine<PW, false>(vW0, (uint32_t) a.ldw * 16u, 0u, sw.base + sw.k * 64, lds0 + RING_W + (uint32_t) (sw.n % NSW) * SZW + (uint32_t) pw0 * 1024u);
            }
            adv(sw);
        };
        // this wave's pieces of step gs have landed when at most its two-steps-ahead requests for step gs + 1 are outstanding
        auto wait_step = [&](int gs) {
            if constexpr (PA + PW > 0) {
                if constexpr (LATE > 0) {
                    const int n_late = NSW == 3 ? sw.n : sa.n;           // (the deep streams advance together)
                    if (n_late > gs + 1) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LATE) : "memory"); return; }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        };

        // prologue: NSx - 1 steps of each image
#pragma unroll
        for (int s0 = 0; s0 < NSA - 1; ++s0) if (sa.n < total) request_a();
#pragma unroll
        for (int s0 = 0; s0 < NSW - 1; ++s0) if (sw.n < total) request_w();
        if (total > 0) wait_step(0);
        __builtin_amdgcn_s_barrier();
        if (a.probe) pt1 = wall_clock64();
        if constexpr (G == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one slot behind

#ifdef WMI_G8_LAB
        long long tc[5] = {0, 0, 0, 0, 0}, tprev = 0;     // cycles: LOAD work, barrier behind LOAD, MFMA work, vmcnt wait, barrier behind MFMA
        const bool stampit = (lab & 2048) != 0;
#define G8_STAMP(i) do { if (stampit) { const long long t_ = clock64(); tc[i] += t_ - tprev; tprev = t_; } } while (0)
        if (stampit) tprev = clock64();
#else
#define G8_STAMP(i) do { } while (0)
#endif
        int gs = 0;
        // One K step; SW = orientation of the accumulator fragments of this tile (SWAPPED, except the V^T third of the encoder's q|k|v:
        // its epilogue wants four consecutive ROWS per lane — time steps of V^T — while q and k want four consecutive columns; a column
        // tile never straddles the thirds, so the choice is per tile and workgroup-uniform)
        auto kstep = [&](auto sw_tag) {
            constexpr bool SW = decltype(sw_tag)::value;
            {
                const unsigned char * stA = smem + (size_t) (gs % NSA) * SZA;
                const unsigned char * stW = smem + (size_t) (gs % NSW) * SZW;
#pragma unroll
                for (int sl = 0; sl < 64 / KS; ++sl) {
                    // ---- LOAD slot: the fragments of NKK k-steps, then this wave's requests (one or two steps ahead: every target stage was
                    // freed by the barrier in front of this slot and is not the one being read)
                    half8 fa[NKK][FM], fb[NKK][FN];
                    if (!(lab & 32) || gs == 0) {
#pragma unroll
                        for (int q = 0; q < NKK; ++q) {
                            const uint32_t kx = (uint32_t) ((sl * NKK + q) << 6);
#pragma unroll
                            for (int j = 0; j < FN; ++j) fb[q][j] = *(const half8 *) (stW + (offB0 ^ kx) + j * 2048);
#pragma unroll
                            for (int i = 0; i < FM; ++i) fa[q][i] = *(const half8 *) (stA + (offA0 ^ kx) + i * 2048);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // the requests go out while the fragment reads are in flight: a DMA instruction holds the wave for ~85 cycles (the CU's
                    // vector memory path takes 1 KiB per ~16-20 cycles from four waves at once), which is as long as the reads take to return —
                    // behind the lgkmcnt wait the two added up to a LOAD slot of 1100-1270 cycles against 875 of MFMA (slot accounting, lab)
                    if (sl == 0 && !(lab & 8)) {
                        if (sa.n < total) request_a();
                        if (sw.n < total) request_w();
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    G8_STAMP(0);
                    if constexpr (G == 1) { if (sl == 64 / KS - 1 && gs + 1 < total) wait_step(gs + 1); }
                    G8_STAMP(3);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    G8_STAMP(1);
                    // ---- MFMA slot
                    __builtin_amdgcn_s_setprio(1);
                    if (!(lab & 16))
#pragma unroll
                    for (int q = 0; q < NKK; ++q)
#pragma unroll
                        for (int i = 0; i < FM; ++i)
#pragma unroll
                            for (int j = 0; j < FN; ++j) {
                                if constexpr (SW)      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[q][j], fa[q][i], acc[i][j], 0, 0, 0);
                                else                   acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q][i], fb[q][j], acc[i][j], 0, 0, 0);
                            }
                    __builtin_amdgcn_s_setprio(0);
                    G8_STAMP(2);
                    if constexpr (G == 0) { if (sl == 64 / KS - 1 && gs + 1 < total) wait_step(gs + 1); }
                    G8_STAMP(3);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    G8_STAMP(4);
                }
            }
        };
        constexpr bool RT_ORIENT = SWAPPED && EPI == EPI_QKV_ENC;      // orientation decided per tile
        for (int ti = 0; ti < n_my; ++ti) {
            int m0, n0; tile_origin(ti, m0, n0);
            const bool sw_tile = SWAPPED && !(RT_ORIENT && n0 >= 2 * a.S);
            if constexpr (RT_ORIENT) {
                if (sw_tile) { for (int kt = 0; kt < nk; ++kt, ++gs) kstep(std::true_type{}); }
                else         { for (int kt = 0; kt < nk; ++kt, ++gs) kstep(std::false_type{}); }
            } else {
                for (int kt = 0; kt < nk; ++kt, ++gs) kstep(std::integral_constant<bool, SWAPPED>{});
            }
            // ---- epilogue of tile ti (gemm_epi.h); the requests for the next tile's first steps are in flight.
            // Both groups run it in the SAME slot (group 0 sits out group 1's last MFMA slot, group 1 takes its extra barrier behind the
            // epilogue): one slot behind each other, each group's epilogue had the partner waiting at the next barrier for all of it.
            if constexpr (G == 0) __builtin_amdgcn_s_barrier();
            if (a.probe && ti == 0) pt2 = wall_clock64();
            if (lab & 512) { m0 = 0; n0 = 0; }              // (lab) every workgroup stores to the same patch: no HBM write traffic
            const int mb = m0 + grp * (BM / 2), nb = n0 + wn * 64;
            const bool interior = m0 + BM <= a.M;          // N is a multiple of 256 here
            if (lab & 4) {                                  // (lab) no epilogue: keep the accumulators alive, store nothing
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) asm volatile("" :: "v"(acc[i][j]));
            } else
            if (SWAPPED && sw_tile) {
                if constexpr (SWAPPED) {
                    if (lab & 1024) { if (interior) epilogue_cols<EPI, FM, FN, false>(a, acc, mb, nb, n0, lane); else epilogue_cols<EPI, FM, FN, true>(a, acc, mb, nb, n0, lane); }
                    else            { if (interior) epilogue_cols_wide<EPI, FM, FN, false>(a, acc, mb, nb, n0, lane); else epilogue_cols_wide<EPI, FM, FN, true>(a, acc, mb, nb, n0, lane); }
                }
            }
            else if constexpr (!SWAPPED || RT_ORIENT) { if (interior) epilogue_rows<EPI, FM, FN, false>(a, acc, mb, nb, lane); else epilogue_rows<EPI, FM, FN, true>(a, acc, mb, nb, lane); }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (a.probe && ti == 0) pte = wall_clock64();
            if constexpr (G == 1) __builtin_amdgcn_s_barrier();
#ifdef WMI_G8_LAB
            if (stampit) tprev = clock64();                // (the epilogue is not part of the slot accounting)
#endif
        }
#ifdef WMI_G8_LAB
        if (stampit && a.probe && lane == 0) {
            unsigned long long * o = a.probe + (size_t) gridDim.x * 5 + ((size_t) blockIdx.x * 8 + wave) * 5;
            for (int i = 0; i < 5; ++i) o[i] = (unsigned long long) tc[i];
        }
#endif
#undef G8_STAMP
        if constexpr (G == 0) __builtin_amdgcn_s_barrier();        // the barrier group 1 spent idle at the start
    };
    if (grp == 0) body(std::integral_constant<int, 0>{}); else body(std::integral_constant<int, 1>{});
    if (a.probe && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long * o = a.probe + (size_t) blockIdx.x * 5;
        o[0] = pt0; o[1] = pt1; o[2] = pt2; o[3] = wall_clock64(); o[4] = pte;      // entry, first tile landed, first tile's K loop done, all done, first epilogue done
    }
}

template <int BM, int EPI, int NSA, int NSW, bool SWAPPED, int KS>
void launch8(const GemmArgs & a, hipStream_t st) {
    const int ntm = (a.M + BM - 1) / BM, ntn = a.N / 256;
    const size_t smem = (size_t) NSA * BM * 128 + (size_t) NSW * 256 * 128;
    static std::atomic<uint64_t> lds_ok{0};
    allow_full_lds((const void *) k_gemm8<BM, EPI, NSA, NSW, SWAPPED, KS>, lds_ok);
    const int n_cu = cu_count_x8();
    const int tiles = ntm * ntn;
    int grid = tiles < n_cu ? (tiles + 7) & ~7 : n_cu;
#ifdef WMI_G8_LAB
    if (a.no_glds & 4096) grid = 64;                        // (lab) a quarter of the CUs: is the epilogue's store rate a chip-wide limit?
#endif
    hipLaunchKernelGGL((k_gemm8<BM, EPI, NSA, NSW, SWAPPED, KS>), dim3(grid), dim3(512), smem, st, a);
}


} // namespace

// bm: rows per tile; ks: k extent of a slot (32 or 64).  false = this epilogue / shape / tile is not served here (the caller keeps gemm()).
// The library instantiates what its dispatch uses (k_gemm.hip: gemm()): the transposed orientation on 192-row tiles (128 as the fallback
// for narrower outputs) for the GELU and cross K/V epilogues; the lab (scratch/lab/gemm8_lab.hip) builds the whole grid of variants.
bool gemm8(int epi, int bm, bool swapped, const GemmArgs & a, hipStream_t st, int ks) {
    if ((a.N % 256) != 0 || (a.K % 64) != 0 || a.M < 1) return false;
#ifdef WMI_G8_LAB
    if (ks != 32 && ks != 64) return false;
#define WMI_G8B(E, SW, KSV)                                                                             \
            if (bm == 96) launch8<96, E, 3, 3, SW, KSV>(a, st); else if (bm == 128) launch8<128, E, 3, 3, SW, KSV>(a, st);  \
            else if (bm == 160) launch8<160, E, 3, 3, SW, KSV>(a, st);                                  \
            else if (bm == 192) launch8<192, E, 2, 3, SW, KSV>(a, st); else if (bm == 256) launch8<256, E, 2, 2, SW, KSV>(a, st); else return false;
#define WMI_G8T(E)                                            /* the one-round tile: 288 rows, 32-deep slots (144 accumulator registers) */ \
    if (epi == E && bm == 288 && ks == 32 && swapped) { launch8<288, E, 2, 2, true, 32>(a, st); return true; }
    WMI_G8T(EPI_F16_BIAS_GELU) WMI_G8T(EPI_QKV_ENC)
    if (epi == EPI_QKV_ENC && swapped && ks == 64 && bm == 192) { launch8<192, EPI_QKV_ENC, 2, 3, true, 64>(a, st); return true; }
#undef WMI_G8T
#define WMI_G8(E)                                                                                       \
    case E:                                                                                             \
        if (swapped) { if (ks == 64) { WMI_G8B(E, true, 64) } else { WMI_G8B(E, true, 32) } }           \
        else         { if (ks == 64) { WMI_G8B(E, false, 64) } else { WMI_G8B(E, false, 32) } }         \
        return true;
    switch (epi) {
        WMI_G8(EPI_F16_BIAS)
        WMI_G8(EPI_F16_BIAS_GELU)
        WMI_G8(EPI_F32_BIAS_RESID)
        WMI_G8(EPI_CROSS_KV)
        default: return false;
    }
#undef WMI_G8
#undef WMI_G8B
#else
    if (!swapped) return false;
    // q|k|v of the lock-step encoder: 288-row tiles on 32-deep slots (42 x 6 = 252 tiles at M = 12 000: one round), orientation per column tile
    if (epi == EPI_QKV_ENC && bm == 288 && ks == 32) { launch8<288, EPI_QKV_ENC, 2, 2, true, 32>(a, st); return true; }
    if (epi == EPI_CROSS_KV && bm == 288 && ks == 32) { launch8<288, EPI_CROSS_KV, 2, 2, true, 32>(a, st); return true; }
    if (epi == EPI_F16_BIAS_GELU && bm == 288 && ks == 32) { launch8<288, EPI_F16_BIAS_GELU, 2, 2, true, 32>(a, st); return true; }
    if (ks != 64) return false;
#define WMI_G8(E)                                                                                       \
    case E:                                                                                             \
        if (bm == 192) launch8<192, E, 2, 3, true, 64>(a, st); else if (bm == 128) launch8<128, E, 3, 3, true, 64>(a, st); else return false;   \
        return true;
    switch (epi) {
        WMI_G8(EPI_F16_BIAS_GELU)
        WMI_G8(EPI_CROSS_KV)
        default: return false;
    }
#undef WMI_G8
#endif
}

}} // namespace wmi::k
