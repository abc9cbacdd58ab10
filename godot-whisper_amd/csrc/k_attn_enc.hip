// Encoder self-attention for gfx950, second form (reference: W/whisper.cpp:1877-1950, soft-max W/ggml.c:11116-11201).
// Own translation unit: built with -mllvm -amdgpu-mfma-vgpr-form (csrc/Makefile) so that the MFMA accumulators the soft-max chain
// reads and rescales every tile are VGPRs, not AGPRs behind v_accvgpr_read / _write pairs.
#include "kernels.h"
#include "wave_ops.h"
#include <atomic>
#include <cstdlib>

namespace wmi { namespace k {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// Tile rows are 128 B (8 chunks of 16 B); chunk c of row r sits in slot c ^ ((r >> 1) & 7).  A 32-row fragment read (ds_read_b128, lane =
// row) is served in the hardware's lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32): 16 rows = 8 even + 8 odd ones, whose 128-byte
// halves of the 256-byte bank window differ by parity — within a parity the key (r >> 1) & 7 takes all eight values, so a group touches every
// bank once.  With the key r & 7 of the 16-row kernels rows r and r + 24 (r + 8 within a group's lanes) met on the same banks: PMC
// SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE (profiles/r03b_pmc_enc8_split.txt).
__device__ __forceinline__ uint32_t lds_off(int row, int chunk) { return (uint32_t) (row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)); }
// fmaxf() canonicalises both operands first (v_max_f32 x, x) because they could be signalling NaNs: 3 instructions per maximum.
// MFMA results are what they are; one v_max3_f32 takes two new values per instruction.
__device__ __forceinline__ float max3(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

// ------------------------------------------------------------------------------------------------
// Encoder attention, second form (round 3): 32 query rows per wavefront on v_mfma_f32_32x32x16_f16, everything that belongs to a
// query row in ONE lane, the soft-max chain in packed / mixed-precision instructions.
//
//   scores   S^T[key][q] = K . Q^T : A = K tile rows from LDS (one ds_read_b128 per MFMA), B = the wavefront's 32 query rows
//            (registers, loaded once).  Lane (q = lane % 32, g = lane / 32) receives keys 8j + 4g + r of each 32-key block.
//   soft-max per element: v_pk_fma_f32 (s * scale - m; scale = 1/8 is a power of two, so the fused form rounds exactly like the
//            reference's separate scale and subtract), v_cvt_pk_f16_f32 (the reference's f16 argument), v_fma_mix_f32
//            (f16 -> f32 times log2 e in one instruction), v_exp_f32, v_cvt_pk_f16_f32 (the reference's f16 result),
//            v_dot2c_f32_f16 against (1, 1) for the row sum: 4 + exp issue slots per element (the first form needs ~14).
//   output   O^T[dv][q] = V^T . P^T : A = V^T tile rows from LDS (the time index of V^T is stored with bits 2 and 3 swapped,
//            kernels.h vt_pos, so the eight values that pair with a lane's numerators are one 16-byte piece), B = the packed
//            numerators exactly as the lane holds them.  The accumulators of a lane all belong to ITS query row: rescaling by
//            exp(m_old - m_new), the final 1 / sum and the row sum itself never leave the lane.
//   tiles    64 keys of K and V^T per key group go global -> LDS by DMA (global_load_lds_dwordx4 from inline asm, XOR swizzle
//            applied to the global address), two stages, one barrier per tile.
//   ONE      true : one sweep, running maximum, accumulators rescaled in f32 when it rises (the departure the decoder's
//                   cross-attention already makes: e is rounded to f16 relative to the maximum SO FAR);
//            false: two sweeps, exact row maximum first (the reference's soft-max argument bit for bit).
//   KS       key groups per workgroup (own tiles, common barriers); partial (m, sum, O^T) combined through LDS at the end.
template <int QW, int KS, bool ONE, int NST, bool WIDE_OUT = true>
__global__ __launch_bounds__(QW * KS * 64) void k_attn_enc2(const __half * __restrict__ q, const __half * __restrict__ k,
                                                            const __half * __restrict__ vt, int T, int Tpad, int S,
                                                            __half * __restrict__ out, float * __restrict__ out32, int xcd_order, int qk_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef float floatx16 __attribute__((ext_vector_type(16)));
    typedef float float2v __attribute__((ext_vector_type(2)));
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    constexpr int STAGE = 16384;                                   // K tile 8 KB + V^T tile 8 KB
    // the output images of the query wavefronts sit behind everything else the kernel keeps in LDS (ring, maxima, combine area)
    constexpr size_t RING_B = (size_t) KS * NST * STAGE, EXTRA_B = (!ONE && KS > 1) ? (size_t) KS * QW * 32 * 4 : 0;
    constexpr size_t COMB_B = KS > 1 ? (size_t) (KS - 1) * QW * 64 * 34 * 4 : 0;
    // (one key group: the images reuse the ring once every wavefront is done with it — 32 KB per workgroup instead of 48: four workgroups
    //  per CU, which is what the registers allow, instead of three)
    constexpr size_t TB_OFF = KS == 1 ? 0 : ((RING_B + EXTRA_B) > COMB_B ? (RING_B + EXTRA_B) : COMB_B);
    constexpr float LOG2E = 1.44269504088896340736f;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: scalar branches
    const int grp = KS == 1 ? 0 : wave / QW, qw = wave - grp * QW;
    const int i = lane & 31, g = lane >> 5;
    // XCD-aware order: workgroup ids go round-robin over the 8 XCDs, so the query blocks of one (chunk, head) — which stream the same
    // K / V^T — would sit on eight different L2s and fetch them eight times (PMC: 213 MB per launch at 8 chunks for 37 MB of operands).
    // The launch id is mapped so that every XCD owns a contiguous run of (chunk, head, query block) triples.
    int bx = blockIdx.x, head = blockIdx.y, bz = blockIdx.z;
    const bool mfma_prio = (xcd_order & 2) != 0;            // A/B: s_setprio 1 around the MFMA clusters (WMI_ATTN_PRIO)
    if (xcd_order & 1) {
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int q8 = nwg / 8, r8 = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        bx = wg % gridDim.x; wg /= gridDim.x; head = wg % gridDim.y; bz = wg / gridDim.y;
    }
    const int q0 = bx * (QW * 32) + qw * 32;
    {
        const size_t zb = bz;
        q += zb * (size_t) qk_rows * S; k += zb * (size_t) qk_rows * S; vt += zb * (size_t) S * Tpad;
        if (out32) out32 += zb * (size_t) T * S; else out += zb * (size_t) T * S;
    }
    unsigned char * const ring = smem + grp * (NST * STAGE);
    const uint32_t ring_lds = lds_addr(ring);

    half8 qf[4];
    {
        int qr = q0 + i; if (qr > T - 1) qr = T - 1;
        const __half * qp = q + (size_t) qr * S + head * 64 + g * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const half8 *) (qp + kk * 16);
        // the loads are consumed HERE as far as the compiler is concerned: otherwise their vmcnt waits sit in front of the first
        // MFMAs of the tile loop, where the counter they test is the tile DMA just issued (the next tile would be waited for
        // before the current one is multiplied)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(qf[kk]));
    }

    const int ntile_all = (T + 63) / 64, ntile_g = (ntile_all + KS - 1) / KS;
    const int kbeg = grp * ntile_g * 64;

    // DMA of one tile: 16 pieces of 1 KB (8 rows x 128 B): 8 of K rows, 8 of V^T rows, dealt round-robin to the group's wavefronts
    // (piece pp * QW + qw: which operand a piece belongs to is known at compile time).  Lane L of a piece lands at L * 16, i.e.
    // (row = L / 8, slot = L % 8), and fetches chunk slot ^ (row & 7) of that row: the XOR swizzle applied to the global address
    constexpr int PPO = 8 / QW;                                    // pieces per wavefront and operand
    const int prow = lane >> 3;                                    // row 8p + prow of the tile: swizzle key (4p + prow / 2) & 7
    const __half * const kcol = k + head * 64;
    const __half * const vrow = vt + (size_t) (head * 64 + prow) * Tpad;
    auto issue = [&](int kt0, int stage) {
        const int kv = kt0 > Tpad - 64 ? Tpad - 64 : kt0;
        const uint32_t dst = ring_lds + stage * STAGE;
#pragma unroll
        for (int pp = 0; pp < PPO; ++pp) {
            const int p = pp * QW + qw;
            const int pch = (lane & 7) ^ ((4 * p + (prow >> 1)) & 7);
            int r = kt0 + p * 8 + prow; if (r > T - 1) r = T - 1;
            glds_asm<16>(kcol + (size_t) r * S + pch * 8, dst + p * 1024);
        }
#pragma unroll
        for (int pp = 0; pp < PPO; ++pp) {
            const int p = pp * QW + qw;
            const int pch = (lane & 7) ^ ((4 * p + (prow >> 1)) & 7);
            glds_asm<16>(vrow + (size_t) (p * 8) * Tpad + kv + pch * 8, dst + 8192 + p * 1024);
        }
    };

    float m = -INFINITY;
    auto tile_scores = [&](const unsigned char * st, int kt0, floatx16 (&s)[2]) {
        if (mfma_prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            s[blk] = floatx16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const half8 kf = *(const half8 *) (st + lds_off(blk * 32 + i, kk * 2 + g));
                s[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], s[blk], 0, 0, 0);
            }
        }
        if (mfma_prio) __builtin_amdgcn_s_setprio(0);
        if (kt0 + 64 > T) {                                        // ragged last tile (and whole tiles past T of the last key group)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int key = kt0 + blk * 32 + 8 * (v >> 2) + 4 * g + (v & 3);
                    if (key >= T) s[blk][v] = -INFINITY;
                }
        }
    };
    auto tile_max = [&](const floatx16 (&s)[2]) -> float {
        float mx = s[0][0];
#pragma unroll
        for (int v = 1; v < 16; v += 2) mx = max3(mx, s[0][v], s[0][v + 1 < 16 ? v + 1 : 0]);
#pragma unroll
        for (int v = 0; v < 16; v += 2) mx = max3(mx, s[1][v], s[1][v + 1]);
        mx = fmaxf(mx, xor_lane<32>(mx));
        return mx * 0.125f;                                        // exact: max and a power-of-two scale commute
    };

    const int nt = ntile_g;
    // NST-deep ring, one barrier per tile: tiles t + 1 .. t + NST - 2 stay in flight while tile t is multiplied (a tile takes
    // ~0.4 us of this wavefront's time, a DMA from L2 / Infinity Cache about as long to arrive).  The wait is counted: tile t has
    // landed when at most (NST - 2) tiles' worth of this wavefront's loads are outstanding; vmcnt retires in order.
    constexpr int LPT = 2 * PPO;                                   // DMA instructions per wavefront and tile
    auto ring_fill = [&]() {
#pragma unroll
        for (int u = 0; u < NST - 1; ++u) if (u < nt) issue(kbeg + u * 64, u);
    };
    auto ring_step = [&](int t) {
        if (NST > 2 && nt - 1 - t >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * LPT) : "memory");
        else                                  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // tile t is there for everyone; the stage multiplied last step is free
        asm volatile("" ::: "memory");
        if (t + NST - 1 < nt) issue(kbeg + (t + NST - 1) * 64, (t + NST - 1) % NST);
    };
    if constexpr (!ONE) {
        // ---- sweep 1: exact row maximum (K tiles only: the V^T half of each stage is fetched too — same DMA routine — and ignored)
        ring_fill();
        for (int t = 0; t < nt; ++t) {
            ring_step(t);
            floatx16 s[2];
            tile_scores(ring + (t % NST) * STAGE, kbeg + t * 64, s);
            m = fmaxf(m, tile_max(s));
        }
        if constexpr (KS > 1) {
            float * xm = (float *) (smem + KS * NST * STAGE);      // [KS][QW][32]
            __syncthreads();
            if (lane < 32) xm[(grp * QW + qw) * 32 + lane] = m;
            __syncthreads();
#pragma unroll
            for (int gg = 0; gg < KS; ++gg) m = fmaxf(m, xm[(gg * QW + qw) * 32 + i]);
        }
        __syncthreads();                                           // every wavefront is done with the ring before it is refilled
    }

    // ---- main sweep
    float l = 0.0f;
    floatx16 o[2];
    o[0] = floatx16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    o[1] = o[0];
    const half2v ones = {(_Float16) 1.0f, (_Float16) 1.0f};
    ring_fill();
    for (int t = 0; t < nt; ++t) {
        ring_step(t);
        const unsigned char * st = ring + (t % NST) * STAGE;
        floatx16 s[2];
        tile_scores(st, kbeg + t * 64, s);
        if constexpr (ONE) {
            const float mn = fmaxf(m, tile_max(s));
            // alpha = exp(m - mn) in f32; nothing seen yet (m = -inf) or nothing valid at all (mn = -inf): factor irrelevant / 1
            const float alpha = mn == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f((m - mn) * LOG2E);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int v = 0; v < 16; ++v) o[mt][v] *= alpha;
        }
        const float negm = m == -INFINITY ? 0.0f : -m;             // a key group without a valid key: s = -inf everywhere, e = 0
        const float2v nm2 = {negm, negm}, sc2 = {0.125f, 0.125f};
        // per 32-key block: numerators, then that block's P.V MFMAs — the matrix pipe works on block 0 while the VALU forms the numerators
        // of block 1 (in program order "all numerators, then all P.V" a wavefront's own MFMAs and VALU work never overlapped)
        const unsigned char * sv = st + 8192;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            half8 pf[2];
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                const float2v s2 = {s[blk][2 * pr], s[blk][2 * pr + 1]};
                const float2v a2 = __builtin_elementwise_fma(s2, sc2, nm2);
                const half2v ah = __builtin_convertvector(a2, half2v);           // the reference's f16 table index
                const uint32_t ab = __builtin_bit_cast(uint32_t, ah);
                float x0, x1;
                asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(x0) : "v"(ab), "v"(LOG2E));
                asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(x1) : "v"(ab), "v"(LOG2E));
                const float2v e2 = {__builtin_amdgcn_exp2f(x0), __builtin_amdgcn_exp2f(x1)};
                const half2v eh = __builtin_convertvector(e2, half2v);           // ... and its f16 entry
                l = __builtin_amdgcn_fdot2(eh, ones, l, false);
                pf[pr >> 2][2 * (pr & 3)] = eh[0];
                pf[pr >> 2][2 * (pr & 3) + 1] = eh[1];
            }
            if (mfma_prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const half8 vf = *(const half8 *) (sv + lds_off(mt * 32 + i, blk * 4 + u * 2 + g));
                    o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u], o[mt], 0, 0, 0);
                }
            if (mfma_prio) __builtin_amdgcn_s_setprio(0);
        }
    }
    l += xor_lane<32>(l);                                          // the two lanes of a query row saw disjoint keys

    if constexpr (KS > 1) {                                        // key groups 1.. hand (m, sum, O^T) to group 0 through the ring's LDS
        __syncthreads();
        float * co = (float *) smem;                               // [(KS-1) * QW][64][33]
        float * cm = (float *) (smem + (size_t) (KS - 1) * QW * 64 * 33 * 4);
        if (grp > 0) {
            float * dst = co + ((size_t) ((grp - 1) * QW + qw) * 64 + lane) * 33;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int v = 0; v < 16; ++v) dst[mt * 16 + v] = o[mt][v];
            dst[32] = l;
            cm[((grp - 1) * QW + qw) * 64 + lane] = m;
        }
        __syncthreads();
        if (grp > 0) return;
        if constexpr (ONE) {
            float mt_all = m;
#pragma unroll
            for (int gg = 0; gg < KS - 1; ++gg) mt_all = fmaxf(mt_all, cm[(gg * QW + qw) * 64 + lane]);
            const float a0 = m == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f((m - mt_all) * LOG2E);
            l *= a0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int v = 0; v < 16; ++v) o[mt][v] *= a0;
#pragma unroll
            for (int gg = 0; gg < KS - 1; ++gg) {
                const float mg = cm[(gg * QW + qw) * 64 + lane];
                const float ag = mg == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f((mg - mt_all) * LOG2E);
                const float * src = co + ((size_t) (gg * QW + qw) * 64 + lane) * 33;
                l += src[32] * ag;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) o[mt][v] += src[mt * 16 + v] * ag;
            }
        } else {
#pragma unroll
            for (int gg = 0; gg < KS - 1; ++gg) {
                const float * src = co + ((size_t) (gg * QW + qw) * 64 + lane) * 33;
                l += src[32];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) o[mt][v] += src[mt * 16 + v];
            }
        }
    }

    // O^T: lane = query row q0 + i, value columns 32 mt + 8 j + 4 g + r
    const int qg = q0 + i;
    if constexpr (WIDE_OUT) {
        if (!out32) {
            // f16 output through a wavefront-private 4 KB LDS image (32 rows x 128 B, 16-byte chunks XOR-swizzled by the row): as 8-byte
            // stores a wave instruction covered 32 rows x 16 B = 32 partial-line requests and the eight of them per wavefront were
            // request-bound like the GEMM epilogues' (gemm_epi.h); read back 16 B per lane an instruction covers 8 rows x 128 B.
            // Wavefront-private: no barrier, only the wave's own lgkmcnt wait.  The stored values are the plain form's.
            if constexpr (KS == 1) __syncthreads();          // the ring's last tile has been read by every wavefront
            unsigned char * tb = smem + TB_OFF + qw * 4096;
            const float inv = (float) (1.0 / (double) l);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    half4 w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = (_Float16) pin_f32(o[mt][4 * j + r] * inv);
                    *(half4 *) (tb + i * 128 + (((4 * mt + j) ^ (i & 7)) << 4) + g * 8) = w;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = p * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = *(const uint4 *) (tb + row * 128 + ((ch ^ (row & 7)) << 4));
                if (q0 + row < T) *(uint4 *) (out + (size_t) (q0 + row) * S + head * 64 + ch * 8) = v;
            }
            return;
        }
    }
    if (qg < T) {
        const float inv = (float) (1.0 / (double) l);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t at = (size_t) qg * S + head * 64 + mt * 32 + 8 * j + 4 * g;
                if (out32) {
                    // a quantised out-projection quantises the f32 tensor (the reference's KQV_merged is f32): no f16 rounding in between
                    float4 w;
                    w.x = o[mt][4 * j] * inv; w.y = o[mt][4 * j + 1] * inv; w.z = o[mt][4 * j + 2] * inv; w.w = o[mt][4 * j + 3] * inv;
                    *(float4 *) (out32 + at) = w;
                } else {
                    half4 w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = (_Float16) pin_f32(o[mt][4 * j + r] * inv);
                    *(half4 *) (out + at) = w;
                }
            }
    }
}

template <int QW, int KS, bool ONE, int NST>
void launch_attn_enc2(const __half * q, const __half * k, const __half * vt, int T, int Tpad, int S, int H, __half * out,
                      hipStream_t st, int B, float * out32, int qk_rows) {
    static std::atomic<uint64_t> lds_ok{0};
    constexpr size_t ring = (size_t) KS * NST * 16384;
    constexpr size_t extra = (!ONE && KS > 1) ? (size_t) KS * QW * 32 * 4 : 0;
    constexpr size_t comb = KS > 1 ? (size_t) (KS - 1) * QW * 64 * 34 * 4 : 0;
    constexpr size_t smem0 = (ring + extra) > comb ? (ring + extra) : comb;
    static const bool narrow = getenv("WMI_ATTN_NARROW_STORES") != nullptr;       // A/B knob: 8-byte output stores
    // A/B knob: 0 = launch order, 1 = XCD runs.  Default: XCD runs for several chunks (operand fetches 213 -> ~40 MB per launch at 8 chunks;
    // time unchanged within the noise: the Infinity Cache was serving the re-fetches), launch order for one chunk (14.3 against 15.0 us)
    static const int xcd_env = getenv("WMI_ATTN_XCD") ? atoi(getenv("WMI_ATTN_XCD")) : -1;
    static const int prio_env = getenv("WMI_ATTN_PRIO") ? atoi(getenv("WMI_ATTN_PRIO")) : 0;      // A/B knob: s_setprio 1 around the MFMA clusters
    const int xcd_order = (xcd_env >= 0 ? (xcd_env & 1) : (B > 1)) | (prio_env ? 2 : 0);
    if (narrow) {
        static_assert(smem0 <= 160 * 1024, "LDS");
        if (smem0 > 48 * 1024) allow_full_lds((const void *) k_attn_enc2<QW, KS, ONE, NST, false>, lds_ok);
        hipLaunchKernelGGL((k_attn_enc2<QW, KS, ONE, NST, false>), dim3((T + QW * 32 - 1) / (QW * 32), H, B), dim3(QW * KS * 64), smem0, st,
                           q, k, vt, T, Tpad, S, out, out32, xcd_order, qk_rows);
        return;
    }
    constexpr size_t smem = KS == 1 ? (smem0 > (size_t) QW * 4096 ? smem0 : (size_t) QW * 4096) : smem0 + (size_t) QW * 4096;      // the query wavefronts' output images: inside the ring (one key group) or behind it
    static_assert(smem <= 160 * 1024, "LDS");
    static std::atomic<uint64_t> lds_ok_w{0};
    if (smem > 48 * 1024) allow_full_lds((const void *) k_attn_enc2<QW, KS, ONE, NST, true>, lds_ok_w);
    hipLaunchKernelGGL((k_attn_enc2<QW, KS, ONE, NST, true>), dim3((T + QW * 32 - 1) / (QW * 32), H, B), dim3(QW * KS * 64), smem, st,
                       q, k, vt, T, Tpad, S, out, out32, xcd_order, qk_rows);
}

}  // namespace


void attn_encoder2(const __half * q, const __half * k, const __half * vt, int T, int Tpad, int S, int H, __half * out, hipStream_t st,
                   int B, float * out32, bool one_sweep, bool split, int qk_chunk_rows) {
    const int qk_rows = qk_chunk_rows > 0 ? qk_chunk_rows : T;
    // one key group wherever the result must not depend on how many chunks share the launch (lock-step "exact" mode) and
    // wherever the grid fills the chip by itself; four key groups of two wavefronts for one or two chunks
    // WMI_ATTN_CFG = <wavefronts per key group><key groups><ring depth>, e.g. 242 (A/B knob; the defaults are the measured best)
    static const int cfg_env = getenv("WMI_ATTN_CFG") ? atoi(getenv("WMI_ATTN_CFG")) : 0;
    const int cfg = cfg_env ? cfg_env : (split ? 242 : 412);
#define WMI_ATTN_CASE(C, QW, KS, NST) case C: if (one_sweep) launch_attn_enc2<QW, KS, true, NST>(q, k, vt, T, Tpad, S, H, out, st, B, out32, qk_rows); \
                                              else           launch_attn_enc2<QW, KS, false, NST>(q, k, vt, T, Tpad, S, H, out, st, B, out32, qk_rows); break;
    switch (cfg) {
        WMI_ATTN_CASE(242, 2, 4, 2)
        WMI_ATTN_CASE(223, 2, 2, 3)
        WMI_ATTN_CASE(224, 2, 2, 4)
        WMI_ATTN_CASE(423, 4, 2, 3)
        WMI_ATTN_CASE(422, 4, 2, 2)
        WMI_ATTN_CASE(222, 2, 2, 2)
        WMI_ATTN_CASE(413, 4, 1, 3)
        WMI_ATTN_CASE(414, 4, 1, 4)
        WMI_ATTN_CASE(213, 2, 1, 3)
        default:
        WMI_ATTN_CASE(412, 4, 1, 2)
    }
#undef WMI_ATTN_CASE
}

}}  // namespace wmi::k
