// Attention kernels for gfx950.
//
// Numerics contract (SURVEY App. B rules 4, 5, 7): scores = f16 Q . f16 K accumulated in f32; row
// max in f32; e = f16(exp(f16(s - max))) — the reference's exp goes through an f16 table, i.e. the
// argument AND the result are rounded to f16; probabilities enter the P.V product as f16.
// To keep that contract without materialising the T x T score matrix (72 MB per layer for base.en)
// the encoder kernel walks the keys twice: sweep 1 finds the exact row max, sweep 2 recomputes the
// scores, forms e with the reference's two roundings and accumulates e.V; the 1/sum is applied to the
// f32 accumulator at the end.  The only departure from the reference is that e is not multiplied by
// 1/sum and re-rounded to f16 before P.V (one rounding fewer).
//
// Encoder kernel layout: workgroup = 4 wavefronts = 64 query rows of one head, 16 rows per
// wavefront.  K and V^T tiles of 64 keys are staged in LDS (XOR-swizzled 128-byte rows) and shared
// by the four wavefronts.  QK^T is computed transposed (A = K tile, B = Q) so that each lane ends up
// holding scores of ONE query row (q = lane & 15): the row reductions are in-register plus two
// cross-group shuffles, and the e values are already in the A-operand slot order of the P.V MFMA
// (slot (g, j) <-> key 32*ks + 4*g + j for j < 4, + 16 for j >= 4; V^T fragments are gathered to match).

#include "kernels.h"
#include "wave_ops.h"
#include "xattn_tail.h"
#include <cstdlib>
#include <atomic>

namespace wmi { namespace k {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float    floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float round_f16(float x) { return __half2float(f2h(x)); }
// the reference's exp-through-f16-table (W/ggml.c:11176-11186)
__device__ __forceinline__ float exp16(float d) { return round_f16(expf(round_f16(d))); }
// encoder variant: hardware exp2 path (v_exp_f32, ~2 ulp) — 18 M evaluations per layer make the libm expf the
// longest VALU chain of the kernel; after the f16 rounding the two agree except on ~0.2 % of (tiny) entries
__device__ __forceinline__ float exp16_fast(float d) { return round_f16(__expf(round_f16(d))); }

__device__ __forceinline__ uint32_t lds_off(int row, int chunk) { return (uint32_t) (row * 128 + ((chunk ^ (row & 7)) << 4)); }

// KS = 2: two groups of NW wavefronts share the 64 query rows and take one half of the keys each (own K / V^T tiles,
// common barriers); the halves exchange the row maxima after sweep 1 (the soft-max stays exact) and add their partial
// sums and outputs at the end.  Used when the grid is small (one or two chunks: 192 workgroups on 256 CUs at one wave per
// SIMD): twice the wavefronts per CU let the MFMA and the exp / packing VALU work of different waves overlap.
template <int NW, int KS>
__global__ __launch_bounds__(NW * 64 * KS) void k_attn_enc(const __half * __restrict__ q, const __half * __restrict__ k,
                                                  const __half * __restrict__ vt, int T, int Tpad, int S, float scale,
                                                  __half * __restrict__ out, float * __restrict__ out32, int qk_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char sK_[KS][64 * 128];
    __shared__ __attribute__((aligned(16))) unsigned char sV_[KS][64 * 128];
    const int half = KS == 1 ? 0 : (int) (threadIdx.x / (NW * 64));       // key half of this wavefront group
    unsigned char * sK = sK_[half], * sV = sV_[half];
    const int tid = threadIdx.x - half * (NW * 64), lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int head = blockIdx.y;
    const int q0 = blockIdx.x * (NW * 16) + wave * 16;
    {   // chunk (lane of a batched encode): activations are [B][T][S], V^T is [B][S][Tpad]
        const size_t zb = blockIdx.z;
        q += zb * (size_t) qk_rows * S; k += zb * (size_t) qk_rows * S; vt += zb * (size_t) S * Tpad;
        if (out32) out32 += zb * (size_t) T * S; else out += zb * (size_t) T * S;
    }

    half8 qf[2];
    {
        int qr = q0 + fr; if (qr > T - 1) qr = T - 1;
        const __half * qp = q + (size_t) qr * S + head * 64 + fq * 8;
        qf[0] = *(const half8 *) (qp);
        qf[1] = *(const half8 *) (qp + 32);
    }
    // staging: 64 rows x 8 chunks of 16 B = 512 chunks, (512 / threads) consecutive chunks per thread
    constexpr int CPT = 512 / (NW * 64);
    const int srow = (tid * CPT) >> 3, sch = (tid * CPT) & 7;

    // global -> registers -> LDS staging, split so that the NEXT tile's loads are in flight while the current tile
    // is multiplied (the tiles live in L2 — K and V^T of one head are 2 x 192 KB — but an unhidden L2 round trip
    // per 64 keys was ~60 % of this kernel)
    // Staging registers are individually named on purpose: an indexed uint4[CPT] here ended up in scratch memory
    // (80-144 B/lane, 50 -> 76 us per layer) even with fully unrolled constant indices.
    uint4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
#define LOAD_K(kt0_) do { int kr_ = (kt0_) + srow; if (kr_ > T - 1) kr_ = T - 1;                                   \
        const uint4 * src_ = (const uint4 *) (k + (size_t) kr_ * S + head * 64 + sch * 8);                          \
        rk0 = src_[0]; rk1 = src_[1]; if constexpr (CPT == 4) { rk2 = src_[2]; rk3 = src_[3]; } } while (0)
#define LOAD_V(kt0_) do { int kv_ = (kt0_); if (kv_ > Tpad - 64) kv_ = Tpad - 64;                                     \
        const uint4 * src_ = (const uint4 *) (vt + (size_t) (head * 64 + srow) * Tpad + kv_ + sch * 8);              \
        rv0 = src_[0]; rv1 = src_[1]; if constexpr (CPT == 4) { rv2 = src_[2]; rv3 = src_[3]; } } while (0)
#define STORE_K() do { *(uint4 *) (sK + lds_off(srow, sch)) = rk0; *(uint4 *) (sK + lds_off(srow, sch + 1)) = rk1;   \
        if constexpr (CPT == 4) { *(uint4 *) (sK + lds_off(srow, sch + 2)) = rk2; *(uint4 *) (sK + lds_off(srow, sch + 3)) = rk3; } } while (0)
// V^T tile: its fragments are 8-byte reads, 16 rows x {lower, upper half of a 16-byte chunk} per half-wave (ds_read_b64: lanes 0-31
// in one LDS cycle, bank = (a / 4) mod 64); rows r and r + 8 share parity and slot, i.e. banks (PMC: SQ_LDS_BANK_CONFLICT = 26 % of
// SQ_LDS_IDX_ACTIVE).  Rows 8-15 of every 16 therefore keep the two halves of each chunk swapped: same 16-byte stores, reads conflict-free.
#define SWAPV(v_) ((srow & 8) ? make_uint4((v_).z, (v_).w, (v_).x, (v_).y) : (v_))
#define STORE_V() do { *(uint4 *) (sV + lds_off(srow, sch)) = SWAPV(rv0); *(uint4 *) (sV + lds_off(srow, sch + 1)) = SWAPV(rv1);   \
        if constexpr (CPT == 4) { *(uint4 *) (sV + lds_off(srow, sch + 2)) = SWAPV(rv2); *(uint4 *) (sV + lds_off(srow, sch + 3)) = SWAPV(rv3); } } while (0)
    auto score_tile = [&](int kt) -> floatx4 {                // S^T for keys kt*16..+15 of the staged tile
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const half8 kf = *(const half8 *) (sK + lds_off(kt * 16 + fr, kk * 4 + fq));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], acc, 0, 0, 0);
        }
        return acc;                                           // acc[r] = S[q = fr][key = kt*16 + fq*4 + r]
    };

    // keys of this group: [kbeg, kend), tile-aligned split; both groups run the same number of tiles (common barriers)
    const int ntile_all = (T + 63) / 64, ntile_h = (ntile_all + KS - 1) / KS;
    const int kbeg = half * ntile_h * 64, kend_loop = kbeg + ntile_h * 64;

    // ---- sweep 1: exact row max
    float m = -INFINITY;
    LOAD_K(kbeg);
    for (int kt0 = kbeg; kt0 < kend_loop; kt0 += 64) {
        STORE_K();
        __syncthreads();
        if (kt0 + 64 < kend_loop) LOAD_K(kt0 + 64);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const floatx4 acc = score_tile(kt);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt0 + kt * 16 + fq * 4 + r;
                if (key < T) m = fmaxf(m, acc[r] * scale);
            }
        }
        __syncthreads();
    }
    m = fmaxf(m, WMI_SHX(m, 16));
    m = fmaxf(m, WMI_SHX(m, 32));
    if (KS == 2) {                                           // exact row max over both key halves
        __shared__ float s_m[2][NW][16];
        if (lane < 16) s_m[half][wave][lane] = m;
        __syncthreads();
        m = fmaxf(s_m[0][wave][fr], s_m[1][wave][fr]);
    }

    // ---- sweep 2: e = exp16(s - max), l = sum e, O += e . V
    float l = 0.0f;
    floatx4 o[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) o[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    LOAD_K(kbeg); LOAD_V(kbeg);
    for (int kt0 = kbeg; kt0 < kend_loop; kt0 += 64) {
        STORE_K(); STORE_V();
        __syncthreads();
        if (kt0 + 64 < kend_loop) { LOAD_K(kt0 + 64); LOAD_V(kt0 + 64); }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 pf;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int kt = ks * 2 + hh;
                const floatx4 acc = score_tile(kt);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt0 + kt * 16 + fq * 4 + r;
                    const float e = key < T ? exp16_fast(acc[r] * scale - m) : 0.0f;
                    l += e;
                    pf[hh * 4 + r] = (_Float16) e;
                }
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int dv = nt * 16 + fr;
                const int ch = ks * 4 + (fq & 1);                             // V^T key order: vt_pos() (bits 2 and 3 of the key swapped)
                const int hv = ((fq >> 1) ^ ((fr >> 3) & 1)) * 8;             // rows 8-15: halves swapped (STORE_V)
                const half4 v0 = *(const half4 *) (sV + lds_off(dv, ch) + hv);
                const half4 v1 = *(const half4 *) (sV + lds_off(dv, ch + 2) + hv);
                half8 vf;
                vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
                o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vf, o[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    l += WMI_SHX(l, 16);
    l += WMI_SHX(l, 32);
    if (KS == 2) {                                           // second half hands its partial sums and outputs to the first
        __shared__ float s_o[NW][64][17];
        if (half == 1) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_o[wave][lane][nt * 4 + r] = o[nt][r];
            s_o[wave][lane][16] = l;
        }
        __syncthreads();
        if (half == 1) return;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[nt][r] += s_o[wave][lane][nt * 4 + r];
        l += s_o[wave][lane][16];
    }
    const float inv = (float) (1.0 / (double) l);

    // o[nt][r]: query row fq*4 + r, value column nt*16 + fr
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = fq * 4 + r;
        const float li = __shfl(inv, qrow);
        const int qg = q0 + qrow;
        if (qg < T) {
            // a quantised out-projection quantises the f32 tensor (the reference's KQV_merged is f32): no f16 rounding in between
            if (out32) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) out32[(size_t) qg * S + head * 64 + nt * 16 + fr] = o[nt][r] * li;
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    out[(size_t) qg * S + head * 64 + nt * 16 + fr] = f2h(o[nt][r] * li);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Decoder attention: one (token, head) per workgroup; keys/values come from an f16 cache laid out
// [cell][S].  Exactly the reference's three steps (scores + mask, soft-max through exp16, P rounded to
// f16, P.V) — the score row (<= 1536 floats) lives in LDS.  HBM-bound: reads 2 * n_kv * 128 B.
__global__ __launch_bounds__(256) void k_attn_dec(const __half * __restrict__ q, int S, const __half * __restrict__ kc,
                                                  const __half * __restrict__ vc, int n_kv_arg,
                                                  const float * __restrict__ mask, int ld_mask, __half * __restrict__ out,
                                                  const int32_t * __restrict__ n_kv_dev, int n_kv_cap, float * __restrict__ out32) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_kv = n_kv_dev ? *n_kv_dev : n_kv_arg;
    float * sc  = (float *) smem;                 // [n_kv]  (sized for n_kv_cap under graph replay)
    float * qs  = sc + (((n_kv_dev ? n_kv_cap : n_kv) + 3) & ~3);   // [64]
    float * red = qs + 64;                        // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x, head = blockIdx.y;

    if (tid < 64) qs[tid] = __half2float(q[(size_t) i * S + head * 64 + tid]);
    __syncthreads();

    float lmax = -INFINITY;
    for (int j = tid; j < n_kv; j += 256) {
        const uint4 * kp = (const uint4 *) (kc + (size_t) j * S + head * 64);
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 u = kp[c];
            const __half2 * h = (const __half2 *) &u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(h[e]);
                dot = fmaf(f.x, qs[c * 8 + e * 2], dot);
                dot = fmaf(f.y, qs[c * 8 + e * 2 + 1], dot);
            }
        }
        const float s = mask ? dot + mask[(size_t) i * ld_mask + j] : dot;
        sc[j] = s;
        lmax = fmaxf(lmax, s);
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, WMI_SHX(lmax, o));
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();

    float lsum = 0.0f;
    for (int j = tid; j < n_kv; j += 256) {
        const float s = sc[j];
        const float e = (s == -INFINITY) ? 0.0f : exp16(s - m);
        sc[j] = e;
        lsum += e;
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) lsum += WMI_SHX(lsum, o);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    const float inv = (float) (1.0 / ((double) red[0] + (double) red[1] + (double) red[2] + (double) red[3]));
    __syncthreads();
    for (int j = tid; j < n_kv; j += 256) sc[j] = round_f16(sc[j] * inv);   // P enters P.V as f16 (App. B rule 1)
    __syncthreads();

    // P.V: lane <-> value column, wavefront <-> key residue class
    float acc = 0.0f;
    const __half * vp = vc + head * 64 + lane;
    // eight value rows requested together (a row per trip was one dependent L2 round trip per key: ~25 in a row for the 100 cells of a
    // five-beam step); accumulated in the same key order
    for (int j0 = wave; j0 < n_kv; j0 += 32) {
        __half vv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int jj = j0 + 4 * t; vv[t] = vp[(size_t) (jj < n_kv ? jj : j0) * S]; }
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int jj = j0 + 4 * t; if (jj < n_kv) acc = fmaf(sc[jj], __half2float(vv[t]), acc); }
    }
    red[wave * 64 + lane] = acc;
    __syncthreads();
    if (tid < 64) {
        const float r = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
        if (out32) out32[(size_t) i * S + head * 64 + tid] = r;
        else       out[(size_t) i * S + head * 64 + tid] = f2h(r);
    }
}

// ------------------------------------------------------------------------------------------------
// Decoder CROSS-attention, split over the key axis.  A decode step at batch 1 has only n_tokens x H = 8
// (token, head) pairs, far too few workgroups for 256 CUs, while each pair has to stream 2 x T x 128 B of
// K/V (384 KB at T = 1500).  The keys are therefore cut into NS slices handled by separate workgroups:
//   pass 1  scores s = q.k for the slice (one key per lane: 8 independent 16-byte loads), slice max
//   pass 2  m = max over slice maxima (exact global max), e = exp16(s - m), partial sum and partial e.V
//   pass 3  combine the NS partial sums / outputs, normalise, round to f16
// i.e. the reference's soft-max with its two f16 roundings, evaluated without a serial pass over T.
constexpr int XS_MAX_SLICES = 16;

__global__ __launch_bounds__(256) void k_xattn_scores(const __half * __restrict__ q, int S, const __half * __restrict__ kc,
                                                      int T, int ks, int ns, float * __restrict__ sc, int ld_sc,
                                                      float * __restrict__ pmax, int64_t kv_row_stride) {
    __shared__ float qs[64];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = blockIdx.x, head = blockIdx.y, i = blockIdx.z, H = gridDim.y;
    if (tid < 64) qs[tid] = __half2float(q[(size_t) i * S + head * 64 + tid]);
    kc += (int64_t) i * kv_row_stride;                      // lock-step chunks: row i attends to its own chunk's cross cache
    __syncthreads();
    float lmax = -INFINITY;
    for (int t = tid; t < ks; t += 256) {
        const int j = slice * ks + t;
        if (j >= T) break;
        const uint4 * kp = (const uint4 *) (kc + (size_t) j * S + head * 64);
        uint4 u[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) u[c] = kp[c];
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const __half2 * h = (const __half2 *) &u[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(h[e]);
                dot = fmaf(f.x, qs[c * 8 + e * 2], dot);
                dot = fmaf(f.y, qs[c * 8 + e * 2 + 1], dot);
            }
        }
        sc[((size_t) i * H + head) * ld_sc + j] = dot;
        lmax = fmaxf(lmax, dot);
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, WMI_SHX(lmax, o));
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    if (tid == 0) pmax[((size_t) i * H + head) * ns + slice] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// k_xattn_scores with the cross-attention query projection folded in (S <= 512: tiny / base): every (slice, head)
// workgroup recomputes LN2(x) and its head's 64 rows of W_cq — 64 KB of weights out of L2 — instead of waiting for a
// separate projection launch.  No LDS round trip in the projection: every wavefront normalises the row in registers
// (lane holds x[8 lane .. 8 lane + 8), the slice its dot products need), loads its 16 weight rows in one go before
// the LayerNorm, and reduces the 16 partial dot products with a halving exchange (17 shuffles instead of 96).
// Roundings as in k_gemv + EPI_Q_SCALED (LN output f16, f32 accumulation, (dot + b) * scale -> f16); the f32
// summation order differs from that kernel's.
template <int NC>                                              // 512-column chunks of the row: S <= 512 NC
__global__ __launch_bounds__(256) void k_xattn_qscores(const float * __restrict__ x32, const float * __restrict__ ln_g,
                                                       const float * __restrict__ ln_b, float eps,
                                                       const __half * __restrict__ wq, const float * __restrict__ bq, float qscale,
                                                       int S, const __half * __restrict__ kc, int T, int ks, int ns,
                                                       float * __restrict__ sc, int ld_sc, float * __restrict__ pmax,
                                                       int64_t kv_row_stride) {
    __shared__ float qs[64];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = blockIdx.x, head = blockIdx.y, i = blockIdx.z, H = gridDim.y;
    kc += (int64_t) i * kv_row_stride;
    bool on[NC]; int c0[NC];                                       // S multiple of 8
#pragma unroll
    for (int t = 0; t < NC; ++t) { on[t] = lane * 8 + 512 * t < S; c0[t] = on[t] ? lane * 8 + 512 * t : 0; }

    // 16 weight rows of this wavefront (independent of x: requested first), then x, gain, bias
    // Loads are unconditional from a clamped column and masked afterwards: written as `on ? load : 0` hipcc put every load
    // in its own exec-masked block — 24 scalar dword loads for x / gain / bias and a vmcnt(0) after the second weight row
    uint4 w[NC][16];
    const __half * wrow0 = wq + (size_t) (head * 64 + wave * 16) * S;
#pragma unroll
    for (int t = 0; t < NC; ++t)
#pragma unroll
        for (int u = 0; u < 16; ++u) w[t][u] = *(const uint4 *) (wrow0 + (size_t) u * S + c0[t]);
    float xv[NC][8], gv[NC][8], bv[NC][8];
#pragma unroll
    for (int t = 0; t < NC; ++t) {
        const float * xr = x32 + (size_t) i * S + c0[t];
        const float4 x0 = *(const float4 *) xr, x1 = *(const float4 *) (xr + 4);
        const float4 g0 = *(const float4 *) (ln_g + c0[t]), g1 = *(const float4 *) (ln_g + c0[t] + 4);
        const float4 b0 = *(const float4 *) (ln_b + c0[t]), b1 = *(const float4 *) (ln_b + c0[t] + 4);
        xv[t][0] = x0.x; xv[t][1] = x0.y; xv[t][2] = x0.z; xv[t][3] = x0.w; xv[t][4] = x1.x; xv[t][5] = x1.y; xv[t][6] = x1.z; xv[t][7] = x1.w;
        gv[t][0] = g0.x; gv[t][1] = g0.y; gv[t][2] = g0.z; gv[t][3] = g0.w; gv[t][4] = g1.x; gv[t][5] = g1.y; gv[t][6] = g1.z; gv[t][7] = g1.w;
        bv[t][0] = b0.x; bv[t][1] = b0.y; bv[t][2] = b0.z; bv[t][3] = b0.w; bv[t][4] = b1.x; bv[t][5] = b1.y; bv[t][6] = b1.z; bv[t][7] = b1.w;
    }
    const float bias = bq ? bq[head * 64 + wave * 16 + ((lane >> 2) & 15)] : 0.0f;
    // Key rows of this wavefront — a quarter of the slice, 8 keys per pass: lane = (key g = lane / 8, 16-byte octet o = lane % 8),
    // so one load instruction covers 8 whole 128-byte rows (a lane per row made every instruction touch 64 cache lines for
    // 16 bytes each: the vector L1 was the bottleneck of the kernel's first 3 us).  Independent of q: requested with everything else.
    constexpr int KPASS = 6;                                        // 4 wavefronts x 6 passes x 8 keys = 192 >= ks for T <= 1536
    const int g = lane >> 3, o = lane & 7;
    const int kpw = (((ks + 3) >> 2) + 7) & ~7;                     // keys per wavefront, whole passes
    const int t0 = wave * kpw;
    uint4 kk[KPASS];
#pragma unroll
    for (int p = 0; p < KPASS; ++p) {
        const int t = t0 + 8 * p + g, j = slice * ks + t;
        const bool ok = 8 * p < kpw && t < ks && j < T;
        kk[p] = *(const uint4 *) (kc + (size_t) (ok ? j : 0) * S + head * 64 + o * 8);
    }
    __builtin_amdgcn_sched_barrier(0);          // keep all loads in flight together (the scheduler would sink them to their uses)
#pragma unroll
    for (int t = 0; t < NC; ++t) {
        if (!on[t]) {
#pragma unroll
            for (int u = 0; u < 16; ++u) w[t][u] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int e = 0; e < 8; ++e) { xv[t][e] = 0.0f; gv[t][e] = 0.0f; bv[t][e] = 0.0f; }
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < NC; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += xv[t][e];
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sum += WMI_SHX(sum, o);
    const float mean = sum / (float) S;
    float sq = 0.0f;
#pragma unroll
    for (int t = 0; t < NC; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) { if (on[t]) { xv[t][e] -= mean; sq += xv[t][e] * xv[t][e]; } }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) sq += WMI_SHX(sq, o);
    const float scl = 1.0f / sqrtf(sq / (float) S + eps);
    float av[NC][8];
#pragma unroll
    for (int t = 0; t < NC; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) av[t][e] = round_f16(__fadd_rn(__fmul_rn(xv[t][e] * scl, gv[t][e]), bv[t][e]));
    float acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        float a = 0.0f;
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            const __half2 * wh = (const __half2 *) &w[t][u];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(wh[e]);
                a = fmaf(f.x, av[t][2 * e], a);
                a = fmaf(f.y, av[t][2 * e + 1], a);
            }
        }
        acc[u] = a;
    }
    // halving exchange: after the steps with masks 32, 16, 8, 4 lane L holds the partial sum of row (L >> 2) & 15 over
    // 4 lanes' worth of columns; masks 2 and 1 finish it
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const bool hi = lane & 32;
        const float keep = hi ? acc[u + 8] : acc[u], send = hi ? acc[u] : acc[u + 8];
        acc[u] = keep + WMI_SHX(send, 32);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool hi = lane & 16;
        const float keep = hi ? acc[u + 4] : acc[u], send = hi ? acc[u] : acc[u + 4];
        acc[u] = keep + WMI_SHX(send, 16);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const bool hi = lane & 8;
        const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2];
        acc[u] = keep + WMI_SHX(send, 8);
    }
    {
        const bool hi = lane & 4;
        const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1];
        acc[0] = keep + WMI_SHX(send, 4);
    }
    acc[0] += WMI_SHX(acc[0], 2);
    acc[0] += WMI_SHX(acc[0], 1);
    if ((lane & 3) == 0) qs[wave * 16 + ((lane >> 2) & 15)] = round_f16((acc[0] + bias) * qscale);
    __syncthreads();

    float qo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qo[e] = qs[o * 8 + e];
    float lmax = -INFINITY;
    for (int p0 = 0; p0 * 8 < kpw; p0 += KPASS) {                   // one trip unless a slice holds more than 192 keys
#pragma unroll
        for (int p = 0; p < KPASS; ++p) {
            const int t = t0 + 8 * (p0 + p) + g, j = slice * ks + t;
            const bool ok = 8 * (p0 + p) < kpw && t < ks && j < T;
            uint4 u = kk[p];
            if (p0 > 0) u = *(const uint4 *) (kc + (size_t) (ok ? j : 0) * S + head * 64 + o * 8);
            const __half2 * h = (const __half2 *) &u;
            float dot = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(h[e]);
                dot = fmaf(f.x, qo[2 * e], dot);
                dot = fmaf(f.y, qo[2 * e + 1], dot);
            }
            dot += WMI_SHX(dot, 1); dot += WMI_SHX(dot, 2); dot += WMI_SHX(dot, 4);     // the 8 octets of a key
            if (ok) {
                if (o == 0) sc[((size_t) i * H + head) * ld_sc + j] = dot;
                lmax = fmaxf(lmax, dot);
            }
        }
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, WMI_SHX(lmax, o));
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    if (tid == 0) pmax[((size_t) i * H + head) * ns + slice] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void k_xattn_pv(const __half * __restrict__ vc, int S, int T, int ks, int ns,
                                                  const float * __restrict__ sc, int ld_sc, const float * __restrict__ pmax,
                                                  float * __restrict__ part_o, float * __restrict__ part_l, int64_t kv_row_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float * e   = (float *) smem;                 // [ks]
    float * red = e + ((ks + 3) & ~3);            // [4][64] partial outputs, then [4] sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = blockIdx.x, head = blockIdx.y, i = blockIdx.z, H = gridDim.y;
    const size_t row = (size_t) i * H + head;
    vc += (int64_t) i * kv_row_stride;
    const int j0 = slice * ks;
    const int cnt = max(0, min(ks, T - j0));
    // e.V: lane = (key sub-index, 16-byte chunk of the 64-wide head slice); 8 keys per wave instruction.  The value rows of this
    // lane (every 32nd key of the slice; at most 8 for slices of <= 256 keys) do not depend on the scores: requested first,
    // unconditionally from a clamped row (as `t < cnt ? load : 0` hipcc split them into three dependent groups)
    const int kg = lane >> 3, ch = lane & 7;
    uint4 vu[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int t = wave * 8 + kg + 32 * it;
        const int jr = min(j0 + (t < cnt ? t : 0), T - 1);
        vu[it] = *(const uint4 *) (vc + (size_t) jr * S + head * 64 + ch * 8);
    }
    float m = -INFINITY;
    if (ns == 8) {                                  // all loads first (a loop makes them eight dependent round trips)
        float pm[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) pm[s2] = pmax[row * 8 + s2];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) m = fmaxf(m, pm[s2]);
    } else {
        for (int s2 = 0; s2 < ns; ++s2) m = fmaxf(m, pmax[row * ns + s2]);
    }
    float lsum = 0.0f;
    for (int t = tid; t < cnt; t += 256) {
        const float v = exp16(sc[row * ld_sc + j0 + t] - m);
        e[t] = v; lsum += v;
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) lsum += WMI_SHX(lsum, o);
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.0f;
    if (cnt <= 256) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int t = wave * 8 + kg + 32 * it;
            if (t < cnt) {
                const __half2 * h = (const __half2 *) &vu[it];
                const float w = e[t];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float2 f = __half22float2(h[p]);
                    acc[2 * p]     = fmaf(w, f.x, acc[2 * p]);
                    acc[2 * p + 1] = fmaf(w, f.y, acc[2 * p + 1]);
                }
            }
        }
    } else
    for (int t = wave * 8 + kg; t < cnt; t += 32) {
        const uint4 u = *(const uint4 *) (vc + (size_t) (j0 + t) * S + head * 64 + ch * 8);
        const __half2 * h = (const __half2 *) &u;
        const float w = e[t];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float2 f = __half22float2(h[p]);
            acc[2 * p]     = fmaf(w, f.x, acc[2 * p]);
            acc[2 * p + 1] = fmaf(w, f.y, acc[2 * p + 1]);
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float v = acc[d];
        v += WMI_SHX(v, 8); v += WMI_SHX(v, 16); v += WMI_SHX(v, 32);
        acc[d] = v;
    }
    if (kg == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) red[wave * 64 + ch * 8 + d] = acc[d];
    }
    __shared__ float lred[4];
    if (lane == 0) lred[wave] = lsum;
    __syncthreads();
    if (tid < 64) part_o[(row * ns + slice) * 64 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    if (tid == 0) part_l[row * ns + slice] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
}

// Scores, soft-max numerators and P.V of a key slice in ONE launch (flash-decoding form): every (slice, head, row) workgroup
// keeps its <= 192 scores in registers, takes the SLICE maximum m_s, e = exp16(s - m_s), and writes (m_s, l_s = sum e,
// o_s = e.V); the consumer rescales by exp(m_s - max m) when it combines (GemvArgs::comb_m, k_xattn_combine).  The two-launch
// form above evaluates the reference's soft-max literally (global maximum before the f16 exponent table); here an element's
// f16 roundings happen relative to its slice's maximum instead — the same two roundings per element, a different argument —
// and the step saves one dependent launch per decoder layer (~3.5 us of ~9).  PROJ: the query projection of k_xattn_qscores
// (LN2(x) . W_cq rows of this head, S <= 512 NC) runs first, else q comes as f16.
// Lane = (key g = lane / 8, 16-byte octet o = lane % 8) for K and V alike; a wavefront owns 6 passes x 8 keys.
template <int NC, bool PROJ>
__global__ __launch_bounds__(256) void k_xattn_fused(const float * __restrict__ x32, const float * __restrict__ ln_g,
                                                     const float * __restrict__ ln_b, float eps,
                                                     const __half * __restrict__ wq, const float * __restrict__ bq, float qscale,
                                                     const __half * __restrict__ q16,
                                                     int S, const __half * __restrict__ kc, const __half * __restrict__ vc, int T, int ks, int ns,
                                                     float * __restrict__ pmax, float * __restrict__ part_o, float * __restrict__ part_l,
                                                     int64_t kv_row_stride, int head_major, const Stamp sp) {
    __shared__ float qs[64];
    __shared__ float red[4], lred[4];
    __shared__ float ored[4][64];
    const unsigned long long ts0 = stamp_t0(sp.base), tc0 = sp.base ? clock64() : 0ull;      // (+ shader-clock counter: the probe derives the effective clock)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // head_major: grid (H, ns, n) — workgroup id % 8 = head % 8, so the 8 key slices of a head share one XCD's L2 and that L2
    // fetches only its head's 64 rows of W_cq (slice-major, every XCD pulled the whole matrix: 2.1x the algorithmic bytes)
    const int slice = head_major ? blockIdx.y : blockIdx.x, head = head_major ? blockIdx.x : blockIdx.y, i = blockIdx.z;
    const int H = head_major ? gridDim.x : gridDim.y;
    const size_t row = (size_t) i * H + head;
    kc += (int64_t) i * kv_row_stride; vc += (int64_t) i * kv_row_stride;
    constexpr int KPASS = 6;                                        // 4 wavefronts x 6 passes x 8 keys = 192 >= ks
    const int g = lane >> 3, o = lane & 7;
    const int kpw = (((ks + 3) >> 2) + 7) & ~7;                     // keys per wavefront, whole passes
    const int t0 = wave * kpw;
    bool ok[KPASS]; size_t off[KPASS];
#pragma unroll
    for (int p = 0; p < KPASS; ++p) {
        const int t = t0 + 8 * p + g, j = slice * ks + t;
        ok[p] = 8 * p < kpw && t < ks && j < T;
        off[p] = (size_t) (ok[p] ? j : 0) * S + head * 64 + o * 8;
    }
    uint4 kk[KPASS], vv[KPASS];
    if constexpr (PROJ) {
        bool on[NC]; int c0[NC];
#pragma unroll
        for (int t = 0; t < NC; ++t) { on[t] = lane * 8 + 512 * t < S; c0[t] = on[t] ? lane * 8 + 512 * t : 0; }
        // Load order (vmcnt retires in order): the residual row first — it was written by the previous launch and is in L2, and the
        // LayerNorm statistics need nothing else — then gain / bias, the 16 query-weight rows, K and V, which come from
        // Infinity Cache / HBM; the statistics run while those are in flight.  (Weights first: LN waited for all 16 rows.)
        float xv[NC][8], gv[NC][8], bv[NC][8];
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            const float * xr = x32 + (size_t) i * S + c0[t];
            const float4 x0 = *(const float4 *) xr, x1 = *(const float4 *) (xr + 4);
            xv[t][0] = x0.x; xv[t][1] = x0.y; xv[t][2] = x0.z; xv[t][3] = x0.w; xv[t][4] = x1.x; xv[t][5] = x1.y; xv[t][6] = x1.z; xv[t][7] = x1.w;
        }
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            const float4 g0 = *(const float4 *) (ln_g + c0[t]), g1 = *(const float4 *) (ln_g + c0[t] + 4);
            const float4 b0 = *(const float4 *) (ln_b + c0[t]), b1 = *(const float4 *) (ln_b + c0[t] + 4);
            gv[t][0] = g0.x; gv[t][1] = g0.y; gv[t][2] = g0.z; gv[t][3] = g0.w; gv[t][4] = g1.x; gv[t][5] = g1.y; gv[t][6] = g1.z; gv[t][7] = g1.w;
            bv[t][0] = b0.x; bv[t][1] = b0.y; bv[t][2] = b0.z; bv[t][3] = b0.w; bv[t][4] = b1.x; bv[t][5] = b1.y; bv[t][6] = b1.z; bv[t][7] = b1.w;
        }
        __builtin_amdgcn_sched_barrier(0);
        uint4 w[NC][16];
        const __half * wrow0 = wq + (size_t) (head * 64 + wave * 16) * S;
#pragma unroll
        for (int t = 0; t < NC; ++t)
#pragma unroll
            for (int u = 0; u < 16; ++u) w[t][u] = *(const uint4 *) (wrow0 + (size_t) u * S + c0[t]);
        // (straight-line: an absent bias reads the weights' first bytes and is masked at its use)
        const float bias_raw = *(bq ? bq + head * 64 + wave * 16 + ((lane >> 2) & 15) : (const float *) wq);
#pragma unroll
        for (int p = 0; p < KPASS; ++p) kk[p] = *(const uint4 *) (kc + off[p]);
        if constexpr (NC == 1) {
#pragma unroll
            for (int p = 0; p < KPASS; ++p) vv[p] = *(const uint4 *) (vc + off[p]);
        }
        __builtin_amdgcn_sched_barrier(0);          // everything this workgroup reads is in flight before the first use
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            if (!on[t]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xv[t][e] = 0.0f; gv[t][e] = 0.0f; bv[t][e] = 0.0f; }
            }
        }
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < NC; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += xv[t][e];
        _Pragma("unroll") for (int x = 32; x > 0; x >>= 1) sum += WMI_SHX(sum, x);
        const float mean = sum / (float) S;
        float sq = 0.0f;
#pragma unroll
        for (int t = 0; t < NC; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) { if (on[t]) { xv[t][e] -= mean; sq += xv[t][e] * xv[t][e]; } }
        _Pragma("unroll") for (int x = 32; x > 0; x >>= 1) sq += WMI_SHX(sq, x);
        const float scl = 1.0f / sqrtf(sq / (float) S + eps);
        float av[NC][8];
#pragma unroll
        for (int t = 0; t < NC; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) av[t][e] = round_f16(__fadd_rn(__fmul_rn(xv[t][e] * scl, gv[t][e]), bv[t][e]));
        __builtin_amdgcn_sched_barrier(0);          // the LayerNorm before anything that waits for the weight rows
#pragma unroll
        for (int t = 0; t < NC; ++t) {               // (columns past S: a select on w is a use of w — it must not sit in front of the LayerNorm)
            if (!on[t]) {
#pragma unroll
                for (int u = 0; u < 16; ++u) w[t][u] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        float acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float a = 0.0f;
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                const __half2 * wh = (const __half2 *) &w[t][u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 f = __half22float2(wh[e]);
                    a = fmaf(f.x, av[t][2 * e], a);
                    a = fmaf(f.y, av[t][2 * e + 1], a);
                }
            }
            acc[u] = a;
        }
        // halving exchange (k_xattn_qscores): lane L ends with the dot product of row (L >> 2) & 15
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool hi = lane & 32;
            const float keep = hi ? acc[u + 8] : acc[u], send = hi ? acc[u] : acc[u + 8];
            acc[u] = keep + WMI_SHX(send, 32);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool hi = lane & 16;
            const float keep = hi ? acc[u + 4] : acc[u], send = hi ? acc[u] : acc[u + 4];
            acc[u] = keep + WMI_SHX(send, 16);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool hi = lane & 8;
            const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2];
            acc[u] = keep + WMI_SHX(send, 8);
        }
        {
            const bool hi = lane & 4;
            const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1];
            acc[0] = keep + WMI_SHX(send, 4);
        }
        acc[0] += WMI_SHX(acc[0], 2);
        acc[0] += WMI_SHX(acc[0], 1);
        const float bias = bq ? bias_raw : 0.0f;
        if ((lane & 3) == 0) qs[wave * 16 + ((lane >> 2) & 15)] = round_f16((acc[0] + bias) * qscale);
        if constexpr (NC > 1) {                      // the projection's weights held the registers: V goes out now, behind the scores
#pragma unroll
            for (int p = 0; p < KPASS; ++p) vv[p] = *(const uint4 *) (vc + off[p]);
        }
    } else {
        if (tid < 64) qs[tid] = __half2float(q16[(size_t) i * S + head * 64 + tid]);
#pragma unroll
        for (int p = 0; p < KPASS; ++p) kk[p] = *(const uint4 *) (kc + off[p]);
#pragma unroll
        for (int p = 0; p < KPASS; ++p) vv[p] = *(const uint4 *) (vc + off[p]);
    }
    __syncthreads();
#if defined(WMI_XA_STAMPS)
    const unsigned long long xa1 = sp.base ? wall_clock64() : 0ull;
#endif

    float qo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qo[e] = qs[o * 8 + e];
    float sv[KPASS];
    float lmax = -INFINITY;
#pragma unroll
    for (int p = 0; p < KPASS; ++p) {
        const __half2 * h = (const __half2 *) &kk[p];
        float dot = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float2 f = __half22float2(h[e]);
            dot = fmaf(f.x, qo[2 * e], dot);
            dot = fmaf(f.y, qo[2 * e + 1], dot);
        }
        dot += WMI_SHX(dot, 1); dot += WMI_SHX(dot, 2); dot += WMI_SHX(dot, 4);     // the 8 octets of a key
        sv[p] = ok[p] ? dot : -INFINITY;
        lmax = fmaxf(lmax, sv[p]);
    }
    _Pragma("unroll") for (int x = 32; x > 4; x >>= 1) lmax = fmaxf(lmax, WMI_SHX(lmax, x));       // the octet lanes of a key agree already
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
#if defined(WMI_XA_STAMPS)
    const unsigned long long xa2 = sp.base ? wall_clock64() : 0ull;
#endif
    float acc[8], lsum = 0.0f;
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.0f;
#pragma unroll
    for (int p = 0; p < KPASS; ++p) {
        const float e = ok[p] ? exp16(sv[p] - m) : 0.0f;
        if (o == 0) lsum += e;
        const __half2 * h = (const __half2 *) &vv[p];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 f = __half22float2(h[q]);
            acc[2 * q]     = fmaf(e, f.x, acc[2 * q]);
            acc[2 * q + 1] = fmaf(e, f.y, acc[2 * q + 1]);
        }
    }
    _Pragma("unroll") for (int x = 32; x > 0; x >>= 1) lsum += WMI_SHX(lsum, x);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float v = acc[d];
        v += WMI_SHX(v, 8); v += WMI_SHX(v, 16); v += WMI_SHX(v, 32);
        acc[d] = v;
    }
    if (g == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) ored[wave][o * 8 + d] = acc[d];
    }
    if (lane == 0) lred[wave] = lsum;
    __syncthreads();
    if (tid < 64) part_o[(row * ns + slice) * 64 + tid] = (ored[0][tid] + ored[1][tid]) + (ored[2][tid] + ored[3][tid]);
    if (tid == 0) { part_l[row * ns + slice] = (lred[0] + lred[1]) + (lred[2] + lred[3]); pmax[row * ns + slice] = m; }
    stamp_end(sp.base, sp.slot, (((int) blockIdx.z * (int) gridDim.y + (int) blockIdx.y) * (int) gridDim.x + (int) blockIdx.x) * 4 + wave, ts0,
#if defined(WMI_XA_STAMPS)
              xa1, xa2);
#else
              tc0 | (1ull << 63), sp.base ? clock64() : 0ull);      // bit 63: a shader-clock pair, not mid points
#endif
}

// ------------------------------------------------------------------------------------------------ the back of the cross-attention, one launch
// LN + cross query + key slices (k_xattn_fused<1, true>), the combine of a head's slices and the out projection + residual
// (k_gemv1<4, 1, false, 3, EPI_F32_BIAS_RESID>) of the one-row step as ONE launch with two few-reader hand-offs (k_front's pattern,
// profiles/r06o_*): a (head, slice) workgroup publishes its partial (m, l, o[64]) as 66 granules {f32, tag}; wavefront 0 of the head's
// slice-0 workgroup gathers the head's ns partials, combines them ONCE (the two-launch form combines in each of its 32 workgroups) and
// publishes the head's 64 values as 32 granules {f16 pair, tag}; the first S / 16 workgroups sweep the S / 2 granules of the row and
// take four rows of the projection per wavefront (weights requested at the start).  Per value the operations and their order are the two
// launches'.  4 <= ns <= 8 key slices, S <= 512.  Tags, spins, status word: MlpPairArgs.  The partials still go to part_o / part_l / pmax.
__global__ __launch_bounds__(256) void k_xback(const XbackArgs a_in, const Stamp sp) {
    // lock-step rows: row z is a one-row problem of its own: shift the per-row operands, everything below is the one-row kernel
    XbackArgs a = a_in;
    {
        const int z = blockIdx.z;
        a.x += (size_t) z * a.S; a.xout += (size_t) z * a.S;
        a.kc += (int64_t) z * a.kv_row_stride; a.vc += (int64_t) z * a.kv_row_stride;
        a.gp += (size_t) z * XBACK_ROW_GRANULES; a.ga += (size_t) z * XBACK_ROW_GRANULES;
    }
    __shared__ float qs[64];
    __shared__ float red[4], lred[4];
    __shared__ float ored[4][64];
    __shared__ __attribute__((aligned(16))) __half act[512];
    const unsigned long long ts0 = stamp_t0(sp.base);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.x, slice = blockIdx.y, H = gridDim.x, ns = a.ns, S = a.S, K = S;
    const int lin = head + H * slice;
    const bool p3 = lin < (S >> 4), comb_wave = slice == 0 && wave == 0;
    const XaKeys keys = xa_keys(slice, a.ks, a.T, S, head, wave, lane);
    int zl = 0; asm volatile("" : "+v"(zl));

    // ---- every load that depends on nothing this launch computes, in one straight line: the row, gain / bias, the 16 query-weight rows of
    // this wavefront, K, V; the projection's four weight rows, bias, residual; the launch tag last (DESIGN hazards 23, 35, 36)
    const bool on = lane * 8 < S; const int c0 = on ? lane * 8 : 0;
    float xv[8], gv[8], bv[8];
    {
        const float * xr = a.x + c0;
        const float4 x0 = *(const float4 *) xr, x1 = *(const float4 *) (xr + 4);
        xv[0] = x0.x; xv[1] = x0.y; xv[2] = x0.z; xv[3] = x0.w; xv[4] = x1.x; xv[5] = x1.y; xv[6] = x1.z; xv[7] = x1.w;
        const float4 g0 = *(const float4 *) (a.ln_g + c0), g1 = *(const float4 *) (a.ln_g + c0 + 4);
        const float4 b0 = *(const float4 *) (a.ln_b + c0), b1 = *(const float4 *) (a.ln_b + c0 + 4);
        gv[0] = g0.x; gv[1] = g0.y; gv[2] = g0.z; gv[3] = g0.w; gv[4] = g1.x; gv[5] = g1.y; gv[6] = g1.z; gv[7] = g1.w;
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    __builtin_amdgcn_sched_barrier(0);
    uint4 w[16];
    const __half * wrow0 = a.wq + (size_t) (head * 64 + wave * 16) * S;
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = *(const uint4 *) (wrow0 + (size_t) u * S + c0);
    const float bias_raw = *(a.bq ? a.bq + head * 64 + wave * 16 + ((lane >> 2) & 15) : (const float *) a.wq);
    uint4 kk[XA_KPASS], vv[XA_KPASS];
#pragma unroll
    for (int p = 0; p < XA_KPASS; ++p) kk[p] = *(const uint4 *) ((const char *) a.kc + keys.off[p]);
#pragma unroll
    for (int p = 0; p < XA_KPASS; ++p) vv[p] = *(const uint4 *) ((const char *) a.vc + keys.off[p]);
    const int gw3 = (p3 ? lin : 0) * 4 + wave, wrow3 = lane >> 4;
    const uint32_t tag_v = a.epoch[a.par + zl] + 1u, other_v = a.epoch[(a.par ^ 1) + zl];
    __builtin_amdgcn_sched_barrier(0);

    // ---- LayerNorm + this head's query (k_xattn_fused<1, true>)
    if (!on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[e] = 0.0f; gv[e] = 0.0f; bv[e] = 0.0f; }
    }
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += xv[e];
    _Pragma("unroll") for (int x = 32; x > 0; x >>= 1) sum += WMI_SHX(sum, x);
    const float mean = sum / (float) S;
    float sq = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { if (on) { xv[e] -= mean; sq += xv[e] * xv[e]; } }
    _Pragma("unroll") for (int x = 32; x > 0; x >>= 1) sq += WMI_SHX(sq, x);
    const float scl = 1.0f / sqrtf(sq / (float) S + a.eps);
    float av[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) av[e] = round_f16(__fadd_rn(__fmul_rn(xv[e] * scl, gv[e]), bv[e]));
    __builtin_amdgcn_sched_barrier(0);
    if (!on) {
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = make_uint4(0u, 0u, 0u, 0u);
    }
    {
        float acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float q = 0.0f;
            const __half2 * wh = (const __half2 *) &w[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(wh[e]);
                q = fmaf(f.x, av[2 * e], q);
                q = fmaf(f.y, av[2 * e + 1], q);
            }
            acc[u] = q;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const bool hi = lane & 32; const float keep = hi ? acc[u + 8] : acc[u], send = hi ? acc[u] : acc[u + 8]; acc[u] = keep + WMI_SHX(send, 32); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const bool hi = lane & 16; const float keep = hi ? acc[u + 4] : acc[u], send = hi ? acc[u] : acc[u + 4]; acc[u] = keep + WMI_SHX(send, 16); }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const bool hi = lane & 8; const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2]; acc[u] = keep + WMI_SHX(send, 8); }
        { const bool hi = lane & 4; const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1]; acc[0] = keep + WMI_SHX(send, 4); }
        acc[0] += WMI_SHX(acc[0], 2);
        acc[0] += WMI_SHX(acc[0], 1);
        const float bias = a.bq ? bias_raw : 0.0f;
        if ((lane & 3) == 0) qs[wave * 16 + ((lane >> 2) & 15)] = round_f16((acc[0] + bias) * a.qscale);
    }
    // the projection's weight rows, bias and residual go out here, where the query's 16 weight rows have left their registers (3 us before
    // their use; 168 registers = three wavefronts per SIMD: lock-step rows need 64 x rows workgroups resident at once).  The residual is
    // read before this workgroup has published anything: no projection of this launch can have overwritten x yet
    uint4 w3[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w3[u] = *(const uint4 *) (a.Wo + (size_t) (gw3 * 4 + u) * K + c0);
    const float bias3 = *(a.bo ? a.bo + gw3 * 4 + wrow3 : (const float *) a.Wo);
    const float resid3 = a.x[gw3 * 4 + wrow3];
    __syncthreads();
    const unsigned long long tm1 = stamp_t0(sp.base);
    // ---- this workgroup's key slice (xattn_tail.h), its partial to memory as before and as 66 granules {f32, tag}
    xa_slice_tail(qs, kk, vv, keys.ok, red, lred, ored, (size_t) blockIdx.z * H + head, ns, slice, a.pmax, a.part_o, a.part_l);
    const uint32_t tag = __builtin_amdgcn_readfirstlane(tag_v);
    if (lin == 0 && blockIdx.z == 0 && tid == 0) {
        if (other_v >= tag) __hip_atomic_fetch_or(a.fault, PAIR_FAULT_PARITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.epoch[a.par ^ 1] = tag;
    }
    {
        unsigned long long * gp = a.gp + (size_t) (head * ns + slice) * 66;
        if (tid < 64 && lin + 1 != a.withhold) {
            const float po = (ored[0][tid] + ored[1][tid]) + (ored[2][tid] + ored[3][tid]);
            __hip_atomic_store(gp + tid, ((unsigned long long) tag << 32) | (unsigned long long) __float_as_uint(po), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 64) {
            const float pl = (lred[0] + lred[1]) + (lred[2] + lred[3]);
            __hip_atomic_store(gp + 64, ((unsigned long long) tag << 32) | (unsigned long long) __float_as_uint(pl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 65) {
            const float pm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            __hip_atomic_store(gp + 65, ((unsigned long long) tag << 32) | (unsigned long long) __float_as_uint(pm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const uint32_t spin_cap = a.spin_cap ? a.spin_cap : (1u << 20);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    unsigned long long tm2 = 0;
    // ---- the head's slices combined once (k_gemv1's combine prologue: maximum of the slice maxima, w = exp(m_s - M), o and l (double) in slice order)
    if (comb_wave) {
        const unsigned long long * hp = a.gp + (size_t) head * ns * 66;
        uint32_t spins = 0; bool landed = false;
        u32x2 po[8]; u32x4 lm[8];
        for (; spins < spin_cap; ++spins) {
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const int sc = s2 < ns ? s2 : 0;
                asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=&v"(po[s2]) : "v"(hp + sc * 66 + lane) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(lm[s2]) : "v"(hp + sc * 66 + 64) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(po[0]), "+v"(po[1]), "+v"(po[2]), "+v"(po[3]), "+v"(po[4]), "+v"(po[5]), "+v"(po[6]), "+v"(po[7]) :: "memory");
            asm volatile("" : "+v"(lm[0]), "+v"(lm[1]), "+v"(lm[2]), "+v"(lm[3]), "+v"(lm[4]), "+v"(lm[5]), "+v"(lm[6]), "+v"(lm[7]));
            bool ok = true;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) ok = ok && po[s2][1] == tag && lm[s2][1] == tag && lm[s2][3] == tag;
            if (__all(ok)) { landed = true; break; }
        }
        if (lane == 0 && (!landed || spins > PAIR_SLOW_POLLS))
            __hip_atomic_fetch_or(a.fault, landed ? PAIR_SLOW : PAIR_FAULT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float o = 0.0f, M = -INFINITY; double l = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) if (s2 < ns) M = fmaxf(M, __uint_as_float(lm[s2][2]));
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            if (s2 < ns) {
                const float pm = __uint_as_float(lm[s2][2]);
                const float wgt = pm > -INFINITY ? __expf(pm - M) : 0.0f;
                o += __uint_as_float(po[s2][0]) * wgt; l += (double) __uint_as_float(lm[s2][0]) * (double) wgt;
            }
        }
        const __half hv = f2h(o * (float) (1.0 / l));
        const uint32_t mine = (uint32_t) __half_as_ushort(hv), other = (uint32_t) WMI_SHX((int) mine, 1);
        if (!(lane & 1))
            __hip_atomic_store(a.ga + (head * 32 + (lane >> 1)), ((unsigned long long) tag << 32) | (unsigned long long) (mine | (other << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tm2 = stamp_t0(sp.base);
    }
    // ---- the out projection + residual on the first S / 16 workgroups: the row swept once per workgroup, four rows per wavefront
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
        uint4 u4 = *(const uint4 *) (act + c0);
        if (!on) u4 = make_uint4(0u, 0u, 0u, 0u);
        float a3[8];
        {
            const __half2 * h = (const __half2 *) &u4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); a3[2 * e] = f.x; a3[2 * e + 1] = f.y; }
        }
        float acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[u] = 0.0f;
            const __half2 * h = (const __half2 *) &w3[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 f = __half22float2(h[e]);
                acc[u] = fmaf(f.x, a3[2 * e], acc[u]);
                acc[u] = fmaf(f.y, a3[2 * e + 1], acc[u]);
            }
        }
        float v;
#pragma unroll
        for (int u = 0; u < 2; ++u) { const bool hi = lane & 32; const float keep = hi ? acc[u + 2] : acc[u], send = hi ? acc[u] : acc[u + 2]; acc[u] = keep + WMI_SHX(send, 32); }
        { const bool hi = lane & 16; const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1]; v = keep + WMI_SHX(send, 16); }
        v += WMI_SHX(v, 8); v += WMI_SHX(v, 4); v += WMI_SHX(v, 2); v += WMI_SHX(v, 1);
        if ((lane & 15) == 0) {
            const float bias = a.bo ? bias3 : 0.0f;
            a.xout[gw3 * 4 + wrow3] = (v + bias) + resid3;
        }
    }
    stamp_end(sp.base, sp.slot, ((int) blockIdx.z * (int) gridDim.x * (int) gridDim.y + lin) * 4 + wave, ts0, tm1, tm2);
}

// weight of slice s2 when partials are relative to their own slice maximum (part_m != null), else 1
__device__ __forceinline__ float slice_weight(const float * part_m, size_t base, int ns, int s2, float M) {
    (void) ns;
    const float ms = part_m[base + s2];
    return ms > -INFINITY ? __expf(ms - M) : 0.0f;
}

__global__ __launch_bounds__(64) void k_xattn_combine(const float * __restrict__ part_o, const float * __restrict__ part_l,
                                                      const float * __restrict__ part_m,
                                                      int ns, int S, __half * __restrict__ out, float * __restrict__ out32) {
    const int head = blockIdx.x, i = blockIdx.y, H = gridDim.x, d = threadIdx.x;
    const size_t row = (size_t) i * H + head;
    float o = 0.0f; double l = 0.0;
    if (part_m) {
        float M = -INFINITY;
        for (int s2 = 0; s2 < ns; ++s2) M = fmaxf(M, part_m[row * ns + s2]);
        for (int s2 = 0; s2 < ns; ++s2) {
            const float w = slice_weight(part_m, row * ns, ns, s2, M);
            o += part_o[(row * ns + s2) * 64 + d] * w; l += (double) part_l[row * ns + s2] * (double) w;
        }
    } else if (ns == 8) {
        float po[8], pl[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) { po[s2] = part_o[(row * 8 + s2) * 64 + d]; pl[s2] = part_l[row * 8 + s2]; }
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) { o += po[s2]; l += (double) pl[s2]; }
    } else {
        for (int s2 = 0; s2 < ns; ++s2) { o += part_o[(row * ns + s2) * 64 + d]; l += (double) part_l[row * ns + s2]; }
    }
    if (out32) out32[(size_t) i * S + head * 64 + d] = o * (float) (1.0 / l);
    else       out[(size_t) i * S + head * 64 + d] = f2h(o * (float) (1.0 / l));
}

} // namespace

// scratch layout: scores [n][H][Tpad] | pmax [n][H][NS] | part_l [n][H][NS] | part_o [n][H][NS][64]
struct XLayout { int ns, ks, ld_sc; float * sc, * pmax, * part_l, * part_o; bool fused; };
static XLayout xattn_layout(int n, int H, int T, float * scratch) {
    // WMI_XATTN_TWO_PASS: the literal two-launch form (global maximum before the f16 exponent), kept for A/B and as the S-agnostic fall-back
    static const bool two_pass = getenv("WMI_XATTN_TWO_PASS") != nullptr;
    XLayout L;
    L.ns = (T + 191) / 192; if (L.ns < 1) L.ns = 1; if (L.ns > XS_MAX_SLICES) L.ns = XS_MAX_SLICES;
    L.ks = (T + L.ns - 1) / L.ns;
    L.ld_sc = (T + 63) & ~63;
    L.sc = scratch;
    L.pmax = L.sc + (size_t) n * H * L.ld_sc;
    L.part_l = L.pmax + (size_t) n * H * L.ns;
    L.part_o = L.part_l + (size_t) n * H * L.ns;
    L.fused = !two_pass && L.ks <= 192;               // one pass keeps a slice's scores in registers: 4 wavefronts x 6 x 8 keys
    return L;
}
static bool xattn_head_major() { static const bool off = getenv("WMI_XATTN_SLICE_MAJOR") != nullptr; return !off; }      // A/B knob
static dim3 xattn_grid(int ns, int H, int n) { return xattn_head_major() ? dim3(H, ns, n) : dim3(ns, H, n); }
static int g_xattn_probe_skip = 0;        // probe only: bit 0 skips the score kernel, bit 1 the P.V kernel (two-pass form)
void set_xattn_probe_skip(int mask) { g_xattn_probe_skip = mask; }

static void xattn_run(const XLayout & L, const __half * q, int n, int S, int H, const __half * kc, const __half * vc, int T,
                      hipStream_t st, int64_t kv_row_stride) {
    if (L.fused) {
        hipLaunchKernelGGL((k_xattn_fused<1, false>), xattn_grid(L.ns, H, n), dim3(256), 0, st, nullptr, nullptr, nullptr, 0.0f, nullptr, nullptr, 0.0f,
                           q, S, kc, vc, T, L.ks, L.ns, L.pmax, L.part_o, L.part_l, kv_row_stride, xattn_head_major() ? 1 : 0, stamp_next());
        return;
    }
    hipLaunchKernelGGL(k_xattn_scores, dim3(L.ns, H, n), dim3(256), 0, st, q, S, kc, T, L.ks, L.ns, L.sc, L.ld_sc, L.pmax, kv_row_stride);
    const size_t smem = (((size_t) L.ks + 3) & ~(size_t) 3) * 4 + 4 * 64 * 4;
    hipLaunchKernelGGL(k_xattn_pv, dim3(L.ns, H, n), dim3(256), smem, st, vc, S, T, L.ks, L.ns, L.sc, L.ld_sc, L.pmax, L.part_o, L.part_l, kv_row_stride);
}

void attn_cross_split(const __half * q, int n, int S, int H, const __half * kc, const __half * vc, int T,
                      float * scratch, __half * out, hipStream_t st, int64_t kv_row_stride, float * out32) {
    const XLayout L = xattn_layout(n, H, T, scratch);
    xattn_run(L, q, n, S, H, kc, vc, T, st, kv_row_stride);
    hipLaunchKernelGGL(k_xattn_combine, dim3(H, n), dim3(64), 0, st, L.part_o, L.part_l, L.fused ? L.pmax : nullptr, L.ns, S, out, out32);
}

void attn_cross_split_partials(const __half * q, int n, int S, int H, const __half * kc, const __half * vc, int T,
                               float * scratch, const float ** po, const float ** pl, const float ** pm, int * pns, hipStream_t st,
                               int64_t kv_row_stride) {
    const XLayout L = xattn_layout(n, H, T, scratch);
    xattn_run(L, q, n, S, H, kc, vc, T, st, kv_row_stride);
    *po = L.part_o; *pl = L.part_l; *pm = L.fused ? L.pmax : nullptr; *pns = L.ns;
}

void attn_cross_combine(const float * part_o, const float * part_l, const float * part_m, int ns, int n, int S, int H, __half * out,
                        hipStream_t st, float * out32) {
    hipLaunchKernelGGL(k_xattn_combine, dim3(H, n), dim3(64), 0, st, part_o, part_l, part_m, ns, S, out, out32);
}

XattnPlan attn_cross_plan(int n, int H, int T, float * scratch) {
    const XLayout L = xattn_layout(n, H, T, scratch);
    XattnPlan P; P.ns = L.ns; P.ks = L.ks; P.pmax = L.pmax; P.part_l = L.part_l; P.part_o = L.part_o; P.fused = L.fused; P.head_major = xattn_head_major();
    return P;
}

void attn_cross_partials_layout(int n, int H, int T, float * scratch, const float ** po, const float ** pl, const float ** pm, int * pns) {
    const XLayout L = xattn_layout(n, H, T, scratch);
    *po = L.part_o; *pl = L.part_l; *pm = L.fused ? L.pmax : nullptr; *pns = L.ns;
}

void attn_cross_qsplit_partials(const float * x32, const float * ln_g, const float * ln_b, float eps, const __half * wq,
                                const float * bq, float qscale, int n, int S, int H, const __half * kc, const __half * vc, int T,
                                float * scratch, const float ** po, const float ** pl, const float ** pm, int * pns, hipStream_t st,
                                int64_t kv_row_stride) {
    const XLayout L = xattn_layout(n, H, T, scratch);
    *po = L.part_o; *pl = L.part_l; *pm = L.fused ? L.pmax : nullptr; *pns = L.ns;
    if (L.fused) {
        if (g_xattn_probe_skip & 1) return;
        auto kern = S <= 512 ? k_xattn_fused<1, true> : S <= 1024 ? k_xattn_fused<2, true> : k_xattn_fused<3, true>;      // S <= 1536
        hipLaunchKernelGGL(kern, xattn_grid(L.ns, H, n), dim3(256), 0, st, x32, ln_g, ln_b, eps, wq, bq, qscale, nullptr, S, kc, vc, T, L.ks, L.ns,
                           L.pmax, L.part_o, L.part_l, kv_row_stride, xattn_head_major() ? 1 : 0, stamp_next());
        return;
    }
    if (!(g_xattn_probe_skip & 1))
    {
        auto kern = S <= 512 ? k_xattn_qscores<1> : S <= 1024 ? k_xattn_qscores<2> : k_xattn_qscores<3>;      // S <= 1536
        hipLaunchKernelGGL(kern, dim3(L.ns, H, n), dim3(256), 0, st, x32, ln_g, ln_b, eps, wq, bq, qscale, S, kc, T, L.ks, L.ns,
                           L.sc, L.ld_sc, L.pmax, kv_row_stride);
    }
    const size_t smem = (((size_t) L.ks + 3) & ~(size_t) 3) * 4 + 4 * 64 * 4;
    if (!(g_xattn_probe_skip & 2))
    hipLaunchKernelGGL(k_xattn_pv, dim3(L.ns, H, n), dim3(256), smem, st, vc, S, T, L.ks, L.ns, L.sc, L.ld_sc, L.pmax, L.part_o, L.part_l, kv_row_stride);
}

bool xback_usable(int S, int H, int T, int rows) {
    if (S > 512 || (S % 64) != 0 || H * 64 != S) return false;
    const XLayout L = xattn_layout(1, H, T, nullptr);
    if (!L.fused || L.ns < 4 || L.ns > 8 || !xattn_head_major()) return false;
    static std::atomic<int> cache[64];
    int dev = 0; (void) hipGetDevice(&dev);
    int v = cache[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0, nb = 0;
        (void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *) k_xback, 256, 0) != hipSuccess) nb = 0;
        v = 1 + std::max(0, (cus - 1) * std::min(nb, 3));       // workgroups resident at once (they wait for each other), one CU left to others
        cache[dev & 63].store(v, std::memory_order_relaxed);
    }
    return H * L.ns * std::max(rows, 1) <= v - 1;
}
void xback(XbackArgs a, int H, float * scratch, hipStream_t st) {
    const int n = a.rows > 1 ? a.rows : 1;
    const XLayout L = xattn_layout(n, H, a.T, scratch);
    a.ks = L.ks; a.ns = L.ns; a.pmax = L.pmax; a.part_o = L.part_o; a.part_l = L.part_l;
    hipLaunchKernelGGL(k_xback, dim3(H, L.ns, n), dim3(256), 0, st, a, stamp_next());
}

size_t attn_cross_scratch_floats(int n, int H, int T) {
    const int ld_sc = (T + 63) & ~63;
    return (size_t) n * H * ((size_t) ld_sc + 2 * XS_MAX_SLICES + (size_t) XS_MAX_SLICES * 64);
}

static bool g_attn_one_group = false;
void set_attn_one_group(bool on) { if (on != g_attn_one_group) bump_mode_epoch(); g_attn_one_group = on; }

void attn_encoder(const __half * q, const __half * k, const __half * vt, int T, int Tpad, int S, int H, float scale,
                  __half * out, hipStream_t st, int B, float * out32, int qk_chunk_rows) {
    const int qk_rows = qk_chunk_rows > 0 ? qk_chunk_rows : T;
    // WMI_ATTN_FORM: 2 (default) = 32-row wavefronts, one sweep with a running maximum; 1 = the same kernel with the exact row
    // maximum found in a first sweep (the reference's soft-max argument); 0 = the round-1/2 kernel (16-row wavefronts, two sweeps)
    static const int form = getenv("WMI_ATTN_FORM") ? atoi(getenv("WMI_ATTN_FORM")) : 2;
    static const int ksplit = getenv("WMI_ATTN_KSPLIT") ? atoi(getenv("WMI_ATTN_KSPLIT")) : -1;      // A/B knob; default: by grid size
    if (form >= 1 && scale == 0.125f && (Tpad % 64) == 0) {
        const bool split = ksplit >= 0 ? ksplit > 1 : (((T + 127) / 128) * H * B < 512 && T >= 512 && !g_attn_one_group);
        attn_encoder2(q, k, vt, T, Tpad, S, H, out, st, B, out32, form == 2, split, qk_rows);
        return;
    }
    static const int nw = getenv("WMI_ATTN_NW") ? atoi(getenv("WMI_ATTN_NW")) : 4;      // wavefronts per workgroup (A/B knob)
    const int nblk = ((T + 63) / 64) * H * B;
    const bool ks2 = ksplit >= 0 ? ksplit == 2 : (nblk <= 512 && T >= 256 && !g_attn_one_group);
    if (nw == 4 && ks2) hipLaunchKernelGGL((k_attn_enc<4, 2>), dim3((T + 63) / 64, H, B), dim3(512), 0, st, q, k, vt, T, Tpad, S, scale, out, out32, qk_rows);
    else if (nw == 4)   hipLaunchKernelGGL((k_attn_enc<4, 1>), dim3((T + 63) / 64, H, B), dim3(256), 0, st, q, k, vt, T, Tpad, S, scale, out, out32, qk_rows);
    else                hipLaunchKernelGGL((k_attn_enc<2, 1>), dim3((T + 31) / 32, H, B), dim3(128), 0, st, q, k, vt, T, Tpad, S, scale, out, out32, qk_rows);
}

void attn_decoder(const __half * q, int n, int S, int H, const __half * kc, const __half * vc, int n_kv,
                  const float * mask, int ld_mask, __half * out, hipStream_t st, const int32_t * n_kv_dev, int n_kv_max, float * out32) {
    const int cap = n_kv_dev ? n_kv_max : n_kv;
    const size_t smem = (((size_t) cap + 3) & ~(size_t) 3) * 4 + 64 * 4 + 256 * 4;
    hipLaunchKernelGGL(k_attn_dec, dim3(n, H), dim3(256), smem, st, q, S, kc, vc, n_kv, mask, ld_mask, out, n_kv_dev, cap, out32);
}

}} // namespace wmi::k
