// Device-side logit filters + greedy pick (reference whisper_process_logits
// W/whisper.cpp:4493-4775 and whisper_sample_token(best) :4777-4830, at temperature 0).
//
// Why on the device: a decode step otherwise ships 207 KB of logits over PCIe and spends ~0.3 ms of host time
// in three passes over 51 864 floats, several times what the GPU needs for the whole step.  Here the filters
// are a byte mask (static suppress list, uploaded once per parameter set) plus a few per-step scalars, and the
// log-soft-max statistics are block reductions; 32 bytes come back.
//
// Semantics kept from the reference: suppress rules in its order (the result is order-independent), "timestamps
// come in pairs", max_initial_ts, monotone timestamps, "if the timestamp mass beats every text token, force a
// timestamp", first-index tie-break of the arg-max loops, tid = 0 when every timestamp probability underflows.
// Difference: sums are tree reductions (the reference adds 51 864 terms sequentially in f32), i.e. p differs by
// ~1e-6 relative; the host implementation (host_logic.cpp) stays the bit-exact definition and is used for
// beam search, t > 0 and logit-filter callbacks.

#include "kernels.h"
#include <algorithm>
#include "wave_ops.h"

namespace wmi { namespace k {

namespace {

constexpr int NB = 64;          // workgroups of the statistics pass
constexpr int NT = 256;

using MaxIdx = FsMaxIdx;
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {      // larger value, then smaller index (first occurrence)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ MaxIdx wave_max(MaxIdx m) {
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) { MaxIdx t; t.v = WMI_SHX(m.v, o); t.i = WMI_SHX(m.i, o); m = better(m, t); }
    return m;
}
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += WMI_SHX(v, o); return v; }

// per-workgroup partial statistics of the filtered logits: maxima with first-index tie-break (all / text /
// timestamps) and sums of exp(l - local max) — combined exactly (online soft-max identity) by the second kernel
using Partial = FsPartial;

__global__ __launch_bounds__(NT) void k_filter_stats(const float * __restrict__ logits, const uint8_t * __restrict__ ban,
                                                     const DecStep * __restrict__ stp, Partial * __restrict__ part, const Stamp sp) {
    const unsigned long long ts0 = stamp_t0(sp.base);
    __shared__ MaxIdx s_all[4], s_txt[4], s_ts[4];
    __shared__ float s_sum[4], s_sum_ts[4];
    __shared__ float b_M;
    const DecStep st = stp[blockIdx.y];                  // grid.y: lock-step chunk (row of the logits matrix)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NV = st.n_vocab, beg = st.beg;
    logits += (size_t) blockIdx.y * NV; part += (size_t) blockIdx.y * NB;
    const bool ban_blank = st.flags & 1, last_ts = st.flags & 2, pen_ts = st.flags & 4;
    const int per = (NV + NB - 1) / NB, i0 = blockIdx.x * per, i1 = min(NV, i0 + per);

    constexpr int MAXE = 4;                            // ceil(51866 / 64 / 256)
    float lv[MAXE]; bool ok[MAXE];
    // the static mask and the logits of all MAXE elements first, from clamped indices (element by element this was
    // mask -> wait -> logit -> wait: eight dependent round trips per thread)
    unsigned char bn[MAXE]; float lraw[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) { const int i = i0 + e * NT + tid, ic = i < i1 ? i : i0; bn[e] = ban[ic]; lraw[e] = logits[ic]; }
    MaxIdx m_all = {-INFINITY, 0x7fffffff}, m_txt = m_all, m_ts = m_all;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int i = i0 + e * NT + tid;
        ok[e] = false; lv[e] = -INFINITY;
        if (i < i1) {
            bool a = !bn[e];
            if (ban_blank && (i == st.eot || i == st.space_id)) a = false;
            if (last_ts) { if (pen_ts) { if (i >= beg) a = false; } else { if (i < st.eot) a = false; } }
            if (i >= st.ts_initial_start) a = false;
            if (i >= beg && i < st.ts_floor_end) a = false;
            if (a) {
                ok[e] = true; lv[e] = st.temperature > 0.0f ? lraw[e] / st.temperature : lraw[e];
                const MaxIdx c = {lv[e], i};
                m_all = better(m_all, c);
                if (i < beg) m_txt = better(m_txt, c); else m_ts = better(m_ts, c);
            }
        }
    }
    m_all = wave_max(m_all); m_txt = wave_max(m_txt); m_ts = wave_max(m_ts);
    if (lane == 0) { s_all[wave] = m_all; s_txt[wave] = m_txt; s_ts[wave] = m_ts; }
    __syncthreads();
    if (tid == 0) {
        MaxIdx a = s_all[0], t = s_txt[0], z = s_ts[0];
        for (int w = 1; w < 4; ++w) { a = better(a, s_all[w]); t = better(t, s_txt[w]); z = better(z, s_ts[w]); }
        s_all[0] = a; s_txt[0] = t; s_ts[0] = z; b_M = a.v;
    }
    __syncthreads();
    const float M = b_M;
    float sum = 0.0f, sum_ts = 0.0f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        if (!ok[e]) continue;
        const float x = expf(lv[e] - M);
        sum += x;
        if (i0 + e * NT + tid >= beg) sum_ts += x;
    }
    sum = wave_sum(sum); sum_ts = wave_sum(sum_ts);
    if (lane == 0) { s_sum[wave] = sum; s_sum_ts[wave] = sum_ts; }
    __syncthreads();
    if (tid == 0) {
        Partial p;
        p.all = s_all[0]; p.txt = s_txt[0]; p.ts = s_ts[0];
        p.sum = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        p.sum_ts = (s_sum_ts[0] + s_sum_ts[1]) + (s_sum_ts[2] + s_sum_ts[3]);
        p.pad[0] = p.pad[1] = 0.0f;
        part[blockIdx.x] = p;
    }
    stamp_end(sp.base, sp.slot, ((int) blockIdx.y * (int) gridDim.x + (int) blockIdx.x) * 4 + wave, ts0);
}

// XU: chunks of 512 columns of the next step's activation row (S <= 512 XU); PPL: partials per lane (1: the 64 of k_filter_stats,
// 12: up to 768 from the vocabulary projection's fused epilogue — with PPL = 1 the arithmetic is exactly the one-partial form)
template <int XU, int PPL>
__global__ __launch_bounds__(64) void k_filter_pick(const Partial * __restrict__ part, int nparts, const DecStep * __restrict__ stp,
                                                    SampleOut * __restrict__ out, SampleOut * __restrict__ out_host, const ChainNext chain, const Stamp sp) {
    const unsigned long long ts0 = stamp_t0(sp.base);
    const int lane = threadIdx.x;
    part += (size_t) blockIdx.x * nparts; stp += blockIdx.x; out += blockIdx.x; if (out_host) out_host += blockIdx.x;
    DecStep * const step_rw = chain.step_rw ? chain.step_rw + blockIdx.x : nullptr;      // lock-step rows: a record and an activation row per chunk
    float * const chain_x = chain.x + (size_t) blockIdx.x * chain.S;
    // the step record and the partials are requested together (field by field at their first use, the record cost three more
    // dependent round trips on this one-wavefront kernel)
    const int4 st0 = *(const int4 *) stp;                  // token, pos, n_kv, kv_head
    const int beg = stp->beg;
    // status of the step's in-launch hand-offs (k_mlp_pair) rides in the tags the host polls: an unconditional load (of a zero pad word when
    // there is no such word) requested with the record
    const uint32_t * const fault_p = chain.fault ? chain.fault : (const uint32_t *) &stp->pad[0];
    const uint32_t fault = *(const volatile uint32_t *) fault_p;
    const int seqv = (stp->seq & SAMPLE_SEQ_MASK) | ((fault & (PAIR_FAULT_TIMEOUT | PAIR_FAULT_PARITY)) ? SAMPLE_TAG_FAULT : 0) | ((fault & PAIR_SLOW) ? SAMPLE_TAG_SLOW : 0);
    Partial pq[PPL];                                     // partials lane, lane + 64, ... (all requested before the first use)
#pragma unroll
    for (int q = 0; q < PPL; ++q) { const int idx = lane + 64 * q; pq[q] = part[idx < nparts ? idx : 0]; }
    MaxIdx la = {-INFINITY, 0x7fffffff}, lt = la, lz = la;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        if (lane + 64 * q >= nparts) { pq[q].all = MaxIdx{-INFINITY, 0x7fffffff}; pq[q].txt = pq[q].all; pq[q].ts = pq[q].all; pq[q].sum = 0.0f; pq[q].sum_ts = 0.0f; }
        la = better(la, pq[q].all); lt = better(lt, pq[q].txt); lz = better(lz, pq[q].ts);
    }
    const MaxIdx a = wave_max(la), t = wave_max(lt), z = wave_max(lz);
    const float M = a.v;
    float lsum = 0.0f, lsum_ts = 0.0f;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const float w = pq[q].all.v > -INFINITY ? expf(pq[q].all.v - M) : 0.0f;      // rescale the local sums to the global max
        if (q == 0) { lsum = pq[q].sum * w; lsum_ts = pq[q].sum_ts * w; } else { lsum += pq[q].sum * w; lsum_ts += pq[q].sum_ts * w; }
    }
    const float sum = wave_sum(lsum), sum_ts = wave_sum(lsum_ts);
    // every lane evaluates the pick (uniform values): no broadcast between the decision and the next step's gathers
    const float lse = logf(sum) + M;
    // timestamp log-mass vs best text token (W/whisper.cpp:4659-4683)
    const float ts_logprob = sum_ts > 0.0f ? logf(sum_ts) + M - lse : -INFINITY;
    const float max_text = t.v > -INFINITY ? t.v - lse : -INFINITY;
    const bool force_ts = ts_logprob > max_text;
    const MaxIdx pick = force_ts ? z : a;
    const int id = pick.i;
    // the next greedy step feeds this pick at the next position (the host checks that before it replays the chained step):
    // its embedding + positional rows are requested now, the result goes to the host while they are in flight
    const int pos1 = st0.y + 1;
    const bool chain_on = step_rw && pos1 < chain.n_pos;
    // lane L owns columns 512 u + 8 L .. + 8 of the row (XU chunks of 512 columns): one 16-byte and two 16-byte loads per chunk
    uint4 trv[XU]; float4 prv[XU][2];
    if (chain_on) {
        const __half * tr = chain.te + (size_t) id * chain.S;
        const float * pr = chain.pe + (size_t) pos1 * chain.S;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int c = lane * 8 + 512 * u, cc = c < chain.S ? c : 0;
            trv[u] = *(const uint4 *) (tr + cc); prv[u][0] = *(const float4 *) (pr + cc); prv[u][1] = *(const float4 *) (pr + cc + 4);
        }
    }
    if (lane == 0) {
        SampleOut r;
        r.id = id; r.plog = pick.v - lse; r.p = expf(r.plog); r.seq0 = seqv; r.seq = seqv;
        // timestamp statistics over the post-filter probabilities (W/whisper.cpp:4793-4809)
        const float p_ts_max = z.v > -INFINITY ? expf(z.v - lse) : 0.0f;
        const double sum_ts_p = (double) sum_ts * (double) expf(M - lse);
        r.tid = p_ts_max > 0.0f ? z.i : 0;
        r.pt = (float) ((double) p_ts_max / (sum_ts_p + 1e-10));
        r.ptsum = (float) sum_ts_p;
        if (r.id >= beg) { r.tid = r.id; r.pt = r.p; }
        *out = r;
        if (out_host) {                                      // two 16-byte stores, each carrying the step's sequence number
            const int4 * h = (const int4 *) &r;
            ((int4 *) out_host)[0] = h[0];
            ((int4 *) out_host)[1] = h[1];
        }
    }
    if (step_rw) {
        if (chain_on) {
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                const int c = lane * 8 + 512 * u;
                if (c < chain.S) {                           // k_dec_embed_step's arithmetic: f32(te) + pe
                    const __half2 * h = (const __half2 *) &trv[u];
                    const float2 f0 = __half22float2(h[0]), f1 = __half22float2(h[1]), f2 = __half22float2(h[2]), f3 = __half22float2(h[3]);
                    *(float4 *) (chain_x + c)     = make_float4(f0.x + prv[u][0].x, f0.y + prv[u][0].y, f1.x + prv[u][0].z, f1.y + prv[u][0].w);
                    *(float4 *) (chain_x + c + 4) = make_float4(f2.x + prv[u][1].x, f2.y + prv[u][1].y, f3.x + prv[u][1].z, f3.y + prv[u][1].w);
                }
            }
        }
        if (lane == 0) *(int4 *) step_rw = make_int4(id, pos1, st0.z + 1, st0.w + 1);      // token, pos, n_kv, kv_head
    }
    stamp_end(sp.base, sp.slot, blockIdx.x, ts0);
}

// ---- draws (beam search / t > 0).  Same filter predicate as k_filter_stats.
__device__ __forceinline__ bool allowed(const DecStep & st, unsigned char bn, int i) {
    const bool ban_blank = st.flags & 1, last_ts = st.flags & 2, pen_ts = st.flags & 4;
    bool a = !bn;
    if (ban_blank && (i == st.eot || i == st.space_id)) a = false;
    if (last_ts) { if (pen_ts) { if (i >= st.beg) a = false; } else { if (i < st.eot) a = false; } }
    if (i >= st.ts_initial_start) a = false;
    if (i >= st.beg && i < st.ts_floor_end) a = false;
    return a;
}
struct RowStats { float M, lse; int force_ts; MaxIdx ts; float sum_ts; };
// the 64 partials of a row -> global max, log-sum-exp, "timestamp mass beats every text token" (all lanes get the result)
__device__ __forceinline__ RowStats row_stats(const Partial * part, int lane) {
    const Partial p = part[lane];
    const MaxIdx a = wave_max(p.all), t = wave_max(p.txt), z = wave_max(p.ts);
    const float M = a.v;
    const float w = p.all.v > -INFINITY ? expf(p.all.v - M) : 0.0f;
    const float sum = wave_sum(p.sum * w), sum_ts = wave_sum(p.sum_ts * w);
    RowStats r;
    r.M = M; r.lse = logf(sum) + M;
    const float ts_logprob = sum_ts > 0.0f ? logf(sum_ts) + M - r.lse : -INFINITY;
    const float max_text = t.v > -INFINITY ? t.v - r.lse : -INFINITY;
    r.force_ts = ts_logprob > max_text ? 1 : 0;
    r.ts = z; r.sum_ts = sum_ts;
    return r;
}
__device__ __forceinline__ float prob_of(const DecStep & st, const RowStats & rs, float lraw, unsigned char bn, int i) {
    if (!allowed(st, bn, i) || (rs.force_ts && i < st.beg)) return 0.0f;
    const float l = st.temperature > 0.0f ? lraw / st.temperature : lraw;
    return expf(l - rs.lse);
}
// block sums of the probabilities in double: grid (NB, rows); block b covers the same index range as in k_filter_stats
__global__ __launch_bounds__(NT) void k_prob_blocks(const float * __restrict__ logits, const uint8_t * __restrict__ ban,
                                                    const DecStep * __restrict__ stp, const Partial * __restrict__ part,
                                                    double * __restrict__ bsum) {
    __shared__ double s_w[4];
    const DecStep st = stp[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NV = st.n_vocab;
    logits += (size_t) blockIdx.y * NV;
    const RowStats rs = row_stats(part + (size_t) blockIdx.y * NB, lane);
    const int per = (NV + NB - 1) / NB, i0 = blockIdx.x * per, i1 = min(NV, i0 + per);
    double acc = 0.0;
    for (int i = i0 + tid; i < i1; i += NT) acc += (double) prob_of(st, rs, logits[i], ban[i], i);
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) acc += WMI_SHX(acc, o);
    if (lane == 0) s_w[wave] = acc;
    __syncthreads();
    if (tid == 0) bsum[(size_t) blockIdx.y * NB + blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
// one wavefront per (draw, row): block by prefix of the 64 block sums, then the element inside the block
__global__ __launch_bounds__(64) void k_draw(const float * __restrict__ logits, const uint8_t * __restrict__ ban,
                                             const DecStep * __restrict__ stp, const Partial * __restrict__ part,
                                             const double * __restrict__ bsum, const double * __restrict__ u, int k,
                                             int tid_default, SampleOut * __restrict__ out) {
    const int lane = threadIdx.x, dr = blockIdx.x, row = blockIdx.y;
    const DecStep st = stp[row];
    const int NV = st.n_vocab;
    logits += (size_t) row * NV;
    const RowStats rs = row_stats(part + (size_t) row * NB, lane);
    // inclusive prefix of the block sums over the lanes
    const double mine = bsum[(size_t) row * NB + lane];
    double pre = mine;
    for (int o = 1; o < 64; o <<= 1) { const double t = __shfl_up(pre, o); if (lane >= o) pre += t; }
    const double total = __shfl(pre, 63);
    const double target = u[(size_t) row * k + dr] * total;
    // first block whose inclusive prefix reaches the target (the last non-empty one if rounding left the target above the total)
    const unsigned long long hit = __ballot(pre >= target && mine > 0.0);
    const unsigned long long nz = __ballot(mine > 0.0);
    int b = hit ? __ffsll((long long) hit) - 1 : (nz ? 63 - __clzll((long long) nz) : 0);
    const double before = __shfl(pre, b) - __shfl(mine, b);
    const int per = (NV + NB - 1) / NB, i0 = b * per, i1 = min(NV, i0 + per);
    // inside the block: lane L owns a contiguous run of elements
    const int run = (per + 63) / 64, j0 = i0 + lane * run, j1 = min(i1, j0 + run);
    double loc = 0.0;
    for (int i = j0; i < j1; ++i) loc += (double) prob_of(st, rs, logits[i], ban[i], i);
    double lpre = loc;
    for (int o = 1; o < 64; o <<= 1) { const double t = __shfl_up(lpre, o); if (lane >= o) lpre += t; }
    const double want = target - before;
    const unsigned long long lh = __ballot(lpre >= want && loc > 0.0), lnz = __ballot(loc > 0.0);
    const int L = lh ? __ffsll((long long) lh) - 1 : (lnz ? 63 - __clzll((long long) lnz) : 0);
    const double lbefore = __shfl(lpre, L) - __shfl(loc, L);
    if (lane == L) {
        int pick = -1, last_nz = -1; double c = lbefore;
        for (int i = j0; i < j1; ++i) {
            const float p = prob_of(st, rs, logits[i], ban[i], i);
            if (p > 0.0f) { last_nz = i; c += (double) p; if (pick < 0 && c >= want) pick = i; }
        }
        if (pick < 0) pick = last_nz >= 0 ? last_nz : 0;
        SampleOut r;
        const float lraw = logits[pick];
        const float l = st.temperature > 0.0f ? lraw / st.temperature : lraw;
        r.id = pick; r.plog = l - rs.lse; r.p = expf(r.plog); r.seq0 = dr; r.seq = dr;
        const float p_ts_max = rs.ts.v > -INFINITY ? expf(rs.ts.v - rs.lse) : 0.0f;
        const double sum_ts_p = (double) rs.sum_ts * (double) expf(rs.M - rs.lse);
        r.tid = p_ts_max > 0.0f ? rs.ts.i : tid_default;      // the reference's initial value: token_beg (top-k) or 0 (single draw)
        r.pt = (float) ((double) p_ts_max / (sum_ts_p + 1e-10));
        r.ptsum = (float) sum_ts_p;
        if (r.id >= st.beg) { r.tid = r.id; r.pt = r.p; }
        out[(size_t) row * k + dr] = r;
    }
}

} // namespace

void filter_draw(const float * logits, const uint8_t * static_ban, const DecStep * step, const double * u, int k, SampleOut * out,
                 void * scratch, hipStream_t st, int n_rows, int tid_default) {
    Partial * part = (Partial *) scratch;
    double * bsum = (double *) ((char *) scratch + filter_scratch_bytes(n_rows));
    hipLaunchKernelGGL(k_filter_stats, dim3(NB, n_rows), dim3(NT), 0, st, logits, static_ban, step, part, Stamp{nullptr, 0});
    hipLaunchKernelGGL(k_prob_blocks, dim3(NB, n_rows), dim3(NT), 0, st, logits, static_ban, step, part, bsum);
    hipLaunchKernelGGL(k_draw, dim3(k, n_rows), dim3(64), 0, st, logits, static_ban, step, part, bsum, u, k, tid_default, out);
}
size_t filter_draw_scratch_bytes(int n_rows) { return filter_scratch_bytes(n_rows) + (size_t) n_rows * NB * sizeof(double); }

void filter_argmax(const float * logits, const uint8_t * static_ban, const DecStep * step, SampleOut * out, void * scratch,
                   hipStream_t st, SampleOut * out_host, int n_rows, const ChainNext * chain, int fused_parts) {
    Partial * part = (Partial *) scratch;
    const bool fused = fused_parts > 0 && fused_parts <= FS_MAX_PARTS && n_rows == 1;
    if (!fused) hipLaunchKernelGGL(k_filter_stats, dim3(NB, n_rows), dim3(NT), 0, st, logits, static_ban, step, part, stamp_next());
    ChainNext cn{};                                         // one row: the greedy step of device.cpp; rows: the lock-step step of batch.cpp
    if (chain) cn = *chain;
    if (cn.step_rw && cn.S > 1536) { const uint32_t * f = cn.fault; cn = ChainNext{}; cn.fault = f; }          // (no such model: the row registers cover 3 chunks; the host then embeds)
    const int xu = cn.step_rw ? (cn.S + 511) / 512 : 1;
    const int np = fused ? fused_parts : NB;
    constexpr int PF = FS_MAX_PARTS / 64;
    if (fused) {
        if (xu <= 1)      hipLaunchKernelGGL((k_filter_pick<1, PF>), dim3(1), dim3(64), 0, st, part, np, step, out, out_host, cn, stamp_next());
        else if (xu == 2) hipLaunchKernelGGL((k_filter_pick<2, PF>), dim3(1), dim3(64), 0, st, part, np, step, out, out_host, cn, stamp_next());
        else              hipLaunchKernelGGL((k_filter_pick<3, PF>), dim3(1), dim3(64), 0, st, part, np, step, out, out_host, cn, stamp_next());
        return;
    }
    if (xu <= 1)      hipLaunchKernelGGL((k_filter_pick<1, 1>), dim3(n_rows), dim3(64), 0, st, part, np, step, out, out_host, cn, stamp_next());
    else if (xu == 2) hipLaunchKernelGGL((k_filter_pick<2, 1>), dim3(n_rows), dim3(64), 0, st, part, np, step, out, out_host, cn, stamp_next());
    else              hipLaunchKernelGGL((k_filter_pick<3, 1>), dim3(n_rows), dim3(64), 0, st, part, np, step, out, out_host, cn, stamp_next());
}
size_t filter_scratch_bytes(int n_rows) { return (size_t) std::max(n_rows * NB, FS_MAX_PARTS) * sizeof(Partial); }

}} // namespace wmi::k
