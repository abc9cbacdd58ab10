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

namespace wmi { namespace k {

namespace {

constexpr int NB = 64;          // workgroups of the statistics pass
constexpr int NT = 256;

struct MaxIdx { float v; int i; };
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {      // larger value, then smaller index (first occurrence)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ MaxIdx wave_max(MaxIdx m) {
    for (int o = 32; o > 0; o >>= 1) { MaxIdx t; t.v = __shfl_xor(m.v, o); t.i = __shfl_xor(m.i, o); m = better(m, t); }
    return m;
}
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// per-workgroup partial statistics of the filtered logits: maxima with first-index tie-break (all / text /
// timestamps) and sums of exp(l - local max) — combined exactly (online soft-max identity) by the second kernel
struct Partial { MaxIdx all, txt, ts; float sum, sum_ts; float pad[2]; };

__global__ __launch_bounds__(NT) void k_filter_stats(const float * __restrict__ logits, const uint8_t * __restrict__ ban,
                                                     const DecStep * __restrict__ stp, Partial * __restrict__ part) {
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
                ok[e] = true; lv[e] = lraw[e];
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
}

__global__ __launch_bounds__(64) void k_filter_pick(const Partial * __restrict__ part, const DecStep * __restrict__ stp,
                                                    SampleOut * __restrict__ out, SampleOut * __restrict__ out_host) {
    const int lane = threadIdx.x;
    part += (size_t) blockIdx.x * NB; stp += blockIdx.x; out += blockIdx.x; if (out_host) out_host += blockIdx.x;
    const Partial p = part[lane];                       // NB == 64: one partial per lane
    const MaxIdx a = wave_max(p.all), t = wave_max(p.txt), z = wave_max(p.ts);
    const float M = a.v;
    const float w = p.all.v > -INFINITY ? expf(p.all.v - M) : 0.0f;      // rescale the local sums to the global max
    const float sum = wave_sum(p.sum * w), sum_ts = wave_sum(p.sum_ts * w);
    if (lane == 0) {
        const int beg = stp->beg;
        const float lse = logf(sum) + M;
        // timestamp log-mass vs best text token (W/whisper.cpp:4659-4683)
        const float ts_logprob = sum_ts > 0.0f ? logf(sum_ts) + M - lse : -INFINITY;
        const float max_text = t.v > -INFINITY ? t.v - lse : -INFINITY;
        const bool force_ts = ts_logprob > max_text;
        const MaxIdx pick = force_ts ? z : a;
        SampleOut r;
        r.id = pick.i; r.plog = pick.v - lse; r.p = expf(r.plog); r.forced_ts = force_ts ? 1 : 0; r.seq = stp->seq;
        // timestamp statistics over the post-filter probabilities (W/whisper.cpp:4793-4809)
        const float p_ts_max = z.v > -INFINITY ? expf(z.v - lse) : 0.0f;
        const double sum_ts_p = (double) sum_ts * (double) expf(M - lse);
        r.tid = p_ts_max > 0.0f ? z.i : 0;
        r.pt = (float) ((double) p_ts_max / (sum_ts_p + 1e-10));
        r.ptsum = (float) sum_ts_p;
        if (r.id >= beg) { r.tid = r.id; r.pt = r.p; }
        *out = r;
        if (out_host) {
            // result straight into pinned host memory; the sequence number goes last, behind a system-scope fence: the host
            // spins on it instead of paying a stream synchronisation per token
            SampleOut body = r; body.seq = out_host->seq;
            *out_host = body;
            __threadfence_system();
            *(volatile int32_t *) &out_host->seq = r.seq;
            __threadfence_system();
        }
    }
}

} // namespace

void filter_argmax(const float * logits, const uint8_t * static_ban, const DecStep * step, SampleOut * out, void * scratch,
                   hipStream_t st, SampleOut * out_host, int n_rows) {
    Partial * part = (Partial *) scratch;
    hipLaunchKernelGGL(k_filter_stats, dim3(NB, n_rows), dim3(NT), 0, st, logits, static_ban, step, part);
    hipLaunchKernelGGL(k_filter_pick, dim3(n_rows), dim3(64), 0, st, part, step, out, out_host);
}
size_t filter_scratch_bytes(int n_rows) { return (size_t) n_rows * NB * sizeof(Partial); }

}} // namespace wmi::k
