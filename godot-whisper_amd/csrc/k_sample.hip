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

constexpr int NT = 1024;

struct MaxIdx { float v; int i; };
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {      // larger value, then smaller index (first occurrence)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ MaxIdx wave_max(MaxIdx m) {
    for (int o = 32; o > 0; o >>= 1) { MaxIdx t; t.v = __shfl_xor(m.v, o); t.i = __shfl_xor(m.i, o); m = better(m, t); }
    return m;
}
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

__global__ __launch_bounds__(NT) void k_filter_argmax(const float * __restrict__ logits, const uint8_t * __restrict__ ban,
                                                      const DecStep * __restrict__ stp, SampleOut * __restrict__ out) {
    __shared__ MaxIdx s_all[16], s_txt[16], s_ts[16];
    __shared__ float s_sum[16], s_sum_ts[16];
    __shared__ float b_M, b_lse, b_sumts;
    __shared__ MaxIdx b_all, b_txt, b_ts;
    const DecStep st = *stp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NV = st.n_vocab, beg = st.beg;
    const bool ban_blank = st.flags & 1, last_ts = st.flags & 2, pen_ts = st.flags & 4;

    auto allowed = [&](int i) -> bool {
        if (ban[i]) return false;
        if (ban_blank && (i == st.eot || i == st.space_id)) return false;
        if (last_ts) { if (pen_ts) { if (i >= beg) return false; } else { if (i < st.eot) return false; } }
        if (i >= st.ts_initial_start) return false;
        if (i >= beg && i < st.ts_floor_end) return false;
        return true;
    };

    // pass 1: maxima (all / text / timestamps) with first-index tie-break
    MaxIdx m_all = {-INFINITY, 0x7fffffff}, m_txt = m_all, m_ts = m_all;
    for (int i = tid; i < NV; i += NT) {
        if (!allowed(i)) continue;
        const MaxIdx c = {logits[i], i};
        m_all = better(m_all, c);
        if (i < beg) m_txt = better(m_txt, c); else m_ts = better(m_ts, c);
    }
    m_all = wave_max(m_all); m_txt = wave_max(m_txt); m_ts = wave_max(m_ts);
    if (lane == 0) { s_all[wave] = m_all; s_txt[wave] = m_txt; s_ts[wave] = m_ts; }
    __syncthreads();
    if (tid == 0) {
        MaxIdx a = s_all[0], t = s_txt[0], z = s_ts[0];
        for (int w = 1; w < NT / 64; ++w) { a = better(a, s_all[w]); t = better(t, s_txt[w]); z = better(z, s_ts[w]); }
        b_all = a; b_txt = t; b_ts = z; b_M = a.v;
    }
    __syncthreads();
    const float M = b_M;

    // pass 2: sum exp(l - M) over all allowed and over the timestamp slice
    float sum = 0.0f, sum_ts = 0.0f;
    for (int i = tid; i < NV; i += NT) {
        if (!allowed(i)) continue;
        const float e = expf(logits[i] - M);
        sum += e;
        if (i >= beg) sum_ts += e;
    }
    sum = wave_sum(sum); sum_ts = wave_sum(sum_ts);
    if (lane == 0) { s_sum[wave] = sum; s_sum_ts[wave] = sum_ts; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.0f, t = 0.0f;
        for (int w = 0; w < NT / 64; ++w) { a += s_sum[w]; t += s_sum_ts[w]; }
        const float lse = logf(a) + M;
        // timestamp log-mass vs best text token (W/whisper.cpp:4659-4683)
        const float ts_logprob = t > 0.0f ? logf(t) + M - lse : -INFINITY;
        const float max_text = b_txt.v > -INFINITY ? b_txt.v - lse : -INFINITY;
        const bool force_ts = ts_logprob > max_text;
        const MaxIdx pick = force_ts ? b_ts : b_all;
        SampleOut r;
        r.id = pick.i; r.plog = pick.v - lse; r.p = expf(r.plog); r.forced_ts = force_ts ? 1 : 0; r.pad = 0;
        // timestamp statistics over the post-filter probabilities (W/whisper.cpp:4793-4809)
        const float p_ts_max = b_ts.v > -INFINITY ? expf(b_ts.v - lse) : 0.0f;
        const double sum_ts_p = (double) t * (double) expf(M - lse);
        r.tid = p_ts_max > 0.0f ? b_ts.i : 0;
        r.pt = (float) ((double) p_ts_max / (sum_ts_p + 1e-10));
        r.ptsum = (float) sum_ts_p;
        if (r.id >= beg) { r.tid = r.id; r.pt = r.p; }
        *out = r;
    }
}

} // namespace

void filter_argmax(const float * logits, const uint8_t * static_ban, const DecStep * step, SampleOut * out, hipStream_t st) {
    hipLaunchKernelGGL(k_filter_argmax, dim3(1), dim3(NT), 0, st, logits, static_ban, step, out);
}

}} // namespace wmi::k
