// Device-side orchestration of the hot path: state arenas, PCM -> mel, encoder, decoder step
// (reference seam: log_mel_spectrogram W/whisper.cpp:2793,
// whisper_encode_internal :2086, whisper_decode_internal :2517).
//
// HBM layout of one context (base.en figures):
//   weights arena  148 MB   one allocation, f16 matrices K-contiguous, f32 vectors
//   self KV        16.5 MB  k,v [L][3*n_text_ctx][S] f16       (same capacity as the reference)
//   cross KV       18.4 MB  k,v [L][T][S] f16 (V is NOT transposed here; see attn_decoder)
//   encoder work   ~16 MB   token-major activations, f32 residual + f16 GEMM operands
// Every buffer is allocated once in init_state and reused; nothing is allocated on the hot path
// except when a longer PCM / mel than ever seen before arrives (grow-only).

#include "wmi.h"
#include <atomic>
#include <thread>
#include <map>
#include <mutex>
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace wmi {

namespace {

template <typename T> bool dalloc(T *& p, size_t n_elems) {
    p = nullptr;
    return HIP_OK(hipMalloc((void **) &p, std::max<size_t>(n_elems, 1) * sizeof(T)));
}
template <typename T> void dfree(T *& p) { if (p) (void) hipFree(p); p = nullptr; }

} // namespace

static bool make_own_queue_stream(int device, hipStream_t * out) {
    uint32_t mask[16]; for (uint32_t & m : mask) m = 0xFFFFFFFFu;
    hipDeviceProp_t prop{};
    const int ncu = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
    const uint32_t words = (uint32_t) std::min(16, (ncu + 31) / 32);
    if (hipExtStreamCreateWithCUMask(out, words, mask) == hipSuccess) return true;
    (void) hipGetLastError();
    return false;
}

// Process-wide pool of such streams, per device.  Measured (profiles/r04d_replica_streams_hw_queues.txt): whether four contexts' launch
// chains run beside each other depends on WHEN their hardware queues were created relative to the process's other queues — the queues a
// process makes first behave, queues made after several contexts have come and gone did not (bench.py: 40 ms per chunk against 18).  So
// the streams are made once, as early as a context exists, never destroyed, and handed from context to context.
// The pool is capped (WMI_OWN_QUEUE_CAP, default 16 per device = four contexts with their three spare queues each): a process that keeps
// more contexts than that alive gets ordinary non-blocking streams for the rest — hardware queues are a finite resource the embedding
// application shares.  These streams are created with the default flags: like every blocking stream they synchronise with the
// legacy NULL stream; an embedder that works on the NULL stream (torch's default stream is one) serialises with a context's launches —
// use a non-blocking stream on the application's side, or WMI_POOLED_MAIN_STREAM=1 for a non-blocking context stream.
static std::mutex g_oq_mu;
static std::map<int, std::vector<hipStream_t>> g_oq_pool;
static std::map<int, int> g_oq_made;
hipStream_t own_queue_stream_get(int device) {
    static const int cap = getenv("WMI_OWN_QUEUE_CAP") ? std::max(0, atoi(getenv("WMI_OWN_QUEUE_CAP"))) : 16;
    {
        std::lock_guard<std::mutex> lk(g_oq_mu);
        auto & v = g_oq_pool[device];
        if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
        if (g_oq_made[device] >= cap) return nullptr;
        g_oq_made[device] += 1;
    }
    hipStream_t s = nullptr;
    if (make_own_queue_stream(device, &s)) return s;
    std::lock_guard<std::mutex> lk(g_oq_mu);
    g_oq_made[device] -= 1;
    return nullptr;
}
void own_queue_stream_put(int device, hipStream_t s) {
    if (!s) return;
    (void) hipStreamSynchronize(s);
    std::lock_guard<std::mutex> lk(g_oq_mu);
    g_oq_pool[device].push_back(s);
}

// ---- which state ctx.state is, per thread (wmi.h: StateSlot)
static thread_local StateInstall * t_state_install = nullptr;
State * StateSlot::get() const {
    for (const StateInstall * p = t_state_install; p; p = p->prev) if (p->slot == this) return p->st;
    return own;
}
StateInstall::StateInstall(const StateSlot & sl, State * s) : slot(&sl), st(s), prev(t_state_install) { t_state_install = this; }
StateInstall::~StateInstall() { t_state_install = prev; }
StateInstall * state_installs_top() { return t_state_install; }
void state_installs_set(StateInstall * top) { t_state_install = top; }

static bool init_state_into(whisper_context & ctx, State * st, bool replica_state, hipStream_t adopt);
bool init_state(whisper_context & ctx, bool replica_state, hipStream_t adopt) {
    State * st = new State();
    ctx.state = st;                                          // the context's own state (free_state releases it, also after a failure here)
    return init_state_into(ctx, st, replica_state, adopt);
}
static bool init_state_into(whisper_context & ctx, State * st, bool replica_state, hipStream_t adopt) {
    const HParams & hp = ctx.model.hp;
    st->device = ctx.device;
    DeviceState & d = st->dev;
    if (!HIP_OK(hipSetDevice(ctx.device))) return false;
    // The context's stream has a hardware queue of its own (a CU-masked stream naming every CU) taken from the process-wide pool above;
    // ordinary streams are multiplexed by the runtime onto a few shared hardware queues in creation order.  What this buys is measured,
    // not understood in full (profiles/r04d_replica_streams_hw_queues.txt): four contexts running beam search side by side (wmi_full_batch
    // replicas) take 18 ms per large-v3 chunk when all four launch on queues the process made early, and 40 ms — hardly better than one after
    // the other — when the calling context's stream was an ordinary one created after an earlier context had run lock-step batches (sixteen
    // more ordinary streams).  One chunk alone measures the same on either kind of stream.  WMI_POOLED_MAIN_STREAM=1: the ordinary kind.
    static const bool pooled = getenv("WMI_POOLED_MAIN_STREAM") != nullptr || getenv("WMI_REPLICA_POOLED_QUEUES") != nullptr;
    if (adopt) { d.stream = adopt; d.stream_own_queue = true; }
    else if (!pooled && (d.stream = own_queue_stream_get(ctx.device))) d.stream_own_queue = true;
    else HIP_TRY(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    if (!replica_state) {
        // as many as wmi_full_batch's default number of replica contexts (WMI_BATCH_REPLICAS, 3); measured on large-v3 q5_1, beam 5 x 8 chunks:
        // queues made here -> 18 ms per chunk on 1 + 3 contexts whatever the process did in between; made when the replicas are first
        // needed -> 18 or 40 ms depending on what else had been created by then (profiles/r04d_replica_streams_hw_queues.txt)
        static const int spares = getenv("WMI_SPARE_QUEUES") ? atoi(getenv("WMI_SPARE_QUEUES"))
                                : getenv("WMI_BATCH_REPLICAS") ? std::max(0, std::min(15, atoi(getenv("WMI_BATCH_REPLICAS")))) : 3;
        for (int i = 0; i < spares; ++i) { hipStream_t sp = own_queue_stream_get(ctx.device); if (sp) ctx.spare_streams.push_back(sp); }
    }

    const size_t S = hp.n_audio_state, T = hp.n_audio_ctx, Lt = hp.n_text_layer, H = hp.n_audio_head;
    const size_t n_self = 3 * (size_t) hp.n_text_ctx;                         // W/whisper.cpp:3009 (factor 3)
    st->kv_self.size = (uint32_t) n_self; st->kv_self.cells.assign(n_self, KVCell());
    bool ok = true;
    ok = ok && dalloc(st->kv_self.k, Lt * n_self * S) && dalloc(st->kv_self.v, Lt * n_self * S);
    ok = ok && dalloc(d.kvc_k, Lt * T * S) && dalloc(d.kvc_v, Lt * T * S);
    d.Tpad = (int) ((T + 63) / 64 * 64);
    const size_t mel_rows = 2 * T + 8;
    ok = ok && dalloc(d.mel_t, mel_rows * hp.n_mels + 1024) && dalloc(d.conv1, (2 * T + 4) * S)
            && dalloc(d.x, T * S) && dalloc(d.embd_conv, T * S) && dalloc(d.xn, T * S)
            && dalloc(d.q, T * S) && dalloc(d.k, T * S) && dalloc(d.vt, S * d.Tpad) && dalloc(d.att, T * S)
            && dalloc(d.h, T * 4 * S) && dalloc(d.rowmax, H * T) && dalloc(d.enc_out, T * S) && dalloc(d.enc_out_h, T * S)
            && dalloc(d.mel_max, 4);
    const size_t n = hp.n_text_ctx;
    if (ctx.model.quantised) {                              // q8 activation rows + f32 attention outputs (device_q.cpp)
        const size_t rows = std::max<size_t>(T, n);
        d.aq_rows = (int) rows;
        ok = ok && dalloc(d.aq, rows * 4 * S) && dalloc(d.ads, 2 * rows * (4 * S / 32)) && dalloc(d.att32, T * S) && dalloc(d.datt32, n * S);
        d.wq16_elems = 8 * S * S;                           // mlp (4 S^2), q|k|v (3 S^2), four layers of cross K | V at a time
        ok = ok && dalloc(d.aq16, rows * 4 * S) && dalloc(d.wq16, d.wq16_elems);
    }
    d.logits_rows_cap = 8;
    // tokens | positions | rows wanting logits | mask: ONE allocation in the order of the pinned staging block, one copy per decode() call
    // (four separate copies cost ~20 us of every batched step)
    ok = ok && dalloc(d.d_tokens, 3 * n + n * n_self);
    if (ok) { d.d_pos = d.d_tokens + n; d.d_rows = d.d_pos + n; d.d_mask = (float *) (d.d_rows + n); }
    ok = ok
            && dalloc(d.dx, n * S) && dalloc(d.dxn, n * S) && dalloc(d.dq, n * S) && dalloc(d.datt, n * S)
            && dalloc(d.dh, n * 4 * S) && dalloc(d.logits, (size_t) d.logits_rows_cap * hp.n_vocab)
            && dalloc(d.xattn, k::attn_cross_scratch_floats((int) n, (int) H, (int) T));
    d.pinned_bytes = std::max<size_t>((size_t) d.logits_rows_cap * hp.n_vocab * 4, n * n_self * 4 + 3 * n * 4 + 4096);
    ok = ok && HIP_OK(hipHostMalloc(&d.pinned, d.pinned_bytes, hipHostMallocDefault));
    // the in-launch hand-off of the one-row step's MLP (k::mlp_pair): 2 S granules + the launch counter, zeroed once (tag 0 = never valid)
    {
        unsigned char * hand = nullptr;
        // (+ 2 S granules behind the words for the one-launch front of a layer, k::front: 3 S / 2 of q|k|v, S / 2 of the attention row)
        // (+ 40 KB behind those for the one-launch back of the cross-attention, k::xback: 8 heads x 8 slices x 66 granules of partials, S / 2 of the row)
        ok = ok && dalloc(hand, (size_t) 32 * S + 64 + 40960);
        if (ok) { d.mlp_hand = hand; d.mlp_arrive = (unsigned long long *) (hand + (size_t) 16 * S); k::fill_zero(hand, (size_t) 32 * S + 64 + 40960, d.stream); }
    }
    if (!ok) { WMI_ERR("%s: device allocation failed\n", __func__); return false; }
    // buffers that are read before being fully written must hold finite values
    k::fill_zero(d.vt, S * d.Tpad * sizeof(__half), d.stream);
    k::fill_zero(d.conv1, (2 * T + 4) * S * sizeof(__half), d.stream);
    k::fill_zero(d.mel_t, (mel_rows * hp.n_mels + 1024) * sizeof(__half), d.stream);
    k::fill_zero(st->kv_self.k, Lt * n_self * S * sizeof(__half), d.stream);
    k::fill_zero(st->kv_self.v, Lt * n_self * S * sizeof(__half), d.stream);
    k::fill_zero(d.kvc_k, Lt * T * S * sizeof(__half), d.stream);
    k::fill_zero(d.kvc_v, Lt * T * S * sizeof(__half), d.stream);
    HIP_TRY(hipStreamSynchronize(d.stream));

    st->batch.token.resize(n); st->batch.pos.resize(n); st->batch.seq_id.resize(n); st->batch.logits.resize(n);
    st->logits.reserve((size_t) hp.n_vocab * 8);
    for (auto & dec : st->decoders) dec.rng = std::mt19937(0);
    WMI_INFO("%s: kv self size = %.2f MB, kv cross size = %.2f MB\n", __func__,
             2.0 * Lt * n_self * S * 2 / 1e6, 2.0 * Lt * T * S * 2 / 1e6);
    return true;
}

State * create_state(whisper_context & ctx) {
    // a further state for the same weights (whisper_init_state).  The context's own state is not touched: another thread may be computing on it.
    State * st = new State();
    if (init_state_into(ctx, st, true, nullptr)) return st;                // (the context's spare streams exist already)
    destroy_state(st);
    return nullptr;
}

void free_state(whisper_context & ctx) {
    destroy_state(ctx.state);
    ctx.state = nullptr;
}

void destroy_state(State * st) {
    if (!st) return;
    (void) hipSetDevice(st->device);
    DeviceState & d = st->dev;
    if (d.stream) (void) hipStreamSynchronize(d.stream);
    dfree(st->kv_self.k); dfree(st->kv_self.v); dfree(d.kvc_k); dfree(d.kvc_v);
    if (d.copy_stream) { (void) hipStreamSynchronize(d.copy_stream); (void) hipStreamDestroy(d.copy_stream); }
    if (d.energy_ev) (void) hipEventDestroy(d.energy_ev);
    for (hipEvent_t & e : d.ph_ev) { if (e) (void) hipEventDestroy(e); e = nullptr; }
    if (d.mel_ev) { (void) hipEventDestroy(d.mel_ev); d.mel_ev = nullptr; }       // (the state of a lock-step call's primary: pcm_to_mel_batch)
    dfree(d.energy); if (d.energy_host) (void) hipHostFree(d.energy_host);
    if (d.ts_host) (void) hipHostFree(d.ts_host);
    dfree(d.pcm); dfree(d.mel); dfree(d.mel_max); dfree(d.mel_t); dfree(d.conv1); dfree(d.x); dfree(d.embd_conv);
    dfree(d.xn); dfree(d.q); dfree(d.k); dfree(d.vt); dfree(d.att); dfree(d.h); dfree(d.rowmax); dfree(d.enc_out);
    dfree(d.enc_out_h); dfree(d.d_tokens); d.d_pos = nullptr; d.d_mask = nullptr; d.d_rows = nullptr; dfree(d.dx); dfree(d.dxn);
    dfree(d.dq); dfree(d.datt); dfree(d.dh); dfree(d.logits); dfree(d.xattn); dfree(d.ban_dev);
    if (d.mlp_hand) { (void) hipFree(d.mlp_hand); d.mlp_hand = nullptr; d.mlp_arrive = nullptr; }
    dfree(d.aq); dfree(d.ads); dfree(d.aq16); dfree(d.wq16); dfree(d.att32); dfree(d.datt32);
    for (auto & sg : d.step_graphs) { if (sg.exec) (void) hipGraphExecDestroy(sg.exec); if (sg.graph) (void) hipGraphDestroy(sg.graph); sg = DeviceState::StepGraph{}; }
    if (d.step_dev) (void) hipFree(d.step_dev);
    if (d.sample_dev) (void) hipFree(d.sample_dev);
    if (d.filter_scratch) (void) hipFree(d.filter_scratch);
    if (d.draw_dev) (void) hipFree(d.draw_dev);
    if (d.draw_scratch) (void) hipFree(d.draw_scratch);
    if (d.draw_host) (void) hipHostFree(d.draw_host);
    if (d.step_host) (void) hipHostFree(d.step_host);
    if (d.sample_host) (void) hipHostFree(d.sample_host);
    if (d.pinned) (void) hipHostFree(d.pinned);
    if (d.stream && d.stream_own_queue) own_queue_stream_put(st->device, d.stream);
    else if (d.stream) (void) hipStreamDestroy(d.stream);
    delete st;
}

// ------------------------------------------------------------------------------------------------ mel
static bool ensure_mel_capacity(DeviceState & d, size_t n_pad, size_t n_mel_elems) {
    if (n_pad > d.pcm_cap) { dfree(d.pcm); if (!dalloc(d.pcm, n_pad + 1024)) return false; d.pcm_cap = n_pad + 1024; }
    if (n_mel_elems > d.mel_cap) { dfree(d.mel); if (!dalloc(d.mel, n_mel_elems)) return false; d.mel_cap = n_mel_elems; }
    return true;
}

static bool phase_events(DeviceState & d) {
    for (hipEvent_t & e : d.ph_ev) if (!e && !HIP_OK(hipEventCreate(&e))) return false;
    return true;
}
int64_t phase_settle(State & st, bool wait) {
    DeviceState & d = st.dev;
    if (!d.ph_mel && !d.ph_enc) return 0;
    if (wait) (void) hipStreamSynchronize(d.stream);
    float ms = 0.0f; double gpu_us = 0.0;
    if (d.ph_mel) { if (hipEventSynchronize(d.ph_ev[1]) == hipSuccess && hipEventElapsedTime(&ms, d.ph_ev[0], d.ph_ev[1]) == hipSuccess) { st.t_mel_us += (int64_t) (ms * 1e3); gpu_us += ms * 1e3; } d.ph_mel = false; }
    if (d.ph_enc) { if (hipEventSynchronize(d.ph_ev[3]) == hipSuccess && hipEventElapsedTime(&ms, d.ph_ev[2], d.ph_ev[3]) == hipSuccess) { st.t_encode_us += (int64_t) (ms * 1e3); gpu_us += ms * 1e3; } d.ph_enc = false; }
    return d.ph_host0 + (int64_t) gpu_us;
}

bool pcm_to_mel(whisper_context & ctx, const float * samples, int n_samples, bool samples_on_device, bool sync, bool defer) {
    if (!compute_ready(ctx, __func__)) return false;
    State & st = *ctx.state; DeviceState & d = st.dev;
    const int64_t t0 = time_us();
    const int n_mel = ctx.model.n_filt_mel;
    hipStream_t ms = d.mel_stream ? d.mel_stream : d.stream;
    // Appendix G: padded = [200 reflect | n | 480000 + 200 zeros]
    const int64_t n_pad = (int64_t) n_samples + 480000 + 400;
    const int n_len = (int) ((n_pad - 400) / 160);
    const int n_len_org = 1 + (n_samples + 200 - 400) / 160;
    const int n_valid = n_samples + 200;
    const int n_fft_frames = std::min(n_valid / 160 + 1, n_len);
    if (!ensure_mel_capacity(d, (size_t) n_pad + (size_t) n_samples, (size_t) n_mel * n_len)) return false;
    defer = defer && ms == d.stream && phase_events(d);
    if (defer) { (void) phase_settle(st, true); HIP_TRY(hipEventRecord(d.ph_ev[0], ms)); }
    const float * src = samples;
    if (!samples_on_device) {                      // stage the borrowed host PCM behind the padded image
        float * stage = d.pcm + n_pad;
        HIP_TRY(hipMemcpyAsync(stage, samples, (size_t) n_samples * 4, hipMemcpyHostToDevice, ms));
        src = stage;
    }
    d.last_pcm = src; d.last_pcm_n = n_samples;
    k::mel_pad(src, n_samples, d.pcm, (int) n_pad, ms, (int *) d.mel_max);
    k::mel_frames(d.pcm, n_valid, n_fft_frames, n_len, n_mel, ctx.w.mel_filters, ctx.w.mel_ranges, ctx.w.mel_taps, d.mel, (int *) d.mel_max, ms);
    k::mel_normalize(d.mel, n_mel * n_len, (const int *) d.mel_max, ms);
    st.mel.n_len = n_len; st.mel.n_len_org = n_len_org; st.mel.n_mel = n_mel;
    if (defer) { HIP_TRY(hipEventRecord(d.ph_ev[1], ms)); d.ph_mel = true; d.ph_host0 = t0; return true; }
    if (sync) HIP_TRY(hipStreamSynchronize(ms));          // lock-step chunks: one sync for all chunks (batch.cpp)
    st.t_mel_us += time_us() - t0;
    return true;
}

static bool ensure_copy_stream(DeviceState & d) {
    if (d.copy_stream) return true;
    // lowest priority: the envelope is needed at emission time only, whatever shares the chip with it goes first
    int prio_lo = 0, prio_hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (!HIP_OK(hipStreamCreateWithPriority(&d.copy_stream, hipStreamNonBlocking, prio_lo))) HIP_TRY(hipStreamCreateWithFlags(&d.copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&d.energy_ev, hipEventDisableTiming));
    return true;
}
static hipStream_t envelope_stream(const DeviceState & d) { return d.energy_wait_stream ? d.energy_wait_stream : d.copy_stream; }

bool pcm_to_mel_batch(whisper_context & ctx, const std::vector<State *> & states, const float * const * pcm, const int * n_samples,
                      bool samples_on_device, bool envelopes) {
    if (!compute_ready(ctx, __func__)) return false;
    DeviceState & pd = ctx.state->dev;                      // the calling state: its stream carries the launches
    hipStream_t s = pd.stream;
    const int n_mel = ctx.model.n_filt_mel;
    const int64_t t0 = time_us();
    k::MelBatch mb{}; int nb = 0; std::vector<State *> in;
    if (envelopes && !ensure_copy_stream(pd)) return false;
    for (size_t r = 0; r < states.size(); ++r) {
        const int n = n_samples[r];
        if (n <= 0) continue;
        if (nb >= 16) return false;
        State & st = *states[r]; DeviceState & d = st.dev;
        const int64_t n_pad = (int64_t) n + 480000 + 400;   // Appendix G, as pcm_to_mel
        const int n_len = (int) ((n_pad - 400) / 160);
        if (!ensure_mel_capacity(d, (size_t) n_pad + (size_t) n, (size_t) n_mel * n_len)) return false;
        const float * src = pcm[r];
        if (!samples_on_device) {
            float * stage = d.pcm + n_pad;
            HIP_TRY(hipMemcpyAsync(stage, pcm[r], (size_t) n * 4, hipMemcpyHostToDevice, s));
            src = stage;
        }
        d.last_pcm = src; d.last_pcm_n = n;
        st.mel.n_len = n_len; st.mel.n_len_org = 1 + (n + 200 - 400) / 160; st.mel.n_mel = n_mel;
        mb.pcm[nb] = src; mb.pad[nb] = d.pcm; mb.mel[nb] = d.mel; mb.gmax[nb] = (int *) d.mel_max; mb.n[nb] = n;
        if (envelopes) {
            // the previous call's envelope of this state may still be in flight on whichever stream wrote it
            if (d.energy_pending && envelope_stream(d)) { HIP_TRY(hipStreamSynchronize(envelope_stream(d))); d.energy_pending = false; }
            const size_t nblk = (size_t) n / 256 + 2;
            if ((size_t) n > d.energy_cap) {                // (energy_cap = capacity of the pinned image AND layout stride of the device image: kept in step with signal_energy_device)
                st.energy = nullptr; st.energy_n = 0; st.energy_bmin = st.energy_bmax = nullptr;
                if (d.energy_host) (void) hipHostFree(d.energy_host);
                d.energy_host = nullptr; d.energy_cap = 0;
                if (!HIP_OK(hipHostMalloc((void **) &d.energy_host, ((size_t) n + 2 * ((size_t) n / 256 + 2)) * 4, hipHostMallocDefault))) return false;
                d.energy_cap = (size_t) n;
            }
            const size_t need = d.energy_cap + 2 * nblk;
            if (d.energy_dev_cap < need) { dfree(d.energy); d.energy_dev_cap = 0; if (!dalloc(d.energy, need)) return false; d.energy_dev_cap = need; }
            mb.energy[nb] = d.energy; mb.bmin[nb] = d.energy + d.energy_cap; mb.bmax[nb] = d.energy + d.energy_cap + nblk;
        }
        in.push_back(&st); ++nb;
    }
    if (nb == 0) return true;
    if (envelopes) {
        // the envelope kernel runs on the side stream BESIDE the mel kernels (both only read the samples) and the main stream takes it back
        // before the encoder: left to run into the encoder, its 15 000 workgroups took CUs from the persistent GEMMs (8 chunks: encoder
        // 1.47 -> 1.55 ms, measured)
        if (!pd.mel_ev) HIP_TRY(hipEventCreateWithFlags(&pd.mel_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(pd.energy_ev, s));           // behind the staging of the samples
        HIP_TRY(hipStreamWaitEvent(pd.copy_stream, pd.energy_ev, 0));
        k::signal_energy_batch(mb, nb, 32, pd.copy_stream);
        HIP_TRY(hipEventRecord(pd.mel_ev, pd.copy_stream));
        for (State * st : in) {
            DeviceState & d = st->dev;
            d.energy_device_only = true; d.energy_unflushed = false; d.energy_pending = true; d.energy_wait_stream = pd.copy_stream;
        }
    }
    k::mel_batch(mb, nb, n_mel, ctx.w.mel_filters, ctx.w.mel_ranges, ctx.w.mel_taps, s);
    if (envelopes) HIP_TRY(hipStreamWaitEvent(s, pd.mel_ev, 0));
    ctx.state->t_mel_us += time_us() - t0;
    return hipGetLastError() == hipSuccess;
}

bool signal_energy_device(whisper_context & ctx, int hw, bool sync, int via_dma) {
    State & st = *ctx.state; DeviceState & d = st.dev;
    const int n = d.last_pcm_n;
    if (!d.last_pcm || n <= 0) return false;
    if (!ensure_copy_stream(d)) return false;
    if (d.energy_pending) { HIP_TRY(hipStreamSynchronize(envelope_stream(d))); d.energy_pending = false; }   // previous envelope still being written
    d.energy_wait_stream = nullptr;
    d.energy_device_only = false;
    if ((size_t) n > d.energy_cap) {
        st.energy = nullptr; st.energy_n = 0; st.energy_bmin = st.energy_bmax = nullptr;
        if (d.energy_host) (void) hipHostFree(d.energy_host);
        d.energy_host = nullptr; d.energy_cap = 0;
        // envelope [n] | block minima [n/256] | block maxima [n/256]
        if (!HIP_OK(hipHostMalloc((void **) &d.energy_host, ((size_t) n + 2 * ((size_t) n / 256 + 2)) * 4, hipHostMallocDefault))) return false;
        d.energy_cap = (size_t) n;
    }
    // The kernel runs on the side stream, behind the staging of the samples, and stores straight into pinned host
    // memory: no memcpy call, and the 1.9 MB of PCIe writes overlap the encoder on the main stream.
    HIP_TRY(hipEventRecord(d.energy_ev, d.mel_stream ? d.mel_stream : d.stream));
    HIP_TRY(hipStreamWaitEvent(d.copy_stream, d.energy_ev, 0));
    {
        const size_t nb = (size_t) n / 256 + 2;
        // via_dma (several chunks per call): the kernel writes a device buffer (~10 us) and the copy engine moves it to the pinned image —
        // as stores from the kernel, 8 x 1875 workgroups sat on PCIe writes in the CUs' wave slots for 0.25-0.35 ms beside the mel kernels
        // or the encoder, whichever they were queued next to.  One chunk: the direct stores (no copy call on the host's critical path).
        const size_t need = d.energy_cap + 2 * nb;
        if (via_dma && d.energy_dev_cap < need) { dfree(d.energy); d.energy_dev_cap = 0; if (dalloc(d.energy, need)) d.energy_dev_cap = need; else via_dma = 0; }
        if (via_dma == 3) {
            k::signal_energy(d.last_pcm, n, hw, d.energy, d.energy + d.energy_cap, d.energy + d.energy_cap + nb, d.copy_stream);
            d.energy_device_only = true; d.energy_unflushed = false;
        } else
        if (via_dma == 2) {
            // the kernel runs now, into HBM (~10 us); the copy to the pinned image is signal_energy_flush()'s thin kernel, later
            k::signal_energy(d.last_pcm, n, hw, d.energy, d.energy + d.energy_cap, d.energy + d.energy_cap + nb, d.copy_stream);
            d.energy_unflushed = true;
        } else
        if (via_dma) {
            k::signal_energy(d.last_pcm, n, hw, d.energy, d.energy + d.energy_cap, d.energy + d.energy_cap + nb, d.copy_stream);
            HIP_TRY(hipMemcpyAsync(d.energy_host, d.energy, (size_t) n * 4, hipMemcpyDeviceToHost, d.copy_stream));
            HIP_TRY(hipMemcpyAsync(d.energy_host + d.energy_cap, d.energy + d.energy_cap, 2 * nb * 4, hipMemcpyDeviceToHost, d.copy_stream));
        } else
        k::signal_energy(d.last_pcm, n, hw, d.energy_host, d.energy_host + d.energy_cap, d.energy_host + d.energy_cap + nb, d.copy_stream);
    }
    d.energy_pending = true;
    return sync ? signal_energy_wait(st) : true;
}

bool signal_energy_flush(State & st) {
    DeviceState & d = st.dev;
    if (!d.energy_pending || !d.energy_unflushed) return true;
    static const int wgs = getenv("WMI_ENVELOPE_COPY_WGS") ? atoi(getenv("WMI_ENVELOPE_COPY_WGS")) : 2;
    const size_t nb = (size_t) d.last_pcm_n / 256 + 2;
    // envelope [n] and the block extrema behind energy_cap: the pinned image has the device buffer's layout
    k::copy_thin(d.energy, d.energy_host, (size_t) d.last_pcm_n * 4, wgs, d.copy_stream);
    k::copy_thin(d.energy + d.energy_cap, d.energy_host + d.energy_cap, 2 * nb * 4, 1, d.copy_stream);
    d.energy_unflushed = false;
    return hipGetLastError() == hipSuccess;
}

bool ts_refine_device(State & st, const k::TsTok * in, int n, k::TsOut * out) {
    DeviceState & d = st.dev;
    constexpr int CAP = 448;
    if (n <= 0) return true;
    hipStream_t es = envelope_stream(d);
    if (n > CAP || !d.energy || !es) return false;
    if (!d.ts_host && !HIP_OK(hipHostMalloc(&d.ts_host, CAP * (sizeof(k::TsTok) + sizeof(k::TsOut)), hipHostMallocDefault))) return false;
    k::TsTok * hin = (k::TsTok *) d.ts_host; k::TsOut * hout = (k::TsOut *) (hin + CAP);
    memcpy(hin, in, (size_t) n * sizeof(k::TsTok));
    const size_t nb = (size_t) d.last_pcm_n / 256 + 2;
    // the kernel reads its records from, and writes its results to, the pinned block; on the envelope's own stream (behind the envelope kernel)
    k::ts_refine(d.energy, d.energy + d.energy_cap, d.energy + d.energy_cap + nb, d.last_pcm_n, hin, hout, n, es);
    HIP_TRY(hipStreamSynchronize(es));
    memcpy(out, hout, (size_t) n * sizeof(k::TsOut));
    return true;
}

bool signal_energy_wait(State & st) {
    DeviceState & d = st.dev;
    if (!d.energy_pending) return true;
    if (d.energy_device_only) {
        HIP_TRY(hipStreamSynchronize(envelope_stream(d)));
        st.energy = nullptr; st.energy_bmin = st.energy_bmax = nullptr; st.energy_n = d.last_pcm_n; st.energy_on_device = true;
        d.energy_pending = false;
        return true;
    }
    st.energy_on_device = false;
    if (d.energy_unflushed && !signal_energy_flush(st)) return false;
    HIP_TRY(hipStreamSynchronize(d.copy_stream));
    st.energy = d.energy_host; st.energy_n = d.last_pcm_n;
    st.energy_bmin = d.energy_host + d.energy_cap; st.energy_bmax = st.energy_bmin + ((size_t) d.last_pcm_n / 256 + 2);
    d.energy_pending = false;
    return true;
}

bool set_mel(whisper_context & ctx, const float * data, int n_len, int n_mel) {
    if (!compute_ready(ctx, __func__)) return false;
    State & st = *ctx.state; DeviceState & d = st.dev;
    if (!ensure_mel_capacity(d, 0, (size_t) n_len * n_mel)) return false;
    HIP_TRY(hipMemcpyAsync(d.mel, data, (size_t) n_len * n_mel * 4, hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipStreamSynchronize(d.stream));
    st.mel.n_len = n_len; st.mel.n_len_org = n_len; st.mel.n_mel = n_mel;
    return true;
}

// ------------------------------------------------------------------------------------------------ encoder
bool encode(whisper_context & ctx, int mel_offset, bool defer) {
    if (!compute_ready(ctx, __func__)) return false;
    State & st = *ctx.state; DeviceState & d = st.dev; const Weights & w = ctx.w; const HParams & hp = ctx.model.hp;
    const int64_t t0 = time_us();
    d.chain_valid = false;
    defer = defer && !ctx.model.quantised && phase_events(d);
    if (defer) {
        if (d.ph_enc) (void) phase_settle(st, true);        // (an encoder pass whose decoder never ran)
        HIP_TRY(hipEventRecord(d.ph_ev[2], d.stream));
        if (!d.ph_mel) d.ph_host0 = t0;
    }
    const int T = st.exp_n_audio_ctx > 0 ? st.exp_n_audio_ctx : hp.n_audio_ctx;
    const int S = hp.n_audio_state, H = hp.n_audio_head, La = hp.n_audio_layer, Lt = hp.n_text_layer, nm = hp.n_mels;
    hipStream_t s = d.stream;
    if (st.mel.n_mel != nm || d.mel == nullptr) { WMI_ERR("%s: no mel spectrogram (n_mel %d, expected %d)\n", __func__, st.mel.n_mel, nm); return false; }

    // conv front-end (row a2): two implicit GEMMs over token-major buffers with a zero row on each side
    const int rows_mel = 2 * T + 6;
    k::mel_slice(d.mel, st.mel.n_len, nm, mel_offset, 2 * T, d.mel_t, nm, rows_mel, s);
    {
        k::GemmArgs a{};
        a.A = d.mel_t; a.lda = nm; a.W = w.conv1_w; a.ldw = w.conv1_k; a.M = 2 * T; a.N = S; a.K = w.conv1_k;
        a.bias = w.conv1_b; a.C = d.conv1 + S; a.ldc = S;                       // rows 1..2T ; rows 0 and 2T+1 stay zero
        k::gemm(k::EPI_F16_BIAS_GELU, a, s);
    }
    // the zero row behind the last frame holds stale data only if an earlier call ran with a larger audio_ctx (one launch fewer otherwise)
    if (d.conv1_max_T > T) k::fill_zero(d.conv1 + (size_t) (2 * T + 1) * S, (size_t) S * sizeof(__half), s);
    d.conv1_max_T = std::max(d.conv1_max_T, T);
    {
        k::GemmArgs a{};
        a.A = d.conv1; a.lda = 2 * S; a.W = w.conv2_w; a.ldw = w.conv2_k; a.M = T; a.N = S; a.K = w.conv2_k;
        a.bias = w.conv2_b; a.C = d.x; a.ldc = S; a.resid = w.e_pe; a.ldr = S; a.aux = d.embd_conv; a.ldaux = S;
        k::gemm(k::EPI_CONV2, a, s);
    }
    // encoder blocks (row a3)
    const float kq_scale = 1.0f / sqrtf((float) S / H);
    if (ctx.model.quantised) { if (!encode_layers_q(ctx, T)) return false; }
    else {
    for (int il = 0; il < La; ++il) {
        const EncLayerW & l = w.enc[il];
        k::layernorm(d.x, T, S, l.ln1_g, l.ln1_b, hp.eps, d.xn, nullptr, s);
        {
            k::GemmArgs a{};
            a.A = d.xn; a.lda = S; a.W = l.w_qkv; a.ldw = S; a.M = T; a.N = 3 * S; a.K = S; a.bias = l.b_qkv;
            a.C = d.q; a.ldc = S; a.aux = d.k; a.ldaux = S; a.aux2 = d.vt; a.ldaux2 = d.Tpad; a.S = S;
            k::gemm(k::EPI_QKV_ENC, a, s);
        }
        k::attn_encoder(d.q, d.k, d.vt, T, d.Tpad, S, H, kq_scale, d.att, s);
        {
            k::GemmArgs a{};
            a.A = d.att; a.lda = S; a.W = l.w_o; a.ldw = S; a.M = T; a.N = S; a.K = S; a.bias = l.b_o;
            a.C = d.x; a.ldc = S; a.resid = d.x; a.ldr = S;
            k::gemm(k::EPI_F32_BIAS_RESID, a, s);
        }
        k::layernorm(d.x, T, S, l.ln2_g, l.ln2_b, hp.eps, d.xn, nullptr, s);
        {
            k::GemmArgs a{};
            a.A = d.xn; a.lda = S; a.W = l.w_fc1; a.ldw = S; a.M = T; a.N = 4 * S; a.K = S; a.bias = l.b_fc1;
            a.C = d.h; a.ldc = 4 * S;
            k::gemm(k::EPI_F16_BIAS_GELU, a, s);
        }
        {
            k::GemmArgs a{};
            a.A = d.h; a.lda = 4 * S; a.W = l.w_fc2; a.ldw = 4 * S; a.M = T; a.N = S; a.K = 4 * S; a.bias = l.b_fc2;
            a.C = d.x; a.ldc = S; a.resid = d.x; a.ldr = S;
            k::gemm(k::EPI_F32_BIAS_RESID, a, s);
        }
    }
    k::layernorm(d.x, T, S, w.e_ln_g, w.e_ln_b, hp.eps, d.enc_out_h, d.enc_out, s);
    // cross-attention K/V of every decoder layer in one GEMM (row a4): N = L * 2S
    {
        k::GemmArgs a{};
        a.A = d.enc_out_h; a.lda = S; a.W = w.w_ckv; a.ldw = S; a.M = T; a.N = Lt * 2 * S; a.K = S; a.bias = w.b_ckv;
        a.C = d.kvc_k; a.ldc = S; a.aux = d.kvc_v; a.ldaux = S; a.S = S; a.layer_stride = (int64_t) T * S;
        a.scale = powf((float) S / H, -0.25f);
        k::gemm(k::EPI_CROSS_KV, a, s);
    }
    }
    st.enc_n_ctx = T;
    st.n_encode++;
    if (defer) {                                            // no host wait: the caller's next launches queue up behind the encoder
        HIP_TRY(hipEventRecord(d.ph_ev[3], s)); d.ph_enc = true;
        return HIP_OK(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (!HIP_OK(hipGetLastError())) return false;
    {
        int64_t dt = time_us() - t0;
        const int64_t done = phase_settle(st, false);       // a deferred log-mel in front of this pass: its GPU time is not the encoder's
        if (done > t0) dt = std::max<int64_t>(0, dt - (done - t0));
        st.t_encode_us += dt;
    }
    return true;
}

static int stamps_reduce(whisper_context & ctx, const unsigned long long * buf, int n, double * out, int cap);

// ------------------------------------------------------------------------------------------------ decoder
bool decode(whisper_context & ctx, const Batch & batch) {
    if (!compute_ready(ctx, __func__)) return false;
    State & st = *ctx.state; DeviceState & d = st.dev; const Weights & w = ctx.w; const HParams & hp = ctx.model.hp;
    const int64_t t0 = time_us();
    d.chain_valid = false;                                  // this path overwrites the activation row a chained greedy step would start from
    KVCache & kv = st.kv_self;
    const int n = batch.n_tokens;
    if (n <= 0) return false;
    // the device and pinned staging buffers hold n_text_ctx rows (the reference's decoder graph is measured for that many, W/whisper.cpp:3098-3110)
    if (n > hp.n_text_ctx) { WMI_ERR("%s: %d tokens in one batch, the model's text context is %d\n", __func__, n, hp.n_text_ctx); return false; }
    if (!kv_find_slot(kv, batch)) return false;                   // W/whisper.cpp:2540
    kv.n = (uint32_t) kv_cell_max(kv);
    const int n_kv = (int) kv.n, kv_head = (int) kv.head, n_ctx = (int) kv.size;
    const int S = hp.n_text_state, H = hp.n_text_head, Lt = hp.n_text_layer, NV = hp.n_vocab;
    const int Tc = st.enc_n_ctx > 0 ? st.enc_n_ctx : (st.exp_n_audio_ctx > 0 ? st.exp_n_audio_ctx : hp.n_audio_ctx);
    hipStream_t s = d.stream;

    // host -> pinned -> device: tokens, positions, mask (W/whisper.cpp:2186-2226), rows wanting logits
    int32_t * p_tok = (int32_t *) d.pinned, * p_pos = p_tok + n, * p_rows = p_pos + n;
    float * p_mask = (float *) (p_rows + n);
    std::vector<int> rows;
    for (int i = 0; i < n; ++i) { p_tok[i] = batch.token[i]; p_pos[i] = batch.pos[i]; if (batch.logits[i]) rows.push_back(i); }
    for (size_t i = 0; i < rows.size(); ++i) p_rows[i] = rows[i];
    for (int j = 0; j < n; ++j) {
        const int32_t pos = batch.pos[j], seq = batch.seq_id[j];
        for (int i = 0; i < n_kv; ++i)
            p_mask[(size_t) j * n_kv + i] = (!kv.cells[i].has(seq) || kv.cells[i].pos > pos) ? -INFINITY : 0.0f;
    }
    // (the device block has the pinned block's layout for THIS n: tokens | positions | rows | mask, packed)
    d.d_pos = d.d_tokens + n; d.d_rows = d.d_pos + n; d.d_mask = (float *) (d.d_rows + n);
    // WMI_DECODE_TRACE=1 (debug): GPU time between the first and the last command of the call (events) beside the host's wall time
    static const bool trace = getenv("WMI_DECODE_TRACE") != nullptr;
    static thread_local hipEvent_t tr0 = nullptr, tr1 = nullptr;      // (probe state per thread: states of one context decode concurrently)
    if (trace) { if (!tr0) { (void) hipEventCreate(&tr0); (void) hipEventCreate(&tr1); } (void) hipEventRecord(tr0, s); }
    HIP_TRY(hipMemcpyAsync(d.d_tokens, p_tok, ((size_t) 3 * n + (size_t) n * n_kv) * 4, hipMemcpyHostToDevice, s));

    if (ctx.model.quantised) {
        // WMI_DECODE_STAMPS=k (debug): in-kernel stamps of the k-th several-row call's launches (start, body, the kernels' two mid marks)
        static const int stamp_call = getenv("WMI_DECODE_STAMPS") ? atoi(getenv("WMI_DECODE_STAMPS")) : -1;
        static thread_local int n_multi = 0;
        unsigned long long * sbuf = nullptr; constexpr int SMAXL = 512;
        if (stamp_call >= 0 && n > 1 && n_multi++ == stamp_call) {
            const size_t bytes = (size_t) SMAXL * k::STAMP_WAVES * 4 * sizeof(unsigned long long);
            if (HIP_OK(hipMalloc((void **) &sbuf, bytes))) { (void) hipMemsetAsync(sbuf, 0, bytes, s); k::stamp_enable(sbuf); }
        }
        const bool layers_ok = decode_layers_q(ctx, n, n_kv, kv_head, Tc, rows);
        if (sbuf) {
            const int nl = std::min(k::stamp_count(), SMAXL);
            k::stamp_enable(nullptr);
            (void) hipStreamSynchronize(s);
            std::vector<double> o((size_t) 6 * nl);
            const int got = stamps_reduce(ctx, sbuf, nl, o.data(), nl);
            for (int i = 0; i < got; ++i)
                if (o[6 * i + 3] > 0)
                    fprintf(stderr, "[wmi] stamps n=%d launch %3d: start %8.2f  last-start +%5.2f  body %5.2f  waves %4d  mark1 +%6.2f  mark2 +%6.2f  gap-before %5.2f\n", n, i,
                            o[6 * i], o[6 * i + 1] - o[6 * i], o[6 * i + 2] - o[6 * i], (int) o[6 * i + 3], o[6 * i + 4] > 0 ? o[6 * i + 4] - o[6 * i] : -1.0,
                            o[6 * i + 5] > 0 ? o[6 * i + 5] - o[6 * i] : -1.0, i ? o[6 * i] - o[6 * (i - 1) + 2] : 0.0);
            (void) hipFree(sbuf);
        }
        if (!layers_ok) return false;
        const int64_t t_enq = time_us();
        if (trace) (void) hipEventRecord(tr1, s);
        HIP_TRY(hipStreamSynchronize(s));
        if (!HIP_OK(hipGetLastError())) return false;
        int64_t dtq = time_us() - t0;
        { const int64_t done = phase_settle(st, false); if (done > t0) dtq = std::max<int64_t>(0, dtq - (done - t0)); }
        if (trace) {
            static thread_local int64_t t_prev_end = 0;
            float ms = 0.f; (void) hipEventElapsedTime(&ms, tr0, tr1);
            fprintf(stderr, "[wmi] decode n=%d n_kv=%d: wall %.0f us (enqueue done at %.0f) | first-to-last command on the GPU %.0f us | since the previous call returned %.0f us\n",
                    n, n_kv, (double) dtq, (double) (t_enq - t0), ms * 1e3, t_prev_end ? (double) (t0 - t_prev_end) : 0.0);
            t_prev_end = time_us();
        }
        if (n == 1)      { st.t_decode_us += dtq; st.n_decode++; }
        else if (n < 16) { st.t_batchd_us += dtq; st.n_batchd += n; }
        else             { st.t_prompt_us += dtq; st.n_prompt += n; }
        return true;
    }
    k::dec_embed(d.d_tokens, d.d_pos, n, S, w.d_te, w.d_pe, d.dx, s);
    const float kq_scale = powf((float) S / H, -0.25f);
    // WMI_DECODE_PATH=gemm|gemv forces one projection path (debug / A-B measurements); default: by batch size
    static const char * force_path = getenv("WMI_DECODE_PATH");
    static const int gemm_mask = getenv("WMI_GEMM_MASK") ? atoi(getenv("WMI_GEMM_MASK")) : 0;   // per-op override (debug)
    const bool skinny_default = (force_path && !strcmp(force_path, "gemm")) ? false : n <= 8;

    // generic projection: y = W . LN?(x) with a fused epilogue, through the weight-streaming kernel
    // (n <= 8) or the MFMA GEMM (prompt / initial_prompt batches)
    auto proj = [&](int op, int epi, const float * ln_g, const float * ln_b, const __half * a16, int K, int N, const __half * W,
                    const float * bias, void * C, int ldc, const float * resid, void * aux, int ldaux, void * aux2,
                    int ldaux2, float scale) {
        const bool skinny = n <= 8 && (skinny_default ? !((gemm_mask >> op) & 1) : false);
        if (skinny) {
            k::GemvArgs g{};
            g.x32 = d.dx; g.ln_g = ln_g; g.ln_b = ln_b; g.eps = hp.eps; g.a16 = a16; g.n = n; g.K = K; g.N = N; g.W = W;
            g.bias = bias; g.epi = epi; g.C = C; g.ldc = ldc; g.resid = resid; g.ldr = S; g.aux = aux; g.ldaux = ldaux;
            g.aux2 = aux2; g.ldaux2 = ldaux2; g.scale = scale; g.S = S; g.rows = nullptr;
            k::gemv(g, s);
        } else {
            const __half * A = a16;
            if (ln_g) { k::layernorm(d.dx, n, S, ln_g, ln_b, hp.eps, d.dxn, nullptr, s); A = d.dxn; }
            k::GemmArgs a{};
            a.A = A; a.lda = K; a.W = W; a.ldw = K; a.M = n; a.N = N; a.K = K; a.bias = bias; a.C = C; a.ldc = ldc;
            a.resid = resid; a.ldr = S; a.aux = aux; a.ldaux = ldaux; a.aux2 = aux2; a.ldaux2 = ldaux2; a.scale = scale; a.S = S;
            k::gemm(epi, a, s);
        }
    };

    for (int il = 0; il < Lt; ++il) {
        const DecLayerW & l = w.dec[il];
        __half * ck = kv.k + ((size_t) il * n_ctx) * S, * cv = kv.v + ((size_t) il * n_ctx) * S;
        // self-attention: q | k -> cache | v -> cache  (W/whisper.cpp:2248-2290)
        proj(0, k::EPI_QKV_DEC, l.ln1_g, l.ln1_b, nullptr, S, 3 * S, l.w_qkv, l.b_qkv, d.dq, S, nullptr,
             ck + (size_t) kv_head * S, S, cv + (size_t) kv_head * S, S, kq_scale);
        k::attn_decoder(d.dq, n, S, H, ck, cv, n_kv, d.d_mask, n_kv, d.datt, s);
        proj(1, k::EPI_F32_BIAS_RESID, nullptr, nullptr, d.datt, S, S, l.w_o, l.b_o, d.dx, S, d.dx, nullptr, 0, nullptr, 0, 0.f);
        // cross-attention against the encoder K/V of this layer, no mask (W/whisper.cpp:2359-2433)
        proj(2, k::EPI_Q_SCALED, l.ln2_g, l.ln2_b, nullptr, S, S, l.w_cq, l.b_cq, d.dq, S, nullptr, nullptr, 0, nullptr, 0, kq_scale);
        static const bool xattn_single = getenv("WMI_XATTN_SINGLE") != nullptr;     // debug: one workgroup per (token, head)
        if (xattn_single) k::attn_decoder(d.dq, n, S, H, d.kvc_k + (size_t) il * Tc * S, d.kvc_v + (size_t) il * Tc * S, Tc, nullptr, 0, d.datt, s);
        else k::attn_cross_split(d.dq, n, S, H, d.kvc_k + (size_t) il * Tc * S, d.kvc_v + (size_t) il * Tc * S, Tc, d.xattn, d.datt, s);
        proj(3, k::EPI_F32_BIAS_RESID, nullptr, nullptr, d.datt, S, S, l.w_co, l.b_co, d.dx, S, d.dx, nullptr, 0, nullptr, 0, 0.f);
        // MLP
        proj(4, k::EPI_F16_BIAS_GELU, l.ln3_g, l.ln3_b, nullptr, S, 4 * S, l.w_fc1, l.b_fc1, d.dh, 4 * S, nullptr, nullptr, 0, nullptr, 0, 0.f);
        proj(5, k::EPI_F32_BIAS_RESID, nullptr, nullptr, d.dh, 4 * S, S, l.w_fc2, l.b_fc2, d.dx, S, d.dx, nullptr, 0, nullptr, 0, 0.f);
    }

    // final LN + logits = d_te . x for the rows that asked for them (the reference computes all rows
    // and copies out the flagged ones, W/whisper.cpp:2498, 2566-2572)
    st.logits.resize((size_t) n * NV);
    for (size_t r0 = 0; r0 < rows.size(); r0 += 8) {
        const int nr = (int) std::min<size_t>(8, rows.size() - r0);
        k::GemvArgs g{};
        g.x32 = d.dx; g.ln_g = w.d_ln_g; g.ln_b = w.d_ln_b; g.eps = hp.eps; g.n = nr; g.K = S; g.N = NV; g.W = w.d_te;
        g.epi = k::EPI_LOGITS; g.C = d.logits; g.ldc = NV; g.rows = d.d_rows + r0;
        k::gemv(g, s);
        if (d.keep_logits_on_device && rows.size() <= 8) continue;        // sample_rows_device() reads them where they are
        HIP_TRY(hipMemcpyAsync(d.pinned, d.logits, (size_t) nr * NV * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int r = 0; r < nr; ++r)
            memcpy(st.logits.data() + (size_t) rows[r0 + r] * NV, (const float *) d.pinned + (size_t) r * NV, (size_t) NV * 4);
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (!HIP_OK(hipGetLastError())) return false;

    int64_t dt = time_us() - t0;                                  // timing buckets, W/whisper.cpp:2583-2592
    { const int64_t done = phase_settle(st, false); if (done > t0) dt = std::max<int64_t>(0, dt - (done - t0)); }      // (the part of this call spent waiting for a deferred log-mel / encoder)
    if (n == 1)      { st.t_decode_us += dt; st.n_decode++; }
    else if (n < 16) { st.t_batchd_us += dt; st.n_batchd += n; }
    else             { st.t_prompt_us += dt; st.n_prompt += n; }
    return true;
}

} // namespace wmi

// ------------------------------------------------------------------------------------------------ greedy fast path
namespace wmi {

bool fast_path_enabled() {
    static const bool off = getenv("WMI_HOST_SAMPLING") != nullptr;      // debug / A-B: force the host filter + sampling path
    return !off;
}

// static part of the logit filters (W/whisper.cpp:4541-4593): one byte per vocabulary entry, rebuilt only when
// the parameters that feed it change
bool upload_static_ban(whisper_context & ctx, const whisper_full_params & params) {
    State & st = *ctx.state; DeviceState & d = st.dev; const Vocab & v = ctx.model.vocab;
    const uint64_t sig = (params.suppress_non_speech_tokens ? 1u : 0u) | (params.no_timestamps ? 2u : 0u) | (params.tdrz_enable ? 4u : 0u) | 8u;
    if (d.ban_dev && d.ban_sig == sig) return true;
    const int n = v.n_vocab;
    std::vector<uint8_t> ban(n, 0);
    auto B = [&](int id) { if (id >= 0 && id < n) ban[id] = 1; };
    B(v.not_); B(v.sot); B(v.nosp); if (!params.tdrz_enable) B(v.solm);
    B(v.translate); B(v.transcribe); B(v.prev);
    for (int i = 0; i < lang_count(); ++i) B(v.sot + 1 + i);
    if (params.no_timestamps) for (int i = v.beg; i < n; ++i) ban[i] = 1;
    if (params.suppress_non_speech_tokens) {
        Decoder tmp; tmp.i_batch = 0;                         // reuse the host filter on an all-zero logit row to harvest the list
        std::vector<float> saved; saved.swap(st.logits);
        st.logits.assign(n, 0.0f);
        for (int i = v.beg; i < n; ++i) st.logits[i] = -1e30f;   // keep the "timestamp mass beats text" rule from firing
        whisper_full_params p2 = params; p2.suppress_blank = false; p2.max_initial_ts = 0.0f; p2.logits_filter_callback = nullptr;
        tmp.sequence.tokens.push_back(whisper_token_data{0, 0, 0.f, 0.f, 0.f, 0.f, -1, -1, 0.f});   // not "initial"
        process_logits(ctx, tmp, p2, 0.0f);
        for (int i = 0; i < v.beg; ++i) if (tmp.logits[i] == -INFINITY) ban[i] = 1;
        st.logits.swap(saved);
    }
    if (!d.ban_dev && !dalloc(d.ban_dev, (size_t) n)) return false;
    HIP_TRY(hipMemcpyAsync(d.ban_dev, ban.data(), (size_t) n, hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipStreamSynchronize(d.stream));
    d.ban_sig = sig;
    return true;
}

// Completion of a step without a stream synchronisation: the last kernel of the step writes the step's sequence number
// into pinned host memory behind its result (k_filter_pick); the host spins on it (~1 us instead of the 10-20 us of an
// interrupt-driven hipStreamSynchronize, paid once per token).  Bounded: after ~2 s it falls back to a real
// synchronisation and reports what the stream says.
bool wait_for_sample(const k::SampleOut * r, int32_t want, hipStream_t s, int32_t * status) {
    // both halves of the record are written by one 16-byte store each and carry the step's number: a half whose tag matches
    // is complete (x86 loads are not reordered with each other: tag first, then the fields).  The tags' upper bits are the status of the
    // step's in-launch hand-offs (k::SAMPLE_TAG_*), handed back through `status`.
    const volatile int32_t * t0 = &r->seq0, * t1 = &r->seq;
    want &= k::SAMPLE_SEQ_MASK;
    auto both = [&]() { const int32_t a = *t0, b = *t1; if ((a & k::SAMPLE_SEQ_MASK) != want || a != b) return false; if (status) *status = a & ~k::SAMPLE_SEQ_MASK; return true; };
    static const bool no_spin = getenv("WMI_NO_SPIN") != nullptr;          // debug / A-B
    if (!no_spin) {
        const int64_t tb = time_us();
        for (uint32_t it = 1;; ++it) {
            if (both()) {
                // the device writes each 16-byte half with one store that carries the tag; the caller reads the fields behind this
                // return — re-read the tags behind a compiler barrier so that "tag, fields, tag" brackets what the caller copies
                asm volatile("" ::: "memory");
                if (both()) return true;
            }
            __builtin_ia32_pause();
            if ((it & 0xFFFF) == 0 && time_us() - tb > 2000000) break;
        }
    }
    if (!HIP_OK(hipStreamSynchronize(s))) return false;
    asm volatile("" ::: "memory");
    return both();
}

// Draws on the device ("argmax / top-k on GPU so only ~k numbers cross PCIe per step").  The logits rows of the
// last decode() stay in d.logits; per row the filters of process_logits (same rules as the greedy step), p = exp(l - lse) and the
// CDF search run in k_sample.hip; the mt19937 generators stay on the host and supply the uniform numbers.
bool sample_rows_device(whisper_context & ctx, const StepFilter * f, const int * rows, int n_rows, float temperature, int k,
                        const double * u, int tid_default, whisper_token_data * out) {
    State & st = *ctx.state; DeviceState & d = st.dev; const Vocab & v = ctx.model.vocab;
    if (n_rows < 1 || n_rows > 8 || k < 1 || k > 8) return false;
    const int64_t t0 = time_us();
    constexpr size_t OFF_U = 8 * sizeof(k::DecStep), OFF_OUT = OFF_U + 64 * sizeof(double), TOTAL = OFF_OUT + 64 * sizeof(k::SampleOut);
    if (!d.draw_dev) {
        if (!HIP_OK(hipMalloc(&d.draw_dev, TOTAL)) || !HIP_OK(hipHostMalloc(&d.draw_host, TOTAL, hipHostMallocDefault)) ||
            !HIP_OK(hipMalloc(&d.draw_scratch, k::filter_draw_scratch_bytes(8)))) return false;
    }
    hipStream_t s = d.stream;
    const int NV = v.n_vocab;
    int space_id = -1; { auto sp = v.token_to_id.find(" "); if (sp != v.token_to_id.end()) space_id = sp->second; }
    k::DecStep * hs = (k::DecStep *) d.draw_host; double * hu = (double *) ((char *) d.draw_host + OFF_U);
    for (int r = 0; r < n_rows; ++r) {
        memset(&hs[r], 0, sizeof(k::DecStep));
        hs[r].flags = (f[r].ban_blank ? 1 : 0) | (f[r].last_ts ? 2 : 0) | (f[r].penult_ts ? 4 : 0);
        hs[r].space_id = space_id; hs[r].eot = v.eot; hs[r].beg = v.beg; hs[r].n_vocab = NV;
        hs[r].ts_floor_end = f[r].ts_floor_end; hs[r].ts_initial_start = f[r].ts_initial_start;
        hs[r].temperature = temperature > 0.0f ? temperature : 0.0f;
        for (int c = 0; c < k; ++c) hu[r * k + c] = u[r * k + c];
    }
    // The kernels read the step records and the uniform numbers straight from the pinned block and write their results into it: two small
    // copies fewer per sampled step (WMI_DRAW_STAGED=1: through the device block, as before)
    static const bool staged = getenv("WMI_DRAW_STAGED") != nullptr;
    if (staged) HIP_TRY(hipMemcpyAsync(d.draw_dev, d.draw_host, OFF_OUT, hipMemcpyHostToDevice, s));
    void * const blk = staged ? d.draw_dev : d.draw_host;
    const k::DecStep * ds = (const k::DecStep *) blk; const double * du = (const double *) ((char *) blk + OFF_U);
    k::SampleOut * dout = (k::SampleOut *) ((char *) blk + OFF_OUT);
    // rows that sit next to each other in d.logits go in one launch; the first step of a window draws every decoder from row 0
    bool contiguous = true;
    for (int r = 0; r < n_rows; ++r) contiguous = contiguous && rows[r] == rows[0] + r;
    if (contiguous) k::filter_draw(d.logits + (size_t) rows[0] * NV, d.ban_dev, ds, du, k, dout, d.draw_scratch, s, n_rows, tid_default);
    else for (int r = 0; r < n_rows; ++r)
        k::filter_draw(d.logits + (size_t) rows[r] * NV, d.ban_dev, ds + r, du + r * k, k, dout + r * k, d.draw_scratch, s, 1, tid_default);
    k::SampleOut * hout = (k::SampleOut *) ((char *) d.draw_host + OFF_OUT);
    if (staged) HIP_TRY(hipMemcpyAsync(hout, dout, (size_t) n_rows * k * sizeof(k::SampleOut), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < n_rows * k; ++i)
        out[i] = whisper_token_data{ hout[i].id, hout[i].tid, hout[i].p, hout[i].plog, hout[i].pt, hout[i].ptsum, -1, -1, 0.0f };
    st.t_sample_us += time_us() - t0; st.n_sample += n_rows;
    return true;
}

// the kernels of one greedy step; every per-step quantity is read from DecStep on the device, so the same launch
// sequence can be replayed as a graph
// probe only (bench kernel 20): which kernel kinds of the step are enqueued — bit 0 embed, 1 qkv, 2 self-attn+out, 3 cross scores,
// 4 cross combine+out, 5 mlp.0, 6 mlp.2, 7 logits, 8 filters
static unsigned g_step_mask = ~0u;

// long_kv: the self cache holds more than 64 cells.  The attention fused into the out projection's prologue is a per-wavefront
// routine for <= 64 keys (every workgroup recomputes it); beyond that its barrier-separated fall-back ran at ~95 us per
// layer (bench: the uncapped transcription), so long caches take the (row, head)-parallel attention kernel + a plain projection.
// chained: the step starts from what the previous step's pick kernel prepared on the device (token, position, cache head, activation
// row): no embedding launch; the host's record (filter flags, sequence number) reaches the device through the extra workgroup of
// the last mlp.2 launch.
static std::atomic<int> g_busy_transcriptions[64];
static thread_local int t_busy_depth = 0;                 // a thread is counted once however its entry points nest (wmi_full_batch -> full(), *_with_state -> full())
BusyScope::BusyScope(int device) : dev(device & 63), counted(t_busy_depth++ == 0) { if (counted) g_busy_transcriptions[dev].fetch_add(1, std::memory_order_relaxed); }
BusyScope::~BusyScope() { --t_busy_depth; if (counted) g_busy_transcriptions[dev].fetch_sub(1, std::memory_order_relaxed); }
int busy_transcriptions(int device) { return g_busy_transcriptions[device & 63].load(std::memory_order_relaxed); }

// per-device turns for greedy steps (see decode_greedy_step): at most `cap` steps of different states / contexts in flight on a device.
// Round 5 took ONE turn per device (a mutex): several contexts' dependent launch chains, interleaved freely, stretched every launch boundary
// (six contexts 7.4 ms per transcription).  Round 6 measured how many chains the device carries well — N threads on N states of one context,
// base.en, ms per transcription (scratch/r06_conc_time.py, profiles/r06f_*):
//     N                        1      2      3      4      6      8
//     one turn (round 5)      3.23   2.98   3.11   2.98   3.09   3.20      the chains simply alternate: nothing gained by the threads
//     two turns               3.24   2.00   1.96   2.46   1.88   2.26
//     three turns             3.23   2.01   1.59   4.52   2.23   3.76
//     no turns                3.24   1.91   1.58   5.96   4.11   4.28
// Three chains side by side are the best the device does, but a fourth anywhere (another thread's encoder counts) breaks it — the runtime
// multiplexes streams onto few hardware queues (DESIGN hazard 21) — so two turns: 1.3 - 1.7 x one transcription at a time, for every N.
// WMI_STEP_SLOTS = 1 is round 5's form.  Not a fair queue: whoever gets a free turn next is as good as anyone for throughput, and a fair
// ticket's next holder may be a descheduled thread.
// The second turn is for steps of the SAME context only (states of one context share one copy of the weights): with separate contexts side
// by side two turns are worse than one — 3 contexts x 150 transcriptions 8.2 - 8.4 ms per transcription against 3.8 with one turn
// (scratch/stress_pair.py) — while N states of one context gain from the second (table above).
namespace {
struct StepSlots { std::atomic<int> used{0}; std::atomic<const void *> owner{nullptr}; };
StepSlots g_step_slots[64];
struct StepTicket {
    StepSlots * sl = nullptr;
    StepTicket(int device, bool take, const void * ctx) {
        if (!take) return;
        static const int cap = getenv("WMI_STEP_SLOTS") ? std::max(1, atoi(getenv("WMI_STEP_SLOTS"))) : 2;
        StepSlots & s = g_step_slots[device & 63];
        for (uint32_t it = 0;; ++it) {
            int u = s.used.load(std::memory_order_acquire);
            if (u == 0) {
                if (s.used.compare_exchange_weak(u, 1, std::memory_order_acquire)) { s.owner.store(ctx, std::memory_order_release); sl = &s; return; }
            } else if (u < cap && s.owner.load(std::memory_order_acquire) == ctx) {
                // (the owner word is written just behind the 0 -> 1 transition: a step that reads the previous owner in that window runs beside
                //  another context's step once — slower, not wrong)
                if (s.used.compare_exchange_weak(u, u + 1, std::memory_order_acquire)) { sl = &s; return; }
            }
            if ((it & 1023) == 1023) std::this_thread::yield(); else __builtin_ia32_pause();      // (a holder is ~150 us from releasing)
        }
    }
    void release() { if (sl) { sl->used.fetch_sub(1, std::memory_order_release); sl = nullptr; } }
    ~StepTicket() { release(); }
};
}

constexpr int PAIR_FAULT_WORD = 4;           // d.mlp_arrive as 32-bit words: [0], [1] the MLP launches' tags, [2], [3] the front launches', [6], [7] the cross-attention back's, [4] the hand-offs' status (k::MlpPairArgs::fault)
static void enqueue_greedy_step(whisper_context & ctx, int Tc, bool long_kv = false, bool chained = false, bool solo = true) {
    if (ctx.model.quantised) { enqueue_greedy_step_q(ctx, Tc); return; }
    State & st = *ctx.state; DeviceState & d = st.dev; const Weights & w = ctx.w; const HParams & hp = ctx.model.hp;
    KVCache & kv = st.kv_self;
    const int S = hp.n_text_state, H = hp.n_text_head, Lt = hp.n_text_layer, NV = hp.n_vocab, n_ctx = (int) kv.size;
    hipStream_t s = d.stream;
    const k::DecStep * stp = (const k::DecStep *) d.step_dev;
    const float kq_scale = powf((float) S / H, -0.25f);
    static const bool dbg = getenv("WMI_DEBUG_SYNC") != nullptr;
    auto chk = [&](const char * what, int il) {
        if (!dbg) return;
        const hipError_t e = hipStreamSynchronize(s);
        fprintf(stderr, "[wmi] greedy step: %s layer %d -> %s\n", what, il, hipGetErrorString(e));
    };
    // step parameters come from pinned host memory, the result goes back into pinned host memory: the replayed
    // graph contains kernels only (memcpy nodes cost tens of microseconds each on this stack)
    const unsigned M = g_step_mask;
    if ((M & 1) && !chained) k::dec_embed_step((const k::DecStep *) d.step_host, (k::DecStep *) d.step_dev, S, w.d_te, w.d_pe, d.dx, s); chk("embed", -1);
    auto gv = [&](int epi, const float * lg, const float * lb, const __half * a16, int K, int N, const __half * W, const float * bias,
                  void * C, int ldc, const float * resid, void * aux, void * aux2, float scale, const int32_t * row_off) {
        k::GemvArgs g{};
        g.x32 = d.dx; g.ln_g = lg; g.ln_b = lb; g.eps = hp.eps; g.a16 = a16; g.n = 1; g.K = K; g.N = N; g.W = W; g.bias = bias;
        g.epi = epi; g.C = C; g.ldc = ldc; g.resid = resid; g.ldr = S; g.aux = aux; g.ldaux = S; g.aux2 = aux2; g.ldaux2 = S;
        g.scale = scale; g.S = S; g.rows = nullptr; g.row_off = row_off;
        k::gemv(g, s);
    };
    // the MLP form is decided once for the whole step: the paired launches' tags alternate between two words from launch to launch (an even
    // number of layers keeps that up across steps), so either every layer pairs or none does
    const k::Knobs & kn = k::knobs();
    const bool paired = (M & 32) && (M & 64) && !kn.no_mlp_pair && solo && d.mlp_hand && (Lt & 1) == 0 && k::mlp_pair_usable(S, chained);
    // the same decision for the front of the layers (k::front): short caches only; its status goes through the MLP pair's word, so it
    // runs where that pair does (the pick kernel reports the word, a fault re-runs the step without either)
    const bool fronted = paired && !long_kv && (M & 2) && (M & 4) && !kn.no_front && k::front_usable(S);
    const bool backed = paired && (M & 8) && (M & 16) && !kn.no_xback && k::xback_usable(S, H, Tc);
    for (int il = 0; il < Lt; ++il) {
        const DecLayerW & l = w.dec[il];
        __half * ck = kv.k + ((size_t) il * n_ctx) * S, * cv = kv.v + ((size_t) il * n_ctx) * S;
        if (fronted) {     // LN + q|k|v, the self-attention (once per head) and the out projection as ONE launch with two hand-offs (k::front)
            k::FrontArgs f{};
            f.x = d.dx; f.xout = d.dx; f.ln_g = l.ln1_g; f.ln_b = l.ln1_b; f.eps = hp.eps; f.S = S; f.Wqkv = l.w_qkv; f.bqkv = l.b_qkv; f.scale = kq_scale;
            f.q16 = d.dq; f.ck = ck; f.cv = cv; f.kv_head = &stp->kv_head; f.n_kv = &stp->n_kv; f.cap = hp.n_text_ctx; f.Wo = l.w_o; f.bo = l.b_o;
            f.gq = (unsigned long long *) ((unsigned char *) d.mlp_hand + (size_t) 16 * S + 64); f.ga = f.gq + 3 * S / 2;
            f.epoch = (uint32_t *) d.mlp_arrive + 2; f.par = il & 1; f.fault = (uint32_t *) d.mlp_arrive + PAIR_FAULT_WORD;
            f.spin_cap = kn.pair_spin_cap; f.withhold = kn.front_withhold;
            k::front(f, s); chk("front", il);
        } else {
        if (M & 2) gv(k::EPI_QKV_DEC, l.ln1_g, l.ln1_b, nullptr, S, 3 * S, l.w_qkv, l.b_qkv, d.dq, S, nullptr, ck, cv, kq_scale, &stp->kv_head); chk("qkv", il);
        if ((M & 4) && long_kv) {
            k::self_attn_rows(d.dq, 1, S, ck, cv, 0, &stp->n_kv, 0, hp.n_text_ctx, d.datt, s, nullptr, true);
            gv(k::EPI_F32_BIAS_RESID, nullptr, nullptr, d.datt, S, S, l.w_o, l.b_o, d.dx, S, d.dx, nullptr, nullptr, 0.f, nullptr); chk("self-attn, out", il);
        } else
        if (M & 4) {   // self-attention over the cache, recomputed in the out-projection's prologue (one launch fewer)
            k::GemvArgs g{};
            g.sa_q = d.dq; g.sa_k = ck; g.sa_v = cv; g.sa_nkv = &stp->n_kv; g.sa_cap = hp.n_text_ctx;
            g.n = 1; g.K = S; g.N = S; g.W = l.w_o; g.bias = l.b_o; g.epi = k::EPI_F32_BIAS_RESID;
            g.C = d.dx; g.ldc = S; g.resid = d.dx; g.ldr = S; g.S = S;
            k::gemv(g, s); chk("self-attn+out", il);
        }
        }
        {   // LN2 + cross query folded into the score kernel; partials combined inside the out-projection's prologue
            static const bool unfused_q = getenv("WMI_XATTN_UNFUSED_Q") != nullptr;          // debug / A-B
            const float * po = nullptr, * pl = nullptr, * pm = nullptr; int ns = 0;
            if (!(M & 8)) { k::attn_cross_partials_layout(1, H, Tc, d.xattn, &po, &pl, &pm, &ns); }
            else if (backed) {   // LN + cross query + key slices, the combine (once per head) and the out projection as ONE launch (k::xback)
                k::XbackArgs xb{};
                xb.x = d.dx; xb.xout = d.dx; xb.ln_g = l.ln2_g; xb.ln_b = l.ln2_b; xb.eps = hp.eps; xb.S = S; xb.wq = l.w_cq; xb.bq = l.b_cq; xb.qscale = kq_scale;
                xb.kc = d.kvc_k + (size_t) il * Tc * S; xb.vc = d.kvc_v + (size_t) il * Tc * S; xb.T = Tc; xb.Wo = l.w_co; xb.bo = l.b_co;
                xb.gp = (unsigned long long *) ((unsigned char *) d.mlp_hand + (size_t) 32 * S + 64); xb.ga = xb.gp + 8 * 8 * 66;
                xb.epoch = (uint32_t *) d.mlp_arrive + 6; xb.par = il & 1; xb.fault = (uint32_t *) d.mlp_arrive + PAIR_FAULT_WORD;
                xb.spin_cap = kn.pair_spin_cap; xb.withhold = kn.xback_withhold;
                k::xback(xb, H, d.xattn, s); chk("cross-attn + out", il);
                goto mlp;
            }
            else if (unfused_q || S > 1536) {                  // the fused kernel keeps a whole row per wavefront in registers: S <= 1536
                gv(k::EPI_Q_SCALED, l.ln2_g, l.ln2_b, nullptr, S, S, l.w_cq, l.b_cq, d.dq, S, nullptr, nullptr, nullptr, kq_scale, nullptr);
                k::attn_cross_split_partials(d.dq, 1, S, H, d.kvc_k + (size_t) il * Tc * S, d.kvc_v + (size_t) il * Tc * S, Tc, d.xattn, &po, &pl, &pm, &ns, s);
            } else
            k::attn_cross_qsplit_partials(d.dx, l.ln2_g, l.ln2_b, hp.eps, l.w_cq, l.b_cq, kq_scale, 1, S, H,
                                          d.kvc_k + (size_t) il * Tc * S, d.kvc_v + (size_t) il * Tc * S, Tc, d.xattn, &po, &pl, &pm, &ns, s);
            chk("cross-attn", il);
            k::GemvArgs g{};
            g.comb_o = po; g.comb_l = pl; g.comb_m = pm; g.comb_ns = ns; g.n = 1; g.K = S; g.N = S; g.W = l.w_co; g.bias = l.b_co; g.epi = k::EPI_F32_BIAS_RESID;
            g.C = d.dx; g.ldc = S; g.resid = d.dx; g.ldr = S; g.S = S;
            if (M & 16) k::gemv(g, s);
        }
        mlp:
        // both MLP projections as ONE launch with an in-launch hand-off of the hidden row (k::mlp_pair; WMI_NO_MLP_PAIR=1: two launches)
        if (paired) {
            k::MlpPairArgs p{};
            p.x = d.dx; p.ln_g = l.ln3_g; p.ln_b = l.ln3_b; p.eps = hp.eps; p.S = S; p.W1 = l.w_fc1; p.b1 = l.b_fc1; p.W2 = l.w_fc2; p.b2 = l.b_fc2;
            p.epoch = (uint32_t *) d.mlp_arrive; p.par = il & 1; p.hand = d.mlp_hand; p.fault = (uint32_t *) d.mlp_arrive + PAIR_FAULT_WORD;
            p.spin_cap = kn.pair_spin_cap; p.withhold = kn.pair_withhold;
            if (chained && il == Lt - 1) { p.step_copy_src = d.step_host; p.step_copy_dst = d.step_dev; }
            k::mlp_pair(p, d.dx, s);
            continue;
        }
        if (M & 32) gv(k::EPI_F16_BIAS_GELU, l.ln3_g, l.ln3_b, nullptr, S, 4 * S, l.w_fc1, l.b_fc1, d.dh, 4 * S, nullptr, nullptr, nullptr, 0.f, nullptr);
        if ((M & 64) && chained && il == Lt - 1) {             // + the workgroup that mirrors the host's step record (filter flags, seq)
            k::GemvArgs g{};
            g.x32 = d.dx; g.eps = hp.eps; g.a16 = d.dh; g.n = 1; g.K = 4 * S; g.N = S; g.W = l.w_fc2; g.bias = l.b_fc2; g.epi = k::EPI_F32_BIAS_RESID;
            g.C = d.dx; g.ldc = S; g.resid = d.dx; g.ldr = S; g.ldaux = S; g.ldaux2 = S; g.S = S;
            g.step_copy_src = d.step_host; g.step_copy_dst = d.step_dev;
            k::gemv(g, s);
        } else
        if (M & 64) gv(k::EPI_F32_BIAS_RESID, nullptr, nullptr, d.dh, 4 * S, S, l.w_fc2, l.b_fc2, d.dx, S, d.dx, nullptr, nullptr, 0.f, nullptr);
    }
    chk("layers", Lt);
    // vocabulary projection; when the whole step runs, the logit filters' statistics pass rides in its epilogue (one partial per
    // workgroup into filter_scratch) and filter_argmax only picks
    int fused_parts = 0;
    {
        k::GemvArgs g{};
        g.x32 = d.dx; g.ln_g = w.d_ln_g; g.ln_b = w.d_ln_b; g.eps = hp.eps; g.n = 1; g.K = S; g.N = NV; g.W = w.d_te;
        g.epi = k::EPI_LOGITS; g.C = d.logits; g.ldc = NV; g.ldr = S; g.ldaux = S; g.ldaux2 = S; g.S = S;
        if ((M & 128) && (M & 256)) { g.fs_ban = d.ban_dev; g.fs_step = stp; g.fs_part = (k::FsPartial *) d.filter_scratch; fused_parts = k::gemv_fused_parts(g); }
        if (!fused_parts) { g.fs_ban = nullptr; g.fs_step = nullptr; g.fs_part = nullptr; }
        if (M & 128) k::gemv(g, s);
        chk("logits", Lt);
    }
    // the pick also prepares the next step on the device (token = pick, position / cache head + 1, x = te[pick] + pe[pos + 1])
    const k::ChainNext cn{ (k::DecStep *) d.step_dev, w.d_te, w.d_pe, d.dx, S, hp.n_text_ctx, paired ? (const uint32_t *) d.mlp_arrive + PAIR_FAULT_WORD : nullptr };
    if (M & 256) k::filter_argmax(d.logits, d.ban_dev, stp, (k::SampleOut *) d.sample_dev, d.filter_scratch, s, (k::SampleOut *) d.sample_host, 1, &cn, fused_parts); chk("filter", Lt);
}

bool decode_greedy_step(whisper_context & ctx, int32_t token, int32_t pos, const StepFilter & f, whisper_token_data & out) {
    if (!compute_ready(ctx, __func__)) return false;
    State & st = *ctx.state; DeviceState & d = st.dev; const HParams & hp = ctx.model.hp; const Vocab & v = ctx.model.vocab;
    const int64_t t0 = time_us();
    KVCache & kv = st.kv_self;
    // same slot bookkeeping as the general path (W/whisper.cpp:2540-2544) so that both paths can be mixed
    Batch & b = st.batch;
    b.n_tokens = 1; b.token[0] = token; b.pos[0] = pos; b.seq_id[0] = 0; b.logits[0] = 1;
    if (!kv_find_slot(kv, b)) return false;
    kv.n = (uint32_t) kv_cell_max(kv);
    // the fast path assumes the plain causal layout of a single greedy sequence: cells [0, head] all visible
    if ((int) kv.n != (int) kv.head + 1) { WMI_ERR("%s: unexpected KV layout (n=%u head=%u)\n", __func__, kv.n, kv.head); return false; }
    const int Tc = st.enc_n_ctx > 0 ? st.enc_n_ctx : hp.n_audio_ctx;

    if (!d.step_dev) {
        if (!HIP_OK(hipMalloc(&d.step_dev, sizeof(k::DecStep))) || !HIP_OK(hipMalloc(&d.filter_scratch, k::filter_scratch_bytes())) || !HIP_OK(hipMalloc(&d.sample_dev, sizeof(k::SampleOut))) ||
            !HIP_OK(hipHostMalloc(&d.step_host, sizeof(k::DecStep), hipHostMallocDefault)) ||
            !HIP_OK(hipHostMalloc(&d.sample_host, sizeof(k::SampleOut), hipHostMallocDefault))) return false;
        memset(d.sample_host, 0, sizeof(k::SampleOut));
    }
    k::DecStep * hs = (k::DecStep *) d.step_host;
    memset(hs, 0, sizeof(*hs));
    hs->token = token; hs->pos = pos; hs->n_kv = (int) kv.n; hs->kv_head = (int) kv.head;
    hs->flags = (f.ban_blank ? 1 : 0) | (f.last_ts ? 2 : 0) | (f.penult_ts ? 4 : 0);
    { auto sp = v.token_to_id.find(" "); hs->space_id = sp != v.token_to_id.end() ? sp->second : -1; }
    hs->eot = v.eot; hs->beg = v.beg; hs->n_vocab = v.n_vocab;
    hs->ts_floor_end = f.ts_floor_end; hs->ts_initial_start = f.ts_initial_start;

    hipStream_t s = d.stream;
    static const bool use_graph = getenv("WMI_NO_GRAPH") == nullptr;
    // two captured forms of the step: caches of <= 64 cells (self-attention inside the out projection) and longer ones
    const bool long_kv = (int) kv.n > 64 && !ctx.model.quantised;          // (the quantised step has one form for every length)
    static const bool no_chain = getenv("WMI_NO_CHAIN") != nullptr;         // debug / A-B
    const bool chained = !no_chain && !ctx.model.quantised && d.chain_valid && token == d.chain_token && pos == d.chain_pos &&
                         (int) kv.head == d.chain_head;
    d.chain_valid = false;
    // alone on the GPU as far as this process knows: the forms with kernels that wait inside a launch (k::mlp_pair); otherwise the plain chain
    // (both are bit-identical, so a transcription may change form from one step to the next)
    // d.pair_off: the in-launch hand-off has timed out once on this state (never again), or has been slow (not for a while: another PROCESS'
    // work on the device is invisible to the count above — the kernel's own poll count is what tells)
    if (d.pair_backoff > 0) --d.pair_backoff;
    const bool solo = busy_transcriptions(ctx.device) <= 1 && !d.pair_off && d.pair_backoff == 0;
    DeviceState::StepGraph & sg = d.step_graphs[(long_kv ? 1 : 0) | (chained ? 2 : 0) | (solo ? 0 : 4)];
    hipGraph_t & graph = sg.graph; hipGraphExec_t & exec = sg.exec; int & graph_T = sg.T;
    if (use_graph && exec && graph_T != Tc) {                           // encoder length changed: the captured step is stale
        (void) hipGraphExecDestroy(exec); (void) hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr;
    }
    // Capture + instantiate costs tens of milliseconds; the streaming node changes audio_ctx on every call (it grows with
    // the buffer), so a step is only captured once the same encoder length has been decoded for a while — until then the
    // launches go out eagerly (the host then pays ~4 us per launch: equal to the replay for the 40-launch short-cache step,
    // twice the replay's time for the long-cache form)
    if (d.step_seen_T != Tc) { d.step_seen_T = Tc; for (auto & g2 : d.step_graphs) g2.seen = 0; }
    // (the count is shared by the forms: the embedding form runs once per window, the chained form for every other step — counted
    // alone it would be captured, at tens of milliseconds, only after 65 windows)
    ++sg.seen;
    int seen_all = 0; for (const auto & g2 : d.step_graphs) seen_all += g2.seen;
    const bool capture_now = use_graph && !exec && !d.step_capture_failed && seen_all > 64;
    if (capture_now) {
        // first use: run once eagerly (lets the launchers set their function attributes), then capture
        // (not for the chained form: a step's pick kernel advances the device-side record, so the step must not run twice — and its
        // kernels are the plain form's, which has run many times by now)
        if (!chained) {
            enqueue_greedy_step(ctx, Tc, long_kv, false, solo);
            HIP_TRY(hipStreamSynchronize(s));
        }
        if (HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal))) {
            enqueue_greedy_step(ctx, Tc, long_kv, chained, solo);
            hipGraph_t g = nullptr;
            const bool ended = HIP_OK(hipStreamEndCapture(s, &g));
            if (ended && g && HIP_OK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0))) {
                graph = g; graph_T = Tc;
            } else {
                // latched: without this every later step paid an eager step, a sync and a new capture attempt
                WMI_WARN("%s: graph capture failed - staying on eager launches\n", __func__);
                if (g) (void) hipGraphDestroy(g);
                exec = nullptr; d.step_capture_failed = true;
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { hipGraph_t junk = nullptr; (void) hipStreamEndCapture(s, &junk); if (junk) (void) hipGraphDestroy(junk); }
            }
        } else d.step_capture_failed = true;
    }
    hs->seq = d.step_seq = (d.step_seq + 1) & k::SAMPLE_SEQ_MASK;
    // Several greedy transcriptions at once on one device (states of one context, replica contexts, an in-process pool): two or three dependent
    // launch chains overlap well, more of them stretch every launch boundary.  A step therefore takes one of the device's (two) turns from its
    // launch to its sample whenever another device call is in flight; alone, nothing is taken (StepTicket above has the measurements).
    // (WMI_NO_STEP_TICKET=1: off.  The general decode() path — beam search, t > 0 — has host work between its steps: no ticket.)
    static const bool no_ticket = getenv("WMI_NO_STEP_TICKET") != nullptr;
    StepTicket ticket(ctx.device, !no_ticket && !solo, &ctx);
    if (use_graph && exec) {
        HIP_TRY(hipGraphLaunch(exec, s));
    } else {
        enqueue_greedy_step(ctx, Tc, long_kv, chained, solo);
    }
    const k::SampleOut * r = (const k::SampleOut *) d.sample_host;
    int32_t status = 0;
    if (!wait_for_sample(r, d.step_seq, s, &status)) return false;
    if (status & k::SAMPLE_TAG_SLOW) {
        // correct, but the launch's workgroups waited for each other for long: the device is shared with work this process does not
        // count (another process, the embedder's own kernels) — two launches for the next steps, then try again
        d.pair_backoff = 512; ++d.pair_slow_events;
        (void) hipMemsetAsync((uint32_t *) d.mlp_arrive + PAIR_FAULT_WORD, 0, sizeof(uint32_t), s);
    }
    if (status & k::SAMPLE_TAG_FAULT) {
        // a hand-off inside a k_mlp_pair launch did not complete (or its tags were out of step): this step's rows are not to be trusted.
        // The step is run again from the host's record in the two-launch form — embedding launch, every cache cell and the device-side
        // record rewritten — and the state keeps that form from here on.  (W/whisper.cpp:2517-2595: a decode either succeeds or reports.)
        if (!d.pair_off) WMI_WARN("%s: in-launch hand-off of the MLP failed (status %#x) - step re-run, staying on the two-launch form\n", __func__, (unsigned) status);
        d.pair_off = true; ++d.pair_fallbacks;
        if (!HIP_OK(hipMemsetAsync((uint32_t *) d.mlp_arrive + PAIR_FAULT_WORD, 0, sizeof(uint32_t), s))) return false;
        hs->seq = d.step_seq = (d.step_seq + 1) & k::SAMPLE_SEQ_MASK;
        enqueue_greedy_step(ctx, Tc, long_kv, false, false);
        status = 0;
        if (!wait_for_sample(r, d.step_seq, s, &status) || (status & k::SAMPLE_TAG_FAULT)) { WMI_ERR("%s: the step's re-run failed\n", __func__); return false; }
    }
    ticket.release();
    // what the pick kernel has left on the device for the next step
    d.chain_valid = !ctx.model.quantised && pos + 1 < hp.n_text_ctx && hp.n_text_state <= 1536;      // (k_filter_pick prepares rows of <= 3 x 512 columns)
    d.chain_token = r->id; d.chain_pos = pos + 1; d.chain_head = (int32_t) kv.head + 1;
    if (k::knobs().debug_sync) fprintf(stderr, "[wmi] step token=%d pos=%d n_kv=%d head=%d flags=%d floor=%d init=%d -> id=%d tid=%d p=%g plog=%g pt=%g ptsum=%g\n",
        token, pos, hs->n_kv, hs->kv_head, hs->flags, hs->ts_floor_end, hs->ts_initial_start, r->id, r->tid, r->p, r->plog, r->pt, r->ptsum);
    out = whisper_token_data{ r->id, r->tid, r->p, r->plog, r->pt, r->ptsum, -1, -1, 0.0f };
    {
        int64_t dt = time_us() - t0;
        const int64_t done = phase_settle(st, false);       // the stream is idle: the step's sample has arrived
        if (done > t0) dt = std::max<int64_t>(0, dt - (done - t0));
        st.t_decode_us += dt; st.n_decode++; st.n_sample++;
    }
    return true;
}

// probe: the kernels of the last greedy step `iters` times back to back with no host round trip in between
// (graph replays when the step is captured, eager launches otherwise); microseconds per step on the GPU, -1 without a step
double bench_greedy_step_chain(whisper_context & ctx, int iters) {
    State & st = *ctx.state; DeviceState & d = st.dev;
    if (!d.step_dev || iters <= 0) return -1.0;
    d.chain_valid = false;                                  // the replays below advance the device-side step record
    const char * mask_env = getenv("WMI_STEP_MASK");
    g_step_mask = mask_env ? (unsigned) strtoul(mask_env, nullptr, 0) : ~0u;
    // bits 9 / 10 (only with an explicit mask): skip scores / P.V inside bit 3 — without this guard the eager chain ran without cross-attention
    k::set_xattn_probe_skip(mask_env ? ((g_step_mask >> 9) & 1) | (((g_step_mask >> 10) & 1) << 1) : 0);
    struct Restore { ~Restore() { g_step_mask = ~0u; k::set_xattn_probe_skip(0); } } restore;
    const int Tc = st.enc_n_ctx > 0 ? st.enc_n_ctx : ctx.model.hp.n_audio_ctx;
    hipStream_t s = d.stream;
    hipEvent_t e0, e1;
    if (!HIP_OK(hipEventCreate(&e0)) || !HIP_OK(hipEventCreate(&e1))) return -1.0;
    const bool long_kv = ((const k::DecStep *) d.step_host)->n_kv > 64 && !ctx.model.quantised;
    // The chain is ALWAYS replayed from a captured graph, also for a masked subset of the step's kernels: eager launches are
    // paced by the host (2.7 us per trivial launch on this stack against 1.63 us for the same chain replayed from a graph,
    // scratch/lab/chain_lab.hip), which is what round 2's per-kind table had measured for every kernel under ~4 us.
    // WMI_CHAIN_EAGER=1 keeps the host-paced form for comparison.
    static const bool eager = getenv("WMI_CHAIN_EAGER") != nullptr;
    hipGraph_t pg = nullptr; hipGraphExec_t pexec = nullptr;
    enqueue_greedy_step(ctx, Tc, long_kv);                      // function attributes, lazy allocations: outside the capture
    (void) hipStreamSynchronize(s);
    // WMI_CHAIN_REPS = r: r copies of the (masked) step inside ONE graph — a replay of a handful of kernels is bounded by the
    // replay's own fixed cost, not by the kernels (a per-kind chain of 6 launches is such a graph)
    const int reps = getenv("WMI_CHAIN_REPS") ? std::max(1, atoi(getenv("WMI_CHAIN_REPS"))) : 1;
    if (!eager && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        for (int r = 0; r < reps; ++r) enqueue_greedy_step(ctx, Tc, long_kv);
        if (hipStreamEndCapture(s, &pg) != hipSuccess || !pg || hipGraphInstantiate(&pexec, pg, nullptr, nullptr, 0) != hipSuccess) pexec = nullptr;
    }
    auto once = [&]() { if (pexec) (void) hipGraphLaunch(pexec, s); else enqueue_greedy_step(ctx, Tc, long_kv); };
    if (reps > 1) iters = (iters + reps - 1) / reps;
    for (int i = 0; i < 4; ++i) once();
    (void) hipStreamSynchronize(s);
    (void) hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) once();
    (void) hipEventRecord(e1, s);
    (void) hipEventSynchronize(e1);
    float ms = 0.0f; (void) hipEventElapsedTime(&ms, e0, e1);
    if (pexec) (void) hipGraphExecDestroy(pexec);
    if (pg) (void) hipGraphDestroy(pg);
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    return (double) ms * 1000.0 / (iters * (pexec ? reps : 1));
}

// reduces the stamp records of n launches (kernels.h: Stamp) to out[6 i + 0..5] — see step_stamps
static int stamps_reduce(whisper_context & ctx, const unsigned long long * buf, int n, double * out, int cap) {
    int ret = -1;
    {
        std::vector<unsigned long long> h((size_t) n * k::STAMP_WAVES * 4);
        if (HIP_OK(hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost))) {
            int khz = 100000; (void) hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx.device);
            const double tick_us = 1000.0 / (double) (khz > 0 ? khz : 100000);
            unsigned long long origin = ~0ull;
            for (size_t i = 0; i < h.size(); i += 4) if (h[i] && h[i] < origin) origin = h[i];
            ret = std::min(n, cap);
            for (int l = 0; l < ret; ++l) {
                unsigned long long mn = ~0ull, mxs = 0, mxe = 0, m1 = 0, m2 = 0; int cnt = 0;
                for (int w2 = 0; w2 < k::STAMP_WAVES; ++w2) {
                    const unsigned long long * r = &h[((size_t) l * k::STAMP_WAVES + w2) * 4];
                    if (!r[0]) continue;
                    mn = std::min(mn, r[0]); mxs = std::max(mxs, r[0]); mxe = std::max(mxe, r[1]); m1 = std::max(m1, r[2]); m2 = std::max(m2, r[3]); ++cnt;
                }
                out[6 * l + 0] = cnt ? (double) (mn - origin) * tick_us : -1.0;
                out[6 * l + 1] = cnt ? (double) (mxs - origin) * tick_us : -1.0;
                out[6 * l + 2] = cnt ? (double) (mxe - origin) * tick_us : -1.0;
                out[6 * l + 3] = (double) cnt;
                if (m1 >> 63) {
                    // not wall-clock mid points but the SHADER clock counter at start / end (k_xattn_fused): effective MHz, averaged
                    double mhz = 0.0; int nm = 0;
                    for (int w2 = 0; w2 < k::STAMP_WAVES; ++w2) {
                        const unsigned long long * r = &h[((size_t) l * k::STAMP_WAVES + w2) * 4];
                        const unsigned long long c0 = r[2] & ~(1ull << 63);
                        if (r[0] && r[1] > r[0] && r[3] > c0) { mhz += (double) (r[3] - c0) / ((double) (r[1] - r[0]) * tick_us); ++nm; }
                    }
                    out[6 * l + 4] = -2.0; out[6 * l + 5] = nm ? mhz / nm : -1.0;
                } else {
                out[6 * l + 4] = m1 ? (double) (m1 - origin) * tick_us : -1.0;      // optional mid points (last wavefront to reach them)
                out[6 * l + 5] = m2 ? (double) (m2 - origin) * tick_us : -1.0;
                }
            }
        }
        }
    return ret;
}

// probe: body / boundary split of the greedy step's dependent launches from in-kernel time stamps (kernels.h: Stamp).  The step is
// captured with stamping on, replayed a few times, and the LAST replay's records are reduced per launch:
// out[6 i + 0..5] = first wavefront start, last wavefront start, last wavefront end (microseconds from the step's first start),
// number of wavefront records, and two optional mid points (k_gemv1: activation row ready, first row tile reduced; -1 if absent).  Returns the number of launches (<= cap), -1 when there is no step to replay.
int step_stamps(whisper_context & ctx, double * out, int cap, bool chained) {
    State & st = *ctx.state; DeviceState & d = st.dev;
    if (!d.step_dev || cap <= 0) return -1;
    if (ctx.model.quantised) chained = false;               // (the block-quantised step has one form)
    // the chained replays advance the device-side record (cache head / n_kv + 1 per replay: four below): refuse near the end of the cache
    if (chained && (int) ((const k::DecStep *) d.step_host)->kv_head + 6 >= (int) st.kv_self.size) return -1;
    const int Tc = st.enc_n_ctx > 0 ? st.enc_n_ctx : ctx.model.hp.n_audio_ctx;
    hipStream_t s = d.stream;
    const bool long_kv = ((const k::DecStep *) d.step_host)->n_kv > 64;
    constexpr int MAXL = 512;
    const size_t bytes = (size_t) MAXL * k::STAMP_WAVES * 4 * sizeof(unsigned long long);
    unsigned long long * buf = nullptr;
    if (!HIP_OK(hipMalloc((void **) &buf, bytes))) return -1;
    (void) hipMemsetAsync(buf, 0, bytes, s);
    d.chain_valid = false;
    enqueue_greedy_step(ctx, Tc, long_kv, false);           // leaves a valid device-side record for the chained form
    (void) hipStreamSynchronize(s);
    hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr; int n = 0;
    k::stamp_enable(buf);
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        enqueue_greedy_step(ctx, Tc, long_kv, chained);
        n = k::stamp_count();
        if (hipStreamEndCapture(s, &g) != hipSuccess || !g || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) ex = nullptr;
    }
    k::stamp_enable(nullptr);
    int ret = -1;
    if (ex && n > 0 && n <= MAXL) {
        for (int i = 0; i < 3; ++i) (void) hipGraphLaunch(ex, s);
        (void) hipStreamSynchronize(s);
        ret = stamps_reduce(ctx, buf, n, out, cap);
    }
    if (ex) (void) hipGraphExecDestroy(ex);
    if (g) (void) hipGraphDestroy(g);
    d.chain_valid = false;                                  // the replays advanced (chained) or reset the device-side record
    (void) hipFree(buf);
    return ret;
}

} // namespace wmi
