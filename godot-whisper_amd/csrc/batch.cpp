// Lock-step transcription of several independent 30 s chunks on ONE GPU (BASELINE config 4:
// 8 chunks per GPU).  Precedent in the reference: whisper_full_parallel (W/whisper.cpp:5809-5935) — shared
// read-only weights, one whisper_state per worker.  Here the workers are not threads but rows of the same
// kernels:
//   * encoder: the chunks are stacked along M, so every projection / MLP GEMM runs once with M = B*T rows
//     (the weights are read once per batch instead of once per chunk, the MFMA tiles fill the chip); attention
//     gets a chunk dimension in grid.z; the conv front-end stays per chunk (overlapping-row implicit GEMM);
//   * decoder: one greedy step advances every chunk by one token: the weight-streaming GEMV carries one
//     activation row per chunk (R = B <= 8), each row with its own self-attention cache, cross cache slice and
//     step record (token, position, filter flags); filters + arg-max run per row on the device.
// The control flow per chunk is the one of full() (seek windows, prompt, state machine, segment emission);
// chunks whose window would need the temperature fallback are re-run alone through full() afterwards, so the
// result of every chunk equals what a separate whisper_full() call returns for it.

#include "wmi.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <mutex>

namespace wmi {

namespace {

constexpr int MAX_LANES = 16;              // columns of k_rows_mfma (the exact mode stays within k_gemv's 8 rows)

template <typename T> bool dalloc(T *& p, size_t n_elems) {
    p = nullptr;
    return HIP_OK(hipMalloc((void **) &p, std::max<size_t>(n_elems, 1) * sizeof(T)));
}
template <typename T> void dfree(T *& p) { if (p) (void) hipFree(p); p = nullptr; }

struct StateSwap {                          // run per-chunk host logic (mel, envelope, emission) against a lane's state (this thread only: wmi.h StateSlot)
    StateInstall inst;
    StateSwap(whisper_context & c, State * lane) : inst(c.state, lane) {}
};

State * new_lane_state(whisper_context & ctx) {
    State * st = new State();
    st->dev.stream = ctx.state->dev.stream;                 // one stream: the lanes are rows of the same launches
    if (!dalloc(st->dev.mel_max, 4)) { delete st; return nullptr; }
    // the chunks' log-mel kernels are latency-bound chains (one workgroup per frame): on their own streams they overlap
    if (!HIP_OK(hipStreamCreateWithFlags(&st->dev.mel_stream, hipStreamNonBlocking)) ||
        !HIP_OK(hipEventCreateWithFlags(&st->dev.mel_ev, hipEventDisableTiming))) { delete st; return nullptr; }
    for (auto & dec : st->decoders) dec.rng = std::mt19937(0);
    return st;
}

void free_lane_state(State * st) {
    DeviceState & d = st->dev;
    if (d.copy_stream) { (void) hipStreamSynchronize(d.copy_stream); (void) hipStreamDestroy(d.copy_stream); }
    if (d.energy_ev) (void) hipEventDestroy(d.energy_ev);
    if (d.mel_stream) { (void) hipStreamSynchronize(d.mel_stream); (void) hipStreamDestroy(d.mel_stream); }
    if (d.mel_ev) (void) hipEventDestroy(d.mel_ev);
    dfree(d.pcm); dfree(d.mel); dfree(d.mel_max); dfree(d.energy);
    if (d.energy_host) (void) hipHostFree(d.energy_host);
    if (d.ts_host) (void) hipHostFree(d.ts_host);
    delete st;
}

bool ensure_batch(whisper_context & ctx, int B) {
    if (!ctx.batch) ctx.batch = new BatchWork();
    BatchWork & w = *ctx.batch;
    if (w.B >= B) return true;
    // grow: drop the old arenas (never on the hot path: the first call decides the size)
    hipStream_t s = ctx.state->dev.stream;
    (void) hipStreamSynchronize(s);
    dfree(w.mel_t); dfree(w.conv1); dfree(w.x); dfree(w.xn); dfree(w.q); dfree(w.k); dfree(w.att); dfree(w.vt); dfree(w.h);
    dfree(w.enc_out_h); dfree(w.kvc_k); dfree(w.kvc_v); dfree(w.self_k); dfree(w.self_v); dfree(w.dx); dfree(w.dq); dfree(w.datt); dfree(w.dh);
    dfree(w.logits); dfree(w.xattn); dfree(w.aq); dfree(w.ads); dfree(w.aq16); dfree(w.wq16); dfree(w.att32); dfree(w.datt32);
    for (auto & rg : w.rows_graph) {                                               // the captured steps hold the old pointers
        if (rg.exec) (void) hipGraphExecDestroy(rg.exec);
        if (rg.graph) (void) hipGraphDestroy(rg.graph);
        rg = BatchWork::RowsGraph{};
    }
    w.chain_valid = false;
    if (w.front_hand) (void) hipFree(w.front_hand);
    if (w.step_dev) (void) hipFree(w.step_dev);
    if (w.sample_dev) (void) hipFree(w.sample_dev);
    if (w.filter_scratch) (void) hipFree(w.filter_scratch);
    if (w.step_host) (void) hipHostFree(w.step_host);
    if (w.sample_host) (void) hipHostFree(w.sample_host);
    w.step_dev = w.sample_dev = w.filter_scratch = w.step_host = w.sample_host = nullptr;

    const HParams & hp = ctx.model.hp;
    const size_t S = hp.n_audio_state, T = hp.n_audio_ctx, Lt = hp.n_text_layer, H = hp.n_audio_head, n_ctx = hp.n_text_ctx;
    w.Tpad = (int) ((T + 63) / 64 * 64);
    w.mel_rows = 2 * T + 8;
    const size_t nb = (size_t) B;
    // LN1's output and q / k keep every chunk on a 16-row boundary (encode_rows: the q|k|v GEMM's V^T epilogue): T rounded up to 16 rows per chunk
    const size_t TP = (T + 15) & ~(size_t) 15;
    bool ok = dalloc(w.mel_t, nb * w.mel_rows * hp.n_mels + 1024) && dalloc(w.conv1, nb * (2 * T + 8) * S + 4 * S)
           && dalloc(w.x, nb * T * S) && dalloc(w.xn, nb * TP * S) && dalloc(w.q, nb * TP * S) && dalloc(w.k, nb * TP * S)
           && dalloc(w.att, nb * T * S) && dalloc(w.vt, nb * S * w.Tpad) && dalloc(w.h, nb * T * 4 * S) && dalloc(w.enc_out_h, nb * T * S)
           && dalloc(w.kvc_k, Lt * nb * T * S) && dalloc(w.kvc_v, Lt * nb * T * S)
           && dalloc(w.self_k, nb * Lt * n_ctx * S) && dalloc(w.self_v, nb * Lt * n_ctx * S)
           && dalloc(w.dx, nb * S) && dalloc(w.dq, nb * S) && dalloc(w.datt, nb * S) && dalloc(w.dh, nb * 4 * S) && dalloc(w.logits, nb * hp.n_vocab)
           && dalloc(w.xattn, k::attn_cross_scratch_floats(B, (int) H, (int) T));
    if (ctx.model.quantised) {                               // q8 activation rows of all chunks + f32 attention outputs (device_q.cpp)
        w.aq_rows = (int) (nb * T);
        ok = ok && dalloc(w.aq, nb * T * 4 * S) && dalloc(w.ads, 2 * nb * T * (4 * S / 32)) && dalloc(w.att32, nb * T * S) && dalloc(w.datt32, nb * S);
        w.wq16_elems = 8 * S * S;
        ok = ok && dalloc(w.aq16, nb * T * 4 * S) && dalloc(w.wq16, w.wq16_elems);
    }
    if (ok && !ctx.model.quantised && S <= 1024) {
        const size_t bytes = nb * 16 * S + 64 + nb * (size_t) k::XBACK_ROW_GRANULES * 8;      // + the rows' cross-attention back (k::xback)
        if (HIP_OK(hipMalloc(&w.front_hand, bytes))) k::fill_zero(w.front_hand, bytes, ctx.state->dev.stream); else w.front_hand = nullptr;
    }
    ok = ok && HIP_OK(hipMalloc(&w.step_dev, nb * sizeof(k::DecStep))) && HIP_OK(hipMalloc(&w.sample_dev, nb * sizeof(k::SampleOut)))
            && HIP_OK(hipMalloc(&w.filter_scratch, k::filter_scratch_bytes(B)))
            && HIP_OK(hipHostMalloc(&w.step_host, nb * sizeof(k::DecStep), hipHostMallocDefault))
            && HIP_OK(hipHostMalloc(&w.sample_host, nb * sizeof(k::SampleOut), hipHostMallocDefault));
    if (!ok) { WMI_ERR("%s: device allocation failed (B = %d)\n", __func__, B); w.B = 0; return false; }
    memset(w.sample_host, 0, nb * sizeof(k::SampleOut));
    w.step_seq = 0;
    k::fill_zero(w.vt, nb * S * w.Tpad * sizeof(__half), s);
    k::fill_zero(w.xn, nb * TP * S * sizeof(__half), s);            // (the chunks' padding rows are GEMM operands: finite)
    k::fill_zero(w.conv1, (nb * (2 * T + 8) * S + 4 * S) * sizeof(__half), s);
    k::fill_zero(w.mel_t, (nb * w.mel_rows * hp.n_mels + 1024) * sizeof(__half), s);
    k::fill_zero(w.self_k, nb * Lt * n_ctx * S * sizeof(__half), s);
    k::fill_zero(w.self_v, nb * Lt * n_ctx * S * sizeof(__half), s);
    k::fill_zero(w.kvc_k, Lt * nb * T * S * sizeof(__half), s);
    k::fill_zero(w.kvc_v, Lt * nb * T * S * sizeof(__half), s);
    HIP_TRY(hipStreamSynchronize(s));
    if (w.lanes.empty()) w.lanes.push_back(ctx.state);
    while ((int) w.lanes.size() < B) {
        State * st = new_lane_state(ctx);
        if (!st) return false;
        w.lanes.push_back(st);
    }
    w.B = B;
    return true;
}

} // namespace

// ---------------------------------------------------------------------------------------------- batched encoder
// rows[r] = lane whose mel feeds chunk row r; seek[r] = its mel frame offset  (declared in wmi.h: the in-situ GEMM probe replays it)
bool encode_rows(whisper_context & ctx, const std::vector<int> & rows, const std::vector<int> & seek, int audio_ctx) {
    BatchWork & b = *ctx.batch; const Weights & w = ctx.w; const HParams & hp = ctx.model.hp;
    const int64_t t0 = time_us();
    const int nb = (int) rows.size();
    const int T = audio_ctx > 0 ? audio_ctx : hp.n_audio_ctx;
    const int S = hp.n_audio_state, H = hp.n_audio_head, La = hp.n_audio_layer, Lt = hp.n_text_layer, nm = hp.n_mels;
    hipStream_t s = ctx.state->dev.stream;
    const int M = nb * T;

    // conv front-end.  The overlapping-row implicit GEMM needs each chunk's zero guard rows, so the chunks are STACKED with a period of
    // R = 2 T + 8 rows in both images — mel slice rows c R .. (row 0 and the rows behind the 2 T frames are zero) and conv1 rows c R ..
    // (row c R = the chunk's leading guard) — and each convolution is ONE launch over all of them: conv1 row m + 1 from mel rows m .. m + 2,
    // conv2 output t of chunk c from conv1 rows c R + 2 t .. + 2 (EPI_CONV2 with rows_per_chunk = R / 2 stores rows t < T at c T + t).
    // The rows conv1 computes across a chunk boundary (c R + 2 T + 1 .. (c + 1) R) are zeroed again before conv2 reads them.
    // Per element the arithmetic is the one-chunk launch's (same operands, same k order).  8 chunks: 32 launches -> 4.
    // WMI_CONV_PER_CHUNK=1 (A/B): one chunk at a time as before.
    static const bool conv_per_chunk = getenv("WMI_CONV_PER_CHUNK") != nullptr;
    const int rows_mel = 2 * T + 6;
    for (int r = 0; r < nb; ++r) {
        State & ls = *b.lanes[rows[r]];
        if (ls.mel.n_mel != nm || ls.dev.mel == nullptr) { WMI_ERR("%s: chunk row %d has no mel spectrogram\n", __func__, r); return false; }
    }
    if (!conv_per_chunk && nb >= 2 && nb <= 16) {
        const int R = 2 * T + 8;
        k::MelSliceBatch mb{};
        for (int r = 0; r < nb; ++r) { State & ls = *b.lanes[rows[r]]; mb.mel[r] = ls.dev.mel; mb.n_len[r] = ls.mel.n_len; mb.offset[r] = seek[r]; }
        k::mel_slice_batch(mb, nb, nm, 2 * T, b.mel_t, nm, R, s);
        {
            k::GemmArgs a{};
            a.A = b.mel_t; a.lda = nm; a.W = w.conv1_w; a.ldw = w.conv1_k; a.M = nb * R - 1; a.N = S; a.K = w.conv1_k;
            a.bias = w.conv1_b; a.C = b.conv1 + S; a.ldc = S;
            k::gemm(k::EPI_F16_BIAS_GELU, a, s);
        }
        k::fill_zero_strided(b.conv1 + (size_t) (2 * T + 1) * S, (size_t) 8 * S * sizeof(__half), (size_t) R * S * sizeof(__half), nb, s);
        {
            k::GemmArgs a{};
            a.A = b.conv1; a.lda = 2 * S; a.W = w.conv2_w; a.ldw = w.conv2_k; a.M = nb * (R / 2) - 1; a.N = S; a.K = w.conv2_k;
            a.bias = w.conv2_b; a.C = b.x; a.ldc = S; a.resid = w.e_pe; a.ldr = S; a.rows_per_chunk = R / 2;
            k::gemm(k::EPI_CONV2, a, s);
        }
    } else
    for (int r = 0; r < nb; ++r) {
        State & ls = *b.lanes[rows[r]];
        __half * mel_t = b.mel_t + (size_t) r * b.mel_rows * nm;
        __half * conv1 = b.conv1 + (size_t) r * (2 * hp.n_audio_ctx + 8) * S;
        k::mel_slice(ls.dev.mel, ls.mel.n_len, nm, seek[r], 2 * T, mel_t, nm, rows_mel, s);
        {
            k::GemmArgs a{};
            a.A = mel_t; a.lda = nm; a.W = w.conv1_w; a.ldw = w.conv1_k; a.M = 2 * T; a.N = S; a.K = w.conv1_k;
            a.bias = w.conv1_b; a.C = conv1 + S; a.ldc = S;
            k::gemm(k::EPI_F16_BIAS_GELU, a, s);
        }
        k::fill_zero(conv1 + (size_t) (2 * T + 1) * S, (size_t) S * sizeof(__half), s);
        {
            k::GemmArgs a{};
            a.A = conv1; a.lda = 2 * S; a.W = w.conv2_w; a.ldw = w.conv2_k; a.M = T; a.N = S; a.K = w.conv2_k;
            a.bias = w.conv2_b; a.C = b.x + (size_t) r * T * S; a.ldc = S; a.resid = w.e_pe; a.ldr = S;
            k::gemm(k::EPI_CONV2, a, s);
        }
    }
    if (ctx.model.quantised) {                               // block-quantised weights: the q8 layer loop over the stacked chunks
        EncBufsQ e{};
        e.T = T; e.nb = nb; e.Tpad = b.Tpad; e.x = b.x; e.q = b.q; e.k = b.k; e.vt = b.vt; e.h = b.h; e.att32 = b.att32;
        e.enc_out = nullptr; e.enc_out_h = b.enc_out_h; e.kvc_k = b.kvc_k; e.kvc_v = b.kvc_v;
        e.A = q8_rows(b, S); e.A4 = q8_rows(b, 4 * S);
        if (!encode_layers_q_on(ctx, e, s)) return false;
        HIP_TRY(hipStreamSynchronize(s));
        if (!HIP_OK(hipGetLastError())) return false;
        b.enc_rows = nb; b.enc_T = T;
        b.t_encode_us += time_us() - t0;
        return true;
    }
    const float kq_scale = 1.0f / sqrtf((float) S / H);
    // Rows per chunk of LN1's output, q and k: T rounded up to 16 (1500 -> 1504), so that every 16-row MFMA fragment of the q|k|v GEMM lies
    // inside one chunk at a 16-step offset — its V^T third then leaves in whole 128-byte lines (gemm_epi.h: epilogue_vt_wide) instead of 32-byte
    // pieces (what the launch cost over a plain epilogue: 41 against 29.5 us at 8 chunks).  The padding rows are zero in xn (never written),
    // finite junk in q / k / V^T (never read as queries or keys: the attention stops at T).  WMI_ENC_NO_ROWPAD=1: off (A/B).
    static const bool no_rowpad = getenv("WMI_ENC_NO_ROWPAD") != nullptr;
    const int TP = (nb >= 2 && !no_rowpad && (T & 15) != 0 && ((T + 15) & ~15) <= b.Tpad) ? (T + 15) & ~15 : T;
    const int MP = nb * TP;
    b.qk_rows = TP;
    for (int il = 0; il < La; ++il) {
        const EncLayerW & l = w.enc[il];
        if (TP != T) k::layernorm(b.x, M, S, l.ln1_g, l.ln1_b, hp.eps, b.xn, nullptr, s, T, TP);
        else         k::layernorm(b.x, M, S, l.ln1_g, l.ln1_b, hp.eps, b.xn, nullptr, s);
        {
            k::GemmArgs a{};
            a.A = b.xn; a.lda = S; a.W = l.w_qkv; a.ldw = S; a.M = MP; a.N = 3 * S; a.K = S; a.bias = l.b_qkv;
            a.C = b.q; a.ldc = S; a.aux = b.k; a.ldaux = S; a.aux2 = b.vt; a.ldaux2 = b.Tpad; a.S = S;
            a.rows_per_chunk = TP; a.chunk_stride_aux2 = (int64_t) S * b.Tpad;
            k::gemm(k::EPI_QKV_ENC, a, s);
        }
        k::attn_encoder(b.q, b.k, b.vt, T, b.Tpad, S, H, kq_scale, b.att, s, nb, nullptr, TP);
        {
            k::GemmArgs a{};
            a.A = b.att; a.lda = S; a.W = l.w_o; a.ldw = S; a.M = M; a.N = S; a.K = S; a.bias = l.b_o;
            a.C = b.x; a.ldc = S; a.resid = b.x; a.ldr = S;
            k::gemm(k::EPI_F32_BIAS_RESID, a, s);
        }
        k::layernorm(b.x, M, S, l.ln2_g, l.ln2_b, hp.eps, b.xn, nullptr, s);
        {
            k::GemmArgs a{};
            a.A = b.xn; a.lda = S; a.W = l.w_fc1; a.ldw = S; a.M = M; a.N = 4 * S; a.K = S; a.bias = l.b_fc1;
            a.C = b.h; a.ldc = 4 * S;
            k::gemm(k::EPI_F16_BIAS_GELU, a, s);
        }
        {
            k::GemmArgs a{};
            a.A = b.h; a.lda = 4 * S; a.W = l.w_fc2; a.ldw = 4 * S; a.M = M; a.N = S; a.K = 4 * S; a.bias = l.b_fc2;
            a.C = b.x; a.ldc = S; a.resid = b.x; a.ldr = S;
            k::gemm(k::EPI_F32_BIAS_RESID, a, s);
        }
    }
    k::layernorm(b.x, M, S, w.e_ln_g, w.e_ln_b, hp.eps, b.enc_out_h, nullptr, s);
    {   // cross K/V of every decoder layer and every chunk: [L][nb*T][S] — chunk r of layer il starts at (il*nb + r)*T*S
        k::GemmArgs a{};
        a.A = b.enc_out_h; a.lda = S; a.W = w.w_ckv; a.ldw = S; a.M = M; a.N = Lt * 2 * S; a.K = S; a.bias = w.b_ckv;
        a.C = b.kvc_k; a.ldc = S; a.aux = b.kvc_v; a.ldaux = S; a.S = S; a.layer_stride = (int64_t) M * S;
        a.scale = powf((float) S / H, -0.25f);
        k::gemm(k::EPI_CROSS_KV, a, s);
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (!HIP_OK(hipGetLastError())) return false;
    b.enc_rows = nb; b.enc_T = T;
    b.t_encode_us += time_us() - t0;
    return true;
}

namespace {

// ---------------------------------------------------------------------------------------------- batched greedy step
// step records are already in b.step_host[0..nb); results land in b.sample_host[0..nb)
// probe only (bench kernel 21): which kernel kinds of the lock-step step are enqueued — bit 0 embed, 1 qkv, 2 self-attention rows,
// 3 out, 4 cross scores + P.V, 5 combine, 6 cross out, 7 mlp.0, 8 mlp.2, 9 logits, 10 filters
static unsigned g_rows_mask = ~0u;

// chained: the step starts from what the previous step's pick kernel left on the device (see BatchWork::chain_*): no embedding
// launch; the rest of the host's step records reaches the filters through extra workgroups of the last layer's self-attention launch
// may this step run the front of its layers (LN + q|k|v, self-attention, out projection) as one launch per layer (k::front, rows on grid.y)?
// Every row's cache <= 64 cells, an even layer count (the launches' tags alternate), no other transcription on the device (the launch's
// workgroups wait for each other: all of them must be resident), no hand-off failure so far, not backing off after a slow one.
static bool rows_fronted(whisper_context & ctx, int nb) {
    BatchWork & b = *ctx.batch; const HParams & hp = ctx.model.hp;
    if (ctx.model.quantised || !b.front_hand || b.front_off || b.front_backoff > 0 || (hp.n_text_layer & 1) || k::knobs().no_front) return false;
    if (busy_transcriptions(ctx.device) > 1 || !k::front_usable(hp.n_text_state, nb)) return false;
    const k::DecStep * hs = (const k::DecStep *) b.step_host;
    for (int r = 0; r < nb; ++r) if (hs[r].n_kv > 64) return false;
    return true;
}

static void enqueue_rows_step(whisper_context & ctx, int nb, bool chained = false, int fronted_in = -1) {
    if (ctx.model.quantised) { enqueue_rows_step_q(ctx, nb); return; }
    BatchWork & b = *ctx.batch; const Weights & w = ctx.w; const HParams & hp = ctx.model.hp;
    const unsigned M = g_rows_mask;
    const int S = hp.n_text_state, H = hp.n_text_head, Lt = hp.n_text_layer, NV = hp.n_vocab, n_ctx = hp.n_text_ctx;
    const int Tc = b.enc_T;
    hipStream_t s = ctx.state->dev.stream;
    const k::DecStep * stp = (const k::DecStep *) b.step_dev;
    const float kq_scale = powf((float) S / H, -0.25f);
    const int step_stride = (int) (sizeof(k::DecStep) / sizeof(int32_t));
    const int64_t cache_stride = (int64_t) Lt * n_ctx * S;             // between the chunks' self caches
    const int64_t cross_layer = (int64_t) b.enc_rows * Tc * S;

    // where the chained form mirrors the rest of the host's records: beside the vocabulary projection when that launch is the
    // matrix-core rows kernel, else beside the last layer's self-attention
    k::GemvArgs glog{};
    {
        glog.n = nb; glog.K = S; glog.N = NV; glog.W = w.d_te; glog.epi = k::EPI_LOGITS; glog.C = b.logits; glog.ldc = NV; glog.ldr = S; glog.S = S;
        glog.eps = hp.eps; glog.lanes = 1; glog.step_stride = step_stride; glog.cache_row_stride = cache_stride;
        glog.x32 = b.dx; glog.ln_g = w.d_ln_g; glog.ln_b = w.d_ln_b;
    }
    const bool mirror_in_logits = chained && (M & 512) && k::gemv_rows_carries_mirror(glog);
    // the front of the layers as one launch per layer (rows_fronted; the caller decides when its graphs depend on it) — not when the
    // step-record mirror has to ride in the last layer's self-attention launch, not for a probe of single kernel kinds
    const bool fronted = (fronted_in < 0 ? rows_fronted(ctx, nb) : fronted_in != 0) && (M & 2) && (M & 4) && (M & 8) && (!chained || mirror_in_logits);
    // ... and the back of the cross-attention the same way where its launch fits (k::xback_usable: 4 - 8 key slices, S <= 512, residency)
    const bool backed = fronted && (M & 16) && (M & 64) && !k::knobs().no_xback && k::xback_usable(S, H, Tc, nb);
    if ((M & 1) && !chained) k::dec_embed_step((const k::DecStep *) b.step_host, (k::DecStep *) b.step_dev, S, w.d_te, w.d_pe, b.dx, s, nb);
    auto base = [&](int K, int N, const __half * W, const float * bias, int epi, void * C, int ldc) {
        k::GemvArgs g{};
        g.n = nb; g.K = K; g.N = N; g.W = W; g.bias = bias; g.epi = epi; g.C = C; g.ldc = ldc; g.ldr = S; g.S = S; g.eps = hp.eps;
        g.lanes = 1; g.step_stride = step_stride; g.cache_row_stride = cache_stride;
        return g;
    };
    for (int il = 0; il < Lt; ++il) {
        const DecLayerW & l = w.dec[il];
        __half * ck = b.self_k + (size_t) il * n_ctx * S, * cv = b.self_v + (size_t) il * n_ctx * S;     // chunk 0; + r * cache_stride
        if (fronted) {   // LN1 + q|k|v, the self-attention (once per head) and the out projection of every row as ONE launch (k::front, rows on grid.y)
            k::FrontArgs f{};
            f.x = b.dx; f.xout = b.dx; f.ln_g = l.ln1_g; f.ln_b = l.ln1_b; f.eps = hp.eps; f.S = S; f.Wqkv = l.w_qkv; f.bqkv = l.b_qkv; f.scale = kq_scale;
            f.q16 = b.dq; f.ck = ck; f.cv = cv; f.kv_head = &stp->kv_head; f.n_kv = &stp->n_kv; f.cap = n_ctx; f.Wo = l.w_o; f.bo = l.b_o;
            f.gq = (unsigned long long *) b.front_hand; f.ga = f.gq + 3 * S / 2;
            uint32_t * words = (uint32_t *) ((unsigned char *) b.front_hand + (size_t) b.B * 16 * S);
            f.epoch = words; f.par = il & 1; f.fault = words + 4; f.spin_cap = k::knobs().pair_spin_cap; f.withhold = k::knobs().front_withhold;
            f.cache_row_stride = cache_stride; f.step_stride = step_stride; f.rows = nb;
            k::front(f, s);
        } else {
        {   // LN1 + q | k -> cache | v -> cache
            k::GemvArgs g = base(S, 3 * S, l.w_qkv, l.b_qkv, k::EPI_QKV_DEC, b.dq, S);
            g.x32 = b.dx; g.ln_g = l.ln1_g; g.ln_b = l.ln1_b; g.aux = ck; g.ldaux = S; g.aux2 = cv; g.ldaux2 = S; g.scale = kq_scale;
            g.row_off = &stp->kv_head;
            if (M & 2) k::gemv(g, s);
        }
        {   // self-attention, one workgroup per chunk row (same arithmetic as the single-row fused prologue), then
            // out projection + residual
            const bool mirror = chained && il == Lt - 1 && !mirror_in_logits;
            k::GemvArgs g = base(S, S, l.w_o, l.b_o, k::EPI_F32_BIAS_RESID, b.dx, S);
            g.resid = b.dx;
            // the out projection takes the self-attention in its prologue (the one-row kernel per row, kernels.h) unless this launch
            // has to carry the step-record mirror or the probe asked for the two kinds separately
            k::GemvArgs gs = g;
            gs.sa_q = b.dq; gs.sa_k = ck; gs.sa_v = cv; gs.sa_nkv = &stp->n_kv; gs.sa_cap = n_ctx;
            if (!mirror && (M & 4) && (M & 8) && k::gemv_rows_take_self_attention(gs)) k::gemv(gs, s);
            else {
                if (M & 4) k::self_attn_rows(b.dq, nb, S, ck, cv, cache_stride, &stp->n_kv, step_stride, n_ctx, b.datt, s, nullptr, false,
                                             mirror ? b.step_host : nullptr, mirror ? b.step_dev : nullptr);
                g.a16 = b.datt;
                if (M & 8) k::gemv(g, s);
            }
        }
        }
        {   // LN2 + cross query (folded into the score kernel) + cross-attention partials over each row's own chunk
            const float * po = nullptr, * pl = nullptr, * pm = nullptr; int ns = 0;
            if (!(M & 16)) k::attn_cross_partials_layout(nb, H, Tc, b.xattn, &po, &pl, &pm, &ns);
            else if (S > 1536) {                            // wider than three 512-column chunks: separate projection launch (see device.cpp)
                k::GemvArgs g = base(S, S, l.w_cq, l.b_cq, k::EPI_Q_SCALED, b.dq, S);
                g.x32 = b.dx; g.ln_g = l.ln2_g; g.ln_b = l.ln2_b; g.scale = kq_scale;
                k::gemv(g, s);
                k::attn_cross_split_partials(b.dq, nb, S, H, b.kvc_k + (size_t) il * cross_layer, b.kvc_v + (size_t) il * cross_layer, Tc,
                                             b.xattn, &po, &pl, &pm, &ns, s, (int64_t) Tc * S);
            } else if (backed) {   // LN2 + cross query + key slices, the combine (once per head) and the out projection of every row as ONE launch (k::xback, rows on grid.z)
                k::XbackArgs xb{};
                xb.x = b.dx; xb.xout = b.dx; xb.ln_g = l.ln2_g; xb.ln_b = l.ln2_b; xb.eps = hp.eps; xb.S = S; xb.wq = l.w_cq; xb.bq = l.b_cq; xb.qscale = kq_scale;
                xb.kc = b.kvc_k + (size_t) il * cross_layer; xb.vc = b.kvc_v + (size_t) il * cross_layer; xb.T = Tc; xb.Wo = l.w_co; xb.bo = l.b_co;
                unsigned char * base = (unsigned char *) b.front_hand + (size_t) b.B * 16 * S;
                xb.gp = (unsigned long long *) (base + 64); xb.ga = xb.gp + 8 * 8 * 66;
                xb.epoch = (uint32_t *) base + 6; xb.par = il & 1; xb.fault = (uint32_t *) base + 4;
                xb.spin_cap = k::knobs().pair_spin_cap; xb.withhold = k::knobs().xback_withhold;
                xb.kv_row_stride = (int64_t) Tc * S; xb.rows = nb;
                k::xback(xb, H, b.xattn, s);
                goto mlp_rows;
            } else
            k::attn_cross_qsplit_partials(b.dx, l.ln2_g, l.ln2_b, hp.eps, l.w_cq, l.b_cq, kq_scale, nb, S, H,
                                          b.kvc_k + (size_t) il * cross_layer, b.kvc_v + (size_t) il * cross_layer, Tc,
                                          b.xattn, &po, &pl, &pm, &ns, s, (int64_t) Tc * S);
            // the out projection combines the key-slice partials in its prologue (all row kernels do): no combine launch
            k::GemvArgs g = base(S, S, l.w_co, l.b_co, k::EPI_F32_BIAS_RESID, b.dx, S);
            g.comb_o = po; g.comb_l = pl; g.comb_m = pm; g.comb_ns = ns; g.resid = b.dx;
            if (M & 64) k::gemv(g, s);
        }
        mlp_rows:
        {
            k::GemvArgs g = base(S, 4 * S, l.w_fc1, l.b_fc1, k::EPI_F16_BIAS_GELU, b.dh, 4 * S);
            g.x32 = b.dx; g.ln_g = l.ln3_g; g.ln_b = l.ln3_b;
            if (M & 128) k::gemv(g, s);
        }
        {
            k::GemvArgs g = base(4 * S, S, l.w_fc2, l.b_fc2, k::EPI_F32_BIAS_RESID, b.dx, S);
            g.a16 = b.dh; g.resid = b.dx;
            if (M & 256) k::gemv(g, s);
        }
    }
    {
        k::GemvArgs g = glog;
        if (mirror_in_logits) { g.rows_mirror_src = b.step_host; g.rows_mirror_dst = b.step_dev; }
        if (M & 512) k::gemv(g, s);
    }
    // the picks also prepare the next step on the device (row r: token = pick, position / cache head + 1, x[r] = te[pick] + pe[pos + 1])
    const k::ChainNext cn{ (k::DecStep *) b.step_dev, w.d_te, w.d_pe, b.dx, S, n_ctx,
                           fronted ? (const uint32_t *) ((const unsigned char *) b.front_hand + (size_t) b.B * 16 * S) + 4 : nullptr };
    if (M & 1024) k::filter_argmax(b.logits, ctx.state->dev.ban_dev, stp, (k::SampleOut *) b.sample_dev, b.filter_scratch, s,
                                   (k::SampleOut *) b.sample_host, nb, S <= 1536 ? &cn : nullptr);
}

bool decode_rows_step(whisper_context & ctx, int nb) {
    BatchWork & b = *ctx.batch;
    const int64_t t0 = time_us();
    hipStream_t s = ctx.state->dev.stream;
    {
        // every per-step quantity reaches the kernels through the step records in pinned memory, so the launch sequence of a
        // (rows, encoder length) pair is static: replayed as ONE graph launch instead of 46 host-paced launches (the host needs
        // ~3 us per launch of this kernarg size, the device 1.6 us between dependent kernels: scratch/lab/chain_lab.hip)
        static const bool use_graph = getenv("WMI_NO_GRAPH") == nullptr;
        static const bool no_chain = getenv("WMI_NO_CHAIN") != nullptr;         // debug / A-B
        const k::DecStep * hs = (const k::DecStep *) b.step_host;
        bool chained = !no_chain && b.chain_valid && b.chain_nb == nb && !ctx.model.quantised;
        for (int r = 0; chained && r < nb; ++r)
            chained = b.chain_row_ok[r] && hs[r].token == b.chain_token[r] && hs[r].pos == b.chain_pos[r] && hs[r].kv_head == b.chain_pos[r] &&
                      hs[r].n_kv == b.chain_pos[r] + 1;
        b.chain_valid = false; b.n_chained += chained ? 1 : 0;
        if (b.front_backoff > 0) --b.front_backoff;
        const bool fronted = rows_fronted(ctx, nb);
        BatchWork::RowsGraph & rg = b.rows_graph[(chained ? 1 : 0) | (fronted ? 2 : 0)];
        // the key includes the epoch of the run-time kernel switches (wmi_set_lockstep_exact: VALU rows / one-group attention):
        // a step captured under the other mode would keep replaying that mode's kernels
        const int epoch = k::mode_epoch();
        if (rg.nb != nb || rg.T != b.enc_T || rg.rows != b.enc_rows || rg.epoch != epoch) {
            if (rg.exec) (void) hipGraphExecDestroy(rg.exec);
            if (rg.graph) (void) hipGraphDestroy(rg.graph);
            rg = BatchWork::RowsGraph{}; rg.nb = nb; rg.T = b.enc_T; rg.rows = b.enc_rows; rg.epoch = epoch;
        }
        ++rg.seen;                                                // (counted over both forms: the embedding form runs once per window)
        const BatchWork::RowsGraph & og = b.rows_graph[(chained ? 0 : 1) | (fronted ? 2 : 0)];
        const int seen_other = (og.nb == nb && og.T == b.enc_T && og.rows == b.enc_rows && og.epoch == epoch) ? og.seen : 0;
        if (use_graph && !rg.exec && !rg.failed && rg.seen + seen_other > 24) {
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                enqueue_rows_step(ctx, nb, chained, fronted ? 1 : 0);
                hipGraph_t g = nullptr;
                if (hipStreamEndCapture(s, &g) == hipSuccess && g && hipGraphInstantiate(&rg.exec, g, nullptr, nullptr, 0) == hipSuccess) rg.graph = g;
                else {
                    WMI_WARN("%s: graph capture failed - staying on eager launches\n", __func__);
                    if (g) (void) hipGraphDestroy(g);
                    rg.exec = nullptr; rg.failed = true;
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { hipGraph_t junk = nullptr; (void) hipStreamEndCapture(s, &junk); if (junk) (void) hipGraphDestroy(junk); }
                }
            } else rg.failed = true;
        }
        if (use_graph && rg.exec) HIP_TRY(hipGraphLaunch(rg.exec, s));
        else enqueue_rows_step(ctx, nb, chained, fronted ? 1 : 0);
    }
    {   // every row's result carries the step's sequence number (set by the caller in the step records)
        const k::DecStep * hs = (const k::DecStep *) b.step_host;
        const k::SampleOut * so = (const k::SampleOut *) b.sample_host;
        int32_t status_all = 0;
        for (int r = 0; r < nb; ++r) { int32_t st1 = 0; if (!wait_for_sample(&so[r], hs[r].seq, s, &st1)) return false; status_all |= st1; }
        if (status_all & (k::SAMPLE_TAG_FAULT | k::SAMPLE_TAG_SLOW)) {
            uint32_t * word = (uint32_t *) ((unsigned char *) b.front_hand + (size_t) b.B * 16 * ctx.model.hp.n_text_state) + 4;
            HIP_TRY(hipMemsetAsync(word, 0, sizeof(uint32_t), s));
            if (status_all & k::SAMPLE_TAG_FAULT) {
                // a hand-off inside a k_front launch did not complete: this step's rows are not to be trusted.  The step is run again from the
                // host's records in the two-launch form (embedding launch: every cache cell and device-side record rewritten), which the batch keeps
                if (!b.front_off) WMI_WARN("%s: in-launch hand-off of a layer's front failed (status %#x) - step re-run, staying on the two-launch form\n", __func__, (unsigned) status_all);
                b.front_off = true; ++b.front_fallbacks;
                ++b.step_seq;
                k::DecStep * hw = (k::DecStep *) b.step_host;
                for (int r = 0; r < nb; ++r) hw[r].seq = b.step_seq;
                enqueue_rows_step(ctx, nb, false, 0);
                for (int r = 0; r < nb; ++r) { int32_t st1 = 0; if (!wait_for_sample(&so[r], hs[r].seq, s, &st1) || (st1 & k::SAMPLE_TAG_FAULT)) { WMI_ERR("%s: the step's re-run failed\n", __func__); return false; } }
            } else b.front_backoff = 512;                        // a slow sweep: something else owns part of the device — two launches for a while
        }
        // what the pick kernel has left on the device for the next step (k_filter_pick prepares rows of <= 3 x 512 columns)
        const HParams & hp = ctx.model.hp;
        b.chain_valid = !ctx.model.quantised && hp.n_text_state <= 1536 && nb <= 16; b.chain_nb = nb;
        for (int r = 0; b.chain_valid && r < nb; ++r) {
            b.chain_token[r] = so[r].id; b.chain_pos[r] = hs[r].pos + 1;
            b.chain_row_ok[r] = hs[r].pos + 1 < hp.n_text_ctx && hs[r].n_kv == hs[r].pos + 1 && hs[r].kv_head == hs[r].pos;
        }
    }
    b.t_decode_us += time_us() - t0; b.n_steps++;
    return true;
}

// per-chunk bookkeeping of one lock-step group
struct Row {
    int chunk = 0, lane = 0;
    int seek = 0, seek_start = 0, seek_end = 0;
    bool live = false;                       // still has windows to decode in lock-step
    bool redo = false;                       // needs the temperature fallback: re-run alone
    std::vector<int32_t> prompt;
    int n_fed = 0;                           // prompt tokens already through the decoder
    int i = 0;                               // sampled tokens accepted in this window
    bool done = false;                       // window finished (completed, failed or length limit)
};

} // namespace

// probe: the kernels of a lock-step step for nb rows, `iters` times back to back with no host round trip; microseconds per step
double bench_rows_step_chain(whisper_context & ctx, int nb, int iters) {
    if (!ctx.batch || ctx.batch->B < nb || !ctx.batch->step_dev || iters <= 0) return -1.0;
    const char * mask_env = getenv("WMI_STEP_MASK");
    g_rows_mask = mask_env ? (unsigned) strtoul(mask_env, nullptr, 0) : ~0u;
    ctx.batch->chain_valid = false;                         // the replays below rewrite the device-side records and activation rows
    hipStream_t s = ctx.state->dev.stream;
    hipEvent_t e0, e1;
    if (!HIP_OK(hipEventCreate(&e0)) || !HIP_OK(hipEventCreate(&e1))) return -1.0;
    // replayed from a captured graph like the product's step (eager launches are paced by the host, see bench_greedy_step_chain)
    static const bool eager = getenv("WMI_CHAIN_EAGER") != nullptr;
    enqueue_rows_step(ctx, nb);
    (void) hipStreamSynchronize(s);
    hipGraph_t pg = nullptr; hipGraphExec_t pexec = nullptr;
    if (!eager && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        enqueue_rows_step(ctx, nb);
        if (hipStreamEndCapture(s, &pg) != hipSuccess || !pg || hipGraphInstantiate(&pexec, pg, nullptr, nullptr, 0) != hipSuccess) pexec = nullptr;
    }
    auto once = [&]() { if (pexec) (void) hipGraphLaunch(pexec, s); else enqueue_rows_step(ctx, nb); };
    for (int i = 0; i < 3; ++i) once();
    (void) hipStreamSynchronize(s);
    (void) hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) once();
    (void) hipEventRecord(e1, s);
    (void) hipEventSynchronize(e1);
    float ms = 0.0f; (void) hipEventElapsedTime(&ms, e0, e1);
    if (pexec) (void) hipGraphExecDestroy(pexec);
    if (pg) (void) hipGraphDestroy(pg);
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    g_rows_mask = ~0u;
    return (double) ms * 1000.0 / iters;
}

// A replica context: the model directory and the device view of the weights copied, the arena itself borrowed (read-only for every
// kernel), a state of its own (caches, activations, stream, graphs).  The reference's precedent is whisper_full_parallel's one
// whisper_state per worker over one shared model (W/whisper.cpp:5837-5858).
static whisper_context * new_replica(whisper_context & ctx) {
    whisper_context * r = nullptr;
    try {
        r = new whisper_context();
        r->params = ctx.params; r->model = ctx.model; r->device = ctx.device;
        r->w = ctx.w; r->w.arena_borrowed = true;
        hipStream_t adopt = nullptr;
        if (!ctx.spare_streams.empty()) { adopt = ctx.spare_streams.back(); ctx.spare_streams.pop_back(); }
        if (!init_state(*r, true, adopt)) { free_state(*r); delete r; return nullptr; }
    } catch (const std::exception & e) {
        WMI_ERR("%s: %s\n", __func__, e.what());
        if (r) { free_state(*r); delete r; }
        return nullptr;
    }
    return r;
}
static void free_replica(whisper_context * r) {
    if (!r) return;
    free_batch(*r);
    free_state(*r);
    free_weights(r->w);                                        // borrowed: forgets the pointer
    delete r;
}

// Surplus replicas are released when fewer are wanted (each holds a whole State — ~0.6 GB for large-v3 — and its stream goes back to the
// process-wide pool of own-queue streams): a caller that lowers wmi_set_batch_replicas gets the memory back without whisper_free.
void trim_replicas(whisper_context & ctx, int keep) {
    if (!ctx.batch) return;
    auto & reps = ctx.batch->replicas;
    while ((int) reps.size() > std::max(keep, 0)) { free_replica(reps.back()); reps.pop_back(); }
}

int ensure_replicas(whisper_context & ctx, int n) {
    if (!ctx.batch) ctx.batch = new BatchWork();
    while ((int) ctx.batch->replicas.size() < n) {
        whisper_context * r = new_replica(ctx);
        if (!r) break;
        ctx.batch->replicas.push_back(r);
    }
    return std::min(n, (int) ctx.batch->replicas.size());
}

void free_batch(whisper_context & ctx) {
    if (!ctx.batch) return;
    BatchWork & w = *ctx.batch;
    dfree(w.mel_t); dfree(w.conv1); dfree(w.x); dfree(w.xn); dfree(w.q); dfree(w.k); dfree(w.att); dfree(w.vt); dfree(w.h);
    dfree(w.enc_out_h); dfree(w.kvc_k); dfree(w.kvc_v); dfree(w.self_k); dfree(w.self_v); dfree(w.dx); dfree(w.dq); dfree(w.datt); dfree(w.dh);
    dfree(w.logits); dfree(w.xattn); dfree(w.aq); dfree(w.ads); dfree(w.aq16); dfree(w.wq16); dfree(w.att32); dfree(w.datt32);
    for (auto & rg : w.rows_graph) {
        if (rg.exec) (void) hipGraphExecDestroy(rg.exec);
        if (rg.graph) (void) hipGraphDestroy(rg.graph);
        rg = BatchWork::RowsGraph{};
    }
    if (w.front_hand) (void) hipFree(w.front_hand);
    if (w.step_dev) (void) hipFree(w.step_dev);
    if (w.sample_dev) (void) hipFree(w.sample_dev);
    if (w.filter_scratch) (void) hipFree(w.filter_scratch);
    if (w.step_host) (void) hipHostFree(w.step_host);
    if (w.sample_host) (void) hipHostFree(w.sample_host);
    for (size_t i = 1; i < w.lanes.size(); ++i) free_lane_state(w.lanes[i]);
    for (whisper_context * r : w.replicas) free_replica(r);
    delete ctx.batch;
    ctx.batch = nullptr;
}

int full_batch(whisper_context & ctx, whisper_full_params params, const float * const * pcm, const int * n_samples, int n_chunks,
               bool on_device) {
    if (!compute_ready(ctx, __func__)) return -2;
    if (n_chunks <= 0) return 0;
    BusyScope busy(ctx.device);                                    // (a lock-step call owns the GPU as much as a transcription does: wmi.h)
    const Vocab & v = ctx.model.vocab;
    const HParams & hp = ctx.model.hp;
    State * primary = ctx.state;
    if (!ctx.batch) ctx.batch = new BatchWork();
    ctx.batch->results.assign(n_chunks, {});
    ctx.batch->redo.assign(n_chunks, 0);
    ctx.batch->t_mel_us = ctx.batch->t_encode_us = ctx.batch->t_decode_us = ctx.batch->t_emit_us = 0; ctx.batch->n_steps = 0; ctx.batch->n_chained = 0;

    auto run_alone = [&](int c) -> int {                          // the general driver, one chunk at a time
        // as on a fresh whisper_state (what every whisper_full_parallel worker gets, W/whisper.cpp:5837-5843): the
        // decoders' generators start from their initial seed (W/whisper.cpp:3077), so a chunk's result does not
        // depend on which chunks were transcribed before it
        for (auto & dec : primary->decoders) dec.rng = std::mt19937(0);
        const int rc = full(ctx, params, on_device ? nullptr : pcm[c], on_device ? pcm[c] : nullptr, n_samples[c]);
        if (rc == 0) ctx.batch->results[c] = std::move(primary->result_all);
        primary->result_all.clear();
        return rc;
    };

    static const bool force_seq = getenv("WMI_BATCH_SEQUENTIAL") != nullptr;     // debug / A-B
    const bool lang_known = params.language && strlen(params.language) > 0 && strcmp(params.language, "auto") != 0 && !params.detect_language;
    const bool distilled = hp.n_text_layer == 2 && !params.no_timestamps;
    const bool lockstep = !force_seq && fast_path_enabled() && params.strategy == WHISPER_SAMPLING_GREEDY && params.temperature < 1e-6f &&
                          lang_known && !distilled && !params.speed_up && !params.logits_filter_callback && !params.grammar_rules && params.n_grammar_rules == 0 && !params.new_segment_callback &&
                          !params.progress_callback && !params.encoder_begin_callback && !params.abort_callback &&
                          ctx.model.n_loaded > 0 && n_chunks > 1;
    if (!lockstep) {
        // Chunks the lock-step rows cannot carry (beam search, t > 0, callbacks ...): the general driver, chunk by chunk — and, where
        // nothing observable depends on the order, on several REPLICA contexts at once: worker w takes the chunks c = w (mod workers),
        // each on its own state and stream, so the short launches of one chunk's decode steps (40-160 workgroups on 256 CUs) run
        // beside another chunk's.  Every chunk is still one whisper_full on a fresh state: results are identical to the sequential
        // loop by construction.  Measured (profiles/r04d_replica_streams_hw_queues.txt): beam 5, 8 chunks, base.en 9.3 -> 3.3 ms per chunk,
        // large-v3 q5_1 46 -> 18 ms per chunk with four contexts.  Not with user callbacks (they would run concurrently, on contexts the
        // caller never saw) or print_realtime (its stdout text would interleave).
        static const int rep_env = getenv("WMI_BATCH_REPLICAS") ? atoi(getenv("WMI_BATCH_REPLICAS")) : 3;
        const int rep_want = ctx.batch->replicas_wanted >= 0 ? ctx.batch->replicas_wanted : rep_env;
        const bool observers = params.logits_filter_callback || params.new_segment_callback || params.progress_callback ||
                               params.encoder_begin_callback || params.abort_callback || params.print_realtime;      // (print_progress only writes percent lines to stderr)
        int n_rep = (!force_seq && !observers && ctx.model.n_loaded > 0) ? std::min(rep_want, n_chunks - 1) : 0;
        if (n_rep < 0) n_rep = 0;
        n_rep = ensure_replicas(ctx, n_rep);                             // (out of memory: fewer workers)
        static const bool dbg_rep = getenv("WMI_DEBUG_TIMING") != nullptr;
        if (dbg_rep) fprintf(stderr, "[wmi] full_batch: %d chunks through the general driver on 1 + %d contexts (wanted %d, observers %d)\n", n_chunks, n_rep, rep_want, (int) observers);
        if (n_rep == 0) {
            for (int c = 0; c < n_chunks; ++c) { const int rc = run_alone(c); if (rc != 0) return rc; ctx.batch->redo[c] = 1; }
            return 0;
        }
        const int workers = n_rep + 1;
        std::vector<int> rets(workers, 0);
        auto work = [&](int w) {
            try {
                whisper_context & wc = w == 0 ? ctx : *ctx.batch->replicas[w - 1];
                (void) hipSetDevice(ctx.device);
                for (int c = w; c < n_chunks; c += workers) {
                    int rc;
                    if (w == 0) rc = run_alone(c);
                    else {
                        for (auto & dec : wc.state->decoders) dec.rng = std::mt19937(0);
                        rc = full(wc, params, on_device ? nullptr : pcm[c], on_device ? pcm[c] : nullptr, n_samples[c]);
                        if (rc == 0) ctx.batch->results[c] = std::move(wc.state->result_all);
                        wc.state->result_all.clear();
                    }
                    if (rc != 0) { rets[w] = rc; return; }
                    ctx.batch->redo[c] = 1;
                }
            } catch (const std::exception & e) {
                WMI_ERR("wmi_full_batch: worker %d: %s\n", w, e.what());
                rets[w] = -10;
            } catch (...) { rets[w] = -10; }
        };
        {
            std::vector<std::thread> th;
            struct Joiner { std::vector<std::thread> & t; ~Joiner() { for (auto & x : t) if (x.joinable()) x.join(); } } joiner{th};
            for (int w = 1; w < workers; ++w) th.emplace_back(work, w);
            work(0);
        }
        for (int w = 0; w < workers; ++w) if (rets[w] != 0) return rets[w];
        return 0;
    }
    if (params.audio_ctx > hp.n_audio_ctx) {
        WMI_ERR("%s: audio_ctx is larger than the maximum allowed (%d > %d)\n", __func__, params.audio_ctx, hp.n_audio_ctx);
        return -5;
    }
    // ---- lock-step GROUPS side by side (round 6).  The chunks are dealt to `groups` contiguous ranges, each a lock-step call of its own — range
    // 0 on this context, the others on replica contexts (own state, stream and work set, the weights borrowed) on host threads of their own.
    // A chunk's result does not depend on which chunks share its launches (per-row arithmetic; the encoder attention's key split follows the
    // grid size: bits can differ between groupings in the default mode as they do between chunk counts, never in the exact mode).
    // Measured, base.en, ms per call with host PCM (scratch/r06_groups_time.py, profiles/r06k_*): one-row transcriptions gain 1.7 x from a second
    // chain, lock-step rows do not — their launches are 4 - 16 x wider and two 4-row chains take as long per step (207 us) as one 8-row chain
    // (212): 8 chunks 5.80 - 5.85 (one group) / 5.82 - 5.90 (two) / 6.21 (three); 16 chunks 9.07 - 9.14 / 8.32 - 8.34 / 8.86 — what two groups
    // of eight win is the width limit of the rows kernels.  Default: two groups from 16 chunks on (WMI_LOCKSTEP_GROUPS / wmi_set_lockstep_groups
    // set the count for every call of >= 4 chunks; 1 = one chain).
    static thread_local bool t_in_group = false;
    if (!t_in_group) {
        static const int groups_env = getenv("WMI_LOCKSTEP_GROUPS") ? atoi(getenv("WMI_LOCKSTEP_GROUPS")) : 0;
        int G = ctx.batch->groups_wanted > 0 ? ctx.batch->groups_wanted : groups_env > 0 ? groups_env : (n_chunks >= 16 ? 2 : 1);
        G = std::max(1, std::min(G, n_chunks / 2));
        if (G > 1) G = ensure_replicas(ctx, G - 1) + 1;                    // (out of memory: fewer groups)
        if (G > 1) {
            std::vector<std::vector<Segment>> all(n_chunks);
            std::vector<int> all_redo(n_chunks, 0), rets(G, 0), c0(G + 1, 0);
            for (int g = 0; g < G; ++g) c0[g + 1] = c0[g] + n_chunks / G + (g < n_chunks % G ? 1 : 0);
            int64_t tm[4] = {0, 0, 0, 0}; int steps = 0, chained = 0;
            std::mutex merge_mu;
            auto work = [&](int g) {
                try {
                    whisper_context & wc = g == 0 ? ctx : *ctx.batch->replicas[g - 1];
                    (void) hipSetDevice(ctx.device);
                    if (g > 0 && wc.batch) wc.batch->groups_wanted = 1;
                    t_in_group = true;
                    struct Off { ~Off() { t_in_group = false; } } off;
                    const int cnt = c0[g + 1] - c0[g];
                    const int rc = full_batch(wc, params, pcm + c0[g], n_samples + c0[g], cnt, on_device);
                    rets[g] = rc;
                    if (rc != 0 || !wc.batch) return;
                    std::lock_guard<std::mutex> lk(merge_mu);
                    for (int i = 0; i < cnt; ++i) { all[c0[g] + i] = std::move(wc.batch->results[i]); all_redo[c0[g] + i] = wc.batch->redo[i]; }
                    tm[0] = std::max(tm[0], wc.batch->t_mel_us); tm[1] = std::max(tm[1], wc.batch->t_encode_us);
                    tm[2] = std::max(tm[2], wc.batch->t_decode_us); tm[3] = std::max(tm[3], wc.batch->t_emit_us);
                    steps = std::max(steps, wc.batch->n_steps); chained = std::max(chained, wc.batch->n_chained);
                } catch (const std::exception & e) {
                    WMI_ERR("wmi_full_batch: lock-step group %d: %s\n", g, e.what());
                    rets[g] = -10;
                } catch (...) { rets[g] = -10; }
            };
            {
                std::vector<std::thread> th;
                struct Joiner { std::vector<std::thread> & t; ~Joiner() { for (auto & x : t) if (x.joinable()) x.join(); } } joiner{th};
                for (int g = 1; g < G; ++g) th.emplace_back(work, g);
                work(0);
            }
            for (int g = 0; g < G; ++g) if (rets[g] != 0) return rets[g];
            BatchWork & bw = *ctx.batch;
            bw.results = std::move(all); bw.redo = std::move(all_redo);
            bw.t_mel_us = tm[0]; bw.t_encode_us = tm[1]; bw.t_decode_us = tm[2]; bw.t_emit_us = tm[3]; bw.n_steps = steps; bw.n_chained = chained;
            bw.groups_last = G;
            return 0;
        }
    }
    ctx.batch->groups_last = 1;
    const int max_lanes = k::rows_valu_enabled() ? 8 : MAX_LANES;
    if (!ensure_batch(ctx, std::min(n_chunks, max_lanes))) return -2;
    BatchWork & b = *ctx.batch;
    if (!upload_static_ban(ctx, params)) return -7;
    // the envelope kernels read the caller's samples on side streams: never return while one is in flight
    struct EnvelopeGuard { BatchWork & b; ~EnvelopeGuard() { for (State * l : b.lanes) (void) signal_energy_wait(*l); } } envelope_guard{b};

    const bool has_fallback = params.temperature_inc > 0.0f && params.temperature + params.temperature_inc < 1.0f + 1e-6f;
    std::vector<int32_t> prompt_user;
    if (!params.prompt_tokens && params.initial_prompt) {
        prompt_user = tokenize(v, params.initial_prompt);
        if (prompt_user.size() > 1024) {
            WMI_ERR("%s: too many resulting tokens: %d (max %d)\n", "whisper_tokenize", (int) prompt_user.size(), 1024);
            prompt_user.clear();
        }
    } else if (params.prompt_tokens && params.prompt_n_tokens > 0) {
        prompt_user.assign(params.prompt_tokens, params.prompt_tokens + params.prompt_n_tokens);
    }
    std::vector<int32_t> prompt_init = { v.sot };
    if (v.is_multilingual()) {
        const int lid = lang_id(params.language);
        prompt_init.push_back(v.sot + 1 + lid);
        prompt_init.push_back(params.translate ? v.translate : v.transcribe);
    }
    if (params.no_timestamps) prompt_init.push_back(v.not_);
    int space_id = -1;
    { auto sp = v.token_to_id.find(" "); if (sp != v.token_to_id.end()) space_id = sp->second; }
    const int n_max = hp.n_text_ctx / 2 - 4;

    const int group = std::min(b.B, max_lanes);
    for (int g0 = 0; g0 < n_chunks; g0 += group) {
        const int ng = std::min(group, n_chunks - g0);
        std::vector<Row> rows(ng);
        static const int env_when_ = getenv("WMI_ENVELOPE_WHEN") ? atoi(getenv("WMI_ENVELOPE_WHEN")) : 0;
        // The envelopes of a lock-step call stay in HBM and the window sums + walks of the token timestamps run there (device.cpp
        // ts_refine_device, k_ts_refine: one workgroup per token, the window sum as order-free integer sums per binade over eight wavefronts —
        // every value equals the host loop's, tests/test_gpu_host_dsp.py).  15 MB of PCIe writes per 8-chunk call are not made: mel phase
        // 0.74 -> 0.57 ms, segments + timestamps unchanged at ~0.25 ms (one device call per window for all chunks), 6.14 -> 5.92 ms per call
        // (profiles/r04g_token_timestamps_on_device.txt).  WMI_TS_DEVICE=0: envelopes to pinned host memory, sums and walks on the host.
        static const bool ts_device = !(getenv("WMI_TS_DEVICE") && atoi(getenv("WMI_TS_DEVICE")) == 0);
        const bool env_interleaved = env_when_ == 0 || env_when_ >= 3;   // each chunk's envelope kernel right behind its mel kernels (3: into HBM, copied out beside the decode steps)
        // ---- per chunk: PCM -> mel, envelope, window bounds (the head of full())
        // The mel kernels of all chunks, and their envelopes where those stay in HBM, are ONE launch per kernel (device.cpp:
        // pcm_to_mel_batch): per chunk on its own stream (3 mel launches + envelope + two event operations, 8 + 8 streams) the phase
        // was ~0.3 ms of host enqueue time per 8-chunk call.  WMI_MEL_PER_CHUNK=1 (A/B): the per-chunk form.
        static const bool mel_per_chunk = getenv("WMI_MEL_PER_CHUNK") != nullptr;
        const bool env_batched = params.token_timestamps && ts_device && env_when_ == 0;
        const bool mel_batched = !mel_per_chunk && ng >= 2 && (env_batched || !params.token_timestamps);
        if (mel_batched) {
            const int64_t tm0 = time_us();
            std::vector<State *> sts(ng);
            for (int r = 0; r < ng; ++r) sts[r] = b.lanes[r];
            if (!pcm_to_mel_batch(ctx, sts, pcm + g0, n_samples + g0, on_device, env_batched)) { WMI_ERR("%s: failed to compute log mel spectrogram\n", __func__); return -2; }
            b.t_mel_us += time_us() - tm0;
        }
        for (int r = 0; r < ng; ++r) {
            Row & row = rows[r]; row.chunk = g0 + r; row.lane = r;
            State & ls = *b.lanes[r];
            StateSwap sw(ctx, &ls);
            ls.result_all.clear();
            ls.ts_failed = false;
            ls.prompt_past = prompt_user;                     // every chunk is an independent transcription (no_context semantics)
            ls.exp_n_audio_ctx = params.audio_ctx;
            if (v.is_multilingual()) ls.lang_id = lang_id(params.language);
            const int64_t tm0 = time_us();
            // kernels of all chunks are queued back to back; one synchronisation after the loop
            if (n_samples[row.chunk] > 0 && !mel_batched) {
                if (!pcm_to_mel(ctx, pcm[row.chunk], n_samples[row.chunk], on_device, false)) { WMI_ERR("%s: failed to compute log mel spectrogram\n", __func__); return -2; }
            }
            if (params.token_timestamps) {
                ls.t_beg = 0; ls.t_last = 0; ls.tid_last = 0;
                if (!mel_batched && env_interleaved && n_samples[row.chunk] > 0 && !signal_energy_device(ctx, 32, false, (ts_device && env_when_ == 0) ? 3 : env_when_ >= 3 ? 2 : 0)) { WMI_ERR("%s: failed to compute the signal envelope\n", __func__); return -2; }
            }
            b.t_mel_us += time_us() - tm0;
        }
        // The |x| envelopes (token timestamps): 1875 workgroups per chunk whose waves sit on stores into pinned host memory (1.9 MB per
        // chunk over PCIe).  WMI_ENVELOPE_WHEN (A/B): 0 = behind each chunk's mel kernels (default), 1 = behind the mel kernels of all
        // chunks (beside the encoder), 2 = behind the first window's encoder (beside the decode steps).  Measured (profiles/
        // r04b_envelope_placement.txt): they cost 0.3-0.4 ms of the 8-chunk call WHEREVER they run — mel phase 0.60 -> 0.24 ms with 1 or 2,
        // and the encoder (its persistent GEMMs want every CU at once) or the decode steps (their 32-byte results queue behind 15 MB of
        // bulk writes) give the same time back; the copy-engine form (WMI_ENVELOPE_DMA) is a blit kernel on this stack (rocprof:
        // __amd_rocclr_copyBuffer, no SDMA), a thin grid (WMI_ENVELOPE_GRID) starves the kernel's 65-deep f64 chains.
        static const int env_dma = getenv("WMI_ENVELOPE_DMA") ? std::max(1, atoi(getenv("WMI_ENVELOPE_DMA"))) : 0;
        auto envelopes = [&]() -> bool {
            if (!params.token_timestamps) return true;
            for (int r = 0; r < ng; ++r) {
                State & ls = *b.lanes[r];
                StateSwap sw(ctx, &ls);
                if (n_samples[g0 + r] > 0 && !signal_energy_device(ctx, 32, false, env_dma)) { WMI_ERR("%s: failed to compute the signal envelope\n", __func__); return false; }
            }
            return true;
        };
        // one synchronisation for the mel kernels of all chunks (keeps the mel / encoder time buckets separate); the
        // envelopes are written to the host by the chunks' side streams and are awaited at emission time
        {
            const int64_t tm0 = time_us();
            for (int r = 0; r < ng; ++r) {                        // the encoder (main stream) waits for every chunk's mel
                DeviceState & ld = b.lanes[r]->dev;
                if (mel_batched || !ld.mel_stream || n_samples[g0 + r] <= 0) continue;      // (batched: the kernels are on the main stream already)
                if (!HIP_OK(hipEventRecord(ld.mel_ev, ld.mel_stream)) || !HIP_OK(hipStreamWaitEvent(primary->dev.stream, ld.mel_ev, 0))) return -2;
            }
            if (!HIP_OK(hipStreamSynchronize(primary->dev.stream))) return -2;
            b.t_mel_us += time_us() - tm0;
        }
        bool env_done = env_interleaved || !params.token_timestamps, env_flushed = false;
        if (env_when_ == 4 && params.token_timestamps) {        // thin copies beside the encoder
            for (int r = 0; r < ng; ++r) if (!signal_energy_flush(*b.lanes[r])) return -2;
            env_flushed = true;
        }
        if (!env_done && env_when_ == 1) { if (!envelopes()) return -2; env_done = true; }
        for (int r = 0; r < ng; ++r) {
            Row & row = rows[r]; State & ls = *b.lanes[r];
            row.seek_start = params.offset_ms / 10;
            row.seek_end = params.duration_ms == 0 ? ls.mel.n_len_org : row.seek_start + params.duration_ms / 10;
            row.seek = row.seek_start;
            row.live = row.seek_end >= row.seek_start + 100;   // < 1 s of audio: nothing to do
        }

        // ---- windows in lock-step
        // (chunks too short for a window never reach the encoder: their envelopes are not needed either — emission only reads them for
        //  decoded tokens)
        while (true) {
            std::vector<int> act;                              // row r of the kernels <-> rows[act[r]]
            for (int r = 0; r < ng; ++r) if (rows[r].live && !rows[r].redo && rows[r].seek + 100 < rows[r].seek_end) act.push_back(r);
            if (act.empty()) break;
            const int nb = (int) act.size();
            {
                std::vector<int> lanes(nb), seeks(nb);
                for (int r = 0; r < nb; ++r) { lanes[r] = rows[act[r]].lane; seeks[r] = rows[act[r]].seek; }
                if (!encode_rows(ctx, lanes, seeks, params.audio_ctx)) { WMI_ERR("%s: failed to encode\n", __func__); return -6; }
                if (!env_done) { if (!envelopes()) return -2; env_done = true; }
                if (env_when_ == 3 && !env_flushed) {            // the thin copies start now: beside the decode steps, done long before emission
                    for (int r = 0; r < ng; ++r) if (!signal_energy_flush(*b.lanes[r])) return -2;
                    env_flushed = true;
                }
                b.chain_valid = false;                             // new windows, possibly other chunks in the rows: every row restarts at cell 0
            }
            for (int r = 0; r < nb; ++r) {
                Row & row = rows[act[r]]; State & ls = *b.lanes[row.lane];
                if (row.seek > row.seek_start && row.seek + 500 >= row.seek_end) ls.prompt_past.clear();
                Decoder & d = ls.decoders[0];
                d.sequence.tokens.clear();
                d.sequence.result_len = 0; d.sequence.sum_logprobs_all = 0.0;
                d.sequence.sum_logprobs = -INFINITY; d.sequence.avg_logprobs = -INFINITY;
                d.sequence.entropy = 0.0; d.sequence.score = -INFINITY;
                d.seek_delta = 100 * WHISPER_CHUNK_SIZE;
                d.failed = false; d.completed = false; d.has_ts = false;
                row.prompt.clear();
                if (!ls.prompt_past.empty() && params.n_max_text_ctx > 0) {      // t = 0 < 0.5
                    const int n_take = std::min(std::min(params.n_max_text_ctx, hp.n_text_ctx / 2), (int) ls.prompt_past.size());
                    row.prompt.push_back(v.prev);
                    row.prompt.insert(row.prompt.end(), ls.prompt_past.end() - n_take, ls.prompt_past.end());
                }
                row.prompt.insert(row.prompt.end(), prompt_init.begin(), prompt_init.end());
                row.n_fed = 0; row.i = 0; row.done = false;
            }

            // ---- decode steps: every row feeds one token (prompt tokens first, one per step, then its own samples)
            while (true) {
                bool any = false;
                k::DecStep * hs = (k::DecStep *) b.step_host;
                for (int r = 0; r < nb; ++r) {
                    Row & row = rows[act[r]]; const Decoder & d = b.lanes[row.lane]->decoders[0];
                    k::DecStep & st = hs[r];
                    memset(&st, 0, sizeof(st));
                    st.seq = b.step_seq + 1;
                    st.space_id = space_id; st.eot = v.eot; st.beg = v.beg; st.n_vocab = v.n_vocab;
                    st.ts_floor_end = v.beg; st.ts_initial_start = v.n_vocab;
                    if (row.done) {                                                                            // idle row: result discarded
                        // it keeps following its own picks while the device-side chain allows, so that the other rows' step stays
                        // chained (rows are independent: own caches, own activation row); otherwise it restarts at cell 0
                        if (b.chain_valid && b.chain_nb == nb && b.chain_row_ok[r] && b.chain_pos[r] + 1 < hp.n_text_ctx) {
                            st.token = b.chain_token[r]; st.pos = b.chain_pos[r]; st.n_kv = st.pos + 1; st.kv_head = st.pos;
                        } else { st.token = v.eot; st.pos = 0; st.n_kv = 1; st.kv_head = 0; }
                        continue;
                    }
                    any = true;
                    const int np = (int) row.prompt.size();
                    if (row.n_fed < np) { st.token = row.prompt[row.n_fed]; st.pos = row.n_fed; }
                    else                { st.token = d.sequence.tokens.back().id; st.pos = np + row.i - 1; }
                    st.n_kv = st.pos + 1; st.kv_head = st.pos;
                    // filter state of the token this step will sample (W/whisper.cpp:4541-4657); see full()'s step_filter
                    const auto & h = d.sequence.tokens;
                    const bool initial = h.empty();
                    const bool last_ts = !h.empty() && h.back().id >= v.beg;
                    const bool penult_ts = h.size() < 2 || h[h.size() - 2].id >= v.beg;
                    st.flags = ((params.suppress_blank && initial) ? 1 : 0) | (last_ts ? 2 : 0) | (penult_ts ? 4 : 0);
                    if (d.has_ts) st.ts_floor_end = v.beg + d.seek_delta / 2;
                    if (initial && params.max_initial_ts > 0.0f) {
                        const float precision = float(WHISPER_CHUNK_SIZE) / hp.n_audio_ctx;
                        st.ts_initial_start = v.beg + (int) std::round(params.max_initial_ts / precision) + 1;
                    }
                }
                if (!any) break;
                ++b.step_seq;
                if (!decode_rows_step(ctx, nb)) { WMI_ERR("%s: failed to decode\n", __func__); return -8; }
                const k::SampleOut * so = (const k::SampleOut *) b.sample_host;
                for (int r = 0; r < nb; ++r) {
                    Row & row = rows[act[r]];
                    if (row.done) continue;
                    const int np = (int) row.prompt.size();
                    if (row.n_fed < np) { row.n_fed++; if (row.n_fed < np) continue; }     // still inside the prompt: discard
                    Decoder & d = b.lanes[row.lane]->decoders[0];
                    const int i = row.i;
                    const whisper_token_data tok{ so[r].id, so[r].tid, so[r].p, so[r].plog, so[r].pt, so[r].ptsum, -1, -1, 0.0f };
                    d.sequence.tokens.push_back(tok);
                    d.sequence.sum_logprobs_all += tok.plog;
                    row.i++;
                    // state machine of one decoder (W/whisper.cpp:5450-5519)
                    int & result_len = d.sequence.result_len;
                    if (tok.id > v.beg) {
                        const int sd_new = 2 * (tok.id - v.beg);
                        if (d.has_ts && d.seek_delta > sd_new && result_len < i) { d.failed = true; row.done = true; continue; }
                        d.seek_delta = sd_new; result_len = i + 1; d.has_ts = true;
                    }
                    if (tok.id == v.eot || (params.max_tokens > 0 && i >= params.max_tokens) ||
                        (d.has_ts && row.seek + d.seek_delta + 100 >= row.seek_end)) {
                        if (result_len == 0) {
                            if (row.seek + d.seek_delta + 100 >= row.seek_end) result_len = i + 1;
                            else { d.failed = true; row.done = true; continue; }
                        }
                        if (params.single_segment) { result_len = i + 1; d.seek_delta = 100 * WHISPER_CHUNK_SIZE; }
                        d.completed = true; row.done = true;
                        continue;
                    }
                    if (i == n_max - 1) {
                        if (result_len == 0 || d.seek_delta < 100 * WHISPER_CHUNK_SIZE / 2) d.failed = true;
                        row.done = true;
                    }
                }
            }

            // ---- rank / fallback decision per row, then segment emission (token timestamps: host CPU work, one
            //      thread per chunk — every chunk owns its State)
            std::vector<int> emit;
            for (int r = 0; r < nb; ++r) {
                Row & row = rows[act[r]]; State & ls = *b.lanes[row.lane];
                Decoder & d = ls.decoders[0];
                if (!d.failed) {
                    d.sequence.tokens.resize(d.sequence.result_len);
                    sequence_score(params, d.sequence);
                    if (d.sequence.result_len > 32 && d.sequence.entropy < params.entropy_thold) { d.failed = true; primary->n_fail_h++; }
                }
                if (has_fallback && (d.failed || d.sequence.avg_logprobs < params.logprob_thold)) { row.redo = true; primary->n_fail_p++; continue; }
                emit.push_back(act[r]);
            }
            {
                const int64_t te0 = time_us();
                auto emit_row = [&](int ri) {
                    Row & row = rows[ri]; State & ls = *b.lanes[row.lane];
                    emit_window(ctx, ls, params, row.seek, row.prompt, prompt_init.size(), ls.decoders[0]);
                    row.seek += ls.decoders[0].seek_delta;
                };
                // token times refined on the device (envelopes in HBM): the chunks' pending segments go in ONE launch behind the per-chunk work
                // (eight threads each launching and synchronising their own small kernel took as long as eight launches in a row)
                std::vector<State *> held;
                if (params.token_timestamps && ts_device) for (int ri : emit) { State * ls = b.lanes[rows[ri].lane]; ls->ts_hold = true; held.push_back(ls); }
                if (params.token_timestamps && !params.print_realtime && emit.size() > 1) {
                    pool_run((int) emit.size(), [&](int e) { emit_row(emit[e]); });       // persistent workers (pool.cpp)
                } else {
                    for (int ri : emit) emit_row(ri);
                }
                for (State * ls : held) ls->ts_hold = false;
                if (!held.empty() && !flush_token_timestamps_of(ctx, held)) { WMI_ERR("%s: failed to refine the token timestamps on the device\n", __func__); return -9; }
                for (int ri : emit) if (b.lanes[rows[ri].lane]->ts_failed) { WMI_ERR("%s: failed to refine the token timestamps on the device\n", __func__); return -9; }
                b.t_emit_us += time_us() - te0;
            }
        }

        // ---- collect; chunks that asked for the temperature fallback go through the general driver alone
        for (int r = 0; r < ng; ++r) {
            Row & row = rows[r];
            if (row.redo) {
                b.redo[row.chunk] = 1;
                const int rc = run_alone(row.chunk);
                if (rc != 0) return rc;
            } else {
                b.results[row.chunk] = std::move(b.lanes[row.lane]->result_all);
                b.lanes[row.lane]->result_all.clear();
            }
        }
    }
    static const bool dbg_t = getenv("WMI_DEBUG_TIMING") != nullptr;
    if (dbg_t) fprintf(stderr, "[wmi] full_batch: %d chunks | mel+envelope %.3f ms | encode %.3f | decode %.3f (%d steps, %d chained) | segments+timestamps %.3f\n",
                       n_chunks, b.t_mel_us / 1e3, b.t_encode_us / 1e3, b.t_decode_us / 1e3, b.n_steps, b.n_chained, b.t_emit_us / 1e3);
    return 0;
}

} // namespace wmi
