// The rest of the whisper.h surface (W/whisper.h v1.5.4): caller-owned states (whisper_init_state and the
// *_with_state / *_from_state families), the deprecated and *_no_state constructors, the loader-callback
// constructors, whisper_full_parallel, the by-reference parameter helpers and the bench entry points.
//
// A whisper_state is this backend's wmi::State: its own KV caches, encoder/decoder activation arenas, HIP stream and
// captured decode graph on the context's GPU — the same ownership split as the reference (W/whisper.cpp:3001-3120:
// weights belong to the context, everything mutable to the state).  The compute code reaches its working set through
// ctx.state, which resolves per THREAD: a *_with_state call installs the caller's state for the calling thread and holds
// that state's lock.  Calls on one state serialise; calls on different states of one context run concurrently, each on its
// state's own stream (round 6; before, they took turns on the context's lock).

#include "wmi.h"
#include "kernels.h"

#include <cstring>
#include <fstream>
#include <thread>

using namespace wmi;

namespace {

inline State * S(struct whisper_state * s) { return reinterpret_cast<State *>(s); }

// A *_with_state call: the caller's state is what ctx.state means on THIS thread for the duration of the call (wmi.h: StateSlot), under
// the STATE's lock — calls on one state serialise, calls on different states of one context run side by side (W/whisper.cpp:5837-5858).
struct StateScope {
    std::unique_lock<std::recursive_mutex> lk; StateInstall inst; BusyScope busy;      // (counted once the lock is held: a waiting caller is not on the GPU)
    StateScope(whisper_context * c, struct whisper_state * s) : lk(S(s)->mu), inst(c->state, S(s)), busy(c->device) {
        if (!c->host_only) (void) hipSetDevice(c->device);
    }
};

bool read_file(const char * path, std::vector<char> & buf) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { WMI_ERR("%s: failed to open '%s'\n", __func__, path); return false; }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    buf.resize((size_t) n);
    if (!f.read(buf.data(), n)) { WMI_ERR("%s: failed to read '%s'\n", __func__, path); return false; }
    return true;
}

// drain a whisper_model_loader (W/whisper.h:108-114) into memory; the loader is closed in every case, as the
// reference does (W/whisper.cpp:3253-3269)
bool read_loader(struct whisper_model_loader * loader, std::vector<char> & buf) {
    if (!loader || !loader->read) return false;
    const size_t step = 8u << 20;
    size_t have = 0;
    for (;;) {
        buf.resize(have + step);
        const size_t got = loader->read(loader->context, buf.data() + have, step);
        have += got;
        if (got < step || (loader->eof && loader->eof(loader->context))) break;
    }
    buf.resize(have);
    if (loader->close) loader->close(loader->context);
    return have > 0;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------- constructors
struct whisper_context * whisper_init_from_buffer_with_params_no_state(void * buffer, size_t buffer_size, struct whisper_context_params params) {
    whisper_context * ctx = init_context(buffer, buffer_size, 0, false);
    if (ctx) ctx->params = params;
    return ctx;
}
struct whisper_context * whisper_init_from_file_with_params_no_state(const char * path, struct whisper_context_params params) {
    WMI_INFO("%s: loading model from '%s'\n", __func__, path);
    std::vector<char> buf;
    if (!read_file(path, buf)) return nullptr;
    return whisper_init_from_buffer_with_params_no_state(buf.data(), buf.size(), params);
}
struct whisper_context * whisper_init_with_params_no_state(struct whisper_model_loader * loader, struct whisper_context_params params) {
    std::vector<char> buf;
    if (!read_loader(loader, buf)) { WMI_ERR("%s: failed to load model\n", __func__); return nullptr; }
    return whisper_init_from_buffer_with_params_no_state(buf.data(), buf.size(), params);
}
struct whisper_context * whisper_init_with_params(struct whisper_model_loader * loader, struct whisper_context_params params) {
    std::vector<char> buf;
    if (!read_loader(loader, buf)) { WMI_ERR("%s: failed to load model\n", __func__); return nullptr; }
    return whisper_init_from_buffer_with_params(buf.data(), buf.size(), params);
}
// deprecated forms: default context parameters (W/whisper.cpp:3316-3338)
struct whisper_context * whisper_init_from_file(const char * path) { return whisper_init_from_file_with_params(path, whisper_context_default_params()); }
struct whisper_context * whisper_init_from_buffer(void * buffer, size_t n) { return whisper_init_from_buffer_with_params(buffer, n, whisper_context_default_params()); }
struct whisper_context * whisper_init(struct whisper_model_loader * loader) { return whisper_init_with_params(loader, whisper_context_default_params()); }
struct whisper_context * whisper_init_from_file_no_state(const char * path) { return whisper_init_from_file_with_params_no_state(path, whisper_context_default_params()); }
struct whisper_context * whisper_init_from_buffer_no_state(void * buffer, size_t n) { return whisper_init_from_buffer_with_params_no_state(buffer, n, whisper_context_default_params()); }
struct whisper_context * whisper_init_no_state(struct whisper_model_loader * loader) { return whisper_init_with_params_no_state(loader, whisper_context_default_params()); }

struct whisper_state * whisper_init_state(struct whisper_context * ctx) {
    if (!ctx) return nullptr;
    if (ctx->host_only) { WMI_ERR("%s: host-only context has no compute path (this backend has no CPU fallback)\n", __func__); return nullptr; }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    State * st = create_state(*ctx);
    if (!st) WMI_ERR("%s: failed to allocate the state on device %d\n", __func__, ctx->device);
    return reinterpret_cast<struct whisper_state *>(st);
}
void whisper_free_state(struct whisper_state * state) { destroy_state(S(state)); }        // NULL-safe, W/whisper.cpp:3340-3368

// not built with OpenVINO: the reference's answer in that configuration (W/whisper.cpp:3122-3134)
int whisper_ctx_init_openvino_encoder(struct whisper_context *, const char *, const char *, const char *) { return 1; }

struct whisper_context_params * whisper_context_default_params_by_ref(void) { return new whisper_context_params(whisper_context_default_params()); }
struct whisper_full_params * whisper_full_default_params_by_ref(enum whisper_sampling_strategy strategy) { return new whisper_full_params(whisper_full_default_params(strategy)); }
void whisper_free_context_params(struct whisper_context_params * params) { delete params; }
void whisper_free_params(struct whisper_full_params * params) { delete params; }

// ---------------------------------------------------------------------------------------------- compute on a caller-owned state
int whisper_pcm_to_mel_with_state(struct whisper_context * ctx, struct whisper_state * state, const float * samples, int n_samples, int) {
    if (!ctx || !state) return -1;
    StateScope sc(ctx, state);
    if (!pcm_to_mel(*ctx, samples, n_samples, false)) { WMI_ERR("%s: failed to compute mel spectrogram\n", __func__); return -1; }
    return 0;
}
// The reference's x2 phase-vocoder variant calls the mel routine with an 800-sample frame, whose 401 bins index the
// 201-bin filterbank out of bounds (W/whisper.cpp:3417-3425 -> :2764-2776) and which whisper_full itself refuses
// (:4973-4976).  There is no defined result to reproduce: fail the way whisper_full does.
int whisper_pcm_to_mel_phase_vocoder_with_state(struct whisper_context *, struct whisper_state *, const float *, int, int) {
    WMI_ERR("%s: failed to compute mel spectrogram (the x2 phase-vocoder front end is not supported)\n", __func__);
    return -1;
}
int whisper_pcm_to_mel_phase_vocoder(struct whisper_context * ctx, const float * samples, int n_samples, int n_threads) {
    return whisper_pcm_to_mel_phase_vocoder_with_state(ctx, ctx ? reinterpret_cast<struct whisper_state *>(ctx->state.get()) : nullptr, samples, n_samples, n_threads);
}
int whisper_set_mel_with_state(struct whisper_context * ctx, struct whisper_state * state, const float * data, int n_len, int n_mel) {
    if (!ctx || !state) return -1;
    if (n_mel != ctx->model.n_filt_mel) { WMI_ERR("%s: invalid number of mel bands: %d (expected %d)\n", __func__, n_mel, ctx->model.n_filt_mel); return -1; }
    StateScope sc(ctx, state);
    return set_mel(*ctx, data, n_len, n_mel) ? 0 : -1;
}
int whisper_encode_with_state(struct whisper_context * ctx, struct whisper_state * state, int offset, int) {
    if (!ctx || !state) return -1;
    StateScope sc(ctx, state);
    if (!encode(*ctx, offset)) { WMI_ERR("%s: failed to eval\n", __func__); return -1; }
    return 0;
}
int whisper_decode_with_state(struct whisper_context * ctx, struct whisper_state * state, const whisper_token * tokens, int n_tokens, int n_past, int) {
    if (!ctx || !state) { WMI_ERR("%s: ERROR state was not loaded.\n", __func__); return -1; }
    StateScope sc(ctx, state);
    State & st = *ctx->state;
    st.batch.prep_legacy(tokens, n_tokens, n_past, 0);
    kv_seq_rm(st.kv_self, 0, n_past, -1);
    if (!decode(*ctx, st.batch)) { WMI_ERR("%s: failed to eval\n", __func__); return 1; }
    return 0;
}
int whisper_lang_auto_detect_with_state(struct whisper_context * ctx, struct whisper_state * state, int offset_ms, int, float * lang_probs) {
    if (!ctx || !state) return -1;
    StateScope sc(ctx, state);
    return lang_auto_detect(*ctx, offset_ms, lang_probs);
}
int whisper_full_with_state(struct whisper_context * ctx, struct whisper_state * state, struct whisper_full_params params, const float * samples, int n_samples) {
    if (!ctx || !state) return -1;
    StateScope sc(ctx, state);
    return full(*ctx, params, samples, nullptr, n_samples);
}

// W/whisper.cpp:5817-5924.  The split, the per-piece parameters, the time offsets, the no-overlap clamp, the callback
// replay, the timing bookkeeping AND the threads are the reference's: pieces 1.. run on host threads of their own, each on
// its own state (own stream), piece 0 on the caller's thread.
int whisper_full_parallel(struct whisper_context * ctx, struct whisper_full_params params, const float * samples, int n_samples, int n_processors) {
    if (n_processors == 1) return whisper_full(ctx, params, samples, n_samples);
    if (!ctx || !ctx->state || n_processors < 1) return -1;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    const int offset_samples = (WHISPER_SAMPLE_RATE * params.offset_ms) / 1000;
    const int per = (n_samples - offset_samples) / n_processors;
    // nothing to split (the reference would hand negative sample counts to its workers here): one plain call
    if (per <= 0) return whisper_full(ctx, params, samples, n_samples);

    std::vector<struct whisper_state *> states;
    for (int i = 0; i < n_processors - 1; ++i) {
        struct whisper_state * st = whisper_init_state(ctx);
        if (!st) { for (auto * s : states) whisper_free_state(s); return -1; }
        states.push_back(st);
    }
    // the pieces 1.. on threads of their own, piece 0 on this one (W/whisper.cpp:5837-5858): every piece computes on its own state, under that
    // state's lock, on that state's stream (round 6; before, the pieces took turns)
    int ret;
    {
        std::vector<std::thread> workers;
        struct Joiner { std::vector<std::thread> & t; ~Joiner() { for (auto & x : t) if (x.joinable()) x.join(); } } joiner{workers};
        static const bool serial = getenv("WMI_PARALLEL_SERIAL") != nullptr;      // debug / A-B: one piece after the other
        for (int i = 0; i < n_processors - 1; ++i) {
            const int start = offset_samples + (i + 1) * per;
            const int n_cur = (i == n_processors - 2) ? n_samples - start : per;
            auto cur = params;
            cur.offset_ms = 0;
            cur.print_progress = false; cur.print_realtime = false;
            cur.new_segment_callback = nullptr; cur.new_segment_callback_user_data = nullptr;
            cur.progress_callback = nullptr;    cur.progress_callback_user_data = nullptr;
            struct whisper_state * st = states[i];
            auto piece = [ctx, st, cur, samples, start, n_cur]() { (void) whisper_full_with_state(ctx, st, cur, samples + start, n_cur); };   // the reference drops the workers' return codes too
            if (serial) piece(); else workers.emplace_back(piece);
        }
        auto cur = params;
        cur.print_realtime = false;
        ret = whisper_full_with_state(ctx, reinterpret_cast<struct whisper_state *>(ctx->state.get()), cur, samples, offset_samples + per);
    }

    const int64_t offset_t = (int64_t) (params.offset_ms / 10.0);
    State & main = *ctx->state;
    for (int i = 0; i < n_processors - 1; ++i) {
        State & si = *S(states[i]);
        for (auto & seg : si.result_all) {
            const int64_t shift = 100 * (int64_t) ((i + 1) * per) / WHISPER_SAMPLE_RATE + offset_t;
            seg.t0 += shift; seg.t1 += shift;
            if (!main.result_all.empty()) seg.t0 = std::max(seg.t0, main.result_all.back().t1);
            main.result_all.push_back(std::move(seg));
            if (params.new_segment_callback)
                params.new_segment_callback(ctx, reinterpret_cast<struct whisper_state *>(ctx->state.get()), 1, params.new_segment_callback_user_data);
        }
        main.t_mel_us += si.t_mel_us; main.t_sample_us += si.t_sample_us; main.t_encode_us += si.t_encode_us;
        main.t_decode_us += si.t_decode_us; main.t_batchd_us += si.t_batchd_us; main.t_prompt_us += si.t_prompt_us;
        main.n_sample += si.n_sample; main.n_encode += si.n_encode; main.n_decode += si.n_decode;
        main.n_batchd += si.n_batchd; main.n_prompt += si.n_prompt;
        whisper_free_state(states[i]);
    }
    main.t_mel_us /= n_processors; main.t_sample_us /= n_processors; main.t_encode_us /= n_processors; main.t_decode_us /= n_processors;

    WMI_WARN("\n");
    WMI_WARN("%s: the audio has been split into %d chunks at the following times:\n", __func__, n_processors);
    for (int i = 0; i < n_processors - 1; ++i) {
        const int64_t t = 100 * (int64_t) ((i + 1) * per) / WHISPER_SAMPLE_RATE + offset_t;       // 10 ms units
        const int64_t msec = t * 10, hr = msec / 3600000, mn = (msec / 60000) % 60, sec = (msec / 1000) % 60;
        WMI_WARN("%s: split %d - %02d:%02d:%02d.%03d\n", __func__, i + 1, (int) hr, (int) mn, (int) sec, (int) (msec % 1000));
    }
    WMI_WARN("%s: the transcription quality may be degraded near these boundaries\n", __func__);
    return ret;
}

// ---------------------------------------------------------------------------------------------- results of a state
int whisper_n_len_from_state(struct whisper_state * state) { return S(state)->mel.n_len_org; }
float * whisper_get_logits_from_state(struct whisper_state * state) { return S(state)->logits.data(); }
int whisper_full_n_segments_from_state(struct whisper_state * state) { return (int) S(state)->result_all.size(); }
int whisper_full_lang_id_from_state(struct whisper_state * state) { return S(state)->lang_id; }
int64_t whisper_full_get_segment_t0_from_state(struct whisper_state * state, int i) { return S(state)->result_all[i].t0; }
int64_t whisper_full_get_segment_t1_from_state(struct whisper_state * state, int i) { return S(state)->result_all[i].t1; }
bool whisper_full_get_segment_speaker_turn_next_from_state(struct whisper_state * state, int i) { return S(state)->result_all[i].speaker_turn_next; }
bool whisper_full_get_segment_speaker_turn_next(struct whisper_context * ctx, int i) { return ctx->state->result_all[i].speaker_turn_next; }
const char * whisper_full_get_segment_text_from_state(struct whisper_state * state, int i) { return S(state)->result_all[i].text.c_str(); }
int whisper_full_n_tokens_from_state(struct whisper_state * state, int i) { return (int) S(state)->result_all[i].tokens.size(); }
const char * whisper_full_get_token_text_from_state(struct whisper_context * ctx, struct whisper_state * state, int i, int j) {
    return ctx->model.vocab.id_to_token[S(state)->result_all[i].tokens[j].id].c_str();
}
whisper_token whisper_full_get_token_id_from_state(struct whisper_state * state, int i, int j) { return S(state)->result_all[i].tokens[j].id; }
whisper_token_data whisper_full_get_token_data_from_state(struct whisper_state * state, int i, int j) { return S(state)->result_all[i].tokens[j]; }
float whisper_full_get_token_p_from_state(struct whisper_state * state, int i, int j) { return S(state)->result_all[i].tokens[j].p; }

const char * whisper_lang_str_full(int id) { return lang_str_full(id); }

// ---------------------------------------------------------------------------------------------- bench entry points
// The reference times host memcpy and ggml_mul_mat on n_threads cores (W/whisper.cpp:6027-6266).  The drop-in reports
// the same two quantities for what does the work here: device-to-device copy bandwidth in HBM, and the f16 MFMA GEMM
// of the encoder (k_gemm, every epilogue writes f16) on the same square sizes.  n_threads is accepted and ignored.
const char * whisper_bench_memcpy_str(int) {
    static std::string s;
    s.clear();
    char line[256];
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { s = "memcpy: no HIP device available\n"; return s.c_str(); }
    const size_t size = 1024ull * 1000 * 1000;       // the reference's 1 GB array
    char * src = nullptr, * dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMalloc(&src, size) != hipSuccess || hipMalloc(&dst, size) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        s = "memcpy: device allocation failed\n";
    } else {
        (void) hipMemset(src, 1, size);
        (void) hipMemcpy(dst, src, size, hipMemcpyDeviceToDevice);           // heat-up
        const int n = 20;
        (void) hipEventRecord(e0, nullptr);
        for (int i = 0; i < n; ++i) (void) hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToDevice, nullptr);
        (void) hipEventRecord(e1, nullptr);
        (void) hipEventSynchronize(e1);
        float ms = 0.0f; (void) hipEventElapsedTime(&ms, e0, e1);
        snprintf(line, sizeof(line), "memcpy: %7.2f GB/s (HBM, device to device; read + write traffic is twice that)\n", (double) n * size / (ms * 1e6));
        s += line;
        unsigned char probe[64] = {0};
        (void) hipMemcpy(probe, dst + size - 64, 64, hipMemcpyDeviceToHost);
        double sum = 0.0; for (unsigned char c : probe) sum += c;
        snprintf(line, sizeof(line), "sum:    %f\n", sum * (double) (size / 64));
        s += line;
    }
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    if (src) (void) hipFree(src);
    if (dst) (void) hipFree(dst);
    return s.c_str();
}
int whisper_bench_memcpy(int n_threads) { fputs(whisper_bench_memcpy_str(n_threads), stderr); return 0; }

const char * whisper_bench_ggml_mul_mat_str(int) {
    static std::string s;
    s.clear();
    char line[256];
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { s = "mul_mat: no HIP device available\n"; return s.c_str(); }
    const size_t sizes[] = { 64, 128, 256, 512, 1024, 2048, 4096 };
    const size_t N_max = 4096;
    __half * a = nullptr, * b = nullptr, * c = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMalloc(&a, N_max * N_max * 2) != hipSuccess || hipMalloc(&b, N_max * N_max * 2) != hipSuccess || hipMalloc(&c, N_max * N_max * 2) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        s = "mul_mat: device allocation failed\n";
    } else {
        (void) hipMemset(a, 0x3c, N_max * N_max * 2);         // f16 0x3c3c = 1.0586: finite operands
        (void) hipMemset(b, 0x3c, N_max * N_max * 2);
        for (const size_t N : sizes) {
            k::GemmArgs g{};
            g.A = a; g.lda = (int) N; g.W = b; g.ldw = (int) N; g.M = g.N = g.K = (int) N; g.bias = nullptr; g.C = c; g.ldc = (int) N;
            k::gemm(k::EPI_F16_BIAS, g, nullptr);             // heat-up
            (void) hipStreamSynchronize(nullptr);
            int n = 0; float ms = 0.0f;
            for (int rep = 8; rep <= 1024 && ms < 20.0f; rep *= 2) {       // grow the run until it lasts 20 ms (the reference stops at 1 s)
                (void) hipEventRecord(e0, nullptr);
                for (int i = 0; i < rep; ++i) k::gemm(k::EPI_F16_BIAS, g, nullptr);
                (void) hipEventRecord(e1, nullptr);
                (void) hipEventSynchronize(e1);
                (void) hipEventElapsedTime(&ms, e0, e1);
                n = rep;
            }
            const double gflops = 2.0 * N * N * N * n / (ms * 1e6);
            snprintf(line, sizeof(line), "%4zu x %4zu: F16  %9.1f GFLOPS (%4d runs) | MFMA f16 x f16 -> f32 accumulate, f16 out\n", N, N, gflops, n);
            s += line;
        }
    }
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    if (a) (void) hipFree(a);
    if (b) (void) hipFree(b);
    if (c) (void) hipFree(c);
    return s.c_str();
}
int whisper_bench_ggml_mul_mat(int n_threads) { fputs(whisper_bench_ggml_mul_mat_str(n_threads), stderr); return 0; }

} // extern "C"
