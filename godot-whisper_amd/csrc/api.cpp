// C ABI of libwhisper_mi355.so: the whisper.h subset (include/whisper_mi355.h) and the device-level
// entry points (include/wmi_device.h).  Nothing throws across the boundary; errors are return codes
// plus the process-global log callback, as in the reference (W/whisper.cpp:6601-6629).

#include <algorithm>
#include "wmi.h"
#include "kernels.h"

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <exception>
#include <fstream>

namespace wmi {

namespace {
void default_log(ggml_log_level, const char * text, void *) { fputs(text, stderr); fflush(stderr); }
ggml_log_callback g_log_cb = default_log;
void * g_log_ud = nullptr;
// one callback invocation at a time: replica contexts and pool workers log from library-created threads, a host's callback (Godot's
// print, src/register_types.cpp:34-58) was written for one caller.  Recursive: a callback that logs does not deadlock itself.
std::recursive_mutex g_log_mu;
}

void log_msg(ggml_log_level lvl, const char * fmt, ...) {
    va_list ap, ap2;
    va_start(ap, fmt);
    va_copy(ap2, ap);
    char buf[1024];
    const int len = vsnprintf(buf, sizeof(buf), fmt, ap);
    std::lock_guard<std::recursive_mutex> lk(g_log_mu);
    if (len < (int) sizeof(buf)) g_log_cb(lvl, buf, g_log_ud);
    else { std::vector<char> big(len + 1); vsnprintf(big.data(), big.size(), fmt, ap2); g_log_cb(lvl, big.data(), g_log_ud); }
    va_end(ap2);
    va_end(ap);
}

bool hip_ok(hipError_t e, const char * what, const char * file, int line) {
    if (e == hipSuccess) return true;
    log_msg(GGML_LOG_LEVEL_ERROR, "HIP error %d (%s) at %s:%d: %s\n", (int) e, hipGetErrorString(e), file, line, what);
    return false;
}

int64_t time_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool compute_ready(const whisper_context & ctx, const char * who) {
    if (ctx.host_only) { WMI_ERR("%s: host-only context has no compute path (this backend has no CPU fallback)\n", who); return false; }
    if (ctx.weights_pending) { WMI_ERR("%s: the weight arena of this context has not been filled yet (wmi_init_from_header without wmi_arena_commit)\n", who); return false; }
    return true;
}

whisper_context * init_context(const void * buffer, size_t size, int device, bool with_state, bool allow_header) {
    const int64_t t0 = time_us();
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        // the product has no CPU path: fail loudly instead of silently degrading
        WMI_ERR("whisper_mi355: no HIP device available - this backend requires an AMD GPU (gfx950)\n");
        return nullptr;
    }
    if (device < 0 || device >= n_dev) { WMI_ERR("whisper_mi355: invalid device %d (have %d)\n", device, n_dev); return nullptr; }
    if (!buffer || size < 4) { WMI_ERR("%s: empty model buffer\n", __func__); return nullptr; }
    whisper_context * ctx = nullptr;
    // nothing may throw across the C boundary: an allocation failure on a hostile size is a load error like any other
    try {
        ctx = new whisper_context();
        ctx->device = device;
        ctx->t_start_us = t0;
        WMI_INFO("%s: loading model from buffer\n", __func__);
        if (!parse_model((const uint8_t *) buffer, size, ctx->model)) { WMI_ERR("%s: failed to load model\n", __func__); delete ctx; return nullptr; }
        if (ctx->model.directory_only && !allow_header) {
            WMI_ERR("%s: this buffer is a header image without tensor payloads - load it with wmi_init_from_header and fill the arena\n", __func__);
            delete ctx; return nullptr;
        }
        ctx->weights_pending = ctx->model.directory_only;
        if (!HIP_OK(hipSetDevice(device))) { delete ctx; return nullptr; }
        if (with_state && !init_state(*ctx)) { free_state(*ctx); delete ctx; return nullptr; }
        hipStream_t ls = ctx->state ? ctx->state->dev.stream : nullptr;
        if (!upload_weights(ctx->model, ctx->model.directory_only ? nullptr : (const uint8_t *) buffer, ctx->w, ls)) {
            WMI_ERR("%s: failed to load model\n", __func__);
            free_state(*ctx); free_weights(ctx->w); delete ctx; return nullptr;
        }
    } catch (const std::exception & e) {
        WMI_ERR("%s: failed to load model (%s)\n", __func__, e.what());
        if (ctx) { free_state(*ctx); free_weights(ctx->w); delete ctx; }
        return nullptr;
    }
    ctx->t_load_us = time_us() - t0;
    return ctx;
}

} // namespace wmi

using namespace wmi;

extern "C" {

// ------------------------------------------------------------------ [host] entry points
struct whisper_context * whisper_init_from_buffer_with_params(void * buffer, size_t buffer_size, struct whisper_context_params params) {
    // params.use_gpu is honoured trivially: there is only a GPU path (the Godot setting defaults to true)
    whisper_context * ctx = init_context(buffer, buffer_size, 0, true);
    if (ctx) ctx->params = params;
    return ctx;
}

void whisper_free(struct whisper_context * ctx) {
    if (!ctx) return;
    if (ctx->host_only) { delete ctx->state; delete ctx; return; }
    (void) hipSetDevice(ctx->device);
    free_batch(*ctx);
    free_state(*ctx);
    for (hipStream_t sp : ctx->spare_streams) own_queue_stream_put(ctx->device, sp);
    ctx->spare_streams.clear();
    free_weights(ctx->w);
    for (float * t : ctx->d_sinc) if (t) (void) hipFree(t);
    if (ctx->vad_res) (void) hipHostFree(ctx->vad_res);
    if (ctx->dsp_scratch) (void) hipFree(ctx->dsp_scratch);
    delete ctx;
}

const char * whisper_print_system_info(void) {
    static std::string s;
    int n_dev = 0; (void) hipGetDeviceCount(&n_dev);
    hipDeviceProp_t p{};
    std::string arch = "none", name = "none"; int cus = 0;
    if (n_dev > 0 && hipGetDeviceProperties(&p, 0) == hipSuccess) { arch = p.gcnArchName; name = p.name; cus = p.multiProcessorCount; }
    s = "HIP = 1 | DEVICES = " + std::to_string(n_dev) + " | ARCH = " + arch + " | CU = " + std::to_string(cus) +
        " | NAME = " + name + " | MFMA_F16 = 1 | CPU_FALLBACK = 0 | ";
    return s.c_str();
}

struct whisper_full_params whisper_full_default_params(enum whisper_sampling_strategy strategy) {   // W/whisper.cpp:4311-4410
    struct whisper_full_params p;
    memset(&p, 0, sizeof(p));
    p.strategy = strategy;
    p.n_threads = 4;  p.n_max_text_ctx = 16384;
    p.no_context = true;  p.print_progress = true;  p.print_timestamps = true;
    p.thold_pt = 0.01f;  p.thold_ptsum = 0.01f;
    p.language = "en";
    p.suppress_blank = true;
    p.temperature = 0.0f;  p.max_initial_ts = 1.0f;  p.length_penalty = -1.0f;
    p.temperature_inc = 0.2f;  p.entropy_thold = 2.4f;  p.logprob_thold = -1.0f;  p.no_speech_thold = 0.6f;
    p.greedy.best_of = -1;  p.beam_search.beam_size = -1;  p.beam_search.patience = -1.0f;
    p.grammar_penalty = 100.0f;
    if (strategy == WHISPER_SAMPLING_GREEDY) p.greedy.best_of = 5;
    else if (strategy == WHISPER_SAMPLING_BEAM_SEARCH) { p.beam_search.beam_size = 5; p.beam_search.patience = -1.0f; }
    return p;
}

// the context's own state is one more whisper_state (api_state.cpp), as in the reference (W/whisper.cpp:5809-5815)
// An entry point that works on the CONTEXT (lock-step work set, DSP scratch, probes) and, through it, on the context's own state: the context's
// lock, then that state's (the order every path takes them in; a *_with_state call holds a state's lock only and never asks for the context's)
struct CtxScope {
    std::unique_lock<std::recursive_mutex> lk, lks;
    explicit CtxScope(struct whisper_context * c) : lk(c->mu) { if (State * st = c->state.own) lks = std::unique_lock<std::recursive_mutex>(st->mu); }
};
static inline struct whisper_state * own_state(struct whisper_context * ctx) { return ctx ? reinterpret_cast<struct whisper_state *>(ctx->state.get()) : nullptr; }

int whisper_full(struct whisper_context * ctx, struct whisper_full_params params, const float * samples, int n_samples) {
    return whisper_full_with_state(ctx, own_state(ctx), params, samples, n_samples);
}

int whisper_full_n_segments(struct whisper_context * ctx) { return (int) ctx->state->result_all.size(); }
int whisper_full_n_tokens(struct whisper_context * ctx, int i) { return (int) ctx->state->result_all[i].tokens.size(); }
const char * whisper_full_get_segment_text(struct whisper_context * ctx, int i) { return ctx->state->result_all[i].text.c_str(); }
const char * whisper_full_get_token_text(struct whisper_context * ctx, int i, int j) {
    return ctx->model.vocab.id_to_token[ctx->state->result_all[i].tokens[j].id].c_str();
}
whisper_token_data whisper_full_get_token_data(struct whisper_context * ctx, int i, int j) { return ctx->state->result_all[i].tokens[j]; }

void whisper_log_set(ggml_log_callback cb, void * user_data) {
    std::lock_guard<std::recursive_mutex> lk(g_log_mu);
    g_log_cb = cb ? cb : default_log; g_log_ud = user_data;
}

// ------------------------------------------------------------------ rest of the whisper.h subset
struct whisper_context_params whisper_context_default_params(void) { struct whisper_context_params p = { true }; return p; }

struct whisper_context * whisper_init_from_file_with_params(const char * path, struct whisper_context_params params) {
    WMI_INFO("%s: loading model from '%s'\n", __func__, path);
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { WMI_ERR("%s: failed to open '%s'\n", __func__, path); return nullptr; }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<char> buf((size_t) n);
    if (!f.read(buf.data(), n)) { WMI_ERR("%s: failed to read '%s'\n", __func__, path); return nullptr; }
    return whisper_init_from_buffer_with_params(buf.data(), buf.size(), params);
}

int whisper_pcm_to_mel(struct whisper_context * ctx, const float * samples, int n_samples, int n_threads) {
    return whisper_pcm_to_mel_with_state(ctx, own_state(ctx), samples, n_samples, n_threads);
}
int whisper_set_mel(struct whisper_context * ctx, const float * data, int n_len, int n_mel) {
    return whisper_set_mel_with_state(ctx, own_state(ctx), data, n_len, n_mel);
}
int whisper_encode(struct whisper_context * ctx, int offset, int n_threads) {
    return whisper_encode_with_state(ctx, own_state(ctx), offset, n_threads);
}
int whisper_decode(struct whisper_context * ctx, const whisper_token * tokens, int n_tokens, int n_past, int n_threads) {
    return whisper_decode_with_state(ctx, own_state(ctx), tokens, n_tokens, n_past, n_threads);
}
int whisper_tokenize(struct whisper_context * ctx, const char * text, whisper_token * tokens, int n_max_tokens) {
    const auto res = tokenize(ctx->model.vocab, text);
    if (n_max_tokens < (int) res.size()) { WMI_ERR("%s: too many resulting tokens: %d (max %d)\n", __func__, (int) res.size(), n_max_tokens); return -1; }
    for (size_t i = 0; i < res.size(); ++i) tokens[i] = res[i];
    return (int) res.size();
}
float * whisper_get_logits(struct whisper_context * ctx) { return ctx->state->logits.data(); }

int whisper_lang_max_id(void) { return lang_max_id(); }
int whisper_lang_id(const char * lang) { return lang_id(lang); }
const char * whisper_lang_str(int id) { return lang_str(id); }
int whisper_lang_auto_detect(struct whisper_context * ctx, int offset_ms, int n_threads, float * lang_probs) {
    return whisper_lang_auto_detect_with_state(ctx, own_state(ctx), offset_ms, n_threads, lang_probs);
}

int whisper_n_len(struct whisper_context * ctx) { return ctx->state->mel.n_len_org; }
int whisper_n_vocab(struct whisper_context * ctx) { return ctx->model.vocab.n_vocab; }
int whisper_n_text_ctx(struct whisper_context * ctx) { return ctx->model.hp.n_text_ctx; }
int whisper_n_audio_ctx(struct whisper_context * ctx) { return ctx->model.hp.n_audio_ctx; }
int whisper_is_multilingual(struct whisper_context * ctx) { return ctx->model.vocab.is_multilingual() ? 1 : 0; }
int whisper_model_n_vocab(struct whisper_context * ctx) { return ctx->model.hp.n_vocab; }
int whisper_model_n_audio_ctx(struct whisper_context * ctx) { return ctx->model.hp.n_audio_ctx; }
int whisper_model_n_audio_state(struct whisper_context * ctx) { return ctx->model.hp.n_audio_state; }
int whisper_model_n_audio_head(struct whisper_context * ctx) { return ctx->model.hp.n_audio_head; }
int whisper_model_n_audio_layer(struct whisper_context * ctx) { return ctx->model.hp.n_audio_layer; }
int whisper_model_n_text_ctx(struct whisper_context * ctx) { return ctx->model.hp.n_text_ctx; }
int whisper_model_n_text_state(struct whisper_context * ctx) { return ctx->model.hp.n_text_state; }
int whisper_model_n_text_head(struct whisper_context * ctx) { return ctx->model.hp.n_text_head; }
int whisper_model_n_text_layer(struct whisper_context * ctx) { return ctx->model.hp.n_text_layer; }
int whisper_model_n_mels(struct whisper_context * ctx) { return ctx->model.hp.n_mels; }
int whisper_model_ftype(struct whisper_context * ctx) { return ctx->model.hp.ftype; }
int whisper_model_type(struct whisper_context * ctx) { return ctx->model.model_type; }
const char * whisper_model_type_readable(struct whisper_context * ctx) {
    static const char * names[] = { "unknown", "tiny", "base", "small", "medium", "large" };
    const int t = ctx->model.model_type;
    return names[t >= 0 && t <= 5 ? t : 0];
}
const char * whisper_token_to_str(struct whisper_context * ctx, whisper_token token) { return ctx->model.vocab.id_to_token.at(token).c_str(); }
whisper_token whisper_token_eot(struct whisper_context * ctx) { return ctx->model.vocab.eot; }
whisper_token whisper_token_sot(struct whisper_context * ctx) { return ctx->model.vocab.sot; }
whisper_token whisper_token_solm(struct whisper_context * ctx) { return ctx->model.vocab.solm; }
whisper_token whisper_token_prev(struct whisper_context * ctx) { return ctx->model.vocab.prev; }
whisper_token whisper_token_nosp(struct whisper_context * ctx) { return ctx->model.vocab.nosp; }
whisper_token whisper_token_not(struct whisper_context * ctx) { return ctx->model.vocab.not_; }
whisper_token whisper_token_beg(struct whisper_context * ctx) { return ctx->model.vocab.beg; }
whisper_token whisper_token_lang(struct whisper_context * ctx, int lang_id) { return ctx->model.vocab.sot + 1 + lang_id; }
whisper_token whisper_token_translate(struct whisper_context * ctx) { return ctx->model.vocab.translate; }
whisper_token whisper_token_transcribe(struct whisper_context * ctx) { return ctx->model.vocab.transcribe; }
int whisper_full_lang_id(struct whisper_context * ctx) { return ctx->state->lang_id; }
int64_t whisper_full_get_segment_t0(struct whisper_context * ctx, int i) { return ctx->state->result_all[i].t0; }
int64_t whisper_full_get_segment_t1(struct whisper_context * ctx, int i) { return ctx->state->result_all[i].t1; }
whisper_token whisper_full_get_token_id(struct whisper_context * ctx, int i, int j) { return ctx->state->result_all[i].tokens[j].id; }
float whisper_full_get_token_p(struct whisper_context * ctx, int i, int j) { return ctx->state->result_all[i].tokens[j].p; }

void whisper_print_timings(struct whisper_context * ctx) {
    const int64_t t_end = time_us();
    WMI_INFO("\n");
    WMI_INFO("%s:     load time = %8.2f ms\n", __func__, ctx->t_load_us / 1000.0f);
    if (ctx->state) {
        const State & s = *ctx->state;
        const int n_sample = std::max(1, s.n_sample), n_encode = std::max(1, s.n_encode), n_decode = std::max(1, s.n_decode),
                  n_batchd = std::max(1, s.n_batchd), n_prompt = std::max(1, s.n_prompt);
        WMI_INFO("%s:     fallbacks = %3d p / %3d h\n", __func__, s.n_fail_p, s.n_fail_h);
        WMI_INFO("%s:      mel time = %8.2f ms\n", __func__, s.t_mel_us / 1000.0f);
        WMI_INFO("%s:   sample time = %8.2f ms / %5d runs (%8.2f ms per run)\n", __func__, 1e-3f * s.t_sample_us, n_sample, 1e-3f * s.t_sample_us / n_sample);
        WMI_INFO("%s:   encode time = %8.2f ms / %5d runs (%8.2f ms per run)\n", __func__, 1e-3f * s.t_encode_us, n_encode, 1e-3f * s.t_encode_us / n_encode);
        WMI_INFO("%s:   decode time = %8.2f ms / %5d runs (%8.2f ms per run)\n", __func__, 1e-3f * s.t_decode_us, n_decode, 1e-3f * s.t_decode_us / n_decode);
        WMI_INFO("%s:   batchd time = %8.2f ms / %5d runs (%8.2f ms per run)\n", __func__, 1e-3f * s.t_batchd_us, n_batchd, 1e-3f * s.t_batchd_us / n_batchd);
        WMI_INFO("%s:   prompt time = %8.2f ms / %5d runs (%8.2f ms per run)\n", __func__, 1e-3f * s.t_prompt_us, n_prompt, 1e-3f * s.t_prompt_us / n_prompt);
    }
    WMI_INFO("%s:    total time = %8.2f ms\n", __func__, (t_end - ctx->t_start_us) / 1000.0f);
}
void whisper_reset_timings(struct whisper_context * ctx) {
    ctx->t_start_us = time_us();
    if (!ctx->state) return;
    State & s = *ctx->state;
    s.t_mel_us = s.t_sample_us = s.t_encode_us = s.t_decode_us = s.t_batchd_us = s.t_prompt_us = 0;
    s.n_sample = s.n_encode = s.n_decode = s.n_batchd = s.n_prompt = 0;
}

// ------------------------------------------------------------------ device-level ABI (include/wmi_device.h)
int wmi_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
const char * wmi_version(void) { return "whisper_mi355 0.1 (gfx950, whisper.cpp v1.5.4 ABI)"; }

struct whisper_context * wmi_init_from_buffer_on_device(const void * buffer, size_t buffer_size, int device) {
    whisper_context * ctx = init_context(buffer, buffer_size, device, true);
    if (ctx) ctx->params.use_gpu = true;
    return ctx;
}

struct whisper_context * wmi_init_from_header(const void * header, size_t header_size, int device) {
    whisper_context * ctx = init_context(header, header_size, device, true, true);
    if (ctx && !ctx->model.directory_only) {          // a full model is not a header image: one meaning per entry point
        WMI_ERR("%s: the buffer holds tensor payloads - use wmi_init_from_buffer_on_device\n", __func__);
        whisper_free(ctx); return nullptr;
    }
    if (ctx) ctx->params.use_gpu = true;
    return ctx;
}
int wmi_arena_commit(struct whisper_context * ctx) {
    if (!ctx || ctx->host_only || !ctx->w.arena) return -1;
    if (!HIP_OK(hipSetDevice(ctx->device)) || !HIP_OK(hipDeviceSynchronize())) return -2;     // whatever stream filled the arena has finished
    ctx->weights_pending = false;
    return 0;
}
int wmi_weights_pending(struct whisper_context * ctx) { return ctx && ctx->weights_pending ? 1 : 0; }

struct whisper_context * wmi_init_host_only(const void * buffer, size_t buffer_size) {
    if (!buffer || buffer_size < 4) return nullptr;
    whisper_context * ctx = nullptr;
    try {
        ctx = new whisper_context();
        ctx->host_only = true;
        ctx->t_start_us = time_us();
        if (!parse_model((const uint8_t *) buffer, buffer_size, ctx->model)) { WMI_ERR("%s: failed to load model\n", __func__); delete ctx; return nullptr; }
        ctx->state = new State();
        for (auto & dec : ctx->state->decoders) dec.rng = std::mt19937(0);
        if (!plan_weights(ctx->model, ctx->w)) { WMI_ERR("%s: failed to load model\n", __func__); delete ctx->state; delete ctx; return nullptr; }
    } catch (const std::exception & e) {
        WMI_ERR("%s: failed to load model (%s)\n", __func__, e.what());
        delete ctx; return nullptr;
    }
    return ctx;
}

size_t wmi_model_header(const void * model, size_t model_size, void * out, size_t cap) {
    if (!model || model_size < 4) return 0;
    try {
        ModelFile mf;
        if (!parse_model((const uint8_t *) model, model_size, mf) || mf.directory_only) return 0;
        const std::vector<uint8_t> img = export_header(mf, (const uint8_t *) model, 0);
        if (out && cap >= img.size()) memcpy(out, img.data(), img.size());
        return img.size();
    } catch (const std::exception &) { return 0; }
}
void * wmi_arena_ptr(struct whisper_context * ctx) { return ctx ? ctx->w.arena : nullptr; }
size_t wmi_weights_bytes(struct whisper_context * ctx, int which) {
    if (!ctx) return 0;
    return which == 0 ? ctx->w.arena_bytes : which == 1 ? ctx->w.matrix_bytes : which == 2 ? (size_t) ctx->w.qtype : which == 3 ? k::qweights_f16_cached_bytes() : 0;
}

// grow-only device staging for the host-pointer forms (a hipMalloc / hipFree pair per call costs more than the kernels: the VAD call
// was 1.7 ms of which the kernel is 0.3)
static float * dsp_scratch(whisper_context * ctx, size_t bytes) {
    if (ctx->dsp_scratch_bytes < bytes) {
        if (ctx->dsp_scratch) (void) hipFree(ctx->dsp_scratch);
        ctx->dsp_scratch = nullptr; ctx->dsp_scratch_bytes = 0;
        if (!HIP_OK(hipMalloc((void **) &ctx->dsp_scratch, bytes))) return nullptr;
        ctx->dsp_scratch_bytes = bytes;
    }
    return ctx->dsp_scratch;
}

int wmi_downmix_stereo(struct whisper_context * ctx, const float * frames, int n_frames, int on_device, float * mono_out) {
    if (!ctx || !ctx->state || ctx->host_only || !frames || !mono_out || n_frames < 0) return -1;
    CtxScope lk(ctx);
    if (!HIP_OK(hipSetDevice(ctx->device))) return -2;
    hipStream_t s = ctx->state->dev.stream;
    if (n_frames == 0) return 0;
    if (on_device) { k::downmix_stereo(frames, n_frames, mono_out, s); return HIP_OK(hipStreamSynchronize(s)) ? 0 : -3; }
    float * d_in = dsp_scratch(ctx, (size_t) n_frames * 12), * d_out = d_in ? d_in + (size_t) n_frames * 2 : nullptr;
    bool ok = d_in && HIP_OK(hipMemcpyAsync(d_in, frames, (size_t) n_frames * 8, hipMemcpyHostToDevice, s));
    if (ok) k::downmix_stereo(d_in, n_frames, d_out, s);
    ok = ok && HIP_OK(hipMemcpyAsync(mono_out, d_out, (size_t) n_frames * 4, hipMemcpyDeviceToHost, s)) && HIP_OK(hipStreamSynchronize(s));
    return ok ? 0 : -3;
}

int wmi_resample(struct whisper_context * ctx, const float * src, int n_frames, int src_rate, int dst_rate, int converter, int on_device,
                 float * dst, int dst_capacity) {
    if (!ctx || !ctx->state || ctx->host_only || !src || !dst || n_frames < 0 || src_rate <= 0 || dst_rate <= 0 || dst_capacity < 0 ||
        converter < 0 || converter > 2) return -1;
    CtxScope lk(ctx);
    if (!HIP_OK(hipSetDevice(ctx->device))) return -2;
    hipStream_t s = ctx->state->dev.stream;
    if (src_rate == dst_rate) {                                                                    // src/speech_to_text.cpp:38-42
        if (n_frames > dst_capacity) return -4;
        const bool ok = n_frames == 0 || (HIP_OK(hipMemcpyAsync(dst, src, (size_t) n_frames * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToHost, s)) &&
                                          HIP_OK(hipStreamSynchronize(s)));
        return ok ? n_frames : -3;
    }
    const double ratio = (double) (uint32_t) dst_rate / (double) (uint32_t) src_rate;              // :25-26
    const long long out_frames = (int) ((uint32_t) n_frames * ratio);
    if (out_frames > dst_capacity) return -4;
    try {                                                                                          // (the plan allocates: nothing throws across the C ABI)
    const k::ResamplePlan pl = k::resample_plan(n_frames, out_frames, ratio, converter);
    if (pl.error) {
        WMI_ERR("wmi_resample: converter error %d (src_simple would report it through src_strerror)\n", -pl.error);
        return pl.error == -10 ? -10 : 0;                                                          // the host returns 0 frames on a converter error (:33-36)
    }
    if (pl.n_out == 0) return 0;
    float *& d_tab = ctx->d_sinc[converter];
    bool ok = true;
    if (!d_tab) {
        const float * coeffs; int count, inc;
        (void) k::sinc_table(converter, &coeffs, &count, &inc);
        ok = HIP_OK(hipMalloc((void **) &d_tab, (size_t) count * 4)) && HIP_OK(hipMemcpyAsync(d_tab, coeffs, (size_t) count * 4, hipMemcpyHostToDevice, s));
        if (!ok) { if (d_tab) { (void) hipFree(d_tab); d_tab = nullptr; } return -3; }
    }
    float * d_in = nullptr, * d_out = nullptr; int * d_pos = nullptr; double * d_frac = nullptr;
    const float * in = src; float * out = dst;
    if (!on_device) {
        d_in = dsp_scratch(ctx, ((size_t) std::max(n_frames, 1) + (size_t) pl.n_out) * 4);
        d_out = d_in ? d_in + std::max(n_frames, 1) : nullptr;
        ok = d_in && HIP_OK(hipMemcpyAsync(d_in, src, (size_t) n_frames * 4, hipMemcpyHostToDevice, s));
        in = d_in; out = d_out;
    }
    if (ok && pl.need_table) {                                                                     // positions from the host's recurrence (index bookkeeping)
        const int * hp; const double * hf;
        k::resample_table(pl, &hp, &hf);
        ok = HIP_OK(hipMalloc((void **) &d_pos, (size_t) pl.n_out * 4)) && HIP_OK(hipMalloc((void **) &d_frac, (size_t) pl.n_out * 8)) &&
             HIP_OK(hipMemcpyAsync(d_pos, hp, (size_t) pl.n_out * 4, hipMemcpyHostToDevice, s)) &&
             HIP_OK(hipMemcpyAsync(d_frac, hf, (size_t) pl.n_out * 8, hipMemcpyHostToDevice, s));
    }
    if (ok) k::resample_launch(pl, in, n_frames, out, d_tab, d_pos, d_frac, s);
    if (ok && !on_device) ok = HIP_OK(hipMemcpyAsync(dst, d_out, (size_t) pl.n_out * 4, hipMemcpyDeviceToHost, s));
    ok = ok && HIP_OK(hipStreamSynchronize(s));
    (void) hipFree(d_pos); (void) hipFree(d_frac);
    return ok ? (int) pl.n_out : -3;
    } catch (...) { WMI_ERR("wmi_resample: out of memory\n"); return -3; }
}

int wmi_selftest_ts_refine(struct whisper_context * ctx, const float * envelope, int n, const int * s0s1, int n_tok, float * sums, float * thold, int * walks) {
    if (!ctx || !envelope || n <= 0 || !s0s1 || n_tok <= 0 || n_tok > 4096 || !sums || !thold || !walks) return -1;
    CtxScope lk(ctx);
    if (!compute_ready(*ctx, __func__)) return -2;
    try {
        (void) hipSetDevice(ctx->device);
        const size_t nb = (size_t) n / 256 + 2;
        std::vector<float> ext(2 * nb, 0.0f);                     // block minima | maxima, as k_signal_energy writes them
        for (size_t b = 0; b * 256 < (size_t) n; ++b) {
            float lo = INFINITY, hi = -INFINITY;
            for (size_t i = b * 256; i < std::min<size_t>((b + 1) * 256, (size_t) n); ++i) { lo = std::min(lo, envelope[i]); hi = std::max(hi, envelope[i]); }
            ext[b] = lo; ext[nb + b] = hi;
        }
        std::vector<k::TsTok> in(n_tok);
        for (int t = 0; t < n_tok; ++t) {
            const int s0 = s0s1[2 * t], s1 = s0s1[2 * t + 1];
            if (s0 < 0 || s0 >= n || s1 < 0 || s1 >= n) return -1;
            in[t] = k::TsTok{ s0, s1, std::max(s0 - 2000, 0), std::min(s1 + 2000, n), nullptr, 0, 0 };
        }
        float * d_en = nullptr, * d_ext = nullptr; k::TsTok * d_in = nullptr; k::TsOut * d_out = nullptr;
        hipStream_t s = ctx->state->dev.stream;
        bool ok = HIP_OK(hipMalloc((void **) &d_en, (size_t) n * 4)) && HIP_OK(hipMalloc((void **) &d_ext, 2 * nb * 4)) &&
                  HIP_OK(hipMalloc((void **) &d_in, (size_t) n_tok * sizeof(k::TsTok))) && HIP_OK(hipMalloc((void **) &d_out, (size_t) n_tok * sizeof(k::TsOut)));
        std::vector<k::TsOut> out(n_tok);
        if (ok) {
            ok = HIP_OK(hipMemcpyAsync(d_en, envelope, (size_t) n * 4, hipMemcpyHostToDevice, s)) &&
                 HIP_OK(hipMemcpyAsync(d_ext, ext.data(), 2 * nb * 4, hipMemcpyHostToDevice, s)) &&
                 HIP_OK(hipMemcpyAsync(d_in, in.data(), (size_t) n_tok * sizeof(k::TsTok), hipMemcpyHostToDevice, s));
            if (ok) k::ts_refine(d_en, d_ext, d_ext + nb, n, d_in, d_out, n_tok, s);
            ok = ok && HIP_OK(hipMemcpyAsync(out.data(), d_out, (size_t) n_tok * sizeof(k::TsOut), hipMemcpyDeviceToHost, s)) && HIP_OK(hipStreamSynchronize(s));
        }
        if (d_en) (void) hipFree(d_en); if (d_ext) (void) hipFree(d_ext); if (d_in) (void) hipFree(d_in); if (d_out) (void) hipFree(d_out);
        if (!ok) return -3;
        for (int t = 0; t < n_tok; ++t) {
            sums[t] = out[t].sum; thold[t] = out[t].thold;
            int * w = walks + 6 * t;
            w[0] = out[t].e0; w[1] = out[t].e1; w[2] = out[t].w_down_above_s0; w[3] = out[t].w_up_below_s0; w[4] = out[t].w_up_above_s1; w[5] = out[t].w_down_below_s1;
        }
        return 0;
    } catch (const std::exception & e) {
        WMI_ERR("%s: %s\n", __func__, e.what());
        return -9;
    }
}

int wmi_selftest_resample_plan(int n_frames, int src_rate, int dst_rate, int converter, long long * frames_gen, long long * frames_used,
                               int * closed_form, int n_pos, long long * pos, double * frac) {
    if (n_frames < 0 || src_rate <= 0 || dst_rate <= 0 || src_rate == dst_rate || n_pos < 0 || (n_pos > 0 && (!pos || !frac))) return -1;
    const double ratio = (double) (uint32_t) dst_rate / (double) (uint32_t) src_rate;
    const long long out_frames = (int) ((uint32_t) n_frames * ratio);
    try {
        const k::ResamplePlan pl = k::resample_plan(n_frames, out_frames, ratio, converter);
        if (pl.error) return pl.error;
        if (frames_gen) *frames_gen = pl.n_out;
        if (frames_used) *frames_used = pl.n_used;
        if (closed_form) *closed_form = pl.need_table ? 0 : 1;
        if (n_pos > 0) k::resample_positions(pl, std::min<long long>(n_pos, out_frames + 1), pos, frac);
        return 0;
    } catch (...) { return -3; }
}

int wmi_vad(struct whisper_context * ctx, const float * pcm, int n_samples, int on_device, float vad_thold, float freq_thold, float * energies) {
    if (!ctx || !ctx->state || ctx->host_only || !pcm || n_samples < 0) return -1;
    const int n_win = WHISPER_SAMPLE_RATE * 3, n_last = (WHISPER_SAMPLE_RATE * 500) / 1000;      // src/speech_to_text.cpp:381-386
    if (n_samples < n_win) return 0;                                                              // not enough accumulated audio
    CtxScope lk(ctx);
    if (!HIP_OK(hipSetDevice(ctx->device))) return -2;
    hipStream_t s = ctx->state->dev.stream;
    // alpha exactly as the host computes it (:54-56): Math_PI is a double constant, rc is rounded to float
    const float rc = (float) (1.0f / (2.0f * 3.14159265358979323846 * freq_thold));
    const float dt = 1.0f / (float) WHISPER_SAMPLE_RATE;
    const float alpha = dt / (rc + dt);
    if (!ctx->vad_res && !HIP_OK(hipHostMalloc((void **) &ctx->vad_res, 16, hipHostMallocDefault))) return -3;
    bool ok = true;
    const float * win = pcm + (n_samples - n_win);
    if (!on_device) {
        float * d_win = dsp_scratch(ctx, (size_t) n_win * 4);
        ok = d_win && HIP_OK(hipMemcpyAsync(d_win, win, (size_t) n_win * 4, hipMemcpyHostToDevice, s));
        win = d_win;
    }
    volatile float * res = ctx->vad_res;                              // the kernel stores its three results straight into pinned host memory
    if (ok) k::vad_window(win, n_win, n_last, alpha, freq_thold > 0.0f, vad_thold, ctx->vad_res, s);
    ok = ok && HIP_OK(hipStreamSynchronize(s));
    if (!ok) return -3;
    if (energies) { energies[0] = res[1]; energies[1] = res[2]; }
    return res[0] != 0.0f ? 1 : 0;
}

int wmi_pcm_to_mel_device(struct whisper_context * ctx, const float * d_samples, int n_samples) {
    CtxScope lk(ctx);
    (void) hipSetDevice(ctx->device);
    return pcm_to_mel(*ctx, d_samples, n_samples, true) ? 0 : -1;
}

int wmi_full_device_pcm(struct whisper_context * ctx, struct whisper_full_params params, const float * d_samples, int n_samples,
                        const float * h_samples) {
    if (!ctx || !ctx->state) return -1;
    CtxScope lk(ctx);
    (void) hipSetDevice(ctx->device);
    return full(*ctx, params, h_samples, d_samples, n_samples);
}

int wmi_set_audio_ctx(struct whisper_context * ctx, int n_audio_ctx) {
    CtxScope lk(ctx);
    if (n_audio_ctx < 0 || n_audio_ctx > ctx->model.hp.n_audio_ctx) return -5;
    ctx->state->exp_n_audio_ctx = n_audio_ctx;
    return 0;
}

int wmi_mel_dims(struct whisper_context * ctx, int * n_len, int * n_len_org, int * n_mel) {
    CtxScope lk(ctx);
    const Mel & m = ctx->state->mel;
    if (n_len) *n_len = m.n_len; if (n_len_org) *n_len_org = m.n_len_org; if (n_mel) *n_mel = m.n_mel;
    return m.n_len * m.n_mel;
}

int wmi_get_tensor(struct whisper_context * ctx, const char * name, float * dst, int n) {
    CtxScope lk(ctx);
    (void) hipSetDevice(ctx->device);
    State & st = *ctx->state; DeviceState & d = st.dev; const HParams & hp = ctx->model.hp;
    const int S = hp.n_audio_state, T = st.enc_n_ctx > 0 ? st.enc_n_ctx : hp.n_audio_ctx, Lt = hp.n_text_layer;
    const void * src = nullptr; size_t count = 0; bool is_half = false;
    const std::string nm(name);
    if (nm == "energy") {                                    // the |x| envelope of the last transcription with token timestamps (k_signal_energy)
        if (!signal_energy_wait(st)) return -1;
        const int cnt = st.energy_n;
        if (!dst) return cnt;
        if (n > cnt) n = cnt;
        if (n <= 0) return 0;
        if (st.energy) { memcpy(dst, st.energy, (size_t) n * 4); return n; }            // pinned host image
        if (!d.energy || !HIP_OK(hipMemcpy(dst, d.energy, (size_t) n * 4, hipMemcpyDeviceToHost))) return -1;
        return n;
    }
    if (nm == "mel")            { src = d.mel; count = (size_t) st.mel.n_len * st.mel.n_mel; }
    else if (nm == "embd_conv") { src = d.embd_conv; count = (size_t) T * S; }
    else if (nm == "embd_enc")  { src = d.enc_out; count = (size_t) T * S; }
    else if (nm == "enc_x")     { src = d.x; count = (size_t) T * S; }
    else if (nm == "cross_k")   { src = d.kvc_k; count = (size_t) Lt * T * S; is_half = true; }
    else if (nm == "cross_v")   { src = d.kvc_v; count = (size_t) Lt * T * S; is_half = true; }
    else if (nm == "self_k")    { src = st.kv_self.k; count = (size_t) Lt * st.kv_self.size * S; is_half = true; }
    else if (nm == "self_v")    { src = st.kv_self.v; count = (size_t) Lt * st.kv_self.size * S; is_half = true; }
    // lock-step work buffers of the last wmi_full_batch group (debug / tests): [L][rows*T][S] and [rows*T][S]
    else if (nm == "batch_cross_k" && ctx->batch) { src = ctx->batch->kvc_k; count = (size_t) Lt * ctx->batch->enc_rows * ctx->batch->enc_T * S; is_half = true; }
    else if (nm == "batch_cross_v" && ctx->batch) { src = ctx->batch->kvc_v; count = (size_t) Lt * ctx->batch->enc_rows * ctx->batch->enc_T * S; is_half = true; }
    else if (nm == "batch_enc_x" && ctx->batch)   { src = ctx->batch->x; count = (size_t) ctx->batch->enc_rows * ctx->batch->enc_T * S; }
    else return -1;
    if (!dst) return (int) count;
    if (!src) return -1;
    if ((size_t) n > count) n = (int) count;
    if (!HIP_OK(hipStreamSynchronize(d.stream))) return -1;
    if (!is_half) { if (!HIP_OK(hipMemcpy(dst, src, (size_t) n * 4, hipMemcpyDeviceToHost))) return -1; }
    else {
        std::vector<__half> tmp(n);
        if (!HIP_OK(hipMemcpy(tmp.data(), src, (size_t) n * 2, hipMemcpyDeviceToHost))) return -1;
        for (int i = 0; i < n; ++i) dst[i] = __half2float(tmp[i]);
    }
    return n;
}

void wmi_get_timings(struct whisper_context * ctx, int64_t * t6, int32_t * n5) {
    CtxScope lk(ctx);
    const State & s = *ctx->state;
    t6[0] = s.t_mel_us; t6[1] = s.t_encode_us; t6[2] = s.t_decode_us; t6[3] = s.t_batchd_us; t6[4] = s.t_prompt_us; t6[5] = s.t_sample_us;
    n5[0] = s.n_encode; n5[1] = s.n_decode; n5[2] = s.n_batchd; n5[3] = s.n_prompt; n5[4] = s.n_sample;
}

int wmi_full_batch(struct whisper_context * ctx, struct whisper_full_params params, const float * const * pcm, const int * n_samples,
                   int n_chunks, int pcm_on_device) {
    if (!ctx || !ctx->state || !pcm || !n_samples || n_chunks < 0) return -1;
    CtxScope lk(ctx);
    (void) hipSetDevice(ctx->device);
    params.no_context = true;                 // chunks are independent transcriptions
    return full_batch(*ctx, params, pcm, n_samples, n_chunks, pcm_on_device != 0);
}

void wmi_set_lockstep_exact(int on) { k::set_rows_valu(on != 0); k::set_attn_one_group(on != 0); }

int wmi_batch_select(struct whisper_context * ctx, int chunk) {
    if (!ctx) return -1;
    CtxScope lk(ctx);
    if (!ctx->state || !ctx->batch || chunk < 0 || chunk >= (int) ctx->batch->results.size()) return -1;
    ctx->state->result_all = ctx->batch->results[chunk];
    return (int) ctx->state->result_all.size();
}

void wmi_get_batch_timings(struct whisper_context * ctx, int64_t * t4, int32_t * n_steps) {
    t4[0] = t4[1] = t4[2] = t4[3] = 0; *n_steps = 0;
    if (!ctx) return;
    CtxScope lk(ctx);
    if (!ctx->batch) return;
    const BatchWork & b = *ctx->batch;
    t4[0] = b.t_mel_us; t4[1] = b.t_encode_us; t4[2] = b.t_decode_us; t4[3] = b.t_emit_us; *n_steps = b.n_steps;
}

int wmi_batch_chunk_mode(struct whisper_context * ctx, int chunk) {
    if (!ctx) return -1;
    CtxScope lk(ctx);
    if (!ctx->batch || chunk < 0 || chunk >= (int) ctx->batch->redo.size()) return -1;
    return ctx->batch->redo[chunk];
}

int wmi_set_lockstep_groups(struct whisper_context * ctx, int n) {
    if (!ctx) return -2;
    CtxScope lk(ctx);
    try { if (!ctx->batch) ctx->batch = new BatchWork(); } catch (const std::exception &) { return -2; }
    const int prev = ctx->batch->groups_wanted;
    ctx->batch->groups_wanted = n < 0 ? 0 : (n > 4 ? 4 : n);
    return prev;
}

int wmi_set_batch_replicas(struct whisper_context * ctx, int n) {
    if (!ctx) return -2;                                    // (-1 is a valid answer: "the previous setting was the default")
    CtxScope lk(ctx);
    try {
        if (!ctx->batch) ctx->batch = new BatchWork();
    } catch (const std::exception &) { return -2; }
    const int prev = ctx->batch->replicas_wanted;
    ctx->batch->replicas_wanted = n < 0 ? -1 : (n > 15 ? 15 : n);
    if (n >= 0) trim_replicas(*ctx, ctx->batch->replicas_wanted);      // fewer wanted: their states and streams are released now
    // the replicas are made now rather than inside the first call that wants them: state allocation (~0.6 GB each for large-v3) stays out
    // of that call, and their hardware queues exist before the context's other streams do (see init_state: the order matters)
    if (n > 0 && compute_ready(*ctx, __func__)) {
        try { (void) hipSetDevice(ctx->device); (void) ensure_replicas(*ctx, ctx->batch->replicas_wanted); } catch (const std::exception &) {}
    }
    return prev;
}

void * wmi_stream(struct whisper_context * ctx) { return (void *) ctx->state->dev.stream; }

int wmi_process_logits(struct whisper_context * ctx, struct whisper_full_params params, const float * raw_logits,
                       const whisper_token * hist, int n_hist, int has_ts, int seek_delta, float temperature,
                       float * out_logits, float * out_logprobs, float * out_probs) {
    State & st = *ctx->state;
    Decoder & d = st.decoders[0];
    const int nv = ctx->model.vocab.n_vocab;
    st.logits.assign(raw_logits, raw_logits + nv);
    d.i_batch = 0;
    d.sequence.tokens.clear();
    for (int i = 0; i < n_hist; ++i) d.sequence.tokens.push_back(whisper_token_data{ hist[i], 0, 0.f, 0.f, 0.f, 0.f, -1, -1, 0.f });
    d.has_ts = has_ts != 0; d.seek_delta = seek_delta;
    d.grammar = params.grammar_rules ? grammar_init(params.grammar_rules, params.n_grammar_rules, params.i_start_rule) : Grammar{};
    for (int i = 0; i < n_hist; ++i) grammar_accept_token(*ctx, d.grammar, hist[i]);       // parse state after the history
    process_logits(*ctx, d, params, temperature);
    memcpy(out_logits, d.logits.data(), (size_t) nv * 4);
    memcpy(out_logprobs, d.logprobs.data(), (size_t) nv * 4);
    memcpy(out_probs, d.probs.data(), (size_t) nv * 4);
    return nv;
}

// worker-pool self-test (no device needed): reps jobs of n_tasks tasks, each task adds its index + 1 once; returns the total
int64_t wmi_selftest_pool(int n_tasks, int reps) {
    std::vector<int> hits((size_t) std::max(n_tasks, 0), 0);
    int64_t total = 0;
    for (int r = 0; r < reps; ++r) {
        pool_run(n_tasks, [&](int i) { hits[i] += 1; pool_run(3, [&](int) {}); });          // a nested call runs inline
        for (int i = 0; i < n_tasks; ++i) total += (int64_t) (i + 1) * (hits[i] == r + 1);
    }
    return total;
}

int wmi_sample_draws(struct whisper_context * ctx, const float * probs, const float * logprobs, int n_draw, int reseed,
                     whisper_token_data * out) {
    Decoder & d = ctx->state->decoders[0];
    const int nv = ctx->model.vocab.n_vocab;
    d.probs.assign(probs, probs + nv); d.logprobs.assign(logprobs, logprobs + nv);
    if (reseed) d.rng = std::mt19937(0);
    for (int i = 0; i < n_draw; ++i) out[i] = sample_token(*ctx, d, false);
    return n_draw;
}

// Kernel-level cross-check: run one decoder projection of layer `layer` through BOTH implementations
// (weight-streaming GEMV and MFMA GEMM) on the same n <= 8 random activation rows and report the
// largest absolute difference over every output (q, cache k, cache v / f16 / f32 results).
double wmi_selftest_proj(struct whisper_context * ctx, int op, int n, int layer) {
    (void) hipSetDevice(ctx->device);
    State & st = *ctx->state; DeviceState & d = st.dev; const HParams & hp = ctx->model.hp;
    const DecLayerW & l = ctx->w.dec[layer];
    const int S = hp.n_text_state;
    hipStream_t s = d.stream;
    if (n < 1 || n > 8) return -1.0;
    // inputs: x (f32 residual) and a16 (f16 activations, also used as the 4S-wide MLP input)
    std::vector<float> hx((size_t) n * S); std::vector<__half> ha((size_t) n * 4 * S);
    std::mt19937 rng(1234 + op); std::normal_distribution<float> nd(0.f, 1.f);
    for (auto & v : hx) v = nd(rng);
    for (auto & v : ha) v = __float2half_rn(nd(rng));
    float * x0 = nullptr; __half * a16 = nullptr; __half * o16[2][3] = {}; float * o32[2] = {};
    bool ok = HIP_OK(hipMalloc((void **) &x0, hx.size() * 4)) && HIP_OK(hipMalloc((void **) &a16, ha.size() * 2));
    for (int v = 0; v < 2 && ok; ++v) {
        for (int t = 0; t < 3 && ok; ++t) ok = HIP_OK(hipMalloc((void **) &o16[v][t], (size_t) n * 4 * S * 2)) && HIP_OK(hipMemset(o16[v][t], 0, (size_t) n * 4 * S * 2));
        ok = ok && HIP_OK(hipMalloc((void **) &o32[v], (size_t) n * S * 4));
    }
    if (!ok) return -1.0;
    (void) hipMemcpy(a16, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    const float kq = powf((float) S / hp.n_text_head, -0.25f);
    for (int v = 0; v < 2; ++v) {
        (void) hipMemcpy(x0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        (void) hipMemcpy(o32[v], hx.data(), hx.size() * 4, hipMemcpyHostToDevice);      // residual input = x
        int epi = 0, K = S, N = S; const float * lg = nullptr, * lb = nullptr; const __half * W = nullptr; const float * bias = nullptr;
        void * C = nullptr; int ldc = S; const float * resid = nullptr; void * aux = nullptr, * aux2 = nullptr; float scale = 0.f;
        switch (op) {
            case 0: epi = k::EPI_QKV_DEC; lg = l.ln1_g; lb = l.ln1_b; N = 3 * S; W = l.w_qkv; bias = l.b_qkv; C = o16[v][0]; aux = o16[v][1]; aux2 = o16[v][2]; scale = kq; break;
            case 1: epi = k::EPI_F32_BIAS_RESID; W = l.w_o; bias = l.b_o; C = o32[v]; resid = o32[v]; break;
            case 2: epi = k::EPI_Q_SCALED; lg = l.ln2_g; lb = l.ln2_b; W = l.w_cq; bias = l.b_cq; C = o16[v][0]; scale = kq; break;
            case 4: epi = k::EPI_F16_BIAS_GELU; lg = l.ln3_g; lb = l.ln3_b; N = 4 * S; W = l.w_fc1; bias = l.b_fc1; C = o16[v][0]; ldc = 4 * S; break;
            case 5: epi = k::EPI_F32_BIAS_RESID; K = 4 * S; W = l.w_fc2; bias = l.b_fc2; C = o32[v]; resid = o32[v]; break;
            default: return -1.0;
        }
        if (v == 0) {
            k::GemvArgs g{};
            g.x32 = x0; g.ln_g = lg; g.ln_b = lb; g.eps = hp.eps; g.a16 = a16; g.n = n; g.K = K; g.N = N; g.W = W; g.bias = bias;
            g.epi = epi; g.C = C; g.ldc = ldc; g.resid = resid; g.ldr = S; g.aux = aux; g.ldaux = S; g.aux2 = aux2; g.ldaux2 = S;
            g.scale = scale; g.S = S;
            k::gemv(g, s);
        } else {
            const __half * A = a16;
            if (lg) { k::layernorm(x0, n, S, lg, lb, hp.eps, d.dxn, nullptr, s); A = d.dxn; }
            k::GemmArgs a{};
            a.A = A; a.lda = K; a.W = W; a.ldw = K; a.M = n; a.N = N; a.K = K; a.bias = bias; a.C = C; a.ldc = ldc;
            a.resid = resid; a.ldr = S; a.aux = aux; a.ldaux = S; a.aux2 = aux2; a.ldaux2 = S; a.scale = scale; a.S = S;
            k::gemm(epi, a, s);
        }
        (void) hipStreamSynchronize(s);
    }
    double worst = 0.0;
    std::vector<__half> h0((size_t) n * 4 * S), h1((size_t) n * 4 * S);
    for (int t = 0; t < 3; ++t) {
        (void) hipMemcpy(h0.data(), o16[0][t], h0.size() * 2, hipMemcpyDeviceToHost);
        (void) hipMemcpy(h1.data(), o16[1][t], h1.size() * 2, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < h0.size(); ++i) worst = std::max(worst, (double) fabsf(__half2float(h0[i]) - __half2float(h1[i])));
    }
    std::vector<float> f0((size_t) n * S), f1((size_t) n * S);
    (void) hipMemcpy(f0.data(), o32[0], f0.size() * 4, hipMemcpyDeviceToHost);
    (void) hipMemcpy(f1.data(), o32[1], f1.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < f0.size(); ++i) worst = std::max(worst, (double) fabsf(f0[i] - f1[i]));
    (void) hipFree(x0); (void) hipFree(a16);
    for (int v = 0; v < 2; ++v) { for (int t = 0; t < 3; ++t) (void) hipFree(o16[v][t]); (void) hipFree(o32[v]); }
    return worst;
}

int wmi_selftest_quant(int device, int qtype, int mode, const void * w_blocks, const float * x, const int32_t * tokens,
                       int M, int N, int K, float * out, int8_t * out_qs, float * out_ds) {
    const k::QGeom g = k::q_geom(qtype);
    if (!g.qb || M < 1 || N < 1 || K < 64 || (K % 64) != 0 || !w_blocks || !out) return -1;
    // mode 0: <= 32 rows through k_qrows; 1: the block-dot GEMM; 2: embedding gather; 3: the f16 form of the GEMM (k_qdequant + k_gemm);
    // 4: the same through the product's route (quantize_rows with the weight expansion in its launch, then qgemm: needs M >= 256)
    if ((mode == 0 && M > 32) || ((mode == 1 || mode == 3 || mode == 4) && (N % 128) != 0) || mode < 0 || mode > 4 || (mode == 2 ? !tokens : !x)) return -1;
    if (!HIP_OK(hipSetDevice(device))) return -2;
    const int nb = K / 32;
    std::vector<uint8_t> tiles(k::q_matrix_bytes(qtype, N, K));
    k::q_repack_host(qtype, (const uint8_t *) w_blocks, N, K, tiles.data());
    uint8_t * d_t = nullptr; float * d_x = nullptr, * d_o = nullptr, * d_z = nullptr; int8_t * d_qs = nullptr; float * d_ds = nullptr; int32_t * d_tok = nullptr;
    __half * d_a16 = nullptr, * d_w16 = nullptr;
    const size_t n_out = (size_t) M * (mode == 2 ? K : N);
    bool ok = HIP_OK(hipMalloc((void **) &d_t, tiles.size() + 4096)) && HIP_OK(hipMalloc((void **) &d_x, (size_t) M * K * 4)) &&
              HIP_OK(hipMalloc((void **) &d_o, n_out * 4)) && HIP_OK(hipMalloc((void **) &d_z, n_out * 4)) &&
              HIP_OK(hipMalloc((void **) &d_qs, (size_t) M * K)) && HIP_OK(hipMalloc((void **) &d_ds, (size_t) M * nb * 8)) &&
              HIP_OK(hipMalloc((void **) &d_tok, (size_t) M * 4 * 2));
    if (mode == 3 || mode == 4) ok = ok && HIP_OK(hipMalloc((void **) &d_a16, (size_t) M * K * 2)) && HIP_OK(hipMalloc((void **) &d_w16, (size_t) N * K * 2));
    hipStream_t st = nullptr;
    ok = ok && HIP_OK(hipStreamCreate(&st));
    if (ok) {
        ok = HIP_OK(hipMemcpy(d_t, tiles.data(), tiles.size(), hipMemcpyHostToDevice)) && HIP_OK(hipMemset(d_z, 0, n_out * 4)) && HIP_OK(hipMemset(d_o, 0, n_out * 4));
        if (x) ok = ok && HIP_OK(hipMemcpy(d_x, x, (size_t) M * K * 4, hipMemcpyHostToDevice));
        const k::QMat W{d_t, qtype};
        if (ok && mode == 2) {
            std::vector<int32_t> tp((size_t) 2 * M, 0);
            for (int i = 0; i < M; ++i) tp[i] = tokens[i];
            ok = HIP_OK(hipMemcpy(d_tok, tp.data(), tp.size() * 4, hipMemcpyHostToDevice));
            k::qdec_embed(d_tok, d_tok + M, M, K, W, d_z, d_o, st);           // "positional embedding" = zeros
        } else if (ok) {
            k::Q8Rows A{d_qs, d_ds, d_ds + (size_t) nb * M, M};
            if (mode == 3 || mode == 4) { A.deq = d_a16; A.wdeq = d_w16; A.wdeq_elems = (size_t) N * K; }
            if (mode == 4) { A.wdeq_ready = k::quantize_rows(d_x, nullptr, M, K, nullptr, nullptr, 0.f, qtype, A, nullptr, nullptr, st, &W, N); A.wdeq_of = W.tiles; if (!A.wdeq_ready) ok = false; }
            else k::quantize_rows(d_x, nullptr, M, K, nullptr, nullptr, 0.f, qtype, A, nullptr, nullptr, st);
            if (mode == 0) {
                k::GemvArgs ga{};
                ga.n = M; ga.K = K; ga.N = N; ga.epi = k::EPI_LOGITS; ga.C = d_o; ga.ldc = N;
                k::qrows(ga, d_x, W, st);
            } else {
                k::GemmArgs a{};
                a.M = M; a.N = N; a.K = K; a.C = d_o; a.ldc = N; a.resid = d_z; a.ldr = N;
                if (mode == 3) { k::qdequant(W, 0, N, K, d_w16, st); a.A = d_a16; a.lda = K; a.W = d_w16; a.ldw = K; k::gemm(k::EPI_F32_BIAS_RESID, a, st); }
                else k::qgemm(k::EPI_F32_BIAS_RESID, a, A, W, st);
            }
        }
        ok = ok && HIP_OK(hipStreamSynchronize(st)) && HIP_OK(hipGetLastError());
        ok = ok && HIP_OK(hipMemcpy(out, d_o, n_out * 4, hipMemcpyDeviceToHost));
        if (ok && mode != 2 && out_qs) ok = HIP_OK(hipMemcpy(out_qs, d_qs, (size_t) M * K, hipMemcpyDeviceToHost));
        if (ok && mode != 2 && out_ds) {                      // device layout: d [nb][M] | s [nb][M]  ->  [M][nb][2]
            std::vector<float> tmp((size_t) 2 * nb * M);
            ok = HIP_OK(hipMemcpy(tmp.data(), d_ds, tmp.size() * 4, hipMemcpyDeviceToHost));
            for (int m = 0; m < M && ok; ++m) for (int b = 0; b < nb; ++b) {
                out_ds[((size_t) m * nb + b) * 2] = tmp[(size_t) b * M + m]; out_ds[((size_t) m * nb + b) * 2 + 1] = tmp[(size_t) nb * M + (size_t) b * M + m];
            }
        }
    }
    if (st) (void) hipStreamDestroy(st);
    (void) hipFree(d_t); (void) hipFree(d_x); (void) hipFree(d_o); (void) hipFree(d_z); (void) hipFree(d_qs); (void) hipFree(d_ds); (void) hipFree(d_tok);
    (void) hipFree(d_a16); (void) hipFree(d_w16);
    return ok ? 0 : -3;
}

int wmi_selftest_seqsum(const float * x, int n, float * out_blocked, float * out_plain) {
    if (!x || n < 0 || !out_blocked || !out_plain) return -1;
    volatile float acc = 0.0f;                              // the definition: one f32 addition after the other
    for (int i = 0; i < n; ++i) acc = acc + x[i];
    *out_plain = acc;
    *out_blocked = seq_sum_f32(x, n);
    return 0;
}

int wmi_step_stamps(struct whisper_context * ctx, double * out, int cap, int chained) {
    if (!ctx || !ctx->state || !out) return -1;
    CtxScope lk(ctx);      // the probe replays the context's step: not beside a transcription
    (void) hipSetDevice(ctx->device);
    try { return step_stamps(*ctx, out, cap, chained != 0); } catch (...) { return -1; }
}

void wmi_reload_knobs(void) { k::reload_knobs(); }

int wmi_pair_status(struct whisper_context * ctx, int32_t * out3, int rearm) {
    if (!ctx || !ctx->state) return -1;
    CtxScope lk(ctx);
    DeviceState & d = ctx->state->dev;
    if (out3) { out3[0] = d.pair_fallbacks; out3[1] = d.pair_slow_events; out3[2] = (d.pair_off ? 1 : 0) | (d.pair_backoff > 0 ? 2 : 0); }
    if (rearm) { d.pair_off = false; d.pair_backoff = 0; }
    if (ctx->batch) {                                        // + the lock-step rows' one-launch front (k_front with the row on grid.y): bit 2 = off
        BatchWork & b = *ctx->batch;
        if (out3) { out3[0] += b.front_fallbacks; out3[2] |= b.front_off ? 4 : 0; }
        if (rearm) { b.front_off = false; b.front_backoff = 0; }
    }
    return 0;
}

double wmi_bench_kernel(struct whisper_context * ctx, int which, int iters) {
    if (!ctx || !ctx->state || iters < 1) return -1.0;
    CtxScope lk(ctx);      // the probes replay the context's kernels on its buffers: not beside a transcription
    (void) hipSetDevice(ctx->device);
    if (which >= 20) k::reload_knobs();                     // (lab scripts flip the step's switches between probe calls of one process)
    State & st = *ctx->state; DeviceState & d = st.dev; const HParams & hp = ctx->model.hp; const Weights & w = ctx->w;
    const int S = hp.n_audio_state, T = hp.n_audio_ctx, H = hp.n_audio_head;
    hipStream_t s = d.stream;
    hipEvent_t e0, e1;
    if (!HIP_OK(hipEventCreate(&e0)) || !HIP_OK(hipEventCreate(&e1))) return -1.0;
    auto once = [&]() {
        switch (which) {
            case 0: {
                if (ctx->model.quantised) {             // block-quantised mlp.0: q8 rows of the last encode x quantised tiles
                    k::GemmArgs a{};
                    a.M = T; a.N = 4 * S; a.K = S; a.bias = w.enc[0].b_fc1; a.C = d.h; a.ldc = 4 * S;
                    k::qgemm(k::EPI_F16_BIAS_GELU, a, q8_rows(d, S), w.enc[0].q_fc1, s);
                    break;
                }
                k::GemmArgs a{};
                a.A = d.xn; a.lda = S; a.W = w.enc[0].w_fc1; a.ldw = S; a.M = T; a.N = 4 * S; a.K = S; a.bias = w.enc[0].b_fc1;
                a.C = d.h; a.ldc = 4 * S;
                k::gemm(k::EPI_F16_BIAS_GELU, a, s);
            } break;
            case 1: {
                k::GemvArgs g{};
                g.x32 = d.dx; g.ln_g = w.d_ln_g; g.ln_b = w.d_ln_b; g.eps = hp.eps; g.n = 1; g.K = S; g.N = hp.n_vocab; g.W = w.d_te;
                g.epi = k::EPI_LOGITS; g.C = d.logits; g.ldc = hp.n_vocab; g.rows = nullptr;
                if (ctx->model.quantised) k::qrows(g, nullptr, w.q_te, s); else
                k::gemv(g, s);
            } break;
            case 2: k::attn_encoder(d.q, d.k, d.vt, T, d.Tpad, S, H, 0.125f, d.att, s); break;
            case 4: {                                  // mlp.0 over the lock-step work buffers: M = chunks * T
                const BatchWork & b = *ctx->batch;
                k::GemmArgs a{};
                a.A = b.xn; a.lda = S; a.W = w.enc[0].w_fc1; a.ldw = S; a.M = b.B * T; a.N = 4 * S; a.K = S; a.bias = w.enc[0].b_fc1;
                a.C = b.h; a.ldc = 4 * S;
                k::gemm(k::EPI_F16_BIAS_GELU, a, s);
            } break;
            case 9: {                                  // cross K / V of every decoder layer over the lock-step work buffers (M = chunks * T)
                const BatchWork & b = *ctx->batch;
                const int Lt = hp.n_text_layer, M = b.B * T;
                k::GemmArgs a{};
                a.A = b.enc_out_h; a.lda = S; a.W = w.w_ckv; a.ldw = S; a.M = M; a.N = Lt * 2 * S; a.K = S; a.bias = w.b_ckv;
                a.C = b.kvc_k; a.ldc = S; a.aux = b.kvc_v; a.ldaux = S; a.S = S; a.layer_stride = (int64_t) M * S;
                a.scale = powf((float) S / H, -0.25f);
                k::gemm(k::EPI_CROSS_KV, a, s);
            } break;
            case 5: {                                  // encoder attention of all lock-step chunks
                const BatchWork & b = *ctx->batch;
                k::attn_encoder(b.q, b.k, b.vt, T, b.Tpad, S, H, 0.125f, b.att, s, b.B, nullptr, b.qk_rows);
            } break;
            default: break;
        }
    };
    if ((which == 4 || which == 5 || which == 9) && (!ctx->batch || ctx->batch->B < 1)) { (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); return -1.0; }
    if (which >= 10 && which <= 12) {                                   // launch-floor probes: chains of trivial dependent kernels
        const int blocks = which == 10 ? 1 : which == 11 ? 32 : 256;
        int * p = (int *) d.mel_max;
        for (int i = 0; i < 8; ++i) k::touch(p, blocks, s);
        (void) hipStreamSynchronize(s);
        (void) hipEventRecord(e0, s);
        for (int i = 0; i < iters; ++i) k::touch(p, blocks, s);
        (void) hipEventRecord(e1, s);
        (void) hipEventSynchronize(e1);
        float ms = 0.0f; (void) hipEventElapsedTime(&ms, e0, e1);
        (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
        return (double) ms * 1000.0 / iters;
    }
    if (which == 6) {
        // vocabulary projection over a ROTATING set of copies of the token embedding, > 256 MiB in total: the matrix of one launch
        // has been evicted from the 256 MiB Infinity Cache by the time it is read again, so bytes / time is an HBM figure
        // (which = 1 re-streams the same 53 MB, i.e. measures L3 + HBM)
        const bool q = ctx->model.quantised;
        const size_t bytes = q ? k::q_matrix_bytes(w.qtype, hp.n_vocab, S) : (size_t) hp.n_vocab * S * 2;
        const int copies = (int) ((size_t) 320 * 1024 * 1024 / bytes) + 2;
        uint8_t * pool = nullptr;
        if (!HIP_OK(hipMalloc((void **) &pool, bytes * copies + 4096))) { (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); return -1.0; }
        const void * src = q ? (const void *) w.q_te.tiles : (const void *) w.d_te;
        for (int c = 0; c < copies; ++c) (void) hipMemcpyAsync(pool + (size_t) c * bytes, src, bytes, hipMemcpyDeviceToDevice, s);
        auto launch = [&](int c) {
            k::GemvArgs g{};
            g.x32 = d.dx; g.ln_g = w.d_ln_g; g.ln_b = w.d_ln_b; g.eps = hp.eps; g.n = 1; g.K = S; g.N = hp.n_vocab;
            g.epi = k::EPI_LOGITS; g.C = d.logits; g.ldc = hp.n_vocab;
            if (q) k::qrows(g, nullptr, k::QMat{pool + (size_t) c * bytes, w.qtype}, s);
            else { g.W = (const __half *) (pool + (size_t) c * bytes); k::gemv(g, s); }
        };
        for (int c = 0; c < copies; ++c) launch(c);
        (void) hipStreamSynchronize(s);
        (void) hipEventRecord(e0, s);
        for (int i = 0; i < iters; ++i) launch(i % copies);
        (void) hipEventRecord(e1, s);
        (void) hipEventSynchronize(e1);
        float ms6 = 0.0f; (void) hipEventElapsedTime(&ms6, e0, e1);
        (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); (void) hipFree(pool);
        return (double) ms6 * 1000.0 / iters;
    }
    if (which == 7 || which == 8) {
        // phase probe of one encoder mlp.0 launch (7: the lock-step work buffers, M = chunks * T; 8: one chunk): per workgroup the
        // wall-clock stamps {entry, first tile landed, K loop done, epilogue done}; prints the averages (us) to stderr
        const bool batch = which == 7;
        if (batch && (!ctx->batch || ctx->batch->B < 1)) { (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); return -1.0; }
        k::GemmArgs a{};
        if (batch) { const BatchWork & b = *ctx->batch; a.A = b.xn; a.M = b.B * T; a.C = b.h; } else { a.A = d.xn; a.M = T; a.C = d.h; }
        a.lda = S; a.W = w.enc[0].w_fc1; a.ldw = S; a.N = 4 * S; a.K = S; a.bias = w.enc[0].b_fc1; a.ldc = 4 * S;
        const int cap = 65536;
        unsigned long long * dp = nullptr;
        if (!HIP_OK(hipMalloc((void **) &dp, (size_t) cap * 5 * 8))) { (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); return -1.0; }
        for (int i = 0; i < 3; ++i) k::gemm(k::EPI_F16_BIAS_GELU, a, s);
        (void) hipMemsetAsync(dp, 0, (size_t) cap * 5 * 8, s);
        a.probe = dp;
        k::gemm(k::EPI_F16_BIAS_GELU, a, s);
        (void) hipStreamSynchronize(s);
        std::vector<unsigned long long> h((size_t) cap * 5);
        (void) hipMemcpy(h.data(), dp, h.size() * 8, hipMemcpyDeviceToHost);
        (void) hipFree(dp);
        int n = 0; unsigned long long tmin = ~0ull, tmax = 0;
        for (int i = 0; i < cap; ++i) if (h[(size_t) i * 5 + 3]) { ++n; tmin = std::min(tmin, h[(size_t) i * 5]); tmax = std::max(tmax, h[(size_t) i * 5 + 3]); }
        double f = 0, l = 0, e = 0;
        std::vector<double> starts;
        for (int i = 0; i < cap; ++i) if (h[(size_t) i * 5 + 3]) {
            const unsigned long long * q = &h[(size_t) i * 5];
            f += (double) (q[1] - q[0]); l += (double) (q[2] - q[1]); e += (double) (q[3] - q[2]);
            starts.push_back((double) (q[0] - tmin) * 0.01);
        }
        std::sort(starts.begin(), starts.end());
        const double tick = 0.01;                                   // wall_clock64: 100 MHz
        fprintf(stderr, "gemm probe (%s): %d workgroups, span %.2f us; per workgroup: entry -> first tile %.2f us, K loop %.2f us, epilogue %.2f us; "
                        "entry times: p10 %.2f p50 %.2f p90 %.2f max %.2f us\n", batch ? "M = chunks x T" : "one chunk", n,
                (double) (tmax - tmin) * tick, f / n * tick, l / n * tick, e / n * tick,
                starts[starts.size() / 10], starts[starts.size() / 2], starts[starts.size() * 9 / 10], starts.back());
        (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
        return (double) (tmax - tmin) * tick;
    }
    if (which == 20) { (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); return bench_greedy_step_chain(*ctx, iters); }
    if (which >= 21 && which <= 36) { (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); return bench_rows_step_chain(*ctx, which - 20, iters); }   // 20 + rows
    if (which == 3) {
        if (d.mel == nullptr) return -1.0;
        const int saved = st.exp_n_audio_ctx;
        encode(*ctx, 0);                                            // warm-up (includes a sync)
        const int64_t t0 = time_us();
        for (int i = 0; i < iters; ++i) encode(*ctx, 0);
        st.exp_n_audio_ctx = saved;
        (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
        return (double) (time_us() - t0) / iters;
    }
    once();
    (void) hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) once();
    (void) hipEventRecord(e1, s);
    (void) hipEventSynchronize(e1);
    float ms = 0.0f;
    (void) hipEventElapsedTime(&ms, e0, e1);
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    return (double) ms * 1000.0 / iters;
}

// One encoder pass (the last single-chunk mel: chunks = 1; the rows of the last wmi_full_batch call: chunks > 1) with a probe slice on
// every gemm() launch: out[6 i + 0..5] = {epilogue id, M, N, K, in-situ microseconds (last workgroup done - first workgroup entered),
// workgroups that reported}.  Returns the number of launches written, -1 when there is nothing to replay.
int wmi_encoder_gemm_stamps(struct whisper_context * ctx, int chunks, double * out, int cap) {
    if (!ctx || !ctx->state || !out || cap < 1 || ctx->model.quantised) return -1;
    CtxScope lk(ctx);
    (void) hipSetDevice(ctx->device);
    State & st = *ctx->state; DeviceState & d = st.dev;
    std::vector<int> rows, seek;
    if (chunks > 1) {
        if (!ctx->batch || ctx->batch->enc_rows < 2) return -1;
        for (int r = 0; r < ctx->batch->enc_rows; ++r) { rows.push_back(r); seek.push_back(0); }
    } else if (d.mel == nullptr) return -1;
    const int saved = st.exp_n_audio_ctx;
    auto pass = [&]() { return chunks > 1 ? encode_rows(*ctx, rows, seek, ctx->batch->enc_T) : encode(*ctx, 0); };
    k::GemmLog log;
    log.cap_words = (size_t) 4 << 20;
    if (!HIP_OK(hipMalloc((void **) &log.buf, log.cap_words * 8))) return -1;
    int ret = -1;
    try {
        if (pass()) {                                                   // warm-up, unstamped
            (void) hipMemsetAsync(log.buf, 0, log.cap_words * 8, d.stream);
            k::gemm_log_install(&log);
            const bool ok = pass();                                     // (both forms synchronise the stream before they return)
            k::gemm_log_install(nullptr);
            std::vector<unsigned long long> h(log.used);
            if (ok && log.used && HIP_OK(hipMemcpy(h.data(), log.buf, log.used * 8, hipMemcpyDeviceToHost))) {
                int khz = 100000; (void) hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device);
                const double tick_us = 1000.0 / (double) (khz > 0 ? khz : 100000);
                ret = 0;
                for (const k::GemmLogEntry & e : log.entries) {
                    if (ret >= cap) break;
                    unsigned long long t0 = ~0ull, t1 = 0; int n = 0;
                    for (int g = 0; g < e.cap; ++g) {
                        const unsigned long long * q = &h[e.off + (size_t) g * 5];
                        if (!q[3]) continue;
                        t0 = std::min(t0, q[0]); t1 = std::max(t1, q[3]); ++n;
                    }
                    double * o = out + (size_t) 6 * ret++;
                    o[0] = e.epi; o[1] = e.M; o[2] = e.N; o[3] = e.K; o[4] = n ? (double) (t1 - t0) * tick_us : -1.0; o[5] = n;
                }
            }
        }
    } catch (...) { ret = -1; }
    k::gemm_log_install(nullptr);
    st.exp_n_audio_ctx = saved;
    (void) hipFree(log.buf);
    return ret;
}

} // extern "C"
