// see speech_to_text.h.  Reference lines are cited per function.
#include "speech_to_text.h"
#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstring>

namespace godot_whisper {

SpeechToText::SpeechToText(const TranscribeSettings & s) : settings(s) {}

SpeechToText::~SpeechToText() {                                        // src/speech_to_text.cpp:348-351
    whisper_free(context_instance);
    context_instance = nullptr;
}

const char * SpeechToText::language_to_code(int lang) const {           // src/speech_to_text.cpp:117-324
    if (lang == Auto) return "auto";
    if (lang >= 1 && lang - 1 <= whisper_lang_max_id()) return whisper_lang_str(lang - 1);   // the enum follows whisper.cpp's table
    return "en";                                                         // "Default to English if unknown language"
}

void SpeechToText::set_language_model(const uint8_t * data, size_t size) {   // src/speech_to_text.cpp:326-346
    whisper_free(context_instance);
    context_instance = nullptr;
    fprintf(stderr, "%s\n", whisper_print_system_info());
    if (data == nullptr || size == 0) return;
    const whisper_context_params context_params{ settings.use_gpu };
    context_instance = whisper_init_from_buffer_with_params((void *) data, size, context_params);
}

std::vector<float> SpeechToText::resample(const std::vector<float> & interleaved_xy, InterpolatorType interpolator_type, int mix_rate) {   // src/speech_to_text.cpp:353-376
    last_resample_error.clear();
    const int64_t buffer_len = (int64_t) (interleaved_xy.size() / 2);
    const uint32_t expected_size = (uint32_t) (buffer_len * WHISPER_SAMPLE_RATE / mix_rate);
    if (!context_instance || buffer_len == 0) return {};
    std::vector<float> buffer_float((size_t) buffer_len);
    if (wmi_downmix_stereo(context_instance, interleaved_xy.data(), (int) buffer_len, 0, buffer_float.data()) != 0) return {};   // _vector2_array_to_float_array
    std::vector<float> resampled_float((size_t) std::max<int64_t>(expected_size, mix_rate == WHISPER_SAMPLE_RATE ? buffer_len : 0) + 1);
    int result_size = wmi_resample(context_instance, buffer_float.data(), (int) buffer_len, mix_rate, WHISPER_SAMPLE_RATE, (int) interpolator_type, 0,
                                   resampled_float.data(), (int) resampled_float.size());                                        // _resample_audio_buffer
    if (result_size < 0) result_size = 0;                                                                                      // converter error: 0 frames
    if ((uint32_t) result_size != expected_size)
        last_resample_error = "size differ exp: " + std::to_string(expected_size) + " res: " + std::to_string(result_size);
    resampled_float.resize((size_t) result_size);
    return resampled_float;
}

bool SpeechToText::voice_activity_detection(const std::vector<float> & buffer) const {     // src/speech_to_text.cpp:378-399
    const int n_samples_vad_window = WHISPER_SAMPLE_RATE * 3;            // the most recent 3 s
    const int vad_last_ms = 500;                                         // energy of the last 500 ms vs the whole window
    if ((int) buffer.size() >= n_samples_vad_window) {
        std::vector<float> window(buffer.end() - n_samples_vad_window, buffer.end());
        return vad_simple(window, WHISPER_SAMPLE_RATE, vad_last_ms, settings.vad_treshold, settings.freq_treshold);
    }
    return false;
}

whisper_full_params SpeechToText::make_params(const std::string & initial_prompt, int audio_ctx) const {   // :402-413
    whisper_full_params p = whisper_full_default_params(WHISPER_SAMPLING_GREEDY);
    p.language = language_to_code(language);
    p.audio_ctx = audio_ctx;
    p.speed_up = settings.speed_up_2x;
    p.split_on_word = true;
    p.token_timestamps = true;
    p.suppress_non_speech_tokens = true;
    p.single_segment = true;
    p.max_tokens = settings.max_tokens;
    p.entropy_thold = settings.entropy_treshold;
    p.initial_prompt = initial_prompt.c_str();          // the caller's string outlives the call (the reference lets it dangle, :413)
    return p;
}

Transcription SpeechToText::collect() const {                            // src/speech_to_text.cpp:424-447
    Transcription out;
    out.ok = true;
    const int n_segments = whisper_full_n_segments(context_instance);
    for (int i = 0; i < n_segments; ++i) {
        const int n_tokens = whisper_full_n_tokens(context_instance, i);
        out.full_text += whisper_full_get_segment_text(context_instance, i);
        for (int j = 0; j < n_tokens; ++j) {
            const whisper_token_data token = whisper_full_get_token_data(context_instance, i, j);
            Token t;
            t.text = whisper_full_get_token_text(context_instance, i, j);
            t.id = token.id; t.p = token.p; t.plog = token.plog; t.pt = token.pt; t.ptsum = token.ptsum;
            t.t0 = token.t0; t.t1 = token.t1; t.tid = token.tid; t.vlen = token.vlen;
            out.tokens.push_back(std::move(t));
        }
    }
    return out;
}

Transcription SpeechToText::transcribe(const std::vector<float> & buffer, const std::string & initial_prompt, int audio_ctx) {
    const whisper_full_params whisper_params = make_params(initial_prompt, audio_ctx);
    if (!context_instance) {
        fprintf(stderr, "ERROR: Context instance is null\n");
        return Transcription();
    }
    const int ret = whisper_full(context_instance, whisper_params, buffer.data(), (int) buffer.size());
    last_return = ret;
    if (ret != 0) {
        fprintf(stderr, "ERROR: Failed to process audio, returned %d\n", ret);
        return Transcription();
    }
    return collect();
}

std::vector<Transcription> SpeechToText::transcribe_batch(const std::vector<std::vector<float>> & buffers,
                                                          const std::string & initial_prompt, int audio_ctx) {
    std::vector<Transcription> out;
    const whisper_full_params whisper_params = make_params(initial_prompt, audio_ctx);
    if (!context_instance) {
        fprintf(stderr, "ERROR: Context instance is null\n");
        return out;
    }
    std::vector<const float *> ptrs; std::vector<int> lens;
    for (const auto & b : buffers) { ptrs.push_back(b.data()); lens.push_back((int) b.size()); }
    const int ret = wmi_full_batch(context_instance, whisper_params, ptrs.data(), lens.data(), (int) buffers.size(), 0);
    last_return = ret;
    if (ret != 0) {
        fprintf(stderr, "ERROR: Failed to process audio, returned %d\n", ret);
        return out;
    }
    for (int c = 0; c < (int) buffers.size(); ++c) {
        wmi_batch_select(context_instance, c);
        out.push_back(collect());
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ VAD
void high_pass_filter(std::vector<float> & data, float cutoff, float sample_rate) {          // src/speech_to_text.cpp:53-66
    const float rc = 1.0f / (2.0f * 3.14159265358979323846 * cutoff);      // Math_PI is a double in the reference: double arithmetic, rounded to float
    const float dt = 1.0f / sample_rate;
    const float alpha = dt / (rc + dt);
    float y = data[0];
    for (size_t i = 1; i < data.size(); i++) {
        y = alpha * (y + data[i] - data[i - 1]);
        data[i] = y;
    }
}

bool vad_simple(std::vector<float> & pcmf32, int sample_rate, int last_ms, float vad_thold, float freq_thold) {   // :69-104
    const int n_samples = (int) pcmf32.size();
    const int n_samples_last = (sample_rate * last_ms) / 1000;
    if (n_samples_last >= n_samples) return false;                        // not enough samples
    if (freq_thold > 0.0f) high_pass_filter(pcmf32, freq_thold, (float) sample_rate);
    float energy_all = 0.0f, energy_last = 0.0f;
    for (int i = 0; i < n_samples; i++) {
        energy_all += fabsf(pcmf32[i]);
        if (i >= n_samples - n_samples_last) energy_last += fabsf(pcmf32[i]);
    }
    energy_all /= n_samples;
    energy_last /= n_samples_last;
    // the host's extra "both energies tiny" clause on top of upstream vad_simple
    if (!(energy_all < 0.0001f && energy_last < 0.0001f) || energy_last > vad_thold * energy_all) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------ GDScript nodes
std::string remove_special_characters(std::string message) {            // addon/audio_stream_to_text.gd:64-88
    const char * pairs[2][2] = { {"[", "]"}, {"<", ">"} };
    for (auto & pr : pairs) {
        while (true) {
            const size_t i = message.find(pr[0]), j = message.find(pr[1]);
            if (i == std::string::npos || j == std::string::npos || j < i) break;
            message.erase(i, j + 1 - i);
        }
    }
    const std::string hallucination = ". you.";
    if (message.size() >= hallucination.size() && message.compare(message.size() - hallucination.size(), hallucination.size(), hallucination) == 0)
        message.replace(message.size() - hallucination.size(), hallucination.size(), ".");
    return message;
}

std::string AudioStreamToText::get_text(const std::vector<float> & pcm16k, const std::string & initial_prompt) {   // :31-62
    const Transcription t = transcribe(pcm16k, initial_prompt, 0);
    if (!t.ok) return std::string();
    return remove_special_characters(t.full_text);
}

static bool ends_with_any_codepoint(const std::string & text, const std::string & set) {
    if (text.empty()) return false;
    size_t i = text.size() - 1;                                           // start of the last UTF-8 code point
    while (i > 0 && ((unsigned char) text[i] & 0xC0) == 0x80) --i;
    const std::string last = text.substr(i);
    for (size_t k = 0; k < set.size();) {
        size_t n = 1;
        const unsigned char c = (unsigned char) set[k];
        if (c >= 0xF0) n = 4; else if (c >= 0xE0) n = 3; else if (c >= 0xC0) n = 2;
        if (set.compare(k, n, last) == 0) return true;
        k += n;
    }
    return false;
}

std::vector<CaptureStreamToText::Update> CaptureStreamToText::stream(const std::vector<float> & pcm16k, int max_calls) {
    // addon/capture_stream_to_text.gd:65-120 — every `transcribe_interval` seconds the WHOLE accumulated buffer is
    // transcribed again with audio_ctx = total_s * 50 + 128 (:84)
    std::vector<Update> out;
    const int sr = WHISPER_SAMPLE_RATE;
    const size_t step = (size_t) std::lround(transcribe_interval * sr);
    size_t start = 0, pos = 0; int calls = 0; int last_tokens = -1;
    while (pos < pcm16k.size()) {
        pos = std::min(pos + step, pcm16k.size());
        const std::vector<float> acc(pcm16k.begin() + start, pcm16k.begin() + pos);
        const double total_s = (double) acc.size() / sr;
        if (total_s < 1.0) continue;
        const bool no_activity = voice_activity_detection(acc);
        const int audio_ctx = std::min((int) (total_s * 50 + 128), 1500);
        const Transcription t = transcribe(acc, "", audio_ctx);
        ++calls;
        if (!t.ok) continue;
        const std::string text = remove_special_characters(t.full_text);
        const int n_tok = (int) t.tokens.size();
        bool finish = false;
        const double total_ms = total_s * 1000;
        if (total_ms > minimum_sentence_ms) {
            if ((ends_with_any_codepoint(text, punctuation_characters) || no_activity) && std::abs(n_tok - last_tokens) <= 1) finish = true;
            if (total_ms > maximum_sentence_ms) finish = true;
        }
        last_tokens = n_tok;
        out.push_back(Update{finish, text, acc.size(), audio_ctx, t.tokens});
        if (finish) {
            start = pos > (size_t) (0.2 * sr) ? pos - (size_t) (0.2 * sr) : 0;    // keep only the last 0.2 s (:111)
            last_tokens = -1;
        }
        if (max_calls >= 0 && calls >= max_calls) break;
    }
    return out;
}

} // namespace godot_whisper
