// Host side of the boundary in the reference's own language: the SpeechToText node (src/speech_to_text.h:20-172,
// src/speech_to_text.cpp:106-450) and the two GDScript call patterns (addon/audio_stream_to_text.gd:31-88,
// addon/capture_stream_to_text.gd:65-120) restated over the C ABI of include/whisper_mi355.h, with std:: types where
// the reference uses Godot's (PackedFloat32Array -> std::vector<float>, String -> std::string, Array of Dictionary ->
// Transcription).  Godot, godot-cpp and SCons are not in this image, so this is what the GDExtension would compile
// to once its Variant plumbing is removed: same methods, same argument meaning, same error behaviour (empty result +
// a message on stderr where the reference ERR_PRINTs).  godot-whisper_amd/host.py is the same mirror in Python and is
// what the parity tests drive; tests/test_host_cpp.py checks that both give identical results.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/whisper_mi355.h"
#include "../../include/wmi_device.h"

namespace godot_whisper {

// ProjectSettings "audio/input/transcribe/*" (src/register_types.cpp:64-69; spelling is the reference's)
struct TranscribeSettings {
    float entropy_treshold = 2.8f;
    float freq_treshold = 200.0f;
    int   max_tokens = 16;
    float vad_treshold = 2.0f;
    bool  use_gpu = true;
    bool  speed_up_2x = false;
};

// one element of the Array transcribe() returns (src/speech_to_text.cpp:431-443)
struct Token {
    std::string text;
    int32_t id = 0, tid = 0;
    float   p = 0, plog = 0, pt = 0, ptsum = 0, vlen = 0;
    int64_t t0 = 0, t1 = 0;
};
// the Array itself: element 0 is the full text, the rest are the token dictionaries (:447)
struct Transcription {
    bool ok = false;              // false <=> the reference returns an empty Array
    std::string full_text;
    std::vector<Token> tokens;
};

class SpeechToText {
public:
    // src/speech_to_text.h:22-128: Auto = 0, then whisper.cpp's language table in its own order
    enum Language { Auto = 0, English = 1 };

    explicit SpeechToText(const TranscribeSettings & settings = TranscribeSettings());
    ~SpeechToText();
    SpeechToText(const SpeechToText &) = delete;
    SpeechToText & operator=(const SpeechToText &) = delete;

    void set_language(int p_language) { language = p_language; }
    int  get_language() const { return language; }
    // set_language_model(Ref<WhisperResource>) + _load_model(): the resource's bytes (src/resource_whisper.h)
    void set_language_model(const uint8_t * data, size_t size);

    // src/speech_to_text.h:151-155, :162 — resample(PackedVector2Array buffer, InterpolatorType): stereo capture frames at the mix rate
    // -> mono 16 kHz.  The reference folds on the CPU and calls libsamplerate's src_simple; here both steps run on the device
    // (wmi_downmix_stereo, wmi_resample: the same arithmetic frame for frame).  `mix_rate` stands in for AudioServer::get_mix_rate().
    enum InterpolatorType { SRC_SINC_BEST_QUALITY = 0, SRC_SINC_MEDIUM_QUALITY = 1, SRC_SINC_FASTEST = 2 };
    std::vector<float> resample(const std::vector<float> & interleaved_xy, InterpolatorType interpolator_type, int mix_rate);
    std::string last_resample_error;      // what the reference ERR_PRINTs ("size differ exp: ... res: ...")

    bool voice_activity_detection(const std::vector<float> & buffer) const;
    Transcription transcribe(const std::vector<float> & buffer, const std::string & initial_prompt, int audio_ctx);
    // not in the reference: several buffers in lock-step on one GPU (include/wmi_device.h: wmi_full_batch); element c is
    // what transcribe(buffers[c], initial_prompt, audio_ctx) returns on a fresh context
    std::vector<Transcription> transcribe_batch(const std::vector<std::vector<float>> & buffers, const std::string & initial_prompt,
                                                int audio_ctx);

    whisper_context * context() { return context_instance; }
    int last_return = 0;          // whisper_full's code of the last transcribe()

protected:
    whisper_full_params make_params(const std::string & initial_prompt, int audio_ctx) const;
    Transcription collect() const;
    const char * language_to_code(int language) const;

    TranscribeSettings settings;
    int language = English;
    whisper_context * context_instance = nullptr;
};

// addon/audio_stream_to_text.gd
class AudioStreamToText : public SpeechToText {
public:
    using SpeechToText::SpeechToText;
    std::string get_text(const std::vector<float> & pcm16k, const std::string & initial_prompt = "");
};
std::string remove_special_characters(std::string message);          // addon/audio_stream_to_text.gd:64-88

// addon/capture_stream_to_text.gd: the streaming loop over a pre-recorded 16 kHz buffer (capture frames go through resample() first)
class CaptureStreamToText : public SpeechToText {
public:
    using SpeechToText::SpeechToText;
    float transcribe_interval = 0.3f;
    int   minimum_sentence_ms = 3000, maximum_sentence_ms = 15000;
    std::string punctuation_characters = ".!?;\xe3\x80\x82\xef\xbc\x9b\xef\xbc\x9f\xef\xbc\x81";
    struct Update { bool finish; std::string text; size_t n_samples; int audio_ctx; std::vector<Token> tokens; };
    std::vector<Update> stream(const std::vector<float> & pcm16k, int max_calls = -1);
};

// src/speech_to_text.cpp:53-104
void high_pass_filter(std::vector<float> & data, float cutoff, float sample_rate);
bool vad_simple(std::vector<float> & pcmf32, int sample_rate, int last_ms, float vad_thold, float freq_thold);

} // namespace godot_whisper
