// Command-line driver of the C++ host mirror (tests/test_host_cpp.py):
//   wmi_host_demo <model.ggml> <pcm.f32> <mode> [language] [initial_prompt] [audio_ctx]
// mode: transcribe | vad | stream | batch (pcm file = several buffers, see below) | resample (pcm file = interleaved stereo frames;
//       language = mix rate, prompt = interpolator type: prints the frame count, an FNV-1a hash of the frames' bits and the error string)
// Prints one JSON document on stdout.  pcm.f32: raw little-endian float32 mono 16 kHz; for `batch` the file starts with
// int32 n, then n x int32 lengths, then the buffers back to back.
#include "speech_to_text.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

using namespace godot_whisper;

static std::vector<uint8_t> slurp(const char * path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void json_string(const std::string & s) {
    putchar('"');
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') printf("\\%c", c);
        else if (c < 0x20) printf("\\u%04x", c);
        else putchar(c);                       // raw bytes >= 0x80 are kept (the test reads the output as latin-1)
    }
    putchar('"');
}
static void json_transcription(const Transcription & t) {
    printf("{\"ok\": %s, \"full_text\": ", t.ok ? "true" : "false"); json_string(t.full_text);
    printf(", \"tokens\": [");
    for (size_t i = 0; i < t.tokens.size(); ++i) {
        const Token & k = t.tokens[i];
        printf("%s{\"text\": ", i ? ", " : ""); json_string(k.text);
        printf(", \"id\": %d, \"tid\": %d, \"p\": %.9g, \"plog\": %.9g, \"pt\": %.9g, \"ptsum\": %.9g, \"t0\": %lld, \"t1\": %lld, \"vlen\": %.9g}",
               k.id, k.tid, k.p, k.plog, k.pt, k.ptsum, (long long) k.t0, (long long) k.t1, k.vlen);
    }
    printf("]}");
}

int main(int argc, char ** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s model pcm mode [language] [prompt] [audio_ctx]\n", argv[0]); return 2; }
    const std::vector<uint8_t> model = slurp(argv[1]);
    const std::vector<uint8_t> raw = slurp(argv[2]);
    const std::string mode = argv[3];
    const int language = argc > 4 ? atoi(argv[4]) : SpeechToText::English;
    const std::string prompt = argc > 5 ? argv[5] : "";
    const int audio_ctx = argc > 6 ? atoi(argv[6]) : 0;
    whisper_log_set([](ggml_log_level, const char *, void *) {}, nullptr);

    if (mode == "vad") {
        std::vector<float> pcm((const float *) raw.data(), (const float *) raw.data() + raw.size() / 4);
        SpeechToText node;
        printf("{\"vad\": %s}\n", node.voice_activity_detection(pcm) ? "true" : "false");
        return 0;
    }
    if (mode == "stream") {
        std::vector<float> pcm((const float *) raw.data(), (const float *) raw.data() + raw.size() / 4);
        CaptureStreamToText node; node.transcribe_interval = 0.7f;
        node.set_language(language); node.set_language_model(model.data(), model.size());
        const auto ups = node.stream(pcm, 8);
        printf("[");
        for (size_t i = 0; i < ups.size(); ++i) {
            printf("%s{\"finish\": %s, \"n_samples\": %zu, \"audio_ctx\": %d, \"text\": ", i ? ", " : "", ups[i].finish ? "true" : "false", ups[i].n_samples, ups[i].audio_ctx);
            json_string(ups[i].text);
            printf(", \"ids\": [");
            for (size_t j = 0; j < ups[i].tokens.size(); ++j) printf("%s%d", j ? ", " : "", ups[i].tokens[j].id);
            printf("]}");
        }
        printf("]\n");
        return 0;
    }
    SpeechToText node;
    node.set_language(language);
    node.set_language_model(model.data(), model.size());
    if (!node.context()) { printf("{\"ok\": false, \"error\": \"no context\"}\n"); return 1; }
    if (mode == "resample") {
        std::vector<float> xy((const float *) raw.data(), (const float *) raw.data() + raw.size() / 4);
        const auto out = node.resample(xy, (SpeechToText::InterpolatorType) atoi(prompt.c_str()), language);
        uint64_t hsh = 1469598103934665603ull;
        for (float v : out) { uint32_t b; memcpy(&b, &v, 4); for (int k = 0; k < 4; ++k) { hsh ^= (b >> (8 * k)) & 0xff; hsh *= 1099511628211ull; } }
        printf("{\"frames\": %zu, \"fnv1a\": \"%016llx\", \"error\": ", out.size(), (unsigned long long) hsh);
        json_string(node.last_resample_error);
        printf("}\n");
        return 0;
    }
    if (mode == "batch") {
        const int32_t * h = (const int32_t *) raw.data();
        const int n = h[0];
        std::vector<std::vector<float>> bufs;
        const float * p = (const float *) (h + 1 + n);
        for (int i = 0; i < n; ++i) { bufs.emplace_back(p, p + h[1 + i]); p += h[1 + i]; }
        const auto res = node.transcribe_batch(bufs, prompt, audio_ctx);
        printf("[");
        for (size_t i = 0; i < res.size(); ++i) { if (i) printf(", "); json_transcription(res[i]); }
        printf("]\n");
        return 0;
    }
    std::vector<float> pcm((const float *) raw.data(), (const float *) raw.data() + raw.size() / 4);
    json_transcription(node.transcribe(pcm, prompt, audio_ctx));
    printf("\n");
    return 0;
}
