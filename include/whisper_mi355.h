/*
 * whisper_mi355.h — the drop-in boundary of libwhisper_mi355.so.
 *
 * This is the subset of the reference's public C API (thirdparty/whisper.cpp/whisper.h,
 * v1.5.4; cited below as W/whisper.h:<line>) that the Godot GDExtension host and the
 * reference's own bench/compare tools call.  Symbol names, argument order, by-value struct
 * layouts and return codes are ABI: a host compiled against W/whisper.h links against this
 * library unchanged.  Everything behind these entry points is new (HIP kernels for gfx950).
 *
 * The 11 entry points the host binds (src/speech_to_text.cpp:332-447, src/register_types.cpp:58)
 * are marked [host].
 */
#ifndef WHISPER_MI355_H
#define WHISPER_MI355_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#define WHISPER_SAMPLE_RATE 16000 /* W/whisper.h:32 */
#define WHISPER_N_FFT       400   /* W/whisper.h:33 */
#define WHISPER_HOP_LENGTH  160   /* W/whisper.h:34 */
#define WHISPER_CHUNK_SIZE  30    /* W/whisper.h:35 */

#define WHISPER_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

struct whisper_context;
struct whisper_state;

typedef int32_t whisper_pos;
typedef int32_t whisper_token;
typedef int32_t whisper_seq_id;

/* W/ggml.h:486-491, :1914 — log levels and callback used by whisper_log_set */
enum ggml_log_level { GGML_LOG_LEVEL_ERROR = 2, GGML_LOG_LEVEL_WARN = 3, GGML_LOG_LEVEL_INFO = 4, GGML_LOG_LEVEL_DEBUG = 5 };
typedef void (*ggml_log_callback)(enum ggml_log_level level, const char * text, void * user_data);

struct whisper_context_params { bool use_gpu; };                         /* W/whisper.h:87-89 */

typedef struct whisper_token_data {                                      /* W/whisper.h:91-106 */
    whisper_token id, tid;
    float p, plog, pt, ptsum;
    int64_t t0, t1;
    float vlen;
} whisper_token_data;

typedef struct whisper_grammar_element { int type; uint32_t value; } whisper_grammar_element; /* W/whisper.h:130-133 */

enum whisper_sampling_strategy { WHISPER_SAMPLING_GREEDY, WHISPER_SAMPLING_BEAM_SEARCH };    /* W/whisper.h:397-400 */

typedef void (*whisper_new_segment_callback)(struct whisper_context *, struct whisper_state *, int n_new, void * ud);
typedef void (*whisper_progress_callback)(struct whisper_context *, struct whisper_state *, int progress, void * ud);
typedef bool (*whisper_encoder_begin_callback)(struct whisper_context *, struct whisper_state *, void * ud);
typedef bool (*whisper_abort_callback)(void * ud);
typedef void (*whisper_logits_filter_callback)(struct whisper_context *, struct whisper_state *,
        const whisper_token_data * tokens, int n_tokens, float * logits, void * ud);

/* W/whisper.h:433-526 — passed BY VALUE; field order is ABI */
struct whisper_full_params {
    enum whisper_sampling_strategy strategy;
    int n_threads, n_max_text_ctx, offset_ms, duration_ms;
    bool translate, no_context, no_timestamps, single_segment;
    bool print_special, print_progress, print_realtime, print_timestamps;
    bool token_timestamps; float thold_pt, thold_ptsum; int max_len; bool split_on_word; int max_tokens;
    bool speed_up, debug_mode; int audio_ctx;
    bool tdrz_enable;
    const char * initial_prompt; const whisper_token * prompt_tokens; int prompt_n_tokens;
    const char * language; bool detect_language;
    bool suppress_blank, suppress_non_speech_tokens;
    float temperature, max_initial_ts, length_penalty;
    float temperature_inc, entropy_thold, logprob_thold, no_speech_thold;
    struct { int best_of; } greedy;
    struct { int beam_size; float patience; } beam_search;
    whisper_new_segment_callback   new_segment_callback;   void * new_segment_callback_user_data;
    whisper_progress_callback      progress_callback;      void * progress_callback_user_data;
    whisper_encoder_begin_callback encoder_begin_callback; void * encoder_begin_callback_user_data;
    whisper_abort_callback         abort_callback;         void * abort_callback_user_data;
    whisper_logits_filter_callback logits_filter_callback; void * logits_filter_callback_user_data;
    const whisper_grammar_element ** grammar_rules; size_t n_grammar_rules, i_start_rule; float grammar_penalty;
};

/* ---- [host] model lifetime (W/whisper.h:151, :205, :391) ---- */
WHISPER_API struct whisper_context * whisper_init_from_buffer_with_params(void * buffer, size_t buffer_size, struct whisper_context_params params);
WHISPER_API void         whisper_free(struct whisper_context * ctx);           /* NULL-safe */
WHISPER_API const char * whisper_print_system_info(void);

/* ---- [host] one transcription (W/whisper.h:532, :537-541) ---- */
WHISPER_API struct whisper_full_params whisper_full_default_params(enum whisper_sampling_strategy strategy);
/* 0 ok; -1 speed_up unsupported; -2 mel; -3 language detect; -4 too many decoders; -5 audio_ctx too large;
 * -6 encode; -7 prompt decode; -8 decode (W/whisper.cpp:4976-5561).  Not re-entrant per context. */
WHISPER_API int whisper_full(struct whisper_context * ctx, struct whisper_full_params params, const float * samples, int n_samples);

/* ---- [host] result walk (W/whisper.h:564, :585-601) ---- */
WHISPER_API int          whisper_full_n_segments(struct whisper_context * ctx);
WHISPER_API int          whisper_full_n_tokens(struct whisper_context * ctx, int i_segment);
WHISPER_API const char * whisper_full_get_segment_text(struct whisper_context * ctx, int i_segment);
WHISPER_API const char * whisper_full_get_token_text(struct whisper_context * ctx, int i_segment, int i_token);
WHISPER_API whisper_token_data whisper_full_get_token_data(struct whisper_context * ctx, int i_segment, int i_token);

/* ---- [host] logging (W/whisper.h:619) ---- */
WHISPER_API void whisper_log_set(ggml_log_callback log_callback, void * user_data);

/* ---- rest of the reference API used by comparison / bench tools (W/examples/bench/bench.cpp:64-135) ---- */
WHISPER_API struct whisper_context_params whisper_context_default_params(void);                                /* :147 */
WHISPER_API struct whisper_context * whisper_init_from_file_with_params(const char * path, struct whisper_context_params params); /* :150 */
WHISPER_API int whisper_pcm_to_mel(struct whisper_context * ctx, const float * samples, int n_samples, int n_threads);           /* :240 */
WHISPER_API int whisper_set_mel(struct whisper_context * ctx, const float * data, int n_len, int n_mel);                         /* :270 */
WHISPER_API int whisper_encode(struct whisper_context * ctx, int offset, int n_threads);                                         /* :286 */
WHISPER_API int whisper_decode(struct whisper_context * ctx, const whisper_token * tokens, int n_tokens, int n_past, int n_threads); /* :303 */
WHISPER_API int whisper_tokenize(struct whisper_context * ctx, const char * text, whisper_token * tokens, int n_max_tokens);     /* :319 */
WHISPER_API float * whisper_get_logits(struct whisper_context * ctx);                                                           /* :366 */
WHISPER_API int whisper_lang_max_id(void);
WHISPER_API int whisper_lang_id(const char * lang);
WHISPER_API const char * whisper_lang_str(int id);
WHISPER_API int whisper_lang_auto_detect(struct whisper_context * ctx, int offset_ms, int n_threads, float * lang_probs);       /* :338 */
WHISPER_API int whisper_n_len(struct whisper_context * ctx);
WHISPER_API int whisper_n_vocab(struct whisper_context * ctx);
WHISPER_API int whisper_n_text_ctx(struct whisper_context * ctx);
WHISPER_API int whisper_n_audio_ctx(struct whisper_context * ctx);
WHISPER_API int whisper_is_multilingual(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_vocab(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_audio_ctx(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_audio_state(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_audio_head(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_audio_layer(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_text_ctx(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_text_state(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_text_head(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_text_layer(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_mels(struct whisper_context * ctx);
WHISPER_API int whisper_model_ftype(struct whisper_context * ctx);
WHISPER_API int whisper_model_type(struct whisper_context * ctx);
WHISPER_API const char * whisper_model_type_readable(struct whisper_context * ctx);
WHISPER_API const char * whisper_token_to_str(struct whisper_context * ctx, whisper_token token);
WHISPER_API whisper_token whisper_token_eot(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_sot(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_solm(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_prev(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_nosp(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_not(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_beg(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_lang(struct whisper_context * ctx, int lang_id);
WHISPER_API whisper_token whisper_token_translate(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_transcribe(struct whisper_context * ctx);
WHISPER_API int whisper_full_lang_id(struct whisper_context * ctx);
WHISPER_API int64_t whisper_full_get_segment_t0(struct whisper_context * ctx, int i_segment);
WHISPER_API int64_t whisper_full_get_segment_t1(struct whisper_context * ctx, int i_segment);
WHISPER_API whisper_token whisper_full_get_token_id(struct whisper_context * ctx, int i_segment, int i_token);
WHISPER_API float whisper_full_get_token_p(struct whisper_context * ctx, int i_segment, int i_token);
WHISPER_API void whisper_print_timings(struct whisper_context * ctx);
WHISPER_API void whisper_reset_timings(struct whisper_context * ctx);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_MI355_H */
