/*
 * whisper_mi355.h — the drop-in boundary of libwhisper_mi355.so.
 *
 * This is the reference's public C API (thirdparty/whisper.cpp/whisper.h, v1.5.4; cited below
 * as W/whisper.h:<line>): every WHISPER_API function of that header is exported, first the ones
 * the Godot GDExtension host and the reference's bench/compare tools call, then the rest
 * (caller-owned states, older constructors, whisper_full_parallel, bench entry points).
 * Symbol names, argument order, by-value struct layouts and return codes are ABI: a host
 * compiled against W/whisper.h links against this library unchanged.  Everything behind these
 * entry points is new (HIP kernels for gfx950).
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

typedef struct whisper_model_loader {                                    /* W/whisper.h:108-114 */
    void * context;
    size_t (*read)(void * ctx, void * output, size_t read_size);
    bool   (*eof)(void * ctx);
    void   (*close)(void * ctx);
} whisper_model_loader;

/* grammar element types (W/whisper.h:116-140): END closes a rule, ALT starts its next alternative, RULE_REF names a rule by
 * index, CHAR / CHAR_NOT open a (negated) character class of code points, CHAR_RNG_UPPER makes the preceding value the low
 * end of a range, CHAR_ALT adds a further member to the class */
enum whisper_gretype {
    WHISPER_GRETYPE_END = 0, WHISPER_GRETYPE_ALT = 1, WHISPER_GRETYPE_RULE_REF = 2, WHISPER_GRETYPE_CHAR = 3,
    WHISPER_GRETYPE_CHAR_NOT = 4, WHISPER_GRETYPE_CHAR_RNG_UPPER = 5, WHISPER_GRETYPE_CHAR_ALT = 6,
};
typedef struct whisper_grammar_element { enum whisper_gretype type; uint32_t value; } whisper_grammar_element; /* W/whisper.h:142-145 */

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
 * -6 encode; -7 prompt decode; -8 decode (W/whisper.cpp:4976-5561); -9 (this backend only) the device-side refinement of token
 * timestamps failed — lock-step calls keep the |x| envelope in HBM, so there is no host form to fall back to.  Not re-entrant per context. */
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

/* ---- the remaining constructors (W/whisper.h:150-193).  The loader forms drain the callbacks into memory and close
 * the loader in every case (W/whisper.cpp:3253-3269); *_no_state leaves the context without a state: create one with
 * whisper_init_state and use the *_with_state calls ---- */
WHISPER_API struct whisper_context * whisper_init_with_params(struct whisper_model_loader * loader, struct whisper_context_params params);            /* :152 */
WHISPER_API struct whisper_context * whisper_init_from_file_with_params_no_state(const char * path_model, struct whisper_context_params params);      /* :156 */
WHISPER_API struct whisper_context * whisper_init_from_buffer_with_params_no_state(void * buffer, size_t buffer_size, struct whisper_context_params params); /* :157 */
WHISPER_API struct whisper_context * whisper_init_with_params_no_state(struct whisper_model_loader * loader, struct whisper_context_params params);   /* :158 */
WHISPER_API struct whisper_context * whisper_init_from_file(const char * path_model);                     /* :160-163, deprecated there */
WHISPER_API struct whisper_context * whisper_init_from_buffer(void * buffer, size_t buffer_size);         /* :164-167 */
WHISPER_API struct whisper_context * whisper_init(struct whisper_model_loader * loader);                  /* :168-171 */
WHISPER_API struct whisper_context * whisper_init_from_file_no_state(const char * path_model);            /* :172-175 */
WHISPER_API struct whisper_context * whisper_init_from_buffer_no_state(void * buffer, size_t buffer_size);/* :176-179 */
WHISPER_API struct whisper_context * whisper_init_no_state(struct whisper_model_loader * loader);         /* :180-183 */
WHISPER_API struct whisper_context_params * whisper_context_default_params_by_ref(void);                  /* :529, free with whisper_free_context_params */
WHISPER_API struct whisper_full_params * whisper_full_default_params_by_ref(enum whisper_sampling_strategy strategy); /* :531, free with whisper_free_params */
WHISPER_API void whisper_free_params(struct whisper_full_params * params);                                /* :206 */
WHISPER_API void whisper_free_context_params(struct whisper_context_params * params);                     /* :207 */
/* always 1: this library is not an OpenVINO build (the reference's answer without WHISPER_USE_OPENVINO, W/whisper.cpp:3122-3134) */
WHISPER_API int whisper_ctx_init_openvino_encoder(struct whisper_context * ctx, const char * model_path, const char * device, const char * cache_dir); /* :198 */

/* ---- caller-owned states (W/whisper.h:185, :205).  A state holds everything mutable (KV caches, activation arenas,
 * its HIP stream and captured decode graph, results) on the context's GPU; the weights stay with the context.
 * Calls on one context serialise on an internal lock; contexts are independent. ---- */
WHISPER_API struct whisper_state * whisper_init_state(struct whisper_context * ctx);      /* NULL on allocation failure or a host-only context */
WHISPER_API void whisper_free_state(struct whisper_state * state);                        /* NULL-safe */
WHISPER_API int whisper_pcm_to_mel_with_state(struct whisper_context * ctx, struct whisper_state * state, const float * samples, int n_samples, int n_threads); /* :234 */
WHISPER_API int whisper_set_mel_with_state(struct whisper_context * ctx, struct whisper_state * state, const float * data, int n_len, int n_mel);               /* :263 */
WHISPER_API int whisper_encode_with_state(struct whisper_context * ctx, struct whisper_state * state, int offset, int n_threads);                               /* :279 */
WHISPER_API int whisper_decode_with_state(struct whisper_context * ctx, struct whisper_state * state, const whisper_token * tokens, int n_tokens, int n_past, int n_threads); /* :295 */
WHISPER_API int whisper_lang_auto_detect_with_state(struct whisper_context * ctx, struct whisper_state * state, int offset_ms, int n_threads, float * lang_probs); /* :344 */
WHISPER_API int whisper_full_with_state(struct whisper_context * ctx, struct whisper_state * state, struct whisper_full_params params, const float * samples, int n_samples); /* :543 */
/* The x2 phase-vocoder front end: always -1.  The reference evaluates it with an 800-sample frame whose 401 bins index
 * the 201-bin filterbank out of bounds (W/whisper.cpp:3417-3425 -> :2764-2776), and its own whisper_full refuses
 * speed_up (:4973-4976): there is no defined result to reproduce. */
WHISPER_API int whisper_pcm_to_mel_phase_vocoder(struct whisper_context * ctx, const float * samples, int n_samples, int n_threads);                            /* :247 */
WHISPER_API int whisper_pcm_to_mel_phase_vocoder_with_state(struct whisper_context * ctx, struct whisper_state * state, const float * samples, int n_samples, int n_threads); /* :253 */
/* Split the audio into n_processors pieces, transcribe each on its own state, merge the segments with the reference's
 * time shift and no-overlap clamp (W/whisper.cpp:5817-5924).  The pieces run one after the other on the GPU; for
 * independent chunks in lock-step use wmi_full_batch (wmi_device.h). */
WHISPER_API int whisper_full_parallel(struct whisper_context * ctx, struct whisper_full_params params, const float * samples, int n_samples, int n_processors); /* :553 */

/* ---- results of a state (W/whisper.h:360-367, :565-602) ---- */
WHISPER_API int     whisper_n_len_from_state(struct whisper_state * state);
WHISPER_API float * whisper_get_logits_from_state(struct whisper_state * state);
WHISPER_API int     whisper_full_n_segments_from_state(struct whisper_state * state);
WHISPER_API int     whisper_full_lang_id_from_state(struct whisper_state * state);
WHISPER_API int64_t whisper_full_get_segment_t0_from_state(struct whisper_state * state, int i_segment);
WHISPER_API int64_t whisper_full_get_segment_t1_from_state(struct whisper_state * state, int i_segment);
WHISPER_API bool    whisper_full_get_segment_speaker_turn_next(struct whisper_context * ctx, int i_segment);
WHISPER_API bool    whisper_full_get_segment_speaker_turn_next_from_state(struct whisper_state * state, int i_segment);
WHISPER_API const char * whisper_full_get_segment_text_from_state(struct whisper_state * state, int i_segment);
WHISPER_API int     whisper_full_n_tokens_from_state(struct whisper_state * state, int i_segment);
WHISPER_API const char * whisper_full_get_token_text_from_state(struct whisper_context * ctx, struct whisper_state * state, int i_segment, int i_token);
WHISPER_API whisper_token whisper_full_get_token_id_from_state(struct whisper_state * state, int i_segment, int i_token);
WHISPER_API whisper_token_data whisper_full_get_token_data_from_state(struct whisper_state * state, int i_segment, int i_token);
WHISPER_API float   whisper_full_get_token_p_from_state(struct whisper_state * state, int i_segment, int i_token);
WHISPER_API const char * whisper_lang_str_full(int id);                                   /* :335: "english" for 0 */

/* ---- bench entry points (W/whisper.h:608-611).  The reference times host memcpy and ggml_mul_mat on n_threads cores;
 * here they report what does the work on this backend: HBM device-to-device copy bandwidth and the f16 MFMA GEMM on
 * the same square sizes (64..4096).  n_threads is accepted and ignored; the *_str forms return a static buffer. ---- */
WHISPER_API int          whisper_bench_memcpy(int n_threads);
WHISPER_API const char * whisper_bench_memcpy_str(int n_threads);
WHISPER_API int          whisper_bench_ggml_mul_mat(int n_threads);
WHISPER_API const char * whisper_bench_ggml_mul_mat_str(int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_MI355_H */
