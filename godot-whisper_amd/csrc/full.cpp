// The transcription driver behind whisper_full() (reference
// W/whisper.cpp:4960-5807) and the token-level timestamp heuristics the Godot host enables
// (row (f)4; W/whisper.cpp:6315-6599).  Control flow stays on the host CPU; the three hot calls —
// pcm_to_mel, encode, decode — run on the GPU.  Decisions (seek window, prompt assembly, temperature
// fallback, beam bookkeeping through the unified KV cache, segment splitting) are reproduced
// decision-for-decision so that the token stream equals the reference's when the logits agree.

#include "wmi.h"

#include <algorithm>
#include <immintrin.h>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace wmi {

namespace {

constexpr int MAX_DECODERS = 8;          // W/whisper.cpp:148

struct BeamCandidate { int decoder_idx; int seek_delta; bool has_ts; Sequence sequence; Grammar grammar; };

const char * tok_str(whisper_context & ctx, int32_t id) { return ctx.model.vocab.id_to_token.at(id).c_str(); }

void emit_segment(whisper_context & ctx, State & st, const whisper_full_params & params, int64_t t0, int64_t t1, const std::string & text,
                  const std::vector<whisper_token_data> & toks, int i0, int i1_excl, bool speaker_turn_next) {
    const int64_t tt0 = params.speed_up ? 2 * t0 : t0, tt1 = params.speed_up ? 2 * t1 : t1;
    if (params.print_realtime) {
        if (params.print_timestamps) {              // hh:mm:ss.mmm of a count of 10 ms units (W/whisper.cpp:2599-2612)
            auto stamp = [](int64_t t, char (&buf)[32]) {
                const int64_t ms = t * 10;
                snprintf(buf, sizeof(buf), "%02d:%02d:%02d.%03d", (int) (ms / 3600000), (int) ((ms / 60000) % 60), (int) ((ms / 1000) % 60), (int) (ms % 1000));
            };
            char b0[32], b1[32];
            stamp(tt0, b0); stamp(tt1, b1);
            printf("[%s --> %s]  %s\n", b0, b1, text.c_str());
        }
        else { printf("%s", text.c_str()); fflush(stdout); }
    }
    Segment seg{tt0, tt1, text, {}, speaker_turn_next};
    seg.tokens.assign(toks.begin() + i0, toks.begin() + i1_excl);
    st.result_all.push_back(std::move(seg));
    int n_new = 1;
    if (params.token_timestamps) {
        static const bool dbg_e = getenv("WMI_DEBUG_EMIT") != nullptr;
        const int64_t e0 = time_us();
        (void) signal_energy_wait(st);             // the envelope's D2H copy ran behind the encoder / decoder
        const int64_t e1 = time_us();
        token_level_timestamps(ctx, st, (int) st.result_all.size() - 1, params.thold_pt, params.thold_ptsum);
        if (dbg_e) fprintf(stderr, "[wmi] emit: envelope wait %lld us, token timestamps %lld us\n", (long long) (e1 - e0), (long long) (time_us() - e1));
        if (params.max_len > 0) n_new = wrap_segment(ctx, st, params.max_len, params.split_on_word);
    }
    if (params.new_segment_callback)
        params.new_segment_callback(&ctx, (whisper_state *) &st, n_new, params.new_segment_callback_user_data);
}

} // namespace

int full(whisper_context & ctx, whisper_full_params params, const float * samples, const float * d_samples, int n_samples) {
    BusyScope busy(ctx.device);                                      // (see wmi.h: kernels that wait inside a launch are used by a lone transcription only)
    State & st = *ctx.state;
    static const bool dbg_t = getenv("WMI_DEBUG_TIMING") != nullptr;
    const int64_t T0 = time_us(); int64_t T_mel = 0, T_energy = 0, T_emit = 0;
    struct Report { bool on; int64_t t0; int64_t * mel, * en, * emit; State * st; ~Report() { if (on) fprintf(stderr,
        "[wmi] full: total %.3f ms | mel %.3f | envelope %.3f | encode %.3f | decode %.3f (+prompt/batch %.3f) | segments+timestamps %.3f\n",
        (time_us() - t0) / 1e3, *mel / 1e3, *en / 1e3, st->t_encode_us / 1e3, st->t_decode_us / 1e3, (st->t_prompt_us + st->t_batchd_us) / 1e3, *emit / 1e3); } }
        report{dbg_t, T0, &T_mel, &T_energy, &T_emit, &st};
    if (dbg_t) { st.t_encode_us = st.t_decode_us = st.t_prompt_us = st.t_batchd_us = 0; }
    const Vocab & v = ctx.model.vocab;
    const HParams & hp = ctx.model.hp;
    st.result_all.clear();
    st.ts_failed = false;
    // the envelope kernel reads the caller's samples on a side stream: never return while it is in flight
    struct EnvelopeGuard { State & st; ~EnvelopeGuard() { (void) signal_energy_wait(st); } } envelope_guard{st};
    static const bool defer_phases = getenv("WMI_PHASE_SYNC") == nullptr;
    struct PhaseGuard { State & st; ~PhaseGuard() { (void) phase_settle(st, true); } } phase_guard{st};      // (a call that ends before any decode step)

    if (n_samples > 0) {
        if (params.speed_up) { WMI_ERR("%s: failed to compute log mel spectrogram\n", __func__); return -1; }
        // (no host wait behind the log-mel and the encoder: the next phase's launches queue up behind them; WMI_PHASE_SYNC=1: wait as before)
        const bool ok_mel = d_samples ? pcm_to_mel(ctx, d_samples, n_samples, true, true, defer_phases) : pcm_to_mel(ctx, samples, n_samples, false, true, defer_phases);
        if (!ok_mel) { WMI_ERR("%s: failed to compute log mel spectrogram\n", __func__); return -2; }
        T_mel = time_us() - T0;
    }

    if (params.language == nullptr || strlen(params.language) == 0 || strcmp(params.language, "auto") == 0 || params.detect_language) {
        std::vector<float> probs(lang_max_id() + 1, 0.0f);
        const int lid = lang_auto_detect(ctx, 0, probs.data());
        if (lid < 0) { WMI_ERR("%s: failed to auto-detect language\n", __func__); return -3; }
        st.lang_id = lid;
        params.language = lang_str(lid);
        WMI_INFO("%s: auto-detected language: %s (p = %f)\n", __func__, params.language, probs[lid]);
        if (params.detect_language) return 0;
    }

    if (params.token_timestamps) {
        st.t_beg = 0; st.t_last = 0; st.tid_last = 0;
        // the |x| envelope is computed on the GPU from the samples pcm_to_mel just staged (bit-identical to the
        // CPU loop, k_signal_energy); the host loop remains only as the definition it is tested against
        const int64_t te0 = time_us();
        struct En { int64_t & acc; int64_t t0; ~En() { acc += time_us() - t0; } } en_timer{T_energy, te0};
        if (n_samples > 0 && !signal_energy_device(ctx, 32, false)) { WMI_ERR("%s: failed to compute the signal envelope\n", __func__); return -2; }
    }

    const int seek_start = params.offset_ms / 10;
    const int seek_end = params.duration_ms == 0 ? st.mel.n_len_org : seek_start + params.duration_ms / 10;
    if (seek_end < seek_start + (params.speed_up ? 50 : 100)) return 0;      // < 1 s of audio: nothing to do

    std::vector<float> temperatures;
    if (params.temperature_inc > 0.0f) for (float t = params.temperature; t < 1.0f + 1e-6f; t += params.temperature_inc) temperatures.push_back(t);
    else temperatures.push_back(params.temperature);

    int n_decoders = 1;
    if (params.strategy == WHISPER_SAMPLING_GREEDY) n_decoders = params.greedy.best_of;
    else if (params.strategy == WHISPER_SAMPLING_BEAM_SEARCH) n_decoders = std::max(params.greedy.best_of, params.beam_search.beam_size);
    n_decoders = std::max(1, n_decoders);
    if (n_decoders > MAX_DECODERS) { WMI_ERR("%s: too many decoders requested (%d), max = %d\n", __func__, n_decoders, MAX_DECODERS); return -4; }
    for (int j = 1; j < n_decoders; ++j) {
        Decoder & d = st.decoders[j];
        d.probs.resize(v.n_vocab); d.logits.resize(v.n_vocab); d.logprobs.resize(v.n_vocab);
        d.rng = std::mt19937(0);
    }

    auto & prompt_past = st.prompt_past;
    if (params.no_context) prompt_past.clear();
    std::vector<int32_t> prompt_tokens_own;
    if (!params.prompt_tokens && params.initial_prompt) {
        prompt_tokens_own = tokenize(v, params.initial_prompt);
        if (prompt_tokens_own.size() > 1024) {             // the reference tokenises into a 1024-slot buffer and gets -1
            WMI_ERR("%s: too many resulting tokens: %d (max %d)\n", "whisper_tokenize", (int) prompt_tokens_own.size(), 1024);
            prompt_tokens_own.clear();                     // (resize(-1) there is undefined; we treat it as "no prompt")
        }
        params.prompt_tokens = prompt_tokens_own.data();
        params.prompt_n_tokens = (int) prompt_tokens_own.size();
    }
    if (params.prompt_tokens && params.prompt_n_tokens > 0) {
        for (int i = 0; i < params.prompt_n_tokens; ++i) prompt_past.push_back(params.prompt_tokens[i]);
        std::rotate(prompt_past.begin(), prompt_past.end() - params.prompt_n_tokens, prompt_past.end());
    }

    if (params.audio_ctx > hp.n_audio_ctx) {
        WMI_ERR("%s: audio_ctx is larger than the maximum allowed (%d > %d)\n", __func__, params.audio_ctx, hp.n_audio_ctx);
        return -5;
    }
    st.exp_n_audio_ctx = params.audio_ctx;

    std::vector<int32_t> prompt_init = { v.sot };
    if (v.is_multilingual()) {
        const int lid = lang_id(params.language);
        st.lang_id = lid;
        prompt_init.push_back(v.sot + 1 + lid);
        prompt_init.push_back(params.translate ? v.translate : v.transcribe);
    }
    if (hp.n_text_layer == 2 && !params.no_timestamps) {   // distilled checkpoints
        WMI_WARN("%s: using distilled model - forcing no_timestamps\n", __func__);
        params.no_timestamps = true;
    }
    if (params.no_timestamps) prompt_init.push_back(v.not_);

    int seek = seek_start;
    std::vector<int32_t> prompt;
    prompt.reserve(hp.n_text_ctx);
    std::vector<std::vector<BeamCandidate>> bc_per_dec(n_decoders);
    std::vector<BeamCandidate> beam_candidates;
    const bool beam = params.strategy == WHISPER_SAMPLING_BEAM_SEARCH;

    while (true) {
        if (params.progress_callback)
            params.progress_callback(&ctx, (whisper_state *) ctx.state.get(), (100 * (seek - seek_start)) / (seek_end - seek_start),
                                     params.progress_callback_user_data);
        if (seek + 100 >= seek_end) break;
        if (params.encoder_begin_callback &&
            !params.encoder_begin_callback(&ctx, (whisper_state *) ctx.state.get(), params.encoder_begin_callback_user_data)) {
            WMI_ERR("%s: encoder_begin_callback returned false - aborting\n", __func__);
            break;
        }
        if (!encode(ctx, seek, defer_phases) || (params.abort_callback && params.abort_callback(params.abort_callback_user_data))) {
            WMI_ERR("%s: failed to encode\n", __func__);
            return -6;
        }
        if (seek > seek_start && seek + 500 >= seek_end) prompt_past.clear();

        int best_decoder_id = 0;
        for (int it = 0; it < (int) temperatures.size(); ++it) {
            const float t_cur = temperatures[it];
            int n_cur = 1;
            if (!beam) { if (t_cur > 0.0f) n_cur = params.greedy.best_of; }
            else       { n_cur = t_cur > 0.0f ? params.greedy.best_of : params.beam_search.beam_size; }
            n_cur = std::max(1, n_cur);

            for (int j = 0; j < n_cur; ++j) {
                Decoder & d = st.decoders[j];
                d.sequence.tokens.clear();
                d.sequence.result_len = 0; d.sequence.sum_logprobs_all = 0.0;
                d.sequence.sum_logprobs = -INFINITY; d.sequence.avg_logprobs = -INFINITY;
                d.sequence.entropy = 0.0; d.sequence.score = -INFINITY;
                d.seek_delta = 100 * WHISPER_CHUNK_SIZE;
                d.failed = false; d.completed = false; d.has_ts = false;
                d.grammar = params.grammar_rules ? grammar_init(params.grammar_rules, params.n_grammar_rules, params.i_start_rule) : Grammar{};
            }

            // prompt = [prev, tail of past text] (only while t < 0.5) + [sot, (lang, task), (notimestamps)]
            prompt.clear();
            if (!prompt_past.empty() && t_cur < 0.5f && params.n_max_text_ctx > 0) {
                const int n_take = std::min(std::min(params.n_max_text_ctx, hp.n_text_ctx / 2), (int) prompt_past.size());
                prompt.push_back(v.prev);
                prompt.insert(prompt.end(), prompt_past.end() - n_take, prompt_past.end());
            }
            prompt.insert(prompt.end(), prompt_init.begin(), prompt_init.end());

            kv_clear(st.kv_self);
            // Greedy, temperature 0, one decoder, no user logit callback: filters + arg-max run on the GPU and each
            // step is one graph replay (device.cpp: decode_greedy_step).  Everything else takes the general path.
            const bool fast = fast_path_enabled() && !beam && t_cur < 1e-6f && n_cur == 1 && !params.logits_filter_callback && !params.grammar_rules &&
                              params.n_grammar_rules == 0 &&
                              ctx.model.n_loaded > 0 && upload_static_ban(ctx, params);
            // Beam search and t > 0 (whisper_sample_token_topk / whisper_sample_token(best = false)): the filters, the soft-max and the
            // CDF search of the draws run on the device as well (device.cpp: sample_rows_device) — the decoders' mt19937 generators stay
            // here and supply the uniform numbers.  User callbacks and grammars need the host arrays and keep the host path.
            const bool no_dev_draw = k::knobs().host_draws;                           // debug / A-B and the tests (wmi_reload_knobs after a change)
            const bool dev_draw = !fast && !no_dev_draw && fast_path_enabled() && (beam || t_cur > 0.0f) && n_cur <= MAX_DECODERS &&
                                  !params.logits_filter_callback && !params.grammar_rules && params.n_grammar_rules == 0 &&
                                  ctx.model.n_loaded > 0 && upload_static_ban(ctx, params);
            st.dev.keep_logits_on_device = dev_draw;
            struct KeepOff { bool & f; ~KeepOff() { f = false; } } keep_off{st.dev.keep_logits_on_device};
            whisper_token_data fast_next{};      // token picked on the device for the upcoming sampling step
            auto step_filter = [&](const Decoder & d) {
                const auto & h = d.sequence.tokens;
                StepFilter f{};
                const bool initial = h.empty();
                f.ban_blank = params.suppress_blank && initial;
                f.last_ts = !h.empty() && h.back().id >= v.beg;
                f.penult_ts = h.size() < 2 || h[h.size() - 2].id >= v.beg;
                f.ts_floor_end = d.has_ts ? v.beg + d.seek_delta / 2 : v.beg;
                f.ts_initial_start = v.n_vocab;
                if (initial && params.max_initial_ts > 0.0f) {
                    const float precision = float(WHISPER_CHUNK_SIZE) / hp.n_audio_ctx;
                    f.ts_initial_start = v.beg + (int) std::round(params.max_initial_ts / precision) + 1;
                }
                return f;
            };
            if (fast) {
                bool ok = true;
                static const bool stepwise = getenv("WMI_PROMPT_STEPWISE") != nullptr;   // debug / A-B: prompt token by token
                if (stepwise) {
                    for (size_t t = 0; ok && t + 1 < prompt.size(); ++t)
                        ok = decode_greedy_step(ctx, prompt[t], (int) t, step_filter(st.decoders[0]), fast_next);
                } else
                if (prompt.size() > 1) {            // all but the last prompt token: plain batch decode, no logits wanted
                    st.batch.prep_legacy(prompt.data(), (int) prompt.size() - 1, 0, 0);
                    st.batch.logits[prompt.size() - 2] = 0;
                    ok = decode(ctx, st.batch);
                }
                ok = ok && decode_greedy_step(ctx, prompt.back(), (int) prompt.size() - 1, step_filter(st.decoders[0]), fast_next);
                if (!ok || (params.abort_callback && params.abort_callback(params.abort_callback_user_data))) {
                    WMI_ERR("%s: failed to decode\n", __func__);
                    return -7;
                }
            } else {
            st.batch.prep_legacy(prompt.data(), (int) prompt.size(), 0, 0);
            if (!decode(ctx, st.batch) || (params.abort_callback && params.abort_callback(params.abort_callback_user_data))) {
                WMI_ERR("%s: failed to decode\n", __func__);
                return -7;
            }
            {
                const int64_t ts = time_us();
                st.decoders[0].i_batch = (int) prompt.size() - 1;
                if (!dev_draw) process_logits(ctx, st.decoders[0], params, t_cur);
                for (int j = 1; j < n_cur; ++j) {
                    Decoder & d = st.decoders[j];
                    kv_seq_cp(st.kv_self, 0, j, -1, -1);
                    if (!dev_draw) { d.probs = st.decoders[0].probs; d.logits = st.decoders[0].logits; d.logprobs = st.decoders[0].logprobs; }
                }
                for (int j = 0; j < n_cur; ++j) st.decoders[j].i_batch = 0;      // device draws: every decoder reads logits row 0 first
                st.t_sample_us += time_us() - ts;
            }
            }

            const int n_max = hp.n_text_ctx / 2 - 4;
            for (int i = 0; i < n_max; ++i) {
                const int64_t ts = time_us();
                if (beam) for (auto & bc : bc_per_dec) bc.clear();

                // sample (each decoder owns its RNG, so the result does not depend on threading): the beams of a step draw on the
                // host worker pool — a 51 866-entry CDF per decoder is ~0.15 ms of one core, five of them in a row were a third of a
                // large-v3 beam step
                std::vector<std::vector<whisper_token_data>> drawn;
                if (dev_draw) {
                    // uniform numbers exactly as std::discrete_distribution would take them from each decoder's generator
                    const int k = beam ? params.beam_search.beam_size : 1;
                    StepFilter fl[MAX_DECODERS]; int rows_[MAX_DECODERS], live[MAX_DECODERS]; double u[MAX_DECODERS * MAX_DECODERS];
                    int nl = 0;
                    for (int j = 0; j < n_cur; ++j) {
                        Decoder & d = st.decoders[j];
                        if (d.completed || d.failed) continue;
                        fl[nl] = step_filter(d); rows_[nl] = i == 0 ? 0 : d.i_batch; live[nl] = j;
                        for (int c = 0; c < std::min(k, (int) MAX_DECODERS); ++c) u[nl * std::min(k, (int) MAX_DECODERS) + c] = std::generate_canonical<double, 53>(d.rng);
                        ++nl;
                    }
                    const int kk = std::min(k, (int) MAX_DECODERS);
                    std::vector<whisper_token_data> res((size_t) nl * kk);
                    if (nl > 0 && !sample_rows_device(ctx, fl, rows_, nl, t_cur, kk, u, beam ? v.beg : 0, res.data())) {
                        WMI_ERR("%s: device sampling failed\n", __func__);
                        return -8;
                    }
                    drawn.resize(n_cur);
                    for (int r = 0; r < nl; ++r) drawn[live[r]].assign(res.begin() + (size_t) r * kk, res.begin() + (size_t) (r + 1) * kk);
                } else
                if (beam && n_cur > 1) {
                    drawn.resize(n_cur);
                    int n_live = 0;
                    for (int j = 0; j < n_cur; ++j) if (!st.decoders[j].completed && !st.decoders[j].failed) ++n_live;
                    pool_run(n_cur, [&](int j) {
                        Decoder & d = st.decoders[j];
                        if (!d.completed && !d.failed) drawn[j] = sample_token_topk(ctx, d, params.beam_search.beam_size, false);
                    });
                    st.n_sample += n_live;
                }
                for (int j = 0; j < n_cur; ++j) {
                    Decoder & d = st.decoders[j];
                    if (d.completed || d.failed) continue;
                    if (fast) {
                        d.sequence.tokens.push_back(fast_next);
                        d.sequence.sum_logprobs_all += fast_next.plog;
                    } else if (!beam) {
                        d.sequence.tokens.push_back(dev_draw ? drawn[j][0] : sample_token(ctx, d, t_cur < 1e-6f));
                        d.sequence.sum_logprobs_all += d.sequence.tokens.back().plog;
                    } else {
                        for (const auto & tok : (drawn.empty() ? sample_token_topk(ctx, d, params.beam_search.beam_size) : drawn[j])) {
                            bc_per_dec[j].push_back({ j, d.seek_delta, d.has_ts, d.sequence, d.grammar });
                            bc_per_dec[j].back().sequence.tokens.push_back(tok);
                            bc_per_dec[j].back().sequence.sum_logprobs_all += tok.plog;
                        }
                    }
                }

                if (beam) {
                    beam_candidates.clear();
                    for (const auto & bc : bc_per_dec) beam_candidates.insert(beam_candidates.end(), bc.begin(), bc.end());
                    std::sort(beam_candidates.begin(), beam_candidates.end(), [](const BeamCandidate & a, const BeamCandidate & b) {
                        return a.sequence.sum_logprobs_all > b.sequence.sum_logprobs_all; });
                    uint32_t cur_c = 0;
                    for (int j = 0; j < n_cur; ++j) {
                        Decoder & d = st.decoders[j];
                        if (d.completed || d.failed) continue;
                        if (cur_c >= beam_candidates.size()) cur_c = 0;
                        BeamCandidate & cur = beam_candidates[cur_c++];
                        while (beam_candidates.size() > cur_c &&
                               beam_candidates[cur_c].sequence.sum_logprobs_all == cur.sequence.sum_logprobs_all && i > 0) ++cur_c;
                        d.seek_delta = cur.seek_delta; d.has_ts = cur.has_ts; d.sequence = cur.sequence; d.grammar = cur.grammar;
                        kv_seq_cp(st.kv_self, cur.decoder_idx, MAX_DECODERS + j, -1, -1);
                    }
                    for (int j = 0; j < n_cur; ++j) {
                        Decoder & d = st.decoders[j];
                        if (d.completed || d.failed) continue;
                        kv_seq_rm(st.kv_self, j, -1, -1);
                        kv_seq_cp(st.kv_self, MAX_DECODERS + j, j, -1, -1);
                        kv_seq_rm(st.kv_self, MAX_DECODERS + j, -1, -1);
                    }
                }

                // per-decoder state machine: timestamps move the window, EOT / limits complete the segment
                for (int j = 0; j < n_cur; ++j) {
                    Decoder & d = st.decoders[j];
                    if (d.completed || d.failed) continue;
                    int & result_len = d.sequence.result_len;
                    const whisper_token_data & tok = d.sequence.tokens.back();
                    if (tok.id > v.beg) {
                        const int sd_new = 2 * (tok.id - v.beg);
                        if (d.has_ts && d.seek_delta > sd_new && result_len < i) { d.failed = true; continue; }   // going back in time
                        d.seek_delta = sd_new; result_len = i + 1; d.has_ts = true;
                    }
                    grammar_accept_token(ctx, d.grammar, tok.id);
                    if (tok.id == v.eot || (params.max_tokens > 0 && i >= params.max_tokens) ||
                        (d.has_ts && seek + d.seek_delta + 100 >= seek_end)) {
                        if (result_len == 0) {
                            if (seek + d.seek_delta + 100 >= seek_end) result_len = i + 1;
                            else { d.failed = true; continue; }
                        }
                        if (params.single_segment) { result_len = i + 1; d.seek_delta = 100 * WHISPER_CHUNK_SIZE; }
                        d.completed = true;
                        continue;
                    }
                    if (ctx.model.n_loaded == 0) { d.seek_delta = 100 * WHISPER_CHUNK_SIZE; d.completed = true; continue; }   // empty test model
                    if (i == n_max - 1 && (result_len == 0 || d.seek_delta < 100 * WHISPER_CHUNK_SIZE / 2)) { d.failed = true; continue; }
                }

                bool all_done = true;
                for (int j = 0; j < n_cur; ++j) if (!st.decoders[j].completed && !st.decoders[j].failed) all_done = false;
                if (all_done) break;
                st.t_sample_us += time_us() - ts;

                if (fast) {
                    Decoder & d0 = st.decoders[0];
                    if (!decode_greedy_step(ctx, d0.sequence.tokens.back().id, (int) prompt.size() + i, step_filter(d0), fast_next) ||
                        (params.abort_callback && params.abort_callback(params.abort_callback_user_data))) {
                        WMI_ERR("%s: failed to decode\n", __func__);
                        return -8;
                    }
                    continue;
                }
                // next step: one token per live decoder, each in its own sequence id
                Batch & b = st.batch;
                b.n_tokens = 0;
                const int n_past = (int) prompt.size() + i;
                for (int j = 0; j < n_cur; ++j) {
                    Decoder & d = st.decoders[j];
                    if (d.failed || d.completed) continue;
                    d.i_batch = b.n_tokens;
                    b.token[b.n_tokens] = d.sequence.tokens.back().id; b.pos[b.n_tokens] = n_past;
                    b.seq_id[b.n_tokens] = j; b.logits[b.n_tokens] = 1;
                    b.n_tokens++;
                }
                if (!decode(ctx, b) || (params.abort_callback && params.abort_callback(params.abort_callback_user_data))) {
                    WMI_ERR("%s: failed to decode\n", __func__);
                    return -8;
                }
                const int64_t ts2 = time_us();
                if (dev_draw) { /* the logits stay on the device: filters and draws happen in sample_rows_device next iteration */ }
                else if (n_cur > 1 && !params.logits_filter_callback && !params.grammar_rules) {
                    // the decoders' filter + log-soft-max passes are independent (own logits row, own arrays): one pool task each
                    pool_run(n_cur, [&](int j) {
                        Decoder & d = st.decoders[j];
                        if (!d.failed && !d.completed) process_logits(ctx, d, params, t_cur);
                    });
                } else
                for (int j = 0; j < n_cur; ++j) {
                    Decoder & d = st.decoders[j];
                    if (d.failed || d.completed) continue;
                    process_logits(ctx, d, params, t_cur);
                }
                st.t_sample_us += time_us() - ts2;
            }

            // rank the finished sequences
            {
                double best_score = -INFINITY;
                for (int j = 0; j < n_cur; ++j) {
                    Decoder & d = st.decoders[j];
                    if (d.failed) continue;
                    d.sequence.tokens.resize(d.sequence.result_len);
                    sequence_score(params, d.sequence);
                    if (d.sequence.result_len > 32 && d.sequence.entropy < params.entropy_thold) { d.failed = true; st.n_fail_h++; continue; }
                    if (best_score < d.sequence.score) { best_score = d.sequence.score; best_decoder_id = j; }
                }
            }
            bool success = true;
            if (it != (int) temperatures.size() - 1) {
                const Decoder & d = st.decoders[best_decoder_id];
                if (d.failed || d.sequence.avg_logprobs < params.logprob_thold) { success = false; st.n_fail_p++; }
            }
            if (success) break;
        }

        // results of this window
        {
            const int64_t tem0 = time_us();
            struct Em { int64_t & acc; int64_t t0; ~Em() { acc += time_us() - t0; } } em_timer{T_emit, tem0};
            const Decoder & best = st.decoders[best_decoder_id];
            emit_window(ctx, st, params, seek, prompt, prompt_init.size(), best);
            if (st.ts_failed) { WMI_ERR("%s: failed to refine the token timestamps on the device\n", __func__); return -9; }
            seek += best.seek_delta;
        }
    }
    return 0;
}

void emit_window(whisper_context & ctx, State & st, const whisper_full_params & params, int seek, const std::vector<int32_t> & prompt,
                 size_t n_prompt_init, const Decoder & best) {
    const Vocab & v = ctx.model.vocab;
    auto & prompt_past = st.prompt_past;
    const int seek_delta = best.seek_delta, result_len = best.sequence.result_len;
    const auto & toks = best.sequence.tokens;

    prompt_past.clear();
    if (prompt.front() == v.prev) prompt_past.insert(prompt_past.end(), prompt.begin() + 1, prompt.end() - n_prompt_init);
    for (int i = 0; i < result_len; ++i) prompt_past.push_back(toks[i].id);

    // envelope in HBM (lock-step calls): the segments' envelope-side refinement is ONE device call per window instead of one per segment
    // (nothing between the segments reads the refined times: no callback, no max_len re-wrap)
    st.ts_defer = params.token_timestamps && params.max_len <= 0 && !params.new_segment_callback && !params.print_realtime;
    st.ts_pending.clear();
    struct Flush { whisper_context & c; State & s; ~Flush() { (void) flush_token_timestamps(c, s); s.ts_defer = false; } } flush_guard{ctx, st};
    if (!toks.empty() && ctx.model.n_loaded > 0) {
        int i0 = 0;
        int64_t t0 = seek + 2 * (toks.front().tid - v.beg);
        std::string text;
        bool speaker_turn_next = false;
        for (int i = 0; i < (int) toks.size(); ++i) {
            if (params.print_special || toks[i].id < v.eot) text += tok_str(ctx, toks[i].id);
            if (params.tdrz_enable && toks[i].id == v.solm) speaker_turn_next = true;
            if (toks[i].id > v.beg && !params.single_segment) {
                const int64_t t1 = seek + 2 * (toks[i].tid - v.beg);
                if (!text.empty()) emit_segment(ctx, st, params, t0, t1, text, toks, i0, i + 1, speaker_turn_next);
                text.clear();
                while (i < (int) toks.size() && toks[i].id > v.beg) ++i;
                --i;
                t0 = t1; i0 = i + 1; speaker_turn_next = false;
            }
        }
        if (!text.empty()) emit_segment(ctx, st, params, t0, seek + seek_delta, text, toks, i0, (int) toks.size(), speaker_turn_next);
    }
}


// ------------------------------------------------------------------ token-level timestamps (W/whisper.cpp:6315-6599)
namespace {
int ts_to_sample(int64_t t, int n_samples) { return std::max(0, std::min(n_samples - 1, (int) ((t * WHISPER_SAMPLE_RATE) / 100))); }
int64_t sample_to_ts(int i) { return (100ll * i) / WHISPER_SAMPLE_RATE; }
float voice_length(const char * s) {
    float r = 0.0f;
    for (; *s; ++s) {
        const char c = *s;
        if (c == ' ') r += 0.01f; else if (c == ',') r += 2.00f;
        else if (c == '.' || c == '!' || c == '?') r += 3.00f;
        else if (c >= '0' && c <= '9') r += 3.00f; else r += 1.00f;
    }
    return r;
}
} // namespace

std::vector<float> signal_energy(const float * signal, int n_samples, int hw) {
    // mean |x| over a (2 hw + 1) window; same left-to-right f32 summation as the reference so the
    // thresholds below see identical values
    std::vector<float> out(n_samples);
    for (int i = 0; i < n_samples; ++i) {
        float sum = 0.0f;
        const int a = std::max(0, i - hw), b = std::min(n_samples - 1, i + hw);
        for (int j = a; j <= b; ++j) sum += fabsf(signal[j]);
        out[i] = sum / (2 * hw + 1);
    }
    return out;
}

// The value of   s = 0; for (i < n) s += p[i];   (f32, left to right — the reference's window sums, W/whisper.cpp:6506-6515), without its
// 4-cycle dependency per element.  While the accumulator stays inside one binade [2^E, 2^(E+1)) it is a multiple of ulp = 2^(E-23), and
// adding x rounds the exact sum to a multiple of that ulp: acc += ulp * rne(x / ulp), an INTEGER addition — unless x / ulp ends in exactly
// .5 (the tie goes to the even neighbour of the running sum, which depends on the sum) or the sum leaves the binade.  So blocks of 128
// elements are converted and added as integers with AVX2 (x / ulp is an exact power-of-two scaling, the conversion rounds to nearest
// even like the adder); a block that contains a tie, a negative or huge element, or that could cross the binade is added the plain way.
// The envelope is non-negative, so the sum is monotone and crosses at most ~30 binades per window.  Bit-identical to the loop
// (tests/test_abi.py::test_sequential_sum_is_exact, random / tie-heavy / denormal / crossing inputs).
float seq_sum_f32(const float * p, int n) {
    float acc = 0.0f;
    int i = 0;
    constexpr int B = 128;                                  // (per lane 16 additions of < 2^23: no 32-bit overflow; 64: 12.2, 128: 10.5, 256: 10.5 us per 36 000 elements)
    while (i < n) {
        uint32_t bits; memcpy(&bits, &acc, 4);
        const int E = (int) ((bits >> 23) & 0xFF) - 127;                        // acc in [2^E, 2^(E+1)) when normal and positive
        if (i + B <= n && (bits >> 31) == 0 && E >= -100 && E <= 100) {
            const uint32_t units = (bits & 0x7FFFFFu) | 0x800000u;               // acc / ulp, in [2^23, 2^24)
            uint32_t sb = (uint32_t) (127 + 23 - E) << 23; float scale; memcpy(&scale, &sb, 4);       // 2^(23 - E) = 1 / ulp
            const __m256 vs = _mm256_set1_ps(scale), half = _mm256_set1_ps(0.5f), big = _mm256_set1_ps(8388608.0f);
            const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7FFFFFFF));
            __m256i isum = _mm256_setzero_si256();
            __m256 bad = _mm256_setzero_ps();
            for (int k = 0; k < B; k += 8) {
                const __m256 t = _mm256_mul_ps(_mm256_loadu_ps(p + i + k), vs);
                const __m256i q = _mm256_cvtps_epi32(t);                         // round to nearest even (MXCSR default)
                const __m256 d = _mm256_and_ps(_mm256_sub_ps(t, _mm256_cvtepi32_ps(q)), absmask);
                bad = _mm256_or_ps(bad, _mm256_cmp_ps(d, half, _CMP_EQ_OQ));     // tie: decided by the parity of the running sum
                bad = _mm256_or_ps(bad, _mm256_cmp_ps(t, big, _CMP_NLT_UQ));     // >= 2^23 units (leaves the binade) or NaN
                bad = _mm256_or_ps(bad, _mm256_cmp_ps(t, _mm256_setzero_ps(), _CMP_LT_OQ));      // negative: the sum could drop a binade
                isum = _mm256_add_epi32(isum, q);
            }
            if (_mm256_movemask_ps(bad) == 0) {
                __m128i s4 = _mm_add_epi32(_mm256_castsi256_si128(isum), _mm256_extracti128_si256(isum, 1));
                s4 = _mm_add_epi32(s4, _mm_shuffle_epi32(s4, 0x4E)); s4 = _mm_add_epi32(s4, _mm_shuffle_epi32(s4, 0xB1));
                const uint64_t total = (uint64_t) units + (uint32_t) _mm_cvtsi128_si32(s4);
                if (total < 0x1000000u) {                                        // every prefix stayed below 2^24 units: same ulp throughout
                    const uint32_t nb = ((uint32_t) (E + 127) << 23) | ((uint32_t) total & 0x7FFFFFu);
                    memcpy(&acc, &nb, 4);
                    i += B;
                    continue;
                }
            }
        }
        const int e = std::min(n, i + B);
        for (; i < e; ++i) acc += p[i];
    }
    return acc;
}

// The envelope stayed in HBM (lock-step calls): the window sums and the four walks a token may take are computed there, one wavefront per
// token (k_ts_refine) — every token's sum and walks depend only on its own t0 / t1 as they stand after the first half of
// token_level_timestamps.  What remains is the reference's sequential pass over a segment's tokens (clamps against the neighbours),
// replayed here on those results:
//   walk_down_while_above(s0)         -> w_down_above_s0                walk_up_while_below(s0, last = s1) -> w_up_below_s0
//   walk_up_while_above(s1, n - 1)    -> w_up_above_s1
//   walk_down_while_below(s1, first)  -> s1 <= first ? s1 : max(first, w_down_below_s1)      (the device walk has first = 0)
// All pending segments of a window go in ONE device call.
void ts_collect(whisper_context & ctx, State & st, std::vector<k::TsTok> & in, std::vector<TsRef> & ref) {
    const Vocab & v = ctx.model.vocab;
    const int n_samples = st.energy_n;
    const int hw = WHISPER_SAMPLE_RATE / 8;
    for (int si : st.ts_pending) {
        auto & tokens = st.result_all[si].tokens;
        for (int j = 0; j < (int) tokens.size(); ++j) {
            if (tokens[j].id >= v.eot) continue;
            const int s0 = ts_to_sample(tokens[j].t0, n_samples), s1 = ts_to_sample(tokens[j].t1, n_samples);
            in.push_back(k::TsTok{ s0, s1, std::max(s0 - hw, 0), std::min(s1 + hw, n_samples), st.dev.energy, n_samples, (int32_t) st.dev.energy_cap });
            ref.push_back(TsRef{ &st, si, j });
        }
    }
    st.ts_pending.clear();
}

void ts_apply(const std::vector<k::TsTok> & in, const std::vector<TsRef> & ref, const std::vector<k::TsOut> & res) {
    for (size_t q = 0; q < ref.size(); ++q) {                    // per state: segments in order, tokens in order — the reference's loop
        auto & tokens = ref[q].st->result_all[ref[q].seg].tokens;
        const int n = (int) tokens.size(), j = ref[q].j;
        const k::TsOut & r = res[q];
        int s0 = in[q].s0, s1 = in[q].s1;
        const int ns = in[q].a1 - in[q].a0;
        if (r.e0 && j > 0) {
            const int k2 = r.w_down_above_s0;
            tokens[j].t0 = sample_to_ts(k2);
            if (tokens[j].t0 < tokens[j - 1].t1) tokens[j].t0 = tokens[j - 1].t1; else s0 = k2;
        } else {
            const int k2 = r.w_up_below_s0;
            s0 = k2;
            tokens[j].t0 = sample_to_ts(k2);
        }
        if (r.e1) {
            const int k2 = r.w_up_above_s1;
            tokens[j].t1 = sample_to_ts(k2);
            // (`j < ns - 1`: the reference's test against the window length, see the host loop below)
            if (j < ns - 1 && j + 1 < n && tokens[j].t1 > tokens[j + 1].t0) tokens[j].t1 = tokens[j + 1].t0; else s1 = k2;
        } else {
            const int k2 = s1 <= s0 ? s1 : std::max(s0, r.w_down_below_s1);
            s1 = k2;
            tokens[j].t1 = sample_to_ts(k2);
        }
        (void) s1;
    }
}

// the pending segments of ONE state, or (lock-step calls) of all chunks at once: one launch, one synchronisation
bool flush_token_timestamps(whisper_context & ctx, State & st) {
    if (st.ts_pending.empty() || st.ts_hold) return true;
    std::vector<State *> one{ &st };
    return flush_token_timestamps_of(ctx, one);
}
bool flush_token_timestamps_of(whisper_context & ctx, const std::vector<State *> & states) {
    std::vector<k::TsTok> in; std::vector<TsRef> ref;
    State * first = nullptr;
    std::vector<State *> involved;
    for (State * s : states) if (s && !s->ts_pending.empty()) { if (!first) first = s; involved.push_back(s); ts_collect(ctx, *s, in, ref); }
    if (!first || in.empty()) return true;
    std::vector<k::TsOut> res(in.size());
    for (size_t q0 = 0; q0 < in.size(); q0 += 448) {            // (the pinned block holds 448 records)
        const int cnt = (int) std::min<size_t>(448, in.size() - q0);
        if (!ts_refine_device(*first, in.data() + q0, cnt, res.data() + q0)) {
            WMI_ERR("%s: timestamp refinement on the device failed\n", __func__);
            for (State * s : involved) s->ts_failed = true;      // reported by whisper_full / wmi_full_batch as -9
            return false;
        }
    }
    ts_apply(in, ref, res);
    return true;
}

void token_level_timestamps(whisper_context & ctx, State & st, int i_segment, float thold_pt, float thold_ptsum) {
    const Vocab & v = ctx.model.vocab;
    Segment & seg = st.result_all[i_segment];
    auto & tokens = seg.tokens;
    const int n_samples = st.energy_n;
    if (n_samples == 0) { WMI_ERR("%s: no signal data available\n", __func__); return; }
    const int64_t t0 = seg.t0, t1 = seg.t1;
    const int n = (int) tokens.size();
    if (n == 0) return;
    if (n == 1) { tokens[0].t0 = t0; tokens[0].t1 = t1; return; }

    for (int j = 0; j < n; ++j) {
        whisper_token_data & tok = tokens[j];
        if (j == 0) {
            if (tok.id == v.beg) {
                tokens[0].t0 = t0; tokens[0].t1 = t0; tokens[1].t0 = t0;
                st.t_beg = t0; st.t_last = t0; st.tid_last = v.beg;
            } else {
                tokens[0].t0 = st.t_last;
            }
        }
        const int64_t tt = st.t_beg + 2 * (tok.tid - v.beg);
        tok.vlen = voice_length(tok_str(ctx, tok.id));
        if (tok.pt > thold_pt && tok.ptsum > thold_ptsum && tok.tid > st.tid_last && tt <= t1) {
            if (j > 0) tokens[j - 1].t1 = tt;
            tok.t0 = tt;
            st.tid_last = tok.tid;
        }
    }
    tokens[n - 2].t1 = t1; tokens[n - 1].t0 = t1; tokens[n - 1].t1 = t1;
    st.t_last = t1;

    // spread unknown intervals proportionally to the voice length
    for (int p0 = 0, p1 = 0;;) {
        while (p1 < n && tokens[p1].t1 < 0) ++p1;
        if (p1 >= n) --p1;
        if (p1 > p0) {
            double psum = 0.0;
            for (int j = p0; j <= p1; ++j) psum += tokens[j].vlen;
            const double dt = (double) (tokens[p1].t1 - tokens[p0].t0);
            for (int j = p0 + 1; j <= p1; ++j) {
                const double ct = tokens[j - 1].t0 + dt * tokens[j - 1].vlen / psum;
                tokens[j - 1].t1 = (int64_t) ct;
                tokens[j].t0 = (int64_t) ct;
            }
        }
        ++p1; p0 = p1;
        if (p1 >= n) break;
    }
    for (int j = 0; j < n - 1; ++j) {
        if (tokens[j].t1 < 0) tokens[j + 1].t0 = tokens[j].t1;
        if (j > 0 && tokens[j - 1].t1 > tokens[j].t0) {
            tokens[j].t0 = tokens[j - 1].t1;
            tokens[j].t1 = std::max(tokens[j].t0, tokens[j].t1);
        }
    }

    // expand / contract by voice activity
    const int hw = WHISPER_SAMPLE_RATE / 8;
    if (st.energy_on_device && st.ts_defer) { st.ts_pending.push_back(i_segment); return; }
    if (st.energy_on_device) {                               // something reads the refined times right away (max_len re-wrap, a callback): now
        st.ts_pending.assign(1, i_segment);
        std::vector<State *> one{ &st };
        (void) flush_token_timestamps_of(ctx, one);          // failure: st.ts_failed, reported by the caller of the window
        return;
    }
    const float * en = st.energy;                    // pinned host memory the GPU wrote (device.cpp: signal_energy_device)
    // The walks below ("move left/right while the envelope stays above/below the threshold") run to the end of
    // the signal on stationary audio — millions of scalar steps per call in the reference.  They are pure
    // searches, so they are done 16 samples at a time with a branch-free block test the compiler vectorises;
    // the element they stop on is the same.
    // The walks are pure searches.  With the per-256-sample block minima / maxima the GPU wrote next to the envelope a
    // whole block is decided by one comparison (all above th <=> block min > th, all below <=> block max < th); the
    // sample the walk stops on is the same as for the reference's one-by-one loop.
    const float * bmin = st.energy_bmin, * bmax = st.energy_bmax;
    // (whole blocks are skipped on the block extrema alone: stepping block by block through `en` touched one cache line per KB of
    //  envelope — 1 875 misses for a walk to the end of a 30 s signal; the 1 875 extrema are 7.5 KB)
    auto walk_down_while_above = [&](int k, float th) {      // while (k > 0 && en[k] > th) --k;
        while (k > 0 && en[k] > th) {
            if ((k & 255) == 255 && bmin && bmin[k >> 8] > th) {
                int b = k >> 8;
                while (b >= 0 && bmin[b] > th) --b;
                if (b < 0) return 0;
                k = (b << 8) + 255; continue;
            }
            --k;
        }
        return k;
    };
    auto walk_up_while_above = [&](int k, float th, int last) {   // while (k < last && en[k] > th) ++k;
        while (k < last && en[k] > th) {
            if ((k & 255) == 0 && k + 256 <= last && bmin && bmin[k >> 8] > th) {
                int b = k >> 8;
                while (((b + 1) << 8) <= last && bmin[b] > th) ++b;
                k = b << 8; continue;
            }
            ++k;
        }
        return k;
    };
    auto walk_up_while_below = [&](int k, float th, int last) {   // while (en[k] < th && k < last) ++k;
        while (en[k] < th && k < last) {
            if ((k & 255) == 0 && k + 256 <= last && bmax && bmax[k >> 8] < th) {
                int b = k >> 8;
                while (((b + 1) << 8) <= last && bmax[b] < th) ++b;
                k = b << 8; continue;
            }
            ++k;
        }
        return k;
    };
    auto walk_down_while_below = [&](int k, float th, int first) { // while (en[k] < th && k > first) --k;
        while (en[k] < th && k > first) {
            if ((k & 255) == 255 && k - 256 >= first && bmax && bmax[k >> 8] < th) {
                int b = k >> 8;
                while ((b << 8) + 255 - 256 >= first && bmax[b] < th) --b;
                k = (b << 8) + 255; continue;
            }
            --k;
        }
        return k;
    };
    // Window sums of the envelope (the threshold of each token).  A token's window depends only on its t0 / t1 as
    // they stand before this loop (iteration j rewrites tokens[j] only, after its own sum), so all sums are taken
    // first.  Each one must stay a sequential left-to-right f32 sum (the reference's rounding), i.e. a 4-cycle
    // dependency chain per add — eight tokens are therefore summed side by side, eight independent chains.
    static const bool dbg_ts = getenv("WMI_DEBUG_EMIT") != nullptr;
    const int64_t ts0 = time_us();
    std::vector<float> win_sum(n, 0.0f);
    {
        std::vector<int> idx;
        for (int j = 0; j < n; ++j) if (tokens[j].id < v.eot) idx.push_back(j);
        // one task per token on the worker pool (up to 8 threads), each an exact blocked sum (seq_sum_f32: ~10x the plain loop).
        // Before: eight chains side by side per task, 90 us for the 17 tokens of a 30 s chunk; the longest window set the time.
        auto sum_one = [&](int t) {
            const int j = idx[t];
            const int a0 = std::max(ts_to_sample(tokens[j].t0, n_samples) - hw, 0);
            const int a1 = std::min(ts_to_sample(tokens[j].t1, n_samples) + hw, n_samples);
            win_sum[j] = a1 > a0 ? seq_sum_f32(en + a0, a1 - a0) : 0.0f;
        };
        pool_run((int) idx.size(), sum_one);
    }
    const int64_t ts1 = time_us();
    struct TsReport { bool on; int64_t a, b; ~TsReport() { if (on) fprintf(stderr, "[wmi] token timestamps: window sums %lld us, walks %lld us\n", (long long) (b - a), (long long) (time_us() - b)); } } ts_report{dbg_ts, ts0, ts1};
    for (int j = 0; j < n; ++j) {
        if (tokens[j].id >= v.eot) continue;
        int s0 = ts_to_sample(tokens[j].t0, n_samples), s1 = ts_to_sample(tokens[j].t1, n_samples);
        const int ss0 = std::max(s0 - hw, 0), ss1 = std::min(s1 + hw, n_samples);
        const int ns = ss1 - ss0;
        const float sum = win_sum[j];
        const float thold = 0.5 * sum / ns;
        {
            int k2 = s0;
            if (en[k2] > thold && j > 0) {
                k2 = walk_down_while_above(k2, thold);
                tokens[j].t0 = sample_to_ts(k2);
                if (tokens[j].t0 < tokens[j - 1].t1) tokens[j].t0 = tokens[j - 1].t1; else s0 = k2;
            } else {
                k2 = walk_up_while_below(k2, thold, s1);
                s0 = k2;
                tokens[j].t0 = sample_to_ts(k2);
            }
        }
        {
            int k2 = s1;
            if (en[k2] > thold) {
                k2 = walk_up_while_above(k2, thold, n_samples - 1);
                tokens[j].t1 = sample_to_ts(k2);
                // The reference tests `j < ns - 1` (window length, not token count) and so reads tokens[n] — one past
                // the end — for the last token (W/whisper.cpp:6561); what it finds there is heap garbage (usually 0,
                // which zeroes the last token's t1).  That read is undefined behaviour and is NOT reproduced: the last
                // token keeps its own end.  tests/test_gpu_parity.py exempts exactly this field.
                if (j < ns - 1 && j + 1 < n && tokens[j].t1 > tokens[j + 1].t0) tokens[j].t1 = tokens[j + 1].t0; else s1 = k2;
            } else {
                k2 = walk_down_while_below(k2, thold, s0);
                s1 = k2;
                tokens[j].t1 = sample_to_ts(k2);
            }
        }
    }
}

int wrap_segment(whisper_context & ctx, State & st, int max_len, bool split_on_word) {          // W/whisper.cpp:4430-4484
    const Vocab & v = ctx.model.vocab;
    Segment seg = st.result_all.back();
    int res = 1, acc = 0;
    std::string text;
    for (int i = 0; i < (int) seg.tokens.size(); ++i) {
        const whisper_token_data & tok = seg.tokens[i];
        if (tok.id >= v.eot) continue;
        const char * txt = tok_str(ctx, tok.id);
        const int cur = (int) strlen(txt);
        const bool may_split = !split_on_word || txt[0] == ' ';
        if (acc + cur > max_len && i > 0 && may_split) {
            Segment & last = st.result_all.back();
            last.text = std::move(text); last.t1 = tok.t0; last.tokens.resize(i); last.speaker_turn_next = false;
            Segment next{tok.t0, seg.t1, std::string(), {}, seg.speaker_turn_next};
            next.tokens.assign(seg.tokens.begin() + i, seg.tokens.end());
            st.result_all.push_back(std::move(next));
            acc = 0; text.clear();
            seg = st.result_all.back();
            i = -1;
            ++res;
        } else {
            acc += cur; text += txt;
        }
    }
    st.result_all.back().text = std::move(text);
    return res;
}

} // namespace wmi
