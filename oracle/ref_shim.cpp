// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// Shim that turns the reference's own whisper.cpp (v1.5.4, vendored under
// /root/reference/thirdparty/whisper.cpp) into a checker library with a few extra
// accessors for tensors that the public whisper.h does not expose.
//
// The reference translation unit is pulled in by the compiler from where it lies
// (-I$(REF) in oracle/Makefile); no reference source text is stored in this repository.
// Build output goes to oracle/_ref/ only (git-ignored, travels to the GPU box as a binary).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the result.

#include "whisper.cpp"   // resolved through -I<reference>/thirdparty/whisper.cpp

#include <cstring>

extern "C" {

// log-mel produced by the last whisper_pcm_to_mel / whisper_full call.
// layout: [n_mel][n_len] f32 (reference: W/whisper.cpp:2779)
int ref_mel_dims(struct whisper_context * ctx, int * n_len, int * n_len_org, int * n_mel) {
    const auto & mel = ctx->state->mel;
    *n_len = mel.n_len; *n_len_org = mel.n_len_org; *n_mel = mel.n_mel;
    return (int) mel.data.size();
}

int ref_mel_copy(struct whisper_context * ctx, float * dst, int n) {
    const auto & mel = ctx->state->mel;
    if (n > (int) mel.data.size()) n = (int) mel.data.size();
    memcpy(dst, mel.data.data(), sizeof(float)*n);
    return n;
}

static int copy_f32_tensor(struct ggml_tensor * t, float * dst, int n) {
    if (!t) return -1;
    const int ne = (int) ggml_nelements(t);
    if (dst == nullptr) return ne;
    if (n > ne) n = ne;
    ggml_backend_tensor_get(t, dst, 0, sizeof(float)*n);
    return n;
}

// conv front-end output, ggml shape [n_ctx (fast), n_state]  (W/whisper.cpp:1724)
int ref_embd_conv(struct whisper_context * ctx, float * dst, int n) {
    return copy_f32_tensor(ctx->state->embd_conv, dst, n);
}

// encoder output, ggml shape [n_state (fast), n_ctx]  (W/whisper.cpp:1985)
int ref_embd_enc(struct whisper_context * ctx, float * dst, int n) {
    return copy_f32_tensor(ctx->state->embd_enc, dst, n);
}

// raw f16 KV caches (bit patterns). which: 0 cross.k 1 cross.v 2 self.k 3 self.v
int ref_kv_copy(struct whisper_context * ctx, int which, uint16_t * dst, int n) {
    struct ggml_tensor * t = nullptr;
    switch (which) {
        case 0: t = ctx->state->kv_cross.k; break;
        case 1: t = ctx->state->kv_cross.v; break;
        case 2: t = ctx->state->kv_self.k;  break;
        case 3: t = ctx->state->kv_self.v;  break;
    }
    if (!t) return -1;
    const int ne = (int) ggml_nelements(t);
    if (dst == nullptr) return ne;
    if (n > ne) n = ne;
    ggml_backend_tensor_get(t, dst, 0, sizeof(uint16_t)*n);
    return n;
}

// set the experimental audio ctx the way whisper_full does (W/whisper.cpp:5102)
void ref_set_audio_ctx(struct whisper_context * ctx, int n) {
    ctx->state->exp_n_audio_ctx = n;
}

// decoder i's post-filter arrays after the last whisper_full (probs / logits / logprobs)
int ref_decoder_probs(struct whisper_context * ctx, int j, int which, float * dst, int n) {
    auto & d = ctx->state->decoders[j];
    const std::vector<float> & v = which == 0 ? d.probs : which == 1 ? d.logits : d.logprobs;
    if (n > (int) v.size()) n = (int) v.size();
    memcpy(dst, v.data(), sizeof(float)*n);
    return n;
}

// timing counters of the reference state (W/whisper.cpp:770-783), microseconds
void ref_timings(struct whisper_context * ctx, int64_t * out6, int32_t * cnt5) {
    auto * s = ctx->state;
    out6[0] = s->t_mel_us;    out6[1] = s->t_encode_us; out6[2] = s->t_decode_us;
    out6[3] = s->t_batchd_us; out6[4] = s->t_prompt_us; out6[5] = s->t_sample_us;
    cnt5[0] = s->n_encode; cnt5[1] = s->n_decode; cnt5[2] = s->n_batchd; cnt5[3] = s->n_prompt; cnt5[4] = s->n_sample;
}

// run process_logits on caller-provided raw logits with a given token history
// (host-logic checker for the product's own filter implementation)
int ref_process_logits(struct whisper_context * ctx, struct whisper_full_params params,
                       const float * raw_logits, const whisper_token * hist, int n_hist,
                       int has_ts, int seek_delta, float temperature,
                       float * out_logits, float * out_logprobs, float * out_probs) {
    auto & st = *ctx->state;
    auto & d  = st.decoders[0];
    const int nv = ctx->vocab.n_vocab;
    st.logits.assign(raw_logits, raw_logits + nv);
    d.i_batch = 0;
    d.sequence.tokens.clear();
    for (int i = 0; i < n_hist; ++i) {
        whisper_token_data td = { hist[i], 0, 0.0f, 0.0f, 0.0f, 0.0f, -1, -1, 0.0f };
        d.sequence.tokens.push_back(td);
    }
    d.has_ts = has_ts != 0;
    d.seek_delta = seek_delta;
    // grammar-constrained decoding: the parse state after the history (what whisper_full keeps per decoder, :5228-5232, :5457)
    if (params.grammar_rules != nullptr) {
        d.grammar = whisper_grammar_init(params.grammar_rules, params.n_grammar_rules, params.i_start_rule);
        for (int i = 0; i < n_hist; ++i) whisper_grammar_accept_token(*ctx, d.grammar, hist[i]);
    } else {
        d.grammar = {};
    }
    whisper_process_logits(*ctx, st, d, params, temperature);
    memcpy(out_logits,   d.logits.data(),   sizeof(float)*nv);
    memcpy(out_logprobs, d.logprobs.data(), sizeof(float)*nv);
    memcpy(out_probs,    d.probs.data(),    sizeof(float)*nv);
    return nv;
}

// draw n tokens from decoder 0's rng the way whisper_sample_token(best=false) does
int ref_sample_draws(struct whisper_context * ctx, const float * probs, const float * logprobs, int n_draw, int reseed,
                     whisper_token_data * out) {
    auto & d = ctx->state->decoders[0];
    const int nv = ctx->vocab.n_vocab;
    d.probs.assign(probs, probs + nv);
    d.logprobs.assign(logprobs, logprobs + nv);
    if (reseed) d.rng = std::mt19937(0);
    for (int i = 0; i < n_draw; ++i) out[i] = whisper_sample_token(*ctx, d, false);
    return n_draw;
}

// tokenizer of the reference (W/whisper.cpp:2899-2947) is already public as whisper_tokenize.

// the reference's GELU evaluated on every f16 bit pattern (= its ggml_table_gelu_f16, W/ggml.c:2229-2231)
int ref_gelu_table(uint16_t * out) {
    struct ggml_init_params ip = { (size_t) 16*1024*1024, nullptr, false };
    struct ggml_context * g = ggml_init(ip);
    if (!g) return -1;
    struct ggml_tensor * x = ggml_new_tensor_1d(g, GGML_TYPE_F32, 65536);
    for (int i = 0; i < 65536; ++i) ((float *) x->data)[i] = ggml_fp16_to_fp32((ggml_fp16_t) i);
    struct ggml_tensor * y = ggml_gelu(g, x);
    struct ggml_cgraph * gf = ggml_new_graph(g);
    ggml_build_forward_expand(gf, y);
    ggml_graph_compute_with_ctx(g, gf, 1);
    for (int i = 0; i < 65536; ++i) out[i] = ggml_fp32_to_fp16(((float *) y->data)[i]);
    ggml_free(g);
    return 65536;
}

size_t ref_sizeof_full_params(void) { return sizeof(struct whisper_full_params); }
size_t ref_sizeof_token_data(void)  { return sizeof(struct whisper_token_data); }

} // extern "C"
