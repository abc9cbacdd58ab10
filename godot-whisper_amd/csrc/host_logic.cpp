// Host-side pieces of the hot path that stay on the CPU:
// unified KV-cache bookkeeping, logit filters, sampling, sequence scoring, the prompt tokenizer and
// the language table.  Behaviour follows whisper.cpp v1.5.4 decision for decision, because greedy
// and sampled token streams must come out identical when fed identical logits
// (tests/test_host_logic.py drives these against the compiled reference).

#include "wmi.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <limits>
#include <regex>

namespace wmi {

// ------------------------------------------------------------------ unified KV cache (W/whisper.cpp:938-1054)
void Batch::prep_legacy(const int32_t * tokens, int n, int n_past, int seq) {
    n_tokens = n;
    if ((int) token.size() < n) { token.resize(n); pos.resize(n); seq_id.resize(n); logits.resize(n); }
    for (int i = 0; i < n; ++i) {
        if (tokens) token[i] = tokens[i];
        pos[i] = n_past + i; seq_id[i] = seq; logits[i] = 0;
    }
    if (n > 0) logits[n - 1] = 1;
}

bool kv_find_slot(KVCache & c, const Batch & b) {
    const uint32_t n_ctx = c.size, n_tokens = (uint32_t) b.n_tokens;
    if (n_tokens > n_ctx) { WMI_ERR("%s: n_tokens=%d > n_ctx=%d\n", __func__, n_tokens, n_ctx); return false; }
    uint32_t tested = 0;
    for (;;) {
        if (c.head + n_tokens > n_ctx) { tested += n_ctx - c.head; c.head = 0; continue; }
        uint32_t busy = n_tokens;                                   // first occupied cell in the window, if any
        for (uint32_t i = 0; i < n_tokens; ++i) if (c.cells[c.head + i].pos >= 0) { busy = i; break; }
        if (busy == n_tokens) break;
        c.head += busy + 1; tested += busy + 1;
        if (tested >= n_ctx) return false;
    }
    for (uint32_t i = 0; i < n_tokens; ++i) {
        c.cells[c.head + i].pos = b.pos[i];
        c.cells[c.head + i].seq.insert(b.seq_id[i]);
    }
    return true;
}

int kv_cell_max(const KVCache & c) {
    for (uint32_t i = c.size - 1; i > 0; --i) if (c.cells[i].pos >= 0 && !c.cells[i].seq.empty()) return (int) i + 1;
    return 1;
}

void kv_clear(KVCache & c) {
    for (auto & cell : c.cells) { cell.pos = -1; cell.seq.clear(); }
    c.head = 0;
}

void kv_seq_rm(KVCache & c, int32_t seq, int32_t p0, int32_t p1) {
    uint32_t new_head = c.size;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = std::numeric_limits<int32_t>::max();
    for (uint32_t i = 0; i < c.size; ++i) {
        KVCell & cell = c.cells[i];
        if (cell.pos < p0 || cell.pos >= p1) continue;
        if (seq < 0) cell.seq.clear();
        else if (cell.has(seq)) cell.seq.erase(seq);
        else continue;
        if (cell.seq.empty()) { cell.pos = -1; if (new_head == c.size) new_head = i; }
    }
    if (new_head != c.size) c.head = new_head;
}

void kv_seq_cp(KVCache & c, int32_t src, int32_t dst, int32_t p0, int32_t p1) {
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = std::numeric_limits<int32_t>::max();
    c.head = 0;
    for (auto & cell : c.cells) if (cell.has(src) && cell.pos >= p0 && cell.pos < p1) cell.seq.insert(dst);
}

// ------------------------------------------------------------------ languages (W/whisper.cpp:244-345)
namespace {
struct Lang { const char * code; const char * name; };
const Lang k_langs[] = {
    {"en","english"},{"zh","chinese"},{"de","german"},{"es","spanish"},{"ru","russian"},{"ko","korean"},{"fr","french"},
    {"ja","japanese"},{"pt","portuguese"},{"tr","turkish"},{"pl","polish"},{"ca","catalan"},{"nl","dutch"},{"ar","arabic"},
    {"sv","swedish"},{"it","italian"},{"id","indonesian"},{"hi","hindi"},{"fi","finnish"},{"vi","vietnamese"},{"he","hebrew"},
    {"uk","ukrainian"},{"el","greek"},{"ms","malay"},{"cs","czech"},{"ro","romanian"},{"da","danish"},{"hu","hungarian"},
    {"ta","tamil"},{"no","norwegian"},{"th","thai"},{"ur","urdu"},{"hr","croatian"},{"bg","bulgarian"},{"lt","lithuanian"},
    {"la","latin"},{"mi","maori"},{"ml","malayalam"},{"cy","welsh"},{"sk","slovak"},{"te","telugu"},{"fa","persian"},
    {"lv","latvian"},{"bn","bengali"},{"sr","serbian"},{"az","azerbaijani"},{"sl","slovenian"},{"kn","kannada"},
    {"et","estonian"},{"mk","macedonian"},{"br","breton"},{"eu","basque"},{"is","icelandic"},{"hy","armenian"},{"ne","nepali"},
    {"mn","mongolian"},{"bs","bosnian"},{"kk","kazakh"},{"sq","albanian"},{"sw","swahili"},{"gl","galician"},{"mr","marathi"},
    {"pa","punjabi"},{"si","sinhala"},{"km","khmer"},{"sn","shona"},{"yo","yoruba"},{"so","somali"},{"af","afrikaans"},
    {"oc","occitan"},{"ka","georgian"},{"be","belarusian"},{"tg","tajik"},{"sd","sindhi"},{"gu","gujarati"},{"am","amharic"},
    {"yi","yiddish"},{"lo","lao"},{"uz","uzbek"},{"fo","faroese"},{"ht","haitian creole"},{"ps","pashto"},{"tk","turkmen"},
    {"nn","nynorsk"},{"mt","maltese"},{"sa","sanskrit"},{"lb","luxembourgish"},{"my","myanmar"},{"bo","tibetan"},
    {"tl","tagalog"},{"mg","malagasy"},{"as","assamese"},{"tt","tatar"},{"haw","hawaiian"},{"ln","lingala"},{"ha","hausa"},
    {"ba","bashkir"},{"jw","javanese"},{"su","sundanese"},{"yue","cantonese"},
};
constexpr int k_n_langs = (int) (sizeof(k_langs) / sizeof(k_langs[0]));
} // namespace

int lang_count()  { return k_n_langs; }
int lang_max_id() { return k_n_langs - 1; }
int lang_id(const char * lang) {
    if (!lang) return -1;
    for (int i = 0; i < k_n_langs; ++i) if (!strcmp(k_langs[i].code, lang)) return i;
    for (int i = 0; i < k_n_langs; ++i) if (!strcmp(k_langs[i].name, lang)) return i;
    WMI_ERR("%s: unknown language '%s'\n", __func__, lang);
    return -1;
}
const char * lang_str(int id) {
    if (id >= 0 && id < k_n_langs) return k_langs[id].code;
    WMI_ERR("%s: unknown language id %d\n", __func__, id);
    return nullptr;
}

const char * lang_str_full(int id) {             // W/whisper.cpp:3558-3567
    if (id >= 0 && id < k_n_langs) return k_langs[id].name;
    WMI_ERR("%s: unknown language id %d\n", __func__, id);
    return nullptr;
}

// ------------------------------------------------------------------ tokenizer (W/whisper.cpp:2899-2947)
std::vector<int32_t> tokenize(const Vocab & vocab, const std::string & text) {
    std::vector<std::string> words;
    {
        static const std::regex re(R"('s|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+)");
        std::string rest = text;
        std::smatch m;
        while (std::regex_search(rest, m, re)) {
            for (const auto & x : m) words.push_back(x);
            rest = m.suffix();
        }
    }
    std::vector<int32_t> out;
    for (const std::string & word : words) {
        const int n = (int) word.size();
        for (int i = 0; i < n;) {                              // greedy longest match against the vocabulary
            int j = n;
            for (; j > i; --j) {
                auto it = vocab.token_to_id.find(word.substr(i, j - i));
                if (it != vocab.token_to_id.end()) { out.push_back(it->second); break; }
            }
            if (j > i) i = j; else { WMI_ERR("unknown token\n"); ++i; }
        }
    }
    return out;
}

// ------------------------------------------------------------------ logit filters (W/whisper.cpp:4493-4775)
namespace {

const char * const k_non_speech[] = {
    "\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^",
    "_", "`", "{", "|", "}", "~", "「", "」", "『", "』", "<<", ">>", "<<<", ">>>", "--",
    "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))", "[[", "]]", "{{", "}}", "♪♪",
    "♪♪♪", "♩", "♪", "♫", "♬", "♭", "♮", "♯",
};

constexpr float NEG_INF = -INFINITY;

void log_softmax(const std::vector<float> & logits, std::vector<float> & logprobs) {
    const int n = (int) logits.size();
    const float mx = *std::max_element(logits.begin(), logits.end());
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) if (logits[i] > NEG_INF) acc += expf(logits[i] - mx);
    const float lse = logf(acc) + mx;
    for (int i = 0; i < n; ++i) logprobs[i] = logits[i] > NEG_INF ? logits[i] - lse : NEG_INF;
}

} // namespace

void process_logits(whisper_context & ctx, Decoder & dec, const whisper_full_params & params, float temperature) {
    const Vocab & v = ctx.model.vocab;
    State & st = *ctx.state;
    const auto & hist = dec.sequence.tokens;
    const bool is_initial = hist.empty();
    const int n = v.n_vocab;

    auto & logits = dec.logits; auto & logprobs = dec.logprobs; auto & probs = dec.probs;
    logits.resize(n); logprobs.resize(n); probs.resize(n);
    memcpy(logits.data(), st.logits.data() + (size_t) dec.i_batch * n, (size_t) n * sizeof(float));
    if (temperature > 0.0f) for (int i = 0; i < n; ++i) logits[i] /= temperature;

    auto ban = [&](int id) { if (id >= 0 && id < n) logits[id] = NEG_INF; };
    auto ban_range = [&](int a, int b) { for (int i = std::max(a, 0); i < std::min(b, n); ++i) logits[i] = NEG_INF; };

    if (params.suppress_blank && is_initial) {
        ban(v.eot);
        auto sp = v.token_to_id.find(" ");
        if (sp != v.token_to_id.end()) ban(sp->second);
    }
    ban(v.not_);
    if (params.no_timestamps) ban_range(v.beg, n);
    ban(v.sot); ban(v.nosp);
    if (!params.tdrz_enable) ban(v.solm);
    ban(v.translate); ban(v.transcribe); ban(v.prev);
    for (int i = 0; i < lang_count(); ++i) ban(v.sot + 1 + i);       // always all 100 ids

    if (params.logits_filter_callback)
        params.logits_filter_callback(&ctx, (whisper_state *) ctx.state.get(), hist.data(), (int) hist.size(), logits.data(),
                                      params.logits_filter_callback_user_data);

    if (params.suppress_non_speech_tokens) {
        for (const char * t : k_non_speech) {
            const std::string forms[2] = { std::string(t), " " + std::string(t) };
            for (const auto & f : forms) { auto it = v.token_to_id.find(f); if (it != v.token_to_id.end()) ban(it->second); }
        }
        for (const char * t : { " -", " '" }) { auto it = v.token_to_id.find(t); if (it != v.token_to_id.end()) ban(it->second); }
    }

    // timestamps come in pairs, except directly before EOT
    {
        const size_t k = hist.size();
        const bool last_ts = k > 0 && hist[k - 1].id >= v.beg;
        const bool penult_ts = k < 2 || hist[k - 2].id >= v.beg;
        if (last_ts) { if (penult_ts) ban_range(v.beg, n); else ban_range(0, v.eot); }
    }
    if (is_initial && params.max_initial_ts > 0.0f) {
        const float precision = float(WHISPER_CHUNK_SIZE) / ctx.model.hp.n_audio_ctx;
        const int tid0 = (int) std::round(params.max_initial_ts / precision);
        ban_range(v.beg + tid0 + 1, n);
    }
    if (dec.has_ts) ban_range(v.beg, v.beg + dec.seek_delta / 2);     // timestamps must not decrease

    log_softmax(logits, logprobs);

    // if the timestamp mass beats every text token, force a timestamp
    {
        float ts_logprob = NEG_INF;
        {
            const float mx = *std::max_element(logprobs.begin() + v.beg, logprobs.end());
            float acc = 0.0f;
            for (int i = v.beg; i < n; ++i) if (logprobs[i] > NEG_INF) acc += expf(logprobs[i] - mx);
            if (acc > 0.0f) ts_logprob = logf(acc) + mx;
        }
        const float max_text = *std::max_element(logprobs.begin(), logprobs.begin() + v.beg);
        if (ts_logprob > max_text) for (int i = 0; i < v.beg; ++i) { logits[i] = NEG_INF; logprobs[i] = NEG_INF; }
        else if (params.n_grammar_rules > 0) {          // W/whisper.cpp:4684-4706: penalise what the grammar cannot take, renormalise
            grammar_penalise(ctx, dec.grammar, params.grammar_penalty, logits);
            log_softmax(logits, logprobs);
        }
    }
    for (int i = 0; i < n; ++i) probs[i] = logits[i] == NEG_INF ? 0.0f : expf(logprobs[i]);
}

// ------------------------------------------------------------------ sampling (W/whisper.cpp:4777-4909)
namespace {
void timestamp_stats(const Vocab & v, const std::vector<float> & probs, int32_t & tid, float & pt, float & ptsum) {
    double sum_ts = 0.0, max_ts = 0.0;
    for (int i = v.beg; i < v.n_vocab; ++i) {
        if (probs[i] == NEG_INF) continue;
        sum_ts += probs[i];
        if (max_ts < probs[i]) { max_ts = probs[i]; tid = i; }
    }
    pt = (float) (max_ts / (sum_ts + 1e-10));
    ptsum = (float) sum_ts;
}
} // namespace

whisper_token_data sample_token(whisper_context & ctx, Decoder & dec, bool best) {
    const Vocab & v = ctx.model.vocab;
    whisper_token_data r = { 0, 0, 0.0f, 0.0f, 0.0f, 0.0f, -1, -1, 0.0f };
    timestamp_stats(v, dec.probs, r.tid, r.pt, r.ptsum);
    if (best) {
        for (int i = 0; i < v.n_vocab; ++i) if (r.p < dec.probs[i]) { r.id = i; r.p = dec.probs[i]; r.plog = dec.logprobs[i]; }
    } else {
        std::discrete_distribution<> dist(dec.probs.begin(), dec.probs.end());
        r.id = dist(dec.rng); r.p = dec.probs[r.id]; r.plog = dec.logprobs[r.id];
    }
    if (r.id >= v.beg) { r.tid = r.id; r.pt = r.p; }
    ctx.state->n_sample++;
    return r;
}

std::vector<whisper_token_data> sample_token_topk(whisper_context & ctx, Decoder & dec, int k, bool count) {
    // v1.5.4 "top-k" = k independent draws from the full distribution; the partial sort
    // the reference performs first has no effect on the result and is not reproduced.
    const Vocab & v = ctx.model.vocab;
    int32_t tid = v.beg; float pt = 0.0f, ptsum = 0.0f;
    timestamp_stats(v, dec.probs, tid, pt, ptsum);
    std::discrete_distribution<> dist(dec.probs.begin(), dec.probs.end());
    std::vector<whisper_token_data> out;
    out.reserve(k);
    for (int i = 0; i < k; ++i) {
        const int id = dist(dec.rng);
        whisper_token_data t = { id, tid, dec.probs[id], dec.logprobs[id], pt, ptsum, -1, -1, 0.0f };
        if (t.id >= v.beg) { t.tid = t.id; t.pt = t.p; }
        out.push_back(t);
    }
    if (count) ctx.state->n_sample++;
    return out;
}

void sequence_score(const whisper_full_params & params, Sequence & seq) {       // W/whisper.cpp:4912-4958
    if (seq.result_len == 0) return;
    double sum = 0.0;
    for (int i = 0; i < seq.result_len; ++i) sum += seq.tokens[i].plog;
    seq.sum_logprobs = sum;
    seq.avg_logprobs = sum / seq.result_len;
    double penalty = seq.result_len;
    if (params.length_penalty > 0.0f) penalty = pow((5.0 + penalty) / 6.0, params.length_penalty);
    seq.score = sum / penalty;
    std::map<int32_t, int> counts;
    int cnt = 0;
    for (int i = std::max(0, seq.result_len - 32); i < seq.result_len; ++i) { counts[seq.tokens[i].id]++; cnt++; }
    double entropy = 0.0;
    for (const auto & kv : counts) { const double p = kv.second / (double) cnt; entropy -= p * log(p); }
    seq.entropy = entropy;
}

// ------------------------------------------------------------------ language detection (W/whisper.cpp:3569-3650)
int lang_auto_detect(whisper_context & ctx, int offset_ms, float * lang_probs) {
    State & st = *ctx.state;
    const int seek = offset_ms / 10;
    if (seek < 0) { WMI_ERR("%s: offset %dms is before the start of the audio\n", __func__, offset_ms); return -1; }
    if (seek >= st.mel.n_len_org) { WMI_ERR("%s: offset %dms is past the end of the audio (%dms)\n", __func__, offset_ms, st.mel.n_len_org * 10); return -2; }
    if (!encode(ctx, seek)) { WMI_ERR("%s: failed to encode\n", __func__); return -6; }
    const int32_t sot = ctx.model.vocab.sot;
    st.batch.prep_legacy(&sot, 1, 0, 0);
    kv_seq_rm(st.kv_self, 0, 0, -1);
    if (!decode(ctx, st.batch)) { WMI_ERR("%s: failed to decode\n", __func__); return -7; }
    std::vector<std::pair<double, int>> cand;
    for (int i = 0; i < lang_count(); ++i) cand.emplace_back((double) st.logits[sot + 1 + i], i);
    // the reference sorts a std::map iteration (keyed by language code) with std::sort; ties are
    // broken by that order, so reproduce it: sort ids by code first, then stable-sort by logit
    std::sort(cand.begin(), cand.end(), [](const std::pair<double, int> & a, const std::pair<double, int> & b) {
        return strcmp(lang_str(a.second), lang_str(b.second)) < 0; });
    std::stable_sort(cand.begin(), cand.end(), [](const std::pair<double, int> & a, const std::pair<double, int> & b) { return a.first > b.first; });
    const double mx = cand[0].first;
    double sum = 0.0;
    for (auto & c : cand) { c.first = exp(c.first - mx); sum += c.first; }
    for (auto & c : cand) c.first /= sum;
    if (lang_probs) for (const auto & c : cand) lang_probs[c.second] = (float) c.first;
    return cand[0].second;
}

} // namespace wmi
