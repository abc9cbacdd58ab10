// Grammar-constrained decoding (whisper_full_params.grammar_rules; W/whisper.cpp:3876-4290, applied at :4684-4706,
// :5228-5232, :5332, :5400, :5457).  A grammar is a set of rules, each a run of elements (W/whisper.h:116-140); the parse
// state is a set of pushdown stacks whose tops rest on character ranges.  Before sampling, every text token whose code
// points no stack can consume is penalised by grammar_penalty; after sampling, the stacks consume the token's code points.
//
// Positions are (rule, offset) indices rather than pointers into the rule arrays, so a Grammar is a plain value: beam
// candidates copy it freely.  Host CPU code like the rest of the sampling logic (host_logic.cpp).

#include "wmi.h"

#include <cstring>

namespace wmi {

namespace {

// element types, W/whisper.h:117-140
constexpr whisper_gretype G_END = WHISPER_GRETYPE_END, G_ALT = WHISPER_GRETYPE_ALT, G_RULE_REF = WHISPER_GRETYPE_RULE_REF,
                          G_CHAR = WHISPER_GRETYPE_CHAR, G_CHAR_NOT = WHISPER_GRETYPE_CHAR_NOT,
                          G_CHAR_RNG_UPPER = WHISPER_GRETYPE_CHAR_RNG_UPPER, G_CHAR_ALT = WHISPER_GRETYPE_CHAR_ALT;

typedef std::vector<std::vector<whisper_grammar_element>> Rules;
typedef std::vector<GrammarPos> Stack;

inline const whisper_grammar_element & at(const Rules & r, GrammarPos p) { return r[p.rule][p.off]; }
inline GrammarPos next(GrammarPos p, int by = 1) { return GrammarPos{ p.rule, p.off + by }; }
inline bool ends_alternative(const whisper_grammar_element & e) { return e.type == G_END || e.type == G_ALT; }

// UTF-8 decoding that may start and end inside a sequence; the result carries a terminating 0.  An invalid byte gives
// the single code point 0 and n_remain = -1 (W/whisper.cpp:3878-3935).
struct Decoded { std::vector<uint32_t> cp; PartialUtf8 tail; };
Decoded decode_utf8(const std::string & text, PartialUtf8 start) {
    static const int seq_len[16] = { 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 2, 2, 3, 4 };       // by the first byte's high nibble
    Decoded d;
    uint32_t value = start.value;
    int n_remain = start.n_remain;
    size_t i = 0;
    const size_t n = strlen(text.c_str());                 // the reference walks a C string: an embedded NUL ends it
    for (; i < n && n_remain > 0; ++i, --n_remain) {
        const uint8_t b = (uint8_t) text[i];
        if ((b >> 6) != 2) { d.cp.assign(1, 0); d.tail = PartialUtf8{ 0, -1 }; return d; }
        value = (value << 6) + (b & 0x3F);
    }
    if (start.n_remain > 0 && n_remain == 0) d.cp.push_back(value);
    while (i < n) {
        const uint8_t first = (uint8_t) text[i];
        n_remain = seq_len[first >> 4] - 1;
        if (n_remain < 0) { d.cp.assign(1, 0); d.tail = PartialUtf8{ 0, n_remain }; return d; }
        value = first & ((1u << (7 - n_remain)) - 1);
        ++i;
        for (; i < n && n_remain > 0; ++i, --n_remain) value = (value << 6) + ((uint8_t) text[i] & 0x3F);
        if (n_remain == 0) d.cp.push_back(value);
    }
    d.cp.push_back(0);
    d.tail = PartialUtf8{ value, n_remain };
    return d;
}

// does chr satisfy the (possibly negated) character class starting at p?  second = the element after the class
std::pair<bool, GrammarPos> match_char(const Rules & r, GrammarPos p, uint32_t chr) {
    const bool positive = at(r, p).type == G_CHAR;
    bool found = false;
    do {
        if (at(r, next(p)).type == G_CHAR_RNG_UPPER) { found = found || (at(r, p).value <= chr && chr <= at(r, next(p)).value); p = next(p, 2); }
        else                                          { found = found || at(r, p).value == chr; p = next(p); }
    } while (at(r, p).type == G_CHAR_ALT);
    return { found == positive, p };
}

// could some completion of the partial sequence satisfy the class at p?  (W/whisper.cpp:3975-4019)
bool match_partial_char(const Rules & r, GrammarPos p, PartialUtf8 partial) {
    const bool positive = at(r, p).type == G_CHAR;
    const int n_remain = partial.n_remain;
    if (n_remain < 0 || (n_remain == 1 && partial.value < 2)) return false;       // invalid, or an overlong 7-bit char
    uint32_t low = partial.value << (n_remain * 6);
    const uint32_t high = low | ((1u << (n_remain * 6)) - 1);
    if (low == 0) { if (n_remain == 2) low = 1u << 11; else if (n_remain == 3) low = 1u << 16; }
    do {
        if (at(r, next(p)).type == G_CHAR_RNG_UPPER) { if (at(r, p).value <= high && low <= at(r, next(p)).value) return positive; p = next(p, 2); }
        else                                          { if (low <= at(r, p).value && at(r, p).value <= high) return positive; p = next(p); }
    } while (at(r, p).type == G_CHAR_ALT);
    return !positive;
}

// expand rule references on top of `stack` until every resulting stack is empty or rests on a character class
void advance_stack(const Rules & r, const Stack & stack, std::vector<Stack> & out) {
    if (stack.empty()) { out.push_back(stack); return; }
    const GrammarPos top = stack.back();
    const int type = at(r, top).type;
    if (type == G_CHAR || type == G_CHAR_NOT) { out.push_back(stack); return; }
    if (type != G_RULE_REF) return;                        // malformed grammar (the reference asserts): drop the stack
    const int rule = (int) at(r, top).value;
    if (rule < 0 || rule >= (int) r.size()) return;
    GrammarPos sub{ rule, 0 };
    for (;;) {                                             // one new stack per alternative of the referenced rule
        Stack ns(stack.begin(), stack.end() - 1);
        if (!ends_alternative(at(r, next(top)))) ns.push_back(next(top));          // what follows the reference
        if (!ends_alternative(at(r, sub))) ns.push_back(sub);                      // a non-empty alternative
        advance_stack(r, ns, out);
        while (!ends_alternative(at(r, sub))) sub = next(sub);
        if (at(r, sub).type != G_ALT) break;
        sub = next(sub);
    }
}

std::vector<Stack> accept_char(const Rules & r, const std::vector<Stack> & stacks, uint32_t chr) {
    std::vector<Stack> out;
    for (const Stack & s : stacks) {
        if (s.empty()) continue;
        const auto m = match_char(r, s.back(), chr);
        if (!m.first) continue;
        Stack ns(s.begin(), s.end() - 1);
        if (!ends_alternative(at(r, m.second))) ns.push_back(m.second);
        advance_stack(r, ns, out);
    }
    return out;
}

struct Candidate { int32_t id; const uint32_t * cp; PartialUtf8 partial; };

std::vector<Candidate> rejects_for_all(const Rules & r, const std::vector<Stack> & stacks, const std::vector<Candidate> & cands);

// the candidates this one stack cannot take (W/whisper.cpp:4114-4163)
std::vector<Candidate> rejects_for_stack(const Rules & r, const Stack & stack, const std::vector<Candidate> & cands) {
    std::vector<Candidate> rejects;
    if (stack.empty()) {                                   // the grammar is complete here: only an exhausted token fits
        for (const Candidate & c : cands) if (*c.cp != 0 || c.partial.n_remain != 0) rejects.push_back(c);
        return rejects;
    }
    const GrammarPos top = stack.back();
    std::vector<Candidate> onward;
    for (const Candidate & c : cands) {
        if (*c.cp == 0) { if (c.partial.n_remain != 0 && !match_partial_char(r, top, c.partial)) rejects.push_back(c); }
        else if (match_char(r, top, *c.cp).first) onward.push_back(Candidate{ c.id, c.cp + 1, c.partial });
        else rejects.push_back(c);
    }
    const GrammarPos after = match_char(r, top, 0).second;
    Stack ns(stack.begin(), stack.end() - 1);
    if (!ends_alternative(at(r, after))) ns.push_back(after);
    std::vector<Stack> next_stacks;
    advance_stack(r, ns, next_stacks);
    for (const Candidate & c : rejects_for_all(r, next_stacks, onward)) rejects.push_back(Candidate{ c.id, c.cp - 1, c.partial });
    return rejects;
}

// a candidate is rejected when every stack rejects it
std::vector<Candidate> rejects_for_all(const Rules & r, const std::vector<Stack> & stacks, const std::vector<Candidate> & cands) {
    if (cands.empty() || stacks.empty()) return {};
    std::vector<Candidate> rejects = rejects_for_stack(r, stacks.front(), cands);
    for (size_t i = 1; i < stacks.size(); ++i) rejects = rejects_for_stack(r, stacks[i], rejects);
    return rejects;
}

} // namespace

Grammar grammar_init(const whisper_grammar_element ** rules, size_t n_rules, size_t i_start_rule) {
    Grammar g;
    g.rules.resize(n_rules);
    for (size_t i = 0; i < n_rules; ++i) {
        for (const whisper_grammar_element * e = rules[i]; e->type != G_END; ++e) g.rules[i].push_back(*e);
        g.rules[i].push_back(whisper_grammar_element{ G_END, 0 });
    }
    if (i_start_rule >= n_rules) return g;
    GrammarPos pos{ (int) i_start_rule, 0 };
    for (;;) {                                             // one initial stack per alternative of the start rule
        Stack s;
        if (!ends_alternative(at(g.rules, pos))) s.push_back(pos);
        advance_stack(g.rules, s, g.stacks);
        while (!ends_alternative(at(g.rules, pos))) pos = next(pos);
        if (at(g.rules, pos).type != G_ALT) break;
        pos = next(pos);
    }
    return g;
}

void grammar_penalise(const whisper_context & ctx, const Grammar & g, float penalty, std::vector<float> & logits) {
    if (g.rules.empty() || g.stacks.empty()) return;
    const Vocab & v = ctx.model.vocab;
    std::vector<Decoded> decoded;
    decoded.reserve(v.eot);
    std::vector<Candidate> cands;
    cands.reserve(v.eot);
    for (int32_t id = 0; id < v.eot; ++id) {
        const std::string & text = v.id_to_token[id];
        if (text.empty()) continue;
        decoded.push_back(decode_utf8(text, g.partial));
        cands.push_back(Candidate{ id, nullptr, decoded.back().tail });
    }
    for (size_t i = 0; i < cands.size(); ++i) cands[i].cp = decoded[i].cp.data();      // after the last push_back: stable addresses
    for (const Candidate & c : rejects_for_all(g.rules, g.stacks, cands)) logits[c.id] -= penalty;
}

void grammar_accept_token(const whisper_context & ctx, Grammar & g, int32_t token) {
    if (g.rules.empty() || g.stacks.empty()) return;
    const std::string & text = ctx.model.vocab.id_to_token[token];
    if (text.rfind("[_", 0) == 0) return;                  // special tokens do not move the grammar
    const Decoded d = decode_utf8(text, g.partial);
    for (size_t i = 0; i + 1 < d.cp.size(); ++i) g.stacks = accept_char(g.rules, g.stacks, d.cp[i]);
    g.partial = d.tail;
}

} // namespace wmi
