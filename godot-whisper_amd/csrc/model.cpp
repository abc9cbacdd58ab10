// ggml Whisper model file -> host directory + device weight arena (SURVEY §8 row a13;
// format: SURVEY Appendix A, reference loader W/whisper.cpp:1102-1640).
//
// The caller's buffer is only borrowed during the call (the Godot host frees its PackedByteArray
// right after whisper_init_from_buffer_with_params, src/speech_to_text.cpp:338-345).  Payloads go
// host -> device through one staging pass.  f16 / f32 matrices are kept f16 and re-laid where a kernel wants a
// different operand order (conv taps, stacked q|k|v rows).  Block-quantised matrices (W/ggml-quants.h:10-47) keep
// their blocks: they are only permuted into the tile layout the kernels stream (kernels.h, k_quant.hip) —
// large-v3 q5_1 occupies ~1.1 GB of HBM instead of the 3.1 GB of an f16 expansion.
//
// The file is untrusted input: every length is checked against the buffer before it is used, every tensor must
// have exactly the rank, dimensions and element count the architecture implies (the reference rejects the same
// files with "wrong shape / size", W/whisper.cpp:1560-1600), and nothing is copied beyond the size reserved for it.

#include "wmi.h"

#include <cmath>
#include <cstring>
#include <new>

namespace wmi {

namespace {

struct Reader {
    const uint8_t * p; size_t n, off = 0; bool ok = true;
    bool eof() const { return off >= n; }
    template <typename T> T get() {
        T v{};
        if (off > n || sizeof(T) > n - off) { ok = false; off = n; return v; }
        memcpy(&v, p + off, sizeof(T)); off += sizeof(T); return v;
    }
    const uint8_t * bytes(size_t k) {
        if (off > n || k > n - off) { ok = false; off = n; return nullptr; }     // (off + k would wrap for a huge k)
        const uint8_t * r = p + off; off += k; return r;
    }
};

// ggml_type ids and block geometry used by Whisper files
enum { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8 };
bool type_geom(int t, int & blck, int & bytes) {
    switch (t) {
        case T_F32:  blck = 1;  bytes = 4;  return true;
        case T_F16:  blck = 1;  bytes = 2;  return true;
        case T_Q4_0: blck = 32; bytes = 18; return true;
        case T_Q4_1: blck = 32; bytes = 20; return true;
        case T_Q5_0: blck = 32; bytes = 22; return true;
        case T_Q5_1: blck = 32; bytes = 24; return true;
        case T_Q8_0: blck = 32; bytes = 34; return true;
        default: return false;
    }
}
// ftype -> weight type (W/ggml.c ggml_ftype_to_ggml_type)
int ftype_to_type(int ftype) {
    switch (ftype) {
        case 0: return T_F32; case 1: return T_F16; case 2: return T_Q4_0; case 3: return T_Q4_1;
        case 7: return T_Q8_0; case 8: return T_Q5_0; case 9: return T_Q5_1;
        default: return -1;
    }
}

float h2f(uint16_t h) { __half v; memcpy(&v, &h, 2); return __half2float(v); }
uint16_t f2h(float f) { __half v = __float2half_rn(f); uint16_t h; memcpy(&h, &v, 2); return h; }

// expand a tensor payload to f32 (vectors) — small tensors only
void to_f32(const FileTensor & t, const uint8_t * src, size_t ne, std::vector<float> & out) {
    out.resize(ne);
    if (t.ttype == T_F32) memcpy(out.data(), src, ne * 4);
    else if (t.ttype == T_F16) { const uint16_t * h = (const uint16_t *) src; for (size_t i = 0; i < ne; ++i) out[i] = h2f(h[i]); }
    else out.assign(ne, 0.0f);
}

// matrix payload -> f16 bit patterns.  (The quantised cases dequantise as W/ggml-quants.c dequantize_row_q* and are only
// reached for 3-D tensors — the quantize tool leaves those f16 — 2-D quantised matrices keep their blocks, see qmat_into.)
void to_f16(const FileTensor & t, const uint8_t * src, size_t ne, std::vector<uint16_t> & out) {
    out.resize(ne);
    switch (t.ttype) {
        case T_F16: memcpy(out.data(), src, ne * 2); break;
        case T_F32: { const float * f = (const float *) src; for (size_t i = 0; i < ne; ++i) out[i] = f2h(f[i]); } break;
        case T_Q8_0:
            for (size_t b = 0; b < ne / 32; ++b) {
                const uint8_t * blk = src + b * 34; uint16_t dh; memcpy(&dh, blk, 2); const float d = h2f(dh);
                const int8_t * qs = (const int8_t *) (blk + 2);
                for (int j = 0; j < 32; ++j) out[b * 32 + j] = f2h(qs[j] * d);
            } break;
        case T_Q4_0:
            for (size_t b = 0; b < ne / 32; ++b) {
                const uint8_t * blk = src + b * 18; uint16_t dh; memcpy(&dh, blk, 2); const float d = h2f(dh);
                const uint8_t * qs = blk + 2;
                for (int j = 0; j < 16; ++j) {
                    out[b * 32 + j]      = f2h(((int) (qs[j] & 0x0F) - 8) * d);
                    out[b * 32 + j + 16] = f2h(((int) (qs[j] >> 4) - 8) * d);
                }
            } break;
        case T_Q4_1:
            for (size_t b = 0; b < ne / 32; ++b) {
                const uint8_t * blk = src + b * 20; uint16_t dh, mh; memcpy(&dh, blk, 2); memcpy(&mh, blk + 2, 2);
                const float d = h2f(dh), m = h2f(mh); const uint8_t * qs = blk + 4;
                for (int j = 0; j < 16; ++j) {
                    out[b * 32 + j]      = f2h((qs[j] & 0x0F) * d + m);
                    out[b * 32 + j + 16] = f2h((qs[j] >> 4) * d + m);
                }
            } break;
        case T_Q5_0:
            for (size_t b = 0; b < ne / 32; ++b) {
                const uint8_t * blk = src + b * 22; uint16_t dh; memcpy(&dh, blk, 2); const float d = h2f(dh);
                uint32_t qh; memcpy(&qh, blk + 2, 4); const uint8_t * qs = blk + 6;
                for (int j = 0; j < 16; ++j) {
                    const uint8_t xh0 = ((qh >> (j + 0)) << 4) & 0x10, xh1 = ((qh >> (j + 12))) & 0x10;
                    out[b * 32 + j]      = f2h(((int) ((qs[j] & 0x0F) | xh0) - 16) * d);
                    out[b * 32 + j + 16] = f2h(((int) ((qs[j] >> 4) | xh1) - 16) * d);
                }
            } break;
        case T_Q5_1:
            for (size_t b = 0; b < ne / 32; ++b) {
                const uint8_t * blk = src + b * 24; uint16_t dh, mh; memcpy(&dh, blk, 2); memcpy(&mh, blk + 2, 2);
                const float d = h2f(dh), m = h2f(mh);
                uint32_t qh; memcpy(&qh, blk + 4, 4); const uint8_t * qs = blk + 8;
                for (int j = 0; j < 16; ++j) {
                    const uint8_t xh0 = ((qh >> (j + 0)) << 4) & 0x10, xh1 = ((qh >> (j + 12))) & 0x10;
                    out[b * 32 + j]      = f2h(((qs[j] & 0x0F) | xh0) * d + m);
                    out[b * 32 + j + 16] = f2h(((qs[j] >> 4) | xh1) * d + m);
                }
            } break;
        default: std::fill(out.begin(), out.end(), (uint16_t) 0);
    }
}

} // namespace

bool parse_model(const uint8_t * buf, size_t n, ModelFile & mf) {
    Reader rd{buf, n};
    const uint32_t magic = rd.get<uint32_t>();
    size_t arena_hint = 0;
    if (magic == 0x686d6977u) {                       // "wmih": a model image without tensor payloads (export_header)
        mf.directory_only = true;
        arena_hint = (size_t) rd.get<uint64_t>();
        (void) arena_hint;
        if (rd.get<uint32_t>() != 0x67676d6c) { WMI_ERR("%s: invalid header image\n", __func__); return false; }
    } else if (magic != 0x67676d6c) { WMI_ERR("%s: invalid model data (bad magic)\n", __func__); return false; }
    HParams & hp = mf.hp;
    hp.n_vocab = rd.get<int32_t>();       hp.n_audio_ctx = rd.get<int32_t>();   hp.n_audio_state = rd.get<int32_t>();
    hp.n_audio_head = rd.get<int32_t>();  hp.n_audio_layer = rd.get<int32_t>(); hp.n_text_ctx = rd.get<int32_t>();
    hp.n_text_state = rd.get<int32_t>();  hp.n_text_head = rd.get<int32_t>();   hp.n_text_layer = rd.get<int32_t>();
    hp.n_mels = rd.get<int32_t>();        hp.ftype = rd.get<int32_t>();
    if (!rd.ok) { WMI_ERR("%s: truncated header\n", __func__); return false; }
    switch (hp.n_audio_layer) { case 4: mf.model_type = 1; break; case 6: mf.model_type = 2; break; case 12: mf.model_type = 3; break;
                                case 24: mf.model_type = 4; break; case 32: mf.model_type = 5; break; default: mf.model_type = 0; }
    const int qntvr = hp.ftype >= 0 ? hp.ftype / 1000 : 0;
    if (hp.ftype >= 0) hp.ftype %= 1000;
    if (ftype_to_type(hp.ftype) < 0) { WMI_ERR("%s: invalid model (bad ftype value %d)\n", __func__, hp.ftype); return false; }
    mf.quantised = hp.ftype >= 2;
    // every size that later becomes a divisor, an allocation or a kernel bound
    auto in = [](int32_t v, int32_t lo, int32_t hi) { return v >= lo && v <= hi; };
    if (!in(hp.n_vocab, 1, 65536) || !in(hp.n_audio_ctx, 1, 1 << 16) || !in(hp.n_text_ctx, 8, 1 << 14) ||
        !in(hp.n_audio_state, 64, 1 << 14) || !in(hp.n_text_state, 64, 1 << 14) || !in(hp.n_audio_head, 1, 256) || !in(hp.n_text_head, 1, 256) ||
        !in(hp.n_audio_layer, 1, 256) || !in(hp.n_text_layer, 1, 256) || !in(hp.n_mels, 1, 256)) {
        WMI_ERR("%s: invalid model (hyper-parameter out of range)\n", __func__);
        return false;
    }
    if (hp.n_audio_state != hp.n_text_state || hp.n_audio_state % hp.n_audio_head != 0 || hp.n_text_state % hp.n_text_head != 0 ||
        hp.n_audio_state / hp.n_audio_head != 64 || hp.n_text_state / hp.n_text_head != 64) {
        WMI_ERR("%s: unsupported geometry (state %d/%d heads %d/%d; head size must be 64)\n", __func__,
                hp.n_audio_state, hp.n_text_state, hp.n_audio_head, hp.n_text_head);
        return false;
    }
    WMI_INFO("%s: n_vocab=%d n_audio_ctx=%d n_audio_state=%d n_audio_head=%d n_audio_layer=%d n_text_ctx=%d n_text_state=%d "
             "n_text_head=%d n_text_layer=%d n_mels=%d ftype=%d qntvr=%d\n", __func__, hp.n_vocab, hp.n_audio_ctx, hp.n_audio_state,
             hp.n_audio_head, hp.n_audio_layer, hp.n_text_ctx, hp.n_text_state, hp.n_text_head, hp.n_text_layer, hp.n_mels, hp.ftype, qntvr);

    mf.n_filt_mel = rd.get<int32_t>(); mf.n_filt_fft = rd.get<int32_t>();
    if (!rd.ok || mf.n_filt_mel != hp.n_mels || mf.n_filt_fft != 201) { WMI_ERR("%s: bad mel filter header (%d x %d for n_mels %d)\n", __func__, mf.n_filt_mel, mf.n_filt_fft, hp.n_mels); return false; }
    {
        const size_t nf = (size_t) mf.n_filt_mel * mf.n_filt_fft;
        const uint8_t * p = rd.bytes(nf * 4);
        if (!p) { WMI_ERR("%s: truncated mel filters\n", __func__); return false; }
        mf.filters.resize(nf); memcpy(mf.filters.data(), p, nf * 4);
    }

    Vocab & v = mf.vocab;
    const int32_t n_vocab_file = rd.get<int32_t>();
    if (!rd.ok || n_vocab_file < 0 || n_vocab_file > 1 << 20) { WMI_ERR("%s: bad vocab size\n", __func__); return false; }
    v.n_vocab = hp.n_vocab;
    v.id_to_token.assign(std::max(hp.n_vocab, n_vocab_file), std::string());
    for (int i = 0; i < n_vocab_file; ++i) {
        const uint32_t len = rd.get<uint32_t>();
        const uint8_t * p = len ? rd.bytes(len) : nullptr;
        if (!rd.ok) { WMI_ERR("%s: truncated vocab\n", __func__); return false; }
        std::string word = len ? std::string((const char *) p, len) : std::string();
        v.token_to_id[word] = i;
        v.id_to_token[i] = word;
    }
    if (v.is_multilingual()) {                        // W/whisper.cpp:1241-1256
        v.eot++; v.sot++;
        const int dt = v.num_languages() - 98;
        v.translate += dt; v.transcribe += dt; v.solm += dt; v.prev += dt; v.nosp += dt; v.not_ += dt; v.beg += dt;
    }
    if (v.beg >= hp.n_vocab || v.eot >= hp.n_vocab) { WMI_ERR("%s: invalid model (n_vocab %d too small for the special tokens)\n", __func__, hp.n_vocab); return false; }
    if (n_vocab_file < hp.n_vocab) {                  // synthesised names for the special ids (W/whisper.cpp:1258-1289)
        for (int i = n_vocab_file; i < hp.n_vocab; ++i) {
            std::string w;
            if (i > v.beg)              w = "[_TT_" + std::to_string(i - v.beg) + "]";
            else if (i == v.eot)        w = "[_EOT_]";
            else if (i == v.sot)        w = "[_SOT_]";
            else if (i == v.translate)  w = "[_TRANSLATE_]";
            else if (i == v.transcribe) w = "[_TRANSCRIBE_]";
            else if (i == v.solm)       w = "[_SOLM_]";
            else if (i == v.prev)       w = "[_PREV_]";
            else if (i == v.nosp)       w = "[_NOSP_]";
            else if (i == v.not_)       w = "[_NOT_]";
            else if (i == v.beg)        w = "[_BEG_]";
            else if (i > v.sot && i <= v.sot + v.num_languages()) {
                const char * ls = lang_str(i - v.sot - 1);
                w = "[_LANG_" + std::string(ls ? ls : "?") + "]";
            } else                      w = "[_extra_token_" + std::to_string(i) + "]";
            v.token_to_id[w] = i;
            v.id_to_token[i] = w;
        }
    }
    mf.header_bytes = rd.off - (mf.directory_only ? 12 : 0);

    // tensor directory
    mf.tensors.clear();
    while (true) {
        const int32_t n_dims = rd.get<int32_t>(), name_len = rd.get<int32_t>(), ttype = rd.get<int32_t>();
        if (!rd.ok) break;                            // clean end of file (or a truncated record head: the count check below catches it)
        if (n_dims < 1 || n_dims > 4 || name_len <= 0 || name_len > 256) { WMI_ERR("%s: corrupt tensor header\n", __func__); return false; }
        FileTensor t; t.n_dims = n_dims; t.ttype = ttype;
        size_t ne = 1;
        for (int i = 0; i < n_dims; ++i) {
            t.ne[i] = rd.get<int32_t>();
            if (!rd.ok || t.ne[i] <= 0 || t.ne[i] > (1 << 24) || ne > ((size_t) 1 << 40) / (size_t) t.ne[i]) { WMI_ERR("%s: corrupt tensor header (dimension)\n", __func__); return false; }
            ne *= (size_t) t.ne[i];
        }
        const uint8_t * nm = rd.bytes(name_len);
        if (!rd.ok) { WMI_ERR("%s: truncated tensor header\n", __func__); return false; }
        t.name.assign((const char *) nm, name_len);
        int blck, bpb;
        if (!type_geom(ttype, blck, bpb)) { WMI_ERR("%s: tensor '%s' has unsupported type %d\n", __func__, t.name.c_str(), ttype); return false; }
        if (t.ne[0] % blck != 0) { WMI_ERR("%s: tensor '%s': row length %lld is not a multiple of the block size %d\n", __func__, t.name.c_str(), (long long) t.ne[0], blck); return false; }
        t.nbytes = ne / blck * bpb;
        if (mf.directory_only) t.offset = 0;
        else {
            t.offset = rd.off;
            if (!rd.bytes(t.nbytes)) { WMI_ERR("%s: tensor '%s' is truncated\n", __func__, t.name.c_str()); return false; }
        }
        if (mf.tensors.count(t.name)) { WMI_ERR("%s: tensor '%s' appears twice\n", __func__, t.name.c_str()); return false; }
        mf.tensors[t.name] = t;
    }
    if (!rd.eof()) { WMI_ERR("%s: trailing bytes after the last tensor\n", __func__); return false; }
    mf.n_loaded = (int) mf.tensors.size();
    // geometry limits of the block-quantised kernels (k_quant.hip): output tiles of 128 columns (k_qgemm has no column guard) and
    // LayerNorm prologues that keep a row of <= 1536 columns in registers (k_qrows, k_q8_rows).  Every published model (384 ..
    // 1280) is inside; anything else is refused here instead of producing silently wrong numbers
    if (mf.quantised && mf.n_loaded > 0 && (hp.n_audio_state % 128 != 0 || hp.n_audio_state > 1536)) {
        WMI_ERR("%s: block-quantised models need n_state %% 128 == 0 and n_state <= 1536 (got %d)\n", __func__, hp.n_audio_state);
        return false;
    }
    {
        const int expected = 7 + 15 * hp.n_audio_layer + 4 + 24 * hp.n_text_layer;   // names at W/whisper.cpp:1354-1510
        if (mf.n_loaded == 0) {
            WMI_WARN("%s: WARN no tensors loaded from model file - assuming empty model for testing\n", __func__);
            // no tensor carries quantised blocks: the zero weights take the f16 path whatever the header's ftype says (the
            // quantised kernels dispatch on the matrices' block type, which an empty model does not have)
            mf.quantised = false;
        } else if (mf.n_loaded != expected) {
            WMI_ERR("%s: ERROR not all tensors loaded from model file - expected %d, got %d\n", __func__, expected, mf.n_loaded);
            return false;
        }
    }
    return true;
}

std::vector<uint8_t> export_header(const ModelFile & mf, const uint8_t * buf, size_t arena_bytes) {
    std::vector<uint8_t> out;
    auto put = [&](const void * p, size_t k) { const uint8_t * b = (const uint8_t *) p; out.insert(out.end(), b, b + k); };
    const uint32_t magic = 0x686d6977u; const uint64_t ab = arena_bytes;
    put(&magic, 4); put(&ab, 8);
    put(buf, mf.header_bytes);                            // the file's own magic, hparams, filters, vocabulary
    for (const auto & kv : mf.tensors) {
        const FileTensor & t = kv.second;
        const int32_t h[3] = { t.n_dims, (int32_t) t.name.size(), t.ttype };
        put(h, 12);
        for (int i = 0; i < t.n_dims; ++i) { const int32_t e = (int32_t) t.ne[i]; put(&e, 4); }
        put(t.name.data(), t.name.size());
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
namespace {

// Two passes over the same sequence of reservations: pass 1 (fill = false) only lays the arena out, pass 2 copies into a
// staging image of exactly that size.  The layout depends on the tensor directory alone, so every rank of a multi-GPU job
// derives the same offsets and the arena of rank 0 can be broadcast as is.
struct Arena {
    uint8_t * host = nullptr;             // staging image (pass 2)
    size_t size = 0;
    bool fill = false;
    size_t reserve(size_t bytes) { const size_t o = (size + 255) & ~(size_t) 255; size = o + bytes; return o; }
};

struct Builder {
    const ModelFile & mf; const uint8_t * buf; Arena & ar; bool ok = true;
    int qtype = 0; size_t matrix_bytes = 0;
    // a tensor of exactly this rank and these dimensions (ggml order: ne[0] fastest), or null + an error
    const FileTensor * find(const std::string & name, std::initializer_list<int64_t> ne) {
        auto it = mf.tensors.find(name);
        if (it == mf.tensors.end() && mf.n_loaded == 0) return nullptr;   // empty test model: zero weights (W/whisper.cpp:1627-1628)
        if (it == mf.tensors.end()) { WMI_ERR("upload_weights: tensor '%s' missing from model file\n", name.c_str()); ok = false; return nullptr; }
        const FileTensor & t = it->second;
        bool good = t.n_dims == (int) ne.size();
        int i = 0;
        for (int64_t e : ne) { good = good && t.ne[i] == e; ++i; }
        if (!good) { WMI_ERR("upload_weights: tensor '%s' has wrong shape in model file\n", name.c_str()); ok = false; return nullptr; }
        return &t;
    }
    bool quantised(const FileTensor * t) const { return t && t->ttype != T_F32 && t->ttype != T_F16; }
    const uint8_t * payload(const FileTensor * t) const { return (ar.fill && buf && t) ? buf + t->offset : nullptr; }

    size_t vec(const std::string & name, int64_t n, bool as_2d = false) {          // f32 vector
        const size_t off = ar.reserve((size_t) n * 4);
        const FileTensor * t = as_2d ? find(name, {1, n}) : find(name, {n});
        if (!t) return off;
        if (t->ttype != T_F32 && t->ttype != T_F16) { WMI_ERR("upload_weights: tensor '%s': vectors must be f32 or f16\n", name.c_str()); ok = false; return off; }
        if (const uint8_t * p = payload(t)) { std::vector<float> f; to_f32(*t, p, (size_t) n, f); memcpy(ar.host + off, f.data(), (size_t) n * 4); }
        return off;
    }
    size_t mat_f32(const std::string & name, int64_t k, int64_t n) {                // f32 matrix (pos. embeddings)
        const size_t off = ar.reserve((size_t) (k * n) * 4);
        const FileTensor * t = find(name, {k, n});
        if (!t) return off;
        if (t->ttype != T_F32 && t->ttype != T_F16) { WMI_ERR("upload_weights: tensor '%s': expected f32 or f16\n", name.c_str()); ok = false; return off; }
        if (const uint8_t * p = payload(t)) { std::vector<float> f; to_f32(*t, p, (size_t) (k * n), f); memcpy(ar.host + off, f.data(), (size_t) (k * n) * 4); }
        return off;
    }
    // A matrix [n][k] that may be stacked from `parts` tensors of [rows_each][k] (q | k | v rows): either f16 rows or quantised
    // tiles — one kind for the whole stack.  Returns the arena offset; `qt` = 0 (f16) or the ggml type of the tiles.
    size_t stack(const std::vector<std::string> & names, int64_t k, int64_t rows_each, int & qt) {
        const int64_t parts = (int64_t) names.size();
        std::vector<const FileTensor *> ts;
        qt = 0; bool any_q = false, any_f = false;
        for (const std::string & nm : names) {
            const FileTensor * t = find(nm, {k, rows_each});
            ts.push_back(t);
            if (t) { if (quantised(t)) { any_q = true; if (qt && qt != t->ttype) ok = false; qt = t->ttype; } else any_f = true; }
        }
        if ((any_q && any_f) || !ok) { if (ok) WMI_ERR("upload_weights: tensors stacked with '%s' mix quantised and plain types\n", names[0].c_str()); ok = false; qt = 0; }
        // the reference allocates every 2-D weight with the type the header's ftype names and rejects a file whose tensor has another
        // size (W/whisper.cpp:1295-1296, 1580-1600): the same rule here
        if (ok && (any_q || any_f) && mf.n_loaded > 0) {
            const int want = ftype_to_type(mf.hp.ftype);
            const bool good = mf.quantised ? (any_q && qt == want) : !any_q;
            if (!good) { WMI_ERR("upload_weights: tensor '%s' has wrong size in model file (type does not match ftype %d)\n", names[0].c_str(), mf.hp.ftype); ok = false; qt = 0; }
        }
        if (qt) {
            // stacked parts must start on a row-group boundary (the last row group of a lone matrix is padded with zero blocks)
            if (k % 64 != 0 || (parts > 1 && rows_each % 32 != 0)) { WMI_ERR("upload_weights: tensor '%s': quantised matrices need K %% 64 == 0 (and stacked rows %% 32 == 0)\n", names[0].c_str()); ok = false; qt = 0; }
            else if (qtype && qtype != qt) { WMI_ERR("upload_weights: more than one quantisation type in one model (%d and %d)\n", qtype, qt); ok = false; qt = 0; }
            else qtype = qt;
        }
        if (qt) {
            const size_t each = k::q_matrix_bytes(qt, rows_each, k);
            const size_t off = ar.reserve(each * parts);
            matrix_bytes += each * parts;
            for (int64_t p = 0; p < parts; ++p)
                if (const uint8_t * src = payload(ts[p])) k::q_repack_host(qt, src, rows_each, k, ar.host + off + each * p);
            return off;
        }
        const size_t each = (size_t) (k * rows_each) * 2;
        const size_t off = ar.reserve(each * parts);
        matrix_bytes += each * parts;
        for (int64_t p = 0; p < parts; ++p)
            if (const uint8_t * src = payload(ts[p])) { std::vector<uint16_t> h; to_f16(*ts[p], src, (size_t) (k * rows_each), h); memcpy(ar.host + off + each * p, h.data(), each); }
        return off;
    }
    size_t mat(const std::string & name, int64_t k, int64_t n, int & qt) { return stack({name}, k, n, qt); }
    // conv weight ggml [3][IC][OC] (tap fastest) -> [OC][tap][IC] f16, row padded with zeros to kpad
    size_t conv(const std::string & name, int64_t ic, int64_t oc, int kpad) {
        const size_t off = ar.reserve((size_t) oc * kpad * 2);
        const FileTensor * t = find(name, {3, ic, oc});
        if (!t) return off;
        matrix_bytes += (size_t) oc * kpad * 2;
        if (const uint8_t * p = payload(t)) {
            std::vector<uint16_t> h; to_f16(*t, p, (size_t) (3 * ic * oc), h);
            uint16_t * dst = (uint16_t *) (ar.host + off);
            for (int64_t o = 0; o < oc; ++o)
                for (int64_t c = 0; c < ic; ++c)
                    for (int tap = 0; tap < 3; ++tap)
                        dst[o * kpad + tap * ic + c] = h[(o * ic + c) * 3 + tap];
        }
        return off;
    }
};

struct EncOff { size_t ln1g, ln1b, ln2g, ln2b, wqkv, bqkv, wo, bo, w1, b1, w2, b2; int tqkv, to, t1, t2; };
struct DecOff { size_t ln1g, ln1b, ln2g, ln2b, ln3g, ln3b, wqkv, bqkv, wo, bo, wcq, bcq, wco, bco, w1, b1, w2, b2; int tqkv, to, tcq, tco, t1, t2; };
struct Plan {
    size_t c1w, c1b, c2w, c2b, epe, elng, elnb, wckv, bckv, dpe, dte, dlng, dlnb, filt, rng, taps;
    int tckv = 0, tte = 0;
    std::vector<EncOff> eo; std::vector<DecOff> dof;
};

// one pass over the architecture's tensors in a fixed order (W/whisper.cpp:1298-1513)
bool build(const ModelFile & mf, const uint8_t * buf, Arena & ar, Plan & pl, Weights & w) {
    const HParams & hp = mf.hp;
    const int64_t S = hp.n_audio_state, La = hp.n_audio_layer, Lt = hp.n_text_layer;
    Builder b{mf, buf, ar};
    pl.eo.assign(La, EncOff{}); pl.dof.assign(Lt, DecOff{});
    w.conv1_k = (int) ((3 * hp.n_mels + 31) / 32 * 32);
    w.conv2_k = (int) (3 * S);
    pl.c1w = b.conv("encoder.conv1.weight", hp.n_mels, S, w.conv1_k);
    pl.c1b = b.vec("encoder.conv1.bias", S, true);
    pl.c2w = b.conv("encoder.conv2.weight", S, S, w.conv2_k);
    pl.c2b = b.vec("encoder.conv2.bias", S, true);
    pl.epe = b.mat_f32("encoder.positional_embedding", S, hp.n_audio_ctx);
    auto qkv_bias = [&](const std::string & p, size_t & off) {
        off = ar.reserve((size_t) 3 * S * 4);
        const size_t q = b.vec(p + "attn.query.bias", S), v = b.vec(p + "attn.value.bias", S);
        if (ar.fill) { memcpy(ar.host + off, ar.host + q, S * 4); memset(ar.host + off + S * 4, 0, S * 4); memcpy(ar.host + off + 2 * S * 4, ar.host + v, S * 4); }
    };
    for (int64_t i = 0; i < La; ++i) {
        const std::string p = "encoder.blocks." + std::to_string(i) + ".";
        EncOff & e = pl.eo[i];
        e.ln1g = b.vec(p + "attn_ln.weight", S); e.ln1b = b.vec(p + "attn_ln.bias", S);
        e.wqkv = b.stack({p + "attn.query.weight", p + "attn.key.weight", p + "attn.value.weight"}, S, S, e.tqkv);
        qkv_bias(p, e.bqkv);
        e.wo = b.mat(p + "attn.out.weight", S, S, e.to); e.bo = b.vec(p + "attn.out.bias", S);
        e.ln2g = b.vec(p + "mlp_ln.weight", S);    e.ln2b = b.vec(p + "mlp_ln.bias", S);
        e.w1 = b.mat(p + "mlp.0.weight", S, 4 * S, e.t1); e.b1 = b.vec(p + "mlp.0.bias", 4 * S);
        e.w2 = b.mat(p + "mlp.2.weight", 4 * S, S, e.t2); e.b2 = b.vec(p + "mlp.2.bias", S);
    }
    pl.elng = b.vec("encoder.ln_post.weight", S); pl.elnb = b.vec("encoder.ln_post.bias", S);

    {   // cross-attention k | v of every decoder layer stacked: [L][2S][S]; bias [L][2S] (k part zero)
        std::vector<std::string> names;
        for (int64_t i = 0; i < Lt; ++i) {
            const std::string p = "decoder.blocks." + std::to_string(i) + ".";
            names.push_back(p + "cross_attn.key.weight"); names.push_back(p + "cross_attn.value.weight");
        }
        pl.wckv = b.stack(names, S, S, pl.tckv);
        pl.bckv = ar.reserve((size_t) Lt * 2 * S * 4);
        if (ar.fill) memset(ar.host + pl.bckv, 0, (size_t) Lt * 2 * S * 4);
    }
    pl.dpe = b.mat_f32("decoder.positional_embedding", S, hp.n_text_ctx);
    pl.dte = b.mat("decoder.token_embedding.weight", S, hp.n_vocab, pl.tte);
    for (int64_t i = 0; i < Lt; ++i) {
        const std::string p = "decoder.blocks." + std::to_string(i) + ".";
        DecOff & d = pl.dof[i];
        d.ln1g = b.vec(p + "attn_ln.weight", S); d.ln1b = b.vec(p + "attn_ln.bias", S);
        d.wqkv = b.stack({p + "attn.query.weight", p + "attn.key.weight", p + "attn.value.weight"}, S, S, d.tqkv);
        qkv_bias(p, d.bqkv);
        d.wo = b.mat(p + "attn.out.weight", S, S, d.to); d.bo = b.vec(p + "attn.out.bias", S);
        d.ln2g = b.vec(p + "cross_attn_ln.weight", S); d.ln2b = b.vec(p + "cross_attn_ln.bias", S);
        d.wcq = b.mat(p + "cross_attn.query.weight", S, S, d.tcq); d.bcq = b.vec(p + "cross_attn.query.bias", S);
        { const size_t v = b.vec(p + "cross_attn.value.bias", S);
          if (ar.fill) memcpy(ar.host + pl.bckv + (size_t) (i * 2 + 1) * S * 4, ar.host + v, S * 4); }
        d.wco = b.mat(p + "cross_attn.out.weight", S, S, d.tco); d.bco = b.vec(p + "cross_attn.out.bias", S);
        d.ln3g = b.vec(p + "mlp_ln.weight", S); d.ln3b = b.vec(p + "mlp_ln.bias", S);
        d.w1 = b.mat(p + "mlp.0.weight", S, 4 * S, d.t1); d.b1 = b.vec(p + "mlp.0.bias", 4 * S);
        d.w2 = b.mat(p + "mlp.2.weight", 4 * S, S, d.t2); d.b2 = b.vec(p + "mlp.2.bias", S);
    }
    pl.dlng = b.vec("decoder.ln.weight", S); pl.dlnb = b.vec("decoder.ln.bias", S);
    pl.filt = ar.reserve(mf.filters.size() * 4);
    if (ar.fill) memcpy(ar.host + pl.filt, mf.filters.data(), mf.filters.size() * 4);
    // per mel filter: the range of 4-tap groups that hold a non-zero weight.  The reference sums all 201 taps
    // (W/whisper.cpp:2759-2768); groups of zeros add exactly +0.0 to its double accumulator, so skipping them is
    // bit-identical and cuts the filterbank work ~8x (triangular filters are narrow).
    const int n_filt = mf.n_filt_mel, n_fft = mf.n_filt_fft;
    pl.rng = ar.reserve((size_t) std::max(n_filt, 1) * 2 * 4);
    // ... and those groups themselves, compact and 16-byte aligned: [n_mel][13][4] = the first 12 groups from g0 (zero-padded), then
    // {tap 200, 0, 0, 0}.  A filter row of the file has 201 floats (804 bytes): read in place, every lane's four taps were four
    // misaligned scalar loads from a different cache line (k_mel.hip).
    pl.taps = ar.reserve((size_t) std::max(n_filt, 1) * 13 * 4 * 4);
    if (ar.fill) {
        int32_t * rng = (int32_t *) (ar.host + pl.rng);
        const int n_groups = (n_fft - 1) / 4 + ((n_fft - 1) % 4 ? 1 : 0);     // 201 taps: groups 0..49 cover taps 0..199
        for (int j = 0; j < n_filt; ++j) {
            const float * f = mf.filters.data() + (size_t) j * n_fft;
            int g0 = n_groups, g1 = 0;
            for (int g = 0; g < n_groups; ++g) {
                bool nz = false;
                for (int t = 4 * g; t < std::min(4 * g + 4, n_fft); ++t) nz = nz || f[t] != 0.0f;
                if (nz) { g0 = std::min(g0, g); g1 = g + 1; }
            }
            if (g0 > g1) g0 = g1 = 0;
            rng[2 * j] = g0; rng[2 * j + 1] = g1;
        }
        float * taps = (float *) (ar.host + pl.taps);
        for (int j = 0; j < n_filt; ++j) {
            const float * f = mf.filters.data() + (size_t) j * n_fft;
            float * t = taps + (size_t) j * 13 * 4;
            for (int q = 0; q < 12; ++q)
                for (int e = 0; e < 4; ++e) {
                    const int g = rng[2 * j] + q, k = 4 * g + e;
                    t[q * 4 + e] = (g < rng[2 * j + 1] && k < n_fft) ? f[k] : 0.0f;
                }
            t[48] = n_fft > 200 ? f[200] : 0.0f; t[49] = t[50] = t[51] = 0.0f;
        }
    }
    ar.reserve(4096);                                  // tail slack: GEMM tiles may over-read clamped rows
    w.qtype = b.qtype; w.matrix_bytes = b.matrix_bytes;
    return b.ok;
}

} // namespace

bool plan_weights(const ModelFile & mf, Weights & w) {
    Plan pl; Arena dry;
    if (!build(mf, nullptr, dry, pl, w)) return false;
    w.arena_bytes = (dry.size + 255) & ~(size_t) 255;
    return true;
}

bool upload_weights(const ModelFile & mf, const uint8_t * host_buf, Weights & w, hipStream_t st) {
    const HParams & hp = mf.hp;
    const int64_t La = hp.n_audio_layer, Lt = hp.n_text_layer;
    Plan pl;
    Arena dry;
    if (!build(mf, nullptr, dry, pl, w)) return false;
    w.arena_bytes = (dry.size + 255) & ~(size_t) 255;
    if (!HIP_OK(hipMalloc(&w.arena, w.arena_bytes))) { w.arena = nullptr; return false; }
    if (host_buf && !mf.directory_only) {
        std::vector<uint8_t> stage;
        try { stage.assign(w.arena_bytes, 0); }
        catch (const std::bad_alloc &) { WMI_ERR("%s: out of host memory staging %.1f MB of weights\n", __func__, w.arena_bytes / 1e6); free_weights(w); return false; }
        Arena ar; ar.host = stage.data(); ar.fill = true;
        Plan pl2;
        if (!build(mf, host_buf, ar, pl2, w) || ar.size != dry.size) { free_weights(w); return false; }
        if (!HIP_OK(hipMemcpyAsync(w.arena, stage.data(), w.arena_bytes, hipMemcpyHostToDevice, st)) ||
            !HIP_OK(hipStreamSynchronize(st))) { free_weights(w); return false; }
    } else {
        // header image: the bytes arrive later (RCCL broadcast / peer copy); until then the arena holds zeros, never stale HBM
        if (!HIP_OK(hipMemsetAsync(w.arena, 0, w.arena_bytes, st)) || !HIP_OK(hipStreamSynchronize(st))) { free_weights(w); return false; }
    }

    uint8_t * base = (uint8_t *) w.arena;
    auto H = [&](size_t o, int qt) { return qt ? (const __half *) nullptr : (const __half *) (base + o); };
    auto Q = [&](size_t o, int qt) { k::QMat m; if (qt) { m.tiles = base + o; m.qtype = qt; } return m; };
    auto F = [&](size_t o) { return (const float *) (base + o); };
    w.conv1_w = (const __half *) (base + pl.c1w); w.conv1_b = F(pl.c1b); w.conv2_w = (const __half *) (base + pl.c2w); w.conv2_b = F(pl.c2b); w.e_pe = F(pl.epe);
    w.enc.resize(La);
    for (int64_t i = 0; i < La; ++i) {
        const EncOff & e = pl.eo[i]; EncLayerW & l = w.enc[i];
        l.ln1_g = F(e.ln1g); l.ln1_b = F(e.ln1b); l.ln2_g = F(e.ln2g); l.ln2_b = F(e.ln2b);
        l.w_qkv = H(e.wqkv, e.tqkv); l.q_qkv = Q(e.wqkv, e.tqkv); l.b_qkv = F(e.bqkv);
        l.w_o = H(e.wo, e.to); l.q_o = Q(e.wo, e.to); l.b_o = F(e.bo);
        l.w_fc1 = H(e.w1, e.t1); l.q_fc1 = Q(e.w1, e.t1); l.b_fc1 = F(e.b1);
        l.w_fc2 = H(e.w2, e.t2); l.q_fc2 = Q(e.w2, e.t2); l.b_fc2 = F(e.b2);
    }
    w.e_ln_g = F(pl.elng); w.e_ln_b = F(pl.elnb);
    w.w_ckv = H(pl.wckv, pl.tckv); w.q_ckv = Q(pl.wckv, pl.tckv); w.b_ckv = F(pl.bckv);
    w.d_pe = F(pl.dpe); w.d_te = H(pl.dte, pl.tte); w.q_te = Q(pl.dte, pl.tte);
    w.dec.resize(Lt);
    for (int64_t i = 0; i < Lt; ++i) {
        const DecOff & d = pl.dof[i]; DecLayerW & l = w.dec[i];
        l.ln1_g = F(d.ln1g); l.ln1_b = F(d.ln1b); l.ln2_g = F(d.ln2g); l.ln2_b = F(d.ln2b); l.ln3_g = F(d.ln3g); l.ln3_b = F(d.ln3b);
        l.w_qkv = H(d.wqkv, d.tqkv); l.q_qkv = Q(d.wqkv, d.tqkv); l.b_qkv = F(d.bqkv);
        l.w_o = H(d.wo, d.to); l.q_o = Q(d.wo, d.to); l.b_o = F(d.bo);
        l.w_cq = H(d.wcq, d.tcq); l.q_cq = Q(d.wcq, d.tcq); l.b_cq = F(d.bcq);
        l.w_co = H(d.wco, d.tco); l.q_co = Q(d.wco, d.tco); l.b_co = F(d.bco);
        l.w_fc1 = H(d.w1, d.t1); l.q_fc1 = Q(d.w1, d.t1); l.b_fc1 = F(d.b1);
        l.w_fc2 = H(d.w2, d.t2); l.q_fc2 = Q(d.w2, d.t2); l.b_fc2 = F(d.b2);
    }
    w.d_ln_g = F(pl.dlng); w.d_ln_b = F(pl.dlnb);
    w.mel_filters = F(pl.filt);
    w.mel_ranges = (const int32_t *) (base + pl.rng);
    w.mel_taps = F(pl.taps);
    WMI_INFO("%s: device weight arena = %.2f MB (matrices %.2f MB%s)\n", __func__, w.arena_bytes / 1e6, w.matrix_bytes / 1e6,
             w.qtype ? ", block-quantised, kept quantised" : "");
    return true;
}

void free_weights(Weights & w) {
    if (w.arena && !w.arena_borrowed) k::qweights_f16_release(w.arena, (const char *) w.arena + w.arena_bytes);      // the f16 images of its matrices
    if (w.arena && !w.arena_borrowed) (void) hipFree(w.arena);
    w.arena = nullptr; w.arena_bytes = 0; w.arena_borrowed = false;
}

} // namespace wmi
