// ggml Whisper model file -> host directory + device weight arena (SURVEY §8 row a13;
// format: SURVEY Appendix A, reference loader W/whisper.cpp:1102-1640).
//
// The caller's buffer is only borrowed during the call (the Godot host frees its PackedByteArray
// right after whisper_init_from_buffer_with_params, src/speech_to_text.cpp:338-345).  Payloads go
// host -> device through one staging pass; matrices are kept f16 and re-laid where a kernel wants a
// different operand order (conv taps, stacked q|k|v rows).  Quantised ggml types are expanded to f16
// at load (block formats: W/ggml-quants.h:10-47).

#include "wmi.h"

#include <cmath>
#include <cstring>

namespace wmi {

namespace {

struct Reader {
    const uint8_t * p; size_t n, off = 0; bool ok = true;
    bool eof() const { return off >= n; }
    template <typename T> T get() {
        T v{};
        if (off + sizeof(T) > n) { ok = false; off = n; return v; }
        memcpy(&v, p + off, sizeof(T)); off += sizeof(T); return v;
    }
    const uint8_t * bytes(size_t k) {
        if (off + k > n) { ok = false; off = n; return nullptr; }
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
void to_f32(const FileTensor & t, const uint8_t * src, std::vector<float> & out) {
    size_t ne = 1; for (int i = 0; i < 4; ++i) ne *= (size_t) t.ne[i];
    out.resize(ne);
    if (t.ttype == T_F32) memcpy(out.data(), src, ne * 4);
    else if (t.ttype == T_F16) { const uint16_t * h = (const uint16_t *) src; for (size_t i = 0; i < ne; ++i) out[i] = h2f(h[i]); }
    else out.assign(ne, 0.0f);
}

// expand a matrix payload to f16 bit patterns (W/ggml-quants.c dequantize_row_q*)
void to_f16(const FileTensor & t, const uint8_t * src, std::vector<uint16_t> & out) {
    size_t ne = 1; for (int i = 0; i < 4; ++i) ne *= (size_t) t.ne[i];
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
    if (rd.get<uint32_t>() != 0x67676d6c) { WMI_ERR("%s: invalid model data (bad magic)\n", __func__); return false; }
    HParams & hp = mf.hp;
    hp.n_vocab = rd.get<int32_t>();       hp.n_audio_ctx = rd.get<int32_t>();   hp.n_audio_state = rd.get<int32_t>();
    hp.n_audio_head = rd.get<int32_t>();  hp.n_audio_layer = rd.get<int32_t>(); hp.n_text_ctx = rd.get<int32_t>();
    hp.n_text_state = rd.get<int32_t>();  hp.n_text_head = rd.get<int32_t>();   hp.n_text_layer = rd.get<int32_t>();
    hp.n_mels = rd.get<int32_t>();        hp.ftype = rd.get<int32_t>();
    if (!rd.ok) { WMI_ERR("%s: truncated header\n", __func__); return false; }
    switch (hp.n_audio_layer) { case 4: mf.model_type = 1; break; case 6: mf.model_type = 2; break; case 12: mf.model_type = 3; break;
                                case 24: mf.model_type = 4; break; case 32: mf.model_type = 5; break; default: mf.model_type = 0; }
    const int qntvr = hp.ftype / 1000;
    hp.ftype %= 1000;
    if (ftype_to_type(hp.ftype) < 0) { WMI_ERR("%s: invalid model (bad ftype value %d)\n", __func__, hp.ftype); return false; }
    if (hp.n_audio_state != hp.n_text_state || hp.n_audio_state % hp.n_audio_head != 0 ||
        hp.n_audio_state / hp.n_audio_head != 64 || hp.n_text_state / hp.n_text_head != 64) {
        WMI_ERR("%s: unsupported geometry (state %d/%d heads %d/%d; head size must be 64)\n", __func__,
                hp.n_audio_state, hp.n_text_state, hp.n_audio_head, hp.n_text_head);
        return false;
    }
    WMI_INFO("%s: n_vocab=%d n_audio_ctx=%d n_audio_state=%d n_audio_head=%d n_audio_layer=%d n_text_ctx=%d n_text_state=%d "
             "n_text_head=%d n_text_layer=%d n_mels=%d ftype=%d qntvr=%d\n", __func__, hp.n_vocab, hp.n_audio_ctx, hp.n_audio_state,
             hp.n_audio_head, hp.n_audio_layer, hp.n_text_ctx, hp.n_text_state, hp.n_text_head, hp.n_text_layer, hp.n_mels, hp.ftype, qntvr);

    mf.n_filt_mel = rd.get<int32_t>(); mf.n_filt_fft = rd.get<int32_t>();
    if (!rd.ok || mf.n_filt_mel <= 0 || mf.n_filt_mel > 256 || mf.n_filt_fft != 201) { WMI_ERR("%s: bad mel filter header\n", __func__); return false; }
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

    // tensor directory
    mf.tensors.clear();
    while (true) {
        const int32_t n_dims = rd.get<int32_t>(), name_len = rd.get<int32_t>(), ttype = rd.get<int32_t>();
        if (rd.eof() && !rd.ok) break;                // clean end of file
        if (!rd.ok) break;
        if (n_dims < 1 || n_dims > 4 || name_len <= 0 || name_len > 256) { WMI_ERR("%s: corrupt tensor header\n", __func__); return false; }
        FileTensor t; t.n_dims = n_dims; t.ttype = ttype;
        size_t ne = 1;
        for (int i = 0; i < n_dims; ++i) { t.ne[i] = rd.get<int32_t>(); ne *= (size_t) t.ne[i]; }
        const uint8_t * nm = rd.bytes(name_len);
        if (!rd.ok) { WMI_ERR("%s: truncated tensor header\n", __func__); return false; }
        t.name.assign((const char *) nm, name_len);
        int blck, bpb;
        if (!type_geom(ttype, blck, bpb)) { WMI_ERR("%s: tensor '%s' has unsupported type %d\n", __func__, t.name.c_str(), ttype); return false; }
        t.nbytes = ne / blck * bpb; t.offset = rd.off;
        if (!rd.bytes(t.nbytes)) { WMI_ERR("%s: tensor '%s' is truncated\n", __func__, t.name.c_str()); return false; }
        mf.tensors[t.name] = t;
    }
    mf.n_loaded = (int) mf.tensors.size();
    {
        const int expected = 7 + 15 * hp.n_audio_layer + 4 + 24 * hp.n_text_layer;   // names at W/whisper.cpp:1354-1510
        if (mf.n_loaded == 0) {
            WMI_WARN("%s: WARN no tensors loaded from model file - assuming empty model for testing\n", __func__);
        } else if (mf.n_loaded != expected) {
            WMI_ERR("%s: ERROR not all tensors loaded from model file - expected %d, got %d\n", __func__, expected, mf.n_loaded);
            return false;
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
namespace {

struct Arena {
    std::vector<uint8_t> host;            // staging image of the whole arena
    size_t reserve(size_t bytes) { const size_t o = (host.size() + 255) & ~(size_t) 255; host.resize(o + bytes, 0); return o; }
};

struct Builder {
    const ModelFile & mf; const uint8_t * buf; Arena & ar; bool ok = true;
    const FileTensor * find(const std::string & name, std::initializer_list<int64_t> ne) {
        auto it = mf.tensors.find(name);
        if (it == mf.tensors.end() && mf.n_loaded == 0) return nullptr;   // empty test model: zero weights (W/whisper.cpp:1627-1628)
        if (it == mf.tensors.end()) { WMI_ERR("upload_weights: tensor '%s' missing from model file\n", name.c_str()); ok = false; return nullptr; }
        int i = 0;
        for (int64_t e : ne) { if (it->second.ne[i] != e) { WMI_ERR("upload_weights: tensor '%s' has wrong shape in model file\n", name.c_str()); ok = false; return nullptr; } ++i; }
        return &it->second;
    }
    size_t vec(const std::string & name, int64_t n, bool as_2d = false) {          // f32 vector
        const size_t off = ar.reserve((size_t) n * 4);
        const FileTensor * t = as_2d ? find(name, {1, n}) : find(name, {n});
        if (!t) return off;
        std::vector<float> f; to_f32(*t, buf + t->offset, f);
        memcpy(ar.host.data() + off, f.data(), (size_t) n * 4);
        return off;
    }
    size_t mat_f32(const std::string & name, int64_t k, int64_t n) {                // f32 matrix (pos. embeddings)
        const size_t off = ar.reserve((size_t) (k * n) * 4);
        const FileTensor * t = find(name, {k, n});
        if (!t) return off;
        std::vector<float> f; to_f32(*t, buf + t->offset, f);
        memcpy(ar.host.data() + off, f.data(), f.size() * 4);
        return off;
    }
    void mat_into(const std::string & name, int64_t k, int64_t n, size_t off) {     // f16 [n][k] at a fixed arena offset
        const FileTensor * t = find(name, {k, n});
        if (!t) return;
        std::vector<uint16_t> h; to_f16(*t, buf + t->offset, h);
        memcpy(ar.host.data() + off, h.data(), h.size() * 2);
    }
    size_t mat(const std::string & name, int64_t k, int64_t n) {
        const size_t off = ar.reserve((size_t) (k * n) * 2);
        mat_into(name, k, n, off);
        return off;
    }
    // conv weight ggml [3][IC][OC] (tap fastest) -> [OC][tap][IC] f16, row padded with zeros to kpad
    size_t conv(const std::string & name, int64_t ic, int64_t oc, int kpad) {
        const size_t off = ar.reserve((size_t) oc * kpad * 2);
        const FileTensor * t = find(name, {3, ic, oc});
        if (!t) return off;
        std::vector<uint16_t> h; to_f16(*t, buf + t->offset, h);
        uint16_t * dst = (uint16_t *) (ar.host.data() + off);
        for (int64_t o = 0; o < oc; ++o)
            for (int64_t c = 0; c < ic; ++c)
                for (int tap = 0; tap < 3; ++tap)
                    dst[o * kpad + tap * ic + c] = h[(o * ic + c) * 3 + tap];
        return off;
    }
};

} // namespace

bool upload_weights(const ModelFile & mf, const uint8_t * host_buf, const void * /*dev_image*/, Weights & w, hipStream_t st) {
    const HParams & hp = mf.hp;
    const int64_t S = hp.n_audio_state, La = hp.n_audio_layer, Lt = hp.n_text_layer;
    Arena ar;
    Builder b{mf, host_buf, ar};

    struct EncOff { size_t ln1g, ln1b, ln2g, ln2b, wqkv, bqkv, wo, bo, w1, b1, w2, b2; };
    struct DecOff { size_t ln1g, ln1b, ln2g, ln2b, ln3g, ln3b, wqkv, bqkv, wo, bo, wcq, bcq, wco, bco, w1, b1, w2, b2; };
    std::vector<EncOff> eo(La); std::vector<DecOff> dof(Lt);

    w.conv1_k = (int) ((3 * hp.n_mels + 31) / 32 * 32);
    w.conv2_k = (int) (3 * S);
    const size_t o_c1w = b.conv("encoder.conv1.weight", hp.n_mels, S, w.conv1_k);
    const size_t o_c1b = b.vec("encoder.conv1.bias", S, true);
    const size_t o_c2w = b.conv("encoder.conv2.weight", S, S, w.conv2_k);
    const size_t o_c2b = b.vec("encoder.conv2.bias", S, true);
    const size_t o_epe = b.mat_f32("encoder.positional_embedding", S, hp.n_audio_ctx);
    for (int64_t i = 0; i < La; ++i) {
        const std::string p = "encoder.blocks." + std::to_string(i) + ".";
        EncOff & e = eo[i];
        e.ln1g = b.vec(p + "attn_ln.weight", S); e.ln1b = b.vec(p + "attn_ln.bias", S);
        e.wqkv = ar.reserve((size_t) 3 * S * S * 2);
        b.mat_into(p + "attn.query.weight", S, S, e.wqkv);
        b.mat_into(p + "attn.key.weight",   S, S, e.wqkv + (size_t) S * S * 2);
        b.mat_into(p + "attn.value.weight", S, S, e.wqkv + (size_t) 2 * S * S * 2);
        e.bqkv = ar.reserve((size_t) 3 * S * 4);
        { const size_t q = b.vec(p + "attn.query.bias", S), v = b.vec(p + "attn.value.bias", S);
          memcpy(ar.host.data() + e.bqkv, ar.host.data() + q, S * 4); memcpy(ar.host.data() + e.bqkv + 2 * S * 4, ar.host.data() + v, S * 4); }
        e.wo = b.mat(p + "attn.out.weight", S, S); e.bo = b.vec(p + "attn.out.bias", S);
        e.ln2g = b.vec(p + "mlp_ln.weight", S);    e.ln2b = b.vec(p + "mlp_ln.bias", S);
        e.w1 = b.mat(p + "mlp.0.weight", S, 4 * S); e.b1 = b.vec(p + "mlp.0.bias", 4 * S);
        e.w2 = b.mat(p + "mlp.2.weight", 4 * S, S); e.b2 = b.vec(p + "mlp.2.bias", S);
    }
    const size_t o_elng = b.vec("encoder.ln_post.weight", S), o_elnb = b.vec("encoder.ln_post.bias", S);

    const size_t o_wckv = ar.reserve((size_t) Lt * 2 * S * S * 2);
    const size_t o_bckv = ar.reserve((size_t) Lt * 2 * S * 4);
    const size_t o_dpe = b.mat_f32("decoder.positional_embedding", S, hp.n_text_ctx);
    const size_t o_dte = b.mat("decoder.token_embedding.weight", S, hp.n_vocab);
    for (int64_t i = 0; i < Lt; ++i) {
        const std::string p = "decoder.blocks." + std::to_string(i) + ".";
        DecOff & d = dof[i];
        d.ln1g = b.vec(p + "attn_ln.weight", S); d.ln1b = b.vec(p + "attn_ln.bias", S);
        d.wqkv = ar.reserve((size_t) 3 * S * S * 2);
        b.mat_into(p + "attn.query.weight", S, S, d.wqkv);
        b.mat_into(p + "attn.key.weight",   S, S, d.wqkv + (size_t) S * S * 2);
        b.mat_into(p + "attn.value.weight", S, S, d.wqkv + (size_t) 2 * S * S * 2);
        d.bqkv = ar.reserve((size_t) 3 * S * 4);
        { const size_t q = b.vec(p + "attn.query.bias", S), v = b.vec(p + "attn.value.bias", S);
          memcpy(ar.host.data() + d.bqkv, ar.host.data() + q, S * 4); memcpy(ar.host.data() + d.bqkv + 2 * S * 4, ar.host.data() + v, S * 4); }
        d.wo = b.mat(p + "attn.out.weight", S, S); d.bo = b.vec(p + "attn.out.bias", S);
        d.ln2g = b.vec(p + "cross_attn_ln.weight", S); d.ln2b = b.vec(p + "cross_attn_ln.bias", S);
        d.wcq = b.mat(p + "cross_attn.query.weight", S, S); d.bcq = b.vec(p + "cross_attn.query.bias", S);
        b.mat_into(p + "cross_attn.key.weight",   S, S, o_wckv + (size_t) (i * 2) * S * S * 2);
        b.mat_into(p + "cross_attn.value.weight", S, S, o_wckv + (size_t) (i * 2 + 1) * S * S * 2);
        { const size_t v = b.vec(p + "cross_attn.value.bias", S);
          memcpy(ar.host.data() + o_bckv + (size_t) (i * 2 + 1) * S * 4, ar.host.data() + v, S * 4); }
        d.wco = b.mat(p + "cross_attn.out.weight", S, S); d.bco = b.vec(p + "cross_attn.out.bias", S);
        d.ln3g = b.vec(p + "mlp_ln.weight", S); d.ln3b = b.vec(p + "mlp_ln.bias", S);
        d.w1 = b.mat(p + "mlp.0.weight", S, 4 * S); d.b1 = b.vec(p + "mlp.0.bias", 4 * S);
        d.w2 = b.mat(p + "mlp.2.weight", 4 * S, S); d.b2 = b.vec(p + "mlp.2.bias", S);
    }
    const size_t o_dlng = b.vec("decoder.ln.weight", S), o_dlnb = b.vec("decoder.ln.bias", S);
    const size_t o_filt = ar.reserve(mf.filters.size() * 4);
    memcpy(ar.host.data() + o_filt, mf.filters.data(), mf.filters.size() * 4);
    // per mel filter: the range of 4-tap groups that hold a non-zero weight.  The reference sums all 201 taps
    // (W/whisper.cpp:2759-2768); groups of zeros add exactly +0.0 to its double accumulator, so skipping them is
    // bit-identical and cuts the filterbank work ~8x (triangular filters are narrow).
    const int n_filt = mf.n_filt_mel, n_fft = mf.n_filt_fft;
    const size_t o_rng = ar.reserve((size_t) std::max(n_filt, 1) * 2 * 4);
    {
        int32_t * rng = (int32_t *) (ar.host.data() + o_rng);
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
    }
    // ... and those groups themselves, compact and 16-byte aligned: [n_mel][13][4] = the first 12 groups from g0 (zero-padded), then
    // {tap 200, 0, 0, 0}.  A filter row of the file has 201 floats (804 bytes): read in place, every lane's four taps were four
    // misaligned scalar loads from a different cache line (k_mel.hip).
    const size_t o_taps = ar.reserve((size_t) std::max(n_filt, 1) * 13 * 4 * 4);
    {
        const int32_t * rng = (const int32_t *) (ar.host.data() + o_rng);
        float * taps = (float *) (ar.host.data() + o_taps);
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

    if (!b.ok) return false;

    w.arena_bytes = ar.host.size();
    if (!HIP_OK(hipMalloc(&w.arena, w.arena_bytes))) { w.arena = nullptr; return false; }
    if (!HIP_OK(hipMemcpyAsync(w.arena, ar.host.data(), w.arena_bytes, hipMemcpyHostToDevice, st)) ||
        !HIP_OK(hipStreamSynchronize(st))) { free_weights(w); return false; }

    uint8_t * base = (uint8_t *) w.arena;
    auto H = [&](size_t o) { return (const __half *) (base + o); };
    auto F = [&](size_t o) { return (const float *) (base + o); };
    w.conv1_w = H(o_c1w); w.conv1_b = F(o_c1b); w.conv2_w = H(o_c2w); w.conv2_b = F(o_c2b); w.e_pe = F(o_epe);
    w.enc.resize(La);
    for (int64_t i = 0; i < La; ++i) {
        const EncOff & e = eo[i]; EncLayerW & l = w.enc[i];
        l.ln1_g = F(e.ln1g); l.ln1_b = F(e.ln1b); l.ln2_g = F(e.ln2g); l.ln2_b = F(e.ln2b);
        l.w_qkv = H(e.wqkv); l.b_qkv = F(e.bqkv); l.w_o = H(e.wo); l.b_o = F(e.bo);
        l.w_fc1 = H(e.w1); l.b_fc1 = F(e.b1); l.w_fc2 = H(e.w2); l.b_fc2 = F(e.b2);
    }
    w.e_ln_g = F(o_elng); w.e_ln_b = F(o_elnb);
    w.w_ckv = H(o_wckv); w.b_ckv = F(o_bckv); w.d_pe = F(o_dpe); w.d_te = H(o_dte);
    w.dec.resize(Lt);
    for (int64_t i = 0; i < Lt; ++i) {
        const DecOff & d = dof[i]; DecLayerW & l = w.dec[i];
        l.ln1_g = F(d.ln1g); l.ln1_b = F(d.ln1b); l.ln2_g = F(d.ln2g); l.ln2_b = F(d.ln2b); l.ln3_g = F(d.ln3g); l.ln3_b = F(d.ln3b);
        l.w_qkv = H(d.wqkv); l.b_qkv = F(d.bqkv); l.w_o = H(d.wo); l.b_o = F(d.bo);
        l.w_cq = H(d.wcq); l.b_cq = F(d.bcq); l.w_co = H(d.wco); l.b_co = F(d.bco);
        l.w_fc1 = H(d.w1); l.b_fc1 = F(d.b1); l.w_fc2 = H(d.w2); l.b_fc2 = F(d.b2);
    }
    w.d_ln_g = F(o_dlng); w.d_ln_b = F(o_dlnb);
    w.mel_filters = F(o_filt);
    w.mel_ranges = (const int32_t *) (base + o_rng);
    w.mel_taps = F(o_taps);
    WMI_INFO("%s: device weight arena = %.2f MB\n", __func__, w.arena_bytes / 1e6);
    return true;
}

void free_weights(Weights & w) {
    if (w.arena) (void) hipFree(w.arena);
    w.arena = nullptr; w.arena_bytes = 0;
}

} // namespace wmi
