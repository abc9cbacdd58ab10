// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  Nothing under godot-whisper_amd/ may call this.
//
// CPU restatement ("port") of the reference's hot path  PCM -> log-mel -> conv -> encoder -> cross K/V
// -> KV-cached decoder step -> logits, written from the algorithm description in SURVEY.md (App. B, G)
// with the reference's rounding points AND its summation order, so that it can stand in for the
// compiled reference on machines where /root/reference does not exist (the GPU box):
//
//   * every mul_mat operand is IEEE f16, products accumulate in f32 through the 4 x 8-lane FMA pattern
//     of ggml's AVX vec_dot_f16 (W/ggml.c:1182-1216, reduce :750-771), leftovers in double;
//   * GELU and exp go through 65536-entry f16 tables (W/ggml.c:1400-1423, 2222-2235, 11176-11186);
//   * LayerNorm uses double accumulators (W/ggml.c:9329-9348);
//   * K/V are stored f16 (W/whisper.cpp:1887-1909, 2057-2066, 2280-2288);
//   * log-mel follows W/whisper.cpp:2614-2887 step by step (table sin/cos, radix-2 down to 25-point DFT);
//   * block-quantised weights (q4_0 q4_1 q5_0 q5_1 q8_0) stay quantised: activations go to q8 blocks and the dot is the
//     reference's integer dot with per-block scales (oracle/port_quants.c; W/ggml-quants.c:837-870, 2442-3560).
//
// Pinning: tests/test_oracle_port.py checks this file against the compiled reference
// (oracle/_ref/libwhisper_ref.so) in the build container and against the committed golden vectors
// (tests/golden/*.npz, produced from the reference by tests/golden/make_goldens.py) everywhere.
//
// Build: make -C oracle port   (g++ -O3 -mavx2 -mfma -mf16c; needs an x86-64 host with F16C/FMA)

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef uint16_t f16;

inline float h2f(f16 h) { return _cvtsh_ss(h); }
inline f16   f2h(float f) { return _cvtss_sh(f, 0); }

f16 g_gelu[65536], g_exp[65536];
extern "C" void port_fill_tables(uint16_t * gelu, uint16_t * expt);   // oracle/port_tables.c (compiled as C, like ggml.c)
// oracle/port_quants.c (compiled as C with ggml's flags): the reference's block-quantised arithmetic
extern "C" {
int   port_q_block_bytes(int type);
int   port_q_act_has_sum(int type);
void  port_quantize_row_q8(const float * x, int k, int with_sum, int8_t * qs, float * d, float * s);
float port_vec_dot_q(int type, int k, const uint8_t * wrow, const int8_t * qs, const float * d, const float * s);
void  port_dequantize_row(int type, const uint8_t * wrow, float * y, int k);
}
void init_tables() {
    static bool done = false;
    if (done) return;
    port_fill_tables(g_gelu, g_exp);
    done = true;
}
inline float gelu_tab(float x) { return h2f(g_gelu[f2h(x)]); }

void parallel_for(int n, int n_threads, const std::function<void(int, int)> & fn) {
    n_threads = std::max(1, std::min(n_threads, n));
    if (n_threads == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int chunk = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        const int a = t * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        th.emplace_back(fn, a, b);
    }
    for (auto & t : th) t.join();
}

// dot product of two f16 vectors with ggml's AVX accumulation pattern
float dot_f16(int n, const f16 * x, const f16 * y) {
    const int np = n & ~31;
    __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
    for (int i = 0; i < np; i += 32) {
        s0 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (x + i))),      _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (y + i))),      s0);
        s1 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (x + i + 8))),  _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (y + i + 8))),  s1);
        s2 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (x + i + 16))), _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (y + i + 16))), s2);
        s3 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (x + i + 24))), _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (y + i + 24))), s3);
    }
    s0 = _mm256_add_ps(s0, s2); s1 = _mm256_add_ps(s1, s3); s0 = _mm256_add_ps(s0, s1);
    const __m128 t0 = _mm_add_ps(_mm256_castps256_ps128(s0), _mm256_extractf128_ps(s0, 1));
    const __m128 t1 = _mm_hadd_ps(t0, t0);
    double sumf = (double) _mm_cvtss_f32(_mm_hadd_ps(t1, t1));
    for (int i = np; i < n; ++i) sumf += (double) (h2f(x[i]) * h2f(y[i]));
    return (float) sumf;
}

void row_to_f16(const float * src, f16 * dst, int n) {
    int i = 0;
    for (; i + 7 < n; i += 8) _mm_storeu_si128((__m128i *) (dst + i), _mm256_cvtps_ph(_mm256_loadu_ps(src + i), 0));
    for (; i < n; ++i) dst[i] = f2h(src[i]);
}

struct Tensor { std::vector<f16> h; std::vector<float> f; std::vector<uint8_t> q; int qtype = 0; int64_t ne[4] = {1, 1, 1, 1}; bool is_f16 = false; };

struct Model {
    int n_vocab, n_audio_ctx, S, H, La, n_text_ctx, St, Ht, Lt, n_mels, ftype;
    int filt_mel = 0; std::vector<float> filters;
    std::map<std::string, Tensor> t;
    const Tensor & get(const std::string & n) const { return t.at(n); }
};

bool load(const uint8_t * p, size_t n, Model & m) {
    size_t off = 0;
    auto rd32 = [&]() { int32_t v = 0; if (off + 4 <= n) memcpy(&v, p + off, 4); off += 4; return v; };
    if ((uint32_t) rd32() != 0x67676d6c) return false;
    m.n_vocab = rd32(); m.n_audio_ctx = rd32(); m.S = rd32(); m.H = rd32(); m.La = rd32(); m.n_text_ctx = rd32();
    m.St = rd32(); m.Ht = rd32(); m.Lt = rd32(); m.n_mels = rd32(); m.ftype = rd32() % 1000;
    m.filt_mel = rd32(); const int nfft = rd32();
    m.filters.resize((size_t) m.filt_mel * nfft); memcpy(m.filters.data(), p + off, m.filters.size() * 4); off += m.filters.size() * 4;
    const int nv = rd32();
    for (int i = 0; i < nv; ++i) { const uint32_t len = (uint32_t) rd32(); off += len; }
    while (off + 12 <= n) {
        const int nd = rd32(), nl = rd32(), tt = rd32();
        Tensor t; size_t ne = 1;
        for (int i = 0; i < nd; ++i) { t.ne[i] = rd32(); ne *= (size_t) t.ne[i]; }
        const std::string name((const char *) p + off, nl); off += nl;
        if (tt == 1) { t.is_f16 = true; t.h.resize(ne); memcpy(t.h.data(), p + off, ne * 2); off += ne * 2; }
        else if (tt == 0) { t.f.resize(ne); memcpy(t.f.data(), p + off, ne * 4); off += ne * 4; }
        else if (port_q_block_bytes(tt) > 0 && ne % 32 == 0) {   // block-quantised 2-D tensor: kept as the file's blocks
            const size_t nbytes = ne / 32 * (size_t) port_q_block_bytes(tt);
            if (off + nbytes > n) return false;
            t.qtype = tt; t.q.assign(p + off, p + off + nbytes); off += nbytes;
        }
        else return false;
        m.t[name] = std::move(t);
    }
    return true;
}

struct Ctx {
    Model m; int n_threads = 4;
    // mel
    std::vector<float> mel; int n_len = 0, n_len_org = 0;
    // encoder results (reference layouts transposed to token-major where noted)
    int T = 0;
    std::vector<float> embd_conv;          // [T][S]
    std::vector<float> embd_enc;           // [T][S]
    std::vector<f16> cross_k, cross_v;     // [L][T][S] each (v un-transposed)
    // self KV (single sequence): [L][n_text_ctx][S]
    std::vector<f16> self_k, self_v; int n_past = 0;
    std::vector<float> logits;             // last row
};

// ------------------------------------------------------------------------------------------------ mel
float g_sin[400], g_cos[400];
void fill_trig() {
    static bool done = false; if (done) return;
    for (int i = 0; i < 400; ++i) { const double th = (2 * M_PI * i) / 400; g_sin[i] = sinf(th); g_cos[i] = cosf(th); }
    done = true;
}
void dft25(const float * in, int N, float * out) {
    const int step = 400 / N;
    for (int k = 0; k < N; ++k) {
        float re = 0, im = 0;
        for (int n = 0; n < N; ++n) { const int idx = (k * n * step) % 400; re += in[n] * g_cos[idx]; im -= in[n] * g_sin[idx]; }
        out[2 * k] = re; out[2 * k + 1] = im;
    }
}
void fft_rec(const std::vector<float> & in, std::vector<float> & out) {
    const int N = (int) in.size();
    out.resize(2 * N);
    if (N == 1) { out[0] = in[0]; out[1] = 0; return; }
    if (N % 2 == 1) { dft25(in.data(), N, out.data()); return; }
    std::vector<float> ev, od, fe, fo;
    for (int i = 0; i < N; ++i) (i % 2 == 0 ? ev : od).push_back(in[i]);
    fft_rec(ev, fe); fft_rec(od, fo);
    const int step = 400 / N;
    for (int k = 0; k < N / 2; ++k) {
        const int idx = k * step;
        const float re = g_cos[idx], im = -g_sin[idx];
        const float ro = fo[2 * k], io = fo[2 * k + 1];
        out[2 * k]               = fe[2 * k] + re * ro - im * io;
        out[2 * k + 1]           = fe[2 * k + 1] + re * io + im * ro;
        out[2 * (k + N / 2)]     = fe[2 * k] - re * ro + im * io;
        out[2 * (k + N / 2) + 1] = fe[2 * k + 1] - re * io - im * ro;
    }
}

void pcm_to_mel(Ctx & c, const float * samples, int n) {
    fill_trig();
    const int n_mel = c.m.filt_mel;
    std::vector<float> hann(400);
    for (int i = 0; i < 400; ++i) hann[i] = 0.5 * (1.0 - cosf((2.0 * M_PI * i) / 400));
    std::vector<float> pad((size_t) n + 480000 + 400, 0.0f);
    std::copy(samples, samples + n, pad.begin() + 200);
    for (int i = 0; i < 200; ++i) pad[i] = samples[std::min(200 - i, n - 1)];
    c.n_len = (int) ((pad.size() - 400) / 160);
    c.n_len_org = 1 + (n + 200 - 400) / 160;
    c.mel.assign((size_t) n_mel * c.n_len, 0.0f);
    const int n_valid = n + 200;
    const int n_fft_frames = std::min(n_valid / 160 + 1, c.n_len);
    parallel_for(c.n_len, c.n_threads, [&](int a, int b) {
        std::vector<float> x(400), X;
        for (int i = a; i < b; ++i) {
            if (i >= n_fft_frames) { for (int j = 0; j < n_mel; ++j) c.mel[(size_t) j * c.n_len + i] = (float) log10(1e-10); continue; }
            const int off = i * 160;
            const int m = std::min(400, n_valid - off);
            for (int j = 0; j < 400; ++j) x[j] = j < m ? hann[j] * pad[off + j] : 0.0f;
            fft_rec(x, X);
            for (int j = 0; j < 400; ++j) X[j] = X[2 * j] * X[2 * j] + X[2 * j + 1] * X[2 * j + 1];
            for (int j = 0; j < n_mel; ++j) {
                const float * f = c.m.filters.data() + (size_t) j * 201;
                double sum = 0.0; int k = 0;
                for (; k < 201 - 3; k += 4) sum += X[k] * f[k] + X[k + 1] * f[k + 1] + X[k + 2] * f[k + 2] + X[k + 3] * f[k + 3];
                for (; k < 201; ++k) sum += X[k] * f[k];
                c.mel[(size_t) j * c.n_len + i] = (float) log10(std::max(sum, 1e-10));
            }
        }
    });
    double mmax = -1e20;
    for (float v : c.mel) if (v > mmax) mmax = v;
    mmax -= 8.0;
    for (float & v : c.mel) { if (v < mmax) v = (float) mmax; v = (float) ((v + 4.0) / 4.0); }
}

// ------------------------------------------------------------------------------------------------ building blocks
// out[j][i] = dot(W[i][:], act[j][:]) for i < N, j < M ; W f16 [N][K], act f16 [M][K] ; out layout [M][N]
void matmul(const Ctx & c, const f16 * W, int N, int K, const f16 * act, int M, float * out) {
    parallel_for(M, c.n_threads, [&](int a, int b) {
        for (int j = a; j < b; ++j) for (int i = 0; i < N; ++i) out[(size_t) j * N + i] = dot_f16(K, W + (size_t) i * K, act + (size_t) j * K);
    });
}
void matmul_f32act(const Ctx & c, const f16 * W, int N, int K, const float * act, int M, float * out) {
    std::vector<f16> a16((size_t) M * K);
    for (int j = 0; j < M; ++j) row_to_f16(act + (size_t) j * K, a16.data() + (size_t) j * K, K);
    matmul(c, W, N, K, a16.data(), M, out);
}
struct Ctx;
void matmul_w(const Ctx & c, const std::string & name, int N, int K, const float * act, int M, float * out);
void add_bias(float * x, int M, int N, const float * b) { for (int j = 0; j < M; ++j) for (int i = 0; i < N; ++i) x[(size_t) j * N + i] += b[i]; }

void layernorm(const float * x, int M, int S, const float * g, const float * b, float * y) {
    for (int j = 0; j < M; ++j) {
        const float * xr = x + (size_t) j * S; float * yr = y + (size_t) j * S;
        double sum = 0.0; for (int i = 0; i < S; ++i) sum += (double) xr[i];
        const float mean = sum / S;
        double sum2 = 0.0;
        for (int i = 0; i < S; ++i) { const float v = xr[i] - mean; yr[i] = v; sum2 += (double) (v * v); }
        const float var = sum2 / S;
        const float sc = 1.0f / sqrtf(var + 1e-5f);
        for (int i = 0; i < S; ++i) yr[i] *= sc;
        for (int i = 0; i < S; ++i) yr[i] = yr[i] * g[i];
        for (int i = 0; i < S; ++i) yr[i] = yr[i] + b[i];
    }
}

// soft-max of one row with an optional additive mask, reference rules (W/ggml.c:11116-11201)
void softmax_row(float * p, int n) {
    float mx = -INFINITY; for (int i = 0; i < n; ++i) mx = std::max(mx, p[i]);
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        if (p[i] == -INFINITY) { p[i] = 0.0f; continue; }
        const float v = h2f(g_exp[f2h(p[i] - mx)]);
        sum += (double) v; p[i] = v;
    }
    const float inv = (float) (1.0 / sum);
    for (int i = 0; i < n; ++i) p[i] *= inv;
}

// attention for nq query rows against nk keys of one layer.  q f32 [nq][S] (rounded to f16 per row as the
// mul_mat operand), k f16 [nk][S] (row = key), vt f16 per head transposed [S][ldv] (column = key).
// mask [nq][nk] or null.  out f32 [nq][S].
void attention(const Ctx & c, const float * q, int nq, const f16 * k, const f16 * vt, int ldv, int nk, int S, int H,
               float post_scale, const float * mask, float * out) {
    std::vector<f16> q16((size_t) nq * S);
    for (int j = 0; j < nq; ++j) row_to_f16(q + (size_t) j * S, q16.data() + (size_t) j * S, S);
    parallel_for(nq * H, c.n_threads, [&](int a, int b) {
        std::vector<float> sc(nk); std::vector<f16> p16(nk);
        for (int w = a; w < b; ++w) {
            const int h = w / nq, j = w % nq;
            for (int i = 0; i < nk; ++i) sc[i] = dot_f16(64, k + (size_t) i * S + h * 64, q16.data() + (size_t) j * S + h * 64);
            if (post_scale != 1.0f) for (int i = 0; i < nk; ++i) sc[i] *= post_scale;
            if (mask) for (int i = 0; i < nk; ++i) sc[i] += mask[(size_t) j * nk + i];
            softmax_row(sc.data(), nk);
            row_to_f16(sc.data(), p16.data(), nk);
            for (int d = 0; d < 64; ++d) out[(size_t) j * S + h * 64 + d] = dot_f16(nk, vt + (size_t) (h * 64 + d) * ldv, p16.data());
        }
    });
}

const f16 * W16(const Ctx & c, const std::string & n) { return c.m.get(n).h.data(); }
const float * F32(const Ctx & c, const std::string & n) { return c.m.get(n).f.data(); }

// out[j][i] = W[i][:] . act[j][:] with the reference's operand rule for the weight's type (SURVEY App. B rule 1):
// f16 weights -> activations rounded to f16; quantised weights -> activations to q8_0 / q8_1 blocks, integer dot
void matmul_w(const Ctx & c, const std::string & name, int N, int K, const float * act, int M, float * out) {
    const Tensor & t = c.m.get(name);
    if (!t.qtype) { matmul_f32act(c, t.h.data(), N, K, act, M, out); return; }
    const int nb = K / 32, has_s = port_q_act_has_sum(t.qtype);
    std::vector<int8_t> qs((size_t) M * K); std::vector<float> d((size_t) M * nb), s((size_t) M * nb);
    for (int j = 0; j < M; ++j) port_quantize_row_q8(act + (size_t) j * K, K, has_s, qs.data() + (size_t) j * K, d.data() + (size_t) j * nb, s.data() + (size_t) j * nb);
    const size_t rb = (size_t) nb * port_q_block_bytes(t.qtype);
    parallel_for(M, c.n_threads, [&](int a, int b) {
        for (int j = a; j < b; ++j) for (int i = 0; i < N; ++i)
            out[(size_t) j * N + i] = port_vec_dot_q(t.qtype, K, t.q.data() + (size_t) i * rb, qs.data() + (size_t) j * K, d.data() + (size_t) j * nb, s.data() + (size_t) j * nb);
    });
}

// ------------------------------------------------------------------------------------------------ encoder
void encode(Ctx & c, int mel_offset, int audio_ctx) {
    init_tables();
    const Model & m = c.m;
    const int T = audio_ctx > 0 ? audio_ctx : m.n_audio_ctx, S = m.S, H = m.H, nm = m.n_mels;
    c.T = T;
    // mel slice [nm][2T], zero beyond n_len
    std::vector<float> mel((size_t) nm * 2 * T, 0.0f);
    {
        const int i0 = std::min(mel_offset, c.n_len), i1 = std::min(mel_offset + 2 * T, c.n_len);
        for (int j = 0; j < nm; ++j) for (int i = i0; i < i1; ++i) mel[(size_t) j * 2 * T + (i - i0)] = c.mel[(size_t) j * c.n_len + i];
    }
    // conv1: im2col rows [t][ic*3 + k] f16, stride 1, pad 1
    auto conv = [&](const float * in, int IC, int IW, int stride, const std::string & wname, const std::string & bname, std::vector<float> & out) {
        const int OW = (IW + 2 - 3) / stride + 1, K = IC * 3;
        std::vector<f16> col((size_t) OW * K);
        for (int t = 0; t < OW; ++t) for (int ic = 0; ic < IC; ++ic) for (int k = 0; k < 3; ++k) {
            const int ii = t * stride + k - 1;
            col[(size_t) t * K + ic * 3 + k] = (ii < 0 || ii >= IW) ? (f16) 0 : f2h(in[(size_t) ic * IW + ii]);
        }
        const f16 * W = W16(c, wname); const float * b = F32(c, bname);
        out.assign((size_t) S * OW, 0.0f);                           // [oc][t]
        parallel_for(S, c.n_threads, [&](int a, int bb) {
            for (int oc = a; oc < bb; ++oc) for (int t = 0; t < OW; ++t) {
                const float v = dot_f16(K, col.data() + (size_t) t * K, W + (size_t) oc * K) + b[oc];
                out[(size_t) oc * OW + t] = gelu_tab(v);
            }
        });
        return OW;
    };
    std::vector<float> c1, c2;
    conv(mel.data(), nm, 2 * T, 1, "encoder.conv1.weight", "encoder.conv1.bias", c1);     // [S][2T]
    conv(c1.data(), S, 2 * T, 2, "encoder.conv2.weight", "encoder.conv2.bias", c2);       // [S][T]
    c.embd_conv.assign((size_t) T * S, 0.0f);
    std::vector<float> x((size_t) T * S);
    const float * pe = F32(c, "encoder.positional_embedding");
    for (int t = 0; t < T; ++t) for (int s = 0; s < S; ++s) {
        c.embd_conv[(size_t) t * S + s] = c2[(size_t) s * T + t];
        x[(size_t) t * S + s] = pe[(size_t) t * S + s] + c2[(size_t) s * T + t];
    }
    std::vector<float> xn((size_t) T * S), q((size_t) T * S), kf((size_t) T * S), vf((size_t) T * S), att((size_t) T * S), h1((size_t) T * 4 * S), tmp((size_t) T * S);
    std::vector<f16> k16((size_t) T * S), vt16((size_t) S * T);
    const float kq = 1.0f / sqrtf((float) S / H);
    for (int il = 0; il < m.La; ++il) {
        const std::string p = "encoder.blocks." + std::to_string(il) + ".";
        layernorm(x.data(), T, S, F32(c, p + "attn_ln.weight"), F32(c, p + "attn_ln.bias"), xn.data());
        matmul_w(c, p + "attn.query.weight", S, S, xn.data(), T, q.data());  add_bias(q.data(), T, S, F32(c, p + "attn.query.bias"));
        matmul_w(c, p + "attn.key.weight",   S, S, xn.data(), T, kf.data());
        matmul_w(c, p + "attn.value.weight", S, S, xn.data(), T, vf.data()); add_bias(vf.data(), T, S, F32(c, p + "attn.value.bias"));
        for (int t = 0; t < T; ++t) row_to_f16(kf.data() + (size_t) t * S, k16.data() + (size_t) t * S, S);
        for (int t = 0; t < T; ++t) for (int s = 0; s < S; ++s) vt16[(size_t) s * T + t] = f2h(vf[(size_t) t * S + s]);
        attention(c, q.data(), T, k16.data(), vt16.data(), T, T, S, H, kq, nullptr, att.data());
        matmul_w(c, p + "attn.out.weight", S, S, att.data(), T, tmp.data()); add_bias(tmp.data(), T, S, F32(c, p + "attn.out.bias"));
        for (size_t i = 0; i < x.size(); ++i) x[i] = tmp[i] + x[i];
        layernorm(x.data(), T, S, F32(c, p + "mlp_ln.weight"), F32(c, p + "mlp_ln.bias"), xn.data());
        matmul_w(c, p + "mlp.0.weight", 4 * S, S, xn.data(), T, h1.data()); add_bias(h1.data(), T, 4 * S, F32(c, p + "mlp.0.bias"));
        for (float & v : h1) v = gelu_tab(v);
        matmul_w(c, p + "mlp.2.weight", S, 4 * S, h1.data(), T, tmp.data()); add_bias(tmp.data(), T, S, F32(c, p + "mlp.2.bias"));
        for (size_t i = 0; i < x.size(); ++i) x[i] = tmp[i] + x[i];
    }
    c.embd_enc.assign((size_t) T * S, 0.0f);
    layernorm(x.data(), T, S, F32(c, "encoder.ln_post.weight"), F32(c, "encoder.ln_post.bias"), c.embd_enc.data());
    // cross K/V
    c.cross_k.assign((size_t) m.Lt * T * S, 0); c.cross_v.assign((size_t) m.Lt * T * S, 0);
    const float ks = powf((float) S / H, -0.25f);
    for (int il = 0; il < m.Lt; ++il) {
        const std::string p = "decoder.blocks." + std::to_string(il) + ".";
        matmul_w(c, p + "cross_attn.key.weight", S, S, c.embd_enc.data(), T, kf.data());
        for (float & v : kf) v *= ks;
        matmul_w(c, p + "cross_attn.value.weight", S, S, c.embd_enc.data(), T, vf.data()); add_bias(vf.data(), T, S, F32(c, p + "cross_attn.value.bias"));
        row_to_f16(kf.data(), c.cross_k.data() + (size_t) il * T * S, T * S);
        row_to_f16(vf.data(), c.cross_v.data() + (size_t) il * T * S, T * S);
    }
    c.n_past = 0;
}

// ------------------------------------------------------------------------------------------------ decoder
void decode(Ctx & c, const int32_t * tokens, int n, int n_past) {
    init_tables();
    const Model & m = c.m;
    const int S = m.St, H = m.Ht, NV = m.n_vocab, NC = m.n_text_ctx, T = c.T;
    if (c.self_k.empty()) { c.self_k.assign((size_t) m.Lt * NC * S, 0); c.self_v.assign((size_t) m.Lt * NC * S, 0); }
    const int n_kv = n_past + n;
    const Tensor & tte = c.m.get("decoder.token_embedding.weight");
    const f16 * te = tte.h.data(); const float * pe = F32(c, "decoder.positional_embedding");
    std::vector<float> x((size_t) n * S), xn((size_t) n * S), q((size_t) n * S), kf((size_t) n * S), vf((size_t) n * S), att((size_t) n * S), tmp((size_t) n * S), h1((size_t) n * 4 * S);
    if (tte.qtype) {                                            // get_rows on a quantised matrix: the row dequantised to f32
        const size_t rb = (size_t) (S / 32) * port_q_block_bytes(tte.qtype);
        std::vector<float> row(S);
        for (int j = 0; j < n; ++j) {
            port_dequantize_row(tte.qtype, tte.q.data() + (size_t) tokens[j] * rb, row.data(), S);
            for (int s = 0; s < S; ++s) x[(size_t) j * S + s] = row[s] + pe[(size_t) (n_past + j) * S + s];
        }
    } else
    for (int j = 0; j < n; ++j) for (int s = 0; s < S; ++s) x[(size_t) j * S + s] = h2f(te[(size_t) tokens[j] * S + s]) + pe[(size_t) (n_past + j) * S + s];
    std::vector<float> mask((size_t) n * n_kv, 0.0f);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n_kv; ++i) if (i > n_past + j) mask[(size_t) j * n_kv + i] = -INFINITY;
    const float ks = powf((float) S / H, -0.25f);
    std::vector<f16> vt;
    for (int il = 0; il < m.Lt; ++il) {
        const std::string p = "decoder.blocks." + std::to_string(il) + ".";
        f16 * ck = c.self_k.data() + (size_t) il * NC * S, * cv = c.self_v.data() + (size_t) il * NC * S;
        layernorm(x.data(), n, S, F32(c, p + "attn_ln.weight"), F32(c, p + "attn_ln.bias"), xn.data());
        matmul_w(c, p + "attn.query.weight", S, S, xn.data(), n, q.data()); add_bias(q.data(), n, S, F32(c, p + "attn.query.bias"));
        for (float & v : q) v *= ks;
        matmul_w(c, p + "attn.key.weight", S, S, xn.data(), n, kf.data());
        for (float & v : kf) v *= ks;
        matmul_w(c, p + "attn.value.weight", S, S, xn.data(), n, vf.data()); add_bias(vf.data(), n, S, F32(c, p + "attn.value.bias"));
        row_to_f16(kf.data(), ck + (size_t) n_past * S, n * S);
        row_to_f16(vf.data(), cv + (size_t) n_past * S, n * S);
        vt.assign((size_t) S * n_kv, 0);
        for (int i = 0; i < n_kv; ++i) for (int s = 0; s < S; ++s) vt[(size_t) s * n_kv + i] = cv[(size_t) i * S + s];
        attention(c, q.data(), n, ck, vt.data(), n_kv, n_kv, S, H, 1.0f, mask.data(), att.data());
        matmul_w(c, p + "attn.out.weight", S, S, att.data(), n, tmp.data()); add_bias(tmp.data(), n, S, F32(c, p + "attn.out.bias"));
        for (size_t i = 0; i < x.size(); ++i) x[i] = tmp[i] + x[i];
        layernorm(x.data(), n, S, F32(c, p + "cross_attn_ln.weight"), F32(c, p + "cross_attn_ln.bias"), xn.data());
        matmul_w(c, p + "cross_attn.query.weight", S, S, xn.data(), n, q.data()); add_bias(q.data(), n, S, F32(c, p + "cross_attn.query.bias"));
        for (float & v : q) v *= ks;
        vt.assign((size_t) S * T, 0);
        const f16 * xv = c.cross_v.data() + (size_t) il * T * S;
        for (int i = 0; i < T; ++i) for (int s = 0; s < S; ++s) vt[(size_t) s * T + i] = xv[(size_t) i * S + s];
        attention(c, q.data(), n, c.cross_k.data() + (size_t) il * T * S, vt.data(), T, T, S, H, 1.0f, nullptr, att.data());
        matmul_w(c, p + "cross_attn.out.weight", S, S, att.data(), n, tmp.data()); add_bias(tmp.data(), n, S, F32(c, p + "cross_attn.out.bias"));
        for (size_t i = 0; i < x.size(); ++i) x[i] = tmp[i] + x[i];
        layernorm(x.data(), n, S, F32(c, p + "mlp_ln.weight"), F32(c, p + "mlp_ln.bias"), xn.data());
        matmul_w(c, p + "mlp.0.weight", 4 * S, S, xn.data(), n, h1.data()); add_bias(h1.data(), n, 4 * S, F32(c, p + "mlp.0.bias"));
        for (float & v : h1) v = gelu_tab(v);
        matmul_w(c, p + "mlp.2.weight", S, 4 * S, h1.data(), n, tmp.data()); add_bias(tmp.data(), n, S, F32(c, p + "mlp.2.bias"));
        for (size_t i = 0; i < x.size(); ++i) x[i] = tmp[i] + x[i];
    }
    layernorm(x.data(), n, S, F32(c, "decoder.ln.weight"), F32(c, "decoder.ln.bias"), xn.data());
    // logits of the last row only (the reference computes all rows and copies out the flagged one)
    c.logits.resize(NV);
    if (tte.qtype) {
        const int nb = S / 32, has_s = port_q_act_has_sum(tte.qtype);
        std::vector<int8_t> qs(S); std::vector<float> d(nb), sa(nb);
        port_quantize_row_q8(xn.data() + (size_t) (n - 1) * S, S, has_s, qs.data(), d.data(), sa.data());
        const size_t rb = (size_t) nb * port_q_block_bytes(tte.qtype);
        parallel_for(NV, c.n_threads, [&](int a, int b) { for (int i = a; i < b; ++i) c.logits[i] = port_vec_dot_q(tte.qtype, S, tte.q.data() + (size_t) i * rb, qs.data(), d.data(), sa.data()); });
    } else {
    std::vector<f16> a16(S); row_to_f16(xn.data() + (size_t) (n - 1) * S, a16.data(), S);
    parallel_for(NV, c.n_threads, [&](int a, int b) { for (int i = a; i < b; ++i) c.logits[i] = dot_f16(S, te + (size_t) i * S, a16.data()); });
    }
    c.n_past = n_past + n;
}

} // namespace

extern "C" {

void * port_init(const void * buf, size_t n, int n_threads) {
    Ctx * c = new Ctx();
    if (!load((const uint8_t *) buf, n, c->m)) { delete c; return nullptr; }
    c->n_threads = n_threads > 0 ? n_threads : 4;
    init_tables();
    return c;
}
void port_free(void * p) { delete (Ctx *) p; }
void port_set_threads(void * p, int n) { ((Ctx *) p)->n_threads = std::max(1, n); }

int port_pcm_to_mel(void * p, const float * pcm, int n) { pcm_to_mel(*(Ctx *) p, pcm, n); return ((Ctx *) p)->n_len; }
int port_set_mel(void * p, const float * mel, int n_len, int n_mel) {
    Ctx & c = *(Ctx *) p; if (n_mel != c.m.filt_mel) return -1;
    c.mel.assign(mel, mel + (size_t) n_len * n_mel); c.n_len = n_len; c.n_len_org = n_len; return 0;
}
int port_mel_dims(void * p, int * n_len, int * n_len_org, int * n_mel) {
    Ctx & c = *(Ctx *) p; *n_len = c.n_len; *n_len_org = c.n_len_org; *n_mel = c.m.filt_mel; return (int) c.mel.size();
}
int port_encode(void * p, int mel_offset, int audio_ctx) { encode(*(Ctx *) p, mel_offset, audio_ctx); return 0; }
int port_decode(void * p, const int32_t * tokens, int n, int n_past, float * logits_last) {
    Ctx & c = *(Ctx *) p; decode(c, tokens, n, n_past);
    if (logits_last) memcpy(logits_last, c.logits.data(), c.logits.size() * 4);
    return 0;
}
// same tensor names / layouts as wmi_get_tensor (include/wmi_device.h)
int port_get_tensor(void * p, const char * name, float * dst, int n) {
    Ctx & c = *(Ctx *) p; const std::string nm(name);
    const std::vector<float> * f = nullptr; const std::vector<f16> * h = nullptr;
    if (nm == "mel") f = &c.mel; else if (nm == "embd_conv") f = &c.embd_conv; else if (nm == "embd_enc") f = &c.embd_enc;
    else if (nm == "cross_k") h = &c.cross_k; else if (nm == "cross_v") h = &c.cross_v; else return -1;
    const int count = (int) (f ? f->size() : h->size());
    if (!dst) return count;
    n = std::min(n, count);
    if (f) memcpy(dst, f->data(), (size_t) n * 4); else for (int i = 0; i < n; ++i) dst[i] = h2f((*h)[i]);
    return n;
}
void port_tables(uint16_t * gelu, uint16_t * expt) { init_tables(); memcpy(gelu, g_gelu, sizeof(g_gelu)); memcpy(expt, g_exp, sizeof(g_exp)); }
int port_n_vocab(void * p) { return ((Ctx *) p)->m.n_vocab; }
int port_hparam(void * p, int which) {
    const Model & m = ((Ctx *) p)->m;
    const int v[] = { m.n_vocab, m.n_audio_ctx, m.S, m.H, m.La, m.n_text_ctx, m.St, m.Ht, m.Lt, m.n_mels };
    return which >= 0 && which < 10 ? v[which] : -1;
}
// greedy transcription without logit filters: PCM -> mel -> encode -> prompt + n_steps argmax steps.
// Used as the CPU-baseline workload when the compiled reference is absent.
int port_greedy(void * p, const float * pcm, int n, int32_t sot, int n_steps, int32_t * out_tokens) {
    Ctx & c = *(Ctx *) p;
    pcm_to_mel(c, pcm, n); encode(c, 0, 0);
    int32_t tok = sot; int n_past = 0;
    for (int i = 0; i <= n_steps; ++i) {
        decode(c, &tok, 1, n_past); n_past += 1;
        tok = (int32_t) (std::max_element(c.logits.begin(), c.logits.begin() + 50256) - c.logits.begin());
        if (out_tokens) out_tokens[i] = tok;
    }
    return n_steps + 1;
}

} // extern "C"
