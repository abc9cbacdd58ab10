// Block-quantised models (BASELINE configs[4]: large-v3 q5_1): the encoder and decoder layer loops with the weights kept in
// their ggml blocks (reference: the same graphs, W/whisper.cpp:1756-2074 and :2148-2505, whose
// mul_mat nodes take the quantised branch W/ggml.c:9841-9857 for every 2-D weight).
//
// What changes against the f16 loops of device.cpp is the operand rule of every projection: the reference quantises the
// f32 activation tensor of the mul_mat to q8 blocks, so
//   * LayerNorm feeds the quantiser in f32 (no f16 rounding in between),
//   * the attention outputs are kept f32 (KQV_merged is an f32 tensor in the reference) and quantised from there,
//   * the GELU output is f16-representable by construction (f16 table), so the f16 buffer is widened exactly.
// The attention products themselves (K, V caches, Q) stay f16 x f16 as in the f16 models, and so does the conv front-end
// (the quantize tool leaves the 3-D conv weights f16, W/examples/common-ggml.cpp:112-131).

#include "wmi.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace wmi {


namespace {

// activation source of a projection
struct Src { const float * x32 = nullptr; const float * ln_g = nullptr, * ln_b = nullptr; const __half * x16 = nullptr; };

// rows * K quantised rows + scales + reduction scratch must fit the 160 KB of LDS for the weight-streaming kernel
bool rows_fit(int n, int K) {
    if (n > 32) return false;
    const int r8 = n <= 8 ? 8 : n <= 16 ? 16 : 32, nb = K / 32;
    const size_t smem = (((size_t) n * (K + 16) + 15) & ~(size_t) 15) + (size_t) 2 * nb * r8 * 4 + (size_t) 16 * 32 * r8 * 4;
    return smem <= 150 * 1024;
}

// mlp.2 of the decoder with K split over two workgroups per row group (kernels.h GemvArgs::ksplit; WMI_Q_KSPLIT=0: off, A/B)
bool fc2_ksplit(k::GemvArgs & g, float * kpart) {
    static const bool off = getenv("WMI_Q_KSPLIT") && atoi(getenv("WMI_Q_KSPLIT")) == 0;
    static const bool dbg = getenv("WMI_DEBUG_KSPLIT") != nullptr;
    const bool ok = !off && kpart && k::qrows_ksplit_ok(g, 2);
    if (dbg) fprintf(stderr, "[wmi] mlp.2 K split: %s (n %d K %d N %d a16 %p epi %d)\n", ok ? "yes" : "no", g.n, g.K, g.N, (const void *) g.a16, g.epi);
    if (!ok) return false;
    g.ksplit = 2; g.kpart = kpart;
    static const bool drop = getenv("WMI_DEBUG_KSPLIT_DROP") != nullptr;      // debug: the consumers ignore the pending half (results must change)
    return !drop;
}

} // namespace

bool encode_layers_q(whisper_context & ctx, int T) {
    DeviceState & d = ctx.state->dev; const int S = ctx.model.hp.n_audio_state;
    EncBufsQ e{};
    e.T = T; e.nb = 1; e.Tpad = d.Tpad; e.x = d.x; e.q = d.q; e.k = d.k; e.vt = d.vt; e.h = d.h; e.att32 = d.att32;
    e.enc_out = d.enc_out; e.enc_out_h = d.enc_out_h; e.kvc_k = d.kvc_k; e.kvc_v = d.kvc_v;
    e.A = q8_rows(d, S); e.A4 = q8_rows(d, 4 * S);
    return encode_layers_q_on(ctx, e, d.stream);
}

bool encode_layers_q_on(whisper_context & ctx, const EncBufsQ & e, hipStream_t s) {
    const Weights & w = ctx.w; const HParams & hp = ctx.model.hp;
    const int S = hp.n_audio_state, H = hp.n_audio_head, La = hp.n_audio_layer, Lt = hp.n_text_layer;
    const int T = e.T, M = e.nb * e.T;
    k::Q8Rows A = e.A, A4 = e.A4;
    A.w_resident_ok = A4.w_resident_ok = true;              // the encoder's matrices: resident f16 images beside the blocks (k_quant.hip)
    const int qt = w.qtype;
    const float kq_scale = 1.0f / sqrtf((float) S / H);
    for (int il = 0; il < La; ++il) {
        const EncLayerW & l = w.enc[il];
        // (each quantiser launch also writes the f16 image of the weights its projection multiplies with, see k::quantize_rows)
        k::Q8Rows Aq = A;
        Aq.wdeq_ready = k::quantize_rows(e.x, nullptr, M, S, l.ln1_g, l.ln1_b, hp.eps, qt, A, nullptr, nullptr, s, &l.q_qkv, 3 * S); Aq.wdeq_of = l.q_qkv.tiles;
        {
            k::GemmArgs a{};
            a.M = M; a.N = 3 * S; a.K = S; a.bias = l.b_qkv;
            a.C = e.q; a.ldc = S; a.aux = e.k; a.ldaux = S; a.aux2 = e.vt; a.ldaux2 = e.Tpad; a.S = S;
            a.rows_per_chunk = T; a.chunk_stride_aux2 = (int64_t) S * e.Tpad;
            k::qgemm(k::EPI_QKV_ENC, a, Aq, l.q_qkv, s);
        }
        k::attn_encoder(e.q, e.k, e.vt, T, e.Tpad, S, H, kq_scale, nullptr, s, e.nb, e.att32);
        Aq.wdeq_ready = k::quantize_rows(e.att32, nullptr, M, S, nullptr, nullptr, 0.f, qt, A, nullptr, nullptr, s, &l.q_o, S); Aq.wdeq_of = l.q_o.tiles;
        {
            k::GemmArgs a{};
            a.M = M; a.N = S; a.K = S; a.bias = l.b_o; a.C = e.x; a.ldc = S; a.resid = e.x; a.ldr = S;
            k::qgemm(k::EPI_F32_BIAS_RESID, a, Aq, l.q_o, s);
        }
        Aq.wdeq_ready = k::quantize_rows(e.x, nullptr, M, S, l.ln2_g, l.ln2_b, hp.eps, qt, A, nullptr, nullptr, s, &l.q_fc1, 4 * S); Aq.wdeq_of = l.q_fc1.tiles;
        {
            k::GemmArgs a{};
            a.M = M; a.N = 4 * S; a.K = S; a.bias = l.b_fc1; a.C = e.h; a.ldc = 4 * S;
            k::qgemm(k::EPI_F16_BIAS_GELU, a, Aq, l.q_fc1, s);
        }
        k::Q8Rows A4q = A4;
        A4q.wdeq_ready = k::quantize_rows(nullptr, e.h, M, 4 * S, nullptr, nullptr, 0.f, qt, A4, nullptr, nullptr, s, &l.q_fc2, S); A4q.wdeq_of = l.q_fc2.tiles;
        {
            k::GemmArgs a{};
            a.M = M; a.N = S; a.K = 4 * S; a.bias = l.b_fc2; a.C = e.x; a.ldc = S; a.resid = e.x; a.ldr = S;
            k::qgemm(k::EPI_F32_BIAS_RESID, a, A4q, l.q_fc2, s);
        }
    }
    // ln_post -> embd_enc (f32, kept for inspection) and its q8 image; cross K/V of every decoder layer in one GEMM
    k::quantize_rows(e.x, nullptr, M, S, w.e_ln_g, w.e_ln_b, hp.eps, qt, A, e.enc_out, e.enc_out_h, s);
    {
        k::GemmArgs a{};
        a.M = M; a.N = Lt * 2 * S; a.K = S; a.bias = w.b_ckv;
        a.C = e.kvc_k; a.ldc = S; a.aux = e.kvc_v; a.ldaux = S; a.S = S; a.layer_stride = (int64_t) M * S;
        a.scale = powf((float) S / H, -0.25f);
        k::qgemm(k::EPI_CROSS_KV, a, A, w.q_ckv, s);
    }
    return true;
}

bool decode_layers_q(whisper_context & ctx, int n, int n_kv, int kv_head, int Tc, const std::vector<int> & rows) {
    State & st = *ctx.state; DeviceState & d = st.dev; const Weights & w = ctx.w; const HParams & hp = ctx.model.hp;
    KVCache & kv = st.kv_self;
    const int S = hp.n_text_state, H = hp.n_text_head, Lt = hp.n_text_layer, NV = hp.n_vocab, n_ctx = (int) kv.size;
    hipStream_t s = d.stream;
    const int qt = w.qtype;
    const float kq_scale = powf((float) S / H, -0.25f);

    k::qdec_embed(d.d_tokens, d.d_pos, n, S, w.q_te, w.d_pe, d.dx, s);

    // y = W . q8(src) with a fused epilogue: <= 32 rows stream the weight tiles once (rows quantised in the kernel's prologue);
    // prompt-sized batches quantise once and go through the tiled GEMM
    static const bool no_pf = getenv("WMI_NO_PREFETCH") != nullptr;            // A/B knob
    const k::QMat * pfW = nullptr; int pfN = 0, pfK = 0;                       // the next weight-streaming launch's matrix (k_qrows prefetch)
    auto next = [&](const k::QMat & W, int N, int K) { pfW = no_pf ? nullptr : &W; pfN = N; pfK = K; };
    auto set_pf = [&](k::GemvArgs & g) {
        if (pfW && pfW->tiles && pfN <= 8192) {
            g.pf_ptr = pfW->tiles; g.pf_group_bytes = (uint32_t) ((size_t) (pfK / 64) * k::q_tile_bytes(pfW->qtype)); g.pf_groups = (uint32_t) ((pfN + 31) / 32);
        }
        pfW = nullptr;
    };
    // pend: the upper-K-half sums of the last mlp.2 that the residual stream d.dx still lacks (GemvArgs::ksplit): taken by the launches
    // that read d.dx until the next out projection has written the row whole again
    const float * pend = nullptr;
    auto proj = [&](int epi, const Src & src, int K, int N, const k::QMat & W, const float * bias, void * C, int ldc,
                    const float * resid, void * aux, int ldaux, void * aux2, int ldaux2, float scale, bool is_fc2 = false) {
        if (rows_fit(n, K)) {
            k::GemvArgs g{};
            set_pf(g);
            g.x32 = src.x32; g.ln_g = src.ln_g; g.ln_b = src.ln_b; g.eps = hp.eps; g.a16 = src.x16; g.n = n; g.K = K; g.N = N;
            g.bias = bias; g.epi = epi; g.C = C; g.ldc = ldc; g.resid = resid; g.ldr = S; g.aux = aux; g.ldaux = ldaux;
            g.aux2 = aux2; g.ldaux2 = ldaux2; g.scale = scale; g.S = S;
            if (pend && (src.ln_g || resid)) g.pend = pend;
            if (resid) pend = nullptr;
            if (is_fc2 && fc2_ksplit(g, d.xattn)) pend = d.xattn;
            k::qrows(g, src.ln_g ? nullptr : src.x32, W, s);
        } else {
            const k::Q8Rows A = q8_rows(d, K);
            k::quantize_rows(src.x32, src.x16, n, K, src.ln_g, src.ln_b, hp.eps, qt, A, nullptr, nullptr, s);
            k::GemmArgs a{};
            a.M = n; a.N = N; a.K = K; a.bias = bias; a.C = C; a.ldc = ldc; a.resid = resid; a.ldr = S;
            a.aux = aux; a.ldaux = ldaux; a.aux2 = aux2; a.ldaux2 = ldaux2; a.scale = scale; a.S = S;
            k::qgemm(epi, a, A, W, s);
        }
    };

    for (int il = 0; il < Lt; ++il) {
        const DecLayerW & l = w.dec[il];
        __half * ck = kv.k + ((size_t) il * n_ctx) * S, * cv = kv.v + ((size_t) il * n_ctx) * S;
        Src ln1; ln1.x32 = d.dx; ln1.ln_g = l.ln1_g; ln1.ln_b = l.ln1_b;
        next(l.q_o, S, S);
        proj(k::EPI_QKV_DEC, ln1, S, 3 * S, l.q_qkv, l.b_qkv, d.dq, S, nullptr, ck + (size_t) kv_head * S, S, cv + (size_t) kv_head * S, S, kq_scale);
        k::attn_decoder(d.dq, n, S, H, ck, cv, n_kv, d.d_mask, n_kv, nullptr, s, nullptr, 0, d.datt32);
        Src att; att.x32 = d.datt32;
        next(l.q_cq, S, S);
        proj(k::EPI_F32_BIAS_RESID, att, S, S, l.q_o, l.b_o, d.dx, S, d.dx, nullptr, 0, nullptr, 0, 0.f);
        Src ln2; ln2.x32 = d.dx; ln2.ln_g = l.ln2_g; ln2.ln_b = l.ln2_b;
        next(l.q_co, S, S);
        if (rows_fit(n, S)) {                     // the out projection combines the key-slice partials in its prologue (one launch fewer)
            const float * po = nullptr, * pl = nullptr, * pm = nullptr; int ns = 0;
            // the cross query inside the attention launch (one launch fewer again), else as its own launch
            if (!k::qattn_cross_qsplit_partials(d.dx, l.ln2_g, l.ln2_b, hp.eps, l.q_cq, l.b_cq, kq_scale, n, S, H, d.kvc_k + (size_t) il * Tc * S,
                                                d.kvc_v + (size_t) il * Tc * S, Tc, d.xattn, &po, &pl, &pm, &ns, s, 0, pfW ? *pfW : k::QMat{}, pfN, pfK)) {
                proj(k::EPI_Q_SCALED, ln2, S, S, l.q_cq, l.b_cq, d.dq, S, nullptr, nullptr, 0, nullptr, 0, kq_scale);
                k::attn_cross_split_partials(d.dq, n, S, H, d.kvc_k + (size_t) il * Tc * S, d.kvc_v + (size_t) il * Tc * S, Tc, d.xattn, &po, &pl, &pm, &ns, s);
            }
            k::GemvArgs g{};
            g.eps = hp.eps; g.n = n; g.K = S; g.N = S; g.bias = l.b_co; g.epi = k::EPI_F32_BIAS_RESID; g.C = d.dx; g.ldc = S; g.resid = d.dx; g.ldr = S;
            g.S = S; g.comb_o = po; g.comb_l = pl; g.comb_m = pm; g.comb_ns = ns;
            next(l.q_fc1, 4 * S, S); set_pf(g);
            k::qrows(g, nullptr, l.q_co, s);
        } else {
        proj(k::EPI_Q_SCALED, ln2, S, S, l.q_cq, l.b_cq, d.dq, S, nullptr, nullptr, 0, nullptr, 0, kq_scale);
        k::attn_cross_split(d.dq, n, S, H, d.kvc_k + (size_t) il * Tc * S, d.kvc_v + (size_t) il * Tc * S, Tc, d.xattn, nullptr, s, 0, d.datt32);
        proj(k::EPI_F32_BIAS_RESID, att, S, S, l.q_co, l.b_co, d.dx, S, d.dx, nullptr, 0, nullptr, 0, 0.f);
        }
        Src ln3; ln3.x32 = d.dx; ln3.ln_g = l.ln3_g; ln3.ln_b = l.ln3_b;
        next(l.q_fc2, S, 4 * S);
        proj(k::EPI_F16_BIAS_GELU, ln3, S, 4 * S, l.q_fc1, l.b_fc1, d.dh, 4 * S, nullptr, nullptr, 0, nullptr, 0, 0.f);
        Src hh; hh.x16 = d.dh;
        if (il + 1 < Lt) next(w.dec[il + 1].q_qkv, 3 * S, S);
        proj(k::EPI_F32_BIAS_RESID, hh, 4 * S, S, l.q_fc2, l.b_fc2, d.dx, S, d.dx, nullptr, 0, nullptr, 0, 0.f, true);
    }

    // final LN + logits for the flagged rows
    st.logits.resize((size_t) n * NV);
    for (size_t r0 = 0; r0 < rows.size(); r0 += 8) {
        const int nr = (int) std::min<size_t>(8, rows.size() - r0);
        k::GemvArgs g{};
        g.x32 = d.dx; g.ln_g = w.d_ln_g; g.ln_b = w.d_ln_b; g.eps = hp.eps; g.n = nr; g.K = S; g.N = NV;
        g.epi = k::EPI_LOGITS; g.C = d.logits; g.ldc = NV; g.rows = d.d_rows + r0; g.pend = pend;
        k::qrows(g, nullptr, w.q_te, s);
        if (d.keep_logits_on_device && rows.size() <= 8) continue;
        HIP_TRY(hipMemcpyAsync(d.pinned, d.logits, (size_t) nr * NV * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int r = 0; r < nr; ++r)
            memcpy(st.logits.data() + (size_t) rows[r0 + r] * NV, (const float *) d.pinned + (size_t) r * NV, (size_t) NV * 4);
    }
    return true;
}

// One greedy decode step of a block-quantised model as a fixed launch sequence (graph-replayable: every per-step quantity
// is read from DecStep on the device, device.cpp: decode_greedy_step): 7 launches per layer —
//   q|k|v (LayerNorm + q8 in the prologue) -> self-attention (f32 out) -> out projection -> cross-attention of the key slices with the
//   LayerNorm + cross query inside (k_xattn_fused_q; WMI_Q_XATTN_TWO_LAUNCHES: the query as its own launch) -> cross out projection
//   (combines the key slices in its prologue) -> mlp.0 -> mlp.2
void enqueue_greedy_step_q(whisper_context & ctx, int Tc) {
    State & st = *ctx.state; DeviceState & d = st.dev; const Weights & w = ctx.w; const HParams & hp = ctx.model.hp;
    KVCache & kv = st.kv_self;
    const int S = hp.n_text_state, H = hp.n_text_head, Lt = hp.n_text_layer, NV = hp.n_vocab, n_ctx = (int) kv.size;
    hipStream_t s = d.stream;
    const k::DecStep * stp = (const k::DecStep *) d.step_dev;
    const float kq_scale = powf((float) S / H, -0.25f);
    k::qdec_embed_step((const k::DecStep *) d.step_host, (k::DecStep *) d.step_dev, S, w.q_te, w.d_pe, d.dx, s);
    // pfW / pfN / pfK: the matrix of the next weight-streaming launch of the chain, prefetched by this one (k_qrows)
    static const bool no_pf = getenv("WMI_NO_PREFETCH") != nullptr;            // A/B knob
    const k::QMat * pfW = nullptr; int pfN = 0, pfK = 0;
    auto next = [&](const k::QMat & W, int N, int K) { pfW = no_pf ? nullptr : &W; pfN = N; pfK = K; };
    const float * pend = nullptr;                             // see decode_layers_q
    auto rows = [&](int epi, const Src & src, int K, int N, const k::QMat & W, const float * bias, void * C, int ldc, const float * resid,
                    void * aux, void * aux2, float scale, const int32_t * row_off, const float * co = nullptr, const float * cl = nullptr, int cns = 0, const float * cm = nullptr) {
        k::GemvArgs g{};
        if (pfW && pfW->tiles && pfN <= 8192) {             // (the vocabulary projection is 50 MB: not prefetched)
            g.pf_ptr = pfW->tiles; g.pf_group_bytes = (uint32_t) ((size_t) (pfK / 64) * k::q_tile_bytes(pfW->qtype)); g.pf_groups = (uint32_t) ((pfN + 31) / 32);
        }
        pfW = nullptr;
        g.x32 = src.x32; g.ln_g = src.ln_g; g.ln_b = src.ln_b; g.eps = hp.eps; g.a16 = src.x16; g.n = 1; g.K = K; g.N = N;
        g.bias = bias; g.epi = epi; g.C = C; g.ldc = ldc; g.resid = resid; g.ldr = S; g.aux = aux; g.ldaux = S; g.aux2 = aux2; g.ldaux2 = S;
        g.scale = scale; g.S = S; g.row_off = row_off; g.comb_o = co; g.comb_l = cl; g.comb_ns = cns; g.comb_m = cm;
        if (pend && (src.ln_g || (resid && !co))) g.pend = pend;
        if (resid && !co && !src.x16) pend = nullptr;                        // the self-attention's out projection writes the row whole again
        if (src.x16 && epi == k::EPI_F32_BIAS_RESID && fc2_ksplit(g, d.xattn)) pend = d.xattn;      // mlp.2
        k::qrows(g, src.ln_g ? nullptr : src.x32, W, s);
#ifdef WMI_QROWS_PROBE
        // (probe build, WMI_Q_DOUBLE=1) the same launch once more on the NEXT layer's matrix of the same kind: cold weights, warm
        // instruction cache — what of a launch's body is instruction fetch?  (results are garbage: timing only)
        static const bool dbl = getenv("WMI_Q_DOUBLE") != nullptr;
        if (dbl && &W >= (const k::QMat *) &w.dec[0] && &W < (const k::QMat *) &w.dec[Lt - 1]) {
            const k::QMat & W2 = *(const k::QMat *) ((const char *) &W + sizeof(DecLayerW));
            k::GemvArgs g2 = g; g2.pf_ptr = nullptr;
            k::qrows(g2, src.ln_g ? nullptr : src.x32, W2, s);
        }
#endif
    };
    for (int il = 0; il < Lt; ++il) {
        const DecLayerW & l = w.dec[il];
        __half * ck = kv.k + ((size_t) il * n_ctx) * S, * cv = kv.v + ((size_t) il * n_ctx) * S;
        Src ln1; ln1.x32 = d.dx; ln1.ln_g = l.ln1_g; ln1.ln_b = l.ln1_b;
        next(l.q_o, S, S);
        rows(k::EPI_QKV_DEC, ln1, S, 3 * S, l.q_qkv, l.b_qkv, d.dq, S, nullptr, ck, cv, kq_scale, &stp->kv_head);
        k::self_attn_rows(d.dq, 1, S, ck, cv, 0, &stp->n_kv, 0, hp.n_text_ctx, nullptr, s, d.datt32);
        Src att; att.x32 = d.datt32;
        next(l.q_cq, S, S);
        rows(k::EPI_F32_BIAS_RESID, att, S, S, l.q_o, l.b_o, d.dx, S, d.dx, nullptr, nullptr, 0.f, nullptr);
        Src ln2; ln2.x32 = d.dx; ln2.ln_g = l.ln2_g; ln2.ln_b = l.ln2_b;
        next(l.q_co, S, S);
        const float * po = nullptr, * pl = nullptr, * pm = nullptr; int ns = 0;
        if (k::qattn_cross_qsplit_partials(d.dx, l.ln2_g, l.ln2_b, hp.eps, l.q_cq, l.b_cq, kq_scale, 1, S, H, d.kvc_k + (size_t) il * Tc * S,
                                           d.kvc_v + (size_t) il * Tc * S, Tc, d.xattn, &po, &pl, &pm, &ns, s, 0, pfW ? *pfW : k::QMat{}, pfN, pfK)) pfW = nullptr;
        else {
            rows(k::EPI_Q_SCALED, ln2, S, S, l.q_cq, l.b_cq, d.dq, S, nullptr, nullptr, nullptr, kq_scale, nullptr);
            k::attn_cross_split_partials(d.dq, 1, S, H, d.kvc_k + (size_t) il * Tc * S, d.kvc_v + (size_t) il * Tc * S, Tc, d.xattn, &po, &pl, &pm, &ns, s);
        }
        next(l.q_fc1, 4 * S, S);
        rows(k::EPI_F32_BIAS_RESID, Src{}, S, S, l.q_co, l.b_co, d.dx, S, d.dx, nullptr, nullptr, 0.f, nullptr, po, pl, ns, pm);
        Src ln3; ln3.x32 = d.dx; ln3.ln_g = l.ln3_g; ln3.ln_b = l.ln3_b;
        next(l.q_fc2, S, 4 * S);
        rows(k::EPI_F16_BIAS_GELU, ln3, S, 4 * S, l.q_fc1, l.b_fc1, d.dh, 4 * S, nullptr, nullptr, nullptr, 0.f, nullptr);
        Src hh; hh.x16 = d.dh;
        if (il + 1 < Lt) next(w.dec[il + 1].q_qkv, 3 * S, S);
        rows(k::EPI_F32_BIAS_RESID, hh, 4 * S, S, l.q_fc2, l.b_fc2, d.dx, S, d.dx, nullptr, nullptr, 0.f, nullptr);
    }
    Src lnf; lnf.x32 = d.dx; lnf.ln_g = w.d_ln_g; lnf.ln_b = w.d_ln_b;
    rows(k::EPI_LOGITS, lnf, S, NV, w.q_te, nullptr, d.logits, NV, nullptr, nullptr, nullptr, 0.f, nullptr);
    k::filter_argmax(d.logits, d.ban_dev, stp, (k::SampleOut *) d.sample_dev, d.filter_scratch, s, (k::SampleOut *) d.sample_host);
}

// One lock-step greedy step of a block-quantised model: nb chunk rows through the same launches (batch.cpp: enqueue_rows_step is
// the f16 form).  Row r has its own self cache (+ r * cache_stride), cross-cache slice and step record.
void enqueue_rows_step_q(whisper_context & ctx, int nb) {
    BatchWork & b = *ctx.batch; const Weights & w = ctx.w; const HParams & hp = ctx.model.hp;
    const int S = hp.n_text_state, H = hp.n_text_head, Lt = hp.n_text_layer, NV = hp.n_vocab, n_ctx = hp.n_text_ctx;
    const int Tc = b.enc_T;
    hipStream_t s = ctx.state->dev.stream;
    const k::DecStep * stp = (const k::DecStep *) b.step_dev;
    const float kq_scale = powf((float) S / H, -0.25f);
    const int step_stride = (int) (sizeof(k::DecStep) / sizeof(int32_t));
    const int64_t cache_stride = (int64_t) Lt * n_ctx * S;
    const int64_t cross_layer = (int64_t) b.enc_rows * Tc * S;
    k::qdec_embed_step((const k::DecStep *) b.step_host, (k::DecStep *) b.step_dev, S, w.q_te, w.d_pe, b.dx, s, nb);
    static const bool no_pf = getenv("WMI_NO_PREFETCH") != nullptr;            // A/B knob
    const k::QMat * pfW = nullptr; int pfN = 0, pfK = 0;                       // the next weight-streaming launch's matrix (k_qrows prefetch)
    auto next = [&](const k::QMat & W, int N, int K) { pfW = no_pf ? nullptr : &W; pfN = N; pfK = K; };
    const float * pend = nullptr;                             // see decode_layers_q
    auto rows = [&](int epi, const Src & src, int K, int N, const k::QMat & W, const float * bias, void * C, int ldc, const float * resid,
                    void * aux, void * aux2, float scale, const int32_t * row_off) {
        k::GemvArgs g{};
        if (pfW && pfW->tiles && pfN <= 8192) {
            g.pf_ptr = pfW->tiles; g.pf_group_bytes = (uint32_t) ((size_t) (pfK / 64) * k::q_tile_bytes(pfW->qtype)); g.pf_groups = (uint32_t) ((pfN + 31) / 32);
        }
        pfW = nullptr;
        g.x32 = src.x32; g.ln_g = src.ln_g; g.ln_b = src.ln_b; g.eps = hp.eps; g.a16 = src.x16; g.n = nb; g.K = K; g.N = N;
        g.bias = bias; g.epi = epi; g.C = C; g.ldc = ldc; g.resid = resid; g.ldr = S; g.aux = aux; g.ldaux = S; g.aux2 = aux2; g.ldaux2 = S;
        g.scale = scale; g.S = S; g.row_off = row_off; g.lanes = 1; g.step_stride = step_stride; g.cache_row_stride = cache_stride;
        if (pend && (src.ln_g || resid)) g.pend = pend;       // (the cross-attention's out projection runs behind the self-attention's: pend is null by then)
        if (resid && !src.x16) pend = nullptr;
        if (src.x16 && epi == k::EPI_F32_BIAS_RESID && fc2_ksplit(g, b.xattn)) pend = b.xattn;      // mlp.2; the cross-attention scratch is idle until the next layer's cross-attention
        return g;
    };
    for (int il = 0; il < Lt; ++il) {
        const DecLayerW & l = w.dec[il];
        __half * ck = b.self_k + (size_t) il * n_ctx * S, * cv = b.self_v + (size_t) il * n_ctx * S;     // chunk 0; + r * cache_stride
        Src ln1; ln1.x32 = b.dx; ln1.ln_g = l.ln1_g; ln1.ln_b = l.ln1_b;
        next(l.q_o, S, S);
        k::qrows(rows(k::EPI_QKV_DEC, ln1, S, 3 * S, l.q_qkv, l.b_qkv, b.dq, S, nullptr, ck, cv, kq_scale, &stp->kv_head), nullptr, l.q_qkv, s);
        k::self_attn_rows(b.dq, nb, S, ck, cv, cache_stride, &stp->n_kv, step_stride, n_ctx, nullptr, s, b.datt32);
        Src att; att.x32 = b.datt32;
        next(l.q_cq, S, S);
        k::qrows(rows(k::EPI_F32_BIAS_RESID, att, S, S, l.q_o, l.b_o, b.dx, S, b.dx, nullptr, nullptr, 0.f, nullptr), att.x32, l.q_o, s);
        Src ln2; ln2.x32 = b.dx; ln2.ln_g = l.ln2_g; ln2.ln_b = l.ln2_b;
        next(l.q_co, S, S);
        const float * po = nullptr, * pl = nullptr, * pm = nullptr; int ns = 0;
        if (k::qattn_cross_qsplit_partials(b.dx, l.ln2_g, l.ln2_b, hp.eps, l.q_cq, l.b_cq, kq_scale, nb, S, H, b.kvc_k + (size_t) il * cross_layer,
                                           b.kvc_v + (size_t) il * cross_layer, Tc, b.xattn, &po, &pl, &pm, &ns, s, (int64_t) Tc * S,
                                           pfW ? *pfW : k::QMat{}, pfN, pfK)) pfW = nullptr;
        else {
            k::qrows(rows(k::EPI_Q_SCALED, ln2, S, S, l.q_cq, l.b_cq, b.dq, S, nullptr, nullptr, nullptr, kq_scale, nullptr), nullptr, l.q_cq, s);
            k::attn_cross_split_partials(b.dq, nb, S, H, b.kvc_k + (size_t) il * cross_layer, b.kvc_v + (size_t) il * cross_layer, Tc,
                                         b.xattn, &po, &pl, &pm, &ns, s, (int64_t) Tc * S);
        }
        {
            next(l.q_fc1, 4 * S, S);
            k::GemvArgs g = rows(k::EPI_F32_BIAS_RESID, Src{}, S, S, l.q_co, l.b_co, b.dx, S, b.dx, nullptr, nullptr, 0.f, nullptr);
            g.comb_o = po; g.comb_l = pl; g.comb_m = pm; g.comb_ns = ns;
            k::qrows(g, nullptr, l.q_co, s);
        }
        Src ln3; ln3.x32 = b.dx; ln3.ln_g = l.ln3_g; ln3.ln_b = l.ln3_b;
        next(l.q_fc2, S, 4 * S);
        k::qrows(rows(k::EPI_F16_BIAS_GELU, ln3, S, 4 * S, l.q_fc1, l.b_fc1, b.dh, 4 * S, nullptr, nullptr, nullptr, 0.f, nullptr), nullptr, l.q_fc1, s);
        Src hh; hh.x16 = b.dh;
        if (il + 1 < Lt) next(w.dec[il + 1].q_qkv, 3 * S, S);
        k::qrows(rows(k::EPI_F32_BIAS_RESID, hh, 4 * S, S, l.q_fc2, l.b_fc2, b.dx, S, b.dx, nullptr, nullptr, 0.f, nullptr), nullptr, l.q_fc2, s);
    }
    Src lnf; lnf.x32 = b.dx; lnf.ln_g = w.d_ln_g; lnf.ln_b = w.d_ln_b;
    k::qrows(rows(k::EPI_LOGITS, lnf, S, NV, w.q_te, nullptr, b.logits, NV, nullptr, nullptr, nullptr, 0.f, nullptr), nullptr, w.q_te, s);
    k::filter_argmax(b.logits, ctx.state->dev.ban_dev, stp, (k::SampleOut *) b.sample_dev, b.filter_scratch, s, (k::SampleOut *) b.sample_host, nb);
}

} // namespace wmi
