"""Deterministic synthetic inputs for the hot path: ggml Whisper model files and 16 kHz PCM.

The file format written here is the one the reference loader parses
(W/whisper.cpp:1112-1633, writer W/models/convert-pt-to-ggml.py:268-339):

    u32 magic 0x67676d6c | 11 x i32 hparams | i32 n_mel, i32 n_fft, f32 filters[n_mel*n_fft]
    i32 n_vocab_in_file, n x {u32 len, bytes} | tensors: {i32 n_dims, i32 name_len, i32 ttype,
    i32 ne[n_dims], name, data}

Real weights are not available offline, so parity and throughput are measured on seeded random
weights of the exact tiny.en / base.en / small / large-v3 shapes.  `logit_scale` (= token
embedding row norm x final-LN gain) makes the output distribution peaked (top-1 / top-2 logit
gaps of order 1) — with plain N(0,1/sqrt(S)) rows the soft-max is flat and greedy arg-max would
flip on 1e-4 noise (SURVEY §7 "hard parts").  The gain is split between d_te and decoder.ln so
the tied embedding does not make the model echo its input token.
"""
from __future__ import annotations

import gzip
import pathlib
import struct

import numpy as np

GOLDEN = pathlib.Path(__file__).resolve().parent.parent / "tests" / "golden"

GGML_MAGIC = 0x67676D6C
GGML_TYPE_F32, GGML_TYPE_F16 = 0, 1

# name: (n_vocab, n_audio_ctx, S, H, L_audio, n_text_ctx, S_text, H_text, L_text, n_mels)
SHAPES = {
    "tiny.en": (51864, 1500, 384, 6, 4, 448, 384, 6, 4, 80),
    "tiny": (51865, 1500, 384, 6, 4, 448, 384, 6, 4, 80),
    "base.en": (51864, 1500, 512, 8, 6, 448, 512, 8, 6, 80),
    "base": (51865, 1500, 512, 8, 6, 448, 512, 8, 6, 80),
    "small": (51865, 1500, 768, 12, 12, 448, 768, 12, 12, 80),
    "medium": (51865, 1500, 1024, 16, 24, 448, 1024, 16, 24, 80),
    "large-v3": (51866, 1500, 1280, 20, 32, 448, 1280, 20, 32, 128),
    # not a Whisper release: a 2-layer, 128-wide model for fast CPU/GPU parity tests
    # (3 text layers: n_text_layer == 2 would trip the "distilled" rule, W/whisper.cpp:5119-5125)
    "micro.en": (51864, 1500, 128, 2, 2, 448, 128, 2, 3, 80),
    "micro": (51865, 1500, 128, 2, 2, 448, 128, 2, 3, 80),
    # large-v3's widths (1280 state, 20 heads, 128 mel bins, 51866 tokens / 100 languages) on 2 + 3 layers: exercises
    # every kernel at the widest shape of BASELINE configs[4] at a size the CPU checker finishes in seconds
    "v3-slice": (51866, 1500, 1280, 20, 2, 448, 1280, 20, 3, 128),
    # medium's widths (1024 state, 16 heads) on 2 + 4 layers: the widest shape the one-launch MLP of the one-row step covers (an even layer count)
    "medium-slice": (51865, 1500, 1024, 16, 2, 448, 1024, 16, 4, 80),
}


def mel_filters(n_mels: int) -> np.ndarray:
    """80-bin bank: the reference's own data (fixture).  Other sizes: Slaney-style bank
    (what librosa.filters.mel(sr=16000, n_fft=400, n_mels=n) computes), checked against the
    80-bin fixture in tests/test_synth.py."""
    if n_mels == 80:
        f = np.fromfile(GOLDEN / "mel_filters_80.f32", dtype="<f4")
        return f.reshape(80, 201)
    return slaney_mel_filters(n_mels)


def slaney_mel_filters(n_mels: int, sr: int = 16000, n_fft: int = 400) -> np.ndarray:
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        mel = f / (200.0 / 3)
        log_t = f >= 1000.0
        mel = np.where(log_t, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), mel)
        return mel

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f = m * (200.0 / 3)
        log_t = m >= 15.0
        return np.where(log_t, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)

    fftfreqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_pts = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_pts)
    ramps = mel_pts[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_pts[2 : n_mels + 2] - mel_pts[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def vocab_blob(multilingual: bool) -> bytes:
    name = "vocab_multi.bin.gz" if multilingual else "vocab_en.bin.gz"
    return gzip.decompress((GOLDEN / name).read_bytes())


def tensor_specs(hp):
    """(name, shape in ggml ne order [fastest first], is_matrix) for every tensor the loader
    expects (W/whisper.cpp:1298-1513)."""
    n_vocab, n_actx, S, H, La, n_tctx, St, Ht, Lt, n_mels = hp
    out = []
    out.append(("encoder.positional_embedding", (S, n_actx), "pe"))
    out.append(("encoder.conv1.weight", (3, n_mels, S), "conv"))
    out.append(("encoder.conv1.bias", (1, S), "bias"))
    out.append(("encoder.conv2.weight", (3, S, S), "conv"))
    out.append(("encoder.conv2.bias", (1, S), "bias"))
    out.append(("encoder.ln_post.weight", (S,), "ln_w"))
    out.append(("encoder.ln_post.bias", (S,), "ln_b"))
    for i in range(La):
        p = f"encoder.blocks.{i}."
        out += [
            (p + "mlp_ln.weight", (S,), "ln_w"), (p + "mlp_ln.bias", (S,), "ln_b"),
            (p + "mlp.0.weight", (S, 4 * S), "mat"), (p + "mlp.0.bias", (4 * S,), "bias"),
            (p + "mlp.2.weight", (4 * S, S), "mat"), (p + "mlp.2.bias", (S,), "bias"),
            (p + "attn_ln.weight", (S,), "ln_w"), (p + "attn_ln.bias", (S,), "ln_b"),
            (p + "attn.query.weight", (S, S), "mat"), (p + "attn.query.bias", (S,), "bias"),
            (p + "attn.key.weight", (S, S), "mat"),
            (p + "attn.value.weight", (S, S), "mat"), (p + "attn.value.bias", (S,), "bias"),
            (p + "attn.out.weight", (S, S), "mat"), (p + "attn.out.bias", (S,), "bias"),
        ]
    out.append(("decoder.positional_embedding", (St, n_tctx), "pe"))
    out.append(("decoder.token_embedding.weight", (St, n_vocab), "te"))
    out.append(("decoder.ln.weight", (St,), "ln_w"))
    out.append(("decoder.ln.bias", (St,), "ln_b"))
    for i in range(Lt):
        p = f"decoder.blocks.{i}."
        out += [
            (p + "mlp_ln.weight", (St,), "ln_w"), (p + "mlp_ln.bias", (St,), "ln_b"),
            (p + "mlp.0.weight", (St, 4 * St), "mat"), (p + "mlp.0.bias", (4 * St,), "bias"),
            (p + "mlp.2.weight", (4 * St, St), "mat"), (p + "mlp.2.bias", (St,), "bias"),
            (p + "attn_ln.weight", (St,), "ln_w"), (p + "attn_ln.bias", (St,), "ln_b"),
            (p + "attn.query.weight", (St, St), "mat"), (p + "attn.query.bias", (St,), "bias"),
            (p + "attn.key.weight", (St, St), "mat"),
            (p + "attn.value.weight", (St, St), "mat"), (p + "attn.value.bias", (St,), "bias"),
            (p + "attn.out.weight", (St, St), "mat"), (p + "attn.out.bias", (St,), "bias"),
            (p + "cross_attn_ln.weight", (St,), "ln_w"), (p + "cross_attn_ln.bias", (St,), "ln_b"),
            (p + "cross_attn.query.weight", (St, St), "mat"), (p + "cross_attn.query.bias", (St,), "bias"),
            (p + "cross_attn.key.weight", (St, St), "mat"),
            (p + "cross_attn.value.weight", (St, St), "mat"), (p + "cross_attn.value.bias", (St,), "bias"),
            (p + "cross_attn.out.weight", (St, St), "mat"), (p + "cross_attn.out.bias", (St,), "bias"),
        ]
    return out


def make_model(shape: str | tuple = "base.en", seed: int = 1234, logit_scale: float = 6.0,
               te_scale: float = 3.0, out_scale: float = 1.0, pe_scale: float = 1.0,
               f32: bool = False, n_audio_ctx: int | None = None) -> bytes:
    """Return the bytes of a synthetic ggml Whisper model."""
    hp = list(SHAPES[shape] if isinstance(shape, str) else shape)
    if n_audio_ctx is not None:
        hp[1] = n_audio_ctx
    n_vocab, n_actx, S, H, La, n_tctx, St, Ht, Lt, n_mels = hp
    rng = np.random.default_rng(seed)
    parts = [struct.pack("<I", GGML_MAGIC), struct.pack("<11i", *hp, 0 if f32 else 1)]
    filt = mel_filters(n_mels)
    parts.append(struct.pack("<2i", n_mels, 201))
    parts.append(np.ascontiguousarray(filt, dtype="<f4").tobytes())
    parts.append(vocab_blob(n_vocab >= 51865))
    wdt = np.float32 if f32 else np.float16
    for name, ne, kind in tensor_specs(hp):
        n = int(np.prod(ne))
        g = rng.standard_normal(n, dtype=np.float32)
        if kind == "mat":
            # block output projections are widened so the residual stream is not dominated by
            # the (tied) token embedding — otherwise the model just repeats its input token
            k = out_scale if (name.endswith("out.weight") or name.endswith("mlp.2.weight")) else 1.0
            data = (g * (k / np.sqrt(ne[0]))).astype(wdt)
        elif kind == "conv":
            data = (g / np.sqrt(ne[0] * ne[1])).astype(wdt)
        elif kind == "te":
            data = (g * (te_scale / np.sqrt(ne[0]))).astype(wdt)
        elif kind == "pe":
            data = (pe_scale * g).astype(np.float32)
        elif kind == "ln_w":
            # the final decoder LN carries the rest of the logit gain: logits = d_te . (gamma*xhat + beta)
            gain = logit_scale / te_scale if name == "decoder.ln.weight" else 1.0
            data = (gain * (1.0 + 0.1 * g)).astype(np.float32)
        elif kind == "ln_b":
            data = (0.02 * g).astype(np.float32)
        else:  # bias
            data = (0.02 * g).astype(np.float32)
        ttype = GGML_TYPE_F32 if data.dtype == np.float32 else GGML_TYPE_F16
        nm = name.encode()
        parts.append(struct.pack("<3i", len(ne), len(nm), ttype))
        parts.append(struct.pack(f"<{len(ne)}i", *ne))
        parts.append(nm)
        parts.append(data.tobytes())
    return b"".join(parts)


def make_pcm(seconds: float = 30.0, seed: int = 1234, gate: bool = False) -> np.ndarray:
    """Seeded synthetic 16 kHz mono f32 PCM: 32 sinusoids in 80..4000 Hz
    plus white noise, clipped to [-1, 1].  `gate` = 2 s on / 1 s off (exercises the VAD)."""
    n = int(round(seconds * 16000))
    rng = np.random.default_rng(seed)
    f = rng.uniform(80.0, 4000.0, size=32)
    ph = rng.uniform(0.0, 2 * np.pi, size=32)
    t = np.arange(n, dtype=np.float64) / 16000.0
    x = np.zeros(n, dtype=np.float64)
    for fk, pk in zip(f, ph):
        x += 0.02 * np.sin(2 * np.pi * fk * t + pk)
    x += 0.01 * rng.standard_normal(n)
    if gate:
        x *= ((np.arange(n) // 16000) % 3 != 2)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def read_wav_mono16(path) -> np.ndarray:
    """Minimal RIFF reader for 16-bit PCM WAV -> float32 (s16/32768, the conversion the
    AudioStreamToText node applies, addon/audio_stream_to_text.gd:44-46)."""
    b = pathlib.Path(path).read_bytes()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    off, fmt, data = 12, None, None
    while off + 8 <= len(b):
        cid, sz = b[off:off + 4], struct.unpack_from("<I", b, off + 4)[0]
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", b, off + 8)
        elif cid == b"data":
            data = b[off + 8: off + 8 + sz]
        off += 8 + sz + (sz & 1)
    assert fmt and data is not None and fmt[0] == 1 and fmt[5] == 16
    x = np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0
    if fmt[1] > 1:
        x = x.reshape(-1, fmt[1]).mean(axis=1)
    return x


# ------------------------------------------------------------------------------------------------
# ggml block quantisation (W/ggml-quants.c quantize_row_q*_reference, block layouts W/ggml-quants.h:10-47),
# applied the way the reference's `quantize` tool does (W/examples/common-ggml.cpp:38-215): every 2-D tensor
# except the conv biases and the two positional embeddings; header ftype = 2000 + ftype.
# tests/test_synth_and_shard.py checks the output byte for byte against the reference tool in the build container.
QTYPES = {"q4_0": (2, 2), "q4_1": (3, 3), "q5_0": (6, 8), "q5_1": (7, 9), "q8_0": (8, 7)}   # name: (ggml_type, ftype)
_SKIP = ("encoder.conv1.bias", "encoder.conv2.bias", "encoder.positional_embedding", "decoder.positional_embedding")


def _f16_bytes(x32: np.ndarray) -> np.ndarray:
    return x32.astype(np.float16).view(np.uint8).reshape(-1, 2)


def quantize_blocks(x: np.ndarray, qtype: str) -> bytes:
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 32)
    nb = x.shape[0]
    one = np.float32(1.0)
    if qtype == "q8_0":
        amax = np.abs(x).max(axis=1)
        d = (amax / np.float32(127.0)).astype(np.float32)
        idv = np.where(d != 0, one / np.where(d != 0, d, one), np.float32(0)).astype(np.float32)
        q = np.round((x * idv[:, None]).astype(np.float32)).astype(np.int8)       # roundf: half away from zero
        x0 = (x * idv[:, None]).astype(np.float32)
        q = np.where(x0 >= 0, np.floor(x0 + np.float32(0.5)), np.ceil(x0 - np.float32(0.5))).astype(np.int8)
        out = np.empty((nb, 34), np.uint8)
        out[:, 0:2] = _f16_bytes(d); out[:, 2:] = q.view(np.uint8)
        return out.tobytes()
    if qtype in ("q4_0", "q5_0"):
        idx = np.abs(x).argmax(axis=1)
        mx = x[np.arange(nb), idx]
        div = np.float32(-8.0) if qtype == "q4_0" else np.float32(-16.0)
        d = (mx / div).astype(np.float32)
        idv = np.where(d != 0, one / np.where(d != 0, d, one), np.float32(0)).astype(np.float32)
        off = np.float32(8.5) if qtype == "q4_0" else np.float32(16.5)
        lim = 15 if qtype == "q4_0" else 31
        xi = np.minimum(lim, ((x * idv[:, None]).astype(np.float32) + off).astype(np.float32).astype(np.int8).astype(np.int32)).astype(np.uint8)
        dh, mh = _f16_bytes(d), None
    else:  # q4_1, q5_1
        mn = x.min(axis=1); mx = x.max(axis=1)
        lv = np.float32(15.0) if qtype == "q4_1" else np.float32(31.0)
        d = ((mx - mn) / lv).astype(np.float32)
        idv = np.where(d != 0, one / np.where(d != 0, d, one), np.float32(0)).astype(np.float32)
        x0 = (((x - mn[:, None]).astype(np.float32) * idv[:, None]).astype(np.float32) + np.float32(0.5)).astype(np.float32)
        if qtype == "q4_1":
            xi = np.minimum(15, x0.astype(np.int8).astype(np.int32)).astype(np.uint8)
        else:
            xi = x0.astype(np.uint8)
        dh, mh = _f16_bytes(d), _f16_bytes(mn)
    lo, hi = xi[:, :16], xi[:, 16:]
    qs = ((lo & 0x0F) | ((hi & 0x0F) << 4)).astype(np.uint8)
    parts = [dh] + ([mh] if mh is not None else [])
    if qtype in ("q5_0", "q5_1"):
        j = np.arange(16, dtype=np.uint32)
        qh = (((lo.astype(np.uint32) & 0x10) >> 4) << j).sum(axis=1, dtype=np.uint32) | \
             (((hi.astype(np.uint32) & 0x10) >> 4) << (j + 16)).sum(axis=1, dtype=np.uint32)
        parts.append(qh.astype("<u4").view(np.uint8).reshape(-1, 4))
    parts.append(qs)
    return np.concatenate(parts, axis=1).tobytes()


def quantize_model(model: bytes, qtype: str = "q5_1") -> bytes:
    gtype, ftype = QTYPES[qtype]
    b = model
    hp = list(struct.unpack_from("<11i", b, 4))
    off = 4 + 44
    n_mel, n_fft = struct.unpack_from("<2i", b, off)
    off += 8 + 4 * n_mel * n_fft
    (nv,) = struct.unpack_from("<i", b, off)
    off += 4
    for _ in range(nv):
        (ln,) = struct.unpack_from("<I", b, off)
        off += 4 + ln
    hp[10] = 2 * 1000 + ftype                       # GGML_QNT_VERSION = 2
    out = [b[:4], struct.pack("<11i", *hp), b[48:off]]
    while off < len(b):
        nd, nl, tt = struct.unpack_from("<3i", b, off); off += 12
        ne = struct.unpack_from(f"<{nd}i", b, off); off += 4 * nd
        name = b[off:off + nl]; off += nl
        n = int(np.prod(ne))
        nbytes = n * (2 if tt == 1 else 4)
        raw = b[off:off + nbytes]; off += nbytes
        if nd == 2 and name.decode() not in _SKIP:
            x = np.frombuffer(raw, np.float16 if tt == 1 else np.float32).astype(np.float32)
            out += [struct.pack("<3i", nd, nl, gtype), struct.pack(f"<{nd}i", *ne), name, quantize_blocks(x, qtype)]
        else:
            out += [struct.pack("<3i", nd, nl, tt), struct.pack(f"<{nd}i", *ne), name, raw]
    return b"".join(out)
