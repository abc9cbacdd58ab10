"""Host-side mirror of the GDExtension node that owns the boundary
(`SpeechToText`, src/speech_to_text.cpp:106-573) and of the two GDScript call patterns
(`AudioStreamToText.get_text`, addon/audio_stream_to_text.gd:31-62;
`CaptureStreamToText.transcribe_thread`, addon/capture_stream_to_text.gd:65-120).

Godot / godot-cpp / SCons are not in this image, so the node is restated in Python over the
same C ABI calls, in the same order, with the same parameter set.  It works over ANY library
that exports whisper.h (`abi.WHISPER_API`): the product `libwhisper_mi355.so`, or —
in tests only — the compiled reference, which is how the parity tests drive both sides
through one code path.
"""
from __future__ import annotations

import ctypes as C
import math
import re

import numpy as np

from . import abi

# ProjectSettings defaults, src/register_types.cpp:64-69 (spelling is the reference's)
SETTINGS = {
    "audio/input/transcribe/entropy_treshold": 2.8,
    "audio/input/transcribe/freq_treshold": 200.0,
    "audio/input/transcribe/max_tokens": 16,
    "audio/input/transcribe/vad_treshold": 2.0,
    "audio/input/transcribe/use_gpu": True,
    "audio/input/transcribe/speed_up_2x": False,
}


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class SpeechToText:
    """One `whisper_context` per node (src/speech_to_text.h:135); not re-entrant."""

    def __init__(self, lib: C.CDLL, settings: dict | None = None):
        self.lib = lib
        self.ctx = None
        self.language = "en"  # Language enum -> code, src/speech_to_text.cpp:117-324
        self.settings = dict(SETTINGS)
        if settings:
            self.settings.update(settings)
        self._keep = []

    # -- set_language_model / _load_model (src/speech_to_text.cpp:326-346)
    def set_language_model(self, model_bytes: bytes | None):
        self.lib.whisper_free(self.ctx)
        self.ctx = None
        if not model_bytes:
            return
        buf = C.create_string_buffer(model_bytes, len(model_bytes))
        cp = abi.whisper_context_params(bool(self.settings["audio/input/transcribe/use_gpu"]))
        self.ctx = self.lib.whisper_init_from_buffer_with_params(C.cast(buf, C.c_void_p), len(model_bytes), cp)
        del buf  # the loader only borrows the buffer during the call (W/whisper.cpp:3231-3240)

    def close(self):
        self.lib.whisper_free(self.ctx)
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- transcribe (src/speech_to_text.cpp:401-450)
    def full_params(self, initial_prompt: str = "", audio_ctx: int = 0) -> abi.whisper_full_params:
        p = self.lib.whisper_full_default_params(abi.WHISPER_SAMPLING_GREEDY)
        lang = self.language.encode()
        prompt = initial_prompt.encode("utf-8")
        self._keep = [lang, prompt]  # the reference lets this dangle (:413); we keep it alive
        p.language = lang
        p.audio_ctx = int(audio_ctx)
        p.speed_up = bool(self.settings["audio/input/transcribe/speed_up_2x"])
        p.split_on_word = True
        p.token_timestamps = True
        p.suppress_non_speech_tokens = True
        p.single_segment = True
        p.max_tokens = int(self.settings["audio/input/transcribe/max_tokens"])
        p.entropy_thold = float(self.settings["audio/input/transcribe/entropy_treshold"])
        p.initial_prompt = prompt
        if getattr(self, "n_threads", 0):            # tests: the CPU reference behind this mirror (results do not depend on the thread count)
            p.n_threads = int(self.n_threads)
        return p

    def transcribe(self, buffer: np.ndarray, initial_prompt: str = "", audio_ctx: int = 0, params=None) -> list:
        if not self.ctx:
            return []  # ERR_PRINT("Context instance is null")
        buffer = np.ascontiguousarray(buffer, dtype=np.float32)
        p = params if params is not None else self.full_params(initial_prompt, audio_ctx)
        ret = self.lib.whisper_full(self.ctx, p, _fptr(buffer), int(buffer.size))
        self.last_ret = ret
        if ret != 0:
            return []  # ERR_PRINT("Failed to process audio, returned ...")
        return self.collect()

    def collect(self) -> list:
        out, full_text = [], b""
        lib, ctx = self.lib, self.ctx
        for i in range(lib.whisper_full_n_segments(ctx)):
            full_text += lib.whisper_full_get_segment_text(ctx, i)
            for j in range(lib.whisper_full_n_tokens(ctx, i)):
                t = lib.whisper_full_get_token_data(ctx, i, j)
                out.append({
                    "text": lib.whisper_full_get_token_text(ctx, i, j), "id": t.id, "p": t.p, "plog": t.plog,
                    "pt": t.pt, "ptsum": t.ptsum, "t0": t.t0, "t1": t.t1, "tid": t.tid, "vlen": t.vlen,
                })
        out.insert(0, full_text)
        return out

    # -- several independent buffers at once (product library only: include/wmi_device.h wmi_full_batch).  The
    #    reference host has no such call; the semantics are "transcribe() of each buffer, on a fresh context".
    def transcribe_batch(self, buffers: list, initial_prompt: str = "", audio_ctx: int = 0, params=None,
                         device_ptrs: list | None = None) -> list:
        if not self.ctx:
            return []
        p = params if params is not None else self.full_params(initial_prompt, audio_ctx)
        n = len(buffers)
        if device_ptrs is None:
            bufs = [np.ascontiguousarray(b, dtype=np.float32) for b in buffers]
            ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
            lens = (C.c_int * n)(*[int(b.size) for b in bufs])
            on_dev = 0
        else:
            ptrs = (C.c_void_p * n)(*device_ptrs)
            lens = (C.c_int * n)(*[int(b) for b in buffers])          # sample counts
            on_dev = 1
        ret = self.lib.wmi_full_batch(self.ctx, p, ptrs, lens, n, on_dev)
        self.last_ret = ret
        if ret != 0:
            return []
        out = []
        self.last_modes = []
        for c in range(n):
            assert self.lib.wmi_batch_select(self.ctx, c) >= 0
            self.last_modes.append(self.lib.wmi_batch_chunk_mode(self.ctx, c))
            out.append(self.collect())
        return out

    # -- resample (src/speech_to_text.cpp:353-376): stereo capture frames at the mix rate -> mono 16 kHz.  The reference goes through
    #    libsamplerate on the CPU; here both steps run on the device (wmi_downmix_stereo, wmi_resample).
    SRC_SINC_BEST_QUALITY, SRC_SINC_MEDIUM_QUALITY, SRC_SINC_FASTEST = 0, 1, 2        # src/speech_to_text.h:151-155

    def resample(self, buffer_xy: np.ndarray, interpolator_type: int = 2, mix_rate: int = 44100) -> np.ndarray:
        xy = np.ascontiguousarray(buffer_xy, dtype=np.float32).reshape(-1, 2)
        n = int(xy.shape[0])
        expected = n * abi.WHISPER_SAMPLE_RATE // int(mix_rate)                        # :356
        mono = np.empty(n, np.float32)
        if n and self.lib.wmi_downmix_stereo(self.ctx, xy.ctypes.data_as(C.c_void_p), n, 0, mono.ctypes.data_as(C.c_void_p)) != 0:
            return np.zeros(0, np.float32)
        out = np.empty(max(expected, n if mix_rate == abi.WHISPER_SAMPLE_RATE else 0, 1), np.float32)
        got = self.lib.wmi_resample(self.ctx, mono.ctypes.data_as(C.c_void_p), n, int(mix_rate), abi.WHISPER_SAMPLE_RATE,
                                    int(interpolator_type), 0, out.ctypes.data_as(C.c_void_p), int(out.size))
        if got < 0:
            got = 0
        if got != expected:                                                            # :368-370
            self.last_resample_warning = f"size differ exp: {expected} res: {got}"
        return out[:got].copy()

    # -- voice_activity_detection (src/speech_to_text.cpp:53-104, 378-399)
    def voice_activity_detection(self, buffer: np.ndarray) -> bool:
        n_win = abi.WHISPER_SAMPLE_RATE * 3
        if buffer.size < n_win:
            return False
        if getattr(self, "device_vad", False) and hasattr(self.lib, "wmi_vad") and self.ctx:
            # the product's device kernel (include/wmi_device.h): same decision, no per-sample host loop
            b = np.ascontiguousarray(buffer[-n_win:], dtype=np.float32)
            r = self.lib.wmi_vad(self.ctx, b.ctypes.data_as(C.c_void_p), int(b.size), 0,
                                 float(self.settings["audio/input/transcribe/vad_treshold"]),
                                 float(self.settings["audio/input/transcribe/freq_treshold"]), None)
            assert r >= 0, r
            return bool(r)
        pcm = np.array(buffer[-n_win:], dtype=np.float32)
        return vad_simple(pcm, abi.WHISPER_SAMPLE_RATE, 500,
                          float(self.settings["audio/input/transcribe/vad_treshold"]),
                          float(self.settings["audio/input/transcribe/freq_treshold"]))


def high_pass_filter(data: np.ndarray, cutoff: float, sample_rate: float) -> None:
    """src/speech_to_text.cpp:53-66, operation for operation.  The reference filters IN PLACE and reads data[i - 1] after it
    has been overwritten, so its "previous input" is the previous OUTPUT: y = alpha * ((y + x_i) - y).  (Math_PI is a double:
    rc is computed in double and rounded to float.)"""
    rc = np.float32(1.0 / (2.0 * math.pi * float(np.float32(cutoff))))
    dt = np.float32(1.0) / np.float32(sample_rate)
    alpha = np.float32(dt / np.float32(rc + dt))
    y = np.float32(data[0])
    for i in range(1, data.size):
        y = np.float32(alpha * np.float32(np.float32(y + np.float32(data[i])) - np.float32(data[i - 1])))
        data[i] = y


def vad_simple(pcm: np.ndarray, sample_rate: int, last_ms: int, vad_thold: float, freq_thold: float) -> bool:
    n = pcm.size
    n_last = (sample_rate * last_ms) // 1000
    if n_last >= n:
        return False
    if freq_thold > 0.0:
        high_pass_filter(pcm, freq_thold, sample_rate)
    a = np.abs(pcm.astype(np.float32))
    # running f32 sums in sample order, as the reference's loop (np.cumsum accumulates sequentially; np.add.reduce would pair)
    e_all = np.float32(np.cumsum(a, dtype=np.float32)[-1]) / np.float32(n)
    e_last = np.float32(np.cumsum(a[n - n_last:], dtype=np.float32)[-1]) / np.float32(max(n_last, 1))
    vad_simple.last_energies = (float(e_all), float(e_last))
    vad_thold = np.float32(vad_thold)
    # note the host's extra "not both < 1e-4" clause vs upstream vad_simple
    if not (e_all < 0.0001 and e_last < 0.0001) or e_last > vad_thold * e_all:
        return False
    return True


class AudioStreamToText(SpeechToText):
    """addon/audio_stream_to_text.gd: one-shot "transcribe this WAV" node."""

    def get_text(self, pcm: np.ndarray, initial_prompt: str = "") -> str:
        tokens = self.transcribe(pcm, initial_prompt, 0)
        if not tokens:
            return ""
        text = tokens.pop(0).decode("utf-8", errors="replace")
        return remove_special_characters(text)


def remove_special_characters(message: str) -> str:
    # addon/audio_stream_to_text.gd:64-88 — drop [..] and <..> spans and a ". you." hallucination
    for a, b in (("[", "]"), ("<", ">")):
        while True:
            i = message.find(a)
            j = message.find(b)
            if i == -1 or j == -1 or j < i:
                break
            message = message[:i] + message[j + 1:]
    message = re.sub(r"\. you\.$", ".", message)
    return message


class CaptureStreamToText(SpeechToText):
    """addon/capture_stream_to_text.gd: the streaming loop, restated over a pre-recorded 16 kHz
    buffer instead of AudioEffectCapture (resampling stays with libsamplerate in the host and is
    out of scope).  Every `interval` seconds of simulated time the whole accumulated buffer is
    re-transcribed with audio_ctx = total_s*50 + 128 (:84)."""

    def __init__(self, lib, settings=None, transcribe_interval: float = 0.3, minimum_sentence_ms: int = 3000,
                 maximum_sentence_ms: int = 15000, punctuation_characters: str = ".!?;。；？！"):
        super().__init__(lib, settings)
        self.interval = transcribe_interval
        self.min_ms = minimum_sentence_ms
        self.max_ms = maximum_sentence_ms
        self.punct = punctuation_characters

    def stream(self, pcm16k: np.ndarray, max_calls: int | None = None):
        """Yield (is_final, text, n_samples_used, audio_ctx, token dicts) per transcribe call."""
        sr = abi.WHISPER_SAMPLE_RATE
        step = int(round(self.interval * sr))
        start, pos, calls = 0, 0, 0
        last_tokens = -1
        while pos < pcm16k.size:
            pos = min(pos + step, pcm16k.size)
            acc = pcm16k[start:pos]
            total_s = acc.size / sr
            if total_s < 1.0:
                continue
            no_activity = self.voice_activity_detection(acc)
            audio_ctx = min(int(total_s * 50 + 128), 1500)
            tokens = self.transcribe(acc, "", audio_ctx)
            calls += 1
            if not tokens:
                continue
            text = remove_special_characters(tokens.pop(0).decode("utf-8", errors="replace"))
            n_tok = len(tokens)
            call_tokens = list(tokens)
            finish = False
            total_ms = total_s * 1000
            if total_ms > self.min_ms:
                ends_punct = len(text) > 0 and text[-1] in self.punct
                if (ends_punct or no_activity) and abs(n_tok - last_tokens) <= 1:
                    finish = True
                if total_ms > self.max_ms:
                    finish = True
            last_tokens = n_tok
            yield finish, text, acc.size, audio_ctx, call_tokens
            if finish:
                start = max(pos - int(0.2 * sr), 0)  # keep only the last 0.2 s (:111)
                last_tokens = -1
            if max_calls is not None and calls >= max_calls:
                return
