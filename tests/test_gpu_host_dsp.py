"""-m gpu: the host-adjacent DSP kernels against oracle/host_dsp.c, the C restatement of the host's C++
(src/speech_to_text.cpp:16-51 resample + stereo -> mono, :53-104 high-pass + energy VAD) — bit-exact: the filter recurrence and
the running f32 energy sums are evaluated in sample order by one lane, every resampled frame by one thread in the converter's
tap order with double accumulators.  The VAD half of that oracle is pinned to the reference's compiled W/examples/common.cpp,
the resampler half to libsamplerate's own test programs (tests/test_oracle_host_dsp.py)."""
import ctypes as C
import pathlib
import struct

import numpy as np
import pytest

from godot_whisper_amd import abi, host, synth

pytestmark = pytest.mark.gpu

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def dsp():
    so = ROOT / "oracle" / "liboracle_dsp.so"
    assert so.exists(), "oracle/liboracle_dsp.so not built (python __graft_entry__.py build)"
    lib = C.CDLL(str(so))
    lib.oracle_resample_audio_buffer.restype = C.c_uint32
    lib.oracle_resample_audio_buffer.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.oracle_vad_simple.restype = C.c_int
    lib.oracle_vad_simple.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]
    lib.oracle_downmix_stereo.restype = None
    lib.oracle_downmix_stereo.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
    return lib


def _table(name):
    raw = (ROOT / "godot-whisper_amd" / "csrc" / "data" / name).read_bytes()
    inc, cnt = struct.unpack("<ii", raw[:8])
    return inc, np.frombuffer(raw[8:], "<f4", cnt).copy()


TABLES = {2: _table("sinc_fastest.bin"), 1: _table("sinc_medium.bin")}


def oracle_resample(dsp, x, src_rate, converter):
    inc, tab = TABLES[converter]
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(max(int(x.size * 16000.0 / src_rate) + 8, x.size + 8), np.float32)
    got = dsp.oracle_resample_audio_buffer(x.ctypes.data, x.size, src_rate, 16000, tab.ctypes.data, tab.size, inc, out.ctypes.data)
    return out[:got]


@pytest.fixture(scope="module")
def node(product_lib):
    n = host.SpeechToText(product_lib); n.set_language_model(synth.make_model("micro.en", seed=1))
    yield n
    n.close()


def test_downmix_is_bit_exact(product_lib, node):
    rng = np.random.default_rng(5)
    fr = rng.uniform(-1, 1, size=(48000 * 2 + 37, 2)).astype(np.float32)
    fr[0] = (1e-38, 1e-38); fr[1] = (3.0e38, 3.0e38)                       # denormal halves, overflow of the float add
    out = np.zeros(fr.shape[0], np.float32)
    assert product_lib.wmi_downmix_stereo(node.ctx, fr.ctypes.data_as(C.c_void_p), fr.shape[0], 0, out.ctypes.data_as(C.c_void_p)) == 0
    with np.errstate(over="ignore"):
        want = ((fr[:, 0] + fr[:, 1]).astype(np.float64) / 2.0).astype(np.float32)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert product_lib.wmi_downmix_stereo(node.ctx, fr.ctypes.data_as(C.c_void_p), 0, 0, out.ctypes.data_as(C.c_void_p)) == 0


@pytest.mark.parametrize("case", ["speech", "gated", "silence", "tiny_noise", "quiet_tail", "no_filter", "short"])
def test_vad_equals_the_host_arithmetic(product_lib, node, dsp, case):
    sr = 16000
    pcm = synth.make_pcm(5.0, seed=11, gate=(case == "gated"))
    thold, freq = 2.0, 200.0
    if case == "silence":
        pcm[:] = 0.0
    elif case == "tiny_noise":
        pcm = (np.random.default_rng(2).standard_normal(5 * sr) * 2e-5).astype(np.float32)
    elif case == "quiet_tail":
        pcm = (np.random.default_rng(3).standard_normal(5 * sr) * 3e-4).astype(np.float32); pcm[-sr:] *= 0.01
    elif case == "no_filter":
        freq = 0.0; pcm = (pcm * 1e-3).astype(np.float32)
    elif case == "short":
        pcm = pcm[: 3 * sr - 1]
    en = np.zeros(2, np.float32)
    got = product_lib.wmi_vad(node.ctx, pcm.ctypes.data_as(C.c_void_p), int(pcm.size), 0, thold, freq, en.ctypes.data_as(C.c_void_p))
    assert got >= 0
    if pcm.size < 3 * sr:
        assert got == 0
        return
    win = np.array(pcm[-3 * sr:], np.float32)
    en_o = np.zeros(2, np.float32)
    want = dsp.oracle_vad_simple(win.ctypes.data, win.size, sr, 500, thold, freq, 0, en_o.ctypes.data)       # the host's form
    assert bool(got) == bool(want), (case, got, want, en, en_o)
    assert en.tobytes() == en_o.tobytes(), (case, en, en_o)
    # and the Python mirror of the host node keeps agreeing with both
    assert bool(host.vad_simple(np.array(pcm[-3 * sr:], np.float32), sr, 500, thold, freq)) == bool(want)


def test_vad_on_device_resident_samples(product_lib, node):
    hip = C.CDLL("libamdhip64.so")
    pcm = (np.random.default_rng(9).standard_normal(4 * 16000) * 2e-5).astype(np.float32)
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), C.c_size_t(pcm.nbytes)) == 0
    try:
        assert hip.hipMemcpy(d, pcm.ctypes.data_as(C.c_void_p), C.c_size_t(pcm.nbytes), 1) == 0          # hipMemcpyHostToDevice
        a = product_lib.wmi_vad(node.ctx, d, int(pcm.size), 1, 2.0, 200.0, None)
        b = product_lib.wmi_vad(node.ctx, pcm.ctypes.data_as(C.c_void_p), int(pcm.size), 0, 2.0, 200.0, None)
        assert a == b and a in (0, 1)
        # down-mix on device-resident frames
        fr = np.random.default_rng(1).uniform(-1, 1, size=(1000, 2)).astype(np.float32)
        din, dout = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(din), C.c_size_t(fr.nbytes)) == 0 and hip.hipMalloc(C.byref(dout), C.c_size_t(4000)) == 0
        assert hip.hipMemcpy(din, fr.ctypes.data_as(C.c_void_p), C.c_size_t(fr.nbytes), 1) == 0
        assert product_lib.wmi_downmix_stereo(node.ctx, din, 1000, 1, dout) == 0
        out = np.zeros(1000, np.float32)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), dout, C.c_size_t(4000), 2) == 0
        assert np.array_equal(out, ((fr[:, 0] + fr[:, 1]).astype(np.float64) / 2.0).astype(np.float32))
        hip.hipFree(din); hip.hipFree(dout)
    finally:
        hip.hipFree(d)


# ------------------------------------------------------------------------------------------------ resampler

def _mic(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = 0.4 * np.sin(2 * np.pi * 0.013 * t + 0.2) + 0.2 * np.sin(2 * np.pi * 0.11 * t) + 0.05 * rng.standard_normal(n)
    return x.astype(np.float32)


def _resample(lib, ctx, x, src_rate, converter):
    x = np.ascontiguousarray(x, np.float32)
    cap = max(int(x.size * 16000.0 / src_rate) + 8, x.size + 8)
    out = np.full(cap, np.nan, np.float32)
    got = lib.wmi_resample(ctx, x.ctypes.data_as(C.c_void_p), int(x.size), src_rate, 16000, converter, 0, out.ctypes.data_as(C.c_void_p), cap)
    return got, out


@pytest.mark.parametrize("src_rate", [48000, 44100, 32000, 22050, 96000, 8000, 11025, 47999])
@pytest.mark.parametrize("converter", [2, 1])
def test_resampler_equals_the_sequential_converter_bit_for_bit(product_lib, node, dsp, src_rate, converter):
    lens = [1, 57, 1000, 14670, 44100, 132301] if converter == 2 else [1000, 44100]
    for i, n in enumerate(lens):
        x = _mic(n, seed=100 + i)
        want = oracle_resample(dsp, x, src_rate, converter)
        got, out = _resample(product_lib, node.ctx, x, src_rate, converter)
        assert got == want.size, (src_rate, n, got, want.size)
        assert out[:got].tobytes() == want.tobytes(), (src_rate, converter, n, float(np.max(np.abs(out[:got] - want))))
        assert np.all(np.isnan(out[got:]))                                       # nothing written past the frames reported


def test_resampler_thirty_seconds_of_capture_frames(product_lib, node, dsp):
    """The streaming node's own use: 30 s of 44.1 kHz stereo capture frames -> mono -> 16 kHz (capture_stream_to_text.gd:76)."""
    n = 44100 * 30
    fr = np.stack([_mic(n, 1), _mic(n, 2)], axis=1)
    mono = np.zeros(n, np.float32)
    dsp.oracle_downmix_stereo(n, fr.ctypes.data, mono.ctypes.data)
    want = oracle_resample(dsp, mono, 44100, 2)
    got = node.resample(fr, host.SpeechToText.SRC_SINC_FASTEST, mix_rate=44100)
    # reference behaviour, reproduced: the converter is asked for int(n * (16000.0 / 44100.0)) = 480 000 frames but stops one early — its
    # termination test `b_current + input_index + 1 / ratio + 1e-20 > b_real_end` (src_sinc.c:389-393) fires for the last frame because
    # 1 / fl(16000 / 44100) = 2.7562500000000001 makes 480 000 steps end at 1 323 000.0000000002 > 1 323 000; the node expects
    # 1 323 000 * 16000 / 44100 = 480 000 (:356) and prints "size differ" (:368-370).
    assert got.size == want.size == 479999
    assert got.tobytes() == want.tobytes()
    assert node.last_resample_warning == "size differ exp: 480000 res: 479999"
    # 48 kHz has no such rounding: 30 s -> exactly 480 000 frames, no warning
    fr48 = np.stack([_mic(48000 * 30, 3), _mic(48000 * 30, 4)], axis=1)
    del node.last_resample_warning
    assert node.resample(fr48, host.SpeechToText.SRC_SINC_FASTEST, mix_rate=48000).size == 480000
    assert not hasattr(node, "last_resample_warning")


def test_resampler_edges(product_lib, node, dsp):
    x = _mic(4800, 5)
    out = np.zeros(4800, np.float32)
    # equal rates copy (src/speech_to_text.cpp:38-42)
    assert product_lib.wmi_resample(node.ctx, x.ctypes.data_as(C.c_void_p), 4800, 16000, 16000, 2, 0, out.ctypes.data_as(C.c_void_p), 4800) == 4800
    assert out.tobytes() == x.tobytes()
    # empty input, capacity too small, best-quality table absent, ratio out of libsamplerate's range (the host gets 0 frames)
    assert product_lib.wmi_resample(node.ctx, x.ctypes.data_as(C.c_void_p), 0, 48000, 16000, 2, 0, out.ctypes.data_as(C.c_void_p), 4800) == 0
    assert product_lib.wmi_resample(node.ctx, x.ctypes.data_as(C.c_void_p), 4800, 48000, 16000, 2, 0, out.ctypes.data_as(C.c_void_p), 10) == -4
    assert product_lib.wmi_resample(node.ctx, x.ctypes.data_as(C.c_void_p), 4800, 48000, 16000, 0, 0, out.ctypes.data_as(C.c_void_p), 4800) == -10
    assert product_lib.wmi_resample(node.ctx, x.ctypes.data_as(C.c_void_p), 4800, 16000 * 300, 16000, 2, 0, out.ctypes.data_as(C.c_void_p), 4800) == 0
    # upsampling (8 kHz telephone audio)
    want = oracle_resample(dsp, x, 8000, 2)
    got, o = _resample(product_lib, node.ctx, x, 8000, 2)
    assert got == want.size == 9600 and o[:got].tobytes() == want.tobytes()


def test_resampler_on_device_pointers_feed_the_transcription(product_lib, node, dsp):
    """Capture frames never touch the CPU: down-mix, resample and VAD on device pointers; the 16 kHz buffer equals the host path."""
    hip = C.CDLL("libamdhip64.so")
    n = 48000 * 4
    fr = np.stack([_mic(n, 7), _mic(n, 8)], axis=1)
    d_fr, d_mono, d_16k = C.c_void_p(), C.c_void_p(), C.c_void_p()
    n16 = n // 3
    assert hip.hipMalloc(C.byref(d_fr), C.c_size_t(fr.nbytes)) == 0 and hip.hipMalloc(C.byref(d_mono), C.c_size_t(4 * n)) == 0
    assert hip.hipMalloc(C.byref(d_16k), C.c_size_t(4 * n16)) == 0
    try:
        assert hip.hipMemcpy(d_fr, fr.ctypes.data_as(C.c_void_p), C.c_size_t(fr.nbytes), 1) == 0
        assert product_lib.wmi_downmix_stereo(node.ctx, d_fr, n, 1, d_mono) == 0
        assert product_lib.wmi_resample(node.ctx, d_mono, n, 48000, 16000, 2, 1, d_16k, n16) == n16
        out = np.zeros(n16, np.float32)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), d_16k, C.c_size_t(4 * n16), 2) == 0
        mono = np.zeros(n, np.float32)
        dsp.oracle_downmix_stereo(n, fr.ctypes.data, mono.ctypes.data)
        assert out.tobytes() == oracle_resample(dsp, mono, 48000, 2).tobytes()
        assert product_lib.wmi_vad(node.ctx, d_16k, n16, 1, 2.0, 200.0, None) in (0, 1)
    finally:
        hip.hipFree(d_fr); hip.hipFree(d_mono); hip.hipFree(d_16k)


# ------------------------------------------------------------------------------------------------ token timestamps, envelope side on the device
def _ts_reference(en, s0, s1, hw=2000):
    """The loops of W/whisper.cpp:6500-6590 for one token on a host envelope (numpy.cumsum in float32 is the sequential left-to-right sum)."""
    n = en.size
    a0, a1 = max(s0 - hw, 0), min(s1 + hw, n)
    total = np.float32(0.0) if a1 <= a0 else np.cumsum(en[a0:a1], dtype=np.float32)[-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        th = np.float32(0.5 * np.float64(total) / np.float64(a1 - a0))
    def walk(k, bound, d, above):
        cont = (lambda x: x > th) if above else (lambda x: x < th)
        while k != bound and (d > 0) == (k < bound) and cont(en[k]):
            k += d
        return k
    return total, th, [int(en[s0] > th), int(en[s1] > th), walk(s0, 0, -1, True), walk(s0, s1, +1, False), walk(s1, n - 1, +1, True), walk(s1, 0, -1, False)]


@pytest.mark.parametrize("kind", ["noise", "ties", "steps", "crossing", "silence", "tiny"])
def test_token_timestamp_sums_and_walks_on_the_device_equal_the_sequential_loops(product_lib, kind):
    """csrc/k_mel.hip k_ts_refine (the opt-in device form of token_level_timestamps' envelope side, WMI_TS_DEVICE=1): per token the
    sequential f32 window sum — taken as order-free integer sums while the running sum stays in one binade, the plain way across ties
    and binade crossings —, the threshold, and the four walks with their block skips.  Every value must equal the plain loops on
    envelopes built to hit those cases: noise, values that are exact half-ulps of the running sum, long constant runs (whole blocks
    skipped on the extrema), sums crossing many binades, all-zero stretches, denormal-sized values."""
    rng = np.random.default_rng({"noise": 1, "ties": 2, "steps": 3, "crossing": 4, "silence": 5, "tiny": 6}[kind])
    n = 200_000
    if kind == "noise":
        en = np.abs(rng.standard_normal(n)).astype(np.float32) * 0.05
    elif kind == "ties":                                   # multiples of 2^-k: sums hit exact ties again and again
        en = (rng.integers(0, 8, n).astype(np.float32) * np.float32(2.0 ** -9)) + np.float32(2.0 ** -13) * rng.integers(0, 2, n).astype(np.float32)
    elif kind == "steps":
        en = np.repeat(np.abs(rng.standard_normal(n // 1000)).astype(np.float32), 1000) * 0.1
    elif kind == "crossing":
        en = (np.float32(1e-6) * np.exp(np.linspace(0, 14, n)).astype(np.float32) * (1 + 0.1 * rng.random(n).astype(np.float32))).astype(np.float32)
    elif kind == "silence":
        en = np.zeros(n, np.float32); en[50_000:50_300] = 0.2; en[120_000:150_000] = np.abs(rng.standard_normal(30_000)).astype(np.float32) * 0.01
    else:
        en = (rng.random(n).astype(np.float32) * np.float32(1e-38)).astype(np.float32)
    toks = []
    for _ in range(24):
        a = int(rng.integers(0, n - 2)); b = int(min(n - 1, a + rng.integers(1, 60_000)))
        toks.append((a, b))
    toks += [(0, n - 1), (n - 1, n - 1), (0, 0), (255, 256), (256, 511), (70_000, 69_000)]       # whole signal, degenerate and reversed tokens
    s0s1 = np.array(toks, np.int32)
    node = host.SpeechToText(product_lib); node.set_language_model(synth.make_model("micro.en", seed=1234))
    try:
        sums = np.zeros(len(toks), np.float32); th = np.zeros(len(toks), np.float32); walks = np.zeros((len(toks), 6), np.int32)
        rc = product_lib.wmi_selftest_ts_refine(node.ctx, en.ctypes.data, n, s0s1.ctypes.data, len(toks), sums.ctypes.data, th.ctypes.data, walks.ctypes.data)
        assert rc == 0
    finally:
        node.close()
    for t, (a, b) in enumerate(toks):
        want_sum, want_th, want_walks = _ts_reference(en, a, b)
        assert sums[t].tobytes() == np.float32(want_sum).tobytes(), (kind, t, a, b, sums[t], want_sum)
        assert th[t].tobytes() == np.float32(want_th).tobytes() or (np.isnan(th[t]) and np.isnan(want_th)), (kind, t, th[t], want_th)
        assert walks[t].tolist() == want_walks, (kind, t, a, b, walks[t].tolist(), want_walks)


# ------------------------------------------------------------------------------------------------ the |x| envelope itself
def _envelope_reference(x, hw=32):
    """get_signal_energy (W/whisper.cpp:6350-6366) step by step: `sum += fabs(signal[i + j])` with a float sum and the double fabs is
    (float) ((double) sum + |x|) per in-range j in order; then sum / (2 hw + 1) as a float division."""
    n = x.size
    ax = np.abs(x.astype(np.float64))
    acc = np.zeros(n, np.float32)
    idx = np.arange(n)
    for j in range(-hw, hw + 1):
        k = idx + j
        ok = (k >= 0) & (k < n)
        step = (acc.astype(np.float64) + ax[np.clip(k, 0, n - 1)]).astype(np.float32)
        acc = np.where(ok, step, acc)
    return (acc / np.float32(2 * hw + 1)).astype(np.float32)


@pytest.mark.parametrize("kind", ["denormal", "mixed", "speech"])
def test_envelope_of_denormal_samples(product_lib, node, kind):
    """csrc/k_mel.hip k_signal_energy runs the reference's 65-step chain as plain f32 additions — equal to its double-add-then-float only
    while the device keeps f32 denormals (Makefile: -fno-gpu-flush-denormals-to-zero).  Samples whose partial sums are denormal, cross
    into the normal range, or sit beside ordinary speech-sized values: every envelope value must have the reference's bits."""
    rng = np.random.default_rng({"denormal": 11, "mixed": 12, "speech": 13}[kind])
    n = 16000 * 3
    if kind == "denormal":                       # |x| in the f32 denormal range: 65 of them sum to at most ~6e-39 < FLT_MIN
        x = (rng.integers(1, 60000, n).astype(np.float64) * 1.4e-45).astype(np.float32) * rng.choice([-1.0, 1.0], n).astype(np.float32)
    elif kind == "mixed":                        # denormals, values around FLT_MIN, zeros and ordinary samples in runs
        x = (rng.integers(0, 1 << 23, n).astype(np.float64) * 1.4e-45).astype(np.float32)
        x[n // 3: n // 3 + 4000] = (rng.standard_normal(4000) * 1.2e-38).astype(np.float32)
        x[n // 2: n // 2 + 4000] = 0.0
        x[2 * n // 3: 2 * n // 3 + 6000] = (0.1 * rng.standard_normal(6000)).astype(np.float32)
    else:
        x = synth.make_pcm(3.0, seed=99)[:n].astype(np.float32)
    assert np.any((np.abs(x) > 0) & (np.abs(x) < 1.17549435e-38)) or kind == "speech"
    p = node.full_params("", 0)                  # host parameter set: token_timestamps on -> the envelope kernel runs
    assert p.token_timestamps
    assert product_lib.whisper_full(node.ctx, p, x.ctypes.data_as(C.POINTER(C.c_float)), int(x.size)) == 0
    got = np.zeros(n, np.float32)
    assert product_lib.wmi_get_tensor(node.ctx, b"energy", got.ctypes.data_as(C.POINTER(C.c_float)), n) == n
    want = _envelope_reference(x)
    assert got.tobytes() == want.tobytes(), (kind, int(np.argmax(got.view(np.uint32) != want.view(np.uint32))))
