"""CPU-side pins of oracle/host_dsp.c and of the product's host-side resampling plan.

* VAD + high-pass filter: bit-exact against the reference's own compiled W/examples/common.cpp (oracle/_ref/libcommon_ref.so).
* SINC resampler: libsamplerate cannot be compiled here (src_sinc.c:36 includes a missing blob), so the restatement is held to
  the reference's own TEST PROGRAMS restated on top of it — thirdparty/libsamplerate/tests/termination_test.c (init_term_test,
  simple_test) and snr_bw_test.c (snr_test with calc_snr.c's peak analysis) for the two converters whose tables are in the tree —
  and to identities of the algorithm.
* the product's plan (frame counts, output positions; host code of libwhisper_mi355.so, no GPU needed) equals the restatement's
  sequential run for every rate pair and length tried, including lengths around the ring-buffer refill boundaries.
"""
import ctypes as C
import math
import pathlib
import struct

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
DATA = ROOT / "godot-whisper_amd" / "csrc" / "data"


def _table(name):
    raw = (DATA / name).read_bytes()
    inc, cnt = struct.unpack("<ii", raw[:8])
    return inc, np.frombuffer(raw[8:], "<f4", cnt).copy()


TABLES = {2: _table("sinc_fastest.bin"), 1: _table("sinc_medium.bin")}


@pytest.fixture(scope="module")
def dsp():
    so = ROOT / "oracle" / "liboracle_dsp.so"
    if not so.exists():
        pytest.skip("oracle/liboracle_dsp.so not built (python __graft_entry__.py build)")
    lib = C.CDLL(str(so))
    lib.oracle_src_simple_mono.restype = C.c_int
    lib.oracle_src_simple_mono.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long,
                                           C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.oracle_resample_audio_buffer.restype = C.c_uint32
    lib.oracle_resample_audio_buffer.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.oracle_high_pass_filter.restype = None
    lib.oracle_high_pass_filter.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float]
    lib.oracle_vad_simple.restype = C.c_int
    lib.oracle_vad_simple.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]
    return lib


def src_simple(dsp, x, ratio, converter, out_frames):
    inc, tab = TABLES[converter]
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(max(out_frames, 1), np.float32)
    gen, used = C.c_long(0), C.c_long(0)
    err = dsp.oracle_src_simple_mono(x.ctypes.data, x.size, ratio, tab.ctypes.data, tab.size, inc, out.ctypes.data, out_frames,
                                     C.byref(gen), C.byref(used))
    return err, out[:gen.value], used.value


# ------------------------------------------------------------------------------------------------ VAD against the compiled reference

@pytest.fixture(scope="module")
def common_ref():
    so = ROOT / "oracle" / "_ref" / "libcommon_ref.so"
    if not so.exists():
        pytest.skip("oracle/_ref/libcommon_ref.so not built (make -C oracle ref needs /root/reference)")
    lib = C.CDLL(str(so))
    lib.ref_high_pass_filter.restype = None
    lib.ref_high_pass_filter.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
    lib.ref_vad_simple.restype = C.c_int
    lib.ref_vad_simple.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
    return lib


def _speechy(rng, n, loud_tail):
    t = np.arange(n) / 16000.0
    x = 0.1 * np.sin(2 * np.pi * 220 * t) * (rng.random(n) < 0.7) + 0.01 * rng.standard_normal(n)
    x[-8000:] *= loud_tail
    return x.astype(np.float32)


@pytest.mark.parametrize("cutoff", [100.0, 200.0, 1234.5])
def test_high_pass_filter_is_the_compiled_reference_bit_for_bit(dsp, common_ref, cutoff):
    rng = np.random.default_rng(7)
    x = _speechy(rng, 48000, 1.0)
    a, b = x.copy(), x.copy()
    dsp.oracle_high_pass_filter(a.ctypes.data, a.size, cutoff, 16000.0)
    common_ref.ref_high_pass_filter(b.ctypes.data, b.size, cutoff, 16000.0)
    assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("tail,thold,freq", [(1.0, 2.0, 200.0), (0.05, 2.0, 200.0), (0.05, 0.6, 100.0), (3.0, 2.0, 0.0), (0.3, 1.0, 200.0)])
def test_vad_decision_and_filtered_samples_equal_the_compiled_reference(dsp, common_ref, tail, thold, freq):
    rng = np.random.default_rng(11)
    x = _speechy(rng, 48000, tail)
    a, b = x.copy(), x.copy()
    en = np.zeros(2, np.float32)
    got = dsp.oracle_vad_simple(a.ctypes.data, a.size, 16000, 500, thold, freq, 1, en.ctypes.data)     # upstream decision
    want = common_ref.ref_vad_simple(b.ctypes.data, b.size, 16000, 500, thold, freq)
    assert a.tobytes() == b.tobytes()
    assert got == want
    # the host's form differs only by its "both energies below 1e-4" clause (src/speech_to_text.cpp:99): loud input -> never "silent"
    c = x.copy()
    host = dsp.oracle_vad_simple(c.ctypes.data, c.size, 16000, 500, thold, freq, 0, en.ctypes.data)
    assert c.tobytes() == b.tobytes()
    assert host == (1 if (en[0] < 1e-4 and en[1] < 1e-4 and not en[1] > np.float32(thold) * en[0]) else 0)


# ------------------------------------------------------------------------------------------------ libsamplerate's own tests, restated

RATIOS = [0.999900, 1.000100, 0.789012, 1.200000, 0.333333, 3.100000, 0.125000, 8.000000, 0.099900, 9.990000, 0.100000, 10.00000]


@pytest.mark.parametrize("converter", [2, 1])
@pytest.mark.parametrize("ratio", RATIOS)
def test_init_term_test_of_termination_test_c(dsp, converter, ratio):
    """thirdparty/libsamplerate/tests/termination_test.c:96-167 (the program runs it for SRC_SINC_FASTEST)."""
    short = 2048
    if ratio >= 1.0:
        out_len, in_len = short, int(math.floor(short / ratio))
    else:
        in_len, out_len = short, int(math.floor(short * ratio))
    in_len -= 10
    assert out_len <= short
    err, out, used = src_simple(dsp, np.ones(short, np.float32)[:in_len], ratio, converter, short)
    assert err == 0
    terminate = int(math.ceil(1 if ratio >= 1.0 else 1.0 / ratio))
    assert abs(ratio * in_len - out.size) <= terminate, (ratio, in_len, out.size)
    assert abs(used - in_len) <= 1
    assert abs(out[0]) >= 0.1


def test_simple_test_of_termination_test_c(dsp):
    """termination_test.c:64-94: 199 030 frames -> 1 000 must not fail."""
    ilen, olen = 199030, 1000
    err, out, used = src_simple(dsp, np.zeros(ilen, np.float32), (1.0 * olen) / ilen, 2, olen)
    assert err == 0 and out.size <= olen


def _gen_windowed_sines(freqs, n):                       # tests/util.c:21-53
    k = np.arange(n, dtype=np.float64)
    out = np.zeros(n, np.float32)
    amp = 1.0 / len(freqs)
    for f in freqs:
        phase = 0.9 * math.pi / len(freqs)
        out = (out + (amp * np.sin(f * (2 * k) * math.pi + phase))).astype(np.float32)
    return (out * (0.5 - 0.5 * np.cos((2 * k) * math.pi / (n - 1)))).astype(np.float32)


def _calculate_snr(data, expected_peaks):                # tests/calc_snr.c:38-231
    n = data.size
    x = data.astype(np.float64)
    while (n & 0x1F) and n < (1 << 15):
        x = np.append(x, 0.0); n += 1
    spec = np.abs(np.fft.rfft(x))
    mag = np.zeros(n)
    mag[1:n // 2] = spec[1:n // 2]
    mag /= mag.max()
    mag = np.where(mag < 1e-15, -200.0, 20.0 * np.log10(np.maximum(mag, 1e-300)))
    half = n // 2
    m = mag.copy()

    def is_peak(a, k):
        return a[k - 1] < a[k] and a[k] >= a[k + 1]

    def smooth(larger, smaller):
        if smaller[1] < larger[1]:
            for k in range(smaller[1] + 1, larger[1]):
                if m[k] < m[k - 1]: m[k] = 0.999 * m[k - 1]
        else:
            for k in range(smaller[1] - 1, larger[1] - 1, -1):
                if m[k] < m[k + 1]: m[k] = 0.999 * m[k + 1]

    first = None
    for k in range(1, half - 1):
        if is_peak(m, k):
            first = (m[k], k); break
    assert first is not None
    prev = first
    k = prev[1] + 1
    while k < half - 1:
        if is_peak(m, k):
            cur = (m[k], k)
            if cur[0] > prev[0]: smooth(cur, prev)
            else: smooth(prev, cur)
            prev = cur
        k += 1
    peaks = []
    for k in range(1, n - 1):
        if is_peak(m, k):
            peaks.append(m[k])
    peaks = sorted(peaks, reverse=True)[:10]
    assert len(peaks) >= expected_peaks
    snr = peaks[0]
    for p in peaks[1:]:
        if abs(snr - p) > 10.0:
            return abs(p)
    return snr


SNR_CASES = {   # tests/snr_bw_test.c:93-123: (freqs, ratio, pass-band peaks, required SNR dB, output peak)
    2: [((0.01111111111,), 3.0, 1, 100.0, 1.0), ((0.01111111111,), 0.6, 1, 99.0, 1.0), ((0.01111111111,), 0.3, 1, 100.0, 1.0),
        ((0.01111111111,), 1.0, 1, 150.0, 1.0), ((0.01111111111,), 1.001, 1, 100.0, 1.0), ((0.011111, 0.324), 1.9999, 2, 97.0, 1.0),
        ((0.012345, 0.457), 0.456789, 1, 100.0, 0.5), ((0.011111, 0.45), 0.6, 1, 97.0, 0.5), ((0.3511111111,), 1.33, 1, 97.0, 1.0)],
    1: [((0.01111111111,), 3.0, 1, 145.0, 1.0), ((0.01111111111,), 0.6, 1, 132.0, 1.0), ((0.01111111111,), 0.3, 1, 138.0, 1.0),
        ((0.01111111111,), 1.0, 1, 157.0, 1.0), ((0.01111111111,), 1.001, 1, 148.0, 1.0), ((0.011111, 0.324), 1.9999, 2, 127.0, 1.0),
        ((0.012345, 0.457), 0.456789, 1, 123.0, 0.5), ((0.011111, 0.45), 0.6, 1, 126.0, 0.5), ((0.43111111111,), 1.33, 1, 121.0, 1.0)],
}


@pytest.mark.parametrize("converter", [2, 1])
@pytest.mark.parametrize("case", range(9))
def test_snr_test_of_snr_bw_test_c(dsp, converter, case):
    """thirdparty/libsamplerate/tests/snr_bw_test.c:175-287 with the reference's own thresholds."""
    freqs, ratio, peaks, need, peak_value = SNR_CASES[converter][case]
    buffer_len, max_spec = 50000, 1 << 15
    if ratio >= 1.0:
        out_len = max_spec
        in_len = min(int(math.ceil(max_spec / ratio)), buffer_len)
    else:
        out_len = int(math.ceil(buffer_len * ratio)) & ~0xF
        out_len = min(out_len, max_spec)
        in_len = int(math.ceil(out_len / ratio))
    x = _gen_windowed_sines(freqs, in_len)
    err, out, _ = src_simple(dsp, x, ratio, converter, out_len)
    assert err == 0
    assert abs(out.size - out_len) <= 4
    assert abs(float(np.abs(out).max()) - peak_value) <= 0.01
    snr = _calculate_snr(out, peaks)
    assert snr >= need, (converter, case, snr, need)


def test_unit_ratio_passes_a_band_limited_signal_through(dsp):
    """Algorithm identity: at ratio 1 every output sits on an input sample (fraction 0) and the filter is a unit-gain low-pass
    (80 % bandwidth): a tone well inside the pass band comes back sample for sample, away from the zero-history edges."""
    k = np.arange(6000)
    x = (0.7 * np.sin(2 * np.pi * 0.05 * k + 0.3)).astype(np.float32)
    err, out, used = src_simple(dsp, x, 1.0, 2, 6000)
    assert err == 0 and out.size >= 5990
    assert np.max(np.abs(out[100:5800] - x[100:5800])) < 1e-4


# ------------------------------------------------------------------------------------------------ the product's plan vs the sequential run

def _plan(lib, n, src_rate, dst_rate, converter, n_pos=0):
    gen, used, closed = C.c_longlong(0), C.c_longlong(0), C.c_int(0)
    pos = np.zeros(max(n_pos, 1), np.int64); frac = np.zeros(max(n_pos, 1), np.float64)
    r = lib.wmi_selftest_resample_plan(n, src_rate, dst_rate, converter, C.byref(gen), C.byref(used), C.byref(closed), n_pos,
                                       pos.ctypes.data, frac.ctypes.data)
    return r, gen.value, used.value, closed.value, pos[:n_pos], frac[:n_pos]


@pytest.fixture(scope="module")
def product_host():
    from godot_whisper_amd import runtime
    return runtime.load_library()


RATES = [48000, 44100, 32000, 22050, 96000, 192000, 8000, 11025, 24000, 88200, 12000, 47999, 16001]


@pytest.mark.parametrize("src_rate", RATES)
@pytest.mark.parametrize("converter", [2, 1])
def test_plan_frame_counts_equal_the_sequential_converter(dsp, product_host, src_rate, converter):
    rng = np.random.default_rng(src_rate)
    ratio = 16000.0 / src_rate
    inc, tab = TABLES[converter]
    lens = [0, 1, 2, 57, 441, 1000, 4096, 14669, 14670, 14671, 14788, 30000, 44100, 48000, 132300, 200001]
    lens += [int(v) for v in rng.integers(1, 300000, 12)]
    if converter == 1:
        lens = lens[:20]
    for n in lens:
        x = np.zeros(n, np.float32)
        out_frames = int(np.uint32(n) * ratio)
        err, out, used = src_simple(dsp, x, ratio, converter, out_frames)
        r, gen, used_p, closed, _, _ = _plan(product_host, n, src_rate, 16000, converter)
        assert r == 0 and err == 0
        assert gen == out.size, (src_rate, n, gen, out.size)
        assert used_p == used, (src_rate, n, used_p, used)


@pytest.mark.parametrize("src_rate", RATES)
def test_plan_positions_are_the_double_recurrence(product_host, src_rate):
    """src_sinc.c:411-416: input_index += 1 / ratio; rem = fmod_one(input_index); b_current += lrint(input_index - rem)."""
    n = 300000
    out_frames = int(np.uint32(n) * (16000.0 / src_rate))
    r, gen, used, closed, pos, frac = _plan(product_host, n, src_rate, 16000, 2, n_pos=out_frames)
    assert r == 0
    inc = 1.0 / (16000.0 / src_rate)
    x, p = 0.0, 0
    want_p = np.zeros(out_frames, np.int64); want_f = np.zeros(out_frames, np.float64)
    for i in range(out_frames):
        want_p[i] = p; want_f[i] = x
        x += inc
        rem = x - round(x)                      # Python's round() is round-half-even = lrint
        if rem < 0.0:
            rem += 1.0
        p += int(round(x - rem))
        x = rem
    assert np.array_equal(pos, want_p), (src_rate, closed)
    assert frac.tobytes() == want_f.tobytes(), (src_rate, closed)
    assert closed == (1 if src_rate in (48000, 44100, 32000, 96000, 192000, 8000, 24000, 88200) else closed)


def test_bad_ratio_and_missing_table(product_host):
    r, *_ = _plan(product_host, 1000, 16000 * 300, 16000, 2)
    assert r == -6                                           # SRC_ERR_BAD_SRC_RATIO
    r, *_ = _plan(product_host, 1000, 48000, 16000, 0)
    assert r == -10                                          # SRC_SINC_BEST_QUALITY: table is a missing blob
