"""-m gpu: the host-adjacent DSP kernels against the restatement of the host's C++
(src/speech_to_text.cpp:45-51 stereo -> mono, :53-104 high-pass + energy VAD; godot-whisper_amd/host.py) — bit-exact:
the filter recurrence and the running f32 energy sums are evaluated in sample order by one lane."""
import ctypes as C

import numpy as np
import pytest

from godot_whisper_amd import host, synth

pytestmark = pytest.mark.gpu


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
def test_vad_equals_the_host_arithmetic(product_lib, node, case):
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
    want = host.vad_simple(np.array(pcm[-3 * sr:], np.float32), sr, 500, thold, freq)
    assert bool(got) == bool(want), (case, got, want, en, host.vad_simple.last_energies)
    ea, el = host.vad_simple.last_energies
    assert np.float32(ea).view(np.uint32) == en[0].view(np.uint32) and np.float32(el).view(np.uint32) == en[1].view(np.uint32), (case, en, ea, el)


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
