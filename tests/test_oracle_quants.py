"""Pin the oracle's restatement of the reference's block-quantised arithmetic (oracle/port_quants.c) before it checks
anything: bit for bit against the reference's own exported functions (quantize_row_q8_0/1, ggml_vec_dot_*,
dequantize_row_*: W/ggml-quants.c) and, for whole models, against the compiled reference's encoder / decoder outputs.
Runs where the compiled reference exists (the build container; on the GPU box the prebuilt library travels along)."""
import ctypes as C

import numpy as np
import pytest

import stage_compare as sc
from godot_whisper_amd import synth
from oracle import port, reflib

pytestmark = pytest.mark.skipif(not (port.available() and reflib.available()), reason="needs oracle/libwhisper_port.so and the compiled reference")

QT = {"q4_0": (2, 18, "q8_0"), "q4_1": (3, 20, "q8_1"), "q5_0": (6, 22, "q8_0"), "q5_1": (7, 24, "q8_1"), "q8_0": (8, 34, "q8_0")}


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def _ref_quantize(ref, x, kind):
    """the reference's row quantiser; returns (qs int8 [k], d f32 [nb], s f32 [nb] or None)"""
    k = x.size; nb = k // 32
    if kind == "q8_1":
        buf = np.zeros(nb * 40, np.uint8)
        ref.quantize_row_q8_1(_p(x), _p(buf), k)
        b = buf.reshape(nb, 40)
        return b[:, 8:].copy().view(np.int8).reshape(-1), b[:, 0:4].copy().view(np.float32).reshape(-1), b[:, 4:8].copy().view(np.float32).reshape(-1), buf
    buf = np.zeros(nb * 34, np.uint8)
    ref.quantize_row_q8_0(_p(x), _p(buf), k)
    b = buf.reshape(nb, 34)
    return b[:, 2:].copy().view(np.int8).reshape(-1), b[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(-1), None, buf


def _setup(ref):
    # the reference converts f16 through a table that ggml_init fills (W/ggml.c:2222-2235): make sure it ran once
    tmp = np.empty(65536, np.uint16)
    assert ref.ref_gelu_table(_p(tmp, C.POINTER(C.c_uint16))) == 65536
    for n in ("quantize_row_q8_1", "quantize_row_q8_0"):
        getattr(ref, n).restype = None; getattr(ref, n).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    pl = port.lib()
    pl.port_quantize_row_q8.restype = None
    pl.port_quantize_row_q8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    for n in ("port_vec_dot_q", "port_vec_dot_q_blockwise"):
        getattr(pl, n).restype = C.c_float
        getattr(pl, n).argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    pl.port_dequantize_row.restype = None; pl.port_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    return pl


def _rows(rng, n, k):
    """activation-like rows: gaussian, a few outliers, an all-zero block, exact ties at .5 after scaling"""
    x = rng.standard_normal((n, k)).astype(np.float32)
    x[0, 5] = 37.0; x[1, :32] = 0.0
    x[2, :32] = np.arange(32, dtype=np.float32) * 0.5 - 8.0          # amax 8 -> id 15.875; several products land on .5 ties
    x[3, :32] = 127.0 * np.sign(rng.standard_normal(32)).astype(np.float32)
    return x


@pytest.mark.parametrize("kind", ["q8_0", "q8_1"])
def test_row_quantiser_is_bit_exact(ref_lib, kind):
    pl = _setup(ref_lib)
    rng = np.random.default_rng(7)
    for k in (128, 512, 1280, 5120):
        x = _rows(rng, 6, k)
        for r in x:
            qs_r, d_r, s_r, _ = _ref_quantize(ref_lib, r, kind)
            qs = np.zeros(k, np.int8); d = np.zeros(k // 32, np.float32); s = np.zeros(k // 32, np.float32)
            pl.port_quantize_row_q8(_p(r), k, 1 if kind == "q8_1" else 0, _p(qs), _p(d), _p(s))
            assert np.array_equal(qs, qs_r)
            assert np.array_equal(d.view(np.uint32), d_r.view(np.uint32))
            if s_r is not None:
                assert np.array_equal(s.view(np.uint32), s_r.view(np.uint32))


@pytest.mark.parametrize("qtype", list(QT))
def test_vec_dot_and_dequantise_are_bit_exact(ref_lib, qtype):
    pl = _setup(ref_lib)
    gtype, bb, akind = QT[qtype]
    dot = getattr(ref_lib, f"ggml_vec_dot_{qtype}_{akind}")
    dot.restype = None; dot.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    deq = getattr(ref_lib, f"dequantize_row_{qtype}")
    deq.restype = None; deq.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(11)
    worst = 0.0
    for k in (128, 1280, 5120):
        w = (rng.standard_normal((9, k)) / np.sqrt(k)).astype(np.float32)
        w[0, :40] *= 30.0
        wq = np.frombuffer(synth.quantize_blocks(w, qtype), np.uint8).reshape(9, k // 32 * bb).copy()
        x = _rows(rng, 5, k)
        for r in x:
            qs, d, s, raw = _ref_quantize(ref_lib, r, akind)
            if s is None:
                s = np.zeros_like(d)
            for i in range(9):
                out = C.c_float()
                dot(k, C.byref(out), _p(wq[i]), _p(raw))
                got = pl.port_vec_dot_q(gtype, k, _p(wq[i]), _p(qs), _p(d), _p(s))
                assert np.float32(got).view(np.uint32) == np.float32(out.value).view(np.uint32), (qtype, k, i)
                # the per-block order (what the GPU kernels do) differs from it by f32 rounding only
                bw = pl.port_vec_dot_q_blockwise(gtype, k, _p(wq[i]), _p(qs), _p(d), _p(s))
                mag = float(np.abs(w[i]).astype(np.float64) @ np.abs(r).astype(np.float64))
                worst = max(worst, abs(bw - out.value) / mag)
        y_r = np.zeros(k, np.float32); y_p = np.zeros(k, np.float32)
        deq(_p(wq[1]), _p(y_r), k); pl.port_dequantize_row(gtype, _p(wq[1]), _p(y_p), k)
        assert np.array_equal(y_r.view(np.uint32), y_p.view(np.uint32))
    assert worst < 3e-6, worst              # f32 rounding over <= 160 block terms


@pytest.mark.parametrize("qtype", ["q5_1", "q8_0", "q4_0", "q4_1", "q5_0"])
def test_port_is_bit_exact_on_a_quantised_model(ref_lib, qtype):
    model = synth.quantize_model(synth.make_model("micro.en", seed=1234), qtype)
    pcm = synth.make_pcm(4.0, seed=5)
    ref = sc.RefSide(ref_lib, model); ps = port.PortSide(model)
    try:
        ref.mel(pcm); ps.mel(pcm)
        er = ref.encode(0, 228); ep = ps.encode(0, 228)
        for k in er:
            assert np.array_equal(er[k], ep[k]), k
        assert np.array_equal(ref.decode([ps.sot], 0), ps.decode([ps.sot], 0))
        assert np.array_equal(ref.decode([1000, 2000, 3000], 1), ps.decode([1000, 2000, 3000], 1))
    finally:
        ref.close(); ps.close()


def test_reference_sensitivity_of_quantised_models():
    """The yardstick of the GPU parity tests for quantised models: the reference arithmetic's own response to a 1e-6 relative
    change of the input.  f16 weights: ~3e-4 on the encoder output; quantised weights (8-bit activation blocks, a
    discontinuous map): an order of magnitude more.  Uses the restatement, which is bit-exact to the reference above."""
    pcm = synth.make_pcm(6.0, seed=9)
    res = {}
    for q in (None, "q5_1"):
        m = synth.make_model("micro.en", seed=1234)
        if q:
            m = synth.quantize_model(m, q)
        outs = []
        for scale in (1.0, 1.0 + 1e-6):
            ps = port.PortSide(m)
            try:
                ps.mel((pcm * np.float32(scale)).astype(np.float32))
                e = ps.encode(0, 428)["embd_enc"]
                outs.append((e, ps.decode([ps.sot], 0)))
            finally:
                ps.close()
        res[q] = (sc.err_stats(outs[1][0], outs[0][0])["rms_rel"], sc.err_stats(outs[1][1], outs[0][1])["rms_rel"])
    assert res[None][0] < 1e-3 and res[None][1] < 1.5e-3, res
    assert 2e-3 < res["q5_1"][0] < 2e-2 and 3e-3 < res["q5_1"][1] < 4e-2, res
