"""-m gpu: the block-quantised kernels (csrc/k_quant.hip) through the C ABI against the oracle's restatement of the
reference's quantised arithmetic (oracle/port_quants.c — pinned bit for bit to the reference's own functions by
tests/test_oracle_quants.py).

Bars: the q8 row quantiser is integer / bit work -> identical quants, identical d and s bit patterns; the block dots are
integer sums (exact) combined in f32 in a different order than the reference's eight-lane AVX2 accumulators ->
|d| <= 4e-6 * sum_k |w_k||x_k|; the dequantising gather is bit-exact."""
import ctypes as C

import numpy as np
import pytest

from godot_whisper_amd import synth
from oracle import port

pytestmark = pytest.mark.gpu

QT = {"q4_0": (2, 18, False), "q4_1": (3, 20, True), "q5_0": (6, 22, False), "q5_1": (7, 24, True), "q8_0": (8, 34, False)}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _port():
    pl = port.lib()
    pl.port_quantize_row_q8.restype = None
    pl.port_quantize_row_q8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    pl.port_vec_dot_q.restype = C.c_float
    pl.port_vec_dot_q.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    pl.port_dequantize_row.restype = None; pl.port_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    return pl


def _oracle(pl, gtype, bb, has_s, wq, x):
    M, K = x.shape; N = wq.shape[0]; nb = K // 32
    qs = np.zeros((M, K), np.int8); d = np.zeros((M, nb), np.float32); s = np.zeros((M, nb), np.float32)
    for j in range(M):
        pl.port_quantize_row_q8(_p(x[j]), K, 1 if has_s else 0, _p(qs[j]), _p(d[j]), _p(s[j]))
    out = np.zeros((M, N), np.float32)
    for j in range(M):
        for i in range(N):
            out[j, i] = pl.port_vec_dot_q(gtype, K, _p(wq[i]), _p(qs[j]), _p(d[j]), _p(s[j]))
    return qs, d, s, out


def _inputs(qtype, M, N, K, seed):
    gtype, bb, has_s = QT[qtype]
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    w[0, :40] *= 20.0
    wq = np.frombuffer(synth.quantize_blocks(w, qtype), np.uint8).reshape(N, K // 32 * bb).copy()
    x = rng.standard_normal((M, K)).astype(np.float32)
    x[0, 5] = 37.0
    if M > 1:
        x[1, :32] = 0.0                                               # an all-zero block: d = 0, id = 0
    if M > 2:
        x[2, :32] = np.arange(32, dtype=np.float32) * 0.5 - 8.0      # products on .5 ties -> round to even
    return gtype, bb, has_s, w, wq, x


@pytest.mark.parametrize("qtype", list(QT))
@pytest.mark.parametrize("mode,M,N,K", [(0, 1, 160, 128), (0, 5, 1280, 1280), (0, 8, 96, 384), (0, 13, 224, 512), (0, 32, 64, 256),
                                        (0, 3, 200, 5120),
                                        (1, 200, 256, 128), (1, 77, 128, 384), (1, 130, 384, 1280), (1, 64, 128, 5120)])
def test_quantised_matmul_against_the_oracle(product_lib, qtype, mode, M, N, K):
    gtype, bb, has_s, w, wq, x = _inputs(qtype, M, N, K, seed=1000 * mode + M + N + K)
    pl = _port()
    qs_o, d_o, s_o, out_o = _oracle(pl, gtype, bb, has_s, wq, x)
    out = np.zeros((M, N), np.float32); qs = np.zeros((M, K), np.int8); ds = np.zeros((M, K // 32, 2), np.float32)
    rc = product_lib.wmi_selftest_quant(0, gtype, mode, _p(wq), _p(x), None, M, N, K, _p(out), _p(qs), _p(ds))
    assert rc == 0
    assert np.array_equal(qs, qs_o), "q8 quants differ"
    assert np.array_equal(ds[:, :, 0].copy().view(np.uint32), d_o.view(np.uint32)), "q8 scales differ"
    if has_s:
        assert np.array_equal(ds[:, :, 1].copy().view(np.uint32), s_o.view(np.uint32)), "q8 sums differ"
    # dequantised operands give the magnitude the f32 rounding bound scales with
    wd = np.zeros((N, K), np.float32)
    for i in range(N):
        pl.port_dequantize_row(gtype, _p(wq[i]), _p(wd[i]), K)
    mag = np.abs(x).astype(np.float64) @ np.abs(wd).astype(np.float64).T
    err = np.abs(out.astype(np.float64) - out_o.astype(np.float64))
    assert np.all(err <= 4e-6 * mag + 1e-30), (float((err / (mag + 1e-30)).max()), float(err.max()))
    # and the quantised product is what it claims to be: close to the f32 product of the dequantised weights with x
    ref = x.astype(np.float64) @ wd.astype(np.float64).T
    assert np.sqrt(((out - ref) ** 2).mean()) <= 2e-2 * np.sqrt((ref ** 2).mean())


@pytest.mark.parametrize("qtype", list(QT))
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (77, 128, 384), (1500, 384, 1280), (260, 128, 5120)])
def test_quantised_matmul_f16_form_against_the_oracle(product_lib, qtype, M, N, K):
    """The large-M form of the quantised projections (k_qdequant + the f16 GEMM): the activation rows are the reference's own
    q8 quants and scales (bit-identical, as above); the product rounds each factor of each term — f16(d_w q_w + m_w), f16(d_a q_a) — to
    f16 before the f32 accumulation, so |out - reference| <= (2^-11 + 2^-11 + f32 slack) * sum_k |w_k||x_k| element by element, and
    ~2^-11 in the rms (independent roundings)."""
    gtype, bb, has_s, w, wq, x = _inputs(qtype, M, N, K, seed=3000 + M + N + K)
    pl = _port()
    qs_o, d_o, s_o, out_o = _oracle(pl, gtype, bb, has_s, wq, x)
    out = np.zeros((M, N), np.float32); qs = np.zeros((M, K), np.int8); ds = np.zeros((M, K // 32, 2), np.float32)
    assert product_lib.wmi_selftest_quant(0, gtype, 3, _p(wq), _p(x), None, M, N, K, _p(out), _p(qs), _p(ds)) == 0
    assert np.array_equal(qs, qs_o), "q8 quants differ"
    assert np.array_equal(ds[:, :, 0].copy().view(np.uint32), d_o.view(np.uint32)), "q8 scales differ"
    wd = np.zeros((N, K), np.float32)
    for i in range(N):
        pl.port_dequantize_row(gtype, _p(wq[i]), _p(wd[i]), K)
    xq = (qs_o.reshape(M, K // 32, 32).astype(np.float64) * d_o[:, :, None].astype(np.float64)).reshape(M, K)
    mag = np.abs(xq) @ np.abs(wd).astype(np.float64).T
    err = np.abs(out.astype(np.float64) - out_o.astype(np.float64))
    assert np.all(err <= 1.0e-3 * mag + 1e-30), (float((err / (mag + 1e-30)).max()), float(err.max()))
    rms = np.sqrt((err ** 2).mean()) / np.sqrt((out_o.astype(np.float64) ** 2).mean())
    print(f"{qtype} f16 form {M}x{N}x{K}: rms-rel {rms:.2e}, worst |err| / sum|w||x| {float((err / (mag + 1e-30)).max()):.2e}")
    assert rms <= 6e-4
    if M >= 256:
        # the product's route (weight expansion inside the row quantiser's launch, then qgemm's dispatch): same bits
        out4 = np.zeros((M, N), np.float32); qs4 = np.zeros((M, K), np.int8)
        assert product_lib.wmi_selftest_quant(0, gtype, 4, _p(wq), _p(x), None, M, N, K, _p(out4), _p(qs4), None) == 0
        assert np.array_equal(qs4, qs_o) and np.array_equal(out4.view(np.uint32), out.view(np.uint32))


@pytest.mark.parametrize("qtype", list(QT))
def test_quantised_embedding_gather_is_bit_exact(product_lib, qtype):
    gtype, bb, has_s = QT[qtype]
    N, K = 300, 384
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    wq = np.frombuffer(synth.quantize_blocks(w, qtype), np.uint8).reshape(N, K // 32 * bb).copy()
    tokens = np.array([0, 31, 32, 33, 299, 150, 64], np.int32)
    out = np.zeros((tokens.size, K), np.float32)
    assert product_lib.wmi_selftest_quant(0, gtype, 2, _p(wq), None, _p(tokens), tokens.size, N, K, _p(out), None, None) == 0
    pl = _port()
    want = np.zeros_like(out)
    for i, t in enumerate(tokens):
        pl.port_dequantize_row(gtype, _p(wq[int(t)]), _p(want[i]), K)
    assert np.array_equal(out, want)                   # value-exact (the gather adds a zero "positional" row: -0 becomes +0)
