"""The oracle must be trusted before it checks anything: pin the CPU restatement (oracle/whisper_port.cpp)
against (a) the golden vectors produced by the reference's own code (tests/golden/hotpath.npz) and
(b) — where the compiled reference is present — the reference itself, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import golden_util as gu
import stage_compare as sc
from oracle import port, reflib

G = np.load(gu.GOLDEN / "hotpath.npz")

pytestmark = pytest.mark.skipif(not port.available(), reason="oracle/libwhisper_port.so not built (python __graft_entry__.py build)")


def close(a, b, rel=1e-6, abs_=1e-6):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rel, atol=abs_)


def check_summary(prefix, x, stride=997):
    s = gu.summary(x, stride)
    assert int(s["n"]) == int(G[f"{prefix}/n"])
    close(s["sample"], G[f"{prefix}/sample"])
    close(s["sum"], G[f"{prefix}/sum"], rel=1e-6, abs_=1e-4)
    close(s["sumsq"], G[f"{prefix}/sumsq"], rel=1e-6, abs_=1e-4)


def check_logits(prefix, l):
    s = gu.logits_summary(l)
    assert list(s["top_ids"][:8]) == list(G[f"{prefix}/top_ids"][:8])
    close(s["top_vals"], G[f"{prefix}/top_vals"], rel=1e-5, abs_=1e-5)
    close(s["sample"], G[f"{prefix}/sample"], rel=1e-5, abs_=1e-5)


@pytest.mark.parametrize("name", list(gu.CASES))
def test_port_matches_reference_goldens(name):
    model, pcm, actx = gu.case_inputs(name)
    ps = port.PortSide(model)
    try:
        mel, n_org = ps.mel(pcm)
        assert list(mel.shape) + [n_org] == list(G[f"{name}/mel_shape"])
        check_summary(f"{name}/mel", mel)
        enc = ps.encode(0, actx)
        for k in ("embd_conv", "embd_enc", "cross_k", "cross_v"):
            check_summary(f"{name}/{k}", enc[k])
        sot = ps.sot
        prompt = [sot] if ps.NV < 51865 else [sot, sot + 1, 50359]
        lg = ps.decode(prompt, 0)
        check_logits(f"{name}/logits_prompt", lg)
        for i, tok in enumerate(G[f"{name}/fed_tokens"]):
            assert int(np.argmax(lg[:50256])) == int(tok)
            lg = ps.decode([int(tok)], len(prompt) + i)
            check_logits(f"{name}/logits_step{i}", lg)
        many = prompt + [int(x) for x in (np.arange(11) * 997 + 1000)]
        check_logits(f"{name}/logits_batch", ps.decode(many, 0))
    finally:
        ps.close()


@pytest.mark.skipif(not reflib.available(), reason="compiled reference absent (only in the build container)")
def test_port_is_bit_exact_against_compiled_reference(ref_lib):
    model, pcm, actx = gu.case_inputs("en4_ctx328")
    ref = sc.RefSide(ref_lib, model); ps = port.PortSide(model)
    try:
        mr, _ = ref.mel(pcm); mp, _ = ps.mel(pcm)
        assert np.array_equal(mr, mp)
        er = ref.encode(0, actx); ep = ps.encode(0, actx)
        for k in er:
            assert np.array_equal(er[k], ep[k]), k
        lr = ref.decode([ps.sot], 0); lp = ps.decode([ps.sot], 0)
        assert np.array_equal(lr, lp)
        lr = ref.decode([1000, 2000, 3000], 1); lp = ps.decode([1000, 2000, 3000], 1)
        assert np.array_equal(lr, lp)
    finally:
        ref.close(); ps.close()


@pytest.mark.skipif(not reflib.available(), reason="compiled reference absent")
def test_port_tables_equal_reference_tables(ref_lib):
    rt = np.empty(65536, np.uint16)
    assert ref_lib.ref_gelu_table(rt.ctypes.data_as(C.POINTER(C.c_uint16))) == 65536
    pg = np.empty(65536, np.uint16); pe = np.empty(65536, np.uint16)
    port.lib().port_tables(pg.ctypes.data_as(C.POINTER(C.c_uint16)), pe.ctypes.data_as(C.POINTER(C.c_uint16)))
    x = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    ok = ~np.isnan(x)
    assert np.array_equal(rt[ok], pg[ok])
