"""The rest of the whisper.h surface on the GPU (include/whisper_mi355.h, second half): caller-owned states, the
*_no_state and loader constructors, whisper_full_parallel against the compiled reference, and the bench entry points.
Everything is called through the C ABI; the reference is used only as the checker."""
import ctypes as C

import numpy as np
import pytest

import golden_util as gu
from godot_whisper_amd import abi, host, synth
from oracle import reflib

pytestmark = pytest.mark.gpu

NEW_SEGMENT_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _params(lib):
    p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_GREEDY)
    p.language = b"en"; p.temperature_inc = 0.0; p.print_progress = False; p.token_timestamps = True
    return p


def _ctx_segments(lib, ctx):
    out = []
    for i in range(lib.whisper_full_n_segments(ctx)):
        toks = [lib.whisper_full_get_token_data(ctx, i, j) for j in range(lib.whisper_full_n_tokens(ctx, i))]
        out.append((lib.whisper_full_get_segment_t0(ctx, i), lib.whisper_full_get_segment_t1(ctx, i),
                    bytes(lib.whisper_full_get_segment_text(ctx, i)), [(t.id, t.tid, t.p, t.plog, t.pt, t.ptsum, t.t0, t.t1, t.vlen) for t in toks]))
    return out


def _state_segments(lib, ctx, st):
    out = []
    for i in range(lib.whisper_full_n_segments_from_state(st)):
        n = lib.whisper_full_n_tokens_from_state(st, i)
        toks = [lib.whisper_full_get_token_data_from_state(st, i, j) for j in range(n)]
        for j, t in enumerate(toks):                              # the per-field getters agree with the struct getter
            assert lib.whisper_full_get_token_id_from_state(st, i, j) == t.id
            assert lib.whisper_full_get_token_p_from_state(st, i, j) == t.p
            assert lib.whisper_full_get_token_text_from_state(ctx, st, i, j) == lib.whisper_token_to_str(ctx, t.id)
        assert lib.whisper_full_get_segment_speaker_turn_next_from_state(st, i) is False
        out.append((lib.whisper_full_get_segment_t0_from_state(st, i), lib.whisper_full_get_segment_t1_from_state(st, i),
                    bytes(lib.whisper_full_get_segment_text_from_state(st, i)), [(t.id, t.tid, t.p, t.plog, t.pt, t.ptsum, t.t0, t.t1, t.vlen) for t in toks]))
    return out


def test_caller_owned_states_are_independent_working_sets(product_lib):
    """Two whisper_states on one context: each *_with_state call works on its own caches and results, bit-identical to the
    same call on the context's own state, and leaves the other states alone (W/whisper.cpp:3001-3120 ownership split)."""
    lib = product_lib
    model = synth.make_model("micro.en", seed=11)
    pcm_a, pcm_b = synth.make_pcm(12.0, seed=21), synth.make_pcm(9.0, seed=22)
    node = host.SpeechToText(lib); node.set_language_model(model); ctx = node.ctx
    p = _params(lib)
    assert lib.whisper_full(ctx, p, _fp(pcm_a), pcm_a.size) == 0
    exp_a = _ctx_segments(lib, ctx)
    assert lib.whisper_full(ctx, p, _fp(pcm_b), pcm_b.size) == 0
    exp_b = _ctx_segments(lib, ctx)
    assert exp_a and exp_b and exp_a != exp_b

    sa, sb = lib.whisper_init_state(ctx), lib.whisper_init_state(ctx)
    assert sa and sb and sa != sb
    assert lib.whisper_full_with_state(ctx, sa, p, _fp(pcm_a), pcm_a.size) == 0
    assert lib.whisper_full_with_state(ctx, sb, p, _fp(pcm_b), pcm_b.size) == 0
    assert _state_segments(lib, ctx, sa) == exp_a                  # not disturbed by the run on sb
    assert _state_segments(lib, ctx, sb) == exp_b
    assert _ctx_segments(lib, ctx) == exp_b                        # the context's own state still holds its last result
    assert lib.whisper_full_lang_id_from_state(sa) == lib.whisper_full_lang_id(ctx)
    assert not lib.whisper_full_get_segment_speaker_turn_next(ctx, 0)

    # stage calls on a state: mel -> encoder -> one decoder call, logits equal to the same calls on the context
    sot = (C.c_int32 * 1)(lib.whisper_token_sot(ctx)); nv = lib.whisper_n_vocab(ctx)
    assert lib.whisper_pcm_to_mel(ctx, _fp(pcm_a), pcm_a.size, 1) == 0 and lib.whisper_encode(ctx, 0, 1) == 0
    assert lib.whisper_decode(ctx, sot, 1, 0, 1) == 0
    want = np.ctypeslib.as_array(lib.whisper_get_logits(ctx), (nv,)).copy()
    assert lib.whisper_pcm_to_mel_with_state(ctx, sb, _fp(pcm_a), pcm_a.size, 1) == 0
    assert lib.whisper_n_len_from_state(sb) == lib.whisper_n_len(ctx) == 1 + (pcm_a.size + 200 - 400) // 160      # W/whisper.cpp:2827
    assert lib.whisper_encode_with_state(ctx, sb, 0, 1) == 0 and lib.whisper_decode_with_state(ctx, sb, sot, 1, 0, 1) == 0
    got = np.ctypeslib.as_array(lib.whisper_get_logits_from_state(sb), (nv,)).copy()
    assert np.array_equal(got, want)
    assert _state_segments(lib, ctx, sa) == exp_a                  # still untouched
    # a mel supplied by the caller goes to the state it names
    mel = np.zeros((80, 3000), np.float32)
    assert lib.whisper_set_mel_with_state(ctx, sa, _fp(mel), 3000, 80) == 0 and lib.whisper_n_len_from_state(sa) == 3000
    assert lib.whisper_set_mel_with_state(ctx, sa, _fp(mel), 3000, 81) == -1
    assert lib.whisper_n_len(ctx) == 1 + (pcm_a.size + 200 - 400) // 160
    probs = (C.c_float * 100)()
    lid = lib.whisper_lang_auto_detect_with_state(ctx, sa, 0, 1, probs)           # soft-max over the 100 ids after <sot> (W/whisper.cpp:3569-3640)
    assert 0 <= lid <= 99 and abs(sum(probs) - 1.0) < 1e-4 and max(probs) == probs[lid]
    assert lib.whisper_lang_auto_detect_with_state(ctx, sa, 30000, 1, probs) == -2 and lib.whisper_lang_auto_detect_with_state(ctx, sa, -10, 1, probs) == -1
    lib.whisper_free_state(sa); lib.whisper_free_state(sb)
    assert lib.whisper_full(ctx, p, _fp(pcm_a), pcm_a.size) == 0 and _ctx_segments(lib, ctx) == exp_a
    node.close()


@pytest.mark.parametrize("strategy", ["greedy", "beam3"])
def test_states_of_one_context_compute_concurrently(product_lib, strategy):
    """whisper_full_with_state from parallel threads on different states of ONE context — what the reference's whisper_full_parallel does
    with its states (W/whisper.cpp:5837-5858).  Every result equals the one a state with the same call history gives when the calls
    take turns (a state carries its decoders' mt19937 generators from call to call, so beam search depends on the history — as in the
    reference), and the calls really run side by side: the two threads finish well before the sum of their calls' solo durations
    (before round 6 every *_with_state call took the context's one lock and the threads took turns)."""
    import os, threading, time
    lib = product_lib
    model = synth.make_model("base.en", seed=4242)
    pcms = [synth.make_pcm(30.0, seed=900 + i) for i in range(4)]
    node = host.SpeechToText(lib); node.set_language_model(model); ctx = node.ctx
    if strategy == "greedy":
        p = _params(lib)
    else:
        p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
        p.language = b"en"; p.temperature_inc = 0.0; p.print_progress = False; p.token_timestamps = True; p.beam_search.beam_size = 3
    p.max_tokens = 24
    reps = 6
    order = [list(range(len(pcms))), list(range(len(pcms)))[::-1]]      # the two threads are never on the same audio at the same time
    # taking turns: two states, thread t's call sequence on state t, one call after the other
    alone = [lib.whisper_init_state(ctx) for _ in range(2)]
    assert all(alone)
    want = [[] for _ in alone]; solo = 0.0
    for t, st in enumerate(alone):
        for rep in range(2 + reps):
            for i in order[t]:
                t0 = time.perf_counter()
                assert lib.whisper_full_with_state(ctx, st, p, _fp(pcms[i]), pcms[i].size) == 0
                if rep >= 2:
                    solo += time.perf_counter() - t0               # the timed calls of the side-by-side run below, alone
                else:
                    want[t].append(_state_segments(lib, ctx, st))
    assert want[0][0] and want[0][0] != want[0][1]
    # side by side: two fresh states, the same call sequences from two threads
    states = [lib.whisper_init_state(ctx) for _ in range(2)]
    assert all(states)
    got = [[] for _ in states]; errs = []
    def work(t, n_reps, check):
        try:
            for rep in range(n_reps):
                for i in order[t]:
                    rc = lib.whisper_full_with_state(ctx, states[t], p, _fp(pcms[i]), pcms[i].size)
                    if check:
                        got[t].append((rc, _state_segments(lib, ctx, states[t])))
                    elif rc != 0:
                        errs.append(rc)
        except Exception as e:                                    # pragma: no cover
            errs.append(e)
    def run(n_reps, check):
        th = [threading.Thread(target=work, args=(t, n_reps, check)) for t in range(len(states))]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        return time.perf_counter() - t0
    run(2, True)                                                  # checked pass: every result read back between the calls
    assert not errs, errs
    for t in range(len(states)):
        assert len(got[t]) == len(want[t])
        for c, ((rc, segs), w) in enumerate(zip(got[t], want[t])):
            assert rc == 0 and segs == w, (strategy, t, c)
    wall = run(reps, False)                                       # timed pass: the calls only
    assert not errs, errs
    print(f"{strategy}: two threads x {reps * len(pcms)} calls: {wall * 1e3:.1f} ms side by side, {solo * 1e3:.1f} ms as the sum of the calls alone ({solo / wall:.2f} x)")
    assert wall < float(os.environ.get("WMI_TEST_CONC_BOUND", "0.8")) * solo, (wall, solo)
    for st in states + alone:
        lib.whisper_free_state(st)
    node.close()


def test_no_state_and_loader_constructors(product_lib):
    """whisper_init_*_no_state leaves the context without a working set (whisper_full refuses), whisper_init_state supplies
    one; the loader-callback constructor reads the same image through read/eof/close (W/whisper.cpp:3178-3338)."""
    lib = product_lib
    model = synth.make_model("micro.en", seed=11); pcm = synth.make_pcm(8.0, seed=23)
    p = _params(lib)
    node = host.SpeechToText(lib); node.set_language_model(model)
    assert lib.whisper_full(node.ctx, p, _fp(pcm), pcm.size) == 0
    want = _ctx_segments(lib, node.ctx)
    node.close()

    buf = C.create_string_buffer(model, len(model))
    ctx = lib.whisper_init_from_buffer_no_state(C.cast(buf, C.c_void_p), len(model))
    assert ctx
    assert lib.whisper_full(ctx, p, _fp(pcm), pcm.size) == -1                     # no state
    st = lib.whisper_init_state(ctx)
    assert st and lib.whisper_full_with_state(ctx, st, p, _fp(pcm), pcm.size) == 0
    assert _state_segments(lib, ctx, st) == want
    lib.whisper_free_state(st); lib.whisper_free(ctx)

    pos = [0]; closed = [0]
    L = abi.whisper_model_loader
    def rd(_, out, n):
        k = min(n, len(model) - pos[0]); C.memmove(out, model[pos[0]:pos[0] + k], k); pos[0] += k; return k
    loader = L(None, L._fields_[1][1](rd), L._fields_[2][1](lambda _: pos[0] >= len(model)),
               L._fields_[3][1](lambda _: closed.__setitem__(0, closed[0] + 1)))
    ctx = lib.whisper_init(C.cast(C.pointer(loader), C.c_void_p))
    assert ctx and closed[0] == 1 and pos[0] == len(model)
    assert lib.whisper_full(ctx, p, _fp(pcm), pcm.size) == 0 and _ctx_segments(lib, ctx) == want
    lib.whisper_free(ctx)
    ctx = lib.whisper_init_from_buffer(C.cast(buf, C.c_void_p), len(model))       # deprecated form = default context params
    assert ctx and lib.whisper_full(ctx, p, _fp(pcm), pcm.size) == 0 and _ctx_segments(lib, ctx) == want
    lib.whisper_free(ctx)


def _ids_and_times(segs):
    return [(a[0], a[1], [t[0] for t in a[3]]) for a in segs]


@pytest.mark.parametrize("n_proc,offset_ms", [(2, 0), (3, 0), (3, 2000)])
def test_full_parallel_is_the_reference_composition(product_lib, n_proc, offset_ms):
    """whisper_full_parallel = the pieces transcribed on their own fresh states + the reference's merge (time shift by the
    piece start, no-overlap clamp, one callback per merged segment; W/whisper.cpp:5817-5924), checked two ways: against
    the composition done by hand with whisper_full on fresh contexts, and against the compiled reference's own call."""
    lib = product_lib
    model = synth.make_model("micro.en", seed=91); pcm = synth.make_pcm(100.0, seed=92, gate=True)
    p = _params(lib); p.offset_ms = offset_ms; p.token_timestamps = False

    calls = []
    cb = NEW_SEGMENT_CB(lambda c, s, n, ud: calls.append(n))
    p_cb = _params(lib); p_cb.offset_ms = offset_ms; p_cb.token_timestamps = False
    p_cb.new_segment_callback = C.cast(cb, C.c_void_p)
    node = host.SpeechToText(lib); node.set_language_model(model)
    assert lib.whisper_full_parallel(node.ctx, p_cb, _fp(pcm), pcm.size, n_proc) == 0
    got = _ctx_segments(lib, node.ctx)
    node.close()
    assert sum(calls) == len(got) and len(got) >= n_proc

    # by hand: the same split, every piece on a fresh context
    off = 16000 * offset_ms // 1000; per = (pcm.size - off) // n_proc
    want = []
    for i in range(n_proc):
        q = _params(lib); q.token_timestamps = False
        if i == 0:
            q.offset_ms = offset_ms; piece = pcm[:off + per]
        else:
            start = off + i * per
            piece = pcm[start:] if i == n_proc - 1 else pcm[start:start + per]
        piece = np.ascontiguousarray(piece)
        node = host.SpeechToText(lib); node.set_language_model(model)
        assert lib.whisper_full(node.ctx, q, _fp(piece), piece.size) == 0
        for (t0, t1, text, toks) in _ctx_segments(lib, node.ctx):
            if i > 0:
                shift = 100 * (i * per) // 16000 + int(offset_ms / 10.0)
                t0 += shift; t1 += shift
                if want:
                    t0 = max(t0, want[-1][1])
            want.append((t0, t1, text, toks))
        node.close()
    assert got == want

    if not reflib.available():
        pytest.skip("hand composition checked; the compiled reference is absent")
    R = reflib.lib()
    quiet = abi.ggml_log_callback(lambda lvl, txt, ud: None)
    R.whisper_log_set(C.cast(quiet, C.c_void_p), None); R._quiet_cb = quiet
    rnode = host.SpeechToText(R); rnode.set_language_model(model)
    pr = _params(R); pr.offset_ms = offset_ms; pr.token_timestamps = False
    assert R.whisper_full_parallel(rnode.ctx, pr, _fp(pcm), pcm.size, n_proc) == 0
    ref = _ctx_segments(R, rnode.ctx)
    rnode.close()
    a, b = _ids_and_times(got), _ids_and_times(ref)
    flat_a = [t for s in a for t in s[2]]; flat_b = [t for s in b for t in s[2]]
    if flat_a == flat_b:
        assert a == b                                                # same tokens => same segments and merged timestamps
    else:                                                            # a near-tie somewhere: the streams part there
        first = next(i for i, (x, y) in enumerate(zip(flat_a, flat_b)) if x != y)
        assert first >= 20, (first, flat_a[:first + 1][-3:], flat_b[:first + 1][-3:])
        assert a[0] == b[0]


def test_bench_entry_points_report_the_device(product_lib):
    """whisper_bench_memcpy_str / whisper_bench_ggml_mul_mat_str (W/whisper.cpp:6027-6266 time the host): here HBM copy
    bandwidth and the MFMA GEMM; sanity floors only, the numbers themselves belong to bench.py and profiles/."""
    lib = product_lib
    s = lib.whisper_bench_memcpy_str(1).decode()
    gbs = float(s.split("memcpy:")[1].split("GB/s")[0])
    assert gbs > 500.0 and "sum:" in s, s
    m = lib.whisper_bench_ggml_mul_mat_str(1).decode()
    lines = [l for l in m.splitlines() if "GFLOPS" in l]
    assert [int(l.split("x")[0]) for l in lines] == [64, 128, 256, 512, 1024, 2048, 4096], m
    gf = [float(l.split("F16")[1].split("GFLOPS")[0]) for l in lines]
    assert gf[-1] > 100e3 and gf[-1] > gf[2], m                      # 4096^3 well above 100 TFLOP/s
    print(s, m)


def test_print_realtime_output_equals_the_reference(product_lib, capfd):
    """params.print_realtime (+ print_timestamps): the lines whisper_full prints while it emits segments
    (W/whisper.cpp:5722-5729, 5769-5776; "[hh:mm:ss.mmm --> hh:mm:ss.mmm]  text") are the reference's, byte for byte."""
    if not reflib.available():
        pytest.skip("needs the compiled reference")
    R = reflib.lib()
    quiet = abi.ggml_log_callback(lambda lvl, txt, ud: None)
    R.whisper_log_set(C.cast(quiet, C.c_void_p), None); R._quiet_cb2 = quiet
    libc = C.CDLL(None)
    model = synth.make_model("micro.en", seed=91); pcm = synth.make_pcm(45.0, seed=93, gate=True)
    outs, toks, segs, texts = [], [], [], []
    for stamps in (True, False):
        for L in (product_lib, R):
            node = host.SpeechToText(L); node.set_language_model(model)
            p = _params(L); p.token_timestamps = False; p.print_realtime = True; p.print_timestamps = stamps
            capfd.readouterr()
            assert L.whisper_full(node.ctx, p, _fp(pcm), pcm.size) == 0
            libc.fflush(None)
            outs.append(capfd.readouterr().out)
            # (id, tid): segment times are printed from tid, the most probable timestamp at that step (W/whisper.cpp:5715-5716) —
            # an arg-max of its own, with its own near-ties
            sg = _ctx_segments(L, node.ctx)
            toks.append([t for s in sg for t in s[3]]); segs.append([len(s[3]) for s in sg]); texts.append([s[2] for s in sg])
            node.close()
    for k in (0, 2):
        assert outs[k].strip(), outs
        ids = [[(t[0], t[1]) for t in toks[k + j]] for j in (0, 1)]
        if ids[0] == ids[1]:
            assert outs[k] == outs[k + 1]
            continue
        # The streams part at a near-tie of the synthetic weights (the arg-max of id or of tid): both picks carry nearly the same
        # probability there, and every line printed from tokens before that point is the reference's
        d = next(i for i, (a, b) in enumerate(zip(ids[0], ids[1])) if a != b)
        ta, tb = toks[k][d], toks[k + 1][d]
        if ta[0] != tb[0]: assert abs(ta[2] - tb[2]) <= 0.05 * max(ta[2], tb[2]), (d, ta, tb)
        else:              assert abs(ta[4] - tb[4]) <= 0.05 * max(ta[4], tb[4]), (d, ta, tb)
        a, b = outs[k].splitlines(), outs[k + 1].splitlines()
        whole = 0; seen = 0
        for na, nb in zip(segs[k], segs[k + 1]):
            if na != nb or seen + na > d: break
            seen += na; whole += 1
        if k == 0: assert a[:whole] == b[:whole], (whole, a, b)          # one line per segment
        else:                                                             # no timestamps: the texts run on without line breaks
            assert texts[k][:whole] == texts[k + 1][:whole]
            assert outs[k].strip() == b"".join(texts[k]).decode("utf-8", "replace").strip()
    assert outs[0].startswith("[00:00:0")


def test_in_process_device_pool_equals_one_context(product_lib):
    """wmi_pool_*: several GPUs behind one host process — here two contexts on the one GPU of the test box.  The second
    context is built from the header image + a device copy of the first one's weight arena (no re-parse); chunks go to
    context c mod 2 and must come back exactly as one context transcribes them."""
    model = synth.make_model("micro.en", seed=1234)
    pcms = [synth.make_pcm(5.0 + i, seed=900 + i) for i in range(5)]
    buf = C.create_string_buffer(model, len(model))
    devs = (C.c_int * 2)(0, 0)
    pool = product_lib.wmi_pool_init(C.cast(buf, C.c_void_p), len(model), devs, 2)
    assert pool and product_lib.wmi_pool_size(pool) == 2
    one = host.SpeechToText(product_lib); one.set_language_model(model)
    try:
        c0, c1 = product_lib.wmi_pool_context(pool, 0), product_lib.wmi_pool_context(pool, 1)
        assert product_lib.wmi_weights_bytes(c0, 0) == product_lib.wmi_weights_bytes(c1, 0) > 0
        assert product_lib.wmi_arena_ptr(c0) != product_lib.wmi_arena_ptr(c1)
        p = one.full_params("", 0); p.temperature_inc = 0.0
        ptrs = (C.c_void_p * len(pcms))(*[b.ctypes.data for b in pcms]); lens = (C.c_int * len(pcms))(*[b.size for b in pcms])
        assert product_lib.wmi_pool_full(pool, p, ptrs, lens, len(pcms)) == 0
        want = one.transcribe_batch(pcms, params=p)
        for c in range(len(pcms)):
            ctx = product_lib.wmi_pool_select(pool, c)
            assert ctx == (c0 if c % 2 == 0 else c1)
            node = host.SpeechToText(product_lib); node.ctx = ctx
            got = node.collect(); node.ctx = None
            g, w = gu.tokens_array(got), gu.tokens_array(want[c])
            assert g.shape == w.shape and np.array_equal(g[:, [0, 1, 6, 7]], w[:, [0, 1, 6, 7]]) and bytes(got[0]) == bytes(want[c][0]), c
        assert product_lib.wmi_pool_select(pool, len(pcms)) is None
        assert product_lib.wmi_pool_device_time_us(pool, 0) > 0
    finally:
        one.close(); product_lib.wmi_pool_free(pool)


@pytest.mark.parametrize("kind", ["beam5", "best_of_t04", "q5_1_beam3"])
def test_full_batch_runs_unlockable_chunks_on_replica_contexts(product_lib, kind):
    """wmi_full_batch with a strategy the lock-step rows cannot carry (beam search, t > 0): the chunks go through the whisper_full
    driver on replica contexts (own state and stream, the weight arena shared) — the worker threads of whisper_full_parallel
    (W/whisper.cpp:5837-5913).  Every chunk must come back exactly as whisper_full returns it on a fresh context, whatever the
    number of replicas (0 = one at a time, 1, default 3), in every field."""
    model = synth.make_model("micro.en", seed=1234)
    if kind == "q5_1_beam3":
        model = synth.quantize_model(model, "q5_1")
    pcms = [synth.make_pcm(6.0 + 1.5 * i, seed=2100 + i) for i in range(7)]

    def params(lib):
        if kind == "best_of_t04":
            p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_GREEDY)
            p.temperature = 0.4; p.greedy.best_of = 4
        else:
            p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
            p.beam_search.beam_size = 5 if kind == "beam5" else 3
        p.language = b"en"; p.temperature_inc = 0.0; p.print_progress = False; p.token_timestamps = True; p.max_tokens = 24
        return p

    want = []
    for pcm in pcms:                                               # whisper_full on a fresh context per chunk
        one = host.SpeechToText(product_lib); one.set_language_model(model)
        assert product_lib.whisper_full(one.ctx, params(product_lib), _fp(pcm), pcm.size) == 0
        want.append(_ctx_segments(product_lib, one.ctx))
        one.close()
    assert sum(len(s[3]) for w in want for s in w) >= 20 and len({tuple(t[0] for s in w for t in s[3]) for w in want}) >= 4
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    try:
        ptrs = (C.c_void_p * len(pcms))(*[b.ctypes.data for b in pcms]); lens = (C.c_int * len(pcms))(*[b.size for b in pcms])
        assert product_lib.wmi_set_batch_replicas(node.ctx, 0) == -1
        for n_rep in (0, 1, -1):
            product_lib.wmi_set_batch_replicas(node.ctx, n_rep)
            assert product_lib.wmi_full_batch(node.ctx, params(product_lib), ptrs, lens, len(pcms), 0) == 0
            for c in range(len(pcms)):
                assert product_lib.wmi_batch_chunk_mode(node.ctx, c) == 1
                assert product_lib.wmi_batch_select(node.ctx, c) == len(want[c])
                assert _ctx_segments(product_lib, node.ctx) == want[c], (kind, n_rep, c)
        # a lock-step call on the same context afterwards still works (replicas stay parked)
        g = _params(product_lib)
        assert product_lib.wmi_full_batch(node.ctx, g, ptrs, lens, len(pcms), 0) == 0
        assert product_lib.wmi_batch_chunk_mode(node.ctx, 0) == 0
    finally:
        node.close()
