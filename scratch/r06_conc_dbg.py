"""Scratch (round 6): two threads x whisper_full_with_state (beam 3) on two states of one context: where do the results leave the solo ones?"""
import ctypes as C, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import abi, host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
model = synth.make_model(os.environ.get("SHAPE", "base.en"), seed=4242)
pcms = [synth.make_pcm(30.0, seed=900 + i) for i in range(4)]
node = host.SpeechToText(lib); node.set_language_model(model); ctx = node.ctx
p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
p.language = b"en"; p.temperature_inc = 0.0; p.print_progress = False; p.token_timestamps = bool(int(os.environ.get("TS", "1"))); p.beam_search.beam_size = 3; p.max_tokens = 24
states = [lib.whisper_init_state(ctx) for _ in range(2)]
def toks(st):
    out = []
    for i in range(lib.whisper_full_n_segments_from_state(st)):
        for j in range(lib.whisper_full_n_tokens_from_state(st, i)):
            t = lib.whisper_full_get_token_data_from_state(st, i, j); out.append((t.id, t.tid, round(t.p, 6), round(t.plog, 5), t.t0, t.t1))
    return out
want = [[None] * 4 for _ in states]
for t, st in enumerate(states):
    for i, pcm in enumerate(pcms):
        assert lib.whisper_full_with_state(ctx, st, p, fp(pcm), pcm.size) == 0
        want[t][i] = toks(st)
print("solo: states agree", want[0] == want[1])
got = [[] for _ in states]
def work(t):
    for rep in range(3):
        for i, pcm in enumerate(pcms if t == 0 else pcms[::-1]):
            rc = lib.whisper_full_with_state(ctx, states[t], p, fp(pcm), pcm.size)
            got[t].append((i if t == 0 else 3 - i, toks(states[t])))
th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
for x in th: x.start()
for x in th: x.join()
bad = 0
for t in range(2):
    for i, g in got[t]:
        w = want[t][i]
        if g != w:
            bad += 1
            k = next((k for k in range(min(len(g), len(w))) if g[k] != w[k]), min(len(g), len(w)))
            print(f"thread {t} pcm {i}: first difference at token {k} of {len(w)}/{len(g)}: want {w[k] if k < len(w) else None} got {g[k] if k < len(g) else None}")
print("mismatches", bad, "of", sum(len(x) for x in got))
