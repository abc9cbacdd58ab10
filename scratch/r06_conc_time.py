"""Scratch (round 6): N threads x whisper_full_with_state (greedy, base.en, host parameter set) on N states of one context:
wall per transcription, with / without the per-device step turn (WMI_NO_STEP_TICKET)."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import abi, host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
model = synth.make_model("base.en", seed=1234)
pcms = [synth.make_pcm(30.0, seed=100 + i) for i in range(8)]
node = host.SpeechToText(lib); node.set_language_model(model); ctx = node.ctx
p = node.full_params("", 0)
for n in (1, 2, 3, 4, 6, 8):
    states = [lib.whisper_init_state(ctx) for _ in range(n)]
    for st in states:
        for i in range(6): assert lib.whisper_full_with_state(ctx, st, p, fp(pcms[i]), pcms[i].size) == 0
    reps = 40
    def work(t):
        for r in range(reps):
            pcm = pcms[(t + r) % 8]
            lib.whisper_full_with_state(ctx, states[t], p, fp(pcm), pcm.size)
    th = [threading.Thread(target=work, args=(t,)) for t in range(n)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    wall = time.perf_counter() - t0
    print(f"{os.environ.get('TAG','')} threads {n}: {wall / reps * 1e3:.2f} ms per round of {n} transcriptions = {wall / (reps * n) * 1e3:.2f} ms per transcription", flush=True)
    for st in states: lib.whisper_free_state(st)
