"""Scratch: soak of the one-launch forms of the decode step (k_front, k_xback, k_mlp_pair): N transcriptions of varying chunks on one
context, then lock-step calls; every result compared with the first of its kind, wmi_pair_status at the end (re-runs and slow hand-offs
must stay 0 on an otherwise idle device)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model(os.environ.get("SHAPE", "base.en"), seed=1234))
pcms = [synth.make_pcm(30.0, seed=1234 + i) for i in range(8)]
ref = {}
n = int(os.environ.get("N", "3000")); bad = 0
t0 = time.time()
for i in range(n):
    r = node.transcribe(pcms[i % 8], "", 0)
    key = [(t["id"], t["p"]) for t in r[1:]]
    if i % 8 in ref:
        bad += key != ref[i % 8]
    else: ref[i % 8] = key
st = (C.c_int32 * 3)(); lib.wmi_pair_status(node.ctx, st, 0)
print(f"{n} transcriptions in {time.time() - t0:.1f} s: {bad} differ from their first run; status (re-runs, slow, switches) = {list(st)}")
refb = None; badb = 0; nb = int(os.environ.get("NBATCH", "300"))
t0 = time.time()
for i in range(nb):
    res = node.transcribe_batch(pcms, "", 0)
    key = [[(t["id"], t["p"]) for t in r[1:]] for r in res]
    if refb is None: refb = key
    else: badb += key != refb
lib.wmi_pair_status(node.ctx, st, 0)
print(f"{nb} lock-step calls of 8 chunks in {time.time() - t0:.1f} s: {badb} differ from the first; status = {list(st)}")
node.close()
