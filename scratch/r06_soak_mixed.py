"""Scratch: lock-step calls on one context while another context transcribes single chunks on a second thread (the one-launch forms switch
off and on as the device gets company): every result against its reference, status words at the end."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("base.en", seed=1234)
a = host.SpeechToText(lib); a.set_language_model(model)
b = host.SpeechToText(lib); b.set_language_model(model)
pcms = [synth.make_pcm(30.0, seed=1234 + i) for i in range(8)]
ref_b = [[(t["id"], t["p"]) for t in b.transcribe(p, "", 0)[1:]] for p in pcms]
ref_a = [[(t["id"], t["p"]) for t in r[1:]] for r in a.transcribe_batch(pcms, "", 0)]
bad = [0, 0]; n = [0, 0]; stop = [False]
def batch():
    while not stop[0]:
        res = a.transcribe_batch(pcms, "", 0)
        bad[0] += [[(t["id"], t["p"]) for t in r[1:]] for r in res] != ref_a; n[0] += 1
def single():
    i = 0
    while not stop[0]:
        r = b.transcribe(pcms[i % 8], "", 0)
        bad[1] += [(t["id"], t["p"]) for t in r[1:]] != ref_b[i % 8]; n[1] += 1; i += 1
        if i % 50 == 0: time.sleep(0.05)              # gaps: the lock-step thread is alone on the device now and then
ts = [threading.Thread(target=batch), threading.Thread(target=single)]
for t in ts: t.start()
time.sleep(float(os.environ.get("SECS", "20"))); stop[0] = True
for t in ts: t.join()
sa = (C.c_int32 * 3)(); sb = (C.c_int32 * 3)(); lib.wmi_pair_status(a.ctx, sa, 0); lib.wmi_pair_status(b.ctx, sb, 0)
print(f"lock-step calls {n[0]} ({bad[0]} differ), single transcriptions {n[1]} ({bad[1]} differ); status lock-step ctx {list(sa)}, single ctx {list(sb)}")
a.close(); b.close()
