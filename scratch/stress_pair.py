"""Scratch: the one-launch MLP under concurrency — N contexts on N host threads transcribe the same chunks at once (each launch's workgroups
wait for each other inside the launch: all of them have to become resident beside the other contexts' launches).  Every result must equal the
single-context result; prints the wall time per transcription."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
shape = os.environ.get("SHAPE", "base.en"); NCTX = int(os.environ.get("NCTX", "6")); REPS = int(os.environ.get("REPS", "150"))
model = synth.make_model(shape, seed=1234)
pcms = [synth.make_pcm(30.0, seed=1234 + i) for i in range(4)]
def toks(node, pcm):
    r = node.transcribe(pcm, "", 0)
    return [(t["id"], t["p"], t["t0"], t["t1"]) for t in r[1:]]
ref_node = host.SpeechToText(lib); ref_node.set_language_model(model)
if not shape.endswith(".en"): ref_node.language = "de"
ref = [toks(ref_node, p) for p in pcms]
nodes = []
for i in range(NCTX):
    n = host.SpeechToText(lib); n.set_language_model(model)
    if not shape.endswith(".en"): n.language = "de"
    nodes.append(n)
bad = [0] * NCTX
sys.setswitchinterval(1e-4)
def work(i):
    # whisper_full directly (the GIL is released for the whole call); the result is read back and compared every 25th time only —
    # per-token ctypes calls on several threads measure the interpreter lock, not the GPU
    node = nodes[i]; p = node.full_params("", 0)
    for rep in range(REPS):
        k = (rep + i) % 4
        ret = lib.whisper_full(node.ctx, p, pcms[k].ctypes.data_as(C.POINTER(C.c_float)), pcms[k].size)
        if ret != 0: bad[i] += 1
        elif rep % 25 == 0 and [(t["id"], t["p"], t["t0"], t["t1"]) for t in node.collect()[1:]] != ref[k]: bad[i] += 1
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(NCTX)]
[t.start() for t in th]; [t.join() for t in th]
dt = time.perf_counter() - t0
print(f"{shape}: {NCTX} contexts x {REPS} transcriptions at once: {sum(bad)} mismatches, {dt / REPS * 1e3:.2f} ms per round of {NCTX} ({dt / REPS / NCTX * 1e3:.2f} ms per transcription)", flush=True)
for n in nodes: n.close()
ref_node.close()
