"""Scratch (round 6): wmi_full_batch of NB chunks, ms per call, by WMI_LOCKSTEP_GROUPS (one process per setting)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model(os.environ.get("SHAPE", "base.en"), seed=1234))
params = node.full_params("", 0)
for nb in (8, 16, 4):
    pcm = [synth.make_pcm(30.0, seed=100 + i) for i in range(nb)]
    ptrs = (C.c_void_p * nb)(*[p.ctypes.data for p in pcm]); lens = (C.c_int * nb)(*[p.size for p in pcm])
    for _ in range(4): assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 0) == 0
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 0) == 0; ts.append(time.perf_counter() - t0)
    t4 = (C.c_int64 * 4)(); ns = C.c_int32(); lib.wmi_get_batch_timings(node.ctx, t4, C.byref(ns))
    print(f"{os.environ.get('TAG','')} {nb} chunks: median {np.median(ts)*1e3:.3f} ms min {min(ts)*1e3:.3f} | phases (max over groups) mel {t4[0]/1e3:.2f} enc {t4[1]/1e3:.2f} dec {t4[2]/1e3:.2f} emit {t4[3]/1e3:.2f} steps {ns.value}", flush=True)
