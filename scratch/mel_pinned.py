#!/usr/bin/env python3
"""Lock-step call, 8 chunks: mel + envelope phase with the caller's PCM in pageable memory (numpy) against pinned memory (torch pin_memory)."""
import ctypes as C, os, sys, time
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
import numpy as np, torch
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("base.en", seed=1234)); node.language = "en"
nb = 8
pcms = [synth.make_pcm(30.0, seed=100 + i) for i in range(nb)]
pinned = [torch.from_numpy(p.copy()).pin_memory() for p in pcms]
p = node.full_params("", 0); p.temperature_inc = 0.0
def run(bufs, label, reps=30):
    ptrs = (C.c_void_p * nb)(*bufs); lens = (C.c_int * nb)(*[480000] * nb)
    for _ in range(5): assert lib.wmi_full_batch(node.ctx, p, ptrs, lens, nb, 0) == 0
    acc = [0.0] * 4; t0 = time.perf_counter()
    for _ in range(reps):
        assert lib.wmi_full_batch(node.ctx, p, ptrs, lens, nb, 0) == 0
        t4 = (C.c_int64 * 4)(); ns = C.c_int32(); lib.wmi_get_batch_timings(node.ctx, t4, C.byref(ns))
        for i in range(4): acc[i] += t4[i]
    dt = (time.perf_counter() - t0) / reps
    print(f"{label}: {dt*1e3:.3f} ms per call | mel+envelope {acc[0]/reps/1e3:.3f} encode {acc[1]/reps/1e3:.3f} decode {acc[2]/reps/1e3:.3f} emit {acc[3]/reps/1e3:.3f}", flush=True)
for _ in range(2):
    run([b.ctypes.data for b in pcms], "pageable PCM")
    run([t.data_ptr() for t in pinned], "pinned PCM  ")
