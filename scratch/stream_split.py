#!/usr/bin/env python3
"""configs[2] (small multilingual, streaming node): where a call's time goes — mel / encode / decode steps / prompt / sampling — per call."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
import numpy as np
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.CaptureStreamToText(lib, transcribe_interval=0.3); node.language = "de"
node.set_language_model(synth.make_model("small", seed=77))
pcm = synth.make_pcm(30.0, seed=21, gate=True)
list(node.stream(pcm[: 16000 * 3]))
rows = []
inner = node.transcribe
def timed(buffer, initial_prompt="", audio_ctx=0, params=None):
    lib.whisper_reset_timings(node.ctx)
    t = time.perf_counter(); r = inner(buffer, initial_prompt, audio_ctx, params); e = time.perf_counter() - t
    t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)(); lib.wmi_get_timings(node.ctx, t6, n5)
    rows.append((e * 1e3, audio_ctx, len(r) - 1, [x / 1e3 for x in t6], list(n5)))
    return r
node.transcribe = timed
for _ in node.stream(pcm): pass
rows.sort(key=lambda x: x[0])
med = rows[len(rows) // 2]
print("calls", len(rows), "median call: %.2f ms, audio_ctx %d, tokens %d | mel %.2f enc %.2f dec %.2f batchd %.2f prompt %.2f sample %.2f | counts %s" % (med[0], med[1], med[2], *med[3], med[4]))
tot = np.array([r[3] for r in rows]).sum(0); wall = sum(r[0] for r in rows)
print("all calls: wall %.1f ms | mel %.1f enc %.1f dec %.1f batchd %.1f prompt %.1f sample %.1f | tokens %d, decode calls %d" % (wall, *tot, sum(r[2] for r in rows), sum(r[4][1] for r in rows)))
for r in rows[-3:]: print("slowest: %.2f ms ctx %d tokens %d" % (r[0], r[1], r[2]), [round(x, 2) for x in r[3]], r[4])
node.close()
