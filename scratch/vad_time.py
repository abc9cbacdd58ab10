"""Scratch: wmi_vad on a device-resident 3 s window, ms per call."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("micro.en", seed=1))
hip = C.CDLL("libamdhip64.so")
pcm = synth.make_pcm(3.0, seed=4)
d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), C.c_size_t(pcm.nbytes)) == 0
assert hip.hipMemcpy(d, pcm.ctypes.data_as(C.c_void_p), C.c_size_t(pcm.nbytes), 1) == 0
for _ in range(3): lib.wmi_vad(node.ctx, d, int(pcm.size), 1, 2.0, 200.0, None)
t0 = time.perf_counter()
for _ in range(20): lib.wmi_vad(node.ctx, d, int(pcm.size), 1, 2.0, 200.0, None)
print("wmi_vad, device-resident 3 s window: %.3f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
node.close()
