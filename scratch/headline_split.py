"""Scratch: where the headline call's wall time goes (base.en, one chunk): variants of the host parameter set."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("base.en", seed=1234))
pcm = [synth.make_pcm(30.0, seed=1234 + i) for i in range(8)]
def run(name, **kw):
    p = node.full_params("", 0)
    for k, v in kw.items(): setattr(p, k, v)
    for i in range(30): lib.whisper_full(node.ctx, p, pcm[i % 8].ctypes.data_as(C.POINTER(C.c_float)), pcm[i % 8].size)
    lib.whisper_reset_timings(node.ctx)
    n = 300; t0 = time.perf_counter()
    for i in range(n): lib.whisper_full(node.ctx, p, pcm[i % 8].ctypes.data_as(C.POINTER(C.c_float)), pcm[i % 8].size)
    dt = (time.perf_counter() - t0) / n
    t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)(); lib.wmi_get_timings(node.ctx, t6, n5)
    ntok = sum(lib.whisper_full_n_tokens(node.ctx, s) for s in range(lib.whisper_full_n_segments(node.ctx)))
    mel, enc, dec = t6[0] / n / 1e3, t6[1] / n / 1e3, (t6[2] + t6[3] + t6[4]) / n / 1e3
    print(f"{name:34s} {dt*1e3:7.3f} ms | mel {mel:.3f} enc {enc:.3f} dec {dec:.3f} ({n5[1]//n} calls) sample {t6[5]/n/1e3:.3f} | other {dt*1e3-mel-enc-dec-t6[5]/n/1e3:.3f} | tokens {ntok}")
run("host params")
run("token_timestamps = false", token_timestamps=False)
run("max_tokens = 1", max_tokens=1)
run("max_tokens = 1, no timestamps", max_tokens=1, token_timestamps=False)
run("max_tokens = 8", max_tokens=8)
node.close()
