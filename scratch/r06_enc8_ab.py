"""Scratch (round 6): 8-chunk (NB) lock-step calls of base.en — encode phase time and the in-situ per-shape GEMM durations, for A/B runs of
one process per environment setting (WMI_ENC_NO_ROWPAD, WMI_GEMM8_QKV, WMI_GEMM_VT_NARROW, ...)."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
import bench
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model(os.environ.get("SHAPE", "base.en"), seed=1234))
params = node.full_params("", 0)
nb = int(os.environ.get("NB", "8"))
pcm = [synth.make_pcm(30.0, seed=100 + i) for i in range(nb)]
ptrs = (C.c_void_p * nb)(*[p.ctypes.data for p in pcm]); lens = (C.c_int * nb)(*[p.size for p in pcm])
enc = []
for _ in range(int(os.environ.get("REPS", "12"))):
    assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 0) == 0
    t = (C.c_int64 * 8)(); n = (C.c_int32 * 8)(); lib.wmi_get_batch_timings(node.ctx, t, n)
    enc.append(t[1] / 1e3)
u = bench.encoder_gemm_utilisation(lib, node.ctx, nb)
tag = os.environ.get("TAG", "")
print("%-28s encode ms (median of %d) %.3f  min %.3f | GEMM agg %.4f (%.1f us)" % (tag, len(enc), float(np.median(enc[2:])), min(enc[2:]), u["frac"], u["gemm_us"]))
for s in u["per_shape"]:
    print("    %-52s M=%6d N=%5d K=%5d  x%d  avg %7.2f us  frac %.3f  wgs %d" % (s["shape"], s["M"], s["N"], s["K"], s["launches"], s["avg_us"], s["frac"], s["workgroups"]))
node.close()
