import ctypes as C, sys, os, time, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import runtime, synth, host
import torch
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("base.en", seed=1234))
params = node.full_params("", 0)
for nb in ((int(os.environ['NB_ONLY']),) if os.environ.get('NB_ONLY') else (8, 16)):
    pcm = [torch.from_numpy(synth.make_pcm(30.0, seed=1234 + i)).cuda() for i in range(nb)]
    ptrs = (C.c_void_p * nb)(*[t.data_ptr() for t in pcm]); lens = (C.c_int * nb)(*[t.numel() for t in pcm])
    for _ in range(3): assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 1) == 0
    t4 = (C.c_int64 * 4)(); ns = C.c_int32(); acc = np.zeros(4)
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 40
    for _ in range(reps):
        assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 1) == 0
        lib.wmi_get_batch_timings(node.ctx, t4, C.byref(ns)); acc += np.array(list(t4), dtype=np.float64)
    dt = (time.perf_counter() - t0) / reps
    print(os.environ.get("WMI_MEL_PER_CHUNK", "batched"), nb, "chunks: %.3f ms per call | mel %.0f enc %.0f dec %.0f emit %.0f us" % (dt * 1e3, *(acc / reps)))
