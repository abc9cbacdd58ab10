import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("base.en", seed=1234)
buf = C.create_string_buffer(model, len(model))
ctx = lib.wmi_init_from_buffer_on_device(C.cast(buf, C.c_void_p), len(model), 0)
us = lib.wmi_bench_kernel(ctx, 1, 300)
print(os.environ.get("WMI_LOGITS_RIF"), os.environ.get("WMI_LOGITS_BLOCKS"), os.environ.get("WMI_LOGITS_NT"), "logits gemv us", round(us, 2), "GB/s", round(51864*512*2/us/1e3, 1))
