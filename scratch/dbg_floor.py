import sys
sys.path.insert(0, "/root/repo")
import __graft_entry__ as e
e.load_package()
from godot_whisper_amd import runtime, synth
import ctypes as C
lib = runtime.require_gpu(); runtime.silence_logs(lib)
mb = synth.make_model("micro.en", seed=1)
buf = C.create_string_buffer(mb, len(mb))
ctx = lib.wmi_init_from_buffer_on_device(C.cast(buf, C.c_void_p), len(mb), 0)
for which, name in ((10, "1 block"), (11, "32 blocks"), (12, "256 blocks")):
    print(name, "trivial dependent kernel chain: %.2f us per kernel" % lib.wmi_bench_kernel(ctx, which, 2000))
