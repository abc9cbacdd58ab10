#!/bin/bash
# Decode-step chain on the GPU (wmi_bench_kernel 20, no host in the loop): graph replay vs eager launches, and HIP runtime knobs on the replay
run() { echo -n "$*: "; env "$@" python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("base.en", seed=1234))
pcm = synth.make_pcm(30.0, seed=1234)
for _ in range(3): node.transcribe(pcm, "", 0)
lib.wmi_bench_kernel.restype = C.c_double
print("%.1f us per step" % lib.wmi_bench_kernel(node.ctx, 20, 200), " touch chains %.2f %.2f" % (lib.wmi_bench_kernel(node.ctx, 10, 500), lib.wmi_bench_kernel(node.ctx, 12, 500)))
PY
}
run X=0
run WMI_NO_GRAPH=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run AMD_OPT_FLUSH=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run HIP_FORCE_DEV_KERNARG=0
run ROC_USE_FGS_KERNARG=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000
