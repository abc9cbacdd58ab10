#!/usr/bin/env python3
"""What in bench.py's earlier legs makes the replica contexts of the configs[4] leg run one after the other?  modes: ctx | full | batch16 | dsp"""
import sys, json, ctypes as C
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
mode = sys.argv[1] if len(sys.argv) > 1 else "ctx"
import torch; torch.cuda.init(); torch.zeros(4, device="cuda")
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
import bench
n0 = host.SpeechToText(lib); n0.set_language_model(synth.make_model("base.en", seed=1234)); n0.language = "en"
g0 = n0.full_params("", 0); g0.temperature_inc = 0.0
if "full" in mode:
    pcm = synth.make_pcm(30.0, seed=1)
    for _ in range(80): n0.transcribe(pcm, params=g0)
import re
m = re.search(r"batch(\d+)x(\d+)", mode)
if m:
    nb, reps = int(m.group(1)), int(m.group(2))
    for _ in range(reps): n0.transcribe_batch([synth.make_pcm(30.0, seed=10 + i) for i in range(nb)], params=g0)
if "dsp" in mode:
    bench.host_dsp_config(lib, n0.ctx, cpu=False)
if 'keep' not in mode: n0.close()
c = bench.config4(lib, cpu=False)
print(mode, {k: c[k].get("ms_per_chunk") for k in ("beam5_8chunks_one_at_a_time", "beam5_8chunks_replicas") if k in c} or c)
