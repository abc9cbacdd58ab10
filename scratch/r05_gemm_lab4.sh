#!/bin/bash
# round 5, GEMM lab call 4: q|k|v with its own epilogue on 288-row tiles (orientation per column tile) vs k_gemm; then the product in situ
cd scratch/lab
L=./gemm8_lab
O=../../gpurun_out/r05_gemm8_d.txt
{
for rep in 1 2; do
LAB_SET=288:32,192:64 timeout 120 $L 8
done
LAB_SET=192:64 timeout 60 $L 0
} > $O 2>&1
grep -v "^$" $O | tail -20
cd ../..
python - <<'PY'
import ctypes as C, json, sys, numpy as np
sys.argv=["bench.py"]
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import runtime, synth, host
import bench, torch
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("base.en", seed=1234)
node = host.SpeechToText(lib); node.set_language_model(model)
params = node.full_params("", 0)
pcm = [torch.from_numpy(synth.make_pcm(30.0, seed=1234 + i)).cuda() for i in range(8)]
ptrs = (C.c_void_p * 8)(*[t.data_ptr() for t in pcm]); lens = (C.c_int * 8)(*[t.numel() for t in pcm])
for _ in range(3): assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, 8, 1) == 0
import os
for g8 in ("1",):
    u = bench.encoder_gemm_utilisation(lib, node.ctx, 8, 2030.0)
    print("batch8 aggregate", u["achieved"], u["frac"], u["gemm_us"])
    for p in u["per_shape"]: print("   ", p["shape"], p["avg_us"], p["tflops"], p["kernel"], p["workgroups"])
t4 = (C.c_int64 * 4)(); ns = C.c_int32()
acc = np.zeros(4)
for _ in range(20):
    assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, 8, 1) == 0
    lib.wmi_get_batch_timings(node.ctx, t4, C.byref(ns)); acc += np.array(list(t4), dtype=np.float64)
print("batch8 timings (us): mel, encode, decode, emit =", acc / 20)
PY
