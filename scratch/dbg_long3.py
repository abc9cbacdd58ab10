# replay window 1 of the long-audio case token by token through both libraries' decode + logit filters and report where the
# filtered arg-max first differs and how close the two candidates are on each side
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import ctypes as C
import numpy as np
import __graft_entry__ as ge
ge.load_package(); ge.load_oracle()
from godot_whisper_amd import host, runtime, synth, abi
from oracle import reflib
import stage_compare as sc
lib = runtime.require_gpu(); runtime.silence_logs(lib)
ref = reflib.lib()
cb = abi.ggml_log_callback(lambda lvl, txt, ud: None); ref.whisper_log_set(C.cast(cb, C.c_void_p), None)
model = synth.make_model("micro.en", seed=91); pcm = synth.make_pcm(150.0, seed=92, gate=True)
prod = sc.ProductSide(lib, model); chk = sc.RefSide(ref, model)
prod.mel(pcm); chk.mel(pcm); prod.encode(0, 0); chk.encode(0, 0)
nodeP = host.SpeechToText(lib); nodeP.ctx = prod.ctx
nodeR = host.SpeechToText(ref); nodeR.ctx = chk.ctx
def params(L):
    p = L.whisper_full_default_params(abi.WHISPER_SAMPLING_GREEDY); p.language = b"en"; p.temperature_inc = 0.0; return p
nv = lib.whisper_n_vocab(prod.ctx); beg = lib.whisper_token_beg(prod.ctx); sot = lib.whisper_token_sot(prod.ctx)
hist = []; has_ts = 0; seek_delta = 3000
toks = [sot]
lp_raw = prod.decode(toks, 0); lr_raw = chk.decode(toks, 0)
for step in range(130):
    outs = []
    for L, ctx, fn, raw in ((lib, prod.ctx, lib.wmi_process_logits, lp_raw), (ref, chk.ctx, ref.ref_process_logits, lr_raw)):
        h = np.asarray(hist, np.int32)
        lo, lpb, pr = (np.empty(nv, np.float32) for _ in range(3))
        fn(ctx, params(L), sc._fptr(np.ascontiguousarray(raw)), h.ctypes.data_as(C.POINTER(C.c_int32)), h.size, has_ts, seek_delta, C.c_float(0.0), sc._fptr(lo), sc._fptr(lpb), sc._fptr(pr))
        outs.append(pr.copy())
    ip, ir = int(np.argmax(outs[0])), int(np.argmax(outs[1]))
    if ip != ir:
        print(f"step {step}: product picks {ip} (p {outs[0][ip]:.4f}, ref gives it {outs[1][ip]:.4f}) ; reference picks {ir} (p {outs[1][ir]:.4f}, product gives it {outs[0][ir]:.4f})")
        break
    tok = ir
    hist.append(tok)
    if tok > beg:
        has_ts = 1; seek_delta = 2 * (tok - beg)
    lp_raw = prod.decode([tok], 1 + step); lr_raw = chk.decode([tok], 1 + step)
else:
    print("no divergence in 130 steps")
print("steps replayed", len(hist), "last tokens", hist[-5:])
