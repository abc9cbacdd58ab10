import sys, os, numpy as np, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as e
e.load_package(); e.load_oracle()
from godot_whisper_amd import runtime, synth, host
from oracle import port
import golden_util as gu, stage_compare as sc
lib = runtime.require_gpu(); runtime.silence_logs(lib)
G = np.load(gu.GOLDEN / "hotpath.npz")
want = G["en30/full_default_greedy/tokens"]
model, pcm, actx = gu.case_inputs("en30")
prod = sc.ProductSide(lib, model); chk = port.PortSide(model)
prod.mel(pcm); chk.mel(pcm); chk.encode(0, 0); prod.encode(0, 0)
node = host.SpeechToText(lib); node.ctx = prod.ctx
p = gu.param_variants(node)["default_greedy"]
nv = prod.NV
def filt(raw, hist, has_ts, sd):
    lo, lp, pr = (np.empty(nv, np.float32) for _ in range(3))
    h = np.asarray(hist, np.int32)
    lib.wmi_process_logits(prod.ctx, p, sc._fptr(np.ascontiguousarray(raw)), h.ctypes.data_as(C.POINTER(C.c_int32)), h.size, has_ts, sd, C.c_float(0.0), sc._fptr(lo), sc._fptr(lp), sc._fptr(pr))
    return pr
lr = chk.decode([50257], 0); lp = prod.decode([50257], 0)
hist = []; has_ts = 0; sd = 3000; beg = 50363
for i in range(0, 110):
    prr = filt(lr, hist, has_ts, sd); prp = filt(lp, hist, has_ts, sd)
    ir, ip = int(np.argmax(prr)), int(np.argmax(prp))
    if i >= 103 or ir != ip or ir != int(want[i,0]):
        top = np.argsort(-prr)[:3]
        print(i, "ref-logits choice", ir, round(float(prr[ir]),4), "prod-logits choice", ip, round(float(prp[ip]),4), "golden", int(want[i,0]), round(want[i,2],4), "top3", [(int(t), round(float(prr[t]),4)) for t in top])
    tok = int(want[i, 0]); hist.append(tok)
    if tok > beg: has_ts = 1; sd = 2*(tok-beg)
    lr = chk.decode([tok], 1 + i); lp = prod.decode([tok], 1 + i)
