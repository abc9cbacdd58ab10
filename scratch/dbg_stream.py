import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as e
e.load_package(); e.load_oracle()
from godot_whisper_amd import runtime, synth, host
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
G = np.load(gu.GOLDEN / "hotpath.npz")
model, pcm, actx = gu.case_inputs("en30")
node = host.SpeechToText(lib); node.set_language_model(model)
p = gu.param_variants(node)["default_greedy"]
got = gu.tokens_array(node.transcribe(pcm, params=p)); want = G["en30/full_default_greedy/tokens"]
n = min(len(got), len(want)); same = got[:n,0]==want[:n,0]
first = n if same.all() else int(np.argmin(same))
print(os.environ.get("WMI_XATTN_SINGLE"), "len", len(got), len(want), "first mismatch", first)
for i in range(max(0, first-3), min(n, first+4)):
    print(i, "got", got[i,0], round(got[i,2],4), "want", want[i,0], round(want[i,2],4))
