"""Scratch: in-kernel stamps of the greedy step of a block-quantised model (large-v3 q5_1 by default): per launch body, gap,
'rows quantised' and 'tiles multiplied' marks."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
entry.load_package(); entry.load_oracle()
from godot_whisper_amd import host, runtime, synth
from oracle import reflib
import test_gpu_large_v3 as tl
lib = runtime.require_gpu(); runtime.silence_logs(lib)
shape = os.environ.get("SHAPE", "large-v3"); qt = os.environ.get("QT", "q5_1")
m = synth.make_model(shape, seed=2024)
if qt != 'f16': m = tl._ref_quantize_model(reflib.lib(), m, qt) if reflib.available() else synth.quantize_model(m, qt)
node = host.SpeechToText(lib); node.set_language_model(m); node.language = "en"
pcm = synth.make_pcm(30.0, seed=7)
for _ in range(2): node.transcribe(pcm, "", 0)
cap = 600
buf = (C.c_double * (6 * cap))()
n = lib.wmi_step_stamps(node.ctx, buf, cap, int(os.environ.get('CHAINED', '0')))
rows = [(buf[6 * i], buf[6 * i + 1], buf[6 * i + 2], int(buf[6 * i + 3]), buf[6 * i + 4], buf[6 * i + 5]) for i in range(max(n, 0))]
rows = [r for r in rows if r[3] > 0]
print(len(rows), "stamped launches; span %.1f us" % (rows[-1][2] - rows[0][0]))
for i, (s0, s1, e1, cnt, m1, m2) in enumerate(rows[:int(os.environ.get('NROWS', '30'))]):
    print("  launch %3d: start %8.2f  last-start +%5.2f  body %5.2f  waves %4d  mark1 +%6.2f  mark2 +%6.2f  gap-before %5.2f" % (i, s0 - rows[0][0], s1 - s0, e1 - s0, cnt, (m1 - s0) if m1 > 0 else -1, (m2 - s0) if m2 > 0 else -1, (s0 - rows[i - 1][2]) if i else 0.0))
node.close()
