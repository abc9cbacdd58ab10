"""-m gpu: launch-structure variants of the greedy decode step give the same transcription.

The step exists in several forms that must not change a single bit of the result: captured graphs vs eager launches
(WMI_NO_GRAPH), chained steps (the pick kernel prepares the next step on the device) vs an embedding launch per step
(WMI_NO_CHAIN).  The switches are read once per process, so every variant runs in its own interpreter.  The two-launch
cross-attention (WMI_XATTN_TWO_PASS: global maximum before the f16 exponent) is a different arithmetic and is compared
margin-aware, like every other f32-order difference."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import json, sys
sys.path.insert(0, ROOT_PLACEHOLDER)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
out = {}
for shape, secs, mt in (("base.en", 30.0, 16), ("micro.en", 41.0, 0), ("micro", 30.0, 0)):
    node = host.SpeechToText(lib); node.set_language_model(synth.make_model(shape, seed=4242))
    if shape == "micro": node.language = "de"
    res = []
    for rep in range(6):                               # enough steps for the graphs to be captured (> 64 per form) and replayed
        p = node.full_params("", 0); p.max_tokens = mt; p.temperature_inc = 0.0
        r = node.transcribe(synth.make_pcm(secs, seed=900 + rep % 2), params=p)
        res.append([[int(t["id"]), int(t["tid"]), float(t["p"]), float(t["plog"]), int(t["t0"]), int(t["t1"])] for t in r[1:]])
    out[shape] = res
    node.close()
print("RESULT" + json.dumps(out))
""".replace("ROOT_PLACEHOLDER", repr(ROOT))


def _run(env_extra):
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", _SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


@pytest.fixture(scope="module")
def default_run():
    return _run({})


@pytest.mark.parametrize("knob", ["WMI_NO_CHAIN", "WMI_NO_GRAPH"])
def test_step_forms_are_bit_identical(default_run, knob):
    other = _run({knob: "1"})
    for shape, runs in default_run.items():
        assert len(runs[0]) > 0, shape
        for a, b in zip(runs, other[shape]):
            assert a == b, (knob, shape)                # ids, tids, probabilities (exact f32 values) and token times


def test_two_pass_cross_attention_agrees_within_the_margin(default_run):
    other = _run({"WMI_XATTN_TWO_PASS": "1"})
    for shape, runs in default_run.items():
        for a, b in zip(runs, other[shape]):
            n = min(len(a), len(b))
            ids_a, ids_b = [t[0] for t in a[:n]], [t[0] for t in b[:n]]
            first = next((i for i in range(n) if ids_a[i] != ids_b[i]), n)
            if first < n:                               # a near-tie of the synthetic weights: both picks carry about the same probability
                assert abs(a[first][2] - b[first][2]) <= 2e-2, (shape, first, a[first], b[first])
            if first:
                pa, pb = np.array([t[2] for t in a[:first]]), np.array([t[2] for t in b[:first]])
                assert np.abs(pa - pb).max() <= 1e-2, shape
