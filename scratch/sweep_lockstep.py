# random compositions of chunks through wmi_full_batch vs one-at-a-time on fresh contexts (exact mode: bit-identical; MFMA mode: up to near-ties)
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import runtime, host, synth
import test_gpu_parity as T
lib = runtime.require_gpu(); runtime.silence_logs(lib)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for case in range(lo, hi):
    rng = np.random.default_rng(5000 + case)
    shape = "micro.en" if case % 2 == 0 else "micro"
    model = synth.make_model(shape, seed=int(rng.integers(1, 10**6)))
    nch = int(rng.integers(2, 20))
    secs = [float(rng.choice([0.4, 1.5, 4.0, 11.0, 22.0, 30.0, 30.0, 47.0])) for _ in range(nch)]
    pcms = [synth.make_pcm(s, seed=int(rng.integers(1, 10**6)), gate=bool(rng.integers(0, 2))) for s in secs]
    actx = 0 if rng.integers(0, 2) else 1100
    prompt = "" if rng.integers(0, 2) else " Well, then."
    exact = bool(case % 3 == 0)
    lib.wmi_set_lockstep_exact(1 if exact else 0)
    def params(node):
        p = node.full_params(prompt, actx)
        if not exact or shape == "micro" or prompt: p.temperature_inc = 0.0
        return p
    want = []
    for b in pcms:
        node = host.SpeechToText(lib); node.set_language_model(model); want.append(node.transcribe(b, params=params(node))); node.close()
    node = host.SpeechToText(lib); node.set_language_model(model)
    got = node.transcribe_batch(pcms, params=params(node)); modes = list(node.last_modes); node.close()
    try:
        assert len(got) == len(want)
        for c, (g, w) in enumerate(zip(got, want)):
            if not w or len(w) <= 1:
                assert not g or len(g) <= 1, (c, g)
                continue
            strict = modes[c] == 1 or (exact and shape == "micro.en" and not prompt and secs[c] <= 30.0)
            T._assert_same_transcription(g, w, (case, c, secs[c], modes[c]), strict)
    except AssertionError as e:
        bad += 1; print(f"case {case} ({shape}, n={nch}, exact={exact}, actx={actx}, prompt={bool(prompt)}): {str(e)[:400]}")
lib.wmi_set_lockstep_exact(0)
print("cases", lo, "..", hi - 1, "failures", bad)
