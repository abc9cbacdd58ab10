"""Scratch: the cross-attention of the block-quantised models with the query projection inside (k_xattn_fused_q) against the two-launch
form (WMI_Q_XATTN_TWO_LAUNCHES=1): results (ids, p, plog, token times of greedy, beam and lock-step calls) as one JSON line, and timings.
Run once per setting and compare the RESULT lines (scratch/r06_qx_ab.sh)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
entry.load_package(); entry.load_oracle()
from godot_whisper_amd import host, runtime, synth
from oracle import reflib
import test_gpu_large_v3 as tl
lib = runtime.require_gpu(); runtime.silence_logs(lib)
out = {}
cases = os.environ.get("CASES", "large-v3:q5_1,base.en:q4_0,small:q8_0,tiny.en:q5_0,medium:q4_1").split(",")
for case in cases:
    shape, qt = case.split(":")
    m = synth.make_model(shape, seed=2024)
    m = tl._ref_quantize_model(reflib.lib(), m, qt) if reflib.available() else synth.quantize_model(m, qt)
    node = host.SpeechToText(lib); node.set_language_model(m); node.language = "en" if shape.endswith(".en") or shape.startswith("large") else "de"
    pcm = synth.make_pcm(30.0, seed=7)
    res = {}
    for name, strat, bs, mt in (("greedy", 0, 1, 16), ("beam3", 1, 3, 12)):
        p = lib.whisper_full_default_params(strat); q = node.full_params("", 0)
        for f in ("language", "audio_ctx", "split_on_word", "token_timestamps", "suppress_non_speech_tokens", "single_segment", "entropy_thold", "initial_prompt"):
            setattr(p, f, getattr(q, f))
        p.max_tokens = mt; p.temperature_inc = 0.0
        if strat == 1: p.beam_search.beam_size = bs
        rr = []
        t0 = time.perf_counter()
        for rep in range(3):
            r = node.transcribe(pcm, params=p)
            rr.append([[int(t["id"]), int(t["tid"]), float(t["p"]), float(t["plog"]), int(t["t0"]), int(t["t1"])] for t in r[1:]])
        res[name] = rr
        print(f"{case} {name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per transcription, {len(rr[0])} tokens", flush=True)
    pcms = [synth.make_pcm(30.0, seed=50 + i) for i in range(4)]
    p = node.full_params("", 0); p.temperature_inc = 0.0
    r = node.transcribe_batch(pcms, params=p)
    res["lockstep4"] = [[[int(t["id"]), float(t["p"]), float(t["plog"])] for t in one[1:]] for one in r]
    print(case, "greedy step chain on the GPU, us per step:", lib.wmi_bench_kernel(node.ctx, 20, 20), flush=True)
    out[case] = res
    node.close()
print("RESULT" + json.dumps(out))
