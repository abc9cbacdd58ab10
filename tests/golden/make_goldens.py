#!/usr/bin/env python3
"""Generate tests/golden/hotpath.npz from the reference's own CPU path (oracle/_ref/libwhisper_ref.so,
compiled from /root/reference by `make -C oracle ref`).  Run in the build container only:

    python tests/golden/make_goldens.py

What is stored are DATA: inputs are regenerated from seeds (godot-whisper_amd/synth.py) or read from the
reference's sample audio (jfk.wav, a data file the reference's ctest uses, W/tests/CMakeLists.txt:15-76);
outputs are compact summaries (moments + strided samples + top-k) of the reference's tensors and the
complete token streams of whisper_full under several parameter sets.
"""
import ctypes as C
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as entry  # noqa: E402

entry.load_package(); entry.load_oracle()
import golden_util as gu  # noqa: E402
import stage_compare as sc  # noqa: E402
from godot_whisper_amd import abi, host  # noqa: E402
from oracle import reflib  # noqa: E402


def main():
    lib = reflib.lib()
    cb = abi.ggml_log_callback(lambda lvl, txt, ud: None)
    lib.whisper_log_set(C.cast(cb, C.c_void_p), None)
    out = {}
    for name in gu.CASES:
        model, pcm, actx = gu.case_inputs(name)
        ref = sc.RefSide(lib, model)
        mel, n_org = ref.mel(pcm)
        out[f"{name}/mel_shape"] = np.asarray(list(mel.shape) + [n_org], np.int64)
        gu.flatten(f"{name}/mel", gu.summary(mel), out)
        enc = ref.encode(0, actx)
        for k in ("embd_conv", "embd_enc", "cross_k", "cross_v"):
            gu.flatten(f"{name}/{k}", gu.summary(enc[k]), out)
        sot = lib.whisper_token_sot(ref.ctx)
        prompt = [sot] if not lib.whisper_is_multilingual(ref.ctx) else [sot, lib.whisper_token_lang(ref.ctx, 0), lib.whisper_token_transcribe(ref.ctx)]
        lg = ref.decode(prompt, 0)
        gu.flatten(f"{name}/logits_prompt", gu.logits_summary(lg), out)
        fed = []
        for i in range(4):
            nxt = int(np.argmax(lg[:50256])); fed.append(nxt)
            lg = ref.decode([nxt], len(prompt) + i)
            gu.flatten(f"{name}/logits_step{i}", gu.logits_summary(lg), out)
        out[f"{name}/fed_tokens"] = np.asarray(fed, np.int32)
        many = prompt + [int(x) for x in (np.arange(11) * 997 + 1000)]
        gu.flatten(f"{name}/logits_batch", gu.logits_summary(ref.decode(many, 0)), out)
        ref.close()
        # whisper_full through the host mirror
        node = host.SpeechToText(lib); node.set_language_model(model)
        for vname, p in gu.param_variants(node).items():
            p.audio_ctx = actx
            r = node.transcribe(pcm, params=p)
            out[f"{name}/full_{vname}/ret"] = np.int64(node.last_ret)
            out[f"{name}/full_{vname}/tokens"] = gu.tokens_array(r) if r else np.zeros((0, 9))
            out[f"{name}/full_{vname}/text"] = np.frombuffer(r[0] if r else b"", np.uint8)
            out[f"{name}/full_{vname}/n_segments"] = np.int64(lib.whisper_full_n_segments(node.ctx))
        # tokenizer
        if name == "en30":
            buf = (C.c_int32 * 1024)()
            for i, text in enumerate(gu.PROMPTS):
                n = lib.whisper_tokenize(node.ctx, text.encode("utf-8"), buf, 1024)
                out[f"tokenize/{i}"] = np.asarray(list(buf[:max(n, 0)]), np.int32)
        # host logic: logit filters and sampling on seeded raw logits
        if name in ("en30", "ml11"):
            nv = lib.whisper_n_vocab(node.ctx)
            beg = lib.whisper_token_beg(node.ctx)
            rng = np.random.default_rng(5)
            hist_cases = [([], 0, 3000), ([100, 200], 0, 3000), ([beg + 10], 1, 20), ([300, beg + 10], 1, 20),
                          ([beg + 5, beg + 5], 1, 10), ([400] * 3, 0, 3000)]
            for ci, (hist, has_ts, sd) in enumerate(hist_cases):
                raw = (rng.standard_normal(nv) * 6.0).astype(np.float32)
                if ci == 2:
                    raw[beg:] += 9.0            # force the "timestamp mass beats text" branch
                for temp in (0.0, 0.6):
                    p = node.full_params("", 0)
                    lo, lp, pr = (np.empty(nv, np.float32) for _ in range(3))
                    h = np.asarray(hist, np.int32)
                    lib.ref_process_logits(node.ctx, p, sc._fptr(raw), h.ctypes.data_as(C.POINTER(C.c_int32)), h.size, has_ts, sd,
                                           C.c_float(temp), sc._fptr(lo), sc._fptr(lp), sc._fptr(pr))
                    key = f"{name}/filters/{ci}_t{temp}"
                    out[key + "/n_neg_inf"] = np.int64(np.isneginf(lo).sum())
                    out[key + "/neg_inf_hash"] = np.int64(np.flatnonzero(np.isneginf(lo)).astype(np.int64).sum())
                    fin = np.isfinite(lp)
                    out[key + "/logprob_sum"] = np.float64(lp[fin].astype(np.float64).sum())
                    out[key + "/prob_sum"] = np.float64(pr.astype(np.float64).sum())
                    top = np.argsort(-pr, kind="stable")[:8].astype(np.int32)
                    out[key + "/top_ids"] = top; out[key + "/top_probs"] = pr[top].copy(); out[key + "/top_logprobs"] = lp[top].copy()
                    if temp > 0:
                        draws = (abi.whisper_token_data * 12)()
                        lib.ref_sample_draws(node.ctx, sc._fptr(pr), sc._fptr(lp), 12, 1, draws)
                        out[key + "/draws"] = np.asarray([[d.id, d.tid] for d in draws], np.int32)
                        out[key + "/draw_stats"] = np.asarray([[d.p, d.plog, d.pt, d.ptsum] for d in draws], np.float64)
        node.close()
    # streaming node (BASELINE config 3 call pattern): growing buffer re-transcribed on a cadence with
    # audio_ctx = total_s*50 + 128 (addon/capture_stream_to_text.gd:84) — ragged encoder lengths on every call
    model, pcm = gu.stream_inputs()
    node = host.CaptureStreamToText(lib, transcribe_interval=gu.STREAM_INTERVAL); node.set_language_model(model)
    for ci, (fin, text, n_used, actx, toks) in enumerate(node.stream(pcm)):
        out[f"stream/{ci}/meta"] = np.asarray([int(fin), n_used, actx], np.int64)
        out[f"stream/{ci}/tokens"] = gu.tokens_array([b""] + toks)
    out["stream/n_calls"] = np.int64(ci + 1)
    node.close()
    path = gu.GOLDEN / "hotpath.npz"
    np.savez_compressed(path, **out)
    print("wrote", path, path.stat().st_size, "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
