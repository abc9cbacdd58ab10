#!/usr/bin/env python3
"""Headline benchmark: realtime factor of PCM -> tokens for base.en on 30 s chunks (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step = one complete transcription of one 30 s chunk of synthetic 16 kHz PCM that is already
resident in HBM, with the Godot host's parameter set (src/speech_to_text.cpp:403-413: greedy,
max_tokens 16, single_segment, token timestamps): log-mel -> conv -> 6 encoder blocks -> cross K/V ->
prompt + <=17 KV-cached decoder steps with the reference's logit filters and sampling.
Rank r transcribes its own chunks (independent units, no collective in the hot loop: SURVEY §8(e));
the only collective is the RCCL broadcast of the ggml model image before timing starts.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

import __graft_entry__ as entry

SHAPE = "base.en"
CHUNK_S = 30.0
# algorithmic work per 30 s chunk: encoder FLOP and decoder bytes per token
ENC_GFLOP = 96.80
DEC_MB_PER_TOKEN = 115.6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default=SHAPE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream-seconds", type=float, default=0.0,
                    help="secondary figure: BASELINE configs[2] — CaptureStreamToText over this many seconds of synthetic microphone "
                         "audio with the `small` multilingual shape (every 0.3 s the grown buffer is transcribed again, ragged audio_ctx)")
    ap.add_argument("--chunks", type=int, default=1,
                    help="chunks per GPU per step: 1 = BASELINE configs[1] (headline); 8 = configs[3]'s per-GPU share, lock-step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # rehearsal of the N > 1 flow on a box with fewer GPUs than ranks (ranks share devices, gloo instead of RCCL):
    # WMI_BENCH_REHEARSAL=1 — never set by the driver, the JSON line says so in "data"
    rehearsal = world > 1 and os.environ.get("WMI_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    entry.load_package()
    from godot_whisper_amd import abi, host, runtime, shard, synth

    lib = runtime.require_gpu()
    runtime.silence_logs(lib)

    # ---- model: rank 0 makes the ggml image; everyone else gets it by ONE RCCL broadcast over xGMI
    model = synth.make_model(args.shape, seed=1234) if rank == 0 else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model_bytes = shard.broadcast_model(model, rank, world, dist, dev)
    torch.cuda.synchronize()
    t_bcast = (time.perf_counter() - t0) if world > 1 else 0.0
    buf = C.create_string_buffer(model_bytes, len(model_bytes))
    ctx = lib.wmi_init_from_buffer_on_device(C.cast(buf, C.c_void_p), len(model_bytes), local_rank)
    assert ctx, "model load failed"
    del buf

    # ---- inputs: a few distinct seeded chunks per rank, already in HBM
    n_distinct = 8
    pcm_host = [synth.make_pcm(CHUNK_S, seed=1234 + 1000 * rank + i) for i in range(n_distinct)]
    pcm_dev = [torch.from_numpy(p).to(dev) for p in pcm_host]
    torch.cuda.synchronize()

    node = host.SpeechToText(lib)
    node.ctx = ctx
    params = node.full_params("", 0)

    def batch_args(i, nb):
        ts = [pcm_dev[(i * nb + j) % n_distinct] for j in range(nb)]
        return (C.c_void_p * nb)(*[t.data_ptr() for t in ts]), (C.c_int * nb)(*[t.numel() for t in ts])

    def step(i):
        if args.chunks > 1:                      # lock-step chunks (include/wmi_device.h: wmi_full_batch)
            ptrs, lens = batch_args(i, args.chunks)
            ret = lib.wmi_full_batch(ctx, params, ptrs, lens, args.chunks, 1)
        else:
            t = pcm_dev[i % n_distinct]
            ret = lib.wmi_full_device_pcm(ctx, params, C.c_void_p(t.data_ptr()), t.numel(), None)
        assert ret == 0, ret

    for i in range(args.warmup):
        step(i)
    t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)()
    lib.whisper_reset_timings(ctx)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    lib.wmi_get_timings(ctx, t6, n5)
    if args.chunks > 1:
        lib.wmi_batch_select(ctx, 0)
    n_tokens = sum(lib.whisper_full_n_tokens(ctx, s) for s in range(lib.whisper_full_n_segments(ctx)))

    # ---- secondary figure in the same run (N = 1, headline configuration only): 8 chunks in lock-step
    batch8 = None
    if args.chunks == 1 and world == 1:
        nb, reps = 8, max(3, args.steps // 8)
        ptrs, lens = batch_args(0, nb)
        for _ in range(2):
            assert lib.wmi_full_batch(ctx, params, ptrs, lens, nb, 1) == 0
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        acc = np.zeros(4); nsteps = 0
        t4 = (C.c_int64 * 4)(); ns = C.c_int32()
        for _ in range(reps):
            assert lib.wmi_full_batch(ctx, params, ptrs, lens, nb, 1) == 0
            lib.wmi_get_batch_timings(ctx, t4, C.byref(ns))
            acc += np.array(list(t4), dtype=np.float64); nsteps += ns.value
        torch.cuda.synchronize()
        tb = (time.perf_counter() - tb0) / reps
        modes = [lib.wmi_batch_chunk_mode(ctx, c) for c in range(nb)]
        batch8 = {"workload": "8 x 30 s chunks per call in lock-step (BASELINE configs[3] per-GPU share), same params",
                  "value": round(nb * CHUNK_S / tb, 1), "unit": "x realtime", "ms_per_call": round(tb * 1e3, 3),
                  "mel_envelope_ms": round(acc[0] / reps / 1e3, 3), "encode_ms": round(acc[1] / reps / 1e3, 3),
                  "decode_ms": round(acc[2] / reps / 1e3, 3), "decode_steps": nsteps // reps,
                  "segments_timestamps_ms": round(acc[3] / reps / 1e3, 3), "chunks_run_alone": int(sum(modes)),
                  "encoder_tflops": round(nb * ENC_GFLOP / (acc[1] / reps / 1e3), 1)}

    stream = None
    if args.stream_seconds > 0 and world == 1:
        stream = stream_config(lib, args.stream_seconds)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    if rank == 0:
        audio_s = CHUNK_S * args.steps * world * args.chunks
        rtf = audio_s / dt_max
        enc_ms = (t6[1] / 1e3) / max(n5[0], 1)
        dec_calls = max(n5[1], 1)
        out = {
            "metric": "realtime-factor (audio-sec/wall-sec), base.en 30 s chunk, PCM->tokens",
            "value": round(rtf, 2), "unit": "x realtime",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt_max / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic" + (" (REHEARSAL: ranks share GPUs, gloo)" if rehearsal else ""),
            "config": {"workload": (f"base.en, single 30 s chunk per step per GPU ({world}x MI355X), greedy decode, host params "
                                    "(max_tokens=16, single_segment, token_timestamps)") if args.chunks == 1 else
                                   (f"base.en, {args.chunks} x 30 s chunks per GPU per step in lock-step, greedy decode, host params"),
                       "chunks_per_gpu_per_step": args.chunks, "tokens_per_chunk": int(n_tokens),
                       "weights": "synthetic seed 1234 (f16 ggml, base.en shape)"},
            "encode_ms": round(enc_ms, 4),
            "decode_ms_per_token": round((t6[2] / 1e3) / dec_calls, 4),
            "mel_ms": round((t6[0] / 1e3) / args.steps, 4),
            "sample_ms_per_step": round((t6[5] / 1e3) / args.steps, 4),
            "weight_bcast_ms": round(1e3 * t_bcast, 3),
        }
        if batch8:
            out["batch8"] = batch8

        if stream:
            out["stream_small"] = stream
        # ---- roofline of the dominant kernel, measured live with HIP events on the context's stream.
        # Dominant by GPU time (profiles/*_kernel_stats.csv) is the decoder's weight-streaming k_gemv; its largest
        # instance — the vocabulary projection, 53.1 MB of f16 weights per launch — is the one reported: HBM bound.
        # The encoder's MFMA GEMM is reported next to it.
        try:
            hp_S = lib.whisper_model_n_audio_state(ctx); T = lib.whisper_model_n_audio_ctx(ctx); NV = lib.whisper_n_vocab(ctx)
            us_gemv = lib.wmi_bench_kernel(ctx, 1, 300)
            us_gemm = lib.wmi_bench_kernel(ctx, 0, 300)
            us_attn = lib.wmi_bench_kernel(ctx, 2, 60)
            alg_bytes = NV * hp_S * 2                       # V*S*2 bytes of d_te per token
            gbs = alg_bytes / (us_gemv * 1e-6) / 1e9
            out["roofline"] = {"kernel": "k_gemv1<8,1> logits = d_te[51864x512] . LN(x)  (f16 weight stream, LN + activations in registers)",
                               "bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(gbs / 8000.0, 4), "traffic": pmc_traffic("k_gemv1<8, 1, false*"),
                               "algorithmic_bytes": alg_bytes, "avg_us": round(us_gemv, 3)}
            flops = 2.0 * T * 4 * hp_S * hp_S
            tf = flops / (us_gemm * 1e-6) / 1e12
            out["roofline_encoder_gemm"] = {"kernel": "k_gemm<64,64,EPI_F16_BIAS_GELU> encoder mlp.0 [1500x2048x512] f16 MFMA",
                                            "bound": "mfma", "achieved": round(tf, 2), "peak": 2500.0, "unit": "TFLOP/s",
                                            "frac": round(tf / 2500.0, 4), "traffic": pmc_traffic("k_gemm<64, 64, 1,* grid=196608"),
                                            "avg_us": round(us_gemm, 3)}
            if batch8:                                  # the same GEMM / attention over the 8 lock-step chunks (M = 12 000)
                us_g8 = lib.wmi_bench_kernel(ctx, 4, 100); us_a8 = lib.wmi_bench_kernel(ctx, 5, 30)
                if us_g8 > 0:
                    tf8 = 8 * flops / (us_g8 * 1e-6) / 1e12
                    out["roofline_encoder_gemm_batch8"] = {"kernel": "k_gemm<128,128,EPI_F16_BIAS_GELU> encoder mlp.0 [12000x2048x512] f16 MFMA, global_load_lds staging",
                                                           "bound": "mfma", "achieved": round(tf8, 2), "peak": 2500.0, "unit": "TFLOP/s",
                                                           "frac": round(tf8 / 2500.0, 4), "traffic": pmc_traffic("k_gemm<128, 128, 1,* grid=385024"),
                                                           "avg_us": round(us_g8, 3)}
                if us_a8 > 0:
                    out["attn_layer_batch8_us"] = round(us_a8, 2)
                    out["attn_layer_batch8_tflops"] = round(8 * 3 * 2.0 * T * T * hp_S / (us_a8 * 1e-6) / 1e12, 1)
            out["attn_layer_us"] = round(us_attn, 2)
            out["attn_layer_tflops"] = round(3 * 2.0 * T * T * hp_S / (us_attn * 1e-6) / 1e12, 1)
            out["encoder_tflops_end_to_end"] = round(ENC_GFLOP / enc_ms, 2)
        except Exception as e:  # pragma: no cover
            out["roofline_error"] = repr(e)
        # ---- 16 chunks in lock-step (after the kernel micro-benchmarks above, which use the 8-chunk work set)
        if batch8 is not None:
            nb16 = 16
            ptrs16, lens16 = batch_args(0, nb16)
            for _ in range(2):
                assert lib.wmi_full_batch(ctx, params, ptrs16, lens16, nb16, 1) == 0
            torch.cuda.synchronize()
            tb0 = time.perf_counter()
            reps16 = 3
            for _ in range(reps16):
                assert lib.wmi_full_batch(ctx, params, ptrs16, lens16, nb16, 1) == 0
            torch.cuda.synchronize()
            tb16 = (time.perf_counter() - tb0) / reps16
            out["batch16"] = {"workload": "16 x 30 s chunks per call in lock-step, same params", "value": round(nb16 * CHUNK_S / tb16, 1),
                              "unit": "x realtime", "ms_per_call": round(tb16 * 1e3, 3)}
        # ---- CPU baseline on this box's host cores (bounded sample), rank 0 / N=1 only
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model_bytes, pcm_host[0])
            # SURVEY §8(d) asks for the host's wider setting too: the same transcription with more worker threads
            wide = min(os.cpu_count() or 4, 32)
            if wide > 4:
                out["cpu_baseline_threads%d" % wide] = cpu_baseline(model_bytes, pcm_host[0], n_threads=wide, budget_s=5.0)
        print(json.dumps(out), flush=True)

    lib.whisper_free(ctx)
    node.ctx = None
    if world > 1:
        dist.destroy_process_group()


def stream_config(lib, seconds: float) -> dict:
    """BASELINE configs[2]: the streaming node's call pattern (addon/capture_stream_to_text.gd:65-120) on the `small`
    multilingual shape: every 0.3 s of simulated time the whole accumulated buffer is transcribed again with
    audio_ctx = total_s * 50 + 128; a sentence ends on punctuation / VAD / 15 s.  Reported: calls, wall per call and how
    many seconds of audio one second of wall sustains (the microphone delivers 1)."""
    from godot_whisper_amd import host, synth
    model = synth.make_model("small", seed=77)
    pcm = synth.make_pcm(seconds, seed=21, gate=True)
    node = host.CaptureStreamToText(lib, transcribe_interval=0.3)
    node.language = "de"
    node.set_language_model(model)
    del model
    list(node.stream(pcm[: 16000 * 3]))                  # warm-up (first-touch)
    # only the library calls are timed: the Python mirror's sample-by-sample VAD loop is host-language overhead
    t_lib = [0.0]; per_call = []
    inner = node.transcribe
    def timed(buffer, initial_prompt="", audio_ctx=0, params=None):
        t = time.perf_counter(); r = inner(buffer, initial_prompt, audio_ctx, params); e = time.perf_counter() - t
        t_lib[0] += e; per_call.append(e)
        return r
    node.transcribe = timed
    calls = 0; finals = 0; samples = 0; ctxs = []
    for fin, text, n, actx, toks in node.stream(pcm):
        calls += 1; finals += int(fin); samples += n; ctxs.append(actx)
    dt = t_lib[0]
    node.close()
    return {"workload": f"small multilingual, {seconds:.0f} s synthetic microphone, transcribe every 0.3 s (grown buffer, ragged audio_ctx)",
            "calls": calls, "sentences": finals, "ms_per_call": round(1e3 * dt / max(calls, 1), 3),
            # calls whose first pass fails the reference's logprob / entropy thresholds go through the temperature fallback
            # (best_of sampling on the general path, W/whisper.cpp:5643-5670): the mean carries them, the median does not
            "ms_per_call_median": round(1e3 * sorted(per_call)[len(per_call) // 2], 3) if per_call else None,
            "audio_ctx_min_max": [int(min(ctxs)), int(max(ctxs))] if ctxs else None,
            "stream_realtime_factor": round(seconds / dt, 1),
            "retranscribed_audio_s_per_wall_s": round(samples / 16000 / dt, 1)}


def pmc_traffic(kernel_key: str):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/*_pmc_summary.json, produced by
    profiles/collect.sh + summarise.py): FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md §HBM) + WRITE_SIZE.
    None when no summary has been committed for this kernel."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        head, _, tail = kernel_key.partition("*")          # "name prefix*suffix": template tails (ring depth, specialisation) may vary
        for k, e in d.items():
            if k.startswith(head) and k.endswith(tail) and "fetch_bytes_x2_corrected" in e:
                best = int(e["fetch_bytes_x2_corrected"] + e.get("write_bytes_raw", 0))
    return best


def cpu_baseline(model_bytes: bytes, pcm: np.ndarray, n_threads: int | None = None, budget_s: float = 10.0) -> dict:
    """The reference's own CPU path (oracle/_ref, kind "reference") when its prebuilt library travelled
    with the snapshot, else this repository's CPU restatement (kind "port").  Sample: the same
    transcription (same model, same 30 s chunk, same host params), run a few times, ~10-30 s of CPU work."""
    entry.load_oracle()
    from godot_whisper_amd import host
    from oracle import reflib
    kind, lib, threads = None, None, None
    if reflib.available():
        import ctypes as C
        from godot_whisper_amd import abi
        lib = reflib.lib()
        cb = abi.ggml_log_callback(lambda lvl, txt, ud: None)
        lib.whisper_log_set(C.cast(cb, C.c_void_p), None)
        lib._cb = cb
        kind = "reference"
    else:
        try:
            from oracle import port
            lib = port.lib()
            kind = "port"
        except Exception as e:
            return {"value": None, "error": repr(e)}
    node = host.SpeechToText(lib)
    node.set_language_model(model_bytes)
    p = node.full_params("", 0)
    if n_threads:
        p.n_threads = int(n_threads)
    threads = int(p.n_threads)
    node.transcribe(pcm, params=p)                      # warm
    t0 = time.perf_counter(); n = 0
    while True:
        node.transcribe(pcm, params=p); n += 1
        if time.perf_counter() - t0 > budget_s or n >= 12:
            break
    dt = (time.perf_counter() - t0) / n
    node.close()
    return {"value": round(CHUNK_S / dt, 2), "unit": "x realtime", "cores": threads, "host_cores": os.cpu_count(),
            "kind": kind, "ms_per_chunk": round(dt * 1e3, 1),
            "sample": f"{n} transcriptions of the same 30 s chunk, n_threads={threads}" + (" (whisper.cpp default min(4,hw))" if not n_threads else "")}


if __name__ == "__main__":
    main()
