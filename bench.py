#!/usr/bin/env python3
"""Headline benchmark: realtime factor of PCM -> tokens for base.en on 30 s chunks (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step = one complete transcription of one 30 s chunk of synthetic 16 kHz PCM through the boundary the Godot
host binds — whisper_full(ctx, params, const float * pcm, n) with HOST samples, i.e. the 1.92 MB H2D copy is inside
the timed region — with the host's parameter set (src/speech_to_text.cpp:403-413: greedy, max_tokens 16,
single_segment, token timestamps): log-mel -> conv -> 6 encoder blocks -> cross K/V -> prompt + <=17 KV-cached
decoder steps with the reference's logit filters and sampling.  Secondary figures in the same line: the same with the
PCM already resident in HBM, the uncapped (max_tokens = 0) transcription, 8 / 16 chunks per call in lock-step,
BASELINE configs[4] (large-v3 q5_1, beam 5), the roofline of the kernel with the largest share of GPU time, of the
whole decode step and of the encoder, and the reference's CPU path on this box's cores.
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

ROOT = entry.ROOT

SHAPE = "base.en"
CHUNK_S = 30.0
# algorithmic work per 30 s chunk: encoder FLOP and decoder bytes per token
ENC_GFLOP = 96.80
DEC_MB_PER_TOKEN = 115.6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--shape", default=SHAPE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream-seconds", type=float, default=600.0,
                    help="secondary figure: BASELINE configs[2] — CaptureStreamToText over this many seconds of synthetic microphone "
                         "audio with the `small` multilingual shape (every 0.3 s the grown buffer is transcribed again, ragged audio_ctx)")
    ap.add_argument("--no-config4", action="store_true", help="skip the BASELINE configs[4] figure (large-v3 q5_1, beam 5: ~1 min of model synthesis)")
    ap.add_argument("--device-pcm", action="store_true", help="headline loop over PCM that is already resident in HBM (round-1 behaviour)")
    ap.add_argument("--profile", action="store_true",
                    help="the run profiles/collect.sh puts under rocprofv3: the headline loop + the per-kernel chains only (short), no secondary "
                         "figures, no CPU baseline, no configs[4]")
    ap.add_argument("--chunks", type=int, default=1,
                    help="chunks per GPU per step: 1 = BASELINE configs[1] (headline); 8 = configs[3]'s per-GPU share, lock-step")
    args = ap.parse_args()
    if args.profile:
        args.no_cpu_baseline = True; args.no_config4 = True; args.stream_seconds = 0.0
    IT = 20 if args.profile else 200                 # launches per micro-benchmark chain

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
    comm_backend = "none (1 rank)" if world == 1 else ("gloo (rehearsal, staged through the host)" if rehearsal else "nccl (RCCL over xGMI)")
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

    # ---- model: rank 0 makes the ggml image, parses it and builds the device weight arena; every other rank gets the ~1 MB header
    # image and then the arena itself by ONE RCCL broadcast over xGMI, straight into its own arena allocation (shard.load_replicated)
    model_bytes = synth.make_model(args.shape, seed=1234) if (rank == 0 or world == 1) else None
    ctx, t_bcast = shard.load_replicated(lib, model_bytes, rank, world, dist, local_rank, dev)
    assert ctx, "model load failed"
    arena_bytes = int(lib.wmi_weights_bytes(ctx, 0))
    # what the collective layer actually saw: ranks, and the device each rank computes on (ordinal + PCI bus id) — at N > 1 the ranks
    # must sit on DISTINCT devices unless this is the declared one-GPU rehearsal
    ranks_seen = dist.get_world_size() if world > 1 else 1
    my_dev = {"rank": rank, "ordinal": int(torch.cuda.current_device()), "name": torch.cuda.get_device_name(dev),
              "pci_bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None)}
    if world > 1:
        devs = [None] * world
        dist.all_gather_object(devs, my_dev)
        ids = [(d["ordinal"], d["pci_bus_id"]) for d in devs]
        assert ranks_seen == world, f"backend reports {ranks_seen} ranks, launcher {world}"
        assert rehearsal or len(set(ids)) == world, f"ranks share devices without WMI_BENCH_REHEARSAL=1: {devs}"
    else:
        devs = [my_dev]

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
        elif args.device_pcm:
            t = pcm_dev[i % n_distinct]
            ret = lib.wmi_full_device_pcm(ctx, params, C.c_void_p(t.data_ptr()), t.numel(), None)
        else:                                    # the reference contract: host samples (W/whisper.h:537-541)
            h = pcm_host[i % n_distinct]
            ret = lib.whisper_full(ctx, params, h.ctypes.data_as(C.POINTER(C.c_float)), int(h.size))
        assert ret == 0, ret

    for i in range(args.warmup):
        step(i)
    t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)()
    lib.whisper_reset_timings(ctx)
    # the interpreter's own housekeeping out of the timed region: a full collection over torch's ~10^6 imported objects is a
    # ~50 ms pause that landed on the process's 288th call whatever the warmup (max_at_step); the objects alive now are
    # parked in the permanent generation, the collector stays on
    import gc
    gc.collect(); gc.freeze()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    per_step = np.empty(args.steps)
    for i in range(args.steps):
        ts = time.perf_counter()
        step(i)                                      # (a step returns when its transcription is complete: host-synchronous)
        per_step[i] = time.perf_counter() - ts
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    lib.wmi_get_timings(ctx, t6, n5)
    if args.chunks > 1:
        lib.wmi_batch_select(ctx, 0)
    n_tokens = sum(lib.whisper_full_n_tokens(ctx, s) for s in range(lib.whisper_full_n_segments(ctx)))

    # spread of the headline: the timed steps themselves, and — when K is small — a 200-step sample of the same loop
    spread = {"steps": int(args.steps), "median_ms": round(1e3 * float(np.median(per_step)), 4), "min_ms": round(1e3 * float(per_step.min()), 4),
              "max_ms": round(1e3 * float(per_step.max()), 4), "max_at_step": int(per_step.argmax())}
    if args.steps < 200 and world == 1 and not args.profile:
        extra = np.empty(200)
        for i in range(200):
            ts = time.perf_counter(); step(i); extra[i] = time.perf_counter() - ts
        spread["sample_200"] = {"mean_ms": round(1e3 * float(extra.mean()), 4), "median_ms": round(1e3 * float(np.median(extra)), 4),
                                "min_ms": round(1e3 * float(extra.min()), 4), "p95_ms": round(1e3 * float(np.percentile(extra, 95)), 4)}

    # ---- secondary figures in the same run (N = 1, headline configuration only)
    dev_pcm = None; uncapped = None
    if args.chunks == 1 and world == 1 and not args.device_pcm and not args.profile:
        reps = max(20, min(300, args.steps // 5))
        for i in range(3):
            t = pcm_dev[i % n_distinct]; assert lib.wmi_full_device_pcm(ctx, params, C.c_void_p(t.data_ptr()), t.numel(), None) == 0
        torch.cuda.synchronize(); td0 = time.perf_counter()
        for i in range(reps):
            t = pcm_dev[i % n_distinct]; assert lib.wmi_full_device_pcm(ctx, params, C.c_void_p(t.data_ptr()), t.numel(), None) == 0
        torch.cuda.synchronize(); td = (time.perf_counter() - td0) / reps
        dev_pcm = {"workload": "same step with the PCM already resident in HBM (wmi_full_device_pcm)", "value": round(CHUNK_S / td, 1),
                   "unit": "x realtime", "ms_per_step": round(td * 1e3, 4)}
        # the uncapped transcription (max_tokens = 0: the decoder runs until EOT / end of audio / n_text_ctx/2 - 4)
        # (temperature_inc = 0: on synthetic weights the 190-token stream fails the entropy threshold and the reference's fallback
        # would re-decode it up to five times with best_of sampling — a property of the random weights, not of the path)
        p0 = node.full_params("", 0); p0.max_tokens = 0; p0.temperature_inc = 0.0
        h = pcm_host[0]
        def full0():
            assert lib.whisper_full(ctx, p0, h.ctypes.data_as(C.POINTER(C.c_float)), int(h.size)) == 0
            return sum(lib.whisper_full_n_tokens(ctx, sg) for sg in range(lib.whisper_full_n_segments(ctx)))
        full0(); full0()
        torch.cuda.synchronize(); tu0 = time.perf_counter(); nrep = 10
        for _ in range(nrep): ntok0 = full0()
        torch.cuda.synchronize(); tu = (time.perf_counter() - tu0) / nrep
        uncapped = {"workload": "same chunk with max_tokens = 0 (no cap on the decoded tokens), whisper_full with host PCM",
                    "value": round(CHUNK_S / tu, 1), "unit": "x realtime", "ms_per_step": round(tu * 1e3, 3), "tokens": int(ntok0)}
    batch8 = None
    if args.chunks == 1 and not args.profile:
        # BASELINE configs[3]: 8 chunks per GPU in lock-step — at N > 1 every rank runs its own 8 (64 chunks on 8 GPUs), the figure
        # is all ranks' audio over the slowest rank's time
        nb, reps = 8, max(3, args.steps // 8)
        ptrs, lens = batch_args(0, nb)
        for _ in range(2):
            assert lib.wmi_full_batch(ctx, params, ptrs, lens, nb, 1) == 0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tb0 = time.perf_counter()
        acc = np.zeros(4); nsteps = 0
        t4 = (C.c_int64 * 4)(); ns = C.c_int32()
        for _ in range(reps):
            assert lib.wmi_full_batch(ctx, params, ptrs, lens, nb, 1) == 0
            lib.wmi_get_batch_timings(ctx, t4, C.byref(ns))
            acc += np.array(list(t4), dtype=np.float64); nsteps += ns.value
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tb = (time.perf_counter() - tb0) / reps
        if world > 1:
            tbm = torch.tensor([tb], dtype=torch.float64, device=dev); dist.all_reduce(tbm, op=dist.ReduceOp.MAX); tb = float(tbm.item())
        modes = [lib.wmi_batch_chunk_mode(ctx, c) for c in range(nb)]
        batch8 = {"workload": f"8 x 30 s chunks per call per GPU in lock-step (BASELINE configs[3]: {8 * world} chunks on {world} GPU(s)), same params",
                  "value": round(world * nb * CHUNK_S / tb, 1), "unit": "x realtime (all ranks)", "ms_per_call": round(tb * 1e3, 3),
                  "mel_envelope_ms": round(acc[0] / reps / 1e3, 3), "encode_ms": round(acc[1] / reps / 1e3, 3),
                  "decode_ms": round(acc[2] / reps / 1e3, 3), "decode_steps": nsteps // reps,
                  "segments_timestamps_ms": round(acc[3] / reps / 1e3, 3), "chunks_run_alone": int(sum(modes)),
                  "encoder_tflops": round(nb * ENC_GFLOP / (acc[1] / reps / 1e3), 1)}

    stream = None
    if args.stream_seconds > 0 and world == 1:
        stream = stream_config(lib, args.stream_seconds)
    host_dsp = None
    if world == 1 and not args.profile and args.chunks == 1:
        host_dsp = host_dsp_config(lib, ctx, cpu=not args.no_cpu_baseline)

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
                                    "(max_tokens=16, single_segment, token_timestamps), whisper_full with "
                                    + ("device-resident PCM" if args.device_pcm else "host PCM (H2D inside the timed region)")) if args.chunks == 1 else
                                   (f"base.en, {args.chunks} x 30 s chunks per GPU per step in lock-step, greedy decode, host params"),
                       "chunks_per_gpu_per_step": args.chunks, "tokens_per_chunk": int(n_tokens),
                       "weights": "synthetic seed 1234 (f16 ggml, base.en shape)"},
            "encode_ms": round(enc_ms, 4),
            "decode_ms_per_token": round((t6[2] / 1e3) / dec_calls, 4),
            "mel_ms": round((t6[0] / 1e3) / args.steps, 4),
            "sample_ms_per_step": round((t6[5] / 1e3) / args.steps, 4),
            "comm_backend": comm_backend,
            "weight_bcast_ms": round(1e3 * t_bcast, 3),
            "weight_bcast_gb_per_s": (round(arena_bytes / t_bcast / 1e9, 1) if (world > 1 and t_bcast > 0) else None),
            "weight_arena_mb": round(arena_bytes / 1e6, 1),
            "ranks_seen": int(ranks_seen), "rank_devices": devs,
            "weight_bcast": ("none (one rank: the file is parsed in place)" if world == 1 else
                             f"one {comm_backend} broadcast of the packed device arena (no re-parse on the other ranks)"),
        }
        if args.chunks > 1:                          # lock-step calls keep their own timers (last call)
            t4 = (C.c_int64 * 4)(); ns = C.c_int32(); lib.wmi_get_batch_timings(ctx, t4, C.byref(ns))
            out["encode_ms"] = round(t4[1] / 1e3, 4); out["mel_ms"] = round(t4[0] / 1e3, 4)
            out["decode_ms_per_token"] = round(t4[2] / 1e3 / max(ns.value, 1), 4)
            out["decode_steps_per_call"] = int(ns.value); out["segments_timestamps_ms"] = round(t4[3] / 1e3, 4)
            out["pcm"] = "device-resident (wmi_full_batch, pcm_on_device = 1)"
        out["headline_spread"] = spread
        if dev_pcm:
            out["device_pcm"] = dev_pcm
        if uncapped:
            out["uncapped_tokens"] = uncapped
        if batch8:
            out["batch8"] = batch8

        if stream:
            out["stream_small"] = stream
        if host_dsp is not None:
            out["host_dsp"] = host_dsp
        # ---- rooflines, measured live with HIP events on the context's stream (wmi_bench_kernel).
        # `roofline` = the kernel kind with the largest share of GPU time in the headline configuration.  The decode step is a
        # chain of dependent kernels; each kind is timed in its own back-to-back chain (which = 20 + WMI_STEP_MASK, no host in
        # the loop), its algorithmic bytes are the weights + cache rows it has to read once.
        try:
          if args.chunks > 1:
            # lock-step configuration (configs[3] per-GPU share): the whole decode step of `chunks` rows in its own chain
            step(0)
            hp_S = lib.whisper_model_n_audio_state(ctx); T = lib.whisper_model_n_audio_ctx(ctx); NV = lib.whisper_n_vocab(ctx)
            Lt = lib.whisper_model_n_text_layer(ctx)
            us_step = lib.wmi_bench_kernel(ctx, 20 + min(args.chunks, 16), 100)
            step_bytes = (DEC_MB_PER_TOKEN - 18.4) * 1e6 + min(args.chunks, 16) * 18.4e6      # weights once, cross K/V per chunk
            out["roofline"] = {"kernel": "lock-step decode step (one-row kernels with the row on grid.y for the layer projections, k_rows_mfma vocabulary projection, per-row attention, filters): %d rows" % min(args.chunks, 16),
                               "bound": "hbm", "achieved": round(step_bytes / (us_step * 1e-6) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(step_bytes / (us_step * 1e-6) / 1e9 / 8000.0, 4), "traffic": None, "algorithmic_bytes": int(step_bytes), "avg_us": round(us_step, 2),
                               "note": "the probe replays the step with its embedding launch; inside a call every step after a window's first is chained on the device (no embedding launch), see decode_ms_per_token"}
          else:
            step(0)                                          # the step record / caches of a headline transcription (17 cells)
            hp_S = lib.whisper_model_n_audio_state(ctx); T = lib.whisper_model_n_audio_ctx(ctx); NV = lib.whisper_n_vocab(ctx)
            Lt = lib.whisper_model_n_text_layer(ctx)
            S2 = hp_S * hp_S * 2
            nkv = max(int(n_tokens) // 2, 1)                                    # mean self-attention cache length over the steps of a chunk
            # Per kernel kind of the step, from time stamps taken INSIDE the kernels of a graph replay of the chained step
            # (wmi_step_stamps: s_memrealtime at a wavefront's first instruction and behind its last store): body = first wavefront
            # start -> last wavefront end, boundary = previous launch's end -> this launch's first start (the device's launch boundary,
            # ~1.3 us), avg_us = body + boundary = the launch's share of the dependent chain.  (Round 2 timed each kind as a chain of
            # its 6 launches per graph replay: a 6-kernel replay is bounded by the replay's own fixed cost, which inflated every
            # launch under ~4 us; profiles/r03*_step_chains*.txt.)
            kinds = [   # (name, launches per step, algorithmic bytes per launch, key in profiles/*_pmc_summary.json)
                ("k_gemv1<4,1,false,1,EPI_QKV_DEC> LN + q|k|v projection", Lt, 3 * S2, "k_gemv1<4, 1, false, 1, 5"),
                ("k_gemv1<2,1,false,2,EPI_F32_BIAS_RESID,1,8> self-attention over the KV cache (one head per wavefront, eight wavefronts) + out projection", Lt, S2 + 2 * nkv * hp_S * 2, "k_gemv1<2, 1, false, 2, 2"),
                ("k_xattn_fused<1,true>: LN + cross query, scores, soft-max numerators and P.V over the cross K/V (one launch)", Lt, S2 + 2 * T * hp_S * 2, "k_xattn_fused<1, true>"),
                ("k_gemv1<4,1,false,3,EPI_F32_BIAS_RESID> cross-attention combine + out projection", Lt, S2, "k_gemv1<4, 1, false, 3, 2"),
                ("k_gemv1<4,1,false,1,EPI_F16_BIAS_GELU> LN + mlp.0", Lt, 4 * S2, "k_gemv1<4, 1, false, 1, 1"),
                ("k_gemv1<2,4,false,0,EPI_F32_BIAS_RESID,0,1> mlp.2 (one-wavefront workgroups)", Lt, 4 * S2, "k_gemv1<2, 4, false, 0, 2"),
            ]
            tail_fused = [("k_gemv1<8,1,false,1,EPI_LOGITS,0,4,true> LN + vocabulary projection + logit-filter statistics (epilogue)", 1, NV * hp_S * 2, "k_gemv1<8, 1, false, 1, 100"),
                          ("k_filter_pick<1,12>: arg-max / timestamp rules over the workgroup partials, result to the host, next step's embedding row", 1, 768 * 40, "k_filter_pick")]
            tail_plain = [("k_gemv1<8,1,false,1,EPI_LOGITS> LN + vocabulary projection", 1, NV * hp_S * 2, "k_gemv1<8, 1, false, 1, 100"),
                          ("k_filter_stats: logit filters + soft-max statistics", 1, NV * 5, "k_filter_stats"),
                          ("k_filter_pick<1,1>: arg-max / timestamp rules, result to the host, next step's embedding row", 1, 64 * 40, "k_filter_pick")]
            cap = 256
            sbuf = (C.c_double * (6 * cap))()
            nst = lib.wmi_step_stamps(ctx, sbuf, cap, 1)
            rows = [(sbuf[6 * i], sbuf[6 * i + 2]) for i in range(max(nst, 0)) if sbuf[6 * i + 3] > 0]
            pair_kind = ("k_mlp_pair<4>: LN + mlp.0 + GELU, in-launch hand-off of the hidden row (tagged 8-byte granules), mlp.2 + residual", Lt, 8 * S2, "k_mlp_pair")
            if len(rows) in (5 * Lt + 2, 5 * Lt + 3):        # both MLP projections in ONE launch (k_mlp_pair, round 5): five launches per layer
                kinds = kinds[:4] + [pair_kind]
            elif len(rows) in (4 * Lt + 2, 4 * Lt + 3):      # + the front of the layer in ONE launch (k_front, round 6): four launches per layer
                kinds = [("k_front: LN + q|k|v (cache write), self-attention once per head, out projection + residual; two in-launch hand-offs (tagged 8-byte granules)",
                          Lt, 4 * S2 + 2 * nkv * hp_S * 2, "k_front")] + kinds[2:4] + [pair_kind]
            elif len(rows) in (3 * Lt + 2, 3 * Lt + 3):      # + the back of the cross-attention in ONE launch (k_xback, round 6): three launches per layer
                kinds = [("k_front: LN + q|k|v (cache write), self-attention once per head, out projection + residual; two in-launch hand-offs (tagged 8-byte granules)",
                          Lt, 4 * S2 + 2 * nkv * hp_S * 2, "k_front"),
                         ("k_xback: LN + cross query + key slices over the cross K/V, combine once per head, out projection + residual; two in-launch hand-offs",
                          Lt, 2 * S2 + 2 * T * hp_S * 2, "k_xback"), pair_kind]
            per_layer = len(kinds)
            tail = tail_fused if len(rows) == per_layer * Lt + 2 else tail_plain
            assert len(rows) == per_layer * Lt + len(tail), ("unexpected launch count of the chained step", len(rows))
            body = {}; gap = {}
            order = [k[0] for k in kinds] * Lt + [k[0] for k in tail]
            prev_end = None
            for nm, (st0, en) in zip(order, rows):
                body.setdefault(nm, []).append(en - st0)
                if prev_end is not None:
                    gap.setdefault(nm, []).append(st0 - prev_end)
                prev_end = en
            span_us = rows[-1][1] - rows[0][0]
            table = []
            for name, nl, alg, pkey in kinds + tail:
                b = float(np.mean(body[name])); g = float(np.mean(gap[name])) if name in gap else 0.0
                us = b + g
                gbs = alg / (us * 1e-6) / 1e9
                table.append({"kernel": name, "launches_per_step": nl, "avg_us": round(us, 3), "body_us": round(b, 3), "boundary_us": round(g, 3),
                              "algorithmic_bytes": int(alg), "achieved": round(gbs, 1), "frac": round(gbs / 8000.0, 4),
                              "achieved_body_only": round(alg / (b * 1e-6) / 1e9, 1), "step_share_us": round(us * nl, 2),
                              "traffic": pmc_traffic(pkey + "*")})
            os.environ["WMI_STEP_MASK"] = "0x1ff"
            us_step = lib.wmi_bench_kernel(ctx, 20, IT)
            os.environ.pop("WMI_STEP_MASK", None)
            tot = sum(t["step_share_us"] for t in table)
            for t in table:
                t["step_share"] = round(t["step_share_us"] / tot, 3)
            dom = max(table, key=lambda t: t["step_share_us"])
            out["roofline"] = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["achieved"], "peak": 8000.0, "unit": "GB/s",
                               "frac": dom["frac"], "traffic": dom["traffic"], "algorithmic_bytes": dom["algorithmic_bytes"], "avg_us": dom["avg_us"],
                               "body_us": dom["body_us"], "boundary_us": dom["boundary_us"],
                               "launches_per_step": dom["launches_per_step"], "share_of_decode_step": dom["step_share"],
                               "note": "dominant kernel kind by its share of the decode step (in-kernel stamps); the step is a dependent chain of "
                                       "small launches: latency-bound, see decode_step_kernels (body vs launch boundary per kind)"}
            out["decode_step_kernels"] = table
            out["decode_step_stamped_span_us"] = round(span_us, 2)
            step_bytes = DEC_MB_PER_TOKEN * 1e6
            out["roofline_decode_step"] = {"bound": "hbm", "achieved": round(step_bytes / (us_step * 1e-6) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                           "frac": round(step_bytes / (us_step * 1e-6) / 1e9 / 8000.0, 4), "algorithmic_bytes": int(step_bytes),
                                           "avg_us": round(us_step, 2), "launches": int(sum(t["launches_per_step"] for t in table) + 1),
                                           "chained_step_span_us": round(span_us, 2), "chained_launches": int(sum(t["launches_per_step"] for t in table)),
                                           "note": "the probe replays the step with its embedding launch; inside a transcription every step after the first is chained (the pick kernel prepares the next step on the device): one launch fewer, see decode_ms_per_token"}
            out["roofline_encoder"] = {"bound": "mfma", "achieved": round(ENC_GFLOP / enc_ms, 2), "peak": 2500.0, "unit": "TFLOP/s",
                                       "frac": round(ENC_GFLOP / enc_ms / 2500.0, 4), "algorithmic_gflop": ENC_GFLOP, "encode_ms": round(enc_ms, 4)}
            # the vocabulary projection alone: on one matrix (Infinity-Cache resident after the first pass) and on a rotating > 256 MiB stream
            us_gemv = lib.wmi_bench_kernel(ctx, 1, IT + IT // 2)
            us_rot = lib.wmi_bench_kernel(ctx, 6, IT + IT // 2)
            us_gemm = lib.wmi_bench_kernel(ctx, 0, IT + IT // 2)
            us_attn = lib.wmi_bench_kernel(ctx, 2, max(IT // 3, 10))
            alg_bytes = NV * hp_S * 2                       # V*S*2 bytes of d_te per token
            out["roofline_logits"] = {"kernel": "k_gemv1<8,1> logits = d_te[51864x512] . LN(x)  (f16 weight stream, LN + activations in registers)",
                                      "bound": "hbm", "achieved": round(alg_bytes / (us_rot * 1e-6) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                      "frac": round(alg_bytes / (us_rot * 1e-6) / 1e9 / 8000.0, 4), "traffic": pmc_traffic("k_gemv1<8, 1, false*"),
                                      "algorithmic_bytes": alg_bytes, "avg_us": round(us_rot, 3),
                                      "stream": "rotating copies of the matrix, > 256 MiB in total (HBM)",
                                      "same_matrix_avg_us": round(us_gemv, 3), "same_matrix_gbs_l3_plus_hbm": round(alg_bytes / (us_gemv * 1e-6) / 1e9, 1)}
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
                    out["roofline_encoder_gemm_batch8"] = {"kernel": "k_gemm8<192,EPI_F16_BIAS_GELU> encoder mlp.0 [12000x2048x512] f16 MFMA: persistent 8-wave ping-pong, 192x256 tiles, LDS-DMA rings",
                                                           "bound": "mfma", "achieved": round(tf8, 2), "peak": 2500.0, "unit": "TFLOP/s",
                                                           "frac": round(tf8 / 2500.0, 4), "traffic": pmc_traffic("k_gemm8<192, 1,*"),
                                                           "avg_us": round(us_g8, 3)}
                us_c8 = lib.wmi_bench_kernel(ctx, 9, 60)
                if us_c8 > 0:
                    fl_c8 = 2.0 * 8 * T * (2 * Lt * hp_S) * hp_S
                    out["roofline_encoder_cross_kv_batch8"] = {"kernel": "k_gemm8<192,EPI_CROSS_KV> cross K/V of all decoder layers [12000x6144x512] f16 MFMA",
                                                               "bound": "mfma", "achieved": round(fl_c8 / (us_c8 * 1e-6) / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s",
                                                               "frac": round(fl_c8 / (us_c8 * 1e-6) / 1e12 / 2500.0, 4), "traffic": pmc_traffic("k_gemm8<192, 6,*"),
                                                               "avg_us": round(us_c8, 3)}
                if us_a8 > 0:
                    out["attn_layer_batch8_us"] = round(us_a8, 2)
                    out["attn_layer_batch8_tflops"] = round(8 * 2 * 2.0 * T * T * hp_S / (us_a8 * 1e-6) / 1e12, 1)      # one sweep: QK^T + P.V
            # north_star's number: aggregate MFMA utilisation over ALL encoder GEMMs (one chunk, and the 8 lock-step chunks), in situ.
            # Denominators: the vendor's dense f16 peak (2.5 PFLOP/s) and the MFMA-only loop measured on this pool's boxes
            # (scratch/lab/mfma_peak.hip: 2.03 PFLOP/s, profiles/README.md)
            try:
                step(0)                                         # the one-chunk mel of a headline transcription
                u1 = encoder_gemm_utilisation(lib, ctx, 1, 2030.0)
                if u1: out["encoder_gemm_mfma_utilisation"] = u1
                if batch8:
                    ptrs8, lens8 = batch_args(0, 8)
                    assert lib.wmi_full_batch(ctx, params, ptrs8, lens8, 8, 1) == 0
                    u8 = encoder_gemm_utilisation(lib, ctx, 8, 2030.0)
                    if u8: out["encoder_gemm_mfma_utilisation_batch8"] = u8
            except Exception as e:  # pragma: no cover
                out["encoder_gemm_mfma_utilisation_error"] = repr(e)
            out["attn_layer_us"] = round(us_attn, 2)
            out["attn_layer_tflops"] = round(2 * 2.0 * T * T * hp_S / (us_attn * 1e-6) / 1e12, 1)               # (the round-2 kernel computed QK^T twice: 3 x)
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
            out["batch16"]["groups"] = "two lock-step groups of 8 side by side (replica context; wmi_set_lockstep_groups default from 16 chunks on)"
            try:
                lib.wmi_set_lockstep_groups(ctx, 1)             # the probe replays ONE 16-row encoder pass on this context's work set
                for _ in range(3):
                    tg0 = time.perf_counter(); assert lib.wmi_full_batch(ctx, params, ptrs16, lens16, nb16, 1) == 0; tg1 = time.perf_counter() - tg0
                out["batch16"]["ms_per_call_one_group"] = round(tg1 * 1e3, 3)
                u16 = encoder_gemm_utilisation(lib, ctx, 16, 2030.0)
                lib.wmi_set_lockstep_groups(ctx, 0)
                if u16: out["encoder_gemm_mfma_utilisation_batch16"] = u16
            except Exception as e:  # pragma: no cover
                out["encoder_gemm_mfma_utilisation_batch16_error"] = repr(e)
        # ---- CPU baseline on this box's host cores (bounded sample), rank 0 / N=1 only
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model_bytes, pcm_host[0])
            # SURVEY §8(d) asks for the host's wider settings too: n_threads = hardware concurrency, and an intermediate count
            # (whisper.cpp's spin-barrier thread pool does not scale to hundreds of threads on a 1500 x 512 problem)
            # (n_threads = all 256 hardware threads of this box was measured once — 545 s per chunk, 0.06x realtime: whisper.cpp's
            # spin-barrier pool collapses — and is not repeated in the default run; profiles/README.md)
            hw = os.cpu_count() or 4
            for nt in sorted({min(hw, 32), min(hw, 64)}):
                if nt > 4:
                    out["cpu_baseline_threads%d" % nt] = cpu_baseline(model_bytes, pcm_host[0], n_threads=nt, budget_s=5.0)
    # ---- BASELINE configs[4]: large-v3 q5_1, beam_size = 5 — each GPU's share of "batch = 8 on 8 GPUs" is one chunk; at N > 1 rank 0
    # synthesises the model, the others receive header image + arena by the same two broadcasts as base.en (all ranks take part)
    c4 = None
    if args.chunks == 1 and not args.no_config4 and not args.profile:
        lib.whisper_free(ctx); node.ctx = None; ctx = None              # release base.en before the 1.2 GB model
        try:
            c4 = config4(lib, cpu=(not args.no_cpu_baseline) and world == 1, rank=rank, world=world, dist=dist, local_rank=local_rank, dev=dev)
        except Exception as e:  # pragma: no cover
            c4 = {"error": repr(e)}
    if rank == 0:
        if c4 is not None:
            out["config4_large_v3_q5_1_beam5"] = c4
        print(json.dumps(out), flush=True)

    if ctx:
        lib.whisper_free(ctx)
    node.ctx = None
    if world > 1:
        dist.destroy_process_group()


def config4(lib, cpu: bool = True, rank: int = 0, world: int = 1, dist=None, local_rank: int = 0, dev=None) -> dict:
    """BASELINE configs[4]: large-v3 (32 + 32 layers, 1280 wide, 128 mels) as q5_1 ggml blocks, beam_size = 5, one 30 s chunk
    per GPU.  The weights stay quantised in HBM (csrc/k_quant.hip); the roofline of a decode step uses the q5_1 bytes.
    The model is made by the product's own quantiser (synth.quantize_model: byte-identical to the reference's tool,
    tests/test_synth_and_shard.py) — nothing under oracle/ prepares the input of a timed leg."""
    import torch
    from godot_whisper_amd import abi, host, shard, synth
    t0 = time.perf_counter()
    model = None
    if rank == 0:
        f16 = synth.make_model("large-v3", seed=2024)
        model = synth.quantize_model(f16, "q5_1")
        del f16
    t_model = time.perf_counter() - t0
    ctx, t_bcast = shard.load_replicated(lib, model, rank, world, dist, local_rank, dev)
    assert ctx, "large-v3 q5_1 load failed"
    node = host.SpeechToText(lib); node.ctx = ctx; node.language = "en"
    file_mb = len(model) / 1e6 if model is not None else 0.0
    arena = lib.wmi_weights_bytes(ctx, 0); mats = lib.wmi_weights_bytes(ctx, 1)
    pcm = synth.make_pcm(CHUNK_S, seed=4321 + 1000 * rank)
    S, T, Lt, NV = 1280, 1500, 32, 51866
    res = {"workload": f"large-v3 q5_1 (synthetic seed 2024), one 30 s chunk per GPU ({world} GPU(s)), whisper_full with host PCM, host params",
           "model_file_mb": round(file_mb, 1), "weights_in_hbm_mb": round(arena / 1e6, 1), "model_synthesis_s": round(t_model, 1),
           "weight_bcast_ms": round(1e3 * t_bcast, 3)}
    q = node.full_params("", 0)
    ps = {}
    for name, strat, bs in (("beam5", abi.WHISPER_SAMPLING_BEAM_SEARCH, 5), ("greedy", abi.WHISPER_SAMPLING_GREEDY, 1)):
        p = lib.whisper_full_default_params(strat)
        for f in ("language", "audio_ctx", "split_on_word", "token_timestamps", "suppress_non_speech_tokens", "single_segment",
                  "max_tokens", "entropy_thold", "initial_prompt"):
            setattr(p, f, getattr(q, f))
        if strat == abi.WHISPER_SAMPLING_BEAM_SEARCH:
            p.beam_search.beam_size = bs
        ps[name] = p
        node.transcribe(pcm, params=p); node.transcribe(pcm, params=p)
        lib.whisper_reset_timings(ctx)
        if world > 1:
            torch.cuda.synchronize(); dist.barrier()
        n = 5; t1 = time.perf_counter()
        for _ in range(n):
            r = node.transcribe(pcm, params=p)
        if world > 1:
            torch.cuda.synchronize(); dist.barrier()
        dt = (time.perf_counter() - t1) / n
        if world > 1:                                   # all ranks' audio over the slowest rank's time
            tm = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(tm, op=dist.ReduceOp.MAX); dt = float(tm.item()) / world
        t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)(); lib.wmi_get_timings(ctx, t6, n5)
        res[name] = {"value": round(CHUNK_S / dt, 1), "unit": "x realtime" + (" (all ranks)" if world > 1 else ""), "ms_per_chunk": round(dt * 1e3, 2), "tokens": max(len(r) - 1, 0),
                     "encode_ms": round(t6[1] / 1e3 / max(n5[0], 1), 3),
                     "decode_ms_total": round((t6[2] + t6[3] + t6[4]) / 1e3 / n, 3), "sample_ms_total": round(t6[5] / 1e3 / n, 3)}
    if world > 1:                                       # the probes below are one-GPU figures (N = 1 runs)
        node.close()
        return res
    # 8 chunks of this model in lock-step on one GPU (greedy; configs[3]'s arrangement with configs[4]'s model)
    try:
        nb8 = 8
        pcms = [synth.make_pcm(CHUNK_S, seed=5000 + i) for i in range(nb8)]
        p8 = node.full_params("", 0); p8.temperature_inc = 0.0
        node.transcribe_batch(pcms, params=p8)
        t1 = time.perf_counter(); reps8 = 3
        for _ in range(reps8):
            node.transcribe_batch(pcms, params=p8)
        dt8 = (time.perf_counter() - t1) / reps8
        t4 = (C.c_int64 * 4)(); ns = C.c_int32(); lib.wmi_get_batch_timings(ctx, t4, C.byref(ns))
        res["lockstep8_greedy"] = {"value": round(nb8 * CHUNK_S / dt8, 1), "unit": "x realtime", "ms_per_call": round(dt8 * 1e3, 2),
                                   "encode_ms": round(t4[1] / 1e3, 2), "decode_ms": round(t4[2] / 1e3, 2), "decode_steps": int(ns.value),
                                   "chunks_run_alone": int(sum(node.last_modes)),
                                   "encoder_tflops_equivalent": round(nb8 * 2588.3 / (t4[1] / 1e3), 1)}
        # the same 8 chunks with beam search: not lock-step — whisper_full per chunk on 1 + 3 replica contexts of this GPU (own state
        # and stream, shared weight arena: include/wmi_device.h wmi_set_batch_replicas), against one chunk at a time
        pb = ps["beam5"]; pb.temperature_inc = 0.0
        for label, n_rep in (("beam5_8chunks_one_at_a_time", 0), ("beam5_8chunks_replicas", -1)):
            lib.wmi_set_batch_replicas(ctx, n_rep)
            node.transcribe_batch(pcms, params=pb)
            t1 = time.perf_counter(); repsb = 2
            for _ in range(repsb):
                node.transcribe_batch(pcms, params=pb)
            dtb = (time.perf_counter() - t1) / repsb
            res[label] = {"value": round(nb8 * CHUNK_S / dtb, 1), "unit": "x realtime", "ms_per_call": round(dtb * 1e3, 2),
                          "ms_per_chunk": round(dtb * 1e3 / nb8, 2), "contexts": 1 if n_rep == 0 else 4}
        node.transcribe(pcm, params=q)                      # back to the one-chunk state for the probes below
    except Exception as e:  # pragma: no cover
        res["lockstep8_error"] = repr(e)
    # decode-step roofline, greedy step chain on the GPU: q5_1 decoder matrices + vocabulary projection + cross K/V (f16) once per step
    dec_bytes = Lt * 14 * S * S * 24 // 32 + (NV + 31) // 32 * 32 * S * 24 // 32 + Lt * 2 * T * S * 2
    us_step = lib.wmi_bench_kernel(ctx, 20, 30)
    if us_step > 0:
        res["roofline_decode_step"] = {"bound": "hbm", "achieved": round(dec_bytes / (us_step * 1e-6) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(dec_bytes / (us_step * 1e-6) / 1e9 / 8000.0, 4), "algorithmic_bytes": int(dec_bytes),
                                       "avg_us": round(us_step, 1), "note": "greedy step, 7 launches per layer (the cross query runs inside the cross-attention launch); q5_1 bytes per token"}
    enc_gflop = 2588.3
    res["roofline_encoder"] = {"bound": "mfma", "achieved": round(enc_gflop / res["greedy"]["encode_ms"], 1), "peak": 2500.0, "unit": "TFLOP/s (2 x MAC; q8 activation rows x dequantised q5_1 blocks as f16 operands on the MFMA)",
                               "frac": round(enc_gflop / res["greedy"]["encode_ms"] / 2500.0, 4), "algorithmic_gflop": enc_gflop}
    us_rot = lib.wmi_bench_kernel(ctx, 6, 100)
    vb = (NV + 31) // 32 * 32 * S * 24 // 32
    res["roofline_logits"] = {"kernel": "k_qrows<q5_1> vocabulary projection, rotating > 256 MiB stream", "bound": "hbm",
                              "achieved": round(vb / (us_rot * 1e-6) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(vb / (us_rot * 1e-6) / 1e9 / 8000.0, 4),
                              "algorithmic_bytes": int(vb), "avg_us": round(us_rot, 2)}
    node.close()
    if cpu:
        try:
            entry.load_oracle()
            from oracle import reflib
            cpu = reflib.available()                    # the scalar port would need minutes for this model: compiled reference only
        except Exception:
            cpu = False
    if cpu:
        # the reference's CPU path on the same model and chunk, beam 5: ONE transcription (tens of seconds), 32 threads — its fastest
        # setting on this box for base.en (bench cpu_baseline_threads32)
        try:
            res["cpu_baseline"] = cpu_baseline(model, pcm, n_threads=min(os.cpu_count() or 4, 32), budget_s=0.0, beam_size=5, warm=False)
        except Exception as e:  # pragma: no cover
            res["cpu_baseline"] = {"value": None, "error": repr(e)}
    del model
    return res


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


def host_dsp_config(lib, ctx, cpu: bool = True) -> dict:
    """the streaming node's DSP in front of transcribe — stereo fold, 16 kHz SINC resampler (SRC_SINC_FASTEST), VAD — on
    30 s of stereo capture frames resident in HBM (device pointers in, device pointers out), HIP events around 20 repetitions.
    cpu: the same resampling through oracle/host_dsp.c (libsamplerate's arithmetic restated; one host core) as the CPU figure."""
    import ctypes as C
    import torch
    res = {"workload": "30 s of stereo capture frames in HBM -> mono -> 16 kHz (SRC_SINC_FASTEST) -> VAD, device pointers throughout"}
    for rate in (48000, 44100):
        n = rate * 30
        rng = np.random.default_rng(rate)
        fr = torch.from_numpy((0.3 * rng.standard_normal((n, 2))).astype(np.float32)).cuda()
        mono = torch.empty(n, dtype=torch.float32, device="cuda")
        n16 = int(np.uint32(n) * (16000.0 / rate))
        out = torch.empty(n16 + 8, dtype=torch.float32, device="cuda")
        def once():
            assert lib.wmi_downmix_stereo(ctx, C.c_void_p(fr.data_ptr()), n, 1, C.c_void_p(mono.data_ptr())) == 0
            got = lib.wmi_resample(ctx, C.c_void_p(mono.data_ptr()), n, rate, 16000, 2, 1, C.c_void_p(out.data_ptr()), n16 + 8)
            assert n16 - 1 <= got <= n16, (got, n16)              # 44.1 kHz: libsamplerate's termination test stops one frame early
            assert lib.wmi_vad(ctx, C.c_void_p(out.data_ptr()), got, 1, 2.0, 200.0, None) in (0, 1)
            return got
        n16 = once(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): once()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 20
        # the resampler alone (the calls above synchronise per call: host-paced)
        t0 = time.perf_counter()
        for _ in range(20):
            lib.wmi_resample(ctx, C.c_void_p(mono.data_ptr()), n, rate, 16000, 2, 1, C.c_void_p(out.data_ptr()), n16 + 8)
        ms_rs = 1e3 * (time.perf_counter() - t0) / 20
        res[f"from_{rate}_hz"] = {"ms_per_30s_fold_resample_vad": round(ms, 3), "ms_per_30s_resample_call": round(ms_rs, 3), "frames_out": n16,
                                  "algorithmic_bytes": int(4 * n + 4 * n16), "resample_gb_per_s": round((4 * n + 4 * n16) / (ms_rs * 1e-3) / 1e9, 1)}
        if cpu and rate == 48000:
            so = ROOT / "oracle" / "liboracle_dsp.so"
            if so.exists():
                import struct
                raw = (ROOT / "godot-whisper_amd" / "csrc" / "data" / "sinc_fastest.bin").read_bytes()
                inc, cnt = struct.unpack("<ii", raw[:8]); tab = np.frombuffer(raw[8:], "<f4", cnt).copy()
                dsp = C.CDLL(str(so))
                dsp.oracle_resample_audio_buffer.restype = C.c_uint32
                dsp.oracle_resample_audio_buffer.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
                x = mono.cpu().numpy(); y = np.zeros(n16 + 8, np.float32)
                t0 = time.perf_counter()
                got = dsp.oracle_resample_audio_buffer(x.ctypes.data, n, rate, 16000, tab.ctypes.data, cnt, inc, y.ctypes.data)
                cpu_ms = 1e3 * (time.perf_counter() - t0)
                same = bool(got == n16 and np.array_equal(y[:n16], out[:n16].cpu().numpy()))
                res["cpu_baseline"] = {"value": round(cpu_ms, 1), "unit": "ms per 30 s (resampler only)", "cores": 1, "kind": "port",
                                       "sample": "one 30 s 48 kHz buffer through oracle/host_dsp.c (libsamplerate src_simple restated, parity unpinned)",
                                       "device_output_identical": same}
    return res


def encoder_gemm_utilisation(lib, ctx, chunks: int, measured_peak_tflops: float | None = None) -> dict | None:
    """`MFMA utilisation` on the encoder GEMMs = (conv + projection + MLP + cross K/V FLOP) / (time in those kernels x
    peak).  Times are IN SITU: one encoder pass with a probe slice on every gemm() launch (wmi_encoder_gemm_stamps: first workgroup
    entered -> last workgroup done, taken by the kernels themselves), not back-to-back loops of one warm shape.  FLOP = 2 M N K of each
    launch as launched (the stacked-chunk conv launches include their guard rows: < 0.4 %), except q|k|v of lock-step chunks, whose
    launch covers 1504 rows per chunk (every chunk on a 16-row boundary): the 1500 real rows are counted."""
    cap = 128
    buf = (C.c_double * (6 * cap))()
    n = lib.wmi_encoder_gemm_stamps(ctx, chunks, buf, cap)
    if n <= 0:
        return None
    names = {(1, "conv1"): "conv1 (implicit GEMM, k3 s1, + GELU)", (3, None): "conv2 (implicit GEMM, k3 s2, + GELU + positional)",
             (4, None): "q|k|v^T", (2, "out"): "attention out + residual", (1, "mlp0"): "mlp.0 + GELU", (2, "mlp2"): "mlp.2 + residual",
             (6, None): "cross K/V, all decoder layers"}
    shapes = {}
    order = []
    seen_conv2 = False                                  # the launches in front of the first conv2 are conv1 (same epilogue id as mlp.0)
    for i in range(n):
        epi, M, N, K, us, wgs = (buf[6 * i + j] for j in range(6))
        epi, M, N, K = int(epi), int(M), int(N), int(K)
        if us <= 0:
            continue
        if epi == 4 and chunks > 1:
            M = min(M, chunks * 1500)                   # q|k|v of lock-step chunks is launched over 1504 rows per chunk (16-row boundaries): count the 1500
        seen_conv2 = seen_conv2 or epi == 3
        if epi == 1:
            key = (1, "mlp0") if seen_conv2 else (1, "conv1")
        elif epi == 2:
            key = (2, "out") if K == N else (2, "mlp2")
        else:
            key = (epi, None)
        e = shapes.setdefault(key, {"shape": names.get(key, str(key)), "M": M, "N": N, "K": K, "launches": 0, "us_sum": 0.0, "gflop_sum": 0.0,
                                    "workgroups": int(wgs)})
        if key not in order:
            order.append(key)
        e["launches"] += 1; e["us_sum"] += us; e["gflop_sum"] += 2.0 * M * N * K / 1e9
    tot_us = sum(e["us_sum"] for e in shapes.values()); tot_gf = sum(e["gflop_sum"] for e in shapes.values())
    per = []
    for key in order:
        e = shapes[key]
        tf = e["gflop_sum"] / (e["us_sum"] * 1e-6) / 1e3
        # (the persistent kernel always launches one workgroup per CU; a smaller grid at M >= 4096 is k_gemm's 192 x 128 tiling of the N = S projections)
        kern = ("k_gemm8 (persistent 8-wave ping-pong)" if (e["workgroups"] == 256 and e["M"] >= 4096) else
                "k_gemm<192,128> (one 8-wave workgroup per CU, four-deep ring)" if (e["workgroups"] < 256 and e["M"] >= 4096 and e["N"] <= 1024) else "k_gemm")
        per.append({"shape": e["shape"], "M": e["M"], "N": e["N"], "K": e["K"], "launches": e["launches"], "avg_us": round(e["us_sum"] / e["launches"], 2),
                    "tflops": round(tf, 1), "frac": round(tf / 2500.0, 4), "share_of_gemm_time": round(e["us_sum"] / tot_us, 3),
                    "kernel": kern, "workgroups": e["workgroups"]})
    agg = tot_gf / (tot_us * 1e-6) / 1e3
    out = {"definition": "sum over the encoder's GEMM launches of 2 M N K / sum of their in-situ durations / dense f16 MFMA peak (SURVEY 8(d); north_star's '>= 40 % on encoder GEMMs')",
           "chunks": chunks, "gemm_gflop": round(tot_gf, 2), "gemm_us": round(tot_us, 1), "achieved": round(agg, 1), "unit": "TFLOP/s",
           "peak": 2500.0, "frac": round(agg / 2500.0, 4), "per_shape": per,
           "timing": "in-kernel wall-clock stamps (100 MHz) of one stamped encoder pass: first workgroup entry -> last workgroup done"}
    if measured_peak_tflops:
        out["measured_peak"] = measured_peak_tflops; out["frac_of_measured_peak"] = round(agg / measured_peak_tflops, 4)
    return out


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


def cpu_baseline(model_bytes: bytes, pcm: np.ndarray, n_threads: int | None = None, budget_s: float = 10.0, beam_size: int = 0,
                 warm: bool = True) -> dict:
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
    if beam_size > 0:                                   # the host parameter set on the reference's beam-search defaults
        from godot_whisper_amd import abi
        q = p; p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
        for f in ("language", "audio_ctx", "split_on_word", "token_timestamps", "suppress_non_speech_tokens", "single_segment",
                  "max_tokens", "entropy_thold", "initial_prompt"):
            setattr(p, f, getattr(q, f))
        p.beam_search.beam_size = beam_size
    if n_threads:
        p.n_threads = int(n_threads)
    threads = int(p.n_threads)
    if warm:
        node.transcribe(pcm, params=p)
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
