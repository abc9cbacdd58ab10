/*
 * wmi_device.h — the thin device-level C ABI of libwhisper_mi355.so (new in this repository).
 *
 * whisper.h hands the library HOST buffers (the Godot host owns PackedFloat32Array /
 * PackedByteArray).  The entry points below are what a batching front end, the bench and the
 * parity tests bind: PCM that is already resident in HBM, access to the tensors the reference
 * keeps in its `whisper_state` (W/whisper.cpp:769-844), and the per-stage timers.
 * Shape follows the reference's encoder-plugin precedent
 * (W/openvino/whisper-openvino-encoder.h:12-27, W/coreml/whisper-encoder.h:14-22):
 * init / encode / free with plain pointers and sizes.  No torch types, no C++ types.
 *
 * Threading: a context owns one HIP stream and one set of activation / cache arenas; every entry point that touches them
 * (whisper_full*, whisper_encode / decode, *_with_state, wmi_full_batch, wmi_get_tensor, wmi_set_audio_ctx, the timers and
 * batch accessors) takes the context's recursive mutex, so calls on ONE context serialise (whisper.h:535 asks callers for
 * that anyway).  Concurrency comes from several contexts — one per device (wmi_pool_*) or several per device — which share
 * nothing but the process-wide switches wmi_set_lockstep_exact / WMI_* environment knobs.
 */
#ifndef WMI_DEVICE_H
#define WMI_DEVICE_H

#include <stddef.h>
#include <stdint.h>

#include "whisper_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Library / device identification; aborts nothing, returns 0 on a box without a usable GPU. */
WHISPER_API int          wmi_device_count(void);
WHISPER_API const char * wmi_version(void);

/* Like whisper_init_from_buffer_with_params but pins the context to HIP device `device`
 * (one process per GPU: pass LOCAL_RANK).  replaces: W/whisper.h:151 for multi-GPU hosts. */
WHISPER_API struct whisper_context * wmi_init_from_buffer_on_device(const void * buffer, size_t buffer_size, int device);

/* Vocabulary + host-side decision logic only (tokenizer, logit filters, sampling): no device is touched
 * and every compute entry point (mel / encode / decode / whisper_full with audio) fails with an error.
 * Exists so the host logic can be unit-tested on machines without a GPU; it is NOT a CPU fallback. */
WHISPER_API struct whisper_context * wmi_init_host_only(const void * buffer, size_t buffer_size);

/* Multi-GPU load without re-parsing ("one RCCL broadcast of the packed weight blob").  The device weight
 * arena — every matrix already in the layout the kernels stream, quantised blocks still quantised — has the same byte layout on
 * every rank, because it is derived from the tensor directory alone:
 *   rank 0:      ctx = wmi_init_from_buffer_on_device(model, n, dev);  h = wmi_model_header(model, n, buf, cap)
 *   every rank:  receives the h bytes of buf (a ~1 MB header image: hyper-parameters, mel filters, vocabulary, tensor directory)
 *   rank != 0:   ctx = wmi_init_from_header(buf, h, dev)     -> arena allocated, ZEROED and laid out; the context is "weights
 *                pending": every compute entry point (mel / encode / decode / whisper_full / wmi_full_batch) fails with an error
 *   every rank:  ncclBroadcast(wmi_arena_ptr(ctx), wmi_weights_bytes(ctx, 0), root 0)   (godot-whisper_amd/shard.py)
 *   rank != 0:   wmi_arena_commit(ctx)                        -> device synchronised, the context may compute
 * The header image is accepted by wmi_init_from_header ONLY: whisper_init_from_buffer* and wmi_init_from_buffer_on_device reject it
 * (a context that "loads fine" and transcribes from an unfilled arena must not exist).
 * wmi_model_header returns the image size (call with out == NULL to size the buffer), 0 for an invalid model.
 * wmi_weights_bytes: which = 0 the whole arena, 1 the matrices only (what a decoded token streams; quantised models: their
 * blocks), 2 the ggml type of the quantised matrices (0: f16), 3 the bytes of resident f16 images of quantised ENCODER matrices held by
 * the process (opt-in, WMI_QENC_F16_CACHE=1: written on first use beside the blocks; 0 by default — the weights are held quantised only). */
WHISPER_API size_t wmi_model_header(const void * model, size_t model_size, void * out, size_t cap);
WHISPER_API void * wmi_arena_ptr(struct whisper_context * ctx);
WHISPER_API struct whisper_context * wmi_init_from_header(const void * header, size_t header_size, int device);
WHISPER_API int    wmi_arena_commit(struct whisper_context * ctx);      /* 0 ok; < 0: no arena / device error */
WHISPER_API int    wmi_weights_pending(struct whisper_context * ctx);   /* 1 while a header-image context waits for its arena */
WHISPER_API size_t wmi_weights_bytes(struct whisper_context * ctx, int which);

/* PCM already in HBM (f32 mono 16 kHz, device pointer on the context's device) -> log-mel in the
 * context.  replaces: whisper_pcm_to_mel (W/whisper.h:240) when the samples never touch the host. */
WHISPER_API int wmi_pcm_to_mel_device(struct whisper_context * ctx, const float * d_samples, int n_samples);

/* whisper_full over device-resident PCM: identical control flow and results, minus the H2D copy.
 * `h_samples_for_timestamps` may be NULL unless params.token_timestamps is set (the timestamp
 * heuristics read the waveform on the host, W/whisper.cpp:5003-5010). */
WHISPER_API int wmi_full_device_pcm(struct whisper_context * ctx, struct whisper_full_params params,
                                    const float * d_samples, int n_samples, const float * h_samples_for_timestamps);

/* Several GPUs behind ONE host process (a Godot host is one process).  Ownership as in whisper_full_parallel
 * (W/whisper.cpp:5837-5913: shared read-only model, one state + one worker thread per piece, results gathered by the caller),
 * with a GPU per worker: devices[0] parses the file, the others receive the header image and a peer-to-peer copy of the
 * weight arena (nothing is re-parsed or re-quantised); chunk c -> devices[c mod n]; the chunks of a device advance in lock-step
 * (wmi_full_batch) on their own host thread; no collective and no cross-device traffic after the load.
 *   wmi_pool_full     host PCM pointers; returns whisper_full's codes (first failing device wins)
 *   wmi_pool_select   routes the whisper_full_n_segments / whisper_full_get_* accessors: returns the context that owns the
 *                     chunk's result, already selected (NULL for a bad index)
 * A device id may repeat (several contexts on one GPU: used by the tests on a one-GPU box). */
struct wmi_pool;
WHISPER_API struct wmi_pool * wmi_pool_init(const void * model, size_t model_size, const int * devices, int n_devices);
WHISPER_API void wmi_pool_free(struct wmi_pool * pool);
WHISPER_API int  wmi_pool_size(struct wmi_pool * pool);
WHISPER_API struct whisper_context * wmi_pool_context(struct wmi_pool * pool, int i);
WHISPER_API int  wmi_pool_full(struct wmi_pool * pool, struct whisper_full_params params, const float * const * pcm,
                               const int * n_samples, int n_chunks);
WHISPER_API struct whisper_context * wmi_pool_select(struct wmi_pool * pool, int chunk);
WHISPER_API int64_t wmi_pool_device_time_us(struct wmi_pool * pool, int i);

/* Host-adjacent DSP of the streaming node on the device, so that raw capture frames need not be touched by the
 * CPU.
 *   wmi_downmix_stereo   interleaved stereo f32 frames [n_frames][2] -> mono (x + y) / 2
 *                        replaces: _vector2_array_to_float_array, src/speech_to_text.cpp:45-51
 *   wmi_vad              SpeechToText::voice_activity_detection (src/speech_to_text.cpp:53-104, 378-399) on the last 3 s of
 *                        `pcm` (16 kHz mono): 1 = no voice activity in the last 500 ms, 0 = activity or fewer than 3 s of audio.
 *                        energies (optional, 2 floats) receives energy_all, energy_last.
 * `on_device` != 0: the input (and `mono_out`) are device pointers on the context's device, else host pointers. */
WHISPER_API int wmi_downmix_stereo(struct whisper_context * ctx, const float * frames, int n_frames, int on_device, float * mono_out);

/* The node's 16 kHz resampler on the device.
 *   replaces: _resample_audio_buffer (src/speech_to_text.cpp:16-43) = libsamplerate's src_simple(&data, interpolator_type, 1)
 *             (thirdparty/libsamplerate/src/samplerate.c:469-483; SINC converters thirdparty/libsamplerate/src/src_sinc.c:283-427)
 * converter: the host's InterpolatorType (src/speech_to_text.h:151-155): 2 = SRC_SINC_FASTEST (what capture_stream_to_text.gd:76 asks
 * for), 1 = SRC_SINC_MEDIUM_QUALITY; 0 = SRC_SINC_BEST_QUALITY answers -10: its coefficient table is a missing blob of the reference
 * checkout.  Mono f32 frames at src_rate -> dst (room for dst_capacity >= int(n_frames * dst_rate / src_rate) frames) at dst_rate
 * (the host passes WHISPER_SAMPLE_RATE).  Returns the frames written — the host's result_size; equal rates copy — or 0 where
 * src_simple reports an error (ratio outside [1/256, 256]); < 0: -1 arguments, -2 / -3 device, -4 dst_capacity too small.
 * Every output frame equals the sequential CPU code bit for bit (double accumulators, taps in its order). */
WHISPER_API int wmi_resample(struct whisper_context * ctx, const float * src, int n_frames, int src_rate, int dst_rate, int converter,
                             int on_device, float * dst, int dst_capacity);
WHISPER_API int wmi_vad(struct whisper_context * ctx, const float * pcm, int n_samples, int on_device, float vad_thold, float freq_thold,
                        float * energies);

/* Several independent chunks on one GPU in lock-step (BASELINE config 4: 8 chunks per GPU).  Every chunk is
 * transcribed as by whisper_full(ctx, params, pcm[c], n_samples[c]) on a freshly initialised context
 * (params.no_context = true, decoder RNGs at their initial seed);
 * the chunks share the weights and advance together: encoder GEMMs over all chunks at once (M = chunks * n_ctx),
 * one decode step = one token for every chunk.  Up to 16 chunks advance together; more are processed in groups of 16.
 * replaces: the per-worker loop of whisper_full_parallel (W/whisper.cpp:5809-5935: shared model, one
 * whisper_state per worker) — workers are rows of the same kernels instead of threads.
 * Lock-step needs greedy sampling at temperature 0 without callbacks and a known language; otherwise, and for
 * any chunk that triggers the temperature fallback, the chunk is run alone through the whisper_full driver.
 * pcm[c] are host pointers, or device pointers when pcm_on_device != 0.  Returns whisper_full's codes. */
WHISPER_API int wmi_full_batch(struct whisper_context * ctx, struct whisper_full_params params, const float * const * pcm,
                               const int * n_samples, int n_chunks, int pcm_on_device);
/* Lock-step projections run on the matrix cores by default (f32 sums in MFMA order), and a one-chunk encoder splits
 * the attention keys over two wavefront groups (partial sums added at the end).  on != 0 keeps the projections on the
 * weight-streaming VALU kernel and the attention on one group for every batch size: wmi_full_batch and whisper_full
 * then agree bit for bit (used by the parity tests; process-wide debug switch, set it before both calls). */
WHISPER_API void wmi_set_lockstep_exact(int on);
/* Make the whisper_full_n_segments / whisper_full_get_* accessors (W/whisper.h:541-575) read chunk `chunk` of the
 * last wmi_full_batch call.  Returns its segment count, -1 for a bad index. */
WHISPER_API int wmi_batch_select(struct whisper_context * ctx, int chunk);
/* 0: chunk `chunk` was decoded in lock-step; 1: it was run alone (fallback, see above); -1: bad index. */
WHISPER_API int wmi_batch_chunk_mode(struct whisper_context * ctx, int chunk);
/* Lock-step calls in GROUPS side by side: wmi_full_batch deals its lock-step chunks to `n` contiguous ranges, each a lock-step call of its
 * own — range 0 on this context, the others on replica contexts (own state, stream and work set; the weights are borrowed) on host threads
 * of their own.
 * n = 0: the default (two groups from 16 chunks on — measured: 16 chunks 9.1 -> 8.3 ms per call, 8 chunks level; WMI_LOCKSTEP_GROUPS
 * overrides), 1: one chain.  A chunk's result does not depend on the grouping in the
 * exact mode (wmi_set_lockstep_exact); in the default mode the usual last-bit differences between chunk counts apply.
 * Returns the previous setting, -2 for a null context. */
WHISPER_API int wmi_set_lockstep_groups(struct whisper_context * ctx, int n);

/* Chunks that cannot advance in lock-step (beam search, temperature > 0, quantised beams ...) are run through the whisper_full
 * driver, up to 1 + n of them at a time: n replica contexts (own whisper_state and stream, the weight arena shared read-only with
 * `ctx`, ~0.15 GB of state each for base.en, ~0.6 GB for large-v3) work beside `ctx`, chunk c on worker c mod (1 + n).
 * replaces: the worker threads of whisper_full_parallel (W/whisper.cpp:5837-5913: one shared model, one whisper_state per worker).
 * Results do not depend on n (every chunk is whisper_full on a fresh state).  n = 0: one chunk at a time; n < 0: the default (3, or
 * WMI_BATCH_REPLICAS).  Not used when params carry user callbacks or print_realtime.  Lowering n releases the surplus replicas at once
 * (their state memory and streams).  With n > 0 whisper_log_set's callback and the print_progress lines can be invoked from the
 * library's worker threads (one call at a time: the library serialises them).  Returns the previous setting (-1 = default), -2 for a
 * null context. */
WHISPER_API int wmi_set_batch_replicas(struct whisper_context * ctx, int n);

/* Wall-clock buckets of the last wmi_full_batch call, microseconds: mel + envelope, encoder, decoder steps,
 * segment emission + token timestamps; n_steps = lock-step decode steps.  (cf. whisper_timings, W/whisper.cpp:3672) */
WHISPER_API void wmi_get_batch_timings(struct whisper_context * ctx, int64_t * t4, int32_t * n_steps);

/* Encoder length override for the bare whisper_encode / whisper_decode calls, i.e. what whisper_full
 * does with params.audio_ctx (W/whisper.cpp:5098-5102); 0 = model default.  Returns -5 if too large. */
WHISPER_API int wmi_set_audio_ctx(struct whisper_context * ctx, int n_audio_ctx);

/* Copy an internal tensor out as f32 (f16 tensors are widened).  Returns the element count, or -1 for
 * an unknown name; with dst == NULL only reports the count.  Names and layouts (row-major):
 *   "mel"        [n_mel][n_len]                 log-mel (W/whisper.cpp:2779)
 *   "embd_conv"  [n_ctx][n_state]               conv front-end output (reference holds the transpose)
 *   "embd_enc"   [n_ctx][n_state]               encoder output
 *   "cross_k"    [n_text_layer][n_ctx][n_state] cross-attention keys (pre-scaled by d^-1/4)
 *   "cross_v"    [n_text_layer][n_ctx][n_state] cross-attention values (reference holds [state][ctx])
 *   "self_k" / "self_v"  [n_text_layer][3*n_text_ctx][n_state]
 *   "enc_x"      [n_ctx][n_state]               residual stream after the last encoder block
 */
WHISPER_API int wmi_get_tensor(struct whisper_context * ctx, const char * name, float * dst, int n);
WHISPER_API int wmi_mel_dims(struct whisper_context * ctx, int * n_len, int * n_len_org, int * n_mel);

/* Stage timers in microseconds, the reference's counters (W/whisper.cpp:770-783):
 * t[0..5] = mel, encode, decode, batchd, prompt, sample ; n[0..4] = n_encode, n_decode, n_batchd, n_prompt, n_sample
 * What a timer measures: whisper_full on an f16 model does not wait behind the log-mel and the encoder (the decoder's launches queue up
 * behind them on the stream); their timers are then the GPU time of each phase, taken from stream events when the first decode step's
 * sample arrives, and the decode timer is the host's wall time minus what spilled over from those phases.  Every other path (the stage
 * calls whisper_pcm_to_mel / whisper_encode / whisper_decode, block-quantised models, lock-step calls, WMI_PHASE_SYNC=1) waits behind each
 * phase and records host wall time.  A consequence for errors: a device fault inside the deferred encoder surfaces at the first decode
 * step's wait, i.e. as whisper_full's "failed to decode" (-7 / -8) rather than "failed to encode" (-6); a faulted device fails every later
 * call either way. */
WHISPER_API void wmi_get_timings(struct whisper_context * ctx, int64_t * t6, int32_t * n5);

/* The context's HIP stream (as void*), so a caller can order its own work / events against the hot path. */
WHISPER_API void * wmi_stream(struct whisper_context * ctx);

/* Draws of beam search and temperature > 0 (whisper_sample_token / _topk, W/whisper.cpp:4777-4909) run on the device by default: the
 * host keeps each decoder's mt19937 and ships its uniform numbers, the device searches the cumulative distribution in double over
 * block sums (k_prob_blocks / k_draw).  CONTRACT: not bit-exact with std::discrete_distribution on the host — the partial sums are
 * associated differently and exp() is the device's, so a uniform number that falls within ~1e-7 of a cell boundary may select the
 * neighbouring token, after which that stream diverges like any sampled stream would (tests: test_device_draws_equal_host_draws
 * pins equality on its fixtures, not in general).  WMI_HOST_DRAWS=1 (environment, read per window) keeps the reference's definition:
 * logits come back to the host, whisper_process_logits + std::discrete_distribution run there, bit-exact given equal logits.
 * In device-draw mode the per-step logits stay in HBM: whisper_get_logits() is only defined after whisper_decode() / with
 * WMI_HOST_DRAWS=1 (greedy whisper_full also keeps them on the device).
 *
 * Host-logic probes used by the parity tests (same decisions as the reference given the same logits). */
WHISPER_API int wmi_process_logits(struct whisper_context * ctx, struct whisper_full_params params, const float * raw_logits,
                                   const whisper_token * hist, int n_hist, int has_ts, int seek_delta, float temperature,
                                   float * out_logits, float * out_logprobs, float * out_probs);
WHISPER_API int wmi_sample_draws(struct whisper_context * ctx, const float * probs, const float * logprobs, int n_draw,
                                 int reseed, whisper_token_data * out);

/* Kernel-level cross-check of the two decoder projection implementations (weight-streaming GEMV vs
 * MFMA GEMM) on identical random inputs; op: 0 self q|k|v, 1 self out, 2 cross q, 4 mlp.0, 5 mlp.2.
 * Returns the largest absolute difference over all outputs (negative on error). */
WHISPER_API double wmi_selftest_proj(struct whisper_context * ctx, int op, int n, int layer);

/* Host half of wmi_resample on its own (no device needed; CPU-side tests): the frame counts src_simple reports for n_frames
 * mono frames at src_rate -> dst_rate (output capacity int(n_frames * ratio) as the host passes it), and the first n_pos output
 * positions (integer sample, fraction) the kernel would use.  Returns 0, or the converter error as wmi_resample logs it.
 * closed_form: 1 when the positions come from the exact 128-bit product, 0 when the host ran the double recurrence. */
/* Test hook for the device form of the token timestamps' envelope side (csrc/k_mel.hip k_ts_refine; opt-in, WMI_TS_DEVICE=1):
 * `envelope` [n] is a host array standing in for the |x| envelope; for each of the n_tok tokens, s0s1[2 t] / s0s1[2 t + 1] are its start / end
 * sample.  Writes per token sums[t] = the sequential f32 sum of envelope[max(s0 - 2000, 0) .. min(s1 + 2000, n)), thold[t], and
 * walks[6 t ..] = { en[s0] > thold, en[s1] > thold, walk_down_while_above(s0), walk_up_while_below(s0, s1), walk_up_while_above(s1, n - 1),
 * walk_down_while_below(s1, 0) } — the loops of W/whisper.cpp:6540-6590.  Returns 0, or < 0 on argument / device errors. */
WHISPER_API int wmi_selftest_ts_refine(struct whisper_context * ctx, const float * envelope, int n, const int * s0s1, int n_tok,
                                       float * sums, float * thold, int * walks);
WHISPER_API int wmi_selftest_resample_plan(int n_frames, int src_rate, int dst_rate, int converter, long long * frames_gen,
                                           long long * frames_used, int * closed_form, int n_pos, long long * pos, double * frac);

/* Block-quantised kernels on caller data (parity tests, no context needed): w_blocks = the ggml blocks of an [N][K] matrix of
 * type `qtype` (ggml_type id: 2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0; W/ggml-quants.h:10-47) exactly as a model file holds them,
 * x = f32 activation rows [M][K], all host pointers.  The rows are quantised to q8 blocks as the reference's mul_mat does
 * (W/ggml-quants.c:837-870) and multiplied with the integer-dot kernels:
 *   mode 0  weight-streaming row kernel (M <= 32)        mode 1  tiled GEMM (N % 128 == 0)
 *   mode 2  x ignored; M token ids in `tokens`: out[M][K] = dequantised rows `tokens[i]` of the matrix (get_rows)
 *   mode 3  the large-M form of mode 1 (N % 128 == 0): the same q8 rows as f16(d_a q_a) against the blocks as f16(d_w q_w + m_w) on the
 *           f16 MFMA GEMM, f32 accumulation (what encode uses from 256 activation rows on; WMI_QGEMM_F16_ROWS=0 keeps mode 1's kernel)
 *   mode 4  mode 3 through the product's own route: the row quantiser's launch also expands the weight blocks, then the projection
 *           (M >= 256; fails with -3 when that route is not taken)
 * out [M][N] f32 (mode 2: [M][K]); out_qs [M][K] int8 and out_ds [M][K/32][2] {d, s} receive the quantised rows when non-NULL.
 * Returns 0, or a negative value on a bad argument / device error. */
WHISPER_API int wmi_selftest_quant(int device, int qtype, int mode, const void * w_blocks, const float * x, const int32_t * tokens,
                                   int M, int N, int K, float * out, int8_t * out_qs, float * out_ds);

/* Kernel micro-benchmarks on synthetic operands (used by bench.py for the roofline line):
 * runs `iters` launches on the context stream between two HIP events, returns average microseconds.
 *   which = 0  encoder MLP-0 GEMM  [T x 4S x S] f16 MFMA   (flops = 2*T*4S*S)
 *   which = 1  logits GEMV         [n_vocab x S] weight stream (bytes = n_vocab*S*2)
 *   which = 2  encoder attention   one layer
 *   which = 3  full encode (conv + L blocks + cross) — same as whisper_encode without the host sync per call
 *   which = 4  encoder MLP-0 GEMM over the lock-step work buffers [chunks*T x 4S x S] (after a wmi_full_batch call)
 *   which = 5  encoder attention, all lock-step chunks, one layer
 *   which = 6  the logits projection over a rotating set of copies of the matrix (> 256 MiB in total): an HBM figure, where
 *              which = 1 re-streams one matrix that the 256 MiB Infinity Cache holds
 *   which = 10..12  chains of trivial dependent kernels on 1 / 32 / 256 workgroups (launch floor)
 *   which = 20  the kernels of the last greedy decode step back to back, no host in the loop (microseconds per step);
 *               the environment variable WMI_STEP_MASK selects kernel kinds (bit 0 embed, 1 q|k|v, 2 self-attention + out,
 *               3 cross scores, 4 cross combine + out, 5 mlp.0, 6 mlp.2, 7 logits, 8 filters)
 *   which = 21..36  the same for the lock-step step of (which - 20) chunks
 */
WHISPER_API double wmi_bench_kernel(struct whisper_context * ctx, int which, int iters);

/* The A/B switches that the launch paths read from the environment (WMI_NO_MLP_PAIR, WMI_PAIR_WPB, WMI_SA_WPB, WMI_GEMV1_WIDE_GENERIC,
 * WMI_HOST_DRAWS, WMI_DEBUG_SYNC, WMI_PAIR_WITHHOLD, WMI_PAIR_SPIN_CAP, WMI_NO_FRONT, WMI_FRONT_WITHHOLD, WMI_NO_XBACK, WMI_XBACK_WITHHOLD) are read ONCE per process; a lab script that flips them between
 * probe calls of one process calls this afterwards.  Not while a transcription runs on another thread. */
WHISPER_API void wmi_reload_knobs(void);

/* Status of the one-launch forms of the greedy step (k_mlp_pair, k_front, k_xback: their workgroups hand rows to each other INSIDE the launch; one status word).  A
 * hand-off that does not complete is reported by the kernel, the step is run again in the two-launch form and the state keeps that form;
 * a slow hand-off (the device is shared with work this process does not count) switches to two launches for the next 512 steps.
 * out3 = { steps re-run, slow hand-offs seen, bit 0: the one-launch form is off for good, bit 1: off for now, bit 2: the lock-step rows'
 * one-launch front (k_front with the row on grid.y, wmi_full_batch) is off; its re-run steps are counted in out3[0] }.  rearm != 0 allows the
 * one-launch forms again (tests).  Returns 0, -1 without a context / state.  Reference behaviour matched: a decode either succeeds or
 * reports, W/whisper.cpp:2517-2595. */
WHISPER_API int wmi_pair_status(struct whisper_context * ctx, int32_t * out3, int rearm);

/* Probe: body / boundary split of the last greedy decode step's dependent launches, from time stamps taken INSIDE the kernels
 * (s_memrealtime at a wavefront's first instruction and behind its last store).  The step is captured as a graph with stamping on
 * and replayed; per launch i, out[6 i + 0..5] = first wavefront start, last wavefront start, last wavefront end (microseconds
 * from the step's first start), the number of wavefront records, and two optional mid points (one-row projections: activation
 * row ready, first row tile reduced; -1 where a kernel has none).  `out` holds 6 * cap doubles.  chained = 1: the form whose first kernel is the q|k|v
 * projection (the pick kernel of the previous step prepared token, position and activation row on the device).
 * Returns the number of launches written (<= cap), -1 when no greedy step has run on this context (or the model is quantised). */
WHISPER_API int wmi_step_stamps(struct whisper_context * ctx, double * out, int cap, int chained);

/* Probe: in-situ duration of every f16 encoder GEMM (conv front-end as implicit GEMMs, q|k|v, out, mlp.0, mlp.2, cross K/V) of ONE encoder
 * pass, from wall-clock stamps the kernels' workgroups take themselves (first workgroup entered -> last workgroup done, stores drained):
 * the cache state is the pass's own, not a back-to-back loop's.  chunks = 1: the context's last mel spectrogram (after whisper_full /
 * whisper_pcm_to_mel); chunks > 1: the rows of the last wmi_full_batch call.  out[6 i + 0..5] = {epilogue id, M, N, K, microseconds,
 * workgroups}; returns the number of launches written (<= cap), -1 when there is nothing to replay or the model is block-quantised.
 * bench.py builds SURVEY section 8(d)'s "MFMA utilisation on encoder GEMMs" from it (sum of 2 M N K / sum of microseconds). */
WHISPER_API int wmi_encoder_gemm_stamps(struct whisper_context * ctx, int chunks, double * out, int cap);

/* Host arithmetic self-test (no device needed): the window sums of the token timestamps are the reference's left-to-right f32 sums
 * (W/whisper.cpp:6506-6515), evaluated in blocks as integer additions wherever that is exact (csrc/full.cpp: seq_sum_f32).
 * out_blocked = that routine's result for x[0..n), out_plain = the plain loop's; they must have the same bits.  Returns 0. */
WHISPER_API int wmi_selftest_seqsum(const float * x, int n, float * out_blocked, float * out_plain);

/* Host worker pool self-test (no device needed): `reps` jobs of `n_tasks` tasks; returns reps * n_tasks * (n_tasks + 1) / 2
 * when every task of every job ran exactly once. */
WHISPER_API int64_t wmi_selftest_pool(int n_tasks, int reps);

#ifdef __cplusplus
}
#endif
#endif /* WMI_DEVICE_H */
