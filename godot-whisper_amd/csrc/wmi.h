// Internal header of libwhisper_mi355.so: host-side model/state structs and the launch
// prototypes of the gfx950 kernels.  Nothing here is ABI; the ABI is include/*.h.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <functional>
#include <mutex>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../../include/whisper_mi355.h"
#include "../../include/wmi_device.h"
#include "kernels.h"

namespace wmi {

// ---------------------------------------------------------------- logging / errors
void log_msg(ggml_log_level lvl, const char * fmt, ...) __attribute__((format(printf, 2, 3)));
#define WMI_ERR(...)  ::wmi::log_msg(GGML_LOG_LEVEL_ERROR, __VA_ARGS__)
#define WMI_WARN(...) ::wmi::log_msg(GGML_LOG_LEVEL_WARN,  __VA_ARGS__)
#define WMI_INFO(...) ::wmi::log_msg(GGML_LOG_LEVEL_INFO,  __VA_ARGS__)

bool hip_ok(hipError_t e, const char * what, const char * file, int line);
#define HIP_OK(expr) ::wmi::hip_ok((expr), #expr, __FILE__, __LINE__)
// use inside functions returning bool
#define HIP_TRY(expr) do { if (!HIP_OK(expr)) return false; } while (0)

int64_t time_us();

// ---------------------------------------------------------------- model (host view)
struct HParams {            // W/whisper.cpp:537-550
    int32_t n_vocab = 51864, n_audio_ctx = 1500, n_audio_state = 384, n_audio_head = 6, n_audio_layer = 4;
    int32_t n_text_ctx = 448, n_text_state = 384, n_text_head = 6, n_text_layer = 4, n_mels = 80, ftype = 1;
    float   eps = 1e-5f;
};

struct Vocab {              // W/whisper.cpp:365-394
    int n_vocab = 51864;
    std::map<std::string, int32_t> token_to_id;
    std::vector<std::string>       id_to_token;   // dense: ids are 0..n_vocab-1
    int32_t eot = 50256, sot = 50257, translate = 50357, transcribe = 50358, solm = 50359,
            prev = 50360, nosp = 50361, not_ = 50362, beg = 50363;
    bool is_multilingual() const { return n_vocab >= 51865; }
    int  num_languages()   const { return n_vocab - 51765 - (is_multilingual() ? 1 : 0); }
};

// a tensor record of the ggml file: where its payload sits in the caller's buffer
struct FileTensor {
    std::string name;
    int32_t     ttype = 0;          // ggml_type
    int32_t     n_dims = 0;
    int64_t     ne[4] = {1, 1, 1, 1};
    size_t      offset = 0, nbytes = 0;     // payload position in the model buffer (offset = 0, nbytes known: directory-only image)
};

struct ModelFile {
    HParams hp;
    int     model_type = 0;         // e_model: 0 unknown, 1 tiny, 2 base, 3 small, 4 medium, 5 large
    int     n_filt_mel = 0, n_filt_fft = 0;
    std::vector<float> filters;     // [n_mel][n_fft]
    Vocab   vocab;
    std::map<std::string, FileTensor> tensors;
    int     n_loaded = 0;
    bool    quantised = false;      // ftype names a block-quantised weight type: the 2-D tensors are q4_0 .. q8_0 blocks
    size_t  header_bytes = 0;       // bytes of the file in front of the first tensor record (hparams, filters, vocabulary)
    bool    directory_only = false; // image without tensor payloads (wmi_export_header): weights arrive as a device arena
};

// parse header + vocab + tensor directory out of an in-memory ggml file (W/whisper.cpp:1102-1640)
bool parse_model(const uint8_t * buf, size_t n, ModelFile & mf);
// the same file without its tensor payloads ("wmih" image): what a rank needs besides the device arena (multi-GPU load)
std::vector<uint8_t> export_header(const ModelFile & mf, const uint8_t * buf, size_t arena_bytes);

// ---------------------------------------------------------------- weights (device view)
// f16 / f32 files: all matrices are f16, row-major [out][in] with `in` contiguous (the ggml ne0 order), which is
// the K-contiguous "B^T" operand layout of the MFMA GEMM.  Block-quantised files (q4_0 q4_1 q5_0 q5_1 q8_0): the 2-D
// tensors keep their blocks, permuted into the tile layout of k_quant.hip (kernels.h); the w_* pointer of such a matrix
// is null and its q_* member holds the tiles.  Vectors are f32.
struct EncLayerW {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const __half *w_qkv;  const float *b_qkv;    // [3S][S]; rows q | k | v; k-bias = 0
    const __half *w_o;    const float *b_o;      // [S][S]
    const __half *w_fc1;  const float *b_fc1;    // [4S][S]
    const __half *w_fc2;  const float *b_fc2;    // [S][4S]
    k::QMat q_qkv, q_o, q_fc1, q_fc2;
};
struct DecLayerW {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
    const __half *w_qkv;  const float *b_qkv;    // self-attn [3S][S]
    const __half *w_o;    const float *b_o;
    const __half *w_cq;   const float *b_cq;     // cross-attn query [S][S]
    const __half *w_co;   const float *b_co;
    const __half *w_fc1;  const float *b_fc1;
    const __half *w_fc2;  const float *b_fc2;
    k::QMat q_qkv, q_o, q_cq, q_co, q_fc1, q_fc2;
};
struct Weights {
    void *  arena = nullptr;  size_t arena_bytes = 0;
    bool    arena_borrowed = false;     // a replica context of wmi_full_batch: the arena belongs to the context it was made from
    // conv front-end, weights re-laid as [oc][k][ic] (+ zero pad to a multiple of 32) for the
    // overlapped-row implicit GEMM
    const __half *conv1_w; const float *conv1_b; int conv1_k = 0;     // K (padded)
    const __half *conv2_w; const float *conv2_b; int conv2_k = 0;
    const float  *e_pe;                                               // [n_audio_ctx][S]
    std::vector<EncLayerW> enc;
    const float  *e_ln_g, *e_ln_b;
    // cross-attention K/V projections of all decoder layers stacked: [L][2S][S] (k rows, v rows)
    const __half *w_ckv;   const float *b_ckv;                        // bias [L][2S] (k part 0)
    k::QMat       q_ckv;
    const float  *d_pe;                                               // [n_text_ctx][S]
    const __half *d_te;                                               // [n_vocab][S]
    k::QMat       q_te;
    int           qtype = 0;                                          // ggml type of the quantised matrices (0: none); one kind per model
    size_t        matrix_bytes = 0;                                   // bytes of all matrices as stored in the arena (decode roofline)
    std::vector<DecLayerW> dec;
    const float  *d_ln_g, *d_ln_b;
    const float  *mel_filters;                                        // [n_mel][201]
    const float   *mel_taps;                                          // [n_mel][13][4]: the first 12 non-zero groups of each filter, then tap 200
    const int32_t *mel_ranges;                                        // [n_mel][2]: non-zero 4-tap groups [g0, g1) of each filter
};

// Builds the device weight arena.  host_buf = the model file (tensor payloads are converted / permuted into a staging
// image and uploaded); host_buf == nullptr (directory-only model): the arena is only allocated and laid out — the same
// offsets for the same tensor directory — and the caller fills it (one RCCL broadcast of rank 0's arena, wmi_arena_ptr).
bool upload_weights(const ModelFile & mf, const uint8_t * host_buf, Weights & w, hipStream_t st);
// layout only (no device): arena_bytes / matrix_bytes / qtype of the arena upload_weights would build
bool plan_weights(const ModelFile & mf, Weights & w);
void free_weights(Weights & w);

// ---------------------------------------------------------------- KV cache bookkeeping (host)
struct KVCell { int32_t pos = -1; std::set<int32_t> seq; bool has(int32_t s) const { return seq.count(s) != 0; } };
struct KVCache {                                     // W/whisper.cpp:639-664
    uint32_t head = 0, size = 0, n = 0;
    std::vector<KVCell> cells;
    __half * k = nullptr;  __half * v = nullptr;     // device: [L][size][S] each
};
struct Batch {                                       // W/whisper.cpp:407-415
    std::vector<int32_t> token, pos, seq_id; std::vector<int8_t> logits;
    int n_tokens = 0;
    void prep_legacy(const int32_t * tokens, int n, int n_past, int seq);
};
bool  kv_find_slot(KVCache & c, const Batch & b);
int   kv_cell_max(const KVCache & c);
void  kv_clear(KVCache & c);
void  kv_seq_rm(KVCache & c, int32_t seq, int32_t p0, int32_t p1);
void  kv_seq_cp(KVCache & c, int32_t src, int32_t dst, int32_t p0, int32_t p1);

// ---------------------------------------------------------------- decode-time structs
struct Sequence {
    std::vector<whisper_token_data> tokens;
    int    result_len = 0;
    double sum_logprobs_all = 0, sum_logprobs = 0, avg_logprobs = 0, entropy = 0, score = 0;
};
// parse state of grammar-constrained decoding (grammar.cpp; W/whisper.cpp:711-728)
struct PartialUtf8 { uint32_t value = 0; int n_remain = 0; };      // n_remain = -1: invalid sequence
struct GrammarPos  { int rule, off; };                              // an element of rules[rule]
struct Grammar {
    std::vector<std::vector<whisper_grammar_element>> rules;
    std::vector<std::vector<GrammarPos>> stacks;                    // every top rests on a character class
    PartialUtf8 partial;                                             // unfinished UTF-8 sequence of the accepted tokens
};
struct Decoder {
    Sequence sequence;
    Grammar  grammar;
    int  i_batch = 0, seek_delta = 0;
    bool failed = false, completed = false, has_ts = false;
    std::vector<float> probs, logits, logprobs;
    std::mt19937 rng{0};
};
struct Segment { int64_t t0, t1; std::string text; std::vector<whisper_token_data> tokens; bool speaker_turn_next; };

struct Mel { int n_len = 0, n_len_org = 0, n_mel = 0; };

// device scratch for one in-flight chunk; sized at init for the model's maxima
struct DeviceState {
    hipStream_t stream = nullptr; bool stream_own_queue = false;   // own_queue: made by make_own_queue_stream — goes back to the process-wide pool, not destroyed
    // mel
    float * pcm = nullptr;      size_t pcm_cap = 0;          // padded PCM
    float * mel = nullptr;      size_t mel_cap = 0;          // [n_mel][n_len] f32 (reference layout)
    float * mel_max = nullptr;                                // 1 float (ordered-int encoded)
    const float * last_pcm = nullptr; int last_pcm_n = 0;     // device copy of the samples of the last pcm_to_mel
    float * energy = nullptr;   size_t energy_cap = 0;        // |x| envelope (device)
    float * energy_host = nullptr;                            // pinned mirror
    size_t  energy_dev_cap = 0;                               // floats allocated at `energy` (envelope + block extrema; the copy-engine form)
    hipStream_t copy_stream = nullptr; hipEvent_t energy_ev = nullptr;   // envelope D2H overlaps the encoder
    // full(): the log-mel and the encoder are enqueued without a host wait behind them (the decoder's first step goes out while they
    // run); their timers are settled from these events at the next point where the stream is known to be idle (phase_settle)
    hipEvent_t ph_ev[4] = {nullptr, nullptr, nullptr, nullptr}; bool ph_mel = false, ph_enc = false; int64_t ph_host0 = 0;
    hipStream_t energy_wait_stream = nullptr;                 // lock-step calls: the stream of the batched envelope launch that wrote this state's envelope (not owned)
    hipStream_t mel_stream = nullptr;  hipEvent_t mel_ev = nullptr;      // lock-step chunks: the mel kernels of the chunks overlap
    bool    energy_pending = false;                            // copy in flight: signal_energy_wait() before reading state.energy
    bool    energy_device_only = false;                        // the envelope stays in HBM: the timestamp walks run there too (ts_refine_device), nothing crosses PCIe
    void *  ts_host = nullptr;                                 // pinned block: TsTok[448] | TsOut[448] (ts_refine_device)
    bool    energy_unflushed = false;                          // the envelope sits in the device buffer `energy`: signal_energy_flush() starts its copy to the pinned image
    // encoder activations, token-major
    __half * mel_t = nullptr;                                 // [2T+2+pad][n_mel_pad] f16, rows -1 and 2T are zero
    __half * conv1 = nullptr;                                 // [2T+2][S] f16 (row 0 and 2T+1 zero)
    float  * x     = nullptr;                                 // [T][S] f32 residual stream
    float  * embd_conv = nullptr;                             // [T][S] f32 (kept for inspection)
    __half * xn    = nullptr;                                 // [T][S] f16 LN output
    __half * q     = nullptr, * k = nullptr;                  // [T][S] f16
    __half * vt    = nullptr;                                 // [S][Tpad] f16 (V transposed, per head rows)
    __half * att   = nullptr;                                 // [T][S] f16
    __half * h     = nullptr;                                 // [T][4S] f16
    float  * rowmax = nullptr;                                // [H][T] f32 (attention pass A)
    // block-quantised models only: activation rows as q8 blocks (the quantised GEMMs' A operand) and f32 attention outputs
    int8_t * aq = nullptr;  float * ads = nullptr;  int aq_rows = 0;   // [rows][4S] int8; scales d [K / 32][rows] | s [K / 32][rows], rows = max(T, n_text_ctx)
    __half * aq16 = nullptr, * wq16 = nullptr; size_t wq16_elems = 0;  // f16 form of the large-M projections (k_quant.hip, k_qdequant): rows as f16(d q) [rows][4S]; one dequantised weight matrix
    float  * att32 = nullptr;                                 // [T][S] f32
    float  * datt32 = nullptr;                                // [n_text_ctx][S] f32
    float  * enc_out = nullptr;                               // [T][S] f32  (embd_enc)
    __half * enc_out_h = nullptr;                             // [T][S] f16
    __half * kvc_k = nullptr, * kvc_v = nullptr;              // cross cache [L][T][S] each
    int      Tpad = 0;
    int      conv1_max_T = 0;                                  // largest audio_ctx the conv1 buffer has held (rows 1..2T written)
    // decoder activations (n <= n_text_ctx)
    int32_t * d_tokens = nullptr, * d_pos = nullptr;          // [n]
    float   * d_mask = nullptr;                               // [n][n_kv_max]
    int32_t * d_rows = nullptr;                               // rows flagged for logits
    float  * dx = nullptr;  __half * dxn = nullptr, * dq = nullptr, * datt = nullptr, * dh = nullptr;
    float  * xattn = nullptr;                                 // cross-attention split scratch
    float  * logits = nullptr;  int logits_rows_cap = 0;      // [rows][n_vocab]
    void   * pinned = nullptr;  size_t pinned_bytes = 0;      // host staging (tokens/pos/mask/logits)
    // greedy fast path: one decode step = one captured HIP graph (SURVEY (f)1 + launch-bound inner loop)
    void   * step_dev = nullptr;   void * step_host = nullptr;       // k::DecStep (device / pinned)
    void   * sample_dev = nullptr; void * sample_host = nullptr;     // k::SampleOut (device / pinned)
    void   * filter_scratch = nullptr;
    uint8_t * ban_dev = nullptr;   uint64_t ban_sig = ~0ull;          // static suppress mask + its parameter signature
    // captured forms of the greedy step, index = (cache longer than 64 cells ? 1 : 0) | (chained ? 2 : 0):
    //   long    f16 models: (row, head)-parallel self-attention + plain out projection instead of the fused prologue (device.cpp)
    //   chained no embedding launch — the previous step's pick kernel left the token, the position, the cache head and the next
    //           activation row on the device; valid while the host feeds exactly that token at that position
    // one-row step, one MLP per launch (k::mlp_pair): the hidden row's granules and the launch counter the tags are derived from
    void * mlp_hand = nullptr; unsigned long long * mlp_arrive = nullptr;
    bool pair_off = false; int pair_backoff = 0;                      // the hand-off failed once (two launches from then on) / was slow (two launches for that many steps)
    int  pair_fallbacks = 0, pair_slow_events = 0;                    // steps re-run in the two-launch form / slow hand-offs seen (wmi_pair_status)
    struct StepGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int T = -1; int seen = 0; };
    StepGraph step_graphs[8];                               // + 4: the forms without kernels that wait inside a launch (several transcriptions in flight)
    bool chain_valid = false; int32_t chain_token = 0, chain_pos = 0, chain_head = 0;
    int32_t step_seq = 0;                                             // sequence number of the last greedy step launched
    // device-side draws (beam search, t > 0): decode() leaves the logits rows in d.logits, sample_rows_device() draws from them
    bool    keep_logits_on_device = false;
    void  * draw_dev = nullptr;  void * draw_host = nullptr;          // DecStep[8] | u[8][8] | SampleOut[8][8] (device / pinned)
    void  * draw_scratch = nullptr;
    bool    step_capture_failed = false;                               // a failed capture is not retried
    int     step_seen_T = -1;                                          // encoder length of recent steps (StepGraph::seen counts them)
};

struct State {
    // every entry point that computes on this state holds this for the duration of the call (api_state.cpp: StateScope): calls on ONE
    // state serialise, calls on different states of one context run concurrently on their own streams (the reference's
    // whisper_full_parallel runs whisper_full_with_state on its states from parallel threads: W/whisper.cpp:5837-5858)
    std::recursive_mutex mu;
    int     device = 0;                           // HIP device of the arenas below
    int64_t t_sample_us = 0, t_encode_us = 0, t_decode_us = 0, t_batchd_us = 0, t_prompt_us = 0, t_mel_us = 0;
    int32_t n_sample = 0, n_encode = 0, n_decode = 0, n_batchd = 0, n_prompt = 0, n_fail_p = 0, n_fail_h = 0;
    KVCache kv_self;
    Mel     mel;
    Batch   batch;
    Decoder decoders[8];
    std::vector<float> logits;                    // [n_tokens][n_vocab] host copy (flagged rows only)
    std::vector<Segment> result_all;
    std::vector<int32_t> prompt_past;
    int     lang_id = 0;
    int64_t t_beg = 0, t_last = 0; int32_t tid_last = 0;
    const float * energy = nullptr; int energy_n = 0;   // |x| envelope of the last PCM (view of dev.energy_host)
    const float * energy_bmin = nullptr, * energy_bmax = nullptr;   // its per-256-sample block extrema
    bool energy_on_device = false;
    bool ts_failed = false;                             // the device-side refinement of token times failed: the call returns -9 instead of unrefined times
    bool ts_hold = false;                               // lock-step: emit_window leaves the pending list to the caller (flush_token_timestamps_of)
    bool ts_defer = false; std::vector<int> ts_pending;     // emit_window: segments whose envelope-side refinement runs as ONE device call at the end of the window                      // energy == nullptr: the envelope lives in dev.energy, use ts_refine_device()
    int32_t exp_n_audio_ctx = 0;
    int     enc_n_ctx = 0;                        // n_ctx of the last encode (cross cache extent)
    DeviceState dev;
};

// Lock-step transcription of several independent chunks on one GPU (8 chunks per GPU; same ownership
// split as whisper_full_parallel, W/whisper.cpp:5837-5858: shared read-only weights, one state per worker).
// Activations carry a chunk dimension (encoder GEMMs see M = B * T rows), every chunk has its own self / cross cache.
struct BatchWork {
    int B = 0;                                               // chunk slots allocated (<= 8: rows of the decode GEMV)
    int Tpad = 0; int qk_rows = 0;                            // qk_rows: rows between the chunks of xn (LN1 output) / q / k in the last encode_rows (T, or T rounded up to 16)
    size_t mel_rows = 0;                                      // rows of one chunk's token-major mel image
    __half * mel_t = nullptr, * conv1 = nullptr;              // [B][mel_rows][n_mel], [B][2T+4][S]
    float  * x = nullptr;                                     // [B*T][S] f32 residual stream
    __half * xn = nullptr, * q = nullptr, * k = nullptr, * att = nullptr, * vt = nullptr, * h = nullptr, * enc_out_h = nullptr;
    __half * kvc_k = nullptr, * kvc_v = nullptr;              // cross cache [L][B*T][S]
    __half * self_k = nullptr, * self_v = nullptr;            // self cache  [B][L][n_ctx][S]
    float  * dx = nullptr; __half * dq = nullptr, * datt = nullptr, * dh = nullptr; float * logits = nullptr, * xattn = nullptr;
    // block-quantised models: q8 activation rows [B*T][4S] + block scales, f32 attention outputs (device_q.cpp)
    int8_t * aq = nullptr; float * ads = nullptr; int aq_rows = 0; float * att32 = nullptr, * datt32 = nullptr;
    __half * aq16 = nullptr, * wq16 = nullptr; size_t wq16_elems = 0;  // as DeviceState's
    void   * step_dev = nullptr, * step_host = nullptr, * sample_dev = nullptr, * sample_host = nullptr, * filter_scratch = nullptr;
    // granules of the one-launch front of a layer for lock-step rows (k::front, rows on grid.y): [B][2 S] granules, then 16 words
    // ([0], [1] the launches' tags, [4] the hand-offs' status); front_off: a hand-off failed once, the two launches from then on
    void   * front_hand = nullptr; bool front_off = false; int front_fallbacks = 0, front_backoff = 0;
    int      enc_rows = 0, enc_T = 0;                         // chunk rows / encoder length of the last batched encode
    int32_t  step_seq = 0;                                    // sequence number of the last lock-step decode step
    // the lock-step step as a captured graph, keyed by what its launches depend on (rows, encoder length, chunk rows of the cross
    // cache); eager until the same key has been decoded for a while (capture + instantiate cost more than a window's steps)
    struct RowsGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int nb = -1, T = -1, rows = -1, epoch = -1, seen = 0; bool failed = false; } rows_graph[4];   // [chained + 2 * fronted (the front of the layers as one launch, k::front)]
    // chained steps: the pick kernel of a step leaves every row's next token, position and cache head in step_dev and the next
    // activation row in dx (as the one-row greedy step does, DeviceState::chain_*); a step whose host records say the same for
    // every row starts without the embedding launch (which reads the records over PCIe in front of everything else)
    bool chain_valid = false; int chain_nb = 0; int32_t chain_token[16] = {}, chain_pos[16] = {}; bool chain_row_ok[16] = {};
    std::vector<State *> lanes;                               // lanes[0] is the context's own state (not owned)
    std::vector<std::vector<Segment>> results;                // per chunk of the last wmi_full_batch call
    std::vector<int> redo;                                    // per chunk: 1 if it was re-run alone (temperature fallback)
    // chunks that cannot advance in lock-step (beam search, t > 0, quantised beams ...) run through the general driver on replica
    // contexts: own state and stream, the weight arena borrowed from this context — see full_batch
    std::vector<whisper_context *> replicas; int replicas_wanted = -1;      // -1: default (WMI_BATCH_REPLICAS, 3)
    int groups_wanted = 0, groups_last = 1;                  // lock-step groups side by side (0: default = WMI_LOCKSTEP_GROUPS, 2); how many the last call used
    int64_t t_mel_us = 0, t_encode_us = 0, t_decode_us = 0, t_emit_us = 0; int n_steps = 0, n_chained = 0;
};

} // namespace wmi

namespace wmi {
// The context's `state` member.  The compute code reaches its working set through ctx.state; which state that is depends on the CALLING
// THREAD: a *_with_state entry point (or a lock-step lane, batch.cpp) installs the caller's state for the thread (StateInstall), every
// other thread keeps seeing its own installation or, with none, the context's own state.  Nothing is swapped inside the context, so two
// threads can compute on two states of one context at the same time.
struct StateSlot {
    State * own = nullptr;                       // the state made with the context (whisper_init_from_*), null for the *_no_state constructors
    State * get() const;
    operator State * () const { return get(); }
    State * operator->() const { return get(); }
    State & operator*() const { return *get(); }
    StateSlot & operator=(State * s) { own = s; return *this; }
};
struct StateInstall {                            // RAII: `st` is what `slot` resolves to on this thread until destruction (nests)
    const StateSlot * slot; State * st; StateInstall * prev;
    StateInstall(const StateSlot & sl, State * s);
    ~StateInstall();
    StateInstall(const StateInstall &) = delete; StateInstall & operator=(const StateInstall &) = delete;
};
// the calling thread's installations, to hand them to a helper thread for the duration of a job (pool.cpp: the tasks of pool_run see
// ctx.state as their caller does; the chain lives on the caller's stack, which outlives the job)
StateInstall * state_installs_top();
void state_installs_set(StateInstall * top);
}

struct whisper_context {
    int64_t t_load_us = 0, t_start_us = 0;
    whisper_context_params params{};
    wmi::ModelFile model;
    wmi::Weights   w;
    wmi::StateSlot state;
    int            device = 0;
    bool           host_only = false;   // vocabulary + host logic only (tests); every compute call fails loudly
    // loaded from a payload-less header image (wmi_init_from_header): the arena is allocated, zeroed and laid out but its bytes
    // have not arrived yet — every compute call fails until wmi_arena_commit() says the broadcast / peer copy has landed
    bool           weights_pending = false;
    wmi::BatchWork * batch = nullptr;   // lazily created by wmi_full_batch
    // streams with a hardware queue of their own, made together with the context's first stream and handed to replica contexts later
    // (init_state: it is the ORDER in which the queues are created that decides whether the replicas run beside each other)
    std::vector<hipStream_t> spare_streams;
    float * d_sinc[3] = {nullptr, nullptr, nullptr};   // resampler coefficient tables on the device, by converter (wmi_resample)
    float * vad_res = nullptr;          // pinned host memory the VAD kernel writes {decision, energy_all, energy_last} into (wmi_vad)
    float * dsp_scratch = nullptr; size_t dsp_scratch_bytes = 0;   // grow-only device staging of the host-pointer forms of wmi_vad / wmi_downmix_stereo / wmi_resample
    // guards what belongs to the CONTEXT: the lock-step work set (wmi_full_batch), the DSP scratch, state creation, the probes.  Compute
    // on a state takes the state's own lock (wmi::State::mu), not this one.
    std::recursive_mutex mu;
};

namespace wmi {

// parse + upload a ggml model image onto `device`; with_state = false leaves ctx->state null (whisper_init_*_no_state)
// allow_header: accept the payload-less "wmih" image (only wmi_init_from_header and the in-process pool pass true; every public
// loader of whisper.h rejects it — such a context would otherwise transcribe from whatever the arena allocation held)
whisper_context * init_context(const void * buffer, size_t size, int device, bool with_state, bool allow_header = false);
// false + an error log when the context cannot compute: host-only (tests) or weights still pending (header image not committed)
bool compute_ready(const whisper_context & ctx, const char * who);
bool init_state(whisper_context & ctx, bool replica_state = false, hipStream_t adopt = nullptr);   // replica_state: no spare streams of its own; adopt: the stream to use
// streams with a hardware queue of their own are kept for the life of the process and handed from context to context (device.cpp)
hipStream_t own_queue_stream_get(int device);
void        own_queue_stream_put(int device, hipStream_t s);     // own_hw_queue: a stream that does not share its hardware queue with other streams (replica contexts)
void free_state(whisper_context & ctx);
State * create_state(whisper_context & ctx);      // a further state for the same weights (whisper_init_state); null on failure
void destroy_state(State * st);

// hot path (device)
// defer (full()): no host wait behind the launches, the time goes to t_mel_us / t_encode_us when the events are settled
bool pcm_to_mel(whisper_context & ctx, const float * samples, int n_samples, bool samples_on_device, bool sync = true, bool defer = false);
// the GPU time of the deferred phases into the state's timers; returns the host time (time_us) at which they were complete on the GPU as far
// as the events tell (0: nothing was pending).  wait: synchronise the stream first (else the caller knows it is idle)
int64_t phase_settle(State & st, bool wait);
// lock-step chunks: log-mel of several states' chunks by ONE launch per kernel on the context's stream (k::mel_batch), and — with
// `envelopes` — their |x| envelopes into HBM by one more (the device-resident form the token timestamps of a lock-step call read).
// Entries with n_samples <= 0 are skipped.  Per chunk the kernels' arithmetic is pcm_to_mel's / signal_energy_device's.
bool pcm_to_mel_batch(whisper_context & ctx, const std::vector<State *> & states, const float * const * pcm, const int * n_samples,
                      bool samples_on_device, bool envelopes);
bool set_mel(whisper_context & ctx, const float * data, int n_len, int n_mel);
bool encode(whisper_context & ctx, int mel_offset, bool defer = false);
// lock-step chunks (batch.cpp): rows[r] = lane whose mel feeds chunk row r, seek[r] = its mel frame offset
bool encode_rows(whisper_context & ctx, const std::vector<int> & rows, const std::vector<int> & seek, int audio_ctx);
bool decode(whisper_context & ctx, const Batch & batch);
// block-quantised models (device_q.cpp): the layer loops of encode() / decode() with the quantised kernels
template <typename BUFS> inline k::Q8Rows q8_rows(const BUFS & d, int K) {
    return k::Q8Rows{d.aq, d.ads, d.ads + (size_t) (K / 32) * d.aq_rows, d.aq_rows, d.aq16, d.wq16, d.wq16_elems};
}
bool encode_layers_q(whisper_context & ctx, int T);
// the same layer loop over caller-supplied buffers: nb chunks stacked along M (lock-step, batch.cpp) — activations [nb*T][S],
// V^T [nb][S][Tpad], cross cache [L][nb*T][S]
struct EncBufsQ {
    int T, nb, Tpad;
    float * x; __half * q, * k, * vt, * h; float * att32; float * enc_out; __half * enc_out_h; __half * kvc_k, * kvc_v;
    k::Q8Rows A, A4;
};
bool encode_layers_q_on(whisper_context & ctx, const EncBufsQ & e, hipStream_t s);
void enqueue_rows_step_q(whisper_context & ctx, int nb);
bool decode_layers_q(whisper_context & ctx, int n, int n_kv, int kv_head, int Tc, const std::vector<int> & rows);
void enqueue_greedy_step_q(whisper_context & ctx, int Tc);
// greedy fast path: decode ONE token of sequence 0 at position `pos` and pick the next token on the device
struct StepFilter { bool ban_blank, last_ts, penult_ts; int ts_floor_end, ts_initial_start; };
bool decode_greedy_step(whisper_context & ctx, int32_t token, int32_t pos, const StepFilter & f, whisper_token_data & out);
bool upload_static_ban(whisper_context & ctx, const whisper_full_params & params);
// k draws per row from the filtered distribution of logits row `rows[r]` of the last decode() (keep_logits_on_device), r < n_rows <= 8;
// u [n_rows][k] uniform numbers in [0, 1) from the decoders' generators.  out [n_rows][k].
bool sample_rows_device(whisper_context & ctx, const StepFilter * f, const int * rows, int n_rows, float temperature, int k,
                        const double * u, int tid_default, whisper_token_data * out);
bool wait_for_sample(const k::SampleOut * r, int32_t want, hipStream_t s, int32_t * status = nullptr);   // spin on a pinned, self-tagged result record (device.cpp)
bool fast_path_enabled();
// host worker pool (pool.cpp): fn(0..n_tasks-1) on a few persistent threads + the caller; nested calls run inline
void pool_run(int n_tasks, const std::function<void(int)> & fn);
double bench_greedy_step_chain(whisper_context & ctx, int iters);
double bench_rows_step_chain(whisper_context & ctx, int nb, int iters);
int    step_stamps(whisper_context & ctx, double * out, int cap, bool chained);
// device calls in flight in this process on one device (full(), wmi_full_batch, every *_with_state entry point — a thread counts once however
// they nest): kernels whose workgroups wait for each other INSIDE a launch (k::mlp_pair) are only used while there is one — beside other
// contexts' launch chains their workgroups become resident at different times and the early ones spin (measured: six contexts at once 7.4 ms
// per transcription against 4.5 with the two-launch form, profiles/r05g_*).  Work this count cannot see (another process, the embedder's own
// kernels) is caught by the kernel itself: a slow hand-off is reported and the step changes form (DeviceState::pair_backoff).
struct BusyScope { int dev; bool counted; explicit BusyScope(int device); ~BusyScope(); };      // (counted per device: an in-process pool runs one context per GPU)
int busy_transcriptions(int device);
// |x| envelope of the last PCM on the GPU; the D2H copy runs on a side stream while the encoder works.
// sync = false: state.energy is valid only after signal_energy_wait()
bool signal_energy_device(whisper_context & ctx, int hw, bool sync = true, int via_dma = 0);   // via_dma 1: kernel -> device buffer -> hipMemcpyAsync; 2: kernel -> device buffer now, signal_energy_flush() later
// via_dma 3: the envelope is computed into HBM and STAYS there (lock-step calls): token_level_timestamps() asks ts_refine_device() for the
// window sums and walks instead of reading it (15 MB of PCIe writes per 8-chunk call, ~0.3 ms of whatever runs beside them, are not made)
bool ts_refine_device(State & st, const k::TsTok * in, int n, k::TsOut * out);
struct TsRef { State * st; int seg, j; };
// false = the device call failed: every state involved is marked ts_failed (the envelope lives only in HBM, there is no host form to fall
// back to) and the transcription call reports it as -9 — never silently unrefined t0 / t1
bool flush_token_timestamps(whisper_context & ctx, State & st);      // full.cpp: the pending segments' envelope-side refinement, one device call
bool flush_token_timestamps_of(whisper_context & ctx, const std::vector<State *> & states);    // ... of several states (lock-step chunks) in ONE device call
bool signal_energy_flush(State & st);             // via_dma 2: a THIN copy kernel moves the envelope to the pinned image (lock-step calls: beside the decode steps)
bool signal_energy_wait(State & st);

// host logic (logits filters, sampling, driver)
void process_logits(whisper_context & ctx, Decoder & dec, const whisper_full_params & params, float temperature);
whisper_token_data sample_token(whisper_context & ctx, Decoder & dec, bool best);
std::vector<whisper_token_data> sample_token_topk(whisper_context & ctx, Decoder & dec, int k, bool count = true);   // count = false: the caller adds to n_sample (parallel decoders)
void sequence_score(const whisper_full_params & params, Sequence & seq);
Grammar grammar_init(const whisper_grammar_element ** rules, size_t n_rules, size_t i_start_rule);
void grammar_penalise(const whisper_context & ctx, const Grammar & g, float penalty, std::vector<float> & logits);
void grammar_accept_token(const whisper_context & ctx, Grammar & g, int32_t token);
std::vector<int32_t> tokenize(const Vocab & vocab, const std::string & text);
int  lang_auto_detect(whisper_context & ctx, int offset_ms, float * lang_probs);
int  full(whisper_context & ctx, whisper_full_params params, const float * samples, const float * d_samples, int n_samples);
// several independent chunks in lock-step (batch.cpp); results per chunk in ctx.batch->results
int  full_batch(whisper_context & ctx, whisper_full_params params, const float * const * pcm, const int * n_samples, int n_chunks, bool on_device);
void free_batch(whisper_context & ctx);
int  ensure_replicas(whisper_context & ctx, int n);     // create up to n replica contexts now; returns how many exist (<= n)
void trim_replicas(whisper_context & ctx, int keep);         // release the replicas beyond `keep` (states, streams back to the pool)
// segment emission of one decoded window: updates prompt_past and appends to state.result_all (W/whisper.cpp:5682-5796)
void emit_window(whisper_context & ctx, State & st, const whisper_full_params & params, int seek, const std::vector<int32_t> & prompt,
                 size_t n_prompt_init, const Decoder & best);
std::vector<float> signal_energy(const float * signal, int n_samples, int hw);
void token_level_timestamps(whisper_context & ctx, State & st, int i_segment, float thold_pt, float thold_ptsum);
float seq_sum_f32(const float * p, int n);            // the left-to-right f32 sum of p[0..n), bit for bit, in blocks (full.cpp)
int  wrap_segment(whisper_context & ctx, State & st, int max_len, bool split_on_word);

int         lang_id(const char * lang);
const char *lang_str(int id);
const char *lang_str_full(int id);
int         lang_max_id();
int         lang_count();

} // namespace wmi
