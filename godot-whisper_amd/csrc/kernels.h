// Launch prototypes of the gfx950 kernels (definitions in k_*.hip).
// Conventions: activations are token-major ([row][feature], feature contiguous); every matrix
// operand is K-contiguous; f16 = IEEE half (the reference's rounding points).
#pragma once
#include <memory>
#include <vector>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <atomic>
#include <cstdint>

namespace wmi { namespace k {

// Kernels with more than 48 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize once per device (a context per
// GPU, launches from several host threads).  The attribute is set to the whole 160 KB of a CU — it permits, the launch decides —
// so it never depends on the sizes seen so far; `mask` = one bit per device, one static per kernel instantiation.
inline void allow_full_lds(const void * kernel, std::atomic<uint64_t> & mask) {
    int dev = 0;
    (void) hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return;
    (void) hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    mask.fetch_or(bit, std::memory_order_release);        // two threads may both set it: idempotent
}

// f32 -> f16 with the value pinned in a register first.  Without the empty asm the AMDGPU back end folds SOME of the
// `cvt(mul)` / `cvt(add)` pairs of an unrolled epilogue into v_fma_mixlo_f16 — one rounding instead of the reference's
// two (f32 op, then f16 store) — and which elements get the fused form depends on their slot in the tile, so identical
// inputs gave results that differed in the last f16 bit with the position of a row in M (found by the lock-step
// bit-exactness test; scratch/lab/pos_test.hip reproduces it).  -ffp-contract=off does not stop this fold.
#if defined(__HIPCC__)
__device__ __forceinline__ float pin_f32(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ __half f2h(float x) { return __float2half_rn(pin_f32(x)); }
// Time index inside an encoder V^T row ([S][Tpad], written by the q|k|v GEMM epilogues, read only by the encoder attention): bits 2
// and 3 of t are swapped, i.e. within every 16 keys the four-key groups sit in the order 0, 2, 1, 3.  A lane of the 32x32x16 MFMA
// holds the soft-max numerators of keys 8j + 4g + r (g = lane / 32): with this order the eight V values that pair with them —
// keys {4g..4g+3} and {8+4g..8+4g+3} of a 16-key slice — are ONE aligned 16-byte piece, so the tile goes global -> LDS by DMA
// untouched and a fragment is a single ds_read_b128.
__host__ __device__ __forceinline__ int vt_pos(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }
#endif

// Compute units of the CURRENT device, rounded down to a multiple of the 8 XCDs (persistent grids: one workgroup per CU).  One slot per
// device ordinal — a process may hold contexts on several GPUs (wmi_pool_*), and a count cached from the first device a kernel saw
// would size the grids of all the others.
inline int cu_count_x8() {
    static std::atomic<int> cache[64];
    int dev = 0;
    (void) hipGetDevice(&dev);
    std::atomic<int> & c = cache[dev & 63];
    int n = c.load(std::memory_order_relaxed);
    if (n > 0) return n;
    n = 256;
    (void) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    n = n > 8 ? n & ~7 : 8;
    c.store(n, std::memory_order_relaxed);
    return n;
}

// Kernel arguments, all requested at the kernel's first instructions.  hipcc places the s_load of a by-value argument struct's field in
// the basic block that first uses it: a kernel with early branches (probe switch, helper workgroups, optional operands) walks through
// half a dozen DEPENDENT scalar-load round trips to the kernarg segment (~0.25-0.3 us each, cold in the scalar cache at every launch)
// before its first vector load goes out — 1.7 us of a 5.8 us launch in k_qrows (in-kernel stamps, profiles/r05c_*).  Naming a field as an
// asm input makes it live here, so the loads of everything named leave together and one wait covers them.
#if defined(__HIPCC__)
#define WMI_ARG_NOW(x) asm volatile("" :: "s"(x))
#endif

// ---------------------------------------------------------------- in-kernel time stamps (probe: wmi_step_stamps)
// Body / boundary split of a dependent chain of launches: when stamping is on, lane 0 of every wavefront of a stamped launch
// writes (s_memrealtime at its first instruction, s_memrealtime behind its last store) to
// base[(slot * STAMP_WAVES + wave index) * 4] (+ two optional mid points); the host takes min start / max end per launch.  Off (base == null) costs one
// scalar compare per kernel.  The launchers draw their slot from stamp_next() in launch order; the switch is THREAD-LOCAL: only launches
// enqueued by the thread that called stamp_enable() are stamped (pool workers and replica contexts launching beside a probe see "off").
constexpr int STAMP_WAVES = 4096;                       // wavefront records per launch
struct Stamp { unsigned long long * base; int slot; };
void  stamp_enable(unsigned long long * base);          // null = off; resets the slot counter
Stamp stamp_next();                                     // {base, slot++} or {null, 0}
int   stamp_count();
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long stamp_t0(const unsigned long long * base) { return base ? wall_clock64() : 0ull; }
__device__ __forceinline__ void stamp_end(unsigned long long * base, int slot, int widx, unsigned long long t0,
                                          unsigned long long tm1 = 0, unsigned long long tm2 = 0) {
    if (base && (threadIdx.x & 63) == 0 && widx < STAMP_WAVES) {
        unsigned long long * s = base + ((size_t) slot * STAMP_WAVES + widx) * 4;       // start, two optional mid points, end
        s[0] = t0; s[1] = wall_clock64(); s[2] = tm1; s[3] = tm2;
    }
}
#endif

// ---------------------------------------------------------------- mel (k_mel.hip)
// pcm_pad: [200 reflect | n_samples | zeros] ; frames [0, n_fft_frames) get an FFT, the rest up to
// n_len the constant log10(1e-10).  mel: [n_mel][n_len] f32.  gmax: ordered-int encoded running max.
void mel_pad(const float * pcm, int n_samples, float * pcm_pad, int n_pad_total, hipStream_t st, int * gmax_reset);   // also resets mel_frames' gmax
void mel_frames(const float * pcm_pad, int n_valid, int n_fft_frames, int n_len, int n_mel,
                const float * filters, const int32_t * ranges, const float * taps, float * mel, int * gmax, hipStream_t st);
void mel_normalize(float * mel, int n, const int * gmax, hipStream_t st);
// lock-step chunks (batch.cpp): the mel kernels and the envelope kernel of up to 16 chunks in ONE launch each — per chunk the samples
// (device), the padded image, the mel image, the running-maximum word and the envelope with its block extrema
struct MelBatch { const float * pcm[16]; float * pad[16]; float * mel[16]; int * gmax[16]; float * energy[16]; float * bmin[16]; float * bmax[16]; int n[16]; };
void mel_batch(const MelBatch & b, int nb, int n_mel, const float * filters, const int32_t * ranges, const float * taps, hipStream_t st);
void signal_energy_batch(const MelBatch & b, int nb, int hw, hipStream_t st);
// token-major f16 slice for the conv front-end: out[r][c], r in [0, rows_total), row r holds frame
// (offset + r - 1); rows outside [1, n_frames] and frames >= n_len are zero.
// lock-step chunks: the slices of up to 16 chunks in one launch (chunk c -> rows c * rows_total ..), and the guard rows between stacked chunks
struct MelSliceBatch { const float * mel[16]; int n_len[16]; int offset[16]; };
void mel_slice_batch(const MelSliceBatch & mb, int nb, int n_mel, int n_frames, __half * out, int ld, int rows_total, hipStream_t st);
void fill_zero_strided(void * p, size_t bytes, size_t stride_bytes, int count, hipStream_t st);
void mel_slice(const float * mel, int n_len, int n_mel, int offset, int n_frames, __half * out, int ld,
               int rows_total, hipStream_t st);

// |x| envelope for the token-level timestamp heuristics (W/whisper.cpp:6352-6366), bit-identical to the CPU loop
// out[n] plus the minimum / maximum of each 256-sample block of out (bmin, bmax: ceil(n / 256) entries)
void signal_energy(const float * pcm, int n, int hw, float * out, float * bmin, float * bmax, hipStream_t st);
// Token-level timestamps, the envelope side (W/whisper.cpp:6500-6590, csrc/full.cpp token_level_timestamps): per token the sequential f32 sum of
// the envelope over its window, the threshold, and the four walks the reference may take from the token's start / end sample.  One
// wavefront per token; every result equals the host loop's (sequential sum in index order; walks stop on the same sample).
struct TsTok { int32_t s0, s1, a0, a1;                      // start / end sample of the token, window [a0, a1) of the sum
               const float * en; int32_t n_samples, ext_off; };   // optional (en != null): this token's own envelope, its length, and the offset of its block minima (maxima: + n_samples / 256 + 2) — tokens of several chunks in one launch
struct TsOut { float sum, thold; int32_t e0, e1;            // e0 = en[s0] > thold, e1 = en[s1] > thold
               int32_t w_down_above_s0, w_up_below_s0, w_up_above_s1, w_down_below_s1; };   // walk results (bounds: 0, s1, n - 1, 0)
void ts_refine(const float * en, const float * bmin, const float * bmax, int n_samples, const TsTok * in, TsOut * out, int n_tok, hipStream_t st);
// device -> pinned host (or anywhere) by `wgs` workgroups only: a deliberately slow copy that keeps the PCIe write queue short
void copy_thin(const void * src, void * dst, size_t bytes, int wgs, hipStream_t st);

// host-adjacent DSP of the streaming node: stereo -> mono and the energy VAD, bit-identical to the host's C++
// (src/speech_to_text.cpp:45-51, 53-104).  res = {no-activity decision, energy_all, energy_last}
void downmix_stereo(const float * frames, int n_frames, float * out, hipStream_t st);
void vad_window(const float * x, int n, int n_last, float alpha, bool filter, float vad_thold, float * res, hipStream_t st);

// SINC resampler (k_resample.hip): src_simple(SRC_SINC_FASTEST = 2 | SRC_SINC_MEDIUM_QUALITY = 1, one channel) of libsamplerate as the host
// calls it (src/speech_to_text.cpp:16-43), bit-identical to the sequential CPU code.  resample_plan replays the converter's index
// state machine on the host (how many frames come out, whether the position recurrence has a closed form), resample_launch
// computes every output frame in its own thread.
struct Stepper;
struct ResamplePlan {
    int error = 0;                       // 0, or libsamplerate's error number negated (-6 ratio, -10 converter, -21 length check), -30 ratio unsupported
    long long n_out = 0, n_used = 0;     // output_frames_gen, input_frames_used
    int half_len = 0, index_inc = 0, increment = 0;
    double float_inc = 0.0, out_scale = 0.0;
    bool need_table = false;             // the (pos, frac) table of resample_table() has to be on the device
    std::shared_ptr<Stepper> stepper;
};
bool sinc_table(int converter, const float ** coeffs, int * count, int * increment);
ResamplePlan resample_plan(long long n_in, long long out_cap, double ratio, int converter);
void resample_table(const ResamplePlan & pl, const int ** pos, const double ** frac);       // host arrays, n_out (+1) entries
void resample_positions(const ResamplePlan & pl, long long n, long long * pos, double * frac);   // first n positions (tests)
void resample_launch(const ResamplePlan & pl, const float * d_in, long long n_in, float * d_out, const float * d_coeffs,
                     const int * d_pos, const double * d_frac, hipStream_t st);

// ---------------------------------------------------------------- GEMM (k_gemm.hip)
enum Epi : int {
    EPI_F16_BIAS = 0,       // C f16 = acc + bias
    EPI_F16_BIAS_GELU,      // C f16 = gelu16(acc + bias)
    EPI_F32_BIAS_RESID,     // C f32 = acc + bias + resid
    EPI_CONV2,              // x f32 = gelu16(acc + bias) + pe ; aux f32 = gelu16(acc + bias)
    EPI_QKV_ENC,            // q | k | v^T split (encoder)
    EPI_QKV_DEC,            // q*s | k*s -> cache | v -> cache (decoder self-attention)
    EPI_CROSS_KV,           // per decoder layer: k*s | v+b into the cross cache
    EPI_Q_SCALED,           // C f16 = (acc + bias) * scale  (decoder cross-attention query)
};
struct GemmArgs {
    const __half * A;  int lda;     // [M][K]
    const __half * W;  int ldw;     // [N][K]
    int M, N, K;                    // K multiple of 32 (weights zero-padded)
    const float * bias;             // [N] or null
    void *  C;   int ldc;           // primary output
    const float * resid; int ldr;   // EPI_F32_BIAS_RESID / EPI_CONV2 (pe)
    void *  aux; int ldaux;         // second output (EPI_CONV2: embd_conv; QKV: k / cache k)
    void *  aux2; int ldaux2;       // third output (QKV: v^T / cache v)
    float   scale;                  // q/k scale
    int     S;                      // split width for the QKV / cross epilogues
    int64_t layer_stride;           // EPI_CROSS_KV: elements between layers in the cross cache
    int     rows_per_chunk;         // EPI_QKV_ENC, batched encode: M = chunks * rows_per_chunk (0: one chunk)
    int64_t chunk_stride_aux2;      //   elements between the chunks' V^T images
    int     no_glds;                // debug: keep 128x128 tiles on the register-staged loop
    unsigned long long * probe;     // probe (wmi_bench_kernel 7): per workgroup {entry, first tile landed, K loop done, epilogue done, SE/CU id}
};
void gemm(int epi, const GemmArgs & a, hipStream_t st);
// Probe (wmi_encoder_gemm_stamps): while a log is installed on the calling THREAD, every gemm() launch that has no probe of its own gets a
// slice of `buf` as GemmArgs::probe (5 words per workgroup: entry ... done, wall-clock ticks) and an entry here — the in-situ duration of
// each GEMM of an encoder pass is then max(done) - min(entry) over its workgroups, with the cache state the pass itself leaves.
struct GemmLogEntry { int epi, M, N, K; size_t off; int cap; };
struct GemmLog { unsigned long long * buf = nullptr; size_t cap_words = 0, used = 0; std::vector<GemmLogEntry> entries; };
void gemm_log_install(GemmLog * log);       // null = off
// k_gemm8.hip: the eight-wavefront ping-pong form for the big grids (M = chunks x 1500): BM x 256 tiles, bm in {96, 128, 160, 192, 256},
// swapped = transposed accumulator fragments (row-major f16 / f32 epilogues), ks = k extent of a slot.  Bit-identical to gemm().
// false = not served (N % 256, K % 64, epilogue): the caller keeps gemm().
bool gemm8(int epi, int bm, bool swapped, const GemmArgs & a, hipStream_t st, int ks = 64);

// ---------------------------------------------------------------- LayerNorm (k_norm.hip)
// y = (x - mean) / sqrt(var + eps) * g + b ; one wave per row. out16 and/or out32 may be null.
void layernorm(const float * x, int rows, int S, const float * g, const float * b, float eps,
               __half * out16, float * out32, hipStream_t st, int rows_per_chunk_in = 0, int rows_per_chunk_out = 0);      // > 0: output row = chunk * out + t for input row chunk * in + t

// ---------------------------------------------------------------- attention (k_attn.hip)
// encoder: q,k [T][S] f16 ; vt [S][Tpad] f16 ; out [T][S] f16 ; scale applied to q.k before softmax
// B > 1: B chunks back to back (q,k,out [B][T][S]; vt [B][S][Tpad]), one grid.z slice each
// out32 != null: the result is written as f32 to out32 instead (models whose out-projection is block-quantised quantise that
// f32 tensor directly, as the reference does); same for the decoder kernels below
// second form (k_attn_enc.hip): 32-row wavefronts; one_sweep = running maximum, else exact maximum first; split = four key groups
void attn_encoder2(const __half * q, const __half * k, const __half * vt, int T, int Tpad, int S, int H, __half * out, hipStream_t st,
                   int B, float * out32, bool one_sweep, bool split, int qk_chunk_rows = 0);
void attn_encoder(const __half * q, const __half * k, const __half * vt, int T, int Tpad, int S, int H,
                  float scale, __half * out, hipStream_t st, int B = 1, float * out32 = nullptr,
                  int qk_chunk_rows = 0);     // rows between the chunks of q and k (0 = T; lock-step chunks start on 16-row boundaries, batch.cpp)
// decoder: one (token, head) per workgroup.  kc/vc: [n_kv][S] caches (this layer), mask: [n][ld_mask] or null
void attn_decoder(const __half * q, int n, int S, int H, const __half * kc, const __half * vc, int n_kv,
                  const float * mask, int ld_mask, __half * out, hipStream_t st,
                  const int32_t * n_kv_dev = nullptr, int n_kv_max = 0,    // n_kv_dev: read n_kv on the device (graph replay)
                  float * out32 = nullptr);

// decoder cross-attention split over the key axis (3 small launches, NS x H x n workgroups); same numerics
void attn_cross_split(const __half * q, int n, int S, int H, const __half * kc, const __half * vc, int T,
                      float * scratch, __half * out, hipStream_t st, int64_t kv_row_stride = 0, float * out32 = nullptr);
// same without the combine launch: the consumer GEMV combines the partials in its prologue (GemvArgs::comb_*)
void attn_cross_split_partials(const __half * q, int n, int S, int H, const __half * kc, const __half * vc, int T,
                               float * scratch, const float ** part_o, const float ** part_l, const float ** part_m, int * ns, hipStream_t st,
                               int64_t kv_row_stride = 0);   // row i reads kc/vc + i * kv_row_stride (lock-step chunks)
// part_m: null when the partials are relative to the row's global maximum (two-launch form), else the slice maxima the
// consumer rescales by (one-launch form, k_xattn_fused) — pass it on as GemvArgs::comb_m / to attn_cross_combine
// the same with the query projection folded into the score kernel: q = (W_cq . LN(x32) + b_cq) * qscale is recomputed per
// (slice, head) workgroup (bit-identical to gemv + EPI_Q_SCALED); saves one launch per decoder layer
// where the one-launch form of the cross-attention puts a step's partials, and its grid: for the form whose query projection is a
// block-quantised matrix (k_quant.hip: qattn_cross_qsplit_partials).  fused = false: the key slices are too long for one launch
struct XattnPlan { int ns, ks; float * pmax, * part_l, * part_o; bool fused, head_major; };
XattnPlan attn_cross_plan(int n, int H, int T, float * scratch);
void attn_cross_qsplit_partials(const float * x32, const float * ln_g, const float * ln_b, float eps, const __half * wq,
                                const float * bq, float qscale, int n, int S, int H, const __half * kc, const __half * vc, int T,
                                float * scratch, const float ** part_o, const float ** part_l, const float ** part_m, int * ns, hipStream_t st,
                                int64_t kv_row_stride = 0);
size_t attn_cross_scratch_floats(int n, int H, int T);

// ---------------------------------------------------------------- decoder small-batch (k_dec.hip)
void dec_embed(const int32_t * tokens, const int32_t * pos, int n, int S, const __half * te, const float * pe,
               float * x, hipStream_t st);
// per-workgroup partial statistics of a row of filtered logits (k_sample.hip: k_filter_stats writes 64 of them per row; the one-row
// vocabulary projection writes one per workgroup from its epilogue, GemvArgs::fs_*): maxima with first-index tie-break over all /
// text / timestamp tokens, and sums of exp(l - all.v) — combined exactly by k_filter_pick (online soft-max identity)
struct FsMaxIdx { float v; int i; };
struct FsPartial { FsMaxIdx all, txt, ts; float sum, sum_ts; float pad[2]; };
constexpr int FS_MAX_PARTS = 768;                 // partials of the fused form (= the vocabulary projection's workgroup cap)
// rows <= 8 "GEMV" path: out[i][o] = sum_k W[o][k] * a[i][k] with the same epilogues as the GEMM.
// If ln_g != null the A operand is LayerNorm(x32) computed in the prologue (fused), else a16.
struct GemvArgs {
    const float * x32; const float * ln_g; const float * ln_b; float eps;   // fused-LN input [n][K]
    const __half * a16;                                                     // or plain f16 input [n][K]
    int n, K, N;
    const __half * W; const float * bias;
    int epi;                                  // Epi (F16_BIAS, F16_BIAS_GELU, F32_BIAS_RESID, QKV_DEC, Q_SCALED) or EPI_LOGITS
    void * C; int ldc;
    const float * resid; int ldr;
    void * aux; int ldaux; void * aux2; int ldaux2;
    float scale; int S;
    const int32_t * rows;                     // optional row gather for the A operand (logits)
    const int32_t * row_off;                  // optional device scalar: aux/aux2 row offset (KV cache head), graph replay
    const float * comb_o; const float * comb_l; int comb_ns;   // optional: A operand = combined split cross-attention partials
    const float * comb_m;                     // slice maxima the partials are relative to (null: the row's global maximum)
    // optional (n == 1): A operand = self-attention output computed in the prologue from q and this layer's KV cache
    const __half * sa_q; const __half * sa_k; const __half * sa_v; const int32_t * sa_nkv; int sa_cap;
    // lock-step chunks (lanes != 0): row r belongs to chunk r with its own KV cache and step record.
    //   aux/aux2 row = row_off[r * step_stride] (no "+ r"), cache base + r * cache_row_stride ; sa_nkv[r * step_stride] ;
    //   sa_k / sa_v + r * cache_row_stride
    int lanes; int step_stride; int64_t cache_row_stride;
    // chained greedy steps: mirror *step_copy_src (DecStep in pinned host memory) into *step_copy_dst by an extra workgroup
    // (one-row f16 rows x plain projection + residual only: the last mlp.2 of the step)
    const void * step_copy_src; void * step_copy_dst;
    // chained lock-step steps: the rows' records (words 4.. of each DecStep) mirrored by an extra workgroup of the vocabulary
    // projection on the matrix cores (k_rows_mfma); ask gemv_rows_carries_mirror() first, the other row kernels ignore these
    const void * rows_mirror_src; void * rows_mirror_dst;
    // weight prefetch for the NEXT launch of a dependent chain (k_qrows): pf_groups row groups of pf_group_bytes each, contiguous at
    // pf_ptr; workgroup i touches one dword per 128-byte line of the groups g = i (mod grid) — the workgroup of the next launch
    // that streams group g sits on the same XCD when both grids are multiples of 8
    const void * pf_ptr; uint32_t pf_group_bytes; uint32_t pf_groups;
    unsigned long long * stamps; int stamp_slot;      // probe (stamp_next()): filled in by gemv() itself
    // greedy step, EPI_LOGITS at one row: the logit filters' statistics pass folded into the epilogue — fs_part[workgroup] gets the
    // partial of the rows the workgroup produced (fs_step: DecStep of the row, fs_ban: static ban bytes); gemv_fused_parts() tells
    // how many partials this launch will write (0: the arguments do not take the fused path — run filter_argmax's own pass)
    const uint8_t * fs_ban; const void * fs_step; FsPartial * fs_part;
    // k_qrows, K split over workgroups (ksplit = 2: mlp.2 of the wide models — N = S is only S / 32 row groups, K = 4 S is 4 S / 64 tiles
    // each): workgroup (row group, half) multiplies its half of K; the lower half runs the epilogue as usual (bias, residual, in
    // place), the upper half leaves its sums in kpart [n][N] f32.  The launches that read that row next take them as `pend` and add
    // them — x = C + kpart, one f32 addition per element, the same in every consumer: the LayerNorm prologue of the next projection /
    // of the vocabulary projection (x32 + pend) and the residual of the next out projection (resid + pend), which writes the row whole again.
    int ksplit; float * kpart;
    const float * pend;
};
int gemv_fused_parts(const GemvArgs & a);
bool gemv_rows_carries_mirror(const GemvArgs & a);
bool gemv_rows_take_self_attention(const GemvArgs & a);      // lock-step rows: will gemv() run this out projection with the self-attention in its prologue?
// lock-step chunks: single-token self-attention of n rows, row r against the cache at kc/vc + r * cache_row_stride with
// n_kv[r * step_stride] cells; same arithmetic as the fused prologue of gemv (GemvArgs::sa_*).  out [n][K] f16
void self_attn_rows(const __half * q, int n, int K, const __half * kc, const __half * vc, int64_t cache_row_stride,
                    const int32_t * n_kv, int step_stride, int cap, __half * out, hipStream_t st, float * out32 = nullptr,
                    bool long_cache = false,    // long_cache: the host knows a row has > 64 cells: four wavefronts per (row, head)
                    const void * mirror_src = nullptr, void * mirror_dst = nullptr);   // chained lock-step steps: + one workgroup per row copying
                                                // words 4.. of the row's DecStep from pinned host memory to the device record (short-cache form only)
// split cross-attention partials -> out [n][S] f16 (the separate form of GemvArgs::comb_*)
void set_xattn_probe_skip(int mask);      // probe only
void attn_cross_partials_layout(int n, int H, int T, float * scratch, const float ** po, const float ** pl, const float ** pm, int * pns);
void attn_cross_combine(const float * part_o, const float * part_l, const float * part_m, int ns, int n, int S, int H, __half * out, hipStream_t st,
                        float * out32 = nullptr);
enum { EPI_LOGITS = 100 };                    // C f32 [n][N] = acc
void gemv(const GemvArgs & a, hipStream_t st);

// One decoder MLP of the one-row step as ONE launch: x += W2 . gelu(W1 . LN(x) + b1) + b2 (k_mlp_pair, k_dec.hip).  The hidden row goes
// from the workgroups that produce it to every workgroup of the same launch through data-tagged 8-byte granules ({two f16, tag}, written
// and read past the caches): no launch boundary between the two projections.  `hand` holds 2 S granules (16 S bytes, zeroed once),
// `epoch` two 32-bit words (zeroed once; the launches' tags, see the kernel).  false = shape not covered (S > 1024: measured slower at S = 1280; or the launch's workgroups
// would not all be resident at once): run the two projections as two launches.
struct MlpPairArgs {
    const float * x; const float * ln_g, * ln_b; float eps; int S;
    const __half * W1; const float * b1; const __half * W2; const float * b2;
    uint32_t * epoch; int par; void * hand;                    // epoch[2] (zeroed once), par = 0 / 1 alternating from launch to launch
    // sticky status word of the hand-off (zeroed once, cleared by the host after it has acted): PAIR_FAULT_TIMEOUT a consumer gave up waiting
    // (its rows of this launch are garbage), PAIR_FAULT_PARITY two launches in a row with the same `par` (stale granules pass the tag test),
    // PAIR_SLOW a sweep needed more than PAIR_SLOW_POLLS polls (correct, but something else owns the GPU: the two-launch form is the faster one).
    // The step's pick kernel reports the word in the sequence tags of SampleOut (ChainNext::fault), the host re-runs / switches form.
    uint32_t * fault; uint32_t spin_cap;                       // spin_cap: polls before a consumer gives up (0 = 1 << 20, about a second)
    int withhold;                                              // tests: wavefront `withhold - 1` never publishes its granules; < 0: wavefront `-withhold - 1` publishes ~0.2 ms late (0 = off)
    const void * step_copy_src; void * step_copy_dst;          // chained greedy steps: see GemvArgs::step_copy_src
    __half * h_out;                                            // optional plain copy of the hidden row (tests / debugging)
};
enum : uint32_t { PAIR_FAULT_TIMEOUT = 1u, PAIR_FAULT_PARITY = 2u, PAIR_SLOW = 4u };
constexpr uint32_t PAIR_SLOW_POLLS = 64;
// decided ONCE per step for all its layers (the launches' parity must alternate: a layer that alone fell back to two launches would break
// it): shape covered, and the widest launch of the step (+ 1 workgroup with the step-record mirror) resident at once on the CURRENT device
bool mlp_pair_usable(int S, bool with_mirror);
void mlp_pair(const MlpPairArgs & a, float * x_inout, hipStream_t st);
// The front of a decoder layer of the one-row step as one launch (k_dec.hip: k_front): LayerNorm + q|k|v with EPI_QKV_DEC's stores, the
// self-attention over a cache of <= 64 cells computed once per head, the out projection + residual (in place: xout = x).  gq: 3 S / 2
// granules, ga: S / 2 granules (8 bytes each, zeroed once); epoch / par / fault / spin_cap / withhold as in MlpPairArgs (own epoch words).
struct FrontArgs {
    const float * x; float * xout; const float * ln_g, * ln_b; float eps; int S;
    const __half * Wqkv; const float * bqkv; float scale; __half * q16; __half * ck, * cv;       // ck / cv: the layer's self cache [cell][S]
    const int32_t * kv_head, * n_kv; int cap;                                                    // the step record's cache head / key count; cells in the cache
    const __half * Wo; const float * bo;
    unsigned long long * gq, * ga; uint32_t * epoch; int par; uint32_t * fault; uint32_t spin_cap; int withhold;
    // lock-step rows (grid.y = row): row y has its own activation row (x + y S), q16 row, self cache (+ y cache_row_stride), step record
    // (+ y step_stride words) and granules (+ y 2 S); rows = 0 / 1: one row
    int64_t cache_row_stride; int step_stride; int rows;
};
// The back of the cross-attention of the one-row step as one launch (k_attn.hip: k_xback): LN + cross query + key slices, the combine of a
// head's slices (once per head) and the out projection + residual (in place: xout = x).  gp: H * ns * 66 granules of partials, ga: S / 2
// granules of the attention row (8 bytes each, zeroed once); epoch / par / fault / spin_cap / withhold as in MlpPairArgs (own epoch words).
// ks / ns / pmax / part_o / part_l are filled in by xback() from the cross-attention's scratch layout.
struct XbackArgs {
    const float * x; float * xout; const float * ln_g, * ln_b; float eps; int S;
    const __half * wq; const float * bq; float qscale; const __half * kc, * vc; int T, ks, ns;
    float * pmax, * part_o, * part_l;
    const __half * Wo; const float * bo;
    unsigned long long * gp, * ga; uint32_t * epoch; int par; uint32_t * fault; uint32_t spin_cap; int withhold;
    // lock-step rows (grid.z = row): row z has its own activation row (x + z S), cross cache (+ z kv_row_stride) and granules
    // (+ z XBACK_ROW_GRANULES); rows = 0 / 1: one row
    int64_t kv_row_stride; int rows;
};
constexpr int XBACK_ROW_GRANULES = 8 * 8 * 66 + 256;        // a row's partials (<= 8 heads x 8 slices x 66) + its attention row (S / 2 <= 256)
bool xback_usable(int S, int H, int T, int rows = 1);
void xback(XbackArgs a, int H, float * scratch, hipStream_t st);
bool front_usable(int S, int rows = 1);
void front(const FrontArgs & a, hipStream_t st);
// A/B switches of the launch paths that are read from the environment: once per process (reload_knobs(): lab scripts that flip them between
// probe calls of one process, exported as wmi_reload_knobs — not while a transcription runs on another thread)
struct Knobs { bool no_mlp_pair; int pair_wpb; int sa_wpb; bool gemv1_wide_generic; bool host_draws; bool debug_sync; int pair_withhold; uint32_t pair_spin_cap; bool no_front; int front_withhold; bool no_xback; int xback_withhold; int front_wpb; };
const Knobs & knobs();
void reload_knobs();
void set_attn_one_group(bool on);              // encoder attention: never split the keys over two wave groups (bit-identical for any batch)
bool rows_valu_enabled();
void set_rows_valu(bool on);                  // lock-step rows: true = VALU kernel (bit-identical to the one-row path), false = MFMA
int  mode_epoch();                             // bumped by every run-time switch that changes which kernels a step launches: a captured
void bump_mode_epoch();                        // step graph is replayed only under the epoch it was captured in

// ---------------------------------------------------------------- device-side logit filters + greedy pick (k_sample.hip)
// One decode step's dynamic inputs; lives in device memory, refreshed by a 64-byte H2D copy per step so that
// the captured HIP graph of the step stays static.
struct DecStep {
    int32_t token, pos, n_kv, kv_head;
    int32_t flags;                 // bit0 ban eot + " " (initial & suppress_blank) ; bit1 last token was a timestamp ; bit2 penultimate too
    int32_t space_id, eot, beg, n_vocab;
    int32_t ts_floor_end;          // timestamps in [beg, ts_floor_end) are banned (monotonic rule) ; = beg when inactive
    int32_t ts_initial_start;      // timestamps in [ts_initial_start, n_vocab) are banned (max_initial_ts) ; = n_vocab when inactive
    int32_t seq;                   // step sequence number, echoed in SampleOut::seq (the host polls pinned memory for it)
    float   temperature;           // > 0: logits are divided by it before the filters (W/whisper.cpp:4500-4506); 0 = as they are
    int32_t pad[3];
};
// Two self-tagged 16-byte halves: the pick kernel writes each half to pinned host memory with ONE 16-byte store, the host accepts
// a half when its tag (seq0 / seq) is the step it waits for — no fence, no read of host memory by the device (the fenced form,
// record + __threadfence_system + seq + fence, cost 0.65 us more per step: profiles/r03a_chain_lab_boundary_and_variants.txt)
struct alignas(16) SampleOut { int32_t id, tid; float p; int32_t seq0; float plog, pt, ptsum; int32_t seq; };
// logits [n_vocab] -> filtered soft-max statistics and the arg-max token (W/whisper.cpp:4493-4830 at temperature 0)
// n_rows > 1: lock-step chunks — logits [n_rows][n_vocab], step[n_rows], out[n_rows]
// chain (one row): the pick kernel also prepares the NEXT greedy step on the device — token = the pick, pos / n_kv / kv_head + 1 in
// *step_rw, and the next activation row x = te[pick] + pe[pos + 1] — so that the next step needs no embedding launch
struct ChainNext { DecStep * step_rw; const __half * te; const float * pe; float * x; int S; int n_pos;
                   const uint32_t * fault; };             // MlpPairArgs::fault of the step's launches (nullptr: none): reported in the tags, see SAMPLE_TAG_*
// SampleOut::seq0 / seq = the step's sequence number (SAMPLE_SEQ_MASK bits) | status of the step's in-launch hand-offs
constexpr int32_t SAMPLE_SEQ_MASK = 0x0FFFFFFF, SAMPLE_TAG_FAULT = 0x40000000, SAMPLE_TAG_SLOW = 0x20000000;
// fused_parts > 0 (one row): scratch already holds that many partials (GemvArgs::fs_part): only the pick kernel runs
void filter_argmax(const float * logits, const uint8_t * static_ban, const DecStep * step, SampleOut * out, void * scratch, hipStream_t st,
                   SampleOut * out_host = nullptr, int n_rows = 1, const ChainNext * chain = nullptr, int fused_parts = 0);
size_t filter_scratch_bytes(int n_rows = 1);
// Draws from the filtered distribution on the device (beam search candidates, t > 0 sampling: whisper_sample_token(best = false)
// and whisper_sample_token_topk, W/whisper.cpp:4777-4909).  The reference draws with std::discrete_distribution on the 51 866
// probabilities: id = first i with cdf(i) >= u, u = generate_canonical(mt19937).  The host keeps the generator and passes the
// uniform numbers u [n_rows][k]; the device filters the logits (same rules as filter_argmax), forms p = exp(l - lse), sums them in
// double (tree order) and finds the k ids per row — k (id, p, plog) and the timestamp statistics come back instead of 207 KB
// of logits per row.  out [n_rows][k] (SampleOut::seq = draw index); scratch: filter_draw_scratch_bytes.
void filter_draw(const float * logits, const uint8_t * static_ban, const DecStep * step, const double * u, int k, SampleOut * out,
                 void * scratch, hipStream_t st, int n_rows, int tid_default);
size_t filter_draw_scratch_bytes(int n_rows);
// first kernel of a replayed step: fetch DecStep from pinned host memory, mirror it on the device, embed the token
void dec_embed_step(const DecStep * host_step, DecStep * dev_step, int S, const __half * te, const float * pe, float * x, hipStream_t st,
                    int n_rows = 1);

// ---------------------------------------------------------------- block-quantised weights (k_quant.hip)
// The reference keeps q4_0 / q4_1 / q5_0 / q5_1 / q8_0 matrices quantised, turns every activation row of a mul_mat into
// q8_0 / q8_1 blocks and takes an integer dot per 32-element block, scaled by d_w * d_a (+ m_w * s_a)
// (SURVEY App. B rule 1; W/ggml-quants.c:837-870, 2442-3560).  Same here: the blocks stay quantised in HBM, the integer
// dots run on v_mfma_i32_32x32x32_i8 (one q block = the K of one instruction, so the dot of a block is exact), the
// per-block f32 scales are applied on the way out of the accumulator.
//
// HBM layout of a quantised matrix [N][K] ("tiles"): a tile = 32 weight rows x 2 blocks (K = 64) = 64 blocks, one per
// lane: lane (n = lane % 32, g = lane / 32) owns block 2 * tp + g of row 32 * tn + n.  A tile stores the 64 quant
// payloads first ([lane][QB bytes]), then the 64 headers ([lane][HB bytes]); tiles of a row group are consecutive
// in tp, row groups follow each other: tile index = tn * (K / 64) + tp.  A wavefront reads a tile with two fully
// coalesced loads; nothing is rewritten, the file's blocks are only permuted (headers of the 2- and 6-byte kinds are
// padded to 4 / 8 bytes).  N is padded to a multiple of 32 with zero blocks.
enum QType : int { QT_NONE = 0, QT_Q4_0 = 2, QT_Q4_1 = 3, QT_Q5_0 = 6, QT_Q5_1 = 7, QT_Q8_0 = 8 };   // ggml_type ids
struct QGeom { int qb, hb, file_bytes; bool has_m; };     // payload / header bytes of a block in the tile layout; *_1 kinds carry m
QGeom  q_geom(int qtype);
inline size_t q_tile_bytes(int qtype) { const QGeom g = q_geom(qtype); return (size_t) 64 * (g.qb + g.hb); }
inline size_t q_matrix_bytes(int qtype, int64_t N, int64_t K) { return (size_t) ((N + 31) / 32) * (size_t) (K / 64) * q_tile_bytes(qtype); }
// host: permute the ggml blocks of rows [0, N) (row stride K / 32 blocks of file_bytes) into the tile layout at dst
void   q_repack_host(int qtype, const uint8_t * src, int64_t N, int64_t K, uint8_t * dst);
struct QMat { const uint8_t * tiles = nullptr; int qtype = QT_NONE; };

// activation rows as q8 blocks in global memory (the GEMM's A operand): qs [M][K] int8; scales block-major d [K / 32][ldm],
// s [K / 32][ldm] (the GEMM fetches the scales of 64 / 128 consecutive rows of one block with one load per wavefront);
// s = d * sum(q) for the q8_1 kinds; for the q8_0 kinds d is rounded to f16 as the reference stores it and s = 0
// deq (optional): the same rows as f16(d * q) [M][K] — the A operand of the f16 form of qgemm; wdeq / wdeq_elems: scratch image for one
// dequantised weight matrix (or a group of cross K | V layers) [rows][K] f16
// wdeq_ready: the image already holds the matrix the next qgemm multiplies with (written by quantize_rows' fused launch)
struct Q8Rows { int8_t * qs; float * d; float * s; int ldm; __half * deq = nullptr; __half * wdeq = nullptr; size_t wdeq_elems = 0; bool wdeq_ready = false;
                const void * wdeq_of = nullptr;        // wdeq_of: the matrix (QMat::tiles) whose f16 image wdeq holds when wdeq_ready — qgemm checks it
                bool w_resident_ok = false; };         // the caller's matrices may be kept as resident f16 images (the encoder's: k_quant.hip qweights_f16_get)
// rows -> q8.  Exactly one source: x32 (+ optional LayerNorm gain/bias: y = LN(x) * g + b in f32, the reference quantises that
// f32 tensor) or x16 (an f16 tensor, e.g. the GELU output, widened exactly).  out32 / out16: optional copy of the LN result.
// W_next / N_next (optional): the [N_next][K] matrix of the projection these rows feed.  When that projection will take the f16 form
// (qgemm), its weight image is written by the same launch; returns true then (the caller sets Q8Rows::wdeq_ready for that qgemm).
bool quantize_rows(const float * x32, const __half * x16, int M, int K, const float * ln_g, const float * ln_b, float eps,
                   int qtype, Q8Rows out, float * out32, __half * out16, hipStream_t st, const QMat * W_next = nullptr, int N_next = 0);

// C[M][N] = A_q8[M][K] . W_q[N][K]^T with the GEMM's epilogues (Epi above; GemmArgs fields A / W / lda / ldw unused).
// N % 128 == 0, K % 64 == 0.
void qgemm(int epi, const GemmArgs & a, Q8Rows A, QMat W, hipStream_t st);
// resident f16 images of the matrices the f16 form multiplies with (k_quant.hip: qweights_f16_get): bytes held by the process, and the
// release of every image whose matrix lies in the arena [lo, hi) (free_weights)
size_t qweights_f16_cached_bytes();
void   qweights_f16_release(const void * lo, const void * hi);
// rows [row0, row0 + rows) of a quantised matrix (row0 % 32 == 0) as f16(d * q + m) [rows][K]
void qdequant(QMat W, int64_t row0, int64_t rows, int K, __half * out, hipStream_t st);

// <= 32 activation rows against a quantised matrix: the weight tiles are streamed once, the rows are quantised in the
// prologue of every workgroup (LayerNorm of x32 if ln_g, plain f32 rows a32, or f16 rows a16) — GemvArgs as for gemv();
// the fused attention prologues (sa_*, comb_*) are not available here.
void qrows(const GemvArgs & a, const float * a32, QMat W, hipStream_t st);
bool qrows_ksplit_ok(const GemvArgs & a, int parts);        // may this launch split K over `parts` workgroups per row group (GemvArgs::ksplit)?
// the cross-attention of a row of a block-quantised model (more rows: WMI_Q_XATTN_ROWS, slower) with the query projection inside (LayerNorm(x32) . W_cq, scaled, f16) — one
// launch instead of qrows(EPI_Q_SCALED) + attn_cross_split_partials, the same bits; partials as attn_cross_split_partials.  pfW / pfN / pfK:
// the next weight-streaming launch's matrix (prefetched).  false: not available for this shape (nothing launched; take the two launches)
bool qattn_cross_qsplit_partials(const float * x32, const float * ln_g, const float * ln_b, float eps, QMat Wcq, const float * bq, float qscale,
                                 int n, int S, int H, const __half * kc, const __half * vc, int T, float * scratch,
                                 const float ** po, const float ** pl, const float ** pm, int * pns, hipStream_t st, int64_t kv_row_stride,
                                 QMat pfW, int pfN, int pfK);

// token embedding gather from a quantised matrix: x[i] = dequant(te[token[i]]) + pe[pos[i]]   (W/ggml.c get_rows, dequantize_row_*)
void qdec_embed(const int32_t * tokens, const int32_t * pos, int n, int S, QMat te, const float * pe, float * x, hipStream_t st);
void qdec_embed_step(const DecStep * host_step, DecStep * dev_step, int S, QMat te, const float * pe, float * x, hipStream_t st, int n_rows = 1);

// misc
void touch(int * p, int blocks, hipStream_t st);     // trivial dependent kernel (launch-floor probe)
void fill_zero(void * p, size_t bytes, hipStream_t st);

}} // namespace wmi::k
