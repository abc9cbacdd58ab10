// log-mel front end on gfx950 (reference W/whisper.cpp:2614-2887, Appendix G).
//
// One workgroup per STFT frame.  The frame transform follows the reference's decomposition
// exactly — radix-2 even/odd splits 400 -> 200 -> 100 -> 50 -> 25 and a direct 25-point DFT at the
// leaves, with the same 400-entry sin/cos table — so that the f32 rounding sequence (and therefore
// log10 of near-silent bins) tracks the CPU path instead of merely approximating it.  The tables
// (Hann window, sin, cos) are computed on the host with the host libm and uploaded once.
// HBM traffic is trivial (1.9 MB PCM in, 1.9 MB mel out per 30 s); the kernel is latency/VALU bound
// and contributes < 1 % of a transcription.

#include "kernels.h"
#include <type_traits>
#include "wave_ops.h"
#include <climits>
#include <cmath>
#include <mutex>

namespace wmi { namespace k {

namespace {

struct MelTables { float hann[400]; float sinv[400]; float cosv[400]; };
__constant__ MelTables c_mel;

std::atomic<uint64_t> g_tables_mask{0};        // one bit per device: a __constant__ symbol has an instance on every GPU of the process

void upload_tables() {
    MelTables t;
    for (int i = 0; i < 400; ++i) {
        t.hann[i] = 0.5 * (1.0 - cosf((2.0 * M_PI * i) / 400));       // W/whisper.cpp:2712-2725 (periodic)
        double theta = (2 * M_PI * i) / 400;                            // W/whisper.cpp:2620-2629
        t.sinv[i] = sinf(theta);
        t.cosv[i] = cosf(theta);
    }
    (void) hipMemcpyToSymbol(HIP_SYMBOL(c_mel), &t, sizeof(t));
}

__device__ inline int enc_ordered(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ inline float dec_ordered(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__device__ __forceinline__ void mel_pad_body(const float * __restrict__ pcm, int n, float * __restrict__ out, int total, int * __restrict__ gmax) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && gmax) *gmax = INT_MIN;     // running maximum of the frame kernel that follows on the stream (was a launch of its own)
    if (i >= total) return;
    float v = 0.0f;
    if (i < 200) {                       // reflect: out[i] = pcm[200 - i]   (W/whisper.cpp:2827)
        int j = 200 - i; if (j > n - 1) j = n - 1; if (j < 0) j = 0;
        v = n > 0 ? pcm[j] : 0.0f;
    } else if (i < 200 + n) {
        v = pcm[i - 200];
    }
    out[i] = v;
}
__global__ void k_mel_pad(const float * __restrict__ pcm, int n, float * __restrict__ out, int total, int * __restrict__ gmax) { mel_pad_body(pcm, n, out, total, gmax); }
// Lock-step chunks: the same kernels with the chunk on grid.y — one launch per kernel for all chunks of a call instead of one per chunk
// (8 chunks: 8 x (3 mel launches + envelope + 2 event operations) on 8 + 8 streams were ~0.3 ms of HOST time in front of the encoder).
// The body of every kernel is the single-chunk kernel's: per chunk the arithmetic and its order are unchanged.
__global__ void k_mel_pad_b(const MelBatch b) { const int c = blockIdx.y; mel_pad_body(b.pcm[c], b.n[c], b.pad[c], b.n[c] + 480400, b.gmax[c]); }

// butterfly stage: for g groups, combine E = src[(g)*half ...], O = src[(g + ngroups)*half ...]
// into dst[g*N + k], dst[g*N + k + half].  Arrays are stored as [leaf][k] complex (re, im interleaved, read and
// written as one 8-byte LDS access).  tw = this stage's twiddles, compact: tw[k] = (cos, sin)(2 pi k / N), k < N/2 —
// consecutive lanes read consecutive 8-byte slots (the shared 400-entry table read at stride 400/N put 64 lanes on
// 64/stride banks).
__device__ inline void butterfly_stage(const float * src, float * dst, int ngroups, int N, int tid, int nthreads,
                                       bool last, const float2 * tw) {
    const int half = N / 2;
    const int total = ngroups * half;
    for (int t = tid; t < total; t += nthreads) {
        const int g = t / half, kk = t % half;
        const float2 * E = (const float2 *) src + (g * half);
        const float2 * O = (const float2 *) src + ((g + ngroups) * half);
        const float2 w = tw[kk];
        const float re =  w.x;
        const float im = -w.y;
        const float2 e = E[kk], o = O[kk];
        float2 * D = (float2 *) dst + (g * N);
        D[kk] = make_float2(fmaf(-im, o.y, fmaf(re, o.x, e.x)), fmaf(im, o.x, fmaf(re, o.y, e.y)));
        if (!last || kk == 0)            // the last stage only needs bins 0..200
            D[kk + half] = make_float2(fmaf(im, o.y, fmaf(-re, o.x, e.x)), fmaf(-im, o.x, fmaf(-re, o.y, e.y)));
    }
}

// frames past the audio: log10(1e-10)  (W/whisper.cpp:2784-2789) — one thread per element, no FFT workgroups
__device__ __forceinline__ void mel_tail_body(float * __restrict__ mel, int n_len, int n_mel, int n_fft_frames, int * __restrict__ gmax) {
    const int w = n_len - n_fft_frames;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < w * n_mel) { const int j = i / w, f = n_fft_frames + (i - j * w); mel[(size_t) j * n_len + f] = -10.0f; }
    if (i == 0 && w > 0) atomicMax(gmax, enc_ordered(-10.0f));
}
__global__ void k_mel_tail(float * __restrict__ mel, int n_len, int n_mel, int n_fft_frames, int * __restrict__ gmax) { mel_tail_body(mel, n_len, n_mel, n_fft_frames, gmax); }

constexpr int MEL_NT = 256;            // threads per frame (4 wavefronts: 1.6 leaf-DFT outputs per thread)
__device__ __forceinline__ void mel_frames_body(const float * __restrict__ pad, int n_valid, int n_fft_frames,
                                                    int n_len, int n_mel, const float * __restrict__ filters,
                                                    const int32_t * __restrict__ ranges, const float * __restrict__ taps,
                                                    float * __restrict__ mel, int * __restrict__ gmax) {
    __shared__ float xin[400];
    __shared__ __attribute__((aligned(8))) float bufA[800];
    __shared__ __attribute__((aligned(8))) float bufB[800];
    __shared__ float pw[204];
    // twiddles in LDS, one compact (cos, sin) table per stage: leaf DFT 25 entries (index (kk m) mod 25 <-> the shared
    // table's 16 (kk m) mod 400), then N = 50, 100, 200, 400 with N/2 entries each.  Same values as the reference's
    // 400-entry table, laid out so that neighbouring lanes hit neighbouring banks.
    __shared__ __attribute__((aligned(8))) float2 tw25[25], tw50[25], tw100[50], tw200[100], tw400[200];
    const int frame = blockIdx.x;
    const int tid = threadIdx.x;
    if (frame >= n_fft_frames) return;

    const int offset = frame * 160;
    int nin = n_valid - offset; if (nin > 400) nin = 400;
    {   // samples, window and twiddles: every load of the thread first, from clamped indices, then the LDS stores (in source
        // order hipcc waited after each table: five dependent round trips at the start of every frame)
        static_assert(MEL_NT == 256, "two samples per thread");
        const int j0 = tid, j1 = tid + 256;                         // j1 < 400 for tid < 144
        const int j1c = j1 < 400 ? j1 : 0, jt = tid < 200 ? tid : 0, j25 = tid < 25 ? tid : 0, j50 = tid < 50 ? tid : 0, j100 = tid < 100 ? tid : 0;
        const float h0 = c_mel.hann[j0], h1 = c_mel.hann[j1c];
        const float p0 = pad[offset + (j0 < nin ? j0 : 0)], p1 = pad[offset + (j1c < nin ? j1c : 0)];
        const float2 w25 = make_float2(c_mel.cosv[16 * j25], c_mel.sinv[16 * j25]), w50 = make_float2(c_mel.cosv[8 * j25], c_mel.sinv[8 * j25]);
        const float2 w100 = make_float2(c_mel.cosv[4 * j50], c_mel.sinv[4 * j50]);
        const float2 w200 = make_float2(c_mel.cosv[2 * j100], c_mel.sinv[2 * j100]);
        const float2 w400 = make_float2(c_mel.cosv[jt], c_mel.sinv[jt]);
        xin[j0] = j0 < nin ? h0 * p0 : 0.0f;
        if (j1 < 400) xin[j1] = j1 < nin ? h1 * p1 : 0.0f;
        if (tid < 25)  { tw25[tid] = w25; tw50[tid] = w50; }
        if (tid < 50)  tw100[tid] = w100;
        if (tid < 100) tw200[tid] = w200;
        if (tid < 200) tw400[tid] = w400;
    }
    // filter rows of this thread's mel bin(s): independent of the transform, requested up front
    int fr0 = 0, fr1 = 0;
    if (tid < n_mel) { fr0 = ranges[2 * tid]; fr1 = ranges[2 * tid + 1]; }
    // ... and the non-zero taps of that bin themselves (FW groups of 4 cover the widest triangle of the 80- and 128-bin
    // banks; wider ranges finish from memory below): in flight during the transform instead of one round trip per group after it
    constexpr int FW = 12;
    float fw[FW][4], f200 = 0.0f;
    {   // from the compact table (model.cpp: 13 aligned float4 per bin): independent of the ranges, 13 loads instead of 49
        const float4 * tp = (const float4 *) taps + (size_t) (tid < n_mel ? tid : 0) * 13;
        float4 tv[FW + 1];
#pragma unroll
        for (int q = 0; q <= FW; ++q) tv[q] = tp[q];
#pragma unroll
        for (int q = 0; q < FW; ++q) { fw[q][0] = tv[q].x; fw[q][1] = tv[q].y; fw[q][2] = tv[q].z; fw[q][3] = tv[q].w; }
        f200 = tv[FW].x;
    }
    __syncthreads();

    // 16 leaf DFTs of 25 points: leaf r holds x[r + 16 m]  (W/whisper.cpp:2634-2654)
    // leaves are stored in the order the combine stages want: slot(r) with bit-reversed 4-bit r
    for (int t = tid; t < 400; t += MEL_NT) {
        const int r = t / 25, kk = t % 25;
        float re = 0.0f, im = 0.0f;
        int idx = 0;                                 // (kk * m) % 25, advanced incrementally
#pragma unroll 5
        for (int m = 0; m < 25; ++m) {
            const float v = xin[r + 16 * m];
            const float2 w = tw25[idx];
            re = fmaf(v, w.x, re);
            im = fmaf(-v, w.y, im);
            idx += kk; if (idx >= 25) idx -= 25;
        }
        bufA[2 * (r * 25 + kk)]     = re;
        bufA[2 * (r * 25 + kk) + 1] = im;
    }
    __syncthreads();
    // N=50: pairs (r, r+8) -> 8 arrays of 50 indexed by r<8 ; N=100: (r, r+4) ; N=200: (r, r+2) ; N=400: (0,1)
    butterfly_stage(bufA, bufB, 8, 50, tid, MEL_NT, false, tw50);   __syncthreads();
    butterfly_stage(bufB, bufA, 4, 100, tid, MEL_NT, false, tw100); __syncthreads();
    butterfly_stage(bufA, bufB, 2, 200, tid, MEL_NT, false, tw200); __syncthreads();
    butterfly_stage(bufB, bufA, 1, 400, tid, MEL_NT, true, tw400);  __syncthreads();

    for (int j = tid; j < 201; j += MEL_NT) {
        const float re = bufA[2 * j], im = bufA[2 * j + 1];
        pw[j] = fmaf(re, re, im * im);
    }
    __syncthreads();

    float vmax = -INFINITY;
    for (int j = tid; j < n_mel; j += MEL_NT) {
        if (j != tid) { fr0 = ranges[2 * j]; fr1 = ranges[2 * j + 1]; }
        const float * f = filters + (size_t) j * 201;
        double sum = 0.0;
        // W/whisper.cpp:2759-2768 sums 50 groups of 4 taps + the last tap; groups outside [fr0, fr1) are all-zero
        // weights and contribute exactly +0.0 (model.cpp: mel_ranges), so they are skipped
        if (j == tid) {
#pragma unroll
            for (int q = 0; q < FW; ++q) {
                if (fr0 + q < fr1) {
                    const int kk = 4 * (fr0 + q);
                    float g = pw[kk] * fw[q][0];
                    g = fmaf(pw[kk + 1], fw[q][1], g);
                    g = fmaf(pw[kk + 2], fw[q][2], g);
                    g = fmaf(pw[kk + 3], fw[q][3], g);
                    sum += (double) g;
                }
            }
        }
        for (int g4 = (j == tid ? fr0 + FW : fr0); g4 < fr1; ++g4) {
            const int kk = 4 * g4;
            float g = pw[kk] * f[kk];
            g = fmaf(pw[kk + 1], f[kk + 1], g);
            g = fmaf(pw[kk + 2], f[kk + 2], g);
            g = fmaf(pw[kk + 3], f[kk + 3], g);
            sum += (double) g;
        }
        sum += (double) (pw[200] * (j == tid ? f200 : f[200]));
        sum = log10(sum > 1e-10 ? sum : 1e-10);
        const float v = (float) sum;
        mel[(size_t) j * n_len + frame] = v;
        vmax = fmaxf(vmax, v);
    }
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, WMI_SHX(vmax, o));
    // one atomic per frame at most, and only when it can raise the maximum: 12 000 atomics on one address (one per wavefront)
    // serialised at the memory side and were most of this kernel's 76 us
    __shared__ float s_vmax[MEL_NT / 64];
    if ((tid & 63) == 0) s_vmax[tid >> 6] = vmax;
    __syncthreads();
    if (tid == 0) {
        float v = s_vmax[0];
        for (int w = 1; w < MEL_NT / 64; ++w) v = fmaxf(v, s_vmax[w]);
        if (v > -INFINITY) {
            const int e = enc_ordered(v);
            if (e > __hip_atomic_load(gmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(gmax, e);
        }
    }
}
__global__ __launch_bounds__(MEL_NT) void k_mel_frames(const float * __restrict__ pad, int n_valid, int n_fft_frames,
                                                    int n_len, int n_mel, const float * __restrict__ filters,
                                                    const int32_t * __restrict__ ranges, const float * __restrict__ taps,
                                                    float * __restrict__ mel, int * __restrict__ gmax) {
    mel_frames_body(pad, n_valid, n_fft_frames, n_len, n_mel, filters, ranges, taps, mel, gmax);
}
// (Appendix G's sizes of a chunk of n samples: padded image n + 480 400, n_len = (n + 480 000) / 160 frames of which the first
//  min((n + 200) / 160 + 1, n_len) hold audio)
__device__ __forceinline__ int mel_n_len(int n) { return (int) (((long long) n + 480000) / 160); }
__device__ __forceinline__ int mel_n_fft(int n) { const int a = (n + 200) / 160 + 1, l = mel_n_len(n); return a < l ? a : l; }
__global__ __launch_bounds__(MEL_NT) void k_mel_frames_b(const MelBatch b, int n_mel, const float * __restrict__ filters,
                                                      const int32_t * __restrict__ ranges, const float * __restrict__ taps) {
    const int c = blockIdx.y, n = b.n[c];
    mel_frames_body(b.pad[c], n + 200, mel_n_fft(n), mel_n_len(n), n_mel, filters, ranges, taps, b.mel[c], b.gmax[c]);
}
__global__ void k_mel_tail_b(const MelBatch b, int n_mel) {
    const int c = blockIdx.y, n = b.n[c];
    mel_tail_body(b.mel[c], mel_n_len(n), n_mel, mel_n_fft(n), b.gmax[c]);
}

__device__ __forceinline__ void mel_normalize_body(float * __restrict__ mel, int n, const int * __restrict__ gmax) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mmax = (double) dec_ordered(*gmax) - 8.0;       // W/whisper.cpp:2855-2871
    float v = mel[i];
    if ((double) v < mmax) v = (float) mmax;
    mel[i] = (float) (((double) v + 4.0) / 4.0);
}
__global__ void k_mel_normalize(float * __restrict__ mel, int n, const int * __restrict__ gmax) { mel_normalize_body(mel, n, gmax); }
__global__ void k_mel_normalize_b(const MelBatch b, int n_mel) { const int c = blockIdx.y; mel_normalize_body(b.mel[c], n_mel * mel_n_len(b.n[c]), b.gmax[c]); }

__global__ void k_mel_slice(const float * __restrict__ mel, int n_len, int n_mel, int offset, int n_frames,
                            __half * __restrict__ out, int ld, int rows_total) {
    // out row r <-> frame offset + r - 1 ; transposes [n_mel][n_len] -> [frame][n_mel] through LDS
    __shared__ float tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        const int fr = offset + r - 1;
        float v = 0.0f;
        if (c < n_mel && r >= 1 && r <= n_frames && fr < n_len && fr >= 0) v = mel[(size_t) c * n_len + fr];
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        if (r < rows_total && c < ld) out[(size_t) r * ld + c] = f2h(c < n_mel ? tile[tx][j] : 0.0f);
    }
}

// the same for up to 16 lock-step chunks in one launch (grid.z = chunk): chunk c's rows are c * rows_total .. (c + 1) * rows_total - 1 of `out`
__global__ void k_mel_slice_batch(const MelSliceBatch mb, int n_mel, int n_frames, __half * __restrict__ out, int ld, int rows_total) {
    __shared__ float tile[32][33];
    const int ch = blockIdx.z;
    const float * __restrict__ mel = mb.mel[ch]; const int n_len = mb.n_len[ch], offset = mb.offset[ch];
    out += (size_t) ch * rows_total * ld;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        const int fr = offset + r - 1;
        float v = 0.0f;
        if (c < n_mel && r >= 1 && r <= n_frames && fr < n_len && fr >= 0) v = mel[(size_t) c * n_len + fr];
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        if (r < rows_total && c < ld) out[(size_t) r * ld + c] = f2h(c < n_mel ? tile[tx][j] : 0.0f);
    }
}
// `count` runs of `words` 32-bit zeros, `stride_words` apart (the guard rows between the stacked chunks of the batched conv front-end)
__global__ void k_fill_zero_strided(uint32_t * p, size_t words, size_t stride_words, int count) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words && (int) blockIdx.y < count) p[(size_t) blockIdx.y * stride_words + i] = 0u;
}

// mean |x| over a (2 hw + 1)-sample window, summed left to right with the reference's exact arithmetic
// (float accumulator, each step rounded through a double add: `sum += fabs(x)`, W/whisper.cpp:6352-6366),
// so the host-side timestamp heuristics see bit-identical thresholds.
// Also written: the minimum and maximum of every 256-sample block of the envelope (bmin / bmax) — the token-timestamp
// heuristics walk outwards from a token "while the envelope stays above / below a threshold" (W/whisper.cpp:6500-6590),
// on stationary audio across the whole signal for every token; with the block extrema the host skips 256 samples per
// comparison and stops on exactly the same sample.
__device__ __forceinline__ void signal_energy_body(const float * __restrict__ x, int n, int hw, float * __restrict__ out,
                                                       float * __restrict__ bmin, float * __restrict__ bmax) {
    __shared__ float s_min[4], s_max[4];
    // block-stride loop: the grid may be thinner than one workgroup per 256 samples (WMI_ENVELOPE_GRID, A/B: the kernel is bound by its
    // stores into pinned host memory and its stalled waves hold slots beside whatever runs next to it — but 64 workgroups starve the
    // 65-deep f64 chains: the lock-step call 6.0 -> 6.5 ms; the one-chunk call within noise)
    for (int blk = blockIdx.x; blk * 256 < n; blk += gridDim.x) {
    if (blk != (int) blockIdx.x) __syncthreads();          // s_min / s_max are reused
    const int i = blk * blockDim.x + threadIdx.x;
    float v = 0.0f;
    // The reference adds in double and stores the sum as float every step: (float) ((double) sum + fabs((double) x)).  That IS the f32 addition
    // sum + |x| — the double sum of two floats rounded to float equals the correctly rounded float sum (53 >= 2 x 24 + 2 bits: the first
    // rounding cannot move the second) — so the 65-step chain runs on f32 adds.  The 256 + 2 hw samples of a block come through LDS once
    // (each is read by up to 65 threads).
    if (hw <= 64) {
        __shared__ float tile[256 + 128];
        const int base = blk * 256 - hw;
        for (int t = threadIdx.x; t < 256 + 2 * hw; t += 256) { const int k = base + t; tile[t] = (k >= 0 && k < n) ? fabsf(x[k]) : 0.0f; }
        __syncthreads();
        if (i < n) {
            float sum = 0.0f;
            // samples outside [0, n) are skipped by the reference: adding their 0.0f placeholders is the same sum (x + 0.0f == x for x >= 0)
            for (int j = 0; j <= 2 * hw; ++j) sum = sum + tile[threadIdx.x + j];
            v = sum / (float) (2 * hw + 1);
            out[i] = v;
        }
        __syncthreads();                                    // (tile is refilled by the next block of the stride loop)
    } else
    if (i < n) {
        float sum = 0.0f;
        for (int j = -hw; j <= hw; ++j) {
            const int k = i + j;
            if (k >= 0 && k < n) sum = (float) ((double) sum + fabs((double) x[k]));
        }
        v = sum / (float) (2 * hw + 1);
        out[i] = v;
    }
    float lo = i < n ? v : INFINITY, hi = i < n ? v : -INFINITY;
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, WMI_SHX(lo, o)); hi = fmaxf(hi, WMI_SHX(hi, o)); }
    if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = lo; s_max[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bmin[blk] = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
        bmax[blk] = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    }
    }
}
__global__ __launch_bounds__(256) void k_signal_energy(const float * __restrict__ x, int n, int hw, float * __restrict__ out,
                                                       float * __restrict__ bmin, float * __restrict__ bmax) { signal_energy_body(x, n, hw, out, bmin, bmax); }
__global__ __launch_bounds__(256) void k_signal_energy_b(const MelBatch b, int hw) {
    const int c = blockIdx.y;
    signal_energy_body(b.pcm[c], b.n[c], hw, b.energy[c], b.bmin[c], b.bmax[c]);
}

// ---------------------------------------------------------------- host-adjacent DSP
// stereo frames -> mono, the streaming node's down-mix in front of the resampler (src/speech_to_text.cpp:45-51):
// out[i] = (float) ((x + y) / 2.0) — a float add, then an exact halving
__global__ void k_downmix(const float2 * __restrict__ frames, int n, float * __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float2 f = frames[i]; out[i] = (float) ((double) (f.x + f.y) / 2.0); }
}

// SpeechToText::voice_activity_detection on a window of n samples (the caller passes the last 3 s): the reference's
// _high_pass_filter + _vad_simple (src/speech_to_text.cpp:53-104), operation for operation.  The filter runs IN PLACE, so its
// "previous input" data[i - 1] is the previous OUTPUT: y_i = alpha * ((y_{i-1} + x_i) - y_{i-1}) — a recurrence through f32
// rounding only, but a recurrence: it and the two running f32 energy sums are evaluated by ONE lane in sample order (48 000
// samples x ~45 cycles of dependent single-lane VALU issue = 0.9 ms per call against a 300 ms cadence); the other lanes stage the samples through LDS.
// res: {decision (1.0 = "no activity"), energy_all, energy_last}
__global__ __launch_bounds__(256) void k_vad(const float * __restrict__ x, int n, int n_last, float alpha, int filter, float vad_thold,
                                             float * __restrict__ res) {
    constexpr int CHUNK = 8192;
    __shared__ __attribute__((aligned(16))) float buf[CHUNK];
    float y = 0.0f, e_all = 0.0f, e_last = 0.0f;
    const int first_last = n - n_last;                      // samples from here on also count towards energy_last
    for (int c0 = 0; c0 < n; c0 += CHUNK) {
        const int m = min(CHUNK, n - c0);
        for (int i = threadIdx.x; i < m; i += 256) buf[i] = x[c0 + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            // eight samples per trip come out of LDS with two 16-byte reads BEFORE the dependent chain touches them (one ds_read_b32 per
            // sample inside the chain was a ~100-cycle LDS round trip per sample: 1.8 ms per 3 s window); the special cases (very first
            // sample, start of the energy_last tail) split the range instead of sitting in the loop body.  The arithmetic and its order
            // are unchanged: y = alpha * ((y + v) - y), e_all += |y|, e_last += |y| over the tail.
            auto run = [&](int i0, int i1, auto tail_tag) {
                constexpr bool TAIL = decltype(tail_tag)::value;
                int i = i0;
                for (; i < i1 && (i & 3); ++i) {                   // up to the next 16-byte boundary
                    float v = buf[i];
                    if (filter) { y = __fmul_rn(alpha, __fsub_rn(__fadd_rn(y, v), y)); v = y; }
                    e_all = __fadd_rn(e_all, fabsf(v));
                    if (TAIL) e_last = __fadd_rn(e_last, fabsf(v));
                }
                for (; i + 8 <= i1; i += 8) {
                    const float4 a4 = *(const float4 *) (buf + i), b4 = *(const float4 *) (buf + i + 4);
                    const float v8[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        float v = v8[u];
                        if (filter) { y = __fmul_rn(alpha, __fsub_rn(__fadd_rn(y, v), y)); v = y; }
                        e_all = __fadd_rn(e_all, fabsf(v));
                        if (TAIL) e_last = __fadd_rn(e_last, fabsf(v));
                    }
                }
                for (; i < i1; ++i) {
                    float v = buf[i];
                    if (filter) { y = __fmul_rn(alpha, __fsub_rn(__fadd_rn(y, v), y)); v = y; }
                    e_all = __fadd_rn(e_all, fabsf(v));
                    if (TAIL) e_last = __fadd_rn(e_last, fabsf(v));
                }
            };
            int i0 = 0;
            if (c0 == 0) {                                         // the filter starts from the first sample itself (y = data[0])
                const float v = buf[0];
                if (filter) y = v;
                e_all = __fadd_rn(e_all, fabsf(v));
                if (first_last <= 0) e_last = __fadd_rn(e_last, fabsf(v));
                i0 = 1;
            }
            const int split = min(max(first_last - c0, i0), m);    // [i0, split): body, [split, m): also energy_last
            run(i0, split, std::false_type{});
            run(split, m, std::true_type{});
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        e_all /= (float) n;
        if (n_last != 0) e_last /= (float) n_last;
        const bool quiet = !(!(e_all < 0.0001f && e_last < 0.0001f) || e_last > vad_thold * e_all);
        res[0] = quiet ? 1.0f : 0.0f; res[1] = e_all; res[2] = e_last;
    }
}

__global__ void k_fill_zero(uint32_t * p, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 0u;
}

__global__ void k_set_int(int * p, int v) { *p = v; }

} // namespace

void mel_pad(const float * pcm, int n_samples, float * pcm_pad, int n_pad_total, hipStream_t st, int * gmax_reset) {
    hipLaunchKernelGGL(k_mel_pad, dim3((n_pad_total + 255) / 256), dim3(256), 0, st, pcm, n_samples, pcm_pad, n_pad_total, gmax_reset);
}

void mel_frames(const float * pcm_pad, int n_valid, int n_fft_frames, int n_len, int n_mel, const float * filters,
                const int32_t * ranges, const float * taps, float * mel, int * gmax, hipStream_t st) {
    {
        int dev = 0; (void) hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        if (!(g_tables_mask.load(std::memory_order_acquire) & bit)) { upload_tables(); g_tables_mask.fetch_or(bit, std::memory_order_release); }
    }
    if (n_fft_frames > 0)
        hipLaunchKernelGGL(k_mel_frames, dim3(n_fft_frames), dim3(MEL_NT), 0, st, pcm_pad, n_valid, n_fft_frames, n_len, n_mel,
                           filters, ranges, taps, mel, gmax);
    const int tail = (n_len - n_fft_frames) * n_mel;
    if (tail > 0) hipLaunchKernelGGL(k_mel_tail, dim3((tail + 255) / 256), dim3(256), 0, st, mel, n_len, n_mel, n_fft_frames, gmax);
}

// lock-step chunks: pad + frames (+ tail) + normalise of nb chunks, one launch each (grid.y = chunk; grid.x = the largest chunk's extent,
// the others' surplus workgroups return at once)
void mel_batch(const MelBatch & b, int nb, int n_mel, const float * filters, const int32_t * ranges, const float * taps, hipStream_t st) {
    {
        int dev = 0; (void) hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        if (!(g_tables_mask.load(std::memory_order_acquire) & bit)) { upload_tables(); g_tables_mask.fetch_or(bit, std::memory_order_release); }
    }
    int n_max = 0, n_min = INT_MAX;
    for (int c = 0; c < nb; ++c) { n_max = std::max(n_max, b.n[c]); n_min = std::min(n_min, b.n[c]); }
    auto n_len = [](int n) { return (int) (((long long) n + 480000) / 160); };
    auto n_fft = [&](int n) { return std::min((n + 200) / 160 + 1, n_len(n)); };
    hipLaunchKernelGGL(k_mel_pad_b, dim3((n_max + 480400 + 255) / 256, nb), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_mel_frames_b, dim3(n_fft(n_max), nb), dim3(MEL_NT), 0, st, b, n_mel, filters, ranges, taps);
    int tail_max = 0;
    for (int c = 0; c < nb; ++c) tail_max = std::max(tail_max, (n_len(b.n[c]) - n_fft(b.n[c])) * n_mel);
    if (tail_max > 0) hipLaunchKernelGGL(k_mel_tail_b, dim3((tail_max + 255) / 256, nb), dim3(256), 0, st, b, n_mel);
    hipLaunchKernelGGL(k_mel_normalize_b, dim3((n_mel * n_len(n_max) + 255) / 256, nb), dim3(256), 0, st, b, n_mel);
}
// ... and their |x| envelopes (b.energy / bmin / bmax)
void signal_energy_batch(const MelBatch & b, int nb, int hw, hipStream_t st) {
    int n_max = 0;
    for (int c = 0; c < nb; ++c) n_max = std::max(n_max, b.n[c]);
    if (n_max > 0) hipLaunchKernelGGL(k_signal_energy_b, dim3((n_max + 255) / 256, nb), dim3(256), 0, st, b, hw);
}

void mel_normalize(float * mel, int n, const int * gmax, hipStream_t st) {
    hipLaunchKernelGGL(k_mel_normalize, dim3((n + 255) / 256), dim3(256), 0, st, mel, n, gmax);
}

void mel_slice(const float * mel, int n_len, int n_mel, int offset, int n_frames, __half * out, int ld, int rows_total,
               hipStream_t st) {
    dim3 grid((rows_total + 31) / 32, (ld + 31) / 32);
    hipLaunchKernelGGL(k_mel_slice, grid, dim3(256), 0, st, mel, n_len, n_mel, offset, n_frames, out, ld, rows_total);
}

void mel_slice_batch(const MelSliceBatch & mb, int nb, int n_mel, int n_frames, __half * out, int ld, int rows_total, hipStream_t st) {
    dim3 grid((rows_total + 31) / 32, (ld + 31) / 32, nb);
    hipLaunchKernelGGL(k_mel_slice_batch, grid, dim3(256), 0, st, mb, n_mel, n_frames, out, ld, rows_total);
}
void fill_zero_strided(void * p, size_t bytes, size_t stride_bytes, int count, hipStream_t st) {
    if (count <= 0 || bytes == 0) return;
    hipLaunchKernelGGL(k_fill_zero_strided, dim3((unsigned) ((bytes / 4 + 255) / 256), count), dim3(256), 0, st, (uint32_t *) p, bytes / 4, stride_bytes / 4, count);
}

void signal_energy(const float * pcm, int n, int hw, float * out, float * bmin, float * bmax, hipStream_t st) {
    static const int thin = getenv("WMI_ENVELOPE_GRID") ? atoi(getenv("WMI_ENVELOPE_GRID")) : 0;      // A/B knob; 0 = one workgroup per block
    const int nblk = (n + 255) / 256;
    hipLaunchKernelGGL(k_signal_energy, dim3(thin > 0 && thin < nblk ? thin : nblk), dim3(256), 0, st, pcm, n, hw, out, bmin, bmax);
}

// ---------------------------------------------------------------- token timestamps: window sums + walks over the envelope (kernels.h TsTok / TsOut)
// walk: the first position p in k, k + dir, k + 2 dir, ... with p == bound or NOT cont(en[p]), cont(x) = above ? x > th : x < th — the
// reference's `while (cont(en[k]) && k != bound) k += dir` — 64 samples per trip, whole 256-sample blocks skipped on the block extrema
// (all above th <=> block minimum > th, all below <=> block maximum < th) as the host walks do.  A NaN threshold stops at once, as there.
__device__ __forceinline__ int ts_walk(const float * __restrict__ en, const float * __restrict__ bext, int k, int bound, float th, int dir, bool above, int lane) {
    for (;;) {
        const int dist = dir > 0 ? bound - k : k - bound;
        if (dist <= 0) return k;
        const bool edge = dir > 0 ? (k & 255) == 0 : (k & 255) == 255;
        if (edge) {
            const int b = (k >> 8) + dir * lane;
            const bool inside = dir > 0 ? ((b + 1) << 8) <= bound : (b >= 0 && (b << 8) > bound);
            bool cont = false;
            if (inside) { const float e = bext[b]; cont = above ? e > th : e < th; }
            const unsigned long long m = __ballot(cont);
            const int nskip = m == ~0ull ? 64 : __builtin_ctzll(~m);
            if (nskip > 0) { k += dir * 256 * nskip; continue; }
        }
        const int to_edge = dir > 0 ? 256 - (k & 255) : (k & 255) + 1;
        int cnt = dist + 1; if (cnt > 64) cnt = 64; if (cnt > to_edge) cnt = to_edge;
        const int p = k + dir * lane;
        bool stop = false;
        if (lane < cnt) { const float x = en[p]; stop = p == bound || !(above ? x > th : x < th); }
        const unsigned long long m = __ballot(stop);
        if (m) return k + dir * __builtin_ctzll(m);
        k += dir * cnt;
    }
}

// One trip of <= 1 024 samples starting at p (n_left of them exist), added to acc by ONE wavefront: per 256-element block the integer form where
// it is exact (see k_ts_refine), else one f32 addition per element in index order.  x[r] = element 64 r + lane of the trip (0 past the end).
__device__ __forceinline__ float ts_trip_exact(float acc, const float (&x)[16], int n_left, int lane) {
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
        const int b0 = sub * 256;
        if (b0 >= n_left) break;
        const uint32_t bits = __float_as_uint(acc);
        const int E = (int) ((bits >> 23) & 0xFF) - 127;
        bool done = false;
        if (b0 + 256 <= n_left && (bits >> 31) == 0 && E >= -100 && E <= 100) {
            const uint32_t units = (bits & 0x7FFFFFu) | 0x800000u;
            const float scale = __uint_as_float((uint32_t) (127 + 23 - E) << 23);
            bool bad = false; uint32_t isum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float tt = x[sub * 4 + k] * scale;
                const int q = __float2int_rn(tt);
                const float d = fabsf(tt - (float) q);
                bad = bad || d == 0.5f || !(tt < 8388608.0f) || tt < 0.0f;
                isum += (uint32_t) q;
            }
            if (__ballot(bad) == 0) {
                _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) isum += (uint32_t) __shfl_xor((int) isum, o);
                const unsigned long long total = (unsigned long long) units + isum;
                if (total < 0x1000000ull) { acc = __uint_as_float(((uint32_t) (E + 127) << 23) | ((uint32_t) total & 0x7FFFFFu)); done = true; }
            }
        }
        if (!done) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c0 = b0 + k * 64;
                const int cnt = n_left - c0 < 64 ? n_left - c0 : 64;
                const float v = x[sub * 4 + k];
                if (cnt >= 64) {
#pragma unroll
                    for (int l = 0; l < 64; ++l) acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
                } else {
                    for (int l = 0; l < cnt; ++l) acc = acc + __shfl(v, l);
                }
            }
        }
    }
    return acc;
}

// One workgroup of TS_W wavefronts per token.
//  * window sum = the loop `s = 0; for (i) s += en[i]` in f32, left to right.  As on the host (full.cpp seq_sum_f32, pinned by
//    tests/test_abi.py::test_sequential_sum_is_exact): while the accumulator stays in one binade [2^E, 2^(E+1)) it is a multiple of
//    ulp = 2^(E-23) and adding x adds rne(x / ulp) units — an integer sum, ORDER-FREE — unless x / ulp is a tie (.5 exactly), negative, NaN
//    or >= 2^23 units, or the total leaves the binade.  So the wavefronts take the next TS_W trips of 1 024 samples side by side, all with
//    the exponent the sum has NOW; the trips are then accepted in order while the running total stays below 2^24 units and no trip saw an
//    exception; the first trip that is not accepted (a binade crossing — ~20 per window, the envelope is non-negative —, a tie, the tail) is
//    added by wavefront 0 the careful way (ts_trip_exact), and the rest start again from there with the new exponent.
//  * threshold and the four walks: wavefront 0.
constexpr int TS_W = 8;
__global__ __launch_bounds__(TS_W * 64) void k_ts_refine(const float * __restrict__ en_, const float * __restrict__ bmin_, const float * __restrict__ bmax_,
                                                         int n_samples_, const TsTok * __restrict__ in, TsOut * __restrict__ out) {
    __shared__ uint32_t s_sum[TS_W]; __shared__ int s_bad[TS_W]; __shared__ float s_acc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TsTok t = in[blockIdx.x];
    // (tokens of several chunks in one launch carry their own envelope)
    const float * __restrict__ en = t.en ? t.en : en_;
    const int n_samples = t.en ? t.n_samples : n_samples_;
    const float * __restrict__ bmin = t.en ? t.en + t.ext_off : bmin_;
    const float * __restrict__ bmax = t.en ? bmin + ((size_t) t.n_samples / 256 + 2) : bmax_;
    float acc = 0.0f;
    {
        const float * p = en + t.a0;
        const int n = t.a1 - t.a0;
        auto fetch = [&](float (&dst)[16], int base) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const int e = base + r * 64 + lane; dst[r] = (e >= 0 && e < n) ? p[e] : 0.0f; }
        };
        float x[16], nx[16]; int nx_base = -1;
        int i = 0;
        while (i < n) {
            const int base = i + wave * 1024;
            if (nx_base == base) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = nx[r];
            } else fetch(x, base);
            nx_base = base + TS_W * 1024;
            if (nx_base < n) fetch(nx, nx_base); else nx_base = -1;          // the next round's trip of this wavefront, if this round is accepted whole
            const uint32_t bits = __float_as_uint(acc);
            const int E = (int) ((bits >> 23) & 0xFF) - 127;
            const bool fast = (bits >> 31) == 0 && E >= -100 && E <= 100;
            bool bad = !fast || base + 1024 > n;
            uint32_t isum = 0;
            if (!bad) {
                const float scale = __uint_as_float((uint32_t) (127 + 23 - E) << 23);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float tt = x[r] * scale;
                    const int q = __float2int_rn(tt);
                    const float d = fabsf(tt - (float) q);
                    bad = bad || d == 0.5f || !(tt < 8388608.0f) || tt < 0.0f;
                    isum += (uint32_t) q;                                    // <= 16 x 2^23
                }
                bad = bad || isum >= 0x1000000u;                             // (then the wave total below cannot wrap: 64 x < 2^24)
                bad = __ballot(bad) != 0;
                if (!bad) { _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) isum += (uint32_t) __shfl_xor((int) isum, o); }
            }
            if (lane == 0) { s_sum[wave] = isum; s_bad[wave] = bad ? 1 : 0; }
            __syncthreads();
            int k = 0;
            if (fast) {
                uint32_t running = (bits & 0x7FFFFFu) | 0x800000u;
                for (; k < TS_W; ++k) {
                    if (s_bad[k] || (unsigned long long) running + s_sum[k] >= 0x1000000ull) break;
                    running += s_sum[k];
                }
                if (k > 0) acc = __uint_as_float(((uint32_t) (E + 127) << 23) | (running & 0x7FFFFFu));
            }
            i += k * 1024;
            const bool slow = k < TS_W && i < n;                              // uniform over the workgroup
            if (slow) {
                if (wave == 0) {
                    float y[16];
                    if (k == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) y[r] = x[r];                // wavefront 0's own trip is the one that failed
                    } else fetch(y, i);
                    const float a2 = ts_trip_exact(acc, y, n - i, lane);
                    if (lane == 0) s_acc = a2;
                }
                __syncthreads();
                acc = s_acc;
                i += n - i < 1024 ? n - i : 1024;
            }
            __syncthreads();                                                  // s_sum / s_bad / s_acc are rewritten next round
        }
    }
    if (wave != 0) return;
    const int ns = t.a1 - t.a0;
    const float th = (float) (0.5 * (double) acc / (double) ns);            // `const float thold = 0.5 * sum / ns;`
    const float x0 = en[t.s0], x1 = en[t.s1];
    const int wa = ts_walk(en, bmin, t.s0, 0, th, -1, true, lane);
    const int wb = ts_walk(en, bmax, t.s0, t.s1, th, +1, false, lane);
    const int wc = ts_walk(en, bmin, t.s1, n_samples - 1, th, +1, true, lane);
    const int wd = ts_walk(en, bmax, t.s1, 0, th, -1, false, lane);
    if (lane == 0) {
        TsOut o; o.sum = acc; o.thold = th; o.e0 = x0 > th ? 1 : 0; o.e1 = x1 > th ? 1 : 0;
        o.w_down_above_s0 = wa; o.w_up_below_s0 = wb; o.w_up_above_s1 = wc; o.w_down_below_s1 = wd;
        out[blockIdx.x] = o;
    }
}
void ts_refine(const float * en, const float * bmin, const float * bmax, int n_samples, const TsTok * in, TsOut * out, int n_tok, hipStream_t st) {
    if (n_tok > 0) hipLaunchKernelGGL(k_ts_refine, dim3(n_tok), dim3(TS_W * 64), 0, st, en, bmin, bmax, n_samples, in, out);
}

// A copy by a handful of workgroups: 16 bytes per lane and trip, four trips in flight.  Used for the |x| envelopes of a lock-step call
// (15 MB to pinned host memory for 8 chunks): as stores of the full-grid envelope kernel — or of the runtime's blit kernel — thousands
// of wavefronts queue megabytes of PCIe writes at once, and the 32-byte results of the decode steps running beside them wait behind
// that queue; a few wavefronts keep it a few KB deep and take as long as the decode phase lets them.
__global__ __launch_bounds__(256) void k_copy_thin(const uint4 * __restrict__ src, uint4 * __restrict__ dst, size_t n16, const unsigned char * __restrict__ tsrc,
                                                   unsigned char * __restrict__ tdst, int tail) {
    const size_t stride = (size_t) gridDim.x * 256;
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && (int) threadIdx.x < tail) tdst[threadIdx.x] = tsrc[threadIdx.x];
}
void copy_thin(const void * src, void * dst, size_t bytes, int wgs, hipStream_t st) {
    if (!bytes) return;
    const size_t n16 = bytes / 16; const int tail = (int) (bytes - n16 * 16);
    hipLaunchKernelGGL(k_copy_thin, dim3(wgs < 1 ? 1 : wgs), dim3(256), 0, st, (const uint4 *) src, (uint4 *) dst, n16,
                       (const unsigned char *) src + n16 * 16, (unsigned char *) dst + n16 * 16, tail);
}

void downmix_stereo(const float * frames, int n_frames, float * out, hipStream_t st) {
    if (n_frames > 0) hipLaunchKernelGGL(k_downmix, dim3((n_frames + 255) / 256), dim3(256), 0, st, (const float2 *) frames, n_frames, out);
}
void vad_window(const float * x, int n, int n_last, float alpha, bool filter, float vad_thold, float * res, hipStream_t st) {
    hipLaunchKernelGGL(k_vad, dim3(1), dim3(256), 0, st, x, n, n_last, alpha, filter ? 1 : 0, vad_thold, res);
}

__global__ void k_touch(int * p, int nblk) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
void touch(int * p, int blocks, hipStream_t st) { hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, st, p, blocks); }

static thread_local Stamp g_stamp{nullptr, 0};       // per host thread: see kernels.h
void  stamp_enable(unsigned long long * base) { g_stamp.base = base; g_stamp.slot = 0; }
Stamp stamp_next() { if (!g_stamp.base) return Stamp{nullptr, 0}; return Stamp{g_stamp.base, g_stamp.slot++}; }
int   stamp_count() { return g_stamp.slot; }

void fill_zero(void * p, size_t bytes, hipStream_t st) {
    (void) hipMemsetAsync(p, 0, bytes, st);
}

}} // namespace wmi::k
