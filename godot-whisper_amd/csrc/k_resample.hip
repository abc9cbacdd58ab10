// SINC resampler of the streaming node on the device.
//
// replaces: _resample_audio_buffer (src/speech_to_text.cpp:16-43) -> src_simple(SRC_SINC_FASTEST | SRC_SINC_MEDIUM_QUALITY,
// 1 channel) of libsamplerate (thirdparty/libsamplerate/src/samplerate.c:469-483, src_sinc.c:283-427, 1166-1239).
//
// libsamplerate walks the output sequentially through a ring buffer; at a fixed ratio that machinery reduces to
//   out[n] = float( (float_inc / index_inc) * (left_n + right_n) )
// over the zero-extended input, where output n sits at input position pos_n + frac_n (the recurrence
// x += 1/ratio; frac = x - floor(x) in double), left/right are the two half-filter sums in DOUBLE, taps visited from the
// far end towards the centre, each coefficient linearly interpolated between table entries at a 12-bit fixed-point filter
// index.  One thread computes one output with exactly that operation order (separate f64 multiply and add: the build runs
// under -ffp-contract=off), so the result equals the sequential CPU code bit for bit.
//
// The position recurrence itself is exact — no rounding ever happens — when every partial sum x + 1/ratio stays
// representable on the grid of 1/ratio's last mantissa bit (inc = m * 2^q, inc + 1 <= 2^(q+53): 48 k, 44.1 k, 32 k, 96 k,
// 192 k -> 16 k all qualify).  Then pos_n / frac_n are the integer and fractional part of n * m * 2^q, evaluated per thread
// with a 64 x 64 -> 128-bit product.  Otherwise (e.g. 22.05 k -> 16 k, where sums cross a binade and are rounded) the host
// runs the recurrence once in double, as index bookkeeping, and the kernel reads the (pos, frac) table.
// How many outputs libsamplerate emits depends on its ring-buffer refills and on a termination test evaluated in buffer
// coordinates in double; plan() replays exactly that index state machine (refill to refill, not sample by sample).

#include "kernels.h"
#include <cmath>
#include <cstring>
#include <algorithm>

extern "C" {
extern const unsigned char wmi_sinc_fastest_bin[];
extern const unsigned char wmi_sinc_medium_bin[];
}

namespace wmi { namespace k {

namespace {

struct ResampleArgs {
    const float * in; long long n_in;
    float * out; long long n_out;
    const float * coeffs; int half_len;
    double float_inc, out_scale;
    int increment;
    unsigned long long m; int sh; double frac_scale;          // closed form: position n = (n * m) >> sh, fraction = low bits * 2^-sh
    const int * pos_tab; const double * frac_tab;             // or the host's recurrence
};

__global__ __launch_bounds__(256) void k_resample(const ResampleArgs a) {
    const long long n = (long long) blockIdx.x * 256 + threadIdx.x;
    if (n >= a.n_out) return;
    long long pos; double frac;
    if (a.pos_tab) { pos = a.pos_tab[n]; frac = a.frac_tab[n]; }
    else {
        const unsigned long long lo = (unsigned long long) n * a.m, hi = __umul64hi((unsigned long long) n, a.m);
        if (a.sh == 0) { pos = (long long) lo; frac = 0.0; }
        else {
            pos = (long long) ((hi << (64 - a.sh)) | (lo >> a.sh));
            frac = (double) (lo & ((1ull << a.sh) - 1ull)) * a.frac_scale;
        }
    }
    const int increment = a.increment;
    const int start_index = (int) __builtin_rint(frac * a.float_inc * 4096.0);     // double_to_fp, src_sinc.c:71-74
    const int max_index = a.half_len << 12;
    const long long last = a.n_in - 1;
    const float * __restrict__ c = a.coeffs;
    const float * __restrict__ x = a.in;

    // left half: taps from the far end towards the centre sample, src_sinc.c:293-314
    int fi = start_index;
    int cnt = (max_index - fi) / increment;
    fi += cnt * increment;
    long long di = pos - cnt;
    double left = 0.0;
    do {
        const double fraction = (double) (fi & 4095) * (1.0 / 4096.0);
        const int ix = fi >> 12;
        const float c0 = c[ix], c1 = c[ix + 1];
        const double ic = (double) c0 + fraction * (double) (c1 - c0);
        const long long dc = di < 0 ? 0 : (di > last ? last : di);
        float v = x[dc];
        v = (di < 0 || di > last) ? 0.0f : v;                 // zero history before the first sample, zero tail after the last
        left += ic * (double) v;
        fi -= increment;
        di += 1;
    } while (fi >= 0);

    // right half, src_sinc.c:316-334
    fi = increment - start_index;
    cnt = (max_index - fi) / increment;
    fi += cnt * increment;
    di = pos + 1 + cnt;
    double right = 0.0;
    do {
        const double fraction = (double) (fi & 4095) * (1.0 / 4096.0);
        const int ix = fi >> 12;
        const float c0 = c[ix], c1 = c[ix + 1];
        const double ic = (double) c0 + fraction * (double) (c1 - c0);
        const long long dc = di > last ? last : di;
        float v = x[dc];
        v = di > last ? 0.0f : v;
        right += ic * (double) v;
        fi -= increment;
        di -= 1;
    } while (fi > 0);

    a.out[n] = (float) (a.out_scale * (left + right));
}

double frac_one(double x) {                                   // thirdparty/libsamplerate/src/common.h:149-158
    const double r = x - (double) lrint(x);
    return r < 0.0 ? r + 1.0 : r;
}

}  // namespace

bool sinc_table(int converter, const float ** coeffs, int * count, int * increment) {
    const unsigned char * blob = converter == 2 ? wmi_sinc_fastest_bin : converter == 1 ? wmi_sinc_medium_bin : nullptr;
    if (!blob) return false;                                  // SRC_SINC_BEST_QUALITY: its table is a missing blob of the reference checkout
    int32_t hdr[2];
    memcpy(hdr, blob, 8);
    *increment = hdr[0]; *count = hdr[1];
    *coeffs = (const float *) (blob + 8);
    return true;
}

// Positions of the outputs: closed form when the recurrence is exact, else the recurrence itself.
struct Stepper {
    bool closed = false;
    unsigned long long m = 0; int sh = 0;
    std::vector<int> pos_tab; std::vector<double> frac_tab;

    void init(double inc, long long cap) {
        int e = 0;
        const double fr = frexp(inc, &e);                                        // inc = fr * 2^e, fr in [0.5, 1)
        unsigned long long mant = (unsigned long long) ldexp(fr, 53);            // 53-bit integer
        int q = e - 53;
        while ((mant & 1ull) == 0) { mant >>= 1; q += 1; }
        closed = q >= -53 && q <= 10 && inc + 1.0 <= ldexp(1.0, q + 53);
        if (closed) {
            if (q > 0) { mant <<= q; q = 0; }
            m = mant; sh = -q;
            return;
        }
        pos_tab.resize((size_t) cap + 1); frac_tab.resize((size_t) cap + 1);
        double x = 0.0; long long pos = 0;
        for (long long n = 0; n <= cap; ++n) {                                   // src_sinc.c:411-416, literally
            pos_tab[(size_t) n] = (int) pos; frac_tab[(size_t) n] = x;
            x += inc;
            const double rem = frac_one(x);
            pos += lrint(x - rem);
            x = rem;
        }
    }
    void at(long long n, long long * pos, double * frac) const {
        if (!closed) { *pos = pos_tab[(size_t) n]; *frac = frac_tab[(size_t) n]; return; }
        const unsigned __int128 t = (unsigned __int128) (unsigned long long) n * m;
        *pos = (long long) (t >> sh);
        *frac = sh == 0 ? 0.0 : ldexp((double) (unsigned long long) (t & (((unsigned __int128) 1 << sh) - 1)), -sh);
    }
    // smallest k in [lo, cap] with pos_k >= target (cap if none)
    long long first_at_or_past(long long target, long long lo, long long cap) const {
        long long hi = cap;
        while (lo < hi) {
            const long long mid = lo + (hi - lo) / 2;
            long long p; double f; at(mid, &p, &f);
            if (p >= target) hi = mid; else lo = mid + 1;
        }
        return lo;
    }
};

// libsamplerate's ring-buffer index state machine without the data: how many frames src_simple emits (and consumes).
// Returns the frame count, or < 0: -6 ratio out of range (SRC_ERR_BAD_SRC_RATIO), -21 internal length check
// (SRC_ERR_SINC_PREPARE_DATA_BAD_LEN), -30 ratio beyond what the zero tail of the ring buffer covers.
static long long plan(long long N, long long cap, double ratio, int half_len, int index_inc, const Stepper & P, int hl, int b_len,
                      long long * used) {
    int b_current = 0, b_end = 0, b_real_end = -1;
    long long in_used = 0, n = 0, pos_n = 0;
    const double terminate = 1.0 / ratio + 1e-20;
    auto refill = [&]() -> int {                                                 // src_sinc.c:1166-1239
        if (b_real_end >= 0) return 0;
        int len;
        if (b_current == 0) { len = b_len - 2 * hl; b_current = b_end = hl; }
        else if (b_end + hl + 1 < b_len) len = std::max(b_len - b_current - hl, 0);
        else {
            len = b_end - b_current;
            b_current = hl; b_end = hl + len;
            len = std::max(b_len - b_current - hl, 0);
        }
        len = (int) std::min<long long>(N - in_used, len);
        if (len < 0 || b_end + len > b_len) return -21;
        b_end += len; in_used += len;
        if (in_used == N && b_end - b_current < 2 * hl) {
            if (b_len - b_end < hl + 5) { len = b_end - b_current; b_current = hl; b_end = hl + len; }
            b_real_end = b_end;
            if (b_end + hl + 5 > b_len) return -30;                              // the reference would run its right taps into stale samples
            b_end += hl + 5;
        }
        return 0;
    };
    while (n < cap) {
        if (b_end - b_current <= hl) {
            const int e = refill();
            if (e) return e;
            if (b_end - b_current <= hl) break;
        }
        const int avail = b_end - b_current - hl;                                // outputs go on while the position has advanced by less
        long long stop = P.first_at_or_past(pos_n + avail, n + 1, cap);
        if (b_real_end >= 0) {                                                   // termination test per output, src_sinc.c:389-393
            auto terminated = [&](long long k) {
                long long p; double f; P.at(k, &p, &f);
                const int bc = b_current + (int) (p - pos_n);
                return bc + f + terminate > b_real_end;
            };
            long long lo = n, hi = stop;                                         // first k in [n, stop) that terminates (stop if none)
            while (lo < hi) { const long long mid = lo + (hi - lo) / 2; if (terminated(mid)) hi = mid; else lo = mid + 1; }
            if (lo < stop) { n = lo; break; }
        }
        long long p; double f; P.at(stop, &p, &f);
        b_current += (int) (p - pos_n);
        pos_n = p; n = stop;
    }
    if (used) *used = in_used;
    return n;
}

// One call of src_simple on the device.  d_in / d_out: device pointers (n_in frames in, room for out_cap frames out).
// d_coeffs: the converter's table on the device.  d_pos / d_frac: device scratch for the (pos, frac) table, filled here when
// the recurrence is not exact (may be null when `need_table` comes back false from resample_plan).
ResamplePlan resample_plan(long long n_in, long long out_cap, double ratio, int converter) {
    ResamplePlan pl;
    const float * coeffs; int count, table_inc;
    if (!sinc_table(converter, &coeffs, &count, &table_inc)) { pl.error = -10; return pl; }     // SRC_ERR_BAD_CONVERTER
    if (ratio < 1.0 / 256 || ratio > 256.0) { pl.error = -6; return pl; }
    if (n_in < 0) n_in = 0;
    if (out_cap < 0) out_cap = 0;
    pl.half_len = count - 2; pl.index_inc = table_inc;
    int b_len = 3 * (int) lrint((pl.half_len + 2.0) / table_inc * 256 + 1);                    // src_sinc.c:213-216
    b_len = std::max(b_len, 4096) + 1;
    double cnt = (pl.half_len + 2.0) / table_inc;
    if (ratio < 1.0) cnt /= ratio;
    const int hl = (int) (lrint(cnt) + 1);
    pl.float_inc = table_inc * (ratio < 1.0 ? ratio : 1.0);
    pl.increment = (int) lrint(pl.float_inc * 4096.0);
    pl.out_scale = pl.float_inc / table_inc;
    pl.stepper = std::make_shared<Stepper>();
    pl.stepper->init(1.0 / ratio, out_cap);
    long long used = 0;
    const long long gen = plan(n_in, out_cap, ratio, pl.half_len, table_inc, *pl.stepper, hl, b_len, &used);
    if (gen < 0) { pl.error = (int) gen; return pl; }
    pl.n_out = gen; pl.n_used = used; pl.need_table = !pl.stepper->closed;
    return pl;
}

void resample_table(const ResamplePlan & pl, const int ** pos, const double ** frac) {
    *pos = pl.stepper->pos_tab.data(); *frac = pl.stepper->frac_tab.data();
}

void resample_positions(const ResamplePlan & pl, long long n, long long * pos, double * frac) {
    for (long long i = 0; i < n; ++i) pl.stepper->at(i, pos + i, frac + i);
}

void resample_launch(const ResamplePlan & pl, const float * d_in, long long n_in, float * d_out, const float * d_coeffs,
                     const int * d_pos, const double * d_frac, hipStream_t st) {
    if (pl.n_out <= 0) return;
    ResampleArgs a;
    a.in = d_in; a.n_in = n_in; a.out = d_out; a.n_out = pl.n_out;
    a.coeffs = d_coeffs; a.half_len = pl.half_len;
    a.float_inc = pl.float_inc; a.out_scale = pl.out_scale; a.increment = pl.increment;
    a.m = pl.stepper->m; a.sh = pl.stepper->sh; a.frac_scale = ldexp(1.0, -pl.stepper->sh);
    a.pos_tab = pl.need_table ? d_pos : nullptr; a.frac_tab = pl.need_table ? d_frac : nullptr;
    hipLaunchKernelGGL(k_resample, dim3((unsigned) ((pl.n_out + 255) / 256)), dim3(256), 0, st, a);
}

}}  // namespace wmi::k
